"""world_size-2 gloo run of the tally exchange: two ranks fill disjoint
blocks, all-reduce them, and both must hold the exact integer sum."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_el, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from raytracer_amd import _abi, distributed
    from raytracer_amd import scene as sc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert distributed.rank_world() == (rank, world)
    first, count = distributed.shard_range(1001, rank, world)
    block = sc.new_tallies(n_el)
    t, arrays = block
    rng = np.random.RandomState(100 + rank)
    for k in _abi.TALLY_ARRAYS + ["spectrum"]:
        arrays[k][:] = rng.randint(0, 2**40, size=arrays[k].shape).astype(np.uint64)
    arrays["energy_fx"][0] = np.uint64(2**63 + 5 + rank)      # exercises the sign bit
    t.n_photons = count
    t.n_steps = 7 * count + rank
    seed = distributed.shared_entropy()
    distributed.allreduce_tallies(block)
    d = sc.tallies_to_dict(*block)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), seed=seed, n_photons=d["n_photons"],
             n_steps=d["n_steps"], **{k: d[k] for k in _abi.TALLY_ARRAYS + ["spectrum"]})
    dist.destroy_process_group()


def test_allreduce_tallies_gloo_world2(tmp_path):
    world, n_el = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), n_el, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    from raytracer_amd import _abi
    expect = {}
    for k in _abi.TALLY_ARRAYS + ["spectrum"]:
        total = None
        for rank in range(world):
            rng_vals = {}
            rng = np.random.RandomState(100 + rank)
            for kk in _abi.TALLY_ARRAYS + ["spectrum"]:
                shape = (n_el,) if kk != "spectrum" else (n_el * _abi.SPECTRUM_BINS,)
                rng_vals[kk] = rng.randint(0, 2**40, size=shape).astype(np.uint64)
            rng_vals["energy_fx"][0] = np.uint64(2**63 + 5 + rank)
            total = rng_vals[k] if total is None else total + rng_vals[k]
        expect[k] = total
    for k in _abi.TALLY_ARRAYS + ["spectrum"]:
        assert np.array_equal(r0[k].reshape(-1), expect[k]), k
        assert np.array_equal(r0[k], r1[k]), k
    assert int(r0["n_photons"]) == 1001 and int(r1["n_photons"]) == 1001
    assert int(r0["n_steps"]) == 7 * 1001 + 1
    assert int(r0["seed"]) == int(r1["seed"])


def _run_ranks(world, argv, extra_env=None, timeout=300):
    """Start `world` processes of tests/dist_worker.py the way a launcher does."""
    import subprocess
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dist_worker.py")] + argv,
                                      env=env))
    codes = [p.wait(timeout=timeout) for p in procs]
    assert codes == [0] * world, codes


def test_sharded_runsimulation_equals_single_rank(tmp_path):
    """RayTrace.runSimulation under two gloo ranks: the ranks take the two
    halves of the id range (the odd photon goes to rank 0), agree on one seed,
    and the all-reduced tallies are exactly those of one rank tracing the whole
    range.  The scene is a stand-in whose tallies are a known function of the
    photon ids, so the test runs without a GPU and checks the host logic:
    shard_range, the seed broadcast, allreduce_tallies, _publish."""
    from raytracer_amd import _abi
    n = 100001
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _run_ranks(1, ["C2", str(n), "5", one, "fake"])
    _run_ranks(2, ["C2", str(n), "5", two, "fake"])
    a, b = np.load(one), np.load(two)
    for k in list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]:
        assert np.array_equal(a[k], b[k]), k
    assert int(a["n_photons"]) == n and int(a["counter"].sum()) == n
    r0, r1 = np.load(two + ".rank0.npz"), np.load(two + ".rank1.npz")
    assert (int(r0["first"]), int(r0["count"])) == (0, 50001)
    assert (int(r1["first"]), int(r1["count"])) == (50001, 50000)
    # no seed given: every rank must still use the same one
    three = str(tmp_path / "three.npz")
    _run_ranks(2, ["C2", "1000", "-1", three, "fake"])
    assert int(np.load(three + ".rank0.npz")["seed"]) == int(np.load(three + ".rank1.npz")["seed"])


def test_shard_range_covers_every_id_once():
    from raytracer_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                first, count = shard_range(n, r, world)
                assert first == nxt and count >= 0
                nxt = first + count
            assert nxt == n


def test_bench_launcher_starts_the_ranks_before_touching_a_gpu():
    """`bench.py --gpus N` without WORLD_SIZE must hand the work to N child
    processes; the parent itself imports neither torch nor the HIP library."""
    import ast
    src = open(os.path.join(REPO, "bench.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "launch_ranks")
    names = {n.names[0].name.split(".")[0] for n in ast.walk(fn) if isinstance(n, (ast.Import, ast.ImportFrom))
             for _ in [0] if isinstance(n, ast.Import)}
    names |= {n.module.split(".")[0] for n in ast.walk(fn) if isinstance(n, ast.ImportFrom) and n.module}
    assert "torch" not in names and "raytracer_amd" not in names, names
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    order = [getattr(n, "lineno") for n in ast.walk(main)
             if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "launch_ranks"]
    first_torch = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Import) and n.names[0].name.startswith("torch"))
    assert order and order[0] < first_torch


def test_devices_list_shards_over_host_threads(monkeypatch):
    """RayTrace.Devices = [...]: the ids are cut into one block per entry, each traced by its own host
    thread on its own device, and the blocks add up to the tallies of one call (stand-in scene, no GPU)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import dist_worker
    from raytracer_amd import RayTrace, _abi, configs
    from raytracer_amd import scene as sc
    monkeypatch.setattr(sc, "Scene", dist_worker.FakeScene)
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job("C2", 100003))
    rt.Seed, rt.Device = 3, 0
    whole = rt.runSimulation()
    rt.Devices = [0, 1, 2]
    rt.Scene().calls.clear()
    split = rt.runSimulation()
    for k in list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]:
        assert np.array_equal(np.asarray(whole[k]), np.asarray(split[k])), k
    calls = sorted(rt.Scene().calls)
    assert [c[:2] for c in calls] == [(0, 33335), (33335, 33334), (66669, 33334)] and {c[2] for c in calls} == {3}


def test_bench_line_describes_the_exchange():
    """The record of the collective comes from torch.distributed itself; without a process group it says so."""
    from raytracer_amd import distributed
    from raytracer_amd import scene as sc
    info = distributed.collective_info(sc.new_tallies(4))
    assert info["backend"] is None and info["world_size"] == 1
    assert info["payload_bytes"] == 8 * (9 + 4 * 4 + 4 * 3800)
