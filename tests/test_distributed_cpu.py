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
