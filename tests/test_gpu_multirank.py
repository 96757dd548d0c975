"""Two ranks on one GPU (gloo): sharded runs equal single-rank runs bit for bit.

The ranks are separate processes started the way a launcher starts them; each
loads librtmc.so and traces its share of the photon ids on GPU 0, then the
integer tallies are all-reduced.  (The RCCL path differs only in the transport
of that one sum.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from raytracer_amd import _abi
from test_distributed_cpu import REPO, _run_ranks

pytestmark = pytest.mark.gpu

KEYS = list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]


def test_two_rank_runsimulation_equals_one_rank(tmp_path):
    n = 300001
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _run_ranks(2, ["C3", str(n), "21", two])
    _run_ranks(1, ["C3", str(n), "21", one])
    a, b = np.load(one), np.load(two)
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert int(b["n_photons"]) == n
    # and both equal a plain rtmc_run over the id range in this process
    s = H.raytrace_for("C3").Scene()
    try:
        t = s.run(n, seed=21)
    finally:
        s.close()
    for k in KEYS:
        assert np.array_equal(np.asarray(t[k]), b[k]), k


def test_bench_launcher_two_ranks(tmp_path):
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself
    and reports them; the tallies it summed over ranks and steps are those of
    one rtmc_run over the same photon ids."""
    dump = str(tmp_path / "bench.npz")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--device", "0", "--steps", "2", "--warmup", "1", "--photons-per-step", "250000",
                          "--no-cpu-baseline", "--dump-tallies", dump],
                         env=env, stdout=subprocess.PIPE, timeout=300, check=True)
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert line["unit"] == "photons/s" and line["value"] > 0
    d = np.load(dump)
    first, n = int(d["first_photon"]), int(d["n_requested"])
    assert int(d["n_photons"]) == n
    assert (first, n) == (2 * 250000, 2 * 2 * 250000)
    s = H.raytrace_for("C2").Scene()
    try:
        t = s.run(n, seed=1, first_photon=first)
    finally:
        s.close()
    for k in KEYS:
        assert np.array_equal(np.asarray(t[k]), d[k]), k


def test_rccl_path_with_one_rank(tmp_path):
    """The launch line the driver uses (`python -m torch.distributed.run ...
    bench.py --gpus N`) with N = 1 and the default backend: the process group
    is created on RCCL (backend "nccl") and the tally block goes through its
    all-reduce on the device.  One rank is all a one-GPU box can hold -- RCCL
    refuses two ranks on one device -- but init, barrier, all-reduce and
    teardown are the calls N ranks make."""
    dump = str(tmp_path / "rccl.npz")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29613",
                          os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--photons-per-step", "300000", "--no-cpu-baseline", "--dump-tallies", dump],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    d = np.load(dump)
    s = H.raytrace_for("C2").Scene()
    try:
        t = s.run(int(d["n_requested"]), seed=1, first_photon=int(d["first_photon"]))
    finally:
        s.close()
    for k in KEYS:
        assert np.array_equal(np.asarray(t[k]), d[k]), k


def test_strong_scaling_mode_two_ranks(tmp_path):
    """`bench.py --total-photons N`: a step is N photon ids cut into one block per rank
    (`distributed.shard_range`).  Two gloo ranks on GPU 0, an odd N: the line says "strong", names
    the backend and the number of ranks the process group really had, and the summed tallies are
    those of one rtmc_run over the same ids."""
    dump = str(tmp_path / "strong.npz")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--device", "0", "--steps", "2", "--warmup", "1", "--total-photons", "400001",
                          "--config", "C4", "--no-cpu-baseline", "--dump-tallies", dump],
                         env=env, stdout=subprocess.PIPE, timeout=300, check=True)
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["collective"]["backend"] == "gloo" and line["collective"]["world_size"] == 2
    assert line["collective"]["payload_bytes"] > 8 * 3800
    assert line["config"]["photons_per_step_total"] == 400001
    d = np.load(dump)
    first, n = int(d["first_photon"]), int(d["n_requested"])
    assert (first, n) == (400001, 2 * 400001) and int(d["n_photons"]) == n
    s = H.raytrace_for("C4").Scene()
    try:
        t = s.run(n, seed=1, first_photon=first)
    finally:
        s.close()
    for k in KEYS:
        assert np.array_equal(np.asarray(t[k]), d[k]), k


def test_one_process_several_devices():
    """RayTrace.Devices = [0, 0]: two host threads, each tracing half of the ids through rtmc_run
    (here on the same GPU, where the library runs them one after the other), summed on the host:
    the tallies of one call over all ids, and the element objects carry them."""
    rt = H.raytrace_for("C3")
    rt.NumberOfPhotons, rt.Seed = 250001, 9
    whole = rt.runSimulation()
    rt.Devices = [0, 0]
    split = rt.runSimulation()
    for k in KEYS:
        assert np.array_equal(np.asarray(whole[k]), np.asarray(split[k])), k
    assert rt.OpticalElements[2].Type.Counter == int(whole["counter"][2])
    rt.Devices = [0, 0, 0]
    rt.NumberOfPhotons = 2          # fewer photons than threads
    assert rt.runSimulation()["n_photons"] == 2
