"""float32 against float64 on one scene of the random-box family (tests/test_gpu_cells.py), as the suite
builds it and unturned, with and without cells: photons that change fate, and iterations.
usage: python tools/debug/rb_moved.py SEED [photons]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from raytracer_amd import RayTrace, _abi, configs  # noqa: E402

seed = int(sys.argv[1])
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3000000


def moved(a, b):
    d = int(np.abs(a["counter"].astype(np.int64) - b["counter"].astype(np.int64)).sum())
    d += sum(abs(int(a[k]) - int(b[k])) for k in ("n_internal_converted", "n_capped", "n_lost"))
    return (d + 1) // 2


def jobs():
    yield "unturned", configs.random_boxes_job(seed, 10)
    rng = np.random.RandomState(500 + seed)
    yield "turned", configs.rotated_job(base="RB", seed=seed, axis=rng.normal(size=3), angle=rng.uniform(0.1, 3.0), n_photons=10)


for label, job in jobs():
    for cells in (True, False):
        if cells:
            os.environ.pop("RTMC_NO_CELLS", None)
        else:
            os.environ["RTMC_NO_CELLS"] = "1"
        rt = RayTrace.RayTrace(demo_scene=False)
        rt.LoadJob(job)
        s = rt.Scene()
        a = s.run(n, seed=100 + seed, precision=_abi.F32)
        b = s.run(n, seed=100 + seed, precision=_abi.F64)
        print("%s, %s: %d of %d photons change fate between float32 and float64; %.2f iterations per photon, stale steps %d / %d"
              % (label, "cells" if cells else "list walk", moved(a, b), n, int(b["n_steps"]) / n, int(a["n_stale_steps"]),
                 int(b["n_stale_steps"])), flush=True)
        s.close()
