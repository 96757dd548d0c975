"""One random scene of tests/test_gpu_fuzz.py: float64 kernels against the CPU checker, first photons
that differ and their step-by-step traces (debug aid).  usage: python tools/dbg_fuzz.py SEED [n]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_fuzz import random_job  # noqa: E402
from raytracer_amd import RayTrace, _abi  # noqa: E402
from raytracer_amd import scene as sc  # noqa: E402
from oracle import oracle  # noqa: E402

seed = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rt = RayTrace.RayTrace(demo_scene=False)
rt.LoadJob(random_job(seed))
for i, oe in enumerate(rt.OpticalElements):
    print(i, oe.Geometry.kind, oe.Type.kind, np.round(oe.Geometry.SupportVector, 3), np.round(oe.Geometry.DirectionVector, 3))
s = rt.Scene()
o = oracle.OracleScene.from_raytrace(rt)
run_seed = 5 + seed
g = s.run(n, seed=run_seed, precision=_abi.F64)
c = o.run(n, seed=run_seed)
for k in ("counter", "bounce", "spectral"):
    print(k, "gpu", g[k].tolist())
    print(k, "cpu", c[k].tolist())
for k in ("n_steps", "n_absorbed_once", "n_absorptions", "n_internal_converted", "n_dumped", "n_capped", "n_lost", "n_stale_steps"):
    print(k, g[k], c[k])
shown = 0
for i in range(n):
    g1 = s.run(1, seed=run_seed, first_photon=i, precision=_abi.F64)
    c1 = o.run(1, seed=run_seed, first_photon=i)
    if g1["n_steps"] != c1["n_steps"] or not np.array_equal(g1["counter"], c1["counter"]) or not np.array_equal(g1["bounce"], c1["bounce"]):
        print("id", i, "gpu steps", g1["n_steps"], "counter", np.nonzero(g1["counter"])[0], "bounce", np.nonzero(g1["bounce"])[0], g1["bounce"][np.nonzero(g1["bounce"])[0]],
              "| cpu steps", c1["n_steps"], "counter", np.nonzero(c1["counter"])[0], "bounce", np.nonzero(c1["bounce"])[0], c1["bounce"][np.nonzero(c1["bounce"])[0]], flush=True)
        shown += 1
        if shown >= 6:
            break
