"""Kernel time vs photon count (fresh process)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import RayTrace, configs
cfg = sys.argv[1]
rt = RayTrace.RayTrace(demo_scene=False)
rt.LoadJob(configs.make_job(cfg, 10))
s = rt.Scene()
s.run(1000000, seed=3)
for n in [1000000, 5000000, 25000000, 50000000, 100000000, 200000000]:
    ts = []
    for i in range(3):
        t = s.run(n, seed=3)
        ts.append(s.last_kernel_ms())
    print("n=%10d  %.3f ms  %.3g photons/s  iters/photon %.2f" % (n, min(ts), n / min(ts) * 1e3, t["n_steps"] / n))
