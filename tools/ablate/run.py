"""Time ablated builds of the trace kernel on the same workload (interleaved rounds)."""
import sys, os, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import RayTrace, configs, _abi, scene as sc
here = os.path.dirname(os.path.abspath(__file__))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
scenes = {}
for f in sorted(glob.glob(os.path.join(here, "lib_*.so"))):
    lib = _abi.load_library(f)
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job(cfg, n))
    scenes[os.path.basename(f)] = sc.Scene(rt.OpticalElements, rt.Photon, rt.MaterialDict, lib=lib)
times = {k: [] for k in scenes}
for rnd in range(4):
    for k, s in scenes.items():
        t = s.run(n, seed=3)
        times[k].append(s.last_kernel_ms())
for k, v in times.items():
    print("%-16s min %.2f ms  median %.2f ms" % (k, min(v[1:]), float(np.median(v[1:]))))
