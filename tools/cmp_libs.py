"""Run the same photons through two builds of librtmc.so and compare tallies."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import RayTrace, configs, _abi, scene as sc
libs = {"cur": _abi.load_library(), "prev": _abi.load_library(sys.argv[1])}
n = int(float(sys.argv[2]))
res = {}
for tag, lib in libs.items():
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job("C2", n))
    s = sc.Scene(rt.OpticalElements, rt.Photon, rt.MaterialDict, lib=lib)
    res[tag] = [s.run(n, seed=3, precision=_abi.F32), s.run(n, seed=3, precision=_abi.F64)]
for i, pr in enumerate(("f32", "f64")):
    a, b = res["cur"][i], res["prev"][i]
    print(pr, "cur ", a["counter"], a["n_internal_converted"], a["n_steps"])
    print(pr, "prev", b["counter"], b["n_internal_converted"], b["n_steps"])
