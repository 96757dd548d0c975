"""Same seed, same photons: float32 vs float64 kernels, per-detector counts."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import RayTrace, configs, _abi
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100000000
rt = RayTrace.RayTrace(demo_scene=False)
rt.LoadJob(configs.make_job(name, n))
s = rt.Scene()
t0 = time.time(); a = s.run(n, seed=3, precision=_abi.F32); t1 = time.time()
b = s.run(n, seed=3, precision=_abi.F64); t2 = time.time()
print("f32 %.2fs f64 %.2fs" % (t1 - t0, t2 - t1))
print("counter f32", a["counter"])
print("counter f64", b["counter"])
d = a["counter"].astype(np.int64) - b["counter"].astype(np.int64)
print("diff", d, "sigma-ish", np.sqrt(np.maximum(1, b["counter"])).astype(int))
for k in ("n_steps", "n_internal_converted", "n_absorbed_once", "n_absorptions", "n_capped", "n_stale_steps"):
    print(k, a[k], b[k], int(a[k]) - int(b[k]))
print("bounce diff", a["bounce"].astype(np.int64) - b["bounce"].astype(np.int64))
