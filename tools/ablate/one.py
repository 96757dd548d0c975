"""Time one build of librtmc.so (RTMC_LIBRARY) in a fresh process."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import RayTrace, configs
n = int(float(sys.argv[1])); cfg = sys.argv[2]
rt = RayTrace.RayTrace(demo_scene=False)
rt.LoadJob(configs.make_job(cfg, n))
s = rt.Scene()
ts = []
for i in range(5):
    s.run(n, seed=3)
    ts.append(s.last_kernel_ms())
print("%-40s min %.2f median %.2f ms" % (os.environ.get("RTMC_LIBRARY", "default"), min(ts[1:]), float(np.median(ts[1:]))))
