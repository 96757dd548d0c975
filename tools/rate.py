"""Kernel photons/s of the named parity configurations on one GPU, without the
checker: the quick A/B companion of results_table.py (RTMC_LIBRARY selects the
build).

usage: python tools/rate.py [photons] [config | RB:<seed> | twin ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50000000
names = sys.argv[2:] or ["C2", "C2r", "C5", "C7", "C8"]
out = []
def scene_of(name):
    """A parity configuration, or RB:<seed> (configs.random_boxes_job) / twin (configs.twin_job)."""
    from raytracer_amd import RayTrace, configs
    if name.startswith("RB:") or name == "twin":
        rt = RayTrace.RayTrace(demo_scene=False)
        rt.LoadJob(configs.twin_job() if name == "twin" else configs.random_boxes_job(int(name[3:])))
        return rt.Scene()
    return H.raytrace_for(name).Scene()


for name in names:
    s = scene_of(name)
    m = n // 50 if name in ("C1_open", "C8") else n
    s.run(min(m, 1000000), seed=3)
    best = 1e30
    for rep in range(3):
        s.run(m, seed=7 + rep)
        best = min(best, s.last_kernel_ms())
    out.append("%s %.4g" % (name, m / (best * 1e-3)))
print(os.environ.get("RTMC_LIBRARY", "default"), "  ".join(out), flush=True)
