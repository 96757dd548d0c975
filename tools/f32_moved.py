"""How many of N photons change fate between the float32 and the float64
kernels on the same variates (RTMC_LIBRARY selects the build).
usage: python tools/f32_moved.py [photons] [config ...]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H
from raytracer_amd import _abi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5000000
out = []
for name in sys.argv[2:] or ["C1", "C2", "C4", "C2r"]:
    s = H.raytrace_for(name).Scene()
    a = s.run(n, seed=17, precision=_abi.F32)
    b = s.run(n, seed=17, precision=_abi.F64)
    moved = int(np.abs(a["counter"].astype(np.int64) - b["counter"].astype(np.int64)).sum())
    moved += abs(int(a["n_internal_converted"]) - int(b["n_internal_converted"]))
    out.append("%s %d" % (name, moved))
    s.close()
print(os.path.basename(os.environ.get("RTMC_LIBRARY", "default")), "moved of %d:" % n, "  ".join(out), flush=True)
