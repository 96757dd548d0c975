"""Find photon ids whose fate differs between the float64 kernels and the CPU checker (debug aid).
usage: python tools/dbg_diff_ids.py CONFIG [n] [seed]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402
from raytracer_amd import _abi  # noqa: E402
from oracle import oracle  # noqa: E402

name = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
rt = H.raytrace_for(name)
s = rt.Scene()
o = oracle.OracleScene.from_raytrace(rt)
KEYS = ("counter", "bounce", "spectral")


def same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in KEYS) and a["n_steps"] == b["n_steps"]


block = 2000
bad = []
for first in range(0, n, block):
    g = s.run(block, seed=seed, first_photon=first, precision=_abi.F64)
    c = o.run(block, seed=seed, first_photon=first)
    if not same(g, c):
        for i in range(first, first + block):
            g1 = s.run(1, seed=seed, first_photon=i, precision=_abi.F64)
            c1 = o.run(1, seed=seed, first_photon=i)
            if not same(g1, c1):
                bad.append(i)
                print("id", i, "gpu: steps", g1["n_steps"], "counter", np.nonzero(g1["counter"])[0], "bounce", g1["bounce"].tolist(),
                      "| cpu: steps", c1["n_steps"], "counter", np.nonzero(c1["counter"])[0], "bounce", c1["bounce"].tolist(), flush=True)
print("differing ids:", bad)
