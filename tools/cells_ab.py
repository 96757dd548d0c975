"""Tallies and kernel rate of the named configurations with and without the cells of the float32 search
(RTMC_NO_CELLS), same photon ids and seed: what differs and what it costs.

usage: python tools/cells_ab.py [photons] [config ...]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
names = sys.argv[2:] or ["C2", "C3", "C4", "C5", "C1", "C6", "C9"]
KEYS = ("counter", "bounce", "spectral", "energy_fx")
SCALARS = ("n_photons", "n_steps", "n_absorptions", "n_internal_converted", "n_dumped", "n_capped", "n_lost", "n_stale_steps")
for name in names:
    m = n // 50 if name in ("C1_open", "C8") else n
    res = {}
    for mode in ("cells", "list"):
        if mode == "list":
            os.environ["RTMC_NO_CELLS"] = "1"
        else:
            os.environ.pop("RTMC_NO_CELLS", None)
        s = H.raytrace_for(name).Scene()
        s.run(min(m, 1000000), seed=3)
        best = 1e30
        for rep in range(3):
            t = s.run(m, seed=11)
            best = min(best, s.last_kernel_ms())
        res[mode] = (t, m / (best * 1e-3), s.cells())
    a, b = res["cells"][0], res["list"][0]
    dist = sum(int(np.abs(a[k].astype(np.int64) - b[k].astype(np.int64)).sum()) for k in KEYS if k != "energy_fx")
    scal = {k: int(a[k]) - int(b[k]) for k in SCALARS if int(a[k]) != int(b[k])}
    print("%-8s cells %.4g  list %.4g photons/s (%+.1f %%); tally distance %d of %d photons; scalars that differ: %s; %s"
          % (name, res["cells"][1], res["list"][1], 100.0 * (res["cells"][1] / res["list"][1] - 1.0), dist, m, scal,
             res["cells"][2]), flush=True)
