"""Debug aid: print where the float32 tape replay leaves the golden trace."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import helpers as H
from raytracer_amd import _abi
np.set_printoptions(linewidth=200, precision=9)
name = sys.argv[1]
prec = _abi.F32 if len(sys.argv) < 3 or sys.argv[2] == "f32" else _abi.F64
_abi.check = lambda rc, lib=None: None   # keep the trace of a replay that ran out of tape
import raytracer_amd.scene as _sc
g = H.load_trace(name)
rt = H.raytrace_for(name)
s = rt.Scene()
tr, batch = H.replay(s, g, precision=prec)
rows, ro = g["rows"], g["row_off"]
shown = 0
for k in range(len(g["n_iter"])):
    r = rows[ro[k]:ro[k + 1]]
    t = tr[k, :len(r)]
    disc = (r[:, 11] != t[:, 11]) | (r[:, 12] != t[:, 12]) | (r[:, 13] != t[:, 13]) | ((r[:, 18] - g["tape_off"][k]) != t[:, 15])
    err = np.abs(r[:, 0:9] - t[:, 0:9]).max(axis=1)
    bad = np.where(disc | ~(err < 2e-4))[0]
    if len(bad):
        i = bad[0]
        print("photon", k, "n_iter", len(r), "first bad iter", i, "discrete" if disc[i] else "vector", "err before", err[max(0, i - 1)])
        for j in range(max(0, i - 1), min(len(r), i + 2)):
            print("  ref ", j, r[j, :15], "tp", r[j, 18] - g["tape_off"][k])
            print("  gpu ", j, t[j, :15], "tp", t[j, 15])
        shown += 1
        if shown >= 4:
            break
