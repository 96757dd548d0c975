"""Shared helpers of the parity tests."""
import json
import os

import numpy as np

from raytracer_amd import RayTrace, configs
from raytracer_amd import scene as sc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACE_CONFIGS = ["C1_open", "C1", "C2", "C3", "C4", "C5", "C6", "C7", "C8", "C9", "C2r"]

# columns of a golden trace row (tests/golden/make_golden.py ROW_FIELDS)
G_L, G_D, G_E = slice(0, 3), slice(3, 6), slice(6, 9)
G_WL, G_DIST, G_STATUS, G_MAT, G_HIT, G_HITD, G_TAPE = 9, 10, 11, 12, 13, 14, 18
# columns of an rtmc trace row (include/rtmc.h)
T_WL, T_DIST, T_STATUS, T_MAT, T_HIT, T_HITD, T_TAPE = 9, 10, 11, 12, 13, 14, 15


def raytrace_for(name, n_photons=10):
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job(name, n_photons))
    return rt


def load_trace(name):
    g = np.load(os.path.join(GOLDEN, "trace_%s.npz" % name))
    return {k: g[k] for k in g.files}


def replay(scene_like, g, **kw):
    """Launch one photon per tape segment and trace as many iterations as the
    longest stored golden photon; returns (trace, batch)."""
    n = len(g["n_iter"])
    ro = g["row_off"]
    n_steps = int((ro[1:] - ro[:-1]).max())
    batch = sc.new_photon_batch(n)
    tr = scene_like.step_tape(batch, g["tape"], g["tape_off"], n_steps, launch=True, trace=True,
                              **kw)
    return tr, batch


def compare_trace(g, tr, rtol, absorber_mats, atol_len=0.0, max_rows=None):
    """Max relative deviation of (L, D, E, wavelength, dist_left) and exact
    agreement of status / material / hit element / tape position, per photon.
    dist_left is only compared inside absorbers: in other media the reference
    carries a stale value over from the previous photon.  `max_rows` limits the
    comparison to the first iterations of every photon."""
    rows, ro = g["rows"], g["row_off"]
    worst = 0.0
    mismatched = []
    for k in range(len(g["n_iter"])):
        r = rows[ro[k]:ro[k + 1]][:max_rows]
        t = tr[k, :len(r)]
        ok = (np.array_equal(r[:, G_STATUS], t[:, T_STATUS])
              and np.array_equal(r[:, G_MAT], t[:, T_MAT])
              and np.array_equal(r[:, G_HIT], t[:, T_HIT])
              and np.array_equal(r[:, G_TAPE] - g["tape_off"][k], t[:, T_TAPE]))
        if not ok:
            mismatched.append(k)
            continue
        ref = np.concatenate([r[:, 0:9], r[:, [G_WL]]], axis=1)
        got = np.concatenate([t[:, 0:9], t[:, [T_WL]]], axis=1)
        err = np.abs(ref - got) / np.maximum(np.abs(ref), 1.0)
        in_abs = np.isin(r[:, G_MAT], absorber_mats)
        if in_abs.any():
            # free paths beyond 1e30 cm (vanishing absorption) are "never": float32 saturates there
            dref = np.minimum(r[in_abs, G_DIST], 1e30)
            dgot = np.minimum(t[in_abs, T_DIST], 1e30)
            derr = np.abs(dref - dgot) / np.maximum(np.abs(dref), 1.0)
            err = np.concatenate([err.reshape(-1), derr])
        e = float(np.max(err))
        if not np.isfinite(e) or e > rtol:
            mismatched.append(k)
        else:
            worst = max(worst, e)
    return worst, mismatched


def ref_tallies(config):
    with open(os.path.join(GOLDEN, "ref_tallies.json")) as fh:
        return [r for r in json.load(fh) if r["config"] == config]
