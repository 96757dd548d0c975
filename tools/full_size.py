"""The BASELINE configurations at their full sizes on one GPU: C3 (co-doped slab, donor->acceptor
transfer) and C4 (stacked slabs with reflective edges) with 1e10 photons, C5 (sun-spectrum source,
per-edge emission spectra) with 1e11, each in one call; prints throughput, fates and the edge
efficiency beside the reference's runs.
usage: python tools/full_size.py  (on the GPU box)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402

for name, n in (("C3", 10 ** 10), ("C4", 10 ** 10), ("C5", 10 ** 11)):
    rt = H.raytrace_for(name)
    s = rt.Scene()
    s.run(10 ** 6, seed=1)
    t0 = time.perf_counter()
    t = s.run(n, seed=2)
    wall = time.perf_counter() - t0
    fates = t["n_dumped"] + t["n_internal_converted"] + t["n_capped"] + t["n_lost"]
    edges = [2, 8, 14] if name == "C4" else [2, 3, 4, 5]      # C4: the one detector edge of each of its three slabs
    eta = float(t["counter"][edges].sum()) / n
    ref = H.ref_tallies(name)
    n_ref = sum(r["n"] for r in ref)
    eta_ref = float(np.sum([r["counter"] for r in ref], axis=0)[edges].sum()) / n_ref
    sig = (eta * (1 - eta) / n_ref) ** 0.5
    print("%s: %.0e photons in %.2f s (kernel %.2f s) = %.3e photons/s; fates add up: %s; spectra add up: %s; "
          "iterations/photon %.3f; edge efficiency %.5f, reference %.5f +- %.5f (%.2f sigma)"
          % (name, n, wall, s.last_kernel_ms() * 1e-3, n / wall, fates == n,
             bool(np.array_equal(t["spectrum"].sum(axis=1), t["counter"])), t["n_steps"] / n, eta, eta_ref, sig,
             (eta - eta_ref) / sig), flush=True)
    s.close()
