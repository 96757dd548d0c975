"""Looks for a bias of the float32 production kernels far below what the reference's own samples
resolve: 2e9 photons on the GPU against 2e8 photons of the float64 CPU checker drawing where and
how the reference draws (the mode the tape replays pin), per configuration; prints every tally as
a fraction of the photons with its difference in standard deviations.
usage: python tools/bias_check.py [config ...]  (on the GPU box; about 25 s of CPU per configuration)"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402
from oracle import oracle  # noqa: E402

n_gpu, n_cpu = 2 * 10 ** 9, 2 * 10 ** 8
for name in sys.argv[1:] or ["C2", "C3", "C9"]:
    rt = H.raytrace_for(name)
    s = rt.Scene()
    gpu = s.run(n_gpu, seed=31)
    cpu = oracle.OracleScene.from_raytrace(rt).run_threads(n_cpu, seed=32, threads=16, reference_order=True)
    rows = [("detector %d" % i, gpu["counter"][i], cpu["counter"][i]) for i in range(s.n_elements) if cpu["counter"][i] or gpu["counter"][i]]
    rows += [("counted %d" % i, gpu["spectral"][i], cpu["spectral"][i]) for i in range(s.n_elements)
             if cpu["spectral"][i] != cpu["counter"][i]]
    rows += [(k, gpu[k], cpu[k]) for k in ("n_internal_converted", "n_absorbed_once", "n_absorptions", "n_steps")]
    worst = 0.0
    print("== %s: %.0e float32 photons (GPU) against %.0e float64 photons (checker, reference order)" % (name, n_gpu, n_cpu))
    for label, x, y in rows:
        pg, pc = float(x) / n_gpu, float(y) / n_cpu
        if label in ("n_absorptions", "n_steps"):      # sums of per-photon counts, not binomial: shown only
            print("   %-22s gpu %.6f  cpu %.6f  per photon" % (label, pg, pc))
            continue
        sig = max((pg * (1 - pg) * (1.0 / n_gpu + 1.0 / n_cpu)) ** 0.5, 1.0 / n_cpu)
        worst = max(worst, abs(pg - pc) / sig)
        print("   %-22s gpu %.6f  cpu %.6f  diff %+.2e = %+.2f sigma" % (label, pg, pc, pg - pc, (pg - pc) / sig))
    print("   worst %.2f sigma; one sigma is about %.1e of the photons" % (worst, (0.25 / n_cpu) ** 0.5), flush=True)
    s.close()
