"""Where a wave of the trace kernel spends its time: runs the named
configurations on the build with cycle stamps (csrc/librtmc_prof.so,
`make -C raytracer_amd/csrc librtmc_prof.so`) and prints, per configuration,
the library's section report: passes / cycles / lanes of the renew block and of
the step, and inside the step the cycles of Philox, search, absorption check
and element operation.

usage: python tools/sections.py [photons] [config ...]   (on the GPU box)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTMC_LIBRARY"] = os.path.join(REPO, "raytracer_amd", "csrc", "librtmc_prof.so")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
for name in sys.argv[2:] or ["C2", "C3", "C4", "C2r"]:
    s = H.raytrace_for(name).Scene()
    print("-- %s: warm-up launch of 1e6 photons (ignore)" % name, file=sys.stderr, flush=True)
    s.run(1000000, seed=3)
    sys.stderr.flush()
    print("== %s, %d photons" % (name, n), file=sys.stderr, flush=True)
    t = s.run(n, seed=7)
    print("[%s] %.3e photons/s (stamped build), %.3f iterations/photon, kernel %.2f ms"
          % (name, n / (s.last_kernel_ms() * 1e-3), t["n_steps"] / t["n_photons"], s.last_kernel_ms()),
          file=sys.stderr, flush=True)
    s.close()
