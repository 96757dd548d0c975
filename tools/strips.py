"""Detector-histogram placement on the segmented-edge scene (62 detectors, 64
elements): throughput with the histogram in LDS against the same scene forced
to count in HBM, plus C2 (10 detectors, 12 elements) for scale.
usage (GPU box): python tools/strips.py [photons]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, REPO)
    from raytracer_amd import RayTrace, configs
    n = int(float(sys.argv[1]))
    for strips in (1, 14, 30):
        rt = RayTrace.RayTrace(demo_scene=False)
        rt.LoadJob(configs.strip_job(10, strips=strips))
        s = rt.Scene()
        s.run(1000000, seed=3)
        best = 1e30
        for rep in range(3):
            t = s.run(n, seed=7 + rep)
            best = min(best, s.last_kernel_ms())
        print("  strips %2d: %3d elements, %3d detectors: %.4g photons/s, edge efficiency %.5f"
              % (strips, s.n_elements, 4 * strips + 6, n / (best * 1e-3),
                 float(t["counter"][2:s.n_elements - 6].sum()) / n), flush=True)
        s.close()
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "1e8"
    for label, limit in (("histogram in LDS while it fits", None), ("histogram in HBM", "0")):
        env = dict(os.environ, RTMC_DEBUG="1")
        if limit is not None:
            env["RTMC_LDS_HIST_LIMIT"] = limit
        print(label, flush=True)
        out = subprocess.run([sys.executable, __file__, n, "child"], env=env, stderr=subprocess.PIPE)
        launches = sorted(set(l for l in out.stderr.decode().splitlines() if "trace launch" in l and "1000000 photons" in l))
        for l in launches:
            print("   ", l)
