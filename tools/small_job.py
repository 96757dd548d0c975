"""Wall time of the reference's own kind of job, end to end: a job folder with 1e3 ... 1e6 photons of
the shipped geometry (C1_open: slab 0.012 cm without a box; C1: with the detector box), parsed,
traced and logged by RayTrace.runSimulation.  usage: python tools/small_job.py  (on the GPU box)"""
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raytracer_amd import RayTrace, configs, prt  # noqa: E402

root = tempfile.mkdtemp()
first = True
for name in ("C1_open", "C1", "C2"):
    for n in (1000, 10000, 100000, 1000000):
        folder = os.path.join(root, "%s_%d" % (name, n))
        os.makedirs(folder)
        prt.dump_job(folder, configs.make_job(name, n))
        for fn in ("LumogenRedAbs.txt", "LumogenRedEm.txt", "OxacinAbs.txt", "OxacinEm.txt"):
            src = os.path.join(REPO, "raytracer_amd", "data", fn)
            if os.path.exists(src):
                shutil.copy(src, folder)
        t0 = time.perf_counter()
        rt = RayTrace.RayTrace(demo_scene=False)
        rt.IOOperator.ParsePath(folder)
        t1 = time.perf_counter()
        rt.Seed = 1
        t = rt.runSimulation()
        t2 = time.perf_counter()
        if first:       # the first call of a process also creates the device context and loads the code object
            first = False
            rt2 = RayTrace.RayTrace(demo_scene=False)
            rt2.IOOperator.ParsePath(folder)
            rt2.Seed = 1
            t1 = time.perf_counter()
            t = rt2.runSimulation()
            t2 = time.perf_counter()
        print("%-8s %8d photons: parse %.1f ms, runSimulation (scene build, upload, trace, log files) %.1f ms, "
              "kernel %.3f ms, capped %d, iterations/photon %.1f"
              % (name, n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), rt.Scene().last_kernel_ms(0), t["n_capped"],
                 t["n_steps"] / t["n_photons"]), flush=True)
shutil.rmtree(root)
