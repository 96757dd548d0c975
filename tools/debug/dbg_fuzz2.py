"""Normal incidence on a tilted plane in one of the random scenes: what numpy, the CPU checker and
the float64 kernels make of the cosine (debug aid).  usage: python tools/dbg_fuzz2.py SEED"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_fuzz import random_job  # noqa: E402
from raytracer_amd import RayTrace, _abi  # noqa: E402
from raytracer_amd import scene as sc  # noqa: E402
from oracle import oracle  # noqa: E402

seed = int(sys.argv[1])
rt = RayTrace.RayTrace(demo_scene=False)
rt.LoadJob(random_job(seed))
s = rt.Scene()
o = oracle.OracleScene.from_raytrace(rt)
raw = np.asarray(rt.Photon.EmissionDirection, dtype=np.float64)
D = raw / np.linalg.norm(raw)
print("raw direction", raw.tolist(), "norm-1", np.linalg.norm(raw) - 1)
for el in (0, 1):
    n = rt.OpticalElements[el].Geometry.DirectionVector
    print("element", el, "numpy dot(D, n) + 1 =", np.dot(D, n) + 1, " dot(n, n) - 1 =", np.dot(n, n) - 1)
# launch on both and compare the direction bits
b = sc.new_photon_batch(1)
tape = np.array([0.3, -0.2, 0.9, 0.5, 0.5, 0.5, 0.5])
off = np.array([0, len(tape)], dtype=np.uint64)
s.step_tape(b, tape, off, 0, launch=True, precision=_abi.F64)
c = sc.new_photon_batch(1)
o.step_tape(c, tape, off, 0, launch=True)
print("gpu D", b["dir"][:, 0].tolist())
print("cpu D", c["dir"][:, 0].tolist())
print("numpy D", D.tolist())
for name, batch in (("gpu", b),):
    for _ in range(2):
        if name == "gpu":
            s.step_tape(batch, tape, off, 1, launch=False, precision=_abi.F64)
        else:
            o.step_tape(batch, tape, off, 1, launch=False)
        print(name, "after a step: el", batch["last_element"][0], "reflected", batch["reflected"][0], "mat", batch["material"][0],
              "loc", batch["loc"][:, 0].tolist(), "dir", batch["dir"][:, 0].tolist())
# the interface's reflectance for the launched photon, both ways
el = rt.OpticalElements[1]
b = sc.new_photon_batch(1)
s.step_tape(b, tape, off, 0, launch=True, precision=_abi.F64)
args = np.zeros((1, _abi.PHOP_ARGS))
args[0, 0:3], args[0, 3:6] = el.Geometry.SupportVector, el.Geometry.DirectionVector
args[0, 6:9] = (1.0, 1.5, 1.0)
b["tape_pos"][:] = 0
out = s.photon_op("reflectance", b, args, precision=_abi.F64)
print("gpu tir, R", out[0].tolist(), "E", b["efield"][:, 0].tolist())
r, refr, refl = oracle.op_interface(el.Geometry.SupportVector, el.Geometry.DirectionVector, 1.0, 1.5,
                                    b["loc"][:, 0].copy(), b["dir"][:, 0].copy(), b["efield"][:, 0].copy())
print("cpu tir, R, sgn", r)
print("materials", [(m.name, m.n) for m in s.packed.materials], "el1 sides", el.Type.MaterialSide1.name, el.Type.MaterialSide2.name)
