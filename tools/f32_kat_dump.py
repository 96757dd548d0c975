"""float32 against float64 on the reference's per-operation vectors (tests/golden/kat.npz,
kat_shapes.npz): runs the interface and lens operations as batches in both precisions and
stores inputs, outputs and the reference's answers for analysis off the GPU box.
usage (GPU box): python tools/f32_kat_dump.py gpurun_out/r3/kat_f32.npz"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from raytracer_amd import Materials, PhotonObject, _abi  # noqa: E402
from raytracer_amd import scene as sc  # noqa: E402

k = np.load(os.path.join(REPO, "tests", "golden", "kat.npz"))
ks = np.load(os.path.join(REPO, "tests", "golden", "kat_shapes.npz"))
photon = PhotonObject.Photon(EmissionMode="Directional Strict", EmissionDirection=[0, 0, 1.0],
                             EmissionWavelengthType="fixed", StartingMaterial=Materials.MaterialObj("kat_air", 1.0))
s = photon._scene()


def batch(L, D, E):
    b = sc.new_photon_batch(len(L))
    b["loc"][:], b["dir"][:], b["efield"][:] = L.T, D.T, E.T
    b["wavelength"][:] = b["start_wavelength"][:] = s.wavelengths()[0]
    return b


def state(b):
    return np.concatenate([b["loc"].T, b["dir"].T, b["efield"].T], axis=1)


n = len(k["if_res"])
args = np.zeros((n, _abi.PHOP_ARGS))
args[:, 0:3], args[:, 3:6] = k["if_S"], k["if_n"]
args[:, 6] = np.where(k["if_inside"], 1.5, 1.0)
args[:, 7] = np.where(k["if_inside"], 1.0, 1.5)
args[:, 8] = k["if_res"][:, 2]
args[:, 9:12] = k["if_D"]
m = len(ks["lens_f"])
largs = np.zeros((m, _abi.PHOP_ARGS))
largs[:, 0:3], largs[:, 3:6], largs[:, 6], largs[:, 9:12] = ks["lens_S"], ks["lens_n"], ks["lens_f"], ks["lens_D"]
out = {}
for tag, prec in (("f32", _abi.F32), ("f64", _abi.F64)):
    b = batch(k["if_L"], k["if_D"], k["if_E"])
    out["refl_R_" + tag] = s.photon_op("reflectance", b, args, precision=prec)
    for op in ("refract", "reflect"):
        b = batch(k["if_L"], k["if_D"], k["if_E"])
        s.photon_op(op, b, args, precision=prec)
        s.photon_op("polarization", b, args, precision=prec)
        out[op + "_" + tag] = state(b)
    b = batch(ks["lens_L"], ks["lens_D"], ks["lens_E"])
    s.photon_op("lens", b, largs, precision=prec)
    s.photon_op("polarization", b, largs, precision=prec)
    out["lens_" + tag] = state(b)
np.savez(sys.argv[1], **out)
print("stored", sys.argv[1])
