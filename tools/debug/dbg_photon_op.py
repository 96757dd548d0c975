"""Runs the single-photon operations named on the command line, one after the other, with
progress lines and a watchdog (debugging aid): python tools/dbg_photon_op.py reflect refract ..."""
import faulthandler
import sys
import time

import numpy as np

sys.path.insert(0, ".")
faulthandler.dump_traceback_later(30, exit=True)
from raytracer_amd import Elements, Materials, PhotonObject, _abi  # noqa: E402


def say(*a):
    print("[%.2f]" % time.time(), *a, flush=True)


air, glass = Materials.MaterialObj("kat_air", 1.0), Materials.MaterialObj("kat_glass", 1.5)
ph = PhotonObject.Photon(EmissionMode="Directional Strict", EmissionDirection=[0, 0, 1.0],
                         EmissionWavelengthType="fixed", StartingMaterial=air)
if "f32" in sys.argv:
    ph.Precision = _abi.F32
iface = Elements.Interface(air, glass, ph)
for op in sys.argv[1:]:
    if op == "f32":
        continue
    ph.Location, ph.Direction = np.array([0.1, 0.2, 0.5]), np.array([0.0, 0.6, -0.8])
    ph.ElectricFieldVector = np.array([1.0, 0.0, 0.0])
    ph.OperationPlane = [np.zeros(3), np.array([0.0, 0.0, 1.0])]
    ph.PrepareExecution()
    say(op, "...")
    if op == "reflectance":
        say(iface.Reflectance(1.0, 1.5))
    elif op == "reflect":
        ph.ReflectOnPlane()
    elif op == "refract":
        ph.RefractOnPlane(1.0, 1.5)
    elif op == "lens":
        ph.FocusThinLens(0.7)
    elif op == "polarization":
        ph.Direction = np.array([0.0, 0.6, 0.8])
        ph.PostProcessExecution()
    elif op == "generate":
        ph.GeneratePhoton()
    elif op == "freepath":
        ph.LamBeerDistance()
    say(op, ph.Location, ph.Direction, ph.ElectricFieldVector)
say("done")
