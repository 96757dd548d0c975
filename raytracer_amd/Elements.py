"""Optical element behaviours (host side): result holders for the tallies.

The reference's `Mirror`, `Interface`, `Lens` and `Dump` objects both carry the
per-element counters and execute the optical operation on the one shared
photon (Elements.py).  Here the operation is device code selected by the
element's type code; these classes keep the constructor signatures and the
counter attributes that user scripts and `SaveLog` read after a run.
"""
from __future__ import annotations

import numpy as np

from . import RTUtils as RTU

SPECTRUM_FIRST_NM = 200
SPECTRUM_BINS = 3800


class _Behaviour:
    kind = "None"

    def __init__(self, Photon=None):
        self.Photon = Photon
        self.BounceCounter = 0

    # The reference's elements operate on the one shared photon, with the plane the
    # caller put into Photon.OperationPlane (RayTrace.py:187).  The methods below keep
    # that protocol for scripts that move a photon by hand; every geometric or optical
    # step inside them is a device routine (see PhotonObject.Photon).
    def _side_of_old_point(self, OldPoint, Point):
        return RTU.CheckSides(OldPoint, Point, self.Photon.OperationPlane)


class Mirror(_Behaviour):
    """Specular reflector; counts bounces (Elements.py:9-28)."""
    kind = "Mirror"
    color, alpha = "blue", 0.05

    def ExecuteOperation(self):
        """Elements.py:18-23."""
        ph = self.Photon
        ph.PrepareExecution()
        ph.ReflectOnPlane()
        self.BounceCounter += 1
        ph.PostProcessExecution()

    def IsValidNextPoint(self, OldPoint, Point):
        """Elements.py:26-28: the photon stays on its side of a mirror."""
        return self._side_of_old_point(OldPoint, Point) == True  # noqa: E712


class Interface(_Behaviour):
    """Boundary between two media: total internal reflection, polarised
    Fresnel reflection or Snell refraction (Elements.py:31-114)."""
    kind = "Interface"
    color, alpha = "yellow", 0.1

    def __init__(self, Material1, Material2, Photon=None):
        super().__init__(Photon)
        self.MaterialSide1, self.MaterialSide2 = Material1, Material2

    def _plane_scalars(self, n1, n2):
        ph = self.Photon
        return ph._operate("reflectance", ph._plane_args(n1, n2, ph.sgn), geometric=True)

    def IsTotalReflect(self, n1, n2):
        """Elements.py:75-81."""
        return bool(self._plane_scalars(n1, n2)[0])

    def Reflectance(self, n1, n2):
        """Elements.py:84-109 (the photon's polarisation comes back normalised, as there)."""
        return float(self._plane_scalars(n1, n2)[1])

    def ExecuteOperation(self):
        """Elements.py:41-73."""
        ph = self.Photon
        ph.PrepareExecution()
        here = ph.CurrentMaterial
        other = self.MaterialSide2 if here.name == self.MaterialSide1.name else self.MaterialSide1
        tir, R = self._plane_scalars(here.n, other.n)
        if tir:
            ph.ReflectOnPlane()
            ph.Reflected = True
            self.BounceCounter += 1
        elif np.random.rand() > R:          # a NaN reflectance compares false: reflected, as in the reference
            ph.RefractOnPlane(here.n, other.n)
            ph.CurrentMaterial = other
            ph.Reflected = False
            ph.LamBeerDistance()
        else:
            ph.ReflectOnPlane()
            ph.Reflected = True
            self.BounceCounter += 1
        ph.PostProcessExecution()

    def IsValidNextPoint(self, OldPoint, Point):
        """Elements.py:113-114: same side iff the photon was reflected."""
        return self._side_of_old_point(OldPoint, Point) == self.Photon.Reflected


class Lens(_Behaviour):
    """Ideal thin lens (Elements.py:117-132)."""
    kind = "Lens"
    color, alpha = "red", 0.1

    def __init__(self, FocalLength, Photon=None):
        super().__init__(Photon)
        self.FocalLength = FocalLength

    def ExecuteOperation(self):
        """Elements.py:125-129."""
        ph = self.Photon
        ph.PrepareExecution()
        ph.FocusThinLens(self.FocalLength)
        ph.PostProcessExecution()

    def IsValidNextPoint(self, OldPoint, Point):
        """Elements.py:131-132: a lens is always crossed."""
        return self._side_of_old_point(OldPoint, Point) == False  # noqa: E712


class Dump(_Behaviour):
    """Detector: stops the photon and logs it (Elements.py:135-169).

    `DetectionSpectrum` is an int (count everything) or a two-column quantum
    efficiency table.  `IncidentSpectrum` is the 3800 x 2 array
    [wavelength 200..3999 nm, count]."""
    kind = "Dump"
    color, alpha = "black", 0.2

    def __init__(self, DetectionSpectrum=1, Photon=None):
        super().__init__(Photon)
        self.DetectionSpectrum = DetectionSpectrum
        self.reset()

    def reset(self):
        self.Counter = 0
        self.EnergyCounter = 0.0
        self.SpectralCounter = 0
        self.IncidentSpectrum = np.zeros((SPECTRUM_BINS, 2))
        self.IncidentSpectrum[:, 0] = np.arange(SPECTRUM_FIRST_NM, SPECTRUM_FIRST_NM + SPECTRUM_BINS)

    def ExecuteOperation(self):
        """Elements.py:152-169: stop the photon and log it (book-keeping only; the move
        to Photon.PointCy and the polarisation update are the photon's)."""
        ph = self.Photon
        ph.PrepareExecution()
        ph.Status = "Dumped"
        self.Counter += 1
        self.EnergyCounter += 1 * ph.StartingWavelength / ph.Wavelength
        self.IncidentSpectrum[RTU.FindClosestListElement(self.IncidentSpectrum[:, 0], ph.Wavelength), 1] += 1
        if isinstance(self.DetectionSpectrum, (int, np.integer)):
            self.SpectralCounter += 1
        else:
            eqe = np.asarray(self.DetectionSpectrum)
            if eqe[RTU.FindClosestListElement(eqe[:, 0], ph.Wavelength), 1] > np.random.rand():
                self.SpectralCounter += 1
        ph.Location = np.array(ph.PointCy, dtype=np.float64)
        ph.PostProcessExecution()
