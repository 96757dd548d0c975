"""Optical element behaviours (host side): result holders for the tallies.

The reference's `Mirror`, `Interface`, `Lens` and `Dump` objects both carry the
per-element counters and execute the optical operation on the one shared
photon (Elements.py).  Here the operation is device code selected by the
element's type code; these classes keep the constructor signatures and the
counter attributes that user scripts and `SaveLog` read after a run.
"""
from __future__ import annotations

import numpy as np

SPECTRUM_FIRST_NM = 200
SPECTRUM_BINS = 3800


class _Behaviour:
    kind = "None"

    def __init__(self, Photon=None):
        self.Photon = Photon
        self.BounceCounter = 0


class Mirror(_Behaviour):
    """Specular reflector; counts bounces (Elements.py:9-28)."""
    kind = "Mirror"
    color, alpha = "blue", 0.05


class Interface(_Behaviour):
    """Boundary between two media: total internal reflection, polarised
    Fresnel reflection or Snell refraction (Elements.py:31-114)."""
    kind = "Interface"
    color, alpha = "yellow", 0.1

    def __init__(self, Material1, Material2, Photon=None):
        super().__init__(Photon)
        self.MaterialSide1, self.MaterialSide2 = Material1, Material2


class Lens(_Behaviour):
    """Ideal thin lens (Elements.py:117-132)."""
    kind = "Lens"
    color, alpha = "red", 0.1

    def __init__(self, FocalLength, Photon=None):
        super().__init__(Photon)
        self.FocalLength = FocalLength


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
