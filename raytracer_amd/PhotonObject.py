"""Photon source description and single-photon state view (host side).

In the reference one mutable `Photon` object is the photon: it is re-launched
N times and every element mutates it (PhotonObject.py).  On the device a
photon is a row of a structure-of-arrays batch; this class keeps the
constructor signature (PhotonObject.py:9) so that job parsing and user
scripts work unchanged, stores the launch parameters the kernels need, and
exposes the state attributes for single-photon inspection through
`RayTrace.TraceTape`.
"""
from __future__ import annotations

import os

import numpy as np

from . import configs

SUN = "Sun spectrum AG15g"
ISOTROPIC = "Directional Isotropic"
STRICT = "Directional Strict"


class Photon:
    def __init__(self, Location=[0, 0, 0], EmissionMode=ISOTROPIC, EmissionDirection=[0, 0, 0],
                 EmissionWavelengthType=SUN, StartingWavelength=450.0, StartingMaterial=None,
                 RespectPolarization=True, StartingLocation=[0, 0, 0], ScaleFactor=1.0, LaunchArea=None):
        self.Location = Location
        self.EmissionMode = EmissionMode
        self.EmissionDirection = EmissionDirection
        self.EmissionWavelengthType = EmissionWavelengthType
        self.StartingWavelength = StartingWavelength
        self.StartingMaterial = StartingMaterial
        self.RespectPolarization = RespectPolarization
        self.StartingLocation = StartingLocation
        self.ScaleFactor = ScaleFactor
        # (width along x, width along y) of a launch rectangle centred on
        # StartingLocation; None: point source.  The reference has this as
        # commented-out lines with fixed sizes (PhotonObject.py:74-84).
        self.LaunchArea = LaunchArea
        # Two-column sun table to use instead of the file `AM15.txt`; None: read the file
        # (set explicitly, e.g. to configs.synthetic_am15(), never substituted silently).
        self.SunTable = None
        self.SunTableFolder = None      # a job folder to look for AM15.txt in besides the working directory
        # state attributes of the reference object (PhotonObject.py:21-37)
        self.PointCy = [0, 0, 0]
        self.OperationPlane = [[0, 0, 0], [0, 0, 0]]
        self.WasElement = False
        self.BeenAbsorbed = [False, 0]
        self.sgn = 1
        self.Reflected = False
        self.Psi = 0
        self.Direction = np.array([0.0, 0.0, 1.0])
        self.CurrentMaterial = StartingMaterial
        self.Status = "Start"
        self.Wavelength = 450.0
        self.DistanceLeftToAbsorption = 0.0
        self.ElectricFieldVector = [1, 0, 0]
        self.OldPoint = [0, 0, 0]
        self.LineColor = "black"

    def sun_table(self):
        """Two-column sun spectrum used when the wavelength type is the sun
        spectrum: `SunTable` if the caller supplied one, else `AM15.txt` from
        the working directory as the reference reads it (PhotonObject.py:41;
        the job folder is searched too).  The reference does not distribute
        that file and fails with IOError without it; so does this."""
        if self.SunTable is not None:
            return np.asarray(self.SunTable, dtype=np.float64)
        for folder in (".", self.SunTableFolder):
            if folder and os.path.isfile(os.path.join(folder, "AM15.txt")):
                return np.loadtxt(os.path.join(folder, "AM15.txt"))
        raise IOError("AM15.txt not found: the sun-spectrum source (%r) reads this two-column table from the "
                      "working directory; supply the file or set Photon.SunTable" % SUN)
