"""Photon source description and single-photon state view (host side).

In the reference one mutable `Photon` object is the photon: it is re-launched
N times and every element mutates it (PhotonObject.py).  On the device a
photon is a row of a structure-of-arrays batch; this class keeps the
constructor signature (PhotonObject.py:9) so that job parsing and user
scripts work unchanged, stores the launch parameters the kernels need, and
exposes the state attributes for single-photon inspection through
`RayTrace.TraceTape`.

The reference's methods that move the one photon by hand (its TestPhoton.py
drives them: GeneratePhoton, ResetPhoton, CheckForAbsorption, ReflectOnPlane
...) are kept too.  Each of them packs this object's state into a one-photon
batch, runs the matching device routine through `rtmc_op_photon` and unpacks
the result; the random numbers come from `numpy.random` as in the reference,
so `numpy.random.seed` makes a hand-stepped photon repeatable.
"""
from __future__ import annotations

import os
import weakref

import numpy as np

from . import RTUtils as RTU
from . import _abi, configs

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

    # ------------------------------------------------------------------
    # single-photon operations (device routines behind the reference's names)
    # ------------------------------------------------------------------
    STATUS = ["Start", "Absorbed", "ReemittedDonor", "ReemittedAcceptor", "InternalConverted", "Dumped",
              "Lost"]                                     # RTMC_ST_* in order
    Precision = _abi.F64
    Device = 0

    def _attach(self, owner):
        """Called by RayTrace: operations then run on the owner's scene."""
        self._owner = weakref.ref(owner)

    def _scene(self):
        owner = getattr(self, "_owner", None)
        owner = owner() if owner is not None else None
        if owner is not None:
            return owner.Scene()
        # a photon on its own (as in the reference's known-answer scripts): a private scene
        # that holds its source, its media and one mirror
        from . import Elements, Geometries, Materials, OpticalObject, scene as _scene
        if self.StartingMaterial is None:
            self.StartingMaterial = Materials.MaterialObj("air", 1.0)
        if self.CurrentMaterial is None:
            self.CurrentMaterial = self.StartingMaterial
        sun = self.EmissionWavelengthType == SUN
        mats = getattr(self, "_private_mats", [])
        for m in (self.StartingMaterial, self.CurrentMaterial):
            if not any(m is x for x in mats):
                mats = mats + [m]
        key = (tuple(id(m) for m in mats), self.EmissionMode,
               tuple(np.asarray(self.EmissionDirection, dtype=float)), self.EmissionWavelengthType,
               None if sun else float(self.StartingWavelength),
               tuple(np.asarray(self.StartingLocation, dtype=float)), bool(self.RespectPolarization),
               float(self.ScaleFactor), tuple(self.LaunchArea) if self.LaunchArea else None)
        if getattr(self, "_private_key", None) != key:
            old = getattr(self, "_private_scene", None)
            if old is not None:
                old.close()
            mirror = OpticalObject.OpticalObject(
                Elements.Mirror(self), Geometries.Rectangle(1.0, 1.0, [0.0, 0.0, -1000.0], [0.0, 0.0, 1.0]))
            self._private_scene = _scene.Scene([mirror], self, {"%d:%s" % (i, m.name): m for i, m in enumerate(mats)})
            self._private_key, self._private_mats = key, mats
        return self._private_scene

    def _pack(self, sc, geometric=False):
        from . import scene as _scene
        b = _scene.new_photon_batch(1)
        for name, value in (("loc", self.Location), ("dir", self.Direction), ("efield", self.ElectricFieldVector),
                            ("old_point", self.OldPoint)):
            try:
                b[name][:, 0] = np.asarray(value, dtype=np.float64).reshape(-1)[:3]
            except (TypeError, ValueError):     # not a point yet (a photon that was never generated)
                b[name][:, 0] = np.asarray(self.StartingLocation, dtype=np.float64)[:3]
        if geometric:            # the operation does not look at the wavelength
            b["wavelength"][0] = b["start_wavelength"][0] = sc.wavelengths()[0]
        else:
            # the device knows the wavelengths a photon of this scene can carry (source lines and table
            # rows); any other value -- the constructor's default on a photon that was never generated --
            # is resolved to the nearest of them, as the reference's lookups would (PhotonObject.py:235,288)
            known = np.sort(sc.wavelengths())
            b["wavelength"][0] = known[RTU.FindClosestListElement(known, float(self.Wavelength))]
            b["start_wavelength"][0] = known[RTU.FindClosestListElement(known, float(self.StartingWavelength))]
        b["dist_left"][0] = self.DistanceLeftToAbsorption
        b["psi"][0] = self.Psi
        mats = sc.packed.materials
        current = self.CurrentMaterial if self.CurrentMaterial is not None else self.StartingMaterial
        idx = [i for i, m in enumerate(mats) if m is current]
        if not idx:
            raise ValueError("the photon's CurrentMaterial %r is not one of the scene's materials"
                             % getattr(current, "name", current))
        b["material"][0] = idx[0]
        b["status"][0] = self.STATUS.index(self.Status) if self.Status in self.STATUS else 0
        b["was_element"][0] = int(bool(self.WasElement))
        b["reflected"][0] = int(bool(self.Reflected))
        return b

    def _unpack(self, sc, b, geometric=False):
        self.Location = b["loc"][:, 0].copy()
        self.Direction = b["dir"][:, 0].copy()
        self.ElectricFieldVector = b["efield"][:, 0].copy()
        if not geometric:
            self.OldPoint = b["old_point"][:, 0].copy()
            self.Wavelength = float(b["wavelength"][0])
            self.DistanceLeftToAbsorption = float(b["dist_left"][0])
            self.Psi = float(b["psi"][0])
            self.CurrentMaterial = sc.packed.materials[int(b["material"][0])]
            self.Status = self.STATUS[int(b["status"][0])]
            self.WasElement = bool(b["was_element"][0])
            self.Reflected = bool(b["reflected"][0])

    def _plane_args(self, a=0.0, b=0.0, c=0.0, old_direction=(0.0, 0.0, 0.0), point=None):
        args = np.zeros((1, _abi.PHOP_ARGS))
        args[0, 0:3] = np.asarray(self.OperationPlane[0] if point is None else point, dtype=np.float64)
        args[0, 3:6] = np.asarray(self.OperationPlane[1], dtype=np.float64)
        args[0, 6:9] = (a, b, c)
        args[0, 9:12] = np.asarray(old_direction, dtype=np.float64)
        return args

    def _operate(self, op, args=None, tape=None, uniform_tape=False, geometric=False):
        sc = self._scene()
        b = self._pack(sc, geometric)
        off = None if tape is None else np.array([0, len(tape)], dtype=np.uint64)
        out = sc.photon_op(op, b, args, tape, off, uniform_tape=uniform_tape, device=self.Device,
                           precision=self.Precision)
        self._unpack(sc, b, geometric)
        return out[0]

    def GeneratePhoton(self):
        """PhotonObject.py:61-97.  The draws are made here, in the reference's order and
        kinds (randn(3) for an isotropic direction, one uniform for the sun-spectrum
        choice, randn(3) for the polarisation), and consumed by the device routine."""
        tape = []
        if self.EmissionMode == ISOTROPIC:
            tape += list(np.random.randn(3))
        if self.LaunchArea:
            tape += [np.random.rand(), np.random.rand()]
        if self.EmissionWavelengthType == SUN:
            tape.append(np.random.random_sample())
        tape += list(np.random.randn(3))
        sc = self._scene()
        b = self._pack(sc, geometric=True)
        sc.photon_op("generate", b, None, np.array(tape), np.array([0, len(tape)], dtype=np.uint64),
                     uniform_tape=False, device=self.Device, precision=self.Precision)
        self._unpack(sc, b)
        self.OldPoint = self.Location.copy()

    def ResetPhoton(self, OperationPlane):
        """PhotonObject.py:49-53."""
        self.GeneratePhoton()
        self.OperationPlane = [np.array(OperationPlane[0], dtype=np.float64),
                               np.array(OperationPlane[1], dtype=np.float64)]
        self.StartingWavelength = self.Wavelength
        self.Reflected = False

    def FindSign(self):
        """PhotonObject.py:56-58."""
        self.sgn = np.sign(-np.dot(np.asarray(self.Direction, dtype=np.float64),
                                   np.asarray(self.OperationPlane[1], dtype=np.float64)))

    def PrepareExecution(self):
        """PhotonObject.py:135-139."""
        self.FindSign()
        self.OldPoint = np.array(self.Location, dtype=np.float64)
        self.OldDirection = np.array(self.Direction, dtype=np.float64)
        self.LineColor = getattr(self.CurrentMaterial, "color", "black")

    def PostProcessExecution(self):
        """PhotonObject.py:141-143."""
        self.UpdatePolarization(self.OldDirection)
        self.WasElement = True

    def UpdatePolarization(self, OldDirection):
        """PhotonObject.py:117-129."""
        self._operate("polarization", self._plane_args(old_direction=OldDirection), geometric=True)

    def ReflectOnPlane(self):
        """PhotonObject.py:145-150."""
        self._operate("reflect", self._plane_args(), geometric=True)

    def RefractOnPlane(self, n1, n2):
        """PhotonObject.py:195-199 (uses Photon.sgn as FindSign left it)."""
        self._operate("refract", self._plane_args(n1, n2, self.sgn), geometric=True)

    def FocusThinLens(self, f):
        """PhotonObject.py:158-186."""
        self._operate("lens", self._plane_args(f), geometric=True)

    def LamBeerDistance(self):
        """PhotonObject.py:283-305: a new free path; the unit exponential is drawn here
        (none if the medium does not absorb, as in the reference)."""
        current = self.CurrentMaterial if self.CurrentMaterial is not None else self.StartingMaterial
        tape = [np.random.exponential(1.0)] if getattr(current, "IsAbsorber", False) else [0.0]
        self._operate("freepath", self._plane_args(point=(0, 0, 0)), tape=np.array(tape))

    def CheckForAbsorption(self, OldPoint, SmallestDistance):
        """PhotonObject.py:206-264, pathway choice and re-emission included.  Which draws
        an absorption needs depends on what happens in it, so the device routine gets a
        run of uniforms and turns them into the variates it needs.  The run is drawn from
        numpy's global generator before the outcome is known: 256 values per call (a few are
        used), so the generator moves on further than the reference's does; a chain of back
        transfers or rejections that needs more is given a longer run of the same stream."""
        self.OldPoint = np.array(OldPoint, dtype=np.float64)
        sc = self._scene()
        start = self._pack(sc)
        start["hit_distance"][0] = SmallestDistance
        args = self._plane_args(point=self.PointCy)
        tape = np.random.random_sample(256)
        while True:
            b = {k: v.copy() for k, v in start.items()}
            try:
                sc.photon_op("absorption", b, args, tape, np.array([0, len(tape)], dtype=np.uint64),
                             uniform_tape=True, device=self.Device, precision=self.Precision)
                break
            except _abi.RtmcError as err:
                if "tape exhausted" not in str(err) or len(tape) >= 1 << 20:
                    raise
                tape = np.concatenate([tape, np.random.random_sample(3 * len(tape))])
        was_absorbed = int(b["status"][0]) != 0
        self._unpack(sc, b)
        if was_absorbed:
            self.BeenAbsorbed[0] = True
