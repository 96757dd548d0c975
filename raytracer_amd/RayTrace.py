"""Driver with the call surface of the reference's `RayTrace` class.

    LSC = RayTrace.RayTrace()
    LSC.IOOperator.ParsePath(folder)      # or LSC.LoadJob(job_dict)
    LSC.runSimulation()

`runSimulation` does what reference RayTrace.py:157-204 does -- trace
`NumberOfPhotons` photons through `OpticalElements`, leave the tallies on the
element objects, write the log -- but the photon loop runs as HIP kernels
through librtmc.so, optionally sharded over the ranks of a
`torch.distributed` job.
"""
from __future__ import annotations

import os

import numpy as np

from . import Elements, Geometries, Materials, OpticalObject
from . import IOOperation as IOO
from . import PhotonObject as PhObj
from . import RTUtils as RTU
from . import _abi, distributed
from . import scene as _scene

SHAPES = ["None", "Rectangle", "Circle", "Cylinder", "Triangle"]   # RayTrace.py:78
TYPES = ["None", "Mirror", "Interface", "Lens", "Dump"]            # RayTrace.py:79


class RayTrace:
    def __init__(self, demo_scene=True):
        self.MaterialDict = {}
        self.NumberOfPhotons = 1
        self.Seed = None            # None: fresh entropy per run, like np.random.seed()
        self.Device = None          # None: LOCAL_RANK or 0
        self.Devices = None         # a list of GPU indices: this process traces on all of them, one host thread each
        self.Precision = _abi.F32
        self.MaxIterations = _scene.REFERENCE_ITERATION_CAP
        self.LastTallies = None
        self._scene = None
        air = Materials.MaterialObj("air", 1.0, color="red")
        if demo_scene:
            self._load_demo(air)
        else:
            self.Elements = np.zeros((0, 2, 3))
            self.ElementProperties, self.ElementGeometry, self.ElementSize = [], [], []
            self.PhArray = [np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]]), "Start", 450.0, air, 0,
                            [1, 0, 0], 450.0]
        self.Photon = PhObj.Photon(self.PhArray, StartingMaterial=air)
        self.lineclr = air.color
        self.IOOperator = IOO.LoadSaveRT("", self)
        self.CreateOpticalElementList()

    # -- scenes -----------------------------------------------------------
    def _load_demo(self, air):
        """The 17-element test scene the reference builds in its constructor
        (RayTrace.py:21-45): two boxes, loose planes, a cylinder detector, a
        triangle mirror and a circular lens."""
        pva = Materials.MaterialObj("pva", 1.51, IsAbsorber=False, conc_a=1e-3, conc_b=5e-3,
                                    color="green", IsAligned=True, AlignVect=[0, 0, 1.0],
                                    AlignS=0.785)
        r = 1.0 / np.sqrt(2)
        diag, anti = [r, r, 0], [r, -r, 0]
        tri_pts = [[-0.7, -0.1, 0.5], [-0.8, 0.6, -0.4], [-0.4, 0.7, -0.2]]
        extra = np.array([[[-0.1, 0, 0], [1, 0, 0]], [[-0.2, 0, 0], [1, 0, 0]],
                          [[0.3, -0.8, -0.5], [0.8, -0.2, 0.5]],
                          [np.asarray(v) for v in RTU.MakeTriangle(*tri_pts)],
                          [[-0.5, -0.3, 0.0], diag]], dtype=np.float64)
        self.Elements = np.concatenate([RTU.MakeBox([0, 0, 0], [1, 1, 1], [1, 0, 0]),
                                        RTU.MakeBox([0.5, 0.5, 0.0], [0.1, 0.2, 0.4], diag),
                                        extra], axis=0)
        iface = [2, air, pva]
        self.ElementProperties = [[1]] * 6 + [list(iface) for _ in range(8)] \
            + [[4, 0, 0, 0, 1], [1], [3, 0.3]]
        self.ElementGeometry = [1] * 14 + [3, 4, 2]
        self.ElementSize = [[1, 1]] * 6 + [[0.4, 0.2, anti]] * 2 + [[0.1, 0.4, diag]] * 2 \
            + [[0.1, 0.2, diag]] * 2 + [[1, 1]] * 2 + [[0.3], tri_pts, [0.2]]
        self.PhArray = [np.array([[-0.5, -0.5, -0.1], [0.0, 0.0, 1.0]]), "Start", 450.0, air, 0,
                        [1, 0, 0], 450.0]

    def LoadJob(self, job):
        """Load a job given as a dict of the six job arrays (configs.py)."""
        self.IOOperator.ParseJob(job)

    def CreateOpticalElementList(self):
        """Turn the parsed job arrays into `OpticalElements`
        (reference RayTrace.py:70-114; same integer codes and defaults)."""
        self.OpticalElements = []
        self._scene = None
        self.Photon._attach(self)       # its single-photon operations run on this scene
        for i, shape_code in enumerate(self.ElementGeometry):
            support, direction = self.Elements[i][0], self.Elements[i][1]
            size = self.ElementSize[i]
            shape = SHAPES[shape_code]
            if shape == "Rectangle":
                axis = size[2] if len(size) > 2 else [0, 0, 0]
                geo = Geometries.Rectangle(size[0], size[1], support, direction, self.Photon, axis)
            elif shape == "Circle":
                geo = Geometries.Circle(size[0], support, direction, self.Photon)
            elif shape == "Cylinder":
                geo = Geometries.Cylinder(size[0], support, direction, self.Photon)
            elif shape == "Triangle":
                geo = Geometries.Triangle(size[0], size[1], size[2], self.Photon)
            else:
                raise ValueError("element %d has no geometry" % i)
            props = self.ElementProperties[i]
            kind = TYPES[props[0]]
            if kind == "Interface":
                typ = Elements.Interface(props[1], props[2], self.Photon)
            elif kind == "Lens":
                typ = Elements.Lens(props[1], self.Photon)
            elif kind == "Dump":
                typ = Elements.Dump(props[4], self.Photon)
            else:                       # "Mirror", and "None" as in the reference
                typ = Elements.Mirror(self.Photon)
            self.OpticalElements.append(OpticalObject.OpticalObject(typ, geo))

    # -- device -----------------------------------------------------------
    def Scene(self):
        """The device-side scene for the current element list (built once)."""
        if self._scene is None:
            self._scene = _scene.Scene(self.OpticalElements, self.Photon, self.MaterialDict,
                                       max_iterations=self.MaxIterations)
        return self._scene

    def runSimulation(self):
        rank, world = distributed.rank_world()
        device = self.Device if self.Device is not None else distributed.local_device()
        seed = self.Seed
        if seed is None:
            seed = distributed.shared_entropy()
        first, count = distributed.shard_range(int(self.NumberOfPhotons), rank, world)
        sc = self.Scene()
        block = _scene.new_tallies(sc.n_elements)
        if self.Devices:
            self._run_on_devices(sc, list(self.Devices), seed, first, count, block)
        elif count > 0:
            sc.run(count, seed=seed, first_photon=first, device=device,
                   precision=self.Precision, tallies=block)
        distributed.allreduce_tallies(block, device)
        t = _scene.tallies_to_dict(*block)
        self.LastTallies = t
        self.LastSeed = seed
        self._publish(t)
        if self.IOOperator.ActivePath and rank == 0:
            self.IOOperator.SaveLog()
        return t

    def _run_on_devices(self, sc, devices, seed, first, count, block):
        """One process, several GPUs: the photon ids [first, first + count) are cut into one contiguous
        block per entry of `devices` (the sequential photon loop of reference RayTrace.py:166, sharded),
        every block is traced by a host thread of its own through rtmc_run on its GPU -- calls on
        different GPUs run side by side, calls on the same GPU one after the other -- and the integer
        tallies are added up on the host."""
        import threading
        parts = [_scene.new_tallies(sc.n_elements) for _ in devices]
        errors = []

        def work(k):
            lo, n = distributed.shard_range(count, k, len(devices))
            try:
                if n > 0:
                    sc.run(n, seed=seed, first_photon=first + lo, device=devices[k],
                           precision=self.Precision, tallies=parts[k])
            except Exception as err:        # reported by the calling thread
                errors.append(err)
        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(devices))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        total, arrays = block
        for part, part_arrays in parts:
            for key in _abi.TALLY_SCALARS:
                setattr(total, key, getattr(total, key) + getattr(part, key))
            for key in arrays:
                arrays[key] += part_arrays[key]

    def CheckElementsForCrossingPoint(self):
        """Nearest valid element on the current trajectory of `self.Photon`
        (reference RayTrace.py:120-154): sets `SmallestDistance`, `FirstEl`,
        `el_n`, `Photon.PointCy` and `OldElement` like the reference does.
        The search itself runs on the device (`rtmc_op_crossing`)."""
        ph = self.Photon
        batch = _scene.new_photon_batch(1)
        old_point = getattr(self, "OldPoint", ph.Location)
        for j in range(3):
            batch["loc"][j, 0] = ph.Location[j]
            batch["dir"][j, 0] = ph.Direction[j]
            batch["old_point"][j, 0] = old_point[j]
        last = getattr(self, "OldElement", None)
        batch["last_element"][0] = self.OpticalElements.index(last) if last is not None else -1
        batch["was_element"][0] = int(bool(ph.WasElement)) if last is not None else 0
        batch["reflected"][0] = int(bool(ph.Reflected))
        device = self.Device if self.Device is not None else distributed.local_device()
        el, dist, pts = self.Scene().crossing(batch, device=device, precision=_abi.F64)
        self.SmallestDistance = float(dist[0])
        ph.PointCy = [0, 0, 0] if el[0] < 0 else pts[0].copy()
        if el[0] >= 0:
            self.FirstEl = self.OpticalElements[int(el[0])]
            self.el_n = int(el[0])
        self.OldElement = getattr(self, "FirstEl", None)
        return self.OldElement

    def _publish(self, t):
        """Copy the tallies onto the element objects the way the reference
        leaves them after a run (Elements.py:15,39,139-144)."""
        for i, oe in enumerate(self.OpticalElements):
            typ = oe.Type
            if typ.kind == "Dump":
                typ.Counter = int(t["counter"][i])
                typ.EnergyCounter = float(t["energy"][i])
                typ.SpectralCounter = int(t["spectral"][i])
                typ.IncidentSpectrum[:, 1] = t["spectrum"][i]
            else:
                typ.BounceCounter = int(t["bounce"][i])
        self.Photon.BeenAbsorbed = [False, int(t["n_absorbed_once"])]

    def OpticalEfficiency(self, elements=None):
        """Sum of detector counts over `elements` (default: all) / photons."""
        t = self.LastTallies
        idx = range(len(self.OpticalElements)) if elements is None else elements
        return sum(int(t["counter"][i]) for i in idx) / float(t["n_photons"])

    def TraceTape(self, tape, tape_off, n_steps, precision=_abi.F64, device=0):
        """Advance freshly launched photons through `n_steps` loop iterations
        with the variates taken from `tape` (one photon per tape segment) and
        return the per-iteration trace, shape (n, n_steps, 16)."""
        sc = self.Scene()
        n = len(tape_off) - 1
        batch = _scene.new_photon_batch(n)
        tr = sc.step_tape(batch, tape, tape_off, n_steps, launch=True, device=device,
                          precision=precision, trace=True)
        return tr, batch
