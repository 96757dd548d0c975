"""Job folders: reading the six `.prt` files and writing the result log.

File names and layouts are those of the reference (IOOperation.py): a job is
`Elements.prt`, `ElementProperties.prt`, `ElementGeometry.prt`,
`ElementSize.prt`, `Materials.prt`, `Photon.prt`; `SaveLog` writes a
time-stamped sub-folder with a copy of the job, `Logfile.txt`,
`DetectorSpectrum<i>.txt` and `AbsorbedPhotons.txt`.  The files are decoded
by `prt.py`, which never unpickles.
"""
from __future__ import annotations

import copy
import datetime
import os

import numpy as np

from . import Materials, prt
from . import PhotonObject as PhObj

# position of each MaterialObj argument in a `Materials.prt` row
# (reference IOOperation.py:81 / RayTracerGui.py:815)
MATERIAL_ROW = ["name", "n", "IsAbsorber", "conc_b", "conc_a", "color", "IsAligned", "AlignVect",
                "AlignS", "R_0", "Q_D", "Q_A", "tau_D", "tau_A", "emax_D", "emax_A", "sp_abs_D",
                "sp_em_D", "sp_abs_A", "sp_em_A", "AbsEmAngle", "FixDist", "FixDistR",
                "FixDistNeighbors"]


_COLUMN_TEXT = {}      # bytes of a column -> its formatted entries (the 3800 bin wavelengths of every detector file)


def _formatted_column(col):
    """'%.18e' of every entry of a float64 column, formatting each distinct bit pattern once."""
    col = np.ascontiguousarray(col, dtype=np.float64)
    key = col.tobytes()
    hit = _COLUMN_TEXT.get(key)
    if hit is not None:
        return hit
    bits, inverse = np.unique(col.view(np.int64), return_inverse=True)
    texts = np.array(["%.18e" % v for v in bits.view(np.float64).tolist()])[inverse]
    if len(bits) > 64:        # a column of many distinct values is worth remembering (the bin wavelengths)
        if len(_COLUMN_TEXT) > 16:
            _COLUMN_TEXT.clear()
        _COLUMN_TEXT[key] = texts
    return texts


def write_two_columns(path, table):
    """The file `np.savetxt(path, table)` writes for an (n, 2) array, byte for byte ('%.18e',
    one blank, newline), without formatting every number anew: a detector spectrum is 3800 rows of
    mostly repeated counts, and ten detectors cost the reference's own job sizes more time in
    `savetxt` than the whole trace takes."""
    table = np.asarray(table, dtype=np.float64)
    if table.ndim != 2 or table.shape[1] != 2 or table.shape[0] == 0:
        np.savetxt(path, table)
        return
    lines = np.char.add(np.char.add(_formatted_column(table[:, 0]), " "), _formatted_column(table[:, 1]))
    with open(path, "w") as fh:
        fh.write("\n".join(lines.tolist()) + "\n")


class LoadSaveRT:
    def __init__(self, Path, RayTracerInstance):
        self.RTI = RayTracerInstance
        self.ActivePath = Path
        self.Materials = []

    def LoadArrays(self):
        job = prt.load_job(self.ActivePath)
        self._take(job)

    def _take(self, job):
        self.Elements = np.array(job["Elements"], dtype=np.float64)
        self.ElementProperties = copy.deepcopy(job["ElementProperties"])
        self.ElementGeometry = list(job["ElementGeometry"])
        self.ElementSize = copy.deepcopy(job["ElementSize"])
        self.Materials = copy.deepcopy(job["Materials"])
        self.PhArray = copy.deepcopy(job["Photon"])
        # not one of the six files: a sun table handed over with an in-memory job (configs.py)
        self.SunTable = job.get("SunTable") if isinstance(job, dict) else None

    def ParsePath(self, Path):
        """Read a job folder and rebuild the tracer's scene from it
        (reference IOOperation.py:73-115)."""
        self.ActivePath = Path
        self.LoadArrays()
        self._build()

    def ParseJob(self, job):
        """Same as ParsePath for a job already in memory."""
        self._take(job)
        self._build()

    def _build(self):
        rti = self.RTI
        for row in self.Materials:
            if row[0] not in rti.MaterialDict:
                rti.MaterialDict[row[0]] = Materials.MaterialObj(**dict(zip(MATERIAL_ROW, row)))
        props = []
        for p in self.ElementProperties:
            if p[0] == 2:
                props.append([2, rti.MaterialDict[p[1]], rti.MaterialDict[p[2]]])
            elif p[0] == 4:
                eqe = 1 if p[3] == "" else Materials.load_table(p[3])
                props.append([4, 0, 0, 0, eqe])
            else:
                props.append(list(p))
        rti.Elements = self.Elements
        rti.ElementGeometry = self.ElementGeometry
        rti.ElementSize = self.ElementSize
        rti.ElementProperties = props
        ph = self.PhArray
        rti.PhArray = ph
        rti.NumberOfPhotons = ph[0]
        rti.Photon = PhObj.Photon(Location=ph[1], EmissionMode=ph[2], EmissionDirection=ph[6],
                                  EmissionWavelengthType=ph[3], StartingWavelength=ph[4],
                                  StartingMaterial=rti.MaterialDict[ph[5]],
                                  RespectPolarization=ph[7], StartingLocation=ph[1],
                                  ScaleFactor=ph[8] if len(ph) == 9 else 1.0)
        rti.Photon.SunTable = getattr(self, "SunTable", None)
        rti.Photon.SunTableFolder = self.ActivePath or None
        rti.lineclr = rti.Photon.StartingMaterial.color
        rti.CreateOpticalElementList()

    def SaveLog(self):
        """Write the job copy and the results (reference IOOperation.py:34-69):
        `Logfile.txt` rows are [Counter, EnergyCounter, SpectralCounter,
        BounceCounter] per element."""
        stamp = datetime.datetime.now().strftime("%Y-%m-%d_%Hh%Mm%Ss")
        folder = os.path.join(self.ActivePath, stamp)
        os.makedirs(folder, exist_ok=True)
        rti = self.RTI
        # The copy of the job holds the element properties as they were read (material and
        # detection-spectrum *names*), so that the log folder is itself a job ParsePath can
        # read; the reference pickles the resolved objects here (IOOperation.py:84-91 replaces
        # the names in place before :44-45 writes them).
        as_read = getattr(self, "ElementProperties", None)
        if as_read is None or len(as_read) != len(rti.ElementProperties):
            as_read = rti.ElementProperties
        prt.dump_job(folder, {"Elements": rti.Elements,
                              "ElementProperties": as_read,
                              "ElementGeometry": rti.ElementGeometry,
                              "ElementSize": rti.ElementSize,
                              "Materials": self.Materials, "Photon": rti.PhArray})
        log = np.zeros((len(rti.OpticalElements), 4))
        for i, oe in enumerate(rti.OpticalElements):
            typ = oe.Type
            if typ.kind == "Dump":
                log[i, :3] = [typ.Counter, typ.EnergyCounter, typ.SpectralCounter]
                write_two_columns(os.path.join(folder, "DetectorSpectrum%d.txt" % i), typ.IncidentSpectrum)
            else:
                log[i, 3] = typ.BounceCounter
        np.savetxt(os.path.join(folder, "Logfile.txt"), log)
        np.savetxt(os.path.join(folder, "AbsorbedPhotons.txt"),
                   np.array([rti.Photon.BeenAbsorbed[1]]))
        self.LastLogFolder = folder
        return folder
