"""Media: refractive index, dye concentrations and processed spectra.

`MaterialObj` keeps the constructor signature and attribute names of the
reference (Materials.py:6) and produces the same processed tables
(Materials.py:38-74); it runs once per job on the host.  Unlike the
reference it does not register itself on `builtins`.
"""
from __future__ import annotations

import os

import numpy as np

from . import RTUtils as RTU

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_packed = None

# directories searched for spectra files after the working directory
SPECTRA_PATH: list = []


def load_table(name):
    """Read a two-column text table.  Bare file names are looked up in the
    working directory (as the reference does, Materials.py:41-44), then in
    SPECTRA_PATH, then among the dye tables packaged in data/spectra.npz."""
    global _packed
    candidates = [name] + [os.path.join(d, name) for d in SPECTRA_PATH]
    for c in candidates:
        if os.path.isfile(c):
            return np.loadtxt(c)
    if _packed is None:
        _packed = dict(np.load(os.path.join(_DATA, "spectra.npz")))
    key = os.path.splitext(os.path.basename(name))[0]
    if key in _packed:
        return np.array(_packed[key], dtype=np.float64)
    raise IOError("spectrum table %r not found" % (name,))


def _clean_absorption(raw):
    """Materials.py:41-61 for one absorption table: clamp at 0, scale the peak
    to 1, lift values in (0, 2e-4) to 2e-4, zero everything beyond 1.5 x the
    peak wavelength, drop zero rows, close with a zero one nm after the end."""
    t = np.maximum(0.0, np.asarray(raw, dtype=np.float64))
    t[:, 1] /= t[:, 1].max()
    small = (t[:, 1] > 0) & (t[:, 1] < 2e-4)
    t[small, 1] = 2e-4
    peak_wl = t[np.argmax(t[:, 1]), 0]
    t[t[:, 0] > 1.5 * peak_wl, 1] = 0.0
    t = t[t[:, 1] > 0]
    return np.vstack([t, [t[-1, 0] + 1.0, 0.0]])


def _clean_emission(raw):
    """Materials.py:42-70 for one emission table: clamp at 0, zero beyond
    2000 nm, drop non-positive rows, normalise the sum to 1."""
    t = np.maximum(0.0, np.asarray(raw, dtype=np.float64))
    t[t[:, 0] > 2000.0, 1] = 0.0
    t = t[t[:, 1] > 0]
    t[:, 1] /= t[:, 1].sum()
    return t


class MaterialObj:
    def __init__(self, name, n, IsAbsorber=False, conc_a=0, conc_b=0, color="blue",
                 IsAligned=False, AlignVect=[1, 0, 0], AlignS=0.1, R_0=5, Q_D=0.9, Q_A=0.9,
                 tau_D=8, tau_A=8, emax_D=23500, emax_A=54000,
                 sp_abs_D="Coumarin1_Abs.txt", sp_em_D="Coumarin1_Em.txt",
                 sp_abs_A="Coumarin6_Abs.txt", sp_em_A="Coumarin6_Em.txt",
                 AbsEmAngle=0, FixDist=False, FixDistR=1, FixDistNeighbors=1):
        self.name = name
        self.n = n
        self.IsAbsorber = IsAbsorber
        self.conc_a, self.conc_b = conc_a, conc_b      # acceptor / donor, mmol/L
        self.R_0 = R_0
        self.Q_D, self.Q_A = Q_D, Q_A
        self.tau_D, self.tau_A = tau_D, tau_A
        self.color = color
        self.IsAligned, self.AlignVect, self.AlignS = IsAligned, AlignVect, AlignS
        self.AbsEmAngle = AbsEmAngle
        self.emax_D, self.emax_A = emax_D, emax_A
        self.FixDist = FixDist
        self.FixDistR = FixDistR if FixDistR != 0 else 1     # Materials.py:32
        self.FixDistNeighbors = FixDistNeighbors
        self.spectra_files = (sp_abs_D, sp_em_D, sp_abs_A, sp_em_A)
        self.ol_DA = self.ol_AD = 0.0
        if IsAbsorber:
            self.sp_abs_D = _clean_absorption(load_table(sp_abs_D))
            self.sp_em_D = _clean_emission(load_table(sp_em_D))
            self.sp_abs_A = _clean_absorption(load_table(sp_abs_A))
            self.sp_em_A = _clean_emission(load_table(sp_em_A))
            self.ol_DA = self.CalcOverlap(self.sp_em_D, self.sp_abs_A, self.emax_A)
            self.ol_AD = self.CalcOverlap(self.sp_em_A, self.sp_abs_D, self.emax_D)

    def CalcOverlap(self, EmD, AbsA, eA):
        """Foerster overlap sum over integer wavelengths (Materials.py:80-84):
        sum_l em(l) * eA * abs(l) * l^4 for l in [int(min em wl), int(max abs wl)),
        both tables read by nearest neighbour."""
        total = 0.0
        for l in range(int(EmD[:, 0].min()), int(AbsA[:, 0].max())):
            em = EmD[RTU.FindClosestListElement(EmD[:, 0], l), 1]
            ab = AbsA[RTU.FindClosestListElement(AbsA[:, 0], l), 1]
            total += em * (eA * ab) * (l ** 4)
        return total
