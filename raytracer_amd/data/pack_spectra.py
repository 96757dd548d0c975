#!/usr/bin/env python3
"""Pack the dye spectra tables into spectra.npz.

The four two-column text tables (wavelength [nm], value) are measurement data
distributed with the reference next to its sources (LumogenRed{Abs,Em}.txt,
Oxacin{Abs,Em}.txt).  Every benchmark configuration needs them, also on
machines where the reference checkout is absent, so their numbers are kept
here as float64 arrays exactly as `numpy.loadtxt` returns them (no cleaning:
that is `Materials.MaterialObj`'s job).  Run in the build container only.
"""
import os
import sys

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
NAMES = ["LumogenRedAbs", "LumogenRedEm", "OxacinAbs", "OxacinEm"]
out = {n: np.loadtxt(os.path.join(SRC, n + ".txt")) for n in NAMES}
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "spectra.npz"), **out)
for n, a in out.items():
    print(n, a.shape, a[:2].tolist())
