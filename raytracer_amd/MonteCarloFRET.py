"""Foerster-transfer constants of a doped medium.

In the reference an `MCFRET` object is built for every absorption event and
recomputes the transfer probabilities each time (MonteCarloFRET.py:13-40,
195-237) although they depend on the material only.  Here the class computes
them once per material on the host; the per-photon pathway selection and
re-emission (MonteCarloFRET.py:63-193, 258-279) is device code.
"""
from __future__ import annotations

import math

AVOGADRO = 6.022e23      # the reference's constant, MonteCarloFRET.py:206
PER_MMOL_L_TO_PER_CM3 = 6.022e17


def _forster_prefactor(Q, n):
    # 9 (2/3) Q ln10 / (128 n^4 N_A pi^5), kappa^2 = 2/3
    return (9.0 * (2.0 / 3.0) * Q * math.log(10.0)) / (128.0 * (n ** 4) * AVOGADRO * (math.pi ** 5))


def _p_from_concentration(Q, n, overlap, conc):
    """1 - exp(-1.42 c / c0) with c0 the critical concentration of the
    Foerster radius R0 (MonteCarloFRET.py:206-215, 225-237).  The factor
    1e17 converts (L/(mol cm)) nm^4 to nm^6/mol."""
    r0_nm = (_forster_prefactor(Q, n) * overlap * 1e17) ** (1.0 / 6.0)
    r0_cm = r0_nm * 1e-7
    if not r0_cm > 0:
        return 0.0, r0_nm
    c0 = 1.0 / ((4.0 / 3) * math.pi * (r0_cm ** 3))
    return 1.0 - math.exp(-1.42 * (conc * PER_MMOL_L_TO_PER_CM3) / c0), r0_nm


class MCFRET(object):
    """Transfer probabilities and pathway tables of one material.

    Accepts either a material or any object with a `CurrentMaterial`
    attribute (the reference passes the photon)."""

    def __init__(self, Photon, InDonor=True, EmPol=True):
        mat = getattr(Photon, "CurrentMaterial", Photon)
        self.Material = mat
        self.InDonor = InDonor
        self.EmPol = EmPol
        self.c_D, self.c_A = mat.conc_b, mat.conc_a
        self.Q_D, self.Q_A = mat.Q_D, mat.Q_A
        self.tau_D, self.tau_A = mat.tau_D, mat.tau_A
        self.overlDA, self.overlAD = mat.ol_DA, mat.ol_AD
        self.P_ET_DA = self.P_ET_AD = 0.0
        self.R_0_DA = self.R_0_AD = 0.0

    def GenerateMolecules(self):
        mat = self.Material
        # donor -> acceptor: the reference only ever reaches the
        # concentration formula (its fixed-distance branch sits behind a raise,
        # MonteCarloFRET.py:198-202, and Materials.py:32 forces FixDistR != 0)
        self.P_ET_DA, self.R_0_DA = _p_from_concentration(self.Q_D, mat.n, self.overlDA, self.c_A)
        if mat.FixDist:
            # acceptor -> donor at a fixed distance (MonteCarloFRET.py:219-222)
            inv_r6 = 1.0 / ((mat.FixDistR * 1e-9) ** 6)
            k_et = (_forster_prefactor(self.Q_A, mat.n) / self.tau_A) * inv_r6 * 1e-54 \
                * self.overlAD * 1e17
            self.P_ET_AD = k_et / (1.0 / self.tau_A + k_et)
        else:
            self.P_ET_AD, self.R_0_AD = _p_from_concentration(self.Q_A, mat.n, self.overlAD,
                                                              self.c_D)
        return self.P_ET_DA, self.P_ET_AD

    def PathwayProbabilities(self):
        """(p, pA): [fluorescence, internal conversion, transfer] for an
        excited donor / acceptor (MonteCarloFRET.py:67-82)."""
        mat = self.Material
        if mat.conc_a > 0:
            self.GenerateMolecules()
            p = [(1 - self.P_ET_DA) * self.Q_D, (1 - self.P_ET_DA) * (1 - self.Q_D), self.P_ET_DA]
            pA = [(1 - self.P_ET_AD) * self.Q_A, (1 - self.P_ET_AD) * (1 - self.Q_A),
                  self.P_ET_AD]
            return p, pA
        return [mat.Q_D, 1 - mat.Q_D], None
