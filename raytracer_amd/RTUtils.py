"""Host-side helpers with the names of the reference's RTUtils module.

Only the photon-independent utilities live here (table lookups, frames, scene
construction helpers); they run once while a scene is prepared.  Everything
the reference evaluates per photon (direction samplers, RotationMatrix, the
side test inside the search) is device code in csrc/; `CheckSides` is kept as
a plain predicate for scripts that call an element's `IsValidNextPoint`.
"""
from __future__ import annotations

import numpy as np


def FindClosestListElement(A, target):
    """Index of the entry of the ascending array `A` nearest to `target`.

    Semantics of reference RTUtils.py:15-22: left insertion point clipped to
    [1, len-1], then one step down when the lower neighbour is strictly
    nearer -- so a tie, and any target at or above the upper neighbour, go to
    the upper index, and duplicates resolve to the first duplicate's upper
    side exactly as `searchsorted` places them.  Accepts scalars or arrays."""
    A = np.asarray(A, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    hi = np.clip(np.searchsorted(A, t, side="left"), 1, len(A) - 1)
    nearer_below = (t - A[hi - 1]) < (A[hi] - t)
    return hi - nearer_below.astype(hi.dtype)


def CheckSides(v1, v2, Plane):
    """True if the points `v1` and `v2` lie on the same side of `Plane` =
    [support, normal]; a point on the plane is on a side of its own
    (reference RTUtils.py:6-11 compares the signs of the two plane equations)."""
    support, normal = (np.asarray(x, dtype=np.float64) for x in Plane)
    offset = float(np.dot(support, normal))
    return bool(np.sign(np.dot(normal, np.asarray(v1, dtype=np.float64)) - offset)
                == np.sign(np.dot(normal, np.asarray(v2, dtype=np.float64)) - offset))


def MakeVectAxis(vector):
    """3x3 matrix whose columns are (vector, n1, n2) with n1, n2 unit vectors
    orthogonal to `vector` (reference RTUtils.py:25-42).  The helper vector is
    x unless `vector` equals x exactly, then y.  `vector` is used as given
    (not normalised)."""
    v = np.asarray(vector, dtype=np.float64)
    helper = np.array([1.0, 0.0, 0.0])
    if np.array_equal(v, helper):
        helper = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(v, helper)
    n1 = n1 / np.linalg.norm(n1)
    n2 = np.cross(v, n1)
    n2 = n2 / np.linalg.norm(n2)
    return np.stack([v, n1, n2], axis=1)


def z_axis_frame(vector):
    """`np.roll(MakeVectAxis(vector), 2, axis=1)`: columns (n1, n2, vector).

    The reference applies it as `row_vector . frame` to turn a hemisphere
    sample about +z into one about `vector` (PhotonObject.py:71-72) and as
    `frame . column_vector` for aligned-dye frames (RTUtils.py:51-52)."""
    m = MakeVectAxis(vector)
    return np.stack([m[:, 1], m[:, 2], m[:, 0]], axis=1)


def RectifySpectrum(Spectrum):
    """Sampling weights of a (wavelength, value) table, reference
    RTUtils.py:152-158: every value but the first is multiplied by the
    distance to the previous wavelength, then the column is normalised."""
    s = np.array(Spectrum, dtype=np.float64, copy=True)
    s[1:, 1] *= np.diff(s[:, 0])
    s[:, 1] /= s[:, 1].sum()
    return s


def MakeTriangle(p1, p2, p3):
    """[centroid, unit normal] of a triangle (reference RTUtils.py:126-133)."""
    a, b, c = (np.asarray(p, dtype=np.float64) for p in (p1, p2, p3))
    normal = np.cross(b - a, c - a)
    return [(a + b + c) / 3.0, normal / np.linalg.norm(normal)]


def MakeBox(M=(0, 0, 0), Size=(0.2, 0.2, 0.2), n_x=(1, 0, 0)):
    """Six [support, normal] pairs of a box centred at `M` whose local x axis
    is `n_x` (reference RTUtils.py:135-149; face pair k uses Size[k])."""
    frame = MakeVectAxis(np.asarray(n_x, dtype=np.float64) / np.linalg.norm(n_x))
    centre = np.asarray(M, dtype=np.float64)
    faces = np.zeros((6, 2, 3))
    for i in range(6):
        local = np.zeros(3)
        local[i // 2] = 1.0 if i % 2 == 0 else -1.0
        normal = frame @ local
        normal /= np.linalg.norm(normal)
        faces[i, 1] = normal
        faces[i, 0] = centre + normal * Size[i // 2]
    return faces
