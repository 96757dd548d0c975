"""Element shapes (host side).

The classes keep the reference's constructor signatures and attribute names
(Geometries.py) and hold what the device needs: for a plane-like shape the
support point, unit normal, the two unit in-plane axes and half extents.
Intersection itself is device code; `FindIntersections` is offered for
single-ray checks and goes through the same kernels.
"""
from __future__ import annotations

import numpy as np

from . import RTUtils as RTU


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


class _Shape:
    kind = "None"
    Photon = None

    def FindIntersections(self):
        """Intersection of the photon's line with this shape, or three NaNs
        (reference Geometries.py:32-52 for the rectangle).  Evaluated on the
        GPU by the same routine the trace kernel uses."""
        from . import scene as _scene
        pts = _scene.intersect_single(self, self.Photon.Location, self.Photon.Direction)
        return pts


class Rectangle(_Shape):
    """Bounded plane.  `Height` is the half extent along `e1`, `Width` along
    `e2` (Geometries.py:43-46).  Without an explicit in-plane axis the pair
    is chosen from three axis-permuted candidates exactly as the reference
    does (Geometries.py:17-29)."""
    kind = "Rectangle"

    def __init__(self, Height, Width, SupportVector, DirectionVector, Photon=None,
                 NormalVector=[0, 0, 0]):
        self.Height, self.Width = float(Height), float(Width)
        self.SupportVector = np.asarray(SupportVector, dtype=np.float64)
        self.DirectionVector = _unit(DirectionVector)
        self.Photon = Photon
        n = self.DirectionVector
        axis = np.asarray(NormalVector, dtype=np.float64)
        if not axis.any():
            a = np.array([-n[1], n[0], 0.0])
            b = np.array([0.0, -n[2], n[1]])
            spare = np.array([-n[2], 0.0, n[0]])
            if not a.any():
                a = spare
            elif not b.any():
                b = spare
            elif np.array_equal(a, b) or np.array_equal(a, -b):
                b = spare
        else:
            a = axis
            b = np.cross(a, n)
        self.e1, self.e2 = _unit(a), _unit(b)

    def OperationPlane(self):
        return [self.SupportVector, self.DirectionVector]


class Circle(_Shape):
    kind = "Circle"

    def __init__(self, Radius, SupportVector, DirectionVector, Photon=None):
        self.Radius = float(Radius)
        self.SupportVector = np.asarray(SupportVector, dtype=np.float64)
        self.DirectionVector = _unit(DirectionVector)
        self.Photon = Photon

    def OperationPlane(self):
        return [self.SupportVector, self.DirectionVector]


class Cylinder(_Shape):
    """Finite tube between two axis points (Geometries.py:90-95); here
    `SupportVector`/`DirectionVector` are the two end points."""
    kind = "Cylinder"

    def __init__(self, Radius, SupportVector, DirectionVector, Photon=None):
        self.Radius = float(Radius)
        self.SupportVector = np.asarray(SupportVector, dtype=np.float64)
        self.DirectionVector = np.asarray(DirectionVector, dtype=np.float64)
        self.Photon = Photon

    def OperationPlane(self):
        """Tangent plane at the photon's candidate point `Photon.PointCy`
        (Geometries.py:138-153, 207-209): the normal points from that point to
        its foot on the axis."""
        point = np.asarray(self.Photon.PointCy, dtype=np.float64)
        axis = _unit(self.DirectionVector - self.SupportVector)
        foot = self.SupportVector + np.dot(point - self.SupportVector, axis) * axis
        return [point, _unit(foot - point)]


class Triangle(_Shape):
    kind = "Triangle"

    def __init__(self, Point1, Point2, Point3, Photon=None):
        self.TrianglePoint1 = np.asarray(Point1, dtype=np.float64)
        self.TrianglePoint2 = np.asarray(Point2, dtype=np.float64)
        self.TrianglePoint3 = np.asarray(Point3, dtype=np.float64)
        self.Photon = Photon
        self.SupportVector, self.DirectionVector = RTU.MakeTriangle(Point1, Point2, Point3)

    def OperationPlane(self):
        return [self.SupportVector, self.DirectionVector]
