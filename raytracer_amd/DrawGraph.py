"""Placeholder for the reference's matplotlib scene viewer.

Drawing every ray segment from inside the photon loop (reference
RayTrace.py:160-161,174,191) has no place in a batched device run; the class
exists so that scripts constructing it keep working, and does nothing.
"""


class DrawElements:
    def __init__(self, OpticalElements=None):
        self.OpticalElements = OpticalElements

    def DrawElementsPoly(self):
        return None

    def DrawLine(self, start, end, color="black"):
        return None

    def DrawPoint(self, point):
        return None
