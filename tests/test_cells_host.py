"""Which closed boxes ("cells") scene creation finds among a scene's rectangles, and which scenes run
the kernels that use them (host logic of rtmc_scene_create: no GPU needed)."""
import helpers as H
from raytracer_amd import RayTrace, configs


def _scene(job):
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(job)
    return rt, rt.Scene()


def test_cells_found_per_configuration():
    expect = {"C1": (2, 2, 2, True), "C2": (2, 2, 2, True), "C3": (2, 2, 2, True), "C4": (4, 3, 4, True),
              "C5": (2, 2, 2, True), "C6": (2, 2, 2, True), "C1_open": (1, 1, 0, True),
              "C8": None, "C2r": (2, 2, 2, True)}       # C2r: the same boxes in a turned frame
    for name, want in expect.items():
        s = H.raytrace_for(name).Scene()
        try:
            c = s.cells()
            if want is None:
                assert not c["in_use"], (name, c)       # other shapes: the kernels with cells do not apply
            else:
                assert (c["cells"], c["usable"], c["source_cell"], c["in_use"]) == want, (name, c)
        finally:
            s.close()
    for job, want in ((configs.nested_job(), (3, 3, True)), (configs.twin_job(), (3, 2, True)), (configs.mirror_box_job(), (2, 2, True)),
                      (configs.near_floor_job(), (2, 0, False)),
                      (configs.tilted_slab_job(), (2, 2, True)),            # a turned slab in an upright box
                      (configs.strip_job(10, strips=14), (2, 2, True)),      # edge faces tiled by 14 detectors each
                      (configs.strip_job(10, strips=30), (2, 2, True)),      # ... by 30: detector histogram in HBM
                      (configs.stacked_job(10, n_slabs=8), (9, 8, True)),    # runs in the hit table beside the cells
                      (configs.flush_detector_job(10), None), (configs.tiled_floor_job(10), None)):
        rt, s = _scene(job)
        try:
            c = s.cells()
            if want is None:   # coplanar overlaps (guard distance): other kernels
                assert not c["in_use"], c
            else:
                assert (c["cells"], c["usable"], c["in_use"]) == want, c
                assert c["source_cell"] != 0     # the source sits in the outermost box
        finally:
            s.close()


def test_cells_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("RTMC_NO_CELLS", "1")
    s = H.raytrace_for("C2").Scene()
    try:
        assert s.cells() == {"cells": 0, "usable": 0, "source_cell": 0, "in_use": False, "kernel_features": 0}
    finally:
        s.close()
