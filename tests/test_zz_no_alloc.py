"""rtmc_run_noalloc (static device scratch, no allocation) against rtmc_run.

Both go through the C ABI; the tallies must be the same integers.  Two calls
in a row, the second with a smaller input, so that whatever the first left in
the scratch array would show."""
import numpy as np
import pytest

import helpers as H
from raytracer_amd import _abi
from raytracer_amd import scene as sc

pytestmark = pytest.mark.gpu

KEYS = ("counter", "spectral", "bounce", "energy_fx", "spectrum")


def _free_device_bytes():
    """hipMemGetInfo of the current device, straight from the HIP runtime."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _same(a, b):
    for key in KEYS:
        assert np.array_equal(a[key], b[key]), key
    for key in _abi.TALLY_SCALARS:
        assert a[key] == b[key], key


def test_noalloc_equals_rtmc_run_twice_in_a_row():
    rt = H.raytrace_for("C3")
    s = rt.Scene()
    try:
        # not resident yet: refused, nothing is allocated behind the caller's back
        with pytest.raises(_abi.RtmcError, match="resident"):
            s.run(10, seed=3, noalloc=True)
        s.run(0, seed=3)                       # uploads the scene
        big = s.run(300000, seed=3, noalloc=True)
        small = s.run(1000, seed=3, first_photon=12345, noalloc=True)
        # a warm call leaves the device's free memory where it was: nothing was allocated (or freed)
        before = _free_device_bytes()
        again = s.run(1000, seed=3, first_photon=12345, noalloc=True)
        assert _free_device_bytes() == before
        _same(again, small)
        _same(big, s.run(300000, seed=3))
        _same(small, s.run(1000, seed=3, first_photon=12345))
        assert big["n_photons"] == 300000 and small["n_photons"] == 1000
        # float64 instantiation and accumulation into a caller's block
        block = sc.new_tallies(s.n_elements)
        s.run(20000, seed=5, precision=_abi.F64, tallies=block, noalloc=True)
        s.run(500, seed=5, first_photon=20000, precision=_abi.F64, tallies=block, noalloc=True)
        _same(sc.tallies_to_dict(*block), s.run(20500, seed=5, precision=_abi.F64))
    finally:
        s.close()


def test_noalloc_after_another_scene_used_the_scratch():
    """Two scenes on one GPU share the scratch array; each call starts clean."""
    a, b = H.raytrace_for("C2").Scene(), H.raytrace_for("C4").Scene()
    try:
        a.run(0), b.run(0)
        ta = a.run(200000, seed=8, noalloc=True)
        tb = b.run(3000, seed=8, noalloc=True)
        ta2 = a.run(700, seed=8, noalloc=True)
        _same(ta, a.run(200000, seed=8))
        _same(tb, b.run(3000, seed=8))
        _same(ta2, a.run(700, seed=8))
    finally:
        a.close(), b.close()


def test_noalloc_refuses_a_tally_block_beyond_the_scratch():
    """126 detectors x the wavelength bins of a spectrum source exceed the 65536 words of
    the static scratch: rtmc_run_noalloc says so, rtmc_run traces the scene
    (its detector histogram then lives in HBM, one copy per workgroup)."""
    from raytracer_amd import RayTrace, configs
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.strip_job(10, strips=30, sun=True))
    s = rt.Scene()
    try:
        t = s.run(50000, seed=2)
        assert t["n_photons"] == 50000 and np.array_equal(t["spectrum"].sum(axis=1), t["counter"])
        with pytest.raises(_abi.RtmcError, match="scratch"):
            s.run(10, noalloc=True)
    finally:
        s.close()
