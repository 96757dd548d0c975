"""The reference's own statistical acceptance check on the device samplers.

TESTEMISSION (reference MonteCarloFRET.py:283-308) draws isotropic emission
directions (RTUtils.py:102-108 GenerateDirectionSine) and counts those outside the
escape cone of n = 1.5: the fraction must be cos(41.81 deg) = 0.7454.  Here the same
count is made three ways, each on 4e6 directions and held to four standard
deviations (and to the reference's own four digits):

  * launches of the hemispherical source in Philox mode (`sphere_direction` folded
    into the hemisphere, PhotonObject.py:68-86), counted by detectors that cover the cone;
  * re-emissions in Philox mode (the same sampler in the renew block of the trace
    kernel), from a source that sits in a thin dye medium;
  * re-emissions through the tape path with the reference's construction (three
    normals, normalised) from uniforms supplied by the host.
"""
import numpy as np
import pytest

from raytracer_amd import RayTrace, _abi, configs
from raytracer_amd import scene as sc

pytestmark = pytest.mark.gpu

N = 4000000
EXPECTED = configs.ESCAPE_CONE_COS      # 0.745356; the reference quotes 0.7454


def _scene(emission):
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.escape_cone_job(10, emission=emission))
    return rt, rt.Scene()


def _check(outside, total, label):
    frac = outside / total
    sigma = np.sqrt(EXPECTED * (1 - EXPECTED) / total)
    print("%s: %d of %d outside the cone = %.5f (expected %.5f, %+.2f sigma)"
          % (label, outside, total, frac, EXPECTED, (frac - EXPECTED) / sigma))
    assert abs(frac - EXPECTED) < 4 * sigma, (label, frac, EXPECTED, sigma)
    assert abs(frac - 0.7454) < 1e-3          # the figure the reference quotes


@pytest.mark.parametrize("precision", [_abi.F32, _abi.F64])
def test_escape_cone_fraction_of_philox_launches(precision, oracle_mod):
    rt, s = _scene(False)
    t = s.run(N, seed=11, precision=precision)
    assert int(t["counter"].sum()) == N and int(t["counter"][1]) == 0     # nothing goes up from a downward hemisphere
    _check(N - int(t["counter"][0]), N, "launch, Philox")
    if precision == _abi.F64:      # the checker on the same variates
        o = oracle_mod.OracleScene.from_raytrace(rt).run(200000, seed=11)
        g = s.run(200000, seed=11, precision=precision)
        assert np.abs(o["counter"].astype(np.int64) - g["counter"].astype(np.int64)).sum() <= 2
    s.close()


@pytest.mark.parametrize("precision", [_abi.F32, _abi.F64])
def test_escape_cone_fraction_of_philox_emissions(precision, oracle_mod):
    rt, s = _scene(True)
    t = s.run(N, seed=12, precision=precision)
    emitted = int(t["counter"].sum())
    assert emitted + int(t["n_internal_converted"]) == N and int(t["n_absorbed_once"]) == N
    assert abs(emitted / N - 0.95) < 4 * np.sqrt(0.95 * 0.05 / N)          # quantum yield of the acceptor
    _check(emitted - int(t["counter"][0]) - int(t["counter"][1]), emitted, "emission, Philox")
    assert abs(int(t["counter"][0]) - int(t["counter"][1])) < 5 * np.sqrt(emitted * 0.25)   # up and down alike
    if precision == _abi.F64:
        o = oracle_mod.OracleScene.from_raytrace(rt).run(200000, seed=12)
        g = s.run(200000, seed=12, precision=precision)
        assert np.abs(o["counter"].astype(np.int64) - g["counter"].astype(np.int64)).sum() <= 2
    s.close()


def test_escape_cone_fraction_of_tape_emissions():
    """CheckForAbsorption on photons that sit in the dye with no free path left: donor/acceptor
    coin, pathway, direction from three normals (RTUtils.py:102-108), polarisation, wavelength,
    free path -- all from uniforms of numpy's generator, in the reference's order."""
    rt, s = _scene(True)
    dye = next(i for i, m in enumerate(s.packed.materials) if m.IsAbsorber)
    rng = np.random.RandomState(5)
    per, batches = 16, 4
    n = N // batches
    outside = emitted = 0
    for _ in range(batches):
        b = sc.new_photon_batch(n)
        b["loc"][2, :] = 1.0
        b["dir"][2, :] = -1.0
        b["efield"][0, :] = 1.0
        b["wavelength"][:] = b["start_wavelength"][:] = 460.0
        b["material"][:] = dye
        b["hit_distance"][:] = 1.0
        args = np.zeros((n, _abi.PHOP_ARGS))
        tape = rng.random_sample(n * per)
        off = np.arange(n + 1, dtype=np.uint64) * per
        s.photon_op("absorption", b, args, tape, off, uniform_tape=True, precision=_abi.F32)
        lit = b["status"] == 3                      # ReemittedAcceptor
        assert set(np.unique(b["status"])) <= {3, 4}
        assert np.abs(np.sqrt((b["dir"][:, lit] ** 2).sum(axis=0)) - 1).max() < 1e-6
        emitted += int(lit.sum())
        outside += int((np.abs(b["dir"][2, lit]) < EXPECTED).sum())
    _check(outside, emitted, "emission, tape")
    s.close()
