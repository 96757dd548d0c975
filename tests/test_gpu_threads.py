"""Host threads sharing one GPU: the library runs their calls one after the
other, so concurrent rtmc_run calls on one scene give the sequential sums."""
import threading

import numpy as np
import pytest

import helpers as H
from raytracer_amd import _abi
from raytracer_amd import scene as sc

pytestmark = pytest.mark.gpu


def test_two_threads_on_one_device_equal_sequential_calls():
    s = H.raytrace_for("C3").Scene()
    other = H.raytrace_for("C2").Scene()
    try:
        n, rounds = 200000, 6
        expect = [s.run(n, seed=3, first_photon=i * n) for i in range(2 * rounds)]
        expect_other = other.run(n, seed=4)
        results, errors = {}, []

        def worker(tid):
            try:
                for r in range(rounds):
                    i = 2 * r + tid
                    # ctypes releases the GIL for the duration of the call
                    results[i] = s.run(n, seed=3, first_photon=i * n, noalloc=(r % 2 == 1))
            except Exception as exc:   # noqa: BLE001
                errors.append(exc)

        def worker_other():
            try:
                for r in range(rounds):
                    results[("o", r)] = other.run(n, seed=4, noalloc=(r % 2 == 0))
            except Exception as exc:   # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(t,)) for t in (0, 1)]
        threads.append(threading.Thread(target=worker_other))
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        keys = list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]
        for i in range(2 * rounds):
            for k in keys:
                assert np.array_equal(np.asarray(results[i][k]), np.asarray(expect[i][k])), (i, k)
        for r in range(rounds):
            for k in keys:
                assert np.array_equal(np.asarray(results[("o", r)][k]), np.asarray(expect_other[k])), (r, k)
    finally:
        s.close(), other.close()


def test_a_small_scene_does_not_shrink_the_lds_of_a_large_one():
    """The LDS limit is an attribute of the kernel function, shared by every scene of the process:
    a scene whose tables need more than the 64 KB default (the sun-spectrum source with the full
    feature set, 80 KB and more), then a small one on kernels they share, then the large one again."""
    import helpers as H
    from raytracer_amd import _abi
    from raytracer_amd import scene as sc
    big_rt, small_rt = H.raytrace_for("C5"), H.raytrace_for("C2")
    big, small = big_rt.Scene(), small_rt.Scene()
    try:
        first = big.run(200000, seed=4)
        # the step-wise kernels are the same functions for every scene
        b = sc.new_photon_batch(4)
        tape = np.full(4 * 64, 0.5)
        off = np.arange(5, dtype=np.uint64) * 64
        big.step_tape(b, tape, off, 3, launch=True, precision=_abi.F32)
        small.run(1000, seed=4)
        b2 = sc.new_photon_batch(4)
        small.step_tape(b2, tape, off, 3, launch=True, precision=_abi.F32)
        again = big.run(200000, seed=4)
        b3 = sc.new_photon_batch(4)
        big.step_tape(b3, tape, off, 3, launch=True, precision=_abi.F32)
        assert np.array_equal(first["counter"], again["counter"]) and first["n_steps"] == again["n_steps"]
        assert np.array_equal(b["loc"], b3["loc"]) and np.array_equal(b["status"], b3["status"])
    finally:
        big.close(), small.close()
