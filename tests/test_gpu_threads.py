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
