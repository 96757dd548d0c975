"""ctypes view of include/rtmc.h and loader of csrc/librtmc.so.

There is no fallback: if the shared library is missing or does not export a
declared symbol, importing a function from here raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "librtmc.so")

ABI_VERSION = 1
SPECTRUM_BINS = 3800
SPECTRUM_FIRST_NM = 200
TRACE_STRIDE = 16
ENERGY_FX_SCALE = 16777216.0
F32, F64 = 0, 1

GEOM = {"Rectangle": 1, "Circle": 2, "Cylinder": 3, "Triangle": 4}
TYPE = {"Mirror": 1, "Interface": 2, "Lens": 3, "Dump": 4}
STATUS = ["Start", "Absorbed", "ReemittedDonor", "ReemittedAcceptor", "InternalConverted",
          "Dumped", "Lost"]

EXPORTS = ["rtmc_version", "rtmc_last_error", "rtmc_scene_create", "rtmc_scene_destroy",
           "rtmc_run", "rtmc_last_kernel_ms", "rtmc_step_tape", "rtmc_op_intersect",
           "rtmc_scene_wavelengths", "rtmc_op_crossing", "rtmc_run_noalloc", "rtmc_op_photon",
           "rtmc_scene_info", "rtmc_scene_cells"]

PHOP = {"generate": 1, "absorption": 2, "reflect": 3, "refract": 4, "lens": 5, "polarization": 6,
        "reflectance": 7, "execute": 8, "freepath": 9}          # RTMC_PHOP_*
PHOP_ARGS = 12

c_double3 = C.c_double * 3
c_double9 = C.c_double * 9


class Element(C.Structure):
    _fields_ = [("geometry", C.c_int32), ("type", C.c_int32),
                ("support", c_double3), ("normal", c_double3),
                ("e1", c_double3), ("e2", c_double3),
                ("half1", C.c_double), ("half2", C.c_double),
                ("material1", C.c_int32), ("material2", C.c_int32),
                ("eqe_table", C.c_int32), ("reserved", C.c_int32),
                ("focal_length", C.c_double), ("vertex3", c_double3)]


class Table(C.Structure):
    _fields_ = [("offset", C.c_int64), ("len", C.c_int64)]


class Material(C.Structure):
    _fields_ = [("name_id", C.c_int32), ("is_absorber", C.c_int32),
                ("is_aligned", C.c_int32), ("fix_dist", C.c_int32),
                ("n", C.c_double), ("conc_a", C.c_double), ("conc_b", C.c_double),
                ("emax_A", C.c_double), ("emax_D", C.c_double),
                ("Q_A", C.c_double), ("Q_D", C.c_double),
                ("tau_A", C.c_double), ("tau_D", C.c_double),
                ("fix_dist_R", C.c_double), ("align_vect", c_double3),
                ("align_S", C.c_double), ("abs_em_angle", C.c_double),
                ("ol_DA", C.c_double), ("ol_AD", C.c_double),
                ("P_ET_DA", C.c_double), ("P_ET_AD", C.c_double),
                ("psi_P", C.c_double), ("psi_Q", C.c_double), ("align_frame", c_double9),
                ("abs_A", Table), ("abs_D", Table), ("em_A", Table), ("em_D", Table)]


class Source(C.Structure):
    _fields_ = [("direction_mode", C.c_int32), ("wavelength_mode", C.c_int32),
                ("location", c_double3), ("direction", c_double3), ("frame", c_double9),
                ("wavelength", C.c_double), ("spectrum", Table),
                ("start_material", C.c_int32), ("respect_polarization", C.c_int32),
                ("scale_factor", C.c_double), ("area", C.c_double * 2)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_elements", C.c_int32),
                ("n_materials", C.c_int32), ("n_tables", C.c_int32),
                ("elements", C.POINTER(Element)), ("materials", C.POINTER(Material)),
                ("tables", C.POINTER(Table)), ("table_data", C.POINTER(C.c_double)),
                ("table_data_len", C.c_int64), ("source", Source),
                ("max_iterations", C.c_int32), ("reserved", C.c_int32)]


_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)

TALLY_SCALARS = ["n_photons", "n_steps", "n_absorbed_once", "n_absorptions",
                 "n_internal_converted", "n_dumped", "n_capped", "n_lost", "n_stale_steps"]
TALLY_ARRAYS = ["counter", "spectral", "bounce", "energy_fx"]


class Tallies(C.Structure):
    _fields_ = ([("n_elements", C.c_int32), ("reserved", C.c_int32)]
                + [(s, C.c_uint64) for s in TALLY_SCALARS]
                + [(a, _u64p) for a in TALLY_ARRAYS]
                + [("spectrum", _u64p)])


class PhotonSoA(C.Structure):
    _fields_ = [("loc", _f64p * 3), ("dir", _f64p * 3), ("efield", _f64p * 3),
                ("old_point", _f64p * 3),
                ("wavelength", _f64p), ("start_wavelength", _f64p), ("dist_left", _f64p),
                ("hit_distance", _f64p), ("psi", _f64p),
                ("material", _i32p), ("status", _i32p), ("last_element", _i32p),
                ("was_element", _i32p), ("reflected", _i32p), ("iterations", _i32p),
                ("tape_pos", _i32p)]


def bind(lib):
    """Attach argument/return types to every export of include/rtmc.h."""
    lib.rtmc_version.restype = C.c_int
    lib.rtmc_version.argtypes = []
    lib.rtmc_last_error.restype = C.c_char_p
    lib.rtmc_last_error.argtypes = []
    lib.rtmc_scene_create.restype = C.c_int
    lib.rtmc_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
    lib.rtmc_scene_destroy.restype = None
    lib.rtmc_scene_destroy.argtypes = [C.c_void_p]
    lib.rtmc_run.restype = C.c_int
    lib.rtmc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                             C.POINTER(Tallies)]
    lib.rtmc_run_noalloc.restype = C.c_int
    lib.rtmc_run_noalloc.argtypes = lib.rtmc_run.argtypes
    lib.rtmc_last_kernel_ms.restype = C.c_double
    lib.rtmc_last_kernel_ms.argtypes = [C.c_void_p, C.c_int]
    lib.rtmc_step_tape.restype = C.c_int
    lib.rtmc_step_tape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32,
                                   C.POINTER(PhotonSoA), _f64p, _u64p, C.c_uint32, C.c_int,
                                   _f64p]
    lib.rtmc_op_intersect.restype = C.c_int
    lib.rtmc_op_intersect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, _f64p, _f64p,
                                      _f64p]
    lib.rtmc_op_crossing.restype = C.c_int
    lib.rtmc_op_crossing.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(PhotonSoA),
                                     _i32p, _f64p, _f64p]
    lib.rtmc_op_photon.restype = C.c_int
    lib.rtmc_op_photon.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(PhotonSoA),
                                   _f64p, _f64p, _f64p, _u64p, C.c_int]
    lib.rtmc_scene_wavelengths.restype = C.c_int
    lib.rtmc_scene_wavelengths.argtypes = [C.c_void_p, C.POINTER(C.c_int64), _f64p, C.c_int64]
    lib.rtmc_scene_info.restype = C.c_int
    lib.rtmc_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.rtmc_scene_cells.restype = C.c_int
    lib.rtmc_scene_cells.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    return lib


_lib = None


class RtmcError(RuntimeError):
    pass


def load_library(path=None):
    """Load librtmc.so (built by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTMC_LIBRARY") or LIB_PATH   # RTMC_LIBRARY: another build, for A/B runs
    if not os.path.isfile(p):
        raise RtmcError("HIP library %s is missing: build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'`" % p)
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RtmcError("librtmc.so does not export %s" % name)
    bind(lib)
    if lib.rtmc_version() != ABI_VERSION:
        raise RtmcError("librtmc.so ABI version %d, expected %d"
                        % (lib.rtmc_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        msg = lib.rtmc_last_error()
        raise RtmcError("rtmc error %d: %s" % (rc, (msg or b"").decode("utf-8", "replace")))


def f64_ptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f64p)


def u64_ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def i32_ptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i32p)
