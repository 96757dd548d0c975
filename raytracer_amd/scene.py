"""Scene packing and the thin object wrapper around the C ABI.

`Scene` turns the host objects (optical elements, materials, photon source)
into the plain arrays of `rtmc_scene_desc`, owns the library handle and
offers `run`, `step_tape` and `intersect` with numpy in / numpy out.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from . import RTUtils as RTU
from .MonteCarloFRET import MCFRET

REFERENCE_ITERATION_CAP = 10001   # `i >= 1e4` tested after the step, i from 0


class TablePool:
    def __init__(self):
        self.chunks = []
        self.n = 0

    def add(self, table):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.float64))
        if t.ndim != 2 or t.shape[1] != 2:
            raise ValueError("a table must have two columns")
        ref = _abi.Table(self.n, t.shape[0])
        self.chunks.append(t.reshape(-1))
        self.n += t.size
        return ref

    def data(self):
        if not self.chunks:
            return np.zeros(1, dtype=np.float64)
        return np.ascontiguousarray(np.concatenate(self.chunks))


def collect_materials(optical_elements, photon, material_dict=None):
    mats = []

    def add(m):
        if m is not None and not any(m is x for x in mats):
            mats.append(m)
    if material_dict:
        for m in material_dict.values():
            add(m)
    add(photon.StartingMaterial)
    for oe in optical_elements:
        if oe.Type.kind == "Interface":
            add(oe.Type.MaterialSide1)
            add(oe.Type.MaterialSide2)
    return mats


def orientation_special_values(S):
    """The six special-function values LambertBeerFullIntegral evaluates for
    an order parameter S (reference PhotonObject.py:273): erf(a -/+ i b1),
    erfi(b1), erf(a -/+ i b2), erfi(b2) with a = 2.2214.../S, b1 = S/sqrt(2),
    b2 = 3 S/sqrt(2).  scipy.special is the library the reference itself calls."""
    from scipy import special
    a = 2.221441469079183 / S
    b1, b2 = 0.7071067811865475 * S, 2.1213203435596424 * S
    return (special.erf(a - 1j * b1), special.erf(a + 1j * b1), special.erfi(b1),
            special.erf(a - 1j * b2), special.erf(a + 1j * b2), special.erfi(b2))


def orientation_integral_constants(S):
    """(P, Q) with LambertBeerFullIntegral(n) = |P (x^2 + y^2) + Q z^2|.

    The reference's expression (PhotonObject.py:273) is linear in x^2 + y^2 and
    z^2; because erf(conj z) = conj erf(z) the imaginary prefactors multiply
    purely imaginary differences and both coefficients are real."""
    if not S > 1e-10:                      # PhotonObject.py:272-276: ol = z^2
        return 0.0, 1.0
    import numpy as np_
    em1, ep1, ei1, em2, ep2, ei2 = orientation_special_values(S)
    c1, c2 = 0.2349964007466563j, 0.07833213358221876j
    e1, e2 = np_.exp(-0.5 * S), np_.exp(-4.5 * S ** 2)
    P = (e1 * (-c1 * em1 + c1 * ep1 + 0.4699928014933126 * ei1)
         + e2 * (c2 * em2 - c2 * ep2 - 0.15666426716443752 * ei2)) / S
    Q = (e1 * ((2 / 3.0) * (-c1 * em1 + c1 * ep1) + 0.313328534328875 * ei1)
         + e2 * (-2 * (c2 * em2 - c2 * ep2) + 0.31332853432887503 * ei2)) / S
    assert abs(P.imag) < 1e-12 * max(1.0, abs(P)) and abs(Q.imag) < 1e-12 * max(1.0, abs(Q))
    return float(P.real), float(Q.real)


def pack_material(m, name_ids, pool):
    out = _abi.Material()
    out.name_id = name_ids.setdefault(m.name, len(name_ids))
    out.is_absorber = int(bool(m.IsAbsorber))
    out.is_aligned = int(bool(m.IsAligned))
    out.fix_dist = int(bool(m.FixDist))
    out.n = float(m.n)
    out.conc_a, out.conc_b = float(m.conc_a), float(m.conc_b)
    out.emax_A, out.emax_D = float(m.emax_A), float(m.emax_D)
    out.Q_A, out.Q_D = float(m.Q_A), float(m.Q_D)
    out.tau_A, out.tau_D = float(m.tau_A), float(m.tau_D)
    out.fix_dist_R = float(m.FixDistR)
    av = np.zeros(3)
    av[:] = np.asarray(m.AlignVect, dtype=np.float64)[:3]
    out.align_vect = _abi.c_double3(*av)
    out.align_S = float(m.AlignS)
    out.abs_em_angle = float(m.AbsEmAngle)
    if m.IsAbsorber and m.IsAligned:
        out.psi_P, out.psi_Q = orientation_integral_constants(float(m.AlignS))
        out.align_frame = _abi.c_double9(*RTU.z_axis_frame(av).reshape(-1))
    if m.IsAbsorber:
        out.ol_DA, out.ol_AD = float(m.ol_DA), float(m.ol_AD)
        if m.conc_a > 0:
            fret = MCFRET(m)
            out.P_ET_DA, out.P_ET_AD = fret.GenerateMolecules()
        out.abs_A = pool.add(m.sp_abs_A)
        out.abs_D = pool.add(m.sp_abs_D)
        out.em_A = pool.add(m.sp_em_A)
        out.em_D = pool.add(m.sp_em_D)
    return out


def pack_element(oe, mats, pool, eqe_tables):
    g, t = oe.Geometry, oe.Type
    e = _abi.Element()
    e.geometry = _abi.GEOM[g.kind]
    e.type = _abi.TYPE[t.kind]
    e.material1 = e.material2 = -1
    e.eqe_table = -1
    if g.kind == "Rectangle":
        e.support = _abi.c_double3(*g.SupportVector)
        e.normal = _abi.c_double3(*g.DirectionVector)
        e.e1 = _abi.c_double3(*g.e1)
        e.e2 = _abi.c_double3(*g.e2)
        e.half1, e.half2 = g.Height, g.Width
    elif g.kind == "Circle":
        e.support = _abi.c_double3(*g.SupportVector)
        e.normal = _abi.c_double3(*g.DirectionVector)
        e.half1 = e.half2 = g.Radius
    elif g.kind == "Triangle":
        e.support = _abi.c_double3(*g.SupportVector)
        e.normal = _abi.c_double3(*g.DirectionVector)
        e.e1 = _abi.c_double3(*g.TrianglePoint1)
        e.e2 = _abi.c_double3(*g.TrianglePoint2)
        e.vertex3 = _abi.c_double3(*g.TrianglePoint3)
    elif g.kind == "Cylinder":
        e.support = _abi.c_double3(*g.SupportVector)
        e.normal = _abi.c_double3(*g.DirectionVector)
        e.half1 = e.half2 = g.Radius
    if t.kind == "Interface":
        e.material1 = next(i for i, m in enumerate(mats) if m is t.MaterialSide1)
        e.material2 = next(i for i, m in enumerate(mats) if m is t.MaterialSide2)
    elif t.kind == "Lens":
        e.focal_length = float(t.FocalLength)
    elif t.kind == "Dump":
        if not isinstance(t.DetectionSpectrum, (int, np.integer)):
            e.eqe_table = len(eqe_tables)
            eqe_tables.append(pool.add(t.DetectionSpectrum))
    return e


def pack_source(photon, mats, pool):
    s = _abi.Source()
    mode = photon.EmissionMode
    if mode == "Directional Strict":
        s.direction_mode = 0
    elif mode == "Directional Isotropic":
        s.direction_mode = 1
    else:
        raise ValueError("unknown EmissionMode %r" % (mode,))
    d = np.asarray(photon.EmissionDirection, dtype=np.float64)
    s.direction = _abi.c_double3(*d)
    s.location = _abi.c_double3(*np.asarray(photon.StartingLocation, dtype=np.float64))
    frame = RTU.z_axis_frame(d) if s.direction_mode == 1 else np.eye(3)
    s.frame = _abi.c_double9(*frame.reshape(-1))
    if photon.EmissionWavelengthType == "Sun spectrum AG15g":
        s.wavelength_mode = 1
        s.spectrum = pool.add(photon.sun_table())
        s.wavelength = 0.0
    else:
        s.wavelength_mode = 0
        s.wavelength = float(photon.StartingWavelength)
    s.start_material = next(i for i, m in enumerate(mats) if m is photon.StartingMaterial)
    s.respect_polarization = int(bool(photon.RespectPolarization))
    s.scale_factor = float(photon.ScaleFactor)
    area = getattr(photon, "LaunchArea", None) or (0.0, 0.0)
    s.area = (C.c_double * 2)(float(area[0]), float(area[1]))
    return s


class PackedScene:
    """SceneDesc plus the buffers it points into (kept alive together)."""

    def __init__(self, optical_elements, photon, material_dict=None,
                 max_iterations=REFERENCE_ITERATION_CAP):
        self.materials = collect_materials(optical_elements, photon, material_dict)
        pool = TablePool()
        name_ids: dict = {}
        self.mat_array = (_abi.Material * len(self.materials))(
            *[pack_material(m, name_ids, pool) for m in self.materials])
        eqe: list = []
        self.el_array = (_abi.Element * len(optical_elements))(
            *[pack_element(oe, self.materials, pool, eqe) for oe in optical_elements])
        self.table_array = (_abi.Table * max(1, len(eqe)))(*eqe)
        source = pack_source(photon, self.materials, pool)
        self.table_data = pool.data()
        d = _abi.SceneDesc()
        d.abi_version = _abi.ABI_VERSION
        d.n_elements = len(optical_elements)
        d.n_materials = len(self.materials)
        d.n_tables = len(eqe)
        d.elements = C.cast(self.el_array, C.POINTER(_abi.Element))
        d.materials = C.cast(self.mat_array, C.POINTER(_abi.Material))
        d.tables = C.cast(self.table_array, C.POINTER(_abi.Table))
        d.table_data = _abi.f64_ptr(self.table_data)
        d.table_data_len = self.table_data.size
        d.source = source
        d.max_iterations = int(max_iterations)
        self.desc = d
        self.n_elements = len(optical_elements)


def new_tallies(n_elements):
    """Zeroed host tally block: (ctypes struct, dict of the numpy arrays it points to)."""
    arrays = {k: np.zeros(n_elements, dtype=np.uint64) for k in _abi.TALLY_ARRAYS}
    arrays["spectrum"] = np.zeros(n_elements * _abi.SPECTRUM_BINS, dtype=np.uint64)
    t = _abi.Tallies()
    t.n_elements = n_elements
    for k, a in arrays.items():
        setattr(t, k, _abi.u64_ptr(a))
    return t, arrays


def tallies_to_dict(t, arrays):
    out = {k: int(getattr(t, k)) for k in _abi.TALLY_SCALARS}
    for k in _abi.TALLY_ARRAYS:
        out[k] = arrays[k].copy()
    out["spectrum"] = arrays["spectrum"].reshape(t.n_elements, _abi.SPECTRUM_BINS).copy()
    out["energy"] = arrays["energy_fx"].astype(np.float64) / _abi.ENERGY_FX_SCALE
    return out


SOA_VEC = ["loc", "dir", "efield", "old_point"]
SOA_F64 = ["wavelength", "start_wavelength", "dist_left", "hit_distance", "psi"]
SOA_I32 = ["material", "status", "last_element", "was_element", "reflected", "iterations",
           "tape_pos"]


def new_photon_batch(n):
    """dict of numpy arrays for `n` photons (vectors are (3, n), C order)."""
    b = {k: np.zeros((3, n), dtype=np.float64) for k in SOA_VEC}
    b.update({k: np.zeros(n, dtype=np.float64) for k in SOA_F64})
    b.update({k: np.zeros(n, dtype=np.int32) for k in SOA_I32})
    b["last_element"][:] = -1
    return b


def batch_struct(b):
    s = _abi.PhotonSoA()
    for k in SOA_VEC:
        setattr(s, k, (C.POINTER(C.c_double) * 3)(*[_abi.f64_ptr(b[k][i]) for i in range(3)]))
    for k in SOA_F64:
        setattr(s, k, _abi.f64_ptr(b[k]))
    for k in SOA_I32:
        setattr(s, k, _abi.i32_ptr(b[k]))
    return s


class Scene:
    """A scene living in librtmc.so."""

    def __init__(self, optical_elements, photon, material_dict=None,
                 max_iterations=REFERENCE_ITERATION_CAP, lib=None):
        self.lib = lib or _abi.load_library()
        self.packed = PackedScene(optical_elements, photon, material_dict, max_iterations)
        self.n_elements = self.packed.n_elements
        handle = C.c_void_p()
        _abi.check(self.lib.rtmc_scene_create(C.byref(self.packed.desc), C.byref(handle)),
                   self.lib)
        self.handle = handle

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rtmc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, n_photons, seed=0, first_photon=0, device=0, precision=_abi.F32,
            tallies=None, noalloc=False):
        """Trace photon ids [first_photon, first_photon + n_photons) and return
        the tallies as a dict (added to `tallies` = (struct, arrays) if given).
        `noalloc`: through rtmc_run_noalloc (the scene must be resident on the
        device already; device tallies in the library's static scratch)."""
        t, arrays = tallies if tallies is not None else new_tallies(self.n_elements)
        entry = self.lib.rtmc_run_noalloc if noalloc else self.lib.rtmc_run
        _abi.check(entry(self.handle, int(device), int(precision),
                         C.c_uint64(seed), C.c_uint64(first_photon),
                         C.c_uint64(n_photons), C.byref(t)), self.lib)
        return tallies_to_dict(t, arrays)

    def last_kernel_ms(self, device=0):
        return float(self.lib.rtmc_last_kernel_ms(self.handle, int(device)))

    def step_tape(self, batch, tape, tape_off, n_steps, launch=False, device=0,
                  precision=_abi.F32, trace=False):
        n = batch["wavelength"].shape[0]
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        tape_off = np.ascontiguousarray(tape_off, dtype=np.uint64)
        assert tape_off.shape[0] == n + 1
        tr = np.full((n, n_steps, _abi.TRACE_STRIDE), np.nan) if trace else None
        s = batch_struct(batch)
        _abi.check(self.lib.rtmc_step_tape(
            self.handle, int(device), int(precision), n, C.byref(s), _abi.f64_ptr(tape),
            _abi.u64_ptr(tape_off), int(n_steps), int(bool(launch)),
            _abi.f64_ptr(tr.reshape(-1)) if trace else None), self.lib)
        return tr

    def intersect(self, loc, direction, device=0, precision=_abi.F32):
        loc = np.ascontiguousarray(loc, dtype=np.float64).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        n = loc.shape[0]
        out = np.zeros((n, self.n_elements, 3), dtype=np.float64)
        _abi.check(self.lib.rtmc_op_intersect(
            self.handle, int(device), int(precision), n, _abi.f64_ptr(loc.reshape(-1)),
            _abi.f64_ptr(direction.reshape(-1)), _abi.f64_ptr(out.reshape(-1))), self.lib)
        return out

    def crossing(self, batch, device=0, precision=_abi.F32):
        """RayTrace.CheckElementsForCrossingPoint for every photon of `batch`:
        (element index or -1, SmallestDistance, PointCy)."""
        n = batch["wavelength"].shape[0]
        el = np.full(n, -1, dtype=np.int32)
        dist = np.zeros(n, dtype=np.float64)
        pts = np.zeros((n, 3), dtype=np.float64)
        s = batch_struct(batch)
        _abi.check(self.lib.rtmc_op_crossing(self.handle, int(device), int(precision), n, C.byref(s),
                                             _abi.i32_ptr(el), _abi.f64_ptr(dist),
                                             _abi.f64_ptr(pts.reshape(-1))), self.lib)
        return el, dist, pts

    def photon_op(self, op, batch, args=None, tape=None, tape_off=None, uniform_tape=True, device=0,
                  precision=_abi.F64):
        """One of the reference's single-photon operations (`_abi.PHOP`) on every
        photon of `batch`, in place, through rtmc_op_photon.  `args`: (n, 12) array
        (point, plane normal, three scalars, old direction).  Returns the (n, 2)
        result array.  Without a tape none of the operation's draws is available."""
        n = batch["wavelength"].shape[0]
        a = np.zeros((n, _abi.PHOP_ARGS)) if args is None else np.ascontiguousarray(args, dtype=np.float64)
        assert a.shape == (n, _abi.PHOP_ARGS)
        if tape is None:
            tape, tape_off = np.zeros(1), np.zeros(n + 1, dtype=np.uint64)
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        tape_off = np.ascontiguousarray(tape_off, dtype=np.uint64)
        assert tape_off.shape[0] == n + 1
        out = np.zeros((n, 2))
        s = batch_struct(batch)
        _abi.check(self.lib.rtmc_op_photon(
            self.handle, int(device), int(precision), int(_abi.PHOP[op]), n, C.byref(s),
            _abi.f64_ptr(a.reshape(-1)), _abi.f64_ptr(out.reshape(-1)), _abi.f64_ptr(tape),
            _abi.u64_ptr(tape_off), int(bool(uniform_tape))), self.lib)
        return out

    def info(self):
        """What the library found in the scene: runs of coplanar rectangles, their members, the float32
        guard mode (0 none, 1 always, 2 until the first absorption), whether the all-features kernel runs."""
        out = (C.c_int32 * 4)()
        _abi.check(self.lib.rtmc_scene_info(self.handle, out), self.lib)
        return {"runs": out[0], "run_members": out[1], "guard": out[2], "all_features_kernel": bool(out[3])}

    def cells(self):
        """The closed boxes among the scene's rectangles, and the empty regions of a box around several others
        that serve as cells too: how many cells, how many of them regions, in how many photons can be confined,
        the one the source sits in (0 none), whether the scene runs the kernels that use them, and the feature
        bits of its float32 trace kernel (the template argument of `trace_kernel<float, …>`)."""
        out = (C.c_int32 * 4)()
        _abi.check(self.lib.rtmc_scene_cells(self.handle, out), self.lib)
        return {"cells": out[0] & 0xFFFF, "regions": out[0] >> 16, "usable": out[1], "source_cell": out[2],
                "in_use": bool(out[3] & 512), "kernel_features": int(out[3])}

    def wavelengths(self):
        cnt = C.c_int64()
        _abi.check(self.lib.rtmc_scene_wavelengths(self.handle, C.byref(cnt), None, 0), self.lib)
        vals = np.zeros(cnt.value, dtype=np.float64)
        _abi.check(self.lib.rtmc_scene_wavelengths(self.handle, C.byref(cnt),
                                                   _abi.f64_ptr(vals), vals.size), self.lib)
        return vals


def intersect_single(geometry, location, direction, precision=_abi.F64):
    """One ray against one shape, through the device routine."""
    from .Elements import Mirror
    from .Materials import MaterialObj
    from .OpticalObject import OpticalObject
    from .PhotonObject import Photon
    air = MaterialObj("air", 1.0)
    ph = Photon(EmissionMode="Directional Strict", EmissionDirection=[0, 0, 1.0],
                EmissionWavelengthType="fixed", StartingMaterial=air)
    sc = Scene([OpticalObject(Mirror(ph), geometry)], ph)
    try:
        return sc.intersect(location, direction, precision=precision)[0, 0]
    finally:
        sc.close()
