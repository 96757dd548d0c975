"""The reference's single-photon methods on the device (`rtmc_op_photon`).

The reference moves one mutable Photon by hand in its TestPhoton.py and in
the scripts behind its known-answer vectors: PrepareExecution, ReflectOnPlane,
RefractOnPlane, FocusThinLens, PostProcessExecution, CheckForAbsorption ...
The host mirror keeps those names; each runs the routine the trace kernel
itself is built from.  Checked here:

  * the reference's own outputs for 512 interface and 200 thin-lens operations
    (tests/golden/kat.npz, kat_shapes.npz), through the host methods and as
    batches in both precisions;
  * the flow of the reference's TestPhoton.py:35-68, written as there;
  * search + CheckForAbsorption + ExecuteOperation, called one after the other,
    reproduce one loop iteration of `rtmc_step_tape` and of the CPU checker on
    every (photon, iteration) of the golden traces, detector, lens, cylinder,
    donor/acceptor transfer and aligned dyes included.
"""
import numpy as np
import pytest

import helpers as H
from raytracer_amd import Elements, Materials, PhotonObject, RayTrace, _abi, configs
from raytracer_amd import scene as sc

pytestmark = pytest.mark.gpu


def _standalone(material):
    return PhotonObject.Photon(EmissionMode="Directional Strict", EmissionDirection=[0, 0, 1.0],
                               EmissionWavelengthType="fixed", StartingMaterial=material)


def test_interface_known_answers_through_the_host_methods(golden_dir):
    """tests/golden/make_golden.py ran exactly these calls on the reference's objects."""
    k = np.load(golden_dir + "/kat.npz")
    air, glass = Materials.MaterialObj("kat_air", 1.0), Materials.MaterialObj("kat_glass", 1.5)
    photon = _standalone(air)
    iface = Elements.Interface(air, glass, photon)
    worst = 0.0
    for i in range(0, len(k["if_res"]), 4):
        res, refl = k["if_res"][i], k["if_refl"][i]
        inside = bool(k["if_inside"][i])
        n1, n2 = (1.5, 1.0) if inside else (1.0, 1.5)

        def fresh():
            photon.Location, photon.Direction = k["if_L"][i].copy(), k["if_D"][i].copy()
            photon.ElectricFieldVector = k["if_E"][i].copy()
            photon.OperationPlane = [k["if_S"][i].copy(), k["if_n"][i].copy()]
            photon.CurrentMaterial = glass if inside else air
            photon.PrepareExecution()
        fresh()
        assert photon.sgn == res[2]
        assert float(iface.IsTotalReflect(n1, n2)) == res[0]
        if not res[0]:
            R = iface.Reflectance(n1, n2)
            assert (np.isnan(R) and np.isnan(res[1])) or R == pytest.approx(res[1], rel=1e-11, abs=1e-14)
        fresh()
        photon.RefractOnPlane(n1, n2)
        photon.PostProcessExecution()
        got = np.concatenate([photon.Location, photon.Direction, photon.ElectricFieldVector])
        good = ~np.isnan(res[3:12])
        assert np.array_equal(np.isnan(got), ~good)
        worst = max(worst, np.abs(got[good] - res[3:12][good]).max(initial=0.0))
        fresh()
        photon.ReflectOnPlane()
        photon.PostProcessExecution()
        got = np.concatenate([photon.Location, photon.Direction, photon.ElectricFieldVector])
        worst = max(worst, np.abs(got - refl).max())
        assert photon.WasElement
    assert worst < 1e-11, worst


def _plane_batch(L, D, E, s):
    n = len(L)
    b = sc.new_photon_batch(n)
    b["loc"][:], b["dir"][:], b["efield"][:] = L.T, D.T, E.T
    b["wavelength"][:] = b["start_wavelength"][:] = s.wavelengths()[0]
    return b


def _kat_conditions(k, ks):
    """Condition numbers of the reference's 512 interface and 200 lens vectors, from their inputs
    alone (the rule is stated in helpers.py)."""
    L, D, S, N = k["if_L"], k["if_D"], k["if_S"], k["if_n"]
    cos_i = np.abs(np.einsum("ij,ij->i", D, N)) / (H._norm(D) * H._norm(N))
    n1, n2 = np.where(k["if_inside"], 1.5, 1.0), np.where(k["if_inside"], 1.0, 1.5)
    sin2_t = (n1 / n2) ** 2 * (1.0 - cos_i ** 2)
    with np.errstate(divide="ignore"):
        k_refraction = 1.0 / np.sqrt(np.clip(1.0 - sin2_t, 0.0, None))   # infinite where no refracted ray exists
    path = np.abs(np.einsum("ij,ij->i", S - L, N)) / (cos_i * H._norm(N))
    c = {"cos_i": cos_i, "sin2_t": sin2_t, "refraction": k_refraction, "landing": H.kappa_landing(L, S, path, cos_i)}
    # thin lens (PhotonObject.py:158-191): v = -(t2 D) - gz z^, t2 = (fo - nz gz) / (D.n)
    L, D, S, N, f = ks["lens_L"], ks["lens_D"], ks["lens_S"], ks["lens_n"], ks["lens_f"]
    dn = np.einsum("ij,ij->i", D, N)
    cos_l = np.abs(dn) / (H._norm(D) * H._norm(N))
    off = np.einsum("ij,ij->i", L - S, N)
    B = L - (off / dn)[:, None] * D
    gz = S[:, 2] - B[:, 2]
    fo = np.where(np.sign(off) == np.sign(f), -f, f)
    t2 = (fo - N[:, 2] * gz) / dn
    v = -(t2[:, None] * D)
    v[:, 2] -= gz
    c["lens_landing"] = H.kappa_landing(L, S, np.abs(off) / cos_l, cos_l)
    c["lens_direction"] = (np.abs(t2) * H._norm(D) / cos_l + c["lens_landing"]) / H._norm(v)
    return c


@pytest.mark.parametrize("precision", [_abi.F64, _abi.F32])
def test_interface_and_lens_known_answers_as_batches(golden_dir, precision):
    """All 512 interface and 200 lens vectors of the reference, none left out.  float64: 1e-11.
    float32: every vector within 1e-6 times its own condition number (helpers.py states the
    rule); the printed lines say how many also meet the plain 1e-6."""
    k = np.load(golden_dir + "/kat.npz")
    ks = np.load(golden_dir + "/kat_shapes.npz")
    cond = _kat_conditions(k, ks)
    f32 = precision == _abi.F32

    def check(label, err, kappa):
        if f32:
            H.report_f32(label, err, kappa)
        else:
            assert np.max(err) < 1e-11, (label, np.max(err))

    photon = _standalone(Materials.MaterialObj("kat_air", 1.0))
    s = photon._scene()
    n = len(k["if_res"])
    args = np.zeros((n, _abi.PHOP_ARGS))
    args[:, 0:3], args[:, 3:6] = k["if_S"], k["if_n"]
    args[:, 6] = np.where(k["if_inside"], 1.5, 1.0)
    args[:, 7] = np.where(k["if_inside"], 1.0, 1.5)
    args[:, 8] = k["if_res"][:, 2]
    args[:, 9:12] = k["if_D"]
    # IsTotalReflect / Reflectance
    b = _plane_batch(k["if_L"], k["if_D"], k["if_E"], s)
    out = s.photon_op("reflectance", b, args, precision=precision)
    # (float32: the two answers meet where sin^2(theta_t) is 1 within its own rounding)
    decided = np.abs(cond["sin2_t"] - 1.0) > (4 * H.F32_REL if f32 else 0.0)
    assert np.array_equal(out[decided, 0], k["if_res"][decided, 0])
    part = (k["if_res"][:, 0] == 0) & ~np.isnan(k["if_res"][:, 1]) & (out[:, 0] == 0)
    check("reflectance", np.abs(out[part, 1] - k["if_res"][part, 1]), cond["refraction"][part])
    if not f32:       # the reference's NaN reflectances are NaN here too
        assert np.array_equal(np.isnan(out[:, 1]), np.isnan(k["if_res"][:, 1]) & (k["if_res"][:, 0] == 0))
    # RefractOnPlane + PostProcessExecution
    b = _plane_batch(k["if_L"], k["if_D"], k["if_E"], s)
    s.photon_op("refract", b, args, precision=precision)
    s.photon_op("polarization", b, args, precision=precision)
    got = np.concatenate([b["loc"].T, b["dir"].T, b["efield"].T], axis=1)
    ref = k["if_res"][:, 3:12]
    good = ~np.isnan(ref).any(axis=1)       # no refracted ray beyond the critical angle: NaN in the reference
    assert np.isnan(got[~good, 3:6]).all()
    check("refraction: landing point", np.abs(got - ref)[good, 0:3].max(axis=1), cond["landing"][good])
    check("refraction: direction", np.abs(got - ref)[good, 3:6].max(axis=1), cond["refraction"][good])
    check("refraction: polarisation", np.abs(got - ref)[good, 6:9].max(axis=1),
          (cond["refraction"] * H.kappa_rotation(k["if_D"], np.where(good[:, None], ref[:, 3:6], k["if_D"])))[good])
    # ReflectOnPlane + PostProcessExecution
    b = _plane_batch(k["if_L"], k["if_D"], k["if_E"], s)
    s.photon_op("reflect", b, args, precision=precision)
    s.photon_op("polarization", b, args, precision=precision)
    got = np.concatenate([b["loc"].T, b["dir"].T, b["efield"].T], axis=1)
    ref = k["if_refl"]
    check("reflection: landing point", np.abs(got - ref)[:, 0:3].max(axis=1), cond["landing"])
    check("reflection: direction", np.abs(got - ref)[:, 3:6].max(axis=1), np.ones(n))
    check("reflection: polarisation", np.abs(got - ref)[:, 6:9].max(axis=1), H.kappa_rotation(k["if_D"], ref[:, 3:6]))
    # FocusThinLens + PostProcessExecution
    m = len(ks["lens_f"])
    args = np.zeros((m, _abi.PHOP_ARGS))
    args[:, 0:3], args[:, 3:6], args[:, 6], args[:, 9:12] = ks["lens_S"], ks["lens_n"], ks["lens_f"], ks["lens_D"]
    b = _plane_batch(ks["lens_L"], ks["lens_D"], ks["lens_E"], s)
    s.photon_op("lens", b, args, precision=precision)
    s.photon_op("polarization", b, args, precision=precision)
    got = np.concatenate([b["loc"].T, b["dir"].T, b["efield"].T], axis=1)
    ref = ks["lens_res"]
    if not f32:
        assert np.abs(got - ref).max() < 1e-10   # (paths of up to 1e3 cm in this set)
        return
    check("lens: landing point", np.abs(got - ref)[:, 0:3].max(axis=1), cond["lens_landing"])
    check("lens: direction", np.abs(got - ref)[:, 3:6].max(axis=1), cond["lens_direction"])
    check("lens: polarisation", np.abs(got - ref)[:, 6:9].max(axis=1),
          cond["lens_direction"] * H.kappa_rotation(ks["lens_D"], ref[:, 3:6]))


@pytest.mark.parametrize("precision", [_abi.F64, _abi.F32])
def test_reflectance_at_exactly_normal_incidence_on_a_tilted_plane(oracle_mod, precision):
    """A direction that is bit for bit the negative of a tilted plane's normal (a strict source
    aimed along a rotated slab's normal): numpy sums D.n to 1 + 2^-52 here, the arccos of the
    reference's Reflectance (Elements.py:87) is NaN -- but its cross product D x n is exactly zero
    (numpy rounds every product on its own), so it takes the normal-incidence branch
    ((n1 - n2)/(n1 + n2))^2 (Elements.py:98,107).  The checker and the float64 routine do the same; a
    fused multiply-subtract in the cross product would leave a rounding residue there and a NaN
    reflectance (always reflected).  No vector of the reference's own pins this case; it follows from
    its source as cited."""
    n = np.array([0.6371277743479907, 0.7070225310439712, 0.3068995597105879])
    D = -n
    E = np.cross(D, [0.0, 0.0, 1.0])
    E /= np.linalg.norm(E)
    assert abs(float(np.dot(D, n))) > 1.0                    # the case in question
    r, _, _ = oracle_mod.op_interface(np.zeros(3), n, 1.0, 1.5, np.array([1.0, 1.0, 1.0]), D, E)
    assert r[0] == 0 and r[1] == pytest.approx(0.04, rel=1e-12)
    photon = _standalone(Materials.MaterialObj("kat_air", 1.0))
    s = photon._scene()
    b = _plane_batch(np.array([[1.0, 1.0, 1.0]]), D[None, :], E[None, :], s)
    args = np.zeros((1, _abi.PHOP_ARGS))
    args[0, 3:6], args[0, 6:9] = n, (1.0, 1.5, 1.0)
    out = s.photon_op("reflectance", b, args, precision=precision)
    assert out[0, 0] == 0 and out[0, 1] == pytest.approx(0.04, rel=1e-12 if precision == _abi.F64 else 1e-5)


def test_the_flow_of_the_references_testphoton():
    """TestPhoton.py:20-68 line by line, on the job its constants belong to: a
    mirror in the plane z = 0 and a photon launched from (0, 0, 0.1) straight down."""
    job = configs.mirror_job()
    RT = RayTrace.RayTrace(demo_scene=False)
    RT.LoadJob(job)
    RT.Photon.GeneratePhoton()
    assert RT.Photon.Location[0] == 0.0 and RT.Photon.Location[1] == 0.0 and RT.Photon.Location[2] == 0.1
    # test_FindIntersections
    assert set(RT.OpticalElements[0].Geometry.FindIntersections()) == {0.0}
    # test_ReflectOnPlane
    RT.Photon.BeenAbsorbed = [False, 0]
    RT.OldPoint = RT.Photon.Location
    RT.Photon.ResetPhoton(RT.Elements[0])
    RT.CheckElementsForCrossingPoint()
    RT.Photon.CheckForAbsorption(RT.OldPoint, RT.SmallestDistance)
    RT.OldPoint = RT.Photon.OldPoint
    RT.Photon.OperationPlane = RT.OldElement.Geometry.OperationPlane()
    RT.Photon.ReflectOnPlane()
    assert set(RT.Photon.Location) == {0.0}
    assert set(RT.Photon.Direction) == {0.0, -1.0}

    RT.Photon.ResetPhoton(RT.Elements[0])
    RT.Photon.Direction = np.array([1.0, 1.0, -1.0])
    RT.CheckElementsForCrossingPoint()
    RT.Photon.CheckForAbsorption(RT.OldPoint, RT.SmallestDistance)
    RT.OldPoint = RT.Photon.OldPoint
    RT.Photon.OperationPlane = RT.OldElement.Geometry.OperationPlane()
    RT.Photon.ReflectOnPlane()
    # (the reference asserts these to 0 decimal places against 0.0909...; the point is (0.1, 0.1, 0))
    assert list(RT.Photon.Location) == pytest.approx([0.1, 0.1, 0.0], abs=1e-15)
    assert round(float(RT.Photon.Location[0]), 0) == round(0.09090909, 0)
    # line-based reflection 2 (n.D) n - D (PhotonObject.py:150): the negative of the mirror direction
    assert list(RT.Photon.Direction) == [-1.0, -1.0, -1.0]
    # Mirror.ExecuteOperation does the same and counts the bounce
    RT.Photon.ResetPhoton(RT.Elements[0])
    RT.Photon.OperationPlane = RT.OpticalElements[0].Geometry.OperationPlane()
    RT.OpticalElements[0].Type.ExecuteOperation()
    assert RT.OpticalElements[0].Type.BounceCounter == 1
    assert list(RT.Photon.Location) == [0.0, 0.0, 0.0] and list(RT.Photon.Direction) == [0.0, 0.0, -1.0]
    assert RT.OpticalElements[0].Type.IsValidNextPoint(RT.Photon.OldPoint, [0.3, 0.0, 2.0])
    assert not RT.OpticalElements[0].Type.IsValidNextPoint(RT.Photon.OldPoint, [0.3, 0.0, -2.0])


@pytest.fixture(scope="module")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            rt = H.raytrace_for(name)
            cache[name] = (rt, rt.Scene())
        return cache[name]
    yield get
    for rt, s in cache.values():
        s.close()


def _stack(states, g, picks):
    big = sc.new_photon_batch(len(picks))
    off = np.zeros(len(picks) + 1, dtype=np.uint64)
    tapes = []
    for j, (i, k) in enumerate(picks):
        for key, arr in states[i].items():
            if arr.ndim == 2:
                big[key][:, j] = arr[:, k]
            else:
                big[key][j] = arr[k]
        seg = g["tape"][g["tape_off"][k]:g["tape_off"][k + 1]]
        tapes.append(seg)
        off[j + 1] = off[j] + len(seg)
    return big, np.concatenate(tapes), off


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C6", "C7", "C8", "C9", "C2r"])
@pytest.mark.parametrize("precision", [_abi.F64, _abi.F32])
def test_search_absorption_and_execution_make_one_iteration(scenes, oracle_mod, name, precision):
    """CheckElementsForCrossingPoint, Photon.CheckForAbsorption and the element's
    ExecuteOperation called one after the other (RayTrace.py:178-189) against one
    iteration of rtmc_step_tape in the same precision, and in float64 against the CPU
    checker's next state, with the reference's variates.  float64: 1e-10.  float32: the two
    routes round differently, so they may differ by twice the bound each of them is held to
    against the reference (1e-6 times the condition number of that photon's step, helpers.py)."""
    from test_gpu_parity import _oracle_states
    g = H.load_trace(name)
    rt, s = scenes(name)
    iters = int(min(25, g["n_iter"].max()))
    states = _oracle_states(oracle_mod, rt, g, iters)
    n = len(g["n_iter"])
    picks = [(i, k) for i in range(iters) for k in range(n)
             if states[i]["status"][k] not in (4, 5, 6) and states[i]["iterations"][k] == i]
    whole, tape, off = _stack(states, g, picks)
    parts = {key: v.copy() for key, v in whole.items()}
    s.step_tape(whole, tape, off, 1, launch=False, precision=precision)
    el, dist, pts = s.crossing(parts, precision=precision)
    found = el >= 0
    args = np.zeros((len(picks), _abi.PHOP_ARGS))
    args[:, 0:3] = pts
    parts["hit_distance"][:] = np.where(found, dist, 0.0)
    s.photon_op("absorption", parts, args, tape, off, uniform_tape=False, precision=precision)
    travelling = parts["status"] == 0
    # the element's operation for the photons that were not absorbed: a batch of its own
    idx = np.nonzero(found & travelling)[0]
    sub = {key: (v[:, idx].copy() if v.ndim == 2 else v[idx].copy()) for key, v in parts.items()}
    sub_off = np.zeros(len(idx) + 1, dtype=np.uint64)
    sub_tape = []
    for j, q in enumerate(idx):
        seg = tape[off[q]:off[q + 1]]
        sub_tape.append(seg)
        sub_off[j + 1] = sub_off[j] + len(seg)
    sub_args = args[idx].copy()
    sub_args[:, 6] = el[idx]
    out = s.photon_op("execute", sub, sub_args, np.concatenate(sub_tape), sub_off, uniform_tape=False,
                      precision=precision)
    for key, v in sub.items():
        if v.ndim == 2:
            parts[key][:, idx] = v
        else:
            parts[key][idx] = v
    assert found.sum() > 0.9 * len(picks) and len(idx) > 0 and (~travelling).sum() >= 0
    cmp = np.nonzero(found)[0]
    for key in ("status", "material", "reflected", "was_element", "tape_pos"):
        assert np.array_equal(parts[key][cmp], whole[key][cmp]), key
    assert np.array_equal(parts["wavelength"][cmp], whole["wavelength"][cmp])
    if precision == _abi.F64:
        bounds = np.full((len(picks), 3), 1e-10)
    else:
        bounds = 2.0 * np.array([H.f32_iteration_bounds(rt, states[i], states[i + 1], k) for i, k in picks])
    for key, col in (("loc", 0), ("dir", 1), ("efield", 1)):
        assert (np.abs(parts[key][:, cmp] - whole[key][:, cmp]).max(axis=0) <= bounds[cmp, col]).all(), key
    absorbing = np.array([s.packed.materials[m].IsAbsorber for m in whole["material"]])
    sel = cmp[absorbing[cmp] & (whole["dist_left"][cmp] < 1e30)]
    if len(sel):
        assert (np.abs(parts["dist_left"][sel] - whole["dist_left"][sel])
                <= np.maximum(bounds[sel, 2], 1e-10 * np.abs(whole["dist_left"][sel]))).all()
    # bounces the elements counted = photons that stayed in their medium at a mirror or an interface
    kinds = np.array([rt.OpticalElements[e].Type.kind for e in el[idx]])
    reflected = (kinds == "Mirror") | ((kinds == "Interface") & (sub["reflected"] == 1))
    assert np.array_equal(out[:, 0] == 1, reflected)
    if precision == _abi.F64:
        for j in cmp:
            i, k = picks[j]
            ref = states[i + 1]
            assert parts["status"][j] == ref["status"][k] and parts["material"][j] == ref["material"][k]
            assert parts["wavelength"][j] == ref["wavelength"][k]
            for key in ("loc", "dir", "efield"):
                assert np.abs(parts[key][:, j] - ref[key][:, k]).max() < 1e-9, (key, i, k)


def test_hand_stepped_photon_reproduces_the_tallies_of_a_run():
    """A photon moved by hand through the reference's loop body (RayTrace.py:166-199)
    with the host methods, 200 photons of the Lumogen Red slab: every photon ends, and
    the fates have the proportions the trace kernel finds (within 5 sigma)."""
    rt = H.raytrace_for("C2")
    np.random.seed(7)
    ph = rt.Photon
    n, dumped, converted = 200, 0, 0
    for _ in range(n):
        ph.ResetPhoton(rt.OpticalElements[0].Geometry.OperationPlane())
        ph.BeenAbsorbed = [False, 0]
        ph.Status = "Start"
        rt.OldPoint = ph.Location
        rt.OldElement = None
        if hasattr(rt, "FirstEl"):
            del rt.FirstEl
        for i in range(2000):
            rt.CheckElementsForCrossingPoint()
            ph.CheckForAbsorption(rt.OldPoint, rt.SmallestDistance)
            rt.OldPoint = ph.OldPoint
            if ph.Status == "Start":
                ph.OperationPlane = rt.OldElement.Geometry.OperationPlane()
                rt.OldElement.Type.ExecuteOperation()
                rt.OldPoint = ph.OldPoint
            if ph.Status in ("InternalConverted", "Dumped"):
                break
        assert ph.Status in ("InternalConverted", "Dumped"), ph.Status
        dumped += ph.Status == "Dumped"
        converted += ph.Status == "InternalConverted"
    t = rt.Scene().run(2000000, seed=3)
    p_ic = t["n_internal_converted"] / 2e6
    assert abs(converted / n - p_ic) < 5 * np.sqrt(p_ic * (1 - p_ic) / n), (converted, p_ic)
    counted = sum(oe.Type.Counter for oe in rt.OpticalElements if oe.Type.kind == "Dump")
    assert counted == dumped == n - converted


def test_hand_driven_photon_with_an_unknown_wavelength():
    """A photon that was never generated keeps the constructor's 450 nm (PhotonObject.py:19), which is
    no wavelength of the scene: the reference resolves it with FindClosestListElement
    (PhotonObject.py:235,288), and so do LamBeerDistance and CheckForAbsorption here."""
    rt = H.raytrace_for("C2")
    ph = rt.Photon
    ph.Wavelength = ph.StartingWavelength = 450.0
    ph.CurrentMaterial = next(m for m in rt.MaterialDict.values() if m.IsAbsorber)
    ph.Location, ph.Direction = np.array([0.0, 0.0, 0.1]), np.array([0.0, 0.0, 1.0])
    ph.ElectricFieldVector = np.array([1.0, 0.0, 0.0])
    np.random.seed(4)
    ph.LamBeerDistance()
    assert 0.0 < ph.DistanceLeftToAbsorption < 10.0
    ph.PointCy = np.array([0.0, 0.0, 0.3])
    ph.DistanceLeftToAbsorption = 0.05
    ph.CheckForAbsorption(ph.Location, 0.2)
    assert ph.Status in ("ReemittedAcceptor", "InternalConverted") and ph.Location[2] == pytest.approx(0.15)
