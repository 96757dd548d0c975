"""Job definitions for the benchmark / parity configurations.

Every function returns a *job*: a dict with the six arrays of a `.prt` job
folder in exactly the layout the reference's GUI writes and
`IOOperation.LoadSaveRT.LoadArrays` reads (reference IOOperation.py:19-31,
field order of a material row at IOOperation.py:81, of the photon list at
IOOperation.py:100-109).  Units: cm, nm, mmol/L.

The same dicts feed three consumers: `prt.dump_job` (to write a job folder),
`RayTrace.RayTrace.LoadJob` (the host mirror) and the golden-vector generator
that runs the reference itself.
"""
from __future__ import annotations

import copy
import os

import numpy as np

AIR_ROW = ["air", 1.0, False, 0.0, 0.0, "blue", False, [0.0, 0.0, 0.0], 0.0, 0.0, 0.0, 0.0,
           0.0, 0.0, 0.0, 0.0, "0", "0", "0", "0", 0, False, 0.0, 1]


EQE_DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "eqe_demo.txt")


def dye_row(name, conc_a=1.0, conc_b=0.0, n=1.5, aligned=False, align_s=0.1,
            align_vect=(0.0, 0.0, 1.0), abs_em_angle=0, fix_dist=False, fix_dist_r=1.0):
    """Material row: Lumogen Red acceptor (+ optional Oxacin donor)."""
    return [name, float(n), True, float(conc_b), float(conc_a), "red", bool(aligned),
            [float(c) for c in align_vect], float(align_s), 5.0,
            0.9, 0.95, 2.5, 6.0, 35731.0, 54000.0,
            "OxacinAbs.txt", "OxacinEm.txt", "LumogenRedAbs.txt", "LumogenRedEm.txt",
            abs_em_angle, bool(fix_dist), float(fix_dist_r), 1]


def _slab_faces(z0, z1, half, mat, edge_types=(4, 4, 4, 4), eqe=""):
    """Six rectangles of one slab: bottom/top interfaces then the four edges.

    Order and orientation follow the shipped job (bottom `[mat, air]`, top
    `[air, mat]`, edges at +x, -x, +y, -y with inward normals)."""
    zc = 0.5 * (z0 + z1)
    hw = 0.5 * (z1 - z0)
    elements = [
        [[0.0, 0.0, z0], [0.0, 0.0, 1.0]],
        [[0.0, 0.0, z1], [0.0, 0.0, 1.0]],
        [[half, 0.0, zc], [-1.0, 0.0, 0.0]],
        [[-half, 0.0, zc], [1.0, 0.0, 0.0]],
        [[0.0, half, zc], [0.0, -1.0, 0.0]],
        [[0.0, -half, zc], [0.0, 1.0, 0.0]],
    ]
    props = [[2, mat, "air"], [2, "air", mat]]
    for t in edge_types:
        props.append([4, 0, 0, eqe] if t == 4 else [1])
    sizes = [[half, half], [half, half]] + [[half, hw]] * 4
    return elements, props, [list(s) for s in sizes]


def _dump_box(cx, cy, zc, hx, hz):
    """Closed box of six detector rectangles written out explicitly."""
    elements = [
        [[cx + hx, cy, zc], [1.0, 0.0, 0.0]],
        [[cx - hx, cy, zc], [-1.0, 0.0, 0.0]],
        [[cx, cy + hx, zc], [0.0, 1.0, 0.0]],
        [[cx, cy - hx, zc], [0.0, -1.0, 0.0]],
        [[cx, cy, zc + hz], [0.0, 0.0, 1.0]],
        [[cx, cy, zc - hz], [0.0, 0.0, -1.0]],
    ]
    props = [[4, 0, 0, ""] for _ in range(6)]
    sizes = [[hx, hz], [hx, hz], [hx, hz], [hx, hz], [hx, hx], [hx, hx]]
    return elements, props, sizes


def _photon(n, start, mode, wl_type, wl, mat, direction, respect_pol=True, scale=1.0):
    return [int(n), [float(c) for c in start], mode, wl_type, wl, mat,
            [float(c) for c in direction], bool(respect_pol), float(scale)]


STRICT = "Directional Strict"
ISOTROPIC = "Directional Isotropic"
FIXED_WL = "Specified Wavelength..."
SUN = "Sun spectrum AG15g"


def slab_job(thickness=0.3, enclosed=True, donor=False, n_photons=10000, sun=False,
             aligned=False, align_s=0.1, abs_em_angle=0, respect_pol=True,
             direction=(0.0, 0.0, -1.0), eqe="", scale=1.0, fix_dist=False, fix_dist_r=1.0):
    """One 5 x 5 cm dye-doped slab, source at (0, 2, 1) pointing down at 460 nm.

    thickness 0.012 + enclosed False is the shipped job's geometry."""
    name = "LRO" if donor else "LR"
    els, props, sizes = _slab_faces(0.0, thickness, 2.5, name, eqe=eqe)
    if enclosed:
        zc = 0.5 * thickness
        e2, p2, s2 = _dump_box(0.0, 0.0, zc, 4.0, 2.0)
        els, props, sizes = els + e2, props + p2, sizes + s2
    mats = [copy.deepcopy(AIR_ROW),
            dye_row(name, conc_a=1.0, conc_b=1.0 if donor else 0.0, aligned=aligned,
                    align_s=align_s, abs_em_angle=abs_em_angle, fix_dist=fix_dist,
                    fix_dist_r=fix_dist_r)]
    photon = _photon(n_photons, (0.0, 2.0, 1.0), STRICT, SUN if sun else FIXED_WL, 460,
                     "air", direction, respect_pol=respect_pol, scale=scale)
    job = {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": [1] * len(els),
        "ElementSize": sizes,
        "Materials": mats,
        "Photon": photon,
    }
    if sun:     # the reference reads AM15.txt, which it does not ship: these configurations say which table they mean
        job["SunTable"] = synthetic_am15()
    return job


def stacked_job(n_photons=10000, n_slabs=3):
    """Stacked slabs (0.3 thick, 0.05 air gaps; three by default); each slab has
    mirrors on three edges and a detector on the +x edge; isotropic source
    above; closed detector box around everything.  Element order: slab 1
    (top), slab 2, ..., box."""
    els, props, sizes = [], [], []
    tops = [0.35 * (n_slabs - 1 - k) for k in range(n_slabs)]
    names = [("LR", "LRO")[k % 2] for k in range(n_slabs)]
    for z0, nm in zip(tops, names):
        e, p, s = _slab_faces(z0, z0 + 0.3, 2.5, nm, edge_types=(4, 1, 1, 1))
        els, props, sizes = els + e, props + p, sizes + s
    top = tops[0] + 0.3
    e, p, s = _dump_box(0.0, 0.0, 0.5, 4.0, 2.0 if n_slabs == 3 else top + 1.5)
    els, props, sizes = els + e, props + p, sizes + s
    mats = [copy.deepcopy(AIR_ROW), dye_row("LR"), dye_row("LRO", conc_b=1.0)]
    photon = _photon(n_photons, (0.0, 0.0, 2.0 if n_slabs == 3 else top + 1.0), ISOTROPIC, FIXED_WL, 460,
                     "air", (0.0, 0.0, -1.0))
    return {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": [1] * len(els),
        "ElementSize": sizes,
        "Materials": mats,
        "Photon": photon,
    }


def rod_job(n_photons=10000):
    """All element shapes and types in one scene: a dye-filled rod (cylinder
    wall + two circular caps, interfaces air/LR) lying along y, a thin lens
    (circle) between the source and the rod, a triangular mirror below, a small
    detector tube beside the rod, and a closed detector box around everything.
    Hemispherical 460 nm source beside the lens.  The lens faces x: the
    reference's thin lens focuses along z only (PhotonObject.py:178) and does
    nothing but reverse the direction vector when its normal is the z axis."""
    from .RTUtils import MakeTriangle
    els, props, geom, sizes = [], [], [], []

    def add(element, prop, g, size):
        els.append(element); props.append(prop); geom.append(g); sizes.append(size)

    p1, p2, r = [0.0, -1.5, 0.0], [0.0, 1.5, 0.0], 0.5
    add([p1, p2], [2, "air", "LR"], 3, [r])                           # rod wall: axis end points
    add([p1, [0.0, 1.0, 0.0]], [2, "air", "LR"], 2, [r])              # caps
    add([p2, [0.0, 1.0, 0.0]], [2, "LR", "air"], 2, [r])
    add([[1.3, 0.0, 0.0], [1.0, 0.0, 0.0]], [3, 0.8], 2, [0.6])       # lens
    tri = [[-1.5, -1.5, -1.0], [1.5, -1.5, -1.2], [0.0, 1.5, -0.8]]
    s, n = MakeTriangle(*tri)
    add([list(s), list(n)], [1], 4, tri)                              # mirror
    add([[-1.5, -1.0, -0.5], [-2.0, 1.0, 0.5]], [4, 0, 0, ""], 3, [0.3])  # detector tube
    e, p, sz = _dump_box(0.0, 0.0, 0.0, 3.0, 3.0)
    for a, b, c in zip(e, p, sz):
        add(a, b, 1, c)
    mats = [copy.deepcopy(AIR_ROW), dye_row("LR")]
    photon = _photon(n_photons, (2.2, 0.1, 0.2), ISOTROPIC, FIXED_WL, 460, "air", (-1.0, 0.05, -0.1))
    return {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": geom,
        "ElementSize": sizes,
        "Materials": mats,
        "Photon": photon,
    }


def rotated_job(base="C2", axis=(0.3, -0.5, 0.8), angle=0.7, n_photons=10000, **kwargs):
    """A rectangle-only job turned rigidly about `axis` by `angle` radians:
    every plane gets a general orientation (explicit in-plane axes keep the
    rectangles' footprints), the physics stays that of the base job."""
    from .Geometries import Rectangle
    builder, kw = CONFIGS[base]
    job = builder(**dict(kw, n_photons=n_photons, **kwargs))
    k = np.asarray(axis, dtype=np.float64)
    k /= np.linalg.norm(k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    els = np.array(job["Elements"], dtype=np.float64)
    sizes = []
    for i in range(len(els)):
        assert job["ElementGeometry"][i] == 1
        size = job["ElementSize"][i]
        rect = Rectangle(size[0], size[1], els[i][0], els[i][1], None,
                         size[2] if len(size) > 2 else [0, 0, 0])
        els[i][0] = R @ els[i][0]
        els[i][1] = R @ rect.DirectionVector
        sizes.append([size[0], size[1], list(R @ rect.e1)])
    job["Elements"] = els
    job["ElementSize"] = sizes
    ph = job["Photon"]
    ph[1] = [float(c) for c in R @ np.asarray(ph[1])]
    ph[6] = [float(c) for c in R @ np.asarray(ph[6])]
    return job


def strip_job(n_photons=10000, strips=14, sun=False, enclosed=True):
    """C2 with every edge of the slab cut into `strips` detectors side by side
    (a segmented edge read-out): 4 * strips + 6 detectors, 4 * strips + 8
    elements.  strips = 1 is C2 itself."""
    thickness, half = 0.3, 2.5
    els, props, sizes = _slab_faces(0.0, thickness, half, "LR")
    els, props, sizes = els[:2], props[:2], sizes[:2]
    w = 2 * half / strips
    for axis, sign in ((0, 1), (0, -1), (1, 1), (1, -1)):      # +x, -x, +y, -y as in _slab_faces
        for k in range(strips):
            c = -half + (k + 0.5) * w
            support = [sign * half, c, 0.5 * thickness] if axis == 0 else [c, sign * half, 0.5 * thickness]
            normal = [-float(sign), 0.0, 0.0] if axis == 0 else [0.0, -float(sign), 0.0]
            els.append([support, normal])
            props.append([4, 0, 0, ""])
            sizes.append([0.5 * w, 0.5 * thickness])
    if enclosed:    # (without the box photons that leave the slab find nothing and re-execute their last element)
        e2, p2, s2 = _dump_box(0.0, 0.0, 0.5 * thickness, 4.0, 2.0)
        els, props, sizes = els + e2, props + p2, sizes + s2
    photon = _photon(n_photons, (0.0, 2.0, 1.0), STRICT, SUN if sun else FIXED_WL, 460, "air", (0.0, 0.0, -1.0))
    job = {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": [1] * len(els),
        "ElementSize": [list(x) for x in sizes],
        "Materials": [copy.deepcopy(AIR_ROW), dye_row("LR")],
        "Photon": photon,
    }
    if sun:
        job["SunTable"] = synthetic_am15()
    return job


def tiled_floor_job(n_photons=10000, nx=7, ny=5):
    """A floor of nx x ny detector tiles of unequal sizes in the plane z = 0 (tiles share their edges;
    every third one is also covered by a larger detector listed after the tiles, which therefore
    never counts where a tile lies), a mirror ceiling, four detector walls, a hemispherical source
    above the floor: many rectangles in one plane, in two dimensions, with overlaps."""
    els, props, sizes = [], [], []
    xs = np.cumsum([0.0] + [0.5 + 0.25 * (i % 3) for i in range(nx)])
    ys = np.cumsum([0.0] + [0.4 + 0.3 * (j % 2) for j in range(ny)])
    xs, ys = xs - 0.5 * xs[-1], ys - 0.5 * ys[-1]
    for j in range(ny):
        for i in range(nx):
            els.append([[0.5 * (xs[i] + xs[i + 1]), 0.5 * (ys[j] + ys[j + 1]), 0.0], [0.0, 0.0, 1.0]])
            props.append([4, 0, 0, ""])
            sizes.append([0.5 * (xs[i + 1] - xs[i]), 0.5 * (ys[j + 1] - ys[j])])
    for k in range(0, nx * ny, 3):      # covers most of tile k and a little of its neighbours
        c = els[k][0]
        els.append([[c[0] + 0.1, c[1] - 0.1, 0.0], [0.0, 0.0, 1.0]])
        props.append([4, 0, 0, ""])
        sizes.append([0.3, 0.3])
    hx, hy = 0.5 * (xs[-1] - xs[0]), 0.5 * (ys[-1] - ys[0])
    els.append([[0.0, 0.0, 2.0], [0.0, 0.0, 1.0]]); props.append([1]); sizes.append([hx + 1.0, hy + 1.0])
    for sign in (1.0, -1.0):
        els.append([[sign * (hx + 1.0), 0.0, 0.75], [sign, 0.0, 0.0]]); props.append([4, 0, 0, ""]); sizes.append([hy + 1.0, 1.25])
        els.append([[0.0, sign * (hy + 1.0), 0.75], [0.0, sign, 0.0]]); props.append([4, 0, 0, ""]); sizes.append([hx + 1.0, 1.25])
    els.append([[0.0, 0.0, -0.5], [0.0, 0.0, 1.0]]); props.append([4, 0, 0, ""]); sizes.append([hx + 1.0, hy + 1.0])
    return {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": [1] * len(els),
        "ElementSize": sizes,
        "Materials": [copy.deepcopy(AIR_ROW)],
        "Photon": _photon(n_photons, (0.1, -0.2, 1.0), ISOTROPIC, FIXED_WL, 460, "air", (0.0, 0.0, -1.0)),
    }


def flush_detector_job(n_photons=10000):
    """C2 plus a detector lying flush on the slab's top face (same plane, same
    footprint), listed after everything else.  In the reference the interface
    (lower index) wins the tie when a photon arrives, and the detector, now at
    distance zero, is then not a candidate (RayTrace.py:134): it never counts."""
    job = slab_job(thickness=0.3, enclosed=True, n_photons=n_photons)
    job["Elements"] = np.concatenate([job["Elements"], np.array([[[0.0, 0.0, 0.3], [0.0, 0.0, 1.0]]])])
    job["ElementProperties"] = job["ElementProperties"] + [[4, 0, 0, ""]]
    job["ElementGeometry"] = job["ElementGeometry"] + [1]
    job["ElementSize"] = job["ElementSize"] + [[2.5, 2.5]]
    return job


def seam_job(n_photons=10, corner=False, start=(0.0, 2.0, 0.75), direction=(0.0, 0.5, -0.5)):
    """Strict sources aimed exactly at a seam, with coordinates that are exact in
    float32 and float64, so that the candidates tie bit for bit.
    corner=False: a 0.25 cm slab, source at (0, 2, 0.75) along (0, 0.5, -0.5): the
    top face z = 0.25 (element 1) and the edge detector y = 2.5 (element 4) are
    both met at line parameter 1.
    corner=True: a detector box listed z+, z-, x+, x-, y+, y- (the reverse of the
    order in which the search visits its groups), source at the origin along
    (1, 1, 0.5): the faces z = 2, x = 4 and y = 4 (elements 0, 2, 4) meet at
    parameter 4."""
    if not corner:
        job = slab_job(thickness=0.25, enclosed=True, n_photons=n_photons, direction=tuple(direction))
        job["Photon"][1] = [float(c) for c in start]
        return job
    e, p, sz = _dump_box(0.0, 0.0, 0.0, 4.0, 2.0)
    order = [4, 5, 0, 1, 2, 3]
    return {
        "Elements": np.array([e[i] for i in order], dtype=np.float64),
        "ElementProperties": [p[i] for i in order],
        "ElementGeometry": [1] * 6,
        "ElementSize": [sz[i] for i in order],
        "Materials": [copy.deepcopy(AIR_ROW)],
        "Photon": _photon(n_photons, (0.0, 0.0, 0.0), STRICT, FIXED_WL, 460, "air", (1.0, 1.0, 0.5)),
    }


def mirror_job(n_photons=7):
    """One mirror in the plane z = 0 in air and a photon launched from (0, 0, 0.1)
    straight down: the job the constants of the reference's TestPhoton.py belong to
    (TestPhoton.py:16,23-25,50-51)."""
    return {
        "Elements": np.array([[[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]]]),
        "ElementProperties": [[1]],
        "ElementGeometry": [1],
        "ElementSize": [[5.0, 5.0]],
        "Materials": [copy.deepcopy(AIR_ROW)],
        "Photon": [n_photons, [0.0, 0.0, 0.1], STRICT, FIXED_WL, 460, "air", [0.0, 0.0, -1.0], True, 1.0],
    }


ESCAPE_CONE_COS = float(np.sqrt(1.0 - (1.0 / 1.5) ** 2))    # cos of the critical angle 41.81 deg of n = 1.5


def escape_cone_job(n_photons=10000, emission=False):
    """The geometry of the reference's manual check TESTEMISSION (MonteCarloFRET.py:283-308): which
    fraction of isotropic directions lies outside the escape cone of a medium with n = 1.5 (41.81
    degrees about the z axis; expected cos(41.81 deg) = 0.7454)?  A point source at (0, 0, 1); two
    round detectors (elements 0 and 1) in the planes z = 0 and z = 2 cover exactly that cone, up
    and down; a closed detector box catches the rest.
    emission=False: a hemispherical source in air (the launch sampler, PhotonObject.py:68-86).
    emission=True: the source sits *inside* a dye medium with n = 1.5 and a vanishing
    concentration: every photon is absorbed on the spot (the free path of a new photon is zero,
    PhotonObject.py:32) and re-emitted isotropically from the source point (MonteCarloFRET.py:92-105,
    170-175) -- or lost to internal conversion --, then flies to a detector unhindered."""
    r = float(np.sqrt(1.0 - ESCAPE_CONE_COS ** 2) / ESCAPE_CONE_COS)   # tan of the critical angle
    els = [[[0.0, 0.0, 0.0], [0.0, 0.0, 1.0]], [[0.0, 0.0, 2.0], [0.0, 0.0, 1.0]]]
    props = [[4, 0, 0, ""], [4, 0, 0, ""]]
    geom, sizes = [2, 2], [[r], [r]]
    e, p, sz = _dump_box(0.0, 0.0, 1.0, 4.0, 1.5)
    els, props, geom, sizes = els + e, props + p, geom + [1] * 6, sizes + sz
    mats = [copy.deepcopy(AIR_ROW)]
    if emission:
        mats.append(dye_row("LRthin", conc_a=1e-9))
    photon = _photon(n_photons, (0.0, 0.0, 1.0), STRICT if emission else ISOTROPIC, FIXED_WL, 460,
                     "LRthin" if emission else "air", (0.0, 0.0, -1.0))
    return {
        "Elements": np.array(els, dtype=np.float64),
        "ElementProperties": props,
        "ElementGeometry": geom,
        "ElementSize": sizes,
        "Materials": mats,
        "Photon": photon,
    }


def nested_job(n_photons=10):
    """The slab in its detector box, and a second detector box around that one whose walls are never
    reached: three cells inside each other."""
    job = make_job("C2", n_photons)
    e, p, s = _dump_box(0.0, 0.0, 0.15, 6.0, 3.0)
    job["Elements"] = np.concatenate([job["Elements"], np.array(e, dtype=np.float64)])
    job["ElementProperties"] = job["ElementProperties"] + p
    job["ElementGeometry"] = job["ElementGeometry"] + [1] * 6
    job["ElementSize"] = job["ElementSize"] + s
    return job


def twin_job(n_photons=10):
    """Two slabs side by side in one detector box (the box holds two cells: photons between them walk
    the list), the source above the first."""
    job = make_job("C2", n_photons)
    els = job["Elements"].copy()
    els[:6, 0, 0] -= 1.0                   # first slab: x from -3.5 to 1.5
    second = job["Elements"][:6].copy()
    second[:, 0, 0] += 2.8                 # its twin, narrower: x from 1.8 to 3.8
    sizes = copy.deepcopy(job["ElementSize"][:6])
    for k in (0, 1, 4, 5):
        sizes[k][0] = 1.0
    second[2, 0, 0] = 3.8
    second[3, 0, 0] = 1.8
    job["Elements"] = np.concatenate([els, second])
    job["ElementProperties"] = job["ElementProperties"] + copy.deepcopy(job["ElementProperties"][:6])
    job["ElementGeometry"] = job["ElementGeometry"] + [1] * 6
    job["ElementSize"] = job["ElementSize"] + sizes
    job["Photon"][1] = [-1.0, 0.5, 1.0]
    job["Photon"][2] = ISOTROPIC
    return job


def near_floor_job(n_photons=10):
    """The slab 2e-4 cm above the floor of its detector box: the floor comes within the margin of the slab's
    box, so neither the slab nor the shell around it is a place for a confined search."""
    job = make_job("C2", n_photons)
    els = job["Elements"].copy()
    sizes = copy.deepcopy(job["ElementSize"])
    floor = -2e-4
    els[6:10, 0, 2] = 0.5 * (floor + 2.15)   # the box of slab_job spans z = 0.15 -+ 2: shorter walls
    for k in range(6, 10):
        sizes[k][1] = 0.5 * (2.15 - floor)
    els[11, 0, 2] = floor
    job["Elements"] = els
    job["ElementSize"] = sizes
    return job


def mirror_box_job(n_photons=10):
    """The slab in a box of five mirrors and a detector floor: what leaves the slab upwards or sideways
    comes back from the walls (a photon that has met a wall no longer "moves away from the slab" and
    walks the list), everything ends on the floor or in the slab."""
    job = make_job("C2", n_photons)
    props = copy.deepcopy(job["ElementProperties"])
    for k in range(6, 11):
        props[k] = [1]
    job["ElementProperties"] = props
    return job


def glass_pair_job(n_photons=10):
    """Two dye slabs of the same height and depth side by side in a detector box.  Their edges along x are
    interfaces like top and bottom (photons leave and enter through them too), those along y detectors (a slab
    that is glass all around keeps a photon for hundreds of iterations), and the edge of the first that faces
    the second is read out by six detector strips: the gap between the two is a region that ends on a tiled face."""
    def box(x0, x1, mat, strips):
        cx, hx, hy, zc, hz = 0.5 * (x0 + x1), 0.5 * (x1 - x0), 1.5, 0.2, 0.2
        e = [[[x0, 0.0, zc], [1.0, 0.0, 0.0]], [[0.0, hy, zc], [0.0, -1.0, 0.0]], [[0.0, -hy, zc], [0.0, 1.0, 0.0]],
             [[cx, 0.0, zc + hz], [0.0, 0.0, 1.0]], [[cx, 0.0, zc - hz], [0.0, 0.0, 1.0]]]
        for k in (1, 2):
            e[k][0][0] = cx
        p = [[2, "air", mat], [4, 0, 0, ""], [4, 0, 0, ""], [2, "air", mat], [2, mat, "air"]]
        s = [[hy, hz], [hx, hz], [hx, hz], [hx, hy], [hx, hy]]
        if strips:
            for k in range(strips):
                w = hy / strips
                e.append([[x1, -hy + (2 * k + 1) * w, zc], [-1.0, 0.0, 0.0]])
                p.append([4, 0, 0, ""])
                s.append([w, hz])
        else:
            e.append([[x1, 0.0, zc], [-1.0, 0.0, 0.0]])
            p.append([2, mat, "air"])
            s.append([hy, hz])
        return e, p, s
    els, props, sizes = box(-3.0, 0.0, "LR", 6)
    e, p, s = box(0.3, 2.3, "LRO", 0)
    els, props, sizes = els + e, props + p, sizes + s
    e, p, s = _dump_box(0.0, 0.0, 0.2, 4.0, 2.0)
    els, props, sizes = els + e, props + p, sizes + s
    return {"Elements": np.array(els, dtype=np.float64), "ElementProperties": props, "ElementGeometry": [1] * len(els),
            "ElementSize": sizes, "Materials": [copy.deepcopy(AIR_ROW), dye_row("LR"), dye_row("LRO", conc_b=1.0)],
            "Photon": _photon(n_photons, (1.2, 0.1, 1.0), ISOTROPIC, FIXED_WL, 460, "air", (0.0, 0.0, -1.0))}


def nested_pair_job(n_photons=10):
    """Three levels of boxes: two Lumogen Red slabs side by side in a detector box, each with a smaller
    co-doped slab embedded in it.  Between the slabs: regions of the detector box; inside each slab: the shell
    around the one box it holds; the edges along y of the outer slabs are detectors, every other face an interface."""
    def box(x0, x1, hy, z0, z1, inner, outer, y_detectors):
        cx, hx, zc, hz = 0.5 * (x0 + x1), 0.5 * (x1 - x0), 0.5 * (z0 + z1), 0.5 * (z1 - z0)
        e = [[[x0, 0.0, zc], [1.0, 0.0, 0.0]], [[x1, 0.0, zc], [-1.0, 0.0, 0.0]],
             [[cx, hy, zc], [0.0, -1.0, 0.0]], [[cx, -hy, zc], [0.0, 1.0, 0.0]],
             [[cx, 0.0, z1], [0.0, 0.0, 1.0]], [[cx, 0.0, z0], [0.0, 0.0, 1.0]]]
        glass = [2, inner, outer]
        p = [list(glass), list(glass)] + [[4, 0, 0, ""] if y_detectors else list(glass) for _ in range(2)] + [[2, outer, inner], list(glass)]
        return e, p, [[hy, hz], [hy, hz], [hx, hz], [hx, hz], [hx, hy], [hx, hy]]
    els, props, sizes = [], [], []
    for x0, x1, i0, i1 in ((-3.0, -0.2, -2.5, -0.8), (0.2, 3.0, 0.7, 2.5)):
        for part in (box(x0, x1, 1.5, 0.0, 0.6, "LR", "air", True), box(i0, i1, 1.0, 0.15, 0.45, "LRO", "LR", False)):
            els, props, sizes = els + part[0], props + part[1], sizes + part[2]
    e, p, s = _dump_box(0.0, 0.0, 0.5, 4.0, 2.0)
    els, props, sizes = els + e, props + p, sizes + s
    return {"Elements": np.array(els, dtype=np.float64), "ElementProperties": props, "ElementGeometry": [1] * len(els),
            "ElementSize": sizes, "Materials": [copy.deepcopy(AIR_ROW), dye_row("LR"), dye_row("LRO", conc_b=1.0)],
            "Photon": _photon(n_photons, (-1.5, 0.2, 1.6), ISOTROPIC, FIXED_WL, 460, "air", (0.0, 0.0, -1.0))}


def tilted_slab_job(n_photons=10, angle=0.2):
    """The slab of C2 turned about the x axis inside its upright detector box: two closed boxes in two
    different frames, one inside the other."""
    job = rotated_job(base="C2", axis=(1.0, 0.0, 0.0), angle=angle, n_photons=n_photons)
    upright = make_job("C2", n_photons)
    els = np.array(job["Elements"], dtype=np.float64)
    els[6:] = np.array(upright["Elements"], dtype=np.float64)[6:]
    sizes = list(job["ElementSize"])
    sizes[6:] = upright["ElementSize"][6:]
    job["Elements"], job["ElementSize"], job["Photon"] = els, sizes, upright["Photon"]
    return job


def random_boxes_job(seed, n_photons=10):
    """Random axis-aligned scene for the closed-box search: one to three dye slabs of random size at random,
    non-overlapping places (stacked with gaps or side by side) inside a detector box that may have mirror
    walls, random edge types (detector, mirror, interface), sometimes a second box around everything, sometimes a loose
    mirror or detector in the shell; strict or hemispherical source above the first slab."""
    rng = np.random.RandomState(7000 + seed)
    els, props, sizes = [], [], []
    n_slabs = int(rng.randint(1, 4))
    stacked = rng.rand() < 0.5
    names = ["LR", "LRO"]
    z_top = 0.0
    x_next = -3.0
    first = None
    for k in range(n_slabs):
        hx, hy = float(rng.randint(4, 11)) / 8.0, float(rng.randint(6, 13)) / 8.0
        th = float(rng.randint(1, 7)) / 16.0
        if stacked:
            cx, cy = 0.0, 0.0
            z1 = z_top
            z0 = z1 - th
            z_top = z0 - float(rng.randint(1, 5)) / 32.0
        else:
            cx, cy = x_next + hx, float(rng.randint(-2, 3)) / 8.0
            x_next = cx + hx + float(rng.randint(1, 5)) / 16.0
            z1 = float(rng.randint(-2, 3)) / 16.0
            z0 = z1 - th
        zc, hw = 0.5 * (z0 + z1), 0.5 * (z1 - z0)
        mat = names[int(rng.randint(0, 2))]
        e = [[[cx, cy, z0], [0.0, 0.0, 1.0]], [[cx, cy, z1], [0.0, 0.0, 1.0]],
             [[cx + hx, cy, zc], [-1.0, 0.0, 0.0]], [[cx - hx, cy, zc], [1.0, 0.0, 0.0]],
             [[cx, cy + hy, zc], [0.0, -1.0, 0.0]], [[cx, cy - hy, zc], [0.0, 1.0, 0.0]]]
        pr = [[2, mat, "air"], [2, "air", mat]]
        for _ in range(4):              # edges: detector, mirror, or glass like top and bottom
            r = rng.rand()
            pr.append([4, 0, 0, ""] if r < 0.45 else ([1] if r < 0.7 else [2, mat, "air"]))
        sz = [[hx, hy], [hx, hy], [hy, hw], [hy, hw], [hx, hw], [hx, hw]]
        els, props, sizes = els + e, props + pr, sizes + sz
        if first is None:
            first = (cx, cy, z1)
    e2, p2, s2 = _dump_box(0.0, 0.0, -0.5, 5.0, 2.0)
    for k in range(5):                      # the floor stays a detector: every photon ends somewhere
        if rng.rand() < 0.3:
            p2[k] = [1]
    els, props, sizes = els + e2, props + p2, sizes + s2
    if rng.rand() < 0.3:
        e3, p3, s3 = _dump_box(0.0, 0.0, -0.5, 6.5, 3.0)
        els, props, sizes = els + e3, props + p3, sizes + s3
    geometry = [1] * len(els)
    if rng.rand() < 0.3:                    # something loose in the shell: the box around it is no place for a confined search
        els.append([[3.5, -3.0, 1.0], [0.0, 0.0, 1.0]])
        props.append([1] if rng.rand() < 0.5 else [4, 0, 0, ""])
        sizes.append([0.5, 0.75])
        geometry.append(1)
    mats = [copy.deepcopy(AIR_ROW), dye_row("LR"), dye_row("LRO", conc_b=1.0)]
    strict = rng.rand() < 0.5
    # (a strict beam with no special relation to the boxes: on a grid of sixteenths it would meet edges and corners
    # exactly, where float32 and float64 part ways whatever the search)
    direction = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)), -1.0) if strict else (0.0, 0.0, -1.0)
    start = (first[0] + float(rng.uniform(-0.12, 0.12)), first[1] + float(rng.uniform(-0.12, 0.12)), first[2] + 0.75)
    photon = _photon(n_photons, start, STRICT if strict else ISOTROPIC, FIXED_WL, 460, "air", direction)
    return {"Elements": np.array(els, dtype=np.float64), "ElementProperties": props, "ElementGeometry": geometry,
            "ElementSize": sizes, "Materials": mats, "Photon": photon}


def synthetic_am15():
    """Two-column stand-in for the sun table the reference expects in
    `AM15.txt` (reference PhotonObject.py:41; the file is not distributed with
    it): spectral irradiance of a 5778 K black body, 300..1100 nm in 1 nm
    steps, normalised to a peak of 1."""
    lam = np.arange(300.0, 1101.0, 1.0)
    x = 1.4387769e7 / (lam * 5778.0)      # hc/(lambda k T), lambda in nm
    irr = lam ** -5.0 / np.expm1(x)
    return np.stack([lam, irr / irr.max()], axis=1)


CONFIGS = {
    # name: (builder, kwargs)
    "C1_open": (slab_job, dict(thickness=0.012, enclosed=False)),
    "C1": (slab_job, dict(thickness=0.012, enclosed=True)),
    "C2": (slab_job, dict(thickness=0.3, enclosed=True)),
    "C3": (slab_job, dict(thickness=0.3, enclosed=True, donor=True)),
    "C4": (stacked_job, dict()),
    "C5": (slab_job, dict(thickness=0.3, enclosed=True, donor=True, sun=True)),
    # aligned acceptor dyes: the shipped job's style (order parameter below the
    # reference's 1e-10 threshold, dipoles along z; oblique incidence, because at
    # normal incidence the field has no z component and nothing is absorbed) ...
    "C6": (slab_job, dict(thickness=0.3, enclosed=True, aligned=True, align_s=1e-99,
                          direction=(0.6, 0.0, -1.0))),
    # ... and a finite order parameter with an absorption/emission dipole angle,
    # donor present, emission polarisation not tied to the dipole
    "C7": (slab_job, dict(thickness=0.3, enclosed=True, donor=True, aligned=True, align_s=0.3,
                          abs_em_angle=10, respect_pol=False)),
    # every shape (cylinder, circle, triangle, rectangle) and type (interface, lens, mirror, detector)
    "C8": (rod_job, dict()),
    # the options the other configurations leave at their defaults: detectors with a
    # quantum-efficiency table, fixed donor-acceptor distance for the back transfer, a scale factor
    # C2 turned rigidly: no plane is axis-aligned, every element takes the general hit test
    "C2r": (rotated_job, dict(base="C2")),
    "C9": (slab_job, dict(thickness=0.3, enclosed=True, donor=True, eqe=EQE_DEMO, scale=2.0,
                          fix_dist=True, fix_dist_r=3.0)),
    # C2 with every edge of the slab read out by 3 / 14 detectors side by side, and a stack of eight slabs
    "S3": (strip_job, dict(strips=3)),
    "S14": (strip_job, dict(strips=14)),
    "stack8": (stacked_job, dict(n_slabs=8)),
    # the stacked slabs and the segmented read-out turned rigidly out of the world's axes
    "C4r": (rotated_job, dict(base="C4")),
    "S14r": (rotated_job, dict(base="strips14")),
    # random slabs in random boxes (seed 0; rotated_job(base="RB", seed=...) turns any of them)
    "RB": (random_boxes_job, dict(seed=0)),
    "tilted": (tilted_slab_job, dict()),
}


# (bases for rotated_job that are not parity configurations of their own)
CONFIGS["strips14"] = (strip_job, dict(strips=14))


def make_job(name, n_photons=None):
    builder, kwargs = CONFIGS[name]
    kwargs = dict(kwargs)
    if n_photons is not None:
        kwargs["n_photons"] = n_photons
    return builder(**kwargs)
