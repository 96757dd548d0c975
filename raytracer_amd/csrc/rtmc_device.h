// rtmc_device.h -- device-side photon transport for gfx950 (MI355X).
//
// One photon lives in the registers of one lane from launch to its end; a
// wave keeps its 64 lanes busy by handing new photon ids to lanes whose
// photon has ended (ballot + prefix count).  Everything a photon looks up is
// staged in LDS at kernel start: the element records used by the optical
// operations, the media, and per wavelength-id the absorption coefficients and
// inverse-CDF thresholds.  The records used by the nearest-hit search are read
// with wave-uniform indices so that they travel through the scalar cache into
// SGPRs.
//
// The code is a template over the arithmetic type: `float` is the production
// instantiation, `double` exists to validate the logic against the float64
// CPU checker.  Variates come either from Philox (production; its launch and
// re-emission code is the `renew_*` family) or from a tape of recorded values
// consumed in the reference's order (parity tests; `launch_in_reference_order`
// and `emit_in_reference_order`).  Both share the loop iteration `photon_step`.
//
// Reference file:line citations name the statement of mohrezaeib/RayTracer
// that a routine stands for.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtmc.h"

namespace rtmc {

// ---------------------------------------------------------------------------
// records shared between host (scene builder) and device
// ---------------------------------------------------------------------------

// One element as the optical operation sees it (first three 16-byte groups,
// gathered per lane from LDS) and as the general hit test sees it (last three
// groups, meaning by shape):
//   rectangle  plane n.x = d0; in-plane coordinates t1 = a.x - c1, t2 = b.x - c2, half extents h1, h2
//   circle     a = centre, h1 = radius^2
//   triangle   a, b = in-plane unit axes; corners in that frame (c1,c2) (h1,h2) (g1,g2)
//   cylinder   a = unit axis, b = an axis point, c1 = radius^2, wall between axial
//              coordinates h1 <= a.(x - b) <= h2; the plane of an operation is the
//              tangent plane at the point hit, so n and d0 are not used
template <typename real>
struct DevElem {
    real nx, ny, nz, d0;
    int32_t type;   // RTMC_TYPE_* | RTMC_GEOM_* << 8
    int32_t info;   // interface: material1 | material2 << 8; dump: (eqe table + 1) | detector slot << 8;
                    // interface and mirror: cell the element is a face of << 16 (7 bits) | its normal points out of
                    // that cell << 23 | cell on the outer side of the face << 24 (DevCell)
    int32_t nid1;   // interface: name id of material1 (Elements.py:45 compares names)
    int32_t slots;  // interface: (absorption slot of material1 + 1) | (slot of material2 + 1) << 16
    real n1v, n2v;  // interface: refractive indices of material1 / material2
    real focal;     // lens: focal length
    real sz;        // z coordinate of the support point (PhotonObject.py:178)
    real ax, ay, az, c1;
    real bx, by, bz, c2;
    real h1, h2, g1, g2;
};

// An element as the nearest-hit search sees it.  The table is grouped by the
// orientation of the normal: first the rectangles normal to x, then y, then z
// (in-plane axes i = k+1, j = k+2 cyclically: plane coordinate p0, centre
// (ci, cj), half extents (hi, hj)), then those of general orientation (only
// `index` is used, the test reads DevElem).  Inside a group the list order of
// the scene is kept.
template <typename real>
struct HitRec {
    real p0, ci, cj, hi, hj;
    int32_t index;  // position in the scene's element list
};

// The search reads the table two records at a time (the twelve words of a
// float32 chunk come in two scalar loads behind one wait); every axis group is padded to a multiple of two with records that
// can never be hit (negative half extents).  Four at a time halves the number
// of waits but keeps 32 SGPRs busy, which the allocator pays for with spills.
constexpr int HIT_CHUNK = 2;
template <typename real>
struct alignas(64) HitChunk {
    HitRec<real> r[HIT_CHUNK];
};

// A rectangle whose normal and in-plane axes are perpendicular to each other is
// axis-aligned in *some* orthonormal frame.  Rectangles that share a frame
// (the faces of a tilted box, a tilted stack of slabs) are tested in that
// frame's coordinates with the cheap axis test after one rotation of the line
// per frame and search.  A frame's header sits in the hit table in front of its
// three groups and has the size of a chunk.
template <typename real>
struct alignas(64) FrameHdr {
    real R[9];     // rows = the frame's axes in world coordinates: x' = R x
    int32_t n[3];  // chunks in the frame's x' | y' | z' groups
};
static_assert(sizeof(FrameHdr<float>) == sizeof(HitChunk<float>) && sizeof(FrameHdr<double>) == sizeof(HitChunk<double>),
              "a frame header takes the place of one chunk");

// Many rectangles of one axis group in one plane (a segmented read-out along an edge, the side
// faces of a tall stack): the plane is intersected once and the rectangles that can contain the
// landing point are looked up instead of being tried one after the other.  Such a *run* has a
// header in the hit table behind the other records of its group (the group's count holds the chunks
// of those in its low half and the number of runs in its high half) and, in the LDS image, a grid of
// cells over the members' bounding box in the plane: a cell names the (at most eight) members that
// touch it, in list order, in two words.  Float32 tables only.
template <typename real>
struct alignas(64) RunHdr {
    real p0;                  // the plane
    real gi0, gj0;            // corner of the grid (in-plane coordinates i, j)
    real inv_wi, inv_wj;      // 1 / cell size
    int32_t ni_1, nj_1;       // cells along i and j, minus one
    int32_t cell_off;         // first cell word of this run (index into Lds::run_cells)
    int32_t mem_off;          // first member record (index into Lds::run_members)
    int32_t pad[7];
};
static_assert(sizeof(RunHdr<float>) == sizeof(HitChunk<float>), "a run header takes the place of one chunk");
// the part of the header the search reads
template <typename real>
struct RunHead {
    real p0, gi0, gj0, inv_wi, inv_wj;
    int32_t ni_1, nj_1, cell_off, mem_off;
};
// A member of a run as the lanes gather it from LDS.
template <typename real>
struct alignas(16) RunMember {
    real ci, cj, hi, hj;
    int32_t index;
    int32_t pad[3];
};
constexpr int RUN_CELL_SLOTS = 8;       // members a cell can name (one byte each in two 32-bit words, 0xFF = none)

// A *cell*: six rectangles of the world frame that close an axis-aligned box.  A photon that is
// inside such a box, with nothing else in it, can only meet one of the three faces its direction
// points at, so its search needs no walk over the element list: see confined_hit.  Cells may be
// nested (a slab inside a detector box); a photon carries the number of the innermost cell it is
// in (Photon::flags) and changes it when it is refracted through a face.  Float32 tables only.
// Record 0 of the table stands for "no cell" and fails every test.
struct alignas(16) DevCell {
    float plo[3], phi[3];  // plane coordinate (HitRec::p0) of the faces normal to x | y | z, low and high side
    float c[3];            // centre of the box
    float hw[3];           // half extents plus a margin: farther out a point is not taken to be in the cell
                           // (negative: the record is not usable, every search from it fails)
    int32_t idlo[3], idhi[3];  // element indices of the low and high faces; -(g + 2): the face is tiled, see TileGrid g
    int32_t frame;             // the orthonormal frame whose coordinates all of the above are in (0: the world's)
    uint32_t pad;
};
static_assert(sizeof(DevCell) == 80, "five 16-byte LDS reads per lane");
// Every cell has two records: [2 cell] for a photon that is simply in it, [2 cell + 1] for one whose last operation
// was on a face of a cell nested in it, from which it now moves away (F_OUTBOUND: refracted out of the nested cell
// or reflected off it from outside).  A record is usable if
//   [2 cell]      nothing but the cell's faces touches its box: every photon in it is confined;
//   [2 cell + 1]  the same, or the box holds exactly one other cell and nothing else: the photon that moves away
//                 from that cell cannot meet it again (it is convex) and is confined to the shell around it.
// A box that holds several cells gets further cells that are no boxes of rectangles but empty *regions* of it: the
// space in front of a face of a nested cell, up to the wall or the neighbour's face it ends on, with open sides
// (face index CONFINED_NONE: the search gives no answer for a photon that leaves through one, and half extents drawn
// in by the margin instead of pushed out, so that what happens at the outline is left to the list walk).  The faces
// of the nested cell name the region as the cell on their outer side (DevElem::info).  Record [2 region + 1] spans
// the whole enclosing box beyond the plane of the face where no nested cell reaches that far.  The number a photon
// carries is a hint that the tests of the search check: wherever it is, a record whose box holds it, and nothing
// else, gives the answer of the list walk.  (Only a shell -- [2 cell + 1] of a box with a cell in it -- relies on
// the photon being where its number says and moving as F_OUTBOUND says: both are set by the operation on the
// nested cell's face, from the geometry of that operation alone.)
// In front of the cell records: what a launched photon starts with.
struct alignas(16) CellHeader {
    float preset_t;          // strict source: line parameter and element of the first hit of every photon,
    int32_t preset_e;        //   found once per scene by the search itself (preset_hit_kernel); -1: none
    uint32_t launch_flags;   // Photon::flags of a launched photon: its cell << F_CELL_SHIFT, | F_PRESET if there is a preset,
                             // | F_LIST if there is none and the source's cell is no place for a confined search
    int32_t n_cells;
};
// A face of a cell that is tiled by several rectangles (DevCell::idlo/idhi < -1 name grid -(id + 2)): the grid
// of a run of coplanar rectangles (RunHdr) without its plane, gathered per lane.
struct alignas(16) TileGrid {
    float gi0, gj0, inv_wi, inv_wj;
    int32_t ni_1, nj_1, cell_off, mem_off;
};
constexpr int32_t CONFINED_NONE = (int32_t)0x80000000;   // confined_hit: no answer from the cell

// One medium.
template <typename real>
struct DevMat {
    real n;
    int32_t flags;     // bit 0: absorber, bit 1: its acceptor dyes are aligned
    int32_t name_id;
    int32_t abs_slot;  // row block in the absorption table, -1 if not an absorber
    int32_t n_p;       // 3: donor pathways incl. transfer, 2: no acceptor present
    uint32_t pT[3];    // cumulative thresholds of the donor pathways (x 2^32)
    uint32_t pAT[3];   // acceptor pathways
    int32_t cdfD_off, cdfD_len, widD_off;  // donor emission: threshold table, first wavelength id
    int32_t cdfA_off, cdfA_len, widA_off;  // acceptor emission
    int32_t gD_off, gA_off;                // their guide tables
    int32_t guide_shift;                   // 32 - the number of leading bits of a variate that name a guide bucket
};

// Absorption data of one (absorbing medium, wavelength id).
template <typename real>
struct DevAbs {
    real b_a;        // acceptor attenuation coefficient [1/cm] (PhotonObject.py:288)
    real b_d;        // donor                                   (PhotonObject.py:289)
    uint32_t coinT;  // ceil(b_d / (b_d + b_a) * 2^32): "absorbed by the donor" threshold
};

// Orientation data of a medium with aligned acceptor dyes (one record per
// medium, zeros for the others).
template <typename real>
struct DevOrient {
    real P, Q;            // orientation integral |P (x^2 + y^2) + Q z^2| of a unit field vector in the alignment frame
    real f[9];            // alignment frame: columns n1, n2, AlignVect (row-major)
    real S;               // width of the Gaussian tilt of the dipoles about AlignVect [rad]
    real sin_ae, cos_ae;  // angle between absorption and emission dipole
    int32_t flags;        // bit 0: emission polarised along the dipole, bit 1: sin_ae/cos_ae in use
    int32_t pad;
};

template <typename real>
struct DevSource {
    real loc[3];
    real dir[3];     // unit (strict mode)
    real frame[9];   // hemisphere frame, row-major
    real inv_scale;  // 1 / ScaleFactor
    int32_t dir_mode, wl_mode, start_mat, wid_fixed;
    int32_t cdf_off, cdf_len;  // source spectrum thresholds; wavelength id = row
    int32_t guide_off;         // its guide table
    int32_t guide_shift;       // 32 - bits per guide bucket number (as DevMat::guide_shift)
    int32_t has_area;          // launch positions spread over a rectangle
    real area[2];              // its full widths along x and y
};

// Offsets (in bytes) into the LDS image.
struct LdsLayout {
    uint32_t elems, mats, orient, abs, cdf, guide, fx, bin, run_cells, run_members, cells, tile_grids, cell_frames, acc, hist, total;
    uint32_t n_wid;   // wavelength ids
    uint32_t n_bins;  // distinct 1-nm detector bins the wavelength ids fall into
    uint32_t n_dump;  // detector elements
    uint32_t hist_in_hbm;  // the detector histogram does not fit beside the tables: it is counted in HBM
};

// Accumulators a workgroup keeps in LDS and flushes once:
//   uint64 acc[]: scalars[ACC_SCALARS] bounce[n_el] spectral[n_el] energy[n_el]
//   uint32 hist[]: n_dump x n_bins detector histogram
enum { ACC_STEPS = 0, ACC_ABS_ONCE, ACC_ABSORPTIONS, ACC_IC, ACC_CAPPED, ACC_LOST, ACC_STALE, ACC_SCALARS };

template <typename real>
struct DevScene {
    const HitChunk<real> *hits;  // global, wave-uniform reads; layout: see nearest_valid_hit
    const DevElem<real> *elems;  // global copy for wave-uniform reads (general elements)
    int32_t n_group[4];          // chunks in the world frame's x | y | z groups; [3] = chunks of the
                                 // general group | number of further frames << 16
    const uint32_t *lds_image;   // elems | mats | abs | cdf, copied to LDS by every workgroup
    const double *wl;            // wavelength of each id

    const uint32_t *eqeT;        // [n_eqe][n_wid] detection thresholds
    DevSource<real> src;
    LdsLayout lay;
    int32_t n_el, n_mat, max_iter;
    int32_t use_guard;
};

struct DevTallies {
    unsigned long long *scalars;    // ACC_SCALARS
    unsigned long long *bounce;     // n_el
    unsigned long long *spectral;   // n_el
    unsigned long long *energy_fx;  // n_el
    unsigned long long *hist;       // n_dump * n_bins
    uint32_t *private_hist;         // scenes that count detector hits in HBM: one n_dump * n_bins block per workgroup
};

// The kernels receive the scene as a pointer to a device-resident DevScene and
// read it through the constant address space: every field is then fetched
// with a scalar load where it is used instead of being held in SGPRs for the
// whole kernel (the struct is larger than the SGPR file can spare).
template <typename real>
using SceneArg = const DevScene<real> __attribute__((address_space(4))) &;

// The few scene values every loop iteration needs, copied out of the scene
// once per kernel so that they stay in SGPRs.
template <typename real>
struct HotScene {
    const HitChunk<real> *hits;
    const DevElem<real> *elems;
    const uint32_t *eqeT;
    int32_t n_group[4];
    int32_t n_wid, max_iter;
    real inv_scale;
    int32_t use_guard;  // float32 kernels with FEAT_GUARD: 0 no guard distance, 1 on every search (the scene has
                        // coplanar overlapping elements), 2 until a photon's first absorption (a strict source
                        // is aimed at a seam: up to there the landing points are the same for every photon)
};

// Code a scene may need beyond rectangles with isotropic dyes; kernels are
// instantiated per combination so that the common case carries none of it.
// FEAT_AREA (launch points spread over a rectangle) is only instantiated together
// with the other two: such a source is rare, and a wave-uniform branch in the
// lean kernels would still cost them the scalar registers of a third Philox block.
// FEAT_SPECTRA: the source draws its wavelength from a spectrum or a detector
// has an efficiency table.  Without it every photon starts at the one source
// wavelength, the energy increment of a detected photon is a table entry, and no
// detector draws; the kernels then hold neither that code nor its pointers.
// FEAT_BACK: some medium allows the back transfer acceptor -> donor
// (MonteCarloFRET.py:187-190), which reopens the pathway choice with a Philox
// block of its own in a loop.
// FEAT_FRAMES: some rectangle lies in a frame other than the world's (the hit
// table then holds frame headers and the search a loop over them); kernels
// with FEAT_SHAPES always have it.
// FEAT_GUARD (float32; only instantiated together with everything else): see guard_distance.
// FEAT_RUNS (float32): some axis group holds three or more rectangles in one plane (the edge faces of
// stacked slabs, a segmented read-out); the hit table then lists them as runs behind the plane they
// share, see RunHdr.
// FEAT_CELLS (float32, rectangle-only scenes of the world frame; not part of FEAT_ALL): some rectangles close
// boxes (DevCell); photons inside them find their next element without walking the list, the others wait for a
// pass of the wave loop that walks it for all of them together.
// FEAT_HBMHIST (with everything else, or with FEAT_CELLS): detectors x wavelength bins exceed what the
// LDS can hold beside the tables, or would cost a kernel with cells a wave per SIMD there; every workgroup
// then counts hits with atomics in a histogram of its own in HBM (one shared histogram is bounded by
// the rate of atomics on its few hot words: 2e9 photons/s whatever the scene) and a second kernel adds
// the copies up.
enum : int { FEAT_ALIGNED = 1, FEAT_SHAPES = 2, FEAT_AREA = 4, FEAT_SPECTRA = 8, FEAT_BACK = 16, FEAT_FRAMES = 32, FEAT_GUARD = 64,
             FEAT_HBMHIST = 128, FEAT_RUNS = 256, FEAT_ALL = 511, FEAT_CELLS = 512 };

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------
// arithmetic helpers
// ---------------------------------------------------------------------------
template <typename real> struct M;
template <> struct M<float> {
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
    // v_sqrt_f32 (1 ulp) without the correctly-rounded fix-up sequence
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }  // correctly rounded
    static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ float log(float x) { return ::logf(x); }
    static __device__ __forceinline__ void sincospi(float x, float *s, float *c) { ::sincospif(x, s, c); }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c) { ::sincosf(x, s, c); }
    static __device__ __forceinline__ float acos(float x) { return ::acosf(x); }
    static __device__ __forceinline__ float asin(float x) { return ::asinf(x); }
    static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
    static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
    // sin/cos of 2 pi x for x in [0, 1): v_sin_f32 / v_cos_f32 take their argument in turns
    static __device__ __forceinline__ void sincos_turns(float x, float *s, float *c)
    {
        *s = __builtin_amdgcn_sinf(x);
        *c = __builtin_amdgcn_cosf(x);
    }
    // -log(1 - u) for the free path of a sampled photon: v_log_f32 (log2) scaled
    static __device__ __forceinline__ float neg_log1m(float u) { return -0.6931471805599453f * __builtin_amdgcn_logf(1.0f - u); }
    static __device__ __forceinline__ float copysign(float m, float s) { return __builtin_copysignf(m, s); }
    static __device__ __forceinline__ float huge() { return 3.0e38f; }
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static constexpr bool has_self_hit_eps = false;  // the reference's 1e-8 cm is below float32 resolution here
};
template <> struct M<double> {
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / __builtin_sqrt(x); }
    static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
    static __device__ __forceinline__ double sqrt_rn(double x) { return __builtin_sqrt(x); }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ void sincospi(double x, double *s, double *c) { ::sincospi(x, s, c); }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double acos(double x) { return ::acos(x); }
    static __device__ __forceinline__ double asin(double x) { return ::asin(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ void sincos_turns(double x, double *s, double *c) { ::sincospi(2.0 * x, s, c); }
    static __device__ __forceinline__ double neg_log1m(double u) { return -::log(1.0 - u); }
    static __device__ __forceinline__ double copysign(double m, double s) { return __builtin_copysign(m, s); }
    static __device__ __forceinline__ double huge() { return 1.0e300; }
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static constexpr bool has_self_hit_eps = true;   // RayTrace.py:134 taken literally
};

template <typename real>
__device__ __forceinline__ int sgn_i(real x) { return (x > real(0)) - (x < real(0)); }

// ---------------------------------------------------------------------------
// variates
// ---------------------------------------------------------------------------
// Philox4x32-10.  key = (seed lo, seed hi); counter = (photon id lo, hi, event, 0).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply per lane pair of the round (v_mad_u64_u32)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // (a xor of three in one instruction: v_bitop3_b32 with the parity truth table)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Philox mode: variates come in blocks of four 32-bit words, one block per
// counter value (photon id lo, hi, loop iteration, purpose); a word is used as
// a 24-bit uniform k/2^24.
//   purpose 0          the iteration's own block, computed for every photon at
//                      the top of the iteration and used by whatever happens in
//                      it: interface [Fresnel test, free path on crossing],
//                      detector [detection test], absorption [donor/acceptor
//                      coin, donor pathway, acceptor pathway, wavelength]
//   purpose r = 1, 2.. the r-th back transfer acceptor -> donor of that absorption
//   BLK_EMIT (+1)      re-emission after that absorption [direction, direction,
//                      polarisation, free path] (aligned acceptors: two blocks)
//   BLK_LAUNCH         launch (iteration 0) [direction, direction, polarisation, wavelength]
// In tape mode the values are read sequentially in the order the reference
// draws them.
enum : uint32_t { BLK_STEP = 0u, BLK_EMIT = 0x80000000u, BLK_LAUNCH = 0xC0000000u };
// BLK_LAUNCH + 1: launch position inside the source's rectangle [x, y] (extended sources only)
template <typename real, bool TAPE>
struct Rng;

template <typename real>
struct Rng<real, false> {
    uint32_t k0, k1, id0, id1;
    uint32_t w[4];
#ifdef RTMC_ABLATE_PHILOX
    __device__ __forceinline__ void block(uint32_t iter, uint32_t purpose)
    {
        uint32_t h = id0 * 0x9E3779B9u + (iter ^ purpose) * 0x85EBCA6Bu + k0;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        w[0] = h; w[1] = h * 0x297A2D39u; w[2] = h ^ 0xA5A5A5A5u; w[3] = h * 0x9E3779B9u;
    }
#else
    __device__ __forceinline__ void block(uint32_t iter, uint32_t purpose) { philox4x32_10(id0, id1, iter, purpose, k0, k1, w); }
#endif
    // the uniform as an integer: floor(u * 2^32)
    __device__ __forceinline__ uint32_t u32(int slot) { return w[slot] & 0xFFFFFF00u; }
    __device__ __forceinline__ real uniform(int slot) { return real(w[slot] >> 8) * real(1.0 / 16777216.0); }
    // unit exponential, numpy legacy form -log(1 - U)
    __device__ __forceinline__ real expo(int slot) { return M<real>::neg_log1m(uniform(slot)); }
    __device__ __forceinline__ void normal3(real &a, real &b, real &c) { a = b = c = real(0); }
    __device__ __forceinline__ bool consume_dead_draws() const { return false; }
};

template <typename real>
struct Rng<real, true> {
    const double *tape;
    uint32_t pos, end;
    bool overrun;
    // false: the tape holds every variate as the reference drew it (uniform, normal or
    // exponential); true: uniforms only, turned into the other kinds here
    bool uniforms_only;
    __device__ __forceinline__ double next()
    {
        if (pos >= end) { overrun = true; return 0.5; }
        return tape[pos++];
    }
    __device__ __forceinline__ double next_normal()
    {
        const double u = next();
        if (!uniforms_only) return u;
        const double lo = 1.1102230246251565e-16;
        return normcdfinv(u < lo ? lo : (u > 1.0 - lo ? 1.0 - lo : u));
    }
    __device__ __forceinline__ void block(uint32_t, uint32_t) {}
    __device__ __forceinline__ uint32_t u32(int)
    {
        double v = next() * 4294967296.0;
        return v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
    }
    __device__ __forceinline__ real uniform(int) { return real(next()); }
    __device__ __forceinline__ real expo(int)
    {
        const double u = next();
        return real(uniforms_only ? -log1p(-u) : u);
    }
    __device__ __forceinline__ void normal3(real &a, real &b, real &c)
    {
        a = real(next_normal()); b = real(next_normal()); c = real(next_normal());
    }
    __device__ __forceinline__ real normal1() { return real(next_normal()); }
    // the reference draws a free path even for a photon that was just lost to
    // internal conversion (PhotonObject.py:250/258); the tape holds that value
    __device__ __forceinline__ bool consume_dead_draws() const { return true; }
};

// ---------------------------------------------------------------------------
// photon
// ---------------------------------------------------------------------------
template <typename real>
struct Photon {
    real Lx, Ly, Lz, Dx, Dy, Dz, Ex, Ey, Ez;
    real dist;       // DistanceLeftToAbsorption
    real psi;        // Photon.Psi (aligned media only; dead otherwise)
    real hitd;       // SmallestDistance of the last iteration
    real best_t;     // signed line parameter of the last candidate
    real oldx, oldy, oldz;  // RayTrace.OldPoint (read back by the step-wise interface only)
    // Admissibility of the next candidate at line parameter t (see photon_step):
    // (t * adm_q > 0) or adm_all.
    real adm_q;
    // Kernels with FEAT_GUARD: no candidate within this distance of the photon (see
    // landing_uncertainty); zero for a photon that does not sit on an element.
    real guard;
    int32_t wid, wid0;
    int32_t mat;
    int32_t last_el;  // RayTrace.FirstEl, -1 none
    int32_t status;
    int32_t iter;
    uint32_t flags;   // F_* bits
};
enum : uint32_t { F_ADM_ALL = 1u, F_WAS_EL = 2u, F_REFLECTED = 4u, F_ABS_ONCE = 8u,
                  // kernels with FEAT_CELLS:
                  F_PRESET = 32u,    // freshly launched from a strict source: its first hit (best_t, last_el) is the scene's preset
                  F_LIST = 64u,      // the photon's last search from its cell failed and its cell record has not changed since:
                                     // the next one walks the list without trying (a photon outside every cell would
                                     // otherwise spend every other pass of the loop on a search that cannot succeed)
                  F_OUTBOUND = 128u, // the photon sits on a face of a cell nested in its own and moves away from it
                  F_CELL_SHIFT = 8u, F_CELL_MASK = 0xFF00u,  // the cell the photon is in (0 = none; 255 = unknown for the rest of its
                                                             // life: it walks the list, see F_LIST)
                  CELL_UNKNOWN = 255u,
                  F_CELL_RECORD_SHIFT = 7u, F_CELL_RECORD_MASK = 0x1FFu };  // 2 cell + outbound: the photon's cell record
template <typename real>
__device__ __forceinline__ void set_flag(Photon<real> &p, uint32_t bit, bool on)
{
    p.flags = on ? (p.flags | bit) : (p.flags & ~bit);
}

// Views into the LDS image.
template <typename real>
struct Lds {
    const DevElem<real> *elems;
    const DevMat<real> *mats;
    const DevOrient<real> *orient;
    const DevAbs<real> *abs;
    const uint32_t *cdf;
    const uint16_t *guide;  // per threshold table 2^bits + 1 entries: thresholds below bucket b's first value
    const uint32_t *fx;    // per wavelength id: llrint(wl_source / wl * 2^24) for a single-line source
    const uint16_t *bin;   // per wavelength id: compact detector bin
    const uint32_t *run_cells;              // grids of the runs of coplanar rectangles
    const RunMember<real> *run_members;     // their members
    const CellHeader *cells;                // header, then the DevCell records (record 0 = no cell)
    const TileGrid *tile_grids;             // grids of the tiled faces of cells
    const float *cell_frames;               // per frame that holds cells: its axes as rows (9 values in 12)
    unsigned long long *acc;
    uint32_t *hist;
};

// Rows of the LDS tables are addressed with the 24-bit multiply: it runs at the full vector
// rate, a 32-bit integer multiply at a quarter of it, and indices and record sizes are far
// below 2^24.
__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
template <typename T>
__device__ __forceinline__ const T &row(const T *base, int index)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + __umul24((unsigned)index, (unsigned)sizeof(T)));
}

// number of thresholds <= v  ==  numpy.searchsorted(cdf, u, side='right')
__device__ __forceinline__ int cdf_pick(const uint32_t *T, int n, uint32_t v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (T[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo < n ? lo : n - 1;
}

// The same count found from a guide table: G[b] = number of thresholds below
// b << shift, so the answer lies in [G[b], G[b + 1]] for b = v >> shift and the
// bisection runs over a handful of entries instead of the table -- over none at all
// where a bucket holds no threshold.  The host gives the tables as many buckets
// (2^8, 2^10 or 2^12) as the LDS can hold without costing the kernel a wave per SIMD.
constexpr int GUIDE_BITS_MIN = 8, GUIDE_BITS_MAX = 12;
__device__ __forceinline__ int cdf_pick_guided(const uint32_t *T, int n, const uint16_t *G, uint32_t v, int shift)
{
    const uint32_t b = v >> shift;
    int lo = G[b], hi = G[b + 1];
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (T[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo < n ? lo : n - 1;
}

template <typename real>
__device__ __forceinline__ void normalize3(real &x, real &y, real &z)
{
    real s = M<real>::rsqrt(x * x + y * y + z * z);
    x *= s; y *= s; z *= s;
}

// PhotonObject.py:295-299 + 267-281: the orientation integral Psi of the
// field vector E, taken in the frame whose z axis is the alignment vector.
template <typename real>
__device__ __forceinline__ real orientation_integral(const DevOrient<real> &o, real Ex, real Ey, real Ez)
{
    real x = Ex * o.f[0] + Ey * o.f[3] + Ez * o.f[6];  // E (row) times frame
    real y = Ex * o.f[1] + Ey * o.f[4] + Ez * o.f[7];
    real z = Ex * o.f[2] + Ey * o.f[5] + Ez * o.f[8];
    real xy = x * x + y * y, zz = z * z;
    real ol = M<real>::abs(o.P * xy + o.Q * zz) * M<real>::rcp(xy + zz);
    return ol < real(1e-10) ? real(0) : ol;
}

// PhotonObject.py:283-305 LamBeerDistance for absorbing medium `mat` (its
// absorption rows start at `slot`), `e` a unit exponential.  In a medium with
// aligned acceptors the acceptor coefficient is weighted by 3 Psi and Psi is
// kept on the photon (PhotonObject.py:291-302).
template <bool ALIGNED, typename real>
__device__ __forceinline__ real free_path(const HotScene<real> &sc, const Lds<real> &lds, int mat, int slot,
                                          Photon<real> &p, real e)
{
    const DevAbs<real> &a = row(lds.abs, mul24(slot, sc.n_wid) + p.wid);
    real b_a = a.b_a;
    if (ALIGNED) {
        if (row(lds.mats, mat).flags & 2) {
            p.psi = orientation_integral(row(lds.orient, mat), p.Ex, p.Ey, p.Ez);
            b_a *= p.psi * real(3);
        }
    }
    real b = b_a + a.b_d;
    real mfp = b > real(0) ? M<real>::rcp(b) : M<real>::huge();
    return sc.inv_scale * (mfp * e);
}

// PhotonObject.py:235-244: was the photon absorbed by the donor?  One uniform.
template <bool ALIGNED, typename real, bool TAPE>
__device__ __forceinline__ bool absorbed_by_donor(const DevMat<real> &m, const DevAbs<real> &a, const Photon<real> &p,
                                                  Rng<real, TAPE> &g)
{
    if (ALIGNED) {
        if (m.flags & 2) {  // PhotonObject.py:238-241
            real b_a = a.b_a * (p.psi * real(3));
            return g.uniform(0) < a.b_d / (a.b_d + b_a);  // 0/0 compares false: acceptor
        }
    }
    return g.u32(0) < a.coinT;
}

// What the next nearest-hit search may accept, from the operation just
// executed.  `side` is the side (-1, 0, +1) of the segment's start point
// (RayTrace.OldPoint) relative to the operation plane, `nd` = n . D_new.
//   Elements.py:26-28   mirror:    candidate on the same side as the old point
//   Elements.py:113-114 interface: same side iff the photon was reflected
//   Elements.py:131-132 lens:      other side
// A candidate at parameter t lies on side sign(t * nd) (the photon itself is on
// the plane).  side == 0 happens when an operation is re-executed in place
// (nothing was found, RayTrace.py:124-154): then no candidate is "on the same
// side" and every candidate is "on another side".
template <typename real>
__device__ __forceinline__ void set_admissibility(Photon<real> &p, int type, int side, real nd)
{
    const bool want_same = type == RTMC_TYPE_MIRROR ? true
                         : (type == RTMC_TYPE_INTERFACE ? (p.flags & F_REFLECTED) != 0u : false);
    set_flag(p, F_ADM_ALL, (side == 0) & !want_same);
    const real q = nd * real(side);  // zero when the old point has no side
    p.adm_q = want_same ? q : -q;
}

// A photon that does not come from an element (launch, re-emission) accepts
// what lies ahead: t > 0 (RayTrace.py:143-150).
template <typename real>
__device__ __forceinline__ void set_admissibility_forward(Photon<real> &p)
{
    p.flags &= ~F_ADM_ALL;
    p.adm_q = real(1);
    p.guard = real(0);
}

template <typename real>
__device__ __forceinline__ void init_photon_common(SceneArg<real> sc, Photon<real> &p)
{
    p.wid0 = p.wid;
    p.mat = sc.src.start_mat;
    p.dist = real(0);
    p.psi = real(0);
    p.hitd = M<real>::inf();
    p.best_t = real(0);
    p.oldx = p.Lx; p.oldy = p.Ly; p.oldz = p.Lz;
    p.last_el = -1;
    p.status = RTMC_ST_START;
    p.iter = 0;
    p.flags = 0u;
    set_admissibility_forward(p);
}

// ---------------------------------------------------------------------------
// launch and re-emission, reference order of draws (tape mode)
// ---------------------------------------------------------------------------

// PhotonObject.py:107-114 UpdatePolarizationRandom: E = unit(D x unit(v)), v ~ N(0,1)^3
template <typename real, bool TAPE>
__device__ __forceinline__ void random_polarization(Photon<real> &p, Rng<real, TAPE> &g)
{
    real vx, vy, vz;
    g.normal3(vx, vy, vz);
    normalize3(vx, vy, vz);
    real ex = p.Dy * vz - p.Dz * vy, ey = p.Dz * vx - p.Dx * vz, ez = p.Dx * vy - p.Dy * vx;
    normalize3(ex, ey, ez);
    p.Ex = ex; p.Ey = ey; p.Ez = ez;
}

// RTUtils.py:102-108 GenerateDirectionSine
template <typename real, bool TAPE>
__device__ __forceinline__ void isotropic_direction(Photon<real> &p, Rng<real, TAPE> &g)
{
    real x, y, z;
    g.normal3(x, y, z);
    normalize3(x, y, z);
    p.Dx = x; p.Dy = y; p.Dz = z;
}

template <typename real, typename Source>
__device__ __forceinline__ void apply_source_frame(const Source &s, Photon<real> &p)
{
    real x = p.Dx, y = p.Dy, z = M<real>::abs(p.Dz);  // PhotonObject.py:70-72
    p.Dx = x * s.frame[0] + y * s.frame[3] + z * s.frame[6];
    p.Dy = x * s.frame[1] + y * s.frame[4] + z * s.frame[7];
    p.Dz = x * s.frame[2] + y * s.frame[5] + z * s.frame[8];
    normalize3(p.Dx, p.Dy, p.Dz);
}

// PhotonObject.py:49-97 ResetPhoton / GeneratePhoton
template <typename real>
__device__ __forceinline__ void launch_in_reference_order(SceneArg<real> sc, const Lds<real> &lds,
                                                          Photon<real> &p, Rng<real, true> &g)
{
    const auto &s = sc.src;
    p.Lx = s.loc[0]; p.Ly = s.loc[1]; p.Lz = s.loc[2];
    if (s.dir_mode == 1) {
        isotropic_direction(p, g);
        apply_source_frame(s, p);
    } else {
        p.Dx = s.dir[0]; p.Dy = s.dir[1]; p.Dz = s.dir[2];
    }
    if (s.has_area) {  // PhotonObject.py:76,84 (commented out in the reference): rand() for x, then for y
        p.Lx += (g.uniform(0) - real(0.5)) * s.area[0];
        p.Ly += (g.uniform(0) - real(0.5)) * s.area[1];
    }
    p.wid = s.wl_mode == 1 ? cdf_pick(lds.cdf + s.cdf_off, s.cdf_len, g.u32(0)) : s.wid_fixed;
    random_polarization(p, g);
    init_photon_common(sc, p);
}

// MonteCarloFRET.py:63-193 RunSimulation, pathway choice for isotropic
// (non-aligned) dyes.  Word use of the iteration's block: 0 coin (already
// consumed), 1 donor pathway, 2 acceptor pathway, 3 wavelength; a back
// transfer opens the next block of this iteration.  Returns 0: internal conversion,
// 1: donor fluoresces, 2: acceptor fluoresces; `v_lambda` is the wavelength
// uniform in Philox mode (in tape mode it is drawn later, in reference order).
template <bool BACK, typename real, bool TAPE>
__device__ __forceinline__ int fret_decide(const DevMat<real> &m, Rng<real, TAPE> &g, bool in_donor,
                                           uint32_t iter, uint32_t &v_lambda)
{
    uint32_t round = 0u;
    for (;;) {
        if (round) g.block(iter, BLK_STEP + round);
        round++;
        v_lambda = TAPE ? 0u : g.u32(3);
        // MonteCarloFRET.py:90: drawn also when the acceptor absorbed
        uint32_t v = g.u32(1);
        int pathway = (v >= m.pT[0]) + (v >= m.pT[1]);
        if (pathway > m.n_p - 1) pathway = m.n_p - 1;
        if (in_donor && pathway == 0) return 1;  // :92-105
        if (in_donor && pathway == 1) return 0;  // :106-109
        if (m.n_p < 3) return 0;  // no acceptor present: the reference has no acceptor pathways here
        uint32_t va = g.u32(2);  // :119
        int pa = (va >= m.pAT[0]) + (va >= m.pAT[1]);
        if (pa == 0) return 2;   // :170-175
        if (!BACK || pa == 1) return 0;  // :182-185 (without BACK no threshold of the scene lets pa reach 2)
        in_donor = true;         // :187-190 back transfer
    }
}

// np.roll(MakeVectAxis(v), 2, axis=1) applied to a column vector
// (RTUtils.py:25-42): the frame with columns n1 = unit(v x x^), n2 = unit(v x n1), v
// (y^ takes the place of x^ when v is exactly x^).
template <typename real>
__device__ __forceinline__ void to_axis_frame(real vx, real vy, real vz, real &Dx, real &Dy, real &Dz)
{
    const bool is_x = (vx == real(1)) & (vy == real(0)) & (vz == real(0));
    real ax = is_x ? -vz : real(0), ay = is_x ? real(0) : vz, az = is_x ? vx : -vy;
    normalize3(ax, ay, az);
    real bx = vy * az - vz * ay, by = vz * ax - vx * az, bz = vx * ay - vy * ax;
    normalize3(bx, by, bz);
    real x = ax * Dx + bx * Dy + vx * Dz;
    real y = ay * Dx + by * Dy + vy * Dz;
    real z = az * Dx + bz * Dy + vz * Dz;
    Dx = x; Dy = y; Dz = z;
}

// MonteCarloFRET.py:124-157: fluorescence of an aligned acceptor from sampled
// angles.  (sx, cx): polar angle of the emission about the dipole, law sin^3
// (RTUtils.py:56-100); az: its azimuth; tilt, phi: polar angle and azimuth of
// the dipole about the alignment vector (RTUtils.py:45-54); ph: azimuth of the
// emission dipole about the absorption dipole (used if the medium has an
// absorption/emission angle).  Azimuths arrive as (sin, cos) pairs.  Returns
// false if the polarisation is still to be drawn (PhotonObject.py:107-114).
template <typename real>
__device__ __forceinline__ bool aligned_emission(const DevOrient<real> &o, real sx, real cx, real s_az, real c_az,
                                                 real tilt, real s_phi, real c_phi, real s_ph, real c_ph,
                                                 Photon<real> &p)
{
    real Dx = sx * c_az, Dy = sx * s_az, Dz = cx;  // RTUtils.py:97-100
    real st, ct;
    M<real>::sincos(tilt, &st, &ct);
    const real mx = st * c_phi, my = st * s_phi, mz = ct;  // RTUtils.py:49
    real vx = o.f[0] * mx + o.f[1] * my + o.f[2] * mz;     // RTUtils.py:51-53
    real vy = o.f[3] * mx + o.f[4] * my + o.f[5] * mz;
    real vz = o.f[6] * mx + o.f[7] * my + o.f[8] * mz;
    normalize3(vx, vy, vz);
    if (o.flags & 2)  // MonteCarloFRET.py:139-149
        to_axis_frame(o.sin_ae * c_ph, o.sin_ae * s_ph, o.cos_ae, Dx, Dy, Dz);
    to_axis_frame(vx, vy, vz, Dx, Dy, Dz);  // :151-152
    normalize3(Dx, Dy, Dz);
    p.Dx = Dx; p.Dy = Dy; p.Dz = Dz;
    if (!(o.flags & 1)) return false;
    // PhotonObject.py:100-105: E = unit(D x (dipole x D))
    real ax = vy * Dz - vz * Dy, ay = vz * Dx - vx * Dz, az = vx * Dy - vy * Dx;
    real ex = Dy * az - Dz * ay, ey = Dz * ax - Dx * az, ez = Dx * ay - Dy * ax;
    normalize3(ex, ey, ez);
    p.Ex = ex; p.Ey = ey; p.Ez = ez;
    return true;
}

// PhotonObject.py:219-258 after the photon has been moved to the absorption
// point: donor/acceptor coin, pathway, re-emission, new free path -- with the
// draws in the reference's order.
template <typename real>
__device__ __forceinline__ void emit_in_reference_order(const HotScene<real> &sc, const Lds<real> &lds,
                                                        Photon<real> &p, Rng<real, true> &g)
{
    const DevMat<real> &m = row(lds.mats, p.mat);
    const DevAbs<real> &a = row(lds.abs, mul24(m.abs_slot, sc.n_wid) + p.wid);
    bool in_donor = absorbed_by_donor<true>(m, a, p, g);  // PhotonObject.py:244
    uint32_t v_lambda;
    int outcome = fret_decide<true>(m, g, in_donor, 0u, v_lambda);
    if (outcome == 2 && (m.flags & 2)) {  // MonteCarloFRET.py:124-162
        const DevOrient<real> &o = row(lds.orient, p.mat);
        real sx, cx, s_az, c_az, s_phi, c_phi, s_ph = real(0), c_ph = real(1);
        for (;;) {  // RTUtils.py:64-68 rejection sampling of sin^3 on [0, pi]
            M<real>::sincospi(g.uniform(0), &sx, &cx);
            const real uy = g.uniform(0);
            if (!(uy > sx * sx * sx) || g.overrun) break;
        }
        M<real>::sincos_turns(g.uniform(0), &s_az, &c_az);  // RTUtils.py:96
        const real tilt = o.S * g.normal1();                // RTUtils.py:47
        M<real>::sincos_turns(g.uniform(0), &s_phi, &c_phi);
        if (o.flags & 2) M<real>::sincos_turns(g.uniform(0), &s_ph, &c_ph);  // MonteCarloFRET.py:141
        if (!aligned_emission(o, sx, cx, s_az, c_az, tilt, s_phi, c_phi, s_ph, c_ph, p)) random_polarization(p, g);
        p.wid = m.widA_off + cdf_pick(lds.cdf + m.cdfA_off, m.cdfA_len, g.u32(3));
        p.status = RTMC_ST_REEMITTED_ACCEPTOR;
    } else if (outcome == 1) {  // direction, wavelength, polarisation (MonteCarloFRET.py:95-103)
        isotropic_direction(p, g);
        p.wid = m.widD_off + cdf_pick(lds.cdf + m.cdfD_off, m.cdfD_len, g.u32(3));
        p.status = RTMC_ST_REEMITTED_DONOR;
        random_polarization(p, g);
    } else if (outcome == 2) {  // direction, polarisation, wavelength (MonteCarloFRET.py:172-175)
        isotropic_direction(p, g);
        random_polarization(p, g);
        p.wid = m.widA_off + cdf_pick(lds.cdf + m.cdfA_off, m.cdfA_len, g.u32(3));
        p.status = RTMC_ST_REEMITTED_ACCEPTOR;
    } else {
        p.status = RTMC_ST_INTERNAL_CONVERTED;
    }
    // PhotonObject.py:250/258: a free path is drawn whatever happened
    p.dist = free_path<true>(sc, lds, p.mat, m.abs_slot, p, g.expo(0));
}

// ---------------------------------------------------------------------------
// launch and re-emission, Philox mode
// ---------------------------------------------------------------------------
// Uniform direction on the sphere from two uniforms.  The reference normalises
// three normals (RTUtils.py:102-108); both are exact samplers of the same law.
template <typename real>
__device__ __forceinline__ void sphere_direction(real u_z, real u_phi, real &x, real &y, real &z)
{
    z = real(1) - real(2) * u_z;
    real r2 = real(1) - z * z;
    real r = M<real>::sqrt(r2 > real(0) ? r2 : real(0));
    real s, c;
    M<real>::sincos_turns(u_phi, &s, &c);
    x = r * c; y = r * s;
}

// Uniform unit vector perpendicular to the unit vector D.  The reference
// normalises D x v for an isotropic v (PhotonObject.py:107-114), which is the
// same law.  Orthonormal basis after Duff et al. (2017).
template <typename real>
__device__ __forceinline__ void perpendicular(real Dx, real Dy, real Dz, real u_psi, real &Ex, real &Ey, real &Ez)
{
    real sg = M<real>::copysign(real(1), Dz);
    real a = real(-1) * M<real>::rcp(sg + Dz);
    real b = Dx * Dy * a;
    real s, c;
    M<real>::sincos_turns(u_psi, &s, &c);
    Ex = c * (real(1) + sg * Dx * Dx * a) + s * b;
    Ey = c * (sg * b) + s * (sg + Dy * Dy * a);
    Ez = c * (-sg * Dx) + s * (-Dy);
}

// Launch (PhotonObject.py:49-97): words 0,1 of the block direction (hemisphere
// source only), 2 polarisation angle, 3 wavelength (spectrum source only).
// Re-emission (MonteCarloFRET.py:92-105,170-175 + PhotonObject.py:250/258):
// words 0,1 direction, 2 polarisation angle, 3 free path.
// Launch or re-emission of the lanes of a wave in one pass: the direction and
// polarisation samplers are run once for both kinds of lane.
template <bool ALIGNED, bool AREA, bool SPECTRA, bool CELLS = false, typename real>
__device__ __forceinline__ void renew_photon(SceneArg<real> sc, const HotScene<real> &hot, const Lds<real> &lds,
                                             Photon<real> &p, Rng<real, false> &g, bool launching)
{
    const auto &s = sc.src;
    const bool hemisphere = s.dir_mode == 1;
    real dx = s.dir[0], dy = s.dir[1], dz = s.dir[2];
    if (!launching | hemisphere) sphere_direction(g.uniform(0), g.uniform(1), dx, dy, dz);
    p.Dx = dx; p.Dy = dy; p.Dz = dz;
    if (launching & hemisphere) apply_source_frame(s, p);
    perpendicular(p.Dx, p.Dy, p.Dz, g.uniform(2), p.Ex, p.Ey, p.Ez);
    // (The free path is stored unconditionally: an if/else that stores to a different
    // member in each arm is merged into one store through a selected address, which
    // keeps those members of the photon record in scratch memory.)
    real dist = real(0);
    if (!launching) dist = free_path<ALIGNED>(hot, lds, p.mat, row(lds.mats, p.mat).abs_slot, p, g.expo(3));
    if (launching) {
        p.Lx = s.loc[0]; p.Ly = s.loc[1]; p.Lz = s.loc[2];
        p.wid = s.wid_fixed;
        if (SPECTRA && s.wl_mode == 1)
            p.wid = cdf_pick_guided(lds.cdf + s.cdf_off, s.cdf_len, lds.guide + s.guide_off, g.u32(3), s.guide_shift);
        if (AREA && s.has_area) {  // a block of its own: words 0, 1 = position along x, y
            g.block(0u, BLK_LAUNCH + 1u);
            p.Lx += (g.uniform(0) - real(0.5)) * s.area[0];
            p.Ly += (g.uniform(1) - real(0.5)) * s.area[1];
        }
        init_photon_common(sc, p);
        if constexpr (CELLS) {   // the cell the source sits in; a strict source: the first hit, the same for every photon
            const CellHeader &h = *lds.cells;
            p.flags = h.launch_flags;
            p.best_t = h.preset_t;
            p.last_el = h.preset_e;
        }
    }
    p.dist = dist;
}

// Fluorescence of an aligned acceptor (MonteCarloFRET.py:124-162) with two
// blocks.  The first (already computed by the caller): word 0 polar angle of the
// emission about the dipole through the closed-form inverse of the sin^3
// distribution function F(c) = (2 - 3c + c^3)/4, c = cos(angle) -- the reference
// samples the same law by rejection (RTUtils.py:64-68) --, word 1 its azimuth,
// words 2,3 the normal variate of the dipole tilt (Box-Muller).  The second:
// word 0 tilt azimuth, 1 emission-dipole azimuth, 2 polarisation angle, 3 free path.
template <typename real>
__device__ __forceinline__ void renew_emit_aligned(const HotScene<real> &sc, const Lds<real> &lds,
                                                   const DevMat<real> &m, Photon<real> &p, Rng<real, false> &g,
                                                   uint32_t iter)
{
    const DevOrient<real> &o = row(lds.orient, p.mat);
    const real pi = real(3.141592653589793);
    const real cx = real(2) * M<real>::cos((M<real>::acos(real(2) * g.uniform(0) - real(1)) + real(4) * pi) * real(1.0 / 3.0));
    const real s2 = real(1) - cx * cx;
    const real sx = M<real>::sqrt(s2 > real(0) ? s2 : real(0));
    real s_az, c_az, s_g, c_g, s_phi, c_phi, s_ph, c_ph;
    M<real>::sincos_turns(g.uniform(1), &s_az, &c_az);
    const real r = M<real>::sqrt(real(-2) * M<real>::log((real(g.w[2] >> 8) + real(1)) * real(1.0 / 16777216.0)));
    M<real>::sincos_turns(g.uniform(3), &s_g, &c_g);
    const real tilt = o.S * (r * c_g);
    g.block(iter, BLK_EMIT + 1u);
    M<real>::sincos_turns(g.uniform(0), &s_phi, &c_phi);
    M<real>::sincos_turns(g.uniform(1), &s_ph, &c_ph);
    if (!aligned_emission(o, sx, cx, s_az, c_az, tilt, s_phi, c_phi, s_ph, c_ph, p))
        perpendicular(p.Dx, p.Dy, p.Dz, g.uniform(2), p.Ex, p.Ey, p.Ez);
    p.dist = free_path<true>(sc, lds, p.mat, m.abs_slot, p, g.expo(3));
}

// ---------------------------------------------------------------------------
// wave-uniform record loads
// ---------------------------------------------------------------------------
// The hit records never change while a kernel runs.  Reading them through the
// constant address space tells the compiler so, and wave-uniform reads then go
// through the scalar cache into SGPRs (s_load) even though the kernel also
// issues atomics to global memory.
template <typename T>
__device__ __forceinline__ T load_uniform(const T *ptr)
{
    static_assert(sizeof(T) % 4 == 0, "record size");
    typedef const uint32_t __attribute__((address_space(4))) *ConstWords;
    const ConstWords w = (ConstWords)(uintptr_t)ptr;
    union { T value; uint32_t words[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) u.words[i] = w[i];
    return u.value;
}

// The same with the address given as a wave-uniform base and a 32-bit byte
// offset, which is the form a scalar load takes its address in: a loop over a
// table then advances and tests one 32-bit register.
template <typename T>
__device__ __forceinline__ T load_uniform_at(const T *base, uint32_t byte_offset)
{
    static_assert(sizeof(T) % 4 == 0, "record size");
    typedef const char __attribute__((address_space(4))) *ConstBytes;
    typedef const uint32_t __attribute__((address_space(4))) *ConstWords;
    const ConstWords w = (ConstWords)((ConstBytes)(uintptr_t)base + byte_offset);
    union { T value; uint32_t words[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) u.words[i] = w[i];
    return u.value;
}

// ---------------------------------------------------------------------------
// nearest valid hit
// ---------------------------------------------------------------------------
// Geometries.py:32-52 Rectangle.FindIntersections for one element of general
// orientation; returns the line parameter t (B = L + t D) and whether B lies
// inside the rectangle.
template <typename real>
__device__ __forceinline__ bool rect_hit(const DevElem<real> &e, real Lx, real Ly, real Lz, real Dx,
                                         real Dy, real Dz, real &t)
{
    real den = Dx * e.nx + Dy * e.ny + Dz * e.nz;
    real num = e.d0 - (Lx * e.nx + Ly * e.ny + Lz * e.nz);
    t = num * M<real>::rcp(den);
    real Bx = Lx + t * Dx, By = Ly + t * Dy, Bz = Lz + t * Dz;
    real t1 = Bx * e.ax + By * e.ay + Bz * e.az - e.c1;
    real t2 = Bx * e.bx + By * e.by + Bz * e.bz - e.c2;
    return (den != real(0)) & (M<real>::abs(t1) <= e.h1) & (M<real>::abs(t2) <= e.h2);
}

// The same test for a rectangle whose normal is coordinate axis k:
// t = (p0 - L_k) / D_k = p0 * (1/D_k) + (-L_k / D_k); the caller passes the two
// per-iteration constants of its axis and the other two coordinate pairs.
// D_k == 0 makes t infinite or NaN and both comparisons false.
template <typename real>
__device__ __forceinline__ bool axis_hit(const HitRec<real> &r, real inv_dk, real m_lk, real Li, real Di,
                                         real Lj, real Dj, real &t)
{
    t = r.p0 * inv_dk + m_lk;
    real Bi = Li + t * Di, Bj = Lj + t * Dj;
    return (M<real>::abs(Bi - r.ci) <= r.hi) & (M<real>::abs(Bj - r.cj) <= r.hj);
}

// Geometries.py:68-80 Circle.FindIntersections
template <typename real>
__device__ __forceinline__ bool circle_hit(const DevElem<real> &e, real Lx, real Ly, real Lz, real Dx, real Dy,
                                           real Dz, real &t)
{
    real den = Dx * e.nx + Dy * e.ny + Dz * e.nz;
    t = (e.d0 - (Lx * e.nx + Ly * e.ny + Lz * e.nz)) * M<real>::rcp(den);
    real x = Lx + t * Dx - e.ax, y = Ly + t * Dy - e.ay, z = Lz + t * Dz - e.az;
    return (den != real(0)) & (x * x + y * y + z * z <= e.h1);
}

// Geometries.py:220-251 Triangle.FindIntersections / CheckPointInTriangle: the
// point and the corners in the in-plane frame, three edge functions of equal sign
template <typename real>
__device__ __forceinline__ bool triangle_hit(const DevElem<real> &e, real Lx, real Ly, real Lz, real Dx, real Dy,
                                             real Dz, real &t)
{
    real den = Dx * e.nx + Dy * e.ny + Dz * e.nz;
    t = (e.d0 - (Lx * e.nx + Ly * e.ny + Lz * e.nz)) * M<real>::rcp(den);
    real Bx = Lx + t * Dx, By = Ly + t * Dy, Bz = Lz + t * Dz;
    real u = Bx * e.ax + By * e.ay + Bz * e.az, v = Bx * e.bx + By * e.by + Bz * e.bz;
    // SclPrdSign(pt, q_i, q_j) = (pt.x - qj.x)(qi.y - qj.y) - (qi.x - qj.x)(pt.y - qj.y)
    bool b1 = (u - e.h1) * (e.c2 - e.h2) - (e.c1 - e.h1) * (v - e.h2) < real(0);
    bool b2 = (u - e.g1) * (e.h2 - e.g2) - (e.h1 - e.g1) * (v - e.g2) < real(0);
    bool b3 = (u - e.c1) * (e.g2 - e.c2) - (e.g1 - e.c1) * (v - e.c2) < real(0);
    return (den != real(0)) & (b1 == b2) & (b2 == b3);
}

// Geometries.py:97-136,156-204 Cylinder.FindIntersections: of the two points
// where the line meets the wall, the one the reference hands to the search.
//   fresh photon (not coming from an element): points behind it are dropped;
//   both left: the nearer one, unless one of them is the photon's own position
//   (within 1e-7), then the other; one left: that one.
// `sitting`: the photon's last operation was on this cylinder.  In float32 the
// 1e-7 test cannot tell the photon's own position; there the root of smaller
// magnitude is taken to be it when the photon sits on the wall.
template <typename real>
__device__ __forceinline__ bool cylinder_hit(const DevElem<real> &e, real Lx, real Ly, real Lz, real Dx, real Dy,
                                             real Dz, bool was_element, bool sitting, real dnorm, real &t)
{
    const real wx = Lx - e.bx, wy = Ly - e.by, wz = Lz - e.bz;
    const real da = Dx * e.ax + Dy * e.ay + Dz * e.az, wa = wx * e.ax + wy * e.ay + wz * e.az;
    const real px = Dx - da * e.ax, py = Dy - da * e.ay, pz = Dz - da * e.az;  // across the axis
    const real qx = wx - wa * e.ax, qy = wy - wa * e.ay, qz = wz - wa * e.az;
    const real a1 = px * px + py * py + pz * pz;
    const real a2 = real(2) * (px * qx + py * qy + pz * qz);
    const real a3 = qx * qx + qy * qy + qz * qz - e.c1;
    const real dsc = a2 * a2 - real(4) * a1 * a3;
    const bool any = (a1 != real(0)) & ((dsc >= real(0)) | (!M<real>::has_self_hit_eps & sitting));
    const real sq = M<real>::sqrt_rn(dsc > real(0) ? dsc : real(0));  // a grazing hit amplifies every rounding
    const real inv = M<real>::rcp(real(2) * a1);
    real t1 = (-a2 + sq) * inv, t2 = (-a2 - sq) * inv;
    if (!M<real>::has_self_hit_eps && sitting) {
        // on the wall a3 is zero up to rounding: one root is the photon's own
        // position and the other is -a2/a1 exactly, which is far better
        // conditioned in float32 for a grazing ray than the root formula
        const real other = -a2 * M<real>::rcp(a1);
        t1 = a2 < real(0) ? other : real(0);
        t2 = a2 < real(0) ? real(0) : other;
    }
    const real x1 = wa + t1 * da, x2 = wa + t2 * da;  // axial coordinates
    bool ok1 = any & !(x1 < e.h1) & !(x1 > e.h2);
    bool ok2 = any & !(x2 < e.h1) & !(x2 > e.h2);
    if (!was_element) { ok1 = ok1 & !(t1 < real(0)); ok2 = ok2 & !(t2 < real(0)); }
    const real d1 = M<real>::abs(t1) * dnorm, d2 = M<real>::abs(t2) * dnorm;
    bool own1, own2;
    if (M<real>::has_self_hit_eps) {
        own1 = d1 <= real(1e-7); own2 = d2 <= real(1e-7);
    } else {
        own1 = sitting & (d1 <= d2); own2 = sitting & (d2 < d1);
    }
    const bool both = ok1 & ok2;
    const bool take1 = both ? (!own1 & (own2 | (d1 < d2))) : ok1;
    const bool take2 = both ? (!own2 & (own1 | !(d1 < d2))) : (ok2 & !ok1);
    t = take1 ? t1 : t2;
    bool hit = take1 | take2;
    if (!M<real>::has_self_hit_eps) hit = hit & !(take1 ? own1 : own2);  // RayTrace.py:134 rejects the own position
    return hit;
}

// Geometries.py:138-153,207-209: unit normal of a cylinder's tangent plane at
// P, pointing from P to the axis.
template <typename real>
__device__ __forceinline__ void cylinder_normal(const DevElem<real> &e, real Px, real Py, real Pz, real &nx, real &ny,
                                                real &nz)
{
    const real wx = Px - e.bx, wy = Py - e.by, wz = Pz - e.bz;
    const real wa = wx * e.ax + wy * e.ay + wz * e.az;
    nx = wa * e.ax - wx; ny = wa * e.ay - wy; nz = wa * e.az - wz;
    normalize3(nx, ny, nz);
}

// FindIntersections of an element of any shape.
template <typename real>
__device__ __forceinline__ bool shape_hit(const DevElem<real> &e, real Lx, real Ly, real Lz, real Dx, real Dy,
                                          real Dz, bool was_element, bool sitting, real dnorm, real &t)
{
    const int geom = e.type >> 8;
    if (geom == RTMC_GEOM_CIRCLE) return circle_hit(e, Lx, Ly, Lz, Dx, Dy, Dz, t);
    if (geom == RTMC_GEOM_TRIANGLE) return triangle_hit(e, Lx, Ly, Lz, Dx, Dy, Dz, t);
    if (geom == RTMC_GEOM_CYLINDER) return cylinder_hit(e, Lx, Ly, Lz, Dx, Dy, Dz, was_element, sitting, dnorm, t);
    return rect_hit(e, Lx, Ly, Lz, Dx, Dy, Dz, t);
}

// Running minimum of the nearest-hit search.
template <typename real>
struct Nearest {
    real t;    // signed line parameter of the best candidate, +inf if none
    int32_t e; // its element index, -1 if none
};

// RayTrace.py:134: a candidate replaces the best so far only if it is strictly
// nearer, so of several elements at exactly the same distance the first one of
// the scene's list wins.  The search does not visit the elements in list order
// (it goes group by group), so an equal distance is decided by the index.
// `guard`: candidates nearer than this are the photon's own position
// (RayTrace.py:134: 1e-8 in float64; float32 uses a multiple of its resolution
// at the scene's size).
template <typename real>
__device__ __forceinline__ void consider(Nearest<real> &best, bool inside, real t, int index, real adm_q,
                                         bool adm_all, int skip, real dnorm, real guard)
{
    const real a = M<real>::abs(t), b = M<real>::abs(best.t);
    const bool nearer = (a < b) | ((a == b) & (index < best.e));
    const bool ok = inside & ((t * adm_q > real(0)) | adm_all) & (index != skip) & nearer & (a * dnorm > guard);
    best.t = ok ? t : best.t;
    best.e = ok ? index : best.e;
}

// RayTrace.py:134: nothing nearer than 1e-8 cm is a candidate -- that is the
// photon's own position.  float64 takes the number as it stands.  float32
// cannot (1e-8 cm is far below its resolution at these coordinates) and rejects
// the element the photon sits on by index instead.  That leaves one case open:
// a *different* element whose plane passes through the photon's position -- a
// detector lying flush on the interface just crossed, or the neighbour across a
// seam the photon has landed on: its line parameter is the rounding residue of
// the last move, of either sign.  Scenes in which photons can get there (the
// host finds coplanar overlapping pairs, and sources aimed at a seam) run
// kernels with GUARD: a photon that sits on an element then accepts no plane
// within the uncertainty of its own position, measured perpendicular to the
// plane for rectangles.  The guard is not free of side effects -- it also hides
// a genuine next element that close, where float64 still tells the two apart:
// applied to every search of every scene it would send 3e-5 to 1e-4 of the
// photons of the stacked-slab and rod scenes elsewhere than float64 does (3e-6
// to 2e-5 without it), a bias that 1e10 photons resolve -- so scenes without
// such elements do not use it.  A strict source aimed at a seam uses it until a
// photon's first absorption: up to there every photon of such a source visits
// the same landing points (only the reflect / cross decisions differ), so the
// guard acts where one of these points lies on a seam and nowhere else.
//
// landing_uncertainty: how far the float32 landing point L' = L + t D, t = -off / (D.n), may lie
// from the true one, along the old direction.  off = L.n - d0 carries two roundings at the size of
// the coordinates, the reciprocal and the product one each at the size of the path, the sum one
// more: at most 2^-24 (3 |coordinates| + 2 path) |D| / |D.n|; 4 units of (|coordinates| + path)
// cover it.
#ifndef RTMC_GUARD_UNITS
#define RTMC_GUARD_UNITS 4.0
#endif
template <typename real>
__device__ __forceinline__ real landing_uncertainty(const Photon<real> &p, real path, real dnorm, real dn)
{
    if constexpr (M<real>::has_self_hit_eps) {
        return real(0);
    } else {
        const float reach = __builtin_fmaxf(__builtin_fmaxf(M<real>::abs(p.Lx), M<real>::abs(p.Ly)), M<real>::abs(p.Lz)) + path;
        return reach * (float(RTMC_GUARD_UNITS / 16777216.0) * dnorm) * M<real>::abs(M<real>::rcp(dn));
    }
}

template <bool GUARD, typename real>
__device__ __forceinline__ real guard_distance(const Photon<real> &p, int scene_mode)
{
    if constexpr (M<real>::has_self_hit_eps) {
        return real(1e-8);
    } else if constexpr (GUARD) {  // kernels that hold the code; the scene says whether it applies
        const bool applies = (scene_mode == 1) | ((scene_mode == 2) & ((p.flags & F_ABS_ONCE) == 0u));
        return applies ? p.guard : real(0);
    } else {
        return real(0);
    }
}

// Folds the result of one group into the overall best, with the same rule.
template <typename real>
__device__ __forceinline__ void merge_nearest(Nearest<real> &best, const Nearest<real> &g)
{
    const real a = M<real>::abs(g.t), b = M<real>::abs(best.t);
    const bool take = (a < b) | ((a == b) & (g.e < best.e));  // "none" is (inf, -1) and never wins a tie
    best.t = take ? g.t : best.t;
    best.e = take ? g.e : best.e;
}

// The float32 test of the two rectangles of one chunk: in-plane offsets di, dj
// against half extents hi, hj.  The five conditions narrow the execution mask
// one compare after the other and the two moves then update the running
// minimum under the narrowed mask: no mask algebra on the scalar unit, which
// this loop otherwise keeps as busy as the vector unit.  `ta` carries the
// admissibility test as a sign: t * adm_q + bias, with bias > 0 when every
// candidate is admissible and otherwise minus the guard distance in the units of
// that product, so that a plane the photon lies in (within rounding) offers no
// candidate.  A line parallel to the plane gives t = inf or NaN, which fails the
// comparisons.  Within a group the records are in list order and the compare is
// strict: the first of equal candidates stays.
// The block saves the execution mask itself, restores it after each record and
// is volatile: wherever the compiler places it, it leaves the mask as it found it.
__device__ __forceinline__ void consider_pair_f32(Nearest<float> &best, const HitChunk<float> &ch, float inv_dk,
                                                  float m_lk, float Li, float Di, float Lj, float Dj, float adm_q,
                                                  float adm_bias, int skip)
{
    const float t0 = ch.r[0].p0 * inv_dk + m_lk, t1 = ch.r[1].p0 * inv_dk + m_lk;
    const float di0 = (Li + t0 * Di) - ch.r[0].ci, dj0 = (Lj + t0 * Dj) - ch.r[0].cj;
    const float di1 = (Li + t1 * Di) - ch.r[1].ci, dj1 = (Lj + t1 * Dj) - ch.r[1].cj;
    const float ta0 = t0 * adm_q + adm_bias, ta1 = t1 * adm_q + adm_bias;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[di0]|, %[hi0]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[dj0]|, %[hj0]\n\t"
                 "v_cmpx_lt_f32_e32 vcc, 0, %[ta0]\n\t"
                 "v_cmpx_ne_u32_e32 vcc, %[idx0], %[skip]\n\t"
                 "v_cmpx_lt_f32_e64 vcc, |%[t0]|, |%[bt]|\n\t"
                 "v_mov_b32_e32 %[bt], %[t0]\n\t"
                 "v_mov_b32_e32 %[be], %[idx0]\n\t"
                 "s_mov_b64 exec, %[sv]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[di1]|, %[hi1]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[dj1]|, %[hj1]\n\t"
                 "v_cmpx_lt_f32_e32 vcc, 0, %[ta1]\n\t"
                 "v_cmpx_ne_u32_e32 vcc, %[idx1], %[skip]\n\t"
                 "v_cmpx_lt_f32_e64 vcc, |%[t1]|, |%[bt]|\n\t"
                 "v_mov_b32_e32 %[bt], %[t1]\n\t"
                 "v_mov_b32_e32 %[be], %[idx1]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [bt] "+v"(best.t), [be] "+v"(best.e), [sv] "=&s"(saved)
                 : [di0] "v"(di0), [dj0] "v"(dj0), [hi0] "s"(ch.r[0].hi), [hj0] "s"(ch.r[0].hj), [ta0] "v"(ta0),
                   [idx0] "s"(ch.r[0].index), [t0] "v"(t0), [di1] "v"(di1), [dj1] "v"(dj1), [hi1] "s"(ch.r[1].hi),
                   [hj1] "s"(ch.r[1].hj), [ta1] "v"(ta1), [idx1] "s"(ch.r[1].index), [t1] "v"(t1), [skip] "v"(skip)
                 : "vcc");
}

// The runs of one axis group (float32): one plane intersection per run, then the members named by
// the grid cell the landing point falls into, gathered per lane from LDS and tried in list order
// (strict "nearer": the first member that contains the point keeps it).  Candidates of different
// planes of one group cannot tie -- their line parameters differ -- unless both directions along
// the line are admissible and the photon sits exactly midway (`ties_possible`: some lane of the
// wave accepts every candidate); only then is every run folded into the group's result with the
// index rule.
__device__ __forceinline__ void scan_runs_f32(const HitChunk<float> *table, const Lds<float> &lds, uint32_t &offset, int n_runs,
                                              float inv_dk, float m_lk, float Li, float Di, float Lj, float Dj, float adm_q,
                                              float adm_bias, int skip, bool ties_possible, Nearest<float> &g)
{
    for (; n_runs != 0; n_runs--) {
        const RunHead<float> h = load_uniform_at(reinterpret_cast<const RunHead<float> *>(table), offset);
        offset += (uint32_t)sizeof(RunHdr<float>);
        const float t = h.p0 * inv_dk + m_lk;
        const float Bi = Li + t * Di, Bj = Lj + t * Dj;
        // cell of the landing point (a point outside the grid is tried against an edge cell's members and fails)
        const int ci = __builtin_amdgcn_fmed3f(__builtin_floorf((Bi - h.gi0) * h.inv_wi), 0.0f, (float)h.ni_1);
        const int cj = __builtin_amdgcn_fmed3f(__builtin_floorf((Bj - h.gj0) * h.inv_wj), 0.0f, (float)h.nj_1);
        // (NaN coordinates -- a line parallel to the plane -- give cell 0 and fail the tests below)
        const uint32_t *cell = lds.run_cells + 2 * (h.cell_off + mul24(cj, h.ni_1 + 1) + ci);
        const bool candidate = (t * adm_q + adm_bias > 0.0f) & (ties_possible ? true : M<float>::abs(t) < M<float>::abs(g.t));
        Nearest<float> r;
        r.t = ties_possible ? M<float>::inf() : g.t;
        r.e = ties_possible ? -1 : g.e;
#pragma unroll
        for (int half = 0; half < 2; half++) {   // (the second word is empty in all but crowded cells)
            uint32_t word = candidate ? cell[half] : 0xFFFFFFFFu;
            while (__ballot((word & 0xFFu) != 0xFFu) != 0ull) {
                const uint32_t slot = word & 0xFFu;
                word = (word >> 8) | 0xFF000000u;
                if (slot != 0xFFu) {
                    const RunMember<float> &m = row(lds.run_members, h.mem_off + (int)slot);
                    const bool ok = (M<float>::abs(Bi - m.ci) <= m.hi) & (M<float>::abs(Bj - m.cj) <= m.hj) & (m.index != skip) &
                                    (M<float>::abs(t) < M<float>::abs(r.t));
                    r.t = ok ? t : r.t;
                    r.e = ok ? m.index : r.e;
                }
            }
        }
        if (ties_possible) merge_nearest(g, r);
        else g = r;
    }
}

// One group of rectangles normal to axis k of the current frame, one chunk of
// records per scalar fetch; the group's nearest candidate is then folded into
// the overall best.
template <bool RUNS, typename real>
__device__ __forceinline__ void scan_axis_group(const HitChunk<real> *table, const Lds<real> &lds, uint32_t &offset, int packed_count, real Lk,
                                                real Dk, real Li, real Di, real Lj, real Dj, real adm_q,
                                                bool adm_all, real guard_q, int skip, real dnorm, real guard,
                                                bool ties_possible, Nearest<real> &best, bool first = false)
{
    if (packed_count == 0) return;
    const int count = RUNS ? (packed_count & 0xFFFF) : packed_count;
    const real inv_dk = M<real>::rcp(Dk), m_lk = -Lk * inv_dk;
    // adm_q is zero when adm_all is set; guard_q = guard * |adm_q| and a plane at
    // line parameter t lies |t D_k| away from the photon
    // (without a guard guard_q is the constant 0 and the product is not computed)
    [[maybe_unused]] real adm_bias = adm_all ? real(1) : real(0);
    if (__builtin_constant_p(guard_q) ? guard_q != real(0) : true)
        adm_bias = adm_all ? real(1) : -(guard_q * M<real>::abs(inv_dk));
#ifndef RTMC_TIE_MODE
#define RTMC_TIE_MODE 2
#endif
    // (the first group of a search starts from "none" anyway and scans straight into the result)
    const bool direct = RTMC_TIE_MODE == 0 || (RTMC_TIE_MODE == 2 && first);
    Nearest<real> g;
    g.t = direct ? best.t : M<real>::inf();
    g.e = direct ? best.e : -1;
    const uint32_t end = offset + (uint32_t)count * (uint32_t)sizeof(HitChunk<real>);
    for (; offset != end; offset += (uint32_t)sizeof(HitChunk<real>)) {
        const HitChunk<real> ch = load_uniform_at(table, offset);
        if constexpr (sizeof(real) == 4 && !M<real>::has_self_hit_eps) {
            consider_pair_f32(g, ch, inv_dk, m_lk, Li, Di, Lj, Dj, adm_q, adm_bias, skip);
        } else {
#pragma unroll
            for (int k = 0; k < HIT_CHUNK; k++) {
                real t;
                const bool inside = axis_hit(ch.r[k], inv_dk, m_lk, Li, Di, Lj, Dj, t);
                consider(g, inside, t, ch.r[k].index, adm_q, adm_all, skip, dnorm, guard);
            }
        }
    }
    if constexpr (RUNS && sizeof(real) == 4 && !M<real>::has_self_hit_eps)
        scan_runs_f32(table, lds, offset, packed_count >> 16, inv_dk, m_lk, Li, Di, Lj, Dj, adm_q, adm_bias, skip, ties_possible, g);
    if (direct) best = g;
    else merge_nearest(best, g);
}

// The three axis groups of one frame; (L, D) is the line in that frame.
template <bool RUNS, typename real>
__device__ __forceinline__ void scan_frame(const HitChunk<real> *table, const Lds<real> &lds, uint32_t &offset, int n0, int n1, int n2,
                                           real Lx, real Ly, real Lz, real Dx, real Dy, real Dz, real adm_q,
                                           bool adm_all, real guard_q, int skip, real dnorm, real guard,
                                           bool ties_possible, Nearest<real> &best, bool first = false)
{
    scan_axis_group<RUNS>(table, lds, offset, n0, Lx, Dx, Ly, Dy, Lz, Dz, adm_q, adm_all, guard_q, skip, dnorm, guard, ties_possible, best, first);
    scan_axis_group<RUNS>(table, lds, offset, n1, Ly, Dy, Lz, Dz, Lx, Dx, adm_q, adm_all, guard_q, skip, dnorm, guard, ties_possible, best);
    scan_axis_group<RUNS>(table, lds, offset, n2, Lz, Dz, Lx, Dx, Ly, Dy, adm_q, adm_all, guard_q, skip, dnorm, guard, ties_possible, best);
}

// RayTrace.py:120-154 CheckElementsForCrossingPoint: the nearest element whose
// point the photon may go to next.  Candidates are ordered by |t| (the line
// parameter); the distance is |t| |D|; equal candidates: see `consider`.
// Admissibility: see set_admissibility.  The element the photon sits on is
// skipped by index, and no element offers a candidate within the guard distance
// of the photon (the reference's `distance > 1e-8`, RayTrace.py:134; in float32
// the guard is a multiple of the coordinate resolution and is measured
// perpendicular to the plane for rectangles).  `dnorm` returns |D|.
//
// Table order: the rectangles of the world frame grouped by normal axis x | y | z,
// then for every further frame a header (rotation, group sizes) and its three
// groups in the frame's coordinates, then the elements tested on their full record.
template <int FEATURES, typename real>
__device__ __forceinline__ Nearest<real> nearest_valid_hit(const HotScene<real> &sc, const Lds<real> &lds, const Photon<real> &p, real &dnorm)
{
    constexpr bool SHAPES = (FEATURES & FEAT_SHAPES) != 0;
    dnorm = M<real>::sqrt(p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz);
    const int skip = (p.flags & F_WAS_EL) ? p.last_el : -1;
    const bool adm_all = (p.flags & F_ADM_ALL) != 0u;
    const real guard = guard_distance<(FEATURES & FEAT_GUARD) != 0>(p, sc.use_guard);
    const real guard_q = (M<real>::has_self_hit_eps || (FEATURES & FEAT_GUARD)) ? guard * M<real>::abs(p.adm_q) : real(0);
    Nearest<real> nearest;
    nearest.t = M<real>::inf();
    nearest.e = -1;
    uint32_t offset = 0;  // bytes into the hit table
    constexpr bool RUNS = (FEATURES & FEAT_RUNS) != 0;
    const bool ties_possible = RUNS ? __ballot(adm_all) != 0ull : false;
    scan_frame<RUNS>(sc.hits, lds, offset, sc.n_group[0], sc.n_group[1], sc.n_group[2], p.Lx, p.Ly, p.Lz, p.Dx, p.Dy, p.Dz,
                     p.adm_q, adm_all, guard_q, skip, dnorm, guard, ties_possible, nearest, true);
    const int n_general = (FEATURES & FEAT_FRAMES) ? (sc.n_group[3] & 0xFFFF) : sc.n_group[3];
    for (int frames = (FEATURES & FEAT_FRAMES) ? (sc.n_group[3] >> 16) : 0; frames != 0; frames--) {
        const FrameHdr<real> f = load_uniform_at(reinterpret_cast<const FrameHdr<real> *>(sc.hits), offset);
        offset += (uint32_t)sizeof(FrameHdr<real>);
        const real Lx = f.R[0] * p.Lx + f.R[1] * p.Ly + f.R[2] * p.Lz, Ly = f.R[3] * p.Lx + f.R[4] * p.Ly + f.R[5] * p.Lz,
                   Lz = f.R[6] * p.Lx + f.R[7] * p.Ly + f.R[8] * p.Lz;
        const real Dx = f.R[0] * p.Dx + f.R[1] * p.Dy + f.R[2] * p.Dz, Dy = f.R[3] * p.Dx + f.R[4] * p.Dy + f.R[5] * p.Dz,
                   Dz = f.R[6] * p.Dx + f.R[7] * p.Dy + f.R[8] * p.Dz;
        scan_frame<RUNS>(sc.hits, lds, offset, f.n[0], f.n[1], f.n[2], Lx, Ly, Lz, Dx, Dy, Dz, p.adm_q, adm_all, guard_q, skip,
                         dnorm, guard, ties_possible, nearest);
    }
    const uint32_t end = offset + (uint32_t)n_general * (uint32_t)sizeof(HitChunk<real>);
    for (; offset != end; offset += (uint32_t)sizeof(HitChunk<real>)) {
        const HitChunk<real> ch = load_uniform_at(sc.hits, offset);
#pragma unroll
        for (int k = 0; k < HIT_CHUNK; k++) {
            const int index = ch.r[k].index;
            if (index < 0) continue;  // padding
            real t;
            const DevElem<real> e = load_uniform(sc.elems + index);
            if (SHAPES) {
                const bool wall = (e.type >> 8) == RTMC_GEOM_CYLINDER;  // may be met again right away
                const bool inside = shape_hit(e, p.Lx, p.Ly, p.Lz, p.Dx, p.Dy, p.Dz, (p.flags & F_WAS_EL) != 0u,
                                              index == skip, dnorm, t);
                consider(nearest, inside, t, index, p.adm_q, adm_all, wall ? -1 : skip, dnorm, guard);
            } else {  // a rectangle whose in-plane axes are not perpendicular
                const bool inside = rect_hit(e, p.Lx, p.Ly, p.Lz, p.Dx, p.Dy, p.Dz, t);
                consider(nearest, inside, t, index, p.adm_q, adm_all, skip, dnorm, guard);
            }
        }
    }
    return nearest;
}

// The search of a photon that is confined to a cell (kernels with FEAT_CELLS).  Inside a closed
// axis-aligned box the line leaves through the nearest of the three faces the photon moves towards;
// the faces behind it lie at the other side of its line parameter and every other element of the
// scene lies beyond the box (the host only marks cells whose box nothing else touches), so that
// face is what the walk over the element list (nearest_valid_hit) would return, with the same line
// parameter: the axis test's p0 * (1/D_k) - L_k/D_k.  "Towards": candidates lie at t adm_q > 0 --
// ahead of a photon that was emitted or refracted, *behind* one that was reflected (the
// reference's reflection returns the negative of the mirror direction, PhotonObject.py:150, and
// its side test then sends the photon backwards along its line).  A lane answers from its cell if
//   its cell record is usable (DevCell) and the photon is in the box, within a margin;
//   the nearest face lies on the admissible side (also rules out adm_q = 0: a re-execution in
//   place, "every candidate"), is not the element the photon sits on, is not tied with another
//   face (equal distances go to the lower list index: left to the list walk) and the landing point
//   is in the box, within the margin.
// Anything else -- a photon in a cell that holds other elements, one that leaked through a seam --
// gets element CONFINED_NONE and waits for the list walk.  Where the two can differ is what the list walk
// itself leaves to rounding: a landing point within an ulp of a face's edge, and a face *behind*
// the photon whose line parameter comes out as a rounding residue of the admissible sign.
// The conditions narrow the execution mask one compare after the other (see consider_pair_f32).
// TILES: a face may be a tiling (a read-out cut into strips); the rectangle that holds the landing
// point is then looked up through the face's grid and tried exactly as scan_runs_f32 tries the members
// of a run: the landing point L + t D against the rectangle's own centre and half extents, list
// order, the first that contains it.
// FRAMES: a cell may lie in a turned frame; the line is then taken into that frame exactly as the list
// walk does it (nearest_valid_hit), one rotation per search.
template <bool TILES, bool FRAMES>
__device__ __forceinline__ void confined_hit(const Lds<float> &lds, const Photon<float> &p, float &dnorm, Nearest<float> &out)
{
    dnorm = M<float>::sqrt(p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz);
    // the whole record at once (five 16-byte reads), then selects: left to itself the compiler selects the
    // address and reads each plane and index on its own, one LDS round trip after the other
    typedef float __attribute__((ext_vector_type(4))) Quad;
    const Quad *rec = reinterpret_cast<const Quad *>(&row(reinterpret_cast<const DevCell *>(lds.cells + 1),
                                                           (int)((p.flags >> F_CELL_RECORD_SHIFT) & F_CELL_RECORD_MASK)));
    Quad q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4));
    const float plo[3] = {q0.x, q0.y, q0.z}, phi[3] = {q0.w, q1.x, q1.y}, cc[3] = {q1.z, q1.w, q2.x}, hw[3] = {q2.y, q2.z, q2.w};
    // (the element indices travel as bit patterns; each component is copied out before it is reinterpreted)
    const float fl[3] = {q3.x, q3.y, q3.z}, fh[3] = {q3.w, q4.x, q4.y};
    const int idlo[3] = {__float_as_int(fl[0]), __float_as_int(fl[1]), __float_as_int(fl[2])};
    const int idhi[3] = {__float_as_int(fh[0]), __float_as_int(fh[1]), __float_as_int(fh[2])};
    float L[3] = {p.Lx, p.Ly, p.Lz}, D[3] = {p.Dx, p.Dy, p.Dz};
    if constexpr (FRAMES) {
        const int frame = __float_as_int(q4.z);
        if (__ballot(frame != 0) != 0ull) {
            const Quad *rows = reinterpret_cast<const Quad *>(&row(reinterpret_cast<const Quad *>(lds.cell_frames), 3 * frame));
            const Quad r0 = rows[0], r1 = rows[1], r2 = rows[2];
            const float R[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
            const float Lx = R[0] * p.Lx + R[1] * p.Ly + R[2] * p.Lz, Ly = R[3] * p.Lx + R[4] * p.Ly + R[5] * p.Lz,
                        Lz = R[6] * p.Lx + R[7] * p.Ly + R[8] * p.Lz;
            const float Dx = R[0] * p.Dx + R[1] * p.Dy + R[2] * p.Dz, Dy = R[3] * p.Dx + R[4] * p.Dy + R[5] * p.Dz,
                        Dz = R[6] * p.Dx + R[7] * p.Dy + R[8] * p.Dz;
            const bool turned = frame != 0;
            L[0] = turned ? Lx : L[0]; L[1] = turned ? Ly : L[1]; L[2] = turned ? Lz : L[2];
            D[0] = turned ? Dx : D[0]; D[1] = turned ? Dy : D[1]; D[2] = turned ? Dz : D[2];
        }
    }
    float t[3], a[3], dL[3], dB[3];
    int id[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float inv = M<float>::rcp(D[k]), m = -L[k] * inv;
        const bool up = D[k] * p.adm_q > 0.0f;   // the photon moves towards the high face
        t[k] = (up ? phi[k] : plo[k]) * inv + m;
        id[k] = up ? idhi[k] : idlo[k];
        a[k] = D[k] == 0.0f ? M<float>::inf() : M<float>::abs(t[k]);   // parallel to this pair of faces: it meets neither
        dL[k] = L[k] - cc[k];
    }
    float ab;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(ab) : "v"(a[0]), "v"(a[1]), "v"(a[2]));
    const float am = __builtin_amdgcn_fmed3f(a[0], a[1], a[2]);
    float tb = a[1] == ab ? t[1] : t[2];
    int ib = a[1] == ab ? id[1] : id[2];
    tb = a[0] == ab ? t[0] : tb;
    ib = a[0] == ab ? id[0] : ib;
#pragma unroll
    for (int k = 0; k < 3; k++) dB[k] = tb * D[k] + dL[k];
    const float ta = tb * p.adm_q;
    const int skip = (p.flags & F_WAS_EL) ? p.last_el : -1;
    int found = CONFINED_NONE;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[lx]|, %[hx]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[ly]|, %[hy]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[lz]|, %[hz]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[bx]|, %[hx]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[by]|, %[hy]\n\t"
                 "v_cmpx_le_f32_e64 vcc, |%[bz]|, %[hz]\n\t"
                 "v_cmpx_lt_f32_e32 vcc, 0, %[ta]\n\t"
                 "v_cmpx_neq_f32_e32 vcc, %[ab], %[am]\n\t"
                 "v_cmpx_ne_u32_e32 vcc, %[ib], %[skip]\n\t"
                 "v_mov_b32_e32 %[e], %[ib]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [e] "+v"(found), [sv] "=&s"(saved)
                 : [lx] "v"(dL[0]), [ly] "v"(dL[1]), [lz] "v"(dL[2]), [bx] "v"(dB[0]), [by] "v"(dB[1]), [bz] "v"(dB[2]),
                   [hx] "v"(hw[0]), [hy] "v"(hw[1]), [hz] "v"(hw[2]), [ta] "v"(ta), [ab] "v"(ab), [am] "v"(am),
                   [ib] "v"(ib), [skip] "v"(skip)
                 : "vcc");
    if constexpr (TILES) {
        const bool tiled = (found < -1) & (found != CONFINED_NONE);
        if (__ballot(tiled) != 0ull) {
            // the landing point in the in-plane coordinates of the face: axes k + 1, k + 2 of its normal axis k
            const bool x = a[0] == ab, y = a[1] == ab;
            const float Li = x ? L[1] : (y ? L[2] : L[0]), Di = x ? D[1] : (y ? D[2] : D[0]);
            const float Lj = x ? L[2] : (y ? L[0] : L[1]), Dj = x ? D[2] : (y ? D[0] : D[1]);
            const float Bi = Li + tb * Di, Bj = Lj + tb * Dj;
            const TileGrid &g = row(lds.tile_grids, tiled ? -found - 2 : 0);
            const int ci = __builtin_amdgcn_fmed3f(__builtin_floorf((Bi - g.gi0) * g.inv_wi), 0.0f, (float)g.ni_1);
            const int cj = __builtin_amdgcn_fmed3f(__builtin_floorf((Bj - g.gj0) * g.inv_wj), 0.0f, (float)g.nj_1);
            const uint32_t *cell = lds.run_cells + 2 * (g.cell_off + mul24(cj, g.ni_1 + 1) + ci);
            int holder = CONFINED_NONE;
#pragma unroll
            for (int half = 0; half < 2; half++) {   // (the second word is empty in all but crowded cells)
                uint32_t word = tiled ? cell[half] : 0xFFFFFFFFu;
                while (__ballot((word & 0xFFu) != 0xFFu) != 0ull) {
                    const uint32_t slot = word & 0xFFu;
                    word = (word >> 8) | 0xFF000000u;
                    if (slot != 0xFFu) {
                        const RunMember<float> &m = row(lds.run_members, g.mem_off + (int)slot);
                        const bool ok = (M<float>::abs(Bi - m.ci) <= m.hi) & (M<float>::abs(Bj - m.cj) <= m.hj) & (m.index != skip) &
                                        (holder == CONFINED_NONE);
                        holder = ok ? m.index : holder;
                    }
                }
            }
            found = tiled ? holder : found;
        }
    }
    const bool preset = (p.flags & F_PRESET) != 0u;   // launched from a strict source: the renew block left the hit on the photon
    out.t = preset ? p.best_t : tb;
    out.e = preset ? p.last_el : found;
}

// The direction updates of PhotonObject.py:150 / :199 and the dot product D.n they and the
// interface start from.  float32 lets the compiler fuse multiplies and adds.  float64 rounds
// every operation on its own, and sums a dot product the way numpy does, as the reference's
// arithmetic and the CPU checker do: whether the cosine of an exactly normal incidence on a
// tilted plane comes out as 1 or as 1 + 2^-52 decides, through the reference's unnormalised
// arccos (Elements.py:87), between a 4 % reflectance and a NaN, so the last bit matters here.
template <typename real>
__device__ __forceinline__ real dot_as_reference(real ax, real ay, real az, real bx, real by, real bz)
{
    if constexpr (M<real>::has_self_hit_eps) {
#pragma clang fp contract(off)
        return __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx));
    } else {
        return ax * bx + ay * by + az * bz;
    }
}

template <typename real>
__device__ __forceinline__ void cross_as_reference(real ax, real ay, real az, real bx, real by, real bz, real &cx,
                                                   real &cy, real &cz)
{
    // numpy's cross: every product rounded on its own (no fused multiply-subtract)
#pragma clang fp contract(off)
    cx = ay * bz - az * by; cy = az * bx - ax * bz; cz = ax * by - ay * bx;
}

template <typename real>
__device__ __forceinline__ void refract_direction(Photon<real> &p, real r, real cc, real sgn, real nx, real ny, real nz)
{
    if constexpr (M<real>::has_self_hit_eps) {
#pragma clang fp contract(off)
        const real k = r * cc - sgn * M<real>::sqrt(real(1) - (r * r) * (real(1) - cc * cc));
        p.Dx = r * p.Dx + k * nx; p.Dy = r * p.Dy + k * ny; p.Dz = r * p.Dz + k * nz;
    } else {
        const real k = r * cc - sgn * M<real>::sqrt(real(1) - (r * r) * (real(1) - cc * cc));
        p.Dx = r * p.Dx + k * nx; p.Dy = r * p.Dy + k * ny; p.Dz = r * p.Dz + k * nz;
    }
}

template <typename real>
__device__ __forceinline__ void reflect_direction(Photon<real> &p, real dn, real nx, real ny, real nz)
{
    if constexpr (M<real>::has_self_hit_eps) {
#pragma clang fp contract(off)
        const real c2 = real(2) * dn;
        p.Dx = c2 * nx - p.Dx; p.Dy = c2 * ny - p.Dy; p.Dz = c2 * nz - p.Dz;
    } else {
        const real c2 = real(2) * dn;
        p.Dx = c2 * nx - p.Dx; p.Dy = c2 * ny - p.Dy; p.Dz = c2 * nz - p.Dz;
    }
}

// PhotonObject.py:149/198 (Elements.py:167 for the detector): the photon moves along its
// line onto the plane; off = (L - S).n, dn = D.n
template <typename real>
__device__ __forceinline__ void move_onto_plane(Photon<real> &p, real off, real dn)
{
    const real t = -off * M<real>::rcp(dn);
    p.Lx += t * p.Dx; p.Ly += t * p.Dy; p.Lz += t * p.Dz;
}

// Elements.py:125-129, PhotonObject.py:158-191 FocusThinLens, for a photon that has been
// moved onto the lens plane (off: its signed distance before the move; sz: z of the lens
// centre).  Also returns D_new x D_old in (cx, cy, cz).
template <typename real>
__device__ __forceinline__ void focus_thin_lens(Photon<real> &p, real f, real off, real dn, real nz, real sz,
                                                real &cx, real &cy, real &cz)
{
    // the focal plane on the far side: S + fo n, fo = -f if the photon came from the side of S + f n
    const real fo = (sgn_i(off) == sgn_i(f)) ? -f : f;
    // parallel ray through (x, y of the hit point, z of the lens centre), PhotonObject.py:178
    const real gz = sz - p.Lz;
    const real t2 = (fo - nz * gz) * M<real>::rcp(dn);  // (FP - PR0).n / (D.n); the hit point is on the plane
    real vx = -(t2 * p.Dx), vy = -(t2 * p.Dy), vz = -(gz + t2 * p.Dz);
    const real iv = M<real>::rsqrt(vx * vx + vy * vy + vz * vz);
    // D_new x D_old = -((0,0,gz) x D_old) / |v|: the part of v along D_old drops out
    // (the two directions are nearly antiparallel, their cross product taken
    // numerically would lose most of its digits in float32)
    cx = gz * p.Dy * iv; cy = -gz * p.Dx * iv; cz = real(0);
    p.Dx = vx * iv; p.Dy = vy * iv; p.Dz = vz * iv;
}

// Elements.py:75-109 IsTotalReflect and Reflectance of an interface n1 -> n2 for a photon on
// the plane with normal n (dn = D.n, dnorm = |D|, sgn = sign(-dn), r = n1/n2).  Returns
// whether the photon is totally reflected; otherwise R is the reflectance for its polarisation.
template <typename real>
__device__ __forceinline__ bool interface_reflectance(Photon<real> &p, real n1, real n2, real r, real dn, real dnorm,
                                                      real sgn, real nx, real ny, real nz, real &R)
{
    const real c = M<real>::abs(dn);  // cos(theta_i) for unit D, as Elements.py:87 takes it
    bool tir;
    real R_perp, R_par;
    if constexpr (M<real>::has_self_hit_eps) {
        // float64 follows the reference's angles literally: Elements.py:77 normalises the
        // cosine, Elements.py:87 does not, so a cosine that rounding has lifted above 1
        // (normal incidence on a tilted plane) makes theta_i, the reflectance and the
        // comparison `rand() > R` NaN / false: the photon is reflected, here as there
        const real nn = M<real>::sqrt(nx * nx + ny * ny + nz * nz);
        const real theta_n = M<real>::acos(c / (dnorm * nn));
        tir = (n1 > n2) & (M<real>::abs(theta_n) > M<real>::asin(n2 / n1));
        const real theta_i = M<real>::acos(c);
        const real theta_t = M<real>::asin((n1 / n2) * M<real>::sin(theta_i));
        const real ci = M<real>::cos(theta_i), ct = M<real>::cos(theta_t);
        const real rs = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct);
        const real rp = (n2 * ci - n1 * ct) / (n2 * ci + n1 * ct);
        R_perp = rs * rs; R_par = rp * rp;
    } else {
        // Elements.py:75-81: theta_i > arcsin(n2/n1)  <=>  cos^2 < 1 - (n2/n1)^2, with the normalised cosine
        const real cn = c * M<real>::rcp(dnorm);
        tir = (n1 > n2) & (cn * cn < real(1) - (n2 * n2) * M<real>::rcp(n1 * n1));
        // Elements.py:84-109 Reflectance, with cos/sin taken algebraically (the physical
        // value also where the reference's unnormalised arccos turns NaN)
        real st2 = r * r * (real(1) - c * c);
        real ct = M<real>::sqrt(real(1) - st2);
        real a1 = n1 * c, a2 = n2 * ct, b1 = n2 * c, b2 = n1 * ct;
        real rs = (a1 - a2) * M<real>::rcp(a1 + a2);
        real rp = (b1 - b2) * M<real>::rcp(b1 + b2);
        R_perp = rs * rs; R_par = rp * rp;
    }
    if (!tir) {
        // Elements.py:93-94 normalises E here; it already is a unit vector (every rotation and
        // every sampler ends with a normalisation), so only the float64 instantiation repeats it
        if (M<real>::has_self_hit_eps) normalize3(p.Ex, p.Ey, p.Ez);
        // D x (sgn n).  float64 rounds the six products on their own, as numpy's cross does: for a
        // direction that is bit for bit the negative of the normal the result is then exactly zero and
        // the reference takes its normal-incidence branch (Elements.py:98,107) -- a fused
        // multiply-subtract would leave the rounding error of one product there instead
        real kx, ky, kz;
        if constexpr (M<real>::has_self_hit_eps) {
            cross_as_reference(p.Dx, p.Dy, p.Dz, sgn * nx, sgn * ny, sgn * nz, kx, ky, kz);
        } else {
            kx = sgn * (p.Dy * nz - p.Dz * ny); ky = sgn * (p.Dz * nx - p.Dx * nz); kz = sgn * (p.Dx * ny - p.Dy * nx);
        }
        real k2 = kx * kx + ky * ky + kz * kz;
        if (!M<real>::has_self_hit_eps) {
            // exactly normal incidence (k = 0): both coefficients equal ((n1 - n2)/(n1 + n2))^2 up to
            // rounding, so float32 takes R_par there and needs no branch
            real dp = (p.Ex * kx + p.Ey * ky + p.Ez * kz);
            real dp2 = k2 != real(0) ? dp * dp * M<real>::rcp(k2) : real(0);
            R = (real(1) - dp2) * R_par + dp2 * R_perp;
        } else if (k2 != real(0)) {
            real dp = (p.Ex * kx + p.Ey * ky + p.Ez * kz);
            real dp2 = dp * dp * M<real>::rcp(k2);
            R = (real(1) - dp2) * R_par + dp2 * R_perp;
        } else {
            real q = (n1 - n2) * M<real>::rcp(n1 + n2);
            R = q * q;
        }
    } else {
        R = real(1);
    }
    return tir;
}

// PhotonObject.py:117-129 UpdatePolarization: rotate E about D_new x D_old by -arccos(D_new . D_old),
// i.e. by the smallest rotation that takes a = D_old/|D_old| to b = D_new/|D_new|:
//   R E = E - s (s.E)/h + 2 b (a.E),  s = a + b,  h = s.s/2 = 1 + a.b
// (no rotation when a = -b: the reference finds a zero cross product there).
// For the line-based reflection (`reflect`, dn = D_old . n) s = 2 (n.a) n exactly, which keeps
// grazing reflections (a close to -b) as accurate as any other.
template <typename real>
__device__ __forceinline__ void rotate_polarization(Photon<real> &p, real oDx, real oDy, real oDz, real dnorm,
                                                    bool reflect, real dn, real nx, real ny, real nz)
{
    const real ia = M<real>::rcp(dnorm);
    const real ib = M<real>::rsqrt(p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz);
    const real k = real(2) * dn * ia;
    const real sx = reflect ? k * nx : oDx * ia + p.Dx * ib;
    const real sy = reflect ? k * ny : oDy * ia + p.Dy * ib;
    const real sz = reflect ? k * nz : oDz * ia + p.Dz * ib;
    const real h = real(0.5) * (sx * sx + sy * sy + sz * sz);
    // h = 0 (a = -b: no rotation) leaves s = 0 and E + 2 b (a.E) with a.E = 0 up to rounding, so
    // float32 guards the division only; float64 skips the update like the reference
    if (!M<real>::has_self_hit_eps || h > real(0)) {
        const real se = (sx * p.Ex + sy * p.Ey + sz * p.Ez) *
                        M<real>::rcp(M<real>::has_self_hit_eps || h > real(1e-30) ? h : real(1e-30));
        const real ae = real(2) * (oDx * p.Ex + oDy * p.Ey + oDz * p.Ez) * (ia * ib);
        real ex = p.Ex - sx * se + p.Dx * ae;
        real ey = p.Ey - sy * se + p.Dy * ae;
        real ez = p.Ez - sz * se + p.Dz * ae;
        normalize3(ex, ey, ez);
        p.Ex = ex; p.Ey = ey; p.Ez = ez;
    }
}

// The same rotation in Rodrigues form about a known axis c = D_new x D_old (the lens).
template <typename real>
__device__ __forceinline__ void rotate_polarization_about(Photon<real> &p, real cx, real cy, real cz, real oDx,
                                                          real oDy, real oDz, real dnorm)
{
    real c2n = cx * cx + cy * cy + cz * cz;
    if (c2n != real(0)) {
        real nn = p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz;
        real iq = M<real>::rsqrt(nn * (dnorm * dnorm));  // 1 / (|D_new| |D_old|)
        real c = (p.Dx * oDx + p.Dy * oDy + p.Dz * oDz) * iq;
        if (c <= real(1) && c >= real(-1)) {
            real ce = cx * p.Ex + cy * p.Ey + cz * p.Ez;
            real coef = c >= real(0) ? iq * iq * M<real>::rcp(real(1) + c) : (real(1) - c) * M<real>::rcp(c2n);
            real xx = cy * p.Ez - cz * p.Ey, xy = cz * p.Ex - cx * p.Ez, xz = cx * p.Ey - cy * p.Ex;
            real ex = c * p.Ex - xx * iq + cx * (ce * coef);
            real ey = c * p.Ey - xy * iq + cy * (ce * coef);
            real ez = c * p.Ez - xz * iq + cz * (ce * coef);
            normalize3(ex, ey, ez);
            p.Ex = ex; p.Ey = ey; p.Ez = ez;
        }
    }
}

// Tally sinks -----------------------------------------------------------------
// LdsTally: the trace kernel (workgroup accumulators in LDS, detector
// histogram in HBM).  NoTally: the step-wise kernel.
template <typename real, bool SPECTRA, bool HBMHIST = false>
struct LdsTally {
    unsigned long long *acc;
    uint32_t *hist;
    uint32_t *ghist;  // HBMHIST kernels: this workgroup's histogram in HBM if the scene keeps it there, else null
    const uint32_t *fx;
    const uint16_t *bin;
    const double *wl;
    int n_el, n_bins;
    bool fixed_source;
    // per-lane totals, flushed once at the end of the kernel
    unsigned long long lane_steps;
    uint32_t lane_abs_once;
#ifdef RTMC_PROFILE_SECTIONS
    // cycle stamps (diagnostic build only): per lane the cycles and the number of passes of every
    // stamped interval it took part in; the host divides the sums.  0 Philox, 1 search, 2 absorption
    // check, 3 element operation (step); 4 endings, 5 pathway + wavelength of absorbed photons,
    // 6 photon ids, 7 Philox, 8 direction / polarisation / free path (renew block)
    enum { N_PROF = 9 };
    unsigned long long prof[N_PROF], prof_t;
    uint32_t prof_n[N_PROF];
    __device__ __forceinline__ void stamp_begin()
    {
        __builtin_amdgcn_sched_barrier(0);
        prof_t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void stamp(int k)
    {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        prof[k] += t - prof_t; prof_n[k]++; prof_t = t;
    }
#else
    __device__ __forceinline__ void stamp_begin() {}
    __device__ __forceinline__ void stamp(int) {}
#endif
    __device__ __forceinline__ void scalar(int which, unsigned long long v = 1ull) { atomicAdd(&acc[which], v); }
    __device__ __forceinline__ void bounce(int el) { atomicAdd(&acc[ACC_SCALARS + el], 1ull); }
    __device__ __forceinline__ void ended(int iter, bool absorbed_once)
    {
        lane_steps += (unsigned long long)iter;
        lane_abs_once += absorbed_once ? 1u : 0u;
    }
    // Elements.py:152-165 Dump tallies.  Counter and the 1-nm spectrum come from
    // the (detector, bin) histogram, which lives in LDS: detector hits are one
    // per photon and scattered global atomics at that rate would bound the kernel.
    __device__ __forceinline__ void dump(int el, int slot, int wid, int wid0, bool counted)
    {
        if (HBMHIST && ghist) atomicAdd(&ghist[mul24(slot, n_bins) + bin[wid]], 1u);
        else atomicAdd(&hist[mul24(slot, n_bins) + bin[wid]], 1u);
        unsigned long long e = (!SPECTRA || fixed_source)
            ? (unsigned long long)fx[wid]
            : (unsigned long long)__double2ll_rn(wl[wid0] / wl[wid] * RTMC_ENERGY_FX_SCALE);
        atomicAdd(&acc[ACC_SCALARS + 2 * n_el + el], e);
        // (without efficiency tables every photon counts and the host takes the counter itself)
        if (SPECTRA && counted) atomicAdd(&acc[ACC_SCALARS + n_el + el], 1ull);
    }
};
struct NoTally {
    __device__ __forceinline__ void scalar(int, unsigned long long = 1ull) {}
    __device__ __forceinline__ void bounce(int) {}
    __device__ __forceinline__ void ended(int, bool) {}
    __device__ __forceinline__ void dump(int, int, int, int, bool) {}
    __device__ __forceinline__ void stamp_begin() {}
    __device__ __forceinline__ void stamp(int) {}
};

enum StepResult { STEP_CONTINUE = 0, STEP_ENDED = 1, STEP_ABSORBED = 2,
                  STEP_WIDE = 3 };  // FEAT_CELLS: nothing done, the photon waits for a pass that walks the element list


// PhotonObject.py:216-233: an absorbed photon moves the rest of its free path
// towards the candidate point of its last search (p.best_t, p.hitd); with no
// candidate the reference aims at PointCy = (0,0,0).
template <typename real, typename Tally>
__device__ __forceinline__ void finish_absorbed_move(Photon<real> &p, Tally &tally)
{
    const bool stale = !(p.hitd < M<real>::inf());
    p.oldx = p.Lx; p.oldy = p.Ly; p.oldz = p.Lz;
    real ux, uy, uz;
    if (!stale) {
        const real dnorm = M<real>::sqrt(p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz);
        real s = (p.best_t < real(0) ? real(-1) : real(1)) * M<real>::rcp(dnorm);
        ux = p.Dx * s; uy = p.Dy * s; uz = p.Dz * s;
    } else {
        ux = -p.Lx; uy = -p.Ly; uz = -p.Lz;
        normalize3(ux, uy, uz);
    }
    p.Lx += ux * p.dist; p.Ly += uy * p.dist; p.Lz += uz * p.dist;
    set_admissibility_forward(p);
    tally.scalar(ACC_ABSORPTIONS);
}

// Elements.py:152-165: the tallies of the detector the photon ended on
// (p.last_el); the efficiency draw is word 0 of the iteration's variates.
template <bool SPECTRA, typename real, bool TAPE, typename Tally>
__device__ __forceinline__ void finish_dumped(const HotScene<real> &sc, const Lds<real> &lds, const Photon<real> &p,
                                              Rng<real, TAPE> &g, Tally &tally)
{
    const int info = row(lds.elems, p.last_el).info;
    bool counted = true;
    const int eqe = (info & 0xFF) - 1;
    if (SPECTRA && eqe >= 0) counted = g.u32(0) < sc.eqeT[mul24(eqe, sc.n_wid) + p.wid];
    tally.dump(p.last_el, info >> 8, p.wid, p.wid0, counted);
}

// What RayTrace.py:187-189 does with the element the search found: OperationPlane, then the
// element's ExecuteOperation (Elements.py:18-24, 41-73, 125-129, 152-169) with
// PrepareExecution before and PostProcessExecution after it.  `m` is the photon's medium,
// `stale` says that the search found nothing and the last element is executed again,
// best_t is the line parameter of the candidate point (cylinders take their tangent
// plane there).  Returns true if the photon ended on a detector.
template <int FEATURES, bool DEFER_ENDINGS, typename real, bool TAPE, typename Tally>
__device__ __forceinline__ bool execute_operation(const HotScene<real> &sc, const Lds<real> &lds, Photon<real> &p,
                                                  Rng<real, TAPE> &g, Tally &tally, const DevMat<real> &m, int el_idx,
                                                  bool stale, real best_t, real dnorm)
{
    constexpr bool ALIGNED = (FEATURES & FEAT_ALIGNED) != 0;
    constexpr bool SHAPES = (FEATURES & FEAT_SHAPES) != 0;
    const DevElem<real> &el = row(lds.elems, el_idx);  // per-lane gather from LDS
    real nx = el.nx, ny = el.ny, nz = el.nz, d0 = el.d0;
    const int type = el.type & 0xFF, info = el.info;
    if (SHAPES) {
        if ((el.type >> 8) == RTMC_GEOM_CYLINDER) {
            // RayTrace.py:187: tangent plane at the candidate point (the origin if nothing was found)
            const real Px = stale ? real(0) : p.Lx + best_t * p.Dx, Py = stale ? real(0) : p.Ly + best_t * p.Dy,
                       Pz = stale ? real(0) : p.Lz + best_t * p.Dz;
            cylinder_normal(el, Px, Py, Pz, nx, ny, nz);
            d0 = nx * Px + ny * Py + nz * Pz;
        }
    }
    // PhotonObject.py:135-139 PrepareExecution
    const real dn = dot_as_reference(p.Dx, p.Dy, p.Dz, nx, ny, nz);
    const real off = p.Lx * nx + p.Ly * ny + p.Lz * nz - d0;  // signed distance of the segment start
    const real sgn = real(sgn_i(-dn));
    const real oDx = p.Dx, oDy = p.Dy, oDz = p.Dz;
    p.oldx = p.Lx; p.oldy = p.Ly; p.oldz = p.Lz;
    if constexpr ((FEATURES & FEAT_CELLS) != 0) p.flags &= ~F_OUTBOUND;  // (set again below if it holds after this operation)
    // a stale re-execution starts on the plane itself: its old point has no side
    const int side = stale ? 0 : sgn_i(off);
    // every operation first moves the photon onto the plane
    // (PhotonObject.py:149/198; Elements.py:167 for the detector)
    if (type == RTMC_TYPE_DUMP && stale) {
        p.Lx = p.Ly = p.Lz = real(0);
    } else {
        move_onto_plane(p, off, dn);
    }
    // (a re-execution in place moves the photon by a rounding residue only)
    if constexpr ((FEATURES & FEAT_GUARD) != 0) p.guard = landing_uncertainty(p, stale ? real(0) : p.hitd, dnorm, dn);
    if (type == RTMC_TYPE_DUMP) {  // Elements.py:152-169
        p.status = RTMC_ST_DUMPED;
        if (!DEFER_ENDINGS) finish_dumped<(FEATURES & FEAT_SPECTRA) != 0>(sc, lds, p, g, tally);
        p.flags |= F_WAS_EL;
        return true;
    }
    bool reflect = true;
    bool have_axis = false;  // D_new x D_old already known
    real cx = real(0), cy = real(0), cz = real(0);
    if (SHAPES) {
        if (type == RTMC_TYPE_LENS) {  // Elements.py:125-129, PhotonObject.py:158-191
            reflect = false;
            have_axis = true;
            focus_thin_lens(p, el.focal, off, dn, nz, el.sz, cx, cy, cz);
        }
    }
    if (type == RTMC_TYPE_INTERFACE) {  // Elements.py:41-73
        // the medium on the far side, chosen by name (Elements.py:45-48)
        const bool from1 = m.name_id == el.nid1;
        const int other = from1 ? (info >> 8) & 0xFF : info & 0xFF;
        const int other_slot = (from1 ? (el.slots >> 16) : (el.slots & 0xFFFF)) - 1;
        const real n1 = m.n, n2 = from1 ? el.n2v : el.n1v;
        const real r = M<real>::has_self_hit_eps ? n1 / n2 : n1 * M<real>::rcp(n2);
        real R;
        const bool tir = interface_reflectance(p, n1, n2, r, dn, dnorm, sgn, nx, ny, nz, R);
        if (!tir) {
            if (g.uniform(0) > R) {  // Elements.py:57-66 crossing
                reflect = false;
                // PhotonObject.py:199 Snell
                refract_direction(p, r, -dn, sgn, nx, ny, nz);
                p.mat = other;
                p.flags &= ~F_REFLECTED;
                if (other_slot >= 0)  // PhotonObject.py:283-305, with E as Reflectance left it
                    p.dist = free_path<ALIGNED>(sc, lds, other, other_slot, p, g.expo(1));
                else
                    p.dist = real(0);
            }
        }
        if (reflect) p.flags |= F_REFLECTED;
    }
    if (reflect) {  // PhotonObject.py:150 (line-based: the negative of the mirror direction)
        reflect_direction(p, dn, nx, ny, nz);
        tally.bounce(el_idx);
    }
    // PhotonObject.py:141-143 PostProcessExecution
#ifndef RTMC_ABLATE_ROTATE
    if (SHAPES && have_axis) rotate_polarization_about(p, cx, cy, cz, oDx, oDy, oDz, dnorm);
    else rotate_polarization(p, oDx, oDy, oDz, dnorm, reflect, dn, nx, ny, nz);
#endif
    p.flags |= F_WAS_EL;  // PhotonObject.py:143
    if constexpr ((FEATURES & FEAT_CELLS) != 0) {
        // The photon now sits on a face of cell c.  Which side of the face it came from follows from its motion
        // along the line (sign(t) D, t being negative for a photon that travels backwards along its line after a
        // reflection) and from whether the element's normal points out of c; a reflection leaves it on that side,
        // a refraction takes it to the other.  Inside: it is in c.  Outside: it is in the cell the host names for
        // this side of the face, and it moves away from c, which it cannot meet again before it has met something
        // else (c is convex).  A re-execution in place (nothing was found) says nothing: the cell stays unknown.
        const uint32_t c = ((uint32_t)info >> 16) & 0x7Fu, outer = (uint32_t)info >> 24;
        if (c != 0u && !stale) {
            const bool normal_outward = (((uint32_t)info >> 23) & 1u) != 0u;
            const bool from_inside = ((best_t < real(0)) != (dn < real(0))) != normal_outward;
            const bool inside = from_inside == reflect;
            p.flags = (p.flags & ~(F_CELL_MASK | F_OUTBOUND)) | ((inside ? c : outer) << F_CELL_SHIFT) | (inside ? 0u : F_OUTBOUND);
        }
    }
    set_admissibility(p, type, side, nx * p.Dx + ny * p.Dy + nz * p.Dz);
    return false;
}

// One iteration of the loop body, RayTrace.py:178-195, up to the point where an
// absorbed photon needs its pathway drawn: the caller does that
// (emit_in_reference_order / the renew block of the trace kernel) and then
// calls finish_absorbed_iteration.
//
// DEFER_ENDINGS: what an absorbed or detected photon still has to do after the
// iteration has decided its fate (the move to the absorption point, the
// detector tallies) is left to finish_absorbed_move / finish_dumped.  Such a
// lane takes no further step before the caller has dealt with it, and a caller
// that deals with the lanes of several iterations at once (the renew block of
// the trace kernel) runs that code once for all of them.
//
// Kernels with FEAT_CELLS: `walk_list` (wave-uniform) says how this pass searches -- by walking the
// element list for every lane in it, or per lane from the cell the photon is confined to; a lane
// that cannot answer from its cell returns STEP_WIDE with its photon untouched.
template <int FEATURES, bool DEFER_ENDINGS, typename real, bool TAPE, typename Tally>
__device__ __forceinline__ StepResult photon_step(const HotScene<real> &sc, const Lds<real> &lds,
                                                  Photon<real> &p, Rng<real, TAPE> &g, Tally &tally, bool walk_list = true)
{
    // the variates of this iteration (every lane computes its block together;
    // which words are used depends on what the iteration turns out to be)
    tally.stamp_begin();
    g.block((uint32_t)p.iter, BLK_STEP);
    tally.stamp(0);
    real dnorm;
    Nearest<real> nearest;
    if constexpr ((FEATURES & FEAT_CELLS) != 0 && sizeof(real) == 4 && !TAPE) {
        if (walk_list) {
            nearest = nearest_valid_hit<FEATURES>(sc, lds, p, dnorm);
            // nothing found: the photon is in the open, outside every closed box.  What the reference does with it
            // from here on (the last operation again and again, in place, which also flips its medium,
            // RayTrace.py:124-154) has no geometric meaning that a cell number could follow.
            if (nearest.e < 0) p.flags |= F_CELL_MASK;
        } else {
            confined_hit<(FEATURES & FEAT_RUNS) != 0, (FEATURES & FEAT_FRAMES) != 0>(lds, p, dnorm, nearest);
            if (nearest.e == CONFINED_NONE) {
                tally.stamp(1);
                return STEP_WIDE;
            }
        }
        p.flags &= ~F_PRESET;
    } else {
        nearest = nearest_valid_hit<FEATURES>(sc, lds, p, dnorm);
    }
    tally.stamp(1);
    const real best_t = nearest.t;
    const int best_e = nearest.e;
    const bool stale = best_e < 0;
    const int el_idx = stale ? p.last_el : best_e;
    p.hitd = stale ? M<real>::inf() : M<real>::abs(best_t) * dnorm;
    p.best_t = best_t;
    p.last_el = el_idx;
    if (stale) tally.scalar(ACC_STALE);
    if (el_idx < 0) {  // nothing hit and no element visited before: the reference raises
        p.status = RTMC_ST_LOST;
        p.iter++;
        return STEP_ENDED;
    }

    // ---- PhotonObject.py:206-264 CheckForAbsorption -------------------------
    const DevMat<real> &m = row(lds.mats, p.mat);
    if (m.flags & 1) {
        if (p.dist > p.hitd) {
            p.dist -= p.hitd;
            p.status = RTMC_ST_START;
        } else {
            p.flags = (p.flags | F_ABS_ONCE) & ~(F_WAS_EL | F_OUTBOUND);
            p.status = RTMC_ST_ABSORBED;
            if (!DEFER_ENDINGS) finish_absorbed_move(p, tally);
            return STEP_ABSORBED;
        }
    } else {
        p.status = RTMC_ST_START;
        p.flags |= F_WAS_EL;
    }

    tally.stamp(2);
    // ---- the element's ExecuteOperation -------------------------------------
    if (execute_operation<FEATURES, DEFER_ENDINGS>(sc, lds, p, g, tally, m, el_idx, stale, best_t, dnorm)) {
        p.iter++;
        return STEP_ENDED;
    }

    tally.stamp(3);
    // ---- RayTrace.py:193-195 ------------------------------------------------
    const bool stop = p.iter >= sc.max_iter - 1;
    p.iter++;
    return stop ? STEP_ENDED : STEP_CONTINUE;
}

// Loop control for an iteration that ended in an absorption, once the pathway
// is known (RayTrace.py:193-195).
template <typename real>
__device__ __forceinline__ StepResult finish_absorbed_iteration(const HotScene<real> &sc, Photon<real> &p)
{
    const bool stop = (p.status == RTMC_ST_INTERNAL_CONVERTED) | (p.iter >= sc.max_iter - 1);
    p.iter++;
    return stop ? STEP_ENDED : STEP_CONTINUE;
}

// book-keeping when a photon has ended, RayTrace.py:197-199 (detected photons
// are counted through the detector histogram)
template <typename real, typename Tally>
__device__ __forceinline__ void count_fate(const Photon<real> &p, Tally &tally)
{
    tally.ended(p.iter, (p.flags & F_ABS_ONCE) != 0u);
    if (p.status != RTMC_ST_DUMPED)
        tally.scalar(p.status == RTMC_ST_INTERNAL_CONVERTED ? ACC_IC
                     : p.status == RTMC_ST_LOST ? ACC_LOST : ACC_CAPPED);
}
#endif  // __HIPCC__

}  // namespace rtmc
