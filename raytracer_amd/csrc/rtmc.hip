// rtmc.hip -- kernels and the C ABI of librtmc.so (gfx950 / MI355X).
//
// Host side: builds the device images of a scene (wavelength ids, absorption
// coefficients per id, integer inverse-CDF thresholds), owns per-device
// buffers and a stream, launches the kernels, folds the device tallies into the
// caller's block.  Device side: see rtmc_device.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rtmc_device.h"

using namespace rtmc;

#define RTMC_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
namespace {

#ifndef RTMC_TRACE_BLOCK
#define RTMC_TRACE_BLOCK 512
#endif
constexpr int TRACE_BLOCK = RTMC_TRACE_BLOCK;  // threads per workgroup of the trace kernel
constexpr uint32_t WORK_CHUNK = 1024; // photon ids a wave takes from the global counter at a time

template <typename real>
__device__ __forceinline__ SceneArg<real> scene_ref(const DevScene<real> *ptr)
{
    return *(const DevScene<real> __attribute__((address_space(4))) *)(uintptr_t)ptr;
}

template <typename real>
__device__ __forceinline__ HotScene<real> hot_scene(SceneArg<real> sc)
{
    HotScene<real> h;
    h.hits = sc.hits; h.elems = sc.elems; h.eqeT = sc.eqeT;
    for (int k = 0; k < 4; k++) h.n_group[k] = sc.n_group[k];
    h.n_wid = (int32_t)sc.lay.n_wid; h.max_iter = sc.max_iter; h.inv_scale = sc.src.inv_scale;
    h.use_guard = sc.use_guard;

    return h;
}

template <typename real>
__device__ __forceinline__ Lds<real> stage_lds(SceneArg<real> sc, unsigned char *smem, bool with_accumulators)
{
    const uint32_t words = sc.lay.acc / 4;
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem);
    const uint32_t *src = sc.lds_image;
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    if (with_accumulators) {  // acc and hist are adjacent
        const uint32_t zero_words = (sc.lay.total - sc.lay.acc) / 4;
        for (uint32_t i = threadIdx.x; i < zero_words; i += blockDim.x) dst[words + i] = 0u;
    }
    __syncthreads();
    Lds<real> l;
    l.elems = reinterpret_cast<const DevElem<real> *>(smem + sc.lay.elems);
    l.mats = reinterpret_cast<const DevMat<real> *>(smem + sc.lay.mats);
    l.orient = reinterpret_cast<const DevOrient<real> *>(smem + sc.lay.orient);
    l.abs = reinterpret_cast<const DevAbs<real> *>(smem + sc.lay.abs);
    l.cdf = reinterpret_cast<const uint32_t *>(smem + sc.lay.cdf);
    l.guide = reinterpret_cast<const uint16_t *>(smem + sc.lay.guide);
    l.fx = reinterpret_cast<const uint32_t *>(smem + sc.lay.fx);
    l.bin = reinterpret_cast<const uint16_t *>(smem + sc.lay.bin);
    l.run_cells = reinterpret_cast<const uint32_t *>(smem + sc.lay.run_cells);
    l.run_members = reinterpret_cast<const RunMember<real> *>(smem + sc.lay.run_members);
    l.cells = reinterpret_cast<const CellHeader *>(smem + sc.lay.cells);
    l.tile_grids = reinterpret_cast<const TileGrid *>(smem + sc.lay.tile_grids);
    l.cell_frames = reinterpret_cast<const float *>(smem + sc.lay.cell_frames);
    l.acc = reinterpret_cast<unsigned long long *>(smem + sc.lay.acc);
    l.hist = reinterpret_cast<uint32_t *>(smem + sc.lay.hist);
    return l;
}

__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Traces photon ids [first, first + n).  Every wave draws blocks of ids from
// `work`; a lane whose photon has ended or has been absorbed goes through the
// *renew* block (new id + launch, or pathway choice + re-emission) before it
// takes part in the next loop iteration.  Launch and re-emission share one
// Philox event and one pass through the direction / polarisation samplers.
// The grid is sized to the machine, not to n.
// NEED_RETIRE: the photon has ended but its tallies are still to be written;
// NEED_EMIT: absorbed, still at the point where its last search began.  Both are
// dealt with at the head of the renew block, for the lanes of all iterations
// since the last pass at once.
// NEED_WIDE (kernels with FEAT_CELLS): the photon's next search has to walk the element list.
enum { NEED_NONE = 0, NEED_LAUNCH = 1, NEED_EMIT = 2, NEED_OFF = 3, NEED_RETIRE = 4, NEED_WIDE = 5 };
// When to run the renew block.  It costs R (about half a loop iteration)
// whatever the number of lanes in it, while every lane that waits for it wastes
// 1/64 of an iteration S per iteration.  Running it every P iterations costs
// R/P + (waste accumulated over P)/P per iteration, which is smallest when the
// accumulated waste equals R: the block runs once the lane-iterations spent
// waiting since the last run reach 64 R/S.  Scenes whose photons end after a few
// iterations get there every other iteration with half the wave waiting; scenes
// whose photons live for hundreds of iterations with three or four lanes waiting.
// Measured (8e8 photons, builds with 8 | 16 | 24 | 32 | 40 | 56 | 80): the single slab is flat
// from 32 to 56 and loses 4 % at 24, 10 % at 16; the three stacked slabs and the thin film, whose
// photons live three times as long, are best at 24 to 32 (+1 % over 40) -- one value serves all.
#ifndef RTMC_RENEW_DEBT
#define RTMC_RENEW_DEBT 32
#endif
constexpr int RENEW_DEBT = RTMC_RENEW_DEBT;  // lane-iterations
// Kernels with FEAT_CELLS: a pass of the loop searches either per lane from the photons' cells (cheap; lanes
// whose photon is not confined to a cell sit it out) or by walking the element list for every lane.  The same
// reasoning as for the renew block: the list walk costs its extra cycles whatever the number of lanes that
// need it, every lane that waits for it wastes 1/64 of a pass, so it runs once the lane-passes spent waiting
// reach a threshold (or when no lane can do without it).
#ifndef RTMC_WIDE_DEBT
#define RTMC_WIDE_DEBT 16
#endif
constexpr int WIDE_DEBT = RTMC_WIDE_DEBT;  // lane-passes

// Register budget: the float32 kernels are compiled for six waves per SIMD
// (three 512-thread workgroups per CU, at most 80 VGPRs; the lean instantiation
// takes 62).  Six waves run 13 % faster than four; seven or eight (which 62
// registers would allow) run slower.  float64 (validation only) is left
// unconstrained.
#ifndef RTMC_TRACE_WAVES
#define RTMC_TRACE_WAVES 6
#endif
// Workgroup size: 512 threads when three workgroups fit a CU.  A scene whose
// tables leave room for only two or one (a source spectrum adds a row per
// wavelength to every per-wavelength table) gets 768 or 1024 threads per
// workgroup instead, so that the SIMDs still hold six or four waves; the host
// picks the size per scene and the float32 kernels are compiled to be launched
// with any of them.  (Wave counts that are not a multiple of four per workgroup
// would leave workgroups unplaceable.)
constexpr int TRACE_BLOCK_MAX = 1024;
template <typename real, int FEATURES>
__global__ void __launch_bounds__(sizeof(real) == 4 ? TRACE_BLOCK_MAX : TRACE_BLOCK)
__attribute__((amdgpu_waves_per_eu((sizeof(real) == 4 ? RTMC_TRACE_WAVES : 1), (sizeof(real) == 4 ? RTMC_TRACE_WAVES : 4))))
trace_kernel(const DevScene<real> *scene, DevTallies tl, uint64_t seed, uint64_t first, uint64_t n,
             unsigned long long *work)
{
    constexpr bool ALIGNED = (FEATURES & FEAT_ALIGNED) != 0;
    constexpr bool CELLS = (FEATURES & FEAT_CELLS) != 0 && sizeof(real) == 4;
    extern __shared__ __align__(16) unsigned char smem[];
    SceneArg<real> sc = scene_ref(scene);
    Lds<real> lds = stage_lds(sc, smem, true);
    const HotScene<real> hot = hot_scene(sc);

    LdsTally<real, (FEATURES & FEAT_SPECTRA) != 0, (FEATURES & FEAT_HBMHIST) != 0> tally;
    tally.ghist = ((FEATURES & FEAT_HBMHIST) && sc.lay.hist_in_hbm)
                      ? tl.private_hist + (size_t)blockIdx.x * (sc.lay.n_dump * sc.lay.n_bins) : nullptr;
    tally.acc = lds.acc; tally.hist = lds.hist; tally.fx = lds.fx; tally.bin = lds.bin; tally.wl = sc.wl;
    tally.n_el = sc.n_el; tally.n_bins = (int)sc.lay.n_bins;
    tally.fixed_source = sc.src.wl_mode == 0;
    tally.lane_steps = 0ull; tally.lane_abs_once = 0u;
#ifdef RTMC_PROFILE_SECTIONS
    for (int k = 0; k < tally.N_PROF; k++) { tally.prof[k] = 0ull; tally.prof_n[k] = 0u; }
    // shader clock against the constant 100 MHz counter, over the life of the wave
    const unsigned long long life_c0 = __builtin_amdgcn_s_memtime(), life_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    Photon<real> p;
    Rng<real, false> g;
    g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32);
    g.id0 = g.id1 = 0;
    int need = NEED_LAUNCH;
    uint64_t chunk_next = 0, chunk_end = 0;  // wave-uniform
    bool exhausted = false;
    int idle_debt = 0;  // wave-uniform: lane-iterations spent waiting for the renew block
    [[maybe_unused]] int wide_debt = 0;  // wave-uniform: lane-passes spent waiting for a walk over the element list
    const uint32_t lane = threadIdx.x & 63;

#ifdef RTMC_PROFILE_SECTIONS
    unsigned long long cyc_renew = 0, cyc_step = 0, n_renew = 0, n_step = 0, lanes_renew = 0, lanes_step = 0, n_walk = 0;
#define SECTION_BEGIN() const unsigned long long sec_t0 = __builtin_readcyclecounter()
#define SECTION_END(acc, cnt, lanes, mask) do { acc += __builtin_readcyclecounter() - sec_t0; cnt++; lanes += __popcll(mask); } while (0)
#else
#define SECTION_BEGIN() do {} while (0)
#define SECTION_END(acc, cnt, lanes, mask) do {} while (0)
#endif
    for (;;) {
        // (Single exit and no `continue`: with two back edges the compiler copied the
        // whole photon record to other registers and back on every iteration.)
        const unsigned long long want = __ballot(need == NEED_LAUNCH || need == NEED_EMIT || need == NEED_RETIRE);
        const unsigned long long stepping = __ballot(need == NEED_NONE || (CELLS && need == NEED_WIDE));
        if ((want | stepping) == 0ull) break;  // every lane is switched off
        idle_debt += __popcll(want);
        if (want != 0ull && (idle_debt >= RENEW_DEBT || stepping == 0ull)) {
            idle_debt = 0;
            SECTION_BEGIN();
#ifdef RTMC_PROFILE_SECTIONS
            const unsigned long long renew_mask = __ballot(need == NEED_LAUNCH || need == NEED_EMIT || need == NEED_RETIRE);
#endif
            // (a) absorbed photons: donor/acceptor coin and pathway (PhotonObject.py:244,
            //     MonteCarloFRET.py:63-193); one event, a back transfer opens another
            bool emitting = false;
            tally.stamp_begin();
            if (need == NEED_RETIRE) {
                if (p.status == RTMC_ST_DUMPED) finish_dumped<(FEATURES & FEAT_SPECTRA) != 0>(hot, lds, p, g, tally);
                count_fate(p, tally);
                need = NEED_LAUNCH;
            }
            tally.stamp(4);
            if (need == NEED_EMIT) {
                finish_absorbed_move(p, tally);
                const DevMat<real> &m = row(lds.mats, p.mat);
                const DevAbs<real> &a = row(lds.abs, mul24(m.abs_slot, hot.n_wid) + p.wid);
                const bool in_donor = absorbed_by_donor<ALIGNED>(m, a, p, g);
                uint32_t v_lambda;
                const int outcome = fret_decide<(FEATURES & FEAT_BACK) != 0>(m, g, in_donor, (uint32_t)p.iter, v_lambda);
                if (outcome != 0) {  // one pick for both dyes: the table is chosen per lane
                    const bool donor = outcome == 1;
                    const int off = donor ? m.cdfD_off : m.cdfA_off, len = donor ? m.cdfD_len : m.cdfA_len;
                    const int goff = donor ? m.gD_off : m.gA_off, woff = donor ? m.widD_off : m.widA_off;
                    p.wid = woff + cdf_pick_guided(lds.cdf + off, len, lds.guide + goff, v_lambda, m.guide_shift);
                    p.status = donor ? RTMC_ST_REEMITTED_DONOR : RTMC_ST_REEMITTED_ACCEPTOR;
                } else {
                    p.status = RTMC_ST_INTERNAL_CONVERTED;
                }
                if (finish_absorbed_iteration(hot, p) == STEP_ENDED) {
                    count_fate(p, tally);
                    need = NEED_LAUNCH;
                } else {
                    emitting = true;
                }
            }
            tally.stamp(5);
            // (b) photon ids for the lanes without a photon
            bool launching = false;
            const unsigned long long idle = __ballot(need == NEED_LAUNCH);
            if (idle != 0ull && !exhausted) {
                if (chunk_next == chunk_end) {
                    unsigned long long s = 0;
                    if (lane == 0) s = atomicAdd(work, (unsigned long long)WORK_CHUNK);
                    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)s);
                    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(s >> 32));
                    s = ((unsigned long long)hi << 32) | lo;
                    if (s >= n) exhausted = true;
                    else { chunk_next = s; chunk_end = (s + WORK_CHUNK < n) ? s + WORK_CHUNK : n; }
                }
                if (!exhausted) {
                    const uint32_t avail = (uint32_t)(chunk_end - chunk_next);
                    const uint32_t rank = lanes_below(idle);
                    if (need == NEED_LAUNCH && rank < avail) {
                        const uint64_t id = first + chunk_next + rank;
                        g.id0 = (uint32_t)id; g.id1 = (uint32_t)(id >> 32);
                        launching = true;
                    }
                    const uint32_t want = (uint32_t)__popcll(idle);
                    chunk_next += want < avail ? want : avail;
                }
            }
            if (exhausted && need == NEED_LAUNCH) need = NEED_OFF;
            tally.stamp(6);
            // (c) one event for both: direction, polarisation, free path / wavelength
            if (launching | emitting) {
                // an emitting photon's iteration count has already moved on
                const uint32_t emit_iter = (uint32_t)p.iter - 1u;
                g.block(launching ? 0u : emit_iter, launching ? BLK_LAUNCH : BLK_EMIT);
                tally.stamp(7);
                bool special = false;  // fluorescence of an aligned acceptor
                if (ALIGNED && emitting)
                    special = (row(lds.mats, p.mat).flags & 2) && p.status == RTMC_ST_REEMITTED_ACCEPTOR;
                if (special)
                    renew_emit_aligned(hot, lds, row(lds.mats, p.mat), p, g, emit_iter);
                else
                    renew_photon<ALIGNED, (FEATURES & FEAT_AREA) != 0, (FEATURES & FEAT_SPECTRA) != 0, CELLS>(sc, hot, lds, p, g, launching);
                need = (CELLS && (p.flags & F_LIST)) ? NEED_WIDE : NEED_NONE;
                tally.stamp(8);
            }
            SECTION_END(cyc_renew, n_renew, lanes_renew, renew_mask);
        }
        SECTION_BEGIN();
#ifdef RTMC_PROFILE_SECTIONS
        const unsigned long long step_mask = __ballot(need == NEED_NONE || (CELLS && need == NEED_WIDE));
#endif
        if constexpr (CELLS) {
            const unsigned long long waiting = __ballot(need == NEED_WIDE);
            wide_debt += __popcll(waiting);
            const bool walk_list = waiting != 0ull && (wide_debt >= WIDE_DEBT || __ballot(need == NEED_NONE) == 0ull);
            if (walk_list) wide_debt = 0;
            if (need == NEED_NONE || (walk_list && need == NEED_WIDE)) {
                const uint32_t record = p.flags & (F_CELL_MASK | F_OUTBOUND);
                const StepResult r = photon_step<FEATURES, true>(hot, lds, p, g, tally, walk_list);
                // F_LIST: set when the search from the cell fails, dropped when the photon's cell record changes
                if (r == STEP_WIDE || (p.flags & F_CELL_MASK) == F_CELL_MASK) p.flags |= F_LIST;   // (cell unknown: for good)
                else if ((p.flags & (F_CELL_MASK | F_OUTBOUND)) != record) p.flags &= ~F_LIST;
                need = r == STEP_ENDED ? NEED_RETIRE
                     : (r == STEP_ABSORBED ? NEED_EMIT : ((p.flags & F_LIST) ? NEED_WIDE : NEED_NONE));
            }
#ifdef RTMC_PROFILE_SECTIONS
            if (lane == 0) { n_walk += walk_list ? 1 : 0; }
#endif
        } else if (need == NEED_NONE) {
            const StepResult r = photon_step<FEATURES, true>(hot, lds, p, g, tally);
            need = r == STEP_ENDED ? NEED_RETIRE : (r == STEP_ABSORBED ? NEED_EMIT : NEED_NONE);
        }
        SECTION_END(cyc_step, n_step, lanes_step, step_mask);
    }
#ifdef RTMC_PROFILE_SECTIONS
    if (lane == 0) {  // per-wave totals behind the work counter
        atomicAdd(&work[8], cyc_renew); atomicAdd(&work[9], n_renew); atomicAdd(&work[10], lanes_renew);
        atomicAdd(&work[11], cyc_step); atomicAdd(&work[12], n_step); atomicAdd(&work[13], lanes_step);
        atomicAdd(&work[14], n_walk);
        atomicAdd(&work[44], __builtin_amdgcn_s_memtime() - life_c0);
        atomicAdd(&work[45], __builtin_amdgcn_s_memrealtime() - life_r0);
    }
    for (int k = 0; k < tally.N_PROF; k++) {
        atomicAdd(&work[16 + k], tally.prof[k]);
        atomicAdd(&work[32 + k], (unsigned long long)tally.prof_n[k]);
    }
#endif

    // flush: per-lane totals into LDS, then the workgroup's accumulators to HBM
    if (tally.lane_steps) atomicAdd(&lds.acc[ACC_STEPS], tally.lane_steps);
    if (tally.lane_abs_once) atomicAdd(&lds.acc[ACC_ABS_ONCE], (unsigned long long)tally.lane_abs_once);
    __syncthreads();
    const int n_el = sc.n_el;
    const int n_acc = ACC_SCALARS + 3 * n_el;
    for (int i = threadIdx.x; i < n_acc; i += blockDim.x) {
        unsigned long long v = lds.acc[i];
        if (v == 0ull) continue;
        if (i < ACC_SCALARS) atomicAdd(&tl.scalars[i], v);
        else if (i < ACC_SCALARS + n_el) atomicAdd(&tl.bounce[i - ACC_SCALARS], v);
        else if (i < ACC_SCALARS + 2 * n_el) atomicAdd(&tl.spectral[i - ACC_SCALARS - n_el], v);
        else atomicAdd(&tl.energy_fx[i - ACC_SCALARS - 2 * n_el], v);
    }
    const int n_hist = sc.lay.hist_in_hbm ? 0 : (int)(sc.lay.n_dump * sc.lay.n_bins);
    for (int i = threadIdx.x; i < n_hist; i += blockDim.x) {
        uint32_t v = lds.hist[i];
        if (v) atomicAdd(&tl.hist[i], (unsigned long long)v);
    }
}

// The first search of a strict source's photons has the same answer for every one of them.  The
// kernels with cells take it from the header of the cell table instead of walking the element
// list; this kernel, run once when a scene is uploaded, finds it with that very walk (one wave, all
// lanes alike) and marks the launch flags if there is a hit.
template <int SEARCH>   // the features of the scene's kernel that the list walk depends on
__global__ void __launch_bounds__(64)
preset_hit_kernel(const DevScene<float> *scene, CellHeader *hdr)
{
    extern __shared__ __align__(16) unsigned char smem[];
    SceneArg<float> sc = scene_ref(scene);
    Lds<float> lds = stage_lds(sc, smem, false);
    const HotScene<float> hot = hot_scene(sc);
    Photon<float> p;
    p.Lx = sc.src.loc[0]; p.Ly = sc.src.loc[1]; p.Lz = sc.src.loc[2];
    p.Dx = sc.src.dir[0]; p.Dy = sc.src.dir[1]; p.Dz = sc.src.dir[2];
    p.Ex = p.Ey = p.Ez = 0.0f;
    p.wid = sc.src.wid_fixed;
    init_photon_common(sc, p);
    float dnorm;
    const Nearest<float> first = nearest_valid_hit<SEARCH>(hot, lds, p, dnorm);
    if (threadIdx.x == 0 && first.e >= 0) {
        hdr->preset_t = first.t;
        hdr->preset_e = first.e;
        hdr->launch_flags = (hdr->launch_flags & ~F_LIST) | F_PRESET;
    }
}

// device view of an rtmc_photon_soa batch (wavelengths already mapped to ids)
struct SoaDev {
    double *loc[3], *dir[3], *efield[3], *old_point[3];
    double *dist_left, *hit_distance, *psi;
    int32_t *wid, *wid0, *material, *status, *last_element, *was_element, *reflected, *iterations,
        *tape_pos;
};

// Tally sink of rtmc_op_photon: what the element counted, for the caller's own counters.
struct OpTally {
    int bounced = 0, counted = 0;
    __device__ __forceinline__ void scalar(int, unsigned long long = 1ull) {}
    __device__ __forceinline__ void bounce(int) { bounced = 1; }
    __device__ __forceinline__ void ended(int, bool) {}
    __device__ __forceinline__ void dump(int, int, int, int, bool was_counted) { counted = was_counted ? 1 : 0; }
    __device__ __forceinline__ void stamp_begin() {}
    __device__ __forceinline__ void stamp(int) {}
};

// One of the single operations of rtmc_op_photon on the photon of this lane.
// a: the RTMC_PHOP_ARGS arguments of the photon, out: its two result values or null.
template <typename real>
__device__ __forceinline__ void photon_operation(const HotScene<real> &sc, const Lds<real> &lds, Photon<real> &p, Rng<real, true> &g,
                                 int op, const double *a, double hit_distance, double *out)
{
    const real nx = real(a[3]), ny = real(a[4]), nz = real(a[5]);
    // PhotonObject.py:149: (S - L).n / (D.n)
    const real dn = dot_as_reference(p.Dx, p.Dy, p.Dz, nx, ny, nz);
    const real off = (p.Lx - real(a[0])) * nx + (p.Ly - real(a[1])) * ny + (p.Lz - real(a[2])) * nz;
    const real dnorm = M<real>::sqrt(p.Dx * p.Dx + p.Dy * p.Dy + p.Dz * p.Dz);
    switch (op) {
    case RTMC_PHOP_ABSORPTION: {  // PhotonObject.py:206-264
        const DevMat<real> &m = row(lds.mats, p.mat);
        if (m.flags & 1) {
            if (p.dist > real(hit_distance)) {
                p.dist -= real(hit_distance);
                p.status = RTMC_ST_START;
            } else {
                p.flags = (p.flags | F_ABS_ONCE) & ~F_WAS_EL;
                p.oldx = p.Lx; p.oldy = p.Ly; p.oldz = p.Lz;
                real ux = real(a[0]) - p.Lx, uy = real(a[1]) - p.Ly, uz = real(a[2]) - p.Lz;
                normalize3(ux, uy, uz);
                p.Lx += ux * p.dist; p.Ly += uy * p.dist; p.Lz += uz * p.dist;
                set_admissibility_forward(p);
                emit_in_reference_order(sc, lds, p, g);
            }
        } else {
            p.status = RTMC_ST_START;
            p.flags |= F_WAS_EL;
        }
        break;
    }
    case RTMC_PHOP_REFLECT:
        move_onto_plane(p, off, dn);
        reflect_direction(p, dn, nx, ny, nz);
        break;
    case RTMC_PHOP_REFRACT: {
        const real n1 = real(a[6]), n2 = real(a[7]);
        move_onto_plane(p, off, dn);
        refract_direction(p, M<real>::has_self_hit_eps ? n1 / n2 : n1 * M<real>::rcp(n2), -dn, real(a[8]), nx, ny, nz);
        break;
    }
    case RTMC_PHOP_LENS: {
        real cx, cy, cz;
        move_onto_plane(p, off, dn);
        focus_thin_lens(p, real(a[6]), off, dn, nz, real(a[2]), cx, cy, cz);
        break;
    }
    case RTMC_PHOP_POLARIZATION: {
        const real ox = real(a[9]), oy = real(a[10]), oz = real(a[11]);
        rotate_polarization(p, ox, oy, oz, M<real>::sqrt(ox * ox + oy * oy + oz * oz), false, real(0), real(0), real(0),
                            real(0));
        break;
    }
    case RTMC_PHOP_REFLECTANCE: {
        const real n1 = real(a[6]), n2 = real(a[7]);
        real R;
        const bool tir = interface_reflectance(p, n1, n2, M<real>::has_self_hit_eps ? n1 / n2 : n1 * M<real>::rcp(n2), dn,
                                               dnorm, real(a[8]), nx, ny, nz, R);
        if (out) { out[0] = tir ? 1.0 : 0.0; out[1] = tir ? 1.0 : (double)R; }
        break;
    }
    case RTMC_PHOP_EXECUTE: {
        const int el = (int)a[6];
        // line parameter of PointCy
        const real best_t = ((real(a[0]) - p.Lx) * p.Dx + (real(a[1]) - p.Ly) * p.Dy + (real(a[2]) - p.Lz) * p.Dz) *
                            M<real>::rcp(dnorm * dnorm);
        OpTally tally;
        p.last_el = el;
        p.best_t = best_t;
        execute_operation<FEAT_ALL, false>(sc, lds, p, g, tally, row(lds.mats, p.mat), el, false, best_t, dnorm);
        if (out) { out[0] = tally.bounced; out[1] = tally.counted; }
        break;
    }
    case RTMC_PHOP_FREEPATH: {
        const DevMat<real> &m = row(lds.mats, p.mat);
        p.dist = (m.flags & 1) ? free_path<true>(sc, lds, p.mat, m.abs_slot, p, g.expo(0)) : real(0);
        break;
    }
    default: break;
    }
}

// The photon of row k of a batch, as the kernels hold it.
template <typename real>
__device__ __forceinline__ void soa_load(const SoaDev &b, uint32_t k, const Lds<real> &lds, Photon<real> &p)
{
    p.Lx = real(b.loc[0][k]); p.Ly = real(b.loc[1][k]); p.Lz = real(b.loc[2][k]);
    p.Dx = real(b.dir[0][k]); p.Dy = real(b.dir[1][k]); p.Dz = real(b.dir[2][k]);
    p.Ex = real(b.efield[0][k]); p.Ey = real(b.efield[1][k]); p.Ez = real(b.efield[2][k]);
    p.oldx = real(b.old_point[0][k]); p.oldy = real(b.old_point[1][k]); p.oldz = real(b.old_point[2][k]);
    p.dist = real(b.dist_left[k]);
    p.psi = real(b.psi[k]);
    {   // length of the segment that brought the photon here (the float32 guard distance scales with it)
        const real sx = p.Lx - p.oldx, sy = p.Ly - p.oldy, sz = p.Lz - p.oldz;
        p.hitd = M<real>::sqrt(sx * sx + sy * sy + sz * sz);
    }
    p.wid = b.wid[k]; p.wid0 = b.wid0[k];
    p.mat = b.material[k]; p.status = b.status[k]; p.last_el = b.last_element[k];
    p.flags = (b.was_element[k] != 0 ? F_WAS_EL : 0u) | (b.reflected[k] != 0 ? F_REFLECTED : 0u);
    p.iter = b.iterations[k];
    p.best_t = real(0);
    set_admissibility_forward(p);
    if ((p.flags & F_WAS_EL) && p.last_el >= 0) {  // side of RayTrace.OldPoint w.r.t. the last operation plane
        const DevElem<real> &le = row(lds.elems, p.last_el);
        real nx = le.nx, ny = le.ny, nz = le.nz;
        double d0 = (double)le.d0;
        if ((le.type >> 8) == RTMC_GEOM_CYLINDER) {  // the photon sits on the tangent plane of its last operation
            cylinder_normal(le, p.Lx, p.Ly, p.Lz, nx, ny, nz);
            d0 = (double)nx * b.loc[0][k] + (double)ny * b.loc[1][k] + (double)nz * b.loc[2][k];
        }
        double s = (double)nx * b.old_point[0][k] + (double)ny * b.old_point[1][k] +
                   (double)nz * b.old_point[2][k] - d0;
        set_admissibility(p, le.type & 0xFF, (s > 0) - (s < 0), nx * p.Dx + ny * p.Dy + nz * p.Dz);
        // uncertainty of the landing point: the segment old point -> here met the plane under |s| / segment
        const real reach = real(s > 0 ? s : -s);
        const bool oblique = (p.hitd > real(0)) & (reach > real(0));
        p.guard = landing_uncertainty(p, p.hitd, oblique ? p.hitd : real(1), oblique ? reach : real(1));
    }
}

template <typename real>
__device__ __forceinline__ void soa_store(const SoaDev &b, uint32_t k, const Photon<real> &p, uint32_t tape_pos,
                                          bool with_hit_distance)
{
    b.loc[0][k] = p.Lx; b.loc[1][k] = p.Ly; b.loc[2][k] = p.Lz;
    b.dir[0][k] = p.Dx; b.dir[1][k] = p.Dy; b.dir[2][k] = p.Dz;
    b.efield[0][k] = p.Ex; b.efield[1][k] = p.Ey; b.efield[2][k] = p.Ez;
    b.old_point[0][k] = p.oldx; b.old_point[1][k] = p.oldy; b.old_point[2][k] = p.oldz;
    b.dist_left[k] = p.dist; b.psi[k] = p.psi;
    if (with_hit_distance) b.hit_distance[k] = p.hitd;
    b.wid[k] = p.wid; b.wid0[k] = p.wid0; b.material[k] = p.mat; b.status[k] = p.status;
    b.last_element[k] = p.last_el; b.was_element[k] = (p.flags & F_WAS_EL) != 0u; b.reflected[k] = (p.flags & F_REFLECTED) != 0u;
    b.iterations[k] = p.iter; b.tape_pos[k] = (int32_t)tape_pos;
}

// rtmc_step_tape: one lane per photon, variates from the tape.
template <typename real>
__global__ void __launch_bounds__(TRACE_BLOCK)
step_tape_kernel(const DevScene<real> *scene, SoaDev b, uint32_t n, uint32_t n_steps, int launch,
                 const double *tape, const unsigned long long *tape_off, double *trace, int *overrun,
                 int32_t *cross_el, double *cross_dist, double *cross_pts)
{
    extern __shared__ __align__(16) unsigned char smem[];
    SceneArg<real> sc = scene_ref(scene);
    Lds<real> lds = stage_lds(sc, smem, false);
    const HotScene<real> hot = hot_scene(sc);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Rng<real, true> g;
    g.tape = tape + tape_off[k];
    g.end = (uint32_t)(tape_off[k + 1] - tape_off[k]);
    g.pos = launch ? 0u : (uint32_t)b.tape_pos[k];
    g.overrun = false;
    g.uniforms_only = false;
    Photon<real> p;
    if (launch) launch_in_reference_order(sc, lds, p, g);
    else soa_load(b, k, lds, p);
    if (cross_el) {  // rtmc_op_crossing: the search only
        real dnorm;
        const Nearest<real> nr = nearest_valid_hit<FEAT_ALL>(hot, lds, p, dnorm);
        const bool none = nr.e < 0;
        cross_el[k] = nr.e;
        cross_dist[k] = none ? __builtin_huge_val() : (double)(M<real>::abs(nr.t) * dnorm);
        cross_pts[3 * k] = none ? 0.0 : (double)(p.Lx + nr.t * p.Dx);
        cross_pts[3 * k + 1] = none ? 0.0 : (double)(p.Ly + nr.t * p.Dy);
        cross_pts[3 * k + 2] = none ? 0.0 : (double)(p.Lz + nr.t * p.Dz);
        return;
    }
    NoTally tally;
    bool alive = !(p.status == RTMC_ST_INTERNAL_CONVERTED || p.status == RTMC_ST_DUMPED ||
                   p.status == RTMC_ST_LOST || p.iter >= sc.max_iter);
    for (uint32_t it = 0; it < n_steps && alive; it++) {
        StepResult r = photon_step<FEAT_ALL, false>(hot, lds, p, g, tally);
        if (r == STEP_ABSORBED) {
            emit_in_reference_order(hot, lds, p, g);
            r = finish_absorbed_iteration(hot, p);
        }
        alive = r == STEP_CONTINUE;
        if (trace) {
            double *r = trace + ((size_t)k * n_steps + it) * RTMC_TRACE_STRIDE;
            r[0] = p.Lx; r[1] = p.Ly; r[2] = p.Lz; r[3] = p.Dx; r[4] = p.Dy; r[5] = p.Dz;
            r[6] = p.Ex; r[7] = p.Ey; r[8] = p.Ez;
            r[9] = sc.wl[p.wid]; r[10] = p.dist; r[11] = p.status; r[12] = p.mat;
            r[13] = p.last_el; r[14] = p.hitd; r[15] = g.pos;
        }
    }
    if (g.overrun) *overrun = 1;
    soa_store(b, k, p, g.pos, true);
}

// rtmc_op_photon: one lane per photon, one operation.
template <typename real>
__global__ void __launch_bounds__(TRACE_BLOCK)
photon_op_kernel(const DevScene<real> *scene, SoaDev b, uint32_t n, int op, const double *op_args, double *op_out,
                 const double *tape, const unsigned long long *tape_off, int *overrun, int uniform_tape)
{
    extern __shared__ __align__(16) unsigned char smem[];
    SceneArg<real> sc = scene_ref(scene);
    Lds<real> lds = stage_lds(sc, smem, false);
    const HotScene<real> hot = hot_scene(sc);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const bool launch = op == RTMC_PHOP_GENERATE;
    Rng<real, true> g;
    g.tape = tape + tape_off[k];
    g.end = (uint32_t)(tape_off[k + 1] - tape_off[k]);
    g.pos = launch ? 0u : (uint32_t)b.tape_pos[k];
    g.overrun = false;
    g.uniforms_only = uniform_tape != 0;
    Photon<real> p;
    if (launch) {
        launch_in_reference_order(sc, lds, p, g);
    } else {
        soa_load(b, k, lds, p);
        photon_operation(hot, lds, p, g, op, op_args + (size_t)k * RTMC_PHOP_ARGS, b.hit_distance[k],
                         op_out ? op_out + 2 * (size_t)k : nullptr);
    }
    if (g.overrun) *overrun = 1;
    soa_store(b, k, p, g.pos, launch);
}

// rtmc_op_intersect: one lane per (ray, element)
template <typename real>
__global__ void intersect_kernel(const DevScene<real> *scene, uint32_t n, const double *loc, const double *dir,
                                 double *out)
{
    SceneArg<real> sc = scene_ref(scene);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * sc.n_el) return;
    const uint32_t k = (uint32_t)(idx / sc.n_el);
    const int e = (int)(idx % sc.n_el);
    real Lx = real(loc[3 * k]), Ly = real(loc[3 * k + 1]), Lz = real(loc[3 * k + 2]);
    real Dx = real(dir[3 * k]), Dy = real(dir[3 * k + 1]), Dz = real(dir[3 * k + 2]);
    real t;
    const real dnorm = M<real>::sqrt(Dx * Dx + Dy * Dy + Dz * Dz);
    bool ok = shape_hit(load_uniform(sc.elems + e), Lx, Ly, Lz, Dx, Dy, Dz, false, false, dnorm, t);
    const double nan = __builtin_nan("");
    out[3 * idx] = ok ? (double)(Lx + t * Dx) : nan;
    out[3 * idx + 1] = ok ? (double)(Ly + t * Dy) : nan;
    out[3 * idx + 2] = ok ? (double)(Lz + t * Dz) : nan;
}

// Adds the workgroups' private detector histograms (HBM) into the tally block.
__global__ void __launch_bounds__(256)
sum_private_hist_kernel(const uint32_t *priv, uint32_t n_copies, uint32_t words, unsigned long long *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    unsigned long long total = 0;
    for (uint32_t c = 0; c < n_copies; c++) total += priv[(size_t)c * words + i];
    out[i] += total;
}

}  // namespace

// ---------------------------------------------------------------------------
// host: errors
// ---------------------------------------------------------------------------
// fixed storage: reporting a failure allocates nothing
static thread_local char g_error[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

// Nothing may propagate out of an entry point: allocation failures and
// anything else thrown by the host code become error codes.
template <typename Body>
static int guarded(Body &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RTMC_ENOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(RTMC_EINVAL, "%s", e.what());
    } catch (...) {
        return fail(RTMC_EINVAL, "unexpected exception");
    }
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(RTMC_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// ---------------------------------------------------------------------------
// host: scene
// ---------------------------------------------------------------------------
namespace {

struct HostTable {
    std::vector<double> x, y;
    bool same(const HostTable &o) const { return x == o.x && y == o.y; }
};

// nearest entry of an ascending table, RTUtils.py:15-22
int64_t nearest_index(const std::vector<double> &A, double t)
{
    int64_t n = (int64_t)A.size();
    int64_t idx = std::lower_bound(A.begin(), A.end(), t) - A.begin();
    idx = std::min<int64_t>(std::max<int64_t>(idx, 1), n - 1);
    if (t - A[idx - 1] < A[idx] - t) idx--;
    return idx;
}

// interval-weighted cumulative distribution of an emission table:
// RTUtils.py:152-158 then the cumulative sum numpy.random.choice searches
std::vector<double> sampling_cdf(const HostTable &t, bool times_lambda)
{
    size_t n = t.x.size();
    std::vector<double> w(n);
    double total = 0;
    for (size_t i = 0; i < n; i++) {
        double v = times_lambda ? t.y[i] * t.x[i] : t.y[i];
        if (i) v *= t.x[i] - t.x[i - 1];
        w[i] = v;
        total += v;
    }
    double run = 0;
    for (size_t i = 0; i < n; i++) { run += w[i] / total; w[i] = run; }
    for (size_t i = 0; i < n; i++) w[i] /= run;
    return w;
}

// in-plane unit axes n1 = unit(v x x^), n2 = unit(v x n1) of MakeVectAxis (RTUtils.py:25-42; y^ if v is x^)
void axis_frame(const double *v, double *n1, double *n2)
{
    const bool is_x = v[0] == 1 && v[1] == 0 && v[2] == 0;
    const double o[3] = {is_x ? 0.0 : 1.0, is_x ? 1.0 : 0.0, 0.0};
    auto cross = [](const double *a, const double *b, double *c) {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto unit = [](double *a) {
        const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        a[0] /= n; a[1] /= n; a[2] /= n;
    };
    cross(v, o, n1); unit(n1);
    cross(v, n1, n2); unit(n2);
}

// u < q  for u = v / 2^32 with integer v   <=>   v < ceil(q * 2^32);
// the same integer serves "cdf entry <= u"  <=>  ceil(cdf * 2^32) <= v.
uint32_t threshold32(double q)
{
    if (!(q > 0)) return 0u;  // also NaN
    double s = std::ceil(q * 4294967296.0);
    return s >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)s;
}

constexpr int MAX_ELEMENTS = 4096;  // beyond what the LDS holds; the real limit is the size of the LDS image
constexpr size_t RUN_MIN_MEMBERS = 6;  // coplanar rectangles of one group from which on they are looked up as a run
constexpr size_t MAX_FRAMES = 64;  // orthonormal frames besides which rectangles take the general test

template <typename real>
struct DeviceImage {
    std::vector<DevElem<real>> elems;
    std::vector<HitChunk<real>> hits;
    int n_group[4];
    bool has_runs = false;           // some axis group lists coplanar rectangles as runs (float32 only)
    int n_runs = 0, n_run_members = 0;
    int n_tiled_faces = 0;           // faces of cells that are tilings looked up through a grid (float32 only)
    std::vector<unsigned char> lds;  // elems | mats | abs | cdf
    LdsLayout lay;
    DevSource<real> src;
    bool back_possible = false;  // some medium's thresholds let the acceptor hand the excitation back
};

// Closed boxes among the rectangles of a scene (see DevCell), found once per scene from its description.
struct CellPlan {
    std::vector<DevCell> cells;          // two records per cell (see DevCell); cell 0 = "no cell"
    std::vector<uint8_t> el_cell;        // per element: the cell it is a face of (0 none)
    std::vector<uint8_t> el_outer;       //              the cell on the outer side of it (0 none)
    std::vector<uint8_t> el_outward;     //              and whether its normal points out of the cell it is a face of
    int start_cell = 0;                  // innermost cell that holds the source
    int n_usable = 0;                    // cells in which some photon can be confined (leaf, simple or region)
    int n_regions = 0;                   // cells that are regions of a box (the last ones)
    // faces that are tilings: per tiled face its rectangles in list order, in the in-plane coordinates of the hit
    // table (axes k+1, k+2 of normal axis k); DevCell::idlo/idhi name list g as -(g + 2)
    struct Tile { double ci, cj, hi, hj; int index; };
    std::vector<std::vector<Tile>> tiles;
    int n_frames_used = 1;               // 1 + the highest frame (FrameSet) some cell lies in
};

struct PerDevice {
    bool ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *elems32 = nullptr, *elems64 = nullptr, *lds32 = nullptr, *lds64 = nullptr;
    void *hits32 = nullptr, *hits64 = nullptr;
    void *scene32 = nullptr, *scene64 = nullptr;  // DevScene<float/double> in device memory
    void *ro_block = nullptr, *rw_block = nullptr; // the two allocations everything above lives in
    double *wl = nullptr;
    uint32_t *eqeT = nullptr;
    unsigned long long *tallies = nullptr;  // scalars | bounce | spectral | energy | hist
    unsigned long long *work = nullptr;
    unsigned long long *host_tallies = nullptr;  // pinned host copy of `tallies`, filled after every launch
    uint32_t *private_hist = nullptr;            // per-workgroup detector histograms of scenes that count in HBM
    size_t private_hist_copies = 0;
    size_t tally_words = 0;
    double last_ms = 0;
    int max_blocks32 = 0, max_blocks64 = 0, n_cu = 0;
    int block32 = TRACE_BLOCK;  // threads per workgroup of the float32 trace kernel for this scene
};

}  // namespace

// ---------------------------------------------------------------------------
// per-GPU state shared by every scene of the process
// ---------------------------------------------------------------------------
// Static device memory for rtmc_run_noalloc: the tally block the trace kernel
// flushes into (at most SCRATCH_TALLY_WORDS 64-bit words) followed by the work
// counter in a 4 KB piece of its own (tallies and counter are hit by atomics at
// very different rates and must not share a cache line).  Every GPU the
// library's code object is loaded on holds its own copy of the array at its own
// address; host code never uses the array's name as an address, only the
// pointer hipGetSymbolAddress returned for that GPU.
constexpr size_t SCRATCH_TALLY_WORDS = 65536;                 // 512 KB
constexpr size_t SCRATCH_WORK_OFFSET = SCRATCH_TALLY_WORDS;   // in words
constexpr size_t SCRATCH_WORDS = SCRATCH_TALLY_WORDS + 512;
static __device__ __align__(256) unsigned long long scratch[SCRATCH_WORDS];

constexpr int MAX_DEVICES = 64;
namespace {
struct GpuShared {
    // Held from the first memset of a launch to the end of its copy-back: the
    // calls of one process on one GPU run one after the other (they share that
    // GPU's scratch array, and the calls on one scene share its tally buffers,
    // stream and events).  Calls on different GPUs do not wait for each other.
    std::mutex launch_mu;
    unsigned long long *scratch_ptr = nullptr;  // device address of `scratch` on this GPU
};
GpuShared g_gpu[MAX_DEVICES];
std::mutex g_gpu_setup_mu;
}  // namespace

// Once per GPU, with that GPU current: the device address of its copy of `scratch`.
static int setup_gpu_shared(int device)
{
    std::lock_guard<std::mutex> lock(g_gpu_setup_mu);
    if (g_gpu[device].scratch_ptr) return RTMC_OK;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RTMC_EDEVICE, "hipGetSymbolAddress returned a null address for the scratch array on device %d", device);
    g_gpu[device].scratch_ptr = (unsigned long long *)ptr;
    return RTMC_OK;
}

struct rtmc_scene {
    int n_el = 0, n_mat = 0, n_eqe = 0, max_iter = 0;
    bool has_aligned = false;       // some medium has aligned acceptor dyes: the kernels with that code are used
    bool has_shapes = false;        // some element is not a rectangle, or is a lens
    bool has_area = false;          // the source spreads its launch points: the kernel with every feature is used
    bool has_spectra = false;       // spectrum source or a detector with an efficiency table
    bool has_back = false;          // some medium's acceptor can hand the excitation back to its donor
    bool has_frames = false;        // some rectangle is tested in a frame other than the world's
    bool has_coplanar = false;      // two elements overlap in one plane: the float32 kernels keep a guard distance
    bool seam_source = false;       // a strict source is aimed at a seam: guard distance until a photon's first absorption
    bool hist_in_hbm = false;       // detectors x bins too large for the LDS in some precision
    bool has_runs = false;          // the float32 hit table lists coplanar rectangles as runs
    bool has_cells = false;         // some rectangles close boxes in which photons can be confined (float32)
    bool has_tiles = false;         // some face of such a box is a tiling of several rectangles
    int n_cells = 0, n_regions = 0, n_cells_usable = 0, start_cell = 0;   // (n_cells counts the regions too)
    int guide_bits = 0;             // bucket bits of the guide tables in the float32 image
    int features() const
    {
        if (has_area || has_coplanar || seam_source || (hist_in_hbm && !cells_in_use())) return FEAT_ALL;
        return (has_aligned ? FEAT_ALIGNED : 0) | (has_shapes ? FEAT_SHAPES | FEAT_FRAMES : 0) |
               (has_spectra ? FEAT_SPECTRA : 0) | (has_back ? FEAT_BACK : 0) | (has_frames ? FEAT_FRAMES : 0) |
               ((has_runs || ((has_tiles || hist_in_hbm || has_frames) && cells_in_use())) ? FEAT_RUNS : 0) |
               (cells_in_use() ? FEAT_CELLS : 0) | (hist_in_hbm ? FEAT_HBMHIST : 0);
    }
    // the kernels with cells exist for rectangle-only scenes; those that also hold the code for runs of coplanar
    // rectangles, tiled faces, turned frames or the detector histogram in HBM, for isotropic dyes only
    bool cells_in_use() const
    {
        return has_cells && !has_shapes && !((has_runs || has_tiles || hist_in_hbm || has_frames) && has_aligned);
    }
    std::vector<rtmc_element> elements;
    std::vector<double> wl;         // wavelength of each id
    std::vector<uint32_t> eqeT;     // [n_eqe][n_wid]
    std::vector<uint32_t> energy_fx;   // single-line source: energy increment of each wavelength id
    std::vector<uint16_t> wid_cbin;    // compact detector bin of each wavelength id
    std::vector<int32_t> bins_used;    // compact bin -> 1-nm bin index (0 = 200 nm)
    std::vector<int32_t> el_slot;      // detector slot of each element, -1 if not a detector
    int n_dump = 0;
    std::vector<int32_t> el_eqe;    // per element: eqe table or -1
    DeviceImage<float> img32;
    DeviceImage<double> img64;
    mutable std::mutex mu;  // guards `dev`
    std::map<int, PerDevice> dev;
};

namespace {

// The orthonormal frames of a scene's rectangles.  A rectangle whose normal and in-plane axes are
// mutually perpendicular unit vectors is axis-aligned in the frame whose axes they are (up to order
// and sign); the world frame is frame 0 and is matched exactly, further frames within 1e-12, in the
// order in which the element list first needs them.
struct FrameSet {
    struct Place { int frame = -1, kn = -1, k1 = -1, k2 = -1; };   // frame and the rows of it that are normal, e1, e2
    std::vector<std::array<double, 9>> r;   // per frame its axes as rows, in world coordinates
    std::vector<Place> place;               // per element; frame -1: tested on its full record
    double coord(int frame, int axis, const double *x) const
    {   // frame 0: the coordinate itself, so that an axis-aligned scene is described exactly
        const double *row = &r[frame][3 * axis];
        return frame == 0 ? x[axis] : row[0] * x[0] + row[1] * x[1] + row[2] * x[2];
    }
};

FrameSet assign_frames(const rtmc_scene_desc *d)
{
    FrameSet fs;
    fs.r.push_back({1, 0, 0, 0, 1, 0, 0, 0, 1});
    fs.place.resize(d->n_elements);
    auto dot3 = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    // row of frame f that v is (plus or minus); -1 if none.  Frame 0 wants the exact unit vector.
    auto row_of = [&](const std::array<double, 9> &f, const double *v, bool exact) {
        for (int a = 0; a < 3; a++) {
            const double c = std::fabs(dot3(&f[3 * a], v));
            double off = 0;
            for (int j = 0; j < 3; j++) off = std::max(off, std::fabs(std::fabs(v[j]) - std::fabs(f[3 * a + j])));
            if (exact ? (c == 1.0 && off == 0.0) : (std::fabs(c - 1.0) <= 1e-12 && off <= 1e-12)) return a;
        }
        return -1;
    };
    for (int i = 0; i < d->n_elements; i++) {
        const rtmc_element &e = d->elements[i];
        const bool orthonormal = e.geometry == RTMC_GEOM_RECTANGLE &&
                                 std::fabs(dot3(e.normal, e.normal) - 1) <= 1e-12 && std::fabs(dot3(e.e1, e.e1) - 1) <= 1e-12 &&
                                 std::fabs(dot3(e.e2, e.e2) - 1) <= 1e-12 && std::fabs(dot3(e.normal, e.e1)) <= 1e-12 &&
                                 std::fabs(dot3(e.normal, e.e2)) <= 1e-12 && std::fabs(dot3(e.e1, e.e2)) <= 1e-12;
        if (!orthonormal) continue;
        size_t fi = 0;
        int kn = -1, k1 = -1, k2 = -1;
        for (; fi < fs.r.size(); fi++) {
            kn = row_of(fs.r[fi], e.normal, fi == 0);
            k1 = row_of(fs.r[fi], e.e1, fi == 0);
            k2 = row_of(fs.r[fi], e.e2, fi == 0);
            if (kn >= 0 && k1 >= 0 && k2 >= 0 && kn != k1 && kn != k2 && k1 != k2) break;
        }
        if (fi == fs.r.size() && fs.r.size() < MAX_FRAMES) {
            fs.r.push_back({e.e1[0], e.e1[1], e.e1[2], e.e2[0], e.e2[1], e.e2[2], e.normal[0], e.normal[1], e.normal[2]});
            k1 = 0; k2 = 1; kn = 2;
        }
        if (fi < fs.r.size()) {
            fs.place[i].frame = (int)fi; fs.place[i].kn = kn; fs.place[i].k1 = k1; fs.place[i].k2 = k2;
        }
    }
    return fs;
}

// Finds the cells of a scene: rectangles of one orthonormal frame (FrameSet) that close a box that
// is axis-aligned in that frame.  A face of the box is one rectangle that spans exactly the box, or
// several that tile it without gap or overlap (a read-out cut into strips); at least one pair of
// opposite faces must be whole rectangles.  A cell is usable for the confined search if nothing but
// its faces comes near its box, or if everything that does belongs to one cell nested in it (see
// DevCell).  Lenses are no faces (a photon passes them without the refraction that moves it from
// cell to cell).
CellPlan find_cells(const rtmc_scene_desc *d, const FrameSet &fs, bool with_regions)
{
    CellPlan plan;
    const int n_el = d->n_elements;
    const int n_frames = (int)fs.r.size();
    plan.el_cell.assign(n_el, 0);
    plan.el_outer.assign(n_el, 0);
    plan.el_outward.assign(n_el, 0);
    DevCell none;
    memset(&none, 0, sizeof none);
    for (int k = 0; k < 3; k++) none.hw[k] = -1.0f;   // no photon is in it
    plan.cells.assign(2, none);
    if (getenv("RTMC_NO_CELLS")) return plan;

    struct Face { int el, axis; double p, c[3], h[3]; };
    std::vector<std::vector<Face>> faces(3 * (size_t)n_frames);   // [3 frame + axis]
    std::vector<int> face_slot(n_el, -1);   // position of an element in its list of faces
    double scale = 1.0;
    for (int i = 0; i < n_el; i++) {
        const rtmc_element &e = d->elements[i];
        for (int a = 0; a < 3; a++) scale = std::max(scale, std::fabs(e.support[a]));
        const FrameSet::Place &pl = fs.place[i];
        if (pl.frame < 0 || e.type == RTMC_TYPE_LENS) continue;
        if (!(e.half1 > 0) || !(e.half2 > 0)) continue;
        Face f;
        f.el = i; f.axis = pl.kn;
        for (int a = 0; a < 3; a++) { f.c[a] = fs.coord(pl.frame, a, e.support); f.h[a] = 0; }
        f.p = f.c[pl.kn];
        f.h[pl.k1] = e.half1; f.h[pl.k2] = e.half2;
        std::vector<Face> &list = faces[3 * (size_t)pl.frame + pl.kn];
        face_slot[i] = (int)list.size();
        list.push_back(f);
    }
    const double tol = 1e-9 * scale;
    struct Box { int frame; double lo[3], hi[3]; std::vector<int> face[3][2]; bool dropped = false; int parent = -1; };
    std::vector<Box> boxes;
    auto near = [&](double a, double b) { return std::fabs(a - b) <= tol; };
    // the rectangles of `axis` of a frame at plane `p` that make up the face lo..hi: one that spans it exactly, or
    // several that lie inside it, do not overlap and add up to its area; empty if there is no such set
    auto find_face = [&](int frame, int axis, double p, const double *lo, const double *hi) {
        std::vector<int> inside;
        const std::vector<Face> &list = faces[3 * (size_t)frame + axis];
        const int i = (axis + 1) % 3, j = (axis + 2) % 3;
        for (const Face &f : list) {
            if (!near(f.p, p)) continue;
            if (f.c[i] - f.h[i] < lo[i] - tol || f.c[i] + f.h[i] > hi[i] + tol || f.c[j] - f.h[j] < lo[j] - tol ||
                f.c[j] + f.h[j] > hi[j] + tol) continue;
            inside.push_back(f.el);
        }
        if (inside.empty() || inside.size() > 255) return std::vector<int>();
        double area = 0;
        for (size_t a = 0; a < inside.size(); a++) {
            const Face &fa = list[face_slot[inside[a]]];
            area += 4.0 * fa.h[i] * fa.h[j];
            for (size_t b = a + 1; b < inside.size(); b++) {   // interiors must not overlap
                const Face &fb = list[face_slot[inside[b]]];
                if (std::fabs(fa.c[i] - fb.c[i]) < fa.h[i] + fb.h[i] - tol && std::fabs(fa.c[j] - fb.c[j]) < fa.h[j] + fb.h[j] - tol)
                    return std::vector<int>();
            }
        }
        const double whole = (hi[i] - lo[i]) * (hi[j] - lo[j]);
        if (std::fabs(area - whole) > 1e-9 * whole) return std::vector<int>();
        return inside;
    };
    for (int frame = 0; frame < n_frames; frame++)
        for (int ax = 0; ax < 3; ax++) {
            const int i = (ax + 1) % 3, j = (ax + 2) % 3;
            const std::vector<Face> &list = faces[3 * (size_t)frame + ax];
            for (const Face &A : list) {
                // the nearest face above A with the same outline closes the box along this axis
                const Face *B = nullptr;
                for (const Face &o : list)
                    if (o.p > A.p + tol && near(o.c[i], A.c[i]) && near(o.c[j], A.c[j]) && near(o.h[i], A.h[i]) && near(o.h[j], A.h[j]) &&
                        (!B || o.p < B->p)) B = &o;
                if (!B) continue;
                Box b;
                b.frame = frame;
                b.lo[ax] = A.p; b.hi[ax] = B->p;
                b.lo[i] = A.c[i] - A.h[i]; b.hi[i] = A.c[i] + A.h[i];
                b.lo[j] = A.c[j] - A.h[j]; b.hi[j] = A.c[j] + A.h[j];
                bool known = false;
                for (const Box &o : boxes) {
                    bool same = o.frame == frame;
                    for (int k = 0; k < 3; k++) same = same && near(o.lo[k], b.lo[k]) && near(o.hi[k], b.hi[k]);
                    known = known || same;
                }
                if (known) continue;
                bool closed = true;
                for (int a = 0; a < 3 && closed; a++) {
                    b.face[a][0] = find_face(frame, a, b.lo[a], b.lo, b.hi);
                    b.face[a][1] = find_face(frame, a, b.hi[a], b.lo, b.hi);
                    closed = !b.face[a][0].empty() && !b.face[a][1].empty();
                }
                if (closed && boxes.size() < 127) boxes.push_back(b);   // (the number of a box takes 7 bits of DevElem::info)
            }
        }
    // a rectangle that is (part of) a face of two boxes (two boxes sharing a wall) would have to move a photon
    // between three cells: neither box becomes a cell
    {
        std::vector<int> uses(n_el, 0);
        for (const Box &b : boxes)
            for (int a = 0; a < 3; a++)
                for (int side = 0; side < 2; side++)
                    for (int e : b.face[a][side]) uses[e]++;
        for (Box &b : boxes)
            for (int a = 0; a < 3; a++)
                for (int side = 0; side < 2; side++)
                    for (int e : b.face[a][side])
                        if (uses[e] > 1) b.dropped = true;
        boxes.erase(std::remove_if(boxes.begin(), boxes.end(), [](const Box &b) { return b.dropped; }), boxes.end());
    }
    if (boxes.empty()) return plan;
    auto extent = [](const Box &b) { return std::max(b.hi[0] - b.lo[0], std::max(b.hi[1] - b.lo[1], b.hi[2] - b.lo[2])); };
    // b strictly inside a, by a's margin; the two may lie in different frames (a turned slab in an upright box): the
    // corners of b are taken into a's frame
    auto inside = [&](const Box &b, const Box &a) {
        const double m = 1e-4 * extent(a);
        for (int corner = 0; corner < 8; corner++) {
            const double q[3] = {corner & 1 ? b.hi[0] : b.lo[0], corner & 2 ? b.hi[1] : b.lo[1], corner & 4 ? b.hi[2] : b.lo[2]};
            double world[3] = {0, 0, 0};
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) world[c] += q[k] * fs.r[b.frame][3 * k + c];
            for (int k = 0; k < 3; k++) {
                const double x = a.frame == b.frame ? q[k] : fs.coord(a.frame, k, world);
                if (!(a.lo[k] + m <= x && x + m <= a.hi[k])) return false;
            }
        }
        return true;
    };
    auto volume = [](const Box &b) { return (b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]); };
    const int n_box = (int)boxes.size();
    for (int i = 0; i < n_box; i++)
        for (int j = 0; j < n_box; j++)
            if (i != j && inside(boxes[i], boxes[j]) && (boxes[i].parent < 0 || volume(boxes[j]) < volume(boxes[boxes[i].parent])))
                boxes[i].parent = j;
    std::vector<int> face_of(n_el, -1);
    for (int i = 0; i < n_box; i++)
        for (int a = 0; a < 3; a++)
            for (int side = 0; side < 2; side++)
                for (int e : boxes[i].face[a][side]) face_of[e] = i;
    // bounding box of any element in the coordinates of a frame
    auto bounds = [&](const rtmc_element &e, int frame, double *lo, double *hi) {
        auto along = [&](int a, const double *v) {   // component of a direction along axis a of the frame
            const double zero[3] = {0, 0, 0};
            return fs.coord(frame, a, v) - fs.coord(frame, a, zero);
        };
        for (int a = 0; a < 3; a++) {
            if (e.geometry == RTMC_GEOM_TRIANGLE) {
                const double q[3] = {fs.coord(frame, a, e.e1), fs.coord(frame, a, e.e2), fs.coord(frame, a, e.vertex3)};
                lo[a] = std::min(q[0], std::min(q[1], q[2])); hi[a] = std::max(q[0], std::max(q[1], q[2]));
            } else if (e.geometry == RTMC_GEOM_CYLINDER) {   // support and normal are the end points of the axis
                const double q[2] = {fs.coord(frame, a, e.support), fs.coord(frame, a, e.normal)};
                lo[a] = std::min(q[0], q[1]) - e.half1; hi[a] = std::max(q[0], q[1]) + e.half1;
            } else {
                const double r = e.geometry == RTMC_GEOM_CIRCLE ? e.half1
                                                                : std::fabs(along(a, e.e1)) * e.half1 + std::fabs(along(a, e.e2)) * e.half2;
                const double c = fs.coord(frame, a, e.support);
                lo[a] = c - r; hi[a] = c + r;
            }
        }
    };
    plan.cells.resize(2 * (1 + n_box), none);
    std::vector<DevCell> box_record(n_box, none);   // the record of each box as a leaf would use it
    std::vector<char> box_fits(n_box, 0), box_clean(n_box, 0);
    for (int i = 0; i < n_box; i++) {
        const Box &b = boxes[i];
        const double m = 1e-4 * extent(b);
        bool dirty = false;
        int children = 0;
        for (int j = 0; j < n_box; j++)
            if (boxes[j].parent == i) children++;
        for (int e = 0; e < n_el && !dirty; e++) {
            if (face_of[e] == i) continue;
            double lo[3], hi[3];
            bounds(d->elements[e], b.frame, lo, hi);
            bool apart = false;
            for (int k = 0; k < 3; k++) apart = apart || hi[k] < b.lo[k] - m || lo[k] > b.hi[k] + m;
            if (apart) continue;
            // it comes near this box: fine only as (part of) a cell nested in this one
            int owner = face_of[e];
            while (owner >= 0 && boxes[owner].parent != i) owner = boxes[owner].parent;
            if (owner < 0) dirty = true;
        }
        bool fits = true;
        DevCell c;
        memset(&c, 0, sizeof c);
        c.frame = b.frame;
        plan.n_frames_used = std::max(plan.n_frames_used, b.frame + 1);
        for (int k = 0; k < 3; k++) {
            int32_t *ids[2] = {&c.idlo[k], &c.idhi[k]};
            float *plane[2] = {&c.plo[k], &c.phi[k]};
            for (int side = 0; side < 2; side++) {
                const std::vector<int> &f = b.face[k][side];
                // (the plane coordinate as the hit table holds it, HitRec::p0)
                *plane[side] = (float)fs.coord(b.frame, k, d->elements[f[0]].support);
                for (int e : f)   // (tiles of one plane must share its float32 coordinate, as the hit table's runs do)
                    if ((float)fs.coord(b.frame, k, d->elements[e].support) != *plane[side]) fits = false;
                if (f.size() == 1) {
                    *ids[side] = f[0];
                } else {   // a tiled face: its rectangles as the hit table describes them, looked up through a grid
                    *ids[side] = -((int32_t)plan.tiles.size() + 2);
                    const int ai = (k + 1) % 3, aj = (k + 2) % 3;
                    std::vector<CellPlan::Tile> list;
                    for (int e : f) {
                        const Face &fe = faces[3 * (size_t)b.frame + k][face_slot[e]];
                        list.push_back({fe.c[ai], fe.c[aj], fe.h[ai], fe.h[aj], e});
                    }
                    std::sort(list.begin(), list.end(), [](const CellPlan::Tile &x, const CellPlan::Tile &y) { return x.index < y.index; });
                    plan.tiles.push_back(list);
                }
            }
            c.c[k] = (float)(0.5 * (b.lo[k] + b.hi[k]));
            c.hw[k] = (float)(0.5 * (b.hi[k] - b.lo[k]) + 1e-5 * extent(b));
            if (!(c.plo[k] < c.phi[k])) fits = false;   // (a box float32 cannot resolve)
        }
        DevCell unusable = c;
        for (int k = 0; k < 3; k++) unusable.hw[k] = -1.0f;
        const bool leaf = fits && !dirty && children == 0, simple = fits && !dirty && children == 1;
        plan.cells[2 * (1 + i)] = leaf ? c : unusable;
        plan.cells[2 * (1 + i) + 1] = (leaf || simple) ? c : unusable;
        if (leaf || simple) plan.n_usable++;
        box_record[i] = c; box_fits[i] = fits; box_clean[i] = !dirty;
        for (int k = 0; k < 3; k++)
            for (int side = 0; side < 2; side++)
                for (int e : b.face[k][side]) {
                    plan.el_cell[e] = (uint8_t)(1 + i);
                    plan.el_outer[e] = (uint8_t)(b.parent < 0 ? 0 : 1 + b.parent);
                    // does the element's normal point out of the box?
                    const double *axis = &fs.r[b.frame][3 * k], *nrm = d->elements[e].normal;
                    const double along = axis[0] * nrm[0] + axis[1] * nrm[1] + axis[2] * nrm[2];
                    plan.el_outward[e] = (side == 1) == (along > 0);
                }
    }
    // A box that holds several others (a stack of slabs in a detector box) confines nobody as a whole, but the
    // space in front of a face of a nested box K does: the face pushed outwards, through empty space, up to the
    // first thing it meets -- the wall of the enclosing box P or the facing side of another nested box.  Such a
    // *region* becomes a cell of its own, named as the cell on the outer side of that face.  Its near side is the
    // face; its far side is the wall or the face it ends on, if that covers the whole cross-section (otherwise,
    // like its four other sides, it is open: the search from the cell gives no answer for a photon that leaves
    // through an open side, CONFINED_NONE, and the list walk takes over).  On the open sides the record's half
    // extents are drawn in by the margin, so that whether a landing point near the outline of the far face lies
    // on that rectangle or beside it is left to the list walk's own test of the rectangle.
    struct Region { int frame, parent; double lo[3], hi[3]; };
    std::vector<Region> regions;
    for (int i = 0; i < n_box; i++) {
        if (!with_regions || !box_fits[i] || !box_clean[i]) continue;
        const Box &P = boxes[i];
        std::vector<int> kids;
        for (int j = 0; j < n_box; j++)
            if (boxes[j].parent == i) kids.push_back(j);
        if (kids.size() < 2) continue;
        const double m = 1e-4 * extent(P);
        bool regular = true;   // all in P's frame, and clear of each other along some axis
        for (size_t a = 0; a < kids.size() && regular; a++) {
            const Box &A = boxes[kids[a]];
            regular = A.frame == P.frame && box_fits[kids[a]];
            for (size_t b = a + 1; b < kids.size() && regular; b++) {
                const Box &B = boxes[kids[b]];
                bool apart = false;
                for (int k = 0; k < 3; k++) apart = apart || A.hi[k] + m <= B.lo[k] || B.hi[k] + m <= A.lo[k];
                regular = apart;
            }
        }
        if (!regular) continue;
        for (int j : kids)
            for (int a = 0; a < 3; a++)
                for (int side = 0; side < 2; side++) {
                    const Box &K = boxes[j];
                    const int i1 = (a + 1) % 3, i2 = (a + 2) % 3;
                    const double p0 = side ? K.hi[a] : K.lo[a];
                    double q = side ? P.hi[a] : P.lo[a];
                    int far_box = i, far_side = side;
                    bool far_whole = true;
                    for (int o : kids) {
                        if (o == j) continue;
                        const Box &O = boxes[o];
                        // (a box whose outline comes within the margin of K's, seen along the axis, counts as in the way)
                        if (!(O.lo[i1] < K.hi[i1] + m && O.hi[i1] > K.lo[i1] - m && O.lo[i2] < K.hi[i2] + m && O.hi[i2] > K.lo[i2] - m))
                            continue;
                        const double plane = side ? O.lo[a] : O.hi[a];
                        if (side ? plane < p0 : plane > p0) continue;   // on the other side of K
                        const bool covers = O.lo[i1] <= K.lo[i1] + tol && O.hi[i1] >= K.hi[i1] - tol && O.lo[i2] <= K.lo[i2] + tol &&
                                            O.hi[i2] >= K.hi[i2] - tol;
                        if (near(plane, q)) far_whole = false;   // two boxes end the region in one plane
                        else if (side ? plane < q : plane > q) { q = plane; far_box = o; far_side = 1 - side; far_whole = covers; }
                    }
                    Region r;
                    r.frame = P.frame; r.parent = i;
                    for (int k = 0; k < 3; k++) { r.lo[k] = K.lo[k]; r.hi[k] = K.hi[k]; }
                    r.lo[a] = std::min(p0, q); r.hi[a] = std::max(p0, q);
                    const DevCell &kr = box_record[j], &fr = box_record[far_box];
                    DevCell c;
                    memset(&c, 0, sizeof c);
                    c.frame = P.frame;
                    const float near_plane = side ? kr.phi[a] : kr.plo[a], far_plane = far_side ? fr.phi[a] : fr.plo[a];
                    const int32_t near_id = side ? kr.idhi[a] : kr.idlo[a];
                    const int32_t far_id = far_whole ? (far_side ? fr.idhi[a] : fr.idlo[a]) : CONFINED_NONE;
                    c.plo[a] = side ? near_plane : far_plane; c.idlo[a] = side ? near_id : far_id;
                    c.phi[a] = side ? far_plane : near_plane; c.idhi[a] = side ? far_id : near_id;
                    const double margin = 1e-5 * extent(P);
                    c.c[a] = (float)(0.5 * (r.lo[a] + r.hi[a]));
                    c.hw[a] = (float)(0.5 * (r.hi[a] - r.lo[a]) + margin);
                    bool usable = c.plo[a] < c.phi[a];
                    for (int k : {i1, i2}) {
                        c.plo[k] = (float)K.lo[k]; c.phi[k] = (float)K.hi[k];
                        c.idlo[k] = c.idhi[k] = CONFINED_NONE;
                        c.c[k] = (float)(0.5 * (K.lo[k] + K.hi[k]));
                        c.hw[k] = (float)(0.5 * (K.hi[k] - K.lo[k]) - margin);
                        usable = usable && c.plo[k] < c.phi[k] && c.hw[k] > 0.0f;
                    }
                    if (!usable || !far_whole) continue;   // (open at the far end too: hardly a search would succeed)
                    // A photon that moves away from the face (F_OUTBOUND) gets a wider record if no nested box reaches
                    // beyond the plane of the face: all of P on that side of the plane is then empty, its five other
                    // sides are walls of P, and the side the photon comes from is one it cannot meet.
                    DevCell out = c;
                    bool outermost = far_box == i;
                    for (int o : kids) outermost = outermost && (side ? boxes[o].hi[a] <= p0 + tol : boxes[o].lo[a] >= p0 - tol);
                    if (outermost) {
                        out = box_record[i];
                        (side ? out.plo[a] : out.phi[a]) = near_plane;
                        (side ? out.idlo[a] : out.idhi[a]) = CONFINED_NONE;
                        out.c[a] = c.c[a]; out.hw[a] = c.hw[a];
                    }
                    int number = -1;
                    for (size_t g = 0; g < regions.size() && number < 0; g++) {   // (the gap between two boxes, from either side)
                        bool same = regions[g].frame == r.frame && regions[g].parent == r.parent;
                        for (int k = 0; k < 3; k++) same = same && near(regions[g].lo[k], r.lo[k]) && near(regions[g].hi[k], r.hi[k]);
                        if (same) number = 1 + n_box + (int)g;
                    }
                    if (number < 0) {
                        if (1 + n_box + regions.size() >= 254) continue;   // (cell numbers 1..254; 255 = unknown)
                        number = 1 + n_box + (int)regions.size();
                        regions.push_back(r);
                        plan.cells.push_back(c);
                        plan.cells.push_back(out);
                        plan.n_usable++;
                    }
                    for (int e : K.face[a][side]) plan.el_outer[e] = (uint8_t)number;
                }
    }
    // the innermost box that holds the source, clear of its faces
    int start = -1;
    for (int i = 0; i < n_box; i++) {
        const Box &b = boxes[i];
        const double m = 1e-4 * extent(b);
        bool in = true;
        for (int k = 0; k < 3; k++) {
            const double x = fs.coord(b.frame, k, d->source.location);
            in = in && b.lo[k] + m < x && x < b.hi[k] - m;
        }
        if (in && (start < 0 || volume(b) < volume(boxes[start]))) start = i;
    }
    plan.n_regions = (int)regions.size();
    plan.start_cell = start < 0 ? 0 : 1 + start;
    for (size_t g = 0; g < regions.size(); g++) {   // ... or a region of that box
        const Region &r = regions[g];
        if (r.parent != start) continue;
        const double m = 1e-4 * extent(boxes[start]);
        bool in = true;
        for (int k = 0; k < 3; k++) {
            const double x = fs.coord(r.frame, k, d->source.location);
            in = in && r.lo[k] + m < x && x < r.hi[k] - m;
        }
        if (!in) continue;
        // A launched photon that leaves the region through an open side has tried the cell in vain and waits for
        // the list walk, where all the photons of a renew pass would otherwise have gone together: the region
        // is the source's cell only if enough of a hemisphere source's photons meet a real side (measured on
        // random scenes: a gain from about half of them on; a strict source's first hit is known anyway).
        const DevCell &rec = plan.cells[2 * (1 + n_box + g)];
        double share = 1.0;
        if (d->source.direction_mode == 1) {
            const int nz = 64, nphi = 128;
            int hits = 0;
            for (int a = 0; a < nz; a++)
                for (int b = 0; b < nphi; b++) {
                    const double z = (a + 0.5) / nz, rho = std::sqrt(1.0 - z * z), phi = 6.283185307179586 * (b + 0.5) / nphi;
                    const double v[3] = {rho * std::cos(phi), rho * std::sin(phi), z};
                    double w[3] = {0, 0, 0};   // PhotonObject.py:71-72: the sample times the source's frame
                    for (int r2 = 0; r2 < 3; r2++)
                        for (int c = 0; c < 3; c++) w[c] += v[r2] * d->source.frame[3 * r2 + c];
                    double best = 1e300;
                    int32_t id = CONFINED_NONE;
                    for (int k = 0; k < 3; k++) {
                        const double *axis = &fs.r[r.frame][3 * k];
                        const double dk = axis[0] * w[0] + axis[1] * w[1] + axis[2] * w[2];
                        if (dk == 0) continue;
                        const double x = fs.coord(r.frame, k, d->source.location);
                        const double t = ((dk > 0 ? (double)rec.phi[k] : (double)rec.plo[k]) - x) / dk;
                        if (t < best) { best = t; id = dk > 0 ? rec.idhi[k] : rec.idlo[k]; }
                    }
                    hits += id != CONFINED_NONE;
                }
            share = (double)hits / (nz * nphi);
        }
        const char *min_share = getenv("RTMC_REGION_START_MIN");
        if (getenv("RTMC_DEBUG")) fprintf(stderr, "[rtmc] source in region %d: %.3f of its photons meet a real side\n", 1 + n_box + (int)g, share);
        if (share >= (min_share ? atof(min_share) : 0.45)) plan.start_cell = 1 + n_box + (int)g;
    }
    return plan;
}

// Waves per SIMD the float32 trace kernel can keep resident beside an LDS image of this size: workgroups of 512,
// 768 or 1024 threads, RTMC_TRACE_WAVES waves at most.
int trace_waves_per_simd(size_t lds_bytes)
{
    int best = 0;
    for (int block = TRACE_BLOCK; block <= TRACE_BLOCK_MAX; block += 256) {
        const int resident = (int)(160 * 1024 / (lds_bytes + 1024));   // (a margin for the allocation granule)
        best = std::max(best, std::min(RTMC_TRACE_WAVES, resident * block / 256));
    }
    return best;
}

template <typename real>
void build_image(const rtmc_scene_desc *d, const std::vector<double> &wl,
                 const std::vector<std::vector<double>> &cdfs, const std::vector<int> &cdf_off,
                 const std::vector<int> &mat_cdfD, const std::vector<int> &mat_cdfA,
                 const std::vector<int> &cdf_wid_off, const std::vector<int> &mat_abs_slot,
                 const std::vector<HostTable> &absA, const std::vector<HostTable> &absD, int src_cdf,
                 const std::vector<uint32_t> &energy_fx, const std::vector<uint16_t> &wid_cbin, int n_bins,
                 const std::vector<int32_t> &el_slot, int n_dump, const FrameSet &fs, const CellPlan &plan,
                 int guide_bits, bool hist_may_move, DeviceImage<real> &img)
{
    const int n_el = d->n_elements, n_mat = d->n_materials;
    const int guide_buckets = 1 << guide_bits;
    const int n_wid = (int)wl.size();
    img.elems.resize(n_el);
    for (int i = 0; i < n_el; i++) {
        const rtmc_element &e = d->elements[i];
        DevElem<real> &o = img.elems[i];
        memset(&o, 0, sizeof o);
        auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
        o.nx = (real)e.normal[0]; o.ny = (real)e.normal[1]; o.nz = (real)e.normal[2];
        o.d0 = (real)dot(e.normal, e.support);
        o.ax = (real)e.e1[0]; o.ay = (real)e.e1[1]; o.az = (real)e.e1[2];
        o.c1 = (real)dot(e.e1, e.support);
        o.bx = (real)e.e2[0]; o.by = (real)e.e2[1]; o.bz = (real)e.e2[2];
        o.c2 = (real)dot(e.e2, e.support);
        o.h1 = (real)e.half1; o.h2 = (real)e.half2;
        o.sz = (real)e.support[2];
        o.focal = (real)e.focal_length;
        if (e.geometry == RTMC_GEOM_CIRCLE) {  // Geometries.py:75
            o.ax = (real)e.support[0]; o.ay = (real)e.support[1]; o.az = (real)e.support[2];
            o.h1 = (real)(e.half1 * e.half1);
        } else if (e.geometry == RTMC_GEOM_TRIANGLE) {  // Geometries.py:232-247
            double n1[3], n2[3];
            axis_frame(e.normal, n1, n2);
            const double *corner[3] = {e.e1, e.e2, e.vertex3};
            real *q[3][2] = {{&o.c1, &o.c2}, {&o.h1, &o.h2}, {&o.g1, &o.g2}};
            for (int c = 0; c < 3; c++) { *q[c][0] = (real)dot(n1, corner[c]); *q[c][1] = (real)dot(n2, corner[c]); }
            o.ax = (real)n1[0]; o.ay = (real)n1[1]; o.az = (real)n1[2];
            o.bx = (real)n2[0]; o.by = (real)n2[1]; o.bz = (real)n2[2];
        } else if (e.geometry == RTMC_GEOM_CYLINDER) {  // Geometries.py:158-173: support/normal are the axis end points
            double a[3] = {e.support[0] - e.normal[0], e.support[1] - e.normal[1], e.support[2] - e.normal[2]};
            const double na = std::sqrt(dot(a, a));
            for (int j = 0; j < 3; j++) a[j] /= na;
            const double x1 = 0.0, x2 = dot(a, e.normal) - dot(a, e.support);  // axial coordinates relative to the first point
            o.ax = (real)a[0]; o.ay = (real)a[1]; o.az = (real)a[2];
            o.bx = (real)e.support[0]; o.by = (real)e.support[1]; o.bz = (real)e.support[2];
            o.c1 = (real)(e.half1 * e.half1);
            o.h1 = (real)std::min(x1, x2); o.h2 = (real)std::max(x1, x2);
            o.nx = o.ny = o.nz = o.d0 = (real)0;
        }
        o.type = e.type | (e.geometry << 8);
        o.info = 0;
        // (interfaces and mirrors: the cell the element is a face of and the cell around that one in the upper
        // half, read by the kernels with FEAT_CELLS only)
        if (e.type == RTMC_TYPE_INTERFACE || e.type == RTMC_TYPE_MIRROR)
            o.info = ((int32_t)plan.el_cell[i] << 16) | ((int32_t)plan.el_outward[i] << 23) | (int32_t)((uint32_t)plan.el_outer[i] << 24);
        if (e.type == RTMC_TYPE_INTERFACE) o.info |= e.material1 | (e.material2 << 8);
        if (e.type == RTMC_TYPE_DUMP) o.info = ((e.eqe_table + 1) & 0xFF) | (el_slot[i] << 8);
    }
    // interface records carry what the operation needs to know about both media
    for (int i = 0; i < n_el; i++) {
        const rtmc_element &e = d->elements[i];
        if (e.type != RTMC_TYPE_INTERFACE) continue;
        DevElem<real> &o = img.elems[i];
        o.nid1 = d->materials[e.material1].name_id;
        o.n1v = (real)d->materials[e.material1].n;
        o.n2v = (real)d->materials[e.material2].n;
        o.slots = (mat_abs_slot[e.material1] + 1) | ((mat_abs_slot[e.material2] + 1) << 16);
    }
    // nearest-hit table (layout: nearest_valid_hit).  A rectangle whose normal and
    // in-plane axes are mutually perpendicular joins the frame whose axes they
    // are (up to order and sign); the world frame is frame 0 and is matched exactly.
    struct Frame { double r[3][3]; std::vector<HitRec<real>> group[3]; };
    std::vector<Frame> frames(fs.r.size());
    for (size_t fi = 0; fi < fs.r.size(); fi++)
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) frames[fi].r[a][c] = fs.r[fi][3 * a + c];
    std::vector<HitRec<real>> general;
    for (int i = 0; i < n_el; i++) {
        const rtmc_element &e = d->elements[i];
        HitRec<real> h;
        memset(&h, 0, sizeof h);
        h.index = i;
        const FrameSet::Place &pl = fs.place[i];
        if (pl.frame < 0) { general.push_back(h); continue; }
        const int kn = pl.kn, ai = (kn + 1) % 3, aj = (kn + 2) % 3;
        h.p0 = (real)fs.coord(pl.frame, kn, e.support);
        h.ci = (real)fs.coord(pl.frame, ai, e.support); h.cj = (real)fs.coord(pl.frame, aj, e.support);
        h.hi = (real)(pl.k1 == ai ? e.half1 : e.half2);
        h.hj = (real)(pl.k1 == ai ? e.half2 : e.half1);
        frames[pl.frame].group[kn].push_back(h);
    }
    auto pad = [](std::vector<HitRec<real>> &recs) {
        while (recs.size() % HIT_CHUNK) {  // padding that no ray can hit
            HitRec<real> h;
            memset(&h, 0, sizeof h);
            h.hi = h.hj = (real)-1;
            h.index = -1;
            recs.push_back(h);
        }
    };
    auto emit = [&](const std::vector<HitRec<real>> &recs) {
        for (size_t c = 0; c < recs.size(); c += HIT_CHUNK) {
            HitChunk<real> ch;
            memset(&ch, 0, sizeof ch);
            for (int k = 0; k < HIT_CHUNK; k++) ch.r[k] = recs[c + k];
            img.hits.push_back(ch);
        }
    };
    img.hits.clear();
    img.has_runs = false;
    img.n_runs = img.n_run_members = 0;
    // float32: RUN_MIN_MEMBERS or more rectangles of a group that lie in one plane (the same p0, bit
    // for bit) leave the group's list and follow it as a run (RunHdr): a grid over their bounding box
    // whose cells name the members that touch them.  A member is entered into every cell its outline,
    // widened by 1e-5 of the box, overlaps: the cell of a landing point is computed in float32 and
    // may come out as a neighbour of the true one.  List order is kept among the records that stay
    // and inside every cell.  A plane whose rectangles crowd more than RUN_CELL_SLOTS into one cell
    // (two words of four one-byte names per cell) at every resolution tried stays in the list.
    struct Run { RunHdr<real> hdr; };
    std::vector<std::vector<Run>> runs(frames.size() * 3);
    std::vector<uint32_t> run_cells;
    std::vector<RunMember<real>> run_members;
    auto build_run = [&](const std::vector<HitRec<real>> &members, Run &out) {
        if (members.size() > 255) return false;
        double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300}, wmin[2] = {1e300, 1e300};
        for (const HitRec<real> &m : members) {
            const double c[2] = {(double)m.ci, (double)m.cj}, h[2] = {(double)m.hi, (double)m.hj};
            for (int a = 0; a < 2; a++) {
                lo[a] = std::min(lo[a], c[a] - h[a]); hi[a] = std::max(hi[a], c[a] + h[a]);
                wmin[a] = std::min(wmin[a], 2 * h[a]);
            }
        }
        if (!(wmin[0] > 0) || !(wmin[1] > 0)) return false;
        // the resolution (cells per smallest member: 1, 2 or 4) whose fullest cell names the fewest
        // members: a wave tries as many members as its fullest lane has to
        int best_fill = RUN_CELL_SLOTS + 1, best_n[2] = {0, 0};
        std::vector<uint32_t> best_cells;
        double best_w[2] = {0, 0};
        for (int refine = 1; refine <= 4; refine *= 2) {
            int n[2];
            double w[2], eps[2];
            for (int a = 0; a < 2; a++) {
                const double ext = hi[a] - lo[a];
                n[a] = (int)std::min(255.0, std::max(1.0, std::ceil(ext / wmin[a] * refine - 1e-9)));
                w[a] = ext / n[a];
                eps[a] = 1e-5 * ext;
            }
            if ((size_t)n[0] * n[1] > 4096) break;
            std::vector<uint32_t> cells(2 * (size_t)n[0] * n[1], 0xFFFFFFFFu);   // two words per cell
            std::vector<int> fill((size_t)n[0] * n[1], 0);
            bool ok = true;
            for (size_t mi = 0; mi < members.size() && ok; mi++) {
                const HitRec<real> &m = members[mi];
                const double c[2] = {(double)m.ci, (double)m.cj}, h[2] = {(double)m.hi, (double)m.hj};
                int first[2], last[2];
                for (int a = 0; a < 2; a++) {
                    first[a] = std::max(0, std::min(n[a] - 1, (int)std::floor((c[a] - h[a] - eps[a] - lo[a]) / w[a])));
                    last[a] = std::max(0, std::min(n[a] - 1, (int)std::floor((c[a] + h[a] + eps[a] - lo[a]) / w[a])));
                }
                for (int cj = first[1]; cj <= last[1] && ok; cj++)
                    for (int ci = first[0]; ci <= last[0] && ok; ci++) {
                        const size_t cell = (size_t)cj * n[0] + ci;
                        if (fill[cell] == RUN_CELL_SLOTS) { ok = false; break; }
                        uint32_t &word = cells[2 * cell + fill[cell] / 4];
                        const int shift = 8 * (fill[cell] % 4);
                        word = (word & ~(0xFFu << shift)) | ((uint32_t)mi << shift);
                        fill[cell]++;
                    }
            }
            if (!ok) continue;
            const int fullest = *std::max_element(fill.begin(), fill.end());
            if (fullest < best_fill) {
                best_fill = fullest; best_cells.swap(cells);
                for (int a = 0; a < 2; a++) { best_n[a] = n[a]; best_w[a] = w[a]; }
            }
        }
        if (best_cells.empty()) return false;
        memset(&out.hdr, 0, sizeof out.hdr);
        out.hdr.p0 = members[0].p0;
        out.hdr.gi0 = (real)lo[0]; out.hdr.gj0 = (real)lo[1];
        out.hdr.inv_wi = (real)(1.0 / best_w[0]); out.hdr.inv_wj = (real)(1.0 / best_w[1]);
        out.hdr.ni_1 = best_n[0] - 1; out.hdr.nj_1 = best_n[1] - 1;
        out.hdr.cell_off = (int32_t)(run_cells.size() / 2);
        out.hdr.mem_off = (int32_t)run_members.size();
        run_cells.insert(run_cells.end(), best_cells.begin(), best_cells.end());
        for (const HitRec<real> &m : members) {
            RunMember<real> rm;
            memset(&rm, 0, sizeof rm);
            rm.ci = m.ci; rm.cj = m.cj; rm.hi = m.hi; rm.hj = m.hj; rm.index = m.index;
            run_members.push_back(rm);
        }
        return true;
    };
    size_t run_min = RUN_MIN_MEMBERS;
    if (const char *e = getenv("RTMC_RUN_MIN")) run_min = std::max(2, atoi(e));  // experiments
    if (sizeof(real) == 4 && !getenv("RTMC_NO_RUNS")) {
        for (size_t fi = 0; fi < frames.size(); fi++)
            for (int k = 0; k < 3; k++) {
                std::vector<HitRec<real>> &recs = frames[fi].group[k], stay;
                std::vector<Run> &rs = runs[fi * 3 + k];
                std::vector<real> planes_done;
                for (const HitRec<real> &h : recs) {
                    if (std::find(planes_done.begin(), planes_done.end(), h.p0) != planes_done.end()) continue;
                    planes_done.push_back(h.p0);
                    std::vector<HitRec<real>> members;
                    for (const HitRec<real> &o : recs)
                        if (o.p0 == h.p0) members.push_back(o);
                    Run r;
                    if (members.size() >= run_min && rs.size() < 32767 && build_run(members, r)) rs.push_back(r);
                    else planes_done.back() = h.p0, stay.insert(stay.end(), members.begin(), members.end());
                }
                // (the records that stay: back into list order)
                std::sort(stay.begin(), stay.end(), [](const HitRec<real> &a, const HitRec<real> &b) { return a.index < b.index; });
                if (stay.size() / HIT_CHUNK + 1 > 65535) { rs.clear(); continue; }  // (counts are 16-bit halves)
                recs.swap(stay);
                if (!rs.empty()) img.has_runs = true;
                img.n_runs += (int)rs.size();
            }
        img.n_run_members = (int)run_members.size();
    }
    // tiled faces of cells: each gets the grid of a run (float32 image only); a tiling the grid cannot hold
    // (more than RUN_CELL_SLOTS rectangles crowd one grid cell) makes its cell unusable
    std::vector<DevCell> cells = plan.cells;
    if (sizeof(real) != 4) cells.resize(2);   // (the float64 kernels walk the list: "no cell" only)
    std::vector<TileGrid> tile_grids(std::max<size_t>(1, plan.tiles.size()));
    memset(tile_grids.data(), 0, sizeof(TileGrid) * tile_grids.size());
    img.n_tiled_faces = 0;
    if (sizeof(real) == 4) {
        for (size_t g = 0; g < plan.tiles.size(); g++) {
            std::vector<HitRec<real>> members;
            for (const CellPlan::Tile &t : plan.tiles[g]) {
                HitRec<real> h;
                memset(&h, 0, sizeof h);
                h.ci = (real)t.ci; h.cj = (real)t.cj; h.hi = (real)t.hi; h.hj = (real)t.hj; h.index = t.index;
                members.push_back(h);
            }
            Run r;
            if (build_run(members, r)) {
                TileGrid &o = tile_grids[g];
                o.gi0 = (float)r.hdr.gi0; o.gj0 = (float)r.hdr.gj0; o.inv_wi = (float)r.hdr.inv_wi; o.inv_wj = (float)r.hdr.inv_wj;
                o.ni_1 = r.hdr.ni_1; o.nj_1 = r.hdr.nj_1; o.cell_off = r.hdr.cell_off; o.mem_off = r.hdr.mem_off;
                img.n_tiled_faces++;
            } else {
                for (DevCell &c : cells) {   // (the cell of the face, and the regions that end on it)
                    bool names = false;
                    for (int k = 0; k < 3; k++) names = names || c.idlo[k] == -((int32_t)g + 2) || c.idhi[k] == -((int32_t)g + 2);
                    if (names)
                        for (int k = 0; k < 3; k++) c.hw[k] = -1.0f;
                }
            }
        }
    }
    auto group_count = [&](size_t fi, int k) {
        return (int)(frames[fi].group[k].size() / HIT_CHUNK) | ((int)runs[fi * 3 + k].size() << 16);
    };
    auto emit_runs = [&](std::vector<Run> &rs) {
        for (Run &r : rs) {
            HitChunk<real> raw;
            static_assert(sizeof(RunHdr<real>) == sizeof(HitChunk<real>), "header = one chunk");
            memcpy(&raw, &r.hdr, sizeof raw);
            img.hits.push_back(raw);
        }
    };
    int extra_frames = 0;
    for (size_t fi = 0; fi < frames.size(); fi++) {
        Frame &f = frames[fi];
        for (int k = 0; k < 3; k++) pad(f.group[k]);
        if (fi == 0) {
            for (int k = 0; k < 3; k++) img.n_group[k] = group_count(fi, k);
        } else {
            static_assert(sizeof(FrameHdr<real>) == sizeof(HitChunk<real>), "header = one chunk");
            FrameHdr<real> hdr;
            memset(&hdr, 0, sizeof hdr);
            for (int a = 0; a < 3; a++)
                for (int c = 0; c < 3; c++) hdr.R[3 * a + c] = (real)f.r[a][c];
            for (int k = 0; k < 3; k++) hdr.n[k] = group_count(fi, k);
            HitChunk<real> raw;
            memcpy(&raw, &hdr, sizeof raw);
            img.hits.push_back(raw);
            extra_frames++;
        }
        for (int k = 0; k < 3; k++) { emit(f.group[k]); emit_runs(runs[fi * 3 + k]); }
    }
    pad(general);
    emit(general);
    img.n_group[3] = (int)(general.size() / HIT_CHUNK) | (extra_frames << 16);
    std::vector<DevMat<real>> mats(n_mat);
    int n_slots = 0;
    for (int i = 0; i < n_mat; i++) {
        const rtmc_material &m = d->materials[i];
        DevMat<real> &o = mats[i];
        memset(&o, 0, sizeof o);
        o.n = (real)m.n;
        o.flags = (m.is_absorber ? 1 : 0) | (m.is_absorber && m.is_aligned ? 2 : 0);
        o.name_id = m.name_id;
        o.abs_slot = mat_abs_slot[i];
        if (!m.is_absorber) continue;
        n_slots = std::max(n_slots, o.abs_slot + 1);
        double p[3], pa[3] = {0, 0, 0};
        if (m.conc_a > 0) {  // MonteCarloFRET.py:67-77
            o.n_p = 3;
            p[0] = (1 - m.P_ET_DA) * m.Q_D; p[1] = (1 - m.P_ET_DA) * (1 - m.Q_D); p[2] = m.P_ET_DA;
            pa[0] = (1 - m.P_ET_AD) * m.Q_A; pa[1] = (1 - m.P_ET_AD) * (1 - m.Q_A); pa[2] = m.P_ET_AD;
        } else {  // MonteCarloFRET.py:82
            o.n_p = 2;
            p[0] = m.Q_D; p[1] = 1 - m.Q_D; p[2] = 0;
        }
        double sp = 0, spa = 0, c = 0, ca = 0;
        for (int j = 0; j < o.n_p; j++) sp += p[j];
        for (int j = 0; j < 3; j++) spa += pa[j];
        for (int j = 0; j < 3; j++) {
            c += p[j];
            ca += pa[j];
            o.pT[j] = j < o.n_p ? threshold32(c / sp) : 0xFFFFFFFFu;
            o.pAT[j] = spa > 0 ? threshold32(ca / spa) : 0xFFFFFFFFu;
        }
        // the acceptor pathway draw is a multiple of 256: a threshold above the largest such value is never reached
        if (o.n_p == 3 && o.pAT[1] <= 0xFFFFFF00u) img.back_possible = true;
        o.cdfD_off = cdf_off[mat_cdfD[i]]; o.cdfD_len = (int)cdfs[mat_cdfD[i]].size();
        o.widD_off = cdf_wid_off[mat_cdfD[i]];
        o.cdfA_off = cdf_off[mat_cdfA[i]]; o.cdfA_len = (int)cdfs[mat_cdfA[i]].size();
        o.gD_off = mat_cdfD[i] * (guide_buckets + 1); o.gA_off = mat_cdfA[i] * (guide_buckets + 1);
        o.guide_shift = 32 - guide_bits;
        o.widA_off = cdf_wid_off[mat_cdfA[i]];
    }
    // aligned acceptors: PhotonObject.py:291-302, MonteCarloFRET.py:124-157
    std::vector<DevOrient<real>> orient(n_mat);
    for (int i = 0; i < n_mat; i++) {
        const rtmc_material &m = d->materials[i];
        DevOrient<real> &o = orient[i];
        memset(&o, 0, sizeof o);
        if (!(m.is_absorber && m.is_aligned)) continue;
        o.P = (real)m.psi_P; o.Q = (real)m.psi_Q;
        for (int j = 0; j < 9; j++) o.f[j] = (real)m.align_frame[j];
        o.S = (real)m.align_S;
        const double th = (m.abs_em_angle / 180.0) * 3.141592653589793;  // MonteCarloFRET.py:140
        o.sin_ae = (real)std::sin(th); o.cos_ae = (real)std::cos(th);
        o.flags = (d->source.respect_polarization ? 1 : 0) | (m.abs_em_angle > 0 ? 2 : 0);
    }
    std::vector<DevAbs<real>> abs((size_t)n_slots * n_wid);
    const double ln10 = std::log(10.0);
    for (int i = 0; i < n_mat; i++) {
        const rtmc_material &m = d->materials[i];
        if (!m.is_absorber) continue;
        const int slot = mat_abs_slot[i];
        for (int w = 0; w < n_wid; w++) {
            // PhotonObject.py:235-236 / 288-289
            double ea = absA[i].y[nearest_index(absA[i].x, wl[w])];
            double ed = absD[i].y[nearest_index(absD[i].x, wl[w])];
            double b_a = m.conc_a * 1e-3 * m.emax_A * ln10 * std::max(1e-99, ea);
            double b_d = m.conc_b * 1e-3 * m.emax_D * ln10 * std::max(1e-99, ed);
            DevAbs<real> &o = abs[(size_t)slot * n_wid + w];
            o.b_a = (real)b_a; o.b_d = (real)b_d;
            o.coinT = threshold32(b_d / (b_d + b_a));
        }
    }
    std::vector<uint32_t> cdfT;
    for (auto &c : cdfs)
        for (double v : c) cdfT.push_back(threshold32(v));
    if (cdfT.empty()) cdfT.push_back(0xFFFFFFFFu);
    // guide tables: thresholds below the first value of every bucket of the top guide_bits bits
    std::vector<uint16_t> guide;
    for (auto &c : cdfs) {
        std::vector<uint32_t> T;
        for (double v : c) T.push_back(threshold32(v));
        size_t cnt = 0;
        for (int b = 0; b <= guide_buckets; b++) {
            const uint64_t first = (uint64_t)b << (32 - guide_bits);
            while (cnt < T.size() && (uint64_t)T[cnt] < first) cnt++;
            guide.push_back((uint16_t)(b == guide_buckets ? T.size() : cnt));
        }
    }
    if (guide.empty()) guide.assign(guide_buckets + 1, 0);

    auto align = [](size_t v) { return (v + 15) & ~(size_t)15; };
    LdsLayout &lay = img.lay;
    size_t off = 0;
    lay.elems = (uint32_t)off; off = align(off + sizeof(DevElem<real>) * std::max(1, n_el));
    lay.mats = (uint32_t)off; off = align(off + sizeof(DevMat<real>) * std::max(1, n_mat));
    lay.orient = (uint32_t)off; off = align(off + sizeof(DevOrient<real>) * std::max(1, n_mat));
    lay.abs = (uint32_t)off; off = align(off + sizeof(DevAbs<real>) * std::max<size_t>(1, abs.size()));
    lay.cdf = (uint32_t)off; off = align(off + 4 * cdfT.size());
    lay.guide = (uint32_t)off; off = align(off + 2 * guide.size());
    lay.fx = (uint32_t)off; off = align(off + 4 * energy_fx.size());
    lay.bin = (uint32_t)off; off = align(off + 2 * wid_cbin.size());
    lay.run_cells = (uint32_t)off; off = align(off + 4 * run_cells.size());
    lay.run_members = (uint32_t)off; off = align(off + sizeof(RunMember<real>) * run_members.size());
    lay.cells = (uint32_t)off; off = align(off + sizeof(CellHeader) + sizeof(DevCell) * cells.size());
    lay.tile_grids = (uint32_t)off; off = align(off + sizeof(TileGrid) * tile_grids.size());
    // axes of the frames that hold cells, rows of three in twelve floats (three 16-byte reads per lane)
    std::vector<float> cell_frames(12 * (size_t)plan.n_frames_used, 0.0f);
    for (int f = 0; f < plan.n_frames_used; f++)
        for (int j = 0; j < 9; j++) cell_frames[12 * f + j] = (float)fs.r[f][j];
    lay.cell_frames = (uint32_t)off; off = align(off + 4 * cell_frames.size());
    lay.acc = (uint32_t)off;
    lay.n_wid = (uint32_t)n_wid;
    lay.n_bins = (uint32_t)n_bins;
    lay.n_dump = (uint32_t)n_dump;
    const size_t n_acc = ACC_SCALARS + 3 * (size_t)n_el;
    lay.hist = (uint32_t)(off + 8 * n_acc);
    // The detector histogram (detectors x bins, 4 bytes each) stays in LDS while a workgroup of the
    // largest size still fits; beyond that it is counted in HBM
    size_t hist_limit = 160 * 1024;
    if (const char *e = getenv("RTMC_LDS_HIST_LIMIT")) hist_limit = (size_t)strtoull(e, nullptr, 10);  // experiments
    const size_t with_hist = align(lay.hist + 4 * (size_t)std::max(1, n_dump * n_bins));
    lay.hist_in_hbm = with_hist > hist_limit ? 1u : 0u;
    // ... or while it does not cost the float32 kernel a wave per SIMD, in scenes whose kernel counts in HBM at no other
    // price (`hist_may_move`: the kernels with closed boxes; a read-out of 62 detectors +19 %, eight stacked slabs +13 %)
    if (sizeof(real) == 4 && hist_may_move && !getenv("RTMC_LDS_HIST_LIMIT") &&
        trace_waves_per_simd(align(lay.hist + 16)) > trace_waves_per_simd(with_hist))
        lay.hist_in_hbm = 1u;
    lay.total = (uint32_t)(lay.hist_in_hbm ? align(lay.hist + 16) : with_hist);
    img.lds.assign(off, 0);
    if (n_el) memcpy(img.lds.data() + lay.elems, img.elems.data(), sizeof(DevElem<real>) * n_el);
    if (n_mat) memcpy(img.lds.data() + lay.mats, mats.data(), sizeof(DevMat<real>) * n_mat);
    if (n_mat) memcpy(img.lds.data() + lay.orient, orient.data(), sizeof(DevOrient<real>) * n_mat);
    if (!abs.empty()) memcpy(img.lds.data() + lay.abs, abs.data(), sizeof(DevAbs<real>) * abs.size());
    memcpy(img.lds.data() + lay.cdf, cdfT.data(), 4 * cdfT.size());
    memcpy(img.lds.data() + lay.guide, guide.data(), 2 * guide.size());
    memcpy(img.lds.data() + lay.fx, energy_fx.data(), 4 * energy_fx.size());
    memcpy(img.lds.data() + lay.bin, wid_cbin.data(), 2 * wid_cbin.size());
    if (!run_cells.empty()) memcpy(img.lds.data() + lay.run_cells, run_cells.data(), 4 * run_cells.size());
    if (!run_members.empty()) memcpy(img.lds.data() + lay.run_members, run_members.data(), sizeof(RunMember<real>) * run_members.size());
    {
        CellHeader hdr;
        memset(&hdr, 0, sizeof hdr);
        hdr.preset_e = -1;
        // (F_PRESET is set on the device, see preset_hit_kernel, which then drops F_LIST)
        const size_t start_cell = 2 * (size_t)plan.start_cell < cells.size() ? (size_t)plan.start_cell : 0;
        hdr.launch_flags = ((uint32_t)start_cell << F_CELL_SHIFT) | (cells[2 * start_cell].hw[0] < 0.0f ? F_LIST : 0u);
        hdr.n_cells = (int32_t)cells.size() / 2 - 1;
        memcpy(img.lds.data() + lay.cells, &hdr, sizeof hdr);
        memcpy(img.lds.data() + lay.cells + sizeof hdr, cells.data(), sizeof(DevCell) * cells.size());
        memcpy(img.lds.data() + lay.tile_grids, tile_grids.data(), sizeof(TileGrid) * tile_grids.size());
        memcpy(img.lds.data() + lay.cell_frames, cell_frames.data(), 4 * cell_frames.size());
    }

    const rtmc_source &s = d->source;
    DevSource<real> &o = img.src;
    memset(&o, 0, sizeof o);
    // PhotonObject.py:86, Direction /= np.linalg.norm(Direction): numpy takes the norm of a 3-vector as
    // sqrt(dot(x, x)) and sums that dot product as fma(x2, x2, fma(x1, x1, x0 * x0)).  The last bit matters:
    // a strict source aimed along a tilted plane's normal meets it with a cosine of exactly 1 or 1 + 2^-52
    // (see interface_reflectance).
    double nrm = std::sqrt(std::fma(s.direction[2], s.direction[2],
                                    std::fma(s.direction[1], s.direction[1], s.direction[0] * s.direction[0])));
    for (int j = 0; j < 3; j++) {
        o.loc[j] = (real)s.location[j];
        o.dir[j] = (real)(nrm > 0 ? s.direction[j] / nrm : 0.0);
    }
    for (int j = 0; j < 9; j++) o.frame[j] = (real)s.frame[j];
    o.inv_scale = (real)(1.0 / s.scale_factor);
    o.dir_mode = s.direction_mode; o.wl_mode = s.wavelength_mode; o.start_mat = s.start_material;
    o.wid_fixed = 0;
    o.has_area = (s.area[0] != 0.0 || s.area[1] != 0.0) ? 1 : 0;
    o.area[0] = (real)s.area[0]; o.area[1] = (real)s.area[1];
    if (src_cdf >= 0) {
        o.cdf_off = cdf_off[src_cdf]; o.cdf_len = (int)cdfs[src_cdf].size();
        o.guide_off = src_cdf * (guide_buckets + 1);
        o.guide_shift = 32 - guide_bits;
    }
}

// Do two flat elements lie in one plane and overlap there?  (Rectangles that
// only share an edge do not count: a photon meets the seam with probability
// zero.)  Footprints are compared through their bounding circles and, for two
// rectangles, through the separating-axis test on their in-plane axes.
bool coplanar_overlap(const rtmc_scene_desc *d)
{
    auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    auto flat = [](const rtmc_element &e) { return e.geometry != RTMC_GEOM_CYLINDER; };
    auto radius = [&](const rtmc_element &e) {
        if (e.geometry == RTMC_GEOM_RECTANGLE) return std::sqrt(e.half1 * e.half1 + e.half2 * e.half2);
        if (e.geometry == RTMC_GEOM_CIRCLE) return e.half1;
        double r = 0;  // triangle: support is the centroid
        for (const double *c : {e.e1, e.e2, e.vertex3}) {
            double v[3] = {c[0] - e.support[0], c[1] - e.support[1], c[2] - e.support[2]};
            r = std::max(r, std::sqrt(dot(v, v)));
        }
        return r;
    };
    for (int i = 0; i < d->n_elements; i++) {
        const rtmc_element &a = d->elements[i];
        if (!flat(a)) continue;
        for (int j = i + 1; j < d->n_elements; j++) {
            const rtmc_element &b = d->elements[j];
            if (!flat(b)) continue;
            const double scale = 1.0 + std::fabs(dot(a.normal, a.support));
            if (std::fabs(std::fabs(dot(a.normal, b.normal)) - 1.0) > 1e-9) continue;           // not parallel
            double gap[3] = {b.support[0] - a.support[0], b.support[1] - a.support[1], b.support[2] - a.support[2]};
            if (std::fabs(dot(a.normal, gap)) > 1e-9 * scale) continue;                           // different planes
            const double ra = radius(a), rb = radius(b), tol = 1e-9 * (ra + rb + scale);
            if (std::sqrt(dot(gap, gap)) >= ra + rb - tol) continue;                              // far apart
            if (a.geometry == RTMC_GEOM_RECTANGLE && b.geometry == RTMC_GEOM_RECTANGLE) {
                bool separated = false;
                const double *axes[4] = {a.e1, a.e2, b.e1, b.e2};
                for (const double *ax : axes) {
                    const double reach_a = a.half1 * std::fabs(dot(ax, a.e1)) + a.half2 * std::fabs(dot(ax, a.e2));
                    const double reach_b = b.half1 * std::fabs(dot(ax, b.e1)) + b.half2 * std::fabs(dot(ax, b.e2));
                    if (std::fabs(dot(ax, gap)) >= reach_a + reach_b - tol) separated = true;
                }
                if (separated) continue;
            }
            return true;
        }
    }
    return false;
}

// Is a strict source aimed at a seam: does its line first land (float64) on a point that belongs to
// a second flat element as well, within what float32 cannot resolve?  Every photon of such a source
// then sits on both elements after its first operation.  Only the first landing point is examined;
// cylinder walls are left out (as first element and as neighbour).
bool strict_source_meets_seam(const rtmc_scene_desc *d)
{
    const rtmc_source &s = d->source;
    if (s.direction_mode != 0 || s.area[0] != 0.0 || s.area[1] != 0.0) return false;
    auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    const double dn = std::sqrt(dot(s.direction, s.direction));
    if (!(dn > 0)) return false;
    const double D[3] = {s.direction[0] / dn, s.direction[1] / dn, s.direction[2] / dn};
    // is point P a point of flat element e, its outline widened by tol?
    auto contains = [&](const rtmc_element &e, const double *P, double tol) {
        const double v[3] = {P[0] - e.support[0], P[1] - e.support[1], P[2] - e.support[2]};
        if (std::fabs(dot(v, e.normal)) > tol) return false;
        if (e.geometry == RTMC_GEOM_RECTANGLE)
            return std::fabs(dot(v, e.e1)) <= e.half1 + tol && std::fabs(dot(v, e.e2)) <= e.half2 + tol;
        if (e.geometry == RTMC_GEOM_CIRCLE) return std::sqrt(dot(v, v)) <= e.half1 + tol;
        // triangle (corners in e1, e2, vertex3): on the inner side of every edge, each moved out by tol
        const double *c[3] = {e.e1, e.e2, e.vertex3};
        int pos = 0, neg = 0;
        for (int k = 0; k < 3; k++) {
            const double *a = c[k], *b = c[(k + 1) % 3];
            const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ap[3] = {P[0] - a[0], P[1] - a[1], P[2] - a[2]};
            const double cr[3] = {ab[1] * ap[2] - ab[2] * ap[1], ab[2] * ap[0] - ab[0] * ap[2], ab[0] * ap[1] - ab[1] * ap[0]};
            const double side = dot(cr, e.normal) / std::max(1e-300, std::sqrt(dot(ab, ab)));  // signed distance from the edge
            if (side > tol) pos++;
            if (side < -tol) neg++;
        }
        return pos == 0 || neg == 0;
    };
    double reach = std::sqrt(dot(s.location, s.location));
    for (int i = 0; i < d->n_elements; i++) reach = std::max(reach, std::sqrt(dot(d->elements[i].support, d->elements[i].support)));
    const double tol = 64.0 / 16777216.0 * (1.0 + reach);   // a few times the float32 guard distance
    double t0 = 1e300;
    int first = -1;
    for (int i = 0; i < d->n_elements; i++) {
        const rtmc_element &e = d->elements[i];
        if (e.geometry == RTMC_GEOM_CYLINDER) continue;
        const double den = dot(D, e.normal);
        if (std::fabs(den) < 1e-12) continue;
        const double w[3] = {e.support[0] - s.location[0], e.support[1] - s.location[1], e.support[2] - s.location[2]};
        const double t = dot(w, e.normal) / den;
        if (!(t > tol) || t >= t0) continue;
        const double P[3] = {s.location[0] + t * D[0], s.location[1] + t * D[1], s.location[2] + t * D[2]};
        if (contains(e, P, tol)) { t0 = t; first = i; }
    }
    if (first < 0) return false;
    const double P[3] = {s.location[0] + t0 * D[0], s.location[1] + t0 * D[1], s.location[2] + t0 * D[2]};
    for (int i = 0; i < d->n_elements; i++)
        if (i != first && d->elements[i].geometry != RTMC_GEOM_CYLINDER && contains(d->elements[i], P, tol)) return true;
    return false;
}

int get_table(const rtmc_scene_desc *d, rtmc_table ref, HostTable &t, const char *what)
{
    if (ref.len < 2 || ref.offset < 0 || ref.offset + 2 * ref.len > d->table_data_len)
        return fail(RTMC_EINVAL, "%s: table (offset %lld, len %lld) outside table_data or shorter than 2 rows",
                    what, (long long)ref.offset, (long long)ref.len);
    t.x.resize(ref.len); t.y.resize(ref.len);
    for (int64_t i = 0; i < ref.len; i++) {
        t.x[i] = d->table_data[ref.offset + 2 * i];
        t.y[i] = d->table_data[ref.offset + 2 * i + 1];
    }
    return RTMC_OK;
}

}  // namespace

RTMC_API int rtmc_version(void) { return RTMC_ABI_VERSION; }
RTMC_API const char *rtmc_last_error(void) { return g_error; }

static int scene_create_impl(const rtmc_scene_desc *d, rtmc_scene **out);

RTMC_API int rtmc_scene_create(const rtmc_scene_desc *d, rtmc_scene **out)
{
    if (!d || !out) return fail(RTMC_EINVAL, "null argument");
    *out = nullptr;
    return guarded([&] { return scene_create_impl(d, out); });
}

static int scene_create_impl(const rtmc_scene_desc *d, rtmc_scene **out)
{
    if (d->abi_version != RTMC_ABI_VERSION)
        return fail(RTMC_EINVAL, "description has ABI version %d, library %d", d->abi_version, RTMC_ABI_VERSION);
    if (d->n_elements < 1 || d->n_elements > MAX_ELEMENTS)
        return fail(RTMC_ELIMIT, "1..%d elements supported, got %d", MAX_ELEMENTS, d->n_elements);
    if (d->n_tables < 0 || d->n_tables > 254) return fail(RTMC_ELIMIT, "0..254 detection tables supported, got %d", d->n_tables);
    if (d->n_tables > 0 && !d->tables) return fail(RTMC_EINVAL, "null detection table array");
    if (d->table_data_len < 0 || (d->table_data_len > 0 && !d->table_data)) return fail(RTMC_EINVAL, "null or negative table data");
    if (d->n_materials < 1 || d->n_materials > 255) return fail(RTMC_ELIMIT, "1..255 materials supported, got %d", d->n_materials);
    if (d->max_iterations < 1) return fail(RTMC_EINVAL, "max_iterations must be positive");
    if (!d->elements || !d->materials) return fail(RTMC_EINVAL, "null element or material array");
    const rtmc_source &src = d->source;
    if (src.start_material < 0 || src.start_material >= d->n_materials) return fail(RTMC_EINVAL, "start material out of range");
    if (!(src.scale_factor > 0)) return fail(RTMC_EINVAL, "scale_factor must be positive");
    if (!(src.area[0] >= 0 && src.area[1] >= 0 && std::isfinite(src.area[0]) && std::isfinite(src.area[1])))
        return fail(RTMC_EINVAL, "source area widths must be finite and not negative");
    if (src.direction_mode != 0 && src.direction_mode != 1) return fail(RTMC_EINVAL, "unknown direction mode");
    bool has_shapes = false;
    for (int i = 0; i < d->n_elements; i++) {
        const rtmc_element &e = d->elements[i];
        if (e.geometry < RTMC_GEOM_RECTANGLE || e.geometry > RTMC_GEOM_TRIANGLE)
            return fail(RTMC_EINVAL, "element %d: unknown geometry code %d", i, e.geometry);
        if (e.type < RTMC_TYPE_MIRROR || e.type > RTMC_TYPE_DUMP)
            return fail(RTMC_EINVAL, "element %d: unknown element type code %d", i, e.type);
        if (e.geometry != RTMC_GEOM_RECTANGLE || e.type == RTMC_TYPE_LENS) has_shapes = true;
        if (e.geometry == RTMC_GEOM_CYLINDER) {
            double len2 = 0;
            for (int j = 0; j < 3; j++) len2 += (e.support[j] - e.normal[j]) * (e.support[j] - e.normal[j]);
            if (!(len2 > 0) || !(e.half1 > 0)) return fail(RTMC_EINVAL, "element %d: cylinder needs two distinct axis points and a radius", i);
        }
        if (e.type == RTMC_TYPE_INTERFACE &&
            (e.material1 < 0 || e.material1 >= d->n_materials || e.material2 < 0 || e.material2 >= d->n_materials))
            return fail(RTMC_EINVAL, "element %d: interface material out of range", i);
        if (e.type == RTMC_TYPE_DUMP && (e.eqe_table < -1 || e.eqe_table >= d->n_tables))
            return fail(RTMC_EINVAL, "element %d: detection table out of range", i);
    }

    std::vector<HostTable> absA(d->n_materials), absD(d->n_materials);
    std::vector<HostTable> em;       // unique emission / source tables
    std::vector<int> em_kind;        // 0 emission, 1 source spectrum
    std::vector<int> mat_cdfD(d->n_materials, -1), mat_cdfA(d->n_materials, -1), mat_abs_slot(d->n_materials, -1);
    int src_cdf = -1, n_slots = 0;
    bool has_aligned = false;
    std::vector<double> wl;
    std::vector<int> cdf_wid_off;

    auto add_table = [&](const HostTable &t, int kind) {
        for (size_t i = 0; i < em.size(); i++)
            if (em_kind[i] == kind && em[i].same(t)) return (int)i;
        em.push_back(t);
        em_kind.push_back(kind);
        cdf_wid_off.push_back((int)wl.size());
        wl.insert(wl.end(), t.x.begin(), t.x.end());
        return (int)em.size() - 1;
    };

    if (src.wavelength_mode == 0) {
        wl.push_back(src.wavelength);  // id 0
    } else if (src.wavelength_mode == 1) {
        HostTable t;
        int rc = get_table(d, src.spectrum, t, "source spectrum");
        if (rc) return rc;
        src_cdf = add_table(t, 1);
    } else
        return fail(RTMC_EINVAL, "unknown wavelength mode");

    for (int i = 0; i < d->n_materials; i++) {
        const rtmc_material &m = d->materials[i];
        if (!m.is_absorber) continue;
        if (m.is_aligned) {
            double f2 = 0;
            for (int j = 0; j < 9; j++) f2 += m.align_frame[j] * m.align_frame[j];
            if (!(f2 > 0) || !std::isfinite(f2) || !std::isfinite(m.psi_P) || !std::isfinite(m.psi_Q) || !(m.align_S >= 0))
                return fail(RTMC_EINVAL, "material %d: aligned dyes need align_frame, psi_P, psi_Q and a non-negative align_S", i);
            has_aligned = true;
        }
        int rc;
        HostTable eD, eA;
        if ((rc = get_table(d, m.abs_A, absA[i], "abs_A")) || (rc = get_table(d, m.abs_D, absD[i], "abs_D")) ||
            (rc = get_table(d, m.em_A, eA, "em_A")) || (rc = get_table(d, m.em_D, eD, "em_D")))
            return rc;
        mat_cdfD[i] = add_table(eD, 0);
        mat_cdfA[i] = add_table(eA, 0);
        mat_abs_slot[i] = n_slots++;
    }
    if (wl.size() > 65535) return fail(RTMC_ELIMIT, "%zu distinct wavelengths, at most 65535 supported", wl.size());

    std::vector<std::vector<double>> cdfs(em.size());
    std::vector<int> cdf_off(em.size());
    int run = 0;
    for (size_t i = 0; i < em.size(); i++) {
        cdfs[i] = sampling_cdf(em[i], em_kind[i] == 1);
        cdf_off[i] = run;
        run += (int)cdfs[i].size();
    }

    std::unique_ptr<rtmc_scene> holder(new rtmc_scene);
    rtmc_scene *s = holder.get();
    s->n_el = d->n_elements; s->n_mat = d->n_materials; s->n_eqe = d->n_tables; s->max_iter = d->max_iterations;
    s->has_aligned = has_aligned;
    s->has_shapes = has_shapes;
    s->has_area = src.area[0] != 0.0 || src.area[1] != 0.0;
    s->has_spectra = src.wavelength_mode == 1;
    for (int i = 0; i < d->n_elements; i++)
        if (d->elements[i].type == RTMC_TYPE_DUMP && d->elements[i].eqe_table >= 0) s->has_spectra = true;
    s->elements.assign(d->elements, d->elements + d->n_elements);
    s->has_coplanar = coplanar_overlap(d);
    s->seam_source = strict_source_meets_seam(d);
    s->wl = wl;
    s->el_eqe.resize(s->n_el);
    for (int i = 0; i < s->n_el; i++) s->el_eqe[i] = d->elements[i].type == RTMC_TYPE_DUMP ? d->elements[i].eqe_table : -1;
    // detector bin of every wavelength: nearest of 200..3999 nm, Elements.py:159;
    // the device histogram only keeps the bins that can occur
    std::vector<double> bins(RTMC_SPECTRUM_BINS);
    for (int i = 0; i < RTMC_SPECTRUM_BINS; i++) bins[i] = RTMC_SPECTRUM_FIRST_NM + i;
    std::vector<int32_t> wid_bin(wl.size());
    for (size_t w = 0; w < wl.size(); w++) wid_bin[w] = (int32_t)nearest_index(bins, wl[w]);
    s->bins_used = wid_bin;
    std::sort(s->bins_used.begin(), s->bins_used.end());
    s->bins_used.erase(std::unique(s->bins_used.begin(), s->bins_used.end()), s->bins_used.end());
    s->wid_cbin.resize(wl.size());
    for (size_t w = 0; w < wl.size(); w++)
        s->wid_cbin[w] = (uint16_t)(std::lower_bound(s->bins_used.begin(), s->bins_used.end(), wid_bin[w]) - s->bins_used.begin());
    s->el_slot.assign(s->n_el, -1);
    s->n_dump = 0;
    for (int i = 0; i < s->n_el; i++)
        if (d->elements[i].type == RTMC_TYPE_DUMP) s->el_slot[i] = s->n_dump++;
    // Elements.py:158 EnergyCounter += wl_start / wl, in units of 2^-24, for a single source line
    // (a source with a spectrum has no such table: the kernels divide the two wavelengths)
    s->energy_fx.assign(src.wavelength_mode == 0 ? wl.size() : 1, 0u);
    if (src.wavelength_mode == 0)
        for (size_t w = 0; w < wl.size(); w++) {
            double fx = std::llrint(wl[0] / wl[w] * RTMC_ENERGY_FX_SCALE);
            if (fx > 4294967295.0) return fail(RTMC_ELIMIT, "wavelength ratio %g too large", wl[0] / wl[w]);
            s->energy_fx[w] = (uint32_t)fx;
        }
    // detection thresholds, Elements.py:164
    s->eqeT.assign(std::max<size_t>(1, (size_t)d->n_tables * wl.size()), 0xFFFFFFFFu);
    for (int t = 0; t < d->n_tables; t++) {
        HostTable q;
        int rc = get_table(d, d->tables[t], q, "detection spectrum");
        if (rc) return rc;
        for (size_t w = 0; w < wl.size(); w++)
            s->eqeT[(size_t)t * wl.size() + w] = threshold32(q.y[nearest_index(q.x, wl[w])]);
    }
    const FrameSet fs = assign_frames(d);
    CellPlan plan = find_cells(d, fs, true);
    auto image32 = [&](int guide_bits, DeviceImage<float> &img) {
        // (the histogram may move to HBM for the sake of occupancy where the scene then still runs a kernel with closed
        // boxes: see rtmc_scene::cells_in_use and features)
        const bool hist_may_move = plan.n_usable > 0 && !s->has_shapes && !s->has_aligned && !s->has_area && !s->has_coplanar &&
                                   !s->seam_source;
        build_image<float>(d, wl, cdfs, cdf_off, mat_cdfD, mat_cdfA, cdf_wid_off, mat_abs_slot, absA, absD, src_cdf, s->energy_fx,
                           s->wid_cbin, (int)s->bins_used.size(), s->el_slot, s->n_dump, fs, plan, guide_bits, hist_may_move, img);
    };
    // The guide tables of the inverse-CDF picks get as many buckets as the LDS can hold without costing the float32
    // kernel a wave per SIMD (workgroups of 512, 768 or 1024 threads, RTMC_TRACE_WAVES waves at most) or moving the
    // detector histogram to HBM: more buckets, fewer steps of bisection (C2 +2 % with 2^12 instead of 2^8).
    auto waves_per_simd = [](size_t lds_bytes) { return trace_waves_per_simd(lds_bytes); };
    image32(GUIDE_BITS_MIN, s->img32);
    if (plan.n_regions > 0) {
        // the regions of a scene with many boxes (two records each): not at the price of a wave per SIMD, of the
        // detector histogram's place in the LDS, or of the scene's fitting at all
        const DeviceImage<float> with = s->img32;
        plan = find_cells(d, fs, false);
        image32(GUIDE_BITS_MIN, s->img32);
        const bool affordable = with.lay.total <= 160 * 1024 && with.lay.hist_in_hbm == s->img32.lay.hist_in_hbm &&
                                waves_per_simd(with.lay.total) >= waves_per_simd(s->img32.lay.total);
        if (getenv("RTMC_DEBUG"))
            fprintf(stderr, "[rtmc] regions: LDS image %u bytes with them (%d waves per SIMD, histogram in %s), %u without (%d, %s): %s\n",
                    with.lay.total, waves_per_simd(with.lay.total), with.lay.hist_in_hbm ? "HBM" : "LDS", s->img32.lay.total,
                    waves_per_simd(s->img32.lay.total), s->img32.lay.hist_in_hbm ? "HBM" : "LDS", affordable ? "kept" : "dropped");
        if (affordable) {
            plan = find_cells(d, fs, true);
            s->img32 = with;
        }
    }
    s->has_cells = plan.n_usable > 0;
    s->n_cells = (int)plan.cells.size() / 2 - 1; s->n_cells_usable = plan.n_usable; s->start_cell = plan.start_cell;
    s->n_regions = plan.n_regions;
    s->has_tiles = !plan.tiles.empty();
    s->guide_bits = GUIDE_BITS_MIN;
    {
        const char *forced = getenv("RTMC_GUIDE_BITS");   // experiments
        for (int bits = GUIDE_BITS_MAX; bits > GUIDE_BITS_MIN; bits -= 2) {
            if (forced && atoi(forced) != bits) continue;
            DeviceImage<float> trial;
            image32(bits, trial);
            const bool fits = trial.lay.total <= 160 * 1024 && trial.lay.hist_in_hbm == s->img32.lay.hist_in_hbm &&
                              waves_per_simd(trial.lay.total) >= waves_per_simd(s->img32.lay.total);
            if (fits || (forced && trial.lay.total <= 160 * 1024)) {
                s->img32 = std::move(trial);
                s->guide_bits = bits;
                break;
            }
        }
        if (forced && atoi(forced) == GUIDE_BITS_MIN) s->guide_bits = GUIDE_BITS_MIN;
    }
    build_image<double>(d, wl, cdfs, cdf_off, mat_cdfD, mat_cdfA, cdf_wid_off, mat_abs_slot, absA, absD, src_cdf, s->energy_fx, s->wid_cbin, (int)s->bins_used.size(), s->el_slot, s->n_dump, fs, plan, GUIDE_BITS_MIN, false, s->img64);
    s->has_back = s->img32.back_possible || s->img64.back_possible;
    s->has_frames = (s->img32.n_group[3] >> 16) != 0;
    s->has_runs = s->img32.has_runs;
    s->hist_in_hbm = s->img32.lay.hist_in_hbm || s->img64.lay.hist_in_hbm;
    if (s->img32.lay.total > 160 * 1024) {
        size_t need = s->img32.lay.total;
        return fail(RTMC_ELIMIT, "scene tables need %zu bytes of LDS, 163840 available", need);
    }
    *out = holder.release();
    return RTMC_OK;
}

static void free_device(PerDevice &p)
{
    if (p.ro_block) (void)hipFree(p.ro_block);
    if (p.rw_block) (void)hipFree(p.rw_block);
    if (p.host_tallies) (void)hipHostFree(p.host_tallies);
    if (p.private_hist) (void)hipFree(p.private_hist);
    if (p.ev0) (void)hipEventDestroy(p.ev0);
    if (p.ev1) (void)hipEventDestroy(p.ev1);
    if (p.stream) (void)hipStreamDestroy(p.stream);
    p = PerDevice();
}

RTMC_API void rtmc_scene_destroy(rtmc_scene *s)
{
    if (!s) return;
    for (auto &kv : s->dev) {
        if (hipSetDevice(kv.first) == hipSuccess) free_device(kv.second);
    }
    delete s;
}

RTMC_API int rtmc_scene_wavelengths(const rtmc_scene *s, int64_t *count, double *values, int64_t capacity)
{
    if (!s || !count) return fail(RTMC_EINVAL, "null argument");
    *count = (int64_t)s->wl.size();
    if (values) {
        if (capacity < *count) return fail(RTMC_EINVAL, "capacity %lld < %lld", (long long)capacity, (long long)*count);
        memcpy(values, s->wl.data(), sizeof(double) * s->wl.size());
    }
    return RTMC_OK;
}

static int upload_scene_structs(rtmc_scene *s, PerDevice &p);

// The trace kernel instantiation for a feature set.  float32: every combination
// of aligned dyes, other shapes, spectra and back transfer, plus the one with
// everything (launch area).  float64 (validation only): spectra and back
// transfer always on, which keeps the number of slow-to-compile kernels down.
template <typename real, int BASE>
static const void *trace_variant(int features)
{
    switch (features & (FEAT_ALIGNED | FEAT_SHAPES | FEAT_FRAMES)) {
    case 0: return (const void *)trace_kernel<real, BASE | 0>;
    case FEAT_ALIGNED: return (const void *)trace_kernel<real, BASE | FEAT_ALIGNED>;
    case FEAT_FRAMES: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES>;
    case FEAT_FRAMES | FEAT_ALIGNED: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_ALIGNED>;
    case FEAT_FRAMES | FEAT_SHAPES: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_SHAPES>;
    default: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_SHAPES | FEAT_ALIGNED>;
    }
}
template <typename real>
static const void *trace_function(int features)
{
    if (features & (FEAT_AREA | FEAT_GUARD)) return (const void *)trace_kernel<real, FEAT_ALL>;
    if (features & FEAT_HBMHIST) {   // many detectors: the kernel with everything, or the kernels with cells that count in HBM
        if (sizeof(real) == 8 || !(features & FEAT_CELLS)) return (const void *)trace_kernel<real, FEAT_ALL>;
        if constexpr (sizeof(real) == 4) {
            constexpr int BASE = FEAT_CELLS | FEAT_RUNS | FEAT_HBMHIST;
            if (features & FEAT_FRAMES) {   // ... in turned frames
                switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
                case 0: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES>;
                case FEAT_SPECTRA: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_SPECTRA>;
                case FEAT_BACK: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_BACK>;
                default: return (const void *)trace_kernel<real, BASE | FEAT_FRAMES | FEAT_SPECTRA | FEAT_BACK>;
                }
            }
            switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
            case 0: return (const void *)trace_kernel<real, BASE>;
            case FEAT_SPECTRA: return (const void *)trace_kernel<real, BASE | FEAT_SPECTRA>;
            case FEAT_BACK: return (const void *)trace_kernel<real, BASE | FEAT_BACK>;
            default: return (const void *)trace_kernel<real, BASE | FEAT_SPECTRA | FEAT_BACK>;
            }
        }
    }
    if constexpr (sizeof(real) == 8) {
        return trace_variant<real, FEAT_SPECTRA | FEAT_BACK>(features);   // (float64 tables hold no runs)
    } else {
        if ((features & FEAT_CELLS) && (features & FEAT_FRAMES)) {  // cells in turned frames (with the code for runs and tiled faces)
            switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
            case 0: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_FRAMES | FEAT_RUNS>;
            case FEAT_SPECTRA: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_FRAMES | FEAT_RUNS | FEAT_SPECTRA>;
            case FEAT_BACK: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_FRAMES | FEAT_RUNS | FEAT_BACK>;
            default: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_FRAMES | FEAT_RUNS | FEAT_SPECTRA | FEAT_BACK>;
            }
        }
        if ((features & FEAT_CELLS) && (features & FEAT_RUNS)) {  // cells, with runs in the hit table or tiled faces
            switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
            case 0: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_RUNS>;
            case FEAT_SPECTRA: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_RUNS | FEAT_SPECTRA>;
            case FEAT_BACK: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_RUNS | FEAT_BACK>;
            default: return (const void *)trace_kernel<real, FEAT_CELLS | FEAT_RUNS | FEAT_SPECTRA | FEAT_BACK>;
            }
        }
        if (features & FEAT_RUNS) {  // runs: the rectangle-only kernels of the world frame, else the one with everything
            if (features & (FEAT_ALIGNED | FEAT_SHAPES | FEAT_FRAMES)) return (const void *)trace_kernel<real, FEAT_ALL>;
            switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
            case 0: return (const void *)trace_kernel<real, FEAT_RUNS>;
            case FEAT_SPECTRA: return (const void *)trace_kernel<real, FEAT_RUNS | FEAT_SPECTRA>;
            case FEAT_BACK: return (const void *)trace_kernel<real, FEAT_RUNS | FEAT_BACK>;
            default: return (const void *)trace_kernel<real, FEAT_RUNS | FEAT_SPECTRA | FEAT_BACK>;
            }
        }
        if (features & FEAT_CELLS) {  // cells: rectangle-only scenes of the world frame (rtmc_scene::cells_in_use)
            const bool aligned = (features & FEAT_ALIGNED) != 0;
            switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
            case 0: return aligned ? (const void *)trace_kernel<real, FEAT_CELLS | FEAT_ALIGNED> : (const void *)trace_kernel<real, FEAT_CELLS>;
            case FEAT_SPECTRA: return aligned ? (const void *)trace_kernel<real, FEAT_CELLS | FEAT_SPECTRA | FEAT_ALIGNED>
                                              : (const void *)trace_kernel<real, FEAT_CELLS | FEAT_SPECTRA>;
            case FEAT_BACK: return aligned ? (const void *)trace_kernel<real, FEAT_CELLS | FEAT_BACK | FEAT_ALIGNED>
                                           : (const void *)trace_kernel<real, FEAT_CELLS | FEAT_BACK>;
            default: return aligned ? (const void *)trace_kernel<real, FEAT_CELLS | FEAT_SPECTRA | FEAT_BACK | FEAT_ALIGNED>
                                    : (const void *)trace_kernel<real, FEAT_CELLS | FEAT_SPECTRA | FEAT_BACK>;
            }
        }
        switch (features & (FEAT_SPECTRA | FEAT_BACK)) {
        case 0: return trace_variant<real, 0>(features);
        case FEAT_SPECTRA: return trace_variant<real, FEAT_SPECTRA>(features);
        case FEAT_BACK: return trace_variant<real, FEAT_BACK>(features);
        default: return trace_variant<real, FEAT_SPECTRA | FEAT_BACK>(features);
        }
    }
}

RTMC_API int rtmc_scene_cells(const rtmc_scene *s, int32_t info[4])
{
    if (!s || !info) return fail(RTMC_EINVAL, "null argument");
    info[0] = s->n_cells | (s->n_regions << 16);
    info[1] = s->n_cells_usable;
    info[2] = s->start_cell;
    info[3] = s->features();
    return RTMC_OK;
}

RTMC_API int rtmc_scene_info(const rtmc_scene *s, int32_t info[4])
{
    if (!s || !info) return fail(RTMC_EINVAL, "null argument");
    info[0] = s->img32.n_runs;
    info[1] = s->img32.n_run_members;
    info[2] = s->has_coplanar ? 1 : (s->seam_source ? 2 : 0);
    info[3] = trace_function<float>(s->features()) == (const void *)trace_kernel<float, FEAT_ALL> ? 1 : 0;
    return RTMC_OK;
}

// Puts the caller's current device back when an entry point returns.
struct DeviceGuard {
    int previous = -1;
    DeviceGuard() { if (hipGetDevice(&previous) != hipSuccess) previous = -1; }
    ~DeviceGuard() { if (previous >= 0) (void)hipSetDevice(previous); }
};

// upload the scene to `device` on first use; leaves that device current
static int init_device(rtmc_scene *s, int device, PerDevice &p);

static int prepare_device(rtmc_scene *s, int device, PerDevice **out)
{
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count || device >= MAX_DEVICES)
        return fail(RTMC_EINVAL, "device %d out of range (%d visible, at most %d supported)", device, count, MAX_DEVICES);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(s->mu);
    PerDevice &p = s->dev[device];
    *out = &p;
    if (p.ready) return RTMC_OK;
    const int rc = init_device(s, device, p);
    if (rc) free_device(p);  // nothing half-built stays behind; the next call starts over
    return rc;
}

static int init_device(rtmc_scene *s, int device, PerDevice &p)
{
    {
        const int rc = setup_gpu_shared(device);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&p.ev0));
    HIP_TRY(hipEventCreate(&p.ev1));
    // Two allocations per device: one for everything the kernels only read, one
    // for what they update with atomics (tallies, work counter).  Sharing a
    // cache line between the two makes every atomic evict read-only data that
    // the scalar loads of all waves keep re-reading.
    auto round_up = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
    struct Piece { void **dst; const void *src; size_t bytes; };
    const Piece pieces[] = {
        {&p.elems32, s->img32.elems.data(), sizeof(DevElem<float>) * s->n_el},
        {&p.elems64, s->img64.elems.data(), sizeof(DevElem<double>) * s->n_el},
        {&p.hits32, s->img32.hits.data(), sizeof(HitChunk<float>) * s->img32.hits.size()},
        {&p.hits64, s->img64.hits.data(), sizeof(HitChunk<double>) * s->img64.hits.size()},
        {&p.lds32, s->img32.lds.data(), s->img32.lds.size()},
        {&p.lds64, s->img64.lds.data(), s->img64.lds.size()},
        {(void **)&p.wl, s->wl.data(), sizeof(double) * s->wl.size()},
        {(void **)&p.eqeT, s->eqeT.data(), 4 * s->eqeT.size()},
        {&p.scene32, nullptr, sizeof(DevScene<float>)},
        {&p.scene64, nullptr, sizeof(DevScene<double>)},
    };
    size_t ro_bytes = 0;
    for (const Piece &pc : pieces) ro_bytes += round_up(pc.bytes);
    HIP_TRY(hipMalloc(&p.ro_block, ro_bytes));
    size_t off = 0;
    for (const Piece &pc : pieces) {
        *pc.dst = (char *)p.ro_block + off;
        if (pc.src && pc.bytes) HIP_TRY(hipMemcpy(*pc.dst, pc.src, pc.bytes, hipMemcpyHostToDevice));
        off += round_up(pc.bytes);
    }
    p.tally_words = ACC_SCALARS + 3 * (size_t)s->n_el + (size_t)std::max(1, s->n_dump) * s->bins_used.size();
    HIP_TRY(hipMalloc(&p.rw_block, round_up(8 * p.tally_words) + 4096));
    p.tallies = (unsigned long long *)p.rw_block;
    p.work = (unsigned long long *)((char *)p.rw_block + round_up(8 * p.tally_words));
    HIP_TRY(hipHostMalloc((void **)&p.host_tallies, 8 * p.tally_words, hipHostMallocDefault));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    p.n_cu = prop.multiProcessorCount;
    const void *trace32 = trace_function<float>(s->features());
    const void *trace64 = trace_function<double>(s->features());
    // The attribute belongs to the function, not to the scene: every scene of the process shares
    // these kernels, so each is allowed the whole LDS of a CU once and for all (a later, smaller
    // scene must not lower the limit under an earlier, larger one).
    const int lds_limit = 160 * 1024;
    for (const void *fn : {trace32, trace64, (const void *)step_tape_kernel<float>, (const void *)photon_op_kernel<float>,
                           (const void *)step_tape_kernel<double>, (const void *)photon_op_kernel<double>,
                           (const void *)preset_hit_kernel<0>, (const void *)preset_hit_kernel<FEAT_RUNS>,
                           (const void *)preset_hit_kernel<FEAT_RUNS | FEAT_FRAMES>})
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit));
    {  // the workgroup size that puts the most waves on a SIMD; the smaller one on a tie
        int best_waves = -1;
        for (int block = TRACE_BLOCK; block <= TRACE_BLOCK_MAX; block += 256) {
            int n = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace32, block, s->img32.lay.total));
            if (n * block > best_waves) { best_waves = n * block; p.max_blocks32 = n; p.block32 = block; }
        }
        if (const char *e = getenv("RTMC_TRACE_THREADS")) {  // experiments
            const int block = atoi(e);
            if (block >= 256 && block <= TRACE_BLOCK_MAX && block % 256 == 0) {
                p.block32 = block;
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.max_blocks32, trace32, block, s->img32.lay.total));
            }
        }
    }
    if (s->img64.lay.total <= (size_t)lds_limit)
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.max_blocks64, trace64, TRACE_BLOCK, s->img64.lay.total));
    if (s->hist_in_hbm) {  // one private histogram per workgroup that can be resident, in either precision
        p.private_hist_copies = (size_t)p.n_cu * (size_t)std::max(1, std::max(p.max_blocks32, p.max_blocks64));
        const size_t words = (size_t)std::max(1, s->n_dump) * s->bins_used.size();
        HIP_TRY(hipMalloc((void **)&p.private_hist, 4 * words * p.private_hist_copies));
    }
    {
        int rc = upload_scene_structs(s, p);
        if (rc) return rc;
    }
    // kernels with cells, strict point source: the first hit of every photon, found once by the search itself
    if ((s->features() & FEAT_CELLS) && s->img32.src.dir_mode == 0 && !s->has_area && !getenv("RTMC_NO_PRESET")) {
        CellHeader *hdr = (CellHeader *)((char *)p.lds32 + s->img32.lay.cells);
        const int search = s->features() & (FEAT_RUNS | FEAT_FRAMES);
        if (search & FEAT_FRAMES)
            hipLaunchKernelGGL(preset_hit_kernel<FEAT_RUNS | FEAT_FRAMES>, dim3(1), dim3(64), s->img32.lay.total, p.stream,
                               (const DevScene<float> *)p.scene32, hdr);
        else if (search & FEAT_RUNS)
            hipLaunchKernelGGL(preset_hit_kernel<FEAT_RUNS>, dim3(1), dim3(64), s->img32.lay.total, p.stream,
                               (const DevScene<float> *)p.scene32, hdr);
        else
            hipLaunchKernelGGL(preset_hit_kernel<0>, dim3(1), dim3(64), s->img32.lay.total, p.stream,
                               (const DevScene<float> *)p.scene32, hdr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(p.stream));
    }
    p.ready = true;
    return RTMC_OK;
}

template <typename real>
static DevScene<real> device_scene(const rtmc_scene *s, const PerDevice &p, const DeviceImage<real> &img);

static int upload_scene_structs(rtmc_scene *s, PerDevice &p)
{
    DevScene<float> a = device_scene<float>(s, p, s->img32);
    DevScene<double> b = device_scene<double>(s, p, s->img64);
    HIP_TRY(hipMemcpy(p.scene32, &a, sizeof a, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.scene64, &b, sizeof b, hipMemcpyHostToDevice));
    return RTMC_OK;
}

template <typename real>
static DevScene<real> device_scene(const rtmc_scene *s, const PerDevice &p, const DeviceImage<real> &img)
{
    DevScene<real> sc;
    sc.hits = (const HitChunk<real> *)(sizeof(real) == 4 ? p.hits32 : p.hits64);
    for (int k = 0; k < 4; k++) sc.n_group[k] = img.n_group[k];
    sc.elems = (const DevElem<real> *)(sizeof(real) == 4 ? p.elems32 : p.elems64);
    sc.lds_image = (const uint32_t *)(sizeof(real) == 4 ? p.lds32 : p.lds64);
    sc.wl = p.wl;
    sc.eqeT = p.eqeT;
    sc.src = img.src;
    sc.lay = img.lay;
    sc.n_el = s->n_el; sc.n_mat = s->n_mat; sc.max_iter = s->max_iter;
    sc.use_guard = s->has_coplanar ? 1 : (s->seam_source ? 2 : 0);
    return sc;
}

RTMC_API double rtmc_last_kernel_ms(const rtmc_scene *s, int device);

// A workgroup counts detector hits in 32-bit LDS words; a launch is kept small
// enough that no workgroup can come near 2^32 photons.
static const uint64_t MAX_PHOTONS_PER_LAUNCH = 100000000000ull;

static uint64_t photons_per_launch_cap()
{
    uint64_t cap = MAX_PHOTONS_PER_LAUNCH;
    if (const char *e = getenv("RTMC_MAX_PHOTONS_PER_LAUNCH"))  // tests exercise the splitting with a small value
        cap = std::max<uint64_t>(1, std::min<uint64_t>(cap, strtoull(e, nullptr, 10)));
    return cap;
}

static int check_run_arguments(const rtmc_scene *s, int precision, uint64_t first_photon, uint64_t n_photons,
                               const rtmc_tallies *out)
{
    if (!s || !out) return fail(RTMC_EINVAL, "null argument");
    if (out->n_elements != s->n_el) return fail(RTMC_EINVAL, "tally block is for %d elements, scene has %d", out->n_elements, s->n_el);
    if (!out->counter || !out->spectral || !out->bounce || !out->energy_fx || !out->spectrum)
        return fail(RTMC_EINVAL, "tally block has a null array");
    if (precision != RTMC_F32 && precision != RTMC_F64) return fail(RTMC_EINVAL, "unknown precision %d", precision);
    if (first_photon + n_photons < first_photon) return fail(RTMC_EINVAL, "photon id range wraps");
    return RTMC_OK;
}

// One launch of the trace kernel with the device tally block at `dev_tallies`
// (p.tally_words words) and the work counter at `dev_work`, then the copy-back
// through the scene's pinned staging buffer and the fold into the caller's
// block.  The device of `p` is current.  Allocates nothing.
static int launch_and_fold(rtmc_scene *s, PerDevice &p, int device, unsigned long long *dev_tallies,
                           unsigned long long *dev_work, int precision, uint64_t seed, uint64_t first_photon,
                           uint64_t n_photons, rtmc_tallies *out, double *kernel_ms)
{
    const bool f32 = precision == RTMC_F32;
    const size_t lds_bytes = f32 ? s->img32.lay.total : s->img64.lay.total;
    const int per_cu = f32 ? p.max_blocks32 : p.max_blocks64;
    if (per_cu < 1) return fail(RTMC_ELIMIT, "scene tables (%zu bytes) do not fit the LDS in this precision", lds_bytes);

    DevTallies tl;
    tl.scalars = dev_tallies;
    tl.bounce = tl.scalars + ACC_SCALARS;
    tl.spectral = tl.bounce + s->n_el;
    tl.energy_fx = tl.spectral + s->n_el;
    tl.hist = tl.energy_fx + s->n_el;
    tl.private_hist = p.private_hist;
    const bool private_hist = (f32 ? s->img32.lay.hist_in_hbm : s->img64.lay.hist_in_hbm) != 0;
    const size_t hist_words = (size_t)std::max(1, s->n_dump) * s->bins_used.size();

    // enough waves to fill the machine, never more than there are blocks of ids
    uint64_t wanted = (n_photons + 63) / 64;                          // waves
    const int block = f32 ? p.block32 : TRACE_BLOCK;
    wanted = (wanted + block / 64 - 1) / (block / 64);                // workgroups
    uint64_t resident = (uint64_t)p.n_cu * (uint64_t)per_cu;
    if (const char *cap = getenv("RTMC_WG_PER_CU"))  // experiments: fewer resident workgroups than fit
        resident = (uint64_t)p.n_cu * (uint64_t)std::max(1, std::min(per_cu, atoi(cap)));
    unsigned grid = (unsigned)std::max<uint64_t>(1, std::min(wanted, resident));

    if (getenv("RTMC_DEBUG"))
        fprintf(stderr, "[rtmc] trace launch: %u workgroups of %d threads (%d per CU x %d CUs), %zu B LDS (guide tables of 2^%d buckets), "
                        "%llu photons\n",
                grid, block, per_cu, p.n_cu, lds_bytes, f32 ? s->guide_bits : GUIDE_BITS_MIN, (unsigned long long)n_photons);
    std::lock_guard<std::mutex> serial(g_gpu[device].launch_mu);
    HIP_TRY(hipMemsetAsync(dev_tallies, 0, 8 * p.tally_words, p.stream));
#ifdef RTMC_PROFILE_SECTIONS
    HIP_TRY(hipMemsetAsync(dev_work, 0, 512, p.stream));
#else
    HIP_TRY(hipMemsetAsync(dev_work, 0, 8, p.stream));
#endif
    if (private_hist) {
        if (grid > p.private_hist_copies) return fail(RTMC_EDEVICE, "more workgroups than private histograms");
        HIP_TRY(hipMemsetAsync(p.private_hist, 0, 4 * hist_words * grid, p.stream));
    }
    HIP_TRY(hipEventRecord(p.ev0, p.stream));
    {
        const void *scene_ptr = f32 ? p.scene32 : p.scene64;
        unsigned long long *work = dev_work;
        void *args[] = {&scene_ptr, &tl, &seed, &first_photon, &n_photons, &work};
        const void *fn = f32 ? trace_function<float>(s->features()) : trace_function<double>(s->features());
        HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(block), args, lds_bytes, p.stream));
    }
    HIP_TRY(hipGetLastError());
    if (private_hist) {
        hipLaunchKernelGGL(sum_private_hist_kernel, dim3((unsigned)((hist_words + 255) / 256)), dim3(256), 0, p.stream,
                           (const uint32_t *)p.private_hist, (uint32_t)grid, (uint32_t)hist_words, tl.hist);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(p.ev1, p.stream));
    HIP_TRY(hipMemcpyAsync(p.host_tallies, dev_tallies, 8 * p.tally_words, hipMemcpyDeviceToHost, p.stream));
    HIP_TRY(hipStreamSynchronize(p.stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p.ev0, p.ev1));
    *kernel_ms += ms;
#ifdef RTMC_PROFILE_SECTIONS
    {
        unsigned long long w[64];
        HIP_TRY(hipMemcpy(w, dev_work, sizeof w, hipMemcpyDeviceToHost));
        const double total = (double)(w[8] + w[11]);
        fprintf(stderr, "[rtmc] sections: renew %llu passes, %.0f cycles each, %.1f lanes; step %llu passes, %.0f cycles each, %.1f lanes; "
                        "renew share of wave time %.3f; lane utilisation %.3f\n",
                w[9], (double)w[8] / (double)std::max(1ull, w[9]), (double)w[10] / (double)std::max(1ull, w[9]),
                w[12], (double)w[11] / (double)std::max(1ull, w[12]), (double)w[13] / (double)std::max(1ull, w[12]),
                (double)w[8] / total,
                ((double)w[8] * (double)w[10] / (double)std::max(1ull, w[9]) +
                 (double)w[11] * (double)w[13] / (double)std::max(1ull, w[12])) / (64.0 * total));
        fprintf(stderr, "[rtmc]   shader clock while the waves ran: %.0f MHz (cycle counter against the 100 MHz counter, all waves)\n",
                100.0 * (double)w[44] / (double)std::max(1ull, w[45]));
        if (s->features() & FEAT_CELLS)
            fprintf(stderr, "[rtmc]   step passes that walked the element list: %llu of %llu\n", w[14], w[12]);
        static const char *names[9] = {"step: philox", "step: search", "step: absorption check", "step: element operation",
                                       "renew: endings", "renew: pathway + wavelength", "renew: photon ids",
                                       "renew: philox", "renew: direction, polarisation, free path"};
        for (int k = 0; k < 9; k++) {
            // cycles per pass of a lane that took part; lanes per pass; share of all wave time
            const double per = (double)w[16 + k] / (double)std::max(1ull, w[32 + k]);
            const double passes = (double)(k < 4 ? w[12] : w[9]);
            const double lanes = (double)w[32 + k] / std::max(1.0, passes);
            fprintf(stderr, "[rtmc]   %-42s %6.0f cycles, %4.1f lanes, %4.1f %% of wave time\n", names[k], per, lanes,
                    100.0 * per * passes / total);
        }
    }
#endif

    const unsigned long long *sc = p.host_tallies;
    const unsigned long long *bounce = sc + ACC_SCALARS, *spectral = bounce + s->n_el,
                             *energy = spectral + s->n_el, *hist = energy + s->n_el;
    const size_t n_bins = s->bins_used.size();
    unsigned long long dumped = 0;
    for (int e = 0; e < s->n_el; e++) {
        out->bounce[e] += bounce[e];
        out->energy_fx[e] += energy[e];
        if (s->el_slot[e] < 0) continue;
        unsigned long long total = 0;
        for (size_t b = 0; b < n_bins; b++) {
            unsigned long long c = hist[(size_t)s->el_slot[e] * n_bins + b];
            if (!c) continue;
            total += c;
            out->spectrum[(size_t)e * RTMC_SPECTRUM_BINS + s->bins_used[b]] += c;
        }
        out->counter[e] += total;
        dumped += total;
        // Elements.py:161-165: without a detection table every photon counts
        out->spectral[e] += s->el_eqe[e] >= 0 ? spectral[e] : total;
    }
    out->n_photons += sc[ACC_IC] + dumped + sc[ACC_CAPPED] + sc[ACC_LOST];
    out->n_steps += sc[ACC_STEPS];
    out->n_absorbed_once += sc[ACC_ABS_ONCE];
    out->n_absorptions += sc[ACC_ABSORPTIONS];
    out->n_internal_converted += sc[ACC_IC];
    out->n_dumped += dumped;
    out->n_capped += sc[ACC_CAPPED];
    out->n_lost += sc[ACC_LOST];
    out->n_stale_steps += sc[ACC_STALE];
    return RTMC_OK;
}

static void set_last_kernel_ms(rtmc_scene *s, PerDevice &p, double ms)
{
    std::lock_guard<std::mutex> lock(s->mu);
    p.last_ms = ms;
}

RTMC_API int rtmc_run(rtmc_scene *s, int device, int precision, uint64_t seed, uint64_t first_photon,
                        uint64_t n_photons, rtmc_tallies *out)
{
    return guarded([&]() -> int {
        int rc = check_run_arguments(s, precision, first_photon, n_photons, out);
        if (rc) return rc;
        DeviceGuard restore_device;
        PerDevice *pp;
        if ((rc = prepare_device(s, device, &pp))) return rc;
        const uint64_t cap = photons_per_launch_cap();
        double ms = 0;
        for (uint64_t done = 0; done < n_photons;) {
            const uint64_t part = std::min(cap, n_photons - done);
            rc = launch_and_fold(s, *pp, device, pp->tallies, pp->work, precision, seed, first_photon + done, part, out, &ms);
            if (rc) return rc;
            done += part;
        }
        if (n_photons) set_last_kernel_ms(s, *pp, ms);
        return RTMC_OK;
    });
}

static int run_noalloc_impl(rtmc_scene *s, int device, int precision, uint64_t seed, uint64_t first_photon,
                            uint64_t n_photons, rtmc_tallies *out)
{
    int rc = check_run_arguments(s, precision, first_photon, n_photons, out);
    if (rc) return rc;
    DeviceGuard restore_device;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count || device >= MAX_DEVICES)
        return fail(RTMC_EINVAL, "device %d out of range (%d visible, at most %d supported)", device, count, MAX_DEVICES);
    HIP_TRY(hipSetDevice(device));
    PerDevice *pp = nullptr;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        auto it = s->dev.find(device);  // never inserts
        if (it != s->dev.end() && it->second.ready) pp = &it->second;
    }
    unsigned long long *base = nullptr;
    {   // (another thread may be uploading its first scene to this GPU right now)
        std::lock_guard<std::mutex> lock(g_gpu_setup_mu);
        base = g_gpu[device].scratch_ptr;
    }
    if (!pp || !base)
        return fail(RTMC_EINVAL, "scene is not resident on device %d: rtmc_run (for zero photons, if nothing is to be "
                                 "traced yet) uploads it, rtmc_run_noalloc never allocates", device);
    if (pp->tally_words > SCRATCH_TALLY_WORDS)
        return fail(RTMC_ELIMIT, "the tally block of this scene has %zu words, the static scratch of rtmc_run_noalloc "
                                 "holds %zu (elements, detectors x wavelength bins): use rtmc_run",
                    pp->tally_words, SCRATCH_TALLY_WORDS);
    const uint64_t cap = photons_per_launch_cap();
    double ms = 0;
    for (uint64_t done = 0; done < n_photons;) {
        const uint64_t part = std::min(cap, n_photons - done);
        rc = launch_and_fold(s, *pp, device, base, base + SCRATCH_WORK_OFFSET, precision, seed, first_photon + done, part, out, &ms);
        if (rc) return rc;
        done += part;
    }
    if (n_photons) set_last_kernel_ms(s, *pp, ms);
    return RTMC_OK;
}

RTMC_API int rtmc_run_noalloc(rtmc_scene *s, int device, int precision, uint64_t seed, uint64_t first_photon,
                                uint64_t n_photons, rtmc_tallies *out)
{
    // (nothing in the body allocates, so nothing in it throws; the wrapper keeps the rule that no entry
    // point lets anything propagate)
    return guarded([&] { return run_noalloc_impl(s, device, precision, seed, first_photon, n_photons, out); });
}

RTMC_API double rtmc_last_kernel_ms(const rtmc_scene *s, int device)
{
    if (!s) return -1.0;
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->dev.find(device);
    return it == s->dev.end() ? -1.0 : it->second.last_ms;
}

namespace {
struct DeviceBuffer {
    void *ptr = nullptr;
    ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
};
}  // namespace

struct PhotonOp {  // rtmc_op_photon's part of a step_tape_impl call
    int op = 0;
    const double *args = nullptr;
    double *out = nullptr;
    int uniform_tape = 0;
};
static int step_tape_impl(rtmc_scene *s, int device, int precision, uint32_t n, rtmc_photon_soa *b,
                          const double *tape, const uint64_t *tape_off, uint32_t n_steps, int launch,
                          double *trace, int32_t *cross_el, double *cross_dist, double *cross_pts,
                          const PhotonOp &po = PhotonOp());

RTMC_API int rtmc_step_tape(rtmc_scene *s, int device, int precision, uint32_t n, rtmc_photon_soa *b,
                              const double *tape, const uint64_t *tape_off, uint32_t n_steps, int launch,
                              double *trace)
{
    return guarded([&] {
        return step_tape_impl(s, device, precision, n, b, tape, tape_off, n_steps, launch, trace, nullptr, nullptr, nullptr);
    });
}

RTMC_API int rtmc_op_crossing(rtmc_scene *s, int device, int precision, uint32_t n, const rtmc_photon_soa *b,
                                int32_t *element, double *distance, double *points)
{
    if (!element || !distance || !points) return fail(RTMC_EINVAL, "null argument");
    if (n == 0) return RTMC_OK;
    return guarded([&] {
        const double no_tape = 0.0;
        std::vector<uint64_t> off(n + 1, 0);
        return step_tape_impl(s, device, precision, n, const_cast<rtmc_photon_soa *>(b), &no_tape, off.data(), 0, 0,
                              nullptr, element, distance, points);
    });
}

RTMC_API int rtmc_op_photon(rtmc_scene *s, int device, int precision, int op, uint32_t n, rtmc_photon_soa *b,
                              const double *args, double *out, const double *tape, const uint64_t *tape_off,
                              int uniform_tape)
{
    if (op < RTMC_PHOP_GENERATE || op > RTMC_PHOP_FREEPATH) return fail(RTMC_EINVAL, "unknown photon operation %d", op);
    if (op != RTMC_PHOP_GENERATE && !args) return fail(RTMC_EINVAL, "null argument");
    if (op == RTMC_PHOP_REFLECTANCE && !out) return fail(RTMC_EINVAL, "this operation needs an output array");
    if (n == 0) return RTMC_OK;
    return guarded([&] {
        if (op == RTMC_PHOP_EXECUTE && s)
            for (uint32_t k = 0; k < n; k++) {
                const double e = args[(size_t)k * RTMC_PHOP_ARGS + 6];
                if (!(e >= 0 && e < s->n_el) || e != (double)(int)e) return fail(RTMC_EINVAL, "photon %u: element out of range", k);
            }
        PhotonOp po;
        po.op = op; po.args = args; po.out = out; po.uniform_tape = uniform_tape;
        return step_tape_impl(s, device, precision, n, b, tape, tape_off, 0, op == RTMC_PHOP_GENERATE, nullptr, nullptr,
                              nullptr, nullptr, po);
    });
}

static int step_tape_impl(rtmc_scene *s, int device, int precision, uint32_t n, rtmc_photon_soa *b,
                          const double *tape, const uint64_t *tape_off, uint32_t n_steps, int launch,
                          double *trace, int32_t *cross_el, double *cross_dist, double *cross_pts, const PhotonOp &po)
{
    const bool crossing = cross_el != nullptr;
    if (!s || !b || !tape || !tape_off) return fail(RTMC_EINVAL, "null argument");
    if (precision != RTMC_F32 && precision != RTMC_F64) return fail(RTMC_EINVAL, "unknown precision %d", precision);
    if (n == 0) return RTMC_OK;
    {   // the columns this call reads or writes must exist
        bool ok = b->last_element && b->was_element && b->reflected;
        for (int j = 0; j < 3; j++) ok = ok && b->loc[j] && b->dir[j] && b->old_point[j];
        if (!crossing) {
            ok = ok && b->wavelength && b->start_wavelength && b->dist_left && b->hit_distance && b->psi && b->material &&
                 b->status && b->iterations && b->tape_pos;
            for (int j = 0; j < 3; j++) ok = ok && b->efield[j];
        }
        if (!ok) return fail(RTMC_EINVAL, "photon batch has a null column");
    }
    for (uint32_t k = 0; k < n; k++) {
        if (tape_off[k + 1] < tape_off[k]) return fail(RTMC_EINVAL, "tape offsets must ascend");
        if (tape_off[k + 1] - tape_off[k] > 0x7FFFFFFFull) return fail(RTMC_ELIMIT, "tape segment too long");
    }
    DeviceGuard restore_device;
    PerDevice *pp;
    int rc = prepare_device(s, device, &pp);
    if (rc) return rc;
    PerDevice &p = *pp;
    const bool f32 = precision == RTMC_F32;
    const size_t lds_bytes = f32 ? s->img32.lay.acc : s->img64.lay.acc;
    if (lds_bytes > 160 * 1024) return fail(RTMC_ELIMIT, "scene tables do not fit the LDS in this precision");

    // wavelengths -> ids
    std::vector<int32_t> wid(n, 0), wid0(n, 0);
    if (!launch) {
        std::map<double, int32_t> ids;
        for (size_t w = s->wl.size(); w-- > 0;) ids[s->wl[w]] = (int32_t)w;
        for (uint32_t k = 0; k < n; k++) {
            if (!crossing) {  // the search does not look at wavelengths
                auto a = ids.find(b->wavelength[k]), c = ids.find(b->start_wavelength[k]);
                if (a == ids.end() || c == ids.end())
                    return fail(RTMC_EINVAL, "photon %u carries a wavelength that is neither a source line nor an emission-table row", k);
                wid[k] = a->second; wid0[k] = c->second;
            }
            if (!crossing && (b->material[k] < 0 || b->material[k] >= s->n_mat)) return fail(RTMC_EINVAL, "photon %u: material out of range", k);
            if (b->last_element[k] < -1 || b->last_element[k] >= s->n_el) return fail(RTMC_EINVAL, "photon %u: element out of range", k);
            uint64_t seg = tape_off[k + 1] - tape_off[k];
            if (!crossing && (b->tape_pos[k] < 0 || (uint64_t)b->tape_pos[k] > seg)) return fail(RTMC_EINVAL, "photon %u: tape position outside its segment", k);
        }
    }
    const size_t nd = 15, ni = 9;  // double / int32 columns
    const size_t tape_len = tape_off[n];
    DeviceBuffer dbuf, ibuf, tbuf, obuf, trbuf, ovbuf, cebuf, cdbuf, cpbuf, oabuf, oobuf;
    if (po.args) HIP_TRY(hipMalloc(&oabuf.ptr, 8 * (size_t)RTMC_PHOP_ARGS * n));
    if (po.out) HIP_TRY(hipMalloc(&oobuf.ptr, 16 * (size_t)n));
    if (crossing) {
        HIP_TRY(hipMalloc(&cebuf.ptr, 4 * (size_t)n));
        HIP_TRY(hipMalloc(&cdbuf.ptr, 8 * (size_t)n));
        HIP_TRY(hipMalloc(&cpbuf.ptr, 24 * (size_t)n));
    }
    HIP_TRY(hipMalloc(&dbuf.ptr, 8 * nd * n));
    HIP_TRY(hipMalloc(&ibuf.ptr, 4 * ni * n));
    HIP_TRY(hipMalloc(&tbuf.ptr, 8 * std::max<size_t>(tape_len, 1)));
    HIP_TRY(hipMalloc(&obuf.ptr, 8 * (n + 1)));
    HIP_TRY(hipMalloc(&ovbuf.ptr, 4));
    if (trace) HIP_TRY(hipMalloc(&trbuf.ptr, 8 * (size_t)n * n_steps * RTMC_TRACE_STRIDE));
    double *dd = (double *)dbuf.ptr;
    int32_t *ii = (int32_t *)ibuf.ptr;
    SoaDev sd;
    const double *hsrc[nd];
    for (int j = 0; j < 3; j++) {
        sd.loc[j] = dd + (size_t)(0 + j) * n; hsrc[0 + j] = b->loc[j];
        sd.dir[j] = dd + (size_t)(3 + j) * n; hsrc[3 + j] = b->dir[j];
        sd.efield[j] = dd + (size_t)(6 + j) * n; hsrc[6 + j] = b->efield[j];
        sd.old_point[j] = dd + (size_t)(9 + j) * n; hsrc[9 + j] = b->old_point[j];
    }
    sd.dist_left = dd + (size_t)12 * n; hsrc[12] = b->dist_left;
    sd.hit_distance = dd + (size_t)13 * n; hsrc[13] = b->hit_distance;
    sd.psi = dd + (size_t)14 * n; hsrc[14] = b->psi;
    int32_t *icols[ni];
    for (size_t j = 0; j < ni; j++) icols[j] = ii + j * n;
    sd.wid = icols[0]; sd.wid0 = icols[1]; sd.material = icols[2]; sd.status = icols[3];
    sd.last_element = icols[4]; sd.was_element = icols[5]; sd.reflected = icols[6];
    sd.iterations = icols[7]; sd.tape_pos = icols[8];
    const int32_t *isrc[ni] = {wid.data(), wid0.data(), b->material, b->status, b->last_element,
                               b->was_element, b->reflected, b->iterations, b->tape_pos};
    std::lock_guard<std::mutex> serial(g_gpu[device].launch_mu);
    if (crossing) {  // only position, direction, old point and the three flags of the last element are read
        HIP_TRY(hipMemsetAsync(dd, 0, 8 * nd * (size_t)n, p.stream));
        HIP_TRY(hipMemsetAsync(ii, 0, 4 * ni * (size_t)n, p.stream));
        for (size_t j : {0, 1, 2, 3, 4, 5, 9, 10, 11}) HIP_TRY(hipMemcpyAsync(dd + j * n, hsrc[j], 8 * (size_t)n, hipMemcpyHostToDevice, p.stream));
        for (size_t j : {4, 5, 6}) HIP_TRY(hipMemcpyAsync(icols[j], isrc[j], 4 * (size_t)n, hipMemcpyHostToDevice, p.stream));
    } else if (!launch) {
        for (size_t j = 0; j < nd; j++) HIP_TRY(hipMemcpyAsync(dd + j * n, hsrc[j], 8 * (size_t)n, hipMemcpyHostToDevice, p.stream));
        for (size_t j = 0; j < ni; j++) HIP_TRY(hipMemcpyAsync(icols[j], isrc[j], 4 * (size_t)n, hipMemcpyHostToDevice, p.stream));
    }
    HIP_TRY(hipMemcpyAsync(tbuf.ptr, tape, 8 * tape_len, hipMemcpyHostToDevice, p.stream));
    HIP_TRY(hipMemcpyAsync(obuf.ptr, tape_off, 8 * (size_t)(n + 1), hipMemcpyHostToDevice, p.stream));
    HIP_TRY(hipMemsetAsync(ovbuf.ptr, 0, 4, p.stream));
    if (po.args) HIP_TRY(hipMemcpyAsync(oabuf.ptr, po.args, 8 * (size_t)RTMC_PHOP_ARGS * n, hipMemcpyHostToDevice, p.stream));
    if (po.out) HIP_TRY(hipMemsetAsync(oobuf.ptr, 0, 16 * (size_t)n, p.stream));
    if (trace) HIP_TRY(hipMemcpyAsync(trbuf.ptr, trace, 8 * (size_t)n * n_steps * RTMC_TRACE_STRIDE, hipMemcpyHostToDevice, p.stream));
    const unsigned grid = (n + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if (po.op != 0) {
        if (f32)
            hipLaunchKernelGGL(photon_op_kernel<float>, dim3(grid), dim3(TRACE_BLOCK), lds_bytes, p.stream,
                               (const DevScene<float> *)p.scene32, sd, n, po.op, (const double *)oabuf.ptr,
                               (double *)oobuf.ptr, (const double *)tbuf.ptr, (const unsigned long long *)obuf.ptr,
                               (int *)ovbuf.ptr, po.uniform_tape);
        else
            hipLaunchKernelGGL(photon_op_kernel<double>, dim3(grid), dim3(TRACE_BLOCK), lds_bytes, p.stream,
                               (const DevScene<double> *)p.scene64, sd, n, po.op, (const double *)oabuf.ptr,
                               (double *)oobuf.ptr, (const double *)tbuf.ptr, (const unsigned long long *)obuf.ptr,
                               (int *)ovbuf.ptr, po.uniform_tape);
    } else if (f32)
        hipLaunchKernelGGL(step_tape_kernel<float>, dim3(grid), dim3(TRACE_BLOCK), lds_bytes, p.stream,
                           (const DevScene<float> *)p.scene32, sd, n, n_steps, launch, (const double *)tbuf.ptr,
                           (const unsigned long long *)obuf.ptr, (double *)trbuf.ptr, (int *)ovbuf.ptr,
                           (int32_t *)cebuf.ptr, (double *)cdbuf.ptr, (double *)cpbuf.ptr);
    else
        hipLaunchKernelGGL(step_tape_kernel<double>, dim3(grid), dim3(TRACE_BLOCK), lds_bytes, p.stream,
                           (const DevScene<double> *)p.scene64, sd, n, n_steps, launch, (const double *)tbuf.ptr,
                           (const unsigned long long *)obuf.ptr, (double *)trbuf.ptr, (int *)ovbuf.ptr,
                           (int32_t *)cebuf.ptr, (double *)cdbuf.ptr, (double *)cpbuf.ptr);
    HIP_TRY(hipGetLastError());
    if (crossing) {  // the batch itself is left as it was
        HIP_TRY(hipMemcpyAsync(cross_el, cebuf.ptr, 4 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
        HIP_TRY(hipMemcpyAsync(cross_dist, cdbuf.ptr, 8 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
        HIP_TRY(hipMemcpyAsync(cross_pts, cpbuf.ptr, 24 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
        HIP_TRY(hipStreamSynchronize(p.stream));
        return RTMC_OK;
    }
    // results
    double *hdst[nd];
    for (int j = 0; j < 3; j++) { hdst[j] = b->loc[j]; hdst[3 + j] = b->dir[j]; hdst[6 + j] = b->efield[j]; hdst[9 + j] = b->old_point[j]; }
    hdst[12] = b->dist_left; hdst[13] = b->hit_distance; hdst[14] = b->psi;
    for (size_t j = 0; j < nd; j++) HIP_TRY(hipMemcpyAsync(hdst[j], dd + j * n, 8 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
    int32_t *idst[ni] = {wid.data(), wid0.data(), b->material, b->status, b->last_element,
                         b->was_element, b->reflected, b->iterations, b->tape_pos};
    for (size_t j = 0; j < ni; j++) HIP_TRY(hipMemcpyAsync(idst[j], icols[j], 4 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
    if (trace) HIP_TRY(hipMemcpyAsync(trace, trbuf.ptr, 8 * (size_t)n * n_steps * RTMC_TRACE_STRIDE, hipMemcpyDeviceToHost, p.stream));
    int overrun = 0;
    if (po.out) HIP_TRY(hipMemcpyAsync(po.out, oobuf.ptr, 16 * (size_t)n, hipMemcpyDeviceToHost, p.stream));
    HIP_TRY(hipMemcpyAsync(&overrun, ovbuf.ptr, 4, hipMemcpyDeviceToHost, p.stream));
    HIP_TRY(hipStreamSynchronize(p.stream));
    for (uint32_t k = 0; k < n; k++) {
        b->wavelength[k] = s->wl[wid[k]];
        b->start_wavelength[k] = s->wl[wid0[k]];
    }
    if (overrun) return fail(RTMC_EINVAL, "variate tape exhausted");
    return RTMC_OK;
}

RTMC_API int rtmc_op_intersect(rtmc_scene *s, int device, int precision, uint32_t n, const double *loc,
                                 const double *dir, double *out_points)
{
    if (!s || !loc || !dir || !out_points) return fail(RTMC_EINVAL, "null argument");
    if (precision != RTMC_F32 && precision != RTMC_F64) return fail(RTMC_EINVAL, "unknown precision %d", precision);
    if (n == 0) return RTMC_OK;
    DeviceGuard restore_device;
    PerDevice *pp;
    int rc = prepare_device(s, device, &pp);
    if (rc) return rc;
    PerDevice &p = *pp;
    std::lock_guard<std::mutex> serial(g_gpu[device].launch_mu);
    DeviceBuffer dl, ddir, dout;
    const size_t total = (size_t)n * s->n_el;
    HIP_TRY(hipMalloc(&dl.ptr, 24 * (size_t)n));
    HIP_TRY(hipMalloc(&ddir.ptr, 24 * (size_t)n));
    HIP_TRY(hipMalloc(&dout.ptr, 24 * total));
    HIP_TRY(hipMemcpyAsync(dl.ptr, loc, 24 * (size_t)n, hipMemcpyHostToDevice, p.stream));
    HIP_TRY(hipMemcpyAsync(ddir.ptr, dir, 24 * (size_t)n, hipMemcpyHostToDevice, p.stream));
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (precision == RTMC_F32)
        hipLaunchKernelGGL(intersect_kernel<float>, dim3(grid), dim3(256), 0, p.stream, (const DevScene<float> *)p.scene32,
                           n, (const double *)dl.ptr, (const double *)ddir.ptr, (double *)dout.ptr);
    else
        hipLaunchKernelGGL(intersect_kernel<double>, dim3(grid), dim3(256), 0, p.stream, (const DevScene<double> *)p.scene64,
                           n, (const double *)dl.ptr, (const double *)ddir.ptr, (double *)dout.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_points, dout.ptr, 24 * total, hipMemcpyDeviceToHost, p.stream));
    HIP_TRY(hipStreamSynchronize(p.stream));
    return RTMC_OK;
}
