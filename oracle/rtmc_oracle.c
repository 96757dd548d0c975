/*
 * rtmc_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A scalar float64 restatement of the reference's photon loop
 * (mohrezaeib/RayTracer, RayTrace.runSimulation and everything it calls), used
 * only as the checker in tests/, in __graft_entry__.smoke() and as the
 * `cpu_baseline` leg of bench.py.  Nothing in raytracer_amd/ loads it.
 *
 * It follows the reference statement by statement -- the arccos/arcsin
 * Fresnel chain, the 1e-8 self-hit rule, the line-based reflection, the
 * stale-element re-execution when nothing is hit -- and evaluates
 * three-component dot products as fma(a2,b2, fma(a1,b1, a0*b0)), which is how
 * numpy 2.2 + OpenBLAS round `np.dot` of 3-vectors on x86-64 (established
 * empirically; numpy is an unpinned third-party dependency of the reference).
 * With that, replaying a recorded variate tape of the reference reproduces its
 * photon positions and directions bit for bit (tests/test_oracle_golden.py).
 * Build with -ffp-contract=off.
 *
 * Two variate sources:
 *   tape   -- the values the reference drew, in its order;
 *   philox -- Philox4x32-10, key = seed, counter = (photon id, iteration, purpose): the
 *             same stream layout as the HIP kernels (csrc/rtmc_device.h), so a
 *             GPU run and an oracle run of the same seed follow the same
 *             photons.
 *
 * Reference file:line citations are given at each function.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rtmc.h"

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* small vector helpers with the reference's rounding                  */
/* ------------------------------------------------------------------ */
static double dot3(const double *a, const double *b)
{
    return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]));
}
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static void cross3(const double *a, const double *b, double *o)
{
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double sign1(double x) { return (x > 0) - (x < 0); } /* np.sign; NaN -> 0 here */

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t len;
    double *x; /* wavelengths */
    double *y; /* values      */
} tab_t;

typedef struct {
    rtmc_material m;
    tab_t abs_A, abs_D, em_A, em_D;
    double *cdf_A, *cdf_D; /* inverse-CDF tables of the emission draws */
    double p[3], pA[3];
    double cdf_p[3], cdf_pA[3];
    int n_p;
    double P_ET_DA, P_ET_AD;
    /* aligned acceptors: special-function values of LambertBeerFullIntegral
       (supplied by the test harness from scipy, the library the reference calls)
       and roll(MakeVectAxis(AlignVect), 2, axis=1) */
    int have_special;
    double complex erf_m1, erf_p1, erf_m2, erf_p2;
    double erfi_1, erfi_2;
    double frame[3][3];
} omat_t;

typedef struct orc_scene {
    int n_el, n_mat, n_tab;
    rtmc_element *el;
    omat_t *mat;
    tab_t *eqe;
    rtmc_source src;
    tab_t sun;
    double *sun_cdf;
    int max_iter;
} orc_scene;

static char g_err[512];
ORC_EXPORT const char *orc_last_error(void) { return g_err; }

static int tab_copy(tab_t *t, const rtmc_scene_desc *d, rtmc_table ref)
{
    t->len = ref.len;
    t->x = t->y = NULL;
    if (ref.len <= 0) return 0;
    if (ref.offset < 0 || ref.offset + 2 * ref.len > d->table_data_len) return -1;
    t->x = (double *)malloc(sizeof(double) * ref.len);
    t->y = (double *)malloc(sizeof(double) * ref.len);
    for (int64_t i = 0; i < ref.len; i++) {
        t->x[i] = d->table_data[ref.offset + 2 * i];
        t->y[i] = d->table_data[ref.offset + 2 * i + 1];
    }
    return 0;
}
static void tab_free(tab_t *t) { free(t->x); free(t->y); }

/* RTUtils.py:15-22 FindClosestListElement */
static int64_t find_closest(const double *A, int64_t n, double t)
{
    int64_t lo = 0, hi = n; /* searchsorted, side='left' */
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (A[mid] < t) lo = mid + 1; else hi = mid;
    }
    int64_t idx = lo;
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    double left = A[idx - 1], right = A[idx];
    if (t - left < right - t) idx -= 1;
    return idx;
}

/* RTUtils.py:152-158 RectifySpectrum followed by the cumulative sum that
 * numpy.random.choice builds (cdf = p.cumsum(); cdf /= cdf[-1]) */
static double *rectified_cdf(const double *x, const double *y, int64_t n, int times_lambda)
{
    double *w = (double *)malloc(sizeof(double) * n);
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double v = times_lambda ? y[i] * x[i] : y[i]; /* PhotonObject.py:44 */
        if (i > 0) v *= (x[i] - x[i - 1]);
        w[i] = v;
        s += v;
    }
    double c = 0.0;
    for (int64_t i = 0; i < n; i++) { c += w[i] / s; w[i] = c; }
    double last = w[n - 1];
    for (int64_t i = 0; i < n; i++) w[i] /= last;
    return w;
}

/* searchsorted(cdf, u, side='right'): number of entries <= u */
static int64_t cdf_pick(const double *cdf, int64_t n, double u)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo < n ? lo : n - 1;
}

/* MonteCarloFRET.py:195-237 GenerateMolecules (photon independent) */
static void generate_molecules(omat_t *M)
{
    const rtmc_material *m = &M->m;
    const double pi = 3.141592653589793;
    {
        double R0 = ((9 * (2 / 3.0) * m->Q_D * log(10.0)) / (128 * pow(m->n, 4) * 6.022e23 * pow(pi, 5))) * m->ol_DA * 1e17;
        R0 = pow(R0, 1 / 6.0);
        R0 *= 1e-7;
        double cA = m->conc_a * 6.022e17;
        double c0 = 1.0 / ((4.0 / 3) * pi * (R0 * R0 * R0));
        M->P_ET_DA = 1.0 - exp(-1.42 * cA / c0);
    }
    if (m->fix_dist) {
        double invR6 = 1 / pow(m->fix_dist_R * 1e-9, 6);
        double k = ((9 * (2 / 3.0) * m->Q_A * log(10.0)) / (128 * pow(m->n, 4) * 6.022e23 * pow(pi, 5) * m->tau_A)) * invR6 * 1e-54 * m->ol_AD * 1e17;
        M->P_ET_AD = k / (1.0 / m->tau_A + k);
    } else {
        double R0 = ((9 * (2 / 3.0) * m->Q_A * log(10.0)) / (128 * pow(m->n, 4) * 6.022e23 * pow(pi, 5))) * m->ol_AD * 1e17;
        R0 = pow(R0, 1 / 6.0);
        R0 *= 1e-7;
        double cD = m->conc_b * 6.022e17;
        if (R0 > 0) {
            double c0 = 1.0 / ((4.0 / 3) * pi * (R0 * R0 * R0));
            M->P_ET_AD = 1.0 - exp(-1.42 * cD / c0);
        } else
            M->P_ET_AD = 0.0;
    }
}

/* np.roll(RTU.MakeVectAxis(v), 2, axis=1) (RTUtils.py:25-42): columns n1, n2, v */
static void z_axis_frame(const double *v, double tm[3][3])
{
    double nv[3] = {1, 0, 0};
    if (v[0] == 1 && v[1] == 0 && v[2] == 0) { nv[0] = 0; nv[1] = 1; }
    double n1[3], n2[3];
    cross3(v, nv, n1);
    double n = norm3(n1);
    n1[0] /= n; n1[1] /= n; n1[2] /= n;
    cross3(v, n1, n2);
    n = norm3(n2);
    n2[0] /= n; n2[1] /= n; n2[2] /= n;
    for (int i = 0; i < 3; i++) { tm[i][0] = n1[i]; tm[i][1] = n2[i]; tm[i][2] = v[i]; }
}
/* np.dot(tm, v) */
static void frame_apply(double tm[3][3], double *v)
{
    double o[3];
    for (int i = 0; i < 3; i++) o[i] = dot3(tm[i], v);
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}

static void cdf3(const double *p, int n, double *cdf)
{
    double c = 0;
    for (int i = 0; i < n; i++) { c += p[i]; cdf[i] = c; }
    for (int i = 0; i < n; i++) cdf[i] /= c;
}

ORC_EXPORT void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    for (int i = 0; i < s->n_mat; i++) {
        tab_free(&s->mat[i].abs_A); tab_free(&s->mat[i].abs_D);
        tab_free(&s->mat[i].em_A); tab_free(&s->mat[i].em_D);
        free(s->mat[i].cdf_A); free(s->mat[i].cdf_D);
    }
    for (int i = 0; i < s->n_tab; i++) tab_free(&s->eqe[i]);
    tab_free(&s->sun);
    free(s->sun_cdf);
    free(s->el); free(s->mat); free(s->eqe);
    free(s);
}

ORC_EXPORT int orc_scene_create(const rtmc_scene_desc *d, orc_scene **out)
{
    if (!d || !out || d->abi_version != RTMC_ABI_VERSION) {
        snprintf(g_err, sizeof g_err, "bad description");
        return RTMC_EINVAL;
    }
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->n_el = d->n_elements; s->n_mat = d->n_materials; s->n_tab = d->n_tables;
    s->el = (rtmc_element *)malloc(sizeof(rtmc_element) * (s->n_el > 0 ? s->n_el : 1));
    memcpy(s->el, d->elements, sizeof(rtmc_element) * s->n_el);
    s->mat = (omat_t *)calloc(s->n_mat > 0 ? s->n_mat : 1, sizeof(omat_t));
    s->eqe = (tab_t *)calloc(s->n_tab > 0 ? s->n_tab : 1, sizeof(tab_t));
    s->src = d->source;
    s->max_iter = d->max_iterations;
    for (int i = 0; i < s->n_el; i++) {
        const rtmc_element *e = &s->el[i];
        if (e->geometry < RTMC_GEOM_RECTANGLE || e->geometry > RTMC_GEOM_TRIANGLE ||
            e->type < RTMC_TYPE_MIRROR || e->type > RTMC_TYPE_DUMP) {
            snprintf(g_err, sizeof g_err, "element %d: unknown geometry %d / type %d", i, e->geometry, e->type);
            orc_scene_destroy(s);
            return RTMC_EUNSUPPORTED;
        }
    }
    for (int i = 0; i < s->n_mat; i++) {
        omat_t *M = &s->mat[i];
        M->m = d->materials[i];
        if (!M->m.is_absorber) continue;
        if (M->m.is_aligned) z_axis_frame(M->m.align_vect, M->frame);
        if (tab_copy(&M->abs_A, d, M->m.abs_A) || tab_copy(&M->abs_D, d, M->m.abs_D) ||
            tab_copy(&M->em_A, d, M->m.em_A) || tab_copy(&M->em_D, d, M->m.em_D)) {
            snprintf(g_err, sizeof g_err, "material %d: table out of range", i);
            orc_scene_destroy(s);
            return RTMC_EINVAL;
        }
        M->cdf_A = rectified_cdf(M->em_A.x, M->em_A.y, M->em_A.len, 0);
        M->cdf_D = rectified_cdf(M->em_D.x, M->em_D.y, M->em_D.len, 0);
        /* MonteCarloFRET.py:64-82 */
        if (M->m.conc_a > 0) {
            generate_molecules(M);
            M->p[0] = (1 - M->P_ET_DA) * M->m.Q_D;
            M->p[1] = (1 - M->P_ET_DA) * (1 - M->m.Q_D);
            M->p[2] = M->P_ET_DA;
            M->pA[0] = (1 - M->P_ET_AD) * M->m.Q_A;
            M->pA[1] = (1 - M->P_ET_AD) * (1 - M->m.Q_A);
            M->pA[2] = M->P_ET_AD;
            M->n_p = 3;
            cdf3(M->pA, 3, M->cdf_pA);
        } else {
            M->p[0] = M->m.Q_D; M->p[1] = 1 - M->m.Q_D;
            M->n_p = 2;
        }
        cdf3(M->p, M->n_p, M->cdf_p);
    }
    for (int i = 0; i < s->n_tab; i++)
        if (tab_copy(&s->eqe[i], d, d->tables[i])) { orc_scene_destroy(s); return RTMC_EINVAL; }
    if (s->src.wavelength_mode == 1) {
        if (tab_copy(&s->sun, d, s->src.spectrum) || s->sun.len < 1) { orc_scene_destroy(s); return RTMC_EINVAL; }
        s->sun_cdf = rectified_cdf(s->sun.x, s->sun.y, s->sun.len, 1);
    }
    *out = s;
    return RTMC_OK;
}

/* erf(a - i b1), erf(a + i b1), erf(a - i b2), erf(a + i b2) as (re, im) pairs,
 * then erfi(b1), erfi(b2): the six scipy.special values PhotonObject.py:273
 * evaluates for this material's order parameter */
ORC_EXPORT int orc_set_aligned_special(orc_scene *s, int mat, const double *v)
{
    if (!s || mat < 0 || mat >= s->n_mat) return RTMC_EINVAL;
    omat_t *M = &s->mat[mat];
    M->erf_m1 = v[0] + v[1] * I; M->erf_p1 = v[2] + v[3] * I;
    M->erf_m2 = v[4] + v[5] * I; M->erf_p2 = v[6] + v[7] * I;
    M->erfi_1 = v[8]; M->erfi_2 = v[9];
    M->have_special = 1;
    return RTMC_OK;
}

/* PhotonObject.py:267-281 LambertBeerFullIntegral */
static double lambert_beer_full_integral(const omat_t *M, const double *nE)
{
    double x = nE[0], y = nE[1], z = nE[2];
    double S = M->m.align_S, ol;
    if (S > 1e-10) {
        double x2 = x * x, y2 = y * y, z2 = z * z;
        double complex t1 =
            (-0.2349964007466563 * I) * (x2 + y2 + (2 / 3.0) * z2) * M->erf_m1 +
            (0.2349964007466563 * I) * (x2 + y2 + (2 / 3.0) * z2) * M->erf_p1 +
            (0.4699928014933126 * x2 + 0.4699928014933126 * y2 + 0.313328534328875 * z2) * M->erfi_1;
        double complex t2 =
            (0.07833213358221876 * I) * (x2 + y2 - 2 * z2) * M->erf_m2 -
            (0.07833213358221876 * I) * (x2 + y2 - 2 * z2) * M->erf_p2 +
            (-0.15666426716443752 * x2 - 0.15666426716443752 * y2 + 0.31332853432887503 * z2) * M->erfi_2;
        ol = cabs(1 / S * (exp(-0.5 * S) * t1 + exp(-4.5 * (S * S)) * t2));
    } else {
        const double ez[3] = {0, 0, 1};
        double d = dot3(nE, ez);
        ol = d * d;
    }
    return ol < 1e-10 ? 0.0 : ol;
}

/* exported for tests */
ORC_EXPORT double orc_orientation_integral(const orc_scene *s, int mat, const double *nE)
{
    return lambert_beer_full_integral(&s->mat[mat], nE);
}

/* exported for tests: transfer probabilities and emission CDFs as restated */
ORC_EXPORT int orc_material_info(const orc_scene *s, int mat, double *pet2, double *cdfA, double *cdfD)
{
    if (mat < 0 || mat >= s->n_mat) return RTMC_EINVAL;
    const omat_t *M = &s->mat[mat];
    pet2[0] = M->P_ET_DA; pet2[1] = M->P_ET_AD;
    if (cdfA && M->cdf_A) memcpy(cdfA, M->cdf_A, sizeof(double) * M->em_A.len);
    if (cdfD && M->cdf_D) memcpy(cdfD, M->cdf_D, sizeof(double) * M->em_D.len);
    return RTMC_OK;
}

/* ------------------------------------------------------------------ */
/* variates                                                            */
/* ------------------------------------------------------------------ */
typedef struct {
    int tape_mode;
    const double *tape;
    uint64_t pos, end;
    int overrun;
    uint32_t key[2], pid[2];
    uint32_t w[4];
    uint32_t buf[4], block;
    int have;
} rng_t;

static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
ORC_EXPORT void orc_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }

/* philox mode: the block of four words for (loop iteration, purpose), laid out
 * as in csrc/rtmc_device.h: purpose 0 = the iteration's own block (interface,
 * detector, absorption), r = r-th back transfer of that absorption,
 * BLK_EMIT (+1) = re-emission, BLK_LAUNCH = launch */
#define BLK_STEP 0u
#define BLK_EMIT 0x80000000u
#define BLK_LAUNCH 0xC0000000u
static void rng_block(rng_t *g, uint32_t iter, uint32_t purpose)
{
    if (g->tape_mode) return;
    uint32_t ctr[4] = {g->pid[0], g->pid[1], iter, purpose};
    philox4x32_10(ctr, g->key, g->w);
}
/* tape_mode 1: the next recorded value, whatever its law.  tape_mode 2
 * ("reference order"): the draws are made where and how the reference makes
 * them (three normals for a direction, rejection sampling, ...) but their
 * values come from a private Philox stream of the photon (counter word 3 =
 * 0xFFFFFFFF keeps it apart from the event blocks); used to check at scale that
 * the samplers of the event layout follow the same laws. */
enum { DRAW_UNIFORM, DRAW_NORMAL, DRAW_EXPONENTIAL };
static double stream_uniform(rng_t *g)
{
    if (g->have == 0) {
        uint32_t ctr[4] = {g->pid[0], g->pid[1], g->block++, 0xFFFFFFFFu};
        philox4x32_10(ctr, g->key, g->buf);
        g->have = 4;
    }
    uint32_t a = g->buf[4 - g->have], b = g->buf[5 - g->have];
    g->have -= 2;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; /* 53 bits, as numpy's random_sample */
}
static double tape_next(rng_t *g, int law)
{
    if (g->tape_mode == 2) {
        if (law == DRAW_UNIFORM) return stream_uniform(g);
        if (law == DRAW_EXPONENTIAL) return -log(1.0 - stream_uniform(g));
        double r = sqrt(-2.0 * log(1.0 - stream_uniform(g)));
        return r * cos(6.283185307179586 * stream_uniform(g));
    }
    if (g->pos >= g->end) { g->overrun = 1; return 0.5; }
    return g->tape[g->pos++];
}
static double u24(uint32_t w) { return (double)(w >> 8) * (1.0 / 16777216.0); }
static double rng_uniform(rng_t *g, int slot) { return g->tape_mode ? tape_next(g, DRAW_UNIFORM) : u24(g->w[slot]); }
/* unit-scale exponential: numpy legacy = -log(1 - U) */
static double rng_exponential(rng_t *g, int slot) { return g->tape_mode ? tape_next(g, DRAW_EXPONENTIAL) : -log(1.0 - u24(g->w[slot])); }
/* three standard normals; philox: Box-Muller on (w0,w1) and (w2,w3) */
static void rng_normal3(rng_t *g, double *o)
{
    if (g->tape_mode) { for (int j = 0; j < 3; j++) o[j] = tape_next(g, DRAW_NORMAL); return; }
    const double two_pi = 6.283185307179586;
    double r0 = sqrt(-2.0 * log(((double)(g->w[0] >> 8) + 1.0) * (1.0 / 16777216.0)));
    double t0 = two_pi * u24(g->w[1]);
    double r1 = sqrt(-2.0 * log(((double)(g->w[2] >> 8) + 1.0) * (1.0 / 16777216.0)));
    double t1 = two_pi * u24(g->w[3]);
    o[0] = r0 * cos(t0); o[1] = r0 * sin(t0); o[2] = r1 * cos(t1);
}

/* ------------------------------------------------------------------ */
/* photon                                                              */
/* ------------------------------------------------------------------ */
typedef struct {
    double L[3], D[3], E[3];
    double wl, wl0, dist_left;
    int mat, status, was_element, reflected;
    double psi;                    /* Photon.Psi, set whenever a free path is drawn in an aligned medium */
    double sgn;
    double op_S[3], op_n[3];       /* Photon.OperationPlane              */
    double old_point[3];           /* Photon.OldPoint                    */
    double old_dir[3];
    double point_cy[3];
    int been_absorbed;
    /* RayTrace loop variables */
    double drv_old_point[3];
    int first_el, old_el;
    double smallest;
    int iter;
    int stale;        /* last iteration found no element */
    int absorbed_now; /* last iteration absorbed the photon */
} ph_t;

/* PhotonObject.py:107-114 UpdatePolarizationRandom */
static void polarization_random(ph_t *p, rng_t *g, const double *dir)
{
    double v[3];
    rng_normal3(g, v);
    double n = norm3(v);
    v[0] /= n; v[1] /= n; v[2] /= n;
    cross3(dir, v, p->E);
    n = norm3(p->E);
    p->E[0] /= n; p->E[1] /= n; p->E[2] /= n;
}

/* RTUtils.py:102-108 GenerateDirectionSine */
static void direction_isotropic(rng_t *g, double *d)
{
    rng_normal3(g, d);
    double n = norm3(d);
    d[0] /= n; d[1] /= n; d[2] /= n;
}

/* Philox mode only: exact samplers of the same laws from uniforms, written
 * like the HIP kernels write them (csrc/rtmc_device.h sphere_direction,
 * perpendicular) so that both follow the same photons. */
static void sphere_direction(double u_z, double u_phi, double *d)
{
    const double two_pi = 6.283185307179586;
    double z = 1.0 - 2.0 * u_z;
    double r2 = 1.0 - z * z;
    double r = sqrt(r2 > 0 ? r2 : 0.0);
    d[0] = r * cos(two_pi * u_phi); d[1] = r * sin(two_pi * u_phi); d[2] = z;
}
static void perpendicular(const double *D, double u_psi, double *E)
{
    const double two_pi = 6.283185307179586;
    double sg = copysign(1.0, D[2]);
    double a = -1.0 / (sg + D[2]);
    double b = D[0] * D[1] * a;
    double s = sin(two_pi * u_psi), c = cos(two_pi * u_psi);
    E[0] = c * (1.0 + sg * D[0] * D[0] * a) + s * b;
    E[1] = c * (sg * b) + s * (sg + D[1] * D[1] * a);
    E[2] = c * (-sg * D[0]) + s * (-D[1]);
}

/* PhotonObject.py:283-305 LamBeerDistance */
static void lambeer_distance(const orc_scene *s, ph_t *p, rng_t *g, int slot)
{
    omat_t *M = &s->mat[p->mat];
    if (M->m.is_absorber) {
        double ea = M->abs_A.y[find_closest(M->abs_A.x, M->abs_A.len, p->wl)];
        double ed = M->abs_D.y[find_closest(M->abs_D.x, M->abs_D.len, p->wl)];
        double b_a = M->m.conc_a * 1e-3 * M->m.emax_A * log(10.0) * fmax(1e-99, ea);
        double b_d = M->m.conc_b * 1e-3 * M->m.emax_D * log(10.0) * fmax(1e-99, ed);
        if (M->m.is_aligned) { /* :291-302 */
            double nE[3];
            for (int j = 0; j < 3; j++) /* np.dot(E, tm): row vector times matrix */
                nE[j] = fma(p->E[2], M->frame[2][j], fma(p->E[1], M->frame[1][j], p->E[0] * M->frame[0][j]));
            double n = norm3(nE);
            nE[0] /= n; nE[1] /= n; nE[2] /= n;
            p->psi = lambert_beer_full_integral(M, nE);
            b_a *= p->psi * 3.0;
        }
        double scale = 1.0 / (b_a + b_d);
        p->dist_left = (1 / s->src.scale_factor) * (scale * rng_exponential(g, slot));
    } else
        p->dist_left = 0.0;
}

/* PhotonObject.py:49-97 ResetPhoton / GeneratePhoton */
static void launch_photon(const orc_scene *s, ph_t *p, rng_t *g)
{
    const rtmc_source *src = &s->src;
    p->was_element = 0;
    memcpy(p->L, src->location, sizeof p->L);
    rng_block(g, 0, BLK_LAUNCH); /* philox: one block for the whole launch: words 0,1 direction, 2 polarisation, 3 wavelength */
    if (src->direction_mode == 1) {
        double d[3];
        if (g->tape_mode) direction_isotropic(g, d);
        else sphere_direction(u24(g->w[0]), u24(g->w[1]), d);
        d[2] = fabs(d[2]);
        for (int j = 0; j < 3; j++) /* row vector times matrix */
            p->D[j] = fma(d[2], src->frame[6 + j], fma(d[1], src->frame[3 + j], d[0] * src->frame[j]));
    } else
        memcpy(p->D, src->direction, sizeof p->D);
    const int has_area = src->area[0] != 0.0 || src->area[1] != 0.0;
    if (has_area && g->tape_mode) { /* PhotonObject.py:76,84 (commented out there): rand() for x, then for y */
        p->L[0] += (tape_next(g, DRAW_UNIFORM) - 0.5) * src->area[0];
        p->L[1] += (tape_next(g, DRAW_UNIFORM) - 0.5) * src->area[1];
    }
    double n = norm3(p->D);
    p->D[0] /= n; p->D[1] /= n; p->D[2] /= n;
    if (src->wavelength_mode == 1)
        p->wl = s->sun.x[cdf_pick(s->sun_cdf, s->sun.len, rng_uniform(g, 3))];
    else
        p->wl = src->wavelength;
    p->mat = src->start_material;
    if (g->tape_mode) polarization_random(p, g, p->D);
    else perpendicular(p->D, u24(g->w[2]), p->E);
    if (has_area && !g->tape_mode) { /* philox: a block of its own, words 0, 1 */
        rng_block(g, 0, BLK_LAUNCH + 1u);
        p->L[0] += (u24(g->w[0]) - 0.5) * src->area[0];
        p->L[1] += (u24(g->w[1]) - 0.5) * src->area[1];
    }
    p->wl0 = p->wl;
    p->reflected = 0;
    if (s->n_el > 0) {
        memcpy(p->op_S, s->el[0].support, sizeof p->op_S);
        memcpy(p->op_n, s->el[0].normal, sizeof p->op_n);
    }
    /* state the reference carries over from the previous photon and that a
       batched engine cannot: see the header of this file */
    p->dist_left = 0.0;
    p->status = RTMC_ST_START;
    p->first_el = p->old_el = -1;
    p->been_absorbed = 0;
    p->iter = 0;
    p->sgn = 1;
    memcpy(p->drv_old_point, p->L, sizeof p->L);
    memcpy(p->old_point, p->L, sizeof p->L);
    memset(p->point_cy, 0, sizeof p->point_cy);
    p->smallest = INFINITY;
}

/* Geometries.py:32-52 Rectangle.FindIntersections; returns 0 for [nan]*3 */
static int rect_intersect(const rtmc_element *e, const double *L, const double *D, double *B)
{
    double dn = dot3(D, e->normal);
    if (dn != 0) {
        double sl[3] = {e->support[0] - L[0], e->support[1] - L[1], e->support[2] - L[2]};
        double t = dot3(sl, e->normal) / dn;
        double d[3];
        for (int j = 0; j < 3; j++) { B[j] = t * D[j] + L[j]; d[j] = B[j] - e->support[j]; }
        double t1 = dot3(e->e1, d), t2 = dot3(e->e2, d);
        if (e->half1 < fabs(t1) || e->half2 < fabs(t2)) return 0;
        /* np.all(point != [nan]*3) is also False for a point that contains no NaN... it is
           True for every array; a NaN coordinate compares unequal as well, so any array passes */
        return 1;
    }
    return 0;
}

/* line-plane intersection shared by the flat shapes:
 * B = ((S - L).n / (D.n)) D + L, Geometries.py:38,72,223 */
static int plane_point(const rtmc_element *e, const double *L, const double *D, double *B)
{
    double dn = dot3(D, e->normal);
    if (dn == 0) return 0;
    double sl[3] = {e->support[0] - L[0], e->support[1] - L[1], e->support[2] - L[2]};
    double t = dot3(sl, e->normal) / dn;
    for (int j = 0; j < 3; j++) B[j] = t * D[j] + L[j];
    return 1;
}

/* Geometries.py:68-80 Circle.FindIntersections */
static int circle_intersect(const rtmc_element *e, const double *L, const double *D, double *B)
{
    if (!plane_point(e, L, D, B)) return 0;
    double d[3] = {B[0] - e->support[0], B[1] - e->support[1], B[2] - e->support[2]};
    return !(e->half1 < norm3(d));
}

/* RTUtils.py:25-42 MakeVectAxis: columns v, n1, n2 */
static void vect_axis(const double *v, double *n1, double *n2)
{
    double nv[3] = {1, 0, 0};
    if (v[0] == 1 && v[1] == 0 && v[2] == 0) { nv[0] = 0; nv[1] = 1; }
    cross3(v, nv, n1);
    double n = norm3(n1);
    n1[0] /= n; n1[1] /= n; n1[2] /= n;
    cross3(v, n1, n2);
    n = norm3(n2);
    n2[0] /= n; n2[1] /= n; n2[2] /= n;
}

/* Geometries.py:220-251 Triangle.FindIntersections / CheckPointInTriangle:
 * e1, e2, vertex3 are the three corners */
static int triangle_intersect(const rtmc_element *e, const double *L, const double *D, double *B)
{
    if (!plane_point(e, L, D, B)) return 0;
    double n1[3], n2[3];
    vect_axis(e->normal, n1, n2); /* transpose(tm) rows: normal, n1, n2; only rows 1, 2 are used */
    const double *c[3] = {e->e1, e->e2, e->vertex3};
    double q[3][2], pt[2] = {dot3(n1, B), dot3(n2, B)};
    for (int i = 0; i < 3; i++) { q[i][0] = dot3(n1, c[i]); q[i][1] = dot3(n2, c[i]); }
#define SCL(p1, p2, p3) (((p1)[0] - (p3)[0]) * ((p2)[1] - (p3)[1]) - ((p2)[0] - (p3)[0]) * ((p1)[1] - (p3)[1]))
    int b1 = SCL(pt, q[0], q[1]) < 0.0;
    int b2 = SCL(pt, q[1], q[2]) < 0.0;
    int b3 = SCL(pt, q[2], q[0]) < 0.0;
#undef SCL
    return b1 == b2 && b2 == b3;
}

/* Geometries.py:156-204 IntersectLineCylinder: support/normal hold the two axis
 * end points.  ok1/ok2 tell which of the two line points exist. */
static void cylinder_line(const rtmc_element *e, const double *L, const double *D, double *p1, int *ok1,
                          double *p2, int *ok2)
{
    *ok1 = *ok2 = 0;
    double r = e->half1;
    double a[3] = {e->support[0] - e->normal[0], e->support[1] - e->normal[1], e->support[2] - e->normal[2]};
    double na = norm3(a);
    a[0] /= na; a[1] /= na; a[2] /= na;
    double n1[3], n2[3];
    vect_axis(a, n1, n2); /* tm = transpose(MakeVectAxis(a)): rows a, n1, n2; its inverse has them as columns */
    double ts[3] = {dot3(a, L), dot3(n1, L), dot3(n2, L)};
    double td[3] = {dot3(a, D), dot3(n1, D), dot3(n2, D)};
    double c1[3] = {dot3(a, e->support), dot3(n1, e->support), dot3(n2, e->support)};
    double c2x = dot3(a, e->normal);
    double a1 = td[1] * td[1] + td[2] * td[2];
    double a2 = 2 * (td[1] * (ts[1] - c1[1]) + td[2] * (ts[2] - c1[2]));
    double a3 = (ts[1] - c1[1]) * (ts[1] - c1[1]) + (ts[2] - c1[2]) * (ts[2] - c1[2]) - r * r;
    double dsc = a2 * a2 - 4 * a1 * a3;
    if (a1 == 0.0 || !(dsc >= 0)) return;
    double t1 = (-a2 + sqrt(dsc)) / (2 * a1), t2 = (-a2 - sqrt(dsc)) / (2 * a1);
    double lo = fmin(c1[0], c2x), hi = fmax(c1[0], c2x);
    double v1[3], v2[3];
    for (int j = 0; j < 3; j++) { v1[j] = ts[j] + t1 * td[j]; v2[j] = ts[j] + t2 * td[j]; }
    for (int j = 0; j < 3; j++) {
        p1[j] = fma(n2[j], v1[2], fma(n1[j], v1[1], a[j] * v1[0]));
        p2[j] = fma(n2[j], v2[2], fma(n1[j], v2[1], a[j] * v2[0]));
    }
    *ok1 = !(v1[0] < lo || v1[0] > hi);
    *ok2 = !(v2[0] < lo || v2[0] > hi);
}

/* Geometries.py:97-136 Cylinder.FindIntersections.  Returns 0 for "no point";
 * the one combination for which the reference leaves its result unassigned
 * (both points within 1e-7 of the photon) also gives 0. */
static int cylinder_intersect(const rtmc_element *e, const double *L, const double *D, int was_element, double *B)
{
    double p1[3], p2[3];
    int ok1, ok2;
    cylinder_line(e, L, D, p1, &ok1, p2, &ok2);
    if (!was_element) { /* :102-107 */
        int k = D[0] != 0 ? 0 : (D[1] != 0 ? 1 : 2);
        if (ok1 && (p1[k] - L[k]) / D[k] < 0) ok1 = 0;
        if (ok2 && (p2[k] - L[k]) / D[k] < 0) ok2 = 0;
    }
    const double *pick = NULL;
    if (ok1 && ok2) {
        double d1v[3] = {p1[0] - L[0], p1[1] - L[1], p1[2] - L[2]};
        double d2v[3] = {p2[0] - L[0], p2[1] - L[1], p2[2] - L[2]};
        double d1 = norm3(d1v), d2 = norm3(d2v);
        if (d1 > 1e-7 && d2 > 1e-7) pick = d1 < d2 ? p1 : p2;
        if (d1 > 1e-7 && d2 <= 1e-7) pick = p1;
        if (d1 <= 1e-7 && d2 > 1e-7) pick = p2;
    } else if (ok1)
        pick = p1;
    else if (ok2)
        pick = p2;
    if (!pick) return 0;
    memcpy(B, pick, 3 * sizeof(double));
    return 1;
}

/* Geometries.py:138-153,207-209 Cylinder.OperationPlane at point P */
static void cylinder_plane(const rtmc_element *e, const double *P, double *S, double *n)
{
    double d[3] = {e->normal[0] - e->support[0], e->normal[1] - e->support[1], e->normal[2] - e->support[2]};
    double nd = norm3(d);
    d[0] /= nd; d[1] /= nd; d[2] /= nd;
    double pa[3] = {P[0] - e->support[0], P[1] - e->support[1], P[2] - e->support[2]};
    double k = dot3(pa, d);
    for (int j = 0; j < 3; j++) n[j] = (e->support[j] + k * d[j]) - P[j];
    double nn = norm3(n);
    for (int j = 0; j < 3; j++) { n[j] /= nn; S[j] = P[j]; }
}

/* Geometry.OperationPlane() of element `el` for a photon whose candidate point is P */
static void operation_plane(const orc_scene *s, int el, const double *P, double *S, double *n)
{
    const rtmc_element *e = &s->el[el];
    if (e->geometry == RTMC_GEOM_CYLINDER) { cylinder_plane(e, P, S, n); return; }
    memcpy(S, e->support, 3 * sizeof(double));
    memcpy(n, e->normal, 3 * sizeof(double));
}

static int element_intersect(const rtmc_element *e, const double *L, const double *D, int was_element, double *B)
{
    switch (e->geometry) {
    case RTMC_GEOM_RECTANGLE: return rect_intersect(e, L, D, B);
    case RTMC_GEOM_CIRCLE: return circle_intersect(e, L, D, B);
    case RTMC_GEOM_TRIANGLE: return triangle_intersect(e, L, D, B);
    case RTMC_GEOM_CYLINDER: return cylinder_intersect(e, L, D, was_element, B);
    }
    return 0;
}

/* RTUtils.py:6-11 CheckSides */
static int check_sides(const double *v1, const double *v2, const double *S, const double *n)
{
    double off = (S[0] * n[0] + S[1] * n[1]) + S[2] * n[2];
    double a = ((n[0] * v1[0] + n[1] * v1[1]) + n[2] * v1[2]) - off;
    double b = ((n[0] * v2[0] + n[1] * v2[1]) + n[2] * v2[2]) - off;
    return sign1(a) == sign1(b);
}

/* Elements.py:26-28,113-114,131-132 IsValidNextPoint of the previous element */
static int valid_next(const orc_scene *s, const ph_t *p, const double *pt)
{
    int same = check_sides(p->drv_old_point, pt, p->op_S, p->op_n);
    int type = s->el[p->old_el].type;
    if (type == RTMC_TYPE_MIRROR) return same == 1;
    if (type == RTMC_TYPE_INTERFACE) return same == p->reflected;
    if (type == RTMC_TYPE_LENS) return same == 0;
    return 0;
}

/* RayTrace.py:120-154 CheckElementsForCrossingPoint */
static void find_crossing(const orc_scene *s, ph_t *p)
{
    p->smallest = INFINITY;
    p->point_cy[0] = p->point_cy[1] = p->point_cy[2] = 0;
    for (int i = 0; i < s->n_el; i++) {
        double pt[3];
        if (!element_intersect(&s->el[i], p->L, p->D, p->was_element, pt)) continue;
        double df[3] = {p->L[0] - pt[0], p->L[1] - pt[1], p->L[2] - pt[2]};
        double dist = norm3(df);
        if (dist < p->smallest && dist > 1e-8) {
            int take;
            if (p->was_element)
                take = p->old_el >= 0 ? valid_next(s, p, pt) : 0;
            else {
                int k = p->D[0] != 0 ? 0 : (p->D[1] != 0 ? 1 : 2);
                take = (pt[k] - p->L[k]) / p->D[k] > 0;
            }
            if (take) {
                p->smallest = dist;
                p->first_el = i;
                memcpy(p->point_cy, pt, sizeof pt);
            }
        }
    }
    p->old_el = p->first_el;
}

/* MonteCarloFRET.py:258-279 EmitWavelength */
static double emit_wavelength(const omat_t *M, int donor, double u)
{
    if (donor) return M->em_D.x[cdf_pick(M->cdf_D, M->em_D.len, u)];
    return M->em_A.x[cdf_pick(M->cdf_A, M->em_A.len, u)];
}

/* philox mode: the re-emission event, words 0,1 direction, 2 polarisation, 3 free path */
static void emit_philox(ph_t *p, rng_t *g)
{
    rng_block(g, (uint32_t)p->iter, BLK_EMIT);
    sphere_direction(u24(g->w[0]), u24(g->w[1]), p->D);
    perpendicular(p->D, u24(g->w[2]), p->E);
}

/* MonteCarloFRET.py:124-162: fluorescence of an aligned acceptor.
 * Tape mode draws in the reference's order (RTUtils.py:56-100 rejection
 * sampling of sin^3, RTUtils.py:45-54 Gaussian tilt, optional azimuth of the
 * emission dipole, optional random polarisation, wavelength).  Philox mode
 * samples the same laws from two events:
 *   event A: w0 polar angle by the closed-form inverse of the sin^3 CDF,
 *            w1 its azimuth, (w2, w3) Box-Muller normal of the tilt;
 *   event B: w0 tilt azimuth, w1 dipole azimuth, w2 polarisation angle,
 *            w3 free path (drawn by the caller). */
static double aligned_acceptor_emission(const orc_scene *s, ph_t *p, rng_t *g)
{
    omat_t *M = &s->mat[p->mat];
    const double pi = 3.141592653589793;
    double sx, cx, psi_a, theta, phi, ph = 0, u_pol = 0, u_lambda = 0;
    if (g->tape_mode) {
        double ux = tape_next(g, DRAW_UNIFORM) * pi, uy = tape_next(g, DRAW_UNIFORM);
        while (uy > pow(sin(ux), 3) && !g->overrun) { ux = tape_next(g, DRAW_UNIFORM) * pi; uy = tape_next(g, DRAW_UNIFORM); }
        psi_a = tape_next(g, DRAW_UNIFORM) * 2 * pi;
        sx = sin(ux); cx = cos(ux);
        theta = M->m.align_S * tape_next(g, DRAW_NORMAL);
        phi = tape_next(g, DRAW_UNIFORM) * 2.0 * pi;
    } else {
        rng_block(g, (uint32_t)p->iter, BLK_EMIT);
        /* F(c) = (2 - 3c + c^3)/4 for c = cos(x); the root of c^3 - 3c + 2 - 4u in [-1, 1] */
        cx = 2.0 * cos((acos(2.0 * u24(g->w[0]) - 1.0) + 4.0 * pi) / 3.0);
        double s2 = 1.0 - cx * cx;
        sx = sqrt(s2 > 0 ? s2 : 0.0);
        psi_a = 2 * pi * u24(g->w[1]);
        double r = sqrt(-2.0 * log(((double)(g->w[2] >> 8) + 1.0) * (1.0 / 16777216.0)));
        theta = M->m.align_S * (r * cos(2 * pi * u24(g->w[3])));
        rng_block(g, (uint32_t)p->iter, BLK_EMIT + 1u);
        phi = 2 * pi * u24(g->w[0]);
        ph = 2 * pi * u24(g->w[1]);
        u_pol = u24(g->w[2]);
        u_lambda = 0; /* the acceptor wavelength comes from the absorption event, see fret_event */
    }
    double D[3] = {sx * cos(psi_a), sx * sin(psi_a), cx};
    double vm[3] = {sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta)};
    frame_apply(M->frame, vm);
    double n = norm3(vm);
    vm[0] /= n; vm[1] /= n; vm[2] /= n;
    double tm[3][3];
    z_axis_frame(vm, tm);
    if (M->m.abs_em_angle > 0) { /* :139-149 */
        double th = (M->m.abs_em_angle / 180.0) * pi;
        if (g->tape_mode) ph = tape_next(g, DRAW_UNIFORM) * 2 * pi;
        double v[3] = {sin(th) * cos(ph), sin(th) * sin(ph), cos(th)};
        double etm[3][3];
        z_axis_frame(v, etm);
        frame_apply(etm, D);
    }
    frame_apply(tm, D);
    n = norm3(D);
    p->D[0] = D[0] / n; p->D[1] = D[1] / n; p->D[2] = D[2] / n;
    if (s->src.respect_polarization) { /* PhotonObject.py:100-105 */
        double v1[3], v2[3];
        cross3(vm, p->D, v1);
        cross3(p->D, v1, v2);
        n = norm3(v2);
        p->E[0] = v2[0] / n; p->E[1] = v2[1] / n; p->E[2] = v2[2] / n;
    } else if (g->tape_mode)
        polarization_random(p, g, p->D);
    else
        perpendicular(p->D, u_pol, p->E);
    if (g->tape_mode) u_lambda = tape_next(g, DRAW_UNIFORM);
    return u_lambda;
}

/* MonteCarloFRET.py:63-193 RunSimulation.  `have_block`:
 * the caller has already opened the draw event (its word 0 was the
 * donor/acceptor coin). */
static void fret_event(const orc_scene *s, ph_t *p, rng_t *g, int in_donor, int have_block)
{
    omat_t *M = &s->mat[p->mat];
    uint32_t round = 0;
    for (;;) {
        double u_lambda = 0;
        if (!have_block) rng_block(g, (uint32_t)p->iter, BLK_STEP + (++round));
        if (!g->tape_mode) u_lambda = u24(g->w[3]); /* saved before later events replace the block */
        have_block = 0;
        /* MonteCarloFRET.py:90 -- drawn even when the acceptor was excited */
        int pathway = (int)cdf_pick(M->cdf_p, M->n_p, rng_uniform(g, 1));
        if (pathway == 0 && in_donor) {
            /* MonteCarloFRET.py:92-105: direction, wavelength, polarisation */
            if (g->tape_mode) {
                direction_isotropic(g, p->D);
                u_lambda = tape_next(g, DRAW_UNIFORM);
                polarization_random(p, g, p->D);
            } else
                emit_philox(p, g);
            p->wl = emit_wavelength(M, 1, u_lambda);
            p->status = RTMC_ST_REEMITTED_DONOR;
            return;
        }
        if (pathway == 1 && in_donor) { /* :106-109 */
            p->status = RTMC_ST_INTERNAL_CONVERTED;
            return;
        }
        /* :110-119 acceptor excited (directly or by transfer) */
        int pa = (int)cdf_pick(M->cdf_pA, 3, rng_uniform(g, 2));
        if (pa == 0 && M->m.is_aligned) { /* :124-162 */
            double ut = aligned_acceptor_emission(s, p, g);
            if (g->tape_mode) u_lambda = ut;
            p->wl = emit_wavelength(M, 0, u_lambda);
            p->status = RTMC_ST_REEMITTED_ACCEPTOR;
            return;
        }
        if (pa == 0) {
            /* :170-175 direction, polarisation, wavelength */
            if (g->tape_mode) {
                direction_isotropic(g, p->D);
                polarization_random(p, g, p->D);
                u_lambda = tape_next(g, DRAW_UNIFORM);
            } else
                emit_philox(p, g);
            p->wl = emit_wavelength(M, 0, u_lambda);
            p->status = RTMC_ST_REEMITTED_ACCEPTOR;
            return;
        }
        if (pa == 1) { /* :182-185 */
            p->status = RTMC_ST_INTERNAL_CONVERTED;
            return;
        }
        in_donor = 1; /* :187-190 back transfer, start over in the donor */
    }
}

/* PhotonObject.py:206-264 CheckForAbsorption */
static void check_absorption(const orc_scene *s, ph_t *p, rng_t *g, const double *old_point, double smallest)
{
    const omat_t *M = &s->mat[p->mat];
    memcpy(p->old_point, old_point, sizeof p->old_point);
    if (M->m.is_absorber) {
        if (p->dist_left > smallest) {
            p->dist_left -= smallest;
            p->status = RTMC_ST_START;
        } else {
            p->been_absorbed = 1;
            p->was_element = 0;
            memcpy(p->old_point, p->L, sizeof p->L);
            p->status = RTMC_ST_ABSORBED;
            double r[3] = {p->point_cy[0] - p->L[0], p->point_cy[1] - p->L[1], p->point_cy[2] - p->L[2]};
            double n = norm3(r);
            for (int j = 0; j < 3; j++) { r[j] /= n; p->L[j] = r[j] * p->dist_left + p->L[j]; }
            double ed = M->abs_D.y[find_closest(M->abs_D.x, M->abs_D.len, p->wl)];
            double ea = M->abs_A.y[find_closest(M->abs_A.x, M->abs_A.len, p->wl)];
            double b_d = M->m.conc_b * 1e-3 * log(10.0) * M->m.emax_D * fmax(1e-99, ed);
            double b_a = M->m.conc_a * 1e-3 * log(10.0) * M->m.emax_A * fmax(1e-99, ea);
            if (M->m.is_aligned) b_a *= p->psi * 3; /* :238-241 */
            int in_donor = rng_uniform(g, 0) < b_d / (b_d + b_a);
            fret_event(s, p, g, in_donor, 1);
            p->absorbed_now = 1;
            /* PhotonObject.py:250/258: a new free path is drawn whatever happened */
            if (g->tape_mode || p->status != RTMC_ST_INTERNAL_CONVERTED)
                lambeer_distance(s, p, g, 3);
        }
    } else {
        p->status = RTMC_ST_START;
        p->was_element = 1;
    }
}

/* PhotonObject.py:135-139 PrepareExecution */
static void prepare_execution(ph_t *p)
{
    p->sgn = sign1(-dot3(p->D, p->op_n));
    memcpy(p->old_point, p->L, sizeof p->L);
    memcpy(p->old_dir, p->D, sizeof p->D);
}

/* PhotonObject.py:145-154 ReflectOnPlane */
static void reflect_on_plane(ph_t *p)
{
    double sl[3] = {p->op_S[0] - p->L[0], p->op_S[1] - p->L[1], p->op_S[2] - p->L[2]};
    double t = dot3(sl, p->op_n) / dot3(p->D, p->op_n);
    for (int j = 0; j < 3; j++) p->L[j] = t * p->D[j] + p->L[j];
    double c = 2 * dot3(p->op_n, p->D);
    for (int j = 0; j < 3; j++) p->D[j] = c * p->op_n[j] - p->D[j];
}

/* PhotonObject.py:195-203 RefractOnPlane */
static void refract_on_plane(ph_t *p, double n1, double n2)
{
    double sl[3] = {p->op_S[0] - p->L[0], p->op_S[1] - p->L[1], p->op_S[2] - p->L[2]};
    double t = dot3(sl, p->op_n) / dot3(p->D, p->op_n);
    for (int j = 0; j < 3; j++) p->L[j] = t * p->D[j] + p->L[j];
    double r = n1 / n2;
    double c = -dot3(p->D, p->op_n);
    double k = r * c - p->sgn * sqrt(1 - (r * r) * (1 - c * c));
    for (int j = 0; j < 3; j++) p->D[j] = r * p->D[j] + k * p->op_n[j];
}

/* RTUtils.py:110-123 RotationMatrix applied to a vector */
static void rotate(const double *axis_in, double theta, double *v)
{
    double nn = sqrt(dot3(axis_in, axis_in));
    double ax[3] = {axis_in[0] / nn, axis_in[1] / nn, axis_in[2] / nn};
    double a = cos(theta / 2.0);
    double sn = sin(theta / 2.0);
    double b = -ax[0] * sn, c = -ax[1] * sn, d = -ax[2] * sn;
    double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    double bc = b * c, ad = a * d, ac = a * c, ab = a * b, bd = b * d, cd = c * d;
    double m[3][3] = {{aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac)},
                      {2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab)},
                      {2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc}};
    double o[3];
    for (int i = 0; i < 3; i++) o[i] = dot3(m[i], v);
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
}

/* PhotonObject.py:117-129,141-143 PostProcessExecution / UpdatePolarization */
static void post_process(ph_t *p)
{
    double cp[3];
    cross3(p->D, p->old_dir, cp);
    if (norm3(cp) != 0) {
        double acs = dot3(p->D, p->old_dir) / (norm3(p->D) * norm3(p->old_dir));
        if (acs <= 1.0 && acs >= -1.0) {
            rotate(cp, -acos(acs), p->E);
            double n = norm3(p->E);
            p->E[0] /= n; p->E[1] /= n; p->E[2] /= n;
        }
    }
    p->was_element = 1;
}

/* Elements.py:75-81 IsTotalReflect */
static int is_total_reflect(const ph_t *p, double n1, double n2)
{
    double mn[3] = {-p->sgn * p->op_n[0], -p->sgn * p->op_n[1], -p->sgn * p->op_n[2]};
    double thetai = acos(dot3(p->D, mn) / (norm3(p->D) * norm3(p->op_n)));
    return n1 > n2 && fabs(thetai) > asin(n2 / n1);
}

/* Elements.py:84-109 Reflectance */
static double reflectance(ph_t *p, double n1, double n2)
{
    double sn[3] = {p->sgn * p->op_n[0], p->sgn * p->op_n[1], p->sgn * p->op_n[2]};
    double thetai = acos(-dot3(p->D, sn));
    double thetat = asin((n1 / n2) * sin(thetai));
    double ci = cos(thetai), ct = cos(thetat);
    double rp = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct);
    double rl = (n2 * ci - n1 * ct) / (n2 * ci + n1 * ct);
    double R_perp = rp * rp, R_par = rl * rl;
    double en = norm3(p->E);
    p->E[0] /= en; p->E[1] /= en; p->E[2] /= en; /* the reference normalises E in place */
    double nee[3];
    cross3(p->D, sn, nee);
    double nrm = norm3(nee);
    if (nrm != 0) {
        nee[0] /= nrm; nee[1] /= nrm; nee[2] /= nrm;
        double dp = dot3(p->E, nee);
        double par[3] = {p->E[0] - dp * nee[0], p->E[1] - dp * nee[1], p->E[2] - dp * nee[2]};
        double parproj = norm3(par);
        double sproj = fabs(dp);
        return (parproj * parproj) * R_par + (sproj * sproj) * R_perp;
    }
    double q = (n1 - n2) / (n1 + n2);
    return q * q;
}

/* Elements.py:41-73 Interface.ExecuteOperation */
static void interface_op(const orc_scene *s, ph_t *p, rng_t *g, int el, rtmc_tallies *t)
{
    const rtmc_element *e = &s->el[el];
    prepare_execution(p);
    int other = (s->mat[p->mat].m.name_id == s->mat[e->material1].m.name_id) ? e->material2 : e->material1;
    double n1 = s->mat[p->mat].m.n, n2 = s->mat[other].m.n;
    if (is_total_reflect(p, n1, n2)) {
        reflect_on_plane(p);
        p->reflected = 1;
        if (t) t->bounce[el]++;
    } else {
        double R = reflectance(p, n1, n2);
        if (rng_uniform(g, 0) > R) {
            refract_on_plane(p, n1, n2);
            p->mat = other;
            p->reflected = 0;
            lambeer_distance(s, p, g, 1);
        } else {
            reflect_on_plane(p);
            p->reflected = 1;
            if (t) t->bounce[el]++;
        }
    }
    post_process(p);
}

/* Elements.py:18-23 Mirror.ExecuteOperation */
static void mirror_op(ph_t *p, int el, rtmc_tallies *t)
{
    prepare_execution(p);
    reflect_on_plane(p);
    if (t) t->bounce[el]++;
    post_process(p);
}

/* PhotonObject.py:158-191 FocusThinLens */
static void focus_thin_lens(ph_t *p, double f)
{
    double sl[3] = {p->op_S[0] - p->L[0], p->op_S[1] - p->L[1], p->op_S[2] - p->L[2]};
    double t = dot3(sl, p->op_n) / dot3(p->D, p->op_n);
    double base[3], fp1[3], fp2[3];
    for (int j = 0; j < 3; j++) {
        base[j] = t * p->D[j] + p->L[j];
        fp1[j] = p->op_S[j] + f * p->op_n[j];
        fp2[j] = p->op_S[j] - f * p->op_n[j];
    }
    /* the focal plane on the far side of the lens */
    const double *fp = check_sides(p->L, fp1, p->op_S, p->op_n) ? fp2 : fp1;
    /* "thin cylindrical lens": the parallel ray starts at the hit point's x, y and the lens centre's z */
    double pr[3] = {base[0], base[1], p->op_S[2]};
    double d2[3] = {fp[0] - pr[0], fp[1] - pr[1], fp[2] - pr[2]};
    double t2 = dot3(d2, p->op_n) / dot3(p->D, p->op_n);
    double dv[3];
    for (int j = 0; j < 3; j++) dv[j] = -((t2 * p->D[j] + pr[j]) - base[j]);
    double n = norm3(dv);
    for (int j = 0; j < 3; j++) { p->L[j] = base[j]; p->D[j] = dv[j] / n; }
}

/* Elements.py:125-129 Lens.ExecuteOperation */
static void lens_op(const orc_scene *s, ph_t *p, int el)
{
    prepare_execution(p);
    focus_thin_lens(p, s->el[el].focal_length);
    post_process(p);
}

/* Elements.py:152-169 Dump.ExecuteOperation */
static void dump_op(const orc_scene *s, ph_t *p, rng_t *g, int el, rtmc_tallies *t)
{
    const rtmc_element *e = &s->el[el];
    prepare_execution(p);
    p->status = RTMC_ST_DUMPED;
    int counted = 1;
    if (e->eqe_table >= 0) {
        const tab_t *q = &s->eqe[e->eqe_table];
        counted = q->y[find_closest(q->x, q->len, p->wl)] > rng_uniform(g, 0);
    }
    if (t) {
        t->counter[el]++;
        double ratio = 1 * p->wl0 / p->wl;
        t->energy_fx[el] += (uint64_t)llrint(ratio * RTMC_ENERGY_FX_SCALE);
        /* IncidentSpectrum bins are the integers 200..3999 */
        double lo = RTMC_SPECTRUM_FIRST_NM;
        int64_t idx;
        { /* FindClosestListElement on arange(200, 4000) */
            double tw = p->wl;
            int64_t ss = (int64_t)ceil(tw - lo); /* searchsorted left: first k with lo+k >= t */
            if (ss < 0) ss = 0;
            if (ss > RTMC_SPECTRUM_BINS) ss = RTMC_SPECTRUM_BINS;
            idx = ss;
            if (idx < 1) idx = 1;
            if (idx > RTMC_SPECTRUM_BINS - 1) idx = RTMC_SPECTRUM_BINS - 1;
            double left = lo + (double)(idx - 1), right = lo + (double)idx;
            if (tw - left < right - tw) idx -= 1;
        }
        t->spectrum[(int64_t)el * RTMC_SPECTRUM_BINS + idx]++;
        if (counted) t->spectral[el]++;
    }
    memcpy(p->L, p->point_cy, sizeof p->L);
    post_process(p);
}

/* one iteration of the loop body, RayTrace.py:178-195; returns 0 when the photon stops */
static int loop_iteration(const orc_scene *s, ph_t *p, rng_t *g, rtmc_tallies *t)
{
    rng_block(g, (uint32_t)p->iter, BLK_STEP); /* philox: this iteration's variates */
    find_crossing(s, p);
    p->stale = !isfinite(p->smallest);
    p->absorbed_now = 0;
    if (p->old_el < 0) { /* the reference would fail with AttributeError (no FirstEl yet) */
        p->status = RTMC_ST_LOST;
        p->iter++;
        return 0;
    }
    check_absorption(s, p, g, p->drv_old_point, p->smallest);
    memcpy(p->drv_old_point, p->old_point, sizeof p->old_point);
    if (p->status == RTMC_ST_START) {
        const rtmc_element *e = &s->el[p->old_el];
        operation_plane(s, p->old_el, p->point_cy, p->op_S, p->op_n); /* RayTrace.py:187 */
        if (e->type == RTMC_TYPE_INTERFACE) interface_op(s, p, g, p->old_el, t);
        else if (e->type == RTMC_TYPE_LENS) lens_op(s, p, p->old_el);
        else if (e->type == RTMC_TYPE_MIRROR) mirror_op(p, p->old_el, t);
        else if (e->type == RTMC_TYPE_DUMP) dump_op(s, p, g, p->old_el, t);
        memcpy(p->drv_old_point, p->old_point, sizeof p->old_point);
    }
    int stop = p->status == RTMC_ST_INTERNAL_CONVERTED || p->status == RTMC_ST_DUMPED ||
               p->iter >= s->max_iter - 1;
    p->iter++;
    return !stop;
}

static void count_fate(const ph_t *p, rtmc_tallies *t)
{
    t->n_photons++;
    t->n_steps += (uint64_t)p->iter;
    if (p->been_absorbed) t->n_absorbed_once++;
    if (p->status == RTMC_ST_INTERNAL_CONVERTED) t->n_internal_converted++;
    else if (p->status == RTMC_ST_DUMPED) t->n_dumped++;
    else if (p->status == RTMC_ST_LOST) t->n_lost++;
    else t->n_capped++;
}

static int run_ids(const orc_scene *s, uint64_t seed, uint64_t first, uint64_t n, rtmc_tallies *out, int mode);

/* RayTrace.py:166-201 for photon ids [first, first+n), event layout of the HIP kernels */
ORC_EXPORT int orc_run(const orc_scene *s, uint64_t seed, uint64_t first, uint64_t n, rtmc_tallies *out)
{
    return run_ids(s, seed, first, n, out, 0);
}

/* the same photons with every draw made where and how the reference makes it */
ORC_EXPORT int orc_run_reference_order(const orc_scene *s, uint64_t seed, uint64_t first, uint64_t n, rtmc_tallies *out)
{
    return run_ids(s, seed, first, n, out, 2);
}

static int run_ids(const orc_scene *s, uint64_t seed, uint64_t first, uint64_t n, rtmc_tallies *out, int mode)
{
    if (!s || !out || out->n_elements != s->n_el) {
        snprintf(g_err, sizeof g_err, "bad arguments");
        return RTMC_EINVAL;
    }
    for (uint64_t k = 0; k < n; k++) {
        uint64_t id = first + k;
        rng_t g;
        memset(&g, 0, sizeof g);
        g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32);
        g.pid[0] = (uint32_t)id; g.pid[1] = (uint32_t)(id >> 32);
        g.tape_mode = mode;
        ph_t p;
        memset(&p, 0, sizeof p);
        launch_photon(s, &p, &g);
        for (;;) {
            int go = loop_iteration(s, &p, &g, out);
            if (p.stale) out->n_stale_steps++;
            if (p.absorbed_now) out->n_absorptions++;
            if (!go) break;
        }
        count_fate(&p, out);
    }
    return RTMC_OK;
}

static void soa_load(const rtmc_photon_soa *b, uint32_t k, ph_t *p, const orc_scene *s)
{
    memset(p, 0, sizeof *p);
    for (int j = 0; j < 3; j++) {
        p->L[j] = b->loc[j][k]; p->D[j] = b->dir[j][k]; p->E[j] = b->efield[j][k];
        p->drv_old_point[j] = b->old_point[j][k];
        p->old_point[j] = b->old_point[j][k];
    }
    p->wl = b->wavelength[k]; p->wl0 = b->start_wavelength[k]; p->dist_left = b->dist_left[k];
    p->psi = b->psi[k];
    p->mat = b->material[k]; p->status = b->status[k];
    p->first_el = p->old_el = b->last_element[k];
    p->was_element = b->was_element[k]; p->reflected = b->reflected[k];
    p->iter = b->iterations[k];
    p->smallest = INFINITY;
    /* the operation plane of the last element; a photon that left a cylinder sits on its tangent plane */
    if (p->old_el >= 0 && p->old_el < s->n_el) operation_plane(s, p->old_el, p->L, p->op_S, p->op_n);
}
static void soa_store(rtmc_photon_soa *b, uint32_t k, const ph_t *p, const rng_t *g, uint64_t tape_base)
{
    for (int j = 0; j < 3; j++) {
        b->loc[j][k] = p->L[j]; b->dir[j][k] = p->D[j]; b->efield[j][k] = p->E[j];
        b->old_point[j][k] = p->drv_old_point[j];
    }
    b->wavelength[k] = p->wl; b->start_wavelength[k] = p->wl0; b->dist_left[k] = p->dist_left;
    b->hit_distance[k] = p->smallest;
    b->psi[k] = p->psi;
    b->material[k] = p->mat; b->status[k] = p->status; b->last_element[k] = p->old_el;
    b->was_element[k] = p->was_element; b->reflected[k] = p->reflected; b->iterations[k] = p->iter;
    b->tape_pos[k] = (int32_t)(g->pos - tape_base);
}

/* same contract as rtmc_step_tape (include/rtmc.h) */
ORC_EXPORT int orc_step_tape(const orc_scene *s, uint32_t n, rtmc_photon_soa *b, const double *tape,
                             const uint64_t *tape_off, uint32_t n_steps, int launch, double *trace)
{
    int overrun = 0;
    for (uint32_t k = 0; k < n; k++) {
        rng_t g;
        memset(&g, 0, sizeof g);
        g.tape_mode = 1; g.tape = tape;
        g.pos = tape_off[k] + (launch ? 0 : (uint64_t)b->tape_pos[k]);
        g.end = tape_off[k + 1];
        ph_t p;
        if (launch) { memset(&p, 0, sizeof p); launch_photon(s, &p, &g); }
        else soa_load(b, k, &p, s);
        int alive = !(p.status == RTMC_ST_INTERNAL_CONVERTED || p.status == RTMC_ST_DUMPED ||
                      p.status == RTMC_ST_LOST || p.iter >= s->max_iter);
        for (uint32_t it = 0; it < n_steps && alive; it++) {
            alive = loop_iteration(s, &p, &g, NULL);
            if (trace) {
                double *r = trace + ((uint64_t)k * n_steps + it) * RTMC_TRACE_STRIDE;
                for (int j = 0; j < 3; j++) { r[j] = p.L[j]; r[3 + j] = p.D[j]; r[6 + j] = p.E[j]; }
                r[9] = p.wl; r[10] = p.dist_left; r[11] = p.status; r[12] = p.mat;
                r[13] = p.old_el; r[14] = p.smallest; r[15] = (double)(g.pos - tape_off[k]);
            }
        }
        overrun |= g.overrun;
        soa_store(b, k, &p, &g, tape_off[k]);
    }
    if (overrun) {
        snprintf(g_err, sizeof g_err, "variate tape exhausted");
        return RTMC_EINVAL;
    }
    return RTMC_OK;
}

/* single operations for the known-answer vectors */
ORC_EXPORT int orc_element_hit(const rtmc_element *e, const double *L, const double *D, int was_element, double *out)
{
    return element_intersect(e, L, D, was_element, out);
}
ORC_EXPORT void orc_cylinder_normal(const rtmc_element *e, const double *P, double *n)
{
    double S[3];
    cylinder_plane(e, P, S, n);
}
/* Lens.ExecuteOperation on a photon state (L, D, E) with plane (S, n) */
ORC_EXPORT void orc_op_lens(const double *S, const double *n, double f, double *state)
{
    ph_t p;
    memset(&p, 0, sizeof p);
    memcpy(p.L, state, 24); memcpy(p.D, state + 3, 24); memcpy(p.E, state + 6, 24);
    memcpy(p.op_S, S, 24); memcpy(p.op_n, n, 24);
    prepare_execution(&p);
    focus_thin_lens(&p, f);
    post_process(&p);
    memcpy(state, p.L, 24); memcpy(state + 3, p.D, 24); memcpy(state + 6, p.E, 24);
}
ORC_EXPORT double orc_psi(double S, const double *special, const double *nE)
{
    omat_t M;
    memset(&M, 0, sizeof M);
    M.m.align_S = S;
    if (special) {
        M.erf_m1 = special[0] + special[1] * I; M.erf_p1 = special[2] + special[3] * I;
        M.erf_m2 = special[4] + special[5] * I; M.erf_p2 = special[6] + special[7] * I;
        M.erfi_1 = special[8]; M.erfi_2 = special[9];
    }
    return lambert_beer_full_integral(&M, nE);
}
ORC_EXPORT void orc_vect_axis(const double *v, double *n1, double *n2) { vect_axis(v, n1, n2); }

/* same contract as rtmc_op_crossing */
ORC_EXPORT int orc_op_crossing(const orc_scene *s, uint32_t n, const rtmc_photon_soa *b, int32_t *element,
                               double *distance, double *points)
{
    for (uint32_t k = 0; k < n; k++) {
        ph_t p;
        soa_load(b, k, &p, s);
        if (p.old_el < 0) p.was_element = 0; /* nothing visited yet */
        find_crossing(s, &p);
        element[k] = isfinite(p.smallest) ? p.first_el : -1;
        distance[k] = p.smallest;
        memcpy(points + 3 * k, p.point_cy, 3 * sizeof(double));
    }
    return RTMC_OK;
}

/* same contract as rtmc_op_intersect */
ORC_EXPORT int orc_op_intersect(const orc_scene *s, uint32_t n, const double *loc, const double *dir, double *out)
{
    for (uint32_t k = 0; k < n; k++)
        for (int i = 0; i < s->n_el; i++) {
            double *o = out + ((uint64_t)k * s->n_el + i) * 3;
            if (!element_intersect(&s->el[i], loc + 3 * k, dir + 3 * k, 0, o)) o[0] = o[1] = o[2] = NAN;
        }
    return RTMC_OK;
}

/* single operations on caller-supplied state, for the known-answer vectors:
 * in/out rows of 9 doubles (L, D, E); plane = S[3], n[3]; returns TIR flag,
 * reflectance and sgn in res[3] */
ORC_EXPORT void orc_op_interface(const double *S, const double *n, double n1, double n2, const double *state,
                                 double *res, double *refracted, double *reflected)
{
    ph_t p;
    memset(&p, 0, sizeof p);
    memcpy(p.L, state, 24); memcpy(p.D, state + 3, 24); memcpy(p.E, state + 6, 24);
    memcpy(p.op_S, S, 24); memcpy(p.op_n, n, 24);
    prepare_execution(&p);
    res[0] = is_total_reflect(&p, n1, n2);
    ph_t q = p;
    res[1] = reflectance(&q, n1, n2);
    res[2] = p.sgn;
    q = p;
    refract_on_plane(&q, n1, n2);
    post_process(&q);
    memcpy(refracted, q.L, 24); memcpy(refracted + 3, q.D, 24); memcpy(refracted + 6, q.E, 24);
    q = p;
    reflect_on_plane(&q);
    post_process(&q);
    memcpy(reflected, q.L, 24); memcpy(reflected + 3, q.D, 24); memcpy(reflected + 6, q.E, 24);
}

ORC_EXPORT int64_t orc_find_closest(const double *A, int64_t n, double t) { return find_closest(A, n, t); }
ORC_EXPORT int orc_check_sides(const double *v1, const double *v2, const double *S, const double *n)
{
    return check_sides(v1, v2, S, n);
}
