/*
 * rtmc.h -- C ABI of librtmc.so, the MI355X photon-transport library.
 *
 * The reference (mohrezaeib/RayTracer) is pure Python and has no FFI of its
 * own; the boundary this library replaces is the body of
 * RayTrace.runSimulation (reference RayTrace.py:157-204): the per-photon
 * `while active:` loop (RayTrace.py:178-195) and everything it calls.  A host
 * program describes the scene once with plain arrays (rtmc_scene_desc), then
 * asks for ranges of photon ids to be traced and receives additive tallies.
 *
 * Conventions
 *   - every function returns 0 on success or a negative RTMC_E* code; the
 *     text of the last failure of the calling thread is rtmc_last_error();
 *   - nothing throws or aborts across this boundary;
 *   - all pointers are host pointers; the library copies what it keeps;
 *   - lengths in cm, wavelengths in nm, concentrations in mmol/L (the units
 *     of the reference's job files);
 *   - a scene handle is immutable after creation and may be used from several
 *     host threads; calls on different devices run concurrently, calls on the
 *     same device are run one after the other by the library.
 */
#ifndef RTMC_H
#define RTMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMC_ABI_VERSION 1

/* error codes */
#define RTMC_OK 0
#define RTMC_EINVAL (-1)    /* malformed description or argument          */
#define RTMC_EUNSUPPORTED (-2) /* feature of the reference not built yet  */
#define RTMC_EDEVICE (-3)   /* HIP runtime failure                        */
#define RTMC_ENOMEM (-4)
#define RTMC_ELIMIT (-5)    /* scene exceeds a compiled-in limit          */

/* geometry codes: index into the reference's `shapes` list, RayTrace.py:78 */
#define RTMC_GEOM_RECTANGLE 1
#define RTMC_GEOM_CIRCLE 2
#define RTMC_GEOM_CYLINDER 3
#define RTMC_GEOM_TRIANGLE 4
/* element type codes: index into the reference's `types` list, RayTrace.py:79 */
#define RTMC_TYPE_MIRROR 1
#define RTMC_TYPE_INTERFACE 2
#define RTMC_TYPE_LENS 3
#define RTMC_TYPE_DUMP 4

/* photon status codes (reference: the strings assigned to Photon.Status) */
#define RTMC_ST_START 0
#define RTMC_ST_ABSORBED 1
#define RTMC_ST_REEMITTED_DONOR 2
#define RTMC_ST_REEMITTED_ACCEPTOR 3
#define RTMC_ST_INTERNAL_CONVERTED 4
#define RTMC_ST_DUMPED 5
#define RTMC_ST_LOST 6 /* no element found and none visited before (the reference raises) */

/* number of 1-nm bins of a detector's incident spectrum, 200..3999 nm
 * (reference Elements.py:143-144) */
#define RTMC_SPECTRUM_BINS 3800
#define RTMC_SPECTRUM_FIRST_NM 200

/* One optical element = reference OpticalObject(Type, Geometry).
 * Rectangle: Geometries.py:8-57.  `e1`/`e2` are the unit in-plane axes the
 * reference derives in the constructor (Geometries.py:17-29, normalised at
 * :40-41); `half1` pairs with e1 ("Height"), `half2` with e2 ("Width").
 * Circle (Geometries.py:60-86): support, unit normal, half1 = Radius.
 * Cylinder (Geometries.py:90-209): `support` and `normal` hold the two end
 * points of the axis as the reference's constructor receives them
 * (SupportVector, DirectionVector), half1 = Radius; the wall has no caps.
 * Triangle (Geometries.py:212-255): e1, e2, vertex3 = the three corners;
 * support = their centroid, normal = unit((e2 - e1) x (vertex3 - e1))
 * (RTUtils.py:126-133). */
typedef struct rtmc_element {
    int32_t geometry;   /* RTMC_GEOM_*                                     */
    int32_t type;       /* RTMC_TYPE_*                                     */
    double support[3];  /* SupportVector                                   */
    double normal[3];   /* DirectionVector, unit length                    */
    double e1[3];
    double e2[3];
    double half1;
    double half2;
    int32_t material1;  /* Interface: MaterialSide1 (index into materials) */
    int32_t material2;  /* Interface: MaterialSide2                        */
    int32_t eqe_table;  /* Dump: index into `tables` of a detection
                           spectrum (Elements.py:161-165), or -1           */
    int32_t reserved;
    double focal_length; /* Lens (Elements.py:117-132)                     */
    double vertex3[3];   /* Triangle: third corner                         */
} rtmc_element;

/* A two-column table (wavelength, value) stored as 2*len doubles,
 * row-major, inside rtmc_scene_desc.table_data. */
typedef struct rtmc_table {
    int64_t offset; /* in doubles */
    int64_t len;    /* rows       */
} rtmc_table;

/* One medium = reference Materials.MaterialObj after its constructor has
 * processed the spectra (Materials.py:38-74).  `abs_*` are the cleaned,
 * peak-normalised absorption tables, `em_*` the cleaned emission tables
 * normalised to unit sum (NOT yet weighted by interval width). */
typedef struct rtmc_material {
    int32_t name_id;     /* materials that the reference would compare equal
                            by name (Elements.py:45) share one id          */
    int32_t is_absorber;
    int32_t is_aligned;
    int32_t fix_dist;
    double n;            /* refractive index                               */
    double conc_a, conc_b;
    double emax_A, emax_D;
    double Q_A, Q_D;
    double tau_A, tau_D;
    double fix_dist_R;
    double align_vect[3];
    double align_S;
    double abs_em_angle;
    double ol_DA, ol_AD; /* Foerster overlap sums (Materials.py:73-74)     */
    double P_ET_DA, P_ET_AD; /* transfer probabilities
                                (MonteCarloFRET.py:195-237)                */
    /* aligned acceptor dyes (is_aligned): the orientation integral of
       PhotonObject.LambertBeerFullIntegral (PhotonObject.py:267-281) is
       |psi_P (x^2 + y^2) + psi_Q z^2| for the unit field vector (x, y, z) in the
       alignment frame; the host evaluates the two real constants from
       align_S with the complex erf/erfi the reference uses (psi_P = 0,
       psi_Q = 1 below the reference's 1e-10 threshold).  align_frame is
       np.roll(MakeVectAxis(align_vect), 2, axis=1), row-major
       (PhotonObject.py:295, RTUtils.py:51). */
    double psi_P, psi_Q;
    double align_frame[9];
    rtmc_table abs_A, abs_D, em_A, em_D;
} rtmc_material;

/* Photon source = the constructor arguments of reference PhotonObject.Photon
 * that GeneratePhoton uses (PhotonObject.py:61-97). */
typedef struct rtmc_source {
    int32_t direction_mode;  /* 0 "Directional Strict", 1 "Directional Isotropic" */
    int32_t wavelength_mode; /* 0 fixed, 1 drawn from `spectrum`           */
    double location[3];
    double direction[3];     /* EmissionDirection as given (not normalised) */
    double frame[9];         /* row-major 3x3; hemisphere sample d becomes
                                d . frame (PhotonObject.py:71-72)          */
    double wavelength;
    rtmc_table spectrum;     /* sun table as read from file (before the
                                x lambda and interval weighting of EmitSun,
                                PhotonObject.py:39-46)                     */
    int32_t start_material;
    int32_t respect_polarization;
    double scale_factor;
    double area[2];          /* full widths along x and y of a launch
                                rectangle centred on `location` (0, 0: point
                                source).  The reference keeps this launch area
                                as commented-out lines with fixed sizes
                                (PhotonObject.py:74-84): two uniforms drawn
                                after the direction, x then y              */
} rtmc_source;

typedef struct rtmc_scene_desc {
    int32_t abi_version;     /* RTMC_ABI_VERSION                           */
    int32_t n_elements;
    int32_t n_materials;
    int32_t n_tables;
    const rtmc_element *elements;   /* list order is tie-break order       */
    const rtmc_material *materials;
    const rtmc_table *tables;       /* detector EQE tables                 */
    const double *table_data;
    int64_t table_data_len;         /* in doubles                          */
    rtmc_source source;
    int32_t max_iterations;  /* loop-iteration cap; the reference's
                                `i >= 1e4` test allows 10001 (RayTrace.py:193) */
    int32_t reserved;
} rtmc_scene_desc;

/* Additive results of tracing a range of photon ids.  The caller allocates
 * the arrays (n_elements entries each; `spectrum` n_elements x
 * RTMC_SPECTRUM_BINS, row per element, all zero for non-detectors) and zeroes
 * them before the first call; every call ADDS to them.
 * Mirrors: Dump.Counter / EnergyCounter / SpectralCounter / IncidentSpectrum
 * (Elements.py:139-165), Mirror/Interface.BounceCounter (Elements.py:15,39),
 * Photon.BeenAbsorbed[1] (RayTrace.py:197-198). */
typedef struct rtmc_tallies {
    int32_t n_elements;
    int32_t reserved;
    uint64_t n_photons;
    uint64_t n_steps;          /* loop iterations executed                  */
    uint64_t n_absorbed_once;  /* photons absorbed at least once            */
    uint64_t n_absorptions;    /* absorption events                         */
    uint64_t n_internal_converted;
    uint64_t n_dumped;
    uint64_t n_capped;         /* stopped by the iteration cap              */
    uint64_t n_lost;           /* RTMC_ST_LOST                              */
    uint64_t n_stale_steps;    /* iterations without a valid hit            */
    uint64_t *counter;         /* per element                               */
    uint64_t *spectral;        /* per element                               */
    uint64_t *bounce;          /* per element                               */
    uint64_t *energy_fx;       /* per element: sum of lambda_start/lambda in
                                  units of 2^-24, rounded per photon        */
    uint64_t *spectrum;        /* n_elements * RTMC_SPECTRUM_BINS           */
} rtmc_tallies;
#define RTMC_ENERGY_FX_SCALE 16777216.0

/* Structure-of-arrays photon batch used by the step-wise entry point.
 * Each pointer addresses `n` values.  Field meaning = the attributes of the
 * reference's Photon object (PhotonObject.py:9-37) plus the two loop
 * variables of RayTrace that outlive an iteration. */
typedef struct rtmc_photon_soa {
    double *loc[3];
    double *dir[3];
    double *efield[3];
    double *old_point[3];   /* RayTrace.OldPoint                          */
    double *wavelength;
    double *start_wavelength;
    double *dist_left;      /* DistanceLeftToAbsorption                   */
    double *hit_distance;   /* out: SmallestDistance of the last iteration */
    double *psi;            /* Photon.Psi: orientation integral of the last
                               free-path draw in an aligned medium
                               (PhotonObject.py:299), used again by the
                               donor/acceptor coin (PhotonObject.py:241)  */
    int32_t *material;
    int32_t *status;        /* RTMC_ST_*                                  */
    int32_t *last_element;  /* RayTrace.FirstEl as an index, -1 = none    */
    int32_t *was_element;
    int32_t *reflected;
    int32_t *iterations;    /* loop iterations done so far                */
    int32_t *tape_pos;      /* in/out: variates consumed from the tape    */
} rtmc_photon_soa;

typedef struct rtmc_scene rtmc_scene;

/* precision of the device arithmetic */
#define RTMC_F32 0 /* production path                                      */
#define RTMC_F64 1 /* same kernels instantiated in double (validation)     */

int rtmc_version(void);
const char *rtmc_last_error(void);

/* Replaces RayTrace.CreateOpticalElementList (RayTrace.py:70-114) and the
 * photon-independent parts of the loop that the reference recomputes for
 * every photon: RectifySpectrum + cumulative sums for the inverse-CDF draws
 * (RTUtils.py:152-158, MonteCarloFRET.py:258-279), per-wavelength absorption
 * coefficients (PhotonObject.py:288-289), nearest-bin lookups
 * (RTUtils.py:15-22). */
int rtmc_scene_create(const rtmc_scene_desc *desc, rtmc_scene **out);
void rtmc_scene_destroy(rtmc_scene *scene);

/* Replaces the photon loop of RayTrace.runSimulation (RayTrace.py:166-201)
 * for photon ids [first_photon, first_photon + n_photons).  Variates come
 * from Philox4x32-10 keyed by `seed` and counted by (photon id, loop iteration,
 * purpose of the draw), so the result does not depend on how a job is split
 * into calls or devices.  Ranges longer than 1e11 photons are traced in
 * several launches.  Blocks until the tallies are on the host. */
int rtmc_run(rtmc_scene *scene, int device, int precision, uint64_t seed,
             uint64_t first_photon, uint64_t n_photons, rtmc_tallies *out);

/* rtmc_run for callers that cannot allocate: tight serving loops, harnesses
 * that forbid hipMalloc once they are running.  It has no counterpart in the
 * reference.  Like rtmc_run it is synchronous: memsets, kernel and copy-back go
 * to the scene's own stream and the call returns when the tallies are on the
 * host, so it cannot become part of a HIP graph a caller is capturing on another
 * stream.  Same arguments, same kernels and bit-identical tallies as
 * rtmc_run, but the device tally block and the work counter -- the only device
 * memory a trace needs beyond the resident scene -- live in a static array of
 * the library (65536 64-bit tally words + the counter, 516 KB on every GPU the
 * library is used on, held until the process ends) and the call allocates
 * nothing, on the device or on the host.
 *   - The scene must already be resident on `device`: one earlier rtmc_run on
 *     it (for zero photons, if nothing is to be traced yet) uploads it.
 *     Otherwise RTMC_EINVAL.
 *   - A scene whose tally block (7 + 3 n_elements + detectors x distinct 1-nm
 *     wavelength bins, in words) exceeds 65536 words is refused with
 *     RTMC_ELIMIT; rtmc_run has no such limit.
 *   - All calls on one GPU share that GPU's array; it is cleared at the start
 *     of every launch and keeps nothing between calls.  The library runs the
 *     calls of one process on one GPU one after the other (a per-GPU lock), so
 *     threads may call it concurrently; it still blocks until the tallies are
 *     on the host. */
int rtmc_run_noalloc(rtmc_scene *scene, int device, int precision, uint64_t seed,
                     uint64_t first_photon, uint64_t n_photons, rtmc_tallies *out);

/* Device time of the trace kernel of the calling thread's last rtmc_run, in
 * milliseconds, measured with HIP events on the stream the kernel ran on. */
double rtmc_last_kernel_ms(const rtmc_scene *scene, int device);

/* Replaces `n_steps` iterations of the loop body (RayTrace.py:178-195:
 * CheckElementsForCrossingPoint, Photon.CheckForAbsorption, the element's
 * ExecuteOperation) for `n` photons whose state the caller supplies.  The
 * variates are not generated but read from `tape` in the order the reference
 * consumes them (photon k uses tape[tape_off[k] + tape_pos[k] ...], never
 * beyond tape_off[k+1]): uniforms where it calls rand()/choice(), standard
 * normals for randn(), unit-scale exponentials for exponential().
 * If `launch` is non-zero the photons are first generated as
 * Photon.ResetPhoton does (PhotonObject.py:49-97) and the input state is
 * ignored.  A photon that ends (Dumped, InternalConverted, cap) stops early.
 * If `trace` is non-NULL it receives, per photon and iteration,
 * RTMC_TRACE_STRIDE doubles (n * n_steps * RTMC_TRACE_STRIDE in all,
 * photon-major; unused rows are left untouched). */
#define RTMC_TRACE_STRIDE 16
/* trace row: loc[3] dir[3] efield[3] wavelength dist_left status material
 *            hit_element hit_distance tape_pos */
int rtmc_step_tape(rtmc_scene *scene, int device, int precision, uint32_t n,
                   rtmc_photon_soa *photons, const double *tape,
                   const uint64_t *tape_off, uint32_t n_steps, int launch,
                   double *trace);

/* Per-operation entry points used for known-answer tests; each replaces one
 * reference method on `n` independent inputs (arrays of n*3 doubles unless
 * noted).  Geometry.FindIntersections for every element of the scene
 * (Geometries.py:32-52): out_points is n * n_elements * 3, NaN where the
 * reference returns [nan, nan, nan]. */
int rtmc_op_intersect(rtmc_scene *scene, int device, int precision, uint32_t n,
                      const double *loc, const double *dir, double *out_points);

/* RayTrace.CheckElementsForCrossingPoint (RayTrace.py:120-154) for `n`
 * photon states: the nearest element whose point the photon may go to next.
 * Read from `photons`: loc, dir, old_point, last_element, was_element,
 * reflected (Cylinder.FindIntersections depends on was_element,
 * Geometries.py:102-107).  element[k] is the index of the element found or -1
 * (the reference then keeps its previous FirstEl), distance[k] its
 * SmallestDistance (+inf if none), points[3k..] Photon.PointCy ((0,0,0) if
 * none). */
int rtmc_op_crossing(rtmc_scene *scene, int device, int precision, uint32_t n,
                     const rtmc_photon_soa *photons, int32_t *element, double *distance,
                     double *points);

/* The single operations the reference's Photon and element objects offer to a
 * script that moves one photon by hand (its TestPhoton.py does), each for `n`
 * photon states of `photons`, which are read and written in place:
 *   RTMC_PHOP_GENERATE      Photon.GeneratePhoton / ResetPhoton (PhotonObject.py:49-97); the input
 *                           state is ignored
 *   RTMC_PHOP_ABSORPTION    Photon.CheckForAbsorption(OldPoint, SmallestDistance)
 *                           (PhotonObject.py:206-264) with old_point and hit_distance of the batch
 *                           and Photon.PointCy = args[0..2]; pathway, re-emission and the new free
 *                           path (MonteCarloFRET.py:63-193, PhotonObject.py:283-305) included
 *   RTMC_PHOP_REFLECT       Photon.ReflectOnPlane (PhotonObject.py:145-150)
 *   RTMC_PHOP_REFRACT       Photon.RefractOnPlane(n1, n2) (PhotonObject.py:195-199)
 *   RTMC_PHOP_LENS          Photon.FocusThinLens(f) (PhotonObject.py:158-186)
 *   RTMC_PHOP_POLARIZATION  Photon.UpdatePolarization(OldDirection) (PhotonObject.py:117-129)
 *   RTMC_PHOP_REFLECTANCE   Interface.IsTotalReflect and Interface.Reflectance (Elements.py:75-109):
 *                           out[2k] = 1 if totally reflected, out[2k+1] = the reflectance
 *                           (NaN where the reference's is)
 *   RTMC_PHOP_FREEPATH      Photon.LamBeerDistance (PhotonObject.py:283-305): one exponential
 *                           variate if the photon's medium absorbs, none otherwise
 *   RTMC_PHOP_EXECUTE       PrepareExecution + the ExecuteOperation of element number args[6] +
 *                           PostProcessExecution (PhotonObject.py:135-143; Elements.py:18-24,
 *                           41-73, 125-129, 152-169) with Photon.PointCy = args[0..2] (cylinders
 *                           take their tangent plane there, Geometries.py:207-209);
 *                           out[2k] = 1 if the element counted a bounce, out[2k+1] = 1 if a
 *                           detector counted the photon for its SpectralCounter
 * `args` holds RTMC_PHOP_ARGS doubles per photon: [0..2] a point (the support vector of
 * Photon.OperationPlane, or PointCy), [3..5] the normal of Photon.OperationPlane, [6..8] the
 * scalars of the call (n1, n2, Photon.sgn | f | element number), [9..11] OldDirection.  `out`
 * (2 doubles per photon) may be NULL for the operations that return nothing.
 * Variates come from `tape` as in rtmc_step_tape (tape_pos of the batch is read and advanced;
 * GENERATE starts at 0).  With `uniform_tape` non-zero every tape value is a uniform in
 * [0, 1) and the library turns it into the normal or exponential variate an operation asks
 * for (inverse distribution function), so that a caller who cannot know which kind the next
 * draw will be can still supply the randomness. */
#define RTMC_PHOP_GENERATE 1
#define RTMC_PHOP_ABSORPTION 2
#define RTMC_PHOP_REFLECT 3
#define RTMC_PHOP_REFRACT 4
#define RTMC_PHOP_LENS 5
#define RTMC_PHOP_POLARIZATION 6
#define RTMC_PHOP_REFLECTANCE 7
#define RTMC_PHOP_EXECUTE 8
#define RTMC_PHOP_FREEPATH 9
#define RTMC_PHOP_ARGS 12
int rtmc_op_photon(rtmc_scene *scene, int device, int precision, int op, uint32_t n,
                   rtmc_photon_soa *photons, const double *args, double *out,
                   const double *tape, const uint64_t *tape_off, int uniform_tape);

/* Number of wavelengths a photon can carry in this scene (source lines or
 * table rows plus every emission-table row) and their values. */
int rtmc_scene_wavelengths(const rtmc_scene *scene, int64_t *count, double *values,
                           int64_t capacity);

/* What rtmc_scene_create found in the scene (no counterpart in the reference; for tests and for
 * users who wonder which kernel their scene runs): info[0] = runs of coplanar rectangles in the float32
 * hit table, info[1] = rectangles in them, info[2] = float32 guard distance: 0 none, 1 on every
 * search (coplanar overlapping elements), 2 until a photon's first absorption (a strict source aimed at a
 * seam), info[3] = 1 if the scene runs the kernel that holds every feature. */
int rtmc_scene_info(const rtmc_scene *scene, int32_t info[4]);

/* The closed boxes ("cells") rtmc_scene_create found among the scene's rectangles (no counterpart in
 * the reference, which tries every element on every search, RayTrace.py:128-150; the float32 kernels
 * answer the search of a photon inside such a box from its six faces): info[0] = cells in its low half,
 * of which -- high half -- regions (empty parts of a box that holds several others, in front of the faces
 * of those, which serve as cells too), info[1] = cells in which photons can be confined (nothing but the
 * faces, or one nested cell, inside; every region), info[2] = the cell the source sits in (0 none), info[3] =
 * the feature bits of the scene's float32 trace kernel (bit 9, value 512: the kernels that use the cells). */
int rtmc_scene_cells(const rtmc_scene *scene, int32_t info[4]);

#ifdef __cplusplus
}
#endif
#endif /* RTMC_H */
