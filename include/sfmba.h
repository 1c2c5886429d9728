/*
 * sfmba.h -- C ABI of the MI355X-native bundle-adjustment back end that drops in
 * behind sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle().
 *
 * Reference interface replaced (all paths relative to the reference checkout):
 *   SfMToyLib/SfMBundleAdjustmentUtils.h:44-49    adjustBundle() declaration
 *   SfMToyLib/SfMBundleAdjustmentUtils.cpp:58-97  SimpleReprojectionError (2 residuals; 6/3/1 params)
 *   SfMToyLib/SfMBundleAdjustmentUtils.cpp:171-179 ceres::Solver::Options + ceres::Solve
 *   SfMToyLib/SfMBundleAdjustmentUtils.cpp:182-185 termination gate (only CONVERGENCE is written back)
 *
 * The reference has no FFI: its "operator API" for this path is the single static C++
 * function above, whose body hands flat double arrays (camera 6-vectors, 3D points, one
 * shared focal) to ceres::Problem / ceres::Solve.  This header is the boundary a
 * maintainer binds instead of Ceres: plain pointers and sizes, no C++/torch types.
 * The C++ shim that keeps the reference signature lives in
 * sfm-toy-library_amd/host/SfMBundleAdjustmentUtils.cpp.
 *
 * Conventions
 *   camera j : cam6[6*j+0..2] = angle-axis (Rodrigues) rotation, cam6[6*j+3..5] = translation,
 *              world->camera: p = Rot(w) * X + t                      (BA.cpp:67-74)
 *   point i  : pt3[3*i+0..2]
 *   focal    : one scalar shared by every camera                      (BA.cpp:92,138,164)
 *   obs k    : (obs_cam[k], obs_pt[k], obs_xy[2k], obs_xy[2k+1]) with the principal point
 *              already subtracted                                     (BA.cpp:149-153)
 *   residual : r = focal * (p.x/p.z, p.y/p.z) - obs                   (BA.cpp:76-86)
 *   cost     : 1/2 * sum ||r||^2  (Ceres convention)
 * Cameras/points that no observation references are not part of the problem and are
 * left untouched (Ceres only sees parameter blocks passed to AddResidualBlock, BA.cpp:160).
 */
#ifndef SFMBA_H_
#define SFMBA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFMBA_ABI_VERSION 6

#if defined(__GNUC__)
#define SFMBA_API __attribute__((visibility("default")))
#else
#define SFMBA_API
#endif

/* ceres::TerminationType values the reference tests against (BA.cpp:182). */
enum {
    SFMBA_CONVERGENCE    = 0,
    SFMBA_NO_CONVERGENCE = 1,
    SFMBA_FAILURE        = 2
};

enum { SFMBA_LINEAR_CHOLESKY = 0,   /* exact Schur + dense LLT == DENSE_SCHUR (BA.cpp:172), always factorised */
       SFMBA_LINEAR_PCG      = 1,   /* exact Schur + two-level block-Jacobi PCG on the dense reduced system, stopped at pcg_tolerance
                                       (inexact Newton: parameters agree with the exact solve to ~1e-8, cost to 1e-12) */
       SFMBA_LINEAR_AUTO     = 2 }; /* THE DEFAULT (library and drop-in shim): the reference's DENSE_SCHUR result at the cost of the
                                       cheapest solver that delivers it.  Up to 256 reduced unknowns (the reference's own data sets):
                                       CHOLESKY.  Above: the same PCG run to a plain relative residual of min(pcg_tolerance, 1e-12)
                                       -- the step then agrees with the factorised solve to ~1e-10 relative, below what the float
                                       containers of adjustBundle() resolve -- and, if the CG has not converged after
                                       pcg_max_iters (0 = min(4 dim, 200)) iterations, the SAME linearisation is solved by CHOLESKY
                                       instead (the matrix is re-formed unpreconditioned; nothing is skipped).  A CG that BREAKS DOWN
                                       (non-positive curvature, NaN) is not retried: the step is invalid, exactly as after a failed
                                       factorisation (radius halved, FAILURE after five in a row).  A linearisation that needed more
                                       CG iterations than a factorisation costs (~3.3 per block column of 64) makes the following
                                       ones OF THE SAME SOLVE go straight to CHOLESKY (the preference is not remembered across
                                       solves: a resident problem solved twice from the same point takes the same path twice).
                                       Sharded solves (sfmba_problem_solve_sharded, sfmba_shard_*) have no fallback: AUTO there is the
                                       CG at 1e-12 with the step forced at pcg_max_iters.  Above 1280 reduced unknowns in
                                       SFMBA_PRECISION_F32J the CG iterates on an fp32-rounded copy of the preconditioned matrix
                                       (pcg_f32_matrix = -1 keeps fp64): the converged step is that of the rounded matrix, ~1e-7
                                       relative from the fp64 one -- inside what F32J promises, not the DENSE_SCHUR digits. */

enum { SFMBA_PRECISION_F64  = 0,    /* everything fp64 (parity mode) */
       SFMBA_PRECISION_F32J = 1 };  /* fp32 Jacobian blocks AND fp32 observation coordinates (BASELINE config 3; the reference's
                                       observations are cv::Point2f anyway, BA.cpp:149-153 -- a caller holding genuinely double
                                       observations loses ~1e-4 px at 2k-px coordinates).  What is summed in which precision: a lane's OWN
                                       partial sum of block products is fp32 (at most 8 pair products of a 6x6 block of the reduced matrix,
                                       ceil(track / 4) point-block products of a point); every sum ACROSS lanes, chunks, points and ranks is
                                       fp64, and so are the residuals, the cost, the reduced system and its solve (ABI v4 summed a block's
                                       up-to-512 pair products in fp32 before widening).  The back-substitution evaluates the Jacobian-type terms of an observation (J x step,
                                       C = B~ L^-T) from an fp32 copy of the camera's R, t and step; the TRIAL residuals, which decide accept / reject, from the fp64 pose.
                                       What to expect (tests/test_gpu_baseline_parity.py): final cost within 1e-6
                                       relative (measured 3e-13) and final RMS within 1e-4 px of the fp64 solve (measured < 1e-9 px),
                                       parameters within ~2e-5 -- EXCEPT points on weakly constrained tracks: a point seen by two
                                       nearly parallel views has almost no depth information, its 3x3 block is ill-conditioned and
                                       the fp32 rounding of its Jacobian moves it by up to ~1e-3..5e-3 scene units along the ray at
                                       (numerically) the same cost (cfg3_banded: 8e-4 and 2.1e-3 seen; the worst of 42 000 points of a 520-camera
                                       path: 0.03 .. 0.07, with 99.9 % of the points within 2e-3).  Use F64 if such points'
                                       coordinates matter beyond that. */

/* The F32J ERROR BUDGET (round 6): what SFMBA_PRECISION_F32J may cost against the library's fp64 mode under the same solver, in the units of the caller.
 * tests/test_gpu_f32j_budget.py measures every line on nine problem shapes, two solvers and two other scene scales (|t| ~ 1e3 and ~ 5e-2) and FAILS beyond
 * it: a faster fp32 kernel that moves a result past one of these numbers is a regression, not a tolerance to widen (rounds 4 - 5 loosened three tests to
 * make room for speed; this is where that stops).  scale = max(1, max |t|) of the solution: the budget does not depend on the units of the scene. */
/*                                             budget     measured, round 6 (profiles/r06_c_f32j_budget.txt: the worst of 22 runs)                          */
#define SFMBA_F32J_BUDGET_COST_REL     1e-8    /* 4.8e-10  final cost, relative (north_star asks 1e-6 at BASELINE config 2) */
#define SFMBA_F32J_BUDGET_RMS_PX       1e-6    /* 1.0e-10  final RMS reprojection error, px (north_star asks 1e-4) */
#define SFMBA_F32J_BUDGET_ROTATION     2e-5    /* 5.5e-6   angle-axis components of a camera, rad; LM iterations and termination type: identical */
#define SFMBA_F32J_BUDGET_TRANSLATION  1e-5    /* 1.6e-6   camera translation / scale */
#define SFMBA_F32J_BUDGET_FOCAL_REL    1e-6    /* 3.0e-9   the shared focal, relative */
#define SFMBA_F32J_BUDGET_POINT_P999   2e-5    /* 1.5e-6   99.9th percentile of the point displacement / scale (the weakly constrained tracks above are the rest) */
/* WHERE THE BUDGET DOES NOT APPLY (round 6, tests/fuzz_parity.py: 1 500 random shapes against the oracle; the fp64 modes follow it to <= 2e-8 of the
 * final cost on every one of them, over runs of up to 343 LM iterations).  F32J is for WELL-DETERMINED problems -- several observations per point, a few
 * hundred points per view, as every reconstruction the reference produces is.  On problems with about as many residuals as parameters (each point seen
 * twice, a handful of points for dozens of views) or dominated by gross outliers, the LM run takes tens of iterations, the trust region grows past ~1e7,
 * the damping falls below the rounding of the fp32 blocks and the reduced matrix stops being positive definite: steps become INVALID.  The library
 * then divides the radius by 8 instead of Ceres' 2 (F32J only; fp64 keeps the reference's rule) and goes on, and the run converges -- but to a final
 * cost that can differ from the fp64 one by 1e-4 .. 1e-3 relative, with a different iteration count, and not reproducibly from one run to the next (the
 * order in which fp64 atomics arrive is enough to tip an accept / reject decision on such a landscape; SFMBA_CREATE_DETERMINISTIC fixes the order).
 * Exactly satisfiable toy problems (final cost ~1e-8 of the initial one) end one LM iteration earlier or later than the oracle in F32J and with
 * the CG at 1e-8: both at a cost of zero for every purpose.  Use the default -- fp64 with SFMBA_LINEAR_AUTO -- when in doubt. */

/* Return codes of every entry point. */
enum {
    SFMBA_OK              = 0,
    SFMBA_ERR_INVALID_ARG = 1,
    SFMBA_ERR_NO_DEVICE   = 2,   /* HIP runtime/device missing: the product path never falls back to CPU */
    SFMBA_ERR_HIP         = 3,
    SFMBA_ERR_ALLOC       = 4,
    SFMBA_ERR_CAPACITY    = 5    /* an output array is too small: the required length was returned, nothing else written */
};

typedef struct sfmba_options {
    int    max_iters;                 /* 500    BA.cpp:174 */
    double max_seconds;               /* 10.0   BA.cpp:176; <= 0 disables the wall-clock limit */
    double function_tolerance;        /* 1e-6   Ceres default */
    double gradient_tolerance;        /* 1e-10  Ceres default */
    double parameter_tolerance;       /* 1e-8   Ceres default */
    double initial_radius;            /* 1e4    Ceres default initial_trust_region_radius */
    double max_radius;                /* 1e16 */
    double min_radius;                /* 1e-32 */
    double min_relative_decrease;     /* 1e-3 */
    double min_lm_diagonal;           /* 1e-6 */
    double max_lm_diagonal;           /* 1e32 */
    int    jacobi_scaling;            /* 1 */
    int    max_consecutive_invalid_steps; /* 5 */
    int    linear_solver;             /* SFMBA_LINEAR_* */
    int    precision;                 /* SFMBA_PRECISION_* */
    double pcg_tolerance;             /* CG residual tolerance (1e-8) */
    int    pcg_max_iters;             /* 0 = 4*dim */
    int    verbose;                   /* 0 silent (BA.cpp:177), 1 per-iteration lines on stderr */
    int    pcg_anchored;              /* 1: inside an LM solve the CG tolerance is anchored to the FIRST iteration's right-hand side,
                                         |r| <= tol * max(|b_k|, |b_first|), never looser than 1e-4 |b_k| -- every LM step is then solved
                                         to the same ABSOLUTE accuracy (dense_solver.hip).  0: plain relative residual. */
    /* ---- ABI v4: behaviour switches that were environment variables only (a C caller could not set them per problem or
       thread-safely).  0 = library default, 1 = on, -1 = off.  ABI v4 let the environment variable named beside each switch
       override the field; since ABI v5 NOTHING below sfmba_problem_create* reads the environment: the fields are the only way
       (what is still read from the environment, when a problem is BUILT: SFMBA_DETERMINISTIC, SFMBA_PAIR_LPB, SFMBA_PAIR_LIMIT,
       SFMBA_BUILD_TIMING). ---- */
    int    pcg_coarse_space;          /* SFMBA_PCG_COARSE         default on : two-level CG preconditioner (8 gauge vectors).  Where the reduced
                                         matrix is sparsely filled (< 1/2 of its blocks) with >= 90 % of the blocks within a quarter of the cyclic camera
                                         order -- views registered along a path -- and >= 32 cameras, the seven similarity vectors are used restricted to
                                         overlapping SEGMENTS of the camera order (eight up to 213 cameras, cameras / 25 <= 20 up to 1007: 3 - 15x fewer CG
                                         iterations there; AUTO keeps that CG above 213 cameras instead of factorising).
                                         1 = the eight global vectors only, 2 = the segments wherever they apply.
                                         The sharded solve keeps the eight global vectors (the choice would have to be agreed between the ranks). */
    int    pcg_symmetric;             /* ABI v6 (the slot ABI v4 called pcg_persistent; reserved in v5)  default on : the streaming CG (d > 1280, no
                                         segmented coarse space, not a deterministic handle) reads ONE triangle of S~ per iteration and uses every entry
                                         twice (k_pcg_iter_sym); the pair pass then writes that triangle only.  -1 = both triangles, the round-5 kernels */
    int    pcg_f32_matrix;            /* SFMBA_PCG_F32_MATRIX     default on : F32J + streaming CG (d > 1280) store S~ in fp32 */
    int    early_linearise;           /* SFMBA_EARLY_LINEARISE    default on : next linearisation enqueued before the host reads the verdict */
    int    shard_two_phase;           /* SFMBA_SHARD_TWO_PHASE    default on : sharded CG path exchanges (A) diagonal data, (B) preconditioned blocks */
    int    shard_f32_exchange;        /* SFMBA_SHARD_F32_EXCHANGE default on : exchange (B) in fp32 where the CG stores S~ in fp32 anyway */
    int    shard_distributed_cg;      /* SFMBA_SHARD_DIST_CG      default off: sharded CG path WITHOUT the redundant solve -- exchange (B) is a
                                         reduce-scatter of the upper-triangle blocks of S~ into ranges of block rows (half the bytes of the
                                         all-reduce), every rank multiplies the blocks it owns, one all-reduce of ld doubles per CG
                                         iteration's partial product (needs sfmba_problem_set_reduce_scatter when world > 1).
                                         2: the same CG with the product formed IMPLICITLY -- no pair pass, no
                                         exchange (B) at all: per CG iteration every rank applies its own points' W V^-1 W^T to the
                                         all-reduced vector (two passes over its observations) and the ranks all-reduce ld doubles
                                         (needs no reduce-scatter; duplicate (camera, point) observations are part of the implicit
                                         product: their cross terms are then NOT added to the diagonal blocks, which stay a preconditioner).
                                         3 (ABI v5; implied by a problem created with SFMBA_CREATE_ROW_SHARDED): block ROWS of S~ per rank.
                                         Every rank holds the whole problem; it eliminates its own range of points, the per-point table is
                                         ALL-GATHERED (88 bytes per point in F32J: 44 MB at 500k points), the camera-diagonal sums of the
                                         rank's share of the camera-major list go through exchange (A) as before, and the pair pass then
                                         forms the blocks of the rank's OWN block rows from ALL their pairs -- at one-GPU efficiency, no
                                         partial block ever crosses a rank, no exchange (B), no unpack.  The CG runs on the owned rows:
                                         per iteration one product launch, ONE all-reduce of ld + 16 (cameras / 4 + 1) doubles (the partial
                                         product and its partial dot products) and one update launch, all multi-workgroup. */
} sfmba_options;

/* Flags of sfmba_problem_create_ex (ABI v4; were environment variables read at create time). */
enum { SFMBA_CREATE_DETERMINISTIC = 1,    /* (SFMBA_DETERMINISTIC=1 in the environment forces it on) every workgroup owns its accumulator slot and multi-chunk
                                             sums are added in a fixed order, so results do not depend on the order fp64 atomics arrive in (bitwise reproducible
                                             run to run; ~30 % slower).  Sharded problems included (ABI v4), and the forms that apply the reduced matrix
                                             implicitly (shard_distributed_cg = 2, SFMBA_CREATE_NO_PAIR_LIST): the per-camera sums of a CG product are then
                                             written per chunk and added in chunk order (implicit_schur.hip).  The symmetric streaming CG (pcg_symmetric, ABI v6:
                                             its products arrive through atomics) is not used on such a handle. */
       SFMBA_CREATE_ROW_SHARDED = 2,      /* ABI v5, sfmba_problem_create_ex only: EVERY rank passes the WHOLE problem (all observations) with its rank / world;
                                             cam_active may be NULL.  The rank owns the points of a contiguous range of point slots (ceil(n / world) each), a
                                             contiguous share of the camera-major list and a balanced range of block rows of the reduced matrix
                                             (sfmba_options.shard_distributed_cg = 3).  Solved with sfmba_problem_solve_sharded (needs sfmba_problem_set_allgather
                                             when world > 1); always through the CG (SFMBA_LINEAR_CHOLESKY is treated as SFMBA_LINEAR_AUTO: CG to 1e-12).  After
                                             a solve every rank holds ALL parameters (the final points are all-gathered): sfmba_problem_get_params returns the
                                             whole solution on every rank. */
       SFMBA_CREATE_NO_PAIR_LIST = 4 };   /* ABI v5: do not build the list of observation pairs (4 bytes per pair of observations of one point: O(sum of squared
                                             track lengths) memory, and 2^31 pairs at most) and never form the reduced camera matrix: sfmba_problem_solve then runs the
                                             two-level CG with the matrix applied IMPLICITLY from the observations -- per CG iteration two passes over them, memory
                                             O(observations) whatever the track lengths.  A problem with 2^31 or more pairs (100 cameras that all see 440 000 points:
                                             the reference adds a residual block per (view, point) with no bound on the track length, BA.cpp:142-166) takes this path
                                             by itself -- no problem the reference's solver accepts is refused for its size.  Every linear_solver setting is served by
                                             that CG: SFMBA_LINEAR_PCG at pcg_tolerance, SFMBA_LINEAR_AUTO / _CHOLESKY at a relative residual of 1e-12 (the DENSE_SCHUR
                                             result to ~1e-10, not bit for bit); max_seconds is checked once per LM iteration; sfmba_problem_build_reduced and
                                             sfmba_problem_append are refused.  A resident handle that GROWS past 2^31 pairs in sfmba_problem_append takes this path
                                             from that append on (the append itself succeeds; the NEXT one is refused with SFMBA_ERR_INVALID_ARG "cannot grow in
                                             place" -- the drop-in shim then rebuilds).  Slower than the formed matrix wherever that fits (~8x per CG iteration at
                                             BASELINE config 5). */

typedef struct sfmba_summary {
    int    termination;               /* SFMBA_CONVERGENCE / NO_CONVERGENCE / FAILURE */
    int    iterations;                /* LM iterations taken (successful + unsuccessful) */
    int    successful_steps;
    int    unsuccessful_steps;
    int    residual_evals;            /* cost-only evaluations */
    int    jacobian_evals;            /* linearisations (residual + Jacobian) */
    int    linear_iters;              /* total PCG iterations (0 for Cholesky) */
    double initial_cost;
    double final_cost;
    double seconds;                   /* solve wall time, excludes H2D/D2H and structure build */
    double setup_seconds;             /* structure build + H2D (sfmba_solve only) */
    char   message[128];
    int    cholesky_fallbacks;        /* ABI v4: LM iterations of an AUTO solve whose CG did not reach 1e-12 and were factorised instead */
} sfmba_summary;

/* One row per LM iteration (row 0 = the initial evaluation), mirrors ceres::IterationSummary. */
typedef struct sfmba_iteration {
    int    iteration;
    int    step_is_valid;
    int    step_is_successful;
    int    linear_iters;
    double cost;
    double cost_change;
    double gradient_max_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;
} sfmba_iteration;

typedef struct sfmba_problem sfmba_problem;   /* opaque, device-resident problem */

SFMBA_API void        sfmba_options_default(sfmba_options* opt);
SFMBA_API int         sfmba_abi_version(void);
SFMBA_API const char* sfmba_last_error(void);
/* Number of visible HIP devices (0 if none / runtime missing). */
SFMBA_API int         sfmba_device_count(void);
/* Device memory of destroyed problems is kept in a bounded cache (<= 8 GB) and handed to the next problem, so that the
 * reference's call pattern -- adjustBundle() re-creating the problem after every added view, SfM.cpp:464-466 -- performs
 * no hipMalloc/hipFree in steady state.  This returns the cached memory to HIP; the number of bytes released. */
SFMBA_API long long   sfmba_release_cache(void);
/* ABI v6.  What the FIRST call of a process pays once -- the HIP context, the first pinned allocation, the first device chunks: 138 ms at BASELINE
 * config 3 against 2.6 - 3 ms for every later adjustBundle() -- can be paid at start-up instead: creates the context on `device`, one stream + pinned
 * block and device chunks for a problem of about `expected_obs` observations (0: the fixed-size pieces only) and leaves them in the cache the first
 * sfmba_problem_create* draws from.  Optional and idempotent; rc as everywhere (SFMBA_ERR_NO_DEVICE without a GPU). */
SFMBA_API int         sfmba_device_warmup(int device, int64_t expected_obs);

/*
 * One-shot solve == the ceres::Problem build + ceres::Solve of BA.cpp:109-179.
 * Parameters are updated in place for CONVERGENCE and NO_CONVERGENCE and left untouched for FAILURE (as Ceres
 * does); the shim applies the reference's "discard unless CONVERGENCE" rule (BA.cpp:182-185).
 * trace may be NULL; at most trace_cap rows are written and *trace_len receives the number written (with trace == NULL:
 * the number of rows the solve produced).
 */
SFMBA_API int sfmba_solve(int n_cam, double* cam6, int n_pt, double* pt3,
                int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                double* focal, const sfmba_options* opt, sfmba_summary* summary,
                sfmba_iteration* trace, int trace_cap, int* trace_len);

/*
 * Resident API: the problem (observation lists, structure, parameters) lives in HBM
 * across calls -- used by bench.py (inputs resident before the timed region) and by
 * the incremental caller (SfM.cpp:464-466 re-runs BA after every added view).
 */
SFMBA_API int  sfmba_problem_create(int device, int precision,
                          int n_cam, const double* cam6, int n_pt, const double* pt3,
                          int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                          double focal, sfmba_problem** out);
/* The same with create flags (SFMBA_CREATE_*); cam_active != NULL makes it a sharded problem exactly like
 * sfmba_problem_create_sharded (rank / world then matter; pass 0 / 1 otherwise). */
SFMBA_API int  sfmba_problem_create_ex(int device, int precision, int flags,
                          int n_cam, const double* cam6, const unsigned char* cam_active, int n_pt, const double* pt3,
                          int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                          double focal, int rank, int world, sfmba_problem** out);
/*
 * Grows a resident problem in place -- the incremental caller re-runs BA after every added view (SfM.cpp:464-466) with a
 * cloud that only ever GROWS (new points, new views of existing points, SfM.cpp:530-629): n_cam >= the previous n_cam and
 * n_pt >= the previous n_pt (new cameras / points at the END of the arrays), n_obs_new NEW observations (of old or new cameras
 * and points; indices into the full arrays).  The observations given before stay.  The parameters of ALL cameras and points
 * and the focal are replaced from cam6 / pt3 / focal (full arrays, as at create time): the caller's containers hold the
 * float-rounded result of the previous solve plus the new entries (BA.cpp:187-221).  The observation list never leaves the
 * device: the new observations are uploaded, merged into the point-major order by a device sort, and the dependent lists
 * (camera-major index, camera-pair lists, launch descriptors) are rebuilt on the device.  The problem solved afterwards is
 * the one sfmba_problem_create would build from the concatenated observation list (old observations first, then the new ones);
 * results agree with that up to floating-point reordering, NOT bit for bit: newly observed cameras / points take the next free
 * slot in order of first observation (create assigns slots in ascending caller index), so the reduced system is a symmetric
 * permutation of create's and sums are taken in a different order.
 * Failure contract: if the call fails after it has begun replacing the device structure (allocation or HIP error), the problem
 * is POISONED -- every later entry point on it returns SFMBA_ERR_INVALID_ARG ("poisoned") without touching the device and the
 * only valid call is sfmba_problem_destroy.  Argument errors (bad sizes, indices out of range) are detected first and leave the
 * problem as it was.
 */
SFMBA_API int  sfmba_problem_append(sfmba_problem* p, int n_cam, const double* cam6, int n_pt, const double* pt3,
                          int64_t n_obs_new, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy, double focal);
/* Restore the parameters given at create (or last append) time.  Nothing is enqueued by the call itself: the first kernel of the next
 * sfmba_problem_solve copies them on the device, and every other entry point that looks at the parameters does so first. */
SFMBA_API int  sfmba_problem_reset(sfmba_problem* p);
/* Overwrite the current parameters from host arrays (full-size arrays, as at create). */
SFMBA_API int  sfmba_problem_set_params(sfmba_problem* p, const double* cam6, const double* pt3, double focal);
SFMBA_API int  sfmba_problem_solve(sfmba_problem* p, const sfmba_options* opt, sfmba_summary* summary,
                         sfmba_iteration* trace, int trace_cap, int* trace_len);
SFMBA_API int  sfmba_problem_get_params(sfmba_problem* p, double* cam6, double* pt3, double* focal);
SFMBA_API void sfmba_problem_destroy(sfmba_problem* p);
/* The HIP stream all kernels of this problem are launched on (hipStream_t as void*). */
SFMBA_API void* sfmba_problem_stream(sfmba_problem* p);
/* Dimension of the reduced camera system: 6 * (#cameras with observations) + 1. */
SFMBA_API int  sfmba_problem_reduced_dim(const sfmba_problem* p);

/*
 * Per-kernel timing with HIP events recorded on the problem's own stream around every launch
 * (bench.py's `roofline` object).  set_profiling resets the counters.
 */
typedef struct sfmba_kernel_time {
    char    name[32];
    double  total_us;
    int64_t launches;
} sfmba_kernel_time;
SFMBA_API int sfmba_problem_set_profiling(sfmba_problem* p, int enable);
SFMBA_API int sfmba_problem_get_profile(sfmba_problem* p, sfmba_kernel_time* out, int cap, int* n);

/*
 * Kernel-level entry points (parity tests call these through the C ABI).
 *   residuals_out : [2*n_obs] in the caller's observation order
 *   cost_out      : 1/2 sum r^2
 */
SFMBA_API int sfmba_problem_eval_residuals(sfmba_problem* p, double* residuals_out, double* cost_out);
/*
 * Jacobian blocks at the current parameters, UNSCALED, caller's observation order:
 *   jc [n_obs][2][6], jp [n_obs][2][3], jf [n_obs][2].  Any pointer may be NULL.
 */
SFMBA_API int sfmba_problem_eval_jacobian(sfmba_problem* p, double* jc, double* jp, double* jf);
/*
 * Damped, Jacobi-scaled reduced camera system at the current parameters for trust-region
 * radius `radius`:  S [dim*dim] row-major (symmetric, both triangles filled), rhs [dim],
 * scale [dim] = Jacobi column scaling of the reduced unknowns (cameras in ascending active
 * order, focal last).  jacobi_scaling follows opt (NULL = defaults).
 */
SFMBA_API int sfmba_problem_build_reduced(sfmba_problem* p, const sfmba_options* opt, double radius,
                                double* S, double* rhs, double* scale);
/* Dense SPD solve on the device (the reduced-system solver in isolation): A [n*n] row-major
 * symmetric, b [n] -> x [n].  method = SFMBA_LINEAR_*.  Returns SFMBA_OK and *info = 0 on success,
 * *info = k > 0 if the leading minor of order k is not positive definite. */
SFMBA_API int sfmba_dense_spd_solve(int device, int n, const double* A, const double* b, double* x,
                          int method, double pcg_tol, int pcg_max_iters, int* info, int* iters);

/*
 * Sharded API (multi-GPU): every rank holds the observations of a disjoint set of points
 * and a replica of all cameras + the focal.  Points are independent given the cameras (the Schur
 * structure), so each rank eliminates its own points and the ONLY exchange per LM iteration is the
 * sum of the partial reduced camera systems:
 *
 *   begin -> [all-reduce SUM of setup_buf] -> setup_finish
 *   repeat: partial_build -> [all-reduce SUM of reduce_buf] -> solve_update
 *                         -> [all-reduce SUM of scalars_buf] -> finish(&done)
 *   end
 *
 * All buffers are DEVICE pointers owned by the problem (wrap them as torch tensors for
 * torch.distributed / RCCL); every phase is enqueued on sfmba_problem_stream().  Every rank solves the
 * reduced system redundantly and takes bit-identical accept/reject decisions (no broadcast).
 *   reduce_buf  = [ upper triangle of S, packed row after row (ld (ld + 1) / 2) | rhs (ld) | udiag (ld) | bc (ld) | scalars (SFMBA_SHARD_SCALARS) ]
 *   setup_buf   = [ udiag (ld) | bc (ld) | scalars ] of the problem's own system buffer (column norms for the Jacobi scaling, ||x||^2)
 *   scalars_buf = the last SFMBA_SHARD_SCALARS doubles (trial cost, model change, step norms; one
 *                 per-rank slot each for the gradient max-norm, gathered through the SUM)
 */
#define SFMBA_SHARD_SCALARS 80
SFMBA_API int     sfmba_problem_create_sharded(int device, int precision,
                          int n_cam, const double* cam6, const unsigned char* cam_active /* [n_cam] globally observed cameras */,
                          int n_pt, const double* pt3,
                          int64_t n_obs, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                          double focal, int rank, int world, sfmba_problem** out);
SFMBA_API int     sfmba_shard_begin(sfmba_problem* p, const sfmba_options* opt);
SFMBA_API int     sfmba_shard_setup_finish(sfmba_problem* p);
SFMBA_API int64_t sfmba_shard_reduce_len(const sfmba_problem* p);       /* doubles in reduce_buf */
SFMBA_API void*   sfmba_shard_reduce_buf(sfmba_problem* p);
SFMBA_API int64_t sfmba_shard_setup_len(const sfmba_problem* p);
SFMBA_API void*   sfmba_shard_setup_buf(sfmba_problem* p);
SFMBA_API void*   sfmba_shard_scalars_buf(sfmba_problem* p);            /* SFMBA_SHARD_SCALARS doubles */
SFMBA_API int     sfmba_shard_partial_build(sfmba_problem* p);          /* linearise own points: partial S / rhs / scalars */
SFMBA_API int     sfmba_shard_solve_update(sfmba_problem* p);           /* after all-reduce #1: solve, back-substitute, trial cost */
SFMBA_API int     sfmba_shard_finish(sfmba_problem* p, int* done);      /* after all-reduce #2: accept/reject, convergence */
SFMBA_API int     sfmba_shard_end(sfmba_problem* p, sfmba_summary* summary);

/*
 * The whole sharded LM loop in one call (the phase functions above remain for callers that drive the choreography themselves).
 * The all-reduces go through `allreduce(ctx, device_buf, n_doubles, hip_stream)` -- in-place SUM over the ranks, enqueued on
 * hip_stream, return 0 on success -- which may be NULL when world == 1.  Per LM iteration, exact solver: two (packed upper triangle
 * of S | rhs | diagonals | scalars, then 80 trial-step scalars).  CG solver: three -- (A) 6x6 diagonal blocks | camera-focal column
 * | rhs | diagonals | scalars, (B) the off-diagonal blocks of the block-Jacobi-PRECONDITIONED matrix (every rank transforms its
 * partial blocks with the factors that follow from (A): the transform is linear in S), (C) the 80 trial-step scalars.  Every rank
 * calls with the same options; what is decided from the reduced sums is then bit-identical on all of them.  sfmba_comm_* is the built-in one:
 * ncclAllReduce (RCCL, xGMI) bound with dlopen at first use; one communicator per rank, created from the 128-byte unique id
 * that rank 0 draws (sfmba_comm_unique_id) and the launcher distributes (bench.py: a torch.distributed broadcast).
 * The host meets the GPU once per LM iteration, at the control kernel's mailbox post; nothing is copied back inside the loop.
 * options.max_seconds is NOT applied in sharded solves (ranks would disagree on a wall clock); max_iters is.
 * Error behaviour is fail-stop, as with RCCL itself: every allocation the loop needs is made BEFORE the rank issues its first
 * collective, so a rank that cannot take part returns an error without having entered one; inside the loop only HIP / collective
 * errors remain.  A rank that returns an error leaves its peers blocked in their next collective: the caller must then tear the
 * job down -- sfmba_comm_abort (ncclCommAbort) on the built-in communicator makes the peers' pending collectives fail so that
 * they return SFMBA_ERR_HIP too.
 */
#define SFMBA_COMM_ID_BYTES 128
typedef struct sfmba_comm sfmba_comm;
typedef int (*sfmba_allreduce_fn)(void* ctx, void* device_buf, int64_t n_doubles, void* hip_stream);
SFMBA_API int  sfmba_comm_unique_id(unsigned char id[SFMBA_COMM_ID_BYTES]);
SFMBA_API int  sfmba_comm_create(const unsigned char id[SFMBA_COMM_ID_BYTES], int rank, int world, int device, sfmba_comm** out);
SFMBA_API void sfmba_comm_destroy(sfmba_comm* comm);
SFMBA_API int  sfmba_comm_size(const sfmba_comm* comm, int* world, int* rank);   /* ncclCommCount / ncclCommUserRank: what RCCL itself says the communicator is (bench.py's n_gpus) */
SFMBA_API int  sfmba_comm_abort(sfmba_comm* comm);      /* ncclCommAbort: call on the ranks that failed; the communicator is unusable afterwards */
SFMBA_API int  sfmba_comm_allreduce(void* comm /* sfmba_comm* */, void* device_buf, int64_t n_doubles, void* hip_stream);   /* an sfmba_allreduce_fn */
SFMBA_API int  sfmba_problem_solve_sharded(sfmba_problem* p, const sfmba_options* opt, sfmba_allreduce_fn allreduce, void* ctx,
                                 sfmba_summary* summary);
/* Optional single-precision all-reduce (same ctx as the fp64 one).  Where the CG stores the preconditioned matrix in fp32 anyway
 * (SFMBA_PRECISION_F32J and more than 1280 reduced unknowns: the streaming CG path) exchange (B) -- by far the largest: 18 Nc (Nc - 1)
 * values, 144 MB in fp64 at 1000 cameras -- is then summed and stored in fp32: half the bytes over xGMI, and the summed buffer is the
 * CG's matrix without a narrowing pass.  Without it (or with options.shard_f32_exchange = -1) every exchange stays fp64. */
typedef int (*sfmba_allreduce_f32_fn)(void* ctx, void* device_buf, int64_t n_floats, void* hip_stream);
SFMBA_API int  sfmba_comm_allreduce_f32(void* comm /* sfmba_comm* */, void* device_buf, int64_t n_floats, void* hip_stream);   /* an sfmba_allreduce_f32_fn */
SFMBA_API int  sfmba_problem_set_allreduce_f32(sfmba_problem* p, sfmba_allreduce_f32_fn allreduce_f32);                        /* NULL: fp64 only */
/* Collectives of the DISTRIBUTED CG (options.shard_distributed_cg, include above): exchange (B) becomes a reduce-scatter of the
 * upper-triangle blocks of S~ in a layout of `world` equal chunks (contiguous ranges of block rows, padded): after the call, stream-ordered,
 * recv_buf (= send_buf + rank * n_values elements: in place) holds the SUM over the ranks of chunk `rank`.  is_f32 != 0: the elements are
 * floats.  Every CG iteration then all-reduces ONE vector of ld doubles through the sfmba_allreduce_fn given to sfmba_problem_solve_sharded
 * (eight of them, in one call, at the coarse-space setup).  Without a reduce-scatter callback the option is ignored when world > 1. */
typedef int (*sfmba_reduce_scatter_fn)(void* ctx, void* send_buf, void* recv_buf, int64_t n_values, int is_f32, void* hip_stream);
SFMBA_API int  sfmba_comm_reduce_scatter(void* comm /* sfmba_comm* */, void* send_buf, void* recv_buf, int64_t n_values, int is_f32, void* hip_stream);   /* an sfmba_reduce_scatter_fn: ncclReduceScatter */
SFMBA_API int  sfmba_problem_set_reduce_scatter(sfmba_problem* p, sfmba_reduce_scatter_fn reduce_scatter);     /* NULL: none */
/* All-gather of the ROW-SHARDED solve (SFMBA_CREATE_ROW_SHARDED): in place -- buf holds `world` slices of bytes_per_rank bytes, slice `rank` is this
 * rank's contribution; after the call, stream-ordered, every slice holds its owner's bytes.  Called twice per linearisation (the two halves of the
 * per-point table) and once at the end of a solve (the final points). */
typedef int (*sfmba_allgather_fn)(void* ctx, void* buf, int64_t bytes_per_rank, void* hip_stream);
SFMBA_API int  sfmba_comm_allgather(void* comm /* sfmba_comm* */, void* buf, int64_t bytes_per_rank, void* hip_stream);   /* an sfmba_allgather_fn: ncclAllGather */
SFMBA_API int  sfmba_problem_set_allgather(sfmba_problem* p, sfmba_allgather_fn allgather);     /* NULL: none */
/* what the last sfmba_problem_solve_sharded() exchanged per linearisation: out = { bytes of (A), bytes of (B), bytes of (C), flags: bit 0 = (B) was fp32, bit 1 = distributed CG (then (B) = the bytes of the whole
 * reduce-scatter buffer, of which a rank receives 1 / world, and every CG iteration adds 8 ld bytes of all-reduce),
 * bit 2 = implicit Schur CG, bit 3 = row-sharded (then (B) = the bytes of the per-point table a rank RECEIVES through the all-gather) } */
SFMBA_API int  sfmba_shard_last_exchange(const sfmba_problem* p, int64_t out[4]);

/*
 * The step in front of bundle adjustment: SfMStereoUtilities::triangulateViews
 * (SfMToyLib/SfMStereoUtilities.cpp:120-206) for n ALIGNED matches -- normalise with K (no distortion), DLT
 * triangulation (cv::triangulatePoints), de-homogenise, re-project into both views, keep[i] = both reprojection errors
 * <= max_reproj_px (the reference's MIN_REPROJECTION_ERROR = 10, :42).  left_xy / right_xy [n][2] pixels, K [9]
 * row-major, P_left / P_right [12] row-major [R|t]; outputs points3d [n][3], keep [n] and (optional) reproj_err [n][2].
 * Host pointers; the computation runs on `device`.
 */
SFMBA_API int sfmba_triangulate(int device, int64_t n, const float* left_xy, const float* right_xy, const float* K,
                                const float* P_left, const float* P_right, float max_reproj_px,
                                float* points3d, unsigned char* keep, float* reproj_err);

/*
 * The two association loops of the incremental pipeline, results identical to the reference's loops
 * entry for entry and in the same order.
 *
 * Shared encodings
 *   cloud views   CSR over the cloud points: view_ptr [n_pt + 1]; entries view_idx / feat_idx = the point's originatingViews
 *                 (SfMCommon.h:87) in ASCENDING view index (std::map iteration order)
 *   match matrix  SfM::mFeatureMatchMatrix (SfM.h:50) flattened: pair p = [pair_left[p]][pair_right[p]], its cv::DMatch list is
 *                 entries pair_ptr[p] .. pair_ptr[p+1] of query_idx / train_idx (/ distance), in list order.  Only entries with
 *                 left <= right are ever consulted by the reference (SfM.cpp:489-490, 555-558); others are ignored here too.
 *
 * sfmba_find_2d3d_matches == SfM::find2D3DMatches (SfMToyLib/SfM.cpp:471-528): for every view v with view_done[v] == 0 and
 * every cloud point, the first originating view (ascending) that has a match to v for the point's feature -- the FIRST such
 * match in list order, matches whose other index is negative skipped (SfM.cpp:508) -- yields one entry
 * (cloud point index, feature index in view v).  Output: out_ptr [n_views + 1] (done views: empty ranges), entries in cloud
 * order inside a view; the reference's points2D / points3D are features[v].points[out_feature] / cloud[out_point].p.
 * *total receives the number of entries; SFMBA_ERR_CAPACITY (out_ptr and *total valid, nothing else written) if cap < *total.
 */
SFMBA_API int sfmba_find_2d3d_matches(int device, int n_views, const unsigned char* view_done,
                int n_pt, const int64_t* view_ptr, const int32_t* view_idx, const int32_t* feat_idx,
                int n_pairs, const int32_t* pair_left, const int32_t* pair_right, const int64_t* pair_ptr,
                const int32_t* query_idx, const int32_t* train_idx,
                int64_t* out_ptr, int32_t* out_point, int32_t* out_feature, int64_t cap, int64_t* total);
/*
 * The O(n^2) part of SfM::mergeNewPointCloud (SfMToyLib/SfM.cpp:538-544): which points of the cloud are closer than max_dist
 * (MERGE_CLOUD_POINT_MIN_MATCH_DISTANCE, SfM.cpp:50) to new point k.  "The cloud" as of new point k is the existing points
 * followed by the new points 0 .. k-1 that were appended before it (SfM.cpp:596-600), so the candidates of k are indices j of
 * the sequence [existing 0 .. n_exist-1, new 0 .. k-1] (j >= n_exist means new point j - n_exist) with
 * cv::norm(seq[j] - new[k]) < max_dist in the reference's arithmetic (float difference, double norm), ASCENDING -- the order
 * the reference's scan meets them in.  cand_ptr [n_new + 1], cand_idx [cap].  The sequential, data-dependent remainder of the
 * function (feature-match confirmation, views added to existing points) is host code: host/SfMAssociation.cpp.
 */
SFMBA_API int sfmba_merge_candidates(int device, int n_exist, const float* exist_xyz, int n_new, const float* new_xyz, float max_dist,
                int64_t* cand_ptr, int32_t* cand_idx, int64_t cap, int64_t* total);

#ifdef __cplusplus
}
#endif
#endif /* SFMBA_H_ */
