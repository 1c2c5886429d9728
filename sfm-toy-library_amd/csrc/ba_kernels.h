// ba_kernels.h -- host-visible declarations of the bundle-adjustment kernels (ba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>

// observations of one camera handled by one workgroup of k_cam_diag (256 threads, SFMBA_CAM_CHUNK / 256 observations per lane); the host
// cuts the camera-major list accordingly.  One per lane: with four (1024-entry chunks, one reduction per four observations) the pass
// ran 46 instead of 40 us at BASELINE config 3 -- it is bound by its dependent loads and the number of waves in flight, not by the
// reduction's instructions
#ifndef SFMBA_CAM_CHUNK
#define SFMBA_CAM_CHUNK 256
#endif

// pairs per wave of k_schur_pairs: a 6x6 block with more pairs is cut into chunks (eight rounds of 64 pairs)
#ifndef SFMBA_PAIR_CHUNK
#define SFMBA_PAIR_CHUNK 512
#endif

namespace sfmba {

// Accumulator slots inside LMState::acc (zeroed by k_lm_control / hipMemsetAsync).
enum {
    ACC_TRIAL_COST = 0,   // sum r^2 at the trial point (x2 of the cost)
    ACC_MODEL = 1,        // model_cost_change
    ACC_STEP2 = 2,        // ||x - x_trial||^2
    ACC_XNEW2 = 3,        // ||x_trial||^2
    ACC_BAD_TRIAL = 4,    // != 0: non-finite residual at the trial point
    ACC_GMAX = 5,         // max |g| (bit pattern, atomicMax)
    ACC_BAD_LIN = 6,      // != 0: non-finite value in the linearisation
    ACC_LIN_COST = 7,     // sum r^2 at the linearisation point
    ACC_COUNT = 8,
    // focal-focal partial sums of the reduced system (same slotted buffer)
    ACC_SFF = 8, ACC_RHSF = 9, ACC_UDF = 10, ACC_BCF = 11,
    SLOT_W = 12,
    NSLOT = 64          // same-address global atomics serialise (~12 ns each): workgroups spread over 64 slots
};

// Device-resident Levenberg-Marquardt state (TrustRegionMinimizer + LevenbergMarquardtStrategy
// bookkeeping [Ceres-upstream]).  One instance per problem, in HBM.
struct LMState {
    int cur;                  // parameter buffer holding the accepted point (0/1)
    int iter;                 // LM iterations taken
    int termination;          // -1 running, else SFMBA_* termination type
    int message;              // message id (host maps to text)
    int consecutive_invalid;
    int successful, unsuccessful, residual_evals, jacobian_evals, linear_iters;
    int x_is_new;             // the current linearisation is at a freshly accepted point
    int lin_info;             // dense solver status of this iteration (0 ok)
    int last_step_successful;
    int mail_seq;             // number of k_lm_control launches since the solve started
    double cost, x_norm, radius, decrease_factor, gmax;
    double focal[2];
    double fscale;            // Jacobi scale of the focal column
    double acc[ACC_COUNT];
    // options (copied from sfmba_options at solve start)
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double max_radius, min_radius, min_relative_decrease, min_diag, max_diag;
    int max_consecutive_invalid;
    int retry;                // set by k_lm_control when it asked for more CG iterations (the LM iteration is not finished yet)
    double invalid_shrink;    // factor on the radius after an INVALID step: 1/2 (Ceres, StepIsInvalid); 1/8 with fp32 Jacobians (init_state)
};

enum {
    MSG_NONE = 0, MSG_GRADIENT_TOL, MSG_PARAMETER_TOL, MSG_FUNCTION_TOL, MSG_MIN_RADIUS,
    MSG_INVALID_STEPS, MSG_INITIAL_EVAL_FAILED, MSG_EVAL_FAILED, MSG_MAX_ITERS, MSG_MAX_TIME
};

struct TraceRow {   // same layout as sfmba_iteration
    int iteration, step_is_valid, step_is_successful, linear_iters;
    double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
};

// Static structure of a problem, all device pointers (built once on the host, see sfmba_api.cpp).
struct DeviceStructure {
    int ncam, npt, nobs;      // active cameras / points, observations
    int pt_base;              // first point slot of the passes that walk the points in slot order without pt_order (k_xnorm): 0, or the first own slot of a
                              // row-sharded rank (whose point passes see npt = its own points through pt_order)
    int d, ld;                // reduced dim 6*ncam+1, padded leading dimension (multiple of 64)
    const int* pt_ptr;        // [npt+1] point-major CSR
    const int* pt_order;      // [npt] point slots sorted by number of observations (lane-group point passes: the quads of a wave loop alike); null = slot order
    const int* obs_cam;       // [nobs] camera slot, point-major order (ascending inside a point)
    const void* obs_xy;       // [nobs] float2 or double2, point-major order
    const int* cam_ptr;       // [ncam+1] camera-major CSR
    const int* cam_obs;       // [nobs] point-major position q of each camera-major entry
    const int* cam_obs_pt;    // [nobs] point slot of that entry
    const void* cam_obs_xy;   // [nobs] float2 / double2: the observation coordinates in camera-major order (re-evaluating camera pass)
    const int* obs_pt;        // [nobs] point slot, point-major order
    int nchunk;
    const int4* chunks;       // [nchunk] {camera slot, begin, end (camera-major entries), 0}: SFMBA_CAM_CHUNK entries each (k_cam_diag)
    const int* chunk_order;   // [nchunk] launch order of the chunks: workgroup b takes chunk chunk_order[b].  Chunks are listed camera by camera (a camera's
                              //       entries ascend in point slot); launched in that order, neighbouring workgroups walk ONE camera's list across the whole
                              //       per-point table and every gather misses L2 once the table outgrows it (BASELINE config 5: 120 bytes of HBM-side traffic per
                              //       observation).  Ordered by the chunk's relative position inside its camera's list instead, the workgroups in flight
                              //       gather from the same window of the table whatever their camera.  Never null.
    int nchunk_coarse;
    const int4* chunks_coarse; // the same list cut every 1024 entries (column-norm pass: a block loops over its chunk); deterministic mode: one chunk per camera
    const int* coarse_order;   // [nchunk_coarse] the same for the coarse chunks
    const int* cam_chunk_ptr;  // [ncam+1] first k_cam_diag chunk of every camera (deterministic mode)
    // camera-pair lists of the reduced-system pass: block b = (ja <= jb); pairs sorted by block
    int nblock;
    const int2* blk_cams;     // [nblock] {ja, jb}
    const int* blk_ptr;       // [nblock+1]
    const int* pair_pt;       // [npair] the point slot of every pair of observations (qa < qb) of one point, grouped by block, ascending point inside a
                              //         block (the pair pass needs nothing else per pair: the cameras follow from the block)
    int pair_lpb;             // lanes per 6x6 block in the pair pass: 64 (k_schur_pairs) or 16 (k_schur_pairs_sub_f), from the mean pairs per block
    int npairwg;
    const int2* pwg_blocks;   // [npairwg] {first block, #blocks <= 4} per workgroup of the pair pass (XCD-grouped rows)
    int pwg_group;            // blocks per workgroup entry (SFMBA_PAIR_WAVES, or 64 / pair_lpb)
    const int4* pwg_desc;     // [npairwg * pwg_group] {block or -1, row camera ja, first pair, last pair + 1}: everything a wave (or lane
                              //       group) needs about its block in ONE load; jb follows from the block index
    const int2* pwg_chunk;    // wave-per-block pass: [npairwg] {row of pair_partial, chunks of the block}: heavy blocks are cut into chunks of
                              //       SFMBA_PAIR_CHUNK pairs, one wave each (consecutive slots); null for the sixteen-lane pass
    int nmulti;               // blocks of more than one chunk ...
    const int* multi_slots;   // ... [nmulti] and the first slot of each (k_schur_combine adds their partial sums)
    int ndupwg;
    const int2* dup_blocks;   // [ndupwg] like pwg_blocks, but only diagonal blocks that have pairs (same camera seeing a point twice)
};

struct DeviceBuffers {
    double* cam[2];           // [ncam][6]
    double* pts[2];           // [npt][3]
    double* camtab[2];        // [ncam][CT_STRIDE]
    double* steptab;          // [ncam][ST_STRIDE]
    float* pu32;              // F32J, every LM loop (rank 0 of a sharded solve adds gradient . step): [ncam][20] fp32 record {R, t of the linearisation point (12), Q dw, dt, first-order flag (8)} written by
                              //         k_cam_update for the FIRST sweep of k_point_update (five 16-byte gathers per observation instead of ten); else null
    double* cscale;           // [6*ncam] Jacobi scale of the camera columns
    double* pscale;           // [npt][3]
    void* PA;                 // [npt] PtRecA<T>: X, L^-1 diag(s_p) -- the per-point table every pass re-evaluates the observations from (sfmba_device.h)
    void* PB;                 // [npt] PtRecB<T>: t = L^-1 b_p, y_f = L^-1 E_f
    double* pt_t;             // [npt][3] L^-1 b_p
    double* pt_M;             // [npt][6] diag(s_p) L^-T (upper triangle, row-major): dX = M z maps the reduced point right-hand side to the unscaled step
    double* S;                // [ld*ld] reduced system: upper triangle of the row-major matrix
    double* rhs;              // [ld]  (overwritten by the solution)
    double* udiag;            // [ld]  diag(J~^T J~) of the reduced unknowns, undamped
    double* bc;               // [ld]  scaled gradient of the reduced unknowns
    double* slots;            // [nslot][SLOT_W] slotted accumulators (cost, norms, gradient max, focal-focal sums)
    int nslot;                // NSLOT (64) normally; deterministic mode: >= the largest grid, so every workgroup of a launch owns its slot
                              // and the sums no longer depend on the order the atomics arrive in
    double* cd_part;          // deterministic mode: [nchunk][48] per-chunk sums of k_cam_diag, added in chunk order by k_finalize; else null
    LMState* st;
    TraceRow* trace;
    int trace_cap;
    int* lin_info;            // dense-solver status word (device), consumed and cleared by k_lm_control
    int* fin_counter;         // arrival counter of k_finalize (last block runs the post-linearisation logic)
    // block-Jacobi PCG: the solution is x~ with z = Lb^-T x~ (dense_solver.hip); k_cam_update applies Lb^-T itself
    const double* pcg_vec;    // x~ buffers (two, selected by pcg_flags[2]); nullptr when the Cholesky path wrote z to rhs
    const double* pcg_linv;
    const int* pcg_flags;
    LMState* st_mirror;       // host-mapped copy of the LM state, refreshed by k_lm_control before every mailbox post (the host then needs no
                              // blocking copy at the end of a solve); the trace rows may live in host-mapped memory too
    const int* cg_gate;       // non-null: k_cam_update / k_point_update / k_lm_control do nothing unless cg_gate[0] (CG done) is set or cg_force
    int cg_force;             //           (the host enqueues them behind a CG batch of guessed length without waiting for it)
    double* pcg_F;            // [d][ld] preconditioned reduced matrix S~ written directly by k_schur_pairs (PCG mode)
    float* pcg_F32;           // the same in fp32 instead (streaming CG path, d > 1280: the matvec is HBM-bound); else null
    int pcg_upper_only;       // 1: the CG reads ONE triangle of S~ (dense_solver.hip, symmetric streaming path): store_block_entry writes the upper block only
    double* pcg_zero;         // ... and its coarse set-up ADDS S~ W~ into this buffer (atomics): k_finalize(pcg = 1) zeroes pcg_zero_n doubles of it; else null
    int pcg_zero_n;
    double* pcg_bt;           // [ld]    Lb^-1 rhs
    double* pcg_binv;         // [ncam*36 + 1] Linv of the diagonal blocks, written by k_finalize (PCG mode)
    double* pair_partial;     // [chunks of multi-chunk blocks][36] their partial sums (factored coordinates); rows: pwg_chunk[].x
    double* pair_G;           // [ncam*36] per-camera factor the factored pair pass applies from both sides of a block (row-major 6 x 6):
                              // Linv D E^T (PCG: k_finalize) or D E^T (exact solver: k_pair_factors), E = diag(R K', I) -- sfmba_device.h
    double* pcg_W;            // [8][ld] gauge vectors in the transformed unknowns (coarse space of the two-level CG preconditioner,
                              //         dense_solver.hip), written by k_finalize (PCG mode); null = not wanted
    int* lm_mailbox;          // host-mapped {seq, termination, message, iter}: polled by the host instead of a D2H copy + sync
    double shared_weight;     // 1 normally; 0 on ranks > 0 of a sharded solve (replicated cameras/focal counted once)
    float* shard_blocks32;    // ... the same in fp32 (exchange (B) in single precision: the streaming CG path stores S~ in fp32 anyway)
    double* shard_blocks;     // sharded CG path: the pair pass (MODE 1) stores the off-diagonal blocks of S~ here (all-reduce layout) instead of pcg_F
    const double* shard_scal; // sharded solve: k_lm_control takes the trial sums from this all-reduced scalar block instead of the slots
    const int* shard_row_shift; // distributed CG: blocks to ADD to a block's list position, per block row (reduce-scatter layout: dist_cg.h); null = none
};

template <typename T> void launch_cam_setup(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int which);
void launch_xnorm(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
template <typename T> void launch_colnorm(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi_scaling, bool clear_udiag = true,
                                         bool points = true, bool finish_xnorm = false, bool with_xnorm = false);
void launch_begin(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, const LMState& st, const double* cam_src = nullptr,
                  const double* pts_src = nullptr);
template <typename T> void launch_point_build(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db,
                                             int ps_mode = 0 /* 0: point scales from db.pscale; 1 / 2: form them here (Jacobi / unit) */);
// mode 0: off-diagonal blocks of S (upper triangle);  mode 1: the same blocks written straight into S~ = Lb^-1 S Lb^-T
// (both triangles) + the per-camera glue of the block-Jacobi transform;  mode 2: diagonal blocks with duplicate pairs.
// Both reduced-system passes re-evaluate every observation from the camera row and the per-point table (nothing is stored per observation).
template <typename T> void launch_schur_pairs(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int mode);
template <typename T> void launch_cam_diag(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
void launch_finalize(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int pcg);
void launch_cd_fold(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);   // deterministic + sharded: chunk sums into the partial system (before the exchange)
void launch_gauge(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);   // gauge vectors from db.pcg_binv (see k_gauge)
void launch_cam_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
template <typename T> void launch_point_update(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
void launch_control(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
void launch_iter0(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
template <typename T> void launch_eval_residuals(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db,
                                                 const int* perm, double* res_out, double* cost_out);
template <typename T> void launch_eval_jacobian(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db,
                                                const int* obs_pt, const int* perm, double* jc, double* jp, double* jf);
// Implicit Schur product of the sharded solve (implicit_schur.hip): out = this rank's part of S~ p~ from its own
// points (+ on rank 0 what every rank knows: identity diagonal blocks, focal row / column); summed over the ranks it is S~ p~.
struct ImplicitProduct {
    DeviceStructure ds;
    DeviceBuffers db;
    double* dtab = nullptr;            // [ncam][8] direction table (two component quads per camera)
    double* spt = nullptr;             // [npt][3] per-point sums
    double* acc = nullptr;             // [ncam][6] per-camera sums
    double* part = nullptr;            // deterministic handles: [nchunk][6] per-chunk sums, added in chunk order by k_imp_out; else null (atomics into acc)
    const double* focal_row = nullptr; const float* focal_row32 = nullptr;     // row d-1 of the CG's matrix (left by the glue)
    int rank = 0;
    bool f32 = false;                  // precision of the Jacobian blocks
};
void launch_implicit_product(hipStream_t s, const ImplicitProduct& ip, const double* p_tilde, double* out, const int* flags);
void launch_pcg_glue(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db);
// sharded mode: move slotted accumulators to / from the all-reduce scalar block
void launch_shard_pack(hipStream_t s, const DeviceBuffers& db, double* scal, int phase, int rank);
void launch_shard_tri(hipStream_t s, double* sys, double* packed, int ld, long long tail, bool unpack);
void launch_shard_unpack(hipStream_t s, const DeviceBuffers& db, const double* scal, int phase, int world);
// two-phase all-reduce of the sharded CG path (ba_kernels.hip): (A) diagonal blocks + vectors + scalars, (B) off-diagonal blocks of S~
long long shard_diag_len(const DeviceStructure& ds);
void launch_shard_diag(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, double* buf, bool unpack, int rank, int world);
long long shard_offdiag_len(const DeviceStructure& ds);
void launch_shard_offdiag(hipStream_t s, const DeviceStructure& ds, double* F, double* buf, bool unpack);
void launch_narrow_matrix(hipStream_t s, const double* src, float* dst, long long n);
void launch_shard_offdiag_f32(hipStream_t s, const DeviceStructure& ds, float* F32, const float* buf);      // unpack of the fp32 exchange
void launch_clear_slots(hipStream_t s, const DeviceBuffers& db);
void launch_shard_xnorm_finish(hipStream_t s, const DeviceBuffers& db);
void launch_colnorm_points_only(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi, int precision_f32);
void launch_colnorm_cams_only(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi, int precision_f32, bool clear_udiag = true);
void launch_colnorm_finish(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, int jacobi);
void launch_mirror_scale(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, double* S_full, double* scale_out);


// structure_build.hip: the problem structure is built on the device
class DeviceArena;
struct PointMajor {            // point-major observation list (build_point_major)
    int* obs_pt = nullptr;     // [2 n] point slot per sorted position, then the caller index of that observation
    int* obs_cam = nullptr;    // [n]
    void* obs_xy = nullptr;    // [n] float2 / double2
    int* pt_ptr = nullptr;     // [npt + 1]
    long long* pair_off = nullptr;   // [npt + 1] pairs of the points before i
};
// All three only enqueue (temporaries from `scratch`, which the caller keeps until the stream has drained)
int build_point_major(hipStream_t s, DeviceArena* arena, DeviceArena* scratch, int n, int npt, int ncam, int xy_bytes, const int* u_pt, const int* u_cam,
                      const int* u_perm, const void* u_xy, PointMajor* out);
// camera-pair lists of the Schur pass (d_pair_off: npt + 1 prefix counts on the device; npair: their total, known to the host)
int build_pair_lists(hipStream_t s, DeviceArena* arena, DeviceArena* scratch, int npt, int nobs, int ncam, int nblock, const int* d_pt_ptr, const int* d_obs_pt,
                     const int* d_obs_cam, const long long* d_pair_off, long long npair, int** d_blk_ptr, int** d_pair_pt);
int build_camera_major(hipStream_t s, DeviceArena* arena, DeviceArena* scratch, int nobs, int ncam, const int* d_obs_cam, const int* d_obs_pt,
                       int** d_cam_obs, int** d_cam_obs_pt, int** d_cam_ptr);
int build_camera_major_xy(hipStream_t s, DeviceArena* arena, int nobs, int xy_bytes, const int* d_cam_obs, const void* d_obs_xy, void** d_cam_obs_xy);
// unsorted observation arrays of a build: the old ones (device) followed by the new ones (one uploaded buffer)
struct StageObs {
    int n_old = 0, n_new = 0;
    const int *old_pt = nullptr, *old_cam = nullptr, *old_perm = nullptr; const void* old_xy = nullptr;
    const int *new_pt = nullptr, *new_cam = nullptr, *new_perm = nullptr; const void* new_xy = nullptr;
    int *u_pt = nullptr, *u_cam = nullptr, *u_perm = nullptr; void* u_xy = nullptr;
};
void launch_stage_obs(hipStream_t s, const StageObs& so, int xy_bytes);
// pair-pass descriptors and the list of diagonal blocks with pairs, from the block CSR on the device
void launch_pair_desc(hipStream_t s, int nwg, int group, const int2* pwg_blocks, const int2* blk_cams, const int* blk_ptr, const int* perm, int4* desc);
void launch_row_order(hipStream_t s, int ncam, int lpb, const int* blk_ptr, int* perm);      // blocks of every block row grouped by rounds of `lpb` pairs
// wave-per-block pair pass: one descriptor per chunk of SFMBA_PAIR_CHUNK pairs (structure_build.hip); counters: two zeroed ints; the slot
// total, the number of multi-chunk blocks and the number of non-empty off-diagonal blocks go to report[4], report[5], report[1]
int build_pair_chunks(hipStream_t s, DeviceArena* scratch, int nwg, int chunk, const int2* pwg_blocks, const int2* blk_cams, const int* blk_ptr,
                      int4* desc, int2* info, int* multi, int* counters, int* report);
void launch_block_fill(hipStream_t s, int nblock, int ncam, const int2* blk_cams, const int* blk_ptr, int* counters, int* report);
void launch_block_mask(hipStream_t s, int ncam, const int* blk_ptr, unsigned* mask);       // per camera: the cameras it shares a non-empty block with (bit mask)
void launch_dup_blocks(hipStream_t s, int ncam, const int* blk_ptr, const long long* pair_total, int2* dup, int* report);

// triangulate.hip: two-view DLT triangulation + reprojection filter (SfMStereoUtilities::triangulateViews), device pointers
void launch_triangulate(hipStream_t s, long long n, const float* d_left, const float* d_right, const float K[9], const float Pl[12],
                        const float Pr[12], float max_err, float* d_points3d, unsigned char* d_keep, float* d_err);

}  // namespace sfmba
