// dense_solver.h -- dense SPD solve of the reduced camera system on the device.
#pragma once
#include <hip/hip_runtime.h>
#include "device_arena.h"
#include "profiler.h"
#include <vector>

namespace sfmba {

constexpr int CHOL_NB = 64;

// Workspace of the reduced-system solver (allocated once per problem).
struct DenseSolver {
    int ld = 0;               // padded dimension (multiple of CHOL_NB, > d)
    int d = 0;                // true dimension
    double* minv = nullptr;   // [ld/NB][NB*NB] inverse-transposed diagonal blocks (L_kk^-T)
    double* y = nullptr;      // [ld] work vector of the back substitution
    // PCG
    double* Sfull = nullptr;  // [ld*ld] full symmetric copy (PCG only, allocated lazily)
    float* Sfull32 = nullptr; // fp32 copy for the streaming path (allocated by dense_pcg_want_f32)
    // launch parameters of the running CG solve (dense_pcg_solve ... dense_pcg_more)
    struct CgRun { int nwg = 0, rows_per_wg = 0; size_t lds = 0; bool fast = false, f32 = false, coarse = false, ml = false, sg = false, sym = false; double tol2 = 0.0; int in = 1, launched = 0, max_iters = 0; int* info = nullptr; } run;
    bool use_f32 = false;     // set by the caller per solve: the preconditioned matrix of THIS solve lives in Sfull32
    bool symmetric = false;   // set by the caller per solve: take the symmetric streaming path (k_sy_vec + k_sy_prod: the UPPER triangle of S~ read once per
                              // iteration) where it applies -- d > 1280, no segmented coarse space
    double* q3 = nullptr;     // [2][ld] the products of the symmetric path by iteration parity (added into with atomics; zeroed by the vector kernel in front)
    double* sym_part = nullptr;   // [2][9][64 slots] its partial sums (p_r . q, (S~ W~)^T p_r), one 128-byte line per accumulator
    int4* sym_tiles = nullptr; int sym_ntiles = 0;   // its tiles of the upper triangle
    int4* sym_ctiles = nullptr; int sym_nctiles = 0; // ... and the taller ones of its coarse set-up (k_sy_coarse)
    double* AWt = nullptr;    // [8][ld] S~ W~ of the symmetric path, vector-major; added into with atomics: the caller's linearisation zeroes it
    double* sym_zero = nullptr; size_t sym_zero_n = 0;   // (= AWt) what the caller's linearisation has to zero
    double* vec = nullptr;    // [9*ld] x[2] r[2] p[2] q[2] btilde
    double* part = nullptr;   // [2][9][1024] per-workgroup partial sums of one iteration (p_r.q, W~^T q), by iteration parity
    // coarse space of the two-level preconditioner (dense_solver.hip): 8 gauge vectors in the transformed unknowns
    double* W = nullptr;      // [8][ld] W~, written by the linearisation (k_finalize); fp32-representable values
    double* AW = nullptr;     // [d][8]  S~ W~
    double* epart = nullptr;  // [72][1024] per-workgroup partials of E = W~^T S~ W~ and c_0 = W~^T b~
    double* coarse = nullptr; // [72] E^-1 (64) and c_0 (8)
    // segmented coarse space (dense_solver.hip, "Segmented coarse space"): 57 hat-restricted gauge vectors, workgroup = camera
    double* mlAW = nullptr;   // [d][64] S~ W~
    double* mlV = nullptr;    // [nc + 1][8][64] per-camera pieces of E
    double* mlU = nullptr;    // [nc + 1][8] per-camera pieces of c_0
    double* mlE = nullptr;    // [64][64] E = W~^T S~ W~
    double* mlEinv = nullptr; // [64][64] E^-1 (rows / columns of dropped vectors zero)
    double* mlC0 = nullptr;   // [64] c_0 = W~^T b~
    double* mlState = nullptr;// [2][3][64] c, mu, p_mu by iteration parity
    // ... on the streaming path (d > 1280): per-camera pieces of E, E, E^-1 (144 x 144), per-camera W~_k . r, |r|^2 partials, {r.z, p.q} by parity
    double *sgV = nullptr, *sgE = nullptr, *sgEinv = nullptr, *sgT = nullptr, *sgRR = nullptr, *sgState = nullptr;
    const unsigned* blk_mask = nullptr;   // [ncam][(ncam + 31) / 32] per camera: cameras with a non-empty block in common (set by the caller; null: dense product)
    double blk_fill = 1.0;    // non-empty off-diagonal blocks / all (set by the caller)
    int last_iters = 0;       // CG iterations of the previous solve
    std::vector<int> hist;    // CG iterations of the previous call per caller key (LM iteration index): sizes the first launch batch
    double* binv = nullptr;   // [ld*6] inverses of the 6x6 diagonal blocks (+1x1 focal)
    double* scal = nullptr;   // [8] rz, pq, bnorm2, rnorm2, ...
    int* flags = nullptr;     // [4] done, iters
    int* h_flags = nullptr;   // pinned host mirror
    volatile int* h_mailbox = nullptr;   // host-mapped {iterations, done}: polled instead of copy + synchronise
    int* d_mailbox = nullptr;
    DeviceArena* arena = nullptr;   // when set, the device arrays above live in (and are released with) this arena
    bool pinned_external = false;   // h_flags / h_mailbox are slices of the caller's pinned block
};

// pinned: optional 128 bytes of pinned, host-mapped memory for {h_flags[4], pad, h_mailbox[16]} (else allocated here)
int  dense_solver_create(DenseSolver* ws, int d, int ld, DeviceArena* arena = nullptr, char* pinned = nullptr);
void dense_solver_destroy(DenseSolver* ws);

// Padded leading dimension for a reduced system of dimension d (room for the augmented rhs row).
inline int dense_padded_dim(int d) { return ((d + 1 + CHOL_NB - 1) / CHOL_NB) * CHOL_NB; }

// Cholesky: S holds the UPPER triangle of the row-major matrix (== lower triangle, column-major),
// padded to ld with an identity diagonal.  rhs[0..d) is overwritten by the solution.
// *info_dev (device int) is set to k>0 if the leading minor k is not positive definite.
void dense_cholesky_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, int* info_dev, Profiler* prof = nullptr);

// Block-Jacobi PCG on the same storage.  Returns the number of iterations (host sync inside).
// pretransformed = true: ws->Sfull, the right-hand side at ws->vec + 8*ld and ws->binv were already written by the
// linearisation kernels (k_finalize / k_schur_pairs mode 1), so the block-Cholesky + transform launches are skipped.
// finish = false leaves the solution in transformed form (x~, see dense_solver.hip) for k_cam_update;
// hist_key >= 0 selects the history slot used to size the first batch of launches.
int dense_pcg_solve(hipStream_t s, DenseSolver* ws, double* S, double* rhs, double tol, int max_iters, int* info_dev,
                    Profiler* prof = nullptr, bool finish = true, int hist_key = -1, bool pretransformed = false, int anchor = 0,
                    bool no_wait = false, bool coarse = false, bool segments = false);
// segments = true (with coarse): the segmented coarse space where it applies (dense_pcg_segments_applicable: d = 6 nc + 1 <= 1280, nc >= 32)
bool dense_pcg_segments_applicable(const DenseSolver* ws);
// ... or its streaming-path form (1280 < d <= 8192): classical PCG in three launches per iteration, up to 20 hats (at most 1007 cameras)
bool dense_pcg_segments_streaming_applicable(const DenseSolver* ws);
// dense_pcg_transform: the block-Jacobi transform on its own (block factors -> ws->binv, S~ -> ws->Sfull / Sfull32, b~), for callers that
// need the factors before the solve (sharded path: gauge vectors are formed from them); follow with dense_pcg_solve(pretransformed = true).
int dense_pcg_transform(hipStream_t s, DenseSolver* ws, double* S, double* rhs, int* info_dev, Profiler* prof = nullptr);
// coarse = true: ws->W holds the 8 gauge vectors of this linearisation (written by k_finalize): two-level preconditioner
// anchor: 0 = relative residual |r| <= tol |b~|; 1 = first solve of an LM run (remembers |b~|); 2 = later solve of the
// same run: |r| <= tol * max(|b~|, |b~_first|), but never looser than max(tol, 1e-4) relative (see dense_solver.hip)
// no_wait variant: dense_pcg_solve(..., no_wait = true) enqueues the first batch of iterations and returns at once (the
// iteration count is then unknown to the host: consumers gate on ws->flags[0] on the device); dense_pcg_more enqueues up to
// n further iterations of the same solve (returns how many; 0 = max_iters reached); dense_pcg_note records the final count.
int dense_pcg_more(hipStream_t s, DenseSolver* ws, int n, Profiler* prof = nullptr);
void dense_pcg_note(DenseSolver* ws, int hist_key, int iters);
int dense_pcg_ensure_workspace(DenseSolver* ws);
// true where ws->symmetric selects the symmetric streaming path (d > 1280): the caller's pair pass may then write the upper triangle of S~ only
bool dense_pcg_symmetric_applicable(const DenseSolver* ws);
// fp32 storage of the preconditioned matrix for the streaming (d > 1280) path; returns the buffer or null if not applicable
float* dense_pcg_want_f32(DenseSolver* ws);
}  // namespace sfmba
