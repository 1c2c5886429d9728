// shard_exchange.hip -- pack / unpack kernels of the exchanges of a sharded solve (one problem, points sharded over the ranks; DESIGN.md
// section 6): the scalar blocks, exchange (A) (diagonal blocks + vectors), exchange (B) (off-diagonal blocks of the preconditioned matrix), and the
// packed upper triangle of the exact solver's single all-reduce.  The reduced camera system these carry is what ceres::Solve forms inside
// SfMToyLib/SfMBundleAdjustmentUtils.cpp:179 on one host.
#include "ba_common.h"
#include <algorithm>

namespace sfmba {

// ------------------------------------------------------------------------------------------
// sharded mode: the slotted accumulators travel through one all-reduce(SUM) as a small scalar block.
//   phase 0 (setup):  [0] ||x||^2  [1] focal column norm^2
//   phase 1 (build):  [0] sum r^2  [1] bad linearisation  [2..5] focal-focal sums  [16 + rank] gradient max-norm
//   phase 2 (update): [0] trial sum r^2  [1] model change  [2] step^2  [3] ||x_trial||^2  [4] bad trial
// ------------------------------------------------------------------------------------------
// scalars of one phase out of the slotted accumulators into the all-reduce block (called by all 64 lanes of ONE wave)
__device__ __forceinline__ void shard_pack_scalars(const DeviceBuffers& db, double* scal, int phase, int rank) {
    double v[6] = { 0, 0, 0, 0, 0, 0 };
    double gmax = 0.0;
    if (phase == 0) { v[0] = slots_take(db, ACC_XNEW2); v[1] = slots_take(db, ACC_UDF); }
    else if (phase == 1) {
        v[0] = slots_take(db, ACC_LIN_COST); v[1] = slots_take(db, ACC_BAD_LIN);
        v[2] = slots_take(db, ACC_SFF); v[3] = slots_take(db, ACC_RHSF); v[4] = slots_take(db, ACC_UDF); v[5] = slots_take(db, ACC_BCF);
        gmax = slots_take(db, ACC_GMAX);
    } else {
        v[0] = slots_take(db, ACC_TRIAL_COST); v[1] = slots_take(db, ACC_MODEL); v[2] = slots_take(db, ACC_STEP2);
        v[3] = slots_take(db, ACC_XNEW2); v[4] = slots_take(db, ACC_BAD_TRIAL);
    }
    // every entry of the block is written exactly once (the sums are uniform over the wave: each lane picks its own)
    for (int e = threadIdx.x & 63; e < SFMBA_SHARD_SCALARS; e += 64) {
        double val = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) val = e == k ? v[k] : val;
        if (phase == 1 && e == 16 + rank) val = gmax;
        scal[e] = val;
    }
}
// and back, summed over the ranks, into slot 0 (all slots are empty after the pack); one thread
__device__ __forceinline__ void shard_unpack_scalars(const DeviceBuffers& db, const double* scal, int phase, int world) {
    double* s0 = db.slots;
    if (phase == 0) { s0[ACC_XNEW2] = scal[0]; s0[ACC_UDF] = scal[1]; }
    else if (phase == 1) {
        s0[ACC_LIN_COST] = scal[0]; s0[ACC_BAD_LIN] = scal[1];
        s0[ACC_SFF] = scal[2]; s0[ACC_RHSF] = scal[3]; s0[ACC_UDF] = scal[4]; s0[ACC_BCF] = scal[5];
        double g = 0.0;
        for (int r = 0; r < world; ++r) { const double v = scal[16 + r]; g = (v > g || v != v) ? v : g; }
        reinterpret_cast<unsigned long long*>(s0)[ACC_GMAX] = (unsigned long long)__double_as_longlong(g);
    } else {
        s0[ACC_TRIAL_COST] = scal[0]; s0[ACC_MODEL] = scal[1]; s0[ACC_STEP2] = scal[2]; s0[ACC_XNEW2] = scal[3]; s0[ACC_BAD_TRIAL] = scal[4];
    }
}

__global__ void k_shard_pack(DeviceBuffers db, double* scal, int phase, int rank) { shard_pack_scalars(db, scal, phase, rank); }

__global__ void k_shard_unpack(DeviceBuffers db, const double* scal, int phase, int world) {
    if (threadIdx.x != 0) return;
    shard_unpack_scalars(db, scal, phase, world);
}

// Sharded mode with the CG solver: TWO all-reduces per linearisation instead of one over the whole reduced system.
//   (A) what the block-Jacobi factors and the LM bookkeeping need -- the 6x6 diagonal blocks, the camera-focal column, right-hand
//       side, undamped diagonal, gradient, scalars (27 ncam + 3 ld + 80 doubles);
//   (B) with the factors known on every rank, the off-diagonal blocks of the PRECONDITIONED matrix: S~ = Linv S Linv^T is linear in
//       S, so every rank transforms its own partial blocks in the pair pass -- exactly what the one-GPU path does -- and the sum over
//       the ranks is S~ (18 ncam (ncam - 1) doubles, as much as the packed triangle of S carried).
// The transform, the block factorisation and the gauge vectors then cost what they cost on one GPU (fused into k_finalize and the
// pair pass) instead of three more kernels over the reduced system behind the all-reduce.
// Layout A: [ncam][21] upper triangles of the diagonal blocks | [ncam][6] S_jf | rhs[ld] udiag[ld] bc[ld] | scalars
__global__ __launch_bounds__(256) void k_shard_diag(DeviceStructure ds, DeviceBuffers db, double* __restrict__ buf, int unpack, int rank, int world) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = ds.ncam, ld = ds.ld, fo = ds.d - 1;
    const long long n_tri = 21ll * nc, n_f = 6ll * nc, n_tail = 3ll * ld;
    double* scal = buf + n_tri + n_f + n_tail;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        if (!unpack) shard_pack_scalars(db, scal, 1, rank);
        else if (threadIdx.x == 0) shard_unpack_scalars(db, scal, 1, world);
    }
    if (e < n_tri) {
        const int j = (int)(e / 21), u = (int)(e - 21ll * j);
        int r = 0, off = u;                          // u-th entry of the row-major upper triangle of a 6 x 6 block
        while (off >= 6 - r) { off -= 6 - r; ++r; }
        double* sp = db.S + (size_t)(6 * j + r) * ld + 6 * j + r + off;
        if (unpack) *sp = buf[e]; else buf[e] = *sp;
    } else if (e < n_tri + n_f) {
        const long long k = e - n_tri;
        double* sp = db.S + (size_t)k * ld + fo;     // row 6 j + a, focal column
        if (unpack) *sp = buf[e]; else buf[e] = *sp;
    } else if (e < n_tri + n_f + n_tail) {
        const long long k = e - n_tri - n_f;         // rhs | udiag | bc are contiguous
        if (unpack) db.rhs[k] = buf[e]; else buf[e] = db.rhs[k];
    }
}
long long shard_diag_len(const DeviceStructure& ds) { return 27ll * ds.ncam + 3ll * ds.ld + SFMBA_SHARD_SCALARS; }
void launch_shard_diag(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db, double* buf, bool unpack, int rank, int world) {
    const long long n = 27ll * ds.ncam + 3ll * ds.ld;
    hipLaunchKernelGGL(k_shard_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, db, buf, unpack ? 1 : 0, rank, world);
}

// Layout B: the 36 entries of every off-diagonal block (ja < jb) of the preconditioned matrix, blocks in list order.  Unpacking
// writes both triangles of the CG's matrix.
__global__ __launch_bounds__(256) void k_shard_offdiag(DeviceStructure ds, double* __restrict__ F, double* __restrict__ buf, int unpack) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(t / 36), e = (int)(t - 36ll * b);
    if (b >= ds.nblock) return;
    const int2 cj = ds.blk_cams[b];
    if (cj.x == cj.y) return;
    const int r = e / 6, c = e - 6 * r;
    const size_t o = (size_t)(b - cj.x - 1) * 36 + e;        // block row ja holds ja + 1 diagonal blocks up to and including its own
    const size_t up = (size_t)(6 * cj.x + r) * ds.ld + 6 * cj.y + c, lo = (size_t)(6 * cj.y + c) * ds.ld + 6 * cj.x + r;
    if (unpack) { const double v = buf[o]; F[up] = v; F[lo] = v; }
    else buf[o] = F[up];
}
// Unpacking layout B through LDS, one workgroup per 8 x 8 super-tile of blocks (48 x 48 entries): an entry per thread, as k_shard_offdiag does it,
// writes the mirrored triangle as scattered 4-byte stores -- one per matrix row -- and ran at 2.3 TB/s (95 us for the 72 + 144 MB of cfg 5).  Here
// the packed blocks of a tile are read as eight contiguous runs, and both triangles leave as 48-entry rows.
template <typename FT>
__global__ __launch_bounds__(256) void k_shard_offdiag_tiled(DeviceStructure ds, FT* __restrict__ F, const FT* __restrict__ buf) {
    __shared__ FT T[48][49];
    const int I = blockIdx.y, J = blockIdx.x;
    if (J < I) return;
    const int tid = threadIdx.x;
    const int ja0 = 8 * I, jb0 = 8 * J;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int idx = tid + 256 * k;                       // 8 block rows x (8 blocks x 36 entries)
        const int jal = idx / 288, rem = idx - 288 * jal, bl = rem / 36, e = rem - 36 * bl;
        const int ja = ja0 + jal, jb = jb0 + bl;
        FT v = (FT)0;
        if (ja < ds.ncam && jb < ds.ncam && jb > ja) {
            const long long b = (long long)ja * ds.ncam - (long long)ja * (ja - 1) / 2 + (jb - ja);       // the block's list position (upper triangle incl. diagonal)
            v = buf[(size_t)(b - ja - 1) * 36 + e];
        }
        T[6 * jal + e / 6][6 * bl + e % 6] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int idx = tid + 256 * k;
        const int row = idx / 48, col = idx - 48 * row;
        {   // upper triangle: rows of camera ja, columns of camera jb
            const int ja = ja0 + row / 6, jb = jb0 + col / 6;
            if (ja < ds.ncam && jb < ds.ncam && jb > ja) F[(size_t)(6 * ja0 + row) * ds.ld + 6 * jb0 + col] = T[row][col];
        }
        {   // mirrored: rows of camera jb, columns of camera ja
            const int jb = jb0 + row / 6, ja = ja0 + col / 6;
            if (ja < ds.ncam && jb < ds.ncam && jb > ja) F[(size_t)(6 * jb0 + row) * ds.ld + 6 * ja0 + col] = T[col][row];
        }
    }
}
long long shard_offdiag_len(const DeviceStructure& ds) { return 36ll * (ds.nblock - ds.ncam); }
void launch_shard_offdiag(hipStream_t s, const DeviceStructure& ds, double* F, double* buf, bool unpack) {
    if (unpack) {
        const int nt = (ds.ncam + 7) / 8;
        hipLaunchKernelGGL(k_shard_offdiag_tiled<double>, dim3(nt, nt), dim3(256), 0, s, ds, F, buf);
        return;
    }
    const long long n = 36ll * ds.nblock;
    hipLaunchKernelGGL(k_shard_offdiag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds, F, buf, 0);
}
// the fp32 exchange: the summed blocks ARE the CG's (fp32) matrix entries
void launch_shard_offdiag_f32(hipStream_t s, const DeviceStructure& ds, float* F32, const float* buf) {
    const int nt = (ds.ncam + 7) / 8;
    hipLaunchKernelGGL(k_shard_offdiag_tiled<float>, dim3(nt, nt), dim3(256), 0, s, ds, F32, buf);
}
__global__ void k_narrow_matrix(const double* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) dst[e] = (float)src[e];
}
void launch_narrow_matrix(hipStream_t s, const double* src, float* dst, long long n) {
    hipLaunchKernelGGL(k_narrow_matrix, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, src, dst, n);
}

__global__ void k_clear_slots(DeviceBuffers db) {
    for (int e = 0; e < SLOT_W; ++e) (void)slots_take(db, e);
    if (threadIdx.x == 0) *db.fin_counter = 0;
}

__global__ void k_shard_xnorm_finish(DeviceBuffers db) {
    const double x2 = slots_take(db, ACC_XNEW2);
    if (threadIdx.x == 0) db.st->x_norm = sqrt(x2);
}

// Sharded mode: the all-reduce carries only what is meaningful -- the upper triangle of S (row r: columns r .. ld-1, packed
// row after row) followed by the tail [rhs | udiag | bc | scalars] -- i.e. ld (ld + 1) / 2 + 3 ld + 80 doubles instead of
// ld^2 + ...: half the xGMI traffic per LM iteration (145 MB instead of 289 MB at 1000 cameras).
__global__ __launch_bounds__(256) void k_shard_tri(double* __restrict__ sys, double* __restrict__ packed, int ld, long long tail, int unpack) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const long long ntri = (long long)ld * (ld + 1) / 2;
    if (c >= r && c < ld) {
        const long long o = (long long)r * ld - (long long)r * (r - 1) / 2 + (c - r);
        if (unpack) sys[(size_t)r * ld + c] = packed[o]; else packed[o] = sys[(size_t)r * ld + c];
    }
    if (r == 0) {       // the tail is contiguous behind S in both layouts
        for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tail; e += (long long)gridDim.x * blockDim.x) {
            if (unpack) sys[(size_t)ld * ld + e] = packed[ntri + e]; else packed[ntri + e] = sys[(size_t)ld * ld + e];
        }
    }
}
void launch_shard_tri(hipStream_t s, double* sys, double* packed, int ld, long long tail, bool unpack) {
    hipLaunchKernelGGL(k_shard_tri, dim3((ld + 255) / 256, ld), dim3(256), 0, s, sys, packed, ld, tail, unpack ? 1 : 0);
}

void launch_shard_pack(hipStream_t s, const DeviceBuffers& db, double* scal, int phase, int rank) {
    hipLaunchKernelGGL(k_shard_pack, dim3(1), dim3(64), 0, s, db, scal, phase, rank);
}
void launch_shard_unpack(hipStream_t s, const DeviceBuffers& db, const double* scal, int phase, int world) {
    hipLaunchKernelGGL(k_shard_unpack, dim3(1), dim3(64), 0, s, db, scal, phase, world);
}
void launch_clear_slots(hipStream_t s, const DeviceBuffers& db) { hipLaunchKernelGGL(k_clear_slots, dim3(1), dim3(64), 0, s, db); }
void launch_shard_xnorm_finish(hipStream_t s, const DeviceBuffers& db) { hipLaunchKernelGGL(k_shard_xnorm_finish, dim3(1), dim3(64), 0, s, db); }

}  // namespace sfmba
