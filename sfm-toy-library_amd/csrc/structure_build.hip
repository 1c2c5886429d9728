// structure_build.hip -- device-side construction of the camera-pair lists of the Schur pass (gfx950).
//
// Counterpart of the reference's AddResidualBlock loop (SfMToyLib/SfMBundleAdjustmentUtils.cpp:142-166): the
// reference hands every observation to ceres::Problem, whose SchurEliminator then walks the (camera, camera)
// cross terms of each point.  Here those cross terms are enumerated ONCE per problem: every pair (qa < qb) of
// observations of one point goes to block (cam(qa), cam(qb)) of the upper triangle of the reduced camera matrix.
//
// The list has N_pt * m(m-1)/2 entries (4.5 M at BASELINE config 3, 22.5 M at config 5); building it on the host
// was 2/3 of the one-shot sfmba_solve() call.  On the device: one lane per observation writes its pairs in point order
// with the block index as key, a stable LSD radix sort (hipCUB / rocPRIM) groups them by block, and a binary search
// per block gives the CSR pointers.  Stability keeps the pairs of a block in ascending point order, so the summation
// order of every S block -- and therefore every bit of the result -- is the same from run to run.
#include "ba_kernels.h"
#include "device_arena.h"

#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

#include <vector>

namespace sfmba {

namespace {

__device__ __forceinline__ unsigned block_of(int ja, int jb, int ncam) {
    return (unsigned)((long long)ja * ncam - (long long)ja * (ja - 1) / 2 + (jb - ja));
}

// One lane per observation a (point-major position): writes its pairs (a, b), b = a+1 .. end of the point -- the block as key, the POINT
// as value (the pair pass re-evaluates both observations from the block's two cameras and the point: it needs nothing else).
__global__ __launch_bounds__(256) void k_pair_gen(int nobs, int ncam, const int* __restrict__ pt_ptr, const int* __restrict__ obs_pt,
                                                  const int* __restrict__ obs_cam, const long long* __restrict__ pair_off,
                                                  unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nobs) return;
    const int i = obs_pt[a];
    const int beg = pt_ptr[i], end = pt_ptr[i + 1];
    const long long la = a - beg, n = end - beg;
    long long o = pair_off[i] + la * (n - 1) - la * (la - 1) / 2;
    const int ca = obs_cam[a];
    for (int b = a + 1; b < end; ++b, ++o) {
        keys[o] = block_of(ca, obs_cam[b], ncam);        // cameras ascend inside a point: ca <= cam(b)
        vals[o] = i;
    }
}

// blk_ptr[b] = first position whose key is >= b  (b = 0 .. nblock)
__global__ void k_block_ptr(int nblock, int npair, const unsigned* __restrict__ keys, int* __restrict__ blk_ptr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nblock) return;
    int lo = 0, hi = npair;
    while (lo < hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (keys[mid] < (unsigned)b) lo = mid + 1; else hi = mid;
    }
    blk_ptr[b] = lo;
}

}  // namespace

// d_pair_off[i] = number of pairs of the points before i (npt + 1 entries, device; build_point_major), npair their total (the caller
// knows it from its per-point observation counts).  On success *d_pair_pt (npair point slots) and *d_blk_ptr (nblock + 1 ints) live in
// `arena`; the sort's temporaries come from `scratch`, which the caller keeps until the stream has drained: nothing here waits
// for the device (the whole structure build is ONE enqueue, sfmba_api.hip build_structure).  Returns 0, or a hipError_t value.
int build_pair_lists(hipStream_t s, DeviceArena* arena, DeviceArena* scratch_arena, int npt, int nobs, int ncam, int nblock, const int* d_pt_ptr, const int* d_obs_pt,
                     const int* d_obs_cam, const long long* d_pair_off, long long npair, int** d_blk_ptr, int** d_pair_pt) {
    *d_blk_ptr = nullptr; *d_pair_pt = nullptr;
    (void)npt;
    hipError_t e;
    DeviceArena& scratch = *scratch_arena;
#define SB_TRY(expr) do { e = (expr); if (e != hipSuccess) return (int)e; } while (0)
#define SB_ALLOC(ptr, ar, T, n) do { ptr = (ar)->alloc_n<T>(n); if (!ptr) return (int)hipErrorOutOfMemory; } while (0)
    const size_t np = (size_t)(npair > 0 ? npair : 1);
    SB_ALLOC(*d_blk_ptr, arena, int, (size_t)nblock + 1);
    SB_ALLOC(*d_pair_pt, arena, int, np);
    if (npair == 0) {
        SB_TRY(hipMemsetAsync(*d_blk_ptr, 0, sizeof(int) * ((size_t)nblock + 1), s));
        SB_TRY(hipMemsetAsync(*d_pair_pt, 0, sizeof(int) * np, s));
        return 0;
    }
    unsigned *d_k0 = nullptr, *d_k1 = nullptr;
    int* d_v0 = nullptr;
    SB_ALLOC(d_k0, &scratch, unsigned, np);
    SB_ALLOC(d_k1, &scratch, unsigned, np);
    SB_ALLOC(d_v0, &scratch, int, np);
    hipLaunchKernelGGL(k_pair_gen, dim3((nobs + 255) / 256), dim3(256), 0, s, nobs, ncam, d_pt_ptr, d_obs_pt, d_obs_cam, d_pair_off, d_k0, d_v0);
    int end_bit = 1;
    while (end_bit < 32 && ((unsigned long long)1 << end_bit) < (unsigned long long)nblock) ++end_bit;
    size_t tmp_bytes = 0;
    SB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, *d_pair_pt, (int)npair, 0, end_bit, s));
    void* d_tmp = scratch.alloc(tmp_bytes ? tmp_bytes : 1);
    if (!d_tmp) return (int)hipErrorOutOfMemory;
    SB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, *d_pair_pt, (int)npair, 0, end_bit, s));
    hipLaunchKernelGGL(k_block_ptr, dim3((nblock + 1 + 255) / 256), dim3(256), 0, s, nblock, (int)npair, d_k1, *d_blk_ptr);
    SB_TRY(hipGetLastError());
#undef SB_TRY
#undef SB_ALLOC
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void k_pm_keys(int n, int cshift, const int* __restrict__ u_pt, const int* __restrict__ u_cam, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { keys[k] = ((unsigned long long)(unsigned)u_pt[k] << cshift) | (unsigned)u_cam[k]; vals[k] = k; }
}
template <typename XY>
__global__ __launch_bounds__(256) void k_pm_gather(int n, int cshift, const unsigned long long* __restrict__ keys, const int* __restrict__ src, const int* __restrict__ u_perm,
                                                   const XY* __restrict__ u_xy, int* __restrict__ obs_pt, int* __restrict__ perm, int* __restrict__ obs_cam, XY* __restrict__ obs_xy) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const unsigned long long key = keys[q];
    const int k = src[q];
    obs_pt[q] = (int)(key >> cshift); obs_cam[q] = (int)(key & (((unsigned long long)1 << cshift) - 1));
    perm[q] = u_perm[k]; obs_xy[q] = u_xy[k];
}
// pt_ptr[i] = first sorted position whose point slot is >= i; cnt[i] = pairs of point i
__global__ __launch_bounds__(256) void k_pm_ptr(int npt, int n, const int* __restrict__ obs_pt, int* __restrict__ pt_ptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > npt) return;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (obs_pt[mid] < i) lo = mid + 1; else hi = mid; }
    pt_ptr[i] = lo;
}
__global__ __launch_bounds__(256) void k_pm_paircount(int npt, const int* __restrict__ pt_ptr, long long* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > npt) return;
    const long long m = i < npt ? pt_ptr[i + 1] - pt_ptr[i] : 0;
    cnt[i] = m * (m - 1) / 2;
}
}  // namespace

// Point-major order of n observations given as unsorted device arrays of (point slot, camera slot, caller index, xy): stable
// radix sort on (point, camera) -- the order adjustBundle() adds its residual blocks in (BA.cpp:142-166: points in cloud order,
// std::map iteration = ascending view inside a point); equal keys keep their input order.  Outputs (from `arena`): the
// sorted arrays, the CSR pointers and the per-point prefix of the pair counts (pair_off[npt] = total number of pairs).  Enqueue
// only: temporaries from the caller's `scratch` arena.
// rocPRIM's radix sort falls back to a merge sort up to 2^20 items: ten merge passes over the 10^6 observations of BASELINE config 3
// (~145 us per sort, two sorts per structure build) where a few Onesweep digit passes over the significant bits do (~40 us).
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, (size_t)1 << 15> OnesweepAbove32k;
static int bits_for(unsigned long long n) { int b = 1; while (b < 63 && ((unsigned long long)1 << b) < n) ++b; return b; }

int build_point_major(hipStream_t s, DeviceArena* arena, DeviceArena* scratch_arena, int n, int npt, int ncam, int xy_bytes, const int* u_pt, const int* u_cam,
                      const int* u_perm, const void* u_xy, PointMajor* out) {
    out->obs_pt = arena->alloc_n<int>((size_t)2 * n);
    out->obs_cam = arena->alloc_n<int>((size_t)n);
    out->obs_xy = arena->alloc((size_t)xy_bytes * (n ? n : 1));
    out->pt_ptr = arena->alloc_n<int>((size_t)npt + 1);
    out->pair_off = arena->alloc_n<long long>((size_t)npt + 1);
    if (!out->obs_pt || !out->obs_cam || !out->obs_xy || !out->pt_ptr || !out->pair_off) return (int)hipErrorOutOfMemory;
    hipError_t e;
    if (n == 0) {
        if ((e = hipMemsetAsync(out->pt_ptr, 0, sizeof(int) * ((size_t)npt + 1), s)) != hipSuccess) return (int)e;
        return (int)hipMemsetAsync(out->pair_off, 0, sizeof(long long) * ((size_t)npt + 1), s);
    }
    DeviceArena& scratch = *scratch_arena;
    unsigned long long* k0 = scratch.alloc_n<unsigned long long>((size_t)n);
    unsigned long long* k1 = scratch.alloc_n<unsigned long long>((size_t)n);
    int* v0 = scratch.alloc_n<int>((size_t)n);
    int* v1 = scratch.alloc_n<int>((size_t)n);
    long long* cnt = scratch.alloc_n<long long>((size_t)npt + 1);
    if (!k0 || !k1 || !v0 || !v1 || !cnt) return (int)hipErrorOutOfMemory;
    // key = point slot above the camera slot, packed without a gap: only the significant bits are sorted (25 at 200 cameras / 10^5 points)
    const int cshift = bits_for((unsigned long long)(ncam > 1 ? ncam : 2));
    const int end_bit = cshift + bits_for((unsigned long long)(npt > 1 ? npt : 2));
    hipLaunchKernelGGL(k_pm_keys, dim3((n + 255) / 256), dim3(256), 0, s, n, cshift, u_pt, u_cam, k0, v0);
    size_t tmp_bytes = 0;
    if ((e = rocprim::radix_sort_pairs<OnesweepAbove32k>(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0u, (unsigned)end_bit, s)) != hipSuccess) return (int)e;
    void* tmp = scratch.alloc(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return (int)hipErrorOutOfMemory;
    if ((e = rocprim::radix_sort_pairs<OnesweepAbove32k>(tmp, tmp_bytes, k0, k1, v0, v1, n, 0u, (unsigned)end_bit, s)) != hipSuccess) return (int)e;
    if (xy_bytes == 8)
        hipLaunchKernelGGL(k_pm_gather<float2>, dim3((n + 255) / 256), dim3(256), 0, s, n, cshift, k1, v1, u_perm, static_cast<const float2*>(u_xy), out->obs_pt, out->obs_pt + n,
                           out->obs_cam, static_cast<float2*>(out->obs_xy));
    else
        hipLaunchKernelGGL(k_pm_gather<double2>, dim3((n + 255) / 256), dim3(256), 0, s, n, cshift, k1, v1, u_perm, static_cast<const double2*>(u_xy), out->obs_pt, out->obs_pt + n,
                           out->obs_cam, static_cast<double2*>(out->obs_xy));
    hipLaunchKernelGGL(k_pm_ptr, dim3((npt + 1 + 255) / 256), dim3(256), 0, s, npt, n, out->obs_pt, out->pt_ptr);
    hipLaunchKernelGGL(k_pm_paircount, dim3((npt + 1 + 255) / 256), dim3(256), 0, s, npt, out->pt_ptr, cnt);
    size_t sb = 0;
    if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, sb, cnt, out->pair_off, npt + 1, s)) != hipSuccess) return (int)e;
    void* st = scratch.alloc(sb ? sb : 1);
    if (!st) return (int)hipErrorOutOfMemory;
    if ((e = hipcub::DeviceScan::ExclusiveSum(st, sb, cnt, out->pair_off, npt + 1, s)) != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

namespace {
__global__ void k_iota_cam(int n, const int* __restrict__ obs_cam, unsigned* __restrict__ keys, int* __restrict__ vals) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) { keys[q] = (unsigned)obs_cam[q]; vals[q] = q; }
}
__global__ void k_gather_pt(int n, const int* __restrict__ cam_obs, const int* __restrict__ obs_pt, int* __restrict__ cam_obs_pt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) cam_obs_pt[e] = obs_pt[cam_obs[e]];
}
}  // namespace

namespace {
template <typename XY>
__global__ __launch_bounds__(256) void k_gather_xy(int n, const int* __restrict__ cam_obs, const XY* __restrict__ obs_xy, XY* __restrict__ cam_obs_xy) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) cam_obs_xy[e] = obs_xy[cam_obs[e]];
}
}  // namespace
// The observation coordinates once more in camera-major order (static: built with the structure): the re-evaluating camera pass reads
// them coalesced and forms the residual itself instead of gathering a stored one by point-major position.
int build_camera_major_xy(hipStream_t s, DeviceArena* arena, int nobs, int xy_bytes, const int* d_cam_obs, const void* d_obs_xy, void** d_cam_obs_xy) {
    *d_cam_obs_xy = arena->alloc((size_t)xy_bytes * (nobs ? nobs : 1));
    if (!*d_cam_obs_xy) return (int)hipErrorOutOfMemory;
    if (nobs == 0) return 0;
    if (xy_bytes == 8) hipLaunchKernelGGL(k_gather_xy<float2>, dim3((nobs + 255) / 256), dim3(256), 0, s, nobs, d_cam_obs, static_cast<const float2*>(d_obs_xy), static_cast<float2*>(*d_cam_obs_xy));
    else hipLaunchKernelGGL(k_gather_xy<double2>, dim3((nobs + 255) / 256), dim3(256), 0, s, nobs, d_cam_obs, static_cast<const double2*>(d_obs_xy), static_cast<double2*>(*d_cam_obs_xy));
    return (int)hipGetLastError();
}

// Camera-major index of the observations on the device: cam_obs[e] = point-major position q, grouped by camera with a stable
// radix sort (ascending q, i.e. ascending point, inside a camera -- the order the host loop produced), cam_obs_pt[e] = its point.
int build_camera_major(hipStream_t s, DeviceArena* arena, DeviceArena* scratch_arena, int nobs, int ncam, const int* d_obs_cam, const int* d_obs_pt,
                       int** d_cam_obs, int** d_cam_obs_pt, int** d_cam_ptr) {
    *d_cam_obs = arena->alloc_n<int>((size_t)nobs);
    *d_cam_obs_pt = arena->alloc_n<int>((size_t)nobs);
    *d_cam_ptr = arena->alloc_n<int>((size_t)ncam + 1);
    if (!*d_cam_obs || !*d_cam_obs_pt || !*d_cam_ptr) return (int)hipErrorOutOfMemory;
    if (nobs == 0) return (int)hipMemsetAsync(*d_cam_ptr, 0, sizeof(int) * ((size_t)ncam + 1), s);
    DeviceArena& scratch = *scratch_arena;
    unsigned* k0 = scratch.alloc_n<unsigned>((size_t)nobs);
    unsigned* k1 = scratch.alloc_n<unsigned>((size_t)nobs);
    int* v0 = scratch.alloc_n<int>((size_t)nobs);
    if (!k0 || !k1 || !v0) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL(k_iota_cam, dim3((nobs + 255) / 256), dim3(256), 0, s, nobs, d_obs_cam, k0, v0);
    int end_bit = 1;
    while (end_bit < 32 && ((unsigned long long)1 << end_bit) < (unsigned long long)ncam) ++end_bit;
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs<OnesweepAbove32k>(nullptr, tmp_bytes, k0, k1, v0, *d_cam_obs, nobs, 0u, (unsigned)end_bit, s);
    if (e != hipSuccess) return (int)e;
    void* tmp = scratch.alloc(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return (int)hipErrorOutOfMemory;
    e = rocprim::radix_sort_pairs<OnesweepAbove32k>(tmp, tmp_bytes, k0, k1, v0, *d_cam_obs, nobs, 0u, (unsigned)end_bit, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_gather_pt, dim3((nobs + 255) / 256), dim3(256), 0, s, nobs, *d_cam_obs, d_obs_pt, *d_cam_obs_pt);
    hipLaunchKernelGGL(k_block_ptr, dim3((ncam + 1 + 255) / 256), dim3(256), 0, s, ncam, nobs, k1, *d_cam_ptr);      // CSR pointers from the sorted keys
    return (int)hipGetLastError();
}

namespace {
// Pair-pass descriptors of every workgroup slot: (block, its row camera, pair range) from the block CSR on the device
__global__ __launch_bounds__(256) void k_pair_desc(int nslot, int group, const int2* __restrict__ pwg_blocks, const int2* __restrict__ blk_cams,
                                                   const int* __restrict__ blk_ptr, const int* __restrict__ perm, int4* __restrict__ desc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslot) return;
    const int2 w = pwg_blocks[t / group];
    const int k = t % group;
    int4 d; d.x = -1; d.y = 0; d.z = 0; d.w = 0;
    if (k < w.y) { const int b = perm ? perm[w.x + k] : w.x + k; d.x = b; d.y = blk_cams[b].x; d.z = blk_ptr[b]; d.w = blk_ptr[b + 1]; }
    desc[t] = d;
}
// Order of the blocks of one block row in the sixteen-lanes-per-block pair pass: by ROUNDS of sixteen pairs (the diagonal block first).  A wave
// of that pass holds four consecutive blocks of the row and loops to the longest of them: with ~45 pairs per block (Poisson) four blocks
// in list order need 4 rounds for 2.8 rounds of work; grouped by rounds a wave's blocks loop alike.  One workgroup per row; the order
// inside a bucket is whatever the atomics give (every block is computed on its own: the order changes no result).
__global__ __launch_bounds__(256) void k_row_order(int ncam, int lpb, const int* __restrict__ blk_ptr, int* __restrict__ perm) {
    constexpr int NBK = 34;
    __shared__ int start[NBK + 1], cursor[NBK];
    const int ja = blockIdx.x, nb = ncam - ja;
    const int b0 = (int)block_of(ja, ja, ncam);
    for (int k = threadIdx.x; k <= NBK; k += blockDim.x) { start[k] = 0; if (k < NBK) cursor[k] = 0; }
    __syncthreads();
    auto bucket = [&](int t) { if (t == 0) return 0; const int n = blk_ptr[b0 + t + 1] - blk_ptr[b0 + t]; const int r = 1 + (n + lpb - 1) / lpb; return r < NBK ? r : NBK - 1; };
    for (int t = threadIdx.x; t < nb; t += blockDim.x) atomicAdd(&start[bucket(t) + 1], 1);
    __syncthreads();
    if (threadIdx.x == 0) { for (int k = 0; k < NBK; ++k) start[k + 1] += start[k]; }
    __syncthreads();
    for (int t = threadIdx.x; t < nb; t += blockDim.x) { const int k = bucket(t); perm[b0 + start[k] + atomicAdd(&cursor[k], 1)] = b0 + t; }
}
// Diagonal blocks that hold pairs (one camera observing a point twice), ascending camera order; their number and the device's own
// total pair count go to host-mapped memory (report[0] = blocks, report[2..3] = pairs), read by the host once the stream has drained.
__global__ __launch_bounds__(256) void k_dup_blocks(int ncam, const int* __restrict__ blk_ptr, const long long* __restrict__ pair_total,
                                                    int2* __restrict__ dup, int* __restrict__ report) {
    __shared__ int sc[256];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < ncam; j0 += 256) {
        const int j = j0 + tid;
        int b = 0, has = 0;
        if (j < ncam) { b = (int)block_of(j, j, ncam); has = blk_ptr[b + 1] > blk_ptr[b]; }
        sc[tid] = has;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? sc[tid - off] : 0;
            __syncthreads();
            sc[tid] += v;
            __syncthreads();
        }
        if (has) { int2 w; w.x = b; w.y = 1; dup[base + sc[tid] - 1] = w; }
        __syncthreads();
        if (tid == 0) base += sc[255];
        __syncthreads();
    }
    if (tid == 0) {
        report[0] = base;
        const long long np = *pair_total;
        report[2] = (int)(np & 0xffffffffll); report[3] = (int)(np >> 32);
        __threadfence_system();
    }
}
}  // namespace

namespace {
template <typename XY>
__global__ __launch_bounds__(256) void k_stage_obs(StageObs so) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= so.n_old + so.n_new) return;
    const bool old = k < so.n_old;
    const int j = old ? k : k - so.n_old;
    so.u_pt[k] = (old ? so.old_pt : so.new_pt)[j];
    so.u_cam[k] = (old ? so.old_cam : so.new_cam)[j];
    so.u_perm[k] = (old ? so.old_perm : so.new_perm)[j];
    static_cast<XY*>(so.u_xy)[k] = static_cast<const XY*>(old ? so.old_xy : so.new_xy)[j];
}
}  // namespace
void launch_stage_obs(hipStream_t s, const StageObs& so, int xy_bytes) {
    const int n = so.n_old + so.n_new;
    if (n <= 0) return;
    if (xy_bytes == 8) hipLaunchKernelGGL(k_stage_obs<float2>, dim3((n + 255) / 256), dim3(256), 0, s, so);
    else hipLaunchKernelGGL(k_stage_obs<double2>, dim3((n + 255) / 256), dim3(256), 0, s, so);
}
// ---- wave-per-block pair pass: heavy blocks are cut into chunks of PAIR_CHUNK pairs, one wave each ----
// Realistic co-visibility concentrates the pairs in few blocks (cameras on a path, tracks of neighbouring cameras: 5 800 non-empty blocks
// of 19 900 with ~1 300 pairs on average and up to 4 650 at BASELINE-config-3 size), and a wave per block then leaves the launch to its
// longest waves (73 rounds of 64 pairs) at a third of the machine's occupancy.  One descriptor per CHUNK, in the order of the host's
// workgroup list; a block of several chunks leaves partial sums behind that k_schur_combine adds (list `multi`: the first slot of every
// such block).  The slot total and the number of multi-chunk blocks go to host-mapped memory (report[4], report[5]).
namespace {
__global__ __launch_bounds__(256) void k_chunk_count(int nwg, int chunk, const int2* __restrict__ pwg_blocks, const int2* __restrict__ blk_cams,
                                                     const int* __restrict__ blk_ptr, int* __restrict__ cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nwg) return;
    int c = 0;
    if (s < nwg) {
        const int2 w = pwg_blocks[s];
        if (w.y > 0) {
            const int2 cj = blk_cams[w.x];
            const int n = blk_ptr[w.x + 1] - blk_ptr[w.x];
            c = cj.x == cj.y ? 1 : (n + chunk - 1) / chunk;
            if (c < 1) c = 1;
        }
    }
    cnt[s] = c;
}
__global__ __launch_bounds__(256) void k_chunk_fill(int nwg, int chunk, const int2* __restrict__ pwg_blocks, const int2* __restrict__ blk_cams,
                                                    const int* __restrict__ blk_ptr, const int* __restrict__ off, int4* __restrict__ desc,
                                                    int2* __restrict__ info, int* __restrict__ multi, int* __restrict__ counters) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nwg) return;
    const int2 w = pwg_blocks[s];
    if (w.y <= 0) return;
    const int b = w.x;
    const int2 cj = blk_cams[b];
    const int pb = blk_ptr[b], pe = blk_ptr[b + 1];
    const int o = off[s], nch = off[s + 1] - o;
    // a block of several chunks gets nch consecutive rows of the partial-sum buffer (pair_partial): which ones is a storage detail (the
    // combine adds them in chunk order), so the rows are handed out by a counter and the buffer is as large as the chunks that need it
    const int prow = nch > 1 ? atomicAdd(&counters[3], nch) : 0;
    for (int c = 0; c < nch; ++c) {
        int4 d; d.x = b; d.y = cj.x;
        d.z = nch == 1 ? pb : pb + c * chunk;
        d.w = nch == 1 ? pe : min(pe, pb + (c + 1) * chunk);
        desc[o + c] = d;
        int2 ci; ci.x = prow + c; ci.y = nch;
        info[o + c] = ci;
    }
    if (nch > 1) multi[atomicAdd(&counters[0], 1)] = o;
}
__global__ void k_chunk_report(int nwg, const int* __restrict__ off, const int* __restrict__ counters, int* __restrict__ report) {
    report[4] = off[nwg];
    report[5] = counters[0];
    report[7] = counters[3];
    __threadfence_system();
}
// number of off-diagonal blocks of the upper triangle that hold at least one pair -> report[1], those close to the diagonal -> report[6] (the fill of the reduced matrix: what
// SFMBA_LINEAR_AUTO reads the co-visibility structure from)
__global__ __launch_bounds__(256) void k_block_fill(int nblock, int ncam, const int2* __restrict__ blk_cams, const int* __restrict__ blk_ptr, int* __restrict__ counters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int has = 0, near = 0;
    if (b < nblock) {
        const int2 cj = blk_cams[b];
        has = cj.x != cj.y && blk_ptr[b + 1] > blk_ptr[b];
        // ... and how many of them couple cameras that are close in the (cyclic) camera order: within a quarter of it.  A camera graph laid out
        // along its index -- views registered one after the other -- has all of its blocks there, a random one half of them.
        const int dist = cj.y - cj.x, cyc = dist < ncam - dist ? dist : ncam - dist;
        near = has && 4 * cyc <= ncam;
    }
    const unsigned long long m = __ballot(has), mn = __ballot(near);
    if ((threadIdx.x & 63) == 0 && m) { atomicAdd(&counters[1], __popcll(m)); if (mn) atomicAdd(&counters[2], __popcll(mn)); }
}
__global__ void k_fill_report(const int* __restrict__ counters, int* __restrict__ report) { report[1] = counters[1]; report[6] = counters[2]; __threadfence_system(); }
}  // namespace

int build_pair_chunks(hipStream_t s, DeviceArena* scratch, int nwg, int chunk, const int2* pwg_blocks, const int2* blk_cams, const int* blk_ptr,
                      int4* desc, int2* info, int* multi, int* counters, int* report) {
    if (nwg <= 0) return 0;
    int* cnt = scratch->alloc_n<int>((size_t)nwg + 1);
    int* off = scratch->alloc_n<int>((size_t)nwg + 1);
    if (!cnt || !off) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL(k_chunk_count, dim3((nwg + 1 + 255) / 256), dim3(256), 0, s, nwg, chunk, pwg_blocks, blk_cams, blk_ptr, cnt);
    size_t tmp_bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, off, nwg + 1, s);
    if (e != hipSuccess) return (int)e;
    void* tmp = scratch->alloc(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return (int)hipErrorOutOfMemory;
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, off, nwg + 1, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_chunk_fill, dim3((nwg + 255) / 256), dim3(256), 0, s, nwg, chunk, pwg_blocks, blk_cams, blk_ptr, off, desc, info, multi, counters);
    hipLaunchKernelGGL(k_chunk_report, dim3(1), dim3(1), 0, s, nwg, off, counters, report);
    return (int)hipGetLastError();
}
// Per camera j the set of cameras c whose block (min, max) of the reduced matrix holds a pair -- and j itself --, as a bit mask of `words` 32-bit words:
// what a block-sparse product of the CG needs (dense_solver.hip, k_sg_q_sparse).  One thread per (camera, word).
__global__ __launch_bounds__(256) void k_block_mask(int ncam, int words, const int* __restrict__ blk_ptr, unsigned* __restrict__ mask) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncam * words) return;
    const int j = t / words, wd = t - words * j;
    unsigned m = 0;
    for (int bit = 0; bit < 32; ++bit) {
        const int c = 32 * wd + bit;
        if (c >= ncam) break;
        const int ja = j < c ? j : c, jb = j < c ? c : j;
        const long long b = (long long)ja * ncam - (long long)ja * (ja - 1) / 2 + (jb - ja);
        if (c == j || blk_ptr[b + 1] > blk_ptr[b]) m |= 1u << bit;
    }
    mask[t] = m;
}
void launch_block_mask(hipStream_t s, int ncam, const int* blk_ptr, unsigned* mask) {
    const int words = (ncam + 31) / 32;
    if (ncam > 0) hipLaunchKernelGGL(k_block_mask, dim3((ncam * words + 255) / 256), dim3(256), 0, s, ncam, words, blk_ptr, mask);
}
void launch_block_fill(hipStream_t s, int nblock, int ncam, const int2* blk_cams, const int* blk_ptr, int* counters, int* report) {
    if (nblock > 0) hipLaunchKernelGGL(k_block_fill, dim3((nblock + 255) / 256), dim3(256), 0, s, nblock, ncam, blk_cams, blk_ptr, counters);
    hipLaunchKernelGGL(k_fill_report, dim3(1), dim3(1), 0, s, counters, report);
}

void launch_row_order(hipStream_t s, int ncam, int lpb, const int* blk_ptr, int* perm) {
    if (ncam > 0) hipLaunchKernelGGL(k_row_order, dim3(ncam), dim3(256), 0, s, ncam, lpb, blk_ptr, perm);
}
void launch_pair_desc(hipStream_t s, int nwg, int group, const int2* pwg_blocks, const int2* blk_cams, const int* blk_ptr, const int* perm, int4* desc) {
    const int nslot = nwg * group;
    if (nslot > 0) hipLaunchKernelGGL(k_pair_desc, dim3((nslot + 255) / 256), dim3(256), 0, s, nslot, group, pwg_blocks, blk_cams, blk_ptr, perm, desc);
}
void launch_dup_blocks(hipStream_t s, int ncam, const int* blk_ptr, const long long* pair_total, int2* dup, int* report) {
    hipLaunchKernelGGL(k_dup_blocks, dim3(1), dim3(256), 0, s, ncam, blk_ptr, pair_total, dup, report);
}

}  // namespace sfmba
