// association.hip -- the two association loops of the incremental pipeline as device-side joins (gfx950).
//
// Reference:
//   SfM::find2D3DMatches     SfMToyLib/SfM.cpp:471-528   for every not-done view x every cloud point x every originating
//                            view: linear scan of that view pair's match list for the point's feature -> O(n^4) on the host
//   SfM::mergeNewPointCloud  SfMToyLib/SfM.cpp:530-629   every new point against every existing point (distance test), then
//                            a feature-match confirmation per close pair -> O(n^2) distance tests
//
// Both are integer / index work; results here are BIT-EXACT with the reference loops (same entries, same order):
//
//   find:   "the first match in list order whose query (train) index equals the feature" is a hash lookup
//           (pair, side, feature) -> smallest list position, built with atomicMin -- order independent, hence
//           deterministic.  One lane per (view, cloud point) walks the point's originating views in ascending order
//           (std::map order) and stops at the first hit; a stable compaction (exclusive scan) restores cloud order.
//   merge:  the distance test of every (existing or earlier-new point, new point) pair is a radius join over a uniform
//           grid (cell edge just above the threshold, 27 neighbour cells, points sorted by cell key); the candidate lists
//           come back in ascending index order, which is the order the reference's scan meets them in.  The data-dependent
//           sequential part (views added to existing points change later decisions, SfM.cpp:582) stays with the caller
//           (host/SfMAssociation.cpp) and only ever touches the candidates.
//
// HBM-bound hash / sort work: no MFMA, coalesced streams over the match lists and the cloud, random 8-byte probes.
#include "association.h"
#include "device_arena.h"

#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

namespace sfmba {

namespace {

typedef unsigned long long u64;
constexpr u64 EMPTY_KEY = ~0ull;

__device__ __forceinline__ u64 mix64(u64 x) {       // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ u64 match_key(int pair, int side, int feature) {
    return ((u64)(unsigned)pair << 33) | ((u64)(unsigned)side << 32) | (u64)(unsigned)feature;
}

__global__ __launch_bounds__(256) void k_table_clear(u64* __restrict__ keys, int* __restrict__ vals, long long n) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { keys[e] = EMPTY_KEY; vals[e] = INT_MAX; }
}

__device__ __forceinline__ void table_insert_min(u64* keys, int* vals, unsigned mask, u64 key, int pos) {
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const u64 prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) { atomicMin(&vals[slot], pos); return; }
        slot = (slot + 1) & mask;
    }
}

__device__ __forceinline__ int table_lookup(const u64* __restrict__ keys, const int* __restrict__ vals, unsigned mask, u64 key) {
    unsigned slot = (unsigned)mix64(key) & mask;
    for (;;) {
        const u64 k = keys[slot];
        if (k == key) return vals[slot];
        if (k == EMPTY_KEY) return -1;
        slot = (slot + 1) & mask;
    }
}

// One lane per match (list position): pair by binary search in pair_ptr, two keys.  A match whose OTHER index is negative
// can never be "found" (SfM.cpp:508: matched2DPointInNewView >= 0), so it is not indexed and later matches stay visible.
__global__ __launch_bounds__(256) void k_index_matches(int n_pairs, const long long* __restrict__ pair_ptr, const int* __restrict__ pair_ok,
                                                       const int* __restrict__ query, const int* __restrict__ train, long long n_match,
                                                       u64* keys, int* vals, unsigned mask) {
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_match) return;
    int lo = 0, hi = n_pairs;             // last pair with pair_ptr[p] <= pos
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pair_ptr[mid] <= pos) lo = mid; else hi = mid; }
    if (!pair_ok[lo]) return;             // lower-triangle or duplicate entry of the pair table: never consulted
    const int q = query[pos], t = train[pos];
    if (t >= 0) table_insert_min(keys, vals, mask, match_key(lo, 0, q), (int)pos);
    if (q >= 0) table_insert_min(keys, vals, mask, match_key(lo, 1, t), (int)pos);
}

// One lane per (not-done view, cloud point): SfM.cpp:483-521.
__global__ __launch_bounds__(256) void k_find_2d3d(int n_views, int nvs, const int* __restrict__ nd_views, int n_pt,
                                                   const long long* __restrict__ view_ptr, const int* __restrict__ view_idx,
                                                   const int* __restrict__ feat_idx, const int* __restrict__ pair_tab,
                                                   const int* __restrict__ query, const int* __restrict__ train,
                                                   const u64* __restrict__ keys, const int* __restrict__ vals, unsigned mask,
                                                   int* __restrict__ hit_feat, int* __restrict__ flags) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)nvs * n_pt) return;
    const int vs = (int)(e / n_pt), i = (int)(e - (long long)vs * n_pt);
    const int view = nd_views[vs];
    int found = -1;
    for (long long o = view_ptr[i]; o < view_ptr[i + 1]; ++o) {          // ascending originating view (std::map order)
        const int ov = view_idx[o], of = feat_idx[o];
        if (ov < 0 || ov >= n_views) continue;
        const bool orig_left = ov < view;
        const int left = orig_left ? ov : view, right = orig_left ? view : ov;
        const int p = pair_tab[(size_t)left * n_views + right];
        if (p < 0) continue;
        const int pos = table_lookup(keys, vals, mask, match_key(p, orig_left ? 0 : 1, of));
        if (pos >= 0) { found = orig_left ? train[pos] : query[pos]; break; }
    }
    hit_feat[e] = found;
    flags[e] = found >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_find_scatter(int nvs, int n_pt, const int* __restrict__ hit_feat, const int* __restrict__ flags,
                                                      const int* __restrict__ pos, int* __restrict__ out_point, int* __restrict__ out_feat,
                                                      long long* __restrict__ out_ptr, long long cap) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)nvs * n_pt;
    if (e >= n) return;
    const int vs = (int)(e / n_pt), i = (int)(e - (long long)vs * n_pt);
    if (i == 0) out_ptr[vs] = pos[e];
    if (e == n - 1) out_ptr[nvs] = (long long)pos[e] + flags[e];
    if (flags[e] && pos[e] < cap) { out_point[pos[e]] = i; out_feat[pos[e]] = hit_feat[e]; }
}

// ---------------------------------------------------------------------------------------------------------------
// radius join
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool finite3(float x, float y, float z) { return fabsf(x) <= FLT_MAX && fabsf(y) <= FLT_MAX && fabsf(z) <= FLT_MAX; }

__device__ __forceinline__ long long cell_of(float x, double inv_cell) {
    double q = floor((double)x * inv_cell);
    q = fmin(fmax(q, -2147483648.0), 2147483647.0);
    return (long long)q;
}

__device__ __forceinline__ u64 cell_key(long long cx, long long cy, long long cz) {
    // lossless inside +-2^20 cells per axis (10^4 scene units at the reference's 0.01 threshold), hashed outside; a collision only
    // costs extra exact distance tests and, at worst, a duplicate candidate that the final unique pass removes
    const long long lim = 1ll << 20;
    if (cx >= -lim && cx < lim && cy >= -lim && cy < lim && cz >= -lim && cz < lim)
        return ((u64)(cx + lim) << 42) | ((u64)(cy + lim) << 21) | (u64)(cz + lim);
    return mix64((u64)cx * 0x9e3779b97f4a7c15ull ^ mix64((u64)cy ^ mix64((u64)cz))) | (1ull << 63);
}

__global__ __launch_bounds__(256) void k_cell_keys(int n, const float* __restrict__ xyz, double inv_cell, u64* __restrict__ keys, int* __restrict__ idx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float x = xyz[3 * (size_t)j], y = xyz[3 * (size_t)j + 1], z = xyz[3 * (size_t)j + 2];
    // a point with a non-finite coordinate is at NaN / infinite distance from everything (never < threshold): parked under a
    // key no query ever asks for
    keys[j] = finite3(x, y, z) ? cell_key(cell_of(x, inv_cell), cell_of(y, inv_cell), cell_of(z, inv_cell)) : EMPTY_KEY;
    idx[j] = j;
}

// cv::norm(existing - new) < max_dist with cv::Point3f operands (SfM.cpp:544): float difference, double norm
__device__ __forceinline__ bool within(const float* __restrict__ xyz, int j, float nx, float ny, float nz, double max_dist) {
    const float dx = xyz[3 * (size_t)j] - nx, dy = xyz[3 * (size_t)j + 1] - ny, dz = xyz[3 * (size_t)j + 2] - nz;
    const double s = (double)dx * (double)dx + (double)dy * (double)dy + (double)dz * (double)dz;
    return __dsqrt_rn(s) < max_dist;
}

// One lane per (new point k, neighbour cell): COUNT = tally hits, else append (k << 32 | j).
template <bool COUNT>
__global__ __launch_bounds__(256) void k_radius_probe(int n_exist, int n_new, const float* __restrict__ xyz, double inv_cell, double max_dist,
                                                      int n_all, const u64* __restrict__ sorted_keys, const int* __restrict__ sorted_idx,
                                                      u64* __restrict__ total, u64* __restrict__ out, u64 cap) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n_new * 27) return;
    const int k = (int)(e / 27), nb = (int)(e - (long long)k * 27);
    const int self = n_exist + k;
    const float nx = xyz[3 * (size_t)self], ny = xyz[3 * (size_t)self + 1], nz = xyz[3 * (size_t)self + 2];
    if (!finite3(nx, ny, nz)) return;
    const long long cx = cell_of(nx, inv_cell) + (nb % 3 - 1), cy = cell_of(ny, inv_cell) + ((nb / 3) % 3 - 1), cz = cell_of(nz, inv_cell) + (nb / 9 - 1);
    const u64 key = cell_key(cx, cy, cz);
    int lo = 0, hi = n_all;
    while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid; }
    unsigned hits = 0;
    for (int r = lo; r < n_all && sorted_keys[r] == key; ++r) {
        const int j = sorted_idx[r];
        if (j >= self) continue;                       // only existing points and EARLIER new points (the cloud as of new point k)
        if (!within(xyz, j, nx, ny, nz, max_dist)) continue;
        if (COUNT) ++hits;
        else {
            const u64 at = atomicAdd(total, 1ull);
            if (at < cap) out[at] = ((u64)(unsigned)k << 32) | (u64)(unsigned)j;
        }
    }
    if (COUNT && hits) atomicAdd(total, (u64)hits);
}

__global__ __launch_bounds__(256) void k_cand_ptr(int n_new, long long n_cand, const u64* __restrict__ sorted, long long* __restrict__ ptr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_new) return;
    const u64 key = (u64)(unsigned)k << 32;
    long long lo = 0, hi = n_cand;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (sorted[mid] < key) lo = mid + 1; else hi = mid; }
    ptr[k] = lo;
}

__global__ __launch_bounds__(256) void k_cand_low(long long n, const u64* __restrict__ sorted, int* __restrict__ idx) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) idx[e] = (int)(unsigned)(sorted[e] & 0xffffffffull);
}

#define AS_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)
#define AS_ALLOC(ptr, T, n) do { ptr = scratch.alloc_n<T>(n); if (!ptr) return (int)hipErrorOutOfMemory; } while (0)

template <typename T>
int upload(DeviceArena& scratch, hipStream_t s, T** d, const T* h, size_t n) {
    *d = scratch.alloc_n<T>(n);
    if (!*d) return (int)hipErrorOutOfMemory;
    if (n) AS_TRY(hipMemcpyAsync(*d, h, sizeof(T) * n, hipMemcpyHostToDevice, s));
    return 0;
}

unsigned grid_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

int assoc_find_2d3d(hipStream_t s, int device, int n_views, const unsigned char* view_done, int n_pt, const int64_t* view_ptr,
                    const int32_t* view_idx, const int32_t* feat_idx, int n_pairs, const int32_t* pair_left, const int32_t* pair_right,
                    const int64_t* pair_ptr, const int32_t* query_idx, const int32_t* train_idx, int64_t* out_ptr, int32_t* out_point,
                    int32_t* out_feature, int64_t cap, int64_t* total) {
    DeviceArena scratch(device);
    std::vector<int> nd;
    for (int v = 0; v < n_views; ++v) if (!view_done[v]) nd.push_back(v);
    const int nvs = (int)nd.size();
    for (int v = 0; v <= n_views; ++v) out_ptr[v] = 0;
    *total = 0;
    const long long n = (long long)nvs * n_pt;
    if (n == 0) return 0;
    if (n >= INT_MAX || (long long)n_views * n_views >= INT_MAX) return ASSOC_ERR_TOO_LARGE;
    const long long n_match = n_pairs > 0 ? pair_ptr[n_pairs] : 0, n_ov = view_ptr[n_pt];
    if (n_match >= INT_MAX) return ASSOC_ERR_TOO_LARGE;
    // pair table: [left][right] -> pair index; only the entries the reference consults (left <= right), first occurrence
    std::vector<int> tab((size_t)n_views * n_views, -1), pair_ok((size_t)std::max(n_pairs, 1), 0);
    for (int p = 0; p < n_pairs; ++p) {
        const int l = pair_left[p], r = pair_right[p];
        if (l < 0 || r < 0 || l >= n_views || r >= n_views || l > r) continue;
        int& slot = tab[(size_t)l * n_views + r];
        if (slot < 0) { slot = p; pair_ok[p] = 1; }
    }
    // the probe sequence works on a 32-bit slot index (mask = 2^bits - 1, bits <= 31): more than 2^29 matches would need
    // bits >= 32, where `1u << bits` is undefined -- refuse instead
    if (4 * n_match > (1ll << 31)) return ASSOC_ERR_TOO_LARGE;
    unsigned bits = 4;
    while ((1ull << bits) < (u64)(4 * std::max(n_match, 1ll))) ++bits;
    const unsigned mask = (1u << bits) - 1u;
    const long long n_slot = 1ll << bits;

    int *d_nd, *d_vidx, *d_fidx, *d_tab, *d_ok, *d_q, *d_t, *d_vals, *d_hit, *d_flag, *d_pos, *d_opt, *d_ofe;
    long long *d_vptr, *d_pptr, *d_optr;
    u64* d_keys;
    int rc;
    if ((rc = upload(scratch, s, &d_nd, nd.data(), (size_t)nvs))) return rc;
    if ((rc = upload(scratch, s, &d_vptr, reinterpret_cast<const long long*>(view_ptr), (size_t)n_pt + 1))) return rc;
    if ((rc = upload(scratch, s, &d_vidx, view_idx, (size_t)n_ov))) return rc;
    if ((rc = upload(scratch, s, &d_fidx, feat_idx, (size_t)n_ov))) return rc;
    if ((rc = upload(scratch, s, &d_tab, tab.data(), tab.size()))) return rc;
    if ((rc = upload(scratch, s, &d_ok, pair_ok.data(), pair_ok.size()))) return rc;
    if ((rc = upload(scratch, s, &d_pptr, reinterpret_cast<const long long*>(pair_ptr), (size_t)n_pairs + 1))) return rc;
    if ((rc = upload(scratch, s, &d_q, query_idx, (size_t)n_match))) return rc;
    if ((rc = upload(scratch, s, &d_t, train_idx, (size_t)n_match))) return rc;
    AS_ALLOC(d_keys, u64, (size_t)n_slot);
    AS_ALLOC(d_vals, int, (size_t)n_slot);
    AS_ALLOC(d_hit, int, (size_t)n);
    AS_ALLOC(d_flag, int, (size_t)n);
    AS_ALLOC(d_pos, int, (size_t)n);
    AS_ALLOC(d_optr, long long, (size_t)nvs + 1);
    const long long ocap = std::min<long long>(std::max<long long>(cap, 0), n);
    AS_ALLOC(d_opt, int, (size_t)std::max<long long>(ocap, 1));
    AS_ALLOC(d_ofe, int, (size_t)std::max<long long>(ocap, 1));

    hipLaunchKernelGGL(k_table_clear, dim3(grid_for(n_slot)), dim3(256), 0, s, d_keys, d_vals, n_slot);
    if (n_match > 0)
        hipLaunchKernelGGL(k_index_matches, dim3(grid_for(n_match)), dim3(256), 0, s, n_pairs, d_pptr, d_ok, d_q, d_t, n_match, d_keys, d_vals, mask);
    hipLaunchKernelGGL(k_find_2d3d, dim3(grid_for(n)), dim3(256), 0, s, n_views, nvs, d_nd, n_pt, d_vptr, d_vidx, d_fidx, d_tab, d_q, d_t,
                       d_keys, d_vals, mask, d_hit, d_flag);
    size_t tmp_bytes = 0;
    AS_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_flag, d_pos, (int)n, s));
    void* d_tmp = scratch.alloc(tmp_bytes ? tmp_bytes : 1);
    if (!d_tmp) return (int)hipErrorOutOfMemory;
    AS_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flag, d_pos, (int)n, s));
    hipLaunchKernelGGL(k_find_scatter, dim3(grid_for(n)), dim3(256), 0, s, nvs, n_pt, d_hit, d_flag, d_pos, d_opt, d_ofe, d_optr, ocap);
    AS_TRY(hipGetLastError());
    std::vector<long long> optr((size_t)nvs + 1);
    AS_TRY(hipMemcpyAsync(optr.data(), d_optr, sizeof(long long) * optr.size(), hipMemcpyDeviceToHost, s));
    AS_TRY(hipStreamSynchronize(s));
    // per-view pointers in the caller's view numbering: done views get empty ranges
    const long long tot = optr[(size_t)nvs];
    *total = tot;
    {
        int vs = 0;
        for (int v = 0; v < n_views; ++v) {
            out_ptr[v] = vs < nvs ? optr[(size_t)vs] : tot;
            if (vs < nvs && nd[(size_t)vs] == v) ++vs;
        }
        out_ptr[n_views] = tot;
    }
    if (tot > cap) return ASSOC_ERR_CAPACITY;
    if (tot > 0) {
        AS_TRY(hipMemcpyAsync(out_point, d_opt, sizeof(int) * (size_t)tot, hipMemcpyDeviceToHost, s));
        AS_TRY(hipMemcpyAsync(out_feature, d_ofe, sizeof(int) * (size_t)tot, hipMemcpyDeviceToHost, s));
        AS_TRY(hipStreamSynchronize(s));
    }
    return 0;
}

int assoc_radius_candidates(hipStream_t s, int device, int n_exist, const float* exist_xyz, int n_new, const float* new_xyz, float max_dist,
                            int64_t* cand_ptr, int32_t* cand_idx, int64_t cap, int64_t* total) {
    for (int k = 0; k <= n_new; ++k) cand_ptr[k] = 0;
    *total = 0;
    if (n_new == 0) return 0;
    if ((long long)n_exist + n_new >= INT_MAX / 32) return ASSOC_ERR_TOO_LARGE;
    if (!(max_dist > 0.0f)) return 0;             // norm >= 0 is never < a non-positive (or NaN) threshold
    const int n_all = n_exist + n_new;
    DeviceArena scratch(device);
    float* d_xyz;
    AS_ALLOC(d_xyz, float, (size_t)3 * n_all);
    if (n_exist) AS_TRY(hipMemcpyAsync(d_xyz, exist_xyz, sizeof(float) * 3 * (size_t)n_exist, hipMemcpyHostToDevice, s));
    AS_TRY(hipMemcpyAsync(d_xyz + 3 * (size_t)n_exist, new_xyz, sizeof(float) * 3 * (size_t)n_new, hipMemcpyHostToDevice, s));
    // cell edge a hair above the threshold: a pair that passes the float-difference test has a true coordinate difference
    // below max_dist (1 + 2^-24), so its cells differ by at most one along every axis (association)
    const double cell = (double)max_dist * (1.0 + 1.0 / 524288.0);
    const double inv_cell = 1.0 / cell;
    u64 *d_k0, *d_k1, *d_total;
    int *d_i0, *d_i1;
    AS_ALLOC(d_k0, u64, (size_t)n_all); AS_ALLOC(d_k1, u64, (size_t)n_all);
    AS_ALLOC(d_i0, int, (size_t)n_all); AS_ALLOC(d_i1, int, (size_t)n_all);
    AS_ALLOC(d_total, u64, 1);
    hipLaunchKernelGGL(k_cell_keys, dim3(grid_for(n_all)), dim3(256), 0, s, n_all, d_xyz, inv_cell, d_k0, d_i0);
    size_t tmp_bytes = 0;
    AS_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k0, d_k1, d_i0, d_i1, n_all, 0, 64, s));
    void* d_tmp = scratch.alloc(tmp_bytes ? tmp_bytes : 1);
    if (!d_tmp) return (int)hipErrorOutOfMemory;
    AS_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_k0, d_k1, d_i0, d_i1, n_all, 0, 64, s));
    AS_TRY(hipMemsetAsync(d_total, 0, sizeof(u64), s));
    const long long n_probe = (long long)n_new * 27;
    hipLaunchKernelGGL(k_radius_probe<true>, dim3(grid_for(n_probe)), dim3(256), 0, s, n_exist, n_new, d_xyz, inv_cell, (double)max_dist, n_all,
                       d_k1, d_i1, d_total, (u64*)nullptr, 0ull);
    AS_TRY(hipGetLastError());
    u64 h_total = 0;
    AS_TRY(hipMemcpyAsync(&h_total, d_total, sizeof(u64), hipMemcpyDeviceToHost, s));
    AS_TRY(hipStreamSynchronize(s));
    if (h_total >= (u64)INT_MAX) return ASSOC_ERR_TOO_LARGE;
    long long n_cand = (long long)h_total;
    long long* d_ptr;
    AS_ALLOC(d_ptr, long long, (size_t)n_new + 1);
    u64* d_sorted = nullptr;
    if (n_cand > 0) {
        u64 *d_c0, *d_c1;
        AS_ALLOC(d_c0, u64, (size_t)n_cand); AS_ALLOC(d_c1, u64, (size_t)n_cand);
        AS_TRY(hipMemsetAsync(d_total, 0, sizeof(u64), s));
        hipLaunchKernelGGL(k_radius_probe<false>, dim3(grid_for(n_probe)), dim3(256), 0, s, n_exist, n_new, d_xyz, inv_cell, (double)max_dist, n_all,
                           d_k1, d_i1, d_total, d_c0, (u64)n_cand);
        size_t sb = 0;
        AS_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sb, d_c0, d_c1, (int)n_cand, 0, 64, s));
        void* st = scratch.alloc(sb ? sb : 1);
        if (!st) return (int)hipErrorOutOfMemory;
        AS_TRY(hipcub::DeviceRadixSort::SortKeys(st, sb, d_c0, d_c1, (int)n_cand, 0, 64, s));
        // duplicates can only come from two neighbour cells colliding in the hashed key range: removed here
        long long* d_nsel;
        AS_ALLOC(d_nsel, long long, 1);
        size_t ub = 0;
        AS_TRY(hipcub::DeviceSelect::Unique(nullptr, ub, d_c1, d_c0, d_nsel, (int)n_cand, s));
        void* ut = scratch.alloc(ub ? ub : 1);
        if (!ut) return (int)hipErrorOutOfMemory;
        AS_TRY(hipcub::DeviceSelect::Unique(ut, ub, d_c1, d_c0, d_nsel, (int)n_cand, s));
        long long nsel = 0;
        AS_TRY(hipMemcpyAsync(&nsel, d_nsel, sizeof(long long), hipMemcpyDeviceToHost, s));
        AS_TRY(hipStreamSynchronize(s));
        n_cand = nsel;
        d_sorted = d_c0;
    }
    hipLaunchKernelGGL(k_cand_ptr, dim3(grid_for(n_new + 1)), dim3(256), 0, s, n_new, n_cand, d_sorted, d_ptr);
    AS_TRY(hipGetLastError());
    AS_TRY(hipMemcpyAsync(cand_ptr, d_ptr, sizeof(long long) * ((size_t)n_new + 1), hipMemcpyDeviceToHost, s));
    *total = n_cand;
    if (n_cand > cap) { AS_TRY(hipStreamSynchronize(s)); return ASSOC_ERR_CAPACITY; }
    if (n_cand > 0) {
        int* d_low;
        AS_ALLOC(d_low, int, (size_t)n_cand);
        hipLaunchKernelGGL(k_cand_low, dim3(grid_for(n_cand)), dim3(256), 0, s, n_cand, d_sorted, d_low);
        AS_TRY(hipMemcpyAsync(cand_idx, d_low, sizeof(int) * (size_t)n_cand, hipMemcpyDeviceToHost, s));
    }
    AS_TRY(hipStreamSynchronize(s));
    return 0;
}

}  // namespace sfmba
