// sfmba_device.h -- device-side camera model shared by the bundle-adjustment kernels (gfx950).
//
// Model (reference SfMToyLib/SfMBundleAdjustmentUtils.cpp:58-97, SimpleReprojectionError):
//   p = Rot(w) X + t ;  r = f * (p.x/p.z, p.y/p.z) - obs
// with ceres::AngleAxisRotatePoint semantics for Rot(w) [Ceres-upstream]: exact Rodrigues when
// theta^2 > DBL_EPSILON, first-order X + w x X otherwise.
// The reference differentiates this functor with Jet<double,10>; here the 2x6 / 2x3 / 2x1 blocks
// are analytic:
//   Aproj = (f/pz) [[1,0,-xp],[0,1,-yp]]          (xp = px/pz, yp = py/pz)
//   d r/d t = Aproj ; d r/d X = Aproj R ; d r/d f = (xp, yp)
//   d r/d w = Aproj G,  G = -R [X]x K'   with K' = (w w^T + (R^T - I)[w]x)/theta^2   (theta^2 > eps)
//                       G = -[X]x                                                    (theta^2 <= eps)
// K' depends on the camera only, so it is tabulated per camera together with R and t.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace sfmba {

// ---- per-camera table (doubles), rebuilt by k_cam_setup whenever camera parameters change ----
constexpr int CT_R = 0;        // R[9] row-major
constexpr int CT_T = 9;        // t[3]
constexpr int CT_K = 12;       // K'[9] row-major (identity when small-angle)
constexpr int CT_SMALL = 21;   // 1.0 if theta^2 <= DBL_EPSILON
constexpr int CT_SCALE = 22;   // Jacobi column scale of the 6 camera parameters
constexpr int CT_QD = 28;      // Q = R K' (3 x 3 row-major; I when small-angle): the rotation part of the pair pass's camera factor E
constexpr int CT_CQ = 37;      // coefficient of [w]x^2 in the inverse right Jacobian of SO(3) (gauge vectors of the two-level CG preconditioner)
constexpr int CT_STRIDE = 40;  // 38 used; whole component quads

// ---- per-camera step table used by the back-substitution / trial-point kernel (component quads like the camera tables) ----
constexpr int ST_DQ = 0;       // Q dw (3): the unscaled rotation step through Q = R K' of the linearisation point (I on the first-order branch);
                               //   the step is SUBTRACTED: trial = current - step
constexpr int ST_DT = 3;       // dt[3]: unscaled translation step
constexpr int ST_SMALL = 6;    // 1.0 if the camera is on the first-order branch at the linearisation point   (quads 0, 1; one pad value)
constexpr int ST_RN = 8;       // R[9] at the trial point
constexpr int ST_TN = 17;      // t[3] at the trial point                     (quads 2, 3, 4)
constexpr int ST_STRIDE = 20;

// The blocks of one observation as the camera pass (and the duplicate-pair pass) evaluate them (obs_record below): 16 values.  Rounds
// 1 - 3 STORED this record per observation (64 B in fp32); since round 4 it only ever lives in registers.
constexpr int YREC = 16;

// ---- per-point table (round 3): what the reduced-system passes need about a POINT to RE-EVALUATE the blocks of one of its
// observations instead of gathering a 64-byte record per observation (the records of BASELINE config 3 are 64 MB gathered ~10 times
// at random; this table is 4.8 MB and stays in every XCD's L2).  Written by k_point_build once per linearisation.
//   PtRecA: X (3 doubles: the projection is evaluated in fp64, exactly as the point pass does) | L = L^-1 diag(s_p) (6 values T:
//           l00 l10 l11 l20 l21 l22 -- C = B L^T row by row)                    64 B (48 + pad: one sector, so that the four lanes of
//           a quad can fetch it with ONE request) in fp32 mode, 80 B (72 + pad) in fp64 mode
//   PtRecB: t = L^-1 b_p (3 T) | y_f = L^-1 E_f (3 T)                           (camera-diagonal pass only)
template <typename T> struct PtRecA;
template <> struct alignas(64) PtRecA<float>  { double X[3]; float L[6]; float pad[4]; };      // one 64-byte sector per point
template <> struct alignas(16) PtRecA<double> { double X[3]; double L[6]; double pad; };
template <typename T> struct PtRecB;
template <> struct alignas(8)  PtRecB<float>  { float t[3]; float yf[3]; };
template <> struct alignas(16) PtRecB<double> { double t[3]; double yf[3]; };
// One entry in three (fp32 mode) 16-byte loads: left to itself the compiler loads the two member arrays separately (16 + 8 and, at an
// 8-byte-aligned offset, 16 + 8 bytes: four requests per lane, and the vector-memory pipe walks one line per lane and request).
__device__ __forceinline__ PtRecA<float> load_ptrec(const PtRecA<float>* p) {
    const int4* q = reinterpret_cast<const int4*>(p);
    const int4 c0 = q[0], c1 = q[1], c2 = q[2];
    PtRecA<float> r;
    r.X[0] = __hiloint2double(c0.y, c0.x); r.X[1] = __hiloint2double(c0.w, c0.z); r.X[2] = __hiloint2double(c1.y, c1.x);
    r.L[0] = __int_as_float(c1.z); r.L[1] = __int_as_float(c1.w);
    r.L[2] = __int_as_float(c2.x); r.L[3] = __int_as_float(c2.y); r.L[4] = __int_as_float(c2.z); r.L[5] = __int_as_float(c2.w);
    return r;
}
__device__ __forceinline__ PtRecA<double> load_ptrec(const PtRecA<double>* p) { return *p; }
static_assert(sizeof(PtRecA<float>) == 64 && sizeof(PtRecA<double>) == 80 && sizeof(PtRecB<float>) == 24 && sizeof(PtRecB<double>) == 48, "point table layout");

// Camera tables are stored in component quads (AoSoA): values 4c .. 4c+3 of camera j are the 32 contiguous bytes at
// tab[(c * ncam + j) * 4].  A wave whose lanes need the same components of 64 different cameras then issues ONE 32-byte
// load per lane and quad instead of four 8-byte loads (the vector-memory pipe counts lines per 16-lane group per
// instruction), while keeping the cameras of one component quad next to each other.
__host__ __device__ __forceinline__ size_t cam_tab_index(int k, int j, int ncam) {
    return ((size_t)(k >> 2) * ncam + j) * 4 + (k & 3);
}
struct CamRow {
    const double* base;   // tab + 4 * j
    int stride;           // ncam
    __device__ __forceinline__ double operator[](int k) const { return base[(size_t)(k >> 2) * stride * 4 + (k & 3)]; }
};

// The table row of ONE camera held in registers: the block passes work on one camera (pair pass: two) per wave, so the row is
// loaded with a wave-uniform index -- scalar loads into SGPRs, no per-lane gather.
struct CamRegs {
    double v[CT_SCALE];     // R, t, K', small-angle flag (the Jacobi scales are applied by the consumers themselves)
    __device__ __forceinline__ double operator[](int k) const { return v[k]; }
};
__device__ __forceinline__ void load_cam_regs(const double* __restrict__ tab, int j_uniform, int ncam, CamRegs& c) {
#pragma unroll
    for (int k = 0; k < CT_SCALE; ++k) c.v[k] = tab[cam_tab_index(k, j_uniform, ncam)];
}

// table entry k as T (converted on the spot)
template <typename T, typename CamPtr> __device__ __forceinline__ T cam_val(const CamPtr& cam, int k) { return (T)cam[k]; }
__device__ __forceinline__ float to_uniform(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double to_uniform(double x) { return x; }       // (already scalar: loaded with a uniform index)

template <typename T> struct ObsXY;
template <> struct ObsXY<float>  { typedef float2 type; };
template <> struct ObsXY<double> { typedef double2 type; };

// ---- cross-lane exchange without the LDS pipe ----
// __shfl_xor compiles to ds_bpermute_b32: measured 24 cycles of SIMD time per shuffle with every wave shuffling (tools/micro/pk_bench.hip;
// a plain VALU instruction: 4-5), i.e. the 38 shuffles of a 36-value halving butterfly were ~10 % of a pair-pass wave and the 94 of the
// camera pass's 24 doubles more than a third of it.  gfx950 has everything needed in the VALU: DPP quad permutes (lane ^ 1, lane ^ 2),
// row_half_mirror (lane ^ 7: a partner across bit 2, which is all a butterfly needs), row_ror:8 (lane ^ 8 inside a row of 16) and the
// v_permlane16_swap / v_permlane32_swap pair for the two top levels (verified on the hardware: tools/micro/permtest).
// "partner(OFF)" below: lane ^ OFF for OFF = 1, 2, 8, 16, 32 and lane ^ 7 for OFF = 4.
template <int OFF> __device__ __forceinline__ int xlane_get_i(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "butterfly level");
    if (OFF == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
    if (OFF == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
    if (OFF == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);      // row_half_mirror
    if (OFF == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);      // row_ror:8
    const bool up = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & OFF) != 0;
    if (OFF == 16) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); return (int)(up ? r[0] : r[1]); }
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(up ? r[0] : r[1]);
}
template <int OFF> __device__ __forceinline__ float xlane_get(float v) { return __int_as_float(xlane_get_i<OFF>(__float_as_int(v))); }
template <int OFF> __device__ __forceinline__ double xlane_get(double v) {
    return __hiloint2double(xlane_get_i<OFF>(__double2hiint(v)), xlane_get_i<OFF>(__double2loint(v)));
}
// v[lane] + v[partner(OFF)].  The two top levels need no select at all: after swap(A, B) the two registers hold, lane by lane, the
// lane's own value and its partner's (in one order or the other) -- their sum is the same either way.
template <int OFF> __device__ __forceinline__ float xlane_add(float v) {
    if (OFF >= 16) {
        const unsigned b = (unsigned)__float_as_int(v);
        if (OFF == 16) { const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false); return __int_as_float((int)r[0]) + __int_as_float((int)r[1]); }
        const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
    }
    return v + xlane_get<OFF>(v);
}
template <int OFF> __device__ __forceinline__ double xlane_add(double v) {
    if (OFF >= 16) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        if (OFF == 16) {
            const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
        }
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v + xlane_get<OFF>(v);
}
// Halving step of the two top levels: lanes with bit OFF clear end up with lo[lane] + lo[partner], the others with hi[lane] + hi[partner]
// -- ONE swap and one add per value pair (the selects of the generic step are done by the swap itself).
template <int OFF> __device__ __forceinline__ float xlane_pairsum(float lo, float hi) {
    static_assert(OFF == 16 || OFF == 32, "swap levels");
    const unsigned a = (unsigned)__float_as_int(lo), b = (unsigned)__float_as_int(hi);
    if (OFF == 16) { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); return __int_as_float((int)r[0]) + __int_as_float((int)r[1]); }
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
}
template <int OFF> __device__ __forceinline__ double xlane_pairsum(double lo, double hi) {
    static_assert(OFF == 16 || OFF == 32, "swap levels");
    const unsigned al = (unsigned)__double2loint(lo), ah = (unsigned)__double2hiint(lo), bl = (unsigned)__double2loint(hi), bh = (unsigned)__double2hiint(hi);
    if (OFF == 16) {
        const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false), h = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
        return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false), h = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
// the sum over the wave, in every lane
template <typename V> __device__ __forceinline__ V wave_allsum(V v) {
    v = xlane_add<32>(v); v = xlane_add<16>(v); v = xlane_add<8>(v); v = xlane_add<4>(v); v = xlane_add<2>(v); v = xlane_add<1>(v);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_allsum(v); }

// Block-wide sum of `v`; result valid in thread 0.  `scratch` holds >= blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) s += scratch[i];
    }
    return s;
}

// atomic max on non-negative doubles through their bit pattern (monotone for x >= 0; NaN maps high)
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (v != v) bits = 0x7ff8000000000000ull;
    atomicMax(reinterpret_cast<unsigned long long*>(addr), bits);
}

// fp64 reciprocal / reciprocal square root: hardware estimate + two Newton steps (full double accuracy).  The
// compiler's generic expansions of 1.0/x and sqrt(x) are ~20-deep dependent fp64 chains; a dependent DFMA costs
// ~32 cycles on gfx950 (tools/micro/microbench.hip), and these chains sit on the critical path of latency-bound waves.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double fast_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// Projection in fp64: p = R X + t, returns xp, yp, 1/pz.  `cam` points at a camera table row.
struct Proj {
    double xp, yp, iz;
};

template <typename CamPtr>
__device__ __forceinline__ Proj project_point(CamPtr cam, int roff, int toff, const double X[3]) {
    const double px = cam[roff + 0] * X[0] + cam[roff + 1] * X[1] + cam[roff + 2] * X[2] + cam[toff + 0];
    const double py = cam[roff + 3] * X[0] + cam[roff + 4] * X[1] + cam[roff + 5] * X[2] + cam[toff + 1];
    const double pz = cam[roff + 6] * X[0] + cam[roff + 7] * X[1] + cam[roff + 8] * X[2] + cam[toff + 2];
    Proj pr;
    pr.iz = fast_rcp(pz);      // pz == 0 gives inf/NaN exactly like the division (caught by the finiteness checks)
    pr.xp = px * pr.iz;
    pr.yp = py * pr.iz;
    return pr;
}

// Unscaled point block B = Aproj R (2x3) in precision T.
template <typename T, typename CamPtr>
__device__ __forceinline__ void point_block(CamPtr cam, const Proj& pr, double focal, T B[6]) {
    const T fz = (T)(focal * pr.iz), xp = (T)pr.xp, yp = (T)pr.yp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T r0 = cam_val<T>(cam, CT_R + c), r1 = cam_val<T>(cam, CT_R + 3 + c), r2 = cam_val<T>(cam, CT_R + 6 + c);
        B[c] = fz * (r0 - xp * r2);
        B[3 + c] = fz * (r1 - yp * r2);
    }
}

// Unscaled camera block A (2x6) = [Aproj G | Aproj] given B = Aproj R (G = -R [X]x K'  =>  Aproj G = -B [X]x K').
template <typename T, typename CamPtr>
__device__ __forceinline__ void camera_block(CamPtr cam, const Proj& pr, double focal, const double Xd[3],
                                             const T B[6], T A[12]) {
    const T fz = (T)(focal * pr.iz), xp = (T)pr.xp, yp = (T)pr.yp;
    const T X0 = (T)Xd[0], X1 = (T)Xd[1], X2 = (T)Xd[2];
    if (cam[CT_SMALL] != 0.0) {
        // G = -[X]x ;  Aproj G, Aproj = fz [[1,0,-xp],[0,1,-yp]] ; [X]x = [[0,-X2,X1],[X2,0,-X0],[-X1,X0,0]]
        // row0 of Aproj [X]x = fz * ( -xp*(-X1) , -X2 - xp*X0 , X1 )   -> careful expansion below
        const T a00 = fz, a02 = -fz * xp, a11 = fz, a12 = -fz * yp;
        // (Aproj [X]x)[r][c] = sum_m Aproj[r][m] [X]x[m][c]
        const T m00 = a02 * (-X1), m01 = a00 * (-X2) + a02 * X0, m02 = a00 * X1;
        const T m10 = a11 * X2 + a12 * (-X1), m11 = a12 * X0, m12 = a11 * (-X0);
        A[0] = -m00; A[1] = -m01; A[2] = -m02;
        A[6] = -m10; A[7] = -m11; A[8] = -m12;
    } else {
        // M = [X]x K' : column c = X x K'[:,c]
        T M[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const T k0 = cam_val<T>(cam, CT_K + c), k1 = cam_val<T>(cam, CT_K + 3 + c), k2 = cam_val<T>(cam, CT_K + 6 + c);
            M[c] = X1 * k2 - X2 * k1;
            M[3 + c] = X2 * k0 - X0 * k2;
            M[6 + c] = X0 * k1 - X1 * k0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[c] = -(B[0] * M[c] + B[1] * M[3 + c] + B[2] * M[6 + c]);
            A[6 + c] = -(B[3] * M[c] + B[4] * M[3 + c] + B[5] * M[6 + c]);
        }
    }
    A[3] = fz;   A[4] = (T)0; A[5] = -fz * xp;
    A[9] = (T)0; A[10] = fz;  A[11] = -fz * yp;
}

// The packed record of one observation (layout: ba_kernels.hip, "Packed per-observation record") from the camera's table row, the
// point and the point's L -- the SAME expressions, in the same order, as the record sweep of k_point_build, so that a pass which
// re-evaluates an observation sees bit for bit what a pass reading the stored record sees.  rec[15] (camera slot) is not set.
template <typename T, typename CamPtr>
__device__ __forceinline__ void obs_record(CamPtr cam, double focal, const double X[3], const T L[6], T rec[YREC]) {
    const Proj pr = project_point(cam, CT_R, CT_T, X);
    T B[6], A[12];
    point_block<T>(cam, pr, focal, B);
    camera_block<T>(cam, pr, focal, X, B, A);
    rec[0] = A[0]; rec[1] = A[1]; rec[2] = A[2]; rec[3] = A[6]; rec[4] = A[7]; rec[5] = A[8];
    rec[6] = (T)(focal * pr.iz); rec[7] = (T)pr.xp; rec[8] = (T)pr.yp;
    const T l00 = L[0], l10 = L[1], l11 = L[2], l20 = L[3], l21 = L[4], l22 = L[5];
#pragma unroll
    for (int r = 0; r < 2; ++r) {   // C = B~ L^-T
        const T b0 = B[3 * r], b1 = B[3 * r + 1], b2 = B[3 * r + 2];
        rec[9 + 3 * r + 0] = b0 * l00;
        rec[9 + 3 * r + 1] = b0 * l10 + b1 * l11;
        rec[9 + 3 * r + 2] = b0 * l20 + b1 * l21 + b2 * l22;
    }
    rec[15] = (T)0;
}

// ---- factored form of an observation for the pair pass -------------------------------------------------------------------------
// The camera block of an observation factors as  A = P [ -[X_g]x | I ] diag(Q, I)  with P = (f / p_z) [[1, 0, -x_p], [0, 1, -y_p]],
// X_g = R X and Q = R K' (A_w = -P R [X]x K' = -P [R X]x R K': R is orthogonal), or X_g = X and Q = I for a camera on the first-order
// branch (theta^2 <= eps: G = -[X]x exactly).  diag(Q, I) is constant over a 6x6 block of the reduced matrix, so the
// pair pass sums the blocks in the factored coordinates and applies Q_i^T . Q_j ONCE per block (k_schur_pairs epilogue):
//   sum_pairs A_a^T (C_a C_b^T) A_b = E_i^T [ sum_pairs G_a^T N G_b ] E_j,   N = P_a^T (C_a C_b^T) P_b  (3 x 3),  G = [ -[X_g]x | I ].
// Per observation that leaves the projection, B = P R and C = B L^T (24 fp32 operations instead of 60: no [X]x K', no B ([X]x K')).
//   g[0..2] X_g | g[3] f / p_z | g[4] x_p | g[5] y_p | g[6..11] C (2 x 3, row-major)
constexpr int GREC = 12;
template <typename T>
struct CamG {
    double v[CT_K];          // R (9), t (3): the projection is evaluated in fp64 like everywhere else
    T vt[9];                 // (T) R, wave-uniform
    double small;
};
template <typename T>
__device__ __forceinline__ void load_cam_g(const double* __restrict__ tab, int j_uniform, int ncam, CamG<T>& c) {
#pragma unroll
    for (int k = 0; k < CT_K; ++k) c.v[k] = tab[cam_tab_index(k, j_uniform, ncam)];
#pragma unroll
    for (int k = 0; k < 9; ++k) c.vt[k] = to_uniform((T)c.v[k]);
    c.small = tab[cam_tab_index(CT_SMALL, j_uniform, ncam)];
}
template <typename T>
__device__ __forceinline__ void obs_factored(const CamG<T>& cam, double focal, const double X[3], const T L[6], T g[GREC]) {
    const double rx = cam.v[CT_R + 0] * X[0] + cam.v[CT_R + 1] * X[1] + cam.v[CT_R + 2] * X[2];
    const double ry = cam.v[CT_R + 3] * X[0] + cam.v[CT_R + 4] * X[1] + cam.v[CT_R + 5] * X[2];
    const double rz = cam.v[CT_R + 6] * X[0] + cam.v[CT_R + 7] * X[1] + cam.v[CT_R + 8] * X[2];
    const double iz = fast_rcp(rz + cam.v[CT_T + 2]);
    const double xpd = (rx + cam.v[CT_T + 0]) * iz, ypd = (ry + cam.v[CT_T + 1]) * iz;
    const bool first_order = cam.small != 0.0;           // wave-uniform
    g[0] = (T)(first_order ? X[0] : rx); g[1] = (T)(first_order ? X[1] : ry); g[2] = (T)(first_order ? X[2] : rz);
    const T fz = (T)(focal * iz), xp = (T)xpd, yp = (T)ypd;
    g[3] = fz; g[4] = xp; g[5] = yp;
    T B[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {                          // B = P R, the expressions of point_block
        B[c] = fz * (cam.vt[c] - xp * cam.vt[6 + c]);
        B[3 + c] = fz * (cam.vt[3 + c] - yp * cam.vt[6 + c]);
    }
    const T l00 = L[0], l10 = L[1], l11 = L[2], l20 = L[3], l21 = L[4], l22 = L[5];
#pragma unroll
    for (int r = 0; r < 2; ++r) {                          // C = B~ L^-T, as obs_record
        const T b0 = B[3 * r], b1 = B[3 * r + 1], b2 = B[3 * r + 2];
        g[6 + 3 * r + 0] = b0 * l00;
        g[6 + 3 * r + 1] = b0 * l10 + b1 * l11;
        g[6 + 3 * r + 2] = b0 * l20 + b1 * l21 + b2 * l22;
    }
}
// The same with the PROJECTION in the precision of the Jacobian blocks as well (small-block pair pass in fp32-Jacobian mode: every value
// the pair product uses is rounded to T anyway, the pass needs no residual, and a camera row held per lane is 12 registers instead of 24;
// measured at BASELINE config 5: 468 -> 431 us per launch.  The wave-per-chunk pass keeps the fp64 projection: with its two rows in
// scalar registers the fp32 form is SLOWER, 62 -> 72 us).  Rt: R (9, row-major) and t (3) in T.
template <typename T>
__device__ __forceinline__ void obs_factored_t(const T (&Rt)[12], bool first_order, T focal, T X0, T X1, T X2, const T L[6], T g[GREC]) {
    const T rx = Rt[0] * X0 + Rt[1] * X1 + Rt[2] * X2;
    const T ry = Rt[3] * X0 + Rt[4] * X1 + Rt[5] * X2;
    const T rz = Rt[6] * X0 + Rt[7] * X1 + Rt[8] * X2;
    const T pz = rz + Rt[11];
    T iz;
    if constexpr (sizeof(T) == 4) { iz = __builtin_amdgcn_rcpf(pz); iz = iz * ((T)2 - pz * iz); }      // estimate + one Newton step
    else iz = (T)fast_rcp((double)pz);
    const T xp = (rx + Rt[9]) * iz, yp = (ry + Rt[10]) * iz;
    g[0] = first_order ? X0 : rx; g[1] = first_order ? X1 : ry; g[2] = first_order ? X2 : rz;
    const T fz = focal * iz;
    g[3] = fz; g[4] = xp; g[5] = yp;
    T B[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        B[c] = fz * (Rt[c] - xp * Rt[6 + c]);
        B[3 + c] = fz * (Rt[3 + c] - yp * Rt[6 + c]);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const T b0 = B[3 * r], b1 = B[3 * r + 1], b2 = B[3 * r + 2];
        g[6 + 3 * r + 0] = b0 * L[0];
        g[6 + 3 * r + 1] = b0 * L[1] + b1 * L[2];
        g[6 + 3 * r + 2] = b0 * L[3] + b1 * L[4] + b2 * L[5];
    }
}
// acc += G_a^T N G_b  (6 x 6 row-major; rows: camera a, columns: camera b; the first three of each are the rotation part)
//   = [[ -[X_a]x N [X_b]x ,  [X_a]x N ],  [ -N [X_b]x ,  N ]]   with  v [X]x = v x X  and  [X]x v = X x v
// (A: the type the lane's sums are kept in -- T, or double with the products widened one by one)
template <typename T, typename A = T>
__device__ __forceinline__ void pair_product_factored(const T ga[GREC], const T gb[GREC], A acc[36]) {
    const T ff = ga[3] * gb[3];
    const T m00 = ff * (ga[6] * gb[6] + ga[7] * gb[7] + ga[8] * gb[8]);
    const T m01 = ff * (ga[6] * gb[9] + ga[7] * gb[10] + ga[8] * gb[11]);
    const T m10 = ff * (ga[9] * gb[6] + ga[10] * gb[7] + ga[11] * gb[8]);
    const T m11 = ff * (ga[9] * gb[9] + ga[10] * gb[10] + ga[11] * gb[11]);
    const T xa = ga[4], ya = ga[5], xb = gb[4], yb = gb[5];
    T N[9];
    N[0] = m00; N[1] = m01; N[2] = -(m00 * xb + m01 * yb);
    N[3] = m10; N[4] = m11; N[5] = -(m10 * xb + m11 * yb);
    N[6] = -(xa * N[0] + ya * N[3]); N[7] = -(xa * N[1] + ya * N[4]); N[8] = -(xa * N[2] + ya * N[5]);
    const T a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
    T Tm[9];                                               // T = [X_a]x N: column c = X_a x N[:, c]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Tm[c] = a1 * N[6 + c] - a2 * N[3 + c];
        Tm[3 + c] = a2 * N[c] - a0 * N[6 + c];
        Tm[6 + c] = a0 * N[3 + c] - a1 * N[c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const T t0 = Tm[3 * r], t1 = Tm[3 * r + 1], t2 = Tm[3 * r + 2];
        // -(T[r, :] [X_b]x) = X_b x T[r, :]
        acc[6 * r + 0] += (A)(b1 * t2 - b2 * t1);
        acc[6 * r + 1] += (A)(b2 * t0 - b0 * t2);
        acc[6 * r + 2] += (A)(b0 * t1 - b1 * t0);
        acc[6 * r + 3] += (A)t0; acc[6 * r + 4] += (A)t1; acc[6 * r + 5] += (A)t2;
        const T n0 = N[3 * r], n1 = N[3 * r + 1], n2 = N[3 * r + 2];
        acc[6 * (3 + r) + 0] += (A)(b1 * n2 - b2 * n1);
        acc[6 * (3 + r) + 1] += (A)(b2 * n0 - b0 * n2);
        acc[6 * (3 + r) + 2] += (A)(b0 * n1 - b1 * n0);
        acc[6 * (3 + r) + 3] += (A)n0; acc[6 * (3 + r) + 4] += (A)n1; acc[6 * (3 + r) + 5] += (A)n2;
    }
}

}  // namespace sfmba
