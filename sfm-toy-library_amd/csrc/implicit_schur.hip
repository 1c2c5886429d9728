// implicit_schur.hip -- the CG product of the sharded solve formed WITHOUT the reduced matrix (options.shard_distributed_cg = 2), and the
// glue of the block-Jacobi transform the pair pass otherwise does on the way.  Replaces, for that form, what ceres::Solve's SchurEliminator
// + dense solve do behind SfMToyLib/SfMBundleAdjustmentUtils.cpp:171-179; DESIGN.md section 6 has the measured price.
#include "ba_common.h"

namespace sfmba {

// ------------------------------------------------------------------------------------------
// Implicit Schur product (sharded solve, options.shard_distributed_cg = 2): q~ = S~ p~ WITHOUT forming S~ --
// nothing of the reduced matrix is exchanged between the ranks, a rank applies its own points' W V^-1 W^T to the vector:
//   S_off y = - sum_points sum_{a != b} A~_a^T C_a C_b^T A~_b y_cam(b),      S~ = Lb^-1 S Lb^-T,  y = Lb^-T p~
// (the diagonal blocks and the focal border of S~ are known on every rank from exchange (A): identity, S~_jf).  Per product:
//   k_imp_dir     per camera: v = D Linv^T p~_j, the direction in the factored coordinates of sfmba_device.h (Q v_w, v_t) -- the layout of
//                 the step table's first two quads; clears the per-camera sums
//   k_imp_points  point-major, one lane per observation (the waves of the point passes): u = A v = P (Q v_w x X_g + v_t), w = C^T u,
//                 s_i = sum_obs w (through the wave's LDS) -> spt[i]
//   k_imp_cams    camera-major, one workgroup per chunk of a camera's observations (the chunks of k_cam_diag_f): e = C (s_i - w),
//                 h = P^T e, sums of X_g x h and h over the camera's observations -> acc[j] (six values per camera)
//   k_imp_out     per camera: q~_j = -Linv D [Q^T a; b] (+ on rank 0 the identity / focal part, as k_dcg_comb adds it)
// Every observation is evaluated twice per product, with the expressions of k_point_update; C = (P R) L~ in the precision of the
// Jacobian blocks.  Pairs of observations of ONE camera on a point (duplicates) live in the diagonal blocks: a problem that has them
// does not take this path (the caller checks ds.ndupwg).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_imp_dir(DeviceStructure ds, DeviceBuffers db, const double* __restrict__ pt, double* __restrict__ dtab,
                                                 double* __restrict__ acc, const int* __restrict__ flags) {
    if (flags && flags[0]) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ds.ncam) return;
    const int cur = db.st->cur;
    const double* Li = db.pcg_binv + (size_t)j * 36;
    double x[6], z[6], Q9[9];
#pragma unroll
    for (int t = 0; t < 6; ++t) x[t] = pt[6 * j + t];
#pragma unroll
    for (int e = 0; e < 9; ++e) Q9[e] = db.camtab[cur][cam_tab_index(CT_QD + e, j, ds.ncam)];
    const double small_cur = db.camtab[cur][cam_tab_index(CT_SMALL, j, ds.ncam)];
#pragma unroll
    for (int c = 0; c < 6; ++c) {          // z = Linv^T x (Linv lower triangular, row-major), then the Jacobi scales
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) if (t >= c) v += Li[t * 6 + c] * x[t];
        z[c] = v * db.cscale[6 * j + c];
    }
    double row[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int e = 0; e < 3; ++e) { row[ST_DQ + e] = Q9[3 * e] * z[0] + Q9[3 * e + 1] * z[1] + Q9[3 * e + 2] * z[2]; row[ST_DT + e] = z[3 + e]; }
    row[ST_SMALL] = small_cur;
#pragma unroll
    for (int e = 0; e < 8; ++e) dtab[cam_tab_index(e, j, ds.ncam)] = row[e];
#pragma unroll
    for (int e = 0; e < 6; ++e) acc[6 * j + e] = 0.0;
}

template <typename T>
__global__ PB_BOUNDS void k_imp_points(DeviceStructure ds, DeviceBuffers db, const double* __restrict__ dtab, double* __restrict__ spt, const int* __restrict__ flags) {
    if (flags && flags[0]) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gw = blockIdx.x * WPB + w;
    const LMState* st = db.st;
    const int cur = st->cur;
    const double focal = st->focal[cur];
    const double* tab = db.camtab[cur];
    const int sub = lane & (PB_LPP - 1);
    const int slot = gw * (64 / PB_LPP) + (lane / PB_LPP);        // four lanes per point, as the point passes
    const bool have = slot < ds.npt;
    const int ip = have ? (ds.pt_order ? ds.pt_order[slot] : slot) : 0;
    const size_t i = (size_t)ip;
    const int q0 = have ? ds.pt_ptr[ip] : 0, q1 = have ? ds.pt_ptr[ip + 1] : 0;
    const PtRecA<T> pa = load_ptrec(reinterpret_cast<const PtRecA<T>*>(db.PA) + i);
    double s[3] = { 0, 0, 0 };
    int q = q0 + sub;
    int j_next = q < q1 ? ds.obs_cam[q] : 0;
    while (__any(q < q1)) {
        const bool act = q < q1;
        const int j = j_next;
        const CamRow ct = { tab + 4 * (size_t)(j), ds.ncam };
        const CamRow drw = { dtab + 4 * (size_t)(j), ds.ncam };
        double Rt[12], dr[8];
#pragma unroll
        for (int e = 0; e < 12; ++e) Rt[e] = ct[CT_R + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) dr[e] = drw[e];
        q += PB_LPP;
        if (q < q1) j_next = ds.obs_cam[q];
        if (act) {
            ImpObs o; T C[6];
            imp_eval<T>(Rt, dr, focal, pa, o, C);
#pragma unroll
            for (int c = 0; c < 3; ++c) s[c] += (double)C[c] * o.u[0] + (double)C[3 + c] * o.u[1];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { s[c] = xlane_add<1>(s[c]); s[c] = xlane_add<2>(s[c]); }
    if (have && sub == 0) { spt[3 * i] = s[0]; spt[3 * i + 1] = s[1]; spt[3 * i + 2] = s[2]; }
}

template <typename T>
__global__ __launch_bounds__(CD_BLK) void k_imp_cams(DeviceStructure ds, DeviceBuffers db, const double* __restrict__ dtab, const double* __restrict__ spt, double* __restrict__ part,
                                                     double* __restrict__ acc, const int* __restrict__ flags) {
    __shared__ double red[CD_BLK / 64][6];
    if (flags && flags[0]) return;
    const int chunk_id = ds.chunk_order[blockIdx.x];
    const int4 ch = ds.chunks[chunk_id];
    const int j = ch.x;
    const LMState* st = db.st;
    const int cur = st->cur;
    const double focal = st->focal[cur];
    CamRegs ct;
    load_cam_regs(db.camtab[cur], j, ds.ncam, ct);
    double dr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = dtab[cam_tab_index(e, j, ds.ncam)];
    const PtRecA<T>* PA = reinterpret_cast<const PtRecA<T>*>(db.PA);
    double v[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll 1
    for (int e = ch.y + threadIdx.x; e < ch.z; e += CD_BLK) {
        const int i = ds.cam_obs_pt[e];
        const PtRecA<T> pa = load_ptrec(PA + i);
        const double s0 = spt[3 * (size_t)i], s1 = spt[3 * (size_t)i + 1], s2 = spt[3 * (size_t)i + 2];
        ImpObs o; T C[6];
        imp_eval<T>(ct, dr, focal, pa, o, C);
        // everything the OTHER observations of the point contribute: s_i - w, then e = C (.), h = P^T e
        const double d0 = s0 - ((double)C[0] * o.u[0] + (double)C[3] * o.u[1]);
        const double d1 = s1 - ((double)C[1] * o.u[0] + (double)C[4] * o.u[1]);
        const double d2 = s2 - ((double)C[2] * o.u[0] + (double)C[5] * o.u[1]);
        const double e0 = (double)C[0] * d0 + (double)C[1] * d1 + (double)C[2] * d2;
        const double e1 = (double)C[3] * d0 + (double)C[4] * d1 + (double)C[5] * d2;
        const double h0 = o.fz * e0, h1 = o.fz * e1, h2 = -o.fz * (o.xp * e0 + o.yp * e1);
        v[0] += o.xg[1] * h2 - o.xg[2] * h1;
        v[1] += o.xg[2] * h0 - o.xg[0] * h2;
        v[2] += o.xg[0] * h1 - o.xg[1] * h0;
        v[3] += h0; v[4] += h1; v[5] += h2;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = wave_allsum(v[k]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) red[w][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < CD_BLK / 64; ++ww) t += red[ww][threadIdx.x];
        // deterministic handles: the chunk's own slot, added in chunk order by k_imp_out (the order fp64 atomics arrive in would show in the CG's iterates)
        if (part) part[(size_t)chunk_id * 6 + threadIdx.x] = t;
        else atomicAdd(&acc[6 * j + threadIdx.x], t);
    }
}

// FF: the type the focal row of S~ is stored in (the CG's matrix: fp64, or fp32 on the streaming path)
template <typename FF>
__global__ __launch_bounds__(1024) void k_imp_out(DeviceStructure ds, DeviceBuffers db, int rank, const double* __restrict__ pt, const double* __restrict__ acc, const double* __restrict__ part,
                                                  const FF* __restrict__ focal_row, double* __restrict__ out, const int* __restrict__ flags) {
    if (flags && flags[0]) return;
    __shared__ double sh[16];
    const int fo = ds.d - 1;
    const int cur = db.st->cur;
    double fdot = 0.0;
    if (rank == 0) { for (int i = threadIdx.x; i < fo; i += blockDim.x) fdot += (double)focal_row[i] * pt[i]; }
    fdot = wave_allsum(fdot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = fdot;
    __syncthreads();
    const double pf = pt[fo];
    for (int j = threadIdx.x; j < ds.ncam; j += blockDim.x) {
        double a[6], g[6], Q9[9];
#pragma unroll
        for (int e = 0; e < 6; ++e) a[e] = acc[6 * j + e];
        if (part) {
            for (int c = ds.cam_chunk_ptr[j]; c < ds.cam_chunk_ptr[j + 1]; ++c) {
#pragma unroll
                for (int e = 0; e < 6; ++e) a[e] += part[(size_t)c * 6 + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) Q9[e] = db.camtab[cur][cam_tab_index(CT_QD + e, j, ds.ncam)];
        // A^T e summed over the camera's observations in the factored coordinates: [Q^T (sum X_g x h); sum h], Jacobi scales, sign of S_off
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g[c] = -db.cscale[6 * j + c] * (Q9[c] * a[0] + Q9[3 + c] * a[1] + Q9[6 + c] * a[2]);
            g[3 + c] = -db.cscale[6 * j + 3 + c] * a[3 + c];
        }
        const double* Li = db.pcg_binv + (size_t)j * 36;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) if (c <= r) v += Li[r * 6 + c] * g[c];
            const int i = 6 * j + r;
            if (rank == 0) v += pt[i] + (double)focal_row[i] * pf;
            out[i] = v;
        }
    }
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
        out[fo] = rank == 0 ? pf + t : 0.0;
    }
}

void launch_implicit_product(hipStream_t s, const ImplicitProduct& ip, const double* p_tilde, double* out, const int* flags) {
    const DeviceStructure& ds = ip.ds;
    hipLaunchKernelGGL(k_imp_dir, dim3((ds.ncam + 255) / 256), dim3(256), 0, s, ds, ip.db, p_tilde, ip.dtab, ip.acc, flags);
    if (ip.f32) {
        hipLaunchKernelGGL(k_imp_points<float>, dim3((ds.npt + WPB * (64 / PB_LPP) - 1) / (WPB * (64 / PB_LPP))), dim3(PBK), 0, s, ds, ip.db, ip.dtab, ip.spt, flags);
        hipLaunchKernelGGL(k_imp_cams<float>, dim3(ds.nchunk), dim3(CD_BLK), 0, s, ds, ip.db, ip.dtab, ip.spt, ip.part, ip.acc, flags);
    } else {
        hipLaunchKernelGGL(k_imp_points<double>, dim3((ds.npt + WPB * (64 / PB_LPP) - 1) / (WPB * (64 / PB_LPP))), dim3(PBK), 0, s, ds, ip.db, ip.dtab, ip.spt, flags);
        hipLaunchKernelGGL(k_imp_cams<double>, dim3(ds.nchunk), dim3(CD_BLK), 0, s, ds, ip.db, ip.dtab, ip.spt, ip.part, ip.acc, flags);
    }
    if (ip.focal_row32) hipLaunchKernelGGL(k_imp_out<float>, dim3(1), dim3(1024), 0, s, ds, ip.db, ip.rank, p_tilde, ip.acc, ip.part, ip.focal_row32, out, flags);
    else hipLaunchKernelGGL(k_imp_out<double>, dim3(1), dim3(1024), 0, s, ds, ip.db, ip.rank, p_tilde, ip.acc, ip.part, ip.focal_row, out, flags);
}

// The glue of the block-Jacobi transform on its own (the pair pass does it on the way when it runs, k_schur_pairs MODE 1): focal row /
// column of S~, b~ = Lb^-1 rhs, 1 / sqrt(S_ff), and the post-linearisation bookkeeping that k_finalize(pcg = 1) leaves to its successor.
__global__ __launch_bounds__(256) void k_pcg_glue(DeviceStructure ds, DeviceBuffers db) {
    if (blockIdx.x == 0 && threadIdx.x < 64) post_linearisation(ds, db);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ds.ncam) return;
    const int fo = ds.d - 1, row0 = 6 * j;
    const double* Li = db.pcg_binv + (size_t)j * 36;
    const double linv_f = 1.0 / sqrt(db.S[(size_t)fo * ds.ld + fo]);
    for (int l = 0; l < 6; ++l) {
        double vf = 0.0, vb = 0.0;
        for (int a = 0; a <= l; ++a) { vf += Li[l * 6 + a] * db.S[(size_t)(row0 + a) * ds.ld + fo]; vb += Li[l * 6 + a] * db.rhs[row0 + a]; }
        vf *= linv_f;
        store_F(db, (size_t)(row0 + l) * ds.ld + fo, vf);
        store_F(db, (size_t)fo * ds.ld + row0 + l, vf);
        db.pcg_bt[row0 + l] = vb;
    }
    if (j == 0) {
        store_F(db, (size_t)fo * ds.ld + fo, 1.0);
        db.pcg_bt[fo] = db.rhs[fo] * linv_f;
        db.pcg_binv[(size_t)ds.ncam * 36] = linv_f;
    }
}
void launch_pcg_glue(hipStream_t s, const DeviceStructure& ds, const DeviceBuffers& db) {
    hipLaunchKernelGGL(k_pcg_glue, dim3((ds.ncam + 255) / 256), dim3(256), 0, s, ds, db);
}

}  // namespace sfmba
