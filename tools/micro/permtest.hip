// permtest.hip -- hardware check of the VALU-only cross-lane exchanges csrc/sfmba_device.h builds every wave reduction on:
// DPP quad permutes (lane ^ 1, lane ^ 2), row_half_mirror (lane ^ 7), row_ror:8 (lane ^ 8) and v_permlane16_swap / v_permlane32_swap
// for the two top levels.  Every helper is compared, lane by lane, with what ds_bpermute (__shfl_xor / __shfl) gives.
//   hipcc --offload-arch=gfx950 -O2 -o permtest permtest.hip && ./permtest        (exit code 0 = all exchanges behave as documented)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../sfm-toy-library_amd/csrc/sfmba_device.h"

using namespace sfmba;

// out[0..5][lane]: partner's value for OFF = 1, 2, 4, 8, 16, 32;  out[6..11]: xlane_add;  out[12], out[13]: xlane_pairsum 16 / 32;
// out[14]: wave_allsum;  the double versions follow at +16
__global__ void k_perm(const float* in, const float* in2, float* outf, double* outd) {
    const int lane = threadIdx.x;
    const float v = in[lane], w = in2[lane];
    outf[0 * 64 + lane] = xlane_get<1>(v);  outf[1 * 64 + lane] = xlane_get<2>(v);  outf[2 * 64 + lane] = xlane_get<4>(v);
    outf[3 * 64 + lane] = xlane_get<8>(v);  outf[4 * 64 + lane] = xlane_get<16>(v); outf[5 * 64 + lane] = xlane_get<32>(v);
    outf[6 * 64 + lane] = xlane_add<1>(v);  outf[7 * 64 + lane] = xlane_add<2>(v);  outf[8 * 64 + lane] = xlane_add<4>(v);
    outf[9 * 64 + lane] = xlane_add<8>(v);  outf[10 * 64 + lane] = xlane_add<16>(v); outf[11 * 64 + lane] = xlane_add<32>(v);
    outf[12 * 64 + lane] = xlane_pairsum<16>(v, w); outf[13 * 64 + lane] = xlane_pairsum<32>(v, w);
    outf[14 * 64 + lane] = wave_allsum(v);
    const double dv = (double)v * 1.000000123, dw = (double)w * 0.999999871;
    outd[0 * 64 + lane] = xlane_get<1>(dv);  outd[1 * 64 + lane] = xlane_get<2>(dv);  outd[2 * 64 + lane] = xlane_get<4>(dv);
    outd[3 * 64 + lane] = xlane_get<8>(dv);  outd[4 * 64 + lane] = xlane_get<16>(dv); outd[5 * 64 + lane] = xlane_get<32>(dv);
    outd[6 * 64 + lane] = xlane_add<1>(dv);  outd[7 * 64 + lane] = xlane_add<2>(dv);  outd[8 * 64 + lane] = xlane_add<4>(dv);
    outd[9 * 64 + lane] = xlane_add<8>(dv);  outd[10 * 64 + lane] = xlane_add<16>(dv); outd[11 * 64 + lane] = xlane_add<32>(dv);
    outd[12 * 64 + lane] = xlane_pairsum<16>(dv, dw); outd[13 * 64 + lane] = xlane_pairsum<32>(dv, dw);
    outd[14 * 64 + lane] = wave_allsum(dv);
}

int main() {
    float h_in[64], h_in2[64];
    for (int l = 0; l < 64; ++l) { h_in[l] = 1.0f + 0.37f * l + 0.001f * l * l; h_in2[l] = -3.0f + 0.11f * l; }
    float *d_in, *d_in2, *d_of; double* d_od;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess) { std::printf("no device\n"); return 2; }
    (void)hipMalloc(&d_in2, sizeof(h_in2)); (void)hipMalloc(&d_of, 15 * 64 * sizeof(float)); (void)hipMalloc(&d_od, 15 * 64 * sizeof(double));
    (void)hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice); (void)hipMemcpy(d_in2, h_in2, sizeof(h_in2), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_perm, dim3(1), dim3(64), 0, 0, d_in, d_in2, d_of, d_od);
    static float of[15 * 64]; static double od[15 * 64];
    if (hipMemcpy(of, d_of, sizeof(of), hipMemcpyDeviceToHost) != hipSuccess) { std::printf("kernel failed\n"); return 2; }
    (void)hipMemcpy(od, d_od, sizeof(od), hipMemcpyDeviceToHost);
    const int offs[6] = { 1, 2, 4, 8, 16, 32 };
    int bad = 0;
    auto partner = [](int lane, int off) { return off == 4 ? lane ^ 7 : lane ^ off; };     // "partner(OFF)" of sfmba_device.h
    double tot = 0.0, totd = 0.0;
    for (int l = 0; l < 64; ++l) { tot += h_in[l]; totd += (double)h_in[l] * 1.000000123; }
    for (int l = 0; l < 64; ++l) {
        for (int k = 0; k < 6; ++k) {
            const int q = partner(l, offs[k]);
            const double dv = (double)h_in[l] * 1.000000123, dq = (double)h_in[q] * 1.000000123;
            if (of[k * 64 + l] != h_in[q]) { ++bad; std::printf("get<%d> lane %d: %g want %g\n", offs[k], l, of[k * 64 + l], h_in[q]); }
            if (of[(6 + k) * 64 + l] != h_in[l] + h_in[q]) { ++bad; std::printf("add<%d> lane %d\n", offs[k], l); }
            if (od[k * 64 + l] != dq) { ++bad; std::printf("get<%d> (double) lane %d\n", offs[k], l); }
            if (std::abs(od[(6 + k) * 64 + l] - (dv + dq)) > 4e-16 * std::abs(dv + dq)) {      /* (the device may fuse the scaling into the add) */ ++bad; std::printf("add<%d> (double) lane %d\n", offs[k], l); }
        }
        for (int k = 0; k < 2; ++k) {       // pairsum<OFF>(lo, hi): lanes with bit OFF clear get lo[lane] + lo[partner], the others hi[lane] + hi[partner]
            const int off = k ? 32 : 16, q = l ^ off;
            const float* src = (l & off) ? h_in2 : h_in;
            if (of[(12 + k) * 64 + l] != src[l] + src[q]) { ++bad; std::printf("pairsum<%d> lane %d: %g want %g\n", off, l, of[(12 + k) * 64 + l], src[l] + src[q]); }
            const double sc = (l & off) ? 0.999999871 : 1.000000123;
            if (std::abs(od[(12 + k) * 64 + l] - ((double)src[l] * sc + (double)src[q] * sc)) > 4e-16 * (std::abs((double)src[l]) + std::abs((double)src[q]))) { ++bad; std::printf("pairsum<%d> (double) lane %d\n", off, l); }
        }
        if (std::abs(of[14 * 64 + l] - (float)tot) > 1e-3f * (float)tot) { ++bad; std::printf("allsum lane %d: %g want %g\n", l, of[14 * 64 + l], tot); }
        if (std::abs(od[14 * 64 + l] - totd) > 1e-12 * totd) { ++bad; std::printf("allsum (double) lane %d\n", l); }
    }
    std::printf(bad ? "permtest: %d mismatches\n" : "permtest: all cross-lane exchanges behave as sfmba_device.h documents (%d mismatches)\n", bad);
    return bad ? 1 : 0;
}
