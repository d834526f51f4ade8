// Issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 (independent accumulators / one dependent chain) at 1, 2 and
// 4 waves per SIMD, and the rate with fp64 VALU work interleaved (the diagonal-tile chains of the wave kernels, csrc/mf_wave.hpp,
// run beside the products).  Backs the cost model of DESIGN 4.13.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o scripts/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) rate(double* out, int iters) {
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    v4f f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
    double a = 1e-3 * threadIdx.x, b = 1.0000001, v0 = 1, v1 = 2, v2 = 3, v3 = 4;
    float af = (float)a, bf = (float)b;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // f64, 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if constexpr (MODE == 1) {   // f64, one dependent chain
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        } else if constexpr (MODE == 2) {   // f32, 4 independent
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f2, 0, 0, 0);
            f3 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f3, 0, 0, 0);
        } else if constexpr (MODE == 3) {   // f32, dependent chain
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f0, 0, 0, 0);
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f0, 0, 0, 0);
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f0, 0, 0, 0);
            f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, f0, 0, 0, 0);
        } else {                            // f64: 4 independent MFMAs + 8 independent fp64 FMAs between them
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            asm volatile("v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            asm volatile("v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            asm volatile("v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            asm volatile("v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
        }
        asm volatile("" : "+v"(a), "+v"(af));
    }
    double s = v0 + v1 + v2 + v3;
    for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e] + f0[e] + f1[e] + f2[e] + f3[e];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int waves_per_simd, double flop_per_mfma) {
    double* out; hipMalloc(&out, 8 * 64 * 8192);
    const int iters = 20000;
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE>), dim3(grid), dim3(64), 0, 0, out, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<MODE>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = double(iters) * 4;
    const double ns = ms * 1e6 / n / waves_per_simd;
    printf("%-44s waves/SIMD=%d: %8.3f ms  %.2f ns per MFMA per SIMD (= %.1f cycles @2.4GHz)  %.1f TFLOP/s chip\n", name, waves_per_simd,
           ms, ns, ns * 2.4, flop_per_mfma * 1024 / ns * 1e-3);
    hipFree(out);
}

// what the instruction computes: D[i][j] = sum_q a(lane(i, q)) b(lane(j, q)); accumulator register e of lane (r, q): which (row, col)?
__global__ void __launch_bounds__(64) layout(double* out) {
    const int r = threadIdx.x & 15, q = threadIdx.x >> 4;
    // A[i][k] = 1 + i + 100 k, B[k][j] = (k == 2) * (j + 1): D[i][j] = (1 + i + 200) (j + 1)
    const double a = 1.0 + r + 100.0 * q, b = (q == 2) ? double(r + 1) : 0.0;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = c[e];
    v4f cf = {0, 0, 0, 0};
    cf = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a, (float)b, cf, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[256 + threadIdx.x * 4 + e] = cf[e];
}

int main() {
    double* d; hipMalloc(&d, 512 * 8);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    double h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    bool ok64 = true, ok32 = true;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int r = l & 15, q = l >> 4;
            ok64 &= h[l * 4 + e] == (1.0 + (q + 4 * e) + 200.0) * (r + 1);          // f64: row q + 4 e, column r
            ok32 &= h[256 + l * 4 + e] == (1.0 + (4 * q + e) + 200.0) * (r + 1);    // f32: row 4 q + e, column r
        }
    printf("accumulator layout f64 (row q + 4 e, col r): %s   f32 (row 4 q + e, col r): %s\n", ok64 ? "OK" : "MISMATCH", ok32 ? "OK" : "MISMATCH");
    if (!ok64) for (int l = 0; l < 64; l += 17) printf("  f64 lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    for (int w : {1, 2, 4}) {
        run<0>("mfma f64 16x16x4, 4 independent accumulators", w, 2048);
        run<1>("mfma f64 16x16x4, one dependent chain", w, 2048);
        run<2>("mfma f32 16x16x4, 4 independent accumulators", w, 2048);
        run<3>("mfma f32 16x16x4, one dependent chain", w, 2048);
        run<4>("mfma f64 + 2 fp64 FMAs per MFMA", w, 2048);
    }
    return 0;
}
