// How fast can ONE wavefront per SIMD issue fp64 / fp32 FMAs on gfx950, vs two per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int NACC>
__global__ void __launch_bounds__(64) fma_chain(T* out, int iters, T a, T b) {
    T acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = T(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename T, int NACC> void run(const char* name, int waves_per_simd) {
    T* out; hipMalloc(&out, 8 * 64 * 4096);
    const int iters = 20000;
    const int grid = 256 * 4 * waves_per_simd;     // one 64-thread block per SIMD (x waves_per_simd)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fma_chain<T, NACC>), dim3(grid), dim3(64), 0, 0, out, 100, T(1.0000001), T(1e-9));
    hipEventRecord(e0);
    hipLaunchKernelGGL((fma_chain<T, NACC>), dim3(grid), dim3(64), 0, 0, out, iters, T(1.0000001), T(1e-9));
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = double(iters) * 16 * NACC;
    const double ns_per_instr = ms * 1e6 / instr_per_wave / waves_per_simd;   // per SIMD
    printf("%s NACC=%2d waves/SIMD=%d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.1f cycles @2.4GHz), %.1f TFLOP/s\n",
           name, NACC, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4,
           2.0 * 64 * instr_per_wave * grid / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<double, 1>("f64", 1); run<double, 2>("f64", 1); run<double, 4>("f64", 1); run<double, 8>("f64", 1);
    run<double, 8>("f64", 2); run<double, 1>("f64", 2); run<double, 2>("f64", 2); run<double, 8>("f64", 4);
    run<float, 1>("f32", 1); run<float, 4>("f32", 1); run<float, 8>("f32", 1); run<float, 8>("f32", 2); run<float, 8>("f32", 4);
    return 0;
}
