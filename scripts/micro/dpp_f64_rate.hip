// Issue rate of v_fmac_f64 with a DPP row_newbcast operand (the only DPP control gfx950 has for 64-bit operations)
// against the plain v_fmac_f64, and a check of what the broadcast delivers.  Backs the "one matrix row per lane of a
// 16-lane DPP row" kernels (csrc/mf_row.hpp): there the broadcast operand replaces every cross-lane move.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/dpp_f64_rate.hip -o scripts/micro/dpp_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define FM(a, s, o, K) "v_fmac_f64_dpp %" #a ", %" #s ", %" #o " row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n"
#define FP(a, s, o) "v_fmac_f64_e32 %" #a ", %" #s ", %" #o "\n"

template <int MODE>
__global__ void __launch_bounds__(64) rate(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double s0 = 1e-9 * threadIdx.x, s1 = 2e-9, o0 = 1.0000001, o1 = 0.9999999;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {          // 8 independent accumulators, plain
            asm volatile(FP(0, 8, 10) FP(1, 9, 11) FP(2, 8, 11) FP(3, 9, 10) FP(4, 8, 10) FP(5, 9, 11) FP(6, 8, 11) FP(7, 9, 10)
                         FP(0, 8, 10) FP(1, 9, 11) FP(2, 8, 11) FP(3, 9, 10) FP(4, 8, 10) FP(5, 9, 11) FP(6, 8, 11) FP(7, 9, 10)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        } else if constexpr (MODE == 1) {   // 8 independent accumulators, DPP broadcast operand
            asm volatile(FM(0, 8, 10, 0) FM(1, 9, 11, 1) FM(2, 8, 11, 2) FM(3, 9, 10, 3) FM(4, 8, 10, 4) FM(5, 9, 11, 5) FM(6, 8, 11, 6) FM(7, 9, 10, 7)
                         FM(0, 8, 10, 8) FM(1, 9, 11, 9) FM(2, 8, 11, 10) FM(3, 9, 10, 11) FM(4, 8, 10, 12) FM(5, 9, 11, 13) FM(6, 8, 11, 14) FM(7, 9, 10, 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        } else if constexpr (MODE == 2) {   // ONE accumulator: the dependent chain, plain
            asm volatile(FP(0, 8, 10) FP(0, 9, 11) FP(0, 8, 11) FP(0, 9, 10) FP(0, 8, 10) FP(0, 9, 11) FP(0, 8, 11) FP(0, 9, 10)
                         FP(0, 8, 10) FP(0, 9, 11) FP(0, 8, 11) FP(0, 9, 10) FP(0, 8, 10) FP(0, 9, 11) FP(0, 8, 11) FP(0, 9, 10)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        } else if constexpr (MODE == 3) {   // ONE accumulator, DPP
            asm volatile(FM(0, 8, 10, 0) FM(0, 9, 11, 1) FM(0, 8, 11, 2) FM(0, 9, 10, 3) FM(0, 8, 10, 4) FM(0, 9, 11, 5) FM(0, 8, 11, 6) FM(0, 9, 10, 7)
                         FM(0, 8, 10, 8) FM(0, 9, 11, 9) FM(0, 8, 11, 10) FM(0, 9, 10, 11) FM(0, 8, 10, 12) FM(0, 9, 11, 13) FM(0, 8, 11, 14) FM(0, 9, 10, 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        } else if constexpr (MODE == 4) {   // two accumulators alternating, DPP (chain of 2)
            asm volatile(FM(0, 8, 10, 0) FM(1, 9, 11, 1) FM(0, 8, 11, 2) FM(1, 9, 10, 3) FM(0, 8, 10, 4) FM(1, 9, 11, 5) FM(0, 8, 11, 6) FM(1, 9, 10, 7)
                         FM(0, 8, 10, 8) FM(1, 9, 11, 9) FM(0, 8, 11, 10) FM(1, 9, 10, 11) FM(0, 8, 10, 12) FM(1, 9, 11, 13) FM(0, 8, 11, 14) FM(1, 9, 10, 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        } else {                            // the accumulator written by one fmac is the DPP source of the next but two (s_nop-free distance 2)
            asm volatile(FM(0, 8, 10, 0) FM(1, 9, 11, 1) FM(2, 0, 11, 2) FM(3, 1, 10, 3) FM(4, 2, 10, 4) FM(5, 3, 11, 5) FM(6, 4, 11, 6) FM(7, 5, 10, 7)
                         FM(0, 6, 10, 8) FM(1, 7, 11, 9) FM(2, 0, 11, 10) FM(3, 1, 10, 11) FM(4, 2, 10, 12) FM(5, 3, 11, 13) FM(6, 4, 11, 14) FM(7, 5, 10, 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(s0), "v"(s1), "v"(o0), "v"(o1));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char* name, int waves_per_simd) {
    double* out; hipMalloc(&out, 8 * 64 * 8192);
    const int iters = 40000;
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE>), dim3(grid), dim3(64), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<MODE>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = double(iters) * 16;
    const double ns = ms * 1e6 / instr_per_wave / waves_per_simd;
    printf("%-34s waves/SIMD=%d: %8.3f ms  %.2f ns per wave-instruction per SIMD = %.2f cycles @2.4GHz\n", name, waves_per_simd, ms, ns, ns * 2.4);
    hipFree(out);
}

// what row_newbcast:K delivers: lane l of every 16-lane row receives lane (l & ~15) + K
__global__ void __launch_bounds__(64) semantics(double* out) {
    double acc = 0.0, src = 100.0 + threadIdx.x, one = 1.0, acc2 = 0.0;
    asm volatile("s_nop 1\n" FM(0, 2, 3, 5) FM(1, 2, 3, 12) : "+v"(acc), "+v"(acc2) : "v"(src), "v"(one));
    out[threadIdx.x] = acc;
    out[64 + threadIdx.x] = acc2;
}

int main() {
    double* d; hipMalloc(&d, 128 * 8);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
    double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok &= h[l] == 100.0 + (l & ~15) + 5 && h[64 + l] == 100.0 + (l & ~15) + 12;
    printf("row_newbcast semantics (lane l gets lane (l & ~15) + K): %s   e.g. lane 37 -> %.0f, %.0f\n", ok ? "OK" : "MISMATCH", h[37], h[64 + 37]);
    for (int w : {1, 2, 4}) {
        run<0>("plain fmac, 8 independent", w);
        run<1>("dpp fmac, 8 independent", w);
        run<2>("plain fmac, dependent chain", w);
        run<3>("dpp fmac, dependent chain", w);
        run<4>("dpp fmac, 2 chains", w);
        run<5>("dpp fmac, src written 2 before", w);
    }
    return ok ? 0 : 1;
}
