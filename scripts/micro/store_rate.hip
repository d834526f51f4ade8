// What does a store instruction cost the wave that issues it?  One workgroup of W wavefronts per CU (256 workgroups); every
// wave issues N buffer_store_dwordx4 (64 lanes x 16 B) to its own region, back to back or with V dependent fp64 FMAs
// between two stores, and stamps s_memtime around the loop.  Patterns: 0 = 1 KB contiguous per instruction, 1 = 64 lanes
// in 64 different 288-B rows (what a lane-per-chunk kernel does when every lane stores its own row), 2 = runs of 144 B
// (the staged pieces of csrc/mf_post_lds.hpp).  Backs DESIGN section 4.10 (who issues the posterior chain's stores).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/store_rate.hip -o scripts/micro/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int PATTERN, int V, bool AGPR>
__global__ void __launch_bounds__(1024) stores(char* buf, long per_wave, int n, unsigned long long* cyc, double* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char* base = buf + ((long)blockIdx.x * nw + wave) * per_wave;
    unsigned off;
    if (PATTERN == 0) off = lane * 16;
    else if (PATTERN == 1) off = lane * 60192u;               // a row per lane, rows a chunk apart (209 x 288 B)
    else off = (lane / 9) * 60192u + (lane % 9) * 16;        // 9 consecutive units per row
    v4i srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)base);
    srd.y = __builtin_amdgcn_readfirstlane((int)(((unsigned long long)base >> 32) & 0xffffu));
    srd.z = 0x7fffffff; srd.w = 0x00020000;
    v4i data = {lane, wave, 3, 4};
    double f = 1.0 + 1e-9 * lane, g = 1.0000001;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < n; ++i) {
        if (AGPR) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "a"(data), "v"(off), "s"(srd) : "memory");
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(data), "v"(off), "s"(srd) : "memory");
        off += (PATTERN == 0) ? 1024 : 288;
#pragma unroll
        for (int k = 0; k < V; ++k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f) : "v"(g));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
    if (f == 12345.678) sink[0] = f;
}

template <int PATTERN, int V, bool AGPR> void run(int W, char* buf, long bytes) {
    const int n = 200;
    const long per_wave = PATTERN == 0 ? (long)n * 1024 : 64L * 60192;
    if (256L * W * per_wave > bytes) { printf("buffer too small\n"); return; }
    unsigned long long* cyc; hipMalloc(&cyc, 8 * 256 * 16); double* sink; hipMalloc(&sink, 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((stores<PATTERN, V, AGPR>), dim3(256), dim3(64 * W), 0, 0, buf, per_wave, n, cyc, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * W);
    hipMemcpy(h.data(), cyc, 8 * 256 * W, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    printf("pattern %d  waves/CU %2d  %3d FMAs between stores  data in %s: %7.0f cycles per store instruction per wave\n", PATTERN, W, V,
           AGPR ? "AGPRs" : "VGPRs", s / h.size() / n);
    hipFree(cyc); hipFree(sink);
}

int main() {
    const long bytes = 12L << 30;
    char* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int W : {1, 2, 3, 4, 8, 16}) run<0, 0, false>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<0, 0, true>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<1, 0, false>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<2, 0, false>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<2, 16, false>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<2, 64, false>(W, buf, bytes);
    for (int W : {1, 3, 4}) run<2, 128, false>(W, buf, bytes);
    return 0;
}
