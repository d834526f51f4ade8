// Probe of the LDS-DMA (buffer_load ... lds) semantics this project relies on (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(v4i srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(v4i srd, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const double* p, double* out, long nbytes, int byteshift) {
    const unsigned long long base = (unsigned long long)p + byteshift;
    v4i srd;
    srd.x = __builtin_amdgcn_readfirstlane((unsigned)base);
    srd.y = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffff);
    srd.z = __builtin_amdgcn_readfirstlane((unsigned)nbytes);
    srd.w = 0x00020000;
    for (int i = threadIdx.x; i < 4096 / 8; i += 64) ((double*)smem)[i] = -7.0;
    __syncthreads();
    unsigned lds0 = (unsigned)(size_t)smem;
    // lanes scattered: lane l reads 16 B at offset ((l*7) % 64) * 16 ; last lane out of range
    unsigned vo = ((threadIdx.x * 7) % 64) * 16;
    if (threadIdx.x == 63) vo = 0xFFFFFFF0u;
    dma16(srd, lds0, vo);
    dma4(srd, lds0 + 2048, threadIdx.x * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096 / 8; i += 64) out[i] = ((double*)smem)[i];
}
int main() {
    const int n = 4096;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    double *d, *o;
    hipMalloc(&d, n * 8); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    for (int shift : {0, 8}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 1024L, shift);
        std::vector<double> r(512);
        hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 63; ++l) {
            double e0 = ((l * 7) % 64) * 2 + shift / 8, e1 = e0 + 1;
            if (r[2 * l] != e0 || r[2 * l + 1] != e1) { if (bad < 4) printf("  lane %d got %g %g want %g %g\n", l, r[2*l], r[2*l+1], e0, e1); ++bad; }
        }
        printf("shift %d: dwordx4 mismatches %d ; OOB lane wrote (%g, %g) [untouched = -7]\n", shift, bad, r[126], r[127]);
        const float* f = (const float*)(r.data() + 256);
        const float* src = (const float*)((const char*)h.data() + shift);
        int badf = 0;
        for (int l = 0; l < 64; ++l) if (f[l] != src[l]) ++badf;
        printf("shift %d: dword mismatches %d\n", shift, badf);
    }
    return 0;
}
