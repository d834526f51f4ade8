// What does a PARTIAL read of a 128-B line cost on gfx950?  Every lane reads `take` bytes (16-B loads) out of each
// `stride`-byte record of a buffer far larger than the Infinity Cache; time and (under rocprofv3 --pmc FETCH_SIZE)
// the counter tell whether the memory side moves whole 128-B lines, 64-B halves or 32-B sectors.
// Build: hipcc --offload-arch=gfx950 -O3 -o line_probe line_probe.hip ; run: ./line_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// record r (stride bytes) -> lane reads units [u0, u0+nu) (16 B each) of it; records dealt to lanes so that a
// wave-instruction touches 64 DIFFERENT records (like the K0 row fetch), grid-stride over records.
__global__ void probe(const uint4* __restrict__ buf, long nrec, int stride16, int u0, int nu, unsigned* sink) {
    unsigned acc = 0;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (long)gridDim.x * blockDim.x) {
        const uint4* p = buf + r * stride16 + u0;
        for (int u = 0; u < nu; ++u) {
            const uint4 v = p[u];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = size_t(4) << 30;
    uint4* buf;
    unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct Case { const char* name; int stride, u0, nu; };
    const Case cases[] = {
        {"full 128B of every 128B line", 128, 0, 8},
        {"16B at +0 of every 128B line", 128, 0, 1},
        {"16B at +64 of every 128B line", 128, 4, 1},
        {"48B at +0 of every 128B line", 128, 0, 3},
        {"64B (first half) of every 128B line", 128, 0, 4},
        {"48B straddling the 64B boundary (+40..)", 128, 2, 3},
        {"16B of every 64B half", 64, 0, 1},
        {"16B of every 32B sector", 32, 0, 1},
        {"16B of every 256B (every other line)", 256, 0, 1},
        {"32B at +0 of every 128B line", 128, 0, 2},
    };
    for (const Case& c : cases) {
        const long nrec = bytes / c.stride;
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(256 * 16), dim3(256), 0, 0, buf, nrec, c.stride / 16, c.u0, c.nu, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
        }
        const double useful = double(nrec) * c.nu * 16, span = double(bytes);
        printf("%-44s %8.3f ms  useful %7.1f GB/s  if-whole-128B-lines %7.1f GB/s\n", c.name, best,
               useful / best / 1e6, (c.stride >= 128 ? double(nrec) * 128 : span) / best / 1e6);
    }
    return 0;
}
