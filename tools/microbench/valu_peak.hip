// Integer VALU issue-rate calibration for gfx950: independent xor / bcnt / min chains, many waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak ; prints wave-instructions per second.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters)
{
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9e3779b9u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = (a[i] ^ seed) + 0x01010101u;                 // xor + add
            if (MODE == 1) a[i] = __builtin_popcount(a[i] ^ seed) + a[(i + 1) & 7]; // xor + bcnt(acc)
            if (MODE == 2) a[i] = min(a[i] ^ 0x55555555u, a[(i + 3) & 7] + 1u);  // xor + add + min
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> static void run(const char *name, int insts_per_iter)
{
    const int blocks = 256 * 8, iters = 20000;
    unsigned *d;
    hipMalloc(&d, blocks * 256 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, winst = waves * iters * insts_per_iter;
    printf("%-18s %.3f ms  %.1f G wave-instr/s  = %.2f cycles per wave-instr per SIMD at 2.4 GHz (1024 SIMDs)\n", name, ms,
           winst / ms * 1e-6, 1024 * 2.4e9 / (winst / (ms * 1e-3)));
    hipFree(d);
}

int main()
{
    run<0>("xor+add", 16);
    run<1>("xor+bcnt", 16);
    run<2>("xor+add+min", 24);
    return 0;
}
