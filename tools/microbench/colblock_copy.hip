// Achievable HBM rate of the blur kernel's ACCESS PATTERN without its arithmetic: a workgroup walks down a column block of an
// image (W bytes wide + 16 either side read, W written), 32 rows per trip, 512 images of 1280 x 375 bytes (pitch 1280), against a
// plain linear copy of the same bytes.  Prints GB/s of bytes read + written (the blur's own figure: 4.2 TB/s at W = 128).
//   hipcc --offload-arch=gfx950 -O3 colblock_copy.hip -o /tmp/colblock_copy && /tmp/colblock_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define PITCH 1280
#define H 375
#define NF 512
template <int W> // W = 128 or 256 output bytes per row
__global__ __launch_bounds__(256) void k_colblock(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int nbx)
{
    const int bid = blockIdx.x, frame = bid / nbx, bx = bid - frame * nbx;
    const uint8_t *S = src + (size_t)frame * PITCH * H;
    uint8_t *D = dst + (size_t)frame * PITCH * H;
    constexpr int CIN = (W + 32) / 16, COUT = W / 16;
    const int xo = min(max(W * bx - 16, 0), PITCH - (W + 32));
    for (int y0 = 0; y0 < H; y0 += 32) {
        uint4 acc = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < 32 * CIN; i += 256) {
            const int r = i / CIN, c = i - r * CIN, y = min(y0 + r, H - 1);
            const uint4 v = *reinterpret_cast<const uint4 *>(S + (size_t)y * PITCH + xo + 16 * c);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        for (int i = threadIdx.x; i < 32 * COUT; i += 256) {
            const int r = i / COUT, c = i - r * COUT, y = y0 + r;
            if (y < H && W * bx + 16 * c < PITCH) *reinterpret_cast<uint4 *>(D + (size_t)y * PITCH + W * bx + 16 * c) = acc;
        }
    }
}
__global__ __launch_bounds__(256) void k_linear(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main()
{
    const size_t bytes = (size_t)NF * PITCH * H;
    uint8_t *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double moved) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.3f ms  %7.1f GB/s (bytes read + written)\n", name, ms / 20, moved / (ms / 20 * 1e-3) / 1e9);
    };
    run("linear copy", [&] { hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); }, 2.0 * bytes);
    run("column blocks of 128 B (+32 read)", [&] { hipLaunchKernelGGL(k_colblock<128>, dim3(NF * 10), dim3(256), 0, 0, a, b, 10); }, bytes * (160.0 / 128 + 1));
    run("column blocks of 256 B (+32 read)", [&] { hipLaunchKernelGGL(k_colblock<256>, dim3(NF * 5), dim3(256), 0, 0, a, b, 5); }, bytes * (288.0 / 256 + 1));
    return 0;
}
