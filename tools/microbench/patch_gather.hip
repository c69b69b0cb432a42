// Achievable rate of the descriptor kernel's ACCESS PATTERN without its arithmetic: one wave per two key points, each staging a
// 37-row x 64-byte aligned patch (148 16-byte loads) of its frame's 1280 x 375 image, 2000 key points per frame, 512 frames,
// frame -> XCD as in the library.  Key points in random order, in raster order and in Z-order (the quadtree's output order is
// close to the last).  Prints ms per 512 frames (the descriptor kernel: 0.345) and the patch bytes per second.
//   hipcc --offload-arch=gfx950 -O3 patch_gather.hip -o /tmp/patch_gather && /tmp/patch_gather
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#define PITCH 1280
#define H 375
#define W 1242
#define NF 512
#define NK 2000
__global__ __launch_bounds__(256) void k_patch(const uint8_t *__restrict__ img, const uint32_t *__restrict__ kps, uint32_t *__restrict__ out, int per_frame)
{
    __shared__ __align__(16) uint8_t patch[4][2][37 * 64];
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3, g = j / per_frame, item = j - g * per_frame, frame = g * 8 + xcd;
    if (frame >= NF) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot0 = (item * 4 + wv) * 2;
    uint4 st[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int slot = min(slot0 + k, NK - 1);
        const uint32_t kp = __builtin_amdgcn_readfirstlane(kps[(size_t)frame * NK + slot]);
        const int x = kp & 0xFFFF, y = kp >> 16;
        const uint8_t *corner = img + (size_t)frame * PITCH * H + (size_t)(y - 18) * PITCH + ((x - 19) & ~15);
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = min(lane + 64 * it, 147), r = q >> 2, dc = q & 3;
            st[k][it] = *reinterpret_cast<const uint4 *>(corner + (size_t)r * PITCH + 16 * dc);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int q = min(lane + 64 * it, 147);
            *reinterpret_cast<uint4 *>(&patch[wv][k][16 * q]) = st[k][it];
        }
    __builtin_amdgcn_wave_barrier();
    uint32_t acc = 0;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 8; ++i) acc += patch[wv][k][(lane * 37 + i * 291) % (37 * 64)];
    if (slot0 < NK) out[((size_t)frame * NK + slot0) * 32 / 4 + (lane & 15)] = acc;
}
static uint32_t zorder(uint32_t x, uint32_t y)
{
    uint32_t z = 0;
    for (int b = 0; b < 12; ++b) z |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
    return z;
}
int main()
{
    const size_t bytes = (size_t)NF * PITCH * H;
    uint8_t *img; uint32_t *d_kp, *out;
    hipMalloc(&img, bytes); hipMemset(img, 7, bytes);
    hipMalloc(&d_kp, (size_t)NF * NK * 4); hipMalloc(&out, (size_t)NF * NK * 32);
    std::mt19937 rng(5);
    std::vector<uint32_t> kp((size_t)NF * NK);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int per_frame = (NK + 7) / 8;
    for (int order = 0; order < 3; ++order) {
        for (int f = 0; f < NF; ++f) {
            std::vector<uint32_t> v(NK);
            for (auto &p : v) p = (19 + rng() % (W - 38)) | ((19 + rng() % (H - 38)) << 16);
            if (order == 1) std::sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return (a >> 16) != (b >> 16) ? (a >> 16) < (b >> 16) : (a & 0xFFFF) < (b & 0xFFFF); });
            if (order == 2) std::sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return zorder(a & 0xFFFF, a >> 16) < zorder(b & 0xFFFF, b >> 16); });
            std::copy(v.begin(), v.end(), kp.begin() + (size_t)f * NK);
        }
        hipMemcpy(d_kp, kp.data(), kp.size() * 4, hipMemcpyHostToDevice);
        auto launch = [&] { hipLaunchKernelGGL(k_patch, dim3(8 * (NF / 8) * per_frame), dim3(256), 0, 0, img, d_kp, out, per_frame); };
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %7.3f ms per 512 frames  %7.1f GB/s of patch bytes (2368 per key point)\n", order == 0 ? "random" : order == 1 ? "raster" : "z-order", ms / 20,
               (double)NF * NK * 2368 / (ms / 20 * 1e-3) / 1e9);
    }
    return 0;
}
