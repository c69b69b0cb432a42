// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this repository's kernels use
// (MI355X_MICROARCH.md, HBM section: 16 B/lane streaming reads are reported at exactly 1/2; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel below reads each byte of a
// W x H image of known size exactly once from the memory system's point of view (overlapping re-reads by neighbouring
// lanes hit L1/L2); the expected FETCH_SIZE is W*H bytes times the factor to be determined.
//   build + run on the GPU box:
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o /tmp/fetch_calib
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int W = 16384, H = 24576; // 384 MiB: larger than the 256 MiB Infinity Cache
constexpr size_t BYTES = (size_t)W * H;

template <typename T> __global__ __launch_bounds__(256) void k_stream(const T *__restrict__ src, unsigned *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i < n; i += stride) {
        const T v = src[i];
        const unsigned *p = reinterpret_cast<const unsigned *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= p[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// FAST-like: one wave = one 30x30 cell, reads its 36 x 40-byte tile with 180 unaligned 8-byte loads
struct __attribute__((packed, aligned(1))) U64u { unsigned long long v; };
__global__ __launch_bounds__(64) void k_fast_like(const uint8_t *__restrict__ img, unsigned *out)
{
    const int cells_x = (W - 40) / 30, cx = blockIdx.x % cells_x, cy = blockIdx.x / cells_x;
    const uint8_t *S = img + (size_t)(cy * 30) * W + cx * 30;
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < 5 * 36; i += 64) {
        const int ty = i / 5, tx = (i - 5 * ty) * 8;
        acc ^= reinterpret_cast<const U64u *>(S + (size_t)ty * W + tx)->v;
    }
    if (acc == 0x123456789abcdefull) out[0] = 1;
}

// blur-like: thread = 4 columns, 32 rows; per row three dword loads covering bytes x-4 .. x+7
__global__ __launch_bounds__(256) void k_blur_like(const uint8_t *__restrict__ img, unsigned *out)
{
    const int groups = W / 4 - 2, g = 1 + (int)(((size_t)blockIdx.x * 256 + threadIdx.x) % groups);
    const int band = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) / groups);
    if (band * 32 + 32 > H) return;
    const uint8_t *p = img + (size_t)band * 32 * W + 4 * g;
    unsigned acc = 0;
    for (int r = 0; r < 32; ++r) {
        const unsigned *q = reinterpret_cast<const unsigned *>(p + (size_t)r * W);
        acc ^= q[-1] ^ q[0] ^ q[1];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    uint8_t *img;
    unsigned *out;
    if (hipMalloc(&img, BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(img, 0x5a, BYTES);
    (void)hipDeviceSynchronize();
    const int blocks = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4 *)img, out, BYTES / 16);
        hipLaunchKernelGGL(k_stream<uint2>, dim3(blocks), dim3(256), 0, 0, (const uint2 *)img, out, BYTES / 8);
        hipLaunchKernelGGL(k_stream<unsigned>, dim3(blocks), dim3(256), 0, 0, (const unsigned *)img, out, BYTES / 4);
        const int cells = ((W - 40) / 30) * ((H - 40) / 30);
        hipLaunchKernelGGL(k_fast_like, dim3(cells), dim3(64), 0, 0, img, out);
        const size_t threads = (size_t)(W / 4 - 2) * (H / 32);
        hipLaunchKernelGGL(k_blur_like, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, img, out);
    }
    (void)hipDeviceSynchronize();
    printf("image %d x %d = %zu bytes (%.1f MiB); every kernel reads each byte once (fast-like/blur-like skip a thin margin)\n",
           W, H, BYTES, BYTES / 1048576.0);
    return 0;
}
