// A C++ caller that works on the NULL stream the way code next to Frame.cpp:20 naturally would (include/orbx.h, "Streams"):
// it fills its device buffers on stream 0, calls the *_device entry points with stream == NULL and reads the results on
// stream 0 -- with NO synchronisation of its own in between.  Two cases:
//   1. orbm_best2_device(..., NULL) right behind asynchronous fills that are still in flight (a long memset runs ahead of them
//      on the same stream, so a call that did not order itself behind stream 0 would read the poison the buffers held before);
//   2. two HANDLES chained with NULL: orbx_extract_batch_device(NULL) -> orbm_best2_device(NULL) on the records it leaves,
//      against the same two steps through the synchronous host entry points.
// Expected answers come from a plain popcount loop in this file.  Exit status 0 = all equal.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbm.h"
#include "orbx.h"

#define HIP_OK(e)                                                                      \
    do {                                                                               \
        hipError_t e_ = (e);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 10;                                                                 \
        }                                                                              \
    } while (0)
#define ORB_OK(e)                                                                      \
    do {                                                                               \
        int e_ = (e);                                                                  \
        if (e_ != 0) {                                                                 \
            std::fprintf(stderr, "%s:%d orbx error %d: %s\n", __FILE__, __LINE__, e_, orbx_last_error()); \
            return 11;                                                                 \
        }                                                                              \
    } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

// ORBMatcher.cpp:148-162: strict '<' in ascending candidate order, both distances start at 256
static void best2_host(const uint8_t *a, int na, const uint8_t *b, int nb, std::vector<int32_t> &bi, std::vector<uint16_t> &bd,
                       std::vector<uint16_t> &sd)
{
    bi.assign(na, -1); bd.assign(na, 256); sd.assign(na, 256);
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            int d = 0;
            for (int k = 0; k < 32; ++k) d += __builtin_popcount(a[32 * i + k] ^ b[32 * j + k]);
            if (d < bd[i]) { sd[i] = bd[i]; bd[i] = (uint16_t)d; bi[i] = j; }
            else if (d < sd[i]) sd[i] = (uint16_t)d;
        }
}

int main(int argc, char **argv)
{
    const int N = 2000;
    orbm_t *mh = nullptr;
    ORB_OK(orbm_create(-1, &mh));

    // ---- case 1
    uint8_t *hA, *hB; int32_t *hn;
    HIP_OK(hipHostMalloc((void **)&hA, N * 32)); HIP_OK(hipHostMalloc((void **)&hB, N * 32)); HIP_OK(hipHostMalloc((void **)&hn, 8));
    uint32_t seed = 12345;
    for (int i = 0; i < N * 32; ++i) { hA[i] = (uint8_t)(lcg(seed) >> 24); hB[i] = (uint8_t)(lcg(seed) >> 24); }
    for (int i = 0; i < N; i += 3) std::memcpy(hB + 32 * ((i * 7) % N), hA + 32 * i, 32), hB[32 * ((i * 7) % N) + (i % 32)] ^= 0x11; // near matches
    hn[0] = N; hn[1] = N;
    uint8_t *dA, *dB, *big; int32_t *dn, *dbi; uint16_t *dbd, *dsd;
    const size_t BIG = (size_t)1 << 30;
    HIP_OK(hipMalloc((void **)&dA, N * 32)); HIP_OK(hipMalloc((void **)&dB, N * 32)); HIP_OK(hipMalloc((void **)&dn, 8));
    HIP_OK(hipMalloc((void **)&dbi, N * 4)); HIP_OK(hipMalloc((void **)&dbd, N * 2)); HIP_OK(hipMalloc((void **)&dsd, N * 2));
    HIP_OK(hipMalloc((void **)&big, BIG));
    int32_t *rbi; uint16_t *rbd, *rsd;
    HIP_OK(hipHostMalloc((void **)&rbi, N * 4)); HIP_OK(hipHostMalloc((void **)&rbd, N * 2)); HIP_OK(hipHostMalloc((void **)&rsd, N * 2));
    std::vector<int32_t> wbi; std::vector<uint16_t> wbd, wsd;
    best2_host(hA, N, hB, N, wbi, wbd, wsd);
    int bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        // poison, then something long on stream 0, then the real fills -- all asynchronous, none waited for
        HIP_OK(hipMemsetAsync(dA, 0xA5, N * 32, nullptr)); HIP_OK(hipMemsetAsync(dB, 0x5A, N * 32, nullptr));
        HIP_OK(hipMemsetAsync(dn, 0, 8, nullptr));
        HIP_OK(hipMemsetAsync(dbi, 0x7F, N * 4, nullptr));
        HIP_OK(hipDeviceSynchronize()); // (the poison is in place; from here on nothing waits)
        HIP_OK(hipMemsetAsync(big, rep, BIG, nullptr));
        HIP_OK(hipMemcpyAsync(dA, hA, N * 32, hipMemcpyHostToDevice, nullptr));
        HIP_OK(hipMemcpyAsync(dB, hB, N * 32, hipMemcpyHostToDevice, nullptr));
        HIP_OK(hipMemcpyAsync(dn, hn, 8, hipMemcpyHostToDevice, nullptr));
        ORB_OK(orbm_best2_device(mh, 1, dA, N, dn, N, dB, N, dn + 1, N, nullptr, nullptr, dbi, dbd, dsd, nullptr));
        HIP_OK(hipMemcpyAsync(rbi, dbi, N * 4, hipMemcpyDeviceToHost, nullptr));
        HIP_OK(hipMemcpyAsync(rbd, dbd, N * 2, hipMemcpyDeviceToHost, nullptr));
        HIP_OK(hipMemcpyAsync(rsd, dsd, N * 2, hipMemcpyDeviceToHost, nullptr));
        HIP_OK(hipStreamSynchronize(nullptr)); // the host reads: the one wait
        for (int i = 0; i < N; ++i) bad += (rbi[i] != wbi[i]) + (rbd[i] != wbd[i]) + (rsd[i] != wsd[i]);
    }
    std::printf("case 1 (fills in flight on stream 0, orbm_best2_device(NULL)): %d mismatches of %d\n", bad, 9 * N);
    if (bad) return 1;

    // ---- case 2: two handles, NULL everywhere
    const int W = 640, H = 360, F = 2;
    orbx_cfg cfg = {1000, 1.2f, 8, 20, 7, W, H, F, 0, -1};
    orbx_t *xh = nullptr;
    ORB_OK(orbx_create(&cfg, &xh));
    const int cap = orbx_max_keypoints(xh, W, H);
    std::vector<uint8_t> img((size_t)F * W * H);
    for (int f = 0; f < F; ++f)      // blocks of random grey levels (corners at the block borders), frame 1 shifted by (3, 2)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int bx = (x + 3 * f) / 12, by = (y + 2 * f) / 10;
                uint32_t s = (uint32_t)(bx * 7919 + by * 104729 + 17);
                img[((size_t)f * H + y) * W + x] = (uint8_t)(40 + (lcg(s) >> 25) + ((x * 3 + y * 5) & 7));
            }
    // reference run: host entry points (synchronous)
    std::vector<orbx_kp> hkp((size_t)F * cap); std::vector<uint8_t> hdesc((size_t)F * cap * 32); int32_t hcnt[F];
    ORB_OK(orbx_extract_batch(xh, img.data(), F, W, H, W, (size_t)W * H, hkp.data(), hdesc.data(), cap, hcnt));
    std::vector<int32_t> w2bi; std::vector<uint16_t> w2bd, w2sd;
    best2_host(hdesc.data(), hcnt[0], hdesc.data() + (size_t)cap * 32, hcnt[1], w2bi, w2bd, w2sd);
    // device chain on NULL: H2D, extract, match, D2H -- no wait until the end
    uint8_t *dimg, *ddesc; orbx_kp *dkp; int32_t *dcnt, *d2bi; uint16_t *d2bd, *d2sd; uint8_t *himg;
    HIP_OK(hipHostMalloc((void **)&himg, img.size())); std::memcpy(himg, img.data(), img.size());
    HIP_OK(hipMalloc((void **)&dimg, img.size())); HIP_OK(hipMalloc((void **)&dkp, sizeof(orbx_kp) * F * cap));
    HIP_OK(hipMalloc((void **)&ddesc, (size_t)F * cap * 32)); HIP_OK(hipMalloc((void **)&dcnt, 4 * F));
    HIP_OK(hipMalloc((void **)&d2bi, 4 * cap)); HIP_OK(hipMalloc((void **)&d2bd, 2 * cap)); HIP_OK(hipMalloc((void **)&d2sd, 2 * cap));
    int32_t *r2bi, *rcnt; uint16_t *r2bd, *r2sd;
    HIP_OK(hipHostMalloc((void **)&r2bi, 4 * cap)); HIP_OK(hipHostMalloc((void **)&r2bd, 2 * cap)); HIP_OK(hipHostMalloc((void **)&r2sd, 2 * cap));
    HIP_OK(hipHostMalloc((void **)&rcnt, 4 * F));
    HIP_OK(hipMemsetAsync(dimg, 0, img.size(), nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemsetAsync(big, 9, BIG, nullptr));
    HIP_OK(hipMemcpyAsync(dimg, himg, img.size(), hipMemcpyHostToDevice, nullptr));
    ORB_OK(orbx_extract_batch_device(xh, dimg, F, W, H, W, (size_t)W * H, dkp, ddesc, cap, dcnt, nullptr));
    ORB_OK(orbm_best2_device(mh, 1, ddesc, cap, dcnt, cap, ddesc + (size_t)cap * 32, cap, dcnt + 1, cap, nullptr, nullptr, d2bi, d2bd,
                             d2sd, nullptr));
    HIP_OK(hipMemcpyAsync(rcnt, dcnt, 4 * F, hipMemcpyDeviceToHost, nullptr));
    HIP_OK(hipMemcpyAsync(r2bi, d2bi, 4 * cap, hipMemcpyDeviceToHost, nullptr));
    HIP_OK(hipMemcpyAsync(r2bd, d2bd, 2 * cap, hipMemcpyDeviceToHost, nullptr));
    HIP_OK(hipMemcpyAsync(r2sd, d2sd, 2 * cap, hipMemcpyDeviceToHost, nullptr));
    HIP_OK(hipStreamSynchronize(nullptr));
    if (rcnt[0] != hcnt[0] || rcnt[1] != hcnt[1] || hcnt[0] < 200) {
        std::printf("case 2: key-point counts %d %d against %d %d\n", rcnt[0], rcnt[1], hcnt[0], hcnt[1]);
        return 2;
    }
    bad = 0;
    for (int i = 0; i < hcnt[0]; ++i) bad += (r2bi[i] != w2bi[i]) + (r2bd[i] != w2bd[i]) + (r2sd[i] != w2sd[i]);
    std::printf("case 2 (orbx_extract_batch_device(NULL) -> orbm_best2_device(NULL), %d x %d key points): %d mismatches\n", hcnt[0],
                hcnt[1], bad);
    if (bad) return 3;
    ORB_OK(orbx_synchronize(xh));
    orbx_destroy(xh);
    orbm_destroy(mh);
    std::printf("null stream ok\n");
    return 0;
}
