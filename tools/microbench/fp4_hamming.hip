// Probe for the dense Hamming match on the FP4 matrix path (v_mfma_f32_32x32x64_f8f6f4, cbsz = blgp = 4):
//  (1) exactness and operand layout: 32 candidates x 32 queries of random 256-bit descriptors, candidates unpacked to
//      FP4 0.0 / 1.0 nibbles, queries to -1.0 / +1.0 nibbles, four MFMAs of K = 64; popcount(q) - acc must be the Hamming
//      distance;  (2) cycles per MFMA, back to back, one wave per SIMD, scaled and unscaled forms, against the i8 form.
// build: hipcc -O3 --offload-arch=gfx950 fp4_hamming.hip -o fp4_hamming
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SCALED> __device__ __forceinline__ v16f mfma_fp4(v4i a, v4i b, v16f c)
{
    v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, B = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    if (SCALED) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0, 0, 0);
}

// lane (n = lane & 31, h = lane >> 5), step t (64 bits = descriptor dwords 2t, 2t+1): operand dword j (j = 0..3) holds 8 nibbles;
// nibble i of dword j of lane half h  <-  bit (4 i + j) of descriptor dword 2 t + h   (the same rule on both operands)
template <int SCALED> __global__ void k_check(const uint32_t *cand, const uint32_t *query, float *out)
{
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    v16f acc = {};
    for (int t = 0; t < 4; ++t) {
        const uint32_t wc = cand[n * 8 + 2 * t + h], wq = query[n * 8 + 2 * t + h];
        v4i a, b;
        for (int j = 0; j < 4; ++j) {
            a[j] = (int)(((wc >> j) & 0x11111111u) << 1);                         // 1 -> 0x2 (+1.0), 0 -> 0x0
            b[j] = (int)(0x22222222u | (((~wq >> j) & 0x11111111u) << 3));       // 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0)
        }
        acc = mfma_fp4<SCALED>(a, b, acc);
    }
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

template <int KIND> __global__ void k_rate(float *out, long long *cyc, int iters)
{
    v4i a = {(int)threadIdx.x * 0x01010101, 0x22222222, 0x2a2a2a2a, 0x02020202}, b = {0x22222222, 0x2a2a2a2a, (int)threadIdx.x, 0x0a0a0a0a};
    v16f acc0 = {}, acc1 = {};
    v16i i0 = {}, i1 = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { acc0 = mfma_fp4<0>(a, b, acc0); acc1 = mfma_fp4<0>(b, a, acc1); }
        if (KIND == 1) { acc0 = mfma_fp4<1>(a, b, acc0); acc1 = mfma_fp4<1>(b, a, acc1); }
        if (KIND == 2) { i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, i0, 0, 0, 0); i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, i1, 0, 0, 0); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + (float)(i0[r] + i1[r]);
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    std::vector<uint32_t> c(256), q(256);
    srand(7);
    for (auto &v : c) v = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
    for (auto &v : q) v = (uint32_t)rand() * 2246822519u ^ (uint32_t)rand();
    for (int t = 0; t < 8; ++t) { q[t] = c[t]; q[8 + t] = ~c[3 * 8 + t]; }   // distance 0 and distance 256 cases
    uint32_t *dc, *dq; float *dout; long long *dcyc;
    (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dq, 1024); (void)hipMalloc(&dout, 64 * 16 * 4 * 1024); (void)hipMalloc(&dcyc, 8 * 4096);
    hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dq, q.data(), 1024, hipMemcpyHostToDevice);
    for (int scaled = 0; scaled < 2; ++scaled) {
        if (scaled) hipLaunchKernelGGL(k_check<1>, dim3(1), dim3(64), 0, 0, dc, dq, dout);
        else hipLaunchKernelGGL(k_check<0>, dim3(1), dim3(64), 0, 0, dc, dq, dout);
        std::vector<float> o(1024);
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int qi = lane & 31, ci = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); // C/D layout: column on the lane, row by register
                int ham = 0, pc = 0;
                for (int t = 0; t < 8; ++t) { ham += __builtin_popcount(c[ci * 8 + t] ^ q[qi * 8 + t]); pc += __builtin_popcount(q[qi * 8 + t]); }
                // acc = sum_k a_k (2 b_k - 1) = 2 popcount(a & b) - popcount(a)  (a = candidate bits, b = query bits),
                // so hamming(a, b) = popcount(b) - acc
                int dot = 0;
                for (int t = 0; t < 8; ++t) dot += __builtin_popcount(c[ci * 8 + t] & q[qi * 8 + t]) - __builtin_popcount(c[ci * 8 + t] & ~q[qi * 8 + t]);
                if ((int)o[lane * 16 + r] != dot || ham != pc - dot) { if (bad < 5) printf("  lane %d r %d: got %g want %d (hamming %d)\n", lane, r, o[lane * 16 + r], dot, ham); ++bad; }
            }
        printf("%s form: %d mismatches of 1024 (acc = popcount(query) - hamming)\n", scaled ? "scaled" : "unscaled", bad);
    }
    const int iters = 4096, blocks = 1024;
    const char *names[3] = {"fp4 32x32x64 unscaled", "fp4 32x32x64 scaled", "i8 32x32x32"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, dout, dcyc, iters);
            if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, dout, dcyc, iters);
            if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, 0, dout, dcyc, iters);
            hipDeviceSynchronize();
        }
        std::vector<long long> cy(blocks);
        hipMemcpy(cy.data(), dcyc, 8 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : cy) s += (double)v;
        printf("%-24s %.1f cycles per MFMA (one wave per SIMD, two accumulators)\n", names[kind], s / blocks / (2.0 * iters));
    }
    return 0;
}
