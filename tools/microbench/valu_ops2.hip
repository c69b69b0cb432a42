// Round-2 issue-cost calibration for gfx950: the op classes a SWAR / f32 / SDWA formulation of the FAST tests would
// use, plus LDS read shapes (byte gathers against unaligned dword / qword reads of a 40-byte-pitch tile).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_ops2.hip -o /tmp/valu_ops2 && /tmp/valu_ops2
// Cycles are printed for 1024 SIMDs at 2.4 GHz (the clock under load is lower: compare rows, not absolutes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, unsigned seed, int iters)    \
    {                                                                                           \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,  \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x5a5a5a5a, c = seed * 9; \
        for (int it = 0; it < iters; ++it) {                                                    \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                     \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_SUBREV(i) "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define A_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_MAXU(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define A_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_MIN3U(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MAXU16(i) "v_max_u16 %" #i ", %" #i ", %8\n"
#define A_MINU_SDWA(i) "v_min_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2\n"
#define A_MAXF_SDWA(i) "v_max_f32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2\n"
#define A_ADD_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2\n"
#define A_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_MIN3F(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3F(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MED3F(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PKMAXF16(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
#define A_PKMINF16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define A_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define A_PKFMAF16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define A_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_PKSUBI16(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define A_PKMAXI16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_MAXF16(i) "v_max_f16 %" #i ", %" #i ", %8\n"
#define A_CVTUB0(i) "v_cvt_f32_ubyte0 %" #i ", %" #i "\n"
#define A_CVTUB2(i) "v_cvt_f32_ubyte2 %" #i ", %" #i "\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define A_CMPSWAP(i) "v_cmp_gt_u32 vcc, %" #i ", %8\n"
#define A_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_PKFMAF32(i) "v_mov_b32 %" #i ", %" #i "\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_MSAD(i) "v_msad_u8 %" #i ", %" #i ", %8, %9\n"

KERNEL(xor, A_XOR) KERNEL(and, A_AND) KERNEL(or, A_OR) KERNEL(add, A_ADD) KERNEL(sub, A_SUB) KERNEL(subrev, A_SUBREV)
KERNEL(lshr, A_LSHR) KERNEL(lshl, A_LSHL) KERNEL(andor, A_ANDOR) KERNEL(or3, A_OR3) KERNEL(bfi, A_BFI) KERNEL(xad, A_XAD)
KERNEL(lshladd, A_LSHLADD) KERNEL(alignbit, A_ALIGNBIT) KERNEL(alignbyte, A_ALIGNBYTE)
KERNEL(minu, A_MINU) KERNEL(maxu, A_MAXU) KERNEL(maxi, A_MAXI) KERNEL(min3u, A_MIN3U) KERNEL(maxu16, A_MAXU16)
KERNEL(minu_sdwa, A_MINU_SDWA) KERNEL(maxf_sdwa, A_MAXF_SDWA) KERNEL(add_sdwa, A_ADD_SDWA)
KERNEL(maxf, A_MAXF) KERNEL(minf, A_MINF) KERNEL(subf, A_SUBF) KERNEL(min3f, A_MIN3F) KERNEL(max3f, A_MAX3F)
KERNEL(med3f, A_MED3F) KERNEL(fma, A_FMA)
KERNEL(pkmaxf16, A_PKMAXF16) KERNEL(pkminf16, A_PKMINF16) KERNEL(pkaddf16, A_PKADDF16) KERNEL(pkfmaf16, A_PKFMAF16)
KERNEL(pkaddu16, A_PKADDU16) KERNEL(pksubi16, A_PKSUBI16) KERNEL(pkmaxi16, A_PKMAXI16) KERNEL(maxf16, A_MAXF16)
KERNEL(cvtub0, A_CVTUB0) KERNEL(cvtub2, A_CVTUB2) KERNEL(perm, A_PERM) KERNEL(bfe, A_BFE) KERNEL(bcnt, A_BCNT)
KERNEL(mbcnt, A_MBCNT) KERNEL(movdpp, A_MOVDPP) KERNEL(adddpp, A_ADDDPP) KERNEL(mov, A_PKFMAF32)
KERNEL(sad, A_SAD) KERNEL(msad, A_MSAD)

template <typename K> static void run(const char *name, K kern)
{
    const int blocks = 256 * 8, iters = 4000;
    unsigned *d;
    (void)hipMalloc(&d, blocks * 256 * sizeof(unsigned));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = blocks * 4.0 * iters * 16.0;
    printf("%-18s %7.3f ms  %7.1f G wave-instr/s  %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, winst / ms * 1e-6,
           1024 * 2.4e9 / (winst / (ms * 1e-3)));
    (void)hipFree(d);
}

// ---- denormal check: are small integers compared correctly by v_max_f32 / v_min3_f32 (needs FP32 denormals on)?
__global__ void k_denorm(const unsigned *in, unsigned *out)
{
    const unsigned a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128];
    unsigned r0, r1, r2;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(a), "v"(b));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0"
                 : "=v"(r2) : "v"(a), "v"(b));
    out[threadIdx.x] = r0; out[threadIdx.x + 64] = r1; out[threadIdx.x + 128] = r2;
}

// ---- LDS read shapes on a 36 x 40-byte tile: 16 reads per iteration at per-lane pseudo-random pixel positions
template <int MODE> __global__ __launch_bounds__(64) void k_lds(unsigned *out, const int *pos, int iters)
{
    __shared__ __align__(16) unsigned char tile[36 * 40 + 64];
    for (int i = threadIdx.x; i < (36 * 40 + 64) / 4; i += 64) reinterpret_cast<unsigned *>(tile)[i] = i * 2654435761u;
    __syncthreads();
    int p = pos[blockIdx.x * 64 + threadIdx.x]; // 3*40+3 .. 32*40+36
    unsigned acc = 0;
    unsigned long long acc2 = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned char *t = &tile[p];
        if (MODE == 0) { // 16 byte gathers (the ring of a FAST candidate)
            acc += t[120] + t[121] + t[82] + t[43] + t[3] + t[-37] + t[-78] + t[-119] + t[-120] + t[-121] + t[-82] + t[-43] +
                   t[-3] + t[37] + t[78] + t[119];
        } else if (MODE == 1) { // 2 unaligned dwords + 5 unaligned qwords covering the same ring rows
            struct __attribute__((packed, aligned(1))) U32 { unsigned v; };
            struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
            acc += reinterpret_cast<const U32 *>(t + 119)->v + reinterpret_cast<const U32 *>(t - 121)->v;
            acc2 += reinterpret_cast<const U64 *>(t + 78)->v + reinterpret_cast<const U64 *>(t + 37)->v +
                    reinterpret_cast<const U64 *>(t - 3)->v + reinterpret_cast<const U64 *>(t - 43)->v +
                    reinterpret_cast<const U64 *>(t - 82)->v;
        } else if (MODE == 2) { // 7 aligned dwords (lower bound for a dword-read formulation)
            const unsigned *q = reinterpret_cast<const unsigned *>(&tile[p & ~3]);
            acc += q[30] + q[20] + q[10] + q[0] + q[-10] + q[-20] + q[-30];
        } else if (MODE == 3) { // 16 aligned dwords
            const unsigned *q = reinterpret_cast<const unsigned *>(&tile[p & ~3]);
            acc += q[30] + q[31] + q[20] + q[11] + q[1] + q[-9] + q[-19] + q[-29] + q[-30] + q[-31] + q[-21] + q[-11] +
                   q[-1] + q[9] + q[19] + q[29];
        }
        p = 3 * 40 + 3 + ((p * 13 + 7 + (int)(acc & 1)) % (29 * 40));
        if (p % 40 > 36) p -= 8;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + (unsigned)acc2 + (unsigned)(acc2 >> 32);
}

template <int MODE> static void run_lds(const char *name, int reads)
{
    const int blocks = 256 * 24, iters = 2000;
    unsigned *d;
    int *pos;
    (void)hipMalloc(&d, blocks * 64 * sizeof(unsigned));
    (void)hipMalloc(&pos, blocks * 64 * sizeof(int));
    std::vector<int> h(blocks * 64);
    srand(7);
    for (auto &v : h) v = (3 + rand() % 30) * 40 + 3 + rand() % 33;
    (void)hipMemcpy(pos, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(64), 0, 0, d, pos, 20);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(64), 0, 0, d, pos, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double rings = (double)blocks * iters; // one 64-lane ring fetch per wave-iteration
    printf("%-30s %7.3f ms  %6.1f ns per wave-ring-fetch per CU  (%d LDS instr per fetch)\n", name, ms,
           ms * 1e6 / (rings / 256.0), reads);
    (void)hipFree(d); (void)hipFree(pos);
}

int main()
{
#define R(n) run(#n, k_##n);
    R(xor) R(and) R(or) R(add) R(sub) R(subrev) R(lshr) R(lshl) R(andor) R(or3) R(bfi) R(xad) R(lshladd) R(alignbit) R(alignbyte)
    R(minu) R(maxu) R(maxi) R(min3u) R(maxu16) R(minu_sdwa) R(maxf_sdwa) R(add_sdwa)
    R(maxf) R(minf) R(subf) R(min3f) R(max3f) R(med3f) R(fma)
    R(pkmaxf16) R(pkminf16) R(pkaddf16) R(pkfmaf16) R(pkaddu16) R(pksubi16) R(pkmaxi16) R(maxf16)
    R(cvtub0) R(cvtub2) R(perm) R(bfe) R(bcnt) R(mbcnt) R(movdpp) R(adddpp) R(mov) R(sad) R(msad)
    {
        unsigned h[192], o[192], *di, *dout;
        for (int i = 0; i < 64; ++i) { h[i] = i * 4 + 1; h[i + 64] = 255 - i; h[i + 128] = (i * 37) & 255; }
        (void)hipMalloc(&di, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
        (void)hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, di, dout);
        (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        int bad0 = 0, bad1 = 0, bad2 = 0;
        for (int i = 0; i < 64; ++i) {
            const unsigned a = h[i], b = h[i + 64], c = h[i + 128];
            bad0 += o[i] != (a > b ? a : b);
            unsigned m = a < b ? a : b; m = m < c ? m : c;
            bad1 += o[i + 64] != m;
            const unsigned a8 = a & 255, b8 = b & 255;
            bad2 += o[i + 128] != (a8 > b8 ? a8 : b8);
        }
        printf("denormal-as-integer compare: v_max_f32 %d bad, v_min3_f32 %d bad, v_max_f32_sdwa(bytes) %d bad of 64\n", bad0, bad1, bad2);
    }
    run_lds<0>("16 x ds_read_u8 (ring gather)", 16);
    run_lds<1>("2 x b32 + 5 x b64 unaligned", 7);
    run_lds<2>("7 x ds_read_b32 aligned", 7);
    run_lds<3>("16 x ds_read_b32 aligned", 16);
    return 0;
}
