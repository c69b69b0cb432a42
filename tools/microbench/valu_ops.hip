// Per-opcode integer VALU issue cost on gfx950: 8 independent chains of one instruction, many waves per SIMD.
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
// Prints cycles per wave64 instruction per SIMD assuming 1024 SIMDs at 2.4 GHz (the real clock under load is lower,
// so compare rows with each other rather than reading the absolute numbers).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

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
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MED3(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define A_PKMINU(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"

#define A_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_MIN3F(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MED3F(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define A_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_SUBI(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
KERNEL(max_f32, A_MAXF)
KERNEL(sub_f32, A_SUBF)
KERNEL(min3_f32, A_MIN3F)
KERNEL(med3_f32, A_MED3F)
KERNEL(cvt_f32_ubyte1, A_CVTUB)
KERNEL(cvt_f32_u32, A_CVTFU)
KERNEL(mul_f32, A_MULF)
KERNEL(max_i32, A_MAXI)
KERNEL(and_b32, A_AND)
KERNEL(lshrrev, A_LSHR)
KERNEL(sub_u32, A_SUBI)
KERNEL(xor, A_XOR)
KERNEL(add, A_ADD)
KERNEL(bcnt, A_BCNT)
KERNEL(min, A_MIN)
KERNEL(min3, A_MIN3)
KERNEL(med3, A_MED3)
KERNEL(perm, A_PERM)
KERNEL(alignbyte, A_ALIGN)
KERNEL(lshl_or, A_LSHLOR)
KERNEL(mad24, A_MAD24)
KERNEL(mul_lo, A_MULLO)
KERNEL(dot4, A_DOT4)
KERNEL(dot2, A_DOT2)
KERNEL(sad, A_SAD)
KERNEL(pk_sub_i16, A_PKSUB)
KERNEL(pk_max_i16, A_PKMAX)
KERNEL(pk_min_u16, A_PKMINU)
KERNEL(cndmask, A_CNDMASK)
KERNEL(fma_f32, A_FMA)
KERNEL(bfe, A_BFE)
KERNEL(add3, A_ADD3)
KERNEL(cvt_i32_f32, A_CVTI)
KERNEL(rndne, A_RNDNE)

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
    printf("%-14s %7.3f ms  %7.1f G wave-instr/s  %.2f cycles/instr/SIMD @2.4GHz\n", name, ms, winst / ms * 1e-6,
           1024 * 2.4e9 / (winst / (ms * 1e-3)));
    (void)hipFree(d);
}

int main()
{
    run("v_xor_b32", k_xor); run("v_add_u32", k_add); run("v_bcnt_u32_b32", k_bcnt); run("v_min_u32", k_min);
    run("v_min3_u32", k_min3); run("v_med3_u32", k_med3); run("v_perm_b32", k_perm); run("v_alignbyte", k_alignbyte);
    run("v_lshl_or_b32", k_lshl_or); run("v_mad_u32_u24", k_mad24); run("v_mul_lo_u32", k_mul_lo);
    run("v_dot4_u32_u8", k_dot4); run("v_dot2_u32_u16", k_dot2); run("v_sad_u8", k_sad);
    run("v_pk_sub_i16", k_pk_sub_i16); run("v_pk_max_i16", k_pk_max_i16); run("v_pk_min_u16", k_pk_min_u16);
    run("v_cndmask_b32", k_cndmask); run("v_fma_f32", k_fma_f32); run("v_bfe_u32", k_bfe); run("v_add3_u32", k_add3);
    run("v_cvt_i32_f32", k_cvt_i32_f32); run("v_rndne_f32", k_rndne);
    run("v_max_f32", k_max_f32); run("v_sub_f32", k_sub_f32); run("v_min3_f32", k_min3_f32); run("v_med3_f32", k_med3_f32);
    run("v_cvt_f32_ubyte1", k_cvt_f32_ubyte1); run("v_cvt_f32_u32", k_cvt_f32_u32); run("v_mul_f32", k_mul_f32);
    run("v_max_i32", k_max_i32); run("v_and_b32", k_and_b32); run("v_lshrrev_b32", k_lshrrev); run("v_sub_u32", k_sub_u32);
    return 0;
}
