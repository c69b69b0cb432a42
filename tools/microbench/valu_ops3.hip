// Round-3 issue-cost calibration: the opcodes of k_fast_strip's hot loops that valu_ops2.hip does not cover (compares,
// selects, v_bitop3, 24-bit multiplies, lane counts, literal / SGPR operand forms), same harness: 2048 x 256 threads,
// 8 independent chains x 16 per iteration.  Output rows feed tools/isa_mix.py (profiles/r03_valu_ops3.txt).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_ops3.hip -o /tmp/valu_ops3 && /tmp/valu_ops3
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, unsigned seed, int iters)    \
    {                                                                                           \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,  \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x5a5a5a5a, c = seed * 9; \
        const unsigned sg = __builtin_amdgcn_readfirstlane(seed * 77u);                        \
        for (int it = 0; it < iters; ++it) {                                                    \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                  \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c), "s"(sg) : "vcc", "s40", "s41");                        \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_AND_LIT(i) "v_and_b32 %" #i ", 0x3f3f3f3f, %" #i "\n"
#define A_AND_SGPR(i) "v_and_b32 %" #i ", %10, %" #i "\n"
#define A_ADD_LIT(i) "v_add_u32 %" #i ", 0x01010101, %" #i "\n"
#define A_ADD_INL(i) "v_add_u32 %" #i ", 7, %" #i "\n"
#define A_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
#define A_CMP32(i) "v_cmp_gt_u32 vcc, %" #i ", %8\n"
#define A_CMP64(i) "v_cmp_gt_u32 s[40:41], %" #i ", %8\n"
#define A_CMPNE32(i) "v_cmp_ne_u32 vcc, 0, %" #i "\n"
#define A_CNDMASK32(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDMASK64(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[40:41]\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xe8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3U(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MBCNTHI(i) "v_mbcnt_hi_u32_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL1(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define A_XOR_LIT(i) "v_xor_b32 %" #i ", 0x80808080, %" #i "\n"
#define A_MINU16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define A_SUBU16(i) "v_sub_u16 %" #i ", %" #i ", %8\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define A_PKMULF32(i) "v_mov_b32 %" #i ", %" #i "\n"
#define A_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"

KERNEL(and, A_AND) KERNEL(and_lit, A_AND_LIT) KERNEL(and_sgpr, A_AND_SGPR) KERNEL(add_lit, A_ADD_LIT) KERNEL(add_inl, A_ADD_INL)
KERNEL(not, A_NOT) KERNEL(cmp32, A_CMP32) KERNEL(cmp64, A_CMP64) KERNEL(cmpne32, A_CMPNE32) KERNEL(cndmask32, A_CNDMASK32)
KERNEL(cndmask64, A_CNDMASK64) KERNEL(bitop3, A_BITOP3) KERNEL(mad24, A_MAD24) KERNEL(mul24, A_MUL24) KERNEL(lshlor, A_LSHLOR)
KERNEL(add3, A_ADD3) KERNEL(max3u, A_MAX3U) KERNEL(mbcnthi, A_MBCNTHI) KERNEL(lshl1, A_LSHL1) KERNEL(ashr, A_ASHR)
KERNEL(xor_lit, A_XOR_LIT) KERNEL(minu16, A_MINU16) KERNEL(subu16, A_SUBU16) KERNEL(dot4, A_DOT4) KERNEL(mulf, A_MULF)
KERNEL(addf, A_ADDF) KERNEL(cvti, A_CVTI) KERNEL(rndne, A_RNDNE)

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

int main()
{
#define R(n) run(#n, k_##n);
    R(and) R(and_lit) R(and_sgpr) R(add_lit) R(add_inl) R(not) R(cmp32) R(cmp64) R(cmpne32) R(cndmask32) R(cndmask64) R(bitop3)
    R(mad24) R(mul24) R(lshlor) R(add3) R(max3u) R(mbcnthi) R(lshl1) R(ashr) R(xor_lit) R(minu16) R(subu16) R(dot4) R(mulf) R(addf)
    R(cvti) R(rndne)
    return 0;
}
