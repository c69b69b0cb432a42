// Matcher cores for gfx950: 256-bit Hamming brute force as HIP kernels (xor + 64-bit popcount,
// wave64 reductions), plus the host-side greedy passes that turn device-computed distances into
// exactly the matches of the reference's sequential loops.
//
// Reference code replaced (paths relative to the reference root):
//   DescriptorDistance                 modules/ORB/ORBMatcher.cpp:17-31
//   SearchByBow inner loop + accept    modules/ORB/ORBMatcher.cpp:136-198
//   SearchForTriangulation             modules/ORB/ORBMatcher.cpp:448-519
//   SearchForInitialization            modules/ORB/ORBMatcher.cpp:33-116 (+ Frame.cpp:97-127 window query)
//   ComputeThreeMaxima                 modules/ORB/ORBMatcher.cpp:594-622
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/orbm.h"
#include "../../include/orbx.h"
#include "orb_math.h"

typedef unsigned long long u64;

extern "C" const char *orbx_last_error(void);
// error text is shared with the extractor (orbx_api.hip owns the thread-local string)
int orbx_set_error(int code, const std::string &msg);
hipError_t orbx_lds_opt_in(const void *kernel, size_t bytes); // orbx_api.hip: dynamic LDS above 64 KB, per kernel and per device
#define M_TRY(expr)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return orbx_set_error(ORBX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct Desc256 { u64 w[4]; };

__device__ __forceinline__ Desc256 load_desc(const uint8_t *p)
{
    const uint4 lo = *reinterpret_cast<const uint4 *>(p), hi = *reinterpret_cast<const uint4 *>(p + 16);
    Desc256 d;
    d.w[0] = (u64)lo.x | ((u64)lo.y << 32); d.w[1] = (u64)lo.z | ((u64)lo.w << 32);
    d.w[2] = (u64)hi.x | ((u64)hi.y << 32); d.w[3] = (u64)hi.z | ((u64)hi.w << 32);
    return d;
}
__device__ __forceinline__ int ham256(const Desc256 &a, const Desc256 &b)
{
    return __popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) + __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]);
}

// dense na x nb distance matrix; 64x64 tile per 256-thread workgroup, B tile staged in LDS
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t *__restrict__ a, int na,
                                                        const uint8_t *__restrict__ b, int nb,
                                                        uint16_t *__restrict__ out)
{
    __shared__ u64 sb[64 * 4];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    {
        const int col = tid >> 2, part = tid & 3; // 8 bytes each
        u64 v = 0;
        if (j0 + col < nb) v = *reinterpret_cast<const u64 *>(b + (size_t)(j0 + col) * 32 + part * 8);
        sb[col * 4 + part] = v;
    }
    __syncthreads();
    const int r = i0 + (tid >> 2), g = tid & 3;
    if (r >= na) return;
    const Desc256 da = load_desc(a + (size_t)r * 32);
    uint16_t *o = out + (size_t)r * nb + j0 + g * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = g * 16 + c;
        if (j0 + col < nb) {
            const u64 *pb = &sb[col * 4];
            o[c] = (uint16_t)(__popcll(da.w[0] ^ pb[0]) + __popcll(da.w[1] ^ pb[1]) + __popcll(da.w[2] ^ pb[2]) +
                              __popcll(da.w[3] ^ pb[3]));
        }
    }
}

// best / second-best per query row.  A workgroup owns 32 query rows (8 per wave, descriptors held
// in scalar registers) and streams the candidate set through LDS in 128-descriptor chunks, so the
// candidates are read from L2 once per 32 rows instead of once per row.  Per pair: 8 xor + 8
// v_bcnt_u32_b32 (accumulating) + a branch-free (best, second) update.  Partial results merge
// exactly like the sequential strict-'<' scan of ORBMatcher.cpp:155-161: best = smallest
// (distance, index) key, second = 2nd smallest distance.
// D = popcount(x) + acc in one instruction
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t bcnt0(uint32_t x)
{
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d) : "v"(x));
    return d;
}
// median of three = the second smallest
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#define B2_ROWS 8
#define B2_CHUNK 128
__global__ __launch_bounds__(256) void k_best2(const uint8_t *__restrict__ a, size_t a_stride,
                                               const int32_t *__restrict__ na_p, int na_max,
                                               const uint8_t *__restrict__ b, size_t b_stride,
                                               const int32_t *__restrict__ nb_p, int nb_max,
                                               const uint8_t *__restrict__ row_ok, const uint8_t *__restrict__ col_ok,
                                               int32_t *__restrict__ best_idx, uint16_t *__restrict__ best,
                                               uint16_t *__restrict__ second)
{
    __shared__ uint4 sb[2][2][B2_CHUNK]; // [buffer][descriptor half][descriptor]
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = na_p ? min(na_p[p], na_max) : na_max, nb = nb_p ? min(nb_p[p], nb_max) : nb_max;
    const int row0 = blockIdx.x * (4 * B2_ROWS) + wave * B2_ROWS;
    if (blockIdx.x * (4 * B2_ROWS) >= na_max) return;
    const uint8_t *A = a + (size_t)p * a_stride * 32;
    const uint8_t *B = b + (size_t)p * b_stride * 32;
    const uint8_t *ok = col_ok ? col_ok + (size_t)p * b_stride : nullptr;

    uint32_t ar[B2_ROWS][8];
#pragma unroll
    for (int r = 0; r < B2_ROWS; ++r) {
        const int row = min(row0 + r, max(na_max - 1, 0)); // clamped rows are computed but never stored
        const uint32_t *pa = reinterpret_cast<const uint32_t *>(A + (size_t)row * 32);
#pragma unroll
        for (int w = 0; w < 8; ++w) ar[r][w] = __builtin_amdgcn_readfirstlane(pa[w]);
    }
    const uint32_t SENT = (256u << 23) | 0x7FFFFFu;
    // the two smallest (distance << 23 | index) keys seen so far: k1 <= k2
    uint32_t k1[B2_ROWS], k2[B2_ROWS];
#pragma unroll
    for (int r = 0; r < B2_ROWS; ++r) { k1[r] = SENT; k2[r] = SENT; }

    const int n_chunks = (nb + B2_CHUNK - 1) / B2_CHUNK;
    auto stage = [&](int chunk, int buf) {
        const int j = chunk * B2_CHUNK + (tid >> 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (j < nb) v = *reinterpret_cast<const uint4 *>(B + (size_t)j * 32 + (tid & 1) * 16);
        sb[buf][tid & 1][tid >> 1] = v;
    };
    if (n_chunks > 0) stage(0, 0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) stage(c + 1, buf ^ 1);
        // full chunks without a candidate mask skip the per-pair validity select (wave-uniform choice)
        const bool all_valid = !ok && (c + 1) * B2_CHUNK <= nb;
#pragma unroll
        for (int h = 0; h < B2_CHUNK / 64; ++h) {
            const int jl = h * 64 + lane, j = c * B2_CHUNK + jl;
            const uint4 lo = sb[buf][0][jl], hi = sb[buf][1][jl];
            const bool valid = all_valid || (j < nb && (!ok || ok[j]));
#pragma unroll
            for (int r = 0; r < B2_ROWS; ++r) {
                // 8 x (xor, accumulate-popcount): one dependent v_bcnt chain per pair, no add tree
                uint32_t d = bcnt0(lo.x ^ ar[r][0]);
                d = bcnt_acc(lo.y ^ ar[r][1], d); d = bcnt_acc(lo.z ^ ar[r][2], d); d = bcnt_acc(lo.w ^ ar[r][3], d);
                d = bcnt_acc(hi.x ^ ar[r][4], d); d = bcnt_acc(hi.y ^ ar[r][5], d); d = bcnt_acc(hi.z ^ ar[r][6], d);
                d = bcnt_acc(hi.w ^ ar[r][7], d);
                uint32_t k = (d << 23) | (uint32_t)j;
                if (!all_valid) k = valid ? k : SENT;
                k2[r] = med3_u32(k, k1[r], k2[r]); // two smallest of {k, k1, k2}
                k1[r] = min(k, k1[r]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < B2_ROWS; ++r) {
        uint32_t kk = k1[r], k2r = k2[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t ok1 = __shfl_xor(kk, o), ok2 = __shfl_xor(k2r, o);
            // two smallest of {kk, k2r, ok1, ok2} with kk <= k2r and ok1 <= ok2
            k2r = min(min(k2r, ok2), max(kk, ok1));
            kk = min(kk, ok1);
        }
        const uint32_t ss = k2r >> 23;
        const int row = row0 + r;
        if (lane == 0 && row < na_max) {
            const size_t orow = (size_t)p * a_stride + row;
            const bool live = row < na && (!row_ok || row_ok[orow]);
            const uint32_t d1 = kk >> 23;
            // a 256-distance candidate never beats the initial 256 of the reference loop
            best_idx[orow] = (live && d1 < 256) ? (int32_t)(kk & 0x7FFFFFu) : -1;
            best[orow] = live ? (uint16_t)min(d1, 256u) : (uint16_t)256;
            second[orow] = live ? (uint16_t)min(ss, 256u) : (uint16_t)256;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same best / second-best search on the MATRIX pipe (an experiment the VALU kernel above stays the parity twin of).
// For bit vectors a (candidate) and b (query):  hamming(a, b) = popcount(b) - sum_k a_k * (2 b_k - 1),
// so with the candidates unpacked to 0/1 bytes and the queries to +1/-1 bytes one v_mfma_i32_32x32x32_i8 chain (8 steps
// of 32 bits) leaves  popcount(b) - hamming  for 32 candidates x 32 queries in the accumulators, exact in i32.  Orientation: candidates are the rows (A operand), queries the columns (B operand): a lane then owns ONE
// query and 16 candidates per tile, so the running two-smallest-keys state lives in the lane and needs no cross-lane
// step until the very end.  Any bit -> k assignment works as long as both operands use the same one (a dot product does
// not care about the order of its terms); here step t takes descriptor dword t, lane half h and byte i of operand
// dword j take bit 8 i + j + 4 h.
//   workgroup = 8 waves = 256 queries; the candidate set streams through LDS 64 descriptors at a time, unpacked by the
//   whole workgroup (2 VALU per dword), double-buffered; per 32 x 32 tile a wave issues 8 ds_read_b128 + 8 MFMA and
//   3 VALU per accumulator register (key = (256 - acc) << 22 | (tile, register), then the med3 / min update) instead of the
//   19 per pair-lane of the VALU kernel.
// ---------------------------------------------------------------------------------------------
typedef int bm_v4i __attribute__((ext_vector_type(4)));
typedef int bm_v16i __attribute__((ext_vector_type(16)));
#ifndef BM_WAVES
#define BM_WAVES 8
#endif
#ifndef BM_TC
#define BM_TC 128  // candidates per stage (four 32-row MFMA tiles)
#endif
#define BM_MAX_CAND 8160 // 255 tiles of 32: (16 tile + register) must stay below 4096 in the key
#define BM_ROWB 272 // bytes per unpacked candidate in LDS: 256 + 16 so that the 32 rows of a tile fall in different banks
// a * m + c with 24-bit signed factors; c is wave-uniform (a VOP3 instruction of gfx9 reads at most one SGPR)
__device__ __forceinline__ uint32_t bm_mad24(int a, int m, uint32_t c)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "s"(c));
    return d;
}
__global__ __launch_bounds__(BM_WAVES * 64) void k_best2_mfma(const uint8_t *__restrict__ a, size_t a_stride,
                                                              const int32_t *__restrict__ na_p, int na_max,
                                                              const uint8_t *__restrict__ b, size_t b_stride,
                                                              const int32_t *__restrict__ nb_p, int nb_max,
                                                              const uint8_t *__restrict__ row_ok,
                                                              int32_t *__restrict__ best_idx, uint16_t *__restrict__ best,
                                                              uint16_t *__restrict__ second)
{
    __shared__ __align__(16) uint8_t sb[2][BM_TC * BM_ROWB];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int na = na_p ? min(na_p[p], na_max) : na_max, nb = nb_p ? min(nb_p[p], nb_max) : nb_max;
    const int row0 = blockIdx.x * (BM_WAVES * 32) + wave * 32;
    const uint8_t *A = a + (size_t)p * a_stride * 32;
    const uint8_t *B = b + (size_t)p * b_stride * 32;

    // ---- this lane's query as the B operand of the 8 steps: +64 / -64 bytes (the candidates are 0 / 64, so the MFMA leaves
    // 4096 x (popcount(query) - hamming) and the low 12 bits are free for the candidate's place)
    bm_v4i bq[8];
    int pcq = 0;
    {
        const int row = min(row0 + n, max(na_max - 1, 0)); // clamped rows are computed but never stored
        const uint32_t *pq = reinterpret_cast<const uint32_t *>(A + (size_t)row * 32);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t w = na_max > 0 ? pq[t] : 0u;
            pcq += __popc(w);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t d = (w >> (j + 4 * h)) & 0x01010101u;
                bq[t][j] = (int)(0xC0C0C0C0u - (d << 7)); // bit set: +64, clear: -64
            }
        }
    }
    // Keys: key = BIAS + 4096 acc + 15 - r   (acc = popcount(query) - hamming, r = accumulator register) is ONE plain
    // addition of a literal per register (a multiply-add in the slow VOP3 class before).  The larger key is the closer
    // candidate, ties go to the lower register.  Tiles are told apart by keeping the running state (the two LARGEST keys
    // seen) in the frame of the tile being processed: true key = key - 16 tile, so the state moves up by 16 when the wave
    // moves to the next tile (two additions per tile).  16 tile + r stays below 4096: at most 255 tiles = 8160 candidates
    // per problem.  (Seeding the accumulators with BIAS + 15 - r instead makes the keys free, but its 16 extra registers
    // cost the second workgroup per CU: 0.57 ms against 0.49.)
    const uint32_t BIAS = 1u << 30, NONE = 1u << 29; // keys are above 2^30 - 2^21; a state below NONE holds no candidate
    uint32_t k1 = 0, k2 = 0; // the two largest keys, in the frame of tile `frame`
    int frame = 0;

    // ---- staging: thread = (candidate tid / 8 (+ multiples of the workgroup's 8 per wave) of the stage, descriptor dword
    // tid % 8) -> 32 unpacked bytes each
    const int sr = tid >> 3, st = tid & 7;
    auto stage = [&](int step, int buf) {
        uint32_t w[BM_TC / (BM_WAVES * 8)];
#pragma unroll
        for (int q = 0; q < BM_TC / (BM_WAVES * 8); ++q) {
            const int j = step * BM_TC + (BM_WAVES * 8) * q + sr;
            w[q] = j < nb ? reinterpret_cast<const uint32_t *>(B + (size_t)j * 32)[st] : 0u;
        }
#pragma unroll
        for (int q = 0; q < BM_TC / (BM_WAVES * 8); ++q) {
            const uint32_t M = 0x40404040u;
            uint4 lo, hi;
            lo.x = (w[q] << 6) & M; lo.y = (w[q] << 5) & M; lo.z = (w[q] << 4) & M; lo.w = (w[q] << 3) & M;
            hi.x = (w[q] << 2) & M; hi.y = (w[q] << 1) & M; hi.z = w[q] & M;        hi.w = (w[q] >> 1) & M;
            uint4 *dst = reinterpret_cast<uint4 *>(&sb[buf][((BM_WAVES * 8) * q + sr) * BM_ROWB + st * 32]);
            dst[0] = lo; dst[1] = hi;
        }
    };
    const int n_steps = (nb + BM_TC - 1) / BM_TC;
    if (n_steps > 0) stage(0, 0);
    __syncthreads();
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps) stage(s + 1, buf ^ 1);
#pragma unroll
        for (int pair = 0; pair < BM_TC / 64; ++pair) { // two tiles at a time: the second one's MFMAs run under the first one's update
            bm_v16i acc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const uint8_t *src = &sb[buf][((2 * pair + sub) * 32 + n) * BM_ROWB + h * 16];
                const bm_v16i zero = {};
                acc[sub] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const bm_v4i *>(src), bq[0], zero, 0, 0, 0);
#pragma unroll
                for (int t = 1; t < 8; ++t)
                    acc[sub] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const bm_v4i *>(src + t * 32), bq[t], acc[sub], 0, 0, 0);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int tile = (BM_TC / 32) * s + 2 * pair + sub, nvalid = nb - tile * 32; // candidates of this tile that exist
                if (nvalid <= 0) continue;
                const uint32_t adv = 16u * (uint32_t)(tile - frame); // into this tile's frame
                k1 += adv; k2 += adv;
                frame = tile;
                if (nvalid >= 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t k = (uint32_t)acc[sub][r] + (BIAS + 15u - (uint32_t)r);
                        k2 = med3_u32(k, k1, k2);
                        k1 = max(k, k1);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        uint32_t k = (uint32_t)acc[sub][r] + (BIAS + 15u - (uint32_t)r);
                        if ((r & 3) + 8 * (r >> 2) + 4 * h >= nvalid) k = 0; // accumulator row of register r (C/D layout)
                        k2 = med3_u32(k, k1, k2);
                        k1 = max(k, k1);
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- decode (distance, candidate) of both states, make them comparable across the lane pair (n, n + 32) that holds
    // one query's two halves -- final key = (acc + 256) << 13 | (8191 - candidate), 0 = none -- and merge
    auto final_key = [&](uint32_t sk) -> uint32_t {
        if (sk < NONE) return 0u;
        const int t2 = (int)(sk - 16u * (uint32_t)frame - BIAS) - 15;       // 4096 acc - (16 tile + r)
        const int acc = (t2 + 4095) >> 12;                                  // ceil: 0 <= 16 tile + r < 4096
        const uint32_t i16 = (uint32_t)(4096 * acc - t2), r = i16 & 15u, tile = i16 >> 4;
        const uint32_t cand = tile * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)h;
        return ((uint32_t)(acc + 256) << 13) | (8191u - cand);
    };
    uint32_t f1 = final_key(k1), f2 = final_key(k2);
    {
        const uint32_t o1 = __shfl_xor(f1, 32), o2 = __shfl_xor(f2, 32);
        f2 = max(max(f2, o2), min(f1, o1));
        f1 = max(f1, o1);
    }
    const int row = row0 + n;
    if (h == 0 && row < na_max) {
        const size_t orow = (size_t)p * a_stride + row;
        const bool live = row < na && (!row_ok || row_ok[orow]);
        // hamming = popcount(query) - acc
        const uint32_t d1 = f1 == 0 ? 256u : (uint32_t)(pcq - ((int)(f1 >> 13) - 256));
        const uint32_t ss = f2 == 0 ? 256u : (uint32_t)(pcq - ((int)(f2 >> 13) - 256));
        // a 256-distance candidate never beats the initial 256 of the reference loop
        best_idx[orow] = (live && d1 < 256) ? (int32_t)(8191u - (f1 & 8191u)) : -1;
        best[orow] = live ? (uint16_t)min(d1, 256u) : (uint16_t)256;
        second[orow] = live ? (uint16_t)min(ss, 256u) : (uint16_t)256;
    }
}

// ---------------------------------------------------------------------------------------------
// The same search on the FP4 matrix path of gfx950: v_mfma_f32_32x32x64_f8f6f4 with cbsz = blgp = 4 takes 64 k-values
// per instruction in the 32 cycles the i8 form needs for 32 (tools/microbench/fp4_hamming.hip: 32.1 cycles either way,
// products exact).  A bit needs no more than a sign and a one, which FP4 (e2m1) has: candidates unpack to 0.0 / 1.0
// nibbles (0x0 / 0x2), queries to -1.0 / +1.0 (0xA / 0x2), and four MFMAs leave popcount(query) - hamming for 32 candidates
// x 32 queries in f32, exact.  Against the i8 kernel: half the MFMAs, half the LDS bytes per candidate (144-byte rows),
// half the LDS reads per tile and half the unpacking; and
//   * a wave owns 64 queries (two B operand sets, 32 registers): every A fragment read from LDS feeds two MFMAs;
//   * the keys cost nothing: the first MFMA of a chain takes (15 - r) / 4096 per accumulator register r as its C operand,
//     (+ 512), so the accumulator IS the key -- larger = closer, ties to the lower register -- and the update is one med3 and
//     one max per register.  Everything is a multiple of 2^-12 below 2^10: exact in f32.  Tiles are told
//     apart as in the i8 kernel: the running state lives in the frame of the current tile (true key = key - 16 tile / 4096).
// Operand layout (probed with exact data): lane (n = lane & 31, h = lane >> 5), step t: nibble i of operand dword j <-
// bit 4 i + j of descriptor dword 2 t + h, the same rule for both operands.
// ---------------------------------------------------------------------------------------------
typedef int bm_v8i __attribute__((ext_vector_type(8)));
typedef float bm_v16f __attribute__((ext_vector_type(16)));
#ifndef BF_WAVES
#define BF_WAVES 8
#endif
#ifndef BF_TC
#define BF_TC 128   // candidates per stage
#endif
#define BF_ROWB 144 // bytes per unpacked candidate in LDS: 128 + 16, so that 8 consecutive rows cover the 32 banks
__device__ __forceinline__ bm_v16f bf_mfma(bm_v4i a, bm_v4i b, bm_v16f c)
{
    const bm_v8i A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0, 0, 0); // scales 0: the unscaled form
}
template <bool RESIDENT> // true: a fixed grid walks the query blocks (ORBM_VAR_BEST2_RESIDENT); false: one workgroup per block
__global__ __launch_bounds__(BF_WAVES * 64, 4) void k_best2_fp4(const uint8_t *__restrict__ a, size_t a_stride,
                                                             const int32_t *__restrict__ na_p, int na_max,
                                                             const uint8_t *__restrict__ b, size_t b_stride,
                                                             const int32_t *__restrict__ nb_p, int nb_max,
                                                             const uint8_t *__restrict__ row_ok,
                                                             int32_t *__restrict__ best_idx, uint16_t *__restrict__ best,
                                                             uint16_t *__restrict__ second, int blocks_x, int n_blocks)
{
    // (256 candidates per stage for the walking form, half the barriers at its half occupancy, was measured: 0.427 ms alone and
    // 2.20 ms per step either way)
    constexpr int TC = BF_TC;
    __shared__ __align__(16) uint8_t sb[2][TC * BF_ROWB];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // One workgroup per (problem, block of 512 queries) -- or, ORBM_VAR_BEST2_RESIDENT, a grid of a fixed number of workgroups
    // per CU that walk the blocks: the kernel then never holds more than that share of a CU's registers and LDS, whatever else
    // is in flight (a caller that runs it beside latency-bound kernels leaves them the rest).
    // (The walking form keeps its few loop-carried values in scratch between blocks -- four dwords stored before and reloaded
    // after a block's 63 tiles; the one-block form is the kernel as it always was, no scratch.)
    int vb = blockIdx.x;
    // (s_setprio 1 / 3 for the walking form, so that it finishes sooner beside the latency-bound kernels: 2.25 / 2.22 ms per step
    // against 2.19 without)
    do {
    const int p = __builtin_amdgcn_readfirstlane(vb / blocks_x); // (uniform: kept in scalar registers, the kernel has no vector register to spare)
    const int na = na_p ? min(na_p[p], na_max) : na_max, nb = nb_p ? min(nb_p[p], nb_max) : nb_max;
    const int row0 = __builtin_amdgcn_readfirstlane((vb - p * blocks_x) * (BF_WAVES * 64) + wave * 64);
    const uint8_t *A = a + (size_t)p * a_stride * 32;
    const uint8_t *B = b + (size_t)p * b_stride * 32;

    // ---- this lane's two queries (rows row0 + n and row0 + 32 + n) as the B operands of the 4 steps
    // (popcount(query), needed only by the decode after the last tile, is taken there from a second read of the row: carried
    // across the tile loop it cost the walking form two of its spills)
    bm_v4i bq[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int row = min(row0 + 32 * g + n, max(na_max - 1, 0)); // clamped rows are computed but never stored
        const uint32_t *pq = reinterpret_cast<const uint32_t *>(A + (size_t)row * 32);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t w0 = na_max > 0 ? pq[2 * t] : 0u, w1 = na_max > 0 ? pq[2 * t + 1] : 0u;
            const uint32_t w = h ? w1 : w0;
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[g][t][j] = (int)(0x22222222u | (((~w >> j) & 0x11111111u) << 3)); // set: +1.0, clear: -1.0
        }
    }
    bm_v16f seed;
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[r] = 512.0f + (float)(15 - r) * (1.0f / 4096.0f);
    // + 512 keeps every key a positive float (acc is in [-256, 256]).  The update is two v_med3_f32 per register -- the
    // max as med3(k, k1, 3e38): fmaxf (and a med3 with +inf, which the compiler folds into it) would first canonicalise its inputs, and an inline-asm integer med3 / max on the bit
    // patterns hides from the compiler that it reads an MFMA result (the wait states between the two are the compiler's
    // to insert).  A state of 0 holds no candidate; the frame shifts move it by less than 1 in total, far below any key.
    float k1[2] = {0.f, 0.f}, k2[2] = {0.f, 0.f}; // the two largest keys per query, in the frame of tile `frame`
    int frame = 0;

    // ---- staging: thread -> (candidate tid / 8 (+ 64 per round) of the stage, descriptor dword tid % 8) -> 16 unpacked bytes
    // (Measured and dropped: the tiles as a rolling pipeline inside the wave -- one query group's MFMA chain issued interleaved,
    // one MFMA : eight v_med3 via sched_group_barrier, with the key update of the other group's finished accumulators --
    // bit-identical, 0.33-0.34 ms against 0.305: per SIMD the kernel's matrix-pipe cycles (39 %) and VALU cycles (55 %) add up to
    // the whole time either way, i.e. the three-operand v_med3 updates do not issue beside this MFMA, whichever wave they
    // come from.)
    // (Measured and dropped: the stage in two halves -- loads of stage s + 1 before the tiles of stage s, unpack + LDS writes after them.
    // The two dwords that stay live across the tile loop push the kernel past its 128 registers (two spills, scratch set up for
    // every wave): 0.33-0.35 ms per 512 problems against 0.305.)
    const int sr = tid >> 3, st = tid & 7;
    auto stage = [&](int step, int buf) {
        uint32_t w[TC / (BF_WAVES * 8)];
#pragma unroll
        for (int q = 0; q < TC / (BF_WAVES * 8); ++q) {
            const int j = step * TC + (BF_WAVES * 8) * q + sr;
            w[q] = j < nb ? reinterpret_cast<const uint32_t *>(B + (size_t)j * 32)[st] : 0u;
        }
#pragma unroll
        for (int q = 0; q < TC / (BF_WAVES * 8); ++q) {
            const uint32_t M = 0x22222222u;
            uint4 o;
            o.x = (w[q] << 1) & M; o.y = w[q] & M; o.z = (w[q] >> 1) & M; o.w = (w[q] >> 2) & M;
            *reinterpret_cast<uint4 *>(&sb[buf][((BF_WAVES * 8) * q + sr) * BF_ROWB + st * 16]) = o;
        }
    };
    // One tile = 32 candidates x this wave's 64 queries.  Full tiles (every candidate exists) run without any per-row test;
    // the one partial tile a problem can end with is done apart, after the loop of its stage, so that its row masks are
    // not computed (the compiler would hoist them under the MFMAs) for the tiles that do not need them.
    auto do_tile = [&](int buf, int tl, int tile, int nvalid, bool partial) {
        const uint8_t *src = &sb[buf][(tl * 32 + n) * BF_ROWB + h * 16];
        bm_v4i af[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const bm_v4i *>(src + t * 32);
        bm_v16f acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            acc[g] = bf_mfma(af[0], bq[g][0], seed);
#pragma unroll
            for (int t = 1; t < 4; ++t) acc[g] = bf_mfma(af[t], bq[g][t], acc[g]);
        }
        const float adv = (float)(16 * (tile - frame)) * (1.0f / 4096.0f); // into this tile's frame
        frame = tile;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            k1[g] += adv;
            k2[g] += adv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float k = acc[g][r];
                if (partial && (r & 3) + 8 * (r >> 2) + 4 * h >= nvalid) k = 0.f; // accumulator row of register r (C/D layout)
                k2[g] = __builtin_amdgcn_fmed3f(k, k1[g], k2[g]);
                k1[g] = __builtin_amdgcn_fmed3f(k, k1[g], 3.0e38f);
            }
        }
    };
    const int n_steps = (nb + TC - 1) / TC, n_full = nb >> 5;
    if (n_steps > 0) stage(0, 0);
    __syncthreads();
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps) stage(s + 1, buf ^ 1);
        const int t0 = (TC / 32) * s, nt = min(n_full - t0, TC / 32);
        for (int tl = 0; tl < nt; ++tl) do_tile(buf, tl, t0 + tl, 32, false);
        if (s + 1 == n_steps && (nb & 31)) do_tile(buf, n_full - t0, n_full, nb & 31, true);
        __syncthreads();
    }
    // ---- decode (distance, candidate) of both states, make them comparable across the lane pair (n, n + 32) that holds
    // one query's two candidate halves -- final key = (acc + 256) << 13 | (8191 - candidate), 0 = none -- and merge
    auto final_key = [&](float sk) -> uint32_t {
        if (sk < 128.0f) return 0u;
        // 4096 acc + 15 - (16 tile + r), exact
        const int K = (int)((sk - 512.0f - (float)(16 * frame) * (1.0f / 4096.0f)) * 4096.0f);
        const int t2 = K - 15;
        const int acc = (t2 + 4095) >> 12;                                           // ceil: 0 <= 16 tile + r < 4096
        const uint32_t i16 = (uint32_t)(4096 * acc - t2), r = i16 & 15u, tile = i16 >> 4;
        const uint32_t cand = tile * 32u + (r & 3u) + 8u * (r >> 2) + 4u * (uint32_t)h;
        return ((uint32_t)(acc + 256) << 13) | (8191u - cand);
    };
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        uint32_t f1 = final_key(k1[g]), f2 = final_key(k2[g]);
        {
            const uint32_t o1 = __shfl_xor(f1, 32), o2 = __shfl_xor(f2, 32);
            f2 = max(max(f2, o2), min(f1, o1));
            f1 = max(f1, o1);
        }
        const int row = row0 + 32 * g + n;
        if (h == 0 && row < na_max) {
            const size_t orow = (size_t)p * a_stride + row;
            const bool live = row < na && (!row_ok || row_ok[orow]);
            const uint4 *pq = reinterpret_cast<const uint4 *>(A + (size_t)row * 32);
            const uint4 q0 = pq[0], q1 = pq[1];
            const int pcq = __popc(q0.x) + __popc(q0.y) + __popc(q0.z) + __popc(q0.w) + __popc(q1.x) + __popc(q1.y) + __popc(q1.z) + __popc(q1.w);
            // hamming = popcount(query) - acc
            const uint32_t d1 = f1 == 0 ? 256u : (uint32_t)(pcq - ((int)(f1 >> 13) - 256));
            const uint32_t ss = f2 == 0 ? 256u : (uint32_t)(pcq - ((int)(f2 >> 13) - 256));
            // a 256-distance candidate never beats the initial 256 of the reference loop
            best_idx[orow] = (live && d1 < 256) ? (int32_t)(8191u - (f1 & 8191u)) : -1;
            best[orow] = live ? (uint16_t)min(d1, 256u) : (uint16_t)256;
            second[orow] = live ? (uint16_t)min(ss, 256u) : (uint16_t)256;
        }
    }
    vb += gridDim.x; // (the stage loop ends on a barrier: the next block may write the staging buffers at once)
    } while (RESIDENT && vb < n_blocks);
}

// distances for explicit candidate lists; one wave per query
__global__ __launch_bounds__(256) void k_hamming_lists(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                       const int32_t *__restrict__ q_idx,
                                                       const int32_t *__restrict__ c_begin,
                                                       const int32_t *__restrict__ c_len,
                                                       const int32_t *__restrict__ out_begin, int n_queries,
                                                       const int32_t *__restrict__ c_idx, uint16_t *__restrict__ out)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_queries) return;
    const Desc256 da = load_desc(a + (size_t)q_idx[q] * 32);
    const int cb = c_begin[q], n = c_len[q], ob = out_begin[q];
    for (int t = lane; t < n; t += 64) out[ob + t] = (uint16_t)ham256(da, load_desc(b + (size_t)c_idx[cb + t] * 32));
}

// The K smallest (distance << 16 | position-in-list) keys of every query's candidate list, one wave per query.
// Candidates whose byte in `cand_free` is 0 are not considered.  Lists longer than 65535 are not supported here
// (the caller falls back to the full distance lists).  Missing entries are 0xFFFFFFFF.
// K = 8 for short candidate lists, 16 when a list is long (many queries then compete for the same few candidates and
// the sequential pass would otherwise run out of entries and recompute rows on the host: 0.92 ms at K = 4, 0.37 at 8,
// 0.24 at 16 for one dense 2000 x 2000 node)
template <int TOPK>
__global__ __launch_bounds__(256) void k_topk_lists(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                    const int32_t *__restrict__ q_idx,
                                                    const int32_t *__restrict__ c_begin,
                                                    const int32_t *__restrict__ c_len, int n_queries,
                                                    const int32_t *__restrict__ c_idx,
                                                    const uint8_t *__restrict__ cand_free, uint32_t *__restrict__ out)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n_queries) return;
    const Desc256 da = load_desc(a + (size_t)q_idx[q] * 32);
    const int cb = c_begin[q], n = c_len[q];
    uint32_t k[TOPK];
#pragma unroll
    for (int i = 0; i < TOPK; ++i) k[i] = 0xFFFFFFFFu;
    for (int t = lane; t < n; t += 64) {
        const int j = c_idx[cb + t];
        if (cand_free && !cand_free[j]) continue;
        uint32_t v = ((uint32_t)ham256(da, load_desc(b + (size_t)j * 32)) << 16) | (uint32_t)t;
#pragma unroll
        for (int i = 0; i < TOPK; ++i) { // sorted insertion
            const uint32_t lo = min(k[i], v);
            v = max(k[i], v);
            k[i] = lo;
        }
    }
    // wave merge: TOPK rounds of "global minimum, owner pops its head"
#pragma unroll
    for (int r = 0; r < TOPK; ++r) {
        uint32_t m = k[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
        if (lane == 0) out[(size_t)q * TOPK + r] = m;
        if (k[0] == m && m != 0xFFFFFFFFu) { // keys are unique (position), so exactly one lane owns the minimum
#pragma unroll
            for (int i = 0; i + 1 < TOPK; ++i) k[i] = k[i + 1];
            k[TOPK - 1] = 0xFFFFFFFFu;
        }
    }
}

// MapPoint::computeDescriptor (modules/BasicObject/MapPoint.cpp:103-152) for many map points at once: group g holds
// the descriptors [off[g], off[g+1]) of one point's observations; the result is the index (inside the group) of the
// descriptor with the least median distance to all of them (self included, distance 0), first one on ties
// (`median < bestMedian` from 256, :141-146).  median = sorted row[(N-1)/2] (:143).
// One wave per group.  The k-th smallest of a row is found without sorting: distances live in 0..256, so a 9-step
// bisection on the value with a counting pass per step gives it; the group's descriptors sit in LDS.
#define MEDOID_MAX 1024
__global__ __launch_bounds__(64) void k_medoid(const uint8_t *__restrict__ desc, const int32_t *__restrict__ off,
                                               int n_groups, int32_t *__restrict__ best_idx)
{
    __shared__ Desc256 sd[MEDOID_MAX];
    const int g = blockIdx.x, lane = threadIdx.x;
    const int b = off[g], n = min(off[g + 1] - b, MEDOID_MAX);
    if (n <= 0) { if (lane == 0) best_idx[g] = -1; return; }
    for (int i = lane; i < n; i += 64) sd[i] = load_desc(desc + (size_t)(b + i) * 32);
    __syncthreads();
    const int kth = (n - 1) / 2;
    uint32_t bestkey = 0xFFFFFFFFu; // (median << 16 | row): the minimum is the least median, first row on ties
    for (int i = lane; i < n; i += 64) {
        const Desc256 di = sd[i];
        int lo = 0, hi = 256; // smallest v with #{j : d_ij <= v} > kth
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < n; ++j) cnt += ham256(di, sd[j]) <= mid;
            if (cnt > kth) hi = mid; else lo = mid + 1;
        }
        bestkey = min(bestkey, ((uint32_t)lo << 16) | (uint32_t)i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bestkey = min(bestkey, (uint32_t)__shfl_xor((int)bestkey, o));
    // bestMedian starts at 256 with a strict '<': a median of 256 never replaces index 0 (:138-146)
    if (lane == 0) best_idx[g] = (bestkey >> 16) >= 256 ? 0 : (int32_t)(bestkey & 0xFFFF);
}

// ---------------------------------------------------------------------------------------------
// Window searches on a device-resident grid: Frame::grid + Frame / KeyFrame::getFeaturesInArea
// (modules/BasicObject/Frame.cpp:33-51, :97-127; KeyFrame.cpp:181-211) and the Hamming distances of every window,
// without the key points ever visiting the host.
// ---------------------------------------------------------------------------------------------
#define ORBM_GRID 40 // Frame::GRID_SIZE (modules/BasicObject/Frame.h:18)

// Frame.cpp:33-51: the 40-px grid as CSR, cell id = cx * rows + cy (grid[cx][cy]), items of a cell in ascending key-point
// index (the push_back order of :45-50).  One workgroup per frame; the same counting sort as k_frame_post.
#define GB_T 1024 // one workgroup builds the grid of a frame
__global__ __launch_bounds__(GB_T) void k_grid_build(const orbx_kp *__restrict__ kps, int n, int img_w, int img_h, int cols,
                                                    int rows, int32_t *__restrict__ cell_start,
                                                    int32_t *__restrict__ cell_items, int32_t *__restrict__ cell_of,
                                                    int32_t *__restrict__ tmp, int32_t *__restrict__ zero_me)
{
    extern __shared__ int32_t g_lds[]; // start[nc + 1], cursor[nc]
    __shared__ int32_t wave_sum[GB_T / 64];
    const int tid = threadIdx.x, nc = cols * rows;
    if (tid == 0 && zero_me) *zero_me = 0; // the next kernel's pool counter
    int32_t *start = g_lds, *cursor = g_lds + nc + 1;
    for (int c = tid; c < nc; c += GB_T) cursor[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += GB_T) {
        const int x = orb_floor_f(kps[i].x), y = orb_floor_f(kps[i].y); // Frame::PosInGrid (Frame.cpp:89-94)
        int c = -1;
        if (x >= 0 && x < img_w && y >= 0 && y < img_h) {
            c = (x / ORBM_GRID) * rows + (y / ORBM_GRID);
            atomicAdd(&cursor[c], 1);
        }
        cell_of[i] = c;
    }
    __syncthreads();
    const int per = (nc + GB_T - 1) / GB_T, c0 = min(tid * per, nc), c1 = min(c0 + per, nc);
    int local = 0;
    for (int c = c0; c < c1; ++c) local += cursor[c];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - local;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_sum[w];
    for (int c = c0; c < c1; ++c) {
        const int cnt = cursor[c];
        start[c] = base; cell_start[c] = base; cursor[c] = 0;
        base += cnt;
    }
    if (tid == GB_T - 1) { start[nc] = base; cell_start[nc] = base; }
    __syncthreads();
    for (int i = tid; i < n; i += GB_T) { // unordered fill, then the rank inside the cell = number of smaller indices
        const int c = cell_of[i];
        if (c >= 0) tmp[start[c] + atomicAdd(&cursor[c], 1)] = i;
    }
    __syncthreads();
    for (int i = tid; i < n; i += GB_T) {
        const int c = cell_of[i];
        if (c < 0) continue;
        const int b = start[c], e = start[c + 1];
        int rank = 0;
        for (int t = b; t < e; ++t) rank += tmp[t] < i;
        cell_items[b + rank] = i;
    }
}

// One wave per query: getFeaturesInArea(x, y, r, min_level, max_level) in the reference's order (cx outer, cy inner,
// cell items ascending: for one cx the cells minCY..maxCY are one contiguous run of the CSR) and the distance of every
// hit.  out[q * cap + p] = distance << 22 | key-point index for list position p < cap; counts[q] = the full list
// length (> cap: the caller's buffer was too small), -1 for a query that is switched off.
//   strict   : KeyFrame::getFeaturesInArea's `< r` window test (KeyFrame.cpp:204) instead of Frame's `<= r`
//   sigma2   : if non-NULL the fuse's chi-square gate (ORBMatcher.cpp:566-567) drops hits from the list
__global__ __launch_bounds__(256) void k_window_lists(const orbx_kp *__restrict__ kps, const uint8_t *__restrict__ desc,
                                                      const int32_t *__restrict__ cell_start,
                                                      const int32_t *__restrict__ cell_items, int cols, int rows,
                                                      const uint8_t *__restrict__ q_desc, const float *__restrict__ q_xy,
                                                      const float *__restrict__ q_r, const int32_t *__restrict__ q_min,
                                                      const int32_t *__restrict__ q_max, const uint8_t *__restrict__ q_ok,
                                                      int nq, int strict, const float *__restrict__ sigma2, int cap,
                                                      int32_t *__restrict__ counts, uint32_t *__restrict__ out,
                                                      int32_t *__restrict__ pool_total, int32_t *__restrict__ offs)
{
    // pool_total == NULL: list q lives at out + q * cap.  Otherwise the lists are packed back to back into out[0 .. cap):
    // a first sweep counts, the wave reserves its block with one atomic (offs[q]), a second sweep writes.
    // The pool reservation is a returning atomic on ONE address, and those are served one at a time (about 8 ns each on this
    // part: 2000 queries reserving wave by wave stand in line for 16 us, as long as the kernel's own work).  The four waves of
    // a workgroup therefore reserve together: every wave -- also one without a query or with a query that is switched off --
    // reaches the two barriers of the pooled form.
    __shared__ int s_need[4], s_off;
    const int wid = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wid, lane = threadIdx.x & 63;
    const bool live = q < nq && q_ok[min(q, nq - 1)];
    if (!pool_total && !live) { if (q < nq && lane == 0) counts[q] = -1; return; }
    const int qc = min(q, nq - 1);
    const float x = q_xy[2 * qc], y = q_xy[2 * qc + 1], r = q_r[qc];
    const int min_level = q_min[qc], max_level = q_max[qc];
    const int minCX = max(0, orb_floor_f(x - r) / ORBM_GRID), maxCX = min(cols - 1, orb_floor_f(x + r) / ORBM_GRID);
    const int minCY = max(0, orb_floor_f(y - r) / ORBM_GRID), maxCY = min(rows - 1, orb_floor_f(y + r) / ORBM_GRID);
    int pos = 0;
    size_t base = (size_t)q * cap;
    int limit = cap;
    const bool window = live && minCX <= maxCX && minCY <= maxCY;
    {
        const Desc256 dq = load_desc(q_desc + (size_t)qc * 32);
        const bool check = min_level > 0 || max_level >= 0; // beCheckLevel (Frame.cpp:107)
        for (int sweep = pool_total ? 0 : 1; sweep < 2; ++sweep) {
        if (sweep == 1 && pool_total) {
            if (lane == 0) s_need[wid] = pos;
            __syncthreads();
            const int n0 = s_need[0], n1 = s_need[1], n2 = s_need[2], n3 = s_need[3];
            if (threadIdx.x == 0) s_off = (n0 + n1 + n2 + n3) ? atomicAdd(pool_total, n0 + n1 + n2 + n3) : 0;
            __syncthreads();
            const int off = s_off + (wid > 0 ? n0 : 0) + (wid > 1 ? n1 : 0) + (wid > 2 ? n2 : 0);
            if (lane == 0 && q < nq) offs[q] = window ? off : 0;
            base = (size_t)off;
            limit = off + pos <= cap ? pos : 0; // a pool that is too small: nothing is written, the host sees total > cap
            if (pos == 0) break;
            pos = 0;
        }
        if (window)
        for (int cx = minCX; cx <= maxCX; ++cx) {
            const int b = cell_start[cx * rows + minCY], e = cell_start[cx * rows + maxCY + 1];
            for (int t0 = b; t0 < e; t0 += 64) {
                const int t = t0 + lane;
                bool hit = false;
                int j = 0;
                if (t < e) {
                    j = cell_items[t];
                    const orbx_kp kp = kps[j];
                    hit = true;
                    if (check) {
                        if (kp.octave < min_level) hit = false;
                        if (max_level >= 0 && kp.octave > max_level) hit = false;
                    }
                    const float ax = fabsf(kp.x - x), ay = fabsf(kp.y - y);
                    hit = hit && (strict ? (ax < r && ay < r) : (ax <= r && ay <= r));
                    if (hit && sigma2) {
                        const float e2 = (x - kp.x) * (x - kp.x) + (y - kp.y) * (y - kp.y);
                        if ((double)e2 > 5.991 * (double)sigma2[kp.octave]) hit = false;
                    }
                }
                const unsigned long long mk = __ballot(hit);
                const int p = pos + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                if (sweep == 1 && hit && p < limit)
                    out[base + p] = ((uint32_t)ham256(dq, load_desc(desc + (size_t)j * 32)) << 22) | (uint32_t)j;
                pos += (int)__popcll(mk);
            }
        }
        }
    }
    if (lane == 0 && q < nq) counts[q] = live ? pos : -1;
}

// ---------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        const size_t old_cap = cap;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = std::max(std::max(bytes, (size_t)4096), 2 * old_cap); // geometric growth: hipFree is costly
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// pinned host staging: one copy in, one copy out per call of a window search
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 16);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// ---------------------------------------------------------------------------------------------
// The greedy pass of SearchByProjection on the device (ORBMatcher.cpp:229-246 frame -> frame, :379-407 map points ->
// frame).  The reference walks the queries in order and lets each take its closest candidate that is still free, where
// "free" depends on what the queries BEFORE it took -- a sequential chain.  Fixed point instead: every query holds a
// tentative choice; one sweep recomputes all choices at once, query i seeing a candidate as free iff no query j < i
// currently holds it (owner[c] = the smallest holder, one atomicMin per query).  Query 0 depends on nothing, so it is
// final after the first sweep; query i is final one sweep after everything before it is: the sweeps reach the sequential
// result in at most nq + 1 rounds (a handful in practice: a round per link of the longest displacement chain), and a
// sweep that changes nothing proves it.  The rotation histogram, ComputeThreeMaxima (:594-622) and the removal of the
// matches outside the three main bins (:261-271) run in the same kernel.  One workgroup per call: every step is a
// barrier apart.  MODE 0 = frame -> frame (best only, TH_HIGH), MODE 1 = map points -> frame (best / second with their
// levels and the ratio test of :402).
// result[0] = matches, [1] = 1 when the packed lists overflowed the pool (nothing done), [2] = sweeps, [3] = list
// entries; MODE 1: [4] = queries switched off (numOutViewAndBad), [5] = fail1, [6] = fail2.
// ---------------------------------------------------------------------------------------------
#define PR_T 1024
template <int MODE>
__global__ __launch_bounds__(PR_T) void k_projection_resolve(const int32_t *__restrict__ counts, const int32_t *__restrict__ offs,
                                                             const uint32_t *__restrict__ pool, int pool_cap,
                                                             const int32_t *__restrict__ total, int nq, int n2,
                                                             const orbx_kp *__restrict__ kps2, const float *__restrict__ q_angle,
                                                             float nn_ratio, int check_orientation,
                                                             int32_t *__restrict__ frame_mp, int32_t *__restrict__ result)
{
    extern __shared__ int32_t pr_lds[];
    int32_t *owner = pr_lds, *assign = pr_lds + n2; // owner[n2], assign[nq]
    __shared__ int s_hist[ORBM_HISTO_LENGTH], s_keep[3], s_n[4];
    const int tid = threadIdx.x;
    const int tot = *total;
    if (tot > pool_cap) { // a window list longer than the pool: the caller repeats the call with a larger one
        if (tid == 0) { result[0] = 0; result[1] = 1; result[2] = 0; result[3] = tot; }
        return;
    }
    for (int i = tid; i < nq; i += PR_T) assign[i] = -1;
    if (tid < ORBM_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid < 4) s_n[tid] = 0;
    // one sweep's outcome for query i: the candidate it takes (-1 none), and for MODE 1 why not (1 = ratio test, 2 = too far)
    auto choose = [&](int i, int *why) -> int {
        const int n = counts[i];
        *why = 0;
        if (n <= 0) return -1;
        const uint32_t *e = pool + offs[i];
        if (MODE == 0) {
            int best = ORBM_TH_HIGH + 1, idx = -1;
            for (int t = 0; t < n; ++t) {
                const int c = (int)(e[t] & 0x3FFFFFu), d = (int)(e[t] >> 22);
                if (owner[c] < i) continue; // taken before this call (-1) or by an earlier query (:235)
                if (d < best) { best = d; idx = c; }
            }
            return best <= ORBM_TH_HIGH ? idx : -1; // :244
        }
        int best = 256, bestLevel = -1, second = 257, secondLevel = -1, idx = -1;
        for (int t = 0; t < n; ++t) {
            const int c = (int)(e[t] & 0x3FFFFFu), d = (int)(e[t] >> 22);
            if (owner[c] < i) continue; // :383
            if (d < best) { second = best; best = d; secondLevel = bestLevel; bestLevel = kps2[c].octave; idx = c; }
            else if (d < second) { second = d; secondLevel = kps2[c].octave; }
        }
        if (best <= ORBM_TH_HIGH) {
            if (bestLevel == secondLevel && (float)best > nn_ratio * (float)second) { *why = 1; return -1; } // :402
            return idx;
        }
        *why = 2;
        return -1;
    };
    int sweeps = 0;
    for (;;) {
        __syncthreads();
        for (int c = tid; c < n2; c += PR_T) owner[c] = frame_mp[c] != -1 ? -1 : INT_MAX;
        __syncthreads();
        for (int i = tid; i < nq; i += PR_T)
            if (assign[i] >= 0) atomicMin(&owner[assign[i]], i);
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < nq; i += PR_T) {
            int why;
            const int c = choose(i, &why);
            if (c != assign[i]) { assign[i] = c; changed = 1; }
        }
        ++sweeps;
        if (!__syncthreads_or(changed) || sweeps > nq + 1) break;
    }
    // ---- the stable choices become the frame's map points; counters; rotation histogram
    // (owner[] of the last sweep is consistent with the stable assign[]: choose() gives the final outcome again)
    for (int i = tid; i < nq; i += PR_T) {
        int why;
        const int c = choose(i, &why);
        if (MODE == 1) {
            if (counts[i] < 0) atomicAdd(&s_n[1], 1);
            if (why == 1) atomicAdd(&s_n[2], 1);
            if (why == 2) atomicAdd(&s_n[3], 1);
        }
        if (c < 0) continue;
        atomicAdd(&s_n[0], 1);
        if (MODE == 0 && check_orientation) {
            const float factor = 1.f / ORBM_HISTO_LENGTH; // :205: only bins 0..12 fill up
            float rot = ORB_FSUB(q_angle[i], kps2[c].angle);
            if (rot < 0.f) rot = ORB_FADD(rot, 360.f);
            int bin = orb_round_f(ORB_FMUL(rot, factor));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            atomicAdd(&s_hist[bin], 1);
        }
    }
    __syncthreads();
    if (MODE == 0 && check_orientation && tid == 0) { // ComputeThreeMaxima (:594-622)
        int max1 = 0, max2 = -1, max3 = -2, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            const int n = s_hist[i];
            if (n > max1) { max3 = max2; max2 = max1; max1 = n; i3 = i2; i2 = i1; i1 = i; }
            else if (n > max2) { max3 = max2; max2 = n; i3 = i2; i2 = i; }
            else if (n > max3) { max3 = n; i3 = i; }
        }
        if (max2 < max1 / 10) { i2 = -1; i3 = -1; }
        else if (max3 < max1 / 10) i3 = -1;
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    for (int i = tid; i < nq; i += PR_T) {
        const int c = assign[i];
        if (c < 0) continue;
        bool keep = true;
        if (MODE == 0 && check_orientation) {
            const float factor = 1.f / ORBM_HISTO_LENGTH;
            float rot = ORB_FSUB(q_angle[i], kps2[c].angle);
            if (rot < 0.f) rot = ORB_FADD(rot, 360.f);
            int bin = orb_round_f(ORB_FMUL(rot, factor));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            keep = bin == s_keep[0] || bin == s_keep[1] || bin == s_keep[2];
            if (!keep) atomicSub(&s_n[0], 1);
        }
        if (keep) frame_mp[c] = i; // :245 / :406
    }
    __syncthreads();
    if (tid == 0) {
        result[0] = s_n[0]; result[1] = 0; result[2] = sweeps; result[3] = tot;
        if (MODE == 1) { result[4] = s_n[1]; result[5] = s_n[2]; result[6] = s_n[3]; }
    }
}
// ---------------------------------------------------------------------------------------------
// SearchForInitialization on the device (ORBMatcher.cpp:33-116).  The reference walks frame 1's level-0 features in order; a
// feature skips every candidate that a feature BEFORE it matched at a distance <= its own (`matchedDistance[idx2] <= dist`, :63 --
// for the best AND the second best), takes the closest of the rest if it passes TH_LOW and the ratio test, and STEALS the
// candidate from whoever held it (:75-78); the loser is not tried again.  The same fixed point as k_projection_resolve: every
// query holds a tentative (candidate, distance); a sweep recomputes all of them at once, query i seeing for candidate c
// matchedDistance = the smallest distance among the queries j < i that currently claim c -- found by walking c's claimant
// chain (head[c] -> next[j] -> ..., rebuilt every sweep with one atomic exchange per claimant; chains are a handful long).
// Query 0 depends on nothing, query i is final one sweep after everything before it: a sweep that changes nothing is the
// sequential result.  Then, as the reference: every ACCEPTED query entered the rotation histogram when it was accepted, also
// the ones robbed later (:84-91: their entries stay and count in ComputeThreeMaxima); a candidate's last claimant holds it;
// matches outside the three main bins go (:96-109); vecPreMatched takes the matched positions (:112-114).
// result[0] = matches, [1] = 1 when the packed lists overflowed the pool (nothing written), [2] = sweeps, [3] = list entries.
// ---------------------------------------------------------------------------------------------
__global__ void k_init_queries(const orbx_kp *__restrict__ kps1, int n1, float window, uint8_t *__restrict__ q_ok,
                               float *__restrict__ q_r, int32_t *__restrict__ q_lv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const int oct = kps1[i].octave;
    q_ok[i] = oct <= 0;  // :48 `if (level1 > 0) continue`
    q_r[i] = window;
    q_lv[i] = oct;       // getFeaturesInArea(..., level1, level1) (:50-52)
}
__global__ __launch_bounds__(PR_T) void k_init_resolve(const int32_t *__restrict__ counts, const int32_t *__restrict__ offs,
                                                       const uint32_t *__restrict__ pool, int pool_cap,
                                                       const int32_t *__restrict__ total, int n1, int n2,
                                                       const orbx_kp *__restrict__ kps1, const orbx_kp *__restrict__ kps2,
                                                       float nn_ratio, int check_orientation, int max_sweeps, int group_lanes,
                                                       int32_t *__restrict__ matches12, float *__restrict__ pre, int32_t *__restrict__ result)
{
    extern __shared__ int32_t ir_lds[];
    // head[n2]; next, the claims (a_c, a_d) and a sweep's new claims (n_c, n_d) [n1 each]
    int32_t *head = ir_lds, *next = ir_lds + n2, *a_c = next + n1, *a_d = a_c + n1, *n_c = a_d + n1, *n_d = n_c + n1;
    __shared__ int s_hist[ORBM_HISTO_LENGTH], s_keep[3], s_n;
    const int tid = threadIdx.x;
    const int tot = *total;
    for (int i = tid; i < n1; i += PR_T) matches12[i] = -1; // :37
    if (tot > pool_cap) {
        if (tid == 0) { result[0] = 0; result[1] = 1; result[2] = 0; result[3] = tot; }
        return;
    }
    for (int i = tid; i < n1; i += PR_T) { a_c[i] = -1; a_d[i] = 0; }
    if (tid < ORBM_HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_n = 0;
    // G lanes per query share the query's window list (1, 4, 16 or 64: a 100-px window over 2000 level-0 features can hold
    // hundreds of candidates, and with one thread per query the longest list sets the sweep's time; with a whole wave per query
    // short lists leave most lanes idle and a lane's claimant chains set the wave's time).  The reference's strict first-wins
    // updates (:64-71) are the two smallest keys (distance << 22 | list position) of the candidates that pass :63.
    const int G = group_lanes > 0 ? group_lanes : (tot >= 1024 * n1 ? 4 : tot >= 48 * n1 ? 16 : tot >= 12 * n1 ? 4 : 1); // (profiles/r06_match_latency.txt)
    const int lane = tid & (G - 1), wv = tid / G;
    auto choose = [&](int i, int *dist_out) -> int { // every lane of the group returns the same answer
        const int n = counts[i];
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        const uint32_t *e = pool + (n > 0 ? offs[i] : 0);
        for (int t0 = lane; t0 < n; t0 += 4 * G) {
            // four list entries requested before the first is used: the lists are re-read from memory in every sweep, and one
            // workgroup has few loads in flight (a dependent load per entry: 0.7 us per entry and lane)
            uint32_t ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ev[u] = t0 + u * G < n ? e[t0 + u * G] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u * G;
                if (t >= n) break;
                const int c = (int)(ev[u] & 0x3FFFFFu), d = (int)(ev[u] >> 22);
                int md = INT_MAX; // matchedDistance[c] as query i finds it: the closest claim of a query before it
                for (int j = head[c]; j >= 0; j = next[j])
                    if (j < i) md = min(md, a_d[j]);
                if (md <= d) continue; // :63
                const uint32_t k = ((uint32_t)d << 22) | (uint32_t)t;
                k2 = min(k2, max(k1, k));
                k1 = min(k1, k);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) {
            const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, o), o2 = (uint32_t)__shfl_xor((int)k2, o);
            k2 = min(max(k1, o1), min(k2, o2));
            k1 = min(k1, o1);
        }
        const int best = k1 == 0xFFFFFFFFu ? INT_MAX - 1 : (int)(k1 >> 22), second = k2 == 0xFFFFFFFFu ? INT_MAX : (int)(k2 >> 22);
        const int idx = k1 == 0xFFFFFFFFu ? -1 : (int)(e[k1 & 0x3FFFFFu] & 0x3FFFFFu);
        *dist_out = best;
        // :74 (second == INT_MAX: the product is 2.1e9 * ratio in float and cvRound saturates the way lrint does on the host)
        return (best <= ORBM_TH_LOW && best < orb_round_f((float)second * nn_ratio)) ? idx : -1;
    };
    int sweeps = 0;
    for (;;) {
        __syncthreads();
        for (int c = tid; c < n2; c += PR_T) head[c] = -1;
        __syncthreads();
        for (int i = tid; i < n1; i += PR_T)
            if (a_c[i] >= 0) next[i] = atomicExch(&head[a_c[i]], i);
        __syncthreads();
        int changed = 0;
        for (int i = wv; i < n1; i += PR_T / G) { // every query reads the OLD claims: the new ones become visible behind a barrier
            int d = 0;
            const int c = choose(i, &d);
            if (lane == 0) { n_c[i] = c; n_d[i] = d; }
        }
        __syncthreads();
        for (int i = tid; i < n1; i += PR_T)
            if (n_c[i] != a_c[i] || (n_c[i] >= 0 && n_d[i] != a_d[i])) { a_c[i] = n_c[i]; a_d[i] = n_d[i]; changed = 1; }
        ++sweeps;
        if (!__syncthreads_or(changed)) break;
        // One workgroup runs the whole search, and a sweep walks every query's window list times the claimant chains.  The fixed
        // point is reached after 3 sweeps on two extracted views and 5 on a crowded scene of near-duplicates (the proof bounds it
        // by n1 + 1); a scene that needs more than ORBM_INIT_MAX_SWEEPS hands the search back instead of holding a CU for
        // milliseconds: result[1] = 2, nothing written (matches12 all -1, pre untouched) -- the host entry point takes it.
        if (sweeps >= max_sweeps) {
            if (tid == 0) { result[0] = 0; result[1] = 2; result[2] = sweeps; result[3] = tot; }
            return;
        }
    }
    // (head / next of the last sweep belong to the stable claims)
    // ---- rotation histogram of every accepted query (:84-91), then the holders: a candidate's LAST claimant (:75-81)
    for (int i = tid; i < n1; i += PR_T) {
        const int c = a_c[i];
        if (c < 0) continue;
        if (check_orientation) {
            const float factor = 1.f / ORBM_HISTO_LENGTH;
            float rot = ORB_FSUB(kps1[i].angle, kps2[c].angle);
            if (rot < 0.f) rot = ORB_FADD(rot, 360.f);
            int bin = orb_round_f(ORB_FMUL(rot, factor));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            atomicAdd(&s_hist[bin], 1);
        }
    }
    __syncthreads();
    if (check_orientation && tid == 0) { // ComputeThreeMaxima (:594-622)
        int max1 = 0, max2 = -1, max3 = -2, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            const int n = s_hist[i];
            if (n > max1) { max3 = max2; max2 = max1; max1 = n; i3 = i2; i2 = i1; i1 = i; }
            else if (n > max2) { max3 = max2; max2 = n; i3 = i2; i2 = i; }
            else if (n > max3) { max3 = n; i3 = i; }
        }
        if (max2 < max1 / 10) { i2 = -1; i3 = -1; }
        else if (max3 < max1 / 10) i3 = -1;
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    for (int i = tid; i < n1; i += PR_T) {
        const int c = a_c[i];
        if (c < 0) continue;
        bool holds = true; // no later query claims c
        for (int j = head[c]; j >= 0; j = next[j]) holds = holds && j <= i;
        if (!holds) continue;
        if (check_orientation) {
            const float factor = 1.f / ORBM_HISTO_LENGTH;
            float rot = ORB_FSUB(kps1[i].angle, kps2[c].angle);
            if (rot < 0.f) rot = ORB_FADD(rot, 360.f);
            int bin = orb_round_f(ORB_FMUL(rot, factor));
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) continue; // :100-106
        }
        matches12[i] = c;
        pre[2 * i] = kps2[c].x; pre[2 * i + 1] = kps2[c].y; // :112-114
        atomicAdd(&s_n, 1);
    }
    __syncthreads();
    if (tid == 0) { result[0] = s_n; result[1] = 0; result[2] = sweeps; result[3] = tot; }
}

// window levels of the two searches: octave - 1 .. octave + hi (:226-229 hi = 1, :367-369 hi = 0)
__global__ void k_projection_levels(const int32_t *__restrict__ lv, int n, int hi, int32_t *__restrict__ lo_out, int32_t *__restrict__ hi_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { lo_out[i] = lv[i] - 1; hi_out[i] = lv[i] + hi; }
}


// ---------------------------------------------------------------------------------------------
// SearchByBow / SearchForTriangulation on device-resident frame records (ORBMatcher.cpp:136-185, :448-506): the node join,
// the candidate lists, the greedy pass, the rotation histogram and ComputeThreeMaxima without a host hop.
//   k_bow_queries   one workgroup: joins the two FeatureVectors (binary search of every node of side 1 in side 2), counts the
//                   usable features per shared node, prefix sum, and writes one query per (shared node, usable feature) in
//                   the reference's order -- nodes ascending, features of a node in list order
//   k_topk_lists    (above) the K closest initially-free candidates of every query, sorted
//   k_bow_resolve   a feature of side 2 belongs to ONE node, so queries of different nodes never compete: one workgroup per
//                   node runs k_projection_resolve's fixed point on the node's queries (query i sees a candidate as free iff
//                   no query j < i of the node currently holds it); a query whose sorted list is used up before it is decided
//                   rescans its node with distances computed on the spot (rare: the list's last distance bounds the unseen
//                   ones, as in the host replay).  Accepted matches go to frame_mp / matches12 and into the histogram.
//   k_bow_finish    ComputeThreeMaxima and the removal of the matches outside the three main bins; the counters.
// MODE 0 = SearchByBow (best / second, :164 ratio test in float), MODE 1 = SearchForTriangulation (best only, strict < TH_LOW,
// `bestIdx2 > 0`, :484).
// ---------------------------------------------------------------------------------------------
#define BQ_T 1024
#define BR_MAX_NODE 4096 // features of one node on either side the resolve kernel keeps state for in LDS (a dense single node: 2000)
struct BowFv { const uint32_t *nodes; const int32_t *off; const uint32_t *idx; const int32_t *n_nodes; }; // a FeatureVector as orbv leaves it
__global__ __launch_bounds__(BQ_T) void k_bow_queries(BowFv f1, BowFv f2, const uint8_t *__restrict__ mask1, int mask_polarity, int n1,
                                                      int max_nodes, int32_t *__restrict__ node_p2, int32_t *__restrict__ node_qbegin,
                                                      int32_t *__restrict__ q_idx, int32_t *__restrict__ c_begin, int32_t *__restrict__ c_len,
                                                      int32_t *__restrict__ q_node, int32_t *__restrict__ n_queries,
                                                      int32_t *__restrict__ result)
{
    // The queries in the reference's order -- node by node, a node's features in stored order -- are the usable entries of
    // side 1's CSR in ENTRY order, so the work is split over the entries, not over the nodes (one thread per node walked a
    // 2000-feature node alone: 0.6 ms for a frame whose features share one node): a thread finds the node of its entry by
    // bisection of the offsets, a block-wide prefix sum of the usable flags numbers the queries.
    __shared__ int s_part[BQ_T];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    const int nn1 = min(*f1.n_nodes, max_nodes), nn2 = *f2.n_nodes;
    for (int p1 = tid; p1 < nn1; p1 += BQ_T) { // node of side 2 with the same id (:180-183), per node of side 1
        const uint32_t id = f1.nodes[p1];
        int lo = 0, hi = nn2;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (f2.nodes[m] < id) lo = m + 1; else hi = m; }
        node_p2[p1] = lo < nn2 && f2.nodes[lo] == id ? lo : -1;
    }
    if (tid == 0) s_base = 0;
    __syncthreads(); // (node_p2 is read below by other threads of this one workgroup: the barrier orders the global writes for them)
    const int n_entries = nn1 > 0 ? f1.off[nn1] : 0;
    for (int a0 = 0; a0 < n_entries; a0 += BQ_T) {
        const int a = a0 + tid;
        int p1 = -1, p2 = -1, i1 = 0;
        bool use = false;
        if (a < n_entries) {
            int lo = 0, hi = nn1; // the node whose entry range holds a: the last p with off[p] <= a
            while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (f1.off[m] <= a) lo = m; else hi = m; }
            p1 = lo;
            p2 = node_p2[p1];
            i1 = (int)f1.idx[a];
            use = p2 >= 0 && ((mask1[i1] != 0) == (mask_polarity != 0)); // usable features of side 1 that side 2 shares the node of
        }
        s_part[tid] = use;
        __syncthreads();
        for (int off = 1; off < BQ_T; off <<= 1) {
            const int v = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        const int q = s_base + s_part[tid] - (int)use;
        if (a < n_entries) {
            if (a == f1.off[p1]) node_qbegin[p1] = q; // first query of the node (or where it would be)
            if (use && q < n1) { q_idx[q] = i1; c_begin[q] = f2.off[p2]; c_len[q] = f2.off[p2 + 1] - f2.off[p2]; q_node[q] = p1; }
        }
        __syncthreads();
        if (tid == BQ_T - 1) s_base += s_part[tid];
        __syncthreads();
    }
    // A node of side 1 WITHOUT entries (a caller-built CSR may hold one; orbv_transform_device never writes one) got no first-query
    // index above: it takes the next node's, so that every node's extent is [node_qbegin[p], node_qbegin[p + 1]) -- empty for such
    // a node -- here and in k_bow_resolve alike, and no stale scratch of an earlier call is ever read as a begin.
    const int total = min(s_base, n1);
    for (int p1 = tid; p1 < nn1; p1 += BQ_T) {
        if (f1.off[p1 + 1] > f1.off[p1]) continue;
        int nx = p1 + 1; // the next node that has entries (its begin was written by the entry loop, and is not written here)
        while (nx < nn1 && f1.off[nx + 1] <= f1.off[nx]) ++nx;
        node_qbegin[p1] = nx < nn1 ? node_qbegin[nx] : total;
    }
    if (tid == 0) node_qbegin[nn1] = total;
    __syncthreads();
    // A shared node with more features on either side than k_bow_resolve keeps state for (BR_MAX_NODE) cannot be resolved on the
    // device.  It is found HERE, before anything is written: the call then reports result[1] = 1 with no query at all, so that
    // k_topk_lists_n, k_bow_resolve and k_bow_finish have nothing to do and frame_mp / matches12 stay exactly as the caller
    // passed them -- the host entry point can still reproduce the reference's loop on them (the guarantee the projection
    // searches give on overflow).  The node extents are the ones k_bow_resolve computes.
    int too_big = 0;
    for (int p1 = tid; p1 < nn1; p1 += BQ_T) {
        const int p2 = node_p2[p1];
        if (p2 < 0) continue;
        const int qb = node_qbegin[p1], qe = min(node_qbegin[p1 + 1], total);
        if (qe - qb > BR_MAX_NODE || f2.off[p2 + 1] - f2.off[p2] > BR_MAX_NODE) too_big = 1;
    }
    too_big = __syncthreads_or(too_big);
    if (tid == 0) {
        *n_queries = too_big ? 0 : total;
        if (too_big) result[1] = 1;
    }
}

// the K smallest keys of every query's list with the query count on the device (k_topk_lists reads it from the host)
template <int TOPK>
__global__ __launch_bounds__(256) void k_topk_lists_n(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                      const int32_t *__restrict__ q_idx, const int32_t *__restrict__ c_begin,
                                                      const int32_t *__restrict__ c_len, const int32_t *__restrict__ n_queries,
                                                      const int32_t *__restrict__ c_idx, const int32_t *__restrict__ frame_mp,
                                                      const uint8_t *__restrict__ busy2, uint32_t *__restrict__ out)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= *n_queries) return;
    const Desc256 da = load_desc(a + (size_t)q_idx[q] * 32);
    const int cb = c_begin[q], n = c_len[q];
    uint32_t k[TOPK];
#pragma unroll
    for (int i = 0; i < TOPK; ++i) k[i] = 0xFFFFFFFFu;
    for (int t = lane; t < n; t += 64) {
        const int j = c_idx[cb + t];
        if (frame_mp ? frame_mp[j] != -1 : busy2[j] != 0) continue; // not free when the call starts (:150 / :466)
        uint32_t v = ((uint32_t)ham256(da, load_desc(b + (size_t)j * 32)) << 16) | (uint32_t)min(t, 65535);
#pragma unroll
        for (int i = 0; i < TOPK; ++i) {
            const uint32_t lo = min(k[i], v);
            v = max(k[i], v);
            k[i] = lo;
        }
    }
#pragma unroll
    for (int r = 0; r < TOPK; ++r) {
        uint32_t m = k[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
        if (lane == 0) out[(size_t)q * TOPK + r] = m;
        if (k[0] == m && m != 0xFFFFFFFFu) {
#pragma unroll
            for (int i = 0; i + 1 < TOPK; ++i) k[i] = k[i + 1];
            k[TOPK - 1] = 0xFFFFFFFFu;
        }
    }
}

#define BR_T 256
#define BR_TOPK 8
template <int MODE>
__global__ __launch_bounds__(BR_T) void k_bow_resolve(BowFv f1, BowFv f2, int max_nodes, const int32_t *__restrict__ node_p2,
                                                      const int32_t *__restrict__ node_qbegin, const int32_t *__restrict__ n_queries,
                                                      const int32_t *__restrict__ q_idx, const uint32_t *__restrict__ topk,
                                                      const uint8_t *__restrict__ desc1, const uint8_t *__restrict__ desc2,
                                                      const orbx_kp *__restrict__ kps1, const orbx_kp *__restrict__ kps2,
                                                      const uint8_t *__restrict__ busy2, float nn_ratio, int check_orientation,
                                                      int32_t *__restrict__ frame_mp, int32_t *__restrict__ matches12,
                                                      int32_t *__restrict__ hist, int32_t *__restrict__ match_list, int32_t *__restrict__ result)
{
    __shared__ int32_t s_owner[BR_MAX_NODE], s_assign[BR_MAX_NODE];
    const int tid = threadIdx.x;
    const int nn1 = min(*f1.n_nodes, max_nodes), nq_all = *n_queries;
    for (int p1 = blockIdx.x; p1 < nn1; p1 += gridDim.x) {
        const int p2 = node_p2[p1];
        if (p2 < 0) continue;
        const int qb = node_qbegin[p1], qe = min(node_qbegin[p1 + 1], nq_all); // (k_bow_queries wrote a begin for EVERY node and for nn1)
        const int nq = qe - qb;
        if (nq <= 0) continue;
        const int cb = f2.off[p2], nc = f2.off[p2 + 1] - cb;
        if (nq > BR_MAX_NODE || nc > BR_MAX_NODE) { // never taken (k_bow_queries leaves no query at all when a node is this large); reported if it ever is
            if (tid == 0) result[1] = 1;
            continue;
        }
        __syncthreads(); // (the previous node's state is no longer read)
        // initially free candidates: SearchByBow -- no map point yet (:150); triangulation -- no map point on side 2 (:466)
        for (int t = tid; t < nc; t += BR_T) {
            const int j = (int)f2.idx[cb + t];
            const bool free0 = MODE == 0 ? frame_mp[j] == -1 : busy2[j] == 0;
            s_owner[t] = free0 ? INT_MAX : -1;
        }
        for (int i = tid; i < nq; i += BR_T) s_assign[i] = -1;
        // one sweep's outcome for query i of the node: the list position it takes, or -1
        auto choose = [&](int i) -> int {
            const uint32_t *e = topk + (size_t)(qb + i) * BR_TOPK;
            int best = MODE == 0 ? 256 : ORBM_TH_LOW, second = 256, bt = -1, found = 0, last_dd = 0;
            bool exhausted = false, resolved = false;
            for (int r = 0; r < BR_TOPK && found < (MODE == 0 ? 2 : 1); ++r) {
                const uint32_t key = e[r];
                if (key == 0xFFFFFFFFu) { exhausted = true; break; }
                const int t = (int)(key & 0xFFFF), dd = (int)(key >> 16);
                last_dd = dd;
                if (s_owner[t] < i) continue; // taken by an earlier query of this call (:150 / :466)
                if (found == 0) { if (dd < best) { best = dd; bt = t; } }
                else second = min(dd, 256);
                ++found;
            }
            if (MODE == 0) {
                resolved = found == 2 || exhausted || (found == 1 && best > ORBM_TH_LOW);
                if (!resolved && found == 1 && (float)best < nn_ratio * (float)min(last_dd, 256)) { second = min(last_dd, 256); resolved = true; }
            } else {
                resolved = found == 1 || exhausted || last_dd >= ORBM_TH_LOW;
            }
            if (!resolved) { // the sorted list is used up: the whole node, distances computed here
                const Desc256 da = load_desc(desc1 + (size_t)q_idx[qb + i] * 32);
                best = MODE == 0 ? 256 : ORBM_TH_LOW; second = 256; bt = -1;
                for (int t = 0; t < nc; ++t) {
                    if (s_owner[t] < i) continue;
                    const int dd = ham256(da, load_desc(desc2 + (size_t)f2.idx[cb + t] * 32));
                    if (dd < best) { second = best; best = dd; bt = t; }
                    else if (MODE == 0 && dd < second) second = dd;
                }
            }
            if (MODE == 0) return (bt >= 0 && best <= ORBM_TH_LOW && (float)best < nn_ratio * (float)second) ? bt : -1; // :164
            return (bt >= 0 && (int)f2.idx[cb + bt] > 0) ? bt : -1; // :484 -- feature 0 of key frame 2 is never accepted
        };
        int sweeps = 0;
        for (;;) {
            __syncthreads();
            int changed = 0;
            for (int i = tid; i < nq; i += BR_T) {
                const int t = choose(i);
                if (t != s_assign[i]) { s_assign[i] = t; changed = 1; }
            }
            ++sweeps;
            const int any = __syncthreads_or(changed);
            if (!any || sweeps > nq + 1) break;
            // owner[t] = the first query that holds t (queries only ever compete inside their node)
            for (int t = tid; t < nc; t += BR_T) if (s_owner[t] >= 0) s_owner[t] = INT_MAX;
            __syncthreads();
            for (int i = tid; i < nq; i += BR_T)
                if (s_assign[i] >= 0) atomicMin(&s_owner[s_assign[i]], i);
        }
        // the stable choices become matches; rotation histogram (:170-174 / :491-495, the reference's 1/30 factor)
        for (int i = tid; i < nq; i += BR_T) {
            const int t = s_assign[i];
            if (t < 0) continue;
            const int i1 = q_idx[qb + i], i2 = (int)f2.idx[cb + t];
            if (MODE == 0) frame_mp[i2] = i1; else matches12[i1] = i2;
            int bin = -1;
            if (check_orientation) {
                const float factor = 1.f / ORBM_HISTO_LENGTH;
                float rot = ORB_FSUB(kps1[i1].angle, kps2[i2].angle);
                if (rot < 0.f) rot = ORB_FADD(rot, 360.f);
                bin = orb_round_f(ORB_FMUL(rot, factor));
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                atomicAdd(&hist[bin], 1);
            }
            const int at = atomicAdd(&result[0], 1);
            match_list[2 * at] = MODE == 0 ? i2 : i1;
            match_list[2 * at + 1] = bin;
        }
        if (tid == 0) atomicMax(&result[2], sweeps);
    }
}

// ComputeThreeMaxima (:594-622) over the call's histogram and the removal of the matches outside the three main bins
template <int MODE>
__global__ __launch_bounds__(256) void k_bow_finish(int check_orientation, const int32_t *__restrict__ hist, const int32_t *__restrict__ match_list,
                                                    int32_t *__restrict__ frame_mp, int32_t *__restrict__ matches12, int32_t *__restrict__ result)
{
    __shared__ int s_keep[3], s_removed;
    const int tid = threadIdx.x, n = result[0];
    if (tid == 0) {
        s_removed = 0;
        int max1 = 0, max2 = -1, max3 = -2, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            const int v = hist[i];
            if (v > max1) { max3 = max2; max2 = max1; max1 = v; i3 = i2; i2 = i1; i1 = i; }
            else if (v > max2) { max3 = max2; max2 = v; i3 = i2; i2 = i; }
            else if (v > max3) { max3 = v; i3 = i; }
        }
        if (max2 < max1 / 10) { i2 = -1; i3 = -1; }
        else if (max3 < max1 / 10) i3 = -1;
        s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
    }
    __syncthreads();
    if (check_orientation)
        for (int k = tid; k < n; k += 256) {
            const int bin = match_list[2 * k + 1];
            if (bin == s_keep[0] || bin == s_keep[1] || bin == s_keep[2]) continue;
            if (MODE == 0) frame_mp[match_list[2 * k]] = -1; else matches12[match_list[2 * k]] = -1;
            atomicAdd(&s_removed, 1);
        }
    __syncthreads();
    if (tid == 0) { result[3] = n; result[0] = n - s_removed; }
}

struct orbm_ctx {
    int device;
    hipStream_t stream;
    bool null_pending = false; // a device call was enqueued on stream 0 (NULL): the next host-pointer call and destroy wait for it
    DevBuf a, b, out, q_idx, c_begin, c_len, out_begin, c_idx, row_ok, col_ok, bidx, bbest, bsecond;
    DevBuf w_in, w_out, w_grid; // window searches: staged inputs, lists, CSR grid + scratch
    PinBuf h_in, h_out;
    int window_on_device = 1;   // ORBM_VAR_WINDOW = 1 keeps the host grid (the parity twin of the device lists)
    int best2_variant = 0;      // ORBM_VAR_BEST2: 0 = fp4, 1 = i8, 2 = valu
    int init_lanes = 0;         // ORBM_VAR_INIT_LANES: lanes per query of k_init_resolve: 0 = by list length (default), 1, 4, 16, 64
    int best2_resident = 0;     // ORBM_VAR_BEST2_RESIDENT: k_best2_fp4 as a grid of this many workgroups per CU (0 = one per block of queries)
    int n_cus = 256;            // the device's CU count (set at create)
    size_t window_last_total = 0; // candidates the previous window search returned (sizes the first copy-out)
};

// A host-pointer entry point starts here: the handle's device and -- the handle's stream being NON-BLOCKING, so that a second
// thread's handle or a legacy-stream operation anywhere in the process never orders against it (include/orbx.h, "Streams") -- a
// wait for NULL-stream device calls of THIS handle that may still be using its scratch.
static hipError_t host_call_begin(orbm_ctx *c)
{
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess && c->null_pending) { e = hipStreamSynchronize((hipStream_t)0); c->null_pending = false; }
    return e;
}

extern "C" int orbm_create(int device, orbm_t **out)
{
    if (!out) return orbx_set_error(ORBX_E_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) return orbx_set_error(ORBX_E_ARG, "device ordinal out of range");
    orbm_ctx *c = new orbm_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { // host-pointer calls only (include/orbx.h, "Streams")
        delete c;
        return orbx_set_error(ORBX_E_NO_DEVICE, "stream creation failed");
    }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->n_cus = cus; }
    *out = c;
    return ORBX_OK;
}

extern "C" int orbm_set_variant(orbm_t *c, int which, int value)
{
    if (!c) return orbx_set_error(ORBX_E_ARG, "null handle");
    if (which == ORBM_VAR_BEST2 && value >= 0 && value <= 2) { c->best2_variant = value; return ORBX_OK; }
    if (which == ORBM_VAR_WINDOW && (value == 0 || value == 1)) { c->window_on_device = !value; return ORBX_OK; }
    if (which == ORBM_VAR_BEST2_RESIDENT && value >= 0 && value <= 2) { c->best2_resident = value; return ORBX_OK; }
    if (which == ORBM_VAR_INIT_LANES && (value == 0 || value == 1 || value == 4 || value == 16 || value == 64)) { c->init_lanes = value; return ORBX_OK; }
    return orbx_set_error(ORBX_E_ARG, "unknown matcher variant switch or value out of range");
}

extern "C" void orbm_destroy(orbm_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->null_pending) (void)hipStreamSynchronize((hipStream_t)0);
    DevBuf *bufs[] = {&c->a, &c->b, &c->out, &c->q_idx, &c->c_begin, &c->c_len, &c->out_begin, &c->c_idx,
                      &c->row_ok, &c->col_ok, &c->bidx, &c->bbest, &c->bsecond, &c->w_in, &c->w_out, &c->w_grid};
    for (DevBuf *d : bufs) d->release();
    c->h_in.release(); c->h_out.release();
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbm_hamming_matrix_device(orbm_t *c, const uint8_t *d_a, int na, const uint8_t *d_b, int nb,
                                          uint16_t *d_out, void *stream)
{
    if (!c || !d_a || !d_b || !d_out || na < 0 || nb < 0) return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (na == 0 || nb == 0) return ORBX_OK;
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    dim3 grid((nb + 63) / 64, (na + 63) / 64);
    hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(256), 0, s, d_a, na, d_b, nb, d_out);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbm_hamming_matrix(orbm_t *c, const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *out)
{
    if (!c || !a || !b || !out || na < 0 || nb < 0) return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (na == 0 || nb == 0) return ORBX_OK;
    M_TRY(host_call_begin(c));
    M_TRY(c->a.need((size_t)na * 32));
    M_TRY(c->b.need((size_t)nb * 32));
    M_TRY(c->out.need((size_t)na * nb * 2));
    M_TRY(hipMemcpyAsync(c->a.p, a, (size_t)na * 32, hipMemcpyHostToDevice, c->stream));
    M_TRY(hipMemcpyAsync(c->b.p, b, (size_t)nb * 32, hipMemcpyHostToDevice, c->stream));
    int rc = orbm_hamming_matrix_device(c, (const uint8_t *)c->a.p, na, (const uint8_t *)c->b.p, nb, (uint16_t *)c->out.p,
                                        c->stream);
    if (rc) return rc;
    M_TRY(hipMemcpyAsync(out, c->out.p, (size_t)na * nb * 2, hipMemcpyDeviceToHost, c->stream));
    M_TRY(hipStreamSynchronize(c->stream));
    return ORBX_OK;
}

extern "C" int orbm_best2_device(orbm_t *c, int n_pairs, const uint8_t *d_a, size_t a_stride, const int32_t *d_na,
                                 int na_max, const uint8_t *d_b, size_t b_stride, const int32_t *d_nb, int nb_max,
                                 const uint8_t *d_row_ok, const uint8_t *d_col_ok, int32_t *d_best_idx, uint16_t *d_best,
                                 uint16_t *d_second, void *stream)
{
    if (!c || !d_a || !d_b || !d_best_idx || !d_best || !d_second || n_pairs < 1 || na_max < 0 || nb_max < 0)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (nb_max >= (1 << 23)) return orbx_set_error(ORBX_E_UNSUPPORTED, "more than 2^23 candidates per problem");
    if (na_max == 0) return ORBX_OK;
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    // the matrix-pipe kernel takes every problem without a candidate mask and at most BM_MAX_CAND (8160) candidates --
    // 16 * tile + register must stay below the 4096 free low bits of its keys; anything else runs k_best2;
    // ORBM_VAR_BEST2 = 2 keeps everything on the VALU kernel (the parity twin), 1 takes the i8 matrix kernel
    const int variant = c->best2_variant == 0 ? 2 : c->best2_variant == 1 ? 1 : 0; // 2 = fp4, 1 = i8, 0 = valu
    const bool use_mfma = variant != 0;
    if (variant == 2 && !d_col_ok && nb_max <= BM_MAX_CAND) {
        const int blocks_x = (na_max + BF_WAVES * 64 - 1) / (BF_WAVES * 64);
        const long long n_blocks = (long long)blocks_x * n_pairs;
        if (n_blocks > 0x7fffffff) return orbx_set_error(ORBX_E_UNSUPPORTED, "too many problems for one launch");
        const long long resident = (long long)c->best2_resident * c->n_cus;
        const int grid = (int)(c->best2_resident > 0 ? std::min(n_blocks, resident) : n_blocks);
        if (c->best2_resident > 0)
            hipLaunchKernelGGL(k_best2_fp4<true>, dim3(grid), dim3(BF_WAVES * 64), 0, s, d_a, a_stride, d_na, na_max, d_b, b_stride, d_nb,
                               nb_max, d_row_ok, d_best_idx, d_best, d_second, blocks_x, (int)n_blocks);
        else
            hipLaunchKernelGGL(k_best2_fp4<false>, dim3(grid), dim3(BF_WAVES * 64), 0, s, d_a, a_stride, d_na, na_max, d_b, b_stride, d_nb,
                               nb_max, d_row_ok, d_best_idx, d_best, d_second, blocks_x, (int)n_blocks);
        M_TRY(hipGetLastError());
        return ORBX_OK;
    }
    if (use_mfma && !d_col_ok && nb_max <= BM_MAX_CAND) {
        dim3 grid((na_max + BM_WAVES * 32 - 1) / (BM_WAVES * 32), n_pairs);
        hipLaunchKernelGGL(k_best2_mfma, grid, dim3(BM_WAVES * 64), 0, s, d_a, a_stride, d_na, na_max, d_b, b_stride, d_nb,
                           nb_max, d_row_ok, d_best_idx, d_best, d_second);
        M_TRY(hipGetLastError());
        return ORBX_OK;
    }
    dim3 grid((na_max + 4 * B2_ROWS - 1) / (4 * B2_ROWS), n_pairs);
    hipLaunchKernelGGL(k_best2, grid, dim3(256), 0, s, d_a, a_stride, d_na, na_max, d_b, b_stride, d_nb, nb_max,
                       d_row_ok, d_col_ok, d_best_idx, d_best, d_second);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbm_best2(orbm_t *c, const uint8_t *a, int na, const uint8_t *b, int nb, const uint8_t *row_ok,
                          const uint8_t *col_ok, int32_t *best_idx, uint16_t *best, uint16_t *second)
{
    if (!c || !a || !best_idx || !best || !second || na < 0 || nb < 0 || (nb > 0 && !b))
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (na == 0) return ORBX_OK;
    M_TRY(host_call_begin(c));
    hipStream_t s = c->stream;
    M_TRY(c->a.need((size_t)na * 32));
    M_TRY(c->b.need((size_t)std::max(nb, 1) * 32));
    M_TRY(c->bidx.need((size_t)na * 4));
    M_TRY(c->bbest.need((size_t)na * 2));
    M_TRY(c->bsecond.need((size_t)na * 2));
    M_TRY(hipMemcpyAsync(c->a.p, a, (size_t)na * 32, hipMemcpyHostToDevice, s));
    if (nb) M_TRY(hipMemcpyAsync(c->b.p, b, (size_t)nb * 32, hipMemcpyHostToDevice, s));
    const uint8_t *d_row = nullptr, *d_col = nullptr;
    if (row_ok) { M_TRY(c->row_ok.need(na)); M_TRY(hipMemcpyAsync(c->row_ok.p, row_ok, na, hipMemcpyHostToDevice, s)); d_row = (const uint8_t *)c->row_ok.p; }
    if (col_ok && nb) { M_TRY(c->col_ok.need(nb)); M_TRY(hipMemcpyAsync(c->col_ok.p, col_ok, nb, hipMemcpyHostToDevice, s)); d_col = (const uint8_t *)c->col_ok.p; }
    int rc = orbm_best2_device(c, 1, (const uint8_t *)c->a.p, na, nullptr, na, (const uint8_t *)c->b.p, std::max(nb, 1),
                               nullptr, nb, d_row, d_col, (int32_t *)c->bidx.p, (uint16_t *)c->bbest.p,
                               (uint16_t *)c->bsecond.p, c->stream);
    if (rc) return rc;
    M_TRY(hipMemcpyAsync(best_idx, c->bidx.p, (size_t)na * 4, hipMemcpyDeviceToHost, s));
    M_TRY(hipMemcpyAsync(best, c->bbest.p, (size_t)na * 2, hipMemcpyDeviceToHost, s));
    M_TRY(hipMemcpyAsync(second, c->bsecond.p, (size_t)na * 2, hipMemcpyDeviceToHost, s));
    M_TRY(hipStreamSynchronize(s));
    return ORBX_OK;
}

// distances for per-query candidate lists; the lists may alias each other (queries of one BoW
// node share the node's candidate list), outputs are disjoint
struct PhaseTrace { // ORBM_TRACE=1: host time stamps of the phases of a window search on stderr
    bool on; std::chrono::steady_clock::time_point t0;
    PhaseTrace() : on(getenv("ORBM_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[orbm] %-24s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - t0).count());
        t0 = t;
    }
};

static int hamming_lists(orbm_ctx *c, const uint8_t *a, int na, const uint8_t *b, int nb,
                         const std::vector<int32_t> &q_idx, const std::vector<int32_t> &c_begin,
                         const std::vector<int32_t> &c_len, const std::vector<int32_t> &out_begin,
                         const int32_t *c_idx, size_t n_cidx, size_t n_out, std::vector<uint16_t> &out)
{
    out.resize(n_out);
    const int nq = (int)q_idx.size();
    if (nq == 0 || n_out == 0) return ORBX_OK;
    M_TRY(host_call_begin(c));
    hipStream_t s = c->stream;
    // one page-locked block up, one down (as topk_lists below): [a][b][q_idx][c_begin][c_len][out_begin][c_idx]
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_a = 0, o_b = al(o_a + (size_t)na * 32), o_q = al(o_b + (size_t)nb * 32), o_cb = al(o_q + (size_t)nq * 4),
                 o_cl = al(o_cb + (size_t)nq * 4), o_ob = al(o_cl + (size_t)nq * 4), o_ci = al(o_ob + (size_t)nq * 4),
                 in_bytes = al(o_ci + n_cidx * 4);
    M_TRY(c->h_in.need(in_bytes));
    M_TRY(c->w_in.need(in_bytes));
    M_TRY(c->h_out.need(n_out * 2));
    M_TRY(c->out.need(n_out * 2));
    uint8_t *hp = (uint8_t *)c->h_in.p;
    memcpy(hp + o_a, a, (size_t)na * 32);
    memcpy(hp + o_b, b, (size_t)nb * 32);
    memcpy(hp + o_q, q_idx.data(), (size_t)nq * 4);
    memcpy(hp + o_cb, c_begin.data(), (size_t)nq * 4);
    memcpy(hp + o_cl, c_len.data(), (size_t)nq * 4);
    memcpy(hp + o_ob, out_begin.data(), (size_t)nq * 4);
    memcpy(hp + o_ci, c_idx, n_cidx * 4);
    M_TRY(hipMemcpyAsync(c->w_in.p, hp, in_bytes, hipMemcpyHostToDevice, s));
    const uint8_t *dp = (const uint8_t *)c->w_in.p;
    hipLaunchKernelGGL(k_hamming_lists, dim3((nq + 3) / 4), dim3(256), 0, s, dp + o_a, dp + o_b, (const int32_t *)(dp + o_q),
                       (const int32_t *)(dp + o_cb), (const int32_t *)(dp + o_cl), (const int32_t *)(dp + o_ob), nq, (const int32_t *)(dp + o_ci),
                       (uint16_t *)c->out.p);
    M_TRY(hipGetLastError());
    M_TRY(hipMemcpyAsync(c->h_out.p, c->out.p, n_out * 2, hipMemcpyDeviceToHost, s));
    M_TRY(hipStreamSynchronize(s));
    memcpy(out.data(), c->h_out.p, n_out * 2);
    return ORBX_OK;
}

// top-K keys per query (see k_topk_lists); `cand_free` has nb bytes (1 = candidate may be used), may be NULL
static int topk_lists(orbm_ctx *c, const uint8_t *a, int na, const uint8_t *b, int nb, const std::vector<int32_t> &q_idx,
                      const std::vector<int32_t> &c_begin, const std::vector<int32_t> &c_len, const int32_t *c_idx,
                      size_t n_cidx, const uint8_t *cand_free, std::vector<uint32_t> &out, int *k_out)
{
    const int nq = (int)q_idx.size();
    int max_len = 0;
    for (int len : c_len) max_len = std::max(max_len, len);
    const int TOPK = max_len > 256 ? 16 : 8;
    *k_out = TOPK;
    out.assign((size_t)nq * TOPK, 0xFFFFFFFFu);
    if (nq == 0 || n_cidx == 0) return ORBX_OK;
    M_TRY(host_call_begin(c));
    hipStream_t s = c->stream;
    // One page-locked block up, one down (round 6; seven copies from pageable memory and one to it before: every one of those is
    // staged by the runtime and holds the calling thread for 10-20 us).  Layout, 16-byte aligned:
    // [a][b][q_idx][c_begin][c_len][c_idx][cand_free]
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_a = 0, o_b = al(o_a + (size_t)na * 32), o_q = al(o_b + (size_t)nb * 32), o_cb = al(o_q + (size_t)nq * 4),
                 o_cl = al(o_cb + (size_t)nq * 4), o_ci = al(o_cl + (size_t)nq * 4), o_fr = al(o_ci + n_cidx * 4),
                 in_bytes = al(o_fr + (cand_free ? (size_t)nb : 0));
    const size_t out_bytes = (size_t)nq * TOPK * 4;
    M_TRY(c->h_in.need(in_bytes));
    M_TRY(c->w_in.need(in_bytes));
    M_TRY(c->h_out.need(out_bytes));
    M_TRY(c->out.need(out_bytes));
    uint8_t *hp = (uint8_t *)c->h_in.p;
    memcpy(hp + o_a, a, (size_t)na * 32);
    memcpy(hp + o_b, b, (size_t)nb * 32);
    memcpy(hp + o_q, q_idx.data(), (size_t)nq * 4);
    memcpy(hp + o_cb, c_begin.data(), (size_t)nq * 4);
    memcpy(hp + o_cl, c_len.data(), (size_t)nq * 4);
    memcpy(hp + o_ci, c_idx, n_cidx * 4);
    if (cand_free) memcpy(hp + o_fr, cand_free, (size_t)nb);
    M_TRY(hipMemcpyAsync(c->w_in.p, hp, in_bytes, hipMemcpyHostToDevice, s));
    const uint8_t *dp = (const uint8_t *)c->w_in.p;
    const uint8_t *d_free = cand_free ? dp + o_fr : nullptr;
    if (TOPK == 16)
        hipLaunchKernelGGL(k_topk_lists<16>, dim3((nq + 3) / 4), dim3(256), 0, s, dp + o_a, dp + o_b, (const int32_t *)(dp + o_q),
                           (const int32_t *)(dp + o_cb), (const int32_t *)(dp + o_cl), nq, (const int32_t *)(dp + o_ci), d_free,
                           (uint32_t *)c->out.p);
    else
        hipLaunchKernelGGL(k_topk_lists<8>, dim3((nq + 3) / 4), dim3(256), 0, s, dp + o_a, dp + o_b, (const int32_t *)(dp + o_q),
                           (const int32_t *)(dp + o_cb), (const int32_t *)(dp + o_cl), nq, (const int32_t *)(dp + o_ci), d_free,
                           (uint32_t *)c->out.p);
    M_TRY(hipGetLastError());
    M_TRY(hipMemcpyAsync(c->h_out.p, c->out.p, out_bytes, hipMemcpyDeviceToHost, s));
    M_TRY(hipStreamSynchronize(s));
    memcpy(out.data(), c->h_out.p, out_bytes);
    return ORBX_OK;
}

// 256-bit Hamming distance on the host: only for the rare rows whose device top-K was used up by earlier matches
static inline int ham256_host(const uint8_t *x, const uint8_t *y)
{
    int d = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned long long u, v;
        memcpy(&u, x + 8 * i, 8);
        memcpy(&v, y + 8 * i, 8);
        d += __builtin_popcountll(u ^ v);
    }
    return d;
}

extern "C" int orbm_hamming_csr(orbm_t *c, const uint8_t *a, int na, const uint8_t *b, int nb, const int32_t *q_idx,
                                const int32_t *off, int n_queries, const int32_t *c_idx, uint16_t *out)
{
    if (!c || !a || !b || !q_idx || !off || !out || n_queries < 0) return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (n_queries == 0) return ORBX_OK;
    std::vector<int32_t> q(q_idx, q_idx + n_queries), cb(off, off + n_queries), cl(n_queries);
    for (int i = 0; i < n_queries; ++i) cl[i] = off[i + 1] - off[i];
    std::vector<uint16_t> tmp;
    const size_t n_out = (size_t)off[n_queries];
    if (n_out && !c_idx) return orbx_set_error(ORBX_E_ARG, "null candidate list");
    int rc = hamming_lists(c, a, na, b, nb, q, cb, cl, cb, c_idx, n_out, n_out, tmp);
    if (rc) return rc;
    if (n_out) memcpy(out, tmp.data(), n_out * 2);
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side greedy passes over device-computed distances
// ---------------------------------------------------------------------------------------------
extern "C" int orbm_distinctive_descriptors_device(orbm_t *c, const uint8_t *d_desc, const int32_t *d_off, int n_groups,
                                                   int32_t *d_best_idx, void *stream)
{
    if (!c || !d_desc || !d_off || !d_best_idx) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n_groups < 0) return orbx_set_error(ORBX_E_ARG, "negative group count");
    if (n_groups == 0) return ORBX_OK;
    M_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    hipLaunchKernelGGL(k_medoid, dim3(n_groups), dim3(64), 0, s, d_desc, d_off, n_groups, d_best_idx);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbm_distinctive_descriptors(orbm_t *c, const uint8_t *desc, const int32_t *off, int n_groups,
                                            int32_t *best_idx)
{
    if (!c || !off || !best_idx) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n_groups < 0) return orbx_set_error(ORBX_E_ARG, "negative group count");
    if (n_groups == 0) return ORBX_OK;
    const int total = off[n_groups];
    for (int g = 0; g < n_groups; ++g) {
        if (off[g + 1] < off[g] || off[g] < 0) return orbx_set_error(ORBX_E_ARG, "group offsets must be non-decreasing");
        if (off[g + 1] - off[g] > MEDOID_MAX)
            return orbx_set_error(ORBX_E_UNSUPPORTED, "more than 1024 observations of one map point");
    }
    if (total > 0 && !desc) return orbx_set_error(ORBX_E_ARG, "null descriptors");
    M_TRY(host_call_begin(c));
    hipStream_t s = c->stream;
    M_TRY(c->a.need((size_t)std::max(total, 1) * 32));
    M_TRY(c->c_begin.need((size_t)(n_groups + 1) * 4));
    M_TRY(c->bidx.need((size_t)n_groups * 4));
    if (total > 0) M_TRY(hipMemcpyAsync(c->a.p, desc, (size_t)total * 32, hipMemcpyHostToDevice, s));
    M_TRY(hipMemcpyAsync(c->c_begin.p, off, (size_t)(n_groups + 1) * 4, hipMemcpyHostToDevice, s));
    int rc = orbm_distinctive_descriptors_device(c, (const uint8_t *)c->a.p, (const int32_t *)c->c_begin.p, n_groups,
                                                 (int32_t *)c->bidx.p, s);
    if (rc) return rc;
    M_TRY(hipMemcpyAsync(best_idx, c->bidx.p, (size_t)n_groups * 4, hipMemcpyDeviceToHost, s));
    M_TRY(hipStreamSynchronize(s));
    return ORBX_OK;
}

extern "C" void orbm_three_maxima(const int32_t *h, int n_bins, int *ind1, int *ind2, int *ind3)
{
    // ORBMatcher.cpp:594-622 (callers initialise the three indices to -1)
    int max1 = 0, max2 = -1, max3 = -2;
    for (int i = 0; i < n_bins; ++i) {
        const int n = h[i];
        if (n > max1) { max3 = max2; max2 = max1; max1 = n; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (n > max2) { max3 = max2; max2 = n; *ind3 = *ind2; *ind2 = i; }
        else if (n > max3) { max3 = n; *ind3 = i; }
    }
    if (max2 < max1 / 10) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < max1 / 10) *ind3 = -1;
}

namespace {
struct RotHist {
    std::vector<int> bins[ORBM_HISTO_LENGTH];
    void add(float a1, float a2, int v)
    {
        // `factor = 1.f / HISTO_LENGTH` as in the reference (:128), so only bins 0..12 fill up
        const float factor = 1.f / ORBM_HISTO_LENGTH;
        float rot = a1 - a2;
        if (rot < 0) rot += 360.f;
        int bin = orb_round_f(rot * factor);
        if (bin == ORBM_HISTO_LENGTH) bin = 0;
        bins[bin].push_back(v);
    }
    void keep3(int *i1, int *i2, int *i3) const
    {
        int32_t sizes[ORBM_HISTO_LENGTH];
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) sizes[i] = (int32_t)bins[i].size();
        *i1 = *i2 = *i3 = -1;
        orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, i1, i2, i3);
    }
};

// merge-walk of two FeatureVectors (:136-185): pairs of positions with equal node id
struct NodePair { int i1, i2; };
void shared_nodes(const orbm_fv *f1, const orbm_fv *f2, std::vector<NodePair> &out)
{
    int i1 = 0, i2 = 0;
    while (i1 < f1->n_nodes && i2 < f2->n_nodes) {
        const uint32_t a = f1->node_ids[i1], b = f2->node_ids[i2];
        if (a == b) { out.push_back({i1, i2}); ++i1; ++i2; }
        else if (a < b) i1 = (int)(std::lower_bound(f1->node_ids + i1, f1->node_ids + f1->n_nodes, b) - f1->node_ids);
        else i2 = (int)(std::lower_bound(f2->node_ids + i2, f2->node_ids + f2->n_nodes, a) - f2->node_ids);
    }
}

// one query per (shared node, usable feature of side 1); candidates = the node's features on side 2
struct NodeQueries {
    std::vector<int32_t> q_idx, c_begin, c_len, out_begin;
    size_t n_out = 0;
};
void build_node_queries(const orbm_fv *f1, const orbm_fv *f2, const std::vector<NodePair> &nodes,
                        const uint8_t *skip1_if_zero, const uint8_t *skip1_if_nonzero, NodeQueries &q)
{
    for (const NodePair &np : nodes) {
        const int cb = f2->offsets[np.i2], cl = f2->offsets[np.i2 + 1] - cb;
        for (int a = f1->offsets[np.i1]; a < f1->offsets[np.i1 + 1]; ++a) {
            const int idx1 = (int)f1->indices[a];
            if (skip1_if_zero && !skip1_if_zero[idx1]) continue;
            if (skip1_if_nonzero && skip1_if_nonzero[idx1]) continue;
            q.q_idx.push_back(idx1); q.c_begin.push_back(cb); q.c_len.push_back(cl);
            q.out_begin.push_back((int32_t)q.n_out);
            q.n_out += (size_t)cl;
        }
    }
}
} // namespace

extern "C" int orbm_search_by_bow(orbm_t *c, float nn_ratio, int check_orientation, const uint8_t *desc1,
                                  const float *angle1, const uint8_t *kf_mp_ok, int n1, const orbm_fv *fv1,
                                  const uint8_t *desc2, const float *angle2, int32_t *frame_mp, int n2,
                                  const orbm_fv *fv2, int *n_matches)
{
    if (!c || !desc1 || !desc2 || !kf_mp_ok || !frame_mp || !fv1 || !fv2 || !n_matches || !angle1 || !angle2)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    *n_matches = 0;
    if (n1 <= 0 || n2 <= 0) return ORBX_OK;
    PhaseTrace tr;
    std::vector<NodePair> nodes;
    shared_nodes(fv1, fv2, nodes);
    NodeQueries q;
    build_node_queries(fv1, fv2, nodes, kf_mp_ok, nullptr, q);
    tr.mark("bow: node queries");
    // The device returns, per key-frame feature, the TOPK closest candidates of its node among the frame features
    // that are free when the call starts.  The reference's sequential loop only ever needs the two closest
    // candidates that are STILL free (:150-161); they are the first two unmatched entries of that sorted list.
    // Only if earlier matches of this call used up the list is the row recomputed on the host.
    bool long_list = false;
    for (int len : q.c_len) long_list = long_list || len > 65535;
    std::vector<uint8_t> free0((size_t)n2);
    for (int j = 0; j < n2; ++j) free0[j] = frame_mp[j] == -1;
    std::vector<uint32_t> topk;
    std::vector<uint16_t> dist;
    int rc, TOPK = 8;
    if (!long_list)
        rc = topk_lists(c, desc1, n1, desc2, n2, q.q_idx, q.c_begin, q.c_len, reinterpret_cast<const int32_t *>(fv2->indices),
                        (size_t)fv2->offsets[fv2->n_nodes], free0.data(), topk, &TOPK);
    else
        rc = hamming_lists(c, desc1, n1, desc2, n2, q.q_idx, q.c_begin, q.c_len, q.out_begin,
                           reinterpret_cast<const int32_t *>(fv2->indices), (size_t)fv2->offsets[fv2->n_nodes], q.n_out, dist);
    if (rc) return rc;
    tr.mark("bow: device top-k");
    RotHist rh;
    int num = 0, n_rescans = 0;
    for (size_t k = 0; k < q.q_idx.size(); ++k) {
        const int idx1 = q.q_idx[k];
        const uint32_t *cand = fv2->indices + q.c_begin[k];
        int bestDist = 256, secondDist = 256, bestIdx2 = -1;
        bool resolved = false;
        if (!long_list) {
            int found = 0;
            bool exhausted = false; // saw the end-of-list sentinel: no further free candidate exists
            int last_dd = 0; // distance of the last list entry looked at: no free candidate outside the list is closer
            for (int r = 0; r < TOPK && found < 2; ++r) {
                const uint32_t key = topk[k * TOPK + r];
                if (key == 0xFFFFFFFFu) { exhausted = true; break; }
                const int idx2 = (int)cand[key & 0xFFFF], dd = (int)(key >> 16);
                last_dd = dd;
                if (frame_mp[idx2] != -1) continue; // taken by an earlier match of this call (:150)
                if (found == 0) { if (dd < 256) { bestDist = dd; bestIdx2 = idx2; } }
                else secondDist = std::min(dd, 256);
                ++found;
            }
            // decided if both were seen, if the list ended, if the best already fails the threshold (:164) -- or if the best
            // passes the ratio test even against the smallest value the unseen second best can have (the list is sorted, so
            // that is the distance of its last entry; the second distance is used for nothing but this test)
            resolved = found == 2 || exhausted || (found == 1 && bestDist > ORBM_TH_LOW);
            if (!resolved && found == 1 && (float)bestDist < nn_ratio * (float)std::min(last_dd, 256)) {
                secondDist = std::min(last_dd, 256);
                resolved = true;
            }
        }
        if (!resolved) {
            ++n_rescans;
            bestDist = 256; secondDist = 256; bestIdx2 = -1;
            for (int t = 0; t < q.c_len[k]; ++t) {
                const int idx2 = (int)cand[t];
                if (frame_mp[idx2] != -1) continue; // :150 -- includes features matched earlier in this call
                const int dd = long_list ? (int)dist[q.out_begin[k] + t]
                                         : ham256_host(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                if (dd < bestDist) { secondDist = bestDist; bestDist = dd; bestIdx2 = idx2; }
                else if (dd < secondDist) secondDist = dd;
            }
        }
        if (bestDist <= ORBM_TH_LOW && (float)bestDist < nn_ratio * (float)secondDist) { // :164
            frame_mp[bestIdx2] = idx1;
            ++num;
            if (check_orientation) rh.add(angle1[idx1], angle2[bestIdx2], bestIdx2);
        }
    }
    if (tr.on) fprintf(stderr, "[orbm] bow: %zu queries, TOPK %d, %d rows rescanned on the host\n", q.q_idx.size(), TOPK, n_rescans);
    tr.mark("bow: greedy replay");
    if (check_orientation) {
        int i1, i2, i3;
        rh.keep3(&i1, &i2, &i3);
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx2 : rh.bins[i]) { frame_mp[idx2] = -1; --num; }
        }
    }
    *n_matches = num;
    return ORBX_OK;
}

extern "C" int orbm_search_for_triangulation(orbm_t *c, int check_orientation, const uint8_t *desc1,
                                             const float *angle1, const uint8_t *has_mp1, int n1, const orbm_fv *fv1,
                                             const uint8_t *desc2, const float *angle2, const uint8_t *has_mp2, int n2,
                                             const orbm_fv *fv2, int32_t *matches12, int *n_matches)
{
    if (!c || !desc1 || !desc2 || !has_mp1 || !has_mp2 || !fv1 || !fv2 || !matches12 || !n_matches || !angle1 || !angle2)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    *n_matches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (n1 <= 0 || n2 <= 0) return ORBX_OK;
    std::vector<NodePair> nodes;
    shared_nodes(fv1, fv2, nodes);
    NodeQueries q;
    build_node_queries(fv1, fv2, nodes, nullptr, has_mp1, q); // :452
    // Same scheme as orbm_search_by_bow: the device returns the TOPK closest initially-eligible candidates per feature,
    // the sequential pass takes the first one not matched earlier in this call (:466-:476 keeps only the best).
    bool long_list = false;
    for (int len : q.c_len) long_list = long_list || len > 65535;
    std::vector<uint8_t> free0((size_t)n2);
    for (int j = 0; j < n2; ++j) free0[j] = !has_mp2[j];
    std::vector<uint32_t> topk;
    std::vector<uint16_t> dist;
    int rc, TOPK = 8;
    if (!long_list)
        rc = topk_lists(c, desc1, n1, desc2, n2, q.q_idx, q.c_begin, q.c_len, reinterpret_cast<const int32_t *>(fv2->indices),
                        (size_t)fv2->offsets[fv2->n_nodes], free0.data(), topk, &TOPK);
    else
        rc = hamming_lists(c, desc1, n1, desc2, n2, q.q_idx, q.c_begin, q.c_len, q.out_begin,
                           reinterpret_cast<const int32_t *>(fv2->indices), (size_t)fv2->offsets[fv2->n_nodes], q.n_out, dist);
    if (rc) return rc;
    std::vector<uint8_t> matched2(n2, 0);
    RotHist rh;
    int num = 0;
    for (size_t k = 0; k < q.q_idx.size(); ++k) {
        const int idx1 = q.q_idx[k];
        const uint32_t *cand = fv2->indices + q.c_begin[k];
        int bestDist = ORBM_TH_LOW, bestIdx2 = -1;
        bool resolved = false;
        if (!long_list) {
            int last_dd = 0;
            for (int r = 0; r < TOPK; ++r) {
                const uint32_t key = topk[k * TOPK + r];
                if (key == 0xFFFFFFFFu) { resolved = true; break; }
                const int idx2 = (int)cand[key & 0xFFFF], dd = (int)(key >> 16);
                last_dd = dd;
                if (matched2[idx2]) continue; // :466 -- matched earlier in this call
                if (dd < bestDist) { bestIdx2 = idx2; bestDist = dd; }
                resolved = true;
                break;
            }
            // the whole list is taken, but it is sorted: nothing outside it is closer than its last entry
            if (!resolved && last_dd >= ORBM_TH_LOW) resolved = true;
        }
        if (!resolved) {
            for (int t = 0; t < q.c_len[k]; ++t) {
                const int idx2 = (int)cand[t];
                if (matched2[idx2] || has_mp2[idx2]) continue; // :466
                const int dd = long_list ? (int)dist[q.out_begin[k] + t]
                                         : ham256_host(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                if (dd < bestDist) { bestIdx2 = idx2; bestDist = dd; }
            }
        }
        if (bestIdx2 > 0) { // :484 -- feature 0 of key frame 2 is never accepted
            matches12[idx1] = bestIdx2;
            matched2[bestIdx2] = 1;
            ++num;
            if (check_orientation) rh.add(angle1[idx1], angle2[bestIdx2], idx1);
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        rh.keep3(&i1, &i2, &i3);
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx1 : rh.bins[i]) { matches12[idx1] = -1; --num; }
        }
    }
    *n_matches = num;
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// window searches: Frame grid + getFeaturesInArea and the distances on the device (host twin kept), greedy pass on the host
// ---------------------------------------------------------------------------------------------
namespace {
// Frame::grid (Frame.cpp:33-51): 40-px cells, grid[x][y] vectors filled in key-point order
struct FrameGrid {
    static const int G = 40;
    int cols, rows;
    std::vector<std::vector<int>> cell;
    FrameGrid(const orbx_kp *kps, int n, int img_w, int img_h)
    {
        cols = img_w % G == 0 ? img_w / G : img_w / G + 1;
        rows = img_h % G == 0 ? img_h / G : img_h / G + 1;
        cell.resize((size_t)cols * rows);
        for (int i = 0; i < n; ++i) {
            const int x = orb_floor_f(kps[i].x), y = orb_floor_f(kps[i].y); // Frame::PosInGrid (:90-95)
            if (x < 0 || x >= img_w || y < 0 || y >= img_h) continue;
            cell[(size_t)(x / G) * rows + y / G].push_back(i);
        }
    }
    // Frame::getFeaturesInArea (Frame.cpp:97-127), appended to `out`
    // strict = KeyFrame::getFeaturesInArea (KeyFrame.cpp:181-211: `< r`), otherwise Frame's (`<= r`)
    void area(const orbx_kp *kps, float x, float y, float r, int minLevel, int maxLevel, std::vector<int32_t> &out,
              bool strict = false) const
    {
        const int minCX = std::max(0, orb_floor_f(x - r) / G), maxCX = std::min(cols - 1, orb_floor_f(x + r) / G);
        if (minCX > maxCX) return;
        const int minCY = std::max(0, orb_floor_f(y - r) / G), maxCY = std::min(rows - 1, orb_floor_f(y + r) / G);
        if (minCY > maxCY) return;
        const bool check = minLevel > 0 || maxLevel >= 0;
        for (int cx = minCX; cx <= maxCX; ++cx)
            for (int cy = minCY; cy <= maxCY; ++cy)
                for (int j : cell[(size_t)cx * rows + cy]) {
                    if (check) {
                        if (kps[j].octave < minLevel) continue;
                        if (maxLevel >= 0 && kps[j].octave > maxLevel) continue;
                    }
                    const float ax = fabsf(kps[j].x - x), ay = fabsf(kps[j].y - y);
                    if (strict ? (ax < r && ay < r) : (ax <= r && ay <= r)) out.push_back(j);
                }
    }
};
struct WindowQueries {
    std::vector<int32_t> q_idx, c_begin, c_len;
    const uint32_t *e = nullptr;      // packed entries: distance << 22 | candidate index
    std::vector<uint32_t> own;        // the entries when they were made on the host
    std::vector<int32_t> c_idx;       // host path scratch
    int idx(size_t k, int t) const { return (int)(e[c_begin[k] + t] & 0x3FFFFFu); }
    int dist(size_t k, int t) const { return (int)(e[c_begin[k] + t] >> 22); }
};
} // namespace

// The candidate lists of a window search, either way: q.q_idx = queries with a non-empty window, their lists
// q.e[q.c_begin[k] .. + q.c_len[k]) (distance << 22 | index) in the reference's getFeaturesInArea order.  On the device
// path q.e is the pinned copy-out buffer of the context, read in place (valid until the context's next call).
//   device path (default): ONE pinned staging copy in, grid build + one wave per query on the device, ONE copy out;
//   host path (ORBM_VAR_WINDOW = 1, or a window longer than `cap`): FrameGrid::area on the host + k_hamming_lists.

static int window_candidates(orbm_ctx *c, bool strict, const float *sigma2, int n_sigma, int cap, const uint8_t *q_desc,
                             const float *q_xy, const float *q_radius, const int32_t *q_min, const int32_t *q_max,
                             const uint8_t *q_ok, int nq, const orbx_kp *kps2, const uint8_t *desc2, int n2, int img_w,
                             int img_h, WindowQueries &q)
{
    q.q_idx.clear(); q.c_begin.clear(); q.c_len.clear(); q.c_idx.clear(); q.own.clear();
    q.e = nullptr;
    if (nq <= 0 || n2 <= 0) return ORBX_OK;
    const int G = ORBM_GRID;
    const int cols = img_w % G == 0 ? img_w / G : img_w / G + 1, rows = img_h % G == 0 ? img_h / G : img_h / G + 1;
    const int nc = cols * rows;
    bool on_device = c->window_on_device && nc > 0 && (size_t)(2 * nc + 1) * 4 <= 60000 && n2 < (1 << 22);
    if (on_device) {
        M_TRY(host_call_begin(c));
        hipStream_t s = c->stream;
        auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
        size_t o = 0;
        const size_t o_kps = o; o = al(o + (size_t)n2 * sizeof(orbx_kp));
        const size_t o_desc = o; o = al(o + (size_t)n2 * 32);
        const size_t o_qd = o; o = al(o + (size_t)nq * 32);
        const size_t o_xy = o; o = al(o + (size_t)nq * 8);
        const size_t o_r = o; o = al(o + (size_t)nq * 4);
        const size_t o_min = o; o = al(o + (size_t)nq * 4);
        const size_t o_max = o; o = al(o + (size_t)nq * 4);
        const size_t o_ok = o; o = al(o + (size_t)nq);
        const size_t o_s2 = o; o = al(o + (size_t)std::max(n_sigma, 1) * 4);
        PhaseTrace tr;
        M_TRY(c->h_in.need(o));
        M_TRY(c->w_in.need(o));
        uint8_t *hp = (uint8_t *)c->h_in.p;
        memcpy(hp + o_kps, kps2, (size_t)n2 * sizeof(orbx_kp));
        memcpy(hp + o_desc, desc2, (size_t)n2 * 32);
        memcpy(hp + o_qd, q_desc, (size_t)nq * 32);
        memcpy(hp + o_xy, q_xy, (size_t)nq * 8);
        memcpy(hp + o_r, q_radius, (size_t)nq * 4);
        memcpy(hp + o_min, q_min, (size_t)nq * 4);
        memcpy(hp + o_max, q_max, (size_t)nq * 4);
        memcpy(hp + o_ok, q_ok, (size_t)nq);
        if (sigma2) memcpy(hp + o_s2, sigma2, (size_t)n_sigma * 4);
        tr.mark("stage inputs");
        M_TRY(hipMemcpyAsync(c->w_in.p, hp, o, hipMemcpyHostToDevice, s));
        const uint8_t *dp = (const uint8_t *)c->w_in.p;
        const size_t g_bytes = ((size_t)nc + 1 + 3 * (size_t)n2) * 4;
        M_TRY(c->w_grid.need(g_bytes));
        int32_t *cell_start = (int32_t *)c->w_grid.p, *cell_items = cell_start + nc + 1, *cell_of = cell_items + n2, *tmp = cell_of + n2;
        // device output: [total, pad x3][counts nq][offs nq][pool]; the lists are packed, `cap` entries per query on average
        const size_t pool_cap = (size_t)nq * cap, head = 16 + (size_t)nq * 8;
        // what one copy brings back (the rest only if needed): the previous call's total is the best guess for this one
        const size_t first = std::min(pool_cap, std::max((size_t)nq * 12 + 256, c->window_last_total + c->window_last_total / 8 + 64));
        M_TRY(c->w_out.need(head + pool_cap * 4));
        M_TRY(c->h_out.need(head + pool_cap * 4));
        int32_t *d_total = (int32_t *)c->w_out.p, *d_counts = d_total + 4, *d_offs = d_counts + nq;
        uint32_t *d_pool = (uint32_t *)(d_offs + nq);
        hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(GB_T), (size_t)(2 * nc + 1) * 4, s, (const orbx_kp *)(dp + o_kps), n2, img_w,
                           img_h, cols, rows, cell_start, cell_items, cell_of, tmp, d_total);
        hipLaunchKernelGGL(k_window_lists, dim3((nq + 3) / 4), dim3(256), 0, s, (const orbx_kp *)(dp + o_kps), dp + o_desc,
                           cell_start, cell_items, cols, rows, dp + o_qd, (const float *)(dp + o_xy), (const float *)(dp + o_r),
                           (const int32_t *)(dp + o_min), (const int32_t *)(dp + o_max), dp + o_ok, nq, strict ? 1 : 0,
                           sigma2 ? (const float *)(dp + o_s2) : nullptr, (int)std::min(pool_cap, (size_t)INT_MAX), d_counts,
                           d_pool, d_total, d_offs);
        M_TRY(hipGetLastError());
        M_TRY(hipMemcpyAsync(c->h_out.p, c->w_out.p, head + first * 4, hipMemcpyDeviceToHost, s));
        tr.mark("enqueue");
        M_TRY(hipStreamSynchronize(s));
        tr.mark("device + first copy");
        const int32_t *h_total = (const int32_t *)c->h_out.p, *counts = h_total + 4, *offs = counts + nq;
        const uint32_t *pool = (const uint32_t *)(offs + nq);
        const size_t total = (size_t)std::max(h_total[0], 0);
        c->window_last_total = total;
        if (total > pool_cap) on_device = false; // windows longer than the pool: redo on the host
        else if (total > first) {
            M_TRY(hipMemcpyAsync((uint8_t *)c->h_out.p + head + first * 4, (uint8_t *)c->w_out.p + head + first * 4,
                                 (total - first) * 4, hipMemcpyDeviceToHost, s));
            M_TRY(hipStreamSynchronize(s));
        }
        tr.mark("second copy");
        if (on_device) {
            q.e = pool;
            for (int i = 0; i < nq; ++i) {
                if (counts[i] <= 0) continue;
                q.q_idx.push_back(i); q.c_begin.push_back(offs[i]); q.c_len.push_back(counts[i]);
            }
            if (tr.on) fprintf(stderr, "[orbm] %zu queries, %zu candidates\n", q.q_idx.size(), total);
            tr.mark("unpack");
            return ORBX_OK;
        }
        q.q_idx.clear(); q.c_begin.clear(); q.c_len.clear(); q.c_idx.clear();
    }
    FrameGrid grid(kps2, n2, img_w, img_h);
    std::vector<int32_t> all;
    for (int i = 0; i < nq; ++i) {
        if (!q_ok[i]) continue;
        const size_t begin = q.c_idx.size();
        if (!sigma2) grid.area(kps2, q_xy[2 * i], q_xy[2 * i + 1], q_radius[i], q_min[i], q_max[i], q.c_idx, strict);
        else {
            all.clear();
            grid.area(kps2, q_xy[2 * i], q_xy[2 * i + 1], q_radius[i], q_min[i], q_max[i], all, strict);
            const float px = q_xy[2 * i], py = q_xy[2 * i + 1];
            for (int32_t j : all) {
                const float e2 = (px - kps2[j].x) * (px - kps2[j].x) + (py - kps2[j].y) * (py - kps2[j].y);
                if ((double)e2 > 5.991 * (double)sigma2[kps2[j].octave]) continue; // ORBMatcher.cpp:566-567
                q.c_idx.push_back(j);
            }
        }
        if (q.c_idx.size() == begin) continue;
        q.q_idx.push_back(i); q.c_begin.push_back((int32_t)begin); q.c_len.push_back((int32_t)(q.c_idx.size() - begin));
    }
    std::vector<uint16_t> dist;
    int rc = hamming_lists(c, q_desc, nq, desc2, n2, q.q_idx, q.c_begin, q.c_len, q.c_begin, q.c_idx.data(), q.c_idx.size(),
                           q.c_idx.size(), dist);
    if (rc) return rc;
    q.own.resize(q.c_idx.size());
    for (size_t t = 0; t < q.c_idx.size(); ++t) q.own[t] = (uint32_t)dist[t] << 22 | (uint32_t)q.c_idx[t];
    q.e = q.own.data();
    return ORBX_OK;
}

extern "C" int orbm_search_for_initialization(orbm_t *c, float nn_ratio, int check_orientation, const void *kps1v,
                                              const uint8_t *desc1, int n1, const void *kps2v, const uint8_t *desc2,
                                              int n2, int img_w, int img_h, float *pre, int32_t *matches12,
                                              int window_size, int *n_matches)
{
    if (!c || !kps1v || !kps2v || !desc1 || !desc2 || !pre || !matches12 || !n_matches)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    const orbx_kp *kps1 = (const orbx_kp *)kps1v, *kps2 = (const orbx_kp *)kps2v;
    *n_matches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (n1 <= 0 || n2 <= 0) return ORBX_OK;
    // window candidates per level-0 feature of frame1 (:46-54): getFeaturesInArea(prematched, windowSize, level1, level1)
    // on frame2's grid (Frame.cpp:33-51, :97-127)
    std::vector<uint8_t> q_ok((size_t)n1);
    std::vector<int32_t> lv((size_t)n1);
    std::vector<float> rad((size_t)n1, (float)window_size);
    for (int idx1 = 0; idx1 < n1; ++idx1) { lv[idx1] = kps1[idx1].octave; q_ok[idx1] = kps1[idx1].octave <= 0; } // :48 `if (level1 > 0) continue`
    WindowQueries wq;
    int rc = window_candidates(c, false, nullptr, 0, 768, desc1, pre, rad.data(), lv.data(), lv.data(), q_ok.data(), n1, kps2, desc2,
                               n2, img_w, img_h, wq);
    const std::vector<int32_t> &q_idx = wq.q_idx, &c_len = wq.c_len;
    if (rc) return rc;
    std::vector<int> matches21(n2, -1), matchedDist(n2, INT_MAX);
    RotHist rh;
    int num = 0;
    for (size_t k = 0; k < q_idx.size(); ++k) {
        const int idx1 = q_idx[k];
        int bestDist = INT_MAX - 1, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int t = 0; t < c_len[k]; ++t) {
            const int idx2 = wq.idx(k, t);
            const int d = wq.dist(k, t);
            if (matchedDist[idx2] <= d) continue; // :63
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = idx2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= ORBM_TH_LOW && bestDist < orb_round_f((float)bestDist2 * nn_ratio)) { // :74
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; --num; }
            matches12[idx1] = bestIdx2;
            matches21[bestIdx2] = idx1;
            matchedDist[bestIdx2] = bestDist;
            ++num;
            if (check_orientation) rh.add(kps1[idx1].angle, kps2[bestIdx2].angle, idx1);
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        rh.keep3(&i1, &i2, &i3);
        for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx1 : rh.bins[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; --num; }
        }
    }
    for (int idx1 = 0; idx1 < n1; ++idx1)
        if (matches12[idx1] >= 0) { pre[2 * idx1] = kps2[matches12[idx1]].x; pre[2 * idx1 + 1] = kps2[matches12[idx1]].y; }
    *n_matches = num;
    return ORBX_OK;
}


extern "C" int orbm_search_by_projection_frame(orbm_t *c, int check_orientation, const uint8_t *q_desc,
                                               const float *q_xy, const float *q_radius, const int32_t *q_octave,
                                               const float *q_angle, const uint8_t *q_ok, int nq, const void *kps2v,
                                               const uint8_t *desc2, int n2, int img_w, int img_h, int32_t *frame_mp,
                                               int *n_matches)
{
    if (!c || !q_desc || !q_xy || !q_radius || !q_octave || !q_angle || !q_ok || !kps2v || !desc2 || !frame_mp || !n_matches)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    *n_matches = 0;
    if (nq <= 0 || n2 <= 0) return ORBX_OK;
    const orbx_kp *kps2 = (const orbx_kp *)kps2v;
    WindowQueries q;
    std::vector<int32_t> lmin((size_t)nq), lmax((size_t)nq);
    for (int i = 0; i < nq; ++i) { lmin[i] = q_octave[i] - 1; lmax[i] = q_octave[i] + 1; } // :226-229
    int rc = window_candidates(c, false, nullptr, 0, 48, q_desc, q_xy, q_radius, lmin.data(), lmax.data(), q_ok, nq, kps2, desc2,
                               n2, img_w, img_h, q);
    if (rc) return rc;
    RotHist rh;
    int num = 0;
    for (size_t k = 0; k < q.q_idx.size(); ++k) {
        const int i = q.q_idx[k];
        int bestDist = ORBM_TH_HIGH + 1, bestIdx2 = -1;
        for (int t = 0; t < q.c_len[k]; ++t) {
            const int idx2 = q.idx(k, t);
            if (frame_mp[idx2] != -1) continue; // :235 -- includes points matched earlier in this call
            const int d = q.dist(k, t);
            if (d < bestDist) { bestDist = d; bestIdx2 = idx2; }
        }
        if (bestDist <= ORBM_TH_HIGH) { // :244
            frame_mp[bestIdx2] = i;
            ++num;
            if (check_orientation) rh.add(q_angle[i], kps2[bestIdx2].angle, bestIdx2);
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        rh.keep3(&i1, &i2, &i3);
        for (int b = 0; b < ORBM_HISTO_LENGTH; ++b) {
            if (b == i1 || b == i2 || b == i3) continue;
            for (int idx2 : rh.bins[b]) { frame_mp[idx2] = -1; --num; }
        }
    }
    *n_matches = num;
    return ORBX_OK;
}

extern "C" int orbm_search_by_projection_points(orbm_t *c, float nn_ratio, const uint8_t *q_desc, const float *q_xy,
                                                const float *q_radius, const int32_t *q_level, const uint8_t *q_ok,
                                                int nq, const void *kps2v, const uint8_t *desc2, int n2, int img_w,
                                                int img_h, int32_t *frame_mp, int *n_matches, int32_t *counters)
{
    if (!c || !q_desc || !q_xy || !q_radius || !q_level || !q_ok || !kps2v || !desc2 || !frame_mp || !n_matches)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    *n_matches = 0;
    int n_out = 0, fail1 = 0, fail2 = 0;
    if (counters) counters[0] = counters[1] = counters[2] = 0;
    if (nq <= 0) return ORBX_OK;
    const orbx_kp *kps2 = (const orbx_kp *)kps2v;
    WindowQueries q;
    std::vector<int32_t> lmin((size_t)nq);
    for (int i = 0; i < nq; ++i) {
        if (!q_ok[i]) ++n_out; // :355-358
        lmin[i] = q_level[i] - 1;  // :367-369
    }
    int rc = window_candidates(c, false, nullptr, 0, 48, q_desc, q_xy, q_radius, lmin.data(), q_level, q_ok, nq, kps2, desc2,
                               std::max(n2, 0), img_w, img_h, q);
    if (rc) return rc;
    int num = 0;
    for (size_t k = 0; k < q.q_idx.size(); ++k) {
        const int i = q.q_idx[k];
        int bestDist = 256, bestLevel = -1, secondDist = 257, secondLevel = -1, bestIdx = -1;
        for (int t = 0; t < q.c_len[k]; ++t) {
            const int idx = q.idx(k, t);
            if (frame_mp[idx] != -1) continue; // :383 -- a slot holding a good MapPoint (also one assigned in this call)
            const int d = q.dist(k, t);
            if (d < bestDist) { secondDist = bestDist; bestDist = d; secondLevel = bestLevel; bestLevel = kps2[idx].octave; bestIdx = idx; }
            else if (d < secondDist) { secondDist = d; secondLevel = kps2[idx].octave; }
        }
        if (bestDist <= ORBM_TH_HIGH) {
            if (bestLevel == secondLevel && (float)bestDist > nn_ratio * (float)secondDist) { ++fail1; continue; } // :402
            frame_mp[bestIdx] = i;
            ++num;
        } else ++fail2;
    }
    if (counters) { counters[0] = n_out; counters[1] = fail1; counters[2] = fail2; }
    *n_matches = num;
    return ORBX_OK;
}

// static ORBMatcher::SearchByProjection(keyFrame, mapPoints, Map*, th) -- the "fuse" of LocalMapping.cpp:282,301
// (modules/ORB/ORBMatcher.cpp:524-592).  Per map point: KeyFrame window (strict test, levels predictLevel-1 ..
// predictLevel, :554-556), chi-square gate on the re-projection error (:566-567), closest descriptor with
// dist < TH_LOW + 1 (:560, :569-574).  Nothing here depends on the observation rewiring of :578-589, which the caller
// replays in map-point order on its own objects.
extern "C" int orbm_search_fuse(orbm_t *c, const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                                const int32_t *q_level, const uint8_t *q_ok, int nq, const void *kpsv,
                                const uint8_t *desc, int n, int img_w, int img_h, const float *sigma2, int n_levels,
                                int32_t *best_idx, int32_t *best_dist, int *n_found)
{
    if (!c || !q_desc || !q_xy || !q_radius || !q_level || !q_ok || !kpsv || !desc || !sigma2 || !best_idx || !best_dist ||
        !n_found || n_levels < 1)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    *n_found = 0;
    for (int i = 0; i < nq; ++i) { best_idx[i] = -1; best_dist[i] = ORBM_TH_LOW + 1; }
    if (nq <= 0 || n <= 0) return ORBX_OK;
    const orbx_kp *kps = (const orbx_kp *)kpsv;
    for (int j = 0; j < n; ++j)
        if (kps[j].octave < 0 || kps[j].octave >= n_levels) return orbx_set_error(ORBX_E_ARG, "key-point octave outside the sigma2 table");
    WindowQueries q;
    std::vector<int32_t> lmin((size_t)nq);
    for (int i = 0; i < nq; ++i) lmin[i] = q_level[i] - 1; // :556
    int rc = window_candidates(c, true, sigma2, n_levels, 48, q_desc, q_xy, q_radius, lmin.data(), q_level, q_ok, nq, kps,
                               desc, n, img_w, img_h, q);
    if (rc) return rc;
    int found = 0;
    for (size_t k = 0; k < q.q_idx.size(); ++k) {
        int bestDist = ORBM_TH_LOW + 1, bestIdx1 = -1;
        for (int t = 0; t < q.c_len[k]; ++t) {
            const int d = q.dist(k, t);
            if (d < bestDist) { bestDist = d; bestIdx1 = q.idx(k, t); }
        }
        best_idx[q.q_idx[k]] = bestIdx1; best_dist[q.q_idx[k]] = bestDist;
        if (bestIdx1 != -1) ++found;
    }
    *n_found = found;
    return ORBX_OK;
}

// SearchByProjection on a device-resident frame record, greedy pass included: window lists (k_window_lists, packed) and
// k_projection_resolve enqueued back to back on one stream; frame_mp is read and written in device memory.
static int projection_device(orbm_ctx *c, int mode, float nn_ratio, int check_orientation, const uint8_t *d_q_desc,
                             const float *d_q_xy, const float *d_q_radius, const int32_t *d_q_level, const float *d_q_angle,
                             const uint8_t *d_q_ok, int nq, const void *d_kps2, const uint8_t *d_desc2,
                             const int32_t *d_cell_start, const int32_t *d_cell_items, int grid_cols, int grid_rows, int n2,
                             int list_cap, int32_t *d_frame_mp, int32_t *d_result, void *stream)
{
    if (!c || !d_q_desc || !d_q_xy || !d_q_radius || !d_q_level || !d_q_ok || !d_kps2 || !d_desc2 || !d_cell_start ||
        !d_cell_items || !d_frame_mp || !d_result || (mode == 0 && !d_q_angle) || grid_cols < 1 || grid_rows < 1 || nq < 0 ||
        n2 < 0 || list_cap < 1)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (n2 >= (1 << 22)) return orbx_set_error(ORBX_E_UNSUPPORTED, "more than 2^22 key points");
    const size_t lds = ((size_t)n2 + (size_t)nq) * 4;
    if (lds > 150 * 1024) return orbx_set_error(ORBX_E_UNSUPPORTED, "nq + n2 above 38400: the greedy pass keeps both in LDS");
    M_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    // scratch: [total, pad x3][counts nq][offs nq][lo nq][hi nq][pool nq * list_cap]
    const size_t pool_cap = (size_t)nq * list_cap, head = 16 + (size_t)nq * 16;
    M_TRY(c->w_out.need(head + pool_cap * 4 + 16));
    int32_t *d_total = (int32_t *)c->w_out.p, *d_counts = d_total + 4, *d_offs = d_counts + nq, *d_lo = d_offs + nq, *d_hi = d_lo + nq;
    uint32_t *d_pool = (uint32_t *)(d_hi + nq);
    M_TRY(hipMemsetAsync(d_total, 0, 16, s));
    if (nq > 0) {
        hipLaunchKernelGGL(k_projection_levels, dim3((nq + 255) / 256), dim3(256), 0, s, d_q_level, nq, mode == 0 ? 1 : 0, d_lo, d_hi);
        hipLaunchKernelGGL(k_window_lists, dim3((nq + 3) / 4), dim3(256), 0, s, (const orbx_kp *)d_kps2, d_desc2, d_cell_start,
                           d_cell_items, grid_cols, grid_rows, d_q_desc, d_q_xy, d_q_radius, d_lo, d_hi, d_q_ok, nq, 0, nullptr,
                           (int)std::min(pool_cap, (size_t)INT_MAX), d_counts, d_pool, d_total, d_offs);
    }
    // once per device, thread-safe (the matcher entry points are re-entrant)
    M_TRY(orbx_lds_opt_in(reinterpret_cast<const void *>(k_projection_resolve<0>), 150 * 1024));
    M_TRY(orbx_lds_opt_in(reinterpret_cast<const void *>(k_projection_resolve<1>), 150 * 1024));
    if (mode == 0)
        hipLaunchKernelGGL(k_projection_resolve<0>, dim3(1), dim3(PR_T), lds, s, d_counts, d_offs, d_pool, (int)std::min(pool_cap, (size_t)INT_MAX),
                           d_total, nq, n2, (const orbx_kp *)d_kps2, d_q_angle, nn_ratio, check_orientation, d_frame_mp, d_result);
    else
        hipLaunchKernelGGL(k_projection_resolve<1>, dim3(1), dim3(PR_T), lds, s, d_counts, d_offs, d_pool, (int)std::min(pool_cap, (size_t)INT_MAX),
                           d_total, nq, n2, (const orbx_kp *)d_kps2, d_q_angle, nn_ratio, check_orientation, d_frame_mp, d_result);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}
extern "C" int orbm_search_by_projection_frame_device(orbm_t *c, int check_orientation, const uint8_t *d_q_desc, const float *d_q_xy,
                                                      const float *d_q_radius, const int32_t *d_q_octave, const float *d_q_angle,
                                                      const uint8_t *d_q_ok, int nq, const void *d_kps2, const uint8_t *d_desc2,
                                                      const int32_t *d_cell_start, const int32_t *d_cell_items, int grid_cols,
                                                      int grid_rows, int n2, int list_cap, int32_t *d_frame_mp, int32_t *d_result,
                                                      void *stream)
{
    return projection_device(c, 0, 0.f, check_orientation, d_q_desc, d_q_xy, d_q_radius, d_q_octave, d_q_angle, d_q_ok, nq, d_kps2,
                             d_desc2, d_cell_start, d_cell_items, grid_cols, grid_rows, n2, list_cap, d_frame_mp, d_result, stream);
}
extern "C" int orbm_search_by_projection_points_device(orbm_t *c, float nn_ratio, const uint8_t *d_q_desc, const float *d_q_xy,
                                                       const float *d_q_radius, const int32_t *d_q_level, const uint8_t *d_q_ok, int nq,
                                                       const void *d_kps2, const uint8_t *d_desc2, const int32_t *d_cell_start,
                                                       const int32_t *d_cell_items, int grid_cols, int grid_rows, int n2, int list_cap,
                                                       int32_t *d_frame_mp, int32_t *d_result, void *stream)
{
    return projection_device(c, 1, nn_ratio, 0, d_q_desc, d_q_xy, d_q_radius, d_q_level, nullptr, d_q_ok, nq, d_kps2, d_desc2,
                             d_cell_start, d_cell_items, grid_cols, grid_rows, n2, list_cap, d_frame_mp, d_result, stream);
}

// The static fuse's per-point search (ORBMatcher.cpp:556-575) on a device-resident key-frame record: the window lists with the
// KeyFrame's strict window test and the chi-square gate (k_window_lists), then the closest hit of every list.  What the
// reference does with the hit (:577-589: observations added, map points replaced) mutates its objects and stays with the caller.
__global__ void k_fuse_levels(const int32_t *__restrict__ lv, int n, int32_t *__restrict__ lo) // predictLevel - 1 (:556)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lo[i] = lv[i] - 1;
}
__global__ __launch_bounds__(256) void k_fuse_best(const int32_t *__restrict__ counts, const uint32_t *__restrict__ lists, int cap, int nq,
                                                   int32_t *__restrict__ best_idx, int32_t *__restrict__ best_dist,
                                                   int32_t *__restrict__ result)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    int found = 0, over = 0;
    if (q < nq) {
        const int n = counts[q];
        int bestDist = ORBM_TH_LOW + 1, bestIdx1 = -1; // :563
        over = n > cap;
        const uint32_t *e = lists + (size_t)q * cap;
        for (int t = 0; t < min(n, cap); ++t) {
            const int d = (int)(e[t] >> 22);
            if (d < bestDist) { bestDist = d; bestIdx1 = (int)(e[t] & 0x3FFFFFu); } // :570-573, list order = the reference's
        }
        best_idx[q] = bestIdx1; best_dist[q] = bestDist;
        found = bestIdx1 != -1;
    }
    const int f = __syncthreads_count(found), o = __syncthreads_or(over);
    if (threadIdx.x == 0) {
        if (f) atomicAdd(&result[0], f);
        if (o) atomicExch(&result[1], 1);
    }
}
extern "C" int orbm_search_fuse_device(orbm_t *c, const uint8_t *d_q_desc, const float *d_q_xy, const float *d_q_radius,
                                       const int32_t *d_q_level, const uint8_t *d_q_ok, int nq, const void *d_kps, const uint8_t *d_desc,
                                       const int32_t *d_cell_start, const int32_t *d_cell_items, int grid_cols, int grid_rows,
                                       const float *d_sigma2, int list_cap, int32_t *d_best_idx, int32_t *d_best_dist,
                                       int32_t *d_result, void *stream)
{
    if (!c || !d_q_desc || !d_q_xy || !d_q_radius || !d_q_level || !d_q_ok || !d_kps || !d_desc || !d_cell_start || !d_cell_items ||
        !d_sigma2 || !d_best_idx || !d_best_dist || !d_result || grid_cols < 1 || grid_rows < 1 || nq < 0 || list_cap < 1)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    M_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    M_TRY(hipMemsetAsync(d_result, 0, 32, s));
    if (nq == 0) return ORBX_OK;
    // scratch: [counts nq][lo nq][lists nq * list_cap]
    const size_t N = (size_t)nq;
    M_TRY(c->w_out.need(N * 8 + N * list_cap * 4 + 16));
    int32_t *d_counts = (int32_t *)c->w_out.p, *d_lo = d_counts + N;
    uint32_t *d_lists = (uint32_t *)(d_lo + N);
    hipLaunchKernelGGL(k_fuse_levels, dim3((nq + 255) / 256), dim3(256), 0, s, d_q_level, nq, d_lo);
    hipLaunchKernelGGL(k_window_lists, dim3((nq + 3) / 4), dim3(256), 0, s, (const orbx_kp *)d_kps, d_desc, d_cell_start, d_cell_items,
                       grid_cols, grid_rows, d_q_desc, d_q_xy, d_q_radius, d_lo, d_q_level, d_q_ok, nq, 1, d_sigma2, list_cap, d_counts,
                       d_lists, nullptr, nullptr);
    hipLaunchKernelGGL(k_fuse_best, dim3((nq + 255) / 256), dim3(256), 0, s, d_counts, d_lists, list_cap, nq, d_best_idx, d_best_dist, d_result);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}

// SearchForInitialization on device-resident records (include/orbm.h)
extern "C" int orbm_search_for_initialization_device(orbm_t *c, float nn_ratio, int check_orientation, const void *d_kps1,
                                                     const uint8_t *d_desc1, int n1, const void *d_kps2, const uint8_t *d_desc2,
                                                     const int32_t *d_cell_start2, const int32_t *d_cell_items2, int grid_cols,
                                                     int grid_rows, int n2, float *d_pre, int window_size, int list_cap,
                                                     int32_t *d_matches12, int32_t *d_result, void *stream)
{
    if (!c || !d_kps1 || !d_desc1 || !d_kps2 || !d_desc2 || !d_cell_start2 || !d_cell_items2 || !d_pre || !d_matches12 || !d_result ||
        grid_cols < 1 || grid_rows < 1 || n1 < 0 || n2 < 0 || list_cap < 1 || window_size < 0)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (n2 >= (1 << 22)) return orbx_set_error(ORBX_E_UNSUPPORTED, "more than 2^22 key points");
    const size_t lds = ((size_t)n2 + 5 * (size_t)n1) * 4;
    if (lds > 150 * 1024) return orbx_set_error(ORBX_E_UNSUPPORTED, "n2 + 5 n1 above 38400: the resolve keeps the claims in LDS");
    M_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    M_TRY(hipMemsetAsync(d_result, 0, 32, s));
    if (n1 == 0) return ORBX_OK;
    if (n2 == 0) { M_TRY(hipMemsetAsync(d_matches12, 0xFF, sizeof(int32_t) * (size_t)n1, s)); return ORBX_OK; }
    // scratch: [total, pad x3][counts n1][offs n1][level n1][radius n1][ok n1 bytes, padded][pool n1 * list_cap]
    const size_t pool_cap = (size_t)n1 * list_cap, N = (size_t)n1, okb = (N + 15) / 16 * 16;
    M_TRY(c->w_out.need(16 + N * 16 + okb + pool_cap * 4 + 16));
    int32_t *d_total = (int32_t *)c->w_out.p, *d_counts = d_total + 4, *d_offs = d_counts + N, *d_lv = d_offs + N;
    float *d_r = (float *)(d_lv + N);
    uint8_t *d_ok = (uint8_t *)(d_r + N);
    uint32_t *d_pool = (uint32_t *)(d_ok + okb);
    M_TRY(hipMemsetAsync(d_total, 0, 16, s));
    hipLaunchKernelGGL(k_init_queries, dim3((n1 + 255) / 256), dim3(256), 0, s, (const orbx_kp *)d_kps1, n1, (float)window_size, d_ok, d_r, d_lv);
    hipLaunchKernelGGL(k_window_lists, dim3((n1 + 3) / 4), dim3(256), 0, s, (const orbx_kp *)d_kps2, d_desc2, d_cell_start2, d_cell_items2,
                       grid_cols, grid_rows, d_desc1, d_pre, d_r, d_lv, d_lv, d_ok, n1, 0, nullptr, (int)std::min(pool_cap, (size_t)INT_MAX),
                       d_counts, d_pool, d_total, d_offs);
    M_TRY(orbx_lds_opt_in(reinterpret_cast<const void *>(k_init_resolve), 150 * 1024));
    hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(PR_T), lds, s, d_counts, d_offs, d_pool, (int)std::min(pool_cap, (size_t)INT_MAX), d_total,
                       n1, n2, (const orbx_kp *)d_kps1, (const orbx_kp *)d_kps2, nn_ratio, check_orientation, ORBM_INIT_MAX_SWEEPS, c->init_lanes, d_matches12,
                       d_pre, d_result);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}

// The window lists of one device-resident frame record (orbx_extract_batch_device -> orbf_frame_post_device): the grid
// is the CSR orbf built, nothing visits the host.  See include/orbm.h.

// SearchByBow / SearchForTriangulation on device-resident records (include/orbm.h)
static int bow_device(orbm_ctx *c, int mode, float nn_ratio, int check_orientation, const uint8_t *d_desc1, const void *d_kps1,
                      const uint8_t *d_mask1, int n1, const uint32_t *d_fv1_nodes, const int32_t *d_fv1_off, const uint32_t *d_fv1_idx,
                      const int32_t *d_n_fv1, const uint8_t *d_desc2, const void *d_kps2, const uint8_t *d_busy2, int32_t *d_frame_mp,
                      int32_t *d_matches12, int n2, const uint32_t *d_fv2_nodes, const int32_t *d_fv2_off, const uint32_t *d_fv2_idx,
                      const int32_t *d_n_fv2, int32_t *d_result, void *stream)
{
    if (!c || !d_desc1 || !d_desc2 || !d_mask1 || !d_fv1_nodes || !d_fv1_off || !d_fv1_idx || !d_n_fv1 || !d_fv2_nodes || !d_fv2_off ||
        !d_fv2_idx || !d_n_fv2 || !d_result || !d_kps1 || !d_kps2 || (mode == 0 ? !d_frame_mp : (!d_matches12 || !d_busy2)))
        return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n1 < 0 || n2 < 0) return orbx_set_error(ORBX_E_ARG, "bad size");
    M_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    M_TRY(hipMemsetAsync(d_result, 0, 32, s));
    if (mode == 1 && n1 > 0) M_TRY(hipMemsetAsync(d_matches12, 0xFF, sizeof(int32_t) * (size_t)n1, s)); // -1 (:457)
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    // scratch (the handle's, one call in flight per handle): node_p2[n1], node_qbegin[n1 + 1], q_idx / c_begin / c_len / q_node [n1],
    // n_queries, hist[30], match_list[2 n1], topk[n1 * K]
    const size_t N = (size_t)n1;
    M_TRY(c->w_grid.need(sizeof(int32_t) * (N * 8 + 64 + N * BR_TOPK + 8)));
    int32_t *node_p2 = (int32_t *)c->w_grid.p, *node_qbegin = node_p2 + N, *q_idx = node_qbegin + N + 1, *c_begin = q_idx + N, *c_len = c_begin + N,
            *q_node = c_len + N, *n_queries = q_node + N, *hist = n_queries + 1, *match_list = hist + 32;
    uint32_t *topk = (uint32_t *)(match_list + 2 * N);
    M_TRY(hipMemsetAsync(n_queries, 0, sizeof(int32_t) * 33, s));
    const BowFv f1 = {d_fv1_nodes, d_fv1_off, d_fv1_idx, d_n_fv1}, f2 = {d_fv2_nodes, d_fv2_off, d_fv2_idx, d_n_fv2};
    hipLaunchKernelGGL(k_bow_queries, dim3(1), dim3(BQ_T), 0, s, f1, f2, d_mask1, mode == 0 ? 1 : 0, n1, n1, node_p2, node_qbegin, q_idx, c_begin,
                       c_len, q_node, n_queries, d_result);
    hipLaunchKernelGGL(k_topk_lists_n<BR_TOPK>, dim3((n1 + 3) / 4), dim3(256), 0, s, d_desc1, d_desc2, q_idx, c_begin, c_len, n_queries,
                       reinterpret_cast<const int32_t *>(d_fv2_idx), mode == 0 ? d_frame_mp : nullptr, d_busy2, topk);
    const int grid = std::min(n1, 1024);
    if (mode == 0)
        hipLaunchKernelGGL(k_bow_resolve<0>, dim3(grid), dim3(BR_T), 0, s, f1, f2, n1, node_p2, node_qbegin, n_queries, q_idx, topk, d_desc1, d_desc2,
                           (const orbx_kp *)d_kps1, (const orbx_kp *)d_kps2, d_busy2, nn_ratio, check_orientation, d_frame_mp, d_matches12, hist, match_list, d_result);
    else
        hipLaunchKernelGGL(k_bow_resolve<1>, dim3(grid), dim3(BR_T), 0, s, f1, f2, n1, node_p2, node_qbegin, n_queries, q_idx, topk, d_desc1, d_desc2,
                           (const orbx_kp *)d_kps1, (const orbx_kp *)d_kps2, d_busy2, nn_ratio, check_orientation, d_frame_mp, d_matches12, hist, match_list, d_result);
    if (mode == 0)
        hipLaunchKernelGGL(k_bow_finish<0>, dim3(1), dim3(256), 0, s, check_orientation, hist, match_list, d_frame_mp, d_matches12, d_result);
    else
        hipLaunchKernelGGL(k_bow_finish<1>, dim3(1), dim3(256), 0, s, check_orientation, hist, match_list, d_frame_mp, d_matches12, d_result);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}
extern "C" int orbm_search_by_bow_device(orbm_t *c, float nn_ratio, int check_orientation, const uint8_t *d_desc1, const void *d_kps1,
                                         const uint8_t *d_kf_mp_ok, int n1, const uint32_t *d_fv1_nodes, const int32_t *d_fv1_off,
                                         const uint32_t *d_fv1_idx, const int32_t *d_n_fv1, const uint8_t *d_desc2, const void *d_kps2,
                                         int32_t *d_frame_mp, int n2, const uint32_t *d_fv2_nodes, const int32_t *d_fv2_off,
                                         const uint32_t *d_fv2_idx, const int32_t *d_n_fv2, int32_t *d_result, void *stream)
{
    return bow_device(c, 0, nn_ratio, check_orientation, d_desc1, d_kps1, d_kf_mp_ok, n1, d_fv1_nodes, d_fv1_off, d_fv1_idx, d_n_fv1, d_desc2,
                      d_kps2, nullptr, d_frame_mp, nullptr, n2, d_fv2_nodes, d_fv2_off, d_fv2_idx, d_n_fv2, d_result, stream);
}
extern "C" int orbm_search_for_triangulation_device(orbm_t *c, int check_orientation, const uint8_t *d_desc1, const void *d_kps1,
                                                    const uint8_t *d_has_mp1, int n1, const uint32_t *d_fv1_nodes, const int32_t *d_fv1_off,
                                                    const uint32_t *d_fv1_idx, const int32_t *d_n_fv1, const uint8_t *d_desc2,
                                                    const void *d_kps2, const uint8_t *d_has_mp2, int n2, const uint32_t *d_fv2_nodes,
                                                    const int32_t *d_fv2_off, const uint32_t *d_fv2_idx, const int32_t *d_n_fv2,
                                                    int32_t *d_matches12, int32_t *d_result, void *stream)
{
    return bow_device(c, 1, 0.f, check_orientation, d_desc1, d_kps1, d_has_mp1, n1, d_fv1_nodes, d_fv1_off, d_fv1_idx, d_n_fv1, d_desc2,
                      d_kps2, d_has_mp2, nullptr, d_matches12, n2, d_fv2_nodes, d_fv2_off, d_fv2_idx, d_n_fv2, d_result, stream);
}

extern "C" int orbm_window_lists_device(orbm_t *c, const void *d_kps, const uint8_t *d_desc, const int32_t *d_cell_start,
                                        const int32_t *d_cell_items, int grid_cols, int grid_rows, const uint8_t *d_q_desc,
                                        const float *d_q_xy, const float *d_q_radius, const int32_t *d_q_min_level,
                                        const int32_t *d_q_max_level, const uint8_t *d_q_ok, int nq, int strict,
                                        const float *d_sigma2, int cap, int32_t *d_counts, uint32_t *d_lists, void *stream)
{
    if (!c || !d_kps || !d_desc || !d_cell_start || !d_cell_items || !d_q_desc || !d_q_xy || !d_q_radius || !d_q_min_level ||
        !d_q_max_level || !d_q_ok || !d_counts || !d_lists || grid_cols < 1 || grid_rows < 1 || cap < 1 || nq < 0)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    if (nq == 0) return ORBX_OK;
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    hipLaunchKernelGGL(k_window_lists, dim3((nq + 3) / 4), dim3(256), 0, s, (const orbx_kp *)d_kps, d_desc, d_cell_start,
                       d_cell_items, grid_cols, grid_rows, d_q_desc, d_q_xy, d_q_radius, d_q_min_level, d_q_max_level, d_q_ok,
                       nq, strict ? 1 : 0, d_sigma2, cap, d_counts, d_lists, nullptr, nullptr);
    M_TRY(hipGetLastError());
    return ORBX_OK;
}
