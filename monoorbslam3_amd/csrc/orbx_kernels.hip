// HIP kernels of the ORB extractor for gfx950 (wave64).
//
// Stage -> reference code it replaces (paths relative to the reference root):
//   k_resize       cv::resize chain            modules/ORB/ORBExtractor.cpp:559-570
//   k_fast_cells   per-cell cv::FAST + retry   modules/ORB/ORBExtractor.cpp:592-617
//   k_blur7        cv::GaussianBlur 7x7 s=2    modules/ORB/ORBExtractor.cpp:527-528
//   k_octree       DistributeOctree            modules/ORB/ORBExtractor.cpp:640-830
//   k_orient_desc  IC_Angle + rBRIEF + output  modules/ORB/ORBExtractor.cpp:18-97, :507-546, :626-637
#include <string.h>

#include <vector>

#include "orbx_internal.h"
#include <atomic>
#include "orb_math.h"
#include "orb_pattern.h"

typedef unsigned long long u64;

// XCD-aware work mapping (speed only, never correctness).  Workgroups are dealt round-robin over the 8 XCDs,
// each with a private 4 MB L2; linear block id -> (xcd = id % 8, j = id / 8) and XCD x takes the frames
// congruent to x (mod 8), so everything that touches one frame's pyramid meets in one L2.
// A call with fewer than 8 frames would leave XCDs idle that way -- a single frame would run on ONE XCD, an eighth of
// the chip, which is what a latency-bound call can least afford -- so below 8 frames the blocks are dealt in plain order
// (frame = id / per_frame) and every frame spreads over all XCDs.
// Launch with orbx_xcd_grid(per_frame, n_frames) blocks.
__device__ __forceinline__ bool xcd_remap(int per_frame, int n_frames, int *frame, int *item)
{
    if (n_frames < 8) {
        const int id = blockIdx.x, f = id / per_frame;
        *item = id - f * per_frame;
        *frame = f;
        return f < n_frames;
    }
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int g = j / per_frame;
    *item = j - g * per_frame;
    *frame = g * 8 + xcd;
    return *frame < n_frames;
}
static inline unsigned orbx_xcd_grid(int per_frame, int n_frames)
{
    if (n_frames < 8) return (unsigned)n_frames * (unsigned)per_frame;
    return 8u * (unsigned)((n_frames + 7) / 8) * (unsigned)per_frame;
}

// Development hook (orbx_dev_set_lds_pad, not declared in include/): extra dynamic LDS per workgroup of a batch kernel, so that a
// tool can cap the kernel's occupancy and measure what a co-resident partner would leave it (profiles/r06_pipeline_budget.md).
// Stage ids: 0 k_resize_lds, 1 k_fast_strip, 2 oct_batch::k_octree_lds, 3 k_orient<false>, 4 k_blur_desc.  0 bytes = as shipped.
static int g_dev_lds_pad[5] = {0, 0, 0, 0, 0};
extern "C" int orbx_dev_set_lds_pad(int stage, int bytes)
{
    if (stage < 0 || stage >= 5 || bytes < 0 || bytes > 159 * 1024) return ORBX_E_ARG;
    g_dev_lds_pad[stage] = bytes;
    return ORBX_OK;
}
template <typename K> static size_t dev_pad(K kernel, int stage, size_t static_and_dynamic)
{
    const size_t pad = (size_t)g_dev_lds_pad[stage];
    if (pad) (void)orbx_lds_opt_in(reinterpret_cast<const void *>(kernel), static_and_dynamic + pad);
    return pad;
}

__constant__ __attribute__((aligned(16))) int8_t c_pattern[ORB_PATTERN_POINTS * 2] = {ORB_PATTERN_INT8_LIST};

// ---------------------------------------------------------------------------------------------
// Pyramid: fixed-point bilinear (11-bit taps), 4 output pixels per thread
// ---------------------------------------------------------------------------------------------
// Each thread produces 4 adjacent output pixels of one row.  Their source columns span at most 8
// bytes (scale < 2), so the two source rows are fetched with one unaligned 64-bit load each and the
// taps pick bytes out of the registers; the 4 x (offset, c0, c1) column taps come in as two 128-bit
// loads.
struct __attribute__((packed, aligned(1))) UnalignedU64 { unsigned long long v; };

#define RS_ROWS 8 // output rows per thread: 2*RS_ROWS independent 64-bit loads in flight per thread
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src, size_t src_fs, int src_pitch, int sw,
                                                int sh, uint8_t *__restrict__ dst, size_t dst_fs, int dst_pitch,
                                                int dw, int dh, const OrbxTap *__restrict__ xtap,
                                                const OrbxTap *__restrict__ ytap, int gx, int gy, int n_frames,
                                                int *__restrict__ zero_counts)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    int frame, blk; // a frame's blocks share one XCD: source rows used by two block rows are fetched once
    if (!xcd_remap(gx * gy, n_frames, &frame, &blk)) return;
    // a call with a few frames saves the memset launch: the first resize clears the frame's FAST candidate counters
    if (zero_counts && blk == 0 && threadIdx.y == 0 && threadIdx.x < ORBX_MAX_LEVELS) zero_counts[frame * ORBX_MAX_LEVELS + threadIdx.x] = 0;
    const int by = blk / gx, bx = blk - by * gx;
    const int dx0 = (bx * 64 + threadIdx.x) * 4;
    // a wave is one row of the 64 x 4 block, so its output rows and their row taps are wave-uniform: scalar loads
    const int dy0 = (by * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * RS_ROWS;
    if (dx0 >= dw || dy0 >= dh) return;
    const uint8_t *S = src + (size_t)frame * src_fs;
    uint8_t *D = dst + (size_t)frame * dst_fs;
    // the tap table is padded to a multiple of 4 entries (host side), 32 bytes per thread
    const uint4 t01 = reinterpret_cast<const uint4 *>(xtap + dx0)[0], t23 = reinterpret_cast<const uint4 *>(xtap + dx0)[1];
    int b0[RS_ROWS], b1[RS_ROWS], sy0[RS_ROWS], sy1[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const OrbxTap ty = ytap[min(dy0 + r, dh - 1)];
        b0[r] = ty.c0; b1[r] = ty.c1;
        sy0[r] = min(max(ty.ofs, 0), sh - 1);
        sy1[r] = min(max(ty.ofs + 1, 0), sh - 1);
    }
    const int ofs[4] = {(int)t01.x, (int)t01.z, (int)t23.x, (int)t23.z};
    const uint32_t cc[4] = {t01.y, t01.w, t23.y, t23.w}; // c0 | c1 << 16
    const int sx0 = ofs[0];
    // All 2 * RS_ROWS source windows are requested together, without a branch: a window that would run past the end of
    // the row (right image edge; level 0 is the caller's buffer, nothing may be read beyond it) is fetched from the last 8
    // bytes of the row instead and shifted into place -- the bytes that fall off are zero, and the taps never select them.
    const int sxl = min(sx0, sw - 8);
    const uint32_t sft = (uint32_t)(sx0 - sxl) * 8u; // 0 .. 56
    unsigned long long w0[RS_ROWS], w1[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        w0[r] = reinterpret_cast<const UnalignedU64 *>(S + (size_t)sy0[r] * src_pitch + sxl)->v;
        w1[r] = reinterpret_cast<const UnalignedU64 *>(S + (size_t)sy1[r] * src_pitch + sxl)->v;
    }
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        w0[r] >>= sft;
        w1[r] >>= sft;
    }
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sel[i] = (uint32_t)(ofs[i] - sx0) * 0x00010001u + 0x0c010c00u; // bytes (o, o+1) -> 16-bit halves
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        if (dy0 + r >= dh) break;
        const uint32_t w0l = (uint32_t)w0[r], w0h = (uint32_t)(w0[r] >> 32), w1l = (uint32_t)w1[r], w1h = (uint32_t)(w1[r] >> 32);
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u16x2 cv = __builtin_bit_cast(u16x2, cc[i]);
            const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(w0h, w0l, sel[i])), cv, 0u, false);
            const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(w1h, w1l, sel[i])), cv, 0u, false);
            const int v = (((int)__umul24((uint32_t)b0[r], (uint32_t)(r0 >> 4)) >> 16) +
                           ((int)__umul24((uint32_t)b1[r], (uint32_t)(r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)v << (8 * i); // v <= 255: the taps of each pass sum to 2048
        }
        // rows of the arena are 64-byte aligned and padded, so the dword store is always in bounds
        *reinterpret_cast<uint32_t *>(D + (size_t)(dy0 + r) * dst_pitch + dx0) = packed;
    }
}

// 4 adjacent output pixels from two 8-byte source windows (w0 = row sy0, w1 = row sy1), k_resize's inner loop
__device__ __forceinline__ uint32_t resize_quad(uint32_t w0l, uint32_t w0h, uint32_t w1l, uint32_t w1h, const uint32_t sel[4],
                                                const uint32_t cc[4], int b0, int b1)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u16x2 cv = __builtin_bit_cast(u16x2, cc[i]);
        const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(w0h, w0l, sel[i])), cv, 0u, false);
        const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(w1h, w1l, sel[i])), cv, 0u, false);
        const int v = (((int)__umul24((uint32_t)b0, (uint32_t)(r0 >> 4)) >> 16) + ((int)__umul24((uint32_t)b1, (uint32_t)(r1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)v << (8 * i);
    }
    return packed;
}

// ---------------------------------------------------------------------------------------------
// The same resampling with the source tile staged through LDS (resident batches).  k_resize's 8-byte windows overlap
// from lane to lane and start at any byte: every window costs the L1 two tag look-ups, 32 per wave instruction, and the
// counters put the kernel at 0.6 look-ups per CU-cycle with 62 % of its wave cycles parked on memory
// (profiles/r03_tcp_counters.txt, r03_sq_breakdown.txt; with the windows forced to aligned addresses -- wrong pixels, a
// timing experiment -- the pyramid of 512 frames took 0.38 instead of 0.47 ms).  Here a workgroup's 256 x 32 output tile
// first brings its (<= 352 x 42) source tile in with 16-byte loads, each byte once, a row per half-wave, and the windows
// are read from LDS (three aligned dwords + two v_alignbit).  Same arithmetic as k_resize, pixel for pixel.
// (Built on top of it and dropped: two levels per launch -- a 128 x 32 tile of level l+2 from an LDS patch of level l+1 from an
// LDS-staged piece of level l, k_resize2's ownership rule, bit-identical -- takes a quarter of the pyramid's HBM traffic away and
// 0.47-0.48 ms per 512 frames against 0.35: three phases and two barriers per workgroup cost more than the bytes saved.)
// ---------------------------------------------------------------------------------------------
#define RL_NC 22                // 16-byte chunks per staged row
#define RL_LP (RL_NC * 16 + 16) // LDS pitch: a window's third dword may lie 12 bytes past the last needed column
#ifndef RL_WAVES
#define RL_WAVES 4              // waves per workgroup = 8-row bands per tile (1 / 2 / 4 measured: 0.344 / 0.351 / 0.350 ms per 512 frames)
#endif
#define RL_NR (RL_WAVES * 10 + 2) // staged rows
#define RL_ROUNDS ((RL_NR + 2 * RL_WAVES - 1) / (2 * RL_WAVES))
__global__ __launch_bounds__(64 * RL_WAVES) void k_resize_lds(const uint8_t *__restrict__ src, size_t src_fs, int src_pitch, int sw,
                                                    int sh, int last_row_bytes, uint8_t *__restrict__ dst, size_t dst_fs,
                                                    int dst_pitch, int dw, int dh, const OrbxTap *__restrict__ xtap,
                                                    const OrbxTap *__restrict__ ytap, int gx, int gy, int n_frames,
                                                    int *__restrict__ zero_counts)
{
    __shared__ __align__(16) uint8_t tile[RL_NR * RL_LP];
    int frame, blk;
    if (!xcd_remap(gx * gy, n_frames, &frame, &blk)) return;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (zero_counts && blk == 0 && tid < ORBX_MAX_LEVELS) zero_counts[frame * ORBX_MAX_LEVELS + tid] = 0;
    const int by = blk / gx, bx = blk - by * gx;
    const int X0 = bx * 256, Y0 = by * RL_WAVES * RS_ROWS, Xl = min(X0 + 255, dw - 1), Yl = min(Y0 + RL_WAVES * RS_ROWS - 1, dh - 1);
    // source tile: columns [colbase, colbase + 16 ncol), rows [r_lo, r_lo + nrow) -- the host checked that it fits
    const int colbase = xtap[X0].ofs & ~15;
    const int ncol = ((min(xtap[Xl].ofs + 1, sw - 1) - colbase) >> 4) + 1;
    const int r_lo = min(max(ytap[Y0].ofs, 0), sh - 1);
    const int nrow = min(max(ytap[Yl].ofs + 1, 0), sh - 1) - r_lo + 1;
    const uint8_t *S = src + (size_t)frame * src_fs;
    uint8_t *D = dst + (size_t)frame * dst_fs;
    {
        struct __attribute__((packed, aligned(1))) U128 { uint32_t w[4]; };
        const int ch = tid & 31, rs = tid >> 5;
        U128 v[RL_ROUNDS];
        bool slow[RL_ROUNDS];
#pragma unroll
        for (int k = 0; k < RL_ROUNDS; ++k) {
            const int row = rs + 2 * RL_WAVES * k, y = r_lo + row, c = colbase + 16 * ch;
            const bool on = ch < ncol && row < nrow;
            // the last row of a caller's image ends at its last pixel: a chunk that would pass it is fetched byte by byte
            slow[k] = on && y == sh - 1 && c + 16 > last_row_bytes;
            v[k].w[0] = v[k].w[1] = v[k].w[2] = v[k].w[3] = 0;
            if (on && !slow[k]) v[k] = *reinterpret_cast<const U128 *>(S + (size_t)y * src_pitch + c);
            if (slow[k])
                for (int q = 0; q < 16; ++q)
                    if (c + q < last_row_bytes) v[k].w[q >> 2] |= (uint32_t)S[(size_t)y * src_pitch + c + q] << (8 * (q & 3));
        }
#pragma unroll
        for (int k = 0; k < RL_ROUNDS; ++k) {
            const int row = rs + 2 * RL_WAVES * k;
            if (ch < ncol && row < nrow)
                *reinterpret_cast<uint4 *>(&tile[row * RL_LP + 16 * ch]) = make_uint4(v[k].w[0], v[k].w[1], v[k].w[2], v[k].w[3]);
        }
    }
    __syncthreads();
    const int dx0 = X0 + 4 * (int)threadIdx.x;
    const int dy0 = Y0 + __builtin_amdgcn_readfirstlane((int)threadIdx.y) * RS_ROWS; // wave-uniform: row taps through the scalar cache
    if (dx0 >= dw || dy0 >= dh) return;
    const uint4 t01 = reinterpret_cast<const uint4 *>(xtap + dx0)[0], t23 = reinterpret_cast<const uint4 *>(xtap + dx0)[1];
    const int ofs[4] = {(int)t01.x, (int)t01.z, (int)t23.x, (int)t23.z};
    const uint32_t cc[4] = {t01.y, t01.w, t23.y, t23.w}; // c0 | c1 << 16
    const int a = ofs[0] - colbase;
    const uint32_t sh8 = (uint32_t)(a & 3) * 8u;
    const uint8_t *tcol = tile + (a & ~3);
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sel[i] = (uint32_t)(ofs[i] - ofs[0]) * 0x00010001u + 0x0c010c00u; // bytes (o, o+1) -> 16-bit halves
    // The horizontal step of a SOURCE row is computed once and kept for the output rows that use it: at scale 1.2 an output row's
    // two source rows are (s, s + 1) and the next one's (s + 1, s + 2) four times out of five, so eight output rows take about
    // 10.4 horizontal rows instead of 16 (LDS reads, window shifts, byte selects and dot products all go down by a third).  Which
    // rows are at hand is wave-uniform (the row taps come through the scalar cache): the choice is a scalar branch.  The value
    // kept is already the ">> 4" of cv::resize's vertical step.  Same integers as resize_quad, pixel for pixel.
    typedef unsigned short rl_u16x2 __attribute__((ext_vector_type(2)));
    auto hrow = [&](int rr, uint32_t g[4]) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(tcol + rr * RL_LP);
        const uint32_t a0 = p[0], a1 = p[1], a2 = p[2];
        const uint32_t wl = __builtin_amdgcn_alignbit(a1, a0, sh8), wh = __builtin_amdgcn_alignbit(a2, a1, sh8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            g[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(rl_u16x2, __builtin_amdgcn_perm(wh, wl, sel[i])), __builtin_bit_cast(rl_u16x2, cc[i]), 0u, false) >> 4;
    };
    int rowA = -1, rowB = -1; // the source rows (tile-relative) whose horizontal results gA / gB hold
    uint32_t gA[4] = {0, 0, 0, 0}, gB[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        if (dy0 + r >= dh) break;
        const OrbxTap ty = ytap[dy0 + r];
        const int r0 = min(max(ty.ofs, 0), sh - 1) - r_lo, r1 = min(max(ty.ofs + 1, 0), sh - 1) - r_lo;
        if (r0 != rowA) {
            if (r0 == rowB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gA[i] = gB[i];
            } else {
                hrow(r0, gA);
            }
            rowA = r0;
        }
        if (r1 != rowB) {
            if (r1 == rowA) { // (the clamped last row: both taps on one source row)
#pragma unroll
                for (int i = 0; i < 4; ++i) gB[i] = gA[i];
            } else {
                hrow(r1, gB);
            }
            rowB = r1;
        }
        // vertical step: ((b0 * g0 >> 16) + (b1 * g1 >> 16) + 2) >> 2.  The "+ 2" rides on the first product as 2 << 16, the two
        // ">> 16" are the word selects of one SDWA addition, and the four ">> 2" are two packed 16-bit shifts.
        uint32_t sm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t p0 = __umul24((uint32_t)ty.c0, gA[i]) + 0x20000u, p1 = __umul24((uint32_t)ty.c1, gB[i]);
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(sm[i]) : "v"(p0), "v"(p1));
        }
        const rl_u16x2 lo = __builtin_bit_cast(rl_u16x2, __builtin_amdgcn_perm(sm[1], sm[0], 0x05040100u)) >> (unsigned short)2;
        const rl_u16x2 hi = __builtin_bit_cast(rl_u16x2, __builtin_amdgcn_perm(sm[3], sm[2], 0x05040100u)) >> (unsigned short)2;
        const uint32_t q = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
        *reinterpret_cast<uint32_t *>(D + (size_t)(dy0 + r) * dst_pitch + dx0) = q; // rows of the arena are 64-byte aligned and padded
    }
}
// whether every tile's source patch fits k_resize_lds's LDS array (host tap tables of the step)
bool orbx_resize_lds_fits(const OrbxTap *xtap, const OrbxTap *ytap, int sw, int sh, int dw, int dh)
{
    for (int X0 = 0; X0 < dw; X0 += 256) {
        const int Xl = std::min(X0 + 255, dw - 1), colbase = xtap[X0].ofs & ~15;
        if (xtap[X0].ofs < 0 || ((std::min(xtap[Xl].ofs + 1, sw - 1) - colbase) >> 4) + 1 > RL_NC) return false;
    }
    for (int Y0 = 0; Y0 < dh; Y0 += RL_WAVES * RS_ROWS) {
        const int Yl = std::min(Y0 + RL_WAVES * RS_ROWS - 1, dh - 1);
        const int r_lo = std::min(std::max(ytap[Y0].ofs, 0), sh - 1);
        if (std::min(std::max(ytap[Yl].ofs + 1, 0), sh - 1) - r_lo + 1 > RL_NR) return false;
    }
    return sw >= 16;
}
void orbx_launch_resize_lds(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh, int last_row_bytes,
                            uint8_t *dst, size_t dst_fs, int dst_pitch, int dw, int dh, const OrbxTap *xtap, const OrbxTap *ytap,
                            int n_frames, int *zero_counts)
{
    const int gx = (dw + 255) / 256, gy = (dh + RL_WAVES * RS_ROWS - 1) / (RL_WAVES * RS_ROWS);
    hipLaunchKernelGGL(k_resize_lds, dim3(orbx_xcd_grid(gx * gy, n_frames)), dim3(64, RL_WAVES), dev_pad(k_resize_lds, 0, 0), s, src, src_fs, src_pitch, sw, sh,
                       last_row_bytes, dst, dst_fs, dst_pitch, dw, dh, xtap, ytap, gx, gy, n_frames, zero_counts);
}

// ---------------------------------------------------------------------------------------------
// Two pyramid levels per launch.  Level l+2 is resampled from level l+1's ROUNDED pixels (ComputePyramid resizes from the
// level before, :565), so the two steps cannot be merged arithmetically -- but a workgroup that owns a 64 x 16 tile of
// level l+2 can compute the (<= 24 x 96) patch of level l+1 it needs from level l into LDS, write the part of that patch
// it OWNS to memory, and resample its tile from LDS: level l+1 is written once and never read back, and the pyramid
// takes four dependent launches instead of seven (what a single-frame call waits for).
// Ownership: tile (bx, by) owns the level-(l+1) columns [cb(bx), cb(bx+1)) and rows [rb(by), rb(by+1)), cb = the first
// source column of the tile's first pixel rounded down to 4 (dword stores), rb = the first source row of its first row;
// the last tiles own up to the padded width / the height.  A patch always covers what its tile owns (scale < 2: the next
// tile's first source column is at most two beyond this tile's last one), and about an eighth of level l+1 is computed
// twice.  Same arithmetic as k_resize, pixel for pixel.
// ---------------------------------------------------------------------------------------------
#define R2_TW 64
#define R2_TH 16
#define R2_RW 104 // LDS pitch of the level-(l+1) patch: 4 * groups computed + 8 spare bytes, a multiple of 4
#define R2_RH 24
struct Resize2Args { // level l (source), l+1 and l+2
    const uint8_t *src; size_t src_fs; int src_pitch, sw, sh;
    uint8_t *d1; size_t d1_fs; int d1_pitch, w1, h1; const OrbxTap *xtap1, *ytap1;
    uint8_t *d2; size_t d2_fs; int d2_pitch, w2, h2; const OrbxTap *xtap2, *ytap2;
    int gx, gy, n_frames; int *zero_counts;
};
__global__ __launch_bounds__(256) void k_resize2(Resize2Args a)
{
    __shared__ __align__(16) uint8_t patch[R2_RH * R2_RW];
    int frame, blk;
    if (!xcd_remap(a.gx * a.gy, a.n_frames, &frame, &blk)) return;
    const int tid = threadIdx.x;
    if (a.zero_counts && blk == 0 && tid < ORBX_MAX_LEVELS) a.zero_counts[frame * ORBX_MAX_LEVELS + tid] = 0;
    const int by = blk / a.gx, bx = blk - by * a.gx;
    const int X0 = bx * R2_TW, Y0 = by * R2_TH, X1 = min(X0 + R2_TW, a.w2), Y1 = min(Y0 + R2_TH, a.h2);
    // the level-(l+1) patch: columns [cb, cend), rows [rb, rend); owned: columns below cn, rows below rn
    const int cb = a.xtap2[X0].ofs & ~3;
    const int cn = bx + 1 < a.gx ? (a.xtap2[X0 + R2_TW].ofs & ~3) : ((a.w1 + 3) & ~3);
    const int cend = max(min(a.xtap2[X1 - 1].ofs + 1, a.w1 - 1) + 1, cn);
    const int rb = max(a.ytap2[Y0].ofs, 0);
    const int rn = by + 1 < a.gy ? max(a.ytap2[Y0 + R2_TH].ofs, 0) : a.h1;
    const int rend = max(min(a.ytap2[Y1 - 1].ofs + 1, a.h1 - 1) + 1, rn);
    const int ncol4 = (cend - cb + 3) >> 2, nrow = rend - rb; // the host checked: 4 ncol4 + 8 <= R2_RW, nrow <= R2_RH
    const uint8_t *S = a.src + (size_t)frame * a.src_fs;
    uint8_t *D1 = a.d1 + (size_t)frame * a.d1_fs, *D2 = a.d2 + (size_t)frame * a.d2_fs;
    // ---- level l -> the patch of level l+1 (item = 4 pixels of one row)
    for (int it = tid; it < ncol4 * nrow; it += 256) {
        const int ry = it / ncol4, g4 = it - ry * ncol4;
        const int y1 = rb + ry, x1 = cb + 4 * g4;
        const OrbxTap ty = a.ytap1[min(y1, a.h1 - 1)];
        const int sy0 = min(max(ty.ofs, 0), a.sh - 1), sy1 = min(max(ty.ofs + 1, 0), a.sh - 1);
        const uint4 t01 = reinterpret_cast<const uint4 *>(a.xtap1 + x1)[0], t23 = reinterpret_cast<const uint4 *>(a.xtap1 + x1)[1];
        const int ofs[4] = {(int)t01.x, (int)t01.z, (int)t23.x, (int)t23.z};
        const uint32_t cc[4] = {t01.y, t01.w, t23.y, t23.w};
        const int sx0 = ofs[0], sxl = min(sx0, a.sw - 8); // a window that would pass the end of the row: k_resize's shift
        const uint32_t sft = (uint32_t)(sx0 - sxl) * 8u;
        unsigned long long w0 = reinterpret_cast<const UnalignedU64 *>(S + (size_t)sy0 * a.src_pitch + sxl)->v;
        unsigned long long w1 = reinterpret_cast<const UnalignedU64 *>(S + (size_t)sy1 * a.src_pitch + sxl)->v;
        w0 >>= sft; w1 >>= sft;
        uint32_t sel[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sel[i] = (uint32_t)(ofs[i] - sx0) * 0x00010001u + 0x0c010c00u;
        const uint32_t q = resize_quad((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32), sel, cc, ty.c0, ty.c1);
        *reinterpret_cast<uint32_t *>(&patch[ry * R2_RW + 4 * g4]) = q;
        if (y1 < rn && x1 < cn) *reinterpret_cast<uint32_t *>(D1 + (size_t)y1 * a.d1_pitch + x1) = q; // rows are padded to 64 bytes
    }
    __syncthreads();
    // ---- the patch -> this tile of level l+2 (thread = 4 pixels of one row)
    const int X = X0 + 4 * (tid & 15), Y = Y0 + (tid >> 4);
    if (X >= a.w2 || Y >= a.h2) return;
    const OrbxTap ty = a.ytap2[Y];
    const int r0 = min(max(ty.ofs, 0), a.h1 - 1) - rb, r1 = min(max(ty.ofs + 1, 0), a.h1 - 1) - rb;
    const uint4 t01 = reinterpret_cast<const uint4 *>(a.xtap2 + X)[0], t23 = reinterpret_cast<const uint4 *>(a.xtap2 + X)[1];
    const int ofs[4] = {(int)t01.x, (int)t01.z, (int)t23.x, (int)t23.z};
    const uint32_t cc[4] = {t01.y, t01.w, t23.y, t23.w};
    const int off = ofs[0] - cb;
    const uint32_t sh8 = (uint32_t)(off & 3) * 8u;
    const uint32_t *p0 = reinterpret_cast<const uint32_t *>(&patch[r0 * R2_RW + (off & ~3)]);
    const uint32_t *p1 = reinterpret_cast<const uint32_t *>(&patch[r1 * R2_RW + (off & ~3)]);
    const uint32_t a0 = p0[0], a1 = p0[1], a2 = p0[2], c0 = p1[0], c1 = p1[1], c2 = p1[2];
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sel[i] = (uint32_t)(ofs[i] - ofs[0]) * 0x00010001u + 0x0c010c00u;
    const uint32_t q = resize_quad(__builtin_amdgcn_alignbit(a1, a0, sh8), __builtin_amdgcn_alignbit(a2, a1, sh8),
                                   __builtin_amdgcn_alignbit(c1, c0, sh8), __builtin_amdgcn_alignbit(c2, c1, sh8), sel, cc, ty.c0, ty.c1);
    *reinterpret_cast<uint32_t *>(D2 + (size_t)Y * a.d2_pitch + X) = q;
}
// whether every tile's patch of level l+1 fits k_resize2's LDS array (host tap tables of levels l+1 -> l+2)
bool orbx_resize2_fits(const OrbxTap *xtap2, const OrbxTap *ytap2, int w1, int h1, int w2, int h2)
{
    const int gx = (w2 + R2_TW - 1) / R2_TW, gy = (h2 + R2_TH - 1) / R2_TH;
    for (int bx = 0; bx < gx; ++bx) {
        const int X0 = bx * R2_TW, X1 = std::min(X0 + R2_TW, w2);
        const int cb = xtap2[X0].ofs & ~3, cn = bx + 1 < gx ? (xtap2[X0 + R2_TW].ofs & ~3) : ((w1 + 3) & ~3);
        const int cend = std::max(std::min(xtap2[X1 - 1].ofs + 1, w1 - 1) + 1, cn);
        if (cn < cb || 4 * ((cend - cb + 3) >> 2) + 8 > R2_RW) return false;
    }
    for (int by = 0; by < gy; ++by) {
        const int Y0 = by * R2_TH, Y1 = std::min(Y0 + R2_TH, h2);
        const int rb = std::max(ytap2[Y0].ofs, 0), rn = by + 1 < gy ? std::max(ytap2[Y0 + R2_TH].ofs, 0) : h1;
        const int rend = std::max(std::min(ytap2[Y1 - 1].ofs + 1, h1 - 1) + 1, rn);
        if (rn < rb || rend - rb > R2_RH) return false;
    }
    return true;
}
void orbx_launch_resize2(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh, uint8_t *d1, size_t d1_fs,
                         int d1_pitch, int w1, int h1, const OrbxTap *xtap1, const OrbxTap *ytap1, uint8_t *d2, size_t d2_fs,
                         int d2_pitch, int w2, int h2, const OrbxTap *xtap2, const OrbxTap *ytap2, int n_frames, int *zero_counts)
{
    Resize2Args a;
    a.src = src; a.src_fs = src_fs; a.src_pitch = src_pitch; a.sw = sw; a.sh = sh;
    a.d1 = d1; a.d1_fs = d1_fs; a.d1_pitch = d1_pitch; a.w1 = w1; a.h1 = h1; a.xtap1 = xtap1; a.ytap1 = ytap1;
    a.d2 = d2; a.d2_fs = d2_fs; a.d2_pitch = d2_pitch; a.w2 = w2; a.h2 = h2; a.xtap2 = xtap2; a.ytap2 = ytap2;
    a.gx = (w2 + R2_TW - 1) / R2_TW; a.gy = (h2 + R2_TH - 1) / R2_TH; a.n_frames = n_frames; a.zero_counts = zero_counts;
    hipLaunchKernelGGL(k_resize2, dim3(orbx_xcd_grid(a.gx * a.gy, n_frames)), dim3(256), 0, s, a);
}

void orbx_launch_resize(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh,
                        uint8_t *dst, size_t dst_fs, int dst_pitch, int dw, int dh, const OrbxTap *xtap,
                        const OrbxTap *ytap, int n_frames, int *zero_counts)
{
    const int gx = (dw + 255) / 256, gy = (dh + 4 * RS_ROWS - 1) / (4 * RS_ROWS);
    hipLaunchKernelGGL(k_resize, dim3(orbx_xcd_grid(gx * gy, n_frames)), dim3(64, 4), 0, s, src, src_fs, src_pitch, sw, sh, dst,
                       dst_fs, dst_pitch, dw, dh, xtap, ytap, gx, gy, n_frames, zero_counts);
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16: threshold-free corner strength S = max(max_arc min(v-p), max_arc min(p-v)) - 1.
// "corner at t" <=> S >= t, and OpenCV's stored score of a corner is S for every t,
// so one strength tile answers both the ini and the min threshold of a cell.
// ---------------------------------------------------------------------------------------------

struct FastSrc {
    const uint8_t *base[ORBX_MAX_LEVELS];
    size_t frame_stride[ORBX_MAX_LEVELS];
    int pitch[ORBX_MAX_LEVELS];
};
struct __attribute__((packed, aligned(1))) UnalignedU32 { uint32_t v; };
typedef short s16x2 __attribute__((ext_vector_type(2)));

// an arc of 9 out of 16 holds one pixel of every opposite pair: exact necessary condition for S >= thr
__device__ __forceinline__ bool fast_pairs(const int d[16], int thr)
{
    int minhi = 1 << 20, maxlo = -(1 << 20);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        minhi = min(minhi, max(d[k], d[k + 8]));
        maxlo = max(maxlo, min(d[k], d[k + 8]));
    }
    return minhi > thr || maxlo < -thr;
}
__device__ __forceinline__ int fast_score(const int d[16])
{
    int m3[16], M3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        m3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        M3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    }
    int best_min = -(1 << 20), best_max = 1 << 20;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        best_min = max(best_min, min(min(m3[k], m3[(k + 3) & 15]), m3[(k + 6) & 15]));
        best_max = min(best_max, max(max(M3[k], M3[(k + 3) & 15]), M3[(k + 6) & 15]));
    }
    return max(best_min, -best_max) - 1;
}

// byte pair (i, j) of the 8 bytes {hi:lo} zero-extended into the two 16-bit halves
#define PERM_SEL(i, j) ((uint32_t)(i) | (0x0cu << 8) | ((uint32_t)(j) << 16) | (0x0cu << 24))
__device__ __forceinline__ s16x2 pk_bytes(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint32_t r = __builtin_amdgcn_perm(hi, lo, sel);
    return __builtin_bit_cast(s16x2, r);
}

// ---------------------------------------------------------------------------------------------
// One WAVE per cell: a 64-thread workgroup owns one 30x30 cell end to end, so no phase waits for
// other waves and ~25 independent waves per CU hide each other's LDS / L2 latency (a 256-thread
// "strip of 8 cells" version of the same pipeline was 1.8x slower).  One launch covers all levels.
//   1. the 36x40 tile goes L2 -> LDS with 64-bit loads (rows overlap with the neighbour cells');
//   2. every pixel takes the 4-point compass test (any 9-arc holds two adjacent compass points), two
//      pixels per instruction in packed 16-bit lanes; survivors are compacted into an LDS queue
//      (popcount + wave prefix);
//   3. the queue takes the exact 8-pair test (ballot + mbcnt compaction in place), its survivors the
//      exact strength, written to a u8 score map;
//   4. strict 3x3 NMS inside the cell from the score map, keepers appended to a list.
// Everything runs at the ini threshold first; a cell left without a keeper is redone at the min
// threshold (reference :604-607).  Both answers are exact for their cells.
// ---------------------------------------------------------------------------------------------
#define FC_TP 40 // tile pitch; tile column tc <-> level x = x0 - 4 + tc, region columns tc in [4, 4 + cw)
#define FC_SP 40 // score-map pitch; score column = tc - 3, so the cell occupies columns 1..cw
struct __attribute__((aligned(8))) FastCell { // 8-byte aligned: one scalar load per cell
    uint16_t level, cy, cx, pad;
};
__device__ __forceinline__ void fastc_load_ring(const uint8_t *t, int d[16])
{
    const int v = t[0];
    d[0] = v - t[3 * FC_TP];      d[1] = v - t[3 * FC_TP + 1];  d[2] = v - t[2 * FC_TP + 2];
    d[3] = v - t[FC_TP + 3];      d[4] = v - t[3];              d[5] = v - t[-FC_TP + 3];
    d[6] = v - t[-2 * FC_TP + 2]; d[7] = v - t[-3 * FC_TP + 1]; d[8] = v - t[-3 * FC_TP];
    d[9] = v - t[-3 * FC_TP - 1]; d[10] = v - t[-2 * FC_TP - 2]; d[11] = v - t[-FC_TP - 3];
    d[12] = v - t[-3];            d[13] = v - t[FC_TP - 3];     d[14] = v - t[2 * FC_TP - 2];
    d[15] = v - t[3 * FC_TP - 1];
}
struct __attribute__((packed, aligned(1))) UnalignedU64b { unsigned long long v; };

// A wave's LDS instructions execute in issue order: between a producer and a consumer stage of ONE wave the only thing
// needed is that the compiler keeps that order.
#define FS_WAVE_ORDER() asm volatile("" ::: "memory")
#ifdef OCT_PROF
// development build (make prof): start / end time stamp (100 MHz) and the CU of every cell's workgroup of frame 0 (tools/fast_cell_times.py)
#define FC_PROF_CELLS 16384
__device__ unsigned long long g_fc_times[3 * FC_PROF_CELLS];
unsigned long long *orbx_dev_fast_cell_times_symbol() { unsigned long long *p = nullptr; (void)hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_fc_times)); return p; }
#endif
// CPW cells per workgroup, one wave each.  The waves share nothing while they work -- every wave has its own tile, score map
// and queues, and orders its own LDS traffic (a wave's LDS instructions execute in issue order) -- and meet once, at the end:
// the keepers of all CPW cells reserve their place in the level's candidate list with ONE atomic.  The reservation is a
// returning atomic on one address per (frame, level), and such atomics are served one at a time, about 8 ns each: with one
// per cell the 2304 cells of a 1920x1080 level 0 stood in line for 18 us after 12 us of work, the 1600 cells of level 1
// for 13 (per-cell time stamps, tools/fast_cell_times.py).
template <int CPW>
__global__ __launch_bounds__(64 * CPW) void k_fast_cells_wave(FastSrc src, const OrbxLevels *__restrict__ levels,
                                                              const FastCell *__restrict__ cells, u64 *__restrict__ cand,
                                                              size_t cand_fs, int *__restrict__ cand_count, int n_cells,
                                                              int n_frames)
{
    static_assert(CPW >= 1 && CPW <= 16, "at most 16 waves per workgroup");
    __shared__ __align__(16) uint8_t s_tile[CPW][36 * FC_TP];
    __shared__ __align__(16) uint8_t s_score[CPW][32 * FC_SP];
    // queue2 aliases queue1: the 8-pair test compacts in place (a batch is read before it is written, and it is
    // written at positions <= the ones just read)
    __shared__ uint16_t s_queue1[CPW][ORBX_CELL * ORBX_CELL];
    __shared__ uint32_t s_keepers[CPW][15 * 15];
    __shared__ int s_nkeep[CPW], s_level[CPW], s_base;

    int frame, group;
    if (!xcd_remap((n_cells + CPW - 1) / CPW, n_frames, &frame, &group)) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cell_id = group * CPW + wid;
    const bool live = cell_id < n_cells; // (a surplus wave of the last group only takes part in the reservation)
#ifdef OCT_PROF
    const unsigned long long fc_t0 = wall_clock64();
#endif
    const FastCell cl = cells[min(cell_id, n_cells - 1)];
    uint8_t *tile = s_tile[wid], *score = s_score[wid];
    uint16_t *queue1 = s_queue1[wid], *queue2 = queue1;
    uint32_t *keepers = s_keepers[wid];
    const int level = cl.level;
    const OrbxLevel &lv = levels->lv[level];
    if (lane == 0) { s_nkeep[wid] = 0; s_level[wid] = live ? level : -1; }
    if (live) {
    const int x0 = ORBX_EDGE + cl.cx * ORBX_CELL, y0 = ORBX_EDGE + cl.cy * ORBX_CELL;
    const int cw = min(ORBX_CELL, lv.w - ORBX_EDGE - x0), ch = min(ORBX_CELL, lv.h - ORBX_EDGE - y0);
    const int th = ch + 6;
    const int pitch = src.pitch[level];
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level] + (size_t)(y0 - 3) * pitch + (x0 - 4);

    for (int i = lane; i < 32 * FC_SP / 16; i += 64) reinterpret_cast<uint4 *>(score)[i] = make_uint4(0, 0, 0, 0);
    {
        // The 36 x 40-byte tile as 5 x th eight-byte items, three per lane, all requested before the first is stored (the
        // index is clamped instead of branching: a surplus lane re-reads the last item).  Items of the last cell column may
        // reach past the end of the image row -- into the row's padding or the next row, never past the frame: the tile's
        // last row is at most h - 17 -- and those bytes are never looked at: the compass test masks the columns beyond the
        // cell and a ring of a cell pixel ends at column w - 17.
        const int n_items = 5 * th;
        unsigned long long v[3];
        int dst[3];
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int i = min(lane + 64 * it, n_items - 1);
            const int ty = i / 5, tx = (i - 5 * ty) * 8;
            dst[it] = ty * FC_TP + tx;
            v[it] = reinterpret_cast<const UnalignedU64b *>(S + (size_t)ty * pitch + tx)->v;
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) *reinterpret_cast<unsigned long long *>(&tile[dst[it]]) = v[it];
    }
    FS_WAVE_ORDER();
    int thr = levels->ini_th;
    for (int pass = 0; pass < 2; ++pass) {
        // ---- compass test: item = (row, run of 4 tile columns 4g..4g+3), g = 1..8.
        // A 9-arc contains one point of every opposite pair, so a corner at threshold T needs
        //   V - max(min(S,N), min(E,W)) > T   (both pairs hold a darker point)   or
        //   min(max(S,N), max(E,W)) - V > T   (both pairs hold a brighter point).
        // Two pixels per instruction in packed i16; m - (T+1) has its sign bit set iff the pixel fails, and the sign
        // bytes of the item's four pixels are gathered with one v_perm.  acc collects them: bit 8j + 4 + it set =
        // pixel j of iteration `it` is NOT a candidate.
        const s16x2 T1 = {(short)(thr + 1), (short)(thr + 1)};
        uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 64 + lane;
            const int r = i >> 3, g = (i & 7) + 1;
            uint32_t w = 0x80808080u; // rows below the cell: no candidates
            if (r < ch) {
                const uint8_t *row = &tile[(r + 3) * FC_TP + 4 * g];
                const uint32_t cm = *reinterpret_cast<const uint32_t *>(row - 4);
                const uint32_t c0 = *reinterpret_cast<const uint32_t *>(row);
                const uint32_t cp = *reinterpret_cast<const uint32_t *>(row + 4);
                const uint32_t up = *reinterpret_cast<const uint32_t *>(row - 3 * FC_TP);
                const uint32_t dn = *reinterpret_cast<const uint32_t *>(row + 3 * FC_TP);
                uint32_t x[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const s16x2 V = pk_bytes(0, c0, h ? PERM_SEL(2, 3) : PERM_SEL(0, 1));
                    const s16x2 E = pk_bytes(cp, c0, h ? PERM_SEL(5, 6) : PERM_SEL(3, 4));
                    const s16x2 Wv = pk_bytes(c0, cm, h ? PERM_SEL(3, 4) : PERM_SEL(1, 2));
                    const s16x2 S = pk_bytes(0, dn, h ? PERM_SEL(2, 3) : PERM_SEL(0, 1));
                    const s16x2 N = pk_bytes(0, up, h ? PERM_SEL(2, 3) : PERM_SEL(0, 1));
                    const s16x2 A = __builtin_elementwise_max(__builtin_elementwise_min(S, N), __builtin_elementwise_min(E, Wv));
                    const s16x2 B = __builtin_elementwise_min(__builtin_elementwise_max(S, N), __builtin_elementwise_max(E, Wv));
                    const s16x2 m = __builtin_elementwise_max(V - A, B - V) - T1;
                    x[h] = __builtin_bit_cast(uint32_t, m);
                }
                w = __builtin_amdgcn_perm(x[1], x[0], 0x07050301u); // high bytes of the four i16
            }
            acc = (w & 0x80808080u) | ((acc >> 1) & 0x7F7F7F7Fu);
        }
        uint32_t mask;
        {
            const int g = (lane & 7) + 1;
            const int nvalid = min(max(cw + 4 - 4 * g, 0), 4); // region columns end at tc = 4 + cw
            const uint32_t vmask = nvalid >= 4 ? 0xF0F0F0F0u : (0xF0F0F0F0u & ((1u << (8 * nvalid)) - 1u));
            mask = ~acc & vmask;
        }
        int q1n;
        {
            const int cnt = __popc(mask);
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            q1n = __shfl(incl, 63);
            int slot = incl - cnt;
            uint32_t m = mask;
            while (m) {
                const int b = __ffs(m) - 1;
                m &= m - 1;
                const int i = ((b & 7) - 4) * 64 + lane; // bit 8j + 4 + it
                queue1[slot++] = (uint16_t)(((i >> 3) << 8) | (4 * ((i & 7) + 1) + (b >> 3))); // (row, tc)
            }
        }
        FS_WAVE_ORDER();
        // ---- exact 8-pair test -> queue 2
        int q2n = 0;
        for (int i0 = 0; i0 < q1n; i0 += 64) {
            const int i = i0 + lane;
            bool ok = false;
            int e = 0;
            if (i < q1n) {
                e = queue1[i];
                int d[16];
                fastc_load_ring(&tile[((e >> 8) + 3) * FC_TP + (e & 255)], d);
                ok = fast_pairs(d, thr);
            }
            const u64 mk = __ballot(ok);
            const int pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            FS_WAVE_ORDER();
            if (ok) queue2[q2n + pre] = (uint16_t)e;
            q2n += (int)__popcll(mk);
        }
        FS_WAVE_ORDER();
        // ---- exact strength
        for (int i = lane; i < q2n; i += 64) {
            const int e = queue2[i], r = e >> 8, tc = e & 255;
            int d[16];
            fastc_load_ring(&tile[(r + 3) * FC_TP + tc], d);
            const int sc = fast_score(d);
            if (sc >= thr) score[(r + 1) * FC_SP + tc - 3] = (uint8_t)sc;
        }
        FS_WAVE_ORDER();
        // ---- strict 3x3 NMS (everything outside the cell is 0 in the score map)
        for (int i = lane; i < q2n; i += 64) {
            const int e = queue2[i], r = e >> 8, tc = e & 255;
            const uint8_t *sp = &score[(r + 1) * FC_SP + tc - 3];
            const int sc = sp[0];
            if (sc > 0 && sc > sp[-1] && sc > sp[1] && sc > sp[-FC_SP - 1] && sc > sp[-FC_SP] && sc > sp[-FC_SP + 1] &&
                sc > sp[FC_SP - 1] && sc > sp[FC_SP] && sc > sp[FC_SP + 1])
                keepers[atomicAdd(&s_nkeep[wid], 1)] = (uint32_t)r | ((uint32_t)(tc - 4) << 8) | ((uint32_t)sc << 16);
        }
        FS_WAVE_ORDER();
        if (__builtin_amdgcn_readfirstlane(s_nkeep[wid]) > 0 || pass == 1) break;
        thr = levels->min_th; // reference :604-607: nothing at the ini threshold -> redo the cell at the min threshold
    }
    }
    FS_WAVE_ORDER();
    const int nk = __builtin_amdgcn_readfirstlane(s_nkeep[wid]);
#ifdef OCT_PROF
    unsigned long long fc_t1 = wall_clock64();
#endif
    // ---- one reservation for the cells of the workgroup (all of one level; a group that straddles two levels -- at most
    // one per level and frame -- lets every wave reserve for itself)
    int base;
    if (CPW > 1) {
        __syncthreads();
        int before = 0, total = 0;
        bool same = true;
        const int lvl0 = s_level[0];
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int n = s_nkeep[k];
            if (k < wid) before += n;
            total += n;
            same = same && (s_level[k] == lvl0 || s_level[k] < 0);
        }
        if (same) {
            if (total == 0) return;
            if (tid == 0) s_base = atomicAdd(&cand_count[frame * ORBX_MAX_LEVELS + level], total);
            __syncthreads();
            base = s_base + before;
        } else {
            base = 0;
            if (lane == 0 && nk > 0) base = atomicAdd(&cand_count[frame * ORBX_MAX_LEVELS + level], nk);
            base = __builtin_amdgcn_readfirstlane(base);
        }
    } else {
        base = 0;
        if (lane == 0 && nk > 0) base = atomicAdd(&cand_count[frame * ORBX_MAX_LEVELS + level], nk);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    u64 *out = cand + (size_t)frame * cand_fs + lv.cand_off + base;
    for (int i = lane; i < nk; i += 64) {
        const uint32_t e = keepers[i];
        const uint32_t x = cl.cx * ORBX_CELL + ((e >> 8) & 255), y = cl.cy * ORBX_CELL + (e & 255);
        out[i] = (u64)(x | (y << 16)) | ((u64)(e >> 16) << 32);
    }
#ifdef OCT_PROF
    if (frame == 0 && lane == 0 && live) {
        const int slot = (cl.level * 4096 + cl.cy * 64 + cl.cx) % FC_PROF_CELLS; // (level, cy, cx) is the key
        g_fc_times[3 * slot] = fc_t0;
        g_fc_times[3 * slot + 1] = wall_clock64();
        g_fc_times[3 * slot + 2] = fc_t1; // the cell's work done, before the reservation
    }
#endif
}


// ---------------------------------------------------------------------------------------------
// FAST, second formulation: one WAVE per STRIP of up to FS_K horizontally adjacent cells, all stages at full lanes.
//   1. the strip's (ch + 6) x (30 K + 7) tile goes L2 -> LDS;
//   2. dense compass test on 6-bit pixels, FOUR pixels per 32-bit operation (SWAR): with p6 = p >> 2 and
//      t6 = ceil((T - 2) / 4), V - p > T implies V6 - p6 >= t6, so bit 7 of the byte (V6 + 128 - t6) - p6 is a
//      superset flag of "darker"; no byte ever borrows (6-bit operands), so plain v_sub / v_add / v_or / v_and
//      -- the cheap VALU class on gfx950 (tools/microbench/valu_ops2.hip) -- do the work.  Items (row, 4-pixel
//      group) with a survivor are queued (one ballot per 256 pixels);
//   3. queued items take the full 16-point test in the same SWAR form: 32 flags, then "9 contiguous of 16" as
//      2 -> 4 -> 8 -> 9 AND-doubling on the even ring positions (47 logic operations per polarity for four pixels).
//      What passes is a superset of the corners of about 1.15 x their number; those pixels are queued;
//   4. queued pixels get the exact strength from the raw tile (17 byte gathers, min3/max3 windows), 64 at a time;
//      corners (S >= T) go to a score map and a list;
//   5. once the pipeline has drained the score map is written over the tile (one empty column between cells, so a
//      neighbour in the next cell reads 0), strict 3x3 NMS from it, ONE returning atomic per strip, keepers to global
//      memory.
// Stages 2-4 run as a rolling pipeline (a stage fires as soon as 64 entries wait), so their lanes are full whatever
// the corner density.  A cell left without a keeper is redone at the min threshold (reference :604-607).
// ---------------------------------------------------------------------------------------------
#ifndef FS_K
#define FS_K 4                       // cells per strip (at most)
#endif
static_assert(FS_K >= 1 && FS_K <= 4, "a strip's tile row is 16 eight-byte items and its compass pass 32 lanes wide: at most 4 cells");
static_assert(4 * ((FS_K * ORBX_CELL + 3) / 4) + 3 < 128, "a pixel entry keeps the tile column in 7 bits (bit 7 is the polarity)");
#define FS_W (FS_K * ORBX_CELL)      // region columns of a full strip
#define FS_G ((FS_W + 3) / 4)        // 4-pixel groups of a full strip
#define FS_LG (FS_G <= 8 ? 8 : FS_G <= 16 ? 16 : 32) // lanes per tile row in the dense stage (a power of two >= FS_G)
#define FS_R (64 / FS_LG)            // rows per dense iteration
#define FS_NQ ((FS_W + 7 + 7) / 8)   // 8-byte items per tile row (<= 16)
#define FS_TP (FS_NQ * 8 + 8)        // tile pitch: 4 + FS_W + 3 bytes, +8 so that consecutive rows start 2 banks apart
#define FS_TROWS (36 + FS_R)         // the dense stage may look one iteration past the last cell row
#define FS_SP ((FS_W + FS_K + 2 + 3) / 4 * 4) // score-map pitch; score column = region column + cell index + 1
#define FS_IQ 128                    // item ring (at most 63 left over + 64 new)
#define FS_CQ 512                    // pixel ring (at most 63 left over + 256 new)
#define FS_LIST (128 * FS_K)         // corners of a pass (the NMS works from this list); more -> narrow passes
#define FS_NARROW (FS_LIST / 30 - 2) // columns of a narrow pass: (FS_NARROW + 2) x 30 pixels always fit the list
struct __attribute__((aligned(8))) FastStrip { // 8-byte aligned: one scalar load per strip
    uint16_t level, cy, cx0, ncells;
};
// (FS_WAVE_ORDER, above: the queues are written and read by the one wave of the workgroup)
// the same barrier, leaving a named comment in the ISA: tools/isa_mix.py counts the instructions between "name_begin" and "name_end"
#define FS_MARK(name) asm volatile("; FSM " name ::: "memory")

// three-input logic in one instruction (v_bitop3_b32: result bit = table[(a << 2) | (b << 1) | c], cheap issue class)
#define B3_AND3 0x80      // a & b & c
#define B3_A_AND_BORC 0xE0 // a & (b | c)
#define B3_A_OR_BANDC 0xF8 // a | (b & c)
// bit 7 of every byte set <=> at least 9 contiguous (cyclic) of the 16 flags x[k] have bit 7 set in that byte.
// A 9-run holds four whole aligned pairs (2s, 2s+1) .. (2s+6, 2s+7) and the flag before or after them: 32 three-input
// operations (the 2 -> 4 -> 8 doubling on the pairs took 40).
__device__ __forceinline__ uint32_t swar_arc9(const uint32_t x[16])
{
    uint32_t a[8], any = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = x[2 * s] & x[2 * s + 1];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t b = __builtin_amdgcn_bitop3_b32(a[s], a[(s + 1) & 7], a[(s + 2) & 7], B3_AND3);
        const uint32_t d = __builtin_amdgcn_bitop3_b32(a[(s + 3) & 7], x[(2 * s + 8) & 15], x[(2 * s + 15) & 15], B3_A_AND_BORC);
        any = s == 0 ? (b & d) : __builtin_amdgcn_bitop3_b32(any, b, d, B3_A_OR_BANDC);
    }
    return any;
}
// bytes 0..n-1 = 0xFF (n <= 0: none, n >= 4: all)
__device__ __forceinline__ uint32_t byte_prefix_mask(int n)
{
    return n <= 0 ? 0u : (n >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n)) - 1u));
}
__device__ __forceinline__ int wave_rank(u64 mk)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
}
// base + rank: the lane count instructions add their third operand, so a queue position costs no separate addition
__device__ __forceinline__ int wave_rank_from(u64 mk, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, (uint32_t)base));
}

// ballot of "byte J of v is not zero" in ONE instruction (the byte select rides on the compare: SDWA), result in a scalar pair
template <int J> __device__ __forceinline__ u64 ballot_byte_nz(uint32_t v)
{
    u64 mk;
    if (J == 0) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(mk) : "v"(v), "s"(0));
    if (J == 1) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD" : "=s"(mk) : "v"(v), "s"(0));
    if (J == 2) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD" : "=s"(mk) : "v"(v), "s"(0));
    if (J == 3) asm volatile("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD" : "=s"(mk) : "v"(v), "s"(0));
    return mk;
}
// (byte J of a, zero-extended) | b in one instruction
template <int J> __device__ __forceinline__ uint32_t or_byte(uint32_t a, uint32_t b)
{
    uint32_t r;
    if (J == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (J == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (J == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    if (J == 3) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 16-bit LDS store by the lanes of `mk` only: the ballot IS the execution mask, so the predicate is not evaluated a second time
// (written as a branch the compiler compares again for the mask).  `addr` = byte address inside LDS.
__device__ __forceinline__ void lds_store_u16_masked(u64 mk, uint32_t addr, uint32_t val)
{
    u64 save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mk), "v"(addr), "v"(val) : "memory");
}
// (a << 1) + b in one instruction, b from a scalar register
__device__ __forceinline__ uint32_t shl1_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

#ifdef OCT_PROF
// development build only (make prof): how often a wave runs each stage of k_fast_strip, summed over the launch
// (tools/fast_mix.py weighs the kernel's instruction mix with these): strips, tile loads, compass steps, 16-point
// batches, strength batches, NMS list batches, passes, items queued, pixels queued
__device__ unsigned long long g_fs_prof[16];
#define FS_COUNT(i, n) do { if (lane == 0) atomicAdd(&g_fs_prof[i], (unsigned long long)(n)); } while (0)
unsigned long long *orbx_dev_fast_prof_symbol() { unsigned long long *p = nullptr; (void)hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_fs_prof)); return p; }
#else
#define FS_COUNT(i, n) do { } while (0)
#endif
__global__ __launch_bounds__(64) void k_fast_strip(FastSrc src, const OrbxLevels *__restrict__ levels,
                                                   const FastStrip *__restrict__ strips, u64 *__restrict__ cand,
                                                   size_t cand_fs, int *__restrict__ cand_count, int n_strips,
                                                   int n_frames)
{
    __shared__ __align__(16) uint8_t tile[FS_TROWS * FS_TP];
    // The score map of the NMS lives in the tile's memory: it is built from the pass's corner list once the pipeline has
    // drained and the pixels are no longer needed (a later pass reloads the tile).  Without a map of its own a wave needs
    // a third less LDS, which is what lets a strip span more cells at the same occupancy.
    static_assert(32 * FS_SP <= FS_TROWS * FS_TP, "the score map must fit into the tile");
    static_assert(FS_NARROW >= 2, "the corner list is too short for a narrow pass");
    uint8_t *const score = tile;
    __shared__ uint32_t iq[FS_IQ];
    __shared__ uint16_t cq[FS_CQ];
    __shared__ uint32_t list[FS_LIST];
    __shared__ int s_cellkeep[FS_K];

    int frame, strip_id;
    if (!xcd_remap(n_strips, n_frames, &frame, &strip_id)) return;
    const FastStrip st = strips[strip_id];
    const int lane = threadIdx.x;
    const int level = st.level;
    const OrbxLevel &lv = levels->lv[level];
    const int x0 = ORBX_EDGE + st.cx0 * ORBX_CELL, y0 = ORBX_EDGE + st.cy * ORBX_CELL;
    const int Ws = min(ORBX_CELL * (int)st.ncells, lv.w - ORBX_EDGE - x0); // region columns of this strip
    const int ch = min(ORBX_CELL, lv.h - ORBX_EDGE - y0);
    const int th = ch + 6;
    const int pitch = src.pitch[level];
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level] + (size_t)(y0 - 3) * pitch + (x0 - 4);

    if (lane < FS_K) s_cellkeep[lane] = 0;
    FS_COUNT(0, 1);
    auto load_tile = [&]() {
        FS_COUNT(1, 1);
        FS_MARK("tile_begin");
        // tile column tc <-> level x = x0 - 4 + tc; region columns are tc in [4, 4 + Ws); the ring of the last region
        // column ends at tc = Ws + 6.  Lane = (8-byte item tx, row ty mod 4); nine row groups, all requested before the
        // first is stored (row and item clamped instead of branching: surplus lanes repeat the last row / item).  The
        // bytes an item may hold beyond Ws + 6 are never looked at for a region pixel, and the loads stay inside the
        // frame (the tile's last row is at most h - 17 and its last byte at most w - 8 of that row).
        // (row clamped in offset space: min(row, th - 1) * pitch + tx = min(row * pitch + tx, (th - 1) * pitch + tx) -- an add and a
        // min per row group where a multiply per row group stood: nine quarter-rate 32-bit multiplies per strip)
        const int n_q = (Ws + 14) >> 3;
        const int tx = min(lane & 15, n_q - 1) * 8, ty0 = lane >> 4;
        unsigned long long v[9];
        int dst[9];
        const int src_max = (th - 1) * pitch + tx, dst_max = (th - 1) * FS_TP + tx;
        int so = ty0 * pitch + tx, dn = ty0 * FS_TP + tx;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int sk = min(so, src_max);
            dst[k] = min(dn, dst_max);
            v[k] = reinterpret_cast<const UnalignedU64b *>(S + sk)->v;
            so += 4 * pitch; dn += 4 * FS_TP;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            *reinterpret_cast<unsigned long long *>(&tile[dst[k]]) = v[k];
        }
        FS_MARK("tile_end");
    };
    __syncthreads();

    int iq_head = 0, iq_tail = 0, cq_head = 0, cq_tail = 0, list_n = 0;
    bool tile_ok = false; // the LDS tile holds the strip's pixels (a pass that found corners builds its score map over them)
    int thr = 0;
    uint32_t Kd = 0; // per byte 128 - t6
    bool both_possible = false; // t6 == 0: a pixel can pass the 6-bit arc test in both polarities

    // ---- stage 4: exact strength of `n` queued pixels (n <= 64)
    auto stage_score = [&](int n) {
        FS_COUNT(4, 1); FS_COUNT(8, n);
        FS_MARK("score_begin");
        const bool live = lane < n;
        const int e = cq[(cq_head + min(lane, n - 1)) & (FS_CQ - 1)];
        cq_head += n;
        // bit 7: the arc test passed the pixel as a BRIGHT corner candidate.  A corner at T >= 0 is dark or bright, never
        // both (two 9-arcs of a 16-ring share two pixels), and its strength is its own polarity's arc measure; with
        // every byte complemented the bright measure is the dark one, so one window pass serves either polarity.
        const int r = (e >> 8) & 31, tc = e & 127;
        int F = (e & 0x80) ? 255 : 0;
        asm("" : "+v"(F)); // kept as one vector register: folded into the 17 exclusive-ors it would make each a three-input operation with a scalar operand (slow class)
        const uint8_t *t = &tile[r * FS_TP + tc - 3]; // ring pixel (dx, dy) at t[(dy + 3) * FS_TP + dx + 3]
#define FS_PX(dx, dy) ((int)t[((dy) + 3) * FS_TP + (dx) + 3] ^ F)
        int p[16];
        p[0] = FS_PX(0, 3);    p[1] = FS_PX(1, 3);    p[2] = FS_PX(2, 2);    p[3] = FS_PX(3, 1);
        p[4] = FS_PX(3, 0);    p[5] = FS_PX(3, -1);   p[6] = FS_PX(2, -2);   p[7] = FS_PX(1, -3);
        p[8] = FS_PX(0, -3);   p[9] = FS_PX(-1, -3);  p[10] = FS_PX(-2, -2); p[11] = FS_PX(-3, -1);
        p[12] = FS_PX(-3, 0);  p[13] = FS_PX(-3, 1);  p[14] = FS_PX(-2, 2);  p[15] = FS_PX(-1, 3);
        const int v = FS_PX(0, 0);
#undef FS_PX
        // S + 1 = V - min_k max_{arc k} p for a dark corner, arcs of 9 as 3 x 3
        int hi3[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) hi3[k] = max(max(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
        int a = 1 << 20;
#pragma unroll
        for (int k = 0; k < 16; ++k) a = min(a, max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]));
        const int sc = v - a - 1;
        const bool corner = live && sc >= thr;
        const int c = tc - 4;
        const u64 mk = __ballot(corner);
        const int pos = wave_rank_from(mk, list_n);
        if (corner && pos < FS_LIST) list[pos] = (uint32_t)r | ((uint32_t)c << 8) | ((uint32_t)sc << 16);
        list_n += (int)__popcll(mk); // may exceed FS_LIST: the pass is then redone in narrow column ranges
        FS_MARK("score_end");
    };
    // ---- stage 3: the full 16-point test of `n` queued items on the 6-bit tile, four pixels per operation
    auto stage_arc = [&](int n) {
        FS_COUNT(3, 1); FS_COUNT(7, n);
        FS_MARK("arc_begin");
        const uint32_t e = iq[(iq_head + min(lane, n - 1)) & (FS_IQ - 1)];
        iq_head += n;
        const int r = e & 31, g = (e >> 8) & 31;
        const uint32_t mc = lane < n ? (e & 0x80808080u) : 0u; // the item's compass survivors
        const uint32_t *t = reinterpret_cast<const uint32_t *>(&tile[r * FS_TP + 4 * g]); // (row r - 3, column tc - 4)
        uint32_t cm[7], c0[7], cp[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) { cm[d] = t[d * (FS_TP / 4)]; c0[d] = t[d * (FS_TP / 4) + 1]; cp[d] = t[d * (FS_TP / 4) + 2]; }
        // ring pixel k of the four pixels = the row's bytes shifted by dx (right shifts take bytes from cp, left ones from
        // cm) and reduced to 6 bits: one v_alignbit by 8 dx + 2 bits and one mask
#define FS_SHR(d, n_) (__builtin_amdgcn_alignbit(cp[d], c0[d], 8 * (n_) + 2) & 0x3F3F3F3Fu)
#define FS_SHL(d, n_) (__builtin_amdgcn_alignbit(c0[d], cm[d], 8 * (4 - (n_)) + 2) & 0x3F3F3F3Fu)
#define FS_SH0(d) ((c0[d] >> 2) & 0x3F3F3F3Fu)
        uint32_t R[16];
        R[0] = FS_SH0(6);     R[1] = FS_SHR(6, 1);  R[2] = FS_SHR(5, 2);  R[3] = FS_SHR(4, 3);
        R[4] = FS_SHR(3, 3);  R[5] = FS_SHR(2, 3);  R[6] = FS_SHR(1, 2);  R[7] = FS_SHR(0, 1);
        R[8] = FS_SH0(0);     R[9] = FS_SHL(0, 1);  R[10] = FS_SHL(1, 2); R[11] = FS_SHL(2, 3);
        R[12] = FS_SHL(3, 3); R[13] = FS_SHL(4, 3); R[14] = FS_SHL(5, 2); R[15] = FS_SHL(6, 1);
        const uint32_t V6 = FS_SH0(3);
#undef FS_SHR
#undef FS_SHL
#undef FS_SH0
        const uint32_t Vd = V6 + Kd, Vb = Kd - V6;
        uint32_t x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = Vd - R[k];   // bit 7: darker (superset)
        const uint32_t md = swar_arc9(x) & mc; // a 9-arc holds two adjacent compass points: nothing outside mc can pass
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = Vb + R[k];   // bit 7: brighter (superset)
        const uint32_t mb = swar_arc9(x) & mc;
        const uint32_t m = md | mb;
        // pixel entry: row << 8 | tile column, bit 7 = candidate of a BRIGHT corner (the exact stage then evaluates that
        // polarity only).  With t6 >= 1 no pixel passes both; at thresholds 0..2 (t6 = 0) the dark entry is queued too.
        // Byte j of `mbj` is the entry's low byte less the column base: 0x80 for bright, + j.
        const uint32_t ent = (uint32_t)(r << 8) | (uint32_t)(4 * g + 4);
        const uint32_t mbj = mb | 0x03020100u;
        const uint32_t cq_base = lds_addr(cq);
#define FS_QUEUE_PIXEL(j, flags, tagsrc)                                                                         \
        {                                                                                                        \
            const u64 mk = ballot_byte_nz<j>(flags);                                                             \
            const uint32_t pos = (uint32_t)wave_rank_from(mk, cq_tail);                                          \
            lds_store_u16_masked(mk, shl1_add(pos & (FS_CQ - 1), cq_base), or_byte<j>(tagsrc, ent));                  \
            cq_tail += (int)__popcll(mk);                                                                        \
        }
        FS_QUEUE_PIXEL(0, m, mbj) FS_QUEUE_PIXEL(1, m, mbj) FS_QUEUE_PIXEL(2, m, mbj) FS_QUEUE_PIXEL(3, m, mbj)
        if (both_possible) {
            const uint32_t both = md & mb;
            FS_QUEUE_PIXEL(0, both, 0x03020100u) FS_QUEUE_PIXEL(1, both, 0x03020100u) FS_QUEUE_PIXEL(2, both, 0x03020100u)
            FS_QUEUE_PIXEL(3, both, 0x03020100u)
        }
#undef FS_QUEUE_PIXEL
        FS_MARK("arc_end");
    };

    // One pass = every corner of the region columns [col_lo, col_hi) at threshold `thr`, and of those with a column in
    // [out_lo, out_hi) the keepers.  Returns false (nothing written) when the corners do not fit the list.
    auto run_pass = [&](int col_lo, int col_hi, int out_lo, int out_hi) -> bool {
        const int t6 = min(max((thr + 1) >> 2, 0), 64); // ceil((thr - 2) / 4) for thr >= 0
        Kd = 0x01010101u * (uint32_t)(128 - t6);
        both_possible = t6 == 0;
        const int arc_batch = both_possible ? 32 : 64; // a batch of items queues at most 256 pixel entries
        iq_head = iq_tail = cq_head = cq_tail = 0;
        list_n = 0;
        FS_COUNT(6, 1);
        // ---- stage 2: compass test.  Lane = (group of four region columns, row mod FS_R), fixed for the pass: the
        // validity mask and every address offset are per-lane constants, a step advances FS_R rows.
        {
            const int g_lo = col_lo >> 2, g_hi = (col_hi + 3) >> 2;
            // lanes per tile row: the smallest power of two that holds the range's groups -- a one-cell retry pass (8 groups)
            // then takes 8 rows per step instead of 2 with three quarters of the wave idle
            const int lg_sh = (g_hi - g_lo) <= 8 ? 3 : (g_hi - g_lo) <= 16 ? 4 : 5, lgm = (1 << lg_sh) - 1, R = 64 >> lg_sh;
            static_assert(FS_LG <= 32 && 36 + 8 <= FS_TROWS + 6, "a step of 8 rows must stay inside the tile");
            const int g = min(g_lo + (lane & lgm), g_hi - 1), rsub = lane >> lg_sh;
            const uint32_t vm = (g_lo + (lane & lgm) < g_hi)
                                    ? (0x80808080u & byte_prefix_mask(col_hi - 4 * g) & ~byte_prefix_mask(col_lo - 4 * g)) : 0u;
            const uint32_t *t = reinterpret_cast<const uint32_t *>(&tile[rsub * FS_TP + 4 * g]); // (row r - 3, column tc - 4)
            uint32_t ent = (uint32_t)rsub | ((uint32_t)g << 8);
            const int t_step = R * (FS_TP / 4);
            // Only the LAST step of a pass can hold rows below the cell (ch = 30 and R = 2 for a full strip: none at all); every
            // other step takes the lanes' validity mask as it is.  In the last one the rows a lane still has below it become a mask
            // arithmetically ((left - 1) >> 31 is all ones once left <= 0: subtract, shift and a three-input logic operation are in
            // the cheap issue class, the compare and select they replace are not, profiles/r03_valu_ops3.txt).  The choice is a
            // scalar branch.
            int rsub_v = rsub;
            asm volatile("" : "+v"(rsub_v));
            for (int r0 = 0; r0 < ch; r0 += R, t += t_step, ent += (uint32_t)R) {
                FS_COUNT(2, 1);
                FS_MARK("compass_begin");
                const uint32_t M6 = 0x3F3F3F3Fu;
                const uint32_t cm = t[3 * (FS_TP / 4)], cr = t[3 * (FS_TP / 4) + 1], cp = t[3 * (FS_TP / 4) + 2];
                const uint32_t up = (t[1] >> 2) & M6, dn = (t[6 * (FS_TP / 4) + 1] >> 2) & M6, c0 = (cr >> 2) & M6;
                const uint32_t E = __builtin_amdgcn_alignbit(cp, cr, 26) & M6, Wv = __builtin_amdgcn_alignbit(cr, cm, 10) & M6;
                const uint32_t Vd = c0 + Kd, Vb = Kd - c0;
                // a 9-arc holds two adjacent compass points: (S or N) and (E or W), all darker or all brighter
                uint32_t m = ((Vd - dn) | (Vd - up)) & ((Vd - E) | (Vd - Wv));
                m |= ((Vb + dn) | (Vb + up)) & ((Vb + E) | (Vb + Wv));
                if (r0 + R <= ch) {
                    m &= vm;
                } else {
                    const int left = (ch - r0) - rsub_v;
                    uint32_t gone; // all ones once left <= 0 (written as an instruction: the compiler would turn the shift back into compare + select)
                    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(gone) : "v"(left - 1));
                    m &= vm & ~gone;
                }
                const bool surv = m != 0;
                const u64 mk = __ballot(surv);
                if (surv) iq[wave_rank_from(mk, iq_tail) & (FS_IQ - 1)] = m | ent;
                iq_tail += (int)__popcll(mk);
                FS_MARK("compass_end");
                while (iq_tail - iq_head >= 64) {
                    stage_arc(arc_batch);
                    FS_WAVE_ORDER();
                    while (cq_tail - cq_head >= 64) { stage_score(64); FS_WAVE_ORDER(); }
                }
                // more corners than the list holds already: the rest of the pass would be thrown away with it
                if (list_n > FS_LIST) { FS_COUNT(9, 1); return false; }
            }
        }
        while (iq_tail > iq_head) {
            stage_arc(min(iq_tail - iq_head, arc_batch));
            FS_WAVE_ORDER();
            while (cq_tail - cq_head >= 64) { stage_score(64); FS_WAVE_ORDER(); }
        }
        while (cq_tail > cq_head) { stage_score(min(cq_tail - cq_head, 64)); FS_WAVE_ORDER(); }

        if (list_n > FS_LIST) { FS_COUNT(9, 1); return false; } // more corners than the list holds: the caller redoes the range in pieces
        if (list_n == 0) return true; // no corner at this threshold: nothing to suppress, and the tile stays intact for the next pass
        tile_ok = false;

        // ---- stage 5: the score map takes the tile's place (every stage that reads pixels has drained), then the strict
        // 3x3 NMS inside each cell, and the keepers with a column in [out_lo, out_hi) leave
        FS_WAVE_ORDER();
        for (int i = lane; i < 32 * FS_SP / 16; i += 64) reinterpret_cast<uint4 *>(score)[i] = make_uint4(0, 0, 0, 0);
        FS_WAVE_ORDER();
        for (int i = lane; i < list_n; i += 64) {
            const uint32_t e = list[i];
            const int r = e & 255, c = (e >> 8) & 255;
            score[(r + 1) * FS_SP + c + ((c * 2185) >> 16) + 1] = (uint8_t)(e >> 16); // c / 30 = c * 2185 >> 16 for c < 150
        }
        FS_WAVE_ORDER();
        int n_keep = 0;
        // keepers are compacted in place: a batch is read before it is written, at positions <= the ones just read
        for (int i0 = 0; i0 < list_n; i0 += 64) {
            FS_COUNT(5, 1);
            FS_MARK("nms_begin");
            const int i = min(i0 + lane, list_n - 1);
            const uint32_t e = list[i];
            const int r = e & 255, c = (e >> 8) & 255, sc = e >> 16;
            const uint8_t *sp = &score[r * FS_SP + c + ((c * 2185) >> 16)]; // (row - 1, column - 1) of the 3 x 3 block
            const bool keep = i0 + lane < list_n && c >= out_lo && c < out_hi && sc > sp[0] && sc > sp[1] && sc > sp[2] &&
                              sc > sp[FS_SP] && sc > sp[FS_SP + 2] && sc > sp[2 * FS_SP] && sc > sp[2 * FS_SP + 1] &&
                              sc > sp[2 * FS_SP + 2];
            if (keep) s_cellkeep[(c * 2185) >> 16] = 1;
            const u64 mk = __ballot(keep);
            FS_WAVE_ORDER();
            if (keep) list[n_keep + wave_rank(mk)] = e;
            n_keep += (int)__popcll(mk);
            FS_MARK("nms_end");
        }
        if (n_keep == 0) return true;
        int base = 0;
        if (lane == 0) base = atomicAdd(&cand_count[frame * ORBX_MAX_LEVELS + level], n_keep);
        base = __builtin_amdgcn_readfirstlane(base);
        u64 *out = cand + (size_t)frame * cand_fs + lv.cand_off + base;
        const uint32_t xb = st.cx0 * ORBX_CELL, yb = st.cy * ORBX_CELL;
        FS_WAVE_ORDER();
        for (int i = lane; i < n_keep; i += 64) {
            const uint32_t e = list[i];
            out[i] = (u64)((xb + ((e >> 8) & 255)) | ((yb + (e & 255)) << 16)) | ((u64)(e >> 16) << 32);
        }
        return true;
    };
    // Work list of the strip, one call site for the pass: the whole strip at the ini threshold; then (reference :604-607)
    // every cell left without a keeper again at the min threshold.  A pass leaves the score map where the pixels were, so
    // each one starts by (re)loading the tile.  A range with more corners than the list holds (noise at a low threshold)
    // is redone in ranges of FS_NARROW columns, each computed with one more column either side so that the 3x3
    // neighbourhoods of its own columns are complete.
    thr = levels->ini_th;
    // narrow >= 0: first column of the next piece of [lo, hi), nw its width.  A range that overflowed is first redone in pieces of
    // two cells (the dense upper levels hold 550-650 corners per four-cell strip: two passes instead of eight thin ones), and
    // only a piece that overflows again in slices of FS_NARROW columns, which always fit.
    int lo = 0, hi = Ws, cell = -1, narrow = -1, nw = FS_NARROW;
    bool retry = false;
    for (;;) {
        int clo = lo, chi = hi, olo = lo, ohi = hi;
        if (narrow >= 0) {
            olo = narrow; ohi = min(narrow + nw, hi);
            clo = max(olo - 1, 0); chi = min(ohi + 1, Ws);
        }
        if (!tile_ok) { load_tile(); tile_ok = true; }
        if (narrow >= 0) FS_COUNT(10, 1);
        if (retry) FS_COUNT(11, 1);
        if (!run_pass(clo, chi, olo, ohi)) {
            if (narrow < 0) { narrow = lo; nw = hi - lo > 2 * ORBX_CELL ? 2 * ORBX_CELL : FS_NARROW; }
            else nw = FS_NARROW; // (cannot happen for a slice of FS_NARROW columns)
            continue;
        }
        if (narrow >= 0) {
            narrow += nw;
            if (narrow < hi) continue;
            narrow = -1;
        }
        FS_WAVE_ORDER();
        if (!retry) { retry = true; thr = levels->min_th; }
        do ++cell; while (cell * ORBX_CELL < Ws && __builtin_amdgcn_readfirstlane(s_cellkeep[min(cell, FS_K - 1)]));
        if (cell * ORBX_CELL >= Ws) break;
        lo = cell * ORBX_CELL;
        // a run of neighbouring cells without a keeper is one retry pass (the flags of the cells ahead are still the main
        // pass's: a retry only ever sets the flags of its own cells)
        while ((cell + 1) * ORBX_CELL < Ws && !__builtin_amdgcn_readfirstlane(s_cellkeep[min(cell + 1, FS_K - 1)])) ++cell;
        hi = min((cell + 1) * ORBX_CELL, Ws);
    }
}

int orbx_build_fast_strips(const OrbxLevels &levels, int level_begin, int level_end, uint16_t *out /* 4 per strip, or NULL */)
{
    int n = 0;
    for (int l = level_begin; l < level_end && l < levels.n_levels; ++l) {
        const OrbxLevel &v = levels.lv[l];
        if (v.n_cols <= 0) continue;
        // strips of a cell row: as equal as possible, none wider than FS_K cells
        const int ns = (v.n_cols + FS_K - 1) / FS_K;
        for (int cy = 0; cy < v.n_rows; ++cy)
            for (int s = 0; s < ns; ++s) {
                const int c0 = (int)((long long)v.n_cols * s / ns), c1 = (int)((long long)v.n_cols * (s + 1) / ns);
                if (out) { out[4 * n] = (uint16_t)l; out[4 * n + 1] = (uint16_t)cy; out[4 * n + 2] = (uint16_t)c0; out[4 * n + 3] = (uint16_t)(c1 - c0); }
                ++n;
            }
    }
    return n;
}

void orbx_launch_fast_strips(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                             const OrbxLevels &levels, const OrbxBuffers &b, const void *d_strips, int n_strips, int n_frames)
{
    if (n_strips <= 0) return;
    FastSrc src;
    for (int l = 0; l < levels.n_levels; ++l) {
        src.base[l] = l == 0 ? l0 : b.img_arena + levels.lv[l].raw_off;
        src.frame_stride[l] = l == 0 ? l0_fs : b.img_frame_stride;
        src.pitch[l] = l == 0 ? l0_pitch : levels.lv[l].pitch;
    }
    hipLaunchKernelGGL(k_fast_strip, dim3(orbx_xcd_grid(n_strips, n_frames)), dim3(64), dev_pad(k_fast_strip, 1, 0), s, src, d_levels,
                       reinterpret_cast<const FastStrip *>(d_strips), b.cand, b.cand_frame_stride, b.cand_count, n_strips,
                       n_frames);
}

int orbx_build_fast_cells(const OrbxLevels &levels, uint16_t *out /* 4 per cell, or NULL to count */)
{
    int n = 0;
    for (int l = 0; l < levels.n_levels; ++l) {
        const OrbxLevel &v = levels.lv[l];
        for (int cy = 0; cy < v.n_rows; ++cy)
            for (int cx = 0; cx < v.n_cols; ++cx) {
                if (out) { out[4 * n] = (uint16_t)l; out[4 * n + 1] = (uint16_t)cy; out[4 * n + 2] = (uint16_t)cx; out[4 * n + 3] = 0; }
                ++n;
            }
    }
    return n;
}

void orbx_launch_fast(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                      const OrbxLevels &levels, const OrbxBuffers &b, const void *d_cells, int n_cells, int n_frames,
                      int cells_per_group)
{
    if (n_cells <= 0) return;
    FastSrc src;
    for (int l = 0; l < levels.n_levels; ++l) {
        src.base[l] = l == 0 ? l0 : b.img_arena + levels.lv[l].raw_off;
        src.frame_stride[l] = l == 0 ? l0_fs : b.img_frame_stride;
        src.pitch[l] = l == 0 ? l0_pitch : levels.lv[l].pitch;
    }
    const FastCell *cells = reinterpret_cast<const FastCell *>(d_cells);
#define ORBX_FAST_CELLS_LAUNCH(CPW)                                                                                          \
    hipLaunchKernelGGL(k_fast_cells_wave<CPW>, dim3(orbx_xcd_grid((n_cells + CPW - 1) / CPW, n_frames)), dim3(64 * CPW), 0, s, \
                       src, d_levels, cells, b.cand, b.cand_frame_stride, b.cand_count, n_cells, n_frames)
    if (cells_per_group >= 16) ORBX_FAST_CELLS_LAUNCH(16);
    else if (cells_per_group >= 8) ORBX_FAST_CELLS_LAUNCH(8);
    else if (cells_per_group >= 4) ORBX_FAST_CELLS_LAUNCH(4);
    else ORBX_FAST_CELLS_LAUNCH(1);
#undef ORBX_FAST_CELLS_LAUNCH
}

// ---------------------------------------------------------------------------------------------
// 7x7 Gaussian, 8.8 fixed-point taps, BORDER_REFLECT_101: H pass exact in u16, V pass rounded >>16
// ---------------------------------------------------------------------------------------------
#define BL_W 256 // tile = 64 lanes x 4 pixels wide ...
#define BL_ROWS 32
#define BL_H (4 * BL_ROWS) // ... and 4 waves x 32 rows high
struct BlurTile {
    uint16_t level, tx, ty, pad;
};

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// horizontal 7-tap sums (exact, <= 65535) of the four pixels whose left neighbourhood starts at byte 0
// of d1: d0|d1|d2 hold level bytes x-4 .. x+7
__device__ __forceinline__ void blur_hrow(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t K0, uint32_t K1, uint32_t o[4])
{
    o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), K0, 0u, false), false);
    o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), K0, 0u, false), false);
    o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), K1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), K0, 0u, false), false);
    o[3] = __builtin_amdgcn_udot4(d2, K1, __builtin_amdgcn_udot4(d1, K0, 0u, false), false);
}

// 7x7 Gaussian as a register sliding window: a thread owns 4 adjacent columns and walks down 32 output
// rows.  Per input row it loads 12 bytes, forms the four horizontal sums with v_alignbyte + v_dot4_u32_u8
// and keeps the last seven rows of sums in registers (ring unrolled by 7, so slots are compile-time);
// every row after the sixth emits one output dword = sat8((sum k_j * H_j + 2^15) >> 16) with 24-bit
// multiply-adds (H < 2^16).  No LDS, no barrier.  BORDER = false is the straight-line interior path; waves
// that touch the left/right image edge take the BORDER = true instantiation (wave-uniform choice).
// one v_mad_u32_u24 (the compiler otherwise splits the 7-tap sum into 24-bit multiplies plus an add tree)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool BORDER>
__device__ __forceinline__ void blur_strip(const uint8_t *__restrict__ S, int pitch, int w, int h, uint8_t *__restrict__ D,
                                           int dpitch, int x, int ys, int ye, const uint32_t k[7], uint32_t K0,
                                           uint32_t K1, bool clamp16)
{
    uint32_t hq[7][4];
    // BORDER_REFLECT_101 column indices of the 12 bytes x-4 .. x+7: the same for every row
    int cidx[12];
    if (BORDER) {
#pragma unroll
        for (int b = 0; b < 12; ++b) cidx[b] = reflect101(x - 4 + b, w);
    }
    auto load_row = [&](int rr, uint32_t o[4]) {
        const uint8_t *row = S + (size_t)reflect101(rr, h) * pitch;
        uint32_t d0, d1, d2;
        if (!BORDER) {
            d0 = reinterpret_cast<const UnalignedU32 *>(row + x - 4)->v;
            d1 = reinterpret_cast<const UnalignedU32 *>(row + x)->v;
            d2 = reinterpret_cast<const UnalignedU32 *>(row + x + 4)->v;
        } else {
            d0 = d1 = d2 = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                d0 |= (uint32_t)row[cidx[b]] << (8 * b);
                d1 |= (uint32_t)row[cidx[4 + b]] << (8 * b);
                d2 |= (uint32_t)row[cidx[8 + b]] << (8 * b);
            }
        }
        blur_hrow(d0, d1, d2, K0, K1, o);
        if (clamp16) { // only the sum-257 tap set can exceed 16 bits
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = min(o[i], 65535u);
        }
    };
    // input row n (level row ys-3+n) goes to slot n%7; from n = 6 on, output row ys+n-6 is complete:
    // its rows y-3 .. y+3 sit in slots (n+1)%7 .. (n+7)%7.  One unrolled-by-7 loop, no separate priming code.
    const int n_rows = ye - ys + 6;
    auto emit = [&](int s, int n) {
        const int y = ys + n - 6;
        uint32_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t acc = 32768u;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = mad_u24(k[j], hq[(s + 1 + j) % 7][i], acc);
            r[i] = min(acc >> 16, 255u);
        }
        // rows of the arena are padded to a multiple of 64 bytes: the dword store stays inside the row
        *reinterpret_cast<uint32_t *>(D + (size_t)y * dpitch + x) = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
    };
    if (!BORDER) {
        // Interior columns: the three dwords of the rows n+1 and n+2 are requested before row n is used, so a wave has
        // two loads in flight while it computes instead of a load -> wait -> compute -> store chain per row.  Row indices
        // are reflected with a closed form (|overshoot| <= 3 < h) and clamped, so a prefetch past the last needed row
        // reads a valid row and is simply not used.
        struct __attribute__((packed, aligned(1))) U96 { uint32_t w[3]; };
        auto fetch = [&](int n) -> U96 {
            int rr = ys - 3 + min(n, n_rows - 1);
            rr = rr < 0 ? -rr : (rr >= h ? 2 * h - 2 - rr : rr);
            rr = min(max(rr, 0), h - 1);
            return *reinterpret_cast<const U96 *>(S + (size_t)rr * pitch + x - 4);
        };
        // a ring of seven register triples indexed like hq (slot = n % 7, compile-time in the unrolled loop): no register
        // copies, which would make the wave wait for the very loads that are meant to stay in flight
        U96 q[7];
        if (ys - 3 >= 0 && ys - 3 + n_rows + 2 <= h) {
            // band in the interior of the level (wave-uniform): every row it touches, the two prefetched beyond it
            // included, exists -- the row pointer just advances by the pitch
            const uint8_t *rp = S + (size_t)(ys - 3) * pitch + x - 4;
            q[0] = *reinterpret_cast<const U96 *>(rp);
            q[1] = *reinterpret_cast<const U96 *>(rp + pitch);
            rp += 2 * (size_t)pitch;
            for (int n0 = 0; n0 < n_rows; n0 += 7) {
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    const int n = n0 + s;
                    q[(s + 2) % 7] = *reinterpret_cast<const U96 *>(rp);
                    rp += (n + 3 < n_rows + 2) ? pitch : 0; // stop at the last existing prefetch row
                    blur_hrow(q[s].w[0], q[s].w[1], q[s].w[2], K0, K1, hq[s]);
                    if (clamp16) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) hq[s][i] = min(hq[s][i], 65535u);
                    }
                    if (n >= 6 && n < n_rows) emit(s, n);
                }
            }
            return;
        }
        q[0] = fetch(0);
        q[1] = fetch(1);
        // the body has no branch on n (rows past the band are clamped re-reads whose results are never stored), so that
        // the compiler's s_waitcnt placement can leave the younger loads in flight
        for (int n0 = 0; n0 < n_rows; n0 += 7) {
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const int n = n0 + s;
                q[(s + 2) % 7] = fetch(n + 2);
                blur_hrow(q[s].w[0], q[s].w[1], q[s].w[2], K0, K1, hq[s]);
                if (clamp16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) hq[s][i] = min(hq[s][i], 65535u);
                }
                if (n >= 6 && n < n_rows) emit(s, n);
            }
        }
        return;
    }
    for (int n0 = 0; n0 < n_rows; n0 += 7) {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int n = n0 + s;
            if (n < n_rows) {
                load_row(ys - 3 + n, hq[s]);
                if (n >= 6) emit(s, n);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_blur_cols(FastSrc src, const OrbxLevels *__restrict__ levels,
                                                   const BlurTile *__restrict__ tiles, uint8_t *__restrict__ arena,
                                                   size_t arena_fs, const int *__restrict__ taps, int n_tiles, int n_frames)
{
    int frame, tile_id; // all tiles of a frame on one XCD: the 6 halo rows between bands are then L2 hits
    if (!xcd_remap(n_tiles, n_frames, &frame, &tile_id)) return;
    const BlurTile t = tiles[tile_id];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, level = t.level;
    const OrbxLevel &lv = levels->lv[level];
    const int w = lv.w, h = lv.h, pitch = src.pitch[level];
    const int x = t.tx * BL_W + 4 * lane;
    const int ys = t.ty * BL_H + wave * BL_ROWS;
    if (x >= w || ys >= h) return;
    const int ye = min(ys + BL_ROWS, h);
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level];
    uint8_t *D = arena + (size_t)frame * arena_fs + lv.blur_off;
    uint32_t k[7];
    uint32_t ksum = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { k[i] = (uint32_t)taps[i]; ksum += k[i]; }
    const uint32_t K0 = k[0] | (k[1] << 8) | (k[2] << 16) | (k[3] << 24), K1 = k[4] | (k[5] << 8) | (k[6] << 16);
    // column groups that touch the left / right image edge are produced by k_blur_edges (reflected columns)
    if (!(x >= 4 && x + 8 <= w)) return;
    blur_strip<false>(S, pitch, w, h, D, lv.pitch, x, ys, ye, k, K0, K1, ksum > 256);
}

// The (at most three) 4-column groups per row that need BORDER_REFLECT_101 on the columns: x = 0 and the
// groups with x + 8 > w.  One lane = one group x 8 output rows; a few thousand lanes per batch.
#define BE_ROWS 8
__global__ __launch_bounds__(64) void k_blur_edges(FastSrc src, const OrbxLevels *__restrict__ levels,
                                                   uint8_t *__restrict__ arena, size_t arena_fs,
                                                   const int *__restrict__ taps, int level_begin)
{
    const int frame = blockIdx.y, level = level_begin + blockIdx.z;
    const OrbxLevel &lv = levels->lv[level];
    const int w = lv.w, h = lv.h, pitch = src.pitch[level];
    const int task = blockIdx.x * 64 + threadIdx.x;
    const int strip = task / 3, which = task - 3 * strip;
    const int ys = strip * BE_ROWS;
    if (ys >= h) return;
    const int last = ((w - 1) >> 2) << 2; // x of the last group
    const int x = which == 0 ? 0 : which == 1 ? last : last - 4;
    if (x < 0 || (x >= 4 && x + 8 <= w)) return;  // interior after all (or no such group)
    if (which != 0 && x == 0) return;            // already covered by which == 0
    if (which == 2 && last - 4 == 0) return;
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level];
    uint8_t *D = arena + (size_t)frame * arena_fs + lv.blur_off;
    uint32_t k[7];
    uint32_t ksum = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { k[i] = (uint32_t)taps[i]; ksum += k[i]; }
    const uint32_t K0 = k[0] | (k[1] << 8) | (k[2] << 16) | (k[3] << 24), K1 = k[4] | (k[5] << 8) | (k[6] << 16);
    blur_strip<true>(S, pitch, w, h, D, lv.pitch, x, ys, min(ys + BE_ROWS, h), k, K0, K1, ksum > 256);
}

// levels [level_begin, level_end) of the blurred pyramid; the tile list is level-major
void orbx_launch_blur(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                      const OrbxLevels &levels, const OrbxBuffers &b, const void *d_tiles, int n_tiles, const int *taps7,
                      int n_frames, int level_begin, int level_end)
{
    if (n_tiles <= 0 || level_begin >= level_end) return;
    FastSrc src;
    for (int l = 0; l < levels.n_levels; ++l) {
        src.base[l] = l == 0 ? l0 : b.img_arena + levels.lv[l].raw_off;
        src.frame_stride[l] = l == 0 ? l0_fs : b.img_frame_stride;
        src.pitch[l] = l == 0 ? l0_pitch : levels.lv[l].pitch;
    }
    int first = 0, count = 0, max_h = 1;
    for (int l = 0; l < level_end; ++l) {
        const int nt = ((levels.lv[l].h + BL_H - 1) / BL_H) * ((levels.lv[l].w + BL_W - 1) / BL_W);
        if (l < level_begin) first += nt;
        else {
            count += nt;
            max_h = levels.lv[l].h > max_h ? levels.lv[l].h : max_h;
        }
    }
    if (count <= 0) return;
    hipLaunchKernelGGL(k_blur_cols, dim3(orbx_xcd_grid(count, n_frames)), dim3(256), 0, s, src, d_levels,
                       reinterpret_cast<const BlurTile *>(d_tiles) + first, b.img_arena, b.img_frame_stride, taps7, count,
                       n_frames);
    const int tasks = 3 * ((max_h + BE_ROWS - 1) / BE_ROWS);
    hipLaunchKernelGGL(k_blur_edges, dim3((tasks + 63) / 64, n_frames, level_end - level_begin), dim3(64), 0, s, src, d_levels,
                       b.img_arena, b.img_frame_stride, taps7, level_begin);
}

// ---------------------------------------------------------------------------------------------
// The same 7x7 Gaussian on the MATRIX pipe (levels at least 64 x 8; the VALU kernels above stay the parity twin and
// serve the small levels).  Both passes are banded matrix products, exact in i8 x i8 -> i32:
//   H pass   H[y][x] = sum_i k_i src[y][x + i - 3]          = (pixels - 128) (32 rows x 64 columns) x band (64 x 32), + 32768
//   V pass   V[y][x] = sum_j k_j H[y + j - 3][x],  H = 256 Hhi + Hlo, each byte (minus 128) its own product
// with BORDER_REFLECT_101 folded into the band matrices of the border tiles (host tables: a tap whose source lies
// outside the image is added to the coefficient of the pixel it reflects to).  One wave owns a strip of 32 columns and
// walks down 32 rows at a time.  The H tile comes out with its column on the lane and its rows in the 16 accumulator
// registers, which is exactly the A-operand layout of a product that sums over those rows (cdna_hip_programming.md,
// "an accumulator tile as the next MFMA's operand"): byte 1 of every accumulator is the signed high byte of H - 32768,
// byte 0 ^ 0x80 the signed low byte, so the hand-over is 24 v_perm + 4 v_xor and no LDS.  Using H as the A operand makes the
// V product come out TRANSPOSED (output row on the lane, four groups of four consecutive columns in the registers), so
// a lane stores four dwords per tile instead of sixteen bytes.  Per 32 x 32 output tile: 6 MFMA and about 110 VALU
// (7 lane-operations per pixel against 23 of the VALU kernel).
// ---------------------------------------------------------------------------------------------
typedef int bl_v4i __attribute__((ext_vector_type(4)));
typedef int bl_v16i __attribute__((ext_vector_type(16)));
#ifndef BM_COLS
#define BM_COLS 128  // output columns of a workgroup: one wave and one 32-column MFMA tile per 32 columns.  (As a plain copy this access
                     // pattern moves 4.2 TB/s -- the kernel's own rate -- and 5.3 TB/s in 256-column blocks, tools/microbench/colblock_copy.hip;
                     // the kernel itself is no faster that way, 0.40 against 0.38 ms per 512 frames: eight-wave workgroups at 88 VGPRs
                     // leave 16 waves per CU instead of 20, and the last block of a level idles more waves.)
#endif
#define BM_WAVES (BM_COLS / 32)
#define BM_T (64 * BM_WAVES)           // threads
#define BM_SRC_W (BM_COLS + 32)       // source bytes staged per row: 16 either side, in 16-byte chunks
#define BM_CH (BM_SRC_W / 16)
#define BM_SRC_P (BM_SRC_W + 16)      // LDS pitch of a staged source row (128 columns: 176, conflict-free 16-byte operand reads)
#define BM_OUT_P (BM_COLS + 4)        // LDS pitch of a finished output row (an odd number of dwords: conflict-free dword writes)
#define BM_OCH (BM_COLS / 16)         // 16-byte chunks of a finished row
static_assert(32 * BM_CH <= 2 * BM_T && 32 * BM_OCH == BM_T, "a thread stages at most two chunks and stores one");
struct __attribute__((aligned(4))) BlurBlock { uint16_t level, bx; };
struct __attribute__((packed, aligned(1))) UnalignedV4 { bl_v4i v; };
struct __attribute__((packed, aligned(1))) UnalignedU4 { uint32_t x, y, z, w; };

// The 160 staged source columns of block bx start at blur_origin(); the 64-column window (two K-steps) of tile column tx
// at blur_window().  `limit` = readable bytes of a row (the width for level 0, the padded pitch for the arena levels).
// The host tables use the same two rules.
__host__ __device__ __forceinline__ int blur_origin(int bx, int limit) { return min(max(BM_COLS * bx - 16, 0), limit - BM_SRC_W); }
__host__ __device__ __forceinline__ int blur_window(int tx, int limit)
{
    const int xo = blur_origin(tx / BM_WAVES, limit);
    return min(max(32 * tx - 16, xo), xo + BM_SRC_W - 64);
}

template <int K_SUM> // the taps sum to 256 (default set), or to 257 (the plain-rounded set: two more operations per pixel)
__global__ __launch_bounds__(BM_T) void k_blur_mfma(FastSrc src, BlurMfmaLevels lv, const BlurBlock *__restrict__ blocks,
                                                   const uint4 *__restrict__ band_h, const uint4 *__restrict__ band_v,
                                                   uint8_t *__restrict__ arena, size_t arena_fs, int n_blocks, int n_frames)
{
    constexpr int hbias = 128 * K_SUM - 32768;                 // H - 128 K  ->  H - 32768
    constexpr uint32_t vbias = 128u * K_SUM * 257u + 32768u;   // the two 128 K of the byte products, and the rounding 2^15
    constexpr bool clamp255 = K_SUM != 256;
    __shared__ __align__(16) uint8_t s_src[2][32 * BM_SRC_P];
    __shared__ __align__(16) uint8_t s_out[2][32 * BM_OUT_P];
    int frame, bid;
    if (!xcd_remap(n_blocks, n_frames, &frame, &bid)) return;
    const BlurBlock bk = blocks[bid];
    const int level = bk.level, tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = lv.w[level], h = lv.h[level], pitch = src.pitch[level], n_ty = lv.n_ty[level], dpitch = lv.dst_pitch[level];
    const int limit = level == 0 ? w : pitch;
    const int xo = blur_origin(bk.bx, limit);
    const int n_tx = (w + 31) >> 5, tx = min(BM_WAVES * (int)bk.bx + wave, n_tx - 1); // a wave past the last tile repeats it (not stored)
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level];
    uint8_t *D = arena + (size_t)frame * arena_fs + lv.dst_off[level];
    // band of the H pass for this wave's tile column: two K-steps of 32 source columns
    const uint4 *bh = band_h + lv.bh_off[level] + (size_t)tx * 128;
    const bl_v4i bh0 = __builtin_bit_cast(bl_v4i, bh[lane]), bh1 = __builtin_bit_cast(bl_v4i, bh[64 + lane]);
    const int a_off = n * BM_SRC_P + (blur_window(tx, limit) - xo) + 16 * hh; // this lane's 16 operand bytes of K-step 0

    // staging: 32 rows x BM_CH 16-byte chunks per H tile (128 columns: 320 chunks), thread t takes chunk t and, the first ones, chunk BM_T + t.
    // H tile T = rows 32 T - 3 .. 32 T + 28 (rows outside the image are clamped: their coefficients are zero).
    const int c0r = tid / BM_CH, c0c = tid - BM_CH * c0r, c1r = (BM_T + tid) / BM_CH, c1c = (BM_T + tid) - BM_CH * c1r;
    const bool second = tid < 32 * BM_CH - BM_T;
    auto g_load = [&](int T, uint4 *u0, uint4 *u1) {
        const int r0 = min(max(32 * T - 3 + c0r, 0), h - 1), r1 = min(max(32 * T - 3 + min(c1r, 31), 0), h - 1);
        const UnalignedU4 a = *reinterpret_cast<const UnalignedU4 *>(S + (size_t)r0 * pitch + xo + 16 * c0c);
        const UnalignedU4 b2 = *reinterpret_cast<const UnalignedU4 *>(S + (size_t)r1 * pitch + xo + 16 * (second ? c1c : 0));
        *u0 = make_uint4(a.x, a.y, a.z, a.w);
        *u1 = make_uint4(b2.x, b2.y, b2.z, b2.w);
    };
    auto s_store = [&](int buf, const uint4 &u0, const uint4 &u1) {
        *reinterpret_cast<uint4 *>(&s_src[buf][c0r * BM_SRC_P + 16 * c0c]) = u0;
        if (second) *reinterpret_cast<uint4 *>(&s_src[buf][c1r * BM_SRC_P + 16 * c1c]) = u1;
    };
    // the H tile staged in `buf`, turned into the two A operands of the V pass
    auto h_tile = [&](int buf, bl_v4i *hi, bl_v4i *lo) {
        bl_v4i a0 = reinterpret_cast<const UnalignedV4 *>(&s_src[buf][a_off])->v;
        bl_v4i a1 = reinterpret_cast<const UnalignedV4 *>(&s_src[buf][a_off + 32])->v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] ^= (int)0x80808080u; a1[k] ^= (int)0x80808080u; } // pixel - 128 as i8
        const bl_v16i zero = {};
        bl_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bh0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bh1, acc, 0, 0, 0);
        // acc = H - 128 K; with hbias it is H - 32768: byte 1 = (H >> 8) - 128 and byte 0 = H & 255, what the V pass takes
        if (hbias) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += hbias;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = (uint32_t)acc[4 * k], q = (uint32_t)acc[4 * k + 1], r2 = (uint32_t)acc[4 * k + 2], t = (uint32_t)acc[4 * k + 3];
            const uint32_t pq1 = __builtin_amdgcn_perm(q, p, 0x0c0c0501u), rt1 = __builtin_amdgcn_perm(t, r2, 0x0c0c0501u);
            const uint32_t pq0 = __builtin_amdgcn_perm(q, p, 0x0c0c0400u), rt0 = __builtin_amdgcn_perm(t, r2, 0x0c0c0400u);
            (*hi)[k] = (int)__builtin_amdgcn_perm(rt1, pq1, 0x05040100u);
            (*lo)[k] = (int)(__builtin_amdgcn_perm(rt0, pq0, 0x05040100u) ^ 0x80808080u);
        }
    };
    uint4 u0, u1;
    g_load(0, &u0, &u1);
    s_store(0, u0, u1);
    g_load(1, &u0, &u1);
    s_store(1, u0, u1);
    g_load(2, &u0, &u1);
    __syncthreads();
    bl_v4i hiP, loP, hiN, loN;
    h_tile(0, &hiP, &loP);
    __syncthreads(); // buffer 0 is free for tile 2
    const uint4 *bv = band_v + lv.bv_off[level];
    const int orow = tid / BM_OCH, ocol = 16 * (tid % BM_OCH); // this thread's 16 bytes of the finished 32 x BM_COLS tile
    // One barrier per trip: trip ty reads source buffer (ty + 1) & 1, so buffer ty & 1 (last read in trip ty - 1) takes
    // tile ty + 2 at the top of the trip, from registers loaded during the trip before; the finished tiles alternate
    // between two output buffers.
    for (int ty = 0; ty < n_ty; ++ty, bv += 128) {
        const bl_v4i bv0 = __builtin_bit_cast(bl_v4i, bv[lane]), bv1 = __builtin_bit_cast(bl_v4i, bv[64 + lane]);
        s_store(ty & 1, u0, u1);
        g_load(ty + 3, &u0, &u1); // in flight under this trip's arithmetic (rows clamped past the image)
        h_tile((ty + 1) & 1, &hiN, &loN);
        const bl_v16i zero = {};
        bl_v16i ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiP, bv0, zero, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiN, bv1, ah, 0, 0, 0);
        bl_v16i al = __builtin_amdgcn_mfma_i32_32x32x32_i8(loP, bv0, zero, 0, 0, 0);
        al = __builtin_amdgcn_mfma_i32_32x32x32_i8(loN, bv1, al, 0, 0, 0);
        // sum k H = 256 (ah + 128 K) + (al + 128 K) with K = sum of the taps; + 2^15 and >> 16 is byte 2 of the total.
        // The product is transposed: this lane holds output row n, columns 8 g + 4 hh + (0..3) of its wave's tile.
        uint8_t *so = s_out[ty & 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = (uint32_t)(ah[4 * g + k] * 256 + al[4 * g + k]) + vbias;
                if (clamp255) v[k] = min(v[k], 0x00FFFFFFu);
            }
            const uint32_t a = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), b2 = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);
            *reinterpret_cast<uint32_t *>(&so[n * BM_OUT_P + 32 * wave + 8 * g + 4 * hh]) = __builtin_amdgcn_perm(b2, a, 0x05040100u);
        }
        __syncthreads(); // the finished tile and the source of the next trip are complete
        {
            const int y = 32 * ty + orow, x = BM_COLS * (int)bk.bx + ocol;
#ifdef BM_PROBE_NOSTORE // timing probe only (tools/build_variant.sh): the blur without its HBM writes
            if (y < h && x < dpitch && frame < 0) {
#else
            if (y < h && x < dpitch) { // columns between the width and the padded pitch take whatever the last tile holds
#endif
                const uint32_t *o = reinterpret_cast<const uint32_t *>(&so[orow * BM_OUT_P + ocol]);
                *reinterpret_cast<uint4 *>(D + (size_t)y * dpitch + x) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        hiP = hiN; loP = loN;
    }
}

static int reflect101_host(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
// Levels 0 .. (return value - 1) are large enough for k_blur_mfma (160 staged source bytes inside every row).
int orbx_blur_mfma_levels(const OrbxLevels &levels)
{
    int n = 0;
    while (n < levels.n_levels && levels.lv[n].w >= BM_SRC_W && levels.lv[n].h >= 8) ++n;
    return n;
}
// Block list (2 x uint16 per workgroup, level-major), the two band tables (16 bytes per lane and K-step) and the per-level
// record of k_blur_mfma.  blocks_before[l] = workgroups of the levels below l.  l0_width_is_limit: level 0 is the caller's
// buffer, nothing beyond a row's width may be read there; the arena levels may be read up to their padded pitch.
void orbx_build_blur_mfma(const OrbxLevels &levels, const int taps[7], std::vector<uint16_t> &blocks, std::vector<uint8_t> &band_h,
                          std::vector<uint8_t> &band_v, BlurMfmaLevels &out, int blocks_before[ORBX_MAX_LEVELS + 1])
{
    blocks.clear(); band_h.clear(); band_v.clear();
    memset(&out, 0, sizeof out);
    const int n_lv = orbx_blur_mfma_levels(levels);
    for (int l = 0; l <= ORBX_MAX_LEVELS; ++l) blocks_before[l] = 0;
    for (int l = 0; l < n_lv; ++l) {
        const OrbxLevel &v = levels.lv[l];
        const int w = v.w, h = v.h, n_tx = (w + 31) / 32, n_ty = (h + 31) / 32, limit = l == 0 ? w : v.pitch;
        out.w[l] = w; out.h[l] = h; out.dst_pitch[l] = v.pitch; out.n_ty[l] = n_ty; out.dst_off[l] = v.blur_off;
        out.bh_off[l] = (int)(band_h.size() / 16); out.bv_off[l] = (int)(band_v.size() / 16);
        blocks_before[l] = (int)(blocks.size() / 2);
        for (int bx = 0; bx < (w + BM_COLS - 1) / BM_COLS; ++bx) { blocks.push_back((uint16_t)l); blocks.push_back((uint16_t)bx); }
        for (int tx = 0; tx < n_tx; ++tx) {
            // H pass: coefficient of source column cw + kk (64-column window, two K-steps) for output column 32 tx + n
            int B[64][32] = {};
            const int x0 = 32 * tx, cw = blur_window(tx, limit);
            for (int n = 0; n < 32 && x0 + n < w; ++n)
                for (int i = 0; i < 7; ++i) B[reflect101_host(x0 + n + i - 3, w) - cw][n] += taps[i];
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 16; ++j) band_h.push_back((uint8_t)(int8_t)B[32 * s + 16 * (lane >> 5) + j][lane & 31]);
        }
        for (int ty = 0; ty < n_ty; ++ty) {
            // V pass: coefficient of H-tile row rho (tile ty + step, rows 32 (ty + step) - 3 ...) for output row 32 ty + n
            int C[2][32][32] = {};
            for (int n = 0; n < 32 && 32 * ty + n < h; ++n)
                for (int i = 0; i < 7; ++i) {
                    const int r = reflect101_host(32 * ty + n + i - 3, h);
                    const int step = r >= 32 * (ty + 1) - 3 ? 1 : 0;
                    C[step][r - (32 * (ty + step) - 3)][n] += taps[i];
                }
            for (int step = 0; step < 2; ++step)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 16; ++j) {
                        const int rho = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5); // accumulator row of register j (C/D layout)
                        band_v.push_back((uint8_t)(int8_t)C[step][rho][lane & 31]);
                    }
        }
    }
    for (int l = n_lv; l <= ORBX_MAX_LEVELS; ++l) blocks_before[l] = (int)(blocks.size() / 2);
}

void orbx_launch_blur_mfma(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels &levels,
                           const OrbxBuffers &b, const BlurMfmaLevels &tab, const void *d_blocks, const int *blocks_before,
                           const void *d_band_h, const void *d_band_v, const int taps[7], int n_frames, int level_begin,
                           int level_end)
{
    const int first = blocks_before[level_begin], count = blocks_before[level_end] - first;
    if (count <= 0) return;
    FastSrc src;
    for (int l = 0; l < levels.n_levels; ++l) {
        src.base[l] = l == 0 ? l0 : b.img_arena + levels.lv[l].raw_off;
        src.frame_stride[l] = l == 0 ? l0_fs : b.img_frame_stride;
        src.pitch[l] = l == 0 ? l0_pitch : levels.lv[l].pitch;
    }
    int K = 0;
    for (int i = 0; i < 7; ++i) K += taps[i];
    if (K == 256)
        hipLaunchKernelGGL(k_blur_mfma<256>, dim3(orbx_xcd_grid(count, n_frames)), dim3(BM_T), 0, s, src, tab,
                           reinterpret_cast<const BlurBlock *>(d_blocks) + first, reinterpret_cast<const uint4 *>(d_band_h),
                           reinterpret_cast<const uint4 *>(d_band_v), b.img_arena, b.img_frame_stride, count, n_frames);
    else // K == 257 (orbx_blur_mfma_levels() returns 0 for any other tap set)
        hipLaunchKernelGGL(k_blur_mfma<257>, dim3(orbx_xcd_grid(count, n_frames)), dim3(BM_T), 0, s, src, tab,
                           reinterpret_cast<const BlurBlock *>(d_blocks) + first, reinterpret_cast<const uint4 *>(d_band_h),
                           reinterpret_cast<const uint4 *>(d_band_v), b.img_arena, b.img_frame_stride, count, n_frames);
}

int orbx_build_blur_tiles(const OrbxLevels &levels, uint16_t *out /* 4 per tile, or NULL to count */)
{
    int n = 0;
    for (int l = 0; l < levels.n_levels; ++l) {
        const OrbxLevel &v = levels.lv[l];
        for (int ty = 0; ty < (v.h + BL_H - 1) / BL_H; ++ty)
            for (int tx = 0; tx < (v.w + BL_W - 1) / BL_W; ++tx) {
                if (out) { out[4 * n] = (uint16_t)l; out[4 * n + 1] = (uint16_t)tx; out[4 * n + 2] = (uint16_t)ty; out[4 * n + 3] = 0; }
                ++n;
            }
    }
    return n;
}

// DistributeOctree (orbx_octree.h), compiled for three workgroup sizes
#ifndef ORBX_OCT_HUGE_PIXELS
#define ORBX_OCT_HUGE_PIXELS 1200000 // level size from which a call with a few frames gives the level 1024 threads (1920 x 1080: levels 0 and 1)
#endif
// (every build names its parameters: the header's defaults would otherwise leak from one inclusion into the next)
namespace oct_wide {
#define OCT_NT ORBX_OCT_THREADS
#define OCT_REG 8
#define OCT_MIN_WAVES 1
#define OCT_PRIO 1 // the call waits for these few workgroups: their waves go first on a CU they share (orbx_octree.h)
#include "orbx_octree.h"
#undef OCT_PRIO
#undef OCT_MIN_WAVES
#undef OCT_REG
#undef OCT_NT
} // namespace oct_wide
namespace oct_batch {
#define OCT_NT ORBX_OCT_THREADS_BATCH
#define OCT_REG 8
#define OCT_MIN_WAVES 4 // four 256-thread workgroups per CU: the register allocator stays at 128 VGPRs (133 without the bound: three workgroups, 0.23 -> 0.27 ms)
#include "orbx_octree.h"
#undef OCT_MIN_WAVES
#undef OCT_REG
#undef OCT_NT
} // namespace oct_batch
namespace oct_huge { // a call with a few frames and a level of a megapixel or more: 1024 threads share its 10 000+ candidates
#define OCT_NT 1024
#define OCT_PYR 1 // passes from a count pyramid instead of candidate sweeps (orbx_octree.h)
#define OCT_PRIO 1
#define OCT_REG 1 // 16 waves per workgroup leave 128 VGPRs a thread: candidates are streamed (they are read twice in all), not held
#define OCT_MIN_WAVES 1
#include "orbx_octree.h"
#undef OCT_MIN_WAVES
#undef OCT_REG
#undef OCT_PRIO
#undef OCT_PYR
#undef OCT_NT
} // namespace oct_huge

// one 16-byte aligned layout for every level: [sort keys 8 (P2 + 16 pad keys)] [node 0/1: 4 M each] [cc: 16 M] [rank, newpos: 2 M each]
// [node_of_rank: 2 NSC] [childpos: 8 NSC]
size_t orbx_octree_lds_bytes(const OrbxLevels &levels)
{
    size_t need = 0;
    for (int l = 0; l < levels.n_levels; ++l) {
        const OrbxLevel &v = levels.lv[l];
        size_t P2 = 1;
        while (P2 < (size_t)v.quota) P2 <<= 1;
        const size_t M = v.node_cap, NSC = (size_t)(v.quota > v.n_ini ? v.quota : v.n_ini) + 4;
        const size_t bytes = 8 * (P2 + 16) + 8 * M + 16 * M + 4 * M + 2 * NSC + 8 * NSC + 64;
        if (bytes > need) need = bytes;
    }
    return need;
}

// Which build of the quadtree serves levels [level_begin, level_end) of a call and with how much dynamic LDS -- host logic only
// (tests/test_abi.py sweeps it across the LDS boundaries through orbx_dev_octree_plan, no GPU needed).
//   kind 0: oct_batch::k_octree_lds (256 threads; resident batches)      kind 1: oct_wide::k_octree_lds (512 threads)
//   kind 2: oct_huge::k_octree_lds (1024 threads, count pyramid) for levels [level_begin, huge_end), kind 1 for the rest
//   kind 3: oct_wide::k_octree (node list in global scratch: quotas too large for the LDS-resident list)
OrbxOctPlan orbx_octree_plan(const OrbxLevels &levels, int n_frames, int level_begin, int level_end, int n_cus)
{
    OrbxOctPlan p;
    p.lds_bytes = orbx_octree_lds_bytes(levels); // the largest level's, whatever the range: one configuration
    p.huge_end = level_begin;
    p.huge_bytes = 0;
    bool small_nodes = true; // 16-bit list positions
    for (int l = 0; l < levels.n_levels; ++l) small_nodes = small_nodes && levels.lv[l].node_cap < 65535;
    if (p.lds_bytes > ORBX_OCT_LDS_LIMIT || !small_nodes) { p.kind = 3; return p; }
    // a resident batch has more workgroups than the chip holds at once: narrower workgroups, more of them per CU
    if ((size_t)n_frames * (size_t)levels.n_levels > 512) { p.kind = 0; return p; }
    p.kind = 1;
    // A level of a megapixel or more in a call with a few frames: 1024 threads and the rest of the CU's LDS for the count
    // pyramid.  The list arrays are laid out for lds_bytes, so the build is only taken when they fit what it is launched with
    // (they always did for the default quotas; a 2N re-quota of a large-quota handle can exceed it: the 512-thread build then).
    int lh = level_begin;
    while (lh < level_end && levels.lv[lh].w * levels.lv[lh].h >= ORBX_OCT_HUGE_PIXELS) ++lh;
    if (lh > level_begin && p.lds_bytes <= ORBX_OCT_HUGE_LDS) {
        p.kind = 2;
        p.huge_bytes = ORBX_OCT_HUGE_LDS;
        // One launch for the whole range while every workgroup finds a CU of its own (the small levels then run beside the
        // large ones: a single frame waits for the longest level, not for the sum).  With more workgroups than CUs a
        // 1024-thread, whole-LDS workgroup per small level would queue behind the others: those levels go to the 512-thread
        // build in a second launch.
        p.huge_end = (size_t)(level_end - level_begin) * (size_t)n_frames > (size_t)(n_cus > 0 ? n_cus : ORBX_N_CUS) ? lh : level_end;
    }
    return p;
}

void orbx_launch_octree(hipStream_t s, const OrbxLevels *d_levels, const OrbxLevels &levels, const OrbxBuffers &b,
                        int n_frames, size_t sort_lds_bytes, int level_begin, int level_end, int n_cus)
{
    if (level_end <= level_begin) return;
    const OrbxOctPlan p = orbx_octree_plan(levels, n_frames, level_begin, level_end, n_cus);
    const int lds = (int)p.lds_bytes;
    dim3 grid(level_end - level_begin, n_frames);
    switch (p.kind) {
    case 0:
        (void)orbx_lds_opt_in(reinterpret_cast<const void *>(oct_batch::k_octree_lds), p.lds_bytes); // per device; a refusal shows as the launch error the caller checks
        hipLaunchKernelGGL(oct_batch::k_octree_lds, grid, dim3(ORBX_OCT_THREADS_BATCH), p.lds_bytes + dev_pad(oct_batch::k_octree_lds, 2, p.lds_bytes), s,
                           d_levels, b, level_begin, lds);
        break;
    case 2:
        (void)orbx_lds_opt_in(reinterpret_cast<const void *>(oct_huge::k_octree_lds), p.huge_bytes);
        hipLaunchKernelGGL(oct_huge::k_octree_lds, dim3(p.huge_end - level_begin, n_frames), dim3(1024), p.huge_bytes, s, d_levels, b,
                           level_begin, (int)p.huge_bytes);
        if (p.huge_end == level_end) break;
        grid = dim3(level_end - p.huge_end, n_frames);
        level_begin = p.huge_end;
        [[fallthrough]];
    case 1:
        (void)orbx_lds_opt_in(reinterpret_cast<const void *>(oct_wide::k_octree_lds), p.lds_bytes);
        hipLaunchKernelGGL(oct_wide::k_octree_lds, grid, dim3(ORBX_OCT_THREADS), p.lds_bytes, s, d_levels, b, level_begin, lds);
        break;
    default:
        // quotas too large for the LDS-resident list: same algorithm with the list in global scratch
        hipLaunchKernelGGL(oct_wide::k_octree, grid, dim3(ORBX_OCT_THREADS), sort_lds_bytes, s, d_levels, b, level_begin);
    }
}

// ---------------------------------------------------------------------------------------------
// Orientation: intensity centroid on the raw level (:18-42), OR_LANES lanes per keypoint, OR_KP keypoints
// per workgroup.  A lane takes OR_TASKS (patch row, left/right half) tasks: one unaligned 128-bit load each,
// bytes outside the circular patch masked off, v_dot4_u32_u8 for sum(|u|*I) and v_sad_u8 for sum(I).
// After a shuffle reduction inside each group the moments go through LDS to OR_KP lanes that evaluate fastAtan2 and
// the double-precision sin/cos once per keypoint (one lane per keypoint instead of one wave).
// ---------------------------------------------------------------------------------------------
#define OR_LANES 8                 // lanes per keypoint
#define OR_KP (256 / OR_LANES)     // keypoints per workgroup
#define OR_TASKS (64 / OR_LANES)   // (row, half) tasks per lane
struct OrientLevels { // what k_orient needs of every level, passed by value so that it sits in the kernel-argument segment
    int kp_off[ORBX_MAX_LEVELS], pitch[ORBX_MAX_LEVELS];
    unsigned long long raw_off[ORBX_MAX_LEVELS];
};
#ifdef OR_NUM_VGPR // experiment (tools/build_variant.sh): a register ceiling, so that more waves fit beside a co-resident kernel
#define OR_VGPR_ATTR __attribute__((amdgpu_num_vgpr(OR_NUM_VGPR)))
#else
#define OR_VGPR_ATTR
#endif
template <bool WITH_ANGLE> // true (calls with a few frames): the lane that holds the moments also does k_angle's work -- one launch less
__global__ __launch_bounds__(256) OR_VGPR_ATTR void k_orient(const uint8_t *__restrict__ l0, size_t l0_fs, int l0_pitch,
                                                const OrbxLevels *__restrict__ levels, OrientLevels tab, OrbxBuffers b,
                                                const int *__restrict__ u_max, int per_frame, int n_frames,
                                                const uint4 *__restrict__ items, int item_levels)
{
    // byte masks of the circular patch, one 16-byte row per (row, half) task: they depend on the task only, so they are
    // built once per workgroup (thread = (task, dword)) instead of per key point
    __shared__ __align__(16) uint32_t s_mask[64][4];
    int frame, blk;
    if (!xcd_remap(per_frame, n_frames, &frame, &blk)) return;
    // (s_setprio 3 here and / or in the batch quadtree, so that they issue ahead of the co-resident match: 2.23-2.26 ms per step
    // against 2.19-2.20; profiles/NEGATIVES.md)
    const int tid = threadIdx.x, sub = tid & (OR_LANES - 1), grp = tid / OR_LANES;
    const int L = levels->n_levels, kc = levels->kcap_total;
    const int *cnts = b.sel_count + frame * ORBX_MAX_LEVELS;
    if (blk == 0 && tid == 0) { // key points of the levels before each level: the descriptor kernel's output index starts there
        int before = 0;
        for (int l = 0; l < L; ++l) { b.sel_prefix[frame * ORBX_MAX_LEVELS + l] = before; before += cnts[l]; }
    }
    const int slot = blk * OR_KP + grp;
    // Everything this thread needs before the patch loads is requested at once and without a branch -- the mask-table
    // entry's u_max, the key-point record (slot clamped), and through the scalar cache the level table and the per-level
    // counts, from which level, pitch and offset are SELECTED rather than gathered: one memory round trip, not four.
    const int mt = tid >> 2, mw = tid & 3, mv = min(mt >> 1, 2 * ORBX_HALF_PATCH) - ORBX_HALF_PATCH;
    const int md_raw = u_max[mv < 0 ? -mv : mv];
    uint2 rec = b.sel[(size_t)frame * kc + min(slot, kc - 1)];
    // levels below item_levels: slot s is the s-th key point of its level in k_desc_bins' (block, trip) order -- neighbours in
    // the image are neighbours in the list, so a workgroup's patches overlap in the L1 -- and the moments go back by that index
    const uint32_t item_xy = items ? items[2 * ((size_t)frame * kc + min(slot, kc - 1))].x : 0u;
    int level = 0, rpitch = l0_pitch, cnt = cnts[0], kp_off = 0;
    size_t raw_off = 0;
#pragma unroll
    for (int l = 1; l < ORBX_MAX_LEVELS; ++l) { // fixed trip count: the table comes in with the kernel arguments, wide scalar loads
        const bool ge = l < L && slot >= tab.kp_off[l];
        level += ge;
        rpitch = ge ? tab.pitch[l] : rpitch;
        raw_off = ge ? (size_t)tab.raw_off[l] : raw_off;
        kp_off = ge ? tab.kp_off[l] : kp_off;
        cnt = ge ? cnts[l] : cnt;
    }
    const bool live = slot < kc && (slot - kp_off) < cnt;
    if (level < item_levels) rec.x = item_xy;
    {
        const int d = mt < 2 * (2 * ORBX_HALF_PATCH + 1) ? md_raw : -1;
        // right half keeps bytes idx <= d (idx = 4w + byte), left half keeps idx >= 16 - d
        const int n_lo = min(max(d + 1 - 4 * mw, 0), 4);        // kept low bytes (right half)
        const int n_hi = min(max(4 * mw + 4 - (16 - d), 0), 4); // kept high bytes (left half)
        const uint32_t m_lo = n_lo >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n_lo)) - 1u);
        const uint32_t m_hi = n_hi <= 0 ? 0u : (0xFFFFFFFFu << (8 * (4 - n_hi)));
        s_mask[mt][mw] = (mt & 1) ? m_lo : m_hi;
    }
    __syncthreads();
    int m10 = 0, m01 = 0;
    if (live) {
        const int x = rec.x & 0xFFFF, y = rec.x >> 16;
        const uint8_t *raw = level == 0 ? l0 + (size_t)frame * l0_fs : b.img_arena + (size_t)frame * b.img_frame_stride + raw_off;
        struct __attribute__((packed, aligned(1))) U128 { uint32_t w[4]; };
        U128 q[OR_TASKS];
        uint4 mk[OR_TASKS];
        static_assert((OR_LANES & 1) == 0, "a lane's tasks all lie in the same half");
        const int half = sub & 1; // task t = sub + OR_LANES*k: row v = t/2 - 15, half = t & 1 (tasks 62, 63 are idle: mask 0)
#pragma unroll
        for (int k = 0; k < OR_TASKS; ++k) {
            const int t = sub + OR_LANES * k, v = min(t >> 1, 2 * ORBX_HALF_PATCH) - ORBX_HALF_PATCH;
            q[k] = *reinterpret_cast<const U128 *>(raw + (size_t)(y + v) * rpitch + x + (half ? 0 : -16));
            mk[k] = *reinterpret_cast<const uint4 *>(s_mask[t]);
        }
        // u-weights of the 16 bytes: the right half holds u = 0..15, the left half u = -16..-1 (as 16 - idx, negated below)
        uint32_t wt[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t wr = (uint32_t)(4 * w) | ((uint32_t)(4 * w + 1) << 8) | ((uint32_t)(4 * w + 2) << 16) | ((uint32_t)(4 * w + 3) << 24);
            const uint32_t wl = (uint32_t)(16 - 4 * w) | ((uint32_t)(15 - 4 * w) << 8) | ((uint32_t)(14 - 4 * w) << 16) | ((uint32_t)(13 - 4 * w) << 24);
            wt[w] = half ? wr : wl;
        }
        uint32_t wsum = 0;
#pragma unroll
        for (int k = 0; k < OR_TASKS; ++k) {
            const int v = min((sub + OR_LANES * k) >> 1, 2 * ORBX_HALF_PATCH) - ORBX_HALF_PATCH;
            const uint32_t m4[4] = {mk[k].x, mk[k].y, mk[k].z, mk[k].w};
            uint32_t ssum = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t wd = q[k].w[w] & m4[w];
                wsum = __builtin_amdgcn_udot4(wd, wt[w], wsum, false);
                ssum = __builtin_amdgcn_sad_u8(wd, 0u, ssum);
            }
            m01 += v * (int)ssum;
        }
        m10 = half ? (int)wsum : -(int)wsum;
    }
#pragma unroll
    for (int o = OR_LANES / 2; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    // the moments leave as they are; k_angle turns them into (angle, cos, sin) with one thread per key point -- done here
    // the long double-precision chain ran on one quarter-filled wave per workgroup while the other three waited
    if (sub == 0 && slot < levels->kcap_total) {
        float4 o = make_float4(__int_as_float(m10), __int_as_float(m01), 0.f, 0.f);
        if (WITH_ANGLE) {
            o.x = orb_fast_atan2((float)m01, (float)m10);
            orb_sincos_deg(o.x, &o.y, &o.z);
        }
        b.kp_ang[(size_t)frame * levels->kcap_total + slot] = o;
    }
}

// IC_Angle's last line (reference :41) and the sine / cosine of computeOrbDescriptor (:54), one thread per key-point slot
__global__ __launch_bounds__(256) void k_angle(const OrbxLevels *__restrict__ levels, OrbxBuffers b, int n_frames, float4 *__restrict__ items,
                                               int item_levels)
{
    const int kc = levels->kcap_total;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)kc * n_frames) return;
    const float4 m = b.kp_ang[i];
    const float ang = orb_fast_atan2((float)__float_as_int(m.y), (float)__float_as_int(m.x));
    float cs, sn;
    orb_sincos_deg(ang, &cs, &sn);
    if (items && b.slot_level[i % (size_t)kc] < item_levels) { // k_blur_desc's record: (xy, out_idx, angle, cos | sin, response, -, -)
        float4 lo = items[2 * i], hi = items[2 * i + 1];
        lo.z = ang; lo.w = cs; hi.x = sn;
        items[2 * i] = lo; items[2 * i + 1] = hi;
    } else {
        b.kp_ang[i] = make_float4(ang, cs, sn, 0.f);
    }
}

// parity tap: the device's (cos, sin) of the descriptor rotation on caller-supplied angles (orbx_tap_sincos)
__global__ __launch_bounds__(256) void k_tap_sincos(const float *__restrict__ ang, int n, float2 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float cs, sn;
    orb_sincos_deg(ang[i], &cs, &sn);
    out[i] = make_float2(cs, sn);
}
void launch_tap_sincos(const float *d_ang, int n, float2 *d_out, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_tap_sincos, dim3((n + 255) / 256), dim3(256), 0, st, d_ang, n, d_out);
}

// ---------------------------------------------------------------------------------------------
// Steered BRIEF on the blurred level (:50-97) + output record (:507-546, :626-632).  One wave64 per
// keypoint, four keypoints per workgroup; lane l owns descriptor bits l, l+64, l+128, l+192 and four
// ballots give the 32 bytes.
// ---------------------------------------------------------------------------------------------
#define DP_W 64 // patch pitch: four ALIGNED 16-byte loads per row starting at floor16(x-19)
#define DP_H 37
#ifndef DP_K
#define DP_K 2  // keypoints per wave: their record / patch loads are all issued before the first use
#endif
struct DescLv { // what k_orient_desc needs of one level: 32 bytes = ONE scalar load from the kernel-argument segment
    int kp_off, pitch;
    float scale;
    int pad;
    unsigned long long blur_off, pad2;
};
struct DescTab { int n_levels, kcap_total, pad[2]; DescLv lv[ORBX_MAX_LEVELS]; };
// ROUNDS > 1 (an experiment that lost, no longer instantiated): a wave takes ROUNDS consecutive pairs of key points and, while it
// samples pair i from LDS, has the patch loads of pair i + 1 in flight, to hide the staging's L2 round trip inside the wave --
// the kernel issues VALU in only a tenth of its wave cycles.  Measured per 512 frames: 1.19 ms (2 rounds) and 1.50 ms (4)
// against 0.62 for one round: every round's key-point records come through a chain of dependent scalar loads, and scalar
// loads share the counter (lgkmcnt, waited to zero) with the LDS gathers of the round being sampled.  One round per wave,
// written this way (no workgroup barrier, 50 VGPRs instead of 65), is what runs.
// Also measured and dropped: the patch rows global -> LDS directly (global_load_lds_dwordx4; the layout is lane-linear, item q at
// byte 16 q, so it fits: no staging registers, no ds_write_b128) -- bit-identical, 45 VGPRs, and 0.40 ms against 0.345; and
// 1 / 3 / 4 key points per wave instead of 2 (-DDP_K): 0.41 / 0.38 / 0.41 ms.
template <int ROUNDS>
__global__ __launch_bounds__(256) void k_orient_desc(DescTab tab, OrbxBuffers b,
                                                     orbx_kp *__restrict__ out_kp, uint8_t *__restrict__ out_desc,
                                                     int cap, int32_t *__restrict__ out_n, int per_frame, int n_frames, int level_min)
{
    // the 37x37 neighbourhood of each keypoint is staged row by row (coalesced, ~45 cache lines) and the
    // 512 rotated samples are byte gathers from LDS instead of ~250 scattered cache-line touches
    __shared__ __align__(16) uint8_t patch[4][DP_K][DP_H * DP_W];
    int frame, blk;
    if (!xcd_remap(per_frame, n_frames, &frame, &blk)) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = tab.n_levels;
    const int *cnts = b.sel_count + frame * ORBX_MAX_LEVELS;
    if (blk == 0 && threadIdx.x == 0) {
        int tot = 0;
        for (int l = 0; l < L; ++l) tot += cnts[l];
        out_n[frame] = tot;
    }
    // this lane's four sampling pairs (the same for every keypoint); loaded first so that their latency
    // overlaps the record / patch loads below
    uint32_t pat[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pat[j] = reinterpret_cast<const uint32_t *>(c_pattern)[lane + 64 * j];
    constexpr int DP_ITEMS = DP_H * (DP_W / 16), DP_IT = (DP_ITEMS + 63) / 64;
    struct Pair { // everything about one round's two key points (wave-uniform)
        bool live[DP_K];
        int level[DP_K], out_idx[DP_K];
        uint2 rec[DP_K];
        float4 ang[DP_K];
    };
    // Everything about the wave's two keypoints is wave-uniform: the slot is forced into an SGPR so that the level table,
    // the per-level counts and the keypoint records come through the scalar cache instead of a chain of dependent vector
    // loads, each with its own s_waitcnt.  All six 128-bit loads of a lane (three per keypoint; the third covers items
    // 128..147, clamped for the other lanes) are in flight together.
    auto fetch = [&](int round, Pair &pr, uint4 (&stage)[DP_K][DP_IT]) {
        const int slot0 = __builtin_amdgcn_readfirstlane(((blk * 4 + wv) * ROUNDS + round) * DP_K);
#pragma unroll
        for (int k = 0; k < DP_K; ++k) {
            const int slot = slot0 + k;
            bool live = slot < tab.kcap_total;
            // level of the slot from a per-geometry table, key points of the levels before from k_orient's prefix: a handful
            // of scalar loads where two loops over the levels ran on the scalar unit
            const int lvl = live ? b.slot_level[slot] : 0;
            pr.level[k] = lvl;
            const DescLv lv = tab.lv[lvl]; // (a slot past the last one is given level 0; an empty slot of a level stages the corner of its own blurred level)
            const int i = slot - lv.kp_off;
            live = live && i < cnts[lvl] && lvl >= level_min; // (levels below level_min are described by k_blur_desc)
            const int oi = i + b.sel_prefix[frame * ORBX_MAX_LEVELS + lvl];
            pr.out_idx[k] = oi;
            live = live && oi < cap;
            pr.live[k] = live;
            pr.rec[k] = make_uint2(0, 0);
            pr.ang[k] = make_float4(0.f, 1.f, 0.f, 0.f);
            // a slot without a keypoint stages the top-left corner of its blurred level (valid memory, never sampled), so
            // that the loads below need no branch
            int x = 19, y = 18;
            if (live) {
                pr.rec[k] = b.sel[(size_t)frame * tab.kcap_total + slot];
                pr.ang[k] = b.kp_ang[(size_t)frame * tab.kcap_total + slot];
                x = pr.rec[k].x & 0xFFFF;
                y = pr.rec[k].x >> 16;
            }
            // arena rows are 64-byte aligned, so one shift (x-19)&15 serves the whole patch: the rows go to LDS
            // as they are (148 aligned 128-bit loads per keypoint) and only the centre index moves
            const uint8_t *corner = b.img_arena + (size_t)frame * b.img_frame_stride + lv.blur_off + (size_t)(y - 18) * lv.pitch + ((x - 19) & ~15);
#pragma unroll
            for (int it = 0; it < DP_IT; ++it) {
                const int q = min(lane + 64 * it, DP_ITEMS - 1), r = q >> 2, dc = q & 3;
                stage[k][it] = *reinterpret_cast<const uint4 *>(corner + (size_t)r * lv.pitch + 16 * dc);
            }
        }
    };
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        px0[j] = (float)(int8_t)(pat[j] & 255); py0[j] = (float)(int8_t)((pat[j] >> 8) & 255);
        px1[j] = (float)(int8_t)((pat[j] >> 16) & 255); py1[j] = (float)(int8_t)(pat[j] >> 24);
    }
    Pair cur;
    uint4 stage[DP_K][DP_IT];
    fetch(0, cur, stage);
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
#pragma unroll
        for (int k = 0; k < DP_K; ++k)
#pragma unroll
            for (int it = 0; it < DP_IT; ++it) {
                const int q = min(lane + 64 * it, DP_ITEMS - 1), r = q >> 2, dc = q & 3;
                *reinterpret_cast<uint4 *>(&patch[wv][k][r * DP_W + 16 * dc]) = stage[k][it];
            }
        // a wave stages and samples only its own two patches, and a wave's LDS instructions execute in issue order: all that
        // is needed between the writes above and the gathers below (and between those and the next round's writes) is that
        // the compiler keeps that order -- no workgroup barrier
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const Pair pr = cur;
        if (round + 1 < ROUNDS) fetch(round + 1, cur, stage); // in flight while this round's pair is sampled
#pragma unroll
        for (int k = 0; k < DP_K; ++k) {
            if (!pr.live[k]) continue;
            const float lscale = tab.lv[pr.level[k]].scale;
            const float a = pr.ang[k].y, bb = pr.ang[k].z;
            // cvRound (ties to even) without v_rndne + v_cvt: adding 1.5 * 2^23 makes the float add itself round to the
            // nearest integer (|value| <= 19 here), the integer is then the low mantissa bits.  The constant's bit pattern M
            // is not subtracted per sample: (M + r) * 64 + (M + c) = r * 64 + c + 65 M, and 65 M is folded into the patch
            // offset (unsigned arithmetic, wraps harmlessly).
            static_assert(DP_W == 64, "row stride folded into a shift");
            const float MAGIC = 12582912.f;
            const uint32_t M = 0x4B400000u;
            const uint8_t *pbytes = &patch[0][0][0];
            const uint32_t centre = (uint32_t)((wv * DP_K + k) * (DP_H * DP_W) + 18 * DP_W + 19 + (((int)(pr.rec[k].x & 0xFFFF) - 19) & 15)) -
                                    65u * M;
            u64 bits[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r0 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px0[j], bb), ORB_FMUL(py0[j], a)), MAGIC));
                const uint32_t c0 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px0[j], a), ORB_FMUL(py0[j], bb)), MAGIC));
                const uint32_t r1 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px1[j], bb), ORB_FMUL(py1[j], a)), MAGIC));
                const uint32_t c1 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px1[j], a), ORB_FMUL(py1[j], bb)), MAGIC));
                const int t0 = pbytes[(r0 << 6) + c0 + centre], t1 = pbytes[(r1 << 6) + c1 + centre];
                bits[j] = __ballot(t0 < t1);
            }
            if (lane < 4) {
                const u64 w = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
                *reinterpret_cast<u64 *>(out_desc + ((size_t)frame * cap + pr.out_idx[k]) * 32 + 8 * lane) = w;
            }
            if (lane == 0) {
                const int x = pr.rec[k].x & 0xFFFF, y = pr.rec[k].x >> 16;
                orbx_kp kp;
                float fx = (float)x, fy = (float)y;
                if (pr.level[k] != 0) { fx = ORB_FMUL(fx, lscale); fy = ORB_FMUL(fy, lscale); }
                kp.x = fx; kp.y = fy; kp.size = lscale; kp.angle = pr.ang[k].x; kp.response = (float)pr.rec[k].y;
                kp.octave = pr.level[k]; kp.class_id = -1;
                out_kp[(size_t)frame * cap + pr.out_idx[k]] = kp;
            }
        }
        asm volatile("" ::: "memory"); // the next round's LDS writes stay behind this round's gathers
    }
}

void orbx_launch_orient_desc(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                             const OrbxLevels &levels, const OrbxBuffers &b, const int *u_max, orbx_kp *out_kp,
                             uint8_t *out_desc, int cap, int32_t *out_n, int n_frames, hipEvent_t blur_done, int desc_level_min,
                             hipEvent_t after_orient, void *d_items, hipEvent_t desc_open)
{
    const int pf_o = (levels.kcap_total + OR_KP - 1) / OR_KP;
    OrientLevels tab;
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) {
        tab.kp_off[l] = l < levels.n_levels ? levels.lv[l].kp_off : 0;
        tab.pitch[l] = l < levels.n_levels ? levels.lv[l].pitch : 0;
        tab.raw_off[l] = l < levels.n_levels ? (unsigned long long)levels.lv[l].raw_off : 0ull;
    }
    // d_items (levels [0, desc_level_min) are described by k_blur_desc): key points of those levels are taken in k_desc_bins' order
    // and their angles go into the item records
    const int item_levels = d_items ? desc_level_min : 0;
    if (n_frames < 24 && !item_levels) { // latency-bound: one launch
        hipLaunchKernelGGL(k_orient<true>, dim3(orbx_xcd_grid(pf_o, n_frames)), dim3(256), 0, s, l0, l0_fs, l0_pitch, d_levels, tab,
                           b, u_max, pf_o, n_frames, nullptr, 0);
    } else {
        hipLaunchKernelGGL(k_orient<false>, dim3(orbx_xcd_grid(pf_o, n_frames)), dim3(256), dev_pad(k_orient<false>, 3, 0), s, l0, l0_fs, l0_pitch, d_levels, tab,
                           b, u_max, pf_o, n_frames, reinterpret_cast<const uint4 *>(item_levels ? d_items : nullptr), item_levels);
        hipLaunchKernelGGL(k_angle, dim3((unsigned)(((size_t)levels.kcap_total * n_frames + 255) / 256)), dim3(256), 0, s, d_levels,
                           b, n_frames, reinterpret_cast<float4 *>(item_levels ? d_items : nullptr), item_levels);
    }
    if (after_orient) (void)hipEventRecord(after_orient, s); // stage timing: orientation | descriptors
    if (desc_level_min < levels.n_levels && blur_done) (void)hipStreamWaitEvent(s, blur_done, 0); // the blurred levels come from a side stream
    // in-step stage timing: the descriptor bracket opens BEHIND the wait for the side stream's blur, so that blur time the
    // wait exposes is not booked on the descriptors
    if (desc_open) (void)hipEventRecord(desc_open, s);
    if (desc_level_min >= levels.n_levels) return; // every level is described by k_blur_desc (orbx_launch_desc_fused)
    DescTab dt;
    dt.n_levels = levels.n_levels; dt.kcap_total = levels.kcap_total; dt.pad[0] = dt.pad[1] = 0;
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) {
        const bool in = l < levels.n_levels;
        dt.lv[l].kp_off = in ? levels.lv[l].kp_off : 0;
        dt.lv[l].pitch = in ? levels.lv[l].pitch : 0;
        dt.lv[l].scale = in ? levels.lv[l].scale : 1.f;
        dt.lv[l].pad = 0; dt.lv[l].pad2 = 0;
        dt.lv[l].blur_off = in ? (unsigned long long)levels.lv[l].blur_off : 0ull;
    }
    const int per_wg = 4 * DP_K; // (the instantiations with 2 / 4 rounds per wave lost -- see the kernel -- and are not built)
    const int pf_d = (levels.kcap_total + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(k_orient_desc<1>, dim3(orbx_xcd_grid(pf_d, n_frames)), dim3(256), 0, s, dt, b, out_kp, out_desc, cap, out_n, pf_d, n_frames,
                       desc_level_min);
}

// ---------------------------------------------------------------------------------------------
// k_blur_desc: the 7x7 Gaussian and the rBRIEF descriptors in ONE pass over the raw level.
//
// The reference blurs a copy of every level only to sample descriptors from it (ORBExtractor.cpp:527-532: the blurred Mat
// has no other reader).  As two kernels that is a full extra pass -- the blurred pyramid written to HBM and read back, one
// 37 x 64-byte patch per key point.  Here a workgroup walks DOWN a block of BD_COLS blurred columns the way k_blur_mfma does
// (same matrix products, same band tables for the rows) but keeps the blurred rows in an LDS ring of four 32-row trips
// instead of storing them: after trip ty the ring holds rows 32 ty - 96 .. 32 ty + 31, and every key point of the block whose
// window ends inside trip ty (y + 18 <= 32 ty + 31, rows from 32 ty - 36 on) is sampled from the ring right there.  Nothing
// blurred ever reaches memory, no patch is staged, and the raw level is read once (plus the 36 / 128 column overlap of the
// blocks: a block serves the key points of its middle BD_CW = 92 columns).  Key points come bucketed by (block, trip) from
// k_desc_bins, one 32-byte record each (position, output slot, angle, cos, sin, response) through the scalar cache.
// Same arithmetic as k_blur_mfma + k_orient_desc, bit for bit (tests: both paths against the oracle, and against each other).
// ---------------------------------------------------------------------------------------------
#ifndef BD_WAVES
#define BD_WAVES 4                       // waves per workgroup = 32-column tile columns of a block (4 / 6 / 8 measured)
#endif
#define BD_COLS (32 * BD_WAVES)          // blurred columns of a workgroup
#define BD_T (64 * BD_WAVES)             // threads
#ifndef BD_ALIGN16
#define BD_ALIGN16 0
#endif
// source bytes staged per row, in 16-byte chunks: 16 either side -- and, with BD_ALIGN16, one chunk more so that the staged row
// can start at a 16-byte boundary of the level (arena rows are 64-byte aligned: every load then is one aligned 16-byte request)
#define BD_SRC_W (BD_COLS + 32 + 16 * BD_ALIGN16)
#define BD_CH (BD_SRC_W / 16)
#define BD_SRC_P (BD_SRC_W + 16)         // LDS pitch of a staged source row
static_assert(32 * BD_CH <= 2 * BD_T, "a thread stages at most two chunks");
#define BD_HALO 18                       // a rotated pattern point lies within +-18 px of its key point (ORBX_EDGE - 1)
#define BD_CW (BD_COLS - 2 * BD_HALO)    // key-point columns a workgroup serves
#define BD_RING 128                      // blurred rows held in LDS: four trips
#define BD_RP (BD_COLS + 4)              // ring pitch (an odd number of dwords: the 32 rows of a tile column land in different banks)
static_assert(BD_RING == 128 && BD_HALO + 1 == ORBX_EDGE, "ring rows are addressed with & 127; the halo is the descriptor's reach");
struct __attribute__((aligned(4))) BdBlock { uint16_t level, bi; };
struct __attribute__((aligned(32))) BdItem { // one key point as k_blur_desc takes it: 32 bytes = one scalar load
    uint32_t xy;      // x | y << 16, level coordinates
    int32_t out_idx;  // record index inside the frame's output (-1: beyond the caller's capacity)
    float angle, cs, sn, response;
    int32_t pad[2];
};
// first blurred column of block bi, the block of a key-point column, the trip that completes a key point's window
__host__ __device__ __forceinline__ int bd_first_col(int bi) { return bi * BD_CW; }
// first staged source column of a block whose first blurred column is B0; `limit` = readable bytes of a row
__host__ __device__ __forceinline__ int bd_origin(int B0, int limit)
{
    const int want = BD_ALIGN16 ? ((B0 - 16) & ~15) : B0 - 16;
    return min(max(want, 0), limit - BD_SRC_W);
}
__host__ __device__ __forceinline__ int bd_block_of(int x) { return (x - BD_HALO) / BD_CW; }
__host__ __device__ __forceinline__ int bd_trip_of(int y) { return (y + BD_HALO) >> 5; }
// Each (block, trip) bucket has an EARLY and a LATE part.  A key point may wait one trip (load balance, below) unless its
// window starts in the first four rows its trip can still see: the trip after would find those rows, four ring blocks back,
// being overwritten by the waves that are already one trip ahead.
__host__ __device__ __forceinline__ int bd_sub_bucket(int x, int y, int n_ty)
{
    return 2 * (bd_block_of(x) * n_ty + bd_trip_of(y)) + (((y + BD_HALO) & 31) >= 4 ? 1 : 0);
}

// Buckets the selected key points of one (frame, level) by (block, trip) -- a counting sort in LDS -- and writes their item
// records (position, output slot, response; k_orient / k_angle, which run over the items in this order, add the angle);
// also the frame's total count (what k_orient_desc's first workgroup does on the two-kernel path).
__global__ __launch_bounds__(256) void k_desc_bins(const OrbxLevels *__restrict__ levels, BdLevels lv, OrbxBuffers b,
                                                   int *__restrict__ bk_start, int bk_stride, BdItem *__restrict__ items, int cap,
                                                   int32_t *__restrict__ out_n)
{
    extern __shared__ int s_bins[]; // [nb + 1] starts, [nb] cursors
    __shared__ int s_part[256];
    const int level = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const int kc = levels->kcap_total, L = levels->n_levels;
    const int *cnts = b.sel_count + frame * ORBX_MAX_LEVELS;
    if (level == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < L; ++l) tot += cnts[l];
        out_n[frame] = tot;
    }
    const int n_ty = lv.n_ty[level], nb = 2 * lv.n_bx[level] * n_ty, kp_off = levels->lv[level].kp_off;
    const int n = min(cnts[level], levels->lv[level].kcap);
    int *start = s_bins, *cursor = s_bins + nb + 1;
    for (int i = tid; i < nb; i += 256) cursor[i] = 0;
    __syncthreads();
    const uint2 *sel = b.sel + (size_t)frame * kc + kp_off;
    for (int i = tid; i < n; i += 256) {
        const uint32_t xy = sel[i].x;
        atomicAdd(&cursor[bd_sub_bucket((int)(xy & 0xFFFF), (int)(xy >> 16), n_ty)], 1);
    }
    __syncthreads();
    // exclusive scan of the nb counts: a contiguous run per thread, then the 256 run totals
    const int per = (nb + 255) / 256;
    int s = 0;
    for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < nb) s += cursor[i]; }
    s_part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int base = s_part[tid] - s;
    for (int k = 0; k < per; ++k) {
        const int i = tid * per + k;
        if (i < nb) { const int c = cursor[i]; start[i] = base; base += c; }
    }
    __syncthreads();
    int *gs = bk_start + (size_t)frame * bk_stride + lv.bucket_base[level];
    for (int i = tid; i < nb; i += 256) { gs[i] = kp_off + start[i]; cursor[i] = start[i]; }
    if (tid == 0) gs[nb] = kp_off + n;
    __syncthreads();
    int before = 0; // key points of the levels below: this level's records start there (k_orient writes the same prefix for k_orient_desc)
    for (int l = 0; l < level; ++l) before += cnts[l];
    for (int i = tid; i < n; i += 256) {
        const uint2 rec = sel[i];
        const int pos = atomicAdd(&cursor[bd_sub_bucket((int)(rec.x & 0xFFFF), (int)(rec.x >> 16), n_ty)], 1);
        BdItem it;
        it.xy = rec.x; it.out_idx = before + i < cap ? before + i : -1;
        it.angle = 0.f; it.cs = 1.f; it.sn = 0.f; // k_angle fills these in (the orientation runs over the items in this order)
        it.response = (float)rec.y; it.pad[0] = it.pad[1] = 0;
        items[(size_t)frame * kc + kp_off + pos] = it;
    }
}

// Five workgroups per CU is what the kernel's 28 KB of LDS allow; left alone the compiler takes 109 VGPRs (four waves per
// SIMD).  Held to five waves it fits in 94 without spilling: 0.565 -> 0.51 ms per 512 frames (six: no gain, LDS-bound).
#ifndef BD_WAVES_PER_EU
#define BD_WAVES_PER_EU 5
#endif
#define BD_OCC __attribute__((amdgpu_waves_per_eu(BD_WAVES_PER_EU, BD_WAVES_PER_EU)))
template <int K_SUM>
__global__ __launch_bounds__(BD_T) BD_OCC void k_blur_desc(FastSrc src, BdLevels lv, const BdBlock *__restrict__ blocks,
                                                   const uint4 *__restrict__ band_h, const uint4 *__restrict__ band_v,
                                                   const int *__restrict__ bk_start, int bk_stride, const BdItem *__restrict__ items,
                                                   int kcap_total, orbx_kp *__restrict__ out_kp, uint8_t *__restrict__ out_desc, int cap,
                                                   int n_blocks, int n_frames)
{
    constexpr int hbias = 128 * K_SUM - 32768;
    constexpr uint32_t vbias = 128u * K_SUM * 257u + 32768u;
    constexpr bool clamp255 = K_SUM != 256;
    __shared__ __align__(16) uint8_t s_src[2][32 * BD_SRC_P];
    __shared__ __align__(16) uint8_t s_ring[BD_RING * BD_RP];
    int frame, bid;
    if (!xcd_remap(n_blocks, n_frames, &frame, &bid)) return;
    const BdBlock bk = blocks[bid];
    const int level = bk.level, tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = lv.w[level], h = lv.h[level], pitch = src.pitch[level], n_ty = lv.n_ty[level];
    const int limit = level == 0 ? w : pitch;
    const int B0 = bd_first_col((int)bk.bi);
    const int xo = bd_origin(B0, limit);
    const int cw = min(max(B0 + 32 * wave - 16, xo), xo + BD_SRC_W - 64); // this wave's 64-column source window (two K-steps)
    (void)w;
    const uint8_t *S = src.base[level] + (size_t)frame * src.frame_stride[level];
    const uint4 *bh = band_h + lv.bh_off[level] + (size_t)((int)bk.bi * BD_WAVES + wave) * 128;
    const bl_v4i bh0 = __builtin_bit_cast(bl_v4i, bh[lane]), bh1 = __builtin_bit_cast(bl_v4i, bh[64 + lane]);
    const int a_off = n * BD_SRC_P + (cw - xo) + 16 * hh;
    // this lane's four sampling pairs (the same for every key point)
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t pat = reinterpret_cast<const uint32_t *>(c_pattern)[lane + 64 * j];
        px0[j] = (float)(int8_t)(pat & 255); py0[j] = (float)(int8_t)((pat >> 8) & 255);
        px1[j] = (float)(int8_t)((pat >> 16) & 255); py1[j] = (float)(int8_t)(pat >> 24);
    }
    const float lscale = lv.scale[level];
    const int *bks = bk_start + (size_t)frame * bk_stride + lv.bucket_base[level] + 2 * (int)bk.bi * n_ty;
    const BdItem *its = items + (size_t)frame * kcap_total;

    const int c0r = tid / BD_CH, c0c = tid - BD_CH * c0r, c1r = (BD_T + tid) / BD_CH, c1c = (BD_T + tid) - BD_CH * c1r;
    const bool second = tid < 32 * BD_CH - BD_T;
    auto g_load = [&](int T, uint4 *u0, uint4 *u1) {
        const int r0 = min(max(32 * T - 3 + c0r, 0), h - 1), r1 = min(max(32 * T - 3 + min(c1r, 31), 0), h - 1);
        const UnalignedU4 a = *reinterpret_cast<const UnalignedU4 *>(S + (size_t)r0 * pitch + xo + 16 * c0c);
        const UnalignedU4 b2 = *reinterpret_cast<const UnalignedU4 *>(S + (size_t)r1 * pitch + xo + 16 * (second ? c1c : 0));
        *u0 = make_uint4(a.x, a.y, a.z, a.w);
        *u1 = make_uint4(b2.x, b2.y, b2.z, b2.w);
    };
    auto s_store = [&](int buf, const uint4 &u0, const uint4 &u1) {
        *reinterpret_cast<uint4 *>(&s_src[buf][c0r * BD_SRC_P + 16 * c0c]) = u0;
        if (second) *reinterpret_cast<uint4 *>(&s_src[buf][c1r * BD_SRC_P + 16 * c1c]) = u1;
    };
    auto h_tile = [&](int buf, bl_v4i *hi, bl_v4i *lo) { // as in k_blur_mfma
        bl_v4i a0 = reinterpret_cast<const UnalignedV4 *>(&s_src[buf][a_off])->v;
        bl_v4i a1 = reinterpret_cast<const UnalignedV4 *>(&s_src[buf][a_off + 32])->v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] ^= (int)0x80808080u; a1[k] ^= (int)0x80808080u; }
        const bl_v16i zero = {};
        bl_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bh0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bh1, acc, 0, 0, 0);
        if (hbias) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += hbias;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t p = (uint32_t)acc[4 * k], q = (uint32_t)acc[4 * k + 1], r2 = (uint32_t)acc[4 * k + 2], t = (uint32_t)acc[4 * k + 3];
            const uint32_t pq1 = __builtin_amdgcn_perm(q, p, 0x0c0c0501u), rt1 = __builtin_amdgcn_perm(t, r2, 0x0c0c0501u);
            const uint32_t pq0 = __builtin_amdgcn_perm(q, p, 0x0c0c0400u), rt0 = __builtin_amdgcn_perm(t, r2, 0x0c0c0400u);
            (*hi)[k] = (int)__builtin_amdgcn_perm(rt1, pq1, 0x05040100u);
            (*lo)[k] = (int)(__builtin_amdgcn_perm(rt0, pq0, 0x05040100u) ^ 0x80808080u);
        }
    };
    // one key point: 512 rotated samples from the ring, four ballots -> 32 bytes, and its output record (reference :50-97, :537-546)
    auto describe = [&](const BdItem &it) {
        if (it.out_idx < 0) return;
        const int x = (int)(it.xy & 0xFFFF), y = (int)(it.xy >> 16);
        float a = it.cs, bb = it.sn;
        // (the record comes through the scalar cache: as scalar operands cos and sin would put the 32 multiplications below into
        // the slow issue class -- v_and with a scalar operand 4.95 cycles against 3.2 / 2.7, profiles/r03_valu_ops3.txt -- two
        // moves make them vector operands)
        asm("" : "+v"(a), "+v"(bb));
        // cvRound through the float adder as in k_orient_desc: bits(v + 1.5 * 2^23) = M + round(v), M's low bits are zero
        const float MAGIC = 12582912.f;
        const uint32_t M = 0x4B400000u;
        u64 bits[4];
        // A window that does not run over the end of the ring (seven in ten) is addressed linearly: with r0 = M + dr, c0 = M + dc the
        // byte is at (top + 18 + dr) * RP + (x - B0) + dc = mad24(r0, RP, c0) + K, two operations per sample (the 24-bit
        // multiply sees 0x400000 + dr, and K takes that constant and M out again); the others wrap row by row with & 127.
        const int top = (y - BD_HALO) & (BD_RING - 1);
        if (top + 2 * BD_HALO < BD_RING) {
            uint32_t K = (uint32_t)((top + BD_HALO) * BD_RP + (x - B0)) - 0x400000u * (uint32_t)BD_RP - M;
            asm("" : "+v"(K)); // a vector register: a scalar operand halves the issue rate of the add below
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r0 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px0[j], bb), ORB_FMUL(py0[j], a)), MAGIC));
                const uint32_t c0 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px0[j], a), ORB_FMUL(py0[j], bb)), MAGIC));
                const uint32_t r1 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px1[j], bb), ORB_FMUL(py1[j], a)), MAGIC));
                const uint32_t c1 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px1[j], a), ORB_FMUL(py1[j], bb)), MAGIC));
                // (v_mad_u32_u24 + v_add instead of the v_mul_u32_u24 + v_add3 the compiler picks: 0.492 against 0.489 ms; the next
                // record requested a key point ahead: no change)
                const int t0 = s_ring[__umul24(r0, BD_RP) + c0 + K];
                const int t1 = s_ring[__umul24(r1, BD_RP) + c1 + K];
                bits[j] = __ballot(t0 < t1);
            }
        } else {
            uint32_t yv = (uint32_t)y, colbase = (uint32_t)(x - B0) - M;
            asm("" : "+v"(yv), "+v"(colbase));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r0 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px0[j], bb), ORB_FMUL(py0[j], a)), MAGIC));
                const uint32_t c0 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px0[j], a), ORB_FMUL(py0[j], bb)), MAGIC));
                const uint32_t r1 = __float_as_uint(ORB_FADD(ORB_FADD(ORB_FMUL(px1[j], bb), ORB_FMUL(py1[j], a)), MAGIC));
                const uint32_t c1 = __float_as_uint(ORB_FADD(ORB_FSUB(ORB_FMUL(px1[j], a), ORB_FMUL(py1[j], bb)), MAGIC));
                const int t0 = s_ring[((r0 + yv) & (BD_RING - 1)) * BD_RP + c0 + colbase];
                const int t1 = s_ring[((r1 + yv) & (BD_RING - 1)) * BD_RP + c1 + colbase];
                bits[j] = __ballot(t0 < t1);
            }
        }
        if (lane < 4) {
            const u64 wd = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
            *reinterpret_cast<u64 *>(out_desc + ((size_t)frame * cap + it.out_idx) * 32 + 8 * lane) = wd;
        }
        if (lane == 0) {
            orbx_kp kp;
            float fx = (float)x, fy = (float)y;
            if (level != 0) { fx = ORB_FMUL(fx, lscale); fy = ORB_FMUL(fy, lscale); }
            kp.x = fx; kp.y = fy; kp.size = lscale; kp.angle = it.angle; kp.response = it.response;
            kp.octave = level; kp.class_id = -1;
            out_kp[(size_t)frame * cap + it.out_idx] = kp;
        }
    };

    uint4 u0, u1;
    g_load(0, &u0, &u1);
    s_store(0, u0, u1);
    g_load(1, &u0, &u1);
    s_store(1, u0, u1);
    g_load(2, &u0, &u1);
    __syncthreads();
    bl_v4i hiP, loP, hiN, loN;
    h_tile(0, &hiP, &loP);
    __syncthreads();
    const uint4 *bv = band_v + lv.bv_off[level];
    int done = bks[0]; // items of this block described so far (the same value in every wave)
    for (int ty = 0; ty < n_ty; ++ty, bv += 128) {
        // Key points to describe in this trip.  The ring keeps a trip's rows for one more trip, so the LATE part of bucket ty may
        // wait until trip ty + 1: pending = what is due now (everything up to the early part of bucket ty) + the late part, and
        // the trip takes the largest multiple of the wave count out of it (at least what is due, everything in the last trip)
        // -- with four or five key points per bucket a fixed one-bucket-per-trip rule makes three waves wait for the one that
        // got two.  The buckets of a block are contiguous in the item array and every wave computes the same cursor, so the rule
        // costs no communication.  The wave takes items it_begin, it_begin + BD_WAVES, ...; the first record is requested now
        // and used after the barrier.
        const int due = bks[2 * ty + 1], avail = bks[2 * ty + 2];
        const int take = ty + 1 == n_ty ? avail - done : max(due - done, (avail - done) / BD_WAVES * BD_WAVES);
        const int it_begin = done + wave, it_end = done + take;
        done = it_end;
        BdItem first;
        first.out_idx = -1;
        if (it_begin < it_end) first = its[it_begin];
        const bl_v4i bv0 = __builtin_bit_cast(bl_v4i, bv[lane]), bv1 = __builtin_bit_cast(bl_v4i, bv[64 + lane]);
        s_store(ty & 1, u0, u1);
        g_load(ty + 3, &u0, &u1);
        h_tile((ty + 1) & 1, &hiN, &loN);
        const bl_v16i zero = {};
        bl_v16i ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiP, bv0, zero, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(hiN, bv1, ah, 0, 0, 0);
        bl_v16i al = __builtin_amdgcn_mfma_i32_32x32x32_i8(loP, bv0, zero, 0, 0, 0);
        al = __builtin_amdgcn_mfma_i32_32x32x32_i8(loN, bv1, al, 0, 0, 0);
        // blurred rows 32 ty .. 32 ty + 31 take ring rows ((ty & 3) << 5) ..: written by this trip, read by this trip's and the
        // next trip's key points, overwritten four trips on -- behind two barriers
        uint8_t *so = &s_ring[((ty & 3) << 5) * BD_RP];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = (uint32_t)(ah[4 * g + k] * 256 + al[4 * g + k]) + vbias;
                if (clamp255) v[k] = min(v[k], 0x00FFFFFFu);
            }
            const uint32_t a = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), b2 = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);
            *reinterpret_cast<uint32_t *>(&so[n * BD_RP + 32 * wave + 8 * g + 4 * hh]) = __builtin_amdgcn_perm(b2, a, 0x05040100u);
        }
        __syncthreads(); // the trip's blurred rows and the source of the next trip are complete
#ifndef BD_PROBE_NODESC // (timing probe: the blur part alone)
        if (it_begin < it_end) {
            describe(first);
            for (int k = it_begin + BD_WAVES; k < it_end; k += BD_WAVES) describe(its[k]);
        }
#endif
        hiP = hiN; loP = loN;
    }
}

// Block list, H-band tables and bucket layout of k_blur_desc for the levels k_blur_mfma could take (the V bands are shared
// with it).  Host side of the plan; `taps` as for orbx_build_blur_mfma.
void orbx_build_blur_desc(const OrbxLevels &levels, const int taps[7], const BlurMfmaLevels &mf, std::vector<uint16_t> &blocks,
                          std::vector<uint8_t> &band_h, BdLevels &out, int *n_fused_levels, int *bk_stride)
{
    blocks.clear(); band_h.clear();
    memset(&out, 0, sizeof out);
    // levels wide enough for a block's staged source row (and for k_blur_mfma, whose V bands are shared) and small enough for
    // k_desc_bins' bucket tables in 64 KB of LDS (two ints per (block, trip, early / late) bucket: beyond about 8000 x 4000 pixels
    // the level keeps the blur pass + k_orient_desc)
    auto buckets_fit = [&](const OrbxLevel &v) {
        const int n_bx = v.w - ORBX_EDGE - 1 >= ORBX_EDGE ? bd_block_of(v.w - ORBX_EDGE - 1) + 1 : 0;
        return (size_t)2 * n_bx * ((v.h + 31) / 32) * 2 * sizeof(int) + sizeof(int) <= (size_t)60 * 1024;
    };
    int n_lv = 0;
    if (levels.n_levels > 0 && buckets_fit(levels.lv[0])) // (level 0 is the largest: if it fits, all do)
        while (n_lv < orbx_blur_mfma_levels(levels) && levels.lv[n_lv].w >= BD_SRC_W) ++n_lv;
    int buckets = 0;
    for (int l = 0; l < n_lv; ++l) {
        const OrbxLevel &v = levels.lv[l];
        const int w = v.w, h = v.h, limit = l == 0 ? w : v.pitch;
        out.w[l] = w; out.h[l] = h; out.n_ty[l] = (h + 31) / 32; out.scale[l] = v.scale;
        out.bv_off[l] = mf.bv_off[l];
        out.bh_off[l] = (int)(band_h.size() / 16);
        // key-point columns are [19, w - 19): the last one decides how many blocks there are
        const int n_bx = w - ORBX_EDGE - 1 >= ORBX_EDGE ? bd_block_of(w - ORBX_EDGE - 1) + 1 : 0;
        out.n_bx[l] = n_bx;
        out.bucket_base[l] = buckets;
        buckets += 2 * n_bx * out.n_ty[l] + 1; // an early and a late part per (block, trip), and the end
        for (int bi = 0; bi < n_bx; ++bi) {
            blocks.push_back((uint16_t)l); blocks.push_back((uint16_t)bi);
            const int B0 = bd_first_col(bi), xo = bd_origin(B0, limit);
            for (int k = 0; k < BD_WAVES; ++k) {
                int B[64][32] = {};
                const int x0 = B0 + 32 * k, cwin = std::min(std::max(x0 - 16, xo), xo + BD_SRC_W - 64);
                for (int n = 0; n < 32 && x0 + n < w; ++n)
                    for (int i = 0; i < 7; ++i) B[reflect101_host(x0 + n + i - 3, w) - cwin][n] += taps[i];
                for (int s = 0; s < 2; ++s)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 16; ++j) band_h.push_back((uint8_t)(int8_t)B[32 * s + 16 * (lane >> 5) + j][lane & 31]);
            }
        }
    }
    *n_fused_levels = n_lv;
    *bk_stride = std::max(buckets, 1);
}

void orbx_launch_desc_bins(hipStream_t s, const OrbxLevels *d_levels, const OrbxBuffers &b, const BdLevels &tab, int n_fused_levels,
                           int *d_bk_start, int bk_stride, void *d_items, int cap, int32_t *out_n, int n_frames)
{
    if (n_fused_levels <= 0) return;
    int max_nb = 1;
    for (int l = 0; l < n_fused_levels; ++l) max_nb = std::max(max_nb, 2 * tab.n_bx[l] * tab.n_ty[l]);
    hipLaunchKernelGGL(k_desc_bins, dim3(n_fused_levels, n_frames), dim3(256), sizeof(int) * (2 * (size_t)max_nb + 1), s, d_levels, tab, b,
                       d_bk_start, bk_stride, reinterpret_cast<BdItem *>(d_items), cap, out_n);
}

void orbx_launch_desc_fused(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels &levels, const OrbxBuffers &b,
                            const BdLevels &tab, int n_fused_levels, const void *d_blocks, int n_blocks, const void *d_band_h,
                            const void *d_band_v, const int *d_bk_start, int bk_stride, const void *d_items, const int taps[7],
                            orbx_kp *out_kp, uint8_t *out_desc, int cap, int n_frames)
{
    if (n_fused_levels <= 0 || n_blocks <= 0) return;
    FastSrc src;
    for (int l = 0; l < levels.n_levels; ++l) {
        src.base[l] = l == 0 ? l0 : b.img_arena + levels.lv[l].raw_off;
        src.frame_stride[l] = l == 0 ? l0_fs : b.img_frame_stride;
        src.pitch[l] = l == 0 ? l0_pitch : levels.lv[l].pitch;
    }
    int K = 0;
    for (int i = 0; i < 7; ++i) K += taps[i];
    if (K == 256)
        hipLaunchKernelGGL(k_blur_desc<256>, dim3(orbx_xcd_grid(n_blocks, n_frames)), dim3(BD_T), dev_pad(k_blur_desc<256>, 4, 0), s, src, tab,
                           reinterpret_cast<const BdBlock *>(d_blocks), reinterpret_cast<const uint4 *>(d_band_h),
                           reinterpret_cast<const uint4 *>(d_band_v), d_bk_start, bk_stride, reinterpret_cast<const BdItem *>(d_items),
                           levels.kcap_total, out_kp, out_desc, cap, n_blocks, n_frames);
    else
        hipLaunchKernelGGL(k_blur_desc<257>, dim3(orbx_xcd_grid(n_blocks, n_frames)), dim3(BD_T), 0, s, src, tab,
                           reinterpret_cast<const BdBlock *>(d_blocks), reinterpret_cast<const uint4 *>(d_band_h),
                           reinterpret_cast<const uint4 *>(d_band_v), d_bk_start, bk_stride, reinterpret_cast<const BdItem *>(d_items),
                           levels.kcap_total, out_kp, out_desc, cap, n_blocks, n_frames);
}
