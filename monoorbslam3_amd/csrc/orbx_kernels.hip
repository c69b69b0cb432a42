// HIP kernels of the ORB extractor for gfx950 (wave64).
//
// Stage -> reference code it replaces (paths relative to the reference root):
//   k_resize       cv::resize chain            modules/ORB/ORBExtractor.cpp:559-570
//   k_fast_cells   per-cell cv::FAST + retry   modules/ORB/ORBExtractor.cpp:592-617
//   k_blur7        cv::GaussianBlur 7x7 s=2    modules/ORB/ORBExtractor.cpp:527-528
//   k_octree       DistributeOctree            modules/ORB/ORBExtractor.cpp:640-830
//   k_orient_desc  IC_Angle + rBRIEF + output  modules/ORB/ORBExtractor.cpp:18-97, :507-546, :626-637
#include "orbx_internal.h"
#include "orb_math.h"
#include "orb_pattern.h"

typedef unsigned long long u64;

__constant__ int8_t c_pattern[ORB_PATTERN_POINTS * 2] = {ORB_PATTERN_INT8_LIST};

// ---------------------------------------------------------------------------------------------
// Pyramid: fixed-point bilinear (11-bit taps), 4 output pixels per thread
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src, size_t src_fs, int src_pitch, int sw,
                                                int sh, uint8_t *__restrict__ dst, size_t dst_fs, int dst_pitch,
                                                int dw, int dh, const OrbxTap *__restrict__ xtap,
                                                const OrbxTap *__restrict__ ytap)
{
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if (dx0 >= dw || dy >= dh) return;
    const uint8_t *S = src + (size_t)blockIdx.z * src_fs;
    uint8_t *D = dst + (size_t)blockIdx.z * dst_fs + (size_t)dy * dst_pitch;
    const OrbxTap ty = ytap[dy];
    int sy0 = min(max(ty.ofs, 0), sh - 1), sy1 = min(max(ty.ofs + 1, 0), sh - 1);
    const uint8_t *S0 = S + (size_t)sy0 * src_pitch, *S1 = S + (size_t)sy1 * src_pitch;
    const int b0 = ty.c0, b1 = ty.c1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = dx0 + i;
        if (dx < dw) {
            const OrbxTap tx = xtap[dx];
            const int sx = tx.ofs, sx1 = min(sx + 1, sw - 1);
            const int r0 = S0[sx] * tx.c0 + S0[sx1] * tx.c1;
            const int r1 = S1[sx] * tx.c0 + S1[sx1] * tx.c1;
            const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 255) << (8 * i);
        }
    }
    // rows of the arena are 64-byte aligned and padded, so the dword store is always in bounds
    *reinterpret_cast<uint32_t *>(D + dx0) = packed;
}

void orbx_launch_resize(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh,
                        uint8_t *dst, size_t dst_fs, int dst_pitch, int dw, int dh, const OrbxTap *xtap,
                        const OrbxTap *ytap, int n_frames)
{
    dim3 block(64, 4), grid((dw + 255) / 256, (dh + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize, grid, block, 0, s, src, src_fs, src_pitch, sw, sh, dst, dst_fs, dst_pitch, dw, dh,
                       xtap, ytap);
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16: threshold-free corner strength S = max(max_arc min(v-p), max_arc min(p-v)) - 1.
// "corner at t" <=> S >= t, and OpenCV's stored score of a corner is S for every t,
// so one strength tile answers both the ini and the min threshold of a cell.
// ---------------------------------------------------------------------------------------------
#define FAST_TP 40 // LDS tile pitch (36 used)

__device__ __forceinline__ int fast_strength(const uint8_t *t /* centre pixel in the LDS tile */, int tlow)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * FAST_TP];      d[1] = v - t[3 * FAST_TP + 1];  d[2] = v - t[2 * FAST_TP + 2];
    d[3] = v - t[FAST_TP + 3];      d[4] = v - t[3];                d[5] = v - t[-FAST_TP + 3];
    d[6] = v - t[-2 * FAST_TP + 2]; d[7] = v - t[-3 * FAST_TP + 1]; d[8] = v - t[-3 * FAST_TP];
    d[9] = v - t[-3 * FAST_TP - 1]; d[10] = v - t[-2 * FAST_TP - 2]; d[11] = v - t[-FAST_TP - 3];
    d[12] = v - t[-3];              d[13] = v - t[FAST_TP - 3];     d[14] = v - t[2 * FAST_TP - 2];
    d[15] = v - t[3 * FAST_TP - 1];
    // an arc of 9 out of 16 holds one pixel of every opposite pair: cheap exact rejection
    int minhi = 1 << 20, maxlo = -(1 << 20);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        minhi = min(minhi, max(d[k], d[k + 8]));
        maxlo = max(maxlo, min(d[k], d[k + 8]));
    }
    if (minhi <= tlow && maxlo >= -tlow) return 0; // S < tlow: irrelevant for both thresholds
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
    const int s = max(best_min, -best_max) - 1;
    return s >= tlow ? s : 0;
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ src, size_t src_fs, int src_pitch,
                                                    OrbxLevel lv, int level, int n_levels, u64 *__restrict__ cand,
                                                    size_t cand_fs, int *__restrict__ cand_count, int ini_th,
                                                    int min_th)
{
    __shared__ uint8_t tile[36 * FAST_TP];
    __shared__ uint8_t score[32 * 32]; // cell + 1-px ring of zeros
    __shared__ int s_n_ini, s_n_emit, s_base;

    const int frame = blockIdx.z;
    const int cx = blockIdx.x, cy = blockIdx.y;
    const int x0 = ORBX_EDGE + cx * ORBX_CELL, y0 = ORBX_EDGE + cy * ORBX_CELL;
    const int cw = min(ORBX_CELL, lv.w - ORBX_EDGE - x0), ch = min(ORBX_CELL, lv.h - ORBX_EDGE - y0);
    const uint8_t *S = src + (size_t)frame * src_fs;
    const int tid = threadIdx.x;
    const int tlow = min(ini_th, min_th);

    for (int i = tid; i < 32 * 32 / 4; i += 256) reinterpret_cast<uint32_t *>(score)[i] = 0;
    if (tid == 0) { s_n_ini = 0; s_n_emit = 0; }
    const int tw = cw + 6, th = ch + 6;
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[ty * FAST_TP + tx] = S[(size_t)(y0 - 3 + ty) * src_pitch + (x0 - 3 + tx)];
    }
    __syncthreads();
    for (int i = tid; i < cw * ch; i += 256) {
        const int py = i / cw, px = i - py * cw;
        score[(py + 1) * 32 + px + 1] = (uint8_t)fast_strength(&tile[(py + 3) * FAST_TP + px + 3], tlow);
    }
    __syncthreads();
    // 3x3 strict NMS inside the cell; remember each thread's (up to 4) survivors
    uint32_t mine[4];
    int n_mine = 0, n_ini = 0;
    for (int i = tid; i < cw * ch; i += 256) {
        const int py = i / cw, px = i - py * cw;
        const uint8_t *s = &score[(py + 1) * 32 + px + 1];
        const int sc = s[0];
        if (sc > 0 && sc > s[-1] && sc > s[1] && sc > s[-33] && sc > s[-32] && sc > s[-31] && sc > s[31] &&
            sc > s[32] && sc > s[33]) {
            mine[n_mine++] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)sc << 16);
            n_ini += sc >= ini_th;
        }
    }
    if (n_ini) atomicAdd(&s_n_ini, n_ini);
    __syncthreads();
    const int thr = s_n_ini > 0 ? ini_th : min_th; // reference :604-607 -- retry the cell at the min threshold
    int keep = 0;
    for (int k = 0; k < n_mine; ++k) keep += (int)(mine[k] >> 16) >= thr;
    int slot = 0;
    if (keep) slot = atomicAdd(&s_n_emit, keep);
    __syncthreads();
    if (tid == 0 && s_n_emit > 0) s_base = atomicAdd(&cand_count[frame * n_levels + level], s_n_emit);
    __syncthreads();
    if (keep) {
        u64 *out = cand + (size_t)frame * cand_fs + lv.cand_off + s_base + slot;
        for (int k = 0; k < n_mine; ++k) {
            const int sc = mine[k] >> 16;
            if (sc >= thr) {
                const uint32_t x = cx * ORBX_CELL + (mine[k] & 255), y = cy * ORBX_CELL + ((mine[k] >> 8) & 255);
                *out++ = (u64)(x | (y << 16)) | ((u64)sc << 32);
            }
        }
    }
}

void orbx_launch_fast(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, const OrbxLevel &lv, int level,
                      int n_levels, const OrbxBuffers &b, int ini_th, int min_th, int n_frames)
{
    if (lv.n_cols <= 0 || lv.n_rows <= 0) return;
    dim3 grid(lv.n_cols, lv.n_rows, n_frames);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, src, src_fs, src_pitch, lv, level, n_levels, b.cand,
                       b.cand_frame_stride, b.cand_count, ini_th, min_th);
}

// ---------------------------------------------------------------------------------------------
// 7x7 Gaussian, 8.8 fixed-point taps, BORDER_REFLECT_101: H pass exact in u16, V pass rounded >>16
// ---------------------------------------------------------------------------------------------
#define BL_W 64
#define BL_H 16
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur7(const uint8_t *__restrict__ src, size_t src_fs, int src_pitch,
                                               uint8_t *__restrict__ dst, size_t dst_fs, int dst_pitch, int w,
                                               int h, const int *__restrict__ taps)
{
    __shared__ uint8_t raw[(BL_H + 6) * (BL_W + 8)];
    __shared__ uint16_t hor[(BL_H + 6) * BL_W];
    const uint8_t *S = src + (size_t)blockIdx.z * src_fs;
    uint8_t *D = dst + (size_t)blockIdx.z * dst_fs;
    const int x0 = blockIdx.x * BL_W, y0 = blockIdx.y * BL_H;
    const int tid = threadIdx.x;
    int k[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) k[i] = taps[i];
    for (int i = tid; i < (BL_H + 6) * (BL_W + 6); i += 256) {
        const int ty = i / (BL_W + 6), tx = i - ty * (BL_W + 6);
        const int sx = reflect101(x0 - 3 + tx, w), sy = reflect101(y0 - 3 + ty, h);
        raw[ty * (BL_W + 8) + tx] = S[(size_t)sy * src_pitch + sx];
    }
    __syncthreads();
    for (int i = tid; i < (BL_H + 6) * BL_W; i += 256) {
        const int ty = i / BL_W, tx = i - ty * BL_W;
        const uint8_t *r = &raw[ty * (BL_W + 8) + tx];
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += (uint32_t)k[j] * r[j];
        hor[i] = (uint16_t)min(acc, 65535u);
    }
    __syncthreads();
    for (int i = tid; i < BL_H * BL_W; i += 256) {
        const int ty = i / BL_W, tx = i - ty * BL_W;
        const int x = x0 + tx, y = y0 + ty;
        if (x < w && y < h) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += (uint32_t)k[j] * hor[(ty + j) * BL_W + tx];
            D[(size_t)y * dst_pitch + x] = (uint8_t)min((acc + (1u << 15)) >> 16, 255u);
        }
    }
}

void orbx_launch_blur(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, uint8_t *dst, size_t dst_fs,
                      int dst_pitch, int w, int h, const int *taps7, int n_frames)
{
    dim3 grid((w + BL_W - 1) / BL_W, (h + BL_H - 1) / BL_H, n_frames);
    hipLaunchKernelGGL(k_blur7, grid, dim3(256), 0, s, src, src_fs, src_pitch, dst, dst_fs, dst_pitch, w, h, taps7);
}

// ---------------------------------------------------------------------------------------------
// DistributeOctree, one workgroup per (frame, level).
//
// The reference's std::list is kept as an array in list order.  Every pass (a main round, or one
// sweep of the final phase) splits a set of nodes in a processing order; children are pushed to
// the list front one by one, so afterwards
//     list' = reverse(children in creation order) ++ (old list minus the split nodes).
// Candidates only carry the position of their node; child occupancy is counted with atomics and
// positions come from block-wide prefix sums.  Nodes never need their points in order: the
// survivor of a node is its strongest point, ties to the earliest candidate in the reference's
// cell-major emission order, which is recomputed from (x, y).
// ---------------------------------------------------------------------------------------------
struct OctCtx {
    const u64 *cand;
    uint32_t *pnode;
    short4 *bnd[2];
    int *cnt[2];
    int *rank, *node_of_rank, *newpos, *childcnt, *childpos;
    u64 *best;
    int n;
};

__device__ __forceinline__ int block_scan_excl(int v, int *total, int *lds)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wid] = x;
    __syncthreads();
    int wprefix = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < ORBX_OCT_THREADS / 64; ++i) {
        const int t = lds[i];
        if (i < wid) wprefix += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return wprefix + x - v;
}

__device__ __forceinline__ int oct_quadrant(u64 c, short4 b)
{
    const int x = (int)(c & 0xFFFF), y = (int)((c >> 16) & 0xFFFF);
    const int midx = b.x + (b.z - b.x) / 2, midy = b.y + (b.w - b.y) / 2; // DivideNode :368-369
    return (x >= midx) + 2 * (y >= midy);                                 // n1,n2,n3,n4 = 0,1,2,3 (:397-407)
}

__device__ __forceinline__ short4 oct_child_bounds(short4 b, int q)
{
    const short midx = (short)(b.x + (b.z - b.x) / 2), midy = (short)(b.y + (b.w - b.y) / 2);
    short4 r;
    r.x = (q & 1) ? midx : b.x;
    r.z = (q & 1) ? b.z : midx;
    r.y = (q & 2) ? midy : b.y;
    r.w = (q & 2) ? b.w : midy;
    return r;
}

// child occupancy of the nodes ranked [0, nrank) in processing order
__device__ void oct_child_counts(const OctCtx &c, int cur, int nrank)
{
    for (int i = threadIdx.x; i < 4 * nrank; i += ORBX_OCT_THREADS) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += ORBX_OCT_THREADS) {
        const int old = c.pnode[p];
        const int r = c.rank[old];
        if (r >= 0) atomicAdd(&c.childcnt[4 * r + oct_quadrant(c.cand[p], c.bnd[cur][old])], 1);
    }
    __syncthreads();
}

// split the nodes ranked [0, nsplit); everything else is carried over behind the new children
__device__ void oct_apply(const OctCtx &c, int cur, int size, int nsplit, int *new_size, int *n_expand, int *lds)
{
    const int nxt = cur ^ 1;
    const int len = 4 * nsplit;
    const int chunk = (len + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
    const int i0 = min(threadIdx.x * chunk, len), i1 = min(i0 + chunk, len);
    int s = 0, e = 0;
    for (int i = i0; i < i1; ++i) {
        s += c.childcnt[i] > 0;
        e += c.childcnt[i] > 1;
    }
    int T, E;
    int ci = block_scan_excl(s, &T, lds);
    block_scan_excl(e, &E, lds);
    const int chunk2 = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
    const int j0 = min(threadIdx.x * chunk2, size), j1 = min(j0 + chunk2, size);
    int u = 0;
    for (int j = j0; j < j1; ++j) {
        const int r = c.rank[j];
        u += !(r >= 0 && r < nsplit);
    }
    int U;
    int ui = block_scan_excl(u, &U, lds);
    for (int i = i0; i < i1; ++i) {
        const int n = c.childcnt[i];
        if (n > 0) {
            const int pos = T - 1 - ci;
            c.bnd[nxt][pos] = oct_child_bounds(c.bnd[cur][c.node_of_rank[i >> 2]], i & 3);
            c.cnt[nxt][pos] = n;
            c.childpos[i] = pos;
            ++ci;
        }
    }
    for (int j = j0; j < j1; ++j) {
        const int r = c.rank[j];
        if (!(r >= 0 && r < nsplit)) {
            const int pos = T + ui;
            c.bnd[nxt][pos] = c.bnd[cur][j];
            c.cnt[nxt][pos] = c.cnt[cur][j];
            c.newpos[j] = pos;
            ++ui;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += ORBX_OCT_THREADS) {
        const int old = c.pnode[p];
        const int r = c.rank[old];
        c.pnode[p] = (r >= 0 && r < nsplit) ? c.childpos[4 * r + oct_quadrant(c.cand[p], c.bnd[cur][old])]
                                            : c.newpos[old];
    }
    __syncthreads();
    *new_size = T + U;
    *n_expand = E;
}

__global__ __launch_bounds__(ORBX_OCT_THREADS) void k_octree(const OrbxLevels *__restrict__ levels, OrbxBuffers b)
{
    extern __shared__ u64 sort_keys[];
    __shared__ int lds[16];
    __shared__ int s_first;

    const int level = blockIdx.x, frame = blockIdx.y;
    const OrbxLevel lv = levels->lv[level];
    const int tid = threadIdx.x;
    const size_t nb = (size_t)frame * b.node_frame_stride + lv.node_off;
    OctCtx c;
    c.cand = b.cand + (size_t)frame * b.cand_frame_stride + lv.cand_off;
    c.pnode = b.pnode + (size_t)frame * b.cand_frame_stride + lv.cand_off;
    c.bnd[0] = b.bnd0 + nb; c.bnd[1] = b.bnd1 + nb;
    c.cnt[0] = b.cnt0 + nb; c.cnt[1] = b.cnt1 + nb;
    c.rank = b.rank + nb; c.node_of_rank = b.node_of_rank + nb; c.newpos = b.newpos + nb;
    c.childcnt = b.childcnt + 4 * nb; c.childpos = b.childpos + 4 * nb;
    c.best = b.best + nb;
    c.n = min(b.cand_count[frame * ORBX_MAX_LEVELS + level], lv.cand_cap);
    int *out_count = &b.sel_count[frame * ORBX_MAX_LEVELS + level];
    if (c.n <= 0 || lv.region_w <= 0 || lv.region_h <= 0) {
        if (tid == 0) *out_count = 0;
        return;
    }
    const int N = lv.quota;

    // ---- initial nodes (:645-686); empty ones are erased, list order kept
    int cur = 0;
    for (int i = tid; i < lv.n_ini; i += ORBX_OCT_THREADS) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) {
        const int idx = (int)(c.cand[p] & 0xFFFF) / lv.h_x;
        c.pnode[p] = idx;
        atomicAdd(&c.childcnt[idx], 1);
    }
    __syncthreads();
    int size;
    {
        const int chunk = (lv.n_ini + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
        const int i0 = min(tid * chunk, lv.n_ini), i1 = min(i0 + chunk, lv.n_ini);
        int s = 0;
        for (int i = i0; i < i1; ++i) s += c.childcnt[i] > 0;
        int pos = block_scan_excl(s, &size, lds);
        for (int i = i0; i < i1; ++i) {
            if (c.childcnt[i] > 0) {
                short4 bb;
                bb.x = (short)(lv.h_x * i); bb.y = 0;
                bb.z = (short)((i == lv.n_ini - 1) ? (lv.w - ORBX_EDGE) : lv.h_x * (i + 1)); // :665 absolute maxX
                bb.w = (short)lv.region_h;
                c.bnd[cur][pos] = bb;
                c.cnt[cur][pos] = c.childcnt[i];
                c.newpos[i] = pos++;
            }
        }
        __syncthreads();
        for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) c.pnode[p] = c.newpos[c.pnode[p]];
        __syncthreads();
    }

    // ---- main rounds (:692-751)
    bool finish = false;
    while (!finish) {
        const int pre = size;
        const int chunk = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
        const int j0 = min(tid * chunk, size), j1 = min(j0 + chunk, size);
        int s = 0;
        for (int j = j0; j < j1; ++j) s += c.cnt[cur][j] > 1;
        int nsplit;
        int r = block_scan_excl(s, &nsplit, lds);
        for (int j = j0; j < j1; ++j) {
            if (c.cnt[cur][j] > 1) { c.rank[j] = r; c.node_of_rank[r] = j; ++r; }
            else c.rank[j] = -1;
        }
        __syncthreads();
        oct_child_counts(c, cur, nsplit);
        int n_expand;
        oct_apply(c, cur, size, nsplit, &size, &n_expand, lds);
        cur ^= 1;
        if (size > N || size == pre) {
            finish = true;
        } else if (size + 3 * n_expand > N) {
            // ---- final phase (:752-809): split in ascending (point count, creation order) until >= N nodes
            while (!finish) {
                const int pre2 = size;
                const int ch2 = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
                const int a0 = min(tid * ch2, size), a1 = min(a0 + ch2, size);
                int k = 0;
                for (int j = a0; j < a1; ++j) k += c.cnt[cur][j] > 1;
                int K;
                int ko = block_scan_excl(k, &K, lds);
                int P = 1;
                while (P < K) P <<= 1;
                for (int j = a0; j < a1; ++j) {
                    c.rank[j] = -1;
                    // created later <=> closer to the list head, so creation order = descending position
                    if (c.cnt[cur][j] > 1) sort_keys[ko++] = ((u64)c.cnt[cur][j] << 32) | (u64)(0xFFFFFFFFu - (uint32_t)j);
                }
                for (int i = K + tid; i < P; i += ORBX_OCT_THREADS) sort_keys[i] = ~0ull;
                __syncthreads();
                for (int kk = 2; kk <= P; kk <<= 1)
                    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                        for (int i = tid; i < P; i += ORBX_OCT_THREADS) {
                            const int ixj = i ^ jj;
                            if (ixj > i) {
                                const u64 x = sort_keys[i], y = sort_keys[ixj];
                                if ((x > y) == ((i & kk) == 0)) { sort_keys[i] = y; sort_keys[ixj] = x; }
                            }
                        }
                        __syncthreads();
                    }
                for (int sidx = tid; sidx < K; sidx += ORBX_OCT_THREADS) {
                    const int pos = (int)(0xFFFFFFFFu - (uint32_t)(sort_keys[sidx] & 0xFFFFFFFFu));
                    c.rank[pos] = sidx;
                    c.node_of_rank[sidx] = pos;
                }
                if (tid == 0) s_first = K;
                __syncthreads();
                oct_child_counts(c, cur, K);
                // first sorted index at which the list reaches N nodes (:802-803)
                const int ch3 = (K + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
                const int b0 = min(tid * ch3, K), b1 = min(b0 + ch3, K);
                int g = 0;
                for (int i = b0; i < b1; ++i) {
                    int ne = 0;
                    for (int q = 0; q < 4; ++q) ne += c.childcnt[4 * i + q] > 0;
                    g += ne - 1;
                }
                int G;
                int acc = size + block_scan_excl(g, &G, lds);
                for (int i = b0; i < b1; ++i) {
                    int ne = 0;
                    for (int q = 0; q < 4; ++q) ne += c.childcnt[4 * i + q] > 0;
                    acc += ne - 1;
                    if (acc >= N) { atomicMin(&s_first, i); break; }
                }
                __syncthreads();
                const int nsplit2 = min(s_first + 1, K);
                __syncthreads();
                int ne2;
                oct_apply(c, cur, size, nsplit2, &size, &ne2, lds);
                cur ^= 1;
                if (size >= N || size == pre2) finish = true;
            }
        }
    }

    // ---- strongest point per node, first in reference emission order on ties (:812-827)
    for (int j = tid; j < size; j += ORBX_OCT_THREADS) c.best[j] = 0;
    __syncthreads();
    const uint32_t ncols = (uint32_t)lv.n_cols;
    for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) {
        const u64 cd = c.cand[p];
        const uint32_t x = (uint32_t)(cd & 0xFFFF), y = (uint32_t)((cd >> 16) & 0xFFFF), resp = (uint32_t)(cd >> 32);
        const uint32_t order = ((y / ORBX_CELL) * ncols + x / ORBX_CELL) * (ORBX_CELL * ORBX_CELL) +
                               (y % ORBX_CELL) * ORBX_CELL + x % ORBX_CELL;
        atomicMax(&c.best[c.pnode[p]], ((u64)resp << 32) | (u64)(0xFFFFFFFFu - order));
    }
    __syncthreads();
    uint2 *sel = b.sel + (size_t)frame * levels->kcap_total + lv.kp_off;
    const int n_out = min(size, lv.kcap);
    for (int j = tid; j < n_out; j += ORBX_OCT_THREADS) {
        const u64 k = c.best[j];
        const uint32_t order = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu);
        const uint32_t cell = order / (ORBX_CELL * ORBX_CELL), in = order % (ORBX_CELL * ORBX_CELL);
        const uint32_t x = (cell % ncols) * ORBX_CELL + in % ORBX_CELL + ORBX_EDGE;
        const uint32_t y = (cell / ncols) * ORBX_CELL + in / ORBX_CELL + ORBX_EDGE;
        sel[j] = make_uint2(x | (y << 16), (uint32_t)(k >> 32));
    }
    if (tid == 0) *out_count = n_out;
}

// ---------------------------------------------------------------------------------------------
// Same algorithm with the node list resident in LDS (the default path).
//
// Node rectangles of the reference form, per initial node, a product grid: the x-interval of a node
// depends only on the left/right choices along its path and the y-interval only on the up/down
// choices (DivideNode halves each axis independently).  So every candidate's whole descent --
// two bits per depth -- is computed ONCE from its (x, y) ("pcode"), and a pass needs no node
// rectangles at all: the quadrant of point p in a node of depth d is bits [31-2d, 30-2d] of
// pcode[p].  Per pass the candidates are streamed coalesced from HBM/L2 (position + code), all
// list state (count|depth per node, ranks, child slots) lives in LDS, and child occupancy uses
// LDS atomics.
// ---------------------------------------------------------------------------------------------
struct OctL {
    const u64 *cand;
    uint32_t *pnode, *pcode;
    uint32_t *node[2]; // depth << 24 | count
    uint32_t *childcnt;
    uint16_t *rank, *newpos, *node_of_rank, *childpos;
    int n;
};
#define OCT_NORANK 0xFFFFu

__device__ __forceinline__ void octl_child_counts(const OctL &c, int cur, int nrank)
{
    for (int i = threadIdx.x; i < 4 * nrank; i += ORBX_OCT_THREADS) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += ORBX_OCT_THREADS) {
        const uint32_t old = c.pnode[p], code = c.pcode[p];
        const uint32_t r = c.rank[old];
        if (r != OCT_NORANK) {
            const uint32_t d = c.node[cur][old] >> 24;
            atomicAdd(&c.childcnt[4 * r + ((code >> (30 - 2 * d)) & 3)], 1u);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void octl_apply(const OctL &c, int cur, int size, int nsplit, int *new_size, int *n_expand,
                                           int *lds)
{
    const int nxt = cur ^ 1;
    const int len = 4 * nsplit;
    const int chunk = (len + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
    const int i0 = min((int)threadIdx.x * chunk, len), i1 = min(i0 + chunk, len);
    int s = 0, e = 0;
    for (int i = i0; i < i1; ++i) {
        s += c.childcnt[i] > 0;
        e += c.childcnt[i] > 1;
    }
    int T, E;
    int ci = block_scan_excl(s, &T, lds);
    block_scan_excl(e, &E, lds);
    const int chunk2 = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
    const int j0 = min((int)threadIdx.x * chunk2, size), j1 = min(j0 + chunk2, size);
    int u = 0;
    for (int j = j0; j < j1; ++j) u += !(c.rank[j] < (uint32_t)nsplit);
    int U;
    int ui = block_scan_excl(u, &U, lds);
    for (int i = i0; i < i1; ++i) {
        const uint32_t n = c.childcnt[i];
        if (n > 0) {
            const int pos = T - 1 - ci;
            const uint32_t pd = c.node[cur][c.node_of_rank[i >> 2]] >> 24;
            c.node[nxt][pos] = ((pd + 1) << 24) | n;
            c.childpos[i] = (uint16_t)pos;
            ++ci;
        }
    }
    for (int j = j0; j < j1; ++j) {
        if (!(c.rank[j] < (uint32_t)nsplit)) {
            const int pos = T + ui;
            c.node[nxt][pos] = c.node[cur][j];
            c.newpos[j] = (uint16_t)pos;
            ++ui;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += ORBX_OCT_THREADS) {
        const uint32_t old = c.pnode[p];
        const uint32_t r = c.rank[old];
        uint32_t np;
        if (r < (uint32_t)nsplit) {
            const uint32_t d = c.node[cur][old] >> 24;
            np = c.childpos[4 * r + ((c.pcode[p] >> (30 - 2 * d)) & 3)];
        } else {
            np = c.newpos[old];
        }
        c.pnode[p] = np;
    }
    __syncthreads();
    *new_size = T + U;
    *n_expand = E;
}

__global__ __launch_bounds__(ORBX_OCT_THREADS) void k_octree_lds(const OrbxLevels *__restrict__ levels, OrbxBuffers b)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int lds[16];
    __shared__ int s_first;

    const int level = blockIdx.x, frame = blockIdx.y;
    const OrbxLevel lv = levels->lv[level];
    const int tid = threadIdx.x;
    OctL c;
    c.cand = b.cand + (size_t)frame * b.cand_frame_stride + lv.cand_off;
    c.pnode = b.pnode + (size_t)frame * b.cand_frame_stride + lv.cand_off;
    c.pcode = b.pcode + (size_t)frame * b.cand_frame_stride + lv.cand_off;
    c.n = min(b.cand_count[frame * ORBX_MAX_LEVELS + level], lv.cand_cap);
    int *out_count = &b.sel_count[frame * ORBX_MAX_LEVELS + level];
    if (c.n <= 0 || lv.region_w <= 0 || lv.region_h <= 0) {
        if (tid == 0) *out_count = 0;
        return;
    }
    const int N = lv.quota, M = lv.node_cap, NSC = max(lv.quota, lv.n_ini) + 4;
    int P2 = 1;
    while (P2 < N) P2 <<= 1;
    u64 *sort_keys = reinterpret_cast<u64 *>(smem);
    c.node[0] = reinterpret_cast<uint32_t *>(sort_keys + P2);
    c.node[1] = c.node[0] + M;
    c.childcnt = c.node[1] + M;
    c.rank = reinterpret_cast<uint16_t *>(c.childcnt + 4 * NSC);
    c.newpos = c.rank + M;
    c.node_of_rank = c.newpos + M;
    c.childpos = c.node_of_rank + NSC;

    // ---- descent code of every candidate + initial nodes (:645-686)
    int cur = 0;
    for (int i = tid; i < lv.n_ini; i += ORBX_OCT_THREADS) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) {
        const u64 cd = c.cand[p];
        const int x = (int)(cd & 0xFFFF), y = (int)((cd >> 16) & 0xFFFF);
        const int idx = x / lv.h_x;
        int ulx = lv.h_x * idx, brx = (idx == lv.n_ini - 1) ? (lv.w - ORBX_EDGE) : lv.h_x * (idx + 1); // :665
        int uly = 0, bry = lv.region_h;
        uint32_t code = 0;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const int midx = ulx + (brx - ulx) / 2, midy = uly + (bry - uly) / 2; // DivideNode :368-369
            const int qx = x >= midx, qy = y >= midy;                             // :397-407
            ulx = qx ? midx : ulx; brx = qx ? brx : midx;
            uly = qy ? midy : uly; bry = qy ? bry : midy;
            code |= (uint32_t)(qx | (qy << 1)) << (30 - 2 * d);
        }
        c.pcode[p] = code;
        c.pnode[p] = idx;
        atomicAdd(&c.childcnt[idx], 1u);
    }
    __syncthreads();
    int size;
    {
        const int chunk = (lv.n_ini + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
        const int i0 = min(tid * chunk, lv.n_ini), i1 = min(i0 + chunk, lv.n_ini);
        int s = 0;
        for (int i = i0; i < i1; ++i) s += c.childcnt[i] > 0;
        int pos = block_scan_excl(s, &size, lds);
        for (int i = i0; i < i1; ++i)
            if (c.childcnt[i] > 0) {
                c.node[cur][pos] = c.childcnt[i]; // depth 0
                c.newpos[i] = (uint16_t)pos++;
            }
        __syncthreads();
        for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) c.pnode[p] = c.newpos[c.pnode[p]];
        __syncthreads();
    }

    // ---- main rounds (:692-751)
    bool finish = false;
    while (!finish) {
        const int pre = size;
        const int chunk = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
        const int j0 = min(tid * chunk, size), j1 = min(j0 + chunk, size);
        int s = 0;
        for (int j = j0; j < j1; ++j) s += (c.node[cur][j] & 0xFFFFFF) > 1;
        int nsplit;
        int r = block_scan_excl(s, &nsplit, lds);
        for (int j = j0; j < j1; ++j) {
            if ((c.node[cur][j] & 0xFFFFFF) > 1) { c.rank[j] = (uint16_t)r; c.node_of_rank[r] = (uint16_t)j; ++r; }
            else c.rank[j] = OCT_NORANK;
        }
        __syncthreads();
        octl_child_counts(c, cur, nsplit);
        int n_expand;
        octl_apply(c, cur, size, nsplit, &size, &n_expand, lds);
        cur ^= 1;
        if (size > N || size == pre) {
            finish = true;
        } else if (size + 3 * n_expand > N) {
            // ---- final phase (:752-809)
            while (!finish) {
                const int pre2 = size;
                const int ch2 = (size + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
                const int a0 = min(tid * ch2, size), a1 = min(a0 + ch2, size);
                int k = 0;
                for (int j = a0; j < a1; ++j) k += (c.node[cur][j] & 0xFFFFFF) > 1;
                int K;
                int ko = block_scan_excl(k, &K, lds);
                int P = 1;
                while (P < K) P <<= 1;
                for (int j = a0; j < a1; ++j) {
                    c.rank[j] = OCT_NORANK;
                    const uint32_t cn = c.node[cur][j] & 0xFFFFFF;
                    if (cn > 1) sort_keys[ko++] = ((u64)cn << 32) | (u64)(0xFFFFFFFFu - (uint32_t)j);
                }
                for (int i = K + tid; i < P; i += ORBX_OCT_THREADS) sort_keys[i] = ~0ull;
                __syncthreads();
                for (int kk = 2; kk <= P; kk <<= 1)
                    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                        for (int i = tid; i < P; i += ORBX_OCT_THREADS) {
                            const int ixj = i ^ jj;
                            if (ixj > i) {
                                const u64 x = sort_keys[i], y = sort_keys[ixj];
                                if ((x > y) == ((i & kk) == 0)) { sort_keys[i] = y; sort_keys[ixj] = x; }
                            }
                        }
                        __syncthreads();
                    }
                for (int sidx = tid; sidx < K; sidx += ORBX_OCT_THREADS) {
                    const int pos = (int)(0xFFFFFFFFu - (uint32_t)(sort_keys[sidx] & 0xFFFFFFFFu));
                    c.rank[pos] = (uint16_t)sidx;
                    c.node_of_rank[sidx] = (uint16_t)pos;
                }
                if (tid == 0) s_first = K;
                __syncthreads();
                octl_child_counts(c, cur, K);
                const int ch3 = (K + ORBX_OCT_THREADS - 1) / ORBX_OCT_THREADS;
                const int b0 = min(tid * ch3, K), b1 = min(b0 + ch3, K);
                int g = 0;
                for (int i = b0; i < b1; ++i) {
                    int ne = 0;
                    for (int q = 0; q < 4; ++q) ne += c.childcnt[4 * i + q] > 0;
                    g += ne - 1;
                }
                int G;
                int acc = size + block_scan_excl(g, &G, lds);
                for (int i = b0; i < b1; ++i) {
                    int ne = 0;
                    for (int q = 0; q < 4; ++q) ne += c.childcnt[4 * i + q] > 0;
                    acc += ne - 1;
                    if (acc >= N) { atomicMin(&s_first, i); break; }
                }
                __syncthreads();
                const int nsplit2 = min(s_first + 1, K);
                __syncthreads();
                int ne2;
                octl_apply(c, cur, size, nsplit2, &size, &ne2, lds);
                cur ^= 1;
                if (size >= N || size == pre2) finish = true;
            }
        }
    }

    // ---- strongest point per node (:812-827); the node arrays are dead now and hold the maxima
    u64 *best = reinterpret_cast<u64 *>(c.node[0]);
    __syncthreads();
    for (int j = tid; j < size; j += ORBX_OCT_THREADS) best[j] = 0;
    __syncthreads();
    const uint32_t ncols = (uint32_t)lv.n_cols;
    for (int p = tid; p < c.n; p += ORBX_OCT_THREADS) {
        const u64 cd = c.cand[p];
        const uint32_t x = (uint32_t)(cd & 0xFFFF), y = (uint32_t)((cd >> 16) & 0xFFFF), resp = (uint32_t)(cd >> 32);
        const uint32_t order = ((y / ORBX_CELL) * ncols + x / ORBX_CELL) * (ORBX_CELL * ORBX_CELL) +
                               (y % ORBX_CELL) * ORBX_CELL + x % ORBX_CELL;
        atomicMax(&best[c.pnode[p]], ((u64)resp << 32) | (u64)(0xFFFFFFFFu - order));
    }
    __syncthreads();
    uint2 *sel = b.sel + (size_t)frame * levels->kcap_total + lv.kp_off;
    const int n_out = min(size, lv.kcap);
    for (int j = tid; j < n_out; j += ORBX_OCT_THREADS) {
        const u64 k = best[j];
        const uint32_t order = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu);
        const uint32_t cell = order / (ORBX_CELL * ORBX_CELL), in = order % (ORBX_CELL * ORBX_CELL);
        const uint32_t x = (cell % ncols) * ORBX_CELL + in % ORBX_CELL + ORBX_EDGE;
        const uint32_t y = (cell / ncols) * ORBX_CELL + in / ORBX_CELL + ORBX_EDGE;
        sel[j] = make_uint2(x | (y << 16), (uint32_t)(k >> 32));
    }
    if (tid == 0) *out_count = n_out;
}

size_t orbx_octree_lds_bytes(const OrbxLevels &levels)
{
    size_t need = 0;
    for (int l = 0; l < levels.n_levels; ++l) {
        const OrbxLevel &v = levels.lv[l];
        size_t P2 = 1;
        while (P2 < (size_t)v.quota) P2 <<= 1;
        const size_t M = v.node_cap, NSC = (size_t)(v.quota > v.n_ini ? v.quota : v.n_ini) + 4;
        const size_t bytes = 8 * P2 + 8 * M + 16 * NSC + 4 * M + 2 * NSC + 8 * NSC + 64;
        if (bytes > need) need = bytes;
    }
    return need;
}

void orbx_launch_octree(hipStream_t s, const OrbxLevels *d_levels, const OrbxLevels &levels, const OrbxBuffers &b,
                        int n_frames, size_t sort_lds_bytes)
{
    dim3 grid(levels.n_levels, n_frames);
    const size_t lds_bytes = orbx_octree_lds_bytes(levels);
    bool small_nodes = true; // 16-bit list positions
    for (int l = 0; l < levels.n_levels; ++l) small_nodes = small_nodes && levels.lv[l].node_cap < 65535;
    if (lds_bytes <= 160 * 1024 - 256 && small_nodes) {
        static size_t configured = 0;
        if (lds_bytes > configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree_lds),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            configured = lds_bytes;
        }
        hipLaunchKernelGGL(k_octree_lds, grid, dim3(ORBX_OCT_THREADS), lds_bytes, s, d_levels, b);
    } else {
        // quotas too large for the LDS-resident list: same algorithm with the list in global scratch
        hipLaunchKernelGGL(k_octree, grid, dim3(ORBX_OCT_THREADS), sort_lds_bytes, s, d_levels, b);
    }
}

// ---------------------------------------------------------------------------------------------
// Orientation (intensity centroid on the raw level) + steered BRIEF (on the blurred level) +
// output record, one 256-thread workgroup per keypoint: thread t owns descriptor bit t and the
// four waves write 8 bytes each from their 64-bit ballot.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_orient_desc(const uint8_t *__restrict__ l0, size_t l0_fs, int l0_pitch,
                                                     const OrbxLevels *__restrict__ levels, OrbxBuffers b,
                                                     const int *__restrict__ u_max, orbx_kp *__restrict__ out_kp,
                                                     uint8_t *__restrict__ out_desc, int cap,
                                                     int32_t *__restrict__ out_n)
{
    __shared__ float s_cs[2];
    __shared__ float s_angle;
    const int frame = blockIdx.y, slot = blockIdx.x, tid = threadIdx.x;
    const int L = levels->n_levels;
    int level = 0;
    for (int l = 1; l < L; ++l) level += slot >= levels->lv[l].kp_off;
    const OrbxLevel lv = levels->lv[level];
    const int i = slot - lv.kp_off;
    const int *cnts = b.sel_count + frame * ORBX_MAX_LEVELS;
    if (slot == 0 && tid == 0) {
        int tot = 0;
        for (int l = 0; l < L; ++l) tot += cnts[l];
        out_n[frame] = tot;
    }
    if (i >= cnts[level]) return;
    int out_idx = i;
    for (int l = 0; l < level; ++l) out_idx += cnts[l];
    if (out_idx >= cap) return;

    const uint2 rec = b.sel[(size_t)frame * levels->kcap_total + slot];
    const int x = rec.x & 0xFFFF, y = rec.x >> 16;
    const uint8_t *raw = level == 0 ? l0 + (size_t)frame * l0_fs : b.img_arena + (size_t)frame * b.img_frame_stride + lv.raw_off;
    const int rpitch = level == 0 ? l0_pitch : lv.pitch;
    const uint8_t *blur = b.img_arena + (size_t)frame * b.img_frame_stride + lv.blur_off;

    if (tid < 64) {
        int m10 = 0, m01 = 0;
        if (tid < 2 * ORBX_HALF_PATCH + 1) {
            const int v = tid - ORBX_HALF_PATCH;
            const int d = u_max[v < 0 ? -v : v];
            const uint8_t *row = raw + (size_t)(y + v) * rpitch + x;
            int rs = 0;
            for (int u = -d; u <= d; ++u) {
                const int val = row[u];
                m10 += u * val;
                rs += val;
            }
            m01 = v * rs;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            m10 += __shfl_down(m10, o);
            m01 += __shfl_down(m01, o);
        }
        if (tid == 0) {
            const float ang = orb_fast_atan2((float)m01, (float)m10);
            float cs, sn;
            orb_sincos_deg(ang, &cs, &sn);
            s_angle = ang;
            s_cs[0] = cs;
            s_cs[1] = sn;
        }
    }
    __syncthreads();
    const float a = s_cs[0], bb = s_cs[1];
    const uint8_t *center = blur + (size_t)y * lv.pitch + x;
    const float px0 = (float)c_pattern[4 * tid], py0 = (float)c_pattern[4 * tid + 1];
    const float px1 = (float)c_pattern[4 * tid + 2], py1 = (float)c_pattern[4 * tid + 3];
    const int r0 = orb_round_f(ORB_FADD(ORB_FMUL(px0, bb), ORB_FMUL(py0, a)));
    const int c0 = orb_round_f(ORB_FSUB(ORB_FMUL(px0, a), ORB_FMUL(py0, bb)));
    const int r1 = orb_round_f(ORB_FADD(ORB_FMUL(px1, bb), ORB_FMUL(py1, a)));
    const int c1 = orb_round_f(ORB_FSUB(ORB_FMUL(px1, a), ORB_FMUL(py1, bb)));
    const int t0 = center[r0 * lv.pitch + c0], t1 = center[r1 * lv.pitch + c1];
    const u64 bits = __ballot(t0 < t1);
    uint8_t *desc = out_desc + ((size_t)frame * cap + out_idx) * 32;
    if ((tid & 63) == 0) *reinterpret_cast<u64 *>(desc + 8 * (tid >> 6)) = bits;
    if (tid == 0) {
        orbx_kp kp;
        float fx = (float)x, fy = (float)y;
        if (level != 0) { fx = ORB_FMUL(fx, lv.scale); fy = ORB_FMUL(fy, lv.scale); }
        kp.x = fx; kp.y = fy; kp.size = lv.scale; kp.angle = s_angle; kp.response = (float)rec.y;
        kp.octave = level; kp.class_id = -1;
        out_kp[(size_t)frame * cap + out_idx] = kp;
    }
}

void orbx_launch_orient_desc(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                             const OrbxLevels &levels, const OrbxBuffers &b, const int *u_max, orbx_kp *out_kp,
                             uint8_t *out_desc, int cap, int32_t *out_n, int n_frames)
{
    dim3 grid(levels.kcap_total, n_frames);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, s, l0, l0_fs, l0_pitch, d_levels, b, u_max, out_kp,
                       out_desc, cap, out_n);
}
