// DistributeOctree kernels.  Included by orbx_kernels.hip once per workgroup size, inside a namespace, with OCT_NT defined:
// 512 threads for calls with a few frames (one workgroup per (frame, level) is all the parallelism there is, so a wide
// workgroup shortens the data-parallel phases) and 256 threads for resident batches (the kernel holds 127 VGPRs, i.e. 16
// waves per CU: four 256-thread workgroups per CU instead of two 512-thread ones hide each other's barrier / LDS latency --
// quadtree of 512 frames 0.28 -> 0.23 ms).  Not a stand-alone header.
#ifndef OCT_NT
#error "define OCT_NT (threads per workgroup, a multiple of 64, at most 1024: the block scans keep 16 wave totals)"
#endif
// The block scans below hold one running total per wave -- two scans at a time -- in a 32-entry half of their LDS array: at most
// 16 waves.  (Round 3's 1024-thread build kept the second scan's totals at a fixed offset of 8 and faulted; the offset now follows
// the wave count, and anything beyond 16 waves is refused at compile time.)
static_assert(OCT_NT % 64 == 0 && OCT_NT >= 64 && OCT_NT <= 1024, "quadtree workgroups are whole waves, at most 16 of them (the block scans keep 2 x 16 wave totals per half of their array)");

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

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (VALU latency; a __shfl_up chain
// is six dependent trips through the LDS crossbar).  Every lane must be active.
__device__ __forceinline__ int wave_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return x;
}

// Workgroup-wide exclusive scans.  `lds` holds 64 ints: calls alternate between its halves (`par`, uniform, flips per call),
// so ONE barrier per scan is enough -- a half is rewritten two calls later, after a barrier every reader has passed.
__device__ __forceinline__ int block_scan_excl(int v, int *total, int *lds, int &par)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int *buf = lds + 32 * par;
    par ^= 1;
    const int x = wave_scan_incl(v);
    if (lane == 63) buf[wid] = x;
    __syncthreads();
    int wprefix = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < OCT_NT / 64; ++i) {
        const int t = buf[i];
        if (i < wid) wprefix += t;
        tot += t;
    }
    *total = tot;
    return wprefix + x - v;
}

// two independent exclusive scans for the price (one barrier) of one
__device__ __forceinline__ void block_scan_excl2(int va, int vb, int *ea, int *eb, int *ta, int *tb, int *lds, int &par)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int *buf = lds + 32 * par;
    par ^= 1;
    const int x = wave_scan_incl(va), z = wave_scan_incl(vb);
    if (lane == 63) { buf[wid] = x; buf[OCT_NT / 64 + wid] = z; }
    __syncthreads();
    int pa = 0, pb = 0, sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < OCT_NT / 64; ++i) {
        const int t = buf[i], u = buf[OCT_NT / 64 + i];
        if (i < wid) { pa += t; pb += u; }
        sa += t; sb += u;
    }
    *ta = sa; *tb = sb;
    *ea = pa + x - va; *eb = pb + z - vb;
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
    for (int i = threadIdx.x; i < 4 * nrank; i += OCT_NT) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += OCT_NT) {
        const int old = c.pnode[p];
        const int r = c.rank[old];
        if (r >= 0) atomicAdd(&c.childcnt[4 * r + oct_quadrant(c.cand[p], (cur ? c.bnd[1] : c.bnd[0])[old])], 1);
    }
    __syncthreads();
}

// split the nodes ranked [0, nsplit); everything else is carried over behind the new children
__device__ void oct_apply(const OctCtx &c, int cur, int size, int nsplit, int *new_size, int *n_expand, int *lds, int &par)
{
    const int nxt = cur ^ 1;
    const int len = 4 * nsplit;
    const int chunk = (len + OCT_NT - 1) / OCT_NT;
    const int i0 = min(threadIdx.x * chunk, len), i1 = min(i0 + chunk, len);
    int s = 0, e = 0;
    for (int i = i0; i < i1; ++i) {
        s += c.childcnt[i] > 0;
        e += c.childcnt[i] > 1;
    }
    int T, E;
    int ci = block_scan_excl(s, &T, lds, par);
    block_scan_excl(e, &E, lds, par);
    const int chunk2 = (size + OCT_NT - 1) / OCT_NT;
    const int j0 = min(threadIdx.x * chunk2, size), j1 = min(j0 + chunk2, size);
    int u = 0;
    for (int j = j0; j < j1; ++j) {
        const int r = c.rank[j];
        u += !(r >= 0 && r < nsplit);
    }
    int U;
    int ui = block_scan_excl(u, &U, lds, par);
    for (int i = i0; i < i1; ++i) {
        const int n = c.childcnt[i];
        if (n > 0) {
            const int pos = T - 1 - ci;
            (nxt ? c.bnd[1] : c.bnd[0])[pos] = oct_child_bounds((cur ? c.bnd[1] : c.bnd[0])[c.node_of_rank[i >> 2]], i & 3);
            (nxt ? c.cnt[1] : c.cnt[0])[pos] = n;
            c.childpos[i] = pos;
            ++ci;
        }
    }
    for (int j = j0; j < j1; ++j) {
        const int r = c.rank[j];
        if (!(r >= 0 && r < nsplit)) {
            const int pos = T + ui;
            (nxt ? c.bnd[1] : c.bnd[0])[pos] = (cur ? c.bnd[1] : c.bnd[0])[j];
            (nxt ? c.cnt[1] : c.cnt[0])[pos] = (cur ? c.cnt[1] : c.cnt[0])[j];
            c.newpos[j] = pos;
            ++ui;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < c.n; p += OCT_NT) {
        const int old = c.pnode[p];
        const int r = c.rank[old];
        c.pnode[p] = (r >= 0 && r < nsplit) ? c.childpos[4 * r + oct_quadrant(c.cand[p], (cur ? c.bnd[1] : c.bnd[0])[old])]
                                            : c.newpos[old];
    }
    __syncthreads();
    *new_size = T + U;
    *n_expand = E;
}

__global__ __launch_bounds__(OCT_NT) void k_octree(const OrbxLevels *__restrict__ levels, OrbxBuffers b, int level0)
{
    extern __shared__ u64 sort_keys[];
    __shared__ int lds[64];
    int par = 0; // which half of `lds` the next scan uses
    __shared__ int s_first;

    const int level = level0 + blockIdx.x, frame = blockIdx.y;
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
    for (int i = tid; i < lv.n_ini; i += OCT_NT) c.childcnt[i] = 0;
    __syncthreads();
    for (int p = tid; p < c.n; p += OCT_NT) {
        const int idx = (int)(c.cand[p] & 0xFFFF) / lv.h_x;
        c.pnode[p] = idx;
        atomicAdd(&c.childcnt[idx], 1);
    }
    __syncthreads();
    int size;
    {
        const int chunk = (lv.n_ini + OCT_NT - 1) / OCT_NT;
        const int i0 = min(tid * chunk, lv.n_ini), i1 = min(i0 + chunk, lv.n_ini);
        int s = 0;
        for (int i = i0; i < i1; ++i) s += c.childcnt[i] > 0;
        int pos = block_scan_excl(s, &size, lds, par);
        for (int i = i0; i < i1; ++i) {
            if (c.childcnt[i] > 0) {
                short4 bb;
                bb.x = (short)(lv.h_x * i); bb.y = 0;
                bb.z = (short)((i == lv.n_ini - 1) ? (lv.w - ORBX_EDGE) : lv.h_x * (i + 1)); // :665 absolute maxX
                bb.w = (short)lv.region_h;
                (cur ? c.bnd[1] : c.bnd[0])[pos] = bb;
                (cur ? c.cnt[1] : c.cnt[0])[pos] = c.childcnt[i];
                c.newpos[i] = pos++;
            }
        }
        __syncthreads();
        for (int p = tid; p < c.n; p += OCT_NT) c.pnode[p] = c.newpos[c.pnode[p]];
        __syncthreads();
    }

    // ---- main rounds (:692-751)
    bool finish = false;
    while (!finish) {
        const int pre = size;
        const int chunk = (size + OCT_NT - 1) / OCT_NT;
        const int j0 = min(tid * chunk, size), j1 = min(j0 + chunk, size);
        int s = 0;
        for (int j = j0; j < j1; ++j) s += (cur ? c.cnt[1] : c.cnt[0])[j] > 1;
        int nsplit;
        int r = block_scan_excl(s, &nsplit, lds, par);
        for (int j = j0; j < j1; ++j) {
            if ((cur ? c.cnt[1] : c.cnt[0])[j] > 1) { c.rank[j] = r; c.node_of_rank[r] = j; ++r; }
            else c.rank[j] = -1;
        }
        __syncthreads();
        oct_child_counts(c, cur, nsplit);
        int n_expand;
        oct_apply(c, cur, size, nsplit, &size, &n_expand, lds, par);
        cur ^= 1;
        if (size > N || size == pre) {
            finish = true;
        } else if (size + 3 * n_expand > N) {
            // ---- final phase (:752-809): split in ascending (point count, creation order) until >= N nodes
            while (!finish) {
                const int pre2 = size;
                const int ch2 = (size + OCT_NT - 1) / OCT_NT;
                const int a0 = min(tid * ch2, size), a1 = min(a0 + ch2, size);
                int k = 0;
                for (int j = a0; j < a1; ++j) k += (cur ? c.cnt[1] : c.cnt[0])[j] > 1;
                int K;
                int ko = block_scan_excl(k, &K, lds, par);
                int P = 1;
                while (P < K) P <<= 1;
                for (int j = a0; j < a1; ++j) {
                    c.rank[j] = -1;
                    // created later <=> closer to the list head, so creation order = descending position
                    if ((cur ? c.cnt[1] : c.cnt[0])[j] > 1) sort_keys[ko++] = ((u64)(cur ? c.cnt[1] : c.cnt[0])[j] << 32) | (u64)(0xFFFFFFFFu - (uint32_t)j);
                }
                for (int i = K + tid; i < P; i += OCT_NT) sort_keys[i] = ~0ull;
                __syncthreads();
                // Bitonic network.  Element i belongs to thread i mod 512, so a wave owns 64 consecutive elements of every
                // 512-block: exchanges at distance < 64 stay inside the wave (its LDS operations complete in order) and need
                // no workgroup barrier -- 6 barriers instead of 45 for 512 keys.
                for (int kk = 2; kk <= P; kk <<= 1)
                    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                        for (int i = tid; i < P; i += OCT_NT) {
                            const int ixj = i ^ jj;
                            if (ixj > i) {
                                const u64 x = sort_keys[i], y = sort_keys[ixj];
                                if ((x > y) == ((i & kk) == 0)) { sort_keys[i] = y; sort_keys[ixj] = x; }
                            }
                        }
                        const int next_jj = jj > 1 ? (jj >> 1) : kk; // the first distance of the next stage is kk
                        if (jj >= 64 || next_jj >= 64 || (jj == 1 && kk == P)) __syncthreads();
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                for (int sidx = tid; sidx < K; sidx += OCT_NT) {
                    const int pos = (int)(0xFFFFFFFFu - (uint32_t)(sort_keys[sidx] & 0xFFFFFFFFu));
                    c.rank[pos] = sidx;
                    c.node_of_rank[sidx] = pos;
                }
                if (tid == 0) s_first = K;
                __syncthreads();
                oct_child_counts(c, cur, K);
                // first sorted index at which the list reaches N nodes (:802-803)
                const int ch3 = (K + OCT_NT - 1) / OCT_NT;
                const int b0 = min(tid * ch3, K), b1 = min(b0 + ch3, K);
                int g = 0;
                for (int i = b0; i < b1; ++i) {
                    int ne = 0;
                    for (int q = 0; q < 4; ++q) ne += c.childcnt[4 * i + q] > 0;
                    g += ne - 1;
                }
                int G;
                int acc = size + block_scan_excl(g, &G, lds, par);
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
                oct_apply(c, cur, size, nsplit2, &size, &ne2, lds, par);
                cur ^= 1;
                if (size >= N || size == pre2) finish = true;
            }
        }
    }

    // ---- strongest point per node, first in reference emission order on ties (:812-827)
    for (int j = tid; j < size; j += OCT_NT) c.best[j] = 0;
    __syncthreads();
    const uint32_t ncols = (uint32_t)lv.n_cols;
    for (int p = tid; p < c.n; p += OCT_NT) {
        const u64 cd = c.cand[p];
        const uint32_t x = (uint32_t)(cd & 0xFFFF), y = (uint32_t)((cd >> 16) & 0xFFFF), resp = (uint32_t)(cd >> 32);
        const uint32_t order = ((y / ORBX_CELL) * ncols + x / ORBX_CELL) * (ORBX_CELL * ORBX_CELL) +
                               (y % ORBX_CELL) * ORBX_CELL + x % ORBX_CELL;
        atomicMax(&c.best[c.pnode[p]], ((u64)resp << 32) | (u64)(0xFFFFFFFFu - order));
    }
    __syncthreads();
    uint2 *sel = b.sel + (size_t)frame * levels->kcap_total + lv.kp_off;
    const int n_out = min(size, lv.kcap);
    for (int j = tid; j < n_out; j += OCT_NT) {
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
    uint32_t *cc;      // [4 x list position] candidates per child quadrant of every node of the CURRENT list
    uint16_t *rank, *newpos, *node_of_rank, *childpos;
    int n;
};
#define OCT_NORANK 0xFFFFu
#define OCT_RANK_BY_COUNTING (OCT_NT >= 1024 ? 1024 : 320) // final-phase keys up to which ranks are counted instead of sorted (one thread per key)

#ifndef OCT_UNROLL
#define OCT_UNROLL 4
#endif
#define OCT_DIGITS0 6 // descent digits computed up front (see more_digits)
#ifndef OCT_REG
#define OCT_REG 8
#endif
// OCT_REG: candidates per thread whose (node, descent code, record) live in registers: the first 4096 of a level
// One pass = one call: the nodes ranked below `nsplit` are replaced by their non-empty children (pushed to the front in
// creation order, i.e. reversed), the others keep their relative order behind them; every candidate moves to its new node
// AND is counted into that node's child quadrants, so the next pass (or the stop search of the final phase) finds the
// child-count table `cc` of the new list ready -- the candidates are streamed once per pass, not twice.
__device__ __forceinline__ void octl_apply(const OctL &c, int cur, int size, int nsplit, int *new_size, int *n_expand,
                                           int *lds, int &par, uint32_t (&rn)[OCT_REG], const uint32_t (&rc)[OCT_REG])
{
    const int nxt = cur ^ 1;
    const uint32_t *ncur = cur ? c.node[1] : c.node[0];
    uint32_t *nnxt = nxt ? c.node[1] : c.node[0];
    const int len = 4 * nsplit;
    const int chunk = (len + OCT_NT - 1) / OCT_NT;
    const int i0 = min((int)threadIdx.x * chunk, len), i1 = min(i0 + chunk, len);
    int s = 0, e = 0;
    for (int i = i0; i < i1; ++i) {
        const uint32_t n = c.cc[4 * c.node_of_rank[i >> 2] + (i & 3)];
        s += n > 0;
        e += n > 1;
    }
    const int chunk2 = (size + OCT_NT - 1) / OCT_NT;
    const int j0 = min((int)threadIdx.x * chunk2, size), j1 = min(j0 + chunk2, size);
    int u = 0;
    for (int j = j0; j < j1; ++j) u += !(c.rank[j] < (uint32_t)nsplit);
    // the three prefix sums of a pass in one sweep: non-empty children and children that will split again share a word
    // (both stay below 2^16: at most four per node of an LDS-resident list), the nodes that stay are the second value
    int ci, ui, TE, U;
    block_scan_excl2(s | (e << 16), u, &ci, &ui, &TE, &U, lds, par);
    ci &= 0xFFFF;
    const int T = TE & 0xFFFF, E = TE >> 16;
    for (int i = i0; i < i1; ++i) {
        const int parent = c.node_of_rank[i >> 2];
        const uint32_t n = c.cc[4 * parent + (i & 3)];
        if (n > 0) {
            const int pos = T - 1 - ci;
            nnxt[pos] = (((ncur[parent] >> 24) + 1) << 24) | n;
            c.childpos[i] = (uint16_t)pos;
            ++ci;
        }
    }
    for (int j = j0; j < j1; ++j) {
        if (!(c.rank[j] < (uint32_t)nsplit)) {
            const int pos = T + ui;
            nnxt[pos] = ncur[j];
            c.newpos[j] = (uint16_t)pos;
            ++ui;
        }
    }
    __syncthreads(); // the old table has been read
    const int nsz = T + U;
    for (int i = threadIdx.x; i < 4 * nsz; i += OCT_NT) c.cc[i] = 0;
    __syncthreads();
    auto visit = [&](uint32_t old_node, uint32_t code) -> uint32_t {
        const uint32_t r = c.rank[old_node];
        uint32_t d = ncur[old_node] >> 24, np;
        if (r < (uint32_t)nsplit) {
            np = c.childpos[4 * r + ((code >> ((30 - 2 * d) & 31)) & 3)];
            ++d;
        } else {
            np = c.newpos[old_node];
        }
        atomicAdd(&c.cc[4 * np + ((code >> ((30 - 2 * d) & 31)) & 3)], 1u); // d = 16 only in one-candidate nodes: never read
        return np;
    };
    {
        // the register-resident candidates, phase by phase: the eight reads of a phase are independent of each other, so a
        // pass costs four LDS round trips, not four per candidate (a slot without a candidate reads node 0, harmlessly)
        uint32_t rr[OCT_REG], dd[OCT_REG];
#pragma unroll
        for (int u2 = 0; u2 < OCT_REG; ++u2) rr[u2] = c.rank[rn[u2]];
#pragma unroll
        for (int u2 = 0; u2 < OCT_REG; ++u2) dd[u2] = ncur[rn[u2]] >> 24;
#pragma unroll
        for (int u2 = 0; u2 < OCT_REG; ++u2) {
            const bool sp = rr[u2] < (uint32_t)nsplit;
            const uint16_t *from = sp ? &c.childpos[4 * rr[u2] + ((rc[u2] >> ((30 - 2 * dd[u2]) & 31)) & 3)] : &c.newpos[rn[u2]];
            dd[u2] += sp;
            rr[u2] = *from;
        }
#pragma unroll
        for (int u2 = 0; u2 < OCT_REG; ++u2)
            if ((int)threadIdx.x + u2 * OCT_NT < c.n) {
                rn[u2] = rr[u2];
                atomicAdd(&c.cc[4 * rr[u2] + ((rc[u2] >> ((30 - 2 * dd[u2]) & 31)) & 3)], 1u);
            }
    }
    // the candidates beyond the register-resident ones: OCT_UNROLL global loads per thread are requested before the first
    // is used (clamped index, no branch around the loads), otherwise every candidate costs a full memory latency
    for (int p0 = OCT_REG * OCT_NT + threadIdx.x; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
        uint32_t old[OCT_UNROLL], code[OCT_UNROLL];
#pragma unroll
        for (int u2 = 0; u2 < OCT_UNROLL; ++u2) {
            const int p = min(p0 + u2 * OCT_NT, c.n - 1);
            old[u2] = c.pnode[p];
            code[u2] = c.pcode[p];
        }
#pragma unroll
        for (int u2 = 0; u2 < OCT_UNROLL; ++u2) {
            const int p = p0 + u2 * OCT_NT;
            if (p >= c.n) break;
            c.pnode[p] = visit(old[u2], code[u2]);
        }
    }
    __syncthreads();
    *new_size = nsz;
    *n_expand = E;
}

#ifdef OCT_PYR
// ---------------------------------------------------------------------------------------------
// Passes without a candidate sweep (OCT_PYR builds: a level of a megapixel with 10 000+ candidates in ONE workgroup).
// What a pass needs from the candidates is only how many of them sit in each child quadrant of the nodes it splits.  The
// descent code makes a node of depth d the set of candidates with a given (initial node, first d digits), so ONE sweep that
// counts the candidates per depth-D path, summed four by four up to depth 0, holds every count a pass can ask for as long as
// it splits nodes above depth D: pyr[off(d) + path], off(d) = n_ini (4^d - 1) / 3, path = initial node * 4^d + digits.
// Nodes carry their path; the list bookkeeping (ranks, scans, order of the new list) is octl_apply's, the sweep is gone.
// Once a node of depth D with more than one candidate appears (corners crowded into a small area) the kernel locates every
// candidate's node from a table of the list's paths and carries on with the sweeping passes.
// ---------------------------------------------------------------------------------------------
struct OctPyr {
    uint32_t *tab;      // the count pyramid; later the table (depth, path) -> list position
    uint32_t *path[2];  // path of every node of the two list copies
    uint32_t *xs, *ys;  // a candidate's depth-D path = xs[x] | ys[y]: the node grid is a product grid (see above), so the D
                        // left/right choices depend on x alone (kept at the even bits, under the initial node's index) and
                        // the D up/down choices on y alone (odd bits) -- two look-ups per candidate instead of a descent
    int D, n_ini;
    __device__ __forceinline__ uint32_t off(int d) const { return (uint32_t)n_ini * (((1u << (2 * d)) - 1u) / 3u); }
};

__device__ __forceinline__ void octl_apply_pyr(const OctL &c, const OctPyr &py, int cur, int size, int nsplit, int *new_size,
                                               int *n_expand, int *deep, int *lds, int &par)
{
    const int nxt = cur ^ 1;
    const uint32_t *ncur = cur ? c.node[1] : c.node[0];
    uint32_t *nnxt = nxt ? c.node[1] : c.node[0];
    const uint32_t *pcur = cur ? py.path[1] : py.path[0]; // (selected, not indexed: an indexed pointer pair would live in scratch)
    uint32_t *pnxt = nxt ? py.path[1] : py.path[0];
    const int len = 4 * nsplit;
    const int chunk = (len + OCT_NT - 1) / OCT_NT;
    const int i0 = min((int)threadIdx.x * chunk, len), i1 = min(i0 + chunk, len);
    int s = 0, e = 0;
    for (int i = i0; i < i1; ++i) {
        const uint32_t n = c.cc[4 * c.node_of_rank[i >> 2] + (i & 3)];
        s += n > 0;
        e += n > 1;
    }
    const int chunk2 = (size + OCT_NT - 1) / OCT_NT;
    const int j0 = min((int)threadIdx.x * chunk2, size), j1 = min(j0 + chunk2, size);
    int u = 0;
    for (int j = j0; j < j1; ++j) u += !(c.rank[j] < (uint32_t)nsplit);
    int ci, ui, TE, U;
    block_scan_excl2(s | (e << 16), u, &ci, &ui, &TE, &U, lds, par);
    ci &= 0xFFFF;
    const int T = TE & 0xFFFF, E = TE >> 16;
    int dp = 0;
    for (int i = i0; i < i1; ++i) {
        const int parent = c.node_of_rank[i >> 2];
        const uint32_t n = c.cc[4 * parent + (i & 3)];
        if (n > 0) {
            const int pos = T - 1 - ci;
            const uint32_t d = (ncur[parent] >> 24) + 1;
            nnxt[pos] = (d << 24) | n;
            pnxt[pos] = 4 * pcur[parent] + (i & 3);
            dp |= (int)d >= py.D && n > 1;
            ++ci;
        }
    }
    for (int j = j0; j < j1; ++j) {
        if (!(c.rank[j] < (uint32_t)nsplit)) {
            const int pos = T + ui;
            nnxt[pos] = ncur[j];
            pnxt[pos] = pcur[j];
            ++ui;
        }
    }
    if (dp) *deep = 1; // (cleared by the caller before the pass)
    __syncthreads(); // the old child counts have been read, the new list is complete
    const int nsz = T + U;
    for (int i = threadIdx.x; i < 4 * nsz; i += OCT_NT) {
        const uint32_t nd = nnxt[i >> 2], d = nd >> 24;
        // children of a node of depth D are below the pyramid: such a node is split only after the switch to sweeps
        c.cc[i] = ((int)d < py.D && (nd & 0xFFFFFF) > 1) ? py.tab[py.off((int)d + 1) + 4 * pnxt[i >> 2] + (i & 3)] : 0u;
    }
    __syncthreads();
    *new_size = nsz;
    *n_expand = E;
}
#endif

#ifndef OCT_MIN_WAVES
#define OCT_MIN_WAVES 1
#endif
__global__ __launch_bounds__(OCT_NT, OCT_MIN_WAVES) void k_octree_lds(const OrbxLevels *__restrict__ levels, OrbxBuffers b, int level0, int dyn_lds_bytes)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int lds[64];
    int par = 0; // which half of `lds` the next scan uses
    __shared__ int s_first;

#ifdef OCT_PRIO
    // A call with a few frames waits for this workgroup while the other stream's FAST waves share its CU: measured with the
    // phase stamps, a level-0 quadtree stood still for 49 us -- the run time of the FAST launch beside it -- at default priority.
    __builtin_amdgcn_s_setprio(3);
#endif
    const int level = level0 + blockIdx.x, frame = blockIdx.y;
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
#ifdef OCT_PROF
    unsigned long long *prof = reinterpret_cast<unsigned long long *>(b.best) + (size_t)level * 64;
    int prof_n = 0;
#define OCT_MARK(tag) do { if (frame == 0 && tid == 0 && prof_n < 31) { prof[2 * prof_n] = wall_clock64(); prof[2 * prof_n + 1] = (tag); ++prof_n; } } while (0)
#else
#define OCT_MARK(tag) do { } while (0)
#endif
    OCT_MARK(0);
    const int N = lv.quota, M = lv.node_cap, NSC = max(lv.quota, lv.n_ini) + 4;
    int P2 = 1;
    while (P2 < N) P2 <<= 1;
    u64 *sort_keys = reinterpret_cast<u64 *>(smem);
    c.node[0] = reinterpret_cast<uint32_t *>(sort_keys + P2 + 16);
    c.node[1] = c.node[0] + M;
    c.cc = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(c.node[1] + M) + 15) & ~(uintptr_t)15); // read four at a time
    c.rank = reinterpret_cast<uint16_t *>(c.cc + 4 * M);
    c.newpos = c.rank + M;
    c.node_of_rank = c.newpos + M;
    c.childpos = c.node_of_rank + NSC;
#ifdef OCT_PYR
    // what the launch gave beyond the list's own arrays: node paths and the count pyramid, as deep as fits (at most 6 digits:
    // a level of 2000 features is done at depth 4-5)
    OctPyr py;
    bool pyr_mode = false;
    __shared__ int s_deep;
    {
        unsigned char *extra = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(c.childpos + 4 * NSC) + 15) & ~(uintptr_t)15);
        const int nxy = lv.w + lv.h; // candidate coordinates are below the level's size whatever their origin
        const long left = (long)dyn_lds_bytes - (long)(extra - smem) - 8L * M - 4L * nxy;
        py.path[0] = reinterpret_cast<uint32_t *>(extra);
        py.path[1] = py.path[0] + M;
        py.xs = py.path[1] + M;
        py.ys = py.xs + lv.w;
        py.tab = py.ys + lv.h;
        py.n_ini = lv.n_ini;
        py.D = 0;
        for (int d = 6; d >= 2; --d)
            if ((long)lv.n_ini * (long)(((1u << (2 * d + 2)) - 1u) / 3u) * 4L <= left) { py.D = d; break; }
        pyr_mode = py.D >= 2;
        if (tid == 0) s_deep = 0;
    }
#endif

    // ---- descent code of every candidate + initial nodes (:645-686)
    int cur = 0;
    for (int i = tid; i < lv.n_ini; i += OCT_NT) c.cc[i] = 0;
    // A node one pixel wide and high holds one candidate and is never divided: the descent stops being interesting after
    // as many halvings as the larger side of an initial node needs to get there (10 for a 1242 x 375 frame, not 16).
    const int last_w = (lv.w - ORBX_EDGE) - lv.h_x * (lv.n_ini - 1);
    int depth = 1;
    for (int side = max(max(lv.h_x, last_w), lv.region_h); side > 1 && depth < 16; side = (side + 1) >> 1) ++depth;
    const uint32_t inv_hx = 0xFFFFFFFFu / (uint32_t)lv.h_x + 1u; // x / h_x = mulhi(x, inv_hx) for 16-bit x
    uint32_t rn[OCT_REG], rc[OCT_REG];
    u64 rcand[OCT_REG];
#pragma unroll
    for (int u = 0; u < OCT_REG; ++u) rcand[u] = c.cand[min(tid + u * OCT_NT, c.n - 1)];
    // initial node and the first `ndig` digits of the descent of one candidate
    auto descend = [&](u64 cd, int ndig, uint32_t *code_out) -> uint32_t {
        const int x = (int)(cd & 0xFFFF), y = (int)((cd >> 16) & 0xFFFF);
        const int idx = (int)__umulhi((uint32_t)x, inv_hx);
        int ulx = lv.h_x * idx, brx = (idx == lv.n_ini - 1) ? (lv.w - ORBX_EDGE) : lv.h_x * (idx + 1); // :665
        int uly = 0, bry = lv.region_h;
        uint32_t code = 0;
        for (int d = 0; d < ndig; ++d) {
            const int midx = ulx + ((brx - ulx) >> 1), midy = uly + ((bry - uly) >> 1); // DivideNode :368-369 (sides are >= 0)
            const int qx = x >= midx, qy = y >= midy;                                   // :397-407
            ulx = qx ? midx : ulx; brx = qx ? brx : midx;
            uly = qy ? midy : uly; bry = qy ? bry : midy;
            code |= (uint32_t)(qx | (qy << 1)) << (30 - 2 * d);
        }
        *code_out = code;
        return (uint32_t)idx;
    };
    // Pass k (counting from 0) reads digits up to k + 1, and a KITTI level is done after four passes: the codes start with
    // OCT_DIGITS0 digits and are recomputed in full in the rare call that goes deeper (corners crowded into a small area).
    int have = min(depth, OCT_DIGITS0), applies = 0;
    auto more_digits = [&]() {
        if (applies + 2 <= have) return;
        have = depth;
#pragma unroll
        for (int u = 0; u < OCT_REG; ++u)
            if (tid + u * OCT_NT < c.n) (void)descend(rcand[u], have, &rc[u]);
        for (int p = OCT_REG * OCT_NT + tid; p < c.n; p += OCT_NT) {
            uint32_t code;
            (void)descend(c.cand[p], have, &code);
            c.pcode[p] = code;
        }
    };
    __syncthreads();
    int size;
#ifdef OCT_PYR
    // Node of a candidate in the current list: the table (depth, path) -> list position takes the pyramid's place, every
    // node's position is handed down to the depth-D paths below it (the list's nodes partition the candidates, so a depth-D
    // path lies under exactly one of them), and a candidate needs ONE look-up.
    auto path_table = [&]() { // (called with the list complete and every reader of the pyramid past a barrier)
        const uint32_t *nc = cur ? c.node[1] : c.node[0];
        const int total = (int)py.off(py.D + 1);
        for (int i = tid; i < total; i += OCT_NT) py.tab[i] = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t *pc = cur ? py.path[1] : py.path[0];
        for (int j = tid; j < size; j += OCT_NT) py.tab[py.off((int)(nc[j] >> 24)) + pc[j]] = (uint32_t)j;
        __syncthreads();
        for (int d = 1; d <= py.D; ++d) {
            const int ne = lv.n_ini << (2 * d);
            const uint32_t *above = py.tab + py.off(d - 1);
            uint32_t *here = py.tab + py.off(d);
            for (int e = tid; e < ne; e += OCT_NT)
                if (here[e] == 0xFFFFFFFFu) here[e] = above[e >> 2];
            __syncthreads();
        }
    };
    const uint32_t *leaf_of = py.tab + py.off(py.D);
    auto locate = [&](u64 cd, int ndig, uint32_t *code_out) -> uint32_t { // ... and the descent code with ndig >= D digits
        const uint32_t idx = descend(cd, ndig, code_out);
        return leaf_of[(idx << (2 * py.D)) + (*code_out >> (32 - 2 * py.D))];
    };
    auto locate_xy = [&](u64 cd) -> uint32_t { // the same from the two path tables (no code: the finish needs none)
        return leaf_of[py.xs[(uint32_t)cd & 0xFFFF] | py.ys[(uint32_t)(cd >> 16) & 0xFFFF]];
    };
    if (pyr_mode) {
        const int total = (int)py.off(py.D + 1);
        const uint32_t offD = py.off(py.D);
        for (int i = tid; i < total; i += OCT_NT) py.tab[i] = 0;
        for (int x = tid; x < lv.w; x += OCT_NT) { // (the same halvings as `descend`, one axis at a time)
            const int idx = (int)__umulhi((uint32_t)x, inv_hx);
            int lo = lv.h_x * idx, hi = (idx == lv.n_ini - 1) ? (lv.w - ORBX_EDGE) : lv.h_x * (idx + 1);
            uint32_t bits = (uint32_t)idx << (2 * py.D);
            for (int d = 0; d < py.D; ++d) {
                const int mid = lo + ((hi - lo) >> 1), q = x >= mid;
                lo = q ? mid : lo; hi = q ? hi : mid;
                bits |= (uint32_t)q << (2 * (py.D - 1 - d));
            }
            py.xs[x] = bits;
        }
        for (int y = tid; y < lv.h; y += OCT_NT) {
            int lo = 0, hi = lv.region_h;
            uint32_t bits = 0;
            for (int d = 0; d < py.D; ++d) {
                const int mid = lo + ((hi - lo) >> 1), q = y >= mid;
                lo = q ? mid : lo; hi = q ? hi : mid;
                bits |= (uint32_t)q << (2 * (py.D - 1 - d) + 1);
            }
            py.ys[y] = bits;
        }
        __syncthreads();
        auto count = [&](u64 cd) {
            atomicAdd(&py.tab[offD + (py.xs[(uint32_t)cd & 0xFFFF] | py.ys[(uint32_t)(cd >> 16) & 0xFFFF])], 1u);
        };
#pragma unroll
        for (int u = 0; u < OCT_REG; ++u) {
            rn[u] = 0; rc[u] = 0;
            if (tid + u * OCT_NT < c.n) count(rcand[u]);
        }
        for (int p0 = OCT_REG * OCT_NT + tid; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
            u64 cdv[OCT_UNROLL];
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u) cdv[u] = c.cand[min(p0 + u * OCT_NT, c.n - 1)];
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u)
                if (p0 + u * OCT_NT < c.n) count(cdv[u]);
        }
        __syncthreads();
        for (int d = py.D - 1; d >= 0; --d) {
            const int ne = lv.n_ini << (2 * d);
            const uint32_t *below = py.tab + py.off(d + 1);
            uint32_t *here = py.tab + py.off(d);
            for (int e = tid; e < ne; e += OCT_NT) here[e] = below[4 * e] + below[4 * e + 1] + below[4 * e + 2] + below[4 * e + 3];
            __syncthreads();
        }
        const int chunk = (lv.n_ini + OCT_NT - 1) / OCT_NT;
        const int i0 = min(tid * chunk, lv.n_ini), i1 = min(i0 + chunk, lv.n_ini);
        int s = 0;
        for (int i = i0; i < i1; ++i) s += py.tab[i] > 0;
        int pos = block_scan_excl(s, &size, lds, par);
        for (int i = i0; i < i1; ++i)
            if (py.tab[i] > 0) {
                c.node[0][pos] = py.tab[i]; // depth 0
                py.path[0][pos] = (uint32_t)i;
                ++pos;
            }
        __syncthreads();
        for (int i = tid; i < 4 * size; i += OCT_NT) c.cc[i] = py.tab[py.off(1) + 4 * py.path[0][i >> 2] + (i & 3)];
        __syncthreads();
    } else
#endif
    {
#pragma unroll
    for (int u = 0; u < OCT_REG; ++u) {
        rn[u] = 0; rc[u] = 0;
        if (tid + u * OCT_NT < c.n) {
            rn[u] = descend(rcand[u], have, &rc[u]);
            atomicAdd(&c.cc[rn[u]], 1u);
        }
    }
    for (int p0 = OCT_REG * OCT_NT + tid; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
        u64 cdv[OCT_UNROLL];
#pragma unroll
        for (int u = 0; u < OCT_UNROLL; ++u) cdv[u] = c.cand[min(p0 + u * OCT_NT, c.n - 1)];
#pragma unroll
        for (int u = 0; u < OCT_UNROLL; ++u) {
            const int p = p0 + u * OCT_NT;
            if (p >= c.n) break;
            uint32_t code;
            const uint32_t idx = descend(cdv[u], have, &code);
            c.pnode[p] = idx;
            c.pcode[p] = code;
            atomicAdd(&c.cc[idx], 1u);
        }
    }
    __syncthreads();
    {
        const int chunk = (lv.n_ini + OCT_NT - 1) / OCT_NT;
        const int i0 = min(tid * chunk, lv.n_ini), i1 = min(i0 + chunk, lv.n_ini);
        int s = 0;
        for (int i = i0; i < i1; ++i) s += c.cc[i] > 0;
        int pos = block_scan_excl(s, &size, lds, par);
        for (int i = i0; i < i1; ++i)
            if (c.cc[i] > 0) {
                c.node[0][pos] = c.cc[i]; // depth 0
                c.newpos[i] = (uint16_t)pos++;
            }
        __syncthreads();
        for (int i = tid; i < 4 * size; i += OCT_NT) c.cc[i] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < OCT_REG; ++u)
            if (tid + u * OCT_NT < c.n) {
                rn[u] = c.newpos[rn[u]];
                atomicAdd(&c.cc[4 * rn[u] + (rc[u] >> 30)], 1u);
            }
        for (int p0 = OCT_REG * OCT_NT + tid; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
            uint32_t pn[OCT_UNROLL], code[OCT_UNROLL];
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u) {
                const int p = min(p0 + u * OCT_NT, c.n - 1);
                pn[u] = c.pnode[p];
                code[u] = c.pcode[p];
            }
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u)
                if (p0 + u * OCT_NT < c.n) {
                    const uint32_t np = c.newpos[pn[u]];
                    c.pnode[p0 + u * OCT_NT] = np;
                    atomicAdd(&c.cc[4 * np + (code[u] >> 30)], 1u);
                }
        }
        __syncthreads();
    }
    }
    OCT_MARK(1);
#ifdef OCT_PYR
    // the switch to sweeping passes: every candidate finds its node (and its full descent code), the child counts of the
    // whole list are counted once
    auto to_sweeps = [&]() {
        path_table();
        have = max(depth, py.D);
        const uint32_t *nc = cur ? c.node[1] : c.node[0];
        for (int i = tid; i < 4 * size; i += OCT_NT) c.cc[i] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < OCT_REG; ++u)
            if (tid + u * OCT_NT < c.n) {
                rn[u] = locate(rcand[u], have, &rc[u]);
                atomicAdd(&c.cc[4 * rn[u] + ((rc[u] >> ((30 - 2 * (nc[rn[u]] >> 24)) & 31)) & 3)], 1u);
            }
        for (int p = OCT_REG * OCT_NT + tid; p < c.n; p += OCT_NT) {
            uint32_t code;
            const uint32_t nd = locate(c.cand[p], have, &code);
            c.pnode[p] = nd;
            c.pcode[p] = code;
            atomicAdd(&c.cc[4 * nd + ((code >> ((30 - 2 * (nc[nd] >> 24)) & 31)) & 3)], 1u);
        }
        __syncthreads();
        pyr_mode = false;
    };
    // one pass in whichever form the level is in
    auto apply = [&](int nsplit, int *n_expand) {
        if (pyr_mode) {
            octl_apply_pyr(c, py, cur, size, nsplit, &size, n_expand, &s_deep, lds, par);
            cur ^= 1; ++applies;
            if (s_deep) to_sweeps(); // (s_deep is written before the barriers of the pass and never cleared: the switch is final)
        } else {
            more_digits();
            octl_apply(c, cur, size, nsplit, &size, n_expand, lds, par, rn, rc);
            cur ^= 1; ++applies;
        }
    };
#else
    auto apply = [&](int nsplit, int *n_expand) {
        more_digits();
        octl_apply(c, cur, size, nsplit, &size, n_expand, lds, par, rn, rc);
        cur ^= 1; ++applies;
    };
#endif

    // ---- main rounds (:692-751)
    bool finish = false;
    while (!finish) {
        const int pre = size;
        const uint32_t *ncur = cur ? c.node[1] : c.node[0];
        const int chunk = (size + OCT_NT - 1) / OCT_NT;
        const int j0 = min(tid * chunk, size), j1 = min(j0 + chunk, size);
        int s = 0;
        for (int j = j0; j < j1; ++j) s += (ncur[j] & 0xFFFFFF) > 1;
        int nsplit;
        int r = block_scan_excl(s, &nsplit, lds, par);
        for (int j = j0; j < j1; ++j) {
            if ((ncur[j] & 0xFFFFFF) > 1) { c.rank[j] = (uint16_t)r; c.node_of_rank[r] = (uint16_t)j; ++r; }
            else c.rank[j] = OCT_NORANK;
        }
        __syncthreads();
        OCT_MARK(2);
        int n_expand;
        apply(nsplit, &n_expand);
        OCT_MARK(4 + (size << 8));
        if (size > N || size == pre) {
            finish = true;
        } else if (size + 3 * n_expand > N) {
            // ---- final phase (:752-809)
            while (!finish) {
                const int pre2 = size;
                const uint32_t *nc2 = cur ? c.node[1] : c.node[0];
                const int ch2 = (size + OCT_NT - 1) / OCT_NT;
                const int a0 = min(tid * ch2, size), a1 = min(a0 + ch2, size);
                int k = 0;
                for (int j = a0; j < a1; ++j) k += (nc2[j] & 0xFFFFFF) > 1;
                int K;
                int ko = block_scan_excl(k, &K, lds, par);
                for (int j = a0; j < a1; ++j) {
                    c.rank[j] = OCT_NORANK;
                    const uint32_t cn = nc2[j] & 0xFFFFFF;
                    if (cn > 1) sort_keys[ko++] = ((u64)cn << 32) | (u64)(0xFFFFFFFFu - (uint32_t)j);
                }
                if (K <= OCT_RANK_BY_COUNTING) {
                    // few keys (all different: they contain the list position): the place of a key in the sorted order is
                    // the number of smaller keys -- K broadcast reads per thread, pipelined, against the 36 dependent
                    // LDS round trips of a 256-key bitonic network
                    if (tid < 16) sort_keys[K + tid] = ~0ull; // the reads below go sixteen keys at a time (the region is padded)
                    __syncthreads();
                    if (tid < K) {
                        const u64 mine = sort_keys[tid];
                        int below = 0;
                        for (int i = 0; i < K; i += 16) {
                            ulonglong2 two[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) two[q] = *reinterpret_cast<const ulonglong2 *>(&sort_keys[i + 2 * q]);
#pragma unroll
                            for (int q = 0; q < 8; ++q) below += (two[q].x < mine) + (two[q].y < mine);
                        }
                        const int pos = (int)(0xFFFFFFFFu - (uint32_t)(mine & 0xFFFFFFFFu));
                        c.rank[pos] = (uint16_t)below;
                        c.node_of_rank[below] = (uint16_t)pos;
                    }
                } else {
                    int P = 1;
                    while (P < K) P <<= 1;
                    for (int i = K + tid; i < P; i += OCT_NT) sort_keys[i] = ~0ull;
                    __syncthreads();
                    // Bitonic network.  Element i belongs to thread i mod 512, so a wave owns 64 consecutive elements of
                    // every 512-block: exchanges at distance < 64 stay inside the wave (its LDS operations complete in
                    // order) and need no workgroup barrier -- 6 barriers instead of 45 for 512 keys.
                    for (int kk = 2; kk <= P; kk <<= 1)
                        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                            for (int i = tid; i < P; i += OCT_NT) {
                                const int ixj = i ^ jj;
                                if (ixj > i) {
                                    const u64 x = sort_keys[i], y = sort_keys[ixj];
                                    if ((x > y) == ((i & kk) == 0)) { sort_keys[i] = y; sort_keys[ixj] = x; }
                                }
                            }
                            const int next_jj = jj > 1 ? (jj >> 1) : kk; // the first distance of the next stage is kk
                            if (jj >= 64 || next_jj >= 64 || (jj == 1 && kk == P)) __syncthreads();
                            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    for (int sidx = tid; sidx < K; sidx += OCT_NT) {
                        const int pos = (int)(0xFFFFFFFFu - (uint32_t)(sort_keys[sidx] & 0xFFFFFFFFu));
                        c.rank[pos] = (uint16_t)sidx;
                        c.node_of_rank[sidx] = (uint16_t)pos;
                    }
                }
                if (tid == 0) s_first = K;
                __syncthreads();
                OCT_MARK(5 + (K << 8));
                const int ch3 = (K + OCT_NT - 1) / OCT_NT;
                const int b0 = min(tid * ch3, K), b1 = min(b0 + ch3, K);
                int g = 0;
                for (int i = b0; i < b1; ++i) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(&c.cc[4 * c.node_of_rank[i]]);
                    g += (q.x > 0) + (q.y > 0) + (q.z > 0) + (q.w > 0) - 1;
                }
                int G;
                int acc = size + block_scan_excl(g, &G, lds, par);
                for (int i = b0; i < b1; ++i) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(&c.cc[4 * c.node_of_rank[i]]);
                    acc += (q.x > 0) + (q.y > 0) + (q.z > 0) + (q.w > 0) - 1;
                    if (acc >= N) { atomicMin(&s_first, i); break; }
                }
                __syncthreads();
                const int nsplit2 = min(s_first + 1, K);
                __syncthreads();
                OCT_MARK(7);
                int ne2;
                apply(nsplit2, &ne2);
                OCT_MARK(8 + (size << 8));
                if (size >= N || size == pre2) finish = true;
            }
        }
    }

    OCT_MARK(9);
    // ---- strongest point per node (:812-827); the node arrays are dead now and hold the maxima
    u64 *best = reinterpret_cast<u64 *>(c.node[0]);
    __syncthreads();
#ifdef OCT_PYR
    if (pyr_mode) path_table(); // no pass has swept the candidates: they find their nodes now
#endif
    for (int j = tid; j < size; j += OCT_NT) best[j] = 0;
    __syncthreads();
    const uint32_t ncols = (uint32_t)lv.n_cols;
    auto offer = [&](u64 cd, uint32_t node) {
        const uint32_t x = (uint32_t)(cd & 0xFFFF), y = (uint32_t)((cd >> 16) & 0xFFFF), resp = (uint32_t)(cd >> 32);
        const uint32_t order = ((y / ORBX_CELL) * ncols + x / ORBX_CELL) * (ORBX_CELL * ORBX_CELL) +
                               (y % ORBX_CELL) * ORBX_CELL + x % ORBX_CELL;
        atomicMax(&best[node], ((u64)resp << 32) | (u64)(0xFFFFFFFFu - order));
    };
#ifdef OCT_PYR
    if (pyr_mode) {
#pragma unroll
        for (int u = 0; u < OCT_REG; ++u)
            if (tid + u * OCT_NT < c.n) offer(rcand[u], locate_xy(rcand[u]));
        for (int p0 = OCT_REG * OCT_NT + tid; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
            u64 cdv[OCT_UNROLL];
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u) cdv[u] = c.cand[min(p0 + u * OCT_NT, c.n - 1)];
#pragma unroll
            for (int u = 0; u < OCT_UNROLL; ++u)
                if (p0 + u * OCT_NT < c.n) offer(cdv[u], locate_xy(cdv[u]));
        }
    } else
#endif
    {
#pragma unroll
    for (int u = 0; u < OCT_REG; ++u)
        if (tid + u * OCT_NT < c.n) offer(rcand[u], rn[u]);
    for (int p0 = OCT_REG * OCT_NT + tid; p0 < c.n; p0 += OCT_UNROLL * OCT_NT) {
        u64 cdv[OCT_UNROLL];
        uint32_t pn[OCT_UNROLL];
#pragma unroll
        for (int u = 0; u < OCT_UNROLL; ++u) {
            const int p = min(p0 + u * OCT_NT, c.n - 1);
            cdv[u] = c.cand[p];
            pn[u] = c.pnode[p];
        }
#pragma unroll
        for (int u = 0; u < OCT_UNROLL; ++u)
            if (p0 + u * OCT_NT < c.n) offer(cdv[u], pn[u]);
    }
    }
    __syncthreads();
    uint2 *sel = b.sel + (size_t)frame * levels->kcap_total + lv.kp_off;
    const int n_out = min(size, lv.kcap);
    for (int j = tid; j < n_out; j += OCT_NT) {
        const u64 k = best[j];
        const uint32_t order = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu);
        const uint32_t cell = order / (ORBX_CELL * ORBX_CELL), in = order % (ORBX_CELL * ORBX_CELL);
        const uint32_t x = (cell % ncols) * ORBX_CELL + in % ORBX_CELL + ORBX_EDGE;
        const uint32_t y = (cell / ncols) * ORBX_CELL + in / ORBX_CELL + ORBX_EDGE;
        sel[j] = make_uint2(x | (y << 16), (uint32_t)(k >> 32));
    }
    if (tid == 0) *out_count = n_out;
    OCT_MARK(10 + (c.n << 8));
#ifdef OCT_PROF
    if (frame == 0 && tid == 0) prof[63] = prof_n;
#endif
}
