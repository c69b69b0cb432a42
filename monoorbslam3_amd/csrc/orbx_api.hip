// C-ABI host side of the MI355X ORB extractor (see include/orbx.h).
//
// Mirrors the reference's ORBExtractor (modules/ORB/ORBExtractor.{h,cpp}): construction builds the
// scale tables, per-level quotas and the circular-patch row table exactly as the constructor at
// ORBExtractor.cpp:424-475 does; extract enqueues the HIP kernels of orbx_kernels.hip.  There is no
// CPU implementation of the pixel path in this library: without a HIP device every call fails.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>

#include "orb_math.h"
#include "orbx_internal.h"

#ifndef ORBX_HANDLE_STREAM_FLAGS // (a build with hipStreamDefault is round 5's blocking handle stream: tools/ab_latency.sh, tools/two_thread_latency.sh)
#define ORBX_HANDLE_STREAM_FLAGS hipStreamNonBlocking
#endif
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
// shared with the matcher translation unit (orbm_matcher.hip)
int orbx_set_error(int code, const std::string &msg) { return fail(code, msg); }
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ORBX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

hipError_t orbx_lds_opt_in(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> configured;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu); // read, set and store as one step: a smaller size can never overwrite a larger one
    size_t &have = configured[std::make_pair(kernel, dev)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

struct orbx_ctx {
    orbx_cfg cfg;
    int device;
    int n_cus = ORBX_N_CUS; // the device's CU count (hipDeviceAttributeMultiprocessorCount at create): decides whether a few-frames
                            // quadtree launch gives every workgroup a CU of its own (orbx_octree_plan)
    // reference tables (ORBExtractor.h:109-121)
    float scale_factors[ORBX_MAX_LEVELS], inv_scale_factors[ORBX_MAX_LEVELS];
    float square_sigmas[ORBX_MAX_LEVELS], inv_square_sigmas[ORBX_MAX_LEVELS];
    float log_scale_factor;
    int quotas[ORBX_MAX_LEVELS];
    int u_max[ORBX_HALF_PATCH + 1];
    int taps[7];
    // geometry of the current frame size
    int cur_w, cur_h;
    OrbxLevels levels;
    size_t l0_stage_pitch; // staging copy of level 0 for the host-pointer API
    // device state
    hipStream_t stream;
    bool null_pending = false; // a device call was enqueued on stream 0 (NULL) since the last host-side wait
    OrbxBuffers buf;
    OrbxLevels *d_levels;
    OrbxTap *d_xtap[ORBX_MAX_LEVELS], *d_ytap[ORBX_MAX_LEVELS];
    bool resize2_ok[ORBX_MAX_LEVELS]; // [l]: levels l and l + 1 can come out of one launch (k_resize2's patch fits)
    bool resize_lds_ok[ORBX_MAX_LEVELS]; // [l]: level l can be made by k_resize_lds (its source tiles fit)
    int resize_lds;                   // ORBX_VAR_RESIZE_LDS: 0 = never, 1 = resident batches (default), 2 = always
    int resize2;                      // ORBX_VAR_RESIZE2: 0 = never two levels per launch, 1 = calls with few frames (default), 2 = always
    int *d_umax, *d_taps;
    hipEvent_t ev_after_fast; bool after_fast_valid; // recorded behind the FAST launches of every call (orbx_stream_wait_fast)
    // timing mode 2: events around the (up to two) launch groups of every stage of a step, on the stream they are launched on
    hipEvent_t ev_in[ORBX_N_STAGES][4]; int ev_in_n[ORBX_N_STAGES];
    uint16_t *d_fast_cells; int n_fast_cells;
    uint16_t *d_fast_strips; int n_fast_strips, n_fast_strips0; // strips of all levels / of level 0
    int fast_variant;                                           // ORBX_VAR_FAST: 0 = by call size (default), 1 = one wave per cell, 2 = strips
    int zero_copy;                                              // ORBX_VAR_ZERO_COPY
    int fast_cell_group;                                        // ORBX_VAR_FAST_CELL_GROUP: cells (one wave each) per workgroup of the few-frames FAST kernel: 1, 4, 8 (default) or 16
    int desc_variant;                                           // ORBX_VAR_DESC: 0 = by call size, 1 = blur pass + k_orient_desc, 2 = k_blur_desc (fused)
    uint16_t *d_blur_tiles; int n_blur_tiles;
    // the Gaussian on the matrix pipe: strip list, band tables, per-level record; levels [0, blur_mfma_levels)
    uint16_t *d_blur_strips; uint8_t *d_band_h, *d_band_v;
    BlurMfmaLevels blur_tab; int blur_strips_before[ORBX_MAX_LEVELS + 1]; int blur_mfma_levels;
    // blur + descriptors in one pass (k_blur_desc): block list, H bands, bucket layout; levels [0, bd_levels)
    uint16_t *d_bd_blocks; uint8_t *d_bd_band_h; int n_bd_blocks, bd_levels, bd_bk_stride;
    BdLevels bd_tab;
    int *d_bd_bk_start; uint8_t *d_bd_items;                    // per frame: bucket starts, 32-byte key-point records
    int alloc_bd_batch, alloc_bd_kcap, alloc_bd_stride;
    int last_fused_levels;                                      // levels of the last call whose blurred copy was never written
    int blur_mfma;                                              // ORBX_VAR_BLUR: 1 = by call size (default), 0 = VALU kernels, 2 = matrix pipe for any batch
    uint8_t *d_l0_stage; size_t l0_stage_fs;
    int *d_slot_level;                                          // level of every key-point slot of the current geometry
    // host-API output staging, one device block: [counts, 256 B aligned][key points][descriptors]; `h_out_block` is its
    // pinned mirror while the block is small (a few frames): the records then come back in one copy and one wait
    uint8_t *d_out_block; uint8_t *h_out_block; uint8_t *h_out_dev; size_t out_block_bytes, out_kp_off, out_desc_off;
    orbx_kp *d_out_kp; uint8_t *d_out_desc; int32_t *d_out_n; int out_cap;
    // capacities actually allocated
    int alloc_batch;
    size_t alloc_img_fs, alloc_cand_fs, alloc_node_fs, alloc_l0_fs;
    int alloc_kcap_total, alloc_out_cap;
    size_t sort_lds_bytes;
    // last call (for the taps)
    const uint8_t *last_l0; size_t last_l0_fs; int last_l0_pitch; int last_frames;
    // sub-batch streams: independent frame ranges run on their own streams so that latency-bound kernels of one
    // range overlap the issue-bound kernels of another
    int n_sub;
    hipStream_t sub[8];
    hipEvent_t ev_fork, ev_join[8];
    // The blur only needs the pyramid, FAST -> quadtree -> orientation only the raw levels: inside a frame range the
    // blur runs on a side stream next to that chain and joins before the descriptor kernel.  Slot 8 serves the
    // unsplit (small batch) case.
    int side_blur, split_level0;
    hipStream_t side[9];
    hipEvent_t ev_pyr[9], ev_blur[9];
    // FAST on level 0 needs no pyramid: its cells start on the side stream at once, next to the (latency-bound) resize
    // chain; the remaining levels' cells follow on the main stream
    int early_fast;
    hipEvent_t ev_start[9], ev_fast0[9];
    // stage timing
    int timing;
    hipEvent_t ev[ORBX_N_STAGES + 1];
    bool ev_valid;
};

// ------------------------------------------------------------------------------------------------
// tables -- reference ORBExtractor.cpp:424-475
// ------------------------------------------------------------------------------------------------
// Host-side wait for everything enqueued through the handle: its own stream and, when a device call was given a NULL stream,
// stream 0 (include/orbx.h, "Streams").
static hipError_t sync_handle(orbx_ctx *c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && c->null_pending) { e = hipStreamSynchronize((hipStream_t)0); c->null_pending = false; }
    return e;
}

static void compute_quotas(orbx_ctx *c, int n_features)
{
    // :443-452.  pow(float,int) is the double overload; the quotient is rounded once to float.
    const float sf = c->cfg.scale_factor;
    float inv2 = 1.0f / (sf * sf);
    float num = (float)n_features * (1 - inv2);
    float nd = (float)((double)num / (1.0 - pow((double)inv2, (double)c->cfg.n_levels)));
    int sum = 0;
    for (int l = 0; l < c->cfg.n_levels - 1; ++l) {
        c->quotas[l] = orb_round_f(nd);
        sum += c->quotas[l];
        nd *= inv2;
    }
    c->quotas[c->cfg.n_levels - 1] = std::max(n_features - sum, 1);
}

static void compute_tables(orbx_ctx *c)
{
    const int L = c->cfg.n_levels;
    const float sf = c->cfg.scale_factor;
    c->log_scale_factor = logf(sf);
    c->scale_factors[0] = c->inv_scale_factors[0] = c->square_sigmas[0] = c->inv_square_sigmas[0] = 1.f;
    for (int i = 1; i < L; ++i) {
        c->scale_factors[i] = c->scale_factors[i - 1] * sf;
        c->inv_scale_factors[i] = 1.f / c->scale_factors[i];
        c->square_sigmas[i] = c->scale_factors[i] * c->scale_factors[i];
        c->inv_square_sigmas[i] = 1.f / c->square_sigmas[i];
    }
    compute_quotas(c, c->cfg.n_features);
    // :460-474
    const int HP = ORBX_HALF_PATCH;
    int v, v0;
    const int vmax = orb_floor_f((float)HP * sqrtf(2.f) / 2 + 1);
    const int vmin = orb_ceil_f((float)HP * sqrtf(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) c->u_max[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (c->u_max[v0] == c->u_max[v0 + 1]) ++v0;
        c->u_max[v] = v0;
        ++v0;
    }
    static const int t0[7] = {18, 34, 48, 56, 48, 34, 18}, t1[7] = {18, 34, 49, 55, 49, 34, 18};
    memcpy(c->taps, c->cfg.blur_variant == 1 ? t1 : t0, sizeof t0);
}

static void level_size(const orbx_ctx *c, int w0, int h0, int l, int *w, int *h)
{
    if (l == 0) { *w = w0; *h = h0; return; }
    const float s = c->inv_scale_factors[l]; // :563-564
    *w = orb_round_f((float)w0 * s);
    *h = orb_round_f((float)h0 * s);
}

// cv::resize INTER_LINEAR tap table for one axis
static void linear_taps(int dn, int sn, bool clamp_ofs, std::vector<OrbxTap> &out)
{
    out.resize(dn);
    const double inv_scale = (double)dn / sn;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = orb_floor_f(f);
        f -= (float)s;
        if (clamp_ofs) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; }
        }
        auto sat = [](int v) { return (int16_t)std::min(std::max(v, -32768), 32767); };
        out[d].ofs = s;
        out[d].c0 = sat(orb_round_f((1.f - f) * 2048.f));
        out[d].c1 = sat(orb_round_f(f * 2048.f));
    }
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// ORBX_VAR_DESC = 0: calls with at least this many level-0 pixels take k_blur_desc.  A workgroup of that kernel walks down a whole
// level, so a small call is bounded by its longest walk, and the blur pass it replaces hides beside FAST on a side stream:
// measured at 1242x375 (tools/desc_crossover.sh) the one-pass kernel loses up to 96 frames (161 k against 165 k frames/s) and
// wins from 128 (174.5 k against 168.4 k; 512 frames: 207 k against 197 k).
// synchronous calls with a few frames: level-0 size from which the side chain (level 0's FAST and quadtree) is issued first
#ifndef ORBX_SIDE_FIRST_PIXELS
#define ORBX_SIDE_FIRST_PIXELS 1200000ull
#endif
#ifndef ORBX_FUSED_DESC_MIN_PIXELS
#define ORBX_FUSED_DESC_MIN_PIXELS 55000000ull
#endif

struct Geometry {
    OrbxLevels levels;
    size_t img_fs, cand_fs, node_fs, l0_fs;
    int l0_pitch;
    size_t sort_lds;
};

static int compute_geometry(const orbx_ctx *c, int w0, int h0, Geometry *g)
{
    memset(g, 0, sizeof *g);
    OrbxLevels &LV = g->levels;
    LV.n_levels = c->cfg.n_levels;
    LV.ini_th = c->cfg.ini_th_fast;
    LV.min_th = c->cfg.min_th_fast;
    size_t img = 0, cand = 0, node = 0;
    int kp = 0, max_quota = 1;
    for (int l = 0; l < LV.n_levels; ++l) {
        OrbxLevel &v = LV.lv[l];
        level_size(c, w0, h0, l, &v.w, &v.h);
        if (v.w < 1 || v.h < 1) return fail(ORBX_E_ARG, "pyramid level collapses to zero size");
        if (v.w > 32000 || v.h > 32000) return fail(ORBX_E_UNSUPPORTED, "image side above 32000 px");
        v.pitch = (int)align_up(v.w, 64);
        v.region_w = v.w - 2 * ORBX_EDGE;
        v.region_h = v.h - 2 * ORBX_EDGE;
        v.quota = c->quotas[l];
        v.scale = c->scale_factors[l];
        max_quota = std::max(max_quota, v.quota);
        if (v.region_w > 0 && v.region_h > 0) {
            v.n_cols = (v.region_w + ORBX_CELL - 1) / ORBX_CELL;
            v.n_rows = (v.region_h + ORBX_CELL - 1) / ORBX_CELL;
            // reference :645-646
            v.n_ini = orb_ceil_f((float)v.region_w / (float)v.region_h);
            v.h_x = orb_ceil_f((float)v.region_w / (float)v.n_ini);
            // at most one strict 3x3 maximum per 2x2 block inside a cell
            const int fullc = v.region_w / ORBX_CELL, remc = v.region_w % ORBX_CELL;
            const int fullr = v.region_h / ORBX_CELL, remr = v.region_h % ORBX_CELL;
            const int colsum = fullc * 15 + (remc + 1) / 2, rowsum = fullr * 15 + (remr + 1) / 2;
            v.cand_cap = colsum * rowsum;
            v.kcap = std::max(v.quota + 3, 4 * v.n_ini);
            // list-size bound: the first main round makes <= 4*nIni nodes; every later main round starts from
            // size + 3*nToExpand <= quota and so ends <= quota; the final phase stops at <= quota + 2
            v.node_cap = std::max(v.quota + 3, 4 * v.n_ini) + 8;
        } else { // level too small for the 19-px border: yields nothing (the reference misbehaves here)
            v.n_cols = v.n_rows = 0; v.n_ini = 1; v.h_x = 1; v.cand_cap = 0; v.kcap = 1; v.node_cap = 8;
        }
        v.kp_off = kp; kp += v.kcap;
        v.cand_off = cand; cand += (size_t)align_up(std::max(v.cand_cap, 1), 2);
        v.node_off = node; node += (size_t)align_up(v.node_cap, 4);
        const size_t bytes = (size_t)v.pitch * v.h;
        if (l > 0) { v.raw_off = img; img += align_up(bytes, 256); }
        v.blur_off = img; img += align_up(bytes, 256);
    }
    LV.kcap_total = kp;
    g->img_fs = img;
    g->cand_fs = cand;
    g->node_fs = node;
    g->l0_pitch = LV.lv[0].pitch;
    g->l0_fs = align_up((size_t)g->l0_pitch * h0, 256);
    size_t p = 1;
    while (p < (size_t)max_quota) p <<= 1;
    g->sort_lds = p * sizeof(unsigned long long);
    if (g->sort_lds > 60 * 1024) return fail(ORBX_E_UNSUPPORTED, "per-level quota above 7680 features");
    return ORBX_OK;
}

template <typename T> static hipError_t dev_alloc(T **p, size_t n)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    return hipMalloc((void **)p, std::max(n, (size_t)1) * sizeof(T));
}

static int ensure_geometry(orbx_ctx *c, int w0, int h0, int batch, int out_cap)
{
    Geometry g;
    const bool same = (w0 == c->cur_w && h0 == c->cur_h);
    if (!same) {
        int rc = compute_geometry(c, w0, h0, &g);
        if (rc) return rc;
    } else {
        g.levels = c->levels;
        g.img_fs = c->alloc_img_fs; g.cand_fs = c->alloc_cand_fs; g.node_fs = c->alloc_node_fs; g.l0_fs = c->alloc_l0_fs;
        g.l0_pitch = (int)c->l0_stage_pitch; g.sort_lds = c->sort_lds_bytes;
    }
    HIP_TRY(hipSetDevice(c->device));
    const bool grow = batch > c->alloc_batch || g.img_fs > c->alloc_img_fs || g.cand_fs > c->alloc_cand_fs ||
                      g.node_fs > c->alloc_node_fs || g.l0_fs > c->alloc_l0_fs ||
                      g.levels.kcap_total > c->alloc_kcap_total;
    if (grow) {
        // arenas are about to be replaced: nothing enqueued earlier (on the handle's streams or the caller's) may
        // still be using them
        HIP_TRY(hipDeviceSynchronize());
        const int B = std::max(batch, c->alloc_batch);
        const size_t img_fs = std::max(g.img_fs, c->alloc_img_fs), cand_fs = std::max(g.cand_fs, c->alloc_cand_fs);
        const size_t node_fs = std::max(g.node_fs, c->alloc_node_fs), l0_fs = std::max(g.l0_fs, c->alloc_l0_fs);
        const int kcap = std::max(g.levels.kcap_total, c->alloc_kcap_total);
        OrbxBuffers &b = c->buf;
        HIP_TRY(dev_alloc(&b.img_arena, img_fs * B));
        HIP_TRY(dev_alloc(&b.cand, cand_fs * B));
        HIP_TRY(dev_alloc(&b.pnode, cand_fs * B));
        HIP_TRY(dev_alloc(&b.pcode, cand_fs * B));
        HIP_TRY(dev_alloc(&b.cand_count, (size_t)ORBX_MAX_LEVELS * B));
        HIP_TRY(dev_alloc(&b.bnd0, node_fs * B));
        HIP_TRY(dev_alloc(&b.bnd1, node_fs * B));
        HIP_TRY(dev_alloc(&b.cnt0, node_fs * B));
        HIP_TRY(dev_alloc(&b.cnt1, node_fs * B));
        HIP_TRY(dev_alloc(&b.rank, node_fs * B));
        HIP_TRY(dev_alloc(&b.node_of_rank, node_fs * B));
        HIP_TRY(dev_alloc(&b.newpos, node_fs * B));
        HIP_TRY(dev_alloc(&b.childcnt, 4 * node_fs * B));
        HIP_TRY(dev_alloc(&b.childpos, 4 * node_fs * B));
        HIP_TRY(dev_alloc(&b.best, node_fs * B));
        HIP_TRY(dev_alloc(&b.sel, (size_t)kcap * B));
        HIP_TRY(dev_alloc(&b.kp_ang, (size_t)kcap * B));
        HIP_TRY(dev_alloc(&b.sel_count, (size_t)ORBX_MAX_LEVELS * B));
        HIP_TRY(dev_alloc(&b.sel_prefix, (size_t)ORBX_MAX_LEVELS * B));
        HIP_TRY(dev_alloc(&c->d_l0_stage, l0_fs * B));
        c->alloc_batch = B; c->alloc_img_fs = img_fs; c->alloc_cand_fs = cand_fs; c->alloc_node_fs = node_fs;
        c->alloc_l0_fs = l0_fs; c->alloc_kcap_total = kcap;
        c->alloc_out_cap = 0; // host-API output staging is per (batch, cap)
    }
    if (out_cap > 0 && (out_cap > c->alloc_out_cap || grow)) {
        HIP_TRY(hipDeviceSynchronize());
        const int cap = std::max(out_cap, c->alloc_out_cap);
        const size_t B = (size_t)c->alloc_batch;
        c->out_kp_off = align_up(B * sizeof(int32_t), 256);
        c->out_desc_off = align_up(c->out_kp_off + B * cap * sizeof(orbx_kp), 256);
        c->out_block_bytes = c->out_desc_off + B * cap * 32;
        HIP_TRY(dev_alloc(&c->d_out_block, c->out_block_bytes));
        c->d_out_n = reinterpret_cast<int32_t *>(c->d_out_block);
        c->d_out_kp = reinterpret_cast<orbx_kp *>(c->d_out_block + c->out_kp_off);
        c->d_out_desc = c->d_out_block + c->out_desc_off;
        if (c->h_out_block) { (void)hipHostFree(c->h_out_block); c->h_out_block = nullptr; }
        c->h_out_dev = nullptr;
        if (c->out_block_bytes <= (size_t)4 << 20) {
            HIP_TRY(hipHostMalloc((void **)&c->h_out_block, c->out_block_bytes, hipHostMallocMapped));
            // the descriptor kernel writes a small call's records straight into this block (no copy back);
            // ORBX_VAR_ZERO_COPY = 0 keeps them in HBM and copies
            HIP_TRY(hipHostGetDevicePointer((void **)&c->h_out_dev, c->h_out_block, 0));
        }
        c->alloc_out_cap = cap;
    }
    if (!same) {
        HIP_TRY(hipDeviceSynchronize()); // the level / tap / cell tables below are read by in-flight kernels
        c->levels = g.levels;
        c->cur_w = w0; c->cur_h = h0;
        c->l0_stage_pitch = g.l0_pitch;
        c->sort_lds_bytes = g.sort_lds;
        HIP_TRY(hipMemcpy(c->d_levels, &c->levels, sizeof(OrbxLevels), hipMemcpyHostToDevice));
        {
            std::vector<int> sl((size_t)std::max(c->levels.kcap_total, 1), 0);
            for (int l = 0; l < c->levels.n_levels; ++l)
                for (int k = c->levels.lv[l].kp_off; k < c->levels.lv[l].kp_off + c->levels.lv[l].kcap && k < c->levels.kcap_total; ++k)
                    sl[k] = l;
            HIP_TRY(dev_alloc(&c->d_slot_level, sl.size()));
            HIP_TRY(hipMemcpy(c->d_slot_level, sl.data(), sl.size() * sizeof(int), hipMemcpyHostToDevice));
            c->buf.slot_level = c->d_slot_level;
        }
        {
            const int nc = orbx_build_fast_cells(c->levels, nullptr);
            std::vector<uint16_t> cells((size_t)std::max(nc, 1) * 4);
            orbx_build_fast_cells(c->levels, cells.data());
            HIP_TRY(dev_alloc(&c->d_fast_cells, cells.size()));
            HIP_TRY(hipMemcpy(c->d_fast_cells, cells.data(), cells.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            c->n_fast_cells = nc;
        }
        {
            const int ns = orbx_build_fast_strips(c->levels, 0, c->levels.n_levels, nullptr);
            std::vector<uint16_t> st((size_t)std::max(ns, 1) * 4);
            orbx_build_fast_strips(c->levels, 0, c->levels.n_levels, st.data());
            HIP_TRY(dev_alloc(&c->d_fast_strips, st.size()));
            HIP_TRY(hipMemcpy(c->d_fast_strips, st.data(), st.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            c->n_fast_strips = ns;
            c->n_fast_strips0 = orbx_build_fast_strips(c->levels, 0, 1, nullptr);
        }
        {
            const int nt = orbx_build_blur_tiles(c->levels, nullptr);
            std::vector<uint16_t> tl((size_t)std::max(nt, 1) * 4);
            orbx_build_blur_tiles(c->levels, tl.data());
            HIP_TRY(dev_alloc(&c->d_blur_tiles, tl.size()));
            HIP_TRY(hipMemcpy(c->d_blur_tiles, tl.data(), tl.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            c->n_blur_tiles = nt;
            std::vector<uint16_t> bs;
            std::vector<uint8_t> bh, bv;
            orbx_build_blur_mfma(c->levels, c->taps, bs, bh, bv, c->blur_tab, c->blur_strips_before);
            c->blur_mfma_levels = orbx_blur_mfma_levels(c->levels);
            HIP_TRY(dev_alloc(&c->d_blur_strips, std::max(bs.size(), (size_t)2)));
            HIP_TRY(dev_alloc(&c->d_band_h, std::max(bh.size(), (size_t)16)));
            HIP_TRY(dev_alloc(&c->d_band_v, std::max(bv.size(), (size_t)16)));
            if (!bs.empty()) HIP_TRY(hipMemcpy(c->d_blur_strips, bs.data(), bs.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            if (!bh.empty()) HIP_TRY(hipMemcpy(c->d_band_h, bh.data(), bh.size(), hipMemcpyHostToDevice));
            if (!bv.empty()) HIP_TRY(hipMemcpy(c->d_band_v, bv.data(), bv.size(), hipMemcpyHostToDevice));
            std::vector<uint16_t> bb;
            std::vector<uint8_t> bdh;
            orbx_build_blur_desc(c->levels, c->taps, c->blur_tab, bb, bdh, c->bd_tab, &c->bd_levels, &c->bd_bk_stride);
            c->n_bd_blocks = (int)(bb.size() / 2);
            HIP_TRY(dev_alloc(&c->d_bd_blocks, std::max(bb.size(), (size_t)2)));
            HIP_TRY(dev_alloc(&c->d_bd_band_h, std::max(bdh.size(), (size_t)16)));
            if (!bb.empty()) HIP_TRY(hipMemcpy(c->d_bd_blocks, bb.data(), bb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            if (!bdh.empty()) HIP_TRY(hipMemcpy(c->d_bd_band_h, bdh.data(), bdh.size(), hipMemcpyHostToDevice));
        }
        std::vector<OrbxTap> taps, ytaps;
        for (int l = 1; l < c->levels.n_levels; ++l) {
            const OrbxLevel &d = c->levels.lv[l], &s = c->levels.lv[l - 1];
            linear_taps(d.w, s.w, true, taps);
            while (taps.size() % 4) taps.push_back(taps.back()); // k_resize reads the column taps four at a time
            HIP_TRY(dev_alloc(&c->d_xtap[l], taps.size()));
            HIP_TRY(hipMemcpy(c->d_xtap[l], taps.data(), taps.size() * sizeof(OrbxTap), hipMemcpyHostToDevice));
            linear_taps(d.h, s.h, false, ytaps);
            HIP_TRY(dev_alloc(&c->d_ytap[l], ytaps.size()));
            HIP_TRY(hipMemcpy(c->d_ytap[l], ytaps.data(), ytaps.size() * sizeof(OrbxTap), hipMemcpyHostToDevice));
            // levels l - 1 and l from one launch (source: level l - 2): the patch of level l - 1 a tile of level l needs
            c->resize2_ok[l - 1] = l >= 2 && orbx_resize2_fits(taps.data(), ytaps.data(), s.w, s.h, d.w, d.h);
            c->resize_lds_ok[l] = orbx_resize_lds_fits(taps.data(), ytaps.data(), s.w, s.h, d.w, d.h);
        }
    }
    // k_blur_desc's per-frame bucket starts and 32-byte item records: grown with the batch, the slot count or the bucket count
    if (c->alloc_batch > c->alloc_bd_batch || c->levels.kcap_total > c->alloc_bd_kcap || c->bd_bk_stride > c->alloc_bd_stride) {
        HIP_TRY(hipDeviceSynchronize());
        c->alloc_bd_batch = std::max(c->alloc_batch, c->alloc_bd_batch);
        c->alloc_bd_kcap = std::max(c->levels.kcap_total, c->alloc_bd_kcap);
        c->alloc_bd_stride = std::max(c->bd_bk_stride, c->alloc_bd_stride);
        HIP_TRY(dev_alloc(&c->d_bd_bk_start, (size_t)c->alloc_bd_stride * c->alloc_bd_batch));
        HIP_TRY(dev_alloc(&c->d_bd_items, (size_t)32 * std::max(c->alloc_bd_kcap, 1) * c->alloc_bd_batch));
    }
    // frame strides of the arenas are the allocated ones
    c->buf.img_frame_stride = c->alloc_img_fs;
    c->buf.cand_frame_stride = c->alloc_cand_fs;
    c->buf.node_frame_stride = c->alloc_node_fs;
    c->l0_stage_fs = c->alloc_l0_fs;
    return ORBX_OK;
}

// ------------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------------
static int create_common(const orbx_cfg *cfg, const int *quotas_override, orbx_t **out)
{
    if (!cfg || !out) return fail(ORBX_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_levels < 1 || cfg->n_levels > ORBX_MAX_LEVELS) return fail(ORBX_E_ARG, "n_levels out of range");
    if (cfg->n_features < 1) return fail(ORBX_E_ARG, "n_features must be >= 1");
    if (!(cfg->scale_factor > 1.0f)) return fail(ORBX_E_ARG, "scale_factor must be > 1");
    if (fabs((double)cfg->scale_factor - 2.0) < 1e-6)
        return fail(ORBX_E_UNSUPPORTED, "scale_factor 2.0 takes cv::resize's INTER_AREA shortcut, not implemented");
    if (cfg->ini_th_fast < 1 || cfg->min_th_fast < 1 || cfg->ini_th_fast > 254 || cfg->min_th_fast > 254)
        return fail(ORBX_E_ARG, "FAST thresholds must be in [1,254]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    orbx_ctx *c = new orbx_ctx(); // value-initialised: every member is zero
    c->cfg = *cfg;
    if (c->cfg.max_batch < 1) c->cfg.max_batch = 1;
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) { delete c; return fail(ORBX_E_ARG, "device ordinal out of range"); }
    c->device = dev;
    compute_tables(c);
    if (quotas_override) memcpy(c->quotas, quotas_override, sizeof(int) * cfg->n_levels);
    c->cur_w = c->cur_h = -1;
    auto cleanup = [&](int code) { orbx_destroy(c); return code; };
    if (hipSetDevice(dev) != hipSuccess) return cleanup(fail(ORBX_E_NO_DEVICE, "hipSetDevice failed"));
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c->n_cus = cus; }
    // The handle's own stream serves the HOST-POINTER entry points only (a NULL stream argument of a device entry point is
    // stream 0 itself) and is NON-BLOCKING: such a call uploads, computes, downloads and waits on it, so nothing of it has to be
    // ordered with the legacy stream -- and a blocking stream would make every legacy-stream operation of any thread of the
    // process (another thread's hipMemcpy, its NULL-stream device calls) a barrier against this handle's work: the reference runs
    // Tracking and LocalMapping side by side (include/orbx.h, "Streams").  The internal side / sub-streams below are forked
    // from and joined into the stream of the call with events.
    if (hipStreamCreateWithFlags(&c->stream, ORBX_HANDLE_STREAM_FLAGS) != hipSuccess)
        return cleanup(fail(ORBX_E_NO_DEVICE, "hipStreamCreate failed"));
    for (int i = 0; i <= ORBX_N_STAGES; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return cleanup(fail(ORBX_E_NO_DEVICE, "hipEventCreate failed"));
    for (int i = 0; i < 8; ++i)
        if (hipStreamCreateWithFlags(&c->sub[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess)
            return cleanup(fail(ORBX_E_NO_DEVICE, "sub-stream creation failed"));
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess)
        return cleanup(fail(ORBX_E_NO_DEVICE, "hipEventCreate failed"));
    for (int i = 0; i < 9; ++i)
        if (hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_pyr[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_blur[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_start[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fast0[i], hipEventDisableTiming) != hipSuccess)
            return cleanup(fail(ORBX_E_NO_DEVICE, "side-stream creation failed"));
    // kernel-choice switches: defaults here, orbx_set_variant() changes them per handle (include/orbx.h)
    c->side_blur = 1;      // the blur on a side stream next to FAST
    c->early_fast = -1;    // level 0's FAST beside the pyramid unless the pyramid is k_resize_lds's (see enqueue)
    c->split_level0 = 1;
    c->fast_variant = 0;   // strips for batches, one wave per cell for a few frames
    c->blur_mfma = 1;      // matrix pipe for batches and levels that are large enough
    c->resize2 = 1;        // two levels per launch for calls with a few frames
    c->resize_lds = 1;     // source tile through LDS for resident batches
    // Sub-batch streams, default 1: with the blur already on a side stream, splitting the batch into frame ranges on more
    // streams loses (tools/overlap_sweep.sh: 103.6 k frames/s with 1, 95.7 k with 2, 94.8 k with 4 at the runtime's default of
    // four hardware queues; more streams than queues alias and serialise).  The knob stays for experiments.
    c->n_sub = 1;
    c->zero_copy = 1;
    c->desc_variant = 0;
    c->fast_cell_group = 8;
    if (hipMalloc((void **)&c->d_levels, sizeof(OrbxLevels)) != hipSuccess ||
        hipMalloc((void **)&c->d_umax, sizeof(int) * 16) != hipSuccess ||
        hipMalloc((void **)&c->d_taps, sizeof(int) * 8) != hipSuccess)
        return cleanup(fail(ORBX_E_NO_DEVICE, "hipMalloc failed"));
    if (hipMemcpy(c->d_umax, c->u_max, sizeof(int) * 16, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_taps, c->taps, sizeof(int) * 7, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(ORBX_E_NO_DEVICE, "hipMemcpy failed"));
    if (cfg->max_width > 0 && cfg->max_height > 0) {
        int rc = ensure_geometry(c, cfg->max_width, cfg->max_height, c->cfg.max_batch, 0);
        if (rc) return cleanup(rc);
    }
    *out = c;
    return ORBX_OK;
}

extern "C" int orbx_create(const orbx_cfg *cfg, orbx_t **out) { return create_common(cfg, nullptr, out); }


extern "C" int orbx_create_requota(const orbx_t *other, int n_features, orbx_t **out)
{
    // reference ORBExtractor.cpp:477-493: same pyramid and thresholds, new quotas
    if (!other) return fail(ORBX_E_ARG, "null argument");
    orbx_cfg cfg = other->cfg;
    cfg.n_features = n_features;
    int rc = create_common(&cfg, nullptr, out);
    if (rc) return rc;
    for (int which = 0; which < ORBX_N_VARIANTS; ++which) { // the twin keeps the kernel choices of the handle it was made from
        int v = 0;
        if (orbx_get_variant(other, which, &v) == ORBX_OK) (void)orbx_set_variant(*out, which, v);
    }
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)sync_handle(c);
    OrbxBuffers &b = c->buf;
    void *ptrs[] = {b.img_arena, b.cand, b.pnode, b.pcode, b.cand_count, b.bnd0, b.bnd1, b.cnt0, b.cnt1, b.rank, b.node_of_rank,
                    b.newpos, b.childcnt, b.childpos, b.best, b.sel, b.kp_ang, b.sel_count, b.sel_prefix, c->d_slot_level, c->d_levels, c->d_umax, c->d_taps,
                    c->d_l0_stage, c->d_out_block, c->d_fast_cells, c->d_fast_strips, c->d_blur_tiles, c->d_blur_strips, c->d_band_h, c->d_band_v, c->d_bd_blocks, c->d_bd_band_h, c->d_bd_bk_start, c->d_bd_items};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->h_out_block) (void)hipHostFree(c->h_out_block);
    for (int l = 0; l < ORBX_MAX_LEVELS; ++l) {
        if (c->d_xtap[l]) (void)hipFree(c->d_xtap[l]);
        if (c->d_ytap[l]) (void)hipFree(c->d_ytap[l]);
    }
    for (int i = 0; i <= ORBX_N_STAGES; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int st = 0; st < ORBX_N_STAGES; ++st)
        for (int i = 0; i < 4; ++i) if (c->ev_in[st][i]) (void)hipEventDestroy(c->ev_in[st][i]);
    if (c->ev_after_fast) (void)hipEventDestroy(c->ev_after_fast);
    for (int i = 0; i < 8; ++i) {
        if (c->sub[i]) { (void)hipStreamSynchronize(c->sub[i]); (void)hipStreamDestroy(c->sub[i]); }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < 9; ++i) {
        if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
        if (c->ev_pyr[i]) (void)hipEventDestroy(c->ev_pyr[i]);
        if (c->ev_blur[i]) (void)hipEventDestroy(c->ev_blur[i]);
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_fast0[i]) (void)hipEventDestroy(c->ev_fast0[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbx_tables(const orbx_t *c, int *n_levels, float *sf, float *isf, float *ss, float *iss, float *lsf,
                           int32_t *quotas, int32_t *u_max16)
{
    if (!c) return fail(ORBX_E_ARG, "null handle");
    const int L = c->cfg.n_levels;
    if (n_levels) *n_levels = L;
    if (sf) memcpy(sf, c->scale_factors, sizeof(float) * L);
    if (isf) memcpy(isf, c->inv_scale_factors, sizeof(float) * L);
    if (ss) memcpy(ss, c->square_sigmas, sizeof(float) * L);
    if (iss) memcpy(iss, c->inv_square_sigmas, sizeof(float) * L);
    if (lsf) *lsf = c->log_scale_factor;
    if (quotas) memcpy(quotas, c->quotas, sizeof(int) * L);
    if (u_max16) memcpy(u_max16, c->u_max, sizeof(int) * 16);
    return ORBX_OK;
}

extern "C" int orbx_level_size(const orbx_t *c, int w0, int h0, int level, int *w, int *h)
{
    if (!c || level < 0 || level >= c->cfg.n_levels || !w || !h) return fail(ORBX_E_ARG, "bad argument");
    level_size(c, w0, h0, level, w, h);
    return ORBX_OK;
}

extern "C" int orbx_max_keypoints(const orbx_t *c, int w0, int h0)
{
    if (!c) return fail(ORBX_E_ARG, "null handle");
    if (w0 == c->cur_w && h0 == c->cur_h) return c->levels.kcap_total;
    Geometry g;
    int rc = compute_geometry(c, w0, h0, &g);
    if (rc) return rc;
    return g.levels.kcap_total;
}

// ------------------------------------------------------------------------------------------------
// the pipeline
// ------------------------------------------------------------------------------------------------
struct PhaseTrace { // ORBX_TRACE=1: host time stamps of the phases of orbx_extract_batch on stderr
    bool on; std::chrono::steady_clock::time_point t0;
    PhaseTrace() : on(getenv("ORBX_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[orbx] %-24s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - t0).count());
        t0 = t;
    }
};

// the same arenas, seen from frame f0 on
static OrbxBuffers offset_buffers(const OrbxBuffers &a, int f0, int kcap_total)
{
    OrbxBuffers b = a;
    const size_t f = (size_t)f0;
    b.img_arena += f * a.img_frame_stride;
    b.cand += f * a.cand_frame_stride; b.pnode += f * a.cand_frame_stride; b.pcode += f * a.cand_frame_stride;
    b.cand_count += f * ORBX_MAX_LEVELS; b.sel_count += f * ORBX_MAX_LEVELS; b.sel_prefix += f * ORBX_MAX_LEVELS;
    const size_t n = f * a.node_frame_stride;
    b.bnd0 += n; b.bnd1 += n; b.cnt0 += n; b.cnt1 += n; b.rank += n; b.node_of_rank += n; b.newpos += n; b.best += n;
    b.childcnt += 4 * n; b.childpos += 4 * n;
    b.sel += f * kcap_total; b.kp_ang += f * kcap_total;
    return b;
}

// timing mode 2: a pair of events on stream `st` around what is launched during the object's life (at most two groups per stage)
struct InStep {
    orbx_ctx *c; int stage; hipStream_t st; int k;
    InStep(orbx_ctx *c_, int stage_, hipStream_t st_) : c(c_), stage(stage_), st(st_), k(-1)
    {
        if (c->timing != 2 || !c->ev_in[0][0] || c->ev_in_n[stage] >= 2) return;
        k = c->ev_in_n[stage]++;
        (void)hipEventRecord(c->ev_in[stage][2 * k], st);
    }
    ~InStep() { if (k >= 0) (void)hipEventRecord(c->ev_in[stage][2 * k + 1], st); }
};

static int enqueue(orbx_ctx *c, hipStream_t s, const uint8_t *d_l0, size_t l0_fs, int l0_pitch, int f0, int n_frames,
                   orbx_kp *d_kp, uint8_t *d_desc, int cap, int32_t *d_n, bool t, int slot, bool latency)
{
    const OrbxLevels &LV = c->levels;
    const int L = LV.n_levels;
    const OrbxBuffers b = offset_buffers(c->buf, f0, LV.kcap_total);
    d_l0 += (size_t)f0 * l0_fs;
    d_kp += (size_t)f0 * cap; d_desc += (size_t)f0 * cap * 32; d_n += f0;
    if (t) HIP_TRY(hipEventRecord(c->ev[0], s));
    auto raw = [&](int l, const uint8_t **p, size_t *fs, int *pitch) {
        if (l == 0) { *p = d_l0; *fs = l0_fs; *pitch = l0_pitch; }
        else { *p = b.img_arena + LV.lv[l].raw_off; *fs = b.img_frame_stride; *pitch = LV.lv[l].pitch; }
    };
    // k_resize_lds pays once the batch is large (1242x375: the two kernels tie up to 192 frames -- 0.18 ms -- and part at 256: 0.206
    // against 0.231, at 512: 0.35 against 0.46; 1920x1080 x 128: 0.37 against 0.52): from about 10^8 level-0 pixels per call
    const bool lds_resize = c->resize_lds == 2 || (c->resize_lds == 1 && (size_t)n_frames * (size_t)LV.lv[0].w * (size_t)LV.lv[0].h >= (size_t)100000000);
    // level l from level l - 1 -- and level l + 1 with it when the two fit one launch (k_resize2): returns the last level made
    auto launch_resize = [&](hipStream_t st, int l, int *zero_counts) -> int {
        const uint8_t *sp; size_t sfs; int spitch;
        raw(l - 1, &sp, &sfs, &spitch);
        // (small calls only: per 512 frames the pyramid takes 0.71 ms this way against 0.47 -- a workgroup's two phases run one
        // after the other -- while a single 1242x375 frame is back 20 us sooner, 157 -> 138 us; ORBX_VAR_RESIZE2 = 2 forces it)
        if ((c->resize2 == 2 || (c->resize2 == 1 && n_frames < 24)) && l + 1 < L && c->resize2_ok[l]) {
            orbx_launch_resize2(st, sp, sfs, spitch, LV.lv[l - 1].w, LV.lv[l - 1].h, b.img_arena + LV.lv[l].raw_off, b.img_frame_stride,
                                LV.lv[l].pitch, LV.lv[l].w, LV.lv[l].h, c->d_xtap[l], c->d_ytap[l], b.img_arena + LV.lv[l + 1].raw_off,
                                b.img_frame_stride, LV.lv[l + 1].pitch, LV.lv[l + 1].w, LV.lv[l + 1].h, c->d_xtap[l + 1],
                                c->d_ytap[l + 1], n_frames, zero_counts);
            return l + 1;
        }
        if (lds_resize && c->resize_lds_ok[l]) {
            orbx_launch_resize_lds(st, sp, sfs, spitch, LV.lv[l - 1].w, LV.lv[l - 1].h, l == 1 ? LV.lv[0].w : spitch,
                                   b.img_arena + LV.lv[l].raw_off, b.img_frame_stride, LV.lv[l].pitch, LV.lv[l].w, LV.lv[l].h, c->d_xtap[l],
                                   c->d_ytap[l], n_frames, zero_counts);
            return l;
        }
        orbx_launch_resize(st, sp, sfs, spitch, LV.lv[l - 1].w, LV.lv[l - 1].h, b.img_arena + LV.lv[l].raw_off, b.img_frame_stride,
                           LV.lv[l].pitch, LV.lv[l].w, LV.lv[l].h, c->d_xtap[l], c->d_ytap[l], n_frames, zero_counts);
        return l;
    };
    const bool side_ok = !t && c->side_blur && slot >= 0;
    // FAST work units (strips of cells, or single cells); both lists are level-major
    // Strips need a few thousand waves in flight to pay (two cells per wave, a long rolling pipeline); a call with a
    // handful of frames is bounded by the longest wave instead, where one short wave per cell finishes sooner
    // (single 1242x375 frame: 31 us against 50; 16 frames: 63 against 69; 32 frames: 110 against 97).  Same candidates either way (tests).  ORBX_VAR_FAST = 1 / 2 force one.
    const bool strips = c->fast_variant == 2 || (c->fast_variant == 0 && n_frames >= 24);
    const uint16_t *d_units = strips ? c->d_fast_strips : c->d_fast_cells;
    const int n_units = strips ? c->n_fast_strips : c->n_fast_cells;
    const int n_cells0 = strips ? c->n_fast_strips0 : LV.lv[0].n_cols * LV.lv[0].n_rows;
    auto launch_fast = [&](hipStream_t st, const uint16_t *units, int n) {
        // timing mode 2: the step keeps its streams, and HIP events on the launch's own stream bracket each launch group, so
        // that the kernels are timed as they run beside the others (orbx_stage_times_in_step_ms)
        InStep ft(c, ORBX_STAGE_FAST, st);
        if (strips) orbx_launch_fast_strips(st, d_l0, l0_fs, l0_pitch, c->d_levels, LV, b, units, n, n_frames);
        else orbx_launch_fast(st, d_l0, l0_fs, l0_pitch, c->d_levels, LV, b, units, n, n_frames, c->fast_cell_group);
    };
    // 7x7 Gaussian of levels [lb, le): on the matrix pipe for the levels that are large enough when the call is a batch
    // (ORBX_VAR_BLUR = 2 forces it for any batch, 0 switches it off), the VALU kernels for the rest
    // Levels [0, fl) get their descriptors from k_blur_desc (blur and sampling in one pass, no blurred copy); the blur pass and
    // k_orient_desc serve the rest -- levels too narrow for it, or all of them on the two-kernel path (ORBX_VAR_DESC = 1).
    const bool big_call = (size_t)n_frames * (size_t)LV.lv[0].w * (size_t)LV.lv[0].h >= (size_t)ORBX_FUSED_DESC_MIN_PIXELS;
    const int fl = (c->desc_variant == 2 || (c->desc_variant == 0 && big_call)) ? c->bd_levels : 0;
    c->last_fused_levels = fl;
    auto launch_blur = [&](hipStream_t st, int lb, int le) {
        lb = std::max(lb, fl);
        if (le <= lb) return;
        InStep bt(c, ORBX_STAGE_BLUR, st);
        const int lm = (c->blur_mfma == 2 || (c->blur_mfma == 1 && n_frames >= 8)) ? std::min(c->blur_mfma_levels, le) : 0;
        if (lm > lb)
            orbx_launch_blur_mfma(st, d_l0, l0_fs, l0_pitch, LV, b, c->blur_tab, c->d_blur_strips, c->blur_strips_before,
                                  c->d_band_h, c->d_band_v, c->taps, n_frames, lb, lm);
        if (le > std::max(lb, lm))
            orbx_launch_blur(st, d_l0, l0_fs, l0_pitch, c->d_levels, LV, b, c->d_blur_tiles, c->n_blur_tiles, c->d_taps, n_frames,
                             std::max(lb, lm), le);
    };
    // orientation, then the descriptors: k_blur_desc for levels [0, fl), k_orient_desc (behind the blur pass) for the others
    auto launch_desc = [&](hipEvent_t blur_done) {
        // (stage timing: mode 1 has the boundary recorded by the launcher; mode 2 brackets orientation and descriptors apart,
        // the launcher's own event standing between them)
        const bool in2 = c->timing == 2 && c->ev_in[0][0];
        if (in2) { (void)hipEventRecord(c->ev_in[ORBX_STAGE_ORIENT][0], s); c->ev_in_n[ORBX_STAGE_ORIENT] = 1; }
        int *bk = c->d_bd_bk_start + (size_t)f0 * c->bd_bk_stride;
        uint8_t *items = c->d_bd_items + (size_t)f0 * 32 * LV.kcap_total;
        if (fl > 0) orbx_launch_desc_bins(s, c->d_levels, b, c->bd_tab, fl, bk, c->bd_bk_stride, items, cap, d_n, n_frames);
        orbx_launch_orient_desc(s, d_l0, l0_fs, l0_pitch, c->d_levels, LV, b, c->d_umax, d_kp, d_desc, cap, d_n, n_frames,
                                fl < L ? blur_done : nullptr, fl, in2 ? c->ev_in[ORBX_STAGE_ORIENT][1] : (t ? c->ev[ORBX_STAGE_ORIENT + 1] : nullptr),
                                fl > 0 ? items : nullptr, in2 ? c->ev_in[ORBX_STAGE_DESC][0] : nullptr);
        if (fl > 0)
            orbx_launch_desc_fused(s, d_l0, l0_fs, l0_pitch, LV, b, c->bd_tab, fl, c->d_bd_blocks, c->n_bd_blocks, c->d_bd_band_h, c->d_band_v,
                                   bk, c->bd_bk_stride, items, c->taps, d_kp, d_desc, cap, n_frames);
        // (the descriptor bracket was opened by the launcher, behind its wait for the side stream's blur)
        if (in2) { (void)hipEventRecord(c->ev_in[ORBX_STAGE_DESC][1], s); c->ev_in_n[ORBX_STAGE_DESC] = 1; }
    };
    // Level 0 needs no pyramid, so its FAST can start on the side stream beside the resizes.  That pays next to k_resize (no LDS,
    // 27 VGPRs: its waves fit between FAST's), not next to k_resize_lds, which wants the LDS that FAST's workgroups fill
    // (512 frames, k_resize_lds with / without the early launch: 194.8 / 198.1 k frames/s; k_resize: 194.9 / 191.0).
    const bool lds_pyramid = lds_resize && L > 1 && c->resize_lds_ok[1];
    const int early_fast = c->early_fast >= 0 ? c->early_fast : (lds_pyramid ? 0 : 1);
    const bool early = side_ok && early_fast && L > 1 && n_cells0 > 0 && n_cells0 < n_units;
    // A call with a few frames is a chain of latency-bound kernels: level 0 (a third of the pixels, the longest quadtree)
    // then runs FAST -> quadtree on the side stream next to resize -> FAST -> quadtree of the other levels, and the blur
    // takes a stream of its own (slot 7 is free whenever the whole batch is on slot 8).
    // (only for the synchronous host entry points: on the asynchronous device entry point, where calls pipeline, the extra
    // cross-stream waits cost throughput -- 4.5 k against 7.0 k single-frame steps per second)
    const bool split = latency && early && slot == 8 && !strips && c->split_level0;
    const int bslot = split ? 7 : slot;
    // the candidate counters are cleared by the first pyramid launch (one stream operation less ahead of every batch) unless
    // FAST starts before or without it: level 0's early launch, or a one-level pyramid
    const bool zero_in_resize = !split && !early && L > 1;
    if (!split && !zero_in_resize) HIP_TRY(hipMemsetAsync(b.cand_count, 0, sizeof(int) * ORBX_MAX_LEVELS * n_frames, s));
    if (split) {
        // The host needs 2-3 us per launch, about what a small resize takes on the device: the chain the call waits for
        // (resize x7 -> FAST -> quadtree of levels 1..) is issued first and back to back, the two side chains (level 0;
        // the blur) afterwards -- they have started long before the main chain needs them.
        // Levels 0 .. G-1 (most of the pixels) do not wait for the end of the resize chain: their FAST and quadtree run on the
        // side stream as soon as level G-1 exists, beside the resizes -- and then the FAST and quadtree -- of the small
        // levels.  (A stream of its own for levels 1 .. G-1 was measured: with a fourth stream the call takes 200 us, not
        // 160 -- the runtime maps streams onto four hardware queues.)
        const int G = std::min(std::max(c->split_level0, 1), L - 1); // first level of the main chain's FAST / quadtree (ORBX_VAR_SPLIT_LEVEL0, default 1: level 0 alone; 4 is 3 us better at 1242x375 and 17 us worse at 1920x1080)
        int cells_before[ORBX_MAX_LEVELS + 1];
        cells_before[0] = 0;
        for (int l = 0; l < L; ++l) cells_before[l + 1] = cells_before[l] + LV.lv[l].n_cols * LV.lv[l].n_rows;
        // Which chain the call ends up waiting for depends on the frame: at 1242x375 it is the main one, at 1920x1080 level 0's
        // (its FAST and its quadtree -- one workgroup, 15 000 candidates -- take 47 + 88 us against 64 + 62 for levels 1..7 together).
        // From a megapixel on the side chain is therefore ISSUED as soon as its inputs exist, ahead of the remaining resizes.
        const bool side_first = (size_t)LV.lv[0].w * (size_t)LV.lv[0].h >= (size_t)ORBX_SIDE_FIRST_PIXELS;
        auto launch_side = [&]() -> int {
            HIP_TRY(hipStreamWaitEvent(c->side[slot], c->ev_start[slot], 0));
            launch_fast(c->side[slot], d_units, cells_before[G]);
            orbx_launch_octree(c->side[slot], c->d_levels, LV, b, n_frames, c->sort_lds_bytes, 0, G, c->n_cus);
            HIP_TRY(hipEventRecord(c->ev_fast0[slot], c->side[slot]));
            return ORBX_OK;
        };
        bool started = false;
        for (int l = 1; l < L;) {
            // the first launch also clears the candidate counters
            const int done = launch_resize(s, l, l == 1 ? b.cand_count : nullptr); // last level this launch produced
            if (!started && done >= std::max(G - 1, 1)) { // counters are zero, levels < G exist
                HIP_TRY(hipEventRecord(c->ev_start[slot], s));
                started = true;
                if (side_first) { int rc = launch_side(); if (rc) return rc; }
            }
            l = done + 1;
        }
        HIP_TRY(hipEventRecord(c->ev_pyr[bslot], s));
        if (cells_before[L] > cells_before[G]) launch_fast(s, d_units + 4 * cells_before[G], cells_before[L] - cells_before[G]);
        orbx_launch_octree(s, c->d_levels, LV, b, n_frames, c->sort_lds_bytes, G, L, c->n_cus);
        if (!side_first) { int rc = launch_side(); if (rc) return rc; }
        HIP_TRY(hipStreamWaitEvent(c->side[bslot], c->ev_pyr[bslot], 0));
        launch_blur(c->side[bslot], 0, L);
        HIP_TRY(hipEventRecord(c->ev_blur[bslot], c->side[bslot]));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_fast0[slot], 0));
        launch_desc(c->ev_blur[bslot]);
        HIP_TRY(hipGetLastError());
        return ORBX_OK;
    }
    if (early) {
        HIP_TRY(hipEventRecord(c->ev_start[slot], s)); // candidate counters are zero from here on
        HIP_TRY(hipStreamWaitEvent(c->side[slot], c->ev_start[slot], 0));
        launch_fast(c->side[slot], d_units, n_cells0);
        HIP_TRY(hipEventRecord(c->ev_fast0[slot], c->side[slot]));
        if (early_fast > 1) // level 0 needs no pyramid for its blur either
            launch_blur(c->side[slot], 0, 1);
    }
    // (Measured and dropped, round 5: the pyramid's last levels -- a chain of small launches, 0.11 ms for a fifth of level 1's
    // pixels -- and their FAST on the side stream beside FAST of the large levels: resize 0.34 -> 0.18-0.30 ms in step, FAST
    // 0.70 -> 0.76-0.77, step 2.21 against 2.18 ms for every split level 3..6; profiles/NEGATIVES.md.)
    {
        InStep rt(c, ORBX_STAGE_RESIZE, s);
        for (int l = 1; l < L;) l = launch_resize(s, l, l == 1 && zero_in_resize ? b.cand_count : nullptr) + 1;
    }
    if (t) HIP_TRY(hipEventRecord(c->ev[1], s));
    // (nothing to fork when k_blur_desc describes every level: no blur pass, and no event pair in the stream between the pyramid and FAST)
    const bool side = !t && c->side_blur && slot >= 0 && fl < L;
    auto fork_blur = [&]() -> int { // the blur only needs the pyramid: side stream, joined before the descriptor kernel
        HIP_TRY(hipEventRecord(c->ev_pyr[bslot], s));
        HIP_TRY(hipStreamWaitEvent(c->side[bslot], c->ev_pyr[bslot], 0));
        launch_blur(c->side[bslot], (early && early_fast > 1) ? 1 : 0, L);
        HIP_TRY(hipEventRecord(c->ev_blur[bslot], c->side[bslot]));
        return ORBX_OK;
    };
    if (side && c->side_blur == 1) { int rc = fork_blur(); if (rc) return rc; } // next to FAST
    if (early) {
        launch_fast(s, d_units + 4 * n_cells0, n_units - n_cells0);
        HIP_TRY(hipStreamWaitEvent(s, c->ev_fast0[slot], 0));
    } else {
        launch_fast(s, d_units, n_units);
    }
    if (t) HIP_TRY(hipEventRecord(c->ev[2], s));
    if (slot == 8) { // (a batch split over sub-streams records the event once, behind the join of all of them: enqueue_batch)
        if (!c->ev_after_fast) HIP_TRY(hipEventCreateWithFlags(&c->ev_after_fast, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_after_fast, s)); // what follows (quadtree, orientation) leaves the vector ALUs mostly idle
        c->after_fast_valid = true;
    }
    if (side && c->side_blur == 2) { int rc = fork_blur(); if (rc) return rc; } // next to the quadtree and orientation
    if (!side)
        launch_blur(s, 0, L);
    if (t) HIP_TRY(hipEventRecord(c->ev[3], s));
    {
        InStep ot(c, ORBX_STAGE_OCTREE, s);
        orbx_launch_octree(s, c->d_levels, LV, b, n_frames, c->sort_lds_bytes, 0, L, c->n_cus);
    }
    if (t) HIP_TRY(hipEventRecord(c->ev[4], s));
    if (side && c->side_blur >= 3) { int rc = fork_blur(); if (rc) return rc; } // next to the orientation only
    launch_desc(side ? c->ev_blur[bslot] : nullptr); // (records ev[5] between orientation and descriptors when timed)
    if (t) { HIP_TRY(hipEventRecord(c->ev[ORBX_N_STAGES], s)); c->ev_valid = true; }
    HIP_TRY(hipGetLastError());
    return ORBX_OK;
}

// whole batch on one stream, or split into frame ranges on the handle's sub-streams (forked from and joined
// back into `s` with events, so the caller still sees one in-order stream)
static int enqueue_batch(orbx_ctx *c, hipStream_t s, const uint8_t *d_l0, size_t l0_fs, int l0_pitch, int n_frames,
                         orbx_kp *d_kp, uint8_t *d_desc, int cap, int32_t *d_n, bool latency)
{
    c->last_l0 = d_l0; c->last_l0_fs = l0_fs; c->last_l0_pitch = l0_pitch; c->last_frames = n_frames;
    for (int i = 0; i < ORBX_N_STAGES; ++i) c->ev_in_n[i] = 0;
    const int ns = c->timing == 1 ? 1 : std::min(c->n_sub, n_frames / 8);
    if (ns <= 1) return enqueue(c, s, d_l0, l0_fs, l0_pitch, 0, n_frames, d_kp, d_desc, cap, d_n, c->timing == 1, 8, latency);
    HIP_TRY(hipEventRecord(c->ev_fork, s));
    for (int i = 0; i < ns; ++i) {
        const int f0 = (int)((long long)n_frames * i / ns), f1 = (int)((long long)n_frames * (i + 1) / ns);
        HIP_TRY(hipStreamWaitEvent(c->sub[i], c->ev_fork, 0));
        int rc = enqueue(c, c->sub[i], d_l0, l0_fs, l0_pitch, f0, f1 - f0, d_kp, d_desc, cap, d_n, false, i, false);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(c->ev_join[i], c->sub[i]));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_join[i], 0));
    }
    // orbx_stream_wait_fast must cover EVERY frame range: with the batch split, the event sits behind the join of all
    // sub-streams (conservative: the whole extraction, not only its FAST stages)
    if (!c->ev_after_fast) HIP_TRY(hipEventCreateWithFlags(&c->ev_after_fast, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_after_fast, s));
    c->after_fast_valid = true;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_t *c, const uint8_t *d_imgs, int n_frames, int width, int height,
                                         int stride, size_t frame_stride, orbx_kp *d_kp, uint8_t *d_desc, int cap,
                                         int32_t *d_n, void *stream)
{
    if (!c || !d_imgs || !d_kp || !d_desc || !d_n) return fail(ORBX_E_ARG, "null argument");
    if (n_frames < 1 || width < 1 || height < 1 || stride < width || cap < 1) return fail(ORBX_E_ARG, "bad size");
    int rc = ensure_geometry(c, width, height, n_frames, 0);
    if (rc) return rc;
    // a NULL stream argument is stream 0 itself (include/orbx.h, "Streams"); the handle remembers it for its host-side waits
    if (!stream) c->null_pending = true;
    return enqueue_batch(c, (hipStream_t)stream, d_imgs, frame_stride, stride, n_frames, d_kp,
                         d_desc, cap, d_n, false);
}

extern "C" int orbx_synchronize(orbx_t *c)
{
    if (!c) return fail(ORBX_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_handle(c));
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_t *c, const uint8_t *imgs, int n_frames, int width, int height, int stride,
                                  size_t frame_stride, orbx_kp *out_kp, uint8_t *out_desc, int cap, int32_t *n_out)
{
    if (!c || !n_out) return fail(ORBX_E_ARG, "null argument");
    for (int f = 0; f < n_frames; ++f) n_out[f] = 0;
    if (!imgs || width <= 0 || height <= 0) return ORBX_OK; // reference :497 -- empty image: silent no-op
    if (n_frames < 1 || stride < width || cap < 1 || !out_kp || !out_desc) return fail(ORBX_E_ARG, "bad argument");
    int rc, dcap; // device-side staging holds every possible keypoint
    if (width == c->cur_w && height == c->cur_h) {
        dcap = c->levels.kcap_total;
    } else {
        Geometry g;
        rc = compute_geometry(c, width, height, &g);
        if (rc) return rc;
        dcap = g.levels.kcap_total;
    }
    rc = ensure_geometry(c, width, height, n_frames, dcap);
    if (rc) return rc;
    // a NULL-stream device call of this handle may still be in flight on the pyramid and scratch this call is about to use
    if (c->null_pending) { HIP_TRY(hipStreamSynchronize((hipStream_t)0)); c->null_pending = false; }
    const int scap = c->alloc_out_cap;
    hipStream_t s = c->stream;
    PhaseTrace tr;
    // Level 0 goes to HBM as it is laid out on the host: dense rows (stride == width, the cv::Mat::clone() case of
    // Frame.cpp:17) are ONE linear copy (a pitched 2-D copy from pageable memory costs milliseconds); the kernels
    // accept any level-0 pitch.  Only a genuinely strided input takes the 2-D copy.
    size_t l0_fs = c->l0_stage_fs;
    int l0_pitch = (int)c->l0_stage_pitch;
    if (stride == width) {
        l0_pitch = width;
        l0_fs = (size_t)width * height;
        if (frame_stride == l0_fs)
            HIP_TRY(hipMemcpyAsync(c->d_l0_stage, imgs, l0_fs * n_frames, hipMemcpyHostToDevice, s));
        else
            for (int f = 0; f < n_frames; ++f)
                HIP_TRY(hipMemcpyAsync(c->d_l0_stage + (size_t)f * l0_fs, imgs + (size_t)f * frame_stride, l0_fs,
                                       hipMemcpyHostToDevice, s));
    } else {
        for (int f = 0; f < n_frames; ++f)
            HIP_TRY(hipMemcpy2DAsync(c->d_l0_stage + (size_t)f * l0_fs, l0_pitch, imgs + (size_t)f * frame_stride, stride,
                                     width, height, hipMemcpyHostToDevice, s));
    }
    tr.mark("copy-in call");
    auto copy_records = [&]() -> int {
        // few frames: the whole fixed-capacity block comes back in one piece (three when the call uses fewer frames than
        // the block was laid out for), one wait, and the live records go to the caller's buffers from pinned memory
        const size_t nb = (size_t)n_frames;
        if (n_frames == c->alloc_batch) {
            HIP_TRY(hipMemcpyAsync(c->h_out_block, c->d_out_block, c->out_block_bytes, hipMemcpyDeviceToHost, s));
        } else {
            HIP_TRY(hipMemcpyAsync(c->h_out_block, c->d_out_block, nb * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(c->h_out_block + c->out_kp_off, c->d_out_block + c->out_kp_off, nb * scap * sizeof(orbx_kp),
                                   hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(c->h_out_block + c->out_desc_off, c->d_out_block + c->out_desc_off, nb * scap * 32,
                                   hipMemcpyDeviceToHost, s));
        }
        return ORBX_OK;
    };
    // (Replaying the call as one captured hipGraph -- kernels on three streams plus the record copy -- was measured and is
    // not used: 295 us per 1242x375 frame against 218 us for the eager launches, ROCm 7.2.)
    if (c->h_out_dev && c->zero_copy) { // the last kernel writes the records into the pinned block itself
        rc = enqueue_batch(c, s, c->d_l0_stage, l0_fs, l0_pitch, n_frames, reinterpret_cast<orbx_kp *>(c->h_out_dev + c->out_kp_off),
                           c->h_out_dev + c->out_desc_off, scap, reinterpret_cast<int32_t *>(c->h_out_dev), true);
        if (rc) return rc;
    } else {
        rc = enqueue_batch(c, s, c->d_l0_stage, l0_fs, l0_pitch, n_frames, c->d_out_kp, c->d_out_desc, scap, c->d_out_n, true);
        if (rc) return rc;
        if (c->h_out_block) { rc = copy_records(); if (rc) return rc; }
    }
    tr.mark("kernel launches");
    int status = ORBX_OK;
    if (c->h_out_block) {
        HIP_TRY(hipStreamSynchronize(s));
        tr.mark("device + records");
        const int32_t *counts = reinterpret_cast<const int32_t *>(c->h_out_block);
        for (int f = 0; f < n_frames; ++f) {
            const int n = counts[f];
            n_out[f] = n;
            if (n == 0) continue; // reference :512 -- outputs untouched
            if (n > cap) { status = ORBX_E_CAPACITY; continue; }
            memcpy(out_kp + (size_t)f * cap, c->h_out_block + c->out_kp_off + (size_t)f * scap * sizeof(orbx_kp), sizeof(orbx_kp) * n);
            memcpy(out_desc + (size_t)f * cap * 32, c->h_out_block + c->out_desc_off + (size_t)f * scap * 32, (size_t)n * 32);
        }
    } else {
        std::vector<int32_t> counts(n_frames);
        HIP_TRY(hipMemcpyAsync(counts.data(), c->d_out_n, sizeof(int32_t) * n_frames, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        tr.mark("device + counts");
        for (int f = 0; f < n_frames; ++f) {
            const int n = counts[f];
            n_out[f] = n;
            if (n == 0) continue; // reference :512 -- outputs untouched
            if (n > cap) { status = ORBX_E_CAPACITY; continue; }
            HIP_TRY(hipMemcpyAsync(out_kp + (size_t)f * cap, c->d_out_kp + (size_t)f * scap, sizeof(orbx_kp) * n,
                                   hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_desc + (size_t)f * cap * 32, c->d_out_desc + (size_t)f * scap * 32, (size_t)n * 32,
                                   hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
    }
    tr.mark("records out");
    if (status == ORBX_E_CAPACITY) return fail(status, "output capacity too small");
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_t *c, const uint8_t *img, int width, int height, int stride, orbx_kp *out_kp,
                            uint8_t *out_desc, int cap, int *n_out)
{
    int32_t n = 0;
    int rc = orbx_extract_batch(c, img, 1, width, height, stride, (size_t)stride * (height > 0 ? height : 0), out_kp,
                                out_desc, cap, &n);
    if (n_out) *n_out = n;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// stage taps
// ------------------------------------------------------------------------------------------------
extern "C" int orbx_tap_level(orbx_t *c, int frame, int level, int blurred, uint8_t *out, size_t out_bytes)
{
    if (!c || !out || c->cur_w < 0 || !c->last_l0) return fail(ORBX_E_ARG, "no extract call yet");
    if (frame < 0 || frame >= c->last_frames || level < 0 || level >= c->levels.n_levels)
        return fail(ORBX_E_ARG, "frame/level out of range");
    const OrbxLevel &v = c->levels.lv[level];
    if (out_bytes < (size_t)v.w * v.h) return fail(ORBX_E_CAPACITY, "buffer too small");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_handle(c));
    if (blurred && level < c->last_fused_levels) {
        // the last call described this level with k_blur_desc, which keeps the blurred rows in LDS only: the blur pass makes
        // the copy now (all frames of the call, so that later taps find it), on the raw level that is still in place
        if (level < c->blur_mfma_levels)
            orbx_launch_blur_mfma(c->stream, c->last_l0, c->last_l0_fs, c->last_l0_pitch, c->levels, c->buf, c->blur_tab, c->d_blur_strips,
                                  c->blur_strips_before, c->d_band_h, c->d_band_v, c->taps, c->last_frames, level, level + 1);
        else
            orbx_launch_blur(c->stream, c->last_l0, c->last_l0_fs, c->last_l0_pitch, c->d_levels, c->levels, c->buf, c->d_blur_tiles,
                             c->n_blur_tiles, c->d_taps, c->last_frames, level, level + 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    const uint8_t *src; size_t pitch;
    if (blurred) { src = c->buf.img_arena + (size_t)frame * c->buf.img_frame_stride + v.blur_off; pitch = v.pitch; }
    else if (level == 0) { src = c->last_l0 + (size_t)frame * c->last_l0_fs; pitch = c->last_l0_pitch; }
    else { src = c->buf.img_arena + (size_t)frame * c->buf.img_frame_stride + v.raw_off; pitch = v.pitch; }
    HIP_TRY(hipMemcpy2D(out, v.w, src, pitch, v.w, v.h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_tap_candidates(orbx_t *c, int frame, int level, uint16_t *xs, uint16_t *ys, uint8_t *resp, int cap,
                                   int *n_out)
{
    if (!c || !n_out || c->cur_w < 0 || !c->last_l0) return fail(ORBX_E_ARG, "no extract call yet");
    if (frame < 0 || frame >= c->last_frames || level < 0 || level >= c->levels.n_levels)
        return fail(ORBX_E_ARG, "frame/level out of range");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_handle(c));
    int n = 0;
    HIP_TRY(hipMemcpy(&n, c->buf.cand_count + frame * ORBX_MAX_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) return fail(ORBX_E_CAPACITY, "buffer too small");
    std::vector<unsigned long long> tmp(std::max(n, 1));
    HIP_TRY(hipMemcpy(tmp.data(), c->buf.cand + (size_t)frame * c->buf.cand_frame_stride + c->levels.lv[level].cand_off,
                      sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        if (xs) xs[i] = (uint16_t)(tmp[i] & 0xFFFF);
        if (ys) ys[i] = (uint16_t)((tmp[i] >> 16) & 0xFFFF);
        if (resp) resp[i] = (uint8_t)(tmp[i] >> 32);
    }
    return ORBX_OK;
}

extern "C" int orbx_tap_level_counts(orbx_t *c, int frame, int32_t *counts)
{
    if (!c || !counts || c->cur_w < 0 || !c->last_l0) return fail(ORBX_E_ARG, "no extract call yet");
    if (frame < 0 || frame >= c->last_frames) return fail(ORBX_E_ARG, "frame out of range");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_handle(c));
    HIP_TRY(hipMemcpy(counts, c->buf.sel_count + frame * ORBX_MAX_LEVELS, sizeof(int) * c->levels.n_levels,
                      hipMemcpyDeviceToHost));
    return ORBX_OK;
}

void launch_tap_sincos(const float *d_ang, int n, float2 *d_out, hipStream_t st);
extern "C" int orbx_tap_sincos(orbx_t *c, const float *angles_deg, int n, float *cos_sin)
{
    if (!c || !angles_deg || !cos_sin || n < 0) return fail(ORBX_E_ARG, "bad argument");
    if (n == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(c->device));
    float *d_in = nullptr;
    float2 *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, sizeof(float) * (size_t)n));
    hipError_t e = hipMalloc(&d_out, sizeof(float2) * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, angles_deg, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_tap_sincos(d_in, n, d_out, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(cos_sin, d_out, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    HIP_TRY(e);
    return ORBX_OK;
}

extern "C" int orbx_set_stage_timing(orbx_t *c, int enable)
{
    if (!c) return fail(ORBX_E_ARG, "null handle");
    c->timing = enable;
    c->ev_valid = false;
    for (int i = 0; i < ORBX_N_STAGES; ++i) c->ev_in_n[i] = 0;
    if (enable == 2 && !c->ev_in[0][0]) {
        HIP_TRY(hipSetDevice(c->device));
        for (int st = ORBX_N_STAGES - 1; st >= 0; --st) // [0][0] last: it marks the set as complete
            for (int i = 3; i >= 0; --i) HIP_TRY(hipEventCreate(&c->ev_in[st][i]));
    }
    return ORBX_OK;
}

extern "C" int orbx_stream_wait_fast(orbx_t *c, void *stream)
{
    if (!c || !stream) return fail(ORBX_E_ARG, "null argument");
    if (!c->after_fast_valid) return fail(ORBX_E_ARG, "no batched extract call enqueued yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->ev_after_fast, 0));
    return ORBX_OK;
}

extern "C" int orbx_stage_times_in_step_ms(orbx_t *c, float *ms)
{
    if (!c || !ms) return fail(ORBX_E_ARG, "null argument");
    if (c->timing != 2 || c->ev_in_n[ORBX_STAGE_FAST] < 1) return fail(ORBX_E_ARG, "no extract call in timing mode 2 yet");
    // a batch split over internal streams re-records the same events once per frame range: only the last range would be timed
    if (c->n_sub > 1) return fail(ORBX_E_UNSUPPORTED, "in-step stage times need ORBX_VAR_STREAMS = 1");
    HIP_TRY(hipSetDevice(c->device));
    for (int st = 0; st < ORBX_N_STAGES; ++st) {
        ms[st] = 0.f;
        for (int i = 0; i < c->ev_in_n[st]; ++i) {
            float t = 0.f;
            hipEvent_t e0 = c->ev_in[st][2 * i];
            HIP_TRY(hipEventSynchronize(c->ev_in[st][2 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&t, e0, c->ev_in[st][2 * i + 1]));
            ms[st] += t;
        }
    }
    return ORBX_OK;
}

extern "C" int orbx_fast_times_in_step_ms(orbx_t *c, float *ms_sum, int *n_launches)
{
    if (!c || !ms_sum) return fail(ORBX_E_ARG, "null argument");
    float ms[ORBX_N_STAGES];
    int rc = orbx_stage_times_in_step_ms(c, ms);
    if (rc) return rc;
    *ms_sum = ms[ORBX_STAGE_FAST];
    if (n_launches) *n_launches = c->ev_in_n[ORBX_STAGE_FAST];
    return ORBX_OK;
}

extern "C" int orbx_stage_times_ms(orbx_t *c, float *ms)
{
    if (!c || !ms) return fail(ORBX_E_ARG, "null argument");
    if (!c->ev_valid) return fail(ORBX_E_ARG, "no timed extract call yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[ORBX_N_STAGES]));
    for (int i = 0; i < ORBX_N_STAGES; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    return ORBX_OK;
}

static int *variant_field(orbx_ctx *c, int which, int *lo, int *hi)
{
    switch (which) {
    case ORBX_VAR_FAST: *lo = 0; *hi = 2; return &c->fast_variant;
    case ORBX_VAR_BLUR: *lo = 0; *hi = 2; return &c->blur_mfma;
    case ORBX_VAR_RESIZE_LDS: *lo = 0; *hi = 2; return &c->resize_lds;
    case ORBX_VAR_RESIZE2: *lo = 0; *hi = 2; return &c->resize2;
    case ORBX_VAR_SIDE_BLUR: *lo = 0; *hi = 3; return &c->side_blur;
    case ORBX_VAR_EARLY_FAST: *lo = -1; *hi = 2; return &c->early_fast;
    case ORBX_VAR_SPLIT_LEVEL0: *lo = 0; *hi = ORBX_MAX_LEVELS; return &c->split_level0;
    case ORBX_VAR_STREAMS: *lo = 1; *hi = 8; return &c->n_sub;
    case ORBX_VAR_ZERO_COPY: *lo = 0; *hi = 1; return &c->zero_copy;
    case ORBX_VAR_DESC: *lo = 0; *hi = 2; return &c->desc_variant;
    case ORBX_VAR_FAST_CELL_GROUP: *lo = 1; *hi = 16; return &c->fast_cell_group;
    default: return nullptr;
    }
}

extern "C" int orbx_set_variant(orbx_t *c, int which, int value)
{
    if (!c) return fail(ORBX_E_ARG, "null handle");
    int lo, hi;
    int *f = variant_field(c, which, &lo, &hi);
    if (!f) return fail(ORBX_E_ARG, "unknown variant switch");
    if (value < lo || value > hi) return fail(ORBX_E_ARG, "variant value out of range");
    // only the instantiated group sizes (orbx_get_variant must report what runs, not a value the launcher rounds down)
    if (which == ORBX_VAR_FAST_CELL_GROUP && value != 1 && value != 4 && value != 8 && value != 16)
        return fail(ORBX_E_ARG, "FAST cell group must be 1, 4, 8 or 16");
    // the first level of the main chain lies inside the pyramid: 0 (one chain) .. n_levels - 1
    if (which == ORBX_VAR_SPLIT_LEVEL0 && value > c->cfg.n_levels - 1)
        return fail(ORBX_E_ARG, "split level must be below the handle's n_levels");
    *f = value;
    return ORBX_OK;
}

extern "C" int orbx_get_variant(const orbx_t *c, int which, int *value)
{
    if (!c || !value) return fail(ORBX_E_ARG, "null argument");
    int lo, hi;
    int *f = variant_field(const_cast<orbx_ctx *>(c), which, &lo, &hi);
    if (!f) return fail(ORBX_E_ARG, "unknown variant switch");
    *value = *f;
    return ORBX_OK;
}

extern "C" int orbx_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(ORBX_E_ARG, "null argument");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return ORBX_OK;
}
extern "C" int orbx_host_unregister(void *ptr)
{
    if (!ptr) return fail(ORBX_E_ARG, "null argument");
    HIP_TRY(hipHostUnregister(ptr));
    return ORBX_OK;
}

// Planning query for tests and tools, no device needed: which quadtree build orbx_launch_octree would take for levels
// [level_begin, level_end) of a call with n_frames frames of w0 x h0 under `cfg` (quotas of n_features, or of `requota` when
// that is > 0: orbx_create_requota's handle), and with how much dynamic LDS.  out[0] = kind (0 batch, 1 wide, 2 huge then wide,
// 3 list in global scratch), out[1] = the list arrays' LDS bytes, out[2] = bytes the 1024-thread build is launched with,
// out[3] = first level of the range that does NOT take the 1024-thread build.
extern "C" int orbx_dev_octree_plan(const orbx_cfg *cfg, int requota, int w0, int h0, int n_frames, int level_begin, int level_end,
                                    int64_t *out)
{
    if (!cfg || !out || cfg->n_levels < 1 || cfg->n_levels > ORBX_MAX_LEVELS || cfg->n_features < 1 || !(cfg->scale_factor > 1.0f))
        return fail(ORBX_E_ARG, "bad argument");
    if (level_begin < 0 || level_end > cfg->n_levels || level_end <= level_begin || n_frames < 1) return fail(ORBX_E_ARG, "bad level range");
    orbx_ctx *c = new orbx_ctx();
    c->cfg = *cfg;
    compute_tables(c);
    if (requota > 0) compute_quotas(c, requota);
    Geometry g;
    int rc = compute_geometry(c, w0, h0, &g);
    delete c;
    if (rc) return rc;
    const OrbxOctPlan p = orbx_octree_plan(g.levels, n_frames, level_begin, level_end);
    out[0] = p.kind; out[1] = (int64_t)p.lds_bytes; out[2] = (int64_t)p.huge_bytes; out[3] = p.huge_end;
    return ORBX_OK;
}

extern "C" const char *orbx_last_error(void) { return g_err.c_str(); }
extern "C" const char *orbx_version(void) { return "orbx 0.1 (gfx950)"; }

#ifdef OCT_PROF
unsigned long long *orbx_dev_fast_prof_symbol();
// development build only (make prof): k_fast_strip's stage counters since the last reset (tools/fast_mix.py)
extern "C" int orbx_dev_fast_prof(orbx_t *c, unsigned long long *out16, int reset)
{
    if (!c || !out16) return fail(ORBX_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long *sym = orbx_dev_fast_prof_symbol();
    if (!sym) return fail(ORBX_E_ARG, "no counter symbol");
    HIP_TRY(hipMemcpy(out16, sym, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(sym, 0, sizeof(unsigned long long) * 16));
    return ORBX_OK;
}
unsigned long long *orbx_dev_fast_cell_times_symbol();
// development build only (make prof): per-cell time stamps of k_fast_cells_wave, frame 0 (tools/fast_cell_times.py)
extern "C" int orbx_dev_fast_cell_times(orbx_t *c, unsigned long long *out, int n_slots, int reset)
{
    if (!c || !out) return fail(ORBX_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long *sym = orbx_dev_fast_cell_times_symbol();
    if (!sym) return fail(ORBX_E_ARG, "no symbol");
    HIP_TRY(hipMemcpy(out, sym, sizeof(unsigned long long) * 3 * n_slots, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(sym, 0, sizeof(unsigned long long) * 3 * n_slots));
    return ORBX_OK;
}
// development build only (make prof): the quadtree kernel's phase time stamps of frame 0 (tools/octree_phases.py)
extern "C" int orbx_dev_octree_phases(orbx_t *c, unsigned long long *out, int n_levels)
{
    if (!c || !out) return fail(ORBX_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_handle(c));
    HIP_TRY(hipMemcpy(out, c->buf.best, sizeof(unsigned long long) * 64 * n_levels, hipMemcpyDeviceToHost));
    return ORBX_OK;
}
#endif
