// Internal declarations shared by the extractor's kernels and its C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/orbx.h"

#define ORBX_EDGE 19        // EDGE_THRESHOLD (reference modules/ORB/ORBExtractor.cpp:15)
#define ORBX_CELL 30        // W              (reference modules/ORB/ORBExtractor.cpp:575)
#define ORBX_HALF_PATCH 15  // HALF_PATCH_SIZE(reference modules/ORB/ORBExtractor.cpp:14)
#ifndef ORBX_OCT_THREADS
#define ORBX_OCT_THREADS 512       // quadtree workgroup of a call with a few frames (at most 512: the block scans keep 8 wave totals)
#endif
#ifndef ORBX_OCT_THREADS_BATCH
#define ORBX_OCT_THREADS_BATCH 256 // ... and of a resident batch (orbx_octree.h)
#endif

// More than 64 KB of dynamic LDS has to be requested per kernel AND per device (the attribute belongs to the device's code
// object).  orbx_lds_opt_in keeps the largest size configured so far per (kernel, device of the calling thread) under a mutex,
// raises it when `bytes` is larger, and returns the runtime's answer -- a handle on a second GPU of the process gets its
// own opt-in, and a failed one is reported instead of being found out by a launch error later.
hipError_t orbx_lds_opt_in(const void *kernel, size_t bytes);

// Geometry of one pyramid level and where its buffers live inside the per-frame arenas.
struct OrbxLevel {
    int w, h;           // level size
    int pitch;          // bytes per row in the arena (multiple of 64)
    int n_cols, n_rows; // FAST cells (30x30) over [19,w-19) x [19,h-19)
    int region_w, region_h;
    int quota;          // n_features_per_level
    int n_ini, h_x;     // DistributeOctree initial nodes (reference :645-646)
    int kcap;           // upper bound on keypoints after the quadtree
    int kp_off;         // prefix sum of kcap (slot offset in the per-frame selected array)
    int node_cap;       // quadtree list capacity
    int cand_cap;       // candidate capacity (sum over cells of ceil(cw/2)*ceil(ch/2))
    float scale;        // scale_factors[level]
    size_t raw_off;     // byte offset of the raw level inside a frame's image arena (level 0: unused)
    size_t blur_off;    // byte offset of the blurred level
    size_t cand_off;    // record offset inside a frame's candidate arena
    size_t node_off;    // node-slot offset inside a frame's quadtree scratch
};

// Per-dst-column / per-dst-row bilinear tap (cv::resize INTER_LINEAR fixed point)
struct OrbxTap {
    int32_t ofs;
    int16_t c0, c1;
};

struct OrbxLevels {
    int n_levels;
    int ini_th, min_th;
    int kcap_total;
    OrbxLevel lv[ORBX_MAX_LEVELS];
};

// Device pointers of one extractor handle (all per-frame arenas are frame-major).
struct OrbxBuffers {
    uint8_t *img_arena;        size_t img_frame_stride;   // raw levels 1.. + blurred levels 0..
    unsigned long long *cand;  size_t cand_frame_stride;  // packed candidates
    int *cand_count;                                      // [frame][level]
    uint32_t *pnode;                                      // [frame][cand slot] quadtree node of each candidate
    uint32_t *pcode;                                      // [frame][cand slot] 16 x 2-bit quadtree descent of each candidate
    // quadtree scratch, per frame `node_frame_stride` node slots
    size_t node_frame_stride;
    short4 *bnd0, *bnd1;
    int *cnt0, *cnt1;
    int *rank, *node_of_rank, *newpos;
    int *childcnt, *childpos;                             // 4 per node slot
    unsigned long long *best;
    uint2 *sel;                                           // [frame][kcap_total] selected keypoints (x | y<<16, response)
    int *sel_count;                                       // [frame][level]
    int *sel_prefix;                                      // [frame][level] key points of the levels before (written by k_orient)
    const int *slot_level;                                // [kcap_total] level of a key-point slot (per geometry, not per frame)
    float4 *kp_ang;                                       // [frame][kcap_total] (angle, cos, sin, -) per selected keypoint
};

void orbx_launch_resize(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh,
                        uint8_t *dst, size_t dst_fs, int dst_pitch, int dw, int dh,
                        const OrbxTap *xtap, const OrbxTap *ytap, int n_frames, int *zero_counts = nullptr);
// the source tile staged through LDS (k_resize_lds, resident batches); usable when orbx_resize_lds_fits.  last_row_bytes: readable
// bytes of the source's last row (its width for a caller's image, its pitch for an arena level)
bool orbx_resize_lds_fits(const OrbxTap *xtap, const OrbxTap *ytap, int sw, int sh, int dw, int dh);
void orbx_launch_resize_lds(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh, int last_row_bytes,
                            uint8_t *dst, size_t dst_fs, int dst_pitch, int dw, int dh, const OrbxTap *xtap, const OrbxTap *ytap,
                            int n_frames, int *zero_counts);
// two levels per launch (k_resize2): src = level l, d1 = level l+1, d2 = level l+2; usable when orbx_resize2_fits
bool orbx_resize2_fits(const OrbxTap *xtap2, const OrbxTap *ytap2, int w1, int h1, int w2, int h2);
void orbx_launch_resize2(hipStream_t s, const uint8_t *src, size_t src_fs, int src_pitch, int sw, int sh, uint8_t *d1, size_t d1_fs,
                         int d1_pitch, int w1, int h1, const OrbxTap *xtap1, const OrbxTap *ytap1, uint8_t *d2, size_t d2_fs,
                         int d2_pitch, int w2, int h2, const OrbxTap *xtap2, const OrbxTap *ytap2, int n_frames, int *zero_counts);
void orbx_launch_fast(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                      const OrbxLevels &levels, const OrbxBuffers &b, const void *d_cells, int n_cells, int n_frames,
                      int cells_per_group);
int orbx_build_fast_cells(const OrbxLevels &levels, uint16_t *out);
// strips of up to three cells (one wave each), level-major; levels [level_begin, level_end)
int orbx_build_fast_strips(const OrbxLevels &levels, int level_begin, int level_end, uint16_t *out);
void orbx_launch_fast_strips(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                             const OrbxLevels &levels, const OrbxBuffers &b, const void *d_strips, int n_strips, int n_frames);
void orbx_launch_blur(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                      const OrbxLevels &levels, const OrbxBuffers &b, const void *d_tiles, int n_tiles, const int *taps7,
                      int n_frames, int level_begin, int level_end);
int orbx_build_blur_tiles(const OrbxLevels &levels, uint16_t *out);
// the Gaussian on the matrix pipe (orbx_kernels.hip, k_blur_mfma): levels 0 .. orbx_blur_mfma_levels() - 1
struct BlurMfmaLevels { // per-level record of k_blur_mfma, passed by value
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], dst_pitch[ORBX_MAX_LEVELS], n_ty[ORBX_MAX_LEVELS];
    unsigned long long dst_off[ORBX_MAX_LEVELS];
    int bh_off[ORBX_MAX_LEVELS], bv_off[ORBX_MAX_LEVELS]; // in uint4, into the band tables
};
int orbx_blur_mfma_levels(const OrbxLevels &levels);
void orbx_build_blur_mfma(const OrbxLevels &levels, const int taps[7], std::vector<uint16_t> &strips, std::vector<uint8_t> &band_h,
                          std::vector<uint8_t> &band_v, BlurMfmaLevels &out, int strips_before[ORBX_MAX_LEVELS + 1]);
void orbx_launch_blur_mfma(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels &levels,
                           const OrbxBuffers &b, const BlurMfmaLevels &tab, const void *d_strips, const int *strips_before,
                           const void *d_band_h, const void *d_band_v, const int taps[7], int n_frames, int level_begin,
                           int level_end);
// blur and descriptors in one pass (k_blur_desc): per-level record, passed by value
struct BdLevels {
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], n_ty[ORBX_MAX_LEVELS], n_bx[ORBX_MAX_LEVELS];
    int bh_off[ORBX_MAX_LEVELS], bv_off[ORBX_MAX_LEVELS]; // in uint4, into the H-band table of k_blur_desc / the V-band table shared with k_blur_mfma
    int bucket_base[ORBX_MAX_LEVELS];                     // first (block, trip) bucket of the level inside a frame's bucket-start array
    float scale[ORBX_MAX_LEVELS];
};
void orbx_build_blur_desc(const OrbxLevels &levels, const int taps[7], const BlurMfmaLevels &mf, std::vector<uint16_t> &blocks,
                          std::vector<uint8_t> &band_h, BdLevels &out, int *n_fused_levels, int *bk_stride);
// k_desc_bins (before the orientation: it fixes the order the key points of levels [0, n_fused_levels) are processed in) and
// k_blur_desc (after it); d_items: 32 bytes per key-point slot and frame, d_bk_start: bk_stride ints per frame
void orbx_launch_desc_bins(hipStream_t s, const OrbxLevels *d_levels, const OrbxBuffers &b, const BdLevels &tab, int n_fused_levels,
                           int *d_bk_start, int bk_stride, void *d_items, int cap, int32_t *out_n, int n_frames);
void orbx_launch_desc_fused(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels &levels, const OrbxBuffers &b,
                            const BdLevels &tab, int n_fused_levels, const void *d_blocks, int n_blocks, const void *d_band_h,
                            const void *d_band_v, const int *d_bk_start, int bk_stride, const void *d_items, const int taps[7],
                            orbx_kp *out_kp, uint8_t *out_desc, int cap, int n_frames);
#define ORBX_OCT_LDS_LIMIT (160 * 1024 - 256)  // the LDS-resident node list of a level must fit this (else the global-scratch build)
#define ORBX_OCT_HUGE_LDS (160 * 1024 - 1024)  // what the 1024-thread build is launched with: list arrays + count pyramid
#define ORBX_N_CUS 256                         // MI355X: the default of the device-less planning query; a handle asks its device (orbx_create)
struct OrbxOctPlan { int kind; size_t lds_bytes, huge_bytes; int huge_end; };
OrbxOctPlan orbx_octree_plan(const OrbxLevels &levels, int n_frames, int level_begin, int level_end, int n_cus = ORBX_N_CUS);
void orbx_launch_octree(hipStream_t s, const OrbxLevels *d_levels, const OrbxLevels &levels, const OrbxBuffers &b,
                        int n_frames, size_t sort_lds_bytes, int level_begin, int level_end, int n_cus = ORBX_N_CUS);
void orbx_launch_orient_desc(hipStream_t s, const uint8_t *l0, size_t l0_fs, int l0_pitch, const OrbxLevels *d_levels,
                             const OrbxLevels &levels, const OrbxBuffers &b, const int *u_max, orbx_kp *out_kp,
                             uint8_t *out_desc, int cap, int32_t *out_n, int n_frames, hipEvent_t blur_done, int desc_level_min = 0,
                             hipEvent_t after_orient = nullptr, void *d_items = nullptr, hipEvent_t desc_open = nullptr);
