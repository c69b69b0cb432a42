/*
 * orbm.h -- C ABI of the MI355X ORB matcher cores (liborbx.so).
 *
 * Drop-in boundary for the reference's ORBMatcher
 * (modules/ORB/ORBMatcher.h:12-52, modules/ORB/ORBMatcher.cpp).  The 256-bit
 * Hamming brute force of every Search* routine runs as HIP kernels (the dense best / second-best search on the
 * matrix pipe; the window searches with Frame::grid and getFeaturesInArea on the device as well); the greedy,
 * order-dependent resolution (which mutates Frame/KeyFrame/MapPoint objects in
 * the reference) consumes the device-computed distances on the host so that the
 * results are identical to the reference's sequential loops.
 * The header-only shim monoorbslam3_amd/compat/ORBMatcher.h maps the
 * reference's Frame/KeyFrame types onto these plain-array entry points.
 *
 * Returns 0 or a negative ORBX_E_* code (orbx.h); text in orbx_last_error().
 * All entry points are re-entrant: a handle owns its (non-blocking) stream and scratch, and the
 * reference calls SearchForTriangulation and the fuse from the LocalMapping thread while
 * Tracking calls SearchByBow / SearchByProjection (LocalMapping.cpp:168, 282, 301; Tracking.cpp:262, 289) -- use one
 * handle per thread; handles never wait for each other (orbx.h, "Streams and threads"; tests/cpp/two_threads.cpp).
 */
#ifndef ORBM_H
#define ORBM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBM_TH_LOW 50        /* modules/ORB/ORBMatcher.cpp:13 */
#define ORBM_TH_HIGH 100      /* modules/ORB/ORBMatcher.cpp:14 */
#define ORBM_HISTO_LENGTH 30  /* modules/ORB/ORBMatcher.cpp:15 */

typedef struct orbm_ctx orbm_t;

int orbm_create(int device, orbm_t **out);
void orbm_destroy(orbm_t *h);
/* Kernel-choice switches per handle (parity twins; no reference counterpart).  They replace the ORBM_BEST2 / ORBM_WINDOW
 * environment variables of earlier builds.  Unknown switch / value out of range: ORBX_E_ARG. */
#define ORBM_VAR_BEST2 0   /* dense best / second-best: 0 FP4 matrix path k_best2_fp4 (default), 1 i8 matrix path k_best2_mfma, 2 VALU k_best2 */
#define ORBM_VAR_WINDOW 1  /* window searches of the host entry points: 0 grid and lists on the device (default), 1 host grid */
#define ORBM_VAR_BEST2_RESIDENT 2 /* k_best2_fp4's grid: 0 one workgroup per (problem, 512 queries) (default); 1 or 2: that many
                                   * workgroups per CU walk the blocks, so the kernel holds a fixed share of every CU (half of the
                                   * registers and 37 KB of LDS per workgroup) -- for a caller that runs it beside other kernels */
#define ORBM_VAR_INIT_LANES 3 /* orbm_search_for_initialization_device: lanes that share one query's window list in the resolve kernel:
                              * 0 chosen from the mean list length (default), or 1, 4, 16, 64 (the parity twins) */
int orbm_set_variant(orbm_t *h, int which, int value);

/* DBoW2::FeatureVector (thirdParty/DBoW2/DBoW2/FeatureVector.h) flattened to CSR:
 * node ids ascending, indices of a node in insertion (ascending feature) order. */
typedef struct orbm_fv {
    int32_t n_nodes;
    const uint32_t *node_ids;
    const int32_t *offsets; /* n_nodes + 1 */
    const uint32_t *indices;
} orbm_fv;

/* ORBMatcher::DescriptorDistance (modules/ORB/ORBMatcher.cpp:17-31) for all pairs:
 * out[i*nb + j] = popcount(a[i] ^ b[j]), 0..256.  Host pointers. */
int orbm_hamming_matrix(orbm_t *h, const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *out);
/* same with device pointers, enqueued on `stream` (hipStream_t; NULL: orbx.h, "Streams") */
int orbm_hamming_matrix_device(orbm_t *h, const uint8_t *d_a, int na, const uint8_t *d_b, int nb, uint16_t *d_out,
                               void *stream);

/* Best / second-best of every query row among the candidate rows, the inner loop of
 * SearchByBow (ORBMatcher.cpp:148-162): strict '<' updates in ascending candidate order,
 * both distances start at 256, best index -1 when no candidate.  row_ok / col_ok
 * (may be NULL) are byte masks: skipped queries return (-1,256,256); masked candidates
 * are not considered.  `n_pairs` independent (A,B) problems are processed in one launch:
 * problem p uses a + p*a_stride ... (strides in descriptors/elements).  Device pointers.
 * Which kernel runs: problems without a candidate mask (d_col_ok == NULL) and nb_max <= 8160 take a matrix-pipe
 * kernel (k_best2_fp4, or k_best2_mfma with ORBM_VAR_BEST2 = 1); a candidate mask, more candidates, or ORBM_VAR_BEST2 = 2
 * take the VALU kernel (k_best2).  All three give the same outputs; only the speed differs. */
int orbm_best2_device(orbm_t *h, int n_pairs, const uint8_t *d_a, size_t a_stride, const int32_t *d_na, int na_max,
                      const uint8_t *d_b, size_t b_stride, const int32_t *d_nb, int nb_max,
                      const uint8_t *d_row_ok, const uint8_t *d_col_ok, int32_t *d_best_idx, uint16_t *d_best,
                      uint16_t *d_second, void *stream);
/* host-pointer convenience wrapper for one problem */
int orbm_best2(orbm_t *h, const uint8_t *a, int na, const uint8_t *b, int nb, const uint8_t *row_ok,
               const uint8_t *col_ok, int32_t *best_idx, uint16_t *best, uint16_t *second);

/* Distances for explicit candidate lists (CSR): for query q (descriptor a[q_idx[q]]) and
 * t in [off[q], off[q+1]): out[t] = hamming(a[q_idx[q]], b[c_idx[t]]).  Host pointers.
 * This is the device primitive under every window / BoW-node search. */
int orbm_hamming_csr(orbm_t *h, const uint8_t *a, int na, const uint8_t *b, int nb, const int32_t *q_idx,
                     const int32_t *off, int n_queries, const int32_t *c_idx, uint16_t *out);

/* ORBMatcher::SearchByBow(keyFrame, frame) (modules/ORB/ORBMatcher.cpp:118-201).
 * kf_mp_ok[i] != 0 <=> keyFrame->getMapPoints()[i] is non-null and not bad (:143).
 * frame_mp[j] (in/out): -1 where frame->map_points[j] is null; on return matched
 * entries hold the key-frame feature index whose MapPoint the reference would assign (:165).
 * Returns the match count through n_matches. */
int orbm_search_by_bow(orbm_t *h, float nn_ratio, int check_orientation,
                       const uint8_t *desc1, const float *angle1, const uint8_t *kf_mp_ok, int n1, const orbm_fv *fv1,
                       const uint8_t *desc2, const float *angle2, int32_t *frame_mp, int n2, const orbm_fv *fv2,
                       int *n_matches);

/* ORBMatcher::SearchForTriangulation(kf1, kf2, matches12) (modules/ORB/ORBMatcher.cpp:417-522),
 * including its `bestIdx2 > 0` acceptance rule (:484).  has_mp1/2[i] != 0 <=> hasMapPoint(i). */
int orbm_search_for_triangulation(orbm_t *h, int check_orientation,
                                  const uint8_t *desc1, const float *angle1, const uint8_t *has_mp1, int n1,
                                  const orbm_fv *fv1,
                                  const uint8_t *desc2, const float *angle2, const uint8_t *has_mp2, int n2,
                                  const orbm_fv *fv2, int32_t *matches12, int *n_matches);

/* ORBMatcher::SearchForInitialization(frame1, frame2, vecPreMatched, matches12, windowSize)
 * (modules/ORB/ORBMatcher.cpp:33-116) with Frame::getFeaturesInArea (Frame.cpp:97-127) over
 * plain arrays.  kps are orbx_kp-layout records (28 bytes).  prematched: n1 (x,y) pairs, in/out. */
int orbm_search_for_initialization(orbm_t *h, float nn_ratio, int check_orientation,
                                   const void *kps1, const uint8_t *desc1, int n1,
                                   const void *kps2, const uint8_t *desc2, int n2,
                                   int img_w, int img_h, float *prematched_xy, int32_t *matches12,
                                   int window_size, int *n_matches);

/* ORBMatcher::SearchByProjection(lastFrame, curFrame, th) and (lastKF, curFrame, th)
 * (modules/ORB/ORBMatcher.cpp:203-274 and :276-348 -- the two bodies are the same loop).  The caller (shim)
 * keeps the camera / pose maths: for every feature i of the last frame it passes q_ok[i] = 1 iff the feature has
 * a live MapPoint whose projection is in front of the camera and inside the image (:213-224), the projection
 * q_xy, the radius th * key_points[i].size, the octave (window levels octave-1 .. octave+1, :226-229), the
 * MapPoint descriptor and the key-point angle.  frame_mp (in/out): -1 where curFrame->map_points[j] is null,
 * any other value = occupied; matched entries are set to the query index i (:245).  kps2 = orbx_kp records. */
int orbm_search_by_projection_frame(orbm_t *h, int check_orientation,
                                    const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                                    const int32_t *q_octave, const float *q_angle, const uint8_t *q_ok, int nq,
                                    const void *kps2, const uint8_t *desc2, int n2, int img_w, int img_h,
                                    int32_t *frame_mp, int *n_matches);

/* ORBMatcher::SearchByProjection(frame, mapPoints, th) (modules/ORB/ORBMatcher.cpp:350-415).  q_ok[i] = 1 iff
 * mp->track_in_view && !mp->isBad() (:355); q_xy = (track_proj_x, track_proj_y); q_radius = the radius computed at
 * :362-365; q_level = track_scale_level (window levels level-1 .. level, :367-369).  frame_mp (in/out): -1 where
 * frame->map_points[j] is null OR bad (such slots may be taken, :383), anything else = occupied by a good point;
 * matched entries are set to the query index.  counters[3] = {numOutViewAndBad, fail1, fail2} (:353, :403, :408). */
int orbm_search_by_projection_points(orbm_t *h, float nn_ratio,
                                     const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                                     const int32_t *q_level, const uint8_t *q_ok, int nq,
                                     const void *kps2, const uint8_t *desc2, int n2, int img_w, int img_h,
                                     int32_t *frame_mp, int *n_matches, int32_t *counters);

/* static ORBMatcher::SearchByProjection(keyFrame, mapPoints, Map *pointMap, th) -- the map-point fuse that
 * LocalMapping.cpp:282,301 calls (modules/ORB/ORBMatcher.h:44-45, ORBMatcher.cpp:524-592).  This entry point is the part
 * of the loop body that reads no MapPoint / KeyFrame state: for map point i with q_ok[i] = 1 (the caller evaluated
 * :534-552: projection in front of the camera and in the image, distance invariance, viewing angle) it walks
 * KeyFrame::getFeaturesInArea(p.x, p.y, radius, predictLevel-1, predictLevel) (KeyFrame.cpp:181-211, STRICT window
 * test), drops key points whose squared re-projection error exceeds 5.991 * sigma2[octave] (:566-567, float against
 * double) and returns the closest descriptor with distance < TH_LOW + 1 (:560, :569-574, first on ties):
 * best_idx[i] = key-point index or -1, best_dist[i] = its distance (TH_LOW + 1 when none).  q_radius[i] =
 * th * scale_factor[predictLevel] (:555), sigma2 = ORBExtractor::getSquareSigmas() (n_levels entries).
 * The caller replays :534 (null / bad / already observed, evaluated live) and :577-591 (addObservation / replace)
 * in map-point order on its own objects -- see compat/ORBMatcher.h. */
int orbm_search_fuse(orbm_t *h, const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                     const int32_t *q_level, const uint8_t *q_ok, int nq,
                     const void *kps, const uint8_t *desc, int n, int img_w, int img_h,
                     const float *sigma2, int n_levels, int32_t *best_idx, int32_t *best_dist, int *n_found);

/* Frame / KeyFrame::getFeaturesInArea (modules/BasicObject/Frame.cpp:97-127, KeyFrame.cpp:181-211) plus the descriptor
 * distance of every hit, for nq queries against ONE device-resident frame record: d_kps / d_desc as written by
 * orbx_extract_batch_device (undistorted key points: orbf's d_kp_un), d_cell_start / d_cell_items = the CSR grid of
 * orbf_frame_post_device (grid_cols x grid_rows cells of 40 px, cell id = cx * grid_rows + cy).  Query q: centre
 * d_q_xy[2q..], radius d_q_radius[q], levels d_q_min_level[q] .. d_q_max_level[q] with the reference's beCheckLevel rule
 * (Frame.cpp:107), d_q_ok[q] = 0 switches it off.  strict != 0 = KeyFrame's `< r` test; d_sigma2 != NULL drops hits whose
 * squared distance to the centre exceeds 5.991 * d_sigma2[octave] (the fuse, ORBMatcher.cpp:566-567).
 * d_lists[q * cap + p] = distance << 22 | key-point index for the p-th hit in the reference's list order (cx outer, cy
 * inner, ascending index inside a cell), p < cap; d_counts[q] = the full list length (may exceed cap), -1 for a query
 * that is off.  Every pointer is device memory; enqueued on `stream` (NULL: orbx.h, "Streams").  This is the primitive under
 * the four window searches above, which wrap it with one staging copy each way when called with host pointers. */
int orbm_window_lists_device(orbm_t *h, const void *d_kps, const uint8_t *d_desc, const int32_t *d_cell_start,
                             const int32_t *d_cell_items, int grid_cols, int grid_rows, const uint8_t *d_q_desc,
                             const float *d_q_xy, const float *d_q_radius, const int32_t *d_q_min_level,
                             const int32_t *d_q_max_level, const uint8_t *d_q_ok, int nq, int strict,
                             const float *d_sigma2, int cap, int32_t *d_counts, uint32_t *d_lists, void *stream);

/* SearchByBow (Tracking.cpp:262 -> ORBMatcher.cpp:118-201) and SearchForTriangulation (LocalMapping.cpp:168 -> :417-522) on
 * device-resident records, greedy pass included: descriptors and key-point records (orbx_kp: the angle is read from them) as
 * orbx_extract_batch_device leaves them, the two
 * FeatureVectors as orbv_transform_device leaves them (node ids ascending, CSR offsets, feature indices, the node COUNT in device
 * memory), so the BoW branch of the tracking thread -- extract, computeBow, SearchByBow -- has no host hop either.
 *   d_kf_mp_ok [n1]   key-frame features that have a live map point (:141-146): the queries
 *   d_frame_mp [n2]   in / out as in the host entry point: -1 = free; a match writes the key-frame feature index
 *   d_has_mp1 / d_has_mp2: features that already have a map point (skipped, :452 / :466); d_matches12 [n1] receives the matches
 * A feature of side 2 lies in one vocabulary node, so the reference's order-dependent loop only couples queries of the same
 * node: every node is resolved by its own workgroup with the fixed point of the projection searches below, on the 8 closest
 * initially-free candidates per query (a query whose list is used up rescans its node on the device).  Rotation histogram
 * (the reference's 1/30 factor) and ComputeThreeMaxima on the device as well.
 * d_result (int32 x 8, device): [0] matches, [1] = 1 if a shared node holds more than 4096 features on a side -- found before
 * anything is written: NOTHING was changed then (frame_mp as passed, matches12 all -1, no match counted), as the projection
 * searches below guarantee on overflow, so the host entry point can take over on the same arrays --, [2] the most sweeps a
 * node needed, [3] matches before the rotation filter.
 * One call in flight per handle (the scratch is the handle's).  Enqueued on `stream` (NULL: orbx.h, "Streams"); no host wait. */
int orbm_search_by_bow_device(orbm_t *h, float nn_ratio, int check_orientation, const uint8_t *d_desc1, const void *d_kps1,
                              const uint8_t *d_kf_mp_ok, int n1, const uint32_t *d_fv1_nodes, const int32_t *d_fv1_off,
                              const uint32_t *d_fv1_idx, const int32_t *d_n_fv1, const uint8_t *d_desc2, const void *d_kps2,
                              int32_t *d_frame_mp, int n2, const uint32_t *d_fv2_nodes, const int32_t *d_fv2_off,
                              const uint32_t *d_fv2_idx, const int32_t *d_n_fv2, int32_t *d_result, void *stream);
int orbm_search_for_triangulation_device(orbm_t *h, int check_orientation, const uint8_t *d_desc1, const void *d_kps1,
                                         const uint8_t *d_has_mp1, int n1, const uint32_t *d_fv1_nodes, const int32_t *d_fv1_off,
                                         const uint32_t *d_fv1_idx, const int32_t *d_n_fv1, const uint8_t *d_desc2,
                                         const void *d_kps2, const uint8_t *d_has_mp2, int n2, const uint32_t *d_fv2_nodes,
                                         const int32_t *d_fv2_off, const uint32_t *d_fv2_idx, const int32_t *d_n_fv2,
                                         int32_t *d_matches12, int32_t *d_result, void *stream);

/* The two SearchByProjection calls of the tracking thread (Tracking.cpp:289-336) on a device-resident frame record, greedy
 * pass included (modules/ORB/ORBMatcher.cpp:229-246 and :379-407): nothing returns to the host between the extraction,
 * orbf_frame_post_device and the matched map points.  Queries (device arrays, as the host entry points above take them):
 * descriptors, projections, radii, octave / predicted level, angle (frame -> frame only), q_ok.  The frame: d_kps2 / d_desc2 /
 * CSR grid as for orbm_window_lists_device, n2 key points.  d_frame_mp [n2], in/out, the meaning of the host entry points:
 * -1 = free, anything else = occupied; matched entries receive the query index.
 * The greedy order of the reference -- query i takes its closest candidate not taken by a query before it -- is reproduced
 * by a fixed-point iteration on the device (every query re-chooses among the candidates no EARLIER query currently holds,
 * until nothing changes); the rotation histogram and ComputeThreeMaxima (:594-622) run there too.
 * list_cap = the pool of window-list entries is nq * list_cap (48 covers the tracking radii; the reference's lists have
 * no bound).  d_result (int32 x 8, device): [0] matches (the return value), [1] = 1 if the lists did not fit the pool --
 * then nothing was changed and the call is to be repeated with a larger list_cap --, [2] sweeps of the fixed point,
 * [3] window-list entries; map points -> frame: [4] numOutViewAndBad, [5] fail1, [6] fail2 (:353-354).
 * nq + n2 <= 38400.  Enqueued on `stream` (NULL: orbx.h, "Streams"); no host synchronisation -- with two provisos: the packed
 * lists live in the handle's scratch, so ONE call may be in flight per handle (use a handle per thread / per stream), and the
 * first call that needs a larger scratch (nq * list_cap grew) reallocates it, which waits for the device once.  When the lists
 * overflow the pool ([1] = 1) d_frame_mp is left as it was: a chain that goes on to orbba_pose_edges_device then optimises the
 * pose on the matches d_frame_mp already held -- size list_cap so that this cannot happen (48 covers the tracking radii at 2000
 * features; the window of a lost-track search needs more), or read d_result back before trusting the pose. */
int orbm_search_by_projection_frame_device(orbm_t *h, int check_orientation, const uint8_t *d_q_desc, const float *d_q_xy,
                                           const float *d_q_radius, const int32_t *d_q_octave, const float *d_q_angle,
                                           const uint8_t *d_q_ok, int nq, const void *d_kps2, const uint8_t *d_desc2,
                                           const int32_t *d_cell_start, const int32_t *d_cell_items, int grid_cols, int grid_rows,
                                           int n2, int list_cap, int32_t *d_frame_mp, int32_t *d_result, void *stream);
int orbm_search_by_projection_points_device(orbm_t *h, float nn_ratio, const uint8_t *d_q_desc, const float *d_q_xy,
                                            const float *d_q_radius, const int32_t *d_q_level, const uint8_t *d_q_ok, int nq,
                                            const void *d_kps2, const uint8_t *d_desc2, const int32_t *d_cell_start,
                                            const int32_t *d_cell_items, int grid_cols, int grid_rows, int n2, int list_cap,
                                            int32_t *d_frame_mp, int32_t *d_result, void *stream);

/* SearchForInitialization (modules/ORB/ORBMatcher.cpp:33-116) on device-resident records: frame 1's key points / descriptors and
 * frame 2's undistorted record + CSR grid as orbf_frame_post_device leaves it.  d_pre [n1][2] is vecPreMatched, in / out (:112-114);
 * d_matches12 [n1] receives the matches (-1 = none).  The window lists (getFeaturesInArea(pre, windowSize, level1, level1) of the
 * level-0 features), the order-dependent matching with its stealing rule (:63, :75-81), the rotation histogram -- robbed queries
 * stay in it, as in the reference -- and ComputeThreeMaxima all run on the device; the result equals the host entry point's.
 * Limits: n2 + 5 n1 <= 38400 (the claims live in LDS), list_cap entries per query on average (pool of n1 * list_cap).
 * d_result (int32 x 8): [0] matches, [1] = 1 when the lists overflowed the pool (d_matches12 all -1, d_pre untouched: repeat with
 * a larger list_cap or use the host entry point), [1] = 2 when the fixed point had not settled after ORBM_INIT_MAX_SWEEPS sweeps
 * (same guarantee: nothing written; use the host entry point), [2] sweeps of the fixed point, [3] list entries.
 * Cost: ONE workgroup resolves the search; a sweep walks every level-0 feature's window list (1, 4 or 16 lanes share a list,
 * chosen from the mean list length: ORBM_VAR_INIT_LANES) and, per entry, the chain of queries that claim that candidate.  Measured
 * device time of the whole call (profiles/r06_match_latency.txt): 0.11 ms on two extracted views (22 entries per list, 3 sweeps; the
 * host entry point 0.17 ms), 1.2 ms on a crowded scene of near-duplicates (900 x 850 features in 260 x 200 px: 390 entries per
 * list, 5 sweeps; host 0.5 ms -- there the host entry point is the faster one), 5 ms with every feature of both frames in ONE window
 * (2000-entry lists; host 10 ms).  The proof bounds the sweeps by n1 + 1; ORBM_INIT_MAX_SWEEPS keeps a pathological scene from
 * holding a CU for longer than that.
 * Enqueued on `stream` (NULL: orbx.h, "Streams"); one call in flight per handle. */
#define ORBM_INIT_MAX_SWEEPS 64
int orbm_search_for_initialization_device(orbm_t *h, float nn_ratio, int check_orientation, const void *d_kps1, const uint8_t *d_desc1,
                                          int n1, const void *d_kps2, const uint8_t *d_desc2, const int32_t *d_cell_start2,
                                          const int32_t *d_cell_items2, int grid_cols, int grid_rows, int n2, float *d_pre,
                                          int window_size, int list_cap, int32_t *d_matches12, int32_t *d_result, void *stream);

/* The per-point search of the static fuse SearchByProjection(keyFrame, mapPoints, Map*, th) (ORBMatcher.cpp:556-575) on a
 * device-resident key-frame record (key points, descriptors, CSR grid): KeyFrame::getFeaturesInArea(p, radius, predictLevel - 1,
 * predictLevel) with its strict window test, the chi-square gate of :566-567 (d_sigma2 = the level table of square sigmas, indexed
 * by the key point's octave) and the closest descriptor below TH_LOW + 1.  d_best_idx [nq] (-1 = none) / d_best_dist [nq]; what
 * the reference does with a hit (:577-589) mutates its objects and stays with the caller (compat/ORBMatcher.h).
 * d_result (int32 x 8): [0] points with a hit, [1] = 1 if a window held more than list_cap hits (that point's answer is then
 * taken from the first list_cap: repeat with a larger list_cap).  Enqueued on `stream` (NULL: orbx.h, "Streams"). */
int orbm_search_fuse_device(orbm_t *h, const uint8_t *d_q_desc, const float *d_q_xy, const float *d_q_radius, const int32_t *d_q_level,
                            const uint8_t *d_q_ok, int nq, const void *d_kps, const uint8_t *d_desc, const int32_t *d_cell_start,
                            const int32_t *d_cell_items, int grid_cols, int grid_rows, const float *d_sigma2, int list_cap,
                            int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_result, void *stream);

/* MapPoint::computeDescriptor (modules/BasicObject/MapPoint.cpp:103-152) for n_groups map points at once.
 * Group g = the descriptors desc[off[g] .. off[g+1]) of one point's observations (the caller skips bad key frames,
 * :115-120).  best_idx[g] = index inside the group of the descriptor with the least median Hamming distance to the
 * group (median = sorted row[(N-1)/2], self distance 0 included; first index on ties, :138-146); -1 for an empty
 * group (the reference returns without touching the descriptor, :122).  At most 1024 observations per point. */
int orbm_distinctive_descriptors(orbm_t *h, const uint8_t *desc, const int32_t *off, int n_groups, int32_t *best_idx);
int orbm_distinctive_descriptors_device(orbm_t *h, const uint8_t *d_desc, const int32_t *d_off, int n_groups,
                                        int32_t *d_best_idx, void *stream);

/* ORBMatcher::ComputeThreeMaxima (modules/ORB/ORBMatcher.cpp:594-622) on bin sizes */
void orbm_three_maxima(const int32_t *hist_sizes, int n_bins, int *ind1, int *ind2, int *ind3);

#ifdef __cplusplus
}
#endif
#endif
