/*
 * orb_ref -- CPU restatement of the reference ORB front-end hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU
 * baseline.  Nothing under monoorbslam3_amd/ may include, link or call it;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference (Whitby-Li/monoORBSLAM3) ships no golden
 * vectors, fixtures or asserting tests for this path,
 * and its pixel arithmetic lives in OpenCV 4.x, which is absent here, so the
 * reference cannot be compiled in this container (oracle/README.md).  The
 * restatement follows the reference's own sources line by line where the
 * arithmetic is in the repo, and OpenCV 4.2's documented/recalled semantics
 * (cv::resize INTER_LINEAR, cv::FAST 9/16, cv::GaussianBlur 8U fixed point,
 * cv::fastAtan2) where it is not.  It is pinned by known-answer tests that
 * are derivable from the reference code alone (tests/test_oracle_kat.py).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference root).
 */
#ifndef ORB_REF_H
#define ORB_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBREF_MAX_LEVELS 16
#define ORBREF_EDGE 19          /* EDGE_THRESHOLD, modules/ORB/ORBExtractor.cpp:15 */
#define ORBREF_HALF_PATCH 15    /* HALF_PATCH_SIZE, modules/ORB/ORBExtractor.cpp:14 */
#define ORBREF_CELL 30          /* W, modules/ORB/ORBExtractor.cpp:575 */

/* cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct orbref_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbref_kp;

typedef struct orbref_cfg {
    int n_features, n_levels, ini_th_fast, min_th_fast;
    float scale_factor, log_scale_factor;
    float scale_factors[ORBREF_MAX_LEVELS];
    float inv_scale_factors[ORBREF_MAX_LEVELS];
    float square_sigmas[ORBREF_MAX_LEVELS];
    float inv_square_sigmas[ORBREF_MAX_LEVELS];
    int n_features_per_level[ORBREF_MAX_LEVELS];
    int u_max[ORBREF_HALF_PATCH + 1];
    int blur_taps[7];           /* 8.8 fixed-point Gaussian taps (Appendix B.3) */
} orbref_cfg;

/* candidate handed to the quadtree: border-relative integer coords + FAST score */
typedef struct orbref_cand {
    float x, y, response;
} orbref_cand;

/* ---- scalar helpers (OpenCV cvRound/cvFloor/cvCeil semantics) ---- */
int orbref_round_f(float v);
int orbref_round_d(double v);
int orbref_floor_f(float v);
int orbref_ceil_f(float v);

/* ---- extractor ---- */
void orbref_cfg_init(orbref_cfg *cfg, int n_features, float scale_factor, int n_levels,
                     int ini_th_fast, int min_th_fast);
void orbref_cfg_requota(orbref_cfg *cfg, int n_features);
void orbref_set_blur_taps(orbref_cfg *cfg, int variant); /* 0: sum-256 set, 1: plain-rounded sum-257 set */
void orbref_level_size(const orbref_cfg *cfg, int w0, int h0, int level, int *w, int *h);
const int8_t *orbref_pattern(void); /* 512 (x,y) int8 pairs */

void orbref_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride);
/* pyramid levels are written tightly packed (stride = level width) into
 * caller-provided buffers levels[l] */
void orbref_pyramid(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                    uint8_t **levels);

/* cv::FAST(sub-image, thr, nms=true) restricted to the pixel box [x0,x1) x [y0,y1)
 * of img; returns keypoints (absolute pixel coords) row-major. */
int orbref_fast_box(const uint8_t *img, int stride, int x0, int y0, int x1, int y1,
                    int threshold, orbref_cand *out, int cap);
/* threshold-free corner strength: max(max_arc min(v-p), max_arc min(p-v)) - 1 */
int orbref_fast_strength(const uint8_t *img, int stride, int x, int y);

/* per-level candidates in reference order (cell row-major, then row-major in
 * cell), coordinates relative to (19,19). */
int orbref_level_candidates(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                            orbref_cand *out, int cap);
int orbref_distribute_octree(const orbref_cand *cands, int n, int minX, int maxX, int minY,
                             int maxY, int n_features, orbref_cand *out, int cap);
float orbref_fast_atan2(float y, float x);
float orbref_ic_angle(const orbref_cfg *cfg, const uint8_t *img, int stride, int x, int y);
void orbref_gaussian_blur7(const orbref_cfg *cfg, const uint8_t *src, int w, int h, int sstride,
                           uint8_t *dst, int dstride);
void orbref_sincos_deg(float angle_deg, float *c, float *s);
void orbref_sincosf(float y, float *sn, float *cs);
void orbref_sincos_deg_n(const float *angle_deg, int n, float *out);
long orbref_sincosf_check_libm(uint32_t first, uint32_t last, uint32_t step);
void orbref_brief(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint8_t *desc32);

/* Full operator(): returns number of keypoints (0 -> outputs untouched), or -1
 * if cap is too small.  per_level_counts may be NULL. */
int orbref_extract(const orbref_cfg *cfg, const uint8_t *img, int w, int h, int stride,
                   orbref_kp *kps, uint8_t *desc, int cap, int *per_level_counts);

/* ---- matcher ---- */
int orbref_hamming(const uint8_t *a, const uint8_t *b);
void orbref_three_maxima(const int *hist_sizes, int n_bins, int *ind1, int *ind2, int *ind3);

/* DBoW2 FeatureVector as CSR: node_ids ascending, offsets[n_nodes+1], indices */
typedef struct orbref_fv {
    int n_nodes;
    const uint32_t *node_ids;
    const int32_t *offsets;
    const uint32_t *indices;
} orbref_fv;

/* SearchByBow (modules/ORB/ORBMatcher.cpp:118-201).  kf_mp_ok[i] != 0 <=> KF
 * feature i has a live MapPoint.  frame_mp[j]: -1 = null, else the KF feature
 * index whose MapPoint was assigned (in/out). */
int orbref_search_by_bow(float nn_ratio, int check_orientation,
                         const uint8_t *desc1, const float *angle1, const uint8_t *kf_mp_ok, int n1,
                         const orbref_fv *fv1,
                         const uint8_t *desc2, const float *angle2, int32_t *frame_mp, int n2,
                         const orbref_fv *fv2);

/* SearchForTriangulation (modules/ORB/ORBMatcher.cpp:417-522) */
int orbref_search_for_triangulation(int check_orientation,
                                    const uint8_t *desc1, const float *angle1, const uint8_t *has_mp1, int n1,
                                    const orbref_fv *fv1,
                                    const uint8_t *desc2, const float *angle2, const uint8_t *has_mp2, int n2,
                                    const orbref_fv *fv2, int32_t *matches12);

/* Frame grid (modules/BasicObject/Frame.cpp:33-51, 90-127) over plain arrays */
typedef struct orbref_grid {
    int cols, rows, img_w, img_h;
    int32_t *cell_start; /* cols*rows+1, column-major cells: cell = cx*rows+cy */
    int32_t *cell_items;
} orbref_grid;
orbref_grid *orbref_grid_build(const orbref_kp *kps, int n, int img_w, int img_h);
void orbref_grid_free(orbref_grid *g);
int orbref_features_in_area(const orbref_grid *g, const orbref_kp *kps, float x, float y, float r,
                            int min_level, int max_level, int32_t *out, int cap);

/* ---- Frame post-processing: modules/BasicObject/Frame.cpp:24-28 ---------------------------------------------------
 * kp.size *= camera->uncertainty(kp.pt) (Pinhole.cpp:55-57: 1; Fisheye.cpp:110-112: scale_mat(y,x) with the float
 * coordinates truncated by the implicit int conversion), then camera->undistortKeyPoints (Pinhole.cpp:59-83:
 * copy when dist[0] == 0, else cv::undistortPoints(pts, K, dist, R = I, P = K); Fisheye.cpp:114-117: copy).
 * cv::undistortPoints is OpenCV's (calib3d/undistort.dispatch.cpp, 4.2): double arithmetic, FIVE fixed-point
 * iterations (TermCriteria(MAX_ITER, 5, 0.01) -- only the count is used), early exit when icdist < 0, result
 * cast to float.  Restated from the published algorithm; unpinned against OpenCV like the rest of this file. */
typedef struct orbref_camera {
    int32_t width, height;
    float fx, fy, cx, cy;
    int32_t n_dist;    /* 0..12 coefficients in OpenCV order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 */
    float dist[12];
    int32_t undistort; /* 1 = Pinhole (RAD_TAN), 0 = Fisheye (key points are copied) */
    const float *size_scale; /* NULL, or height*width floats (Fisheye::scale_mat) */
} orbref_camera;
void orbref_undistort_point(const orbref_camera *cam, float u, float v, float *xu, float *yu);
/* raw (in/out: size scaled), un (out: copy of raw with pt undistorted) */
void orbref_frame_post(const orbref_camera *cam, orbref_kp *raw, int n, orbref_kp *un);

/* ---- DBoW2 vocabulary transform: thirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259 (called with levelsup = 4
 * from Frame::computeBow, modules/BasicObject/Frame.cpp:168-178), BowVector.cpp:32-90, FeatureVector.cpp:31-45,
 * FORB.cpp:81-101, ScoringObject.h:73-92.  Nodes are given in the order loadFromTextFile (:1338-1420) creates them:
 * node 0 = root, node i = line i of the text file; children of a node in ascending id (push_back order, :1393);
 * word ids count the lines flagged as leaves, in file order (:1408-1414). */
typedef struct orbref_voc orbref_voc;
orbref_voc *orbref_voc_build(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                             const uint8_t *is_leaf, const uint8_t *desc, const double *weight);
void orbref_voc_free(orbref_voc *v);
int orbref_voc_n_words(const orbref_voc *v);
/* transform(feature, word_id, weight, &nid, levelsup) (:1218-1259).  *nid is left untouched when the descent ends
 * above level L - levelsup (the reference leaves it uninitialised there). */
void orbref_voc_transform_feature(const orbref_voc *v, const uint8_t *desc, int levelsup, uint32_t *word, double *weight,
                                  uint32_t *nid);
/* transform(features, BowVector, FeatureVector, levelsup) (:1127-1201).  Outputs are the two std::maps flattened in
 * key order: bow_ids/bow_vals [n_words_out]; fv_nodes [n_fv], fv_off [n_fv + 1], fv_idx [fv_off[n_fv]].
 * All arrays need room for n entries (fv_off n + 1). */
void orbref_voc_transform(const orbref_voc *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_ids,
                          double *bow_vals, int *n_words_out, uint32_t *fv_nodes, int32_t *fv_off, uint32_t *fv_idx,
                          int *n_fv);

/* MapPoint::computeDescriptor (modules/BasicObject/MapPoint.cpp:103-152): index of the descriptor with the least
 * median distance to the n descriptors (sorted row[(n-1)/2]); -1 when n == 0 */
int orbref_distinctive_descriptor(const uint8_t *desc, int n);

/* The best / second-best loop of SearchByBow (modules/ORB/ORBMatcher.cpp:148-162) over all rows of b for every
 * row of a: strict '<' in ascending candidate order, both distances start at 256, best index -1 without candidates. */
void orbref_best2(const uint8_t *a, int na, const uint8_t *b, int nb, int32_t *best_idx, uint16_t *best,
                  uint16_t *second);

/* SearchForInitialization (modules/ORB/ORBMatcher.cpp:33-116) */
int orbref_search_for_initialization(float nn_ratio, int check_orientation,
                                     const orbref_kp *kps1, const uint8_t *desc1, int n1,
                                     const orbref_kp *kps2, const uint8_t *desc2, int n2,
                                     int img_w, int img_h,
                                     float *prematched_xy /* n1*2 in/out */, int32_t *matches12,
                                     int window_size);

/* SearchByProjection(lastFrame|lastKF, curFrame, th) (modules/ORB/ORBMatcher.cpp:203-348) with the camera maths
 * done by the caller; see include/orbm.h for the argument meaning. */
int orbref_search_by_projection_frame(int check_orientation, const uint8_t *q_desc, const float *q_xy,
                                      const float *q_radius, const int32_t *q_octave, const float *q_angle,
                                      const uint8_t *q_ok, int nq, const orbref_kp *kps2, const uint8_t *desc2, int n2,
                                      int img_w, int img_h, int32_t *frame_mp);
/* SearchByProjection(frame, mapPoints, th) (modules/ORB/ORBMatcher.cpp:350-415) */
int orbref_search_by_projection_points(float nn_ratio, const uint8_t *q_desc, const float *q_xy, const float *q_radius,
                                       const int32_t *q_level, const uint8_t *q_ok, int nq, const orbref_kp *kps2,
                                       const uint8_t *desc2, int n2, int img_w, int img_h, int32_t *frame_mp,
                                       int32_t *counters);

/* KeyFrame::getFeaturesInArea (modules/BasicObject/KeyFrame.cpp:181-211): strict window test */
int orbref_keyframe_features_in_area(const orbref_grid *g, const orbref_kp *kps, float x, float y, float r,
                                     int min_level, int max_level, int32_t *out, int cap);
/* static SearchByProjection(keyFrame, mapPoints, Map*, th) (modules/ORB/ORBMatcher.cpp:524-592), the part that reads no
 * MapPoint / KeyFrame state: per map point the closest key point of the window; see include/orbm.h orbm_search_fuse.
 * Returns the number of points with a candidate. */
int orbref_search_fuse(const uint8_t *q_desc, const float *q_xy, const float *q_radius, const int32_t *q_level,
                       const uint8_t *q_ok, int nq, const orbref_kp *kps, const uint8_t *desc, int n, int img_w,
                       int img_h, const float *sigma2, int32_t *best_idx, int32_t *best_dist);

#ifdef __cplusplus
}
#endif
#endif
