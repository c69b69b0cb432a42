/*
 * orbba.h -- C ABI of the optional local-BA linearisation kernels (liborbx.so).
 *
 * Replaces, for the monocular projection edges only, what g2o does per LM iteration inside
 * Optimize::localBundleAdjustment (modules/Backend/Optimize.cpp:766-951):
 *   EdgeSE3Project3D::computeError / linearizeOplus   modules/Backend/G2oTypes.h:247-251, G2oTypes.cpp:36-47
 *   Pinhole::project / getProjJacobian                 modules/Sensor/Pinhole.cpp:28-32, :49-53
 *   Huber kernel, delta = sqrtf(5.991)                 modules/Backend/Optimize.cpp:857, :880-882
 *   g2o BaseBinaryEdge::constructQuadraticForm         (g2o 20201223, not vendored): H_ii += J_i^T W J_i,
 *                                                      H_ij += J_i^T W J_j, b_i -= J_i^T W e, W = rho'(chi2) * Omega
 * orbba_linearize delivers one iteration's blocks to a host solver; orbba_optimize / orbba_local_bundle_adjustment
 *  keep the whole Levenberg-Marquardt loop on the device: Schur complement over the marginalised
 * points, dense Cholesky of the reduced pose system, back-substitution, VertexSE3 / Vertex3D updates -- the algorithm
 * of g2o 20201223's OptimizationAlgorithmLevenberg + BlockSolver_6_3 (not vendored in the reference; restated from
 * its published sources, checked against the numpy restatement in oracle/ba_ref.py to a stated tolerance).
 * Everything is IEEE double.  Host pointers in and out.
 * Threads (orbx.h, "Streams and threads"): the entry points take no handle and are re-entrant.  The reference runs poseOptimize
 * on the Tracking thread (Tracking.cpp:273-358) while localBundleAdjustment runs on LocalMapping's (LocalMapping.cpp:45-52): every
 * host-pointer call leases a non-blocking stream, a device arena and a page-locked staging block from a per-device pool for its
 * duration -- one copy up, one copy down, one small read-back per LM trial; no stream-0 operation, no allocation in steady state.
 */
#ifndef ORBBA_H
#define ORBBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbba_problem {
    double fx, fy, cx, cy;      /* intrinsics (both camera models) */
    double huber_delta;         /* (double)sqrtf(5.991) in the reference; <= 0 disables the robust kernel */
    int32_t n_poses, n_points, n_edges;
    const double *pose_R;       /* n_poses x 9, row-major R_cw (VertexSE3 estimate, G2oTypes.h:96-116) */
    const double *pose_t;       /* n_poses x 3, t_cw */
    const uint8_t *pose_fixed;  /* n_poses, 1 = fixed key frame (Optimize.cpp:830,:841): no pose blocks */
    const double *points;       /* n_points x 3, world coordinates (Vertex3D, G2oTypes.h:140-161) */
    const int32_t *edge_pose;   /* n_edges */
    const int32_t *edge_point;  /* n_edges, non-decreasing (edges are built map point by map point, Optimize.cpp:860-889) */
    const double *edge_z;       /* n_edges x 2 measured pixel (kp.pt) */
    const double *edge_inv_sigma2; /* n_edges, 1/kp.size^2 (Optimize.cpp:877) */
    /* camera->project / getProjJacobian (G2oTypes.cpp:42): 0 = Pinhole (modules/Sensor/Pinhole.cpp:28-53), 1 = Fisheye, the
     * Kannala-Brandt model of modules/Sensor/Fisheye.cpp:35-49, :83-108 with dist_coeffs k1..k4 as the reference reads them
     * (float values widened to double).  A zero-initialised tail is the pinhole camera. */
    int32_t camera_model;
    double fisheye_k[4];
} orbba_problem;

typedef struct orbba_result {
    double *chi2;     /* n_edges: e^T Omega e (before the robust kernel) */
    double *error;    /* n_edges x 2: z - project(R P + t) */
    double *H_pp;     /* n_poses x 36, row-major 6x6 (rotation block first, as VertexSE3::oplusImpl) */
    double *b_p;      /* n_poses x 6 */
    double *H_ll;     /* n_points x 9 */
    double *b_l;      /* n_points x 3 */
    double *H_lp;     /* n_edges x 18, row-major 3x6 = J_point^T W J_pose (zero for fixed poses) */
    float kernel_ms;  /* device time of the two kernels (HIP events) */
} orbba_result;

/* any output pointer may be NULL */
int orbba_linearize(const orbba_problem *p, orbba_result *r, int device);

/* g2o::OptimizationAlgorithmLevenberg's knobs; a zero field takes g2o's default */
typedef struct orbba_lm_options {
    int32_t max_iterations;     /* optimizer.optimize(n) */
    int32_t max_trials;         /* _maxTrialsAfterFailure, default 10 */
    double tau;                 /* computeLambdaInit: lambda0 = tau * max |H_jj|, default 1e-5 */
    double good_step_lower;     /* default 1/3 */
    double good_step_upper;     /* default 2/3 */
    double user_lambda_init;    /* > 0: fixed initial lambda */
    const uint8_t *edge_active; /* n_edges, 0 = edge at level 1 (Optimize.cpp:900-902); NULL = all active */
} orbba_lm_options;

typedef struct orbba_lm_result {
    double *pose_R;  /* n_poses x 9, optimised (fixed poses unchanged) */
    double *pose_t;  /* n_poses x 3 */
    double *points;  /* n_points x 3 */
    double *chi2;    /* n_edges: e^T Omega e at the final estimate, also for edges that options.edge_active switches off.
                      * orbba_local_bundle_adjustment overwrites the entries of the edges it demoted after its first
                      * round with their first-round value: g2o never recomputes the error of a level-1 edge, so that is
                      * what e->chi2() returns at Optimize.cpp:919 */
    int32_t iterations, trials; /* outer LM iterations run, linear solves tried */
    double lambda;              /* final damping */
    double chi2_initial, chi2_final; /* activeRobustChi2 before / after */
    float device_ms;            /* HIP-event time of the whole loop (includes the small host decisions) */
} orbba_lm_result;

/* optimizer.initializeOptimization(); optimizer.optimize(max_iterations) for the graph of
 * Optimize::localBundleAdjustment (Optimize.cpp:811-893): VertexSE3 poses (fixed ones constant), marginalised
 * Vertex3D points, EdgeSE3Project3D edges with the Huber kernel of p->huber_delta (<= 0: none). */
int orbba_optimize(const orbba_problem *p, const orbba_lm_options *o, orbba_lm_result *r, int device);

/* Optimize.cpp:892-922: optimize(5) with Huber; edges with chi2 > 5.991 go to level 1 and the kernel is dropped;
 * optimize(10); outlier[e] = 1 for every edge demoted after the first round (its stale e->chi2() stays above 5.991, see
 * orbba_lm_result.chi2) and for every still-active edge whose chi2 at the final estimate exceeds 5.991 -- the
 * observations the reference then erases (:917-935).  outlier may be NULL. */
int orbba_local_bundle_adjustment(const orbba_problem *p, orbba_lm_result *r, uint8_t *outlier, int device);

/* Optimize::poseOptimize (modules/Backend/Optimize.cpp:444-545) for a batch of frames at once: per frame one
 * VertexSE3 and one EdgeSE3Project3DOnlyPose (G2oTypes.h:209-236, G2oTypes.cpp:27-34) per matched map point; four rounds
 * of optimize(10), each restarted from the frame's initial pose on the edges the previous round left as inliers
 * (chi2 <= 5.991), Huber kernel throughout (the `iter == 2` test at :518 never fires).  Frames with fewer than three
 * correspondences come back unchanged with n_inliers = 0 (:491). */
typedef struct orbba_pose_problem {
    double fx, fy, cx, cy;
    double huber_delta;             /* (double)sqrtf(5.991), :466 */
    int32_t n_frames;
    int32_t rounds, iterations;     /* 0 = the reference's 4 and 10 (:495) */
    const int32_t *edge_off;        /* n_frames + 1: frame f owns edges [edge_off[f], edge_off[f+1]) */
    const double *pose_R, *pose_t;  /* n_frames x 9 / x 3: frame->T_cw */
    const double *points;           /* n_edges x 3: mp->getPos() */
    const double *edge_z;           /* n_edges x 2: kp.pt */
    const double *edge_inv_sigma2;  /* n_edges: 1 / kp.size^2 (:478) */
    int32_t camera_model;           /* as in orbba_problem */
    double fisheye_k[4];
} orbba_pose_problem;

typedef struct orbba_pose_result {
    double *pose_R, *pose_t;  /* n_frames x 9 / x 3 */
    uint8_t *inlier;          /* n_edges: 0 = the map point the reference drops from the frame (:531-537) */
    int32_t *n_inliers;       /* n_frames: the return value of poseOptimize */
    double *chi2;             /* n_edges at the final pose, may be NULL */
    float kernel_ms;
} orbba_pose_result;

int orbba_pose_optimize_batch(const orbba_pose_problem *p, orbba_pose_result *r, int device);

/* The same with every array of both structs in DEVICE memory (chi2 must be given), enqueued on `stream` (hipStream_t; NULL: orbx.h, "Streams" --
 * no handle here: stream 0 itself) without a copy or a wait: the last step of the tracking chain extract -> frame record -> SearchByProjection -> pose
 * (Tracking.cpp:289-336) when the steps before it left their results on the device. */
int orbba_pose_optimize_batch_device(const orbba_pose_problem *p, orbba_pose_result *r, void *stream);
/* poseOptimize's edges (Optimize.cpp:468-490) from a device-resident frame: for every key point i, in index order, whose
 * d_frame_mp[i] (as orbm_search_by_projection_*_device leaves it) is a query index q in [0, nq): the map point position
 * d_q_points[3q..] (float, mp->getPos()), the measurement kp.pt of d_kps[i] (orbx_kp records, undistorted) and
 * invSigma2 = 1.f / kp.size / kp.size (:479).  d_edge_off receives {0, n_edges} (one frame); d_points / d_edge_z /
 * d_edge_inv_sigma2 need room for n2 edges; d_edge_kp (may be NULL) receives i per edge -- the vecIndices of :464 that the
 * caller uses to drop the outliers from the frame (:531-537).  Enqueued on `stream` (NULL: orbx.h, "Streams").
 * d_frame_mp must hold indices of ONE query set: when two searches filled it (Tracking.cpp:289-336 runs frame -> frame and then
 * map points -> frame on the same frame_mp), give both searches one shared index space -- concatenated query arrays with
 * q_ok masks selecting each search's part -- and pass the concatenated d_q_points here. */
int orbba_pose_edges_device(int n2, int nq, const int32_t *d_frame_mp, const void *d_kps, const float *d_q_points,
                            int32_t *d_edge_off, double *d_points, double *d_edge_z, double *d_edge_inv_sigma2,
                            int32_t *d_edge_kp, void *stream);

/* Kernel-choice switches (parity twins; no reference counterpart).  The BA entry points take no handle, so a switch holds
 * for the process and is read per call.  Unknown switch / value out of range: ORBX_E_ARG. */
#define ORBBA_VAR_CHOL 0      /* reduced pose system: 0 solved in LDS when it fits (default), 1 the global-memory kernel */
#define ORBBA_VAR_POSE_LDS 1  /* poseOptimize: edges of a frame staged in LDS, 0 .. 3000 (default 3000; 0 = never) */
int orbba_set_variant(int which, int value);

#ifdef __cplusplus
}
#endif
#endif
