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
 * The Schur complement, the reduced solve and the LM loop stay with g2o.
 * Everything is IEEE double.  Host pointers in and out; the call uploads, runs two kernels
 * (per-edge linearise, fixed-order block reduction) and downloads.
 */
#ifndef ORBBA_H
#define ORBBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbba_problem {
    double fx, fy, cx, cy;      /* Pinhole intrinsics */
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

#ifdef __cplusplus
}
#endif
#endif
