/*
 * orbf.h -- C ABI of the per-frame post-processing that follows ORB extraction (liborbx.so).
 *
 * Replaces the host loops of the reference's Frame constructor, modules/BasicObject/Frame.cpp:24-51:
 *   :24-26  kp.size *= camera->uncertainty(kp.pt)          (Pinhole.cpp:55-57 = 1, Fisheye.cpp:110-112 = scale_mat)
 *   :28     camera->undistortKeyPoints(raw, undistorted)    (Pinhole.cpp:59-83 -> cv::undistortPoints, Fisheye: copy)
 *   :32-51  assignment of the undistorted key points to the 40-px grid (PosInGrid :89-94)
 * so that a batch of frames leaves the GPU as complete "frame records" (raw + undistorted key points, descriptors,
 * grid index) and the window searches of orbm.h can consume the grid without a host pass.
 * SURVEY.md 8f rank 2 (+ the grid half of rank 1).
 *
 * Returns 0 or a negative ORBX_E_* code (orbx.h); text in orbx_last_error().
 * A handle owns device scratch that every call uses: keep one call in flight per handle (calls enqueued on ONE stream
 * are fine, they are ordered); use one handle per stream / host thread otherwise.
 */
#ifndef ORBF_H
#define ORBF_H

#include <stddef.h>
#include <stdint.h>

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBF_GRID_SIZE 40 /* modules/BasicObject/Frame.h:18 */
#define ORBF_MAX_DIST 12

/* Camera::Camera (modules/Sensor/Camera.cpp:17-22) reduced to what Frame.cpp:24-51 reads */
typedef struct orbf_camera {
    int32_t width, height;    /* image size: grid dimensions (Frame.cpp:32-40) and the PosInGrid bounds */
    float fx, fy, cx, cy;     /* mat_K */
    int32_t n_dist;           /* 0..12 coefficients, OpenCV order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 */
    float dist[ORBF_MAX_DIST];
    int32_t undistort;        /* 1 = Pinhole/RAD_TAN (undistorts unless dist[0] == 0, Pinhole.cpp:62); 0 = Fisheye (copy) */
    const float *size_scale;  /* NULL (Pinhole), or host height*width f32 map = Fisheye::scale_mat; copied at create */
} orbf_camera;

typedef struct orbf_ctx orbf_t;

int orbf_create(const orbf_camera *cam, int device, orbf_t **out);
void orbf_destroy(orbf_t *h);
/* GRID_COLS / GRID_ROWS of Frame.cpp:32-40; n_cells = cols * rows, cell id = cx * rows + cy (grid[cx][cy]) */
int orbf_grid_dims(const orbf_t *h, int *cols, int *rows);

/* Device pointers, enqueued on `stream` (hipStream_t; NULL: orbx.h, "Streams").
 *   d_kp_raw  [n_frames][cap]  in/out: orbx_extract_batch_device's records; `size` is scaled in place (:24-26)
 *   d_n       [n_frames]       key-point counts
 *   d_kp_un   [n_frames][cap]  out: copy of raw with pt undistorted (:28)
 *   d_cell_start [n_frames][n_cells + 1], d_cell_items [n_frames][cap]  out: CSR grid, items of a cell in
 *             ascending key-point index (the push_back order of :45-50); key points outside the image are in no cell */
int orbf_frame_post_device(orbf_t *h, int n_frames, orbx_kp *d_kp_raw, const int32_t *d_n, int cap, orbx_kp *d_kp_un,
                           int32_t *d_cell_start, int32_t *d_cell_items, void *stream);
/* one frame, host pointers */
int orbf_frame_post(orbf_t *h, orbx_kp *kp_raw, int n, orbx_kp *kp_un, int32_t *cell_start, int32_t *cell_items);

#ifdef __cplusplus
}
#endif
#endif
