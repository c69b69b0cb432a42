// The three loops of the reference's Frame constructor that follow the extractor call (modules/BasicObject/Frame.cpp:24-51)
// as one call on the GPU (include/orbf.h): kp.size *= camera->uncertainty(kp.pt); camera->undistortKeyPoints(raw, un);
// grid[x][y].push_back(i).  In Frame::Frame a maintainer replaces lines 24-51 by
//     static FramePost post(Camera::getCamera() ...);  post(raw_key_points, key_points, grid, GRID_COLS, GRID_ROWS);
#pragma once
#include <cstring>
#include <vector>

#include "orbf.h"

#if defined(ORBX_SHIM_USE_CV_MIRROR)
#include "cv_mirror.h"
#else
#include <opencv2/core.hpp>
#endif

namespace mono_orb_slam3 {
    class FramePost {
    public:
        // width/height, mat_K and dist_coeffs of Camera (Camera.cpp:17-22); pinhole = RAD_TAN model (Pinhole.cpp:59-83),
        // otherwise Fisheye: key points are copied and `size_scale` (Fisheye::scale_mat, height x width floats) scales kp.size
        FramePost(int width, int height, float fx, float fy, float cx, float cy, const std::vector<float> &dist_coeffs,
                  bool pinhole = true, const float *size_scale = nullptr) {
            orbf_camera cam;
            std::memset(&cam, 0, sizeof cam);
            cam.width = width, cam.height = height;
            cam.fx = fx, cam.fy = fy, cam.cx = cx, cam.cy = cy;
            cam.n_dist = (int32_t) (dist_coeffs.size() < ORBF_MAX_DIST ? dist_coeffs.size() : ORBF_MAX_DIST);
            for (int i = 0; i < cam.n_dist; ++i) cam.dist[i] = dist_coeffs[(size_t) i];
            cam.undistort = pinhole ? 1 : 0;
            cam.size_scale = size_scale;
            ok_ = orbf_create(&cam, /*device*/0, &h_) == ORBX_OK;
            if (ok_) orbf_grid_dims(h_, &cols_, &rows_);
        }
        FramePost(const FramePost &) = delete;
        FramePost &operator=(const FramePost &) = delete;
        ~FramePost() { orbf_destroy(h_); }

        bool ok() const { return ok_; }
        int gridCols() const { return cols_; } // Frame::GRID_COLS (Frame.cpp:32-40)
        int gridRows() const { return rows_; }

        // raw: in/out (size scaled), un: out (= Frame::key_points), grid: out (= Frame::grid, [GRID_COLS][GRID_ROWS])
        bool operator()(std::vector<cv::KeyPoint> &raw, std::vector<cv::KeyPoint> &un,
                        std::vector<std::vector<std::vector<size_t>>> &grid) const {
            static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_kp), "cv::KeyPoint layout");
            const int n = (int) raw.size();
            un.resize(raw.size());
            grid.assign((size_t) cols_, std::vector<std::vector<size_t>>((size_t) rows_));
            if (!ok_) return false;
            std::vector<int32_t> start((size_t) cols_ * rows_ + 1), items((size_t) (n > 0 ? n : 1));
            if (orbf_frame_post(h_, reinterpret_cast<orbx_kp *>(raw.data()), n, reinterpret_cast<orbx_kp *>(un.data()),
                                start.data(), items.data()) != ORBX_OK)
                return false;
            for (int cx = 0; cx < cols_; ++cx)
                for (int cy = 0; cy < rows_; ++cy) {
                    const int c = cx * rows_ + cy;
                    grid[(size_t) cx][(size_t) cy].assign(items.begin() + start[(size_t) c], items.begin() + start[(size_t) c + 1]);
                }
            return true;
        }

    private:
        orbf_t *h_ = nullptr;
        bool ok_ = false;
        int cols_ = 0, rows_ = 0;
    };
}
