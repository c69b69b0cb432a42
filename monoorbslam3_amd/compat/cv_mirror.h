// Minimal stand-ins for the few OpenCV types the shims touch, used ONLY when OpenCV is absent
// (this repo's own syntax / smoke checks: -DORBX_SHIM_USE_CV_MIRROR).  In a real integration the
// reference's OpenCV headers are used instead and this file is not included.
// Layouts follow OpenCV 4: cv::KeyPoint is {Point2f pt; float size, angle, response; int octave, class_id} = 28 bytes.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {
    struct Point2f {
        float x = 0, y = 0;
        Point2f() = default;
        Point2f(float x_, float y_) : x(x_), y(y_) {}
    };
    struct KeyPoint {
        Point2f pt;
        float size = 0, angle = -1, response = 0;
        int octave = 0, class_id = -1;
    };
    class Mat {
    public:
        int rows = 0, cols = 0;
        unsigned char *data = nullptr;
        size_t step = 0;
        Mat() = default;
        Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
        Mat(int r, int c, int /*type*/, void *ext, size_t step_ = 0) : rows(r), cols(c), data((unsigned char *) ext), step(step_ ? step_ : (size_t) c) {}
        void create(int r, int c, int /*type*/) {
            store_ = std::make_shared<std::vector<unsigned char>>((size_t) r * c);
            rows = r, cols = c, step = (size_t) c, data = store_->data();
        }
        bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
        int type() const { return CV_8UC1; }
        unsigned char *ptr(int y = 0) { return data + (size_t) y * step; }
        const unsigned char *ptr(int y = 0) const { return data + (size_t) y * step; }
        template<typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t) y * step); }
        template<typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t) y * step); }
        Mat row(int y) const { Mat m; m.rows = 1, m.cols = cols, m.step = step, m.data = data + (size_t) y * step, m.store_ = store_; return m; }
        bool isContinuous() const { return step == (size_t) cols; }
    private:
        std::shared_ptr<std::vector<unsigned char>> store_;
    };
}
