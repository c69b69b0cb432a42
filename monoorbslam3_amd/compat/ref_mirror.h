// Minimal stand-ins for the reference's Frame / KeyFrame / MapPoint / Map / Camera / Pose / Logger and
// DBoW2::FeatureVector, exposing only the members the shims touch (modules/BasicObject/{Frame,KeyFrame,MapPoint,
// Pose,Map}.h, modules/Sensor/Camera.h, modules/Log/Logger.h, thirdParty/DBoW2/DBoW2/FeatureVector.h) with the
// reference's signatures, plus the three Eigen operations ORBMatcher.cpp uses on them.  Used ONLY for this repo's
// syntax / behaviour checks (-DORBX_SHIM_USE_REF_MIRROR): they let every member function Tracking.cpp and
// LocalMapping.cpp call compile and run here without OpenCV / Eigen; a real integration includes the reference's own
// headers and none of this.
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <ostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "cv_mirror.h"

namespace Eigen { // just enough of Vector3f / Matrix3f for `R * Pw + t`, `Pw - Ow`, `.norm()`, `.dot()`, `[i]`
    struct Vector3f {
        float v[3] = {0, 0, 0};
        Vector3f() = default;
        Vector3f(float x, float y, float z) { v[0] = x, v[1] = y, v[2] = z; }
        float operator[](int i) const { return v[i]; }
        float &operator[](int i) { return v[i]; }
        Vector3f operator-(const Vector3f &o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
        Vector3f operator+(const Vector3f &o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
        float dot(const Vector3f &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
        float norm() const { return std::sqrt(dot(*this)); }
    };
    struct Matrix3f {
        float m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        Vector3f operator*(const Vector3f &p) const {
            return {m[0][0] * p[0] + m[0][1] * p[1] + m[0][2] * p[2], m[1][0] * p[0] + m[1][1] * p[1] + m[1][2] * p[2],
                    m[2][0] * p[0] + m[2][1] * p[1] + m[2][2] * p[2]};
        }
    };
}

namespace DBoW2 {
    typedef unsigned int NodeId;
    typedef unsigned int WordId;
    typedef double WordValue;
    class BowVector : public std::map<WordId, WordValue> {};
    class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {
    public:
        void addFeature(NodeId id, unsigned int i_feature) { (*this)[id].push_back(i_feature); }
    };
}

namespace mono_orb_slam3 {
    // modules/BasicObject/Pose.h:11-32
    struct Pose {
        Eigen::Matrix3f R;
        Eigen::Vector3f t;
        [[nodiscard]] Eigen::Vector3f map(const Eigen::Vector3f &P) const { return R * P + t; }
    };

    // modules/Sensor/Camera.h (pinhole stand-in: project + isInImage are what ORBMatcher.cpp calls)
    class Camera {
    public:
        int width = 0, height = 0;
        float fx = 1, fy = 1, cx = 0, cy = 0;
        static const Camera *getCamera() { return instance(); }
        static Camera *instance() { static Camera cam; return &cam; }
        cv::Point2f project(const Eigen::Vector3f &Pc) const { return {fx * Pc[0] / Pc[2] + cx, fy * Pc[1] / Pc[2] + cy}; }
        bool isInImage(const cv::Point2f &p) const { return p.x >= 0 && p.x < (float) width && p.y >= 0 && p.y < (float) height; }
    };

    // modules/Log/Logger.h:56,62
    struct Logger {
        std::ostringstream text;
        template<typename T> Logger &operator<<(const T &v) { text << v; return *this; }
    };
    inline Logger tracker_logger;
    inline const std::string titles[3] = {"[a] ", "[b] ", "[c] "};

    class KeyFrame;
    class Map {
    public:
        int erased = 0;
    };

    // modules/BasicObject/MapPoint.h: the members read or called by ORBMatcher.cpp
    class MapPoint : public std::enable_shared_from_this<MapPoint> {
    public:
        bool bad = false;
        Eigen::Vector3f pos, normal = {0, 0, 1};
        float min_distance = 0, max_distance = 1e30f;
        int predict_level = 0;
        cv::Mat descriptor;
        std::map<std::shared_ptr<KeyFrame>, size_t> observations;
        std::shared_ptr<MapPoint> replaced_by;
        // Tracking fields (MapPoint.h, "Tracking" block)
        bool track_in_view = false;
        int track_scale_level = 0;
        float track_view_cos = 1.f, track_proj_x = -1, track_proj_y = -1;

        bool isBad() const { return bad; }
        Eigen::Vector3f getPos() const { return pos; }
        Eigen::Vector3f getAverageDirection() const { return normal; }
        float getMinDistanceInvariance() const { return min_distance; }
        float getMaxDistanceInvariance() const { return max_distance; }
        int predictScaleLevel(const float &) const { return predict_level; }
        cv::Mat getDescriptor() const { return descriptor; }
        bool isObserveKeyFrame(const std::shared_ptr<KeyFrame> &kf) const { return observations.count(kf) != 0; }
        int getNumObs() const { return (int) observations.size(); }
        void addObservation(const std::shared_ptr<KeyFrame> &kf, size_t idx) { observations[kf] = idx; }
        inline void replace(const std::shared_ptr<MapPoint> &mapPoint); // after KeyFrame
    };

    class Frame {
    public:
        int num_kps = 0;
        int img_cols = 0, img_rows = 0;
        std::vector<cv::KeyPoint> key_points;
        cv::Mat descriptors;
        DBoW2::FeatureVector feature_vector;
        std::vector<std::shared_ptr<MapPoint>> map_points;
        Pose T_cw;
    };

    class KeyFrame : public Frame, public std::enable_shared_from_this<KeyFrame> {
    public:
        std::vector<std::shared_ptr<MapPoint>> getMapPoints() const { return map_points; }
        bool hasMapPoint(int idx) const { return map_points[idx] != nullptr; }
        std::shared_ptr<MapPoint> getMapPoint(size_t idx) const { return map_points[idx]; }
        void addMapPoint(const std::shared_ptr<MapPoint> &mp, size_t idx) { map_points[idx] = mp; }
        void eraseMapPoint(size_t idx) { map_points[idx] = nullptr; }
        Pose getPose() const { return T_cw; }
        Eigen::Vector3f getCameraCenter() const { return center; }
        Eigen::Vector3f center;
    };

    // MapPoint.cpp:233-264 reduced to the bookkeeping the fuse's result depends on
    inline void MapPoint::replace(const std::shared_ptr<MapPoint> &mapPoint) {
        if (mapPoint.get() == this) return;
        auto obs = observations;
        observations.clear();
        bad = true;
        replaced_by = mapPoint;
        for (const auto &o : obs) {
            if (!mapPoint->isObserveKeyFrame(o.first)) {
                o.first->addMapPoint(mapPoint, o.second);
                mapPoint->addObservation(o.first, o.second);
            } else o.first->eraseMapPoint(o.second);
        }
    }
}
