// Minimal stand-ins for the reference's Frame / KeyFrame / MapPoint / DBoW2::FeatureVector, exposing only
// the members ORBMatcher touches (modules/BasicObject/{Frame,KeyFrame,MapPoint}.h,
// thirdParty/DBoW2/DBoW2/FeatureVector.h).  Used ONLY for this repo's syntax / smoke checks
// (-DORBX_SHIM_USE_REF_MIRROR); a real integration includes the reference's own headers.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "cv_mirror.h"

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
    class MapPoint {
    public:
        bool bad = false;
        bool isBad() const { return bad; }
    };

    class Frame {
    public:
        int num_kps = 0;
        int img_cols = 0, img_rows = 0;
        std::vector<cv::KeyPoint> key_points;
        cv::Mat descriptors;
        DBoW2::FeatureVector feature_vector;
        std::vector<std::shared_ptr<MapPoint>> map_points;
    };

    class KeyFrame : public Frame {
    public:
        std::vector<std::shared_ptr<MapPoint>> getMapPoints() const { return map_points; }
        bool hasMapPoint(int idx) const { return map_points[idx] != nullptr; }
    };
}
