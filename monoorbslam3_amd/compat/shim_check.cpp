// Syntax check of the reference-signature shims against the mirror types (no OpenCV / reference headers here).
#define ORBX_SHIM_USE_CV_MIRROR
#define ORBX_SHIM_USE_REF_MIRROR
#include "ORBExtractor.h"
#include "ORBMatcher.h"
#include "ORBVocabulary.h"
#include "FramePost.h"

int shim_check_instantiate() {
    using namespace mono_orb_slam3;
    // signatures exactly as the reference's call sites use them (Frame.cpp:20, Tracking.cpp:24,262,606, LocalMapping.cpp:168)
    void (ORBExtractor::*op)(const cv::Mat &, std::vector<cv::KeyPoint> &, cv::Mat &) = &ORBExtractor::operator();
    int (ORBMatcher::*bow)(const std::shared_ptr<KeyFrame> &, const std::shared_ptr<Frame> &) const = &ORBMatcher::SearchByBow;
    int (ORBMatcher::*tri)(const std::shared_ptr<KeyFrame> &, const std::shared_ptr<KeyFrame> &, std::vector<int> &) const =
            &ORBMatcher::SearchForTriangulation;
    int (ORBMatcher::*ini)(const std::shared_ptr<Frame> &, const std::shared_ptr<Frame> &, std::vector<cv::Point2f> &,
                           std::vector<int> &, int) const = &ORBMatcher::SearchForInitialization;
    int (*dd)(const cv::Mat &, const cv::Mat &) = &ORBMatcher::DescriptorDistance;
    // Tracking.cpp:289,321,424: the three const window searches
    int (ORBMatcher::*pf)(const std::shared_ptr<Frame> &, const std::shared_ptr<Frame> &, float) const = &ORBMatcher::SearchByProjection;
    int (ORBMatcher::*pk)(const std::shared_ptr<KeyFrame> &, const std::shared_ptr<Frame> &, float) const = &ORBMatcher::SearchByProjection;
    int (ORBMatcher::*pp)(const std::shared_ptr<Frame> &, const std::vector<std::shared_ptr<MapPoint>> &, float) const =
            &ORBMatcher::SearchByProjection;
    // ORBMatcher.h:44-45: the static fuse, exactly as LocalMapping.cpp:282,301 calls it (three arguments, th defaulted)
    int (*fuse)(const std::shared_ptr<KeyFrame> &, const std::vector<std::shared_ptr<MapPoint>> &, Map *, float) =
            &ORBMatcher::SearchByProjection;
    float (*sf)(int) = &ORBExtractor::getScaleFactor;
    // Frame.cpp:175-176 and ORBVocabulary.cpp:10-25
    void (Vocabulary::*tr)(const std::vector<cv::Mat> &, DBoW2::BowVector &, DBoW2::FeatureVector &, int) const = &Vocabulary::transform;
    bool (*mk)(const std::string &) = &ORBVocabulary::createORBVocabulary;
    const Vocabulary *(*get)() = &ORBVocabulary::getORBVocabulary;
    bool (FramePost::*fp)(std::vector<cv::KeyPoint> &, std::vector<cv::KeyPoint> &,
                          std::vector<std::vector<std::vector<size_t>>> &) const = &FramePost::operator();
    return op && bow && tri && ini && dd && pf && pk && pp && fuse && sf && tr && mk && get && fp;
}

// The call sites of the reference's two threads, spelled as they are there, so that a missing declaration or a
// changed default argument breaks this file (the header swap broke LocalMapping.cpp:282,301).
namespace {
    struct CallSites : mono_orb_slam3::ORBMatcher { // ComputeThreeMaxima is protected (ORBMatcher.h:47-48)
        static void threeMaxima(std::vector<int> *h, int &a, int &b, int &c) { ComputeThreeMaxima(h, a, b, c); }
    };
}
int shim_check_call_sites(const std::shared_ptr<mono_orb_slam3::KeyFrame> &kf,
                          const std::shared_ptr<mono_orb_slam3::KeyFrame> &current_kf,
                          const std::shared_ptr<mono_orb_slam3::Frame> &last_frame,
                          const std::shared_ptr<mono_orb_slam3::Frame> &current_frame, mono_orb_slam3::Map *point_map) {
    using namespace mono_orb_slam3;
    int n = 0;
    {   // LocalMapping.cpp:168
        ORBMatcher matcher(0.6, false);
        std::vector<int> matches12;
        n += matcher.SearchForTriangulation(kf, current_kf, matches12);
    }
    {   // LocalMapping.cpp:280-282, :301
        std::vector<std::shared_ptr<MapPoint>> curMapPoints = current_kf->getMapPoints();
        n += ORBMatcher::SearchByProjection(kf, curMapPoints, point_map);
        std::vector<std::shared_ptr<MapPoint>> fuseMapPoints;
        ORBMatcher::SearchByProjection(current_kf, fuseMapPoints, point_map);
    }
    {   // Tracking.cpp:262, :289, :295, :321, :327, :424
        ORBMatcher matcher(0.7, true);
        n += matcher.SearchByBow(kf, current_frame);
        ORBMatcher matcher2(0.9, true);
        n += matcher2.SearchByProjection(last_frame, current_frame, 7);
        n += matcher2.SearchByProjection(last_frame, current_frame, 14);
        n += matcher2.SearchByProjection(kf, current_frame, 10);
        std::vector<std::shared_ptr<MapPoint>> local_map_points;
        ORBMatcher matcher3(0.8);
        n += matcher3.SearchByProjection(current_frame, local_map_points, 3);
    }
    {   // ORBMatcher.cpp:99, :188, :509 (inside the class)
        std::vector<int> rotHist[ORBM_HISTO_LENGTH];
        int ind1 = -1, ind2 = -1, ind3 = -1;
        CallSites::threeMaxima(rotHist, ind1, ind2, ind3);
        n += ind1;
    }
    return n;
}
