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
    float (*sf)(int) = &ORBExtractor::getScaleFactor;
    // Frame.cpp:175-176 and ORBVocabulary.cpp:10-25
    void (Vocabulary::*tr)(const std::vector<cv::Mat> &, DBoW2::BowVector &, DBoW2::FeatureVector &, int) const = &Vocabulary::transform;
    bool (*mk)(const std::string &) = &ORBVocabulary::createORBVocabulary;
    const Vocabulary *(*get)() = &ORBVocabulary::getORBVocabulary;
    bool (FramePost::*fp)(std::vector<cv::KeyPoint> &, std::vector<cv::KeyPoint> &,
                          std::vector<std::vector<std::vector<size_t>>> &) const = &FramePost::operator();
    return op && bow && tri && ini && dd && sf && tr && mk && get && fp;
}
