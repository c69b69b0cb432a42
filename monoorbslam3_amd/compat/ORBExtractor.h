// Drop-in replacement for the reference's modules/ORB/ORBExtractor.h (header-only shim over liborbx.so).
//
// Same namespace, class name, constructors, operator() and static getters as the reference
// (modules/ORB/ORBExtractor.h:27-122), so System / Tracking / Frame compile and call it unchanged
// (Frame.cpp:20 is the only operator() call site).  All pixel work happens in the HIP library behind
// include/orbx.h; this header only converts between cv::Mat / cv::KeyPoint and plain buffers.
//
// Build: with OpenCV present, include <opencv2/core/core.hpp> first (as the reference does).  Without
// OpenCV (this repo's own checks) define ORBX_SHIM_USE_CV_MIRROR to get the minimal cv:: mirror types.
#ifndef MONO_ORB_SLAM3_ORBEXTRACTOR_H
#define MONO_ORB_SLAM3_ORBEXTRACTOR_H

#include <cassert>
#include <cmath>
#include <cstring>
#include <iostream>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef ORBX_SHIM_USE_CV_MIRROR
#include "cv_mirror.h"
#else
#include <opencv2/core/core.hpp>
#endif

#include "orbx.h"

namespace mono_orb_slam3 {
    typedef std::pair<unsigned int, unsigned int> Match; // kept: declared in the reference header (:14)

    class ORBExtractor {
    public:
        // reference ORBExtractor.h:29-30 / ORBExtractor.cpp:424-475
        explicit ORBExtractor(int nFeatures = 1000, float scaleFactor = 1.2, int nLevels = 8, int iniThFast = 20,
                              int minThFast = 10) {
            orbx_cfg cfg;
            std::memset(&cfg, 0, sizeof cfg);
            cfg.n_features = nFeatures;
            cfg.scale_factor = scaleFactor;
            cfg.n_levels = nLevels;
            cfg.ini_th_fast = iniThFast;
            cfg.min_th_fast = minThFast;
            cfg.max_batch = 1;
            cfg.device = -1;
            check(orbx_create(&cfg, &handle_));
            n_features = nFeatures, ini_th_fast = iniThFast, min_th_fast = minThFast;
            publishTables(scaleFactor); // the reference keeps the pyramid tables in static members written by this ctor
        }

        // reference ORBExtractor.h:32 / ORBExtractor.cpp:477-493 (the 2N "initial" extractor, Tracking.cpp:24)
        ORBExtractor(int nFeatures, const ORBExtractor &orbExtractor) {
            check(orbx_create_requota(orbExtractor.handle_, nFeatures, &handle_));
            n_features = nFeatures, ini_th_fast = orbExtractor.ini_th_fast, min_th_fast = orbExtractor.min_th_fast;
            loadQuotas();
        }

        ORBExtractor(const ORBExtractor &) = delete;
        ORBExtractor &operator=(const ORBExtractor &) = delete;

        ~ORBExtractor() { orbx_destroy(handle_); }

        // reference ORBExtractor.h:38-39 / ORBExtractor.cpp:495-547
        void operator()(const cv::Mat &image, std::vector<cv::KeyPoint> &keyPoints, cv::Mat &descriptors) {
            if (image.empty()) return;                  // :497
            assert(image.type() == CV_8UC1);            // :499
            static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_kp), "cv::KeyPoint must be the 28-byte record");
            const int cap = orbx_max_keypoints(handle_, image.cols, image.rows);
            if (cap < 0) throw std::runtime_error(orbx_last_error());
            kp_buf_.resize((size_t) cap);
            desc_buf_.resize((size_t) cap * 32);
            int n = 0;
            check(orbx_extract(handle_, image.data, image.cols, image.rows, (int) image.step, kp_buf_.data(),
                               desc_buf_.data(), cap, &n));
            if (n == 0) return;                         // :512 -- outputs untouched
            descriptors.create(n, 32, CV_8U);           // :514
            for (int i = 0; i < n; ++i) std::memcpy(descriptors.ptr(i), desc_buf_.data() + (size_t) i * 32, 32);
            keyPoints.clear();                          // :517
            keyPoints.resize((size_t) n);
            std::memcpy(static_cast<void *>(keyPoints.data()), kp_buf_.data(), sizeof(orbx_kp) * (size_t) n);
        }

        // reference ORBExtractor.h:42 / ORBExtractor.cpp:549-557
        void print() const {
            std::cout << std::endl << "ORB Pyramid Information: " << std::endl;
            std::cout << " - Features: " << n_features << "(at initial stage)" << std::endl;
            std::cout << " - ScaleFactor: " << scale_factor << std::endl;
            std::cout << " - Levels: " << n_levels << std::endl;
            std::cout << " - IniThFAST: " << ini_th_fast << std::endl;
            std::cout << " - MinThFAST: " << min_th_fast << std::endl;
            std::cout << std::endl;
        }

        // static getters, reference ORBExtractor.h:44-86
        inline static float getScaleFactor(int level = 0) {
            assert(level >= 0 && level < n_levels);
            return scale_factors[level];
        }
        inline static float getLogScaleFactor() { return log_sale_factor; }
        inline static float getMaxScaleFactor() { return scale_factors[n_levels - 1]; }
        inline static std::vector<float> getScaleFactors() { return scale_factors; }
        inline static float getInvScaleFactor(int level) {
            assert(level >= 0 && level < n_levels);
            return inv_scale_factors[level];
        }
        inline static std::vector<float> getInvScaleFactors() { return inv_scale_factors; }
        inline static int getNumLevels() { return n_levels; }
        inline static std::vector<float> getSquareSigmas() { return square_sigmas; }
        inline static float getSquareSigma(int level) {
            assert(level >= 0 && level < n_levels);
            return square_sigmas[level];
        }
        inline static float getInvSquareSigma(int level) {
            assert(level >= 0 && level < n_levels);
            return inv_square_sigmas[level];
        }

        // reference ORBExtractor.h:88: a public member nobody outside the class reads (grep).  The pyramid lives in
        // HBM; fetchPyramid() copies the levels of the last operator() call here for tools that want them.
        std::vector<cv::Mat> image_pyramid;

        void fetchPyramid(int width, int height) {
            image_pyramid.resize((size_t) n_levels);
            for (int l = 0; l < n_levels; ++l) {
                int w = 0, h = 0;
                check(orbx_level_size(handle_, width, height, l, &w, &h));
                image_pyramid[l].create(h, w, CV_8U);
                std::vector<unsigned char> tmp((size_t) w * h);
                check(orbx_tap_level(handle_, 0, l, 0, tmp.data(), tmp.size()));
                for (int y = 0; y < h; ++y) std::memcpy(image_pyramid[l].ptr(y), tmp.data() + (size_t) y * w, (size_t) w);
            }
        }

        orbx_t *handle() const { return handle_; }
        const std::vector<int> &featuresPerLevel() const { return n_features_per_level; }

    protected:
        static void check(int rc) {
            if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx: ") + orbx_last_error());
        }

        void loadQuotas() {
            int L = 0;
            int32_t q[ORBX_MAX_LEVELS];
            int32_t um[16];
            check(orbx_tables(handle_, &L, nullptr, nullptr, nullptr, nullptr, nullptr, q, um));
            n_features_per_level.assign(q, q + L);
            u_max.assign(um, um + 16);
        }

        void publishTables(float scaleFactor) {
            int L = 0;
            float sf[ORBX_MAX_LEVELS], isf[ORBX_MAX_LEVELS], ss[ORBX_MAX_LEVELS], iss[ORBX_MAX_LEVELS], lsf = 0;
            check(orbx_tables(handle_, &L, sf, isf, ss, iss, &lsf, nullptr, nullptr));
            n_levels = L;
            scale_factor = scaleFactor; // stored unconditionally, also for one level (ORBExtractor.cpp:427)
            log_sale_factor = lsf;
            scale_factors.assign(sf, sf + L);
            inv_scale_factors.assign(isf, isf + L);
            square_sigmas.assign(ss, ss + L);
            inv_square_sigmas.assign(iss, iss + L);
            loadQuotas();
        }

        orbx_t *handle_ = nullptr;
        std::vector<orbx_kp> kp_buf_;
        std::vector<unsigned char> desc_buf_;

        int n_features;
        int ini_th_fast;
        int min_th_fast;

        // pyramid information (static in the reference, ORBExtractor.h:109-115)
        inline static float scale_factor = 1.f;
        inline static float log_sale_factor = 1.f;
        inline static int n_levels = 1;
        inline static std::vector<float> scale_factors;
        inline static std::vector<float> inv_scale_factors;
        inline static std::vector<float> square_sigmas;
        inline static std::vector<float> inv_square_sigmas;

        std::vector<int> n_features_per_level;
        std::vector<int> u_max;
    };

} // mono_orb_slam3

#endif //MONO_ORB_SLAM3_ORBEXTRACTOR_H
