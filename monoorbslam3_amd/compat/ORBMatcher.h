// Drop-in replacement for the reference's modules/ORB/ORBMatcher.h (header-only shim over liborbx.so).
//
// Same class, constructor and method signatures as the reference (modules/ORB/ORBMatcher.h:12-52) for the
// routines on the north-star path: DescriptorDistance, SearchForInitialization, SearchByBow and
// SearchForTriangulation.  The Hamming brute force runs in HIP (include/orbm.h); the side effects on
// Frame::map_points / matches12 are applied here so that the reference's objects stay the owners.
// The three SearchByProjection overloads and the static fuse variant are "next" rows: they
// keep the reference's own implementation until their wrappers land.
//
// Build: in a real integration include the reference's BasicObject headers before this one.  For this
// repo's checks define ORBX_SHIM_USE_REF_MIRROR to get minimal mirror types (ref_mirror.h).
#ifndef MONO_ORB_SLAM3_ORBMATCHER_H
#define MONO_ORB_SLAM3_ORBMATCHER_H

#include <climits>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef ORBX_SHIM_USE_REF_MIRROR
#include "ref_mirror.h"
#else
#include "BasicObject/Frame.h"
#include "BasicObject/Map.h"
#endif

#include "orbm.h"
#include "orbx.h"

namespace mono_orb_slam3 {
    class ORBMatcher {
    public:
        // reference ORBMatcher.h:14
        explicit ORBMatcher(float nnRatio = 0.6, bool checkOrientation = true)
                : nn_ratio(nnRatio), be_check_orientation(checkOrientation) {}

        // reference ORBMatcher.h:18 / ORBMatcher.cpp:17-31.  A single pair is host work (MapPoint.cpp:130 calls it in a
        // tight loop on already-resident rows): the same xor + popcount the device kernels use.
        static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
            const unsigned int *pa = a.ptr<unsigned int>();
            const unsigned int *pb = b.ptr<unsigned int>();
            int dist = 0;
            for (int i = 0; i < 8; ++i) dist += __builtin_popcount(pa[i] ^ pb[i]);
            return dist;
        }

        // reference ORBMatcher.h:21-23 / ORBMatcher.cpp:33-116
        int SearchForInitialization(const std::shared_ptr<Frame> &frame1, const std::shared_ptr<Frame> &frame2,
                                    std::vector<cv::Point2f> &vecPreMatched, std::vector<int> &matches12,
                                    int windowSize = 100) const {
            matches12.assign((size_t) frame1->num_kps, -1);
            std::vector<float> pre((size_t) frame1->num_kps * 2);
            for (int i = 0; i < frame1->num_kps; ++i) pre[2 * i] = vecPreMatched[i].x, pre[2 * i + 1] = vecPreMatched[i].y;
            int n = 0;
            check(orbm_search_for_initialization(handle(), nn_ratio, be_check_orientation, frame1->key_points.data(),
                                                 rows(frame1->descriptors), frame1->num_kps, frame2->key_points.data(),
                                                 rows(frame2->descriptors), frame2->num_kps, imageCols(*frame2),
                                                 imageRows(*frame2), pre.data(), matches12.data(), windowSize, &n));
            for (int i = 0; i < frame1->num_kps; ++i) vecPreMatched[i] = cv::Point2f(pre[2 * i], pre[2 * i + 1]);
            return n;
        }

        // reference ORBMatcher.h:26 / ORBMatcher.cpp:118-201
        [[nodiscard]] int SearchByBow(const std::shared_ptr<KeyFrame> &keyFrame, const std::shared_ptr<Frame> &frame) const {
            const std::vector<std::shared_ptr<MapPoint>> mapPoints = keyFrame->getMapPoints();
            const int n1 = keyFrame->num_kps, n2 = frame->num_kps;
            std::vector<unsigned char> ok((size_t) n1);
            std::vector<float> a1((size_t) n1), a2((size_t) n2);
            for (int i = 0; i < n1; ++i) {
                ok[i] = mapPoints[i] != nullptr && !mapPoints[i]->isBad(); // :143
                a1[i] = keyFrame->key_points[i].angle;
            }
            const int32_t OCCUPIED = INT32_MAX;
            std::vector<int32_t> mp((size_t) n2);
            for (int j = 0; j < n2; ++j) {
                mp[j] = frame->map_points[j] != nullptr ? OCCUPIED : -1;    // :150
                a2[j] = frame->key_points[j].angle;
            }
            Csr f1(keyFrame->feature_vector), f2(frame->feature_vector);
            int n = 0;
            check(orbm_search_by_bow(handle(), nn_ratio, be_check_orientation, rows(keyFrame->descriptors), a1.data(),
                                     ok.data(), n1, &f1.fv, rows(frame->descriptors), a2.data(), mp.data(), n2, &f2.fv, &n));
            for (int j = 0; j < n2; ++j)
                if (mp[j] >= 0 && mp[j] != OCCUPIED) frame->map_points[j] = mapPoints[mp[j]]; // :165 (after the :187-198 filter)
            return n;
        }

        // reference ORBMatcher.h:40-42 / ORBMatcher.cpp:417-522
        int SearchForTriangulation(const std::shared_ptr<KeyFrame> &keyFrame1, const std::shared_ptr<KeyFrame> &keyFrame2,
                                   std::vector<int> &matches12) const {
            const int n1 = keyFrame1->num_kps, n2 = keyFrame2->num_kps;
            std::vector<unsigned char> h1((size_t) n1), h2((size_t) n2);
            std::vector<float> a1((size_t) n1), a2((size_t) n2);
            for (int i = 0; i < n1; ++i) h1[i] = keyFrame1->hasMapPoint(i), a1[i] = keyFrame1->key_points[i].angle;
            for (int j = 0; j < n2; ++j) h2[j] = keyFrame2->hasMapPoint(j), a2[j] = keyFrame2->key_points[j].angle;
            Csr f1(keyFrame1->feature_vector), f2(keyFrame2->feature_vector);
            matches12.assign((size_t) n1, -1);
            int n = 0;
            check(orbm_search_for_triangulation(handle(), be_check_orientation, rows(keyFrame1->descriptors), a1.data(),
                                                h1.data(), n1, &f1.fv, rows(keyFrame2->descriptors), a2.data(), h2.data(),
                                                n2, &f2.fv, matches12.data(), &n));
            return n;
        }

    protected:
        // DBoW2::FeatureVector (an ordered map node -> feature indices) flattened for the C ABI
        struct Csr {
            std::vector<uint32_t> ids, idx;
            std::vector<int32_t> off;
            orbm_fv fv;
            template<typename FV> explicit Csr(const FV &v) {
                off.push_back(0);
                for (const auto &kv : v) {
                    ids.push_back((uint32_t) kv.first);
                    for (unsigned int i : kv.second) idx.push_back(i);
                    off.push_back((int32_t) idx.size());
                }
                fv.n_nodes = (int32_t) ids.size();
                fv.node_ids = ids.data(), fv.offsets = off.data(), fv.indices = idx.data();
            }
        };

        static const unsigned char *rows(const cv::Mat &d) {
            if (!d.isContinuous()) throw std::runtime_error("descriptors must be a continuous n x 32 matrix");
            return d.ptr();
        }
#ifdef ORBX_SHIM_USE_REF_MIRROR
        static int imageCols(const Frame &f) { return f.img_cols; }
        static int imageRows(const Frame &f) { return f.img_rows; }
#else
        static int imageCols(const Frame &f) { return f.img.cols; }
        static int imageRows(const Frame &f) { return f.img.rows; }
#endif
        static void check(int rc) {
            if (rc != ORBX_OK) throw std::runtime_error(std::string("orbm: ") + orbx_last_error());
        }
        // one orbm handle (HIP stream + scratch) per host thread: Tracking and LocalMapping call concurrently
        static orbm_t *handle() {
            struct Holder {
                orbm_t *h = nullptr;
                Holder() { check(orbm_create(-1, &h)); }
                ~Holder() { orbm_destroy(h); }
            };
            static thread_local Holder holder;
            return holder.h;
        }

        float nn_ratio;
        bool be_check_orientation;
    };
} // mono_orb_slam3

#endif //MONO_ORB_SLAM3_ORBMATCHER_H
