// Drop-in replacement for the reference's modules/ORB/ORBMatcher.h (header-only shim over liborbx.so).
//
// Same class, constructor and method signatures as the reference (modules/ORB/ORBMatcher.h:12-52) for the
// routines on the north-star path: DescriptorDistance, SearchForInitialization, SearchByBow and
// SearchForTriangulation, and (inside the reference tree, where Camera / Pose exist) the three const
// SearchByProjection overloads.  The Hamming brute force runs in HIP (include/orbm.h); the side effects on
// Frame::map_points / matches12 are applied here so that the reference's objects stay the owners.
// The static fuse SearchByProjection(keyFrame, mapPoints, Map*, th) (ORBMatcher.h:44-45, ORBMatcher.cpp:524-592) gets its
// per-point window search from orbm_search_fuse and replays the observation rewiring (:577-591) here, in map-point
// order, on the reference's own MapPoint / KeyFrame objects.  ComputeThreeMaxima (ORBMatcher.h:48) forwards to
// orbm_three_maxima.  With these two the header declares every member of the reference's class, so the bodies in
// modules/ORB/ORBMatcher.cpp are no longer compiled (drop the file from the target).
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
#include "ORBExtractor.h" // the reference reaches it through BasicObject/Frame.h
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

        // The projection maths below follows the reference call for call (it needs its Camera / Pose / MapPoint classes;
        // this repo's checks run it on the mirror types of ref_mirror.h); candidate gathering, Hamming distances and the
        // greedy pass run behind the C ABI.

        // reference ORBMatcher.h:28-30 / ORBMatcher.cpp:203-274
        [[nodiscard]] int SearchByProjection(const std::shared_ptr<Frame> &lastFrame, const std::shared_ptr<Frame> &curFrame,
                                             float th = 5) const {
            return projectionFromFrame(lastFrame->num_kps, lastFrame->map_points, lastFrame->key_points, curFrame, th);
        }

        // reference ORBMatcher.h:32-33 / ORBMatcher.cpp:276-348
        [[nodiscard]] int SearchByProjection(const std::shared_ptr<KeyFrame> &lastKF, const std::shared_ptr<Frame> &curFrame,
                                             float th = 5) const {
            return projectionFromFrame(lastKF->num_kps, lastKF->getMapPoints(), lastKF->key_points, curFrame, th);
        }

        // reference ORBMatcher.h:35-37 / ORBMatcher.cpp:350-415
        [[nodiscard]] int SearchByProjection(const std::shared_ptr<Frame> &frame,
                                             const std::vector<std::shared_ptr<MapPoint>> &mapPoints, float th = 3) const {
            const int nq = (int) mapPoints.size(), n2 = frame->num_kps;
            std::vector<unsigned char> desc((size_t) nq * 32, 0), ok((size_t) nq, 0);
            std::vector<float> xy((size_t) nq * 2, 0.f), radius((size_t) nq, 0.f);
            std::vector<int32_t> level((size_t) nq, 0);
            for (int i = 0; i < nq; ++i) {
                const auto &mp = mapPoints[i];
                if (!mp->track_in_view || mp->isBad()) continue;                    // :355
                ok[i] = 1;
                level[i] = mp->track_scale_level;
                float r = th;
                if (mp->track_view_cos > 0.998) r *= 2.5f; else r *= 4.f;          // :362-364
                radius[i] = r * ORBExtractor::getScaleFactor(level[i]);            // :365
                xy[2 * i] = mp->track_proj_x, xy[2 * i + 1] = mp->track_proj_y;
                const cv::Mat d = mp->getDescriptor();
                std::memcpy(&desc[(size_t) i * 32], d.ptr(), 32);
            }
            const int32_t OCCUPIED = INT32_MAX;
            std::vector<int32_t> fmp((size_t) n2);
            for (int j = 0; j < n2; ++j)
                fmp[j] = (frame->map_points[j] && !frame->map_points[j]->isBad()) ? OCCUPIED : -1; // :383
            int n = 0;
            int32_t counters[3];
            check(orbm_search_by_projection_points(handle(), nn_ratio, desc.data(), xy.data(), radius.data(), level.data(),
                                                   ok.data(), nq, frame->key_points.data(), rows(frame->descriptors), n2,
                                                   imageCols(*frame), imageRows(*frame), fmp.data(), &n, counters));
            for (int j = 0; j < n2; ++j)
                if (fmp[j] >= 0 && fmp[j] != OCCUPIED) frame->map_points[j] = mapPoints[fmp[j]]; // :406
            tracker_logger << titles[0] << "out view and bad " << counters[0] << ", fail1 " << counters[1] << ", fail2 "
                           << counters[2] << "\n";                                                // :411-412
            return n;
        }

        // reference ORBMatcher.h:44-45 / ORBMatcher.cpp:524-592 (LocalMapping.cpp:282,301).  Static, like the reference's.
        static int SearchByProjection(const std::shared_ptr<KeyFrame> &keyFrame,
                                      const std::vector<std::shared_ptr<MapPoint>> &mapPoints, Map *pointMap, float th = 3) {
            (void) pointMap; // the reference does not read it either (:524-592); MapPoint::replace holds its own Map*
            const Camera *camera = Camera::getCamera();
            const Pose Tcw = keyFrame->getPose();
            const Eigen::Vector3f Ow = keyFrame->getCameraCenter();
            const int nq = (int) mapPoints.size(), n1 = keyFrame->num_kps;
            std::vector<unsigned char> desc((size_t) nq * 32, 0), ok((size_t) nq, 0);
            std::vector<float> xy((size_t) nq * 2, 0.f), radius((size_t) nq, 0.f);
            std::vector<int32_t> level((size_t) nq, 0);
            // :534-556 for every point, except the three tests on mutable state (null / bad / already observed), which are
            // evaluated live in the replay loop below: nothing computed here is changed by an earlier point's rewiring
            for (int i = 0; i < nq; ++i) {
                const std::shared_ptr<MapPoint> &mp = mapPoints[i];
                if (mp == nullptr) continue;
                const Eigen::Vector3f Pw = mp->getPos();
                const Eigen::Vector3f Pc = Tcw.R * Pw + Tcw.t;                      // :537
                if (Pc[2] < 0) continue;
                const cv::Point2f p = camera->project(Pc);
                if (!camera->isInImage(p)) continue;                               // :540-541
                const Eigen::Vector3f OP = Pw - Ow;
                const float distance = OP.norm();
                const float maxDistance = mp->getMaxDistanceInvariance();
                const float minDistance = mp->getMinDistanceInvariance();
                if (distance < minDistance || distance > maxDistance) continue;    // :547
                const Eigen::Vector3f Pn = mp->getAverageDirection();
                if (OP.dot(Pn) < 0.5 * distance) continue;                         // :550
                const int predictLevel = mp->predictScaleLevel(distance);
                ok[i] = 1;
                level[i] = predictLevel;
                radius[i] = th * ORBExtractor::getScaleFactor(predictLevel);       // :553
                xy[2 * i] = p.x, xy[2 * i + 1] = p.y;
                const cv::Mat d = mp->getDescriptor();
                std::memcpy(&desc[(size_t) i * 32], d.ptr(), 32);
            }
            const std::vector<float> sigma2 = ORBExtractor::getSquareSigmas();
            std::vector<int32_t> bestIdx((size_t) nq, -1), bestDist((size_t) nq, 0);
            int found = 0;
            check(orbm_search_fuse(handle(), desc.data(), xy.data(), radius.data(), level.data(), ok.data(), nq,
                                   keyFrame->key_points.data(), rows(keyFrame->descriptors), n1, imageCols(*keyFrame),
                                   imageRows(*keyFrame), sigma2.data(), (int) sigma2.size(), bestIdx.data(),
                                   bestDist.data(), &found));
            int numMatch = 0;
            for (int i = 0; i < nq; ++i) {
                const std::shared_ptr<MapPoint> &mp = mapPoints[i];
                if (mp == nullptr || mp->isBad() || mp->isObserveKeyFrame(keyFrame)) continue; // :534, live
                if (!ok[i] || bestIdx[i] < 0) continue;                                        // :535-558, :577
                const int bestIdx1 = bestIdx[i];
                std::shared_ptr<MapPoint> mp1 = keyFrame->getMapPoint(bestIdx1);                // :578
                if (mp1 == nullptr) {
                    mp->addObservation(keyFrame, bestIdx1);
                    keyFrame->addMapPoint(mp, bestIdx1);
                } else if (!mp1->isBad()) {
                    if (mp1->getNumObs() > mp->getNumObs()) mp->replace(mp1);
                    else mp1->replace(mp);
                }
                numMatch++;                                                                     // :588
            }
            return numMatch;
        }

    protected:
        // reference ORBMatcher.h:48 / ORBMatcher.cpp:594-622
        static void ComputeThreeMaxima(std::vector<int> *histo, int &ind1, int &ind2, int &ind3) {
            int32_t sizes[ORBM_HISTO_LENGTH];
            for (int i = 0; i < ORBM_HISTO_LENGTH; ++i) sizes[i] = (int32_t) histo[i].size();
            orbm_three_maxima(sizes, ORBM_HISTO_LENGTH, &ind1, &ind2, &ind3);
        }

        int projectionFromFrame(int nq, const std::vector<std::shared_ptr<MapPoint>> &mapPoints,
                                const std::vector<cv::KeyPoint> &keyPoints, const std::shared_ptr<Frame> &curFrame,
                                float th) const {
            const Camera *camera = Camera::getCamera();
            const Pose &Tcw = curFrame->T_cw;
            const int n2 = curFrame->num_kps;
            std::vector<unsigned char> desc((size_t) nq * 32, 0), ok((size_t) nq, 0);
            std::vector<float> xy((size_t) nq * 2, 0.f), radius((size_t) nq, 0.f), angle((size_t) nq, 0.f);
            std::vector<int32_t> octave((size_t) nq, 0);
            for (int i = 0; i < nq; ++i) {
                const std::shared_ptr<MapPoint> &mp = mapPoints[i];
                if (mp == nullptr || mp->isBad()) continue;                         // :213-217
                const Eigen::Vector3f Pc = Tcw.map(mp->getPos());                   // :219-220
                if (Pc[2] < 0) continue;
                const cv::Point2f p = camera->project(Pc);
                if (!camera->isInImage(p)) continue;                                // :223-224
                ok[i] = 1;
                xy[2 * i] = p.x, xy[2 * i + 1] = p.y;
                octave[i] = keyPoints[i].octave;
                radius[i] = th * keyPoints[i].size;                                 // :228
                angle[i] = keyPoints[i].angle;
                const cv::Mat d = mp->getDescriptor();
                std::memcpy(&desc[(size_t) i * 32], d.ptr(), 32);
            }
            const int32_t OCCUPIED = INT32_MAX;
            std::vector<int32_t> fmp((size_t) n2);
            for (int j = 0; j < n2; ++j) fmp[j] = curFrame->map_points[j] ? OCCUPIED : -1; // :235
            int n = 0;
            check(orbm_search_by_projection_frame(handle(), be_check_orientation, desc.data(), xy.data(), radius.data(),
                                                  octave.data(), angle.data(), ok.data(), nq, curFrame->key_points.data(),
                                                  rows(curFrame->descriptors), n2, imageCols(*curFrame), imageRows(*curFrame),
                                                  fmp.data(), &n));
            for (int j = 0; j < n2; ++j)
                if (fmp[j] >= 0 && fmp[j] != OCCUPIED) curFrame->map_points[j] = mapPoints[fmp[j]]; // :245
            return n;
        }

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
