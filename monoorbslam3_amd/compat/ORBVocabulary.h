// Drop-in for the reference's modules/ORB/ORBVocabulary.h + .cpp and for the one DBoW2 entry point the SLAM calls on
// it: Vocabulary::transform(features, BowVector&, FeatureVector&, levelsup) (Frame.cpp:168-178, KeyFrame::computeBow),
// i.e. thirdParty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1201, plus loadFromTextFile (:1338-1420).
// The tree lives on the GPU (include/orbv.h); BowVector / FeatureVector stay the reference's std::map types, so
// ORBMatcher and everything else that reads them is untouched.
#pragma once
#include <string>
#include <vector>

#include "orbv.h"

#if defined(ORBX_SHIM_USE_REF_MIRROR)
#include "ref_mirror.h"
#else
#include <opencv2/core.hpp>
#include "DBoW2/BowVector.h"
#include "DBoW2/FeatureVector.h"
#endif

namespace mono_orb_slam3 {
    // stands where `typedef DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> Vocabulary;` stood (ORBVocabulary.h:12)
    class Vocabulary {
    public:
        Vocabulary() = default;
        Vocabulary(const Vocabulary &) = delete;
        Vocabulary &operator=(const Vocabulary &) = delete;
        ~Vocabulary() { orbv_destroy(h_); }

        // TemplatedVocabulary::loadFromTextFile (:1338-1420): false when the file is missing or malformed
        bool loadFromTextFile(const std::string &filename) {
            orbv_destroy(h_);
            h_ = nullptr;
            if (orbv_load_text(filename.c_str(), /*device*/0, &h_) != ORBX_OK) return false;
            int n_words = 0;
            orbv_info(h_, nullptr, nullptr, nullptr, nullptr, nullptr, &n_words);
            n_words_ = (unsigned) n_words;
            return true;
        }

        bool empty() const { return n_words_ == 0; }   // :546-549
        unsigned int size() const { return n_words_; }  // :539-542

        // TemplatedVocabulary::transform(features, v, fv, levelsup) (:1127-1201)
        void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const {
            v.clear();
            fv.clear();
            const int n = (int) features.size();
            if (empty() || n == 0) return;
            std::vector<unsigned char> desc((size_t) n * 32);
            for (int i = 0; i < n; ++i) std::memcpy(&desc[(size_t) i * 32], features[(size_t) i].ptr(0), 32);
            std::vector<uint32_t> bow_ids((size_t) n), fv_nodes((size_t) n), fv_idx((size_t) n);
            std::vector<double> bow_vals((size_t) n);
            std::vector<int32_t> fv_off((size_t) n + 1);
            int32_t n_words = 0, n_fv = 0;
            if (orbv_transform(h_, desc.data(), n, levelsup, bow_ids.data(), bow_vals.data(), &n_words, fv_nodes.data(),
                               fv_off.data(), fv_idx.data(), &n_fv) != ORBX_OK)
                return; // as an empty vocabulary: both maps stay empty (the C ABI keeps the reason in orbx_last_error())
            for (int i = 0; i < n_words; ++i) v.insert(v.end(), DBoW2::BowVector::value_type(bow_ids[(size_t) i], bow_vals[(size_t) i]));
            for (int r = 0; r < n_fv; ++r)
                fv.insert(fv.end(), DBoW2::FeatureVector::value_type(
                        fv_nodes[(size_t) r], std::vector<unsigned int>(fv_idx.begin() + fv_off[(size_t) r],
                                                                        fv_idx.begin() + fv_off[(size_t) r + 1])));
        }

    private:
        orbv_t *h_ = nullptr;
        unsigned n_words_ = 0;
    };

    // modules/ORB/ORBVocabulary.h:14-24, ORBVocabulary.cpp:8-25
    class ORBVocabulary {
    public:
        static bool createORBVocabulary(const std::string &path) {
            Vocabulary *&voc = slot();
            if (voc == nullptr) {
                voc = new Vocabulary();
                const bool sign = voc->loadFromTextFile(path);
                if (!sign) {
                    delete voc;
                    voc = nullptr;
                }
                return sign;
            }
            return false;
        }

        static const Vocabulary *getORBVocabulary() { return slot(); }

    private:
        ORBVocabulary() = default;

        static Vocabulary *&slot() {
            static Vocabulary *vocabulary = nullptr;
            return vocabulary;
        }
    };
}
