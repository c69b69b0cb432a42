// The steps either side of the extractor, driven from C++ the way the reference's Frame constructor and
// Frame::computeBow do (Frame.cpp:20-51, :168-178), through the shims compat/FramePost.h and compat/ORBVocabulary.h,
// followed by ORBMatcher::SearchByBow on the resulting FeatureVectors.
// Input: raw file "W H" + two frames; vocabulary text file.  Output: binary dump checked by tests/test_shim_gpu.py.
#define ORBX_SHIM_USE_CV_MIRROR
#define ORBX_SHIM_USE_REF_MIRROR
#include <cstdio>
#include <cstdlib>

#include "FramePost.h"
#include "ORBExtractor.h"
#include "ORBMatcher.h"
#include "ORBVocabulary.h"

using namespace mono_orb_slam3;

struct Rec {
    std::vector<cv::KeyPoint> raw, un;
    cv::Mat desc;
    std::vector<std::vector<std::vector<size_t>>> grid;
    DBoW2::BowVector bow;
    DBoW2::FeatureVector fv;
};

static void write_vec(FILE *f, const void *p, size_t bytes) { std::fwrite(p, 1, bytes, f); }

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    int W = 0, H = 0;
    if (!in || std::fscanf(in, "%d %d\n", &W, &H) != 2) return 3;
    std::vector<unsigned char> img[2];
    for (auto &v : img) {
        v.resize((size_t) W * H);
        if (std::fread(v.data(), 1, v.size(), in) != v.size()) return 4;
    }
    std::fclose(in);
    if (!ORBVocabulary::createORBVocabulary(argv[2])) return 5;
    if (ORBVocabulary::createORBVocabulary(argv[2])) return 6; // second call: already created (ORBVocabulary.cpp:11,20)
    const Vocabulary *voc = ORBVocabulary::getORBVocabulary();

    ORBExtractor extractor(1000, 1.2f, 8, 20, 7);
    FramePost post(W, H, 458.654f, 457.296f, 367.215f, 248.375f, {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f});
    if (!post.ok()) return 7;
    Rec rec[2];
    auto kf = std::make_shared<KeyFrame>();
    auto fr = std::make_shared<Frame>();
    Frame *frames[2] = {kf.get(), fr.get()};
    for (int k = 0; k < 2; ++k) {
        cv::Mat m(H, W, CV_8UC1, img[k].data());
        extractor(m, rec[k].raw, rec[k].desc);                 // Frame.cpp:20
        if (!post(rec[k].raw, rec[k].un, rec[k].grid)) return 8; // Frame.cpp:24-51
        std::vector<cv::Mat> rows(rec[k].raw.size());           // Frame.cpp:170-173
        for (size_t i = 0; i < rows.size(); ++i) rows[i] = rec[k].desc.row((int) i);
        voc->transform(rows, rec[k].bow, rec[k].fv, 2);        // Frame.cpp:176 (levelsup 4 on ORBvoc's L = 6)
        Frame &f = *frames[k];
        f.num_kps = (int) rec[k].un.size();
        f.img_cols = W, f.img_rows = H;
        f.key_points = rec[k].un;
        f.descriptors = rec[k].desc;
        f.feature_vector = rec[k].fv;
        f.map_points.assign(rec[k].un.size(), nullptr);
    }
    for (int i = 0; i < kf->num_kps; ++i) kf->map_points[i] = std::make_shared<MapPoint>();
    ORBMatcher m07(0.7f, true);
    const int n_bow = m07.SearchByBow(kf, fr);

    FILE *out = std::fopen(argv[3], "wb");
    for (int k = 0; k < 2; ++k) {
        const int n = (int) rec[k].raw.size(), nw = (int) rec[k].bow.size(), nf = (int) rec[k].fv.size();
        int hdr[6] = {n, nw, nf, post.gridCols(), post.gridRows(), n_bow};
        write_vec(out, hdr, sizeof hdr);
        write_vec(out, rec[k].raw.data(), sizeof(cv::KeyPoint) * (size_t) n);
        write_vec(out, rec[k].un.data(), sizeof(cv::KeyPoint) * (size_t) n);
        for (int i = 0; i < n; ++i) write_vec(out, rec[k].desc.ptr(i), 32);
        for (int cx = 0; cx < post.gridCols(); ++cx)
            for (int cy = 0; cy < post.gridRows(); ++cy) {
                const auto &cell = rec[k].grid[(size_t) cx][(size_t) cy];
                const int cnt = (int) cell.size();
                write_vec(out, &cnt, 4);
                for (size_t v : cell) { const int iv = (int) v; write_vec(out, &iv, 4); }
            }
        for (const auto &kv : rec[k].bow) { write_vec(out, &kv.first, 4); write_vec(out, &kv.second, 8); }
        for (const auto &kv : rec[k].fv) {
            const int cnt = (int) kv.second.size();
            write_vec(out, &kv.first, 4);
            write_vec(out, &cnt, 4);
            write_vec(out, kv.second.data(), 4 * (size_t) cnt);
        }
    }
    std::fclose(out);
    std::printf("records smoke: %zu + %zu keypoints, %zu / %zu words, %zu / %zu nodes, bow matches %d\n", rec[0].raw.size(),
                rec[1].raw.size(), rec[0].bow.size(), rec[1].bow.size(), rec[0].fv.size(), rec[1].fv.size(), n_bow);
    return 0;
}
