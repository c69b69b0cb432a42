// End-to-end run of the C++ shims (compat/ORBExtractor.h, compat/ORBMatcher.h) the way the reference's
// Tracking thread would call them.  Input: raw file "W H" + two frames of W*H bytes.  Output: binary dump
// (counts, keypoints, descriptors, matches) that tests/test_shim_gpu.py compares with the ctypes path.
#define ORBX_SHIM_USE_CV_MIRROR
#define ORBX_SHIM_USE_REF_MIRROR
#include <cstdio>
#include <cstdlib>

#include "ORBExtractor.h"
#include "ORBMatcher.h"

using namespace mono_orb_slam3;

static void fill(Frame &f, ORBExtractor &ex, cv::Mat &img, int prefix_bits) {
    ex(img, f.key_points, f.descriptors);
    f.num_kps = (int) f.key_points.size();
    f.img_cols = img.cols, f.img_rows = img.rows;
    f.map_points.assign((size_t) f.num_kps, nullptr);
    for (int i = 0; i < f.num_kps; ++i) {
        const unsigned char *d = f.descriptors.ptr(i);
        unsigned key = (d[0] | (d[1] << 8)) & ((1u << prefix_bits) - 1);
        f.feature_vector.addFeature(key, (unsigned) i);
    }
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    int W = 0, H = 0;
    if (!in || std::fscanf(in, "%d %d\n", &W, &H) != 2) return 3;
    std::vector<unsigned char> a((size_t) W * H), b((size_t) W * H);
    if (std::fread(a.data(), 1, a.size(), in) != a.size() || std::fread(b.data(), 1, b.size(), in) != b.size()) return 4;
    std::fclose(in);
    cv::Mat ia(H, W, CV_8UC1, a.data()), ib(H, W, CV_8UC1, b.data());

    ORBExtractor extractor(1000, 1.2f, 8, 20, 7);
    ORBExtractor initial(2000, extractor); // Tracking.cpp:24
    auto kf = std::make_shared<KeyFrame>();
    auto fr = std::make_shared<Frame>();
    fill(*kf, initial, ia, 6);
    fill(*fr, initial, ib, 6);
    for (int i = 0; i < kf->num_kps; ++i)
        if (i % 3 != 0) kf->map_points[i] = std::make_shared<MapPoint>();

    std::vector<cv::Point2f> pre((size_t) kf->num_kps);
    for (int i = 0; i < kf->num_kps; ++i) pre[i] = kf->key_points[i].pt;
    std::vector<int> m_ini;
    ORBMatcher m09(0.9f, true);
    const int n_ini = m09.SearchForInitialization(kf, fr, pre, m_ini, 100);

    ORBMatcher m07(0.7f, true);
    const int n_bow = m07.SearchByBow(kf, fr);
    std::vector<int> bow_assign((size_t) fr->num_kps, -1);
    for (int j = 0; j < fr->num_kps; ++j)
        if (fr->map_points[j])
            for (int i = 0; i < kf->num_kps; ++i)
                if (kf->map_points[i] == fr->map_points[j]) bow_assign[j] = i;

    auto kf2 = std::make_shared<KeyFrame>();
    fill(*kf2, initial, ib, 6);
    std::vector<int> m_tri;
    ORBMatcher m06(0.6f, false);
    const int n_tri = m06.SearchForTriangulation(kf, kf2, m_tri);

    FILE *out = std::fopen(argv[2], "wb");
    int hdr[6] = {kf->num_kps, fr->num_kps, n_ini, n_bow, n_tri, ORBExtractor::getNumLevels()};
    std::fwrite(hdr, sizeof(int), 6, out);
    std::fwrite(kf->key_points.data(), sizeof(cv::KeyPoint), (size_t) kf->num_kps, out);
    for (int i = 0; i < kf->num_kps; ++i) std::fwrite(kf->descriptors.ptr(i), 1, 32, out);
    std::fwrite(fr->key_points.data(), sizeof(cv::KeyPoint), (size_t) fr->num_kps, out);
    for (int i = 0; i < fr->num_kps; ++i) std::fwrite(fr->descriptors.ptr(i), 1, 32, out);
    std::fwrite(m_ini.data(), sizeof(int), m_ini.size(), out);
    std::fwrite(bow_assign.data(), sizeof(int), bow_assign.size(), out);
    std::fwrite(m_tri.data(), sizeof(int), m_tri.size(), out);
    std::fclose(out);
    std::printf("shim smoke: %d + %d keypoints, init %d, bow %d, tri %d, dist(0,0)=%d\n", kf->num_kps, fr->num_kps, n_ini,
                n_bow, n_tri, ORBMatcher::DescriptorDistance(kf->descriptors.row(0), fr->descriptors.row(0)));
    return 0;
}
