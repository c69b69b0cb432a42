// End-to-end run of the C++ shims (compat/ORBExtractor.h, compat/ORBMatcher.h) the way the reference's
// Tracking thread would call them.  Input: raw file "W H" + two frames of W*H bytes.  Output: binary dump
// (counts, keypoints, descriptors, matches) that tests/test_shim_gpu.py compares with the ctypes path.
#define ORBX_SHIM_USE_CV_MIRROR
#define ORBX_SHIM_USE_REF_MIRROR
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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

    // ---- the static fuse (LocalMapping.cpp:282,301) on mirror MapPoints: map points = the key frame's own features seen
    // again (projection exactly representable, so the Python side reproduces it bit for bit), some slots already
    // taken, some points null / bad / listed twice
    auto elsewhere = [] { // another key frame that observes a point in its slot 0
        auto k = std::make_shared<KeyFrame>();
        k->map_points.assign(1, nullptr);
        return k;
    };
    Camera *cam = Camera::instance();
    cam->width = W, cam->height = H, cam->fx = 512.f, cam->fy = 512.f, cam->cx = 376.f, cam->cy = 240.f;
    auto kf3 = std::make_shared<KeyFrame>();
    fill(*kf3, initial, ia, 6);
    const int n3 = kf3->num_kps;
    std::vector<std::shared_ptr<MapPoint>> fusePoints;
    for (int i = 0; i < n3; ++i) {
        if (i % 11 == 3) { fusePoints.push_back(nullptr); continue; }
        auto mp = std::make_shared<MapPoint>();
        const cv::KeyPoint &kp = kf3->key_points[i];
        const float u = std::floor(kp.pt.x * 8.f) / 8.f + float((i * 37) % 17 - 8) / 8.f;
        const float v = std::floor(kp.pt.y * 8.f) / 8.f + float((i * 53) % 13 - 6) / 8.f;
        mp->pos = Eigen::Vector3f((u - cam->cx) / 128.f, (v - cam->cy) / 128.f, 4.f); // projects to exactly (u, v)
        mp->normal = mp->pos;                                                          // OP . Pn = |OP|^2 >= 0.5 |OP|
        mp->predict_level = std::min(kp.octave + (i % 5 == 0 ? 1 : 0), 7);
        mp->descriptor.create(1, 32, CV_8U);
        std::memcpy(mp->descriptor.ptr(), kf3->descriptors.ptr(i), 32);
        mp->descriptor.ptr()[i % 32] ^= (unsigned char) (1u << (i % 8));
        mp->descriptor.ptr()[(i * 7) % 32] ^= (unsigned char) (1u << ((i / 3) % 8));
        mp->bad = i % 29 == 7;
        for (int o = 0; o < i % 4; ++o) mp->addObservation(elsewhere(), 0); // numObs 0..3
        fusePoints.push_back(mp);
        if (i % 17 == 5) fusePoints.push_back(mp); // listed twice: the second visit finds it observed (:534)
    }
    std::vector<std::shared_ptr<MapPoint>> foreign; // map points the key frame holds before the fuse
    for (int i = 0; i < n3; ++i) {
        if (i % 6 != 1) continue;
        auto mp = std::make_shared<MapPoint>();
        mp->bad = i % 5 == 0;
        for (int o = 0; o < (i / 6) % 4; ++o) mp->addObservation(elsewhere(), 0);
        mp->addObservation(kf3, (size_t) i);
        kf3->map_points[i] = mp;
        foreign.push_back(mp);
    }
    Map point_map;
    const int n_fuse = ORBMatcher::SearchByProjection(kf3, fusePoints, &point_map);
    std::vector<int> fuse_slot((size_t) n3, -1), fuse_bad(fusePoints.size(), 0);
    for (int i = 0; i < n3; ++i) {
        if (!kf3->map_points[i]) continue;
        fuse_slot[i] = -2; // a foreign point
        for (size_t k = 0; k < fusePoints.size(); ++k)
            if (fusePoints[k] == kf3->map_points[i]) { fuse_slot[i] = (int) k; break; }
    }
    for (size_t k = 0; k < fusePoints.size(); ++k) fuse_bad[k] = fusePoints[k] ? (int) fusePoints[k]->bad : -1;

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
    int fhdr[3] = {n_fuse, n3, (int) fusePoints.size()};
    std::fwrite(fhdr, sizeof(int), 3, out);
    std::fwrite(fuse_slot.data(), sizeof(int), fuse_slot.size(), out);
    std::fwrite(fuse_bad.data(), sizeof(int), fuse_bad.size(), out);
    std::fclose(out);
    std::printf("shim smoke: %d + %d keypoints, init %d, bow %d, tri %d, dist(0,0)=%d\n", kf->num_kps, fr->num_kps, n_ini,
                n_bow, n_tri, ORBMatcher::DescriptorDistance(kf->descriptors.row(0), fr->descriptors.row(0)));
    return 0;
}
