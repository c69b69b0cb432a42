// The reference's two-thread model at the C ABI (include/orbx.h, "Streams and threads"; SURVEY 8b "Threading").
// Tracking runs on the thread that calls System::Track (System.cpp:96): ORBExtractor::operator() (Frame.cpp:20),
// computeBow (Frame.cpp:168), SearchByProjection (Tracking.cpp:289-336), poseOptimize (Tracking.cpp:305).  LocalMapping is a
// std::thread of its own (System.cpp:55): computeBow (LocalMapping.cpp:90), SearchForTriangulation (:168), the fuse's
// SearchByProjection (:282, :301), localBundleAdjustment (:45-52).  Both run at once and share ONE vocabulary object.
//
// Thread T loops  orbx_extract -> orbv_transform -> orbm_search_by_projection_frame -> orbba_pose_optimize_batch,
// thread M loops  orbv_transform -> orbm_search_for_triangulation -> orbm_search_fuse -> orbba_local_bundle_adjustment,
// each on handles of its own (the vocabulary handle is shared, as in the reference).  Every output of every iteration must
// equal, byte for byte, what the same calls return when one thread runs them alone.
// Then the same two threads run DEVICE chains, each on a non-blocking stream of its own (what orbx.h asks of a second thread that
// wants the *_device entry points): T copies a frame up, extracts it and matches it against a resident view (orbx_extract_batch_device
// -> orbm_best2_device), M computes both bag-of-words records and the triangulation matches (orbv_transform_device x2 ->
// orbm_search_for_triangulation_device); results equal the single-thread device chains and the host entry points.
//   two_threads [--iters N]                correctness (exit 0 = all equal)
//   two_threads --latency N [--own-voc]    per-call latency of T alone and with M running (p50 / p90, microseconds);
//                                          --own-voc gives each thread its own vocabulary handle (a library older than
//                                          round 6 has no re-entrant orbv_transform)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "orbba.h"
#include "orbm.h"
#include "orbv.h"
#include "orbx.h"

#define ORB_OK(e)                                                                                                  \
    do {                                                                                                           \
        int e_ = (e);                                                                                              \
        if (e_ != 0) {                                                                                             \
            std::fprintf(stderr, "%s:%d orbx error %d: %s\n", __FILE__, __LINE__, e_, orbx_last_error());          \
            std::exit(11);                                                                                         \
        }                                                                                                          \
    } while (0)

#define HIP_OK(e)                                                                                                  \
    do {                                                                                                           \
        hipError_t e_ = (e);                                                                                       \
        if (e_ != hipSuccess) {                                                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                         \
            std::exit(10);                                                                                         \
        }                                                                                                          \
    } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
static double unif(uint32_t &s) { return (double)(lcg(s) >> 8) / (double)(1u << 24); } // [0, 1)

static const int W = 1242, H = 375, NF = 2000; // BASELINE config 2

// blocks of random grey levels (corners at the block borders) on a ramp; view `f` is shifted by (3 f, 2 f)
static std::vector<uint8_t> make_image(int f)
{
    std::vector<uint8_t> img((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int bx = (x + 3 * f) / 12, by = (y + 2 * f) / 10;
            uint32_t s = (uint32_t)(bx * 7919 + by * 104729 + 17);
            img[(size_t)y * W + x] = (uint8_t)(40 + (lcg(s) >> 25) + ((x * 3 + y * 5) & 7));
        }
    return img;
}

struct View {
    std::vector<uint8_t> img, desc;
    std::vector<orbx_kp> kp;
    int n = 0;
    std::vector<float> angle;
};

// a k-ary tree of random descriptors, nodes in loadFromTextFile's order (orbv.h)
static orbv_t *make_vocabulary(int k, int L)
{
    std::vector<int32_t> parent(1, -1);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    std::vector<int> level(1, 0);
    uint32_t s = 4242;
    for (size_t i = 0; i < parent.size(); ++i) {
        if (level[i] == L) continue;
        for (int c = 0; c < k; ++c) {
            parent.push_back((int32_t)i);
            level.push_back(level[i] + 1);
            leaf.push_back(level[i] + 1 == L);
            for (int b = 0; b < 32; ++b) desc.push_back((uint8_t)(lcg(s) >> 24));
            weight.push_back(0.25 + unif(s));
        }
    }
    orbv_t *v = nullptr;
    ORB_OK(orbv_create(k, L, ORBV_L1_NORM, ORBV_TF_IDF, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), 0, &v));
    return v;
}

struct Bow {
    std::vector<uint32_t> ids, nodes, idx;
    std::vector<double> vals;
    std::vector<int32_t> off;
    int32_t n_words = 0, n_fv = 0;
    void size(int n) { ids.assign(n, 0); nodes.assign(n, 0); idx.assign(n, 0); vals.assign(n, 0.0); off.assign(n + 1, 0); }
    orbm_fv fv() const { orbm_fv f = {n_fv, nodes.data(), off.data(), idx.data()}; return f; }
    bool operator==(const Bow &o) const
    {
        return n_words == o.n_words && n_fv == o.n_fv && !memcmp(ids.data(), o.ids.data(), 4 * (size_t)n_words) &&
               !memcmp(vals.data(), o.vals.data(), 8 * (size_t)n_words) && !memcmp(nodes.data(), o.nodes.data(), 4 * (size_t)n_fv) &&
               !memcmp(off.data(), o.off.data(), 4 * (size_t)(n_fv + 1)) && idx == o.idx;
    }
};
static void compute_bow(orbv_t *v, const View &f, Bow &b)
{
    b.size(f.n);
    ORB_OK(orbv_transform(v, f.desc.data(), f.n, /*levelsup*/ 1, b.ids.data(), b.vals.data(), &b.n_words, b.nodes.data(), b.off.data(),
                          b.idx.data(), &b.n_fv));
}

// ---- what thread T computes in one iteration
struct TOut {
    std::vector<orbx_kp> kp; std::vector<uint8_t> desc; int n = 0;
    Bow bow;
    std::vector<int32_t> frame_mp; int n_proj = 0;
    double R[9], t[3]; std::vector<uint8_t> inlier; int32_t n_inl = 0; std::vector<double> chi2;
    bool operator==(const TOut &o) const
    {
        return n == o.n && !memcmp(kp.data(), o.kp.data(), sizeof(orbx_kp) * (size_t)n) && !memcmp(desc.data(), o.desc.data(), 32 * (size_t)n) &&
               bow == o.bow && frame_mp == o.frame_mp && n_proj == o.n_proj && !memcmp(R, o.R, sizeof R) && !memcmp(t, o.t, sizeof t) &&
               inlier == o.inlier && n_inl == o.n_inl && !memcmp(chi2.data(), o.chi2.data(), 8 * chi2.size());
    }
};
// ---- and thread M
struct MOut {
    Bow bow;
    std::vector<int32_t> m12; int n_tri = 0;
    std::vector<int32_t> fuse_idx, fuse_dist; int n_fuse = 0;
    std::vector<double> R, t, P, chi2; std::vector<uint8_t> outlier; int its = 0, trials = 0; double lambda = 0, chi_i = 0, chi_f = 0;
    bool operator==(const MOut &o) const
    {
        return bow == o.bow && m12 == o.m12 && n_tri == o.n_tri && fuse_idx == o.fuse_idx && fuse_dist == o.fuse_dist && n_fuse == o.n_fuse &&
               !memcmp(R.data(), o.R.data(), 8 * R.size()) && !memcmp(t.data(), o.t.data(), 8 * t.size()) &&
               !memcmp(P.data(), o.P.data(), 8 * P.size()) && !memcmp(chi2.data(), o.chi2.data(), 8 * chi2.size()) && outlier == o.outlier &&
               its == o.its && trials == o.trials && !memcmp(&lambda, &o.lambda, 8) && !memcmp(&chi_i, &o.chi_i, 8) && !memcmp(&chi_f, &o.chi_f, 8);
    }
};

// ---- the shared, read-only scene
struct Scene {
    View a, b;
    int cap = 0;
    float sigma2[ORBX_MAX_LEVELS]; int n_levels = 0;
    // projection queries: view a's features looked for in view b
    std::vector<float> q_xy, q_radius_proj, q_radius_fuse, q_angle; std::vector<int32_t> q_octave; std::vector<uint8_t> q_ok, has_mp_a, has_mp_b;
    // poseOptimize: one frame
    std::vector<int32_t> pose_off; std::vector<double> pose_R, pose_t, pose_P, pose_z, pose_w;
    // local BA
    int NP = 0, NL = 0, NE = 0;
    std::vector<double> ba_R, ba_t, ba_P, ba_z, ba_w; std::vector<uint8_t> ba_fix; std::vector<int32_t> ba_ep, ba_el;
};
static const double FX = 718.856, FY = 718.856, CX = 607.19, CY = 185.22;

static void rot_y(double a, double *R) { const double c = std::cos(a), s = std::sin(a); const double r[9] = {c, 0, s, 0, 1, 0, -s, 0, c}; memcpy(R, r, sizeof r); }

static void build_scene(Scene &S, orbx_t *xh)
{
    S.cap = orbx_max_keypoints(xh, W, H);
    for (int f = 0; f < 2; ++f) {
        View &v = f ? S.b : S.a;
        v.img = make_image(f);
        v.kp.resize(S.cap); v.desc.resize((size_t)S.cap * 32);
        ORB_OK(orbx_extract(xh, v.img.data(), W, H, W, v.kp.data(), v.desc.data(), S.cap, &v.n));
        v.angle.resize(v.n);
        for (int i = 0; i < v.n; ++i) v.angle[i] = v.kp[i].angle;
    }
    float sq[ORBX_MAX_LEVELS];
    ORB_OK(orbx_tables(xh, &S.n_levels, nullptr, nullptr, sq, nullptr, nullptr, nullptr, nullptr));
    memcpy(S.sigma2, sq, sizeof sq);
    const int n = S.a.n;
    S.q_xy.resize(2 * (size_t)n); S.q_radius_proj.resize(n); S.q_radius_fuse.resize(n); S.q_angle.resize(n); S.q_octave.resize(n);
    S.q_ok.resize(n); S.has_mp_a.resize(n); S.has_mp_b.resize(S.b.n);
    for (int i = 0; i < n; ++i) {
        const orbx_kp &k = S.a.kp[i];
        S.q_xy[2 * i] = k.x - 3.f; S.q_xy[2 * i + 1] = k.y - 2.f; // where view b shows the same corner
        S.q_radius_proj[i] = 15.f * k.size;                       // th * key_points[i].size (ORBMatcher.cpp:226)
        S.q_radius_fuse[i] = 3.f * k.size;                        // th * scale_factor[predictLevel] (:555)
        S.q_angle[i] = k.angle; S.q_octave[i] = k.octave;
        S.q_ok[i] = (i % 7) != 0;
        S.has_mp_a[i] = (i % 5) == 0;
    }
    for (int j = 0; j < S.b.n; ++j) S.has_mp_b[j] = (j % 6) == 0;
    // poseOptimize: 400 points in front of a camera near the identity, pixel noise, a few gross outliers
    uint32_t s = 777;
    const int ne = 400;
    S.pose_off = {0, ne};
    S.pose_R.resize(9); rot_y(0.01, S.pose_R.data());
    S.pose_t = {0.05, -0.02, 0.03};
    for (int e = 0; e < ne; ++e) {
        const double X = -12 + 24 * unif(s), Y = -4 + 8 * unif(s), Z = 6 + 30 * unif(s);
        S.pose_P.insert(S.pose_P.end(), {X, Y, Z});
        double u = FX * X / Z + CX + (unif(s) - 0.5), v = FY * Y / Z + CY + (unif(s) - 0.5);
        if (e % 23 == 0) { u += 25; v -= 18; }
        S.pose_z.insert(S.pose_z.end(), {u, v});
        const double sz = std::pow(1.2, e % 8);
        S.pose_w.push_back(1.0 / (sz * sz));
    }
    // local BA: 8 key frames on an arc (2 fixed), 600 points, every point seen by every key frame; perturbed estimates
    S.NP = 8; S.NL = 600;
    std::vector<double> Rg((size_t)9 * S.NP), tg((size_t)3 * S.NP), Pg((size_t)3 * S.NL);
    for (int p = 0; p < S.NP; ++p) {
        rot_y(0.02 * p, &Rg[9 * p]);
        const double c[3] = {0.4 * p, 0.0, 0.05 * p}; // camera centre; t = -R c
        for (int r = 0; r < 3; ++r) tg[3 * p + r] = -(Rg[9 * p + 3 * r] * c[0] + Rg[9 * p + 3 * r + 1] * c[1] + Rg[9 * p + 3 * r + 2] * c[2]);
    }
    for (int l = 0; l < S.NL; ++l) { Pg[3 * l] = -8 + 16 * unif(s); Pg[3 * l + 1] = -3 + 6 * unif(s); Pg[3 * l + 2] = 8 + 20 * unif(s); }
    S.ba_fix.assign(S.NP, 0); S.ba_fix[0] = S.ba_fix[1] = 1;
    for (int l = 0; l < S.NL; ++l)
        for (int p = 0; p < S.NP; ++p) {
            const double *R = &Rg[9 * p], *t = &tg[3 * p], *P = &Pg[3 * l];
            const double X = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0], Y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1],
                         Z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
            double u = FX * X / Z + CX + (unif(s) - 0.5), v = FY * Y / Z + CY + (unif(s) - 0.5);
            if ((l * S.NP + p) % 97 == 0) { u += 12; v += 9; }
            S.ba_ep.push_back(p); S.ba_el.push_back(l);
            S.ba_z.insert(S.ba_z.end(), {u, v});
            const double sz = std::pow(1.2, (l + p) % 8);
            S.ba_w.push_back(1.0 / (sz * sz));
        }
    S.NE = (int)S.ba_ep.size();
    S.ba_R = Rg; S.ba_t = tg; S.ba_P = Pg;
    for (int p = 2; p < S.NP; ++p) for (int r = 0; r < 3; ++r) S.ba_t[3 * p + r] += 0.02 * (unif(s) - 0.5);
    for (double &x : S.ba_P) x += 0.05 * (unif(s) - 0.5);
}

using Clock = std::chrono::steady_clock;
static double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct THandles { orbx_t *x; orbm_t *m; orbv_t *v; };
// one iteration of the Tracking-like thread; lat (may be null) receives the four calls' host times in microseconds
static void t_iteration(const Scene &S, const THandles &h, TOut &o, double *lat)
{
    o.kp.assign(S.cap, orbx_kp{}); o.desc.assign((size_t)S.cap * 32, 0);
    Clock::time_point t0 = Clock::now();
    ORB_OK(orbx_extract(h.x, S.b.img.data(), W, H, W, o.kp.data(), o.desc.data(), S.cap, &o.n)); // Frame.cpp:20
    if (lat) lat[0] = us_since(t0);
    View cur; cur.n = o.n; cur.desc = o.desc;
    t0 = Clock::now();
    compute_bow(h.v, cur, o.bow); // Frame.cpp:168
    if (lat) lat[1] = us_since(t0);
    o.frame_mp.assign(o.n, -1);
    t0 = Clock::now();
    ORB_OK(orbm_search_by_projection_frame(h.m, 1, S.a.desc.data(), S.q_xy.data(), S.q_radius_proj.data(), S.q_octave.data(), S.q_angle.data(),
                                           S.q_ok.data(), S.a.n, o.kp.data(), o.desc.data(), o.n, W, H, o.frame_mp.data(), &o.n_proj)); // Tracking.cpp:289
    if (lat) lat[2] = us_since(t0);
    orbba_pose_problem p = {};
    p.fx = FX; p.fy = FY; p.cx = CX; p.cy = CY; p.huber_delta = (double)std::sqrt(5.991f);
    p.n_frames = 1; p.edge_off = S.pose_off.data(); p.pose_R = S.pose_R.data(); p.pose_t = S.pose_t.data(); p.points = S.pose_P.data();
    p.edge_z = S.pose_z.data(); p.edge_inv_sigma2 = S.pose_w.data();
    const int ne = S.pose_off[1];
    o.inlier.assign(ne, 0); o.chi2.assign(ne, 0.0);
    orbba_pose_result r = {};
    r.pose_R = o.R; r.pose_t = o.t; r.inlier = o.inlier.data(); r.n_inliers = &o.n_inl; r.chi2 = o.chi2.data();
    t0 = Clock::now();
    ORB_OK(orbba_pose_optimize_batch(&p, &r, -1)); // Tracking.cpp:305
    if (lat) lat[3] = us_since(t0);
}

struct MHandles { orbm_t *m; orbv_t *v; };
static void m_iteration(const Scene &S, const MHandles &h, MOut &o, double *lat)
{
    Clock::time_point t0 = Clock::now();
    compute_bow(h.v, S.a, o.bow); // LocalMapping.cpp:90
    Bow bow_b;
    compute_bow(h.v, S.b, bow_b);
    if (lat) lat[0] = us_since(t0);
    const orbm_fv fa = o.bow.fv(), fb = bow_b.fv();
    o.m12.assign(S.a.n, -1);
    t0 = Clock::now();
    ORB_OK(orbm_search_for_triangulation(h.m, 0, S.a.desc.data(), S.a.angle.data(), S.has_mp_a.data(), S.a.n, &fa, S.b.desc.data(), S.b.angle.data(),
                                         S.has_mp_b.data(), S.b.n, &fb, o.m12.data(), &o.n_tri)); // LocalMapping.cpp:168
    if (lat) lat[1] = us_since(t0);
    o.fuse_idx.assign(S.a.n, -1); o.fuse_dist.assign(S.a.n, 0);
    t0 = Clock::now();
    ORB_OK(orbm_search_fuse(h.m, S.a.desc.data(), S.q_xy.data(), S.q_radius_fuse.data(), S.q_octave.data(), S.q_ok.data(), S.a.n, S.b.kp.data(),
                            S.b.desc.data(), S.b.n, W, H, S.sigma2, S.n_levels, o.fuse_idx.data(), o.fuse_dist.data(), &o.n_fuse)); // LocalMapping.cpp:282
    if (lat) lat[2] = us_since(t0);
    orbba_problem p = {};
    p.fx = FX; p.fy = FY; p.cx = CX; p.cy = CY; p.huber_delta = (double)std::sqrt(5.991f);
    p.n_poses = S.NP; p.n_points = S.NL; p.n_edges = S.NE; p.pose_R = S.ba_R.data(); p.pose_t = S.ba_t.data(); p.pose_fixed = S.ba_fix.data();
    p.points = S.ba_P.data(); p.edge_pose = S.ba_ep.data(); p.edge_point = S.ba_el.data(); p.edge_z = S.ba_z.data(); p.edge_inv_sigma2 = S.ba_w.data();
    o.R.assign((size_t)9 * S.NP, 0); o.t.assign((size_t)3 * S.NP, 0); o.P.assign((size_t)3 * S.NL, 0); o.chi2.assign(S.NE, 0); o.outlier.assign(S.NE, 0);
    orbba_lm_result r = {};
    r.pose_R = o.R.data(); r.pose_t = o.t.data(); r.points = o.P.data(); r.chi2 = o.chi2.data();
    t0 = Clock::now();
    ORB_OK(orbba_local_bundle_adjustment(&p, &r, o.outlier.data(), -1)); // LocalMapping.cpp:45-52
    if (lat) lat[3] = us_since(t0);
    o.its = r.iterations; o.trials = r.trials; o.lambda = r.lambda; o.chi_i = r.chi2_initial; o.chi_f = r.chi2_final;
}

static void pct(std::vector<double> v, double *p50, double *p90)
{
    std::sort(v.begin(), v.end());
    *p50 = v[v.size() / 2];
    *p90 = v[std::min(v.size() - 1, v.size() * 9 / 10)];
}


// ---- device chains, one non-blocking stream per thread
template <typename T> static T *dmalloc(size_t n) { void *p = nullptr; HIP_OK(hipMalloc(&p, std::max(n, (size_t)1) * sizeof(T))); return (T *)p; }
template <typename T> static T *hmalloc(size_t n) { void *p = nullptr; HIP_OK(hipHostMalloc(&p, std::max(n, (size_t)1) * sizeof(T), hipHostMallocDefault)); return (T *)p; }
struct TDev { // thread T: frame up, extract, best / second-best against view a's descriptors
    hipStream_t s; orbx_t *x; orbm_t *m; int cap, na;
    uint8_t *h_img, *d_img, *d_desc, *d_desc_a; orbx_kp *d_kp; int32_t *d_n, *d_na, *d_bi; uint16_t *d_bd, *d_sd;
    orbx_kp *h_kp; uint8_t *h_desc; int32_t *h_n, *h_bi; uint16_t *h_bd, *h_sd;
    void init(const Scene &S, orbx_t *xh, orbm_t *mh)
    {
        HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        x = xh; m = mh; cap = S.cap; na = S.a.n;
        h_img = hmalloc<uint8_t>((size_t)W * H); memcpy(h_img, S.b.img.data(), (size_t)W * H);
        d_img = dmalloc<uint8_t>((size_t)W * H); d_kp = dmalloc<orbx_kp>(cap); d_desc = dmalloc<uint8_t>((size_t)cap * 32); d_n = dmalloc<int32_t>(1);
        d_desc_a = dmalloc<uint8_t>((size_t)na * 32); d_na = dmalloc<int32_t>(1);
        d_bi = dmalloc<int32_t>(cap); d_bd = dmalloc<uint16_t>(cap); d_sd = dmalloc<uint16_t>(cap);
        h_kp = hmalloc<orbx_kp>(cap); h_desc = hmalloc<uint8_t>((size_t)cap * 32); h_n = hmalloc<int32_t>(1);
        h_bi = hmalloc<int32_t>(cap); h_bd = hmalloc<uint16_t>(cap); h_sd = hmalloc<uint16_t>(cap);
        HIP_OK(hipMemcpy(d_desc_a, S.a.desc.data(), (size_t)na * 32, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_na, &na, 4, hipMemcpyHostToDevice));
    }
    void run()
    {
        HIP_OK(hipMemcpyAsync(d_img, h_img, (size_t)W * H, hipMemcpyHostToDevice, s));
        ORB_OK(orbx_extract_batch_device(x, d_img, 1, W, H, W, (size_t)W * H, d_kp, d_desc, cap, d_n, s));
        ORB_OK(orbm_best2_device(m, 1, d_desc, cap, d_n, cap, d_desc_a, na, d_na, na, nullptr, nullptr, d_bi, d_bd, d_sd, s));
        HIP_OK(hipMemcpyAsync(h_n, d_n, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_kp, d_kp, sizeof(orbx_kp) * (size_t)cap, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_desc, d_desc, (size_t)cap * 32, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_bi, d_bi, 4 * (size_t)cap, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_bd, d_bd, 2 * (size_t)cap, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_sd, d_sd, 2 * (size_t)cap, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    struct Out { int n; std::vector<orbx_kp> kp; std::vector<uint8_t> desc; std::vector<int32_t> bi; std::vector<uint16_t> bd, sd; };
    Out out() const
    {
        Out o; o.n = *h_n;
        o.kp.assign(h_kp, h_kp + o.n); o.desc.assign(h_desc, h_desc + (size_t)o.n * 32);
        o.bi.assign(h_bi, h_bi + o.n); o.bd.assign(h_bd, h_bd + o.n); o.sd.assign(h_sd, h_sd + o.n);
        return o;
    }
};
static bool same(const TDev::Out &a, const TDev::Out &b)
{
    return a.n == b.n && !memcmp(a.kp.data(), b.kp.data(), sizeof(orbx_kp) * (size_t)a.n) && a.desc == b.desc && a.bi == b.bi && a.bd == b.bd && a.sd == b.sd;
}
struct MDev { // thread M: both views' bag of words and the triangulation matches, records resident on the device
    hipStream_t s; orbv_t *v; orbm_t *m; int n[2], capv;
    uint8_t *d_desc[2], *d_has[2]; orbx_kp *d_kp[2]; int32_t *d_cnt[2], *d_nw[2], *d_nfv[2], *d_off[2], *d_m12, *d_res, *h_m12, *h_res;
    uint32_t *d_ids[2], *d_nodes[2], *d_idx[2]; double *d_vals[2];
    void init(const Scene &S, orbv_t *voc, orbm_t *mh)
    {
        HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        v = voc; m = mh; capv = std::max(S.a.n, S.b.n);
        for (int k = 0; k < 2; ++k) {
            const View &f = k ? S.b : S.a;
            const std::vector<uint8_t> &has = k ? S.has_mp_b : S.has_mp_a;
            n[k] = f.n;
            d_desc[k] = dmalloc<uint8_t>((size_t)capv * 32); d_kp[k] = dmalloc<orbx_kp>(capv); d_has[k] = dmalloc<uint8_t>(capv); d_cnt[k] = dmalloc<int32_t>(1);
            d_ids[k] = dmalloc<uint32_t>(capv); d_vals[k] = dmalloc<double>(capv); d_nw[k] = dmalloc<int32_t>(1); d_nodes[k] = dmalloc<uint32_t>(capv);
            d_off[k] = dmalloc<int32_t>(capv + 1); d_idx[k] = dmalloc<uint32_t>(capv); d_nfv[k] = dmalloc<int32_t>(1);
            HIP_OK(hipMemcpy(d_desc[k], f.desc.data(), (size_t)f.n * 32, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_kp[k], f.kp.data(), sizeof(orbx_kp) * (size_t)f.n, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_has[k], has.data(), f.n, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_cnt[k], &f.n, 4, hipMemcpyHostToDevice));
        }
        d_m12 = dmalloc<int32_t>(capv); d_res = dmalloc<int32_t>(8); h_m12 = hmalloc<int32_t>(capv); h_res = hmalloc<int32_t>(8);
    }
    void run()
    {
        for (int k = 0; k < 2; ++k)
            ORB_OK(orbv_transform_device(v, 1, d_desc[k], d_cnt[k], capv, /*levelsup*/ 1, d_ids[k], d_vals[k], d_nw[k], d_nodes[k], d_off[k], d_idx[k],
                                         d_nfv[k], s));
        ORB_OK(orbm_search_for_triangulation_device(m, 0, d_desc[0], d_kp[0], d_has[0], n[0], d_nodes[0], d_off[0], d_idx[0], d_nfv[0], d_desc[1],
                                                    d_kp[1], d_has[1], n[1], d_nodes[1], d_off[1], d_idx[1], d_nfv[1], d_m12, d_res, s));
        HIP_OK(hipMemcpyAsync(h_m12, d_m12, 4 * (size_t)n[0], hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(h_res, d_res, 32, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
    }
};

int main(int argc, char **argv)
{
    int iters = 12, latency = 0;
    bool own_voc = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--latency") && i + 1 < argc) latency = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--own-voc")) own_voc = true;
    }
    orbx_cfg cfg = {NF, 1.2f, 8, 20, 7, W, H, 1, 0, -1};
    orbx_t *xh = nullptr;
    ORB_OK(orbx_create(&cfg, &xh));
    orbm_t *mt = nullptr, *mm = nullptr;
    ORB_OK(orbm_create(-1, &mt));
    ORB_OK(orbm_create(-1, &mm));
    orbv_t *voc = make_vocabulary(8, 2), *voc_m = own_voc ? make_vocabulary(8, 2) : voc;
    Scene S;
    build_scene(S, xh);
    std::printf("views: %d / %d key points; local BA %d poses x %d points, %d edges\n", S.a.n, S.b.n, S.NP, S.NL, S.NE);
    if (S.a.n < 1500 || S.b.n < 1500) { std::printf("too few key points\n"); return 2; }
    const THandles th = {xh, mt, voc};
    const MHandles mh = {mm, voc_m};

    // the single-thread answers (the second pass warms every scratch and shows the calls repeat themselves)
    TOut t_ref, t_again;
    MOut m_ref, m_again;
    t_iteration(S, th, t_ref, nullptr);
    m_iteration(S, mh, m_ref, nullptr);
    t_iteration(S, th, t_again, nullptr);
    m_iteration(S, mh, m_again, nullptr);
    if (!(t_ref == t_again) || !(m_ref == m_again)) { std::printf("single-thread calls do not repeat themselves\n"); return 3; }
    std::printf("single thread: projection matches %d, pose inliers %d; triangulation matches %d, fuse hits %d, BA %d iterations chi2 %.3f -> %.3f\n",
                t_ref.n_proj, t_ref.n_inl, m_ref.n_tri, m_ref.n_fuse, m_ref.its, m_ref.chi_i, m_ref.chi_f);
    if (t_ref.n_proj < 100 || t_ref.n_inl < 100 || m_ref.n_tri < 20 || m_ref.n_fuse < 100 || !(m_ref.chi_f < m_ref.chi_i)) {
        std::printf("the scene does not exercise the calls\n");
        return 4;
    }

    if (latency > 0) {
        static const char *tn[4] = {"orbx_extract", "orbv_transform", "orbm_search_by_projection_frame", "orbba_pose_optimize_batch"};
        static const char *mn[4] = {"orbv_transform x2", "orbm_search_for_triangulation", "orbm_search_fuse", "orbba_local_bundle_adjustment"};
        std::vector<double> alone[4], with_m[4], m_alone[4], m_with[4];
        TOut o; MOut mo;
        for (int i = 0; i < latency; ++i) { double l[4]; t_iteration(S, th, o, l); for (int k = 0; k < 4; ++k) alone[k].push_back(l[k]); }
        for (int i = 0; i < std::max(latency / 8, 8); ++i) { double l[4]; m_iteration(S, mh, mo, l); for (int k = 0; k < 4; ++k) m_alone[k].push_back(l[k]); }
        std::atomic<bool> stop{false}, started{false};
        std::thread m([&] {
            MOut lo;
            while (!stop.load()) { double l[4]; m_iteration(S, mh, lo, l); started.store(true); for (int k = 0; k < 4; ++k) m_with[k].push_back(l[k]); }
        });
        while (!started.load()) std::this_thread::yield();
        for (int i = 0; i < latency; ++i) { double l[4]; t_iteration(S, th, o, l); for (int k = 0; k < 4; ++k) with_m[k].push_back(l[k]); }
        stop.store(true);
        m.join();
        std::printf("thread T, %d iterations, microseconds per call (host clock around the C call)\n", latency);
        std::printf("%-34s %10s %10s %12s %12s %8s\n", "call", "alone p50", "alone p90", "with M p50", "with M p90", "p50 x");
        for (int k = 0; k < 4; ++k) {
            double a50, a90, w50, w90;
            pct(alone[k], &a50, &a90); pct(with_m[k], &w50, &w90);
            std::printf("%-34s %10.1f %10.1f %12.1f %12.1f %8.2f\n", tn[k], a50, a90, w50, w90, w50 / a50);
        }
        std::printf("thread M, %zu iterations alone, %zu beside T\n", m_alone[0].size(), m_with[0].size());
        std::printf("%-34s %10s %10s %12s %12s %8s\n", "call", "alone p50", "alone p90", "with T p50", "with T p90", "p50 x");
        for (int k = 0; k < 4; ++k) {
            double a50, a90, w50, w90;
            pct(m_alone[k], &a50, &a90); pct(m_with[k], &w50, &w90);
            std::printf("%-34s %10.1f %10.1f %12.1f %12.1f %8.2f\n", mn[k], a50, a90, w50, w90, w50 / a50);
        }
        return 0;
    }

    // both threads at once, every iteration against the single-thread answers
    std::atomic<int> bad_t{0}, bad_m{0};
    std::thread tt([&] { TOut o; for (int i = 0; i < iters; ++i) { t_iteration(S, th, o, nullptr); if (!(o == t_ref)) ++bad_t; } });
    std::thread tm([&] { MOut o; for (int i = 0; i < iters; ++i) { m_iteration(S, mh, o, nullptr); if (!(o == m_ref)) ++bad_m; } });
    tt.join();
    tm.join();
    std::printf("two threads, %d iterations each: T mismatches %d, M mismatches %d\n", iters, bad_t.load(), bad_m.load());
    if (bad_t.load() || bad_m.load()) return 1;

    // ---- device chains: each thread on a non-blocking stream of its own (M has a vocabulary handle of its own: the device transform uses
    // the handle's one scratch)
    orbv_t *voc_dev = make_vocabulary(8, 2);
    TDev td; MDev md;
    td.init(S, xh, mt);
    md.init(S, voc_dev, mm);
    td.run();
    const TDev::Out td_ref = td.out();
    md.run();
    const std::vector<int32_t> m12_dev(md.h_m12, md.h_m12 + S.a.n);
    const int tri_dev = md.h_res[0];
    // ... against the host entry points: the frame extracted on the device is the frame orbx_extract returns, its best / second best are
    // orbm_best2's, and the device triangulation search is the host one
    {
        std::vector<int32_t> bi(t_ref.n); std::vector<uint16_t> bd(t_ref.n), sd(t_ref.n);
        ORB_OK(orbm_best2(mt, t_ref.desc.data(), t_ref.n, S.a.desc.data(), S.a.n, nullptr, nullptr, bi.data(), bd.data(), sd.data()));
        const bool ext = td_ref.n == t_ref.n && !memcmp(td_ref.kp.data(), t_ref.kp.data(), sizeof(orbx_kp) * (size_t)t_ref.n) &&
                         !memcmp(td_ref.desc.data(), t_ref.desc.data(), 32 * (size_t)t_ref.n);
        if (!ext || bi != td_ref.bi || bd != td_ref.bd || sd != td_ref.sd || md.h_res[1] != 0 || tri_dev != m_ref.n_tri || m12_dev != m_ref.m12) {
            std::printf("device chains differ from the host entry points (extract %d, triangulation %d against %d, flag %d)\n", (int)ext, tri_dev, m_ref.n_tri,
                        md.h_res[1]);
            return 5;
        }
    }
    std::atomic<int> bad_td{0}, bad_md{0};
    std::thread dt([&] { for (int i = 0; i < iters; ++i) { td.run(); if (!same(td.out(), td_ref)) ++bad_td; } });
    std::thread dm([&] {
        for (int i = 0; i < iters; ++i) {
            md.run();
            if (md.h_res[0] != tri_dev || md.h_res[1] != 0 || memcmp(md.h_m12, m12_dev.data(), 4 * m12_dev.size())) ++bad_md;
        }
    });
    dt.join();
    dm.join();
    std::printf("device chains on two streams, %d iterations each: T mismatches %d, M mismatches %d (best-2 of %d x %d, %d triangulation matches)\n", iters,
                bad_td.load(), bad_md.load(), td_ref.n, S.a.n, tri_dev);
    if (bad_td.load() || bad_md.load()) return 6;
    orbv_destroy(voc_dev);
    orbx_destroy(xh); orbm_destroy(mt); orbm_destroy(mm);
    if (voc_m != voc) orbv_destroy(voc_m);
    orbv_destroy(voc);
    std::printf("two threads ok\n");
    return 0;
}
