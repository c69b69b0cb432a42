// Golden-vector dumper for a machine that HAS OpenCV 4.2 (this container does not: SURVEY.md 8c).
//
// The oracle (oracle/orb_ref.c) restates OpenCV's 8-bit resize / FAST / GaussianBlur / fastAtan2 from their published
// algorithms; nothing in this repository can check that restatement against OpenCV itself.  This program produces the
// missing evidence: it runs the OpenCV calls the reference makes (ORBExtractor.cpp:565 resize, :601/:605 FAST, :528
// GaussianBlur, :41 fastAtan2, :56-62 cosf/sinf/cvRound) on seeded images and writes inputs and outputs as .npy files.
// Copy the output directory to tests/golden/opencv/ and tests/test_opencv_golden.py compares the oracle with every
// file; until then that test is skipped and parity against the original stays "unpinned".
//
//   g++ -O2 -std=c++17 tools/dump_opencv_golden.cpp -o dump_opencv_golden $(pkg-config --cflags --libs opencv4)
//   ./dump_opencv_golden tests/golden/opencv
//
// NOT compiled in this repository's build (no OpenCV here), and it links nothing from the reference.
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

static void write_npy(const std::string &path, const char *descr, const std::vector<int> &shape, const void *data,
                      size_t bytes)
{
    std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    dict += "), }";
    while ((10 + dict.size() + 1) % 64) dict += ' ';
    dict += '\n';
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    const uint16_t hl = (uint16_t)dict.size();
    fwrite(magic, 1, 8, f);
    fwrite(&hl, 2, 1, f);
    fwrite(dict.data(), 1, dict.size(), f);
    fwrite(data, 1, bytes, f);
    fclose(f);
}

static void write_u8(const std::string &path, const cv::Mat &m)
{
    cv::Mat c = m.isContinuous() ? m : m.clone();
    write_npy(path, "|u1", {c.rows, c.cols}, c.data, (size_t)c.rows * c.cols);
}

// Seeded test image: blocks, discs and noise so that every FAST threshold and both polarities occur.
static cv::Mat make_image(int w, int h, uint32_t seed)
{
    cv::Mat img(h, w, CV_8UC1);
    uint32_t s = seed;
    auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img.at<uint8_t>(y, x) = (uint8_t)(96 + ((x / 37 + y / 29) % 3) * 20 + rnd() % 5);
    for (int k = 0; k < w * h / 900; ++k) {
        const int cx = rnd() % w, cy = rnd() % h, r = 3 + rnd() % 24, v = rnd() % 256, disc = rnd() & 1;
        for (int y = std::max(0, cy - r); y < std::min(h, cy + r); ++y)
            for (int x = std::max(0, cx - r); x < std::min(w, cx + r); ++x)
                if (!disc || (x - cx) * (x - cx) + (y - cy) * (y - cy) < r * r) img.at<uint8_t>(y, x) = (uint8_t)v;
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = img.at<uint8_t>(y, x) + (int)(rnd() % 7) - 3;
            img.at<uint8_t>(y, x) = (uint8_t)std::min(255, std::max(0, v));
        }
    return img;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "opencv_golden";
    printf("OpenCV %s -> %s/\n", CV_VERSION, dir.c_str());
    const int sizes[4][2] = {{752, 480}, {1242, 375}, {321, 243}, {97, 61}};
    for (int k = 0; k < 4; ++k) {
        const int w = sizes[k][0], h = sizes[k][1];
        const cv::Mat img = make_image(w, h, 20261004u + k);
        const std::string tag = dir + "/img" + std::to_string(k);
        write_u8(tag + "_image.npy", img);
        // ORBExtractor.cpp:559-570: level sizes cvRound(w * inv_scale), chained from the previous level
        cv::Mat prev = img;
        float scale = 1.f;
        for (int l = 1; l < 8; ++l) {
            scale *= 1.2f;
            const float inv = 1.f / scale;
            const cv::Size sz(cvRound((float)w * inv), cvRound((float)h * inv));
            if (sz.width < 40 || sz.height < 40) break;
            cv::Mat dst;
            cv::resize(prev, dst, sz, 0, 0, cv::INTER_LINEAR);
            write_u8(tag + "_pyr" + std::to_string(l) + ".npy", dst);
            prev = dst;
        }
        // ORBExtractor.cpp:527-528
        cv::Mat blur;
        cv::GaussianBlur(img, blur, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
        write_u8(tag + "_blur.npy", blur);
        // ORBExtractor.cpp:601/:605 on the whole image: rows of (x, y, response)
        for (int th : {20, 7}) {
            std::vector<cv::KeyPoint> kps;
            cv::FAST(img, kps, th, true);
            std::vector<int32_t> rows;
            for (const auto &kp : kps) {
                rows.push_back((int32_t)kp.pt.x);
                rows.push_back((int32_t)kp.pt.y);
                rows.push_back((int32_t)kp.response);
            }
            write_npy(tag + "_fast" + std::to_string(th) + ".npy", "<i4", {(int)kps.size(), 3}, rows.data(), rows.size() * 4);
        }
    }
    // fastAtan2 (ORBExtractor.cpp:41) on the integer moments' range, and on a dense small grid
    {
        std::vector<float> in, out;
        uint32_t s = 7;
        auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return (int)(s >> 8); };
        for (int y = -40; y <= 40; ++y)
            for (int x = -40; x <= 40; ++x) { in.push_back((float)y); in.push_back((float)x); }
        for (int i = 0; i < 20000; ++i) { in.push_back((float)(rnd() % 400001 - 200000)); in.push_back((float)(rnd() % 400001 - 200000)); }
        for (size_t i = 0; i < in.size(); i += 2) out.push_back(cv::fastAtan2(in[i], in[i + 1]));
        write_npy(dir + "/atan2_in.npy", "<f4", {(int)in.size() / 2, 2}, in.data(), in.size() * 4);
        write_npy(dir + "/atan2_out.npy", "<f4", {(int)out.size()}, out.data(), out.size() * 4);
    }
    // cosf/sinf of angle*pi/180 as computeOrbDescriptor evaluates them (ORBExtractor.cpp:56-57), and cvRound ties
    {
        std::vector<float> ang, cs;
        for (int i = 0; i < 36000; ++i) ang.push_back((float)i * 0.01f);
        const float factor = (float)(CV_PI / 180.);
        // columns: cosf, sinf (what `cos(float)` resolves to with <cmath>'s overloads), then the double routines rounded
        for (float a : ang) {
            const float r = a * factor;
            cs.push_back(std::cos(r));
            cs.push_back(std::sin(r));
            cs.push_back((float)::cos((double)r));
            cs.push_back((float)::sin((double)r));
        }
        write_npy(dir + "/sincos_in.npy", "<f4", {(int)ang.size()}, ang.data(), ang.size() * 4);
        write_npy(dir + "/sincos_out.npy", "<f4", {(int)ang.size(), 4}, cs.data(), cs.size() * 4);
        std::vector<float> rin;
        std::vector<int32_t> rout;
        for (int i = -64; i <= 64; ++i) for (float f : {0.f, 0.25f, 0.5f, 0.75f}) rin.push_back((float)i + f);
        for (float v : rin) rout.push_back(cvRound(v));
        write_npy(dir + "/round_in.npy", "<f4", {(int)rin.size()}, rin.data(), rin.size() * 4);
        write_npy(dir + "/round_out.npy", "<i4", {(int)rout.size()}, rout.data(), rout.size() * 4);
    }
    return 0;
}
