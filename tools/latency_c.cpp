// Host-to-host latency of orbx_extract as a C++ caller sees it (no Python in the loop).
//   usage: latency_c frame.bin W H [n_features] [reps] [pinned] [which=value ...]     (frame.bin: W*H bytes, e.g. written by
//   tools/latency_c.sh; which=value: orbx_set_variant(handle, which, value), the ORBX_VAR_* numbers of include/orbx.h)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include "../include/orbx.h"

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s frame.bin W H [n_features] [reps]\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = argc > 4 ? atoi(argv[4]) : 2000, reps = argc > 5 ? atoi(argv[5]) : 200;
    std::vector<uint8_t> img((size_t)w * h);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    orbx_cfg cfg = {};
    cfg.n_features = nf; cfg.scale_factor = 1.2f; cfg.n_levels = 8; cfg.ini_th_fast = 20; cfg.min_th_fast = 7;
    cfg.max_width = w; cfg.max_height = h; cfg.max_batch = 1; cfg.device = 0;
    orbx_t *ex = nullptr;
    if (orbx_create(&cfg, &ex)) { fprintf(stderr, "orbx_create: %s\n", orbx_last_error()); return 1; }
    bool pinned = false;
    for (int a = 6; a < argc; ++a)
        if (std::string(argv[a]) == "pinned") { pinned = true; for (int b = a; b + 1 < argc; ++b) argv[b] = argv[b + 1]; --argc; break; }
    if (pinned && orbx_host_register(img.data(), img.size())) { fprintf(stderr, "orbx_host_register: %s\n", orbx_last_error()); return 1; }
    for (int a = 6; a < argc; ++a) {
        int which = 0, value = 0;
        if (sscanf(argv[a], "%d=%d", &which, &value) != 2 || orbx_set_variant(ex, which, value)) { fprintf(stderr, "bad switch %s: %s\n", argv[a], orbx_last_error()); return 2; }
    }
    const int cap = orbx_max_keypoints(ex, w, h);
    std::vector<orbx_kp> kp(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    for (int i = 0; i < 20; ++i)
        if (orbx_extract(ex, img.data(), w, h, w, kp.data(), desc.data(), cap, &n)) { fprintf(stderr, "orbx_extract: %s\n", orbx_last_error()); return 1; }
    std::vector<double> us(reps);
    for (int i = 0; i < reps; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        orbx_extract(ex, img.data(), w, h, w, kp.data(), desc.data(), cap, &n);
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(us.begin(), us.end());
    for (int a = 6; a < argc; ++a) printf("[%s] ", argv[a]);
    if (pinned) printf("[frame in page-locked memory] ");
    printf("%dx%d nf=%d: orbx_extract host-to-host median %.1f us  p10 %.1f  p90 %.1f  (%d key points, %d calls)\n", w, h, nf,
           us[reps / 2], us[reps / 10], us[reps * 9 / 10], n, reps);
    if (pinned) orbx_host_unregister(img.data());
    orbx_destroy(ex);
    return 0;
}
