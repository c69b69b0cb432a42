// Frame post-processing on the device: key-point size scaling, undistortion and the 40-px grid index
// (reference modules/BasicObject/Frame.cpp:24-51; C ABI in include/orbf.h).
//
// One 256-thread workgroup per frame.  The grid is a counting sort whose order inside a cell must be the ascending
// key-point index (the reference push_back()s in index order, Frame.cpp:45-50): cells are filled with LDS atomics in
// arbitrary order first, then every key point finds its rank by counting the smaller indices of its own cell.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/orbf.h"
#include "orb_math.h"

int orbx_set_error(int code, const std::string &msg);

#define F_TRY(expr)                                                                                              \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return orbx_set_error(ORBX_E_NO_DEVICE, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

struct OrbfCam {
    int width, height, cols, rows;
    double fx, fy, cx, cy, ifx, ify;
    double k[ORBF_MAX_DIST];
    int undistort; // already folded: RAD_TAN and dist[0] != 0
    const float *size_scale;
};

struct orbf_ctx {
    int device = 0;
    OrbfCam cam{};
    float *d_scale = nullptr;
    hipStream_t stream = nullptr;
    bool null_pending = false; // a device call was enqueued on stream 0 (NULL): the next host-pointer call and destroy wait for it
    int32_t *d_cell_of = nullptr, *d_tmp = nullptr;
    size_t scratch_items = 0;
    // host-convenience staging
    orbx_kp *d_raw = nullptr, *d_un = nullptr;
    int32_t *d_start = nullptr, *d_items = nullptr, *d_n = nullptr;
    size_t stage_cap = 0;
};

// cv::undistortPoints(src, dst, K, dist, noArray(), K) for one point: OpenCV 4.2 cvUndistortPointsInternal with
// TermCriteria(MAX_ITER, 5, 0.01), i.e. exactly five iterations, all in double, result rounded to float.
__device__ __forceinline__ void undistort_point(const OrbfCam &c, float uf, float vf, float *xu, float *yu)
{
    const double u = (double)uf, v = (double)vf;
    double x = ORB_DMUL(ORB_DSUB(u, c.cx), c.ifx), y = ORB_DMUL(ORB_DSUB(v, c.cy), c.ify);
    const double x0 = x, y0 = y;
    const double *k = c.k;
    for (int j = 0; j < 5; ++j) {
        const double r2 = ORB_DADD(ORB_DMUL(x, x), ORB_DMUL(y, y));
        const double num = ORB_DADD(1., ORB_DMUL(ORB_DADD(ORB_DMUL(ORB_DADD(ORB_DMUL(k[7], r2), k[6]), r2), k[5]), r2));
        const double den = ORB_DADD(1., ORB_DMUL(ORB_DADD(ORB_DMUL(ORB_DADD(ORB_DMUL(k[4], r2), k[1]), r2), k[0]), r2));
        const double icdist = ORB_DDIV(num, den);
        if (icdist < 0) {
            x = x0;
            y = y0;
            break;
        }
        // 2*k2*x*y + k3*(r2 + 2*x*x) + k8*r2 + k9*r2*r2, left to right as C evaluates it
        double dx = ORB_DMUL(ORB_DMUL(ORB_DMUL(2., k[2]), x), y);
        dx = ORB_DADD(dx, ORB_DMUL(k[3], ORB_DADD(r2, ORB_DMUL(ORB_DMUL(2., x), x))));
        dx = ORB_DADD(dx, ORB_DMUL(k[8], r2));
        dx = ORB_DADD(dx, ORB_DMUL(ORB_DMUL(k[9], r2), r2));
        double dy = ORB_DMUL(k[2], ORB_DADD(r2, ORB_DMUL(ORB_DMUL(2., y), y)));
        dy = ORB_DADD(dy, ORB_DMUL(ORB_DMUL(ORB_DMUL(2., k[3]), x), y));
        dy = ORB_DADD(dy, ORB_DMUL(k[10], r2));
        dy = ORB_DADD(dy, ORB_DMUL(ORB_DMUL(k[11], r2), r2));
        x = ORB_DMUL(ORB_DSUB(x0, dx), icdist);
        y = ORB_DMUL(ORB_DSUB(y0, dy), icdist);
    }
    // P = K, R = I: xx = fx*x + 0*y + cx, ww = 1/(0*x + 0*y + 1) = 1 -- the zero terms do not change a finite value
    *xu = (float)ORB_DADD(ORB_DMUL(c.fx, x), c.cx);
    *yu = (float)ORB_DADD(ORB_DMUL(c.fy, y), c.cy);
}

__global__ __launch_bounds__(256) void k_frame_post(OrbfCam cam, orbx_kp *__restrict__ kp_raw,
                                                    const int32_t *__restrict__ n_kp, int cap,
                                                    orbx_kp *__restrict__ kp_un, int32_t *__restrict__ cell_start,
                                                    int32_t *__restrict__ cell_items, int32_t *__restrict__ cell_of,
                                                    int32_t *__restrict__ tmp)
{
    extern __shared__ int32_t lds[]; // start[nc + 1], cursor[nc]
    __shared__ int32_t wave_sum[4];
    const int f = blockIdx.x, tid = threadIdx.x, nc = cam.cols * cam.rows;
    int32_t *start = lds, *cursor = lds + nc + 1;
    const int n = min(n_kp[f], cap);
    kp_raw += (size_t)f * cap;
    kp_un += (size_t)f * cap;
    cell_items += (size_t)f * cap;
    cell_of += (size_t)f * cap;
    tmp += (size_t)f * cap;
    cell_start += (size_t)f * (nc + 1);
    for (int c = tid; c < nc; c += 256) cursor[c] = 0;
    __syncthreads();
    // 1. size scaling, undistortion, cell histogram
    for (int i = tid; i < n; i += 256) {
        const float *src = reinterpret_cast<const float *>(kp_raw + i);
        float v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = src[k];
        if (cam.size_scale) { // Fisheye::uncertainty: scale_mat.at<float>(p.y, p.x), float -> int truncation
            v[2] = ORB_FMUL(v[2], cam.size_scale[(size_t)(int)v[1] * cam.width + (int)v[0]]);
            reinterpret_cast<float *>(kp_raw + i)[2] = v[2];
        }
        if (cam.undistort) undistort_point(cam, v[0], v[1], &v[0], &v[1]);
        float *dst = reinterpret_cast<float *>(kp_un + i);
#pragma unroll
        for (int k = 0; k < 7; ++k) dst[k] = v[k];
        const int x = orb_floor_f(v[0]), y = orb_floor_f(v[1]); // Frame::PosInGrid (Frame.cpp:89-94)
        int c = -1;
        if (x >= 0 && x < cam.width && y >= 0 && y < cam.height) {
            c = (x / ORBF_GRID_SIZE) * cam.rows + (y / ORBF_GRID_SIZE);
            atomicAdd(&cursor[c], 1);
        }
        cell_of[i] = c;
    }
    __syncthreads();
    // 2. exclusive scan of the histogram: thread t owns cells [t*per, (t+1)*per)
    const int per = (nc + 255) / 256, c0 = min(tid * per, nc), c1 = min(c0 + per, nc);
    int local = 0;
    for (int c = c0; c < c1; ++c) local += cursor[c];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - local;
    for (int w = 0; w < (tid >> 6); ++w) base += wave_sum[w];
    for (int c = c0; c < c1; ++c) {
        const int cnt = cursor[c];
        start[c] = base;
        cell_start[c] = base;
        cursor[c] = 0;
        base += cnt;
    }
    if (tid == 255) {
        start[nc] = base; // thread 255 owns the tail (or an empty range at the very end)
        cell_start[nc] = base;
    }
    __syncthreads();
    // 3. unordered fill
    for (int i = tid; i < n; i += 256) {
        const int c = cell_of[i];
        if (c >= 0) tmp[start[c] + atomicAdd(&cursor[c], 1)] = i;
    }
    __syncthreads();
    // 4. rank inside the cell = number of smaller indices
    for (int i = tid; i < n; i += 256) {
        const int c = cell_of[i];
        if (c < 0) continue;
        const int b = start[c], e = start[c + 1];
        int rank = 0;
        for (int t = b; t < e; ++t) rank += tmp[t] < i;
        cell_items[b + rank] = i;
    }
}

static int grid_dim(int v) { return v % ORBF_GRID_SIZE == 0 ? v / ORBF_GRID_SIZE : v / ORBF_GRID_SIZE + 1; }

extern "C" int orbf_create(const orbf_camera *cam, int device, orbf_t **out)
{
    if (!cam || !out) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (cam->width <= 0 || cam->height <= 0 || cam->n_dist < 0 || cam->n_dist > ORBF_MAX_DIST)
        return orbx_set_error(ORBX_E_ARG, "bad camera: size must be positive and n_dist in 0..12");
    if (cam->fx == 0.f || cam->fy == 0.f) return orbx_set_error(ORBX_E_ARG, "bad camera: zero focal length");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return orbx_set_error(ORBX_E_ARG, "device index out of range");
    F_TRY(hipSetDevice(device));
    orbf_ctx *c = new orbf_ctx();
    c->device = device;
    OrbfCam &k = c->cam;
    k.width = cam->width;
    k.height = cam->height;
    k.cols = grid_dim(cam->width);
    k.rows = grid_dim(cam->height);
    k.fx = cam->fx; k.fy = cam->fy; k.cx = cam->cx; k.cy = cam->cy;
    k.ifx = 1. / k.fx;
    k.ify = 1. / k.fy;
    for (int i = 0; i < ORBF_MAX_DIST; ++i) k.k[i] = i < cam->n_dist ? (double)cam->dist[i] : 0.;
    k.undistort = cam->undistort && cam->n_dist > 0 && cam->dist[0] != 0.f; // Pinhole.cpp:62
    if ((size_t)(2 * k.cols * k.rows + 1) * 4 > 60 * 1024) {
        delete c;
        return orbx_set_error(ORBX_E_ARG, "image too large for the grid kernel (more than 7679 cells)");
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); // host-pointer calls only (include/orbx.h, "Streams")
    if (e == hipSuccess && cam->size_scale) {
        const size_t bytes = (size_t)cam->width * cam->height * 4;
        e = hipMalloc(&c->d_scale, bytes);
        if (e == hipSuccess) e = hipMemcpy(c->d_scale, cam->size_scale, bytes, hipMemcpyHostToDevice);
        k.size_scale = c->d_scale;
    }
    if (e != hipSuccess) {
        orbf_destroy(c);
        return orbx_set_error(ORBX_E_NO_DEVICE, std::string("orbf_create: ") + hipGetErrorString(e));
    }
    *out = c;
    return ORBX_OK;
}

extern "C" void orbf_destroy(orbf_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->null_pending) (void)hipStreamSynchronize((hipStream_t)0);
    for (void *p : {(void *)c->d_scale, (void *)c->d_cell_of, (void *)c->d_tmp, (void *)c->d_raw, (void *)c->d_un,
                    (void *)c->d_start, (void *)c->d_items, (void *)c->d_n})
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbf_grid_dims(const orbf_t *c, int *cols, int *rows)
{
    if (!c || !cols || !rows) return orbx_set_error(ORBX_E_ARG, "null argument");
    *cols = c->cam.cols;
    *rows = c->cam.rows;
    return ORBX_OK;
}

extern "C" int orbf_frame_post_device(orbf_t *c, int n_frames, orbx_kp *d_kp_raw, const int32_t *d_n, int cap,
                                      orbx_kp *d_kp_un, int32_t *d_cell_start, int32_t *d_cell_items, void *stream)
{
    if (!c || !d_kp_raw || !d_n || !d_kp_un || !d_cell_start || !d_cell_items)
        return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n_frames < 0 || cap <= 0) return orbx_set_error(ORBX_E_ARG, "n_frames must be >= 0 and cap positive");
    if (n_frames == 0) return ORBX_OK;
    F_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream; // NULL is stream 0 itself (include/orbx.h, "Streams")
    if (!stream) c->null_pending = true;
    const size_t need = (size_t)n_frames * cap;
    if (need > c->scratch_items) { // scratch grows geometrically; nothing may be in flight on the old one
        F_TRY(hipDeviceSynchronize());
        if (c->d_cell_of) (void)hipFree(c->d_cell_of);
        if (c->d_tmp) (void)hipFree(c->d_tmp);
        c->d_cell_of = c->d_tmp = nullptr;
        c->scratch_items = 0;
        const size_t grow = need + need / 2;
        F_TRY(hipMalloc(&c->d_cell_of, grow * 4));
        F_TRY(hipMalloc(&c->d_tmp, grow * 4));
        c->scratch_items = grow;
    }
    const int nc = c->cam.cols * c->cam.rows;
    hipLaunchKernelGGL(k_frame_post, dim3(n_frames), dim3(256), (size_t)(2 * nc + 1) * 4, s, c->cam, d_kp_raw, d_n, cap,
                       d_kp_un, d_cell_start, d_cell_items, c->d_cell_of, c->d_tmp);
    F_TRY(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbf_frame_post(orbf_t *c, orbx_kp *kp_raw, int n, orbx_kp *kp_un, int32_t *cell_start,
                               int32_t *cell_items)
{
    if (!c || !kp_raw || !kp_un || !cell_start || !cell_items) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (n < 0) return orbx_set_error(ORBX_E_ARG, "negative key-point count");
    F_TRY(hipSetDevice(c->device));
    // the handle's stream is non-blocking: NULL-stream device calls of this handle still in flight use the same scratch
    if (c->null_pending) { F_TRY(hipStreamSynchronize((hipStream_t)0)); c->null_pending = false; }
    const int nc = c->cam.cols * c->cam.rows, cap = n > 0 ? n : 1;
    if ((size_t)cap > c->stage_cap) {
        F_TRY(hipStreamSynchronize(c->stream));
        for (void **p : {(void **)&c->d_raw, (void **)&c->d_un, (void **)&c->d_start, (void **)&c->d_items, (void **)&c->d_n})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        c->stage_cap = 0;
        const size_t grow = (size_t)cap + cap / 2 + 64;
        F_TRY(hipMalloc(&c->d_raw, grow * sizeof(orbx_kp)));
        F_TRY(hipMalloc(&c->d_un, grow * sizeof(orbx_kp)));
        F_TRY(hipMalloc(&c->d_items, grow * 4));
        F_TRY(hipMalloc(&c->d_start, (size_t)(nc + 1) * 4));
        F_TRY(hipMalloc(&c->d_n, 4));
        c->stage_cap = grow;
    }
    hipStream_t s = c->stream;
    const int32_t n32 = n;
    F_TRY(hipMemcpyAsync(c->d_raw, kp_raw, (size_t)n * sizeof(orbx_kp), hipMemcpyHostToDevice, s));
    F_TRY(hipMemcpyAsync(c->d_n, &n32, 4, hipMemcpyHostToDevice, s));
    int rc = orbf_frame_post_device(c, 1, c->d_raw, c->d_n, cap, c->d_un, c->d_start, c->d_items, s);
    if (rc) return rc;
    F_TRY(hipMemcpyAsync(kp_raw, c->d_raw, (size_t)n * sizeof(orbx_kp), hipMemcpyDeviceToHost, s));
    F_TRY(hipMemcpyAsync(kp_un, c->d_un, (size_t)n * sizeof(orbx_kp), hipMemcpyDeviceToHost, s));
    F_TRY(hipMemcpyAsync(cell_start, c->d_start, (size_t)(nc + 1) * 4, hipMemcpyDeviceToHost, s));
    F_TRY(hipMemcpyAsync(cell_items, c->d_items, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    F_TRY(hipStreamSynchronize(s));
    return ORBX_OK;
}
