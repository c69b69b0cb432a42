// Optional second kernel set: per-edge reprojection residual + analytic Jacobians and the block J^T W J
// reduction of local bundle adjustment, in double precision (see include/orbba.h for the reference
// lines replaced).  Two kernels:
//   k_ba_edges        thread = edge: residual, chi2, Huber weight, J_point (2x3), J_pose (2x6); writes H_lp
//                     (3x6) and the edge's contribution to its pose block (21 upper-triangular + 6) and to
//                     its point block (6 + 3) into edge-major scratch;
//   k_ba_reduce_pose  workgroup = pose: sums its edges' contributions in a fixed order (CSR built on the
//                     host), wave shuffles + LDS, so results are reproducible run to run;
//   k_ba_reduce_point thread = point: sums its (contiguous) edges.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/orbba.h"
#include "../../include/orbx.h"

int orbx_set_error(int code, const std::string &msg);
hipError_t orbx_lds_opt_in(const void *kernel, size_t bytes); // orbx_api.hip: dynamic LDS above 64 KB, per kernel and per device
#define B_TRY(expr)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return orbx_set_error(ORBX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// kernel-choice switches of the BA entry points (include/orbba.h: orbba_set_variant); the entry points take no handle, so the
// switches are per process and read per call
static std::atomic<int> g_ba_chol{0};        // ORBBA_VAR_CHOL: 0 = reduced system in LDS when it fits, 1 = global-memory kernel
static std::atomic<int> g_ba_pose_lds{3000}; // ORBBA_VAR_POSE_LDS: edges of a frame k_pose_optimize stages in LDS (0 .. 3000)
extern "C" int orbba_set_variant(int which, int value)
{
    if (which == ORBBA_VAR_CHOL && (value == 0 || value == 1)) { g_ba_chol.store(value); return ORBX_OK; }
    if (which == ORBBA_VAR_POSE_LDS && value >= 0 && value <= 3000) { g_ba_pose_lds.store(value); return ORBX_OK; }
    return orbx_set_error(ORBX_E_ARG, "unknown BA variant switch or value out of range");
}

struct BaCam { double fx, fy, cx, cy, delta; int model; double k[4]; };
static BaCam make_cam(double fx, double fy, double cx, double cy, double delta, int model, const double *k)
{
    BaCam c = {fx, fy, cx, cy, delta, model, {k[0], k[1], k[2], k[3]}};
    return c;
}
// camera->project(Pc) (G2oTypes.h:250 through Camera): Pinhole.cpp:28-32, or the Kannala-Brandt model of Fisheye.cpp:35-49
__device__ __forceinline__ void ba_project(const BaCam &cam, double X, double Y, double Z, double *u, double *v)
{
    if (cam.model == 0) {
        *u = cam.fx * (X / Z) + cam.cx;
        *v = cam.fy * (Y / Z) + cam.cy;
        return;
    }
    const double a = X / Z, b = Y / Z;
    const double r = sqrt(a * a + b * b);
    const double theta = atan(r);
    const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta2 * theta3, theta7 = theta2 * theta5,
                 theta9 = theta2 * theta7;
    const double theta_d = theta + cam.k[0] * theta3 + cam.k[1] * theta5 + cam.k[2] * theta7 + cam.k[3] * theta9;
    *u = cam.fx * theta_d * a / r + cam.cx; // a point on the optical axis (r = 0) is 0 / 0 here as in the reference
    *v = cam.fy * theta_d * b / r + cam.cy;
}
// camera->getProjJacobian(Pc) (G2oTypes.cpp:42): Pinhole.cpp:49-53 or Fisheye.cpp:83-108; row-major 2 x 3
__device__ __forceinline__ void ba_proj_jacobian(const BaCam &cam, double X, double Y, double Z, double Jp[6])
{
    if (cam.model == 0) {
        Jp[0] = cam.fx / Z; Jp[1] = 0.0; Jp[2] = -cam.fx * X / (Z * Z);
        Jp[3] = 0.0; Jp[4] = cam.fy / Z; Jp[5] = -cam.fy * Y / (Z * Z);
        return;
    }
    const double x2 = X * X, y2 = Y * Y, z2 = Z * Z;
    const double r2 = x2 + y2, r = sqrt(r2), r3 = r2 * r;
    const double theta = atan2(r, Z);
    const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta,
                 theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
    const double f = theta + theta3 * cam.k[0] + theta5 * cam.k[1] + theta7 * cam.k[2] + theta9 * cam.k[3];
    const double fd = 1 + 3 * cam.k[0] * theta2 + 5 * cam.k[1] * theta4 + 7 * cam.k[2] * theta6 + 9 * cam.k[3] * theta8;
    Jp[0] = cam.fx * (fd * Z * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
    Jp[3] = cam.fy * (fd * Z * Y * X / (r2 * (r2 + z2)) - f * Y * X / r3);
    Jp[1] = cam.fx * (fd * Z * Y * X / (r2 * (r2 + z2)) - f * Y * X / r3);
    Jp[4] = cam.fy * (fd * Z * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
    Jp[2] = -cam.fx * fd * X / (r2 + z2);
    Jp[5] = -cam.fy * fd * Y / (r2 + z2);
}

__global__ __launch_bounds__(256) void k_ba_edges(BaCam cam, int n_edges, const double *__restrict__ pose_R,
                                                  const double *__restrict__ pose_t, const uint8_t *__restrict__ pose_fixed,
                                                  const double *__restrict__ points, const int *__restrict__ edge_pose,
                                                  const int *__restrict__ edge_point, const double *__restrict__ edge_z,
                                                  const double *__restrict__ edge_w,
                                                  const uint8_t *__restrict__ edge_active, double *__restrict__ chi2_out,
                                                  double *__restrict__ err_out, double *__restrict__ Hlp,
                                                  double *__restrict__ Cpp /* n_edges x 27 */,
                                                  double *__restrict__ Cll /* n_edges x 9 */)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int ip = edge_pose[e], il = edge_point[e];
    double R[9], t[3], P[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = pose_R[9 * ip + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { t[i] = pose_t[3 * ip + i]; P[i] = points[3 * il + i]; }
    // Pc = Tcw.map(Pw)  (G2oTypes.cpp:41)
    const double X = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
    const double Y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
    const double Z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
    // camera->project and the residual (G2oTypes.h:250)
    double u, v;
    ba_project(cam, X, Y, Z, &u, &v);
    const double ex = edge_z[2 * e] - u, ey = edge_z[2 * e + 1] - v;
    const double om = edge_w[e];
    const double chi2 = om * (ex * ex + ey * ey);
    // g2o RobustKernelHuber: rho'(chi2) = 1 inside delta^2, delta / sqrt(chi2) outside
    double rw = 1.0;
    if (cam.delta > 0.0 && chi2 > cam.delta * cam.delta) rw = cam.delta / sqrt(chi2);
    // an edge at level 1 (Optimize.cpp:900-902) is not part of the active set: its chi2 is still reported
    const double W = (edge_active && !edge_active[e]) ? 0.0 : rw * om;
    // camera->getProjJacobian (G2oTypes.cpp:42)
    double Jp[6];
    ba_proj_jacobian(cam, X, Y, Z, Jp);
    // J_point = -Jp * R  (G2oTypes.cpp:44)
    double Jl[6];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            Jl[3 * r + c] = -(Jp[3 * r] * R[c] + Jp[3 * r + 1] * R[3 + c] + Jp[3 * r + 2] * R[6 + c]);
    // J_pose = [Jp * Hat(Pc), -Jp]  (G2oTypes.cpp:45-46); Hat(Pc) = [0 -Z Y; Z 0 -X; -Y X 0]
    double Jq[12];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double a = Jp[3 * r], b = Jp[3 * r + 1], c = Jp[3 * r + 2];
        Jq[6 * r + 0] = b * Z - c * Y;
        Jq[6 * r + 1] = -a * Z + c * X;
        Jq[6 * r + 2] = a * Y - b * X;
        Jq[6 * r + 3] = -a; Jq[6 * r + 4] = -b; Jq[6 * r + 5] = -c;
    }
    if (chi2_out) chi2_out[e] = chi2;
    if (err_out) { err_out[2 * e] = ex; err_out[2 * e + 1] = ey; }
    const bool fixed = pose_fixed[ip] != 0;
    // point block: H_ll += Jl^T W Jl (6 unique), b_l -= Jl^T W e
    if (!Cll || !Cpp) return; // error-only evaluation (an LM trial): the linearisation of the iteration is kept
    double *cl = Cll + (size_t)e * 9;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) cl[k++] = W * (Jl[i] * Jl[j] + Jl[3 + i] * Jl[3 + j]);
#pragma unroll
    for (int i = 0; i < 3; ++i) cl[6 + i] = -W * (Jl[i] * ex + Jl[3 + i] * ey);
    // pose block (21 unique + 6) and the off-diagonal block H_lp = Jl^T W Jq (3x6)
    double *cp = Cpp + (size_t)e * 27;
    k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) cp[k++] = fixed ? 0.0 : W * (Jq[i] * Jq[j] + Jq[6 + i] * Jq[6 + j]);
#pragma unroll
    for (int i = 0; i < 6; ++i) cp[21 + i] = fixed ? 0.0 : -W * (Jq[i] * ex + Jq[6 + i] * ey);
    if (Hlp) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                Hlp[(size_t)e * 18 + 6 * i + j] = fixed ? 0.0 : W * (Jl[i] * Jq[j] + Jl[3 + i] * Jq[6 + j]);
    }
}

__global__ __launch_bounds__(256) void k_ba_reduce_pose(const int *__restrict__ pose_off, const int *__restrict__ pose_edges,
                                                        const double *__restrict__ Cpp, double *__restrict__ Hpp,
                                                        double *__restrict__ bp)
{
    __shared__ double red[4][27];
    const int ip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    for (int i = pose_off[ip] + tid; i < pose_off[ip + 1]; i += 256) {
        const double *c = Cpp + (size_t)pose_edges[i] * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] += c[k];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (tid < 27) {
        const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        if (tid < 21) {
            int i = 0, k = tid; // unpack the upper-triangular index
            while (k >= 6 - i) { k -= 6 - i; ++i; }
            const int j = i + k;
            if (Hpp) { Hpp[(size_t)ip * 36 + 6 * i + j] = v; Hpp[(size_t)ip * 36 + 6 * j + i] = v; }
        } else if (bp) bp[(size_t)ip * 6 + tid - 21] = v;
    }
}

__global__ __launch_bounds__(256) void k_ba_reduce_point(int n_points, const int *__restrict__ point_off,
                                                         const double *__restrict__ Cll, double *__restrict__ Hll,
                                                         double *__restrict__ bl)
{
    const int il = blockIdx.x * blockDim.x + threadIdx.x;
    if (il >= n_points) return;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    for (int e = point_off[il]; e < point_off[il + 1]; ++e)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += Cll[(size_t)e * 9 + k];
    if (Hll) {
        double *h = Hll + (size_t)il * 9;
        h[0] = acc[0]; h[1] = acc[1]; h[2] = acc[2];
        h[3] = acc[1]; h[4] = acc[3]; h[5] = acc[4];
        h[6] = acc[2]; h[7] = acc[4]; h[8] = acc[5];
    }
    if (bl) { bl[(size_t)il * 3] = acc[6]; bl[(size_t)il * 3 + 1] = acc[7]; bl[(size_t)il * 3 + 2] = acc[8]; }
}

namespace {
// The device side of one host-pointer call.  The BA entry points take no handle and run on two of the reference's threads
// (poseOptimize on Tracking, Tracking.cpp:273-358; localBundleAdjustment on LocalMapping, LocalMapping.cpp:45-52), so a call
// leases a workspace for its duration: a NON-BLOCKING stream, one device arena, one page-locked staging block and two events,
// pooled per device.  Concurrent calls never share a stream or scratch, none of them touches stream 0 (include/orbx.h,
// "Streams": a legacy-stream operation is a barrier against every blocking stream of every thread), and a steady-state call
// allocates nothing (hipMalloc / hipFree wait for the whole device).  Inputs go up in ONE copy from the staging block, results
// come back in one, and the LM loop reads one small block per trial.
struct BaWork {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint8_t *d = nullptr, *h = nullptr;
    size_t d_bytes = 0, h_bytes = 0;
    hipError_t need(size_t dev_bytes, size_t host_bytes)
    {
        hipError_t e = hipSuccess;
        if (dev_bytes > d_bytes) {
            if (d) (void)hipFree(d);
            d = nullptr; d_bytes = 0;
            const size_t want = dev_bytes + dev_bytes / 4;
            if ((e = hipMalloc((void **)&d, want)) != hipSuccess) return e;
            d_bytes = want;
        }
        if (host_bytes > h_bytes) {
            if (h) (void)hipHostFree(h);
            h = nullptr; h_bytes = 0;
            const size_t want = host_bytes + host_bytes / 4;
            if ((e = hipHostMalloc((void **)&h, want, hipHostMallocDefault)) != hipSuccess) return e;
            h_bytes = want;
        }
        return e;
    }
};
std::mutex g_work_mu;
std::vector<BaWork *> g_work_idle; // never destroyed: the HIP runtime may be gone when static destructors run

struct BaLease {
    BaWork *w = nullptr;
    ~BaLease()
    {
        if (!w) return;
        (void)hipStreamSynchronize(w->stream); // an error return may leave work in flight: the next lessee must not meet it
        std::lock_guard<std::mutex> lock(g_work_mu);
        g_work_idle.push_back(w);
    }
    hipError_t acquire()
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        {
            std::lock_guard<std::mutex> lock(g_work_mu);
            for (size_t i = 0; i < g_work_idle.size(); ++i)
                if (g_work_idle[i]->device == dev) {
                    w = g_work_idle[i];
                    g_work_idle.erase(g_work_idle.begin() + (long)i);
                    return hipSuccess;
                }
        }
        BaWork *n = new BaWork();
        n->device = dev;
        if ((e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&n->e0)) != hipSuccess ||
            (e = hipEventCreate(&n->e1)) != hipSuccess) {
            if (n->e0) (void)hipEventDestroy(n->e0);
            if (n->stream) (void)hipStreamDestroy(n->stream);
            delete n;
            return e;
        }
        w = n;
        return hipSuccess;
    }
};
// offsets of a call's arrays in the arena (and, for the staged ones, in the page-locked block), 256-byte aligned
struct Layout {
    size_t n = 0;
    size_t add(size_t bytes) { const size_t o = n; n += (bytes + 255) & ~(size_t)255; return o; }
};
inline void put(uint8_t *base, size_t off, const void *src, size_t bytes) { if (bytes) memcpy(base + off, src, bytes); }
} // namespace

extern "C" int orbba_linearize(const orbba_problem *p, orbba_result *r, int device)
{
    if (!p || !r) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (p->n_poses < 1 || p->n_points < 1 || p->n_edges < 0) return orbx_set_error(ORBX_E_ARG, "bad sizes");
    if (!p->pose_R || !p->pose_t || !p->pose_fixed || !p->points || (p->n_edges && (!p->edge_pose || !p->edge_point || !p->edge_z || !p->edge_inv_sigma2)))
        return orbx_set_error(ORBX_E_ARG, "null input array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device >= 0) B_TRY(hipSetDevice(device));
    const int NP = p->n_poses, NL = p->n_points, NE = p->n_edges;
    // host-side index structures: edges per pose (CSR, edge order) and the contiguous range of each point
    std::vector<int> pose_off(NP + 1, 0), pose_edges(std::max(NE, 1)), point_off(NL + 1, 0);
    for (int e = 0; e < NE; ++e) {
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= NP || p->edge_point[e] < 0 || p->edge_point[e] >= NL)
            return orbx_set_error(ORBX_E_ARG, "edge index out of range");
        if (e && p->edge_point[e] < p->edge_point[e - 1]) return orbx_set_error(ORBX_E_ARG, "edges must be grouped by point");
        pose_off[p->edge_pose[e] + 1]++;
        point_off[p->edge_point[e] + 1]++;
    }
    for (int i = 0; i < NP; ++i) pose_off[i + 1] += pose_off[i];
    for (int i = 0; i < NL; ++i) point_off[i + 1] += point_off[i];
    {
        std::vector<int> fill(pose_off.begin(), pose_off.end() - 1);
        for (int e = 0; e < NE; ++e) pose_edges[fill[p->edge_pose[e]]++] = e;
    }
    // arena: [inputs, one copy up][edge-major scratch][outputs, one copy down: the small blocks first, H_lp last]
    Layout L;
    const size_t oR = L.add(72 * (size_t)NP), ot = L.add(24 * (size_t)NP), ofix = L.add(NP), oP = L.add(24 * (size_t)NL),
                 oep = L.add(4 * (size_t)NE), oel = L.add(4 * (size_t)NE), oz = L.add(16 * (size_t)NE), ow = L.add(8 * (size_t)NE),
                 ope = L.add(4 * (size_t)NE), opo = L.add(4 * (size_t)(NP + 1)), olo = L.add(4 * (size_t)(NL + 1));
    const size_t in_bytes = L.n;
    const size_t oCpp = L.add(216 * (size_t)NE), oCll = L.add(72 * (size_t)NE);
    const size_t out_begin = L.n;
    const size_t oHpp = L.add(288 * (size_t)NP), obp = L.add(48 * (size_t)NP), oHll = L.add(72 * (size_t)NL), obl = L.add(24 * (size_t)NL),
                 ochi = L.add(8 * (size_t)NE), oerr = L.add(16 * (size_t)NE);
    const size_t out_small_end = L.n;
    const size_t oHlp = L.add(144 * (size_t)NE);
    const size_t out_end = (r->H_lp && NE) ? L.n : out_small_end;
    BaLease lease;
    B_TRY(lease.acquire());
    BaWork *w = lease.w;
    B_TRY(w->need(L.n, std::max(in_bytes, out_end - out_begin)));
    hipStream_t s = w->stream;
    uint8_t *d = w->d, *h = w->h;
    put(h, oR, p->pose_R, 72 * (size_t)NP); put(h, ot, p->pose_t, 24 * (size_t)NP); put(h, ofix, p->pose_fixed, NP);
    put(h, oP, p->points, 24 * (size_t)NL);
    if (NE) {
        put(h, oep, p->edge_pose, 4 * (size_t)NE); put(h, oel, p->edge_point, 4 * (size_t)NE); put(h, oz, p->edge_z, 16 * (size_t)NE);
        put(h, ow, p->edge_inv_sigma2, 8 * (size_t)NE); put(h, ope, pose_edges.data(), 4 * (size_t)NE);
    }
    put(h, opo, pose_off.data(), 4 * (size_t)(NP + 1)); put(h, olo, point_off.data(), 4 * (size_t)(NL + 1));
    B_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    const BaCam cam = make_cam(p->fx, p->fy, p->cx, p->cy, p->huber_delta, p->camera_model, p->fisheye_k);
    auto D = [&](size_t off) { return reinterpret_cast<double *>(d + off); };
    auto I = [&](size_t off) { return reinterpret_cast<int *>(d + off); };
    B_TRY(hipEventRecord(w->e0, s));
    if (NE)
        hipLaunchKernelGGL(k_ba_edges, dim3((NE + 255) / 256), dim3(256), 0, s, cam, NE, D(oR), D(ot), d + ofix, D(oP), I(oep), I(oel), D(oz),
                           D(ow), (const uint8_t *)nullptr, D(ochi), D(oerr), D(oHlp), D(oCpp), D(oCll));
    hipLaunchKernelGGL(k_ba_reduce_pose, dim3(NP), dim3(256), 0, s, I(opo), I(ope), D(oCpp), D(oHpp), D(obp));
    hipLaunchKernelGGL(k_ba_reduce_point, dim3((NL + 255) / 256), dim3(256), 0, s, NL, I(olo), D(oCll), D(oHll), D(obl));
    B_TRY(hipEventRecord(w->e1, s));
    B_TRY(hipGetLastError());
    B_TRY(hipMemcpyAsync(h, d + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, s));
    B_TRY(hipStreamSynchronize(s));
    float ms = 0;
    B_TRY(hipEventElapsedTime(&ms, w->e0, w->e1));
    r->kernel_ms = ms;
    auto O = [&](size_t off) { return h + (off - out_begin); }; // an output's place in the staging block
    if (r->chi2 && NE) memcpy(r->chi2, O(ochi), 8 * (size_t)NE);
    if (r->error && NE) memcpy(r->error, O(oerr), 16 * (size_t)NE);
    if (r->H_lp && NE) memcpy(r->H_lp, O(oHlp), 144 * (size_t)NE);
    if (r->H_pp) memcpy(r->H_pp, O(oHpp), 288 * (size_t)NP);
    if (r->b_p) memcpy(r->b_p, O(obp), 48 * (size_t)NP);
    if (r->H_ll) memcpy(r->H_ll, O(oHll), 72 * (size_t)NL);
    if (r->b_l) memcpy(r->b_l, O(obl), 24 * (size_t)NL);
    return ORBX_OK;
}


// =====================================================================================================================
// Levenberg-Marquardt with marginalised points on the device: what g2o's
// OptimizationAlgorithmLevenberg + BlockSolver_6_3 do for the graph of Optimize::localBundleAdjustment
// (modules/Backend/Optimize.cpp:811-911).  All f64, fixed-order reductions, the host only takes the accept/reject
// decision of each trial from three scalars.
//   k_lm_points     thread = point: (H_ll + lambda I)^-1 and its product with b_l
//   k_lm_schur      workgroup = one 6x6 block (i, j >= i) of the reduced system over the free poses:
//                   S_ij = [i == j](H_pp + lambda I) - sum_l H_pl(i,l) H_ll^-1 H_pl(j,l)^T,  rhs_i = b_p - sum_l H_pl H_ll^-1 b_l
//   k_lm_chol_solve single workgroup: in-place Cholesky of S and the two triangular solves
//   k_lm_backsub    thread = point: x_l = H_ll^-1 (b_l - sum_e H_pl^T x_p); partial sums of x (lambda x + b)
//   k_lm_update_*   VertexSE3::oplusImpl (exp(update) * estimate, G2oTypes.h:112-115) and Vertex3D::oplusImpl (:155-158)
//   k_lm_chi2       partial sums of the robustified chi2 over the active edges
// =====================================================================================================================
__global__ __launch_bounds__(256) void k_lm_points(int n_points, double lam, const double *__restrict__ Hll,
                                                   const double *__restrict__ bl, double *__restrict__ inv,
                                                   double *__restrict__ tl)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= n_points) return;
    const double *h = Hll + (size_t)l * 9;
    const double a = h[0] + lam, b = h[1], c = h[2], d = h[4] + lam, e = h[5], f = h[8] + lam; // symmetric
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    double m[9];
    m[0] = c00 * id; m[1] = c01 * id; m[2] = c02 * id;
    m[3] = m[1]; m[4] = (a * f - c * c) * id; m[5] = (b * c - a * e) * id;
    m[6] = m[2]; m[7] = m[5]; m[8] = (a * d - b * b) * id;
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[(size_t)l * 9 + k] = m[k];
    const double b0 = bl[3 * l], b1 = bl[3 * l + 1], b2 = bl[3 * l + 2];
    tl[3 * l] = m[0] * b0 + m[1] * b1 + m[2] * b2;
    tl[3 * l + 1] = m[3] * b0 + m[4] * b1 + m[5] * b2;
    tl[3 * l + 2] = m[6] * b0 + m[7] * b1 + m[8] * b2;
}

__global__ __launch_bounds__(256) void k_lm_schur(int n_free, int n_points, double lam, const int *__restrict__ free_pose,
                                                  const int *__restrict__ edge_of /* [n_free][n_points] */,
                                                  const double *__restrict__ Hpp, const double *__restrict__ bp,
                                                  const double *__restrict__ Hlp, const double *__restrict__ inv,
                                                  const double *__restrict__ tl, double *__restrict__ S,
                                                  double *__restrict__ rhs)
{
    __shared__ double red[4][42];
    // block index -> (i, j >= i)
    int i = 0, rem = blockIdx.x;
    while (rem >= n_free - i) { rem -= n_free - i; ++i; }
    const int j = i + rem, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double acc[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) acc[k] = 0.0;
    const int *ei = edge_of + (size_t)i * n_points, *ej = edge_of + (size_t)j * n_points;
    for (int l = tid; l < n_points; l += 256) {
        const int ea = ei[l], eb = ej[l];
        if (ea < 0 || eb < 0) continue;
        const double *A = Hlp + (size_t)ea * 18, *B = Hlp + (size_t)eb * 18, *m = inv + (size_t)l * 9;
        double Cm[18]; // H_ll^-1 * B (3x6)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) Cm[6 * r + c] = m[3 * r] * B[c] + m[3 * r + 1] * B[6 + c] + m[3 * r + 2] * B[12 + c];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[6 * r + c] += A[r] * Cm[c] + A[6 + r] * Cm[6 + c] + A[12 + r] * Cm[12 + c];
        if (i == j) {
            const double *t = tl + (size_t)l * 3;
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[36 + r] += A[r] * t[0] + A[6 + r] * t[1] + A[12 + r] * t[2];
        }
    }
#pragma unroll
    for (int k = 0; k < 42; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    const int n = 6 * n_free;
    if (tid < 36) {
        const int r = tid / 6, c = tid % 6;
        const double sum = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        double v = -sum;
        if (i == j) v += Hpp[(size_t)free_pose[i] * 36 + tid] + (r == c ? lam : 0.0);
        S[(size_t)(6 * i + r) * n + 6 * j + c] = v;
        if (i != j) S[(size_t)(6 * j + c) * n + 6 * i + r] = v;
    } else if (tid < 42 && i == j) {
        const int r = tid - 36;
        rhs[6 * i + r] = bp[(size_t)free_pose[i] * 6 + r] - ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    }
}

// S (n x n, row-major, lower triangle used) -> L in place; x = S^-1 rhs.  *ok = 0 if a pivot is not positive.
// The same solve with the system resident in LDS (n <= CH_MAX_N, i.e. up to 23 free key frames: (n + 1)^2 doubles).  The global-
// memory kernel below pays three workgroup barriers and as many memory round trips per column (0.52 ms for n = 114, two thirds
// of a local-BA iteration); here a column costs ONE barrier:
//   * the right-hand side rides along as row n of the matrix, so the forward substitution L y = rhs falls out of the
//     factorisation (row n of L is y^T);
//   * step k reads the unscaled column k of the lower triangle, scales on the fly (every thread computes 1 / sqrt(d_k) itself)
//     and writes the scaled column -- L's column k -- into ROW k of the upper triangle, which no step reads: nothing is read
//     and written in one step (1 / L[k][k] goes to an array of its own: a slow thread may still be reading d_k);
//   * thread (tr, tc) of a (CH_T / 16) x 16 grid owns the elements r = tr mod CH_T / 16, c = tc mod 16: no index division;
//   * L^T x = y is solved by ONE wave without barriers (lane owns x[lane], x[lane + 64], x[lane + 128]; x[k] is broadcast
//     with a lane read; the column of the step after is requested before the current one is used).
// Operation order differs from LAPACK's (the oracle) at rounding level only; the LM tests compare at 1e-6.
// Measured for n = 114 (tools/ba_kernel_stats.sh): 0.52 ms -> 0.13; of that 0.04 are the load, the barriers and the back
// substitution, 0.09 the element updates.  Variants that lost: the next pivot's reciprocal square root on an extra wave
// (0.14: the chain is 0.1 us of a step's 1.2), a step's operands of a row -- or of the whole step -- requested before the first
// store (0.17 / 0.39: nine predicated accesses per row cost more than the two or three real ones they wait for).
#ifndef CH_T
#define CH_T 512
#endif
#define CH_MAX_N 140
__global__ __launch_bounds__(CH_T) void k_lm_chol_solve_lds(int n, const double *__restrict__ S, const double *__restrict__ rhs,
                                                            double *__restrict__ x, int *__restrict__ ok)
{
    extern __shared__ __align__(16) double A[]; // (n + 1) rows x P columns, then 1 / L[k][k] for every k
    const int tid = threadIdx.x, P = n + 1;
    double *dinv = A + (size_t)(n + 1) * P;
    for (int i = tid; i < n * n; i += CH_T) {
        const int r = i / n, c = i - r * n;
        if (c <= r) A[r * P + c] = S[i];
    }
    for (int i = tid; i < n; i += CH_T) A[n * P + i] = rhs[i];
    __syncthreads();
    constexpr int TR = CH_T / 16;
    const int tr = tid >> 4, tc = tid & 15;
    bool bad = false;
    for (int k = 0; k < n; ++k) {
        const double d = A[k * P + k];
        if (!(d > 0.0)) { bad = true; break; } // (every thread reads the same value: a uniform exit)
        const double inv = 1.0 / sqrt(d);
        // first row / column of this thread's residue class beyond k
        const int r0 = k + 1 + ((tr - (k + 1)) & (TR - 1)), c0 = k + 1 + ((tc - (k + 1)) & 15);
        for (int r = r0; r <= n; r += TR) {
            const double lr = A[r * P + k] * inv;
            const int cend = r < n ? r : n - 1;
            for (int c = c0; c <= cend; c += 16) A[r * P + c] -= lr * (A[c * P + k] * inv);
            if (tc == 0) A[k * P + r] = lr; // L[r][k]; r = n: y[k]
        }
        if (tid == 0) dinv[k] = inv;
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) *ok = 0;
        for (int i = tid; i < n; i += CH_T) x[i] = 0.0;
        return;
    }
    if (tid >= 64) return;
    // L^T x = y: L[k][r] (r < k) lies at A[r][k]; y[k] at A[k][n]
    constexpr int SL = (CH_MAX_N + 63) / 64;
    double y[SL], a[SL], an[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int r = min(tid + 64 * q, n - 1);
        y[q] = A[r * P + n];
        an[q] = A[r * P + n - 1];
    }
    double di = dinv[n - 1];
    for (int k = n - 1; k >= 0; --k) {
        const double dk = di;
#pragma unroll
        for (int q = 0; q < SL; ++q) a[q] = an[q];
        if (k > 0) { // the step after's operands: independent of this step's result
            di = dinv[k - 1];
#pragma unroll
            for (int q = 0; q < SL; ++q) an[q] = A[min(tid + 64 * q, n - 1) * P + k - 1];
        }
        const int kq = k >> 6, kl = k & 63; // wave-uniform
        double ysel = y[0];
#pragma unroll
        for (int q = 1; q < SL; ++q) ysel = q == kq ? y[q] : ysel;
        const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(ysel) & 0xFFFFFFFFll), kl);
        const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(ysel) >> 32), kl);
        const double xk = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo) * dk;
#pragma unroll
        for (int q = 0; q < SL; ++q) {
            const int r = tid + 64 * q;
            if (r < k) y[q] -= a[q] * xk;
            else if (r == k) y[q] = xk;
        }
    }
#pragma unroll
    for (int q = 0; q < SL; ++q)
        if (tid + 64 * q < n) x[tid + 64 * q] = y[q];
    if (tid == 0) *ok = 1;
}

__global__ __launch_bounds__(256) void k_lm_chol_solve(int n, double *__restrict__ S, const double *__restrict__ rhs,
                                                       double *__restrict__ x, int *__restrict__ ok)
{
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int i = tid; i < n; i += 256) x[i] = rhs[i];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            const double d = S[(size_t)k * n + k];
            if (!(d > 0.0)) bad = 1;
            else S[(size_t)k * n + k] = sqrt(d);
        }
        __syncthreads();
        if (bad) break;
        const double dk = S[(size_t)k * n + k];
        for (int r = k + 1 + tid; r < n; r += 256) S[(size_t)r * n + k] /= dk;
        __syncthreads();
        const int m = n - k - 1;
        for (int idx = tid; idx < m * m; idx += 256) {
            const int r = k + 1 + idx / m, c = k + 1 + idx % m;
            if (c <= r) S[(size_t)r * n + c] -= S[(size_t)r * n + k] * S[(size_t)c * n + k];
        }
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) *ok = 0;
        for (int i = tid; i < n; i += 256) x[i] = 0.0;
        return;
    }
    for (int k = 0; k < n; ++k) { // L y = rhs
        if (tid == 0) x[k] /= S[(size_t)k * n + k];
        __syncthreads();
        const double yk = x[k];
        for (int r = k + 1 + tid; r < n; r += 256) x[r] -= S[(size_t)r * n + k] * yk;
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; --k) { // L^T x = y
        if (tid == 0) x[k] /= S[(size_t)k * n + k];
        __syncthreads();
        const double xk = x[k];
        for (int r = tid; r < k; r += 256) x[r] -= S[(size_t)k * n + r] * xk;
        __syncthreads();
    }
    if (tid == 0) *ok = 1;
}

__device__ __forceinline__ double block_sum_256(double v, double *red4)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

__global__ __launch_bounds__(256) void k_lm_backsub(int n_points, double lam, const int *__restrict__ point_off,
                                                    const int *__restrict__ edge_pose, const int *__restrict__ pose_slot,
                                                    const double *__restrict__ Hlp, const double *__restrict__ xp,
                                                    const double *__restrict__ bl, const double *__restrict__ inv,
                                                    double *__restrict__ xl, double *__restrict__ partial)
{
    __shared__ double red4[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    double sc = 0.0;
    if (l < n_points) {
        double r0 = bl[3 * l], r1 = bl[3 * l + 1], r2 = bl[3 * l + 2];
        for (int e = point_off[l]; e < point_off[l + 1]; ++e) {
            const int s = pose_slot[edge_pose[e]];
            if (s < 0) continue;
            const double *A = Hlp + (size_t)e * 18, *q = xp + 6 * s;
#pragma unroll
            for (int c = 0; c < 6; ++c) { r0 -= A[c] * q[c]; r1 -= A[6 + c] * q[c]; r2 -= A[12 + c] * q[c]; }
        }
        const double *m = inv + (size_t)l * 9;
        const double x0 = m[0] * r0 + m[1] * r1 + m[2] * r2, x1 = m[3] * r0 + m[4] * r1 + m[5] * r2,
                     x2 = m[6] * r0 + m[7] * r1 + m[8] * r2;
        xl[3 * l] = x0; xl[3 * l + 1] = x1; xl[3 * l + 2] = x2;
        sc = x0 * (lam * x0 + bl[3 * l]) + x1 * (lam * x1 + bl[3 * l + 1]) + x2 * (lam * x2 + bl[3 * l + 2]);
    }
    const double tot = block_sum_256(sc, red4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// one workgroup: poses <- exp(x_p) * poses for the free poses, and their share of computeScale()
__global__ __launch_bounds__(256) void k_lm_update_poses(int n_free, double lam, const int *__restrict__ free_pose,
                                                         const double *__restrict__ xp, const double *__restrict__ bp,
                                                         double *__restrict__ R, double *__restrict__ t,
                                                         double *__restrict__ scale_out)
{
    __shared__ double red4[4];
    double sc = 0.0;
    for (int i = threadIdx.x; i < n_free; i += 256) {
        const int ip = free_pose[i];
        const double *u = xp + 6 * i;
        const double wx = u[0], wy = u[1], wz = u[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) sc += u[k] * (lam * u[k] + bp[(size_t)ip * 6 + k]);
        // g2o SE3Quat::exp
        const double th = sqrt(wx * wx + wy * wy + wz * wz);
        const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double K2[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
        double ra, rb, va, vb;
        if (th < 0.00001) { ra = 1.0; rb = 0.5; va = 0.5; vb = 1.0 / 6.0; }
        else {
            ra = sin(th) / th; rb = (1.0 - cos(th)) / (th * th);
            va = rb; vb = (th - sin(th)) / (th * th * th);
        }
        double E[9], V[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double I = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
            E[k] = I + ra * K[k] + rb * K2[k];
            V[k] = I + va * K[k] + vb * K2[k];
        }
        double Rn[9], tn[3];
        const double *Ro = R + (size_t)ip * 9, *to = t + (size_t)ip * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Rn[3 * r + c] = E[3 * r] * Ro[c] + E[3 * r + 1] * Ro[3 + c] + E[3 * r + 2] * Ro[6 + c];
            tn[r] = (E[3 * r] * to[0] + E[3 * r + 1] * to[1] + E[3 * r + 2] * to[2]) +
                    (V[3 * r] * u[3] + V[3 * r + 1] * u[4] + V[3 * r + 2] * u[5]);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) R[(size_t)ip * 9 + k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[(size_t)ip * 3 + k] = tn[k];
    }
    const double tot = block_sum_256(sc, red4);
    if (threadIdx.x == 0) *scale_out = tot;
}

__global__ __launch_bounds__(256) void k_lm_update_points(int n3, const double *__restrict__ xl, double *__restrict__ P)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n3) P[i] += xl[i];
}

__global__ __launch_bounds__(256) void k_lm_chi2(int n_edges, double delta, const double *__restrict__ chi2,
                                                 const uint8_t *__restrict__ active, double *__restrict__ partial)
{
    __shared__ double red4[4];
    const int e = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (e < n_edges && (!active || active[e])) {
        v = chi2[e];
        if (delta > 0.0 && v > delta * delta) v = 2.0 * delta * sqrt(v) - delta * delta; // RobustKernelHuber rho[0]
    }
    const double tot = block_sum_256(v, red4);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// computeLambdaInit (g2o OptimizationAlgorithmLevenberg): max |H_jj| over the free poses' and the points' diagonal entries.
// One workgroup; a maximum does not depend on the order it is taken in.
__global__ __launch_bounds__(256) void k_lm_diag_max(int n_free, const int *__restrict__ free_pose, const double *__restrict__ Hpp,
                                                     int n_points, const double *__restrict__ Hll, double *__restrict__ out)
{
    __shared__ double red4[4];
    double mx = 0.0;
    for (int i = threadIdx.x; i < 6 * n_free; i += 256) mx = fmax(mx, fabs(Hpp[(size_t)36 * free_pose[i / 6] + 7 * (i % 6)]));
    for (int i = threadIdx.x; i < 3 * n_points; i += 256) mx = fmax(mx, fabs(Hll[(size_t)9 * (i / 3) + 4 * (i % 3)]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmax(fmax(red4[0], red4[1]), fmax(red4[2], red4[3]));
}

extern "C" int orbba_optimize(const orbba_problem *p, const orbba_lm_options *o, orbba_lm_result *r, int device)
{
    if (!p || !o || !r) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (p->n_poses < 1 || p->n_points < 1 || p->n_edges < 1) return orbx_set_error(ORBX_E_ARG, "bad sizes");
    if (!p->pose_R || !p->pose_t || !p->pose_fixed || !p->points || !p->edge_pose || !p->edge_point || !p->edge_z ||
        !p->edge_inv_sigma2)
        return orbx_set_error(ORBX_E_ARG, "null input array");
    if (o->max_iterations < 0) return orbx_set_error(ORBX_E_ARG, "negative iteration count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device >= 0) B_TRY(hipSetDevice(device));
    const int NP = p->n_poses, NL = p->n_points, NE = p->n_edges;
    const double tau = o->tau > 0 ? o->tau : 1e-5, lower = o->good_step_lower > 0 ? o->good_step_lower : 1.0 / 3.0,
                 upper = o->good_step_upper > 0 ? o->good_step_upper : 2.0 / 3.0;
    const int max_trials = o->max_trials > 0 ? o->max_trials : 10;
    // ---- index structures
    std::vector<int> pose_off(NP + 1, 0), pose_edges(NE), point_off(NL + 1, 0), free_pose, pose_slot(NP, -1);
    for (int i = 0; i < NP; ++i)
        if (!p->pose_fixed[i]) { pose_slot[i] = (int)free_pose.size(); free_pose.push_back(i); }
    const int NF = (int)free_pose.size(), N = 6 * NF;
    if (NF < 1) return orbx_set_error(ORBX_E_ARG, "every pose is fixed: nothing to optimise");
    if ((size_t)NF * NL > (size_t)1 << 28) return orbx_set_error(ORBX_E_UNSUPPORTED, "free poses x points table too large");
    std::vector<int> edge_of((size_t)NF * NL, -1);
    for (int e = 0; e < NE; ++e) {
        const int ip = p->edge_pose[e], il = p->edge_point[e];
        if (ip < 0 || ip >= NP || il < 0 || il >= NL) return orbx_set_error(ORBX_E_ARG, "edge index out of range");
        if (e && il < p->edge_point[e - 1]) return orbx_set_error(ORBX_E_ARG, "edges must be grouped by point");
        pose_off[ip + 1]++;
        point_off[il + 1]++;
        if (pose_slot[ip] >= 0) {
            int &slot = edge_of[(size_t)pose_slot[ip] * NL + il];
            if (slot >= 0) return orbx_set_error(ORBX_E_ARG, "a point is observed twice by the same key frame");
            slot = e;
        }
    }
    for (int i = 0; i < NP; ++i) pose_off[i + 1] += pose_off[i];
    for (int i = 0; i < NL; ++i) point_off[i + 1] += point_off[i];
    {
        std::vector<int> fill(pose_off.begin(), pose_off.end() - 1);
        for (int e = 0; e < NE; ++e) pose_edges[fill[p->edge_pose[e]]++] = e;
    }
    const int EB = (NE + 255) / 256, LB = (NL + 255) / 256;
    // arena: [estimates R t P -- in and out][other inputs][scratch][chi2 out][read-back block]
    Layout L;
    const size_t oR = L.add(72 * (size_t)NP), ot = L.add(24 * (size_t)NP), oP = L.add(24 * (size_t)NL);
    const size_t est_bytes = L.n;
    const size_t ofix = L.add(NP), oep = L.add(4 * (size_t)NE), oel = L.add(4 * (size_t)NE), oz = L.add(16 * (size_t)NE),
                 ow = L.add(8 * (size_t)NE), oact = L.add(NE), opo = L.add(4 * (size_t)(NP + 1)), ope = L.add(4 * (size_t)NE),
                 olo = L.add(4 * (size_t)(NL + 1)), ofree = L.add(4 * (size_t)NF), oslot = L.add(4 * (size_t)NP),
                 oeo = L.add((size_t)4 * NF * NL);
    const size_t in_bytes = L.n;
    const size_t oRb = L.add(est_bytes), // push() / pop(): R, t, P kept in the same relative layout
                 oHlp = L.add(144 * (size_t)NE), oCpp = L.add(216 * (size_t)NE), oCll = L.add(72 * (size_t)NE),
                 oHpp = L.add(288 * (size_t)NP), obp = L.add(48 * (size_t)NP), oHll = L.add(72 * (size_t)NL), obl = L.add(24 * (size_t)NL),
                 oinv = L.add(72 * (size_t)NL), otl = L.add(24 * (size_t)NL), oS = L.add((size_t)8 * N * N), orhs = L.add(8 * (size_t)N),
                 oxp = L.add(8 * (size_t)N), oxl = L.add(24 * (size_t)NL);
    const size_t ochi = L.add(8 * (size_t)NE);
    // what the host reads per trial, ONE copy: [chi2 partials at the iteration's estimate: EB][at the trial's: EB]
    // [computeScale partials of the points: LB][of the poses: 1][max |H_jj|: 1][Cholesky flag: int in 8 bytes]
    const int RB_CUR = 0, RB_TRY = EB, RB_SCL = 2 * EB, RB_POSE = 2 * EB + LB, RB_MAX = RB_POSE + 1, RB_FLAG = RB_POSE + 2, RB_N = RB_POSE + 3;
    const size_t orb = L.add(8 * (size_t)RB_N);
    // page-locked block: [read-back][inputs up / results down]
    Layout HL;
    const size_t hrb = HL.add(8 * (size_t)RB_N), hio = HL.add(std::max(in_bytes, est_bytes + 8 * (size_t)NE + 256));
    BaLease lease;
    B_TRY(lease.acquire());
    BaWork *w = lease.w;
    B_TRY(w->need(L.n, HL.n));
    hipStream_t s = w->stream;
    uint8_t *d = w->d, *h = w->h + hio;
    double *const rb = reinterpret_cast<double *>(w->h + hrb);
    put(h, oR, p->pose_R, 72 * (size_t)NP); put(h, ot, p->pose_t, 24 * (size_t)NP); put(h, oP, p->points, 24 * (size_t)NL);
    put(h, ofix, p->pose_fixed, NP); put(h, oep, p->edge_pose, 4 * (size_t)NE); put(h, oel, p->edge_point, 4 * (size_t)NE);
    put(h, oz, p->edge_z, 16 * (size_t)NE); put(h, ow, p->edge_inv_sigma2, 8 * (size_t)NE);
    if (o->edge_active) put(h, oact, o->edge_active, NE);
    put(h, opo, pose_off.data(), 4 * (size_t)(NP + 1)); put(h, ope, pose_edges.data(), 4 * (size_t)NE);
    put(h, olo, point_off.data(), 4 * (size_t)(NL + 1)); put(h, ofree, free_pose.data(), 4 * (size_t)NF);
    put(h, oslot, pose_slot.data(), 4 * (size_t)NP); put(h, oeo, edge_of.data(), (size_t)4 * NF * NL);
    B_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    auto D = [&](size_t off) { return reinterpret_cast<double *>(d + off); };
    auto I = [&](size_t off) { return reinterpret_cast<int *>(d + off); };
    const uint8_t *d_active = o->edge_active ? d + oact : nullptr;
    const BaCam cam = make_cam(p->fx, p->fy, p->cx, p->cy, p->huber_delta, p->camera_model, p->fisheye_k);
    B_TRY(hipEventRecord(w->e0, s));

    // errors at the current estimate (+ the whole linearisation when `full`); activeRobustChi2 as partial sums into the
    // read-back block's slot `slot`
    auto evaluate = [&](bool full, int slot) {
        hipLaunchKernelGGL(k_ba_edges, dim3(EB), dim3(256), 0, s, cam, NE, D(oR), D(ot), d + ofix, D(oP), I(oep), I(oel), D(oz), D(ow),
                           d_active, D(ochi), (double *)nullptr, full ? D(oHlp) : nullptr, full ? D(oCpp) : nullptr,
                           full ? D(oCll) : nullptr);
        if (full) {
            hipLaunchKernelGGL(k_ba_reduce_pose, dim3(NP), dim3(256), 0, s, I(opo), I(ope), D(oCpp), D(oHpp), D(obp));
            hipLaunchKernelGGL(k_ba_reduce_point, dim3(LB), dim3(256), 0, s, NL, I(olo), D(oCll), D(oHll), D(obl));
        }
        hipLaunchKernelGGL(k_lm_chi2, dim3(EB), dim3(256), 0, s, NE, cam.delta, D(ochi), d_active, D(orb) + slot);
    };
    // the read-back block in host memory: the call's only waits besides the final one
    auto read_back = [&]() -> int {
        B_TRY(hipGetLastError());
        B_TRY(hipMemcpyAsync(rb, d + orb, 8 * (size_t)RB_N, hipMemcpyDeviceToHost, s));
        B_TRY(hipStreamSynchronize(s));
        return ORBX_OK;
    };
    auto sum = [&](int from, int n) { double v = 0.0; for (int b = 0; b < n; ++b) v += rb[from + b]; return v; };

    double lam = 0.0, ni = 2.0, chi_initial = 0.0, current = 0.0;
    int its = 0, trials_total = 0, rc = ORBX_OK;
    for (int it = 0; it < o->max_iterations; ++it) {
        evaluate(true, RB_CUR);
        bool have_current = false;
        if (it == 0) {
            if (o->user_lambda_init > 0) lam = o->user_lambda_init;
            else { // computeLambdaInit: tau * max |H_jj| over the free vertices
                hipLaunchKernelGGL(k_lm_diag_max, dim3(1), dim3(256), 0, s, NF, I(ofree), D(oHpp), NL, D(oHll), D(orb) + RB_MAX);
                if ((rc = read_back())) return rc;
                lam = tau * rb[RB_MAX];
                current = sum(RB_CUR, EB);
                have_current = true;
            }
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            // push(): keep the estimates
            B_TRY(hipMemcpyAsync(d + oRb, d + oR, est_bytes, hipMemcpyDeviceToDevice, s)); // R, t, P and their copies are laid out alike
            hipLaunchKernelGGL(k_lm_points, dim3(LB), dim3(256), 0, s, NL, lam, D(oHll), D(obl), D(oinv), D(otl));
            hipLaunchKernelGGL(k_lm_schur, dim3(NF * (NF + 1) / 2), dim3(256), 0, s, NF, NL, lam, I(ofree), I(oeo), D(oHpp), D(obp), D(oHlp),
                               D(oinv), D(otl), D(oS), D(orhs));
            int *const d_flag = reinterpret_cast<int *>(D(orb) + RB_FLAG);
            // ORBBA_VAR_CHOL = 1: the global-memory kernel, the parity twin (read per call)
            bool in_lds = N <= CH_MAX_N && g_ba_chol.load() == 0;
            if (in_lds) {
                const size_t lds = sizeof(double) * ((size_t)(N + 1) * (N + 1) + N);
                // per device; when the opt-in or the launch is refused the global-memory kernel solves the system instead --
                // a launch that did not happen would leave the previous iteration's step in xp / the flag
                in_lds = orbx_lds_opt_in(reinterpret_cast<const void *>(k_lm_chol_solve_lds), lds) == hipSuccess;
                if (in_lds) {
                    hipLaunchKernelGGL(k_lm_chol_solve_lds, dim3(1), dim3(CH_T), lds, s, N, D(oS), D(orhs), D(oxp), d_flag);
                    in_lds = hipGetLastError() == hipSuccess;
                }
            }
            if (!in_lds) {
                hipLaunchKernelGGL(k_lm_chol_solve, dim3(1), dim3(256), 0, s, N, D(oS), D(orhs), D(oxp), d_flag);
                B_TRY(hipGetLastError());
            }
            hipLaunchKernelGGL(k_lm_backsub, dim3(LB), dim3(256), 0, s, NL, lam, I(olo), I(oep), I(oslot), D(oHlp), D(oxp), D(obl), D(oinv),
                               D(oxl), D(orb) + RB_SCL);
            hipLaunchKernelGGL(k_lm_update_poses, dim3(1), dim3(256), 0, s, NF, lam, I(ofree), D(oxp), D(obp), D(oR), D(ot), D(orb) + RB_POSE);
            hipLaunchKernelGGL(k_lm_update_points, dim3((3 * NL + 255) / 256), dim3(256), 0, s, 3 * NL, D(oxl), D(oP));
            evaluate(false, RB_TRY);
            if ((rc = read_back())) return rc;
            if (!have_current) { current = sum(RB_CUR, EB); have_current = true; }
            if (it == 0 && qmax == 0) chi_initial = current;
            double temp = sum(RB_TRY, EB);
            int ok2 = 0;
            memcpy(&ok2, rb + RB_FLAG, sizeof(int));
            if (!ok2) temp = std::numeric_limits<double>::max();
            double scale = rb[RB_POSE]; // poses first, then the points block by block
            for (int b = 0; b < LB; ++b) scale += rb[RB_SCL + b];
            scale += 1e-3;
            rho = (current - temp) / scale;
            ++trials_total;
            if (rho > 0 && std::isfinite(temp)) {
                double alpha = 1.0 - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, upper);
                lam *= std::max(lower, alpha);
                ni = 2.0;
                current = temp;
            } else {
                lam *= ni;
                ni *= 2;
                // pop(): restore
                B_TRY(hipMemcpyAsync(d + oR, d + oRb, est_bytes, hipMemcpyDeviceToDevice, s));
                if (!std::isfinite(lam)) break;
            }
            ++qmax;
        } while (rho < 0 && qmax < max_trials);
        ++its;
        if (qmax == max_trials || rho == 0 || !std::isfinite(lam)) break; // Terminate
    }
    evaluate(false, RB_TRY);
    B_TRY(hipEventRecord(w->e1, s));
    B_TRY(hipGetLastError());
    B_TRY(hipMemcpyAsync(rb, d + orb, 8 * (size_t)RB_N, hipMemcpyDeviceToHost, s));
    B_TRY(hipMemcpyAsync(h, d, est_bytes, hipMemcpyDeviceToHost, s));
    if (r->chi2) B_TRY(hipMemcpyAsync(h + est_bytes, d + ochi, 8 * (size_t)NE, hipMemcpyDeviceToHost, s));
    B_TRY(hipStreamSynchronize(s));
    const double final_chi = sum(RB_TRY, EB);
    float ms = 0;
    B_TRY(hipEventElapsedTime(&ms, w->e0, w->e1));
    r->iterations = its;
    r->trials = trials_total;
    r->lambda = lam;
    r->chi2_initial = o->max_iterations > 0 ? chi_initial : final_chi;
    r->chi2_final = final_chi;
    r->device_ms = ms;
    if (r->pose_R) memcpy(r->pose_R, h + oR, 72 * (size_t)NP);
    if (r->pose_t) memcpy(r->pose_t, h + ot, 24 * (size_t)NP);
    if (r->points) memcpy(r->points, h + oP, 24 * (size_t)NL);
    if (r->chi2) memcpy(r->chi2, h + est_bytes, 8 * (size_t)NE);
    return ORBX_OK;
}

// The optimisation part of Optimize::localBundleAdjustment (Optimize.cpp:892-922)
extern "C" int orbba_local_bundle_adjustment(const orbba_problem *p, orbba_lm_result *r, uint8_t *outlier, int device)
{
    if (!p || !r) return orbx_set_error(ORBX_E_ARG, "null argument");
    const int NP = p->n_poses, NL = p->n_points, NE = p->n_edges;
    if (NP < 1 || NL < 1 || NE < 1) return orbx_set_error(ORBX_E_ARG, "bad sizes");
    std::vector<double> R1((size_t)9 * NP), t1((size_t)3 * NP), P1((size_t)3 * NL), chi((size_t)NE);
    orbba_lm_options o1 = {};
    o1.max_iterations = 5; // optimizer.optimize(5), :893
    orbba_lm_result r1 = {};
    r1.pose_R = R1.data(); r1.pose_t = t1.data(); r1.points = P1.data(); r1.chi2 = chi.data();
    int rc = orbba_optimize(p, &o1, &r1, device);
    if (rc) return rc;
    std::vector<uint8_t> active((size_t)NE);
    for (int e = 0; e < NE; ++e) active[e] = !(chi[e] > 5.991); // :899-902
    orbba_problem p2 = *p;
    p2.pose_R = R1.data(); p2.pose_t = t1.data(); p2.points = P1.data();
    p2.huber_delta = 0.0; // setRobustKernel(nullptr), :904
    orbba_lm_options o2 = {};
    o2.max_iterations = 10; // :909
    o2.edge_active = active.data();
    std::vector<double> chi2v((size_t)NE);
    double *user_chi = r->chi2;
    r->chi2 = chi2v.data();
    rc = orbba_optimize(&p2, &o2, r, device);
    r->chi2 = user_chi;
    if (rc) return rc;
    r->device_ms += r1.device_ms;
    r->iterations += r1.iterations;
    r->trials += r1.trials;
    r->chi2_initial = r1.chi2_initial;
    // An edge demoted to level 1 at :899-902 is not part of optimize(10)'s active set: g2o never calls computeError()
    // on it again, so e->chi2() at :919 still evaluates the error vector of the first round -- above 5.991 by
    // construction -- and the observation is always erased, whatever its residual at the final estimate would be
    // (poseOptimize has an explicit computeError() for exactly that reason, Optimize.cpp:510; this loop has none).
    for (int e = 0; e < NE; ++e)
        if (!active[e]) chi2v[e] = chi[e];
    if (user_chi) std::copy(chi2v.begin(), chi2v.end(), user_chi);
    if (outlier)
        for (int e = 0; e < NE; ++e) outlier[e] = !active[e] || chi2v[e] > 5.991; // :917-921
    return ORBX_OK;
}


// =====================================================================================================================
// Optimize::poseOptimize (modules/Backend/Optimize.cpp:444-545) for a batch of frames: one workgroup runs one frame's
// whole procedure -- four rounds of (restart from the initial pose, up to ten Levenberg-Marquardt iterations on the
// inlier edges with the Huber kernel, re-classification of every edge at chi2 <= 5.991) -- without leaving the kernel.
// The 6x6 system is reduced over the edges in a fixed order (strided per thread, wave shuffles, four partials) and then
// solved redundantly by every lane, so all control flow is uniform across the workgroup.
// =====================================================================================================================
__device__ __forceinline__ void se3_exp_apply(const double *u, double *R, double *t)
{
    const double wx = u[0], wy = u[1], wz = u[2];
    const double th = sqrt(wx * wx + wy * wy + wz * wz);
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    double ra, rb, va, vb;
    if (th < 0.00001) { ra = 1.0; rb = 0.5; va = 0.5; vb = 1.0 / 6.0; }
    else {
        ra = sin(th) / th; rb = (1.0 - cos(th)) / (th * th);
        va = rb; vb = (th - sin(th)) / (th * th * th);
    }
    double E[9], V[9], Rn[9], tn[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double I = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
        E[k] = I + ra * K[k] + rb * K2[k];
        V[k] = I + va * K[k] + vb * K2[k];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) Rn[3 * r + c] = E[3 * r] * R[c] + E[3 * r + 1] * R[3 + c] + E[3 * r + 2] * R[6 + c];
        tn[r] = (E[3 * r] * t[0] + E[3 * r + 1] * t[1] + E[3 * r + 2] * t[2]) + (V[3 * r] * u[3] + V[3 * r + 1] * u[4] + V[3 * r + 2] * u[5]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rn[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = tn[k];
}

// all 256 threads get the sum of v[0..K) over the workgroup, summed in a fixed order
template <int K> __device__ __forceinline__ void block_sum_vals(double *v, double (*red)[28])
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        v[k] = x;
    }
    __syncthreads(); // the previous round's readers are done
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

// solves (H + lam I) x = b for the symmetric 6x6 H given by its 21 upper-triangular entries; false if not positive definite
__device__ __forceinline__ bool solve6(const double *h21, double lam, const double *b, double *x)
{
    double L[36];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) { L[6 * j + i] = h21[k] + (i == j ? lam : 0.0); ++k; } // lower triangle
    for (int c = 0; c < 6; ++c) {
        double d = L[6 * c + c];
        for (int m = 0; m < c; ++m) d -= L[6 * c + m] * L[6 * c + m];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        L[6 * c + c] = d;
        for (int r = c + 1; r < 6; ++r) {
            double v = L[6 * r + c];
            for (int m = 0; m < c; ++m) v -= L[6 * r + m] * L[6 * c + m];
            L[6 * r + c] = v / d;
        }
    }
    double y[6];
    for (int r = 0; r < 6; ++r) {
        double v = b[r];
        for (int m = 0; m < r; ++m) v -= L[6 * r + m] * y[m];
        y[r] = v / L[6 * r + r];
    }
    for (int r = 5; r >= 0; --r) {
        double v = y[r];
        for (int m = r + 1; m < 6; ++m) v -= L[6 * m + r] * x[m];
        x[r] = v / L[6 * r + r];
    }
    return true;
}

__global__ __launch_bounds__(256) void k_pose_optimize(BaCam cam, int rounds, int iterations, int cap, const int *__restrict__ edge_off,
                                                       const double *__restrict__ R0, const double *__restrict__ t0,
                                                       const double *__restrict__ Pw, const double *__restrict__ z,
                                                       const double *__restrict__ w, double *__restrict__ R_out,
                                                       double *__restrict__ t_out, uint8_t *__restrict__ inlier,
                                                       int *__restrict__ n_inliers, double *__restrict__ chi2_out)
{
    __shared__ double red[4][28];
    // A frame's edges (map point, measurement, weight, inlier flag: 49 bytes each) are read in every one of up to 4 x 10 x 2 passes:
    // up to `cap` of them are staged in LDS once (structure of arrays), so a pass costs LDS latency instead of an L2 round trip per
    // operand.  Same operations in the same order as from global memory (a frame with more edges keeps reading it from there).
    extern __shared__ __align__(16) double sm[];
    const int f = blockIdx.x, tid = threadIdx.x, e0 = edge_off[f], n = edge_off[f + 1] - e0;
    const bool staged = n <= cap;
    double *const sPx = sm, *const sPy = sPx + cap, *const sPz = sPy + cap, *const szu = sPz + cap, *const szv = szu + cap, *const sww = szv + cap;
    uint8_t *const sin_ = reinterpret_cast<uint8_t *>(sww + cap);
    double Ri[9], ti[3], R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Ri[k] = R0[(size_t)f * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = ti[k] = t0[(size_t)f * 3 + k];
    for (int e = tid; e < n; e += 256) {
        inlier[e0 + e] = 1;
        if (staged) {
            sPx[e] = Pw[(size_t)(e0 + e) * 3]; sPy[e] = Pw[(size_t)(e0 + e) * 3 + 1]; sPz[e] = Pw[(size_t)(e0 + e) * 3 + 2];
            szu[e] = z[(size_t)(e0 + e) * 2]; szv[e] = z[(size_t)(e0 + e) * 2 + 1]; sww[e] = w[e0 + e];
            sin_[e] = 1;
        }
    }
    __syncthreads();
    auto is_in = [&](int e) -> bool { return staged ? sin_[e] != 0 : inlier[e0 + e] != 0; };
    auto weight = [&](int e) -> double { return staged ? sww[e] : w[e0 + e]; };
    const double d2 = cam.delta * cam.delta;

    // residual of edge e at pose (R, t); returns chi2, optionally the pieces needed for the Jacobian
    auto edge_chi = [&](int e, const double *Rc, const double *tc, double *ex, double *ey, double *Pc) -> double {
        double P[3], zu, zv;
        if (staged) { P[0] = sPx[e]; P[1] = sPy[e]; P[2] = sPz[e]; zu = szu[e]; zv = szv[e]; }
        else {
            const double *Pg = Pw + (size_t)(e0 + e) * 3;
            P[0] = Pg[0]; P[1] = Pg[1]; P[2] = Pg[2]; zu = z[(size_t)(e0 + e) * 2]; zv = z[(size_t)(e0 + e) * 2 + 1];
        }
        const double X = Rc[0] * P[0] + Rc[1] * P[1] + Rc[2] * P[2] + tc[0];
        const double Y = Rc[3] * P[0] + Rc[4] * P[1] + Rc[5] * P[2] + tc[1];
        const double Z = Rc[6] * P[0] + Rc[7] * P[1] + Rc[8] * P[2] + tc[2];
        double u, v;
        ba_project(cam, X, Y, Z, &u, &v);
        *ex = zu - u;
        *ey = zv - v;
        Pc[0] = X; Pc[1] = Y; Pc[2] = Z;
        return weight(e) * (*ex * *ex + *ey * *ey);
    };
    auto robust_sum = [&](const double *Rc, const double *tc) -> double { // activeRobustChi2
        double s[1] = {0.0};
        for (int e = tid; e < n; e += 256) {
            if (!is_in(e)) continue;
            double ex, ey, Pc[3];
            double c = edge_chi(e, Rc, tc, &ex, &ey, Pc);
            if (cam.delta > 0.0 && c > d2) c = 2.0 * cam.delta * sqrt(c) - d2;
            s[0] += c;
        }
        block_sum_vals<1>(s, red);
        return s[0];
    };

    if (n >= 3) { // fewer than three correspondences: nothing is done (:491)
        for (int round = 0; round < rounds; ++round) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = Ri[k]; // vPose->setEstimate(Tcw) (:497)
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = ti[k];
            double cnt[1] = {0.0};
            for (int e = tid; e < n; e += 256) cnt[0] += is_in(e) ? 1.0 : 0.0;
            block_sum_vals<1>(cnt, red);
            double lam = 0.0, ni = 2.0;
            if (cnt[0] > 0.0) {
                for (int it = 0; it < iterations; ++it) {
                    // computeActiveErrors + buildSystem
                    double acc[28];
#pragma unroll
                    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
                    for (int e = tid; e < n; e += 256) {
                        if (!is_in(e)) continue;
                        double ex, ey, Pc[3];
                        const double c = edge_chi(e, R, t, &ex, &ey, Pc);
                        double rw = 1.0, rc = c;
                        if (cam.delta > 0.0 && c > d2) { rw = cam.delta / sqrt(c); rc = 2.0 * cam.delta * sqrt(c) - d2; }
                        const double W = rw * weight(e), X = Pc[0], Y = Pc[1], Z = Pc[2];
                        double Jp[6];
                        ba_proj_jacobian(cam, X, Y, Z, Jp);
                        double Jq[12];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const double a = Jp[3 * r], b = Jp[3 * r + 1], cc = Jp[3 * r + 2];
                            Jq[6 * r + 0] = b * Z - cc * Y;
                            Jq[6 * r + 1] = -a * Z + cc * X;
                            Jq[6 * r + 2] = a * Y - b * X;
                            Jq[6 * r + 3] = -a; Jq[6 * r + 4] = -b; Jq[6 * r + 5] = -cc;
                        }
                        int k = 0;
#pragma unroll
                        for (int i = 0; i < 6; ++i)
#pragma unroll
                            for (int j = i; j < 6; ++j) acc[k++] += W * (Jq[i] * Jq[j] + Jq[6 + i] * Jq[6 + j]);
#pragma unroll
                        for (int i = 0; i < 6; ++i) acc[21 + i] += -W * (Jq[i] * ex + Jq[6 + i] * ey);
                        acc[27] += rc;
                    }
                    block_sum_vals<28>(acc, red);
                    double current = acc[27];
                    const double *h21 = acc, *b = acc + 21;
                    if (it == 0) { // computeLambdaInit
                        double mx = 0.0;
                        int k = 0;
                        for (int i = 0; i < 6; ++i) { mx = fmax(mx, fabs(h21[k])); k += 6 - i; }
                        lam = 1e-5 * mx;
                        ni = 2.0;
                    }
                    double rho = 0.0;
                    int qmax = 0;
                    do {
                        double Rb[9], tb[3], x[6];
#pragma unroll
                        for (int k = 0; k < 9; ++k) Rb[k] = R[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) tb[k] = t[k];
                        const bool ok2 = solve6(h21, lam, b, x);
                        if (!ok2)
#pragma unroll
                            for (int k = 0; k < 6; ++k) x[k] = 0.0;
                        se3_exp_apply(x, R, t);
                        double temp = robust_sum(R, t);
                        if (!ok2) temp = 1.7976931348623157e308;
                        double scale = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) scale += x[k] * (lam * x[k] + b[k]);
                        scale += 1e-3;
                        rho = (current - temp) / scale;
                        if (rho > 0 && isfinite(temp)) {
                            double alpha = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                            alpha = fmin(alpha, 2.0 / 3.0);
                            lam *= fmax(1.0 / 3.0, alpha);
                            ni = 2.0;
                            current = temp;
                        } else {
                            lam *= ni;
                            ni *= 2;
#pragma unroll
                            for (int k = 0; k < 9; ++k) R[k] = Rb[k];
#pragma unroll
                            for (int k = 0; k < 3; ++k) t[k] = tb[k];
                            if (!isfinite(lam)) break;
                        }
                        ++qmax;
                    } while (rho < 0 && qmax < 10);
                    if (qmax == 10 || rho == 0 || !isfinite(lam)) break;
                }
            }
            // :503-516 -- every edge is re-evaluated at the round's pose
            __syncthreads();
            for (int e = tid; e < n; e += 256) {
                double ex, ey, Pc[3];
                const double c = edge_chi(e, R, t, &ex, &ey, Pc);
                inlier[e0 + e] = !(c > 5.991);
                if (staged) sin_[e] = !(c > 5.991);
            }
            __syncthreads();
        }
    }
    double cnt[1] = {0.0};
    for (int e = tid; e < n; e += 256) {
        double ex, ey, Pc[3];
        const double c = edge_chi(e, R, t, &ex, &ey, Pc);
        if (chi2_out) chi2_out[e0 + e] = c;
        cnt[0] += (n >= 3 && is_in(e)) ? 1.0 : 0.0;
    }
    block_sum_vals<1>(cnt, red);
    if (tid == 0) {
        n_inliers[f] = (int)cnt[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) R_out[(size_t)f * 9 + k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t_out[(size_t)f * 3 + k] = t[k];
    }
}

// edges of a frame staged in LDS by k_pose_optimize (49 bytes each; ORBBA_VAR_POSE_LDS = 0: never, the parity twin)
static int pose_lds_cap() { return g_ba_pose_lds.load(); }
static size_t pose_lds_bytes(int cap) { return ((size_t)cap * 49 + 15) / 16 * 16; }
static hipError_t pose_lds_configure(int cap)
{
    return orbx_lds_opt_in(reinterpret_cast<const void *>(k_pose_optimize), pose_lds_bytes(cap)); // per device, largest size wins
}
extern "C" int orbba_pose_optimize_batch(const orbba_pose_problem *p, orbba_pose_result *r, int device)
{
    if (!p || !r) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (p->n_frames < 1 || !p->edge_off || !p->pose_R || !p->pose_t) return orbx_set_error(ORBX_E_ARG, "bad sizes / null arrays");
    const int B = p->n_frames, NE = p->edge_off[B];
    for (int f = 0; f < B; ++f)
        if (p->edge_off[f + 1] < p->edge_off[f] || p->edge_off[0] != 0)
            return orbx_set_error(ORBX_E_ARG, "edge offsets must start at 0 and be non-decreasing");
    if (NE > 0 && (!p->points || !p->edge_z || !p->edge_inv_sigma2)) return orbx_set_error(ORBX_E_ARG, "null edge array");
    if (!r->pose_R || !r->pose_t || !r->n_inliers || (NE > 0 && !r->inlier)) return orbx_set_error(ORBX_E_ARG, "null output array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return orbx_set_error(ORBX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device >= 0) B_TRY(hipSetDevice(device));
    // arena: [inputs, one copy up][results, one copy down]
    Layout L;
    const size_t ooff = L.add(4 * (size_t)(B + 1)), oR0 = L.add(72 * (size_t)B), ot0 = L.add(24 * (size_t)B), oP = L.add(24 * (size_t)NE),
                 oz = L.add(16 * (size_t)NE), ow = L.add(8 * (size_t)NE);
    const size_t in_bytes = L.n;
    const size_t oR = L.add(72 * (size_t)B), ot = L.add(24 * (size_t)B), oni = L.add(4 * (size_t)B), oin = L.add(NE);
    const size_t out_small_end = L.n;
    const size_t ochi = L.add(8 * (size_t)NE);
    const size_t out_end = (NE && r->chi2) ? L.n : out_small_end;
    BaLease lease;
    B_TRY(lease.acquire());
    BaWork *w = lease.w;
    B_TRY(w->need(L.n, std::max(in_bytes, out_end - in_bytes)));
    hipStream_t s = w->stream;
    uint8_t *d = w->d, *h = w->h;
    put(h, ooff, p->edge_off, 4 * (size_t)(B + 1)); put(h, oR0, p->pose_R, 72 * (size_t)B); put(h, ot0, p->pose_t, 24 * (size_t)B);
    if (NE) { put(h, oP, p->points, 24 * (size_t)NE); put(h, oz, p->edge_z, 16 * (size_t)NE); put(h, ow, p->edge_inv_sigma2, 8 * (size_t)NE); }
    B_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    const BaCam cam = make_cam(p->fx, p->fy, p->cx, p->cy, p->huber_delta, p->camera_model, p->fisheye_k);
    auto D = [&](size_t off) { return reinterpret_cast<double *>(d + off); };
    B_TRY(hipEventRecord(w->e0, s));
    int max_n = 0;
    for (int f = 0; f < B; ++f) max_n = std::max(max_n, p->edge_off[f + 1] - p->edge_off[f]);
    const int cap = std::min(pose_lds_cap(), max_n); // (no more LDS than the largest frame needs: more workgroups per CU for a big batch)
    B_TRY(pose_lds_configure(cap));
    hipLaunchKernelGGL(k_pose_optimize, dim3(B), dim3(256), pose_lds_bytes(cap), s, cam, p->rounds > 0 ? p->rounds : 4,
                       p->iterations > 0 ? p->iterations : 10, cap, reinterpret_cast<int *>(d + ooff), D(oR0), D(ot0), D(oP), D(oz), D(ow),
                       D(oR), D(ot), d + oin, reinterpret_cast<int *>(d + oni), D(ochi));
    B_TRY(hipEventRecord(w->e1, s));
    B_TRY(hipGetLastError());
    B_TRY(hipMemcpyAsync(h, d + in_bytes, out_end - in_bytes, hipMemcpyDeviceToHost, s));
    B_TRY(hipStreamSynchronize(s));
    float ms = 0;
    B_TRY(hipEventElapsedTime(&ms, w->e0, w->e1));
    r->kernel_ms = ms;
    auto O = [&](size_t off) { return h + (off - in_bytes); }; // a result's place in the staging block
    memcpy(r->pose_R, O(oR), 72 * (size_t)B);
    memcpy(r->pose_t, O(ot), 24 * (size_t)B);
    memcpy(r->n_inliers, O(oni), 4 * (size_t)B);
    if (NE) memcpy(r->inlier, O(oin), NE);
    if (NE && r->chi2) memcpy(r->chi2, O(ochi), 8 * (size_t)NE);
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------
// The tracking chain without a host hop (Tracking.cpp:289-336): the matched map points of a device-resident frame record
// (orbm_search_by_projection_*_device wrote frame_mp) become the edges of poseOptimize (Optimize.cpp:468-490) on the
// device, and orbba_pose_optimize_batch_device runs k_pose_optimize on device arrays.
// ---------------------------------------------------------------------------------------------
// edges in key-point order (:470): for key point i with frame_mp[i] in [0, nq): the map point's position, kp.pt, 1 / size^2
__global__ __launch_bounds__(1024) void k_pose_edges(int n2, int nq, const int32_t *__restrict__ frame_mp,
                                                     const orbx_kp *__restrict__ kps, const float *__restrict__ q_points,
                                                     int32_t *__restrict__ edge_off, double *__restrict__ points,
                                                     double *__restrict__ z, double *__restrict__ w, int32_t *__restrict__ edge_kp)
{
    __shared__ int s_wave[16], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n2; i0 += 1024) {
        const int i = i0 + tid;
        const int q = i < n2 ? frame_mp[i] : -1;
        const bool on = q >= 0 && q < nq;
        const unsigned long long mk = __ballot(on);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (lane == 0) s_wave[wv] = (int)__popcll(mk);
        __syncthreads();
        int before = s_base;
        for (int k = 0; k < wv; ++k) before += s_wave[k];
        if (on) {
            const int e = before + rank;
            const orbx_kp kp = kps[i];
            points[3 * e] = (double)q_points[3 * q]; points[3 * e + 1] = (double)q_points[3 * q + 1]; points[3 * e + 2] = (double)q_points[3 * q + 2];
            z[2 * e] = (double)kp.x; z[2 * e + 1] = (double)kp.y;           // setMeasurement({kp.pt.x, kp.pt.y}) (:478)
            w[e] = (double)__fdiv_rn(__fdiv_rn(1.f, kp.size), kp.size);     // invSigma2 = 1.f / kp.size / kp.size (:479)
            if (edge_kp) edge_kp[e] = i;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += s_wave[k]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) { edge_off[0] = 0; edge_off[1] = s_base; }
}
extern "C" int orbba_pose_edges_device(int n2, int nq, const int32_t *d_frame_mp, const void *d_kps, const float *d_q_points,
                                       int32_t *d_edge_off, double *d_points, double *d_edge_z, double *d_edge_inv_sigma2,
                                       int32_t *d_edge_kp, void *stream)
{
    if (n2 < 0 || nq < 0 || !d_frame_mp || !d_kps || !d_q_points || !d_edge_off || !d_points || !d_edge_z || !d_edge_inv_sigma2)
        return orbx_set_error(ORBX_E_ARG, "bad argument");
    hipLaunchKernelGGL(k_pose_edges, dim3(1), dim3(1024), 0, (hipStream_t)stream, n2, nq, d_frame_mp, (const orbx_kp *)d_kps,
                       d_q_points, d_edge_off, d_points, d_edge_z, d_edge_inv_sigma2, d_edge_kp);
    B_TRY(hipGetLastError());
    return ORBX_OK;
}
extern "C" int orbba_pose_optimize_batch_device(const orbba_pose_problem *p, orbba_pose_result *r, void *stream)
{
    if (!p || !r) return orbx_set_error(ORBX_E_ARG, "null argument");
    if (p->n_frames < 1 || !p->edge_off || !p->pose_R || !p->pose_t || !p->points || !p->edge_z || !p->edge_inv_sigma2 ||
        !r->pose_R || !r->pose_t || !r->n_inliers || !r->inlier || !r->chi2)
        return orbx_set_error(ORBX_E_ARG, "null array (every pointer is device memory here, chi2 included)");
    const BaCam cam = make_cam(p->fx, p->fy, p->cx, p->cy, p->huber_delta, p->camera_model, p->fisheye_k);
    const int cap = pose_lds_cap(); // (the edge counts are on the device: the full capacity, one workgroup per CU)
    B_TRY(pose_lds_configure(cap));
    hipLaunchKernelGGL(k_pose_optimize, dim3(p->n_frames), dim3(256), pose_lds_bytes(cap), (hipStream_t)stream, cam, p->rounds > 0 ? p->rounds : 4,
                       p->iterations > 0 ? p->iterations : 10, cap, p->edge_off, p->pose_R, p->pose_t, p->points, p->edge_z,
                       p->edge_inv_sigma2, r->pose_R, r->pose_t, r->inlier, r->n_inliers, r->chi2);
    B_TRY(hipGetLastError());
    r->kernel_ms = 0.f;
    return ORBX_OK;
}
