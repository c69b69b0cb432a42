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
#include <string>
#include <vector>

#include "../../include/orbba.h"
#include "../../include/orbx.h"

int orbx_set_error(int code, const std::string &msg);
#define B_TRY(expr)                                                                                    \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return orbx_set_error(ORBX_E_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct BaCam { double fx, fy, cx, cy, delta; };

__global__ __launch_bounds__(256) void k_ba_edges(BaCam cam, int n_edges, const double *__restrict__ pose_R,
                                                  const double *__restrict__ pose_t, const uint8_t *__restrict__ pose_fixed,
                                                  const double *__restrict__ points, const int *__restrict__ edge_pose,
                                                  const int *__restrict__ edge_point, const double *__restrict__ edge_z,
                                                  const double *__restrict__ edge_w, double *__restrict__ chi2_out,
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
    // Pinhole::project (Pinhole.cpp:28-32) and the residual (G2oTypes.h:250)
    const double u = cam.fx * (X / Z) + cam.cx, v = cam.fy * (Y / Z) + cam.cy;
    const double ex = edge_z[2 * e] - u, ey = edge_z[2 * e + 1] - v;
    const double om = edge_w[e];
    const double chi2 = om * (ex * ex + ey * ey);
    // g2o RobustKernelHuber: rho'(chi2) = 1 inside delta^2, delta / sqrt(chi2) outside
    double rw = 1.0;
    if (cam.delta > 0.0 && chi2 > cam.delta * cam.delta) rw = cam.delta / sqrt(chi2);
    const double W = rw * om;
    // Pinhole::getProjJacobian (Pinhole.cpp:49-53)
    const double Jp[6] = {cam.fx / Z, 0.0, -cam.fx * X / (Z * Z), 0.0, cam.fy / Z, -cam.fy * Y / (Z * Z)};
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
struct Dev {
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max(bytes, (size_t)8)); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
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
    Dev dR, dt, dfix, dP, dep, del, dz, dw, dchi, derr, dHlp, dCpp, dCll, dpo, dpe, dlo, dHpp, dbp, dHll, dbl;
    B_TRY(dR.alloc(sizeof(double) * 9 * NP)); B_TRY(dt.alloc(sizeof(double) * 3 * NP)); B_TRY(dfix.alloc(NP));
    B_TRY(dP.alloc(sizeof(double) * 3 * NL)); B_TRY(dep.alloc(sizeof(int) * NE)); B_TRY(del.alloc(sizeof(int) * NE));
    B_TRY(dz.alloc(sizeof(double) * 2 * NE)); B_TRY(dw.alloc(sizeof(double) * NE)); B_TRY(dchi.alloc(sizeof(double) * NE));
    B_TRY(derr.alloc(sizeof(double) * 2 * NE)); B_TRY(dHlp.alloc(sizeof(double) * 18 * NE));
    B_TRY(dCpp.alloc(sizeof(double) * 27 * NE)); B_TRY(dCll.alloc(sizeof(double) * 9 * NE));
    B_TRY(dpo.alloc(sizeof(int) * (NP + 1))); B_TRY(dpe.alloc(sizeof(int) * std::max(NE, 1))); B_TRY(dlo.alloc(sizeof(int) * (NL + 1)));
    B_TRY(dHpp.alloc(sizeof(double) * 36 * NP)); B_TRY(dbp.alloc(sizeof(double) * 6 * NP));
    B_TRY(dHll.alloc(sizeof(double) * 9 * NL)); B_TRY(dbl.alloc(sizeof(double) * 3 * NL));
    B_TRY(hipMemcpy(dR.p, p->pose_R, sizeof(double) * 9 * NP, hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(dt.p, p->pose_t, sizeof(double) * 3 * NP, hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(dfix.p, p->pose_fixed, NP, hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(dP.p, p->points, sizeof(double) * 3 * NL, hipMemcpyHostToDevice));
    if (NE) {
        B_TRY(hipMemcpy(dep.p, p->edge_pose, sizeof(int) * NE, hipMemcpyHostToDevice));
        B_TRY(hipMemcpy(del.p, p->edge_point, sizeof(int) * NE, hipMemcpyHostToDevice));
        B_TRY(hipMemcpy(dz.p, p->edge_z, sizeof(double) * 2 * NE, hipMemcpyHostToDevice));
        B_TRY(hipMemcpy(dw.p, p->edge_inv_sigma2, sizeof(double) * NE, hipMemcpyHostToDevice));
        B_TRY(hipMemcpy(dpe.p, pose_edges.data(), sizeof(int) * NE, hipMemcpyHostToDevice));
    }
    B_TRY(hipMemcpy(dpo.p, pose_off.data(), sizeof(int) * (NP + 1), hipMemcpyHostToDevice));
    B_TRY(hipMemcpy(dlo.p, point_off.data(), sizeof(int) * (NL + 1), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    B_TRY(hipEventCreate(&e0)); B_TRY(hipEventCreate(&e1));
    const BaCam cam = {p->fx, p->fy, p->cx, p->cy, p->huber_delta};
    B_TRY(hipEventRecord(e0, 0));
    if (NE)
        hipLaunchKernelGGL(k_ba_edges, dim3((NE + 255) / 256), dim3(256), 0, 0, cam, NE, dR.as<double>(), dt.as<double>(),
                           dfix.as<uint8_t>(), dP.as<double>(), dep.as<int>(), del.as<int>(), dz.as<double>(), dw.as<double>(),
                           dchi.as<double>(), derr.as<double>(), dHlp.as<double>(), dCpp.as<double>(), dCll.as<double>());
    hipLaunchKernelGGL(k_ba_reduce_pose, dim3(NP), dim3(256), 0, 0, dpo.as<int>(), dpe.as<int>(), dCpp.as<double>(),
                       dHpp.as<double>(), dbp.as<double>());
    hipLaunchKernelGGL(k_ba_reduce_point, dim3((NL + 255) / 256), dim3(256), 0, 0, NL, dlo.as<int>(), dCll.as<double>(),
                       dHll.as<double>(), dbl.as<double>());
    B_TRY(hipEventRecord(e1, 0));
    B_TRY(hipEventSynchronize(e1));
    B_TRY(hipGetLastError());
    float ms = 0;
    B_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    r->kernel_ms = ms;
    if (r->chi2 && NE) B_TRY(hipMemcpy(r->chi2, dchi.p, sizeof(double) * NE, hipMemcpyDeviceToHost));
    if (r->error && NE) B_TRY(hipMemcpy(r->error, derr.p, sizeof(double) * 2 * NE, hipMemcpyDeviceToHost));
    if (r->H_lp && NE) B_TRY(hipMemcpy(r->H_lp, dHlp.p, sizeof(double) * 18 * NE, hipMemcpyDeviceToHost));
    if (r->H_pp) B_TRY(hipMemcpy(r->H_pp, dHpp.p, sizeof(double) * 36 * NP, hipMemcpyDeviceToHost));
    if (r->b_p) B_TRY(hipMemcpy(r->b_p, dbp.p, sizeof(double) * 6 * NP, hipMemcpyDeviceToHost));
    if (r->H_ll) B_TRY(hipMemcpy(r->H_ll, dHll.p, sizeof(double) * 9 * NL, hipMemcpyDeviceToHost));
    if (r->b_l) B_TRY(hipMemcpy(r->b_l, dbl.p, sizeof(double) * 3 * NL, hipMemcpyDeviceToHost));
    return ORBX_OK;
}
