// Scalar arithmetic shared by the host-side tables and the HIP kernels.
//
// Everything here must produce the same bits on the host (gcc/clang x86-64) and
// on gfx950, so it only uses IEEE-754 +,-,*,/ on float/double, written through
// the ORB_F* wrappers that forbid FMA contraction on the device (the build also
// passes -ffp-contract=off).  Reference semantics being reproduced:
//   cvRound/cvFloor/cvCeil      (used at modules/ORB/ORBExtractor.cpp:21,56,61-62,398-403,448,564,645-646)
//   cv::fastAtan2               (modules/ORB/ORBExtractor.cpp:41)
//   cos/sin(angle * CV_PI/180)  (modules/ORB/ORBExtractor.cpp:53-54)
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORB_HD __host__ __device__ __forceinline__
#else
#define ORB_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ORB_FMUL(a, b) __fmul_rn((a), (b))
#define ORB_FADD(a, b) __fadd_rn((a), (b))
#define ORB_FSUB(a, b) __fsub_rn((a), (b))
#define ORB_FDIV(a, b) __fdiv_rn((a), (b))
#define ORB_DMUL(a, b) __dmul_rn((a), (b))
#define ORB_DADD(a, b) __dadd_rn((a), (b))
#define ORB_DSUB(a, b) __dsub_rn((a), (b))
#define ORB_DDIV(a, b) __ddiv_rn((a), (b))
#else
#define ORB_FMUL(a, b) ((a) * (b))
#define ORB_FADD(a, b) ((a) + (b))
#define ORB_FSUB(a, b) ((a) - (b))
#define ORB_FDIV(a, b) ((a) / (b))
#define ORB_DMUL(a, b) ((a) * (b))
#define ORB_DADD(a, b) ((a) + (b))
#define ORB_DSUB(a, b) ((a) - (b))
#define ORB_DDIV(a, b) ((a) / (b))
#endif

// cvRound(float): round half to even (SSE cvtss2si under the default MXCSR)
ORB_HD int orb_round_f(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)lrintf(v);
#endif
}
ORB_HD int orb_floor_f(float v) { int i = (int)v; return i - (i > v); }
ORB_HD int orb_ceil_f(float v) { int i = (int)v; return i + (i < v); }

// cv::fastAtan2(y, x) in degrees, OpenCV 4.2 scalar polynomial
ORB_HD float orb_fast_atan2(float y, float x)
{
    const float rad2deg = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2, t;
    if (ax >= ay) {
        c = ORB_FDIV(ay, ORB_FADD(ax, eps));
        c2 = ORB_FMUL(c, c);
        t = ORB_FADD(ORB_FMUL(p7, c2), p5);
        t = ORB_FADD(ORB_FMUL(t, c2), p3);
        t = ORB_FADD(ORB_FMUL(t, c2), p1);
        a = ORB_FMUL(t, c);
    } else {
        c = ORB_FDIV(ax, ORB_FADD(ay, eps));
        c2 = ORB_FMUL(c, c);
        t = ORB_FADD(ORB_FMUL(p7, c2), p5);
        t = ORB_FADD(ORB_FMUL(t, c2), p3);
        t = ORB_FADD(ORB_FMUL(t, c2), p1);
        a = ORB_FSUB(90.f, ORB_FMUL(t, c));
    }
    if (x < 0) a = ORB_FSUB(180.f, a);
    if (y < 0) a = ORB_FSUB(360.f, a);
    return a;
}

// glibc's sincosf on [0, 2*pi] -- what `cos(angle)`, `sin(angle)` on a float resolve to at
// modules/ORB/ORBExtractor.cpp:53-54 (`using namespace std` at :11 -> std::cos(float) -> cosf).
// glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c; not vendored by the
// reference, pinned here by the container's glibc 2.35): |y| < pi/4 takes the degree-7 / degree-8
// polynomials directly; otherwise n = round(y * 2/pi) by a 2^24-scaled double -> int32 conversion
// (the !TOINT_INTRINSICS branch x86-64 builds), x = y - n*(pi/2) in double, polynomial picked by the
// quadrant, rounded once to float.  tests/test_oracle_kat.py compares the oracle's copy of this with
// the host libm's sinf / cosf / sincosf on every float in [0, 2*pi] (0 mismatches; the x86-64
// FMA and SSE2 ifunc variants agree on that whole range, so the separate multiply-add form is used).
ORB_HD float orb_sincosf_poly(double x, double x2, double sg, int odd)
{
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if (!odd) {
        const double x3 = ORB_DMUL(x, x2);
        const double s1 = ORB_DADD(S2, ORB_DMUL(x2, S3));
        const double x7 = ORB_DMUL(x3, x2);
        const double s = ORB_DADD(x, ORB_DMUL(x3, S1));
        return (float)ORB_DADD(s, ORB_DMUL(x7, s1));
    }
    // __sincosf_table[1] holds the cosine coefficients negated (exact), selected when n & 2
    const double x4 = ORB_DMUL(x2, x2);
    const double c2 = ORB_DADD(sg * C3, ORB_DMUL(x2, sg * C4));
    const double c1 = ORB_DADD(sg * C0, ORB_DMUL(x2, sg * C1));
    const double x6 = ORB_DMUL(x4, x2);
    const double c = ORB_DADD(c1, ORB_DMUL(x4, sg * C2));
    return (float)ORB_DADD(c, ORB_DMUL(x6, c2));
}
ORB_HD void orb_sincosf(float y, float *sn, float *cs)
{
    union { float f; uint32_t u; } bits;
    bits.f = y;
    const uint32_t top = (bits.u >> 20) & 0x7ff;
    const double x = (double)y;
    if (top < 0x3f4) {                      // abstop12(y) < abstop12(0x1.921FB6p-1f)
        if (top < 0x398) {                  // < abstop12(0x1p-12f)
            *sn = y;
            *cs = 1.0f;
            return;
        }
        const double x2 = ORB_DMUL(x, x);
        *sn = orb_sincosf_poly(x, x2, 1.0, 0);
        *cs = orb_sincosf_poly(x, x2, 1.0, 1);
        return;
    }
    // reduce_fast; valid below 120.0f, and the callers stay within [0, 2*pi]
    const double r = ORB_DMUL(x, 0x1.45F306DC9C883p+23);
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = ORB_DSUB(x, ORB_DMUL((double)n, 0x1.921FB54442D18p0));
    const double sign = ((n + 1) & 2) ? -1.0 : 1.0;     // sign[n & 3] = {1, -1, -1, 1}
    const double sg = (n & 2) ? -1.0 : 1.0;
    const double xs = ORB_DMUL(xr, sign), x2 = ORB_DMUL(xr, xr);
    *sn = orb_sincosf_poly(xs, x2, sg, n & 1);
    *cs = orb_sincosf_poly(xs, x2, sg, (n ^ 1) & 1);
}

// (cosf, sinf) of angle_deg * (float)(CV_PI/180.f)  (modules/ORB/ORBExtractor.cpp:16,53-54)
ORB_HD void orb_sincos_deg(float angle_deg, float *cs, float *sn)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    orb_sincosf(ORB_FMUL(angle_deg, factorPI), sn, cs);
}
