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

// (cosf, sinf) of angle_deg * (float)(CV_PI/180.f): double-precision Cody-Waite
// reduction + degree-13/12 kernel polynomials, rounded once to float.
ORB_HD void orb_sincos_deg(float angle_deg, float *cs, float *sn)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float a = ORB_FMUL(angle_deg, factorPI);
    const double x = (double)a;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double kd = floor(ORB_DADD(ORB_DMUL(x, two_over_pi), 0.5));
    const int k = (int)kd;
    const double r = ORB_DSUB(ORB_DSUB(x, ORB_DMUL(kd, pio2_hi)), ORB_DMUL(kd, pio2_lo));
    const double z = ORB_DMUL(r, r);
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6;
    ps = ORB_DADD(ORB_DMUL(ps, z), S5);
    ps = ORB_DADD(ORB_DMUL(ps, z), S4);
    ps = ORB_DADD(ORB_DMUL(ps, z), S3);
    ps = ORB_DADD(ORB_DMUL(ps, z), S2);
    ps = ORB_DADD(ORB_DMUL(ps, z), S1);
    const double s = ORB_DADD(r, ORB_DMUL(r, ORB_DMUL(z, ps)));
    double pc = C6;
    pc = ORB_DADD(ORB_DMUL(pc, z), C5);
    pc = ORB_DADD(ORB_DMUL(pc, z), C4);
    pc = ORB_DADD(ORB_DMUL(pc, z), C3);
    pc = ORB_DADD(ORB_DMUL(pc, z), C2);
    pc = ORB_DADD(ORB_DMUL(pc, z), C1);
    const double c = ORB_DSUB(1.0, ORB_DSUB(ORB_DMUL(0.5, z), ORB_DMUL(z, ORB_DMUL(z, pc))));
    double cv, sv;
    switch (k & 3) {
    case 0: cv = c; sv = s; break;
    case 1: cv = -s; sv = c; break;
    case 2: cv = -c; sv = -s; break;
    default: cv = s; sv = -c; break;
    }
    *cs = (float)cv;
    *sn = (float)sv;
}
