/* Scalar numeric primitives shared by the HIP kernels and their host-side planners.
 *
 * Everything here is written so that host (g++/clang x86-64) and device (gfx950) evaluate the same
 * sequence of individually rounded IEEE operations: no FMA contraction (the library is built with
 * -ffp-contract=off and the hot spots use explicit __f*_rn/__d*_rn intrinsics on the device),
 * round-half-to-even conversions, correctly rounded division.
 *
 * Reference call sites: cvRound -> src/extractors/ORBextractor.cpp:21,55,59-60,926;
 * cos/sin(float) -> :53; cv::fastAtan2 -> :43.
 */
#ifndef TB_MATH_H
#define TB_MATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TB_HD __host__ __device__ inline
#else
#define TB_HD inline
#endif

/* The __f*_rn / __d*_rn intrinsics of ROCm 7.2 are plain operators (no rounding-mode variants are
 * built in), so contraction must be off for the whole translation unit, not only through them. */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace tbm {

#if defined(__HIP_DEVICE_COMPILE__)
#define TB_FMUL(a, b) __fmul_rn((a), (b))
#define TB_FADD(a, b) __fadd_rn((a), (b))
#define TB_FSUB(a, b) __fsub_rn((a), (b))
#define TB_DMUL(a, b) __dmul_rn((a), (b))
#define TB_DADD(a, b) __dadd_rn((a), (b))
#define TB_DSUB(a, b) __dsub_rn((a), (b))
#define TB_FDIV(a, b) __fdiv_rn((a), (b))
#else
#define TB_FMUL(a, b) ((a) * (b))
#define TB_FADD(a, b) ((a) + (b))
#define TB_FSUB(a, b) ((a) - (b))
#define TB_DMUL(a, b) ((a) * (b))
#define TB_DADD(a, b) ((a) + (b))
#define TB_DSUB(a, b) ((a) - (b))
#define TB_FDIV(a, b) ((a) / (b))
#endif

/* cvRound: round-half-to-even (cvtss2si) */
TB_HD int cv_round(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)nearbyintf(v);
#endif
}
TB_HD int cv_round(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_rn(v);
#else
    return (int)nearbyint(v);
#endif
}
TB_HD int cv_floor(float v) { int i = (int)v; return i - (i > v); }
TB_HD int cv_floor(double v) { int i = (int)v; return i - (i > v); }

TB_HD uint32_t abstop12(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (__float_as_uint(x) >> 20) & 0x7ff;
#else
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
#endif
}

/* cosf/sinf for |y| < 120: double-precision pi/2 reduction + polynomial pair, rounded once to float.
 * Same algorithm as the libm the reference links (glibc >= 2.28); tests/test_oracle_math.py pins the
 * oracle's copy against libm bit for bit, tests/test_gpu_* pin this copy against the oracle. */
#define TB_SC_HPI_INV 0x1.45F306DC9C883p+23
#define TB_SC_HPI 0x1.921FB54442D18p0
#define TB_SC_C1 (-0x1.ffffffd0c621cp-2)
#define TB_SC_C2 0x1.55553e1068f19p-5
#define TB_SC_C3 (-0x1.6c087e89a359dp-10)
#define TB_SC_C4 0x1.99343027bf8c3p-16
#define TB_SC_S1 (-0x1.555545995a603p-3)
#define TB_SC_S2 0x1.1107605230bc4p-7
#define TB_SC_S3 (-0x1.994eb3774cf24p-13)

/* sgn = +1 or -1 flips the cosine polynomial (second table of the libm algorithm) */
TB_HD float sincos_poly(double x, double x2, double csgn, int n) {
    if ((n & 1) == 0) {
        double x3 = TB_DMUL(x, x2);
        double s1 = TB_DADD(TB_SC_S2, TB_DMUL(x2, TB_SC_S3));
        double x7 = TB_DMUL(x3, x2);
        double s = TB_DADD(x, TB_DMUL(x3, TB_SC_S1));
        return (float)TB_DADD(s, TB_DMUL(x7, s1));
    }
    double x4 = TB_DMUL(x2, x2);
    double c2 = TB_DADD(csgn * TB_SC_C3, TB_DMUL(x2, csgn * TB_SC_C4));
    double c1 = TB_DADD(csgn * 1.0, TB_DMUL(x2, csgn * TB_SC_C1));
    double x6 = TB_DMUL(x4, x2);
    double c = TB_DADD(c1, TB_DMUL(x4, csgn * TB_SC_C2));
    return (float)TB_DADD(c, TB_DMUL(x6, c2));
}

TB_HD void sincosf_rn(float y, float* sn, float* cs) {
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = TB_DMUL(x, x);
        if (abstop12(y) < abstop12(0x1p-12f)) {
            *cs = 1.0f;
            *sn = y;
            return;
        }
        *cs = sincos_poly(x, x2, 1.0, 1);
        *sn = sincos_poly(x, x2, 1.0, 0);
        return;
    }
    double r = TB_DMUL(x, TB_SC_HPI_INV);
    int n = ((int32_t)r + 0x800000) >> 24;
    double xr = TB_DSUB(x, TB_DMUL((double)n, TB_SC_HPI));
    const int q = n & 3;
    const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
    const double csgn = (n & 2) ? -1.0 : 1.0;
    double xs = TB_DMUL(xr, s);
    double x2 = TB_DMUL(xr, xr);
    *cs = sincos_poly(xs, x2, csgn, n ^ 1);
    *sn = sincos_poly(xs, x2, csgn, n);
}

/* cv::fastAtan2(y, x): degrees in [0,360) (OpenCV 3.3 scalar polynomial) */
TB_HD float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180.0 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180.0 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180.0 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180.0 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = TB_FDIV(ay, TB_FADD(ax, eps));
        c2 = TB_FMUL(c, c);
        a = TB_FMUL(TB_FADD(TB_FMUL(TB_FADD(TB_FMUL(TB_FADD(TB_FMUL(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = TB_FDIV(ax, TB_FADD(ay, eps));
        c2 = TB_FMUL(c, c);
        a = TB_FSUB(90.f, TB_FMUL(TB_FADD(TB_FMUL(TB_FADD(TB_FMUL(TB_FADD(TB_FMUL(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = TB_FSUB(180.f, a);
    if (y < 0) a = TB_FSUB(360.f, a);
    return a;
}

}  // namespace tbm
#endif
