/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the tracking hot path.
 *
 * Scalar numeric primitives of the reference path, restated for the CPU.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/; the product (trackingbench_slam_amd/) must not.
 *
 * PARITY STATUS: "parity unpinned" w.r.t. genuine OpenCV 3.3 -- the reference
 * holds no golden vectors and OpenCV is not in the image.
 * What IS pinned here: orc_cosf/orc_sinf agree bit-for-bit with this
 * container's glibc 2.35 cosf/sinf (the libm the reference's `cos(float)` /
 * `sin(float)` calls at src/extractors/ORBextractor.cpp:53 bind to) on every
 * sampled float in [0, 2*pi] (tests/test_oracle_math.py).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

/* cvRound(float/double) on x86-64 = cvtss2si / cvtsd2si = round-half-to-even
 * (reference uses it at ORBextractor.cpp:21,55,59-60,926). */
static inline int cv_round(float v) { return (int)std::nearbyintf(v); }
static inline int cv_round(double v) { return (int)std::nearbyint(v); }
static inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(float v) { int i = (int)v; return i + (i < v); }

/* ---- cosf/sinf: restatement of the glibc >= 2.28 single-precision algorithm
 * (double-precision range reduction by pi/2 and a degree-7/8 polynomial pair).
 * Written with explicit, un-fused multiply and add so the HIP copy can be
 * bit-identical.  Valid for |y| < 120 (angles here are in [0, 2*pi]). */
struct sincos_tab {
    double sign[4];
    double hpi_inv, hpi;
    double c0, c1, c2, c3, c4;
    double s1, s2, s3;
};

static const sincos_tab k_sincos[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

static inline uint32_t abstop12(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}

static inline float sincos_poly(double x, double x2, const sincos_tab* p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = p->s2 + x2 * p->s3;
        double x7 = x3 * x2;
        double s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    }
    double x4 = x2 * x2;
    double c2 = p->c3 + x2 * p->c4;
    double c1 = p->c0 + x2 * p->c1;
    double x6 = x4 * x2;
    double c = c1 + x4 * p->c2;
    return (float)(c + x6 * c2);
}

static inline double sincos_reduce(double x, const sincos_tab* p, int* np) {
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * p->hpi;
}

static inline float orc_cosf(float y) {
    double x = y;
    const sincos_tab* p = &k_sincos[0];
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincos_poly(x, x * x, p, 1);
    }
    x = sincos_reduce(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &k_sincos[1];
    return sincos_poly(x * s, x * x, p, n ^ 1);
}

static inline float orc_sinf(float y) {
    double x = y;
    const sincos_tab* p = &k_sincos[0];
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincos_poly(x, x * x, p, 0);
    }
    x = sincos_reduce(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &k_sincos[1];
    return sincos_poly(x * s, x * x, p, n);
}

/* cv::fastAtan2(y, x) -> degrees in [0, 360): OpenCV 3.3 scalar polynomial
 * (SURVEY.md App. A.3 [memory]); called at ORBextractor.cpp:43. */
static inline float fast_atan2(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace orc
#endif
