/* TEST INFRASTRUCTURE ONLY -- CPU oracle of the RANSAC stage behind LocalBA::AddMapPointsByStereo:
 *   Matcher::rejectWithF (src/matchers/matcher.cpp:853-881) = cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, 1.0, 0.99, mask)
 *   Matcher::searchByOPFlow(..., reject = true) (matcher.cpp:724-768)
 *   LocalBA::AddMapPointsByStereo (src/mapping/LocalBA.cpp:46-68): depth = bf / |x_tracked - x_key|
 *
 * cv::findFundamentalMat is OpenCV 3.3 (calib3d/fundam.cpp + ptsetreg.cpp), NOT in the reference tree and not installed
 * here: restated from its published structure, PARITY UNPINNED (the reference holds no vector for it). What is restated
 * step by step:
 *   - RANSACPointSetRegistrator::run: cv::RNG seeded with (uint64)-1 (multiply-with-carry, 4164903690), getSubset draws
 *     7 distinct indices with rng.uniform(0, count) (a duplicate redraws that index), FMEstimatorCallback::checkSubset
 *     rejects a sample whose 7th point is collinear with two earlier ones in either image (the whole sample is redrawn),
 *     up to three models per sample, findInliers with err <= (float)(threshold^2), the best model is the first with
 *     goodCount > max(best so far, 6), RANSACUpdateNumIters(0.99, outlier ratio, 7, niters) after every improvement,
 *     1000 iterations at most;
 *   - FMEstimatorCallback::computeError: max of the two squared point-to-epipolar-line distances, in double, stored as float;
 *   - run7Point: the 7 x 9 epipolar system, its two-dimensional null space {f1, f2}, det(lambda f1 + (1 - lambda) f2) = 0
 *     as a cubic in lambda, F scaled to F[8] = 1;
 *   - findFundamentalMat's dispatch: fewer than 7 points -> nothing; exactly 7 -> the 7-point solver, every point an
 *     inlier; 8..14 -> LMeDSPointSetRegistrator (restated in round 3, see orc_find_fundamental_ransac); 15 and more -> RANSAC.
 * Two numerical routines are NOT OpenCV's, on purpose: they are built from + - * / sqrt only, so that the HIP kernel
 * reproduces them bit for bit (libm's acos / cos / pow differ between host and device):
 *   - the null space comes from Gauss-Jordan elimination with row pivoting (OpenCV: Jacobi SVD whose last two right
 *     singular vectors are completed from a random start, cv::RNG(0x12345678) -- another basis of the same plane; the
 *     singular members of the pencil, i.e. the candidate F, are the same up to rounding);
 *   - the cubic's real roots come from bracketing by its stationary points and bisection, in ascending order (OpenCV:
 *     cv::solveCubic's trigonometric / Cardano forms, in its own order -- the order only decides ties between models of
 *     one sample).
 * UB in the reference that is given a defined meaning (SURVEY App. C policy): rejectWithF indexes an empty fund_status
 * when findFundamentalMat was not called (at most 8 keys) or returned no mask (fewer than 7 tracked points, no model):
 * here the status flags are left as they are.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <vector>
#include <algorithm>

#include "oracle.h"

namespace {

struct CvRng {                       /* cv::RNG, core/operations.hpp */
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

/* haveCollinearPoints (fundam.cpp): is the last of `count` points on a line through two earlier ones (or on top of one)? */
bool have_collinear(const float* p, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

/* real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3], ascending; + - * / sqrt only (see the header) */
int cubic_roots(const double c[4], double r[3]) {
    const double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) return 0;
            r[0] = -a3 / a2;
            return 1;
        }
        double d = a2 * a2 - 4 * a1 * a3;
        if (d < 0) return 0;
        d = sqrt(d);
        const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
        double x0, x1;
        if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; } else { x0 = q2 / a1; x1 = a3 / q2; }
        if (!(d > 0)) { r[0] = x0; return 1; }
        r[0] = x0 < x1 ? x0 : x1;
        r[1] = x0 < x1 ? x1 : x0;
        return 2;
    }
    const double a = a1 / a0, b = a2 / a0, cc = a3 / a0;
    auto p = [&](double x) { return ((x + a) * x + b) * x + cc; };
    double M = fabs(a);
    if (fabs(b) > M) M = fabs(b);
    if (fabs(cc) > M) M = fabs(cc);
    M = M + 1.0;                                     /* Cauchy bound: every root lies in (-M, M) */
    auto bisect = [&](double lo, double hi) {        /* p(lo) <= 0 <= p(hi) or the reverse */
        const bool rising = p(lo) <= 0;
        for (int it = 0; it < 128; it++) {
            const double mid = 0.5 * (lo + hi);
            if (mid == lo || mid == hi) break;
            const double v = p(mid);
            if ((v <= 0) == rising) lo = mid; else hi = mid;
        }
        return 0.5 * (lo + hi);
    };
    const double disc = a * a - 3 * b;
    if (!(disc > 0)) { r[0] = bisect(-M, M); return 1; }
    const double s = sqrt(disc);
    const double xl = (-a - s) / 3, xh = (-a + s) / 3;  /* local maximum, local minimum */
    const double pl = p(xl), ph = p(xh);
    int n = 0;
    if (pl >= 0) r[n++] = pl == 0 ? xl : bisect(-M, xl);
    if (pl > 0 && ph < 0) r[n++] = bisect(xl, xh);
    if (ph <= 0) r[n++] = ph == 0 ? xh : bisect(xh, M);
    return n;
}

/* FMEstimatorCallback::run7Point restated (null space and cubic as in the header): up to 3 matrices, 9 doubles each */
int run_7point(const float* m1, const float* m2, double* F) {
    double A[7][9];
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
        A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
        A[i][6] = x0; A[i][7] = y0; A[i][8] = 1;
    }
    /* Gauss-Jordan with row pivoting, column by column; columns without a pivot are free */
    int pivcol[7], npiv = 0, freecol[9], nfree = 0;
    for (int col = 0; col < 9; col++) {
        if (npiv == 7) { freecol[nfree++] = col; continue; }
        int best = npiv;
        double bv = fabs(A[npiv][col]);
        for (int r = npiv + 1; r < 7; r++)
            if (fabs(A[r][col]) > bv) { bv = fabs(A[r][col]); best = r; }
        double scale = 0;                         /* largest entry of the remaining rows: the pivot's yardstick */
        for (int r = npiv; r < 7; r++)
            for (int k = col; k < 9; k++) scale = fabs(A[r][k]) > scale ? fabs(A[r][k]) : scale;
        if (!(bv > 1e-12 * scale)) { freecol[nfree++] = col; continue; }
        if (best != npiv)
            for (int k = 0; k < 9; k++) std::swap(A[best][k], A[npiv][k]);
        const double inv = 1.0 / A[npiv][col];
        for (int k = 0; k < 9; k++) A[npiv][k] *= inv;
        for (int r = 0; r < 7; r++) {
            if (r == npiv) continue;
            const double f = A[r][col];
            if (f == 0) continue;
            for (int k = 0; k < 9; k++) A[r][k] -= f * A[npiv][k];
        }
        pivcol[npiv++] = col;
    }
    if (nfree != 2) return 0;                     /* rank-deficient sample: no model (OpenCV would still return some) */
    double f1[9], f2[9];
    for (int k = 0; k < 9; k++) f1[k] = f2[k] = 0;
    f1[freecol[0]] = 1;
    f2[freecol[1]] = 1;
    for (int r = 0; r < 7; r++) { f1[pivcol[r]] = -A[r][freecol[0]]; f2[pivcol[r]] = -A[r][freecol[1]]; }
    /* f1, f2 span the null space; find lambda with det(lambda f1 + (1 - lambda) f2) = 0 (fundam.cpp run7Point) */
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    double c[4], t0, t1, t2;
    t0 = f2[4] * f2[8] - f2[5] * f2[7];
    t1 = f2[3] * f2[8] - f2[5] * f2[6];
    t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    double roots[3];
    const int n = cubic_roots(c, roots);
    for (int k = 0; k < n; k++) {
        double lambda = roots[k], mu = 1;
        const double s = f1[8] * lambda + f2[8];
        double* Fk = F + 9 * k;
        if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda *= mu; Fk[8] = 1; } else Fk[8] = 0;
        for (int i = 0; i < 8; i++) Fk[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* FMEstimatorCallback::computeError + findInliers for one model */
int find_inliers(const float* m1, const float* m2, int n, const double* F, float t, uint8_t* mask, float* errs = nullptr) {
    int good = 0;
    for (int i = 0; i < n; i++) {
        const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
        const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
        a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
        const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
        const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
        const float err = (float)std::max(e1, e2);  /* std::max(a, b) = (a < b) ? b : a: decides the NaN cases */
        if (errs) errs[i] = err;
        const int f = err <= t;
        mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

/* RANSACUpdateNumIters (ptsetreg.cpp). pow() / log() are libm's here and the device library's in k_ransac.hip: the result
 * is an integer (a rounded quotient, or the old budget if the quotient is not below it), so a last-bit difference between
 * the two libraries only shows if the quotient lies within ~1e-15 of a rounding boundary */
int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p;
    if (num < DBL_MIN) num = DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)nearbyint(num / denom);
}

}  // namespace

extern "C" {

/* cv::findFundamentalMat(pts1, pts2, FM_RANSAC, thresh, conf, mask): returns 1 and fills mask (n bytes) / F (9 doubles,
 * nullable) / *iters (nullable: RANSAC iterations executed) when a mask comes back, 0 when OpenCV returns none (fewer than
 * 7 points, or no model). 8..14 points: the LMedS registrator, 15 and more: RANSAC, as fundam.cpp dispatches. */
int orc_find_fundamental_ransac(const float* pts1, const float* pts2, int n, double thresh, double conf, uint8_t* mask, double* Fout,
                                int* iters) {
    if (iters) *iters = 0;
    if (n < 7) return 0;
    double F[27];
    if (n == 7) {
        const int nm = run_7point(pts1, pts2, F);
        if (nm <= 0) return 0;
        for (int i = 0; i < n; i++) mask[i] = 1;
        if (Fout) memcpy(Fout, F, 9 * sizeof(double));
        return 1;
    }
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    const float t = (float)(thresh * thresh);
    CvRng rng((uint64_t)-1);
    /* fundam.cpp: RANSAC needs 15 points; 8..14 go to LMeDSPointSetRegistrator(cb, 7, conf, 1000)::run -- the same getSubset /
     * runKernel, a fixed number of iterations, the model with the smallest median error (errors converted to float and
     * sorted as integers; an even count averages the two middle floats), then inliers within sigma */
    const bool lmeds = n < 15;
    int niters = lmeds ? std::max(update_num_iters(conf, 0.45, 7, 1000), 3) : 1000, maxGood = 0, iter = 0;
    double minMedian = DBL_MAX;
    std::vector<uint8_t> cur(n), best(n, 0);
    std::vector<float> errs(n);
    double bestF[9] = {0};
    float ms1[14], ms2[14];
    int idx[7];
    for (iter = 0; iter < niters; iter++) {
        /* getSubset(m1, m2, ms1, ms2, rng, 10000) */
        int attempts = 0, i = 0;
        for (; attempts < 10000; attempts++) {
            for (i = 0; i < 7 && attempts < 10000;) {
                int idx_i = 0;
                for (;;) {
                    idx_i = idx[i] = rng.uniform(0, n);
                    int j = 0;
                    for (; j < i; j++)
                        if (idx_i == idx[j]) break;
                    if (j == i) break;
                }
                ms1[2 * i] = pts1[2 * idx_i]; ms1[2 * i + 1] = pts1[2 * idx_i + 1];
                ms2[2 * i] = pts2[2 * idx_i]; ms2[2 * i + 1] = pts2[2 * idx_i + 1];
                i++;
            }
            if (i == 7 && (have_collinear(ms1, 7) || have_collinear(ms2, 7))) continue;
            break;
        }
        if (!(i == 7 && attempts < 10000)) {
            if (iter == 0) return 0;
            break;
        }
        const int nm = run_7point(ms1, ms2, F);
        if (nm <= 0) continue;
        for (int k = 0; k < nm; k++) {
            if (lmeds) {
                find_inliers(pts1, pts2, n, F + 9 * k, t, cur.data(), errs.data());
                std::vector<int32_t> bits(n);
                memcpy(bits.data(), errs.data(), (size_t)n * 4);
                std::sort(bits.begin(), bits.end());          /* std::sort(errf.ptr<int>(), errf.ptr<int>() + count) */
                memcpy(errs.data(), bits.data(), (size_t)n * 4);
                const double median = n % 2 != 0 ? (double)errs[n / 2] : (double)(errs[n / 2 - 1] + errs[n / 2]) * 0.5;
                if (median < minMedian) { minMedian = median; memcpy(bestF, F + 9 * k, sizeof bestF); }
                continue;
            }
            const int good = find_inliers(pts1, pts2, n, F + 9 * k, t, cur.data());
            if (good > std::max(maxGood, 6)) {
                std::swap(cur, best);
                memcpy(bestF, F + 9 * k, sizeof bestF);
                maxGood = good;
                niters = update_num_iters(conf, (double)(n - good) / n, 7, niters);
            }
        }
    }
    if (iters) *iters = iter;
    if (lmeds) {
        if (!(minMedian < DBL_MAX)) return 0;
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(minMedian);
        sigma = sigma > 0.001 ? sigma : 0.001;
        find_inliers(pts1, pts2, n, bestF, (float)(sigma * sigma), mask);
        if (Fout) memcpy(Fout, bestF, sizeof bestF);
        return 1;
    }
    if (maxGood <= 0) return 0;
    memcpy(mask, best.data(), n);
    if (Fout) memcpy(Fout, bestF, sizeof bestF);
    return 1;
}

/* Matcher::rejectWithF (matcher.cpp:853-881): status[i] of the n keys; cur / last: n (x, y) pairs. Returns the number of
 * flags cleared. */
int orc_reject_with_f(const float* cur_pts, const float* last_pts, int n, uint8_t* status) {
    std::vector<int> id;
    std::vector<float> p1, p2;
    for (int i = 0; i < n; i++)
        if (status[i]) {
            id.push_back(i);
            p1.push_back(cur_pts[2 * i]); p1.push_back(cur_pts[2 * i + 1]);
            p2.push_back(last_pts[2 * i]); p2.push_back(last_pts[2 * i + 1]);
        }
    if (!(n > 8)) return 0;                      /* findFundamentalMat not called: fund_status stays empty (UB in the reference) */
    const int m = (int)id.size();
    std::vector<uint8_t> fund(m > 0 ? m : 1, 1);
    const int rc = orc_find_fundamental_ransac(p1.data(), p2.data(), m, 1.0, 0.99, fund.data(), nullptr, nullptr);
    if (rc != 1) return 0;                       /* no mask came back */
    int cleared = 0;
    for (int i = 0; i < m; i++)
        if (fund[i] == 0) { status[id[i]] = 0; cleared++; }
    return cleared;
}

/* LocalBA::AddMapPointsByStereo (LocalBA.cpp:46-68): searchByOPFlow(stereo_frame, current_frame, pts, true, true), then
 * Depth[left_id] = bf / fabsf(pts[right_id].x - key[left_id].x) with left_id == right_id == the key index (the drawing
 * and imshow of :56-67 are dropped). fx is an unused argument of the reference. */
int orc_add_map_points_by_stereo(const uint8_t* img_stereo, const uint8_t* img_current, int w, int h, int stride,
                                 const tb_camera* cam_stereo, const float* keys_xy, int n, float bf, float* depth) {
    std::vector<float> cur((size_t)2 * (n > 0 ? n : 1));
    std::vector<int32_t> idx(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) depth[i] = -1.0f;
    const int m = orc_search_by_opflow(img_stereo, img_current, w, h, stride, cam_stereo, keys_xy, n, 1, 1, cur.data(), idx.data());
    if (m < 0) return m;
    for (int k = 0; k < m; k++) {
        const int i = idx[k];
        depth[i] = bf / fabsf(cur[2 * i] - keys_xy[2 * i]);
    }
    return m;
}

}  // extern "C"
