/* SURVEY 8(f) row 2, second part / row a16 -- the RANSAC stage of the optical-flow matcher and the stereo depths:
 *   Matcher::rejectWithF (src/matchers/matcher.cpp:853-881) = cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, 1.0, 0.99, mask)
 *   LocalBA::AddMapPointsByStereo's depth step (src/mapping/LocalBA.cpp:54-66): depth = bf / |x_tracked - x_key|
 *
 * cv::findFundamentalMat is OpenCV 3.3 (calib3d/fundam.cpp, ptsetreg.cpp), not part of the reference tree: restated,
 * PARITY UNPINNED. The CPU restatement (oracle_fund.cpp, test infrastructure) says line by line what is OpenCV's structure (cv::RNG((uint64)-1) sampling with
 * getSubset's duplicate redraws and collinearity retries, up to three 7-point models per sample, symmetric epipolar
 * distance against (float)(threshold^2), "first strictly better" model update, RANSACUpdateNumIters) and which two numerical
 * routines are deliberately not (null space by Gauss-Jordan instead of a Jacobi SVD with random completion, cubic roots
 * by bracketing + bisection instead of cv::solveCubic): only + - * / sqrt, so that this kernel and the oracle agree bit for bit.
 *
 * MI355X mapping: RANSAC is sequential only in two thin places -- the random stream (every sample's indices depend on how
 * many draws its predecessors took) and the budget update (a better model shortens the loop). Everything else of an
 * iteration is independent of the other iterations. One workgroup per image pair works in batches of 64 iterations:
 *   (a) ONE lane draws the batch's 64 samples from the generator, in order;
 *   (b) 64 lanes solve the 64 seven-point problems (the 7 x 9 systems live in LDS, one column of doubles per lane);
 *   (c) the four wavefronts count the inliers of the up to 192 models over all points (lane = point, ballot + popcount);
 *   (d) ONE lane replays the reference's loop over the batch -- model by model in iteration order, strictly-better rule,
 *       budget update -- and stops where the sequential loop would have stopped.
 * The result is the sequential algorithm's, iteration for iteration; models past the stopping point were computed for
 * nothing (typically the first sample already fits most points and the budget drops to a few dozen iterations).
 * Bound: latency of (a) and FP64 vector work of (c); a side path of the tracker, no SURVEY 8(d) row.
 */
#include <float.h>
#include <algorithm>
#include "tb_internal.h"
#include "tb_device.h"

#define RS_B 64            /* iterations per batch */
#define RS_T 256

struct RsShared {
    double A[63 * RS_B];           /* 7 x 9 systems, element e of lane h at A[e * RS_B + h] */
    double F[RS_B * 27];           /* up to three models per iteration */
    double bestF[9];
    float ms1[RS_B * 14], ms2[RS_B * 14];
    int nm[RS_B];                  /* models of iteration h; -1: getSubset failed there */
    int good[RS_B * 3];
    double med[RS_B * 3];          /* LMedS: median error of model k of iteration h */
    double minMedian;
    unsigned long long rng;
    unsigned long long rngBefore[RS_B];   /* generator state in front of sample h */
    int idx[RS_B * 7];                    /* drawn indices */
    int niters, maxGood, iter, done, found, firstBad, wsum[8];
};

__device__ __forceinline__ unsigned rs_next(unsigned long long& st) {
    st = (unsigned long long)(unsigned)st * 4164903690u + (unsigned)(st >> 32);
    return (unsigned)st;
}

__device__ bool rs_collinear(const float* p) {   /* haveCollinearPoints(m, 7): the 7th point against every earlier pair */
    const int i = 6;
    for (int j = 0; j < i; j++) {
        const double dx1 = (double)p[2 * j] - (double)p[2 * i], dy1 = (double)p[2 * j + 1] - (double)p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = (double)p[2 * k] - (double)p[2 * i], dy2 = (double)p[2 * k + 1] - (double)p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

/* real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3], ascending (oracle_fund.cpp cubic_roots, operation for operation) */
__device__ int rs_cubic_roots(const double c[4], double r[3]) {
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
    M = M + 1.0;
    auto bisect = [&](double lo, double hi) {
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
    const double xl = (-a - s) / 3, xh = (-a + s) / 3;
    const double pl = p(xl), ph = p(xh);
    int n = 0;
    if (pl >= 0) r[n++] = pl == 0 ? xl : bisect(-M, xl);
    if (pl > 0 && ph < 0) r[n++] = bisect(xl, xh);
    if (ph <= 0) r[n++] = ph == 0 ? xh : bisect(xh, M);
    return n;
}

/* run7Point of one sample (oracle_fund.cpp run_7point, operation for operation); A = this lane's column of the LDS systems */
__device__ int rs_run_7point(const float* m1, const float* m2, double* A /* stride RS_B */, double* F) {
#define RA(r, c) A[((r) * 9 + (c)) * RS_B]
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        RA(i, 0) = x1 * x0; RA(i, 1) = x1 * y0; RA(i, 2) = x1;
        RA(i, 3) = y1 * x0; RA(i, 4) = y1 * y0; RA(i, 5) = y1;
        RA(i, 6) = x0; RA(i, 7) = y0; RA(i, 8) = 1;
    }
    int pivcol[7], npiv = 0, freecol[9], nfree = 0;
    for (int col = 0; col < 9; col++) {
        if (npiv == 7) { freecol[nfree++] = col; continue; }
        int best = npiv;
        double bv = fabs(RA(npiv, col));
        for (int r = npiv + 1; r < 7; r++)
            if (fabs(RA(r, col)) > bv) { bv = fabs(RA(r, col)); best = r; }
        double scale = 0;
        for (int r = npiv; r < 7; r++)
            for (int k = col; k < 9; k++) scale = fabs(RA(r, k)) > scale ? fabs(RA(r, k)) : scale;
        if (!(bv > 1e-12 * scale)) { freecol[nfree++] = col; continue; }
        if (best != npiv)
            for (int k = 0; k < 9; k++) { const double t = RA(best, k); RA(best, k) = RA(npiv, k); RA(npiv, k) = t; }
        const double inv = 1.0 / RA(npiv, col);
        for (int k = 0; k < 9; k++) RA(npiv, k) *= inv;
        for (int r = 0; r < 7; r++) {
            if (r == npiv) continue;
            const double f = RA(r, col);
            if (f == 0) continue;
            for (int k = 0; k < 9; k++) RA(r, k) -= f * RA(npiv, k);
        }
        pivcol[npiv++] = col;
    }
    if (nfree != 2) return 0;
    double f1[9], f2[9];
    for (int k = 0; k < 9; k++) f1[k] = f2[k] = 0;
    /* dynamic indices into small private arrays: written as selects so that they stay in registers */
    for (int k = 0; k < 9; k++) { f1[k] = (k == freecol[0]) ? 1.0 : f1[k]; f2[k] = (k == freecol[1]) ? 1.0 : f2[k]; }
    for (int r = 0; r < 7; r++) {
        const double v1 = -RA(r, freecol[0]), v2 = -RA(r, freecol[1]);
        for (int k = 0; k < 9; k++) { f1[k] = (k == pivcol[r]) ? v1 : f1[k]; f2[k] = (k == pivcol[r]) ? v2 : f2[k]; }
    }
#undef RA
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
    const int n = rs_cubic_roots(c, roots);
    for (int k = 0; k < n; k++) {
        double lambda = roots[k], mu = 1;
        const double s = f1[8] * lambda + f2[8];
        double* Fk = F + 9 * k;
        if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda *= mu; Fk[8] = 1; } else Fk[8] = 0;
        for (int i = 0; i < 8; i++) Fk[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* FMEstimatorCallback::computeError of one point pair: (float)std::max(d1^2 s1, d2^2 s2) -- std::max(a, b) = (a < b) ? b : a,
 * which is what decides when one of the two is NaN (a degenerate F with a^2 + b^2 == 0) */
__device__ __forceinline__ float rs_error(const double* F, float x1f, float y1f, float x2f, float y2f) {
    const double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
    double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)((e1 < e2) ? e2 : e1);
}
/* inlier iff the error <= t (findInliers) */
__device__ __forceinline__ bool rs_inlier(const double* F, float x1f, float y1f, float x2f, float y2f, float t) {
    return rs_error(F, x1f, y1f, x2f, y2f) <= t;
}

/* RANSACUpdateNumIters (ptsetreg.cpp) */
__device__ int rs_update_iters(double p, double ep, int model_points, int max_iters) {
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p;
    if (num < DBL_MIN) num = DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : __double2int_rn(num / denom);
}

/* One workgroup per pair. pts1 / pts2: the tracked positions and the keys they were tracked from (n (x, y) pairs per
 * pair at stride pts_pitch); status: in/out flags. mode 0 = Matcher::rejectWithF (clears the flags of the outliers);
 * mode 1 = cv::findFundamentalMat itself on ALL n points (mask to status, F and iteration count out; host test form).
 * work: per pair pts_pitch x (2 + 2 floats + 1 int) of compacted points. flags[pair]: 0 ok, 1 no mask came back.
 * 8..14 tracked points: cv::findFundamentalMat switches to LMeDSPointSetRegistrator (fundam.cpp: RANSAC only from 15
 * points on) -- the same sampling and seven-point models over a FIXED number of iterations,
 * max(RANSACUpdateNumIters(conf, 0.45, 7, 1000), 3) = 300 at conf 0.99, the model with the smallest median error wins (the
 * errors as floats, sorted; an even count takes the mean of the two middle ones), then inliers within
 * sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(min median), at least 0.001. Here: stage (c) gives every model 16 lanes
 * (lane = point, rank by comparison with the other lanes), stage (d) replays "first strictly smaller median". */
__global__ void __launch_bounds__(RS_T)
k_ransac_f(const float* __restrict__ pts1, const float* __restrict__ pts2, uint8_t* __restrict__ status,
           const int32_t* __restrict__ counts, int pts_pitch, int mode, double thresh, double conf, float* __restrict__ work,
           int32_t* __restrict__ flags, double* __restrict__ Fout, int32_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    RsShared& S = *reinterpret_cast<RsShared*>(rs_smem);
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts ? min(counts[pair], pts_pitch) : pts_pitch;
    const float* c1 = pts1 + (size_t)pair * pts_pitch * 2;
    const float* c2 = pts2 + (size_t)pair * pts_pitch * 2;
    uint8_t* st = status + (size_t)pair * pts_pitch;
    float* p1 = work + (size_t)pair * pts_pitch * 5;
    float* p2 = p1 + (size_t)pts_pitch * 2;
    int* id = reinterpret_cast<int*>(p2 + (size_t)pts_pitch * 2);
    if (tid == 0) { flags[pair] = 0; if (iters_out) iters_out[pair] = 0; }
    if (mode == 0 && !(n > 8)) return;      /* matcher.cpp:870: findFundamentalMat is not called */

    /* tracked points, in index order (matcher.cpp:859-867) */
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += RS_T) {
        const int i = i0 + tid;
        const bool v = i < n && (mode == 1 || st[i] != 0);
        const unsigned long long bm = __ballot(v);
        if (lane == 0) S.wsum[wave] = __popcll(bm);
        __syncthreads();
        int off = m;
        for (int w = 0; w < wave; w++) off += S.wsum[w];
        const int tot = S.wsum[0] + S.wsum[1] + S.wsum[2] + S.wsum[3];
        if (v) {
            const int k = off + __popcll(bm & ((1ull << lane) - 1));
            id[k] = i;
            p1[2 * k] = c1[2 * i]; p1[2 * k + 1] = c1[2 * i + 1];
            p2[2 * k] = c2[2 * i]; p2[2 * k + 1] = c2[2 * i + 1];
        }
        m += tot;
        __syncthreads();
    }
    __threadfence_block();
    if (m < 7) { if (tid == 0 && mode == 1) flags[pair] = 1; return; }   /* no mask comes back: flags stay (UB in the reference) */
    if (m == 7) {                            /* the 7-point solver alone, every point an inlier */
        if (mode == 1) {
            if (tid == 0) {
                const int nm = rs_run_7point(p1, p2, S.A, S.F);
                S.found = nm > 0;
                if (nm > 0 && Fout) for (int k = 0; k < 9; k++) Fout[(size_t)pair * 9 + k] = S.F[k];
            }
            __syncthreads();
            if (tid < 7) st[tid] = S.found ? 1 : st[tid];
            if (tid == 0 && !S.found) flags[pair] = 1;
        }
        return;
    }
    const bool lmeds = m < 15;
    if (thresh <= 0) thresh = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    float t = (float)(thresh * thresh);
    if (tid == 0) {
        S.rng = ~0ull; S.niters = lmeds ? max(rs_update_iters(conf, 0.45, 7, 1000), 3) : 1000;
        S.maxGood = 0; S.iter = 0; S.done = 0; S.found = 0; S.minMedian = DBL_MAX;
    }
    __syncthreads();

    while (!S.done) {
        const int iter0 = S.iter;
        const int B = min(RS_B, S.niters - iter0);
        /* (a) the batch's samples: getSubset(m1, m2, ms1, ms2, rng, 10000) for B iterations in order. The generator is
         * serial, the collinearity test of checkSubset is not: one lane draws the index sets as if every sample passed
         * (7 distinct draws each), all lanes test their sample, and only if one fails -- rare -- that sample is redone
         * the sequential way (its retries consume draws) and the later ones are drawn again behind it. */
        auto draw7 = [&](unsigned long long& rng, int* idx) {
            for (int i = 0; i < 7;) {
                const int v = (int)(rs_next(rng) % (unsigned)m);
                bool dup = false;
                for (int j = 0; j < 7; j++) dup = dup || (j < i && idx[j] == v);
                if (dup) continue;
                idx[i++] = v;
            }
        };
        auto gather = [&](int h) {
            for (int i = 0; i < 7; i++) {
                const int v = S.idx[h * 7 + i];
                S.ms1[h * 14 + 2 * i] = p1[2 * v]; S.ms1[h * 14 + 2 * i + 1] = p1[2 * v + 1];
                S.ms2[h * 14 + 2 * i] = p2[2 * v]; S.ms2[h * 14 + 2 * i + 1] = p2[2 * v + 1];
            }
        };
        if (tid == 0) {
            unsigned long long rng = S.rng;
            for (int h = 0; h < B; h++) { S.rngBefore[h] = rng; draw7(rng, S.idx + h * 7); S.nm[h] = 0; }
            S.rng = rng;
        }
        __syncthreads();
        for (int from = 0; from < B;) {
            bool bad = false;
            if (tid >= from && tid < B && S.nm[tid] == 0) {
                gather(tid);
                bad = rs_collinear(S.ms1 + tid * 14) || rs_collinear(S.ms2 + tid * 14);
            }
            if (tid < 64) {   /* B <= 64: the samples' lanes are wavefront 0 */
                const unsigned long long bm = __ballot(bad);
                if (tid == 0) S.firstBad = bm ? (int)__ffsll((long long)bm) - 1 : -1;
            }
            __syncthreads();
            const int fb = S.firstBad;
            if (fb < 0) break;
            if (tid == 0) {   /* sample fb the sequential way, from the generator state in front of it */
                unsigned long long rng = S.rngBefore[fb];
                int attempts = 0;
                bool ok = false;
                for (; attempts < 10000; attempts++) {
                    draw7(rng, S.idx + fb * 7);
                    gather(fb);
                    if (rs_collinear(S.ms1 + fb * 14) || rs_collinear(S.ms2 + fb * 14)) continue;
                    ok = true;
                    break;
                }
                S.nm[fb] = ok ? 0 : -1;
                if (!ok) { for (int q = fb + 1; q < B; q++) S.nm[q] = -1; }
                else for (int h = fb + 1; h < B; h++) { S.rngBefore[h] = rng; draw7(rng, S.idx + h * 7); S.nm[h] = 0; }
                S.rng = rng;
            }
            __syncthreads();
            if (S.nm[fb] < 0) break;
            from = fb + 1;
        }
        __syncthreads();
        /* (b) one seven-point problem per lane */
        if (tid < B && S.nm[tid] == 0) S.nm[tid] = rs_run_7point(S.ms1 + tid * 14, S.ms2 + tid * 14, S.A + tid, S.F + tid * 27);
        __syncthreads();
        /* (c), LMedS: median error of every model over the m <= 14 points. Sixteen lanes per model, lane = point: the rank of a
         * point's error among the others (float bit patterns as integers, as OpenCV sorts them; ties by point index, which
         * does not change the sorted values) tells which lanes hold the middle elements */
        if (lmeds) {
            const int sub = lane >> 4, pi = lane & 15;
            for (int j0 = wave * 4; j0 < B * 3; j0 += 16) {
                const int j = j0 + sub, h = min(j, B * 3 - 1) / 3, k = min(j, B * 3 - 1) - 3 * h;
                const bool on = j < B * 3 && k < S.nm[h];      /* S.nm < 0 (failed sample): off */
                const double* F = S.F + h * 27 + k * 9;
                double Fr[9];
#pragma unroll
                for (int q = 0; q < 9; q++) Fr[q] = F[q];
                const int ip = min(pi, m - 1);
                const int key = __float_as_int(rs_error(Fr, p1[2 * ip], p1[2 * ip + 1], p2[2 * ip], p2[2 * ip + 1]));
                int rank = 0;
                for (int o = 0; o < 14; o++) {
                    const int ko = __shfl(key, (lane & 48) + o, 64);
                    rank += (o < m && (ko < key || (ko == key && o < pi))) ? 1 : 0;
                }
                /* the elements of rank m/2 - 1 and m/2, broadcast inside the model's 16 lanes */
                const unsigned long long ba = __ballot(pi < m && rank == m / 2), bb = __ballot(pi < m && rank == m / 2 - 1);
                const int la = (int)__ffsll((long long)((ba >> (lane & 48)) & 0xffffull)) - 1, lb = (int)__ffsll((long long)((bb >> (lane & 48)) & 0xffffull)) - 1;
                const float ea = __int_as_float(__shfl(key, (lane & 48) + max(la, 0), 64));
                const float eb = __int_as_float(__shfl(key, (lane & 48) + max(lb, 0), 64));
                const double median = (m & 1) ? (double)ea : (double)(eb + ea) * 0.5;
                if (on && pi == 0) S.med[j] = median;
            }
        } else
        /* (c) inliers of every model over all points: models dealt to the wavefronts, lane = point */
        for (int j = wave; j < B * 3; j += 4) {
            const int h = j / 3, k = j - 3 * h;
            if (k >= S.nm[h]) continue;
            const double* F = S.F + h * 27 + k * 9;
            double Fr[9];
#pragma unroll
            for (int q = 0; q < 9; q++) Fr[q] = F[q];
            int good = 0;
            for (int i0 = 0; i0 < m; i0 += 64) {
                const int i = i0 + lane;
                const bool in = i < m && rs_inlier(Fr, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], t);
                good += __popcll(__ballot(in));
            }
            if (lane == 0) S.good[j] = good;
        }
        __syncthreads();
        /* (d) the reference's loop over the batch, in order */
        if (tid == 0) {
            int iter = iter0, niters = S.niters, maxGood = S.maxGood;
            bool stop = false;
            for (int h = 0; h < B && !stop; h++) {
                if (!(iter < niters)) { stop = true; break; }
                if (S.nm[h] < 0) { stop = true; break; }      /* getSubset failed: return false at iteration 0, else leave the loop */
                for (int k = 0; k < S.nm[h]; k++) {
                    if (lmeds) {   /* LMeDSPointSetRegistrator::run: the first strictly smaller median */
                        if (S.med[h * 3 + k] < S.minMedian) {
                            S.minMedian = S.med[h * 3 + k];
                            for (int q = 0; q < 9; q++) S.bestF[q] = S.F[h * 27 + k * 9 + q];
                        }
                        continue;
                    }
                    const int good = S.good[h * 3 + k];
                    if (good > max(maxGood, 6)) {
                        for (int q = 0; q < 9; q++) S.bestF[q] = S.F[h * 27 + k * 9 + q];
                        maxGood = good;
                        niters = rs_update_iters(conf, (double)(m - good) / m, 7, niters);
                    }
                }
                iter++;
            }
            S.iter = iter; S.niters = niters; S.maxGood = maxGood;
            S.done = stop || !(iter < niters);
        }
        __syncthreads();
    }
    if (tid == 0 && iters_out) iters_out[pair] = S.iter;
    if (lmeds) {
        if (!(S.minMedian < DBL_MAX)) { if (tid == 0 && mode == 1) flags[pair] = 1; return; }   /* no model: no mask comes back */
        double sigma = 2.5 * 1.4826 * (1 + 5. / (m - 7)) * sqrt(S.minMedian);
        sigma = sigma > 0.001 ? sigma : 0.001;
        t = (float)(sigma * sigma);
    } else if (S.maxGood <= 0) { if (tid == 0 && mode == 1) flags[pair] = 1; return; }   /* no mask comes back */
    if (tid < 9 && Fout) Fout[(size_t)pair * 9 + tid] = S.bestF[tid];
    double Fr[9];
#pragma unroll
    for (int q = 0; q < 9; q++) Fr[q] = S.bestF[q];
    for (int i = tid; i < m; i += RS_T) {
        const bool in = rs_inlier(Fr, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], t);
        if (mode == 1) st[id[i]] = in ? 1 : 0;
        else if (!in) st[id[i]] = 0;
    }
}

size_t tbk_ransac_work_bytes(int npairs, int pts_pitch) { return (size_t)npairs * (size_t)std::max(pts_pitch, 1) * 5 * sizeof(float); }

int tbk_ransac_f(tb_ctx* ctx, int npairs, const float* d_pts1, const float* d_pts2, uint8_t* d_status, const int32_t* d_counts,
                 int pts_pitch, int mode, double thresh, double conf, void* d_work, int32_t* d_flags, double* d_F, int32_t* d_iters) {
    if (npairs <= 0 || pts_pitch <= 0) return TB_OK;
    const size_t lds = sizeof(RsShared);
    TB_HIP(ctx, hipFuncSetAttribute((const void*)k_ransac_f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tb_prof_begin(ctx, "k_ransac_f");
    hipLaunchKernelGGL(k_ransac_f, dim3(npairs), dim3(RS_T), lds, ctx->stream, d_pts1, d_pts2, d_status, d_counts, pts_pitch, mode,
                       thresh, conf, (float*)d_work, d_flags, d_F, d_iters);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* LocalBA::AddMapPointsByStereo, LocalBA.cpp:54-66: Depth[i] = bf / fabsf(pts[i].x - key[i].x) for the matched keys, -1 else */
__global__ void __launch_bounds__(256)
k_stereo_depth(const float* __restrict__ cur, const float* __restrict__ keys, const uint8_t* __restrict__ status,
               const int32_t* __restrict__ counts, int pts_pitch, float bf, float* __restrict__ depth) {
    const int pair = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts ? min(counts[pair], pts_pitch) : pts_pitch;
    if (i >= pts_pitch) return;
    const size_t o = (size_t)pair * pts_pitch + i;
    float d = -1.0f;
    if (i < n && status[o]) d = TB_FDIV(bf, fabsf(TB_FSUB(cur[2 * o], keys[2 * o])));
    depth[o] = d;
}

int tbk_stereo_depth(tb_ctx* ctx, int npairs, const float* d_cur, const float* d_keys, const uint8_t* d_status, const int32_t* d_counts,
                     int pts_pitch, float bf, float* d_depth) {
    if (npairs <= 0 || pts_pitch <= 0) return TB_OK;
    tb_prof_begin(ctx, "k_stereo_depth");
    hipLaunchKernelGGL(k_stereo_depth, dim3((pts_pitch + 255) / 256, npairs), dim3(256), 0, ctx->stream, d_cur, d_keys, d_status, d_counts,
                       pts_pitch, bf, d_depth);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
