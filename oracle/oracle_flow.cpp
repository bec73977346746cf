/* TEST INFRASTRUCTURE ONLY -- CPU oracle of the optical-flow matcher (first part).
 *
 * Reference call site: Matcher::searchByOPFlow, src/matchers/matcher.cpp:724-768:
 *     cv::calcOpticalFlowPyrLK(img2, img1, F2->GetCVKeys(), cur_points, status, err, cv::Size(21, 21), 3);
 *     status[i] &= IsInFrame(Vector2i(cur_points[i]))            (:746-748, CameraModel.h:33-39)
 *     [reject: rejectWithF -> cv::findFundamentalMat(FM_RANSAC)]  (:750-754, NOT restated: needs cv::RNG + the 7-point
 *                                                                  solver; reject=true is unsupported)
 *     matches = {(i, i) : status[i]}                              (:756-766)
 *
 * cv::calcOpticalFlowPyrLK lives in OpenCV 3.3 (imgproc/video, lkpyramid.cpp), which is NOT in /root/reference and not
 * installed here: the algorithm below is restated from its published structure (pyrDown 5x5 Gaussian pyramid with
 * BORDER_REFLECT_101 padding, Scharr derivatives with zero padding, W_BITS = 14 fixed-point bilinear weights,
 * iterative 2x2 solve, COUNT 30 + EPS 0.01 criteria, minEigThreshold 1e-4, final L1 error) -- PARITY UNPINNED: no
 * golden vector of the reference covers it.  One deliberate difference: OpenCV accumulates the window sums
 * (A11, A12, A22, b1, b2) in float, in an order that depends on its SIMD path; here they are exact 64-bit integer sums
 * converted to float once, so that the result does not depend on the summation order (and the HIP kernel can match it
 * bit for bit). */
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline int refl101(int p, int n) { /* BORDER_REFLECT_101, |offset| < n */
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}

struct Level { int w, h; std::vector<uint8_t> px; };

inline int at(const Level& L, int x, int y) { return L.px[(size_t)refl101(y, L.h) * L.w + refl101(x, L.w)]; }

/* cv::pyrDown, 8-bit: separable [1 4 6 4 1] / 16 twice, (sum + 128) >> 8, size ((w + 1) / 2, (h + 1) / 2) */
Level pyr_down(const Level& s) {
    Level d;
    d.w = (s.w + 1) / 2; d.h = (s.h + 1) / 2;
    d.px.resize((size_t)d.w * d.h);
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < d.h; y++)
        for (int x = 0; x < d.w; x++) {
            int sum = 0;
            for (int j = -2; j <= 2; j++)
                for (int i = -2; i <= 2; i++) sum += k[i + 2] * k[j + 2] * at(s, 2 * x + i, 2 * y + j);
            d.px[(size_t)y * d.w + x] = (uint8_t)((sum + 128) >> 8);
        }
    return d;
}

/* Scharr derivatives at an integer position (calcSharrDeriv); zero outside the image (the derivative buffer is padded
 * with BORDER_CONSTANT) */
inline void scharr(const Level& L, int x, int y, int* dx, int* dy) {
    if (x < 0 || x >= L.w || y < 0 || y >= L.h) { *dx = 0; *dy = 0; return; }
    const int a00 = at(L, x - 1, y - 1), a01 = at(L, x, y - 1), a02 = at(L, x + 1, y - 1);
    const int a10 = at(L, x - 1, y), a12 = at(L, x + 1, y);
    const int a20 = at(L, x - 1, y + 1), a21 = at(L, x, y + 1), a22 = at(L, x + 1, y + 1);
    *dx = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
    *dy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
}

inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

struct Weights { int w00, w01, w10, w11; };
inline Weights weights(float a, float b) {
    Weights W;
    W.w00 = (int)lrintf((1.f - a) * (1.f - b) * 16384.f);
    W.w01 = (int)lrintf(a * (1.f - b) * 16384.f);
    W.w10 = (int)lrintf((1.f - a) * b * 16384.f);
    W.w11 = 16384 - W.w00 - W.w01 - W.w10;
    return W;
}

}  // namespace

extern "C" {

/* One pyramid level of cv::buildOpticalFlowPyramid (exposed for the kernel's own test). */
int orc_pyr_down(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride) {
    Level s; s.w = w; s.h = h; s.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(&s.px[(size_t)y * w], src + (size_t)y * stride, w);
    Level d = pyr_down(s);
    for (int y = 0; y < d.h; y++) memcpy(dst + (size_t)y * dstride, &d.px[(size_t)y * d.w], d.w);
    return 0;
}

/* cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(win, win), max_level) with the default
 * criteria (30 iterations, eps 0.01), flags 0, minEigThreshold 1e-4 -- the call of matcher.cpp:744. */
int orc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts, int n,
                            int win, int max_level, float* next_pts, uint8_t* status, float* err) {
    if (win < 3 || !(win & 1) || max_level < 0) return -1;
    std::vector<Level> P(1), Q(1);
    P[0].w = Q[0].w = w; P[0].h = Q[0].h = h;
    P[0].px.resize((size_t)w * h); Q[0].px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        memcpy(&P[0].px[(size_t)y * w], prev + (size_t)y * stride, w);
        memcpy(&Q[0].px[(size_t)y * w], next + (size_t)y * stride, w);
    }
    if (w <= win || h <= win) return -1;
    for (int l = 1; l <= max_level; l++) { /* buildOpticalFlowPyramid stops before a level no larger than the window */
        const int lw = (P[l - 1].w + 1) / 2, lh = (P[l - 1].h + 1) / 2;
        if (lw <= win || lh <= win) { max_level = l - 1; break; }
        P.push_back(pyr_down(P[l - 1]));
        Q.push_back(pyr_down(Q[l - 1]));
    }
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (win - 1) * 0.5f;
    std::vector<int> Iw((size_t)win * win), Ix((size_t)win * win), Iy((size_t)win * win);
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0; }
    for (int level = max_level; level >= 0; level--) {
        const Level& I = P[level];
        const Level& J = Q[level];
        for (int i = 0; i < n; i++) {
            float px = prev_pts[2 * i] * (float)(1. / (1 << level)), py = prev_pts[2 * i + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == max_level) { nx = px; ny = py; }
            else { nx = next_pts[2 * i] * 2.f; ny = next_pts[2 * i + 1] * 2.f; }
            next_pts[2 * i] = nx; next_pts[2 * i + 1] = ny;
            px -= half; py -= half;
            int ix = (int)floorf(px), iy = (int)floorf(py);
            if (ix < -win || ix >= I.w || iy < -win || iy >= I.h) {
                if (level == 0) { status[i] = 0; if (err) err[i] = 0; }
                continue;
            }
            Weights W = weights(px - ix, py - iy);
            long long A11 = 0, A12 = 0, A22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int X = ix + x, Y = iy + y;
                    int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
                    scharr(I, X, Y, &dx00, &dy00); scharr(I, X + 1, Y, &dx01, &dy01);
                    scharr(I, X, Y + 1, &dx10, &dy10); scharr(I, X + 1, Y + 1, &dx11, &dy11);
                    const int iv = descale(at(I, X, Y) * W.w00 + at(I, X + 1, Y) * W.w01 + at(I, X, Y + 1) * W.w10 + at(I, X + 1, Y + 1) * W.w11, 9);
                    const int xv = descale(dx00 * W.w00 + dx01 * W.w01 + dx10 * W.w10 + dx11 * W.w11, 14);
                    const int yv = descale(dy00 * W.w00 + dy01 * W.w01 + dy10 * W.w10 + dy11 * W.w11, 14);
                    Iw[(size_t)y * win + x] = iv; Ix[(size_t)y * win + x] = xv; Iy[(size_t)y * win + x] = yv;
                    A11 += (long long)xv * xv; A12 += (long long)xv * yv; A22 += (long long)yv * yv;
                }
            const float a11 = (float)A11 * FLT_SCALE, a12 = (float)A12 * FLT_SCALE, a22 = (float)A22 * FLT_SCALE;
            float D = a11 * a22 - a12 * a12;
            const float minEig = (a22 + a11 - sqrtf((a11 - a22) * (a11 - a22) + 4.f * a12 * a12)) / (float)(2 * win * win);
            if (minEig < 1e-4f || D < FLT_EPSILON) {
                if (level == 0) status[i] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < 30; j++) {
                const int jx = (int)floorf(nx), jy = (int)floorf(ny);
                if (jx < -win || jx >= J.w || jy < -win || jy >= J.h) {
                    if (level == 0) status[i] = 0;
                    break;
                }
                W = weights(nx - jx, ny - jy);
                long long B1 = 0, B2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = jx + x, Y = jy + y;
                        const int diff = descale(at(J, X, Y) * W.w00 + at(J, X + 1, Y) * W.w01 + at(J, X, Y + 1) * W.w10 + at(J, X + 1, Y + 1) * W.w11, 9) -
                                         Iw[(size_t)y * win + x];
                        B1 += (long long)diff * Ix[(size_t)y * win + x];
                        B2 += (long long)diff * Iy[(size_t)y * win + x];
                    }
                const float b1 = (float)B1 * FLT_SCALE, b2 = (float)B2 * FLT_SCALE;
                const float dx = (a12 * b2 - a22 * b1) * D, dy = (a12 * b1 - a11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[2 * i] = nx + half; next_pts[2 * i + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= 0.01 * 0.01) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                    next_pts[2 * i] -= dx * 0.5f; next_pts[2 * i + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[i] && level == 0) { /* L1 error of the final position */
                const float fx = next_pts[2 * i] - half, fy = next_pts[2 * i + 1] - half;
                const int jx = (int)floorf(fx), jy = (int)floorf(fy);
                if (jx < -win || jx >= J.w || jy < -win || jy >= J.h) { status[i] = 0; continue; }
                W = weights(fx - jx, fy - jy);
                long long E = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = jx + x, Y = jy + y;
                        const int diff = descale(at(J, X, Y) * W.w00 + at(J, X + 1, Y) * W.w01 + at(J, X, Y + 1) * W.w10 + at(J, X + 1, Y + 1) * W.w11, 9) -
                                         Iw[(size_t)y * win + x];
                        E += std::abs(diff);
                    }
                if (err) err[i] = (float)E / (float)(32 * win * win);
            }
        }
    }
    return max_level;
}

/* Matcher::searchByOPFlow(F1, F2, cur_points, equalized, reject = false), matcher.cpp:724-768: img2 / keys2 belong
 * to F2 (the frame whose keys are tracked), img1 / cam1 to F1. Returns the number of matches (queryIdx = trainIdx = i). */
int orc_clahe(const uint8_t* src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst, int dstride);

int orc_search_by_opflow(const uint8_t* img1, const uint8_t* img2, int w, int h, int stride, const tb_camera* cam1,
                         const float* keys2_xy, int n, int equalized, int reject, float* cur_points, int32_t* match_idx) {
    std::vector<uint8_t> status(n > 0 ? n : 1), eq;
    if (equalized) { /* matcher.cpp:736-739: img1 = F1->Equalize() */
        eq.resize((size_t)stride * h);
        if (orc_clahe(img1, w, h, stride, 3.0, 8, 8, eq.data(), stride) < 0) return -1;
        img1 = eq.data();
    }
    std::vector<float> err(n > 0 ? n : 1);
    if (orc_optical_flow_pyr_lk(img2, img1, w, h, stride, keys2_xy, n, 21, 3, cur_points, status.data(), err.data()) < 0) return -1;
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;
        const float x = cur_points[2 * i], y = cur_points[2 * i + 1];
        bool in = std::fabs(x) < 2147483648.f && std::fabs(y) < 2147483648.f; /* else cvttss2si -> INT_MIN: not in frame */
        if (in) {
            const int u = (int)x, v = (int)y;
            in = u >= 0 && u < (int)((float)cam1->width * 1.f) && v >= 0 && v < (int)((float)cam1->height * 1.f);
        }
        if (!in) status[i] = 0;
    }
    if (reject) { /* matcher.cpp:751-755: rejectWithF(cur_points, F2->GetCVKeys(), status) */
        const int rc = orc_reject_with_f(cur_points, keys2_xy, n, status.data());
        if (rc == -2) return -2;
    }
    int m = 0;
    for (int i = 0; i < n; i++)
        if (status[i]) match_idx[m++] = i;
    return m;
}

}  // extern "C"

/* Frame::Equalize, src/types/Frame.cpp:453-458: cv::createCLAHE(3.0, Size(8, 8))->apply(level 0, out).
 * cv::CLAHE is OpenCV 3.3 (imgproc/clahe.cpp), not in the reference tree: restated from its published structure, PARITY
 * UNPINNED -- per-tile histogram, clip at max(1, (int)(clipLimit * tileArea / 256)), the excess redistributed as
 * excess / 256 to every bin plus one to each of the first (excess % 256) bins (the 3.3-era rule; later releases spread
 * the remainder with a stride), LUT = saturate(cvRound(cumsum * 255 / tileArea)), then per pixel the bilinear blend of
 * the four neighbouring tiles' LUT values in float. Images whose size is not a multiple of the tile grid are extended
 * to the right / bottom with BORDER_REFLECT_101 for the histograms only. */
extern "C" int orc_clahe(const uint8_t* src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst,
                         int dstride) {
    if (w < 1 || h < 1 || tiles_x < 1 || tiles_y < 1) return -1;
    int ew = w, eh = h;
    if (w % tiles_x || h % tiles_y) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    if (ew - w >= w || eh - h >= h) return -1; /* reflection would leave the image */
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    const float lutScale = (float)255 / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / 256); if (clip < 1) clip = 1; }
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            int hist[256] = {0};
            for (int y = ty * th; y < (ty + 1) * th; y++)
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[src[(size_t)refl101(y, h) * stride + refl101(x, w)]]++;
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++)
                    if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
                const int batch = clipped / 256, residual = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                for (int i = 0; i < residual; i++) hist[i]++;
            }
            uint8_t* L = &lut[((size_t)ty * tiles_x + tx) * 256];
            int sum = 0;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                const long r = lrintf((float)sum * lutScale);
                L[i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = ty1 < 0 ? 0 : ty1; ty2 = ty2 > tiles_y - 1 ? tiles_y - 1 : ty2;
        for (int x = 0; x < w; x++) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            tx1 = tx1 < 0 ? 0 : tx1; tx2 = tx2 > tiles_x - 1 ? tiles_x - 1 : tx2;
            const int v = src[(size_t)y * stride + x];
            const float l11 = lut[((size_t)ty1 * tiles_x + tx1) * 256 + v], l12 = lut[((size_t)ty1 * tiles_x + tx2) * 256 + v];
            const float l21 = lut[((size_t)ty2 * tiles_x + tx1) * 256 + v], l22 = lut[((size_t)ty2 * tiles_x + tx2) * 256 + v];
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            const long r = lrintf(res);
            dst[(size_t)y * dstride + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    return 0;
}
