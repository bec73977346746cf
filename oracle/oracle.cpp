/* TEST INFRASTRUCTURE ONLY -- CPU oracle, extractor stages.
 *
 * A dependency-free restatement of the reference's pyramid + ORB / FAST
 * extraction path, with OpenCV 3.3 / fast_lib semantics restated from their
 * published algorithms (neither library is in the image).
 * PARITY STATUS: parity unpinned against genuine OpenCV -- see orc_math.h.
 * Never linked, imported or executed by the product path.
 */
#include "oracle.h"
#include "orc_math.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

using namespace orc;

namespace {

const int8_t k_pattern[1024] = {
#include "orb_pattern.inc"
};

/* OpenCV makeOffsets(pixel, step, 16): Bresenham circle r=3, (dx,dy) in ring order. */
const int k_ring[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                           {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* Largest t for which p is a FAST corner with `arc` contiguous ring pixels all
 * brighter than p+t or all darker than p-t; -1 if it is a corner for no t >= 0.
 * Equals OpenCV cornerScore<16> (arc 9) and fast_corner_score_10 (arc 10) on corners. */
inline int fast_score(const uint8_t* p, int stride, int arc) {
    int d[32];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = d[k + 16] = v - p[k_ring[k][1] * stride + k_ring[k][0]];
    int best = 0; /* max over arcs of min |d| with a common sign */
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int i = 1; i < arc; i++) {
            mn = std::min(mn, d[k + i]);
            mx = std::max(mx, d[k + i]);
        }
        best = std::max(best, mn);
        best = std::max(best, -mx);
    }
    return best - 1;
}

const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

/* ORBExtractor::ORBExtractor umax table, ORBextractor.cpp:389-404 */
void check_umax() {
    static bool done = false;
    if (done) return;
    int umax[17] = {0};
    const int H = 15;
    int v, v0, vmax = cv_floor(H * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(H * std::sqrt(2.f) / 2);
    const double hp2 = H * H;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = H, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
    for (v = 0; v <= H; ++v) assert(umax[v] == k_umax[v]);
    (void)umax;
    done = true;
}

inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

}  // namespace

extern "C" {

/* Frame::Frame, Frame.cpp:18-29: all float32, level 0 = 1. */
int orc_scale_factors(int n, float scale, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2) {
    if (n < 1 || !sf) return TB_EINVAL;
    for (int i = 0; i < n; i++) {
        sf[i] = 1.f;
        if (inv_sf) inv_sf[i] = 1.f;
        if (sigma2) sigma2[i] = 1.f;
        if (inv_sigma2) inv_sigma2[i] = 1.f;
    }
    float isf = 1.f;
    for (int i = 1; i < n; i++) {
        sf[i] = sf[i - 1] * scale;
        isf = isf / scale;
        if (inv_sf) inv_sf[i] = isf;
        float s2 = sf[i] * sf[i];
        if (sigma2) sigma2[i] = s2;
        if (inv_sigma2) inv_sigma2[i] = 1.f / s2;
    }
    return TB_OK;
}

/* Frame::ComputePyramid, Frame.cpp:423-424: cv::Size(image.cols * scale, image.rows * scale),
 * int * float -> float -> int (truncation), always from the level-0 size. */
int orc_pyramid_sizes(int w, int h, int n, const float* sf, int* ws, int* hs) {
    if (n < 1 || !sf || !ws || !hs) return TB_EINVAL;
    for (int i = 0; i < n; i++) {
        if (i == 0) {
            ws[i] = w;
            hs[i] = h;
        } else {
            ws[i] = (int)((float)w * sf[i]);
            hs[i] = (int)((float)h * sf[i]);
        }
    }
    return TB_OK;
}

/* cv::resize 8UC1 INTER_LINEAR, OpenCV 3.3 (SURVEY App. A.1 [memory]):
 * scale = 1/(dst/src) in double; fx = (float)((dx+0.5)*scale-0.5); 11-bit coefficients;
 * vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2. */
int orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                         uint8_t* dst, int dw, int dh, int dstride) {
    if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return TB_EINVAL;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), xofs1(dw);
    std::vector<short> ialpha(2 * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xofs1[dx] = std::min(sx + 1, sw - 1);
        ialpha[2 * dx] = (short)cv_round((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = (short)cv_round(fx * 2048.f);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        short b0 = (short)cv_round((1.f - fy) * 2048.f);
        short b1 = (short)cv_round(fy * 2048.f);
        int sy0 = std::min(std::max(sy, 0), sh - 1);
        int sy1 = std::min(std::max(sy + 1, 0), sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            row0[dx] = S0[xofs[dx]] * ialpha[2 * dx] + S0[xofs1[dx]] * ialpha[2 * dx + 1];
            row1[dx] = S1[xofs[dx]] * ialpha[2 * dx] + S1[xofs1[dx]] * ialpha[2 * dx + 1];
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
    return TB_OK;
}

int orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int arc, int16_t* out) {
    if (!img || !out || (arc != 9 && arc != 10)) return TB_EINVAL;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            if (y < 3 || y >= h - 3 || x < 3 || x >= w - 3) out[(size_t)y * w + x] = -1;
            else out[(size_t)y * w + x] = (int16_t)fast_score(img + (size_t)y * stride + x, stride, arc);
        }
    return TB_OK;
}

/* cv::FAST(img, kps, th, nms), TYPE_9_16 (SURVEY App. A.2 [memory]): scan rows 3..h-4, cols 3..w-4;
 * with nms keep a corner iff its score is strictly greater than the 8 neighbours' scores, where
 * non-corner / unscanned neighbours count 0; raster output order. */
int orc_fast9(const uint8_t* img, int w, int h, int stride, int th, int nms, tb_corner* out, int cap) {
    if (!img || w < 0 || h < 0) return TB_EINVAL;
    th = std::min(std::max(th, 0), 255);
    if (w < 7 || h < 7) return 0;
    std::vector<uint8_t> sc((size_t)w * h, 0), isc((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = fast_score(img + (size_t)y * stride + x, stride, 9);
            if (s >= th) {
                isc[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = (uint8_t)s;
            }
        }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!isc[(size_t)y * w + x]) continue;
            int s = sc[(size_t)y * w + x];
            bool keep = true;
            if (nms) {
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        if (!(s > sc[(size_t)(y + dy) * w + x + dx])) { keep = false; break; }
                    }
            }
            if (keep) {
                if (out) {
                    if (n >= cap) return TB_ECAPACITY;
                    out[n].x = x; out[n].y = y; out[n].score = s;
                }
                n++;
            }
        }
    return n;
}

/* fast::fast_corner_detect_10(_sse2) + fast_corner_score_10 + fast_nonmax_3x3 (uzh-rpg fast,
 * un-vendored; published algorithm restated): FAST-10 at barrier th over x in [3,w-3), y in [3,h-3);
 * score = largest barrier still a corner; a corner survives iff no adjacent CORNER has score >= its own. */
int orc_fast10_nms(const uint8_t* img, int w, int h, int stride, int th, tb_corner* out, int cap) {
    if (!img || w < 0 || h < 0) return TB_EINVAL;
    if (w < 7 || h < 7) return 0;
    std::vector<int16_t> sc((size_t)w * h, -1);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = fast_score(img + (size_t)y * stride + x, stride, 10);
            if (s >= th) sc[(size_t)y * w + x] = (int16_t)s;
        }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t)y * w + x];
            if (s < 0) continue;
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    int q = sc[(size_t)(y + dy) * w + x + dx];
                    if (q >= 0 && q >= s) { keep = false; break; }
                }
            if (keep) {
                if (out) {
                    if (n >= cap) return TB_ECAPACITY;
                    out[n].x = x; out[n].y = y; out[n].score = s;
                }
                n++;
            }
        }
    return n;
}

/* cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on 8U (SURVEY App. A.5 [memory]):
 * float kernel exp(-x^2/8)/sum -> x256 rounded = 18,34,49,55,49,34,18; exact int32 separable
 * accumulation; dst = saturate((acc + 2^15) >> 16). */
int orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (!src || !dst || w < 1 || h < 1) return TB_EINVAL;
    int k[7];
    {
        double kd[7], sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = i - 3;
            kd[i] = std::exp(-0.5 * x * x / 4.0);
            sum += kd[i];
        }
        for (int i = 0; i < 7; i++) {
            float kf = (float)(kd[i] / sum); /* getGaussianKernel(..., CV_32F) */
            k[i] = cv_round((double)kf * 256.0);
        }
        assert(k[0] == 18 && k[1] == 34 && k[2] == 49 && k[3] == 55);
    }
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++) acc += k[i] * src[(size_t)y * sstride + reflect101(x + i - 3, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++) acc += k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    return TB_OK;
}

/* IC_Angle, ORBextractor.cpp:17-44 */
float orc_ic_angle(const uint8_t* img, int stride, float x, float y) {
    check_umax();
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (std::ptrdiff_t)cv_round(y) * stride + cv_round(x);
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0;
        int d = k_umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, ORBextractor.cpp:48-87 */
void orc_orb_descriptor(const uint8_t* img, int stride, float x, float y, float angle_deg, uint8_t* desc) {
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = orc_cosf(angle), b = orc_sinf(angle);
    const uint8_t* center = img + (std::ptrdiff_t)cv_round(y) * stride + cv_round(x);
    const int8_t* pat = k_pattern;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int j = 0; j < 8; j++, pat += 4) {
            float x0 = pat[0], y0 = pat[1], x1 = pat[2], y1 = pat[3];
            float r0 = x0 * b, r0b = y0 * a, c0 = x0 * a, c0b = y0 * b;
            float r1 = x1 * b, r1b = y1 * a, c1 = x1 * a, c1b = y1 * b;
            int t0 = center[cv_round(r0 + r0b) * stride + cv_round(c0 - c0b)];
            int t1 = center[cv_round(r1 + r1b) * stride + cv_round(c1 - c1b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBExtractor::operator() quota, ORBextractor.cpp:919-930 (factor is sf[1]). */
int orc_orb_quotas(int nlevels, const float* sf, int target, int* quotas) {
    if (nlevels < 2 || !sf || !quotas) return TB_EINVAL; /* sf[1] is read: nlevels==1 is UB in the reference */
    float nDesired = target * (1 - sf[1]) / (1 - (float)std::pow((double)sf[1], (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        quotas[level] = cv_round(nDesired);
        sum += quotas[level];
        nDesired *= sf[1];
    }
    quotas[nlevels - 1] = std::max(target - sum, 0);
    return TB_OK;
}

/* ComputeKeyPointsOctTree cell loop, ORBextractor.cpp:747-804. Output coordinates are relative
 * to (minBorderX, minBorderY) = (16,16) exactly as vToDistributeKeys holds them. */
int orc_orb_candidates(const uint8_t* img, int w, int h, int stride, float init_th, float min_th,
                       tb_corner* out, int cap) {
    if (!img) return TB_EINVAL;
    const int minBorderX = 19 - 3, minBorderY = minBorderX;
    const int maxBorderX = w - 19 + 3, maxBorderY = h - 19 + 3;
    const float W = 30;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return 0; /* reference divides by zero here; no cells -> no keypoints */
    const int wCell = (int)std::ceil(width / (float)nCols);
    const int hCell = (int)std::ceil(height / (float)nRows);
    int n = 0;
    std::vector<tb_corner> cell(70 * 70);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)minBorderY + (float)i * (float)hCell;
        float maxY = iniY + (float)hCell + 6.f;
        if (iniY >= (float)maxBorderY - 3.f) continue;
        if (maxY > (float)maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = minBorderX + (float)(j * wCell);
            float maxX = iniX + (float)wCell + 6.f;
            if (iniX >= (float)maxBorderX - 6.f) continue;
            if (maxX > (float)maxBorderX) maxX = (float)maxBorderX;
            const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
            const uint8_t* roi = img + (size_t)y0 * stride + x0;
            int m = orc_fast9(roi, x1 - x0, y1 - y0, stride, (int)init_th, 1, cell.data(), (int)cell.size());
            if (m == 0) m = orc_fast9(roi, x1 - x0, y1 - y0, stride, (int)min_th, 1, cell.data(), (int)cell.size());
            if (m < 0) return m;
            for (int k = 0; k < m; k++) {
                if (out) {
                    if (n >= cap) return TB_ECAPACITY;
                    out[n].x = cell[k].x + j * wCell;
                    out[n].y = cell[k].y + i * hCell;
                    out[n].score = cell[k].score;
                }
                n++;
            }
        }
    }
    return n;
}

}  // extern "C"

namespace {

struct Key {
    float x, y, response;
    int idx; /* index into the candidate array */
};
struct ExitKey { float x, y; };

/* ExtractorNode, ORBextractor.h:9-22 */
struct Node {
    std::vector<Key> keys;
    std::vector<ExitKey> exit_keys;
    int ULx = 0, ULy = 0, URx = 0, URy = 0, BLx = 0, BLy = 0, BRx = 0, BRy = 0;
    std::list<Node>::iterator lit;
    bool no_more = false;
    long seq = 0; /* creation order: replaces the pointer tie-break of ORBextractor.cpp:643 */
};

/* ExtractorNode::DivideNode, ORBextractor.cpp:416-491 */
void divide_node(const Node& p, Node& n1, Node& n2, Node& n3, Node& n4) {
    const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;

    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;

    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;

    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;

    for (const Key& kp : p.keys) {
        if (kp.x < (float)n1.URx) {
            if (kp.y < (float)n1.BRy) n1.keys.push_back(kp);
            else n3.keys.push_back(kp);
        } else if (kp.y < (float)n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    for (const ExitKey& kp : p.exit_keys) {
        if (kp.x < (float)n1.URx) {
            if (kp.y < (float)n1.BRy) n1.exit_keys.push_back(kp);
            else n3.exit_keys.push_back(kp);
        } else if (kp.y < (float)n1.BRy) n2.exit_keys.push_back(kp);
        else n4.exit_keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.no_more = true;
    if (n2.keys.size() == 1) n2.no_more = true;
    if (n3.keys.size() == 1) n3.no_more = true;
    if (n4.keys.size() == 1) n4.no_more = true;
}

}  // namespace

extern "C" {

/* ORBExtractor::DistributeOctTree, ORBextractor.cpp:494-733.
 * Deviation: ties between equal-size nodes in the "expand largest first"
 * phase are broken by creation order (later-created first) instead of by heap address;
 * nIni < 1 (portrait level, reference crashes) is clamped to 1. */
int orc_distribute_octtree(const tb_corner* cand, int ncand, const tb_keypoint* exit_keys, int nexit,
                           int min_x, int max_x, int min_y, int max_y, int quota,
                           tb_corner* out, int cap) {
    const int N = quota;
    if (ncand == 0) return 0;
    int nIni = (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y));
    if (nIni < 1) nIni = 1;
    const float hX = static_cast<float>(max_x - min_x) / nIni;
    long seq = 0;

    std::list<Node> lNodes;
    std::vector<Node*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = max_y - min_y;
        ni.BRx = ni.URx; ni.BRy = max_y - min_y;
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (int k = 0; k < ncand; k++) {
        Key kp{(float)cand[k].x, (float)cand[k].y, (float)cand[k].score, k};
        int bin = (int)(kp.x / hX);
        if (bin >= nIni) bin = nIni - 1; /* cannot happen for in-range candidates; guards the oracle */
        vpIniNodes[bin]->keys.push_back(kp);
    }
    for (int k = 0; k < nexit; k++) vpIniNodes[0]->exit_keys.push_back(ExitKey{exit_keys[k].x, exit_keys[k].y});

    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->keys.size() == 1) { lit->no_more = true; lit++; }
        else if (lit->keys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }

    bool bFinish = false;
    typedef std::pair<int, std::pair<long, Node*>> SizeNode; /* (size, (seq, node)) */
    std::vector<SizeNode> vSizeAndNode;

    auto push_child = [&](Node& c, std::vector<SizeNode>& vec, int* nToExpand) {
        if (c.keys.size() > 0) {
            c.seq = seq++;
            lNodes.push_front(c);
            if (c.keys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                vec.push_back(std::make_pair((int)c.keys.size(), std::make_pair(lNodes.front().seq, &lNodes.front())));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndNode.clear();
        while (lit != lNodes.end()) {
            if (lit->no_more) { lit++; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            push_child(n1, vSizeAndNode, &nToExpand);
            push_child(n2, vSizeAndNode, &nToExpand);
            push_child(n3, vSizeAndNode, &nToExpand);
            push_child(n4, vSizeAndNode, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeNode> vPrev = vSizeAndNode;
                vSizeAndNode.clear();
                std::sort(vPrev.begin(), vPrev.end(),
                          [](const SizeNode& a, const SizeNode& b) {
                              if (a.first != b.first) return a.first < b.first;
                              return a.second.first < b.second.first;
                          });
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    Node* parent = vPrev[j].second.second;
                    divide_node(*parent, n1, n2, n3, n4);
                    push_child(n1, vSizeAndNode, nullptr);
                    push_child(n2, vSizeAndNode, nullptr);
                    push_child(n3, vSizeAndNode, nullptr);
                    push_child(n4, vSizeAndNode, nullptr);
                    lNodes.erase(parent->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }

    int n = 0;
    for (Node& node : lNodes) {
        const Key* best = &node.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < node.keys.size(); k++)
            if (node.keys[k].response > maxResponse) {
                best = &node.keys[k];
                maxResponse = node.keys[k].response;
            }
        bool accept = true;
        for (const ExitKey& pt : node.exit_keys)
            if ((pt.x - best->x) * (pt.x - best->x) + (pt.y - best->y) * (pt.y - best->y) < 400) accept = false;
        if (accept) {
            if (out) {
                if (n >= cap) return TB_ECAPACITY;
                out[n] = cand[best->idx];
            }
            n++;
        }
    }
    return n;
}

/* ORBExtractor::operator() (ORBextractor.cpp:906-978) and AddPoints (:840-904). */
int orc_orb_extract(const uint8_t* const* levels, const int* ws, const int* hs, const int* strides,
                    int nlevels, const float* sf, int target, float init_th, float min_th,
                    const tb_keypoint* exit_keys, int nexit, int use_quotas, int* quotas_inout,
                    tb_keypoint* kps, uint8_t* desc, int cap) {
    if (!levels || !ws || !hs || !strides || !sf || !quotas_inout || nlevels < 1) return TB_EINVAL;
    if (!levels[0] || ws[0] < 1 || hs[0] < 1) return 0; /* images.at(0).empty(): silent return */
    if (!use_quotas) {
        int rc = orc_orb_quotas(nlevels, sf, target, quotas_inout);
        if (rc) return rc;
    }
    int total = 0;
    std::vector<std::vector<tb_keypoint>> all(nlevels);
    for (int level = 0; level < nlevels; level++) {
        const int w = ws[level], h = hs[level];
        const int minBX = 16, minBY = 16, maxBX = w - 16, maxBY = h - 16;
        /* NMS survivors are never 8-adjacent, so at most one per 2x2 block */
        std::vector<tb_corner> cand((size_t)w * h / 4 + 64);
        int nc = orc_orb_candidates(levels[level], w, h, strides[level], init_th, min_th, cand.data(), (int)cand.size());
        if (nc < 0) return nc;
        std::vector<tb_corner> sel(nc + 1);
        int ns = orc_distribute_octtree(cand.data(), nc, exit_keys, nexit, minBX, maxBX, minBY, maxBY,
                                        quotas_inout[level], sel.data(), nc);
        if (ns < 0) return ns;
        const int scaledPatchSize = (int)(31 * sf[level]);
        for (int i = 0; i < ns; i++) {
            tb_keypoint kp;
            kp.x = (float)sel[i].x + (float)minBX;
            kp.y = (float)sel[i].y + (float)minBY;
            kp.size = (float)scaledPatchSize;
            kp.angle = -1.f;
            kp.response = (float)sel[i].score;
            kp.octave = level;
            kp.class_id = -1;
            all[level].push_back(kp);
        }
    }
    for (int level = 0; level < nlevels; level++)
        for (tb_keypoint& kp : all[level]) kp.angle = orc_ic_angle(levels[level], strides[level], kp.x, kp.y);

    for (int level = 0; level < nlevels; level++) total += (int)all[level].size();
    if (total > cap) return TB_ECAPACITY;

    int offset = 0;
    for (int level = 0; level < nlevels; level++) {
        std::vector<tb_keypoint>& v = all[level];
        if (v.empty()) continue;
        const int w = ws[level], h = hs[level];
        std::vector<uint8_t> blur((size_t)w * h);
        orc_gaussian7(levels[level], w, h, strides[level], blur.data(), w);
        for (size_t i = 0; i < v.size(); i++)
            orc_orb_descriptor(blur.data(), w, v[i].x, v[i].y, v[i].angle, desc + (size_t)(offset + i) * 32);
        if (level != 0) {
            float scale = sf[level];
            for (tb_keypoint& kp : v) { kp.x *= scale; kp.y *= scale; }
        }
        for (size_t i = 0; i < v.size(); i++) kps[offset + i] = v[i];
        offset += (int)v.size();
    }
    return total;
}

/* FASTExtractor::shiTomasiScore, FASTextractor.cpp:87-127 */
float orc_shi_tomasi(const uint8_t* img, int w, int h, int stride, int u, int v) {
    float dXX = 0.0, dYY = 0.0, dXY = 0.0;
    const int halfbox_size = 4, box_size = 8, box_area = 64;
    const int x_min = u - halfbox_size, x_max = u + halfbox_size;
    const int y_min = v - halfbox_size, y_max = v + halfbox_size;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.0;
    for (int y = y_min; y < y_max; ++y) {
        const uint8_t* ptr_left = img + (std::ptrdiff_t)stride * y + x_min - 1;
        const uint8_t* ptr_right = img + (std::ptrdiff_t)stride * y + x_min + 1;
        const uint8_t* ptr_top = img + (std::ptrdiff_t)stride * (y - 1) + x_min;
        const uint8_t* ptr_bottom = img + (std::ptrdiff_t)stride * (y + 1) + x_min;
        for (int x = 0; x < box_size; ++x, ++ptr_left, ++ptr_right, ++ptr_top, ++ptr_bottom) {
            float dx = (float)*ptr_right - (float)*ptr_left;
            float dy = (float)*ptr_bottom - (float)*ptr_top;
            dXX += dx * dx;
            dYY += dy * dy;
            dXY += dx * dy;
        }
    }
    dXX = dXX / (2.f * box_area);
    dYY = dYY / (2.f * box_area);
    dXY = dXY / (2.f * box_area);
    return 0.5f * (dXX + dYY - std::sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

/* FASTExtractor::operator(), FASTextractor.cpp:8-80.
 * Deviation: the per-cell slot array covers every reachable index k instead of holding
 * `target` entries, where the reference's `.at(k)` would throw (e.g. 640x480, N=1000 -> 1036 cells);
 * occupancy beyond the caller's array reads as free. Detector threshold 20 is hard-coded as in the reference. */
int orc_fastgrid_extract(const uint8_t* const* levels, const int* ws, const int* hs, const int* strides,
                         int nlevels, const float* inv_sf, int target, float threshold,
                         const uint8_t* occupancy, int nocc, tb_keypoint* kps, int cap) {
    if (!levels || !ws || !hs || !strides || !inv_sf || nlevels < 1 || target < 1) return TB_EINVAL;
    if (!levels[0] || ws[0] < 1 || hs[0] < 1) return 0;
    const int cell_size = (int)std::sqrt((float)ws[0] * (float)hs[0] / (float)target);
    if (cell_size < 1) return TB_EINVAL;
    const int grid_n_cols = (int)((float)ws[0] / (float)cell_size);
    const int grid_n_rows = (int)((float)hs[0] / (float)cell_size);
    /* every reachable k = row*cols + col with row <= rows, col <= cols */
    const int ncell = std::max((grid_n_rows + 2) * (grid_n_cols + 1), target);
    std::vector<tb_keypoint> grid(ncell);
    for (tb_keypoint& g : grid) { g.x = g.y = 0; g.size = 0; g.angle = -1; g.response = 0; g.octave = 0; g.class_id = -1; }
    for (int L = 0; L < nlevels; L++) {
        std::vector<tb_corner> corners((size_t)ws[L] * hs[L] / 4 + 64);
        int nc = orc_fast10_nms(levels[L], ws[L], hs[L], strides[L], 20, corners.data(), (int)corners.size());
        if (nc < 0) return nc;
        corners.resize(nc);
        for (const tb_corner& c : corners) {
            const int k = static_cast<int>(((float)c.y * inv_sf[L]) / (float)cell_size) * grid_n_cols +
                          static_cast<int>(((float)c.x * inv_sf[L]) / (float)cell_size);
            if (k < 0 || k >= ncell) continue; /* outside the slot array: reference reads/writes out of range */
            if (occupancy && k < nocc && occupancy[k]) continue;
            const float score = orc_shi_tomasi(levels[L], ws[L], hs[L], strides[L], c.x, c.y);
            if (score > grid[k].response) {
                grid[k].x = (float)c.x * inv_sf[L];
                grid[k].y = (float)c.y * inv_sf[L];
                grid[k].size = 1;
                grid[k].angle = 0;
                grid[k].response = score;
                grid[k].octave = L;
                grid[k].class_id = -1;
            }
        }
    }
    int n = 0;
    for (const tb_keypoint& g : grid)
        if (g.response > (float)threshold) {
            if (kps) {
                if (n >= cap) return TB_ECAPACITY;
                kps[n] = g;
            }
            n++;
        }
    return n;
}

}  // extern "C"
