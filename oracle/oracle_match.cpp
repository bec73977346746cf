/* TEST INFRASTRUCTURE ONLY -- CPU oracle, matcher stages (see orc_math.h for the usage rule).
 * Restates src/matchers/matcher.cpp:168-228, :299-395, :793-851 and the Frame lookup grid of
 * src/types/Frame.cpp:30-31,187-265. cv::BFMatcher semantics are OpenCV 3.3's batchDistance
 * restated from memory (library absent): PARITY UNPINNED against genuine OpenCV.
 */
#include "oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

extern "C" {

/* Matcher::DescriptorDistance, matcher.cpp:793-808 (8 x 32-bit SWAR popcount). */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        int32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (int)(((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101u) >> 24) & 0xff);
    }
    return dist;
}

/* Matcher::ComputeThreeMaxima, matcher.cpp:810-851. i1..i3 must be initialised by the caller
 * (the reference passes -1,-1,-1 at :379). */
void orc_three_maxima(const int* sizes, int L, int* ind1, int* ind2, int* ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

/* cv::BFMatcher(NORM_HAMMING, crossCheck)::match(query=d1, train=d2), OpenCV 3.3 [memory]:
 *  - without cross-check: for each query the nearest train (first index on ties);
 *  - with cross-check (batchDistance(..., crosscheck=true)): for each TRAIN row i take its nearest
 *    query idx (first on ties); query idx then keeps the train with the smallest such distance
 *    (first i on ties); queries nobody points at produce no match.
 *  Output ordered by queryIdx; distance is the integer Hamming distance as float. */
int orc_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int crosscheck,
                 tb_match* out, int cap) {
    if (n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2)) return TB_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    std::vector<int> dist(n1, INT_MAX), nidx(n1, -1);
    if (!crosscheck) {
        for (int q = 0; q < n1; q++)
            for (int t = 0; t < n2; t++) {
                int d = orc_descriptor_distance(d1 + 32 * (size_t)q, d2 + 32 * (size_t)t);
                if (d < dist[q]) { dist[q] = d; nidx[q] = t; }
            }
    } else {
        for (int t = 0; t < n2; t++) {
            int best = INT_MAX, bq = -1;
            for (int q = 0; q < n1; q++) {
                int d = orc_descriptor_distance(d2 + 32 * (size_t)t, d1 + 32 * (size_t)q);
                if (d < best) { best = d; bq = q; }
            }
            if (best < dist[bq]) { dist[bq] = best; nidx[bq] = t; }
        }
    }
    int n = 0;
    for (int q = 0; q < n1; q++) {
        if (nidx[q] < 0) continue;
        if (out) {
            if (n >= cap) return TB_ECAPACITY;
            out[n].queryIdx = q;
            out[n].trainIdx = nidx[q];
            out[n].imgIdx = 0;
            out[n].distance = (float)dist[q];
        }
        n++;
    }
    return n;
}

/* Matcher::searchByBF whole-set branch, matcher.cpp:178-182,205-218: cross-checked BF match, then
 * keep m.distance < fmin(ratio * d_min, minTh). (An empty match list dereferences end() in the
 * reference; here it returns 0 matches.) */
int orc_search_by_bf(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float ratio, float min_th,
                     tb_match* out, int cap) {
    std::vector<tb_match> m((size_t)std::max(n1, 1));
    int n = orc_bf_match(d1, n1, d2, n2, 1, m.data(), (int)m.size());
    if (n <= 0) return n;
    float min_distance = m[0].distance;
    for (int i = 1; i < n; i++)
        if (m[i].distance < min_distance) min_distance = m[i].distance;
    const float lim = std::fmin(ratio * min_distance, min_th);
    int k = 0;
    for (int i = 0; i < n; i++)
        if (m[i].distance < lim) {
            if (out) {
                if (k >= cap) return TB_ECAPACITY;
                out[k] = m[i];
            }
            k++;
        }
    return k;
}

/* Matcher::searchByViolence, matcher.cpp:299-395, over Frame::AssignFeaturesToGrid /
 * GetFeaturesInArea / PosInGrid (Frame.cpp:187-265). The grid's inverse factors are swapped in the
 * reference (Frame.cpp:30-31: "HeightInv" = 120/cols, "WidthInv" = 36/rows); reproduced as is. */
int orc_search_by_violence(const tb_keypoint* k1, const uint8_t* d1, int n1,
                           const tb_keypoint* k2, const uint8_t* d2, int n2,
                           int img2_w, int img2_h, int min_level, int max_level, float r,
                           int th_low, float nratio, int histo_len, int check_orientation,
                           tb_match* out, int cap) {
    const int GRID_ROWS = 36, GRID_COLS = 120;
    if (n1 < 0 || n2 < 0 || histo_len < 1) return TB_EINVAL;
    const float heightInv = (float)GRID_COLS / (float)img2_w;
    const float widthInv = (float)GRID_ROWS / (float)img2_h;
    std::vector<std::vector<int>> grid((size_t)GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n2; i++) {
        int posX = (int)std::round(k2[i].x * widthInv);
        int posY = (int)std::round(k2[i].y * heightInv);
        if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
        grid[(size_t)posX * GRID_ROWS + posY].push_back(i);
    }
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist(histo_len);
    const float factor = 1.f / (float)histo_len;
    std::vector<int> cand;
    for (int i1 = 0; i1 < n1; i1++) {
        const float x = k1[i1].x, y = k1[i1].y;
        cand.clear();
        do {
            const int nMinCellX = std::max(0, (int)std::floor((x - r) * widthInv));
            if (nMinCellX >= GRID_COLS) break;
            const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x + r) * widthInv));
            if (nMaxCellX < 0) break;
            const int nMinCellY = std::max(0, (int)std::floor((y - r) * heightInv));
            if (nMinCellY >= GRID_ROWS) break;
            const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y + r) * heightInv));
            if (nMaxCellY < 0) break;
            const bool bCheckLevels = (min_level > 0) || (max_level >= 0);
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                    for (int j : grid[(size_t)ix * GRID_ROWS + iy]) {
                        if (bCheckLevels) {
                            if (k2[j].octave < min_level) continue;
                            if (max_level >= 0 && k2[j].octave > max_level) continue;
                        }
                        const float distx = k2[j].x - x, disty = k2[j].y - y;
                        if (std::fabs(distx) < r && std::fabs(disty) < r) cand.push_back(j);
                    }
        } while (0);
        if (cand.empty()) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : cand) {
            int dist = orc_descriptor_distance(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= th_low && (float)bestDist < (float)bestDist2 * nratio) {
            tb_match m{i1, bestIdx2, -1, (float)bestDist};
            matches.push_back(m);
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0) rot += 360.f;
                int bin = (int)std::roundf(rot * factor);
                if (bin == histo_len) bin = 0;
                if (bin < 0 || bin >= histo_len) return TB_EUNSUPPORTED; /* reference asserts */
                rotHist[bin].push_back((int)matches.size() - 1);
            }
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes(histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        orc_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good = matches;
    }
    if (out) {
        if ((int)good.size() > cap) return TB_ECAPACITY;
        std::copy(good.begin(), good.end(), out);
    }
    return (int)good.size();
}

}  // extern "C"

/* ------------------------------------------------------------------------------------------------
 * Matcher::searchByProjection, both overloads (matcher.cpp:406-617), with
 * Frame::GetFeaturesInArea (Frame.cpp:202-255), Frame::IsInFrustum (Frame.cpp:370-412),
 * PinholeCamera::World2Cam (CameraModel.cpp:63-93) and CameraModel::IsInFrame (CameraModel.h:33-39).
 * PARITY UNPINNED: the reference holds no vectors for these functions and its float results depend on the Eigen
 * version and compiler flags of its build. Restated here with Eigen 3.3's fixed-size reduction order
 * (c0 + (c1 + c2), redux_novec_unroller) and no FMA contraction; the HIP path is checked against this. */
namespace {
struct Grid {
    static const int ROWS = 36, COLS = 120;
    float heightInv, widthInv; /* swapped in the reference (Frame.cpp:30-31); kept */
    std::vector<std::vector<int>> cells;
    Grid(const tb_keypoint* k, int n, int img_w, int img_h) : cells((size_t)COLS * ROWS) {
        heightInv = (float)COLS / (float)img_w;
        widthInv = (float)ROWS / (float)img_h;
        for (int i = 0; i < n; i++) { /* Frame::AssignFeaturesToGrid + PosInGrid, Frame.cpp:187-200,257-265 */
            const int posX = (int)std::round(k[i].x * widthInv), posY = (int)std::round(k[i].y * heightInv);
            if (posX < 0 || posX >= COLS || posY < 0 || posY >= ROWS) continue;
            cells[(size_t)posX * ROWS + posY].push_back(i);
        }
    }
    /* Frame::GetFeaturesInArea, Frame.cpp:202-255 */
    void area(const tb_keypoint* k, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const {
        out.clear();
        const int nMinCellX = std::max(0, (int)std::floor((x - r) * widthInv));
        if (nMinCellX >= COLS) return;
        const int nMaxCellX = std::min(COLS - 1, (int)std::ceil((x + r) * widthInv));
        if (nMaxCellX < 0) return;
        const int nMinCellY = std::max(0, (int)std::floor((y - r) * heightInv));
        if (nMinCellY >= ROWS) return;
        const int nMaxCellY = std::min(ROWS - 1, (int)std::ceil((y + r) * heightInv));
        if (nMaxCellY < 0) return;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int j : cells[(size_t)ix * ROWS + iy]) {
                    if (bCheckLevels) {
                        if (k[j].octave < minLevel) continue;
                        if (maxLevel >= 0 && k[j].octave > maxLevel) continue;
                    }
                    const float distx = k[j].x - x, disty = k[j].y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(j);
                }
    }
};

/* Rcw * X + tcw, row-major 4x4 T; each coefficient c0 + (c1 + c2), then + t */
inline void se3_map(const float* T, const float* X, float* Pc) {
    for (int i = 0; i < 3; i++) {
        const float c0 = T[4 * i] * X[0], c1 = T[4 * i + 1] * X[1], c2 = T[4 * i + 2] * X[2];
        Pc[i] = (c0 + (c1 + c2)) + T[4 * i + 3];
    }
}
/* PinholeCamera::World2Cam(xyz_c), CameraModel.cpp:63-93 */
inline void world2cam(const tb_camera* cam, const float* Pc, float* px) {
    const float x = Pc[0] / Pc[2], y = Pc[1] / Pc[2];
    if (!cam->has_distortion) {
        px[0] = cam->fx * x + cam->cx;
        px[1] = cam->fy * y + cam->cy;
    } else {
        const float* md = cam->d;
        const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const float a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const float cdist = 1 + md[0] * r2 + md[1] * r4 + md[4] * r6;
        const float xd = x * cdist + md[2] * a1 + md[3] * a2;
        const float yd = y * cdist + md[2] * a3 + md[3] * a1;
        px[0] = xd * cam->fx + cam->cx;
        px[1] = yd * cam->fy + cam->cy;
    }
}
/* IsInFrame(uv.cast<int>()), CameraModel.h:33-39: the cast truncates; a non-finite or out-of-int-range
 * coordinate converts to INT_MIN on x86 (cvttss2si) and fails the test */
inline bool in_frame(const tb_camera* cam, const float* px) {
    if (!(std::fabs(px[0]) < 2147483648.f) || !(std::fabs(px[1]) < 2147483648.f)) return false;
    const int u = (int)px[0], v = (int)px[1];
    return u >= 0 && u < (int)((float)cam->width * 1.f) && v >= 0 && v < (int)((float)cam->height * 1.f);
}
}  // namespace

extern "C" {

/* Matcher::searchByProjection(F1, F2), matcher.cpp:406-531. mp2 / mp2_desc are aligned with F2's keys. */
int orc_search_by_projection(const float Tcw1[16], const tb_camera* cam1, int img1_w, int img1_h,
                             const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                             const tb_keypoint* k2, const tb_mappoint* mp2, const uint8_t* mp2_desc, int n2,
                             const float* scale_factors, int nlevels, float nratio, int th_high, int histo_len,
                             int check_orientation, tb_match* out, int cap) {
    if (n1 < 0 || n2 < 0 || histo_len < 1 || nlevels < 1) return TB_EINVAL;
    const Grid grid(k1, n1, img1_w, img1_h);
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist(histo_len);
    const float factor = 1.0f / (float)histo_len;
    std::vector<int> cand;
    for (int i2 = 0; i2 < n2; i2++) {
        if (mp2[i2].bad) continue;
        float Pc[3], uv[2];
        se3_map(Tcw1, mp2[i2].pos, Pc);
        const float invzc = 1.0f / Pc[2];
        if (invzc < 0) continue;
        world2cam(cam1, Pc, uv);
        if (!in_frame(cam1, uv)) continue;
        const int nLastOctave = k2[i2].octave;
        if (nLastOctave < 0 || nLastOctave >= nlevels) return TB_EINVAL; /* the reference indexes out of range */
        const float radius = nratio * scale_factors[nLastOctave];
        grid.area(k1, uv[0], uv[1], radius, nLastOctave - 1, nLastOctave + 1, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx1 = -1;
        for (int i1 : cand) {
            if (taken1 && taken1[i1]) continue; /* F1->GetMapPoint(i1) with Observations() > 0 */
            const int dist = orc_descriptor_distance(mp2_desc + 32 * (size_t)i2, d1 + 32 * (size_t)i1);
            if (dist < bestDist) { bestDist = dist; bestIdx1 = i1; }
        }
        if (bestDist <= th_high && bestIdx1 >= 0) {
            tb_match m{bestIdx1, i2, -1, (float)bestDist};
            matches.push_back(m);
            if (check_orientation) {
                float rot = k2[i2].angle - k1[bestIdx1].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::roundf(rot * factor);
                if (bin == histo_len) bin = 0;
                if (bin < 0 || bin >= histo_len) return TB_EUNSUPPORTED; /* reference asserts */
                rotHist[bin].push_back((int)matches.size() - 1);
            }
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes(histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        orc_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good = matches;
    }
    if (out) {
        if ((int)good.size() > cap) return TB_ECAPACITY;
        std::copy(good.begin(), good.end(), out);
    }
    return (int)good.size();
}

/* Matcher::searchByProjection(map, F1, radio), matcher.cpp:539-617, over Frame::IsInFrustum (Frame.cpp:370-412;
 * viewingCosLimit enters as 0.5, the predicted level is the constant 0 of the reference's TODO). */
int orc_search_by_projection_map(const float Tcw1[16], const tb_camera* cam1, int img1_w, int img1_h,
                                 const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                                 const tb_mappoint* mps, const uint8_t* mp_desc, int nmp,
                                 const float* scale_factors, int nlevels, float nratio, float radio, int th_high,
                                 tb_match* out, int cap) {
    if (n1 < 0 || nmp < 0 || nlevels < 1) return TB_EINVAL;
    const Grid grid(k1, n1, img1_w, img1_h);
    /* Frame::SetPose, Frame.cpp:50-62: mOw = -Rcw^T * tcw */
    float Ow[3];
    for (int i = 0; i < 3; i++) {
        const float c0 = -Tcw1[i] * Tcw1[3], c1 = -Tcw1[4 + i] * Tcw1[7], c2 = -Tcw1[8 + i] * Tcw1[11];
        Ow[i] = c0 + (c1 + c2);
    }
    const bool bFactor = nratio != 1.0;
    std::vector<tb_match> matches;
    std::vector<int> cand;
    for (int im = 0; im < nmp; im++) {
        const tb_mappoint& mp = mps[im];
        if (mp.bad) continue;
        /* IsInFrustum */
        float Pc[3], uv[2];
        se3_map(Tcw1, mp.pos, Pc);
        if (Pc[2] < 0.0f) continue;
        world2cam(cam1, Pc, uv);
        if (!in_frame(cam1, uv)) continue;
        const float PO[3] = {mp.pos[0] - Ow[0], mp.pos[1] - Ow[1], mp.pos[2] - Ow[2]};
        const float dist3 = std::sqrt(PO[0] * PO[0] + (PO[1] * PO[1] + PO[2] * PO[2]));
        if (dist3 < mp.min_dist || dist3 > mp.max_dist) continue;
        const float viewCos = (PO[0] * mp.normal[0] + (PO[1] * mp.normal[1] + PO[2] * mp.normal[2])) / dist3;
        if (viewCos < 0.5f) continue;
        const int nPredictedLevel = 0;
        float r = 4.f;
        if (viewCos > 0.998) r = 2.5;
        if (bFactor) r *= nratio;
        grid.area(k1, uv[0], uv[1], r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
            if (taken1 && taken1[idx]) continue;
            const int dist = orc_descriptor_distance(mp_desc + 32 * (size_t)im, d1 + 32 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k1[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = k1[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= th_high && bestIdx >= 0) {
            if (bestLevel == bestLevel2 && (float)bestDist > radio * (float)bestDist2) continue;
            tb_match m{bestIdx, im, -1, (float)bestDist};
            matches.push_back(m);
        }
    }
    if (out) {
        if ((int)matches.size() > cap) return TB_ECAPACITY;
        std::copy(matches.begin(), matches.end(), out);
    }
    return (int)matches.size();
}

/* SURVEY 8(f) row 4 -- Matcher::searchByBow(F1, F2, MapPointOnly), matcher.cpp:619-721. The frames' DBoW2 feature
 * vectors (Frame::GetFeatureVector(): std::map<NodeId, std::vector<unsigned>>, filled by voc->transform(.., 4),
 * Frame.cpp:269) are INPUTS here: node ids ascending (the map's order), the nodes' feature lists as CSR (start[nn + 1],
 * items in insertion order). DBoW2 and its vocabulary are not part of this path (the tree holds no vocabulary file).
 * Walk of the two sorted node lists as in :637-698 (the lower_bound jumps = skipping smaller ids); per feature of F1 in a
 * shared node: best / second-best Hamming distance over F2's features of that node (:645-669), accepted if
 * best < TH_LOW and best < nRatio * second (:671-673); rotation histogram with factor 1 / HISTO_LENGTH and
 * round(rot * factor) (:677-686, as written there), the three fullest bins kept, in bin order (:701-717). */
int orc_search_by_bow(const tb_keypoint* k1, const uint8_t* d1, int n1, const uint32_t* nodes1, const int32_t* start1,
                      const uint32_t* items1, int nn1, const tb_keypoint* k2, const uint8_t* d2, int n2, const uint8_t* has_mp2,
                      const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2, int map_point_only,
                      int th_low, float nratio, int histo_len, int check_orientation, tb_match* out, int cap) {
    if (n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || histo_len < 1) return TB_EINVAL;
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist((size_t)histo_len);
    const float factor = 1.0f / (float)histo_len;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            for (int p1 = start1[a]; p1 < start1[a + 1]; p1++) {
                const int idx1 = (int)items1[p1];
                if (idx1 < 0 || idx1 >= n1) return TB_EINVAL;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int p2 = start2[b]; p2 < start2[b + 1]; p2++) {
                    const int idx2 = (int)items2[p2];
                    if (idx2 < 0 || idx2 >= n2) return TB_EINVAL;
                    if (map_point_only && !(has_mp2 && has_mp2[idx2])) continue;
                    const int dist = orc_descriptor_distance(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < th_low && bestIdx2 >= 0) {   /* bestIdx2 < 0 (TH_LOW > 256): the reference indexes key -1 */
                    if ((float)bestDist1 < nratio * (float)bestDist2) {
                        tb_match m = {idx1, bestIdx2, -1, (float)bestDist1};
                        matches.push_back(m);
                        if (check_orientation) {
                            float rot = k1[idx1].angle - k2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == histo_len) bin = 0;
                            if (bin < 0 || bin >= histo_len) return TB_EUNSUPPORTED;   /* the reference asserts */
                            rotHist[bin].push_back((int)matches.size() - 1);
                        }
                    }
                }
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;      /* lower_bound(f2it->first) */
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes((size_t)histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        orc_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good.swap(matches);
    }
    if ((int)good.size() > cap) return TB_ECAPACITY;
    for (size_t i = 0; i < good.size(); i++) out[i] = good[i];
    return (int)good.size();
}


/* Stereo tracks -> PoseOptimization's inputs (the composition bench.py times; see k_stereo_obs in k_match.hip): per left <->
 * right match Depth = bf / |x_right - x_left| (LocalBA.cpp:60-64), X = ((x - cx) / fx, (y - cy) / fy, 1) * Depth of the LEFT key
 * (test/test_vo.cpp:257-267), observed at the RIGHT key's pixel with invSigma2[octave] (LocalBA.cpp:333-363). Rows in match
 * order; matches without disparity or with an octave outside the table are dropped. Returns the number of rows. */
int orc_stereo_tracks_to_obs(const tb_keypoint* kl, const tb_keypoint* kr, const tb_match* matches, int nmatches, const float K[4],
                             float bf, const float* inv_sigma2, int nlevels, tb_obs* obs, int cap) {
    int n = 0;
    for (int i = 0; i < nmatches; i++) {
        const tb_keypoint a = kl[matches[i].queryIdx], b = kr[matches[i].trainIdx];
        const float depth = bf / fabsf(b.x - a.x);
        const float nx = (a.x - K[2]) / K[0], ny = (a.y - K[3]) / K[1];
        if (!std::isfinite(depth) || b.octave < 0 || b.octave >= nlevels) continue;
        if (n >= cap) break;
        tb_obs o;
        o.u = b.x; o.v = b.y;
        o.X = nx * depth; o.Y = ny * depth; o.Z = depth;
        o.inv_sigma2 = inv_sigma2[b.octave];
        obs[n++] = o;
    }
    return n;
}


/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) for every descriptor
 * (third_part/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1260), distances by FORB::distance (FORB.cpp:81-101, a SWAR popcount
 * of the XOR). The tree comes as flat arrays (tb_vocabulary). Where a branch ends above level L - levelsup the reference
 * leaves *nid unset (its caller passes an uninitialised NodeId): defined here as the leaf. */
int orc_bow_transform(const tb_vocabulary* V, const uint8_t* desc, int n, int levelsup, int32_t* word_ids, double* weights,
                      int32_t* node_ids) {
    if (!V || V->nnodes < 1) return -1;
    const int nid_level = V->L - levelsup;
    for (int f = 0; f < n; f++) {
        const uint8_t* a = desc + 32 * (size_t)f;
        int final_id = 0, level = 0, nid = 0;
        bool nid_set = nid_level <= 0;
        while (V->child_start[final_id + 1] > V->child_start[final_id]) {
            level++;
            const int c0 = V->child_start[final_id], c1 = V->child_start[final_id + 1];
            int best = V->child_items[c0];
            int best_d = orc_descriptor_distance(a, V->desc + 32 * (size_t)best);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = V->child_items[c];
                const int d = orc_descriptor_distance(a, V->desc + 32 * (size_t)id);
                if (d < best_d) { best_d = d; best = id; }
            }
            final_id = best;
            if (level == nid_level) { nid = final_id; nid_set = true; }
        }
        if (!nid_set) nid = final_id;
        word_ids[f] = V->word_id[final_id];
        weights[f] = V->weight[final_id];
        node_ids[f] = nid;
    }
    return 0;
}

}  // extern "C"
