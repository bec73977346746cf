/* TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle (see orc_math.h
 * for the usage rule and the parity status).  Every function restates one
 * stage of the reference hot path and cites the reference file:line it
 * follows in oracle.cpp / oracle_match.cpp / oracle_pose.cpp.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/tb_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Frame::Frame scale vectors, Frame.cpp:18-29. Arrays of n floats. */
int orc_scale_factors(int n, float scale, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2);
/* Frame::ComputePyramid sizes, Frame.cpp:423-424. */
int orc_pyramid_sizes(int w, int h, int n, const float* sf, int* ws, int* hs);
/* cv::resize 8U INTER_LINEAR (OpenCV 3.3 fixed point), SURVEY App. A.1. */
int orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                         uint8_t* dst, int dw, int dh, int dstride);
/* cv::FAST(img, kps, th, nms) TYPE_9_16, SURVEY App. A.2. Returns count (or <0). */
int orc_fast9(const uint8_t* img, int w, int h, int stride, int th, int nms,
              tb_corner* out, int cap);
/* fast::fast_corner_detect_10 + fast_corner_score_10 + fast_nonmax_3x3 (FASTextractor.cpp:36-51). */
int orc_fast10_nms(const uint8_t* img, int w, int h, int stride, int th,
                   tb_corner* out, int cap);
/* FAST score map S(p) = max t such that p is a FAST-{arc} corner at threshold t (or -1). */
int orc_fast_score_map(const uint8_t* img, int w, int h, int stride, int arc, int16_t* out);
/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) 8U, SURVEY App. A.5. */
int orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* IC_Angle, ORBextractor.cpp:17-44. */
float orc_ic_angle(const uint8_t* img, int stride, float x, float y);
/* computeOrbDescriptor, ORBextractor.cpp:48-87 (img = blurred level). */
void orc_orb_descriptor(const uint8_t* img, int stride, float x, float y, float angle_deg, uint8_t* desc32);
/* per-level quota, ORBextractor.cpp:919-930. */
int orc_orb_quotas(int nlevels, const float* sf, int target, int* quotas);
/* cell-grid FAST candidates of one level, ORBextractor.cpp:747-804 (coords relative to the 16-px border). */
int orc_orb_candidates(const uint8_t* img, int w, int h, int stride, float init_th, float min_th,
                       tb_corner* out, int cap);
/* DistributeOctTree, ORBextractor.cpp:494-733. cand coords relative to the border; returns count. */
int orc_distribute_octtree(const tb_corner* cand, int ncand, const tb_keypoint* exit_keys, int nexit,
                           int min_x, int max_x, int min_y, int max_y, int quota,
                           tb_corner* out, int cap);
/* ORBExtractor::operator() (quotas_inout computed and written) / AddPoints (use_quotas=1: read). */
int orc_orb_extract(const uint8_t* const* levels, const int* ws, const int* hs, const int* strides,
                    int nlevels, const float* sf, int target, float init_th, float min_th,
                    const tb_keypoint* exit_keys, int nexit, int use_quotas, int* quotas_inout,
                    tb_keypoint* kps, uint8_t* desc, int cap);
/* FASTExtractor::operator(), FASTextractor.cpp:8-80. occupancy may be NULL. */
int orc_fastgrid_extract(const uint8_t* const* levels, const int* ws, const int* hs, const int* strides,
                         int nlevels, const float* inv_sf, int target, float threshold,
                         const uint8_t* occupancy, int nocc, tb_keypoint* kps, int cap);
float orc_shi_tomasi(const uint8_t* img, int w, int h, int stride, int u, int v);

/* Matcher::DescriptorDistance, matcher.cpp:793-808. */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
/* Matcher::ComputeThreeMaxima, matcher.cpp:810-851 (histogram given as bin sizes). */
void orc_three_maxima(const int* sizes, int L, int* i1, int* i2, int* i3);
/* cv::BFMatcher(NORM_HAMMING, crossCheck).match, OpenCV 3.3 batchDistance semantics [memory]. */
int orc_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int crosscheck,
                 tb_match* out, int cap);
/* Matcher::searchByBF whole-set branch, matcher.cpp:168-228. */
int orc_search_by_bf(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float ratio, float min_th,
                     tb_match* out, int cap);
/* Matcher::searchByViolence, matcher.cpp:299-395 with Frame grid Frame.cpp:187-265.
 * img2_w/img2_h = level-0 size of F2 (grid factors, Frame.cpp:30-31). */
int orc_search_by_violence(const tb_keypoint* k1, const uint8_t* d1, int n1,
                           const tb_keypoint* k2, const uint8_t* d2, int n2,
                           int img2_w, int img2_h, int min_level, int max_level, float radius,
                           int th_low, float nratio, int histo_len, int check_orientation,
                           tb_match* out, int cap);

/* SURVEY 8(f) row 4 -- Matcher::searchByBow, matcher.cpp:619-721; the frames' DBoW2 feature vectors are inputs (node ids
 * ascending, CSR feature lists); has_mp2 (nullable): F2->GetMapPoint(i) != nullptr, read when map_point_only */
int orc_search_by_bow(const tb_keypoint* k1, const uint8_t* d1, int n1, const uint32_t* nodes1, const int32_t* start1,
                      const uint32_t* items1, int nn1, const tb_keypoint* k2, const uint8_t* d2, int n2, const uint8_t* has_mp2,
                      const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2, int map_point_only,
                      int th_low, float nratio, int histo_len, int check_orientation, tb_match* out, int cap);

/* SURVEY 8(f) row 1 -- Matcher::searchByProjection(F1, F2), matcher.cpp:406-531 (PARITY UNPINNED, see the .cpp).
 * mp2 / mp2_desc are aligned with F2's keys (bad != 0: no usable map point); taken1[i] != 0: F1's key i already
 * has a map point with observations; img1_w/h = level-0 size of F1 (grid factors). */
int orc_search_by_projection(const float Tcw1[16], const tb_camera* cam1, int img1_w, int img1_h,
                             const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                             const tb_keypoint* k2, const tb_mappoint* mp2, const uint8_t* mp2_desc, int n2,
                             const float* scale_factors, int nlevels, float nratio, int th_high, int histo_len,
                             int check_orientation, tb_match* out, int cap);
/* Matcher::searchByProjection(map, F1, radio), matcher.cpp:539-617 + Frame::IsInFrustum, Frame.cpp:370-412. */
int orc_search_by_projection_map(const float Tcw1[16], const tb_camera* cam1, int img1_w, int img1_h,
                                 const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                                 const tb_mappoint* mps, const uint8_t* mp_desc, int nmp,
                                 const float* scale_factors, int nlevels, float nratio, float radio, int th_high,
                                 tb_match* out, int cap);

/* LocalBA::PoseOptimization, LocalBA.cpp:291-490 (g2o LM restated).
 * K = fx,fy,cx,cy. Tcw_in: row-major 4x4 float (vertex reset value). outlier: in/out flags.
 * Returns nInitialCorrespondences - nBad (>=0) or <0 on error. */
int orc_bow_transform(const tb_vocabulary* V, const uint8_t* desc, int n, int levelsup, int32_t* word_ids, double* weights,
                      int32_t* node_ids);
int orc_stereo_tracks_to_obs(const tb_keypoint* kl, const tb_keypoint* kr, const tb_match* matches, int nmatches, const float K[4],
                             float bf, const float* inv_sigma2, int nlevels, tb_obs* obs, int cap);
int orc_pose_opt(const double K[4], const float Tcw_in[16], const tb_obs* obs, int n,
                 uint8_t* outlier, float Tcw_out[16], double* stats /* nullable, 8 doubles */);

/* North-star extension with NO reference counterpart: multi-keyframe local BA,
 * LM + Schur complement in double. poses: nkf x 16 row-major Tcw (float in/out), pts: npt x 3. */
int orc_local_ba(const double K[4], int nkf, int nfixed, float* poses, int npt, float* pts,
                 const tb_ba_obs* obs, int nobs, int iters, double* stats /* nullable, 8 doubles */);

/* ---- oracle_flow.cpp: Matcher::searchByOPFlow (matcher.cpp:724-768) and the cv::calcOpticalFlowPyrLK call it makes
 * (OpenCV 3.3, not in the tree: restated, PARITY UNPINNED -- see the file header). */
int orc_pyr_down(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride);
/* returns the top pyramid level used (<= max_level), or < 0 on error */
int orc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, int stride, const float* prev_pts, int n,
                            int win, int max_level, float* next_pts, uint8_t* status, float* err /* nullable */);
/* Frame::Equalize (Frame.cpp:453-458) = cv::CLAHE(clip_limit, tiles).apply, restated; parity unpinned */
int orc_clahe(const uint8_t* src, int w, int h, int stride, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst, int dstride);
/* returns the number of matches, -1 on error, -2 when reject is asked for with 8..14 tracked points (OpenCV's LMedS branch) */
int orc_search_by_opflow(const uint8_t* img1, const uint8_t* img2, int w, int h, int stride, const tb_camera* cam1,
                         const float* keys2_xy, int n, int equalized, int reject, float* cur_points, int32_t* match_idx);

/* ---- oracle_fund.cpp: Matcher::rejectWithF (matcher.cpp:853-881) = cv::findFundamentalMat(FM_RANSAC, 1.0, 0.99), OpenCV 3.3
 * restated, PARITY UNPINNED (see the file header for what is OpenCV's structure and what is deliberately not), and
 * LocalBA::AddMapPointsByStereo (LocalBA.cpp:46-68). */
int orc_find_fundamental_ransac(const float* pts1, const float* pts2, int n, double thresh, double conf, uint8_t* mask,
                                double* F /* 9, nullable */, int* iters /* nullable */);
int orc_reject_with_f(const float* cur_pts, const float* last_pts, int n, uint8_t* status);
/* stereo = F1 (tracked into, equalised), current = F2 (its keys are tracked): depth[i] = bf / |x_tracked - x_key| for the
 * surviving matches, -1 elsewhere. Returns the number of depths set, -1 on error, -2 as above. */
int orc_add_map_points_by_stereo(const uint8_t* img_stereo, const uint8_t* img_current, int w, int h, int stride,
                                 const tb_camera* cam_stereo, const float* keys_xy, int n, float bf, float* depth);

#ifdef __cplusplus
}
#endif
#endif
