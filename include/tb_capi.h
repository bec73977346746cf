/* C ABI of the MI355X tracking hot path (libtb_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  The reference has no
 * FFI of its own (it is one C++ static library, CMakeLists.txt:30-45); the functions below are what a
 * binding for its hot-path operators would bind, one per operator, each citing the reference interface
 * it replaces.  include/extractors, include/matchers, include/mapping hold the C++ header shims that
 * keep the reference's class signatures on top of this ABI.
 *
 * Conventions: return 0 (TB_OK) or a negative TB_E* code (tb_types.h); tb_last_error(ctx) gives the text.
 * Outputs are caller-allocated with stated capacities.  A tb_ctx owns one GPU and one HIP stream and is
 * not thread-safe; contexts on different GPUs are independent.  "host" pointers are ordinary memory,
 * "dev" pointers are HIP device memory of the context's GPU.  There is NO CPU fallback: every compute
 * entry point fails with TB_EDEVICE when no gfx950 device is usable.
 */
#ifndef TB_CAPI_H
#define TB_CAPI_H

#include "tb_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tb_ctx tb_ctx;
typedef struct tb_extractor tb_extractor;

/* ---------------------------------------------------------------- context */
int tb_create(int device, tb_ctx** out);
void tb_destroy(tb_ctx* ctx);
const char* tb_last_error(const tb_ctx* ctx);
const char* tb_strerror(int code);
const char* tb_version(void);
/* Run everything on an existing HIP stream (e.g. torch's current stream); NULL = the context's own. */
int tb_set_stream(tb_ctx* ctx, void* hip_stream);
int tb_synchronize(tb_ctx* ctx);
/* Per-kernel timing with HIP events on the context's stream (measurement aid for bench.py: the roofline
 * figure needs the dominant kernel's average launch duration over the timed region). enable(1) resets the
 * accumulators; report() synchronises and writes one line per kernel: "name calls total_ms\n". */
int tb_profile_enable(tb_ctx* ctx, int on);
/* Time only the named kernel while profiling is enabled (NULL or "": all of them). Two event records per launch cost host
 * time and a queue packet each: with several hundred launches per step, timing every kernel inside a measured region costs
 * ~2 % of the region; timing the one kernel a roofline figure is about does not. */
int tb_profile_only(tb_ctx* ctx, const char* kernel);
/* Exchange helper ("counts first, then the live records"): the first counts[f] rows of every frame of a
 * [nframes][cap][row_bytes] record array, frame after frame, to the front of dst (same total size); *total (nullable, device,
 * int64) = the number of rows written. row_bytes a multiple of 4. Device pointers, asynchronous on the context's stream. */
int tb_pack_rows_dev(tb_ctx* ctx, const void* src, int row_bytes, int cap, const int32_t* counts, int nframes, void* dst,
                     long long* total);
/* A hint for launch shapes: `peers` contexts (this one included) are expected to run their kernels on this GPU at the same time
 * -- e.g. a batch's local-BA windows split over several contexts, each on its own stream and host thread. Kernels whose grid is
 * sized to fill the chip in one resident round (the local-BA Schur kernel) then take 1 / peers of it. Default 1. Results do not
 * depend on it beyond the order of floating-point partial sums (the number of partial systems per window follows the grid). */
int tb_set_concurrency(tb_ctx* ctx, int peers);
/* Measurement helper: copies `bytes` (a multiple of 16, 16-byte aligned device pointers) with a 16-byte-per-lane
 * kernel and returns the average seconds per copy over `reps` (HIP events on the context's stream, one untimed copy first). The
 * streaming bandwidth of the device is 2 * bytes / seconds. */
int tb_measure_copy_seconds(tb_ctx* ctx, const void* d_src, void* d_dst, size_t bytes, int reps, double* seconds);
/* Test hook: on != 0 sends every block of the cell-wise FAST kernel down its any-density path (no candidate lists; the
 * results are the same). A context setting, not an environment variable: nothing outside the caller changes which kernels run. */
int tb_debug_force_dense_fast(tb_ctx* ctx, int on);
int tb_profile_report(tb_ctx* ctx, char* buf, int cap);

/* ---------------------------------------------------------------- a1/a2/a3: host-side scalar set-up
 * Frame::Frame scale vectors (src/types/Frame.cpp:18-29), Frame::ComputePyramid sizes (:423-424),
 * ORBExtractor::operator() per-level quota (src/extractors/ORBextractor.cpp:919-930). Pure host math. */
int tb_scale_factors(int nlevels, float scale, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2);
int tb_pyramid_sizes(int width, int height, int nlevels, const float* sf, int* widths, int* heights);
int tb_orb_quotas(int nlevels, const float* sf, int target, int* quotas);

/* ---------------------------------------------------------------- batched extractor (device resident)
 * One plan = one image geometry (width x height, nlevels, scale vector) and up to max_images frames in
 * flight.  Replaces Frame::ComputePyramid (Frame.cpp:414-427) + ORBExtractor::operator()/AddPoints
 * (ORBextractor.cpp:906-978, :840-904) + FASTExtractor::operator() (FASTextractor.cpp:8-80) for a
 * whole batch of frames per call. max_target bounds `target` of later calls. */
int tb_extractor_create(tb_ctx* ctx, int width, int height, int nlevels, const float* sf,
                        const int* widths, const int* heights, /* per-level sizes; NULL = Frame.cpp:423-424 */
                        int max_images, int max_target, tb_extractor** out);
void tb_extractor_destroy(tb_extractor* ex);
/* Level-0 images: n frames, row stride `stride` bytes, frame pitch `pitch` bytes. The host form copies
 * (PCIe); the dev form only records the pointer (frames are read in place, they must stay valid). */
int tb_extractor_set_images_host(tb_extractor* ex, const uint8_t* images, int n, int stride, size_t pitch);
int tb_extractor_set_images_dev(tb_extractor* ex, const uint8_t* dev_images, int n, int stride, size_t pitch);
/* Caller-built pyramid for frame `index` (the reference's extractors take std::vector<cv::Mat>&). */
int tb_extractor_set_levels_host(tb_extractor* ex, int index, const uint8_t* const* levels, const int* strides);
/* a2: levels 1..n-1 of every frame by the cv::resize INTER_LINEAR chain. */
int tb_extractor_build_pyramid(tb_extractor* ex, int n);
int tb_extractor_get_level_host(tb_extractor* ex, int index, int level, uint8_t* out, int out_stride);
/* a3-a10: ORB extraction of frames [0,n). quota_mode 0 = operator() (quotas from target), 1 = AddPoints
 * (reuse the quotas of the last quota_mode-0 call; TB_ESTATE if none). exit keys (host, may be NULL)
 * apply to every frame of the call. Results stay on the device. */
int tb_extractor_orb(tb_extractor* ex, int n, int target, float init_th, float min_th, int quota_mode,
                     const tb_keypoint* exit_keys, int n_exit);
/* a11: FASTExtractor grid extraction (no descriptors). occupancy: host bytes, may be NULL. */
int tb_extractor_fastgrid(tb_extractor* ex, int n, const float* inv_sf, int target, float threshold,
                          const uint8_t* occupancy, int n_occupancy);
/* Results of the last extraction call. counts: n ints (keypoints per frame). */
int tb_extractor_counts_host(tb_extractor* ex, int n, int* counts);
int tb_extractor_results_host(tb_extractor* ex, int index, tb_keypoint* kps, uint8_t* desc, int cap, int* count);
/* Device views (valid until the next extraction call): keypoints [max_images][kp_capacity],
 * descriptors [max_images][kp_capacity][32], counts [max_images]. */
int tb_extractor_results_dev(tb_extractor* ex, const tb_keypoint** kps, const uint8_t** desc,
                             const int32_t** counts, int* kp_capacity);
/* Copy the results of frames [0,n) into caller-owned device buffers (e.g. torch tensors that feed the
 * RCCL gather): kps [n][cap], desc [n][cap][32], counts [n]; cap >= the plan's kp_capacity is not required,
 * rows beyond cap are dropped (counts are clamped). Asynchronous on the context's stream. */
int tb_extractor_copy_results_dev(tb_extractor* ex, int n, tb_keypoint* kps, uint8_t* desc, int32_t* counts, int cap);
/* Stage probes for parity tests: FAST candidates of one level after the cell loop (a4). */
int tb_extractor_candidates_host(tb_extractor* ex, int index, int level, tb_corner* out, int cap, int* count);

/* ---------------------------------------------------------------- single-frame operator forms (host buffers)
 * What the header shims call; each wraps a cached plan. */
/* Frame::ComputePyramid: levels[i] (i>=1) receive widths[i] x heights[i] bytes at strides[i]. */
int tb_pyramid(tb_ctx* ctx, const uint8_t* image, int width, int height, int stride, int nlevels,
               const float* sf, uint8_t* const* levels_out, const int* strides_out);
/* cv::FAST(img, kps, th, nms) TYPE_9_16 on a whole image (the primitive inside a4; raster order). */
int tb_fast_detect(tb_ctx* ctx, const uint8_t* image, int width, int height, int stride, int threshold,
                   int nms, tb_corner* out, int cap, int* count);
/* ORBExtractor::operator() (use_quotas 0) / AddPoints (use_quotas 1), ORBextractor.cpp:906-978,:840-904 */
int tb_orb_extract(tb_ctx* ctx, const uint8_t* const* levels, const int* widths, const int* heights,
                   const int* strides, int nlevels, const float* sf, int target, float init_th, float min_th,
                   const tb_keypoint* exit_keys, int n_exit, int use_quotas, int* quotas_inout,
                   tb_keypoint* kps, uint8_t* desc, int cap, int* count);
/* FASTExtractor::operator(), FASTextractor.cpp:8-80 */
int tb_fastgrid_extract(tb_ctx* ctx, const uint8_t* const* levels, const int* widths, const int* heights,
                        const int* strides, int nlevels, const float* inv_sf, int target, float threshold,
                        const uint8_t* occupancy, int n_occupancy, tb_keypoint* kps, int cap, int* count);

/* ---------------------------------------------------------------- matchers
 * Matcher::DescriptorDistance / ComputeThreeMaxima (matcher.cpp:793-851): host helpers. */
int tb_descriptor_distance(const uint8_t* a, const uint8_t* b);
void tb_three_maxima(const int* bin_sizes, int nbins, int* ind1, int* ind2, int* ind3);
/* cv::BFMatcher(NORM_HAMMING, crossCheck).match (the call inside searchByBF, matcher.cpp:207) */
int tb_match_bf(tb_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int crosscheck,
                tb_match* out, int cap, int* count);
/* Matcher::searchByBF whole-set branch, matcher.cpp:168-228 */
int tb_search_by_bf(tb_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, float ratio,
                    float min_th, tb_match* out, int cap, int* count);
/* Batched device form: npairs descriptor-set pairs, set p of side s at desc_s + p*set_pitch bytes with
 * counts_s[p] rows; matches to out + p*cap, counts to out_counts[p]. All pointers are device memory. */
int tb_search_by_bf_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* desc1, const int32_t* counts1,
                              const uint8_t* desc2, const int32_t* counts2, size_t set_pitch,
                              float ratio, float min_th, tb_match* out, int cap, int32_t* out_counts);
/* Matcher::searchByViolence, matcher.cpp:299-395 (+ Frame grid, Frame.cpp:187-265). */
int tb_search_by_violence(tb_ctx* ctx, const tb_keypoint* k1, const uint8_t* d1, int n1,
                          const tb_keypoint* k2, const uint8_t* d2, int n2, int img2_width, int img2_height,
                          int min_level, int max_level, float radius, int th_low, float nratio,
                          int histo_len, int check_orientation, tb_match* out, int cap, int* count);

/* SURVEY 8(f) row 4 -- Matcher::searchByBow(F1, F2, MapPointOnly), matcher.cpp:619-721. The frames' DBoW2 feature vectors
 * (Frame::GetFeatureVector(), a std::map<NodeId, std::vector<unsigned>> filled by voc->transform(.., 4), Frame.cpp:269)
 * are INPUTS: nodesX = node ids in ascending order (the map's order), startX[nnX + 1] / itemsX = the nodes' feature index
 * lists as CSR, in insertion order. DBoW2 and its vocabulary stay outside this library (the reference tree ships no
 * vocabulary file). has_mp2 (nullable, n2 bytes): F2->GetMapPoint(i) != nullptr, read when map_point_only is set.
 * th_low / nratio / histo_len / check_orientation = the Matcher's TH_LOW / nRatio / HISTO_LENGTH / checkOrientation.
 * Matches: queryIdx = F1 key, trainIdx = F2 key, imgIdx = -1, distance = Hamming, in the reference's order. */
int tb_search_by_bow(tb_ctx* ctx, const tb_keypoint* k1, const uint8_t* d1, int n1, const uint32_t* nodes1, const int32_t* start1,
                     const uint32_t* items1, int nn1, const tb_keypoint* k2, const uint8_t* d2, int n2, const uint8_t* has_mp2,
                     const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2, int map_point_only, int th_low,
                     float nratio, int histo_len, int check_orientation, tb_match* out, int cap, int* count);

/* SURVEY 8(f) row 1 -- Matcher::searchByProjection(F1, F2), matcher.cpp:406-531 (+ Frame::GetFeaturesInArea,
 * Frame.cpp:202-255; PinholeCamera::World2Cam, CameraModel.cpp:63-93; CameraModel::IsInFrame, CameraModel.h:33-39).
 * F1 = current frame: pose Tcw1 (row-major 4x4), camera, level-0 image size (lookup-grid factors), keys k1,
 * descriptors d1 (n1 x 32), taken1[i] != 0 iff F1->GetMapPoint(i) has Observations() > 0 (nullable = none).
 * F2 = reference frame: keys k2 (octave, angle) and, aligned with them, its map points mp2 (bad != 0 also for "no
 * map point") with descriptors mp2_desc (n2 x 32). scale_factors = F1->GetScaleFactors() (nlevels entries).
 * Matches: queryIdx = F1 key, trainIdx = i2, imgIdx = -1, distance = Hamming; order as the reference emits them. */
int tb_search_by_projection(tb_ctx* ctx, const float Tcw1[16], const tb_camera* cam1, int img1_width, int img1_height,
                            const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                            const tb_keypoint* k2, const tb_mappoint* mp2, const uint8_t* mp2_desc, int n2,
                            const float* scale_factors, int nlevels, float nratio, int th_high, int histo_len,
                            int check_orientation, tb_match* out, int cap, int* count);
/* Matcher::searchByProjection(map, F1, radio), matcher.cpp:539-617 (+ Frame::IsInFrustum, Frame.cpp:370-412, entered
 * with viewingCosLimit 0.5; the predicted level is the reference's constant 0). mps = map->GetAllMapPoints() in
 * order; trainIdx = index into mps. */
int tb_search_by_projection_map(tb_ctx* ctx, const float Tcw1[16], const tb_camera* cam1, int img1_width, int img1_height,
                                const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1,
                                const tb_mappoint* mps, const uint8_t* mp_desc, int nmp,
                                const float* scale_factors, int nlevels, float nratio, float radio, int th_high,
                                tb_match* out, int cap, int* count);

/* SURVEY 8(f) row 3 -- Frame::AssignFeaturesToGrid (Frame.cpp:187-200, PosInGrid :257-265) for a batch of frames, on
 * the device: frame f has counts[f] keys at keys + f*key_pitch; its 120x36 lookup grid comes back as CSR,
 * cell_start + f*4321 (cell c = posX*36 + posY, last entry = total) and cell_items + f*key_pitch (key indices, inside a
 * cell in insertion = index order). img_* = level-0 image size (grid factors, swapped as in Frame.cpp:30-31).
 * Device pointers, asynchronous on the context's stream. */
int tb_frame_grid_batch_dev(tb_ctx* ctx, int nframes, const tb_keypoint* keys, const int32_t* counts, int key_pitch,
                            int img_width, int img_height, int32_t* cell_start, int32_t* cell_items);
/* Batched, device-resident Matcher::searchByProjection(F1, F2) (matcher.cpp:406-531): pair p reads Tcw1 + 16p, F1's
 * keys / descriptors / taken flags at stride pitch1 (n1[p] valid) with the lookup grid built by
 * tb_frame_grid_batch_dev, F2's keys and key-aligned map points / descriptors at stride pitch2 (n2[p] valid); cam1 and
 * scale_factors are host arrays shared by all pairs. Matches go to out + p*cap in the reference's order, their
 * number to out_counts[p] (if it exceeds cap the list is truncated, the count is not); flags[p] != 0 reports what the
 * host form returns as an error (1: a key octave outside the scale factors, 2: a rotation bin outside the histogram,
 * where the reference asserts). Device pointers, asynchronous, no host synchronisation. histo_len <= 1024. */
int tb_search_by_projection_batch_dev(tb_ctx* ctx, int npairs, const float* Tcw1, const tb_camera* cam1, int img1_width,
                                      int img1_height, const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1,
                                      const int32_t* n1, int pitch1, const int32_t* cell_start, const int32_t* cell_items,
                                      const tb_keypoint* k2, const tb_mappoint* mp2, const uint8_t* mp2_desc,
                                      const int32_t* n2, int pitch2, const float* scale_factors, int nlevels, float nratio,
                                      int th_high, int histo_len, int check_orientation, tb_match* out, int cap,
                                      int32_t* out_counts, int32_t* flags);

/* Batched, device-resident Matcher::searchByProjection(map, F1, radio) (matcher.cpp:539-617): pair p = one current
 * frame (pose, keys, descriptors, taken flags, lookup grid as above) against a map of nmp[p] points at
 * mps + p*mp_pitch, mp_desc + p*mp_pitch*32; mp_pitch = 0 matches every frame against ONE shared map. max_nmp bounds
 * nmp[]. Matches (queryIdx = F1 key, trainIdx = map point index) to out + p*cap in map order, counts to
 * out_counts[p] (truncated list, untruncated count). Device pointers, asynchronous. */
int tb_search_by_projection_map_batch_dev(tb_ctx* ctx, int npairs, const float* Tcw1, const tb_camera* cam1, int img1_width,
                                          int img1_height, const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1,
                                          const int32_t* n1, int pitch1, const int32_t* cell_start, const int32_t* cell_items,
                                          const tb_mappoint* mps, const uint8_t* mp_desc, const int32_t* nmp, int mp_pitch,
                                          int max_nmp, const float* scale_factors, int nlevels, float nratio, float radio,
                                          int th_high, tb_match* out, int cap, int32_t* out_counts, int32_t* flags);
/* Batched, device-resident Matcher::searchByViolence (matcher.cpp:299-395): pair p matches F1's keys (k1 / d1 at stride
 * pitch1, n1[p] valid) against F2's (stride pitch2, n2[p] valid) through F2's lookup grid from
 * tb_frame_grid_batch_dev (built over k2 with img2_*). Matches (queryIdx = F1 key, trainIdx = F2 key) go to
 * out + p*cap in the reference's order, their number to out_counts[p] (truncated list, untruncated count);
 * flags[p] = 2 reports a rotation bin outside the histogram (the reference asserts). Device pointers, asynchronous. */
int tb_search_by_violence_batch_dev(tb_ctx* ctx, int npairs, const tb_keypoint* k1, const uint8_t* d1, const int32_t* n1,
                                    int pitch1, const tb_keypoint* k2, const uint8_t* d2, const int32_t* n2, int pitch2,
                                    const int32_t* cell_start2, const int32_t* cell_items2, int img2_width, int img2_height,
                                    int min_level, int max_level, float radius, int th_low, float nratio, int histo_len,
                                    int check_orientation, tb_match* out, int cap, int32_t* out_counts, int32_t* flags);

/* SURVEY 8(f) row 4, second half -- the DBoW2 transform behind Frame::SetBow (src/types/Frame.cpp:267-270:
 * voc->transform(descriptors, mBowVec, mFeatVec, 4); third_part/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1260, FORB.cpp:81-101).
 * The reference tree ships no vocabulary file, so the caller supplies the tree (tb_vocabulary: the arrays of an ORBvoc-style
 * text file, TemplatedVocabulary::loadFromTextFile :1338-1420); tb_vocab_create uploads it once per context.
 * tb_bow_transform: per descriptor the word it falls into (word_ids), that word's weight (weights; 0 = a stopped word, which
 * enters neither vector) and its ancestor at level L - levelsup (node_ids: the key of the frame's FeatureVector; the root when
 * L - levelsup <= 0; where a branch ends above that level the reference leaves the id unset -- here it is the leaf).
 * Host pointers. The BowVector / FeatureVector containers are built from these arrays (shim: Frame::SetBow). */
typedef struct tb_vocab tb_vocab;
int tb_vocab_create(tb_ctx* ctx, const tb_vocabulary* host, tb_vocab** out);
void tb_vocab_destroy(tb_vocab* v);
int tb_bow_transform(tb_ctx* ctx, const tb_vocab* voc, const uint8_t* desc, int n, int levelsup, int32_t* word_ids,
                     double* weights, int32_t* node_ids);
/* Batched, device-resident: frame f has counts[f] descriptors at desc + f * desc_pitch * 32. Outputs [nframes][desc_pitch]:
 * word_ids / node_ids / weights (each nullable), and fv_keys (nullable, uint64): the frame's FeatureVector as a sorted list --
 * (node id << 32 | feature index) of every feature whose word is not stopped, ascending, i.e. the std::map's node order
 * with every node's features in insertion order; fv_counts[f] entries. Feeds tb_search_by_bow_batch_dev. Device pointers,
 * asynchronous on the context's stream; desc_pitch <= 8192. */
int tb_bow_transform_batch_dev(tb_ctx* ctx, const tb_vocab* voc, int nframes, const uint8_t* desc, const int32_t* counts,
                               int desc_pitch, int levelsup, int32_t* word_ids, int32_t* node_ids, double* weights,
                               uint64_t* fv_keys, int32_t* fv_counts);
/* Batched, device-resident Matcher::searchByBow(F1, F2, MapPointOnly) (matcher.cpp:619-721) on feature vectors in the
 * sorted-list form above: pair p matches frame p of side 1 against frame p of side 2 (keys / descriptors [npairs][pitchX],
 * fv keys [npairs][pitchX] with fv_countsX[p] entries; has_mp2 nullable [npairs][pitch2]). Matches [npairs][cap] in the
 * reference's order, out_counts[p]; flags[p] != 0: a rotation bin outside the histogram (the reference asserts). */
int tb_search_by_bow_batch_dev(tb_ctx* ctx, int npairs, const tb_keypoint* k1, const uint8_t* d1, int pitch1,
                               const uint64_t* fv1, const int32_t* fv_counts1, const tb_keypoint* k2, const uint8_t* d2, int pitch2,
                               const uint64_t* fv2, const int32_t* fv_counts2, const uint8_t* has_mp2, int map_point_only,
                               int th_low, float nratio, int histo_len, int check_orientation, tb_match* out, int cap,
                               int32_t* out_counts, int32_t* flags);

/* Stereo tracks -> PoseOptimization's inputs, batched and device-resident (round 3). For frame f and each of its
 * match_counts[f] left <-> right matches (queryIdx = left key, trainIdx = right key; the output of
 * tb_search_by_bf_batch_dev): Depth = bf / |x_right - x_left| (LocalBA::AddMapPointsByStereo, LocalBA.cpp:60-64), the map
 * point = the left key back-projected with that depth (test/test_vo.cpp:257-267), observed at the right key's pixel with
 * invSigma2[octave of the right key] (LocalBA.cpp:333-363) -- one tb_obs row per match, in match order; matches without
 * disparity or with an octave outside the table are dropped. keys_left / keys_right: [nframes][key_pitch] records; K = fx, fy,
 * cx, cy; inv_sigma2: nlevels floats (host; tb_scale_factors). obs [nframes][obs_pitch], obs_counts [nframes]. Device
 * pointers except K / inv_sigma2; asynchronous on the context's stream. PoseOptimization started at the identity on these rows
 * finds the right camera's pose. */
int tb_stereo_tracks_to_obs_batch_dev(tb_ctx* ctx, int nframes, const tb_keypoint* keys_left, const tb_keypoint* keys_right,
                                      int key_pitch, const tb_match* matches, const int32_t* match_counts, int match_pitch,
                                      const float K[4], float bf, const float* inv_sigma2, int nlevels, tb_obs* obs, int obs_pitch,
                                      int32_t* obs_counts);

/* ---------------------------------------------------------------- pose optimisation / local BA
 * LocalBA::PoseOptimization, LocalBA.cpp:291-490. K = fx,fy,cx,cy. Tcw_in/out: row-major 4x4.
 * outlier: n in/out flags (Frame::GetOutlier/SetOutlier). *n_inliers = nInitialCorrespondences - nBad.
 * stats (nullable, 8 doubles): LM iterations, final robust chi2, final lambda, nBad, t[3], q.w. */
int tb_pose_opt(tb_ctx* ctx, const double K[4], const float Tcw_in[16], const tb_obs* obs, int n,
                uint8_t* outlier, float Tcw_out[16], int* n_inliers, double* stats);
/* Batched device form: problem p reads obs + p*obs_pitch (counts[p] rows), Tcw_in + 16p, outlier +
 * p*obs_pitch; writes Tcw_out + 16p, n_inliers[p], stats + 8p (nullable). Device pointers. */
int tb_pose_opt_batch_dev(tb_ctx* ctx, int nproblems, const double K[4], const float* Tcw_in,
                          const tb_obs* obs, const int32_t* counts, int obs_pitch, uint8_t* outlier,
                          float* Tcw_out, int32_t* n_inliers, double* stats);
/* SURVEY 8(f) row 2, first part -- the optical-flow matcher.
 * tb_optical_flow_pyr_lk replaces the call cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err,
 * Size(win, win), max_level) of matcher.cpp:744 (default criteria: 30 iterations / eps 0.01, flags 0,
 * minEigThreshold 1e-4). win must be 21 (the reference's), max_level 0..5. Host pointers; prev_pts / next_pts are
 * n (x, y) pairs; err nullable; *top_level (nullable) = coarsest pyramid level used. OpenCV is not part of the
 * reference tree: the routine is restated, parity UNPINNED (oracle/oracle_flow.cpp says what was restated and the one
 * deliberate difference, exact integer window sums).
 * tb_search_by_opflow replaces Matcher::searchByOPFlow(F1, F2, cur_points, equalized, reject), matcher.cpp:724-768:
 * tracks F2's keys (keys2_xy) from img2 into img1, clears the points that leave F1's frame (cam1->width / height,
 * CameraModel.h:33-39) and returns DMatch(i, i) records (distance FLT_MAX, imgIdx -1, as a default-constructed
 * cv::DMatch). equalized != 0: img1 goes through tb_clahe(3.0, 8 x 8) first (F1->Equalize(), matcher.cpp:736-739).
 * reject != 0: Matcher::rejectWithF (matcher.cpp:853-881) = cv::findFundamentalMat(FM_RANSAC, 1.0, 0.99) clears the
 * flags of the epipolar outliers before the matches are listed (tb_reject_with_f below). cur_points: n (x, y) pairs out. */
/* Frame::Equalize, Frame.cpp:453-458: cv::createCLAHE(clip_limit = 3.0, Size(tiles_x, tiles_y) = 8 x 8)->apply(src, dst)
 * (OpenCV 3.3 routine restated, parity unpinned). Host pointers; dst has the size of src. tb_clahe_dev: device pointers,
 * asynchronous on the context's stream. */
int tb_clahe(tb_ctx* ctx, const uint8_t* src, int width, int height, int stride, double clip_limit, int tiles_x, int tiles_y,
             uint8_t* dst, int dst_stride);
int tb_clahe_dev(tb_ctx* ctx, const uint8_t* src, int width, int height, int stride, double clip_limit, int tiles_x, int tiles_y,
                 uint8_t* dst, int dst_stride);
int tb_optical_flow_pyr_lk(tb_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height, int stride,
                           const float* prev_pts, int n, int win, int max_level, float* next_pts, uint8_t* status,
                           float* err, int* top_level);
int tb_search_by_opflow(tb_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int width, int height, int stride,
                        const tb_camera* cam1, const float* keys2_xy, int n, int equalized, int reject,
                        float* cur_points, tb_match* out, int cap, int* count);
/* Batched device-resident searchByOPFlow: npairs (F1, F2) image pairs of one geometry (pair p at img1 / img2 +
 * p * image_pitch bytes), F2's keys of pair p at keys2_xy + p * pts_pitch (x, y) records (counts[p] of them; counts
 * nullable = pts_pitch each). Outputs per pair: cur_points and status (1 = matched, after the IsInFrame filter) at slot
 * p * pts_pitch, DMatch(i, i) records in index order at out + p * cap, their number in out_counts[p] (clamped to cap).
 * cam1 is a HOST pointer (only width / height are read). Asynchronous on the context's stream. */
int tb_search_by_opflow_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* img1, const uint8_t* img2, int width, int height,
                                  int stride, size_t image_pitch, const tb_camera* cam1, const float* keys2_xy,
                                  const int32_t* counts, int pts_pitch, int equalized, int reject, float* cur_points,
                                  uint8_t* status, tb_match* out, int cap, int32_t* out_counts);
/* Device-resident form of the tracker: images, points and outputs in HBM, asynchronous on the context's stream. */
int tb_optical_flow_pyr_lk_dev(tb_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height, int stride,
                               const float* prev_pts, int n, int win, int max_level, float* next_pts, uint8_t* status,
                               float* err);
/* Batched device form: npairs image pairs of one geometry in one launch per stage (pair p at prev / next +
 * p * image_pitch bytes; its points, results, status and err at slot p * pts_pitch, counts[p] <= pts_pitch of them,
 * counts nullable = pts_pitch each). */
int tb_optical_flow_pyr_lk_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* prev, const uint8_t* next, int width, int height,
                                     int stride, size_t image_pitch, const float* prev_pts, const int32_t* counts,
                                     int pts_pitch, int win, int max_level, float* next_pts, uint8_t* status, float* err);

/* SURVEY 8(f) row 2, second part / row a16 -- the RANSAC stage and the stereo depths.
 * tb_find_fundamental_ransac replaces cv::findFundamentalMat(pts1, pts2, cv::FM_RANSAC, thresh, conf, mask) as
 * Matcher::rejectWithF calls it (matcher.cpp:872): *ok = 1 and mask (n bytes) / F (9 doubles row-major, nullable) / *iters
 * (nullable: RANSAC iterations run) when a mask comes back, *ok = 0 when OpenCV returns none (fewer than 7 points, no model).
 * As in OpenCV, 8..14 points go to the LMedS registrator (fixed iteration count, smallest median error, inliers within
 * sigma; *iters = its iteration count), 15 and more to RANSAC -- host and batched entry points alike. OpenCV 3.3 routines restated, PARITY UNPINNED: the sampling
 * (cv::RNG((uint64)-1), getSubset, collinearity retries), error measure, model update and iteration budget follow OpenCV's
 * structure; the 7-point solver's null space and cubic roots are computed with + - * / sqrt only (oracle/oracle_fund.cpp).
 * tb_reject_with_f replaces Matcher::rejectWithF(cur_pts, last_pts, status) (matcher.cpp:853-881): n (x, y) pairs each,
 * status in/out. At most 8 keys, fewer than 7 tracked points or no model: the reference reads an empty vector (UB); here
 * the flags are left as they are. Host pointers. */
int tb_find_fundamental_ransac(tb_ctx* ctx, const float* pts1, const float* pts2, int n, double thresh, double conf, uint8_t* mask,
                               double* F, int* iters, int* ok);
int tb_reject_with_f(tb_ctx* ctx, const float* cur_pts, const float* last_pts, int n, uint8_t* status);
/* Batched, device-resident Matcher::rejectWithF: pair p has counts[p] keys (all pts_pitch of them when counts is null);
 * cur_pts / last_pts [npairs][pts_pitch][2] floats, status [npairs][pts_pitch] in/out. One workgroup per pair; every pair
 * takes the branch the host form takes for its number of tracked points (none / seven-point / LMedS / RANSAC).
 * Asynchronous on the context's stream. */
int tb_reject_with_f_batch_dev(tb_ctx* ctx, int npairs, const float* cur_pts, const float* last_pts, const int32_t* counts,
                               int pts_pitch, uint8_t* status);
/* LocalBA::AddMapPointsByStereo(current_frame, stereo_frame, bf, fx), LocalBA.cpp:46-68: searchByOPFlow(stereo, current,
 * pts, equalized = true, reject = true), then depth[i] = bf / fabsf(pts[i].x - key[i].x) for the surviving keys i of the
 * current frame and -1 for the others (fx is unused by the reference; its drawing and imshow are dropped). img_stereo /
 * img_current: level-0 images; cam_stereo: the stereo frame's camera (width / height for IsInFrame); keys_xy: the current
 * frame's keys. *n_depth = number of depths set. Host pointers. */
int tb_add_map_points_by_stereo(tb_ctx* ctx, const uint8_t* img_stereo, const uint8_t* img_current, int width, int height, int stride,
                                const tb_camera* cam_stereo, const float* keys_xy, int n, float bf, float* depth, int* n_depth);
/* Batched device form: pair p's images at + p * image_pitch, its keys / tracked points / status / depths at slot
 * p * pts_pitch (counts[p] valid, counts nullable = pts_pitch each). cam_stereo is a HOST pointer. Asynchronous. */
int tb_add_map_points_by_stereo_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* img_stereo, const uint8_t* img_current, int width,
                                          int height, int stride, size_t image_pitch, const tb_camera* cam_stereo, const float* keys_xy,
                                          const int32_t* counts, int pts_pitch, float bf, float* cur_points, uint8_t* status,
                                          float* depth);

/* Multi-keyframe local BA -- north-star extension, NO reference counterpart.
 * poses: nkf x 16 (in/out, first nfixed held), pts: npt x 3 (in/out). A point is observed at most once per
 * keyframe (repeated (kf, pt) pairs are rejected like out-of-range indices); at most 64 free keyframes (128 in all),
 * 2^25 points and 99 iterations per window (TB_EUNSUPPORTED beyond). Windows with up to 10 free keyframes run the
 * MFMA-tiled Schur path; 11..64 (reduced systems up to 384 x 384) the generic large-window kernels. stats (nullable, 8 doubles):
 * iterations, initial chi2, final chi2, final lambda. */
int tb_local_ba(tb_ctx* ctx, const double K[4], int nkf, int nfixed, float* poses, int npt, float* pts,
                const tb_ba_obs* obs, int nobs, int iters, double* stats);
/* Batched device form: nwindows equally sized windows; window w uses poses + w*nkf*16, pts + w*npt*3,
 * obs + w*obs_pitch (obs_counts[w] rows, GROUPED BY ASCENDING POINT INDEX), stats + 8w (nullable;
 * stats[7] = -1 flags a window whose observations were out of range / not grouped / repeated). Device pointers.
 * Synchronises the stream once (LM termination is data dependent). */
int tb_local_ba_batch_dev(tb_ctx* ctx, int nwindows, const double K[4], int nkf, int nfixed, float* poses, int npt,
                          float* pts, const tb_ba_obs* obs, const int32_t* obs_counts, int obs_pitch, int iters,
                          double* stats);

/* ---- multi-GPU batch entry (SURVEY.md section 8(b) `tb_batch_run`, 8(e): frames are independent units through
 * extract -> left/right match, sharded as contiguous blocks of frames, one exchange step at the end).
 * The in-process counterpart of trackingbench_slam_amd/dist.py for a C++ host that holds one context per GPU:
 * frame f of the batch goes to context f * ngpu / nframes; every context's chain -- Frame::ComputePyramid (Frame.cpp:414-427),
 * ORBExtractor::operator() (ORBextractor.cpp:906-978) on the left and the right image, Matcher::searchByBF left <-> right
 * (matcher.cpp:168-228) -- is queued on its own stream before any context is waited for, so the GPUs work concurrently; then
 * the per-frame track records are gathered into the caller's HOST arrays (no collective: the records of a shard come straight
 * from its GPU).  ctxs may name the same device more than once (the shards then share it).
 *   left, right   host frames [nframes][height][stride] (pitch bytes apart)
 *   kps / desc    [2][nframes][cap] records / [2][nframes][cap][32] bytes: side 0 = left, 1 = right
 *   counts        [2][nframes]; matches [nframes][cap] (queryIdx = left key, trainIdx = right key), match_counts [nframes]
 * cap must hold every frame's keypoints and matches (TB_ECAPACITY otherwise). */
typedef struct tb_batch_params {
    int width, height, nlevels;
    float scale;                /* Frame::Frame scale step (0.8) */
    int target;                 /* keypoints per image */
    float init_th, min_th;      /* FAST thresholds */
    float bf_ratio, bf_min_th;  /* searchByBF */
} tb_batch_params;
int tb_batch_run(tb_ctx** ctxs, int ngpu, const tb_batch_params* p, int nframes, const uint8_t* left, const uint8_t* right,
                 int stride, size_t pitch, int cap, tb_keypoint* kps, uint8_t* desc, int32_t* counts, tb_match* matches,
                 int32_t* match_counts);

#ifdef __cplusplus
}
#endif
#endif /* TB_CAPI_H */
