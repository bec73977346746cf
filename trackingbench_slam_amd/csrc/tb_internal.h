/* Internal declarations of libtb_hip.so (not part of the C ABI). */
#ifndef TB_INTERNAL_H
#define TB_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/tb_capi.h"

#define TB_MAX_LEVELS 16
#define TB_BORDER 16          /* EDGE_THRESHOLD - 3, ORBextractor.cpp:749 */
#define TB_NODE_CAP_MAX 2048  /* quadtree list capacity that fits LDS (k_octree.hip) */

struct tb_ctx {
    int device = 0;
    int num_cu = 256;   /* compute units of the device (tb_create): launch shapes that aim at one resident round */
    std::vector<std::pair<std::string, hipGraphExec_t>> ba_graphs; /* captured local-BA calls of small batches (k_ba.hip) */
    int peers = 1;      /* tb_set_concurrency: contexts expected to keep this GPU busy at the same time */
    int dbg_fast_dense = 0; /* tb_debug_force_dense_fast: every FAST block takes the any-density path (test hook) */
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, tb_extractor*> plans; /* cached single-frame plans */
    std::set<tb_extractor*> live;               /* every plan created on this context */
    /* per-kernel HIP-event timing (tb_profile_*) */
    bool prof = false;
    bool prof_open = false;   /* the last tb_prof_begin recorded its start event */
    std::string prof_only;    /* non-empty: time only this kernel (tb_profile_only) */
    struct ProfRec { const char* name; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::map<std::string, std::pair<long, double>> prof_acc;
    /* grow-only device scratch for the matcher / pose entry points */
    void* scratch[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

int tb_fail(tb_ctx* ctx, int code, const char* fmt, ...);
void tb_prof_begin(tb_ctx* ctx, const char* name);
void tb_prof_end(tb_ctx* ctx);
int tb_scratch(tb_ctx* ctx, int slot, size_t bytes, void** out);

#define TB_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return tb_fail((ctx), TB_EDEVICE, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                               \
    } while (0)

/* First statement of every entry point that takes a context or a plan: the HIP "current device" is per host thread and
 * the caller may hold contexts on several GPUs (or drive one context from several threads in turn), so every call binds
 * its thread to the context's device before it launches, copies or allocates. A null context falls through to the
 * function's own argument check. */
#define TB_ENTER(ctx)                                                                          \
    do {                                                                                       \
        tb_ctx* c_ = (ctx);                                                                    \
        if (c_) {                                                                              \
            hipError_t e_ = hipSetDevice(c_->device);                                          \
            if (e_ != hipSuccess) { /* tb_fail itself makes no HIP call and must not come back here */ \
                c_->err = std::string("hipSetDevice: ") + hipGetErrorString(e_);               \
                return TB_EDEVICE;                                                             \
            }                                                                                  \
        }                                                                                      \
    } while (0)

/* Geometry of one pyramid level inside a plan (host and device copies are identical). */
struct LevelGeom {
    int w, h, stride;     /* stride of the slab copy of this level */
    int nCols, nRows;     /* 30-px cell grid, ORBextractor.cpp:757-763 */
    int wCell, hCell;
    int cellBase;         /* index of this level's first entry in the cell table */
    int nCells;           /* valid (not skipped) cells of this level */
    int candCap;          /* capacity of the per-image candidate array of this level */
    int nodeCap;          /* quadtree list capacity = quota + 3 + 4*nIni (this call) */
    int nodeCapAlloc;     /* slots reserved for this level in the selection array */
    int nIni;             /* DistributeOctTree initial nodes, ORBextractor.cpp:498 */
    int quota;            /* mnFeaturesPerLevel[level] of the current call */
    int selBase;          /* first slot of this level in the per-image selection array */
    float hX;             /* (maxX-minX)/nIni as float, ORBextractor.cpp:500 */
    float patchSize;      /* (float)(int)(31*sf[level]), ORBextractor.cpp:812 */
    float sf;             /* mvScaleFactor[level] */
    float inv_sf;         /* FASTExtractor scale argument (fastgrid only) */
    unsigned long long off;     /* byte offset of the level inside one image's slab */
    unsigned long long candOff; /* element offset of the level inside one image's candidate array */
};

struct PlanGeom {
    int nlevels;
    int width, height;
    int selCap;                      /* slots per image in the selection / result arrays */
    unsigned long long slabBytes;    /* bytes per image slab */
    unsigned long long candPerImage; /* candidate records per image */
    /* level-0 source: external frames (dev pointer) or the slab */
    const uint8_t* img0;
    unsigned long long img0_pitch;
    int img0_stride;
    int pad_;
    LevelGeom lv[TB_MAX_LEVELS];
};

/* One FAST work item (k_fast_blocks): a block of ncx x ncy adjacent 30-px cells of one level, staged in LDS once.
 * The ROI is the union of the cells' cv::FAST ROIs (ORBextractor.cpp:765-786): cell (i, j) of the block scans columns
 * [x0 + 3 + j wCell, min(x0 + 3 + (j + 1) wCell, x1 - 3)) and rows likewise -- the cells' scan regions tile the ROI's. */
struct FastBlock {
    int16_t level;
    int16_t ncx, ncy;       /* cells of this block (<= FB_MAX_CX x FB_MAX_CY) */
    int16_t sA;             /* stage-1 lane map, fixed per block (host arithmetic, k_fast.hip): first 16-byte tile segment */
    int16_t x0, y0, x1, y1; /* ROI in absolute level coordinates, [x0,x1) x [y0,y1) */
    int16_t nss, rowsPer;   /*   that holds scanned columns, number of such segments, rows per 64-lane pass = 64 / nss, */
    int16_t nPass;          /*   passes over the scanned rows, */
    uint16_t invNss;        /*   ceil(32768 / nss): lane / nss == (lane * invNss) >> 15 for lane < 64 */
};
#define FB_S 160            /* LDS row stride of a block tile (bytes, 10 x 16) */
#ifndef FB_TH
#define FB_TH 68            /* tile rows */
#endif
#define FB_MAX_CX 4
#ifndef FB_MAX_CY
#define FB_MAX_CY 2
#endif

/* resize tables, built on the host with the oracle-identical double/float arithmetic */
struct ResizeX { int16_t sx, sx1, a0, a1; };
struct ResizeY { int32_t sy0, sy1; int16_t b0, b1; };

struct tb_extractor {
    tb_ctx* ctx = nullptr;
    PlanGeom g;                /* host copy; img0* fields updated per call */
    int max_images = 0, max_target = 0;
    std::vector<float> sf;
    bool have_quotas = false;
    int quotas[TB_MAX_LEVELS];
    int last_n = 0;
    bool last_was_orb = false;
    /* device memory */
    uint8_t* d_slab = nullptr;          /* [max_images][slabBytes] */
    uint8_t* d_img0_copy = nullptr;     /* [max_images][h][stride0] for host-provided frames */
    FastBlock* d_blocks = nullptr; int nBlocksTotal = 0;   /* FAST work items of one image, level-major */
    ResizeX* d_rx[TB_MAX_LEVELS]; ResizeY* d_ry[TB_MAX_LEVELS];
    uint32_t* d_cand = nullptr;         /* [max_images][candPerImage] packed score<<24|y<<12|x */
    int32_t* d_candCount = nullptr;     /* [max_images][TB_MAX_LEVELS] */
    uint32_t* d_knode = nullptr;        /* [max_images][candPerImage] quadtree scratch */
    uint32_t* d_sel = nullptr;          /* [max_images][selCap] selected keypoints, packed */
    int32_t* d_selCount = nullptr;      /* [max_images][TB_MAX_LEVELS] */
    tb_keypoint* d_kps = nullptr;       /* [max_images][selCap] */
    uint8_t* d_desc = nullptr;          /* [max_images][selCap][32] */
    int32_t* d_counts = nullptr;        /* [max_images] */
    float* d_exit = nullptr; int exitCap = 0;   /* exit keys (x,y) */
    int32_t* d_enode = nullptr; size_t enodeCap = 0;
    /* fastgrid scratch */
    unsigned long long* d_gridBest = nullptr; size_t gridBestCap = 0;
    uint8_t* d_occ = nullptr; size_t occCap = 0;
};

/* kernel launchers (k_*.hip) */
int tbk_resize_level(tb_extractor* ex, int level, int n);
int tbk_fast_cells(tb_extractor* ex, int n, int init_th, int min_th);
int tbk_octree(tb_extractor* ex, int n, int n_exit);
int tbk_describe(tb_extractor* ex, int n);
int tbk_fast_image(tb_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, int th, int nms, int arc,
                   uint32_t* d_out, int cap, int32_t* d_count);
int tbk_fastgrid(tb_extractor* ex, int n, int target, float threshold, int n_occ);
int tbk_bf_batch(tb_ctx* ctx, int npairs, const uint8_t* d1, const int32_t* c1, const uint8_t* d2,
                 const int32_t* c2, size_t set_pitch, int max_n, int crosscheck, int filter, float ratio,
                 float min_th, tb_match* out, int cap, int32_t* out_counts, unsigned long long* d_tbest,
                 unsigned long long* d_qbest);
int tbk_window_match(tb_ctx* ctx, const tb_keypoint* d_k1, const uint8_t* d_d1, int n1, const tb_keypoint* d_k2,
                     const uint8_t* d_d2, int n2, const int32_t* d_cellStart, const int32_t* d_cellItems,
                     float widthInv, float heightInv, int min_level, int max_level, float r,
                     int32_t* d_best /* n1 x 4: bestDist, bestDist2, bestIdx, #candidates */);
int tbk_violence_batch(tb_ctx* ctx, int npairs, const tb_keypoint* d_k1, const uint8_t* d_d1, const int32_t* d_n1, int pitch1,
                       const tb_keypoint* d_k2, const uint8_t* d_d2, const int32_t* d_n2, int pitch2, const int32_t* d_cellStart,
                       const int32_t* d_cellItems, int img2_w, int img2_h, int min_level, int max_level, float radius, int th_low,
                       float nratio, int histo_len, int check_orientation, int32_t* d_best, tb_match* d_out, int cap,
                       int32_t* d_out_counts, int32_t* d_flags);
/* searchByBow's search over shared vocabulary nodes (queries: int4 idx1, start2, end2, 0; best: int4) */
int tbk_bow_search(tb_ctx* ctx, int nq, const void* d_queries, const uint8_t* d_d1, const uint8_t* d_d2, const uint32_t* d_items2,
                   const uint8_t* d_has_mp2, int map_point_only, void* d_best);
/* device-resident lookup grids and the batched searchByProjection(F1, F2) on them */
int tbk_grid_build_batch(tb_ctx* ctx, int nframes, const tb_keypoint* d_keys, const int32_t* d_counts, int key_pitch, int img_w,
                         int img_h, int32_t* d_cellStart, int32_t* d_cellItems);
int tbk_projection_batch(tb_ctx* ctx, int npairs, const float* d_Tcw, const tb_camera* cam, int img_w, int img_h,
                         const tb_keypoint* d_k1, const uint8_t* d_d1, const uint8_t* d_taken1, const int32_t* d_n1, int pitch1,
                         const int32_t* d_cellStart, const int32_t* d_cellItems, const tb_keypoint* d_k2, const tb_mappoint* d_mp2,
                         const uint8_t* d_mp2d, const int32_t* d_n2, int pitch2, const float* sf, int nlevels, float nratio,
                         int th_high, int histo_len, int check_orientation, int32_t* d_best, tb_match* d_out, int cap,
                         int32_t* d_out_counts, int32_t* d_flags, int map_mode, float radio, int max_n2);
/* searchByProjection: project nq map points into F1 and search F1's lookup grid; best[6 nq] */
int tbk_projection_search(tb_ctx* ctx, int map_overload, const float Tcw[16], const tb_camera* cam, const tb_keypoint* d_k2,
                          const tb_mappoint* d_mp, const uint8_t* d_mpdesc, int nq, const float* d_sf, int nlevels, float sf0,
                          float nratio, const tb_keypoint* d_k1, const uint8_t* d_d1, const uint8_t* d_taken1,
                          const int32_t* d_cellStart, const int32_t* d_cellItems, float widthInv, float heightInv,
                          void* d_queries, int32_t* d_best, int* d_flag);
int tbk_bow_transform(tb_ctx* ctx, int nnodes, int L, const int32_t* d_child_start, const int32_t* d_child_items, const uint8_t* d_vdesc,
                      const int32_t* d_word_id, const double* d_weight, int nframes, const uint8_t* d_desc, const int32_t* d_counts,
                      int desc_pitch, int levelsup, int32_t* d_word_ids, int32_t* d_node_ids, double* d_weights,
                      unsigned long long* d_fv_keys, int32_t* d_fv_counts);
int tbk_bow_search_batch(tb_ctx* ctx, int npairs, const tb_keypoint* d_k1, const uint8_t* d_d1, int pitch1, const unsigned long long* d_fv1,
                         const int32_t* d_n1, const tb_keypoint* d_k2, const uint8_t* d_d2, int pitch2, const unsigned long long* d_fv2,
                         const int32_t* d_n2, const uint8_t* d_has_mp2, int map_point_only, int th_low, float nratio, int histo_len,
                         int check_orientation, tb_match* d_out, int cap, int32_t* d_out_counts, int32_t* d_flags, int32_t* d_best);
int tbk_pack_rows(tb_ctx* ctx, const void* d_src, int row_bytes, int cap, const int32_t* d_counts, int nframes, void* d_dst, long long* d_total);
int tbk_copy16(tb_ctx* ctx, const void* d_src, void* d_dst, size_t bytes);
int tbk_stereo_obs(tb_ctx* ctx, int nframes, const tb_keypoint* d_kl, const tb_keypoint* d_kr, int key_pitch, const tb_match* d_matches,
                   const int32_t* d_match_counts, int match_pitch, const float K[4], float bf, const float* d_inv_sigma2, int nlevels,
                   tb_obs* d_obs, int obs_pitch, int32_t* d_obs_counts);
int tbk_pose_batch(tb_ctx* ctx, int nproblems, const double K[4], const float* Tcw_in, const tb_obs* obs,
                   const int32_t* counts, int obs_pitch, uint8_t* outlier, float* Tcw_out, int32_t* n_inliers,
                   double* stats, double* d_err);
int tbk_local_ba_batch(tb_ctx* ctx, int W, const double K[4], int nkf, int nfixed, float* d_poses, int npt, float* d_pts,
                       const tb_ba_obs* d_obs, const int32_t* d_counts, int obs_pitch, int iters, double* d_stats, void* d_work,
                       size_t work_bytes);
size_t tbk_local_ba_work_bytes(const tb_ctx* ctx, int W, int nkf, int nfixed, int npt, int obs_pitch);
int tbk_clahe(tb_ctx* ctx, int nimg, const uint8_t* d_src, int w, int h, int stride, size_t spitch, double clip_limit, int tiles_x,
              int tiles_y, uint8_t* d_dst, int dstride, size_t dpitch, uint8_t* d_lut);
int tbk_flow_accept(tb_ctx* ctx, int npairs, const float* d_cur, uint8_t* d_status, const int32_t* d_counts, int pts_pitch, int width,
                    int height, tb_match* d_out, int cap, int32_t* d_out_counts);
/* RANSAC fundamental matrix (k_ransac.hip): mode 0 = Matcher::rejectWithF on the flagged points, 1 = cv::findFundamentalMat on all */
size_t tbk_ransac_work_bytes(int npairs, int pts_pitch);
int tbk_ransac_f(tb_ctx* ctx, int npairs, const float* d_pts1, const float* d_pts2, uint8_t* d_status, const int32_t* d_counts,
                 int pts_pitch, int mode, double thresh, double conf, void* d_work, int32_t* d_flags, double* d_F, int32_t* d_iters);
int tbk_stereo_depth(tb_ctx* ctx, int npairs, const float* d_cur, const float* d_keys, const uint8_t* d_status, const int32_t* d_counts,
                     int pts_pitch, float bf, float* d_depth);
size_t tbk_lk_work_bytes(int w, int h, int max_level, int npairs);
int tbk_lk_track(tb_ctx* ctx, int npairs, const uint8_t* d_prev, const uint8_t* d_next, int w, int h, int stride, size_t image_pitch,
                 const float* d_prev_pts, const int32_t* d_counts, int n, int pts_pitch, int win, int max_level, float* d_next_pts,
                 uint8_t* d_status, float* d_err, void* d_work, int* top_level);

#endif
