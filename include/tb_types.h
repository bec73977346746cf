/* Plain-data types that cross the C ABI of the tracking hot path.
 *
 * Layouts mirror the OpenCV 3.x types the reference's operator API passes
 * (reference: include/extractors/ORBextractor.h:38-52, include/matchers/matcher.h:39-62,
 * include/mapping/LocalBA.h:16-22) so a binding can reinterpret_cast vectors of
 * cv::KeyPoint / cv::DMatch without a copy.
 */
#ifndef TB_TYPES_H
#define TB_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct tb_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} tb_keypoint;

/* == cv::DMatch (16 bytes) */
typedef struct tb_match {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} tb_match;

/* One FAST corner: integer pixel position inside the scanned image + score. */
typedef struct tb_corner {
    int32_t x, y, score;
} tb_corner;

/* One motion-only BA observation (reference: src/mapping/LocalBA.cpp:333-363):
 * pixel (Feature::px), world point (MapPoint::GetWorldPos), information scale
 * (Frame::GetInverseScaleSigmaSquares()[octave]). */
typedef struct tb_obs {
    float u, v;
    float X, Y, Z;
    float inv_sigma2;
} tb_obs;

/* Pinhole camera as the projection matchers use it (reference PinholeCamera, CameraModel.cpp:63-93 and
 * CameraModel.h:33-39): focal lengths, principal point, image size for IsInFrame, and the radial-tangential
 * coefficients d[0..4] (k1 k2 p1 p2 k3) applied iff has_distortion. */
typedef struct tb_camera {
    float fx, fy, cx, cy;
    int32_t width, height;
    int32_t has_distortion;
    float d[5];
} tb_camera;

/* A map point as Matcher::searchByProjection / Frame::IsInFrustum read it (reference MapPoint: GetWorldPos,
 * GetNormal, Get{Min,Max}DistanceInvariance, isBad). bad != 0 also stands for "this key has no map point" where
 * the array is aligned with a frame's keys. Descriptors travel separately (32 bytes each). */
typedef struct tb_mappoint {
    float pos[3];
    float normal[3];
    float min_dist, max_dist;
    int32_t bad;
} tb_mappoint;

/* A DBoW2 vocabulary as flat arrays (reference third_part/DBoW2/DBoW2/TemplatedVocabulary.h:297-329, the tree that
 * Frame::SetBow walks, src/types/Frame.cpp:267-270). Node 0 is the root; the children of node n are
 * child_items[child_start[n] .. child_start[n + 1]) in the vocabulary's order (the order decides ties: the first child with
 * the smallest distance wins, TemplatedVocabulary.h:1231-1244); a node without children is a word and carries word_id / weight.
 * desc: 32 bytes per node (FORB::TDescriptor). k / L as in the vocabulary file's first line; weighting: 0 TF_IDF, 1 TF, 2 IDF,
 * 3 BINARY; scoring: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT (BowVector.h:36-53). */
typedef struct tb_vocabulary {
    int32_t nnodes, k, L;
    int32_t weighting, scoring;
    const int32_t* child_start;   /* nnodes + 1 */
    const int32_t* child_items;   /* child_start[nnodes] node ids */
    const uint8_t* desc;          /* nnodes x 32 */
    const int32_t* word_id;       /* nnodes, read for leaves */
    const double* weight;         /* nnodes, read for leaves (WordValue) */
} tb_vocabulary;

/* One local-BA observation: keyframe index, point index, pixel, information scale. */
typedef struct tb_ba_obs {
    int32_t kf, pt;
    float u, v;
    float inv_sigma2;
} tb_ba_obs;

/* Error codes (0 = ok). */
enum {
    TB_OK = 0,
    TB_EINVAL = -1,       /* bad argument (null pointer, non-positive size, ...) */
    TB_ENOMEM = -2,       /* host or device allocation failed */
    TB_ECAPACITY = -3,    /* caller-provided output capacity too small */
    TB_EUNSUPPORTED = -4, /* reference behaviour undefined/broken for this input */
    TB_EDEVICE = -5,      /* HIP runtime error; see tb_last_error() */
    TB_ESTATE = -6        /* call sequence error (e.g. AddPoints before operator()) */
};

#ifdef __cplusplus
}
#endif
#endif /* TB_TYPES_H */
