/* Minimal stand-ins for the OpenCV 3.x types the reference's hot-path headers name (cv::Mat of CV_8UC1,
 * cv::KeyPoint, cv::DMatch, cv::Point2f, cv::OutputArray). Used only when <opencv2/core.hpp> is absent
 * (it is absent in the build image); layouts of KeyPoint / DMatch match OpenCV's so the
 * C ABI can take vectors of them directly. With real OpenCV installed this header is not included.
 */
#ifndef TB_COMPAT_CV_LITE_H
#define TB_COMPAT_CV_LITE_H

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

typedef unsigned char uchar;   /* OpenCV's cvdef.h declares it at global scope (the reference writes std::vector<uchar>) */

namespace cv {

using ::uchar;

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_& operator*=(T s) { x *= s; y *= s; return *this; }
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct KeyPoint {
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.4e38f) {}
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
};
static_assert(sizeof(DMatch) == 16, "cv::DMatch layout");

/* 8-bit single-channel matrix with shared ownership (enough for images and 32-byte descriptor rows). */
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c), data((uchar*)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && own_) return;
        rows = r; cols = c; step = (size_t)c;
        own_.reset(new std::vector<uchar>((size_t)r * c, 0));
        data = own_->data();
    }
    void release() { own_.reset(); rows = cols = 0; step = 0; data = nullptr; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Mat clone() const {
        Mat m(rows, cols, CV_8UC1);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols);
        return m;
    }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(data + (size_t)r * step)[c]; }
    Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.step = step; m.data = data + (size_t)r * step; m.own_ = own_; return m; }
private:
    std::shared_ptr<std::vector<uchar>> own_;
};

/* The reference passes `cv::OutputArray descriptors`; only create()/release()/getMat() are used. */
class _OutputArray {
public:
    _OutputArray() : m_(nullptr) {}
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { if (m_) m_->create(r, c, t); }
    void release() const { if (m_) m_->release(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool needed() const { return m_ != nullptr; }
private:
    Mat* m_;
};
typedef const _OutputArray& OutputArray;
inline _OutputArray noArray() { return _OutputArray(); }

}  // namespace cv
#endif
