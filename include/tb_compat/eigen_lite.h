/* Minimal stand-ins for the Eigen fixed-size types named by the reference's hot-path headers
 * (Matrix4f, Matrix3f, Vector3f, Vector2f), used only when <Eigen/Core> is absent. */
#ifndef TB_COMPAT_EIGEN_LITE_H
#define TB_COMPAT_EIGEN_LITE_H

namespace Eigen {

template <typename T, int R, int C> struct Matrix {
    T m[R * C]; /* row-major storage; access only through (r, c) */
    Matrix() { for (int i = 0; i < R * C; i++) m[i] = T(0); }
    T& operator()(int r, int c) { return m[r * C + c]; }
    const T& operator()(int r, int c) const { return m[r * C + c]; }
    T& operator()(int i) { return m[i]; }
    const T& operator()(int i) const { return m[i]; }
    T& operator[](int i) { return m[i]; }
    const T& operator[](int i) const { return m[i]; }
    T& x() { return m[0]; }
    T& y() { return m[1]; }
    T& z() { return m[2]; }
    const T& x() const { return m[0]; }
    const T& y() const { return m[1]; }
    const T& z() const { return m[2]; }
    static Matrix Identity() { Matrix r; for (int i = 0; i < (R < C ? R : C); i++) r(i, i) = T(1); return r; }
    static Matrix Zero() { return Matrix(); }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> t; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) t(c, r) = (*this)(r, c); return t; }
};
template <typename T, int R, int K, int C>
Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b) {
    Matrix<T, R, C> r;
    for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) { T s = T(0); for (int k = 0; k < K; k++) s += a(i, k) * b(k, j); r(i, j) = s; }
    return r;
}
template <typename T, int R, int C> Matrix<T, R, C> operator-(const Matrix<T, R, C>& a) {
    Matrix<T, R, C> r; for (int i = 0; i < R * C; i++) r.m[i] = -a.m[i]; return r;
}
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<double, 3, 1> Vector3d;

}  // namespace Eigen
#endif
