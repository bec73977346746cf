/* FP64 SE(3) helpers shared by the pose-optimisation and local-BA kernels: the g2o SE3Quat /
 * Eigen quaternion operations that VertexSE3Expmap and EdgeSE3ProjectXYZ(OnlyPose) use, restated for
 * the device (see k_pose.hip for the reference call sites). */
#ifndef TB_SE3_H
#define TB_SE3_H
#include "tb_device.h"

struct PoSE3 { double qx, qy, qz, qw, tx, ty, tz; };

__device__ inline void po_quat_from_R(const double* R, PoSE3& s) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        s.qw = 0.5 * t;
        t = 0.5 / t;
        s.qx = (R[7] - R[5]) * t;
        s.qy = (R[2] - R[6]) * t;
        s.qz = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) { /* Eigen picks the largest diagonal entry: i = 0 */
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        s.qx = 0.5 * t;
        t = 0.5 / t;
        s.qw = (R[7] - R[5]) * t;
        s.qy = (R[3] + R[1]) * t;
        s.qz = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) { /* i = 1 */
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        s.qy = 0.5 * t;
        t = 0.5 / t;
        s.qw = (R[2] - R[6]) * t;
        s.qz = (R[7] + R[5]) * t;
        s.qx = (R[1] + R[3]) * t;
    } else { /* i = 2 */
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        s.qz = 0.5 * t;
        t = 0.5 / t;
        s.qw = (R[3] - R[1]) * t;
        s.qx = (R[2] + R[6]) * t;
        s.qy = (R[5] + R[7]) * t;
    }
}
__device__ inline void po_normalize(PoSE3& s) {
    if (s.qw < 0) { s.qx = -s.qx; s.qy = -s.qy; s.qz = -s.qz; s.qw = -s.qw; }
    const double n = sqrt(s.qx * s.qx + s.qy * s.qy + s.qz * s.qz + s.qw * s.qw);
    s.qx /= n; s.qy /= n; s.qz /= n; s.qw /= n;
}
__device__ inline void po_rot(const PoSE3& s, const double* v, double* o) {
    double uv0 = s.qy * v[2] - s.qz * v[1], uv1 = s.qz * v[0] - s.qx * v[2], uv2 = s.qx * v[1] - s.qy * v[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    const double c0 = s.qy * uv2 - s.qz * uv1, c1 = s.qz * uv0 - s.qx * uv2, c2 = s.qx * uv1 - s.qy * uv0;
    o[0] = v[0] + s.qw * uv0 + c0;
    o[1] = v[1] + s.qw * uv1 + c1;
    o[2] = v[2] + s.qw * uv2 + c2;
}
__device__ inline void po_map(const PoSE3& s, const double* X, double* o) {
    po_rot(s, X, o);
    o[0] += s.tx; o[1] += s.ty; o[2] += s.tz;
}
__device__ inline void po_to_R(const PoSE3& s, double* R) {
    const double tx = 2 * s.qx, ty = 2 * s.qy, tz = 2 * s.qz;
    const double twx = tx * s.qw, twy = ty * s.qw, twz = tz * s.qw;
    const double txx = tx * s.qx, txy = ty * s.qx, txz = tz * s.qx;
    const double tyy = ty * s.qy, tyz = tz * s.qy, tzz = tz * s.qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ inline PoSE3 po_from_Rt(const double* R, const double* t) {
    PoSE3 s;
    po_quat_from_R(R, s);
    po_normalize(s);
    s.tx = t[0]; s.ty = t[1]; s.tz = t[2];
    return s;
}
/* SE3Quat::exp(update) * T */
__device__ inline PoSE3 po_exp_mul(const double* u, const PoSE3& T) {
    const double om[3] = {u[0], u[1], u[2]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
    double a, b, c, d;
    if (theta < 0.00001) { a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0; }
    else {
        a = sin(theta) / theta;
        b = (1 - cos(theta)) / (theta * theta);
        c = b;
        d = (theta - sin(theta)) / (theta * theta * theta);
    }
    double R[9], V[9], t[3];
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + c * Om[i] + d * Om2[i];
    }
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    const PoSE3 E = po_from_Rt(R, t);
    PoSE3 r;
    r.qw = E.qw * T.qw - E.qx * T.qx - E.qy * T.qy - E.qz * T.qz;
    r.qx = E.qw * T.qx + E.qx * T.qw + E.qy * T.qz - E.qz * T.qy;
    r.qy = E.qw * T.qy + E.qy * T.qw + E.qz * T.qx - E.qx * T.qz;
    r.qz = E.qw * T.qz + E.qz * T.qw + E.qx * T.qy - E.qy * T.qx;
    const double tt[3] = {T.tx, T.ty, T.tz};
    double rt[3];
    po_rot(E, tt, rt);
    r.tx = E.tx + rt[0]; r.ty = E.ty + rt[1]; r.tz = E.tz + rt[2];
    po_normalize(r);
    return r;
}
/* un-pivoted Cholesky solve of the 6x6 system; false if not positive definite */
__device__ inline bool po_chol6(const double* H, double lambda, const double* b, double* x) {
    double A[36];
    for (int i = 0; i < 36; i++) A[i] = H[i];
    for (int i = 0; i < 6; i++) A[i * 6 + i] += lambda;
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) return false;
        d = sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * 6 + k] * y[k];
        y[i] = s / A[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < 6; k++) s -= A[k * 6 + i] * x[k];
        x[i] = s / A[i * 6 + i];
    }
    return true;
}

__device__ __forceinline__ double po_wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
/* block sum of NV per-thread values; result valid in every thread; red: LDS >= 4*NV doubles */
template <int NV>
__device__ __forceinline__ void po_block_sum(double (&v)[NV], double* red) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = po_wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = (red[i] + red[NV + i]) + (red[2 * NV + i] + red[3 * NV + i]);
}

#endif
