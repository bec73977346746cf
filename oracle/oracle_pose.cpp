/* TEST INFRASTRUCTURE ONLY -- CPU oracle, pose optimisation (see orc_math.h for the usage rule).
 *
 * orc_pose_opt restates LocalBA::PoseOptimization (src/mapping/LocalBA.cpp:291-490) including the
 * g2o pieces it drives -- OptimizationAlgorithmLevenberg::solve, SparseOptimizer::optimize,
 * BaseUnaryEdge::constructQuadraticForm, RobustKernelHuber, EdgeSE3ProjectXYZOnlyPose,
 * VertexSE3Expmap/SE3Quat -- from their published algorithms (g2o is not in the image, version
 * unpinned: SURVEY.md 8c). PARITY UNPINNED against genuine g2o. Known liberties: the dense 6x6
 * solve is an un-pivoted Cholesky (g2o: Eigen LDLT), SE3Quat::exp uses the second-order small-angle
 * branch, Eigen's internal product association order is not modelled.
 *
 * orc_local_ba is a north-star EXTENSION with no reference counterpart.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Quat { double x, y, z, w; };
struct SE3 { Quat r; double t[3]; };

/* Eigen::Quaterniond(Matrix3d) */
Quat quat_from_R(const double R[9]) {
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[7] - R[5]) * t;
        q.y = (R[2] - R[6]) * t;
        q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        double c[3];
        c[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
        c[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        c[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q.x = c[0]; q.y = c[1]; q.z = c[2];
    }
    return q;
}

/* SE3Quat::normalizeRotation */
void quat_normalize(Quat& q) {
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

Quat quat_mul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

/* Eigen QuaternionBase::_transformVector */
void quat_rot(const Quat& q, const double v[3], double out[3]) {
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
    out[0] = v[0] + q.w * uv[0] + c[0];
    out[1] = v[1] + q.w * uv[1] + c[1];
    out[2] = v[2] + q.w * uv[2] + c[2];
}

/* Eigen QuaternionBase::toRotationMatrix */
void quat_to_R(const Quat& q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

SE3 se3_from_Rt(const double R[9], const double t[3]) {
    SE3 s;
    s.r = quat_from_R(R);
    quat_normalize(s.r);
    s.t[0] = t[0]; s.t[1] = t[1]; s.t[2] = t[2];
    return s;
}

void se3_map(const SE3& T, const double X[3], double out[3]) {
    quat_rot(T.r, X, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}

void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

/* SE3Quat::exp(update), update = [omega, upsilon] */
SE3 se3_exp(const double u[6]) {
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double Om2[9];
    mat3_mul(Om, Om, Om2);
    double R[9], V[9];
    double a, b, c, d;
    if (theta < 0.00001) {
        a = 1.0; b = 0.5; c = 0.5; d = 1.0 / 6.0;
    } else {
        a = std::sin(theta) / theta;
        b = (1 - std::cos(theta)) / (theta * theta);
        c = b;
        d = (theta - std::sin(theta)) / (theta * theta * theta);
    }
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * Om[i] + b * Om2[i];
        V[i] = I + c * Om[i] + d * Om2[i];
    }
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    return se3_from_Rt(R, t);
}

/* SE3Quat::operator* */
SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 r;
    r.r = quat_mul(a.r, b.r);
    double rt[3];
    quat_rot(a.r, b.t, rt);
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    quat_normalize(r.r);
    return r;
}

/* symmetric positive-definite solve (n <= 512), in place on copies. Returns false if not PD. */
bool chol_solve(int n, std::vector<double> A, std::vector<double> b, std::vector<double>& x) {
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k];
        x[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

/* RobustKernelHuber::robustify */
inline void huber(double e, double delta, double rho[3]) {
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e; rho[1] = 1.; rho[2] = 0.;
    } else {
        const double sqrte = std::sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

/* EdgeSE3ProjectXYZOnlyPose::linearizeOplus (pose block; shared with EdgeSE3ProjectXYZ) */
inline void jac_pose(const double pc[3], double fx, double fy, double J[12]) {
    const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
    J[0] = x * y * invz_2 * fx;
    J[1] = -(1 + (x * x * invz_2)) * fx;
    J[2] = y * invz * fx;
    J[3] = -invz * fx;
    J[4] = 0;
    J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy;
    J[7] = -x * y * invz_2 * fy;
    J[8] = -x * invz * fy;
    J[9] = 0;
    J[10] = -invz * fy;
    J[11] = y * invz_2 * fy;
}

}  // namespace

extern "C" {

int orc_pose_opt(const double K[4], const float Tcw_in[16], const tb_obs* obs, int n,
                 uint8_t* outlier, float Tcw_out[16], double* stats) {
    if (!K || !Tcw_in || !Tcw_out || n < 0 || (n && (!obs || !outlier))) return TB_EINVAL;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    for (int i = 0; i < 16; i++) Tcw_out[i] = Tcw_in[i];
    if (stats) for (int i = 0; i < 8; i++) stats[i] = 0;
    const int nInitialCorrespondences = n;
    if (nInitialCorrespondences < 3) return 0; /* LocalBA.cpp:401 */

    double R0[9], t0[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R0[i * 3 + j] = (double)Tcw_in[i * 4 + j];
        t0[i] = (double)Tcw_in[i * 4 + 3];
    }
    const double delta = (double)std::sqrt(5.991f); /* deltaMono, float sqrtf */
    const float chi2Mono = 5.991f;

    std::vector<double> Xw(3 * (size_t)n), ob(2 * (size_t)n), w(n), err(2 * (size_t)n);
    std::vector<int> level(n, 0);
    std::vector<char> robust(n, 1);
    for (int i = 0; i < n; i++) {
        Xw[3 * i] = obs[i].X; Xw[3 * i + 1] = obs[i].Y; Xw[3 * i + 2] = obs[i].Z;
        ob[2 * i] = obs[i].u; ob[2 * i + 1] = obs[i].v;
        w[i] = (double)obs[i].inv_sigma2;
    }
    SE3 est = se3_from_Rt(R0, t0);

    auto compute_error = [&](int i, const SE3& T) {
        double pc[3];
        se3_map(T, &Xw[3 * i], pc);
        const double px = pc[0] / pc[2] * fx + cx, py = pc[1] / pc[2] * fy + cy;
        err[2 * i] = ob[2 * i] - px;
        err[2 * i + 1] = ob[2 * i + 1] - py;
    };
    auto chi2_of = [&](int i) {
        return err[2 * i] * (w[i] * err[2 * i]) + err[2 * i + 1] * (w[i] * err[2 * i + 1]);
    };
    for (int i = 0; i < n; i++) compute_error(i, est);

    int nBad = 0;
    double total_iters = 0, last_chi = 0, lambda = 0;
    std::vector<int> active;
    for (int it = 0; it < 4; it++) {
        est = se3_from_Rt(R0, t0); /* LocalBA.cpp:426-428: every round restarts from the input pose */
        active.clear();
        for (int i = 0; i < n; i++)
            if (level[i] == 0) active.push_back(i);

        auto active_errors = [&](const SE3& T) { for (int i : active) compute_error(i, T); };
        auto active_robust_chi2 = [&]() {
            double chi = 0;
            for (int i : active) {
                double c = chi2_of(i);
                if (robust[i]) { double rho[3]; huber(c, delta, rho); chi += rho[0]; }
                else chi += c;
            }
            return chi;
        };

        if (!active.empty()) {
            /* SparseOptimizer::optimize(10) + OptimizationAlgorithmLevenberg::solve */
            double ni = 2;
            bool ok = true;
            for (int iter = 0; iter < 10 && ok; iter++) {
                active_errors(est);
                double currentChi = active_robust_chi2();
                double tempChi = currentChi;
                double H[36] = {0}, b[6] = {0};
                for (int i : active) { /* buildSystem: linearizeOplus + constructQuadraticForm */
                    double pc[3], J[12];
                    se3_map(est, &Xw[3 * i], pc);
                    jac_pose(pc, fx, fy, J);
                    double r1 = 1.0;
                    if (robust[i]) { double rho[3]; huber(chi2_of(i), delta, rho); r1 = rho[1]; }
                    const double wo = w[i], ww = r1 * w[i];
                    for (int a = 0; a < 6; a++) {
                        b[a] -= r1 * ((J[a] * wo) * err[2 * i] + (J[6 + a] * wo) * err[2 * i + 1]);
                        for (int c = 0; c < 6; c++)
                            H[a * 6 + c] += (J[a] * ww) * J[c] + (J[6 + a] * ww) * J[6 + c];
                    }
                }
                if (iter == 0) { lambda = 1e-4; ni = 2; } /* setUserLambdaInit(0.0001), LocalBA.cpp:303 */
                double rho = 0;
                int qmax = 0;
                do {
                    SE3 backup = est; /* push() */
                    std::vector<double> A(H, H + 36), rhs(b, b + 6), x(6, 0.0);
                    for (int a = 0; a < 6; a++) A[a * 6 + a] += lambda;
                    bool ok2 = chol_solve(6, A, rhs, x);
                    if (!ok2) std::fill(x.begin(), x.end(), 0.0);
                    est = se3_mul(se3_exp(x.data()), est); /* VertexSE3Expmap::oplusImpl */
                    active_errors(est);
                    tempChi = active_robust_chi2();
                    if (!ok2) tempChi = std::numeric_limits<double>::max();
                    rho = currentChi - tempChi;
                    double scale = 0;
                    for (int a = 0; a < 6; a++) scale += x[a] * (lambda * x[a] + b[a]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && std::isfinite(tempChi)) {
                        double alpha = 1. - std::pow(2 * rho - 1, 3);
                        alpha = std::min(alpha, 2. / 3.);
                        double scaleFactor = std::max(1. / 3., alpha);
                        lambda *= scaleFactor;
                        ni = 2;
                        currentChi = tempChi;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                        est = backup; /* pop(): estimate restored, edge errors stay as computed */
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                total_iters += 1;
                last_chi = currentChi;
                if (qmax == 10 || rho == 0) ok = false; /* Terminate */
            }
        }

        nBad = 0;
        for (int i = 0; i < n; i++) { /* LocalBA.cpp:434-461 */
            if (outlier[i]) compute_error(i, est);
            const float chi2 = (float)chi2_of(i);
            if (chi2 > chi2Mono) { outlier[i] = 1; level[i] = 1; nBad++; }
            else { outlier[i] = 0; level[i] = 0; }
            if (it == 2) robust[i] = 0;
        }
        if (n < 10) break; /* optimizer.edges().size()<10, LocalBA.cpp:477 */
    }

    double R[9];
    quat_to_R(est.r, R);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Tcw_out[i * 4 + j] = (float)R[i * 3 + j];
        Tcw_out[i * 4 + 3] = (float)est.t[i];
    }
    Tcw_out[12] = Tcw_out[13] = Tcw_out[14] = 0.f;
    Tcw_out[15] = 1.f;
    if (stats) {
        stats[0] = total_iters;
        stats[1] = last_chi;
        stats[2] = lambda;
        stats[3] = nBad;
        stats[4] = est.t[0]; stats[5] = est.t[1]; stats[6] = est.t[2];
        stats[7] = est.r.w;
    }
    return nInitialCorrespondences - nBad;
}

/* Multi-keyframe local BA (extension, no reference counterpart): vertices = nkf SE3 poses (the first
 * nfixed held fixed) + npt points; edges = reprojection (Huber, delta^2 = 5.991); g2o-style LM
 * (tau = 1e-5, same accept/reject rule as above) on the Schur-reduced pose system. */
int orc_local_ba(const double K[4], int nkf, int nfixed, float* poses, int npt, float* pts,
                 const tb_ba_obs* obs, int nobs, int iters, double* stats) {
    if (!K || !poses || !pts || !obs || nkf < 1 || npt < 1 || nobs < 1 || nfixed < 0 || nfixed > nkf)
        return TB_EINVAL;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double delta = (double)std::sqrt(5.991f);
    const int nfree = nkf - nfixed, np = 6 * nfree;
    std::vector<SE3> T(nkf);
    for (int k = 0; k < nkf; k++) {
        double R[9], t[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) R[i * 3 + j] = poses[k * 16 + i * 4 + j];
            t[i] = poses[k * 16 + i * 4 + 3];
        }
        T[k] = se3_from_Rt(R, t);
    }
    std::vector<double> P(3 * (size_t)npt);
    for (int i = 0; i < 3 * npt; i++) P[i] = pts[i];
    for (int e = 0; e < nobs; e++)
        if (obs[e].kf < 0 || obs[e].kf >= nkf || obs[e].pt < 0 || obs[e].pt >= npt) return TB_EINVAL;

    std::vector<double> err(2 * (size_t)nobs);
    auto errors = [&](const std::vector<SE3>& TT, const std::vector<double>& PP) {
        double chi = 0;
        for (int e = 0; e < nobs; e++) {
            double pc[3];
            se3_map(TT[obs[e].kf], &PP[3 * obs[e].pt], pc);
            err[2 * e] = (double)obs[e].u - (pc[0] / pc[2] * fx + cx);
            err[2 * e + 1] = (double)obs[e].v - (pc[1] / pc[2] * fy + cy);
            const double wgt = obs[e].inv_sigma2;
            double c = err[2 * e] * (wgt * err[2 * e]) + err[2 * e + 1] * (wgt * err[2 * e + 1]);
            double rho[3];
            huber(c, delta, rho);
            chi += rho[0];
        }
        return chi;
    };

    double lambda = 0, ni = 2, chi0 = 0, chi_last = 0;
    int done = 0, trials = 0; /* trials: LM steps tried, rejected ones included (stats[4]) */
    bool ok = true;
    std::vector<double> Hpp((size_t)np * np), bp(np), Hll(9 * (size_t)npt), bl(3 * (size_t)npt),
        Hpl((size_t)nobs * 18);
    for (int iter = 0; iter < iters && ok; iter++) {
        double currentChi = errors(T, P);
        if (iter == 0) chi0 = currentChi;
        std::fill(Hpp.begin(), Hpp.end(), 0.0);
        std::fill(bp.begin(), bp.end(), 0.0);
        std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(bl.begin(), bl.end(), 0.0);
        std::fill(Hpl.begin(), Hpl.end(), 0.0);
        for (int e = 0; e < nobs; e++) {
            const int k = obs[e].kf, l = obs[e].pt;
            double pc[3], Jp[12], Jl[6], R[9];
            se3_map(T[k], &P[3 * l], pc);
            jac_pose(pc, fx, fy, Jp);
            quat_to_R(T[k].r, R);
            const double x = pc[0], y = pc[1], z = pc[2];
            const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
            for (int a = 0; a < 2; a++)
                for (int c = 0; c < 3; c++)
                    Jl[a * 3 + c] = -1. / z * (tmp[a * 3] * R[c] + tmp[a * 3 + 1] * R[3 + c] + tmp[a * 3 + 2] * R[6 + c]);
            const double wgt = obs[e].inv_sigma2;
            double c2 = err[2 * e] * (wgt * err[2 * e]) + err[2 * e + 1] * (wgt * err[2 * e + 1]);
            double rho[3];
            huber(c2, delta, rho);
            const double ww = rho[1] * wgt;
            for (int a = 0; a < 3; a++) {
                bl[3 * l + a] -= ww * (Jl[a] * err[2 * e] + Jl[3 + a] * err[2 * e + 1]);
                for (int c = 0; c < 3; c++) Hll[9 * (size_t)l + a * 3 + c] += ww * (Jl[a] * Jl[c] + Jl[3 + a] * Jl[3 + c]);
            }
            if (k >= nfixed) {
                const int o = 6 * (k - nfixed);
                for (int a = 0; a < 6; a++) {
                    bp[o + a] -= ww * (Jp[a] * err[2 * e] + Jp[6 + a] * err[2 * e + 1]);
                    for (int c = 0; c < 6; c++)
                        Hpp[(size_t)(o + a) * np + o + c] += ww * (Jp[a] * Jp[c] + Jp[6 + a] * Jp[6 + c]);
                    for (int c = 0; c < 3; c++)
                        Hpl[(size_t)e * 18 + a * 3 + c] = ww * (Jp[a] * Jl[c] + Jp[6 + a] * Jl[3 + c]);
                }
            }
        }
        if (iter == 0) {
            double maxDiag = 0;
            for (int a = 0; a < np; a++) maxDiag = std::max(maxDiag, std::fabs(Hpp[(size_t)a * np + a]));
            for (int l = 0; l < npt; l++)
                for (int a = 0; a < 3; a++) maxDiag = std::max(maxDiag, std::fabs(Hll[9 * (size_t)l + a * 4]));
            lambda = 1e-5 * maxDiag;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            /* Schur complement with lambda on both diagonals */
            std::vector<double> S(Hpp), rhs(bp), Hinv(9 * (size_t)npt);
            for (int a = 0; a < np; a++) S[(size_t)a * np + a] += lambda;
            bool ok2 = true;
            for (int l = 0; l < npt; l++) {
                double A[9];
                for (int a = 0; a < 9; a++) A[a] = Hll[9 * (size_t)l + a];
                A[0] += lambda; A[4] += lambda; A[8] += lambda;
                const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
                                   A[2] * (A[3] * A[7] - A[4] * A[6]);
                if (!(std::fabs(det) > 0)) { ok2 = false; break; }
                const double id = 1.0 / det;
                double* I = &Hinv[9 * (size_t)l];
                I[0] = (A[4] * A[8] - A[5] * A[7]) * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
                I[3] = (A[5] * A[6] - A[3] * A[8]) * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
                I[6] = (A[3] * A[7] - A[4] * A[6]) * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
            }
            std::vector<double> xp(np, 0.0), xl(3 * (size_t)npt, 0.0);
            if (ok2) {
                /* per point: gather its observations from free keyframes */
                std::vector<std::vector<int>> byPt(npt);
                for (int e = 0; e < nobs; e++)
                    if (obs[e].kf >= nfixed) byPt[obs[e].pt].push_back(e);
                for (int l = 0; l < npt; l++) {
                    const double* I = &Hinv[9 * (size_t)l];
                    for (int e1 : byPt[l]) {
                        const int o1 = 6 * (obs[e1].kf - nfixed);
                        double Y[18]; /* Hpl * Hll^-1 */
                        for (int a = 0; a < 6; a++)
                            for (int c = 0; c < 3; c++)
                                Y[a * 3 + c] = Hpl[(size_t)e1 * 18 + a * 3] * I[c] + Hpl[(size_t)e1 * 18 + a * 3 + 1] * I[3 + c] +
                                               Hpl[(size_t)e1 * 18 + a * 3 + 2] * I[6 + c];
                        for (int a = 0; a < 6; a++)
                            rhs[o1 + a] -= Y[a * 3] * bl[3 * l] + Y[a * 3 + 1] * bl[3 * l + 1] + Y[a * 3 + 2] * bl[3 * l + 2];
                        for (int e2 : byPt[l]) {
                            const int o2 = 6 * (obs[e2].kf - nfixed);
                            for (int a = 0; a < 6; a++)
                                for (int c = 0; c < 6; c++)
                                    S[(size_t)(o1 + a) * np + o2 + c] -=
                                        Y[a * 3] * Hpl[(size_t)e2 * 18 + c * 3] + Y[a * 3 + 1] * Hpl[(size_t)e2 * 18 + c * 3 + 1] +
                                        Y[a * 3 + 2] * Hpl[(size_t)e2 * 18 + c * 3 + 2];
                        }
                    }
                }
                if (np > 0) ok2 = chol_solve(np, S, rhs, xp);
                if (ok2) {
                    std::vector<double> r(bl);
                    for (int e = 0; e < nobs; e++) {
                        if (obs[e].kf < nfixed) continue;
                        const int o = 6 * (obs[e].kf - nfixed), l = obs[e].pt;
                        for (int c = 0; c < 3; c++)
                            for (int a = 0; a < 6; a++) r[3 * l + c] -= Hpl[(size_t)e * 18 + a * 3 + c] * xp[o + a];
                    }
                    for (int l = 0; l < npt; l++) {
                        const double* I = &Hinv[9 * (size_t)l];
                        for (int a = 0; a < 3; a++)
                            xl[3 * l + a] = I[a * 3] * r[3 * l] + I[a * 3 + 1] * r[3 * l + 1] + I[a * 3 + 2] * r[3 * l + 2];
                    }
                }
            }
            if (!ok2) { std::fill(xp.begin(), xp.end(), 0.0); std::fill(xl.begin(), xl.end(), 0.0); }
            std::vector<SE3> Tn(T);
            std::vector<double> Pn(P);
            for (int k = nfixed; k < nkf; k++) Tn[k] = se3_mul(se3_exp(&xp[6 * (k - nfixed)]), T[k]);
            for (int i = 0; i < 3 * npt; i++) Pn[i] += xl[i];
            double tempChi = errors(Tn, Pn);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int a = 0; a < np; a++) scale += xp[a] * (lambda * xp[a] + bp[a]);
            for (int a = 0; a < 3 * npt; a++) scale += xl[a] * (lambda * xl[a] + bl[a]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                T = Tn;
                P = Pn;
            } else {
                lambda *= ni;
                ni *= 2;
                errors(T, P);
            }
            qmax++;
            trials++;
        } while (rho < 0 && qmax < 10);
        done++;
        chi_last = currentChi;
        if (qmax == 10 || rho == 0) ok = false;
    }
    for (int k = 0; k < nkf; k++) {
        double R[9];
        quat_to_R(T[k].r, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) poses[k * 16 + i * 4 + j] = (float)R[i * 3 + j];
            poses[k * 16 + i * 4 + 3] = (float)T[k].t[i];
        }
        poses[k * 16 + 12] = poses[k * 16 + 13] = poses[k * 16 + 14] = 0.f;
        poses[k * 16 + 15] = 1.f;
    }
    for (int i = 0; i < 3 * npt; i++) pts[i] = (float)P[i];
    if (stats) {
        stats[0] = done; stats[1] = chi0; stats[2] = chi_last; stats[3] = lambda;
        stats[4] = trials;
        stats[5] = stats[6] = stats[7] = 0;
    }
    return done;
}

}  // extern "C"
