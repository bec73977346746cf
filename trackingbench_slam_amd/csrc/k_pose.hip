/* a15 -- motion-only pose optimisation, LocalBA::PoseOptimization (src/mapping/LocalBA.cpp:291-490).
 *
 * g2o's machinery (OptimizationAlgorithmLevenberg::solve, SparseOptimizer::optimize,
 * BaseUnaryEdge::constructQuadraticForm, RobustKernelHuber, EdgeSE3ProjectXYZOnlyPose,
 * VertexSE3Expmap / SE3Quat) restated for one workgroup per frame: edges are spread over the lanes,
 * every sum over edges (robust chi2, the 21 + 6 entries of J'WJ and J'We) is a fixed-shape FP64 tree
 * reduction (per-lane strided partials -> wave shuffle tree -> 4 wave partials), so results are
 * reproducible run to run; the 6x6 solve, the exp-map update and the LM accept/reject logic run on
 * lane 0.  Four rounds x <= 10 LM iterations x <= 10 trials, Huber delta = sqrt(5.991), chi2 gate
 * 5.991, pose reset to the input every round, robust kernel dropped in the last round, edge errors left
 * "as last evaluated" exactly like g2o leaves them (a rejected trial's errors survive the pop()).
 *
 * Bound: latency / FP64 VALU (~150 flop per edge per evaluation, ~12 MFLOP per frame); not HBM.
 */
#include "tb_internal.h"
#include "tb_device.h"

#define PO_T 256

#include "tb_se3.h"

struct PoShared {
    PoSE3 est, backup;
    double H[36], b[6], x[6];
    double lambda, ni, currentChi, tempChi, rho;
    int qmax, again, ok;
};

__global__ void __launch_bounds__(PO_T)
k_pose(int nproblems, double fx, double fy, double cx, double cy, const float* __restrict__ Tcw_in,
       const tb_obs* __restrict__ obsAll, const int32_t* __restrict__ counts, int obs_pitch, uint8_t* __restrict__ outlierAll,
       float* __restrict__ Tcw_out, int32_t* __restrict__ n_inliers, double* __restrict__ stats, double* __restrict__ errAll) {
    __shared__ PoShared S;
    __shared__ double red[4 * 28];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[p], obs_pitch);
    const tb_obs* obs = obsAll + (size_t)p * obs_pitch;
    uint8_t* outlier = outlierAll + (size_t)p * obs_pitch;
    double* err = errAll + (size_t)p * obs_pitch * 3; /* e0, e1, level */
    const float* Tin = Tcw_in + 16 * p;
    float* Tout = Tcw_out + 16 * p;
    if (tid < 16) Tout[tid] = Tin[tid];
    if (stats && tid < 8) stats[8 * p + tid] = 0;
    if (n < 3) { /* LocalBA.cpp:401 */
        if (tid == 0) n_inliers[p] = 0;
        return;
    }
    double R0[9], t0[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R0[i * 3 + j] = (double)Tin[i * 4 + j];
        t0[i] = (double)Tin[i * 4 + 3];
    }
    const PoSE3 est0 = po_from_Rt(R0, t0);
    const double delta = (double)sqrtf(5.991f);
    const float chi2Mono = 5.991f;

    auto edge_error = [&](int i, const PoSE3& T) {
        const tb_obs o = obs[i];
        const double X[3] = {(double)o.X, (double)o.Y, (double)o.Z};
        double pc[3];
        po_map(T, X, pc);
        err[3 * i] = (double)o.u - (pc[0] / pc[2] * fx + cx);
        err[3 * i + 1] = (double)o.v - (pc[1] / pc[2] * fy + cy);
    };
    auto edge_chi2 = [&](int i) {
        const double w = (double)obs[i].inv_sigma2;
        return err[3 * i] * (w * err[3 * i]) + err[3 * i + 1] * (w * err[3 * i + 1]);
    };
    auto robust_chi = [&](double c, bool robust) {
        if (!robust) return c;
        const double dsqr = delta * delta;
        if (c <= dsqr) return c;
        return 2 * sqrt(c) * delta - dsqr;
    };

    for (int i = tid; i < n; i += PO_T) {
        edge_error(i, est0);
        err[3 * i + 2] = 0.0; /* level 0 */
    }
    int nBad = 0;
    double total_iters = 0;
    for (int it = 0; it < 4; it++) {
        const bool robust = it < 3; /* setRobustKernel(nullptr) after round 2, LocalBA.cpp:459 */
        __syncthreads();
        if (tid == 0) { S.est = est0; S.ok = 1; S.ni = 2; }
        double na[1] = {0};
        for (int i = tid; i < n; i += PO_T) na[0] += (err[3 * i + 2] == 0.0) ? 1.0 : 0.0;
        po_block_sum<1>(na, red);
        const bool have_active = na[0] > 0;
        __syncthreads();
        for (int iter = 0; iter < 10 && have_active; iter++) {
            if (!S.ok) break;
            /* computeActiveErrors + activeRobustChi2 + buildSystem in one sweep */
            const PoSE3 est = S.est;
            double acc[28];
            for (int k = 0; k < 28; k++) acc[k] = 0;
            for (int i = tid; i < n; i += PO_T) {
                if (err[3 * i + 2] != 0.0) continue;
                const tb_obs o = obs[i];
                const double X[3] = {(double)o.X, (double)o.Y, (double)o.Z};
                double pc[3];
                po_map(est, X, pc);
                const double e0 = (double)o.u - (pc[0] / pc[2] * fx + cx);
                const double e1 = (double)o.v - (pc[1] / pc[2] * fy + cy);
                err[3 * i] = e0; err[3 * i + 1] = e1;
                const double w = (double)o.inv_sigma2;
                const double c = e0 * (w * e0) + e1 * (w * e1);
                double r1 = 1.0;
                if (robust && c > delta * delta) r1 = delta / sqrt(c);
                acc[27] += robust_chi(c, robust);
                const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
                double J[12];
                J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
                J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
                J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy;
                J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
                const double ww = r1 * w;
                int k = 0;
                for (int a = 0; a < 6; a++) {
                    acc[21 + a] -= r1 * ((J[a] * w) * e0 + (J[6 + a] * w) * e1);
                    for (int cc = a; cc < 6; cc++) acc[k++] += (J[a] * ww) * J[cc] + (J[6 + a] * ww) * J[6 + cc];
                }
            }
            po_block_sum<28>(acc, red);
            if (tid == 0) {
                int k = 0;
                for (int a = 0; a < 6; a++)
                    for (int cc = a; cc < 6; cc++) { S.H[a * 6 + cc] = acc[k]; S.H[cc * 6 + a] = acc[k]; k++; }
                for (int a = 0; a < 6; a++) S.b[a] = acc[21 + a];
                S.currentChi = acc[27];
                if (iter == 0) { S.lambda = 1e-4; S.ni = 2; } /* setUserLambdaInit(0.0001), LocalBA.cpp:303 */
                S.rho = 0; S.qmax = 0; S.again = 1;
            }
            __syncthreads();
            while (S.again) {
                if (tid == 0) {
                    S.backup = S.est;
                    double x[6] = {0, 0, 0, 0, 0, 0};
                    const bool ok2 = po_chol6(S.H, S.lambda, S.b, x);
                    if (!ok2) for (int a = 0; a < 6; a++) x[a] = 0;
                    for (int a = 0; a < 6; a++) S.x[a] = x[a];
                    S.est = po_exp_mul(x, S.est);
                    S.tempChi = ok2 ? 0.0 : -1.0;
                }
                __syncthreads();
                const PoSE3 trial = S.est;
                double chi[1] = {0};
                for (int i = tid; i < n; i += PO_T) {
                    if (err[3 * i + 2] != 0.0) continue;
                    edge_error(i, trial);
                    chi[0] += robust_chi(edge_chi2(i), robust);
                }
                po_block_sum<1>(chi, red);
                if (tid == 0) {
                    double tempChi = (S.tempChi < 0) ? 1.7976931348623157e308 : chi[0];
                    double rho = S.currentChi - tempChi;
                    double scale = 0;
                    for (int a = 0; a < 6; a++) scale += S.x[a] * (S.lambda * S.x[a] + S.b[a]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        double alpha = 1. - pow(2 * rho - 1, 3);
                        alpha = fmin(alpha, 2. / 3.);
                        S.lambda *= fmax(1. / 3., alpha);
                        S.ni = 2;
                        S.currentChi = tempChi;
                    } else {
                        S.lambda *= S.ni;
                        S.ni *= 2;
                        S.est = S.backup;
                    }
                    S.qmax++;
                    S.rho = rho;
                    S.again = (rho < 0 && S.qmax < 10) ? 1 : 0;
                    if (!S.again && (S.qmax == 10 || rho == 0)) S.ok = 0;
                }
                __syncthreads();
            }
            total_iters += 1;
            __syncthreads();
        }
        __syncthreads();
        /* classify, LocalBA.cpp:434-461 */
        const PoSE3 fin = S.est;
        double bad[1] = {0};
        for (int i = tid; i < n; i += PO_T) {
            if (outlier[i]) edge_error(i, fin);
            const float chi2 = (float)edge_chi2(i);
            if (chi2 > chi2Mono) { outlier[i] = 1; err[3 * i + 2] = 1.0; bad[0] += 1.0; }
            else { outlier[i] = 0; err[3 * i + 2] = 0.0; }
        }
        po_block_sum<1>(bad, red);
        nBad = (int)bad[0];
        if (n < 10) break; /* optimizer.edges().size() < 10, LocalBA.cpp:477 */
    }
    __syncthreads();
    if (tid == 0) {
        double R[9];
        po_to_R(S.est, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) Tout[i * 4 + j] = (float)R[i * 3 + j];
            Tout[i * 4 + 3] = (float)(i == 0 ? S.est.tx : (i == 1 ? S.est.ty : S.est.tz));
        }
        Tout[12] = Tout[13] = Tout[14] = 0.f;
        Tout[15] = 1.f;
        n_inliers[p] = n - nBad;
        if (stats) {
            double* st = stats + 8 * p;
            st[0] = total_iters; st[1] = S.currentChi; st[2] = S.lambda; st[3] = nBad;
            st[4] = S.est.tx; st[5] = S.est.ty; st[6] = S.est.tz; st[7] = S.est.qw;
        }
    }
}

int tbk_pose_batch(tb_ctx* ctx, int nproblems, const double K[4], const float* Tcw_in, const tb_obs* obs,
                   const int32_t* counts, int obs_pitch, uint8_t* outlier, float* Tcw_out, int32_t* n_inliers,
                   double* stats, double* d_err) {
    if (nproblems <= 0) return TB_OK;
    tb_prof_begin(ctx, "k_pose");
    hipLaunchKernelGGL(k_pose, dim3(nproblems), dim3(PO_T), 0, ctx->stream, nproblems, K[0], K[1], K[2], K[3], Tcw_in, obs,
                       counts, obs_pitch, outlier, Tcw_out, n_inliers, stats, d_err);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
