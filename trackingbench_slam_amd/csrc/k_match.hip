/* a12 / a13 / a14 -- Hamming matchers.
 *
 * a12 Matcher::searchByBF (src/matchers/matcher.cpp:168-228) = cv::BFMatcher(NORM_HAMMING,
 *     crossCheck=true).match + "d < fmin(ratio*d_min, minTh)" filter.  OpenCV 3.3 cross-check semantics
 *     (batchDistance, restated): every TRAIN row takes its nearest query (first index on ties); a query
 *     keeps the train with the smallest such distance (first train on ties).  Both "first on ties"
 *     rules are an integer minimum over the packed word (distance << 32 | index), so the whole thing is
 *     two rounds of 64-bit atomicMin -- order independent, bit exact.
 * a13 Matcher::DescriptorDistance (:793-808): 256-bit Hamming = 4 x (xor64 + popcount64).
 * a14 Matcher::searchByViolence (:299-395): per F1 key, best / second-best over the 120x36 lookup grid
 *     window of F2 (Frame.cpp:202-255), traversal order (ix, iy, insertion) preserved per thread.
 *
 * Bound: integer VALU (v_bcnt accumulate), not HBM: a 2000 x 2000 pair is 4e6 popcount-256 on 128 KB
 * of descriptors that sit in LDS / L2.
 */
#include "tb_internal.h"
#include "tb_device.h"

#define BF_T 256
#define BF_QC 512 /* queries staged per block */

struct Desc256 { unsigned long long w[4]; };

__device__ __forceinline__ int bf_dist(const Desc256& a, const unsigned long long* __restrict__ q) {
    return __popcll(a.w[0] ^ q[0]) + __popcll(a.w[1] ^ q[1]) + __popcll(a.w[2] ^ q[2]) + __popcll(a.w[3] ^ q[3]);
}

/* For every "row" descriptor (one per thread) the nearest "col" descriptor of a staged chunk:
 * rbest[pair][row] = min(dist << 32 | col). grid (row tiles, col chunks, pairs). */
__global__ void __launch_bounds__(BF_T)
k_bf_nn(const uint8_t* __restrict__ rows, const int32_t* __restrict__ rowCounts, const uint8_t* __restrict__ cols,
        const int32_t* __restrict__ colCounts, size_t set_pitch, int max_n, unsigned long long* __restrict__ rbest) {
    __shared__ __attribute__((aligned(16))) unsigned long long q[BF_QC * 4];
    const int p = blockIdx.z;
    const int nr = min(rowCounts[p], max_n), nc = min(colCounts[p], max_n);
    const int c0 = blockIdx.y * BF_QC;
    if (c0 >= nc || (int)(blockIdx.x * BF_T) >= nr) return;
    const int cn = min(BF_QC, nc - c0);
    const unsigned long long* csrc = reinterpret_cast<const unsigned long long*>(cols + (size_t)p * set_pitch) + (size_t)c0 * 4;
    for (int i = threadIdx.x; i < cn * 4; i += BF_T) q[i] = csrc[i];
    __syncthreads();
    const int r = blockIdx.x * BF_T + threadIdx.x;
    if (r >= nr) return;
    const unsigned long long* rsrc = reinterpret_cast<const unsigned long long*>(rows + (size_t)p * set_pitch) + (size_t)r * 4;
    Desc256 d;
    d.w[0] = rsrc[0]; d.w[1] = rsrc[1]; d.w[2] = rsrc[2]; d.w[3] = rsrc[3];
    int best = 0x7fffffff, bi = 0;
    for (int c = 0; c < cn; c++) {
        const int dist = bf_dist(d, q + 4 * c);
        if (dist < best) { best = dist; bi = c; }
    }
    atomicMin(&rbest[(size_t)p * max_n + r], ((unsigned long long)best << 32) | (unsigned)(c0 + bi));
}

/* cross-check: train t -> its nearest query q; qbest[q] = min(dist << 32 | t) */
__global__ void __launch_bounds__(BF_T)
k_bf_cross(const int32_t* __restrict__ trainCounts, int max_n, const unsigned long long* __restrict__ tbest,
           unsigned long long* __restrict__ qbest) {
    const int p = blockIdx.y, t = blockIdx.x * BF_T + threadIdx.x;
    if (t >= min(trainCounts[p], max_n)) return;
    const unsigned long long v = tbest[(size_t)p * max_n + t];
    if (v == ~0ull) return;
    const unsigned q = (unsigned)v;
    atomicMin(&qbest[(size_t)p * max_n + q], (v & 0xffffffff00000000ull) | (unsigned)t);
}

/* one block per pair: d_min, filter, compaction in query order (searchByBF :209-218) */
__global__ void __launch_bounds__(BF_T)
k_bf_finalize(const int32_t* __restrict__ queryCounts, int max_n, const unsigned long long* __restrict__ qbest, int filter,
              float ratio, float min_th, tb_match* __restrict__ out, int cap, int32_t* __restrict__ outCounts) {
    __shared__ int flags[BF_T];
    __shared__ int tmp[8];
    __shared__ int s_min, running;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int nq = min(queryCounts[p], max_n);
    const unsigned long long* qb = qbest + (size_t)p * max_n;
    if (tid == 0) { s_min = 0x7fffffff; running = 0; }
    __syncthreads();
    int mn = 0x7fffffff;
    for (int qi = tid; qi < nq; qi += BF_T) {
        const unsigned long long v = qb[qi];
        if (v != ~0ull) mn = min(mn, (int)(v >> 32));
    }
    atomicMin(&s_min, mn);
    __syncthreads();
    float lim = 3.0e38f;
    if (filter) lim = fminf(TB_FMUL(ratio, (float)s_min), min_th);
    tb_match* o = out + (size_t)p * cap;
    for (int base = 0; base < nq; base += BF_T) {
        const int qi = base + tid;
        unsigned long long v = ~0ull;
        if (qi < nq) v = qb[qi];
        const float dist = (float)(int)(v >> 32);
        const int f = (v != ~0ull && (!filter || dist < lim)) ? 1 : 0;
        flags[tid] = f;
        __syncthreads();
        const int total = tb_block_excl_scan(flags, BF_T, tmp);
        if (f) {
            const int slot = running + flags[tid];
            if (slot < cap) {
                tb_match m;
                m.queryIdx = qi; m.trainIdx = (int)(unsigned)v; m.imgIdx = 0; m.distance = dist;
                o[slot] = m;
            }
        }
        __syncthreads();
        if (tid == 0) running += total;
        __syncthreads();
    }
    if (tid == 0) outCounts[p] = running; /* may exceed cap: the host reports TB_ECAPACITY */
}

int tbk_bf_batch(tb_ctx* ctx, int npairs, const uint8_t* d1, const int32_t* c1, const uint8_t* d2, const int32_t* c2,
                 size_t set_pitch, int max_n, int crosscheck, int filter, float ratio, float min_th, tb_match* out, int cap,
                 int32_t* out_counts, unsigned long long* d_tbest, unsigned long long* d_qbest) {
    if (npairs <= 0 || max_n <= 0) return TB_OK;
    const size_t bytes = (size_t)npairs * max_n * sizeof(unsigned long long);
    TB_HIP(ctx, hipMemsetAsync(d_qbest, 0xff, bytes, ctx->stream));
    dim3 grid((max_n + BF_T - 1) / BF_T, (max_n + BF_QC - 1) / BF_QC, npairs);
    if (crosscheck) {
        TB_HIP(ctx, hipMemsetAsync(d_tbest, 0xff, bytes, ctx->stream));
        /* rows = train (d2), cols = query (d1) */
        tb_prof_begin(ctx, "k_bf_nn");
        hipLaunchKernelGGL(k_bf_nn, grid, dim3(BF_T), 0, ctx->stream, d2, c2, d1, c1, set_pitch, max_n, d_tbest);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
        tb_prof_begin(ctx, "k_bf_cross");
        hipLaunchKernelGGL(k_bf_cross, dim3((max_n + BF_T - 1) / BF_T, npairs), dim3(BF_T), 0, ctx->stream, c2, max_n, d_tbest,
                           d_qbest);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
    } else {
        /* rows = query, cols = train: qbest[q] = (dist, nearest train) directly */
        tb_prof_begin(ctx, "k_bf_nn");
        hipLaunchKernelGGL(k_bf_nn, grid, dim3(BF_T), 0, ctx->stream, d1, c1, d2, c2, set_pitch, max_n, d_qbest);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
    }
    tb_prof_begin(ctx, "k_bf_finalize");
    hipLaunchKernelGGL(k_bf_finalize, dim3(npairs), dim3(BF_T), 0, ctx->stream, c1, max_n, d_qbest, filter, ratio, min_th, out,
                       cap, out_counts);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* a14: window search. One thread per F1 key; F2's 120x36 grid arrives as CSR (cellStart, cellItems). */
__global__ void __launch_bounds__(256)
k_window(const tb_keypoint* __restrict__ k1, const uint8_t* __restrict__ d1, int n1, const tb_keypoint* __restrict__ k2,
         const uint8_t* __restrict__ d2, const int32_t* __restrict__ cellStart, const int32_t* __restrict__ cellItems,
         float widthInv, float heightInv, int min_level, int max_level, float r, int32_t* __restrict__ best) {
    const int GRID_ROWS = 36, GRID_COLS = 120;
    const int i1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i1 >= n1) return;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx = -1, ncand = 0;
    const float x = k1[i1].x, y = k1[i1].y;
    /* Frame::GetFeaturesInArea, Frame.cpp:202-255 */
    const int nMinCellX = max(0, (int)floorf(TB_FMUL(TB_FSUB(x, r), widthInv)));
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(TB_FMUL(TB_FADD(x, r), widthInv)));
    const int nMinCellY = max(0, (int)floorf(TB_FMUL(TB_FSUB(y, r), heightInv)));
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(TB_FMUL(TB_FADD(y, r), heightInv)));
    if (nMinCellX < GRID_COLS && nMaxCellX >= 0 && nMinCellY < GRID_ROWS && nMaxCellY >= 0) {
        const bool bCheckLevels = (min_level > 0) || (max_level >= 0);
        const unsigned long long* a = reinterpret_cast<const unsigned long long*>(d1) + (size_t)i1 * 4;
        Desc256 da;
        da.w[0] = a[0]; da.w[1] = a[1]; da.w[2] = a[2]; da.w[3] = a[3];
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c = ix * GRID_ROWS + iy;
                for (int s = cellStart[c]; s < cellStart[c + 1]; s++) {
                    const int j = cellItems[s];
                    const tb_keypoint kp = k2[j];
                    if (bCheckLevels) {
                        if (kp.octave < min_level) continue;
                        if (max_level >= 0 && kp.octave > max_level) continue;
                    }
                    if (!(fabsf(TB_FSUB(kp.x, x)) < r && fabsf(TB_FSUB(kp.y, y)) < r)) continue;
                    ncand++;
                    const int dist = bf_dist(da, reinterpret_cast<const unsigned long long*>(d2) + (size_t)j * 4);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
            }
    }
    best[4 * i1] = bestDist;
    best[4 * i1 + 1] = bestDist2;
    best[4 * i1 + 2] = bestIdx;
    best[4 * i1 + 3] = ncand;
}

int tbk_window_match(tb_ctx* ctx, const tb_keypoint* d_k1, const uint8_t* d_d1, int n1, const tb_keypoint* d_k2,
                     const uint8_t* d_d2, int n2, const int32_t* d_cellStart, const int32_t* d_cellItems, float widthInv,
                     float heightInv, int min_level, int max_level, float r, int32_t* d_best) {
    if (n1 <= 0) return TB_OK;
    tb_prof_begin(ctx, "k_window");
    hipLaunchKernelGGL(k_window, dim3((n1 + 255) / 256), dim3(256), 0, ctx->stream, d_k1, d_d1, n1, d_k2, d_d2, d_cellStart,
                       d_cellItems, widthInv, heightInv, min_level, max_level, r, d_best);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ------------------------------------------------------------------------------------------------
 * SURVEY 8(f) row 1 -- Matcher::searchByProjection, both overloads (matcher.cpp:406-617).
 * Two steps on the device: (1) one thread per map point projects it into F1 and derives its search window
 * (k_project_frame: matcher.cpp:431-458; k_project_map: Frame::IsInFrustum, Frame.cpp:370-412, and
 * matcher.cpp:558-567), (2) one thread per query walks F1's 120x36 lookup grid in the reference's order
 * (ix, iy, insertion) and keeps best / second best with their levels (k_window_q). Float arithmetic: one rounding
 * per reference operation, fixed-size Eigen 3.3 reduction order c0 + (c1 + c2), no FMA contraction. */
#pragma clang fp contract(off)
struct ProjPose { float T[16]; };
struct ProjQuery { float u, v, r; int32_t minL, maxL; };  /* r < 0: no search for this map point */

__device__ __forceinline__ void pj_se3_map(const ProjPose& P, const float* X, float* Pc) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float c0 = P.T[4 * i] * X[0], c1 = P.T[4 * i + 1] * X[1], c2 = P.T[4 * i + 2] * X[2];
        Pc[i] = (c0 + (c1 + c2)) + P.T[4 * i + 3];
    }
}
__device__ __forceinline__ void pj_world2cam(const tb_camera& cam, const float* Pc, float* px) {
    const float x = Pc[0] / Pc[2], y = Pc[1] / Pc[2];
    if (!cam.has_distortion) {
        px[0] = cam.fx * x + cam.cx;
        px[1] = cam.fy * y + cam.cy;
    } else {
        const float r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const float a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const float cdist = 1 + cam.d[0] * r2 + cam.d[1] * r4 + cam.d[4] * r6;
        const float xd = x * cdist + cam.d[2] * a1 + cam.d[3] * a2;
        const float yd = y * cdist + cam.d[2] * a3 + cam.d[3] * a1;
        px[0] = xd * cam.fx + cam.cx;
        px[1] = yd * cam.fy + cam.cy;
    }
}
__device__ __forceinline__ bool pj_in_frame(const tb_camera& cam, const float* px) {
    if (!(fabsf(px[0]) < 2147483648.f) || !(fabsf(px[1]) < 2147483648.f)) return false; /* x86 cast -> INT_MIN */
    const int u = (int)px[0], v = (int)px[1];
    return u >= 0 && u < (int)((float)cam.width * 1.f) && v >= 0 && v < (int)((float)cam.height * 1.f);
}

__global__ void __launch_bounds__(256)
k_project_frame(ProjPose P, tb_camera cam, const tb_keypoint* __restrict__ k2, const tb_mappoint* __restrict__ mp2, int n2,
                const float* __restrict__ sf, int nlevels, float nratio, ProjQuery* __restrict__ q, int* __restrict__ bad_octave) {
    const int i2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i2 >= n2) return;
    ProjQuery o = {0.f, 0.f, -1.f, 0, 0};
    const tb_mappoint mp = mp2[i2];
    if (!mp.bad) {
        float Pc[3], uv[2];
        pj_se3_map(P, mp.pos, Pc);
        const float invzc = 1.0f / Pc[2];
        if (!(invzc < 0)) {
            pj_world2cam(cam, Pc, uv);
            if (pj_in_frame(cam, uv)) {
                const int oct = k2[i2].octave;
                if (oct < 0 || oct >= nlevels) *bad_octave = 1; /* benign race: every writer stores 1 */
                else { o.u = uv[0]; o.v = uv[1]; o.r = nratio * sf[oct]; o.minL = oct - 1; o.maxL = oct + 1; }
            }
        }
    }
    q[i2] = o;
}

__global__ void __launch_bounds__(256)
k_project_map(ProjPose P, tb_camera cam, const tb_mappoint* __restrict__ mps, int nmp, float sf0, float nratio,
              ProjQuery* __restrict__ q) {
    const int im = blockIdx.x * blockDim.x + threadIdx.x;
    if (im >= nmp) return;
    ProjQuery o = {0.f, 0.f, -1.f, 0, 0};
    const tb_mappoint mp = mps[im];
    if (!mp.bad) {
        float Pc[3], uv[2], Ow[3];
#pragma unroll
        for (int i = 0; i < 3; i++) { /* Frame::SetPose: mOw = -Rcw^T tcw */
            const float c0 = -P.T[i] * P.T[3], c1 = -P.T[4 + i] * P.T[7], c2 = -P.T[8 + i] * P.T[11];
            Ow[i] = c0 + (c1 + c2);
        }
        pj_se3_map(P, mp.pos, Pc);
        if (!(Pc[2] < 0.0f)) {
            pj_world2cam(cam, Pc, uv);
            if (pj_in_frame(cam, uv)) {
                const float PO[3] = {mp.pos[0] - Ow[0], mp.pos[1] - Ow[1], mp.pos[2] - Ow[2]};
                const float dist3 = sqrtf(PO[0] * PO[0] + (PO[1] * PO[1] + PO[2] * PO[2]));
                if (!(dist3 < mp.min_dist || dist3 > mp.max_dist)) {
                    const float viewCos = (PO[0] * mp.normal[0] + (PO[1] * mp.normal[1] + PO[2] * mp.normal[2])) / dist3;
                    if (!(viewCos < 0.5f)) {
                        float r = 4.f;
                        if ((double)viewCos > 0.998) r = 2.5f;
                        if ((double)nratio != 1.0) r *= nratio;
                        o.u = uv[0]; o.v = uv[1]; o.r = r * sf0; o.minL = -1; o.maxL = 0;
                    }
                }
            }
        }
    }
    q[im] = o;
}

/* best[6 q]: bestDist, bestDist2, bestIdx, bestLevel, bestLevel2, candidates in the window */
__global__ void __launch_bounds__(256)
k_window_q(const ProjQuery* __restrict__ q, const uint8_t* __restrict__ qd, int nq, const tb_keypoint* __restrict__ k1,
           const uint8_t* __restrict__ d1, const uint8_t* __restrict__ taken1, const int32_t* __restrict__ cellStart,
           const int32_t* __restrict__ cellItems, float widthInv, float heightInv, int32_t* __restrict__ best) {
    const int GRID_ROWS = 36, GRID_COLS = 120;
    const int iq = blockIdx.x * blockDim.x + threadIdx.x;
    if (iq >= nq) return;
    int bestDist = 256, bestDist2 = 256, bestIdx = -1, bestLevel = -1, bestLevel2 = -1, ncand = 0;
    const ProjQuery w = q[iq];
    if (w.r >= 0.f) {
        const float x = w.u, y = w.v, r = w.r;
        const int nMinCellX = max(0, (int)floorf((x - r) * widthInv));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x + r) * widthInv));
        const int nMinCellY = max(0, (int)floorf((y - r) * heightInv));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y + r) * heightInv));
        if (nMinCellX < GRID_COLS && nMaxCellX >= 0 && nMinCellY < GRID_ROWS && nMaxCellY >= 0) {
            const bool bCheckLevels = (w.minL > 0) || (w.maxL >= 0);
            const unsigned long long* a = reinterpret_cast<const unsigned long long*>(qd) + (size_t)iq * 4;
            Desc256 da;
            da.w[0] = a[0]; da.w[1] = a[1]; da.w[2] = a[2]; da.w[3] = a[3];
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const int c = ix * GRID_ROWS + iy;
                    for (int s = cellStart[c]; s < cellStart[c + 1]; s++) {
                        const int j = cellItems[s];
                        const tb_keypoint kp = k1[j];
                        if (bCheckLevels) {
                            if (kp.octave < w.minL) continue;
                            if (w.maxL >= 0 && kp.octave > w.maxL) continue;
                        }
                        if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) continue;
                        ncand++;
                        if (taken1[j]) continue;
                        const int dist = bf_dist(da, reinterpret_cast<const unsigned long long*>(d1) + (size_t)j * 4);
                        if (dist < bestDist) {
                            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kp.octave; bestIdx = j;
                        } else if (dist < bestDist2) {
                            bestLevel2 = kp.octave; bestDist2 = dist;
                        }
                    }
                }
        }
    }
    int32_t* o = best + (size_t)iq * 6;
    o[0] = bestDist; o[1] = bestDist2; o[2] = bestIdx; o[3] = bestLevel; o[4] = bestLevel2; o[5] = ncand;
}

int tbk_projection_search(tb_ctx* ctx, int map_overload, const float Tcw[16], const tb_camera* cam, const tb_keypoint* d_k2,
                          const tb_mappoint* d_mp, const uint8_t* d_mpdesc, int nq, const float* d_sf, int nlevels, float sf0,
                          float nratio, const tb_keypoint* d_k1, const uint8_t* d_d1, const uint8_t* d_taken1,
                          const int32_t* d_cellStart, const int32_t* d_cellItems, float widthInv, float heightInv,
                          void* d_queries, int32_t* d_best, int* d_flag) {
    if (nq <= 0) return TB_OK;
    ProjPose P;
    for (int i = 0; i < 16; i++) P.T[i] = Tcw[i];
    ProjQuery* q = (ProjQuery*)d_queries;
    tb_prof_begin(ctx, "k_project");
    if (map_overload)
        hipLaunchKernelGGL(k_project_map, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, P, *cam, d_mp, nq, sf0, nratio, q);
    else
        hipLaunchKernelGGL(k_project_frame, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, P, *cam, d_k2, d_mp, nq, d_sf,
                           nlevels, nratio, q, d_flag);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    tb_prof_begin(ctx, "k_window_q");
    hipLaunchKernelGGL(k_window_q, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, q, d_mpdesc, nq, d_k1, d_d1, d_taken1,
                       d_cellStart, d_cellItems, widthInv, heightInv, d_best);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ------------------------------------------------------------------------------------------------
 * SURVEY 8(f) row 3 -- Frame::AssignFeaturesToGrid (Frame.cpp:187-200, PosInGrid :257-265) on the device, and the
 * batched, device-resident form of searchByProjection(F1, F2) on top of it (no host round trip: projection,
 * window search, acceptance, rotation histogram and the ordered match list all stay in HBM).
 *
 * k_grid_build: one workgroup per frame. The 120x36 lookup grid as CSR: LDS histogram of the keys' cells, block
 * scan, scatter, then every cell's (short) item list is put back into key-index order -- the order
 * std::vector::push_back gives the reference, which decides ties in the matchers. */
#define GRID_CELLS (120 * 36)
__global__ void __launch_bounds__(256)
k_grid_build(const tb_keypoint* __restrict__ keys, const int32_t* __restrict__ counts, int key_pitch, float widthInv,
             float heightInv, int32_t* __restrict__ cellStart, int32_t* __restrict__ cellItems) {
    __shared__ int hist[GRID_CELLS];
    __shared__ int tmp[8];
    const int f = blockIdx.x, tid = threadIdx.x;
    const tb_keypoint* k = keys + (size_t)f * key_pitch;
    const int n = min(counts[f], key_pitch);
    int32_t* start = cellStart + (size_t)f * (GRID_CELLS + 1);
    int32_t* items = cellItems + (size_t)f * key_pitch;
    for (int c = tid; c < GRID_CELLS; c += 256) hist[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int posX = (int)roundf(k[i].x * widthInv), posY = (int)roundf(k[i].y * heightInv);
        if (posX >= 0 && posX < 120 && posY >= 0 && posY < 36) atomicAdd(&hist[posX * 36 + posY], 1);
    }
    __syncthreads();
    const int total = tb_block_excl_scan(hist, GRID_CELLS, tmp);
    for (int c = tid; c < GRID_CELLS; c += 256) start[c] = hist[c];
    if (tid == 0) start[GRID_CELLS] = total;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int posX = (int)roundf(k[i].x * widthInv), posY = (int)roundf(k[i].y * heightInv);
        if (posX >= 0 && posX < 120 && posY >= 0 && posY < 36) items[atomicAdd(&hist[posX * 36 + posY], 1)] = i;
    }
    __syncthreads(); /* hist[c] is now the END of cell c; the global writes below read what this block wrote */
    __threadfence_block();
    for (int c = tid; c < GRID_CELLS; c += 256) { /* insertion sort of the cell's items by key index */
        const int b = start[c], e = hist[c];
        for (int i = b + 1; i < e; i++) {
            const int v = items[i];
            int j = i - 1;
            while (j >= b && items[j] > v) { items[j + 1] = items[j]; j--; }
            items[j + 1] = v;
        }
    }
}

struct ProjBatch {
    const float* Tcw;                /* [npairs][16] */
    tb_camera cam;
    const tb_keypoint* k1; const uint8_t* d1; const uint8_t* taken1; const int32_t* n1; int pitch1;
    const int32_t* cellStart; const int32_t* cellItems;
    const tb_keypoint* k2; const tb_mappoint* mp2; const uint8_t* mp2d; const int32_t* n2; int pitch2;
    float sf[TB_MAX_LEVELS * 2]; int nlevels;
    float nratio, widthInv, heightInv, radio;
    int th_high, histo_len, check_orientation;
    int max_n2;   /* rows of `best` per pair = most map points any pair has; pitch2 may be 0 (one map shared by all pairs) */
    int map_mode; /* 0: searchByProjection(F1, F2); 1: searchByProjection(map, F1, radio) -- mp2 / mp2d are the map, k2 unused */
    int32_t* best;                   /* [npairs][pitch2][6] */
    tb_match* out; int cap; int32_t* out_counts; int32_t* flags; /* flags[p]: 1 = octave outside the table, 2 = bin outside the histogram */
};

/* projection + window search of one map point of pair blockIdx.y (k_project_frame + k_window_q fused) */
__global__ void __launch_bounds__(256)
k_proj_search_batch(ProjBatch B) {
    const int p = blockIdx.y, i2 = blockIdx.x * blockDim.x + threadIdx.x;
    const int n2 = min(B.n2[p], B.max_n2);
    if (i2 >= n2) return;
    ProjPose P;
#pragma unroll
    for (int i = 0; i < 16; i++) P.T[i] = B.Tcw[(size_t)p * 16 + i];
    const tb_mappoint mp = B.mp2[(size_t)p * B.pitch2 + i2];
    int bestDist = 256, bestDist2 = 256, bestIdx = -1, bestLevel = -1, bestLevel2 = -1, ncand = 0;
    bool search = false;
    float x = 0, y = 0, r = 0;
    int minL = 0, maxL = 0;
    if (!mp.bad && !B.map_mode) {
        float Pc[3], uv[2];
        pj_se3_map(P, mp.pos, Pc);
        const float invzc = 1.0f / Pc[2];
        if (!(invzc < 0)) {
            pj_world2cam(B.cam, Pc, uv);
            if (pj_in_frame(B.cam, uv)) {
                const int oct = B.k2[(size_t)p * B.pitch2 + i2].octave;
                if (oct < 0 || oct >= B.nlevels) B.flags[p] = 1; /* benign race */
                else { x = uv[0]; y = uv[1]; r = B.nratio * B.sf[oct]; minL = oct - 1; maxL = oct + 1; search = true; }
            }
        }
    } else if (!mp.bad) { /* Frame::IsInFrustum (Frame.cpp:370-412) + the window of matcher.cpp:558-567, as k_project_map */
        float Pc[3], uv[2], Ow[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float c0 = -P.T[i] * P.T[3], c1 = -P.T[4 + i] * P.T[7], c2 = -P.T[8 + i] * P.T[11];
            Ow[i] = c0 + (c1 + c2);
        }
        pj_se3_map(P, mp.pos, Pc);
        if (!(Pc[2] < 0.0f)) {
            pj_world2cam(B.cam, Pc, uv);
            if (pj_in_frame(B.cam, uv)) {
                const float PO[3] = {mp.pos[0] - Ow[0], mp.pos[1] - Ow[1], mp.pos[2] - Ow[2]};
                const float dist3 = sqrtf(PO[0] * PO[0] + (PO[1] * PO[1] + PO[2] * PO[2]));
                if (!(dist3 < mp.min_dist || dist3 > mp.max_dist)) {
                    const float viewCos = (PO[0] * mp.normal[0] + (PO[1] * mp.normal[1] + PO[2] * mp.normal[2])) / dist3;
                    if (!(viewCos < 0.5f)) {
                        float rr = 4.f;
                        if ((double)viewCos > 0.998) rr = 2.5f;
                        if ((double)B.nratio != 1.0) rr *= B.nratio;
                        x = uv[0]; y = uv[1]; r = rr * B.sf[0]; minL = -1; maxL = 0; search = true;
                    }
                }
            }
        }
    }
    if (search) {
        const int GRID_ROWS = 36, GRID_COLS = 120;
        const int nMinCellX = max(0, (int)floorf((x - r) * B.widthInv));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x + r) * B.widthInv));
        const int nMinCellY = max(0, (int)floorf((y - r) * B.heightInv));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y + r) * B.heightInv));
        if (nMinCellX < GRID_COLS && nMaxCellX >= 0 && nMinCellY < GRID_ROWS && nMaxCellY >= 0) {
            const bool bCheckLevels = (minL > 0) || (maxL >= 0);
            const int32_t* cs = B.cellStart + (size_t)p * (GRID_CELLS + 1);
            const int32_t* ci = B.cellItems + (size_t)p * B.pitch1;
            const tb_keypoint* k1 = B.k1 + (size_t)p * B.pitch1;
            const uint8_t* d1 = B.d1 + (size_t)p * B.pitch1 * 32;
            const uint8_t* tk = B.taken1 + (size_t)p * B.pitch1;
            const unsigned long long* a = reinterpret_cast<const unsigned long long*>(B.mp2d) + ((size_t)p * B.pitch2 + i2) * 4;
            Desc256 da;
            da.w[0] = a[0]; da.w[1] = a[1]; da.w[2] = a[2]; da.w[3] = a[3];
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const int c = ix * GRID_ROWS + iy;
                    for (int s = cs[c]; s < cs[c + 1]; s++) {
                        const int j = ci[s];
                        const tb_keypoint kp = k1[j];
                        if (bCheckLevels) {
                            if (kp.octave < minL) continue;
                            if (maxL >= 0 && kp.octave > maxL) continue;
                        }
                        if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) continue;
                        ncand++;
                        if (tk[j]) continue;
                        const int dist = bf_dist(da, reinterpret_cast<const unsigned long long*>(d1) + (size_t)j * 4);
                        if (dist < bestDist) {
                            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kp.octave; bestIdx = j;
                        } else if (dist < bestDist2) {
                            bestLevel2 = kp.octave; bestDist2 = dist;
                        }
                    }
                }
        }
    }
    int32_t* o = B.best + ((size_t)p * B.max_n2 + i2) * 6;
    o[0] = bestDist; o[1] = bestDist2; o[2] = bestIdx; o[3] = bestLevel; o[4] = bestLevel2; o[5] = ncand;
}

/* acceptance, rotation histogram, ComputeThreeMaxima and the ordered match list (matcher.cpp:483-530): one
 * workgroup per pair. The reference's output order -- kept bins ascending, inside a bin the order of acceptance --
 * is a stable partition: one ordered compaction pass per kept bin (at most three). */
__global__ void __launch_bounds__(256)
k_proj_accept_batch(ProjBatch B) {
    __shared__ int hist[1024];
    __shared__ int sflag[256];
    __shared__ int tmp[8];
    __shared__ int keep[3];
    __shared__ int srun;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n2 = min(B.n2[p], B.max_n2);
    const int32_t* best = B.best + (size_t)p * B.max_n2 * 6;
    const tb_keypoint* k1 = B.k1 + (size_t)p * B.pitch1;
    const tb_keypoint* k2 = B.k2 + (size_t)p * B.pitch2;
    tb_match* out = B.out + (size_t)p * B.cap;
    const float factor = 1.0f / (float)B.histo_len;
    auto accepted = [&](int i2, int& bin) -> bool {
        if (i2 >= n2) return false;
        const int bd = best[6 * (size_t)i2], bi = best[6 * (size_t)i2 + 2];
        if (best[6 * (size_t)i2 + 5] == 0 || bi < 0 || bd > B.th_high) return false;
        if (B.map_mode && best[6 * (size_t)i2 + 3] == best[6 * (size_t)i2 + 4] &&
            (float)bd > B.radio * (float)best[6 * (size_t)i2 + 1]) return false; /* matcher.cpp:608-609 */
        bin = 0;
        if (B.check_orientation) {
            float rot = k2[i2].angle - k1[bi].angle;
            if (rot < 0.0f) rot += 360.0f;
            bin = (int)roundf(rot * factor);
            if (bin == B.histo_len) bin = 0;
            if (bin < 0 || bin >= B.histo_len) { B.flags[p] = 2; return false; } /* the reference asserts */
        }
        return true;
    };
    if (tid == 0) { keep[0] = B.check_orientation ? -1 : 0; keep[1] = keep[2] = -1; srun = 0; }
    for (int b = tid; b < B.histo_len; b += 256) hist[b] = 0;
    __syncthreads();
    if (B.check_orientation) {
        for (int i2 = tid; i2 < n2; i2 += 256) { int bin; if (accepted(i2, bin)) atomicAdd(&hist[bin], 1); }
        __syncthreads();
        if (tid == 0) { /* Matcher::ComputeThreeMaxima, matcher.cpp:810-851 */
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < B.histo_len; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            /* kept bins in ascending order */
            int a = i1, b = i2, c = i3, t;
            if (a < 0) a = 1 << 30; if (b < 0) b = 1 << 30; if (c < 0) c = 1 << 30;
            if (a > b) { t = a; a = b; b = t; } if (b > c) { t = b; b = c; c = t; } if (a > b) { t = a; a = b; b = t; }
            keep[0] = a < (1 << 30) ? a : -1; keep[1] = b < (1 << 30) ? b : -1; keep[2] = c < (1 << 30) ? c : -1;
        }
        __syncthreads();
    }
    for (int kb = 0; kb < 3; kb++) {
        const int want = keep[kb];
        if (want < 0) continue;
        for (int e0 = 0; e0 < n2; e0 += 256) {
            const int i2 = e0 + tid;
            int bin = 0;
            const int f = (accepted(i2, bin) && (!B.check_orientation || bin == want)) ? 1 : 0;
            sflag[tid] = f;
            __syncthreads();
            const int total = tb_block_excl_scan(sflag, 256, tmp);
            const int slot = srun + sflag[tid];
            if (f && slot < B.cap) {
                tb_match m;
                m.queryIdx = best[6 * (size_t)i2 + 2]; m.trainIdx = i2; m.imgIdx = -1; m.distance = (float)best[6 * (size_t)i2];
                out[slot] = m;
            }
            __syncthreads();
            if (tid == 0) srun += total;
            __syncthreads();
        }
    }
    if (tid == 0) B.out_counts[p] = srun; /* may exceed cap: the list is truncated, the count is not */
}

int tbk_grid_build_batch(tb_ctx* ctx, int nframes, const tb_keypoint* d_keys, const int32_t* d_counts, int key_pitch, int img_w,
                         int img_h, int32_t* d_cellStart, int32_t* d_cellItems) {
    if (nframes <= 0) return TB_OK;
    const float heightInv = 120.f / (float)img_w, widthInv = 36.f / (float)img_h; /* swapped in the reference; kept */
    tb_prof_begin(ctx, "k_grid_build");
    hipLaunchKernelGGL(k_grid_build, dim3(nframes), dim3(256), 0, ctx->stream, d_keys, d_counts, key_pitch, widthInv, heightInv,
                       d_cellStart, d_cellItems);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

int tbk_projection_batch(tb_ctx* ctx, int npairs, const float* d_Tcw, const tb_camera* cam, int img_w, int img_h,
                         const tb_keypoint* d_k1, const uint8_t* d_d1, const uint8_t* d_taken1, const int32_t* d_n1, int pitch1,
                         const int32_t* d_cellStart, const int32_t* d_cellItems, const tb_keypoint* d_k2, const tb_mappoint* d_mp2,
                         const uint8_t* d_mp2d, const int32_t* d_n2, int pitch2, const float* sf, int nlevels, float nratio,
                         int th_high, int histo_len, int check_orientation, int32_t* d_best, tb_match* d_out, int cap,
                         int32_t* d_out_counts, int32_t* d_flags, int map_mode, float radio, int max_n2) {
    if (npairs <= 0 || max_n2 <= 0) return TB_OK;
    ProjBatch B;
    B.map_mode = map_mode; B.radio = radio; B.max_n2 = max_n2;
    B.Tcw = d_Tcw; B.cam = *cam;
    B.k1 = d_k1; B.d1 = d_d1; B.taken1 = d_taken1; B.n1 = d_n1; B.pitch1 = pitch1;
    B.cellStart = d_cellStart; B.cellItems = d_cellItems;
    B.k2 = d_k2; B.mp2 = d_mp2; B.mp2d = d_mp2d; B.n2 = d_n2; B.pitch2 = pitch2;
    for (int i = 0; i < TB_MAX_LEVELS * 2; i++) B.sf[i] = i < nlevels ? sf[i] : 0.f;
    B.nlevels = nlevels; B.nratio = nratio;
    B.heightInv = 120.f / (float)img_w; B.widthInv = 36.f / (float)img_h;
    B.th_high = th_high; B.histo_len = histo_len; B.check_orientation = check_orientation;
    B.best = d_best; B.out = d_out; B.cap = cap; B.out_counts = d_out_counts; B.flags = d_flags;
    TB_HIP(ctx, hipMemsetAsync(d_flags, 0, (size_t)npairs * sizeof(int32_t), ctx->stream));
    tb_prof_begin(ctx, "k_proj_search");
    hipLaunchKernelGGL(k_proj_search_batch, dim3((max_n2 + 255) / 256, npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    tb_prof_begin(ctx, "k_proj_accept");
    hipLaunchKernelGGL(k_proj_accept_batch, dim3(npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Batched, device-resident Matcher::searchByViolence (matcher.cpp:299-395) on the device-built lookup grids: pair =
 * blockIdx.y; window search per F1 key (k_window's body), then acceptance (th_low, nratio), rotation histogram,
 * ComputeThreeMaxima and the reference's output order, one workgroup per pair. */
struct VioBatch {
    const tb_keypoint* k1; const uint8_t* d1; const int32_t* n1; int pitch1;
    const tb_keypoint* k2; const uint8_t* d2; const int32_t* n2; int pitch2;
    const int32_t* cellStart; const int32_t* cellItems; /* grids of the F2 frames */
    float widthInv, heightInv, r, nratio;
    int min_level, max_level, th_low, histo_len, check_orientation;
    int32_t* best;            /* [npairs][pitch1][4]: bestDist, bestDist2, bestIdx, candidates */
    tb_match* out; int cap; int32_t* out_counts; int32_t* flags;
};

__global__ void __launch_bounds__(256)
k_window_batch(VioBatch B) {
    const int GRID_ROWS = 36, GRID_COLS = 120;
    const int p = blockIdx.y, i1 = blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = min(B.n1[p], B.pitch1);
    if (i1 >= n1) return;
    const tb_keypoint* k1 = B.k1 + (size_t)p * B.pitch1;
    const tb_keypoint* k2 = B.k2 + (size_t)p * B.pitch2;
    const uint8_t* d2 = B.d2 + (size_t)p * B.pitch2 * 32;
    const int32_t* cs = B.cellStart + (size_t)p * (GRID_CELLS + 1);
    const int32_t* ci = B.cellItems + (size_t)p * B.pitch2;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx = -1, ncand = 0;
    const float x = k1[i1].x, y = k1[i1].y, r = B.r;
    const int nMinCellX = max(0, (int)floorf((x - r) * B.widthInv));
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x + r) * B.widthInv));
    const int nMinCellY = max(0, (int)floorf((y - r) * B.heightInv));
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y + r) * B.heightInv));
    if (nMinCellX < GRID_COLS && nMaxCellX >= 0 && nMinCellY < GRID_ROWS && nMaxCellY >= 0) {
        const bool bCheckLevels = (B.min_level > 0) || (B.max_level >= 0);
        const unsigned long long* a = reinterpret_cast<const unsigned long long*>(B.d1) + ((size_t)p * B.pitch1 + i1) * 4;
        Desc256 da;
        da.w[0] = a[0]; da.w[1] = a[1]; da.w[2] = a[2]; da.w[3] = a[3];
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const int c = ix * GRID_ROWS + iy;
                for (int s = cs[c]; s < cs[c + 1]; s++) {
                    const int j = ci[s];
                    const tb_keypoint kp = k2[j];
                    if (bCheckLevels) {
                        if (kp.octave < B.min_level) continue;
                        if (B.max_level >= 0 && kp.octave > B.max_level) continue;
                    }
                    if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) continue;
                    ncand++;
                    const int dist = bf_dist(da, reinterpret_cast<const unsigned long long*>(d2) + (size_t)j * 4);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
            }
    }
    int32_t* o = B.best + ((size_t)p * B.pitch1 + i1) * 4;
    o[0] = bestDist; o[1] = bestDist2; o[2] = bestIdx; o[3] = ncand;
}

__global__ void __launch_bounds__(256)
k_violence_accept_batch(VioBatch B) {
    __shared__ int hist[1024];
    __shared__ int sflag[256];
    __shared__ int tmp[8];
    __shared__ int keep[3];
    __shared__ int srun;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n1 = min(B.n1[p], B.pitch1);
    const int32_t* best = B.best + (size_t)p * B.pitch1 * 4;
    const tb_keypoint* k1 = B.k1 + (size_t)p * B.pitch1;
    const tb_keypoint* k2 = B.k2 + (size_t)p * B.pitch2;
    tb_match* out = B.out + (size_t)p * B.cap;
    const float factor = 1.f / (float)B.histo_len;
    auto accepted = [&](int i1, int& bin) -> bool { /* matcher.cpp:352-377 */
        if (i1 >= n1) return false;
        const int bd = best[4 * (size_t)i1], bd2 = best[4 * (size_t)i1 + 1], bi = best[4 * (size_t)i1 + 2];
        if (best[4 * (size_t)i1 + 3] == 0) return false;
        if (!(bd <= B.th_low && (float)bd < (float)bd2 * B.nratio)) return false;
        bin = 0;
        if (B.check_orientation) {
            float rot = k1[i1].angle - k2[bi].angle;
            if (rot < 0) rot += 360.f;
            bin = (int)roundf(rot * factor);
            if (bin == B.histo_len) bin = 0;
            if (bin < 0 || bin >= B.histo_len) { B.flags[p] = 2; return false; } /* the reference asserts */
        }
        return true;
    };
    if (tid == 0) { keep[0] = B.check_orientation ? -1 : 0; keep[1] = keep[2] = -1; srun = 0; }
    for (int b = tid; b < B.histo_len; b += 256) hist[b] = 0;
    __syncthreads();
    if (B.check_orientation) {
        for (int i1 = tid; i1 < n1; i1 += 256) { int bin; if (accepted(i1, bin)) atomicAdd(&hist[bin], 1); }
        __syncthreads();
        if (tid == 0) { /* Matcher::ComputeThreeMaxima, matcher.cpp:810-851 */
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < B.histo_len; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            int a = i1 < 0 ? (1 << 30) : i1, b = i2 < 0 ? (1 << 30) : i2, c = i3 < 0 ? (1 << 30) : i3, t;
            if (a > b) { t = a; a = b; b = t; }
            if (b > c) { t = b; b = c; c = t; }
            if (a > b) { t = a; a = b; b = t; }
            keep[0] = a < (1 << 30) ? a : -1; keep[1] = b < (1 << 30) ? b : -1; keep[2] = c < (1 << 30) ? c : -1;
        }
        __syncthreads();
    }
    for (int kb = 0; kb < 3; kb++) {
        const int want = keep[kb];
        if (want < 0) continue;
        for (int e0 = 0; e0 < n1; e0 += 256) {
            const int i1 = e0 + tid;
            int bin = 0;
            const int f = (accepted(i1, bin) && (!B.check_orientation || bin == want)) ? 1 : 0;
            sflag[tid] = f;
            __syncthreads();
            const int total = tb_block_excl_scan(sflag, 256, tmp);
            const int slot = srun + sflag[tid];
            if (f && slot < B.cap) {
                tb_match m;
                m.queryIdx = i1; m.trainIdx = best[4 * (size_t)i1 + 2]; m.imgIdx = -1; m.distance = (float)best[4 * (size_t)i1];
                out[slot] = m;
            }
            __syncthreads();
            if (tid == 0) srun += total;
            __syncthreads();
        }
    }
    if (tid == 0) B.out_counts[p] = srun;
}

int tbk_violence_batch(tb_ctx* ctx, int npairs, const tb_keypoint* d_k1, const uint8_t* d_d1, const int32_t* d_n1, int pitch1,
                       const tb_keypoint* d_k2, const uint8_t* d_d2, const int32_t* d_n2, int pitch2, const int32_t* d_cellStart,
                       const int32_t* d_cellItems, int img2_w, int img2_h, int min_level, int max_level, float radius, int th_low,
                       float nratio, int histo_len, int check_orientation, int32_t* d_best, tb_match* d_out, int cap,
                       int32_t* d_out_counts, int32_t* d_flags) {
    if (npairs <= 0 || pitch1 <= 0) return TB_OK;
    VioBatch B;
    B.k1 = d_k1; B.d1 = d_d1; B.n1 = d_n1; B.pitch1 = pitch1;
    B.k2 = d_k2; B.d2 = d_d2; B.n2 = d_n2; B.pitch2 = pitch2;
    B.cellStart = d_cellStart; B.cellItems = d_cellItems;
    B.heightInv = 120.f / (float)img2_w; B.widthInv = 36.f / (float)img2_h; /* swapped in the reference; kept */
    B.r = radius; B.nratio = nratio; B.min_level = min_level; B.max_level = max_level;
    B.th_low = th_low; B.histo_len = histo_len; B.check_orientation = check_orientation;
    B.best = d_best; B.out = d_out; B.cap = cap; B.out_counts = d_out_counts; B.flags = d_flags;
    TB_HIP(ctx, hipMemsetAsync(d_flags, 0, (size_t)npairs * sizeof(int32_t), ctx->stream));
    tb_prof_begin(ctx, "k_window_batch");
    hipLaunchKernelGGL(k_window_batch, dim3((pitch1 + 255) / 256, npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    tb_prof_begin(ctx, "k_violence_accept");
    hipLaunchKernelGGL(k_violence_accept_batch, dim3(npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- Matcher::searchByBow (matcher.cpp:619-721), the search over shared vocabulary nodes.
 * One thread per QUERY = one feature of F1 that lies in a node both frames have (the host's walk of the two sorted node
 * lists, matcher.cpp:637-698, yields the queries in the reference's emission order): best / second-best Hamming
 * distance over F2's features of that node in list order (:645-669). best[q] = {bestDist1, bestDist2, bestIdx2, 0}.
 * Bound: gather latency (a node holds a handful of features); no SURVEY 8(d) row. */
__global__ void __launch_bounds__(256)
k_bow_search(int nq, const int4* __restrict__ queries /* idx1, start2, end2, 0 */, const uint8_t* __restrict__ d1,
             const uint8_t* __restrict__ d2, const uint32_t* __restrict__ items2, const uint8_t* __restrict__ has_mp2,
             int map_point_only, int4* __restrict__ best) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int4 qu = queries[q];
    Desc256 a;
    const unsigned long long* pa = reinterpret_cast<const unsigned long long*>(d1 + 32 * (size_t)qu.x);
    a.w[0] = pa[0]; a.w[1] = pa[1]; a.w[2] = pa[2]; a.w[3] = pa[3];
    int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
    for (int p2 = qu.y; p2 < qu.z; p2++) {
        const int idx2 = (int)items2[p2];
        if (map_point_only && !(has_mp2 && has_mp2[idx2])) continue;
        const int dist = bf_dist(a, reinterpret_cast<const unsigned long long*>(d2 + 32 * (size_t)idx2));
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
        else if (dist < bestDist2) bestDist2 = dist;
    }
    best[q] = make_int4(bestDist1, bestDist2, bestIdx2, 0);
}

int tbk_bow_search(tb_ctx* ctx, int nq, const void* d_queries, const uint8_t* d_d1, const uint8_t* d_d2, const uint32_t* d_items2,
                   const uint8_t* d_has_mp2, int map_point_only, void* d_best) {
    if (nq <= 0) return TB_OK;
    tb_prof_begin(ctx, "k_bow_search");
    hipLaunchKernelGGL(k_bow_search, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, nq, (const int4*)d_queries, d_d1, d_d2, d_items2,
                       d_has_mp2, map_point_only, (int4*)d_best);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- stereo tracks -> pose-optimisation observations (round 3; SURVEY 3.5 / LocalBA.cpp:46-68, :333-363).
 * The reference's per-frame flow is stereo depth -> map points -> PoseOptimization (test/test_vo.cpp:716,761,800): a key of
 * the current frame gets Depth = bf / |x_other - x_key| (LocalBA.cpp:60-64), the test back-projects it through the pinhole
 * model (test_vo.cpp:257-267: norm = ((u - cx) / fx, (v - cy) / fy, 1), X = norm * depth), and PoseOptimization reads per
 * edge the pixel, the point and invSigma2[octave] (LocalBA.cpp:333-363). Here the two keys of a left <-> right match of
 * searchByBF play those roles: X from the LEFT key and the disparity, observed at the RIGHT key's pixel, so the pose that
 * PoseOptimization finds from the identity is the right camera's (a translation by the baseline bf / fx along x, plus whatever
 * the vertical mismatches of the matches ask for). One workgroup per frame; rows in match order; a match without
 * disparity (infinite depth) or with an octave outside the table is dropped by an ordered compaction. float32 arithmetic,
 * one operation per statement (no contraction), the same in oracle_match.cpp. */
__global__ void __launch_bounds__(256)
k_stereo_obs(const tb_keypoint* __restrict__ kl, const tb_keypoint* __restrict__ kr, int key_pitch, const tb_match* __restrict__ matches,
             const int32_t* __restrict__ match_counts, int match_pitch, float fx, float fy, float cx, float cy, float bf,
             const float* __restrict__ inv_sigma2, int nlevels, tb_obs* __restrict__ obs, int obs_pitch, int32_t* __restrict__ obs_counts) {
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(match_counts[f], match_pitch);
    const tb_keypoint* L = kl + (size_t)f * key_pitch;
    const tb_keypoint* Rk = kr + (size_t)f * key_pitch;
    const tb_match* M = matches + (size_t)f * match_pitch;
    tb_obs* O = obs + (size_t)f * obs_pitch;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        bool ok = false;
        tb_obs o = {0, 0, 0, 0, 0, 0};
        if (i < n) {
            const tb_match m = M[i];
            const tb_keypoint a = L[m.queryIdx], b = Rk[m.trainIdx];
            const float depth = bf / fabsf(b.x - a.x);               /* LocalBA.cpp:64 */
            const float nx = (a.x - cx) / fx, ny = (a.y - cy) / fy;  /* test_vo.cpp:257-261 */
            o.u = b.x; o.v = b.y;
            o.X = nx * depth; o.Y = ny * depth; o.Z = depth;
            ok = isfinite(depth) && b.octave >= 0 && b.octave < nlevels;
            o.inv_sigma2 = ok ? inv_sigma2[b.octave] : 0.f;
        }
        const unsigned long long bm = __ballot(ok);
        if (lane == 0) wsum[wave] = __popcll(bm);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wsum[w];
        const int at = off + __popcll(bm & ((1ull << lane) - 1));
        if (ok && at < obs_pitch) O[at] = o;
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) obs_counts[f] = min(base, obs_pitch);
}

int tbk_stereo_obs(tb_ctx* ctx, int nframes, const tb_keypoint* d_kl, const tb_keypoint* d_kr, int key_pitch, const tb_match* d_matches,
                   const int32_t* d_match_counts, int match_pitch, const float K[4], float bf, const float* d_inv_sigma2, int nlevels,
                   tb_obs* d_obs, int obs_pitch, int32_t* d_obs_counts) {
    if (nframes <= 0) return TB_OK;
    tb_prof_begin(ctx, "k_stereo_obs");
    hipLaunchKernelGGL(k_stereo_obs, dim3(nframes), dim3(256), 0, ctx->stream, d_kl, d_kr, key_pitch, d_matches, d_match_counts, match_pitch,
                       K[0], K[1], K[2], K[3], bf, d_inv_sigma2, nlevels, d_obs, obs_pitch, d_obs_counts);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- SURVEY 8(f) row 4, second half: the DBoW2 transform (TemplatedVocabulary::transform, TemplatedVocabulary.h:1218-1260).
 * One thread per descriptor walks the tree: at every level the Hamming distance (FORB::distance, FORB.cpp:81-101) to each
 * child of the current node, the first child with the smallest distance wins (strict <, :1238). The children of a node are
 * consecutive 32-byte rows gathered through the L2 (a 10^6-node ORB vocabulary is 32 MB; the upper levels stay cached, the
 * leaves are one 320-byte gather per feature); ~k L = 60 distances per feature. Bound: gather latency; no SURVEY 8(d) row. */
struct BowVocab {
    int nnodes, L;
    const int32_t* child_start;
    const int32_t* child_items;
    const uint8_t* desc;
    const int32_t* word_id;
    const double* weight;
};
__global__ void __launch_bounds__(256)
k_bow_transform(BowVocab V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int desc_pitch, int levelsup,
                int32_t* __restrict__ word_ids, int32_t* __restrict__ node_ids, double* __restrict__ weights) {
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts ? min(counts[f], desc_pitch) : desc_pitch;
    if (i >= n) return;
    const size_t at = (size_t)f * desc_pitch + i;
    Desc256 a;
    const unsigned long long* pa = reinterpret_cast<const unsigned long long*>(desc + 32 * at);
    a.w[0] = pa[0]; a.w[1] = pa[1]; a.w[2] = pa[2]; a.w[3] = pa[3];
    const int nid_level = V.L - levelsup;
    int final_id = 0, level = 0, nid = 0;
    bool nid_set = nid_level <= 0;   /* root (TemplatedVocabulary.h:1227) */
    int c0 = V.child_start[0], c1 = V.child_start[1];
    while (c1 > c0) {
        level++;
        int best = V.child_items[c0];
        int best_d = bf_dist(a, reinterpret_cast<const unsigned long long*>(V.desc + 32 * (size_t)best));
        for (int c = c0 + 1; c < c1; c++) {
            const int id = V.child_items[c];
            const int dd = bf_dist(a, reinterpret_cast<const unsigned long long*>(V.desc + 32 * (size_t)id));
            if (dd < best_d) { best_d = dd; best = id; }
        }
        final_id = best;
        if (level == nid_level) { nid = final_id; nid_set = true; }
        c0 = V.child_start[final_id]; c1 = V.child_start[final_id + 1];
    }
    if (!nid_set) nid = final_id; /* the branch ended above level L - levelsup: the reference leaves *nid unset */
    if (word_ids) word_ids[at] = V.word_id[final_id];
    if (weights) weights[at] = V.weight[final_id];
    if (node_ids) node_ids[at] = nid;
}

/* The frame's FeatureVector as a sorted key list: (node id << 32 | feature index) of the features whose word is not
 * stopped (w > 0, TemplatedVocabulary.h:1159), ascending -- the std::map's node order, each node's features in insertion
 * order. One workgroup per frame: keys into LDS (padding = all ones), bitonic sort, count of real keys. */
#define BOW_MAXN 8192
__global__ void __launch_bounds__(1024)
k_bow_fv_sort(const int32_t* __restrict__ node_ids, const double* __restrict__ weights, const int32_t* __restrict__ counts,
              int desc_pitch, unsigned long long* __restrict__ keys_out, int32_t* __restrict__ fv_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
    __shared__ int wsum[16];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = counts ? min(counts[f], desc_pitch) : desc_pitch;
    int m = 1;
    while (m < n) m <<= 1;
    int mine = 0;
    for (int i = tid; i < m; i += 1024) {
        unsigned long long k = ~0ull;
        if (i < n && weights[(size_t)f * desc_pitch + i] > 0) { k = ((unsigned long long)(unsigned)node_ids[(size_t)f * desc_pitch + i] << 32) | (unsigned)i; mine++; }
        sk[i] = k;
    }
    mine = tb_wave_sum(mine);
    if ((tid & 63) == 0) wsum[tid >> 6] = mine;
    __syncthreads();
    for (int k2 = 2; k2 <= m; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = sk[i], y = sk[l];
                    const bool up = (i & k2) == 0;
                    if ((x > y) == up) { sk[i] = y; sk[l] = x; }
                }
            }
            __syncthreads();
        }
    int total = 0;
    for (int w = 0; w < 16; w++) total += wsum[w];
    for (int i = tid; i < total; i += 1024) keys_out[(size_t)f * desc_pitch + i] = sk[i];
    if (tid == 0) fv_counts[f] = total;
}

/* Batched searchByBow, search stage: one thread per entry of F1's feature vector (= the reference's emission order once the
 * entries of nodes F2 does not have are dropped): the node's entries of F2 by binary search, then best / second best Hamming
 * distance in list order (matcher.cpp:645-669). best[pos] = {bestDist1, bestDist2, bestIdx2, 1 if F2 has the node}. */
struct BowBatch {
    const tb_keypoint *k1, *k2;
    const uint8_t *d1, *d2, *has_mp2;
    const unsigned long long *fv1, *fv2;
    const int32_t *n1, *n2;
    int pitch1, pitch2, map_point_only, th_low, histo_len, check_orientation, cap;
    float nratio;
    int32_t* best;
    tb_match* out;
    int32_t *out_counts, *flags;
};
__global__ void __launch_bounds__(256)
k_bow_search_batch(BowBatch B) {
    const int p = blockIdx.y, pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = min(B.n1[p], B.pitch1), n2 = min(B.n2[p], B.pitch2);
    if (pos >= n1) return;
    const unsigned long long key = B.fv1[(size_t)p * B.pitch1 + pos];
    const unsigned node = (unsigned)(key >> 32), idx1 = (unsigned)key;
    const unsigned long long* F2 = B.fv2 + (size_t)p * B.pitch2;
    int lo = 0, hi = n2;
    const unsigned long long want = (unsigned long long)node << 32;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (F2[mid] < want) lo = mid + 1; else hi = mid; }
    int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256, found = 0;
    if (idx1 < (unsigned)B.pitch1) {
        Desc256 a;
        const unsigned long long* pa = reinterpret_cast<const unsigned long long*>(B.d1 + 32 * ((size_t)p * B.pitch1 + idx1));
        a.w[0] = pa[0]; a.w[1] = pa[1]; a.w[2] = pa[2]; a.w[3] = pa[3];
        for (int q = lo; q < n2 && (unsigned)(F2[q] >> 32) == node; q++) {
            found = 1;
            const unsigned idx2 = (unsigned)F2[q];
            if (idx2 >= (unsigned)B.pitch2) continue;
            if (B.map_point_only && !(B.has_mp2 && B.has_mp2[(size_t)p * B.pitch2 + idx2])) continue;
            const int dist = bf_dist(a, reinterpret_cast<const unsigned long long*>(B.d2 + 32 * ((size_t)p * B.pitch2 + idx2)));
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
    }
    reinterpret_cast<int4*>(B.best)[(size_t)p * B.pitch1 + pos] = make_int4(bestDist1, bestDist2, bestIdx2, found);
}
/* acceptance (matcher.cpp:671-689), rotation histogram, ComputeThreeMaxima and the reference's output order (kept bins in
 * ascending order, emission order inside a bin: matcher.cpp:703-717) -- the structure of k_violence_accept_batch */
__global__ void __launch_bounds__(256)
k_bow_accept_batch(BowBatch B) {
    __shared__ int hist[1024];
    __shared__ int sflag[256];
    __shared__ int tmp[8];
    __shared__ int keep[3];
    __shared__ int srun;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int n1 = min(B.n1[p], B.pitch1);
    const int32_t* best = B.best + (size_t)p * B.pitch1 * 4;
    const unsigned long long* F1 = B.fv1 + (size_t)p * B.pitch1;
    const tb_keypoint* k1 = B.k1 + (size_t)p * B.pitch1;
    const tb_keypoint* k2 = B.k2 + (size_t)p * B.pitch2;
    tb_match* out = B.out + (size_t)p * B.cap;
    const float factor = 1.f / (float)B.histo_len;
    auto accepted = [&](int pos, int& bin) -> bool {
        if (pos >= n1) return false;
        const int bd = best[4 * (size_t)pos], bd2 = best[4 * (size_t)pos + 1], bi = best[4 * (size_t)pos + 2];
        if (best[4 * (size_t)pos + 3] == 0 || bi < 0) return false;
        if (!(bd < B.th_low && (float)bd < B.nratio * (float)bd2)) return false;
        bin = 0;
        if (B.check_orientation) {
            float rot = k1[(unsigned)F1[pos]].angle - k2[bi].angle;
            if (rot < 0) rot += 360.f;
            bin = (int)roundf(rot * factor);
            if (bin == B.histo_len) bin = 0;
            if (bin < 0 || bin >= B.histo_len) { B.flags[p] = 2; return false; } /* the reference asserts */
        }
        return true;
    };
    if (tid == 0) { keep[0] = B.check_orientation ? -1 : 0; keep[1] = keep[2] = -1; srun = 0; B.flags[p] = 0; }
    for (int b = tid; b < B.histo_len; b += 256) hist[b] = 0;
    __syncthreads();
    if (B.check_orientation) {
        for (int pos = tid; pos < n1; pos += 256) { int bin; if (accepted(pos, bin)) atomicAdd(&hist[bin], 1); }
        __syncthreads();
        if (tid == 0) { /* Matcher::ComputeThreeMaxima, matcher.cpp:810-851 */
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < B.histo_len; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            int a = i1 < 0 ? (1 << 30) : i1, b = i2 < 0 ? (1 << 30) : i2, c = i3 < 0 ? (1 << 30) : i3, t;
            if (a > b) { t = a; a = b; b = t; }
            if (b > c) { t = b; b = c; c = t; }
            if (a > b) { t = a; a = b; b = t; }
            keep[0] = a < (1 << 30) ? a : -1; keep[1] = b < (1 << 30) ? b : -1; keep[2] = c < (1 << 30) ? c : -1;
        }
        __syncthreads();
    }
    for (int kb = 0; kb < 3; kb++) {
        const int want = keep[kb];
        if (want < 0) continue;
        for (int e0 = 0; e0 < n1; e0 += 256) {
            const int pos = e0 + tid;
            int bin = 0;
            const int f = (accepted(pos, bin) && (!B.check_orientation || bin == want)) ? 1 : 0;
            sflag[tid] = f;
            __syncthreads();
            const int total = tb_block_excl_scan(sflag, 256, tmp);
            const int slot = srun + sflag[tid];
            if (f && slot < B.cap) {
                tb_match m;
                m.queryIdx = (int)(unsigned)F1[pos]; m.trainIdx = best[4 * (size_t)pos + 2]; m.imgIdx = -1; m.distance = (float)best[4 * (size_t)pos];
                out[slot] = m;
            }
            __syncthreads();
            if (tid == 0) srun += total;
            __syncthreads();
        }
    }
    if (tid == 0) B.out_counts[p] = srun;
}

int tbk_bow_transform(tb_ctx* ctx, int nnodes, int L, const int32_t* d_child_start, const int32_t* d_child_items, const uint8_t* d_vdesc,
                      const int32_t* d_word_id, const double* d_weight, int nframes, const uint8_t* d_desc, const int32_t* d_counts,
                      int desc_pitch, int levelsup, int32_t* d_word_ids, int32_t* d_node_ids, double* d_weights,
                      unsigned long long* d_fv_keys, int32_t* d_fv_counts) {
    if (nframes <= 0 || desc_pitch <= 0) return TB_OK;
    BowVocab V = {nnodes, L, d_child_start, d_child_items, d_vdesc, d_word_id, d_weight};
    tb_prof_begin(ctx, "k_bow_transform");
    hipLaunchKernelGGL(k_bow_transform, dim3((desc_pitch + 255) / 256, nframes), dim3(256), 0, ctx->stream, V, d_desc, d_counts, desc_pitch,
                       levelsup, d_word_ids, d_node_ids, d_weights);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    if (d_fv_keys) {
        int m = 1;
        while (m < desc_pitch) m <<= 1;
        const size_t lds = (size_t)m * sizeof(unsigned long long);
        TB_HIP(ctx, hipFuncSetAttribute((const void*)k_bow_fv_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        tb_prof_begin(ctx, "k_bow_fv_sort");
        hipLaunchKernelGGL(k_bow_fv_sort, dim3(nframes), dim3(1024), lds, ctx->stream, d_node_ids, d_weights, d_counts, desc_pitch, d_fv_keys,
                           d_fv_counts);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
    }
    return TB_OK;
}

int tbk_bow_search_batch(tb_ctx* ctx, int npairs, const tb_keypoint* d_k1, const uint8_t* d_d1, int pitch1, const unsigned long long* d_fv1,
                         const int32_t* d_n1, const tb_keypoint* d_k2, const uint8_t* d_d2, int pitch2, const unsigned long long* d_fv2,
                         const int32_t* d_n2, const uint8_t* d_has_mp2, int map_point_only, int th_low, float nratio, int histo_len,
                         int check_orientation, tb_match* d_out, int cap, int32_t* d_out_counts, int32_t* d_flags, int32_t* d_best) {
    if (npairs <= 0) return TB_OK;
    BowBatch B;
    B.k1 = d_k1; B.k2 = d_k2; B.d1 = d_d1; B.d2 = d_d2; B.has_mp2 = d_has_mp2; B.fv1 = d_fv1; B.fv2 = d_fv2; B.n1 = d_n1; B.n2 = d_n2;
    B.pitch1 = pitch1; B.pitch2 = pitch2; B.map_point_only = map_point_only; B.th_low = th_low; B.histo_len = histo_len;
    B.check_orientation = check_orientation; B.cap = cap; B.nratio = nratio; B.best = d_best; B.out = d_out; B.out_counts = d_out_counts;
    B.flags = d_flags;
    tb_prof_begin(ctx, "k_bow_search_batch");
    hipLaunchKernelGGL(k_bow_search_batch, dim3((pitch1 + 255) / 256, npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    tb_prof_begin(ctx, "k_bow_accept_batch");
    hipLaunchKernelGGL(k_bow_accept_batch, dim3(npairs), dim3(256), 0, ctx->stream, B);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
