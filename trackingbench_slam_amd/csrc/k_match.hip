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
