/* SURVEY 8f row 2, first part -- pyramidal Lucas-Kanade tracking for Matcher::searchByOPFlow.
 *
 * Reference call site: src/matchers/matcher.cpp:744
 *     cv::calcOpticalFlowPyrLK(img2, img1, F2->GetCVKeys(), cur_points, status, err, cv::Size(21, 21), 3);
 * The routine itself is OpenCV 3.3 (not in the reference tree): restated from its published structure -- 5x5 Gaussian
 * pyrDown pyramid read with BORDER_REFLECT_101 padding, Scharr derivatives (zero outside the image), W_BITS = 14
 * fixed-point bilinear weights, iterative 2x2 solve, 30 iterations / eps 0.01, minEigThreshold 1e-4, L1 error.
 * The window sums (A11, A12, A22, b1, b2, error) are exact 64-bit INTEGER sums converted to float once (OpenCV adds
 * float products in a SIMD-path dependent order): the result does not depend on the order of summation, so a
 * wavefront tree sum and a sequential loop agree bit for bit.
 *
 *   k_pyr_down  : thread per destination pixel, 25 taps, (sum + 128) >> 8
 *   k_lk_track  : ONE WAVEFRONT PER POINT, all pyramid levels in one launch. Per level the 24 x 24 source patch goes to
 *                 LDS once (reflect-101 indexing), the 22 x 22 Scharr derivatives are formed from it in LDS, every lane
 *                 keeps its 7 of the 441 interpolated window samples (I, Ix, Iy) in registers; the iterations read the
 *                 second image from a 34 x 34 region cached in LDS and reduce two sums over the wavefront on the DPP
 *                 path. No derivative images and no padded pyramid copies exist in HBM.
 * Bound: latency / LDS (a few KB per point per iteration); no 8d row. */
#include "tb_internal.h"
#include "tb_device.h"

#define LK_WIN 21
#define LK_PW (LK_WIN + 3)      /* source patch: window + 1 for the bilinear neighbour + 1 on each side for Scharr */
#define LK_DW (LK_WIN + 1)      /* derivative / second-image patch */
#define LK_NPX (LK_WIN * LK_WIN)
#define LK_PER ((LK_NPX + 63) / 64)
#define LK_MAX_LEVELS 6
#define LK_JM 6                 /* margin of the cached second-image region: most iterations move the window by < 1 px */
#define LK_JS (LK_DW + 2 * LK_JM)

struct LkLevels {
    const uint8_t* prev[LK_MAX_LEVELS];
    const uint8_t* next[LK_MAX_LEVELS];
    int w[LK_MAX_LEVELS], h[LK_MAX_LEVELS], stride[LK_MAX_LEVELS];
    size_t pitch[LK_MAX_LEVELS]; /* bytes between the same level of consecutive pairs (batched form) */
    int top;                     /* coarsest level used */
    int pts_pitch;               /* points per pair slot (batched form) */
};

__device__ __forceinline__ int lk_refl(int p, int n) { /* BORDER_REFLECT_101 for |offset| < n */
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

__global__ void __launch_bounds__(256)
k_pyr_down(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t spitch, uint8_t* __restrict__ dst, int dw, int dh,
           int dstride, size_t dpitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    src += (size_t)blockIdx.z * spitch; /* blockIdx.z = image of the batch */
    dst += (size_t)blockIdx.z * dpitch;
    int col[5];
#pragma unroll
    for (int i = 0; i < 5; i++) col[i] = lk_refl(2 * x + i - 2, sw);
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t* row = src + (size_t)lk_refl(2 * y + j - 2, sh) * sstride;
        const int kj = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
        sum += kj * ((int)row[col[0]] + 4 * (int)row[col[1]] + 6 * (int)row[col[2]] + 4 * (int)row[col[3]] + (int)row[col[4]]);
    }
    dst[(size_t)y * dstride + x] = (uint8_t)((sum + 128) >> 8);
}

/* Wave sum of one int per lane on the DPP data path (row shifts inside the 16-lane rows, then row_bcast 15 / 31 into
 * lane 63): six dependent VALU adds instead of six LDS-permute round trips. Integer, so any order gives the same sum. */
__device__ __forceinline__ int lk_dpp_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); /* row_shr:8 -> lane 15 of a row holds its sum */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1, 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2, 3 */
    return __builtin_amdgcn_readlane(v, 63);
}
/* per-lane 32-bit partial sums (|a| < 2^28) whose wave total needs 64 bits: 16-bit halves summed separately */
__device__ __forceinline__ long long lk_wave_sum(int a) {
    const int lo = lk_dpp_sum(a & 0xffff), hi = lk_dpp_sum(a >> 16);
    return ((long long)hi << 16) + lo;
}
__device__ __forceinline__ int lk_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
struct LkW { int w00, w01, w10, w11; };
__device__ __forceinline__ LkW lk_weights(float a, float b) {
    LkW W;
    W.w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    W.w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    W.w10 = __float2int_rn((1.f - a) * b * 16384.f);
    W.w11 = 16384 - W.w00 - W.w01 - W.w10;
    return W;
}

__global__ void __launch_bounds__(64)
k_lk_track(LkLevels L, const float* __restrict__ prev_pts, const int32_t* __restrict__ counts, int n, float* __restrict__ next_pts,
           uint8_t* __restrict__ status, float* __restrict__ err) {
    __shared__ int Ip[LK_PW * LK_PW];        /* source patch, position (x, y) of the window at [(y + 1) * LK_PW + x + 1] */
    __shared__ int dX[LK_DW * LK_DW], dY[LK_DW * LK_DW];
    __shared__ int Jc[LK_JS * LK_JS]; /* cached region of the second image around the current window */
    const int pair = blockIdx.y, lane = threadIdx.x;
    if ((int)blockIdx.x >= (counts ? min(counts[pair], n) : n)) return;
    const int i = pair * L.pts_pitch + blockIdx.x; /* record index */
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (LK_WIN - 1) * 0.5f;
    const float ptx = prev_pts[2 * i], pty = prev_pts[2 * i + 1];
    float outx = 0.f, outy = 0.f, errv = 0.f;
    int st = 1;
    /* this lane's window samples: p = lane + 64 k */
    int wy[LK_PER], wx[LK_PER];
#pragma unroll
    for (int k = 0; k < LK_PER; k++) { const int p = lane + 64 * k; wy[k] = p / LK_WIN; wx[k] = p - wy[k] * LK_WIN; }
    for (int level = L.top; level >= 0; level--) {
        const uint8_t* I = L.prev[level] + (size_t)pair * L.pitch[level];
        const uint8_t* J = L.next[level] + (size_t)pair * L.pitch[level];
        const int w = L.w[level], h = L.h[level], stride = L.stride[level];
        const float sc = (float)(1. / (1 << level));
        float px = ptx * sc, py = pty * sc, nx, ny;
        if (level == L.top) { nx = px; ny = py; }
        else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;
        px -= half; py -= half;
        const int ix = (int)floorf(px), iy = (int)floorf(py);
        if (ix < -LK_WIN || ix >= w || iy < -LK_WIN || iy >= h) { /* wave-uniform */
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }
        __syncthreads(); /* the previous level is done with the LDS patches */
        for (int t = lane; t < LK_PW * LK_PW; t += 64) {
            const int yy = t / LK_PW, xx = t - yy * LK_PW;
            Ip[t] = I[(size_t)lk_refl(iy - 1 + yy, h) * stride + lk_refl(ix - 1 + xx, w)];
        }
        __syncthreads();
        for (int t = lane; t < LK_DW * LK_DW; t += 64) {
            const int yy = t / LK_DW, xx = t - yy * LK_DW;
            const int X = ix + xx, Y = iy + yy;
            int gx = 0, gy = 0;
            if (X >= 0 && X < w && Y >= 0 && Y < h) {
                const int* c = Ip + (yy + 1) * LK_PW + xx + 1;
                const int a00 = c[-LK_PW - 1], a01 = c[-LK_PW], a02 = c[-LK_PW + 1], a10 = c[-1], a12 = c[1];
                const int a20 = c[LK_PW - 1], a21 = c[LK_PW], a22 = c[LK_PW + 1];
                gx = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
                gy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
            }
            dX[t] = gx; dY[t] = gy;
        }
        __syncthreads();
        LkW W = lk_weights(px - (float)ix, py - (float)iy);
        int Iw[LK_PER], Ix[LK_PER], Iy[LK_PER];
        int A11l = 0, A12l = 0, A22l = 0; /* |Ix|, |Iy| <= 4080: 7 products per lane fit 32 bits */
#pragma unroll
        for (int k = 0; k < LK_PER; k++) {
            Iw[k] = Ix[k] = Iy[k] = 0;
            if (lane + 64 * k < LK_NPX) {
                const int* c = Ip + (wy[k] + 1) * LK_PW + wx[k] + 1;
                const int d = wy[k] * LK_DW + wx[k];
                Iw[k] = lk_descale(c[0] * W.w00 + c[1] * W.w01 + c[LK_PW] * W.w10 + c[LK_PW + 1] * W.w11, 9);
                Ix[k] = lk_descale(dX[d] * W.w00 + dX[d + 1] * W.w01 + dX[d + LK_DW] * W.w10 + dX[d + LK_DW + 1] * W.w11, 14);
                Iy[k] = lk_descale(dY[d] * W.w00 + dY[d + 1] * W.w01 + dY[d + LK_DW] * W.w10 + dY[d + LK_DW + 1] * W.w11, 14);
                A11l += Ix[k] * Ix[k]; A12l += Ix[k] * Iy[k]; A22l += Iy[k] * Iy[k];
            }
        }
        const long long A11 = lk_wave_sum(A11l), A12 = lk_wave_sum(A12l), A22 = lk_wave_sum(A22l);
        const float a11 = (float)A11 * FLT_SCALE, a12 = (float)A12 * FLT_SCALE, a22 = (float)A22 * FLT_SCALE;
        float D = a11 * a22 - a12 * a12;
        const float minEig = (a22 + a11 - sqrtf((a11 - a22) * (a11 - a22) + 4.f * a12 * a12)) / (float)(2 * LK_WIN * LK_WIN);
        if (minEig < 1e-4f || D < 1.1920929e-07f /* FLT_EPSILON */) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        /* the second image's patch at (jx, jy) -> per-lane residuals against the stored window */
        /* residuals of the window at (jx, jy) of the second image against the stored window. The LK_JS x LK_JS region around
         * the window is staged in LDS and reused while the window stays inside it: a reload per iteration was one
         * global-memory latency and two barriers on the critical path of every iteration (5.9 -> 4.9 ms per 128 k points).
         * Its margin may reach past a single reflection on small levels: clamped there, never read back. */
        int cx0 = 0, cy0 = 0;
        bool cached = false;
        auto residuals = [&](int jx, int jy, const LkW& Wj, int& s1, int& s2, int& sabs) {
            if (!cached || jx < cx0 || jx + LK_DW > cx0 + LK_JS || jy < cy0 || jy + LK_DW > cy0 + LK_JS) { /* wave-uniform */
                cx0 = jx - LK_JM; cy0 = jy - LK_JM;
                cached = true;
                __syncthreads();
                for (int t = lane; t < LK_JS * LK_JS; t += 64) {
                    const int yy = t / LK_JS, xx = t - yy * LK_JS;
                    const int yr = min(max(lk_refl(cy0 + yy, h), 0), h - 1), xr = min(max(lk_refl(cx0 + xx, w), 0), w - 1);
                    Jc[t] = J[(size_t)yr * stride + xr];
                }
                __syncthreads();
            }
            const int* Jp = Jc + (jy - cy0) * LK_JS + (jx - cx0);
            /* per lane the 7 products fit 32 bits (|diff| <= 8160, |Ix| <= 4080); the wave sum needs 64 */
            int a1 = 0, a2 = 0, ab = 0;
#pragma unroll
            for (int k = 0; k < LK_PER; k++)
                if (lane + 64 * k < LK_NPX) {
                    const int* c = Jp + wy[k] * LK_JS + wx[k];
                    const int diff = lk_descale(c[0] * Wj.w00 + c[1] * Wj.w01 + c[LK_JS] * Wj.w10 + c[LK_JS + 1] * Wj.w11, 9) - Iw[k];
                    a1 += diff * Ix[k]; a2 += diff * Iy[k];
                    ab += diff < 0 ? -diff : diff;
                }
            s1 = a1; s2 = a2; sabs = ab;
        };
        for (int j = 0; j < 30; j++) {
            const int jx = (int)floorf(nx), jy = (int)floorf(ny);
            if (jx < -LK_WIN || jx >= w || jy < -LK_WIN || jy >= h) {
                if (level == 0) st = 0;
                break;
            }
            const LkW Wj = lk_weights(nx - (float)jx, ny - (float)jy);
            int r1, r2, ra;
            residuals(jx, jy, Wj, r1, r2, ra);
            const long long B1 = lk_wave_sum(r1), B2 = lk_wave_sum(r2);
            const float b1 = (float)B1 * FLT_SCALE, b2 = (float)B2 * FLT_SCALE;
            const float dx = (a12 * b2 - a22 * b1) * D, dy = (a12 * b1 - a11 * b2) * D;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            if ((double)dx * dx + (double)dy * dy <= 0.01 * 0.01) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && level == 0) { /* L1 error at the final position */
            const float fx = outx - half, fy = outy - half;
            const int jx = (int)floorf(fx), jy = (int)floorf(fy);
            if (jx < -LK_WIN || jx >= w || jy < -LK_WIN || jy >= h) st = 0;
            else {
                const LkW Wj = lk_weights(fx - (float)jx, fy - (float)jy);
                int r1, r2, ra;
                residuals(jx, jy, Wj, r1, r2, ra);
                const long long E = lk_dpp_sum(ra); /* <= 64 * 7 * 8160 */
                errv = (float)E / (float)(32 * LK_WIN * LK_WIN);
            }
        }
    }
    if (lane == 0) {
        next_pts[2 * i] = outx; next_pts[2 * i + 1] = outy;
        status[i] = (uint8_t)st;
        if (err) err[i] = errv;
    }
}

/* Device entry: npairs image pairs (pair p at prev / next + p * image_pitch bytes), points of pair p at
 * prev_pts + p * pts_pitch (x, y) records, counts[p] of them (counts nullable: n each); `work` holds the pyramids above
 * level 0 (tbk_lk_work_bytes). Returns the coarsest level used through *top_level. */
size_t tbk_lk_work_bytes(int w, int h, int max_level, int npairs) {
    size_t total = 0;
    for (int l = 1; l <= max_level && l < LK_MAX_LEVELS; l++) {
        w = (w + 1) / 2; h = (h + 1) / 2;
        total += 2 * (size_t)npairs * (((size_t)w * h + 255) & ~(size_t)255);
    }
    return total + 256;
}

int tbk_lk_track(tb_ctx* ctx, int npairs, const uint8_t* d_prev, const uint8_t* d_next, int w, int h, int stride, size_t image_pitch,
                 const float* d_prev_pts, const int32_t* d_counts, int n, int pts_pitch, int win, int max_level, float* d_next_pts,
                 uint8_t* d_status, float* d_err, void* d_work, int* top_level) {
    if (win != LK_WIN) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: window %d (this build: 21, the reference's)", win);
    if (max_level < 0 || max_level >= LK_MAX_LEVELS) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: max_level %d (0..5)", max_level);
    if (w <= win || h <= win) return tb_fail(ctx, TB_EINVAL, "optical flow: image not larger than the window");
    if (npairs > 65535) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: more than 65535 pairs per call");
    LkLevels L;
    memset(&L, 0, sizeof L);
    L.prev[0] = d_prev; L.next[0] = d_next; L.w[0] = w; L.h[0] = h; L.stride[0] = stride; L.pitch[0] = image_pitch;
    L.top = 0;
    L.pts_pitch = pts_pitch;
    uint8_t* p = (uint8_t*)d_work;
    for (int l = 1; l <= max_level; l++) { /* cv::buildOpticalFlowPyramid stops before a level no larger than the window */
        const int lw = (L.w[l - 1] + 1) / 2, lh = (L.h[l - 1] + 1) / 2;
        if (lw <= win || lh <= win) break;
        const size_t bytes = ((size_t)lw * lh + 255) & ~(size_t)255;
        uint8_t* a = p; p += bytes * npairs;
        uint8_t* b = p; p += bytes * npairs;
        dim3 grid((lw + 63) / 64, (lh + 3) / 4, npairs);
        tb_prof_begin(ctx, "k_pyr_down");
        hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, ctx->stream, L.prev[l - 1], L.w[l - 1], L.h[l - 1], L.stride[l - 1],
                           L.pitch[l - 1], a, lw, lh, lw, bytes);
        hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, ctx->stream, L.next[l - 1], L.w[l - 1], L.h[l - 1], L.stride[l - 1],
                           L.pitch[l - 1], b, lw, lh, lw, bytes);
        tb_prof_end(ctx);
        L.prev[l] = a; L.next[l] = b; L.w[l] = lw; L.h[l] = lh; L.stride[l] = lw; L.pitch[l] = bytes;
        L.top = l;
    }
    if (top_level) *top_level = L.top;
    if (n > 0 && npairs > 0) {
        tb_prof_begin(ctx, "k_lk_track");
        hipLaunchKernelGGL(k_lk_track, dim3(n, npairs), dim3(64), 0, ctx->stream, L, d_prev_pts, d_counts, n, d_next_pts, d_status, d_err);
        tb_prof_end(ctx);
    }
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- Frame::Equalize, src/types/Frame.cpp:453-458: cv::createCLAHE(3.0, Size(8, 8))->apply(level 0).
 * cv::CLAHE is OpenCV 3.3 (not in the reference tree): restated, parity unpinned (DESIGN.md section 2 lists what: tile
 * histograms over the reflect-101 extended image, clip at max(1, (int)(clip * area / 256)), excess / 256 to every bin
 * plus one to each of the first excess % 256 bins, LUT = saturate(cvRound(cumsum * 255 / area))).
 *   k_clahe_lut   : one workgroup per tile -- LDS histogram (integer atomics), clip, redistribute, prefix sum, LUT
 *   k_clahe_apply : thread per pixel -- bilinear blend of the four neighbouring tiles' LUT entries: (l11 xa1 + l12 xa) ya1 +
 *                   (l21 xa1 + l22 xa) ya in float, no contraction */
__global__ void __launch_bounds__(256)
k_clahe_lut(const uint8_t* __restrict__ src, int w, int h, int stride, size_t spitch, int tw, int th, int clip, float lutScale,
            uint8_t* __restrict__ lut) {
    __shared__ int hist[256];
    __shared__ int tmp[8];
    const int tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
    src += (size_t)blockIdx.z * spitch;                               /* blockIdx.z = image of the batch */
    lut += (size_t)blockIdx.z * gridDim.x * gridDim.y * 256;
    hist[tid] = 0;
    __syncthreads();
    for (int p = tid; p < tw * th; p += 256) {
        const int yy = p / tw, xx = p - yy * tw;
        atomicAdd(&hist[src[(size_t)lk_refl(ty * th + yy, h) * stride + lk_refl(tx * tw + xx, w)]], 1);
    }
    __syncthreads();
    int mine = hist[tid];
    if (clip > 0) {
        const int over = mine > clip ? mine - clip : 0;
        mine -= over;
        int tot = tb_wave_sum(over);
        if ((tid & 63) == 0) tmp[tid >> 6] = tot;
        __syncthreads();
        const int clipped = tmp[0] + tmp[1] + tmp[2] + tmp[3];
        const int batch = clipped / 256, residual = clipped - batch * 256;
        mine += batch + (tid < residual ? 1 : 0);
        __syncthreads();
    }
    hist[tid] = mine;
    __syncthreads();
    const int before = tb_block_excl_scan(hist, 256, tmp); /* in place; returns the total */
    (void)before;
    const int sum = hist[tid] + mine;                      /* inclusive */
    const int r = __float2int_rn((float)sum * lutScale);
    lut[((size_t)ty * gridDim.x + tx) * 256 + tid] = (uint8_t)min(max(r, 0), 255);
}

__global__ void __launch_bounds__(256)
k_clahe_apply(const uint8_t* __restrict__ src, int w, int h, int stride, size_t spitch, int tiles_x, int tiles_y, float inv_tw,
              float inv_th, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst, int dstride, size_t dpitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    src += (size_t)blockIdx.z * spitch;
    dst += (size_t)blockIdx.z * dpitch;
    lut += (size_t)blockIdx.z * tiles_x * tiles_y * 256;
    const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya, xa = txf - (float)tx1, xa1 = 1.0f - xa;
    const int ty2 = min(ty1 + 1, tiles_y - 1), tx2 = min(tx1 + 1, tiles_x - 1);
    ty1 = max(ty1, 0); tx1 = max(tx1, 0);
    const int v = src[(size_t)y * stride + x];
    const float l11 = lut[((size_t)ty1 * tiles_x + tx1) * 256 + v], l12 = lut[((size_t)ty1 * tiles_x + tx2) * 256 + v];
    const float l21 = lut[((size_t)ty2 * tiles_x + tx1) * 256 + v], l22 = lut[((size_t)ty2 * tiles_x + tx2) * 256 + v];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    dst[(size_t)y * dstride + x] = (uint8_t)min(max(__float2int_rn(res), 0), 255);
}

int tbk_clahe(tb_ctx* ctx, int nimg, const uint8_t* d_src, int w, int h, int stride, size_t spitch, double clip_limit, int tiles_x,
              int tiles_y, uint8_t* d_dst, int dstride, size_t dpitch, uint8_t* d_lut) {
    if (tiles_x < 1 || tiles_y < 1 || tiles_x * tiles_y > 65535 || nimg < 1 || nimg > 65535)
        return tb_fail(ctx, TB_EINVAL, "CLAHE: bad tile grid / image count");
    int ew = w, eh = h;
    if (w % tiles_x || h % tiles_y) { ew = w + (tiles_x - w % tiles_x); eh = h + (tiles_y - h % tiles_y); }
    if (ew - w >= w || eh - h >= h) return tb_fail(ctx, TB_EINVAL, "CLAHE: image smaller than its tile grid");
    const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
    int clip = 0;
    if (clip_limit > 0.0) { clip = (int)(clip_limit * area / 256); if (clip < 1) clip = 1; }
    tb_prof_begin(ctx, "k_clahe");
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x, tiles_y, nimg), dim3(256), 0, ctx->stream, d_src, w, h, stride, spitch, tw, th, clip,
                       (float)255 / (float)area, d_lut);
    hipLaunchKernelGGL(k_clahe_apply, dim3((w + 63) / 64, (h + 3) / 4, nimg), dim3(256), 0, ctx->stream, d_src, w, h, stride, spitch,
                       tiles_x, tiles_y, 1.0f / (float)tw, 1.0f / (float)th, d_lut, d_dst, dstride, dpitch);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* Matcher::searchByOPFlow's bookkeeping on the device (matcher.cpp:746-766): a point stays matched when the tracker kept it
 * and its truncated position lies in F1's frame (CameraModel.h:33-39); the matches DMatch(i, i) come out in index order.
 * One wavefront per pair, ballot-ranked compaction. status is updated like the reference's vector. */
__global__ void __launch_bounds__(64)
k_flow_accept(const float* __restrict__ cur, uint8_t* __restrict__ status, const int32_t* __restrict__ counts, int pts_pitch, int width,
              int height, tb_match* __restrict__ out, int cap, int32_t* __restrict__ out_counts) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int n = counts ? min(counts[pair], pts_pitch) : pts_pitch;
    const float* c = cur + (size_t)pair * pts_pitch * 2;
    uint8_t* st = status + (size_t)pair * pts_pitch;
    tb_match* o = out + (size_t)pair * cap;
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool keep = false;
        if (i < n && st[i]) {
            const float x = c[2 * i], y = c[2 * i + 1];
            if (fabsf(x) < 2147483648.f && fabsf(y) < 2147483648.f) { /* out-of-range converts to INT_MIN on x86: not in frame */
                const int u = (int)x, v = (int)y;
                keep = u >= 0 && u < (int)((float)width * 1.f) && v >= 0 && v < (int)((float)height * 1.f);
            }
            if (!keep) st[i] = 0;
        }
        const unsigned long long b = __ballot(keep);
        const int at = m + __popcll(b & ((1ull << lane) - 1ull));
        if (keep && at < cap) { tb_match r; r.queryIdx = i; r.trainIdx = i; r.imgIdx = -1; r.distance = 3.402823466e+38f; o[at] = r; }
        m += __popcll(b);
    }
    if (lane == 0) out_counts[pair] = min(m, cap);
}

int tbk_flow_accept(tb_ctx* ctx, int npairs, const float* d_cur, uint8_t* d_status, const int32_t* d_counts, int pts_pitch, int width,
                    int height, tb_match* d_out, int cap, int32_t* d_out_counts) {
    tb_prof_begin(ctx, "k_flow_accept");
    hipLaunchKernelGGL(k_flow_accept, dim3(npairs), dim3(64), 0, ctx->stream, d_cur, d_status, d_counts, pts_pitch, width, height, d_out,
                       cap, d_out_counts);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
