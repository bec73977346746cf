/* a4 / a11 -- FAST corner detection + score + 3x3 non-max suppression.
 *
 * Reference: ComputeKeyPointsOctTree cell loop calling cv::FAST(ROI, initTh, nms) with the minTH retry
 * (src/extractors/ORBextractor.cpp:747-804) and FASTExtractor's fast_corner_detect_10 /
 * fast_corner_score_10 / fast_nonmax_3x3 (src/extractors/FASTextractor.cpp:36-51).
 *
 * Formulation (hard part 3): the FAST score S(p) = largest t for which p is still a
 * corner is threshold independent, a pixel is a corner at t iff S(p) >= t, and 3x3 NMS on S restricted
 * to the scanned region equals OpenCV's / fast_lib's NMS.  So one pass scores a region once and the
 * per-cell adaptive threshold becomes a block-wide reduction:
 *   emit {p : local strict maximum, S(p) >= th},  th = initTh if that set is non-empty else minTh.
 *
 * One workgroup per work item (a 30-px cell ROI, or a 58x58 tile in whole-image mode). The ROI
 * (<= 66x66 u8) is staged in LDS with row-contiguous loads; phase A runs the 16-compare arc test on
 * every scanned pixel and ballot-compacts the few survivors into an LDS list; phases B-D (exact score,
 * NMS, emit) touch only that list.  Roofline: HBM (1 read per pixel) but in practice the
 * 16-tap ring test makes phase A LDS/VALU-issue bound; see DESIGN.md.
 */
#include "tb_internal.h"
#include "tb_device.h"

#define FT_TS 68   /* LDS tile row stride (bytes) */
#define FT_TH 66   /* max ROI rows / cols */
#define FT_LIST 3600

struct FastRegion {
    int lx0, ly0, lx1, ly1; /* loaded pixels */
    int sx0, sy0, sx1, sy1; /* scored pixels (scan region) */
    int ox0, oy0, ox1, oy1; /* output pixels */
};

__device__ __forceinline__ void ft_ring(const uint8_t* c, int r[16]) {
    r[0] = c[3 * FT_TS];       r[1] = c[3 * FT_TS + 1];  r[2] = c[2 * FT_TS + 2];  r[3] = c[FT_TS + 3];
    r[4] = c[3];               r[5] = c[-FT_TS + 3];     r[6] = c[-2 * FT_TS + 2]; r[7] = c[-3 * FT_TS + 1];
    r[8] = c[-3 * FT_TS];      r[9] = c[-3 * FT_TS - 1]; r[10] = c[-2 * FT_TS - 2]; r[11] = c[-FT_TS - 3];
    r[12] = c[-3];             r[13] = c[FT_TS - 3];     r[14] = c[2 * FT_TS - 2]; r[15] = c[3 * FT_TS - 1];
}

template <int ARC>
__device__ __forceinline__ bool ft_is_corner(const uint8_t* c, int th) {
    int r[16];
    ft_ring(c, r);
    const int v = c[0];
    const int hi = v + th, lo = v - th;
    uint32_t B = 0, D = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        B |= (uint32_t)(r[k] > hi) << k;
        D |= (uint32_t)(r[k] < lo) << k;
    }
    B |= B << 16;
    D |= D << 16;
    uint32_t xb = B & (B >> 1), xd = D & (D >> 1);
    xb &= xb >> 2; xd &= xd >> 2;
    xb &= xb >> 4; xd &= xd >> 4;
    xb &= B >> 8;  xd &= D >> 8;
    if (ARC == 10) { xb &= B >> 9; xd &= D >> 9; }
    return ((xb | xd) & 0xffffu) != 0;
}

/* exact score: max over the 16 arcs of min |d| with a common sign, minus 1 */
template <int ARC>
__device__ __forceinline__ int ft_score(const uint8_t* c) {
    int r[16];
    ft_ring(c, r);
    const int v = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - r[k];
    int best = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
#pragma unroll
        for (int i = 1; i < ARC; i++) {
            mn = min(mn, d[(k + i) & 15]);
            mx = max(mx, d[(k + i) & 15]);
        }
        best = max(best, max(mn, -mx));
    }
    return best - 1;
}

/* Shared body. Emits packed records score<<24 | (y-oy_bias)<<12 | (x-ox_bias) into out[] through a
 * wave-aggregated atomic on *count. two_th: cell mode (initTh with minTh retry). */
template <int ARC>
__device__ __forceinline__ void ft_process(const uint8_t* __restrict__ img, int stride, const FastRegion R,
                                           int th_hi, int th_lo, bool two_th, bool nms, int ox_bias, int oy_bias,
                                           uint32_t* __restrict__ out, int cap, int* __restrict__ count) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[FT_TH * FT_TS];
    __shared__ __attribute__((aligned(16))) uint8_t sc[FT_TH * FT_TS];
    __shared__ uint16_t clist[FT_LIST];
    __shared__ int nlist, any_hi;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lw = R.lx1 - R.lx0, lh = R.ly1 - R.ly0;
    if (tid == 0) { nlist = 0; any_hi = 0; }
    /* stage pixels (row-contiguous) and clear the score map */
    for (int i = tid; i < lh * FT_TS; i += T) {
        const int y = i / FT_TS, x = i - y * FT_TS;
        tile[i] = (x < lw) ? img[(size_t)(R.ly0 + y) * stride + R.lx0 + x] : 0;
        sc[i] = 0;
    }
    __syncthreads();
    const int sw = R.sx1 - R.sx0, sh = R.sy1 - R.sy0;
    const int npx = sw > 0 && sh > 0 ? sw * sh : 0;
    const int thA = two_th ? min(th_hi, th_lo) : th_hi;
    /* phase A: arc test on every scanned pixel, compact survivors */
    for (int base = 0; base < npx; base += T) {
        const int i = base + tid;
        bool c = false;
        int idx = 0;
        if (i < npx) {
            const int y = i / sw, x = i - y * sw;
            idx = (R.sy0 - R.ly0 + y) * FT_TS + (R.sx0 - R.lx0 + x);
            c = ft_is_corner<ARC>(tile + idx, thA);
        }
        const unsigned long long m = __ballot(c);
        int wbase = 0;
        if (tb_lane() == 0 && m) wbase = atomicAdd(&nlist, __popcll(m));
        wbase = __shfl(wbase, 0, TB_WAVE);
        if (c) {
            const int slot = wbase + __popcll(m & ((1ull << tb_lane()) - 1));
            if (slot < FT_LIST) clist[slot] = (uint16_t)idx;
        }
    }
    __syncthreads();
    const int n = min(nlist, FT_LIST);
    /* phase B: exact scores of the survivors */
    for (int i = tid; i < n; i += T) {
        const int idx = clist[i];
        int s = ft_score<ARC>(tile + idx);
        sc[idx] = (uint8_t)min(max(s, 0), 255);
    }
    __syncthreads();
    /* phase C: 3x3 strict maximum; neighbours outside the scored region hold 0 */
    const int oxa = R.ox0 - R.lx0, oxb = R.ox1 - R.lx0, oya = R.oy0 - R.ly0, oyb = R.oy1 - R.ly0;
    int hit_hi = 0;
    for (int i = tid; i < n; i += T) {
        const int idx = clist[i];
        const int y = idx / FT_TS, x = idx - y * FT_TS;
        const int s = sc[idx];
        bool keep = (x >= oxa && x < oxb && y >= oya && y < oyb) && s > 0;
        if (keep && nms) {
            keep = s > sc[idx - 1] && s > sc[idx + 1] && s > sc[idx - FT_TS - 1] && s > sc[idx - FT_TS] &&
                   s > sc[idx - FT_TS + 1] && s > sc[idx + FT_TS - 1] && s > sc[idx + FT_TS] && s > sc[idx + FT_TS + 1];
        }
        if (!keep) clist[i] = 0xffff;
        else if (s >= th_hi) hit_hi = 1;
    }
    if (two_th) {
        if (hit_hi) any_hi = 1; /* benign race: all writers store 1 */
    }
    __syncthreads();
    const int th = two_th ? (any_hi ? th_hi : th_lo) : th_hi;
    /* phase D: emit */
    for (int base = 0; base < n; base += T) {
        const int i = base + tid;
        bool e = false;
        uint32_t rec = 0;
        if (i < n) {
            const int idx = clist[i];
            if (idx != 0xffff) {
                const int s = sc[idx];
                if (s >= th) {
                    const int y = idx / FT_TS, x = idx - y * FT_TS;
                    e = true;
                    rec = ((uint32_t)s << 24) | ((uint32_t)(R.ly0 + y - oy_bias) << 12) | (uint32_t)(R.lx0 + x - ox_bias);
                }
            }
        }
        const unsigned long long m = __ballot(e);
        int wbase = 0;
        if (tb_lane() == 0 && m) wbase = atomicAdd(count, __popcll(m));
        wbase = __shfl(wbase, 0, TB_WAVE);
        if (e) {
            const int slot = wbase + __popcll(m & ((1ull << tb_lane()) - 1));
            if (slot < cap) out[slot] = rec;
        }
    }
}

/* ---- cell mode (a4): ONE WORKGROUP PER BLOCK OF CELLS (up to 4 x 2 cells of ~30 px, FastBlock in tb_internal.h).
 *
 * The reference runs cv::FAST on every 30-px cell's ROI separately (ORBextractor.cpp:765-797). The cells' SCAN regions
 * tile the level, their ROIs overlap by 6 px, and three things are per cell: the 3x3 non-max suppression sees only
 * scores of its own cell's scan region (outside counts 0), the threshold is initTh if that leaves a corner after NMS and
 * minTh otherwise, and nothing else. So a block of cells is processed as ONE image tile with the cell structure applied
 * only where it matters:
 *
 *   stage 0  the block's ROI (<= 130 x 66 px) -> LDS with 16-byte row-segment loads, once: the pixels P and a
 *            QUANTISED copy Q = P >> 2 (6 bits per byte).
 *   stage 1  cardinal pre-test on every scanned pixel, 16 pixels per lane, byte-parallel on Q: with q = P >> 2 and
 *            t'' = (t + 1) >> 2,  X > V + t  implies  qX - qV >= t''  and  X < V - t  implies  qV - qX >= t''
 *            (floor((a + b) / 4) >= floor(a / 4) + floor(b / 4)), and both are ONE 32-bit add / subtract for four
 *            pixels, the verdict in bit 7 of each byte, no carries between bytes:
 *                bright: (qX + (128 - t'' - qV)) & 0x80        dark: ((128 - t'' + qV) - qX) & 0x80
 *            A 9-arc holds two adjacent cardinals, so (b0|b8)&(b4|b12) | (d0|d8)&(d4|d12) is necessary; the quantised
 *            form admits a few more pixels than the exact one and never loses a corner: 18 vector instructions per 4
 *            pixels instead of 65 for the exact 16-bit SWAR form. Lanes with a hit store one 8-byte record.
 *   stage 2  Q is dead: its LDS becomes the (zeroed) score map. The records are expanded into a pixel list (wave
 *            prefix sum on the DPP path) and every listed pixel gets its exact FAST score, two pixels per lane in packed
 *            16-bit halves, the batches dealt round-robin to the four wavefronts; corners (score >= initTh) go into the
 *            score map and the corner list.
 *   stage 3  3x3 strict maximum per corner, neighbours of another cell masked by per-column / per-row cell tables;
 *            one bit per cell records "a corner survived".
 *   retry    cells without a survivor are redone at minTh by one wavefront each (exact cardinal test from P, same
 *            score routine); their pass-1 corners lost the NMS and lose it again (the scores that beat them are
 *            still in the map), so only the new corners (minTh <= score < initTh) are appended and suppressed.
 *   emit     every wavefront counts the survivors in its share of the list, reserves their range with one global atomic
 *            and writes the records.
 *
 * LDS: P 10.6 KB + Q / score map 10.6 KB + lists 8.7 KB = 30.2 KB -> five blocks (20 wavefronts) per CU. The pixel and
 * corner lists are sized for ordinary images (3072 listed pixels, 1360 corners per block: six times / ten times the
 * average of textured frames); a block that overflows them -- dense noise, thresholds near zero -- is redone by
 * fb_dense(): scores of all scanned pixels straight into the map, then per cell threshold choice, NMS and emission by
 * scanning the map. No list, any density, same results (tests force it with TB_FAST_DENSE=1). */
#ifndef FB_NW
#define FB_NW 4                                    /* wavefronts per block */
#endif
#define FB_NT (64 * FB_NW)
#define FB_SEGS (FB_S / 16)
#define FB_NSEG_ALL (FB_SEGS * FB_TH)              /* 680 16-byte segments */
#define FB_PX (FB_S * FB_TH)                       /* 10880 */
#ifndef FB_LIST_CAP
#define FB_LIST_CAP 3072
#endif
#define FB_REC_CAP FB_NSEG_ALL                     /* one 4-byte record per stage-1 work item: cannot overflow */
#define FB_CL_CAP (FB_REC_CAP * 2)                 /* the corner list takes over the records' LDS */
#define FB_OFF_R FB_PX                             /* Q, then the score map */
#define FB_OFF_LIST (2 * FB_PX)
#define FB_OFF_REC (FB_OFF_LIST + FB_LIST_CAP * 2)
#define FB_OFF_CL FB_OFF_REC
#define FB_OFF_COL (FB_OFF_REC + FB_REC_CAP * 4)
#define FB_OFF_ROW (FB_OFF_COL + FB_S)
#define FB_OFF_MISC (FB_OFF_ROW + 80)
#ifndef FB_PAD_LDS
#define FB_PAD_LDS 0
#endif
#define FB_LDS_BYTES (FB_OFF_MISC + 64 + FB_PAD_LDS)
#define FB_RETRY_CAP 512                           /* pixels per retry strip (wave-private lists over LIST) */

typedef _Float16 ft_h2 __attribute__((ext_vector_type(2)));
typedef unsigned ft_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ft_h2 ft_h2_bits(uint32_t u) { return __builtin_bit_cast(ft_h2, u); }
__device__ __forceinline__ ft_h2 ft_min3(ft_h2 a, ft_h2 b, ft_h2 c) { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); }
__device__ __forceinline__ ft_h2 ft_max3(ft_h2 a, ft_h2 b, ft_h2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }

__device__ __forceinline__ void ft_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

/* exact FAST-9 score + 1 of the pixel at tile pointer p (row stride FB_S): max over the 16 arcs of min(d) and of min(-d),
 * d = centre - ring; a pixel is a corner at t iff the result - 1 >= t.
 * BOTH SIGNS RIDE IN ONE REGISTER as two half-precision numbers of the binade [1024, 2048), where one ulp is 1 and the
 * bit pattern 0x6600 + x IS the number 1536 + x: with K = (0x6600 - v) << 16 | (0x6600 + v), one 24-bit multiply-add per
 * ring pixel, r * 0xffff + K, leaves 1536 + d in the low half and 1536 - d in the high half (no borrow: 0x6600 + v >= r).
 * The minimum over a 9-arc is two rounds of gfx950's three-input packed minimum (v_pk_minimum3_f16: 3 x 3), the maximum
 * over the arcs a tree of v_pk_maximum3_f16 -- 16 + 32 + 8 vector instructions per pixel where the 16-bit integer
 * min / max network of two-input instructions took ~105. All operands are positive normal numbers of one binade, so the
 * floating-point order is the integer order of the differences and no rounding ever happens. */
__device__ __forceinline__ int fb_score1(const uint8_t* p) {
    constexpr int S = FB_S, C = 3 * FB_S + 3;
    const uint8_t* q = p - C;                      /* every ring offset non-negative: immediate DS offsets */
    const uint32_t v = q[C];
    const uint32_t K = ((0x6600u - v) << 16) | (0x6600u + v);
    constexpr int off[16] = {C + 3 * S, C + 3 * S + 1, C + 2 * S + 2, C + S + 3, C + 3, C - S + 3, C - 2 * S + 2, C - 3 * S + 1,
                             C - 3 * S, C - 3 * S - 1, C - 2 * S - 2, C - S - 3, C - 3, C + S - 3, C + 2 * S - 2, C + 3 * S - 1};
    ft_h2 e[16], m3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = ft_h2_bits(__umul24((uint32_t)q[off[k]], 0xffffu) + K);
#pragma unroll
    for (int k = 0; k < 16; k++) m3[k] = ft_min3(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
    ft_h2 m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = ft_min3(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);   /* min over the 9-arc starting at k */
    const ft_h2 a0 = ft_max3(m9[0], m9[1], m9[2]), a1 = ft_max3(m9[3], m9[4], m9[5]), a2 = ft_max3(m9[6], m9[7], m9[8]);
    const ft_h2 a3 = ft_max3(m9[9], m9[10], m9[11]), a4 = ft_max3(m9[12], m9[13], m9[14]);
    const ft_h2 b = ft_max3(ft_max3(a0, a1, a2), ft_max3(a3, a4, m9[15]), m9[15]);
    const uint32_t bits = __builtin_bit_cast(uint32_t, b);
    return (int)max(bits & 0xffffu, bits >> 16) - 0x6600;
}

/* exact cardinal pre-test at threshold th (one pixel per lane; retry and dense paths) */
__device__ __forceinline__ bool fb_cardinal(const uint8_t* c, int th) {
    const int v = c[0], hi = v + th, lo = v - th;
    const int p0 = c[3 * FB_S], p8 = c[-3 * FB_S], p4 = c[3], p12 = c[-3];
    return (((p0 > hi) | (p8 > hi)) & ((p4 > hi) | (p12 > hi))) | (((p0 < lo) | (p8 < lo)) & ((p4 < lo) | (p12 < lo)));
}

/* idx -> row of the tile: idx / 160 = (idx >> 5) / 5, exact for idx < 160 * 204 */
__device__ __forceinline__ int fb_row(int idx) { return ((idx >> 5) * 205) >> 10; }

/* 3x3 strict maximum inside the corner's own cell: neighbours across a cell boundary (and outside the scan region,
 * where the map is 0 anyway) do not count. colinfo / rowinfo: bits 0-1 cell index, bit 4 first, bit 5 last column /
 * row of its cell. */
__device__ __forceinline__ bool fb_nms_keep(const uint8_t* sc, const uint8_t* colinfo, const uint8_t* rowinfo, int idx,
                                            int* cell) {
    const int ty = fb_row(idx), tx = idx - ty * FB_S;
    const int ci = colinfo[tx], ri = rowinfo[ty];
    *cell = (ri & 3) * FB_MAX_CX + (ci & 3);
    const int s = sc[idx];
    const bool L = !(ci & 16), R = !(ci & 32), U = !(ri & 16), D = !(ri & 32);
    int m = 0;
    m = max(m, L ? (int)sc[idx - 1] : 0);
    m = max(m, R ? (int)sc[idx + 1] : 0);
    m = max(m, U ? (int)sc[idx - FB_S] : 0);
    m = max(m, D ? (int)sc[idx + FB_S] : 0);
    m = max(m, (U && L) ? (int)sc[idx - FB_S - 1] : 0);
    m = max(m, (U && R) ? (int)sc[idx - FB_S + 1] : 0);
    m = max(m, (D && L) ? (int)sc[idx + FB_S - 1] : 0);
    m = max(m, (D && R) ? (int)sc[idx + FB_S + 1] : 0);
    return s > m;
}

struct FbGeom {                       /* wave-uniform geometry of the block's tile */
    int ax0, y0;                      /* image column of tile column 0, image row of tile row 0 */
    int scanX0, scanX1, scanY1;       /* scanned tile columns [scanX0, scanX1), rows [3, scanY1) */
    int wCell, hCell, ncx, ncy;
};

__device__ __forceinline__ uint32_t fb_record(const uint8_t* SC, const FbGeom& G, int idx) {
    const int ty = fb_row(idx), tx = idx - ty * FB_S;
    return ((uint32_t)SC[idx] << 24) | ((uint32_t)(G.y0 + ty - TB_BORDER) << 12) | (uint32_t)(G.ax0 + tx - TB_BORDER);
}

/* wave-level emission of the lanes with e set: one returning global atomic, then the records */
__device__ __forceinline__ void fb_emit_wave(bool e, uint32_t rec, int lane, int* __restrict__ count, uint32_t* __restrict__ out,
                                             int cap) {
    const unsigned long long m = __ballot(e);
    if (!m) return;
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(count, __popcll(m));
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    if (e) {
        const int slot = wbase + __popcll(m & ((1ull << lane) - 1));
        if (slot < cap) out[slot] = rec;
    }
}

/* Any-density path: no lists. Exact scores of ALL scanned pixels at the lower threshold go straight into the (cleared)
 * map; then, cell by cell (one wavefront each), the reference's two tries: survivors of the NMS at initTh, or, if there
 * are none, at minTh. The NMS reads true scores: a neighbour below the threshold in force scores below the corner anyway. */
__device__ __forceinline__ void fb_dense(uint8_t* P, uint8_t* SC, const uint8_t* colinfo, const uint8_t* rowinfo, const FbGeom G,
                                      int init_th, int min_th, int* __restrict__ count, uint32_t* __restrict__ out, int cap) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < FB_PX / 16; i += FB_NT) reinterpret_cast<uint4*>(SC)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int tlow = min(init_th, min_th);
    const int sw = G.scanX1 - G.scanX0;
    for (int y = 3 + wave; y < G.scanY1; y += FB_NW)
        for (int x0 = 0; x0 < sw; x0 += 64) {
            const int x = x0 + lane;
            if (x < sw) {
                const int idx = y * FB_S + G.scanX0 + x;
                if (fb_cardinal(P + idx, tlow)) {
                    const int s = fb_score1(P + idx) - 1;
                    if (s >= tlow && s > 0) SC[idx] = (uint8_t)min(s, 255);
                }
            }
        }
    __syncthreads();
    for (int cellId = wave; cellId < FB_MAX_CX * FB_MAX_CY; cellId += FB_NW) {
        const int cx = cellId & (FB_MAX_CX - 1), cy = cellId / FB_MAX_CX;
        if (cx >= G.ncx || cy >= G.ncy) continue;
        const int X0 = G.scanX0 + cx * G.wCell, X1 = min(X0 + G.wCell, G.scanX1);
        const int Y0 = 3 + cy * G.hCell, Y1 = min(Y0 + G.hCell, G.scanY1);
        int th = init_th;
        for (int pass = 0; pass < 2; pass++) {
            bool any = false;
            for (int y = Y0; y < Y1; y++)
                for (int x0 = X0; x0 < X1; x0 += 64) {
                    const int x = x0 + lane, idx = y * FB_S + x;
                    int cell;
                    const bool k = x < X1 && SC[idx] >= th && SC[idx] > 0 && fb_nms_keep(SC, colinfo, rowinfo, idx, &cell);
                    if (pass == 0) any = any || (__ballot(k) != 0);
                    else fb_emit_wave(k, k ? fb_record(SC, G, idx) : 0u, lane, count, out, cap);
                }
            if (pass == 0 && !any) th = min_th;    /* min_th >= init_th finds nothing either: a subset */
        }
    }
}

/* verdict-word masks: pixel j = 4 d + k of a lane's 16 sits in bit 8 k + 7 - d; ge[n] = pixels j >= n, lt[n] = pixels j < n */
struct FbMask { uint32_t ge[17], lt[17]; };
constexpr FbMask fb_make_mask() {
    FbMask m{};
    for (int n = 0; n <= 16; n++)
        for (int j = 0; j < 16; j++) {
            const uint32_t bit = 1u << (8 * (j & 3) + 7 - (j >> 2));
            if (j >= n) m.ge[n] |= bit;
            if (j < n) m.lt[n] |= bit;
        }
    return m;
}
__constant__ FbMask c_fbmask = fb_make_mask();

#ifdef FB_TIMING   /* debug build (make EXTRA=-DFB_TIMING): per-stage shader clocks of wavefront 0 of every 64th block */
__device__ unsigned long long fb_times[16];
#define FB_T(i) do { const unsigned long long t1_ = __builtin_readcyclecounter(); dt_[i] = t1_ - t0_; t0_ = t1_; } while (0)
extern "C" int tb_debug_fast_times(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_times), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(fb_times), z, sizeof z); }
    return 0;
}
#else
#define FB_T(i) do { } while (0)
#endif

#ifndef FB_MINW
#define FB_MINW 5
#endif
/* FB_NW = 4 wavefronts per block, five blocks (30 KB of LDS each) per CU. Measured with FB_NW = 8 (4 blocks = 32 wavefronts per CU, the
 * hardware limit, instead of 20): 4.32 ms against 3.40 per 1024 images -- every wavefront pays ~300 instructions of fixed cost (block
 * decode, lane maps, stage transitions, emission) whatever its share of the tile, and the kernel is bound by instruction issue, not by
 * latency. */
__global__ void __launch_bounds__(FB_NT, FB_MINW)
k_fast_blocks(PlanGeom g, const uint8_t* __restrict__ slab, const FastBlock* __restrict__ blocks, int nBlocks, int nImages, int by_image,
              uint32_t* __restrict__ cand, int32_t* __restrict__ candCount, int init_th, int min_th, int force_dense) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const P = smem;
    uint8_t* const Q = smem + FB_OFF_R;      /* stage 0-1 */
    uint8_t* const SC = smem + FB_OFF_R;     /* stage 2-  */
    uint16_t* const LIST = reinterpret_cast<uint16_t*>(smem + FB_OFF_LIST);
    uint32_t* const REC = reinterpret_cast<uint32_t*>(smem + FB_OFF_REC);   /* stage 1-2a */
    uint16_t* const CL = reinterpret_cast<uint16_t*>(smem + FB_OFF_CL);     /* stage 2b-: same LDS */
    uint8_t* const colinfo = smem + FB_OFF_COL;
    uint8_t* const rowinfo = smem + FB_OFF_ROW;
    int* const misc = reinterpret_cast<int*>(smem + FB_OFF_MISC);
    /* misc: 0 records, 1 listed pixels, 2 corners, 3 cells with a survivor, 4 overflow */

    /* Workgroup -> (image, block). Consecutive workgroup ids go round the 8 XCDs, each with its own L2.
     *   by_image = 0 (few images): workgroup (x, y) = block x of image y, every XCD sees every level of every image.
     *   by_image = 1 (batches): a 1-D grid in which XCD k works through images k, k + 8, ... block by block -- an image's
     *     blocks share their 6-px overlaps and the partial 64-byte lines at their edges through ONE L2, and every XCD
     *     gets whole images, i.e. the same mix of levels.
     * Measured per 1024 images of 1280x720: an XCD-contiguous run of each image's blocks (round-2 first attempt) 4.62 ms at
     * 1.03x the algorithmic HBM bytes -- the runs hold different pyramid levels, the XCD with the light ones idles; plain
     * order 3.95 ms but 2.13x the bytes (neighbours land on different L2s); by image: see DESIGN.md section 4a. */
    int bid, b;
    if (by_image) {
        const unsigned L = blockIdx.x, xcd = L & 7u, j = L >> 3;
        const unsigned grp = j / (unsigned)nBlocks;
        bid = (int)(j - grp * (unsigned)nBlocks);
        b = (int)(grp * 8u + xcd);
        if (b >= nImages) return;
    } else {
        bid = (int)blockIdx.x;
        b = (int)blockIdx.y;
    }
    if (bid >= nBlocks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    /* the block descriptor through the SCALAR cache (constant address space: s_load instead of a vector load + six
     * v_readfirstlane): it heads the chain descriptor -> level geometry -> tile loads that every block waits out */
    FastBlock blk;
    {
        static_assert(sizeof(FastBlock) == 24, "six dwords");
        const __attribute__((address_space(4))) uint32_t* cb =
            reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(blocks + bid));
        uint32_t wds[6];
#pragma unroll
        for (int i = 0; i < 6; i++) wds[i] = cb[i];
        __builtin_memcpy(&blk, wds, sizeof blk);
    }
    const LevelGeom& L = g.lv[blk.level];
    int stride;
    const uint8_t* img = tb_level_ptr(g, slab, b, blk.level, &stride);
    const int rw = blk.x1 - blk.x0, rh = blk.y1 - blk.y0;
    FbGeom G;
    G.ax0 = blk.x0 & ~15;                                   /* LDS column 0 = image column ax0 */
    G.y0 = blk.y0;
    const int cx0 = blk.x0 - G.ax0;
    const int nseg = (cx0 + rw + 15) >> 4;                  /* 16-byte segments per row that hold ROI pixels */
    G.wCell = L.wCell; G.hCell = L.hCell; G.ncx = blk.ncx; G.ncy = blk.ncy;
    G.scanX0 = cx0 + 3; G.scanX1 = cx0 + rw - 3; G.scanY1 = rh - 3;
    const int wCell = G.wCell, hCell = G.hCell, scanX0 = G.scanX0, scanX1 = G.scanX1, scanY1 = G.scanY1;
    uint32_t* const out = cand + (size_t)b * g.candPerImage + L.candOff;
    int* const count = candCount + b * TB_MAX_LEVELS + blk.level;
    const int candCap = L.candCap;
#ifdef FB_TIMING
    unsigned long long t0_ = __builtin_readcyclecounter();
    unsigned long long dt_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    /* stage 1's lane map (from the host, FastBlock) and the lane's valid-pixel mask -- tile columns [scanX0, scanX1) in the
     * layout of the verdict word: two table reads, issued here so that they arrive behind stage 0's loads */
    const int nss = blk.nss, rowsPer = blk.rowsPer;
    const int rr = (lane * (int)blk.invNss) >> 15, ss = lane - rr * nss;
    const int col16 = 16 * (blk.sA + ss);
    const uint32_t valid = c_fbmask.ge[min(max(scanX0 - col16, 0), 16)] & c_fbmask.lt[min(max(scanX1 - col16, 0), 16)];

    /* ---- stage 0 */
    if (tid < 8) misc[tid] = 0;
    if (wave == FB_NW - 1) {   /* cell tables: one wavefront, three columns and a row per lane (the others go straight to the loads) */
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int tx = lane + 64 * j;
            if (tx < FB_S) {
                const int rel = tx - scanX0;
                int info = 0x80;
                if (rel >= 0 && tx < scanX1) {
                    const int k = (rel >= wCell) + (rel >= 2 * wCell) + (rel >= 3 * wCell);
                    info = k | ((rel == k * wCell) ? 16 : 0) | ((rel == (k + 1) * wCell - 1 || tx == scanX1 - 1) ? 32 : 0);
                }
                colinfo[tx] = (uint8_t)info;
            }
        }
        for (int ty = lane; ty < FB_TH; ty += 64) {
            const int rel = ty - 3;
            int info = 0x80;
            if (rel >= 0 && ty < scanY1) {
                const int k = (rel >= hCell) ? 1 : 0;
                info = k | ((rel == k * hCell) ? 16 : 0) | ((rel == (k + 1) * hCell - 1 || ty == scanY1 - 1) ? 32 : 0);
            }
            rowinfo[ty] = (uint8_t)info;
        }
    }
    {
        const bool wide = ((stride & 15) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
        const uint32_t M6 = 0x3f3f3f3fu;
        if (wide) {
            /* Tile row r, 16-byte segment s is LDS segment 10 r + s (FB_S = 160): thread tid takes segments tid, tid + FB_NT, ... One buffer resource per block, base = (row y0, column ax0), bounded by the end of the ROI's last row
             * (padding included: rows are stride bytes apart inside one allocation): rows below the ROI read as zero without
             * touching memory, and so do the segments right of the ROI (offset out of range). All three loads are in flight
             * before the first LDS store; no 64-bit addresses. */
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(img) + (size_t)blk.y0 * stride + G.ax0, 0, rh * stride - G.ax0, 0x00020000);
            constexpr int NLD = (FB_NSEG_ALL + FB_NT - 1) / FB_NT;
            ft_u4 v[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int sidx = tid + FB_NT * k;
                const int row = (sidx * 6554) >> 16, seg = sidx - row * FB_SEGS;      /* sidx / 10 */
                v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, seg < nseg ? row * stride + 16 * seg : 0x7ffffff0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NLD; k++)
                if (k < NLD - 1 || tid < FB_NSEG_ALL - FB_NT * (NLD - 1)) {
                    const int o = 16 * (tid + FB_NT * k);
                    *reinterpret_cast<ft_u4*>(P + o) = v[k];
                    *reinterpret_cast<ft_u4*>(Q + o) = (v[k] >> 2) & M6;
                }
        } else {                        /* caller-owned level 0 with an odd stride: bytes, bounded by the row */
#pragma unroll 1
            for (int sidx = tid; sidx < FB_NSEG_ALL; sidx += FB_NT) {
                const int row = (sidx * 6554) >> 16, seg = sidx - row * FB_SEGS;
                uint32_t w4[4] = {0, 0, 0, 0};
                if (row < rh && seg < nseg) {
                    const uint8_t* src = img + (size_t)(blk.y0 + row) * stride + (G.ax0 + 16 * seg);
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        if (G.ax0 + 16 * seg + j < L.w) w4[j >> 2] |= (uint32_t)src[j] << (8 * (j & 3));
                }
                *reinterpret_cast<uint4*>(P + 16 * sidx) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                *reinterpret_cast<uint4*>(Q + 16 * sidx) = make_uint4((w4[0] >> 2) & M6, (w4[1] >> 2) & M6, (w4[2] >> 2) & M6, (w4[3] >> 2) & M6);
            }
        }
    }
    __syncthreads();
    FB_T(0);

    /* ---- stage 1: quantised cardinal test, 16 pixels per lane */
    {
        const int tq = (init_th + 1) >> 2;
        const uint32_t K = (uint32_t)(128 - tq) * 0x01010101u;
        const int nPass = blk.nPass;
        for (int p = wave; p < nPass; p += FB_NW) {
            const int r = 3 + p * rowsPer + rr;
            uint32_t Gm = 0;
            const int base = r * FB_S + col16;
            if (rr < rowsPer && r < scanY1) {
                const uint4 C = *reinterpret_cast<const uint4*>(Q + base);
                const uint4 T = *reinterpret_cast<const uint4*>(Q + base - 3 * FB_S);
                const uint4 B = *reinterpret_cast<const uint4*>(Q + base + 3 * FB_S);
                const uint32_t Lw = *reinterpret_cast<const uint32_t*>(Q + base - 4);
                const uint32_t Rw = *reinterpret_cast<const uint32_t*>(Q + base + 16);
                const uint32_t c[6] = {Lw, C.x, C.y, C.z, C.w, Rw};
                const uint32_t t[4] = {T.x, T.y, T.z, T.w};
                const uint32_t bb[4] = {B.x, B.y, B.z, B.w};
                uint32_t F[4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t X12 = __builtin_amdgcn_alignbyte(c[d + 1], c[d], 1);      /* column - 3 */
                    const uint32_t X4 = __builtin_amdgcn_alignbyte(c[d + 2], c[d + 1], 3);   /* column + 3 */
                    const uint32_t A = K - c[d + 1], E = K + c[d + 1];
                    const uint32_t b08 = (t[d] + A) | (bb[d] + A), b412 = (X4 + A) | (X12 + A);
                    const uint32_t d08 = (E - t[d]) | (E - bb[d]), d412 = (E - X4) | (E - X12);
                    F[d] = (b08 & b412) | (d08 & d412);
                }
                /* 16 verdicts -> one word: dword d's land in bit 7 - d of every byte */
                Gm = ((F[0] & 0x80808080u) | ((F[1] >> 1) & 0x40404040u) | ((F[2] >> 2) & 0x20202020u) | ((F[3] >> 3) & 0x10101010u)) & valid;
            }
            const bool hit = Gm != 0;
            const unsigned long long bm = __ballot(hit);
            if (bm) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&misc[0], __popcll(bm));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                /* the verdicts sit in bits 4-7 of every byte: the segment number (10 bits) travels in the low nibbles */
                const uint32_t pos = (uint32_t)(base >> 4);
                const uint32_t enc = (pos & 0xfu) | ((pos << 4) & 0xf00u) | ((pos << 8) & 0x30000u);
                if (hit) REC[wbase + __popcll(bm & ((1ull << lane) - 1))] = Gm | enc;
            }
        }
    }
    __syncthreads();
    FB_T(1);

    /* ---- stage 2a: Q -> zeroed score map; records -> pixel list */
    const int nrec = misc[0];
    bool dense = force_dense != 0;
    if (!dense) {
        for (int i = tid; i < FB_PX / 16; i += FB_NT) reinterpret_cast<uint4*>(SC)[i] = make_uint4(0, 0, 0, 0);
        for (int i0 = 0; i0 < nrec; i0 += FB_NT) {
            const int i = i0 + tid;
            uint32_t Gm = 0;
            int px0 = 0;
            if (i < nrec) {
                const uint32_t rec = REC[i];
                Gm = rec & 0xf0f0f0f0u;
                px0 = (int)((rec & 0xfu) | ((rec >> 4) & 0xf0u) | ((rec >> 8) & 0x300u)) << 4;
            }
            const int c = __popc(Gm);
            const int incl = tb_wave_incl_scan_dpp(c);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            int wbase = 0;
            if (lane == 0 && tot) wbase = atomicAdd(&misc[1], tot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (wbase + tot <= FB_LIST_CAP) {
                int slot = wbase + incl - c;
                while (Gm) {
                    const int bit = __ffs((int)Gm) - 1;
                    Gm &= Gm - 1;
                    LIST[slot++] = (uint16_t)(px0 + 4 * (7 - (bit & 7)) + (bit >> 3));
                }
            }
        }
    }
    __syncthreads();
    FB_T(2);

    /* ---- stage 2b: exact scores, one pixel per lane, batches of 64 dealt to the wavefronts */
    const int npx = misc[1];
    dense = dense || npx > FB_LIST_CAP;
    if (!dense) {
        for (int base = 64 * wave; base < npx; base += FB_NT) {
            const int i = base + lane;
            const bool v = i < npx;
            const int idx = v ? LIST[i] : (3 * FB_S + 4);
            const int s = fb_score1(P + idx) - 1;
            const bool c = v && s >= init_th && s > 0;
            const unsigned long long m = __ballot(c);
            if (m) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&misc[2], __popcll(m));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (wbase + __popcll(m) <= FB_CL_CAP && c) {
                    CL[wbase + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)idx;
                    SC[idx] = (uint8_t)min(s, 255);
                }
            }
        }
    }
    __syncthreads();
    FB_T(3);

    /* ---- stage 3: NMS inside the cells; which cells keep a corner at initTh? */
    const int ncl = misc[2];
    dense = dense || ncl > FB_CL_CAP;
    if (!dense) {
        for (int i0 = 0; i0 < ncl; i0 += FB_NT) {
            const int i = i0 + tid;
            int cellbit = 0;
            if (i < ncl) {
                int cell;
                if (fb_nms_keep(SC, colinfo, rowinfo, CL[i], &cell)) cellbit = 1 << cell;
                else CL[i] = 0xffff;
            }
            const int anyw = tb_wave_or_dpp(cellbit);
            if (lane == 63 && anyw) atomicOr(&misc[3], anyw);
        }
    }
    __syncthreads();
    FB_T(4);

    /* ---- retry at minTh, one wavefront per cell without a survivor */
    int nAll = ncl;
    if (!dense && min_th < init_th) {
        const uint32_t rowm = (1u << blk.ncx) - 1u;
        const uint32_t need = (rowm | (blk.ncy > 1 ? rowm << FB_MAX_CX : 0u)) & ~(uint32_t)misc[3];
        if (need) {
            uint16_t* const rl = LIST + wave * (FB_RETRY_CAP + 64);       /* the pixel list is dead by now */
            int k = 0;
            for (int cellId = 0; cellId < FB_MAX_CX * FB_MAX_CY; cellId++) {
                if (!((need >> cellId) & 1u)) continue;
                if ((k++ & 3) != wave) continue;      /* wave-private lists for four wavefronts: the others have none to do */
                const int cx = cellId & (FB_MAX_CX - 1), cy = cellId / FB_MAX_CX;
                const int X0 = scanX0 + cx * wCell, X1 = min(X0 + wCell, scanX1);
                const int Y0 = 3 + cy * hCell, Y1 = min(Y0 + hCell, scanY1);
                const int rowsStrip = max(FB_RETRY_CAP / max(X1 - X0, 1), 1);
                for (int ys = Y0; ys < Y1; ys += rowsStrip) {
                    int n = 0;
                    const int ye = min(ys + rowsStrip, Y1);
                    for (int y = ys; y < ye; y++)
                        for (int xb = X0; xb < X1; xb += 64) {               /* cells are at most 59 px wide: one pass */
                            const int x = xb + lane, idx = y * FB_S + x;
                            const bool hit = x < X1 && fb_cardinal(P + idx, min_th);
                            const unsigned long long bm = __ballot(hit);
                            if (hit) rl[n + __popcll(bm & ((1ull << lane) - 1))] = (uint16_t)idx;
                            n += __popcll(bm);
                        }
                    ft_lds_fence();
                    for (int base = 0; base < n; base += 64) {
                        const int i = base + lane;
                        const bool v = i < n;
                        const int idx = v ? rl[i] : (3 * FB_S + 4);
                        const int s = fb_score1(P + idx) - 1;
                        /* corners at minTh that pass 1 has not listed already */
                        const bool c = v && s >= min_th && s > 0 && s < init_th;
                        const unsigned long long m = __ballot(c);
                        if (m) {
                            int wbase = 0;
                            if (lane == 0) wbase = atomicAdd(&misc[2], __popcll(m));
                            wbase = __builtin_amdgcn_readfirstlane(wbase);
                            if (wbase + __popcll(m) <= FB_CL_CAP && c) {
                                CL[wbase + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)idx;
                                SC[idx] = (uint8_t)s;
                            }
                        }
                    }
                    ft_lds_fence();
                }
            }
            __syncthreads();
            nAll = misc[2];
            dense = nAll > FB_CL_CAP;
            if (!dense) {
                for (int i = ncl + tid; i < nAll; i += FB_NT) {
                    int cell;
                    if (!fb_nms_keep(SC, colinfo, rowinfo, CL[i], &cell)) CL[i] = 0xffff;
                }
            }
            __syncthreads();
        }
    }
    FB_T(5);

#ifdef FB_TIMING
    if (dense && tid == 0) { atomicAdd(&fb_times[14], 1ull); atomicMax(&fb_times[15], (unsigned long long)npx); atomicMax(&fb_times[13], (unsigned long long)nrec); atomicMax(&fb_times[7], (unsigned long long)ncl); }
#endif
    if (dense) {   /* block-uniform: every operand of the decision came out of LDS behind a barrier */
        fb_dense(P, SC, colinfo, rowinfo, G, init_th, min_th, count, out, candCap);
        return;
    }

    /* ---- emit: every wavefront its share of the list, one global atomic each */
    {
        int mine = 0;
        for (int i0 = 64 * wave; i0 < nAll; i0 += FB_NT) {
            const int i = i0 + lane;
            mine += __popcll(__ballot(i < nAll && CL[i] != 0xffff));
        }
        if (mine) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(count, mine);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            for (int i0 = 64 * wave; i0 < nAll; i0 += FB_NT) {
                const int i = i0 + lane;
                const int idx = i < nAll ? CL[i] : 0xffff;
                const bool e = idx != 0xffff;
                const unsigned long long m = __ballot(e);
                if (e) {
                    const int slot = wbase + __popcll(m & ((1ull << lane) - 1));
                    if (slot < candCap) out[slot] = fb_record(SC, G, idx);
                }
                wbase += __popcll(m);
            }
        }
    }
    FB_T(6);
#ifdef FB_TIMING
    if (tid == 0 && (bid & 63) == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&fb_times[i], dt_[i]);
        atomicAdd(&fb_times[8], 1ull); atomicAdd(&fb_times[9], (unsigned long long)nrec); atomicAdd(&fb_times[10], (unsigned long long)npx);
        atomicAdd(&fb_times[11], (unsigned long long)ncl); atomicAdd(&fb_times[12], (unsigned long long)(nAll - ncl));
    }
#endif
}

int tbk_fast_cells(tb_extractor* ex, int n, int init_th, int min_th) {
    tb_ctx* ctx = ex->ctx;
    TB_HIP(ctx, hipMemsetAsync(ex->d_candCount, 0, sizeof(int32_t) * TB_MAX_LEVELS * n, ctx->stream));
    if (ex->nBlocksTotal == 0) return TB_OK;
    /* batches: an image per XCD at a time (see the kernel); a few images: plain (block, image) order so that all XCDs work */
    const int by_image = n >= 64 ? 1 : 0;
    const dim3 grid = by_image ? dim3((unsigned)ex->nBlocksTotal * 8u * (unsigned)((n + 7) / 8)) : dim3(ex->nBlocksTotal, n);
    static_assert(FB_LDS_BYTES - FB_PAD_LDS <= 32 * 1024, "five blocks per CU");
    /* test hook (tb_debug_force_dense_fast): every block down the any-density path (same results, no lists) */
    const int force_dense = ctx->dbg_fast_dense;
    tb_prof_begin(ctx, "k_fast_cells");
    hipLaunchKernelGGL(k_fast_blocks, grid, dim3(FB_NT), FB_LDS_BYTES, ctx->stream, ex->g, ex->d_slab, ex->d_blocks, ex->nBlocksTotal,
                       n, by_image, ex->d_cand, ex->d_candCount, init_th, min_th, force_dense);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- whole-image mode: 58x58 output tiles, scan region [3,w-3) x [3,h-3) */
#define FT_OUT 58

template <int ARC>
__global__ void __launch_bounds__(256)
k_fast_image(const uint8_t* __restrict__ img, int w, int h, int stride, size_t pitch, int th, int nms,
             uint32_t* __restrict__ out, int cap, size_t out_pitch, int32_t* __restrict__ count, int count_stride) {
    const int b = blockIdx.z;
    FastRegion R;
    R.ox0 = 3 + blockIdx.x * FT_OUT; R.oy0 = 3 + blockIdx.y * FT_OUT;
    R.ox1 = min(R.ox0 + FT_OUT, w - 3); R.oy1 = min(R.oy0 + FT_OUT, h - 3);
    R.sx0 = max(R.ox0 - 1, 3); R.sy0 = max(R.oy0 - 1, 3);
    R.sx1 = min(R.ox1 + 1, w - 3); R.sy1 = min(R.oy1 + 1, h - 3);
    R.lx0 = R.sx0 - 3; R.ly0 = R.sy0 - 3; R.lx1 = R.sx1 + 3; R.ly1 = R.sy1 + 3;
    ft_process<ARC>(img + (size_t)b * pitch, stride, R, th, th, false, nms != 0, 0, 0, out + (size_t)b * out_pitch, cap,
                    count + (size_t)b * count_stride);
}

int tbk_fast_image(tb_ctx* ctx, const uint8_t* d_img, int w, int h, int stride, int th, int nms, int arc,
                   uint32_t* d_out, int cap, int32_t* d_count) {
    TB_HIP(ctx, hipMemsetAsync(d_count, 0, sizeof(int32_t), ctx->stream));
    if (w < 7 || h < 7) return TB_OK;
    dim3 grid((w - 6 + FT_OUT - 1) / FT_OUT, (h - 6 + FT_OUT - 1) / FT_OUT, 1);
    tb_prof_begin(ctx, "k_fast_image");
    if (arc == 9)
        hipLaunchKernelGGL(k_fast_image<9>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, (size_t)0, th, nms,
                           d_out, cap, (size_t)0, d_count, 0);
    else
        hipLaunchKernelGGL(k_fast_image<10>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, (size_t)0, th, nms,
                           d_out, cap, (size_t)0, d_count, 0);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- a11: FASTExtractor grid selection.
 * Per level: FAST-10 (th 20) + NMS over the whole level (above), then every surviving corner bids for
 * its grid cell with (Shi-Tomasi score, first-seen order) -- FASTextractor.cpp:53-69.  "First seen" in
 * the reference is (level, raster order); the bid key reproduces it so the atomicMax is order free:
 *   key = score_bits<<32 | ~(level<<26 | y<<13 | x)     (score > 0 so its float bits sort as uints). */
__device__ __forceinline__ float ft_shi_tomasi(const uint8_t* img, int w, int h, int stride, int u, int v) {
    /* FASTExtractor::shiTomasiScore, FASTextractor.cpp:87-127 (float accumulation in scan order) */
    float dXX = 0.f, dYY = 0.f, dXY = 0.f;
    const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
    if (x_min < 1 || x_max >= w - 1 || y_min < 1 || y_max >= h - 1) return 0.f;
    for (int y = y_min; y < y_max; ++y) {
        const uint8_t* row = img + (size_t)y * stride;
        for (int x = x_min; x < x_min + 8; ++x) {
            const float dx = (float)row[x + 1] - (float)row[x - 1];
            const float dy = (float)row[x + stride] - (float)row[x - stride];
            dXX = TB_FADD(dXX, TB_FMUL(dx, dx));
            dYY = TB_FADD(dYY, TB_FMUL(dy, dy));
            dXY = TB_FADD(dXY, TB_FMUL(dx, dy));
        }
    }
    dXX = TB_FDIV(dXX, 128.f);
    dYY = TB_FDIV(dYY, 128.f);
    dXY = TB_FDIV(dXY, 128.f);
    const float tr = TB_FADD(dXX, dYY);
    const float disc = TB_FSUB(TB_FMUL(tr, tr), TB_FMUL(4.f, TB_FSUB(TB_FMUL(dXX, dYY), TB_FMUL(dXY, dXY))));
    return TB_FMUL(0.5f, TB_FSUB(tr, sqrtf(disc)));
}

__global__ void __launch_bounds__(256)
k_fastgrid_bid(PlanGeom g, const uint8_t* __restrict__ slab, const uint32_t* __restrict__ cand,
               const int32_t* __restrict__ candCount, int level, int cell_size, int grid_cols, int ncell,
               const uint8_t* __restrict__ occ, int n_occ, unsigned long long* __restrict__ best) {
    const int b = blockIdx.y;
    const LevelGeom& L = g.lv[level];
    const int n = min(candCount[b * TB_MAX_LEVELS + level], L.candCap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t rec = cand[(size_t)b * g.candPerImage + L.candOff + i];
    const int x = rec & 0xfff, y = (rec >> 12) & 0xfff;
    const float fx = TB_FMUL((float)x, L.inv_sf), fy = TB_FMUL((float)y, L.inv_sf);
    const int k = (int)TB_FDIV(fy, (float)cell_size) * grid_cols + (int)TB_FDIV(fx, (float)cell_size);
    if (k < 0 || k >= ncell) return;
    if (occ && k < n_occ && occ[k]) return;
    int stride;
    const uint8_t* img = tb_level_ptr(g, slab, b, level, &stride);
    const float score = ft_shi_tomasi(img, L.w, L.h, stride, x, y);
    if (!(score > 0.f)) return;
    const uint32_t order = ((uint32_t)level << 26) | ((uint32_t)y << 13) | (uint32_t)x;
    const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (uint32_t)(~order);
    atomicMax(&best[(size_t)b * ncell + k], key);
}

__global__ void __launch_bounds__(256)
k_fastgrid_emit(PlanGeom g, const unsigned long long* __restrict__ best, int ncell, float threshold,
                tb_keypoint* __restrict__ kps, int32_t* __restrict__ counts) {
    /* one block per image: cells in index order -> keyPoints order of FASTextractor.cpp:72-78 */
    __shared__ int flags[256];
    __shared__ int tmp[8];
    __shared__ int running;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += 256) {
        const int k = base + tid;
        unsigned long long key = 0;
        float score = 0.f;
        if (k < ncell) {
            key = best[(size_t)b * ncell + k];
            score = __uint_as_float((uint32_t)(key >> 32));
        }
        const int f = (k < ncell && key != 0 && score > threshold) ? 1 : 0;
        flags[tid] = f;
        __syncthreads();
        const int total = tb_block_excl_scan(flags, 256, tmp);
        if (f) {
            const uint32_t order = ~(uint32_t)key;
            const int level = order >> 26, y = (order >> 13) & 0x1fff, x = order & 0x1fff;
            tb_keypoint kp;
            kp.x = TB_FMUL((float)x, g.lv[level].inv_sf);
            kp.y = TB_FMUL((float)y, g.lv[level].inv_sf);
            kp.size = 1.f; kp.angle = 0.f; kp.response = score; kp.octave = level; kp.class_id = -1;
            kps[(size_t)b * g.selCap + running + flags[tid]] = kp;
        }
        __syncthreads();
        if (tid == 0) running += total;
        __syncthreads();
    }
    if (tid == 0) counts[b] = running;
}

int tbk_fastgrid(tb_extractor* ex, int n, int target, float threshold, int n_occ) {
    tb_ctx* ctx = ex->ctx;
    const PlanGeom& g = ex->g;
    const int cell_size = (int)sqrtf((float)g.width * (float)g.height / (float)target);
    if (cell_size < 1) return tb_fail(ctx, TB_EINVAL, "fastgrid: cell size < 1");
    const int grid_cols = (int)((float)g.width / (float)cell_size);
    const int grid_rows = (int)((float)g.height / (float)cell_size);
    int ncell = (grid_rows + 2) * (grid_cols + 1);
    if (ncell < target) ncell = target;
    if (ncell > g.selCap) return tb_fail(ctx, TB_ECAPACITY, "fastgrid: %d grid cells exceed plan capacity %d", ncell, g.selCap);
    const size_t need = (size_t)n * ncell;
    if (need > ex->gridBestCap) {
        if (ex->d_gridBest) hipFree(ex->d_gridBest);
        ex->d_gridBest = nullptr;
        TB_HIP(ctx, hipMalloc(&ex->d_gridBest, need * sizeof(unsigned long long)));
        ex->gridBestCap = need;
    }
    TB_HIP(ctx, hipMemsetAsync(ex->d_gridBest, 0, need * sizeof(unsigned long long), ctx->stream));
    TB_HIP(ctx, hipMemsetAsync(ex->d_candCount, 0, sizeof(int32_t) * TB_MAX_LEVELS * n, ctx->stream));
    for (int l = 0; l < g.nlevels; l++) {
        const LevelGeom& L = g.lv[l];
        if (L.w < 7 || L.h < 7) continue;
        /* detect (image mode, all frames of the batch in grid.z) */
        dim3 grid((L.w - 6 + FT_OUT - 1) / FT_OUT, (L.h - 6 + FT_OUT - 1) / FT_OUT, n);
        const uint8_t* base;
        size_t pitch;
        int stride;
        if (l == 0 && g.img0) { base = g.img0; pitch = g.img0_pitch; stride = g.img0_stride; }
        else { base = ex->d_slab + L.off; pitch = g.slabBytes; stride = L.stride; }
        tb_prof_begin(ctx, "k_fast_image");
        hipLaunchKernelGGL(k_fast_image<10>, grid, dim3(256), 0, ctx->stream, base, L.w, L.h, stride, pitch, 20, 1,
                           ex->d_cand + L.candOff, L.candCap, (size_t)g.candPerImage, ex->d_candCount + l, TB_MAX_LEVELS);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
        dim3 bgrid((L.candCap + 255) / 256, n);
        tb_prof_begin(ctx, "k_fastgrid_bid");
        hipLaunchKernelGGL(k_fastgrid_bid, bgrid, dim3(256), 0, ctx->stream, g, ex->d_slab, ex->d_cand, ex->d_candCount, l,
                           cell_size, grid_cols, ncell, n_occ > 0 ? ex->d_occ : nullptr, n_occ, ex->d_gridBest);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
    }
    tb_prof_begin(ctx, "k_fastgrid_emit");
    hipLaunchKernelGGL(k_fastgrid_emit, dim3(n), dim3(256), 0, ctx->stream, g, ex->d_gridBest, ncell, threshold, ex->d_kps,
                       ex->d_counts);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
