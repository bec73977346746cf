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

/* ---- cell mode: one WAVEFRONT per 30-px cell, four cells per workgroup, no workgroup barriers.
 * Per wave: ROI rows staged into a private LDS tile (32-bit loads when the level is 4-byte aligned);
 *   stage 1  cardinal test (ring positions 0/4/8/12 hold >= 2 adjacent members of any 9-arc) on every
 *            pixel, two ROI rows per 64-lane pass, survivors ballot-compacted into an LDS list;
 *   stage 2  full 16-position arc test on the list, compaction in place;
 *   stage 3  exact scores of the corners into a private score map;
 *   stage 4  3x3 strict maximum, initTh / minTh choice by wave ballot, emission through one atomic per
 *            wave-pass.  LDS per wave = tile + score map + list (sized by the plan's largest ROI). */
struct FastCellsArgs {
    int nCells;        /* cells per image */
    int tileStride;    /* bytes per LDS tile row (multiple of 4) */
    int tileRows;
    int listCap;       /* interior pixels of the largest ROI */
    int waveBytes;     /* LDS bytes per wave */
};

typedef short ft_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ft_s16x2 ft_pack(int a, int b) { return (ft_s16x2){(short)a, (short)b}; }
__device__ __forceinline__ ft_s16x2 ft_min(ft_s16x2 a, ft_s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ ft_s16x2 ft_max(ft_s16x2 a, ft_s16x2 b) { return __builtin_elementwise_max(a, b); }

__device__ __forceinline__ void ft_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

template <int ARC>
__device__ __forceinline__ bool ft_is_corner_s(const uint8_t* c, int S, int th) {
    const int v = c[0];
    const int hi = v + th, lo = v - th;
    int r[16];
    r[0] = c[3 * S];      r[1] = c[3 * S + 1];   r[2] = c[2 * S + 2];   r[3] = c[S + 3];
    r[4] = c[3];          r[5] = c[-S + 3];      r[6] = c[-2 * S + 2];  r[7] = c[-3 * S + 1];
    r[8] = c[-3 * S];     r[9] = c[-3 * S - 1];  r[10] = c[-2 * S - 2]; r[11] = c[-S - 3];
    r[12] = c[-3];        r[13] = c[S - 3];      r[14] = c[2 * S - 2];  r[15] = c[3 * S - 1];
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

template <int ARC>
__device__ __forceinline__ int ft_score_s(const uint8_t* c, int S) {
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * S];      d[1] = v - c[3 * S + 1];   d[2] = v - c[2 * S + 2];   d[3] = v - c[S + 3];
    d[4] = v - c[3];          d[5] = v - c[-S + 3];      d[6] = v - c[-2 * S + 2];  d[7] = v - c[-3 * S + 1];
    d[8] = v - c[-3 * S];     d[9] = v - c[-3 * S - 1];  d[10] = v - c[-2 * S - 2]; d[11] = v - c[-S - 3];
    d[12] = v - c[-3];        d[13] = v - c[S - 3];      d[14] = v - c[2 * S - 2];  d[15] = v - c[3 * S - 1];
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

template <int ST> /* ST = LDS tile row stride in bytes (0 = runtime value): a constant folds every ring offset into the
                     ds_read immediate field */
__global__ void __launch_bounds__(256)
k_fast_cells(PlanGeom g, const uint8_t* __restrict__ slab, const CellDesc* __restrict__ cells,
             uint32_t* __restrict__ cand, int32_t* __restrict__ candCount, int init_th, int min_th, FastCellsArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    /* wave-uniform values are made scalar so the cell / level descriptors come through the scalar cache */
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cellId = blockIdx.x * 4 + wave;
    if (cellId >= A.nCells) return;
    const int b = blockIdx.y;
    const CellDesc c = cells[cellId];
    const LevelGeom& L = g.lv[c.level];
    int stride;
    const uint8_t* img = tb_level_ptr(g, slab, b, c.level, &stride);
    const int S = ST ? ST : A.tileStride;
    uint8_t* tile = smem + (size_t)wave * A.waveBytes;
    uint8_t* sc = tile + S * A.tileRows;
    uint16_t* list = reinterpret_cast<uint16_t*>(sc + S * A.tileRows);

    /* ---- stage 0: ROI rows -> LDS. LDS column 0 = image column ax0 (x0 rounded down to 4) */
    const int rw = c.x1 - c.x0, rh = c.y1 - c.y0;
    const int ax0 = c.x0 & ~3, cx0 = c.x0 - ax0;
    const int nd = (cx0 + rw + 3) >> 2; /* dwords per row */
    const bool aligned = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(img) & 3) == 0) && (ax0 + nd * 4 <= stride);
    if (aligned) {
        const int per = 64 / nd; /* rows per pass (nd <= 18) */
        const int rr = lane / nd, dd = lane - rr * nd;
        const uint8_t* src = img + (size_t)c.y0 * stride + ax0 + 4 * dd;
        for (int y0 = 0; y0 < rh; y0 += 8 * per) { /* up to 8 row loads in flight per lane before the LDS stores */
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int y = y0 + j * per + rr;
                v[j] = 0;
                if (rr < per && y < rh) v[j] = *reinterpret_cast<const uint32_t*>(src + (size_t)y * stride);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int y = y0 + j * per + rr;
                if (rr < per && y < rh) {
                    *reinterpret_cast<uint32_t*>(tile + y * S + 4 * dd) = v[j];
                    *reinterpret_cast<uint32_t*>(sc + y * S + 4 * dd) = 0;
                }
            }
        }
    } else {
        for (int y = 0; y < rh; y++)
            for (int x = lane; x < nd * 4; x += 64) {
                const int ax = ax0 + x;
                tile[y * S + x] = (ax >= c.x0 && ax < c.x1) ? img[(size_t)(c.y0 + y) * stride + ax] : 0;
                sc[y * S + x] = 0;
            }
    }
    ft_lds_fence();

    /* ---- stage 1: cardinal test on every scanned pixel, FOUR pixels per lane, SWAR on 16-bit sub-lanes.
     * With V the centre, X a ring pixel, t the threshold, per 16-bit lane:
     *   X > V + t  <=>  bit 9 of  X + (511 - t - V)        (value in [1, 766]: no carry between lanes)
     *   X < V - t  <=>  bit 9 of  (V + 511 - t) - X        (value in [1, 766]: no borrow)
     * A 9-arc holds two adjacent cardinals, i.e. one of {0, 8} and one of {4, 12}. */
    const int sw = rw - 6, sh = rh - 6;
    /* The cell is tried at init_th first, as the reference does (ORBextractor.cpp:785-797): only where that leaves no
     * corner after NMS is it redone at min_th. NMS does not depend on the threshold (a neighbour below the threshold
     * scores below any corner at it), so the first pass is exact -- and on textured frames it sends a quarter of the
     * pixels into the score stage that a single pass at min_th would (2.6 % of the pixels are corners at 80, 10 % at 30). */
    int thA = init_th;
    int nlist = 0, ncorner = 0;
    auto nms_keep = [&](int idx) -> bool {
        const int s = sc[idx];
        return s > 0 && s > sc[idx - 1] && s > sc[idx + 1] && s > sc[idx - S - 1] && s > sc[idx - S] && s > sc[idx - S + 1] &&
               s > sc[idx + S - 1] && s > sc[idx + S] && s > sc[idx + S + 1];
    };
    for (int pass = 0; pass < 2; pass++) {
    nlist = 0;
    ncorner = 0;
    if (sw > 0 && sh > 0) {
        const int cA = cx0 + 3, cB = cx0 + rw - 3;   /* scanned LDS columns [cA, cB) */
        const int gA = cA >> 2, ng = ((cB - 1) >> 2) - gA + 1;
        const int rowsPer = 64 / ng;
        const int rr = lane / ng, gg = lane - rr * ng;
        const uint32_t K = (uint32_t)(511 - thA) * 0x00010001u;
        const uint32_t M = 0x00ff00ffu;
        const int col0 = 4 * (gA + gg);
        /* valid pixels of this lane's group */
        uint32_t vmask = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) vmask |= (uint32_t)((col0 + i >= cA) && (col0 + i < cB)) << i;
        for (int y = 0; y < sh; y += rowsPer) {
            const int yy = y + rr;
            uint32_t m = 0;
            const int base = (3 + yy) * S + col0;
            if (rr < rowsPer && yy < sh) {
                const uint32_t C = *reinterpret_cast<const uint32_t*>(tile + base);
                const uint32_t Lw = *reinterpret_cast<const uint32_t*>(tile + base - 4);
                const uint32_t Rw = *reinterpret_cast<const uint32_t*>(tile + base + 4);
                const uint32_t Tt = *reinterpret_cast<const uint32_t*>(tile + base - 3 * S);
                const uint32_t Bt = *reinterpret_cast<const uint32_t*>(tile + base + 3 * S);
                const uint32_t X12 = __builtin_amdgcn_alignbyte(C, Lw, 1);
                const uint32_t X4 = __builtin_amdgcn_alignbyte(Rw, C, 3);
                const uint32_t VE = C & M, VO = (C >> 8) & M;
                const uint32_t AE = K - VE, AO = K - VO, CE = K + VE, CO = K + VO;
                uint32_t x, xe, xo;
                x = Bt; xe = x & M; xo = (x >> 8) & M;
                uint32_t b08e = xe + AE, b08o = xo + AO, d08e = CE - xe, d08o = CO - xo;
                x = Tt; xe = x & M; xo = (x >> 8) & M;
                b08e |= xe + AE; b08o |= xo + AO; d08e |= CE - xe; d08o |= CO - xo;
                x = X4; xe = x & M; xo = (x >> 8) & M;
                uint32_t b4e = xe + AE, b4o = xo + AO, d4e = CE - xe, d4o = CO - xo;
                x = X12; xe = x & M; xo = (x >> 8) & M;
                b4e |= xe + AE; b4o |= xo + AO; d4e |= CE - xe; d4o |= CO - xo;
                const uint32_t re = ((b08e & b4e) | (d08e & d4e)) & 0x02000200u;
                const uint32_t ro = ((b08o & b4o) | (d08o & d4o)) & 0x02000200u;
                m = (((re >> 9) & 1u) | (((ro >> 9) & 1u) << 1) | (((re >> 25) & 1u) << 2) | (((ro >> 25) & 1u) << 3)) & vmask;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool pass = (m >> i) & 1u;
                const unsigned long long bm = __ballot(pass);
                if (pass) list[nlist + __popcll(bm & ((1ull << lane) - 1))] = (uint16_t)(base + i);
                nlist += __popcll(bm);
            }
        }
    }
    ft_lds_fence();

    /* ---- stage 2 + 3: exact FAST score of every survivor, TWO survivors per lane in packed 16-bit halves.
     * score = max over the 16 arcs of min(d) and of min(-d), minus 1, with d = centre - ring; the min over a
     * 9-arc is a doubling network (2, 4, 8, 8+1), every step one v_pk_min_i16 / v_pk_max_i16 for both pixels.
     * A survivor is a corner at thA iff score >= thA; corners are compacted in place. */
    for (int base = 0; base < nlist; base += 128) {
        const int iA = base + lane, iB = base + 64 + lane;
        const bool vA = iA < nlist, vB = iB < nlist;
        const int idxA = vA ? list[iA] : (3 * S + 4), idxB = vB ? list[iB] : (3 * S + 4);
        const uint8_t* pA = tile + idxA;
        const uint8_t* pB = tile + idxB;
        const ft_s16x2 vv = ft_pack(pA[0], pB[0]);
        ft_s16x2 d[16];
        d[0] = vv - ft_pack(pA[3 * S], pB[3 * S]);           d[1] = vv - ft_pack(pA[3 * S + 1], pB[3 * S + 1]);
        d[2] = vv - ft_pack(pA[2 * S + 2], pB[2 * S + 2]);   d[3] = vv - ft_pack(pA[S + 3], pB[S + 3]);
        d[4] = vv - ft_pack(pA[3], pB[3]);                   d[5] = vv - ft_pack(pA[-S + 3], pB[-S + 3]);
        d[6] = vv - ft_pack(pA[-2 * S + 2], pB[-2 * S + 2]); d[7] = vv - ft_pack(pA[-3 * S + 1], pB[-3 * S + 1]);
        d[8] = vv - ft_pack(pA[-3 * S], pB[-3 * S]);         d[9] = vv - ft_pack(pA[-3 * S - 1], pB[-3 * S - 1]);
        d[10] = vv - ft_pack(pA[-2 * S - 2], pB[-2 * S - 2]); d[11] = vv - ft_pack(pA[-S - 3], pB[-S - 3]);
        d[12] = vv - ft_pack(pA[-3], pB[-3]);                d[13] = vv - ft_pack(pA[S - 3], pB[S - 3]);
        d[14] = vv - ft_pack(pA[2 * S - 2], pB[2 * S - 2]);  d[15] = vv - ft_pack(pA[3 * S - 1], pB[3 * S - 1]);
        ft_s16x2 lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { lo2[k] = ft_min(d[k], d[(k + 1) & 15]); hi2[k] = ft_max(d[k], d[(k + 1) & 15]); }
#pragma unroll
        for (int k = 0; k < 16; k++) { lo4[k] = ft_min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = ft_max(hi2[k], hi2[(k + 2) & 15]); }
        ft_s16x2 best = (ft_s16x2){0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const ft_s16x2 lo9 = ft_min(ft_min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);   /* min over the 9-arc starting at k */
            const ft_s16x2 hi9 = ft_max(ft_max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
            best = ft_max(best, ft_max(lo9, -hi9));
        }
        const int sA = (int)best.x - 1, sB = (int)best.y - 1;
        const bool cA = vA && sA >= thA && sA > 0, cB = vB && sB >= thA && sB > 0;
        ft_lds_fence(); /* every lane has read its two list entries before the compaction overwrites them */
        const unsigned long long mA = __ballot(cA);
        if (cA) { list[ncorner + __popcll(mA & ((1ull << lane) - 1))] = (uint16_t)idxA; sc[idxA] = (uint8_t)min(sA, 255); }
        ncorner += __popcll(mA);
        const unsigned long long mB = __ballot(cB);
        if (cB) { list[ncorner + __popcll(mB & ((1ull << lane) - 1))] = (uint16_t)idxB; sc[idxB] = (uint8_t)min(sB, 255); }
        ncorner += __popcll(mB);
        ft_lds_fence();
    }

    /* ---- stage 4: NMS; does this pass leave a corner? (list[] holds the corners at thA, sc[] their scores, 0 elsewhere) */
    bool any = false;
    for (int base = 0; base < ncorner; base += 64) {
        const int i = base + lane;
        const bool keep = (i < ncorner) && nms_keep(list[i]);
        any = any || (__ballot(keep) != 0);
    }
    if (any || pass == 1 || min_th >= init_th) break;
    /* nothing at init_th: forget this pass's scores (corners that lost the NMS) and redo the cell at min_th */
    for (int base = 0; base < ncorner; base += 64) {
        const int i = base + lane;
        if (i < ncorner) sc[list[i]] = 0;
    }
    ft_lds_fence();
    thA = min_th;
    }
    for (int base = 0; base < ncorner; base += 64) { /* drop the corners that lose the NMS */
        const int i = base + lane;
        if (i < ncorner && !nms_keep(list[i])) list[i] = 0xffff; /* only the scores are read across lanes */
    }
    ft_lds_fence();
    uint32_t* out = cand + (size_t)b * g.candPerImage + L.candOff;
    int* count = candCount + b * TB_MAX_LEVELS + c.level;
    /* ONE returning atomic per cell (a returning atomic per 64-lane pass cost 30-50 % of this kernel: every
     * wave stalls on the round trip to a counter shared by the ~900 cells of its level): count the survivors
     * first, reserve the range, then write */
    int total = 0;
    for (int base = 0; base < ncorner; base += 64) {
        const int i = base + lane;
        bool e = false;
        if (i < ncorner) {
            e = list[i] != 0xffff;
        }
        total += __popcll(__ballot(e));
    }
    ft_lds_fence();
    if (total == 0) return;
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(count, total);
    wbase = __shfl(wbase, 0, TB_WAVE);
    for (int base = 0; base < ncorner; base += 64) {
        const int i = base + lane;
        bool e = false;
        uint32_t rec = 0;
        if (i < ncorner) {
            const int idx = list[i];
            if (idx != 0xffff) {
                const int y = idx / S, x = idx - y * S;
                e = true;
                rec = ((uint32_t)sc[idx] << 24) | ((uint32_t)(c.y0 + y - TB_BORDER) << 12) | (uint32_t)(ax0 + x - TB_BORDER);
            }
        }
        const unsigned long long m = __ballot(e);
        if (e) {
            const int slot = wbase + __popcll(m & ((1ull << lane) - 1));
            if (slot < L.candCap) out[slot] = rec;
        }
        wbase += __popcll(m);
    }
}

int tbk_fast_cells(tb_extractor* ex, int n, int init_th, int min_th) {
    tb_ctx* ctx = ex->ctx;
    TB_HIP(ctx, hipMemsetAsync(ex->d_candCount, 0, sizeof(int32_t) * TB_MAX_LEVELS * n, ctx->stream));
    if (ex->nCellsTotal == 0) return TB_OK;
    FastCellsArgs A;
    A.nCells = ex->nCellsTotal;
    A.tileStride = (ex->maxRoiW + 3 + 3 + 4) & ~3; /* ROI + alignment slack, multiple of 4 */
    A.tileRows = ex->maxRoiH;
    A.listCap = (ex->maxRoiW - 6) * (ex->maxRoiH - 6);
    if (A.listCap < 64) A.listCap = 64;
    dim3 grid((ex->nCellsTotal + 3) / 4, n);
    void (*kern)(PlanGeom, const uint8_t*, const CellDesc*, uint32_t*, int32_t*, int, int, FastCellsArgs) = k_fast_cells<0>;
    if (A.tileStride <= 44) { A.tileStride = 44; kern = k_fast_cells<44>; }
    else if (A.tileStride <= 48) { A.tileStride = 48; kern = k_fast_cells<48>; }
    else if (A.tileStride <= 56) { A.tileStride = 56; kern = k_fast_cells<56>; }
    else if (A.tileStride <= 76) { A.tileStride = 76; kern = k_fast_cells<76>; }
    A.waveBytes = (2 * A.tileStride * A.tileRows + 2 * A.listCap + 15) & ~15;
    const size_t lds = 4 * (size_t)A.waveBytes;
    TB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    tb_prof_begin(ctx, "k_fast_cells");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, ex->g, ex->d_slab, ex->d_cells, ex->d_cand, ex->d_candCount,
                       init_th, min_th, A);
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
