/* a6 + a7 + a8 + a10 -- orientation, Gaussian blur and steered-BRIEF descriptor, fused per keypoint.
 *
 * Reference: IC_Angle (src/extractors/ORBextractor.cpp:17-44), GaussianBlur(7x7, sigma 2,
 * BORDER_REFLECT_101) on a clone of every level (:959-960), computeOrbDescriptor (:48-87) and the
 * coordinate rescale + level-major concatenation (:969-976).
 *
 * The reference blurs whole levels (2 x 2.48 MB of HBM traffic per 720p frame) and then gathers 512
 * taps per keypoint.  Every tap lies within 19 px of the keypoint, and the 8U blur is an exact integer
 * function of the 7x7 neighbourhood (kernel 18,34,49,55,49,34,18; (acc + 2^15) >> 16), so the blurred
 * 39x39 patch can be rebuilt bit-exactly from the 45x45 source patch that the orientation disc (radius
 * 15) needs anyway.  One wavefront per keypoint:
 *   stage 45x45 source patch in LDS (reflect-101 at level borders) -> integer moments (wave reduce) ->
 *   fastAtan2 -> separable blur in LDS (u16 intermediate is exact: 255*257 = 65535) -> 256 rotated tests,
 *   4 x 64-lane ballots = 32 descriptor bytes.
 * No blurred level is ever written to HBM.  Roofline: HBM/L2 gather of 2 KB per keypoint (SURVEY 8d
 * "orient + describe" row); VALU work ~370 multiply-adds per lane.
 */
#include "tb_internal.h"
#include "tb_device.h"

__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

#define DS_P 45      /* source patch edge */
#define DS_PS 52     /* source patch row stride: 13 dwords (odd) -> one-row-per-lane accesses hit distinct banks */
#define DS_B 39      /* blurred patch edge */
#define DS_HS 42     /* h-pass row stride in u16 (21 dwords, odd) */
#define DS_BS 40     /* blurred patch row stride */

__device__ __forceinline__ int ds_reflect(int i, int n) {
    /* BORDER_REFLECT_101; n >= 2 and |overshoot| < n for every level that can hold a keypoint */
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ int ds_byte(const uint32_t* w, int i) { /* byte i of a register array, i compile-time */
    return (int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}

/* Instruction budget per keypoint (one wavefront), the quantity this kernel is bound by:
 *   staging   interior keypoints copy 45 rows as 13 aligned dwords each (the sub-dword phase of the patch is
 *             kept as a column shift), border keypoints take the per-byte reflect-101 path;
 *   moments   one lane per disc row pair, bytes unpacked from dword LDS reads;
 *   h-pass    one lane per patch ROW: 13 dword reads, 39 sliding-window outputs, fully unrolled;
 *   v-pass    one lane per patch COLUMN: 45 u16 reads, 39 sliding-window outputs, fully unrolled;
 *   tests     4 x 64 rotated comparisons -> 4 ballots. */
__global__ void __launch_bounds__(64)
k_describe(PlanGeom g, const uint8_t* __restrict__ slab, const uint32_t* __restrict__ sel,
           const int32_t* __restrict__ selCount, tb_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
           int32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint8_t src[DS_P * DS_PS + 16];
    __shared__ __attribute__((aligned(16))) unsigned short hp[DS_P * DS_HS];
    __shared__ __attribute__((aligned(16))) uint8_t bl[DS_B * DS_BS];
    const int b = blockIdx.y, slot = blockIdx.x, lane = threadIdx.x;
    const int32_t* sc = selCount + b * TB_MAX_LEVELS;
    /* slot -> (level, index), level-major output base */
    int level = 0, base = 0;
    for (int l = 0; l < g.nlevels; l++) {
        if (slot >= g.lv[l].selBase) level = l;
    }
    for (int l = 0; l < level; l++) base += sc[l];
    if (slot == 0 && lane == 0) {
        int tot = 0;
        for (int l = 0; l < g.nlevels; l++) tot += sc[l];
        counts[b] = tot;
    }
    const LevelGeom& G = g.lv[level];
    const int idx = slot - G.selBase;
    if (idx >= sc[level]) return;
    const uint32_t rec = sel[(size_t)b * g.selCap + slot];
    const int kx = (int)(rec & 0xfff) + TB_BORDER, ky = (int)((rec >> 12) & 0xfff) + TB_BORDER;
    int stride;
    const uint8_t* img = tb_level_ptr(g, slab, b, level, &stride);

    /* 1. stage the source patch: patch column c lives at LDS column c + sh */
    const int x0 = kx - 22, y0 = ky - 22;
    const bool interior = x0 >= 0 && y0 >= 0 && kx + 22 < G.w && ky + 22 < G.h && ((stride & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(img) & 3) == 0) && ((x0 & ~3) + 52 <= stride);
    const int sh = interior ? (x0 & 3) : 0;
    if (interior) {
        const int rr = lane / 13, dd = lane - rr * 13; /* 4 rows x 13 dwords per pass, lanes 52..63 idle */
        const uint8_t* srcp = img + (size_t)y0 * stride + (x0 & ~3) + 4 * dd;
        uint32_t v[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int r = 4 * j + rr;
            v[j] = 0;
            if (rr < 4 && r < DS_P) v[j] = *reinterpret_cast<const uint32_t*>(srcp + (size_t)r * stride);
        }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int r = 4 * j + rr;
            if (rr < 4 && r < DS_P) *reinterpret_cast<uint32_t*>(src + r * DS_PS + 4 * dd) = v[j];
        }
    } else {
        for (int e = lane; e < DS_P * DS_P; e += 64) {
            const int r = e / DS_P, c = e - r * DS_P;
            const int yy = ds_reflect(y0 + r, G.h), xx = ds_reflect(x0 + c, G.w);
            src[r * DS_PS + c] = img[(size_t)yy * stride + xx];
        }
    }
    __syncthreads();

    /* 2. IC_Angle: integer moments over the radius-15 disc; lane = (row v, half): 62 lanes active */
    int m10 = 0, m01 = 0;
    {
        const int v = (lane >> 1) - 15, half = lane & 1; /* half 0: u in [-15,-1], half 1: u in [0,15] */
        if (lane < 62) {
            const int av = v < 0 ? -v : v;
            /* umax[] of ORBextractor.cpp:389-404 = 15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3 */
            const unsigned long long UM = 0x3689ABCDDEEEFFFFull; /* nibble av = umax[av] */
            const int um = (int)((UM >> (4 * av)) & 0xf);
            const uint8_t* row = src + (22 + v) * DS_PS + 22 + sh;
            int sI = 0, sU = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int u = half ? k : -(k + 1);
                const int au = half ? k : k + 1;
                if (au <= um && au <= 15) {
                    const int I = row[u];
                    sI += I;
                    sU += u * I;
                }
            }
            m10 = sU;
            m01 = v * sI;
        }
    }
    m10 = tb_wave_sum(m10);
    m01 = tb_wave_sum(m01);
    const float angle = tbm::fast_atan2((float)m01, (float)m10);

    /* 3a. horizontal 7-tap pass, one lane per patch row (exact integers, u16 result) */
    if (lane < DS_P) {
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(src + lane * DS_PS);
        uint32_t w[13];
#pragma unroll
        for (int j = 0; j < 13; j++) w[j] = rw[j];
        /* fold the sub-dword phase away: four code paths with compile-time byte indices */
        unsigned short* out = hp + lane * DS_HS;
#define DS_HROW(SH)                                                                                                  \
    _Pragma("unroll") for (int c = 0; c < DS_B; c += 2) {                                                            \
        const int a0 = 18 * (ds_byte(w, c + SH) + ds_byte(w, c + 6 + SH)) + 34 * (ds_byte(w, c + 1 + SH) + ds_byte(w, c + 5 + SH)) + \
                       49 * (ds_byte(w, c + 2 + SH) + ds_byte(w, c + 4 + SH)) + 55 * ds_byte(w, c + 3 + SH);                          \
        int a1 = 0;                                                                                                  \
        if (c + 1 < DS_B)                                                                                            \
            a1 = 18 * (ds_byte(w, c + 1 + SH) + ds_byte(w, c + 7 + SH)) + 34 * (ds_byte(w, c + 2 + SH) + ds_byte(w, c + 6 + SH)) +   \
                 49 * (ds_byte(w, c + 3 + SH) + ds_byte(w, c + 5 + SH)) + 55 * ds_byte(w, c + 4 + SH);                                \
        *reinterpret_cast<uint32_t*>(out + c) = (uint32_t)a0 | ((uint32_t)a1 << 16);                                 \
    }
        if (sh == 0) { DS_HROW(0) } else if (sh == 1) { DS_HROW(1) } else if (sh == 2) { DS_HROW(2) } else { DS_HROW(3) }
#undef DS_HROW
    }
    __syncthreads();
    /* 3b. vertical pass, one lane per patch column */
    if (lane < DS_B) {
        int h[DS_P];
#pragma unroll
        for (int r = 0; r < DS_P; r++) h[r] = hp[r * DS_HS + lane];
#pragma unroll
        for (int r = 0; r < DS_B; r++) {
            const int acc = 18 * (h[r] + h[r + 6]) + 34 * (h[r + 1] + h[r + 5]) + 49 * (h[r + 2] + h[r + 4]) + 55 * h[r + 3];
            bl[r * DS_BS + lane] = (uint8_t)min((acc + (1 << 15)) >> 16, 255);
        }
    }
    __syncthreads();

    /* 4. steered BRIEF, ORBextractor.cpp:52-84 */
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, bsin;
    tbm::sincosf_rn(TB_FMUL(angle, factorPI), &bsin, &a);
    const uint8_t* center = bl + 19 * DS_BS + 19;
    const size_t out = (size_t)b * g.selCap + base + idx;
    unsigned long long* d64 = reinterpret_cast<unsigned long long*>(desc + out * 32);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = j * 64 + lane;
        const float x0f = (float)c_pattern[4 * t], y0f = (float)c_pattern[4 * t + 1];
        const float x1f = (float)c_pattern[4 * t + 2], y1f = (float)c_pattern[4 * t + 3];
        const int r0 = tbm::cv_round(TB_FADD(TB_FMUL(x0f, bsin), TB_FMUL(y0f, a)));
        const int c0 = tbm::cv_round(TB_FSUB(TB_FMUL(x0f, a), TB_FMUL(y0f, bsin)));
        const int r1 = tbm::cv_round(TB_FADD(TB_FMUL(x1f, bsin), TB_FMUL(y1f, a)));
        const int c1 = tbm::cv_round(TB_FSUB(TB_FMUL(x1f, a), TB_FMUL(y1f, bsin)));
        const int t0 = center[r0 * DS_BS + c0], t1 = center[r1 * DS_BS + c1];
        const unsigned long long bits = __ballot(t0 < t1);
        if (lane == 0) d64[j] = bits;
    }

    /* 5. keypoint record; coordinates scaled by sf[level] for level != 0 (ORBextractor.cpp:969-974) */
    if (lane == 0) {
        tb_keypoint kp;
        kp.x = (float)kx;
        kp.y = (float)ky;
        if (level != 0) {
            kp.x = TB_FMUL(kp.x, G.sf);
            kp.y = TB_FMUL(kp.y, G.sf);
        }
        kp.size = G.patchSize;
        kp.angle = angle;
        kp.response = (float)(rec >> 24);
        kp.octave = level;
        kp.class_id = -1;
        kps[out] = kp;
    }
}

int tbk_describe(tb_extractor* ex, int n) {
    tb_ctx* ctx = ex->ctx;
    dim3 grid(ex->g.selCap, n);
    tb_prof_begin(ctx, "k_describe");
    hipLaunchKernelGGL(k_describe, grid, dim3(64), 0, ctx->stream, ex->g, ex->d_slab, ex->d_sel, ex->d_selCount,
                       ex->d_kps, ex->d_desc, ex->d_counts);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
