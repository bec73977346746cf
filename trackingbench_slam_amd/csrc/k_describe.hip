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
#define DS_PS 48     /* source patch row stride */
#define DS_B 39      /* blurred patch edge */
#define DS_BS 40     /* blurred / h-pass row stride */

__device__ __forceinline__ int ds_reflect(int i, int n) {
    /* BORDER_REFLECT_101; n >= 2 and |overshoot| < n for every level that can hold a keypoint */
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__global__ void __launch_bounds__(64)
k_describe(PlanGeom g, const uint8_t* __restrict__ slab, const uint32_t* __restrict__ sel,
           const int32_t* __restrict__ selCount, tb_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
           int32_t* __restrict__ counts) {
    __shared__ __attribute__((aligned(16))) uint8_t src[DS_P * DS_PS];
    __shared__ __attribute__((aligned(16))) unsigned short hp[DS_P * DS_BS];
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

    /* 1. stage the source patch */
    for (int e = lane; e < DS_P * DS_P; e += 64) {
        const int r = e / DS_P, c = e - r * DS_P;
        const int yy = ds_reflect(ky - 22 + r, G.h), xx = ds_reflect(kx - 22 + c, G.w);
        src[r * DS_PS + c] = img[(size_t)yy * stride + xx];
    }
    __syncthreads();

    /* 2. IC_Angle: integer moments over the radius-15 disc */
    int m10 = 0, m01 = 0;
    for (int e = lane; e < 31 * 31; e += 64) {
        const int v = e / 31 - 15, u = e - (e / 31) * 31 - 15;
        const int av = v < 0 ? -v : v;
        /* umax[] of ORBextractor.cpp:389-404 = 15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3 */
        const unsigned long long UM = 0x3689ABCDDEEEFFFFull; /* nibble av = umax[av] */
        const int um = (int)((UM >> (4 * av)) & 0xf);
        const int au = u < 0 ? -u : u;
        if (au <= um) {
            const int I = src[(22 + v) * DS_PS + 22 + u];
            m10 += u * I;
            m01 += v * I;
        }
    }
    m10 = tb_wave_sum(m10);
    m01 = tb_wave_sum(m01);
    const float angle = tbm::fast_atan2((float)m01, (float)m10);

    /* 3. separable 7x7 blur, exact integers */
    for (int e = lane; e < DS_P * DS_B; e += 64) {
        const int r = e / DS_B, c = e - r * DS_B;
        const uint8_t* p = src + r * DS_PS + c;
        const int acc = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3];
        hp[r * DS_BS + c] = (unsigned short)acc;
    }
    __syncthreads();
    for (int e = lane; e < DS_B * DS_B; e += 64) {
        const int r = e / DS_B, c = e - r * DS_B;
        const unsigned short* p = hp + r * DS_BS + c;
        const int acc = 18 * ((int)p[0] + p[6 * DS_BS]) + 34 * ((int)p[DS_BS] + p[5 * DS_BS]) +
                        49 * ((int)p[2 * DS_BS] + p[4 * DS_BS]) + 55 * (int)p[3 * DS_BS];
        const int v = (acc + (1 << 15)) >> 16;
        bl[r * DS_BS + c] = (uint8_t)min(v, 255);
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
        const float x0 = (float)c_pattern[4 * t], y0 = (float)c_pattern[4 * t + 1];
        const float x1 = (float)c_pattern[4 * t + 2], y1 = (float)c_pattern[4 * t + 3];
        const int r0 = tbm::cv_round(TB_FADD(TB_FMUL(x0, bsin), TB_FMUL(y0, a)));
        const int c0 = tbm::cv_round(TB_FSUB(TB_FMUL(x0, a), TB_FMUL(y0, bsin)));
        const int r1 = tbm::cv_round(TB_FADD(TB_FMUL(x1, bsin), TB_FMUL(y1, a)));
        const int c1 = tbm::cv_round(TB_FSUB(TB_FMUL(x1, a), TB_FMUL(y1, bsin)));
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
