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

/* bit_pattern_31_ (ORBextractor.cpp:90-348) as floats, four per test (x0, y0, x1, y1): one 16-byte load per lane */
struct DsPattern { float v[1024]; };
constexpr DsPattern ds_make_pattern() {
    constexpr int8_t raw[1024] = {
#include "orb_pattern.inc"
    };
    DsPattern p{};
    for (int i = 0; i < 1024; i++) p.v[i] = (float)raw[i];
    return p;
}
__constant__ __attribute__((aligned(16))) DsPattern c_patternf = ds_make_pattern();

/* IC_Angle disc (umax[] of ORBextractor.cpp:389-404 = 15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3) as byte weights for
 * v_dot4_u32_u8: row |v| of the disc spans u in [-umax, umax]; byte i of the 32 bytes that start at u = -15 weighs
 * 1 (sum of intensities) and u + 16 (first moment, biased to stay unsigned) inside the disc, 0 outside. */
struct DsDisc { uint32_t w1[16][8], wu[16][8]; };
constexpr DsDisc ds_make_disc() {
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    DsDisc d{};
    for (int av = 0; av < 16; av++)
        for (int i = 0; i < 32; i++) {
            const int u = i - 15;
            const bool in = (u >= -umax[av]) && (u <= umax[av]);
            d.w1[av][i >> 2] |= (uint32_t)(in ? 1 : 0) << (8 * (i & 3));
            d.wu[av][i >> 2] |= (uint32_t)(in ? (u + 16) : 0) << (8 * (i & 3));
        }
    return d;
}
__constant__ __attribute__((aligned(16))) DsDisc c_disc = ds_make_disc();

#define DS_P 45      /* source patch edge */
#define DS_PS 84     /* source patch row stride: 21 dwords (odd) -> one-row-per-lane accesses hit distinct banks; = the h-pass row (below) */
#define DS_B 39      /* blurred patch edge */
#define DS_HS 42     /* h-pass row stride in u16 (21 dwords, odd) */
#define DS_BS 40     /* blurred patch row stride */

typedef unsigned short ds_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int ds_reflect(int i, int n) {
    /* BORDER_REFLECT_101; n >= 2 and |overshoot| < n for every level that can hold a keypoint */
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

/* four bytes of a register array starting at byte o (compile-time o) */
__device__ __forceinline__ uint32_t ds_window(const uint32_t* w, int o) {
    return (o & 3) ? __builtin_amdgcn_alignbyte(w[(o >> 2) + 1], w[o >> 2], o & 3) : w[o >> 2];
}

__device__ __forceinline__ uint32_t ds_udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ds_u16x2, a), __builtin_bit_cast(ds_u16x2, b), c, false);
}

/* horizontal 7-tap pass of one patch row held in 13 dwords, patch column 0 at byte SH: 39 exact u16 sums
 * (18,34,49,55,49,34,18; at most 255 * 257 = 65535) as two v_dot4_u32_u8 each */
template <int SH>
__device__ __forceinline__ void ds_hrow(const uint32_t* w, unsigned short* out) {
    constexpr uint32_t K1 = 18u | (34u << 8) | (49u << 16) | (55u << 24), K2 = 49u | (34u << 8) | (18u << 16);
    uint32_t a[DS_B + 1];
#pragma unroll
    for (int c = 0; c < DS_B; c++)
        a[c] = __builtin_amdgcn_udot4(ds_window(w, c + SH + 4), K2, __builtin_amdgcn_udot4(ds_window(w, c + SH), K1, 0u, false), false);
    a[DS_B] = 0;
#pragma unroll
    for (int c = 0; c < DS_B; c += 2) *reinterpret_cast<uint32_t*>(out + c) = a[c] | (a[c + 1] << 16);
}

/* vertical 7-tap pass of N outputs of one column: h[i] = u16 sums of rows r0 + i (N + 6 of them), exact 32-bit
 * accumulation as three v_dot2_u32_u16 and one multiply-add, (acc + 2^15) >> 16 saturated (the kernel sums to 257) */
template <int N>
__device__ __forceinline__ void ds_vcol(const uint32_t* h, uint8_t* out) {
    constexpr uint32_t KA = 18u | (34u << 16), KB = 49u | (55u << 16), KC = 49u | (34u << 16);
    uint32_t pr[N + 5];
#pragma unroll
    for (int i = 0; i < N + 5; i++) pr[i] = h[i] | (h[i + 1] << 16);
#pragma unroll
    for (int j = 0; j < N; j++) {
        uint32_t acc = 18u * h[j + 6] + (1u << 15);
        acc = ds_udot2(pr[j], KA, acc);
        acc = ds_udot2(pr[j + 2], KB, acc);
        acc = ds_udot2(pr[j + 4], KC, acc);
        out[j * DS_BS] = (uint8_t)min(acc >> 16, 255u);   /* rows past the patch land in the spare row of bl[] */
    }
}

/* Instruction budget per keypoint (one wavefront), the quantity this kernel is bound by (~1080 vector instructions in
 * round 1, blur passes 51 % of them on 45 / 39 of the 64 lanes):
 *   staging   interior keypoints copy 45 rows as 13 aligned dwords each (the sub-dword phase of the patch is
 *             kept as a column shift), border keypoints take the per-byte reflect-101 path;
 *   moments   one lane per disc row: eight 4-byte windows against the disc's weight table, 16 v_dot4_u32_u8;
 *   h-pass    one lane per patch ROW: 13 dword reads, 39 outputs of two v_dot4_u32_u8 each (4 multiply-adds per
 *             instruction on the packed bytes), fully unrolled;
 *   v-pass    the 39 x 39 outputs over ALL lanes: columns 0-31 as two half columns of 20 rows per lane, then columns
 *             32-38 as eight 5-row pieces; u16 inputs, three v_dot2_u32_u16 + one multiply-add per output;
 *   tests     4 x 64 rotated comparisons -> 4 ballots. */
#define DS_KPB 4     /* keypoints (wavefronts) per workgroup */

__device__ __forceinline__ void ds_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

__global__ void __launch_bounds__(64 * DS_KPB)
k_describe(PlanGeom g, const uint8_t* __restrict__ slab, const uint32_t* __restrict__ sel,
           const int32_t* __restrict__ selCount, tb_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
           int32_t* __restrict__ counts, int nImages, int by_image, int slotGroups) {
    /* ONE patch buffer per keypoint, three tenants in turn: the source patch (rows of 84 bytes, 52 used), the h-pass sums (u16, the
     * same 84-byte rows: every lane has its whole source row in registers before it stores), the blurred patch (40-byte rows from
     * offset 0: every lane has all its h-pass operands in registers before the first store). 3.8 KB per keypoint instead of 7.7:
     * the kernel's occupancy is bound by LDS (20 -> 32 wavefronts per CU, now the register limit). */
    static_assert(DS_PS == DS_HS * 2, "source rows and h-pass rows share their LDS");
    __shared__ __attribute__((aligned(16))) uint8_t buf_[DS_KPB][DS_P * DS_PS + 16];
    __shared__ int mom[DS_KPB][2];
    __shared__ float rot[DS_KPB][4];
    /* A workgroup = DS_KPB wavefronts = DS_KPB consecutive slots of one image, one keypoint per wavefront with its own patch
     * buffers. The wavefronts meet twice: the orientation of a keypoint (fastAtan2, then sinf / cosf in the double-precision form
     * that matches glibc bit for bit: ~100 wave-uniform vector instructions, an eighth of the kernel when every wavefront ran
     * them on 64 identical lanes) is computed for all DS_KPB keypoints at once on DS_KPB lanes of wavefront 0, while the others are in
     * their blur passes.
     * workgroup -> (image, slot group). by_image (batches): a 1-D grid in which XCD k (workgroup id mod 8) takes images k, k + 8,
     * ... group by group, so that all patch gathers of an image go through ONE L2 (its pyramid, 2.5 MB at 1280x720, fits the
     * 4 MB); otherwise (group, image) order, every XCD works on every image. */
    int b, sg;
    if (by_image) {
        const unsigned L = blockIdx.x, j = L >> 3, grp = j / (unsigned)slotGroups;
        sg = (int)(j - grp * (unsigned)slotGroups);
        b = (int)(grp * 8u + (L & 7u));
        if (b >= nImages) return;
    } else {
        b = blockIdx.y; sg = blockIdx.x;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = sg * DS_KPB + wave;
    uint8_t* const src = buf_[wave];
    unsigned short* const hp = reinterpret_cast<unsigned short*>(buf_[wave]);
    uint8_t* const bl = buf_[wave];
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
    const bool active = slot < g.selCap && idx < sc[level];      /* wave-uniform; idle wavefronts still meet the barriers */
    uint32_t rec = 0;
    int kx = 0, ky = 0, sh = 0;
    if (active) {
        rec = sel[(size_t)b * g.selCap + slot];
        kx = (int)(rec & 0xfff) + TB_BORDER; ky = (int)((rec >> 12) & 0xfff) + TB_BORDER;
        int stride;
        const uint8_t* img = tb_level_ptr(g, slab, b, level, &stride);

        /* 1. stage the source patch: patch column c lives at LDS column c + sh */
        const int x0 = kx - 22, y0 = ky - 22;
        const bool interior = x0 >= 0 && y0 >= 0 && kx + 22 < G.w && ky + 22 < G.h && ((stride & 3) == 0) &&
                              ((reinterpret_cast<uintptr_t>(img) & 3) == 0) && ((x0 & ~3) + 52 <= stride);
        sh = interior ? (x0 & 3) : 0;
        if (interior) {
            const int rr = (lane * 5042) >> 16, dd = lane - rr * 13; /* lane / 13: 4 rows x 13 dwords per pass, lanes 52..63 idle */
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
        ds_wave_fence();

        /* 2. IC_Angle: integer moments over the radius-15 disc, lane = disc row v = lane - 15 */
        int m10 = 0, m01 = 0;
        if (lane < 31) {
            const int v = lane - 15, av = v < 0 ? -v : v;
            const int o = 7 + sh;                                  /* byte of u = -15 in the row (patch column 22 - 15) */
            const uint32_t* rw = reinterpret_cast<const uint32_t*>(src + (22 + v) * DS_PS) + (o >> 2);
            const uint4* t1 = reinterpret_cast<const uint4*>(c_disc.w1[av]);
            const uint4* tu = reinterpret_cast<const uint4*>(c_disc.wu[av]);
            const uint4 a0 = t1[0], a1 = t1[1], u0 = tu[0], u1 = tu[1];
            const uint32_t w1[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const uint32_t wu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            uint32_t w[9];
#pragma unroll
            for (int k = 0; k < 9; k++) w[k] = rw[k];
            uint32_t sI = 0, sW = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t X = __builtin_amdgcn_alignbyte(w[k + 1], w[k], (uint32_t)(o & 3));
                sI = __builtin_amdgcn_udot4(X, w1[k], sI, false);
                sW = __builtin_amdgcn_udot4(X, wu[k], sW, false);
            }
            m10 = (int)sW - 16 * (int)sI;    /* sum u I = sum (u + 16) I - 16 sum I */
            m01 = v * (int)sI;
        }
        m10 = tb_wave_incl_scan_dpp(m10);
        m01 = tb_wave_incl_scan_dpp(m01);
        if (lane == 63) { mom[wave][0] = m10; mom[wave][1] = m01; }
    }
    __syncthreads();

    /* 2b. orientations of the workgroup's keypoints, one lane each (ORBextractor.cpp:43, :52-55) */
    if (wave == 0 && lane < DS_KPB) {
        const float angle = tbm::fast_atan2((float)mom[lane][1], (float)mom[lane][0]);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float a, bsin;
        tbm::sincosf_rn(TB_FMUL(angle, factorPI), &bsin, &a);
        rot[lane][0] = angle; rot[lane][1] = a; rot[lane][2] = bsin;
    }

    if (active) {
        /* 3a. horizontal 7-tap pass, one lane per patch row (exact integers, u16 result) */
        if (lane < DS_P) {
            const uint32_t* rw = reinterpret_cast<const uint32_t*>(src + lane * DS_PS);
            uint32_t w[14];
#pragma unroll
            for (int j = 0; j < 13; j++) w[j] = rw[j];
            w[13] = 0;
            unsigned short* out = hp + lane * DS_HS;
            /* the sub-dword phase of the patch: four code paths with compile-time byte offsets */
            if (sh == 0) ds_hrow<0>(w, out); else if (sh == 1) ds_hrow<1>(w, out); else if (sh == 2) ds_hrow<2>(w, out); else ds_hrow<3>(w, out);
        }
        ds_wave_fence();
        /* 3b. vertical pass over all 64 lanes: all operands first (the outputs overwrite them) */
        {
            /* columns 0..31: lane = (column, upper / lower half): rows [0, 20) and [20, 39) */
            const int c = lane & 31, r0 = 20 * (lane >> 5);
            /* columns 32..38: lane = (column, one of eight 5-row pieces) */
            const int cc = lane & 7, q0 = 5 * (lane >> 3);
            const int c2 = 32 + min(cc, 6);
            uint32_t h[26], h2[11];
#pragma unroll
            for (int i = 0; i < 26; i++) h[i] = hp[min(r0 + i, DS_P - 1) * DS_HS + c];
#pragma unroll
            for (int i = 0; i < 11; i++) h2[i] = hp[min(q0 + i, DS_P - 1) * DS_HS + c2];
            ds_wave_fence();
            ds_vcol<20>(h, bl + r0 * DS_BS + c);          /* the lower half's 20th output is row 39: the spare row */
            ds_vcol<5>(h2, bl + q0 * DS_BS + c2);         /* lanes with cc == 7 repeat column 38 (same values); rows reach 39 at most */
        }
    }
    __syncthreads();
    if (!active) return;

    /* 4. steered BRIEF, ORBextractor.cpp:52-84 */
    const float angle = rot[wave][0], a = rot[wave][1], bsin = rot[wave][2];
    const uint8_t* center = bl + 19 * DS_BS + 19;
    const size_t out = (size_t)b * g.selCap + base + idx;
    unsigned long long* d64 = reinterpret_cast<unsigned long long*>(desc + out * 32);
    const float4* pat = reinterpret_cast<const float4*>(c_patternf.v);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float4 pt = pat[j * 64 + lane];
        const float x0f = pt.x, y0f = pt.y, x1f = pt.z, y1f = pt.w;
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
    const int by_image = n >= 64 ? 1 : 0;
    const int slotGroups = (ex->g.selCap + DS_KPB - 1) / DS_KPB;
    const dim3 grid = by_image ? dim3((unsigned)slotGroups * 8u * (unsigned)((n + 7) / 8)) : dim3(slotGroups, n);
    tb_prof_begin(ctx, "k_describe");
    hipLaunchKernelGGL(k_describe, grid, dim3(64 * DS_KPB), 0, ctx->stream, ex->g, ex->d_slab, ex->d_sel, ex->d_selCount,
                       ex->d_kps, ex->d_desc, ex->d_counts, n, by_image, slotGroups);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
