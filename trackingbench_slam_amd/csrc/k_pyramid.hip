/* a2 -- image pyramid: cv::resize 8UC1 INTER_LINEAR chain of Frame::ComputePyramid
 * (reference src/types/Frame.cpp:414-427), OpenCV 3.3 fixed-point arithmetic.
 *
 * Roofline: HBM-bound, algorithmic bytes = src px + dst px per level.
 * The source-index / 11-bit coefficient tables are built once per plan on the host with the same
 * double/float arithmetic as the CPU restatement, so the kernel is pure integer work:
 *   D = ((b0*((S0[sx]*a0+S0[sx1]*a1)>>4))>>16) + ((b1*((S1[sx]*a0+S1[sx1]*a1)>>4))>>16) + 2) >> 2.
 * One thread produces 4 consecutive destination pixels (one 32-bit store); consecutive lanes cover
 * consecutive 16-byte... 4-byte groups of a row, so stores are fully coalesced and the two source rows
 * are read as contiguous ~5*64-byte spans per wave (L1/L2 hits for the second tap).
 */
#include "tb_internal.h"
#include "tb_device.h"

__global__ void __launch_bounds__(256)
k_resize(PlanGeom g, uint8_t* __restrict__ slab, const ResizeX* __restrict__ rx, const ResizeY* __restrict__ ry,
         int level) {
    const int b = blockIdx.y;
    const LevelGeom& D = g.lv[level];
    const int groupsPerRow = D.stride >> 2;
    const int group = blockIdx.x * blockDim.x + threadIdx.x;
    if (group >= groupsPerRow * D.h) return;
    const int dy = group / groupsPerRow;
    const int dx0 = (group - dy * groupsPerRow) << 2;

    int sstride;
    const uint8_t* src = tb_level_ptr(g, slab, b, level - 1, &sstride);
    uint8_t* dst = slab + (size_t)b * g.slabBytes + D.off + (size_t)dy * D.stride;

    const ResizeY yy = ry[dy];
    const uint8_t* S0 = src + (size_t)yy.sy0 * sstride;
    const uint8_t* S1 = src + (size_t)yy.sy1 * sstride;
    const int b0 = yy.b0, b1 = yy.b1;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = dx0 + i;
        if (dx < D.w) {
            const ResizeX xx = rx[dx];
            const int r0 = S0[xx.sx] * xx.a0 + S0[xx.sx1] * xx.a1;
            const int r1 = S1[xx.sx] * xx.a0 + S1[xx.sx1] * xx.a1;
            int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            packed |= (uint32_t)v << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + dx0) = packed;
}

/* LDS-staged variant: one wavefront per tile of 256 x RS_ROWS destination pixels. Every source row the
 * tile needs is copied once, as aligned dwords (coalesced), into LDS; the wave's four coefficient entries per
 * lane stay in registers across the tile's rows, and the 16 taps of every output group come from LDS instead
 * of 16 scattered global byte loads. Requires 4-byte aligned source rows; tile footprint given by the launcher. */
#define RS_ROWS 8
#define RS_MAXR 14 /* source rows of a tile kept in registers while they are fetched (8 destination rows at scale >= 0.73) */

__global__ void __launch_bounds__(64)
k_resize_lds(PlanGeom g, uint8_t* __restrict__ slab, const ResizeX* __restrict__ rx, const ResizeY* __restrict__ ry,
             int level, int rowBytes, int nImages, int by_image, int tilesX, int tilesY) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rows[];
    /* workgroup -> (image, tile). by_image (batches): 1-D grid, XCD k (workgroup id mod 8) takes images k, k + 8, ... tile by
     * tile: vertically adjacent tiles share up to 5 of their 13 source rows, through ONE L2 this way. */
    int b, tx, ty;
    if (by_image) {
        const unsigned L = blockIdx.x, j = L >> 3, T = (unsigned)(tilesX * tilesY), grp = j / T, t = j - grp * T;
        b = (int)(grp * 8u + (L & 7u));
        if (b >= nImages) return;
        ty = (int)(t / (unsigned)tilesX); tx = (int)(t - (unsigned)ty * (unsigned)tilesX);
    } else {
        b = blockIdx.z; tx = blockIdx.x; ty = blockIdx.y;
    }
    const int dy0 = ty * RS_ROWS, lane = threadIdx.x;
    const LevelGeom& D = g.lv[level];
    const int X0 = tx * 256;
    int sstride;
    const uint8_t* src = tb_level_ptr(g, slab, b, level - 1, &sstride);
    const int dy1 = min(dy0 + RS_ROWS, D.h) - 1;
    const int xl = min(X0 + 255, D.w - 1);
    const int sxa = rx[X0].sx & ~3, sxb = rx[xl].sx1;
    const int ndw = ((sxb - sxa) >> 2) + 1;
    const int sya = ry[dy0].sy0, syb = ry[dy1].sy1;
    const int nrows = syb - sya + 1;
    if (nrows <= RS_MAXR && ndw <= 128) {
        /* every source dword of the tile in flight at once (a load -> LDS store loop per row is one memory latency per
         * row and wave), then the LDS stores */
        uint32_t v[RS_MAXR][2];
        const uint8_t* S = src + (size_t)sya * sstride + sxa + 4 * lane;
#pragma unroll
        for (int r = 0; r < RS_MAXR; r++) {
            v[r][0] = (r < nrows && lane < ndw) ? *reinterpret_cast<const uint32_t*>(S + (size_t)r * sstride) : 0u;
            v[r][1] = (r < nrows && lane + 64 < ndw) ? *reinterpret_cast<const uint32_t*>(S + (size_t)r * sstride + 256) : 0u;
        }
#pragma unroll
        for (int r = 0; r < RS_MAXR; r++) {
            if (r < nrows && lane < ndw) *reinterpret_cast<uint32_t*>(rows + r * rowBytes + 4 * lane) = v[r][0];
            if (r < nrows && lane + 64 < ndw) *reinterpret_cast<uint32_t*>(rows + r * rowBytes + 4 * lane + 256) = v[r][1];
        }
    } else {
        for (int r = 0; r < nrows; r++) {
            const uint8_t* S = src + (size_t)(sya + r) * sstride + sxa;
            for (int dd = lane; dd < ndw; dd += 64)
                *reinterpret_cast<uint32_t*>(rows + r * rowBytes + 4 * dd) = *reinterpret_cast<const uint32_t*>(S + 4 * dd);
        }
    }
    __syncthreads();
    const int dx0 = X0 + 4 * lane;
    if (dx0 >= D.stride) return;
    int o0[4], o1[4], a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = min(dx0 + i, D.w - 1);
        const ResizeX xx = rx[dx];
        o0[i] = xx.sx - sxa; o1[i] = xx.sx1 - sxa; a0[i] = xx.a0; a1[i] = xx.a1;
    }
    /* horizontal interpolation of one source row at the lane's four columns: (S[sx] a0 + S[sx1] a1) >> 4 */
    auto hrow = [&](int sy, int (&hh)[4]) {
        const uint8_t* R = rows + (sy - sya) * rowBytes;
#pragma unroll
        for (int i = 0; i < 4; i++) hh[i] = (R[o0[i]] * a0[i] + R[o1[i]] * a1[i]) >> 4;
    };
    /* vertical blend + store of one destination row. No clamp: a0 + a1 and b0 + b1 are 2047..2049 (each coefficient rounded on
     * its own), so h <= 32655 and ((b0 h0) >> 16) + ((b1 h1) >> 16) <= 1020, i.e. the result is 0..255 as it stands */
    auto emit = [&](int dy, int b0, int b1, const int (&h0)[4], const int (&h1)[4]) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (dx0 + i < D.w) packed |= (uint32_t)((((b0 * h0[i]) >> 16) + ((b1 * h1[i]) >> 16) + 2) >> 2) << (8 * i);
        *reinterpret_cast<uint32_t*>(slab + (size_t)b * g.slabBytes + D.off + (size_t)dy * D.stride + dx0) = packed;
    };
    /* two destination rows per trip: at scale 0.8 the second one's upper source row is the first one's lower row four times
     * out of five (wave-uniform test) -- three horizontal passes instead of four (the kernel is bound by vector instructions:
     * 85 % of the SIMD cycles issue one) */
    for (int dy = dy0; dy <= dy1; dy += 2) {
        const ResizeY ya = ry[dy];
        int hA[4], hB[4];
        hrow(ya.sy0, hA);
        hrow(ya.sy1, hB);
        emit(dy, ya.b0, ya.b1, hA, hB);
        if (dy + 1 <= dy1) {
            const ResizeY yb = ry[dy + 1];
            int hC[4];
            if (yb.sy0 == ya.sy1) {
                hrow(yb.sy1, hC);
                emit(dy + 1, yb.b0, yb.b1, hB, hC);
            } else {
                hrow(yb.sy0, hA);
                hrow(yb.sy1, hC);
                emit(dy + 1, yb.b0, yb.b1, hA, hC);
            }
        }
    }
}

int tbk_resize_level(tb_extractor* ex, int level, int n) {
    tb_ctx* ctx = ex->ctx;
    const LevelGeom& D = ex->g.lv[level];
    const LevelGeom& Sg = ex->g.lv[level - 1];
    /* source alignment check for the LDS-staged kernel */
    const bool ext0 = (level - 1 == 0) && ex->g.img0 != nullptr;
    const int sstride = ext0 ? ex->g.img0_stride : Sg.stride;
    const uintptr_t sbase = ext0 ? reinterpret_cast<uintptr_t>(ex->g.img0) : 0;
    const unsigned long long spitch = ext0 ? ex->g.img0_pitch : 0;
    const bool aligned = (sstride % 4 == 0) && (sbase % 4 == 0) && (spitch % 4 == 0) && (sstride >= ((Sg.w + 3) & ~3));
    const double rx_ratio = (double)Sg.w / (double)D.w, ry_ratio = (double)Sg.h / (double)D.h;
    const int rowBytes = (((int)(256.0 * rx_ratio) + 16 + 3) & ~3) + 4;
    const int srows = (int)(RS_ROWS * ry_ratio) + 3;
    const size_t lds = (size_t)rowBytes * srows;
    if (aligned && lds <= 48 * 1024) {
        const int tilesX = (D.stride + 255) / 256, tilesY = (D.h + RS_ROWS - 1) / RS_ROWS;
        const int by_image = n >= 64 ? 1 : 0;
        const dim3 grid = by_image ? dim3((unsigned)(tilesX * tilesY) * 8u * (unsigned)((n + 7) / 8)) : dim3(tilesX, tilesY, n);
        tb_prof_begin(ctx, "k_resize");
        hipLaunchKernelGGL(k_resize_lds, grid, dim3(64), lds, ctx->stream, ex->g, ex->d_slab, ex->d_rx[level], ex->d_ry[level], level,
                           rowBytes, n, by_image, tilesX, tilesY);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
        return TB_OK;
    }
    const int groups = (D.stride >> 2) * D.h;
    dim3 grid((groups + 255) / 256, n);
    tb_prof_begin(ctx, "k_resize");
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, ex->ctx->stream, ex->g, ex->d_slab, ex->d_rx[level],
                       ex->d_ry[level], level);
    tb_prof_end(ctx);
    TB_HIP(ex->ctx, hipGetLastError());
    return TB_OK;
}

/* ---- measurement helper ("confirm on the box with a device-to-device copy microbench"): 16 bytes per lane, four
 * independent loads in flight per thread, one workgroup per 16 KB tile (no loop: the dispatcher keeps the CUs fed) -- the float4
 * copy the hardware guide quotes 6.29 TB/s for. A grid-stride loop over 8 workgroups per CU with one load in flight measured
 * 5.06 TB/s. bench.py reports what it measures with this next to the 8 TB/s spec peak. */
typedef unsigned int tb_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
k_copy16(const tb_u4* __restrict__ src, tb_u4* __restrict__ dst, size_t n16) {
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x;
    tb_u4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (i0 + 256 * k < n16) ? __builtin_nontemporal_load(src + i0 + 256 * k) : (tb_u4){0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (i0 + 256 * k < n16) __builtin_nontemporal_store(v[k], dst + i0 + 256 * k);
}

int tbk_copy16(tb_ctx* ctx, const void* d_src, void* d_dst, size_t bytes) {
    const size_t n16 = bytes / 16;
    if (n16 == 0) return TB_OK;
    const unsigned blocks = (unsigned)((n16 + 1023) / 1024);
    hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, ctx->stream, (const tb_u4*)d_src, (tb_u4*)d_dst, n16);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

/* ---- exchange helper: the live rows of a [F][cap] record array, frame after frame, to the front of a packed
 * array -- what a rank sends in the track gather instead of capacity-sized tensors. One workgroup per frame: its offset is
 * the sum of the counts before it (F is a few hundred: every workgroup adds them up itself), then 16 bytes per lane.
 * row_bytes is a multiple of 4 (28, 32, 16). total_out[0] = all live rows. */
__global__ void __launch_bounds__(256)
k_pack_rows(const uint8_t* __restrict__ src, int row_bytes, int cap, const int32_t* __restrict__ counts, int nframes,
            uint8_t* __restrict__ dst, long long* __restrict__ total_out) {
    __shared__ long long red[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    long long before = 0, all = 0;
    for (int i = tid; i < nframes; i += 256) {
        const long long c = min(max(counts[i], 0), cap);
        all += c;
        if (i < f) before += c;
    }
    for (int pass = 0; pass < 2; pass++) {
        long long v = pass == 0 ? before : all;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        v = red[0] + red[1] + red[2] + red[3];
        if (pass == 0) before = v; else all = v;
    }
    if (f == 0 && tid == 0 && total_out) total_out[0] = all;
    const int n = min(max(counts[f], 0), cap);
    const size_t nb = (size_t)n * row_bytes;
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + (size_t)f * cap * row_bytes);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + (size_t)before * row_bytes);
    for (size_t i = tid; i < nb / 4; i += 256) d4[i] = s4[i];
}

int tbk_pack_rows(tb_ctx* ctx, const void* d_src, int row_bytes, int cap, const int32_t* d_counts, int nframes, void* d_dst, long long* d_total) {
    if (nframes <= 0) return TB_OK;
    hipLaunchKernelGGL(k_pack_rows, dim3(nframes), dim3(256), 0, ctx->stream, (const uint8_t*)d_src, row_bytes, cap, d_counts, nframes,
                       (uint8_t*)d_dst, d_total);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
