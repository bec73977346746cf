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

int tbk_resize_level(tb_extractor* ex, int level, int n) {
    tb_ctx* ctx = ex->ctx;
    const LevelGeom& D = ex->g.lv[level];
    const int groups = (D.stride >> 2) * D.h;
    dim3 grid((groups + 255) / 256, n);
    tb_prof_begin(ctx, "k_resize");
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, ex->ctx->stream, ex->g, ex->d_slab, ex->d_rx[level],
                       ex->d_ry[level], level);
    tb_prof_end(ctx);
    TB_HIP(ex->ctx, hipGetLastError());
    return TB_OK;
}
