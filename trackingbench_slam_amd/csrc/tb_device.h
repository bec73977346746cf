/* Device-side helpers shared by the kernels (gfx950: 64-lane wavefronts). */
#ifndef TB_DEVICE_H
#define TB_DEVICE_H

#include "tb_internal.h"
#include "tb_math.h"

#define TB_WAVE 64

__device__ __forceinline__ const uint8_t* tb_level_ptr(const PlanGeom& g, const uint8_t* slab, int b, int l,
                                                       int* stride) {
    if (l == 0 && g.img0 != nullptr) {
        *stride = g.img0_stride;
        return g.img0 + (size_t)b * g.img0_pitch;
    }
    *stride = g.lv[l].stride;
    return slab + (size_t)b * g.slabBytes + g.lv[l].off;
}

__device__ __forceinline__ int tb_lane() { return threadIdx.x & (TB_WAVE - 1); }

/* inclusive wave scan (sum) */
__device__ __forceinline__ int tb_wave_incl_scan(int v) {
    const int lane = tb_lane();
#pragma unroll
    for (int d = 1; d < TB_WAVE; d <<= 1) {
        int t = __shfl_up(v, d, TB_WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

/* inclusive wave scan (sum) on the DPP path: Hillis-Steele inside every row of 16 lanes (row_shr 1, 2, 4, 8; lanes
 * shifted in from outside the row read 0), then the row totals travel down with row_bcast 15 / 31. Six vector adds, no
 * LDS round trips; lane 63 ends up with the wave total. */
__device__ __forceinline__ int tb_wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1, 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2, 3 */
    return v;
}
/* bitwise OR over the wave, result in lane 63 (same network) */
__device__ __forceinline__ int tb_wave_or_dpp(int v) {
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ int tb_wave_sum(int v) {
#pragma unroll
    for (int d = TB_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, TB_WAVE);
    return v;
}

__device__ __forceinline__ int tb_wave_max_i(int v) {
#pragma unroll
    for (int d = TB_WAVE / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, TB_WAVE));
    return v;
}

/* In-place exclusive scan of arr[0..n) in LDS by the whole block; returns the total.
 * tmp: LDS, >= (blockDim.x/64 + 1) ints.  Deterministic (fixed chunking).  Ends with a barrier. */
__device__ inline int tb_block_excl_scan(int* arr, int n, int* tmp) {
    const int T = blockDim.x, tid = threadIdx.x;
    const int per = (n + T - 1) / T;
    const int beg = min(tid * per, n), end = min(beg + per, n);
    int s = 0;
    for (int i = beg; i < end; i++) s += arr[i];
    const int incl = tb_wave_incl_scan(s);
    const int wave = tid >> 6, lane = tid & 63, nw = T >> 6;
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < nw; w++) { int t = tmp[w]; tmp[w] = acc; acc += t; }
        tmp[nw] = acc;
    }
    __syncthreads();
    int run = tmp[wave] + incl - s;
    for (int i = beg; i < end; i++) { int t = arr[i]; arr[i] = run; run += t; }
    const int total = tmp[nw];
    __syncthreads();
    return total;
}

#endif
