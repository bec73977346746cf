/* Micro-benchmark behind DESIGN.md section 4 (round 3): v_mfma_f64_4x4x4_4b_f64 on gfx950 -- four independent 4x4x4 FP64
 * block products per instruction, one double of A, B and C/D per lane.
 *   1. operand / result lane maps, found by probing: A one-hot in lane x, B = lane + 1 -> which (output lane, B lane) pairs
 *      light up;  checked against  A[blk][i][k] in lane 16 blk + 4 k + i,  B[blk][k][j] in lane 16 blk + 4 k + j,
 *      D[blk][i][j] in lane 16 blk + 4 i + j  (and the transposed alternative).
 *   2. issue rate against v_mfma_f64_16x16x4 (2048 flop) : 512 flop per instruction.
 *   3. the same loop with one or two ds_read_b64 per MFMA (operands from LDS), the shape of the Schur kernel's inner loop.
 *   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f64_4x4.hip -o tools/ubench/mfma_f64_4x4 && tools/ubench/mfma_f64_4x4 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef double d4 __attribute__((ext_vector_type(4)));
#define N_IT 4096
#define N_ACC 8

__global__ void k_probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

/* DPP row rotate: which lane does lane x read with row_ror:n? (the Schur kernel rotates the four row blocks of an operand) */
__global__ void k_ror(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x120 + 4, 0xf, 0xf, false);       /* row_ror:4 */
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x120 + 12, 0xf, 0xf, false); /* row_ror:12 */
}

template <int MODE> /* 0 = 4x4x4 from registers, 1 = 16x16x4 from registers, 2 = 4x4x4 + one LDS read each, 3 = 4x4x4 + two LDS reads each */
__global__ void __launch_bounds__(256) k_rate(double* out, double a, double b) {
    __shared__ double sh[4 * 64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 8; i += 256) sh[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    const double* mine = sh + wave * 512;
    double acc1[N_ACC];
    d4 acc4[N_ACC];
    for (int i = 0; i < N_ACC; i++) { acc1[i] = 0; acc4[i] = (d4){0, 0, 0, 0}; }
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < N_ACC; i++) {
            if (MODE == 0) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
            if (MODE == 1) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
            if (MODE == 2) {
                const double x = mine[((it + i) & 7) * 64 + lane];
                acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b, acc1[i], 0, 0, 0);
            }
            if (MODE == 3) {
                const double x = mine[((it + i) & 7) * 64 + lane], y = mine[((it + i + 3) & 7) * 64 + (lane ^ 16)];
                acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc1[i], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < N_ACC; i++) s += acc1[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(double* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *da, *db, *dd, ha[64], hb[64], hd[64];
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    /* probe: triples (output lane, A lane, B lane) */
    int outA[64][4], outB[64][4], nfound[64];
    memset(nfound, 0, sizeof nfound);
    for (int x = 0; x < 64; x++) {
        for (int l = 0; l < 64; l++) { ha[l] = (l == x) ? 1.0 : 0.0; hb[l] = l + 1; }
        hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++)
            if (hd[l] != 0 && nfound[l] < 4) { outA[l][nfound[l]] = x; outB[l][nfound[l]] = (int)hd[l] - 1; nfound[l]++; }
    }
    int okN = 1, okT = 1;
    for (int l = 0; l < 64; l++) {
        if (nfound[l] != 4) { okN = okT = 0; continue; }
        for (int t = 0; t < 4; t++) {
            const int al = outA[l][t], bl = outB[l][t];
            /* measured on gfx950: A[blk][i][k] in lane 16 k + 4 blk + i, B[blk][k][j] in lane 16 k + 4 blk + j,
               D[blk][i][j] in lane 16 i + 4 blk + j (okN); okT = the transposed output, which the hardware does not use */
            const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
            const int ak = al >> 4, ablk = (al >> 2) & 3, ai = al & 3, bk = bl >> 4, bblk = (bl >> 2) & 3, bj = bl & 3;
            const bool sameblk = ablk == blk && bblk == blk && ak == bk;
            if (!(sameblk && ai == i && bj == j)) okN = 0;
            if (!(sameblk && ai == j && bj == i)) okT = 0;
        }
    }
    printf("{\"layout_A_16k_4blk_i__B_16k_4blk_j__D_16i_4blk_j\": %s, \"layout_D_transposed\": %s,\n", okN ? "true" : "false", okT ? "true" : "false");
    if (!okN && !okT) {
        printf(" \"triples\": [");
        for (int l = 0; l < 64; l++) { printf("[%d", l); for (int t = 0; t < nfound[l]; t++) printf(",%d,%d", outA[l][t], outB[l][t]); printf("]%s", l < 63 ? "," : ""); }
        printf("],\n");
    }
    {
        int *dr, hr[128];
        hipMalloc(&dr, 512);
        hipLaunchKernelGGL(k_ror, dim3(1), dim3(64), 0, 0, dr);
        hipMemcpy(hr, dr, 512, hipMemcpyDeviceToHost);
        printf(" \"row_ror4_lane0_5_17_reads\": [%d, %d, %d], \"row_ror12_lane0_5_17_reads\": [%d, %d, %d],\n", hr[0], hr[5], hr[17], hr[64], hr[64 + 5], hr[64 + 17]);
    }
    const int blocks = 256 * 2;
    double* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(double));
    const float t0 = run<0>(d_out, blocks), t1 = run<1>(d_out, blocks), t2 = run<2>(d_out, blocks), t3 = run<3>(d_out, blocks);
    const double n = (double)N_ACC * N_IT * blocks * 4; /* wave-instructions */
    printf(" \"mfma4x4x4_ms\": %.4f, \"mfma4x4x4_tflops\": %.1f, \"mfma16x16x4_ms\": %.4f, \"mfma16x16x4_tflops\": %.1f,\n"
           " \"mfma4x4x4_1lds_ms\": %.4f, \"mfma4x4x4_1lds_tflops\": %.1f, \"mfma4x4x4_2lds_ms\": %.4f, \"mfma4x4x4_2lds_tflops\": %.1f,\n"
           " \"clocks_per_4x4x4_at_2p4GHz\": %.1f}\n",
           t0, 512 * n / t0 / 1e9, t1, 2048 * n / t1 / 1e9, t2, 512 * n / t2 / 1e9, t3, 512 * n / t3 / 1e9,
           t0 * 1e-3 * 2.4e9 / (n / 1024.0));
    return 0;
}
