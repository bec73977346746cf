/* Micro-benchmark behind DESIGN.md section 4: do the FP64 matrix (v_mfma_f64_16x16x4) and FP64 vector (v_fma_f64) pipes
 * of gfx950 overlap?  Three kernels with the same loop count and full occupancy of one wave per SIMD x WAVES:
 * MFMA only, vector FMA only, both interleaved (independent dependency chains).  If the pipes were independent the mixed
 * kernel would take max(a, b); if they share the FP64 datapath it takes a + b.
 *   hipcc -O3 --offload-arch=gfx950 tools/ubench/fp64_pipes.hip -o /tmp/fp64_pipes && /tmp/fp64_pipes */
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
#define N_IT 4096
#define N_MFMA 8  /* independent accumulators */
#define N_FMA 16  /* independent vector chains */

template <int MODE> /* 1 = MFMA, 2 = vector FMA, 3 = both */
__global__ void __launch_bounds__(256) k(double* out, double a, double b) {
    d4 acc[N_MFMA];
    double v[N_FMA];
    for (int i = 0; i < N_MFMA; i++) acc[i] = (d4){0, 0, 0, 0};
    for (int i = 0; i < N_FMA; i++) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < N_IT; it++) {
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < N_MFMA; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < N_FMA; i++) v[i] = fma(v[i], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < N_MFMA; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < N_FMA; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(double* d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 2; /* two 4-wave blocks per CU: two waves per SIMD */
    double* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(double));
    const float tm = run<1>(d_out, blocks), tv = run<2>(d_out, blocks), tb = run<3>(d_out, blocks);
    const double mf = 2.0 * 16 * 16 * 4 * N_MFMA * (double)N_IT * blocks * 4; /* flops: per wave-MFMA 2*16*16*4 */
    const double vf = 2.0 * 64 * N_FMA * (double)N_IT * blocks * 4;
    printf("{\"mfma_only_ms\": %.4f, \"mfma_tflops\": %.1f, \"vector_only_ms\": %.4f, \"vector_tflops\": %.1f, "
           "\"both_ms\": %.4f, \"sum_ms\": %.4f, \"max_ms\": %.4f}\n",
           tm, mf / tm / 1e9, tv, vf / tv / 1e9, tb, tm + tv, tm > tv ? tm : tv);
    hipFree(d_out);
    return 0;
}
