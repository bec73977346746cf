/* Micro-benchmark: issue rate of the vector instructions the hot loops are made of (gfx950, per SIMD, 16 independent chains, four
 * wavefronts per SIMD). Result (profiles/r02n_ubench_int_ops.json): plain 32-bit integer / logic / shift instructions run at about
 * 1.7x the rate of packed 16-bit, 24-bit multiply-add, dot-product and 64-bit ones -- an "instruction" is not a unit of cost.
 *   hipcc -O3 --offload-arch=gfx950 tools/ubench/int_ops.hip -o tools/ubench/int_ops && tools/ubench/int_ops */
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
#define N_CH 16

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* out, unsigned a32, unsigned long long a64) {
    unsigned v[N_CH];
    unsigned long long w[N_CH];
    for (int i = 0; i < N_CH; i++) { v[i] = threadIdx.x + i; w[i] = threadIdx.x * 77ull + i; }
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < N_CH; i++) {
            if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(a64));
            if (MODE == 2) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(v[i]) : "v"(a32));
            if (MODE == 5) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 7) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 8) asm volatile("v_dot4_u32_u8 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 9) asm volatile("v_dot2_u32_u16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 10) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 13) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 14) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 16) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 17) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 18) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[i]) : "v"(a32));
            if (MODE == 20) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(v[i]) : "v"(a32));
            if (MODE == 21) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 23) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 24) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]) : "v"(a32));
            if (MODE == 25) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i]) : "v"(a32));
            if (MODE == 26) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a32) : "vcc");
            if (MODE == 27) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 28) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a32));
            if (MODE == 29) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a32));
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < N_CH; i++) s += v[i] + w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(unsigned long long* d_out, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 3u, 5ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 3u, 5ull);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 4;
    unsigned long long* d_out;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(unsigned long long));
    const char* names[30] = {"v_add_u32", "v_lshl_add_u64", "v_pk_minimum3_f16", "v_mad_u32_u24", "v_bitop3_b32", "v_pk_min_i16", "v_alignbyte_b32", "v_perm_b32", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_bcnt_u32_b32", "v_xor_b32", "v_mul_u32_u24", "v_lshl_or_b32", "v_min3_u32", "v_min_u32", "v_and_or_b32", "v_lshl_add_u32", "v_add3_u32", "v_lshrrev_b32", "v_bfe_u32", "v_med3_i32", "v_mul_f32", "v_fma_f32", "v_rndne_f32", "v_cvt_i32_f32", "v_cndmask_b32", "v_mul_lo_u32", "v_sad_u8", "v_mov_b32_dpp"};
    float t[30] = {run<0>(d_out, blocks), run<1>(d_out, blocks), run<2>(d_out, blocks), run<3>(d_out, blocks), run<4>(d_out, blocks), run<5>(d_out, blocks), run<6>(d_out, blocks), run<7>(d_out, blocks), run<8>(d_out, blocks), run<9>(d_out, blocks), run<10>(d_out, blocks), run<11>(d_out, blocks), run<12>(d_out, blocks), run<13>(d_out, blocks), run<14>(d_out, blocks), run<15>(d_out, blocks), run<16>(d_out, blocks), run<17>(d_out, blocks), run<18>(d_out, blocks), run<19>(d_out, blocks), run<20>(d_out, blocks), run<21>(d_out, blocks), run<22>(d_out, blocks), run<23>(d_out, blocks), run<24>(d_out, blocks), run<25>(d_out, blocks), run<26>(d_out, blocks), run<27>(d_out, blocks), run<28>(d_out, blocks), run<29>(d_out, blocks)};
    printf("{");
    for (int i = 0; i < 30; i++) {
        const double winst = (double)N_IT * N_CH * blocks * 4;                 /* wave-instructions */
        printf("\"%s\": %.0f%s", names[i], winst / 1024.0 / (t[i] * 1e3), i < 29 ? ", " : "");
    }
    printf("}\n");   /* wave-instructions per SIMD per microsecond */
    return 0;
}
