// Issue rate of the VALU instruction classes the blend kernels are made of (instructions per SIMD-cycle, wave64).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kUnroll = 8;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed) {
    float a[kUnroll];
    f32x2 p[kUnroll];
    for (int i = 0; i < kUnroll; ++i) { a[i] = seed + threadIdx.x + i; p[i] = (f32x2){a[i], a[i] + 1.f}; }
    const float m = 1.0000001f, c = 1e-9f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (MODE == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % kUnroll]));
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            if (MODE == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % kUnroll]));
            if (MODE == 3) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 4) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (MODE == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (MODE == 6) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) % kUnroll]));
            if (MODE == 7) p[i] = p[i] * m2;
            if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(0xAAAAAAAAAAAAAAAAull));
            if (MODE == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < kUnroll; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 12345.f) out[0] = s;
}

template <int MODE>
double run(const char* name, float* d) {
    const int blocks = 256 * 8;     // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    rate_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)kIters * kUnroll * 8;          // 8 waves per SIMD
    printf("%-28s %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (4 cycles @2.4GHz = 1.67 ns)\n", name, ms,
           ms * 1e6 / instr_per_simd);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32 (asm)", d);
    run<10>("v_mul_f32 (asm)", d);
    run<11>("v_pk_fma_f32 (asm)", d);
    run<12>("v_mov_b32 (asm)", d);
    run<1>("v_pk_fma_f32", d);
    run<7>("v_pk_mul_f32", d);
    run<2>("v_permlane32_swap_b32", d);
    run<6>("v_permlane16_swap_b32", d);
    run<3>("v_add_f32_dpp row_ror", d);
    run<4>("v_exp_f32", d);
    run<5>("v_rcp_f32", d);
    run<8>("v_cndmask_b32", d);
    run<9>("v_med3_f32", d);
    return 0;
}
