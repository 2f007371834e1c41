// Sustained shader clock and VALU issue cost on the MI355X, measured three ways in one kernel:
//   s_memtime      (shader-clock ticks, MI355X_MICROARCH.md "s_memtime tick = shader cycle")
//   s_memrealtime  (constant 100 MHz wall clock)
//   hipEvent wall time around the launch
// -> effective clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, and cycles per wave64 VALU instruction per SIMD at
// 1, 2, 4 and 8 waves per SIMD.  This settles which denominator the VALU-issue fraction in bench.py must use.
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int kIters = 8192, kUnroll = 16;

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(uint64_t* stamps, float seed) {
    float a[kUnroll];
    uint32_t sc[4] = {1, 2, 3, 4};
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 l4 = {0, 0, 0, 0};
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed;
    const uint32_t ldsaddr = (uint32_t)(threadIdx.x & 7) * 16;
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + threadIdx.x + i;
    const float m = 1.0000001f, c = 1e-9f;
    const uint64_t t0 = __builtin_readcyclecounter();          // s_memtime
    const uint64_t r0 = wall_clock64();                        // s_memrealtime, 100 MHz
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(0xAAAAAAAAAAAAAAAAull));
            if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[i & 3]));
            if (MODE == 6) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[i & 3])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); }
            if (MODE == 7) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sc[i & 3]) : "v"(a[i]));
            if (MODE == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"(ldsaddr));
            if (MODE == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < kUnroll; ++i) s += a[i];
    s += (float)(sc[0] + sc[1] + sc[2] + sc[3]) + l4[0] + l4[3];
    if (s == 12345.f) stamps[0] = (uint64_t)s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;       // 4-wave workgroups: `waves_per_simd` workgroups per CU
    const size_t n_waves = (size_t)blocks * 4;
    uint64_t* d;
    hipMalloc(&d, n_waves * 16);
    probe_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(n_waves * 2);
    hipMemcpy(h.data(), d, n_waves * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc(n_waves), real(n_waves);
    for (size_t i = 0; i < n_waves; ++i) { cyc[i] = (double)h[2 * i]; real[i] = (double)h[2 * i + 1]; }
    std::sort(cyc.begin(), cyc.end()); std::sort(real.begin(), real.end());
    const double mc = cyc[n_waves / 2], mr = real[n_waves / 2];
    const double instr_per_simd = (double)kIters * kUnroll * waves_per_simd;
    const double ghz = mc / (mr * 10.0);           // ticks per ns: realtime tick = 10 ns
    printf("%-14s waves/SIMD %d: wall %.3f ms | median wave: %.0f shader ticks in %.1f us -> clock %.3f GHz | "
           "%.2f shader cycles and %.3f ns per wave64 instruction per SIMD\n",
           name, waves_per_simd, ms, mc, mr / 100.0, ghz, mc * 1.0 / (kIters * kUnroll) / 1.0 / waves_per_simd * 1.0,
           mr * 10.0 / instr_per_simd);
    hipFree(d);
}

int main(int argc, char** argv) {
    if (argc > 1) {      // "pmc": a short fixed sequence for rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
        run<0>("v_fma_f32", 4); run<0>("v_fma_f32", 8); run<1>("v_exp_f32", 8); run<2>("v_add_f32_dpp", 8);
        return 0;
    }
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w);
    for (int w : {1, 8}) run<1>("v_exp_f32", w);
    for (int w : {1, 8}) run<4>("v_rcp_f32", w);
    for (int w : {1, 8}) run<2>("v_add_f32_dpp", w);
    for (int w : {1, 8}) run<3>("v_cndmask_b32", w);
    for (int w : {1, 4, 8}) run<5>("s_add_u32", w);
    for (int w : {4, 8}) run<6>("s_add+v_fma pair", w);
    for (int w : {4, 8}) run<7>("v_readlane_b32", w);
    for (int w : {4, 8}) run<8>("ds_read_b128 bcast", w);
    for (int w : {4, 8}) run<9>("v_cmp+v_cndmask vcc", w);
    return 0;
}
