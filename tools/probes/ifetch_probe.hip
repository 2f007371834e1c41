// Instruction-supply probe for the MI355X: the same ALU work encoded in more or fewer bytes, 8 waves per SIMD on every CU.
// If two bodies with identical ALU cost differ in time by their size in bytes, the instruction fetch path bounds them.
//   M0  v_fmac_f32_e32                  4 B   1 vector op
//   M1  v_fma_f32 (VOP3)                8 B   1 vector op
//   M2  s_mov_b32 s, s                  4 B   1 scalar op
//   M3  s_mov_b32 s, literal            8 B   1 scalar op
//   M4  v_fmac_e32 + s_mov s,s          8 B   1 vector + 1 scalar
//   M5  v_fmac_e32 + s_mov literal     12 B   1 vector + 1 scalar
//   M6  v_fma VOP3 + s_mov literal     16 B   1 vector + 1 scalar
//   M7  v_fma VOP3 + 2 x s_mov literal 24 B   1 vector + 2 scalar
//   M8  v_fmac_e32 + 2 x s_mov s,s     12 B   1 vector + 2 scalar
// Build: hipcc --offload-arch=gfx950 -O3 -o ifetch_probe ifetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 4096, kUnroll = 32;

template <int MODE>
__global__ __launch_bounds__(256) void probe_kernel(float* out, float seed) {
    float a[8];
    uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    const float m = 1.0000001f, c = 1e-9f;
    // M9 / M10: the v_fmac loop of M0 with only the lower 32 lanes / only 16 lanes enabled — does a half-empty wave64
    // instruction cost half?
    if (MODE == 9) asm volatile("s_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0" ::: "memory");
    if (MODE == 10) asm volatile("s_mov_b32 exec_lo, 0xffff\n\ts_mov_b32 exec_hi, 0" ::: "memory");
    if (MODE == 11) asm volatile("s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, -1" ::: "memory");
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            float& x = a[i & 7];
            if (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 8 || MODE >= 9) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
            if (MODE == 1 || MODE == 6 || MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
            if (MODE == 2 || MODE == 4 || MODE == 8) asm volatile("s_mov_b32 %0, %1" : "=s"(s0) : "s"(s1));
            if (MODE == 8) asm volatile("s_mov_b32 %0, %1" : "=s"(s2) : "s"(s3));
            if (MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7) asm volatile("s_mov_b32 %0, 0x12345678" : "=s"(s0));
            if (MODE == 7) asm volatile("s_mov_b32 %0, 0x23456789" : "=s"(s2));
        }
    }
    if (MODE >= 9) asm volatile("s_mov_b64 exec, -1" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    s += (float)(s0 + s1 + s2 + s3);
    if (s == 12345.f) out[0] = s;
}

template <int MODE>
void run(const char* name, int bytes, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;
    float* d;
    hipMalloc(&d, 4);
    probe_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe_kernel<MODE><<<blocks, 256>>>(d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_simd = (double)kIters * kUnroll * waves_per_simd;
    const double ns = ms * 1e6 / groups_per_simd;
    printf("%-34s %2d B/group  waves/SIMD %d: %.3f ms  %.3f ns per group per SIMD  -> %.2f B/ns per SIMD, %.1f B/ns per CU\n",
           name, bytes, waves_per_simd, ms, ns, bytes / ns, 4 * bytes / ns);
    hipFree(d);
}

int main() {
    for (int w : {2, 8}) {
        run<0>("v_fmac_e32", 4, w);
        run<1>("v_fma VOP3", 8, w);
        run<2>("s_mov s,s", 4, w);
        run<3>("s_mov literal", 8, w);
        run<4>("v_fmac_e32 + s_mov s,s", 8, w);
        run<5>("v_fmac_e32 + s_mov literal", 12, w);
        run<6>("v_fma VOP3 + s_mov literal", 16, w);
        run<7>("v_fma VOP3 + 2 s_mov literal", 24, w);
        run<8>("v_fmac_e32 + 2 s_mov s,s", 12, w);
        run<9>("v_fmac_e32, lanes 0-31 only", 4, w);
        run<11>("v_fmac_e32, lanes 32-63 only", 4, w);
        run<10>("v_fmac_e32, lanes 0-15 only", 4, w);
    }
    return 0;
}
