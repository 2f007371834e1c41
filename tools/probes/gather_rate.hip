// gather_rate.hip — how fast does ONE compute unit (and the chip) process DIVERGENT global loads whose lines are on-die?
// Every lane of a wave reads a different 48-byte record of a table (the blend kernels' splat gather, the sorts' depth-key gather):
//   x4    one dwordx4 per lane per record            x4x3  the record's three dwordx4 pieces (three instructions)
//   x1    one dword per lane                          seq   the same number of bytes fully coalesced (reference)
// Prints lane-accesses per shader cycle per compute unit at 1, 4, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void k(const float4* __restrict__ table, const uint32_t* __restrict__ idx, int n_idx, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    uint32_t base = (uint32_t)(wave * 977u) % (uint32_t)(n_idx - 64 * iters - 64);
    for (int it = 0; it < iters; it += 4) {
        uint32_t id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = idx[base + (it + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { const float4 v = table[3 * (size_t)id[u]]; acc += v.x + v.w; }
            if (MODE == 1) {
                const float4 a = table[3 * (size_t)id[u]], b = table[3 * (size_t)id[u] + 1], c = table[3 * (size_t)id[u] + 2];
                acc += a.x + b.y + c.z;
            }
            if (MODE == 2) { acc += reinterpret_cast<const float*>(table)[12 * (size_t)id[u]]; }
            if (MODE == 4) {       // the same three pieces of a record that starts on a 64-byte line (64-byte stride)
                const float4 a = table[4 * (size_t)id[u]], b = table[4 * (size_t)id[u] + 1], c = table[4 * (size_t)id[u] + 2];
                acc += a.x + b.y + c.z;
            }
            if (MODE == 5) {       // two pieces of a 48-byte record (what a cull test needs)
                const float4 a = table[3 * (size_t)id[u]], b = table[3 * (size_t)id[u] + 1];
                acc += a.x + b.y;
            }
            if (MODE == 3) { const float4 v = table[(size_t)(base + (it + u) * 64 + lane) % (size_t)(3 * 2000)]; acc += v.x + v.w; }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 200000, n_idx = 1 << 22;        // records of the table: 200 000 = the S2 splat table
    std::vector<uint32_t> h(n_idx);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n_idx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % P); }
    float4* table; uint32_t* idx; float* out;
    hipMalloc(&table, (size_t)P * 64); hipMemset(table, 0, (size_t)P * 64);
    hipMalloc(&idx, (size_t)n_idx * 4); hipMemcpy(idx, h.data(), (size_t)n_idx * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const int iters = 256;
    const char* names[6] = {"x4 (one dwordx4 per lane, random record)", "x4x3 (three dwordx4 of one random record)",
                            "x1 (one dword per lane, random record)", "seq (dwordx4, coalesced)",
                            "x4x3 of a 64-byte-aligned record", "x4x2 (two dwordx4 of a 48-byte record)"};
    for (int wps : {1, 4, 8}) {
        for (int mode = 0; mode < 6; ++mode) {
            const int waves = cus * 4 * wps;
            dim3 grid(waves), block(64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, table, idx, n_idx, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, table, idx, n_idx, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, table, idx, n_idx, iters, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, table, idx, n_idx, iters, out);
                if (mode == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, table, idx, n_idx, iters, out);
                if (mode == 5) hipLaunchKernelGGL(k<5>, grid, block, 0, 0, table, idx, n_idx, iters, out);
            };
            for (int w = 0; w < 3; ++w) launch();
            hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            const double lane_acc = (double)waves * iters * 64 * ((mode == 1 || mode == 4) ? 3 : mode == 5 ? 2 : 1);
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf("%d waves/SIMD  %-46s %8.1f us  %6.2f lane-accesses / cycle / CU = %5.2f cycles per RECORD  (%.0f GB/s of payload)\n", wps, names[mode],
                   ms * 1e3, lane_acc / cyc / cus, cyc * cus / ((double)waves * iters * 64), lane_acc * (mode == 2 ? 4 : 16) / (ms * 1e-3) / 1e9);
        }
    }
    printf("(clock %.2f GHz nominal, %d CUs; table %d records x 48 B = %.1f MB)\n", ghz, cus, P, P * 48 / 1e6);
    return 0;
}
