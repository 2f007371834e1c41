// Prints, for each cross-lane primitive used by the blend kernels, which source lane every lane reads.
// Build: hipcc --offload-arch=gfx950 -o dpp_probe dpp_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    int v = lane;
    out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x124, 0xF, 0xF, false);   // row_ror:4
    out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x128, 0xF, 0xF, false);   // row_ror:8
    out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x104, 0xF, 0xF, false);   // row_shl:4
    out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x124, 0xF, 0x5, false);   // row_ror:4 bank_mask 0x5
    {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
        out[5 * 64 + lane] = (int)r[0]; out[6 * 64 + lane] = (int)r[1];
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
        out[7 * 64 + lane] = (int)r[0]; out[8 * 64 + lane] = (int)r[1];
    }
}
int main() {
    int* d; hipMalloc(&d, 9 * 64 * 4);
    probe<<<1, 64>>>(d);
    int h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[9] = {"row_ror:4", "row_ror:8", "row_shl:4", "row_shr:4", "row_ror:4 bank_mask:0x5",
                            "permlane16_swap r[0] (a=lane,b=100+lane)", "permlane16_swap r[1]",
                            "permlane32_swap r[0]", "permlane32_swap r[1]"};
    for (int k = 0; k < 9; ++k) {
        printf("%s\n", names[k]);
        for (int l = 0; l < 64; ++l) printf("%4d%s", h[k * 64 + l], (l % 16 == 15) ? "\n" : "");
    }
    return 0;
}
