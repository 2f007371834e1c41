// matchloss.hip — fused match loss from the rendered depth + its depth gradient (include/scg_matchloss.h).
// One 1024-thread workgroup per view pair (M <= ~2000 matches: 2 per thread); the only reduction is the masked
// mean, done in LDS; the gradient goes to the depth image with 4 atomics per match.
#include "scg_common.h"
#include "../../include/scg_matchloss.h"

namespace scg {

constexpr int kMlThreads = 1024;

__device__ __forceinline__ float fetch_or_zero(const float* __restrict__ d, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? d[(size_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(kMlThreads) void match_loss_pair_kernel(
    const float* __restrict__ depth, int H, int W, const float2* __restrict__ uv0, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ cam_rays_d, const float* __restrict__ mask0,
    const float* __restrict__ mask1, const float* __restrict__ intr1, const float* __restrict__ w2c1,
    const float2* __restrict__ uv1, int M, float width, float height, float* __restrict__ loss,
    float* __restrict__ grad_depth) {
    __shared__ float s_num[kMlThreads / kWave], s_den[kMlThreads / kWave];
    __shared__ float s_inv;
    float K[9], E[12];
#pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = intr1[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) E[i] = w2c1[i];       // first three rows of the 4x4

    float num = 0.f, den = 0.f;
    // pass 1: per-match loss, mask and d(loss_i)/d(sampled depth); kept in registers (<= 2 matches per thread
    // for M <= 2048, more are handled by the strided loop with a second evaluation in pass 2)
    for (int i = threadIdx.x; i < M; i += kMlThreads) {
        const float2 p = uv0[i];
        // grid_sample(align_corners=False): ix = ((2u/W - 1 + 1) * W - 1) / 2
        const float nx = (p.x / width) * 2.f - 1.f, ny = (p.y / height) * 2.f - 1.f;
        const float ix = ((nx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((ny + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float d = fetch_or_zero(depth, x0, y0, W, H) * (wx0 * wy0) + fetch_or_zero(depth, x0 + 1, y0, W, H) * (wx1 * wy0) +
                        fetch_or_zero(depth, x0, y0 + 1, W, H) * (wx0 * wy1) + fetch_or_zero(depth, x0 + 1, y0 + 1, W, H) * (wx1 * wy1);
        const float z = d / cam_rays_d[3 * i + 2];
        const float wxp = rays_o[3 * i] + rays_d[3 * i] * z, wyp = rays_o[3 * i + 1] + rays_d[3 * i + 1] * z,
                    wzp = rays_o[3 * i + 2] + rays_d[3 * i + 2] * z;
        const float cx = E[0] * wxp + E[1] * wyp + E[2] * wzp + E[3];
        const float cy = E[4] * wxp + E[5] * wyp + E[6] * wzp + E[7];
        const float cz = E[8] * wxp + E[9] * wyp + E[10] * wzp + E[11];
        const float X = K[0] * cx + K[1] * cy + K[2] * cz, Y = K[3] * cx + K[4] * cy + K[5] * cz,
                    Z = K[6] * cx + K[7] * cy + K[8] * cz;
        const float inv = 1.f / (Z + 1e-8f);
        const float px = X * inv, py = Y * inv;
        const bool in_img = (px > 0.f) && (px < width) && (py > 0.f) && (py < height);
        const bool valid = (!mask0 || !mask1) ? true : (mask0[i] * mask1[i] > 0.f);
        const float m = (in_img && valid) ? 1.f : 0.f;
        const float2 q = uv1[i];
        const float li = 0.5f * (fabsf(px - q.x) / width + fabsf(py - q.y) / height);
        num += li * m; den += m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { num += __shfl_down(num, off, kWave); den += __shfl_down(den, off, kWave); }
    if (lane_id() == 0) { s_num[wave_id()] = num; s_den[wave_id()] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tn = 0.f, td = 0.f;
        for (int k = 0; k < kMlThreads / kWave; ++k) { tn += s_num[k]; td += s_den[k]; }
        const float inv = 1.f / (td + 1e-8f);
        s_inv = inv;
        atomicAdd(loss, tn * inv);
    }
    if (!grad_depth) return;
    __syncthreads();
    const float inv_cnt = s_inv;
    // pass 2: gradient.  Everything is linear in the sampled depth d up to the perspective divide:
    //   (X,Y,Z) = a + b*z,  z = d / cam_rays_d.z
    for (int i = threadIdx.x; i < M; i += kMlThreads) {
        const float2 p = uv0[i];
        const float nx = (p.x / width) * 2.f - 1.f, ny = (p.y / height) * 2.f - 1.f;
        const float ix = ((nx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((ny + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const float d = fetch_or_zero(depth, x0, y0, W, H) * (wx0 * wy0) + fetch_or_zero(depth, x0 + 1, y0, W, H) * (wx1 * wy0) +
                        fetch_or_zero(depth, x0, y0 + 1, W, H) * (wx0 * wy1) + fetch_or_zero(depth, x0 + 1, y0 + 1, W, H) * (wx1 * wy1);
        const float rz = cam_rays_d[3 * i + 2];
        const float z = d / rz;
        const float rdx = rays_d[3 * i], rdy = rays_d[3 * i + 1], rdz = rays_d[3 * i + 2];
        const float wxp = rays_o[3 * i] + rdx * z, wyp = rays_o[3 * i + 1] + rdy * z, wzp = rays_o[3 * i + 2] + rdz * z;
        const float cx = E[0] * wxp + E[1] * wyp + E[2] * wzp + E[3];
        const float cy = E[4] * wxp + E[5] * wyp + E[6] * wzp + E[7];
        const float cz = E[8] * wxp + E[9] * wyp + E[10] * wzp + E[11];
        const float X = K[0] * cx + K[1] * cy + K[2] * cz, Y = K[3] * cx + K[4] * cy + K[5] * cz,
                    Z = K[6] * cx + K[7] * cy + K[8] * cz;
        // d(cam)/dz = R1 . rays_d ; d(X,Y,Z)/dz = K . that
        const float bx = E[0] * rdx + E[1] * rdy + E[2] * rdz, by = E[4] * rdx + E[5] * rdy + E[6] * rdz,
                    bz = E[8] * rdx + E[9] * rdy + E[10] * rdz;
        const float dX = K[0] * bx + K[1] * by + K[2] * bz, dY = K[3] * bx + K[4] * by + K[5] * bz,
                    dZ = K[6] * bx + K[7] * by + K[8] * bz;
        const float inv = 1.f / (Z + 1e-8f);
        const float px = X * inv, py = Y * inv;
        const bool in_img = (px > 0.f) && (px < width) && (py > 0.f) && (py < height);
        const bool valid = (!mask0 || !mask1) ? true : (mask0[i] * mask1[i] > 0.f);
        if (!(in_img && valid)) continue;
        const float dpx = (dX - px * dZ) * inv, dpy = (dY - py * dZ) * inv;
        const float2 q = uv1[i];
        const float sx = (px > q.x) ? 1.f : ((px < q.x) ? -1.f : 0.f), sy = (py > q.y) ? 1.f : ((py < q.y) ? -1.f : 0.f);
        const float dl_dz = 0.5f * (sx * dpx / width + sy * dpy / height);
        const float g = inv_cnt * dl_dz / rz;                 // d(term)/d(sampled depth)
        if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) unsafeAtomicAdd(&grad_depth[(size_t)y0 * W + x0], g * (wx0 * wy0));
        if (x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H) unsafeAtomicAdd(&grad_depth[(size_t)y0 * W + x0 + 1], g * (wx1 * wy0));
        if (x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H) unsafeAtomicAdd(&grad_depth[(size_t)(y0 + 1) * W + x0], g * (wx0 * wy1));
        if (x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H)
            unsafeAtomicAdd(&grad_depth[(size_t)(y0 + 1) * W + x0 + 1], g * (wx1 * wy1));
    }
}

}  // namespace scg

using namespace scg;

extern "C" int scg_match_loss_pair(const float* depth, int32_t H, int32_t W, const float* uv0, const float* rays_o,
                                   const float* rays_d, const float* cam_rays_d, const float* mask0,
                                   const float* mask1, const float* intr1, const float* w2c1, const float* uv1,
                                   int32_t M, float width, float height, float* loss, float* grad_depth, void* stream) {
    if (H <= 0 || W <= 0 || M < 0 || !(width > 0.f) || !(height > 0.f)) return fail(SCG_E_RANGE, "match loss dims out of range");
    if (M == 0) return 0;
    if (!depth || !uv0 || !rays_o || !rays_d || !cam_rays_d || !intr1 || !w2c1 || !uv1 || !loss)
        return fail(SCG_E_NULL, "match loss pointer is NULL");
    if ((mask0 == nullptr) != (mask1 == nullptr)) return fail(SCG_E_EXCLUSIVE, "pass both masks or neither");
    if ((reinterpret_cast<uintptr_t>(uv0) & 7u) || (reinterpret_cast<uintptr_t>(uv1) & 7u))
        return fail(SCG_E_ALIGN, "uv0 / uv1 must be 8-byte aligned");
    hipLaunchKernelGGL(match_loss_pair_kernel, dim3(1), dim3(kMlThreads), 0, reinterpret_cast<hipStream_t>(stream), depth,
                       H, W, reinterpret_cast<const float2*>(uv0), rays_o, rays_d, cam_rays_d, mask0, mask1, intr1, w2c1,
                       reinterpret_cast<const float2*>(uv1), M, width, height, loss, grad_depth);
    return check_hip(hipGetLastError(), "match_loss_pair_kernel");
}
