// matchloss.hip — fused match loss from the rendered depth + its depth gradient (include/scg_matchloss.h).
// One 1024-thread workgroup per view pair (M <= ~2000 matches: 2 per thread); the only reduction is the masked
// mean, done in LDS; the gradient goes to the depth image with 4 atomics per match.
#include "scg_common.h"
#include "../../include/scg_matchloss.h"

namespace scg {

constexpr int kMlThreads = 1024;
constexpr int kMlGroups = 8;                  // workgroups per pair: each scatters an eighth of the gradient (see the kernel)

__device__ __forceinline__ float fetch_or_zero(const float* __restrict__ d, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? d[(size_t)y * W + x] : 0.f;
}

// What one match contributes: its loss term, whether it counts, and d(term)/d(sampled depth) before the division by the
// number of counting matches (which only the whole workgroup knows).
struct MatchTerm {
    float li, m, dl_dd;        // loss term, 1 / 0, gradient per unit of sampled depth (x 1 / count later)
    int x0, y0;                // top-left tap of the bilinear sample
    float wx1, wy1;            // its weights
};
// A match's inputs as they lie in memory (every load independent of every other), then where its four depth taps are
struct MatchIn { float2 p, q; float ox, oy, oz, rdx, rdy, rdz, rz, mk; };
struct MatchTaps { int x0, y0; float wx1, wy1; size_t a00, a10, a01, a11; };

__device__ __forceinline__ MatchIn load_match(int i, const float2* __restrict__ uv0, const float* __restrict__ rays_o,
                                              const float* __restrict__ rays_d, const float* __restrict__ cam_rays_d,
                                              const float* __restrict__ m0, const float* __restrict__ m1, bool masked,
                                              const float2* __restrict__ uv1) {
    MatchIn r;
    r.p = uv0[i]; r.q = uv1[i];
    r.ox = rays_o[3 * i]; r.oy = rays_o[3 * i + 1]; r.oz = rays_o[3 * i + 2];
    r.rdx = rays_d[3 * i]; r.rdy = rays_d[3 * i + 1]; r.rdz = rays_d[3 * i + 2];
    r.rz = cam_rays_d[3 * i + 2];
    const float a = m0[i], b = m1[i];                 // (unmasked: any readable array — the product is not used)
    r.mk = masked ? a * b : 1.f;
    return r;
}

__device__ __forceinline__ MatchTaps taps_of(float2 p, int H, int W, float width, float height) {
    MatchTaps t;
    // grid_sample(align_corners=False): ix = ((2u/W - 1 + 1) * W - 1) / 2
    const float nx = (p.x / width) * 2.f - 1.f, ny = (p.y / height) * 2.f - 1.f;
    const float ix = ((nx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((ny + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.wx1 = ix - fx; t.wy1 = iy - fy;
    // unconditional loads at clamped coordinates, zeroed afterwards where a tap lies outside the image (a load behind each bounds
    // test is a branch per tap, and the compiler keeps them in order: four round trips instead of one)
    const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
    const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
    t.a00 = (size_t)ya * W + xa; t.a10 = (size_t)ya * W + xb; t.a01 = (size_t)yb * W + xa; t.a11 = (size_t)yb * W + xb;
    return t;
}

__device__ __forceinline__ MatchTerm match_term(const MatchIn& r, const MatchTaps& tp, float d00, float d10, float d01, float d11,
                                                int H, int W, const float (&K)[9], const float (&E)[12], float width,
                                                float height) {
    MatchTerm t;
    t.x0 = tp.x0; t.y0 = tp.y0; t.wx1 = tp.wx1; t.wy1 = tp.wy1;
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    const bool in_xa = t.x0 >= 0 && t.x0 < W, in_xb = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool in_ya = t.y0 >= 0 && t.y0 < H, in_yb = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    d00 = (in_xa && in_ya) ? d00 : 0.f; d10 = (in_xb && in_ya) ? d10 : 0.f;
    d01 = (in_xa && in_yb) ? d01 : 0.f; d11 = (in_xb && in_yb) ? d11 : 0.f;
    const float d = d00 * (wx0 * wy0) + d10 * (t.wx1 * wy0) + d01 * (wx0 * t.wy1) + d11 * (t.wx1 * t.wy1);
    const float z = d / r.rz;
    const float wxp = r.ox + r.rdx * z, wyp = r.oy + r.rdy * z, wzp = r.oz + r.rdz * z;
    const float cx = E[0] * wxp + E[1] * wyp + E[2] * wzp + E[3];
    const float cy = E[4] * wxp + E[5] * wyp + E[6] * wzp + E[7];
    const float cz = E[8] * wxp + E[9] * wyp + E[10] * wzp + E[11];
    const float X = K[0] * cx + K[1] * cy + K[2] * cz, Y = K[3] * cx + K[4] * cy + K[5] * cz,
                Z = K[6] * cx + K[7] * cy + K[8] * cz;
    const float inv = 1.f / (Z + 1e-8f);
    const float px = X * inv, py = Y * inv;
    const bool in_img = (px > 0.f) && (px < width) && (py > 0.f) && (py < height);
    t.m = (in_img && r.mk > 0.f) ? 1.f : 0.f;
    t.li = 0.5f * (fabsf(px - r.q.x) / width + fabsf(py - r.q.y) / height);
    // gradient.  Everything is linear in the sampled depth d up to the perspective divide: (X,Y,Z) = a + b*z, z = d / cam_rays_d.z;
    // d(cam)/dz = R1 . rays_d ; d(X,Y,Z)/dz = K . that
    const float bx = E[0] * r.rdx + E[1] * r.rdy + E[2] * r.rdz, by = E[4] * r.rdx + E[5] * r.rdy + E[6] * r.rdz,
                bz = E[8] * r.rdx + E[9] * r.rdy + E[10] * r.rdz;
    const float dX = K[0] * bx + K[1] * by + K[2] * bz, dY = K[3] * bx + K[4] * by + K[5] * bz,
                dZ = K[6] * bx + K[7] * by + K[8] * bz;
    const float dpx = (dX - px * dZ) * inv, dpy = (dY - py * dZ) * inv;
    const float sx = (px > r.q.x) ? 1.f : ((px < r.q.x) ? -1.f : 0.f), sy = (py > r.q.y) ? 1.f : ((py < r.q.y) ? -1.f : 0.f);
    t.dl_dd = 0.5f * (sx * dpx / width + sy * dpy / height);
    return t;
}

__device__ __forceinline__ void scatter_depth_gradient(const MatchTerm& t, float inv_cnt, float rz, int H, int W,
                                                       float* __restrict__ grad_depth) {
    if (t.m == 0.f) return;
    const float g = inv_cnt * t.dl_dd / rz;                 // d(term)/d(sampled depth)
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    const int x0 = t.x0, y0 = t.y0;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) unsafeAtomicAdd(&grad_depth[(size_t)y0 * W + x0], g * (wx0 * wy0));
    if (x0 + 1 >= 0 && x0 + 1 < W && y0 >= 0 && y0 < H) unsafeAtomicAdd(&grad_depth[(size_t)y0 * W + x0 + 1], g * (t.wx1 * wy0));
    if (x0 >= 0 && x0 < W && y0 + 1 >= 0 && y0 + 1 < H) unsafeAtomicAdd(&grad_depth[(size_t)(y0 + 1) * W + x0], g * (wx0 * t.wy1));
    if (x0 + 1 >= 0 && x0 + 1 < W && y0 + 1 >= 0 && y0 + 1 < H)
        unsafeAtomicAdd(&grad_depth[(size_t)(y0 + 1) * W + x0 + 1], g * (t.wx1 * t.wy1));
}

// Round 5: ONE pass.  The first version evaluated every match twice (loss, then gradient once the count of counting matches was
// known) and fetched each input where it was first used: ~10 dependent round trips to memory for a single workgroup, 32 us per
// pair in the trace of a training iteration.  Now every per-match input of a thread's (up to) two matches is requested at the
// top, the bilinear taps behind them, the terms and their gradients stay in registers across the reduction, and only the four
// atomics per match follow it.  Matches beyond 2 x 1024 (the reference samples <= 2 000 per pair) take the old two-pass route.
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
    const bool masked = mask0 && mask1;
    const float* m0 = masked ? mask0 : cam_rays_d;
    const float* m1 = masked ? mask1 : cam_rays_d;
    auto term_of = [&](int i) {
        const MatchIn r = load_match(i, uv0, rays_o, rays_d, cam_rays_d, m0, m1, masked, uv1);
        const MatchTaps tp = taps_of(r.p, H, W, width, height);
        return match_term(r, tp, depth[tp.a00], depth[tp.a10], depth[tp.a01], depth[tp.a11], H, W, K, E, width, height);
    };
    // the thread's first two matches, kept in registers: the inputs of BOTH in one round trip (clamped indices: no branch around a
    // load), then the eight depth taps in a second one
    const int i0 = threadIdx.x, i1 = threadIdx.x + kMlThreads;
    const int last = M - 1;
    const MatchIn r0 = load_match(min(i0, last), uv0, rays_o, rays_d, cam_rays_d, m0, m1, masked, uv1);
    const MatchIn r1 = load_match(min(i1, last), uv0, rays_o, rays_d, cam_rays_d, m0, m1, masked, uv1);
    const MatchTaps p0 = taps_of(r0.p, H, W, width, height), p1 = taps_of(r1.p, H, W, width, height);
    const float a0 = depth[p0.a00], b0 = depth[p0.a10], c0 = depth[p0.a01], e0 = depth[p0.a11];
    const float a1 = depth[p1.a00], b1 = depth[p1.a10], c1 = depth[p1.a01], e1 = depth[p1.a11];
    MatchTerm t0 = match_term(r0, p0, a0, b0, c0, e0, H, W, K, E, width, height);
    MatchTerm t1 = match_term(r1, p1, a1, b1, c1, e1, H, W, K, E, width, height);
    const float rz0 = r0.rz, rz1 = r1.rz;
    if (i0 >= M) t0.m = 0.f;
    if (i1 >= M) t1.m = 0.f;
    float num = t0.li * t0.m, den = t0.m;
    num += t1.li * t1.m; den += t1.m;
    for (int i = threadIdx.x + 2 * kMlThreads; i < M; i += kMlThreads) {
        const MatchTerm t = term_of(i);
        num += t.li * t.m; den += t.m;
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
        // (one hardware add, not waited for: atomicAdd(float*) is a compare-and-swap loop; every workgroup knows the sum, one adds it)
        if (blockIdx.x == 0) unsafeAtomicAdd(loss, tn * inv);
    }
    if (!grad_depth) return;
    __syncthreads();
    const float inv_cnt = s_inv;
    // The gradient's atomics are memory-side transactions (the image was zeroed by another kernel: device-scope adds go past the
    // L2s), ~2.5 ns each from ONE compute unit: 8 000 of them were 20 of the kernel's 23 us.  So the launch has kMlGroups
    // workgroups; each evaluates ALL matches (it needs their count; the loads are the cheap part) and scatters the gradient of
    // its share — waves w with w % kMlGroups == blockIdx.x.
    if ((wave_id() % kMlGroups) != (int)blockIdx.x % kMlGroups) return;
    scatter_depth_gradient(t0, inv_cnt, rz0, H, W, grad_depth);
    scatter_depth_gradient(t1, inv_cnt, rz1, H, W, grad_depth);
    for (int i = threadIdx.x + 2 * kMlThreads; i < M; i += kMlThreads)
        scatter_depth_gradient(term_of(i), inv_cnt, cam_rays_d[3 * i + 2], H, W, grad_depth);
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
    hipLaunchKernelGGL(match_loss_pair_kernel, dim3(grad_depth ? kMlGroups : 1), dim3(kMlThreads), 0, reinterpret_cast<hipStream_t>(stream), depth,
                       H, W, reinterpret_cast<const float2*>(uv0), rays_o, rays_d, cam_rays_d, mask0, mask1, intr1, w2c1,
                       reinterpret_cast<const float2*>(uv1), M, width, height, loss, grad_depth);
    return check_hip(hipGetLastError(), "match_loss_pair_kernel");
}
