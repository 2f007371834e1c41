// loss.hip — fused L1 + SSIM image loss and its gradient (include/scg_loss.h).
//
// One 256-thread workgroup per (channel, 32x32 output tile).  The 42x42 input halo tile (zero outside the image =
// conv2d zero padding) goes through LDS; the 11x11 Gaussian window is applied separably: a horizontal pass over
// the 42 halo rows into LDS, then a vertical pass.  Both passes are REGISTER-BLOCKED (round 5): a thread produces four
// adjacent outputs of a row (a column) from the 14 values they share instead of 4 x 11 — the first version (16x16 tiles,
// one output per thread and pass) read LDS ~100 times per pixel and channel and took 38 + 32 us at 1008x756 for bytes
// that cost 7 + 9 us; every output still adds its eleven taps in the same order.  The 1-D window is computed on the host
// exactly as utils/loss_utils.py:46-48 does (fp32 normalisation).  HBM traffic: forward reads 2 images and writes 3 maps,
// backward reads 3 maps + 2 images and writes the gradient: ~13 floats per pixel-channel for both.
#include <math.h>

#include "scg_common.h"
#include "../../include/scg_loss.h"

namespace scg {

constexpr int kWin = 11;
constexpr int kHalo = kWin / 2;                 // 5
// (the tile's height is a build parameter for the A/B that chose it: 32 / 16 / 8 rows = 3 / 4-5 / 8 workgroups per compute
//  unit and 42.6 / 47.2 / 61.3 us forward, 30.0 / 31.8 / 35.8 backward — the halo rows cost more than the occupancy buys,
//  profiles/r05_loss_tile_height.txt)
#ifndef SCG_LOSS_TILE_Y
#define SCG_LOSS_TILE_Y 32
#endif
constexpr int kLT = 32;                         // output tile: columns
constexpr int kLTY = SCG_LOSS_TILE_Y;           // ... and rows
constexpr int kLH = kLT + 2 * kHalo;            // 42 halo columns
constexpr int kLHY = kLTY + 2 * kHalo;          // halo rows
constexpr int kQ = 4;                           // adjacent outputs per thread in the horizontal pass
constexpr int kSpan = kQ + kWin - 1;            // 14 inputs feed them
constexpr int kQV = kLTY * kLT / kBlock;        // ... and in the vertical pass: every thread one column, kQV rows
constexpr int kSpanV = kQV + kWin - 1;
static_assert(kBlock == kLT * (kLTY / kQV) && kQV >= 1, "the vertical pass gives every thread one column and kQV rows");
constexpr float kC1 = 0.01f * 0.01f;
constexpr float kC2 = 0.03f * 0.03f;

struct Window { float g[kWin]; };

static Window make_window() {
    // gaussian(11, 1.5): exp(-(x-5)^2 / (2*1.5^2)) in double, stored fp32, normalised in fp32
    Window w;
    float s = 0.f;
    for (int x = 0; x < kWin; ++x) {
        w.g[x] = (float)exp(-(double)((x - kWin / 2) * (x - kWin / 2)) / (2.0 * 1.5 * 1.5));
        s += w.g[x];
    }
    for (int x = 0; x < kWin; ++x) w.g[x] = w.g[x] / s;
    return w;
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if (lane_id() == 0) s_red[wave_id()] = v;
    __syncthreads();
    const float t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kBlock) void image_loss_forward_kernel(const float* __restrict__ img,
                                                                    const float* __restrict__ gt, int H, int W,
                                                                    Window win, float2* __restrict__ partials,
                                                                    float* __restrict__ dmaps) {
    __shared__ float s_x[kLHY][kLH + 1];
    __shared__ float s_y[kLHY][kLH + 1];
    __shared__ float s_h[5][kLHY][kLT + 1];       // horizontal pass: x, y, xx, yy, xy
    __shared__ float s_red[4];
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLTY;
    const size_t plane = (size_t)c * H * W;
    // the halo tile: every load of the thread issued before the first is waited for (clamped coordinates, zeroed afterwards
    // where the position lies outside the image: a load behind its bounds test is a branch, and seven of them in a row are
    // seven round trips)
    constexpr int kLoads = (kLHY * kLH + kBlock - 1) / kBlock;
    float hx[kLoads], hy[kLoads];
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
        const int k = min((int)threadIdx.x + it * kBlock, kLHY * kLH - 1);
        const int ly = k / kLH, lx = k - ly * kLH;
        const int gy = min(max(y0 + ly - kHalo, 0), H - 1), gx = min(max(x0 + lx - kHalo, 0), W - 1);
        hx[it] = img[plane + (size_t)gy * W + gx];
        hy[it] = gt[plane + (size_t)gy * W + gx];
    }
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
        const int k = (int)threadIdx.x + it * kBlock;
        const int ly = k / kLH, lx = k - ly * kLH;
        const int gy = y0 + ly - kHalo, gx = x0 + lx - kHalo;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        if (k < kLHY * kLH) { s_x[ly][lx] = in ? hx[it] : 0.f; s_y[ly][lx] = in ? hy[it] : 0.f; }
    }
    __syncthreads();
    // horizontal pass: item = (halo row, group of four adjacent columns)
    for (int k = threadIdx.x; k < kLHY * (kLT / kQ); k += kBlock) {
        const int ly = k / (kLT / kQ), lx = (k - ly * (kLT / kQ)) * kQ;
        // (the products are formed once per input value, then convolved — the reference's order: conv2d(img1 * img1, window),
        //  utils/loss_utils.py:80-82)
        float vx[kSpan], vy[kSpan], xx[kSpan], yy[kSpan], xy[kSpan];
#pragma unroll
        for (int u = 0; u < kSpan; ++u) {
            vx[u] = s_x[ly][lx + u]; vy[u] = s_y[ly][lx + u];
            xx[u] = vx[u] * vx[u]; yy[u] = vy[u] * vy[u]; xy[u] = vx[u] * vy[u];
        }
#pragma unroll
        for (int j = 0; j < kQ; ++j) {
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const float g = win.g[t];
                a += g * vx[j + t]; b += g * vy[j + t]; aa += g * xx[j + t]; bb += g * yy[j + t]; ab += g * xy[j + t];
            }
            s_h[0][ly][lx + j] = a; s_h[1][ly][lx + j] = b; s_h[2][ly][lx + j] = aa; s_h[3][ly][lx + j] = bb; s_h[4][ly][lx + j] = ab;
        }
    }
    __syncthreads();
    // vertical pass: thread = (column, group of four adjacent rows)
    const int lx = threadIdx.x & (kLT - 1), ly0 = (threadIdx.x / kLT) * kQV;
    const int gx = x0 + lx;
    float l1 = 0.f, ss = 0.f;
    float col[5][kSpanV];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int u = 0; u < kSpanV; ++u) col[q][u] = s_h[q][ly0 + u][lx];
#pragma unroll
    for (int j = 0; j < kQV; ++j) {
        const int ly = ly0 + j, gy = y0 + ly;
        if (gx < W && gy < H) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const float g = win.g[t];
                m1 += g * col[0][j + t]; m2 += g * col[1][j + t];
                e11 += g * col[2][j + t]; e22 += g * col[3][j + t]; e12 += g * col[4][j + t];
            }
            const float m1m2 = m1 * m2, m1s = m1 * m1, m2s = m2 * m2;
            const float s1 = e11 - m1s, s2 = e22 - m2s, s12 = e12 - m1m2;
            const float A1 = 2.f * m1m2 + kC1, A2 = 2.f * s12 + kC2;
            const float B1 = m1s + m2s + kC1, B2 = s1 + s2 + kC2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            ss += S;
            const float vx = s_x[ly + kHalo][lx + kHalo], vy = s_y[ly + kHalo][lx + kHalo];
            l1 += fabsf(vx - vy);
            if (dmaps) {
                // S as a function of the three convolutions that depend on img: m1 = w*x, e11 = w*x^2, e12 = w*xy
                const float dS_ds1 = -S / B2;
                const float dS_ds12 = 2.f * A1 * inv;
                const float dS_dm1 = 2.f * m2 * A2 * inv - 2.f * m1 * S / B1 + dS_ds1 * (-2.f * m1) + dS_ds12 * (-m2);
                const size_t n = (size_t)gridDim.z * H * W;
                const size_t p = plane + (size_t)gy * W + gx;
                dmaps[p] = dS_dm1; dmaps[n + p] = dS_ds1; dmaps[2 * n + p] = dS_ds12;
            }
        }
    }
    const float t_l1 = block_sum(l1, s_red);
    const float t_ss = block_sum(ss, s_red);
    if (threadIdx.x == 0)
        partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float2(t_l1, t_ss);
}

// fixed-order reduction of the per-workgroup partial sums (single 1024-thread workgroup)
// lambda_dssim >= 0: also sums[2] = (1 - lambda) * L1 + lambda * (1 - SSIM), the training loss (train.py:160-161) — the
// operations of the reference's expression in its order, each rounded on its own (no contraction)
__global__ __launch_bounds__(1024) void image_loss_reduce_kernel(const float2* __restrict__ partials, int n,
                                                                 float* __restrict__ sums, float inv_count,
                                                                 float lambda_dssim) {
    __shared__ float s_a[16], s_b[16];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) { const float2 p = partials[i]; a += p.x; b += p.y; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, kWave); b += __shfl_down(b, off, kWave); }
    if (lane_id() == 0) { s_a[wave_id()] = a; s_b[wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int k = 0; k < 16; ++k) { ta += s_a[k]; tb += s_b[k]; }
        sums[0] = ta; sums[1] = tb;
        if (lambda_dssim >= 0.f) {
#pragma clang fp contract(off)
            const float l1 = ta * inv_count, ssim = tb * inv_count;
            const float a = (1.0f - lambda_dssim) * l1, b = lambda_dssim * (1.0f - ssim);
            sums[2] = a + b;
        }
    }
}

__global__ __launch_bounds__(kBlock) void image_loss_backward_kernel(const float* __restrict__ img,
                                                                     const float* __restrict__ gt,
                                                                     const float* __restrict__ dmaps, int H, int W,
                                                                     Window win, const float* __restrict__ weights,
                                                                     float* __restrict__ d_img, float scale_l1,
                                                                     float scale_ssim, int scaled) {
    // scaled: weights[0] is the upstream gradient of the COMBINED loss (a device scalar), the two factors are the caller's
    const float w_l1 = scaled ? weights[0] * scale_l1 : weights[0], w_ssim = scaled ? weights[0] * scale_ssim : weights[1];
    __shared__ float s_m[3][kLHY][kLH + 1];
    __shared__ float s_h[3][kLHY][kLT + 1];
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLTY;
    const size_t plane = (size_t)c * H * W;
    const size_t n = (size_t)gridDim.z * H * W;
    constexpr int kLoads = (kLHY * kLH + kBlock - 1) / kBlock;          // (as in the forward: all loads first)
    float ha[kLoads], hb[kLoads], hd[kLoads];
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
        const int k = min((int)threadIdx.x + it * kBlock, kLHY * kLH - 1);
        const int ly = k / kLH, lx = k - ly * kLH;
        const int gy = min(max(y0 + ly - kHalo, 0), H - 1), gx = min(max(x0 + lx - kHalo, 0), W - 1);
        const size_t p = plane + (size_t)gy * W + gx;
        ha[it] = dmaps[p]; hb[it] = dmaps[n + p]; hd[it] = dmaps[2 * n + p];
    }
#pragma unroll
    for (int it = 0; it < kLoads; ++it) {
        const int k = (int)threadIdx.x + it * kBlock;
        const int ly = k / kLH, lx = k - ly * kLH;
        const int gy = y0 + ly - kHalo, gx = x0 + lx - kHalo;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        if (k < kLHY * kLH) { s_m[0][ly][lx] = in ? ha[it] : 0.f; s_m[1][ly][lx] = in ? hb[it] : 0.f; s_m[2][ly][lx] = in ? hd[it] : 0.f; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kLHY * (kLT / kQ); k += kBlock) {
        const int ly = k / (kLT / kQ), lx = (k - ly * (kLT / kQ)) * kQ;
        float v[3][kSpan];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int u = 0; u < kSpan; ++u) v[q][u] = s_m[q][ly][lx + u];
#pragma unroll
        for (int j = 0; j < kQ; ++j) {
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const float g = win.g[t];
                a += g * v[0][j + t]; b += g * v[1][j + t]; d += g * v[2][j + t];
            }
            s_h[0][ly][lx + j] = a; s_h[1][ly][lx + j] = b; s_h[2][ly][lx + j] = d;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & (kLT - 1), ly0 = (threadIdx.x / kLT) * kQV;
    const int gx = x0 + lx;
    float col[3][kSpanV];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int u = 0; u < kSpanV; ++u) col[q][u] = s_h[q][ly0 + u][lx];
#pragma unroll
    for (int j = 0; j < kQV; ++j) {
        const int gy = y0 + ly0 + j;
        if (gx < W && gy < H) {
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int t = 0; t < kWin; ++t) {
                const float g = win.g[t];
                a += g * col[0][j + t]; b += g * col[1][j + t]; d += g * col[2][j + t];
            }
            const size_t p = plane + (size_t)gy * W + gx;
            const float vx = img[p], vy = gt[p];
            const float diff = vx - vy;
            const float sgn = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
            d_img[p] = w_l1 * sgn + w_ssim * (a + 2.f * vx * b + vy * d);
        }
    }
}

}  // namespace scg

using namespace scg;

extern "C" {

size_t scg_image_loss_dmaps_bytes(int32_t C, int32_t H, int32_t W) {
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)3 * C * H * W * sizeof(float);
}

static int check_dims(int32_t C, int32_t H, int32_t W) {
    if (C <= 0 || C > 65535 || H <= 0 || W <= 0 || (int64_t)H * W > (1ll << 31)) return fail(SCG_E_RANGE, "image dims out of range");
    return 0;
}

size_t scg_image_loss_scratch_bytes(int32_t C, int32_t H, int32_t W) {
    if (C <= 0 || H <= 0 || W <= 0) return 256;
    return (size_t)C * ((H + kLTY - 1) / kLTY) * ((W + kLT - 1) / kLT) * sizeof(float2) + 256;
}

static int image_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* sums,
                              float* dmaps, void* scratch, size_t scratch_bytes, float lambda_dssim, void* stream) {
    int rc = check_dims(C, H, W);
    if (rc) return rc;
    if (!img || !gt || !sums || !scratch) return fail(SCG_E_NULL, "image_loss_forward pointer is NULL");
    if (scratch_bytes < scg_image_loss_scratch_bytes(C, H, W)) return fail(SCG_E_SCRATCH, "image loss scratch too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    static const Window win = make_window();
    const dim3 grid((W + kLT - 1) / kLT, (H + kLTY - 1) / kLTY, C);
    float2* partials = reinterpret_cast<float2*>(scratch);
    hipLaunchKernelGGL(image_loss_forward_kernel, grid, dim3(kBlock), 0, s, img, gt, H, W, win, partials, dmaps);
    hipLaunchKernelGGL(image_loss_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, (int)(grid.x * grid.y * grid.z),
                       sums, 1.0f / ((float)C * (float)H * (float)W), lambda_dssim);
    return check_hip(hipGetLastError(), "image_loss_forward_kernel");
}

int scg_image_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* sums,
                           float* dmaps, void* scratch, size_t scratch_bytes, void* stream) {
    return image_loss_forward(img, gt, C, H, W, sums, dmaps, scratch, scratch_bytes, -1.0f, stream);
}

int scg_image_loss_forward_combined(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                                    float* sums3, float* dmaps, void* scratch, size_t scratch_bytes, void* stream) {
    if (!(lambda_dssim >= 0.f && lambda_dssim <= 1.f)) return fail(SCG_E_RANGE, "lambda_dssim must lie in [0, 1]");
    return image_loss_forward(img, gt, C, H, W, sums3, dmaps, scratch, scratch_bytes, lambda_dssim, stream);
}

int scg_image_loss_backward(const float* img, const float* gt, const float* dmaps, int32_t C, int32_t H, int32_t W,
                            const float* weights, float* d_img, void* stream) {
    int rc = check_dims(C, H, W);
    if (rc) return rc;
    if (!img || !gt || !dmaps || !d_img || !weights) return fail(SCG_E_NULL, "image_loss_backward pointer is NULL");
    static const Window win = make_window();
    const dim3 grid((W + kLT - 1) / kLT, (H + kLTY - 1) / kLTY, C);
    hipLaunchKernelGGL(image_loss_backward_kernel, grid, dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), img,
                       gt, dmaps, H, W, win, weights, d_img, 0.f, 0.f, 0);
    return check_hip(hipGetLastError(), "image_loss_backward_kernel");
}

int scg_image_loss_backward_combined(const float* img, const float* gt, const float* dmaps, int32_t C, int32_t H, int32_t W,
                                     const float* upstream, float lambda_dssim, float* d_img, void* stream) {
    int rc = check_dims(C, H, W);
    if (rc) return rc;
    if (!img || !gt || !dmaps || !d_img || !upstream) return fail(SCG_E_NULL, "image_loss_backward pointer is NULL");
    if (!(lambda_dssim >= 0.f && lambda_dssim <= 1.f)) return fail(SCG_E_RANGE, "lambda_dssim must lie in [0, 1]");
    static const Window win = make_window();
    const dim3 grid((W + kLT - 1) / kLT, (H + kLTY - 1) / kLTY, C);
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    hipLaunchKernelGGL(image_loss_backward_kernel, grid, dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), img,
                       gt, dmaps, H, W, win, upstream, d_img, (1.0f - lambda_dssim) * inv_n, -lambda_dssim * inv_n, 1);
    return check_hip(hipGetLastError(), "image_loss_backward_kernel");
}

}  // extern "C"
