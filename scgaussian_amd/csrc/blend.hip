// blend.hip — 16x16-tile alpha blending of colour + depth + alpha, forward and per-pixel backward
// (outputs consumed at reference gaussian_renderer/__init__.py:100, depth at
// scene/gaussian_model.py:246-263, alpha at train.py:168).
//
// MI355X mapping
//   * one 256-thread workgroup (4 x wave64) per tile; wave q owns the 8x8-pixel QUADRANT q of the tile,
//     one pixel per lane, so everything a wave decides (skip a splat, leave the loop) is wave-uniform.
//   * the tile's sorted list is walked in chunks of 256: thread t gathers the 48-byte splat record of list
//     entry t (three 16-byte loads from one record; the table is L2 / Infinity-Cache resident) into LDS and,
//     while it holds the record in registers, tests it against the four quadrants: it evaluates the exact
//     minimum of the conic's quadratic form over each 8x8 pixel rectangle and compares it with the
//     alpha >= 1/255 cut-off 2*ln(255*opacity) (with a safety margin).  A ballot turns the results into one
//     64-bit mask per (quadrant, loader wave).
//   * a consumer wave then iterates ONLY over the set bits of its masks with scalar bit scans; records are
//     read with wave-uniform (broadcast) LDS reads.  With 32 waves per CU sharing ONE LDS pipe the forward is
//     LDS-issue bound (SQ_LDS_IDX_ACTIVE ~ 65 % of the kernel), so the record is laid out for the cheapest reads:
//     {x,y,conic_a,conic_b} = one ds_read_b128 (4 LDS cycles), {conic_c,opacity} = one ds_read_b64 (2), and
//     {r,g,b,depth} = one ds_read_b128 issued by contributing lanes only — never a ds_read_b96 (8 cycles).  The test is conservative, so the skipped splats are
//     exactly ones every pixel of the quadrant would have skipped itself: results are unchanged, only the
//     ~4x redundant work of the loose 3-sigma tile rectangle disappears.
//   * blockIdx -> tile is XCD-aware (xcd_tile_remap): an XCD's private L2 sees a contiguous band of tiles.
//   * backward: per-splat partial gradients are summed across the 64 lanes with DPP adds, accumulated across
//     the tile's 4 waves with LDS float atomics, and flushed with ONE set of global atomics per splat per
//     tile (hardware global_atomic_add_f32).
#include "scg_common.h"

namespace scg {

constexpr int kChunk = kBlock;        // list entries staged per round

// Conservative "does this splat reach any pixel of the rectangle [x0,x1]x[y0,y1]" test.
// g = splat centre, (ca,cb,cc) = conic, thr = cut-off of the quadratic form q(d) = ca dx^2 + 2 cb dx dy + cc dy^2.
// A pixel blends the splat only if q <= 2 ln(255 opacity) (alpha >= 1/255) — and q >= 0 (power <= 0).
// q is convex, so when the centre lies outside the rectangle its minimum over the rectangle is attained on
// an edge facing the centre; on an edge it is a clamped 1-D parabola minimum: exact, no sampling.
__device__ __forceinline__ bool rect_hit(float gx, float gy, float ca, float cb, float cc, float inv_ca, float inv_cc,
                                         float thr, float x0, float y0, float x1, float y1) {
    const float dx0 = x0 - gx, dx1 = x1 - gx, dy0 = y0 - gy, dy1 = y1 - gy;
    const bool inx = (dx0 <= 0.f) && (dx1 >= 0.f);
    const bool iny = (dy0 <= 0.f) && (dy1 >= 0.f);
    if (inx && iny) return true;
    float qmin = 3.0e38f;
    if (!inx) {
        const float dx = (dx0 > 0.f) ? dx0 : dx1;
        const float dy = fminf(fmaxf(-cb * dx * inv_cc, dy0), dy1);
        qmin = ca * dx * dx + 2.f * cb * dx * dy + cc * dy * dy;
    }
    if (!iny) {
        const float dy = (dy0 > 0.f) ? dy0 : dy1;
        const float dx = fminf(fmaxf(-cb * dy * inv_ca, dx0), dx1);
        qmin = fminf(qmin, ca * dx * dx + 2.f * cb * dx * dy + cc * dy * dy);
    }
    return !(qmin > thr);       // NaN -> hit (never cull on a malformed conic)
}

// 4-bit quadrant mask of one splat against the tile at pixel origin (tx0, ty0).
__device__ __forceinline__ uint32_t quadrant_hits(const float4& a, const float4& b, float tx0, float ty0) {
    // alpha = min(0.99, o*exp(power)) >= 1/255  <=>  q <= 2 ln(255 o).  Margin: 0.1 % + 0.01 absolute on q
    // (fp32 evaluation error of q is < 1e-4 here), so no pixel that would pass its own test is culled.
    // record layout: a = {x, y, conic_a, conic_b}, b = {conic_c, opacity, -, -}, c = {r, g, b, depth}
    const float L = __logf(255.0f * b.y);
    if (!(L >= -0.01f)) return (b.y != b.y) ? 0xFu : 0u;   // opacity < 1/255 never blends; NaN -> keep
    const float thr = 2.0f * L * 1.001f + 0.01f;
    // v_rcp_f32 (1 ulp) is enough: the clamped 1-D minimiser only has to be near the true one — any point of
    // the edge gives an UPPER bound of the minimum, and the margin in thr covers the difference.  (An upper
    // bound could only cull too little... it is the cut-off side that must stay conservative: q at the
    // approximate minimiser >= true minimum, so the 0.1 % + 0.01 margin is what keeps the test safe.)
    const float inv_ca = __builtin_amdgcn_rcpf(a.z);
    const float inv_cc = __builtin_amdgcn_rcpf(b.x);
    uint32_t hits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = tx0 + (float)((q & 1) * 8);
        const float y0 = ty0 + (float)((q >> 1) * 8);
        if (rect_hit(a.x, a.y, a.z, a.w, b.x, inv_ca, inv_cc, thr, x0, y0, x0 + 7.f, y0 + 7.f)) hits |= (1u << q);
    }
    return hits;
}

// Move a wave-uniform 64-bit value into SGPRs (readfirstlane returns a SIGNED int: go through uint32_t,
// or the low half sign-extends into the high half).
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void blend_forward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ point_list,
                                                               const float4* __restrict__ splats,
                                                               float* __restrict__ out_color,
                                                               float* __restrict__ out_depth,
                                                               float* __restrict__ out_alpha,
                                                               float* __restrict__ final_T,
                                                               uint32_t* __restrict__ n_contrib) {
    __shared__ float4 s_a[kChunk];
    __shared__ float4 s_b[kChunk];
    __shared__ float4 s_c[kChunk];
    __shared__ uint64_t s_mask[4][4];          // [consumer quadrant][loader wave]

    const int n_tiles = f.gx * f.gy;
    const int tile = xcd_tile_remap(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int tile_x = tile % f.gx, tile_y = tile / f.gx;
    const int w = wave_id(), lane = lane_id();
    const int px = tile_x * kTile + (w & 1) * 8 + (lane & 7);
    const int py = tile_y * kTile + (w >> 1) * 8 + (lane >> 3);
    const bool inside = (px < f.W) && (py < f.H);
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tile_x * kTile), ty0 = (float)(tile_y * kTile);

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dz = 0.f, Aa = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (int base = 0; base < n; base += kChunk) {
        // workgroup-wide early exit (also the barrier that protects the LDS chunk from being overwritten)
        if (__syncthreads_and(done)) break;

        const int k = base + (int)threadIdx.x;
        uint32_t hits = 0;
        if (k < n) {
            const uint32_t id = point_list[range.x + k];
            const float4 a = splats[3 * (size_t)id + 0];
            const float4 b = splats[3 * (size_t)id + 1];
            const float4 c = splats[3 * (size_t)id + 2];
            s_a[threadIdx.x] = a; s_b[threadIdx.x] = b; s_c[threadIdx.x] = c;
            hits = quadrant_hits(a, b, tx0, ty0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t m = __ballot((hits >> q) & 1u);
            if (lane == 0) s_mask[q][w] = m;
        }
        __syncthreads();

        const bool wave_done = __all(done);
        if (!wave_done) {
            for (int lw = 0; lw < 4; ++lw) {
                uint64_t m = s_mask[w][lw];
                m = uniform_u64(m);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int j = lw * kWave + bit;
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const float dx = a.x - pxf, dy = a.y - pyf;
                    const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                    const float alpha = fminf(kAlphaMax, b.y * __expf(power));
                    bool ok = !done && (power <= 0.0f) && (alpha >= kAlphaMin);
                    const float test_T = T * (1.0f - alpha);
                    if (ok && test_T < kTEps) { done = true; ok = false; }
                    if (ok) {
                        const float4 c = s_c[j];
                        const float wgt = alpha * T;
                        Cr += c.x * wgt; Cg += c.y * wgt; Cb += c.z * wgt;
                        Dz += c.w * wgt; Aa += wgt;
                        T = test_T;
                        last = (uint32_t)(base + j + 1);
                    }
                }
                if (__all(done)) break;
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)py * f.W + px;
        const size_t hw = (size_t)f.H * f.W;
        out_color[pix] = Cr + T * f.bg[0];
        out_color[hw + pix] = Cg + T * f.bg[1];
        out_color[2 * hw + pix] = Cb + T * f.bg[2];
        out_depth[pix] = Dz;
        out_alpha[pix] = Aa;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

int launch_blend_forward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                         const float* splats, float* out_color, float* out_depth, float* out_alpha,
                         float* final_T, uint32_t* n_contrib, hipStream_t stream) {
    const int n_tiles = f.gx * f.gy;
    const int grid = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(grid), dim3(kBlock), 0, stream, f,
                       reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                       out_color, out_depth, out_alpha, final_T, n_contrib);
    return check_hip(hipGetLastError(), "blend_forward_kernel");
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF,
                                                                  false));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_move<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xF>(v);    // row_half_mirror
    v += dpp_move<0x140, 0xF>(v);    // row_mirror           -> every lane holds its row's sum
    v += dpp_move<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}

// --- transposing reductions -------------------------------------------------------------------------
// quad_transpose4(a,b,c,d): every lane returns the sum over its QUAD (4 lanes) of ONE of the four inputs,
// chosen by lane&3 (0:a 1:b 2:c 3:d).  Two butterfly levels: 4 selects + 2 DPP adds, then 2 selects + 1 DPP add.
__device__ __forceinline__ float quad_transpose4(float a, float b, float c, float d, int lane) {
    const bool odd = lane & 1;
    const float ab = (odd ? b : a) + dpp_move<0xB1, 0xF>(odd ? a : b);     // partner = lane^1
    const float cd = (odd ? d : c) + dpp_move<0xB1, 0xF>(odd ? c : d);
    const bool hi = lane & 2;
    return (hi ? cd : ab) + dpp_move<0x4E, 0xF>(hi ? ab : cd);             // partner = lane^2
}
// two inputs: lane&1 selects (0:a 1:b); summed over the quad.
__device__ __forceinline__ float quad_transpose2(float a, float b, int lane) {
    const bool odd = lane & 1;
    float v = (odd ? b : a) + dpp_move<0xB1, 0xF>(odd ? a : b);
    v += dpp_move<0x4E, 0xF>(v);
    return v;
}
// Sum over the 16 quads of the wave while keeping lane&3 (which identifies the quantity a lane carries):
// row_ror:4 / row_ror:8 sum the 4 quads of a 16-lane row; v_permlane16_swap / v_permlane32_swap (gfx950) add
// the same lane position of the other rows.  Every lane ends up with the wave total of its quantity.
__device__ __forceinline__ float quads_sum_all(float v) {
    v += dpp_move<0x124, 0xF>(v);    // row_ror:4
    v += dpp_move<0x128, 0xF>(v);    // row_ror:8
    {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // rows (0,0,2,2) + (1,1,3,3)
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // halves (lo,lo) + (hi,hi)
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}

constexpr int kGradSlots = 10;   // dx dy ddepth dopacity | dca dcb dcc | dr dg db

__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
    float* __restrict__ dsplats) {
    __shared__ float4 s_a[kChunk];
    __shared__ float4 s_b[kChunk];
    __shared__ float4 s_c[kChunk];
    __shared__ float s_grad[kChunk * kGradSlots];
    __shared__ uint32_t s_id[kChunk];
    __shared__ uint64_t s_mask[4][4];
    __shared__ uint32_t s_max[4];

    const int n_tiles = f.gx * f.gy;
    const int tile = xcd_tile_remap(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int tile_x = tile % f.gx, tile_y = tile / f.gx;
    const int w = wave_id(), lane = lane_id();
    const int px = tile_x * kTile + (w & 1) * 8 + (lane & 7);
    const int py = tile_y * kTile + (w >> 1) * 8 + (lane >> 3);
    const bool inside = (px < f.W) && (py < f.H);
    const float pxf = (float)px, pyf = (float)py;
    const float tx0 = (float)(tile_x * kTile), ty0 = (float)(tile_y * kTile);

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T_final = 1.0f, dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dA = 0.f;
    uint32_t last = 0;
    if (inside) {
        const size_t pix = (size_t)py * f.W + px;
        const size_t hw = (size_t)f.H * f.W;
        T_final = final_T[pix];
        last = n_contrib[pix];
        dC0 = dL_dcolor[pix]; dC1 = dL_dcolor[hw + pix]; dC2 = dL_dcolor[2 * hw + pix];
        if (dL_ddepth) dD = dL_ddepth[pix];
        if (dL_dalpha) dA = dL_dalpha[pix];
    }
    const float bg_dot = f.bg[0] * dC0 + f.bg[1] * dC1 + f.bg[2] * dC2;

    // highest list index any pixel of the tile blended
    uint32_t mx = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down((int)mx, off, kWave));
    if (lane == 0) s_max[w] = mx;
    __syncthreads();
    const uint32_t tile_last = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    if (tile_last == 0 || n == 0) return;
    const uint32_t wave_last = __shfl((int)mx, 0, kWave);   // lane 0 holds the wave max after shfl_down tree

    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_z = 0.f, acc_a = 0.f;
    float last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f, last_z = 0.f;

    const int first_chunk = ((int)tile_last - 1) / kChunk;
    for (int chunk = first_chunk; chunk >= 0; --chunk) {
        const int base = chunk * kChunk;
        const int k = base + (int)threadIdx.x;
        uint32_t hits = 0;
        uint32_t id = 0;
        if (k < n && k < (int)tile_last) {
            id = point_list[range.x + k];
            const float4 a = splats[3 * (size_t)id + 0];
            const float4 b = splats[3 * (size_t)id + 1];
            const float4 c = splats[3 * (size_t)id + 2];
            s_a[threadIdx.x] = a; s_b[threadIdx.x] = b; s_c[threadIdx.x] = c;
            hits = quadrant_hits(a, b, tx0, ty0);
        }
        s_id[threadIdx.x] = id;
#pragma unroll
        for (int s = 0; s < kGradSlots; ++s) s_grad[s * kChunk + threadIdx.x] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t m = __ballot((hits >> q) & 1u);
            if (lane == 0) s_mask[q][w] = m;
        }
        __syncthreads();

        if ((uint32_t)base < wave_last) {
            for (int lw = 3; lw >= 0; --lw) {
                uint64_t m = s_mask[w][lw];
                m = uniform_u64(m);
                while (m) {
                    const int bit = 63 - __builtin_clzll(m);
                    m &= ~(1ull << bit);
                    const int j = lw * kWave + bit;
                    const uint32_t pos = (uint32_t)(base + j);          // 0-based list index
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const float dx = a.x - pxf, dy = a.y - pyf;
                    const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                    const float G = __expf(power);
                    const float alpha = fminf(kAlphaMax, b.y * G);
                    const bool ok = (pos < last) && (power <= 0.0f) && (alpha >= kAlphaMin);
                    if (__ballot(ok) == 0ull) continue;                   // wave-uniform

                    float g_x = 0.f, g_y = 0.f, g_z = 0.f, g_o = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f;
                    float g_r = 0.f, g_g = 0.f, g_b = 0.f;
                    if (ok) {
                        const float4 c = s_c[j];
                        const float inv_one_m = __builtin_amdgcn_rcpf(1.0f - alpha);    // 1-alpha >= 0.01
                        T = T * inv_one_m;
                        const float wgt = alpha * T;
                        acc_r = last_alpha * last_r + (1.f - last_alpha) * acc_r;
                        acc_g = last_alpha * last_g + (1.f - last_alpha) * acc_g;
                        acc_b = last_alpha * last_b + (1.f - last_alpha) * acc_b;
                        acc_z = last_alpha * last_z + (1.f - last_alpha) * acc_z;
                        acc_a = last_alpha + (1.f - last_alpha) * acc_a;
                        last_r = c.x; last_g = c.y; last_b = c.z; last_z = c.w;
                        float dL_dalpha_ = (c.x - acc_r) * dC0 + (c.y - acc_g) * dC1 + (c.z - acc_b) * dC2 +
                                           (c.w - acc_z) * dD + (1.f - acc_a) * dA;
                        dL_dalpha_ *= T;
                        last_alpha = alpha;
                        dL_dalpha_ -= (T_final * inv_one_m) * bg_dot;
                        g_r = wgt * dC0; g_g = wgt * dC1; g_b = wgt * dC2;
                        g_z = wgt * dD;
                        const float dL_dG = b.y * dL_dalpha_;
                        const float gdx = G * dx, gdy = G * dy;
                        g_x = dL_dG * (-gdx * a.z - gdy * a.w);
                        g_y = dL_dG * (-gdy * b.x - gdx * a.w);
                        g_ca = -0.5f * gdx * dx * dL_dG;
                        g_cb = -gdx * dy * dL_dG;
                        g_cc = -0.5f * gdy * dy * dL_dG;
                        g_o = G * dL_dalpha_;
                    }
                    // 10 partial gradients -> 3 registers by two transposing butterfly levels inside each quad
                    // (lane&3 selects WHICH gradient a lane carries), then plain sums over the 16 quads.
                    // Result: lanes 0..3 hold the wave totals of (g_x,g_y,g_z,g_o) / (g_ca,g_cb,g_cc,g_r) /
                    // (g_g,g_b,g_g,g_b): 3 LDS atomics with distinct addresses instead of 10.
                    const float t0 = quad_transpose4(g_x, g_y, g_z, g_o, lane);
                    const float t1 = quad_transpose4(g_ca, g_cb, g_cc, g_r, lane);
                    const float t2 = quad_transpose2(g_g, g_b, lane);
                    const float r0 = quads_sum_all(t0);
                    const float r1 = quads_sum_all(t1);
                    const float r2 = quads_sum_all(t2);
                    if (lane < 4) {
                        const int k = lane;
                        atomicAdd(&s_grad[k * kChunk + j], r0);
                        atomicAdd(&s_grad[(4 + k) * kChunk + j], r1);
                        if (k < 2) atomicAdd(&s_grad[(8 + k) * kChunk + j], r2);
                    }
                }
            }
        }
        __syncthreads();

        // flush: thread t owns list entry t of the chunk
        if (hits) {
            float g[kGradSlots];
            bool any = false;
#pragma unroll
            for (int s = 0; s < kGradSlots; ++s) { g[s] = s_grad[s * kChunk + threadIdx.x]; any |= (g[s] != 0.f); }
#ifdef SCG_EXP_NOFLUSH
            any = any && (g[0] == 12345.f);
#endif
            if (any) {
                float* dst = dsplats + (size_t)id * SCG_SPLAT_FLOATS;
                unsafeAtomicAdd(dst + 0, g[0]); unsafeAtomicAdd(dst + 1, g[1]);
                unsafeAtomicAdd(dst + 2, g[2]); unsafeAtomicAdd(dst + 3, g[3]);
                unsafeAtomicAdd(dst + 4, g[4]); unsafeAtomicAdd(dst + 5, g[5]); unsafeAtomicAdd(dst + 6, g[6]);
                unsafeAtomicAdd(dst + 8, g[7]); unsafeAtomicAdd(dst + 9, g[8]); unsafeAtomicAdd(dst + 10, g[9]);
            }
        }
        __syncthreads();
    }
}

int launch_blend_backward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                          const float* splats, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                          float* dsplats, hipStream_t stream) {
    int rc = check_hip(hipMemsetAsync(dsplats, 0, (size_t)f.P * SCG_SPLAT_FLOATS * sizeof(float), stream),
                       "dsplats memset");
    if (rc) return rc;
    const int n_tiles = f.gx * f.gy;
    const int grid = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(blend_backward_kernel, dim3(grid), dim3(kBlock), 0, stream, f,
                       reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                       final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dsplats);
    return check_hip(hipGetLastError(), "blend_backward_kernel");
}

}  // namespace scg
