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

// One splat against ONE 8x8 pixel rectangle at pixel origin (x0, y0) — same test, same margins.
__device__ __forceinline__ bool splat_hits_rect(const float4& a, const float4& b, float x0, float y0) {
    const float L = __logf(255.0f * b.y);
    if (!(L >= -0.01f)) return b.y != b.y;
    const float thr = 2.0f * L * 1.001f + 0.01f;
    return rect_hit(a.x, a.y, a.z, a.w, b.x, __builtin_amdgcn_rcpf(a.z), __builtin_amdgcn_rcpf(b.x), thr, x0, y0,
                    x0 + 7.f, y0 + 7.f);
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
// ONE WAVE PER WORKGROUP: workgroup (tile, q) owns the 8x8-pixel quadrant q of a tile and walks the tile's list
// on its own — no workgroup barriers to wait at for a slower sibling quadrant, no LDS atomics, 4x more (and
// smaller) workgroups to balance over the 256 CUs, 2.5 KiB of LDS per wave so registers alone set the occupancy.
// The kernel is VALU-issue bound (profiles/README.md), so the inner loop minimises vector instructions per
// (splat, quadrant):
//   * one scalar recurrence instead of five.  With d_i = c_i . dL/dC (colour, depth and alpha channels folded
//     into one dot product) the "colour behind splat i" term of the classic formulation collapses to
//         B_i     = (sum_{j>i} w_j d_j + T_final * bg.dL/dC) / T_{i+1}
//         B_{i-1} = alpha_i d_i + (1 - alpha_i) B_i ,     B_last = bg . dL/dC
//         dL/dalpha_i = T_i (d_i - B_i)
//     i.e. the background behaves like one more, opaque, splat behind the list.
//   * the conic is staged pre-multiplied by 0.5*log2(e): e' = ca' dx + cb' dy and h' = cb' dx + cc' dy give
//     G = exp2(-(dx e' + dy h')) with a bare v_exp_f32, and the same e', h' are the position gradient; the
//     constant factors (-1/(0.5 log2 e), -0.5, 1/opacity) are applied once per sum, after the reduction.
//   * no divergent branch: a lane that does not blend the splat runs the update with alpha = 0 (a no-op on
//     its state) and contributes zeros.
//   * 10 partial gradients x 64 lanes -> 10 sums in ONE register by a fully transposing butterfly: every level
//     halves the number of live registers while it adds lanes, 2 instructions per output
//       lanes ^32 : v_permlane32_swap + add   (10 -> 5)      lanes ^16 : v_permlane16_swap + add   (5 -> 3)
//       banks ^2  : bank-masked row_ror:8 adds (3 -> 2)      banks ^1  : bank-masked row_shl/shr:4  (2 -> 1)
//     and two quad_perm adds finish inside the 4-lane bank: 24 VALU instead of 10 x 6.  Ten lanes then own ten
//     different sums and issue ONE global_atomic_add_f32 on the 48-byte gradient record of the splat.
constexpr float kHalfLog2e = 0.72134752044448170f;   // 0.5 * log2(e)

// lanes < 32 return (a[l] + a[l+32]), lanes >= 32 return (b[l-32] + b[l])
__device__ __forceinline__ float transpose_add_32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);       // r0 = [a.lo, b.lo], r1 = [a.hi, b.hi]
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// 16-lane rows 0,2 return a (row pair 0+1 / 2+3 added), rows 1,3 return b
__device__ __forceinline__ float transpose_add_16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                                    false, false);       // r0 = [a0,b0,a2,b2], r1 = [a1,b1,a3,b3]
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Sums over the 64 lanes of ten registers.  Lane (row r = lane>>4, bank b = (lane>>2)&3) returns
//   b == 0 : sum of v[ {0,2,1,3}[r] ]     b == 2 : sum of v[ {4,6,5,7}[r] ]     b odd : sum of v[ 8 + (r>>1) ]
// (cross-lane semantics pinned by tools/probes/dpp_probe.hip).
__device__ __forceinline__ float wave_reduce10(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                               float v7, float v8, float v9) {
    const float u0 = transpose_add_32(v0, v1), u1 = transpose_add_32(v2, v3), u2 = transpose_add_32(v4, v5);
    const float u3 = transpose_add_32(v6, v7), u4 = transpose_add_32(v8, v9);
    const float w0 = transpose_add_16(u0, u1), w1 = transpose_add_16(u2, u3), w2 = transpose_add_16(u4, u4);
    float y, x0, x1;
    asm("s_nop 1\n\t"                                                     // VALU write -> DPP read: 2 wait states
        "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"   // banks 0,1 <- w0 (+ bank^2)
        "v_add_f32_dpp %2, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"   //              w2 (+ bank^2)
        "v_add_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"   // banks 2,3 <- w1 (+ bank^2)
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"   // banks 0,2 <- x0 (+ bank+1)
        "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"   // banks 1,3 <- x1 (+ bank-1)
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(y), "=&v"(x0), "=&v"(x1)
        : "v"(w0), "v"(w1), "v"(w2));
    return y;
}

__global__ __launch_bounds__(kWave) void blend_backward_kernel(
    FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
    float* __restrict__ dsplats) {
    __shared__ float4 s_a[kWave];              // x, y, ca', cb'        (conic pre-multiplied by 0.5 log2 e)
    __shared__ float4 s_b[kWave];              // cc', opacity, 1/opacity, -
    __shared__ float4 s_c[kWave];              // r, g, b, depth

    // 4 consecutive groups of 8 workgroups = the 4 quadrants of 8 tiles, one tile per XCD (b % 8 picks the XCD)
    const int n_tiles = f.gx * f.gy;
    const int wg = blockIdx.x;
    const int quad = (wg >> 3) & 3;
    const int tile = xcd_tile_remap(((wg >> 5) << 3) | (wg & 7), n_tiles);
    if (tile >= n_tiles) return;
    const int tile_x = tile % f.gx, tile_y = tile / f.gx;
    const int lane = threadIdx.x;
    const int qx0 = tile_x * kTile + (quad & 1) * 8, qy0 = tile_y * kTile + (quad >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = (px < f.W) && (py < f.H);
    const float pxf = (float)px, pyf = (float)py;

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    float T = 1.0f, dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dA = 0.f;
    uint32_t last = 0;
    if (inside) {
        const size_t pix = (size_t)py * f.W + px;
        const size_t hw = (size_t)f.H * f.W;
        T = final_T[pix];
        last = n_contrib[pix];
        dC0 = dL_dcolor[pix]; dC1 = dL_dcolor[hw + pix]; dC2 = dL_dcolor[2 * hw + pix];
        if (dL_ddepth) dD = dL_ddepth[pix];
        if (dL_dalpha) dA = dL_dalpha[pix];
    }
    float behind = f.bg[0] * dC0 + f.bg[1] * dC1 + f.bg[2] * dC2;      // B_last

    // highest list index any pixel of the quadrant blended
    uint32_t mx = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down((int)mx, off, kWave));
    const int limit = min(n, __builtin_amdgcn_readfirstlane((int)mx));
    if (limit <= 0) return;

    // which of the ten sums this lane owns after wave_reduce10, where it goes in the 12-float gradient record
    // (dx dy ddepth dopacity | dca dcb dcc - | dr dg db -) and the constant factor it still needs
    const int row = lane >> 4, bank = (lane >> 2) & 3;
    int slot = -1;
    float scale = 1.0f;
    if ((lane & 3) == 0) {
        if (bank == 0) slot = (row == 0) ? 0 : (row == 1) ? 2 : (row == 2) ? 1 : 3;
        else if (bank == 2) slot = (row == 0) ? 4 : (row == 1) ? 6 : (row == 2) ? 5 : 8;
        else if (bank == 1 && (row & 1) == 0) slot = (row == 0) ? 9 : 10;
        if (slot == 0 || slot == 1) scale = -1.0f / kHalfLog2e;
        if (slot == 4 || slot == 6) scale = -0.5f;
        if (slot == 5) scale = -1.0f;
    }
    const bool owns_opacity = (slot == 3);

    for (int chunk = (limit - 1) / kWave; chunk >= 0; --chunk) {
        const int base = chunk * kWave;
        const int k = base + lane;
        bool hit = false;
        uint32_t id = 0;
        if (k < limit) {
            id = point_list[range.x + k];
            const float4 a = splats[3 * (size_t)id + 0];
            const float4 b = splats[3 * (size_t)id + 1];
            hit = splat_hits_rect(a, b, (float)qx0, (float)qy0);
            if (hit) {
                s_a[lane] = make_float4(a.x, a.y, kHalfLog2e * a.z, kHalfLog2e * a.w);
                s_b[lane] = make_float4(kHalfLog2e * b.x, b.y, 1.0f / b.y, 0.f);
                s_c[lane] = splats[3 * (size_t)id + 2];
            }
        }
        uint64_t m = __ballot(hit);
        // every gather has landed before the loop (vmcnt(0)): the only VMEM traffic inside it are fire-and-forget
        // atomics, which must never be waited for
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();

        while (m) {
            const int j = 63 - __builtin_clzll(m);
            m &= ~(1ull << j);
            const uint32_t pos = (uint32_t)(base + j);              // 0-based list index
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            asm("" ::"v"(b.w));                                     // keep it one ds_read_b128 (a b96 costs twice the LDS cycles)
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float e = a.z * dx + a.w * dy;
            const float h = a.w * dx + b.x * dy;
            const float t = dx * e + dy * h;                        // -log2 G
            const float oG = b.y * __builtin_amdgcn_exp2f(-t);
            const bool ok = (pos < last) && (t >= 0.0f) && (oG >= kAlphaMin);
            if (__ballot(ok) == 0ull) continue;                     // wave-uniform

            const float4 c = s_c[j];
            const float q0 = ok ? oG : 0.0f;                        // alpha before the 0.99 clamp, 0 if skipped
            const float alpha = __builtin_amdgcn_fmed3f(q0, 0.0f, kAlphaMax);
            const float one_m = 1.0f - alpha;                       // >= 0.01
            T *= __builtin_amdgcn_rcpf(one_m);                      // transmittance in front of this splat
            const float d = __builtin_fmaf(c.x, dC0, __builtin_fmaf(c.y, dC1, __builtin_fmaf(c.z, dC2, __builtin_fmaf(c.w, dD, dA))));
            const float q = q0 * ((d - behind) * T);                // opacity * G * dL/dalpha
            behind = __builtin_fmaf(one_m, behind, alpha * d);
            const float wgt = alpha * T;
            const float qdx = q * dx, qdy = q * dy;
            const float sum = wave_reduce10(q * e, q * h, wgt * dD, q,                  // dx dy ddepth dopacity
                                            qdx * dx, qdx * dy, qdy * dy, wgt * dC0,    // dca dcb dcc dr
                                            wgt * dC1, wgt * dC2);                      // dg db
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
            if (slot >= 0)
                unsafeAtomicAdd(dsplats + (size_t)sid * SCG_SPLAT_FLOATS + slot, sum * (owns_opacity ? b.z : scale));
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
    const int grid = ((n_tiles + 7) / 8) * 8 * 4;          // (tile, quadrant) workgroups of one wave
    hipLaunchKernelGGL(blend_backward_kernel, dim3(grid), dim3(kWave), 0, stream, f,
                       reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                       final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dsplats);
    return check_hip(hipGetLastError(), "blend_backward_kernel");
}

}  // namespace scg
