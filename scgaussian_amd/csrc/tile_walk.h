// tile_walk.h — how the tile-first binning cuts the Gaussians into slices and walks their tile rectangles.  Shared by the
// binning kernels (binning_tiles.hip) and the geometry kernel that builds the tile histogram while it produces the
// rectangles (geometry.hip: the one-call path).
#pragma once

#include "scg_common.h"

namespace scg {

constexpr uint32_t kCoopThreshold = 48;        // rectangles larger than this are walked by the whole wave

// q = n / d, r = n % d for n < 2^24, 0 < d < 2^16 (one v_rcp_f32 + fix-up instead of the ~50-instruction u32 divide)
__device__ __forceinline__ void divmod_small(uint32_t n, uint32_t d, uint32_t& q, uint32_t& r) {
    q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
    int rem = (int)n - (int)(q * d);
    if (rem < 0) { q -= 1; rem += (int)d; }
    if (rem >= (int)d) { q += 1; rem -= (int)d; }
    r = (uint32_t)rem;
}

// Visit every (tile, Gaussian id) instance of Gaussians [ga, gb) with one wave.  Small rectangles: one Gaussian per
// lane, each lane walks its own rectangle (no search, no division).  Large rectangles (a background blob can cover
// the whole screen) are walked by all 64 lanes together so that no lane serialises thousands of tiles.
// The visiting order is unspecified: callers only count / allocate slots.
// One rectangle per lane ({min_x | min_y << 16, width | height << 16}, zero size = nothing) owned by Gaussian `g`:
// visit every tile of every lane's rectangle.  Small rectangles are walked by their own lane (no search, no
// division); large ones (a background blob can cover the whole screen) by all 64 lanes together so that no lane
// serialises thousands of tiles.  g0 = id of lane 0's Gaussian when ids are consecutive, else pass ids per lane.
template <class F>
__device__ __forceinline__ void walk_rects(uint2 r, uint32_t g, int grid_x, F&& f) {
    const int lane = lane_id();
    const uint32_t wd = r.y & 0xFFFFu, ht = r.y >> 16;
    const uint32_t cnt = wd * ht;
    if (cnt && cnt <= kCoopThreshold) {
        uint32_t row_tile = (r.x >> 16) * (uint32_t)grid_x + (r.x & 0xFFFFu);
        for (uint32_t y = 0; y < ht; ++y, row_tile += (uint32_t)grid_x)
            for (uint32_t x = 0; x < wd; ++x) f(row_tile + x, g);
    }
    uint64_t big = __ballot(cnt > kCoopThreshold);
    while (big) {
        const int l = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t bx = (uint32_t)__shfl((int)r.x, l, kWave);
        const uint32_t by = (uint32_t)__shfl((int)r.y, l, kWave);
        const uint32_t bg = (uint32_t)__shfl((int)g, l, kWave);
        const uint32_t bw = by & 0xFFFFu, bn = bw * (by >> 16);
        const uint32_t org = (bx >> 16) * (uint32_t)grid_x + (bx & 0xFFFFu);
        for (uint32_t k = lane; k < bn; k += kWave) {
            uint32_t qy, qx;
            divmod_small(k, bw, qy, qx);
            f(org + qy * (uint32_t)grid_x + qx, bg);
        }
    }
}

// Workgroups of the histogram / scatter kernels have 16 waves: the per-lane work is a chain of dependent LDS
// atomics (and, in the scatter, a store behind each), so it is latency bound and needs many waves per SIMD.
constexpr int kBinThreads = 1024;
constexpr int kBinWaves = kBinThreads / kWave;

// wave w of workgroup b owns Gaussians [(16b+w) P / 16B, (16b+w+1) P / 16B): id order is depth-random, so equal
// Gaussian counts are balanced in instance count up to statistical noise.
__device__ __forceinline__ void wave_slice(uint32_t P, uint32_t nblocks, uint32_t b, uint32_t w, uint32_t& ga,
                                           uint32_t& gb) {
    const uint64_t slots = (uint64_t)nblocks * kBinWaves;
    const uint64_t s = (uint64_t)b * kBinWaves + w;
    ga = (uint32_t)(s * P / slots);
    gb = (uint32_t)((s + 1) * P / slots);
}

// The same cut on 256-Gaussian boundaries (the geometry kernel that histograms its own rectangles sums tiles_touched per
// 256 Gaussians like geometry_forward_kernel does): slice b = blocks [b NB / B, (b + 1) NB / B) of NB = ceil(P / 256).
__host__ __device__ __forceinline__ void block_slice(uint32_t P, uint32_t nblocks, uint32_t b, uint32_t& blk_a, uint32_t& blk_b) {
    const uint64_t nb256 = ((uint64_t)P + kBlock - 1) / kBlock;
    blk_a = (uint32_t)((uint64_t)b * nb256 / nblocks);
    blk_b = (uint32_t)((uint64_t)(b + 1) * nb256 / nblocks);
}

}  // namespace scg
