// blend.hip — 16x16-tile alpha blending of colour + depth + alpha, forward and per-pixel backward
// (outputs consumed at reference gaussian_renderer/__init__.py:100, depth at
// scene/gaussian_model.py:246-263, alpha at train.py:168).
//
// MI355X mapping
//   * ONE WAVE PER WORKGROUP: workgroup (tile, q) owns the 8x8-pixel QUADRANT q of a 16x16 tile, one pixel per
//     lane, and walks the tile's sorted list on its own.  Everything a wave decides (skip a splat, stop) is
//     wave-uniform; there is no workgroup barrier to wait at for a slower sibling quadrant, no LDS atomics,
//     4x more (and smaller) workgroups to balance over the 256 CUs, and with 2.5 KiB of LDS per wave the
//     registers alone set the occupancy (8 waves per SIMD).
//   * the list is walked in chunks of 64: lane l gathers the 48-byte splat record of list entry l (the table is
//     L2 / Infinity-Cache resident) and, while it holds the record in registers, tests it against the quadrant:
//     it evaluates the exact minimum of the conic's quadratic form over the 8x8 pixel rectangle and compares it
//     with the alpha >= 1/255 cut-off 2*ln(255*opacity) (with a safety margin).  Only hits are staged in LDS
//     (and only hits fetch their colour); the ballot of the test is the work list, held in two SGPRs.
//   * the wave then iterates ONLY over the set bits with scalar bit scans; records are read with wave-uniform
//     (broadcast) LDS reads: {x,y} = one ds_read_b64, {conic_a,2 conic_b,conic_c,opacity} = one ds_read_b128,
//     {r,g,b,depth} = one ds_read_b128 (only by trips somebody contributes to) — never a ds_read_b96 (twice the LDS
//     cycles).  The forward's trip is hand-written (see the comment in front of it).  The
//     test is conservative, so the skipped splats are exactly ones every pixel of the quadrant would have
//     skipped itself: results are unchanged, only the ~4x redundant work of the loose 3-sigma tile rectangle
//     disappears.
//   * the conic is staged pre-multiplied by 0.5*log2(e), so the Gaussian is a bare v_exp_f32 of the quadratic form.
//   * blockIdx -> (tile, quadrant) is XCD-aware (quadrant_workgroup): an XCD's private L2 sees a contiguous band
//     of tiles and all four quadrants of a tile.
//   * backward ("v3": per-pixel weights transposed through LDS, per-splat sums by lanes that own a (splat row, pixel
//     group), one atomic instruction per four splats) — see the comment in front of backward_walk.  Round 3 cut the walk
//     into 128-entry segments fed from checkpoints of the forward (branch exp/segmented-backward): same time (141.7 vs
//     142.2 us at S2 — the kernel is bound by instruction issue on every SIMD, not by the drain of its launch) and 6x the
//     rounding error, so the quadrant's whole list stays one work item.  Round 4 hand-wrote the walk (four copies of the trip,
//     one per row of the open block; -5.5 %, nearly all of it scalar instructions; the flush included): see backward_walk.
//   * what bounds the two kernels (profiles/README.md, round 2): not HBM (traffic is below the algorithmic bytes) and not
//     the instruction fetch path (tools/probes/ifetch_probe: the same work in twice the bytes costs the same) — the
//     vector pipe.  It is 46-59 % busy at the measured instruction costs, a scalar instruction costs a SIMD 4 cycles
//     (twice a plain vector one) and overlaps a vector one only partly, and the forward answers to the vector
//     instruction count of its trip almost 1:1 (four v_mov more per trip: +6 %; thirteen scalar instructions and
//     branches less: -4 %).  Hence the hand-written forward trip below.
#include "scg_common.h"
#include "tile_sort.h"

namespace scg {

constexpr float kHalfLog2e = 0.72134752044448170f;   // 0.5 * log2(e)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Conservative "does this splat reach any pixel of the 8x8 quadrant at pixel origin (x0, y0)" test.
// record: a = {x, y, conic_a, conic_b}, b = {conic_c, opacity, cull_thr, cull_slope}.
// A pixel blends the splat only if q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= 2 ln(255 opacity) (alpha >= 1/255) —
// cull_thr is that bound with a 0.1 % + 0.01 margin (the fp32 evaluation error of q is < 1e-4 here), so no pixel
// that would pass its own test is culled.  q is convex: with n = the point of the rectangle nearest to the centre
// (0 along an axis on which the centre is inside), the minimum over the rectangle lies on the edge x = n.x or on
// the edge y = n.y, and on an edge it is a clamped 1-D parabola minimum — exact, no sampling, no branches; a
// centre inside the rectangle gives q = 0.  NaN anywhere -> hit (never cull on a malformed record).
__device__ __forceinline__ bool splat_hits_rect(const float4& a, const float4& b, float x0, float y0) {
    const float dx0 = x0 - a.x, dy0 = y0 - a.y, dx1 = dx0 + 7.0f, dy1 = dy0 + 7.0f;
    const float nx = __builtin_amdgcn_fmed3f(0.0f, dx0, dx1), ny = __builtin_amdgcn_fmed3f(0.0f, dy0, dy1);
    const float cb2 = a.w + a.w;
    const float ya = __builtin_amdgcn_fmed3f(b.w * nx, dy0, dy1);                        // -cb/cc * nx, clamped
    const float qa = nx * (a.z * nx + cb2 * ya) + b.x * ya * ya;
    const float xb = __builtin_amdgcn_fmed3f(-a.w * ny * __builtin_amdgcn_rcpf(a.z), dx0, dx1);
    const float qb = ny * (b.x * ny + cb2 * xb) + a.z * xb * xb;
    return !((qa > b.z) && (qb > b.z));
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// (tile, quadrant) of workgroup wg: 4 consecutive groups of 8 workgroups are the 4 quadrants of 8 tiles, one tile per
// XCD (wg % 8 picks the XCD); the tile is slot wg / 32 of that XCD's band in the launch order behind the ranges
// (scg_common.h: tile_order_slots) — a contiguous band of tiles per XCD, longest lists first.
__device__ __forceinline__ int quadrant_workgroup(int wg, int n_tiles, const uint2* __restrict__ ranges, int& quad) {
    quad = (wg >> 3) & 3;
    const int per = (n_tiles + 7) >> 3;
    const uint32_t* order = reinterpret_cast<const uint32_t*>(ranges) + 2 * (size_t)n_tiles;
    return (int)order[(wg & 7) * per + (wg >> 5)];
}

// The LDS operations of ONE wave execute in order: staging stores and the trips' broadcast reads of a wave's own record
// planes need no hardware barrier, only the compiler must not reorder them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One wave blends the 8x8 quadrant `quad` of `tile` front to back.  s_rec: this wave's three planes of 64 16-byte records,
// addressed by the trip's hand-written code with one register:
//   [0] r, g, b, depth      [1] ca', 2 cb', cc', opacity (conic pre-multiplied by 0.5 log2 e)      [2] x, y, -, -
// (x, y NOT next to the conic in one record: asked for that, the compiler keeps x, y in registers beside the scaled
//  conic, copies them there right behind the NEXT chunk's gather and waits for the gather on the spot — in front
//  of the blending it was issued early to hide behind)
// list: the tile's sorted ids — in global memory (point_list + range.x) or, behind the sort of the kernel that sorts its own
// tile, in LDS (an address_space(3) pointer: the two chunk-ahead id fetches are LDS reads then, not a trip to the L2)
typedef const uint32_t __attribute__((address_space(3))) * LdsIds;
template <class IdList>
__device__ __forceinline__ void forward_walk(float4* s_rec, const FrameDev& f, int tile, int quad, int lane, uint2 range,
                                             IdList list,
                                             const float4* __restrict__ splats, float* __restrict__ out_color,
                                             float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                             float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    const int tile_x = tile % f.gx, tile_y = tile / f.gx;
    const int qx0 = tile_x * kTile + (quad & 1) * 8, qy0 = tile_y * kTile + (quad >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = (px < f.W) && (py < f.H);
    const float pxf = (float)px, pyf = (float)py;

    const int n = (int)(range.y - range.x);
    // accumulated alpha is not carried: sum_i alpha_i T_i telescopes to 1 - T.
    // A pixel that has terminated (or lies outside the image) carries its transmittance NEGATED: every later test
    // T - alpha T >= 1e-4 then fails by itself, so no per-lane "done" mask has to be maintained on the scalar pipe
    // (the forward is co-bound by it: ~0.8 scalar instructions per vector instruction before this).
    float T = inside ? 1.0f : -1.0f;
    // the background colour is read HERE, into scalar registers: at the end of the walk the load was a memory round trip
    // every wave waited for right before its output stores
    int bgi0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f.bg[0]));
    int bgi1 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f.bg[1]));
    int bgi2 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f.bg[2]));
    asm volatile("" : "+s"(bgi0), "+s"(bgi1), "+s"(bgi2));
    f32x2 Crg = {0.f, 0.f}, Cbz = {0.f, 0.f};
    uint32_t last = 0;
    int n_blended = 0;                   // list entries this wave blends: the tile's cost for the next render of this camera
#ifndef SCG_FWD_TRIP_CXX
    // LDS offset of the record planes (the low half of a generic LDS address is the offset) and the full EXEC mask
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(s_rec);
    const uint64_t exec_all = __builtin_amdgcn_read_exec();
#endif

    // software pipeline over the chunks: list ids are fetched two chunks ahead and the 48-byte records one chunk
    // ahead, so both gathers are in flight while the wave blends the current chunk (lanes past the end of the
    // list re-fetch its last entry and never report a hit)
    uint32_t id_next = 0;
    float4 ra, rb, rc;
    if (n > 0) {
        // (unconditional, index-clamped loads: a select around a load makes the compiler wait for it on the spot)
        const uint32_t id0 = list[min(lane, n - 1)];
        id_next = list[min(kWave + lane, n - 1)];
        ra = splats[3 * (size_t)id0 + 0]; rb = splats[3 * (size_t)id0 + 1]; rc = splats[3 * (size_t)id0 + 2];
    }

    for (int base = 0; base < n; base += kWave) {
        if (__all(T < 0.0f)) break;
        const bool hit = (base + lane < n) && splat_hits_rect(ra, rb, (float)qx0, (float)qy0);
        if (hit) {
            *reinterpret_cast<float2*>(&s_rec[2 * kWave + lane]) = make_float2(ra.x, ra.y);
            s_rec[kWave + lane] = make_float4(kHalfLog2e * ra.z, 2.0f * kHalfLog2e * ra.w, kHalfLog2e * rb.x, rb.y);
            s_rec[lane] = rc;
        }
        uint64_t m = __ballot(hit);
        if (base + kWave < n) {
            ra = splats[3 * (size_t)id_next + 0]; rb = splats[3 * (size_t)id_next + 1];
            rc = splats[3 * (size_t)id_next + 2];
            id_next = list[min(base + 2 * kWave + lane, n - 1)];
        }
        wave_sync();

        n_blended += __builtin_popcountll(m);                           // (scalar, once per chunk)
#ifdef SCG_FWD_TRIP_CXX
        // The trip as the compiler writes it: 10 scalar instructions and 3 branches per trip.  Built as libscg_raster_cxx.so
        // (scgaussian_amd/build.py) and held against the hand-written trip BIT FOR BIT by tests/test_gpu_parity.py — the same
        // operations in the same order (every fused multiply-add explicit), so any difference is a broken assumption of the
        // inline assembly, not arithmetic.
        while (m) {
            const int j = __builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));         // one scalar op instead of the 64-bit m & (m - 1)
            const float2 c = *reinterpret_cast<const float2*>(&s_rec[2 * kWave + j]);
            const float4 q = s_rec[kWave + j];
            const float dx = c.x - pxf, dy = c.y - pyf;
            const float u = __builtin_fmaf(q.y, dy, q.x * dx);      // ca' dx + 2 cb' dy
            const float t = __builtin_fmaf(u, dx, (q.z * dy) * dy); // -log2 G = ca' dx^2 + 2 cb' dx dy + cc' dy^2
            const float alpha = fminf(kAlphaMax, q.w * __builtin_amdgcn_exp2f(-t));
            const float wgt = T * alpha;
            const float test_T = __builtin_fmaf(-T, alpha, T);      // T (1 - alpha)
            if ((t >= 0.0f) && (alpha >= kAlphaMin)) {
                const bool contributes = test_T >= kTEps;
                const float T_prev = T;
                T = contributes ? test_T : -fabsf(T_prev);
                if (contributes) {
                    const float4 col = s_rec[j];
                    const f32x2 ww = {wgt, wgt};
                    Crg = __builtin_elementwise_fma((f32x2){col.x, col.y}, ww, Crg);
                    Cbz = __builtin_elementwise_fma((f32x2){col.z, col.w}, ww, Cbz);
                    last = (uint32_t)(base + j + 1);
                }
            }
        }
#else
        // The trip, hand-written.  A scalar instruction costs a SIMD 4 cycles (tools/probes/ifetch_probe: twice a plain
        // vector one) and the compiler's trip spends ten of them plus three branches on bit scan, address, two nested
        // exec-mask regions and the list index.  Here the tests narrow EXEC themselves (v_cmpx), terminating a pixel is
        // "set the sign on the lanes that blend, overwrite with T(1-alpha) on the lanes that contribute", the list index is
        // kept chunk-local, and one s_mov restores EXEC: 4 scalar instructions and the loop branch per trip.
        //   v48,v49 dx,dy | v[52:55] ca' 2cb' cc' opacity | v56 alpha -> weight | v57 T(1-alpha) | v58..v60 partial sums, t
        //   v[44:47] r g b depth | v63 LDS address.     (trans result v56 is first read two instructions later: gfx950's
        //   one-wait-state forwarding hazard; no DPP, no lane-select reads of freshly written SGPRs)
        int last_j = -1;
        while (m) {
            const int j = __builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));         // one scalar op instead of the 64-bit m & (m - 1)
            asm volatile(
                "v_lshl_add_u32 v63, %[j], 4, %[lds]\n\t"
                "ds_read_b64 v[48:49], v63 offset:2048\n\t"
                "ds_read_b128 v[52:55], v63 offset:1024\n\t"
                "s_waitcnt lgkmcnt(1)\n\t"
                "v_sub_f32_e32 v48, v48, %[px]\n\t"
                "v_sub_f32_e32 v49, v49, %[py]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mul_f32_e32 v58, v52, v48\n\t"                  // ca' dx
                "v_mul_f32_e32 v59, v54, v49\n\t"                  // cc' dy
                "v_fmac_f32_e32 v58, v53, v49\n\t"                 // ca' dx + 2 cb' dy
                "v_mul_f32_e32 v60, v59, v49\n\t"                  // cc' dy^2
                "v_fmac_f32_e32 v60, v58, v48\n\t"                 // t = ca' dx^2 + 2 cb' dx dy + cc' dy^2 = -log2 G
                "v_exp_f32_e64 v56, -v60\n\t"
                "v_cmpx_le_f32_e32 vcc, 0, v60\n\t"                // EXEC: t >= 0
                "v_mul_f32_e32 v56, v55, v56\n\t"
                "v_min_f32_e32 v56, 0x3f7d70a4, v56\n\t"           // alpha = min(0.99, opacity G)
                "v_cmpx_le_f32_e32 vcc, %[amin], v56\n\t"          // EXEC: ... and alpha >= 1/255
                "v_fma_f32 v57, -%[T], v56, %[T]\n\t"              // T (1 - alpha)
                "v_mul_f32_e32 v56, %[T], v56\n\t"                 // weight = T alpha
                "v_or_b32_e32 %[T], 0x80000000, %[T]\n\t"          // lanes that blend: terminated (idempotent) ...
                "v_cmpx_le_f32_e32 vcc, %[eps], v57\n\t"           // EXEC: ... and T (1 - alpha) >= 1e-4
                "v_mov_b32_e32 %[T], v57\n\t"                      // ... unless they contribute
                "ds_read_b128 v[44:47], v63\n\t"
                "v_mov_b32_e32 %[lj], %[j]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_pk_fma_f32 %[crg], v[44:45], v[56:57], %[crg] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[cbz], v[46:47], v[56:57], %[cbz] op_sel_hi:[1,0,1]\n\t"
                "s_mov_b64 exec, %[all]"
                : [T] "+v"(T), [crg] "+v"(Crg), [cbz] "+v"(Cbz), [lj] "+v"(last_j)
                : [j] "s"(j), [lds] "v"(lds_base), [px] "v"(pxf), [py] "v"(pyf), [amin] "s"(kAlphaMin), [eps] "s"(kTEps),
                  [all] "s"(exec_all)
                : "memory", "vcc", "v44", "v45", "v46", "v47", "v48", "v49", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
                  "v59", "v60", "v63");
        }
        if (last_j >= 0) last = (uint32_t)(base + last_j + 1);
#endif
        wave_sync();
    }

    // (the busiest quadrant's trips; with the list length — the sort's share — mixed in, n_blended + n / 16: no change, round 5)
    if (f.cost_out && lane == 0) atomicMax(f.cost_out + tile, (uint32_t)n_blended);
    if (inside) {
        T = fabsf(T);
        const size_t pix = (size_t)py * f.W + px;
        const size_t hw = (size_t)f.H * f.W;
        out_color[pix] = Crg[0] + T * __builtin_bit_cast(float, bgi0);
        out_color[hw + pix] = Crg[1] + T * __builtin_bit_cast(float, bgi1);
        out_color[2 * hw + pix] = Cbz[0] + T * __builtin_bit_cast(float, bgi2);
        out_depth[pix] = Cbz[1];
        out_alpha[pix] = 1.0f - T;
        if (final_T) final_T[pix] = T;              // (both null: a render nobody will differentiate — scg_forward's
        if (n_contrib) n_contrib[pix] = last;       //  SCG_FORWARD_NO_BACKWARD_STATE)
    }
}


__global__ __launch_bounds__(kWave) void blend_forward_kernel(FrameDev f, const uint2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ point_list,
                                                              const float4* __restrict__ splats,
                                                              float* __restrict__ out_color,
                                                              float* __restrict__ out_depth,
                                                              float* __restrict__ out_alpha,
                                                              float* __restrict__ final_T,
                                                              uint32_t* __restrict__ n_contrib,
                                                              float4* __restrict__ zero_fill, uint32_t zero_vec) {
    __shared__ float4 s_rec[3 * kWave];
    // optional: clear the gradient records of the coming backward here (one coalesced 16-byte store per lane and
    // trip) instead of a separate memset launch in front of blend_backward
    if (zero_fill) {
        for (uint32_t i = blockIdx.x * kWave + threadIdx.x; i < zero_vec; i += gridDim.x * kWave)
            zero_fill[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int n_tiles = f.gx * f.gy;
    int quad;
    const int tile = quadrant_workgroup(blockIdx.x, n_tiles, ranges, quad);
    if (tile >= n_tiles) return;
    const uint2 range = ranges[tile];
    forward_walk(s_rec, f, tile, quad, (int)threadIdx.x, range, point_list + range.x, splats, out_color, out_depth, out_alpha,
                 final_T, n_contrib);
}

// ---- the forward blend that sorts its own tile (the one-call path, scg_forward) ------------------------------------
// One workgroup of FOUR quadrant waves per tile: together they sort the tile's list segment in LDS (the binning stage's
// per-tile bucket sort, tile_sort.h), write the canonical order to point_list (the backward needs it too), and then every wave
// walks the list on its own exactly as blend_forward_kernel does.  A latency-bound sort next to an
// issue-bound blend: the tiles of a compute unit are in different phases, the sort's waiting fills the blend's idle issue
// slots, and one launch with its ramp and drain disappears (S3: 47.9 us of tile_sort_kernel + 134.9 us of blend before).
// Lists longer than kFusedMaxN were sorted by the rare-size kernel before this launch (long_presorted) and are walked from
// global memory — or, when the caller skipped that launch on the promise that the frame has no such list
// (SCG_FORWARD_SKIP_RARE_SORT) and one shows up after all, sorted here by the tile's own workgroup through global scratch
// (sort_long_list with 256 threads and 1 024 LDS counters: the same result, 2-7x slower than the 1 024-thread kernel on a
// clustered scene — profiles/README.md round 3 — which is why it is only the fallback).  f.long_out, a host-visible word,
// receives the number of such lists: the caller's next render of this camera launches the rare-size kernel again.
template <int NW>
__device__ __forceinline__ void fused_long_list_fallback(unsigned char* smem, const uint32_t* __restrict__ depth_keys,
                                                      uint32_t* __restrict__ list, int n, uint64_t* __restrict__ keys,
                                                      uint64_t* __restrict__ keys2) {
    if (!sort_long_list<NW * kWave, kFusedLongBuckets, 2>(smem, depth_keys, list, n, keys, keys2)) {
        __syncthreads();                              // heavily tied depths: the bitonic network on the composites
        for (int i = threadIdx.x; i < n; i += NW * kWave) {
            const uint32_t id = list[i];
            keys[i] = ((uint64_t)depth_keys[id] << 32) | (uint64_t)id;
        }
        __syncthreads();
        bitonic_sort_asc(keys, n, true);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NW * kWave) list[i] = (uint32_t)keys[i];
    }
}

// MAX_N / CNT: list entries the workgroup sorts in LDS and its bucket counters — <kFusedMaxN, kFusedCounters> at eight waves
// per SIMD for the usual frames, <kFusedDenseMaxN, kFusedDenseCounters> (36 KiB: four workgroups per compute unit) with NW = 8
// sorting waves (four of them leave before the walk) for dense ones, whose 8-wave sort kernel was 49 us of a 210 us forward at a
// million Gaussians on 960x540 (round 5).
template <int MAX_N, int CNT, int WPE, int NW = 4>
__global__ __launch_bounds__(NW * kWave) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void tile_blend_forward_kernel(
    FrameDev f, const uint2* __restrict__ ranges, uint32_t* __restrict__ point_list,
    const uint32_t* __restrict__ depth_keys, int id_bits, const float4* __restrict__ splats,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float4* __restrict__ zero_fill, uint32_t zero_vec,
    uint64_t* __restrict__ spill, uint64_t* __restrict__ spill2, const uint32_t* __restrict__ class_counts,
    int long_presorted) {
    // LDS: the sort's arrays — the ids first: they stay, in their final order, for the walk — and, over everything behind the ids,
    // the four quadrant waves' record planes (12 KiB).  Half as many buckets as list entries can be (two entries per bucket on
    // average at a full list).  18.2 KiB for the usual variant: eight workgroups per compute unit by LDS as by wave slots (until
    // round 5 the planes lay over the ids too: 16.2 KiB, nine by LDS — measured with 1 024-entry lists: the ninth buys nothing)
    typedef TileSortLds<NW, MAX_N, CNT> Lds;
    constexpr size_t kIdsBytes = sizeof(uint32_t) * MAX_N, kPlanesBytes = 4 * 3 * kWave * sizeof(float4);
    constexpr size_t kLdsBytes = sizeof(Lds) > kIdsBytes + kPlanesBytes ? sizeof(Lds) : kIdsBytes + kPlanesBytes;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    Lds& L = *reinterpret_cast<Lds*>(smem);
    static_assert(offsetof(Lds, id) == 0, "the sorted ids lie in front of the record planes");
    static_assert(kLdsBytes >= (2 * kFusedLongBuckets + 4 + 2 * 4) * sizeof(uint32_t), "the fallback's counters must fit");
    if (zero_fill) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_vec; i += gridDim.x * blockDim.x)
            zero_fill[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // how many lists of this frame are longer than this kernel sorts in LDS (counted by the scatter's publishing workgroups,
    // complete before this launch started): told to the host for its next render of this camera
    if (f.long_out && blockIdx.x == 0 && threadIdx.x == 0) {
        f.long_out[0] = class_counts[0] + class_counts[1];
        f.long_out[1] = class_counts[3];                       // ... and how many VERY long ones (beyond 16 384 entries)
    }
    const int n_tiles = f.gx * f.gy, per = (n_tiles + 7) >> 3;
    const uint32_t* order = reinterpret_cast<const uint32_t*>(ranges) + 2 * (size_t)n_tiles;
    const int tile = (int)order[(blockIdx.x & 7) * per + (blockIdx.x >> 3)];      // band = XCD, slot in the band's launch order
    if (tile >= n_tiles) return;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
#ifdef SCG_PROBE_TIMELINE                    // tools/probes/blend_timeline.py: per-workgroup clocks behind the other kernels' logs
    const uint32_t tp0 = (uint32_t)wall_clock64();
#endif
    const bool ids_in_lds = n >= 2 && n <= MAX_N;
    if (ids_in_lds) sort_one_tile<NW, MAX_N, CNT, true>(L, range, depth_keys, point_list, id_bits);
    else if (n > MAX_N && !long_presorted)
        fused_long_list_fallback<NW>(smem, depth_keys, point_list + range.x, n, spill + range.x, spill2 + range.x);
    // the sorted ids are in L.id (lists this workgroup sorted in LDS; on their way to point_list for the backward) or in
    // point_list (visible to the whole workgroup behind the barrier); the rest of the sort's LDS is free
#ifdef SCG_PROBE_TIMELINE
    const uint32_t tp1 = (uint32_t)wall_clock64();
    uint32_t pr[6];
    for (int k = 0; k < 6; ++k) pr[k] = L.probe[k];
    __syncthreads();
#else
    // (a list sorted in LDS: the sort's last barrier stands behind its last read of the scratch the record planes overwrite and
    //  behind the last write of L.id — no second one)
    if (!ids_in_lds) __syncthreads();
#endif
    const int quad = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & (kWave - 1);
    if (NW > 4 && quad >= 4) return;                 // the waves that only helped to sort
#ifdef SCG_PROBE_TIMELINE
    const uint32_t tp2 = (uint32_t)wall_clock64();
#endif
    float4* planes = reinterpret_cast<float4*>(smem + kIdsBytes) + quad * 3 * kWave;
    if (ids_in_lds)
        forward_walk(planes, f, tile, quad, lane, range, reinterpret_cast<LdsIds>((uintptr_t)(uint32_t)reinterpret_cast<uintptr_t>(&L.id[0])),
                     splats, out_color, out_depth, out_alpha, final_T, n_contrib);
    else
        forward_walk(planes, f, tile, quad, lane, range, point_list + range.x, splats, out_color, out_depth, out_alpha, final_T,
                     n_contrib);
#ifdef SCG_PROBE_TIMELINE
    if (f.cost_out && lane == 0) {
        uint32_t* tl = f.cost_out + n_tiles + 65792 + 32768 + ((size_t)blockIdx.x * 4 + quad) * 8;
        tl[0] = tp0; tl[1] = tp1; tl[2] = tp2; tl[3] = (uint32_t)wall_clock64(); tl[4] = (uint32_t)n; tl[5] = (uint32_t)tile;
        tl[6] = 0u; tl[7] = 0xB1E9D000u;
        if (quad == 0) {        // the sort's internal clocks: a second record behind the kernel's (index n_slots * 4 + blockIdx)
            uint32_t* t2 = f.cost_out + n_tiles + 65792 + 32768 + ((size_t)tile_order_slots(n_tiles) * 4 + blockIdx.x) * 8;
            for (int k = 0; k < 6; ++k) t2[k] = pr[k];
            t2[6] = tp0; t2[7] = 0x50B70000u;
        }
    }
#endif
}

int launch_tile_blend_forward(const FrameDev& f, const uint32_t* ranges, uint32_t* point_list, const uint32_t* depth_keys,
                              const float* splats, float* out_color, float* out_depth, float* out_alpha,
                              float* final_T, uint32_t* n_contrib, float* dsplats_zero, void* bin_scratch, int64_t R,
                              bool long_presorted, hipStream_t stream) {
    const int n_tiles = f.gx * f.gy;
    int id_bits = 8;
    while (id_bits < 32 && (1ll << id_bits) < (long long)f.P) id_bits += 8;
    const TileBinningLayout BL = tile_binning_layout(f.P, R, n_tiles);
    uint64_t* spill = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(bin_scratch) + BL.spill);
    const bool dense = fused_max_list(R, n_tiles) == kFusedDenseMaxN;
    auto kernel = dense ? tile_blend_forward_kernel<kFusedDenseMaxN, kFusedDenseCounters, kFusedDenseWpe, kFusedDenseWaves>
                        : tile_blend_forward_kernel<kFusedMaxN, kFusedCounters, 8>;
    hipLaunchKernelGGL(kernel, dim3(tile_order_slots(n_tiles)), dim3((dense ? kFusedDenseWaves : 4) * kWave), 0, stream, f,
                       reinterpret_cast<const uint2*>(ranges), point_list, depth_keys, id_bits,
                       reinterpret_cast<const float4*>(splats), out_color, out_depth, out_alpha, final_T, n_contrib,
                       reinterpret_cast<float4*>(dsplats_zero), (uint32_t)((size_t)f.P * SCG_DSPLAT_FLOATS / 4),
                       spill, spill + R,
                       reinterpret_cast<const uint32_t*>(reinterpret_cast<char*>(bin_scratch) + BL.class_counts),
                       long_presorted ? 1 : 0);
    return check_hip(hipGetLastError(), "tile_blend_forward_kernel");
}

int launch_blend_forward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                         const float* splats, float* out_color, float* out_depth, float* out_alpha,
                         float* final_T, uint32_t* n_contrib, float* dsplats_zero, hipStream_t stream) {
    const int n_tiles = f.gx * f.gy;
    const int grid = ((n_tiles + 7) / 8) * 8 * 4;          // (tile, quadrant) workgroups of one wave
    hipLaunchKernelGGL(blend_forward_kernel, dim3(grid), dim3(kWave), 0, stream, f,
                       reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                       out_color, out_depth, out_alpha, final_T, n_contrib, reinterpret_cast<float4*>(dsplats_zero),
                       (uint32_t)((size_t)f.P * SCG_DSPLAT_FLOATS / 4));
    return check_hip(hipGetLastError(), "blend_forward_kernel");
}


// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// One wave owns the 8x8-pixel quadrant q of a tile and walks (a segment of) the tile's list back to front on its own — no
// workgroup barriers to wait at for a slower sibling quadrant, no LDS atomics, 5 KiB of LDS per wave so registers alone set
// the occupancy.  The kernel is vector-issue bound (profiles/README.md), so the walk minimises vector instructions per
// (splat, quadrant):
//   * one scalar recurrence instead of five.  With d_i = c_i . dL/dC (colour, depth and alpha channels folded
//     into one dot product) the "colour behind splat i" term of the classic formulation collapses to
//         B_i     = (sum_{j>i} w_j d_j + T_final * bg.dL/dC) / T_{i+1}
//         B_{i-1} = alpha_i d_i + (1 - alpha_i) B_i ,     B_last = bg . dL/dC
//         dL/dalpha_i = T_i (d_i - B_i)
//     i.e. the background behaves like one more, opaque, splat behind the list.
//   * the conic is staged pre-multiplied by 0.5*log2(e): G = exp2(-(ca' dx^2 + 2 cb' dx dy + cc' dy^2)) is a bare v_exp_f32
//     of a five-instruction quadratic form; the constant factors (-1/(0.5 log2 e), -0.5, 1/opacity) are applied once per
//     Gaussian by the geometry backward (the record holds raw sums).
//   * no divergent branch: a lane that does not blend the splat runs the update with alpha = 0 (a no-op on
//     its state) and contributes zeros.
//   * the ten per-splat sums are contractions over the 64 pixels of only TWO per-pixel weights,
//         q_p = opacity G dL/dalpha   (against 1, dx, dy, dx^2, dx dy, dy^2)        w_p = alpha T   (against dL/dC_rgb, dL/dD)
//     so the per-pixel loop only computes q and w and parks them in LDS (one 8-byte store per lane).  After FOUR splats
//     the [4 x 64] block is read back TRANSPOSED: lane = (splat row r = lane >> 4, pixel group g = lane & 15) owns the 4
//     pixels {g, g+16, g+32, g+48} of the quadrant — all in ONE pixel column, so dx is a per-lane constant and the three
//     x-moments follow from the y-sums after the loop — and accumulates its splat's ten sums over them (9 instructions per
//     pixel, 4 pixels, for 4 splats at once).  What is left to reduce are the 16 lanes of a row: the four in-row levels of
//     a transposing butterfly (row_reduce10: 22 DPP / select instructions, no cross-row step), once per FOUR splats, and ONE
//     global_atomic_add_f32 instruction per four splats (10 lanes of each row, distinct records; never two lanes of one
//     instruction on the same address: measured 4x the kernel time when they are).
//
// The record it writes holds RAW sums (geometry_backward_kernel applies the conic map and the constant factors once
// per Gaussian):  [0] sum q dx  [1] sum q dy  [2] dL/ddepth  [3] sum q | [4] sum q dx^2  [5] sum q dx dy  [6] sum q dy^2 | [8..10] dL/drgb
constexpr int kSlots = 4;                        // splats per transposed step = DPP rows of the wave
constexpr int kWStride = 2 * kWave;              // floats per row: 64 x (q, w).  (A pad of 32 floats makes the transposed reads of
                                                 // rows r, r+1 conflict-free and costs LDS: measured 1 % (S2) to 6 % (S4) SLOWER.)

// Sums over the 16 lanes of every DPP row of ten registers.  Lane (bank b = (lane >> 2) & 3, q = lane & 3) of a row returns:
//   q == 0 : sum of v[b]     q == 1 : sum of v[4 + b]     q >= 2 : sum of v[8 + (b & 1)]
// The four levels transpose while they add (10 -> 5 -> 3 -> 2 -> 1 registers: bank-masked row_shl/shr:4 and row_ror:8 adds,
// then select + quad_perm adds).  Order of the levels chosen by measured instruction cost (tools/probes/valu_rate.hip: DPP add
// 1.4, v_cndmask 1 fma-slots); hand-scheduled: every DPP read is at least two instructions behind the write of its source
// (cross-lane semantics pinned by tools/probes/dpp_probe.hip).
// odd, hi: the lane masks lane & 1 and lane & 2 (0xAAAA..., 0xCCCC...) in scalar register pairs the caller keeps alive (as
// literals the compiler re-assembles the pairs with two scalar moves per call).
__device__ __forceinline__ float row_reduce10(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                              float v7, float v8, float v9, uint64_t odd, uint64_t hi) {
    float y, t0, t1, t2, t3, t4, t5, t6, t7;
    asm("s_nop 1\n\t"
        // bank ^ 1: even banks keep the first input of a pair, odd banks the second
        "v_add_f32_dpp %1, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %13, %13 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %15, %15 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %17, %17 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %16, %16 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %18, %18 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        // bank ^ 2: banks 0,1 keep the first, banks 2,3 the second      -> t5 = v0..v3, t6 = v4..v7 by bank, t7 = v8|v9
        "v_add_f32_dpp %6, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        // lane ^ 1: even lanes keep t5, odd lanes t6; t7 is plainly summed
        "v_cndmask_b32 %1, %6, %7, %19\n\t"                          // keep = odd ? t6 : t5
        "v_cndmask_b32 %2, %7, %6, %19\n\t"                          // send = odd ? t5 : t6
        "v_add_f32_dpp %3, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %4, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        // lane ^ 2: lanes 0,1 of a quad keep that, lanes 2,3 the v8|v9 register
        "v_cndmask_b32 %5, %4, %3, %20\n\t"                          // keep = hi ? t3 : t4
        "v_cndmask_b32 %1, %3, %4, %20\n\t"                          // send = hi ? t4 : t3
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(y), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(v8), "v"(v9), "s"(odd), "s"(hi));
    return y;
}

struct BwdLds {
    float4 a[kWave];                              // x, y, ca', 2 cb'      (conic pre-multiplied by 0.5 log2 e)
    float4 b[kWave];                              // cc', opacity, Gaussian id (bits), -
    float4 c[kWave];                              // r, g, b, depth
    __attribute__((aligned(16))) float w[kSlots * kWStride];
};

// Per-pixel state a walk starts from: the pixel's upstream gradients, the index behind its last contributor, and the
// transmittance / "colour behind" at the END of the range that is walked.
struct BwdPixel {
    float T, behind, dC0, dC1, dC2, dD, dA;
    uint32_t last;
};

// Walks list entries [first, end) of `tile` for quadrant `quad` back to front (first is a multiple of 64) and adds the
// per-Gaussian sums to dsplats.  Everything is wave-uniform except the pixel state.
// id_top: this lane's list entry of the TOP chunk, point_list[list_begin + min(top_base + 63 - lane, end - 1)] — loaded by
// the caller together with the pixel state, so that the walk starts one memory round trip earlier.
__device__ __forceinline__ uint32_t top_chunk_index(int end) {
    return (uint32_t)min(((end - 1) / kWave) * kWave + (kWave - 1 - (int)threadIdx.x), end - 1);
}
// Returns the number of trips (list entries that passed the quadrant's cull): the wave's work.
__device__ __forceinline__ int backward_walk(BwdLds& L, const FrameDev& f, int tile, int quad, int first, int end,
                                              BwdPixel px, uint32_t list_begin, uint32_t id_top,
                                              const uint32_t* __restrict__ point_list,
                                              const float4* __restrict__ splats, float* __restrict__ dsplats) {
    const int lane = threadIdx.x;
    const int tile_x = tile % f.gx, tile_y = tile / f.gx;
    const int qx0 = tile_x * kTile + (quad & 1) * 8, qy0 = tile_y * kTile + (quad >> 1) * 8;
    const float pxf = (float)(qx0 + (lane & 7)), pyf = (float)(qy0 + (lane >> 3));
    float T = px.T, behind = px.behind;
    const float dC0 = px.dC0, dC1 = px.dC1, dC2 = px.dC2, dD = px.dD, dA = px.dA;
    const uint32_t last = px.last;

    // ---- the transposed role of this lane: splat row r, pixel group g -> pixels g + 16 i (i = 0..3): column g & 7, rows
    // (g >> 3) + 2 i.  Their upstream gradients are fetched once (they live in those pixels' lanes: through LDS).
    const int row = lane >> 4, grp = lane & 15;
    float4 fc[4];
    {
        float4* tmp = reinterpret_cast<float4*>(L.w);
        __syncthreads();                                     // (an earlier walk of this wave may still read the weight rows)
        tmp[lane] = make_float4(dC0, dC1, dC2, dD);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) fc[i] = tmp[grp + 16 * i];
        __syncthreads();
    }
    const float gx_pix = (float)(qx0 + (grp & 7)), gy_pix = (float)(qy0 + (grp >> 3));
    float* w_store = L.w + 2 * lane;
#ifdef SCG_FWD_TRIP_CXX
    const float* w_load = L.w + row * kWStride + 2 * grp;
#else
    // (ONE register holds the address of the lane's transposed view: the hand-written block's LDS offset; a second copy as a
    //  pointer for the partial flush behind the loop was a register more)
    const uint32_t lds_wl = (uint32_t)reinterpret_cast<uintptr_t>(L.w + row * kWStride + 2 * grp);
    typedef const float __attribute__((address_space(3))) * LdsFloats;
#define w_load reinterpret_cast<LdsFloats>((uintptr_t)lds_wl)
#endif
    // which of the ten row sums this lane owns after row_reduce10, and where it goes in the 16-float record
    // (64 bytes, line aligned: one memory-side atomic transaction per record and flush)
    const int bank = (lane >> 2) & 3, qq_ = lane & 3;
    int out_slot = -1;
    if (qq_ < 2 || (qq_ == 2 && bank < 2)) {
        const int quantity = (qq_ == 0) ? bank : (qq_ == 1) ? 4 + bank : 8 + bank;    // order of row_reduce10's arguments
        out_slot = (quantity < 7) ? quantity : quantity + 1;
    }
    const uint32_t out_bytes = (uint32_t)(out_slot < 0 ? 0 : out_slot) * 4u;

    int slot = 0;                          // rows of the open block in use
    // centre and id of the splat in this lane's row of the open block (picked up when the splat is blended: the values
    // are in every lane then, the lanes of row `slot` keep them)
    float my_x = 0.f, my_y = 0.f;
    uint32_t my_id = 0;

    // the lanes that own one of the ten sums after row_reduce10, all four rows; the full EXEC mask
    const uint64_t out_lanes = __ballot(out_slot >= 0);
    const uint64_t lanes_all = __builtin_amdgcn_read_exec();
    uint64_t odd_lanes = 0xAAAAAAAAAAAAAAAAull, hi_lanes = 0xCCCCCCCCCCCCCCCCull;
    asm volatile("" : "+s"(odd_lanes), "+s"(hi_lanes));
    auto flush = [&](int rows, float dx, float dy0) {
        // the (q, w) stores of the trips and the transposed reads below are different lanes' views of one LDS block: the
        // wave's LDS operations execute in order; the fence keeps the compiler from reordering them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (the first pixel starts the sums: no 0 + x, no fma(.., 0) — two instructions per flush)
#ifdef SCG_FWD_TRIP_CXX
        const float2 qw0 = *reinterpret_cast<const float2*>(w_load);
#else
        const float2 qw0 = make_float2(w_load[0], w_load[1]);
#endif
        const float qy0 = qw0.x * dy0;
        float Sq = qw0.x, Sy = qy0, Syy = qy0 * dy0, Rr = qw0.y * fc[0].x, Gg = qw0.y * fc[0].y, Bb = qw0.y * fc[0].z,
              Dz = qw0.y * fc[0].w;
#pragma unroll
        for (int i = 1; i < 4; ++i) {
#ifdef SCG_FWD_TRIP_CXX
            const float2 qw = *reinterpret_cast<const float2*>(w_load + 32 * i);
#else
            const float2 qw = make_float2(w_load[32 * i], w_load[32 * i + 1]);
#endif
            const float dy = dy0 - (float)(2 * i);
            const float qy = qw.x * dy;
            Sq += qw.x;
            Sy += qy;
            Syy = __builtin_fmaf(qy, dy, Syy);
            Rr = __builtin_fmaf(qw.y, fc[i].x, Rr);
            Gg = __builtin_fmaf(qw.y, fc[i].y, Gg);
            Bb = __builtin_fmaf(qw.y, fc[i].z, Bb);
            Dz = __builtin_fmaf(qw.y, fc[i].w, Dz);
        }
        const float Sx = dx * Sq, Sxx = dx * Sx, Sxy = dx * Sy;
        const float sum = row_reduce10(Sx, Sy, Dz, Sq,              // sum q dx, sum q dy, ddepth, sum q
                                       Sxx, Sxy, Syy, Rr,           // second moments, dr
                                       Gg, Bb, odd_lanes, hi_lanes); // dg db
        // ONE atomic instruction: the owning lanes of the first `rows` rows, a 64-byte line per record.  The EXEC mask is set by
        // hand: the compiler's `if` costs s_and_saveexec + two branches + s_or per flush.  Fire-and-forget (never waited for).
        const uint32_t rec = my_id * (uint32_t)(SCG_DSPLAT_FLOATS * 4) + out_bytes;
        const uint64_t lanes = (rows >= kSlots) ? out_lanes : (out_lanes & ((1ull << (16 * rows)) - 1ull));
        asm volatile("s_mov_b64 exec, %[lanes]\n\t"
                     "global_atomic_add_f32 %[off], %[val], %[base]\n\t"
                     "s_mov_b64 exec, %[all]"
                     :: [lanes] "s"(lanes), [off] "v"(rec), [val] "v"(sum), [base] "s"(dsplats), [all] "s"(lanes_all)
                     : "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // Instruction diet of the walk (a scalar instruction costs a SIMD four cycles, twice a plain vector one — tools/probes/
    // ifetch_probe.hip — and a single wave issues one instruction of any kind per ~5.6 cycles):
    //   * lane l of a chunk stages list entry base + 63 - l: ascending bit order (s_ff1) IS back-to-front order, no 63 - clz;
    //   * "this entry lies behind the pixel's last contributor" is tested against last - base - 63, once per chunk;
    //   * the weight rows are addressed by a pointer that advances with the slot (no scalar multiply per trip);
    //   * the 0.99 clamp lives in a scalar register (v_med3_f32 takes no literal).
    float amax_s;
    asm volatile("s_mov_b32 %0, 0x3f7d70a4" : "=s"(amax_s));       // 0.99f
    const int chunk_bot = first / kWave;
    uint32_t id = id_top;
    int n_trips = 0;
#ifdef SCG_FWD_TRIP_CXX
    float* w_ptr = w_store;
#else
    // LDS offsets of the record planes and of this lane's (q, w) column (the low half of a generic LDS address is the offset),
    // the full EXEC mask, and the lanes of the four rows of the open block
    static_assert(offsetof(BwdLds, b) == 1024 && offsetof(BwdLds, c) == 2048 && kWStride * sizeof(float) == 512,
                  "the hand-written walk addresses the planes with immediate offsets");
    const uint32_t lds_rec = (uint32_t)reinterpret_cast<uintptr_t>(&L.a[0]);
    const uint32_t lds_w = (uint32_t)reinterpret_cast<uintptr_t>(w_store);
    const uint64_t row0 = 0x000000000000FFFFull, row1 = 0x00000000FFFF0000ull, row2 = 0x0000FFFF00000000ull,
                   row3 = 0xFFFF000000000000ull;
    // the upstream gradients of the lane's four pixels and of its own pixel in FIXED registers (the block below names them)
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const f32x16 fcv = {fc[0].x, fc[0].y, fc[0].z, fc[0].w, fc[1].x, fc[1].y, fc[1].z, fc[1].w,
                        fc[2].x, fc[2].y, fc[2].z, fc[2].w, fc[3].x, fc[3].y, fc[3].z, fc[3].w};
    const f32x4 dCv = {dC0, dC1, dC2, dD};
#endif
    for (int chunk = (end - 1) / kWave; chunk >= chunk_bot; --chunk) {
        const int base = chunk * kWave;
        const int k = base + (kWave - 1 - lane);
        // entry base + 63 - j is blended by this pixel iff it is below `last`:  j > base + 63 - last
#ifdef SCG_FWD_TRIP_CXX
        const int first_j = base + (kWave - 1) - (int)last;        // may be negative: then every j passes
#else
        const int first_j = max(base + (kWave - 1) - (int)last, -1);   // (-1: every j passes — and no lane passes j = -1)
#endif
        // this chunk's records (the id is valid for every lane: clamped), and — one round trip ahead — the next chunk's ids
        const float4 a = splats[3 * (size_t)id + 0];
        const float4 b = splats[3 * (size_t)id + 1];
        // (the third piece of the record rides along: fetched inside `if (hit)` it was a second dependent round trip per chunk —
        //  round 6, same process: S2 100.3 -> 99.9 us, S3 199.0 -> 198.0, S4 76.3 -> 74.9)
        const float4 c_rec = splats[3 * (size_t)id + 2];
        uint32_t id_next = id;
        if (chunk > chunk_bot) id_next = point_list[list_begin + (uint32_t)(k - kWave)];
        const bool hit = (k < end) && splat_hits_rect(a, b, (float)qx0, (float)qy0);
        if (hit) {
            L.a[lane] = make_float4(a.x, a.y, kHalfLog2e * a.z, 2.0f * kHalfLog2e * a.w);   // (2 cb': five-instruction quadratic form below)
            L.b[lane] = make_float4(kHalfLog2e * b.x, b.y, __builtin_bit_cast(float, id), 0.f);
            L.c[lane] = c_rec;
        }
        id = id_next;
        uint64_t m = __ballot(hit);
        n_trips += __builtin_popcountll(m);                          // (scalar, once per chunk)
        // every gather has landed before the loop (vmcnt(0)): the only VMEM traffic inside it are fire-and-forget
        // atomics, which must never be waited for
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();

#ifdef SCG_FWD_TRIP_CXX
        // The walk as the compiler writes it (libscg_raster_cxx.so, scgaussian_amd/build.py): 13 scalar / branch instructions per
        // trip, four of them for the slot counter of the open block; the same operations in the same order as the hand-written
        // walk below, which tests/test_gpu_parity.py holds against this one.
        while (m) {
            const int j = __builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));
            const float4 a = L.a[j];
            const float4 b = L.b[j];
            const float4 c = L.c[j];                                // with a and b: one LDS round trip per trip, not two
            asm("" ::"v"(b.w));                                     // keep it one ds_read_b128 (a b96 costs twice the LDS cycles)
            const float dx = a.x - pxf, dy = a.y - pyf;
            // t = ca' dx^2 + 2 cb' dx dy + cc' dy^2 in five plain instructions (u = ca' dx + 2cb' dy; t = u dx + (cc' dy) dy): the
            // per-splat sums are formed from (q, w) later, so e = ca' dx + cb' dy and h are not needed here.  The empty asm keeps
            // the SLP vectoriser from pairing the two products (packed: two register copies + a hazard nop per trip).
            float u = a.z * dx;
            asm("" : "+v"(u));
            u = __builtin_fmaf(a.w, dy, u);
            const float t = __builtin_fmaf(u, dx, (b.x * dy) * dy);     // -log2 G
            const float oG = b.y * __builtin_amdgcn_exp2f(-t);
            const bool ok = (j > first_j) && (t >= 0.0f) && (oG >= kAlphaMin);
            if (__ballot(ok) == 0ull) continue;                     // wave-uniform

            const float q0 = ok ? oG : 0.0f;                        // alpha before the 0.99 clamp, 0 if skipped
            const float alpha = __builtin_amdgcn_fmed3f(q0, 0.0f, amax_s);
            const float one_m = 1.0f - alpha;                       // >= 0.01
            T *= __builtin_amdgcn_rcpf(one_m);                      // transmittance in front of this splat
            const float d = __builtin_fmaf(c.x, dC0, __builtin_fmaf(c.y, dC1, __builtin_fmaf(c.z, dC2, __builtin_fmaf(c.w, dD, dA))));
            const float q = q0 * ((d - behind) * T);                // opacity * G * dL/dalpha
            // B_{i-1} = (1 - alpha) B_i + alpha d_i: the product in place, then the fused add in place — written as
            // fma(one_m, behind, alpha * d) the compiler needs a register copy per trip (round 4: 118.7 -> 116.9 us at S2)
            behind *= one_m;
            asm("" : "+v"(behind));
            behind = __builtin_fmaf(alpha, d, behind);
            const float wgt = alpha * T;
            *reinterpret_cast<float2*>(w_ptr) = make_float2(q, wgt);
            w_ptr += kWStride;
            const bool mine = (row == slot);
            my_x = mine ? a.x : my_x;
            my_y = mine ? a.y : my_y;
            my_id = mine ? __builtin_bit_cast(uint32_t, b.z) : my_id;
            if (++slot == kSlots) {
                flush(kSlots, my_x - gx_pix, my_y - gy_pix);
                slot = 0;
                w_ptr = w_store;
            }
        }
#else
        // The walk, hand-written (round 4).  What the compiler cannot do here (an unrolled slot needs `goto` into loop bodies,
        // which its structuriser turns into a state machine of scalar moves — profiles/README.md): FOUR COPIES OF THE TRIP, one
        // per row of the open block, entered at the row the previous chunk stopped in.  A copy loops until a splat is blended
        // (then falls through to the next row) or the chunk is exhausted, so there is no slot counter to advance, compare and
        // branch on, the row's (q, w) store has an immediate offset and "the lanes of this row keep the splat" is three moves
        // under a constant EXEC mask (measured against three selects on that mask: 110.8 vs 111.5 us).  The three tests narrow EXEC themselves (v_cmpx) and ONE branch asks whether anybody is left; the update
        // runs on the blending lanes only (the others hand zeros to the row sums: the two v_mov in front of the tests).
        //   v[48:51] x, y, ca' -> u -> alpha, 2cb' -> G -> opacity G | v[52:55] cc' -> t -> 1-alpha, opacity, id, - |
        //   v[44:47] r g b depth | v56 dx -> 1/(1-alpha) | v57 dy -> d | v[58:59] q, w | v60 LDS address
        //   (a transcendental's result is first read two instructions later: gfx950's forwarding hazard)
#define SCG_BWD_ROW(K, NEXT)                                                                                                 \
            ".Ltrip" #K "_%=:\n\t"                                                                                         \
            "s_ff1_i32_b64 s90, %[m]\n\t"                          /* lowest set bit = next entry back to front; -1: none */ \
            "s_bitset0_b64 %[m], s90\n\t"                                                                                 \
            "v_lshl_add_u32 v60, s90, 4, %[rec]\n\t"                                                                      \
            "ds_read_b128 v[48:51], v60\n\t"                                                                               \
            "ds_read_b128 v[52:55], v60 offset:1024\n\t"                                                                   \
            "ds_read_b128 v[44:47], v60 offset:2048\n\t"                                                                   \
            "v_mov_b32_e32 v58, 0\n\t"                                                                                     \
            "v_mov_b32_e32 v59, 0\n\t"                                                                                     \
            "v_cmpx_gt_i32_e32 vcc, s90, %[fj]\n\t"                /* EXEC: the entry lies in front of the pixel's last */ \
            "s_waitcnt lgkmcnt(2)\n\t"                                                                                     \
            "v_sub_f32_e32 v56, v48, %[px]\n\t"                                                                            \
            "v_sub_f32_e32 v57, v49, %[py]\n\t"                                                                            \
            "v_mul_f32_e32 v50, v56, v50\n\t"                       /* ca' dx */                                           \
            "s_waitcnt lgkmcnt(1)\n\t"                                                                                     \
            "v_mul_f32_e32 v52, v57, v52\n\t"                       /* cc' dy */                                           \
            "v_fmac_f32_e32 v50, v51, v57\n\t"                      /* u = ca' dx + 2 cb' dy */                            \
            "v_mul_f32_e32 v52, v57, v52\n\t"                       /* cc' dy^2 */                                         \
            "v_fmac_f32_e32 v52, v50, v56\n\t"                      /* t = -log2 G */                                      \
            "v_exp_f32_e64 v51, -v52\n\t"                                                                                  \
            "v_cmpx_le_f32_e32 vcc, 0, v52\n\t"                     /* EXEC: ... and t >= 0 */                             \
            "v_mul_f32_e32 v51, v53, v51\n\t"                       /* opacity G */                                        \
            "v_cmpx_le_f32_e32 vcc, 0x3b808081, v51\n\t"               /* EXEC: ... and opacity G >= 1/255 */                 \
            "s_cbranch_execz .Lnone" #K "_%=\n\t"                                                                          \
            "v_min_f32_e32 v50, 0x3f7d70a4, v51\n\t"                /* alpha = min(0.99, opacity G) */                     \
            "v_sub_f32_e32 v52, 1.0, v50\n\t"                       /* 1 - alpha >= 0.01 */                                \
            "v_rcp_f32_e32 v56, v52\n\t"                                                                                   \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
            "v_fma_f32 v57, v47, v27, %[dA]\n\t"                  /* d = c . dL/dC (depth and alpha channels folded in) */ \
            "v_fmac_f32_e32 v57, v46, v26\n\t"                                                                          \
            "v_fmac_f32_e32 v57, v45, v25\n\t"                                                                          \
            "v_fmac_f32_e32 v57, v44, v24\n\t"                                                                          \
            "v_mul_f32_e32 %[T], %[T], v56\n\t"                     /* transmittance in front of this splat */             \
            "v_sub_f32_e32 v58, v57, %[bh]\n\t"                                                                            \
            "v_mul_f32_e32 v58, v58, %[T]\n\t"                                                                             \
            "v_mul_f32_e32 v58, v51, v58\n\t"                       /* q = opacity G dL/dalpha */                          \
            "v_mul_f32_e32 %[bh], %[bh], v52\n\t"                   /* B_{i-1} = (1 - alpha) B_i + alpha d_i */            \
            "v_fmac_f32_e32 %[bh], v50, v57\n\t"                                                                           \
            "v_mul_f32_e32 v59, v50, %[T]\n\t"                      /* w = alpha T */                                      \
            "s_mov_b64 exec, %[row" #K "]\n\t"                        /* the lanes of row K keep centre and id ... */       \
            "v_sub_f32_e32 %[mdx], v48, %[px]\n\t"                    /* (offsets from the lane's pixel column — grp & 7 == lane & 7: its own pixel's — / first row) */ \
            "v_sub_f32_e32 %[mdy], v49, %[gy]\n\t"                                                                        \
            "v_mov_b32_e32 %[mi], v54\n\t"                                                                                 \
            "s_mov_b64 exec, -1\n\t"                            /* ... and every lane parks its (q, w) */              \
            "ds_write_b64 %[wst], v[58:59] offset:" NEXT "\n\t"
        // nobody blends the entry: next entry of the same row — or, if there was no entry (the bit scan of an empty mask says
        // -1, no lane passes the first test: first_j >= -1), the chunk is exhausted with K rows of the block open
#define SCG_BWD_NONE(K)                                                                                                      \
            ".Lnone" #K "_%=:\n\t"                                                                                         \
            "s_mov_b64 exec, -1\n\t"                                                                                   \
            "s_cmp_lt_i32 s90, 0\n\t"                                                                                     \
            "s_cbranch_scc0 .Ltrip" #K "_%=\n\t"                                                                           \
            "s_mov_b32 %[slot], " #K "\n\t"                                                                                \
            "s_branch .Lend_%=\n\t"
        {
            // (everything the block names by number is pinned by its constraint: v[24:27] = dL/dC of the lane's pixel,
            //  v[28:43] = dL/dC of the four pixels of its transposed role; v44-v61 and s90 are scratch)
            asm volatile(
                // enter at the row the open block has reached
                "s_cmp_eq_u32 %[slot], 0\n\t"
                "s_cbranch_scc1 .Ltrip0_%=\n\t"
                "s_cmp_eq_u32 %[slot], 1\n\t"
                "s_cbranch_scc1 .Ltrip1_%=\n\t"
                "s_cmp_eq_u32 %[slot], 2\n\t"
                "s_cbranch_scc1 .Ltrip2_%=\n\t"
                "s_branch .Ltrip3_%=\n\t"
                SCG_BWD_NONE(0)
                SCG_BWD_NONE(1)
                SCG_BWD_NONE(2)
                SCG_BWD_NONE(3)
                SCG_BWD_ROW(0, "0")
                SCG_BWD_ROW(1, "512")
                SCG_BWD_ROW(2, "1024")
                SCG_BWD_ROW(3, "1536")
                // ---- the block is full: the flush.  Lane (row r, pixel group g) reads the (q, w) of its four pixels of splat r
                // (one pixel column: dx is the lane's constant, the x-moments follow from the y-sums), forms the ten sums, the 16
                // lanes of a row are reduced by the transposing butterfly (see row_reduce10: the same 23 instructions), ONE atomic
                // instruction adds the 4 x 10 sums to the four records.  The same operations in the same order as flush() above.
                "ds_read2_b64 v[44:47], %[wld] offset1:16\n\t"            // q0 w0 q1 w1
                "ds_read2_b64 v[48:51], %[wld] offset0:32 offset1:48\n\t" // q2 w2 q3 w3
                "v_add_f32_e32 v52, -2.0, %[mdy]\n\t"                     // dy of pixels 1..3 (two rows down each)
                "v_add_f32_e32 v53, -4.0, %[mdy]\n\t"
                "v_add_f32_e32 v54, 0xc0c00000, %[mdy]\n\t"
                "s_waitcnt lgkmcnt(1)\n\t"
                "v_mul_f32_e32 v55, v44, %[mdy]\n\t"                      // Sy = q0 dy0          (Sq = v44 in place)
                "v_mul_f32_e32 v57, v45, v28\n\t"                         // dr = w0 dL/dr(p0)
                "v_mul_f32_e32 v58, v45, v29\n\t"
                "v_mul_f32_e32 v59, v45, v30\n\t"
                "v_mul_f32_e32 v60, v45, v31\n\t"                         // ddepth
                "v_mul_f32_e32 v56, v55, %[mdy]\n\t"                      // Syy = q0 dy0^2
                "v_mul_f32_e32 v61, v46, v52\n\t"                         // q1 dy1
                "v_add_f32_e32 v44, v44, v46\n\t"
                "v_fmac_f32_e32 v57, v47, v32\n\t"
                "v_fmac_f32_e32 v58, v47, v33\n\t"
                "v_fmac_f32_e32 v59, v47, v34\n\t"
                "v_fmac_f32_e32 v60, v47, v35\n\t"
                "v_add_f32_e32 v55, v55, v61\n\t"
                "v_fmac_f32_e32 v56, v61, v52\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mul_f32_e32 v61, v48, v53\n\t"                         // q2 dy2
                "v_add_f32_e32 v44, v44, v48\n\t"
                "v_fmac_f32_e32 v57, v49, v36\n\t"
                "v_fmac_f32_e32 v58, v49, v37\n\t"
                "v_fmac_f32_e32 v59, v49, v38\n\t"
                "v_fmac_f32_e32 v60, v49, v39\n\t"
                "v_add_f32_e32 v55, v55, v61\n\t"
                "v_fmac_f32_e32 v56, v61, v53\n\t"
                "v_mul_f32_e32 v61, v50, v54\n\t"                         // q3 dy3
                "v_add_f32_e32 v44, v44, v50\n\t"
                "v_fmac_f32_e32 v57, v51, v40\n\t"
                "v_fmac_f32_e32 v58, v51, v41\n\t"
                "v_fmac_f32_e32 v59, v51, v42\n\t"
                "v_fmac_f32_e32 v60, v51, v43\n\t"
                "v_add_f32_e32 v55, v55, v61\n\t"
                "v_fmac_f32_e32 v56, v61, v54\n\t"
                "v_mul_f32_e32 v45, %[mdx], v44\n\t"                      // Sx  = dx Sq
                "v_mul_f32_e32 v47, %[mdx], v55\n\t"                      // Sxy = dx Sy
                "v_mul_f32_e32 v46, %[mdx], v45\n\t"                      // Sxx = dx Sx
                // row_reduce10(Sx v45, Sy v55, ddepth v60, Sq v44 | Sxx v46, Sxy v47, Syy v56, dr v57 | dg v58, db v59)
                "s_nop 1\n\t"
                "v_add_f32_dpp v48, v45, v45 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp v49, v60, v60 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp v50, v46, v46 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp v51, v56, v56 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp v52, v58, v58 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp v48, v55, v55 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp v49, v44, v44 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp v50, v47, v47 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp v51, v57, v57 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp v52, v59, v59 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp v53, v48, v48 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_add_f32_dpp v54, v50, v50 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_add_f32_dpp v53, v49, v49 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                "v_add_f32_dpp v54, v51, v51 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                "v_add_f32_dpp v61, v52, v52 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_e64 v48, v53, v54, %[odd]\n\t"
                "v_cndmask_b32_e64 v49, v54, v53, %[odd]\n\t"
                "v_add_f32_dpp v50, v61, v61 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 0\n\t"
                "v_add_f32_dpp v51, v49, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_e64 v52, v51, v50, %[hi]\n\t"
                "v_cndmask_b32_e64 v48, v50, v51, %[hi]\n\t"
                "v_lshl_add_u32 v44, %[mi], 6, %[ob]\n\t"                 // the record's line + this lane's slot in it
                "s_nop 0\n\t"
                "v_add_f32_dpp v45, v48, v52 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b64 exec, %[lanes]\n\t"
                "global_atomic_add_f32 v44, v45, %[base]\n\t"             // fire-and-forget
                "s_mov_b64 exec, -1\n\t"
                "s_branch .Ltrip0_%=\n\t"
                // (the exhausted trip leaves LDS reads into v44-v61 in flight: they must have landed before the compiler's code,
                //  which may reuse those registers and does not see memory operations inside this block, continues)
                ".Lend_%=:\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [m] "+s"(m), [slot] "+s"(slot), [T] "+v"(T), [bh] "+v"(behind), [mdx] "+v"(my_x), [mdy] "+v"(my_y),
                  [mi] "+v"(my_id)
                : [rec] "v"(lds_rec), [wst] "v"(lds_w), [wld] "v"(lds_wl), [fj] "v"(first_j), [px] "v"(pxf), [py] "v"(pyf),
                  [gy] "v"(gy_pix), [ob] "v"(out_bytes), [dA] "v"(dA), "{v[24:27]}"(dCv), "{v[28:43]}"(fcv),
                  [row0] "s"(row0), [row1] "s"(row1), [row2] "s"(row2), [row3] "s"(row3), [lanes] "s"(out_lanes),
                  [odd] "s"(odd_lanes), [hi] "s"(hi_lanes), [base] "s"(dsplats)
                : "memory", "vcc", "scc", "s90", "s91", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
                  "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61");
        }
#undef SCG_BWD_NONE
#undef SCG_BWD_ROW
#endif
        __syncthreads();                                            // the staged records are overwritten next
    }
#ifdef SCG_FWD_TRIP_CXX
    if (slot > 0) flush(slot, my_x - gx_pix, my_y - gy_pix);
#else
    if (slot > 0) flush(slot, my_x, my_y);                  // (the hand-written walk keeps the offsets from the lane's pixel column / row)
#undef w_load
#endif
    return n_trips;
}

// The pixel's upstream gradients and forward state, for the walk that ends at its LAST contributor.
__device__ __forceinline__ BwdPixel load_pixel_final(const FrameDev& f, int px, int py, const float* __restrict__ final_T,
                                                     const uint32_t* __restrict__ n_contrib,
                                                     const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                                                     const float* __restrict__ dL_dalpha) {
    BwdPixel s;
    s.T = 1.0f; s.dC0 = s.dC1 = s.dC2 = s.dD = s.dA = 0.f; s.last = 0;
    if ((px < f.W) && (py < f.H)) {
        const size_t pix = (size_t)py * f.W + px;
        const size_t hw = (size_t)f.H * f.W;
        s.T = final_T[pix];
        s.last = n_contrib[pix];
        s.dC0 = dL_dcolor[pix]; s.dC1 = dL_dcolor[hw + pix]; s.dC2 = dL_dcolor[2 * hw + pix];
        if (dL_ddepth) s.dD = dL_ddepth[pix];
        if (dL_dalpha) s.dA = dL_dalpha[pix];
    }
    s.behind = f.bg[0] * s.dC0 + f.bg[1] * s.dC1 + f.bg[2] * s.dC2;      // B_last
    return s;
}

// Workgroup (tile, quadrant) in the binning stage's launch order.  SIX waves per SIMD (73 registers: the hand-written walk pins
// 38 of them) since round 4's flush moved into the block — found in round 5 by the list-scheduler replay of the kernel's own
// wave clocks, which matched the forward on its slot count and missed the backward by 20 % on the 7 168 slots it was believed
// to have (tools/probes/backward_timeline.py counts 6 144 waves alive).  A seventh costs nothing to get (-DSCG_BWD_WAVES=7:
// 72 registers, no spill) and buys nothing: S2 99.7 vs 99.9 us, S4 75.8 vs 75.7 (profiles/r05_ab_backward_7_waves.txt) — the
// waves wait for issue slots, not for each other's latencies.
#ifdef SCG_BWD_WAVES
#define SCG_BWD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SCG_BWD_WAVES, SCG_BWD_WAVES)))
#else
#define SCG_BWD_OCCUPANCY
#endif
__global__ __launch_bounds__(kWave) SCG_BWD_OCCUPANCY void blend_backward_kernel(
    FrameDev f, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ splats, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
    float* __restrict__ dsplats) {
    __shared__ BwdLds L;
    const int n_tiles = f.gx * f.gy;
    // (tile, quadrant) from the backward's own launch order (behind the tiles' order): band = XCD = blockIdx & 7, entry
    // blockIdx >> 3 of the band's run — the quadrants that took longest in the camera's previous backward first
    const int slots = tile_order_slots(n_tiles);
    const uint32_t* order_q = reinterpret_cast<const uint32_t*>(ranges) + 2 * (size_t)n_tiles + slots;
    const uint32_t entry = order_q[(size_t)(blockIdx.x & 7) * (4 * (slots >> 3)) + (blockIdx.x >> 3)];
    const int tile = (int)(entry >> 2), quad = (int)(entry & 3u);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x;
    const int px = (tile % f.gx) * kTile + (quad & 1) * 8 + (lane & 7), py = (tile / f.gx) * kTile + (quad >> 1) * 8 + (lane >> 3);
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const BwdPixel s = load_pixel_final(f, px, py, final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha);
    // highest list index any pixel of the quadrant blended
    uint32_t mx = s.last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down((int)mx, off, kWave));
    const int limit = min(n, __builtin_amdgcn_readfirstlane((int)mx));
#ifdef SCG_PROBE_TIMELINE                // tools/probes/backward_timeline.py
    const uint32_t tp0 = (uint32_t)wall_clock64();
#endif
    if (limit <= 0) {
        if (f.bcost_out && lane == 0) f.bcost_out[4 * tile + quad] = 0u;
        return;
    }
    const int trips = backward_walk(L, f, tile, quad, 0, limit, s, range.x, point_list[range.x + top_chunk_index(limit)],
                                    point_list, splats, dsplats);
    // this quadrant's work: the next step's launch order of this camera (ScgFrame.bwd_cost_out).  (Its TRIPS, not its time: a
    // wave's time depends on when it ran — round-5 timeline: waves ordered by their previous time came out uncorrelated.)
    // (trips alone: with the entries walked mixed in — trips + entries / 4 — the same or worse, S2 101.1 vs 100.5 us)
    if (f.bcost_out && lane == 0) f.bcost_out[4 * tile + quad] = (uint32_t)trips;
#ifdef SCG_PROBE_TIMELINE
    if (f.cost_out && lane == 0) {
        uint32_t* tl = f.cost_out + n_tiles + 65792 + 32768 + (size_t)(n_tiles + 8) * 40 + (size_t)blockIdx.x * 4;
        tl[0] = tp0; tl[1] = (uint32_t)wall_clock64(); tl[2] = ((uint32_t)min(trips, 65535) << 16) | (uint32_t)min(limit, 65535);
        tl[3] = 0xBAC00000u | (uint32_t)(4 * tile + quad);
    }
#endif
}

int launch_blend_backward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                          const float* splats, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                          float* dsplats, bool dsplats_prezeroed, hipStream_t stream, hipEvent_t started, hipEvent_t done) {
    if (!dsplats_prezeroed) {
        const int rc = check_hip(hipMemsetAsync(dsplats, 0, (size_t)f.P * SCG_DSPLAT_FLOATS * sizeof(float), stream),
                                 "dsplats memset");
        if (rc) return rc;
    }
    const int n_tiles = f.gx * f.gy;
    const int grid = ((n_tiles + 7) / 8) * 8 * 4;          // (tile, quadrant) workgroups of one wave
    if (started || done) {
        // the stage timer's events ride on this dispatch: no barrier packets around the dominant kernel (scg_common.h)
        hipExtLaunchKernelGGL(blend_backward_kernel, dim3(grid), dim3(kWave), 0u, stream, started, done, 0u, f,
                              reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                              final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dsplats);
    } else {
        hipLaunchKernelGGL(blend_backward_kernel, dim3(grid), dim3(kWave), 0, stream, f,
                           reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(splats),
                           final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dsplats);
    }
    return check_hip(hipGetLastError(), "blend_backward_kernel");
}

}  // namespace scg
