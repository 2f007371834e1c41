// binning_tiles.hip — tile-first binning (the default binning path).
//
// The reference pipeline duplicates every Gaussian once per touched tile and sorts the R = sum(tiles touched)
// 12-byte (tile<<32|depth, id) pairs with a 6-pass global radix sort: ~150*R bytes of HBM traffic and 18+ kernel
// launches.  The result that matters — for every tile its Gaussians ordered by (depth, id), plus the tile ranges
// — is produced here with five launches and ~12*R bytes (three on the one-call path scg_forward, where the geometry kernel
// builds the histogram — geometry.hip, geometry_hist_kernel — and the forward blend sorts its own tiles — blend.hip):
//
//   tile_hist_kernel      every workgroup owns a slice of the Gaussians, walks their tile rectangles (generated on
//                         the fly, never materialised) and histograms them over the Tn tiles in LDS -> table[B][Tn]
//   table_colscan_kernel  per tile: exclusive prefix over the workgroups (in place) + tile total (+ 64-tile sums)
//   tile_scatter_kernel   workgroup (band of tile rows, Gaussian slice): scans the tile totals of its band (tile starts),
//                         walks the slice's rectangles again and drops each instance's Gaussian id into its tile's segment
//                         (slot = segment start + slice prefix + LDS cursor); order inside a segment is arbitrary.  One
//                         band = one XCD's L2: full-line write-backs instead of one 32-byte sector per 4-byte store.  Its
//                         first eight workgroups publish the tile starts = the tile RANGES (identifyTileRanges for
//                         free), the work lists of the rarer sort sizes and the launch order of the blend kernels
//   tile_sort_kernel      one workgroup per tile: one-pass bucket sort of the segment by depth in LDS (4-pass LSD radix
//                         for heavily tied depths; ties come out in ascending id): (depth, id) is a total order, so the
//                         arbitrary scatter order cannot show and the output is bit-identical to the reference's stable
//                         sort.  <4 waves, 2048 entries> normally, <8 waves, 4096 entries> for dense scenes.
//   tile_sort_rare_kernel the tiles whose list is longer (work lists): 16-wave LDS bucket/radix sort up to 8192 entries,
//                         beyond that the same bucket sort with the (depth, id) composites in global scratch; the
//                         bitonic network (LDS up to 16384 entries, else global) only as the fallback for tied depths.
//
// LDS does the work a global sort would do through HBM: a tile's list (a few hundred to a few thousand entries)
// fits the 160 KiB LDS of a CU with room to spare.
#include "scg_common.h"
#include "tile_sort.h"
#include "tile_walk.h"

#include <mutex>

namespace scg {

constexpr int kTileBlocksSmall = 128;          // slices (= 16-wave workgroups of the histogram pass) of a small frame
constexpr int kTileBlocksLarge = 256;          // ... of a frame of a million instances or more: one per compute unit — the
                                               // geometry kernel that histograms its own rectangles (117 registers) fits
                                               // one 16-wave workgroup per compute unit, a second round would double it
constexpr int64_t kLargeFrame = 1ll << 20;
constexpr int kLargeScene = 100000;           // ... or of 100 000 Gaussians or more (the geometry is the longer part there)
constexpr int kSortSmallMax = 2048;            // entries sorted by tile_sort_kernel<4,..> (radix, 16 KiB of key/id LDS)
constexpr int kSortDenseMax = 4096;            // ... by its 8-wave variant, launched instead when the AVERAGE list is long
constexpr int kSortMidMax = 8192;              // entries the rare kernel's 16-wave LDS sort takes (96 KiB)
constexpr int kVeryLong = 16384;               // a frame with a list beyond this is worth the split + 8-wave path (counted for the
                                               // caller's next render of the camera: ScgFrame.long_lists_out[1])
constexpr int kSort8Max = 4096;                // entries the 8-wave work-list sort (tile_sort_list8_kernel) takes; longer lists are
                                               // SPLIT by depth into parts of at most this many entries (tile_split_long_kernel)
constexpr int kSortBigLdsMax = 16384;          // entries the bitonic fallback keeps in LDS (128 KiB)
constexpr int kMaxDynLds = 152 * 1024;         // dynamic LDS ceiling requested for the big-LDS kernels (static LDS
                                               // of the same kernel + this must stay <= 160 KiB)

// q = n / d, r = n % d for n < 2^24, 0 < d < 2^16 (one v_rcp_f32 + fix-up instead of the ~50-instruction uThe visiting order is unspecified: callers only count / allocate slots.
template <class F>
__device__ __forceinline__ void visit_instances(const uint2* __restrict__ rects, int grid_x, uint32_t ga, uint32_t gb,
                                                F&& f) {
    const int lane = lane_id();
    for (uint32_t g0 = ga; g0 < gb; g0 += kWave) {
        const uint32_t g = g0 + lane;
        uint2 r = make_uint2(0u, 0u);
        if (g < gb) r = rects[g];
        walk_rects(r, g, grid_x, f);
    }
}

// ---------------------------------------------------------------------------------------------------
// per-workgroup tile histogram
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBinThreads) void tile_hist_kernel(const uint2* __restrict__ rects, uint32_t P,
                                                                int grid_x, int n_tiles,
                                                                uint32_t* __restrict__ table,
                                                                uint32_t* __restrict__ class_counts,
                                                                uint32_t* __restrict__ len_hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    if (blockIdx.x == 0) {                                     // counters of the later kernels of this stage
        if (threadIdx.x < 4) class_counts[threadIdx.x] = 0u;         // mid tiles, big tiles, segments of split big lists, -
        len_hist[threadIdx.x] = 0u;                            // 8 x 64 length histogram + 8 x 64 cursors
    }
    for (int t = threadIdx.x; t < n_tiles; t += kBinThreads) hist[t] = 0;
    __syncthreads();
    uint32_t ga, gb;
    wave_slice(P, gridDim.x, blockIdx.x, wave_id(), ga, gb);
    visit_instances(rects, grid_x, ga, gb, [&](uint32_t tile, uint32_t) { atomicAdd(&hist[tile], 1u); });
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * n_tiles;
    for (int t = threadIdx.x; t < n_tiles; t += kBinThreads) row[t] = hist[t];
}

// Column scan of table[B][Tn]: per tile the exclusive prefix over workgroups (in place) and the tile total.
// Workgroup = 64 tiles x 16 row-groups (1024 threads).  A thread's <= 64 rows are loaded in ONE fully unrolled,
// predicated batch (all loads in flight together: the table was just written by other CUs, so every row is an
// L2 / HBM round trip) and stay in registers for the write pass.
constexpr int kColTiles = 64;
constexpr int kColGroups = kBinThreads / kColTiles;     // 16
constexpr int kColRowsMax = 16;                          // kTileBlocksLarge / kColGroups

constexpr int kBands8 = 8;                                // XCDs
constexpr int kLenClasses = 64;                         // per XCD band: tiles binned by list length >> shift

__device__ __forceinline__ int len_class(uint32_t total, int shift) {
    return (int)min((uint32_t)(kLenClasses - 1), total >> shift);
}
// Launch-order class of a tile.  Without a hint: its list length (the column scan knows it).  With the cost the blend
// forward recorded the last time this camera was rendered (ScgFrame.tile_cost_in: list entries its busiest quadrant
// blended): 16 classes per octave over 32..512 entries (about 5 % apart), everything longer in the top class — a
// wave's time follows the entries it blends, not the length of the list it stops early in (profiles/README.md: the
// length even correlates negatively on uniform scenes).
__device__ __forceinline__ int order_class(const uint32_t* __restrict__ cost_in, int t, uint32_t total, int shift) {
    if (!cost_in) return len_class(total, shift);
    if (total == 0u) return 0;
    const int c = (int)(__float_as_uint((float)cost_in[t]) >> 19) - ((127 + 5) << 4);
    return max(0, min(kLenClasses - 1, c));
}

__global__ __launch_bounds__(kBinThreads) void table_colscan_kernel(uint32_t* __restrict__ table, int nblocks,
                                                                    int n_tiles, uint32_t* __restrict__ tile_total,
                                                                    uint32_t* __restrict__ len_hist, int len_shift,
                                                                    const uint32_t* __restrict__ cost_in,
                                                                    uint8_t* __restrict__ tile_class,
                                                                    uint32_t* __restrict__ tile_part) {
    __shared__ uint32_t s_part[kColGroups][kColTiles];
    const int c = threadIdx.x & (kColTiles - 1);
    const int q = threadIdx.x / kColTiles;
    const int t = blockIdx.x * kColTiles + c;
    const int b0 = (int)((int64_t)nblocks * q / kColGroups), b1 = (int)((int64_t)nblocks * (q + 1) / kColGroups);
    uint32_t v[kColRowsMax];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kColRowsMax; ++k) {
        v[k] = 0;
        if (t < n_tiles && b0 + k < b1) v[k] = table[(size_t)(b0 + k) * n_tiles + t];
    }
#pragma unroll
    for (int k = 0; k < kColRowsMax; ++k) sum += v[k];
    s_part[q][c] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int k = 0; k < q; ++k) run += s_part[k][c];
    // length histogram of the workgroup's 64 tiles, aggregated in LDS: neighbouring tiles have similar lengths, so a
    // workgroup touches a handful of (band, class) counters — one global atomic each instead of one per tile
    __shared__ uint32_t s_len[kBands8 * kLenClasses];
    if (threadIdx.x < kBands8 * kLenClasses) s_len[threadIdx.x] = 0u;
    __syncthreads();
    if (t < n_tiles && q == kColGroups - 1) {
        tile_total[t] = run + sum;
        const int per = (n_tiles + 7) >> 3;
        // the class is decided HERE and stored: tile_start_kernel must place the tile in the class this histogram counted it
        // in, whatever happens to the hint buffer in between (it is caller memory)
        const int cls = order_class(cost_in, t, run + sum, len_shift);
        tile_class[t] = (uint8_t)cls;
        atomicAdd(&s_len[(t / per) * kLenClasses + cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kBands8 * kLenClasses && s_len[threadIdx.x]) atomicAdd(&len_hist[threadIdx.x], s_len[threadIdx.x]);
    // the sum of this workgroup's 64 tile totals: the scatter workgroups add these up instead of waiting for a scan of all tiles
    if (q == kColGroups - 1) s_part[0][c] = (t < n_tiles) ? run + sum : 0u;      // (row 0 was last read before the barrier above)
    __syncthreads();
    if (threadIdx.x < kColTiles) {
        uint32_t tot = s_part[0][threadIdx.x];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off, kWave);
        if (threadIdx.x == 0) tile_part[blockIdx.x] = tot;
    }
#pragma unroll
    for (int k = 0; k < kColRowsMax; ++k) {
        if (t < n_tiles && b0 + k < b1) table[(size_t)(b0 + k) * n_tiles + t] = run;
        run += v[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// tile starts / ranges (identifyTileRanges for free) and the scatter of ids into the tile segments: ONE launch
// ---------------------------------------------------------------------------------------------------
// A scan of the Tn tile totals used to be a launch of its own between the column scan and the scatter (5-6 us for
// microseconds of work).  Now every scatter workgroup scans the totals of ITS band of tile rows itself (a few hundred values;
// what lies before the band it adds up from the column scan's 64-tile sums), and the first eight workgroups of the launch
// publish what the later kernels need — tile starts, ranges, work lists of the rarer sort sizes, launch order of the blend
// kernels — one eighth of the tiles each, beside the scattering ones.
constexpr int kBands = 8;
constexpr int kQueue = 2 * kWave;
constexpr int kScatterThreads = 256;           // per (band, slice): small workgroups, so a wave sees enough Gaussians
constexpr int kScatterWaves = kScatterThreads / kWave;   // of its slice to fill its queue

// sum of tile_total[0 .. t_first) for the whole workgroup (256 threads; s_red: kScatterWaves words of LDS)
__device__ __forceinline__ uint32_t totals_before(const uint32_t* __restrict__ tile_total,
                                                  const uint32_t* __restrict__ tile_part, int t_first, uint32_t* s_red) {
    const int full = t_first / kColTiles;
    uint32_t v = 0;
    for (int k = threadIdx.x; k < full; k += kScatterThreads) v += tile_part[k];
    for (int k = full * kColTiles + (int)threadIdx.x; k < t_first; k += kScatterThreads) v += tile_total[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, kWave);
    __syncthreads();                                            // (s_red may still be read from an earlier use)
    if (lane_id() == 0) s_red[wave_id()] = v;
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScatterWaves; ++k) sum += s_red[k];
    return sum;
}

// exclusive prefix of v over the workgroup's 256 threads; total = sum over all of them (s_red as above)
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t* s_red, uint32_t& total) {
    const int lane = lane_id(), w = wave_id();
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t n = (uint32_t)__shfl_up((int)inc, off, kWave);
        if (lane >= off) inc += n;
    }
    __syncthreads();
    if (lane == kWave - 1) s_red[w] = inc;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < kScatterWaves; ++k) {
        const uint32_t sk = s_red[k];
        if (k < w) before += sk;
        total += sk;
    }
    return before + inc - v;
}

// Workgroup x of the first eight: tiles [x per, (x + 1) per) of the eighth of the image whose launch order it builds.
// capacity = number of entries point_list can hold.  The caller may pass an UPPER-BOUND GUESS instead of the exact
// instance count (to launch without waiting for the host read of num_rendered); if the guess is too small nothing is
// written past it and the published ranges are clipped to it, so every later kernel stays in bounds — the caller
// detects the overflow from num_rendered and runs the stage again.
__device__ __forceinline__ void publish_tile_starts(int x, int n_tiles, const uint32_t* __restrict__ tile_total,
                                                    const uint32_t* __restrict__ tile_part,
                                                    uint32_t* __restrict__ tile_start, uint2* __restrict__ ranges,
                                                    uint32_t capacity, uint32_t* __restrict__ class_counts,
                                                    uint32_t* __restrict__ mid_tiles, uint32_t* __restrict__ big_tiles,
                                                    uint32_t small_max, const uint32_t* __restrict__ len_hist,
                                                    const uint8_t* __restrict__ tile_class,
                                                    uint32_t* __restrict__ cost_out, const uint32_t* __restrict__ bcost_in,
                                                    uint8_t* q_class /* LDS, 4 * ceil(n_tiles / 8) bytes */,
                                                    uint32_t* __restrict__ nr_out) {
    __shared__ uint32_t s_red[kScatterWaves];
    __shared__ uint32_t s_first[kLenClasses];      // first slot of a length class in this band's run: longer classes first
    __shared__ uint32_t s_cnt[kLenClasses];
    const int per = (n_tiles + 7) >> 3;
    const int t0 = x * per, t1 = min(t0 + per, n_tiles);
    const int lane = lane_id();
    if (threadIdx.x < kLenClasses) {
        uint32_t before = 0;
        for (int k = (int)threadIdx.x + 1; k < kLenClasses; ++k) before += len_hist[x * kLenClasses + k];
        s_first[threadIdx.x] = before;
        s_cnt[threadIdx.x] = 0u;
    }
    uint32_t carry = totals_before(tile_total, tile_part, min(t0, n_tiles), s_red);       // (its barriers publish s_first / s_cnt)
    // launch order of the blend kernels (behind the ranges): tile t goes to its XCD band, longer lists first; the order
    // inside a length class is whatever the LDS atomics give (it changes scheduling only)
    uint32_t* order = reinterpret_cast<uint32_t*>(ranges) + 2 * (size_t)n_tiles;
    for (int k0 = t0; k0 < t1; k0 += kScatterThreads) {
        const int t = k0 + (int)threadIdx.x;
        const bool in = t < t1;
        const uint32_t cnt = in ? tile_total[t] : 0u;
        uint32_t total;
        const uint32_t run = carry + block_exclusive_256(cnt, s_red, total);
        carry += total;
        uint32_t len = 0;
        if (in) {
            tile_start[t] = run;
            const uint32_t lo = min(run, capacity), hi = min(run + cnt, capacity);
            ranges[t] = (hi > lo) ? make_uint2(lo, hi) : make_uint2(0u, 0u);
            len = hi - lo;
            if (t == n_tiles - 1) {
                tile_start[n_tiles] = run + cnt;
                // ScgFrame.num_rendered_out: the count BEFORE the lists were clipped to the capacity — what a caller that never
                // reads the host inside a step compares with its capacity afterwards
                if (nr_out) *nr_out = run + cnt;
            }
            if (cost_out) cost_out[t] = 0u;                     // the blend forward takes the maximum over the tile's waves
            const int c = (int)tile_class[t];
            order[t0 + s_first[c] + atomicAdd(&s_cnt[c], 1u)] = (uint32_t)t;
        }
        // tiles whose list does not fit the common 4-wave sort go on work lists for the rarer sizes: one atomic per wave
        // and list (in a dense scene EVERY tile is on a list: per-tile atomics on one counter cost 15 us at S4)
        const bool is_big = len > (uint32_t)kSortMidMax, is_mid = !is_big && len > small_max;
        const uint64_t m_mid = __ballot(is_mid), m_big = __ballot(is_big);
        const uint64_t m_split = __ballot(len > (uint32_t)kVeryLong);            // (rare: one atomic per wave that has any)
        if (lane == 0 && m_split) atomicAdd(&class_counts[3], (uint32_t)__popcll(m_split));
        uint32_t base_mid = 0, base_big = 0;
        if (lane == 0) {
            if (m_mid) base_mid = atomicAdd(&class_counts[0], (uint32_t)__popcll(m_mid));
            if (m_big) base_big = atomicAdd(&class_counts[1], (uint32_t)__popcll(m_big));
        }
        base_mid = (uint32_t)__shfl((int)base_mid, 0, kWave);
        base_big = (uint32_t)__shfl((int)base_big, 0, kWave);
        const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
        if (is_mid) mid_tiles[base_mid + (uint32_t)__popcll(m_mid & below)] = (uint32_t)t;
        if (is_big) big_tiles[base_big + (uint32_t)__popcll(m_big & below)] = (uint32_t)t;
    }
    // slots of the band's run that no tile took (the padded tail of the last band(s)): "no tile"
    for (int k = max(t0, t1) + (int)threadIdx.x; k < t0 + per; k += kScatterThreads) order[k] = (uint32_t)n_tiles;

    // Launch order of the blend BACKWARD's (tile, quadrant) waves, behind the tiles' order: 4 entries per slot, entry = 4 tile +
    // quadrant, this band's in [4 t0, 4 (t0 + per)).  With the trips every quadrant's wave made in the camera's previous
    // backward (ScgFrame.bwd_cost_in): most first — 64 classes, 16 per octave over 16 .. 256 trips, the order inside a class
    // whatever the LDS atomics give.  Without: the quadrants follow the tiles' order above.
    uint32_t* order_q = order + (size_t)tile_order_slots(n_tiles);
    const int slots8 = tile_order_slots(n_tiles) >> 3;             // (= per: slots of a band)
    const int nq = 4 * max(t1 - t0, 0);
    __syncthreads();                                               // (the band's tile order above is complete; s_cnt / s_first free)
    if (!bcost_in) {
        for (int e = threadIdx.x; e < 4 * slots8; e += kScatterThreads) {
            const uint32_t t = order[t0 + (e >> 2)];
            order_q[4 * (size_t)t0 + e] = (t < (uint32_t)n_tiles) ? 4u * t + (uint32_t)(e & 3) : 4u * (uint32_t)n_tiles;
        }
        return;
    }
    auto qclass = [&](int e) {                                     // e: quadrant of the band, 4 (tile - t0) + quadrant
        const uint32_t c = bcost_in[4 * (size_t)t0 + e];
        const int k = (int)(__float_as_uint((float)c) >> 19) - ((127 + 4) << 4);
        return max(0, min(kLenClasses - 1, k));
    };
    if (threadIdx.x < kLenClasses) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    // A quadrant's class is decided ONCE, from one read of the hint, and kept in LDS for the placement pass: the hint is the
    // caller's buffer (another stream's backward of the same camera may be writing it), and a class that differed between the
    // count and the placement would leave order_q short of a permutation — waves skipped or run twice, wrong gradients.
    // With one read per quadrant any content of the buffer yields a permutation ("never enters a result").
    for (int e = threadIdx.x; e < nq; e += kScatterThreads) {
        const int c = qclass(e);
        q_class[e] = (uint8_t)c;
        atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kLenClasses) {
        uint32_t before = 0;
        for (int k = (int)threadIdx.x + 1; k < kLenClasses; ++k) before += s_cnt[k];
        s_first[threadIdx.x] = before;
    }
    __syncthreads();
    if (threadIdx.x < kLenClasses) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    // (quadrant by quadrant, NOT tile by tile: with the four quadrants of a tile kept together — ordered by the tile's busiest
    // quadrant — the backward takes what it takes without a hint, 105.7 vs 101.3 us at S2, 83.6 vs 76.3 at S4, same process)
    for (int e = threadIdx.x; e < nq; e += kScatterThreads) {
        const int c = (int)q_class[e];                             // (written by this very thread above)
        order_q[4 * (size_t)t0 + s_first[c] + atomicAdd(&s_cnt[c], 1u)] = 4u * (uint32_t)t0 + (uint32_t)e;
    }
    for (int e = nq + (int)threadIdx.x; e < 4 * slots8; e += kScatterThreads) order_q[4 * (size_t)t0 + e] = 4u * (uint32_t)n_tiles;
}

// Scatter.  A 4-byte store per instance into a tile segment chosen by the instance is the worst case for a
// write-back L2: with workgroups walking Gaussian slices over the WHOLE image a 64-byte line of a segment collects
// its 16 ids over the whole kernel, the 22 MB list does not stay in a 4 MiB L2, and every store leaves the L2 as its
// own 32-byte sector (measured 208 MB of fabric writes for 22 MB of ids, 84 us at S3).  So the image is cut into 8
// bands of tile rows and workgroup (band, slice) scatters only the instances of its slice that fall into its band:
// workgroup % 8 = band = the XCD it runs on (MI355X_MICROARCH.md), so ONE L2 sees all stores to a band's segments
// (1/8 of the list: fits) and writes complete lines back.  Every band re-reads the slice's rectangles (8 bytes per
// Gaussian — cheap) and compacts the ones that touch the band into a per-wave LDS queue, so that the lanes walking
// rectangles are all busy.  slot = tile start + slice prefix (the column-scanned table) + LDS cursor; the order
// inside a segment is arbitrary (the per-tile sort makes it canonical).
__device__ __forceinline__ void band_rows(int gy, int band, int& r0, int& r1) {
    r0 = (int)((int64_t)gy * band / kBands);
    r1 = (int)((int64_t)gy * (band + 1) / kBands);
}

__global__ __launch_bounds__(kScatterThreads) void tile_scatter_kernel(const uint2* __restrict__ rects, uint32_t P,
                                                                   int grid_x, int grid_y, int n_slices,
                                                                   const uint32_t* __restrict__ table,
                                                                   const uint32_t* __restrict__ tile_total,
                                                                   const uint32_t* __restrict__ tile_part,
                                                                   uint32_t* __restrict__ point_list,
                                                                   uint32_t capacity,
                                                                   uint32_t* __restrict__ tile_start,
                                                                   uint2* __restrict__ ranges,
                                                                   uint32_t* __restrict__ class_counts,
                                                                   uint32_t* __restrict__ mid_tiles,
                                                                   uint32_t* __restrict__ big_tiles, uint32_t small_max,
                                                                   const uint32_t* __restrict__ len_hist,
                                                                   const uint8_t* __restrict__ tile_class,
                                                                   uint32_t* __restrict__ cost_out, int block_slices,
                                                                   const uint32_t* __restrict__ bcost_in,
                                                                   uint32_t* __restrict__ nr_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cursor = reinterpret_cast<uint32_t*>(smem);      // next free slot of this workgroup in each tile of the band
    __shared__ uint2 s_qrect[kScatterWaves][kQueue];
    __shared__ uint32_t s_qid[kScatterWaves][kQueue];
    __shared__ uint32_t s_red[2 * kScatterWaves];
    const int n_tiles = grid_x * grid_y;
    if (blockIdx.x < kBands) {                                 // the eight publishing workgroups (see above)
        publish_tile_starts((int)blockIdx.x, n_tiles, tile_total, tile_part, tile_start, ranges, capacity, class_counts,
                            mid_tiles, big_tiles, small_max, len_hist, tile_class, cost_out, bcost_in,
                            smem /* the cursors' space: a publishing workgroup scatters nothing; >= 4 ceil(n_tiles / 8) bytes */,
                            nr_out);
        return;
    }
#ifdef SCG_PROBE_TIMELINE
    uint32_t* tl = cost_out ? cost_out + n_tiles + ((size_t)blockIdx.x * kScatterWaves + wave_id()) * 8 : nullptr;
    const uint32_t tl0 = (uint32_t)wall_clock64();
    uint32_t tl_walk = 0, tl_nwalk = 0, tl_most = 0;
#endif
    const int wg = (int)blockIdx.x - kBands;                   // (wg & 7 == blockIdx.x & 7: the band is still the XCD)
    const int band = wg & (kBands - 1), slice = wg >> 3;
    int r0, r1;
    band_rows(grid_y, band, r0, r1);
    const int t_lo = r0 * grid_x, nt = (r1 - r0) * grid_x;
    if (nt == 0) return;
    // slot = tile start + slice prefix: the tile starts of the band are scanned here (exclusive scan of its tile totals
    // behind the sum of everything in front of the band)
    // The band's tile totals go through LDS (coalesced read), every thread scans `share` consecutive tiles in place, and ONE
    // exchange between the waves carries both the tiles in front of a wave and what lies before the band (64-tile sums of the
    // column scan + the tiles between the last full 64 and the band); then the slice's prefix is added (coalesced read).
    const int w = wave_id(), lane = lane_id();
    // the slice = the Gaussians of workgroup `slice` of tile_hist_kernel (its 16 wave slices) or of geometry_hist_kernel
    // (its 256-Gaussian blocks), split over 4 waves here
    uint32_t sa, sb, dummy;
    if (block_slices) {
        block_slice(P, (uint32_t)n_slices, (uint32_t)slice, sa, sb);
        sa = min(sa * (uint32_t)kBlock, P);
        sb = min(sb * (uint32_t)kBlock, P);
    } else {
        wave_slice(P, (uint32_t)n_slices, (uint32_t)slice, 0u, sa, dummy);
        wave_slice(P, (uint32_t)n_slices, (uint32_t)slice, (uint32_t)kBinWaves - 1u, dummy, sb);
    }
    const uint32_t ga = sa + (uint32_t)((uint64_t)(sb - sa) * w / kScatterWaves);
    const uint32_t gb = sa + (uint32_t)((uint64_t)(sb - sa) * (w + 1) / kScatterWaves);
    // the rectangles are fetched TWO rounds ahead (a round = 64 Gaussians, one load per lane: without the prefetch every
    // round waits for its own trip to the Infinity Cache — the rectangles were written by another XCD's compute units —
    // and a wave has three to fifteen rounds; four rounds ahead measured the same, round-5 timeline of the kernel)
    constexpr int kAhead = 2;
    auto fetch = [&](uint32_t g) { return (g < gb) ? rects[g] : make_uint2(0u, 0u); };
    uint2 r_q[kAhead];
#pragma unroll
    for (int a = 0; a < kAhead; ++a) r_q[a] = fetch(ga + (uint32_t)(a * kWave + lane));
    const uint32_t* row = table + (size_t)slice * n_tiles + t_lo;
    // (round 5: everything the workgroup reads from global memory before its walk — the band's tile totals, what lies in
    // front of the band, the slice's prefix of every tile, the first rectangles — is requested HERE, in one round trip; the
    // slice prefixes were fetched behind the scan, the rectangles behind that)
    constexpr int kRowRegs = 8;                                // tiles per thread kept in registers (bands up to 2 048 tiles)
    uint32_t row_pre[kRowRegs];
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) {
        const int k = j * kScatterThreads + (int)threadIdx.x;
        row_pre[j] = (k < nt) ? row[k] : 0u;
    }
    {
        // what lies in front of the band: 64-tile sums of the column scan + the tiles between the last full 64 and the band.
        // (Round 6: the usual sizes — up to 16 384 tiles — need ONE value of each per thread; they are requested here, with
        // clamped indices and no branch, in the same round trip as everything above: as loops with a data-dependent trip
        // count each was a load + wait of its own behind the batch below.)
        const int full = t_lo / kColTiles;
        uint32_t before = 0;
        const int tail_k = full * kColTiles + (int)threadIdx.x;
        const uint32_t part0 = tile_part[min((int)threadIdx.x, max(full - 1, 0))];
        const uint32_t tail0 = tile_total[min(tail_k, n_tiles - 1)];
        // (the band's tile totals likewise: up to kRowRegs per thread in one batch — as a loop the compiler's remainder code was
        //  two-then-one loads, each waited for)
        uint32_t tot_pre[kRowRegs];
#pragma unroll
        for (int j = 0; j < kRowRegs; ++j) tot_pre[j] = tile_total[t_lo + min(j * kScatterThreads + (int)threadIdx.x, nt - 1)];
#pragma unroll
        for (int j = 0; j < kRowRegs; ++j) {
            const int k = j * kScatterThreads + (int)threadIdx.x;
            if (k < nt) cursor[k] = tot_pre[j];
        }
        for (int k = kRowRegs * kScatterThreads + (int)threadIdx.x; k < nt; k += kScatterThreads) cursor[k] = tile_total[t_lo + k];
        if ((int)threadIdx.x < full) before += part0;
        if (tail_k < t_lo) before += tail0;
        for (int k = (int)threadIdx.x + kScatterThreads; k < full; k += kScatterThreads) before += tile_part[k];
        // (the tail holds fewer than 64 tiles: one value per thread covers it)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off, kWave);
        __syncthreads();
        const int share = (nt + kScatterThreads - 1) / kScatterThreads;
        const int k_a = min((int)threadIdx.x * share, nt), k_b = min(k_a + share, nt);
        uint32_t mine = 0;
        for (int k = k_a; k < k_b; ++k) mine += cursor[k];
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t n = (uint32_t)__shfl_up((int)inc, off, kWave);
            if (lane >= off) inc += n;
        }
        if (lane == kWave - 1) { s_red[w] = inc; s_red[kScatterWaves + w] = before; }
        __syncthreads();
        uint32_t start = inc - mine;                            // exclusive prefix inside the wave
#pragma unroll
        for (int k = 0; k < kScatterWaves; ++k) start += s_red[kScatterWaves + k] + (k < w ? s_red[k] : 0u);
        for (int k = k_a; k < k_b; ++k) {
            const uint32_t cnt = cursor[k];
            cursor[k] = start;
            start += cnt;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kRowRegs; ++j) {
            const int k = j * kScatterThreads + (int)threadIdx.x;
            if (k < nt) cursor[k] += row_pre[j];
        }
        for (int k = kRowRegs * kScatterThreads + (int)threadIdx.x; k < nt; k += kScatterThreads) cursor[k] += row[k];
        if (threadIdx.x < kWave) cursor[nt + threadIdx.x] = 0u;       // the dump slots of scatter_rects (the spare row)
    }
    __syncthreads();

#ifdef SCG_PROBE_TIMELINE
    const uint32_t tl1 = (uint32_t)wall_clock64();
#endif
    uint2* qrect = s_qrect[w];
    uint32_t* qid = s_qid[w];
    auto drop = [&](uint32_t tile, uint32_t id) {
        const uint32_t pos = atomicAdd(&cursor[tile - (uint32_t)t_lo], 1u);
        if (pos < capacity) point_list[pos] = id;
    };
    // One rectangle per lane (clipped to the band), every instance dropped into its tile's segment.  A lane's walk is a chain of
    // LDS atomics that return the slot — ~150 cycles each, and the wave walks as long as its largest rectangle (up to 48 tiles):
    // the atomics of FOUR tiles are issued back to back and waited for once (round 5: the walk was a third of the kernel).
    // No branch around an atomic (the compiler would wait behind each): a lane whose rectangle has ended adds to a dump slot
    // in the spare row behind the band's cursors.
    auto scatter_rects = [&](uint2 r, uint32_t g) {
#ifdef SCG_PROBE_TIMELINE
        const uint32_t tw0 = (uint32_t)wall_clock64();
#endif
        const uint32_t wd = r.y & 0xFFFFu, ht = r.y >> 16;
        const uint32_t cnt = wd * ht;
        const uint32_t dump = (uint32_t)nt + (uint32_t)lane;
        uint32_t small = (cnt <= kCoopThreshold) ? cnt : 0u;
        uint32_t x = 0, row_tile = (r.x >> 16) * (uint32_t)grid_x + (r.x & 0xFFFFu) - (uint32_t)t_lo;
        // (wave-uniform trip count: the largest rectangle of the wave)
        uint32_t most = small;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, off, kWave));
        for (uint32_t k = 0; k < most; k += 4) {
            uint32_t slot[4], pos[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                slot[u] = (k + u < small) ? row_tile + x : dump;
                if (++x == wd) { x = 0; row_tile += (uint32_t)grid_x; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) pos[u] = atomicAdd(&cursor[slot[u]], 1u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + u < small && pos[u] < capacity) point_list[pos[u]] = g;
        }
        walk_rects(cnt > kCoopThreshold ? r : make_uint2(0u, 0u), g, grid_x, drop);     // large rectangles: by the whole wave
#ifdef SCG_PROBE_TIMELINE
        tl_walk += (uint32_t)wall_clock64() - tw0; tl_nwalk += 1; tl_most += most;
#endif
    };
    int qn = 0;                                                // wave-uniform queue length
    for (uint32_t g0 = ga; g0 < gb; g0 += kWave) {
        const uint32_t g = g0 + lane;
        const uint2 r = r_q[0];
#pragma unroll
        for (int a = 0; a + 1 < kAhead; ++a) r_q[a] = r_q[a + 1];
        r_q[kAhead - 1] = fetch(g + kAhead * kWave);
        // clip the rectangle's rows to the band
        const int y0 = max((int)(r.x >> 16), r0), y1 = min((int)(r.x >> 16) + (int)(r.y >> 16), r1);
        const bool touches = (r.y & 0xFFFFu) != 0u && y1 > y0;
        const uint64_t m = __ballot(touches);
        if (touches) {
            const int at = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            qrect[at] = make_uint2((r.x & 0xFFFFu) | ((uint32_t)y0 << 16), (r.y & 0xFFFFu) | ((uint32_t)(y1 - y0) << 16));
            qid[at] = g;
        }
        qn += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (qn >= kWave) {                                     // a full wave of work: walk it, keep the remainder
            const uint2 qr = qrect[lane];
            const uint32_t qg = qid[lane];
            const uint2 mr = qrect[kWave + lane];
            const uint32_t mg = qid[kWave + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            qn -= kWave;
            if (lane < qn) { qrect[lane] = mr; qid[lane] = mg; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            scatter_rects(qr, qg);
        }
    }
    if (qn > 0) {
        uint2 qr = make_uint2(0u, 0u);
        uint32_t qg = 0;
        if (lane < qn) { qr = qrect[lane]; qg = qid[lane]; }
        scatter_rects(qr, qg);
    }
#ifdef SCG_PROBE_TIMELINE
    if (tl && lane == 0) {
        tl[0] = tl0; tl[1] = tl1; tl[2] = (uint32_t)wall_clock64(); tl[3] = tl_walk; tl[4] = tl_nwalk; tl[5] = tl_most;
        tl[6] = gb - ga; tl[7] = 0xC0FFEEu;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// per-tile sort on (depth bits, id)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sort_tile_in_lds(uint64_t* s_keys, const uint32_t* __restrict__ depth_keys,
                                                 uint32_t* __restrict__ list, int n) {
    for (int i = threadIdx.x; i < n; i += (int)blockDim.x) {
        const uint32_t id = list[i];
        s_keys[i] = ((uint64_t)depth_keys[id] << 32) | (uint64_t)id;
    }
    __syncthreads();
    bitonic_sort_asc(s_keys, n, false);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += (int)blockDim.x) list[i] = (uint32_t)s_keys[i];
}

// The common case: one 4-wave workgroup per tile, lists of 2..2048 entries (20 KiB of LDS, 7 workgroups per CU).
// Dense scenes (a million Gaussians on a small image: the AVERAGE list has thousands of entries, S4: 2 100) would send
// half of their tiles to the rare kernel, whose 16-wave workgroups run one per compute unit: for them the host launches
// the 8-wave variant instead (lists up to 4096 entries, 49 KiB of LDS, 3 workgroups per CU).
template <int NW, int MAX_N>
__global__ __launch_bounds__(NW * kWave) void tile_sort_kernel(const uint2* __restrict__ ranges,
                                                               const uint32_t* __restrict__ depth_keys,
                                                               uint32_t* __restrict__ point_list, int id_bits) {
    __shared__ TileSortLds<NW, MAX_N> L;
    const uint2 r = ranges[blockIdx.x];
    const int n = (int)(r.y - r.x);
    if (n < 2 || n > MAX_N) return;
    sort_one_tile<NW, MAX_N, MAX_N>(L, r, depth_keys, point_list, id_bits);
}

// ---- long lists: sort_long_list<threads, buckets> of tile_sort.h (shared with the forward blend's fallback) ----------
constexpr int kRareThreads = 16 * kWave;
constexpr int kLongBuckets = 8192;

// One workgroup sorts one long list completely (the path of a frame whose long lists were not split beforehand, and the
// fallback of the split for heavily tied depths).
__device__ __forceinline__ void sort_big_tile(unsigned char* smem, const uint2 r, const uint32_t* __restrict__ depth_keys,
                                              uint32_t* __restrict__ point_list, uint64_t* __restrict__ spill,
                                              uint64_t* __restrict__ spill2) {
    const int n = (int)(r.y - r.x);
    uint32_t* list = point_list + r.x;
    uint64_t* keys = spill + r.x;
    if (!sort_long_list<kRareThreads, kLongBuckets>(smem, depth_keys, list, n, keys, spill2 + r.x)) {
        __syncthreads();
        if (n <= kSortBigLdsMax) {
            sort_tile_in_lds(reinterpret_cast<uint64_t*>(smem), depth_keys, list, n);
        } else {
            for (int i = threadIdx.x; i < n; i += kRareThreads) {
                const uint32_t id = list[i];
                keys[i] = ((uint64_t)depth_keys[id] << 32) | (uint64_t)id;
            }
            __syncthreads();
            bitonic_sort_asc(keys, n, true);
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += kRareThreads) list[i] = (uint32_t)keys[i];
        }
    }
}

// ---- the rarer list sizes when the camera's previous render is known (scg_forward, SCG_FORWARD_RARE_8WAVE) --------------------
// tile_sort_rare_kernel below runs ONE 16-wave workgroup per compute unit (96-128 KiB of LDS, 128 registers): fine for a handful
// of lists, a queue of several rounds for a scene with a heavy tail (hundreds of tiles between 1 536 and 8 192 entries), and a
// list of 45 000 entries keeps one workgroup busy for ~170 us of dependent passes over global memory while 255 compute units
// idle.  The path for frames whose previous render told the caller what to expect:
//   tile_split_long_kernel   (only when lists beyond kSort8Max = 4 096 entries were seen) one 16-wave workgroup per such list
//                            runs the FIRST TWO passes of the long-list bucket sort — composites + 8 192-bucket histogram, then
//                            the ids scattered into bucket order, in place — and cuts the bucket-ordered list into SEGMENTS of
//                            consecutive buckets (a segment = the buckets that start in [m T, (m + 1) T), T = kSegTarget: never
//                            empty, at most T + kLongBucketMax = 2 048 entries).  Segments are depth-disjoint and in depth
//                            order: sorting each one on (depth, id) sorts the list.  Heavily tied depths (a bucket beyond
//                            kLongBucketMax entries) are sorted on the spot by the old path.
//   tile_sort_list8_kernel   8-wave workgroups, 49 KiB of LDS, THREE per compute unit, walk one work list: the segments, then
//                            the tiles of kFusedMaxN + 1 .. 4 096 entries — the per-tile bucket / radix sort of tile_sort.h.
//                            A list beyond 4 096 entries that was NOT split (the scene changed since the previous render) is
//                            sorted by the same workgroup through global scratch: correct, slow, and the next render's hint
//                            knows about it.
constexpr int kSegTarget = 1024;                // (measured, 60 %-clustered scene, binning stage: 1 024 -> 131 us, 2 048 -> 146,
                                                //  3 072 -> 142: parts of at most 2 048 entries take the sort's 4-keys-per-thread form)
constexpr int kSplitBatch = 16;                 // entries per thread in flight in the split's two passes (8 / 16 / 24: 129.9 / 129.0 /
                                                // 127.9 us of binning — the passes wait for ONE compute unit's rate of divergent
                                                // gathers and scattered stores, ~1 per cycle: 45 000 entries = 2 x 20 us)
constexpr int kSegMaxPerTile = 8192;            // lists up to 8 M entries (longer ones: the unsplit path)
static_assert(kSegTarget + kLongBucketMax <= kSort8Max, "a segment must fit the 8-wave sort");
static_assert(kLongBucketMax <= kSegTarget, "consecutive bucket starts must be at most a segment apart: no segment index is skipped");

__device__ __forceinline__ bool split_long_list(unsigned char* smem, const uint2 r, const uint32_t* __restrict__ depth_keys,
                                                uint32_t* __restrict__ point_list, uint64_t* __restrict__ A,
                                                uint32_t* __restrict__ seg_count, uint2* __restrict__ segments) {
    constexpr int T = kRareThreads, NW = T / kWave, PER = kLongBuckets / T;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);                      // [kLongBuckets + 1] counts -> starts
    uint32_t* cur = cnt + kLongBuckets + 4;                                  // [kLongBuckets] running cursors
    uint32_t* red = cur + kLongBuckets;                                      // [2 * NW] reductions (+ 1: the segments' base)
    uint32_t* pm = red + 2 * NW + 4;                                         // [kSegMaxPerTile + 1] segment boundaries
    const int n = (int)(r.y - r.x);
    uint32_t* list = point_list + r.x;
    const int t = threadIdx.x, w = wave_id(), lane = lane_id();
    const int n_seg = (n + kSegTarget - 1) / kSegTarget;
    if (n_seg > kSegMaxPerTile) return false;
    uint32_t kmin, kmax;
    {
        const uint32_t key = depth_keys[list[(int)(((int64_t)t * n) / T)]];    // range from a sample: the map only has to be monotone
        kmin = key; kmax = key;
    }
    for (int b = t; b < kLongBuckets; b += T) cnt[b] = 0u;
    for (int m = t; m <= n_seg; m += T) pm[m] = (uint32_t)n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, kWave));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, kWave));
    }
    if (lane == 0) { red[2 * w] = kmin; red[2 * w + 1] = kmax; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) { kmin = min(kmin, red[2 * k]); kmax = max(kmax, red[2 * k + 1]); }
    const int sh = __builtin_clz((kmax - kmin) | 1u);
    auto bucket_of = [&](uint64_t comp) {
        const uint32_t key = (uint32_t)(comp >> 32);
        return __umulhi((min(max(key, kmin), kmax) - kmin) << sh, (uint32_t)kLongBuckets);
    };
    // (one workgroup, two dependent gathers per entry: kSplitBatch entries per thread in flight — ids, then keys, then the stores)
    for (int i0 = t; i0 < n; i0 += kSplitBatch * T) {
        uint32_t id[kSplitBatch], key[kSplitBatch];
#pragma unroll
        for (int u = 0; u < kSplitBatch; ++u) id[u] = list[min(i0 + u * T, n - 1)];
#pragma unroll
        for (int u = 0; u < kSplitBatch; ++u) key[u] = depth_keys[id[u]];
#pragma unroll
        for (int u = 0; u < kSplitBatch; ++u) {
            if (i0 + u * T < n) {
                const uint64_t comp = ((uint64_t)key[u] << 32) | (uint64_t)id[u];
                A[i0 + u * T] = comp;
                atomicAdd(&cnt[bucket_of(comp)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t c[PER], sum = 0, cmax = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { c[j] = cnt[t * PER + j]; sum += c[j]; cmax = max(cmax, c[j]); }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, kWave);
        if (lane >= off) incl += up;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off, kWave));
    __syncthreads();
    if (lane == kWave - 1) red[w] = incl;
    if (lane == 0) red[NW + w] = cmax;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int k = 0; k < w; ++k) base += red[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) cmax = max(cmax, red[NW + k]);
    if (cmax > (uint32_t)kLongBucketMax) return false;             // uniform: tied depths, the caller sorts the list itself
    // bucket starts (cnt[b], cnt[NB] = n) and cursors
#pragma unroll
    for (int j = 0; j < PER; ++j) { cnt[t * PER + j] = base; cur[t * PER + j] = base; base += c[j]; }
    if (t == T - 1) cnt[kLongBuckets] = base;                        // = n
    __syncthreads();
    // segment m begins at the FIRST bucket start at or behind m * kSegTarget.  Consecutive starts are at most
    // kLongBucketMax <= kSegTarget apart (static_assert above), so no m is skipped and the boundaries are strictly increasing: every segment is
    // non-empty and shorter than kSegTarget + kLongBucketMax.  (Empty buckets share their start with the next one: the
    // racing writes carry the same value.)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = t * PER + j;
        const uint32_t s_b = cnt[b];
        if (b == 0) {
            pm[0] = 0u;
        } else if (s_b < (uint32_t)n) {
            const uint32_t m = s_b / (uint32_t)kSegTarget;
            if (m * (uint32_t)kSegTarget > cnt[b - 1]) pm[m] = s_b;
        }
    }
    __syncthreads();
    for (int i0 = t; i0 < n; i0 += kSplitBatch * T) {                  // ids in bucket order, in place (read from A)
        uint64_t comp[kSplitBatch];
#pragma unroll
        for (int u = 0; u < kSplitBatch; ++u) comp[u] = A[min(i0 + u * T, n - 1)];
#pragma unroll
        for (int u = 0; u < kSplitBatch; ++u)
            if (i0 + u * T < n) list[atomicAdd(&cur[bucket_of(comp[u])], 1u)] = (uint32_t)comp[u];
    }
    // the valid boundaries are a prefix pm[0 .. n_valid); segment m = [pm[m], pm[m + 1]) (the last one ends at n)
    int n_valid = 0;
    for (int m0 = 0; m0 < n_seg; m0 += T) n_valid += __syncthreads_count((m0 + t < n_seg) && pm[m0 + t] < (uint32_t)n);
    if (t == 0) red[2 * NW] = atomicAdd(seg_count, (uint32_t)n_valid);
    __syncthreads();
    const uint32_t out = red[2 * NW];
    for (int m = t; m < n_valid; m += T)
        segments[out + m] = make_uint2(r.x + pm[m], r.x + ((m + 1 < n_valid) ? pm[m + 1] : (uint32_t)n));
    return true;
}

__global__ __launch_bounds__(kRareThreads) void tile_split_long_kernel(const uint2* __restrict__ ranges,
                                                                       const uint32_t* __restrict__ depth_keys,
                                                                       uint32_t* __restrict__ point_list,
                                                                       uint64_t* __restrict__ spill,
                                                                       uint64_t* __restrict__ spill2,
                                                                       uint32_t* __restrict__ class_counts,
                                                                       const uint32_t* __restrict__ mid_tiles,
                                                                       const uint32_t* __restrict__ big_tiles,
                                                                       uint2* __restrict__ segments) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t n_mid = class_counts[0], n_big = class_counts[1];
    // the long lists first (dealt round-robin), then the mid-size work list, most of whose entries are skipped at once
    for (uint32_t i = blockIdx.x; i < n_big + n_mid; i += gridDim.x) {
        const uint2 r = ranges[i < n_big ? big_tiles[i] : mid_tiles[i - n_big]];
        if (r.y - r.x <= (uint32_t)kSort8Max) continue;                       // tile_sort_list8_kernel sorts it as it is
        if (!split_long_list(smem, r, depth_keys, point_list, spill + r.x, class_counts + 2, segments)) {
            __syncthreads();
            sort_big_tile(smem, r, depth_keys, point_list, spill, spill2);
            __syncthreads();
            if (threadIdx.x == 0) {                                            // sorted already: an empty marker keeps the
                const uint32_t at = atomicAdd(class_counts + 2, 1u);          // "was split" bookkeeping uniform
                segments[at] = make_uint2(r.x, r.x);
            }
        }
        __syncthreads();
    }
}

constexpr int kList8Threads = 8 * kWave;
constexpr int kList8LongBuckets = 4096;

// (a function of its own: its registers must not count against the common path's 80)
__device__ __noinline__ void sort_unsplit_list8(unsigned char* smem, const uint2 r, const uint32_t* __restrict__ depth_keys,
                                                uint32_t* __restrict__ point_list, uint64_t* __restrict__ spill,
                                                uint64_t* __restrict__ spill2) {
    const int n = (int)(r.y - r.x);
    uint32_t* list = point_list + r.x;
    uint64_t* keys = spill + r.x;
    if (!sort_long_list<kList8Threads, kList8LongBuckets, 2>(smem, depth_keys, list, n, keys, spill2 + r.x)) {
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kList8Threads) {
            const uint32_t id = list[k];
            keys[k] = ((uint64_t)depth_keys[id] << 32) | (uint64_t)id;
        }
        __syncthreads();
        bitonic_sort_asc(keys, n, true);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += kList8Threads) list[k] = (uint32_t)keys[k];
    }
}

__global__ __launch_bounds__(kList8Threads) __attribute__((amdgpu_waves_per_eu(6, 6))) void tile_sort_list8_kernel(const uint2* __restrict__ ranges,
                                                                        const uint32_t* __restrict__ depth_keys,
                                                                        uint32_t* __restrict__ point_list, int id_bits,
                                                                        uint64_t* __restrict__ spill,
                                                                        uint64_t* __restrict__ spill2,
                                                                        const uint32_t* __restrict__ class_counts,
                                                                        const uint32_t* __restrict__ mid_tiles,
                                                                        const uint32_t* __restrict__ big_tiles,
                                                                        const uint2* __restrict__ segments, int presplit) {
    __shared__ TileSortLds<8, kSort8Max> L;
    static_assert(sizeof(L) >= (2 * kList8LongBuckets + 4 + 2 * 8 + 8) * sizeof(uint32_t), "the unsplit fallback's counters must fit");
    const uint32_t n_mid = class_counts[0], n_big = class_counts[1];
    const uint32_t n_seg = presplit ? class_counts[2] : 0u;
    for (uint32_t i = blockIdx.x; i < n_seg + n_mid + n_big; i += gridDim.x) {
        uint2 r;
        if (i < n_seg) r = segments[i];
        else if (i < n_seg + n_mid) r = ranges[mid_tiles[i - n_seg]];
        else r = ranges[big_tiles[i - n_seg - n_mid]];
        const int n = (int)(r.y - r.x);
        if (n >= 2 && n <= kSort8Max) {
            sort_one_tile<8, kSort8Max, kSort8Max>(L, r, depth_keys, point_list, id_bits);
        } else if (n > kSort8Max && !presplit) {
            // not split beforehand (the caller's expectation was wrong): this workgroup sorts the whole list through global scratch
            sort_unsplit_list8(reinterpret_cast<unsigned char*>(&L), r, depth_keys, point_list, spill, spill2);
        }
        __syncthreads();
    }
}

// The rarer list sizes without such knowledge (the staged calls; the first render of a camera): one launch, a small fixed grid of
// 16-wave workgroups walking the work lists the scatter's publishing workgroups built (so the launch costs next to nothing when
// they are empty):
//   1 537 / 2 049 .. 8 192 entries: the bucket / radix sort of tile_sort.h with 8 keys per thread (96 KiB of LDS);
//   longer lists: sort_big_tile (one workgroup per list: the bucket sort with the entries in global scratch; spill holds two
//   copies of R composites, a tile uses spill + its range start: tiles never overlap; heavily tied depths: the bitonic network
//   on the 64-bit key, in 128 KiB of LDS up to 16 384 entries, else in global scratch).
constexpr size_t kRareLds = (size_t)kSortBigLdsMax * sizeof(uint64_t) > sizeof(TileSortLds<16, kSortMidMax>)
                                ? (size_t)kSortBigLdsMax * sizeof(uint64_t) : sizeof(TileSortLds<16, kSortMidMax>);
static_assert(kRareLds >= (2 * kLongBuckets + 4 + 2 * 16 + 8 + kSegMaxPerTile + 1) * sizeof(uint32_t), "split_long_list's LDS");

__global__ __launch_bounds__(kRareThreads) void tile_sort_rare_kernel(const uint2* __restrict__ ranges,
                                                                      const uint32_t* __restrict__ depth_keys,
                                                                      uint32_t* __restrict__ point_list, int id_bits,
                                                                      uint64_t* __restrict__ spill,
                                                                      uint64_t* __restrict__ spill2,
                                                                      const uint32_t* __restrict__ class_counts,
                                                                      const uint32_t* __restrict__ mid_tiles,
                                                                      const uint32_t* __restrict__ big_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TileSortLds<16, kSortMidMax>& L = *reinterpret_cast<TileSortLds<16, kSortMidMax>*>(smem);
    const uint32_t n_mid = class_counts[0], n_big = class_counts[1];
    for (uint32_t i = blockIdx.x; i < n_mid; i += gridDim.x) {
        sort_one_tile<16, kSortMidMax, kSortMidMax>(L, ranges[mid_tiles[i]], depth_keys, point_list, id_bits);
        __syncthreads();
    }
    // the long lists start on the LAST workgroups, so the first ones do not stack on top of a mid-size list
    for (uint32_t t = gridDim.x - 1 - blockIdx.x; t < n_big; t += gridDim.x) {
        sort_big_tile(smem, ranges[big_tiles[t]], depth_keys, point_list, spill, spill2);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void rebuild_keys_kernel(const uint32_t* __restrict__ tile_start, int n_tiles,
                                                              const uint32_t* __restrict__ point_list,
                                                              const uint32_t* __restrict__ depth_keys, int64_t R,
                                                              uint64_t* __restrict__ keys_sorted) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= R) return;
    int lo = 0, hi = n_tiles;            // invariant: tile_start[lo] <= s < tile_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= (uint32_t)s) lo = mid; else hi = mid;
    }
    keys_sorted[s] = ((uint64_t)(uint32_t)lo << 32) | (uint64_t)depth_keys[point_list[s]];
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Per-DEVICE one-time setup (a process may rasterize on several GPUs, from several threads): the dynamic-LDS
// attributes of the two big-LDS kernels are a property of the (kernel, device) pair, and the rare-sort grid is sized
// by that device's CU count.
struct DeviceSetup {
    std::once_flag once;
    bool ok = false;
    int n_cus = 256;
};
constexpr int kMaxDevices = 64;
static DeviceSetup g_device_setup[kMaxDevices];

static const DeviceSetup* device_setup() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    DeviceSetup& d = g_device_setup[dev];
    std::call_once(d.once, [&] {
        int v = 0;
        const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_hist_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_sort_rare_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRareLds);
        if (e1 == hipSuccess)
            e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_split_long_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRareLds);
        const hipError_t e2 = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        hipError_t e3 = geometry_hist_set_max_lds(kMaxDynLds);
        if (e3 == hipSuccess) e3 = geometry_hist_model_set_max_lds(kMaxDynLds);
        d.n_cus = (e2 == hipSuccess && v > 0) ? v : 256;
        d.ok = (e0 == hipSuccess && e1 == hipSuccess && e3 == hipSuccess);
    });
    return d.ok ? &d : nullptr;
}

int tile_binning_blocks(int P, int64_t R) {
    return (R >= kLargeFrame || P >= kLargeScene) ? kTileBlocksLarge : kTileBlocksSmall;
}

bool tile_binning_supported(int n_tiles, int64_t R) {
    // one uint32 per tile in LDS (histogram / cursors) and a bounded table
    return (size_t)n_tiles * 4 <= (size_t)kMaxDynLds - 2048 && (size_t)kTileBlocksLarge * n_tiles * 4 <= (1ull << 30);
}

TileBinningLayout tile_binning_layout(int P, int64_t R, int n_tiles) {
    TileBinningLayout L;
    size_t off = 0;
    const int nb = tile_binning_blocks(P, R);
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    L.table = take((size_t)nb * n_tiles * 4);
    L.tile_total = take((size_t)n_tiles * 4);
    L.tile_start = take((size_t)(n_tiles + 1) * 4);
    L.class_counts = take(16);
    L.mid_tiles = take((size_t)n_tiles * 4);
    L.big_tiles = take((size_t)n_tiles * 4);
    L.len_hist = take((size_t)2 * kBands8 * kLenClasses * 4);
    L.tile_part = take((size_t)((n_tiles + kColTiles - 1) / kColTiles) * 4);   // 64-tile sums of the tile totals (column scan)
    L.tile_class = take((size_t)n_tiles);   // launch-order class of every tile, decided once (column scan) and reused
    L.spill = take((size_t)R * 16);          // two copies of the 64-bit (depth, id) composites: only touched by tiles
                                             // with more than kSortMidMax entries
    L.segments = take(((size_t)R / 1024 + 2 * (size_t)n_tiles + 8) * sizeof(uint2));   // parts of split long lists (>= 1 024 entries each)
    L.total = off;
    L.nblocks = nb;
    return L;
}

// (round 5: dense frames defer their sort to the forward blend as well — its 3 584-entry, eight-wave variant)
bool tile_binning_defers_sort(int64_t R, int n_tiles) { return true; }

// geometry_hist_kernel keeps n_tiles + a few words of LDS like tile_hist_kernel and two of its workgroups share a compute unit
bool tile_binning_hist_in_geometry(const FrameDev& f, int64_t R) {
    const int n_tiles = f.gx * f.gy;
    return f.P > 0 && tile_binning_supported(n_tiles, R) &&
           (size_t)(n_tiles + (f.P + kBlock - 1) / kBlock / tile_binning_blocks(f.P, R) + 2) * 4 + 16 +
                   (size_t)kBinThreads * 64                                        // + the sixteen waves' output stages (64 KiB)
               <= (size_t)kMaxDynLds - 2048;
}

int launch_geometry_hist_binned(const FrameDev& f, int64_t R, const float* means3D, const float* opacities, const float* shs,
                                const float* colors_precomp, const float* scales, const float* rotations,
                                const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped, uint32_t* rects,
                                uint32_t* depth_keys, uint32_t* block_sums, void* bin_scratch, hipStream_t stream) {
    const TileBinningLayout L = tile_binning_layout(f.P, R, f.gx * f.gy);
    char* base = reinterpret_cast<char*>(bin_scratch);
    if (!device_setup()) return fail(SCG_E_RANGE, "tile binning: device setup failed (hipFuncSetAttribute / device query)");
    return launch_geometry_hist(f, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, splats, radii,
                                clamped, rects, depth_keys, block_sums, L.nblocks,
                                reinterpret_cast<uint32_t*>(base + L.table), reinterpret_cast<uint32_t*>(base + L.class_counts),
                                reinterpret_cast<uint32_t*>(base + L.len_hist), stream);
}

int launch_geometry_hist_binned_model(const FrameDev& f, int64_t R, const ScgModel& m, float* splats, int32_t* radii,
                                      uint8_t* clamped, uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums,
                                      void* bin_scratch, hipStream_t stream) {
    const TileBinningLayout L = tile_binning_layout(f.P, R, f.gx * f.gy);
    char* base = reinterpret_cast<char*>(bin_scratch);
    if (!device_setup()) return fail(SCG_E_RANGE, "tile binning: device setup failed (hipFuncSetAttribute / device query)");
    return launch_geometry_hist_model(f, m, splats, radii, clamped, rects, depth_keys, block_sums, L.nblocks,
                                      reinterpret_cast<uint32_t*>(base + L.table), reinterpret_cast<uint32_t*>(base + L.class_counts),
                                      reinterpret_cast<uint32_t*>(base + L.len_hist), stream);
}

int launch_tile_binning(const FrameDev& f, int64_t R, const uint32_t* rects, const uint32_t* depth_keys,
                        uint32_t* point_list, uint32_t* ranges, uint64_t* keys_sorted, void* scratch,
                        bool* defer_sort, bool hist_done, bool skip_rare, bool rare8, bool split_long, hipStream_t stream) {
    const int P = f.P;
    const int n_tiles = f.gx * f.gy;
    const TileBinningLayout L = tile_binning_layout(P, R, n_tiles);
    char* base = reinterpret_cast<char*>(scratch);
    uint32_t* table = reinterpret_cast<uint32_t*>(base + L.table);
    uint32_t* tile_total = reinterpret_cast<uint32_t*>(base + L.tile_total);
    uint32_t* tile_start = reinterpret_cast<uint32_t*>(base + L.tile_start);
    uint64_t* spill = reinterpret_cast<uint64_t*>(base + L.spill);
    uint32_t* class_counts = reinterpret_cast<uint32_t*>(base + L.class_counts);
    uint32_t* mid_tiles = reinterpret_cast<uint32_t*>(base + L.mid_tiles);
    uint32_t* big_tiles = reinterpret_cast<uint32_t*>(base + L.big_tiles);
    uint2* segments = reinterpret_cast<uint2*>(base + L.segments);
    uint32_t* len_hist = reinterpret_cast<uint32_t*>(base + L.len_hist);
    uint8_t* tile_class = reinterpret_cast<uint8_t*>(base + L.tile_class);
    uint32_t* tile_part = reinterpret_cast<uint32_t*>(base + L.tile_part);
    const uint2* rects2 = reinterpret_cast<const uint2*>(rects);
    uint2* ranges2 = reinterpret_cast<uint2*>(ranges);

    const DeviceSetup* ds = device_setup();
    if (!ds) return fail(SCG_E_RANGE, "tile binning: device setup failed (hipFuncSetAttribute / device query)");
    const int nb = L.nblocks;
    const bool dense = R / n_tiles >= kDenseMeanList;       // R = the capacity the lists were sized for
    // the caller's forward blend sorts the common tiles itself: only the lists it does not take are sorted here
    const bool deferred = defer_sort && *defer_sort && tile_binning_defers_sort(R, n_tiles) && !keys_sorted;
    if (defer_sort) *defer_sort = deferred;
    const size_t lds_tiles = (size_t)n_tiles * sizeof(uint32_t);
    if (!hist_done)
        hipLaunchKernelGGL(tile_hist_kernel, dim3(nb), dim3(kBinThreads), lds_tiles, stream, rects2, (uint32_t)P, f.gx,
                           n_tiles, table, class_counts, len_hist);
    // length classes: the average list lands around class 16..31
    int len_shift = 0;
    while (((R / n_tiles) >> len_shift) >= 32) ++len_shift;
    hipLaunchKernelGGL(table_colscan_kernel, dim3((n_tiles + kColTiles - 1) / kColTiles), dim3(kBinThreads), 0, stream,
                       table, nb, n_tiles, tile_total, len_hist, len_shift, f.cost_in, tile_class, tile_part);
    const size_t lds_band = ((size_t)((f.gy + kBands - 1) / kBands + 1) * f.gx + kWave) * sizeof(uint32_t);   // + 64 dump slots
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(kBands + nb * kBands), dim3(kScatterThreads), lds_band, stream, rects2,
                       (uint32_t)P, f.gx, f.gy, nb, table, tile_total, tile_part, point_list, (uint32_t)R, tile_start, ranges2,
                       class_counts, mid_tiles, big_tiles,
                       (uint32_t)(deferred ? fused_max_list(R, n_tiles) : dense ? kSortDenseMax : kSortSmallMax), len_hist, tile_class,
                       f.cost_out, hist_done ? 1 : 0, f.bcost_in, f.nr_out);
    int id_bits = 8;
    while (id_bits < 32 && (1ll << id_bits) < (long long)P) id_bits += 8;
    if (deferred) {
        // (the forward blend sorts: nothing here)
    } else if (dense)
        hipLaunchKernelGGL((tile_sort_kernel<8, kSortDenseMax>), dim3(n_tiles), dim3(8 * kWave), 0, stream, ranges2,
                           depth_keys, point_list, id_bits);
    else
        hipLaunchKernelGGL((tile_sort_kernel<4, kSortSmallMax>), dim3(n_tiles), dim3(4 * kWave), 0, stream, ranges2,
                           depth_keys, point_list, id_bits);
    // one 128-KiB-LDS workgroup fits a compute unit at a time: more workgroups than CUs would only queue.  (An idle
    // launch — no list of a rare size, the usual case — costs 4.2 us whatever the grid: measured with 512 and 256.)
    // Not launched at all when the caller promises a frame without such lists (skip_rare: the forward blend that sorts its own
    // tiles has a fallback for a list that is longer after all).
    const int n_cus = ds->n_cus;
    if (!(deferred && skip_rare)) {
        if (rare8) {
            // the camera's previous render is known: lists beyond kSort8Max entries (if it had any) are partitioned by depth,
            // then 8-wave workgroups, three per compute unit, sort the parts and the mid-size tiles
            if (split_long)
                hipLaunchKernelGGL(tile_split_long_kernel, dim3(n_tiles < n_cus ? n_tiles : n_cus), dim3(kRareThreads), kRareLds,
                                   stream, ranges2, depth_keys, point_list, spill, spill + R, class_counts, mid_tiles, big_tiles,
                                   segments);
            hipLaunchKernelGGL(tile_sort_list8_kernel, dim3(3 * n_cus), dim3(kList8Threads), 0, stream, ranges2, depth_keys,
                               point_list, id_bits, spill, spill + R, class_counts, mid_tiles, big_tiles, segments,
                               split_long ? 1 : 0);
        } else {
            hipLaunchKernelGGL(tile_sort_rare_kernel, dim3(n_tiles < n_cus ? n_tiles : n_cus), dim3(kRareThreads), kRareLds,
                               stream, ranges2, depth_keys, point_list, id_bits, spill, spill + R, class_counts, mid_tiles,
                               big_tiles);
        }
    }
    if (keys_sorted) {
        const int kb = (int)((R + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(rebuild_keys_kernel, dim3(kb), dim3(kBlock), 0, stream, tile_start, n_tiles, point_list,
                           depth_keys, R, keys_sorted);
    }
    return check_hip(hipGetLastError(), "tile binning");
}

}  // namespace scg
