// binning.hip — tile binning: inclusive scan, duplicateWithKeys, stable LSD radix sort of
// (uint64 key, uint32 value) pairs, identifyTileRanges.
//
// All integer work; results must be bit-identical to oracle/torch_rasterizer.py::bin_and_sort.
// HBM-bound streaming / scatter kernels: 16-byte lane-contiguous loads, wave64 ballot ranking, LDS
// histograms.  No MFMA (nothing here is a contraction).
#include "scg_common.h"

namespace scg {

// ===================================================================================================
// inclusive scan of uint32: 256 elements per workgroup-chunk
//   phase 1  block sums         (fused into geometry_forward_kernel, or scan_block_sums_kernel)
//   phase 2  exclusive scan of the block sums, single workgroup
//   phase 3  local scan + block prefix
// ===================================================================================================
constexpr int kScanChunk = kBlock;   // one element per thread: matches the geometry kernel's blocking

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t n = __shfl_up(v, off, kWave);
        if (lane_id() >= off) v += n;
    }
    return v;
}

// inclusive scan over the 256 threads of a workgroup; returns the inclusive value, total in *total
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t* total) {
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == kWave - 1) s_wave[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0;
    const int w = wave_id();
    const uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
    if (w > 0) base += w0;
    if (w > 1) base += w1;
    if (w > 2) base += w2;
    if (total) *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return inc + base;
}

__global__ __launch_bounds__(kBlock) void scan_block_sums_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                                 uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_wave[4];
    const int64_t i = (int64_t)blockIdx.x * kScanChunk + threadIdx.x;
    uint32_t s = (i < n) ? in[i] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
    if (lane_id() == 0) s_wave[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// single workgroup: exclusive scan of nb block sums, in place
__global__ __launch_bounds__(kBlock) void scan_spine_kernel(uint32_t* __restrict__ block_sums, int nb) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += kBlock) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t inc = block_inclusive_scan(v, s_wave, &total);
        if (i < nb) block_sums[i] = carry + inc - v;
        carry += total;
    }
}

__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ out, int64_t n,
                                                            const uint32_t* __restrict__ block_prefix,
                                                            uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_wave[4];
    const int64_t i = (int64_t)blockIdx.x * kScanChunk + threadIdx.x;
    const uint32_t v = (i < n) ? in[i] : 0u;
    const uint32_t inc = block_inclusive_scan(v, s_wave, nullptr) + block_prefix[blockIdx.x];
    if (i < n) out[i] = inc;
    if (total_out && i == n - 1) *total_out = inc;
}

size_t scan_scratch_bytes(int64_t n) {
    const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
    return (size_t)(nb > 0 ? nb : 1) * sizeof(uint32_t);
}

int launch_inclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_out, void* scratch,
                          hipStream_t stream) {
    if (n <= 0) return 0;
    uint32_t* block_sums = reinterpret_cast<uint32_t*>(scratch);
    const int nb = (int)((n + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(kBlock), 0, stream, in, n, block_sums);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(kBlock), 0, stream, block_sums, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kBlock), 0, stream, in, out, n, block_sums, total_out);
    return check_hip(hipGetLastError(), "scan");
}

// ===================================================================================================
// duplicateWithKeys — wave-cooperative expansion.
// A wave owns 64 consecutive Gaussians, whose instances occupy one contiguous output range
// [offsets[first]-count[first], offsets[last]).  The wave walks that range 64 slots at a time; each lane
// finds the owning Gaussian of its slot by binary search over the wave's 64 running ends (in LDS), so
// key/value stores are fully coalesced and a Gaussian covering hundreds of tiles does not serialise
// its lane.  Order = Gaussian id, then tile row, then tile column (matches the oracle).
// ===================================================================================================
__global__ __launch_bounds__(kBlock) void duplicate_keys_kernel(FrameDev f, const uint2* __restrict__ rects,
                                                                const uint32_t* __restrict__ depth_keys,
                                                                const uint32_t* __restrict__ point_offsets,
                                                                uint64_t* __restrict__ keys,
                                                                uint32_t* __restrict__ vals) {
    __shared__ uint32_t s_end[4][kWave];      // inclusive end, relative to the wave's base
    __shared__ uint32_t s_minx[4][kWave];
    __shared__ uint32_t s_miny[4][kWave];
    __shared__ uint32_t s_w[4][kWave];
    __shared__ uint32_t s_depth[4][kWave];
    const int w = wave_id();
    const int lane = lane_id();
    const int i = blockIdx.x * kBlock + threadIdx.x;

    uint32_t cnt = 0, end = 0, minx = 0, miny = 0, width = 1, dbits = 0;
    if (i < f.P) {
        end = point_offsets[i];
        const uint2 r = rects[i];
        const uint32_t wd = r.y & 0xFFFFu, ht = r.y >> 16;
        if (wd * ht > 0) {
            minx = r.x & 0xFFFFu; miny = r.x >> 16; width = wd;
            cnt = wd * ht;
            dbits = depth_keys[i];
        }
    }
    // lanes past P carry end = 0: give them the previous valid end so the ends stay monotone
    const uint32_t start = end - cnt;
    const uint32_t wave_base = __shfl(start, 0, kWave);
    // last valid lane's end
    const int last_valid = min(kWave - 1, f.P - 1 - (blockIdx.x * kBlock + w * kWave));
    if (last_valid < 0) return;                    // whole wave past P (wave-uniform)
    const uint32_t wave_end = __shfl(end, last_valid, kWave);
    const uint32_t total = wave_end - wave_base;

    s_end[w][lane] = (lane <= last_valid) ? (end - wave_base) : total;
    s_minx[w][lane] = minx; s_miny[w][lane] = miny; s_w[w][lane] = width; s_depth[w][lane] = dbits;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const int first_id = blockIdx.x * kBlock + w * kWave;
    for (uint32_t o = lane; o < total; o += kWave) {
        // smallest l with s_end[l] > o
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            if (s_end[w][lo + step - 1] <= o) lo += step;
        }
        const uint32_t owner_end = s_end[w][lo];
        // count of the owner = end - previous end
        const uint32_t prev_end = (lo > 0) ? s_end[w][lo - 1] : 0u;
        const uint32_t local = o - prev_end;
        (void)owner_end;
        const uint32_t wd = s_w[w][lo];
        const uint32_t ty = s_miny[w][lo] + local / wd;
        const uint32_t tx = s_minx[w][lo] + local % wd;
        const uint64_t key = ((uint64_t)(ty * (uint32_t)f.gx + tx) << 32) | (uint64_t)s_depth[w][lo];
        keys[(size_t)wave_base + o] = key;
        vals[(size_t)wave_base + o] = (uint32_t)(first_id + lo);
    }
}

int launch_duplicate_keys(const FrameDev& f, const uint32_t* rects, const uint32_t* depth_keys,
                          const uint32_t* point_offsets, uint64_t* keys, uint32_t* vals, hipStream_t stream) {
    const int blocks = (f.P + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3(blocks), dim3(kBlock), 0, stream, f,
                       reinterpret_cast<const uint2*>(rects), depth_keys, point_offsets, keys, vals);
    return check_hip(hipGetLastError(), "duplicate_keys_kernel");
}

// tiles touched per Gaussian (from the packed rectangles), in id order -> inclusive offsets
__global__ __launch_bounds__(kBlock) void rect_counts_kernel(const uint2* __restrict__ rects, int P,
                                                             uint32_t* __restrict__ counts) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < P) { const uint2 r = rects[i]; counts[i] = (r.y & 0xFFFFu) * (r.y >> 16); }
}

int launch_rect_counts_scan(const uint32_t* rects, int P, uint32_t* offsets_out, uint32_t* block_sums,
                            hipStream_t stream) {
    const int blocks = (P + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(rect_counts_kernel, dim3(blocks), dim3(kBlock), 0, stream,
                       reinterpret_cast<const uint2*>(rects), P, offsets_out);
    return launch_inclusive_scan(offsets_out, offsets_out, P, nullptr, block_sums, stream);
}

// single workgroup: total of nb block sums -> *total_out (num_rendered)
__global__ __launch_bounds__(kBlock) void total_kernel(const uint32_t* __restrict__ block_sums, int nb,
                                                       uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_wave[4];
    uint32_t s = 0;
    for (int i = threadIdx.x; i < nb; i += kBlock) s += block_sums[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
    if (lane_id() == 0) s_wave[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) *total_out = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

int launch_total_from_block_sums(uint32_t* block_sums, int nb, uint32_t* total_out, hipStream_t stream) {
    hipLaunchKernelGGL(total_kernel, dim3(1), dim3(kBlock), 0, stream, block_sums, nb, total_out);
    return check_hip(hipGetLastError(), "total_kernel");
}

// ===================================================================================================
// stable LSD radix sort, 8-bit digits.  Per pass:
//   sort_hist_kernel     per-workgroup digit histogram            -> counts[digit][block]
//   sort_rowscan_kernel  one workgroup per digit: exclusive scan of its row, row total
//   sort_scatter_kernel  stable ranking inside the workgroup (wave64 ballot match) + scatter
// A workgroup owns kSortItems*256 consecutive pairs; wave w owns a contiguous slice of them and walks it
// 64 pairs at a time, so (wave, step, lane) order == memory order and the ranking is stable.
// ===================================================================================================
constexpr int kSortItems = 8;                              // pairs per thread
constexpr int kSortTile = kSortItems * kBlock;             // 2048 pairs per workgroup
constexpr int kRadix = 256;

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, int shift) { return (uint32_t)(key >> shift) & 0xffu; }

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void sort_hist_kernel(const KeyT* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t* __restrict__ counts, int nblocks) {
    __shared__ uint32_t s_hist[kRadix];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const int64_t idx = base + j * kBlock + threadIdx.x;
        if (idx < n) atomicAdd(&s_hist[digit_of(keys[idx], shift)], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_hist[threadIdx.x];
}

// grid = 256 workgroups (one per digit).  In-place exclusive scan of counts[digit][0..nblocks),
// row total to totals[digit].
__global__ __launch_bounds__(kBlock) void sort_rowscan_kernel(uint32_t* __restrict__ counts, int nblocks,
                                                              uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[4];
    uint32_t* row = counts + (size_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += kBlock) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nblocks) ? row[i] : 0u;
        uint32_t total;
        const uint32_t inc = block_inclusive_scan(v, s_wave, &total);
        if (i < nblocks) row[i] = carry + inc - v;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// vals_in == nullptr: the value of pair i is i (first pass of an index sort).
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void sort_scatter_kernel(const KeyT* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              KeyT* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                              const uint32_t* __restrict__ counts, int nblocks,
                                                              const uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave_cnt[4][kRadix];   // running per-wave digit counters
    __shared__ uint32_t s_base[kRadix];          // global base of (digit, this block)
    __shared__ uint32_t s_scan[4];
    const int w = wave_id();
    const int lane = lane_id();
    const int t = threadIdx.x;

    // global digit bases: exclusive scan over the 256 row totals, plus this block's row offset
    {
        const uint32_t tot = totals[t];
        const uint32_t inc = block_inclusive_scan(tot, s_scan, nullptr);
        s_base[t] = inc - tot + counts[(size_t)t * nblocks + blockIdx.x];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_wave_cnt[k][t] = 0;
    __syncthreads();

    // wave w owns pairs [base + w*kSortItems*64, +kSortItems*64)
    const int64_t wbase = (int64_t)blockIdx.x * kSortTile + (int64_t)w * (kSortItems * kWave);
    KeyT key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const int64_t idx = wbase + j * kWave + lane;
        const bool valid = idx < n;
        key[j] = valid ? keys_in[idx] : (KeyT)~(KeyT)0;
        val[j] = valid ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;
    }
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const int64_t idx = wbase + j * kWave + lane;
        const bool valid = idx < n;
        const uint32_t d = digit_of(key[j], shift);
        // lanes of this wave holding the same digit (invalid lanes match nobody)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t vote = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? vote : ~vote;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t prior = valid ? s_wave_cnt[w][d] : 0u;
        rank[j] = prior + before;
        __builtin_amdgcn_wave_barrier();
        // the highest peer lane publishes the new running count
        if (valid && (peers >> lane) == 1ull) s_wave_cnt[w][d] = prior + before + 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // exclusive prefix of the per-wave totals across waves, per digit
    {
        const uint32_t c0 = s_wave_cnt[0][t], c1 = s_wave_cnt[1][t], c2 = s_wave_cnt[2][t];
        const uint32_t b = s_base[t];
        __syncthreads();
        s_wave_cnt[0][t] = b;
        s_wave_cnt[1][t] = b + c0;
        s_wave_cnt[2][t] = b + c0 + c1;
        s_wave_cnt[3][t] = b + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSortItems; ++j) {
        const int64_t idx = wbase + j * kWave + lane;
        if (idx < n) {
            const uint32_t d = digit_of(key[j], shift);
            const uint32_t dst = s_wave_cnt[w][d] + rank[j];
            keys_out[dst] = key[j];
            vals_out[dst] = val[j];
        }
    }
}

int sort_num_passes(int end_bit) { return (end_bit + 7) / 8; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t sort_scratch_bytes(int64_t n) {
    const int64_t nblocks = (n + kSortTile - 1) / kSortTile;
    return align_up((size_t)(nblocks > 0 ? nblocks : 1) * kRadix * sizeof(uint32_t), 256) + kRadix * sizeof(uint32_t);
}

template <typename KeyT>
static int sort_passes(const KeyT* k0, const uint32_t* v0, KeyT* ka, uint32_t* va, KeyT* kb, uint32_t* vb, int64_t n,
                       int end_bit, void* scratch, hipStream_t stream, bool first_out_is_b) {
    // pass 0 reads (k0, v0) [v0 may be nullptr = implicit iota]; outputs alternate between the a and b pairs
    const int nblocks = (int)((n + kSortTile - 1) / kSortTile);
    uint32_t* counts = reinterpret_cast<uint32_t*>(scratch);
    uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(scratch) +
                                                   align_up((size_t)nblocks * kRadix * sizeof(uint32_t), 256));
    const int passes = sort_num_passes(end_bit);
    const KeyT* kin = k0;
    const uint32_t* vin = v0;
    bool out_b = first_out_is_b;
    for (int p = 0; p < passes; ++p) {
        KeyT* kout = out_b ? kb : ka;
        uint32_t* vout = out_b ? vb : va;
        const int shift = 8 * p;
        hipLaunchKernelGGL(sort_hist_kernel<KeyT>, dim3(nblocks), dim3(kBlock), 0, stream, kin, n, shift, counts, nblocks);
        hipLaunchKernelGGL(sort_rowscan_kernel, dim3(kRadix), dim3(kBlock), 0, stream, counts, nblocks, totals);
        hipLaunchKernelGGL(sort_scatter_kernel<KeyT>, dim3(nblocks), dim3(kBlock), 0, stream, kin, vin, kout, vout, n,
                           shift, counts, nblocks, totals);
        kin = kout; vin = vout;
        out_b = !out_b;
    }
    return check_hip(hipGetLastError(), "sort");
}

// Sorts (keys_a, vals_a); the result lands in the `b` buffers when result_in_b, else in the `a` buffers (one copy
// pass is inserted when the pass count has the wrong parity for that).
int launch_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int64_t n,
                      int end_bit, void* scratch, hipStream_t stream, bool result_in_b) {
    if (n <= 0) return 0;
    const int passes = sort_num_passes(end_bit);
    int rc = sort_passes<uint64_t>(keys_a, vals_a, keys_a, vals_a, keys_b, vals_b, n, end_bit, scratch, stream,
                                   /*first_out_is_b=*/true);
    if (rc) return rc;
    const bool in_b = (passes & 1) != 0;
    if (in_b != result_in_b) {
        uint64_t* ksrc = in_b ? keys_b : keys_a; uint64_t* kdst = in_b ? keys_a : keys_b;
        uint32_t* vsrc = in_b ? vals_b : vals_a; uint32_t* vdst = in_b ? vals_a : vals_b;
        rc = check_hip(hipMemcpyAsync(kdst, ksrc, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream),
                       "sort copy keys");
        if (rc) return rc;
        rc = check_hip(hipMemcpyAsync(vdst, vsrc, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream),
                       "sort copy vals");
        if (rc) return rc;
    }
    return 0;
}

// ===================================================================================================
// identifyTileRanges: boundaries of equal key>>32 runs.  ranges must be zeroed by the caller.
// ===================================================================================================
__global__ __launch_bounds__(kBlock) void tile_ranges_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                             uint2* __restrict__ ranges) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t tile = (uint32_t)(keys[i] >> 32);
    if (i == 0) {
        ranges[tile].x = 0;
    } else {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (prev != tile) {
            ranges[prev].y = (uint32_t)i;
            ranges[tile].x = (uint32_t)i;
        }
    }
    if (i == n - 1) ranges[tile].y = (uint32_t)n;
}

// launch order = identity: slot i of the order array (see tile_order_slots) holds tile i, padding slots hold n_tiles
// ... and the (tile, quadrant) order of the blend backward behind it: entry j = 4 tile + quadrant = j, padding 4 n_tiles
__global__ __launch_bounds__(kBlock) void tile_order_identity_kernel(uint32_t* __restrict__ order, int slots, int n_tiles) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < slots) order[i] = (uint32_t)min(i, n_tiles);
    if (i < 4 * slots) order[slots + i] = (uint32_t)min(i, 4 * n_tiles);
}

int launch_tile_ranges(const uint64_t* keys_sorted, int64_t n, uint32_t* ranges, int n_tiles, hipStream_t stream) {
    int rc = check_hip(hipMemsetAsync(ranges, 0, (size_t)n_tiles * 2 * sizeof(uint32_t), stream), "ranges memset");
    if (rc) return rc;
    const int slots = tile_order_slots(n_tiles);
    hipLaunchKernelGGL(tile_order_identity_kernel, dim3((4 * slots + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       ranges + 2 * (size_t)n_tiles, slots, n_tiles);
    if (n <= 0) return check_hip(hipGetLastError(), "tile_order_identity_kernel");
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks), dim3(kBlock), 0, stream, keys_sorted, n,
                       reinterpret_cast<uint2*>(ranges));
    return check_hip(hipGetLastError(), "tile_ranges_kernel");
}

}  // namespace scg
