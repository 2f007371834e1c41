// tile_sort.h — per-tile sort of a list segment on (depth bits, Gaussian id) in LDS: the one-pass bucket sort (common case)
// with the LSD radix sort as the fallback for heavily tied depths.  Device code shared by the binning stage's sort kernels
// (binning_tiles.hip) and the forward blend that sorts its own tile on the way (blend.hip, tile_blend_forward_kernel).
#pragma once

#include "scg_common.h"

namespace scg {

// ---- small tiles: LSD radix sort in LDS -------------------------------------------------------------
// The bitonic network moves O(n log^2 n) keys through LDS and is LDS-bandwidth bound (measured: 135 us for the
// 8160 tiles of S3); an LSD radix sort moves 4 x n.  ITEMS keys per thread live in registers; wave w owns the
// contiguous index slice [w*ITEMS*64, (w+1)*ITEMS*64), so (wave, step, lane) order == index order and the
// wave64 ballot ranking is stable.  The sort key is the 32-bit depth; ties in depth must come out in ascending
// id, but the scatter order is arbitrary — so after the 4 depth passes the (rare) tiles that contain an
// out-of-order tie are redone with the id bits as additional leading passes.
constexpr int kRadixBins = 256;

// CNT: words of the counter array — as many buckets as list entries (the default) or fewer (denser buckets, less LDS: the
// forward blend that sorts its own tile wants every workgroup slot of the compute unit), never fewer than the radix
// fallback's NW * 256 counters.
// The ids come FIRST: the forward blend that sorts its own tile keeps the sorted list there (KEEP below) while its waves' record
// planes live in the bytes behind it (the rest of the sort's scratch, which that kernel pads to the planes' size).
template <int NW, int MAX_N, int CNT = MAX_N>
struct TileSortLds {
    __attribute__((aligned(16))) uint32_t id[MAX_N];
    // radix passes: cnt[w * 256 + digit]; bucket sort: cnt[bucket] + one end sentinel
    __attribute__((aligned(16))) uint32_t cnt[CNT + 4];
    uint32_t scan[NW];
    uint32_t red[2 * NW];
    uint32_t key[MAX_N];
#ifdef SCG_PROBE_TIMELINE
    uint32_t probe[8];
#endif
};

// One LSD pass over the workgroup's NW*64*ITEMS keys (NW waves; the first 256 threads own the 256 digits).
template <int NW, int MAX_N, int CNT, int ITEMS>
__device__ __forceinline__ void lds_radix_pass(TileSortLds<NW, MAX_N, CNT>& L, uint32_t (&key)[ITEMS], uint32_t (&id)[ITEMS],
                                               int shift, bool digit_from_id) {
    static_assert(NW * kRadixBins <= CNT + 4, "radix counters must fit the bucket array");
    const int w = wave_id(), lane = lane_id(), t = threadIdx.x;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
    for (int k = t; k < NW * kRadixBins; k += NW * kWave) L.cnt[k] = 0;
    __syncthreads();
    uint32_t rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = ((digit_from_id ? id[j] : key[j]) >> shift) & 0xffu;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t vote = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? vote : ~vote;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t prior = L.cnt[w * kRadixBins + d];
        rank[j] = prior + before;
        __builtin_amdgcn_wave_barrier();
        if ((peers >> lane) == 1ull) L.cnt[w * kRadixBins + d] = prior + before + 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // digit bases: exclusive scan of the digit totals (threads 0..255 = waves 0..3), then the per-wave prefixes
    uint32_t tot = 0, v = 0;
    if (t < kRadixBins) {
#pragma unroll
        for (int k = 0; k < NW; ++k) tot += L.cnt[k * kRadixBins + t];
        v = tot;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t n = __shfl_up(v, off, kWave);
            if (lane >= off) v += n;
        }
        if (lane == kWave - 1) L.scan[w] = v;
    }
    __syncthreads();
    if (t < kRadixBins) {
        uint32_t base = v - tot;
        for (int k = 0; k < w; ++k) base += L.scan[k];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const uint32_t c = L.cnt[k * kRadixBins + t];
            L.cnt[k * kRadixBins + t] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = ((digit_from_id ? id[j] : key[j]) >> shift) & 0xffu;
        const uint32_t dst = L.cnt[w * kRadixBins + d] + rank[j];
        L.key[dst] = key[j];
        L.id[dst] = id[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = w * (ITEMS * kWave) + j * kWave + lane;
        key[j] = L.key[idx];
        id[j] = L.id[idx];
    }
    __syncthreads();
}

// ---- the common case: one-pass bucket sort -----------------------------------------------------------------
// A tile's depths are spread out: with as many buckets as list entries, a monotone map key -> bucket
// (offset by the tile's minimum, scaled by its range) leaves one to three entries per bucket.  So: count per bucket
// with LDS atomics (the returned arrival number places the entry inside its bucket), exclusive scan of the counts,
// scatter into LDS, and every entry finds its final rank by comparing (depth, id) with the few entries of its own
// bucket — O(n) instead of four 8-ballot radix passes (the radix sort was 93 % VALU-bound: ~600 lane-instructions
// per entry).  The result is the total order on (depth, id), i.e. exactly what the stable sort produces.  Lists
// whose fullest bucket exceeds kBucketMax entries (heavily tied depths) go to the radix sort below.
constexpr int kBucketMax = 24;

// BPT buckets per thread: ITEMS (one bucket per possible entry) normally; fewer = denser buckets in a smaller counter array.
#ifdef SCG_PROBE_TIMELINE
#define SCG_TP(k) if (threadIdx.x == 0) L.probe[k] = (uint32_t)wall_clock64();
#else
#define SCG_TP(k)
#endif
// KEEP: the sorted ids also stay in L.id (the caller walks the list from LDS) and leave for `list` as coalesced stores nobody
// waits for, instead of one scattered 4-byte store per entry that the workgroup's next barrier has to see completed.
template <int NW, int MAX_N, int CNT, int ITEMS, int BPT, bool KEEP = false>
__device__ __forceinline__ bool sort_tile_bucket(TileSortLds<NW, MAX_N, CNT>& L, const uint32_t* __restrict__ depth_keys,
                                                 uint32_t* __restrict__ list, int n) {
    constexpr int T = NW * kWave;
    constexpr int B = BPT * T;                                 // buckets
    static_assert(B + 1 <= CNT + 4, "bucket counters + end sentinel must fit the counter array");
    const int w = wave_id(), lane = lane_id(), t = threadIdx.x;
    uint32_t key[ITEMS], id[ITEMS];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    // Every id of a group is requested before one of them is waited for, then every depth key (round 6).  The loads used to sit
    // inside `if (idx < n)`, one branch per entry: the compiler cannot move a load out of its branch, and the load counter retires
    // in order — waiting for entry j + 1's id also waited for entry j's key: 2 ITEMS dependent round trips per thread where two
    // are needed (the ISA showed load / wait / load / wait ...; S3 forward blend 158.0 -> 146.6 us, S2 76.0 -> 72.7, S4 103.8 ->
    // 93.8: profiles/README.md round 6).  Index-clamped loads need no branch: a thread's slots beyond the list re-read the last
    // entry (n >= 2 here), which changes neither minimum nor maximum, and everything per entry below is predicated on the slot
    // being real.  (In groups of at most four entries: seven gather addresses in flight at once spill registers in the dense
    // frames' kernel, which lives on 64; as it is, that kernel parks one base pointer in scratch: two reloads per tile.)
    constexpr int GROUP = ITEMS > 6 ? 4 : ITEMS;
#pragma unroll
    for (int j0 = 0; j0 < ITEMS; j0 += GROUP) {
#pragma unroll
        for (int j = j0; j < j0 + GROUP && j < ITEMS; ++j) id[j] = list[min(j * T + t, n - 1)];
#pragma unroll
        for (int j = j0; j < j0 + GROUP && j < ITEMS; ++j) key[j] = depth_keys[id[j]];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { kmin = min(kmin, key[j]); kmax = max(kmax, key[j]); }
#pragma unroll
    for (int j = 0; j < BPT; ++j) L.cnt[j * T + t] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, kWave));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, kWave));
    }
    if (lane == 0) { L.red[2 * w] = kmin; L.red[2 * w + 1] = kmax; }
    SCG_TP(0)
    __syncthreads();
    SCG_TP(1)
#pragma unroll
    for (int k = 0; k < NW; ++k) { kmin = min(kmin, L.red[2 * k]); kmax = max(kmax, L.red[2 * k + 1]); }
    // monotone map: (key - kmin) normalised to 32 bits, times B / 2^32
    const int sh = __builtin_clz((kmax - kmin) | 1u);
    // (an instantiation with sixteen items per thread — the dense frames' forward blend had one: four waves for up to 4 096
    // entries, until its sort got eight waves — recomputes the bucket where it is used instead of keeping sixteen more
    // registers alive: three instructions)
    constexpr bool kKeepBucket = ITEMS < 16;
    auto bucket_of = [&](uint32_t k) { return __umulhi((k - kmin) << sh, (uint32_t)B); };
    uint32_t bucket[kKeepBucket ? ITEMS : 1], arrival[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t b = bucket_of(key[j]);
        if (kKeepBucket) bucket[j] = b;
        if (j * T + t < n) arrival[j] = atomicAdd(&L.cnt[b], 1u);
    }
    SCG_TP(2)
    __syncthreads();
    SCG_TP(3)
    // exclusive scan of the B counts (thread t owns BPT consecutive buckets); fullest bucket
    uint32_t c[BPT], sum = 0, cmax = 0;
#pragma unroll
    for (int j = 0; j < BPT; ++j) { c[j] = L.cnt[t * BPT + j]; sum += c[j]; cmax = max(cmax, c[j]); }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, kWave);
        if (lane >= off) incl += up;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off, kWave));
    if (lane == kWave - 1) L.scan[w] = incl;
    if (lane == 0) L.red[w] = cmax;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int k = 0; k < w; ++k) base += L.scan[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) cmax = max(cmax, L.red[k]);
    if (cmax > (uint32_t)kBucketMax) return false;             // uniform: every thread sees the same maximum
#pragma unroll
    for (int j = 0; j < BPT; ++j) { L.cnt[t * BPT + j] = base; base += c[j]; }
    if (t == T - 1) L.cnt[B] = base;                            // = n: end of the last bucket
    __syncthreads();
    SCG_TP(4)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (j * T + t < n) {
            const uint32_t pos = L.cnt[kKeepBucket ? bucket[j] : bucket_of(key[j])] + arrival[j];
            L.key[pos] = key[j];
            L.id[pos] = id[j];
        }
    }
    __syncthreads();
    SCG_TP(5)
    uint32_t final_rank[KEEP ? ITEMS : 1];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (j * T + t < n) {
            const uint32_t bj = kKeepBucket ? bucket[j] : bucket_of(key[j]);
            const uint32_t s = L.cnt[bj], e = L.cnt[bj + 1];
            uint32_t rank = s;
            for (uint32_t p = s; p < e; ++p) {
                const uint32_t kk = L.key[p], ii = L.id[p];
                rank += ((kk < key[j]) || (kk == key[j] && ii < id[j])) ? 1u : 0u;
            }
            if (KEEP) final_rank[j] = rank;
            else list[rank] = id[j];
        }
    }
    if (KEEP) {
        __syncthreads();                                        // every bucket has been read: L.id may take the final order
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (j * T + t < n) L.id[final_rank[j]] = id[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (j * T + t < n) list[j * T + t] = L.id[j * T + t];
    }
    return true;
}

template <int NW, int MAX_N, int CNT, int ITEMS, bool KEEP = false>
__device__ __forceinline__ void sort_tile_radix(TileSortLds<NW, MAX_N, CNT>& L, const uint32_t* __restrict__ depth_keys,
                                                uint32_t* __restrict__ list, int n, int id_bits) {
    // one bucket per possible entry when the counter array has room for that, else half as many (two entries per bucket)
    // ... or as many as the counter array holds (the dense frames' 3 584-entry lists: three buckets per thread, 1 536 of 2 048 counters)
    constexpr int kHalf = ITEMS > 1 ? ITEMS / 2 : 1;
    constexpr int BPT = (ITEMS * NW * kWave + 1 <= CNT + 4) ? ITEMS : (kHalf * NW * kWave + 1 <= CNT + 4) ? kHalf : CNT / (NW * kWave);
    static_assert(BPT >= 1, "at least one bucket per thread");
    if (sort_tile_bucket<NW, MAX_N, CNT, ITEMS, BPT, KEEP>(L, depth_keys, list, n)) return;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    uint32_t key[ITEMS], id[ITEMS];
    auto load = [&]() {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int idx = w * (ITEMS * kWave) + j * kWave + lane;
            key[j] = 0xFFFFFFFFu; id[j] = 0xFFFFFFFFu;      // padding: larger than any real (depth, id)
            if (idx < n) { id[j] = list[idx]; key[j] = depth_keys[id[j]]; }
        }
    };
    load();
    for (int p = 0; p < 4; ++p) lds_radix_pass<NW, MAX_N, CNT, ITEMS>(L, key, id, 8 * p, false);
    // out-of-order tie?  (L.key / L.id hold the sorted sequence)
    bool bad = false;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = w * (ITEMS * kWave) + j * kWave + lane;
        if (idx > 0 && idx < n && L.key[idx - 1] == key[j] && L.id[idx - 1] > id[j]) bad = true;
    }
    if (__syncthreads_or(bad)) {
        load();                                              // LSD over (id bits, then depth bits)
        for (int sh = 0; sh < id_bits; sh += 8) lds_radix_pass<NW, MAX_N, CNT, ITEMS>(L, key, id, sh, true);
        for (int p = 0; p < 4; ++p) lds_radix_pass<NW, MAX_N, CNT, ITEMS>(L, key, id, 8 * p, false);
    }
    // (the last pass left the sorted sequence in L.key / L.id as well: a KEEP caller walks it from there)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int idx = w * (ITEMS * kWave) + j * kWave + lane;
        if (idx < n) list[idx] = id[j];
    }
}

// One workgroup of NW waves sorts one list of n <= MAX_N = NW*64*8 entries.  <4, 2048>: 20 KiB of LDS, 7 workgroups
// per CU — the common kernel; <8, 4096>: its dense-scene variant (49 KiB); <16, 8192>: the rare kernel's 16-wave sort
// (96 KiB).
template <int NW, int MAX_N, int CNT, bool KEEP = false>
__device__ __forceinline__ void sort_one_tile(TileSortLds<NW, MAX_N, CNT>& L, const uint2 r,
                                              const uint32_t* __restrict__ depth_keys,
                                              uint32_t* __restrict__ point_list, int id_bits) {
    const int n = (int)(r.y - r.x);
    uint32_t* list = point_list + r.x;
    constexpr int per = NW * kWave;
    static_assert(MAX_N % per == 0, "MAX_N must be a multiple of the workgroup size");
    if (n <= per) sort_tile_radix<NW, MAX_N, CNT, 1, KEEP>(L, depth_keys, list, n, id_bits);
    else if (n <= 2 * per) sort_tile_radix<NW, MAX_N, CNT, 2, KEEP>(L, depth_keys, list, n, id_bits);
    else if (n <= 4 * per && MAX_N >= 4 * per) sort_tile_radix<NW, MAX_N, CNT, 4, KEEP>(L, depth_keys, list, n, id_bits);
    else sort_tile_radix<NW, MAX_N, CNT, MAX_N / per, KEEP>(L, depth_keys, list, n, id_bits);
}


__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// In-place ascending sort of keys[0..n) with the bitonic network in its all-ascending form (each merge starts
// with a mirrored "flip" stage, then half-cleaners), so the array needs NO padding: a comparator whose upper
// index is >= n is a comparison with a virtual +inf and never moves anything.  `keys` may be LDS or global
// memory.  Stages spanning <= 128 keys stay inside one wave's window (consecutive threads own consecutive
// comparators): they are separated by wave barriers only.
template <typename Ptr>
__device__ __forceinline__ void bitonic_sort_asc(Ptr keys, int n, bool global_mem) {
    const int np = next_pow2(n);
    const int half = np >> 1;
    auto sync = [&](bool cross) {
        if (cross || global_mem) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    for (int k = 2; k <= np; k <<= 1) {
        // flip stage: element off of the lower half of each k-block against its mirror in the upper half
        {
            const int hk = k >> 1;
            for (int i = threadIdx.x; i < half; i += (int)blockDim.x) {
                const int blk = i / hk, off = i - blk * hk;
                const int a = blk * k + off;
                const int b = blk * k + (k - 1 - off);
                if (b < n) {
                    const uint64_t x = keys[a], y = keys[b];
                    if (x > y) { keys[a] = y; keys[b] = x; }
                }
            }
            // next stage: j = k/4 (span k/2), or the next flip (span 2k) when k == 2
            const int next_span = (k >= 4) ? (k >> 1) : (k << 1);
            sync(k > 2 * kWave || next_span > 2 * kWave);
        }
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < half; i += (int)blockDim.x) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a + j;
                if (b < n) {
                    const uint64_t x = keys[a], y = keys[b];
                    if (x > y) { keys[a] = y; keys[b] = x; }
                }
            }
            const int next_span = (j > 1) ? j : (k << 1);   // next half-cleaner spans 2*(j/2) = j; else next flip
            sync(2 * j > 2 * kWave || next_span > 2 * kWave);
        }
    }
}


// ---- long lists: the same one-pass bucket sort with the entries in global memory --------------------------------
// A tile behind a dense cluster can hold tens of thousands of entries (real captures do this; the synthetic uniform
// scenes do not).  One workgroup, kLongBuckets buckets counted in LDS, the 64-bit (depth, id) composites in two
// global scratch copies: A = list order, B = bucket order.  Four passes of n items over the workgroup's threads — O(n), where the
// bitonic network this replaces is O(n log^2 n) compare-exchanges through global memory (1.5 ms for a 45 k list).
// Returns false (nothing written to `list`) when a bucket exceeds kLongBucketMax entries: heavily tied depths, left
// to the bitonic network.
// T threads, NB buckets (a multiple of T): <1024, 8192> in the rare-size kernel (64 KiB of LDS), <256, 1024> in the forward
// blend that sorts its own tiles (8.2 KiB: lists beyond its LDS sort, walked right afterwards by the same workgroup).
constexpr int kLongBucketMax = 1024;     // ranking is O(bucket size) per entry: beyond this the depths are too tied

template <int T, int NB, int BATCH = 8>
__device__ __forceinline__ bool sort_long_list(unsigned char* smem, const uint32_t* __restrict__ depth_keys,
                                               uint32_t* __restrict__ list, int n, uint64_t* __restrict__ A,
                                               uint64_t* __restrict__ B) {
    constexpr int kLongBuckets = NB;
    static_assert(NB % T == 0, "buckets per thread must be whole");
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);                      // [kLongBuckets + 1] counts -> starts
    uint32_t* cur = cnt + kLongBuckets + 4;                                  // [kLongBuckets] running cursors
    uint32_t* red = cur + kLongBuckets;                                      // [2 * 16] reductions
    const int t = threadIdx.x, w = wave_id(), lane = lane_id();
    constexpr int NW = T / kWave;
    // depth range of the list from a SAMPLE of 1024 entries (one per thread): the map key -> bucket only has to be
    // monotone, keys outside the sampled range are clamped into the first / last bucket — so the gather, the composite
    // copy A and the histogram are ONE pass over the list instead of two
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    {
        const int i = (int)(((int64_t)t * n) / T);
        const uint32_t key = depth_keys[list[i]];
        kmin = key; kmax = key;
    }
    for (int b = t; b < kLongBuckets; b += T) cnt[b] = 0u;
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
        const uint32_t clamped = min(max(key, kmin), kmax);
        return __umulhi((clamped - kmin) << sh, (uint32_t)kLongBuckets);
    };
    // (one workgroup per list: every pass keeps several independent memory operations in flight per thread)
    // (BATCH entries per thread in flight: the ids, then their keys — two dependent gathers —, then the stores; eight in the
    //  16-wave kernels, fewer where the registers belong to somebody else: the fallbacks of the 8-wave sort and of the blend)
    for (int i0 = t; i0 < n; i0 += BATCH * T) {
        uint32_t id[BATCH], key[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) id[u] = list[min(i0 + u * T, n - 1)];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) key[u] = depth_keys[id[u]];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            if (i0 + u * T < n) {
                const uint64_t comp = ((uint64_t)key[u] << 32) | (uint64_t)id[u];
                A[i0 + u * T] = comp;
                atomicAdd(&cnt[bucket_of(comp)], 1u);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the counts (8 consecutive buckets per thread) + fullest bucket
    constexpr int PER = kLongBuckets / T;
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
    __syncthreads();                                               // red[] is reused
    if (lane == kWave - 1) red[w] = incl;
    if (lane == 0) red[NW + w] = cmax;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int k = 0; k < w; ++k) base += red[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) cmax = max(cmax, red[NW + k]);
    if (cmax > (uint32_t)kLongBucketMax) return false;             // uniform
#pragma unroll
    for (int j = 0; j < PER; ++j) { cnt[t * PER + j] = base; cur[t * PER + j] = base; base += c[j]; }
    if (t == T - 1) cnt[kLongBuckets] = base;                       // = n
    __syncthreads();
    for (int i0 = t; i0 < n; i0 += BATCH * T) {
        uint64_t comp[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) comp[u] = A[min(i0 + u * T, n - 1)];
#pragma unroll
        for (int u = 0; u < BATCH; ++u)
            if (i0 + u * T < n) B[atomicAdd(&cur[bucket_of(comp[u])], 1u)] = comp[u];
    }
    __threadfence_block();
    __syncthreads();
#pragma unroll 2
    for (int i = t; i < n; i += T) {                               // i = position in bucket order
        const uint64_t comp = B[i];
        const uint32_t b = bucket_of(comp);
        const uint32_t s0 = cnt[b], e0 = cnt[b + 1];
        uint32_t rank = s0;
        for (uint32_t p = s0; p < e0; p += 4) {                    // four peers per trip, loads independent
            const uint64_t p0 = B[p], p1 = B[min(p + 1, e0 - 1)], p2 = B[min(p + 2, e0 - 1)], p3 = B[min(p + 3, e0 - 1)];
            rank += (p0 < comp) ? 1u : 0u;
            rank += (p + 1 < e0 && p1 < comp) ? 1u : 0u;
            rank += (p + 2 < e0 && p2 < comp) ? 1u : 0u;
            rank += (p + 3 < e0 && p3 < comp) ? 1u : 0u;
        }
        list[rank] = (uint32_t)comp;
    }
    return true;
}

}  // namespace scg
