// scg_common.h — internal declarations shared by the HIP translation units of libscg_raster.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/scg_raster.h"

namespace scg {

constexpr int kWave = 64;
constexpr int kTile = SCG_TILE;            // 16
constexpr int kTilePix = kTile * kTile;    // 256
constexpr int kBlock = 256;                // 4 waves per workgroup

constexpr float kNearZ = 0.2f;             // view-space cull plane
constexpr float kLowpass = 0.3f;           // px^2 added to the 2D covariance diagonal
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 1e-4f;

// Kernel-argument copy of ScgFrame plus host-derived scalars.  The fp32 host scalars are computed in
// exactly this form (oracle/torch_rasterizer.py host_scalars mirrors it):
//   focal_x = (float)W / (2.0f * tanfovx);  limx = 1.3f * tanfovx
struct FrameDev {
    int P, D, M, W, H;
    int gx, gy;                 // tile grid
    float tanfovx, tanfovy, focal_x, focal_y, limx, limy, mod;
    const float* view;
    const float* proj;
    const float* campos;
    const float* bg;
    const uint32_t* cost_in;    // launch-order hint (previous render of this camera) or nullptr
    uint32_t* cost_out;         // receives this render's per-tile cost, or nullptr
    uint32_t* long_out;         // two host-visible words: tiles with more than kFusedMaxN / more than 16 384 entries, or nullptr
    const uint32_t* bcost_in;   // per (tile, quadrant): ticks the blend backward's wave ran in the camera's previous step, or nullptr
    uint32_t* bcost_out;        // receives this step's, or nullptr
    uint32_t* nr_out;           // one device-accessible word that receives num_rendered as the binning stage counted it, or nullptr
};

inline FrameDev make_frame_dev(const ScgFrame* f) {
    FrameDev d;
    d.P = f->P; d.D = f->sh_degree; d.M = f->sh_coeffs; d.W = f->width; d.H = f->height;
    d.gx = (f->width + kTile - 1) / kTile;
    d.gy = (f->height + kTile - 1) / kTile;
    d.tanfovx = f->tanfovx; d.tanfovy = f->tanfovy;
    d.focal_x = (float)f->width / (2.0f * f->tanfovx);
    d.focal_y = (float)f->height / (2.0f * f->tanfovy);
    d.limx = 1.3f * f->tanfovx;
    d.limy = 1.3f * f->tanfovy;
    d.mod = f->scale_modifier;
    d.view = f->viewmatrix; d.proj = f->projmatrix; d.campos = f->campos; d.bg = f->bg;
    d.cost_in = f->tile_cost_in; d.cost_out = f->tile_cost_out; d.long_out = f->long_lists_out;
    d.bcost_in = f->bwd_cost_in; d.bcost_out = f->bwd_cost_out;
    d.nr_out = f->num_rendered_out;
    return d;
}

// error plumbing (api.hip)
int fail(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int validate_frame(const ScgFrame* f, bool need_bg);

// stage launchers (each returns 0 or an error code; all work goes on `stream`)
int launch_geometry_forward(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                            const float* colors_precomp, const float* scales, const float* rotations,
                            const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped,
                            uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, hipStream_t stream);
int launch_geometry_backward(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations,
                             const float* cov3D_precomp, const int32_t* radii, const uint8_t* clamped,
                             const float* dsplats, float* dmeans3D, float* dmeans2D, float* dopac, float* dshs,
                             float* dcolors, float* dscales, float* drots, float* dcov3D, int flags /* SCG_BACKWARD_* */,
                             hipStream_t stream);
// the model path (ScgModel: the reference model's raw parameter tensors; geometry.hip ModelSource)
int launch_geometry_forward_model(const FrameDev& f, const ScgModel& m, float* splats, int32_t* radii, uint8_t* clamped,
                                  uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, hipStream_t stream);
int launch_geometry_hist_model(const FrameDev& f, const ScgModel& m, float* splats, int32_t* radii, uint8_t* clamped,
                               uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, int nblocks, uint32_t* table,
                               uint32_t* class_counts, uint32_t* len_hist, hipStream_t stream);
int launch_geometry_hist_binned_model(const FrameDev& f, int64_t R, const ScgModel& m, float* splats, int32_t* radii,
                                      uint8_t* clamped, uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums,
                                      void* bin_scratch, hipStream_t stream);
hipError_t geometry_hist_model_set_max_lds(int bytes);
int launch_geometry_backward_model(const FrameDev& f, const ScgModel& m, const ScgModelGrads& g, const int32_t* radii,
                                   const uint8_t* clamped, const float* dsplats, float* dmeans2D, int flags, hipStream_t stream);
int launch_model_activate(const ScgModel& m, float* means3D, float* opacities, float* scales, float* rotations,
                          hipStream_t stream);

size_t scan_scratch_bytes(int64_t n);
int launch_inclusive_scan(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_out, void* scratch,
                          hipStream_t stream);

size_t sort_scratch_bytes(int64_t n);
int launch_sort_pairs(uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int64_t n,
                      int end_bit, void* scratch, hipStream_t stream, bool result_in_b);
int sort_num_passes(int end_bit);

int launch_duplicate_keys(const FrameDev& f, const uint32_t* rects, const uint32_t* depth_keys,
                          const uint32_t* point_offsets, uint64_t* keys, uint32_t* vals, hipStream_t stream);
int launch_rect_counts_scan(const uint32_t* rects, int P, uint32_t* offsets_out, uint32_t* block_sums,
                            hipStream_t stream);
int launch_total_from_block_sums(uint32_t* block_sums, int nb, uint32_t* total_out, hipStream_t stream);

// depth-first tile binning (binning_tiles.hip)
struct TileBinningLayout {
    size_t table, tile_total, tile_start, class_counts, mid_tiles, big_tiles, len_hist, tile_class, tile_part, spill, segments, total;
    int nblocks;
};
int tile_binning_blocks(int P, int64_t R);
bool tile_binning_supported(int n_tiles, int64_t R);
TileBinningLayout tile_binning_layout(int P, int64_t R, int n_tiles);
// defer_sort (in/out): the caller's forward blend sorts the tiles itself (launch_tile_blend_forward) — the common per-tile
// sort kernel is then not launched and lists longer than kFusedMaxN go to the rare-size kernel; cleared when the stage
// sorted everything after all (dense scenes: their 8-wave sort stays a kernel of its own).
constexpr int kFusedMaxN = SCG_FUSED_MAX_LIST;   // list entries the sorting forward blend takes (1 536)
constexpr int kFusedLongBuckets = 1024;     // buckets of its global-memory fallback sort of longer lists (8.2 KiB of the same LDS)
constexpr int kFusedCounters = 1024;        // ... with this many bucket counters (= the radix fallback's 4 x 256): 16.2 KiB of sort arrays, 18.2 KiB with the record planes behind the ids
// dense frames (an average list of kDenseMeanList entries or more: a million Gaussians on a small image) — round 5: the
// forward blend sorts their tiles too, with room for 3 584 entries (36.4 KiB of LDS: four workgroups per compute unit) and
// EIGHT waves per workgroup: all eight sort (7 entries per thread at a full list), the upper four leave behind the sort's
// last barrier and the four quadrant waves blend.  Two rounds of workgroups at the S4 scene either way (2 040 tiles on 1 024
// slots); what a round costs is the sort's two dependent gathers (ids, then a depth key per id), and twice the threads have
// twice as many in flight: S4 blend 111.0 -> 104.0 us same process (profiles/r05_ab_dense_8wave_sort.txt).  2 048 counters:
// the radix fallback's 8 x 256.
constexpr int kFusedDenseMaxN = 3584;
constexpr int kFusedDenseCounters = 2048;
constexpr int kFusedDenseWaves = 8, kFusedDenseWpe = 8;
constexpr int kDenseMeanList = 1100;        // average list length (capacity / tiles) from which a frame counts as dense
__host__ __device__ inline int fused_max_list(int64_t R, int n_tiles) {
    return (R / (n_tiles > 0 ? n_tiles : 1) >= kDenseMeanList) ? kFusedDenseMaxN : kFusedMaxN;
}
// hist_done: the slice histograms (table[B][Tn], slices = block_slice of tile_walk.h) were built by
// launch_geometry_hist on the same scratch — the stage starts at the column scan.
// skip_rare: with a deferred sort, do not launch the rare-size kernel either (SCG_FORWARD_SKIP_RARE_SORT: the forward blend's
// fallback takes a long list that shows up after all).  rare8: sort the rarer list sizes with 8-wave workgroups, three per
// compute unit (SCG_FORWARD_RARE_8WAVE); split_long: with rare8, partition the lists beyond 4 096 entries by depth first
// (tile_split_long_kernel: their parts are sorted by all compute units; SCG_FORWARD_SPLIT_LONG_LISTS).
int launch_tile_binning(const FrameDev& f, int64_t R, const uint32_t* rects, const uint32_t* depth_keys,
                        uint32_t* point_list, uint32_t* ranges, uint64_t* keys_sorted, void* scratch,
                        bool* defer_sort, bool hist_done, bool skip_rare, bool rare8, bool split_long, hipStream_t stream);
// geometry_forward + the tile histogram of the tile-first binning in one kernel (the one-call path).  `bin_scratch`, R as
// for launch_tile_binning, which must follow with hist_done = true.  Returns > 0 (a code of fail()) on error.
bool tile_binning_hist_in_geometry(const FrameDev& f, int64_t R);
int launch_geometry_hist_binned(const FrameDev& f, int64_t R, const float* means3D, const float* opacities, const float* shs,
                                const float* colors_precomp, const float* scales, const float* rotations,
                                const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped, uint32_t* rects,
                                uint32_t* depth_keys, uint32_t* block_sums, void* bin_scratch, hipStream_t stream);
int launch_geometry_hist(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                         const float* colors_precomp, const float* scales, const float* rotations,
                         const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped, uint32_t* rects,
                         uint32_t* depth_keys, uint32_t* block_sums, int nblocks, uint32_t* table, uint32_t* class_counts,
                         uint32_t* len_hist, hipStream_t stream);
hipError_t geometry_hist_set_max_lds(int bytes);      // dynamic-LDS ceiling of every geometry_hist_kernel instantiation
bool tile_binning_defers_sort(int64_t R, int n_tiles);      // what launch_tile_binning answers to *defer_sort = true
// bin_scratch, R: the binning stage's scratch and the capacity it was laid out for (the fallback sort's spill copies and the
// count of long lists live there); long_presorted: the rare-size kernel ran in front of this launch.
int launch_tile_blend_forward(const FrameDev& f, const uint32_t* ranges, uint32_t* point_list, const uint32_t* depth_keys,
                              const float* splats, float* out_color, float* out_depth, float* out_alpha,
                              float* final_T, uint32_t* n_contrib, float* dsplats_zero, void* bin_scratch, int64_t R,
                              bool long_presorted, hipStream_t stream);
int launch_tile_ranges(const uint64_t* keys_sorted, int64_t n, uint32_t* ranges, int n_tiles, hipStream_t stream);

int launch_blend_forward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                         const float* splats, float* out_color, float* out_depth, float* out_alpha,
                         float* final_T, uint32_t* n_contrib, float* dsplats_zero, hipStream_t stream);
int launch_blend_backward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                          const float* splats, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                          float* dsplats, bool dsplats_prezeroed, hipStream_t stream, hipEvent_t started = nullptr,
                          hipEvent_t done = nullptr);
// started / done: events attached to the kernel's OWN dispatch packet (hipExtLaunchKernelGGL): they carry the kernel's begin /
// end time stamps and complete with it, and — unlike hipEventRecord between two launches, which costs the queue a barrier
// packet of its own (~5.8 us of idle GPU per record in a rocprofv3 trace of the training step) — leave the stream's packets
// back to back: bench.py's live timing of the dominant kernel no longer stretches the step it measures (S2, 200 steps with
// the two events per step: 0.2556 -> 0.2505 ms).  Not used for the forward's num_rendered event: there the attached form saves
// the GPU 1 us and costs the host 2 (S1, host-bound: 0.1344 -> 0.1366 ms per step).

// ---- device helpers -------------------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Tile rectangle of a splat: [min, max) in tile units; decision D4 of the oracle (saturate in float,
// then cast).  Must stay a chain of single fp32 operations (bit-exact tile assignment).
__device__ __forceinline__ void tile_rect(float px, float py, float radius, int gx, int gy, int& minx,
                                          int& miny, int& maxx, int& maxy) {
    minx = (int)fminf(fmaxf(truncf((px - radius) * 0.0625f), 0.0f), (float)gx);
    maxx = (int)fminf(fmaxf(truncf((px + radius + 15.0f) * 0.0625f), 0.0f), (float)gx);
    miny = (int)fminf(fmaxf(truncf((py - radius) * 0.0625f), 0.0f), (float)gy);
    maxy = (int)fminf(fmaxf(truncf((py + radius + 15.0f) * 0.0625f), 0.0f), (float)gy);
}

// XCD-aware tile mapping: workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md), each XCD has
// a private 4 MiB L2.  Give each XCD a contiguous band of tile rows so that the splat records shared by
// neighbouring tiles stay in one L2.  Pure permutation of [0, n_tiles): placement changes speed only.
// The blend kernels take their tiles from a LAUNCH ORDER array stored behind the (tiles, 2) ranges: 8 bands of
// `per` = ceil(tiles / 8) slots, band x = the tiles [x per, (x+1) per) of XCD x, each band ordered by decreasing list
// length by the binning stage (longest lists start first: the short ones fill the tail of the launch); padding
// slots hold n_tiles.  Pure permutation: placement and order change speed only.
__host__ __device__ __forceinline__ int tile_order_slots(int n_tiles) { return ((n_tiles + 7) >> 3) << 3; }

__device__ __forceinline__ int xcd_tile_remap(int b, int n_tiles) {
    const int xcd = b & 7;
    const int slot = b >> 3;
    const int per = (n_tiles + 7) >> 3;        // tiles per XCD band
    const int t = xcd * per + slot;
    return t;                                   // may be >= n_tiles for the padded tail: caller checks
}
#endif

}  // namespace scg
