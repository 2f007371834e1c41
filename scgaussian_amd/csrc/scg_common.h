// scg_common.h — internal declarations shared by the HIP translation units of libscg_raster.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
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
};

// Unit table of the segmented blend backward (blend.hip).  The dominant kernel of a training step walks every quadrant's list
// back to front; as ONE work item per quadrant the items are tens of microseconds long and the launch ends in a long drain.
// So the list is cut into SEGMENTS of kSeg entries and a work item (one single-wave workgroup) walks one segment:
//   * the binning stage (tile_start_kernel) writes the table: for tile t with n list entries, 4 quadrants x ceil(n / kSeg)
//     descriptors {tile, quadrant | segment << 2} at slots [ubase(t), ubase(t+1)), ubase(t) = 4 (range.x / kSeg + t) — the
//     slots a tile does not need (<= 8) hold an "unused" marker; ubase is monotone, so XCD band x (tiles [x per, (x+1) per))
//     owns the contiguous region [ubase(x per), ubase((x+1) per)), whose bounds sit behind the launch order in `ranges`;
//   * the forward blend stores every pixel's accumulators (r, g, b, depth, T) at each kSeg-th list entry it walks past (a
//     CHECKPOINT: five planes of 64 floats per quadrant; slot of tile t, boundary i = 1, 2, ...: range.x / kSeg + i - 1 —
//     tiles never collide: floor(x1 / kSeg) - floor(x0 / kSeg) >= floor((n - 1) / kSeg)) and, at its end, how far the
//     quadrant's pixels got (qlimit[4 tile + quadrant] = index behind the last entry any of them blended);
//   * backward workgroup b serves band b % 8 (the XCD it runs on, whose L2 holds the band's splat records since the
//     forward), unit b / 8 of the band's region: a unit whose segment lies behind qlimit retires at once, the others walk
//     <= kSeg entries starting from the checkpoint behind them.
// No atomics anywhere: a counter shared by the workgroups of a launch costs ~0.3 us per increment on this part (measured:
// queue tickets from 8 counters made the kernel 9x slower), the hardware's own workgroup dispatch is the dynamic scheduler.
constexpr int kSegChunks = 2;                           // 64-entry chunks per backward unit
constexpr int kSeg = kSegChunks * kWave;                // list entries per unit = distance of the forward's checkpoints
constexpr int kCkptFloats = 4 * 5 * kWave;              // per slot: 4 quadrants x {r, g, b, depth, T} x 64 pixels
constexpr uint32_t kUnitUnused = 0xFFFFFFFFu;
constexpr int kTailBandStart = 0;                       // [0..8]   first unit slot of each XCD band's region, [8] = end
constexpr int kTailValid = 24;                          // [24]     1 when the table is valid (tile-first binning ran)
constexpr int kTailWords = 32;

struct BwdQueue {
    uint2* units;                   // bwd_units_capacity(capacity, n_tiles) descriptors
    uint32_t* qlimit;               // 4 n_tiles
    float* ckpt;                    // bwd_ckpt_slots(capacity) x kCkptFloats
};
inline size_t bwd_units_capacity(int64_t capacity, int n_tiles) { return 4 * ((size_t)capacity / kSeg + (size_t)n_tiles + 1); }
inline size_t bwd_ckpt_slots(int64_t capacity) { return (size_t)capacity / kSeg + 1; }
inline BwdQueue no_bwd_queue() { BwdQueue q; q.units = nullptr; q.qlimit = nullptr; q.ckpt = nullptr; return q; }

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
    d.cost_in = f->tile_cost_in; d.cost_out = f->tile_cost_out;
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
                             float* dcolors, float* dscales, float* drots, float* dcov3D, bool accumulate,
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
    size_t table, tile_total, tile_start, class_counts, mid_tiles, big_tiles, len_hist, tile_class, spill, total;
    int nblocks;
};
int tile_binning_blocks(int64_t R);
bool tile_binning_supported(int n_tiles, int64_t R);
TileBinningLayout tile_binning_layout(int P, int64_t R, int n_tiles);
int launch_tile_binning(const FrameDev& f, int64_t R, const uint32_t* rects, const uint32_t* depth_keys,
                        uint32_t* point_list, uint32_t* ranges, uint64_t* keys_sorted, void* scratch,
                        uint2* bwd_units /* unit table of the segmented blend backward, or nullptr */, hipStream_t stream);
int launch_tile_ranges(const uint64_t* keys_sorted, int64_t n, uint32_t* ranges, int n_tiles, hipStream_t stream);

// bq.units == nullptr: no checkpoints, no queue (render only / whole-list backward)
int launch_blend_forward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                         const float* splats, float* out_color, float* out_depth, float* out_alpha,
                         float* final_T, uint32_t* n_contrib, float* dsplats_zero, const BwdQueue& bq,
                         hipStream_t stream);
// bq.units != nullptr (the forward filled it): segmented backward from the queue; needs the forward's colour / depth images
int launch_blend_backward(const FrameDev& f, const uint32_t* ranges, const uint32_t* point_list,
                          const float* splats, const float* final_T, const uint32_t* n_contrib,
                          const float* out_color, const float* out_depth,
                          const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                          float* dsplats, bool dsplats_prezeroed, const BwdQueue& bq, int64_t capacity,
                          hipStream_t stream);

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
// the kTailWords control words of the backward's work queue behind the launch order
__host__ __device__ __forceinline__ size_t ranges_tail_offset(int n_tiles) { return 2 * (size_t)n_tiles + (size_t)tile_order_slots(n_tiles); }

__device__ __forceinline__ int xcd_tile_remap(int b, int n_tiles) {
    const int xcd = b & 7;
    const int slot = b >> 3;
    const int per = (n_tiles + 7) >> 3;        // tiles per XCD band
    const int t = xcd * per + slot;
    return t;                                   // may be >= n_tiles for the padded tail: caller checks
}
#endif

}  // namespace scg
