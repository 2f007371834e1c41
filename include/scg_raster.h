/* scg_raster.h — C ABI of the MI355X-native differentiable Gaussian rasterizer (libscg_raster.so).
 *
 * This is the drop-in boundary for the reference's rasterizer extension: the Python package
 * `diff_gaussian_rasterization` imported at reference gaussian_renderer/__init__.py:15 (upstream: a
 * torch C++/CUDA extension exposing rasterize_gaussians / rasterize_gaussians_backward, absent from
 * /root/reference — README.md:23).  Each entry point below names the reference call site whose work
 * it performs.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / ATen types cross this boundary.
 *   - Every buffer (inputs, outputs, saved state, scratch) is OWNED BY THE CALLER and lives in device
 *     memory unless stated otherwise.  The library never allocates or frees device memory and keeps no
 *     device-resident state between calls: it is re-entrant per device / stream.  The only process-wide state is
 *     host-side and write-once: a per-device record behind std::call_once (that device's compute-unit count and
 *     "the dynamic-LDS attributes of the big-LDS kernels were set", csrc/binning_tiles.hip device_setup) and the
 *     thread-local message of scg_last_error().
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*); no entry point synchronises the
 *     host.  The one host read the path needs (num_rendered, to size the binning buffers) is made by
 *     the caller after scg_geometry_forward — see `num_rendered_out`.
 *   - Return value: 0 = ok; < 0 = invalid argument (SCG_E_*); > 0 = a hipError_t.  Nothing throws.
 *     scg_last_error() returns a thread-local message for the last non-zero return.
 *   - All floats are fp32, ids int32/uint32, sort keys uint64.  Pointers must be 16-byte aligned where
 *     a record is a multiple of 16 bytes (splats, dsplats, rotations); torch allocations satisfy this.
 *   - Matrices are the reference's already-transposed row-vector matrices (scene/cameras.py:60-62):
 *     the flat 16-float buffer is read as [x y z 1] . M.
 */
#ifndef SCG_RASTER_H
#define SCG_RASTER_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared in the headers under include/ are exported. */
#ifndef SCG_API
#define SCG_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define SCG_TILE 16                 /* 16x16-pixel tiles (BASELINE.json north_star) */
#define SCG_SPLAT_FLOATS 12         /* per-Gaussian screen-space record: 3 x float4 */
#define SCG_DSPLAT_FLOATS 16        /* per-Gaussian gradient record: 64 bytes, 64-byte aligned (see below) */
#define SCG_ABI_VERSION 10

enum {
    SCG_OK = 0,
    SCG_E_NULL = -1,        /* required pointer is NULL */
    SCG_E_RANGE = -2,       /* size / degree / dimension out of range */
    SCG_E_EXCLUSIVE = -3,   /* must pass exactly one of (shs | colors_precomp) and of (scales+rotations | cov3D_precomp) */
    SCG_E_SCRATCH = -4,     /* scratch buffer too small */
    SCG_E_ALIGN = -5        /* pointer not sufficiently aligned */
};

/* Per-view constants: the scalar/tensor fields of GaussianRasterizationSettings
 * (reference gaussian_renderer/__init__.py:38-51) plus the Gaussian count and SH record size. */
typedef struct ScgFrame {
    int32_t P;               /* number of Gaussians */
    int32_t sh_degree;       /* active degree 0..3            (:47 sh_degree) */
    int32_t sh_coeffs;       /* M: coefficients per record; shs is (P, M, 3), M >= (sh_degree+1)^2 */
    int32_t width;           /* image_width                   (:40) */
    int32_t height;          /* image_height                  (:39) */
    float tanfovx;           /*                               (:41) */
    float tanfovy;           /*                               (:42) */
    float scale_modifier;    /*                               (:44) */
    int32_t prefiltered;     /* accepted, ignored (reference passes False, :49) */
    int32_t debug;           /* (:50) accepted by the library; the BINDING implements upstream's behaviour: on a failing call it
                              * saves the call's arguments (snapshot_fw.dump / snapshot_bw.dump) and re-raises */
    const float* viewmatrix; /* device, 16 floats              (:45) */
    const float* projmatrix; /* device, 16 floats              (:46) */
    const float* campos;     /* device, 3 floats               (:48) */
    const float* bg;         /* device, 3 floats               (:43) */
    /* Launch-order hint of the blend kernels (ABI 5; both optional, never enter a result).  The blend forward records in
     * tile_cost_out[tile] what the tile cost (blended list entries of its busiest 8x8 quadrant); a caller that renders the
     * same camera again — a training loop does, thousands of times — hands those Tn = ceil(W/16)*ceil(H/16) words back as
     * tile_cost_in, and the tiles that were expensive start first, so the launch drains faster.  NULL in: order by list
     * length.  NULL out: nothing is recorded.  The two must not alias. */
    const uint32_t* tile_cost_in;
    uint32_t* tile_cost_out;
    /* ABI 8, optional: TWO words of HOST-VISIBLE (pinned) memory that scg_forward's forward blend overwrites with [0] the
     * number of tiles whose list is longer than it sorts itself in LDS (SCG_FUSED_MAX_LIST entries) and [1] the number whose
     * list exceeds 16 384 entries.  A caller that renders the same camera again reads them (no synchronisation: they hold the
     * counts of the latest COMPLETED render): while [0] is 0 it passes SCG_FORWARD_SKIP_RARE_SORT, while [1] is not 0
     * SCG_FORWARD_RARE_8WAVE | SCG_FORWARD_SPLIT_LONG_LISTS.  NULL: nothing is recorded. */
    uint32_t* long_lists_out;
    /* ABI 9, both optional, never enter a result: launch order of the blend BACKWARD at quadrant granularity.  Every (tile, 8x8
     * quadrant) wave of scg_blend_backward / scg_backward records its work (list entries that passed its cull) in
     * bwd_cost_out[4 tile + quadrant] — 4 Tn words; a caller that renders the same camera again hands them back as bwd_cost_in, and
     * the binning stage orders each XCD band's quadrants by decreasing work for the next backward (a round-5 replay of the measured wave lives: the backward's
     * makespan on the chip's 7 168 wave slots is 102 us in tile order, 84 us longest-first).  NULL in: the quadrants follow the
     * tiles' order.  NULL out: nothing is recorded.  The two must not alias. */
    const uint32_t* bwd_cost_in;
    uint32_t* bwd_cost_out;
    /* ABI 10, optional: ONE word at any device-accessible address (device memory, or pinned host memory) that the binning
     * stage overwrites with num_rendered as IT counted it (the total of the tile lists before they were clipped to the
     * capacity).  A caller that must not read the host inside a step — a step captured in a hipGraph — launches scg_forward
     * with a capacity of its own choosing, never calls scg_wait_num_rendered, and compares this word with the capacity when it
     * next looks (after the step, or at the next render of the camera): larger = the lists of that render were clipped, its
     * images and gradients are incomplete, run it again with room for this count.  NULL: nothing is written. */
    uint32_t* num_rendered_out;
} ScgFrame;

/* Layout of one splat record (SCG_SPLAT_FLOATS floats, 48 bytes), written by scg_geometry_forward and
 * gathered by the blend kernels:
 *   [0] x_pix   [1] y_pix    [2] conic_a   [3] conic_b
 *   [4] conic_c [5] opacity  [6] cull_thr  [7] cull_slope
 *   [8] r       [9] g        [10] b        [11] depth (view z)
 * (everything a pixel needs to decide whether the splat contributes sits in the first 6 floats; colour + depth
 * are read by contributing lanes only.  cull_thr = 2 ln(255 opacity) * 1.001 + 0.01 is the largest value of the
 * conic's quadratic form at which alpha can still reach 1/255, cull_slope = -conic_b / conic_c; the two only
 * steer the blend kernels' conservative per-quadrant culling and never enter a blended value.)
 * The per-Gaussian gradient record `dsplats` written by scg_blend_backward (SCG_DSPLAT_FLOATS floats = 64 bytes; the
 * buffer must be 64-byte aligned) holds RAW SUMS over the pixels that blended the Gaussian, not derivatives (ABI >= 5):
 * with q = opacity * G * dL/dalpha of a pixel and (dx, dy) = splat centre - pixel,
 *   [0] sum q dx   [1] sum q dy   [2] dL/ddepth   [3] sum q       |
 *   [4] sum q dx^2 [5] sum q dx dy [6] sum q dy^2 [7] -           |
 *   [8] dL/dr      [9] dL/dg      [10] dL/db      [11..15] -
 * One record = one 64-byte line (ABI >= 7): the float atomics of the blend backward are executed at the memory side of
 * the fabric, one transaction per line touched, and a 48-byte record at a 48-byte stride straddles two lines every
 * other time (measured: blend backward 143 -> 130 us at 200 k Gaussians @ 1008x756, profiles/README.md round 3).
 * scg_geometry_backward applies the conic map and the constant factors once per Gaussian
 * (dL/dx_pix = -(a S_x + b S_y), dL/dy_pix = -(b S_x + c S_y), dL/dconic_a = -S_xx / 2, dL/dconic_b = -S_xy,
 *  dL/dconic_c = -S_yy / 2, dL/dopacity = S_q / opacity; csrc/geometry.hip is the single source for these).  A consumer
 * of the staged API that wants derivatives must apply the same map. */

SCG_API const char* scg_last_error(void);
SCG_API int32_t scg_abi_version(void);
/* sizeof(ScgFrame) / sizeof(ScgWorkspaceLayout) / sizeof(ScgStageEvents) as this library was compiled: a binding that
 * declares the structs itself (ctypes, cgo, JNA) compares them with its own before the first call. */
SCG_API size_t scg_struct_bytes(int32_t which /* 0 ScgFrame, 1 ScgWorkspaceLayout, 2 ScgStageEvents, 3 ScgModel, 4 ScgModelGrads */);

/* ---- stage 1: per-Gaussian geometry (replaces the preprocess step of upstream rasterize_gaussians;
 *      inputs as passed at reference gaussian_renderer/__init__.py:100-108) ---------------------------
 * Frustum cull (view z <= 0.2), projection, 3D covariance from scale/rotation (or cov3D_precomp),
 * EWA 2D covariance + conic + radius + tile rectangle, SH -> RGB (or colors_precomp), and the total number
 * of tile instances.
 *   means3D (P,3)  opacities (P)  shs (P,M,3)|NULL  colors_precomp (P,3)|NULL
 *   scales (P,3)+rotations (P,4) | cov3D_precomp (P,6)
 * Outputs: splats (P,12)  radii (P) int32  clamped (P) uint8 bit c set when channel c was clamped at 0
 *          rects (P,2) uint32: {min_x | min_y << 16, width | height << 16} of the touched tile rectangle
 *                (0,0 for culled Gaussians; width*height = tiles touched)
 *          depth_keys (P) uint32: float bits of the view-space depth, 0xFFFFFFFF for culled Gaussians
 *          num_rendered_out: 1 uint32, any device-accessible address (device or pinned host memory), or NULL;
 *          receives R = sum of tiles touched.  The caller reads it (after synchronising `stream`) to size
 *          point_list and the binning scratch.
 * scratch: scg_geometry_scratch_bytes(P) bytes, any device-accessible address.  On return its first
 *          ceil(P/256) uint32 hold per-workgroup partial sums of the tiles touched (their total is R): a caller
 *          that passes pinned host memory here and num_rendered_out = NULL gets R on the host with no extra
 *          kernel and no copy (one wait on `stream`, one host-side sum). */
SCG_API size_t scg_geometry_scratch_bytes(int32_t P);
SCG_API int scg_geometry_forward(const ScgFrame* frame,
                         const float* means3D, const float* opacities,
                         const float* shs, const float* colors_precomp,
                         const float* scales, const float* rotations, const float* cov3D_precomp,
                         float* splats, int32_t* radii, uint8_t* clamped,
                         uint32_t* rects, uint32_t* depth_keys, uint32_t* num_rendered_out,
                         void* scratch, size_t scratch_bytes, void* stream);

/* ---- stage 2: tile binning (duplicateWithKeys + radix sort + identifyTileRanges of upstream) --------
 * Produces the reference's result — for every tile the list of Gaussians touching it, ordered by depth
 * with ties in ascending Gaussian id, i.e. the stable sort of (key = tile<<32 | float_bits(depth), id) —
 * without materialising the R 64-bit keys (SCG_BINNING_AUTO): the instances are generated from the tile
 * rectangles, counted and scattered by tile (tile-first), and every tile's list is then sorted on (depth, id) in
 * LDS.  When the tile count does not fit the LDS histograms (or on request, SCG_BINNING_GLOBAL_SORT) the
 * reference's scheme — duplicate, global 64-bit radix sort, range detection — is used; both give identical outputs.
 *   num_rendered: R as read from num_rendered_out — or, on the SCG_BINNING_AUTO path when
 *          scg_binning_accepts_bound() says so, any UPPER BOUND of it (the capacity of point_list): this lets the
 *          caller enqueue stages 2-3 without waiting for the host read of num_rendered_out.  If the bound turns
 *          out smaller than R, nothing is written out of bounds and the ranges are clipped to it (the images are
 *          then wrong): the caller must compare with num_rendered_out afterwards and run stages 2-3 again.
 * Outputs: point_list (R) uint32 sorted Gaussian ids;
 *          ranges: scg_ranges_words(width, height) uint32 = the (tiles,2) tile ranges (untouched tiles: 0,0) followed
 *          by the LAUNCH ORDER of the tiles for the blend kernels (8 bands of ceil(tiles/8) slots, one band per XCD,
 *          longest lists first; a permutation of the tiles padded with `tiles`) and by the launch order of the blend
 *          backward's (tile, quadrant) waves (4 entries per slot: 4 tile + quadrant, padded with 4 tiles) — scheduling
 *          only, never a result
 *          keys_sorted (R) uint64 or NULL (debug / parity tests: the sorted 64-bit keys)
 * scratch: scg_binning_scratch_bytes(P, R, width, height, algo) bytes. */
enum { SCG_BINNING_AUTO = 0, SCG_BINNING_GLOBAL_SORT = 1 };
SCG_API size_t scg_binning_scratch_bytes(int32_t P, int64_t num_rendered, int32_t width, int32_t height, int32_t algo);
SCG_API size_t scg_ranges_words(int32_t width, int32_t height);      /* uint32 words the `ranges` buffer must hold */
/* 1 when scg_binning(..., algo) for this image size / bound runs the tile-first path, which accepts an upper bound
 * for num_rendered; 0 when it runs the global sort, which needs the exact value. */
SCG_API int32_t scg_binning_accepts_bound(int64_t num_rendered_bound, int32_t width, int32_t height, int32_t algo);
SCG_API int scg_binning(const ScgFrame* frame, int64_t num_rendered,
                const uint32_t* rects, const uint32_t* depth_keys,
                uint32_t* point_list, uint32_t* ranges, uint64_t* keys_sorted, int32_t algo,
                void* scratch, size_t scratch_bytes, void* stream);

/* Stable LSD radix sort of (uint64 key, uint32 value) pairs on key bits [0, end_bit).  Exposed for the
 * parity tests ("bit-exact sort indices").  On return the sorted pairs are in keys_out / vals_out;
 * keys_in / vals_in are clobbered.  scratch: scg_sort_scratch_bytes(n). */
SCG_API size_t scg_sort_scratch_bytes(int64_t n);
SCG_API int scg_sort_pairs(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                   int64_t n, int32_t end_bit, void* scratch, size_t scratch_bytes, void* stream);

/* Inclusive prefix sum of n uint32 (in may equal out).  total_out (optional) receives the last element.
 * scratch: scg_scan_scratch_bytes(n). */
SCG_API size_t scg_scan_scratch_bytes(int64_t n);
SCG_API int scg_inclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_out,
                           void* scratch, size_t scratch_bytes, void* stream);

/* ---- stage 3: 16x16-tile forward alpha blend (upstream render forward; outputs unpacked at
 *      reference gaussian_renderer/__init__.py:100) --------------------------------------------------
 * Front-to-back over each tile's sorted list: colour (3,H,W) incl. background, depth (1,H,W) = expected
 * view z (un-normalised), alpha (1,H,W) = 1 - T_final; plus the per-pixel state the backward needs:
 * final_T (H,W) float, n_contrib (H,W) uint32 (list index + 1 of the last blended Gaussian). */
SCG_API int scg_blend_forward(const ScgFrame* frame, const uint32_t* ranges, const uint32_t* point_list,
                      const float* splats,
                      float* out_color, float* out_depth, float* out_alpha,
                      float* final_T, uint32_t* n_contrib,
                      float* dsplats_zero /* NULL, or the (P,16) gradient record buffer of the coming backward: cleared here */,
                      void* stream);

/* ---- stage 4: per-pixel backward (upstream render backward; autograd hands over dL/dcolor,
 *      dL/ddepth, dL/dalpha — reference train.py:170, scene/gaussian_model.py:259-280, train.py:168) ---
 * Back-to-front replay per 8x8 quadrant of a tile (one wave each); the ten partial gradients of a Gaussian are
 * reduced across the 64 lanes of the wave and added to its gradient record with one hardware float atomic
 * instruction per Gaussian per quadrant.
 * dL_dcolor (3,H,W), dL_ddepth (1,H,W)|NULL, dL_dalpha (1,H,W)|NULL.
 * Output: dsplats (P,16), zero-initialised by this call unless dsplats_prezeroed != 0 (the buffer was handed to
 *         scg_blend_forward as dsplats_zero and not touched since), then accumulated. */
SCG_API int scg_blend_backward(const ScgFrame* frame, const uint32_t* ranges, const uint32_t* point_list,
                       const float* splats, const float* final_T, const uint32_t* n_contrib,
                       const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                       float* dsplats, int32_t dsplats_prezeroed, void* stream);

/* ---- stage 5: per-Gaussian geometry backward (upstream preprocess backward) ------------------------
 * Chain rule from dsplats to the inputs of stage 1.  Any output pointer that does not apply to the
 * chosen input path must be NULL (dL_dshs with colors_precomp, dL_dscales/rotations with cov3D_precomp
 * and vice versa).  dL_dmeans2D is (P,3): xy in NDC units (pixel gradient x 0.5*W, 0.5*H), z = 0 — the
 * slot consumed at reference scene/gaussian_model.py:932-934.  All outputs are fully written (zeros for
 * culled Gaussians and for SH coefficients above the active degree).
 * accumulate != 0 (ABI 6): every parameter gradient is ADDED to what its output buffer already holds instead of
 * overwriting it — the second, third ... view of the same Gaussians inside one training step (BASELINE cfg5: K views
 * per rank per step, one gradient exchange per K views) costs no separate add pass over 236 bytes per Gaussian.
 * dL_dmeans2D belongs to the view and is always overwritten. */
SCG_API int scg_geometry_backward(const ScgFrame* frame,
                          const float* means3D, const float* opacities,
                          const float* shs, const float* colors_precomp,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          const int32_t* radii, const uint8_t* clamped, const float* dsplats,
                          float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities,
                          float* dL_dshs, float* dL_dcolors_precomp,
                          float* dL_dscales, float* dL_drotations, float* dL_dcov3D_precomp,
                          int32_t accumulate /* SCG_BACKWARD_* bits; 1 = accumulate (ABI 6 callers pass 0 or 1) */, void* stream);
enum { SCG_BACKWARD_ACCUMULATE = 1,
       /* ABI 10: the caller promises that the SH-coefficient gradients ABOVE the active degree already hold zeros in the output
        * buffer(s) — a gradient arena the caller owns and keeps from step to step, written only by these kernels.  The kernel then
        * leaves them alone instead of storing zeros over zeros: at degree 0 that is 180 of the 192 bytes of every record (the
        * reference trains its first 1 000 iterations at degree 0, the next 1 000 at degree 1: train.py:129). */
       SCG_BACKWARD_SH_TAIL_ZERO = 2 };

/* ---- the whole path in ONE call per direction (the fast path of the Python binding) ------------------------------
 * The five stages above stay available (parity tests drive them one by one); a training step, though, is bound by
 * host time once the scene is small (the reference starts from <= 12 k Gaussians), so the binding normally issues
 * exactly two calls per step: scg_forward (stages 1-3) and scg_backward (stages 4-5).
 *
 * All internal state of a forward — splats, rects, depth keys, clamp flags, point_list, ranges (+ launch order),
 * final_T, n_contrib and the binning scratch — lives in ONE caller-owned `workspace` whose layout the library
 * reports; the same workspace is handed to scg_backward, which reads the saved state from it.
 *
 * `capacity` is an UPPER BOUND of num_rendered chosen by the caller (the size point_list is laid out for): the
 * stages are enqueued back to back without a host read.  The geometry stage leaves ceil(P/256) per-workgroup partial
 * sums of num_rendered in `partial_sums` (any device-accessible address: the binding passes pinned host memory) and,
 * if `event` (a hipEvent_t) is given, records it right behind that stage; scg_wait_num_rendered() blocks on the event
 * and returns the total.  If it exceeds `capacity` the images of this call are incomplete (nothing was written out
 * of bounds, the lists were clipped): the caller calls scg_forward again with capacity >= that total.
 * Only the tile-first binning path accepts a bound: check scg_binning_accepts_bound(capacity, w, h, SCG_BINNING_AUTO).
 */
typedef struct ScgWorkspaceLayout {      /* byte offsets into `workspace`; all 256-byte aligned */
    uint64_t splats;        /* (P,12) float  */
    uint64_t rects;         /* (P,2) uint32  */
    uint64_t depth_keys;    /* (P) uint32    */
    uint64_t clamped;       /* (P) uint8     */
    uint64_t point_list;    /* (capacity) uint32 */
    uint64_t ranges;        /* scg_ranges_words(w,h) uint32 */
    uint64_t final_T;       /* (H,W) float   */
    uint64_t n_contrib;     /* (H,W) uint32  */
    uint64_t bin_scratch;   /* scg_binning_scratch_bytes(...) */
    uint64_t total;         /* bytes the workspace must hold */
    uint64_t partial_words; /* uint32 words `partial_sums` must hold */
} ScgWorkspaceLayout;
SCG_API int scg_workspace_layout(int32_t P, int64_t capacity, int32_t width, int32_t height, ScgWorkspaceLayout* out);

/* Optional per-stage timing of the one-call entry points: hipEvent_t pairs (timing enabled) recorded on `stream`
 * right before / after a stage's launches; NULL entries are skipped, a NULL struct costs nothing.
 * scg_forward: [0] geometry  [1] binning  [2] blend forward.   scg_backward: [0] blend backward  [1] geometry backward.
 * scg_backward's pair [0] rides on the blend-backward kernel's own dispatch packet (the kernel's begin / end time stamps, no
 * barrier packets in the stream: timing the dominant kernel does not stretch the step); the clearing of a gradient-record
 * buffer that was not pre-zeroed runs in front of that interval. */
typedef struct ScgStageEvents {
    void* begin[3];
    void* end[3];
} ScgStageEvents;

/* scg_forward leaves the per-tile sort to the forward blend (ABI 6): one workgroup of four quadrant waves per tile sorts the
 * tile's list segment in LDS, writes the canonical order to point_list, and blends — one launch and its drain less than sort
 * kernel + blend kernel, the latency-bound sort hidden behind other tiles' blending; its geometry kernel (ABI 7) builds the binning
 * stage's slice histograms where it produces the tile rectangles.  Options a caller may pass (bits 0 and 1 are reserved for the
 * library's own A/B tests of those two fusions — csrc/scg_debug.h, not part of this interface): */
enum {
       /* the render will not be differentiated: final_T / n_contrib, the backward's per-pixel state (8 of the 28 bytes a pixel
        * costs), are not written.  scg_backward on such a workspace reads uninitialised state — the library cannot tell (the
        * workspace is caller memory it keeps no record of); the Python binding remembers the option and raises instead. */
       SCG_FORWARD_NO_BACKWARD_STATE = 4,
       /* the caller expects no list longer than SCG_FUSED_MAX_LIST entries (ScgFrame.long_lists_out of this camera's previous
        * render said 0): the rare-size sort kernel — an idle 4 us launch in such frames — is not launched.  Only a promise
        * about SPEED: a list that is longer after all is sorted by the forward blend's own workgroup through global scratch
        * (same result, slower), and long_lists_out tells the caller to drop the option at the next render. */
       SCG_FORWARD_SKIP_RARE_SORT = 8,
       /* the camera's previous render had VERY long lists (long_lists_out[1] > 0: beyond 16 384 entries, tiles behind a dense
        * cluster).  One 16-wave workgroup per such list is the critical path of the whole stage; instead every list beyond
        * 4 096 entries is partitioned by depth first (SPLIT_LONG_LISTS: one more launch) and the parts, like all the other
        * lists beyond SCG_FUSED_MAX_LIST entries, are sorted by 8-wave workgroups, three per compute unit (RARE_8WAVE).  The
        * two go together.  Same result whatever the options; a wrong expectation costs time only. */
       SCG_FORWARD_RARE_8WAVE = 16,
       SCG_FORWARD_SPLIT_LONG_LISTS = 32,
       /* ABI 9: `partial_sums` is HOST memory the device can write (pinned, host-coherent: what hipHostMalloc returns by default)
        * and the caller will collect num_rendered WITHOUT an event: scg_forward fills the ceil(P/256) words with
        * SCG_PARTIAL_SUM_ARMED on the host before it launches anything, and scg_wait_num_rendered(NULL, ...) watches the words
        * arrive (every one is written by exactly one workgroup of the geometry kernel).  The event record cost the queue a
        * barrier packet behind the geometry kernel (~6 us of idle GPU per forward in a rocprofv3 trace of the training step) and
        * the host an event wake-up; pass event = NULL with this option. */
       SCG_FORWARD_ARM_PARTIAL_SUMS = 64 };
#define SCG_PARTIAL_SUM_ARMED 0xFFFFFFFFu   /* no sum of tiles touched of 256 Gaussians reaches this */
#define SCG_FUSED_MAX_LIST 1536     /* list entries the sorting forward blend takes in LDS (3 584 in dense frames: an average
                                     * of 1 100 or more entries per tile) */
/* 1 when scg_forward with this capacity / image size / options sorts inside the forward blend (always, unless the reserved A/B
 * bit asks for the separate kernels): where the sort's time and bytes are accounted. */
SCG_API int32_t scg_forward_sorts_in_blend(int64_t capacity, int32_t width, int32_t height, int32_t options);
SCG_API int scg_forward(const ScgFrame* frame,
                const float* means3D, const float* opacities,
                const float* shs, const float* colors_precomp,
                const float* scales, const float* rotations, const float* cov3D_precomp,
                int64_t capacity, void* workspace, size_t workspace_bytes,
                int32_t* radii, float* out_color, float* out_depth, float* out_alpha,
                uint32_t* partial_sums, void* event,
                float* dsplats_zero /* NULL, or the (P,16) gradient records of the coming backward: cleared here */,
                int32_t options /* 0, or SCG_FORWARD_* bits */,
                const ScgStageEvents* stage_events, void* stream);

/* Blocks until `event` has completed — or, with event = NULL, until none of the ceil(P/256) words holds
 * SCG_PARTIAL_SUM_ARMED any more (scg_forward's SCG_FORWARD_ARM_PARTIAL_SUMS; words that were never armed are taken as they
 * are) — then returns the sum of the partial sums (= num_rendered); < 0 on error (see scg_last_error; the event-less wait polls
 * busily for the first ~100 000 polls, then yields the core between polls, and gives up after 120 s — the caller must synchronise
 * the stream before it releases the call's buffers then: the kernels may still be writing them).  `partial_sums_host` must be
 * host-readable (pinned) memory. */
SCG_API int64_t scg_wait_num_rendered(void* event, const uint32_t* partial_sums_host, int32_t P);

/* hipEvent_t helpers (a binding without its own HIP bindings needs nothing else): timing = 0 for the `event` above,
 * 1 for ScgStageEvents entries; elapsed time between two completed timing events in milliseconds. */
SCG_API int scg_event_create(void** event_out, int32_t timing);
SCG_API int scg_event_destroy(void* event);
SCG_API int scg_event_elapsed_ms(void* begin, void* end, float* ms_out);

SCG_API int scg_backward(const ScgFrame* frame,
                 const float* means3D, const float* opacities,
                 const float* shs, const float* colors_precomp,
                 const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii, int64_t capacity, const void* workspace,
                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                 float* dsplats, int32_t dsplats_prezeroed,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities,
                 float* dL_dshs, float* dL_dcolors_precomp,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D_precomp,
                 int32_t accumulate /* SCG_BACKWARD_* bits, see scg_geometry_backward */,
                 const ScgStageEvents* stage_events, void* stream);

/* ---- ABI 10: the path as the reference's MODEL hands it over (reference scene/gaussian_model.py:105-152, 452-468, 491-509) ----
 * The operator above takes ACTIVATED tensors; the reference's render() produces them on every call with ~25 small torch
 * launches (and as many in backward): get_xyz = cat(rayo + rayd * zval, bg_xyz) three times, get_features = cat(cat(f_dc,
 * bg_f_dc), cat(f_rest, bg_f_rest), dim 1) — a 192 B / Gaussian copy whose backward slices the SH gradient twice more —,
 * sigmoid / exp / normalize + cat for opacity / scaling / rotation.  These entry points read the model's RAW parameter tensors
 * where they lie and write the gradients of the RAW parameters: the activations, their derivatives and the concatenations
 * happen in registers of the two geometry kernels.
 *
 * A model is two SETS of Gaussians, indexed one behind the other (Gaussian i < ray.count is ray-bound, the rest background):
 *   ray-bound   position = rayo + rayd * zval (zval trainable; :126-131)     free (background)   position = xyz (trainable)
 * both with  features_dc (n,1,3) | features_rest (n,15,3)  raw SH, coefficient-major          (:134-142)
 *            opacity (n,1)   logit:  sigmoid            (:144-152)
 *            scaling (n,3)   log:    exp                (:105-113)
 *            rotation (n,4)  raw quaternion (r,x,y,z): q / max(|q|, 1e-12)   (:115-124, torch.nn.functional.normalize)
 * Either set may be empty (count 0, pointers ignored).  ScgFrame.P must equal ray.count + bg.count, sh_coeffs 16. */
typedef struct ScgModelSet {
    int32_t count;
    const float* zval;            /* (n,1)   ray-bound set; NULL in a free set */
    const float* rayo;            /* (n,3)   "  */
    const float* rayd;            /* (n,3)   "  */
    const float* xyz;             /* (n,3)   free set; NULL in a ray-bound set */
    const float* features_dc;     /* (n,1,3) */
    const float* features_rest;   /* (n,15,3), 16-byte aligned */
    const float* opacity;         /* (n,1) */
    const float* scaling;         /* (n,3) */
    const float* rotation;        /* (n,4), 16-byte aligned */
} ScgModelSet;
typedef struct ScgModel { ScgModelSet ray, bg; } ScgModel;

/* Gradient buffers of one set, same shapes as the parameters (zval for a ray-bound set, xyz for a free one).  All of them are
 * fully written (zeros for culled Gaussians) unless SCG_BACKWARD_ACCUMULATE / SCG_BACKWARD_SH_TAIL_ZERO say otherwise. */
typedef struct ScgModelGradSet {
    float* zval;
    float* xyz;
    float* features_dc;
    float* features_rest;         /* 16-byte aligned */
    float* opacity;
    float* scaling;
    float* rotation;              /* 16-byte aligned */
} ScgModelGradSet;
typedef struct ScgModelGrads { ScgModelGradSet ray, bg; } ScgModelGrads;

/* The model's activated getters in ONE launch (get_xyz, get_opacity, get_scaling, get_rotation of reference
 * scene/gaussian_model.py:105-152), computed by the very device functions the two geometry kernels use: what a caller needs
 * them for outside the render (densification reads get_xyz / get_scaling), and what the parity tests feed the oracle with
 * (the rasterizer's integer outputs are then bit-exact against it; the activations themselves are compared with torch's).
 * Any output may be NULL.  means3D (P,3), opacities (P,1), scales (P,3), rotations (P,4). */
SCG_API int scg_model_activate(const ScgModel* model, float* means3D, float* opacities, float* scales, float* rotations,
                               void* stream);

/* scg_forward / scg_backward with the model in place of the seven input tensors (SH colours, scale + rotation covariance: the
 * reference's default pipe, gaussian_renderer/__init__.py:64-68, 78-85).  Everything else — workspace, capacity, options,
 * num_rendered, stage events — as there; struct index 3 / 4 of scg_struct_bytes = ScgModel / ScgModelGrads. */
SCG_API int scg_forward_model(const ScgFrame* frame, const ScgModel* model,
                              int64_t capacity, void* workspace, size_t workspace_bytes,
                              int32_t* radii, float* out_color, float* out_depth, float* out_alpha,
                              uint32_t* partial_sums, void* event, float* dsplats_zero, int32_t options,
                              const ScgStageEvents* stage_events, void* stream);
SCG_API int scg_backward_model(const ScgFrame* frame, const ScgModel* model,
                               const int32_t* radii, int64_t capacity, const void* workspace,
                               const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                               float* dsplats, int32_t dsplats_prezeroed,
                               const ScgModelGrads* grads, float* dL_dmeans2D,
                               int32_t flags /* SCG_BACKWARD_* */, const ScgStageEvents* stage_events, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCG_RASTER_H */
