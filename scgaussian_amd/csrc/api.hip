// api.hip — the extern "C" surface declared in include/scg_raster.h: argument validation, error
// reporting, scratch carving, and the per-stage launch sequences.  No device memory is allocated or freed
// here and no entry point synchronises the host.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "scg_common.h"
#include "scg_debug.h"

#include <atomic>
#include <chrono>
#include <thread>

namespace scg {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
}

int validate_frame(const ScgFrame* f, bool need_bg) {
    if (!f) return fail(SCG_E_NULL, "frame is NULL");
    if (f->P < 0) return fail(SCG_E_RANGE, "P = %d < 0", f->P);
    if (f->width <= 0 || f->height <= 0 || f->width > 65535 * SCG_TILE || f->height > 65535 * SCG_TILE)
        return fail(SCG_E_RANGE, "image size %d x %d out of range", f->width, f->height);
    if (f->sh_degree < 0 || f->sh_degree > 3) return fail(SCG_E_RANGE, "sh_degree %d not in 0..3", f->sh_degree);
    if (!(f->tanfovx > 0.f) || !(f->tanfovy > 0.f)) return fail(SCG_E_RANGE, "tanfov must be positive");
    if (!f->viewmatrix || !f->projmatrix || !f->campos) return fail(SCG_E_NULL, "viewmatrix/projmatrix/campos is NULL");
    if (need_bg && !f->bg) return fail(SCG_E_NULL, "bg is NULL");
    return 0;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned64(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 63u) == 0; }
int validate_model(const ScgFrame* f, const ScgModel* m);

static int validate_inputs(const ScgFrame* f, const float* means3D, const float* opacities, const float* shs,
                           const float* colors_precomp, const float* scales, const float* rotations,
                           const float* cov3D_precomp) {
    if (f->P == 0) return 0;
    if (!means3D || !opacities) return fail(SCG_E_NULL, "means3D/opacities is NULL");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(SCG_E_EXCLUSIVE, "Please provide exactly one of either SHs or precomputed colors!");
    const bool has_sr = scales != nullptr && rotations != nullptr;
    if ((scales != nullptr) != (rotations != nullptr) || has_sr == (cov3D_precomp != nullptr))
        return fail(SCG_E_EXCLUSIVE,
                    "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (shs && f->sh_coeffs < (f->sh_degree + 1) * (f->sh_degree + 1))
        return fail(SCG_E_RANGE, "sh_coeffs %d < (sh_degree+1)^2 = %d", f->sh_coeffs,
                    (f->sh_degree + 1) * (f->sh_degree + 1));
    if (rotations && !aligned16(rotations)) return fail(SCG_E_ALIGN, "rotations must be 16-byte aligned");
    return 0;
}

static int validate_model_set(const ScgModelSet& m, bool ray, const char* name) {
    if (m.count < 0) return fail(SCG_E_RANGE, "%s.count = %d < 0", name, m.count);
    if (m.count == 0) return 0;
    if (ray ? (!m.zval || !m.rayo || !m.rayd) : !m.xyz)
        return fail(SCG_E_NULL, ray ? "%s: zval / rayo / rayd is NULL" : "%s: xyz is NULL", name);
    if (!m.features_dc || !m.features_rest || !m.opacity || !m.scaling || !m.rotation)
        return fail(SCG_E_NULL, "%s: features_dc / features_rest / opacity / scaling / rotation is NULL", name);
    if (!aligned16(m.rotation) || !aligned16(m.features_rest) || !aligned16(m.features_dc))
        return fail(SCG_E_ALIGN, "%s: rotation / features_rest / features_dc must be 16-byte aligned", name);
    return 0;
}

int validate_model(const ScgFrame* f, const ScgModel* m) {
    int rc = validate_model_set(m->ray, true, "model.ray");
    if (rc) return rc;
    if ((rc = validate_model_set(m->bg, false, "model.bg"))) return rc;
    if ((int64_t)m->ray.count + m->bg.count != f->P)
        return fail(SCG_E_RANGE, "model holds %d + %d Gaussians, frame.P = %d", m->ray.count, m->bg.count, f->P);
    if (f->P > 0 && f->sh_coeffs != 16)
        return fail(SCG_E_RANGE, "the model path reads (n,1,3) + (n,15,3) SH records: frame.sh_coeffs must be 16, not %d", f->sh_coeffs);
    return 0;
}

}  // namespace scg

using namespace scg;

extern "C" {

const char* scg_last_error(void) { return g_err; }
int32_t scg_abi_version(void) { return SCG_ABI_VERSION; }

size_t scg_struct_bytes(int32_t which) {
    switch (which) {
        case 0: return sizeof(ScgFrame);
        case 1: return sizeof(ScgWorkspaceLayout);
        case 2: return sizeof(ScgStageEvents);
        case 3: return sizeof(ScgModel);
        case 4: return sizeof(ScgModelGrads);
        default: return 0;
    }
}

size_t scg_geometry_scratch_bytes(int32_t P) { return align_up(scan_scratch_bytes(P > 0 ? P : 1), 256); }

int scg_geometry_forward(const ScgFrame* frame, const float* means3D, const float* opacities, const float* shs,
                         const float* colors_precomp, const float* scales, const float* rotations,
                         const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped,
                         uint32_t* rects, uint32_t* depth_keys, uint32_t* num_rendered_out, void* scratch,
                         size_t scratch_bytes, void* stream) {
    int rc = validate_frame(frame, false);
    if (rc) return rc;
    rc = validate_inputs(frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (frame->P == 0)
        return num_rendered_out ? check_hip(hipMemsetAsync(num_rendered_out, 0, sizeof(uint32_t), s), "memset R") : 0;
    if (!splats || !radii || !clamped || !rects || !depth_keys || !scratch)
        return fail(SCG_E_NULL, "output/scratch pointer is NULL");
    if (!aligned16(splats)) return fail(SCG_E_ALIGN, "splats must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(rects) & 7u) != 0) return fail(SCG_E_ALIGN, "rects must be 8-byte aligned");
    if (scratch_bytes < scg_geometry_scratch_bytes(frame->P))
        return fail(SCG_E_SCRATCH, "geometry scratch: %zu < %zu bytes", scratch_bytes, scg_geometry_scratch_bytes(frame->P));
    const FrameDev f = make_frame_dev(frame);
    uint32_t* block_sums = reinterpret_cast<uint32_t*>(scratch);
    rc = launch_geometry_forward(f, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, splats,
                                 radii, clamped, rects, depth_keys, block_sums, s);
    if (rc || !num_rendered_out) return rc;
    return launch_total_from_block_sums(block_sums, (frame->P + kBlock - 1) / kBlock, num_rendered_out, s);
}

// ---- binning -----------------------------------------------------------------------------------------
struct LegacyLayout { size_t keys0, keys1, vals0, offsets, scan, sortscr, total; };

static LegacyLayout legacy_layout(int P, int64_t R) {
    LegacyLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    L.keys0 = take((size_t)R * sizeof(uint64_t));
    L.keys1 = take((size_t)R * sizeof(uint64_t));
    L.vals0 = take((size_t)R * sizeof(uint32_t));
    L.offsets = take((size_t)P * sizeof(uint32_t));
    L.scan = take(scan_scratch_bytes(P));
    L.sortscr = take(sort_scratch_bytes(R));
    L.total = off;
    return L;
}

static int key_bits(int n_tiles) {
    int bits = 0;
    while ((1ll << bits) < (long long)n_tiles) ++bits;      // bits needed to represent tile ids 0..n_tiles-1
    return 32 + (bits > 0 ? bits : 1);
}

static int n_tiles_of(int32_t width, int32_t height) {
    return ((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
}

static bool use_tile_path(int n_tiles, int64_t R, int32_t algo) {
    if (algo == SCG_BINNING_GLOBAL_SORT) return false;
    return tile_binning_supported(n_tiles, R);
}

size_t scg_binning_scratch_bytes(int32_t P, int64_t num_rendered, int32_t width, int32_t height, int32_t algo) {
    const int64_t R = num_rendered > 0 ? num_rendered : 1;
    const int Pp = P > 0 ? P : 1;
    const int n_tiles = n_tiles_of(width, height);
    if (use_tile_path(n_tiles, R, algo)) return tile_binning_layout(Pp, R, n_tiles).total;
    return legacy_layout(Pp, R).total;
}

size_t scg_ranges_words(int32_t width, int32_t height) {
    const int n_tiles = ((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    return (size_t)2 * n_tiles + (size_t)5 * tile_order_slots(n_tiles);      // ranges | tile order | (tile, quadrant) order
}

int32_t scg_forward_sorts_in_blend(int64_t capacity, int32_t width, int32_t height, int32_t options) {
    if (width <= 0 || height <= 0 || capacity <= 0 || (options & SCG_DEBUG_SEPARATE_SORT)) return 0;
    const int n_tiles = n_tiles_of(width, height);
    if (!use_tile_path(n_tiles, capacity, SCG_BINNING_AUTO)) return 0;
    return tile_binning_defers_sort(capacity, n_tiles) ? 1 : 0;
}

int32_t scg_binning_accepts_bound(int64_t num_rendered_bound, int32_t width, int32_t height, int32_t algo) {
    if (width <= 0 || height <= 0) return 0;
    return use_tile_path(n_tiles_of(width, height), num_rendered_bound > 0 ? num_rendered_bound : 1, algo) ? 1 : 0;
}

int scg_binning(const ScgFrame* frame, int64_t num_rendered, const uint32_t* rects, const uint32_t* depth_keys,
                uint32_t* point_list, uint32_t* ranges, uint64_t* keys_sorted, int32_t algo, void* scratch,
                size_t scratch_bytes, void* stream) {
    int rc = validate_frame(frame, false);
    if (rc) return rc;
    if (!ranges) return fail(SCG_E_NULL, "ranges is NULL");
    if (num_rendered < 0 || num_rendered > 0xFFFFFFFFll) return fail(SCG_E_RANGE, "num_rendered out of range");
    if (algo != SCG_BINNING_AUTO && algo != SCG_BINNING_GLOBAL_SORT) return fail(SCG_E_RANGE, "unknown binning algo %d", algo);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev f = make_frame_dev(frame);
    const int n_tiles = f.gx * f.gy;
    if (num_rendered == 0 || frame->P == 0) return launch_tile_ranges(nullptr, 0, ranges, n_tiles, s);
    if (!rects || !depth_keys || !point_list || !scratch) return fail(SCG_E_NULL, "binning pointer is NULL");
    const size_t need = scg_binning_scratch_bytes(frame->P, num_rendered, frame->width, frame->height, algo);
    if (scratch_bytes < need) return fail(SCG_E_SCRATCH, "binning scratch: %zu < %zu bytes", scratch_bytes, need);

    if (use_tile_path(n_tiles, num_rendered, algo))
        return launch_tile_binning(f, num_rendered, rects, depth_keys, point_list, ranges, keys_sorted, scratch, nullptr, false, false, false, false, s);

    // global 64-bit key sort (the reference's scheme): duplicateWithKeys + 6-pass radix sort + identifyTileRanges
    const LegacyLayout L = legacy_layout(frame->P, num_rendered);
    char* base = reinterpret_cast<char*>(scratch);
    uint64_t* keys0 = reinterpret_cast<uint64_t*>(base + L.keys0);
    uint64_t* keys1 = reinterpret_cast<uint64_t*>(base + L.keys1);
    uint32_t* vals0 = reinterpret_cast<uint32_t*>(base + L.vals0);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(base + L.offsets);
    rc = launch_rect_counts_scan(rects, frame->P, offsets, reinterpret_cast<uint32_t*>(base + L.scan), s);
    if (rc) return rc;
    const int end_bit = key_bits(n_tiles);
    const bool odd = (sort_num_passes(end_bit) & 1) != 0;
    // the sorted pairs must end in (keys1, point_list): start in the other pair when the pass count is odd
    uint64_t* ka = odd ? keys0 : keys1;
    uint32_t* va = odd ? vals0 : point_list;
    uint64_t* kb = odd ? keys1 : keys0;
    uint32_t* vb = odd ? point_list : vals0;
    rc = launch_duplicate_keys(f, rects, depth_keys, offsets, ka, va, s);
    if (rc) return rc;
    rc = launch_sort_pairs(ka, va, kb, vb, num_rendered, end_bit, base + L.sortscr, s, /*result_in_b=*/odd);
    if (rc) return rc;
    rc = launch_tile_ranges(keys1, num_rendered, ranges, n_tiles, s);
    if (rc) return rc;
    if (keys_sorted)
        return check_hip(hipMemcpyAsync(keys_sorted, keys1, (size_t)num_rendered * sizeof(uint64_t),
                                        hipMemcpyDeviceToDevice, s), "copy keys_sorted");
    return 0;
}

size_t scg_sort_scratch_bytes(int64_t n) { return align_up(sort_scratch_bytes(n > 0 ? n : 1), 256); }

int scg_sort_pairs(uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                   int32_t end_bit, void* scratch, size_t scratch_bytes, void* stream) {
    if (n < 0 || n > 0xFFFFFFFFll) return fail(SCG_E_RANGE, "n out of range");
    if (end_bit < 1 || end_bit > 64) return fail(SCG_E_RANGE, "end_bit %d not in 1..64", end_bit);
    if (n == 0) return 0;
    if (!keys_in || !vals_in || !keys_out || !vals_out || !scratch) return fail(SCG_E_NULL, "sort pointer is NULL");
    if (scratch_bytes < scg_sort_scratch_bytes(n)) return fail(SCG_E_SCRATCH, "sort scratch too small");
    return launch_sort_pairs(keys_in, vals_in, keys_out, vals_out, n, end_bit, scratch,
                             reinterpret_cast<hipStream_t>(stream), /*result_in_b=*/true);
}

size_t scg_scan_scratch_bytes(int64_t n) { return align_up(scan_scratch_bytes(n > 0 ? n : 1), 256); }

int scg_inclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_out, void* scratch,
                           size_t scratch_bytes, void* stream) {
    if (n < 0 || n > 0x7FFFFFFFll * 256) return fail(SCG_E_RANGE, "n out of range");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) return total_out ? check_hip(hipMemsetAsync(total_out, 0, sizeof(uint32_t), s), "memset total") : 0;
    if (!in || !out || !scratch) return fail(SCG_E_NULL, "scan pointer is NULL");
    if (scratch_bytes < scg_scan_scratch_bytes(n)) return fail(SCG_E_SCRATCH, "scan scratch too small");
    return launch_inclusive_scan(in, out, n, total_out, scratch, s);
}

int scg_blend_forward(const ScgFrame* frame, const uint32_t* ranges, const uint32_t* point_list,
                      const float* splats, float* out_color, float* out_depth, float* out_alpha, float* final_T,
                      uint32_t* n_contrib, float* dsplats_zero, void* stream) {
    int rc = validate_frame(frame, true);
    if (rc) return rc;
    if (!ranges || !out_color || !out_depth || !out_alpha || !final_T || !n_contrib)
        return fail(SCG_E_NULL, "blend_forward pointer is NULL");
    if (splats && !aligned16(splats)) return fail(SCG_E_ALIGN, "splats must be 16-byte aligned");
    if (dsplats_zero && !aligned64(dsplats_zero)) return fail(SCG_E_ALIGN, "dsplats_zero must be 64-byte aligned");
    const FrameDev f = make_frame_dev(frame);
    return launch_blend_forward(f, ranges, point_list, splats, out_color, out_depth, out_alpha, final_T, n_contrib,
                                dsplats_zero, reinterpret_cast<hipStream_t>(stream));
}

static int blend_backward_checked(const ScgFrame* frame, const uint32_t* ranges, const uint32_t* point_list,
                                  const float* splats, const float* final_T, const uint32_t* n_contrib,
                                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, float* dsplats,
                                  int32_t dsplats_prezeroed, void* stream, hipEvent_t started, hipEvent_t done) {
    int rc = validate_frame(frame, true);
    if (rc) return rc;
    if (frame->P == 0) return 0;
    if (!ranges || !final_T || !n_contrib || !dL_dcolor || !dsplats || !splats)
        return fail(SCG_E_NULL, "blend_backward pointer is NULL");
    if (!aligned16(splats)) return fail(SCG_E_ALIGN, "splats must be 16-byte aligned");
    if (!aligned64(dsplats)) return fail(SCG_E_ALIGN, "dsplats must be 64-byte aligned (one record = one line)");
    if (frame->P > 60000000) return fail(SCG_E_RANGE, "blend_backward addresses gradient records with 32-bit offsets: P <= 60e6");
    const FrameDev f = make_frame_dev(frame);
    return launch_blend_backward(f, ranges, point_list, splats, final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha,
                                 dsplats, dsplats_prezeroed != 0, reinterpret_cast<hipStream_t>(stream), started, done);
}

int scg_blend_backward(const ScgFrame* frame, const uint32_t* ranges, const uint32_t* point_list,
                       const float* splats, const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                       const float* dL_ddepth, const float* dL_dalpha, float* dsplats, int32_t dsplats_prezeroed,
                       void* stream) {
    return blend_backward_checked(frame, ranges, point_list, splats, final_T, n_contrib, dL_dcolor, dL_ddepth, dL_dalpha,
                                  dsplats, dsplats_prezeroed, stream, nullptr, nullptr);
}

int scg_geometry_backward(const ScgFrame* frame, const float* means3D, const float* opacities, const float* shs,
                          const float* colors_precomp, const float* scales, const float* rotations,
                          const float* cov3D_precomp, const int32_t* radii, const uint8_t* clamped,
                          const float* dsplats, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities,
                          float* dL_dshs, float* dL_dcolors_precomp, float* dL_dscales, float* dL_drotations,
                          float* dL_dcov3D_precomp, int32_t accumulate, void* stream) {
    int rc = validate_frame(frame, false);
    if (rc) return rc;
    rc = validate_inputs(frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp);
    if (rc) return rc;
    if (frame->P == 0) return 0;
    if (!radii || !clamped || !dsplats || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacities)
        return fail(SCG_E_NULL, "geometry_backward pointer is NULL");
    if ((shs != nullptr) != (dL_dshs != nullptr) || (colors_precomp != nullptr) != (dL_dcolors_precomp != nullptr))
        return fail(SCG_E_EXCLUSIVE, "dL_dshs / dL_dcolors_precomp must match the colour input that was used");
    const bool has_sr = scales != nullptr;
    if (has_sr != (dL_dscales != nullptr) || has_sr != (dL_drotations != nullptr) ||
        (cov3D_precomp != nullptr) != (dL_dcov3D_precomp != nullptr))
        return fail(SCG_E_EXCLUSIVE, "dL_dscales/dL_drotations / dL_dcov3D_precomp must match the covariance input used");
    if (!aligned64(dsplats) || (dL_drotations && !aligned16(dL_drotations)))
        return fail(SCG_E_ALIGN, "dsplats must be 64-byte, dL_drotations 16-byte aligned");
    const FrameDev f = make_frame_dev(frame);
    return launch_geometry_backward(f, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, radii,
                                    clamped, dsplats, dL_dmeans3D, dL_dmeans2D, dL_dopacities, dL_dshs,
                                    dL_dcolors_precomp, dL_dscales, dL_drotations, dL_dcov3D_precomp,
                                    accumulate & (SCG_BACKWARD_ACCUMULATE | SCG_BACKWARD_SH_TAIL_ZERO),
                                    reinterpret_cast<hipStream_t>(stream));
}

// ---- the whole path in one call per direction ---------------------------------------------------------------------
static inline int mark(const ScgStageEvents* ev, int stage, bool end, hipStream_t s) {
    if (!ev) return 0;
    void* e = end ? ev->end[stage] : ev->begin[stage];
    return e ? check_hip(hipEventRecord(reinterpret_cast<hipEvent_t>(e), s), "stage event record") : 0;
}

int scg_workspace_layout(int32_t P, int64_t capacity, int32_t width, int32_t height, ScgWorkspaceLayout* out) {
    if (!out) return fail(SCG_E_NULL, "layout is NULL");
    if (P < 0 || capacity < 0 || capacity > 0xFFFFFFFFll || width <= 0 || height <= 0)
        return fail(SCG_E_RANGE, "workspace layout: P / capacity / image size out of range");
    const int64_t cap = capacity > 0 ? capacity : 1;
    const size_t Pp = (size_t)(P > 0 ? P : 1), hw = (size_t)width * height;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return (uint64_t)o; };
    out->splats = take(Pp * SCG_SPLAT_FLOATS * sizeof(float));
    out->rects = take(Pp * 2 * sizeof(uint32_t));
    out->depth_keys = take(Pp * sizeof(uint32_t));
    out->clamped = take(Pp);
    out->point_list = take((size_t)cap * sizeof(uint32_t));
    out->ranges = take(scg_ranges_words(width, height) * sizeof(uint32_t));
    out->final_T = take(hw * sizeof(float));
    out->n_contrib = take(hw * sizeof(uint32_t));
    out->bin_scratch = take(scg_binning_scratch_bytes(P, cap, width, height, SCG_BINNING_AUTO));
    out->total = (uint64_t)off;
    out->partial_words = (uint64_t)(scg_geometry_scratch_bytes(P) / sizeof(uint32_t));
    return 0;
}

// scg_forward / scg_forward_model: the same launch sequence behind two geometry kernels (`model` != NULL: the reference model's
// raw parameter tensors instead of the seven activated inputs)
static int forward_impl(const ScgFrame* frame, const ScgModel* model, const float* means3D, const float* opacities,
                        const float* shs, const float* colors_precomp, const float* scales, const float* rotations,
                        const float* cov3D_precomp, int64_t capacity, void* workspace, size_t workspace_bytes, int32_t* radii,
                        float* out_color, float* out_depth, float* out_alpha, uint32_t* partial_sums, void* event,
                        float* dsplats_zero, int32_t options, const ScgStageEvents* stage_events, void* stream) {
    int rc = validate_frame(frame, true);
    if (rc) return rc;
    rc = model ? validate_model(frame, model)
               : validate_inputs(frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp);
    if (rc) return rc;
    if (!workspace || !out_color || !out_depth || !out_alpha || !partial_sums || (frame->P > 0 && !radii))
        return fail(SCG_E_NULL, "scg_forward: workspace / output / partial_sums pointer is NULL");
    if (!aligned16(workspace)) return fail(SCG_E_ALIGN, "workspace must be 16-byte aligned");
    if (dsplats_zero && !aligned64(dsplats_zero)) return fail(SCG_E_ALIGN, "dsplats_zero must be 64-byte aligned");
    ScgWorkspaceLayout L;
    rc = scg_workspace_layout(frame->P, capacity, frame->width, frame->height, &L);
    if (rc) return rc;
    if (workspace_bytes < L.total) return fail(SCG_E_SCRATCH, "workspace: %zu < %llu bytes", workspace_bytes, (unsigned long long)L.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev f = make_frame_dev(frame);
    const int n_tiles = f.gx * f.gy;
    char* base = reinterpret_cast<char*>(workspace);
    float* splats = reinterpret_cast<float*>(base + L.splats);
    uint32_t* rects = reinterpret_cast<uint32_t*>(base + L.rects);
    uint32_t* depth_keys = reinterpret_cast<uint32_t*>(base + L.depth_keys);
    uint8_t* clamped = reinterpret_cast<uint8_t*>(base + L.clamped);
    uint32_t* point_list = reinterpret_cast<uint32_t*>(base + L.point_list);
    uint32_t* ranges = reinterpret_cast<uint32_t*>(base + L.ranges);
    // a render that will not be differentiated does not write the backward's per-pixel state (8 of the 28 bytes per pixel)
    const bool keep_state = !(options & SCG_FORWARD_NO_BACKWARD_STATE);
    float* final_T = keep_state ? reinterpret_cast<float*>(base + L.final_T) : nullptr;
    uint32_t* n_contrib = keep_state ? reinterpret_cast<uint32_t*>(base + L.n_contrib) : nullptr;
    const bool empty = frame->P == 0 || capacity == 0;
    if (!empty && !use_tile_path(n_tiles, capacity, SCG_BINNING_AUTO))
        return fail(SCG_E_RANGE, "scg_forward needs the tile-first binning path (scg_binning_accepts_bound); use the staged calls");
    if (options & SCG_FORWARD_ARM_PARTIAL_SUMS) {               // (host memory: written here, on the host, before the launch)
        const int nb = frame->P > 0 ? (frame->P + kBlock - 1) / kBlock : 1;
        volatile uint32_t* w = partial_sums;
        for (int i = 0; i < nb; ++i) w[i] = SCG_PARTIAL_SUM_ARMED;
        std::atomic_thread_fence(std::memory_order_release);
    }
    if ((rc = mark(stage_events, 0, false, s))) return rc;
    // the slice histograms of the binning stage are built by the geometry kernel itself (one launch and the re-read of the
    // rectangles less) unless the caller or the shape says otherwise
    const bool hist_in_geometry = !empty && !(options & SCG_DEBUG_SEPARATE_HIST) && tile_binning_hist_in_geometry(f, capacity);
    if (frame->P == 0) {
        rc = check_hip(hipMemsetAsync(partial_sums, 0, sizeof(uint32_t), s), "memset partial sums");
    } else if (hist_in_geometry) {
        rc = model ? launch_geometry_hist_binned_model(f, capacity, *model, splats, radii, clamped, rects, depth_keys, partial_sums,
                                                       base + L.bin_scratch, s)
                   : launch_geometry_hist_binned(f, capacity, means3D, opacities, shs, colors_precomp, scales, rotations,
                                                 cov3D_precomp, splats, radii, clamped, rects, depth_keys, partial_sums,
                                                 base + L.bin_scratch, s);
    } else {
        rc = model ? launch_geometry_forward_model(f, *model, splats, radii, clamped, rects, depth_keys, partial_sums, s)
                   : launch_geometry_forward(f, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, splats,
                                             radii, clamped, rects, depth_keys, partial_sums, s);
    }
    if (rc) return rc;
    if ((rc = mark(stage_events, 0, true, s))) return rc;
    if (event) {
        rc = check_hip(hipEventRecord(reinterpret_cast<hipEvent_t>(event), s), "event record");
        if (rc) return rc;
    }
    if ((rc = mark(stage_events, 1, false, s))) return rc;
    // the forward blend sorts the tiles' lists itself (one launch and its drain less) unless the binning stage decides
    // otherwise (dense scenes)
    bool fused_sort = !empty && !(options & SCG_DEBUG_SEPARATE_SORT);
    const bool skip_rare = (options & SCG_FORWARD_SKIP_RARE_SORT) != 0;
    if (empty && frame->num_rendered_out)                       // (nothing is binned: the count the caller looks at later is 0)
        if ((rc = check_hip(hipMemsetAsync(frame->num_rendered_out, 0, sizeof(uint32_t), s), "memset num_rendered_out"))) return rc;
    rc = empty ? launch_tile_ranges(nullptr, 0, ranges, n_tiles, s)
               : launch_tile_binning(f, capacity, rects, depth_keys, point_list, ranges, nullptr, base + L.bin_scratch,
                                     &fused_sort, hist_in_geometry, skip_rare, (options & SCG_FORWARD_RARE_8WAVE) != 0,
                                     (options & SCG_FORWARD_SPLIT_LONG_LISTS) != 0, s);
    if (rc) return rc;
    if ((rc = mark(stage_events, 1, true, s))) return rc;
    if ((rc = mark(stage_events, 2, false, s))) return rc;
    rc = fused_sort ? launch_tile_blend_forward(f, ranges, point_list, depth_keys, splats, out_color, out_depth, out_alpha,
                                                final_T, n_contrib, frame->P ? dsplats_zero : nullptr, base + L.bin_scratch,
                                                capacity, !skip_rare, s)
                    : launch_blend_forward(f, ranges, point_list, splats, out_color, out_depth, out_alpha, final_T, n_contrib,
                                           frame->P ? dsplats_zero : nullptr, s);
    if (rc) return rc;
    return mark(stage_events, 2, true, s);
}

int scg_forward(const ScgFrame* frame, const float* means3D, const float* opacities, const float* shs,
                const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                int64_t capacity, void* workspace, size_t workspace_bytes, int32_t* radii, float* out_color,
                float* out_depth, float* out_alpha, uint32_t* partial_sums, void* event, float* dsplats_zero,
                int32_t options, const ScgStageEvents* stage_events, void* stream) {
    return forward_impl(frame, nullptr, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, capacity,
                        workspace, workspace_bytes, radii, out_color, out_depth, out_alpha, partial_sums, event, dsplats_zero,
                        options, stage_events, stream);
}

int scg_forward_model(const ScgFrame* frame, const ScgModel* model, int64_t capacity, void* workspace, size_t workspace_bytes,
                      int32_t* radii, float* out_color, float* out_depth, float* out_alpha, uint32_t* partial_sums, void* event,
                      float* dsplats_zero, int32_t options, const ScgStageEvents* stage_events, void* stream) {
    if (!model) return fail(SCG_E_NULL, "model is NULL");
    return forward_impl(frame, model, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, capacity, workspace,
                        workspace_bytes, radii, out_color, out_depth, out_alpha, partial_sums, event, dsplats_zero, options,
                        stage_events, stream);
}

int scg_model_activate(const ScgModel* model, float* means3D, float* opacities, float* scales, float* rotations, void* stream) {
    if (!model) return fail(SCG_E_NULL, "model is NULL");
    ScgFrame fr;
    memset(&fr, 0, sizeof(fr));
    fr.P = model->ray.count + model->bg.count;
    fr.sh_coeffs = 16;
    const int rc = validate_model(&fr, model);
    if (rc) return rc;
    if (rotations && !aligned16(rotations)) return fail(SCG_E_ALIGN, "rotations must be 16-byte aligned");
    return launch_model_activate(*model, means3D, opacities, scales, rotations, reinterpret_cast<hipStream_t>(stream));
}

int64_t scg_wait_num_rendered(void* event, const uint32_t* partial_sums_host, int32_t P) {
    if (!partial_sums_host) { fail(SCG_E_NULL, "partial_sums_host is NULL"); return SCG_E_NULL; }
    const int nb = P > 0 ? (P + kBlock - 1) / kBlock : 1;
    if (event) {
        const hipError_t e = hipEventSynchronize(reinterpret_cast<hipEvent_t>(event));
        if (e != hipSuccess) { check_hip(e, "event synchronize"); return -(int64_t)e - 1000; }
    } else {
        // No event: the words were armed (SCG_FORWARD_ARM_PARTIAL_SUMS) with a value no sum can take, and every one of them is
        // written by exactly one workgroup of the geometry kernel, straight into this (pinned, host-coherent) memory: the host
        // watches them arrive.  No barrier packet in the queue behind the geometry kernel, no event wake-up on the host.
        const volatile uint32_t* w = partial_sums_host;
        const auto t0 = std::chrono::steady_clock::now();
        // (a geometry kernel takes 10-100 us: the first ~100 000 polls are a busy wait; a wait that lasts longer — a profiler that
        //  serialises kernels, a debugger, a GPU shared with another process — yields the core between polls, and gives up only
        //  after two minutes.  The caller must synchronise the stream before it releases the call's buffers then: the kernels may
        //  still be writing them — the Python binding does, rasterizer.forward_fused's except path.)
        for (int i = nb - 1; i >= 0; --i) {
            uint64_t spins = 0;
            while (w[i] == SCG_PARTIAL_SUM_ARMED) {
                if (spins < 100000u) __builtin_ia32_pause(); else std::this_thread::yield();
                if ((++spins & 0xFFFFu) == 0 &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                    fail(SCG_E_RANGE, "scg_wait_num_rendered: the geometry stage's partial sums did not arrive within 120 s "
                                      "(kernel fault, or partial_sums is not host-coherent memory)");
                    return SCG_E_RANGE;
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    int64_t total = 0;
    for (int i = 0; i < nb; ++i) total += partial_sums_host[i];
    return total;
}

int scg_event_create(void** event_out, int32_t timing) {
    if (!event_out) return fail(SCG_E_NULL, "event_out is NULL");
    hipEvent_t ev;
    const int rc = check_hip(hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming), "event create");
    if (rc) return rc;
    *event_out = reinterpret_cast<void*>(ev);
    return 0;
}

int scg_event_destroy(void* event) {
    return event ? check_hip(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)), "event destroy") : 0;
}

int scg_event_elapsed_ms(void* begin, void* end, float* ms_out) {
    if (!begin || !end || !ms_out) return fail(SCG_E_NULL, "event / ms_out is NULL");
    return check_hip(hipEventElapsedTime(ms_out, reinterpret_cast<hipEvent_t>(begin), reinterpret_cast<hipEvent_t>(end)),
                     "event elapsed time");
}

int scg_backward(const ScgFrame* frame, const float* means3D, const float* opacities, const float* shs,
                 const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii, int64_t capacity, const void* workspace, const float* dL_dcolor,
                 const float* dL_ddepth, const float* dL_dalpha, float* dsplats, int32_t dsplats_prezeroed,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities, float* dL_dshs, float* dL_dcolors_precomp,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D_precomp, int32_t accumulate,
                 const ScgStageEvents* stage_events, void* stream) {
    if (!frame) return fail(SCG_E_NULL, "frame is NULL");
    if (frame->P == 0) return 0;
    if (!workspace) return fail(SCG_E_NULL, "workspace is NULL");
    ScgWorkspaceLayout L;
    int rc = scg_workspace_layout(frame->P, capacity, frame->width, frame->height, &L);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const char* base = reinterpret_cast<const char*>(workspace);
    // the blend backward's two stage events ride on its dispatch packet (begin / end time stamps of the kernel itself)
    hipEvent_t bb_begin = stage_events ? reinterpret_cast<hipEvent_t>(stage_events->begin[0]) : nullptr;
    hipEvent_t bb_end = stage_events ? reinterpret_cast<hipEvent_t>(stage_events->end[0]) : nullptr;
    rc = blend_backward_checked(frame, reinterpret_cast<const uint32_t*>(base + L.ranges),
                                reinterpret_cast<const uint32_t*>(base + L.point_list),
                                reinterpret_cast<const float*>(base + L.splats),
                                reinterpret_cast<const float*>(base + L.final_T),
                                reinterpret_cast<const uint32_t*>(base + L.n_contrib), dL_dcolor, dL_ddepth, dL_dalpha,
                                dsplats, dsplats_prezeroed, stream, bb_begin, bb_end);
    if (rc) return rc;
    if ((rc = mark(stage_events, 1, false, s))) return rc;
    rc = scg_geometry_backward(frame, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, radii,
                               reinterpret_cast<const uint8_t*>(base + L.clamped), dsplats, dL_dmeans3D, dL_dmeans2D,
                               dL_dopacities, dL_dshs, dL_dcolors_precomp, dL_dscales, dL_drotations, dL_dcov3D_precomp,
                               accumulate, stream);
    if (rc) return rc;
    return mark(stage_events, 1, true, s);
}

int scg_backward_model(const ScgFrame* frame, const ScgModel* model, const int32_t* radii, int64_t capacity, const void* workspace,
                       const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, float* dsplats,
                       int32_t dsplats_prezeroed, const ScgModelGrads* grads, float* dL_dmeans2D, int32_t flags,
                       const ScgStageEvents* stage_events, void* stream) {
    if (!frame) return fail(SCG_E_NULL, "frame is NULL");
    if (!model || !grads) return fail(SCG_E_NULL, "model / grads is NULL");
    int rc = validate_frame(frame, false);
    if (rc) return rc;
    if ((rc = validate_model(frame, model))) return rc;
    if (frame->P == 0) return 0;
    if (!workspace || !radii || !dL_dmeans2D) return fail(SCG_E_NULL, "workspace / radii / dL_dmeans2D is NULL");
    for (int k = 0; k < 2; ++k) {
        const ScgModelSet& ms = k ? model->bg : model->ray;
        const ScgModelGradSet& gs = k ? grads->bg : grads->ray;
        if (ms.count == 0) continue;
        if ((k == 0 ? !gs.zval : !gs.xyz) || !gs.features_dc || !gs.features_rest || !gs.opacity || !gs.scaling || !gs.rotation)
            return fail(SCG_E_NULL, "grads.%s: a gradient buffer of a non-empty set is NULL", k ? "bg" : "ray");
        if (!aligned16(gs.rotation) || !aligned16(gs.features_rest) || !aligned16(gs.features_dc))
            return fail(SCG_E_ALIGN, "grads.%s: rotation / features_rest / features_dc must be 16-byte aligned", k ? "bg" : "ray");
    }
    ScgWorkspaceLayout L;
    rc = scg_workspace_layout(frame->P, capacity, frame->width, frame->height, &L);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const char* base = reinterpret_cast<const char*>(workspace);
    hipEvent_t bb_begin = stage_events ? reinterpret_cast<hipEvent_t>(stage_events->begin[0]) : nullptr;
    hipEvent_t bb_end = stage_events ? reinterpret_cast<hipEvent_t>(stage_events->end[0]) : nullptr;
    rc = blend_backward_checked(frame, reinterpret_cast<const uint32_t*>(base + L.ranges),
                                reinterpret_cast<const uint32_t*>(base + L.point_list),
                                reinterpret_cast<const float*>(base + L.splats),
                                reinterpret_cast<const float*>(base + L.final_T),
                                reinterpret_cast<const uint32_t*>(base + L.n_contrib), dL_dcolor, dL_ddepth, dL_dalpha,
                                dsplats, dsplats_prezeroed, stream, bb_begin, bb_end);
    if (rc) return rc;
    if ((rc = mark(stage_events, 1, false, s))) return rc;
    const FrameDev f = make_frame_dev(frame);
    rc = launch_geometry_backward_model(f, *model, *grads, radii, reinterpret_cast<const uint8_t*>(base + L.clamped), dsplats,
                                        dL_dmeans2D, flags & (SCG_BACKWARD_ACCUMULATE | SCG_BACKWARD_SH_TAIL_ZERO), s);
    if (rc) return rc;
    return mark(stage_events, 1, true, s);
}

}  // extern "C"
