// geometry.hip — per-Gaussian stage of the rasterizer, forward and backward.
//
// Forward: frustum cull, projection, cov3D from scale/rotation, EWA cov2D -> conic/radius/tile rect,
// SH -> RGB fused (utils/sh_utils.py:57-103 polynomials; the python twin lives at reference
// gaussian_renderer/__init__.py:79-83), the packed tile rectangle + depth sort key per Gaussian, and the
// per-block partial sums of tiles touched (their total is num_rendered).
// Backward: analytic chain rule from the splat-record gradients back to means3D / SH / opacity /
// scales / rotations (or the precomputed colour / cov3D inputs).
//
// This translation unit is compiled with -ffp-contract=off: every value that decides an INTEGER
// (radius, tile rectangle, depth key bits) is a chain of single correctly-rounded fp32 operations in the
// same order as oracle/torch_rasterizer.py::preprocess, so tile assignment is bit-exact.
//
// One thread per Gaussian, 256-thread workgroups (4 waves).  Loads of the (P,3)/(P,4) arrays are
// lane-contiguous; the 192-byte SH record is read with 16-byte loads (12 per lane at degree 3).
#include "scg_common.h"
#include "tile_walk.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace scg {

__constant__ const float kC0 = 0.28209479177387814f;
__constant__ const float kC1 = 0.4886025119029199f;
__constant__ const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__constant__ const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

struct Mat16 { float m[16]; };

__device__ __forceinline__ Mat16 load16(const float* __restrict__ p) {
    Mat16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.m[i] = p[i];
    return r;
}

// Quantities of the forward projection that the backward recomputes identically.
struct Proj {
    float tx, ty, tz;            // view-space position
    float hx, hy, m_w;           // homogeneous clip x, y and 1/(w+eps)
    float t_x, t_y;              // clamped view x, y used by the Jacobian
    bool cl_x, cl_y;             // whether the 1.3*tanfov clamp was active
    float cov[6];                // 3D covariance, packed xx xy xz yy yz zz
    float T00, T01, T02, T10, T11, T12;   // Tm = J . Wm
    float u0, u1, u2, v0, v1, v2;         // Sigma . T0^T, Sigma . T1^T
    float A, B, C, det, det_inv;          // 2D covariance (+low-pass), determinant
    float L[9];                  // R.S (only when built from scale/rotation)
    float R[9];
};

// The per-Gaussian inputs that depend on nothing computed: ALL loaded together at the top of a Gaussian's work, in front of
// the cull (round 5).  Round 4's kernels fetched them where they were first used — means3D, then (behind the cull test) scales
// and rotations, then the SH record, then the opacity: four dependent round trips to HBM per Gaussian, each with a few hundred
// bytes of a wave in flight; the streaming kernels ran at 0.49-0.57 of the HBM roofline on latency, not on bytes.
struct GeoIn {
    float x, y, z, opacity;
    float a[7];                  // rotation (r, x, y, z) + scales (not yet multiplied by the modifier) — or cov3D_precomp's six
                                 // floats in a[0..5]: one set of registers for the two exclusive input paths
    float rgb[3];                // colors_precomp path
};

template <bool WITH_RGB>
__device__ __forceinline__ GeoIn load_geo_in(int i, const float* __restrict__ means3D, const float* __restrict__ opacities,
                                             const float* __restrict__ colors_precomp, const float* __restrict__ scales,
                                             const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp) {
    GeoIn g;
    g.x = means3D[3 * (size_t)i + 0];
    g.y = means3D[3 * (size_t)i + 1];
    g.z = means3D[3 * (size_t)i + 2];
    g.opacity = opacities[i];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) g.a[k] = cov3D_precomp[6 * (size_t)i + k];
        g.a[6] = 0.f;
    } else {
        const float4 q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
        g.a[0] = q.x; g.a[1] = q.y; g.a[2] = q.z; g.a[3] = q.w;
        g.a[4] = scales[3 * (size_t)i + 0];
        g.a[5] = scales[3 * (size_t)i + 1];
        g.a[6] = scales[3 * (size_t)i + 2];
    }
    if (WITH_RGB) {
        g.rgb[0] = colors_precomp[3 * (size_t)i + 0];
        g.rgb[1] = colors_precomp[3 * (size_t)i + 1];
        g.rgb[2] = colors_precomp[3 * (size_t)i + 2];
    } else {
        g.rgb[0] = g.rgb[1] = g.rgb[2] = 0.f;
    }
    return g;
}

__device__ __forceinline__ void cov3d_from_scale_rot(const GeoIn& g, float mod, Proj& p) {
    const float r = g.a[0], x = g.a[1], y = g.a[2], z = g.a[3];
    p.R[0] = 1.0f - 2.0f * (y * y + z * z);
    p.R[1] = 2.0f * (x * y - r * z);
    p.R[2] = 2.0f * (x * z + r * y);
    p.R[3] = 2.0f * (x * y + r * z);
    p.R[4] = 1.0f - 2.0f * (x * x + z * z);
    p.R[5] = 2.0f * (y * z - r * x);
    p.R[6] = 2.0f * (x * z - r * y);
    p.R[7] = 2.0f * (y * z + r * x);
    p.R[8] = 1.0f - 2.0f * (x * x + y * y);
    const float s0 = mod * g.a[4];
    const float s1 = mod * g.a[5];
    const float s2 = mod * g.a[6];
    float* L = p.L;
    L[0] = p.R[0] * s0; L[1] = p.R[1] * s1; L[2] = p.R[2] * s2;
    L[3] = p.R[3] * s0; L[4] = p.R[4] * s1; L[5] = p.R[5] * s2;
    L[6] = p.R[6] * s0; L[7] = p.R[7] * s1; L[8] = p.R[8] * s2;
    p.cov[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
    p.cov[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
    p.cov[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
    p.cov[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
    p.cov[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
    p.cov[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

// Shared forward math: returns false when the Gaussian is behind the near plane.
__device__ __forceinline__ bool project(const FrameDev& f, const Mat16& V, const Mat16& PM, float x, float y,
                                        float z, Proj& p) {
    p.tx = V.m[0] * x + V.m[4] * y + V.m[8] * z + V.m[12];
    p.ty = V.m[1] * x + V.m[5] * y + V.m[9] * z + V.m[13];
    p.tz = V.m[2] * x + V.m[6] * y + V.m[10] * z + V.m[14];
    if (!(p.tz > kNearZ)) return false;
    p.hx = PM.m[0] * x + PM.m[4] * y + PM.m[8] * z + PM.m[12];
    p.hy = PM.m[1] * x + PM.m[5] * y + PM.m[9] * z + PM.m[13];
    const float hw = PM.m[3] * x + PM.m[7] * y + PM.m[11] * z + PM.m[15];
    p.m_w = 1.0f / (hw + 1e-7f);
    return true;
}

__device__ __forceinline__ void cov2d(const FrameDev& f, const Mat16& V, Proj& p) {
    const float txtz = p.tx / p.tz;
    const float tytz = p.ty / p.tz;
    p.cl_x = (txtz < -f.limx) || (txtz > f.limx);
    p.cl_y = (tytz < -f.limy) || (tytz > f.limy);
    p.t_x = fminf(fmaxf(txtz, -f.limx), f.limx) * p.tz;
    p.t_y = fminf(fmaxf(tytz, -f.limy), f.limy) * p.tz;
    const float tz2 = p.tz * p.tz;
    const float J00 = f.focal_x / p.tz;
    const float J02 = -(f.focal_x * p.t_x) / tz2;
    const float J11 = f.focal_y / p.tz;
    const float J12 = -(f.focal_y * p.t_y) / tz2;
    p.T00 = J00 * V.m[0] + J02 * V.m[2];
    p.T01 = J00 * V.m[4] + J02 * V.m[6];
    p.T02 = J00 * V.m[8] + J02 * V.m[10];
    p.T10 = J11 * V.m[1] + J12 * V.m[2];
    p.T11 = J11 * V.m[5] + J12 * V.m[6];
    p.T12 = J11 * V.m[9] + J12 * V.m[10];
    const float* c = p.cov;
    p.u0 = c[0] * p.T00 + c[1] * p.T01 + c[2] * p.T02;
    p.u1 = c[1] * p.T00 + c[3] * p.T01 + c[4] * p.T02;
    p.u2 = c[2] * p.T00 + c[4] * p.T01 + c[5] * p.T02;
    p.v0 = c[0] * p.T10 + c[1] * p.T11 + c[2] * p.T12;
    p.v1 = c[1] * p.T10 + c[3] * p.T11 + c[4] * p.T12;
    p.v2 = c[2] * p.T10 + c[4] * p.T11 + c[5] * p.T12;
    p.A = p.T00 * p.u0 + p.T01 * p.u1 + p.T02 * p.u2 + kLowpass;
    p.B = p.T00 * p.v0 + p.T01 * p.v1 + p.T02 * p.v2;
    p.C = p.T10 * p.v0 + p.T11 * p.v1 + p.T12 * p.v2 + kLowpass;
    p.det = p.A * p.C - p.B * p.B;
    p.det_inv = 1.0f / p.det;
}

// Load the first K coefficients (3 floats each) of one SH record into sh[(3K + 3) / 4 * 4].
template <int K>
__device__ __forceinline__ void load_sh(const float* __restrict__ rec, bool vec16, float* sh) {
    constexpr int n = 3 * K;
    if (vec16) {
        constexpr int nv = (n + 3) / 4;
        const float4* r4 = reinterpret_cast<const float4*>(rec);
#pragma unroll
        for (int i = 0; i < nv; ++i) {
            const float4 v = r4[i];
            sh[4 * i + 0] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < n; ++i) sh[i] = rec[i];
#pragma unroll
        for (int i = n; i < (n + 3) / 4 * 4; ++i) sh[i] = 0.f;    // (every element defined on both paths: the array stays in registers)
    }
}

// ---------------------------------------------------------------------------------------------------
// Where a Gaussian's inputs come from: the kernels below are written once over a SOURCE.
//   TensorSource  the operator's seven input tensors, ACTIVATED values (reference gaussian_renderer/__init__.py:100-108)
//   ModelSource   the reference model's RAW parameter tensors (scene/gaussian_model.py:452-468; include/scg_raster.h ScgModel):
//                 position = rayo + rayd * zval or bg_xyz, sigmoid / exp / normalize in registers, the SH record read from
//                 its two parts (features_dc, features_rest) — what the reference's getters (:105-152) compute with ~25 torch
//                 launches and a 192 B / Gaussian concatenation per render() call
// ---------------------------------------------------------------------------------------------------
struct TensorSource {
    const float* means3D; const float* opacities; const float* shs; const float* colors_precomp;
    const float* scales; const float* rotations; const float* cov3D_precomp;
    int M, sh_vec16;
    __device__ __forceinline__ bool has_cov() const { return cov3D_precomp != nullptr; }
    template <bool WITH_RGB>
    __device__ __forceinline__ GeoIn load(int i) const {
        return load_geo_in<WITH_RGB>(i, means3D, opacities, colors_precomp, scales, rotations, cov3D_precomp);
    }
    template <int K>
    __device__ __forceinline__ void load_sh(int i, float* sh) const { scg::load_sh<K>(shs + (size_t)i * M * 3, sh_vec16 != 0, sh); }
};

// The reference's activations (scene/gaussian_model.py:37-51): ONE definition for the forward, the backward's recomputation and
// scg_model_activate — the same instructions, hence the same bits, wherever an activated value is needed.
constexpr float kNormalizeEps = 1e-12f;                       // torch.nn.functional.normalize's eps
__device__ __forceinline__ float act_opacity(float logit) { return 1.0f / (1.0f + expf(-logit)); }
__device__ __forceinline__ float act_scale(float log_s) { return expf(log_s); }
__device__ __forceinline__ float quat_denominator(float r, float x, float y, float z) {
    return fmaxf(sqrtf(r * r + x * x + y * y + z * z), kNormalizeEps);
}

constexpr int kRestCoeffs = 15;                               // features_rest is (n, 15, 3): M = 16 records in two parts

struct ModelSource {
    ScgModel m;
    __device__ __forceinline__ bool has_cov() const { return false; }
    // (lane's set and index inside it: the two sets are indexed one behind the other)
    __device__ __forceinline__ bool locate(int i, int& j) const {
        const bool ray = i < m.ray.count;
        j = ray ? i : i - m.ray.count;
        return ray;
    }
    // A Gaussian's raw parameters as loaded (no arithmetic: the backward issues these loads next to its other loads and
    // activates behind them), and their activation.  den_out / rayd_out (backward only): the quaternion's denominator
    // max(|q|, eps) and the ray direction — what the derivatives of normalize and of rayo + rayd * zval need.
    struct Raw { float o[3], d[3], zv, logit, ls[3]; float4 q; bool ray; };
    __device__ __forceinline__ Raw load_raw(int i) const {
        int j;
        Raw r;
        r.ray = locate(i, j);
        r.zv = 0.f; r.o[0] = r.o[1] = r.o[2] = 0.f;
        if (r.ray) {
            r.zv = m.ray.zval[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) { r.o[c] = m.ray.rayo[3 * (size_t)j + c]; r.d[c] = m.ray.rayd[3 * (size_t)j + c]; }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) r.d[c] = m.bg.xyz[3 * (size_t)j + c];
        }
        const float* op = r.ray ? m.ray.opacity : m.bg.opacity;
        const float* sc = r.ray ? m.ray.scaling : m.bg.scaling;
        const float* ro = r.ray ? m.ray.rotation : m.bg.rotation;
        r.logit = op[j];
        r.q = *reinterpret_cast<const float4*>(ro + 4 * (size_t)j);
#pragma unroll
        for (int c = 0; c < 3; ++c) r.ls[c] = sc[3 * (size_t)j + c];
        return r;
    }
    static __device__ __forceinline__ GeoIn activate(const Raw& r, float* den_out = nullptr, float* rayd_out = nullptr) {
        GeoIn g;
        // rayo + rayd * zval: two roundings (this file is compiled with contraction off), as torch evaluates it (:127)
        g.x = r.ray ? r.o[0] + r.d[0] * r.zv : r.d[0];
        g.y = r.ray ? r.o[1] + r.d[1] * r.zv : r.d[1];
        g.z = r.ray ? r.o[2] + r.d[2] * r.zv : r.d[2];
        g.opacity = act_opacity(r.logit);
        const float den = quat_denominator(r.q.x, r.q.y, r.q.z, r.q.w);
        g.a[0] = r.q.x / den; g.a[1] = r.q.y / den; g.a[2] = r.q.z / den; g.a[3] = r.q.w / den;
        g.a[4] = act_scale(r.ls[0]); g.a[5] = act_scale(r.ls[1]); g.a[6] = act_scale(r.ls[2]);
        g.rgb[0] = g.rgb[1] = g.rgb[2] = 0.f;
        if (den_out) *den_out = den;
        if (rayd_out) { rayd_out[0] = r.ray ? r.d[0] : 0.f; rayd_out[1] = r.ray ? r.d[1] : 0.f; rayd_out[2] = r.ray ? r.d[2] : 0.f; }
        return g;
    }
    template <bool WITH_RGB>
    __device__ __forceinline__ GeoIn load(int i) const { return activate(load_raw(i)); }
    // coefficient 0 from features_dc (12 bytes per Gaussian: a dense stream at degree 0), 1 .. K-1 from the head of the
    // Gaussian's 180-byte features_rest record
    template <int K>
    __device__ __forceinline__ void load_sh(int i, float* sh) const {
        int j;
        const bool ray = locate(i, j);
        const float* dc = (ray ? m.ray.features_dc : m.bg.features_dc) + 3 * (size_t)j;
        sh[0] = dc[0]; sh[1] = dc[1]; sh[2] = dc[2];
        if (K > 1) {
            const float* rest = (ray ? m.ray.features_rest : m.bg.features_rest) + 3 * kRestCoeffs * (size_t)j;
#pragma unroll
            for (int k = 0; k < 3 * (K - 1); ++k) sh[3 + k] = rest[k];
        }
#pragma unroll
        for (int k = 3 * K; k < (3 * K + 3) / 4 * 4; ++k) sh[k] = 0.f;
    }
};

// rgb = eval_sh(deg, sh, dir) in the operation order of oracle eval_sh_rgb (== utils/sh_utils.py:57-103).
template <int DEG>
__device__ __forceinline__ void eval_sh(const float* sh, float x, float y, float z, float* rgb) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float res = kC0 * sh[c];
        if (DEG > 0) {
            res = res - (kC1 * y) * sh[3 + c] + (kC1 * z) * sh[6 + c] - (kC1 * x) * sh[9 + c];
            if (DEG > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                const float xy = x * y, yz = y * z, xz = x * z;
                res = res + (kC2[0] * xy) * sh[12 + c] + (kC2[1] * yz) * sh[15 + c] +
                      (kC2[2] * (2.0f * zz - xx - yy)) * sh[18 + c] + (kC2[3] * xz) * sh[21 + c] +
                      (kC2[4] * (xx - yy)) * sh[24 + c];
                if (DEG > 2) {
                    res = res + ((kC3[0] * y) * (3.0f * xx - yy)) * sh[27 + c] +
                          ((kC3[1] * xy) * z) * sh[30 + c] +
                          ((kC3[2] * y) * (4.0f * zz - xx - yy)) * sh[33 + c] +
                          ((kC3[3] * z) * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * sh[36 + c] +
                          ((kC3[4] * x) * (4.0f * zz - xx - yy)) * sh[39 + c] +
                          ((kC3[5] * z) * (xx - yy)) * sh[42 + c] +
                          ((kC3[6] * x) * (xx - 3.0f * yy)) * sh[45 + c];
                }
            }
        }
        rgb[c] = res;
    }
}

// ---------------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------------
// One wave's LDS stage: 64 splat records of three float4 (48-byte stride: conflict-free 16-byte writes), then 64 uint4
// {radius, clamp bits, rectangle} — 4 KiB.
constexpr int kStageVec = 4 * kWave;

// Store what geometry_forward_one parked in the stage for the 64 Gaussians from `first` on.  Why not right away: vmcnt counts
// loads and stores alike and retires them in order, so a wave that stores its outputs and then fetches its next chunk waits
// for the WRITES to be acknowledged before the first load it needs counts as arrived (round-5 probe: the forward without its
// output stores 62 -> 41 us at a million Gaussians; transposing the records through LDS alone changed nothing).  The looping
// kernel therefore flushes chunk k while the loads of the SH records of chunk k + 1 are already in flight.
__device__ __forceinline__ void flush_stage(const float4* stage, int first, int P, float4* __restrict__ splats,
                                            int32_t* __restrict__ radii, uint8_t* __restrict__ clamped,
                                            uint2* __restrict__ rects, uint32_t* __restrict__ depth_keys) {
    const int lane = lane_id();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = min(kWave, P - first);                          // Gaussians of the chunk (<= 0: nothing)
    float4* dst = splats + 3 * (size_t)first;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = k * kWave + lane;
        if (idx < 3 * n) dst[idx] = stage[idx];
    }
    if (lane < n) {
        const uint4 u = reinterpret_cast<const uint4*>(stage + 3 * kWave)[lane];
        const float tz = stage[3 * lane + 2].w;                   // depth of a visible Gaussian (> 0.2), 0 of a culled one
        const int i = first + lane;
        radii[i] = (int32_t)u.x;
        clamped[i] = (uint8_t)u.y;
        rects[i] = make_uint2(u.z, u.w);
        depth_keys[i] = u.x ? __float_as_uint(tz) : 0xFFFFFFFFu;
    }
    // (the stage is rewritten by the wave's next chunk: the reads above are complete before a later write can execute — the
    // LDS queue of a wave is in order; the fence keeps the compiler from moving them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Everything geometry_forward does for one Gaussian per lane: cull, project, covariance, conic, radius, tile rectangle,
// colour; PARKS the splat record, radius, clamp bits and rectangle in the wave's LDS stage (flush_stage stores them).
// Returns tiles_touched.
//   in       the lane's inputs (load_geo_in), fetched by the caller — at the top of the kernel, or one chunk ahead
//   valid    lane has a Gaussian (i < f.P); the function is called by ALL lanes of the wave
//   i        index of the lane's Gaussian; the lanes of a wave hold 64 CONSECUTIVE Gaussians (i - lane = the first one)
//   between  called once by all lanes, in uniform control flow, with the lane's rectangle — AFTER the loads of the SH
//            record were issued and BEFORE they are consumed: the one-call path's kernel walks the rectangle into its
//            tile histogram there (LDS atomics next to HBM latency)
//   DEG      active SH degree, -1: colours come precomputed (in.rgb).  A template parameter so that the record's registers are
//            plain registers from the issue of the loads to their use (a run-time degree sent the array to scratch memory)
template <int DEG, class Src, class Between>
__device__ __forceinline__ uint32_t geometry_forward_one(
    const FrameDev& f, const Mat16& V, const Mat16& PM, int i, bool valid, const GeoIn& in,
    const Src& src, float4* stage /* LDS, kStageVec float4 of this wave */, Between&& between
#ifdef SCG_PROBE_TIMELINE                       // tools/probes/geometry_timeline.py: per-wave phase clocks
    , uint32_t* g_tp
#endif
    ) {
    uint32_t my_tiles = 0;
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa, sc = sa;
    int radius_i = 0;
    uint8_t clamp_bits = 0;
    uint2 rect = make_uint2(0u, 0u);          // {minx | miny<<16, width | height<<16} in tiles
    float px = 0.f, py = 0.f, con_a = 0.f, con_b = 0.f, con_c = 0.f, tz = 0.f;

    Proj p;
    bool ok = valid && project(f, V, PM, in.x, in.y, in.z, p);
    if (ok) {
        if (src.has_cov()) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.cov[k] = in.a[k];
        } else {
            cov3d_from_scale_rot(in, f.mod, p);
        }
        cov2d(f, V, p);
        ok = (p.det != 0.0f);
    }
    if (ok) {
        con_a = p.C * p.det_inv;
        con_b = -p.B * p.det_inv;
        con_c = p.A * p.det_inv;
        const float mid = 0.5f * (p.A + p.C);
        const float lam1 = mid + sqrtf(fmaxf(mid * mid - p.det, 0.1f));
        const float radius_f = ceilf(3.0f * sqrtf(lam1));
        const float ndc_x = p.hx * p.m_w;
        const float ndc_y = p.hy * p.m_w;
        px = ((ndc_x + 1.0f) * (float)f.W - 1.0f) * 0.5f;
        py = ((ndc_y + 1.0f) * (float)f.H - 1.0f) * 0.5f;
        int minx, miny, maxx, maxy;
        tile_rect(px, py, radius_f, f.gx, f.gy, minx, miny, maxx, maxy);
        const int tiles = (maxx - minx) * (maxy - miny);
        ok = tiles > 0;
        if (ok) {
            my_tiles = (uint32_t)tiles;
            radius_i = (int)fminf(fmaxf(radius_f, 0.0f), 2.0e9f);
            rect = make_uint2((uint32_t)minx | ((uint32_t)miny << 16),
                              (uint32_t)(maxx - minx) | ((uint32_t)(maxy - miny) << 16));
            tz = p.tz;
        }
    }
    // the visible lanes' SH records: loads issued here, consumed behind `between`.  No branch around the loads (the array
    // must not cross control flow): lanes without a visible Gaussian read record 0 — one cache line for all of them
    constexpr bool has_colors = DEG < 0;
    constexpr int K = has_colors ? 1 : (DEG + 1) * (DEG + 1);
    float sh[(3 * K + 3) / 4 * 4];
#ifdef SCG_PROBE_TIMELINE
    { float pin = con_a; asm volatile("" : "+v"(pin)); }          // the cull / covariance chain is done (inputs have arrived)
    const uint32_t tp_math = (uint32_t)wall_clock64();
#endif
    if (!has_colors) src.template load_sh<K>(ok ? i : 0, sh);
    between(rect);
#ifdef SCG_PROBE_TIMELINE
    const uint32_t tp_between = (uint32_t)wall_clock64();
    asm volatile("" : "+v"(sh[0]), "+v"(sh[(3 * K + 3) / 4 * 4 - 1]));        // the SH record has arrived
    const uint32_t tp_sh = (uint32_t)wall_clock64();
    g_tp[0] += tp_math - g_tp[4]; g_tp[1] += tp_between - tp_math; g_tp[2] += tp_sh - tp_between; g_tp[4] = tp_sh;
#endif
    if (ok) {
        float rgb[3];
        if (has_colors) {
            rgb[0] = in.rgb[0]; rgb[1] = in.rgb[1]; rgb[2] = in.rgb[2];
        } else {
            float dx = in.x - f.campos[0], dy = in.y - f.campos[1], dz = in.z - f.campos[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx = dx / len; dy = dy / len; dz = dz / len;
            eval_sh<(DEG < 0 ? 0 : DEG)>(sh, dx, dy, dz, rgb);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rgb[c] = rgb[c] + 0.5f;
                if (rgb[c] < 0.0f) { clamp_bits |= (uint8_t)(1u << c); rgb[c] = 0.0f; }
            }
        }
        sa = make_float4(px, py, con_a, con_b);
        // [6] cut-off of the conic's quadratic form for alpha >= 1/255 (with the blend kernels' safety margin:
        // 0.1 % + 0.01), [7] slope of the minimiser along a vertical edge — both only steer the blend
        // kernels' conservative 8x8-quadrant culling, never a blended value
        const float opa = in.opacity;
        sb = make_float4(con_c, opa, 2.0f * logf(255.0f * opa) * 1.001f + 0.01f, -con_b / con_c);
        sc = make_float4(rgb[0], rgb[1], rgb[2], tz);
    }
    // The outputs of the wave's 64 Gaussians are PARKED in its LDS stage; flush_stage() stores them — the 48-byte splat
    // records (contiguous in memory for the wave) as three fully coalesced 16-byte stores per lane.
    const int lane = lane_id();
    stage[3 * lane + 0] = sa;
    stage[3 * lane + 1] = sb;
    stage[3 * lane + 2] = sc;
    reinterpret_cast<uint4*>(stage + 3 * kWave)[lane] = make_uint4((uint32_t)radius_i, (uint32_t)clamp_bits, rect.x, rect.y);
    return my_tiles;
}

template <int DEG, class Src>
__device__ __forceinline__ void geometry_forward_body(
    const FrameDev& f, const Src& src, float4* __restrict__ splats,
    int32_t* __restrict__ radii, uint8_t* __restrict__ clamped, uint2* __restrict__ rects,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_wave_sum[kBlock / kWave];
    __shared__ float4 s_stage[(kBlock / kWave) * kStageVec];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < f.P;
    const GeoIn in = src.template load<(DEG < 0)>(valid ? i : f.P - 1);
    const Mat16 V = load16(f.view);
    const Mat16 PM = load16(f.proj);
    float4* stage = s_stage + kStageVec * wave_id();
#ifdef SCG_PROBE_TIMELINE
    uint32_t tp_dummy[5] = {0u, 0u, 0u, 0u, 0u};
#endif
    const uint32_t my_tiles = geometry_forward_one<DEG>(f, V, PM, i, valid, in, src, stage,
                                                        [](uint2) {}
#ifdef SCG_PROBE_TIMELINE
                                                        , tp_dummy
#endif
                                                        );
    flush_stage(stage, i - lane_id(), f.P, splats, radii, clamped, rects, depth_keys);

    // per-block sum of tiles_touched: first phase of the inclusive scan, fused here
    uint32_t s = my_tiles;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
    if (lane_id() == 0) s_wave_sum[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_wave_sum[0] + s_wave_sum[1] + s_wave_sum[2] + s_wave_sum[3];
}

template <int DEG>
__global__ __launch_bounds__(kBlock) void geometry_forward_kernel(
    FrameDev f, const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, float4* __restrict__ splats,
    int32_t* __restrict__ radii, uint8_t* __restrict__ clamped, uint2* __restrict__ rects,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums, int sh_vec16) {
    const TensorSource src{means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, f.M, sh_vec16};
    geometry_forward_body<DEG>(f, src, splats, radii, clamped, rects, depth_keys, block_sums);
}

template <int DEG>
__global__ __launch_bounds__(kBlock) void geometry_forward_model_kernel(
    FrameDev f, ScgModel model, float4* __restrict__ splats, int32_t* __restrict__ radii, uint8_t* __restrict__ clamped,
    uint2* __restrict__ rects, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums) {
    const ModelSource src{model};
    geometry_forward_body<DEG>(f, src, splats, radii, clamped, rects, depth_keys, block_sums);
}

// The one-call path's variant: the tile histogram of the tile-first binning (binning_tiles.hip: table[B][Tn]) is built
// WHERE THE RECTANGLES ARE PRODUCED.  Workgroup b of B (16 waves) owns the 256-Gaussian blocks block_slice(b) — the slices
// the scatter kernel walks again —, its waves take the blocks' 64-Gaussian chunks in turn, and every lane drops its
// rectangle's tiles into the workgroup's LDS histogram while the loads of its SH record are in flight: tile_hist_kernel, its
// launch and its re-read of the rectangles are gone.  block_sums as above (one per 256 Gaussians, summed through LDS).
// (Round 5 tried fetching a wave's NEXT chunk's inputs — means3D / opacity / scale / rotation, 11 registers — a chunk ahead:
// nothing at S2 / S4, the registers spilled once the splat stage was in: dropped.)
// dynamic LDS of geometry_hist_kernel: [n_tiles] histogram, [max_blocks] block sums, then the sixteen waves' output stages
__host__ __device__ __forceinline__ size_t geometry_hist_stage_offset(int n_tiles, int max_blocks) {
    return ((size_t)(n_tiles + max_blocks) * sizeof(uint32_t) + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t geometry_hist_lds_bytes(int n_tiles, int max_blocks) {
    return geometry_hist_stage_offset(n_tiles, max_blocks) + (size_t)(kBinThreads / kWave) * kStageVec * sizeof(float4);
}

template <int DEG, class Src>
__device__ __forceinline__ void geometry_hist_body(
    const FrameDev& f, const Src& src, float4* __restrict__ splats,
    int32_t* __restrict__ radii, uint8_t* __restrict__ clamped, uint2* __restrict__ rects,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ table, uint32_t* __restrict__ class_counts, uint32_t* __restrict__ len_hist, int max_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const int n_tiles = f.gx * f.gy;
    uint32_t* s_blk = hist + n_tiles;                          // tiles_touched of this workgroup's 256-Gaussian blocks
    // ... behind them (16-byte aligned) the waves' output stages: 4 KiB each
    float4* stage = reinterpret_cast<float4*>(smem + geometry_hist_stage_offset(n_tiles, max_blocks)) + kStageVec * wave_id();
    uint32_t blk_a, blk_b;
    block_slice((uint32_t)f.P, gridDim.x, blockIdx.x, blk_a, blk_b);
    const int lane = lane_id();
    uint32_t chunk = 4u * blk_a + (uint32_t)wave_id();
    auto fetch = [&](uint32_t c) {
        const int g = (int)(c * kWave) + lane;
        return src.template load<(DEG < 0)>(min(g, f.P - 1));
    };
    const Mat16 V = load16(f.view);
    const Mat16 PM = load16(f.proj);
    if (blockIdx.x == 0) {                                     // counters of the later kernels of the binning stage
        if (threadIdx.x < 4) class_counts[threadIdx.x] = 0u;
        len_hist[threadIdx.x] = 0u;                            // 8 x 64 length histogram (+ as many unused words)
    }
    for (int t = threadIdx.x; t < n_tiles + max_blocks; t += kBinThreads) hist[t] = 0;
    // The workgroup's 64-Gaussian chunks are handed out by an LDS counter (round 5): every wave starts on the chunk of its number
    // and takes the next free one when it is done — chunks differ (visible Gaussians, rectangle sizes) and a wave has only two
    // to four of them: with a fixed assignment the workgroup waited 6-7 us for its slowest wave (timeline: mean wave 42 us,
    // slowest 49 at a million Gaussians).  The counter sits in the last (spare) word of the block-sum area.
    uint32_t* next_chunk = s_blk + (max_blocks - 1);
    __syncthreads();
    if (threadIdx.x == 0) *next_chunk = 4u * blk_a + (uint32_t)kBinWaves;
    __syncthreads();
    int pending = -1;                                          // first Gaussian of the chunk parked in the stage, -1: none
#ifdef SCG_PROBE_TIMELINE
    uint32_t g_tp[5] = {0u, 0u, 0u, 0u, (uint32_t)wall_clock64()};
    const uint32_t tp_begin = g_tp[4];
    uint32_t tp_iters = 0;
#endif
    while (chunk < 4u * blk_b) {
        const int i = (int)(chunk * kWave) + lane;
        const GeoIn in = fetch(chunk);
        // (the next chunk's number: asked for now, needed at the end of this one)
        uint32_t grabbed = 0;
        if (lane == 0) grabbed = atomicAdd(next_chunk, 1u);
        const uint32_t my_tiles = geometry_forward_one<DEG>(
            f, V, PM, i, i < f.P, in, src, stage, [&](uint2 rect) {
                // (the loads of this chunk's SH records are in flight: now the previous chunk's outputs leave, then the histogram)
                if (pending >= 0) flush_stage(stage, pending, f.P, splats, radii, clamped, rects, depth_keys);
                walk_rects(rect, (uint32_t)i, f.gx, [&](uint32_t tile, uint32_t) { atomicAdd(&hist[tile], 1u); });
            }
#ifdef SCG_PROBE_TIMELINE
            , g_tp
#endif
            );
#ifdef SCG_PROBE_TIMELINE
        { const uint32_t t = (uint32_t)wall_clock64(); g_tp[3] += t - g_tp[4]; g_tp[4] = t; ++tp_iters; }
#endif
        pending = i - lane;
        uint32_t s = my_tiles;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, kWave);
        if (lane == 0 && s) atomicAdd(&s_blk[(chunk >> 2) - blk_a], s);
        chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)grabbed);
    }
    if (pending >= 0) flush_stage(stage, pending, f.P, splats, radii, clamped, rects, depth_keys);
#ifdef SCG_PROBE_TIMELINE
    const uint32_t tp_loop_end = (uint32_t)wall_clock64();
#endif
    __syncthreads();
#ifdef SCG_PROBE_TIMELINE
    if (f.cost_out && lane == 0) {
        uint32_t* tl = f.cost_out + n_tiles + 65792 + ((size_t)blockIdx.x * kBinWaves + wave_id()) * 8;   // (behind the scatter's log)
        tl[0] = tp_begin; tl[1] = g_tp[0]; tl[2] = g_tp[1]; tl[3] = g_tp[2]; tl[4] = g_tp[3]; tl[5] = tp_loop_end;
        tl[6] = (uint32_t)wall_clock64(); tl[7] = 0xC0FFEE00u | tp_iters;
    }
#endif
    uint32_t* row = table + (size_t)blockIdx.x * n_tiles;
    for (int t = threadIdx.x; t < n_tiles; t += kBinThreads) row[t] = hist[t];
    for (uint32_t k = threadIdx.x; k < blk_b - blk_a; k += kBinThreads) block_sums[blk_a + k] = s_blk[k];
}

template <int DEG>
__global__ __launch_bounds__(kBinThreads) void geometry_hist_kernel(
    FrameDev f, const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, float4* __restrict__ splats,
    int32_t* __restrict__ radii, uint8_t* __restrict__ clamped, uint2* __restrict__ rects,
    uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums, int sh_vec16,
    uint32_t* __restrict__ table, uint32_t* __restrict__ class_counts, uint32_t* __restrict__ len_hist, int max_blocks) {
    const TensorSource src{means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, f.M, sh_vec16};
    geometry_hist_body<DEG>(f, src, splats, radii, clamped, rects, depth_keys, block_sums, table, class_counts, len_hist, max_blocks);
}

// ... reading the reference model's raw parameter tensors (ScgModel): no activation / concatenation launches in front of it
template <int DEG>
__global__ __launch_bounds__(kBinThreads) void geometry_hist_model_kernel(
    FrameDev f, ScgModel model, float4* __restrict__ splats, int32_t* __restrict__ radii, uint8_t* __restrict__ clamped,
    uint2* __restrict__ rects, uint32_t* __restrict__ depth_keys, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ table, uint32_t* __restrict__ class_counts, uint32_t* __restrict__ len_hist, int max_blocks) {
    const ModelSource src{model};
    geometry_hist_body<DEG>(f, src, splats, radii, clamped, rects, depth_keys, block_sums, table, class_counts, len_hist, max_blocks);
}

// ---------------------------------------------------------------------------------------------------
// backward kernel
// ---------------------------------------------------------------------------------------------------
// The K = (DEG + 1)^2 basis values at a unit direction and their derivatives by its three components (the polynomials of
// utils/sh_utils.py:57-103 and their analytic gradients).
template <int DEG>
__device__ __forceinline__ void sh_basis_grads(float x, float y, float z, float* basis, float* bx, float* by, float* bz) {
    basis[0] = kC0; bx[0] = by[0] = bz[0] = 0.f;
    if (DEG > 0) {
        basis[1] = -kC1 * y; bx[1] = 0.f; by[1] = -kC1; bz[1] = 0.f;
        basis[2] = kC1 * z;  bx[2] = 0.f; by[2] = 0.f; bz[2] = kC1;
        basis[3] = -kC1 * x; bx[3] = -kC1; by[3] = 0.f; bz[3] = 0.f;
    }
    if (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        basis[4] = kC2[0] * xy;                    bx[4] = kC2[0] * y; by[4] = kC2[0] * x; bz[4] = 0.f;
        basis[5] = kC2[1] * yz;                    bx[5] = 0.f; by[5] = kC2[1] * z; bz[5] = kC2[1] * y;
        basis[6] = kC2[2] * (2.0f * zz - xx - yy); bx[6] = kC2[2] * -2.0f * x; by[6] = kC2[2] * -2.0f * y; bz[6] = kC2[2] * 4.0f * z;
        basis[7] = kC2[3] * xz;                    bx[7] = kC2[3] * z; by[7] = 0.f; bz[7] = kC2[3] * x;
        basis[8] = kC2[4] * (xx - yy);             bx[8] = kC2[4] * 2.0f * x; by[8] = kC2[4] * -2.0f * y; bz[8] = 0.f;
        if (DEG > 2) {
            basis[9] = kC3[0] * y * (3.0f * xx - yy);
            bx[9] = kC3[0] * 6.0f * xy; by[9] = kC3[0] * (3.0f * xx - 3.0f * yy); bz[9] = 0.f;
            basis[10] = kC3[1] * xy * z;
            bx[10] = kC3[1] * yz; by[10] = kC3[1] * xz; bz[10] = kC3[1] * xy;
            basis[11] = kC3[2] * y * (4.0f * zz - xx - yy);
            bx[11] = kC3[2] * -2.0f * xy; by[11] = kC3[2] * (4.0f * zz - xx - 3.0f * yy); bz[11] = kC3[2] * 8.0f * yz;
            basis[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            bx[12] = kC3[3] * -6.0f * xz; by[12] = kC3[3] * -6.0f * yz; bz[12] = kC3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            basis[13] = kC3[4] * x * (4.0f * zz - xx - yy);
            bx[13] = kC3[4] * (4.0f * zz - 3.0f * xx - yy); by[13] = kC3[4] * -2.0f * xy; bz[13] = kC3[4] * 8.0f * xz;
            basis[14] = kC3[5] * z * (xx - yy);
            bx[14] = kC3[5] * 2.0f * xz; by[14] = kC3[5] * -2.0f * yz; bz[14] = kC3[5] * (xx - yy);
            basis[15] = kC3[6] * x * (xx - 3.0f * yy);
            bx[15] = kC3[6] * (3.0f * xx - 3.0f * yy); by[15] = kC3[6] * -6.0f * xy; bz[15] = 0.f;
        }
    }
}

// d(rgb)/d(sh) and d(rgb)/d(dir) for all active coefficients.  dsh_out may be nullptr-free: always written.
// STREAM: the coefficients are read one at a time from `rec` right where they are used instead of all 3K up front (the
// record sits in LDS: nothing to batch, and 48 registers less) — same operations in the same order.
template <int DEG, bool STREAM = false>
__device__ __forceinline__ void sh_backward(const float* rec, bool vec16, float x, float y, float z,
                                            const float* dRGB, float* dsh_rec, int M,
                                            float& ddx, float& ddy, float& ddz, bool acc,     // rec may alias dsh_rec (LDS slot);
                                            int sw = -1) {                                     // acc: add to dsh_rec (global records only)
    // sw >= 0: the record's twelve 16-byte pieces are stored with the low two bits of the piece index XORed with sw (the
    // bank-conflict swizzle of the 192-byte LDS slots, see geometry_backward_kernel)
    auto at = [&](int j) { return sw < 0 ? j : ((((j >> 2) ^ sw) << 2) | (j & 3)); };
    constexpr int K = (DEG + 1) * (DEG + 1);
    float sh[STREAM ? 3 : (3 * K + 3) / 4 * 4];
    if (!STREAM) load_sh<K>(rec, vec16, sh);
    float basis[K];
    float bx[K], by[K], bz[K];
    sh_basis_grads<DEG>(x, y, z, basis, bx, by, bz);
    ddx = ddy = ddz = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s0, s1, s2;
        if (STREAM) { s0 = rec[at(3 * k)]; s1 = rec[at(3 * k + 1)]; s2 = rec[at(3 * k + 2)]; }
        else { s0 = sh[3 * k]; s1 = sh[3 * k + 1]; s2 = sh[3 * k + 2]; }
        const float g = s0 * dRGB[0] + s1 * dRGB[1] + s2 * dRGB[2];
        ddx += g * bx[k]; ddy += g * by[k]; ddz += g * bz[k];
        if (acc) {
            dsh_rec[at(3 * k + 0)] += basis[k] * dRGB[0];
            dsh_rec[at(3 * k + 1)] += basis[k] * dRGB[1];
            dsh_rec[at(3 * k + 2)] += basis[k] * dRGB[2];
        } else {
            dsh_rec[at(3 * k + 0)] = basis[k] * dRGB[0];
            dsh_rec[at(3 * k + 1)] = basis[k] * dRGB[1];
            dsh_rec[at(3 * k + 2)] = basis[k] * dRGB[2];
        }
    }
    if (!acc)
        for (int k = 3 * K; k < 3 * M; ++k) dsh_rec[at(k)] = 0.f;
}

// Chain rule from a Gaussian's gradient record (the blend backward's raw sums) to its position, opacity and covariance inputs:
// everything of the geometry backward except the colour part.  `in` holds ACTIVATED values (the model path's kernel applies the
// activations' derivatives afterwards); ga / gb are the record's first two float4 and are consumed.
__device__ __forceinline__ void geometry_backward_one(const FrameDev& f, const Mat16& V, const Mat16& PM, const GeoIn& in,
                                                      bool has_cov, float4& ga, float4& gb, float* dm, float* dm2, float& d_op,
                                                      float* ds, float* dq, float* dcov) {
    const float x = in.x, y = in.y, z = in.z;
    // the blend backward leaves RAW SUMS over the pixels (q = opacity G dL/dalpha, d = splat centre - pixel):
    //   [0] sum q dx  [1] sum q dy  [2] dL/ddepth  [3] sum q | [4] sum q dx^2  [5] sum q dx dy  [6] sum q dy^2 | [8..10] dL/drgb
    // with G = exp(-(a dx^2 + 2 b dx dy + c dy^2) / 2):
    //   dL/dx = -(a S_x + b S_y)   dL/dy = -(b S_x + c S_y)   dL/dopacity = S_q / opacity
    //   dL/da = -S_xx / 2          dL/db = -S_xy              dL/dc = -S_yy / 2
    const float opac = in.opacity;
    d_op = (opac > 0.0f) ? ga.w / opac : 0.0f;

    Proj p;
    project(f, V, PM, x, y, z, p);
    if (has_cov) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.cov[k] = in.a[k];
    } else {
        cov3d_from_scale_rot(in, f.mod, p);
    }
    cov2d(f, V, p);
    {
        const float con_a = p.C * p.det_inv, con_b = -p.B * p.det_inv, con_c = p.A * p.det_inv;
        const float sx = ga.x, sy = ga.y;
        ga.x = -(con_a * sx + con_b * sy);
        ga.y = -(con_b * sx + con_c * sy);
        gb.x *= -0.5f; gb.y = -gb.y; gb.z *= -0.5f;
    }

    // conic = inverse(cov2D):  a = C/det, b = -B/det, c = A/det
    const float di2 = p.det_inv * p.det_inv;
    const float dA = di2 * (-p.C * p.C * gb.x + p.B * p.C * gb.y - p.B * p.B * gb.z);
    const float dC = di2 * (-p.B * p.B * gb.x + p.A * p.B * gb.y - p.A * p.A * gb.z);
    const float dB = di2 * (2.0f * p.B * p.C * gb.x - (p.A * p.C + p.B * p.B) * gb.y + 2.0f * p.A * p.B * gb.z);

    // cov2D = Tm Sigma Tm^T  ->  dSigma (full symmetric) and dTm
    const float T0[3] = {p.T00, p.T01, p.T02};
    const float T1[3] = {p.T10, p.T11, p.T12};
    float Ms[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            Ms[3 * j + k] = dA * T0[j] * T0[k] + 0.5f * dB * (T0[j] * T1[k] + T1[j] * T0[k]) + dC * T1[j] * T1[k];
    const float u[3] = {p.u0, p.u1, p.u2};
    const float v[3] = {p.v0, p.v1, p.v2};
    float dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dT0[k] = 2.0f * dA * u[k] + dB * v[k];
        dT1[k] = dB * u[k] + 2.0f * dC * v[k];
    }
    // Tm = J . Wm,  Wm[i][j] = V[4j+i]
    const float dJ00 = dT0[0] * V.m[0] + dT0[1] * V.m[4] + dT0[2] * V.m[8];
    const float dJ02 = dT0[0] * V.m[2] + dT0[1] * V.m[6] + dT0[2] * V.m[10];
    const float dJ11 = dT1[0] * V.m[1] + dT1[1] * V.m[5] + dT1[2] * V.m[9];
    const float dJ12 = dT1[0] * V.m[2] + dT1[1] * V.m[6] + dT1[2] * V.m[10];
    const float itz = 1.0f / p.tz;
    const float itz2 = itz * itz;
    const float itz3 = itz2 * itz;
    const float dtx = p.cl_x ? 0.f : -f.focal_x * itz2 * dJ02;
    const float dty = p.cl_y ? 0.f : -f.focal_y * itz2 * dJ12;
    float dtz = -f.focal_x * itz2 * dJ00 - f.focal_y * itz2 * dJ11 +
                2.0f * f.focal_x * p.t_x * itz3 * dJ02 + 2.0f * f.focal_y * p.t_y * itz3 * dJ12;
    dtz += ga.z;                                        // depth = view z
#pragma unroll
    for (int j = 0; j < 3; ++j)
        dm[j] = V.m[4 * j] * dtx + V.m[4 * j + 1] * dty + V.m[4 * j + 2] * dtz;

    // pixel -> NDC -> mean3D
    const float dndc_x = ga.x * 0.5f * (float)f.W;
    const float dndc_y = ga.y * 0.5f * (float)f.H;
    dm2[0] = dndc_x; dm2[1] = dndc_y;
    const float mw2 = p.m_w * p.m_w;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dm[j] += (PM.m[4 * j] * p.m_w - PM.m[4 * j + 3] * p.hx * mw2) * dndc_x +
                 (PM.m[4 * j + 1] * p.m_w - PM.m[4 * j + 3] * p.hy * mw2) * dndc_y;
    }

    // cov3D -> (scales, rotations) or the precomputed input (before the colour part: the rotation / scale matrices and
    // dSigma are dead by the time the SH record is worked on — register pressure)
    if (has_cov) {
        dcov[0] = Ms[0]; dcov[1] = 2.0f * Ms[1]; dcov[2] = 2.0f * Ms[2];
        dcov[3] = Ms[4]; dcov[4] = 2.0f * Ms[5]; dcov[5] = Ms[8];
    } else {
        const float* L = p.L;
        const float* R = p.R;
        const float S[3] = {f.mod * in.a[4], f.mod * in.a[5], f.mod * in.a[6]};
        float dL[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                dL[3 * a + b] = 2.0f * (Ms[3 * a + 0] * L[0 + b] + Ms[3 * a + 1] * L[3 + b] + Ms[3 * a + 2] * L[6 + b]);
        float dR[9];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            ds[b] = f.mod * (dL[b] * R[b] + dL[3 + b] * R[3 + b] + dL[6 + b] * R[6 + b]);
            dR[b] = dL[b] * S[b]; dR[3 + b] = dL[3 + b] * S[b]; dR[6 + b] = dL[6 + b] * S[b];
        }
        const float r = in.a[0], qx = in.a[1], qy = in.a[2], qz = in.a[3];
        dq[0] = 2.0f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
        dq[1] = 2.0f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.0f * qx * dR[4] - r * dR[5] + qz * dR[6] + r * dR[7] - 2.0f * qx * dR[8]);
        dq[2] = 2.0f * (-2.0f * qy * dR[0] + qx * dR[1] + r * dR[2] + qx * dR[3] + qz * dR[5] - r * dR[6] + qz * dR[7] - 2.0f * qy * dR[8]);
        dq[3] = 2.0f * (-2.0f * qz * dR[0] - r * dR[1] + qx * dR[2] + r * dR[3] - 2.0f * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
    }
}

// STAGED (SH path with the usual 16-coefficient records): the 192-byte SH record of a Gaussian is 12 x 16 bytes at a
// 192-byte stride between threads, and its gradient record was written as 48 scalar stores per thread — every memory
// instruction of a wave touched 64 different cache lines.  Here the workgroup moves its records (contiguous in memory)
// between HBM and LDS with fully coalesced 16-byte accesses, and each thread works on its own record in LDS.
// What runs (BLOCK = 64): single-wave workgroups of SIXTY Gaussians, 192-byte slots whose 16-byte pieces are XOR-swizzled
// against bank conflicts — fourteen workgroups = 840 Gaussians in flight per compute unit (see RECS in the kernel); the
// 256-thread form (13-float4 slot stride: conflict-free without a swizzle, 53 KiB, three workgroups = 768 Gaussians per
// compute unit) is what round 2 shipped and what the template still instantiates for BLOCK = 256.
constexpr int kShVec = 12;                     // float4 per 16-coefficient record
constexpr int kShSlot = 13;                    // LDS slot stride in float4 (256-thread workgroups)

template <bool STAGED, int BLOCK, bool STAGE_IN = STAGED>
__global__ __launch_bounds__(BLOCK) void geometry_backward_kernel(
    FrameDev f, const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped, const float4* __restrict__ dsplats,
    float* __restrict__ dmeans3D, float* __restrict__ dmeans2D, float* __restrict__ dopac,
    float* __restrict__ dshs, float* __restrict__ dcolors, float* __restrict__ dscales, float* __restrict__ drots,
    float* __restrict__ dcov3D, int sh_vec16, int flags) {
    const int accumulate = flags & SCG_BACKWARD_ACCUMULATE;
    // SCG_BACKWARD_SH_TAIL_ZERO: the records' coefficients above the active degree already hold zeros (the caller's promise about
    // its own gradient buffer): the staged form then stores only the 16-byte pieces that hold an active coefficient — one of the
    // record's twelve at degree 0, three at degree 1, seven at degree 2
    const int live_pieces = (flags & SCG_BACKWARD_SH_TAIL_ZERO) ? (3 * (f.D + 1) * (f.D + 1) + 3) / 4 : kShVec;
    // accumulate != 0: every parameter gradient is ADDED to what the output buffers hold (a second view of the same
    // Gaussians in one training step: no separate add pass over 236 bytes per Gaussian); dmeans2D is per view and
    // always overwritten.  A Gaussian is owned by one thread: plain read-add-write, no atomics.
    constexpr int kSlot = (BLOCK == kWave) ? kShVec : kShSlot;
    // Gaussians per workgroup.  The single-wave form takes SIXTY, not 64: LDS is handed out in 1 280-byte granules, 64 slots
    // of 192 bytes are ten of them and only twelve such workgroups fit a compute unit (768 Gaussians — what three 256-thread
    // workgroups hold); 60 slots are nine granules, fourteen workgroups, 840 Gaussians in flight per compute unit, and the
    // benchmark's 200 000 (S2) fit one round of workgroups (measured: 23.1 us at 196 608 Gaussians, 28.0 at 197 632 before).
    constexpr int RECS = (BLOCK == kWave) ? 60 : BLOCK;
    __shared__ float4 s_sh[STAGED ? RECS * kSlot : 1];
    // 16-byte piece idx of the workgroup's records -> its place in LDS.  The 192-byte slots of the single-wave form would put
    // lanes t, t + 4, t + 8, ... on the same banks (192 t mod 256 has period 4): the low two bits of the piece index are XORed
    // with (t >> 2) & 3, which spreads every group of sixteen such lanes over four bank groups (what the 208-byte slots do).
    constexpr bool kSwizzle = (BLOCK == kWave);
    auto slot_piece = [&](int idx) {
        const int r = idx / kShVec, pc = idx % kShVec;
        return r * kSlot + (kSwizzle ? (pc ^ ((r >> 2) & 3)) : pc);
    };
    const int sw = kSwizzle ? (((int)threadIdx.x >> 2) & 3) : -1;
    const int block_first = blockIdx.x * RECS;
    const bool owner = (int)threadIdx.x < RECS;                        // (the other lanes only help moving the records)
    const int i = owner ? block_first + (int)threadIdx.x : f.P;
    const int n_vec = min(RECS, f.P - block_first) * kShVec;           // float4 of this workgroup's records
    if (!STAGED && i >= f.P) return;
    // Round 5: EVERY load of the workgroup is issued here, before anything is waited for — the twelve coalesced pieces of the
    // SH records and each thread's own inputs (radius, position, gradient record, opacity, scale, rotation, clamp bits).
    // Round 4's kernel waited for each staged piece before it fetched the next (a store to LDS behind every load) and fetched
    // the per-Gaussian inputs where they were first used: ~17 dependent round trips to HBM per workgroup.
    // STAGE_IN: the SH records go through LDS on their way in as well (degree 3; lower degrees read a short prefix of the
    // record: not worth staging — a compile-time choice, so that `stage` stays in registers)
    constexpr bool stage_sh = STAGED && STAGE_IN;
    float4 stage[stage_sh ? kShVec : 1];
    if (stage_sh) {
        const float4* src = reinterpret_cast<const float4*>(shs) + (size_t)block_first * kShVec;
#pragma unroll
        for (int k = 0; k < kShVec; ++k) stage[k] = src[min(k * BLOCK + (int)threadIdx.x, n_vec - 1)];
    }
    const int ic = min(i, f.P - 1);                    // (lanes without a Gaussian load somebody's: no branch around the loads)
    int radius = radii[ic];
    GeoIn in = load_geo_in<false>(ic, means3D, opacities, nullptr, scales, rotations, cov3D_precomp);
    constexpr int kRec = SCG_DSPLAT_FLOATS / 4;            // float4s per gradient record (one 64-byte line)
    float4 ga = dsplats[kRec * (size_t)ic + 0];
    float4 gb = dsplats[kRec * (size_t)ic + 1];
    float4 gc = dsplats[kRec * (size_t)ic + 2];   // d/drgb
    uint8_t cb = clamped[ic];
    const Mat16 V = load16(f.view);
    const Mat16 PM = load16(f.proj);
    float cam_x = f.campos[0], cam_y = f.campos[1], cam_z = f.campos[2];
    // (nothing moves across: the scheduler would otherwise trade loads in flight for registers and issue the per-thread
    // loads behind the LDS writes of the staged pieces)
    __builtin_amdgcn_sched_barrier(0);
    if (STAGED) {
        if (stage_sh) {
#pragma unroll
            for (int k = 0; k < kShVec; ++k)      // (no branch: a lane beyond the last piece rewrites that piece with its own value —
                s_sh[slot_piece(min(k * BLOCK + (int)threadIdx.x, n_vec - 1))] = stage[k];    // a branch makes the compiler sink
                                                                                              // each load behind it, one wait per piece)
        }
    }
    // the per-thread inputs are USED here, in front of the branch they are needed in: the compiler otherwise sinks their loads
    // into it, behind the wait for the staged pieces (a second round trip to HBM)
    {
        int rr = radius, cc = cb;
        asm volatile("" : "+v"(rr), "+v"(cc), "+v"(ga.x), "+v"(ga.y), "+v"(ga.z), "+v"(ga.w), "+v"(gb.x), "+v"(gb.y), "+v"(gb.z));
        radius = rr; cb = (uint8_t)cc;
        asm volatile("" : "+v"(gc.x), "+v"(gc.y), "+v"(gc.z), "+v"(in.x), "+v"(in.y), "+v"(in.z), "+v"(in.opacity));
        asm volatile("" : "+v"(in.a[0]), "+v"(in.a[1]), "+v"(in.a[2]), "+v"(in.a[3]), "+v"(in.a[4]), "+v"(in.a[5]), "+v"(in.a[6]));
        asm volatile("" : "+v"(cam_x), "+v"(cam_y), "+v"(cam_z));
    }
    if (STAGED) __syncthreads();
    float* my_slot = reinterpret_cast<float*>(&s_sh[STAGED ? min((int)threadIdx.x, RECS - 1) * kSlot : 0]);

    float dm[3] = {0.f, 0.f, 0.f};
    float dm2[2] = {0.f, 0.f};
    float d_op = 0.f;
    float ds[3] = {0.f, 0.f, 0.f};
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dcol[3] = {0.f, 0.f, 0.f};
    bool sh_written = false;

    if (i < f.P && radius > 0) {
        const float x = in.x, y = in.y, z = in.z;
        geometry_backward_one(f, V, PM, in, cov3D_precomp != nullptr, ga, gb, dm, dm2, d_op, ds, dq, dcov);

        // colour
        if (colors_precomp) {
            dcol[0] = gc.x; dcol[1] = gc.y; dcol[2] = gc.z;
        } else {
            const float dRGB[3] = {(cb & 1) ? 0.f : gc.x, (cb & 2) ? 0.f : gc.y, (cb & 4) ? 0.f : gc.z};
            float dx = x - cam_x, dy = y - cam_y, dz = z - cam_z;
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            const float ilen = 1.0f / len;
            dx *= ilen; dy *= ilen; dz *= ilen;
            float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f;
            const float* rec = stage_sh ? my_slot : shs + (size_t)i * f.M * 3;
            float* drec = STAGED ? my_slot : dshs + (size_t)i * f.M * 3;
            switch (f.D) {
                case 0: sh_backward<0>(rec, sh_vec16, dx, dy, dz, dRGB, drec, f.M, gx_, gy_, gz_, !STAGED && accumulate,
                                        STAGED ? sw : -1); break;
                case 1: sh_backward<1>(rec, sh_vec16, dx, dy, dz, dRGB, drec, f.M, gx_, gy_, gz_, !STAGED && accumulate,
                                        STAGED ? sw : -1); break;
                case 2: sh_backward<2>(rec, sh_vec16, dx, dy, dz, dRGB, drec, f.M, gx_, gy_, gz_, !STAGED && accumulate,
                                        STAGED ? sw : -1); break;
                default: sh_backward<3, stage_sh>(rec, sh_vec16, dx, dy, dz, dRGB, drec, f.M, gx_, gy_, gz_, !STAGED && accumulate,
                                                STAGED ? sw : -1); break;
            }
            sh_written = true;
            // through dir = d / |d|
            const float dot = dx * gx_ + dy * gy_ + dz * gz_;
            dm[0] += (gx_ - dx * dot) * ilen;
            dm[1] += (gy_ - dy * dot) * ilen;
            dm[2] += (gz_ - dz * dot) * ilen;
        }
    }

    if (STAGED) {
        if (!sh_written && owner) {
#pragma unroll
            for (int k = 0; k < kShVec; ++k) s_sh[threadIdx.x * kSlot + k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dshs) + (size_t)block_first * kShVec;
#pragma unroll
        for (int k = 0; k < kShVec; ++k) {
            const int idx = k * BLOCK + (int)threadIdx.x;
            if (idx < n_vec && idx % kShVec < live_pieces) {
                float4 v = s_sh[slot_piece(idx)];
                if (accumulate) {
                    const float4 o = dst[idx];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                dst[idx] = v;
            }
        }
        // the three (P, 3) outputs — position, screen-space position, scale gradients: 12-byte records, 4-byte stores at a
        // 12-byte stride when every thread stores its own — go through the same LDS (free now) and leave lane-contiguous
        __syncthreads();
        float* sm = reinterpret_cast<float*>(s_sh);
        constexpr int kArr = 3 * RECS;
        if (owner) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sm[3 * threadIdx.x + c] = dm[c];
                sm[kArr + 3 * threadIdx.x + c] = c < 2 ? dm2[c] : 0.f;
                sm[2 * kArr + 3 * threadIdx.x + c] = ds[c];
            }
        }
        __syncthreads();
        const int n3 = 3 * min(RECS, f.P - block_first);
        float* o_m3 = dmeans3D + 3 * (size_t)block_first;
        float* o_m2 = dmeans2D + 3 * (size_t)block_first;
        float* o_sc = dscales ? dscales + 3 * (size_t)block_first : nullptr;
#pragma unroll
        for (int k = 0; k < (kArr + BLOCK - 1) / BLOCK; ++k) {
            const int idx = k * BLOCK + (int)threadIdx.x;
            if (idx < n3) {
                float v = sm[idx], w = sm[2 * kArr + idx];
                if (accumulate) {                      // (a culled Gaussian adds zeros)
                    v += o_m3[idx];
                    if (o_sc) w += o_sc[idx];
                }
                o_m3[idx] = v;
                o_m2[idx] = sm[kArr + idx];
                if (o_sc) o_sc[idx] = w;
            }
        }
        if (i >= f.P) return;
    } else if (dshs && !sh_written && !accumulate) {
        float* drec = dshs + (size_t)i * f.M * 3;
        for (int k = 0; k < 3 * f.M; ++k) drec[k] = 0.f;
    }
    if (!STAGED) {
        dmeans2D[3 * (size_t)i + 0] = dm2[0];
        dmeans2D[3 * (size_t)i + 1] = dm2[1];
        dmeans2D[3 * (size_t)i + 2] = 0.f;
    }
    if (accumulate) {
        if (radius <= 0) return;                   // nothing to add
        if (!STAGED) {
            dm[0] += dmeans3D[3 * (size_t)i + 0]; dm[1] += dmeans3D[3 * (size_t)i + 1]; dm[2] += dmeans3D[3 * (size_t)i + 2];
        }
        d_op += dopac[i];
        if (dcolors) {
            dcol[0] += dcolors[3 * (size_t)i + 0]; dcol[1] += dcolors[3 * (size_t)i + 1]; dcol[2] += dcolors[3 * (size_t)i + 2];
        }
        if (dscales) {
            if (!STAGED) {
                ds[0] += dscales[3 * (size_t)i + 0]; ds[1] += dscales[3 * (size_t)i + 1]; ds[2] += dscales[3 * (size_t)i + 2];
            }
            const float4 o = *reinterpret_cast<const float4*>(drots + 4 * (size_t)i);
            dq[0] += o.x; dq[1] += o.y; dq[2] += o.z; dq[3] += o.w;
        }
        if (dcov3D) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov[k] += dcov3D[6 * (size_t)i + k];
        }
    }
    if (!STAGED) {
        dmeans3D[3 * (size_t)i + 0] = dm[0];
        dmeans3D[3 * (size_t)i + 1] = dm[1];
        dmeans3D[3 * (size_t)i + 2] = dm[2];
    }
    dopac[i] = d_op;
    if (dcolors) {
        dcolors[3 * (size_t)i + 0] = dcol[0]; dcolors[3 * (size_t)i + 1] = dcol[1]; dcolors[3 * (size_t)i + 2] = dcol[2];
    }
    if (dscales) {
        if (!STAGED) {
            dscales[3 * (size_t)i + 0] = ds[0]; dscales[3 * (size_t)i + 1] = ds[1]; dscales[3 * (size_t)i + 2] = ds[2];
        }
        *reinterpret_cast<float4*>(drots + 4 * (size_t)i) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
    if (dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dcov3D[6 * (size_t)i + k] = dcov[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel of the model path: gradients of the reference model's RAW parameters (ScgModelGrads)
// ---------------------------------------------------------------------------------------------------
// Same shape as the staged kernel above — single-wave workgroups of SIXTY Gaussians, every load issued up front, the SH records
// and their gradients moved between HBM and LDS with coalesced 16-byte accesses — with three differences:
//   * a workgroup lies inside ONE set (ray-bound or background): the sets' tensors are separate allocations;
//   * the SH record arrives in two parts, features_dc (12 B) and features_rest (180 B), and the gradient leaves the same way:
//     LDS mirrors memory (3-word and 45-word record strides: odd, so a thread walking its own record meets no bank conflict);
//     60 records = 675 + 45 float4 = the same nine 1 280-byte LDS granules, fourteen workgroups per compute unit;
//   * the activations' derivatives (sigmoid, exp, normalize, rayd . dL/dxyz) are applied in registers before the stores.
constexpr int kModelRecs = 60;
constexpr int kRestFloats = 3 * kRestCoeffs;                               // 45 floats per features_rest record
constexpr int kRestVecs = kModelRecs * kRestFloats / 4;                    // 675 float4 per workgroup
constexpr int kRestRounds = (kRestVecs + kWave - 1) / kWave;               // 11 rounds of 64 lanes
constexpr int kDcVecs = kModelRecs * 3 / 4;                                // 45 float4 per workgroup

template <int DEG, bool STREAM>
__device__ __forceinline__ void sh_backward_model(const float* sh, float* dc_t, float* rest_t, float x, float y, float z,
                                                  const float* dRGB, float& ddx, float& ddy, float& ddz) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    float basis[K], bx[K], by[K], bz[K];
    sh_basis_grads<DEG>(x, y, z, basis, bx, by, bz);
    ddx = ddy = ddz = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float* slot = (k == 0) ? dc_t : rest_t + 3 * (k - 1);
        float s0, s1, s2;
        if (STREAM) { s0 = slot[0]; s1 = slot[1]; s2 = slot[2]; }
        else { s0 = sh[3 * k]; s1 = sh[3 * k + 1]; s2 = sh[3 * k + 2]; }
        const float g = s0 * dRGB[0] + s1 * dRGB[1] + s2 * dRGB[2];
        ddx += g * bx[k]; ddy += g * by[k]; ddz += g * bz[k];
        slot[0] = basis[k] * dRGB[0];
        slot[1] = basis[k] * dRGB[1];
        slot[2] = basis[k] * dRGB[2];
    }
    if (DEG > 0)                                                           // (degree 0 never touches the features_rest slots)
        for (int w = 3 * (K - 1); w < kRestFloats; ++w) rest_t[w] = 0.f;
}

// (amdgpu_waves_per_eu: LDS lets fourteen of these single-wave workgroups share a compute unit, i.e. 4 waves per SIMD = 128
// registers; left alone the compiler settled at 140 and twelve)
template <bool STAGE_IN>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(4, 4))) void geometry_backward_model_kernel(
    FrameDev f, ScgModel m, ScgModelGrads g, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
    const float4* __restrict__ dsplats, float* __restrict__ dmeans2D, int flags, int blocks_ray) {
    __shared__ float4 s_rest4[kRestVecs];
    __shared__ float4 s_dc4[kDcVecs];
    float* s_rest = reinterpret_cast<float*>(s_rest4);
    float* s_dc = reinterpret_cast<float*>(s_dc4);
    const bool accumulate = (flags & SCG_BACKWARD_ACCUMULATE) != 0;
    // SH gradients above the active degree: stored as zeros — unless they are known to hold zeros already, or are being added to
    const bool skip_tail = (flags & (SCG_BACKWARD_ACCUMULATE | SCG_BACKWARD_SH_TAIL_ZERO)) != 0;
    const bool ray = (int)blockIdx.x < blocks_ray;                         // workgroup-uniform: the set this workgroup lies in
    const ScgModelSet set = ray ? m.ray : m.bg;
    const ScgModelGradSet gs = ray ? g.ray : g.bg;
    const int first = ((int)blockIdx.x - (ray ? 0 : blocks_ray)) * kModelRecs;    // the workgroup's first Gaussian inside its set
    const int n = min(kModelRecs, set.count - first);
    const int t = (int)threadIdx.x;
    const bool owner = t < n;
    const int j = first + min(t, n - 1);                                   // (lanes without a Gaussian load the last one's inputs)
    const int i = (ray ? 0 : m.ray.count) + j;                             // index among all Gaussians
    const int n_rest = n * kRestFloats, nv = n_rest >> 2;                  // floats / whole float4 of the workgroup's rest records
    const int n_dc = n * 3, nvd = n_dc >> 2;
    const int K = (f.D + 1) * (f.D + 1);
    const int active = 3 * (K - 1);                                        // floats of a rest record that can be non-zero
    const ModelSource src{m};

    // ---- every load of the workgroup, issued before anything is waited for.  The SH records of a full workgroup go STRAIGHT
    // into LDS (global_load_lds_dwordx4: wave-uniform LDS base + lane x 16 bytes — exactly the linear image of memory the
    // kernel keeps): no staging registers (they pushed the kernel to 136 registers and 12 instead of 14 workgroups per compute
    // unit) and no ds_write pass.  A last, partial workgroup of a set (one per set) fills its LDS with a plain loop.
    const bool full = n == kModelRecs;
    if constexpr (STAGE_IN) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        if (full) {
            const float4* src4 = reinterpret_cast<const float4*>(set.features_rest + (size_t)first * kRestFloats);
#pragma unroll
            for (int k = 0; k < kRestRounds - 1; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)(src4 + k * kWave + t), (lptr_t)(s_rest4 + k * kWave), 16, 0, 0);
            if ((kRestRounds - 1) * kWave + t < kRestVecs)
                __builtin_amdgcn_global_load_lds((gptr_t)(src4 + (kRestRounds - 1) * kWave + t),
                                                 (lptr_t)(s_rest4 + (kRestRounds - 1) * kWave), 16, 0, 0);
            if (t < kDcVecs)
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const float4*>(set.features_dc + (size_t)first * 3) + t),
                                                 (lptr_t)s_dc4, 16, 0, 0);
        } else {
            for (int w = t; w < n_rest; w += kWave) s_rest[w] = set.features_rest[(size_t)first * kRestFloats + w];
            for (int w = t; w < n_dc; w += kWave) s_dc[w] = set.features_dc[(size_t)first * 3 + w];
        }
    }
    int radius = radii[i];
    ModelSource::Raw raw = src.load_raw(i);
    constexpr int kRec = SCG_DSPLAT_FLOATS / 4;
    float4 ga = dsplats[kRec * (size_t)i + 0];
    float4 gb = dsplats[kRec * (size_t)i + 1];
    float4 gc = dsplats[kRec * (size_t)i + 2];
    uint8_t cb = clamped[i];
    const Mat16 V = load16(f.view);
    const Mat16 PM = load16(f.proj);
    float cam_x = f.campos[0], cam_y = f.campos[1], cam_z = f.campos[2];
    __builtin_amdgcn_sched_barrier(0);
    {
        int rr = radius, cc = cb;
        asm volatile("" : "+v"(rr), "+v"(cc), "+v"(ga.x), "+v"(ga.y), "+v"(ga.z), "+v"(ga.w), "+v"(gb.x), "+v"(gb.y), "+v"(gb.z));
        radius = rr; cb = (uint8_t)cc;
        asm volatile("" : "+v"(gc.x), "+v"(gc.y), "+v"(gc.z), "+v"(raw.zv), "+v"(raw.logit), "+v"(raw.q.x), "+v"(raw.q.y), "+v"(raw.q.z), "+v"(raw.q.w));
        asm volatile("" : "+v"(raw.o[0]), "+v"(raw.o[1]), "+v"(raw.o[2]), "+v"(raw.d[0]), "+v"(raw.d[1]), "+v"(raw.d[2]));
        asm volatile("" : "+v"(raw.ls[0]), "+v"(raw.ls[1]), "+v"(raw.ls[2]), "+v"(cam_x), "+v"(cam_y), "+v"(cam_z));
    }
    __syncthreads();
    float den;
    const GeoIn in = ModelSource::activate(raw, &den);
    float* dc_t = s_dc + 3 * min(t, kModelRecs - 1);
    float* rest_t = s_rest + kRestFloats * min(t, kModelRecs - 1);

    float dm[3] = {0.f, 0.f, 0.f};
    float dm2[2] = {0.f, 0.f};
    float d_op = 0.f;
    float ds[3] = {0.f, 0.f, 0.f};
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dcov[6];
    bool sh_written = false;
    const bool visible = owner && radius > 0;
    if (visible) {
        geometry_backward_one(f, V, PM, in, false, ga, gb, dm, dm2, d_op, ds, dq, dcov);
        // ---- the activations' derivatives (reference scene/gaussian_model.py:37-51), applied right here: the activated values
        // and the quaternion's norm are dead before the colour part starts (registers: 14 workgroups per compute unit)
        //   opacity = sigmoid(l):        dL/dl = dL/do . o (1 - o)
        //   scale   = exp(s):            dL/ds = dL/dscale . scale
        //   rot     = q / max(|q|, eps): dL/dq = (dL/drot - rot (rot . dL/drot)) / |q|      (dL/drot / eps where the clamp is active)
        d_op = d_op * in.opacity * (1.0f - in.opacity);
        ds[0] *= in.a[4]; ds[1] *= in.a[5]; ds[2] *= in.a[6];
        {
            const bool clamped_norm = !(den > kNormalizeEps);
            const float dotq = clamped_norm ? 0.f : in.a[0] * dq[0] + in.a[1] * dq[1] + in.a[2] * dq[2] + in.a[3] * dq[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) dq[k] = (dq[k] - in.a[k] * dotq) / den;
        }
        const float dRGB[3] = {(cb & 1) ? 0.f : gc.x, (cb & 2) ? 0.f : gc.y, (cb & 4) ? 0.f : gc.z};
        float dx = in.x - cam_x, dy = in.y - cam_y, dz = in.z - cam_z;
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float ilen = 1.0f / len;
        dx *= ilen; dy *= ilen; dz *= ilen;
        float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f;
        if constexpr (STAGE_IN) {                                          // degree 3: the records are in LDS
            sh_backward_model<3, true>(nullptr, dc_t, rest_t, dx, dy, dz, dRGB, gx_, gy_, gz_);
        } else {                                                           // degrees 0-2: a short head of the record, read directly
            switch (f.D) {
                case 0: { float sh[4]; src.load_sh<1>(i, sh); sh_backward_model<0, false>(sh, dc_t, rest_t, dx, dy, dz, dRGB, gx_, gy_, gz_); break; }
                case 1: { float sh[12]; src.load_sh<4>(i, sh); sh_backward_model<1, false>(sh, dc_t, rest_t, dx, dy, dz, dRGB, gx_, gy_, gz_); break; }
                default: { float sh[28]; src.load_sh<9>(i, sh); sh_backward_model<2, false>(sh, dc_t, rest_t, dx, dy, dz, dRGB, gx_, gy_, gz_); break; }
            }
        }
        sh_written = true;
        const float dot = dx * gx_ + dy * gy_ + dz * gz_;
        dm[0] += (gx_ - dx * dot) * ilen;
        dm[1] += (gy_ - dy * dot) * ilen;
        dm[2] += (gz_ - dz * dot) * ilen;
    }
    if (owner && !sh_written) {                                            // a culled Gaussian: zero SH gradients
        dc_t[0] = dc_t[1] = dc_t[2] = 0.f;
        if (f.D > 0)
            for (int w = 0; w < kRestFloats; ++w) rest_t[w] = 0.f;
    }
    //   xyz = rayo + rayd zval:  dL/dzval = rayd . dL/dxyz   (background set: dL/dxyz itself).  The direction is read AGAIN here
    //   (it was loaded for the position at the top: an L2 hit) instead of being kept in registers across the colour part.
    float d_zval = 0.f;
    if (ray)
        d_zval = set.rayd[3 * (size_t)j + 0] * dm[0] + set.rayd[3 * (size_t)j + 1] * dm[1] + set.rayd[3 * (size_t)j + 2] * dm[2];
    const float d_logit = d_op;
    const float* d_ls = ds;
    const float* d_q = dq;

    // ---- SH gradients: LDS -> memory, coalesced
    __syncthreads();
    {
        float4* dst = reinterpret_cast<float4*>(gs.features_dc + (size_t)first * 3);
        if (t < nvd) {
            float4 v = s_dc4[t];
            if (accumulate) { const float4 o = dst[t]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            dst[t] = v;
        }
        if (4 * nvd + t < n_dc) {
            float* d1 = gs.features_dc + (size_t)first * 3 + 4 * nvd + t;
            *d1 = s_dc[4 * nvd + t] + (accumulate ? *d1 : 0.f);
        }
    }
    if (f.D > 0 || !skip_tail) {
        float4* dst = reinterpret_cast<float4*>(gs.features_rest + (size_t)first * kRestFloats);
        const bool zeros = f.D == 0;                                       // degree 0: nothing of it is in LDS, all of it is zero
#pragma unroll
        for (int k = 0; k < kRestRounds; ++k) {
            const int idx = k * kWave + t;
            // a float4 of the region holds words [w0, w0 + 4) of a record (the last ones may belong to the next record's head):
            // with the tails left alone it is stored only if one of them can be non-zero
            const int w0 = (4 * idx) % kRestFloats;
            const bool live = !skip_tail || w0 < active || w0 + 3 >= kRestFloats;
            if (idx < nv && live) {
                float4 v = zeros ? make_float4(0.f, 0.f, 0.f, 0.f) : s_rest4[idx];
                if (accumulate) { const float4 o = dst[idx]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                dst[idx] = v;
            }
        }
        if (4 * nv + t < n_rest) {
            float* d1 = gs.features_rest + (size_t)first * kRestFloats + 4 * nv + t;
            *d1 = (zeros ? 0.f : s_rest[4 * nv + t]) + (accumulate ? *d1 : 0.f);
        }
    }
    // ---- the (n, 3) outputs leave lane-contiguous through the same LDS: position (background set), screen-space position, scale
    __syncthreads();
    constexpr int kArr = 3 * kModelRecs;
    if (owner) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s_rest[3 * t + c] = dm[c];
            s_rest[kArr + 3 * t + c] = c < 2 ? dm2[c] : 0.f;
            s_rest[2 * kArr + 3 * t + c] = d_ls[c];
        }
    }
    __syncthreads();
    {
        float* o_xyz = ray ? nullptr : gs.xyz + 3 * (size_t)first;
        float* o_m2 = dmeans2D + 3 * ((size_t)(ray ? 0 : m.ray.count) + first);
        float* o_sc = gs.scaling + 3 * (size_t)first;
#pragma unroll
        for (int k = 0; k < (kArr + kWave - 1) / kWave; ++k) {
            const int idx = k * kWave + t;
            if (idx < n_dc) {
                float v = s_rest[idx], w = s_rest[2 * kArr + idx];
                if (accumulate) {
                    if (o_xyz) v += o_xyz[idx];
                    w += o_sc[idx];
                }
                if (o_xyz) o_xyz[idx] = v;
                o_m2[idx] = s_rest[kArr + idx];
                o_sc[idx] = w;
            }
        }
    }
    if (!owner) return;
    if (accumulate) {
        if (ray) gs.zval[j] += d_zval;
        gs.opacity[j] += d_logit;
        float4* dr = reinterpret_cast<float4*>(gs.rotation + 4 * (size_t)j);
        const float4 o = *dr;
        *dr = make_float4(o.x + d_q[0], o.y + d_q[1], o.z + d_q[2], o.w + d_q[3]);
    } else {
        if (ray) gs.zval[j] = d_zval;
        gs.opacity[j] = d_logit;
        *reinterpret_cast<float4*>(gs.rotation + 4 * (size_t)j) = make_float4(d_q[0], d_q[1], d_q[2], d_q[3]);
    }
}

// The model's activated getters (scg_model_activate): one thread per Gaussian, the device functions the kernels above use.
__global__ __launch_bounds__(kBlock) void model_activate_kernel(ScgModel m, int P, float* __restrict__ means3D,
                                                                float* __restrict__ opacities, float* __restrict__ scales,
                                                                float* __restrict__ rotations) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const ModelSource src{m};
    const GeoIn in = src.load<false>(i);
    if (means3D) { means3D[3 * (size_t)i] = in.x; means3D[3 * (size_t)i + 1] = in.y; means3D[3 * (size_t)i + 2] = in.z; }
    if (opacities) opacities[i] = in.opacity;
    if (scales) { scales[3 * (size_t)i] = in.a[4]; scales[3 * (size_t)i + 1] = in.a[5]; scales[3 * (size_t)i + 2] = in.a[6]; }
    if (rotations) *reinterpret_cast<float4*>(rotations + 4 * (size_t)i) = make_float4(in.a[0], in.a[1], in.a[2], in.a[3]);
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// degree -> kernel instantiation (-1: precomputed colours)
template <class F>
static inline void by_degree(int deg, F&& f) {
    switch (deg) {
        case -1: f(std::integral_constant<int, -1>()); break;
        case 0: f(std::integral_constant<int, 0>()); break;
        case 1: f(std::integral_constant<int, 1>()); break;
        case 2: f(std::integral_constant<int, 2>()); break;
        default: f(std::integral_constant<int, 3>()); break;
    }
}

int launch_geometry_forward(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                            const float* colors_precomp, const float* scales, const float* rotations,
                            const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped,
                            uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, hipStream_t stream) {
    const int blocks = (f.P + kBlock - 1) / kBlock;
    const int vec16 = (shs && aligned16(shs) && ((f.M * 3 * 4) % 16 == 0)) ? 1 : 0;
    by_degree(colors_precomp ? -1 : f.D, [&](auto deg) {
        hipLaunchKernelGGL(geometry_forward_kernel<decltype(deg)::value>, dim3(blocks), dim3(kBlock), 0, stream, f, means3D,
                           opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, reinterpret_cast<float4*>(splats),
                           radii, clamped, reinterpret_cast<uint2*>(rects), depth_keys, block_sums, vec16);
    });
    return check_hip(hipGetLastError(), "geometry_forward_kernel");
}

// the dynamic-LDS ceiling of every instantiation of geometry_hist_kernel (binning_tiles.hip device_setup, once per device)
hipError_t geometry_hist_set_max_lds(int bytes) {
    hipError_t rc = hipSuccess;
    for (int deg = -1; deg <= 3; ++deg)
        by_degree(deg, [&](auto d) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(geometry_hist_kernel<decltype(d)::value>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) rc = e;
        });
    return rc;
}

int launch_geometry_hist(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                         const float* colors_precomp, const float* scales, const float* rotations,
                         const float* cov3D_precomp, float* splats, int32_t* radii, uint8_t* clamped, uint32_t* rects,
                         uint32_t* depth_keys, uint32_t* block_sums, int nblocks, uint32_t* table, uint32_t* class_counts,
                         uint32_t* len_hist, hipStream_t stream) {
    const int vec16 = (shs && aligned16(shs) && ((f.M * 3 * 4) % 16 == 0)) ? 1 : 0;
    const int n_tiles = f.gx * f.gy;
    const int nb256 = (f.P + kBlock - 1) / kBlock;
    const int max_blocks = nb256 / nblocks + 2;                // 256-Gaussian blocks of one workgroup's slice, at most
    const size_t lds = geometry_hist_lds_bytes(n_tiles, max_blocks);
    by_degree(colors_precomp ? -1 : f.D, [&](auto deg) {
        hipLaunchKernelGGL(geometry_hist_kernel<decltype(deg)::value>, dim3(nblocks), dim3(kBinThreads), lds, stream, f, means3D,
                           opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, reinterpret_cast<float4*>(splats),
                           radii, clamped, reinterpret_cast<uint2*>(rects), depth_keys, block_sums, vec16, table, class_counts,
                           len_hist, max_blocks);
    });
    return check_hip(hipGetLastError(), "geometry_hist_kernel");
}

int launch_geometry_backward(const FrameDev& f, const float* means3D, const float* opacities, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations,
                             const float* cov3D_precomp, const int32_t* radii, const uint8_t* clamped,
                             const float* dsplats, float* dmeans3D, float* dmeans2D, float* dopac, float* dshs,
                             float* dcolors, float* dscales, float* drots, float* dcov3D, int flags,
                             hipStream_t stream) {
    const int vec16 = (shs && aligned16(shs) && ((f.M * 3 * 4) % 16 == 0)) ? 1 : 0;
    const bool staged = vec16 && dshs && aligned16(dshs) && f.M == 16;
    const int block = staged ? kWave : kBlock;
    const int recs = staged ? 60 : kBlock;                     // Gaussians per workgroup (geometry_backward_kernel: RECS)
    const int blocks = (f.P + recs - 1) / recs;
    auto kernel = !staged ? geometry_backward_kernel<false, kBlock>
                          : (f.D >= 3 ? geometry_backward_kernel<true, kWave, true> : geometry_backward_kernel<true, kWave, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(block), 0, stream, f, means3D, opacities, shs, colors_precomp, scales, rotations,
                       cov3D_precomp, radii, clamped, reinterpret_cast<const float4*>(dsplats), dmeans3D, dmeans2D, dopac,
                       dshs, dcolors, dscales, drots, dcov3D, vec16, flags);
    return check_hip(hipGetLastError(), "geometry_backward_kernel");
}

// ---- model path ------------------------------------------------------------------------------------------------------------
int launch_geometry_forward_model(const FrameDev& f, const ScgModel& m, float* splats, int32_t* radii, uint8_t* clamped,
                                  uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, hipStream_t stream) {
    const int blocks = (f.P + kBlock - 1) / kBlock;
    by_degree(f.D, [&](auto deg) {
        hipLaunchKernelGGL(geometry_forward_model_kernel<decltype(deg)::value>, dim3(blocks), dim3(kBlock), 0, stream, f, m,
                           reinterpret_cast<float4*>(splats), radii, clamped, reinterpret_cast<uint2*>(rects), depth_keys,
                           block_sums);
    });
    return check_hip(hipGetLastError(), "geometry_forward_model_kernel");
}

hipError_t geometry_hist_model_set_max_lds(int bytes) {
    hipError_t rc = hipSuccess;
    for (int deg = 0; deg <= 3; ++deg)
        by_degree(deg, [&](auto d) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(geometry_hist_model_kernel<decltype(d)::value>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) rc = e;
        });
    return rc;
}

int launch_geometry_hist_model(const FrameDev& f, const ScgModel& m, float* splats, int32_t* radii, uint8_t* clamped,
                               uint32_t* rects, uint32_t* depth_keys, uint32_t* block_sums, int nblocks, uint32_t* table,
                               uint32_t* class_counts, uint32_t* len_hist, hipStream_t stream) {
    const int n_tiles = f.gx * f.gy;
    const int nb256 = (f.P + kBlock - 1) / kBlock;
    const int max_blocks = nb256 / nblocks + 2;
    const size_t lds = geometry_hist_lds_bytes(n_tiles, max_blocks);
    by_degree(f.D, [&](auto deg) {
        hipLaunchKernelGGL(geometry_hist_model_kernel<decltype(deg)::value>, dim3(nblocks), dim3(kBinThreads), lds, stream, f, m,
                           reinterpret_cast<float4*>(splats), radii, clamped, reinterpret_cast<uint2*>(rects), depth_keys,
                           block_sums, table, class_counts, len_hist, max_blocks);
    });
    return check_hip(hipGetLastError(), "geometry_hist_model_kernel");
}

int launch_geometry_backward_model(const FrameDev& f, const ScgModel& m, const ScgModelGrads& g, const int32_t* radii,
                                   const uint8_t* clamped, const float* dsplats, float* dmeans2D, int flags, hipStream_t stream) {
    const int blocks_ray = (m.ray.count + kModelRecs - 1) / kModelRecs;
    const int blocks_bg = (m.bg.count + kModelRecs - 1) / kModelRecs;
    if (blocks_ray + blocks_bg == 0) return 0;
    auto kernel = f.D >= 3 ? geometry_backward_model_kernel<true> : geometry_backward_model_kernel<false>;
    hipLaunchKernelGGL(kernel, dim3(blocks_ray + blocks_bg), dim3(kWave), 0, stream, f, m, g, radii, clamped,
                       reinterpret_cast<const float4*>(dsplats), dmeans2D, flags, blocks_ray);
    return check_hip(hipGetLastError(), "geometry_backward_model_kernel");
}

int launch_model_activate(const ScgModel& m, float* means3D, float* opacities, float* scales, float* rotations,
                          hipStream_t stream) {
    const int P = m.ray.count + m.bg.count;
    if (P == 0) return 0;
    hipLaunchKernelGGL(model_activate_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, m, P, means3D, opacities,
                       scales, rotations);
    return check_hip(hipGetLastError(), "model_activate_kernel");
}

}  // namespace scg
