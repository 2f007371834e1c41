// knn.hip — mean squared distance to the 3 nearest neighbours (include/scg_knn.h).
//
// Brute force on purpose: the call happens once per training run on <= ~12 k points in SCGaussian (2 000 matches x
// 6 ordered pairs, data_preprocess/get_match_info.py:376) and a few 100 k for COLMAP initialisations; the exact
// all-pairs search is ~10 VALU per pair and the MI355X sustains > 10^12 pairs/s, so no spatial index is needed.
// Queries: one per thread.  Candidates: streamed through LDS as float4 tiles, read with wave-uniform (broadcast)
// ds_read_b128.  Long candidate ranges are split across blockIdx.y; partial top-3 lists are merged by a second
// tiny kernel (keeps the GPU full when N is small).
#include "scg_common.h"
#include "../../include/scg_knn.h"

namespace scg {

constexpr int kKnnTile = 1024;          // candidates per LDS tile
constexpr int kKnnSplitMax = 64;        // max candidate-range splits (blockIdx.y)

__device__ __forceinline__ void insert3(float d, float& b0, float& b1, float& b2) {
    // keep b0 <= b1 <= b2 = the three smallest seen so far
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
        } else {
            b2 = d;
        }
    }
}

__global__ __launch_bounds__(kBlock) void knn3_partial_kernel(const float* __restrict__ pts, int n, int splits,
                                                              float* __restrict__ partial /* (splits, n, 3) */) {
    __shared__ float4 s_p[kKnnTile];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < n) { qx = pts[3 * (size_t)i]; qy = pts[3 * (size_t)i + 1]; qz = pts[3 * (size_t)i + 2]; }
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
    const int per = (n + splits - 1) / splits;
    const int j_begin = blockIdx.y * per;
    const int j_end = min(n, j_begin + per);
    for (int base = j_begin; base < j_end; base += kKnnTile) {
        __syncthreads();
        for (int k = threadIdx.x; k < kKnnTile; k += kBlock) {
            const int j = base + k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < j_end) v = make_float4(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2], 0.f);
            s_p[k] = v;
        }
        __syncthreads();
        const int cnt = min(kKnnTile, j_end - base);
        for (int k = 0; k < cnt; ++k) {
            const float4 c = s_p[k];
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (base + k != i) insert3(d, b0, b1, b2);
        }
    }
    if (i < n) {
        float* o = partial + ((size_t)blockIdx.y * n + i) * 3;
        o[0] = b0; o[1] = b1; o[2] = b2;
    }
}

__global__ __launch_bounds__(kBlock) void knn3_merge_kernel(const float* __restrict__ partial, int n, int splits,
                                                            float* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;
    for (int s = 0; s < splits; ++s) {
        const float* p = partial + ((size_t)s * n + i) * 3;
        insert3(p[0], b0, b1, b2); insert3(p[1], b0, b1, b2); insert3(p[2], b0, b1, b2);
    }
    // mean over the neighbours that exist (n - 1 < 3 for tiny inputs)
    const int have = min(3, n - 1);
    float sum = 0.f;
    if (have > 0) sum += b0;
    if (have > 1) sum += b1;
    if (have > 2) sum += b2;
    out[i] = have > 0 ? sum / (float)have : 0.f;
}

}  // namespace scg

using namespace scg;

extern "C" {

size_t scg_knn3_scratch_bytes(int64_t n) {
    if (n <= 0) return 256;
    int splits = (int)((256 * 8 * (int64_t)kBlock + n - 1) / n);      // aim for >= 2048 workgroups
    if (splits < 1) splits = 1;
    if (splits > kKnnSplitMax) splits = kKnnSplitMax;
    return (size_t)splits * (size_t)n * 3 * sizeof(float) + 256;
}

int scg_knn3_mean_dist2_ws(const float* points, int64_t n, float* mean_dist2, void* scratch, size_t scratch_bytes,
                           void* stream) {
    if (n < 0 || n > 0x7FFFFFFFll) return fail(SCG_E_RANGE, "n out of range");
    if (n == 0) return 0;
    if (!points || !mean_dist2 || !scratch) return fail(SCG_E_NULL, "knn pointer is NULL");
    if (scratch_bytes < scg_knn3_scratch_bytes(n)) return fail(SCG_E_SCRATCH, "knn scratch too small");
    int splits = (int)((256 * 8 * (int64_t)kBlock + n - 1) / n);
    if (splits < 1) splits = 1;
    if (splits > kKnnSplitMax) splits = kKnnSplitMax;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    float* partial = reinterpret_cast<float*>(scratch);
    hipLaunchKernelGGL(knn3_partial_kernel, dim3(blocks, splits), dim3(kBlock), 0, s, points, (int)n, splits, partial);
    hipLaunchKernelGGL(knn3_merge_kernel, dim3(blocks), dim3(kBlock), 0, s, partial, (int)n, splits, mean_dist2);
    return check_hip(hipGetLastError(), "knn3");
}

}  // extern "C"
