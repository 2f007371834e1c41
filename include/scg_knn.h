/* scg_knn.h — C ABI of the nearest-neighbour initialisation helper.
 *
 * Replaces the third-party extension the reference imports at scene/gaussian_model.py:20
 * (`from simple_knn._C import distCUDA2`, cloned at install time per README.md:24, absent from /root/reference)
 * and calls once at scene/gaussian_model.py:444 to initialise the Gaussian scales:
 *     dist2 = clamp_min(distCUDA2(points), 1e-7);  scales = log(sqrt(dist2))
 * Semantics [UPSTREAM-RECALL]: for every point the MEAN of the squared Euclidean distances to its 3 nearest
 * OTHER points (duplicates at distance 0 count).  With fewer than 4 points the missing neighbours are ignored
 * (mean over the available ones; 0 for a single point).
 *
 * Same conventions as scg_raster.h: extern "C", device pointers owned by the caller, work enqueued on `stream`,
 * 0 = ok / < 0 = SCG_E_* / > 0 = hipError_t, scg_last_error() for the message. */
#ifndef SCG_KNN_H
#define SCG_KNN_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared in the headers under include/ are exported. */
#ifndef SCG_API
#define SCG_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#include <stddef.h>

/* points (N,3) fp32 -> mean_dist2 (N) fp32.  Exact brute-force search, LDS-tiled: 256 queries per workgroup, the
 * candidate set streamed through LDS in tiles of 1024 points (~10 VALU per pair); the candidate range is split
 * over blockIdx.y when N is small so the chip stays full, partial top-3 lists are merged by a second kernel.
 * scratch: scg_knn3_scratch_bytes(N) bytes, caller-owned. */
SCG_API size_t scg_knn3_scratch_bytes(int64_t n);
SCG_API int scg_knn3_mean_dist2_ws(const float* points, int64_t n, float* mean_dist2, void* scratch, size_t scratch_bytes,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCG_KNN_H */
