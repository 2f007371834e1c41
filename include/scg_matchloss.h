/* scg_matchloss.h — C ABI of the fused depth-consumer step.
 *
 * The reference's GaussianModel.get_matchloss_from_renderdepth (scene/gaussian_model.py:241-282), called every
 * training iteration on the rasterizer's depth output (train.py:164): for each matched view pair, bilinear-sample
 * the rendered depth at the <= 2000 match pixels of view 0, lift them along their rays, project into view 1 and
 * take the normalised L1 distance to the matched pixels, averaged over the matches that project inside the image
 * and are valid in both masks.  In the reference this is ~30 small torch kernels per pair and as many again in
 * backward; here it is ONE kernel per pair that also produces the gradient w.r.t. the depth image (the only
 * differentiable input), scattered to the four bilinear taps of every match.
 *
 * Same conventions as scg_raster.h (caller-owned device buffers, stream-ordered, int status). */
#ifndef SCG_MATCHLOSS_H
#define SCG_MATCHLOSS_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared in the headers under include/ are exported. */
#ifndef SCG_API
#define SCG_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* One view pair.
 *   depth (H,W) fp32                      rendered depth of view 0 (rasterizer output, squeezed)
 *   uv0 (M,2)                             match pixels in view 0            (:253 match_data["uv"])
 *   rays_o (M,3), rays_d (M,3)            world-space rays of those pixels  (:261)
 *   cam_rays_d (M,3)                      camera-space ray directions; only .z is used (:262)
 *   mask0, mask1 (M) or NULL              blender masks of both views; valid = mask0*mask1 > 0 (:249-251)
 *   intr1 (3,3), w2c1 (4,4) row-major     intrinsics / world-to-camera of view 1 (:265)
 *   uv1 (M,2)                             matched pixels in view 1          (:273)
 *   width, height                         normalisation and in-image test   (:244,:271,:277)
 * Outputs (both ACCUMULATED, the caller zeroes them once per view):
 *   loss (1)            += sum_i(l_i m_i) / (sum_i m_i + 1e-8)
 *   grad_depth (H,W)    += d(that term)/d(depth), or NULL when no gradient is needed */
SCG_API int scg_match_loss_pair(const float* depth, int32_t H, int32_t W, const float* uv0, const float* rays_o,
                        const float* rays_d, const float* cam_rays_d, const float* mask0, const float* mask1,
                        const float* intr1, const float* w2c1, const float* uv1, int32_t M, float width, float height,
                        float* loss, float* grad_depth, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCG_MATCHLOSS_H */
