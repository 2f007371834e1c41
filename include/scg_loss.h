/* scg_loss.h — C ABI of the fused image loss.
 *
 * The reference's per-iteration image loss (train.py:160-161):
 *     Ll1  = l1_loss(image, gt_image)                                   utils/loss_utils.py:40
 *     loss = (1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image, gt_image))    utils/loss_utils.py:56-94
 * where ssim() is the 11x11 Gaussian-window (sigma 1.5) SSIM, computed by the reference with five zero-padded
 * grouped conv2d calls per iteration (plus their backward).  Here the whole loss and its gradient w.r.t. `image`
 * are two kernels: separable 11-tap convolutions staged through LDS, SSIM map and its three derivative maps in the
 * forward, three more separable convolutions in the backward.
 *
 * Same conventions as scg_raster.h (caller-owned device buffers, stream-ordered, int status). */
#ifndef SCG_LOSS_H
#define SCG_LOSS_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared in the headers under include/ are exported. */
#ifndef SCG_API
#define SCG_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of the `dmaps` buffer: three (C,H,W) fp32 derivative maps saved by the forward for the backward */
SCG_API size_t scg_image_loss_dmaps_bytes(int32_t C, int32_t H, int32_t W);

/* bytes of the forward's scratch (per-workgroup partial sums, reduced in fixed order by a second tiny kernel:
 * no same-address atomics, bitwise reproducible) */
SCG_API size_t scg_image_loss_scratch_bytes(int32_t C, int32_t H, int32_t W);

/* Forward.  img, gt: (C,H,W) fp32.  sums: 2 floats written by this call:
 *   sums[0] = sum over all elements of |img - gt|,  sums[1] = sum over all elements of the SSIM map
 * (the caller divides by C*H*W: L1 = sums[0]/N, SSIM = sums[1]/N).  dmaps: see above (may be NULL when no
 * gradient is needed). */
SCG_API int scg_image_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* sums,
                           float* dmaps, void* scratch, size_t scratch_bytes, void* stream);

/* Backward.  d_img (C,H,W) = w[0] * sign(img - gt) + w[1] * d(sum of SSIM map)/d(img), with the two weights read
 * from DEVICE memory (`weights`, 2 floats) so that the upstream gradients — device scalars under autograd — never
 * force a host read.  For loss = (1-l)*L1 + l*(1-SSIM) with upstream gradient g:  w[0] = g*(1-l)/N,  w[1] = -g*l/N. */
SCG_API int scg_image_loss_backward(const float* img, const float* gt, const float* dmaps, int32_t C, int32_t H, int32_t W,
                            const float* weights, float* d_img, void* stream);

/* The training loss in one piece (round 5): like scg_image_loss_forward, and sums3[2] = (1 - lambda_dssim) * sums3[0] / N +
 * lambda_dssim * (1 - sums3[1] / N) — train.py:160-161, the reference's expression in its order of operations — written by the
 * same reduction kernel; scg_image_loss_backward_combined takes the upstream gradient of THAT scalar (`upstream`: one float in
 * DEVICE memory) and forms the two weights itself: d_img = upstream * ((1-l)/N * sign(img - gt) - l/N * d(sum SSIM)/d(img)).
 * A training iteration's image loss is then two library calls and no tensor arithmetic around them (the separate form costs
 * the caller's framework about a dozen one-element kernels per iteration). */
SCG_API int scg_image_loss_forward_combined(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                            float lambda_dssim, float* sums3, float* dmaps, void* scratch,
                                            size_t scratch_bytes, void* stream);
SCG_API int scg_image_loss_backward_combined(const float* img, const float* gt, const float* dmaps, int32_t C, int32_t H,
                                             int32_t W, const float* upstream, float lambda_dssim, float* d_img,
                                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCG_LOSS_H */
