"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's utils/geo_check.py:25-128 with cv2.remap replaced by
an explicit float64 bilinear sampler (INTER_LINEAR, BORDER_CONSTANT 0; cv2 itself is not installed here, and its
5-bit fixed-point weights are not reproduced) — "parity unpinned": the reference function is dead code without tests.
Follows the reference statement by statement so that a reader can diff the two.
"""
import numpy as np


def get_pairs(c2ws, num_select=10):                                             # :25-31
    dists = np.linalg.norm(c2ws[:, None, :3, 3] - c2ws[None, :, :3, 3], axis=-1)
    eyes = np.eye(dists.shape[0])
    dists[eyes > 0] = 1e3
    sorted_vids = np.argsort(dists, axis=1, kind="stable")
    return sorted_vids[:, :num_select]


def remap_linear(img, x, y):
    """cv2.remap(img, x, y, INTER_LINEAR) with the default BORDER_CONSTANT (0), exact weights."""
    H, W = img.shape
    out = np.zeros(x.shape, dtype=np.float64)
    for r in range(x.shape[0]):
        for c in range(x.shape[1]):
            xf, yf = float(x[r, c]), float(y[r, c])
            if not (np.isfinite(xf) and np.isfinite(yf)):
                continue
            x0, y0 = int(np.floor(xf)), int(np.floor(yf))
            fx, fy = xf - x0, yf - y0
            acc = 0.0
            for dy, wy in ((0, 1 - fy), (1, fy)):
                for dx, wx in ((0, 1 - fx), (1, fx)):
                    xx, yy = x0 + dx, y0 + dy
                    if 0 <= xx < W and 0 <= yy < H:
                        acc += wy * wx * float(img[yy, xx])
            out[r, c] = acc
    return out.astype(np.float32)


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):   # :91-128
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(intrinsics_src, xyz_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([height, width]).astype(np.float32)
    y_src = xy_src[1].reshape([height, width]).astype(np.float32)
    sampled_depth_src = remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(intrinsics_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled_depth_src.reshape([-1]))
    xyz_reprojected = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_reprojected = xyz_reprojected[2].reshape([height, width]).astype(np.float32)
    K_xyz_reprojected = np.matmul(intrinsics_ref, xyz_reprojected)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_reprojected = K_xyz_reprojected[:2] / K_xyz_reprojected[2:3]
    x_reprojected = xy_reprojected[0].reshape([height, width]).astype(np.float32)
    y_reprojected = xy_reprojected[1].reshape([height, width]).astype(np.float32)
    return depth_reprojected, x_reprojected, y_reprojected, x_src, y_src


def geocheck(intrs, c2ws, depths, dist_thresh=1.0, depth_thresh=0.01, view_thresh=5, num_src=15):               # :33-88
    num_cams = intrs.shape[0]
    pairs = get_pairs(c2ws, num_src)
    filter_masks, filter_depths = [], []
    for i in range(num_cams):
        geo_mask_sum = 0
        depth_est_sum = 0
        depth_ref = depths[i]
        width, height = depth_ref.shape[1], depth_ref.shape[0]
        x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
        for j in pairs[i]:
            depth_reprojected, x2d, y2d, _, _ = reproject_with_depth(depth_ref, intrs[i], c2ws[i], depths[j], intrs[j], c2ws[j])
            with np.errstate(divide="ignore", invalid="ignore"):
                dist = np.sqrt((x2d - x_ref) ** 2 + (y2d - y_ref) ** 2)
                relative_depth_diff = np.abs(depth_reprojected - depth_ref) / depth_ref
            mask = np.logical_and(dist < dist_thresh, relative_depth_diff < depth_thresh)
            depth_reprojected[~mask] = 0
            geo_mask_sum += mask.astype(np.int32)
            depth_est_sum += depth_reprojected
        depth_est_averaged = (depth_est_sum + depth_ref) / (geo_mask_sum + 1)
        final_mask = geo_mask_sum > view_thresh
        filter_masks.append(final_mask)
        filter_depths.append(depth_est_averaged * final_mask.astype(np.float32))
    return np.stack(filter_depths, axis=0), np.stack(filter_masks, axis=0).astype(np.float32)
