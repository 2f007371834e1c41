"""TEST INFRASTRUCTURE ONLY — float64 numpy restatement of the cross-view depth consistency check of the reference
(utils/geo_check.py: nearest source views :25-31, the reference->source->reference round trip :91-128, the
consistency filter and depth averaging :33-88).  Written independently of the reference's code shape (row-vector
point arrays, one explicit bilinear sampler instead of cv2.remap — cv2 is not installable here and its 5-bit
fixed-point interpolation weights are NOT reproduced), so this pins the torch implementation in
scgaussian_amd/geo_check.py against a second derivation, not against the reference's bytes: "parity unpinned"
(the reference function is dead code without tests or golden data).
"""
import numpy as np


def get_pairs(cam_mats, num_select=10):
    """For every camera the indices of its `num_select` nearest other cameras (translation column distance; a
    camera is never its own neighbour: its self-distance is replaced by 1e3 before sorting) — :25-31."""
    centres = cam_mats[:, :3, 3]
    d = np.sqrt(((centres[:, None, :] - centres[None, :, :]) ** 2).sum(-1))
    np.fill_diagonal(d, 1e3)
    return np.argsort(d, axis=1, kind="stable")[:, :num_select]


def bilinear_zero_border(img, xs, ys):
    """Sample img (H, W) at real pixel coordinates (integer = pixel centre); taps outside the image contribute 0;
    non-finite coordinates sample 0.  The exact-weight counterpart of cv2.remap(INTER_LINEAR, BORDER_CONSTANT)."""
    H, W = img.shape
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    good = np.isfinite(xs) & np.isfinite(ys)
    xs = np.where(good, xs, -5.0)
    ys = np.where(good, ys, -5.0)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    wx1, wy1 = xs - x0, ys - y0
    acc = np.zeros(xs.shape, dtype=np.float64)
    for oy, wy in ((0, 1.0 - wy1), (1, wy1)):
        for ox, wx in ((0, 1.0 - wx1), (1, wx1)):
            xi, yi = x0 + ox, y0 + oy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            acc += np.where(ok, img[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)], 0.0) * wx * wy
    return acc.astype(np.float32)


def _pixel_grid(H, W):
    v, u = np.divmod(np.arange(H * W), W)
    return u.astype(np.float64), v.astype(np.float64)


def _project(K, pts):
    """pts (N,3) camera-space -> (u, v) with the perspective division the reference applies (no guard on z)."""
    h = pts @ K.T
    with np.errstate(divide="ignore", invalid="ignore"):
        return h[:, 0] / h[:, 2], h[:, 1] / h[:, 2]


def _lift(K, u, v, depth):
    """pixels + depth -> camera-space points (N,3): depth * K^-1 [u v 1]^T."""
    rays = np.stack([u, v, np.ones_like(u)], axis=1) @ np.linalg.inv(K).T
    return rays * depth[:, None]


def _move(E_to, E_from, pts):
    """camera `from` -> camera `to` with the 4x4 matrices used as the reference uses them: E_to · E_from^-1."""
    M = E_to @ np.linalg.inv(E_from)
    return pts @ M[:3, :3].T + M[:3, 3]


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """Round trip of every reference pixel through the source view (:91-128).  Returns, as (H, W) float32 maps: the
    depth of the returned point in the reference camera, its reference pixel coordinates, and where the pixel
    landed in the source view."""
    H, W = depth_ref.shape
    u, v = _pixel_grid(H, W)
    in_src = _move(E_src, E_ref, _lift(K_ref, u, v, depth_ref.reshape(-1).astype(np.float64)))
    us, vs = _project(K_src, in_src)
    us32 = us.reshape(H, W).astype(np.float32)
    vs32 = vs.reshape(H, W).astype(np.float32)
    seen = bilinear_zero_border(depth_src, us32, vs32).reshape(-1).astype(np.float64)
    back = _move(E_ref, E_src, _lift(K_src, us, vs, seen))
    ub, vb = _project(K_ref, back)
    as_map = lambda a: a.reshape(H, W).astype(np.float32)      # noqa: E731
    return as_map(back[:, 2]), as_map(ub), as_map(vb), us32, vs32


def geocheck(intrs, c2ws, depths, dist_thresh=1.0, depth_thresh=0.01, view_thresh=5, num_src=15):
    """A pixel of view i survives when, in MORE than `view_thresh` of its `num_src` nearest views, the round trip
    lands within `dist_thresh` pixels of where it started and within `depth_thresh` relative depth; its depth
    becomes the mean of its own depth and the consistent round-trip depths (:33-88)."""
    n, H, W = depths.shape
    neighbours = get_pairs(c2ws, num_src)
    gu, gv = np.meshgrid(np.arange(W), np.arange(H))
    kept_depth = np.zeros((n, H, W), dtype=np.float64)
    kept_mask = np.zeros((n, H, W), dtype=np.float32)
    for i in range(n):
        votes = np.zeros((H, W), dtype=np.int32)
        depth_sum = np.zeros((H, W), dtype=np.float64)
        for j in neighbours[i]:
            d_back, u_back, v_back, _, _ = reproject_with_depth(depths[i], intrs[i], c2ws[i], depths[j], intrs[j], c2ws[j])
            with np.errstate(divide="ignore", invalid="ignore"):
                moved = np.hypot(u_back - gu, v_back - gv)
                rel = np.abs(d_back - depths[i]) / depths[i]
            agree = (moved < dist_thresh) & (rel < depth_thresh)
            votes += agree
            depth_sum += np.where(agree, d_back, 0.0)
        keep = votes > view_thresh
        kept_mask[i] = keep
        kept_depth[i] = (depth_sum + depths[i]) / (votes + 1) * keep
    return kept_depth, kept_mask
