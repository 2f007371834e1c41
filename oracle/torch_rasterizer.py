"""TEST INFRASTRUCTURE ONLY — pure-PyTorch fp32 CPU oracle of the rasterizer hot path.

PARITY UNPINNED against the reference's CUDA rasterizer (absent from
/root/reference; see oracle/__init__.py).  The algorithm follows SURVEY.md
Appendix A, constrained by the reference's call sites:

* I/O contract ............ gaussian_renderer/__init__.py:28-118
* matrix conventions ...... scene/cameras.py:54-63, utils/graphics_utils.py:51-71
* SH -> RGB ............... utils/sh_utils.py:26-112, gaussian_renderer/__init__.py:79-83
* cov3D ................... utils/general_utils.py:70-116 (no in-kernel quaternion normalisation;
                            python normalises at scene/gaussian_model.py:118)
* depth = camera z, un-normalised, differentiable ... scene/gaussian_model.py:246-263
* alpha (1,H,W) in [0,1], differentiable ............ train.py:168
* means2D gradient slot (NDC units) ................. scene/gaussian_model.py:932-934

Forward is an explicit restatement (project, bin, stable sort, per-tile front-to-back
blend).  Backward is torch.autograd through that forward: an independent derivation of
every gradient the HIP kernels compute analytically.

Every arithmetic step that feeds an INTEGER result (radius, tile rectangle, sort key)
is written as a chain of single fp32 operations in a fixed order, so that the HIP
kernel (compiled with -ffp-contract=off for that stage) reproduces it bit for bit.

Design decisions held fixed between this oracle and the HIP kernels (each one is
[UPSTREAM-RECALL] in SURVEY.md, i.e. unverifiable here):
  D1  alpha = min(0.99, opacity*G) is straight-through in backward (the clamp has
      unit gradient).
  D2  the 1.3*tanfov clamp of t.x/t.z, t.y/t.z: a clamped component is a constant in
      backward (no gradient to t.x / t.y, and none to t.z through the clamp product).
  D3  cull is view-space z <= 0.2 only (prefiltered=False at gaussian_renderer/__init__.py:49).
  D4  tile rectangle from float->int truncation, clamped to the grid; saturating in float
      before the cast so out-of-range values are defined.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

TILE = 16
ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.99
T_EPS = 1e-4
NEAR_Z = 0.2
LOWPASS = 0.3

# utils/sh_utils.py:26-43 (values, as fp32 constants in the kernel)
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class Settings(NamedTuple):
    """Same 12 fields, same order, as the reference's GaussianRasterizationSettings
    (gaussian_renderer/__init__.py:38-51)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def host_scalars(W: int, H: int, tanfovx: float, tanfovy: float):
    """fp32 host-side scalars, computed exactly as the C-ABI host code computes them."""
    f = np.float32
    tfx, tfy = f(tanfovx), f(tanfovy)
    focal_x = f(W) / (f(2.0) * tfx)
    focal_y = f(H) / (f(2.0) * tfy)
    limx = f(1.3) * tfx
    limy = f(1.3) * tfy
    return float(focal_x), float(focal_y), float(limx), float(limy)


def eval_sh_rgb(deg: int, sh: torch.Tensor, dirs: torch.Tensor):
    """sh (P,M,3) coefficient-major / channel-minor (scene/gaussian_model.py:142);
    dirs (P,3) unit.  Polynomials of utils/sh_utils.py:57-103 (deg <= 3)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def cov3d_from_scale_rot(scales, rotations, mod: float):
    """(P,6) [xx,xy,xz,yy,yz,zz] of (R S)(R S)^T; quaternion (r,x,y,z), NOT normalised here.
    Mirrors utils/general_utils.py:84-116 + strip_symmetric :70-82."""
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R00 = 1.0 - 2.0 * (y * y + z * z)
    R01 = 2.0 * (x * y - r * z)
    R02 = 2.0 * (x * z + r * y)
    R10 = 2.0 * (x * y + r * z)
    R11 = 1.0 - 2.0 * (x * x + z * z)
    R12 = 2.0 * (y * z - r * x)
    R20 = 2.0 * (x * z - r * y)
    R21 = 2.0 * (y * z + r * x)
    R22 = 1.0 - 2.0 * (x * x + y * y)
    s0, s1, s2 = mod * scales[:, 0], mod * scales[:, 1], mod * scales[:, 2]
    L00, L01, L02 = R00 * s0, R01 * s1, R02 * s2
    L10, L11, L12 = R10 * s0, R11 * s1, R12 * s2
    L20, L21, L22 = R20 * s0, R21 * s1, R22 * s2
    xx = L00 * L00 + L01 * L01 + L02 * L02
    xy = L00 * L10 + L01 * L11 + L02 * L12
    xz = L00 * L20 + L01 * L21 + L02 * L22
    yy = L10 * L10 + L11 * L11 + L12 * L12
    yz = L10 * L20 + L11 * L21 + L12 * L22
    zz = L20 * L20 + L21 * L21 + L22 * L22
    return torch.stack([xx, xy, xz, yy, yz, zz], dim=1)


def _sqrt_ieee(t: torch.Tensor) -> torch.Tensor:
    """Correctly rounded fp32 square root (numpy: the hardware instruction), for values that decide integers."""
    return torch.from_numpy(np.sqrt(t.detach().numpy().astype(np.float32, copy=False)))


def _straight_through(value_fwd: torch.Tensor, value_bwd: torch.Tensor):
    """forward value_fwd, gradient of value_bwd."""
    return value_bwd + (value_fwd - value_bwd).detach()


def preprocess(means3D, means2D, opacities, settings: Settings, shs=None, colors_precomp=None,
               scales=None, rotations=None, cov3D_precomp=None):
    """Per-Gaussian stage.  Returns a dict of per-Gaussian tensors;
    float outputs are differentiable w.r.t. the inputs."""
    H, W = int(settings.image_height), int(settings.image_width)
    P = means3D.shape[0]
    focal_x, focal_y, limx, limy = host_scalars(W, H, settings.tanfovx, settings.tanfovy)
    V = settings.viewmatrix.reshape(16).to(torch.float32)
    PM = settings.projmatrix.reshape(16).to(torch.float32)
    campos = settings.campos.reshape(3).to(torch.float32)
    gx_tiles = (W + TILE - 1) // TILE
    gy_tiles = (H + TILE - 1) // TILE

    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    # transformPoint4x3 on the flat (already transposed) view matrix: [x y z 1] . V
    tx = V[0] * x + V[4] * y + V[8] * z + V[12]
    ty = V[1] * x + V[5] * y + V[9] * z + V[13]
    tz = V[2] * x + V[6] * y + V[10] * z + V[14]
    in_front = tz > NEAR_Z

    hx = PM[0] * x + PM[4] * y + PM[8] * z + PM[12]
    hy = PM[1] * x + PM[5] * y + PM[9] * z + PM[13]
    hw = PM[3] * x + PM[7] * y + PM[11] * z + PM[15]
    m_w = 1.0 / (hw + 1e-7)
    ndc_x = hx * m_w + means2D[:, 0]          # means2D is the all-zero gradient slot
    ndc_y = hy * m_w + means2D[:, 1]

    if cov3D_precomp is not None:
        cov3D = cov3D_precomp
    else:
        cov3D = cov3d_from_scale_rot(scales, rotations, float(settings.scale_modifier))
    c_xx, c_xy, c_xz, c_yy, c_yz, c_zz = cov3D.unbind(1)

    # EWA projection with the 1.3*tanfov clamp (decision D2: clamped component is a constant)
    txtz = tx / tz
    tytz = ty / tz
    cl_x = (txtz < -limx) | (txtz > limx)
    cl_y = (tytz < -limy) | (tytz > limy)
    # forward value: clamp(tx/tz)*tz with the exact fp32 rounding the kernel performs.  Gradient:
    # clamped -> constant (D2); un-clamped -> clamp() is the identity and d/dtz of (tx/tz)*tz
    # cancels analytically, so the gradient is routed through tx / ty directly.
    t_x_val = (torch.clamp(txtz, -limx, limx) * tz).detach()
    t_y_val = (torch.clamp(tytz, -limy, limy) * tz).detach()
    t_x = torch.where(cl_x, t_x_val, _straight_through(t_x_val, tx))
    t_y = torch.where(cl_y, t_y_val, _straight_through(t_y_val, ty))

    tz2 = tz * tz
    # NB: `python_float / tensor` is evaluated by torch as reciprocal(tensor) * float (two roundings);
    # divide tensor by tensor so that J00 = focal_x / tz is ONE correctly-rounded fp32 division.
    J00 = torch.full_like(tz, focal_x) / tz
    J02 = -(focal_x * t_x) / tz2
    J11 = torch.full_like(tz, focal_y) / tz
    J12 = -(focal_y * t_y) / tz2
    # Wm[i][j] = w2c[i][j] = V[j*4+i];  Tm = J . Wm  (2x3)
    T00 = J00 * V[0] + J02 * V[2]
    T01 = J00 * V[4] + J02 * V[6]
    T02 = J00 * V[8] + J02 * V[10]
    T10 = J11 * V[1] + J12 * V[2]
    T11 = J11 * V[5] + J12 * V[6]
    T12 = J11 * V[9] + J12 * V[10]
    # u = Sigma . T0^T, v = Sigma . T1^T
    u0 = c_xx * T00 + c_xy * T01 + c_xz * T02
    u1 = c_xy * T00 + c_yy * T01 + c_yz * T02
    u2 = c_xz * T00 + c_yz * T01 + c_zz * T02
    v0 = c_xx * T10 + c_xy * T11 + c_xz * T12
    v1 = c_xy * T10 + c_yy * T11 + c_yz * T12
    v2 = c_xz * T10 + c_yz * T11 + c_zz * T12
    A = T00 * u0 + T01 * u1 + T02 * u2 + LOWPASS
    B = T00 * v0 + T01 * v1 + T02 * v2
    C = T10 * v0 + T11 * v1 + T12 * v2 + LOWPASS

    det = A * C - B * B
    det_ok = det != 0.0
    det_inv = 1.0 / torch.where(det_ok, det, torch.ones_like(det))
    con_a = C * det_inv
    con_b = -B * det_inv
    con_c = A * det_inv

    # The radius decides integers (tile rectangle, visibility): its two square roots must be the correctly rounded fp32
    # ones the kernel computes.  torch.sqrt is NOT that on every host: on the GPU box's EPYC 9575F it returns
    # sqrt(0x42571c73) = 0x40eaaaac where IEEE (numpy, the build container's Xeon, the MI355X) gives 0x40eaaaab, which moved
    # ceil(3 sqrt(lambda)) of one S3 Gaussian from 22 to 23 (a one-off script of round 3, tools/debug/s3_chain.py in the history).  numpy's sqrt is the hardware
    # instruction; nothing differentiable depends on the radius.
    with torch.no_grad():
        mid = 0.5 * (A + C)
        lam1 = mid + _sqrt_ieee(torch.clamp_min(mid * mid - det, 0.1))
        radius_f = torch.ceil(3.0 * _sqrt_ieee(lam1))

    px = ((ndc_x + 1.0) * W - 1.0) * 0.5
    py = ((ndc_y + 1.0) * H - 1.0) * 0.5

    with torch.no_grad():
        def _tile(vf, grid):
            return torch.clamp(torch.trunc(vf * 0.0625), 0.0, float(grid)).to(torch.int32)
        rmin_x = _tile(px - radius_f, gx_tiles)
        rmax_x = _tile(px + radius_f + 15.0, gx_tiles)
        rmin_y = _tile(py - radius_f, gy_tiles)
        rmax_y = _tile(py + radius_f + 15.0, gy_tiles)
        tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
        visible = in_front & det_ok & (tiles > 0)
        radii = torch.where(visible, torch.clamp(radius_f, 0.0, 2.0e9).to(torch.int32),
                            torch.zeros(P, dtype=torch.int32))
        tiles_touched = torch.where(visible, tiles, torch.zeros_like(tiles))

    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        d = means3D - campos[None, :]
        dlen = torch.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        dirs = d / dlen[:, None]
        raw = eval_sh_rgb(int(settings.sh_degree), shs, dirs) + 0.5
        clamped = (raw < 0.0).detach()
        rgb = torch.clamp_min(raw, 0.0)

    return dict(xy=torch.stack([px, py], 1), depth=tz, conic=torch.stack([con_a, con_b, con_c], 1),
                opacity=opacities.reshape(-1), rgb=rgb, clamped=clamped, cov3D=cov3D,
                radii=radii, tiles_touched=tiles_touched, visible=visible,
                rect=torch.stack([rmin_x, rmin_y, rmax_x, rmax_y], 1),
                grid=(gx_tiles, gy_tiles))


def bin_and_sort(pre, W: int, H: int):
    """Tile binning: inclusive scan, duplicateWithKeys, stable sort of
    (tile<<32 | depth_bits, id), tile ranges.  Integer-exact numpy."""
    gx_tiles, gy_tiles = pre["grid"]
    tiles_touched = pre["tiles_touched"].numpy().astype(np.int64)
    offsets = np.cumsum(tiles_touched)
    R = int(offsets[-1]) if offsets.size else 0
    rect = pre["rect"].numpy()
    depth_bits = pre["depth"].detach().numpy().astype(np.float32).view(np.uint32).astype(np.uint64)
    vis = np.nonzero(tiles_touched > 0)[0]
    keys = np.empty(R, dtype=np.uint64)
    vals = np.empty(R, dtype=np.uint32)
    if R:
        # vectorised expansion in the order: Gaussian id, then y, then x
        cnt = tiles_touched[vis]
        owner = np.repeat(vis, cnt)
        start = np.repeat(offsets[vis] - cnt, cnt)
        local = np.arange(R, dtype=np.int64) - start
        w = (rect[owner, 2] - rect[owner, 0]).astype(np.int64)
        ty_ = rect[owner, 1].astype(np.int64) + local // w
        tx_ = rect[owner, 0].astype(np.int64) + local % w
        tile = (ty_ * gx_tiles + tx_).astype(np.uint64)
        keys[:] = (tile << np.uint64(32)) | depth_bits[owner]
        vals[:] = owner.astype(np.uint32)
    order = np.argsort(keys, kind="stable")
    keys_sorted = keys[order]
    point_list = vals[order]
    n_tiles = gx_tiles * gy_tiles
    ranges = np.zeros((n_tiles, 2), dtype=np.uint32)
    if R:
        tile_of = (keys_sorted >> np.uint64(32)).astype(np.int64)
        starts = np.searchsorted(tile_of, np.arange(n_tiles), side="left")
        ends = np.searchsorted(tile_of, np.arange(n_tiles), side="right")
        touched = ends > starts
        ranges[touched, 0] = starts[touched]
        ranges[touched, 1] = ends[touched]
    return dict(point_offsets=offsets.astype(np.uint32), num_rendered=R, keys_unsorted=keys,
                vals_unsorted=vals, keys_sorted=keys_sorted, point_list=point_list, ranges=ranges)


def blend(pre, binning, settings: Settings, tile_stride: int = 1, tiles=None):
    """16x16-tile front-to-back alpha blend of colour + depth + alpha.
    tile_stride > 1 blends only every tile_stride-th tile (others show the background): the bounded
    sample used by bench.py's cpu_baseline leg.  `tiles` (a collection of tile indices) names the blended
    tiles explicitly instead (full-size parity tests: a stride subset plus the tiles with the longest lists)."""
    if tiles is not None:
        tiles = frozenset(int(t) for t in tiles)
    H, W = int(settings.image_height), int(settings.image_width)
    gx_tiles, gy_tiles = pre["grid"]
    bg = settings.bg.reshape(3).to(torch.float32)
    xy, conic, opac, rgb, depth = pre["xy"], pre["conic"], pre["opacity"], pre["rgb"], pre["depth"]
    ranges = binning["ranges"]
    plist = torch.from_numpy(binning["point_list"].astype(np.int64))

    Hp, Wp = gy_tiles * TILE, gx_tiles * TILE
    # padded per-tile layout: (tiles_y, tiles_x, 256, ch); cropped at the end
    out_c = [[None] * gx_tiles for _ in range(gy_tiles)]
    out_d = [[None] * gx_tiles for _ in range(gy_tiles)]
    out_a = [[None] * gx_tiles for _ in range(gy_tiles)]
    final_T = torch.ones(gy_tiles, gx_tiles, TILE * TILE)
    n_contrib = torch.zeros(gy_tiles, gx_tiles, TILE * TILE, dtype=torch.int32)

    loc = torch.arange(TILE * TILE)
    lx = (loc % TILE).to(torch.float32)
    ly = (loc // TILE).to(torch.float32)
    zero_c = torch.zeros(TILE * TILE, 3)
    zero_1 = torch.zeros(TILE * TILE)
    for tyi in range(gy_tiles):
        for txi in range(gx_tiles):
            s, e = int(ranges[tyi * gx_tiles + txi, 0]), int(ranges[tyi * gx_tiles + txi, 1])
            t_idx = tyi * gx_tiles + txi
            if e <= s or ((t_idx not in tiles) if tiles is not None else (t_idx % tile_stride) != 0):
                out_c[tyi][txi] = zero_c + bg[None, :]
                out_d[tyi][txi] = zero_1
                out_a[tyi][txi] = zero_1
                continue
            ids = plist[s:e]
            pxf = lx + float(txi * TILE)
            pyf = ly + float(tyi * TILE)
            g_xy = xy[ids]
            g_con = conic[ids]
            dx = g_xy[None, :, 0] - pxf[:, None]
            dy = g_xy[None, :, 1] - pyf[:, None]
            power = -0.5 * (g_con[None, :, 0] * dx * dx + g_con[None, :, 2] * dy * dy) - g_con[None, :, 1] * dx * dy
            G = torch.exp(power)
            oG = opac[ids][None, :] * G
            alpha = _straight_through(torch.clamp_max(oG, ALPHA_MAX), oG)        # decision D1
            valid = (power <= 0.0) & (alpha >= ALPHA_MIN)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            T_after = torch.cumprod(1.0 - a_eff, dim=1)
            T_before = torch.cat([torch.ones(TILE * TILE, 1), T_after[:, :-1]], dim=1)
            live = T_after.detach() >= T_EPS                    # monotone in the list index
            contrib = valid & live
            wgt = torch.where(contrib, alpha * T_before, torch.zeros_like(alpha))
            n_live = live.sum(1)
            Tf = torch.where(n_live > 0, T_after.gather(1, (n_live - 1).clamp_min(0)[:, None])[:, 0],
                             torch.ones(TILE * TILE))
            idx = torch.arange(1, e - s + 1, dtype=torch.int32)[None, :]
            n_contrib[tyi, txi] = torch.where(contrib, idx, torch.zeros_like(idx)).max(1).values
            final_T[tyi, txi] = Tf.detach()
            out_c[tyi][txi] = wgt @ rgb[ids] + Tf[:, None] * bg[None, :]
            out_d[tyi][txi] = wgt @ depth[ids]
            out_a[tyi][txi] = wgt.sum(1)

    def _assemble(cells, ch):
        rows = []
        for tyi in range(gy_tiles):
            row = [c.reshape(TILE, TILE, ch) for c in cells[tyi]]
            rows.append(torch.cat(row, dim=1))
        img = torch.cat(rows, dim=0)           # (Hp, Wp, ch)
        return img[:H, :W].permute(2, 0, 1).contiguous()

    color = _assemble(out_c, 3)
    depth_img = _assemble([[c[:, None] for c in r] for r in out_d], 1)
    alpha_img = _assemble([[c[:, None] for c in r] for r in out_a], 1)
    fT = final_T.reshape(gy_tiles, gx_tiles, TILE, TILE).permute(0, 2, 1, 3).reshape(Hp, Wp)[:H, :W]
    nC = n_contrib.reshape(gy_tiles, gx_tiles, TILE, TILE).permute(0, 2, 1, 3).reshape(Hp, Wp)[:H, :W]
    return color, depth_img, alpha_img, fT.contiguous(), nC.contiguous()


def rasterize(means3D, means2D, opacities, settings: Settings, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None, return_aux: bool = False, tile_stride: int = 1):
    """Full forward; returns (color(3,H,W), radii(P,) int32, depth(1,H,W), alpha(1,H,W)) — the
    4-tuple unpacked at gaussian_renderer/__init__.py:100 — plus aux when asked."""
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    H, W = int(settings.image_height), int(settings.image_width)
    pre = preprocess(means3D, means2D, opacities, settings, shs, colors_precomp, scales, rotations, cov3D_precomp)
    binning = bin_and_sort(pre, W, H)
    color, depth, alpha, final_T, n_contrib = blend(pre, binning, settings, tile_stride)
    if return_aux:
        aux = dict(pre=pre, binning=binning, final_T=final_T, n_contrib=n_contrib)
        return color, pre["radii"], depth, alpha, aux
    return color, pre["radii"], depth, alpha
