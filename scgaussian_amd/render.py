"""`render()` twin of the reference's render glue (gaussian_renderer/__init__.py:20-118), bound to the
MI355X rasterizer.  The reference's own `gaussian_renderer.render` runs unchanged on top of the drop-in
`diff_gaussian_rasterization` package; this twin exists so that the glue's behaviour (input selection by
`pipe.convert_SHs_python` / `pipe.compute_cov3D_python` / `override_color`, the 6-key result dict) can be
exercised without the reference tree (which is not present on the GPU box).

`pc` is any object with the reference model's read interface (scene/gaussian_model.py:105-155):
get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_covariance(scaling_modifier),
active_sh_degree, max_sh_degree.  `viewpoint_camera` needs FoVx, FoVy, image_height, image_width,
world_view_transform, full_proj_transform, camera_center (scene/cameras.py:19-72).
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

from . import model_path
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

# utils/sh_utils.py:26-43
_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


# render() hands a model that carries the reference's raw tensors to the rasterizer's model path (activations and
# concatenations inside the geometry kernels); False: always through the getters, like the reference (module switch for A/B runs)
MODEL_FAST_PATH = True


class PipelineParams(NamedTuple):
    """arguments/__init__.py:64-69 defaults."""
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False


def model_fast_path_available(pc, pipe=None, override_color=None) -> bool:
    """True when render() would hand `pc` to the rasterizer's model path (model_path.rasterize_model): it carries the reference
    model's raw tensors in a form the kernels take as they are, and the call uses the reference's default pipe (SH colours and
    scale + rotation covariance evaluated by the rasterizer: gaussian_renderer/__init__.py:64-68, 78-85)."""
    if not MODEL_FAST_PATH or override_color is not None:
        return False
    if pipe is not None and (pipe.convert_SHs_python or pipe.compute_cov3D_python):
        return False
    if int(getattr(pc, "max_sh_degree", 3)) != 3:
        return False
    return model_path.model_for(pc) is not None


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis values (P, (deg+1)^2) at unit directions (P,3), in the coefficient order and sign
    convention of utils/sh_utils.py:57-103 (degrees 0-3)."""
    assert 0 <= deg <= 3
    x, y, z = dirs.unbind(-1)
    cols = [torch.full_like(x, _C0)]
    if deg >= 1:
        cols += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        cols += [_C2[0] * x * y, _C2[1] * y * z, _C2[2] * (2.0 * zz - xx - yy), _C2[3] * x * z, _C2[4] * (xx - yy)]
        if deg >= 3:
            cols += [_C3[0] * y * (3.0 * xx - yy), _C3[1] * x * y * z, _C3[2] * y * (4.0 * zz - xx - yy),
                     _C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), _C3[4] * x * (4.0 * zz - xx - yy),
                     _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=-1)


def sh_to_rgb(deg: int, features: torch.Tensor, xyz: torch.Tensor, campos: torch.Tensor) -> torch.Tensor:
    """Colours the rasterizer would compute in-kernel, in torch: features (P,M,3) coefficient-major
    (scene/gaussian_model.py:142).  Used for the `convert_SHs_python` switch
    (gaussian_renderer/__init__.py:78-83): max(basis(dir) . sh + 0.5, 0)."""
    d = xyz - campos.reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    B = sh_basis(deg, d)                                            # (P,K)
    return torch.clamp_min(torch.einsum("pk,pkc->pc", B, features[:, :B.shape[1]]) + 0.5, 0.0)


def store_color_ply(path: str, xyz, rgb255) -> None:
    """Twin of scene/dataset_readers.py:127-142 `storePly`: x y z nx ny nz (float32, normals zero) + red green blue
    (uchar, the float colours cast the way numpy casts them there)."""
    import numpy as np
    from .ply_io import write_vertex_ply
    xyz = np.asarray(xyz, dtype=np.float32)
    cols = np.concatenate((xyz, np.zeros_like(xyz), np.asarray(rgb255, dtype=np.float32)), axis=1)
    write_vertex_ply(path, ["x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"], cols,
                     dtypes=["f4"] * 6 + ["u1"] * 3)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
           override_color: Optional[torch.Tensor] = None, save_color_pcd: bool = False,
           color_pcd_save_path: Optional[str] = None, reference_call_pattern: bool = False):
    """Render the scene; background tensor must be on the GPU.  Returns the reference's dict:
    render, rendered_depth, rendered_alpha, viewspace_points, visibility_filter, radii.
    `save_color_pcd` / `color_pcd_save_path` (gaussian_renderer/__init__.py:20, 89-96; passed by render.py:135): also
    write `<color_pcd_save_path>/point_cloud_color.ply` — every Gaussian with its view-dependent colour
    max(SH(dir) + 0.5, 0) * 255 as seen from this camera.
    `reference_call_pattern` (bench.py's `render_glue` leg): evaluate the model's getters exactly as often as the reference's
    render() does — `pc.get_xyz` THREE times (gaussian_renderer/__init__.py:28 twice, :55), each one `rayo + rayd * zval` and
    a concatenation (scene/gaussian_model.py:126-131) — so that the glue around the operator is timed at the reference's
    own cost; by default the position is computed once.

    MODEL PATH (round 6): a `pc` that carries the reference model's raw tensors (scene/gaussian_model.py:452-468) is rendered
    WITHOUT its getters when the call uses the default pipe — the geometry kernels read `_zval / _rayo / _rayd / _features_dc /
    _features_rest / _opacity / _scaling / _rotation / bg_*` where they lie and the backward writes the gradients of those very
    tensors (model_path.py).  Same dict, same values up to the rounding of the activations, same `.grad` slots
    (`viewspace_points.grad[:, :2]` for the densification statistics).  MODEL_FAST_PATH = False, another pipe, an override colour,
    a color-ply request or tensors the kernels cannot take as they are: through the getters, as the reference does."""
    if MODEL_FAST_PATH and override_color is None and not save_color_pcd and not reference_call_pattern \
            and not (pipe.convert_SHs_python or pipe.compute_cov3D_python) and int(getattr(pc, "max_sh_degree", 3)) == 3:
        raster_settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
            projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
            campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
        margs = model_path.model_for(pc, raster_settings)
        if margs is not None:
            # the screen-space gradient slot (gaussian_renderer/__init__.py:28): a leaf whose VALUE nobody reads — no fill launch
            screenspace_points = torch.empty((margs.P, 3), dtype=torch.float32, device=margs.device,
                                             requires_grad=torch.is_grad_enabled())
            rendered_image, radii, rendered_depth, rendered_alpha = model_path.rasterize_model(
                raster_settings, screenspace_points, _args=margs)
            return {"render": rendered_image, "rendered_depth": rendered_depth, "rendered_alpha": rendered_alpha,
                    "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
    if reference_call_pattern:
        _ = pc.get_xyz.dtype                          # (:28 reads the dtype off a second evaluation)
        _ = pc.get_xyz                                # (:28 zeros_like(pc.get_xyz, ...))
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            colors_precomp = sh_to_rgb(pc.active_sh_degree, pc.get_features, xyz, viewpoint_camera.camera_center)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    if save_color_pcd:
        import os
        pcd_color = sh_to_rgb(pc.active_sh_degree, pc.get_features, xyz, viewpoint_camera.camera_center)
        store_color_ply(os.path.join(color_pcd_save_path, "point_cloud_color.ply"), xyz.detach().cpu().numpy(),
                        pcd_color.detach().cpu().numpy() * 255)

    rendered_image, radii, rendered_depth, rendered_alpha = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=pc.get_opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    return {"render": rendered_image, "rendered_depth": rendered_depth, "rendered_alpha": rendered_alpha,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def render_views(viewpoint_cameras, pc, pipe, bg_color: torch.Tensor, scaling_modifier: float = 1.0):
    """K views of one model in ONE autograd node (BASELINE cfg5's batched multi-view step; no counterpart in the reference, whose
    train.py:143 renders one view per iteration): a list of render()'s dicts, one per camera.  `viewspace_points` of every dict is
    the camera's (P, 3) slice of one (K, P, 3) leaf — after backward its gradient is `dicts[0]["viewspace_points_all"].grad[k]`
    (the densification statistics are per view, scene/gaussian_model.py:932-934).  The views' parameter gradients are summed in the
    kernels (scg_backward_model `accumulate`) and land in one arena: one gradient exchange per K views
    (parallel.GradBucket.reduce_grads).  Needs the model path (a model that carries the reference's raw tensors, the default pipe)."""
    if pipe.convert_SHs_python or pipe.compute_cov3D_python or int(getattr(pc, "max_sh_degree", 3)) != 3:
        raise ValueError("render_views needs the default pipe (SH colours and scale + rotation covariance evaluated by the rasterizer)")
    settings = [GaussianRasterizationSettings(
        image_height=int(c.image_height), image_width=int(c.image_width), tanfovx=math.tan(c.FoVx * 0.5),
        tanfovy=math.tan(c.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=c.world_view_transform,
        projmatrix=c.full_proj_transform, sh_degree=pc.active_sh_degree, campos=c.camera_center, prefiltered=False,
        debug=pipe.debug) for c in viewpoint_cameras]
    margs = model_path.model_for(pc, settings[0]) if settings else None
    if margs is None or not all(model_path._accepts(margs.P, st) for st in settings):
        raise ValueError("render_views needs a model the model path can take as it is (model_path.model_for) on a ROCm GPU")
    K = len(settings)
    points = torch.empty((K, margs.P, 3), dtype=torch.float32, device=margs.device, requires_grad=torch.is_grad_enabled())
    outs = model_path.rasterize_model_views(settings, points, _args=margs)
    return [{"render": c, "rendered_depth": d, "rendered_alpha": a, "viewspace_points": points[k], "viewspace_points_all": points,
             "visibility_filter": r > 0, "radii": r} for k, (c, r, d, a) in enumerate(outs)]

