#!/usr/bin/env python3
"""Generate tests/golden/ref_model.npz by RUNNING the reference's own Python — scene/gaussian_model.py,
gaussian_renderer/__init__.py, scene/cameras.py, utils/geo_check.py — imported read-only from /root/reference in the
build container.  Only input/output vectors are committed; no reference source travels.

How the reference modules are made importable here (no GPU, several third-party packages absent):
  * import-time-only dependencies (plyfile, simple_knn, cv2, imageio, skimage, pytorch3d, torchvision, lpips) are
    satisfied by empty stub modules from a meta-path finder: nothing of them is CALLED by the functions exercised,
    with one exception stated below;
  * `device="cuda"` factory calls and `.cuda()` are redirected to the CPU (the reference hard-codes them);
  * `diff_gaussian_rasterization` (the un-vendored CUDA extension, README.md:23) is provided by a module of that name
    backed by the CPU oracle (oracle/torch_rasterizer.py), so the reference's `render()` runs end to end: what the
    fixture pins is the reference's GLUE (input selection by the pipe switches, activations, concatenation of ray-bound
    and background Gaussians, the result dict, which tensor receives which gradient) around that rasterizer;
  * utils/geo_check.py calls cv2.remap at run time: it is provided by scipy.ndimage.map_coordinates(order=1,
    mode="grid-constant": taps outside the image are 0 and ARE interpolated with, like cv2 BORDER_CONSTANT) — a third-party bilinear sampler (cv2's 5-bit fixed-point weights are not
    reproduced); everything else in geocheck / reproject_with_depth / get_pairs is the reference's own code.

Pins:
  a3  render()                                   gaussian_renderer/__init__.py:20-118   (4 switch combinations + override_color)
  a2  activated getters + get_covariance         scene/gaussian_model.py:105-152
  f2  get_matchloss_from_renderdepth             scene/gaussian_model.py:241-282        (value and d/d depth)
  f4  construct_list_of_attributes(_bg)          scene/gaussian_model.py:531-565
  f4  geocheck / reproject_with_depth / get_pairs utils/geo_check.py:25-128
  a15 add_densification_stats                    scene/gaussian_model.py:932-934
  cameras  Camera.intr / w2c / matrices          scene/cameras.py:54-72

Run:  python tests/golden/make_golden_model.py      (needs /root/reference; CPU only)
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "ref_model.npz")

STUB_ROOTS = {"plyfile", "simple_knn", "cv2", "imageio", "skimage", "pytorch3d", "torchvision", "lpips", "lpipsPyTorch"}


class _Missing:
    def __init__(self, name):
        self._n = name

    def __call__(self, *a, **k):
        raise RuntimeError(f"stub {self._n} was called: this path must not be exercised by the fixture generator")

    def __getattr__(self, k):
        return _Missing(self._n + "." + k)


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Missing(self.__name__ + "." + k)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def _cpu_device_shim():
    for name in ("zeros", "ones", "tensor", "empty", "zeros_like", "ones_like", "rand", "randn", "full"):
        orig = getattr(torch, name)

        def wrapped(*a, __orig=orig, **kw):
            if "device" in kw and "cuda" in str(kw["device"]):
                kw["device"] = "cpu"
            return __orig(*a, **kw)
        setattr(torch, name, wrapped)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None


def _install_oracle_rasterizer():
    """`diff_gaussian_rasterization` backed by the CPU oracle."""
    sys.path.insert(0, ROOT)
    from oracle import torch_rasterizer as orc

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            return orc.rasterize(means3D, means2D, opacities, self.raster_settings, shs=shs,
                                 colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                                 cov3D_precomp=cov3D_precomp)

    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings = orc.Settings
    m.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = m


def _rot(ax, ay):
    cx, sx, cy, sy = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    return Ry @ Rx


def main():
    sys.meta_path.insert(0, _Finder())
    _cpu_device_shim()
    _install_oracle_rasterizer()
    sys.path.insert(0, REF)
    from scene.gaussian_model import GaussianModel
    from scene.cameras import Camera
    import gaussian_renderer
    import utils.geo_check as ref_geo
    from scgaussian_amd import synthetic as syn

    g = torch.Generator().manual_seed(77)
    out = {}

    # ---------------------------------------------------------------- cameras (the reference's own Camera class)
    W, H = 96, 64
    fovy = math.radians(48.0)
    fovx = 2 * math.atan(math.tan(fovy / 2) * W / H)
    cams = []
    for i, (ax, ay, T) in enumerate([(0.0, 0.0, (0.0, 0.0, 0.0)), (0.04, -0.10, (0.7, 0.05, 0.1)), (-0.05, 0.09, (-0.6, -0.1, 0.05))]):
        R = _rot(ax, ay)                           # camera->world rotation, stored transposed as the loaders do
        cam = Camera(colmap_id=i, R=R, T=np.array(T, dtype=np.float64), FoVx=fovx, FoVy=fovy, image=None,
                     gt_alpha_mask=None, image_name=f"view{i}", uid=i, dtumask=None, near_far=None, blendermask=None,
                     height_in=H, width_in=W)
        cams.append(cam)
    for k in ("world_view_transform", "full_proj_transform", "camera_center", "intr", "w2c"):
        out["cam_" + k] = np.stack([getattr(c, k).numpy() for c in cams])
    out["cam_R"] = np.stack([c.R for c in cams])
    out["cam_T"] = np.stack([c.T for c in cams])
    out["cam_fov"] = np.array([fovx, fovy])
    out["cam_wh"] = np.array([W, H])

    # ---------------------------------------------------------------- a ray-bound model with background Gaussians
    Pr, Pb = 900, 200
    gm = GaussianModel(3)
    c2w0 = torch.linalg.inv(cams[0].w2c)
    uv = torch.stack([torch.rand(Pr, generator=g) * W, torch.rand(Pr, generator=g) * H], 1)
    cam_rays = (torch.linalg.inv(cams[0].intr) @ torch.cat([uv, torch.ones(Pr, 1)], 1).t()).t()
    cam_rays = cam_rays / cam_rays.norm(dim=1, keepdim=True)
    raw = dict(
        rayo=c2w0[:3, 3][None].repeat(Pr, 1).contiguous(),
        rayd=(c2w0[:3, :3] @ cam_rays.t()).t().contiguous(),
        zval=(torch.rand(Pr, 1, generator=g) * 6 + 3),
        features_dc=torch.rand(Pr, 1, 3, generator=g) * 3 - 1.5,
        features_rest=torch.randn(Pr, 15, 3, generator=g) * 0.15,
        opacity=torch.randn(Pr, 1, generator=g) * 2,
        scaling=torch.randn(Pr, 3, generator=g) * 0.5 - 2.6,
        rotation=torch.randn(Pr, 4, generator=g),
        bg_xyz=torch.stack([torch.randn(Pb, generator=g) * 3, torch.randn(Pb, generator=g) * 2,
                            torch.rand(Pb, generator=g) * 8 + 4], 1),
        bg_features_dc=torch.rand(Pb, 1, 3, generator=g) * 3 - 1.5,
        bg_features_rest=torch.randn(Pb, 15, 3, generator=g) * 0.15,
        bg_opacity=torch.randn(Pb, 1, generator=g) * 2,
        bg_scaling=torch.randn(Pb, 3, generator=g) * 0.5 - 2.2,
        bg_rotation=torch.randn(Pb, 4, generator=g),
    )
    for k, v in raw.items():
        out["raw_" + k] = v.numpy().copy()

    def fresh_model(active_deg):
        m = GaussianModel(3)
        m.active_sh_degree = active_deg
        names = dict(rayo="_rayo", rayd="_rayd", zval="_zval", features_dc="_features_dc", features_rest="_features_rest",
                     opacity="_opacity", scaling="_scaling", rotation="_rotation")
        for k, v in raw.items():
            t = v.clone().requires_grad_(k not in ("rayo", "rayd"))
            setattr(m, names.get(k, k), t)
        return m

    # ---- a2: activated getters --------------------------------------------------------------------------------
    m = fresh_model(3)
    with torch.no_grad():
        for name in ("get_xyz", "get_features", "get_opacity", "get_scaling", "get_rotation"):
            out["getter_" + name] = getattr(m, name).numpy().copy()
    # get_covariance pairs the concatenated scaling with the ray-bound `_rotation` only (scene/gaussian_model.py:151-152):
    # defined when there are no background Gaussians
    m_nobg = fresh_model(3)
    for k in ("bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"):
        setattr(m_nobg, k, getattr(m_nobg, k)[:0].detach())
    with torch.no_grad():
        for mod in (1.0, 0.6):
            out[f"getter_cov_nobg_mod{mod}"] = m_nobg.get_covariance(mod).numpy().copy()

    # ---- f4: attribute name lists -------------------------------------------------------------------------------
    out["attr_names"] = np.array(m.construct_list_of_attributes())
    out["attr_names_bg"] = np.array(m.construct_list_of_attributes_bg())

    # ---- a3: render() through the reference's glue ------------------------------------------------------------------
    class Pipe:
        def __init__(self, sh, cov):
            self.convert_SHs_python, self.compute_cov3D_python, self.debug = sh, cov, False

    dc, dd, da = syn.make_upstream_grads(W, H, seed=5)
    out["up_dc"], out["up_dd"], out["up_da"] = dc.numpy(), dd.numpy(), da.numpy()
    bg = torch.tensor([0.15, 0.3, 0.05])
    out["render_bg"] = bg.numpy()
    grad_names = ("_zval", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "bg_xyz",
                  "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation")
    cases = {"plain": (False, False, 3, 1.0, 0, True), "shpy": (True, False, 2, 1.0, 1, True),
             "covpy": (False, True, 3, 0.8, 2, False), "bothpy": (True, True, 1, 1.0, 1, False)}
    for tag, (sh_py, cov_py, deg, mod, cam_i, with_bg) in cases.items():
        mm = fresh_model(deg)
        if not with_bg:                              # get_covariance is only defined without background Gaussians
            for k in ("bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"):
                setattr(mm, k, getattr(mm, k)[:0].detach().requires_grad_(True))
        pkg = gaussian_renderer.render(cams[cam_i], mm, Pipe(sh_py, cov_py), bg, scaling_modifier=mod)
        assert set(pkg) == {"render", "rendered_depth", "rendered_alpha", "viewspace_points", "visibility_filter", "radii"}
        loss = (pkg["render"] * dc).sum() + (pkg["rendered_depth"] * dd).sum() + (pkg["rendered_alpha"] * da).sum()
        loss.backward()
        out[f"render_{tag}_cfg"] = np.array([int(sh_py), int(cov_py), deg, cam_i, int(with_bg)])
        out[f"render_{tag}_mod"] = np.float32(mod)
        out[f"render_{tag}_image"] = pkg["render"].detach().numpy()
        out[f"render_{tag}_depth"] = pkg["rendered_depth"].detach().numpy()
        out[f"render_{tag}_alpha"] = pkg["rendered_alpha"].detach().numpy()
        out[f"render_{tag}_radii"] = pkg["radii"].numpy()
        out[f"render_{tag}_visibility"] = pkg["visibility_filter"].numpy()
        out[f"render_{tag}_grad_viewspace"] = pkg["viewspace_points"].grad.numpy()
        for n in grad_names:
            t = getattr(mm, n)
            if t.grad is not None:
                out[f"render_{tag}_grad_{n.lstrip('_')}"] = t.grad.numpy()
        if tag == "plain":                           # a15: the densification statistics the reference derives
            P_all = pkg["radii"].shape[0]
            mm.xyz_gradient_accum = torch.zeros((P_all, 1))
            mm.denom = torch.zeros((P_all, 1))
            mm.add_densification_stats(pkg["viewspace_points"], pkg["visibility_filter"])
            out["densify_accum"] = mm.xyz_gradient_accum.numpy()
            out["densify_denom"] = mm.denom.numpy()
    # override_color
    mm = fresh_model(3)
    oc = torch.rand(Pr + Pb, 3, generator=g)
    out["render_override_colors"] = oc.numpy()
    pkg = gaussian_renderer.render(cams[0], mm, Pipe(False, False), bg, override_color=oc)
    out["render_override_image"] = pkg["render"].detach().numpy()
    # save_color_pcd (gaussian_renderer/__init__.py:20, 89-96 -> scene/dataset_readers.py:127-142 storePly): the
    # reference's own storePly runs; the two plyfile classes it hands its structured array to are replaced by
    # capturing objects, so the fixture pins the ARRAY it builds (x y z nx ny nz float32, red green blue uchar)
    import scene.dataset_readers as ref_readers
    captured = {}

    class _CapElement:
        @staticmethod
        def describe(elements, name):
            return (name, elements)

    class _CapData:
        def __init__(self, elements):
            self.elements = elements

        def write(self, path):
            captured[os.path.basename(path)] = self.elements[0][1]

    ref_readers.PlyElement, ref_readers.PlyData = _CapElement, _CapData
    mm = fresh_model(2)
    with torch.no_grad():
        # colours kept inside [0, 1]: above 1 the reference's uchar conversion is numpy-version dependent (1.x wraps,
        # 2.x raises OverflowError in `elements[:] = list(map(tuple, attributes))`)
        for k in ("_features_dc", "_features_rest", "bg_features_dc", "bg_features_rest"):
            getattr(mm, k).mul_(0.25)
        gaussian_renderer.render(cams[2], mm, Pipe(False, False), bg, save_color_pcd=True, color_pcd_save_path="/tmp/scg_golden")
    el = captured["point_cloud_color.ply"]
    out["color_pcd_cfg"] = np.array([2, 2])          # active SH degree, camera index
    out["color_pcd_xyz_normals"] = np.stack([el[k] for k in ("x", "y", "z", "nx", "ny", "nz")], 1)
    out["color_pcd_rgb"] = np.stack([el[k] for k in ("red", "green", "blue")], 1)
    assert out["color_pcd_rgb"].dtype == np.uint8 and out["color_pcd_xyz_normals"].dtype == np.float32

    # ---- f2: match loss on a rendered depth ---------------------------------------------------------------------
    M = 300
    names = [c.image_name for c in cams]
    view_gs = {}
    for i, c in enumerate(cams):
        view_gs[names[i]] = {"width": W, "height": H, "intr": c.intr, "w2c": c.w2c, "match_infos": {}}
    for i in range(3):
        for j in range(3):
            if i == j:
                continue
            uvm = torch.stack([torch.rand(M, generator=g) * (W + 6) - 3, torch.rand(M, generator=g) * (H + 6) - 3], 1)
            cr = (torch.linalg.inv(cams[i].intr) @ torch.cat([uvm, torch.ones(M, 1)], 1).t()).t()
            cr = cr / cr.norm(dim=1, keepdim=True)
            c2w = torch.linalg.inv(cams[i].w2c)
            view_gs[names[i]]["match_infos"][names[j]] = {
                "uv": uvm.contiguous(), "rays_o": c2w[:3, 3][None].repeat(M, 1).contiguous(),
                "rays_d": (c2w[:3, :3] @ cr.t()).t().contiguous(), "cam_rays_d": cr.contiguous(),
                "blender_mask": (torch.rand(M, generator=g) > 0.2).float()}
    mm = fresh_model(3)
    mm.view_gs = view_gs
    for i in range(3):
        for j in range(3):
            if i != j:
                for k, v in view_gs[names[i]]["match_infos"][names[j]].items():
                    out[f"match_{i}{j}_{k}"] = v.numpy()
    for i in range(3):
        depth = (torch.rand(1, H, W, generator=g) * 5 + 2.5).requires_grad_(True)
        ml = mm.get_matchloss_from_renderdepth(cams[i], depth, None)
        ml.backward()
        out[f"match_depth{i}"] = depth.detach().numpy()
        out[f"match_loss{i}"] = np.float32(ml.item())
        out[f"match_grad{i}"] = depth.grad.numpy()

    # ---- f4: cross-view depth consistency (utils/geo_check.py), cv2.remap = scipy bilinear with a zero border ------
    from scipy.ndimage import map_coordinates
    ref_geo.cv2.INTER_LINEAR = 1

    def remap(src, map_x, map_y, interpolation=1):
        return map_coordinates(src.astype(np.float64), [map_y.astype(np.float64), map_x.astype(np.float64)], order=1,
                               mode="grid-constant", cval=0.0).astype(np.float32)
    ref_geo.cv2.remap = remap
    rng = np.random.default_rng(3)
    n, gh, gw, f = 8, 20, 28, 30.0
    K = np.array([[f, 0, gw / 2], [0, f, gh / 2], [0, 0, 1.0]])
    intrs = np.repeat(K[None], n, 0)
    exts = np.zeros((n, 4, 4))
    depths = np.zeros((n, gh, gw), dtype=np.float32)
    nrm, d0 = np.array([0.1, -0.05, 1.0]), 6.0
    for i in range(n):
        ang = 0.06 * (i - n / 2)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.25 * (i - n / 2) + 0.017 * i * i, 0.05 * i, 0.0])      # irregular spacing: no distance ties in get_pairs
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, t
        exts[i] = E
        ys, xs = np.mgrid[0:gh, 0:gw]
        rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(gh * gw)])
        z = (d0 + nrm @ (R.T @ t)) / (nrm @ (R.T @ rays))
        depths[i] = z.reshape(gh, gw)
    bad = rng.random(depths.shape) < 0.04
    depths[bad] *= rng.uniform(1.05, 1.6, size=int(bad.sum())).astype(np.float32)
    out["geo_intrs"], out["geo_exts"], out["geo_depths"] = intrs, exts, depths
    out["geo_pairs4"] = ref_geo.get_pairs(exts, 4)
    rp = ref_geo.reproject_with_depth(depths[2], intrs[2], exts[2], depths[5], intrs[5], exts[5])
    for k, v in zip(("depth", "x", "y", "xsrc", "ysrc"), rp):
        out["geo_reproject_" + k] = np.asarray(v)
    with np.errstate(all="ignore"):
        fd, fm = ref_geo.geocheck(intrs, exts, depths.copy(), dist_thresh=1.0, depth_thresh=0.01, view_thresh=3)
    out["geo_filtered_depths"], out["geo_masks"] = fd.astype(np.float32), fm.astype(np.float32)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
