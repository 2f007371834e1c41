"""The oracle's importable sub-pieces against vectors produced by the reference's own Python
(tests/golden/make_golden.py): SH->RGB, cov3D, camera matrices.  CPU only."""
import math

import numpy as np
import torch

from oracle import torch_rasterizer as orc
from scgaussian_amd import synthetic as syn


def test_sh_to_rgb_matches_reference(ref_pieces):
    shs = torch.from_numpy(ref_pieces["sh_shs"])
    xyz = torch.from_numpy(ref_pieces["sh_xyz"])
    campos = torch.from_numpy(ref_pieces["sh_campos"])
    d = xyz - campos[None]
    dirs = d / torch.sqrt((d * d).sum(1, keepdim=True))
    for deg in range(4):
        raw = orc.eval_sh_rgb(deg, shs, dirs) + 0.5
        rgb = torch.clamp_min(raw, 0.0)
        np.testing.assert_allclose(rgb.numpy(), ref_pieces[f"sh_rgb_deg{deg}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(raw.numpy(), ref_pieces[f"sh_raw_deg{deg}"], rtol=2e-5, atol=2e-6)


def test_cov3d_matches_reference(ref_pieces):
    scales = torch.from_numpy(ref_pieces["cov_scales"])
    rot = torch.from_numpy(ref_pieces["cov_rot"])
    for mod in (1.0, 0.37):
        cov = orc.cov3d_from_scale_rot(scales, rot, mod)
        ref = ref_pieces[f"cov_sym_mod{mod}"]
        # off-diagonals cancel: tolerance relative to each matrix's largest entry
        scale = np.abs(ref).max(axis=1, keepdims=True)
        assert np.all(np.abs(cov.numpy() - ref) <= 2e-6 * scale)


def test_camera_matrices_match_reference(ref_pieces):
    for i in range(ref_pieces["cam_R"].shape[0]):
        cam = syn.make_camera(ref_pieces["cam_R"][i], ref_pieces["cam_T"][i], float(ref_pieces["cam_fovx"][i]),
                              float(ref_pieces["cam_fovy"][i]), 64, 48)
        np.testing.assert_allclose(cam.world_view_transform.numpy(), ref_pieces["cam_wv"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), ref_pieces["cam_full"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), ref_pieces["cam_center"][i], rtol=1e-5, atol=1e-6)


def _settings(cam, deg=3, bg=(0.0, 0.0, 0.0), mod=1.0):
    return orc.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                        torch.tensor(bg), mod, cam.world_view_transform, cam.full_proj_transform, deg,
                        cam.camera_center, False, False)


def test_oracle_switch_equivalences():
    """The reference's own self-consistency switches: convert_SHs_python and
    compute_cov3D_python must give the same image as the in-rasterizer paths."""
    W, H, P = 96, 64, 600
    sc = syn.make_scene(P, W, H, seed=3, log_scale_mean=-3.0)
    cam = syn.orbit_camera(W, H, 12.0, -7.0, 7.0)
    st = _settings(cam, deg=2, bg=(1.0, 1.0, 1.0))
    z2 = torch.zeros(P, 3)
    c0, r0, d0, a0 = orc.rasterize(sc.means3D, z2, sc.opacities, st, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    d = sc.means3D - cam.camera_center[None]
    dirs = d / d.norm(dim=1, keepdim=True)
    colors = torch.clamp_min(orc.eval_sh_rgb(2, sc.shs, dirs) + 0.5, 0.0)
    cov = orc.cov3d_from_scale_rot(sc.scales, sc.rotations, 1.0)
    c1, r1, d1, a1 = orc.rasterize(sc.means3D, z2, sc.opacities, st, colors_precomp=colors, cov3D_precomp=cov)
    assert torch.equal(r0, r1)
    torch.testing.assert_close(c0, c1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d0, d1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a0, a1, rtol=1e-5, atol=1e-6)
    assert (r0 > 0).sum() > 100
    assert float(a0.max()) <= 1.0 + 1e-5


def test_oracle_blend_against_scalar_loop():
    """Vectorised per-tile blend vs a literal per-pixel sequential loop (Appendix A wording)."""
    W, H, P = 40, 24, 120
    sc = syn.make_scene(P, W, H, seed=5, log_scale_mean=-2.5)
    cam = syn.default_camera(W, H)
    st = _settings(cam, deg=1, bg=(0.2, 0.5, 0.9))
    c, r, d, a, aux = orc.rasterize(sc.means3D, torch.zeros(P, 3), sc.opacities, st, shs=sc.shs, scales=sc.scales,
                                    rotations=sc.rotations, return_aux=True)
    pre, binning = aux["pre"], aux["binning"]
    gx = pre["grid"][0]
    xy, con, op, rgb, dep = [pre[k].numpy() for k in ("xy", "conic", "opacity", "rgb", "depth")]
    bg = np.array([0.2, 0.5, 0.9], dtype=np.float32)
    f = np.float32
    for (px, py) in [(0, 0), (13, 7), (39, 23), (20, 12), (31, 3)]:
        tile = (py // 16) * gx + px // 16
        s, e = binning["ranges"][tile]
        T = f(1.0); C = np.zeros(3, f); D = f(0); A = f(0); last = 0
        for k, gid in enumerate(binning["point_list"][s:e]):
            dx = f(xy[gid, 0] - f(px)); dy = f(xy[gid, 1] - f(py))
            power = f(f(-0.5) * f(f(con[gid, 0] * dx * dx) + f(con[gid, 2] * dy * dy)) - f(con[gid, 1] * dx * dy))
            if power > 0:
                continue
            alpha = min(f(0.99), f(op[gid] * np.exp(power, dtype=f)))
            if alpha < f(1.0 / 255.0):
                continue
            test_T = f(T * f(1 - alpha))
            if test_T < f(1e-4):
                break
            wgt = f(alpha * T)
            C += rgb[gid] * wgt; D += dep[gid] * wgt; A += wgt
            T = test_T; last = k + 1
        np.testing.assert_allclose(c[:, py, px].numpy(), C + T * bg, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(d[0, py, px].item(), D, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(a[0, py, px].item(), A, rtol=2e-5, atol=1e-6)
        assert int(aux["n_contrib"][py, px]) == last
        np.testing.assert_allclose(aux["final_T"][py, px].item(), T, rtol=1e-6)


def test_oracle_binning_invariants():
    W, H, P = 200, 120, 3000
    sc = syn.make_scene(P, W, H, seed=2)
    cam = syn.default_camera(W, H)
    st = _settings(cam)
    c, r, d, a, aux = orc.rasterize(sc.means3D, torch.zeros(P, 3), sc.opacities, st, shs=sc.shs, scales=sc.scales,
                                    rotations=sc.rotations, return_aux=True)
    b = aux["binning"]
    keys = b["keys_sorted"]
    assert np.all(keys[1:] >= keys[:-1])
    # stability: equal keys keep ascending Gaussian id
    eq = keys[1:] == keys[:-1]
    assert np.all(b["point_list"][1:][eq] > b["point_list"][:-1][eq])
    # ranges partition the sorted list by tile
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tile_of):
        s, e = b["ranges"][t]
        assert np.all(tile_of[s:e] == t) and (s == 0 or tile_of[s - 1] != t) and (e == len(keys) or tile_of[e] != t)
    assert b["num_rendered"] == int(aux["pre"]["tiles_touched"].sum())
    assert torch.equal(r > 0, aux["pre"]["tiles_touched"] > 0)
