"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
against the committed golden fixtures, and through size-independent properties at BASELINE sizes.

Bar (BASELINE.json north_star): tile assignment and sort indices bit-exact; rendered RGB + depth + alpha
and all gradients within 1e-4 relative (fp32)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

import parity_utils as pu
from scgaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = pu.REL_TOL


def _dev():
    assert torch.cuda.is_available(), "these tests need a GPU (run with gpurun)"
    return torch.device("cuda")


def _tiles_touched(fs) -> np.ndarray:
    r = pu.as_u32(fs["rects"])
    return ((r[:, 1] & 0xFFFF) * (r[:, 1] >> 16)).astype(np.int64)


def _stages(sc, cam, deg, bg, mod=1.0, mode="sh_sr", algo=0):
    from scgaussian_amd import rasterizer as R
    st = pu.hip_settings(cam, deg, bg, mod)
    lv = {k: v.to(_dev()) for k, v in pu.run_oracle_inputs(sc, cam, deg, mod, mode).items()}
    fs = R.forward_stages(st, lv["means3D"], lv["opacities"], shs=lv.get("shs"), colors_precomp=lv.get("colors_precomp"),
                          scales=lv.get("scales"), rotations=lv.get("rotations"), cov3D_precomp=lv.get("cov3D_precomp"),
                          want_keys=True, binning_algo=algo)
    torch.cuda.synchronize()
    return fs


CONFIGS = [
    # P, W, H, deg, bg, mod, camera, seed, log_scale_mean
    (2000, 200, 120, 3, (0.0, 0.0, 0.0), 1.0, ("default",), 0, -4.0),
    (3000, 250, 130, 2, (1.0, 1.0, 1.0), 0.8, ("orbit", 15.0, -8.0, 7.5), 3, -4.0),
    (1500, 97, 61, 1, (0.2, 0.4, 0.6), 1.0, ("orbit", -25.0, 12.0, 6.0), 4, -3.0),
    (1200, 64, 48, 0, (0.0, 0.0, 0.0), 1.3, ("default",), 5, -2.5),      # big splats: long per-tile lists
    (10000, 256, 256, 3, (0.0, 0.0, 0.0), 1.0, ("default",), 0, -4.0),   # BASELINE cfg1 / S1
]


def _cam(spec, W, H):
    return syn.default_camera(W, H) if spec[0] == "default" else syn.orbit_camera(W, H, *spec[1:])


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"P{c[0]}_{c[1]}x{c[2]}_d{c[3]}")
def test_forward_matches_oracle(cfg):
    P, W, H, deg, bg, mod, camspec, seed, lsm = cfg
    sc = syn.make_scene(P, W, H, seed=seed, log_scale_mean=lsm)
    cam = _cam(camspec, W, H)
    o = pu.run_oracle(sc, cam, deg, bg, mod)
    fs = _stages(sc, cam, deg, bg, mod)
    pre, b = o["aux"]["pre"], o["aux"]["binning"]
    # ---- integers: bit-exact -------------------------------------------------------------
    assert torch.equal(fs["radii"].cpu(), o["radii"])
    assert fs["num_rendered"] == b["num_rendered"]
    assert np.array_equal(np.cumsum(_tiles_touched(fs)).astype(np.uint32), b["point_offsets"])
    assert np.array_equal(pu.as_u32(fs["rects"]).astype(np.int64)[:, 0] & 0xFFFF, pre["rect"].numpy()[:, 0] * (o["radii"] > 0).numpy())
    for algo_fs in (fs, _stages(sc, cam, deg, bg, mod, algo=1)):      # depth-first binning and global 64-bit sort
        assert np.array_equal(algo_fs["keys_sorted"].cpu().numpy().view(np.uint64), b["keys_sorted"])
        assert np.array_equal(pu.as_u32(algo_fs["point_list"]), b["point_list"])
        assert np.array_equal(pu.as_u32(algo_fs["ranges"]), b["ranges"])
    vis = (o["radii"] > 0).numpy()
    depth_keys = pu.as_u32(fs["depth_keys"])
    assert np.array_equal(depth_keys[vis], pre["depth"].detach().numpy().view(np.uint32)[vis])
    assert np.all(depth_keys[~vis] == 0xFFFFFFFF)
    sp = fs["splats"].cpu().numpy()
    # values that feed integers are themselves bit-exact
    assert np.array_equal(sp[vis, 0:2], pre["xy"].detach().numpy()[vis])
    assert np.array_equal(sp[vis, 11], pre["depth"].detach().numpy()[vis])
    assert np.array_equal(sp[vis, 2:5], pre["conic"].detach().numpy()[vis])
    assert np.array_equal(sp[vis, 5], pre["opacity"].detach().numpy()[vis])
    assert pu.nrm_err(sp[vis, 8:11], pre["rgb"].detach().numpy()[vis]) < 1e-6
    # clamp flags agree wherever the colour is not within rounding of zero
    cl = fs["clamped"].cpu().numpy()
    for c in range(3):
        ref = pre["clamped"].numpy()[:, c]
        mism = ((cl >> c) & 1).astype(bool)[vis] != ref[vis]
        assert mism.sum() <= 1
    # ---- images: within tolerance ----------------------------------------------------------
    # pixels where a knife-edge alpha >= 1/255 / T >= 1e-4 decision fell differently are a class of their own: counted,
    # bounded, and excluded from the element-wise comparison (they still enter the tensor-scale one)
    flips = pu.threshold_flips(fs["n_contrib"], o["aux"]["n_contrib"])
    assert float(flips.float().mean()) < 1e-4
    for k, ref_img in (("color", o["color"]), ("depth", o["depth"]), ("alpha", o["alpha"])):
        assert pu.nrm_err(fs[k], ref_img) < TOL, k
        pu.assert_close(fs[k], ref_img, ("forward", k, P), mask=flips[None], frac_max=0.0)
    pu.assert_close(fs["final_T"], o["aux"]["final_T"], ("forward", "final_T", P), mask=flips, frac_max=0.0)


def _backward_cases():
    """All four input modes on the small configurations; the full BASELINE cfg1 / S1 scene in the default mode."""
    for cfg in CONFIGS[:4]:
        for mode in ("sh_sr", "col_sr", "sh_cov", "col_cov"):
            yield pytest.param(cfg, mode, id=f"P{cfg[0]}_{cfg[1]}x{cfg[2]}_d{cfg[3]}-{mode}")
    yield pytest.param(CONFIGS[4], "sh_sr", id="S1_full-sh_sr")


@pytest.mark.parametrize("cfg,mode", list(_backward_cases()))
def test_backward_matches_oracle_autograd(cfg, mode):
    P, W, H, deg, bg, mod, camspec, seed, lsm = cfg
    sc = syn.make_scene(P, W, H, seed=seed, log_scale_mean=lsm)
    cam = _cam(camspec, W, H)
    grads = syn.make_upstream_grads(W, H, seed=seed + 50)
    o = pu.run_oracle(sc, cam, deg, bg, mod, mode, grads=grads)
    h = pu.run_hip(sc, cam, deg, bg, mod, mode, grads=grads, leaves_from={k: v.detach() for k, v in o["leaves"].items()})
    assert torch.equal(h["radii"].cpu(), o["radii"])
    assert pu.nrm_err(h["color"], o["color"]) < TOL
    for k, g_ref in o["grads"].items():
        g = h["grads"][k]
        assert g is not None and g.shape == g_ref.shape, k
        pu.assert_close(g, g_ref, ("backward", mode, P, k))
    # means2D gradient slot: z component is zero, culled Gaussians get zero everywhere
    assert float(h["grads"]["means2D"][:, 2].abs().max()) == 0.0
    culled = (o["radii"] == 0)
    for k, g in h["grads"].items():
        assert float(g.cpu()[culled].abs().sum()) == 0.0, k


def test_partial_upstream_grads_and_alpha_only():
    """dL/ddepth / dL/dalpha absent (None) — e.g. the alpha loss exists only on DTU (reference train.py:167-168)."""
    P, W, H, deg = 1500, 112, 80, 2
    sc = syn.make_scene(P, W, H, seed=8, log_scale_mean=-3.2)
    cam = syn.orbit_camera(W, H, 8.0, 4.0, 7.0)
    dc, dd, da = syn.make_upstream_grads(W, H, seed=9)
    for grads in [(dc, None, None), (torch.zeros_like(dc), dd, None), (torch.zeros_like(dc), None, da)]:
        o = pu.run_oracle(sc, cam, deg, (0.5, 0.5, 0.5), grads=grads)
        h = pu.run_hip(sc, cam, deg, (0.5, 0.5, 0.5), grads=grads)
        for k, g_ref in o["grads"].items():
            pu.assert_close(h["grads"][k], g_ref, ("partial upstream", k))


def test_reference_switch_equivalences_on_hip():
    """The reference's own self-consistency switches: convert_SHs_python and compute_cov3D_python
    must render the same image as the in-kernel paths (gaussian_renderer/__init__.py:64-65,78-83)."""
    P, W, H, deg = 4000, 208, 144, 3
    sc = syn.make_scene(P, W, H, seed=21)
    cam = syn.orbit_camera(W, H, -12.0, 6.0, 7.0)
    base = pu.run_hip(sc, cam, deg, (0.0, 0.0, 0.0), mode="sh_sr")
    for mode in ("col_sr", "sh_cov", "col_cov"):
        other = pu.run_hip(sc, cam, deg, (0.0, 0.0, 0.0), mode=mode)
        assert torch.equal(base["radii"], other["radii"])
        assert pu.nrm_err(other["color"], base["color"]) < 1e-5
        assert pu.nrm_err(other["depth"], base["depth"]) < 1e-5
        assert pu.nrm_err(other["alpha"], base["alpha"]) < 1e-5


def test_against_committed_golden_fixtures():
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(HERE, "golden", "make_oracle_golden.py"))
    mog = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mog)
    gold = np.load(os.path.join(HERE, "golden", "oracle_small.npz"))
    for name, cfg in mog.CASES.items():
        sc, cam, grads = mog.make_case(cfg)
        fs = _stages(sc, cam, cfg["deg"], cfg["bg"], cfg["mod"], cfg["mode"])
        assert np.array_equal(fs["radii"].cpu().numpy(), gold[f"{name}_radii"])
        assert np.array_equal(np.cumsum(_tiles_touched(fs)).astype(np.uint32), gold[f"{name}_point_offsets"])
        assert np.array_equal(fs["keys_sorted"].cpu().numpy().view(np.uint64), gold[f"{name}_keys_sorted"])
        assert np.array_equal(pu.as_u32(fs["point_list"]), gold[f"{name}_point_list"])
        assert np.array_equal(pu.as_u32(fs["ranges"]), gold[f"{name}_ranges"])
        for k in ("color", "depth", "alpha", "final_T"):
            assert pu.nrm_err(fs[k], gold[f"{name}_{k}"]) < TOL, (name, k)
        assert (fs["n_contrib"].cpu().numpy() != gold[f"{name}_n_contrib"]).mean() < 1e-4
        h = pu.run_hip(sc, cam, cfg["deg"], cfg["bg"], cfg["mod"], cfg["mode"], grads=grads)
        for k, g in h["grads"].items():
            pu.assert_close(g, gold[f"{name}_grad_{k}"], ("golden", name, k))


# ---------------------------------------------------------------------------------------------------
# standalone integer primitives through the C ABI
# ---------------------------------------------------------------------------------------------------
def _sort_gpu(keys: np.ndarray, vals: np.ndarray, end_bit: int):
    from scgaussian_amd import _lib
    lib = _lib.load()
    dev = _dev()
    n = keys.size
    k_in = torch.from_numpy(keys.view(np.int64).copy()).to(dev)
    v_in = torch.from_numpy(vals.view(np.int32).copy()).to(dev)
    k_out, v_out = torch.empty_like(k_in), torch.empty_like(v_in)
    scratch = torch.empty(lib.scg_sort_scratch_bytes(n), dtype=torch.uint8, device=dev)
    _lib.check(lib.scg_sort_pairs(k_in.data_ptr(), v_in.data_ptr(), k_out.data_ptr(), v_out.data_ptr(), n, end_bit,
                                  scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream), "sort")
    torch.cuda.synchronize()
    return k_out.cpu().numpy().view(np.uint64), v_out.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 2047, 2048, 2049, 6000, 100_003, 1_000_000])
@pytest.mark.parametrize("end_bit", [8, 13, 41, 45, 64])
def test_radix_sort_pairs_is_the_stable_sort(n, end_bit):
    rng = np.random.default_rng(n * 131 + end_bit)
    mask = np.uint64((1 << end_bit) - 1) if end_bit < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    keys &= mask
    if n > 100:           # heavy duplicates + long equal runs (ties must keep input order)
        keys[: n // 3] = keys[0]
        keys[n // 2: n // 2 + n // 5] &= np.uint64(0xFF)
    vals = np.arange(n, dtype=np.uint32)
    ks, vs = _sort_gpu(keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ks, keys[order])
    assert np.array_equal(vs, vals[order])


@pytest.mark.parametrize("n", [1, 255, 256, 257, 4096, 65_537, 1_000_001])
def test_inclusive_scan_u32(n):
    from scgaussian_amd import _lib
    lib = _lib.load()
    dev = _dev()
    rng = np.random.default_rng(n)
    x = rng.integers(0, 300, size=n, dtype=np.uint32)
    t_in = torch.from_numpy(x.view(np.int32).copy()).to(dev)
    t_out = torch.empty_like(t_in)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.scg_scan_scratch_bytes(n), dtype=torch.uint8, device=dev)
    _lib.check(lib.scg_inclusive_scan_u32(t_in.data_ptr(), t_out.data_ptr(), n, total.data_ptr(), scratch.data_ptr(),
                                          scratch.numel(), torch.cuda.current_stream().cuda_stream), "scan")
    torch.cuda.synchronize()
    ref = np.cumsum(x.astype(np.uint64)).astype(np.uint32)
    assert np.array_equal(t_out.cpu().numpy().view(np.uint32), ref)
    assert int(total.item()) & 0xFFFFFFFF == int(ref[-1])


# ---------------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------------
def test_everything_culled_renders_background():
    P, W, H = 300, 70, 50
    sc = syn.make_scene(P, W, H, seed=1)
    sc = sc._replace(means3D=sc.means3D * torch.tensor([1.0, 1.0, -1.0]))     # all behind the camera
    cam = syn.default_camera(W, H)
    bg = (0.25, 0.5, 0.75)
    grads = syn.make_upstream_grads(W, H)
    h = pu.run_hip(sc, cam, 3, bg, grads=grads)
    assert int((h["radii"] != 0).sum()) == 0
    assert torch.allclose(h["color"].cpu(), torch.tensor(bg)[:, None, None].expand(3, H, W))
    assert float(h["depth"].abs().max()) == 0.0 and float(h["alpha"].abs().max()) == 0.0
    for k, g in h["grads"].items():
        assert float(g.abs().max()) == 0.0, k


def test_single_and_huge_gaussians():
    """P = 1; and a splat whose rectangle covers every tile (hundreds of instances from one Gaussian: the
    wave-cooperative duplicateWithKeys path) — against the oracle."""
    W, H = 330, 200
    cam = syn.default_camera(W, H)
    means = torch.tensor([[0.0, 0.0, 5.0], [0.3, -0.2, 6.0], [-0.5, 0.4, 4.0]])
    scales = torch.tensor([[2.5, 2.0, 0.5], [0.02, 0.03, 0.02], [0.8, 0.05, 0.05]])
    rots = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.9, 0.1, 0.3, 0.2], [0.7, 0.0, 0.0, 0.714]])
    rots = rots / rots.norm(dim=1, keepdim=True)
    opac = torch.tensor([[0.6], [0.9], [0.8]])
    g = torch.Generator().manual_seed(2)
    shs = torch.cat([torch.rand(3, 1, 3, generator=g), torch.randn(3, 15, 3, generator=g) * 0.1], 1)
    for sel in ([0], [1], [0, 1, 2]):
        sc = syn.Scene(means[sel], scales[sel], rots[sel], opac[sel], shs[sel])
        grads = syn.make_upstream_grads(W, H, seed=4)
        o = pu.run_oracle(sc, cam, 3, (0.1, 0.1, 0.1), grads=grads)
        fs = _stages(sc, cam, 3, (0.1, 0.1, 0.1))
        b = o["aux"]["binning"]
        assert torch.equal(fs["radii"].cpu(), o["radii"])
        assert np.array_equal(pu.as_u32(fs["point_list"]), b["point_list"])
        assert np.array_equal(pu.as_u32(fs["ranges"]), b["ranges"])
        if 0 in sel:
            assert b["num_rendered"] >= ((W + 15) // 16) * ((H + 15) // 16)     # covers every tile
        fs1 = _stages(sc, cam, 3, (0.1, 0.1, 0.1), algo=1)
        assert np.array_equal(pu.as_u32(fs1["point_list"]), b["point_list"])
        h = pu.run_hip(sc, cam, 3, (0.1, 0.1, 0.1), grads=grads)
        assert pu.nrm_err(h["color"], o["color"]) < TOL
        for k, g_ref in o["grads"].items():
            pu.assert_close(h["grads"][k], g_ref, ("single/huge", tuple(sel), k))


def test_opacity_edge_values():
    """Opacities 0, far below / just below / exactly at / just above the 1/255 blending cut-off, and exactly 1 (alpha
    clamps at 0.99): the per-quadrant culling threshold is derived from log(255 * opacity) — -inf, negative, ~0 —
    and must never drop a contribution the per-pixel test would keep."""
    W, H = 96, 64
    cam = syn.default_camera(W, H)
    P = 240
    sc = syn.make_scene(P, W, H, seed=9, log_scale_mean=-2.6)
    edge = torch.tensor([0.0, 1e-6, 1.0 / 255.0 - 1e-5, 1.0 / 255.0, 1.0 / 255.0 + 1e-5, 0.004, 0.0045, 1.0])
    opac = edge.repeat(P // edge.numel())[:, None].contiguous()
    sc = syn.Scene(sc.means3D, sc.scales, sc.rotations, opac, sc.shs)
    grads = syn.make_upstream_grads(W, H, seed=6)
    o = pu.run_oracle(sc, cam, 3, (0.0, 0.0, 0.0), grads=grads)
    h = pu.run_hip(sc, cam, 3, (0.0, 0.0, 0.0), grads=grads)
    assert torch.equal(h["radii"].cpu(), o["radii"])
    for k in ("color", "depth", "alpha"):
        assert pu.nrm_err(h[k], o[k]) < TOL, k
    for k, g_ref in o["grads"].items():
        pu.assert_close(h["grads"][k], g_ref, ("opacity edges", k))
    # pixels' contributor counts are exact integers: the culling never changed who blends
    fs = _stages(sc, cam, 3, (0.0, 0.0, 0.0))
    assert np.array_equal(pu.as_u32(fs["n_contrib"]), o["aux"]["n_contrib"].numpy().astype(np.uint32))


def test_empty_input():
    W, H = 40, 24
    cam = syn.default_camera(W, H)
    sc = syn.Scene(torch.zeros(0, 3), torch.zeros(0, 3), torch.zeros(0, 4), torch.zeros(0, 1), torch.zeros(0, 16, 3))
    from scgaussian_amd import rasterizer as R
    st = pu.hip_settings(cam, 3, (0.3, 0.2, 0.1))
    dev = _dev()
    fs = R.forward_stages(st, sc.means3D.to(dev), sc.opacities.to(dev), shs=sc.shs.to(dev), scales=sc.scales.to(dev),
                          rotations=sc.rotations.to(dev))
    torch.cuda.synchronize()
    assert fs["num_rendered"] == 0
    assert torch.allclose(fs["color"].cpu(), torch.tensor([0.3, 0.2, 0.1])[:, None, None].expand(3, H, W))


# ---------------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties (the oracle is too slow there)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["S2", "S3"])
def test_full_size_properties(name):
    w = syn.WORKLOADS[name]
    P, W, H = w["P"], w["width"], w["height"]
    sc = syn.make_scene(P, W, H, seed=0)
    cam = syn.default_camera(W, H)
    fs = _stages(sc, cam, 3, (0.0, 0.0, 0.0))
    R_ = fs["num_rendered"]
    keys = fs["keys_sorted"].cpu().numpy().view(np.uint64)
    plist = pu.as_u32(fs["point_list"])
    ranges = pu.as_u32(fs["ranges"]).astype(np.int64)
    radii = fs["radii"].cpu().numpy()
    # tiles touched: sum == R; > 0 exactly for visible Gaussians
    tiles = _tiles_touched(fs)
    assert tiles.sum() == R_ and np.array_equal(tiles > 0, radii > 0)
    # sortedness + stability of ties
    assert np.all(keys[1:] >= keys[:-1])
    eq = keys[1:] == keys[:-1]
    assert np.all(plist[1:][eq] > plist[:-1][eq])
    # every Gaussian appears exactly tiles_touched times (the sort is a permutation of the duplicates)
    assert np.array_equal(np.bincount(plist, minlength=P), tiles)
    # depth bits in the key are the Gaussian's depth
    depth_bits = fs["splats"][:, 11].contiguous().cpu().numpy().view(np.uint32)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), depth_bits[plist])
    # ranges partition the list by tile id
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    n_tiles = ranges.shape[0]
    counts = np.bincount(tile_of, minlength=n_tiles)
    assert np.array_equal(ranges[:, 1] - ranges[:, 0], counts)
    touched = counts > 0
    assert np.array_equal(ranges[touched, 0], np.searchsorted(tile_of, np.nonzero(touched)[0], side="left"))
    # blend: alpha + T_final == 1, colour bounded by accumulated weight * max rgb, determinism of the forward
    a, T = fs["alpha"][0], fs["final_T"]
    assert float((a + T - 1).abs().max()) < 1e-5
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-6
    fs2 = _stages(sc, cam, 3, (0.0, 0.0, 0.0), algo=1)        # the global 64-bit sort gives the identical list
    assert torch.equal(fs["point_list"], fs2["point_list"]) and torch.equal(fs["ranges"], fs2["ranges"])
    assert torch.equal(fs["keys_sorted"], fs2["keys_sorted"])
    assert torch.equal(fs["color"], fs2["color"]) and torch.equal(fs["depth"], fs2["depth"])
    # white background adds exactly T_final to every channel
    fs_w = _stages(sc, cam, 3, (1.0, 1.0, 1.0))
    assert pu.nrm_err(fs_w["color"] - fs["color"], T[None].expand(3, H, W)) < 1e-5


def test_full_size_backward_is_linear_in_upstream_grads():
    """S2: backward(a*g1 + b*g2) == a*backward(g1) + b*backward(g2) (float atomics: tolerance, not bitwise)."""
    w = syn.WORKLOADS["S2"]
    P, W, H = w["P"], w["width"], w["height"]
    sc = syn.make_scene(P, W, H, seed=0)
    cam = syn.default_camera(W, H)
    g1 = syn.make_upstream_grads(W, H, seed=1)
    g2 = syn.make_upstream_grads(W, H, seed=2)
    g12 = tuple(0.7 * x - 1.9 * y for x, y in zip(g1, g2))
    h1 = pu.run_hip(sc, cam, 3, (0.2, 0.3, 0.4), grads=g1)
    h2 = pu.run_hip(sc, cam, 3, (0.2, 0.3, 0.4), grads=g2)
    h12 = pu.run_hip(sc, cam, 3, (0.2, 0.3, 0.4), grads=g12)
    for k in h1["grads"]:
        comb = 0.7 * h1["grads"][k] - 1.9 * h2["grads"][k]
        assert torch.isfinite(h12["grads"][k]).all()
        assert pu.nrm_err(h12["grads"][k], comb) < TOL, k


# ---------------------------------------------------------------------------------------------------
# render() twin (gaussian_renderer/__init__.py:20-118): switches, result dict, gradient slots
# ---------------------------------------------------------------------------------------------------
class _MockModel:
    """Read interface of the reference's GaussianModel (scene/gaussian_model.py:105-155)."""

    def __init__(self, sc, active_sh_degree, dev):
        self.max_sh_degree = 3
        self.active_sh_degree = active_sh_degree
        self._xyz = sc.means3D.to(dev).requires_grad_(True)
        self._features = sc.shs.to(dev).requires_grad_(True)
        self._opacity = sc.opacities.to(dev).requires_grad_(True)
        self._scaling = sc.scales.to(dev).requires_grad_(True)
        self._rotation = sc.rotations.to(dev).requires_grad_(True)

    get_xyz = property(lambda s: s._xyz)
    get_features = property(lambda s: s._features)
    get_opacity = property(lambda s: s._opacity)
    get_scaling = property(lambda s: s._scaling)
    get_rotation = property(lambda s: s._rotation)

    def get_covariance(self, scaling_modifier=1.0):
        from oracle.torch_rasterizer import cov3d_from_scale_rot
        return cov3d_from_scale_rot(self._scaling, self._rotation, scaling_modifier)


def test_render_twin_switches_and_dict():
    from scgaussian_amd.render import PipelineParams, render
    dev = _dev()
    P, W, H = 3000, 160, 112
    sc = syn.make_scene(P, W, H, seed=31)
    cam = syn.orbit_camera(W, H, 9.0, -4.0, 7.0).to(dev)
    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    pc = _MockModel(sc, 2, dev)
    base = render(cam, pc, PipelineParams(), bg)
    assert set(base) == {"render", "rendered_depth", "rendered_alpha", "viewspace_points", "visibility_filter", "radii"}
    assert base["render"].shape == (3, H, W) and base["rendered_depth"].shape == (1, H, W)
    assert base["rendered_alpha"].shape == (1, H, W) and base["radii"].dtype == torch.int32
    assert torch.equal(base["visibility_filter"], base["radii"] > 0)
    for pipe in (PipelineParams(convert_SHs_python=True), PipelineParams(compute_cov3D_python=True),
                 PipelineParams(True, True)):
        other = render(cam, pc, pipe, bg)
        assert torch.equal(other["radii"], base["radii"])
        assert pu.nrm_err(other["render"], base["render"]) < 1e-5
        assert pu.nrm_err(other["rendered_depth"], base["rendered_depth"]) < 1e-5
    # gradients flow to the model tensors and to the viewspace slot, identically through both colour paths
    loss = base["render"].sum() + 0.1 * base["rendered_depth"].sum() + base["rendered_alpha"].mean()
    loss.backward()
    g_xyz = pc._xyz.grad.clone()
    vs = base["viewspace_points"].grad
    assert vs is not None and vs.shape == (P, 3) and float(vs[:, 2].abs().max()) == 0.0
    assert float(vs[base["radii"] > 0][:, :2].abs().sum()) > 0
    pc2 = _MockModel(sc, 2, dev)
    o2 = render(cam, pc2, PipelineParams(True, True), bg)
    (o2["render"].sum() + 0.1 * o2["rendered_depth"].sum() + o2["rendered_alpha"].mean()).backward()
    assert pu.nrm_err(pc2._xyz.grad, g_xyz) < 1e-4
    assert pu.nrm_err(pc2._features.grad, pc._features.grad) < 1e-4
    assert pu.nrm_err(pc2._scaling.grad, pc._scaling.grad) < 1e-4
    # override_color path
    oc = render(cam, pc, PipelineParams(), bg, override_color=torch.ones(P, 3, device=dev) * 0.5)
    assert pu.nrm_err(oc["render"][0], 0.5 * base["rendered_alpha"][0]) < 1e-5


def test_tile_sort_handles_long_lists_and_depth_ties():
    """Many Gaussians stacked on one tile (lists of 1 500 .. 20 000 entries: one case per tile-sort path) or spread
    over a few tiles (mixed lengths in one launch) with heavily duplicated depths (ties must come out in ascending id): both binning
    algorithms must agree with each other bit for bit, and with the numpy stable sort."""
    dev = _dev()
    from scgaussian_amd import rasterizer as R
    W, H = 64, 48
    cam = syn.default_camera(W, H)
    longest_seen = []
    for P, spread in ((1500, 0.02), (6000, 0.02), (12000, 0.02), (20000, 0.02), (40000, 0.5)):
        g = torch.Generator().manual_seed(P)
        xy = (torch.rand(P, 2, generator=g) - 0.5) * spread
        z = torch.randint(0, 37, (P,), generator=g).float() * 0.25 + 3.0          # only 37 distinct depths
        means = torch.cat([xy * z[:, None], z[:, None]], 1)
        sc = syn.Scene(means, torch.full((P, 3), 0.004), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                       torch.full((P, 1), 0.02), torch.rand(P, 16, 3, generator=g) * 0.1)
        outs = []
        for algo in (0, 1):
            fs = _stages(sc, cam, 0, (0.0, 0.0, 0.0), algo=algo)
            outs.append(fs)
        a, b = outs
        assert a["num_rendered"] == b["num_rendered"] > 0
        assert torch.equal(a["point_list"], b["point_list"])
        assert torch.equal(a["ranges"], b["ranges"])
        assert torch.equal(a["keys_sorted"], b["keys_sorted"])
        keys = a["keys_sorted"].cpu().numpy().view(np.uint64)
        plist = pu.as_u32(a["point_list"])
        assert np.all(keys[1:] >= keys[:-1])
        eq = keys[1:] == keys[:-1]
        assert eq.sum() > 100 and np.all(plist[1:][eq] > plist[:-1][eq])
        counts = pu.as_u32(a["ranges"])[:, 1].astype(np.int64) - pu.as_u32(a["ranges"])[:, 0]
        longest_seen += [int(c) for c in counts if c > 0]
        assert torch.equal(a["color"], b["color"])
    # the same stacks with CONTINUOUS depths (no ties): the long lists take the global-memory bucket sort instead of the
    # bitonic fallback — still the stable sort, bit for bit
    for P in (9000, 30000, 70000):
        g = torch.Generator().manual_seed(P + 1)
        xy = (torch.rand(P, 2, generator=g) - 0.5) * 0.02
        z = torch.rand(P, generator=g) * 9.0 + 3.0
        means = torch.cat([xy * z[:, None], z[:, None]], 1)
        sc = syn.Scene(means, torch.full((P, 3), 0.004), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                       torch.full((P, 1), 0.02), torch.rand(P, 16, 3, generator=g) * 0.1)
        a, b = (_stages(sc, cam, 0, (0.0, 0.0, 0.0), algo=algo) for algo in (0, 1))
        assert a["num_rendered"] == b["num_rendered"] >= P
        assert torch.equal(a["point_list"], b["point_list"])
        assert torch.equal(a["keys_sorted"], b["keys_sorted"])
        keys = a["keys_sorted"].cpu().numpy().view(np.uint64)
        assert np.all(keys[1:] >= keys[:-1])
    # every sort path was exercised: 4-wave radix (<= 2048), 16-wave radix (<= 8192), LDS bitonic (<= 16384), global
    ls = np.array(longest_seen)
    assert ((ls > 1) & (ls <= 2048)).any() and ((ls > 2048) & (ls <= 8192)).any(), sorted(set(longest_seen))[-8:]
    assert ((ls > 8192) & (ls <= 16384)).any() and (ls > 16384).any(), sorted(set(longest_seen))[-8:]


def test_speculative_launch_and_overflow_retry():
    """Stages 2-3 are normally enqueued with an upper-bound guess of num_rendered (no host stall).  The result must
    be identical to the exact-size path, also when the guess was too small (clipped lists -> automatic re-run) and
    when it was far too large."""
    from scgaussian_amd import rasterizer as R
    dev = _dev()
    P, W, H = 6000, 240, 160
    sc = syn.make_scene(P, W, H, seed=17).to(dev)
    cam = syn.default_camera(W, H)
    st = pu.hip_settings(cam, 3, (0.1, 0.2, 0.3))
    kw = dict(shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    old = R.SPECULATIVE_LAUNCH
    try:
        R.SPECULATIVE_LAUNCH = False
        exact = R.forward_stages(st, sc.means3D, sc.opacities, **kw)
        Rn = exact["num_rendered"]
        assert Rn > 1000
        for hint in (Rn // 3, Rn, Rn + 1, 4 * Rn):
            spec = R.forward_stages(st, sc.means3D, sc.opacities, capacity_hint=hint, **kw)
            torch.cuda.synchronize()
            assert spec["num_rendered"] == Rn
            assert torch.equal(spec["color"], exact["color"]) and torch.equal(spec["depth"], exact["depth"])
            assert torch.equal(spec["point_list"], exact["point_list"]) and torch.equal(spec["ranges"], exact["ranges"])
            assert torch.equal(spec["n_contrib"], exact["n_contrib"])
        # the module-level switch: second call uses the learned hint
        R.SPECULATIVE_LAUNCH = True
        R._SPEC_STATE.clear()
        a = R.forward_stages(st, sc.means3D, sc.opacities, **kw)
        b = R.forward_stages(st, sc.means3D, sc.opacities, **kw)
        assert torch.equal(a["color"], exact["color"]) and torch.equal(b["color"], exact["color"])
        # backward through a speculative forward
        grads = syn.make_upstream_grads(W, H, seed=3)
        h1 = pu.run_hip(sc.to("cpu"), cam, 3, (0.1, 0.2, 0.3), grads=grads)
        R.SPECULATIVE_LAUNCH = False
        h0 = pu.run_hip(sc.to("cpu"), cam, 3, (0.1, 0.2, 0.3), grads=grads)
        for k in h0["grads"]:
            assert pu.nrm_err(h1["grads"][k], h0["grads"][k]) < 1e-5, k
    finally:
        R.SPECULATIVE_LAUNCH = old


def test_a_camera_is_known_by_its_matrix_not_by_the_address_of_its_tensor():
    """the per-camera host hints (capacity of point_list, tile costs, long-list counts) are keyed by the
    CONTENT of the view matrix.  Camera B (many instances) is rendered, then camera A (less than half of them); A's tensors are
    freed and B's matrices are uploaded again — into fresh tensors, wherever the allocator puts them, A's old address
    included.  The re-uploaded B finds B's capacity: ONE scg_forward, no overflow re-run (keyed by data_ptr it started from
    A's bound, or from the shape's latest bound = A's, and ran twice)."""
    from scgaussian_amd import _lib, rasterizer as R
    dev = _dev()
    P, W, H = 40000, 320, 208                                   # (173 k instances from near, 62 k from far: the bound A
    sc = syn.make_scene(P, W, H, seed=21, log_scale_mean=-3.0).to(dev)      #  leaves behind, 74 k, is far below B's count)
    near, far = syn.orbit_camera(W, H, 5.0, 2.0, 7.0), syn.orbit_camera(W, H, 5.0, 2.0, 200.0)
    R._SPEC_STATE.clear()
    R._FRAME_CACHE.clear()
    lib = _lib.load()
    calls = []
    real = lib.scg_forward

    def counting(*a):
        calls.append(1)
        return real(*a)

    def render(st):
        out = R.forward_fused(st, sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None, False)
        if out is None:                                         # first sight of the shape: the staged path sets the bound
            R.forward_stages(st, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
            out = R.forward_fused(st, sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None, False)
        torch.cuda.synchronize()
        return out[4]["num_rendered"], out[0].clone()

    st_b = pu.hip_settings(near, 3, (0.0, 0.0, 0.0))
    r_b, img_b = render(st_b)
    st_a = pu.hip_settings(far, 3, (0.0, 0.0, 0.0))
    r_a, _ = render(st_a)
    assert r_b > 2 * r_a, (r_b, r_a)                           # (beyond the head room of A's bound: checked below)
    r_a2, _ = render(st_a)                                      # the shape's latest bound is A's now
    spec = R._spec_state(dev)
    assert spec.hint[(P, W, H)] < r_b
    key_b = R._camera_key(st_b.viewmatrix)
    del st_a
    R._FRAME_CACHE.clear()
    R._CAM_KEYS.clear()                                         # (nothing keeps A's or B's tensors alive any more)
    st_b2 = pu.hip_settings(near, 3, (0.0, 0.0, 0.0))           # B again, in new tensors
    assert st_b2.viewmatrix.data_ptr() != st_b.viewmatrix.data_ptr() and R._camera_key(st_b2.viewmatrix) == key_b
    lib.scg_forward = counting
    try:
        r_b2, img_b2 = render(st_b2)
    finally:
        lib.scg_forward = real
    assert r_b2 == r_b and len(calls) == 1, (r_b2, r_b, calls)
    assert torch.equal(img_b2, img_b)
    # a matrix rewritten in place is another camera: its key follows the tensor's version counter
    st_b2.viewmatrix.copy_(pu.hip_settings(far, 3, (0.0, 0.0, 0.0)).viewmatrix)
    assert R._camera_key(st_b2.viewmatrix) != key_b


def test_one_call_path_equals_the_staged_path_and_recovers_from_a_small_bound():
    """scg_forward / scg_backward (one library call per direction, the binding's fast path) against the five staged
    calls: forward outputs bit-identical, gradients equal up to the order of the float atomics; a capacity that is too
    small is detected from num_rendered and the call repeated; the stage-event hook times the stages."""
    from scgaussian_amd import rasterizer as R
    dev = _dev()
    P, W, H = 5000, 320, 208
    sc = syn.make_scene(P, W, H, seed=12, log_scale_mean=-3.4).to(dev)
    cam = syn.orbit_camera(W, H, 11.0, 3.0, 7.0)
    st = pu.hip_settings(cam, 3, (0.3, 0.2, 0.1))
    R._SPEC_STATE.clear()
    assert R.forward_fused(st, sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None, False) is None
    exact = R.forward_stages(st, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
    Rn = exact["num_rendered"]
    spec = R._spec_state(sc.means3D.device)
    assert spec.hint[(P, W, H)] >= Rn
    for cap in (None, 4096, Rn - 1, Rn):                     # learned capacity, far too small, one short, exact
        if cap is not None:
            spec.hint[(P, W, H)] = cap                                   # the shape's bound ...
            for k in [k for k in spec.cam_hint if k[:2] == (W, H)]:      # ... and this camera's
                spec.cam_hint[k] = (cap, spec.cam_hint[k][1], P)
        out = R.forward_fused(st, sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None, True)
        torch.cuda.synchronize()
        c, radii, d, a, state = out
        assert state["num_rendered"] == Rn and state["cap"] >= Rn
        assert torch.equal(c, exact["color"]) and torch.equal(d, exact["depth"]) and torch.equal(a, exact["alpha"])
        assert torch.equal(radii, exact["radii"])
        ws, plan = state["ws"], state["plan"]
        pl = ws[plan.point_list: plan.point_list + 4 * Rn].view(torch.int32)
        assert torch.equal(pl, exact["point_list"])
        nt = exact["ranges"].numel()
        assert torch.equal(ws[plan.ranges: plan.ranges + 4 * nt].view(torch.int32).view(-1, 2), exact["ranges"])
        assert spec.hint[(P, W, H)] >= Rn
    # the camera's bound outlives a change of the Gaussian count (densification): rescaled by the ratio of the counts, so the
    # first render at the new count takes the one-call path with room for its instances — no staged re-run, no overflow
    P2 = P + 700
    sc2 = syn.make_scene(P2, W, H, seed=12, log_scale_mean=-3.4).to(dev)
    assert (W, H, R._camera_key(st.viewmatrix)) in spec.cam_hint and (P2, W, H) not in spec.hint
    out2 = R.forward_fused(st, sc2.means3D, sc2.opacities, sc2.shs, None, sc2.scales, sc2.rotations, None, False)
    assert out2 is not None and out2[4]["cap"] >= out2[4]["num_rendered"] > 0
    assert spec.cam_hint[(W, H, R._camera_key(st.viewmatrix))][2] == P2
    # a forward that ran without backward state cannot be differentiated: the binding says so instead of returning garbage
    with pytest.raises(Exception, match="without backward state"):
        R.backward_fused(out2[4]["inputs"], out2[1], out2[4], torch.zeros(3, H, W, device=dev), None, None)
    # gradients: autograd through the one-call path vs through the staged path
    grads = syn.make_upstream_grads(W, H, seed=3)
    timer = R.StageTimer()
    R.set_stage_timer(timer)
    try:
        h1 = pu.run_hip(sc.to("cpu"), cam, 3, (0.3, 0.2, 0.1), grads=grads)
    finally:
        R.set_stage_timer(None)
    times = timer.summary()
    assert set(times) == {"geometry_forward", "binning", "blend_forward", "blend_backward", "geometry_backward"}
    assert all(0.0 < ms < 50.0 for ms, _ in times.values())
    old = R.SPECULATIVE_LAUNCH
    R.SPECULATIVE_LAUNCH = False
    try:
        h0 = pu.run_hip(sc.to("cpu"), cam, 3, (0.3, 0.2, 0.1), grads=grads)
    finally:
        R.SPECULATIVE_LAUNCH = old
    assert torch.equal(h1["color"], h0["color"])
    for k in h0["grads"]:
        assert pu.nrm_err(h1["grads"][k], h0["grads"][k]) < 1e-5, k


def test_huge_image_falls_back_to_global_sort_and_million_gaussians():
    """(1) an image with more tiles than the LDS histograms can hold (7680x4320 = 129 600 tiles) silently takes the
    global-sort path and still matches the explicit request for it; (2) S4-sized input (1 M Gaussians) runs and
    satisfies the list invariants."""
    from scgaussian_amd import _lib
    lib = _lib.load()
    assert lib.scg_binning_accepts_bound(1000, 7680, 4320, 0) == 0
    assert lib.scg_binning_accepts_bound(1000, 1920, 1080, 0) == 1
    W, H, P = 7680, 4320, 3000
    sc = syn.make_scene(P, W, H, seed=9, log_scale_mean=-2.5)
    cam = syn.default_camera(W, H)
    a = _stages(sc, cam, 1, (0.0, 0.0, 0.0), algo=0)
    b = _stages(sc, cam, 1, (0.0, 0.0, 0.0), algo=1)
    assert a["num_rendered"] == b["num_rendered"] > P
    assert torch.equal(a["point_list"], b["point_list"]) and torch.equal(a["ranges"], b["ranges"])
    assert torch.equal(a["color"], b["color"])
    keys = a["keys_sorted"].cpu().numpy().view(np.uint64)
    assert np.all(keys[1:] >= keys[:-1])
    del a, b
    w = syn.WORKLOADS["S4"]
    sc = syn.make_scene(w["P"], w["width"], w["height"], seed=0)
    fs = _stages(sc, syn.default_camera(w["width"], w["height"]), 3, (0.0, 0.0, 0.0))
    keys = fs["keys_sorted"].cpu().numpy().view(np.uint64)
    plist = pu.as_u32(fs["point_list"])
    assert np.all(keys[1:] >= keys[:-1])
    eq = keys[1:] == keys[:-1]
    assert np.all(plist[1:][eq] > plist[:-1][eq])
    assert np.array_equal(np.bincount(plist, minlength=w["P"]), _tiles_touched(fs))
    assert float((fs["alpha"][0] + fs["final_T"] - 1).abs().max()) < 1e-5


def test_mark_visible_is_the_near_plane_test_of_the_geometry_stage():
    dev = _dev()
    from scgaussian_amd import rasterizer as R
    W, H = 96, 64
    cam = syn.orbit_camera(W, H, 25.0, -10.0, 6.0)
    sc = syn.make_scene(4000, W, H, seed=11)
    means = sc.means3D.clone()
    means[:500, 2] = torch.linspace(-3.0, 0.6, 500)               # behind / just in front of the camera plane
    st = pu.hip_settings(cam, 3, (0.0, 0.0, 0.0))
    rast = R.GaussianRasterizer(st)
    vis = rast.markVisible(means.to(dev))
    z = (torch.cat([means, torch.ones(len(means), 1)], 1) @ cam.world_view_transform)[:, 2]
    assert torch.equal(vis.cpu(), z > 0.2)
    _, radii, _, _ = rast(means3D=means.to(dev), means2D=torch.zeros_like(means).to(dev), opacities=sc.opacities.to(dev),
                          shs=sc.shs.to(dev), scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
    assert not bool(((radii > 0) & ~vis).any())                    # nothing invisible is ever rasterized
    assert int(vis.sum()) > 1000


def _random_config(seed):
    """Odd image sizes (partial tiles on both axes, widths that are not a multiple of 8 or 16), varying splat scale,
    SH degree, background, scale modifier and camera — drawn from a seeded generator."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(17, 180)), int(rng.integers(9, 130))
    P = int(rng.integers(1, 1800))
    deg = int(rng.integers(0, 4))
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3).round(2))
    mod = float(rng.choice([1.0, 0.7, 1.4]))
    cam = ("default",) if seed % 2 == 0 else ("orbit", float(rng.uniform(-30, 30)), float(rng.uniform(-15, 15)), float(rng.uniform(5.5, 8.0)))
    return P, W, H, deg, bg, mod, cam, seed, float(rng.uniform(-4.2, -2.6))


@pytest.mark.parametrize("seed", range(6))
def test_random_configurations_forward_and_backward(seed):
    P, W, H, deg, bg, mod, camspec, sd, lsm = _random_config(seed)
    sc = syn.make_scene(P, W, H, seed=sd, log_scale_mean=lsm)
    cam = _cam(camspec, W, H)
    grads = syn.make_upstream_grads(W, H, seed=20 + seed)
    o = pu.run_oracle(sc, cam, deg, bg, mod, grads=grads)
    fs = _stages(sc, cam, deg, bg, mod)
    b = o["aux"]["binning"]
    assert torch.equal(fs["radii"].cpu(), o["radii"]), (P, W, H)
    assert np.array_equal(pu.as_u32(fs["point_list"]), b["point_list"])
    assert np.array_equal(pu.as_u32(fs["ranges"]), b["ranges"])
    h = pu.run_hip(sc, cam, deg, bg, mod, grads=grads)
    for k in ("color", "depth", "alpha"):
        assert pu.nrm_err(h[k], o[k]) < TOL, (k, P, W, H, deg)
    for k, g_ref in o["grads"].items():
        assert pu.nrm_err(h["grads"][k], g_ref) < TOL, (k, P, W, H, deg)


def test_host_state_isolation_interleaved_calls_retain_graph_and_side_stream():
    """The host wrapper caches frame structs, reuses a pinned scratch buffer and hands a pre-cleared gradient buffer
    from the forward to the backward: interleave two scenes, run a backward twice (retain_graph) and run on a
    non-default stream — every result must equal the plain sequential one."""
    dev = _dev()
    from scgaussian_amd import rasterizer as R

    def build(seed, W, H, P):
        sc = syn.make_scene(P, W, H, seed=seed)
        cam = syn.orbit_camera(W, H, 10.0 * seed, 5.0, 7.0)
        st = pu.hip_settings(cam, 3, (0.1, 0.2, 0.3))
        leaves = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
        ups = [u.to(dev) for u in syn.make_upstream_grads(W, H, seed=seed)]
        return R.GaussianRasterizer(st), leaves, ups

    def fwd(rast, lv):
        m, o, s, sc_, r = lv
        return rast(means3D=m, means2D=torch.zeros_like(m), opacities=o, shs=s, scales=sc_, rotations=r)

    def grads_of(lv):
        g = [p.grad.clone() for p in lv]
        for p in lv:
            p.grad = None
        return g

    A, B = build(1, 120, 72, 1500), build(2, 88, 104, 2300)
    ref = []
    for rast, lv, ups in (A, B):                                  # plain sequential reference
        c, _, d, a = fwd(rast, lv)
        torch.autograd.backward([c, d, a], ups)
        ref.append((c.detach().clone(), grads_of(lv)))
    # interleaved: forward A, forward B, backward A, backward B
    outs = [fwd(rast, lv) for rast, lv, _ in (A, B)]
    for (c, _, d, a), (_, lv, ups) in zip(outs, (A, B)):
        torch.autograd.backward([c, d, a], ups, retain_graph=True)
    for i, (_, lv, ups) in enumerate((A, B)):
        assert torch.equal(outs[i][0], ref[i][0])
        for g, r in zip(grads_of(lv), ref[i][1]):
            assert pu.nrm_err(g, r) < 1e-6
    # the same graphs once more (the pre-cleared buffer is gone: memset path) — same gradients
    for (c, _, d, a), (_, lv, ups), (_, gref) in zip(outs, (A, B), ref):
        torch.autograd.backward([c, d, a], ups)
        for g, r in zip(grads_of(lv), gref):
            assert pu.nrm_err(g, r) < 1e-6
    # side stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        rast, lv, ups = A
        c, _, d, a = fwd(rast, lv)
        torch.autograd.backward([c, d, a], ups)
    s.synchronize()
    assert torch.equal(c, ref[0][0])
    for g, r in zip(grads_of(lv), ref[0][1]):
        assert pu.nrm_err(g, r) < 1e-6


@pytest.mark.parametrize("what", ["means_nan", "means_inf", "scale_nan", "scale_inf", "scale_zero", "scale_negative",
                                  "rot_nan", "rot_zero", "opacity_nan"])
def test_non_finite_and_degenerate_gaussians_are_contained(what):
    """Every 7th Gaussian is malformed.  The path terminates, the image stays finite, the healthy Gaussians get
    finite gradients, and the healthy part renders exactly as if the malformed Gaussians were absent whenever the
    geometry stage rejects them (radius 0)."""
    dev = _dev()
    from scgaussian_amd import rasterizer as R
    P, W, H = 5000, 160, 96
    cam = syn.default_camera(W, H)
    sc = syn.make_scene(P, W, H, seed=3)
    idx = torch.arange(0, P, 7)
    poison = {"means_nan": (sc.means3D, slice(None), float("nan")), "means_inf": (sc.means3D, 0, float("inf")),
              "scale_nan": (sc.scales, slice(None), float("nan")), "scale_inf": (sc.scales, 1, float("inf")),
              "scale_zero": (sc.scales, slice(None), 0.0), "scale_negative": (sc.scales, slice(None), -0.05),
              "rot_nan": (sc.rotations, slice(None), float("nan")), "rot_zero": (sc.rotations, slice(None), 0.0),
              "opacity_nan": (sc.opacities, slice(None), float("nan"))}[what]
    poison[0][idx, poison[1]] = poison[2]
    st = pu.hip_settings(cam, 3, (0.1, 0.2, 0.3))
    rast = R.GaussianRasterizer(st)
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m, sh, op, s, r = params
    c, radii, d, a = rast(means3D=m, means2D=torch.zeros_like(m), shs=sh, opacities=op, scales=s, rotations=r)
    (c.sum() + d.sum() + a.sum()).backward()
    torch.cuda.synchronize()
    bad = torch.zeros(P, dtype=torch.bool, device=dev)
    bad[idx.to(dev)] = True
    for t in (c, d, a):
        assert bool(torch.isfinite(t).all())
    for p in params:
        assert bool(torch.isfinite(p.grad[~bad]).all())
    if int((radii[bad] > 0).sum()) == 0:
        keep = ~bad
        c2, radii2, d2, a2 = rast(means3D=m[keep].detach(), means2D=torch.zeros_like(m[keep]), shs=sh[keep].detach(),
                                  opacities=op[keep].detach(), scales=s[keep].detach(), rotations=r[keep].detach())
        assert torch.equal(radii[keep], radii2)
        assert torch.equal(c, c2) and torch.equal(d, d2) and torch.equal(a, a2)
    else:
        assert what in ("rot_zero", "opacity_nan", "scale_zero", "scale_negative")


def test_non_fp32_and_non_contiguous_inputs_get_gradients_of_their_own_dtype_and_layout():
    """The kernels compute in fp32 on contiguous buffers; inputs in another dtype or layout are converted on the way
    in and their gradients come back in the input's dtype and shape (as autograd requires)."""
    dev = _dev()
    from scgaussian_amd import rasterizer as R
    P, W, H = 1500, 96, 64
    cam = syn.default_camera(W, H)
    sc = syn.make_scene(P, W, H, seed=8)
    st = pu.hip_settings(cam, 3, (0.0, 0.0, 0.0))
    rast = R.GaussianRasterizer(st)
    ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]

    def run(means, shs, opac, scales, rots):
        leaves = [means, shs, opac, scales, rots]
        c, _, d, a = rast(means3D=means, means2D=torch.zeros_like(means), shs=shs, opacities=opac, scales=scales,
                          rotations=rots)
        torch.autograd.backward([c, d, a], ups)
        return [c, d, a], leaves

    base = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    out0, _ = run(*base)
    g0 = [p.grad.clone() for p in base]
    # float64 leaves
    f64 = [t.detach().double().requires_grad_(True) for t in base]
    out1, _ = run(*f64)
    for a, b in zip(out0, out1):
        assert torch.equal(a, b)
    for p, g in zip(f64, g0):
        assert p.grad.dtype == torch.float64 and p.grad.shape == p.shape
        assert pu.nrm_err(p.grad.float(), g) < 1e-6            # (the backward's atomics make the last bits run-dependent)
    # a transposed (non-contiguous) SH leaf and an expanded opacity
    sh_t = base[1].detach().transpose(1, 2).contiguous().requires_grad_(True)          # (P, 3, 16) storage
    out2, _ = run(base[0].detach().requires_grad_(True), sh_t.transpose(1, 2), base[2].detach().requires_grad_(True),
                  base[3].detach().requires_grad_(True), base[4].detach().requires_grad_(True))
    for a, b in zip(out0, out2):
        assert torch.equal(a, b)
    assert pu.nrm_err(sh_t.grad.transpose(1, 2), g0[1]) < 1e-6


def test_second_backward_over_one_forward_repeats_the_first():
    """retain_graph: a second backward over the same saved forward state gives the same gradients (lists of several hundred
    entries per tile)."""
    from scgaussian_amd import rasterizer as R
    P, W, H = 6000, 200, 120
    sc = syn.make_scene(P, W, H, seed=21, log_scale_mean=-2.6)
    cam = syn.orbit_camera(W, H, -7.0, 4.0, 7.0)
    grads = syn.make_upstream_grads(W, H, seed=4)
    dev = _dev()
    lv = {k: v.detach().to(dev).requires_grad_(True) for k, v in pu.run_oracle_inputs(sc, cam, 3, 1.0, "sh_sr").items()}
    kw = {k: v for k, v in lv.items() if k not in ("means3D", "means2D", "opacities")}
    c, r, d, a = R.GaussianRasterizer(pu.hip_settings(cam, 3, (0.3, 0.1, 0.2)))(means3D=lv["means3D"], means2D=lv["means2D"],
                                                                                  opacities=lv["opacities"], **kw)
    loss = (c * grads[0].to(dev)).sum() + (d * grads[1].to(dev)).sum() + (a * grads[2].to(dev)).sum()
    g1 = torch.autograd.grad(loss, list(lv.values()), retain_graph=True)
    g2 = torch.autograd.grad(loss, list(lv.values()))
    for x, y, k in zip(g1, g2, lv):
        assert float(x.abs().max()) > 0
        pu.assert_close(y, x, ("second backward over one forward", k))


_CXX_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import parity_utils as pu
from scgaussian_amd import synthetic as syn, rasterizer as R, _lib
assert _lib.LIB_PATH.endswith("_cxx.so"), _lib.LIB_PATH
out = {}
for i, (P, W, H, scale, deg) in enumerate([(4000, 208, 120, -4.0, 3), (9000, 256, 192, -3.0, 2), (800, 77, 45, -3.5, 0)]):
    sc = syn.make_scene(P, W, H, seed=31 + i, log_scale_mean=scale).to("cuda")
    cam = syn.orbit_camera(W, H, 6.0 - 5 * i, 2.0, 7.0)
    fs = R.forward_stages(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales,
                          rotations=sc.rotations)
    for k in ("color", "depth", "alpha", "final_T", "n_contrib", "radii"):
        out[f"{i}_{k}"] = fs[k].cpu().numpy()
    # the ONE-CALL path too: tile_blend_forward_kernel inlines the same trip under a 64-register budget (waves_per_eu 8)
    for k, v in pu.one_call_forward(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc).items():
        out[f"{i}_fused_{k}"] = v
    # ... and the backward walk (hand-written since round 4): gradients of a fixed upstream
    for k, v in pu.gradients_for_fixed_upstream(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc, W, H, seed=50 + i).items():
        out[f"{i}_grad_{k}"] = v
np.savez(sys.argv[2], **out)
"""


def test_hand_written_forward_trip_equals_the_compiler_written_one_bit_for_bit(tmp_path):
    """csrc/blend.hip's forward trip and backward walk are hand-written ISA with hard-coded registers (v44..v63);
    -DSCG_FWD_TRIP_CXX builds both from C++.  The two libraries must produce bit-identical color / depth / alpha / final_T /
    n_contrib — and the same gradients up to the order of the float atomics (the per-pixel arithmetic of the two backward
    walks is the same operations in the same order): a toolchain change that breaks the inline assembly's assumptions shows
    up here."""
    import os
    import subprocess
    import sys
    from scgaussian_amd import build as B, rasterizer as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = B.build_cxx_trip_variant()
    npz = str(tmp_path / "cxx.npz")
    env = dict(os.environ, SCG_LIB_PATH=cxx)
    res = subprocess.run([sys.executable, "-c", _CXX_CHILD, root, npz], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    ref = np.load(npz)
    for i, (P, W, H, scale, deg) in enumerate([(4000, 208, 120, -4.0, 3), (9000, 256, 192, -3.0, 2), (800, 77, 45, -3.5, 0)]):
        sc = syn.make_scene(P, W, H, seed=31 + i, log_scale_mean=scale).to(_dev())
        cam = syn.orbit_camera(W, H, 6.0 - 5 * i, 2.0, 7.0)
        fs = R.forward_stages(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc.means3D, sc.opacities, shs=sc.shs,
                              scales=sc.scales, rotations=sc.rotations)
        for k in ("color", "depth", "alpha", "final_T", "n_contrib", "radii"):
            assert np.array_equal(fs[k].cpu().numpy(), ref[f"{i}_{k}"]), (i, k)
        assert int(fs["n_contrib"].max()) > 20
        # the fused sort + blend kernel of the one-call path (the instantiation the bench runs) against its compiler-written twin
        for k, v in pu.one_call_forward(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc).items():
            assert np.array_equal(v, ref[f"{i}_fused_{k}"]), (i, "one-call", k)
            if k in ("color", "depth", "alpha", "final_T", "n_contrib"):
                assert np.array_equal(v.reshape(ref[f"{i}_{k}"].shape), ref[f"{i}_{k}"]), (i, "one-call vs staged", k)
        for k, v in pu.gradients_for_fixed_upstream(pu.hip_settings(cam, deg, (0.2, 0.4, 0.1)), sc, W, H, seed=50 + i).items():
            assert float(np.abs(v).max()) > 0, (i, k)
            # (two runs of ONE library differ by up to 2.5e-6 of the tensor maximum here — the order of the float atomics,
            #  tools/probes/atomics_noise.py: rotations of scene 1 — so the bar is 2e-5, a fifth of the suite's)
            pu.assert_close(v, ref[f"{i}_grad_{k}"], ("hand-written vs compiler-written backward walk", i, k), rel=2e-5)


def _fused_vs_staged(sc, cam, deg, bg):
    """One-call forward (the geometry kernel builds the slice histograms, the blend sorts its own tiles) against the staged
    calls (histogram kernel, sort kernel + blend kernel) and against the one-call path with the sort or the histogram kept
    apart: sorted lists, ranges and every image bit for bit."""
    from scgaussian_amd import rasterizer as R
    dev = _dev()
    scd = sc.to(dev)
    st = pu.hip_settings(cam, deg, bg)
    R._SPEC_STATE.clear()
    exact = R.forward_stages(st, scd.means3D, scd.opacities, shs=scd.shs, scales=scd.scales, rotations=scd.rotations)
    Rn = exact["num_rendered"]
    res = {}
    old = R.FUSED_SORT, R.FUSED_HIST
    try:
        # (sort inside the blend, histogram inside the geometry kernel), the sort kept apart, the histogram kept apart
        for fused in ((True, True), (False, True), (True, False)):
            R.FUSED_SORT, R.FUSED_HIST = fused
            out = R.forward_fused(st, scd.means3D, scd.opacities, scd.shs, None, scd.scales, scd.rotations, None, True)
            assert out is not None
            torch.cuda.synchronize()
            c, radii, d, a, state = out
            ws, plan = state["ws"], state["plan"]
            res[fused] = dict(color=c, depth=d, alpha=a, radii=radii,
                              point_list=ws[plan.point_list: plan.point_list + 4 * Rn].view(torch.int32).clone(),
                              final_T=ws[plan.final_T: plan.final_T + 4 * exact["final_T"].numel()].view(torch.float32).clone(),
                              n_contrib=ws[plan.n_contrib: plan.n_contrib + 4 * exact["n_contrib"].numel()].view(torch.int32).clone())
        # a render that will not be differentiated (SCG_FORWARD_NO_BACKWARD_STATE: final_T / n_contrib are not written)
        out = R.forward_fused(st, scd.means3D, scd.opacities, scd.shs, None, scd.scales, scd.rotations, None, False)
        torch.cuda.synchronize()
        for k, v in zip(("color", "radii", "depth", "alpha"), out[:4]):
            assert torch.equal(v, exact[k]), ("render without backward state", k)
    finally:
        R.FUSED_SORT, R.FUSED_HIST = old
    for fused in res:
        r = res[fused]
        assert torch.equal(r["point_list"], exact["point_list"]), fused
        for k in ("color", "depth", "alpha", "radii"):
            assert torch.equal(r[k], exact[k]), (fused, k)
        assert torch.equal(r["final_T"].view_as(exact["final_T"]), exact["final_T"]), fused
        assert torch.equal(r["n_contrib"].view_as(exact["n_contrib"]), exact["n_contrib"]), fused
    counts = pu.as_u32(exact["ranges"])[:, 1].astype(np.int64) - pu.as_u32(exact["ranges"])[:, 0]
    return counts


@pytest.mark.parametrize("P,W,H", [(1, 33, 17), (63, 40, 40), (255, 100, 30), (257, 64, 64), (1000, 300, 20), (70000, 199, 150),
                                   (500, 16, 16), (500, 9, 7), (3000, 130, 8), (120000, 48, 48)])
def test_one_call_path_equals_the_staged_calls_on_odd_shapes(P, W, H):
    """Gaussian counts around the 256-Gaussian blocks the histogramming geometry kernel cuts its slices on (most slices empty,
    a last block that is not full), image shapes with a partial tile on both axes, one wide and flat, a single tile, fewer tiles than
    the eight bands of the scatter, 256 slices on nine tiles: scg_forward against the
    staged calls bit for bit (lists, ranges, images, state)."""
    sc = syn.make_scene(P, W, H, seed=P + W, log_scale_mean=-3.0)
    counts = _fused_vs_staged(sc, syn.default_camera(W, H), 2, (0.3, 0.2, 0.1))
    assert len(counts) == ((W + 15) // 16) * ((H + 15) // 16)


def test_forward_blend_that_sorts_its_own_tiles_equals_sort_kernel_plus_blend_kernel():
    """scg_forward's fused sort + blend (tile_blend_forward_kernel) on scenes whose lists cover every case: ordinary lists,
    lists beyond the fused kernel's 1 536 entries (sorted by the rare-size kernel first: 16-wave LDS sort, global bucket
    sort), heavily tied depths (radix fallback inside the fused kernel), empty tiles and a ragged image border."""
    W, H = 64, 48
    cam = syn.default_camera(W, H)
    seen = []
    # (the last two: DENSE frames — thousands of entries in every tile: the blend's 3 584-entry variant sorts them with eight waves, round 5)
    for P, spread, ties in ((900, 0.02, True), (1500, 0.02, False), (1500, 0.02, True), (6000, 0.02, False), (20000, 0.5, True),
                            (30000, 0.02, False), (40000, 0.5, False), (40000, 0.5, True)):
        g = torch.Generator().manual_seed(7 * P + int(ties))
        xy = (torch.rand(P, 2, generator=g) - 0.5) * spread
        z = (torch.randint(0, 37, (P,), generator=g).float() * 0.25 + 3.0) if ties else (torch.rand(P, generator=g) * 9.0 + 3.0)
        means = torch.cat([xy * z[:, None], z[:, None]], 1)
        sc = syn.Scene(means, torch.full((P, 3), 0.004), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                       torch.full((P, 1), 0.02), torch.rand(P, 16, 3, generator=g) * 0.1)
        seen += [int(c) for c in _fused_vs_staged(sc, cam, 0, (0.1, 0.0, 0.2)) if c > 0]
    ls = np.array(seen)
    assert ((ls > 1) & (ls <= 1536)).any() and ((ls > 1536) & (ls <= 8192)).any() and (ls > 8192).any(), sorted(set(seen))[-8:]
    assert ((ls > 2048) & (ls <= 3584)).any(), sorted(set(seen))
    # ordinary scenes: odd image sizes (ragged last tile row / column), several hundred entries per tile
    for i, (P, W, H, scale) in enumerate([(9000, 250, 187, -3.0), (4000, 208, 120, -4.0), (30000, 333, 201, -3.3)]):
        sc = syn.make_scene(P, W, H, seed=40 + i, log_scale_mean=scale)
        counts = _fused_vs_staged(sc, syn.orbit_camera(W, H, 4.0 * i, -2.0, 7.0), 3, (0.0, 0.2, 0.1))
        assert counts.max() > 100


def test_skipped_rare_sort_launch_and_the_forward_blends_fallback_for_a_long_list_that_shows_up_after_all():
    """SCG_FORWARD_SKIP_RARE_SORT (ABI 8): while the previous render of a camera found no list beyond the forward blend's own
    LDS sort (ScgFrame.long_lists_out = 0) the binding does not launch the rare-size sort kernel.  The promise is about speed
    only: when the same camera then sees long lists after all (same Gaussian count, different positions), the tile's own
    workgroup sorts them through global scratch — outputs bit-identical to the staged calls — and the word tells the binding
    to launch the kernel again at the next render."""
    from scgaussian_amd import rasterizer as R
    dev = _dev()
    W, H, P = 320, 208, 40000              # 260 tiles: the average list stays short (the fused sort + blend kernel runs) even
    cam = syn.default_camera(W, H)         # when a cluster fills a few tiles with thousands of entries
    st = pu.hip_settings(cam, 1, (0.1, 0.0, 0.2))
    g = torch.Generator().manual_seed(3)
    from scgaussian_amd import _lib as _L
    lib = _L.load()

    def scene(spread, tied):
        xy = (torch.rand(P, 2, generator=g) - 0.5) * spread
        z = (torch.randint(0, 23, (P,), generator=g).float() * 0.3 + 3.0) if tied else (torch.rand(P, generator=g) * 9.0 + 3.0)
        means = torch.cat([xy * z[:, None], z[:, None]], 1)
        return syn.Scene(means, torch.full((P, 3), 0.004), torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1),
                         torch.full((P, 1), 0.02), torch.rand(P, 16, 3, generator=g) * 0.1).to(dev)

    def one_call(sc):
        out = R.forward_fused(st, sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None, True)
        assert out is not None
        torch.cuda.synchronize()
        state = out[4]
        ws, plan, Rn = state["ws"], state["plan"], state["num_rendered"]
        return out, ws[plan.point_list: plan.point_list + 4 * Rn].view(torch.int32).clone(), state["frame"]

    def check(sc, out, pl):
        saved = dict(R._spec_state(dev).hint), dict(R._spec_state(dev).cam_hint)
        exact = R.forward_stages(st, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        R._spec_state(dev).hint, R._spec_state(dev).cam_hint = saved          # (the staged call is the checker, not a render)
        assert torch.equal(pl, exact["point_list"])
        for k, v in zip(("color", "radii", "depth", "alpha"), out[:4]):
            assert torch.equal(v, exact[k]), k
        cnt = pu.as_u32(exact["ranges"])
        return int((cnt[:, 1].astype(np.int64) - cnt[:, 0]).max())

    old = R.SKIP_IDLE_RARE_SORT
    R.SKIP_IDLE_RARE_SORT = True
    R._SPEC_STATE.clear()
    R._FRAME_CACHE.clear()
    R._CAM_HINTS.clear()
    try:
        wide = scene(2.5, False)                                # lists of a few hundred entries at most
        R.forward_stages(st, wide.means3D, wide.opacities, shs=wide.shs, scales=wide.scales, rotations=wide.rotations)
        out, pl, fr = one_call(wide)                            # first one-call render of the frame: word unknown (-1 -> launch)
        assert check(wide, out, pl) <= 1536 and int(fr.long_np[0]) == 0 and R._rare_options(fr.long_np) == 8
        out, pl, fr2 = one_call(wide)                           # second: the launch is skipped (word == 0)
        assert fr2 is fr and check(wide, out, pl) <= 1536 and int(fr.long_np[0]) == 0
        # another background tensor (reference train.py:141 random_background): another frame, the camera's hints carry over
        st_bg = st._replace(bg=torch.tensor([0.5, 0.5, 0.5], device=wide.means3D.device))
        fr_bg = R._frame_for(st_bg, P, 16, wide.means3D.device, forward=True)
        assert fr_bg is not fr and fr_bg.hints is fr.hints and R._rare_options(fr_bg.long_np) == 8
        for tied in (False, True):                              # ... and now the promise is wrong: lists of thousands of entries
            fr.long_np[0] = 0
            dense = scene(0.05, tied)
            # room for the dense scene's instances (the capacity is the caller's business, not what is tested here)
            sp = R._spec_state(dev)
            sp.cam_hint.clear()
            sp.hint[(P, W, H)] = 200_000
            assert lib.scg_forward_sorts_in_blend(200_000, W, H, 0) == 1
            out, pl, fr3 = one_call(dense)
            assert fr3 is fr
            longest = check(dense, out, pl)
            assert longest > 16384, longest
            # the next render of this camera knows: lists beyond the blend's own sort, and very long ones among them
            assert int(fr.long_np[0]) > 0 and int(fr.long_np[1]) > 0
            assert R._rare_options(fr.long_np) == (16 | 32)     # -> split by depth + 8-wave work-list sort
            out, pl, _ = one_call(dense)
            assert check(dense, out, pl) == longest
            # ... and with a bound that is too small for this scene while the split path is on: the lists are clipped (the split
            # and the 8-wave sort work on clipped ranges, nothing is written out of bounds), the binding sees num_rendered
            # and renders again with room for it
            assert R._rare_options(fr.long_np) == (16 | 32)
            sp.cam_hint[(W, H, R._camera_key(st.viewmatrix))] = (30_000, 30_000, P)
            out, pl, _ = one_call(dense)
            assert out[4]["cap"] >= out[4]["num_rendered"] > 30_000
            assert check(dense, out, pl) == longest
    finally:
        R.SKIP_IDLE_RARE_SORT = old
