"""BASELINE configs at their real sizes, HIP path against the CPU oracle.

* backward at S2 (cfg2: 200 k Gaussians @ 1008x756), S4 (cfg5 per-view shape: 1 M @ 960x540) and S3 (the north-star
  point: 500 k @ 1920x1080, ~713 instances per tile) on a TILE SUBSET: the
  upstream gradients are zeroed outside every k-th tile, the oracle blends (and differentiates) only those tiles
  (orc.blend(tile_stride=k)), and all gradients are compared — 32-bit offset arithmetic, atomics under the
  contention of a full-size launch and long per-tile lists are all exercised at the size the bench runs;
* cfg3 (DTU-style): several views rendered with the HIP path, the masked alpha term of train.py:167-168 in the loss,
  the rendered depths fed to the structure-consistency check (utils/geo_check.py:33-88) ON THE DEVICE and compared
  with the float64 numpy oracle of it;
* autograd hygiene: in-place modification between forward and backward raises; two forwards in flight on one device (two threads, two streams)
  raise instead of racing."""
import math
import threading

import numpy as np
import pytest
import torch

import parity_utils as pu
from oracle import geo_check_oracle as geo_orc
from oracle import torch_rasterizer as orc
from scgaussian_amd import geo_check as gc
from scgaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _tile_mask(W, H, sel_tiles):
    """Pixel mask (H, W) of the selected tiles (indices into the row-major tile grid)."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    sel = np.zeros(gx * gy, dtype=bool)
    sel[np.asarray(sorted(sel_tiles), dtype=np.int64)] = True
    sel = sel.reshape(gy, gx)
    m = np.repeat(np.repeat(sel, 16, axis=0), 16, axis=1)[:H, :W]
    return torch.from_numpy(m), int(sel.sum())


def _one_call_lists(sc, cam, deg, bg):
    """num_rendered, point_list, ranges and radii as the ONE-CALL path (scg_forward: the path bench.py times) leaves them.
    The capacity of the shape is known from the render before (run_hip): forward_fused applies."""
    o = pu.one_call_forward(pu.hip_settings(cam, deg, bg), sc.to(torch.device("cuda")))
    return int(o["num_rendered"]), o["point_list"], o["ranges"], torch.from_numpy(o["radii"])


# (workload, tile stride, clustered variant or None).  The clustered case: 30 % of S2's Gaussians around one screen point —
# lists of > 8 192 entries (rare-size sort kernels, long walks) beside nearly empty tiles; its three longest tiles are added to
# the stride subset.
@pytest.mark.parametrize("name,stride,clustered", [("S2", 29, None), ("S4", 19, None), ("S3", 61, None),
                                                   ("S2", 29, "clustered30")])
def test_full_size_backward_on_a_tile_subset_matches_the_oracle(name, stride, clustered):
    w = syn.WORKLOADS[name]
    P, W, H, deg = w["P"], w["width"], w["height"], 3
    sc = syn.make_scene(P, W, H, seed=0) if clustered is None else \
        syn.make_clustered_scene(P, W, H, *syn.CLUSTERED[clustered], seed=0)
    cam = syn.default_camera(W, H)
    bg = (0.2, 0.1, 0.3)
    # oracle: full preprocess + binning
    st = pu.oracle_settings(cam, deg, bg)
    leaves = {k: v.clone().requires_grad_(True) for k, v in
              dict(means3D=sc.means3D, means2D=torch.zeros(P, 3), opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
                   rotations=sc.rotations).items()}
    pre = orc.preprocess(leaves["means3D"], leaves["means2D"], leaves["opacities"], st, shs=leaves["shs"],
                         scales=leaves["scales"], rotations=leaves["rotations"])
    binning = orc.bin_and_sort(pre, W, H)
    counts = binning["ranges"][:, 1].astype(np.int64) - binning["ranges"][:, 0]
    sel = set(range(0, len(counts), stride))
    if clustered is not None:
        assert counts.max() > 8192, counts.max()
        sel |= set(int(t) for t in np.argsort(counts)[-3:])
    mask, n_sel = _tile_mask(W, H, sel)
    grads = tuple(g * mask for g in syn.make_upstream_grads(W, H, seed=3))
    # ... blend + autograd on the selected tiles only
    c, d, a, fT, nC = orc.blend(pre, binning, st, tiles=sel)
    torch.autograd.backward([c, d, a], list(grads))
    # HIP path: the whole image, the same (masked) upstream gradients
    h = pu.run_hip(sc, cam, deg, bg, grads=grads)
    assert torch.equal(h["radii"].cpu(), pre["radii"])
    # north_star "bit-exact tile assignment and sort indices" AT THE SIZE THE BENCH RUNS, on the path it times: the one-call
    # forward's instance count, sorted id list and per-tile ranges against the oracle's stable sort of (tile << 32 | depth, id)
    Rn, point_list, ranges, radii1 = _one_call_lists(sc, cam, deg, bg)
    assert Rn == binning["num_rendered"], (Rn, binning["num_rendered"])
    assert torch.equal(radii1.cpu(), pre["radii"])
    assert np.array_equal(ranges, binning["ranges"])
    assert np.array_equal(point_list, binning["point_list"])
    m3 = mask[None]
    for k, ref_img in (("color", c), ("depth", d), ("alpha", a)):
        got = h[k].cpu() * m3
        pu.assert_close(got, ref_img.detach() * m3, (name, clustered, "subset image", k))
    n_inst = int(counts[sorted(sel)].sum())
    assert n_inst > 20_000, n_inst
    for k, leaf in leaves.items():
        assert leaf.grad is not None and float(leaf.grad.abs().max()) > 0, k
        pu.assert_close(h["grads"][k], leaf.grad, (name, clustered, f"{n_sel} tiles / {n_inst} instances", k))


def _dtu_like_views(W, H, n):
    return [syn.orbit_camera(W, H, yaw, pitch, 7.0) for yaw, pitch in
            [(0.0, 0.0), (7.0, 2.0), (-6.0, -3.0), (12.0, -1.0), (-11.0, 4.0)][:n]]


def test_cfg3_depth_render_masked_alpha_loss_and_geo_check():
    """BASELINE cfg3 as one workload at a DTU-like size."""
    from scgaussian_amd.rasterizer import GaussianRasterizer
    dev = torch.device("cuda")
    P, W, H, deg = 30_000, 400, 300, 3
    g = torch.Generator().manual_seed(11)
    # an object: Gaussians on a bumpy surface patch in front of the cameras (opaque enough to give a depth map)
    u, v = torch.rand(P, generator=g) * 2 - 1, torch.rand(P, generator=g) * 2 - 1
    zs = 7.0 + 0.6 * torch.sin(2.5 * u) * torch.cos(2.0 * v)
    means = torch.stack([2.2 * u, 1.6 * v, zs], 1).contiguous()
    base = syn.make_scene(P, W, H, seed=12, log_scale_mean=-3.3, log_scale_std=0.3)
    sc = syn.Scene(means, base.scales, base.rotations, torch.sigmoid(torch.randn(P, 1, generator=g) + 2.5), base.shs)
    views = _dtu_like_views(W, H, 5)
    bg = (0.0, 0.0, 0.0)

    leaves_cpu = dict(means3D=sc.means3D, means2D=torch.zeros(P, 3), opacities=sc.opacities, shs=sc.shs, scales=sc.scales,
                      rotations=sc.rotations)
    depths, alphas = [], []
    for vi, cam in enumerate(views):
        st = pu.hip_settings(cam, deg, bg)
        lv = {k: t.detach().to(dev).requires_grad_(True) for k, t in leaves_cpu.items()}
        c, radii, d, a = GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], opacities=lv["opacities"],
                                                shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"])
        depths.append(d.detach()[0])
        alphas.append(a.detach()[0])
        if vi == 0:
            # train.py:150-168: background mask from a dark ground-truth image, alpha pushed to 0 there
            gt = (c.detach() * (a.detach() > 0.6)).clamp(0, 1)
            bg_mask = gt.max(0, keepdim=True).values < 30 / 255
            assert 0.02 < float(bg_mask.float().mean()) < 0.98
            target = gt * 0.8 + 0.1             # a smooth image term (|x| has a kink exactly where render == target)
            loss = ((c - target) ** 2).mean() + a[bg_mask].mean()
            loss.backward()
            # oracle, same loss (mask and target are data)
            ol = {k: t.clone().requires_grad_(True) for k, t in leaves_cpu.items()}
            oc, orad, od, oa = orc.rasterize(ol["means3D"], ol["means2D"], ol["opacities"], pu.oracle_settings(cam, deg, bg),
                                             shs=ol["shs"], scales=ol["scales"], rotations=ol["rotations"])
            oloss = ((oc - target.cpu()) ** 2).mean() + oa[bg_mask.cpu()].mean()
            oloss.backward()
            assert torch.equal(radii.cpu(), orad)
            pu.assert_close(d, od.detach(), ("cfg3", "depth"))
            pu.assert_close(a, oa.detach(), ("cfg3", "alpha"))
            assert abs(float(loss) - float(oloss)) <= 1e-5 * abs(float(oloss))
            for k in ol:
                pu.assert_close(lv[k].grad, ol[k].grad, ("cfg3", "grad", k))
    torch.cuda.synchronize()

    # structure consistency on the RENDERED depths, on the device, vs the float64 numpy oracle on the same depths
    def intr(cam):
        fx, fy = W / (2 * math.tan(cam.FoVx / 2)), H / (2 * math.tan(cam.FoVy / 2))
        return torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1.0]], dtype=torch.float64)
    intrs = torch.stack([intr(c) for c in views])
    w2cs = torch.stack([c.world_view_transform.t().double() for c in views])
    dstack = torch.stack(depths)
    # un-normalised expected depth -> a depth map where the surface is opaque; holes stay 0 (the check must cope)
    astack = torch.stack(alphas)
    dmaps = torch.where(astack > 0.9, dstack / astack.clamp_min(1e-6), torch.zeros_like(dstack))
    gd, gm = gc.geocheck(intrs.to(dev), w2cs.to(dev), dmaps, dist_thresh=1.0, depth_thresh=0.01, view_thresh=2, num_src=4)
    with np.errstate(all="ignore"):
        od_, om_ = geo_orc.geocheck(intrs.numpy(), w2cs.numpy(), dmaps.cpu().numpy(), dist_thresh=1.0, depth_thresh=0.01,
                                    view_thresh=2, num_src=4)
    gm_c, gd_c = gm.cpu().numpy(), gd.cpu().numpy()
    assert (gm_c != om_).mean() < 0.003                       # threshold ties may flip a pixel
    same = gm_c == om_
    assert np.allclose(gd_c[same], od_[same], rtol=1e-4, atol=1e-4)
    assert 0.2 < om_.mean() < 1.0                             # the check keeps a real share of the surface


def test_in_place_update_between_forward_and_backward_raises():
    from scgaussian_amd.rasterizer import GaussianRasterizer
    P, W, H = 500, 64, 48
    sc = syn.make_scene(P, W, H, seed=2, log_scale_mean=-3.0).to("cuda")
    cam = syn.default_camera(W, H)
    st = pu.hip_settings(cam, 3, (0.0, 0.0, 0.0))
    means = sc.means3D.clone().requires_grad_(True)
    scales = sc.scales.clone().requires_grad_(True)
    c, _, d, a = GaussianRasterizer(st)(means3D=means, means2D=torch.zeros_like(means, requires_grad=True),
                                        opacities=sc.opacities, shs=sc.shs, scales=scales, rotations=sc.rotations)
    with torch.no_grad():
        scales.mul_(1.01)                       # e.g. an optimizer step between forward and backward
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        c.sum().backward()


def test_two_forwards_in_flight_on_one_device_from_two_threads():
    """The operator is re-entrant per device and stream (a second forward in flight used to raise): two
    threads, each on its own stream with its own scene, render and differentiate concurrently; every result equals the one the
    same scene gives alone."""
    from scgaussian_amd import rasterizer as R
    W, H = 160, 96
    cams = [syn.default_camera(W, H), syn.orbit_camera(W, H, 9.0, -3.0, 7.0)]
    scenes = [syn.make_scene(P, W, H, seed=30 + i, log_scale_mean=-3.3).to("cuda") for i, P in enumerate((2500, 4100))]
    ups = [tuple(t.to("cuda") for t in syn.make_upstream_grads(W, H, seed=5 + i)) for i in range(2)]

    def run(i, n, out, stream=None):
        sc, st = scenes[i], pu.hip_settings(cams[i], 3, (0.1, 0.2, 0.3))
        rast = R.GaussianRasterizer(st)
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            for _ in range(n):
                leaves = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
                m, sh, op, s_, r_ = leaves
                c, radii, d, a = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=op, shs=sh,
                                      scales=s_, rotations=r_)
                torch.autograd.backward([c, d, a], list(ups[i]))
                out.append((c.detach().clone(), radii.clone(), [t.grad.clone() for t in leaves]))
            if stream is not None:
                stream.synchronize()

    import contextlib
    alone = [[], []]
    for i in range(2):
        run(i, 2, alone[i])
    torch.cuda.synchronize()
    both = [[], []]
    errs = []

    def worker(i):
        try:
            run(i, 12, both[i], torch.cuda.Stream())
        except Exception as e:                              # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        c0, r0, g0 = alone[i][-1]
        for c, r, g in both[i]:
            assert torch.equal(c, c0) and torch.equal(r, r0)
            for a_, b_ in zip(g, g0):
                assert pu.nrm_err(a_, b_) < 1e-4                # (float atomics: the order of the sums differs run to run)


def test_tile_cost_hint_reorders_the_launch_and_changes_no_result():
    """ScgFrame.tile_cost_in / _out (ABI 5): the blend forward records what every tile cost, the next render of the same
    camera launches the expensive tiles first.  Every forward output must be bit-identical with and without the hint,
    the recorded cost is the busiest quadrant's number of blended list entries (<= the tile's list length), and the
    launch order stays a permutation of the tiles."""
    from scgaussian_amd import rasterizer as R
    P, W, H = 6000, 200, 136
    sc = syn.make_scene(P, W, H, seed=11, log_scale_mean=-3.2).to("cuda")
    st = pu.hip_settings(syn.default_camera(W, H), 3, (0.1, 0.2, 0.3))
    args = dict(shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    runs = []
    for i in range(3):
        fs = R.forward_stages(st, sc.means3D, sc.opacities, **args)
        fr = fs["frame"]
        torch.cuda.synchronize()
        n_tiles = fr.n_tiles
        slots = (n_tiles + 7) // 8 * 8
        words = fs["arenas"][1].view(1, (R._lib.load().scg_ranges_words(W, H),), torch.int32)
        assert words.numel() == 2 * n_tiles + 5 * slots
        order, order_q = words[2 * n_tiles: 2 * n_tiles + slots], words[2 * n_tiles + slots:]
        # the blend backward's (tile, quadrant) order behind it: a permutation of the 4 n_tiles quadrants, every band's in its band
        real_q = order_q[order_q < 4 * n_tiles]
        assert real_q.numel() == 4 * n_tiles and torch.equal(torch.sort(real_q).values,
                                                             torch.arange(4 * n_tiles, device=order_q.device, dtype=order_q.dtype))
        per_q = slots // 8
        bq = order_q.view(8, 4 * per_q)
        for b in range(8):
            v = bq[b][bq[b] < 4 * n_tiles] // 4
            assert bool(((v >= b * per_q) & (v < (b + 1) * per_q)).all())
        runs.append(dict(color=fs["color"].clone(), depth=fs["depth"].clone(), alpha=fs["alpha"].clone(),
                         n_contrib=fs["n_contrib"].clone(), point_list=fs["point_list"].clone(),
                         ranges=fs["ranges"].clone(), order=order.clone(), hinted=bool(fr.c.tile_cost_in),
                         cost=fr.hints.cost[fr.hints.cur].clone()))
    assert [r["hinted"] for r in runs] == [False, True, True]
    for r in runs[1:]:
        for k in ("color", "depth", "alpha", "n_contrib", "point_list", "ranges"):
            assert torch.equal(r[k], runs[0][k]), k
        assert torch.equal(r["cost"], runs[0]["cost"])               # the cost is a property of the frame, not of the order
    lens = (runs[0]["ranges"][:, 1] - runs[0]["ranges"][:, 0])
    cost = runs[0]["cost"]
    assert int(cost.max()) > 0 and bool((cost <= lens).all()) and bool((cost[lens == 0] == 0).all())
    n_tiles = lens.numel()
    for r in runs:
        o = r["order"]
        real = o[o < n_tiles]
        assert real.numel() == n_tiles and torch.equal(torch.sort(real).values, torch.arange(n_tiles, device=o.device, dtype=o.dtype))
    # with the hint the order follows the cost classes: inside an XCD band costs do not increase by more than a class step
    per = (n_tiles + 7) // 8
    o = runs[1]["order"][: 8 * per].view(8, per)
    for b in range(8):
        t = o[b][o[b] < n_tiles].long()
        c = cost[t].float()
        c = c[c >= 32]                                               # classes start at 32 entries
        if c.numel() > 1:
            assert bool((c[1:] <= c[:-1] * 1.07 + 1).all())


@pytest.mark.parametrize("switch", ["EVENTLESS_WAIT", "BWD_ORDER_HINT", "TILE_COST_HINT", "SPLIT_LONG_LISTS", "RARE_8WAVE",
                                    "SKIP_IDLE_RARE_SORT"])
def test_every_switch_of_the_binding_keeps_the_results(switch):
    """The binding's module switches choose between two ways to the same numbers (an event or the armed words behind
    num_rendered, launch orders from the previous render of the camera or the plain ones, how the rare long lists are sorted):
    three renders + backwards of one camera through the public operator with the switch off equal the ones with it on,
    images bit for bit, gradients up to the order of the float atomics."""
    from scgaussian_amd import rasterizer as R
    P, W, H = 9000, 256, 176
    sc = syn.make_scene(P, W, H, seed=77, log_scale_mean=-2.9).to("cuda")
    st = pu.hip_settings(syn.orbit_camera(W, H, 25.0, 4.0, 6.0), 3, (0.3, 0.1, 0.2))
    up = tuple(t.to("cuda") for t in syn.make_upstream_grads(W, H, seed=9))

    def three_steps():
        R._SPEC_STATE.clear()
        R._CAM_HINTS.clear()
        rast = R.GaussianRasterizer(st)
        res = None
        for _ in range(3):                                   # the third render runs on the hints of the second
            leaves = [t.detach().clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
            m, sh, op, s_, r_ = leaves
            c, radii, d, a = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=op, shs=sh,
                                  scales=s_, rotations=r_)
            torch.autograd.backward([c, d, a], list(up))
            res = ([t.detach().clone() for t in (c, radii, d, a)], [t.grad.clone() for t in leaves])
        torch.cuda.synchronize()
        return res

    old = getattr(R, switch)
    try:
        setattr(R, switch, True)
        on = three_steps()
        setattr(R, switch, False)
        off = three_steps()
    finally:
        setattr(R, switch, old)
        R._SPEC_STATE.clear()
        R._CAM_HINTS.clear()
    for a_, b_ in zip(on[0], off[0]):
        assert torch.equal(a_, b_), switch
    for a_, b_ in zip(on[1], off[1]):
        assert float(a_.abs().max()) > 0
        assert pu.nrm_err(a_, b_) < 1e-4, switch
