"""The model path (ABI 10: scg_forward_model / scg_backward_model / scg_model_activate) against the oracle COMPOSED WITH THE
REFERENCE'S ACTIVATIONS (scene/gaussian_model.py:105-152, pinned by tests/golden/ref_model.npz).

How parity is established (DESIGN.md "Model path"):
  1. the kernels' activations (scg_model_activate: the device functions both geometry kernels use) equal the reference
     getters' values to fp32 rounding (<= 2e-6 relative; exp / sigmoid / normalize of two libraries);
  2. fed with exactly those activated values, the CPU oracle and the model path agree like the operator does: radii,
     num_rendered, the sorted id lists and the tile ranges BIT FOR BIT, images at the suite's bar;
  3. every raw-parameter gradient equals torch.autograd through (reference getter -> oracle) at the suite's bar, the getter's
     VALUE replaced by the kernel's (so both sides differentiate at the same point) and its derivative kept.
"""
import math
import os

import numpy as np
import pytest
import torch

import parity_utils as pu
from oracle import torch_rasterizer as orc
from scgaussian_amd import model_path as mp
from scgaussian_amd import ply_io
from scgaussian_amd import rasterizer as R
from scgaussian_amd import render as rmod
from scgaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda"


def _model(P, W, H, ray_fraction, seed=0):
    sc = syn.make_scene(P, W, H, seed=seed)
    return sc, syn.make_raw_model(sc, ray_fraction=ray_fraction, seed=seed + 3)


def _on_device(model, grad=False):
    m = model.to(DEV)
    if grad:
        m.requires_grad_()
    return m


def _oracle_side(model_cpu, act):
    """Raw CPU leaves, the reference getters on them (autograd), the getters' VALUES replaced by the kernel's activated values
    `act` = (xyz, opacity, scaling, rotation) so that the oracle runs at the very point the kernels run at."""
    leaves = ply_io.RayBoundModel(**{k: (v.clone().requires_grad_(k not in ("rayo", "rayd")) if isinstance(v, torch.Tensor) else v)
                                     for k, v in model_cpu.__dict__.items()})
    xyz, opa, sca, rot = (a.cpu() for a in act)

    def at(getter, value):
        return getter + (value - getter).detach()
    inputs = dict(means3D=at(leaves.get_xyz, xyz), opacities=at(leaves.get_opacity, opa), scales=at(leaves.get_scaling, sca),
                  rotations=at(leaves.get_rotation, rot), shs=leaves.get_features)
    return leaves, inputs


def test_activations_equal_the_reference_getters():
    """scg_model_activate vs the reference GaussianModel's own getters (values produced by the reference's code: ref_model.npz)."""
    ref = np.load(os.path.join(HERE, "golden", "ref_model.npz"))
    kw = {k: torch.from_numpy(ref["raw_" + k]).to(DEV) for k in
          ("features_dc", "features_rest", "opacity", "scaling", "rotation", "zval", "rayo", "rayd", "bg_xyz", "bg_features_dc",
           "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation")}
    m = ply_io.RayBoundModel(**kw)
    for with_bg in (True, False):
        t = mp.tensors_of(m if with_bg else ply_io.RayBoundModel(**{k: (v[:0] if k.startswith("bg_") else v) for k, v in kw.items()}))
        assert mp.supported(t)
        xyz, opa, sca, rot = mp.activate(**t)
        n = None if with_bg else t["zval"].shape[0]
        for got, name in ((xyz, "get_xyz"), (opa, "get_opacity"), (sca, "get_scaling"), (rot, "get_rotation")):
            want = ref["getter_" + name][:n]
            assert got.shape == want.shape, name
            assert np.allclose(got.cpu().numpy(), want, rtol=2e-6, atol=1e-7), (name, np.abs(got.cpu().numpy() - want).max())
    # xyz = rayo + rayd * zval is two correctly rounded operations on both sides: bit for bit
    nr = kw["zval"].shape[0]
    want = (torch.from_numpy(ref["raw_rayo"]) + torch.from_numpy(ref["raw_rayd"]) * torch.from_numpy(ref["raw_zval"])).numpy()
    assert np.array_equal(mp.activate(**mp.tensors_of(m))[0][:nr].cpu().numpy(), want)


@pytest.mark.parametrize("deg,ray_fraction", [(3, 0.6), (0, 0.6), (1, 1.0), (2, 0.0), (3, 0.37)])
def test_model_forward_integers_bit_exact_and_images_against_the_oracle(deg, ray_fraction):
    P, W, H = 30_000, 400, 304
    sc, model = _model(P, W, H, ray_fraction)
    cam = syn.orbit_camera(W, H, 5.0, -3.0, 7.0)
    bg = (0.3, 0.1, 0.2)
    md = _on_device(model)
    t = mp.tensors_of(md)
    st = pu.hip_settings(cam, deg, bg)
    assert mp.supported(t, st)
    act = mp.activate(**t)
    # oracle at the kernels' activated values
    shs = model.get_features
    o = orc.rasterize(act[0].cpu(), torch.zeros(P, 3), act[1].cpu(), pu.oracle_settings(cam, deg, bg), shs=shs,
                      scales=act[2].cpu(), rotations=act[3].cpu(), return_aux=True)
    oc, orad, od, oa, aux = o
    # the one-call model forward (twice: the first call establishes the capacity by its retry)
    for _ in range(2):
        out = R.forward_fused(st, None, None, None, None, None, None, None, True, model=mp._ModelArgs(t))
    assert out is not None
    torch.cuda.synchronize()
    c, radii, d, a, state = out
    ws, plan, Rn = state["ws"], state["plan"], int(state["num_rendered"])
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def words(off, n, dtype):
        return ws[off: off + 4 * n].view(dtype).cpu().numpy()
    assert torch.equal(radii.cpu(), orad)
    binning = aux["binning"]
    assert Rn == binning["num_rendered"]
    assert np.array_equal(words(plan.ranges, 2 * n_tiles, torch.int32).view(np.uint32).reshape(n_tiles, 2), binning["ranges"])
    assert np.array_equal(words(plan.point_list, Rn, torch.int32).view(np.uint32), binning["point_list"])
    flips = pu.threshold_flips(words(plan.n_contrib, H * W, torch.int32).reshape(H, W), aux["n_contrib"])
    assert float(flips.float().mean()) < 1e-4
    for got, want, name in ((c, oc, "color"), (d, od, "depth"), (a, oa, "alpha")):
        pu.assert_close(got, want, ("model forward", deg, ray_fraction, name), mask=flips[None])


@pytest.mark.parametrize("deg,ray_fraction,P", [(3, 0.6, 12_000), (0, 0.6, 12_000), (1, 1.0, 8_000), (2, 0.0, 8_000),
                                                (3, 0.5, 119)])
def test_model_backward_equals_autograd_through_reference_getters_and_oracle(deg, ray_fraction, P):
    W, H = 256, 192
    sc, model = _model(P, W, H, ray_fraction, seed=2)
    model.active_sh_degree = deg
    cam = syn.orbit_camera(W, H, -4.0, 2.0, 7.0)
    bg = (0.1, 0.2, 0.3)
    grads = syn.make_upstream_grads(W, H, seed=5)
    md = _on_device(model, grad=True)
    md.active_sh_degree = deg
    t = mp.tensors_of(md)
    act = mp.activate(**t)
    # oracle side
    leaves, inputs = _oracle_side(model, act)
    m2 = torch.zeros(P, 3, requires_grad=True)
    oc, orad, od, oa = orc.rasterize(inputs["means3D"], m2, inputs["opacities"], pu.oracle_settings(cam, deg, bg),
                                     shs=inputs["shs"], scales=inputs["scales"], rotations=inputs["rotations"])
    ((oc * grads[0]).sum() + (od * grads[1]).sum() + (oa * grads[2]).sum()).backward()
    # HIP: render() takes the model path by itself
    camd = cam.to(DEV)
    bgd = torch.tensor(bg, device=DEV)
    assert rmod.model_fast_path_available(md, rmod.PipelineParams())
    out = rmod.render(camd, md, rmod.PipelineParams(), bgd)
    loss = (out["render"] * grads[0].to(DEV)).sum() + (out["rendered_depth"] * grads[1].to(DEV)).sum() + \
        (out["rendered_alpha"] * grads[2].to(DEV)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(out["radii"].cpu(), orad)
    assert torch.equal(out["visibility_filter"].cpu(), orad > 0)
    for name, got, want in (("render", out["render"], oc), ("depth", out["rendered_depth"], od), ("alpha", out["rendered_alpha"], oa)):
        assert pu.nrm_err(got, want.detach()) < 1e-4, name
    pu.assert_close(out["viewspace_points"].grad, m2.grad, ("model backward", deg, ray_fraction, "means2D"))
    names = ["zval", "features_dc", "features_rest", "opacity", "scaling", "rotation"]
    if md.bg_xyz.shape[0]:
        names += ["bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"]
    for n in names:
        g_hip, g_ref = getattr(md, n).grad, getattr(leaves, n).grad
        if getattr(md, n).shape[0] == 0:
            continue
        assert g_hip is not None and g_ref is not None, n
        assert float(g_ref.abs().max()) > 0 or "rest" in n, n
        pu.assert_close(g_hip, g_ref, ("model backward", deg, ray_fraction, P, n))
    # every gradient is a view of ONE arena (what parallel.GradBucket all-reduces in place)
    params = [p for p in md.parameters() if p.shape[0] > 0]
    assert R.grad_arena(params) is not None
    assert md.rayo.grad is None and md.rayd.grad is None


def test_render_model_path_equals_getter_path_and_accumulates_over_views():
    P, W, H, deg = 20_000, 320, 240, 3
    sc, model = _model(P, W, H, 0.55, seed=4)
    cams = [syn.orbit_camera(W, H, 5.0, -3.0, 7.0).to(DEV), syn.default_camera(W, H).to(DEV)]
    bg = torch.tensor((0.0, 0.0, 0.0), device=DEV)
    ups = [tuple(g.to(DEV) for g in syn.make_upstream_grads(W, H, seed=20 + i)) for i in range(2)]
    pipe = rmod.PipelineParams()

    def run(fast):
        md = _on_device(model, grad=True)
        rmod.MODEL_FAST_PATH = fast
        try:
            outs = []
            for cam, up in zip(cams, ups):                     # two views, gradients accumulated by autograd (no zero_grad)
                o = rmod.render(cam, md, pipe, bg)
                torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], list(up))
                outs.append(o)
        finally:
            rmod.MODEL_FAST_PATH = True
        torch.cuda.synchronize()
        return md, outs
    m_fast, o_fast = run(True)
    m_slow, o_slow = run(False)
    for a, b in zip(o_fast, o_slow):
        # two libraries' exp / sigmoid / normalize differ in the last bit: a radius may move by one on a knife edge
        assert int((a["radii"] != b["radii"]).sum()) <= 2
        for k in ("render", "rendered_depth", "rendered_alpha"):
            assert pu.nrm_err(a[k], b[k]) < 1e-4, k
    for pf, ps in zip(m_fast.parameters(), m_slow.parameters()):
        assert pu.nrm_err(pf.grad, ps.grad) < 2e-4


def test_sh_tail_zero_promise_of_the_pooled_gradient_arena():
    """Degree 0 / 1 steps (train.py:129): the second backward on a kept arena is told that the SH gradients above the active
    degree already hold zeros (SCG_BACKWARD_SH_TAIL_ZERO) and leaves them alone; a torch write through a gradient (version
    counter) or a degree change in either direction is noticed."""
    P, W, H = 9_000, 208, 160
    sc, model = _model(P, W, H, 0.5, seed=6)
    cam = syn.default_camera(W, H).to(DEV)
    bg = torch.zeros(3, device=DEV)
    up = tuple(g.to(DEV) for g in syn.make_upstream_grads(W, H, seed=9))
    md = _on_device(model, grad=True)
    pipe = rmod.PipelineParams()
    flags_seen = []
    orig = R._sh_tail_promise

    def spy(pa, n_active):
        f = orig(pa, n_active)
        flags_seen.append(f)
        return f
    R._sh_tail_promise = spy
    try:
        def step(deg):
            md.active_sh_degree = deg
            for p in md.parameters():
                p.grad = None
            o = rmod.render(cam, md, pipe, bg)
            torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], list(up))
            torch.cuda.synchronize()
            return {n: getattr(md, n).grad for n in ("features_dc", "features_rest", "bg_features_rest", "zval")}
        g0 = step(0)
        assert flags_seen[-1] == 0                                  # a fresh arena: the kernel writes the zeros
        assert float(g0["features_rest"].abs().max()) == 0 and float(g0["features_dc"].abs().max()) > 0
        ref_dc = g0["features_dc"].clone()
        g0 = None
        g1 = step(0)
        assert flags_seen[-1] == 2                                  # kept arena, untouched: tails left alone ...
        assert float(g1["features_rest"].abs().max()) == 0 and float(g1["bg_features_rest"].abs().max()) == 0
        assert pu.nrm_err(g1["features_dc"], ref_dc) < 1e-5        # ... and the active part written as before
        g1 = None
        g2 = step(1)                                                # degree raised: coefficients 1-3 now written, 4-15 still zero
        assert flags_seen[-1] == 2
        assert float(g2["features_rest"][:, :3].abs().max()) > 0 and float(g2["features_rest"][:, 3:].abs().max()) == 0
        g2["features_rest"].mul_(1.0)                               # a torch write through a gradient: the promise is off
        g2 = None
        g3 = step(1)
        assert flags_seen[-1] == 0
        assert float(g3["features_rest"][:, 3:].abs().max()) == 0
        g3 = None
        g4 = step(0)                                                # degree LOWERED: coefficients 1-3 hold old values -> rewritten
        assert flags_seen[-1] == 0
        assert float(g4["features_rest"].abs().max()) == 0
        held = g4                                                   # the caller keeps these gradients: the arena is not reused
        g5 = step(0)
        assert g5["zval"].untyped_storage().data_ptr() != held["zval"].untyped_storage().data_ptr()
    finally:
        R._sh_tail_promise = orig
    # the operator's tensor path takes the same promise for its (P, 16, 3) SH gradient
    scd = sc.to(DEV)
    leaves = [t.clone().requires_grad_(True) for t in (scd.means3D, scd.shs, scd.opacities, scd.scales, scd.rotations)]
    st = pu.hip_settings(syn.default_camera(W, H), 0, (0.0, 0.0, 0.0))
    for it in range(3):
        for p in leaves:
            p.grad = None
        c, _, d, a = R.GaussianRasterizer(st)(means3D=leaves[0], means2D=torch.zeros_like(leaves[0], requires_grad=True),
                                              opacities=leaves[2], shs=leaves[1], scales=leaves[3], rotations=leaves[4])
        torch.autograd.backward([c, d, a], list(up))
        torch.cuda.synchronize()
        assert float(leaves[1].grad[:, 1:].abs().max()) == 0 and float(leaves[1].grad[:, :1].abs().max()) > 0, it


def test_render_remembers_the_validated_model_and_notices_a_densification():
    """render() keeps the validated _ModelArgs on the model object (model_path.model_for) while its tensors are the same objects
    at the same addresses; replaced tensors (what densify_and_prune does: new Parameters of another length,
    scene/gaussian_model.py:898-931) or a `.data` swap are noticed at the next render."""
    P, W, H = 6_000, 256, 192
    sc, model = _model(P, W, H, 0.5, seed=6)
    md = _on_device(model)
    cam, bg, pipe = syn.orbit_camera(W, H, 1.0, 1.0, 7.0).to(DEV), torch.zeros(3, device=DEV), rmod.PipelineParams()

    def both():
        with torch.no_grad():
            fast = rmod.render(cam, md, pipe, bg)
            rmod.MODEL_FAST_PATH = False
            try:
                slow = rmod.render(cam, md, pipe, bg)
            finally:
                rmod.MODEL_FAST_PATH = True
        torch.cuda.synchronize()
        assert fast["radii"].shape == slow["radii"].shape and int((fast["radii"] != slow["radii"]).sum()) <= 2
        assert pu.nrm_err(fast["render"], slow["render"]) < 1e-4
        return fast
    both()
    a1 = mp.model_for(md)
    assert a1 is not None and mp.model_for(md) is a1                                 # remembered
    # densification: the background set grows (clone) — new tensors under the same attribute names
    n_add = 500
    for name in ("bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"):
        t = getattr(md, name)
        setattr(md, name, torch.cat([t, t[:n_add] + (0.05 if name == "bg_xyz" else 0.0)]).contiguous())
    out = both()
    a2 = mp.model_for(md)
    assert a2 is not a1 and a2.P == P + n_add and out["radii"].shape[0] == P + n_add
    # a `.data` swap keeps the tensor object: the address gives it away
    md.zval.data = md.zval.data.clone() * 1.05
    both()
    assert mp.model_for(md) is not a2
    # a model the kernels cannot take as it is (a non-contiguous tensor): through the getters, same values
    md.scaling = md.scaling.t().contiguous().t()
    assert mp.model_for(md) is None and not rmod.model_fast_path_available(md, pipe)


def test_render_views_sums_the_views_gradients_in_one_node():
    """render_views (K views of one model in one autograd node: cfg5's batched step on the model path) against K separate
    render() calls whose gradients autograd accumulates; one arena, per-view screen-space gradients; a view that does not reach
    the loss contributes nothing."""
    P, W, H = 15_000, 320, 240
    sc, model = _model(P, W, H, 0.6, seed=8)
    cams = [syn.orbit_camera(W, H, yaw, 1.0, 7.0).to(DEV) for yaw in (-6.0, 0.0, 7.0)]
    bg, pipe = torch.zeros(3, device=DEV), rmod.PipelineParams()
    ups = [tuple(g.to(DEV) for g in syn.make_upstream_grads(W, H, seed=40 + i)) for i in range(3)]
    used = (0, 2)                                                   # the middle view's outputs do not reach the loss

    def separate():
        md = _on_device(model, grad=True)
        outs = []
        for k in used:
            o = rmod.render(cams[k], md, pipe, bg)
            torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], list(ups[k]))
            outs.append(o)
        torch.cuda.synchronize()
        return md, outs
    m_sep, o_sep = separate()
    md = _on_device(model, grad=True)
    outs = rmod.render_views(cams, md, pipe, bg)
    ts, gs = [], []
    for k in used:
        ts += [outs[k]["render"], outs[k]["rendered_depth"], outs[k]["rendered_alpha"]]
        gs += list(ups[k])
    torch.autograd.backward(ts, gs)
    torch.cuda.synchronize()
    for o, k in zip(o_sep, used):
        assert torch.equal(outs[k]["render"], o["render"]) and torch.equal(outs[k]["radii"], o["radii"])
        pu.assert_close(outs[0]["viewspace_points_all"].grad[k], o["viewspace_points"].grad, ("render_views", "means2D", k))
    assert float(outs[0]["viewspace_points_all"].grad[1].abs().max()) == 0.0
    for pv, ps in zip(md.parameters(), m_sep.parameters()):
        assert pu.nrm_err(pv.grad, ps.grad) < 2e-5
    assert R.grad_arena([p for p in md.parameters() if p.shape[0] > 0]) is not None


def test_a_model_whose_background_set_is_the_references_empty_initialisation():
    """create_from_pcd leaves `bg_xyz`, `bg_features_dc`, ... as `nn.Parameter(torch.empty(0).cuda())` — ONE-dimensional empties
    (scene/gaussian_model.py:462-467) — until the first densification: the model path takes them as "no background set"."""
    P, W, H = 5_000, 256, 192
    sc, model = _model(P, W, H, 1.0, seed=9)
    md = _on_device(model, grad=True)
    for name in ("bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"):
        setattr(md, name, torch.nn.Parameter(torch.empty(0, device=DEV)))
    cam, bg, pipe = syn.orbit_camera(W, H, 2.0, 1.0, 7.0).to(DEV), torch.zeros(3, device=DEV), rmod.PipelineParams()
    assert rmod.model_fast_path_available(md, pipe)
    out = rmod.render(cam, md, pipe, bg)
    (out["render"].sum() + out["rendered_depth"].sum()).backward()
    torch.cuda.synchronize()
    assert md.zval.grad is not None and float(md.zval.grad.abs().max()) > 0 and md.bg_xyz.grad is None
    rmod.MODEL_FAST_PATH = False
    try:
        with torch.no_grad():
            slow = rmod.render(cam, md, pipe, bg)
    finally:
        rmod.MODEL_FAST_PATH = True
    assert pu.nrm_err(out["render"].detach(), slow["render"]) < 1e-4 and int((out["radii"] != slow["radii"]).sum()) <= 2
