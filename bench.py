#!/usr/bin/env python3
"""bench.py — throughput of the rasterizer hot path on MI355X (contract: see the repo's DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S2]

A "step" is one training iteration of the hot path over one view of a synthetic scene: forward
(geometry -> binning -> 16x16-tile blend) + backward (per-pixel backward -> geometry backward) through the
drop-in GaussianRasterizer, with fixed synthetic upstream gradients dL/dcolor, dL/ddepth, dL/dalpha, all
inputs resident in HBM.  N > 1: one process per GPU, data-parallel over views, ONE RCCL all-reduce of the
236 B/Gaussian parameter-gradient bucket per step (inside the timed region).

Rank 0 prints ONE JSON line:
  metric/value  train iterations (= views) per second, whole job
  render_mpix_per_sec   forward-only megapixels per second (the other half of BASELINE.json's metric)
  roofline      dominant kernel's algorithmic bytes / mean HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline  the pure-PyTorch CPU oracle timed on this box's host cores on a bounded sample (N=1 only)
  s3_forward    the north-star point: 500k Gaussians @ 1920x1080, forward only, per-stage times + roofline
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import sys
import time

import torch
import torch.utils._python_dispatch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scgaussian_amd import parallel as par                       # noqa: E402
from scgaussian_amd import rasterizer as R                        # noqa: E402
from scgaussian_amd import synthetic as syn                       # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # same guide: measured float4-copy ceiling (79 % of spec)
N_VIEWS = 3                    # LLFF 3-view training (scene/dataset_readers.py:165)


def sort_in_blend(R_, W, H):
    """True when the one-call forward sorts the tiles' lists inside the forward blend (tile_blend_forward_kernel): the
    sort's 12 bytes per instance (ids read, depth keys gathered, ids written) then belong to `blend_forward`."""
    from scgaussian_amd import _lib
    cap = R._capacity_for(int(R_))
    return _lib.load().scg_forward_sorts_in_blend(cap, W, H, 0 if R.FUSED_SORT else 1) == 1


def algorithmic_bytes(P, V, R_, W, H, deg):
    """Bytes each stage must move at minimum (BASELINE.md §2 / SURVEY §8d for the per-Gaussian and per-pixel stages).
    `binning` is priced for the algorithm that RUNS (tile-first binning, csrc/binning_tiles.hip), not for the
    reference's 6-pass global radix sort it replaces: rectangles read once per band (8) by the scatter (8 B each; once more
    by the histogram kernel where it is a kernel of its own), the [B][Tn] slice table written (by the geometry kernel on
    the one-call path), column-scanned (read + write) and read by the scatter,
    4 B per instance written by the scatter, read + written by the per-tile sort, plus its 4 B depth-key gather; tile
    totals / starts / ranges / launch order (20 B per tile).  `binning_reference_scheme` keeps the old figure for
    comparison with a CUDA-style pipeline."""
    K = (deg + 1) ** 2
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    passes = math.ceil((32 + max(1, math.ceil(math.log2(max(Tn, 2))))) / 8)
    B = 256 if (R._capacity_for(R_) >= (1 << 20) or P >= 100000) else 128              # tile_binning_blocks()
    sort_bytes = 12 * R_                              # per-tile sort: 4 B ids read + 4 B keys gathered + 4 B ids written
    fused = sort_in_blend(R_, W, H)
    # the one-call path's geometry kernel builds the slice histograms itself: the histogram's read of the rectangles is gone
    # and the table is written by the geometry stage
    hist = (0, 4 * B * Tn) if R.FUSED_HIST else (8 * P + 4 * B * Tn, 0)
    return {
        "geometry_forward": 52 * P + (12 * K + 67) * V + 8 * P + hist[1],               # preprocess + scan (+ histogram rows)
        "binning": 8 * P * 8 + 4 * B * Tn * 3 + hist[0] + 4 * R_ + 20 * Tn + (0 if fused else sort_bytes),
        "binning_reference_scheme": 20 * V + 12 * R_ + 24 * R_ * passes + 8 * R_ + 8 * R_ + 8 * Tn,
        "blend_forward": 44 * R_ + 8 * Tn + 28 * W * H + (sort_bytes if fused else 0),
        # a render that is not differentiated leaves the backward's per-pixel state (final_T, n_contrib) unwritten
        "blend_forward_render": 44 * R_ + 8 * Tn + 20 * W * H + (sort_bytes if fused else 0),
        "blend_backward": 124 * R_ + 28 * W * H,
        "geometry_backward": (371 + 12 * K) * V,
    }


# which BASELINE.json config a synthetic workload stands for (BASELINE.json `configs`, SURVEY §8d sizes)
WORKLOAD_CONFIG = {
    "S1": "BASELINE configs[0] shape: 10 k Gaussians, 256x256 — the reference's own scene size at the start of training",
    "S2": "BASELINE configs[1] shape: LLFF-fern 3-view training, 200 k Gaussians @ 1008x756",
    "S2r8": "BASELINE configs[1] Gaussian count at 504x378 (what `-r 8` yields on LLFF)",
    "S3": "north-star roofline point: 500 k Gaussians @ 1920x1080",
    "S4": "BASELINE configs[4] per-view shape: >= 1 M Gaussians @ 960x540, multi-view step with gradient all-reduce",
}


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:                                  # noqa: BLE001
        return f"unavailable ({type(e).__name__})"


def backward_with_given_grads(outs, grads):
    """torch.autograd.backward(outs, grads) without its Python-side validation of every (output, gradient) pair
    (torch/autograd/__init__.py _make_grads: shape comparison through sym_eq, ~20 us of host time for three image-sized
    gradients): the bench hands the engine fixed upstream gradients whose shapes it built itself.  A training loop calls
    loss.backward() on a scalar and never pays that check; at the reference's own scene size (S1, host-bound) it is 15 % of
    the step.  SECONDARY figure of the host-bound small legs only (`small_workloads.*.ms_per_step_engine_direct`); their primary
    `ms_per_step`, the headline loop and the clustered legs go through the public torch.autograd.backward."""
    global ENGINE_DIRECT_OK
    if ENGINE_DIRECT_OK:
        try:
            torch.autograd.Variable._execution_engine.run_backward(tuple(outs), tuple(grads), False, False, (),
                                                                   allow_unreachable=True, accumulate_grad=True)
            return
        except TypeError:                                      # another torch version's engine signature: said so in the line
            ENGINE_DIRECT_OK = False
    torch.autograd.backward(list(outs), list(grads))


ENGINE_DIRECT_OK = True


def make_views(W, H):
    return [syn.default_camera(W, H), syn.orbit_camera(W, H, 6.0, 0.0, 7.0), syn.orbit_camera(W, H, -6.0, 2.0, 7.0)]


def settings_for(cam, deg, bg, dev):
    camd = cam.to(dev)
    return R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2),
                                           math.tan(cam.FoVy / 2), bg, 1.0, camd.world_view_transform,
                                           camd.full_proj_transform, deg, camd.camera_center, False, False)


def render_once(sett, params):
    """One forward through the path the binding takes in production: the one-call fast path when a capacity is known
    for the shape, the staged path otherwise (first call).  Returns (radii, num_rendered)."""
    means, shs, opac, scales, rots = params
    out = R.forward_fused(sett, means, opac, shs, None, scales, rots, None, False)
    if out is not None:
        return out[1], out[4]["num_rendered"]
    fs = R.forward_stages(sett, means, opac, shs=shs, scales=scales, rotations=rots)
    return fs["radii"], fs["num_rendered"]


def forward_only(setts, params, steps, warmup, timer):
    """Render leg.  Timed region: the `steps` renders, wall clock between two synchronisations, NO events inside (two event
    records per render cost a 0.13-0.24 ms render 3-5 %: round 4 moved them out); the dominant kernel (blend_forward) is timed
    with HIP events in a second pass of the same renders, every stage in a third."""
    with torch.no_grad():
        R.set_stage_timer(None)
        for i in range(warmup):
            render_once(setts[i % len(setts)], params)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fs = None
        for i in range(steps):
            fs = render_once(setts[i % len(setts)], params)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timed = R.StageTimer(only=("blend_forward",))
        R.set_stage_timer(timed)
        for i in range(steps):
            render_once(setts[i % len(setts)], params)
        dominant = timed.summary()
        timer.reset()
        R.set_stage_timer(timer)
        for i in range(steps):
            render_once(setts[i % len(setts)], params)
        stages = timer.summary()
        stages.update(dominant)
    return dt, {"radii": fs[0], "num_rendered": fs[1]}, stages


_PMC = None


def _pmc_traffic():
    """Per-kernel PMC results collected with rocprofv3 in separate passes (tools/pmc.sh -> tools/pmc_summary.py),
    committed as profiles/pmc_summary.json (counters cannot be collected from inside this process).  The file is stamped
    with the sha256 of the kernel sources (and of the .so) it was collected on: counters of OTHER kernels than the ones
    being timed are not printed — `source` says why.  Returns (per-workload dict, source string)."""
    global _PMC
    if _PMC is not None:
        return _PMC
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        _PMC = ({}, "none: profiles/pmc_summary.json absent")
        return _PMC
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary as ps
        from scgaussian_amd import _lib
        st = data.get("_stamp") or {}
        src_now, lib_now = ps.kernel_source_sha256(), ps.lib_sha256(_lib.LIB_PATH)
        if st.get("kernel_source_sha256") and st["kernel_source_sha256"] == src_now:
            how = "same .so" if st.get("lib_sha256") == lib_now else "same kernel sources, rebuilt .so"
            _PMC = (data, f"profiles/pmc_summary.json ({data.get('_source')}, commit {str(st.get('commit'))[:12]}): {how}")
        else:
            _PMC = ({}, "stale: profiles/pmc_summary.json was collected on other kernel sources "
                        f"(stamp {str(st.get('kernel_source_sha256'))[:12]}, running {src_now[:12]}) - counters withheld")
    except Exception as e:                                  # noqa: BLE001
        _PMC = ({}, f"unverifiable: {type(e).__name__}: {e}")
    return _PMC


def roofline_for(stage, ms, alg_bytes, workload=None):
    """achieved = algorithmic bytes / mean HIP-event duration.  `traffic` = HBM bytes per launch from the committed PMC
    passes (profiles/pmc_summary.json, produced by tools/pmc_summary.py from tools/pmc.sh output).  The blend kernels
    are bound by vector-instruction issue, not HBM: `valu` reports how full the vector pipe is, from SQ_INSTS_VALU of
    the same PMC pass, the shader clock of that pass (GRBM_GUI_ACTIVE / duration) and the per-class issue costs
    measured by tools/probes/clock_probe.hip — once with every instruction at the plain 2-cycle wave64 rate of the
    SIMD-32 (a lower bound), once weighted by the kernel's static instruction mix (DPP / select ~3.4, transcendental 8,
    packed fp32 4 cycles)."""
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    data, source = _pmc_traffic()
    pmc = data.get(workload or "", {}).get(stage, {})
    return {"kernel": stage, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4),
            "traffic": pmc.get("hbm_bytes"), "traffic_source": source,
            "algorithmic_bytes": int(alg_bytes), "mean_ms": round(ms, 4),
            "valu": {k: pmc[k] for k in ("insts_valu", "shader_clock_ghz", "issue_frac_all_plain_2cyc",
                                          "issue_frac_mix_weighted", "cycles_per_inst_mix") if k in pmc} or None}


CPU_THREADS_CAP = 16     # measured on the 256-core GPU box: the per-tile torch ops of the oracle peak at 16
                         # threads (8: 0.046, 16: 0.067, 32: 0.038, 64: 0.016 iters/s); 256 threads does not finish


def cpu_baseline(P, W, H, deg, tile_stride=8):
    """The CPU oracle (oracle/torch_rasterizer.py — a 'port': the reference has no CPU rasterizer) on the host
    cores: full preprocess + binning of the workload, every tile_stride-th tile blended fwd+bwd, blend time
    extrapolated to all tiles by instance count."""
    from oracle import torch_rasterizer as orc           # checker / baseline leg only
    cores = min(os.cpu_count() or 1, CPU_THREADS_CAP)
    torch.set_num_threads(cores)
    sc = syn.make_scene(P, W, H, seed=0)
    cam = syn.default_camera(W, H)
    st = orc.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3), 1.0,
                      cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
    dc, dd, da = syn.make_upstream_grads(W, H)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, torch.zeros(P, 3), sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m, m2, sh, op, s, r = leaves
    t0 = time.perf_counter()
    pre = orc.preprocess(m, m2, op, st, shs=sh, scales=s, rotations=r)
    binning = orc.bin_and_sort(pre, W, H)
    t1 = time.perf_counter()
    c, d, a, _, _ = orc.blend(pre, binning, st, tile_stride=tile_stride)
    t2 = time.perf_counter()
    torch.autograd.backward([c, d, a], [dc, dd, da])
    t3 = time.perf_counter()
    rng = binning["ranges"].astype("int64")
    counts = rng[:, 1] - rng[:, 0]
    r_sample = int(counts[::tile_stride].sum())
    r_total = int(counts.sum())
    scale = r_total / max(r_sample, 1)
    t_pre = t1 - t0
    t_blend_fwd = (t2 - t1) * scale
    # backward = blend backward of the sample (extrapolated) + preprocess backward (full); they are not separable
    # in one autograd call, so the whole backward is extrapolated by the blend ratio only for its blend share:
    # measured: the preprocess backward is <5 % of the sample's backward, so extrapolating all of it over-states
    # the CPU time slightly (conservative for the CPU: stated in DESIGN.md).
    t_bwd = (t3 - t2) * scale
    t_full = t_pre + t_blend_fwd + t_bwd
    return {"value": round(1.0 / t_full, 5), "unit": "iters/s", "cores": cores, "host_cpus": os.cpu_count(),
            "kind": "port",
            "sample": (f"P={P} {W}x{H} deg{deg}: one full fwd+bwd of the workload (all {r_total} instances, every tile "
                       f"blended, nothing extrapolated)") if tile_stride == 1 else
                      (f"P={P} {W}x{H} deg{deg}: full preprocess+binning, every {tile_stride}th tile blended fwd+bwd "
                       f"({r_sample} of {r_total} instances), blend+backward time extrapolated x{scale:.2f}"),
            "measured_s": round(t3 - t0, 2), "fwd_value": round(1.0 / (t_pre + t_blend_fwd), 5),
            "fwd_unit": "renders/s"}


def _synthetic_match_pairs(views, depth_img, M, dev):
    """Matches between view 0 and the other two views, built from a rendered depth map (synthetic stand-in for the
    reference's match_data.npy: <= 2000 matches per ordered pair, data_preprocess/get_match_info.py:376)."""
    cam0 = views[0]
    W, H = cam0.image_width, cam0.image_height

    def intr(cam):
        fx, fy = W / (2 * math.tan(cam.FoVx / 2)), H / (2 * math.tan(cam.FoVy / 2))
        return torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]], dtype=torch.float32)

    g = torch.Generator().manual_seed(5)
    K0 = intr(cam0)
    c2w0 = torch.linalg.inv(cam0.world_view_transform.t())
    pairs = []
    depth_cpu = depth_img.detach().float().cpu().reshape(H, W)
    for cam1 in views[1:]:
        uv0 = torch.stack([torch.rand(M, generator=g) * W, torch.rand(M, generator=g) * H], 1)
        cr = (torch.linalg.inv(K0) @ torch.cat([uv0, torch.ones(M, 1)], 1).t()).t()
        cr = cr / cr.norm(dim=1, keepdim=True)
        rd = (c2w0[:3, :3] @ cr.t()).t().contiguous()
        ro = c2w0[:3, 3][None].repeat(M, 1).contiguous()
        d = torch.nn.functional.grid_sample(depth_cpu[None, None], torch.stack([uv0[:, 0] / W * 2 - 1, uv0[:, 1] / H * 2 - 1], -1)[None, None],
                                            align_corners=False).reshape(-1).clamp_min(0.5)
        w2c1 = cam1.world_view_transform.t().contiguous()
        world = ro + rd * (d / cr[:, 2])[:, None]
        xyz = intr(cam1) @ (w2c1 @ torch.cat([world, torch.ones(M, 1)], 1).t())[:3]
        uv1 = (xyz[:2] / (xyz[2:] + 1e-8)).t().contiguous() + torch.randn(M, 2, generator=g)
        pairs.append({k: v.to(dev) for k, v in dict(uv0=uv0.contiguous(), rays_o=ro, rays_d=rd, cam_rays_d=cr.contiguous(),
                                                     mask0=torch.ones(M), mask1=torch.ones(M), intr1=intr(cam1), w2c1=w2c1,
                                                     uv1=uv1).items()})
    return pairs


def full_iteration_leg(P, W, H, deg, dev, steps):
    """One iteration of the reference's main training stage reduced to GPU work (train.py:143-208): render ->
    0.8 L1 + 0.2 (1-SSIM) -> + 0.3 match loss on the rendered depth -> backward -> Adam.  Timed with the fused loss
    kernels of this repo and, for comparison, with the reference's torch formulation of the image loss (five
    grouped conv2d + autograd).  Extra information: NOT the headline value."""
    import numpy as np
    import torch.nn.functional as F
    from scgaussian_amd import losses
    from scgaussian_amd.match_loss import match_loss_from_depth
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    views = make_views(W, H)
    bg = torch.zeros(3, device=dev)
    setts = [settings_for(v, deg, bg, dev) for v in views]
    params = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    means, shs, opac, scales, rots = params
    with torch.no_grad():
        gt = [R.GaussianRasterizer(s)(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs,
                                       scales=scales, rotations=rots) for s in setts]
    pairs = _synthetic_match_pairs(views, gt[0][2], 2000, dev)
    targets = [(g_[0] + 0.05 * torch.randn_like(g_[0])).clamp(0, 1) for g_ in gt]
    # the reference's optimizer (scene/gaussian_model.py:511: torch.optim.Adam(l, lr=0.0, eps=1e-15), torch's default multi-tensor
    # implementation) and the same optimizer with fused=True (one kernel over all parameter tensors): the caller's choice, timed
    # here because behind a 0.24 ms rasterizer step the optimizer is the next largest piece of an iteration
    opts = {"default": torch.optim.Adam(params, lr=1e-4, eps=1e-15)}
    try:
        opts["fused"] = torch.optim.Adam(params, lr=1e-4, eps=1e-15, fused=True)
    except (RuntimeError, TypeError):
        pass
    gauss = torch.tensor([np.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32)
    gauss = gauss / gauss.sum()
    win = (gauss[:, None] @ gauss[None, :]).expand(3, 1, 11, 11).contiguous().to(dev)

    def torch_image_loss(x, y):
        conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]      # noqa: E731
        mu1, mu2 = conv(x), conv(y)
        s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
        smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        return 0.8 * (x - y).abs().mean() + 0.2 * (1 - smap.mean())

    def iteration(i, fused, opt):
        v = i % len(setts)
        means2D = torch.zeros_like(means, requires_grad=True)
        c, radii, d, a = R.GaussianRasterizer(setts[v])(means3D=means, means2D=means2D, opacities=opac, shs=shs,
                                                         scales=scales, rotations=rots)
        loss = losses.image_loss(c, targets[v], 0.2) if fused else torch_image_loss(c, targets[v])
        if v == 0:
            loss = loss + 0.3 * match_loss_from_depth(d, pairs, float(W), float(H))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    out = {}
    legs = [("fused_losses", True, "default"), ("torch_image_loss", False, "default")]
    if "fused" in opts:
        legs.insert(1, ("fused_losses_fused_adam", True, "fused"))
    for name, fused, which in legs:
        for i in range(3):
            iteration(i, fused, opts[which])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            iteration(i, fused, opts[which])
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    out["workload"] = f"{P} Gaussians, {W}x{H}: render + 0.8 L1 + 0.2 (1-SSIM) + 0.3 match loss (2 pairs x 2000) + backward + Adam"
    out["iters_per_sec_fused"] = round(1e3 / out["fused_losses_ms"], 2)
    out["iters_per_sec_torch_image_loss"] = round(1e3 / out["torch_image_loss_ms"], 2)
    if "fused_losses_fused_adam_ms" in out:
        out["iters_per_sec_fused_losses_fused_adam"] = round(1e3 / out["fused_losses_fused_adam_ms"], 2)
    return out


def moving_scene_leg(P, W, H, deg, dev, iters=2000, densify_every=100):
    """Speculation and hints under a MOVING scene: `iters` training iterations with Adam steps at the
    reference's learning rates (arguments/__init__.py:76-84) and, every `densify_every` iterations (train.py:195: 100), a random
    clone of 5 % of the Gaussians + a random prune of 4 % (new parameter tensors, new optimizer state, P changes: what
    densify_and_prune does to the operator's inputs, scene/gaussian_model.py:898-931).  Counters from the binding
    (rasterizer.speculation_stats): how often the capacity missed (the forward ran twice), how many forwards fell to the staged
    path, how many had their camera's launch-order hints; the step time next to the same loop WITHOUT densification."""
    from scgaussian_amd import losses
    views = make_views(W, H)
    bg = torch.zeros(3, device=dev)
    setts = [settings_for(v, deg, bg, dev) for v in views]
    lrs = {"means3D": 1.6e-4 * 5.0, "shs": 2.5e-3, "opacities": 5e-2, "scales": 5e-3, "rotations": 1e-3}

    def fresh(sc_tensors):
        ps = [t.detach().clone().requires_grad_(True) for t in sc_tensors]
        opt = torch.optim.Adam([{"params": [p_], "lr": lr} for p_, lr in zip(ps, lrs.values())], eps=1e-15)
        return ps, opt

    def run(densify):
        sc = syn.make_scene(P, W, H, seed=0).to(dev)
        ps, opt = fresh((sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations))
        with torch.no_grad():
            targets = [(R.GaussianRasterizer(s_)(means3D=ps[0], means2D=torch.zeros_like(ps[0]), opacities=ps[2], shs=ps[1],
                                                  scales=ps[3], rotations=ps[4])[0] + 0.1 * torch.randn(3, H, W, device=dev)
                        ).clamp(0, 1) for s_ in setts]
        g = torch.Generator(device=dev).manual_seed(1)
        for i in range(6):                                   # every camera seen: the loop below starts with hints, like iteration 7
            step(i, ps, opt, targets)
        torch.cuda.synchronize()
        R.speculation_stats(reset=True)
        Pmin = Pmax = ps[0].shape[0]
        t0 = time.perf_counter()
        for i in range(iters):
            step(i, ps, opt, targets)
            if densify and (i + 1) % densify_every == 0:
                with torch.no_grad():
                    n = ps[0].shape[0]
                    keep = torch.rand(n, device=dev, generator=g) >= 0.04
                    clone = torch.rand(n, device=dev, generator=g) < 0.05
                    new = []
                    for k, p_ in enumerate(ps):
                        extra = p_[clone]
                        if k == 0:
                            extra = extra + 0.01 * torch.randn(extra.shape, device=dev, generator=g)
                        new.append(torch.cat([p_[keep], extra]))
                ps, opt = fresh(new)
                Pmin, Pmax = min(Pmin, ps[0].shape[0]), max(Pmax, ps[0].shape[0])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        st = R.speculation_stats()
        return ms, st, (Pmin, Pmax)

    def step(i, ps, opt, targets):
        v = i % len(setts)
        m_, f_, o_, s_, r_ = ps
        c, _, _, _ = R.GaussianRasterizer(setts[v])(means3D=m_, means2D=torch.zeros_like(m_, requires_grad=True), opacities=o_,
                                                    shs=f_, scales=s_, rotations=r_)
        loss = losses.image_loss(c, targets[v], 0.2)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    R.set_stage_timer(None)
    static_ms, static_st, _ = run(False)
    moving_ms, st, (Pmin, Pmax) = run(True)
    n_fw = max(st["one_call_forwards"] + st["staged_forwards"], 1)
    return {"workload": f"{P} Gaussians at the start ({Pmin} .. {Pmax} during the run), {W}x{H}, SH degree {deg}: render + fused "
                        f"0.8 L1 + 0.2 (1-SSIM) + backward + Adam (reference learning rates), {iters} iterations, clone 5 % + prune "
                        f"4 % at random every {densify_every}",
            "iterations": iters, "densifications": iters // densify_every,
            "ms_per_iteration": round(moving_ms, 4), "ms_per_iteration_static_scene": round(static_ms, 4),
            "moving_over_static": round(moving_ms / static_ms, 4),
            "overflow_retries": st["overflow_retries"], "overflow_retries_per_1000_steps": round(1e3 * st["overflow_retries"] / iters, 2),
            "retry_rate": round(st["overflow_retries"] / n_fw, 5),
            "staged_forwards": st["staged_forwards"], "one_call_forwards": st["one_call_forwards"],
            "tile_cost_hint_rate": round(st["tile_cost_hints"] / max(st["one_call_forwards"], 1), 4),
            "bwd_order_hint_rate": round(st["bwd_order_hints"] / max(st["one_call_forwards"], 1), 4),
            "static_scene": {"overflow_retries": static_st["overflow_retries"], "staged_forwards": static_st["staged_forwards"],
                             "tile_cost_hint_rate": round(static_st["tile_cost_hints"] / max(static_st["one_call_forwards"], 1), 4)},
            "moving_includes": "the densification's own torch work (masks, cat, a new Adam) every 100th iteration"}


class _OpCounter(torch.utils._python_dispatch.TorchDispatchMode):
    """Counts the aten operators dispatched while it is active (views and metadata queries excluded): on a GPU every one of
    them is (at least) one kernel launch.  The rasterizer's own launches are not aten operators — the library reports them."""
    _FREE = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.slice", "aten.select",
             "aten.split", "aten.split_with_sizes", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze",
             "aten.as_strided", "aten.empty", "aten.empty_like", "aten.empty_strided", "aten.new_empty", "aten.set_",
             "aten.is_", "aten.sym_", "aten.size", "aten.stride", "aten.storage_offset", "aten.unbind", "aten.narrow",
             "aten.permute", "aten._local_scalar_dense", "aten.lift_fresh", "aten.result_type", "aten.item")

    def __init__(self):
        super().__init__()
        self.n = 0
        self.names = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(name.startswith(f) for f in self._FREE):
            self.n += 1
            self.names[name] = self.names.get(name, 0) + 1
        return func(*args, **(kwargs or {}))


def render_glue_leg(name, deg, dev, steps):
    """The path AS THE REFERENCE CALLS IT: a training step through `render()` on a model in the reference's raw
    parameterisation (scene/gaussian_model.py:452-468: ray-bound set + background set, logits, log-scales, un-normalised
    quaternions, features_dc + features_rest), the getters evaluated as often as gaussian_renderer/__init__.py:28,55-57,67-68,85
    evaluates them, backward down to `zval` — next to the bare operator on the same (pre-activated) scene, which is what the
    headline times.  `model_path`: the same render() with scgaussian_amd.render's fast path for models that carry the
    reference's raw tensors (activations and concatenations inside the geometry kernels, raw-parameter gradients in one arena)."""
    from scgaussian_amd import render as rmod
    w = syn.WORKLOADS[name]
    Ps, Ws, Hs = w["P"], w["width"], w["height"]
    sc = syn.make_scene(Ps, Ws, Hs, seed=0)
    model = syn.make_raw_model(sc).to(dev).requires_grad_()
    model.active_sh_degree = deg
    cams = [v.to(dev) for v in make_views(Ws, Hs)]
    pipe = rmod.PipelineParams()
    bg = torch.zeros(3, device=dev)
    us = [tuple(t.to(dev) for t in syn.make_upstream_grads(Ws, Hs, seed=10 + i)) for i in range(N_VIEWS)]
    params = model.parameters()
    with torch.no_grad():
        leaves = [model.get_xyz, model.get_features, model.get_opacity, model.get_scaling, model.get_rotation]
    leaves = [t.detach().clone().requires_grad_(True) for t in leaves]
    rs = [R.GaussianRasterizer(settings_for(v, deg, bg, dev)) for v in make_views(Ws, Hs)]

    def glue_step(i, fast):
        for p_ in params:
            p_.grad = None
        rmod.MODEL_FAST_PATH = fast
        o = rmod.render(cams[i % 3], model, pipe, bg, reference_call_pattern=not fast)
        torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], list(us[i % 3]))

    def bare_step(i):
        for p_ in leaves:
            p_.grad = None
        m_, f_, o_, s_, r_ = leaves
        c_, _, d_, a_ = rs[i % 3](means3D=m_, means2D=torch.zeros_like(m_, requires_grad=True), opacities=o_, shs=f_,
                                  scales=s_, rotations=r_)
        torch.autograd.backward([c_, d_, a_], list(us[i % 3]))

    n = max(30, steps)
    R.set_stage_timer(None)

    def timed(fn):
        for i in range(12):
            fn(i)
        reps = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / n * 1e3)
        return sorted(reps)[1], [round(r_, 4) for r_ in reps]

    def ops(fn):
        with _OpCounter() as oc:
            fn(0)
        torch.cuda.synchronize()
        return oc.n, dict(sorted(oc.names.items(), key=lambda kv: -kv[1])[:12])

    gc.collect()
    gc.disable()
    try:
        bare_ms, bare_reps = timed(bare_step)
        glue_ms, glue_reps = timed(lambda i: glue_step(i, False))
        rmod.MODEL_FAST_PATH = True                          # (glue_step(…, False) left the switch off: ask with it on)
        has_fast = rmod.model_fast_path_available(model, pipe)
        fast_ms, fast_reps = timed(lambda i: glue_step(i, True)) if has_fast else (None, None)
    finally:
        gc.enable()
        rmod.MODEL_FAST_PATH = True
    n_bare, _ = ops(bare_step)
    n_glue, top_glue = ops(lambda i: glue_step(i, False))
    n_fast, top_fast = ops(lambda i: glue_step(i, True)) if has_fast else (None, None)
    out = {"workload": f"{name}: {Ps} Gaussians ({model.zval.shape[0]} ray-bound + {model.bg_xyz.shape[0]} background), "
                       f"{Ws}x{Hs}, SH degree {deg}, fwd+bwd per view through render() down to zval",
           "bare_operator_ms": round(bare_ms, 4), "bare_operator_repetitions": bare_reps,
           "reference_glue_ms": round(glue_ms, 4), "reference_glue_repetitions": glue_reps,
           "glue_delta_ms": round(glue_ms - bare_ms, 4), "glue_delta_frac_of_bare": round(glue_ms / bare_ms - 1.0, 4),
           "torch_ops_per_step": {"bare_operator": n_bare, "reference_glue": n_glue, "model_path": n_fast},
           "reference_glue_top_ops": top_glue,
           "library_launches_per_step": "4 forward + 2 backward (frames without long lists) in every variant",
           "model_path_ms": None if fast_ms is None else round(fast_ms, 4), "model_path_repetitions": fast_reps,
           "model_path_delta_frac_of_bare": None if fast_ms is None else round(fast_ms / bare_ms - 1.0, 4),
           "model_path_top_ops": top_fast}
    return out


def by_sh_degree_leg(P, W, H, dev, steps, workload):
    """The SH degrees the reference TRAINS at (train.py:129: the active degree starts at 0 and rises every 1 000 iterations of a
    2 000-iteration run — degree 0 and 1, degree 2 on the last iteration; degree 3 only when a saved scene is rendered): the
    training step of the headline workload per degree, stage times, and the two per-Gaussian kernels against the roofline of
    THEIR OWN algorithmic bytes (the forward reads 12 K of each record's 192 bytes, the backward writes 12 K + zeros)."""
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    ps = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
    for p_ in ps:
        p_.requires_grad_(True)
    m_, f_, o_, s_, r_ = ps
    bg = torch.zeros(3, device=dev)
    us = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10 + i)) for i in range(N_VIEWS)]
    out = {}
    for deg in (0, 1, 2):
        rs = [R.GaussianRasterizer(settings_for(v, deg, bg, dev)) for v in make_views(W, H)]
        radii_box = [None]

        def st(i):
            for p_ in ps:
                p_.grad = None
            c_, radii_box[0], d_, a_ = rs[i % 3](means3D=m_, means2D=torch.zeros_like(m_, requires_grad=True), opacities=o_,
                                                 shs=f_, scales=s_, rotations=r_)
            torch.autograd.backward([c_, d_, a_], list(us[i % 3]))
        R.set_stage_timer(None)
        n = max(30, steps)
        for i in range(20):
            st(i)
        reps = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                st(i)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / n * 1e3)
        tm = R.StageTimer()
        R.set_stage_timer(tm)
        for i in range(12):
            st(i)
        stg = tm.summary()
        R.set_stage_timer(None)
        V = int((radii_box[0] > 0).sum().item())
        st0 = R._spec_state(dev).cam_hint
        R_ = max((v[1] for k, v in st0.items() if k[0] == W and k[1] == H), default=0)
        alg = algorithmic_bytes(P, V, R_, W, H, deg)
        out[str(deg)] = {"ms_per_step": round(sorted(reps)[1], 4), "ms_per_step_repetitions": [round(r, 4) for r in reps],
                         "stage_ms": {k: round(v[0], 4) for k, v in stg.items()},
                         "roofline": {k: roofline_for(k, stg[k][0], alg[k], f"{workload}_deg{deg}")
                                      for k in ("geometry_forward", "geometry_backward") if k in stg},
                         "sh_bytes_per_gaussian": {"forward_read": 12 * (deg + 1) ** 2, "backward_read": 12 * (deg + 1) ** 2,
                                                   "backward_written_active": 12 * (deg + 1) ** 2,
                                                   "backward_written_zeros": 192 - 12 * (deg + 1) ** 2}}
    return out


def cpu_baseline_guarded(P, W, H, deg, tile_stride, budget_s=150):
    """Run the CPU leg in a child process with a wall-clock budget so a slow host can never stall the bench."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print('CPUBASE ' + json.dumps(bench.cpu_baseline(%d,%d,%d,%d,%d)))" % (ROOT, P, W, H, deg, tile_stride))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s, env=env)
        for line in res.stdout.splitlines():
            if line.startswith("CPUBASE "):
                return json.loads(line[len("CPUBASE "):])
        return {"value": None, "unit": "iters/s", "cores": min(os.cpu_count() or 1, CPU_THREADS_CAP), "kind": "port",
                "sample": "failed: " + (res.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "iters/s", "cores": min(os.cpu_count() or 1, CPU_THREADS_CAP), "kind": "port",
                "sample": f"exceeded the {budget_s}s budget"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the LAUNCHER — it
    starts N copies of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one rank per visible GPU
    on RCCL; `--dist-backend gloo`: the ranks share the visible GPUs round-robin), passes rank 0's stdout (the one JSON
    line) through and exits with the first non-zero status of any rank.  A launch that cannot give every RCCL rank a GPU
    of its own is refused: it never degrades to fewer ranks than asked for (the reference pins cuda:0 and has
    no launcher, utils/general_utils.py:139)."""
    import socket
    import subprocess
    n = args.gpus
    backend = args.dist_backend or "nccl"
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU")
    if backend == "nccl" and n_dev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {n_dev} GPU(s) visible and RCCL places one rank per device — refusing "
                         f"to run fewer ranks than asked for (use --dist-backend gloo to let ranks share a GPU: a code-path "
                         f"check, not a scaling number)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SCG_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py launcher: rank {r} exited with status {code}; stopping the others", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 200 x 0.4 ms: long enough to average host hiccups out
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=5,
                    help="the K timed steps are repeated this many times back to back: value_median / _min / _max (BASELINE.md: "
                         "median of >= 5 runs); `value` is the first block")
    ap.add_argument("--workload", default="S2", choices=sorted(syn.WORKLOADS))
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--sustained-steps", type=int, default=1000,
                    help="untimed steps before the `sustained` repetition of the K timed steps (0: skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-s3", action="store_true")
    ap.add_argument("--no-full-iteration", action="store_true")
    ap.add_argument("--no-small", action="store_true", help="skip the S1 / S2r8 training-step legs")
    ap.add_argument("--no-rccl-floor", action="store_true", help="skip the RCCL world-of-one all-reduce of the gradient arena")
    ap.add_argument("--rccl-debug", action="store_true", help="NCCL_DEBUG=INFO for the ranks (which algorithm / protocol RCCL picks)")
    ap.add_argument("--rccl-algo", default=None, help="NCCL_ALGO for the ranks (e.g. Ring, Tree): which all-reduce algorithm RCCL "
                                                     "is ASKED for; recorded in config.rccl_requested (default: its tuner decides)")
    ap.add_argument("--rccl-proto", default=None, help="NCCL_PROTO for the ranks (e.g. Simple, LL, LL128); recorded likewise")
    ap.add_argument("--no-render-glue", action="store_true", help="skip the render()-on-the-reference's-model legs")
    ap.add_argument("--no-moving-scene", action="store_true", help="skip the 2 000-iteration densify / prune leg")
    ap.add_argument("--no-graph", action="store_true", help="skip the captured-step (hipGraph replay) legs")
    ap.add_argument("--no-by-degree", action="store_true", help="skip the SH degree 0 / 1 / 2 legs of the headline workload")
    ap.add_argument("--no-clustered", action="store_true", help="skip the non-uniform (clustered) scenes of the headline shape")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (default nccl = RCCL; gloo lets the N>1 path be "
                         "exercised with several ranks sharing one GPU)")
    ap.add_argument("--cpu-tile-stride", type=int, default=1)     # every tile: ~12 s of CPU work on S2, nothing extrapolated
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="K views per rank per step in ONE autograd node (GaussianRasterizerViews): the parameter gradients "
                         "of views 2..K are accumulated in the kernel, ONE gradient exchange per K views (BASELINE cfg5)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.rccl_debug:
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING")
    # the exchange rides on whatever all-reduce RCCL's tuner picks for the bucket size; a request goes to the ranks through
    # the environment (set before the process group exists; launch_ranks' children inherit it) and into the JSON line
    if args.rccl_algo:
        os.environ["NCCL_ALGO"] = args.rccl_algo
    if args.rccl_proto:
        os.environ["NCCL_PROTO"] = args.rccl_proto
    env_world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    if env_world != args.gpus:
        # a launcher (torchrun, ours) that disagrees with --gpus must never degrade to a silent single-GPU run
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: start `python bench.py --gpus N` (it launches its "
                         f"own N ranks) or `torchrun --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])           # does not return: this process only starts and reaps the N ranks
    # stdout carries the ONE JSON line and nothing else: libraries that write to the C-level stdout (RCCL prints a version
    # banner when a process group comes up) are sent to stderr; the line itself goes out through the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import scgaussian_amd
    scgaussian_amd.single_gpu_host_setup()     # one GPU per process: backward on the calling thread
    n_dev = torch.cuda.device_count()
    env_local = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world > 1 and args.dist_backend != "gloo" and env_local >= n_dev:
        raise SystemExit(f"rank with LOCAL_RANK={env_local} but only {n_dev} GPU(s) visible: RCCL needs a device per rank "
                         f"(--dist-backend gloo shares devices)")
    if args.dist_backend == "gloo":
        # several ranks on one GPU (test mode): pick the device ourselves; RCCL cannot place two ranks on one device,
        # so this mode always exchanges through gloo
        torch.cuda.set_device(env_local % n_dev)
        args.dist_backend = "gloo"
        rank, world, local_rank = par.init_from_env("gloo")
    else:
        # an explicit --dist-backend with one process runs the N-GPU exchange path in a world of one (RCCL group
        # creation, ReduceOp.AVG on the gradient arena in place): the code the first 8-GPU launch will execute
        forced = args.dist_backend is not None and int(os.environ.get("WORLD_SIZE", "1")) == 1
        rank, world, local_rank = par.init_from_env(args.dist_backend, force=forced)
    import torch.distributed as dist
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1
    if world != args.gpus or ranks_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} rank(s) (WORLD_SIZE={world})")
    dev = torch.device("cuda", (local_rank % n_dev) if world > 1 else 0)
    torch.cuda.set_device(dev)

    wl = syn.WORKLOADS[args.workload]
    P, W, H, deg = wl["P"], wl["width"], wl["height"], args.sh_degree
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
    for p in params:
        p.requires_grad_(True)
    means, shs, opac, scales, rots = params
    bg = torch.zeros(3, device=dev)
    views = make_views(W, H)
    setts = [settings_for(v, deg, bg, dev) for v in views]
    rasts = [R.GaussianRasterizer(s) for s in setts]
    ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10 + i)) for i in range(N_VIEWS)]
    bucket = par.GradBucket(params, active_dim1={1: (deg + 1) ** 2}) if par.exchanging() else None
    K = max(1, args.views_per_step)
    multi = R.GaussianRasterizerViews([setts[0]] * K) if K > 1 else None
    # Timed region: HIP events only around the dominant kernel (blend_backward: the roofline line) and the exchange
    # step; the other stages are timed in a second, untimed pass of the same steps — ten extra event records per step
    # would cost more host time than some stages take on the GPU.
    timer = R.StageTimer()
    timed = R.StageTimer(only=("blend_backward", "grad_allreduce"))
    R.set_stage_timer(timed)
    EVENT_EVERY = 4 if args.steps >= 8 else 1

    def train_step(step):
        for p in params:
            p.grad = None
        if multi is None:
            v = par.view_for(step, rank, world, N_VIEWS)
            means2D = torch.zeros_like(means, requires_grad=True)
            c, radii, d, a = rasts[v](means3D=means, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots)
            torch.autograd.backward([c, d, a], list(ups[v]))
        else:
            # K consecutive views of this rank in one node: forward x K, backward x K accumulating in the kernel
            vs = [(par.view_for(step, rank, world, 1 << 30) * K + k) % N_VIEWS for k in range(K)]
            multi.raster_settings_list = [setts[v] for v in vs]
            means2D = torch.zeros((K,) + tuple(means.shape), device=dev, requires_grad=True)
            outs = multi(means3D=means, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots)
            torch.autograd.backward([t for o in outs for t in (o[0], o[2], o[3])], [g for v in vs for g in ups[v]])
            radii = outs[-1][1]
        if bucket is not None:
            with timed("grad_allreduce"):
                bucket.reduce_grads(params)
        return radii

    gc.collect()
    gc.disable()                 # a 0.3 ms step creates no reference cycles worth a collector pause inside the timed region
    for i in range(args.warmup):
        train_step(i)
    timed.reset()                # (nothing slow between the warm-up and the timed steps: an idle gap of a few ms lets the
    par.barrier()                #  GPU fall back to its low-power clocks, which a short run then pays for)
    torch.cuda.synchronize()
    # median of >= 5 runs.  The K-step block is timed `--blocks` (5) times back to back, each bracketed by a
    # barrier + synchronize and reduced with MAX over ranks; `value` stays the FIRST block (the K steps directly behind the W
    # warm-up steps: comparable with every earlier round's line), `value_median / _min / _max` are over all blocks.
    block_dts = []
    for b in range(max(1, args.blocks)):
        t0 = time.perf_counter()
        for i in range(args.steps):
            # the dominant kernel is timed live, with events on its own dispatch packet — on every EVENT_EVERY-th step: a dispatch
            # that carries a completion signal costs the queue ~7 us in front of it and ~5 us behind (rocprofv3 trace of this loop,
            # profiles/README.md round 5: 11.8 us of a 280 us step), which is the MEASUREMENT's time, not the path's
            R.set_stage_timer(timed if (i % EVENT_EVERY == 0 and b == 0) else None)
            radii = train_step(args.warmup + b * args.steps + i)
        torch.cuda.synchronize()
        par.barrier()
        block_dts.append(par.max_over_ranks(time.perf_counter() - t0, dev))
    dt = block_dts[0]
    gc.enable()
    dominant_ms = timed.summary()
    # second pass, untimed: every stage
    R.set_stage_timer(timer)
    for i in range(args.steps):
        train_step(args.warmup + args.steps + i)
    stage_ms = timer.summary()
    stage_ms.update({k: v for k, v in dominant_ms.items()})       # the timed region's own numbers win

    # The same K steps once more on a device that has been busy for a while.  MI355X's power management is still raising
    # the clocks during the first ~0.3 s of load after idle (measured, profiles/README.md: 0.305 ms per step after 5 warm-up
    # steps, 0.285 after 50, 0.270 after 1000 — the kernels themselves run 13 % faster); `value` above is the protocol's
    # figure (W warm-up steps after process start), `sustained` is what a training run that lasts minutes sees.
    sustained, sustained_stage = None, {}
    if args.sustained_steps > 0:
        R.set_stage_timer(None)
        gc.collect()
        gc.disable()
        for i in range(args.sustained_steps):
            train_step(args.warmup + 2 * args.steps + i)
        par.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            train_step(args.warmup + 2 * args.steps + args.sustained_steps + i)
        torch.cuda.synchronize()
        par.barrier()
        dt_s = par.max_over_ranks(time.perf_counter() - t0, dev)
        # the dominant kernel at these clocks: the same K steps once more with its two events per step (not inside the
        # interval above: the event records cost the step 3-5 % — the main region pays that, its roofline line is measured live)
        timed_s = R.StageTimer(only=("blend_backward", "grad_allreduce"))
        R.set_stage_timer(timed_s)
        for i in range(5):
            train_step(i)
        timed_s.reset()
        for i in range(args.steps):
            train_step(5 + i)
        torch.cuda.synchronize()
        gc.enable()
        sustained_stage = timed_s.summary()
        sustained = {"value": round(world * K * args.steps / dt_s, 3), "unit": "iters/s",
                     "ms_per_step": round(dt_s / args.steps * 1e3, 4), "steps": args.steps,
                     "after_untimed_steps": args.warmup + (len(block_dts) + 1) * args.steps + args.sustained_steps,
                     "note": "same protocol, same K steps, device at its sustained clocks (the first ~0.3 s of load after "
                             "idle run at lower clocks); `value` is the figure after the W warm-up steps the caller asked for"}
        R.set_stage_timer(timer)

    # forward-only (render) leg, same workload
    dt_f, fs, stage_ms_f = forward_only(setts, params, args.steps, args.warmup, timer)
    dt_f = par.max_over_ranks(dt_f, dev)

    if rank != 0:
        par.shutdown()
        return
    V = int((radii > 0).sum().item())
    R_ = int(fs["num_rendered"])
    alg = algorithmic_bytes(P, V, R_, W, H, deg)
    kern = {k: v for k, v in stage_ms.items() if k in alg}
    dominant = max(kern, key=lambda k: kern[k][0])
    ms_per_step = dt / args.steps * 1e3
    out = {
        # one iteration of the reference = one view rendered and differentiated (train.py:143-170).  With K > 1 views per
        # autograd node a "step" is K views behind ONE optimizer step: the metric is then named for what it counts.
        "metric": "train_iters_per_sec" if K == 1 else "train_views_per_sec",
        "value": round(world * K * args.steps / dt, 3), "unit": "iters/s" if K == 1 else "views/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_median": round(world * K * args.steps / sorted(block_dts)[len(block_dts) // 2], 3),
        "value_min": round(world * K * args.steps / max(block_dts), 3),
        "value_max": round(world * K * args.steps / min(block_dts), 3),
        "value_blocks": {"blocks": len(block_dts), "steps_per_block": args.steps,
                         "ms_per_step": [round(d_ / args.steps * 1e3, 4) for d_ in block_dts],
                         "note": "K-step blocks timed back to back (barrier + synchronize around each, MAX over ranks); `value` = "
                                 "block 0, the K steps directly behind the W warm-up steps"},
        "optimizer_steps_per_sec": round(args.steps / dt, 3),
        "views_per_sec": round(world * K * args.steps / dt, 3),
        "value_is": ("cold: the K timed steps directly after the W warm-up steps the caller asked for (device still "
                     "raising its clocks when W is small); `sustained` = the same K steps on a warm device"),
        "config": {"workload": f"{args.workload}: {P} Gaussians, {W}x{H}, SH degree {deg}, fwd+bwd per view "
                               f"({WORKLOAD_CONFIG.get(args.workload, 'no BASELINE config of this shape')})",
                   "gaussians": P, "width": W, "height": H, "sh_degree": deg, "visible": V, "num_rendered": R_,
                   "views": N_VIEWS, "parallelism": f"dp{world}-over-views" if world > 1 else "single",
                   "ranks_seen": ranks_seen,
                   "rccl_version": rccl_version(),
                   "launched_by": ("bench.py's own launcher" if os.environ.get("SCG_BENCH_LAUNCHED") else
                                   "external launcher (torchrun)" if world > 1 else "single process"),
                   "grad_bucket_bytes": bucket.nbytes if bucket else 0,
                   "views_per_step": K,
                   "grad_bucket_bytes_per_view": (bucket.nbytes // K) if bucket else 0,
                   "dist_backend": (args.dist_backend or "nccl") if bucket is not None else None,
                   "exchange": "one all-reduce of the flat gradient arena per step (= per K views per rank), inside the timed region"
                               if bucket is not None else None,
                   # "arena": all-reduce of the gradient arena in place; "mixed" (SH degree < 3): the 44 B / Gaussian of the
                   # other gradients in place + the packed ACTIVE SH coefficients; "packed": pack / all-reduce / unpack
                   "exchange_path": getattr(bucket, "last_path", None) if bucket is not None else None,
                   "rccl_requested": {"NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"),
                                      "note": "what the ranks ASKED RCCL for (--rccl-algo / --rccl-proto or the caller's "
                                              "environment); null = RCCL's tuner decides; --rccl-debug prints what it picked"},
                   "host": "scgaussian_amd.single_gpu_host_setup(): autograd backward on the calling thread",
                   "per_tile_sort": ("inside the forward blend (tile_blend_forward_kernel: its time and 12 B / instance are in "
                                     "blend_forward)" if sort_in_blend(R_, W, H) else "tile_sort_kernel (binning stage)"),
                   "tile_order": ("cost recorded by the previous render of the same camera (ScgFrame.tile_cost_in; the "
                                  "bench cycles through its views like a training loop)" if R.TILE_COST_HINT
                                  else "list length (rasterizer.TILE_COST_HINT = False)")},
        "sustained": sustained,
        "ms_per_view": round(ms_per_step / K, 4),
        "render_mpix_per_sec": round(world * args.steps * W * H / dt_f / 1e6, 2),
        "render_ms": round(dt_f / args.steps * 1e3, 4),
        # SURVEY §8d "unit of work": time per tile instance and per Gaussian, one view per rank
        "unit_ns": {"train_per_instance": round(ms_per_step / K * 1e6 / max(R_, 1), 4),
                    "train_per_gaussian": round(ms_per_step / K * 1e6 / max(P, 1), 3),
                    "render_per_instance": round(dt_f / args.steps * 1e9 / max(R_, 1), 4),
                    "render_per_gaussian": round(dt_f / args.steps * 1e9 / max(P, 1), 3)},
        "stage_ms": {k: round(v[0], 4) for k, v in stage_ms.items()},
        "stage_ms_forward_only": {k: round(v[0], 4) for k, v in stage_ms_f.items()},
        "roofline": roofline_for(dominant, kern[dominant][0], alg[dominant], args.workload),
        "roofline_is": ("cold: measured live inside the K timed steps behind `value` (HIP events on the kernel's own dispatch "
                        f"packet, every {EVENT_EVERY}th step: {dominant_ms.get(dominant, (0, 0))[1]} launches timed); "
                        "`roofline_sustained` = warm twin"),
        # the same kernel in the `sustained` repetition (device at its sustained clocks)
        "roofline_sustained": (roofline_for(dominant, sustained_stage[dominant][0], alg[dominant], args.workload)
                               if sustained is not None and dominant in sustained_stage else None),
        "roofline_all": {k: roofline_for(k, kern[k][0], alg[k], args.workload) for k in kern},
    }
    # SURVEY §8e link budget next to the measured step, in EVERY line (N >= 1): what the exchange of this workload's gradient
    # arena (236 B per Gaussian at SH degree 3) would cost at 2 / 4 / 8 GPUs, and what K views per exchange buy
    arena_bytes = bucket.nbytes if bucket is not None else sum(int(p.numel()) * 4 for p in params)
    ar_ms = stage_ms.get("grad_allreduce", (0.0, 0))[0] if bucket is not None else 0.0
    compute_ms = max(ms_per_step - ar_ms, 1e-6)
    model = {}
    for n in (2, 4, 8):
        m = par.exchange_time_model(arena_bytes, n)
        model[str(n)] = {"ring_ms": round(m["ring_s"] * 1e3, 4), "all_links_ms": round(m["all_links_s"] * 1e3, 4),
                         "predicted_efficiency_ring": round(compute_ms / (compute_ms + m["ring_s"] * 1e3), 3),
                         "predicted_efficiency_all_links": round(compute_ms / (compute_ms + m["all_links_s"] * 1e3), 3)}
    # bytes exchanged per Gaussian by SH degree (the reference raises the degree every 1 000 iterations, train.py:129): 11 floats
    # of means / opacity / scales / rotations + 3 (deg + 1)^2 active SH coefficients — the `mixed` path of GradBucket.reduce_grads
    by_degree = {str(d): {"bytes_per_gaussian": 4 * (11 + 3 * (d + 1) ** 2), "bucket_bytes": 4 * (11 + 3 * (d + 1) ** 2) * P,
                          "ring_ms_8": round(par.exchange_time_model(4 * (11 + 3 * (d + 1) ** 2) * P, 8)["ring_s"] * 1e3, 4),
                          "all_links_ms_8": round(par.exchange_time_model(4 * (11 + 3 * (d + 1) ** 2) * P, 8)["all_links_s"] * 1e3, 4)}
                 for d in range(4)}
    out["exchange_model"] = {"bucket_bytes": arena_bytes, "views_per_step": K, "sh_degree": deg,
                             "bytes_exchanged_by_sh_degree": by_degree,
                             "measured_allreduce_ms": round(ar_ms, 4) if bucket is not None else None,
                             "compute_ms_per_step": round(compute_ms, 4), "by_world_size": model,
                             "note": "xGMI 7 links x 153 GB/s per GPU; the exchange is not overlappable with this step's "
                                     "compute (all gradients become final in the last kernel); efficiency = compute / "
                                     "(compute + exchange) per step of K views.  Which algorithm / protocol RCCL picks for "
                                     "this size on an 8-GPU xGMI node is its tuning model's decision and unmeasured here "
                                     "(no node): --rccl-debug prints it (NCCL_DEBUG=INFO) on the first real launch"}
    out["roofline_all"]["blend_forward(render)"] = roofline_for("blend_forward", stage_ms_f["blend_forward"][0],
                                                                alg["blend_forward_render"], args.workload)

    # The secondary legs must never cost the headline line: a failure in one of them is reported in its slot.
    def s3_leg():
        w3 = syn.WORKLOADS["S3"]
        P3, W3, H3 = w3["P"], w3["width"], w3["height"]
        sc3 = syn.make_scene(P3, W3, H3, seed=0).to(dev)
        setts3 = [settings_for(v, deg, bg, dev) for v in make_views(W3, H3)]
        p3 = [sc3.means3D, sc3.shs, sc3.opacities, sc3.scales, sc3.rotations]
        n3 = max(10, args.steps // 2)
        # (the scene above took seconds of CPU time with an idle GPU.)  COLD twin first: three untimed renders (one per view:
        # they establish the capacities, ~1 ms of load), then 10 timed renders — what a short process sees.  WARM: the leg
        # then warms the device up itself (300 renders: the main leg's W is the caller's, this one's is not; see `sustained`)
        dt3c, _, sm3c = forward_only(setts3, p3, 10, 3, timer)
        w3 = 300
        dt3, fs3, sm3 = forward_only(setts3, p3, n3, w3, timer)
        V3, R3 = int((fs3["radii"] > 0).sum().item()), int(fs3["num_rendered"])
        alg3 = algorithmic_bytes(P3, V3, R3, W3, H3, deg)
        Tn3 = ((W3 + 15) // 16) * ((H3 + 15) // 16)
        # SURVEY §8d's figure for the render forward, without the per-tile sort's 12 B / instance that the fused kernel also
        # moves; 20 B per pixel: a render nobody differentiates writes colour, depth and alpha only (28 with final_T, n_contrib)
        contract = 44 * R3 + 8 * Tn3 + 20 * W3 * H3

        def roof(ms):
            r = roofline_for("blend_forward", ms, alg3["blend_forward_render"], "S3")
            r["frac_contract"] = round(contract / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            r["contract_bytes"] = int(contract)
            r["contract"] = "SURVEY 8d render fwd: 44 R + 8 Tn + bytes_per_pixel W H, bytes_per_pixel = 20 (no backward state written)"
            r["frac_is"] = "bytes the fused sort + blend kernel moves (contract + 12 R of the per-tile sort) / time / 8 TB/s"
            # second yardstick: the kernel is vector-issue bound, so the best `frac_contract` THIS instruction
            # stream could reach is the contract's bytes over the time its vector instructions need at 100 % issue — counted
            # instructions (PMC pass of these kernels) x the measured cycles per instruction of their mix / 1 024 SIMDs
            v = r.get("valu") or {}
            if v.get("insts_valu") and v.get("cycles_per_inst_mix") and v.get("shader_clock_ghz"):
                floor_s = v["insts_valu"] * v["cycles_per_inst_mix"] / (1024 * v["shader_clock_ghz"] * 1e9)
                r["valu_floor_ms"] = round(floor_s * 1e3, 4)
                r["valu_floor_frac"] = round(contract / floor_s / 1e9 / HBM_PEAK_GBS, 4)
                r["target_reachable"] = (
                    f"north_star asks for frac_contract >= 0.60 (<= {contract / 0.6 / HBM_PEAK_GBS / 1e9 * 1e3:.4f} ms); at 100 % vector "
                    f"issue this kernel's {v['insts_valu'] / 1e6:.1f} M vector instructions take {floor_s * 1e3:.4f} ms = "
                    f"{r['valu_floor_frac']:.2f}: " + ("reachable" if r["valu_floor_frac"] >= 0.6 else
                    "NOT reachable with exact fp32 per-pixel evaluation of every (splat, quadrant) pair that passes the cull — the "
                    "kernel does not wait for HBM (traffic below the contract's bytes)"))
            else:
                r["valu_floor_frac"] = None
                r["target_reachable"] = "unknown in this run: no counters of these kernels (profiles/pmc_summary.json stale)"
            return r
        return {
            "workload": f"S3: {P3} Gaussians, {W3}x{H3}, SH degree {deg}, forward only (north-star roofline point)",
            "visible": V3, "num_rendered": R3, "renders": n3, "warmup_renders": w3, "render_ms": round(dt3 / n3 * 1e3, 4),
            "render_mpix_per_sec": round(n3 * W3 * H3 / dt3 / 1e6, 2),
            "stage_ms": {k: round(v[0], 4) for k, v in sm3.items()},
            "roofline": roof(sm3["blend_forward"][0]),
            "roofline_is": "warm: after this leg's own 300 warm-up renders (the S2 headline `roofline` beside it is cold)",
            "cold": {"renders": 10, "after_untimed_renders": 3, "render_ms": round(dt3c / 10 * 1e3, 4),
                     "stage_ms": {k: round(v[0], 4) for k, v in sm3c.items()},
                     "roofline": roof(sm3c["blend_forward"][0])},
            "roofline_all": {k: roofline_for(k, sm3[k][0], alg3["blend_forward_render" if k == "blend_forward" else k], "S3")
                             for k in sm3 if k in alg3},
        }

    def clustered_leg(name):
        """Training step on a NON-uniform scene of the headline shape (a share of the Gaussians in one screen region: long
        per-tile lists next to nearly empty tiles — what COLMAP-initialised scenes look like, reference
        scene/dataset_readers.py:145-249), same protocol as the small legs."""
        frac, spread = syn.CLUSTERED[name]
        scs = syn.make_clustered_scene(P, W, H, frac, spread, seed=0).to(dev)
        ps = [scs.means3D, scs.shs, scs.opacities, scs.scales, scs.rotations]
        for p_ in ps:
            p_.requires_grad_(True)
        ms_, shs_, op_, sc_, ro_ = ps
        rs = [R.GaussianRasterizer(settings_for(v, deg, bg, dev)) for v in make_views(W, H)]

        def st(i):
            for p_ in ps:
                p_.grad = None
            c_, _, d_, a_ = rs[i % 3](means3D=ms_, means2D=torch.zeros_like(ms_, requires_grad=True), opacities=op_,
                                      shs=shs_, scales=sc_, rotations=ro_)
            torch.autograd.backward([c_, d_, a_], list(ups[i % 3]))
        R.set_stage_timer(None)
        n = max(30, args.steps)
        for i in range(20):
            st(i)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for i in range(n):
            st(i)
        torch.cuda.synchronize()
        ms_step = (time.perf_counter() - t0_) / n * 1e3
        tm = R.StageTimer()
        R.set_stage_timer(tm)
        for i in range(12):
            st(i)
        stg = tm.summary()
        R.set_stage_timer(None)
        with torch.no_grad():
            fs_ = R.forward_stages(rs[0].raster_settings, ms_, op_, shs=shs_, scales=sc_, rotations=ro_)
            rng = fs_["ranges"].to(torch.int64)
            ln = (rng[:, 1] - rng[:, 0]).float()
            longest, mean_len = int(ln.max().item()), float(ln.mean().item())
            p99 = float(torch.quantile(ln, 0.99).item())
        return {"workload": f"{name}: {P} Gaussians, {W}x{H}, {frac:.0%} of them around one screen point (sigma {spread} NDC), "
                            f"fwd+bwd per view",
                "ms_per_step": round(ms_step, 4), "iters_per_sec": round(1e3 / ms_step, 1),
                "num_rendered_view0": int(fs_["num_rendered"]), "longest_list": longest, "mean_list": round(mean_len, 1),
                "p99_list": round(p99, 1),
                "stage_ms": {k: round(v[0], 4) for k, v in stg.items()}}

    def rccl_floor_leg():
        """What the exchange step costs BEFORE there is a peer: an RCCL all-reduce of the gradient arena's size in a world of
        one (process group on the RCCL backend, ReduceOp.AVG and SUM in place) — the floor under every N > 1 exchange, and
        what a training loop that all-reduces unconditionally would pay on one GPU (parallel.py skips the exchange when the
        world is one: _exchanging())."""
        import torch.distributed as dist
        if dist.is_initialized():
            return {"skipped": "a process group already exists: `grad_allreduce` in stage_ms is the measurement"}
        par.init_from_env("nccl", force=True)
        try:
            buf = torch.zeros(arena_bytes // 4, dtype=torch.float32, device=dev)
            res = {}
            for name, op in (("avg", dist.ReduceOp.AVG), ("sum", dist.ReduceOp.SUM)):
                for _ in range(5):
                    dist.all_reduce(buf, op=op)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    dist.all_reduce(buf, op=op)
                e1.record()
                torch.cuda.synchronize()
                res[f"allreduce_{name}_ms"] = round(e0.elapsed_time(e1) / 20, 4)
            res.update(bytes=arena_bytes, world=1, backend="nccl (RCCL %s)" % rccl_version(),
                       implied_gbs=round(2 * arena_bytes / (res["allreduce_avg_ms"] * 1e-3) / 1e9, 1),
                       note="in-place all-reduce in a world of one = one device-side pass over the buffer (read + write); "
                            "parallel.GradBucket.reduce_grads does not call it unless the world is > 1 or forced")
            return res
        finally:
            par.shutdown()
            par._EXCHANGE_AT_WORLD_ONE = False

    def guarded(fn):
        try:
            return fn()
        except Exception as e:                      # noqa: BLE001 - reported, not swallowed
            return {"error": f"{type(e).__name__}: {e}"}

    if world == 1 and not args.no_s3 and args.workload != "S3":
        out["s3_forward"] = guarded(s3_leg)

    def small_leg(name, k_views=1):
        """Training step of a smaller named workload (the reference's own regime: host-bound sizes), same protocol.
        k_views > 1: the three views of a step in ONE autograd node (GaussianRasterizerViews) — at these sizes the step is
        Python / autograd time, and the node pays it once per k views."""
        w = syn.WORKLOADS[name]
        Ps, Ws, Hs = w["P"], w["width"], w["height"]
        scs = syn.make_scene(Ps, Ws, Hs, seed=0).to(dev)
        ps = [scs.means3D, scs.shs, scs.opacities, scs.scales, scs.rotations]
        for p_ in ps:
            p_.requires_grad_(True)
        ms_, shs_, op_, sc_, ro_ = ps
        rs = [R.GaussianRasterizer(settings_for(v, deg, bg, dev)) for v in make_views(Ws, Hs)]
        us = [tuple(t.to(dev) for t in syn.make_upstream_grads(Ws, Hs, seed=10 + i)) for i in range(N_VIEWS)]
        tm = R.StageTimer()

        node = R.GaussianRasterizerViews([r_.raster_settings for r_ in rs][:k_views]) if k_views > 1 else None

        public_api = [False]

        def st(i):
            for p_ in ps:
                p_.grad = None
            if node is None:
                c_, _, d_, a_ = rs[i % 3](means3D=ms_, means2D=torch.zeros_like(ms_, requires_grad=True), opacities=op_,
                                          shs=shs_, scales=sc_, rotations=ro_)
                if public_api[0]:
                    torch.autograd.backward([c_, d_, a_], list(us[i % 3]))
                else:
                    backward_with_given_grads((c_, d_, a_), us[i % 3])
            else:
                m2 = torch.zeros((k_views,) + tuple(ms_.shape), device=dev, requires_grad=True)
                outs_ = node(means3D=ms_, means2D=m2, opacities=op_, shs=shs_, scales=sc_, rotations=ro_)
                ts_, gs_ = [t for o in outs_ for t in (o[0], o[2], o[3])], [g for v in range(k_views) for g in us[v]]
                if public_api[0]:
                    torch.autograd.backward(ts_, gs_)
                else:
                    backward_with_given_grads(ts_, gs_)
        R.set_stage_timer(None)
        n = max(50, args.steps)
        gc.collect()
        gc.disable()

        def median_of_three():                 # these legs share the process with much larger ones, and an allocator
            for i in range(20):                # reshuffle behind them can land in one repetition
                st(i)
            reps_ = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(n):
                    st(i)
                torch.cuda.synchronize()
                reps_.append((time.perf_counter() - t0_) / n * 1e3)
            return sorted(reps_)[1], reps_
        # PRIMARY: the public API — torch.autograd.backward(outputs, upstream gradients) — what any caller can reproduce
        public_api[0] = True
        ms_step, reps = median_of_three()
        # secondary: the same step with the gradients handed to the engine directly (no Python-side validation of them)
        public_api[0] = False
        ms_direct, _ = median_of_three()
        gc.enable()
        R.set_stage_timer(tm)
        for i in range(20):
            st(i)
        stg = tm.summary()
        R.set_stage_timer(None)
        return {"workload": f"{name}: {Ps} Gaussians, {Ws}x{Hs}, fwd+bwd per view" +
                            (f", {k_views} views per autograd node" if k_views > 1 else ""),
                "ms_per_step": round(ms_step, 4), "ms_per_view": round(ms_step / k_views, 4),
                "ms_per_step_repetitions": [round(r, 4) for r in reps],
                "backward_call": "torch.autograd.backward(outputs, fixed upstream gradients): the public API",
                "ms_per_step_engine_direct": round(ms_direct, 4) if ENGINE_DIRECT_OK else None,
                "engine_direct_is": ("the same step with the gradients handed to the autograd engine directly (private "
                                     "torch.autograd.Variable._execution_engine.run_backward: no Python-side validation of three "
                                     "image-sized gradients; a training loop calls loss.backward() on a scalar and never pays it)"
                                     if ENGINE_DIRECT_OK else "unavailable: this torch version's engine signature differs"),
                "iters_per_sec": round(k_views * 1e3 / ms_step, 1),
                "gpu_stage_sum_ms": round(sum(v[0] for v in stg.values()), 4),
                "stage_ms": {k: round(v[0], 4) for k, v in stg.items()}}

    def graph_leg(name, through_render=False):
        """The training step CAPTURED in a hipGraph and replayed (scgaussian_amd.graph_step.CapturedStep: the forward runs without
        its host read, the count of every replay is looked at before the next one) next to the eager step of the same process, and
        the eager step without the host read (rasterizer.no_host_read()).  One captured step per view, replayed in the training
        loop's order.  `through_render`: the step goes through render() on the reference's raw model (the model path)."""
        from scgaussian_amd import graph_step as gstep
        from scgaussian_amd import render as rmod
        w = syn.WORKLOADS[name]
        Ps, Ws, Hs = w["P"], w["width"], w["height"]
        scs = syn.make_scene(Ps, Ws, Hs, seed=0)
        us = [tuple(t.to(dev) for t in syn.make_upstream_grads(Ws, Hs, seed=10 + i)) for i in range(N_VIEWS)]
        if through_render:
            model = syn.make_raw_model(scs).to(dev).requires_grad_()
            model.active_sh_degree = deg
            ps = model.parameters()
            cams = [v.to(dev) for v in make_views(Ws, Hs)]
            pipe = rmod.PipelineParams()

            def fn(i):
                o = rmod.render(cams[i % 3], model, pipe, bg)
                torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], list(us[i % 3]))
                return o["render"]
        else:
            scd = scs.to(dev)
            ps = [scd.means3D, scd.shs, scd.opacities, scd.scales, scd.rotations]
            for p_ in ps:
                p_.requires_grad_(True)
            ms_, shs_, op_, sc_, ro_ = ps
            rs = [R.GaussianRasterizer(settings_for(v, deg, bg, dev)) for v in make_views(Ws, Hs)]

            def fn(i):
                c_, _, d_, a_ = rs[i % 3](means3D=ms_, means2D=torch.zeros_like(ms_, requires_grad=True), opacities=op_,
                                          shs=shs_, scales=sc_, rotations=ro_)
                torch.autograd.backward([c_, d_, a_], list(us[i % 3]))
                return c_

        def eager(i):
            for p_ in ps:
                p_.grad = None
            fn(i)
        R.set_stage_timer(None)
        n = max(50, args.steps)

        def median_of_three(step_fn):
            for i in range(20):
                step_fn(i)
            reps_ = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(n):
                    step_fn(i)
                torch.cuda.synchronize()
                reps_.append((time.perf_counter() - t0_) / n * 1e3)
            return sorted(reps_)[1], [round(r, 4) for r in reps_]
        gc.collect()
        gc.disable()
        try:
            eager_ms, eager_reps = median_of_three(eager)
            with R.no_host_read():
                nohost_ms, nohost_reps = median_of_three(eager)
            torch.cuda.synchronize()
            stats0 = R.settle_counts()
            steps_ = [gstep.CapturedStep(lambda i=i: fn(i), params=ps) for i in range(3)]
            graph_ms, graph_reps = median_of_three(lambda i: steps_[i % 3].replay())
            torch.cuda.synchronize()
            stats1 = R.settle_counts()
            res = {"workload": f"{name}: {Ps} Gaussians, {Ws}x{Hs}, SH degree {deg}, fwd+bwd per view" +
                               (" through render() on the reference's raw model (model path)" if through_render else ""),
                   "eager_ms": round(eager_ms, 4), "eager_repetitions": eager_reps,
                   "eager_no_host_read_ms": round(nohost_ms, 4), "eager_no_host_read_repetitions": nohost_reps,
                   "graph_ms": round(graph_ms, 4), "graph_repetitions": graph_reps,
                   "graph_iters_per_sec": round(1e3 / graph_ms, 1),
                   "captured_forwards": sum(len(s_.words) for s_ in steps_),
                   "replays": sum(s_.replays for s_ in steps_), "overflows": sum(s_.overflows for s_ in steps_),
                   "recaptures": sum(s_.recaptures for s_ in steps_),
                   "no_host_read_renders": stats1["renders"] - stats0["renders"] + 0,
                   "eager_no_host_read_overflows": stats0["overflows"],
                   "is": "one hipGraph per view (forward + backward, six kernels), replayed in the training loop's order; static "
                         "parameters (the moving-scene counters are in `full_iteration`)"}
            for s_ in steps_:
                s_.close()
            for p_ in ps:
                p_.grad = None
            return res
        finally:
            gc.enable()

    if world == 1 and not args.no_small and args.workload == "S2":
        out["small_workloads"] = {n: guarded(lambda n=n: small_leg(n)) for n in ("S1", "S2r8")}
        out["small_workloads"]["S1_3views_per_node"] = guarded(lambda: small_leg("S1", 3))
    if world == 1 and not args.no_graph and args.workload == "S2":
        out["captured_step"] = {"S1_graph": guarded(lambda: graph_leg("S1")), "S2_graph": guarded(lambda: graph_leg("S2")),
                                "S1_graph_render": guarded(lambda: graph_leg("S1", True)),
                                "S2_graph_render": guarded(lambda: graph_leg("S2", True))}

    if world == 1 and not args.no_render_glue and args.workload == "S2":
        R.set_stage_timer(None)
        out["render_glue"] = {f"{n}_deg{d}": guarded(lambda n=n, d=d: render_glue_leg(n, d, dev, args.steps))
                              for n, d in (("S2", deg), ("S2", 1), ("S1", deg), ("S1", 0))}
    if world == 1 and not args.no_by_degree:
        out["by_sh_degree"] = guarded(lambda: by_sh_degree_leg(P, W, H, dev, args.steps, args.workload))

    if world == 1 and not args.no_clustered and args.workload == "S2":
        out["clustered"] = {n: guarded(lambda n=n: clustered_leg(n)) for n in ("clustered30", "clustered60")}

    if world == 1 and not args.no_full_iteration:
        R.set_stage_timer(None)
        out["full_iteration"] = guarded(lambda: full_iteration_leg(P, W, H, deg, dev, max(10, args.steps // 2)))

    if world == 1 and not args.no_moving_scene and args.workload == "S2":
        out["moving_scene"] = guarded(lambda: moving_scene_leg(P, W, H, deg, dev))

    if world == 1 and bucket is None and not args.no_rccl_floor:
        out["exchange_model"]["rccl_world_of_one"] = guarded(rccl_floor_leg)

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = guarded(lambda: cpu_baseline_guarded(P, W, H, deg, args.cpu_tile_stride))
    else:
        out["cpu_baseline"] = None
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()
    par.shutdown()


if __name__ == "__main__":
    main()
