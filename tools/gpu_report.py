#!/usr/bin/env python3
"""Exploratory parity report: HIP path vs CPU oracle on seeded scenes, printing error statistics for every
intermediate and gradient (no asserts).  Run on the GPU box:  python tools/gpu_report.py [P W H deg]"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_rasterizer as orc            # noqa: E402  (checker only)
from scgaussian_amd import synthetic as syn           # noqa: E402
from scgaussian_amd import rasterizer as R            # noqa: E402


def stats(name, a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-30)
    rel_el = d / (np.abs(b) + 1e-4 * scale)
    print(f"  {name:14s} max|ref|={scale:.3e} max|d|={d.max():.3e} nrm={d.max()/scale:.2e} "
          f"l2rel={np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30):.2e} frac(el_rel>1e-3)={np.mean(rel_el>1e-3):.2e}")


def run(P, W, H, deg, cam, bg, mod=1.0, seed=0):
    print(f"== P={P} {W}x{H} deg={deg} bg={bg} mod={mod}")
    sc = syn.make_scene(P, W, H, seed=seed)
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    bg_t = torch.tensor(bg, dtype=torch.float32)
    st_cpu = orc.Settings(H, W, tanx, tany, bg_t, mod, cam.world_view_transform, cam.full_proj_transform, deg,
                          cam.camera_center, False, False)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, torch.zeros(P, 3), sc.shs, sc.opacities, sc.scales, sc.rotations)]
    m, m2, sh, op, s, r = leaves
    t0 = time.time()
    c, rad, d, a, aux = orc.rasterize(m, m2, op, st_cpu, shs=sh, scales=s, rotations=r, return_aux=True)
    dc, dd, da = syn.make_upstream_grads(W, H)
    ((c * dc).sum() + (d * dd).sum() + (a * da).sum()).backward()
    print(f"  oracle fwd+bwd {time.time()-t0:.2f}s  R={aux['binning']['num_rendered']} V={(rad>0).sum().item()}")

    dev = torch.device("cuda")
    camd = cam.to(dev)
    st = R.GaussianRasterizationSettings(H, W, tanx, tany, bg_t.to(dev), mod, camd.world_view_transform,
                                         camd.full_proj_transform, deg, camd.camera_center, False, False)
    g = [t.detach().to(dev).requires_grad_(True) for t in (sc.means3D, torch.zeros(P, 3), sc.shs, sc.opacities, sc.scales, sc.rotations)]
    gm, gm2, gsh, gop, gs, gr = g
    fs = R.forward_stages(st, gm.detach(), gop.detach(), shs=gsh.detach(), scales=gs.detach(), rotations=gr.detach(), want_keys=True)
    torch.cuda.synchronize()
    pre, b = aux["pre"], aux["binning"]
    print("  R gpu", fs["num_rendered"], "radii equal", torch.equal(fs["radii"].cpu(), rad),
          "n_mismatch", (fs["radii"].cpu() != rad).sum().item())
    off = fs["point_offsets"].cpu().numpy().astype(np.uint32)
    print("  offsets equal", np.array_equal(off, b["point_offsets"]))
    if fs["num_rendered"] == b["num_rendered"]:
        k = fs["keys_sorted"].cpu().numpy().view(np.uint64)
        print("  keys equal", np.array_equal(k, b["keys_sorted"]), " point_list equal",
              np.array_equal(fs["point_list"].cpu().numpy().astype(np.uint32), b["point_list"]),
              " ranges equal", np.array_equal(fs["ranges"].cpu().numpy().astype(np.uint32), b["ranges"]))
    sp = fs["splats"].cpu().numpy()
    vis = rad.numpy() > 0
    stats("xy", sp[vis, 0:2], pre["xy"].detach().numpy()[vis])
    print("   xy bit-equal:", np.array_equal(sp[vis, 0:2], pre["xy"].detach().numpy()[vis]),
          " depth bit-equal:", np.array_equal(sp[vis, 11], pre["depth"].detach().numpy()[vis]),
          " conic bit-equal:", np.array_equal(sp[vis, 2:5], pre["conic"].detach().numpy()[vis]),
          " rgb bit-equal:", np.array_equal(sp[vis, 8:11], pre["rgb"].detach().numpy()[vis]))
    stats("conic", sp[vis, 2:5], pre["conic"].detach().numpy()[vis])
    stats("rgb", sp[vis, 8:11], pre["rgb"].detach().numpy()[vis])
    stats("color", fs["color"].cpu().numpy(), c.detach().numpy())
    stats("depth", fs["depth"].cpu().numpy(), d.detach().numpy())
    stats("alpha", fs["alpha"].cpu().numpy(), a.detach().numpy())
    stats("final_T", fs["final_T"].cpu().numpy(), aux["final_T"].numpy())
    nc = fs["n_contrib"].cpu().numpy(); nco = aux["n_contrib"].numpy()
    print("  n_contrib mismatches", int((nc != nco).sum()), "of", nc.size)

    rast = R.GaussianRasterizer(st)
    t0 = time.time()
    gc, grad_, gd, ga = rast(means3D=gm, means2D=gm2, opacities=gop, shs=gsh, scales=gs, rotations=gr)
    ((gc * dc.to(dev)).sum() + (gd * dd.to(dev)).sum() + (ga * da.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    print(f"  hip fwd+bwd (incl. torch glue, first call) {time.time()-t0:.3f}s")
    for name, cpu_t, gpu_t in zip(("means3D", "means2D", "shs", "opacities", "scales", "rotations"), leaves, g):
        stats("d_" + name, gpu_t.grad.cpu().numpy(), cpu_t.grad.numpy())


if __name__ == "__main__":
    if len(sys.argv) >= 5:
        P, W, H, deg = map(int, sys.argv[1:5])
        run(P, W, H, deg, syn.default_camera(W, H), (0.0, 0.0, 0.0))
    else:
        run(2000, 200, 120, 3, syn.default_camera(200, 120), (0.0, 0.0, 0.0))
        run(3000, 250, 130, 2, syn.orbit_camera(250, 130, 15.0, -8.0, 7.5), (1.0, 1.0, 1.0), mod=0.8, seed=3)
        run(10000, 256, 256, 3, syn.default_camera(256, 256), (0.0, 0.0, 0.0))
