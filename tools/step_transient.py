"""Wall time of the first training steps of a fresh process, one by one (each step synchronised): how long until the
steady state bench.py's short default run (--steps 20 --warmup 5) is measured in?"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scgaussian_amd
from scgaussian_amd import rasterizer as R, synthetic as syn
scgaussian_amd.single_gpu_host_setup()
dev = torch.device("cuda", 0)
w = syn.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "S2"]
P, W, H = w["P"], w["width"], w["height"]
sc = syn.make_scene(P, W, H, seed=0).to(dev)
params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
for p in params: p.requires_grad_(True)
means, shs, opac, scales, rots = params
bg = torch.zeros(3, device=dev)
views = [syn.default_camera(W, H), syn.orbit_camera(W, H, 6.0, 0.0, 7.0), syn.orbit_camera(W, H, -6.0, 2.0, 7.0)]
def sett(cam):
    c = cam.to(dev)
    return R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0,
                                           c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False, False)
rasts = [R.GaussianRasterizer(sett(v)) for v in views]
ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10 + i)) for i in range(3)]
torch.cuda.synchronize()
ts = []
for i in range(40):
    t0 = time.perf_counter()
    for p in params: p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rasts[i % 3](means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], list(ups[i % 3]))
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (synchronised):", " ".join(f"{t:.3f}" for t in ts))
