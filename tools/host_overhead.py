#!/usr/bin/env python3
"""How much of the training step is host (Python + launch) time?  Times the S2 step loop twice: wall time with a
final synchronize, and the host time spent enqueueing (no sync).  If the two are close the step is host-bound."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scgaussian_amd import synthetic as syn, rasterizer as R

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "S2"
wl = dict(P=2000, width=128, height=96) if name == "tiny" else syn.WORKLOADS[name]   # tiny: GPU work ~ launch floors
sc = syn.make_scene(wl["P"], wl["width"], wl["height"])
import math
cam = syn.default_camera(wl["width"], wl["height"])
st = R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                     torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                     cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
rast = R.GaussianRasterizer(st)
params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
means, opac, shs, scales, rots = params
ups = [u.to(dev) for u in syn.make_upstream_grads(cam.image_width, cam.image_height)]

def step():
    for p in params:
        p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rast(means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], ups)

for _ in range(5):
    step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/N:.3f} ms/step   wall {1e3*(t2-t0)/N:.3f} ms/step   tail drain {1e3*(t2-t1):.3f} ms")
# forward only
with torch.no_grad():
    for _ in range(3):
        rast(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        rast(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales, rotations=rots)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"forward: host enqueue {1e3*(t1-t0)/N:.3f} ms   wall {1e3*(t2-t0)/N:.3f} ms")
