import sys, os, math, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scgaussian_amd import synthetic as syn, rasterizer as R
dev = torch.device("cuda", 0)
P, W, H, lsm = 50_000, 1008, 756, -1.5
sc = syn.make_scene(P, W, H, seed=0, log_scale_mean=lsm)
cam = syn.default_camera(W, H)
st = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                     cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
rast = R.GaussianRasterizer(st)
params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
means, opac, shs, scales, rots = params
ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]
orig = R.forward_fused
def traced(*a, **k):
    t0 = time.perf_counter(); out = orig(*a, **k); t1 = time.perf_counter()
    print("   forward_fused", "None" if out is None else ("R", out[4]["num_rendered"], "cap", out[4]["cap"]), f"{(t1-t0)*1e3:.3f} ms")
    return out
R.forward_fused = traced
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in params: p.grad = None
    c, radii, d, a = rast(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales, rotations=rots)
    t1 = time.perf_counter()
    torch.autograd.backward([c, d, a], ups)
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(i, f"fwd host {(t1-t0)*1e3:.3f} bwd host {(t2-t1)*1e3:.3f} total {(t3-t0)*1e3:.3f} ms", "mem", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
