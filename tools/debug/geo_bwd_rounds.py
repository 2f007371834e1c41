"""geometry_backward stage time against the Gaussian count around the round boundaries of its workgroups (S2 image)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scgaussian_amd import synthetic as syn, rasterizer as R
dev = torch.device("cuda", 0)
W, H = 1008, 756
cam = syn.default_camera(W, H)
for P in [int(a) for a in sys.argv[1:]] or [180000, 196608, 197632, 200000, 212992, 214016, 230000]:
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
    for p in params: p.requires_grad_(True)
    means, shs, opac, scales, rots = params
    camd = cam.to(dev)
    st = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                         camd.world_view_transform, camd.full_proj_transform, 3, camd.camera_center, False, False)
    rast = R.GaussianRasterizer(st)
    ups = [t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10)]
    tm = R.StageTimer()
    def step():
        for p in params: p.grad = None
        c, r, d, a = rast(means3D=means, means2D=torch.zeros_like(means, requires_grad=True), opacities=opac, shs=shs, scales=scales, rotations=rots)
        torch.autograd.backward([c, d, a], ups)
    for _ in range(30): step()
    R.set_stage_timer(tm)
    for _ in range(40): step()
    R.set_stage_timer(None)
    s = tm.summary()
    print(P, "geometry_backward %.1f us" % (s["geometry_backward"][0] * 1e3), "geometry_forward %.1f" % (s["geometry_forward"][0] * 1e3), flush=True)
