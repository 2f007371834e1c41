"""What is in a full training iteration besides the rasterizer?  (bench.full_iteration_leg's iteration, 40 x per optimizer variant,
under tools/kstats.sh; then host time per iteration without the GPU in the way: wall clock of the enqueue loop vs the synchronised loop)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import scgaussian_amd
from scgaussian_amd import rasterizer as R, synthetic as syn, losses
from scgaussian_amd.match_loss import match_loss_from_depth
import bench
scgaussian_amd.single_gpu_host_setup()
dev = torch.device("cuda", 0)
w = syn.WORKLOADS["S2"]; P, W, H = w["P"], w["width"], w["height"]
sc = syn.make_scene(P, W, H, seed=0).to(dev)
views = bench.make_views(W, H)
bg = torch.zeros(3, device=dev)
setts = [bench.settings_for(v, 3, bg, dev) for v in views]
params = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
means, shs, opac, scales, rots = params
with torch.no_grad():
    gt = [R.GaussianRasterizer(s)(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales, rotations=rots) for s in setts]
pairs = bench._synthetic_match_pairs(views, gt[0][2], 2000, dev)
targets = [(g_[0] + 0.05 * torch.randn_like(g_[0])).clamp(0, 1) for g_ in gt]
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
opt = torch.optim.Adam(params, lr=1e-4, eps=1e-15, fused=(mode == "fused"))
def iteration(i, with_opt=True, with_loss=True):
    v = i % len(setts)
    means2D = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = R.GaussianRasterizer(setts[v])(means3D=means, means2D=means2D, opacities=opac, shs=shs, scales=scales, rotations=rots)
    if with_loss:
        loss = losses.image_loss(c, targets[v], 0.2)
        if v == 0:
            loss = loss + 0.3 * match_loss_from_depth(d, pairs, float(W), float(H))
    else:
        loss = c.sum()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if with_opt:
        opt.step()
for name, kw in (("full", {}), ("no optimizer", dict(with_opt=False)), ("no optimizer, sum() loss", dict(with_opt=False, with_loss=False))):
    for i in range(6): iteration(i, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60): iteration(i, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:8s} {name:28s} enqueue {(t1 - t0) / 60 * 1e3:.4f} ms / iteration, with the GPU {(t2 - t0) / 60 * 1e3:.4f} ms")
