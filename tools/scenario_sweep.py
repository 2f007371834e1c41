import sys, os, math, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scgaussian_amd import synthetic as syn, rasterizer as R
dev = torch.device("cuda", 0)
for name, P, W, H, lsm in (("4K image", 200_000, 3840, 2160, -4.0), ("tiny image, 1M", 1_000_000, 256, 256, -4.0), ("8K image (global-sort fallback)", 100_000, 7680, 4320, -4.0), ("big splats", 50_000, 1008, 756, -1.5)):
    sc = syn.make_scene(P, W, H, seed=0, log_scale_mean=lsm)
    cam = syn.default_camera(W, H)
    st = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                         1.0, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                         cam.camera_center.to(dev), False, False)
    rast = R.GaussianRasterizer(st)
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    means, opac, shs, scales, rots = params
    ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]
    def step():
        for p in params: p.grad = None
        c, radii, d, a = rast(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales, rotations=rots)
        torch.autograd.backward([c, d, a], ups)
    for _ in range(3): step()
    timer = R.StageTimer(); R.set_stage_timer(timer)
    for _ in range(5): step()
    stages = {k: round(v[0] * 1e3) for k, v in timer.summary().items()}
    R.set_stage_timer(None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    fs = R.forward_stages(st, means.detach(), opac.detach(), shs=shs.detach(), scales=scales.detach(), rotations=rots.detach())
    rng = fs["ranges"].cpu().numpy().astype("int64"); ln = rng[:, 1] - rng[:, 0]
    print(f"{name}: P {P} {W}x{H}  step {dt*1e3:.3f} ms  R {fs['num_rendered']}  list mean {ln.mean():.0f} max {ln.max()}  stages(us) {stages}")
    del params, means, opac, shs, scales, rots, ups, fs
    torch.cuda.empty_cache()
