"""How long is the geometry + histogram kernel without the SH records (colors_precomp path = the kernel a split-off colour pass
would leave)?  Run under rocprofv3 --kernel-trace --stats."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scgaussian_amd import rasterizer as R, synthetic as syn
import bench
dev = torch.device("cuda", 0)
for name in sys.argv[1:]:
    w = syn.WORKLOADS[name]
    P, W, H = w["P"], w["width"], w["height"]
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    sett = bench.settings_for(syn.default_camera(W, H), 3, torch.zeros(3, device=dev), dev)
    rast = R.GaussianRasterizer(sett)
    rgb = torch.rand(P, 3, device=dev)
    with torch.no_grad():
        for _ in range(30):
            rast(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
        for _ in range(30):
            rast(means3D=sc.means3D, means2D=torch.zeros_like(sc.means3D), opacities=sc.opacities, colors_precomp=rgb, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
