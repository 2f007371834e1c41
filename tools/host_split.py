import sys, os, math, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scgaussian_amd import synthetic as syn, rasterizer as R, _lib
dev = torch.device("cuda", 0)
wl = dict(P=2000, width=128, height=96)
sc = syn.make_scene(wl["P"], wl["width"], wl["height"])
cam = syn.default_camera(wl["width"], wl["height"])
st = R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                     torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                     cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
rast = R.GaussianRasterizer(st)
params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
means, opac, shs, scales, rots = params
ups = [u.to(dev) for u in syn.make_upstream_grads(cam.image_width, cam.image_height)]
lib = _lib.load()
acc = {}
class Wrap:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
        return g
_lib._LIB = Wrap(lib) if hasattr(_lib, "_LIB") else None
orig_load = _lib.load
w = Wrap(lib)
_lib.load = lambda: w
R._lib.load = _lib.load
tf = [0.0]; tb = [0.0]
ofs, obs = R.forward_stages, R.backward_stages
def fs(*a, **k):
    t0 = time.perf_counter(); r = ofs(*a, **k); tf[0] += time.perf_counter() - t0; return r
def bs(*a, **k):
    t0 = time.perf_counter(); r = obs(*a, **k); tb[0] += time.perf_counter() - t0; return r
R.forward_stages, R.backward_stages = fs, bs
def step():
    for p in params: p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rast(means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], ups)
for _ in range(20): step()
torch.cuda.synchronize(); acc.clear(); tf[0] = tb[0] = 0
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("per step us: total %.1f  forward_stages %.1f  backward_stages %.1f" % ((t1 - t0) / N * 1e6, tf[0] / N * 1e6, tb[0] / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-32s %.1f us" % (k, v / N * 1e6))
