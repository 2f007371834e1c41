"""The oracle's radius chain for one Gaussian of S3, evaluated by torch on the whole batch and on a slice of one: which
operation depends on the host CPU's vector path?"""
import sys, os, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_utils as pu
from oracle import torch_rasterizer as orc
from scgaussian_amd import synthetic as syn

w = syn.WORKLOADS["S3"]; P, W, H = w["P"], w["width"], w["height"]
sc = syn.make_scene(P, W, H, seed=0); cam = syn.default_camera(W, H)
st = pu.oracle_settings(cam, 3, (0.2, 0.1, 0.3))
i = 28960
print(torch.__config__.show().split("CPU capability usage")[1][:20], "threads", torch.get_num_threads())
os.system("grep -m1 'model name' /proc/cpuinfo")


def chain(m, s, r):
    out = {}
    focal_x, focal_y, limx, limy = orc.host_scalars(W, H, st.tanfovx, st.tanfovy)
    V = st.viewmatrix.reshape(16).to(torch.float32)
    x, y, z = m[:, 0], m[:, 1], m[:, 2]
    tx = V[0] * x + V[4] * y + V[8] * z + V[12]
    ty = V[1] * x + V[5] * y + V[9] * z + V[13]
    tz = V[2] * x + V[6] * y + V[10] * z + V[14]
    cov3D = orc.cov3d_from_scale_rot(s, r, 1.0)
    c_xx, c_xy, c_xz, c_yy, c_yz, c_zz = cov3D.unbind(1)
    txtz = tx / tz; tytz = ty / tz
    t_x = torch.clamp(txtz, -limx, limx) * tz
    t_y = torch.clamp(tytz, -limy, limy) * tz
    tz2 = tz * tz
    J00 = torch.full_like(tz, focal_x) / tz
    J02 = -(focal_x * t_x) / tz2
    J11 = torch.full_like(tz, focal_y) / tz
    J12 = -(focal_y * t_y) / tz2
    T00 = J00 * V[0] + J02 * V[2]; T01 = J00 * V[4] + J02 * V[6]; T02 = J00 * V[8] + J02 * V[10]
    T10 = J11 * V[1] + J12 * V[2]; T11 = J11 * V[5] + J12 * V[6]; T12 = J11 * V[9] + J12 * V[10]
    u0 = c_xx * T00 + c_xy * T01 + c_xz * T02; u1 = c_xy * T00 + c_yy * T01 + c_yz * T02; u2 = c_xz * T00 + c_yz * T01 + c_zz * T02
    v0 = c_xx * T10 + c_xy * T11 + c_xz * T12; v1 = c_xy * T10 + c_yy * T11 + c_yz * T12; v2 = c_xz * T10 + c_yz * T11 + c_zz * T12
    A = T00 * u0 + T01 * u1 + T02 * u2 + 0.3
    B = T00 * v0 + T01 * v1 + T02 * v2
    C = T10 * v0 + T11 * v1 + T12 * v2 + 0.3
    det = A * C - B * B
    mid = 0.5 * (A + C)
    inner = mid * mid - det
    sq = torch.sqrt(torch.clamp_min(inner, 0.1))
    lam1 = mid + sq
    s2 = torch.sqrt(lam1)
    rad = 3.0 * s2
    for k, v in dict(tx=tx, ty=ty, tz=tz, cxx=c_xx, cxy=c_xy, cxz=c_xz, cyy=c_yy, cyz=c_yz, czz=c_zz, txtz=txtz, t_x=t_x, J00=J00, J02=J02,
                     J11=J11, J12=J12, T00=T00, T01=T01, T02=T02, T10=T10, T11=T11, T12=T12, u0=u0, u1=u1, u2=u2, v0=v0, v1=v1,
                     v2=v2, A=A, B=B, C=C, det=det, mid=mid, inner=inner, sq=sq, lam1=lam1, s2=s2, rad=rad).items():
        out[k] = v
    return out


full = chain(sc.means3D, sc.scales, sc.rotations)
one = chain(sc.means3D[i:i + 1], sc.scales[i:i + 1], sc.rotations[i:i + 1])
hx = lambda t: hex(struct.unpack("I", struct.pack("f", float(t)))[0])
for k in full:
    a, b = full[k][i], one[k][0]
    print(f"{k:5s} batch {hx(a)} single {hx(b)} {'   <-- differs' if hx(a) != hx(b) else ''}")
