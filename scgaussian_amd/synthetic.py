"""Seeded synthetic scenes and camera records for parity tests and bench.py.

Generator spec: BASELINE.md §2 / SURVEY.md §8d.  Everything is generated on the CPU with a
``torch.Generator`` (seed 0 is what the reference's ``safe_state`` uses,
utils/general_utils.py:136-138) and uploaded bit-identically, so the CPU oracle and the HIP
path see the same fp32 inputs.

Camera matrices follow the reference's conventions (scene/cameras.py:54-63,
utils/graphics_utils.py:38-71): ``viewmatrix`` is world->camera TRANSPOSED (row-vector
convention), ``projmatrix`` is ``viewmatrix @ P^T``, camera centre is
``inverse(viewmatrix)[3, :3]``.  These are this package's own formulas; tests/golden pins them
against the reference's helpers.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch


class CameraRecord(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # (4,4) = w2c^T
    full_proj_transform: torch.Tensor    # (4,4) = w2c^T @ P^T
    camera_center: torch.Tensor          # (3,)
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device))


def world_to_view(R: np.ndarray, T: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """w2c 4x4 (fp32).  R is camera->world rotation (stored transposed, COLMAP style), T the
    w2c translation — same meaning as utils/graphics_utils.py:38-49."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(T, dtype=np.float64)
    Rt[3, 3] = 1.0
    translate = np.asarray(translate, dtype=np.float64)
    if np.any(translate != 0.0) or scale != 1.0:
        C2W = np.linalg.inv(Rt)
        C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
        Rt = np.linalg.inv(C2W)
    return Rt.astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """Perspective matrix P (column-vector convention, z_sign=+1), utils/graphics_utils.py:51-71."""
    tan_y = math.tan(fovY / 2)
    tan_x = math.tan(fovX / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(R: np.ndarray, T: np.ndarray, FoVx: float, FoVy: float, width: int, height: int,
                znear: float = 0.01, zfar: float = 100.0) -> CameraRecord:
    wv = torch.from_numpy(world_to_view(R, T)).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, FoVx, FoVy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return CameraRecord(int(height), int(width), float(FoVx), float(FoVy), wv, full, center, znear, zfar)


def default_camera(width: int, height: int, fovy_deg: float = 50.0) -> CameraRecord:
    """R = I, T = 0, looking down +z."""
    fovy = math.radians(fovy_deg)
    fovx = 2.0 * math.atan(math.tan(fovy / 2) * width / height)
    return make_camera(np.eye(3), np.zeros(3), fovx, fovy, width, height)


def orbit_camera(width: int, height: int, yaw_deg: float, pitch_deg: float, dist: float,
                 target=(0.0, 0.0, 7.0), fovy_deg: float = 50.0) -> CameraRecord:
    """A rotated/translated camera looking at ``target`` — exercises every view-matrix entry."""
    fovy = math.radians(fovy_deg)
    fovx = 2.0 * math.atan(math.tan(fovy / 2) * width / height)
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    tgt = np.asarray(target, dtype=np.float64)
    fwd = np.array([math.sin(yaw) * math.cos(pitch), math.sin(pitch), math.cos(yaw) * math.cos(pitch)])
    pos = tgt - dist * fwd
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_R = np.stack([right, down, fwd], axis=1)          # columns = camera axes in world
    w2c_R = c2w_R.T
    T = -w2c_R @ pos
    return make_camera(c2w_R, T, fovx, fovy, width, height)


class Scene(NamedTuple):
    means3D: torch.Tensor     # (P,3)
    scales: torch.Tensor      # (P,3) activated
    rotations: torch.Tensor   # (P,4) normalised (r,x,y,z)
    opacities: torch.Tensor   # (P,1) activated
    shs: torch.Tensor         # (P,16,3)

    def to(self, device):
        return Scene(*[t.to(device) for t in self])


def make_scene(P: int, width: int, height: int, seed: int = 0, fovy_deg: float = 50.0,
               log_scale_mean: float = -4.0, log_scale_std: float = 0.5) -> Scene:
    g = torch.Generator().manual_seed(seed)
    fovy = math.radians(fovy_deg)
    tanfovy = math.tan(fovy / 2)
    tanfovx = tanfovy * width / height
    z = torch.rand(P, generator=g) * 10.0 + 2.0
    x = z * tanfovx * (torch.rand(P, generator=g) * 2.2 - 1.1)
    y = z * tanfovy * (torch.rand(P, generator=g) * 2.2 - 1.1)
    means = torch.stack([x, y, z], 1).contiguous()
    scales = torch.exp(torch.randn(P, 3, generator=g) * log_scale_std + log_scale_mean)
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 2.0)
    dc = torch.rand(P, 1, 3, generator=g) * 3.0 - 1.5
    rest = torch.randn(P, 15, 3, generator=g) * 0.1
    shs = torch.cat([dc, rest], 1).contiguous()
    return Scene(means, scales, rot, opac, shs)


def make_clustered_scene(P: int, width: int, height: int, frac: float, spread: float, seed: int = 0,
                         centre=(0.25, -0.2), fovy_deg: float = 50.0) -> Scene:
    """A NON-uniform scene: the first `frac` of make_scene's Gaussians are moved into one screen region (normal
    distribution of width `spread` in NDC around `centre`, depths 4..8) — long per-tile lists next to nearly empty tiles,
    what scenes initialised from COLMAP points and match rays look like (reference scene/dataset_readers.py:145-249,
    scene/gaussian_model.py:362-468).  frac = 0 is make_scene itself."""
    sc = make_scene(P, width, height, seed=seed, fovy_deg=fovy_deg)
    n = int(P * frac)
    if n:
        g = torch.Generator().manual_seed(seed + 5)
        z = torch.rand(n, generator=g) * 4.0 + 4.0
        tany = math.tan(math.radians(fovy_deg) / 2)
        tanx = tany * width / height
        x = z * tanx * (centre[0] + spread * torch.randn(n, generator=g))
        y = z * tany * (centre[1] + spread * torch.randn(n, generator=g))
        sc.means3D[:n] = torch.stack([x, y, z], 1)
    return sc


# the clustered variants the bench and the parity suite use: name -> (share of the Gaussians, width in NDC)
CLUSTERED = {"clustered30": (0.3, 0.08), "clustered60": (0.6, 0.05), "clustered90": (0.9, 0.03)}


def make_upstream_grads(width: int, height: int, seed: int = 1):
    """dL/dcolor, dL/ddepth, dL/dalpha seeds."""
    g = torch.Generator().manual_seed(seed)
    hw = width * height
    dc = torch.randn(3, height, width, generator=g) / (3.0 * hw)
    dd = torch.randn(1, height, width, generator=g) / hw
    da = torch.randn(1, height, width, generator=g) / hw
    return dc, dd, da


# Named workloads.
WORKLOADS = {
    "S1": dict(P=10_000, width=256, height=256),
    "S2": dict(P=200_000, width=1008, height=756),
    "S2r8": dict(P=200_000, width=504, height=378),
    "S3": dict(P=500_000, width=1920, height=1080),
    "S4": dict(P=1_000_000, width=960, height=540),
}


def make_raw_model(sc: Scene, ray_fraction: float = 0.6, origin=(0.0, 0.0, 0.0), seed: int = 3):
    """The scene in the reference model's RAW parameterisation (scene/gaussian_model.py:452-468, 491-509): the first
    `ray_fraction` of the Gaussians ray-bound (`xyz = rayo + rayd * zval`, rays from `origin`), the rest free background
    Gaussians; opacity as a logit, scales as logs, quaternions UN-normalised (each scaled by a random factor in 0.5 .. 2),
    SH split into `features_dc` (P,1,3) and `features_rest` (P,15,3).  The activated getters of the returned
    `ply_io.RayBoundModel` reproduce `sc` up to the rounding of the activations."""
    from .ply_io import RayBoundModel
    P = sc.means3D.shape[0]
    nr = int(round(P * ray_fraction))
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor(origin, dtype=torch.float32)
    d = sc.means3D[:nr] - o
    zval = d.norm(dim=1, keepdim=True)
    rayd = (d / zval).contiguous()
    rayo = o[None].repeat(nr, 1).contiguous()
    opa = sc.opacities.clamp(1e-6, 1 - 1e-6)
    logit = torch.log(opa / (1 - opa))
    logs = torch.log(sc.scales)
    rot = sc.rotations * (0.5 + 1.5 * torch.rand(P, 1, generator=g))
    dc, rest = sc.shs[:, :1].contiguous(), sc.shs[:, 1:].contiguous()
    return RayBoundModel(
        features_dc=dc[:nr].contiguous(), features_rest=rest[:nr].contiguous(), opacity=logit[:nr].contiguous(),
        scaling=logs[:nr].contiguous(), rotation=rot[:nr].contiguous(), zval=zval.contiguous(), rayo=rayo, rayd=rayd,
        bg_xyz=sc.means3D[nr:].contiguous(), bg_features_dc=dc[nr:].contiguous(), bg_features_rest=rest[nr:].contiguous(),
        bg_opacity=logit[nr:].contiguous(), bg_scaling=logs[nr:].contiguous(), bg_rotation=rot[nr:].contiguous(),
        max_sh_degree=3)
