"""On-disk format of a trained SCGaussian scene: the data either side of the rasterizer.

The reference stores its ray-bound model with the `plyfile` package (absent here and on the GPU box) as binary
little-endian PLY files (scene/gaussian_model.py:531-609 `save_ply`, :653-756 `load_ply`):

  point_cloud.ply      one `vertex` element, all properties `float`, in this order
                       x y z nx ny nz | f_dc_0..2 | f_rest_0..44 | opacity | scale_0..2 | rot_0..3 |
                       zval_0 | rayo_0..2 | rayd_0..2           (x y z = rayo + rayd * zval, normals = 0)
  point_cloud_bg.ply   the free background Gaussians (only when there are any), same idea with a `b` prefix:
                       bx by bz bnx bny bnz | bf_dc_0..2 | bf_rest_0..44 | bopacity | bscale_0..2 | brot_0..3
  point_cloud_color.ply  x y z nx ny nz (float) red green blue (uchar) of all Gaussians, for viewers
                       (scene/dataset_readers.py:127-142 `storePly`; colour = f_dc * 255 cast to uchar)

SH coefficients are stored channel-major (`transpose(1, 2).flatten`: all R coefficients, then G, then B — :573-574),
raw (pre-activation) opacity / scale / rotation.  This module reads and writes exactly that layout with numpy only
and hands the tensors to the rasterizer through the same activations the reference applies
(scene/gaussian_model.py:105-155: exp, normalize, sigmoid, rayo + rayd * zval).

The byte layout follows the PLY specification and the reference's property order; it is not pinned against files
written by `plyfile` itself (not installable here) — "parity unpinned", the round trip and the header are tested.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
_NP_TO_PLY = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "i2": "short", "u2": "ushort",
              "i4": "int", "u4": "uint"}


# --------------------------------------------------------------------------------------------------------------
# minimal PLY container: one element, scalar properties, ascii or binary
# --------------------------------------------------------------------------------------------------------------
def write_vertex_ply(path: str, names: List[str], columns: np.ndarray, dtypes: Optional[List[str]] = None) -> None:
    """Write a single `vertex` element: columns (N, len(names)); dtypes are numpy codes ('f4', 'u1'...)."""
    columns = np.asarray(columns)
    n = columns.shape[0]
    dtypes = dtypes or ["f4"] * len(names)
    rec = np.empty(n, dtype=[(nm, "<" + dt) for nm, dt in zip(names, dtypes)])
    for i, nm in enumerate(names):
        rec[nm] = columns[:, i]            # numpy casts like `elements[:] = list(map(tuple, attributes))` does
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    header += [f"property {_NP_TO_PLY[dt]} {nm}" for nm, dt in zip(names, dtypes)]
    header.append("end_header")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(rec.tobytes())


def read_vertex_ply(path: str) -> Dict[str, np.ndarray]:
    """Read the first element of a PLY file (binary little/big endian or ascii) into {property: array}."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen_elements = None, None, [], False, 0
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen_elements += 1
                in_first = seen_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not part of the SCGaussian format")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: incomplete PLY header")
        if fmt == "ascii":
            rows = np.loadtxt(fh, max_rows=count, ndmin=2, dtype=np.float64)
            return {nm: rows[:, i].astype(dt) for i, (nm, dt) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, order + t) for nm, t in props])
        raw = fh.read(dt.itemsize * count)
        if len(raw) != dt.itemsize * count:
            raise ValueError(f"{path}: truncated vertex data")
        rec = np.frombuffer(raw, dtype=dt, count=count)
        return {nm: np.ascontiguousarray(rec[nm]).astype(rec[nm].dtype.newbyteorder("=")) for nm, _ in props}


# --------------------------------------------------------------------------------------------------------------
# the ray-bound model
# --------------------------------------------------------------------------------------------------------------
def _empty(*shape):
    return torch.zeros(shape, dtype=torch.float32)


@dataclass
class RayBoundModel:
    """Raw (pre-activation) tensors of the reference's GaussianModel; shapes as in scene/gaussian_model.py."""
    features_dc: torch.Tensor                 # (P, 1, 3)
    features_rest: torch.Tensor               # (P, 15, 3)
    opacity: torch.Tensor                     # (P, 1)   logit
    scaling: torch.Tensor                     # (P, 3)   log
    rotation: torch.Tensor                    # (P, 4)   unnormalised quaternion (r, x, y, z)
    zval: torch.Tensor                        # (P, 1)   depth along the matched ray
    rayo: torch.Tensor                        # (P, 3)   ray origin (fixed)
    rayd: torch.Tensor                        # (P, 3)   ray direction (fixed)
    bg_xyz: torch.Tensor = field(default_factory=lambda: _empty(0, 3))
    bg_features_dc: torch.Tensor = field(default_factory=lambda: _empty(0, 1, 3))
    bg_features_rest: torch.Tensor = field(default_factory=lambda: _empty(0, 15, 3))
    bg_opacity: torch.Tensor = field(default_factory=lambda: _empty(0, 1))
    bg_scaling: torch.Tensor = field(default_factory=lambda: _empty(0, 3))
    bg_rotation: torch.Tensor = field(default_factory=lambda: _empty(0, 4))
    max_sh_degree: int = 3
    _active_sh_degree: Optional[int] = None

    # ---- the reference's activated getters (scene/gaussian_model.py:105-155): the background set is concatenated only when
    # there is one, like there (`if hasattr(self, "bg_xyz") and self.bg_xyz.shape[0] > 0`) ----
    def _with_bg(self, t: torch.Tensor, bg: torch.Tensor) -> torch.Tensor:
        return torch.cat([t, bg.to(t.device)]) if bg.shape[0] > 0 else t

    @property
    def get_xyz(self) -> torch.Tensor:
        return self._with_bg(self.rayo + self.rayd * self.zval, self.bg_xyz)

    @property
    def get_features(self) -> torch.Tensor:
        dc, rest = self.features_dc, self.features_rest
        if self.bg_features_dc.shape[0] > 0:
            dc = torch.cat([dc, self.bg_features_dc.to(dc.device)])
            rest = torch.cat([rest, self.bg_features_rest.to(dc.device)])
        return torch.cat((dc, rest), dim=1)

    @property
    def get_opacity(self) -> torch.Tensor:
        opa = torch.sigmoid(self.opacity)
        return torch.cat([opa, torch.sigmoid(self.bg_opacity.to(opa.device))]) if self.bg_opacity.shape[0] > 0 else opa

    @property
    def get_scaling(self) -> torch.Tensor:
        scal = torch.exp(self.scaling)
        return torch.cat([scal, torch.exp(self.bg_scaling.to(scal.device))]) if self.bg_scaling.shape[0] > 0 else scal

    @property
    def get_rotation(self) -> torch.Tensor:
        rot = torch.nn.functional.normalize(self.rotation)
        if self.bg_rotation.shape[0] > 0:
            rot = torch.cat([rot, torch.nn.functional.normalize(self.bg_rotation.to(rot.device))])
        return rot

    def get_covariance(self, scaling_modifier: float = 1.0) -> torch.Tensor:
        """(P,6) upper triangle [xx,xy,xz,yy,yz,zz] of (R S)(R S)^T, as scene/gaussian_model.py:37-41,151-152 builds it
        for pipe.compute_cov3D_python: S = scaling_modifier * get_scaling and the RAW `_rotation` — build_rotation
        (utils/general_utils.py:84-105) normalises the quaternion itself, so raw and normalised give the same matrix.
        (With background Gaussians the reference passes only the ray-bound `_rotation` next to the concatenated scaling,
        a shape mismatch; here the background rotations are appended so the call is defined.)"""
        q = torch.cat([self.rotation, self.bg_rotation.to(self.rotation.device)])
        q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
        r, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        cov = L @ L.transpose(1, 2)
        return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)

    @property
    def active_sh_degree(self) -> int:
        """load_ply sets active = max (:710); a model that is being trained starts at 0 and is raised every 1 000 iterations
        (train.py:129 -> oneupSHdegree, scene/gaussian_model.py:157-159): assignable."""
        return self.max_sh_degree if self._active_sh_degree is None else self._active_sh_degree

    @active_sh_degree.setter
    def active_sh_degree(self, deg: int) -> None:
        self._active_sh_degree = int(deg)

    def oneupSHdegree(self) -> None:
        if self.active_sh_degree < self.max_sh_degree:
            self._active_sh_degree = self.active_sh_degree + 1

    def parameters(self) -> List[torch.Tensor]:
        """The trainable tensors in the order of the reference's two optimizers' parameter groups
        (scene/gaussian_model.py:491-509): zval, f_dc, f_rest, opacity, scaling, rotation, then — when there is a
        background set — bg_xyz, bg_f_dc, bg_f_rest, bg_opacity, bg_scaling, bg_rotation."""
        ps = [self.zval, self.features_dc, self.features_rest, self.opacity, self.scaling, self.rotation]
        if self.bg_xyz.shape[0] > 0:
            ps += [self.bg_xyz, self.bg_features_dc, self.bg_features_rest, self.bg_opacity, self.bg_scaling, self.bg_rotation]
        return ps

    def to(self, device) -> "RayBoundModel":
        kw = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()}
        return RayBoundModel(**kw)

    def requires_grad_(self, flag: bool = True) -> "RayBoundModel":
        for p in self.parameters():
            p.requires_grad_(flag)
        return self


def _attribute_names(n_dc: int, n_rest: int, prefix: str = "", ray_bound: bool = True) -> List[str]:
    """construct_list_of_attributes / construct_list_of_attributes_bg (scene/gaussian_model.py:531-565)."""
    p = prefix
    names = [p + "x", p + "y", p + "z", p + "nx", p + "ny", p + "nz"]
    names += [f"{p}f_dc_{i}" for i in range(n_dc)]
    names += [f"{p}f_rest_{i}" for i in range(n_rest)]
    names.append(p + "opacity")
    names += [f"{p}scale_{i}" for i in range(3)]
    names += [f"{p}rot_{i}" for i in range(4)]
    if ray_bound:
        names += ["zval_0"] + [f"rayo_{i}" for i in range(3)] + [f"rayd_{i}" for i in range(3)]
    return names


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().astype(np.float32)


def _flat_sh(t: torch.Tensor) -> np.ndarray:
    """(P, K, 3) -> (P, 3K) channel-major, as `transpose(1, 2).flatten(start_dim=1)` (:573)."""
    return _np(t.detach().transpose(1, 2).flatten(start_dim=1).contiguous())


def save_ply(path: str, model: RayBoundModel, write_color_ply: bool = True) -> None:
    """Twin of GaussianModel.save_ply: `path` is .../point_cloud.ply; the bg / colour files go next to it."""
    xyz = _np(model.rayo + model.rayd * model.zval)
    f_dc, f_rest = _flat_sh(model.features_dc), _flat_sh(model.features_rest)
    attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, _np(model.opacity), _np(model.scaling),
                            _np(model.rotation), _np(model.zval), _np(model.rayo), _np(model.rayd)), axis=1)
    write_vertex_ply(path, _attribute_names(f_dc.shape[1], f_rest.shape[1]), attrs)
    folder = os.path.dirname(os.path.abspath(path))
    all_xyz, all_color = xyz, f_dc
    if model.bg_xyz.shape[0] > 0:
        bg_xyz = _np(model.bg_xyz)
        bg_dc, bg_rest = _flat_sh(model.bg_features_dc), _flat_sh(model.bg_features_rest)
        bg_attrs = np.concatenate((bg_xyz, np.zeros_like(bg_xyz), bg_dc, bg_rest, _np(model.bg_opacity),
                                   _np(model.bg_scaling), _np(model.bg_rotation)), axis=1)
        write_vertex_ply(os.path.join(folder, "point_cloud_bg.ply"),
                         _attribute_names(bg_dc.shape[1], bg_rest.shape[1], prefix="b", ray_bound=False), bg_attrs)
        all_xyz, all_color = np.concatenate([xyz, bg_xyz]), np.concatenate([f_dc, bg_dc])
    if write_color_ply:
        cols = np.concatenate((all_xyz, np.zeros_like(all_xyz), all_color * 255), axis=1)
        write_vertex_ply(os.path.join(folder, "point_cloud_color.ply"),
                         ["x", "y", "z", "nx", "ny", "nz", "red", "green", "blue"], cols,
                         dtypes=["f4"] * 6 + ["u1"] * 3)


def _columns(props: Dict[str, np.ndarray], prefix: str) -> np.ndarray:
    """All properties `<prefix><int>` ordered by their integer suffix (the reference sorts the same way, :672-704)."""
    names = [n for n in props if n.startswith(prefix) and n[len(prefix):].lstrip("_").isdigit()]
    names.sort(key=lambda n: int(n.split("_")[-1]))
    if not names:
        raise ValueError(f"PLY file has no '{prefix}*' properties")
    return np.stack([props[n] for n in names], axis=1).astype(np.float32)


def _sh_from_columns(props, dc_prefix: str, rest_prefix: str, max_sh_degree: int) -> Tuple[torch.Tensor, torch.Tensor]:
    dc = _columns(props, dc_prefix)                                           # (P, 3) = one coefficient x RGB
    rest = _columns(props, rest_prefix)
    want = 3 * (max_sh_degree + 1) ** 2 - 3
    if rest.shape[1] != want:
        raise ValueError(f"expected {want} {rest_prefix}* properties for SH degree {max_sh_degree}, found {rest.shape[1]}")
    dc_t = torch.from_numpy(dc.reshape(-1, 3, 1)).transpose(1, 2).contiguous()
    rest_t = torch.from_numpy(rest.reshape(rest.shape[0], 3, -1)).transpose(1, 2).contiguous()
    return dc_t, rest_t


def load_ply(path: str, max_sh_degree: int = 3, device="cpu") -> RayBoundModel:
    """Twin of GaussianModel.load_ply (scene/gaussian_model.py:653-756), including the optional bg file."""
    props = read_vertex_ply(path)
    dc, rest = _sh_from_columns(props, "f_dc_", "f_rest_", max_sh_degree)
    kw = dict(features_dc=dc, features_rest=rest,
              opacity=torch.from_numpy(props["opacity"].astype(np.float32)[:, None].copy()),
              scaling=torch.from_numpy(_columns(props, "scale_")), rotation=torch.from_numpy(_columns(props, "rot")),
              zval=torch.from_numpy(_columns(props, "zval")), rayo=torch.from_numpy(_columns(props, "rayo")),
              rayd=torch.from_numpy(_columns(props, "rayd")), max_sh_degree=max_sh_degree)
    bg_path = os.path.join(os.path.dirname(os.path.abspath(path)), "point_cloud_bg.ply")
    if os.path.exists(bg_path):
        bp = read_vertex_ply(bg_path)
        bdc, brest = _sh_from_columns(bp, "bf_dc_", "bf_rest_", max_sh_degree)
        kw.update(bg_xyz=torch.from_numpy(np.stack((bp["bx"], bp["by"], bp["bz"]), axis=1).astype(np.float32)),
                  bg_features_dc=bdc, bg_features_rest=brest,
                  bg_opacity=torch.from_numpy(bp["bopacity"].astype(np.float32)[:, None].copy()),
                  bg_scaling=torch.from_numpy(_columns(bp, "bscale_")), bg_rotation=torch.from_numpy(_columns(bp, "brot")))
    return RayBoundModel(**kw).to(device)
