"""The rasterizer fed with the reference MODEL's raw parameter tensors (ABI 10: scg_forward_model / scg_backward_model).

The reference's render() (gaussian_renderer/__init__.py:20-118) turns the model into the operator's activated inputs on every
call: `get_xyz` = cat(rayo + rayd * zval, bg_xyz) three times (:28 twice, :55), `get_features` = cat(cat(f_dc, bg_f_dc),
cat(f_rest, bg_f_rest), dim=1) — a 192 B / Gaussian copy (:85; scene/gaussian_model.py:134-142) —, sigmoid / exp / normalize +
cat for opacity / scaling / rotation (:57, :67-68; scene/gaussian_model.py:105-152): 67 torch launches per training step forward
and backward around a 6-launch rasterizer step, measured at 0.70 ms per step against 0.25 ms for the bare operator at 200 k
Gaussians (bench.py `render_glue`, round 6).  Here the two geometry kernels read the raw tensors where they lie and write the
gradients of the RAW parameters into one flat arena: activations, their derivatives and the concatenations happen in registers.

    color, radii, depth, alpha = rasterize_model(settings, means2D, zval, rayo, rayd, features_dc, features_rest, opacity,
                                                 scaling, rotation, bg_xyz, bg_features_dc, ..., bg_rotation)

`render.render()` takes this path by itself for a model that carries these tensors (reference attribute names `_zval`,
`_features_dc`, ..., `bg_xyz`, ... or ply_io.RayBoundModel's) with the reference's default pipe; everything else goes through the
getters as before.  The gradient arena puts `features_rest` / `bg_features_rest` LAST: the other ten tensors' gradients are one
contiguous span, which parallel.GradBucket all-reduces in place at every SH degree.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from . import rasterizer as R
from ._lib import ScgModel, ScgModelGrads, check, ptr

# order of the operator's tensor arguments (behind means2D) = order of the gradients backward returns
_RAY = ("zval", "rayo", "rayd", "features_dc", "features_rest", "opacity", "scaling", "rotation")
_BG = ("bg_xyz", "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation")
ARG_NAMES = _RAY + _BG
# segments of the gradient arena: the two large SH tensors last (see the module docstring); rayo / rayd are constants
_ARENA_ORDER = ("zval", "opacity", "scaling", "rotation", "features_dc", "bg_xyz", "bg_opacity", "bg_scaling", "bg_rotation",
                "bg_features_dc", "features_rest", "bg_features_rest")
_LAYOUTS = {}


class _ModelArgs:
    """ScgModel plus the tensors whose storage it points at (kept alive for the call and saved for backward)."""

    def __init__(self, tensors: dict):
        t = tensors
        self.tensors = tuple(t[n] for n in ARG_NAMES)
        nr, nb = int(t["zval"].shape[0]), int(t["bg_xyz"].shape[0])
        self.n_ray, self.n_bg, self.P = nr, nb, nr + nb
        self.device = (t["zval"] if nr else t["bg_xyz"]).device
        m = ScgModel()
        m.ray.count, m.bg.count = nr, nb
        if nr:
            for f, n in (("zval", "zval"), ("rayo", "rayo"), ("rayd", "rayd"), ("features_dc", "features_dc"),
                         ("features_rest", "features_rest"), ("opacity", "opacity"), ("scaling", "scaling"),
                         ("rotation", "rotation")):
                setattr(m.ray, f, t[n].data_ptr())
        if nb:
            for f, n in (("xyz", "bg_xyz"), ("features_dc", "bg_features_dc"), ("features_rest", "bg_features_rest"),
                         ("opacity", "bg_opacity"), ("scaling", "bg_scaling"), ("rotation", "bg_rotation")):
                setattr(m.bg, f, t[n].data_ptr())
        self.c = m
        self.ref = C.byref(m)


def _usable(t: torch.Tensor, device, align16: bool) -> bool:
    return (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == device and t.is_contiguous()
            and (t.numel() == 0 or not align16 or t.data_ptr() % 16 == 0))


def supported(tensors: dict, settings=None) -> bool:
    """True when the model path can take these tensors as they are: fp32, contiguous, one CUDA device, the reference's shapes
    ((n,1) zval / opacity, (n,3) rays / scaling / positions, (n,4) rotation, (n,1,3) + (n,15,3) SH), 16-byte aligned where the
    kernels use 16-byte accesses — and, with `settings`, an image the tile-first binning takes.  Anything else: the getters."""
    try:
        z, bx = tensors["zval"], tensors["bg_xyz"]
        nr, nb = int(z.shape[0]), int(bx.shape[0])
        if nr + nb == 0:
            return False
        dev = (z if nr else bx).device
        if dev.type != "cuda":
            return False
        shapes = {"zval": (nr, 1), "rayo": (nr, 3), "rayd": (nr, 3), "features_dc": (nr, 1, 3), "features_rest": (nr, 15, 3),
                  "opacity": (nr, 1), "scaling": (nr, 3), "rotation": (nr, 4), "bg_xyz": (nb, 3), "bg_features_dc": (nb, 1, 3),
                  "bg_features_rest": (nb, 15, 3), "bg_opacity": (nb, 1), "bg_scaling": (nb, 3), "bg_rotation": (nb, 4)}
        for n, shp in shapes.items():
            t = tensors[n]
            if shp[0] == 0:
                if t.shape[0] != 0:
                    return False
                continue
            if tuple(t.shape) != shp or not _usable(t, dev, n.endswith(("features_dc", "features_rest", "rotation"))):
                return False
        if settings is not None:
            lib = _lib.load()
            if lib.scg_binning_accepts_bound(R._capacity_for(4 * (nr + nb)), int(settings.image_width),
                                             int(settings.image_height), 0) != 1:
                return False
        return True
    except (KeyError, AttributeError, IndexError):
        return False


_ACCEPTS = {}                             # (P, W, H) -> the tile-first binning takes the model path's first-render bound


def _accepts(P: int, settings) -> bool:
    key = (P, int(settings.image_width), int(settings.image_height))
    v = _ACCEPTS.get(key)
    if v is None:
        if len(_ACCEPTS) > 256:
            _ACCEPTS.clear()
        v = _ACCEPTS[key] = _lib.load().scg_binning_accepts_bound(R._capacity_for(4 * P), key[1], key[2], 0) == 1
    return v


_ATTRS = None


def model_for(pc, settings=None) -> Optional[_ModelArgs]:
    """The _ModelArgs of `pc` when the model path can render it (tensors_of + supported), else None.  The answer is remembered ON
    the model object and reused while its fourteen tensors are the same objects at the same addresses (a training loop renders
    the same parameters thousands of times between two densifications; the checks cost as much host time as the forward's
    launches at the reference's scene size): validated per call by identity + data_ptr, rebuilt after any change."""
    global _ATTRS
    cached = pc.__dict__.get("_scg_model_args") if hasattr(pc, "__dict__") else None
    if cached is not None:
        attrs, args = cached
        ok = True
        for a, n, t, ptr_ in zip(attrs, ARG_NAMES, args.tensors, args.ptrs):
            if a is None:                                    # (a stand-in for a set the model did not have: still without it?)
                if isinstance(getattr(pc, n, None), torch.Tensor):
                    ok = False
                    break
                continue
            if getattr(pc, a, None) is not t or t.data_ptr() != ptr_:
                ok = False
                break
        if ok and args.n_ray == args.tensors[0].shape[0] and args.n_bg == args.tensors[8].shape[0]:
            return args if (settings is None or _accepts(args.P, settings)) else None
    t = tensors_of(pc)
    if t is None or not supported(t):
        return None
    args = _ModelArgs(t)
    args.ptrs = tuple(x.data_ptr() for x in args.tensors)
    # which attribute of pc each tensor came from (None: an empty stand-in tensors_of made up for a model without a bg set)
    attrs = []
    for n, x in zip(ARG_NAMES, args.tensors):
        found = None
        for a in (_REFERENCE_ATTRS.get(n) or (n,)):
            if getattr(pc, a, None) is x:
                found = a
                break
        attrs.append(found)
    try:
        pc.__dict__["_scg_model_args"] = (tuple(attrs), args)
    except (AttributeError, TypeError):
        pass
    return args if (settings is None or _accepts(args.P, settings)) else None


def _grad_arena(model: _ModelArgs, into):
    """{name: gradient tensor} for the trainable tensors of the non-empty sets, all views of ONE flat arena (pooled: see
    rasterizer._take_arena), + "_pooled" and "_c" (the ScgModelGrads struct).  `into`: an earlier call's result — reused."""
    if into is not None:
        return into
    key = (model.n_ray, model.n_bg)
    lay = _LAYOUTS.get(key)
    if lay is None:
        if len(_LAYOUTS) > 64:
            _LAYOUTS.clear()
        names, sizes, shapes = [], [], []
        by_name = dict(zip(ARG_NAMES, model.tensors))
        for n in _ARENA_ORDER:
            t = by_name[n]
            if t.shape[0] == 0:
                continue
            names.append(n)
            sizes.append((t.numel() + 3) // 4 * 4)               # 16-byte aligned segments
            shapes.append(tuple(t.shape))
        lay = _LAYOUTS[key] = (tuple(names), sizes, tuple(shapes), max(sum(sizes), 4))
    names, sizes, shapes, total = lay
    dev = model.device
    arena, pooled = R._take_arena(("model", key, dev.index), total, dev)
    out = {"_pooled": pooled}
    views = arena.split_with_sizes(sizes) if total == sum(sizes) else arena[: sum(sizes)].split_with_sizes(sizes)
    for n, v, shp, sz in zip(names, views, shapes, sizes):
        out[n] = (v if sz == math.prod(shp) else v[:math.prod(shp)]).view(shp)
    # the struct of raw pointers: a pooled arena's segments stay where they are — built once per arena
    g = getattr(pooled, "cstruct", None) if pooled is not None else None
    if g is None:
        g = ScgModelGrads()
        base, off = arena.data_ptr(), 0
        for n, sz in zip(names, sizes):
            if n.startswith("bg_"):
                setattr(g.bg, n[3:], base + 4 * off)
            else:
                setattr(g.ray, n, base + 4 * off)
            off += sz
        if pooled is not None:
            pooled.cstruct = g
    out["_c"] = g
    return out


def backward_fused_model(model: _ModelArgs, radii, state, dL_dcolor, dL_ddepth, dL_dalpha, timer=None, into=None,
                         d_means2D_out=None):
    """Stages 4-5 of the model path in one library call (scg_backward_model).  Returns {argument name: gradient} (views of one
    arena) + "means2D"; `into` as in rasterizer.backward_fused (the kernel adds to an earlier view's gradients)."""
    lib = _lib.load()
    if not state.get("has_backward_state", True):
        raise _lib.ScgError("this forward ran without backward state (SCG_FORWARD_NO_BACKWARD_STATE): it cannot be differentiated")
    dev = model.device
    fr = state["frame"]
    if fr.hints is not None and fr.hints.bcost is not None:
        fr.hints.bwritten = True
    H, W = fr.H, fr.W
    dL_dcolor = R._f32c(dL_dcolor, dev)
    if dL_dcolor is None:
        dL_dcolor = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
    dL_ddepth = R._f32c(dL_ddepth, dev)
    dL_dalpha = R._f32c(dL_dalpha, dev)
    timer = timer or R._ACTIVE_TIMER
    with R._on_device(dev):
        stage_ev = timer.stage_events("backward") if isinstance(timer, R.StageTimer) else None
        stream = R._stream(dev)
        dsplats = state.get("dsplats_zeroed")
        state["dsplats_zeroed"] = None
        prezeroed = dsplats is not None
        keep = None
        if dsplats is None:
            keep = torch.empty((model.P, R.DSPLAT_FLOATS), dtype=torch.float32, device=dev)
            dsplats = keep.data_ptr()
        out = _grad_arena(model, into)
        d_means2D = d_means2D_out if d_means2D_out is not None else torch.empty((model.P, 3), dtype=torch.float32, device=dev)
        flags = 1 if into is not None else R._sh_tail_promise(out.get("_pooled"), (fr.c.sh_degree + 1) ** 2)
        check(lib.scg_backward_model(fr.ref, model.ref, radii.data_ptr(), state["cap"], state["ws"].data_ptr(),
                                     dL_dcolor.data_ptr(), ptr(dL_ddepth), ptr(dL_dalpha), dsplats, int(prezeroed),
                                     C.byref(out["_c"]), d_means2D.data_ptr(), flags, stage_ev, stream), "scg_backward_model")
    out = dict(out)
    out["means2D"] = d_means2D
    return out


class _RasterizeModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2D, *args):
        *tensors, raster_settings, model = args
        if model is None:
            model = _ModelArgs(dict(zip(ARG_NAMES, tensors)))
        needs_grad = any(ctx.needs_input_grad)
        fused = R.forward_fused(raster_settings, None, None, None, None, None, None, None, needs_grad, model=model)
        if fused is None:
            raise _lib.ScgError("the model path needs the tile-first binning (scg_binning_accepts_bound): check "
                                "model_path.supported(tensors, settings) first and render through the getters otherwise")
        color, radii, depth, alpha, state = fused
        state.pop("inputs")
        ctx.model = model
        ctx.fused_state = state
        ctx.raster_settings = raster_settings
        if needs_grad:
            ctx.save_for_backward(*model.tensors, radii)         # in-place updates between forward and backward trip the version check
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        saved = ctx.saved_tensors                                # raises if a parameter was modified in place since forward
        g = backward_fused_model(ctx.model, saved[-1], ctx.fused_state, grad_color, grad_depth, grad_alpha)
        return (g["means2D"],) + tuple(g.get(n) for n in ARG_NAMES) + (None, None)


def rasterize_model(raster_settings, means2D: torch.Tensor, _args: Optional[_ModelArgs] = None, **tensors):
    """(color (3,H,W), radii (P,) int32, depth (1,H,W), alpha (1,H,W)) of the model given by its raw tensors (keyword
    arguments named as in ARG_NAMES; `means2D` (P,3): the screen-space gradient slot, reference gaussian_renderer/__init__.py:28).
    Differentiable with respect to every tensor but rayo / rayd.  `_args`: the validated _ModelArgs of these tensors
    (model_for), in place of the keyword arguments."""
    R._require_cuda(means2D)
    ts = _args.tensors if _args is not None else tuple(tensors[n] for n in ARG_NAMES)
    return _RasterizeModel.apply(means2D, *ts, raster_settings, _args)


class _RasterizeModelViews(torch.autograd.Function):
    """K views of one model in ONE autograd node (BASELINE cfg5's "multi-view batched step", rasterizer._RasterizeViews on the
    model path): forward = the K views one after the other; backward = per view blend backward + geometry backward, the second and
    later views ADDING their raw-parameter gradients to the first one's in the kernel (scg_backward_model `accumulate`), so the
    node returns the gradient of the SUM over views from one arena — one exchange per K views.  Outputs, flat: color_0, radii_0,
    depth_0, alpha_0, color_1, ...; means2D is (K, P, 3)."""

    @staticmethod
    def forward(ctx, means2D, *args):
        *tensors, settings_list, model = args
        if model is None:
            model = _ModelArgs(dict(zip(ARG_NAMES, tensors)))
        needs_grad = any(ctx.needs_input_grad)
        outs, states = [], []
        for st_ in settings_list:
            fused = R.forward_fused(st_, None, None, None, None, None, None, None, needs_grad, model=model)
            if fused is None:
                raise _lib.ScgError("the model path needs the tile-first binning (scg_binning_accepts_bound) for every view")
            color, radii, depth, alpha, state = fused
            state.pop("inputs")
            states.append(state)
            outs += [color, radii, depth, alpha]
        ctx.model, ctx.states, ctx.K = model, states, len(settings_list)
        if needs_grad:
            ctx.save_for_backward(*model.tensors, *outs[1::4])
        ctx.mark_non_differentiable(*outs[1::4])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        K, model = ctx.K, ctx.model
        saved = ctx.saved_tensors
        radii_all = saved[len(saved) - K:]
        d_means2D = torch.empty((K, model.P, 3), dtype=torch.float32, device=model.device)
        acc = None
        for k in range(K):
            g_color, _, g_depth, g_alpha = grads[4 * k: 4 * k + 4]
            if g_color is None and g_depth is None and g_alpha is None:
                d_means2D[k].zero_()
                continue                                         # this view's outputs did not reach the loss
            acc = backward_fused_model(model, radii_all[k], ctx.states[k], g_color, g_depth, g_alpha, into=acc,
                                       d_means2D_out=d_means2D[k])
        if acc is None:
            return (None,) * (len(ARG_NAMES) + 3)
        return (d_means2D,) + tuple(acc.get(n) for n in ARG_NAMES) + (None, None)


def rasterize_model_views(settings_list, means2D: torch.Tensor, _args: Optional[_ModelArgs] = None, **tensors):
    """[(color, radii, depth, alpha)] * K for the K views `settings_list` of one model (raw tensors as in rasterize_model);
    `means2D` is (K, P, 3): one screen-space gradient slot per view.  One autograd node: the views' gradients are summed in the
    kernels and land in one arena."""
    R._require_cuda(means2D)
    K = len(settings_list)
    if means2D.dim() != 3 or means2D.shape[0] != K:
        raise ValueError(f"means2D must be ({K}, P, 3): one screen-space gradient slot per view")
    ts = _args.tensors if _args is not None else tuple(tensors[n] for n in ARG_NAMES)
    flat = _RasterizeModelViews.apply(means2D, *ts, list(settings_list), _args)
    return [tuple(flat[4 * k: 4 * k + 4]) for k in range(K)]


def activate(**tensors):
    """The model's activated getters in ONE launch (scg_model_activate): (xyz (P,3), opacity (P,1), scaling (P,3), rotation (P,4))
    as reference scene/gaussian_model.py:105-152 computes them, by the device functions the geometry kernels use.  No autograd."""
    model = _ModelArgs({n: tensors[n].detach() for n in ARG_NAMES})
    dev = model.device
    P = model.P
    out = [torch.empty((P, k), dtype=torch.float32, device=dev) for k in (3, 1, 3, 4)]
    with R._on_device(dev):
        check(_lib.load().scg_model_activate(model.ref, *(o.data_ptr() for o in out), R._stream(dev)), "scg_model_activate")
    return tuple(out)


# ---- the reference model's attribute names -> ARG_NAMES -------------------------------------------------------------------
_REFERENCE_ATTRS = {"zval": ("_zval", "zval"), "rayo": ("_rayo", "rayo"), "rayd": ("_rayd", "rayd"),
                    "features_dc": ("_features_dc", "features_dc"), "features_rest": ("_features_rest", "features_rest"),
                    "opacity": ("_opacity", "opacity"), "scaling": ("_scaling", "scaling"), "rotation": ("_rotation", "rotation")}


def tensors_of(pc) -> Optional[dict]:
    """The raw tensors of a reference GaussianModel (scene/gaussian_model.py:452-468: `_zval`, `_rayo`, `_rayd`, `_features_dc`,
    `_features_rest`, `_opacity`, `_scaling`, `_rotation`, `bg_xyz`, `bg_features_dc`, ...) or of a ply_io.RayBoundModel (the same
    names without the underscore), keyed by ARG_NAMES; None when `pc` does not carry them."""
    out = {}
    for name, cands in _REFERENCE_ATTRS.items():
        t = None
        for a in cands:
            t = getattr(pc, a, None)
            if isinstance(t, torch.Tensor):
                break
        if not isinstance(t, torch.Tensor):
            return None
        out[name] = t
    dev = out["zval"].device
    for name in _BG:
        t = getattr(pc, name, None)
        if not isinstance(t, torch.Tensor):                       # a model without a background set (hasattr(self, "bg_xyz") false)
            k = {"bg_xyz": (0, 3), "bg_features_dc": (0, 1, 3), "bg_features_rest": (0, 15, 3), "bg_opacity": (0, 1),
                 "bg_scaling": (0, 3), "bg_rotation": (0, 4)}[name]
            t = torch.empty(k, dtype=torch.float32, device=dev)
        out[name] = t
    return out
