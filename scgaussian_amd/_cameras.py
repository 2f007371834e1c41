"""Which camera is this?  (Split out of rasterizer.py in round 6.)

The binding keeps per-camera history — the capacity a forward is launched with, launch-order hints, long-list counts — and looks a
camera up by `_camera_key(viewmatrix)`: the content of its view matrix, a name the caller gave it (tag_camera), never the address of
a tensor alone.
"""
from __future__ import annotations

import weakref

import torch

# A camera is identified by the CONTENT of its view matrix, not by the address of the tensor that holds it: an address is
# recycled by the allocator as soon as the tensor dies, and the next camera that lands on it would inherit the capacity,
# tile costs and long-list counts of another view.  The sixteen floats of a DEVICE tensor are read ONCE per
# tensor (one small device-to-host copy the first time a view-matrix tensor is seen, again only after an in-place write to it:
# the version counter says so; a write through `.data` or a raw pointer does not bump it — the key then goes stale, which costs
# hints, never a result).  The table holds the tensor WEAKLY (round 6; it used to keep up to 2 048 tensors, and whatever storage
# they were views of, alive) when it is a view into a LARGE storage — the entry goes when the tensor does; a matrix in a small
# storage of its own is kept alive by a detached alias (64 bytes), so its address cannot be recycled while the entry exists.
# The reference builds each camera's matrices once and keeps them for the whole run (scene/cameras.py:60-63).
# A caller that builds a NEW camera per frame (render_video.py:130-style) would pay that copy — a device synchronisation — per
# frame; two ways around it, neither reads the device: pass the view matrix as a CPU tensor (its content is the key; the
# binding uploads it), or name the camera: tag_camera(viewmatrix, camera_id).
_CAM_KEYS = {}                                   # data_ptr -> (weak reference to the tensor, version at the read, content bytes)
_CAM_KEYS_MAX = 2048


def tag_camera(viewmatrix: torch.Tensor, camera_id) -> torch.Tensor:
    """Name the camera this view-matrix tensor belongs to (any hashable with a stable repr: the reference's `Camera.uid`, a frame
    counter's "video" for a fly-through whose frames may share hints): the rasterizer then keys its per-camera hints (capacity,
    tile costs) by that name and never reads the tensor's content back from the device.  Returns the tensor."""
    viewmatrix._scg_camera_id = b"id:" + repr(camera_id).encode()
    return viewmatrix


def _camera_key(vm) -> bytes:
    if not isinstance(vm, torch.Tensor):
        return b""
    tagged = getattr(vm, "_scg_camera_id", None)
    if tagged is not None:
        return tagged
    if not vm.is_cuda:                           # host memory: the content itself, no copy to wait for
        return vm.detach().reshape(-1).to(torch.float32).numpy().tobytes()
    p = vm.data_ptr()
    ent = _CAM_KEYS.get(p)
    if ent is not None and ent[1] == vm._version and ent[0]() is not None:
        return ent[2]
    key = vm.detach().reshape(-1).to("cpu", torch.float32).numpy().tobytes()
    if ent is None and len(_CAM_KEYS) >= _CAM_KEYS_MAX:          # bounded: the oldest entries go (insertion order)
        for k in list(_CAM_KEYS)[: _CAM_KEYS_MAX // 4]:
            del _CAM_KEYS[k]
    if vm.untyped_storage().nbytes() <= 4096:
        # a matrix in a storage of its own (the usual case): a detached alias keeps the 64 bytes alive, so the address cannot be
        # recycled and callers that build a new VIEW object of the same memory per call (`cam.w2v.T`) are recognised by address
        alias = vm.detach()
        _CAM_KEYS[p] = ((lambda a=alias: a), vm._version, key)
    else:
        # a view into something large (a table of all cameras' matrices): held weakly — the entry goes when the tensor does
        def _gone(_ref, p=p):
            e = _CAM_KEYS.get(p)
            if e is not None and e[0] is _ref:
                del _CAM_KEYS[p]
        _CAM_KEYS[p] = (weakref.ref(vm, _gone), vm._version, key)
    return key
