"""Host-side mirror of the reference's rasterizer operator API, on top of the C-ABI HIP library.

Mirrors (names, argument meaning, error behaviour) the package the reference imports at
gaussian_renderer/__init__.py:15 and uses at :38-53 and :100-108:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None)
        -> (color (3,H,W), radii (P,) int32, depth (1,H,W), alpha (1,H,W))

PyTorch is plumbing here (device memory, the current stream, autograd graph edges); every stage of the
computation runs in hand-written HIP kernels behind include/scg_raster.h.  There is no fallback path.

Concurrency: the operator is re-entrant per device and stream, like the upstream extension — every forward takes its own
pinned scratch (the partial sums of num_rendered) from a per-device pool, so two threads, each on its own stream, may be
inside a forward at the same time (round 5; before, the second one raised).  The host-side hint tables (capacities, tile
costs) are plain dictionaries updated under the GIL: a concurrent update costs a hint, never a result.  Forward and
backward may run on different threads (autograd's device thread) and on any stream; several forwards may be
outstanding before their backwards run (each keeps its own saved state).  The inputs are saved with
ctx.save_for_backward: modifying one in place between forward and backward raises autograd's version-counter error,
as it does with the reference's operator.
"""
from __future__ import annotations

import contextlib
import math
import os
import ctypes as C
import threading
from typing import Callable, NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import ScgFrame, check, ptr

SPLAT_FLOATS = 12
DSPLAT_FLOATS = 16         # gradient record of a Gaussian: one 64-byte line (scg_raster.h SCG_DSPLAT_FLOATS)
TILE = 16


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.dtype == torch.float32 and t.device == device and t.is_contiguous():     # the usual case: nothing to do
        return t
    if t.device != device:
        if not t.is_cuda and device.type == "cuda" and t.numel() <= 64:
            # a camera's matrices / background handed over in host memory (a caller that builds a camera per frame): through a
            # pinned block and an asynchronous copy — a pageable-memory upload would wait for everything queued on the stream
            return t.detach().to(torch.float32).contiguous().pin_memory().to(device, non_blocking=True)
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _Frame:
    """ScgFrame plus the tensors whose storage it points at (kept alive for the call)."""

    def __init__(self, settings: GaussianRasterizationSettings, P: int, M: int, device):
        self.keep = [_f32c(settings.viewmatrix, device), _f32c(settings.projmatrix, device),
                     _f32c(settings.campos, device), _f32c(settings.bg, device)]
        if any(k is None for k in self.keep[:3]):
            raise ValueError("viewmatrix / projmatrix / campos must be non-empty tensors")
        self.vm_src = settings.viewmatrix        # the caller's tensor: what names the camera (host content / tag_camera / content)
        self.c = ScgFrame(P=int(P), sh_degree=int(settings.sh_degree), sh_coeffs=int(M),
                          width=int(settings.image_width), height=int(settings.image_height),
                          tanfovx=float(settings.tanfovx), tanfovy=float(settings.tanfovy),
                          scale_modifier=float(settings.scale_modifier),
                          prefiltered=int(bool(settings.prefiltered)), debug=int(bool(settings.debug)),
                          viewmatrix=ptr(self.keep[0]), projmatrix=ptr(self.keep[1]), campos=ptr(self.keep[2]),
                          bg=ptr(self.keep[3]))
        self.H, self.W = int(settings.image_height), int(settings.image_width)
        self.n_tiles = ((self.W + TILE - 1) // TILE) * ((self.H + TILE - 1) // TILE)
        # what the previous render of this CAMERA left for the next one (launch-order hint, long-list counts): shared by every
        # frame of the camera — a frame is (camera, background, ...): a training loop with random backgrounds
        # (reference train.py:141) builds a new frame per iteration, its hints must not start from nothing each time
        self.hints = None
        self.cam_key = None              # identity of the camera (content of its view matrix, _camera_key): set at the first forward
        self.ref = C.byref(self.c)

    @property
    def long_np(self):
        """The camera's two ScgFrame.long_lists_out words as a numpy view (None before the first forward / when switched off)."""
        return None if self.hints is None else self.hints.long_np

    def next_forward(self, device):
        """Called once per forward: what the previous render of this camera recorded becomes the hint, the other buffer
        receives this render's costs.  A hint never changes a result, only which kernels are launched and in which order
        tiles start."""
        # (two dictionary look-ups: the matrix itself is read once per tensor; the camera's record moves to the recently-used
        # end of its table, and one that was evicted meanwhile — its pinned words belong to another camera now — is replaced)
        self.cam_key = cam = _camera_key(self.vm_src)
        h = self.hints = _hints_for(cam, self.W, self.H, device, self.n_tiles)
        if h.cost is not None:
            cur = _stream(device)
            if cur not in h.streams:
                h.streams.add(cur)
                ts = torch.cuda.current_stream(device)
                for t in (*h.cost, *h.bcost):
                    t.record_stream(ts)
        self.c.long_lists_out = h.long_ptr
        if h.cost is not None:
            self.c.tile_cost_in = h.cost[h.cur].data_ptr() if h.cost_valid else None
            h.cur ^= 1
            self.c.tile_cost_out = h.cost[h.cur].data_ptr()
            h.cost_valid = True
            # the blend backward's own order (ScgFrame.bwd_cost_in / _out, per quadrant): the buffer the camera's latest
            # BACKWARD wrote becomes the hint; it is swapped only when a backward ran since (renders in between keep it)
            if h.bwritten:
                h.bcur ^= 1
                h.bwritten, h.bvalid = False, True
            self.c.bwd_cost_in = h.bcost[h.bcur ^ 1].data_ptr() if (h.bvalid and BWD_ORDER_HINT) else None
            self.c.bwd_cost_out = h.bcost[h.bcur].data_ptr()
        else:                            # (a cached frame whose camera's record was made with TILE_COST_HINT off)
            self.c.tile_cost_in = self.c.tile_cost_out = self.c.bwd_cost_in = self.c.bwd_cost_out = None


class _CamHints:
    """Per (camera, image size, device): two buffers of per-tile costs swapped at every forward (ScgFrame.tile_cost_in /
    _out) and the two pinned words of ScgFrame.long_lists_out (ABI 8: the number of tiles whose list is longer than the
    forward blend sorts itself / longer than 16 384 entries; -1: no render has completed yet).  Not per Gaussian count: the
    buffers are per TILE, and a record that died at every densification would leave the camera without history each time."""
    __slots__ = ("cost", "cur", "cost_valid", "long_np", "long_ptr", "slot", "bcost", "bcur", "bwritten", "bvalid", "streams")


from ._cameras import _CAM_KEYS, _CAM_KEYS_MAX, _camera_key, tag_camera          # noqa: E402,F401 - camera identity (split out in round 6)


_CAM_HINTS = {}                                  # (camera content, W, H, device) -> _CamHints, least recently used first
_CAM_HINTS_MAX = 1024


def _hints_for(cam_key, W, H, device, n_tiles):
    key = (cam_key, W, H, device.index)
    h = _CAM_HINTS.pop(key, None)
    if h is None:
        if len(_CAM_HINTS) >= _CAM_HINTS_MAX:                    # the least recently used quarter goes; their pinned words are
            for k in list(_CAM_HINTS)[: _CAM_HINTS_MAX // 4]:    # handed out again (a render of theirs finished long ago: the
                _release_long_words(_CAM_HINTS.pop(k))           # records in use sit at the other end of the table)
        h = _CamHints()
        h.cost = [torch.zeros(n_tiles, dtype=torch.int32, device=device) for _ in range(2)] if TILE_COST_HINT else None
        h.cur, h.cost_valid = 0, False
        # ... and two of per-QUADRANT times of the blend backward's waves (4 words per tile)
        h.bcost = [torch.zeros(4 * n_tiles, dtype=torch.int32, device=device) for _ in range(2)] if TILE_COST_HINT else None
        h.bcur, h.bwritten, h.bvalid = 0, False, False
        # raw streams these buffers have been used on: the one they were allocated on + every other one, each told to the caching
        # allocator once (Tensor.record_stream), so that an evicted record's memory is not handed out while a kernel of another
        # stream still writes it (the operator is re-entrant per stream since round 5)
        h.streams = {_stream(device)}
        h.long_np = h.long_ptr = h.slot = None
        if SKIP_IDLE_RARE_SORT or RARE_8WAVE:
            h.slot, h.long_np, h.long_ptr = _long_words()
    _CAM_HINTS[key] = h                                          # (re-inserted: most recently used last)
    return h


_FRAME_CACHE = {}
# ScgFrame.long_lists_out words come from ONE pinned allocation per process (a pinned allocation per camera would cost a render
# loop over hundreds of distinct cameras ~50 us each): one slot of two words per live hint record (_CAM_HINTS), returned to
# the free list when the record is evicted.  The words are a HINT (which sort kernels to launch) — a wrong one costs time,
# never a result (the forward blend sorts a list nobody sorted for it,
# tests/test_gpu_parity.py::test_skipped_rare_sort_launch_...).
_LONG_POOL = None
_LONG_FREE = []


def _long_words():
    """(slot index, numpy view of the slot's two words, device-visible address of the slot)."""
    global _LONG_POOL
    if _LONG_POOL is None:
        t = torch.full((2 * (_CAM_HINTS_MAX + 1),), -1, dtype=torch.int32).pin_memory()
        _LONG_POOL = (t, t.numpy(), t.data_ptr())
        _LONG_FREE.extend(range(_CAM_HINTS_MAX, -1, -1))
    _t, arr, base = _LONG_POOL
    if not _LONG_FREE:                   # records that left the table without passing _release_long_words (a cleared table)
        used = {h.slot for h in _CAM_HINTS.values()}
        _LONG_FREE.extend(i for i in range(_CAM_HINTS_MAX, -1, -1) if i not in used)
    i = _LONG_FREE.pop()
    view = arr[2 * i: 2 * i + 2]
    view[:] = -1                         # "no render of this camera has completed yet"
    return i, view, base + 8 * i


def _release_long_words(h):
    if h.slot is not None:
        _LONG_FREE.append(h.slot)
        h.slot = h.long_np = h.long_ptr = None


# ---- module switches (round 5: no environment variables in the product — tests and the same-process A/B tool
# tools/ab_inproc.py flip these attributes; a caller never needs to) ------------------------------------------------------
# Order the blend kernels' tiles by what they cost the last time the same camera was rendered (False: by list length always).
TILE_COST_HINT = True
# ... and the blend backward's (tile, quadrant) waves by how long each took in the camera's previous backward
# (ScgFrame.bwd_cost_in; False: they follow the tiles' order)
BWD_ORDER_HINT = True
# Skip the launch of the rare-size sort kernel while the previous render of the same camera found no list beyond the forward
# blend's own sort (scg_raster.h SCG_FORWARD_SKIP_RARE_SORT; False: always launch it).
SKIP_IDLE_RARE_SORT = True
# ... and partition the lists beyond 4 096 entries (kSort8Max) by depth first when that render found very long ones
# (SCG_FORWARD_SPLIT_LONG_LISTS; False: one workgroup sorts each long list as before)
SPLIT_LONG_LISTS = True
# ... sorted, like the other lists beyond the forward blend's own sort, by 8-wave workgroups three per compute unit
# (SCG_FORWARD_RARE_8WAVE; False: the 16-wave rare-size kernel, one workgroup per compute unit, as on a first render)
RARE_8WAVE = True


def _rare_options(words) -> int:
    """scg_forward option bits from what the camera's previous render left in ScgFrame.long_lists_out."""
    if words is None or words[0] < 0:
        return 0                                             # nothing known yet: the rare-size kernel as in the staged calls
    if words[0] == 0:
        return 8 if SKIP_IDLE_RARE_SORT else 0               # SCG_FORWARD_SKIP_RARE_SORT
    # very long lists (beyond 16 384 entries): split everything beyond 4 096 by depth + 8-wave work-list sort; without them the
    # 16-wave rare-size kernel finishes its handful of lists in one round (measured: 30 %-clustered scene 64.6 vs 73.3 us)
    if RARE_8WAVE and SPLIT_LONG_LISTS and words[1] > 0:
        return 16 | 32
    return 0


def _frame_for(settings: GaussianRasterizationSettings, P: int, M: int, device, forward: bool = False) -> _Frame:
    """ScgFrame structs are cached by what they contain (pointers + scalars): building the ctypes struct costs more
    host time than some of the kernels it describes take.  Only frames whose four small tensors are used as they
    are (fp32, contiguous, on the device) are cached — the cached frame keeps them alive, so their addresses cannot be
    recycled, and their CONTENT is read by the kernels at launch time, not here."""
    small = (settings.viewmatrix, settings.projmatrix, settings.campos, settings.bg)
    for t in small:
        if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.device == device and t.is_contiguous()
                and t.numel() > 0):
            return _Frame(settings, P, M, device)
    key = (small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr(), small[3].data_ptr(),
           settings.image_height, settings.image_width, settings.tanfovx, settings.tanfovy,
           settings.scale_modifier, settings.sh_degree, settings.prefiltered, settings.debug, P, M, device.index)
    fr = _FRAME_CACHE.get(key)
    if fr is None:
        if len(_FRAME_CACHE) > 256:
            _FRAME_CACHE.clear()
        fr = _FRAME_CACHE[key] = _Frame(settings, P, M, device)
    if forward:
        fr.next_forward(device)
    return fr


def _stream(device) -> int:
    """Raw hipStream_t of torch's current stream on `device` (the C call behind torch.cuda.current_stream)."""
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already the current device (the common case skips it)."""

    def __init__(self, device):
        self.ctx = None
        if device.index is not None and torch._C._cuda_getDevice() != device.index:
            self.ctx = torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _require_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.ScgError("scgaussian_amd rasterizer needs tensors on a ROCm GPU ('cuda' device); "
                            "there is no CPU path in the product (the CPU oracle lives under oracle/, tests only)")


def _no_timer(name: str):
    return contextlib.nullcontext()


_ACTIVE_TIMER: Callable = _no_timer


def set_stage_timer(timer: Optional[Callable]):
    """Install a StageTimer used by every rasterizer call that does not pass its own (bench.py); None removes it."""
    global _ACTIVE_TIMER
    _ACTIVE_TIMER = timer if timer is not None else _no_timer


class StageTimer:
    """Optional per-stage timing with events recorded on the stream the kernels are launched on (torch's
    current stream).  Usage: t = StageTimer(); forward_stages(..., timer=t); t.summary() after a sync.

    The one-call entry points (scg_forward / scg_backward) record their stages' events themselves
    (include/scg_raster.h ScgStageEvents): `stage_events()` hands them a struct of raw hipEvent_t pairs from this
    timer's pool, so the per-stage times of the REAL fast path are measured, not those of a staged replay."""

    FORWARD = ("geometry_forward", "binning", "blend_forward")
    BACKWARD = ("blend_backward", "geometry_backward")

    def __init__(self, only=None):
        self.events = {}
        self.raw = {}                                            # name -> [(begin handle, end handle)]
        self.only = set(only) if only is not None else None      # restrict to these stages (less event traffic)
        self._pool = []
        self._raw_pool = {}                                      # device index -> [hipEvent_t]: an event belongs to a device
        self._structs = []                                       # keeps the ctypes structs alive until summary()

    def _event(self):
        if not self._pool:                                       # created in batches, off the per-stage path
            self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(256)]
        return self._pool.pop()

    def _raw_event(self, dev):
        pool = self._raw_pool.setdefault(dev, [])
        if not pool:                                             # created on the CURRENT device: callers are inside
            lib = _lib.load()                                    # `with _on_device(dev)`
            for _ in range(256):
                h = C.c_void_p()
                check(lib.scg_event_create(C.byref(h), 1), "scg_event_create")
                pool.append(h.value)
        return pool.pop()

    def stage_events(self, direction: str):
        """ScgStageEvents (by reference) for one scg_forward ('forward') / scg_backward ('backward') call, or None."""
        names = self.FORWARD if direction == "forward" else self.BACKWARD
        wanted = [i for i, n in enumerate(names) if self.only is None or n in self.only]
        if not wanted:
            return None
        ev = _lib.ScgStageEvents()
        dev = torch._C._cuda_getDevice()
        for i in wanted:
            a, b = self._raw_event(dev), self._raw_event(dev)
            ev.begin[i], ev.end[i] = a, b
            self.raw.setdefault(names[i], []).append((a, b, dev))
        self._structs.append(ev)
        return C.byref(ev)

    def __call__(self, name: str):
        if self.only is not None and name not in self.only:
            return contextlib.nullcontext()
        return self._timed(name)

    @contextlib.contextmanager
    def _timed(self, name: str):
        a, b = self._event(), self._event()
        a.record()
        try:
            yield
        finally:
            b.record()
            self.events.setdefault(name, []).append((a, b))

    def summary(self):
        """{stage: (mean ms, calls)}; synchronises."""
        torch.cuda.synchronize()
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in self.events.items()}
        if self.raw:
            lib = _lib.load()
            ms = C.c_float()
            for k, pairs in self.raw.items():
                for a, b, _dev in pairs:
                    check(lib.scg_event_elapsed_ms(a, b, C.byref(ms)), "scg_event_elapsed_ms")
                    out.setdefault(k, []).append(ms.value)
        return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}

    def reset(self):
        for pairs in self.raw.values():                          # completed events go back to their device's pool
            for a, b, dev in pairs:
                self._raw_pool.setdefault(dev, []).extend((a, b))
        self.events, self.raw, self._structs = {}, {}, []

    def close(self):
        """Destroy the raw hipEvents of this timer (pool and recorded pairs)."""
        self.reset()
        pools, self._raw_pool = self._raw_pool, {}
        try:
            lib = _lib.load()
            for pool in pools.values():
                for h in pool:
                    lib.scg_event_destroy(h)
        except Exception:                                        # interpreter shutdown: the driver reclaims them
            pass

    def __del__(self):
        self.close()


_ELEMENT_SIZE = {torch.float32: 4, torch.int32: 4, torch.uint8: 1, torch.int64: 8}


class _Arena:
    """One device allocation carved into aligned sub-buffers addressed by raw pointer (internal state never
    becomes a tensor unless a caller asks for a view): fewer torch allocations / tensor objects per call."""

    def __init__(self, sizes, device):
        self.offsets = []
        off = 0
        for nbytes in sizes:
            self.offsets.append(off)
            off += (int(nbytes) + 255) // 256 * 256
        self.buf = torch.empty((max(off, 256),), dtype=torch.uint8, device=device)
        self.base = self.buf.data_ptr()

    def ptr(self, i: int) -> int:
        return self.base + self.offsets[i]

    def view(self, i: int, shape, dtype) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ELEMENT_SIZE[dtype]
        return self.buf[self.offsets[i]: self.offsets[i] + nbytes].view(dtype).view(shape)


class _SpecState:
    """Per-device state of the speculative launch: pinned host memory for the partial sums of num_rendered, an event,
    and the capacity (upper bound of num_rendered) in use per (P, W, H)."""

    def __init__(self, device):
        self.hint = {}                   # (P, W, H) -> capacity: the latest bound of ANY camera of this shape
        self.cam_hint = {}               # (W, H, camera) -> (capacity, num_rendered, P it was taken at)
        # pinned host scratch of the forwards IN FLIGHT on this device: one _PinnedSums per forward, taken from / returned to a
        # free list (round 5: the operator is re-entrant — two threads, each on its own stream, may be inside a forward at the
        # same time, as with the upstream extension; one shared scratch made the second one raise)
        self.free = []
        self.pool_lock = threading.Lock()
        self.plans = {}                  # (P, W, H, capacity) -> _Plan
        self.pending = {}                # (W, H, camera) -> [_CountWord]: no-host-read renders whose count nobody has looked at yet

    def take(self, nbytes: int) -> "_PinnedSums":
        """Pinned host memory used as the geometry stage's scratch for ONE forward: the kernel writes its per-workgroup
        partial sums of num_rendered straight to the host, so the speculative path needs neither a total kernel nor a D2H
        copy.  give_back() when the count has been read."""
        with self.pool_lock:
            ps = self.free.pop() if self.free else None
        if ps is None or ps.nbytes < nbytes:
            ps = _PinnedSums(nbytes)
        return ps

    def give_back(self, ps: "_PinnedSums"):
        with self.pool_lock:
            if len(self.free) < 8:
                self.free.append(ps)

    def plan(self, lib, P, W, H, cap):
        key = (P, W, H, cap)
        pl = self.plans.get(key)
        if pl is None:
            if len(self.plans) > 64:
                self.plans.clear()
            pl = self.plans[key] = _Plan(lib, P, W, H, cap)
        return pl



class _PinnedSums:
    """The pinned words one forward's geometry kernel writes its partial sums of num_rendered to, and the events of the paths
    that wait on one (the staged path; the one-call path with EVENTLESS_WAIT off)."""
    __slots__ = ("t", "np", "ptr", "nbytes", "event", "raw_event")

    def __init__(self, nbytes: int):
        self.t = torch.zeros(((nbytes + 3) // 4 + 1024,), dtype=torch.int32).pin_memory()
        self.np = self.t.numpy()
        self.ptr = self.t.data_ptr()
        self.nbytes = self.t.numel() * 4
        self.event = None                # torch.cuda.Event of the staged path
        self.raw_event = None            # hipEvent_t (timing disabled) of the one-call path

    def torch_event(self):
        if self.event is None:
            self.event = torch.cuda.Event()
        return self.event

    def event_handle(self):
        if self.raw_event is None:
            h = C.c_void_p()
            check(_lib.load().scg_event_create(C.byref(h), 0), "scg_event_create")
            self.raw_event = h.value
        return self.raw_event


class _Plan:
    """Workspace layout of one (P, W, H, capacity): byte offsets reported by the library, looked up once."""

    __slots__ = ("total", "final_T", "n_contrib", "point_list", "ranges", "splats", "rects", "depth_keys", "clamped",
                 "partial_bytes", "accepts", "fused", "_shape")

    def __init__(self, lib, P, W, H, cap):
        L = _lib.ScgWorkspaceLayout()
        check(lib.scg_workspace_layout(P, cap, W, H, C.byref(L)), "scg_workspace_layout")
        self.total = int(L.total)
        for k in ("final_T", "n_contrib", "point_list", "ranges", "splats", "rects", "depth_keys", "clamped"):
            setattr(self, k, int(getattr(L, k)))
        self.partial_bytes = int(L.partial_words) * 4
        self.accepts = lib.scg_binning_accepts_bound(cap, W, H, 0) == 1
        # ScgFrame.long_lists_out is written by the forward blend that sorts its own tiles; a frame whose sort is a kernel of
        # its own (the library's A/B bit, FUSED_SORT = False) leaves the words alone (fused[options]: which one runs)
        self.fused = {}
        self._shape = (cap, W, H)

    def sorts_in_blend(self, lib, options: int) -> bool:
        v = self.fused.get(options)
        if v is None:
            v = self.fused[options] = lib.scg_forward_sorts_in_blend(*self._shape, options) == 1
        return v


from ._counts import _SPEC_STATE, _capacity_for, _next_capacity          # noqa: E402 - the capacity policy: _counts.py
# what the speculation cost so far (speculation_stats(); bench.py `moving_scene`): forwards on the one-call path, how many of them
# had to be repeated because num_rendered exceeded the capacity, forwards that took the staged path (first sight of a shape, images
# beyond the tile-first binning), and how many one-call forwards had a launch-order hint of their camera's previous render
_SPEC_STATS = {"one_call_forwards": 0, "overflow_retries": 0, "staged_forwards": 0, "tile_cost_hints": 0, "bwd_order_hints": 0}


def speculation_stats(reset: bool = False) -> dict:
    """Counters of the speculative launch since the process started (or the last reset): see _SPEC_STATS."""
    out = dict(_SPEC_STATS)
    if reset:
        for k in _SPEC_STATS:
            _SPEC_STATS[k] = 0
    return out


SPECULATIVE_LAUNCH = True       # module switch (tests flip it to cover both paths)
# scg_forward lets the forward blend sort the tiles' lists itself and the geometry kernel build the binning stage's slice
# histograms; False passes the library's A/B bits (csrc/scg_debug.h: SCG_DEBUG_SEPARATE_SORT / _HIST — not part of the public
# header) so that the suite can hold the fused kernels against the separate ones bit for bit
FUSED_SORT = True
FUSED_HIST = True
# scg_forward's partial sums of num_rendered are collected by watching the pinned words (SCG_FORWARD_ARM_PARTIAL_SUMS) instead
# of waiting on an event recorded behind the geometry kernel (module switch for same-process A/B runs)
EVENTLESS_WAIT = True


def _spec_state(device) -> _SpecState:
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _SPEC_STATE.get(key)
    if st is None:
        st = _SPEC_STATE[key] = _SpecState(device)
    return st


def forward_stages(settings: GaussianRasterizationSettings, means3D, opacities, shs=None, colors_precomp=None,
                   scales=None, rotations=None, cov3D_precomp=None, want_keys: bool = False,
                   timer: Optional[Callable] = None, binning_algo: int = 0, capacity_hint: Optional[int] = None,
                   prepare_backward: bool = False):
    """Run the forward stages through the C ABI and return every intermediate (used by the autograd
    function and, with want_keys=True, by the parity tests).

    Host/GPU overlap: the binning buffers are sized by num_rendered, which only the GPU knows.  Instead of
    stalling on that read, stages 2-3 are enqueued at once with an upper-bound guess (1.25 x the largest count seen
    for this problem shape); num_rendered travels to pinned host memory on the same stream and is checked after
    the launches.  A guess that was too small (rare) re-runs stages 2-3 with the exact size.  `capacity_hint`
    overrides the guess (tests)."""
    _require_cuda(means3D)
    spec = _spec_state(means3D.device)
    return _forward_stages_locked(spec, settings, means3D, opacities, shs, colors_precomp, scales, rotations,
                                  cov3D_precomp, want_keys, timer, binning_algo, capacity_hint, prepare_backward)


def _forward_stages_locked(spec, settings, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                           want_keys, timer, binning_algo, capacity_hint, prepare_backward):
    lib = _lib.load()
    _SPEC_STATS["staged_forwards"] += 1
    timer = timer or _ACTIVE_TIMER
    dev = means3D.device
    means3D = _f32c(means3D, dev)
    P = 0 if means3D is None else means3D.shape[0]
    opacities = _f32c(opacities, dev)
    shs = _f32c(shs, dev)
    colors_precomp = _f32c(colors_precomp, dev)
    scales = _f32c(scales, dev)
    rotations = _f32c(rotations, dev)
    cov3D_precomp = _f32c(cov3D_precomp, dev)
    M = shs.shape[1] if shs is not None else 0
    fr = _frame_for(settings, P, M, dev, forward=True)
    H, W = fr.H, fr.W
    with _on_device(dev):
        stream = _stream(dev)
        # per-Gaussian state: [0] splats  [1] rects  [2] depth_keys  [3] clamped  [4] geometry scratch  [5] num_rendered
        gscratch = lib.scg_geometry_scratch_bytes(P)
        ga = _Arena([P * SPLAT_FLOATS * 4, P * 8, P * 4, P, gscratch, 4], dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        key = (P, W, H)
        guess = capacity_hint if capacity_hint is not None else spec.hint.get(key)
        speculative = (SPECULATIVE_LAUNCH or capacity_hint is not None) and guess is not None and P > 0 and \
            not want_keys and \
            lib.scg_binning_accepts_bound(int(guess), W, H, binning_algo) == 1
        pinned = spec.take(gscratch) if speculative else None
        with timer("geometry_forward"):
            # speculative: partial sums of num_rendered go straight to pinned host memory, no on-device total
            check(lib.scg_geometry_forward(fr.ref, ptr(means3D), ptr(opacities), ptr(shs), ptr(colors_precomp),
                                           ptr(scales), ptr(rotations), ptr(cov3D_precomp), ga.ptr(0), ptr(radii),
                                           ga.ptr(3), ga.ptr(1), ga.ptr(2), None if speculative else ga.ptr(5),
                                           pinned.ptr if speculative else ga.ptr(4), gscratch, stream),
                  "scg_geometry_forward")
        if speculative:
            pinned.torch_event().record()
            R = None
            cap = int(guess)
        else:
            nr = ga.view(5, (1,), torch.int32)
            R = int(nr.item()) & 0xFFFFFFFF          # the one host read of the path (sizes the binning buffers)
            cap = R

        img = torch.empty((5, H, W), dtype=torch.float32, device=dev)          # one allocation, three views
        color, depth, alpha = img[0:3], img[3:4], img[4:5]

        # gradient records of the coming backward: cleared by the forward blend kernel (no memset launch later)
        dsplats = torch.empty((P, DSPLAT_FLOATS), dtype=torch.float32, device=dev) if prepare_backward and P > 0 else None

        def bin_and_blend(capacity):
            # [0] point_list  [1] ranges  [2] final_T  [3] n_contrib  [4] binning scratch  [5] keys (debug)
            scratch_bytes = lib.scg_binning_scratch_bytes(P, capacity, W, H, binning_algo)
            ba = _Arena([capacity * 4, lib.scg_ranges_words(W, H) * 4, H * W * 4, H * W * 4, scratch_bytes,
                         capacity * 8 if want_keys else 0], dev)
            with timer("binning"):
                check(lib.scg_binning(fr.ref, capacity, ga.ptr(1), ga.ptr(2), ba.ptr(0), ba.ptr(1),
                                      ba.ptr(5) if want_keys else None, binning_algo, ba.ptr(4), scratch_bytes, stream),
                      "scg_binning")
            with timer("blend_forward"):
                check(lib.scg_blend_forward(fr.ref, ba.ptr(1), ba.ptr(0), ga.ptr(0), ptr(color), ptr(depth), ptr(alpha),
                                            ba.ptr(2), ba.ptr(3), ptr(dsplats), stream), "scg_blend_forward")
            return ba

        ba = bin_and_blend(cap)
        # everything that does not need num_rendered is done BEFORE the wait: what follows the wait sits on the
        # critical path of a training step (the GPU has ~0.13 ms of forward queued, the host needs longer than that
        # to get from here to the launch of the backward)
        out = _LazyViews(dict(color=color, depth=depth, alpha=alpha, radii=radii, num_rendered=R, dsplats_zeroed=dsplats,
                              arenas=(ga, ba), capacity=cap, n_tiles=fr.n_tiles, hw=(H, W), P=P, frame=fr,
                              ptrs=dict(splats=ga.ptr(0), clamped=ga.ptr(3), point_list=ba.ptr(0), ranges=ba.ptr(1),
                                        final_T=ba.ptr(2), n_contrib=ba.ptr(3)),
                              inputs=(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)),
                         want_keys)
        if speculative:
            pinned.torch_event().synchronize()
            R = int(pinned.np[: (P + 255) // 256].sum(dtype="int64"))
            spec.give_back(pinned)
            if R > cap:                              # the guess was too small: lists were clipped, run again
                ba = bin_and_blend(R)
                cap = R
                out["arenas"] = (ga, ba)
                out["capacity"] = cap
                out["ptrs"].update(point_list=ba.ptr(0), ranges=ba.ptr(1), final_T=ba.ptr(2), n_contrib=ba.ptr(3))
            out["num_rendered"] = R
        if capacity_hint is None:
            spec.hint[key] = _next_capacity(spec.hint.get(key), R)
    return out


class _LazyViews(dict):
    """forward_stages result: internal buffers become tensors only when somebody indexes them (tests, tools)."""

    _SPECS = {
        "splats": (0, 0, lambda d: (d["P"], SPLAT_FLOATS), torch.float32),
        "rects": (0, 1, lambda d: (d["P"], 2), torch.int32),
        "depth_keys": (0, 2, lambda d: (d["P"],), torch.int32),
        "clamped": (0, 3, lambda d: (d["P"],), torch.uint8),
        "point_list": (1, 0, lambda d: (d["num_rendered"],), torch.int32),
        "ranges": (1, 1, lambda d: (d["n_tiles"], 2), torch.int32),
        "final_T": (1, 2, lambda d: d["hw"], torch.float32),
        "n_contrib": (1, 3, lambda d: d["hw"], torch.int32),
        "keys_sorted": (1, 5, lambda d: (d["num_rendered"],), torch.int64),
    }

    def __init__(self, d, want_keys):
        super().__init__(d)
        self._want_keys = want_keys

    def __missing__(self, k):
        spec = self._SPECS.get(k)
        if spec is None or (k == "keys_sorted" and not self._want_keys):
            if k == "keys_sorted":
                return None
            raise KeyError(k)
        arena_i, slot, shape_fn, dtype = spec
        v = dict.__getitem__(self, "arenas")[arena_i].view(slot, shape_fn(self), dtype)
        self[k] = v
        return v


# Order of the segments in the gradient arena.  The SH gradients come LAST of the usual five: the other four (11 floats per
# Gaussian) are then one contiguous span, which parallel.GradBucket all-reduces in place when only the active SH coefficients
# of a degree-limited step are exchanged (round 5).
_GRAD_ORDER = ("means3D", "opacities", "scales", "rotations", "shs", "colors_precomp", "cov3D_precomp")
_GRAD_INDEX = (0, 1, 4, 5, 2, 3, 6)            # position of each of those in the `inputs` 7-tuple
_GRAD_LAYOUTS = {}


# ---- gradient arenas that are KEPT from step to step ------------------------------------------------------------------------
# A backward writes every parameter gradient into one flat fp32 arena and autograd keeps views of it as the .grad tensors.  The
# arenas of a layout are pooled: one is handed out again once nobody outside the pool references its storage any more (the
# previous step's .grad tensors are gone: optimizer.zero_grad(set_to_none=True), p.grad = None).  What that buys (round 6): the
# arena REMEMBERS that its SH-coefficient gradients above the active degree hold zeros — written by the kernel the first time —
# and the next backward is told to leave them alone (SCG_BACKWARD_SH_TAIL_ZERO): at degree 0 the geometry backward stores 12
# instead of 192 bytes of SH gradient per Gaussian (the reference trains 1 000 iterations at degree 0 and 1 000 at degree 1,
# train.py:129).  The promise holds while (a) nobody but the pool and this step's autograd references the storage and (b) no
# torch operation wrote through any view of it since (the views share the arena's version counter: zero_grad(set_to_none=False),
# an in-place all-reduce or clip bump it) — otherwise the kernel writes the zeros again.
ARENA_POOL = True                        # module switch (tests / A-B runs)
_ARENA_POOLS = {}                        # layout key -> [ _PooledArena ]
_ARENA_POOL_DEPTH = 3                    # arenas kept per layout (a step holds one; gradient accumulation over two steps: two)
_use_count = getattr(torch._C, "_storage_Use_Count", None)


class _PooledArena:
    __slots__ = ("arena", "storage", "version", "zero_from", "cstruct")

    def __init__(self, total, dev):
        self.arena = torch.empty((total,), dtype=torch.float32, device=dev)
        self.storage = self.arena.untyped_storage()
        self.version = -1
        self.zero_from = None            # SH coefficients >= this index hold zeros (None: unknown)
        self.cstruct = None              # model_path: the ScgModelGrads struct of this arena's segments

    def free(self) -> bool:
        return _use_count(self.storage._cdata) == 2          # the arena tensor and the wrapper above, nobody else


def _take_arena(key, total, dev):
    """(arena tensor, pooled record or None).  Not pooled: no use-count query in this torch, the switch is off, or a stream
    capture is in progress (a captured step's buffers belong to its graph's memory pool and are replayed in place)."""
    if not ARENA_POOL or _use_count is None or (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        return torch.empty((total,), dtype=torch.float32, device=dev), None
    pool = _ARENA_POOLS.get(key)
    if pool is None:
        if len(_ARENA_POOLS) > 32:
            _ARENA_POOLS.clear()
        pool = _ARENA_POOLS[key] = []
    for pa in pool:
        if pa.free():
            return pa.arena, pa
    pa = _PooledArena(total, dev)
    if len(pool) < _ARENA_POOL_DEPTH:
        pool.append(pa)
    return pa.arena, pa


def _sh_tail_promise(pa, n_active: int) -> int:
    """SCG_BACKWARD_SH_TAIL_ZERO (2) when the pooled arena is known to hold zeros in every SH coefficient >= n_active; records
    what this backward leaves behind (called once per backward, before the launch)."""
    if pa is None:
        return 0
    ok = pa.zero_from is not None and pa.zero_from <= n_active and pa.version == pa.arena._version
    pa.zero_from = n_active              # after this backward: written below n_active, zeros (kept or written) from there on
    pa.version = pa.arena._version
    return 2 if ok else 0



def _grad_outputs(inputs, into, d_means2D_out, dev):
    """Output tensors of a backward: every parameter gradient a view of ONE flat fp32 arena (16-byte aligned segments;
    data-parallel training all-reduces the arena in place instead of packing / unpacking a bucket, parallel.GradBucket) —
    or, when `into` is the result of an earlier backward over the same inputs, those very tensors (the kernel then ADDS
    to them: views-per-step accumulation).  dL/dmeans2D belongs to the view: always a tensor of its own."""
    d_means2D = d_means2D_out if d_means2D_out is not None else torch.empty_like(inputs[0])
    if into is not None:
        out = dict(into)
        out["means2D"] = d_means2D
        return out
    # the layout (segment sizes, shapes) depends on the inputs' shapes only: looked up, not rebuilt per step
    key = tuple(None if t is None else t.shape for t in inputs)
    lay = _GRAD_LAYOUTS.get(key)
    if lay is None:
        if len(_GRAD_LAYOUTS) > 64:
            _GRAD_LAYOUTS.clear()
        names, sizes, shapes, exact = [], [], [], []
        for n, i in zip(_GRAD_ORDER, _GRAD_INDEX):
            t = inputs[i]
            if t is not None:
                names.append(n)
                sizes.append((t.numel() + 3) // 4 * 4)
                shapes.append(tuple(t.shape))
                exact.append(sizes[-1] == t.numel())
        lay = _GRAD_LAYOUTS[key] = (tuple(names), sizes, tuple(shapes), tuple(exact), max(sum(sizes), 4))
    names, sizes, shapes, exact, total = lay
    arena, pooled = _take_arena(("tensors", key, dev.index), total, dev)
    out = dict.fromkeys(_GRAD_ORDER)
    out["_pooled"] = pooled
    if names:
        for n, v, shp, ex in zip(names, arena.split_with_sizes(sizes) if total == sum(sizes) else
                                 arena[: sum(sizes)].split_with_sizes(sizes), shapes, exact):
            out[n] = (v if ex else v[: math.prod(shp)]).view(shp)
    out["means2D"] = d_means2D
    return out


def backward_stages(settings: GaussianRasterizationSettings, inputs, saved, dL_dcolor, dL_ddepth, dL_dalpha,
                    want_dsplats: bool = False, timer: Optional[Callable] = None, into=None, d_means2D_out=None):
    """Blend backward + geometry backward through the C ABI.  `inputs` is the 7-tuple of contiguous fp32
    input tensors, `saved` the forward state: {"ptrs": raw device pointers of splats / clamped / point_list /
    ranges / final_T / n_contrib, "arenas": the allocations that own them, "radii": tensor} — a forward_stages
    result can be passed as is.  `into`: the dict an earlier call returned for the same inputs — the parameter
    gradients of this view are added to it in the kernel (scg_geometry_backward accumulate)."""
    lib = _lib.load()
    timer = timer or _ACTIVE_TIMER
    means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp = inputs
    dev = means3D.device
    P = means3D.shape[0]
    M = shs.shape[1] if shs is not None else 0
    fr = saved.get("frame") if isinstance(saved, dict) else None
    if fr is None:
        fr = _frame_for(settings, P, M, dev)
    if fr.hints is not None and fr.hints.bcost is not None:
        fr.hints.bwritten = True
    H, W = fr.H, fr.W
    dL_dcolor = _f32c(dL_dcolor, dev)
    if dL_dcolor is None:
        dL_dcolor = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
    dL_ddepth = _f32c(dL_ddepth, dev)
    dL_dalpha = _f32c(dL_dalpha, dev)
    with _on_device(dev):
        stream = _stream(dev)
        # the forward may have left a cleared gradient-record buffer behind (usable once)
        dsplats = saved.pop("dsplats_zeroed", None) if isinstance(saved, dict) else None
        prezeroed = dsplats is not None
        if dsplats is None:
            dsplats = torch.empty((P, DSPLAT_FLOATS), dtype=torch.float32, device=dev)
        with timer("blend_backward"):
            sp = saved["ptrs"]
            check(lib.scg_blend_backward(fr.ref, sp["ranges"], sp["point_list"], sp["splats"], sp["final_T"],
                                         sp["n_contrib"], ptr(dL_dcolor), ptr(dL_ddepth), ptr(dL_dalpha),
                                         ptr(dsplats), int(prezeroed), stream), "scg_blend_backward")
        out = _grad_outputs(inputs, into, d_means2D_out, dev)
        flags = 1 if into is not None else _sh_tail_promise(out.get("_pooled"), (fr.c.sh_degree + 1) ** 2)
        with timer("geometry_backward"):
            check(lib.scg_geometry_backward(fr.ref, ptr(means3D), ptr(opacities), ptr(shs), ptr(colors_precomp),
                                            ptr(scales), ptr(rotations), ptr(cov3D_precomp), ptr(saved["radii"]),
                                            saved["ptrs"]["clamped"], ptr(dsplats), ptr(out["means3D"]),
                                            ptr(out["means2D"]), ptr(out["opacities"]), ptr(out["shs"]),
                                            ptr(out["colors_precomp"]), ptr(out["scales"]), ptr(out["rotations"]),
                                            ptr(out["cov3D_precomp"]), flags, stream),
                  "scg_geometry_backward")
    if want_dsplats:
        out["dsplats"] = dsplats
    return out


# ---------------------------------------------------------------------------------------------------------------------
# a forward that never reads the host (round 6; SURVEY §8b "or none, if the caller passes a capacity and the kernel reports
# overflow"; ScgFrame.num_rendered_out)
# ---------------------------------------------------------------------------------------------------------------------
# The default forward spins on the pinned partial sums of num_rendered before it returns (scg_wait_num_rendered): it hands back
# a result that is known to be complete, and it cannot be captured in a hipGraph.  With NO_HOST_READ (or whenever torch's current
# stream is capturing) the forward is launched with the capacity the camera's earlier renders established and returns at once;
# the binning stage leaves the count it found in one pinned word (ScgFrame.num_rendered_out) and the binding looks at that word
# the NEXT time it renders the camera (or when somebody asks: settle_counts()): a count beyond the capacity means that render's
# lists were clipped (its images and gradients miss the tail of the tile order) — it is counted in overflow_stats(), the camera's
# capacity is raised, and the next render is complete again: one step late, as the reference's own loop would notice a bad
# iteration one `loss.item()` late (train.py:176).  graph_step.CapturedStep builds on this: forward + backward captured once
# and replayed, the word checked before every replay, the step re-captured with room for the count after an overflow.
NO_HOST_READ = False
_CAPTURE_RECORD = None                   # graph_step's record of the forwards captured right now (None: no capture of ours)


@contextlib.contextmanager
def no_host_read(enabled: bool = True):
    """`with no_host_read():` — every forward inside is launched without waiting for num_rendered (see NO_HOST_READ)."""
    global NO_HOST_READ
    prev, NO_HOST_READ = NO_HOST_READ, bool(enabled)
    try:
        yield
    finally:
        NO_HOST_READ = prev


# the pinned count words, what a settled count does to a camera's capacity, the counters: _counts.py
from ._counts import (_ANON_CAPTURED, _COUNT_ARMED, _COUNT_FREE, _COUNT_SLOTS, _OVERFLOW, _QUARANTINE, _CountWord,   # noqa: E402,F401
                      _count_pool, _count_word, _settle_camera, _settle_word, overflow_stats, settle_counts)


# ---------------------------------------------------------------------------------------------------------------------
# the fast path: ONE C-ABI call per direction (include/scg_raster.h scg_forward / scg_backward)
# ---------------------------------------------------------------------------------------------------------------------
def forward_fused(settings: GaussianRasterizationSettings, means3D, opacities, shs, colors_precomp, scales, rotations,
                  cov3D_precomp, prepare_backward: bool, timer: Optional[Callable] = None, model=None):
    """Stages 1-3 in one library call, laid out in one workspace allocation, enqueued without a host read: the capacity
    (upper bound of num_rendered) comes from the previous calls of this problem shape.  Returns None when the fast path
    does not apply (no capacity known yet, image too large for the tile-first binning): the caller then takes the
    staged path, which also establishes the capacity.  Otherwise (color, radii, depth, alpha, state).
    `model` (model_path._ModelArgs): the reference model's raw parameter tensors in place of the seven activated inputs
    (scg_forward_model); a model's first render starts from a generous bound instead of the staged path."""
    dev = means3D.device if model is None else model.device
    spec = _spec_state(dev)
    P = means3D.shape[0] if model is None else model.P
    H, W = int(settings.image_height), int(settings.image_width)
    # the capacity is remembered per CAMERA (views of one scene can differ by more than 2x in num_rendered: a bound
    # shared by all of them would shrink after the cheap view and overflow on the expensive one, every other step); a
    # camera seen for the first time starts from the latest bound of any camera of this shape
    # (the camera = the content of its view matrix, _camera_key: an address can be recycled for another camera)
    cam = _camera_key(settings.viewmatrix)
    ckey = (W, H, cam)
    # a forward that must not read the host: asked for (NO_HOST_READ) or inside a stream capture (a host read cannot be captured)
    capturing = torch.cuda.is_current_stream_capturing()
    no_read = NO_HOST_READ or capturing
    if spec.pending:
        _settle_camera(spec, ckey)                           # counts of this camera's earlier no-host-read renders that have arrived
    # ... keyed WITHOUT the Gaussian count: densification changes P every ~100 iterations, and a per-camera entry that
    # died with every change of P would leave hundreds of cameras on the shared fallback again.  The entry remembers the
    # count it was taken at; after a change of P the camera's last num_rendered is rescaled by the ratio of the counts.
    ent = spec.cam_hint.get(ckey)
    if ent is not None:
        cap_c, R_c, P_c = ent
        cap = cap_c if P_c == P else _capacity_for(int(R_c * (P / max(P_c, 1))) + 1)
        if capturing:                                        # a captured step keeps its capacity for every replay: more head room
            cap = max(cap, _capacity_for(int(R_c * (P / max(P_c, 1)) * 1.25) + 1))
    else:
        cap = spec.hint.get((P, W, H))
    if cap is None and (model is not None or no_read):
        cap = _capacity_for(4 * P)                           # (too small: the retry below repeats the call with room for the count;
    #                                                           without a host read the camera's next render has room)
    if cap is None or P == 0 or not SPECULATIVE_LAUNCH:
        if capturing:
            raise _lib.ScgError("a rasterizer forward inside a stream capture needs the one-call path (P > 0, SPECULATIVE_LAUNCH)")
        return None
    lib = _lib.load()
    plan = spec.plan(lib, P, W, H, cap)
    if not plan.accepts:
        if capturing:
            raise _lib.ScgError("a rasterizer forward inside a stream capture needs the tile-first binning "
                                "(scg_binning_accepts_bound): this image / capacity takes the staged path, which reads the host")
        return None
    pinned = None if no_read else spec.take(plan.partial_bytes)   # this forward's pinned words (returned once num_rendered is read)
    try:
        if model is None:
            means3D = _f32c(means3D, dev)
            opacities = _f32c(opacities, dev)
            shs = _f32c(shs, dev)
            colors_precomp = _f32c(colors_precomp, dev)
            scales = _f32c(scales, dev)
            rotations = _f32c(rotations, dev)
            cov3D_precomp = _f32c(cov3D_precomp, dev)
            M = shs.shape[1] if shs is not None else 0
        else:
            M = 16
        fr = _frame_for(settings, P, M, dev, forward=True)
        _SPEC_STATS["one_call_forwards"] += 1
        if fr.c.tile_cost_in:
            _SPEC_STATS["tile_cost_hints"] += 1
        if fr.c.bwd_cost_in:
            _SPEC_STATS["bwd_order_hints"] += 1
        timer = timer or _ACTIVE_TIMER
        with _on_device(dev):
            stage_ev = timer.stage_events("forward") if isinstance(timer, StageTimer) else None
            stream = _stream(dev)
            # num_rendered without an event (ABI 9): the pinned words are armed by scg_forward and watched by
            # scg_wait_num_rendered — no barrier packet behind the geometry kernel, no event wake-up
            ev = None if EVENTLESS_WAIT else pinned.event_handle()
            img = torch.empty((5, H, W), dtype=torch.float32, device=dev)          # colour | depth | alpha
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            # the gradient records of the coming backward (cleared by the forward blend) live behind the workspace in the same
            # allocation: one torch.empty less per step (they are only ever addressed by raw pointer)
            ds_bytes = (P * DSPLAT_FLOATS * 4 + 64) if prepare_backward else 0
            ip = img.data_ptr()
            hw4 = H * W * 4
            if model is None:
                inputs = (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp)
                in_ptrs = tuple(None if t is None else t.data_ptr() for t in inputs)
            else:
                inputs = model.tensors
            cw = None
            while True:
                # (without a host read the geometry kernel's partial sums of num_rendered go to device memory behind the records:
                # nobody reads them — the count comes from the binning stage, ScgFrame.num_rendered_out)
                ws = torch.empty((plan.total + ds_bytes + (plan.partial_bytes + 256 if no_read else 0),), dtype=torch.uint8,
                                 device=dev)
                wp = ws.data_ptr()
                dsplats = ((wp + plan.total + 63) & ~63) if prepare_backward else None       # 64-byte aligned records
                options = (0 if FUSED_SORT else 1) | (0 if FUSED_HIST else 2) | (0 if prepare_backward else 4) | \
                    _rare_options(fr.long_np) | (64 if (ev is None and not no_read) else 0)
                if fr.long_np is not None and fr.long_np[0] >= 0 and not plan.sorts_in_blend(lib, options):
                    # nobody writes the words in this frame: what an earlier, sparser frame of the camera left there is stale
                    fr.long_np[:] = -1
                    options &= ~(8 | 16 | 32)
                if no_read:
                    cw = _count_word(cap, P, ckey, dev.index)
                    fr.c.num_rendered_out = cw.ptr
                    sums, ev = (wp + plan.total + ds_bytes + 255) & ~255, None
                else:
                    fr.c.num_rendered_out = None
                    sums = pinned.ptr
                try:
                    if model is None:
                        check(lib.scg_forward(fr.ref, *in_ptrs, cap, wp, plan.total, radii.data_ptr(), ip,
                                              ip + 3 * hw4, ip + 4 * hw4, sums, ev,
                                              dsplats, options, stage_ev,
                                              stream), "scg_forward")
                    else:
                        check(lib.scg_forward_model(fr.ref, model.ref, cap, wp, plan.total, radii.data_ptr(), ip,
                                                    ip + 3 * hw4, ip + 4 * hw4, sums, ev, dsplats, options, stage_ev,
                                                    stream), "scg_forward_model")
                finally:
                    fr.c.num_rendered_out = None
                if no_read:
                    # launched, not waited for: the count is looked at when this camera is rendered next (or by settle_counts /
                    # the captured step that owns the word)
                    _OVERFLOW["renders"] += 1
                    if capturing:
                        cw.captured = True
                        (_CAPTURE_RECORD if _CAPTURE_RECORD is not None else _ANON_CAPTURED).append(cw)
                    else:
                        spec.pending.setdefault(ckey, []).append(cw)
                    R = None
                    break
                R = lib.scg_wait_num_rendered(ev, pinned.ptr, P)
                if R < 0:
                    check(int(R), "scg_wait_num_rendered")
                if R <= cap:
                    break
                # the bound was too small (rare: the scene grew by > 12 % since this camera's last render): lists were
                # clipped, run again with room for the real count
                _SPEC_STATS["overflow_retries"] += 1
                cap = _capacity_for(R)
                plan = spec.plan(lib, P, W, H, cap)
                if not plan.accepts:                 # the larger bound no longer fits the tile-first binning:
                    spec.cam_hint.pop((W, H, cam), None)         # the staged path (global sort) takes over
                    spec.hint.pop((P, W, H), None)
                    spec.give_back(pinned)
                    return None
                stage_ev = None
            if R is not None:
                nxt = _next_capacity(cap, R)
                spec.hint[(P, W, H)] = nxt
                spec.cam_hint.pop((W, H, cam), None)             # (re-inserted: the dict's order is the eviction order)
                spec.cam_hint[(W, H, cam)] = (nxt, R, P)
            for table in (spec.hint, spec.cam_hint):             # bounded: the OLDEST entries go, never the ones just written
                if len(table) > 1024:
                    for k in list(table)[:128]:
                        del table[k]
        # "num_rendered": None for a forward that did not read the host ("count_word": where its count arrives)
        state = {"ws": ws, "cap": cap, "plan": plan, "frame": fr, "dsplats_zeroed": dsplats, "num_rendered": R,
                 "inputs": inputs, "has_backward_state": bool(prepare_backward), "count_word": cw}
        if pinned is not None:
            spec.give_back(pinned)               # (num_rendered has been read: no kernel writes these words any more)
        return img[0:3], radii, img[3:4], img[4:5], state
    except BaseException:
        # kernels of the failed call may still be writing ws / img / radii and the pinned words: let them finish before Python
        # frees the buffers, and keep the pinned block out of circulation for good (dropping the reference hands it
        # back to torch's caching host allocator)
        try:
            if not capturing:
                torch.cuda.current_stream(dev).synchronize()
        except Exception:                        # noqa: BLE001 - the original error matters more
            pass
        if pinned is not None:
            _QUARANTINE.append(pinned)
        raise


def backward_fused(inputs, radii, state, dL_dcolor, dL_ddepth, dL_dalpha, timer: Optional[Callable] = None, into=None,
                   d_means2D_out=None):
    """Stages 4-5 in one library call; every parameter gradient is a view of ONE flat fp32 allocation (see
    _grad_outputs).  Returns the same dict as backward_stages; `into` as there."""
    lib = _lib.load()
    if not state.get("has_backward_state", True):
        raise _lib.ScgError("this forward ran without backward state (prepare_backward=False -> scg_forward's "
                            "SCG_FORWARD_NO_BACKWARD_STATE: final_T / n_contrib were not written): it cannot be differentiated")
    means3D = inputs[0]
    dev = means3D.device
    fr = state["frame"]
    if fr.hints is not None and fr.hints.bcost is not None:
        fr.hints.bwritten = True                             # (this backward records its quadrants' times: the next forward's hint)
    H, W = fr.H, fr.W
    dL_dcolor = _f32c(dL_dcolor, dev)
    if dL_dcolor is None:
        dL_dcolor = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
    dL_ddepth = _f32c(dL_ddepth, dev)
    dL_dalpha = _f32c(dL_dalpha, dev)
    timer = timer or _ACTIVE_TIMER
    with _on_device(dev):
        stage_ev = timer.stage_events("backward") if isinstance(timer, StageTimer) else None
        stream = _stream(dev)
        dsplats = state.get("dsplats_zeroed")               # raw pointer into the forward's allocation (state["ws"])
        state["dsplats_zeroed"] = None                      # usable once
        prezeroed = dsplats is not None
        keep = None
        if dsplats is None:                                 # a second backward over one forward: records of its own
            keep = torch.empty((means3D.shape[0], DSPLAT_FLOATS), dtype=torch.float32, device=dev)
            dsplats = keep.data_ptr()
        out = _grad_outputs(inputs, into, d_means2D_out, dev)
        flags = 1 if into is not None else _sh_tail_promise(out.get("_pooled"), (fr.c.sh_degree + 1) ** 2)
        check(lib.scg_backward(fr.ref, *(None if t is None else t.data_ptr() for t in inputs), radii.data_ptr(),
                               state["cap"], state["ws"].data_ptr(), dL_dcolor.data_ptr(), ptr(dL_ddepth), ptr(dL_dalpha),
                               dsplats, int(prezeroed), out["means3D"].data_ptr(), out["means2D"].data_ptr(),
                               out["opacities"].data_ptr(), ptr(out["shs"]), ptr(out["colors_precomp"]), ptr(out["scales"]),
                               ptr(out["rotations"]), ptr(out["cov3D_precomp"]), flags, stage_ev, stream),
              "scg_backward")
    return out


def grad_arena(params):
    """The flat fp32 tensor that holds every `p.grad` of the latest rasterizer backward, when they all live in ONE
    storage (backward_stages writes every parameter gradient into one allocation and autograd keeps those views as
    `.grad` when it was None) and TILE a range of it; else None.  The range starts at the first of the given gradients
    and ends behind the last: a subset of the parameters (only the opacities, say) yields only its own span, and a subset
    with another parameter's gradient in between yields None — an all-reduce of the result never touches a gradient
    that was not asked for.  Nothing is registered anywhere: the arena is rebuilt from the gradients' shared storage, so
    it lives exactly as long as a gradient does."""
    if not params or params[0].grad is None:
        return None
    st = params[0].grad.untyped_storage()
    base = st.data_ptr()
    begin, end, covered = None, 0, 0
    for p in params:
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base:
            return None
        off = g.storage_offset()
        begin = off if begin is None else min(begin, off)
        end = max(end, off + g.numel())
        covered += g.numel()
    # segments are padded to 16 bytes (<= 3 floats each): anything more between them is somebody else's memory
    if end * 4 > st.nbytes() or (end - begin) - covered > 3 * len(params):
        return None
    return torch.empty((0,), dtype=torch.float32, device=params[0].grad.device).set_(st, begin, (end - begin,))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        needs_grad = any(ctx.needs_input_grad)
        _require_cuda(means3D)
        fused = forward_fused(raster_settings, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp,
                              needs_grad)
        if fused is not None:
            color, radii, depth, alpha, state = fused
            inputs = state.pop("inputs")
            ctx.fused_state = state
        else:
            st = forward_stages(raster_settings, means3D, opacities, sh, colors_precomp, scales, rotations,
                                cov3Ds_precomp, prepare_backward=needs_grad)
            color, radii, depth, alpha, inputs = st["color"], st["radii"], st["depth"], st["alpha"], st["inputs"]
            ctx.fused_state = None
            # raw pointers into the two arenas (kept alive by the reference to `arenas`)
            ctx.saved_state = {"ptrs": st["ptrs"], "arenas": st["arenas"],
                               "dsplats_zeroed": st["dsplats_zeroed"], "frame": st["frame"]}
        ctx.raster_settings = raster_settings
        ctx.inputs_present = tuple(t is not None for t in inputs)
        ctx.shapes = (means3D.shape, means2D.shape, None if sh is None else sh.shape, opacities.shape)
        # inputs that are not fp32 (converted for the kernels): their gradients are cast back in backward
        odd = tuple(None if (t is None or t.dtype == torch.float32) else t.dtype
                    for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
        ctx.odd_dtypes = odd if any(d is not None for d in odd) else None
        # the inputs (geometry_backward recomputes the forward chain from them) and radii go through autograd's saved
        # tensors: an in-place update between forward and backward trips the version counter instead of silently
        # pairing new parameter values with the splats / radii of the old ones
        if needs_grad:
            ctx.save_for_backward(*[t for t in inputs if t is not None], radii)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)       # missing output gradients arrive as None, not as zero-filled tensors
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        saved = ctx.saved_tensors                            # raises if an input was modified in place since forward
        it = iter(saved)
        inputs = tuple(next(it) if present else None for present in ctx.inputs_present)
        radii = saved[-1]
        try:
            if ctx.fused_state is not None:
                g = backward_fused(inputs, radii, ctx.fused_state, grad_color, grad_depth, grad_alpha)
            else:
                state = dict(ctx.saved_state, radii=radii)
                g = backward_stages(ctx.raster_settings, inputs, state, grad_color, grad_depth, grad_alpha)
                ctx.saved_state["dsplats_zeroed"] = None     # usable once (a second backward memsets its own)
        except Exception:
            if ctx.raster_settings.debug:
                names = ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")
                _debug_dump(DEBUG_SNAPSHOT_BW, "backward", ctx.raster_settings,
                            dict(zip(names, inputs), radii=radii, grad_color=grad_color, grad_depth=grad_depth,
                                 grad_alpha=grad_alpha))
            raise
        means_shape, means2d_shape, sh_shape, opac_shape = ctx.shapes

        def _shape(t, shape):
            return t if (t is None or t.shape == shape) else t.reshape(shape)
        # order = forward argument order
        grads = (_shape(g["means3D"], means_shape), _shape(g["means2D"], means2d_shape), _shape(g["shs"], sh_shape),
                 g["colors_precomp"], _shape(g["opacities"], opac_shape), g["scales"], g["rotations"],
                 g["cov3D_precomp"])
        if ctx.odd_dtypes is not None:
            grads = tuple(t if (t is None or d is None) else t.to(d) for t, d in zip(grads, ctx.odd_dtypes))
        return grads + (None,)


def _none_if_empty(t):
    return None if (t is None or (isinstance(t, torch.Tensor) and t.numel() == 0)) else t


DEBUG_SNAPSHOT_FW = "snapshot_fw.dump"     # file names of upstream's debug dumps [UPSTREAM-RECALL]
DEBUG_SNAPSHOT_BW = "snapshot_bw.dump"


def _debug_dump(path, what, raster_settings, tensors):
    """`debug=True` (arguments/__init__.py:68 --debug -> gaussian_renderer/__init__.py:50): when a library call fails,
    the arguments of the failing call are saved with torch.save before the error is re-raised — what upstream's extension
    does with its snapshot_fw.dump / snapshot_bw.dump [UPSTREAM-RECALL] — so the failure can be replayed off line."""
    try:
        cpu = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tensors.items()}
        cpu["raster_settings"] = tuple(v.detach().cpu() if isinstance(v, torch.Tensor) else v for v in raster_settings)
        torch.save(cpu, path)
        print(f"\nAn error occured in {what}. Writing {path} for debugging.")
    except Exception as e:                                   # noqa: BLE001 - the original error matters more
        print(f"\nAn error occured in {what}; the debug snapshot could not be written ({e}).")


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    args = (means3D, means2D, _none_if_empty(sh), _none_if_empty(colors_precomp), opacities, _none_if_empty(scales),
            _none_if_empty(rotations), _none_if_empty(cov3Ds_precomp), raster_settings)
    if not raster_settings.debug:
        return _RasterizeGaussians.apply(*args)
    try:
        return _RasterizeGaussians.apply(*args)
    except Exception:
        _debug_dump(DEBUG_SNAPSHOT_FW, "forward", raster_settings,
                    dict(means3D=means3D, means2D=means2D, sh=args[2], colors_precomp=args[3], opacities=opacities,
                         scales=args[5], rotations=args[6], cov3Ds_precomp=args[7]))
        raise


class GaussianRasterizer(nn.Module):
    """Drop-in for the module constructed at reference gaussian_renderer/__init__.py:53."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Upstream's frustum-culling helper ([UPSTREAM-RECALL]; not called by the reference): True for points whose
        view-space z exceeds the 0.2 near cut the geometry stage applies (the same single test — DESIGN decision D3)."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.to(positions.device, torch.float32)      # row-vector convention
            z = positions[:, 0] * vm[0, 2] + positions[:, 1] * vm[1, 2] + positions[:, 2] * vm[2, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        shs, colors_precomp = _none_if_empty(shs), _none_if_empty(colors_precomp)
        scales, rotations, cov3D_precomp = _none_if_empty(scales), _none_if_empty(rotations), _none_if_empty(cov3D_precomp)
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)


# ---------------------------------------------------------------------------------------------------------------------
# K views of the same Gaussians in ONE autograd node (BASELINE cfg5: "multi-view batched step")
# ---------------------------------------------------------------------------------------------------------------------
class _RasterizeViews(torch.autograd.Function):
    """forward: the K views one after the other (each with its own saved state); backward: per view blend backward +
    geometry backward, the second and later views ADDING their parameter gradients to the first one's in the kernel
    (scg_backward `accumulate`), so the node returns the gradient of the SUM over views from one flat arena — no add pass
    per view over 236 bytes per Gaussian, one chain-rule pass through whatever produced the inputs, one gradient
    exchange per K views (parallel.GradBucket.reduce_grads all-reduces that arena in place).
    Outputs, flat: color_0, radii_0, depth_0, alpha_0, color_1, ...; means2D is (K, P, 3): one screen-space gradient slot
    per view (the densification statistics are per view, scene/gaussian_model.py:932-934)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings_list):
        needs_grad = any(ctx.needs_input_grad)
        _require_cuda(means3D)
        outs, states, inputs = [], [], None
        for st_ in settings_list:
            fused = forward_fused(st_, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp, needs_grad)
            if fused is not None:
                color, radii, depth, alpha, state = fused
                inputs = state.pop("inputs")
                states.append(("fused", state))
            else:
                fs = forward_stages(st_, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp,
                                    prepare_backward=needs_grad)
                color, radii, depth, alpha, inputs = fs["color"], fs["radii"], fs["depth"], fs["alpha"], fs["inputs"]
                states.append(("staged", {"ptrs": fs["ptrs"], "arenas": fs["arenas"],
                                          "dsplats_zeroed": fs["dsplats_zeroed"], "frame": fs["frame"]}))
            outs += [color, radii, depth, alpha]
        ctx.settings_list = tuple(settings_list)
        ctx.states = states
        ctx.inputs_present = tuple(t is not None for t in inputs)
        ctx.shapes = (means3D.shape, means2D.shape, None if sh is None else sh.shape, opacities.shape)
        if needs_grad:
            ctx.save_for_backward(*[t for t in inputs if t is not None], *outs[1::4])
        ctx.mark_non_differentiable(*outs[1::4])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        K = len(ctx.settings_list)
        saved = ctx.saved_tensors
        it = iter(saved)
        inputs = tuple(next(it) if present else None for present in ctx.inputs_present)
        radii_all = saved[len(saved) - K:]
        means_shape, means2d_shape, sh_shape, opac_shape = ctx.shapes
        P = inputs[0].shape[0]
        # scg_backward writes a view's (P, 3) slot completely (zeros for culled Gaussians): no fill launch over K x P x 12 bytes;
        # only the slot of a view whose outputs did not reach the loss is cleared here
        d_means2D = torch.empty((K, P, 3), dtype=torch.float32, device=inputs[0].device)
        acc = None
        for k in range(K):
            g_color, _, g_depth, g_alpha = grads[4 * k: 4 * k + 4]
            if g_color is None and g_depth is None and g_alpha is None:
                d_means2D[k].zero_()
                continue                                             # this view's outputs did not reach the loss
            kind, state = ctx.states[k]
            if kind == "fused":
                acc = backward_fused(inputs, radii_all[k], state, g_color, g_depth, g_alpha, into=acc,
                                     d_means2D_out=d_means2D[k])
            else:
                acc = backward_stages(ctx.settings_list[k], inputs, dict(state, radii=radii_all[k]), g_color, g_depth,
                                      g_alpha, into=acc, d_means2D_out=d_means2D[k])
                state["dsplats_zeroed"] = None
        if acc is None:
            return (None,) * 9

        def _shape(t, shape):
            return None if t is None else t.reshape(shape)
        return (_shape(acc["means3D"], means_shape), d_means2D.reshape(means2d_shape), _shape(acc["shs"], sh_shape),
                acc["colors_precomp"], _shape(acc["opacities"], opac_shape), acc["scales"], acc["rotations"],
                acc["cov3D_precomp"], None)


class GaussianRasterizerViews(nn.Module):
    """K views of one set of Gaussians per call — the batched multi-view step of BASELINE cfg5; no counterpart in the
    reference, whose train.py:143 renders one view per iteration.  Same keyword arguments as GaussianRasterizer, except
    that `means2D` is (K, P, 3) (a screen-space gradient slot per view); returns a list of K
    (color, radii, depth, alpha) tuples.  The loss may use any subset of the views."""

    def __init__(self, raster_settings_list):
        super().__init__()
        self.raster_settings_list = list(raster_settings_list)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        shs, colors_precomp = _none_if_empty(shs), _none_if_empty(colors_precomp)
        scales, rotations, cov3D_precomp = _none_if_empty(scales), _none_if_empty(rotations), _none_if_empty(cov3D_precomp)
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        K = len(self.raster_settings_list)
        if means2D.dim() != 3 or means2D.shape[0] != K:
            raise ValueError(f"means2D must be ({K}, P, 3): one screen-space gradient slot per view")
        flat = _RasterizeViews.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                     self.raster_settings_list)
        return [tuple(flat[4 * k: 4 * k + 4]) for k in range(K)]
