"""The capacity policy and the counts of renders that were launched without a host read.  (Split out of rasterizer.py in round 6.)

A forward is launched with a CAPACITY — an upper bound of num_rendered, the number of (Gaussian, tile) instances — and learns the
real count either at once (the default: the host waits for the geometry stage's partial sums) or LATER: with rasterizer.no_host_read()
/ inside a stream capture the binning stage leaves its count in a pinned word (ScgFrame.num_rendered_out) that is looked at the next
time the camera is rendered.  This module owns those words and what a count that has arrived does to the camera's capacity;
rasterizer.py decides when a forward runs in which mode.
"""
from __future__ import annotations

import torch

from . import _lib

_SPEC_STATE = {}                         # device index -> rasterizer._SpecState (capacities per shape and per camera, pending counts)


def _capacity_for(R: int) -> int:
    """Upper bound of num_rendered to lay point_list out for, given the latest count: ~12-25 % head room, quantised to
    1/16 of its magnitude so that the value (and the cached workspace plan keyed by it) stays put from step to step."""
    need = int(R * 1.125) + 4096
    g = 1 << max(12, need.bit_length() - 4)
    return (need + g - 1) // g * g


def _next_capacity(cur, R: int) -> int:
    """Keep the capacity in use while the count sits comfortably inside it; otherwise re-derive it from the count."""
    if cur is not None and R + (R >> 5) <= cur <= 2 * R + 65536:
        return cur
    return _capacity_for(R)



_COUNT_ARMED = 0xFFFFFFFF                # "the binning stage of this render has not written its count yet"
_COUNT_POOL = None                       # one pinned allocation of count words per process
_COUNT_FREE = []
_COUNT_SLOTS = 4096
_OVERFLOW = {"renders": 0, "overflows": 0, "settled": 0}
_ANON_CAPTURED = []                      # count words of forwards captured by somebody else's graph (kept: the graph writes them)
_QUARANTINE = []                         # pinned blocks of forwards that failed after their launch (never handed out again)


class _CountWord:
    """One pinned word that a render's binning stage overwrites with num_rendered + what the binding needs to judge it later."""
    __slots__ = ("slot", "np", "ptr", "cap", "P", "key", "device_index", "captured")

    def value(self):
        v = int(self.np[0])
        return None if v == _COUNT_ARMED else v


def _count_pool():
    """The process's pinned count words (allocated at the first use — graph_step asks BEFORE it starts a capture: a pinned
    allocation inside a capture is not allowed)."""
    global _COUNT_POOL
    if _COUNT_POOL is None:
        t = torch.full((_COUNT_SLOTS,), -1, dtype=torch.int32).pin_memory()
        _COUNT_POOL = (t, t.numpy().view("uint32"), t.data_ptr())
        _COUNT_FREE.extend(range(_COUNT_SLOTS - 1, -1, -1))
    return _COUNT_POOL


def _count_word(cap, P, key, device_index) -> _CountWord:
    _count_pool()
    if not _COUNT_FREE:                                      # every slot is waiting for its render: let the device catch up
        torch.cuda.synchronize()
        settle_counts()
        if not _COUNT_FREE:
            raise _lib.ScgError("no_host_read: more than %d renders (or captured steps) hold a count word" % _COUNT_SLOTS)
    _t, arr, base = _COUNT_POOL
    w = _CountWord()
    w.slot = _COUNT_FREE.pop()
    w.np = arr[w.slot: w.slot + 1]
    w.np[0] = _COUNT_ARMED
    w.ptr = base + 4 * w.slot
    w.cap, w.P, w.key, w.device_index, w.captured = int(cap), int(P), key, device_index, False
    return w


def _settle_word(spec, w: _CountWord, R: int):
    """The count of a render that was launched without a host read has arrived: the camera's capacity follows it."""
    _OVERFLOW["settled"] += 1
    if R > w.cap:
        _OVERFLOW["overflows"] += 1
    W, H, cam = w.key
    ent = spec.cam_hint.get(w.key)
    cur = ent[0] if (ent is not None and ent[2] == w.P) else w.cap
    nxt = _next_capacity(max(cur, w.cap) if R <= w.cap else None, R)
    spec.hint[(w.P, W, H)] = nxt
    spec.cam_hint.pop(w.key, None)
    spec.cam_hint[w.key] = (nxt, R, w.P)


def _settle_camera(spec, key):
    """Look (without waiting) at the count words of this camera's earlier no-host-read renders, oldest first."""
    q = spec.pending.get(key)
    if not q:
        return
    while q:
        w = q[0]
        R = w.value()
        if R is None:
            break
        q.pop(0)
        _settle_word(spec, w, R)
        _COUNT_FREE.append(w.slot)
    if not q:
        spec.pending.pop(key, None)


def settle_counts(device=None) -> dict:
    """Look at every outstanding count word (after a synchronisation of the caller's all of them have arrived) and return
    overflow_stats().  Never waits."""
    for idx, spec in list(_SPEC_STATE.items()):
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        for key in list(spec.pending):
            _settle_camera(spec, key)
    for w in _ANON_CAPTURED:                                 # words a foreign graph's replays write: the latest count, once each
        R = w.value()
        if R is not None:
            w.np[0] = _COUNT_ARMED
            spec = _SPEC_STATE.get(w.device_index)
            if spec is not None:
                _settle_word(spec, w, R)
    return overflow_stats()


def overflow_stats() -> dict:
    """{"renders": forwards launched without a host read, "settled": of those, counts looked at so far, "overflows": of those,
    renders whose lists were clipped (their result was incomplete; the next render of the camera had room again)}."""
    return dict(_OVERFLOW)
