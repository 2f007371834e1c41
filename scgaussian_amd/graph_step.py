"""A training step captured once and replayed (hipGraph through torch.cuda.CUDAGraph) — VERDICT r5 item 3.

At the reference's own scene size (<= 12 k Gaussians, data_preprocess/get_match_info.py:376) a training step is 65 us of GPU
work behind 130-200 us of host work (Python, autograd bookkeeping, six launches, the spin on num_rendered).  The rasterizer's
forward can run without its host read (rasterizer.NO_HOST_READ: the capacity comes from the camera's earlier renders, the count
arrives in a pinned word that is looked at later), which makes forward + backward capturable:

    step = CapturedStep(lambda: loss_and_backward(view))      # runs the closure eagerly (warm-up), then captures it
    for it in range(n):
        out = step.replay()                                   # the closure's return value: static tensors, overwritten per replay
        optimizer.step()                                      # (.grad tensors are static too: allocated inside the capture)

What a replay cannot do, and how it is handled:
  * the capacity (upper bound of num_rendered) is frozen at capture time, 25 % above the camera's last count.  Every replay leaves
    its count in the pinned word of each captured forward; `replay()` looks at the words BEFORE launching: a count beyond the
    capacity means an EARLIER replay's lists were clipped (its images and gradients were incomplete) — `overflows` is incremented
    and the step is captured again with room for the count, so the replay that follows is complete: recovery one step late.
  * launch-order hints (tile costs, long-list counts) are frozen with the graph; they never enter a result.
  * the closure must be capturable in everything else it does: no `.item()`, no allocation of pinned memory, no new cameras
    (a camera's first render reads its view matrix once: the warm-up runs take care of that).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import rasterizer as R


class CapturedStep:
    """`fn()` — forward + backward (any number of rasterizer calls), returning a tensor / tuple / dict of tensors — captured in a
    hipGraph.  `warmup` eager runs come first (they establish every camera's capacity and hints; their side effects on `.grad`
    are cleared by setting the gradients of `params` to None before the capture, as torch's whole-network capture recipe does).

    replay() -> the captured return value (the same tensor objects every time, rewritten by the replay); `p.grad` of every tensor in
    `params` is the gradient THIS step's graph wrote (each captured step owns the gradient tensors of its capture: with one step per
    training view over the same parameters, the optimizer must see the replayed view's, not the last captured one's).
    Attributes: `overflows` (replays whose lists were clipped, detected one replay late), `recaptures`, `capacities`."""

    def __init__(self, fn: Callable[[], object], params=None, warmup: int = 3, stream: Optional[torch.cuda.Stream] = None):
        self.fn = fn
        self.params = list(params) if params is not None else None
        self.warmup = int(warmup)
        self.stream = stream
        self.overflows = 0
        self.recaptures = -1
        self.replays = 0
        self.grads = None
        self.graph = None
        self.outputs = None
        self.words = []
        self._capture()

    # ---- capture --------------------------------------------------------------------------------------------------------------
    def _clear_grads(self):
        if self.params is not None:
            for p in self.params:
                p.grad = None

    def _capture(self):
        torch.cuda.synchronize()
        side = self.stream or torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                     # eager, WITH the host read: exact capacities, camera keys, hints
                self._clear_grads()
                self.fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        R.settle_counts()
        R._count_pool()
        self._clear_grads()
        # the words of the graph that is being replaced stay reserved until the old graph is gone (it may still be running)
        old_words, old_graph = self.words, self.graph
        record = []
        graph = torch.cuda.CUDAGraph()
        prev, R._CAPTURE_RECORD = R._CAPTURE_RECORD, record
        try:
            with torch.cuda.graph(graph, stream=side):
                outputs = self.fn()
        finally:
            R._CAPTURE_RECORD = prev
        self.graph, self.outputs, self.words = graph, outputs, record
        # the gradients this graph writes: allocated inside ITS capture.  Several captured steps over the same parameters (one per
        # training view) each have their own; replay() points .grad at this step's before it returns
        self.grads = [p.grad for p in self.params] if self.params is not None else None
        self.capacities = [w.cap for w in record]
        self.recaptures += 1
        if old_graph is not None:
            torch.cuda.synchronize()
            del old_graph
            for w in old_words:
                R._COUNT_FREE.append(w.slot)

    # ---- replay ---------------------------------------------------------------------------------------------------------------
    def check(self) -> bool:
        """Look (without waiting) at the counts the latest completed replay left: True when one of them exceeded its capacity.
        Raises the cameras' capacities so that the next capture has room."""
        clipped = False
        for w in self.words:
            Rn = w.value()
            if Rn is None:
                continue
            w.np[0] = R._COUNT_ARMED
            spec = R._SPEC_STATE.get(w.device_index if w.device_index is not None else torch.cuda.current_device())
            if spec is not None:
                R._settle_word(spec, w, Rn)
            if Rn > w.cap:
                clipped = True
        return clipped

    def replay(self):
        if self.check():
            self.overflows += 1
            self._capture()
        self.graph.replay()
        self.replays += 1
        if self.grads is not None:
            for p, g in zip(self.params, self.grads):
                p.grad = g
        return self.outputs

    def __call__(self):
        return self.replay()

    def close(self):
        """Give the count words back (after the graph's last replay has finished)."""
        if self.graph is not None:
            torch.cuda.synchronize()
            self.graph = None
            for w in self.words:
                R._COUNT_FREE.append(w.slot)
            self.words = []

    def __del__(self):
        try:
            self.close()
        except Exception:                                    # noqa: BLE001 - interpreter shutdown
            pass
