"""Data parallelism over training VIEWS (no counterpart in the reference, which is a single
process pinned to cuda:0 — utils/general_utils.py:139).

One process per GPU, replicated Gaussians, rank r renders view (step*world + r) mod n_views.  The only
exchange step of the path is the sum of the Gaussian parameter gradients: they are packed into ONE flat
fp32 bucket (59 floats = 236 B per Gaussian with SH degree 3 storage: means 3 + sh 48 + opacity 1 +
scales 3 + rotations 4) and reduced with a single all-reduce — on MI355X that is RCCL over xGMI
(backend "nccl"); on the CPU tests it is gloo.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so
one large bucket per step amortises the per-collective latency; there is nothing to overlap it with (all
gradients become final in the last kernel of the backward).

Densification statistics (scene/gaussian_model.py:932-934, train.py:192) must also agree across ranks so
that every replica takes identical densify/prune decisions: xyz_gradient_accum (sum), denom (sum),
max_radii2D (max).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


# Exchange steps normally collapse to nothing in a world of one.  `init_from_env(..., force=True)` turns them on anyway:
# a single process then walks the exact code path of an N-GPU job — process-group creation on the RCCL backend with a
# bound device, ReduceOp.AVG on the gradient arena in place, the densification-state reductions, the RNG broadcast —
# so the first contact with a real multi-GPU node cannot fail on any of them (tests/test_gpu_train_loop.py,
# `bench.py --gpus 1 --dist-backend nccl`).
_EXCHANGE_AT_WORLD_ONE = False


def _exchanging(group=None) -> bool:
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or _EXCHANGE_AT_WORLD_ONE)


def exchanging(group=None) -> bool:
    """True when the exchange steps of this module do anything (a world of more than one rank, or a forced world of one)."""
    return _exchanging(group)


# SURVEY §8e link budget of one MI355X in an 8-GPU node: 7 xGMI links x ~153 GB/s, point to point
XGMI_LINK_GBS = 153.0
XGMI_LINKS = 7


def exchange_time_model(nbytes: int, world: int):
    """Lower bounds (seconds) of one all-reduce of `nbytes` per rank over xGMI: a ring moves 2 (N-1)/N of the bucket over
    ONE link per GPU; a direct reduce-scatter / all-gather spreads the same volume over all N-1 links a GPU has to its
    peers (what RCCL can reach at best on a fully connected node)."""
    if world <= 1:
        return {"ring_s": 0.0, "all_links_s": 0.0}
    vol = 2.0 * (world - 1) / world * nbytes
    return {"ring_s": vol / (XGMI_LINK_GBS * 1e9), "all_links_s": vol / (XGMI_LINK_GBS * 1e9 * min(world - 1, XGMI_LINKS))}


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, world, local_rank).  Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* when
    WORLD_SIZE > 1 (or when `force`: a world of one with every exchange step executed); backend defaults to nccl
    (= RCCL) on GPU, gloo on CPU."""
    global _EXCHANGE_AT_WORLD_ONE
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if force:
        _EXCHANGE_AT_WORLD_ONE = True
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket                      # a world of one has nobody to agree with: any free port (a fixed one may
            with socket.socket() as sk:        # still be in TIME_WAIT from the previous run)
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """Index of the training view rank `rank` renders at `step`: consecutive views across ranks, wrapping."""
    return (step * world + rank) % n_views


class GradBucket:
    """Flat fp32 bucket over a fixed list of parameter tensors: pack grads -> one all-reduce -> unpack.

    `active_dim1` optionally limits a parameter to its first k entries along dim 1: the SH tensor (P,16,3) only has
    non-zero gradients in its first (active_sh_degree+1)^2 coefficients (the rasterizer writes zeros above), and the
    reference trains at degrees 0..2 for most of its schedule (train.py:129, arguments/__init__.py:73) — at degree 0
    the bucket shrinks from 236 to 56 bytes per Gaussian."""

    def __init__(self, params: Sequence[torch.Tensor], active_dim1: Optional[Dict[int, int]] = None):
        # a limit that covers the whole dimension is no limit (keeps the zero-copy path of reduce_grads available)
        self.active = {i: int(k) for i, k in (active_dim1 or {}).items() if int(k) < params[i].shape[1]}
        self.full_shapes = [tuple(p.shape) for p in params]
        self.shapes = []
        for i, p in enumerate(params):
            shp = list(p.shape)
            if i in self.active:
                shp[1] = min(int(self.active[i]), shp[1])
            self.shapes.append(tuple(shp))
        self.sizes = [int(torch.Size(s).numel()) for s in self.shapes]
        self.total = sum(self.sizes)
        dev = params[0].device if params else torch.device("cpu")
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.views: List[torch.Tensor] = []
        off = 0
        for n, shp in zip(self.sizes, self.shapes):
            self.views.append(self.flat[off:off + n].view(shp))
            off += n

    @property
    def nbytes(self) -> int:
        return self.total * 4

    def _active(self, i: int, t: torch.Tensor) -> torch.Tensor:
        return t[:, : self.shapes[i][1]] if i in self.active else t

    def pack(self, grads: Iterable[Optional[torch.Tensor]]):
        dst, src = [], []
        for i, (v, g) in enumerate(zip(self.views, grads)):
            if g is None:
                v.zero_()
            else:
                dst.append(v)
                src.append(self._active(i, g.reshape(self.full_shapes[i])))
        if dst:
            torch._foreach_copy_(dst, src)          # one multi-tensor launch instead of one copy kernel per parameter

    def all_reduce_mean(self, group=None):
        if _exchanging(group):
            self.last_bytes = self.total * 4
            if dist.get_backend(group) == "nccl":   # RCCL averages in the collective: no separate scaling kernel
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.div_(dist.get_world_size(group))
        return self.views

    def _all_reduce_mean(self, tensors: Sequence[torch.Tensor], group=None):
        """In place, every (non-empty) tensor of the list; on RCCL the collectives of one call are issued as ONE group (a single
        launch) where this torch offers the coalescing context with the signature used here — else one collective per tensor."""
        tensors = [t for t in tensors if t.numel() > 0]
        self.last_bytes = sum(int(t.numel()) * t.element_size() for t in tensors)
        if not tensors:
            return
        if dist.get_backend(group) == "nccl":
            done = False
            if len(tensors) > 1 and hasattr(dist, "_coalescing_manager"):
                try:
                    with dist._coalescing_manager(group, device=tensors[0].device, async_ops=False):
                        for t in tensors:
                            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                    done = True
                except TypeError:                   # an older torch: _coalescing_manager(group, reqs) — nothing was issued yet
                    done = False
            if not done:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        else:
            n = dist.get_world_size(group)
            for t in tensors:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                t.div_(n)

    def _limited_flat(self):
        """Flat buffer + views for the limited parameters only (the active SH coefficients): what the mixed path packs."""
        if getattr(self, "_lim", None) is None:
            idx = sorted(self.active)
            sizes = [self.sizes[i] for i in idx]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.flat.device)
            views, off = [], 0
            for i, n in zip(idx, sizes):
                views.append(flat[off:off + n].view(self.shapes[i]))
                off += n
            self._lim = (idx, flat, views)
        return self._lim

    def exchanged_bytes(self, path: Optional[str] = None) -> int:
        """Bytes per rank one reduce_grads hands to the collective(s) on `path` ("arena", "mixed", "packed"; default: the path
        the latest reduce_grads took, else "packed").  "packed" and "mixed" exchange the ACTIVE elements (`total`); "arena"
        all-reduces the gradient arena where it lies, i.e. the full tensors incl. the 16-byte padding of its segments — the
        figure of the latest call is in `last_bytes`."""
        path = path or getattr(self, "last_path", None) or "packed"
        if path == "arena":
            return sum(((int(torch.Size(s).numel()) + 3) // 4 * 4) * 4 for s in self.full_shapes)
        return self.total * 4

    def reduce_grads(self, params: Sequence[torch.Tensor], group=None):
        """In-place: p.grad <- mean over ranks of p.grad, for every parameter.

        Three paths, `last_path` says which ran:
          "arena"   no limit: the rasterizer's backward wrote all gradients into one flat arena that autograd kept as the
                    .grad views — all-reduced where it lies (no pack / unpack passes over 236 B per Gaussian);
          "mixed"   SH degree < 3 (most of the reference's schedule, train.py:129): the contiguous span of the unlimited
                    gradients (means, opacity, scales, rotations: 44 B per Gaussian) is all-reduced in place, only the
                    ACTIVE SH coefficients are packed (12-108 B per Gaussian, one strided copy each way) — 56-152 B
                    exchanged instead of 236, and no full pack / unpack;
          "packed"  gradients that do not live in one arena (they came from elsewhere): pack -> all-reduce -> unpack."""
        self.last_path = None
        if _exchanging(group):
            from .rasterizer import grad_arena
            if not self.active:
                arena = grad_arena(list(params))
                if arena is not None:
                    self._all_reduce_mean([arena], group)
                    self.last_path = "arena"
                    return
            else:
                dense = [p for i, p in enumerate(params) if i not in self.active]
                arena = grad_arena(dense) if dense else None
                idx, lim_flat, lim_views = self._limited_flat()
                if (arena is not None or not dense) and all(params[i].grad is not None for i in idx):
                    src = [self._active(i, params[i].grad.reshape(self.full_shapes[i])) for i in idx]
                    torch._foreach_copy_(lim_views, src)
                    self._all_reduce_mean(([arena] if arena is not None else []) + [lim_flat], group)
                    torch._foreach_copy_(src, lim_views)
                    self.last_path = "mixed"
                    return
        self.pack([p.grad for p in params])
        self.all_reduce_mean(group)
        dst = []
        for i, p in enumerate(params):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            dst.append(self._active(i, p.grad))
        torch._foreach_copy_(dst, list(self.views))
        self.last_path = "packed"


def reduce_densification_stats(xyz_gradient_accum: torch.Tensor, denom: torch.Tensor, max_radii2D: torch.Tensor,
                               group=None):
    """Make the densification state identical on all ranks: sums for the accumulators, max for the radii."""
    if _exchanging(group):
        both = torch.cat([xyz_gradient_accum.reshape(-1), denom.reshape(-1)]).float()
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        n = xyz_gradient_accum.numel()
        xyz_gradient_accum.copy_(both[:n].view_as(xyz_gradient_accum))
        denom.copy_(both[n:].view_as(denom))
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)
    return xyz_gradient_accum, denom, max_radii2D


_SYNC_RNG_CALLS = 0


def sync_rng(seed: Optional[int] = None, group=None, device=None) -> int:
    """Replica-consistent random numbers: every rank seeds the default generator OF THE DEVICE THE
    DENSIFICATION SAMPLES ARE DRAWN ON with the same value — rank 0's `seed` (drawn there when None), broadcast.  Called
    before a densification step it makes `torch.normal(mean, std)` in densify_and_split (scene/gaussian_model.py:875: CUDA
    tensors, default generator, no `generator=` argument to hand a private one to) produce identical samples on every
    replica (same device type, same shapes, same Philox stream), so the replicas' Gaussian sets stay bit-identical without
    exchanging the samples.  ONLY that generator is touched: the CPU generator (view shuffling, random backgrounds — the
    reference seeds it once, utils/general_utils.py:136-138) and the other devices' generators keep their streams.
    `device`: default = the current CUDA device when there is one, else the CPU.  Returns the seed in use."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    exchanging_ = _exchanging(group)
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()] \
        if device.type == "cuda" else torch.default_generator
    if seed is None and (not exchanging_ or dist.get_rank(group) == 0):
        # derived from the generator's current seed and a call counter: reproducible run to run, advances no stream
        global _SYNC_RNG_CALLS
        _SYNC_RNG_CALLS += 1
        seed = ((gen.initial_seed() * 6364136223846793005 + _SYNC_RNG_CALLS * 1442695040888963407) >> 20) & 0x7FFFFFFF
    if exchanging_:
        dev = _collective_device(group)
        t = torch.tensor([int(seed) if seed is not None else 0], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0, group=group)
        seed = int(t.item())
    gen.manual_seed(int(seed))
    return int(seed)


def broadcast_from_rank0(*tensors: torch.Tensor, group=None):
    """In place: every tensor takes rank 0's content.  The belt-and-braces alternative to sync_rng for values that
    must agree bit for bit across replicas (freshly sampled positions, a pruning mask) whatever produced them."""
    if _exchanging(group):
        for t in tensors:
            dist.broadcast(t, src=0, group=group)
    return tensors if len(tensors) != 1 else tensors[0]


def _collective_device(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def barrier():
    if _exchanging():
        dist.barrier()


def shutdown():
    """Destroy the process group init_from_env created (no-op without one)."""
    if dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(value: float, device) -> float:
    if _exchanging():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)
