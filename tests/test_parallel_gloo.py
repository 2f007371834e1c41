"""world_size-2 gloo tests of the view-parallel exchange step (CPU): gradient bucket all-reduce,
densification-state reduction, view assignment.  The rasterizer itself needs a GPU; here the per-rank
"gradients" are deterministic functions of (rank, view) so the reduced result is known in closed form."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scgaussian_amd import parallel as par


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_grads(P, view):
    g = torch.Generator().manual_seed(1000 + view)
    return [torch.randn(P, 3, generator=g), torch.randn(P, 16, 3, generator=g), torch.randn(P, 1, generator=g),
            torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)]


def _worker(rank, world, port, P, steps, ret, K=3):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, lr = par.init_from_env("gloo")
    assert (r, w) == (rank, world)
    params = [torch.zeros(P, 3), torch.zeros(P, 16, 3), torch.zeros(P, 1), torch.zeros(P, 3), torch.zeros(P, 4)]
    bucket = par.GradBucket(params)
    assert bucket.nbytes == P * 59 * 4          # 236 B per Gaussian
    n_views = 3
    for step in range(steps):
        v = par.view_for(step, rank, world, n_views)
        for p, g in zip(params, _fake_grads(P, v)):
            p.grad = g.clone()
        bucket.reduce_grads(params)
        views = [par.view_for(step, rr, world, n_views) for rr in range(world)]
        expect = [sum(gs) / world for gs in zip(*[_fake_grads(P, vv) for vv in views])]
        for p, e in zip(params, expect):
            assert torch.allclose(p.grad, e, atol=1e-6), (rank, step)
    # views-per-step K (BASELINE cfg5): every rank ACCUMULATES the gradients of its K views (the rasterizer's backward
    # does that in the kernel, tests/test_gpu_views.py), then ONE exchange: the result is the mean over all N*K views
    mine = [rank * K + k for k in range(K)]
    for i, p in enumerate(params):
        p.grad = sum(_fake_grads(P, 50 + v)[i] for v in mine) / K
    bucket.reduce_grads(params)
    for i, p in enumerate(params):
        e = sum(_fake_grads(P, 50 + v)[i] for v in range(world * K)) / (world * K)
        assert torch.allclose(p.grad, e, atol=1e-6), (rank, i)
    # SH-degree-limited steps (most of the reference's schedule, train.py:129) with the gradients where the rasterizer's
    # backward leaves them — views of ONE flat arena, the SH segment last (rasterizer._GRAD_ORDER): the span of the other
    # four is all-reduced in place, only the active SH coefficients are packed ("mixed"); the inactive ones stay untouched
    from scgaussian_amd.rasterizer import grad_arena
    order = (0, 2, 3, 4, 1)                                       # means, opacity, scales, rotations | shs
    for k_active in (1, 4, 9, 16):
        sizes = [params[i].numel() for i in order]
        arena = torch.zeros(sum(sizes))
        gs = _fake_grads(P, 200 + rank)
        for i, seg in zip(order, arena.split(sizes)):
            seg.copy_(gs[i].reshape(-1))
            params[i].grad = seg.view(params[i].shape)
        params[1].grad[:, k_active:] = float(rank + 1)            # marker: must not travel
        assert grad_arena(params) is not None and grad_arena([params[i] for i in (0, 2, 3, 4)]) is not None
        b = par.GradBucket(params, active_dim1={1: k_active})
        assert b.nbytes == P * (11 + 3 * k_active) * 4
        b.reduce_grads(params)
        assert b.last_path == ("mixed" if k_active < 16 else "arena"), b.last_path
        for i, p in enumerate(params):
            e = sum(_fake_grads(P, 200 + rr)[i] for rr in range(world)) / world
            if i == 1:
                assert torch.allclose(p.grad[:, :k_active], e[:, :k_active], atol=1e-6), (rank, k_active)
                if k_active < 16:
                    assert torch.all(p.grad[:, k_active:] == float(rank + 1))
            else:
                assert torch.allclose(p.grad, e, atol=1e-6), (rank, k_active, i)
            assert p.grad.untyped_storage().data_ptr() == arena.untyped_storage().data_ptr()     # still the arena's views
    for p in params:
        p.grad = None
    # The REFERENCE MODEL's parameters (scene/gaussian_model.py:491-509: zval, f_dc, f_rest, opacity, scaling, rotation + the six
    # bg_* tensors), gradients where the model path's backward leaves them (model_path._grad_arena: one arena, the two
    # features_rest segments last): "arena" at degree 3, and at a lower degree the other ten tensors' span in place + the packed
    # active coefficients of the two features_rest tensors ("mixed") — never pack / unpack over 236 B per Gaussian
    from scgaussian_amd import model_path as mp_
    from scgaussian_amd import synthetic as syn
    sc = syn.make_scene(P, 64, 48, seed=3)
    model = syn.make_raw_model(sc, ray_fraction=0.6)
    mparams = model.parameters()
    args = mp_._ModelArgs(mp_.tensors_of(model))
    f_rest_idx = [i for i, p in enumerate(mparams) if p.dim() == 3 and p.shape[1] == 15]
    assert len(f_rest_idx) == 2

    def fake(i, who):
        return torch.randn(mparams[i].shape, generator=torch.Generator().manual_seed(7000 + 31 * who + i))
    for deg in (3, 0, 1, 2):
        k_rest = (deg + 1) ** 2 - 1
        views = mp_._grad_arena(args, None)
        by_id = {id(t): n for n, t in zip(mp_.ARG_NAMES, args.tensors)}
        for i, p in enumerate(mparams):
            g = views[by_id[id(p)]]
            g.copy_(fake(i, rank))
            if i in f_rest_idx:
                g[:, k_rest:] = float(rank + 1)                       # marker above the active degree: must not travel
            p.grad = g
        assert grad_arena(mparams) is not None
        b = par.GradBucket(mparams, active_dim1={i: k_rest for i in f_rest_idx})
        b.reduce_grads(mparams)
        assert b.last_path == ("arena" if deg == 3 else "mixed"), (deg, b.last_path)
        if deg < 3:                                               # bytes handed to the collective: 44 + 12 (deg+1)^2 per Gaussian
            assert b.last_bytes <= P * (11 + 3 * (deg + 1) ** 2) * 4 + 16 * len(mparams), (deg, b.last_bytes)
        for i, p in enumerate(mparams):
            e = sum(fake(i, rr) for rr in range(world)) / world
            if i in f_rest_idx and deg < 3:
                assert torch.allclose(p.grad[:, :k_rest], e[:, :k_rest], atol=1e-6), (rank, deg, i)
                assert torch.all(p.grad[:, k_rest:] == float(rank + 1)), (rank, deg, i)
            else:
                assert torch.allclose(p.grad, e, atol=1e-6), (rank, deg, i)
        for p in mparams:
            p.grad = None
        del views
    # densification state: sums and max
    acc = torch.full((P, 1), float(rank + 1))
    den = torch.full((P, 1), 2.0 * (rank + 1))
    rad = torch.arange(P, dtype=torch.float32) * (1 if rank == 0 else -1) + rank
    par.reduce_densification_stats(acc, den, rad)
    assert torch.all(acc == sum(range(1, world + 1)))
    assert torch.all(den == 2.0 * sum(range(1, world + 1)))
    exp_rad = torch.maximum(torch.arange(P, dtype=torch.float32), -torch.arange(P, dtype=torch.float32) + (world - 1))
    assert torch.equal(rad, exp_rad)
    assert par.max_over_ranks(float(rank), torch.device("cpu")) == world - 1
    # replica-consistent densification RNG (scene/gaussian_model.py:875 torch.normal in densify_and_split):
    # the replicas start from DIFFERENT generator states (as they would after rendering different views) ...
    torch.manual_seed(100 + rank)
    stds = torch.rand(P, 3, generator=torch.Generator().manual_seed(3)) + 0.1      # replicated scaling
    diverged = torch.normal(mean=torch.zeros(2 * P, 3), std=stds.repeat(2, 1))
    # ... sync_rng puts them on one stream: the samples of the split are bit-identical everywhere
    seed = par.sync_rng()
    samples = torch.normal(mean=torch.zeros(2 * P, 3), std=stds.repeat(2, 1))
    gathered = [torch.empty_like(samples) for _ in range(world)]
    dist.all_gather(gathered, samples)
    assert all(torch.equal(gathered[0], g) for g in gathered), "densification samples differ between replicas"
    seeds = [None] * world
    dist.all_gather_object(seeds, seed)
    assert len(set(seeds)) == 1
    gathered_div = [torch.empty_like(diverged) for _ in range(world)]
    dist.all_gather(gathered_div, diverged)
    assert not torch.equal(gathered_div[0], gathered_div[1])      # (the test would be vacuous otherwise)
    # explicit seed, and the broadcast alternative for values that must agree whatever produced them
    assert par.sync_rng(4242) == 4242
    mask = (torch.rand(P, generator=torch.Generator().manual_seed(50 + rank)) > 0.5)
    new_xyz = torch.randn(P, 3, generator=torch.Generator().manual_seed(60 + rank))
    par.broadcast_from_rank0(mask, new_xyz)
    assert torch.equal(mask, torch.rand(P, generator=torch.Generator().manual_seed(50)) > 0.5)
    assert torch.equal(new_xyz, torch.randn(P, 3, generator=torch.Generator().manual_seed(60)))
    par.barrier()
    ret[rank] = 1
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_view_parallel_grad_bucket_world2():
    world, P, steps = 2, 257, 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, steps, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: 1, 1: 1}


@pytest.mark.timeout(300)
def test_view_parallel_exchange_world8_three_views_two_views_per_step():
    """BASELINE cfg5 at the node's size, on CPU: EIGHT ranks over gloo, three training views, K = 2 views accumulated per
    exchange — view_for's assignment, GradBucket's all-reduce (mean over the 8 / the 16 views), reduce_densification_stats
    (SUM, MAX over eight ranks), sync_rng (eight diverged generators onto one stream), broadcast_from_rank0, barrier,
    max_over_ranks: every exchange step the first 8-GPU launch executes, with eight real peers."""
    world, P, steps = 8, 97, 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, steps, ret, 2)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: 1 for r in range(world)}


def test_view_assignment_covers_views_evenly():
    for world in (1, 2, 4, 8):
        seen = {}
        for step in range(12):
            for r in range(world):
                v = par.view_for(step, r, world, 3)
                seen[v] = seen.get(v, 0) + 1
        assert set(seen) == {0, 1, 2}
        assert max(seen.values()) - min(seen.values()) <= 0 if (12 * world) % 3 == 0 else 1


def test_bucket_with_active_sh_coefficients():
    """Only the first (deg+1)^2 SH coefficients travel; the rest of the gradient stays as the rasterizer wrote it."""
    P = 11
    params = [torch.zeros(P, 3), torch.zeros(P, 16, 3), torch.zeros(P, 1)]
    for deg, k in ((0, 1), (1, 4), (2, 9), (3, 16)):
        b = par.GradBucket(params, active_dim1={1: k})
        assert b.nbytes == P * (3 + 3 * k + 1) * 4
        g = [torch.randn(P, 3), torch.randn(P, 16, 3), torch.randn(P, 1)]
        g[1][:, k:] = 0.0
        for p, gg in zip(params, g):
            p.grad = gg.clone()
        b.reduce_grads(params)
        for p, gg in zip(params, g):
            assert torch.equal(p.grad, gg)


def test_single_process_bucket_is_identity():
    params = [torch.randn(5, 3), torch.randn(5, 2, 3)]
    for p in params:
        p.grad = torch.randn_like(p)
    ref = [p.grad.clone() for p in params]
    b = par.GradBucket(params)
    b.reduce_grads(params)
    for p, r in zip(params, ref):
        assert torch.equal(p.grad, r)


def test_world_of_one_helpers_are_identity_without_a_process_group():
    assert par.sync_rng(7) == 7
    a = torch.rand(4)
    assert par.sync_rng(7) == 7 and torch.equal(torch.rand(4), a)
    t = torch.arange(5.0)
    assert par.broadcast_from_rank0(t) is t
