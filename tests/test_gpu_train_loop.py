"""End-to-end: the gradients of the HIP path drive Adam to fit a perturbed scene back to its ground-truth renders
(the reference's train.py loop reduced to the hot path)."""
import importlib.util
import os

import pytest
import torch

import parity_utils as pu
from scgaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fit_improves_psnr():
    spec = importlib.util.spec_from_file_location("fit_synthetic", os.path.join(ROOT, "examples", "fit_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.fit(iters=150, P=3000, W=192, H=128, verbose=False)
    first, last = hist[0], hist[-1]
    assert last[1] < 0.6 * first[1], hist          # loss down by > 40 %
    assert last[2] > first[2] + 3.0, hist          # PSNR up by > 3 dB


def _dp_worker(rank, world, port, ret):
    """One data-parallel step on rank's view; both ranks share GPU 0 (gloo stages CUDA tensors through the host)."""
    import os
    import torch.distributed as dist
    from scgaussian_amd import parallel as par, rasterizer as R
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    par.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    P, W, H = 3000, 160, 96
    sc = syn.make_scene(P, W, H, seed=5)
    cams = [syn.orbit_camera(W, H, yaw, 3.0, 7.0) for yaw in (-8.0, 9.0)]
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]

    dens = {}                                                # view -> (|dL/dmean2D| of the visible Gaussians, visible, radii)

    def grads_for(view):
        for p in params:
            p.grad = None
        st = pu.hip_settings(cams[view], 3, (0.0, 0.0, 0.0))
        means, shs, opac, scales, rots = params
        means2D = torch.zeros_like(means, requires_grad=True)       # screenspace_points of the reference's render()
        c, radii, d, a = R.GaussianRasterizer(st)(means3D=means, means2D=means2D, shs=shs, opacities=opac,
                                                  scales=scales, rotations=rots)
        torch.autograd.backward([c, d, a], ups)
        vis = radii > 0
        dens[view] = (torch.norm(means2D.grad[:, :2], dim=-1, keepdim=True) * vis[:, None], vis, radii.float())
        return [p.grad.clone() for p in params]

    single = [grads_for(v) for v in range(world)]
    expect = [sum(g) / world for g in zip(*single)]
    grads_for(par.view_for(0, rank, world, world))
    assert R.grad_arena(params) is not None                 # autograd kept the arena views as .grad
    bucket = par.GradBucket(params, active_dim1={1: 16})
    assert not bucket.active
    bucket.reduce_grads(params)                             # zero-copy path: all-reduce of the arena in place
    for p, e in zip(params, expect):
        assert pu.nrm_err(p.grad, e) < 1e-6
    # fallback path (gradients that are not arena views) gives the same
    for p, g in zip(params, single[par.view_for(0, rank, world, world)]):
        p.grad = g.clone()
    assert R.grad_arena(params) is None
    bucket.reduce_grads(params)
    for p, e in zip(params, expect):
        assert pu.nrm_err(p.grad, e) < 1e-6
    # densification state on the GPU path (reference scene/gaussian_model.py:932-934 add_densification_stats, train.py:192):
    # every rank accumulates the statistics of ITS view from the HIP path's outputs (means2D.grad, radii), the exchange
    # makes them what one process that had rendered both views would hold
    my = par.view_for(0, rank, world, world)
    g2d, vis, radii = dens[my]
    acc = torch.zeros(P, 1, device=dev); den = torch.zeros(P, 1, device=dev); rad = torch.zeros(P, device=dev)
    acc[vis] += g2d[vis]
    den[vis] += 1
    rad[vis] = torch.max(rad[vis], radii[vis])
    par.reduce_densification_stats(acc, den, rad)
    e_acc = sum(dens[v][0] for v in range(world))
    e_den = sum(dens[v][1].float()[:, None] for v in range(world))
    e_rad = torch.stack([dens[v][2] * dens[v][1] for v in range(world)]).max(0).values
    assert acc.is_cuda and pu.nrm_err(acc, e_acc) < 1e-6 and torch.equal(den, e_den) and torch.equal(rad, e_rad)
    assert float(den.max()) == world and float(acc.abs().max()) > 0
    ret[rank] = 1
    dist.destroy_process_group()


def test_data_parallel_step_zero_copy_bucket_two_ranks_one_gpu():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: 1, 1: 1}


def _rccl_world1_worker(port, ret):
    """The N-GPU exchange path executed by ONE process on the RCCL backend (a world of one is legal): process-group
    creation with a bound device, ReduceOp.AVG on the gradient arena in place, the bucket path, the densification-state
    reductions, the RNG broadcast, barrier and max-over-ranks.  Identity is the expected result of every one of them."""
    import os
    import torch.distributed as dist
    from scgaussian_amd import parallel as par, rasterizer as R
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rank, world, local_rank = par.init_from_env("nccl", force=True)
    assert (rank, world, local_rank) == (0, 1, 0)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and par._exchanging()
    dev = torch.device("cuda", 0)
    P, W, H = 3000, 160, 96
    sc = syn.make_scene(P, W, H, seed=5)
    cam = syn.orbit_camera(W, H, -8.0, 3.0, 7.0)
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]
    means, shs, opac, scales, rots = params
    means2D = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = R.GaussianRasterizer(pu.hip_settings(cam, 3, (0.0, 0.0, 0.0)))(
        means3D=means, means2D=means2D, shs=shs, opacities=opac, scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], ups)
    want = [p.grad.clone() for p in params]
    arena = R.grad_arena(params)
    assert arena is not None and arena.is_cuda
    # a subset of the parameters yields only its own span of the arena (never another parameter's gradient)
    sub = R.grad_arena([opac])
    assert sub is not None and sub.numel() == opac.numel() and sub.data_ptr() == opac.grad.data_ptr()
    assert R.grad_arena([means, opac]) is None or R.grad_arena([means, opac]).numel() <= means.numel() + opac.numel() + 6
    bucket = par.GradBucket(params, active_dim1={1: 16})
    bucket.reduce_grads(params)                              # RCCL all-reduce (AVG) of the arena where it lies
    torch.cuda.synchronize()
    for p, w_ in zip(params, want):
        assert torch.equal(p.grad, w_)
    bucket3 = par.GradBucket(params, active_dim1={1: 4})     # SH-degree-limited bucket: pack -> RCCL AVG -> unpack
    bucket3.reduce_grads(params)
    torch.cuda.synchronize()
    for p, w_ in zip(params, want):
        assert torch.equal(p.grad, w_)
    vis = radii > 0
    acc = torch.norm(means2D.grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
    den = vis.float()[:, None].clone()
    rad = radii.float() * vis
    a0, d0, r0 = acc.clone(), den.clone(), rad.clone()
    par.reduce_densification_stats(acc, den, rad)            # RCCL SUM + MAX
    assert torch.equal(acc, a0) and torch.equal(den, d0) and torch.equal(rad, r0) and float(acc.abs().max()) > 0
    cpu_state = torch.get_rng_state()
    seed = par.sync_rng()                                    # RCCL broadcast of the seed, CUDA generator seeded
    assert torch.equal(torch.get_rng_state(), cpu_state)     # ... and ONLY that generator: the CPU stream is the training loop's
    s1 = torch.normal(mean=torch.zeros(64, 3, device=dev), std=torch.ones(64, 3, device=dev))
    assert par.sync_rng(seed) == seed
    s2 = torch.normal(mean=torch.zeros(64, 3, device=dev), std=torch.ones(64, 3, device=dev))
    assert torch.equal(s1, s2)
    t = torch.arange(8.0, device=dev)
    assert torch.equal(par.broadcast_from_rank0(t), torch.arange(8.0, device=dev))
    par.barrier()
    assert par.max_over_ranks(3.5, dev) == 3.5
    ret[0] = 1
    dist.destroy_process_group()


def test_rccl_backend_world_of_one_executes_every_exchange_step():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_world1_worker, args=(port, ret))
    p.start()
    p.join(240)
    assert p.exitcode == 0, p.exitcode
    assert dict(ret) == {0: 1}


def _run_bench(args, timeout=900):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks_when_asked_for_more_than_one_gpu():
    """`python bench.py --gpus 2` as a PLAIN subprocess (no torchrun, no WORLD_SIZE): the bench starts its two ranks itself,
    they meet in one process group (gloo: both share this box's GPU — a code-path check, not a scaling number), rank 0 prints
    ONE line whose n_gpus is the size of the group the ranks actually saw."""
    res, line = _run_bench(["--gpus", "2", "--dist-backend", "gloo", "--workload", "S1", "--steps", "5", "--warmup", "2",
                            "--sustained-steps", "0"])
    assert res.returncode == 0, res.stderr[-3000:]
    assert line is not None and len([ln for ln in res.stdout.splitlines() if ln.startswith("{")]) == 1
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2
    assert line["config"]["launched_by"].startswith("bench.py")
    assert line["config"]["dist_backend"] == "gloo" and line["config"]["grad_bucket_bytes"] > 0
    assert line["stage_ms"].get("grad_allreduce", 0) > 0                 # the exchange ran inside the timed region
    assert line["value"] > 0


def test_bench_refuses_an_rccl_launch_it_cannot_place_and_a_world_size_that_disagrees():
    """RCCL places one rank per device: `--gpus N` with fewer than N visible GPUs must exit non-zero with a message — never
    run fewer ranks than asked for.  The same for an outer launcher whose WORLD_SIZE disagrees with --gpus."""
    import subprocess
    import sys
    n = torch.cuda.device_count() + 1
    res, line = _run_bench(["--gpus", str(n), "--workload", "S1", "--steps", "2", "--warmup", "1"], timeout=300)
    assert res.returncode != 0 and line is None
    assert "refusing" in res.stderr and "GPU" in res.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
