"""K views of the same Gaussians in one autograd node (GaussianRasterizerViews; BASELINE cfg5 "multi-view batched step"):
the images are those of K single-view calls bit for bit, the parameter gradients — accumulated IN THE KERNEL by the
second and later views (scg_backward accumulate, include/scg_raster.h) — are those of the sum of the K single-view
backward passes, they live in one flat arena, and two data-parallel ranks that each accumulate K views and exchange once
hold the mean over all N*K views."""
import os
import socket

import pytest
import torch

import parity_utils as pu
from scgaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _scene(P=4000, W=176, H=112, seed=8):
    sc = syn.make_scene(P, W, H, seed=seed)
    cams = [syn.orbit_camera(W, H, yaw, pitch, 7.0) for yaw, pitch in ((-9.0, 2.0), (0.0, 0.0), (8.0, -3.0), (14.0, 1.0))]
    return sc, cams, W, H


def _leaves(sc, mode, cam, deg, dev):
    return {k: v.detach().to(dev).requires_grad_(True) for k, v in pu.run_oracle_inputs(sc, cam, deg, 1.0, mode).items()
            if k != "means2D"}


def _kw(lv):
    return {k: v for k, v in lv.items() if k not in ("means3D", "opacities")}


@pytest.mark.parametrize("mode", ["sh_sr", "col_cov"])
@pytest.mark.parametrize("use", [(0, 1, 2), (0, 2)])
def test_views_node_equals_the_sum_of_single_view_calls(mode, use):
    from scgaussian_amd import rasterizer as R
    dev = torch.device("cuda")
    sc, cams, W, H = _scene()
    deg, bg, K = 3, (0.1, 0.3, 0.2), 3
    setts = [pu.hip_settings(c, deg, bg) for c in cams[:K]]
    ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=20 + k)) for k in range(K)]
    P = sc.means3D.shape[0]

    # reference: K single-view calls, autograd sums the gradients
    lv = _leaves(sc, mode, cams[0], deg, dev)
    m2 = [torch.zeros(P, 3, device=dev, requires_grad=True) for _ in range(K)]
    singles = [R.GaussianRasterizer(setts[k])(means3D=lv["means3D"], means2D=m2[k], opacities=lv["opacities"], **_kw(lv))
               for k in range(K)]
    loss = sum((singles[k][0] * ups[k][0]).sum() + (singles[k][2] * ups[k][1]).sum() + (singles[k][3] * ups[k][2]).sum()
               for k in use)
    loss.backward()

    # one node
    lv2 = _leaves(sc, mode, cams[0], deg, dev)
    m2s = torch.zeros(K, P, 3, device=dev, requires_grad=True)
    outs = R.GaussianRasterizerViews(setts)(means3D=lv2["means3D"], means2D=m2s, opacities=lv2["opacities"], **_kw(lv2))
    assert len(outs) == K
    for k in range(K):
        for a, b in zip(outs[k], singles[k]):
            assert torch.equal(a, b)                          # images and radii: bit-identical
    loss2 = sum((outs[k][0] * ups[k][0]).sum() + (outs[k][2] * ups[k][1]).sum() + (outs[k][3] * ups[k][2]).sum()
                for k in use)
    loss2.backward()
    torch.cuda.synchronize()
    for name in lv:
        assert lv2[name].grad is not None, name
        pu.assert_close(lv2[name].grad, lv[name].grad, ("views node", mode, use, name))
    for k in range(K):
        want = m2[k].grad if k in use else torch.zeros(P, 3, device=dev)
        pu.assert_close(m2s.grad[k], want, ("views node", mode, use, "means2D", k))
    # one flat arena behind all parameter gradients (what the data-parallel exchange all-reduces in place)
    assert R.grad_arena(list(lv2.values())) is not None


def test_views_node_against_the_oracle():
    """The accumulated gradient against the CPU oracle's autograd over the same K views (not only against our own
    single-view path)."""
    from oracle import torch_rasterizer as orc
    from scgaussian_amd import rasterizer as R
    dev = torch.device("cuda")
    sc, cams, W, H = _scene(P=2500, W=128, H=96)
    deg, bg, K = 2, (0.0, 0.0, 0.0), 2
    ups = [syn.make_upstream_grads(W, H, seed=30 + k) for k in range(K)]
    ol = {k: v.clone().requires_grad_(True) for k, v in pu.run_oracle_inputs(sc, cams[0], deg, 1.0, "sh_sr").items()
          if k != "means2D"}
    P = sc.means3D.shape[0]
    loss = 0.0
    for k in range(K):
        c, r, d, a = orc.rasterize(ol["means3D"], torch.zeros(P, 3), ol["opacities"], pu.oracle_settings(cams[k], deg, bg),
                                   shs=ol["shs"], scales=ol["scales"], rotations=ol["rotations"])
        loss = loss + (c * ups[k][0]).sum() + (d * ups[k][1]).sum() + (a * ups[k][2]).sum()
    loss.backward()
    lv = _leaves(sc, "sh_sr", cams[0], deg, dev)
    m2s = torch.zeros(K, P, 3, device=dev, requires_grad=True)
    outs = R.GaussianRasterizerViews([pu.hip_settings(c, deg, bg) for c in cams[:K]])(
        means3D=lv["means3D"], means2D=m2s, opacities=lv["opacities"], **_kw(lv))
    sum((outs[k][0] * ups[k][0].to(dev)).sum() + (outs[k][2] * ups[k][1].to(dev)).sum() +
        (outs[k][3] * ups[k][2].to(dev)).sum() for k in range(K)).backward()
    torch.cuda.synchronize()
    for name in lv:
        pu.assert_close(lv[name].grad, ol[name].grad, ("views node vs oracle", name))


def _dp_views_worker(rank, world, port, ret):
    """N = 2 ranks (sharing GPU 0, exchange through gloo) x K = 2 views each: accumulate in the kernel, ONE exchange."""
    import torch.distributed as dist
    from scgaussian_amd import parallel as par, rasterizer as R
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    par.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    sc, cams, W, H = _scene(P=3000, W=160, H=96)
    K, deg = 2, 3
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
    means, shs, opac, scales, rots = params
    ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=40 + v)) for v in range(world * K)]
    P = means.shape[0]

    def view_loss(out, v):
        return ((out[0] * ups[v][0]).sum() + (out[2] * ups[v][1]).sum() + (out[3] * ups[v][2]).sum()) / K

    # what one process that rendered all N*K views would hold: the mean over N*K views
    for p in params:
        p.grad = None
    for v in range(world * K):
        out = R.GaussianRasterizer(pu.hip_settings(cams[v], deg, (0.0, 0.0, 0.0)))(
            means3D=means, means2D=torch.zeros(P, 3, device=dev, requires_grad=True), opacities=opac, shs=shs, scales=scales,
            rotations=rots)
        (view_loss(out, v) / world).backward()
    expect = [p.grad.clone() for p in params]
    # this rank's K views in one node, one backward, one exchange
    for p in params:
        p.grad = None
    mine = [par.view_for(0, rank, world, world) * K + k for k in range(K)]       # rank r: views rK .. rK+K-1
    outs = R.GaussianRasterizerViews([pu.hip_settings(cams[v], deg, (0.0, 0.0, 0.0)) for v in mine])(
        means3D=means, means2D=torch.zeros(K, P, 3, device=dev, requires_grad=True), opacities=opac, shs=shs, scales=scales,
        rotations=rots)
    sum(view_loss(outs[k], mine[k]) for k in range(K)).backward()
    assert R.grad_arena(params) is not None
    bucket = par.GradBucket(params)
    bucket.reduce_grads(params)                              # zero-copy: the accumulated arena, all-reduced in place
    for p, e in zip(params, expect):
        assert pu.nrm_err(p.grad, e) < 2e-6, pu.nrm_err(p.grad, e)
    ret[rank] = 1
    dist.destroy_process_group()


def test_two_ranks_accumulate_k_views_then_one_exchange_equals_the_mean_over_all_views():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_dp_views_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: 1, 1: 1}


@pytest.mark.parametrize("M,deg", [(9, 2), (4, 1), (16, 1)])
def test_views_node_with_short_sh_records_and_low_active_degree(M, deg):
    """SH storage of fewer than 16 coefficients takes the un-staged gradient path of the geometry backward (the accumulation
    then happens coefficient by coefficient in global memory), and an active degree below the stored one leaves the upper
    coefficients' gradients at exactly zero also when accumulating."""
    from scgaussian_amd import rasterizer as R
    dev = torch.device("cuda")
    sc, cams, W, H = _scene(P=2500, W=144, H=96)
    K, bg = 3, (0.0, 0.1, 0.2)
    setts = [pu.hip_settings(c, deg, bg) for c in cams[:K]]
    ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=60 + k)) for k in range(K)]
    P = sc.means3D.shape[0]
    shs_cpu = sc.shs[:, :M].contiguous()

    def leaves():
        return dict(means3D=sc.means3D.to(dev).requires_grad_(True), opacities=sc.opacities.to(dev).requires_grad_(True),
                    shs=shs_cpu.to(dev).requires_grad_(True), scales=sc.scales.to(dev).requires_grad_(True),
                    rotations=sc.rotations.to(dev).requires_grad_(True))
    a = leaves()
    loss = 0.0
    for k in range(K):
        o = R.GaussianRasterizer(setts[k])(means3D=a["means3D"], means2D=torch.zeros(P, 3, device=dev, requires_grad=True),
                                           opacities=a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
        loss = loss + (o[0] * ups[k][0]).sum() + (o[2] * ups[k][1]).sum() + (o[3] * ups[k][2]).sum()
    loss.backward()
    b = leaves()
    outs = R.GaussianRasterizerViews(setts)(means3D=b["means3D"], means2D=torch.zeros(K, P, 3, device=dev, requires_grad=True),
                                            opacities=b["opacities"], shs=b["shs"], scales=b["scales"], rotations=b["rotations"])
    sum((outs[k][0] * ups[k][0]).sum() + (outs[k][2] * ups[k][1]).sum() + (outs[k][3] * ups[k][2]).sum() for k in range(K)).backward()
    torch.cuda.synchronize()
    for n in a:
        pu.assert_close(b[n].grad, a[n].grad, ("views node, short SH records", M, deg, n))
    active = (deg + 1) ** 2
    if active < M:
        assert float(b["shs"].grad[:, active:].abs().max()) == 0.0
    assert float(b["shs"].grad[:, :active].abs().max()) > 0.0
