#!/usr/bin/env python3
"""Run lengths of the scatter: how many ids does ONE slice of the Gaussians (= one scatter workgroup per band,
tile_scatter_kernel) drop into ONE tile's segment?  Those ids land in consecutive slots of point_list — a run — and the run
length is what a variant that writes runs instead of single 4-byte stores could gain from.  The count of a (slice, tile) pair is
the entry of the stage's own table[B][Tn] before the column scan; here it is rebuilt from the rectangles the geometry stage wrote
(staged forward), with the library's own cut of the Gaussians into slices (tile_walk.h block_slice, B = 256 for these frames).

usage: scatter_runs.py [S2 S3 S4 clustered30 clustered60]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scgaussian_amd import rasterizer as R
from scgaussian_amd import synthetic as syn

dev = torch.device("cuda", 0)
CLUSTERED = syn.CLUSTERED


def frame(name):
    if name in CLUSTERED:
        w = syn.WORKLOADS["S2"]
        frac, spread = CLUSTERED[name]
        sc = syn.make_clustered_scene(w["P"], w["width"], w["height"], frac, spread, seed=0)
    else:
        w = syn.WORKLOADS[name]
        sc = syn.make_scene(w["P"], w["width"], w["height"], seed=0)
    return sc.to(dev), w["width"], w["height"]


def slices_of(P, B):
    nb256 = (P + 255) // 256
    cut = torch.tensor([(b * nb256 // B) * 256 for b in range(B + 1)], device=dev).clamp_(max=P)
    return torch.bucketize(torch.arange(P, device=dev), cut, right=True) - 1


def main():
    names = sys.argv[1:] or ["S2", "S3", "S4", "clustered30", "clustered60"]
    for name in names:
        sc, W, H = frame(name)
        P = sc.means3D.shape[0]
        cam = syn.default_camera(W, H)
        st = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                             cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                             cam.camera_center.to(dev), False, False)
        with torch.no_grad():
            fs = R.forward_stages(st, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        torch.cuda.synchronize()
        rects = fs["rects"].to(torch.int64) & 0xFFFFFFFF
        x0, y0, w, h = rects[:, 0] & 0xFFFF, rects[:, 0] >> 16, rects[:, 1] & 0xFFFF, rects[:, 1] >> 16
        gx, gy = (W + 15) // 16, (H + 15) // 16
        n_tiles = gx * gy
        cnt = w * h
        Rn = int(cnt.sum())
        assert Rn == int(fs["num_rendered"]), (Rn, fs["num_rendered"])
        B = 256 if (Rn >= (1 << 20) or P >= 100_000) else 128
        ids = torch.repeat_interleave(torch.arange(P, device=dev), cnt)
        first = torch.cumsum(cnt, 0) - cnt
        k = torch.arange(Rn, device=dev) - first[ids]
        tile = (y0[ids] + k // w[ids]) * gx + x0[ids] + k % w[ids]
        sl = slices_of(P, B)[ids]
        # a scatter workgroup = (band of tile rows, slice): 8 bands
        runs = torch.bincount(sl * n_tiles + tile, minlength=B * n_tiles)
        runs = runs[runs > 0]
        hist = torch.bincount(runs.clamp(max=33))
        n_runs = int(runs.numel())
        by_id = lambda lo: float(runs[runs >= lo].sum()) / Rn                               # noqa: E731
        print(f"{name}: P {P}, {W}x{H}, {n_tiles} tiles, {B} slices, num_rendered {Rn}: {n_runs} (slice, tile) runs, "
              f"mean run {Rn / n_runs:.2f} ids (median {int(runs.median())}, p90 {int(runs.float().quantile(0.9))}, max {int(runs.max())}); "
              f"ids in runs of >= 2: {by_id(2):.3f}, >= 4: {by_id(4):.3f}, >= 8: {by_id(8):.3f}, >= 16: {by_id(16):.3f}")
        print("   runs by length 1..32, 33+: " + " ".join(str(int(v)) for v in hist[1:].tolist()))
        # what a 64-lane store instruction of the flat (tile-ordered) flush would touch: 64 consecutive staged ids = how many lines?
        lines_direct = Rn                                                                   # one line request per id today
        starts = torch.cumsum(runs, 0) - runs
        # upper bound of line requests of a flush in run order: every run of n ids touches ceil((n + 15) / 16) lines at worst
        lines_flush = int(((runs + 30) // 16).sum())
        print(f"   line requests: {lines_direct} today (one per id); <= {lines_flush} with the ids of a run stored by neighbouring "
              f"lanes ({lines_direct / lines_flush:.2f}x fewer)")


if __name__ == "__main__":
    main()
