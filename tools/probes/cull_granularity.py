"""How many blend trips would a finer cull granularity save?  (8x4 half-wave lists, 4x4 second-level cull.)

For every entry of every tile list of a named workload, evaluates the kernel's exact ellipse-vs-rectangle test
(blend.hip: splat_hits_rect) against the 8x8 quadrants the kernels use today, against 8x4 half quadrants and against 4x4
blocks, and reports
  trips_8x8          = sum of quadrant hits                      (what one wave per quadrant walks today)
  trips_half_lb      = sum over quadrants of max(hits of the upper half, hits of the lower half)
                       (a wave that blends TWO splats per trip, one per 32-lane half: LOWER bound, no chunk imbalance)
  lanes_8x8 / 8x4 / 4x4 = share of the lanes of a trip whose 4x4 block the splat reaches, at each granularity
Early termination is ignored on both sides.  Profiling aid; runs on the GPU through forward_stages.
"""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from scgaussian_amd import rasterizer as R, synthetic as syn  # noqa: E402


def hits_rect(a, b, x0, y0, w, h):
    """splat_hits_rect of blend.hip for rectangles of pixel centres [x0, x0+w-1] x [y0, y0+h-1] (vectorised)."""
    dx0 = x0 - a[:, 0]; dy0 = y0 - a[:, 1]
    dx1 = dx0 + (w - 1); dy1 = dy0 + (h - 1)
    zero = torch.zeros_like(dx0)
    med3 = lambda p, q, r: torch.maximum(torch.minimum(p, q), torch.minimum(torch.maximum(p, q), r))
    nx = med3(zero, dx0, dx1); ny = med3(zero, dy0, dy1)
    ca, cb, cc, thr, slope = a[:, 2], a[:, 3], b[:, 0], b[:, 2], b[:, 3]
    ya = med3(slope * nx, dy0, dy1)
    qa = nx * (ca * nx + 2 * cb * ya) + cc * ya * ya
    xb = med3(-cb * ny / ca, dx0, dx1)
    qb = ny * (cc * ny + 2 * cb * xb) + ca * xb * xb
    return ~((qa > thr) & (qb > thr))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "S2"
    w = syn.WORKLOADS[name]
    P, W, H = w["P"], w["width"], w["height"]
    dev = torch.device("cuda:0")
    sc = syn.make_scene(P, W, H, seed=0).to(dev)
    setts = bench.settings_for(bench.make_views(W, H)[0], 3, torch.zeros(3, device=dev), dev)
    fs = R.forward_stages(setts, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
    splats = fs["splats"].view(P, 3, 4)
    plist = fs["point_list"].long()
    ranges = fs["ranges"].long()
    n_tiles = ranges.shape[0]
    gx = (W + 15) // 16
    lens = ranges[:, 1] - ranges[:, 0]
    tile_of = torch.repeat_interleave(torch.arange(n_tiles, device=dev), lens)
    a = splats[plist, 0]; b = splats[plist, 1]
    tx = (tile_of % gx).float() * 16; ty = (tile_of // gx).float() * 16
    out = {"workload": name, "entries": int(plist.numel())}
    trips8 = 0; half_lb = 0; lanes = {"8x8": [0, 0], "8x4": [0, 0], "4x4": [0, 0]}
    for q in range(4):
        x0 = tx + (q & 1) * 8; y0 = ty + (q >> 1) * 8
        h8 = hits_rect(a, b, x0, y0, 8, 8)
        hA = hits_rect(a, b, x0, y0, 8, 4) & h8
        hB = hits_rect(a, b, x0, y0 + 4, 8, 4) & h8
        blocks = [hits_rect(a, b, x0 + 4 * (k & 1), y0 + 4 * (k >> 1), 4, 4) & h8 for k in range(4)]
        nb = sum(x.long() for x in blocks)                       # 4x4 blocks of the quadrant the splat reaches
        trips8 += int(h8.sum())
        HA = torch.zeros(n_tiles, device=dev, dtype=torch.long).index_add_(0, tile_of, hA.long())
        HB = torch.zeros(n_tiles, device=dev, dtype=torch.long).index_add_(0, tile_of, hB.long())
        half_lb += int(torch.maximum(HA, HB).sum())
        lanes["8x8"][0] += int(nb[h8].sum()); lanes["8x8"][1] += 4 * int(h8.sum())
        # 8x4 halves: a half is walked when it is hit; its two 4x4 blocks are the lanes
        for hh, ks in ((hA, (0, 1)), (hB, (2, 3))):
            lanes["8x4"][0] += int(sum(blocks[k][hh].long().sum() for k in ks)); lanes["8x4"][1] += 2 * int(hh.sum())
        lanes["4x4"][0] += int(nb.sum()); lanes["4x4"][1] += int(nb.sum())
    out["trips_8x8"] = trips8
    out["hit_rate_8x8"] = round(trips8 / (4 * plist.numel()), 4)
    out["trips_half_lower_bound"] = half_lb
    out["half_over_8x8"] = round(half_lb / trips8, 4)
    out["blocks_reached_per_trip_lanes"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in lanes.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
