#!/usr/bin/env python3
"""Generate tests/golden/oracle_small.npz: outputs of the CPU oracle (oracle/torch_rasterizer.py) on two
small seeded scenes — integer artefacts, images and all gradients.  The scenes are regenerated from their
seeds by scgaussian_amd.synthetic, so only expected outputs are stored.

    python tests/golden/make_oracle_golden.py        (CPU only)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scgaussian_amd import synthetic as syn      # noqa: E402
import parity_utils as pu                        # noqa: E402

CASES = {
    "a": dict(P=700, W=96, H=64, deg=3, bg=(0.0, 0.0, 0.0), mod=1.0, seed=11, cam=("default",), mode="sh_sr",
              log_scale_mean=-3.2),
    "b": dict(P=900, W=120, H=72, deg=1, bg=(0.3, 0.6, 0.9), mod=0.9, seed=12, cam=("orbit", 14.0, -6.0, 7.5),
              mode="sh_sr", log_scale_mean=-3.0),
    "c": dict(P=500, W=80, H=48, deg=2, bg=(1.0, 1.0, 1.0), mod=1.0, seed=13, cam=("orbit", -20.0, 9.0, 6.5),
              mode="col_cov", log_scale_mean=-3.0),
}


def make_case(cfg):
    sc = syn.make_scene(cfg["P"], cfg["W"], cfg["H"], seed=cfg["seed"], log_scale_mean=cfg["log_scale_mean"])
    if cfg["cam"][0] == "default":
        cam = syn.default_camera(cfg["W"], cfg["H"])
    else:
        cam = syn.orbit_camera(cfg["W"], cfg["H"], *cfg["cam"][1:])
    grads = syn.make_upstream_grads(cfg["W"], cfg["H"], seed=cfg["seed"] + 100)
    return sc, cam, grads


def main():
    out = {}
    for name, cfg in CASES.items():
        sc, cam, grads = make_case(cfg)
        o = pu.run_oracle(sc, cam, cfg["deg"], cfg["bg"], cfg["mod"], cfg["mode"], grads=grads)
        b = o["aux"]["binning"]
        out[f"{name}_color"] = o["color"].numpy()
        out[f"{name}_depth"] = o["depth"].numpy()
        out[f"{name}_alpha"] = o["alpha"].numpy()
        out[f"{name}_radii"] = o["radii"].numpy()
        out[f"{name}_point_offsets"] = b["point_offsets"]
        out[f"{name}_keys_sorted"] = b["keys_sorted"]
        out[f"{name}_point_list"] = b["point_list"]
        out[f"{name}_ranges"] = b["ranges"]
        out[f"{name}_final_T"] = o["aux"]["final_T"].numpy()
        out[f"{name}_n_contrib"] = o["aux"]["n_contrib"].numpy()
        for k, g in o["grads"].items():
            out[f"{name}_grad_{k}"] = g.numpy()
        print(name, "R =", b["num_rendered"], "V =", int((o["radii"] > 0).sum()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
