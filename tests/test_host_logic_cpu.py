"""Host-side logic of the binding that needs no GPU (round 6): the capacity policy, what a settled count does to a camera's capacity,
camera identity without a device read, the model path's attribute mapping and its refusal of models the kernels cannot take, the
counters' summary per SH degree."""
import os
import sys
import types

import pytest
import torch

from scgaussian_amd import model_path as mp
from scgaussian_amd import rasterizer as R
from scgaussian_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capacity_policy_head_room_and_stability():
    for count in (0, 1, 4095, 22_000, 1_285_028, 5_455_250, 123_456_789):
        cap = R._capacity_for(count)
        assert cap >= count + count // 8 and cap <= 2 * count + 65536 + 8192, (count, cap)
        # a count that moves by a per cent keeps the capacity (the workspace plan keyed by it stays cached)
        assert R._next_capacity(cap, int(count * 1.01)) == cap
        assert R._next_capacity(cap, count) == cap
        # growth beyond the head room, or a count that fell to less than half: re-derived
        assert R._next_capacity(cap, cap) > cap
        if count > 200_000:
            assert R._next_capacity(cap, count // 3) < cap
    assert R._next_capacity(None, 1000) == R._capacity_for(1000)


def test_a_settled_count_moves_the_cameras_capacity():
    spec = types.SimpleNamespace(hint={}, cam_hint={}, pending={})
    key = (640, 480, b"camera")
    w = R._CountWord()
    w.slot, w.np, w.ptr, w.cap, w.P, w.key, w.device_index, w.captured = 0, None, 0, 100_000, 5000, key, 0, False
    before = dict(R._OVERFLOW)
    R._settle_word(spec, w, 80_000)                              # inside the capacity: kept
    assert spec.cam_hint[key] == (100_000, 80_000, 5000) and spec.hint[(5000, 640, 480)] == 100_000
    assert R._OVERFLOW["overflows"] == before["overflows"] and R._OVERFLOW["settled"] == before["settled"] + 1
    R._settle_word(spec, w, 150_000)                             # clipped: counted, room for the count next time
    assert R._OVERFLOW["overflows"] == before["overflows"] + 1
    assert spec.cam_hint[key][0] >= 150_000 + 150_000 // 8 and spec.cam_hint[key][1:] == (150_000, 5000)


def test_camera_identity_without_a_device_read():
    a, b = torch.eye(4), torch.eye(4)
    b[3, 2] = 1.5
    assert R._camera_key(a) == R._camera_key(a.clone()) != R._camera_key(b)      # host memory: the content itself
    assert len(R._CAM_KEYS) == 0 or all(not isinstance(k, bytes) for k in R._CAM_KEYS)
    R.tag_camera(b, ("scene", 7))
    assert R._camera_key(b) == b"id:('scene', 7)"
    assert R._camera_key(None) == b""


def test_model_path_maps_the_reference_models_attribute_names():
    """scene/gaussian_model.py:452-468 keeps `_zval`, `_features_dc`, ... with a leading underscore and the background set without;
    a model without a background set gets empty stand-ins; on the CPU the model path declines (render() takes the getters)."""
    sc = syn.make_scene(50, 64, 48)
    m = syn.make_raw_model(sc, ray_fraction=1.0)
    ref_like = types.SimpleNamespace(**{"_" + k: getattr(m, k) for k in ("zval", "rayo", "rayd", "features_dc", "features_rest",
                                                                          "opacity", "scaling", "rotation")})
    t = mp.tensors_of(ref_like)
    assert t is not None and t["zval"] is m.zval and t["bg_xyz"].shape == (0, 3) and t["bg_features_rest"].shape == (0, 15, 3)
    assert tuple(t) == mp.ARG_NAMES
    assert mp.tensors_of(types.SimpleNamespace(_zval=m.zval)) is None           # not a Gaussian model
    assert not mp.supported(t)                                                   # CPU tensors: no kernels for them
    assert mp.model_for(m) is None and mp.model_for(ref_like) is None
    bad = dict(t, scaling=t["scaling"].double())
    assert not mp.supported(bad)


def test_counter_summary_keeps_the_degrees_apart(tmp_path, monkeypatch):
    """tools/pmc_summary.py `<dirs> <dirs>@deg0`: the second collection's workloads are stored as S2_deg0 — the kernels carry
    the same names at every degree, so they must never be averaged into the headline workload's entry."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as ps

    def collection(d, fetch_kib):
        p = d / "p1"
        p.mkdir(parents=True)
        rows = ["Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value"]
        for i, (name, grid) in enumerate((("void scg::geometry_hist_kernel<3>(scg::FrameDev)", 262144),
                                          ("void scg::tile_blend_forward_kernel<1536, 1024, 8, 4>(scg::FrameDev)", 774144),
                                          ("scg::blend_backward_kernel(scg::FrameDev)", 774144)), start=1):
            rows.append(f'{i},"{name}",{grid},FETCH_SIZE,{fetch_kib}')
            rows.append(f'{i},"{name}",{grid},WRITE_SIZE,10')
        (p / "c_counter_collection.csv").write_text("\n".join(rows) + "\n")
        return str(d)
    a, b = collection(tmp_path / "pmc", 1000), collection(tmp_path / "pmc_deg0", 400)
    out = tmp_path / "summary.json"
    monkeypatch.setattr(ps, "static_mix", lambda: {})
    monkeypatch.setattr(sys, "argv", ["pmc_summary.py", a, b + "@deg0", "--out", str(out)])
    ps.main()
    import json
    d = json.loads(out.read_text())
    assert d["S2"]["blend_backward"]["hbm_bytes"] == (2 * 1000 + 10) * 1024
    assert d["S2_deg0"]["blend_backward"]["hbm_bytes"] == (2 * 400 + 10) * 1024
    assert d["S2"]["geometry_forward"]["hbm_bytes"] != d["S2_deg0"]["geometry_forward"]["hbm_bytes"]
    assert "kernel_source_sha256" in d["_stamp"]
