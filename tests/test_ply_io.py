"""On-disk scene format: PLY round trip, header layout, activations, render-from-file."""
import os

import numpy as np
import pytest
import torch

from scgaussian_amd import ply_io


def _model(P=37, Nb=11, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    rayd = torch.nn.functional.normalize(r(P, 3))
    return ply_io.RayBoundModel(features_dc=r(P, 1, 3), features_rest=r(P, 15, 3) * 0.1, opacity=r(P, 1),
                                scaling=r(P, 3) - 3, rotation=r(P, 4), zval=torch.rand(P, 1, generator=g) * 5 + 1,
                                rayo=r(P, 3), rayd=rayd, bg_xyz=r(Nb, 3) * 10, bg_features_dc=r(Nb, 1, 3),
                                bg_features_rest=r(Nb, 15, 3) * 0.1, bg_opacity=r(Nb, 1), bg_scaling=r(Nb, 3) - 2,
                                bg_rotation=r(Nb, 4))


def test_round_trip_and_header(tmp_path):
    m = _model()
    path = os.path.join(tmp_path, "point_cloud", "iteration_30000", "point_cloud.ply")
    ply_io.save_ply(path, m)
    folder = os.path.dirname(path)
    assert sorted(os.listdir(folder)) == ["point_cloud.ply", "point_cloud_bg.ply", "point_cloud_color.ply"]
    head = open(path, "rb").read(4096).split(b"end_header\n")[0].decode().split("\n")
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [h.split()[2] for h in head[3:] if h.startswith("property")]
    # reference order: scene/gaussian_model.py:531-549
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9:54] == [f"f_rest_{i}" for i in range(45)]
    assert names[54:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "zval_0",
                          "rayo_0", "rayo_1", "rayo_2", "rayd_0", "rayd_1", "rayd_2"]
    assert all(h.split()[1] == "float" for h in head[3:] if h.startswith("property"))
    assert os.path.getsize(path) == len("\n".join(head)) + len("end_header\n") + 37 * 69 * 4    # 69 float columns

    back = ply_io.load_ply(path)
    for k in ("features_dc", "features_rest", "opacity", "scaling", "rotation", "zval", "rayo", "rayd", "bg_xyz",
              "bg_features_dc", "bg_features_rest", "bg_opacity", "bg_scaling", "bg_rotation"):
        assert torch.equal(getattr(back, k), getattr(m, k)), k
    # x y z column = rayo + rayd * zval, SH stored channel-major (transpose(1,2).flatten)
    props = ply_io.read_vertex_ply(path)
    xyz = np.stack([props["x"], props["y"], props["z"]], 1)
    assert np.array_equal(xyz, (m.rayo + m.rayd * m.zval).numpy())
    assert np.array_equal(props["f_rest_15"], m.features_rest[:, 0, 1].numpy())      # first G coefficient
    assert np.array_equal(props["f_dc_2"], m.features_dc[:, 0, 2].numpy())
    col = ply_io.read_vertex_ply(os.path.join(folder, "point_cloud_color.ply"))
    assert col["red"].dtype == np.uint8 and col["x"].shape == (48,)


def test_getters_match_reference_activations(tmp_path):
    m = _model(P=5, Nb=3)
    assert m.get_xyz.shape == (8, 3) and torch.equal(m.get_xyz[:5], m.rayo + m.rayd * m.zval)
    assert m.get_features.shape == (8, 16, 3)
    assert torch.allclose(m.get_opacity, torch.sigmoid(torch.cat([m.opacity, m.bg_opacity])))
    assert torch.allclose(m.get_scaling, torch.exp(torch.cat([m.scaling, m.bg_scaling])))
    assert torch.allclose(m.get_rotation.norm(dim=1), torch.ones(8))
    assert m.active_sh_degree == 3


def test_reads_ascii_big_endian_and_reordered_files(tmp_path):
    m = _model(P=4, Nb=0)
    path = os.path.join(tmp_path, "a", "point_cloud.ply")
    ply_io.save_ply(path, m, write_color_ply=False)
    assert not os.path.exists(os.path.join(tmp_path, "a", "point_cloud_bg.ply"))
    props = ply_io.read_vertex_ply(path)
    names = list(props)[::-1]                                  # reversed property order, ascii encoding
    asc = os.path.join(tmp_path, "b", "point_cloud.ply")
    os.makedirs(os.path.dirname(asc))
    with open(asc, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment written by a test\nelement vertex 4\n")
        fh.write("".join(f"property float {n}\n" for n in names) + "end_header\n")
        for i in range(4):
            fh.write(" ".join(repr(float(props[n][i])) for n in names) + "\n")
    back = ply_io.load_ply(asc)
    assert torch.equal(back.rotation, m.rotation) and torch.equal(back.features_rest, m.features_rest)
    big = os.path.join(tmp_path, "c", "point_cloud.ply")
    os.makedirs(os.path.dirname(big))
    rec = np.empty(4, dtype=[(n, ">f4") for n in props])
    for n in props:
        rec[n] = props[n]
    with open(big, "wb") as fh:
        fh.write(("ply\nformat binary_big_endian 1.0\nelement vertex 4\n" +
                  "".join(f"property float {n}\n" for n in props) + "end_header\n").encode())
        fh.write(rec.tobytes())
    assert torch.equal(ply_io.load_ply(big).zval, m.zval)
    with pytest.raises(ValueError):
        ply_io.load_ply(path, max_sh_degree=2)                 # 45 f_rest columns do not fit degree 2


@pytest.mark.gpu
def test_render_from_loaded_file_matches_render_from_memory(tmp_path):
    from scgaussian_amd import synthetic as syn
    from scgaussian_amd.render import render, PipelineParams
    dev = torch.device("cuda", 0)
    m = _model(P=600, Nb=200, seed=3)
    m.rayo = torch.zeros(600, 3)
    m.rayd = torch.nn.functional.normalize(torch.randn(600, 3) * torch.tensor([0.4, 0.3, 0.05]) + torch.tensor([0, 0, 1.0]))
    m.zval = torch.rand(600, 1) * 6 + 3
    m.bg_xyz = torch.randn(200, 3) * torch.tensor([3.0, 2.0, 1.0]) + torch.tensor([0, 0, 9.0])
    path = os.path.join(tmp_path, "point_cloud.ply")
    ply_io.save_ply(path, m)
    loaded = ply_io.load_ply(path, device=dev)
    cam = syn.default_camera(96, 64).to(dev)
    bg = torch.zeros(3, device=dev)
    a = render(cam, m.to(dev), PipelineParams(), bg)
    b = render(cam, loaded, PipelineParams(), bg)
    assert int((a["radii"] > 0).sum()) > 100
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["rendered_depth"], b["rendered_depth"])
