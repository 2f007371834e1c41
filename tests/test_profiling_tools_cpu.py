"""The profiling tools' bookkeeping (no GPU): which workload a dispatch belongs to, and what the counters' stamp covers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_the_dense_forward_blend_is_keyed_by_its_four_quadrant_waves():
    """tile_blend_forward_kernel<3584, 2048, 8, 8> launches eight waves per tile, four of which only sort: the tools key a
    forward by tiles x 4 x 64 whatever the variant, and the slice kernels in front of it inherit that key."""
    import _workload_tag as wt
    dense = "void scg::tile_blend_forward_kernel<3584, 2048, 8, 8>(scg::FrameDev, HIP_vector_type<unsigned int, 2u> const*)"
    usual = "void scg::tile_blend_forward_kernel<1536, 1024, 8, 4>(scg::FrameDev, HIP_vector_type<unsigned int, 2u> const*)"
    assert wt.norm_grid(dense, 1044480) == 522240
    assert wt.norm_grid(usual, 774144) == 774144
    assert wt.norm_grid("scg::blend_backward_kernel(scg::FrameDev)", 522240) == 522240
    rows = [dict(Kernel_Name="void scg::geometry_hist_kernel<3>(scg::FrameDev)", Dispatch_Id="1", Grid_Size_X="262144"),
            dict(Kernel_Name="scg::tile_scatter_kernel(HIP_vector_type<unsigned int, 2u> const*)", Dispatch_Id="2", Grid_Size_X="526336"),
            dict(Kernel_Name=dense, Dispatch_Id="3", Grid_Size_X="1044480"),
            dict(Kernel_Name="void scg::geometry_hist_kernel<3>(scg::FrameDev)", Dispatch_Id="4", Grid_Size_X="262144"),
            dict(Kernel_Name=usual, Dispatch_Id="5", Grid_Size_X="774144")]
    assert wt.tags(rows, "Grid_Size_X") == {1: 522240, 2: 522240, 4: 774144}


def test_the_counters_stamp_covers_the_paths_kernels_and_nothing_next_to_it(tmp_path, monkeypatch):
    """A change to the image-loss / match-loss / 3-NN kernels (same library, no counters) must not withhold the path's counters
    from the bench line; a change to one of the path's kernels must."""
    import pmc_summary as ps
    csrc = tmp_path / "scgaussian_amd" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    for name in ("blend.hip", "geometry.hip", "scg_common.h", "loss.hip", "matchloss.hip", "knn.hip"):
        (csrc / name).write_text("// " + name)
    (tmp_path / "include" / "scg_raster.h").write_text("// abi")
    monkeypatch.setattr(ps, "ROOT", str(tmp_path))
    base = ps.kernel_source_sha256()
    (csrc / "loss.hip").write_text("// another loss kernel")
    (csrc / "matchloss.hip").write_text("// another match loss kernel")
    assert ps.kernel_source_sha256() == base
    (csrc / "blend.hip").write_text("// another blend kernel")
    assert ps.kernel_source_sha256() != base
