"""`simple_knn._C` twin: distCUDA2(points (N,3) float cuda tensor) -> (N,) float tensor with the mean squared
distance of every point to its 3 nearest other points (call site: reference scene/gaussian_model.py:444).

N < 4 (fewer than three other points): this twin returns the mean over the neighbours that exist (0 for N = 1);
upstream's kernel leaves FLT_MAX-scale values in the missing slots [UPSTREAM-RECALL].  The reference clamps the result
to >= 1e-7 and initialises scales from it (:444-445), so for such degenerate inputs this twin gives finite scales where
upstream gives astronomically large ones — a documented divergence, never reached by the reference's pipelines
(<= 2 000 matches x 6 pairs at init, data_preprocess/get_match_info.py:376)."""
import torch

from scgaussian_amd import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.ScgError("distCUDA2 needs a tensor on the ROCm GPU ('cuda'); there is no CPU path")
    pts = points.detach()
    if pts.dtype != torch.float32:
        pts = pts.float()
    pts = pts.contiguous()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("points must be (N,3)")
    n = pts.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        nbytes = lib.scg_knn3_scratch_bytes(n)
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=pts.device)
        _lib.check(lib.scg_knn3_mean_dist2_ws(pts.data_ptr(), n, out.data_ptr(), scratch.data_ptr(), nbytes,
                                              torch.cuda.current_stream(pts.device).cuda_stream), "scg_knn3_mean_dist2_ws")
    return out
