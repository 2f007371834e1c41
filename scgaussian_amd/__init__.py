"""scgaussian_amd — MI355X-native differentiable Gaussian rasterizer (hand-written HIP for gfx950)
behind the reference's GaussianRasterizer / GaussianRasterizationSettings API
(gaussian_renderer/__init__.py:15,38-53,100-108)."""
__version__ = "0.1.0"


def single_gpu_host_setup() -> None:
    """One line for a training script that drives ONE GPU from ONE Python thread (the reference's train.py):
    run autograd's backward on the calling thread.  PyTorch hands CUDA backward nodes to a per-device worker thread;
    the two hand-offs per `loss.backward()` cost more host time than this rasterizer's whole forward at the
    reference's own scene size (measured on the MI355X box, 10 k Gaussians @ 256x256: 0.237 -> 0.148 ms per training
    step).  A process that runs backward passes of several devices concurrently should not call this."""
    import torch
    torch.autograd.set_multithreading_enabled(False)
