"""fullsubnet_plus_amd - MI355X (gfx950) native FullSubNet+ inference forward.

Drop-in for the reference's plugin point (``[model].path`` in config/inference.toml:27):

    [model]
    path = "fullsubnet_plus_amd.model.FullSubNet_Plus"     # or, for the original FullSubNet
    path = "fullsubnet_plus_amd.fullsubnet.Model"          # (reference: fullsubnet.model.fullsubnet.Model)

The arithmetic runs in hand-written HIP kernels behind the C ABI of ``libfsnp_hip.so``
(include/fsnp.h); PyTorch is only the tensor container.  There is no CPU fallback.
"""
from .model import FullSubNet, FullSubNet_Plus, Model  # noqa: F401

__all__ = ["FullSubNet_Plus", "Model", "FullSubNet"]
