"""Plugin path of the original FullSubNet, mirroring the reference's ``fullsubnet.model.fullsubnet.Model``
(speech_enhance/fullsubnet/model/fullsubnet.py:12; selected by ``[model].path`` in the inference TOML)."""
from .model import FullSubNet

Model = FullSubNet

__all__ = ["Model", "FullSubNet"]
