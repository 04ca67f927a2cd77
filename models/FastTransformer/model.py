"""Drop-in plugin module: ``importlib.import_module("models.FastTransformer.model").TransformerModel``
(the lookup every reference driver performs: train.py:49-50, inference.py:57-58, speed_test.py:34-35).
The implementation lives in transformerupscaler_amd.fast_transformer (MI355X HIP path)."""
from transformerupscaler_amd.fast_transformer import (TransformerModel, WindowAttention,  # noqa: F401
                                                      WindowTransformerBlock)
from .utils import BasicConv, Upsampler  # noqa: F401

__all__ = ["TransformerModel"]
