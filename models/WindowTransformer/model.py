"""Drop-in plugin module ``models.WindowTransformer.model`` (reference models/WindowTransformer/model.py);
implementation in transformerupscaler_amd.window_transformer (MI355X HIP path, inference)."""
from transformerupscaler_amd.window_transformer import TransformerModel  # noqa: F401

__all__ = ["TransformerModel"]
