"""Drop-in plugin module ``models.ResidualTransformer.model`` (reference models/ResidualTransformer/model.py);
implementation in transformerupscaler_amd.residual_transformer (MI355X HIP path, inference)."""
from transformerupscaler_amd.residual_transformer import TransformerBlock, TransformerModel  # noqa: F401

__all__ = ["TransformerModel"]
