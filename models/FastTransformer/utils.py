"""Parameter-holder counterparts of the reference's EDSR helpers (utils.py:13-98)."""
from transformerupscaler_amd.fast_transformer import BasicConv, Upsampler  # noqa: F401
