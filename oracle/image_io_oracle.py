"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the frame conversions either side of the model
(SURVEY.md 8(f) rank 1).  Only tests/ may import this.

  to_tensor    torchvision ``ToTensor`` on a uint8 HWC image: ``img.permute(2, 0, 1).float().div(255)``
               (third-party torchvision 0.21 ``functional.to_tensor``; reference call sites
               data_handling/data_class.py:61-71, inference.py:65-75)
  to_frame     ``(x * 255).clamp(0, 255).to(torch.uint8).permute(1, 2, 0)`` (+ ``[..., [2, 1, 0]]`` for BGR), reference
               app_overlay.py:381-388; ToPILImage's ``pic.mul(255).byte()`` for inputs already in [0, 1]
               (inference.py:123-124)
Pinned by tests/test_image_io.py against those torch expressions evaluated here (bit-exact)."""
import numpy as np


def to_tensor(frames_u8: np.ndarray, bgr: bool = False) -> np.ndarray:
    """uint8 [B][H][W][3] -> float32 [B][3][H][W]."""
    f = frames_u8[..., ::-1] if bgr else frames_u8
    return (np.ascontiguousarray(f.transpose(0, 3, 1, 2)).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def to_frame(x: np.ndarray, bgr: bool = False) -> np.ndarray:
    """float32 [B][3][H][W] -> uint8 [B][H][W][3], truncating like tensor.to(torch.uint8)."""
    v = np.clip(x.astype(np.float32) * np.float32(255.0), np.float32(0.0), np.float32(255.0))
    out = np.ascontiguousarray(v.astype(np.uint8).transpose(0, 2, 3, 1))
    return np.ascontiguousarray(out[..., ::-1]) if bgr else out
