"""Training-step harness with the semantics of the reference loop (train.py:110-146): zero_grad,
forward with ``res_out = HR size, require_ratio=False``, antialiased Resize to the HR size when shapes
differ, L1 loss, backward, Adam step.  Equal-shaped samples are batched (the reference loops over
samples at B=1; mean of per-sample L1 means == batch L1 mean for equal shapes)."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .autograd import l1_loss, resize_aa

use_torch_adam = False          # A/B attribute


def make_optimizer(model, lr: float = 1e-4):
    """train.py:104 -- Adam, default betas/eps, no weight decay (parameters without grad are skipped).  `optim.Adam` is
    torch.optim.Adam with the whole update in one HIP launch; harness.use_torch_adam = True selects torch's own step (A/B)."""
    if use_torch_adam:
        return torch.optim.Adam(model.parameters(), lr=lr)
    from .optim import Adam
    return Adam(model.parameters(), lr=lr)


def train_step(model, optimizer, lr_batch: torch.Tensor, hr_batch: torch.Tensor) -> torch.Tensor:
    optimizer.zero_grad(set_to_none=True)                                    # train.py:113
    out = model(lr_batch, res_out=tuple(hr_batch.shape[2:]), require_ratio=False)      # train.py:124
    if tuple(out.shape[2:]) != tuple(hr_batch.shape[2:]):
        out = resize_aa(out, tuple(hr_batch.shape[2:]))                      # train.py:127-130
    loss = l1_loss(out, hr_batch, fuse_into_model_backward=True)             # train.py:103,132,136 (HIP forward + backward; `out` feeds nothing else)
    loss.backward()                                                          # train.py:138 (bf16 needs no GradScaler)
    optimizer.step()                                                         # train.py:139
    return loss.detach()


def save_checkpoint(model, checkpoint_dir: str, epoch: int, optimizer=None) -> str:
    """train.py:152-156: weights only, ``model_epoch_{n}.pth`` -- the file the reference's drivers load
    (``model.load_state_dict(torch.load(path))``, train.py:90, inference.py:97, speed_test.py:45), in both directions.
    The reference drops the optimizer state; with `optimizer` it goes to a sidecar ``optim_epoch_{n}.pt`` so a resumed
    run continues Adam's moments without changing the weight file's format."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, f"model_epoch_{epoch}.pth")
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(checkpoint_dir, f"optim_epoch_{epoch}.pt"))
    return path


def load_latest_checkpoint(model, checkpoint_dir: str, optimizer=None, map_location=None) -> int:
    """Resume as train.py:86-92 does (latest ``*_<epoch>.pth``, strict load); returns the epoch (0 if none)."""
    from tools.utils import get_latest_checkpoint
    try:
        path, epoch = get_latest_checkpoint(checkpoint_dir)
    except (FileNotFoundError, OSError):
        return 0
    model.load_state_dict(torch.load(path, map_location=map_location))
    side = os.path.join(checkpoint_dir, f"optim_epoch_{epoch}.pt")
    if optimizer is not None and os.path.exists(side):
        optimizer.load_state_dict(torch.load(side, map_location=map_location))
    return epoch
