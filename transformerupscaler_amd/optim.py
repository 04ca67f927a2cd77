"""`Adam`: torch.optim.Adam (reference train.py:104 `optim.Adam(model.parameters(), lr=args.lr)`, stepped at train.py:139) with the
update of ALL parameters in one HIP launch (`tup_adam_step`, csrc/pack_plan.hip) instead of torch's ~13 multi-tensor launches per
step.  A drop-in subclass: same constructor, same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter -- a
checkpoint written by one loads into the other), same semantics for parameters without a gradient (skipped: no state, no step;
SURVEY Q3).  Options the kernel does not implement (weight decay, amsgrad, maximize, capturable / differentiable) and parameters it
cannot take (not fp32 / not on the GPU / sparse gradients) fall through to torch's own step, per parameter group."""
from __future__ import annotations

import math
import struct
from typing import List

import torch

from . import _lib

_CHUNK = 4096


class Adam(torch.optim.Adam):
    def _fusable(self, group) -> bool:
        return (group.get("weight_decay", 0) == 0 and not group.get("amsgrad", False) and not group.get("maximize", False)
                and not group.get("capturable", False) and not group.get("differentiable", False)
                and not torch.is_tensor(group["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        recs: List[bytes] = []
        sizes: List[int] = []
        updated: list = []                                     # parameters the launch writes (their version is bumped below)
        keep: list = []                                        # tensors the asynchronous launch reads: alive until the next step
        device = None
        fallback_groups = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            ok = self._fusable(group) and all(
                p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                and not p.grad.is_sparse and p.grad.device == p.device for p in ps)
            if ok and ps:
                if device is None:
                    device = ps[0].device
                ok = all(p.device == device for p in ps)
            if not ok:
                fallback_groups.append(group)
                continue
            beta1, beta2 = group["betas"]
            for p in ps:
                st = self.state[p]
                if len(st) == 0:                              # same lazy state as torch.optim.Adam
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                recs.append(struct.pack("<QQQQqffffff", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                        p.numel(), group["lr"] / bc1, 1.0 / math.sqrt(bc2), beta2, 1.0 - beta1, 1.0 - beta2, group["eps"]))
                sizes.append(p.numel())
                updated.append(p)
                if g is not p.grad:
                    keep.append(g)
        if recs:
            key = tuple(sizes)
            cache = self.__dict__.setdefault("_chunk_cache", {})
            chunks = cache.get((key, device))
            if chunks is None:                                # (segment, first element) per 4096 elements; changes only with the grad set
                tab = []
                for si, n in enumerate(sizes):
                    tab += [(si, off) for off in range(0, n, _CHUNK)]
                chunks = cache[(key, device)] = torch.tensor(tab, dtype=torch.int32).to(device)
            # gradient pointers change every step: the table goes up through one of two pinned staging buffers (a pageable
            # source would make the copy synchronous and stall the host behind the whole backward)
            raw = b"".join(recs)
            stage = self.__dict__.setdefault("_stage", [None, None, 0])
            events = self.__dict__.setdefault("_stage_events", [None, None])
            slot = stage[2] = stage[2] ^ 1
            # the host may run several steps ahead of the GPU (no per-step sync in a training loop): a slot is rewritten only
            # after the asynchronous upload that last read it has executed
            if events[slot] is not None:
                events[slot].synchronize()
            if stage[slot] is None or stage[slot].numel() * 8 < len(raw):
                stage[slot] = torch.empty(len(raw) // 8, dtype=torch.int64).pin_memory()
            host = stage[slot][:len(raw) // 8]
            host.copy_(torch.frombuffer(bytearray(raw), dtype=torch.int64))
            from .ops import _stream
            with torch.cuda.device(device):
                segs = host.to(device, non_blocking=True)
                if events[slot] is None:
                    events[slot] = torch.cuda.Event()
                events[slot].record()
                _lib.call("tup_adam_step", segs.data_ptr(), chunks.data_ptr(), chunks.shape[0], _stream())
            keep.append(segs)
            # the launch writes the parameters through raw pointers: tell autograd (saved-tensor checks) and every cache keyed on
            # (data_ptr, _version) -- the models' packed-weight caches -- that they changed, as an in-place torch update would
            for p in updated:
                torch.autograd.graph.increment_version(p)
        self.__dict__["_keep"] = keep
        if fallback_groups:
            saved = self.param_groups
            self.param_groups = fallback_groups
            try:
                super().step()
            finally:
                self.param_groups = saved
        return loss
