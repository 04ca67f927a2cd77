"""Single-node data-parallel training: one process per GPU, RCCL all-reduce of gradient buckets over
xGMI, launched from inside the backward so the reduction overlaps the remaining backward kernels.

The reference has no distributed code at all; this is the DP layer required by
BASELINE.json.  Design for MI355X: the whole active gradient set is 4.49 M floats (17.9 MB fp32), so it
lives in ONE flat fp32 buffer cut into a few multi-MB buckets in backward-completion order (tail convs
-> blocks 5..0 -> patch_embed -> up-branch -> conv2/conv1); each bucket is one ring/tree all-reduce
on a dedicated HIP stream (xGMI is point-to-point, so few large messages beat many small ones).
Only parameters active at the training scale are reduced; the other scales' upsamplers keep
``grad is None`` and Adam skips them exactly as in the reference.  All ranks must train the
same scale in a step.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import autograd as _ag
from .weights import active_param_names, param_shapes


class GradReducer:
    def __init__(self, scale: Optional[int], device, process_group=None, bucket_mb: float = 6.0, names: Optional[List[str]] = None,
                 shapes: Optional[Dict[str, tuple]] = None):
        """scale: FastTransformer training scale (selects the active parameter set); pass scale=None with explicit
        `names` + `shapes` for a model whose parameters are all active (ResidualTransformer)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        shapes = dict(shapes) if shapes is not None else param_shapes()
        self.names = list(names) if names is not None else active_param_names(scale)
        self.shapes = {n: shapes[n] for n in self.names}
        self.numel = {n: int(torch.Size(self.shapes[n]).numel()) for n in self.names}
        self.device = torch.device(device)
        self.flat = torch.zeros(sum(self.numel.values()), dtype=torch.float32, device=self.device)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.offset: Dict[str, int] = {}        # assigned lazily in arrival order (= backward order)
        self._cursor = 0
        self._bucket_start = 0
        self._works = []
        self._layout_frozen = False
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.bucket_ranges: List[tuple] = []

    # ---- called from the backward (autograd.grad_ready_hook) ----
    def on_ready(self, names: List[str], grads: Dict[str, torch.Tensor]) -> None:
        for n in names:
            if n not in self.numel:
                continue
            if n not in self.offset:
                if self._layout_frozen:
                    raise RuntimeError(f"gradient {n} arrived that was not part of the first step's layout")
                self.offset[n] = self._cursor
                self._cursor += self.numel[n]
            o = self.offset[n]
            self.flat[o:o + self.numel[n]].copy_(grads[n].reshape(-1))
            self._filled = max(getattr(self, "_filled", 0), o + self.numel[n])
        if self._filled - self._bucket_start >= self.bucket_elems:
            self._launch(self._bucket_start, self._filled)
            self._bucket_start = self._filled

    def _launch(self, a: int, b: int) -> None:
        if b <= a:
            return
        if not self._layout_frozen:
            self.bucket_ranges.append((a, b))
        if self.world == 1:
            return
        view = self.flat[a:b]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- called at the end of the backward ----
    def finish(self) -> Dict[str, torch.Tensor]:
        """Flush the last bucket, wait for every all-reduce, average, and hand back per-parameter views."""
        self._launch(self._bucket_start, getattr(self, "_filled", 0))
        for w in self._works:
            w.wait()
        if self.comm_stream is not None and self.world > 1:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._works = []
        if self.world > 1:
            self.flat[:self._cursor].mul_(1.0 / self.world)
        out = {n: self.flat[o:o + self.numel[n]].view(self.shapes[n]) for n, o in self.offset.items()}
        self._layout_frozen = True
        self._bucket_start = 0
        self._filled = 0
        return out


class DataParallel:
    """Attach a GradReducer to the model's backward.  Usage (mirrors train.py's loop, one process per GPU):

        dp = DataParallel(model, scale=2)            # after dist.init_process_group("nccl")
        loss = criterion(resize(model(lr, ...)), hr); loss.backward(); optimizer.step()

    Parameters are broadcast from rank 0 at construction so every replica starts identical."""

    def __init__(self, module, scale: Optional[int] = None, process_group=None, bucket_mb: float = 6.0):
        self.module = module
        dev = next(module.parameters()).device
        if scale is None:        # every parameter is active (ResidualTransformer): take names / shapes from the module
            named = {n: tuple(p.shape) for n, p in module.named_parameters() if p.requires_grad}
            self.reducer = GradReducer(None, dev, process_group, bucket_mb, names=list(named), shapes=named)
        else:
            self.reducer = GradReducer(scale, dev, process_group, bucket_mb)
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            for p in module.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
        module._grad_reducer = self.reducer

    def detach(self):
        self.module._grad_reducer = None
