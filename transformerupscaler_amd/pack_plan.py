"""Re-packing the weights after an optimizer step as two gather launches (training path).

`packing.pack_state_dict(..., backward=True)` turns the module's parameters into the layouts the kernels read.  Every one of
its outputs is a pure gather of parameter elements (permute / index_select / flip / cat / zero padding / dtype cast) -- no
arithmetic -- so there is one integer map per output: element i <- parameter element map[i] (or zero).  The map is not
written by hand a second time: it is DERIVED from packing.py by tracing.  The pack is run once per base-256 digit of the
(1-based) flat parameter index with every parameter holding that digit of its own indices (0..255 are exact in bf16, so the
digits survive the casts); the packed outputs then hold the digits of their source indices -- three packs, on the host.  The plan is checked against a real pack of the real weights (bit-equal) before its
first use, so a packing function that stops being a pure gather is caught (the module then keeps the torch path).

Per step this replaces ~75 aten launches (reference train.py:139 moves every weight, so the pack runs once per step) by
`tup_pack_gather` x 2 (csrc/pack_plan.hip)."""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch

from . import _lib


class PackPlan:
    def __init__(self, params: List[Tuple[str, torch.Tensor]], pack_fn: Callable[[Dict[str, torch.Tensor]], Dict[str, torch.Tensor]]):
        """params: (name, parameter) in a fixed order; pack_fn: state-dict-like {name: tensor} -> {key: packed tensor}."""
        dev = params[0][1].device
        self.names = [n for n, _ in params]
        self.shapes = [tuple(p.shape) for _, p in params]
        sizes = [p.numel() for _, p in params]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        total = offs[-1]
        if total + 1 >= 2 ** 31:
            raise ValueError("parameter count exceeds the int32 maps")
        self.total = total
        # ---- trace: one pack per base-256 digit of (flat index + 1), on the host (packing.py is device-agnostic; this keeps the
        #      tracing launches off the GPU -- the plan is built inside the first training step) ----
        ndig = (int(total + 1).bit_length() + 7) // 8         # base-256 digits: 0..255 are exact in bf16 (8-bit significand)
        cpu = torch.device("cpu")
        ids = torch.arange(1, total + 1, dtype=torch.int64)
        acc: Dict[str, torch.Tensor] = {}
        self.consts: Dict[str, object] = {}
        meta: Dict[str, Tuple[torch.dtype, Tuple[int, ...]]] = {}
        for d in range(ndig):
            plane = ((ids >> (8 * d)) & 255).to(torch.float32)
            sd = {n: plane[offs[i]:offs[i + 1]].view(self.shapes[i]) for i, n in enumerate(self.names)}
            out = pack_fn(sd)
            for k, t in out.items():
                if not torch.is_tensor(t):                # constants of the layout (e.g. a block count) pass through
                    self.consts[k] = t
                    continue
                if d == 0:
                    if t.dtype not in (torch.bfloat16, torch.float32):
                        raise TypeError(f"packed tensor {k}: unexpected dtype {t.dtype}")
                    meta[k] = (t.dtype, tuple(t.shape))
                    acc[k] = torch.zeros(t.numel(), dtype=torch.int64)
                v = t.reshape(-1).to(device=cpu, dtype=torch.float32)
                vi = v.to(torch.int64)
                if not bool(((vi.to(torch.float32) == v) & (vi >= 0) & (vi <= 255)).all()):
                    raise ValueError(f"packed tensor {k} is not a pure gather of the parameters")
                acc[k] |= vi << (8 * d)
        # ---- layout: all bf16 outputs in one buffer, all fp32 outputs in another (each tensor 256-byte aligned) ----
        self.views: Dict[str, Tuple[torch.dtype, int, Tuple[int, ...]]] = {}
        maps = {torch.bfloat16: [], torch.float32: []}
        fill = {torch.bfloat16: 0, torch.float32: 0}
        for k, (dt, shape) in meta.items():
            align = 128 if dt == torch.bfloat16 else 64
            pad = (-fill[dt]) % align
            if pad:
                maps[dt].append(torch.full((pad,), -1, dtype=torch.int32))
                fill[dt] += pad
            self.views[k] = (dt, fill[dt], shape)
            maps[dt].append((acc[k] - 1).to(torch.int32))
            fill[dt] += acc[k].numel()
        self.map_bf16 = (torch.cat(maps[torch.bfloat16]) if maps[torch.bfloat16] else torch.empty(0, dtype=torch.int32)).to(dev)
        self.map_f32 = (torch.cat(maps[torch.float32]) if maps[torch.float32] else torch.empty(0, dtype=torch.int32)).to(dev)
        self.offs = torch.tensor(offs, device=dev, dtype=torch.int32)
        self._ptrs_host: List[int] = []
        self._ptrs = torch.zeros(len(params), device=dev, dtype=torch.int64)
        self.device = dev

    def matches(self, params: List[Tuple[str, torch.Tensor]]) -> bool:
        return (len(params) == len(self.names) and all(n == m for (n, _), m in zip(params, self.names))
                and all(tuple(p.shape) == s for (_, p), s in zip(params, self.shapes)) and params[0][1].device == self.device)

    def run(self, params: List[Tuple[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        """The packed tensors of the parameters' current values (fresh buffers: the previous ones may still be referenced by
        a saved-for-backward context)."""
        from .ops import _stream
        for _, p in params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("PackPlan expects contiguous fp32 parameters")
        ptrs = [p.data_ptr() for _, p in params]
        if ptrs != self._ptrs_host:                       # optimizer steps are in place: the table changes only on .to() / load
            self._ptrs.copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._ptrs_host = ptrs
        out_bf16 = torch.empty(self.map_bf16.numel(), device=self.device, dtype=torch.bfloat16)
        out_f32 = torch.empty(self.map_f32.numel(), device=self.device, dtype=torch.float32)
        n = len(params)
        _lib.call("tup_pack_gather", self._ptrs.data_ptr(), self.offs.data_ptr(), n, self.map_bf16.data_ptr(), out_bf16.data_ptr(),
                  out_bf16.numel(), 1, _stream())
        _lib.call("tup_pack_gather", self._ptrs.data_ptr(), self.offs.data_ptr(), n, self.map_f32.data_ptr(), out_f32.data_ptr(),
                  out_f32.numel(), 0, _stream())
        bufs = {torch.bfloat16: out_bf16, torch.float32: out_f32}
        pk = dict(self.consts)
        for k, (dt, off, shape) in self.views.items():
            numel = 1
            for s in shape:
                numel *= s
            pk[k] = bufs[dt][off:off + numel].view(shape)
        return pk


_PLANS: Dict[tuple, object] = {}          # process-wide: a plan depends on the layout (names, shapes, pack function), not on the values


def packed_with_plan(module, key, sd: Dict[str, torch.Tensor], pack_fn):
    """pack_fn(sd) through a PackPlan: built once per process and layout (`key` names the pack function, e.g. ("ft", scale)),
    verified bit for bit against pack_fn on the weights it is first used with; None when no plan applies (CPU tensors, a
    non-gather pack) -- the caller then runs pack_fn itself.  `module` only scopes the key (its class name is part of it)."""
    params = list(sd.items())
    if not params or not params[0][1].is_cuda or any(p.dtype != torch.float32 or not p.is_contiguous() for _, p in params):
        return None
    ck = (type(module).__name__, key, params[0][1].device, tuple(n for n, _ in params), tuple(tuple(p.shape) for _, p in params))
    plan = _PLANS.get(ck)
    if plan is False:
        return None
    if plan is not None:
        return plan.run(params)
    try:
        plan = PackPlan(params, pack_fn)
        ref, got = pack_fn(sd), plan.run(params)
        ok = set(ref) == set(got) and all(
            (ref[k].dtype == got[k].dtype and torch.equal(ref[k], got[k])) if torch.is_tensor(ref[k]) else ref[k] == got[k] for k in ref)
    except (ValueError, TypeError):
        ok = False
    _PLANS[ck] = plan if ok else False
    return got if ok else None
