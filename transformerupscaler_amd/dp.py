"""Single-node data-parallel training: one process per GPU, RCCL all-reduce of gradient buckets over
xGMI, launched from inside the backward so the reduction overlaps the remaining backward kernels.

The reference has no distributed code at all; this is the DP layer required by
BASELINE.json.  Design for MI355X: the whole active gradient set is 4.49 M floats (17.9 MB fp32), so it
lives in ONE flat fp32 buffer with a STATIC layout in backward-completion order (tail convs -> blocks 5..0
-> patch_embed -> up-branch -> conv2/conv1) cut into a few multi-MB buckets; each bucket is one all-reduce
on a dedicated HIP stream (xGMI is point-to-point, so few large messages beat many small ones).  Buckets
are always issued in index order, so every rank issues the same sequence of collectives whatever order its
gradients arrive in.

Two layouts:
 * ``scale=s``  -- every rank trains scale ``s`` in every step (the benchmark configuration).  Only the
   parameters active at that scale are in the buffer; the other scales' upsamplers keep ``grad is None``
   and Adam skips them exactly as in the reference.  A gradient outside the layout raises.
 * ``scales=(2, 3, 4, 6)`` -- ranks / steps may train different scales (the reference dataset mixes
   scales inside one step, data_handling/data_class.py:34-45, train.py:119-133; SURVEY §8(e) "wrinkle").
   The buffer spans the union; a rank contributes zeros for what it did not produce ("missing grads are
   zeros for the reduce only") and the ranks OR their active-parameter bitmasks on the HOST (a small CPU
   collective over a gloo side group, started in begin() and collected in finish(): no device readback inside
   the backward), so a parameter gets a gradient iff at least one rank produced one (otherwise ``None`` ->
   Adam skips it on all ranks).

One backward node = one begin() / on_ready()... / finish() episode; with train.py's per-sample loop
(several forwards, one ``loss.backward()``) every sample's node runs its own episode, so all ranks must
run the same number of forward calls per step.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .weights import VALID_SCALES, active_param_names, param_shapes


def _ft_backward_key(name: str):
    """Sort key = the order in which autograd.backward_train finalises FastTransformer gradients."""
    if name.startswith("final_upscale_conv."):
        return (0, 0)
    if name.startswith("final_upscale.upsamplers."):
        return (1, -int(name.split(".")[3]))
    if name.startswith("decoder_conv2."):
        return (2, 0)
    if name.startswith("decoder_conv1."):
        return (3, 0)
    if name.startswith("patch_unembed."):
        return (4, 0)
    if name.startswith("window_blocks."):
        return (5, -int(name.split(".")[1]))
    if name.startswith("patch_embed."):
        return (6, 0)
    if name.startswith("up1_conv."):
        return (7, 0)
    if name.startswith("up1.upsamplers."):
        return (8, -int(name.split(".")[3]))
    if name.startswith("conv2."):
        return (9, 0)
    if name.startswith("conv1."):
        return (10, 0)
    return (11, 0)


class GradReducer:
    def __init__(self, scale: Optional[int], device, process_group=None, bucket_mb: float = 6.0, names: Optional[List[str]] = None,
                 shapes: Optional[Dict[str, tuple]] = None, scales: Optional[Sequence[int]] = None):
        """scale: fixed FastTransformer training scale; scales: several scales (mixed-scale layout); or pass scale=None with
        explicit `names` (+ `shapes`) in expected backward order for a model whose parameters are all active
        (ResidualTransformer, WindowTransformer)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        shapes = dict(shapes) if shapes is not None else param_shapes()
        if names is not None:
            layout = list(names)
            self.mixed = False
        elif scales is not None:
            scales = tuple(int(s) for s in scales)
            if not scales or any(s not in VALID_SCALES for s in scales):
                raise ValueError(f"scales must be a non-empty subset of {VALID_SCALES}")
            seen = set()
            layout = [n for s in scales for n in active_param_names(s) if not (n in seen or seen.add(n))]
            layout.sort(key=_ft_backward_key)
            self.mixed = len(scales) > 1
        else:
            layout = sorted(active_param_names(scale), key=_ft_backward_key)
            self.mixed = False
        self.names: List[str] = layout
        self.shapes = {n: tuple(shapes[n]) for n in layout}
        self.numel = {n: int(torch.Size(self.shapes[n]).numel()) for n in layout}
        self.device = torch.device(device)
        # static layout, 64-float (256 B) aligned segments
        self.offset: Dict[str, int] = {}
        cur = 0
        for n in layout:
            self.offset[n] = cur
            cur += (self.numel[n] + 63) // 64 * 64
        self.param_floats = cur
        self.index = {n: i for i, n in enumerate(layout)}
        self.total_floats = total = cur
        # mixed layout: which parameters got a gradient on SOME rank is exchanged on the host (62 names per int64 word, OR-reduced)
        self._host_group = None
        if self.mixed and dist.is_initialized() and self.world > 1:
            if dist.get_backend(process_group) == "gloo":
                self._host_group = process_group
            else:                                            # collective call: every rank builds its reducer at the same point
                ranks = dist.get_process_group_ranks(process_group if process_group is not None else dist.group.WORLD)
                self._host_group = dist.new_group(ranks=ranks, backend="gloo")
        self._mask_words = (len(layout) + 61) // 62
        self._mask_work = None
        self._mask = None
        self.flat = None          # allocated per backward episode: finish() hands out views of it, which stay valid (and may be
                                  # accumulated into by train.py's per-sample loop) while the next episode fills a fresh buffer
        # buckets of >= bucket_mb in layout order
        bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        self.bucket_ranges: List[tuple] = []
        self.bucket_of: Dict[str, int] = {}
        a = 0
        for n in layout:
            self.bucket_of[n] = len(self.bucket_ranges)
            end = self.offset[n] + (self.numel[n] + 63) // 64 * 64
            if end - a >= bucket_elems:
                self.bucket_ranges.append((a, end))
                a = end
        if a < cur or not self.bucket_ranges:
            self.bucket_ranges.append((a, cur))
        for n in layout:                                     # a trailing short bucket was merged into a new last one above
            self.bucket_of[n] = min(self.bucket_of[n], len(self.bucket_ranges) - 1)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._works: list = []
        self._in_step = False
        self._active: Optional[frozenset] = None
        self._arrived: set = set()
        self._pending: List[int] = []
        self._next_bucket = 0
        self._mask_cache: Dict[frozenset, torch.Tensor] = {}
        self.launched_order: List[int] = []                  # bucket indices in issue order of the last episode (tests)

    # ---- episode start: called by the autograd node before its first gradient ----
    def begin(self, active: Optional[Iterable[str]] = None) -> None:
        if self._in_step:
            raise RuntimeError("GradReducer.begin() called while a backward episode is still open (finish() missing)")
        if active is None:
            if self.mixed:
                raise RuntimeError("a mixed-scale GradReducer needs begin(active_names) before the first gradient")
            act = frozenset(self.names)
        else:
            act = frozenset(active)
            stray = [n for n in act if n not in self.offset]
            if stray:
                raise RuntimeError(f"gradients for {stray[:3]}{'...' if len(stray) > 3 else ''} are not in this reducer's layout: "
                                   "the step's scale differs from DataParallel(scale=...); build it with scales=(...) for mixed-scale steps")
        self._in_step, self._active, self._arrived = True, act, set()
        self._pending = [0] * len(self.bucket_ranges)
        for n in act:
            self._pending[self.bucket_of[n]] += 1
        self._next_bucket = 0
        self.launched_order = []
        # segments this rank will not fill count as zeros in the sum; the alignment gaps between segments are reduced too but
        # never read
        alloc = torch.zeros if len(act) != len(self.names) else torch.empty
        self.flat = alloc(self.total_floats, dtype=torch.float32, device=self.device)
        if self.comm_stream is not None and dist.is_initialized():
            self.flat.record_stream(self.comm_stream)        # the all-reduces run on the side stream
        if self.mixed:
            mask = self._mask_cache.get(act)
            if mask is None:
                words = [0] * self._mask_words
                for n in act:
                    words[self.index[n] // 62] |= 1 << (self.index[n] % 62)
                mask = self._mask_cache[act] = torch.tensor(words, dtype=torch.int64)
            self._mask = mask.clone()
            if self._host_group is not None or (dist.is_initialized() and self.world > 1):
                self._mask_work = dist.all_reduce(self._mask, op=dist.ReduceOp.BOR, group=self._host_group, async_op=True)
        self._advance()

    # ---- called from the backward as soon as a group of gradients is final ----
    def on_ready(self, names: List[str], grads: Dict[str, torch.Tensor]) -> None:
        if not self._in_step:
            self.begin(None)
        dst, src = [], []
        for n in names:
            if n not in self.offset:
                raise RuntimeError(f"gradient {n} is not in this reducer's layout (scale mismatch with DataParallel(scale=...))")
            if n not in self._active:
                raise RuntimeError(f"gradient {n} was not announced in begin() for this backward")
            if n in self._arrived:
                raise RuntimeError(f"gradient {n} arrived twice in one backward")
            o = self.offset[n]
            dst.append(self.flat[o:o + self.numel[n]].view(self.shapes[n]))
            src.append(grads[n])
            self._arrived.add(n)
            self._pending[self.bucket_of[n]] -= 1
        if len(dst) == 1:
            dst[0].copy_(src[0])
        elif dst:
            torch._foreach_copy_(dst, src)                   # one launch per ready group (a block's 13 gradients) instead of 13
        self._advance()

    def _advance(self) -> None:
        """Issue every bucket whose predecessors are issued and whose own gradients are all in (index order only)."""
        nb = len(self.bucket_ranges)
        while self._next_bucket < nb and self._pending[self._next_bucket] == 0:
            self._launch(self._next_bucket)
            self._next_bucket += 1

    def _launch(self, k: int) -> None:
        a, b = self.bucket_ranges[k]
        self.launched_order.append(k)
        if b <= a or not dist.is_initialized():
            return                      # no process group: single process, nothing to reduce
        # (a 1-rank group still issues its collectives: the RCCL / stream / event path is then the one N ranks run)
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
        """Wait for every all-reduce, average, and hand back per-parameter views of the flat buffer."""
        if not self._in_step:
            self.begin(None)
        missing = self._active - self._arrived
        if missing:
            self._abort()
            raise RuntimeError(f"backward finished without gradients for {sorted(missing)[:3]} (announced in begin())")
        self._advance()
        assert self._next_bucket == len(self.bucket_ranges)
        for w in self._works:
            w.wait()
        if self.comm_stream is not None and dist.is_initialized():
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._works = []
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        if self.mixed:
            if self._mask_work is not None:
                self._mask_work.wait()                       # host collective, started in begin(): no device sync here
                self._mask_work = None
            words = self._mask.tolist()
            have = [n for n in self.names if (words[self.index[n] // 62] >> (self.index[n] % 62)) & 1]
        else:
            have = [n for n in self.names if n in self._active]
        out = {n: self.flat[self.offset[n]:self.offset[n] + self.numel[n]].view(self.shapes[n]) for n in have}
        self._in_step, self._active = False, None
        return out

    def _abort(self):
        for w in self._works:
            w.wait()
        self._works = []
        if self._mask_work is not None:
            self._mask_work.wait()
            self._mask_work = None
        self._in_step, self._active = False, None


class DataParallel:
    """Attach a GradReducer to the model's backward.  Usage (mirrors train.py's loop, one process per GPU):

        torch.cuda.set_device(local_rank); dist.init_process_group("nccl")       # "nccl" is RCCL on ROCm
        model = TransformerModel().to(f"cuda:{local_rank}")
        dp = DataParallel(model, scale=2)            # or scales=(2, 3, 4, 6) for mixed-scale steps
        loss = criterion(resize(model(lr, ...)), hr); loss.backward(); optimizer.step()

    Parameters are broadcast from rank 0 at construction so every replica starts identical (and the module's
    packed-weight cache is dropped, since the broadcast writes the parameters in place)."""

    def __init__(self, module, scale: Optional[int] = None, process_group=None, bucket_mb: float = 6.0,
                 scales: Optional[Sequence[int]] = None):
        self.module = module
        dev = next(module.parameters()).device
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            raise RuntimeError(f"the module lives on {dev} but the current device is cuda:{torch.cuda.current_device()}: "
                               "call torch.cuda.set_device(local_rank) before building the model (kernels launch on the current device)")
        if scale is None and scales is None:   # every parameter is active (ResidualTransformer): names / shapes from the module
            named = [(n, tuple(p.shape)) for n, p in module.named_parameters() if p.requires_grad]
            named.reverse()                    # backward finalises parameters roughly in reverse definition order
            self.reducer = GradReducer(None, dev, process_group, bucket_mb, names=[n for n, _ in named], shapes=dict(named))
        else:
            self.reducer = GradReducer(scale, dev, process_group, bucket_mb, scales=scales)
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():
                for p in module.parameters():
                    dist.broadcast(p.data, src=0, group=process_group)
        if hasattr(module, "invalidate_packed"):
            module.invalidate_packed()
        module._grad_reducer = self.reducer

    def detach(self):
        self.module._grad_reducer = None
