"""Round-3 parity additions (VERDICT r2 "next round" #8 and ADVICE r2): the optimizer really moves the weights the kernels
use, training at the benchmarked per-rank batch is value-checked, and the fused-L1 hand-off fails safe.

All through the plugin surface and the C ABI; the oracle / torch.optim.Adam are the checkers."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ft(det_sd, train=False):
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda()
    return m.train() if train else m.eval()


def _rt(train=False):
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    m = m.cuda()
    return m.train() if train else m.eval()


# ---------------------------------------------------------------- the optimizer step reaches the kernels ------------------
@pytest.mark.parametrize("which", ["FastTransformer", "ResidualTransformer"])
def test_fused_adam_steps_repack_and_match_torch_adam(which, det_sd, monkeypatch):
    """optim.Adam writes the parameters through raw pointers; the packed-weight caches are keyed on
    (data_ptr, _version).  After a step the cache must be rebuilt (the next forward runs on the MOVED weights), and three
    harness steps must follow torch.optim.Adam's trajectory: same losses, same parameters."""
    from transformerupscaler_amd import harness
    from transformerupscaler_amd.autograd import l1_loss
    from transformerupscaler_amd.optim import Adam

    if which == "FastTransformer":
        g = torch.Generator().manual_seed(31)
        lr_b = torch.rand((2, 3, 64, 96), generator=g).cuda()
        hr_b = torch.rand((2, 3, 96, 144), generator=g).cuda()
        make = lambda: _ft(det_sd)                                      # .eval(): no dropout, deterministic trajectory
        step = lambda m, opt: harness.train_step(m, opt, lr_b, hr_b)
        packed = lambda m: m.packed(2, backward=True)[0]
    else:
        g = torch.Generator().manual_seed(32)
        lr_b = torch.rand((1, 3, 720, 1280), generator=g).cuda()
        hr_b = torch.rand((1, 3, 1440, 2560), generator=g).cuda()
        make = lambda: _rt()

        def step(m, opt):
            opt.zero_grad(set_to_none=True)
            loss = l1_loss(m(lr_b, upscale_factor=2), hr_b, fuse_into_model_backward=True)
            loss.backward()
            opt.step()
            return loss.detach()
        packed = lambda m: m.packed(backward=True)

    lr_rate = 1e-3                        # larger than train.py's 1e-4 so three steps move the loss well above rounding
    runs = {}
    for tag in ("fused", "torch"):
        m = make()
        opt = Adam(m.parameters(), lr=lr_rate) if tag == "fused" else torch.optim.Adam(m.parameters(), lr=lr_rate)
        losses, pk_ids = [], []
        for _ in range(3):
            losses.append(float(step(m, opt)))
            pk_ids.append(packed(m))                # the objects themselves: kept alive, so their ids cannot be recycled
        runs[tag] = (m, losses, pk_ids)
    mf, lf, idf = runs["fused"]
    mt, lt, _ = runs["torch"]
    assert len({id(p) for p in idf}) == 3, "the packed-weight cache was not rebuilt after optim.Adam.step()"
    assert abs(lf[1] - lf[0]) > 1e-5 and abs(lf[2] - lf[1]) > 1e-6, f"the loss does not move: {lf} (frozen packed weights?)"
    for a, b in zip(lf, lt):
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3) + 2e-5, (lf, lt)
    # parameters: Adam's first steps are +-lr per element whatever the gradient's size, so an element whose gradient is at the
    # level of the fp32-atomics order noise may flip; everything else must agree closely
    tot = bad = 0
    for (k, pf), (_, pt) in zip(mf.named_parameters(), mt.named_parameters()):
        d = (pf.detach() - pt.detach()).abs()
        tot += d.numel()
        bad += int((d > 0.2 * lr_rate).sum())
    assert bad <= 0.01 * tot, (bad, tot)
    # and the version counter moved exactly once per step for every updated parameter
    p0 = next(p for p in mf.parameters() if p.grad is not None)
    v = p0._version
    step(mf, Adam(mf.parameters(), lr=lr_rate))
    assert p0._version == v + 1


def test_fused_adam_host_runs_ahead_without_corrupting_steps(det_sd):
    """the per-step segment table goes up through two pinned slots; the host of a training loop runs many
    steps ahead of the GPU.  Twelve steps issued back to back (no sync) must equal the same twelve steps with a device
    synchronisation after each."""
    from transformerupscaler_amd import harness
    from transformerupscaler_amd.optim import Adam
    g = torch.Generator().manual_seed(41)
    lr_b = torch.rand((1, 3, 48, 64), generator=g).cuda()
    hr_b = torch.rand((1, 3, 72, 96), generator=g).cuda()
    res = {}
    for tag in ("async", "sync"):
        m = _ft(det_sd)
        opt = Adam(m.parameters(), lr=1e-3)
        harness.train_step(m, opt, lr_b, hr_b)                  # builds the pack plan, sizes the staging slots
        torch.cuda.synchronize()
        if tag == "async":
            torch.cuda._sleep(int(1.5e9))                       # the GPU idles ~0.7 s while the host queues the steps below
        losses = []
        for _ in range(8):
            losses.append(harness.train_step(m, opt, lr_b, hr_b))
            if tag == "sync":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res[tag] = ([float(v) for v in losses], [p.detach().clone() for p in m.parameters()])
    la, pa = res["async"]
    ls, ps = res["sync"]
    # the same trajectory: the first step to rounding (same weights), the later ones to the divergence that the weight-gradient
    # kernels' fp32 atomics (order-dependent last bits) seed and Adam's first, sign-like +-lr steps amplify (1.5e-4 seen at the
    # third loss) -- a step run with another step's lr / bc1 or gradient pointers would be off by percent
    for i, (a, b) in enumerate(zip(la, ls)):
        assert abs(a - b) <= (1e-5 if i < 1 else 2e-3) * abs(b), (i, la, ls)
    tot = bad = 0
    for a, b in zip(pa, ps):
        d = (a - b).abs()
        tot += d.numel(); bad += int((d > 1e-3).sum())          # one full +-lr step apart
    assert bad <= 0.02 * tot, (bad, tot)


# ---------------------------------------------------------------- training at the benchmarked batch -----------------------
def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def test_train_batch4_720p_equals_mean_of_single_image_steps(det_sd):
    """bench.py's FastTransformer training step runs 4 images of 720x1280 per rank; gradients had only been
    value-checked at B = 1 (train_720p.npz).  The B = 4 step's gradient of every parameter must equal the mean of the four
    B = 1 gradients (each of which is the fixture-checked path) to 1e-3 relative; dropout off."""
    from transformerupscaler_amd.autograd import l1_loss, resize_aa
    m = _ft(det_sd)
    g = torch.Generator().manual_seed(4321)
    lr_b = torch.rand((4, 3, 720, 1280), generator=g).cuda()
    hr_b = torch.rand((4, 3, 1080, 1920), generator=g).cuda()

    def step(a, b):
        m.zero_grad(set_to_none=True)
        out = resize_aa(m(a, res_out=(1080, 1920), require_ratio=False), (1080, 1920))
        loss = l1_loss(out, b)
        loss.backward()
        return float(loss), _grads(m)

    loss4, g4 = step(lr_b, hr_b)
    acc, losses = {}, []
    for i in range(4):
        li, gi = step(lr_b[i:i + 1], hr_b[i:i + 1])
        losses.append(li)
        for k, v in gi.items():
            acc[k] = acc.get(k, 0) + v.double() / 4
    assert abs(loss4 - sum(losses) / 4) <= 1e-5
    assert set(g4) == set(acc)
    worst = ("", 0.0)
    for k, v in g4.items():
        rel = (v.double() - acc[k]).norm().item() / max(acc[k].norm().item(), 1e-20)
        if rel > worst[1]:
            worst = (k, rel)
        assert rel <= 1e-3, (k, rel)
    print("B=4 vs mean of four B=1 steps at 720p: worst relative L2", worst)


def test_rt_x6_batch2_equals_mean_of_single_image_steps():
    """The same for BASELINE configs[4]'s per-rank batch: ResidualTransformer x6, 2 images of 720x1280 -> 4320x7680, L1."""
    from transformerupscaler_amd.autograd import l1_loss
    m = _rt()
    g = torch.Generator().manual_seed(9876)
    lr_b = torch.rand((2, 3, 720, 1280), generator=g).cuda()
    hr_b = torch.rand((2, 3, 4320, 7680), generator=g).cuda()

    def step(a, b, fuse):
        m.zero_grad(set_to_none=True)
        loss = l1_loss(m(a, upscale_factor=6), b, fuse_into_model_backward=fuse)
        loss.backward()
        return float(loss), _grads(m)

    loss2, g2 = step(lr_b, hr_b, True)                 # as bench.py runs it (loss gradient formed inside the bicubic backward)
    acc, losses = {}, []
    for i in range(2):
        li, gi = step(lr_b[i:i + 1], hr_b[i:i + 1], False)
        losses.append(li)
        for k, v in gi.items():
            acc[k] = acc.get(k, 0) + v.double() / 2
    assert abs(loss2 - sum(losses) / 2) <= 1e-5
    for k, v in g2.items():
        rel = (v.double() - acc[k]).norm().item() / max(acc[k].norm().item(), 1e-20)
        assert rel <= 1e-3, (k, rel)


# ---------------------------------------------------------------- fused L1 hand-off fails safe ----------------------------
def test_fused_l1_refusal_leaves_the_reducer_usable():
    """when the model node refuses the fused-L1 stand-in (the output has a second consumer) with a gradient
    reducer attached, the refusal must come before the reducer opens its episode: the next backward works."""
    from transformerupscaler_amd.autograd import l1_loss
    from transformerupscaler_amd.dp import DataParallel
    m = _rt()
    dp = DataParallel(m)
    g = torch.Generator().manual_seed(5)
    lr_b = torch.rand((1, 3, 720, 1280), generator=g).cuda()
    hr_b = torch.rand((1, 3, 1440, 2560), generator=g).cuda()
    out = m(lr_b, upscale_factor=2)
    total = l1_loss(out, hr_b, fuse_into_model_backward=True) + 1e-3 * out.mean()
    with pytest.raises(RuntimeError, match="consumer besides the loss"):
        total.backward()
    assert not dp.reducer._in_step
    m.zero_grad(set_to_none=True)
    l1_loss(m(lr_b, upscale_factor=2), hr_b, fuse_into_model_backward=True).backward()       # the reducer is not stuck
    assert all(p.grad is not None for p in m.parameters())
    dp.detach()


def test_fused_l1_is_not_used_when_the_output_gradient_is_observed():
    """under the fusion the gradient flowing into `out` is a stand-in; an `out` that retains its gradient or
    carries a hook gets the materialised sign(out - target) / numel instead."""
    from transformerupscaler_amd.autograd import l1_loss
    m = _rt()
    g = torch.Generator().manual_seed(6)
    lr_b = torch.rand((1, 3, 720, 1280), generator=g).cuda()
    hr_b = torch.rand((1, 3, 1440, 2560), generator=g).cuda()
    out = m(lr_b, upscale_factor=2)
    out.retain_grad()
    l1_loss(out, hr_b, fuse_into_model_backward=True).backward()
    ref = torch.sign(out.detach() - hr_b) / out.numel()
    assert torch.allclose(out.grad, ref, atol=1e-12)
    g_plain = _grads(m)
    m.zero_grad(set_to_none=True)
    seen = []
    out = m(lr_b, upscale_factor=2)
    out.register_hook(lambda gr: seen.append(gr.detach().clone()))
    l1_loss(out, hr_b, fuse_into_model_backward=True).backward()
    assert len(seen) == 1 and torch.allclose(seen[0], ref, atol=1e-12)
    m.zero_grad(set_to_none=True)
    l1_loss(m(lr_b, upscale_factor=2), hr_b, fuse_into_model_backward=True).backward()          # fused path: same gradients
    for k, v in _grads(m).items():
        rel = (v - g_plain[k]).norm().item() / max(g_plain[k].norm().item(), 1e-20)
        assert rel <= 1e-3, (k, rel)


# ---------------------------------------------------------------- fused L1 in the FastTransformer training step -----------
@pytest.mark.parametrize("path", ["harness_resize", "model_resize", "no_resize"])
def test_fast_transformer_fused_l1_equals_materialised_gradient(path, det_sd):
    """autograd.l1_loss(..., fuse_into_model_backward=True) for the FastTransformer step (train.py:124-138): the loss gradient is formed
    inside the first backward kernel -- of the Resize of train.py:127-130 (`harness_resize`: model output 96 x 144 squashed to 72 x 108),
    of the model's own Resize + clamp (`model_resize`, require_ratio=True) or of its clamp (`no_resize`) -- and every parameter
    gradient equals the one computed from the materialised sign(out - target) / numel."""
    from transformerupscaler_amd.autograd import l1_loss, resize_aa
    g = torch.Generator().manual_seed(77)
    lr_b = torch.rand((2, 3, 48, 72), generator=g).cuda()
    hw = (96, 144) if path == "no_resize" else (72, 108)
    hr_b = torch.rand((2, 3) + hw, generator=g).cuda()
    res = {}
    for fuse in (False, True):
        m = _ft(det_sd)
        if path == "harness_resize":
            out = resize_aa(m(lr_b, res_out=hw, require_ratio=False), hw)
        elif path == "model_resize":
            out = m(lr_b, res_out=hw, require_ratio=True)
        else:
            out = m(lr_b, upscale_factor=2)
        assert tuple(out.shape[2:]) == hw
        loss = l1_loss(out, hr_b, fuse_into_model_backward=fuse)
        loss.backward()
        res[fuse] = (float(loss), _grads(m))
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * abs(res[False][0])
    worst = max((((res[True][1][k] - v).norm() / v.norm().clamp_min(1e-20)).item(), k) for k, v in res[False][1].items())
    print("fused vs materialised L1 gradient, worst relative L2:", worst)
    assert worst[0] <= 1e-3, worst          # same arithmetic; what differs is the order of the weight-gradient kernels' fp32 atomics
