"""Training path on the MI355X: gradients of every parameter from the hand-written backward vs the
reference-generated fixture (tests/golden/train_g36x44.npz) and vs the oracle's autograd; Adam step
semantics (parameters of unused scales untouched).

Tolerances.  Activations / activation-gradients are bf16 with fp32 accumulation.
 * With a smooth cotangent (loss = sum(out * R), R fixed) every parameter gradient must match the
   oracle's fp32 autograd to the limits of CALIB below.  They are derived from a committed calibration, not from
   this path's own result: tests/golden/calib_bf16_autocast.json (tests/golden/make_golden_r2.py --only calib) holds how far
   the REFERENCE graph itself moves when run under torch bf16 autocast on the CPU with the same weights / inputs /
   cotangent: median over parameters 6.6-7.9 %, worst parameter 9.4-24 % relative L2, forward max |diff| 6.1e-3..6.7e-3.
   Limit: median <= 1.25 x the largest calibrated median, worst <= min(15 %, largest calibrated worst).  The error is
   dominated by ReLU / clamp gate flips caused by the bf16 forward, which every parameter gradient inherits;
   tests/test_hip_parity_r2.py::test_grads_mask_matched_vs_oracle removes the flips and holds the kernels to 2 % / 5 %.
 * With train.py's L1 loss the cotangent is sign(out - hr)/N, which is discontinuous: a forward
   difference of 1e-3 (bf16) flips the sign on the ~0.2 % of pixels with |out - hr| < 1e-3 and that
   alone is a ~4-5 % relative-L2 change of the cotangent.  The fixture comparison therefore allows
   10 % relative L2 / sampled error (the loss value itself must agree to 2e-3)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fast_transformer_oracle as O

pytestmark = pytest.mark.gpu


def _calib_limits():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "calib_bf16_autocast.json")) as f:
        c = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    med = 1.25 * max(v["grad_rel_l2_median"] for v in c.values())
    worst = min(0.15, max(v["grad_rel_l2_worst"] for v in c.values()))
    return med, worst


MED_LIMIT, WORST_LIMIT = _calib_limits()


def make_model(det_sd):
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    return m.to("cuda")


def train_loss(model, lr, hr):
    """train.py:117-136 with equal-shaped samples batched (mean of per-sample means == batch mean)."""
    from transformerupscaler_amd.autograd import resize_aa
    out = model(lr, res_out=tuple(hr.shape[2:]), require_ratio=False)
    if tuple(out.shape[2:]) != tuple(hr.shape[2:]):
        out = resize_aa(out, tuple(hr.shape[2:]))
    return F.l1_loss(out, hr)


def test_train_step_grads_match_reference(det_sd, golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "train_g36x44.npz")))
    model = make_model(det_sd).eval()            # eval graph with grads: dropout off, as the fixture
    loss = train_loss(model, torch.from_numpy(d["lr"]).cuda(), torch.from_numpy(d["hr"]).cuda())
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-3, (loss.item(), float(d["loss"]))
    none = set(d["none_grads"].tolist())
    worst = {}
    for k, p in model.named_parameters():
        if k in none:
            assert p.grad is None, f"{k} must not receive a gradient at scale 2"
            continue
        assert p.grad is not None, k
        st = d["gstat_" + k]
        g = p.grad.detach().double().cpu().flatten()
        samp = g[torch.from_numpy(d["gidx_" + k])].float().numpy()
        ref = d["gval_" + k]
        e_s = np.abs(samp - ref).max() / max(st[2], 1e-12)
        e_n = abs(g.norm().item() - st[1]) / max(st[1], 1e-12)
        worst[k] = (e_s, e_n)
        if "gfull_" + k in d:
            full = torch.from_numpy(d["gfull_" + k]).double().flatten()
            rel = (g - full).norm().item() / max(full.norm().item(), 1e-12)
            assert rel <= 0.10, f"{k}: relative L2 error {rel:.4f}"
        assert e_s <= 0.10, f"{k}: sampled max err {e_s:.4f} of max|g|"
        assert e_n <= 0.05, f"{k}: norm err {e_n:.4f}"
    print("worst sampled:", max(v[0] for v in worst.values()), "worst norm:", max(v[1] for v in worst.values()))


@pytest.mark.parametrize("scale,shape,kw", [(4, (1, 3, 20, 28), dict(upscale_factor=4)),
                                            (3, (2, 3, 24, 40), dict(res_out=(70, 100))),
                                            (6, (1, 3, 16, 24), dict(upscale_factor=6))])
def test_grads_fixed_cotangent_vs_oracle(det_sd, scale, shape, kw):
    """Scales 3/4/6 incl. the in-model resize + clamp path, against the oracle's autograd (CPU fp32),
    with a smooth loss sum(out * R) so only the backward kernels are under test."""
    x = torch.rand(shape, generator=torch.Generator().manual_seed(scale))
    leaf = {k: v.clone().requires_grad_(True) for k, v in det_sd.items()}
    yo = O.forward(leaf, x, **kw)
    R = torch.rand(tuple(yo.shape), generator=torch.Generator().manual_seed(99)) - 0.5
    (yo * R).sum().backward()
    model = make_model(det_sd).eval()
    y = model(x.cuda(), **kw)
    (y * R.cuda()).sum().backward()
    assert (y.detach().cpu() - yo.detach()).abs().max() <= 4e-3
    errs = {}
    for k, p in model.named_parameters():
        ref = leaf[k].grad
        if ref is None:
            assert p.grad is None, k
            continue
        g = p.grad.detach().cpu().double()
        errs[k] = (g - ref.double()).norm().item() / max(ref.double().norm().item(), 1e-12)
    bad = {k: round(v, 4) for k, v in errs.items() if v > WORST_LIMIT}
    med = sorted(errs.values())[len(errs) // 2]
    print(f"scale {scale}: relative L2 median {med:.4f} worst {max(errs.values()):.4f} ({max(errs, key=errs.get)})")
    assert not bad, bad
    assert med <= MED_LIMIT, (med, MED_LIMIT)


def test_grads_fixed_cotangent_train_geometry(det_sd, golden_dir):
    """train.py geometry (model at x2 without ratio, external antialiased resize) with a smooth loss."""
    from transformerupscaler_amd.autograd import resize_aa
    d = dict(np.load(os.path.join(golden_dir, "train_g36x44.npz")))
    lr = torch.from_numpy(d["lr"])
    R = torch.rand((2, 3, 54, 66), generator=torch.Generator().manual_seed(5)) - 0.5
    leaf = {k: v.clone().requires_grad_(True) for k, v in det_sd.items()}
    (O.aa_resize(O.forward(leaf, lr, res_out=(54, 66), require_ratio=False), (54, 66)) * R).sum().backward()
    model = make_model(det_sd).eval()
    (resize_aa(model(lr.cuda(), res_out=(54, 66), require_ratio=False), (54, 66)) * R.cuda()).sum().backward()
    errs = {}
    for k, p in model.named_parameters():
        if leaf[k].grad is None:
            assert p.grad is None, k
            continue
        g, ref = p.grad.detach().cpu().double(), leaf[k].grad.double()
        errs[k] = (g - ref).norm().item() / max(ref.norm().item(), 1e-12)
    bad = {k: round(v, 4) for k, v in errs.items() if v > WORST_LIMIT}
    med = sorted(errs.values())[len(errs) // 2]
    print(f"train geometry: relative L2 median {med:.4f} worst {max(errs.values()):.4f} ({max(errs, key=errs.get)})")
    assert not bad, bad
    assert med <= MED_LIMIT, (med, MED_LIMIT)


def test_adam_step_skips_unused_scales(det_sd, golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "train_g36x44.npz")))
    model = make_model(det_sd).eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)          # train.py:104
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt.zero_grad()
    train_loss(model, torch.from_numpy(d["lr"]).cuda(), torch.from_numpy(d["hr"]).cuda()).backward()
    opt.step()
    none = set(d["none_grads"].tolist())
    for k, p in model.named_parameters():
        if k in none:
            assert torch.equal(p.detach(), before[k]), k
        elif "adam_" + k in d:
            # first Adam step moves every element by ~lr*sign(g): compare the update direction where the
            # reference gradient is significant (e.g. the key bias of qkv has an exactly-zero true gradient)
            ref = torch.from_numpy(d["adam_" + k]).cuda()
            gref = torch.from_numpy(d["gfull_" + k]).cuda()
            sel = gref.abs() > 0.05 * gref.abs().max()
            agree = ((p.detach() - before[k]).sign() == (ref - before[k]).sign())[sel].float().mean().item()
            assert agree >= 0.97, f"{k}: only {agree:.3f} of the significant update signs agree"
    # second forward uses the updated weights (pack cache invalidated by the in-place update)
    l2 = train_loss(model, torch.from_numpy(d["lr"]).cuda(), torch.from_numpy(d["hr"]).cuda()).item()
    assert l2 < float(d["loss"]) + 1e-3


@pytest.mark.parametrize("shape", [(2, 3, 54, 66), (1, 3, 5, 7), (3,)])       # the last two: numel % 4 != 0, < one vector
def test_l1_loss_kernels_match_torch(shape):
    from transformerupscaler_amd.autograd import l1_loss
    g = torch.Generator().manual_seed(3)
    a = torch.rand(shape, generator=g).cuda().requires_grad_(True)
    b = torch.rand(shape, generator=g).cuda()
    b.view(-1)[:2] = a.detach().view(-1)[:2]                 # exact ties: sign(0) = 0
    b.view(-1)[-1] = a.detach().view(-1)[-1]
    loss = l1_loss(a, b)
    (loss * 3.0).backward()
    a2 = a.detach().clone().requires_grad_(True)
    ref = F.l1_loss(a2, b)
    (ref * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 1e-6
    assert torch.equal(a.grad, a2.grad)
    with pytest.raises(ValueError):
        l1_loss(a, b.reshape(1, -1))


@pytest.mark.parametrize("scale", [2, 3, 4, 6])
def test_pack_plan_equals_torch_pack(det_sd, scale):
    """pack_plan.PackPlan (two tup_pack_gather launches, maps traced from packing.py) == packing.pack_state_dict(backward=True),
    bit for bit, on the loaded weights and again after an in-place update of every parameter (what Adam does, train.py:139);
    and the module's training path uses it."""
    from transformerupscaler_amd import packing
    from transformerupscaler_amd.pack_plan import PackPlan
    model = make_model(det_sd)
    params = list(model.named_parameters())
    plan = PackPlan(params, lambda d: packing.pack_state_dict(d, scale, backward=True))
    g = torch.Generator(device="cuda").manual_seed(scale)
    for rnd in range(2):
        ref = packing.pack_state_dict(dict(params), scale, backward=True)
        got = plan.run(params)
        assert set(ref) == set(got)
        for k in ref:
            assert ref[k].dtype == got[k].dtype and ref[k].shape == got[k].shape, k
            assert torch.equal(ref[k], got[k]), k
            assert got[k].data_ptr() % 16 == 0, k                  # the kernels read them with 16-byte loads
        with torch.no_grad():
            for _, p in params:
                p.add_(torch.randn(p.shape, device="cuda", generator=g) * 0.01)
    pk, _, _ = model.packed(scale, backward=True)
    from transformerupscaler_amd import pack_plan
    assert any(isinstance(v, PackPlan) and k[1] == scale for k, v in pack_plan._PLANS.items())      # the module's path uses a plan
    ref = packing.pack_state_dict(dict(params), scale, backward=True)
    assert all(torch.equal(pk[k], ref[k]) for k in ref)


def test_fused_adam_equals_torch_adam():
    """optim.Adam (one tup_adam_step launch) against torch.optim.Adam on the same parameters and gradients: three steps with a
    parameter that has no gradient in step 2 (skipped: its step count and moments must not move), two parameter groups
    with different learning rates, and the state_dict of one loading into the other."""
    from transformerupscaler_amd.optim import Adam
    g = torch.Generator(device="cuda").manual_seed(7)
    shapes = [(64, 3, 3, 3), (192,), (5000,), (768, 192), (1,)]
    base = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    pa = [torch.nn.Parameter(b.clone()) for b in base]
    pb = [torch.nn.Parameter(b.clone()) for b in base]
    oa = Adam([{"params": pa[:3], "lr": 1e-3}, {"params": pa[3:], "lr": 3e-4}])
    ob = torch.optim.Adam([{"params": pb[:3], "lr": 1e-3}, {"params": pb[3:], "lr": 3e-4}])
    for step in range(3):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if step == 1 and i == 2:
                x.grad = y.grad = None
                continue
            gr = torch.randn(x.shape, device="cuda", generator=g)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), (step, i, (x - y).abs().max().item())
            sa, sb = oa.state[x], ob.state[y]
            assert float(sa["step"]) == float(sb["step"])
            assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=1e-8)
            assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-10)
    assert float(oa.state[pa[2]]["step"]) == 2.0
    # interchange: torch's state into the fused optimizer and back
    oa2 = Adam([{"params": pa[:3], "lr": 1e-3}, {"params": pa[3:], "lr": 3e-4}])
    oa2.load_state_dict(ob.state_dict())
    ob2 = torch.optim.Adam([{"params": pb[:3], "lr": 1e-3}, {"params": pb[3:], "lr": 3e-4}])
    ob2.load_state_dict(oa.state_dict())
    for x, y in zip(pa, pb):
        gr = torch.randn(x.shape, device="cuda", generator=g)
        x.grad, y.grad = gr.clone(), gr.clone()
    oa2.step(); ob2.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=4e-6, atol=2e-7)
    # options the kernel does not implement run torch's own step
    pc = [torch.nn.Parameter(base[0].clone())]
    oc = Adam(pc, lr=1e-3, weight_decay=0.1)
    pc[0].grad = torch.ones_like(pc[0])
    oc.step()
    pd = [torch.nn.Parameter(base[0].clone())]
    od = torch.optim.Adam(pd, lr=1e-3, weight_decay=0.1)
    pd[0].grad = torch.ones_like(pd[0])
    od.step()
    assert torch.equal(pc[0], pd[0])
