"""ResidualTransformer training path on the MI355X: kernel-level backward checks against torch autograd and the
gradients of every parameter against the oracle's autograd (CPU fp32) on reduced token grids.  Tolerances as in
test_hip_train.py (bf16 activations / activation gradients; calibration there)."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import residual_transformer_oracle as R
from transformerupscaler_amd.weights import rt_deterministic_state_dict

pytestmark = pytest.mark.gpu


def bf(t):
    return t.to(torch.bfloat16).float()


def rnd(shape, seed, scale=1.0, shift=0.0):
    return (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * scale + shift


def small_model(th, tw, seed=0):
    """The plugin with a (th x tw)-token pos_embed (the reference fixes 45 x 80; the kernels take any grid)."""
    sd = rt_deterministic_state_dict(seed)
    sd["pos_embed"] = sd["pos_embed"][:, : th * tw].clone()
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.pos_embed = nn.Parameter(torch.empty(1, th * tw, 128))
    m.num_tokens = th * tw
    m.load_state_dict(sd)
    return m.to("cuda"), sd


def test_bicubic_backward():
    from transformerupscaler_amd import ops
    a = rnd((2, 3, 20, 28), 1, 0.5, 0.5)
    b = rnd((2, 3, 10, 14), 2, 0.3).requires_grad_(True)
    size = (47, 66)
    pre = F.interpolate(a, size=size, mode="bicubic", align_corners=False) + F.interpolate(b, size=size, mode="bicubic", align_corners=False)
    out = pre.clamp(0, 1)
    gout = rnd(out.shape, 3)
    out.backward(gout)
    got_out = ops.rt_bicubic_sum(a.cuda(), b.detach().cuda(), size, clamp=True)
    gb = ops.rt_bicubic_bwd(gout.cuda(), got_out, (10, 14)).cpu()
    assert (gb - b.grad).abs().max() <= 2e-4 * max(1.0, b.grad.abs().max().item())


def test_strided_conv_backward():
    from transformerupscaler_amd import ops, packing
    x = bf(rnd((2, 64, 36, 80), 1)).requires_grad_(True)
    w = bf(rnd((64, 64, 3, 3), 2, 0.06)).requires_grad_(True)
    b = rnd((64,), 3, 0.2).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=2, padding=1)
    g = bf(rnd(y.shape, 4))
    y.backward(g)
    xg = x.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    gg = g.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    dwp, db = ops.conv_c64_wgrad_s2d(xg, gg, 2)
    dw = packing.unpack_conv_c64_stride2_wgrad(dwp).cpu()
    assert (dw - w.grad).abs().max() <= 2e-2 + 1e-2 * w.grad.abs().max()
    assert (db.cpu() - b.grad).abs().max() <= 2e-2 + 1e-3 * b.grad.abs().max()
    gx = ops.conv_c64(gg, packing.pack_conv_c64_stride2_dgrad(w.detach()).cuda(), None, 2).float().cpu().permute(0, 3, 1, 2)
    assert (gx - x.grad).abs().max() <= 1.5e-2 + 1e-2 * x.grad.abs().max()


@pytest.mark.parametrize("shape,res_out", [((1, 3, 64, 96), (96, 144)), ((2, 3, 160, 256), (320, 512))])
def test_grads_fixed_cotangent_vs_oracle(shape, res_out):
    B, _, H, W = shape
    th, tw = H // 16, W // 16
    model, sd = small_model(th, tw)
    model.eval()                                           # dropout off: the oracle's graph
    x = torch.rand(shape, generator=torch.Generator().manual_seed(7))
    cot = rnd((B, 3) + res_out, 8)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = R.forward(leaf, x, res_out=res_out)
    (ref * cot).sum().backward()
    out = model(x.cuda(), res_out=res_out)
    assert (out.detach().cpu() - ref.detach()).abs().max() < 2e-2
    (out * cot.cuda()).sum().backward()
    rels = {}
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        gr, g = leaf[k].grad.double().flatten(), p.grad.double().cpu().flatten()
        rels[k] = ((g - gr).norm() / gr.norm().clamp_min(1e-12)).item()
    worst = max(rels, key=rels.get)
    med = float(np.median(list(rels.values())))
    print("worst", worst, rels[worst], "median", med)
    assert rels[worst] <= 0.15, (worst, rels[worst])
    assert med <= 0.10, med


def test_train_mode_dropout_step():
    """.train(): dropout masks are a pure function of (seed, site, index), so the forward and backward of one call agree:
    the directional derivative along the parameter gradient matches a finite difference of the same-seed loss."""
    model, sd = small_model(4, 6)
    model.train()
    x = torch.rand((1, 3, 64, 96), generator=torch.Generator().manual_seed(9)).cuda()
    cot = rnd((1, 3, 96, 144), 10).cuda()
    calls = model._dropout_calls
    out = model(x, res_out=(96, 144))
    (out * cot).sum().backward()
    model.eval()
    with torch.no_grad():
        out_eval = model(x, res_out=(96, 144))
    assert (out.detach() - out_eval).abs().max() > 1e-4                         # dropout did something
    # same seed again -> identical output
    model.train()
    model._dropout_calls = calls
    model.zero_grad()
    out2 = model(x, res_out=(96, 144))
    assert torch.equal(out2.detach(), out.detach())
    for p in model.parameters():
        assert p.grad is None or torch.isfinite(p.grad).all()


def test_full_size_train_grads_match_reference_fixture(golden_dir):
    """720x1280 -> 1080x1920, L1 loss, eval graph: every parameter gradient against the fixture generated from the
    real reference module (tests/golden/make_golden_rt.py).  L1's sign cotangent: tolerance as in test_hip_train.py."""
    import os
    d = dict(np.load(os.path.join(golden_dir, "rt_train_1080p.npz")))
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(4321)
    lr = torch.rand((1, 3, 720, 1280), generator=g).cuda()
    hr = torch.rand((1, 3, 1080, 1920), generator=g).cuda()
    loss = F.l1_loss(m(lr, res_out=(1080, 1920)), hr)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-3
    worst_s = worst_n = 0.0
    for k, p in m.named_parameters():
        st = d["gstat_" + k]
        gr = p.grad.detach().double().cpu().flatten()
        e_s = np.abs(gr[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() / max(st[2], 1e-12)
        e_n = abs(gr.norm().item() - st[1]) / max(st[1], 1e-12)
        worst_s, worst_n = max(worst_s, e_s), max(worst_n, e_n)
        if "gfull_" + k in d:
            full = torch.from_numpy(d["gfull_" + k]).double().flatten()
            rel = (gr - full).norm().item() / max(full.norm().item(), 1e-12)
            assert rel <= 0.10, f"{k}: relative L2 error {rel:.4f}"
        assert e_s <= 0.10, f"{k}: sampled max err {e_s:.4f} of max|g|"
        assert e_n <= 0.05, f"{k}: norm err {e_n:.4f}"
    print("worst sampled", worst_s, "worst norm", worst_n)


def test_l1_gradient_fused_into_model_backward():
    """autograd.l1_loss(..., fuse_into_model_backward=True): the loss gradient is formed inside the bicubic backward's row pass
    instead of being materialised.  Same loss and (up to the order of a few fp32 sums: none here, the row pass is identical) the
    same parameter gradients as the two-node path; a model output with a second consumer is refused loudly."""
    from transformerupscaler_amd.autograd import l1_loss
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    m = m.cuda().eval()                                   # eval: no dropout, the two runs see the same forward
    g = torch.Generator().manual_seed(31)
    lr = torch.rand((1, 3, 720, 1280), generator=g).cuda()       # the position embedding fixes the token grid (model.py:140)
    hr = torch.rand((1, 3, 1440, 2560), generator=g).cuda()
    grads = []
    for fuse in (False, True):
        m.zero_grad(set_to_none=True)
        loss = l1_loss(m(lr, upscale_factor=2), hr, fuse_into_model_backward=fuse)
        loss.backward()
        grads.append((loss.item(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert grads[0][0] == grads[1][0]
    assert set(grads[0][1]) == set(grads[1][1]) and len(grads[0][1]) > 50
    for k in grads[0][1]:
        a, b = grads[0][1][k], grads[1][1][k]
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-9 + 1e-6 * a.abs().max().item()), k
    # a second consumer of the model output: its gradient is not the loss's stand-in
    m.zero_grad(set_to_none=True)
    out = m(lr, upscale_factor=2)
    total = l1_loss(out, hr, fuse_into_model_backward=True) + 1e-3 * out.mean()
    with pytest.raises(RuntimeError, match="consumer besides the loss"):
        total.backward()
