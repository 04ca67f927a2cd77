"""Round-2 parity additions (VERDICT r1 "next round" item 1): every BASELINE.json configuration at its own size and
batch, the 0.01 dB PSNR target, gradients with the ReLU / clamp decisions factored out, and the reference's caller patterns.

All through the plugin surface (models/<Name>/model.py) and the C ABI; the oracle / fixtures are the checkers."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fast_transformer_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# forward tolerance: 4x what the bf16 path measures against the fp32 reference (max |diff| 9.5e-4, 74 dB)
MAX_ABS, MIN_PSNR = 4e-3, 62.0


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


@pytest.fixture(scope="module")
def model(det_sd):
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    return m.to("cuda").eval()


def check_patches(y0, d, max_abs=MAX_ABS, min_psnr=MIN_PSNR):
    worst, se, n = 0.0, 0.0, 0
    for i, (a, b) in enumerate(zip(d["ys"].tolist(), d["xs"].tolist())):
        diff = y0[:, a:a + 32, b:b + 32] - torch.from_numpy(d["patches"][i])
        worst = max(worst, diff.abs().max().item())
        se += (diff.double() ** 2).sum().item(); n += diff.numel()
    ps = 10 * np.log10(1.0 / max(se / n, 1e-20))
    assert worst <= max_abs, worst
    assert ps >= min_psnr, ps
    return worst, ps


# ---------------------------------------------------------------- config 2 at its batch -----------------------------------
def test_config2_batch8_720p(model, golden_dir):
    """BASELINE configs[1] exactly as benchmarked: 8 images of 720x1280 -> 1080x1920 in one call (1,920 windows per
    whole-block launch).  Image 0 is the fixture input: its output must match the reference patches; every image of the
    batch must equal its own single-image forward bit for bit."""
    d = dict(np.load(os.path.join(golden_dir, "fwd_720p_to_1080p.npz")))
    x0 = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234))
    xr = torch.rand((7, 3, 720, 1280), generator=torch.Generator().manual_seed(4242))
    x = torch.cat([x0, xr]).cuda()
    with torch.no_grad():
        yb = model(x, res_out=(1080, 1920))
        assert tuple(yb.shape) == (8, 3, 1080, 1920)
        worst, ps = check_patches(yb[0].cpu(), d)
        print(f"config 2 at B=8: image 0 vs reference patches max|d| {worst:.2e} PSNR {ps:.1f} dB")
        assert abs(yb[0].double().mean().item() - d["stats"][0]) < 5e-4
        # (single images are 240-window launches, which engine routes to the 16x16x32 whole-block kernel; for the bit-for-bit check the
        # B = 1 forwards run the kernel the batch ran: a window's result does not depend on how many windows the launch carries)
        from transformerupscaler_amd import engine
        keep = engine.STREAM_MIN_WINDOWS
        engine.STREAM_MIN_WINDOWS = 1
        try:
            for i in range(8):
                assert torch.equal(yb[i:i + 1], model(x[i:i + 1], res_out=(1080, 1920))), f"image {i} differs from its B=1 forward"
        finally:
            engine.STREAM_MIN_WINDOWS = keep
        y1 = model(x[:1], res_out=(1080, 1920))          # ... and the routed single-image forward (the other kernel) to bf16 noise
        assert (y1 - yb[:1]).abs().max().item() <= 2e-3


def test_config4_batch4_540p_x4(model, golden_dir):
    """BASELINE configs[3]: 4 images 540x960 -> 2160x3840 (two-stage x4, reflect-padded 544 rows)."""
    d = dict(np.load(os.path.join(golden_dir, "fwd_540p_x4.npz")))
    x0 = torch.rand((1, 3, 540, 960), generator=torch.Generator().manual_seed(1234))
    xr = torch.rand((3, 3, 540, 960), generator=torch.Generator().manual_seed(77))
    x = torch.cat([x0, xr]).cuda()
    with torch.no_grad():
        yb = model(x, upscale_factor=4)
        assert tuple(yb.shape) == (4, 3, 2160, 3840)
        worst, ps = check_patches(yb[0].cpu(), d)
        print(f"config 4 at B=4: max|d| {worst:.2e} PSNR {ps:.1f} dB")
        from transformerupscaler_amd import engine        # (B = 1 on the kernel the batch ran, as in test_config2_batch8_720p)
        keep = engine.STREAM_MIN_WINDOWS
        engine.STREAM_MIN_WINDOWS = 1
        try:
            for i in (0, 3):
                assert torch.equal(yb[i:i + 1], model(x[i:i + 1], upscale_factor=4))
        finally:
            engine.STREAM_MIN_WINDOWS = keep
        # the six blocks' launch handing patch_unembed bf16 tokens (the default) changes nothing: that GEMM rounds fp32 tokens to the
        # same bf16 values on load
        keep16 = engine.stream_bf16_tokens
        engine.stream_bf16_tokens = False
        try:
            assert torch.equal(yb, model(x, upscale_factor=4))
        finally:
            engine.stream_bf16_tokens = keep16


# ---------------------------------------------------------------- ΔPSNR <= 0.01 dB ----------------------------------------
def test_delta_psnr_vs_reference(model, golden_dir):
    """north_star: "output PSNR within 0.01 dB of the CPU reference" -- PSNR against the HR target as inference.py:129-146
    prints it, build vs the value the real reference produced (tests/golden/make_golden_r2.py)."""
    d = dict(np.load(os.path.join(golden_dir, "psnr_cases.npz")))
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 1080, 1920), generator=g)
    res = {}
    with torch.no_grad():
        res["720p"] = (psnr(model(x.cuda(), res_out=(1080, 1920)).cpu(), hr), float(d["psnr_720p"]))
        x = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(11))
        hr = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(12))
        res["256"] = (psnr(model(x.cuda(), upscale_factor=2).cpu(), hr), float(d["psnr_256"]))
        hr = torch.from_numpy(d["real_hr_u8"]).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
        y = model(torch.from_numpy(d["real_lr"]).cuda(), upscale_factor=2).cpu()
        res["real"] = (psnr(y, hr), float(d["psnr_real"]))
        ref = torch.from_numpy(d["real_y_f16"].astype(np.float32))
        assert (y - ref).abs().max().item() <= MAX_ABS and psnr(y, ref) >= MIN_PSNR
    for k, (got, want) in res.items():
        print(f"PSNR vs HR [{k}]: build {got:.5f} dB, reference {want:.5f} dB, delta {got - want:+.5f}")
        assert abs(got - want) <= 0.01, (k, got, want)


# ---------------------------------------------------------------- training at 720p ----------------------------------------
def _grad_family(name: str) -> str:
    """Parameter families whose gradient errors differ by nature (limits per family, not one global number)."""
    if name.endswith("relative_position_bias_table"):
        return "relpos_table"          # sums with near-total cancellation over windows (see the gate-matched test below)
    if name.startswith("conv1.") or name.endswith(".bias") and not name.startswith("window_blocks."):
        return "cnn_bias+conv1"        # sit behind the most ReLU / clamp gates that flip under bf16
    if name.startswith("window_blocks."):
        return "blocks"
    return "cnn_weights"


# Limits of the fixture comparison = the reference's own gradients (fp32) against this path's (bf16 operands, gates that flip):
# (sampled max error / max |g|, norm error, full relative L2), per family.  Set from the values the 720p and crop fixtures measure on
# the MI355X (printed by the test) times 1.3, rounded up; the calibration file (tests/golden/calib_bf16_autocast.json) puts the
# reference's OWN bf16-autocast run at 6.6-7.9 % median / 9.4-24 % worst relative L2 on the same weights, i.e. above every limit here.
GRAD_LIMITS = {"relpos_table": (0.11, 0.03, 0.11), "cnn_bias+conv1": (0.11, 0.03, 0.11), "blocks": (0.08, 0.02, 0.08), "cnn_weights": (0.08, 0.02, 0.08)}


def _check_grads_vs_fixture(model, d):
    none = set(d["none_grads"].tolist()) if "none_grads" in d else set()
    worst = {}
    for k, p in model.named_parameters():
        if k in none:
            assert p.grad is None, f"{k} must not receive a gradient"
            continue
        assert p.grad is not None, k
        st = d["gstat_" + k]
        g = p.grad.detach().double().cpu().flatten()
        e_s = np.abs(g[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() / max(st[2], 1e-12)
        e_n = abs(g.norm().item() - st[1]) / max(st[1], 1e-12)
        e_f = 0.0
        if "gfull_" + k in d:
            full = torch.from_numpy(d["gfull_" + k]).double().flatten()
            e_f = (g - full).norm().item() / max(full.norm().item(), 1e-12)
        fam = _grad_family(k)
        w = worst.setdefault(fam, [0.0, 0.0, 0.0, "", "", ""])
        for i, e in enumerate((e_s, e_n, e_f)):
            if e > w[i]:
                w[i], w[3 + i] = e, k
    for fam, w in sorted(worst.items()):
        print(f"   {fam}: worst sampled {w[0]:.4f} ({w[3]}), worst norm {w[1]:.4f} ({w[4]}), worst full rel-L2 {w[2]:.4f} ({w[5]})")
    for fam, w in worst.items():
        ls, ln, lf = GRAD_LIMITS[fam]
        assert w[0] <= ls, f"{fam} / {w[3]}: sampled max err {w[0]:.4f} of max|g| (limit {ls})"
        assert w[1] <= ln, f"{fam} / {w[4]}: norm err {w[1]:.4f} (limit {ln})"
        assert w[2] <= lf, f"{fam} / {w[5]}: relative L2 error {w[2]:.4f} (limit {lf})"


def test_train_720p_grads_match_reference(det_sd, golden_dir):
    """BASELINE configs[2] geometry (one rank's sample): train.py:113-140 step at 720x1280 -> 1440x2560 -> Resize 1080x1920,
    L1, dropout off -- every parameter gradient against the real reference's (tests/golden/train_720p.npz)."""
    from transformerupscaler_amd.autograd import resize_aa
    d = dict(np.load(os.path.join(golden_dir, "train_720p.npz")))
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(int(d["seed"]))
    lr = torch.rand((1, 3, 720, 1280), generator=g).cuda()
    hr = torch.rand((1, 3, 1080, 1920), generator=g).cuda()
    out = m(lr, res_out=(1080, 1920), require_ratio=False)
    assert tuple(out.shape[2:]) == (1440, 2560)
    loss = F.l1_loss(resize_aa(out, (1080, 1920)), hr)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 1e-3, (loss.item(), float(d["loss"]))
    _check_grads_vs_fixture(m, d)


# ---------------------------------------------------------------- ResidualTransformer x6 (config 5) -----------------------
@pytest.fixture(scope="module")
def rt_model():
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    return m.cuda().eval()


def test_rt_x6_forward_matches_reference(rt_model, golden_dir):
    """BASELINE configs[4] geometry: 720x1280 -> 4320x7680, 28 reference patches incl. the four corners."""
    d = dict(np.load(os.path.join(golden_dir, "rt_fwd_x6.npz")))
    x = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = rt_model(x.cuda(), upscale_factor=6)
    assert tuple(y.shape) == (1, 3, 4320, 7680)
    worst, ps = check_patches(y[0].cpu(), d)
    print(f"RT x6: max|d| {worst:.2e} PSNR {ps:.1f} dB")
    assert abs(y.double().mean().item() - d["stats"][0]) < 5e-4
    assert np.abs(y[0].double().mean(dim=(0, 2)).float().cpu().numpy() - d["row_means"]).max() < 2e-3
    assert np.abs(y[0].double().mean(dim=(0, 1)).float().cpu().numpy() - d["col_means"]).max() < 2e-3
    with torch.no_grad():                      # batch 2 (the per-GPU batch of config 5) == per-sample
        x2 = torch.cat([x, torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(5))]).cuda()
        y2 = rt_model(x2, upscale_factor=6)
        assert torch.equal(y2[0:1], y)


def test_rt_x6_train_grads_match_reference(golden_dir):
    from transformerupscaler_amd.autograd import l1_loss
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    d = dict(np.load(os.path.join(golden_dir, "rt_train_x6.npz")))
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(9876)
    lr = torch.rand((1, 3, 720, 1280), generator=g).cuda()
    hr = torch.rand((1, 3, 4320, 7680), generator=g).cuda()
    loss = l1_loss(m(lr, upscale_factor=6), hr)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 1e-3, (loss.item(), float(d["loss"]))
    _check_grads_vs_fixture(m, d)


# ---------------------------------------------------------------- gradients with the gates factored out -------------------
@pytest.mark.parametrize("scale,shape,kw", [(2, (2, 3, 36, 44), dict(res_out=(54, 66))), (4, (1, 3, 20, 28), dict(upscale_factor=4)),
                                            (3, (2, 3, 24, 40), dict(res_out=(70, 100))), (6, (1, 3, 16, 24), dict(upscale_factor=6)),
                                            # the benchmarked size (BASELINE configs[2], one sample): 14,400 tokens, 240 windows
                                            (2, (1, 3, 720, 1280), dict(res_out=(1080, 1920)))])
def test_grads_mask_matched_vs_oracle(det_sd, scale, shape, kw):
    """Backward kernels alone: the oracle's autograd is evaluated at the ReLU / clamp gates the HIP forward actually took
    (oracle.forward(masks=...)), so a gate flipped by bf16 rounding -- which dominates the plain comparison -- no longer
    counts and the limits can sit at kernel-arithmetic level."""
    from transformerupscaler_amd import autograd as AG, engine
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    x = torch.rand(shape, generator=torch.Generator().manual_seed(scale))
    res_out, s = engine.resolve_scale(shape[2], shape[3], kw.get("res_out", (0, 0)), kw.get("upscale_factor"))
    assert s == scale
    pk, ft, fn = m.packed(scale, backward=True)
    out, sv = AG.forward_train(pk, ft, x.cuda(), scale, res_out, True)
    R = torch.rand(tuple(out.shape), generator=torch.Generator().manual_seed(99)) - 0.5
    grads = AG.backward_train(pk, ft, fn, sv, scale, R.cuda())
    nchw = lambda t: t.float().permute(0, 3, 1, 2).cpu()
    masks = {"feat1": nchw(sv["feat1"]) > 0, "feat": nchw(sv["feat"]) > 0, "dec": nchw(sv["dec"]) > 0,
             "upscaled_input": sv["ui"].cpu() > 0, "clamp": (sv["pre"].cpu() >= 0) & (sv["pre"].cpu() <= 1)}
    leaf = {k: v.clone().requires_grad_(True) for k, v in det_sd.items()}
    yo = O.forward(leaf, x, masks=masks, **kw)
    (yo * R).sum().backward()
    errs = {}
    for k, v in leaf.items():
        if v.grad is None:
            assert k not in grads or grads[k] is None, k
            continue
        g, ref = grads[k].detach().cpu().double(), v.grad.double()
        errs[k] = (g.reshape(ref.shape) - ref).norm().item() / max(ref.norm().item(), 1e-12)
    med = sorted(errs.values())[len(errs) // 2]
    worst = max(errs, key=errs.get)
    print(f"scale {scale} mask-matched: relative L2 median {med:.4f} worst {errs[worst]:.4f} ({worst})")
    assert med <= 0.012, med                   # measured 0.0069-0.0076 over the five geometries
    # The relative-position tables are the one family whose gradient is a sum with almost complete cancellation (every row of
    # dS sums to zero; the table entry collects dS over all windows and all (i, j) of one offset).  The kernel accumulates dS in
    # fp32 BEFORE any rounding (attention_bwd.hip: dbacc += ds), so what shows amplified is not the reduction but the one-sweep
    # form of the softmax Jacobian: rowsum(P o dP) is taken as dO . O with the bf16 O (and the bf16-rounded P behind it) of the
    # forward, which leaves every row of dS a small non-zero sum that does not cancel across windows.  The exact form needs a
    # second sweep over the keys (or 64 more registers in a 470-register kernel): +0.1-0.2 ms per step, not taken.  At crop size the
    # tables measure 3.9-4.9 %, at 720p (240 windows) 7.0-8.2 %; everything else 0.9-1.05 % at every size.
    tables = {k: v for k, v in errs.items() if k.endswith("relative_position_bias_table")}
    rest = {k: v for k, v in errs.items() if k not in tables}
    w_rest, w_tab = max(rest, key=rest.get), max(tables, key=tables.get)
    print(f"   worst outside the relative-position tables {rest[w_rest]:.4f} ({w_rest}); worst table {tables[w_tab]:.4f} ({w_tab})")
    assert rest[w_rest] <= 0.02, (w_rest, rest[w_rest])                      # measured <= 0.0105; was 0.05
    assert tables[w_tab] <= (0.10 if shape[2] >= 720 else 0.065), (w_tab, tables[w_tab])


# ---------------------------------------------------------------- caller patterns -----------------------------------------
def test_inference_under_fp16_autocast(model, golden_dir):
    """inference.py:117-122: ``with torch.no_grad(), torch.autocast("cuda", torch.float16): model(lr, upscale_factor=s)``."""
    d = dict(np.load(os.path.join(golden_dir, "fwd_g68x84_s2_b2.npz")))
    x = torch.from_numpy(d["x"]).cuda()
    with torch.no_grad():
        y32 = model(x, upscale_factor=2)
        with torch.autocast("cuda", dtype=torch.float16):
            y16 = model(x, upscale_factor=2)
    assert y16.dtype == torch.float16 and y32.dtype == torch.float32
    assert torch.equal(y16, y32.half())              # same kernels; the module only casts the result to the autocast dtype
    ref = torch.from_numpy(d["y"])
    assert (y16.float().cpu() - ref).abs().max().item() <= MAX_ABS + 5e-4      # + fp16 rounding of [0,1]


def _per_sample_step(model, lr, hr, autocast=False):
    """train.py:113-140: zero_grad; per-sample forward (B=1) inside the autocast context; external Resize; L1; mean of
    the per-sample losses; ONE backward."""
    from transformerupscaler_amd.autograd import resize_aa
    model.zero_grad(set_to_none=True)
    losses = []
    ctx = torch.autocast("cuda", dtype=torch.float16) if autocast else torch.autocast("cuda", enabled=False)
    with ctx:
        for i in range(lr.shape[0]):
            out = model(lr[i:i + 1], res_out=tuple(hr.shape[2:]), require_ratio=False)
            if tuple(out.shape[2:]) != tuple(hr.shape[2:]):
                out = resize_aa(out.float(), tuple(hr.shape[2:]))
            losses.append(F.l1_loss(out.float(), hr[i:i + 1]))
    loss = sum(losses) / len(losses)
    loss.backward()
    return loss.detach()


def test_per_sample_loop_matches_reference_and_batched(det_sd, golden_dir):
    """The reference's own loop shape (several B=1 forwards, one backward) gives the fixture's gradients, equals the batched
    harness step, is unchanged by a world-1 DataParallel wrapper, and works under train.py's autocast wrapping."""
    from transformerupscaler_amd.dp import DataParallel
    d = dict(np.load(os.path.join(golden_dir, "train_g36x44.npz")))
    lr, hr = torch.from_numpy(d["lr"]).cuda(), torch.from_numpy(d["hr"]).cuda()
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    loss = _per_sample_step(m, lr, hr)
    assert abs(loss.item() - float(d["loss"])) < 1e-3
    _check_grads_vs_fixture(m, d)
    g_loop = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # batched step (harness semantics)
    from transformerupscaler_amd.autograd import resize_aa
    m.zero_grad(set_to_none=True)
    F.l1_loss(resize_aa(m(lr, res_out=(54, 66), require_ratio=False), (54, 66)), hr).backward()
    for k, p in m.named_parameters():
        if p.grad is None:
            assert k not in g_loop
            continue
        rel = (p.grad - g_loop[k]).norm().item() / max(g_loop[k].norm().item(), 1e-12)
        assert rel <= 2e-2, (k, rel)
    # the same loop through DataParallel (world 1: reducer episodes per sample, no collective)
    dp = DataParallel(m, scale=2)
    loss_dp = _per_sample_step(m, lr, hr)
    assert abs(loss_dp.item() - loss.item()) < 1e-6
    for k, p in m.named_parameters():
        if p.grad is not None:
            rel = (p.grad - g_loop[k]).norm().item() / max(g_loop[k].norm().item(), 1e-12)
            assert rel <= 1e-3, (k, rel)                    # same kernels; fp32 atomics leave order-dependent last bits
    dp.detach()
    # train.py:65-73,117 autocast wrapping: output comes back fp16, gradients still flow to fp32 parameters
    loss_ac = _per_sample_step(m, lr, hr, autocast=True)
    assert abs(loss_ac.item() - loss.item()) < 2e-3
    assert all(p.grad is None or p.grad.dtype == torch.float32 for p in m.parameters())


def test_dp_scale_mismatch_raises_and_mixed_scales_work(det_sd):
    """a DataParallel(scale=2) model called at another scale must not train silently without those parameters."""
    from transformerupscaler_amd.dp import DataParallel
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    x = torch.rand((1, 3, 16, 24), generator=torch.Generator().manual_seed(1)).cuda()
    dp = DataParallel(m, scale=2)
    with pytest.raises(RuntimeError, match="layout"):
        m(x, upscale_factor=3).sum().backward()
    dp.detach()
    ref = {}
    for s in (2, 3):                                        # reference: the two scales without a reducer
        m.zero_grad(set_to_none=True)
        m(x, upscale_factor=s).sum().backward()
        for k, p in m.named_parameters():
            if p.grad is not None:
                ref[k] = ref.get(k, 0) + p.grad.clone()
    dp = DataParallel(m, scales=(2, 3, 4, 6))
    m.zero_grad(set_to_none=True)
    (m(x, upscale_factor=2).sum() + m(x, upscale_factor=3).sum()).backward()     # one step, two scales (train.py:119-133)
    for k, p in m.named_parameters():
        if k in ref:
            rel = (p.grad - ref[k]).norm().item() / max(ref[k].norm().item(), 1e-12)
            assert rel <= 1e-3, (k, rel)                    # fp32 atomics in the weight-gradient kernels: order-dependent last bits
        else:
            assert p.grad is None, k                        # scales 4 / 6 stay untouched
    dp.detach()


def test_invalidate_packed_after_data_write(det_sd):
    """``p.data.copy_`` does not bump the version counter the packed-weight cache is keyed on."""
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    x = torch.rand((1, 3, 16, 24), generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        y0 = m(x, upscale_factor=2)
        m.conv1.weight.data.mul_(0.5)
        m.invalidate_packed()
        y1 = m(x, upscale_factor=2)
        assert not torch.equal(y0, y1)
        m.conv1.weight.mul_(2.0)                 # a versioned in-place write is picked up by itself
        y2 = m(x, upscale_factor=2)
        assert torch.equal(y0, y2)


def test_speed_test_cli_graph():
    """the synced speed_test.py (reference surface speed_test.py:77-88) incl. hipGraph replay runs on
    the box and reports a latency; its eager and graph runs agree on the geometry."""
    env = dict(os.environ)
    outs = {}
    for tag, extra in (("eager", []), ("graph", ["--graph"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "speed_test.py"), "--model", "FastTransformer", "--frames", "20",
                            "--warmup", "3", "--res_in", "720", "--res_out", "2160", "3840", *extra],
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
        print(tag, outs[tag]["latency_ms"], outs[tag]["images_per_sec"])
    assert outs["graph"]["graph"] is True and outs["eager"]["graph"] is False
    assert outs["graph"]["res_out"] == [2160, 3840] == outs["eager"]["res_out"]
    assert 0 < outs["graph"]["latency_ms"]["p50"] < 50.0


def test_input_gradient_matches_oracle(det_sd):
    """an input that requires grad gets d loss / d x (the reference's autograd supplies it), not None."""
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    m = m.cuda().eval()
    x = torch.rand((1, 3, 20, 28), generator=torch.Generator().manual_seed(8))
    R = torch.rand((1, 3, 40, 56), generator=torch.Generator().manual_seed(9)) - 0.5
    xo = x.clone().requires_grad_(True)
    (O.forward(det_sd, xo, upscale_factor=2) * R).sum().backward()
    xg = x.cuda().requires_grad_(True)
    (m(xg, upscale_factor=2) * R.cuda()).sum().backward()
    assert xg.grad is not None and xg.grad.shape == x.shape
    rel = (xg.grad.cpu().double() - xo.grad.double()).norm().item() / xo.grad.double().norm().item()
    print("input-gradient relative L2", rel)
    assert rel <= 0.10, rel


@pytest.mark.parametrize("name", ["FastTransformer", "ResidualTransformer", "WindowTransformer"])
def test_train_mode_without_grad_applies_dropout(name, det_sd):
    """the reference's nn.Dropout layers are active whenever the module is in .train(), with or without gradients;
    all three plugins now behave the same: a no_grad forward in .train() equals the same-seed training forward (and differs
    from .eval())."""
    from transformerupscaler_amd import weights as Wt
    m = importlib.import_module(f"models.{name}.model").TransformerModel()
    sd = {"FastTransformer": lambda: det_sd, "ResidualTransformer": lambda: Wt.rt_deterministic_state_dict(0),
          "WindowTransformer": lambda: Wt.wt_deterministic_state_dict(0)}[name]()
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    shape = (1, 3, 720, 1280) if name == "ResidualTransformer" else (1, 3, 64, 96)
    kw = dict(res_out=(1080, 1920)) if name == "ResidualTransformer" else dict(upscale_factor=2)
    x = torch.rand(shape, generator=torch.Generator().manual_seed(4)).cuda()
    m.eval()
    with torch.no_grad():
        y_eval = m(x, **kw)
    m.train()
    calls = m._dropout_calls
    with torch.no_grad():
        y_ng = m(x, **kw)
    m._dropout_calls = calls
    y_g = m(x, **kw)                                   # same seed through the autograd path
    assert torch.equal(y_ng, y_g.detach())
    assert (y_ng - y_eval).abs().max().item() > 1e-5   # dropout did something
