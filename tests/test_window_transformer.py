"""WindowTransformer plugin: the CPU oracle against fixtures generated from the real reference
module (tests/golden/make_golden_wt.py), the plugin surface, and the HIP path against the same fixtures (GPU)."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import window_transformer_oracle as WO
from transformerupscaler_amd.weights import wt_deterministic_state_dict, wt_param_shapes

CASES = [("g88x120_x2", dict(upscale_factor=2)), ("g128x128_res", dict(res_out=(200, 168))),
         ("g91x125_res", dict(res_out=(190, 260)))]          # odd input size (models/WindowTransformer/model.py:205 accepts it)


@pytest.mark.parametrize("tag,kw", CASES)
def test_oracle_matches_reference_fixture(golden_dir, tag, kw):
    d = dict(np.load(os.path.join(golden_dir, f"wt_fwd_{tag}.npz")))
    cap = {}
    with torch.no_grad():
        y = WO.forward(wt_deterministic_state_dict(0), torch.from_numpy(d["x"]), capture=cap, **kw)
    assert np.abs(y[0, :, :24, :24].numpy() - d["out_f32_patch"]).max() <= 2e-5
    assert np.abs(y.numpy() - d["out"].astype(np.float32)).max() <= 1e-3          # fp16 storage
    assert np.abs(cap["residual"].numpy() - d["residual"]).max() <= 2e-5
    assert np.abs(cap["block0"][:2].numpy() - d["block0_head"]).max() <= 2e-5


def test_plugin_surface():
    m = importlib.import_module("models.WindowTransformer.model").TransformerModel()
    sd = m.state_dict()
    params = [k for k in sd if not k.endswith("relative_position_index")]
    assert params == list(wt_param_shapes().keys())
    assert all(tuple(sd[k].shape) == s for k, s in wt_param_shapes().items())
    assert sum(p.numel() for p in m.parameters()) == 2763651
    idx = sd["window_blocks.0.attn.relative_position_index"]
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (64, 64) and int(idx[0, 0]) == 112
    res = m.load_state_dict(wt_deterministic_state_dict(0), strict=False)
    assert not res.unexpected_keys
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 3, 64, 64))               # CPU tensor: no silent fallback


@pytest.fixture(scope="module")
def wt_model():
    m = importlib.import_module("models.WindowTransformer.model").TransformerModel()
    m.load_state_dict(wt_deterministic_state_dict(0), strict=False)
    return m.cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", CASES)
def test_hip_matches_reference_fixture(golden_dir, wt_model, tag, kw):
    d = dict(np.load(os.path.join(golden_dir, f"wt_fwd_{tag}.npz")))
    with torch.no_grad():
        y = wt_model(torch.from_numpy(d["x"]).cuda(), **kw).float().cpu()
    ref = torch.from_numpy(d["out"].astype(np.float32))
    err = (y - ref).abs().max().item()
    psnr = 10 * np.log10(1.0 / max(((y - ref) ** 2).mean().item(), 1e-20))
    print(tag, "max abs", err, "PSNR", psnr)
    assert err <= 4e-3 and psnr >= 62.0            # 4x the measured error of the sibling plugins (printed above)
    assert np.abs(y[0, :, :24, :24].numpy() - d["out_f32_patch"]).max() <= 4e-3


@pytest.mark.gpu
def test_hip_1080p_matches_reference_fixture(golden_dir, wt_model):
    d = dict(np.load(os.path.join(golden_dir, "wt_fwd_1080p.npz")))
    x = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234)).cuda()
    with torch.no_grad():
        y = wt_model(x, res_out=(1080, 1920)).float().cpu()
    worst, se = 0.0, 0.0
    for i, (a, b) in enumerate(zip(d["ys"].tolist(), d["xs"].tolist())):
        diff = y[0, :, a:a + 32, b:b + 32].numpy() - d["patches"][i]
        worst = max(worst, np.abs(diff).max()); se += (diff ** 2).mean()
    psnr = 10 * np.log10(1.0 / max(se / 16, 1e-20))
    print("1080p max abs", worst, "PSNR", psnr)
    assert worst <= 4e-3 and psnr >= 62.0
    assert abs(y.double().mean().item() - d["stats"][0]) < 2e-3
    assert np.abs(y[0].double().mean(dim=(0, 2)).float().numpy() - d["row_means"]).max() < 5e-3


def test_oracle_backward_matches_reference_fixture(golden_dir):
    import torch.nn.functional as F
    d = dict(np.load(os.path.join(golden_dir, "wt_train_g88x120.npz")))
    leaf = {k: v.clone().requires_grad_(True) for k, v in wt_deterministic_state_dict(0).items()}
    loss = F.l1_loss(WO.forward(leaf, torch.from_numpy(d["lr"]), upscale_factor=2), torch.from_numpy(d["hr"]))
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 1e-6
    for k, v in leaf.items():
        gr = v.grad.double().flatten()
        st = d["gstat_" + k]
        assert abs(gr.norm().item() - st[1]) <= 2e-3 * st[1] + 1e-12, k
        assert np.abs(gr[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() <= 2e-3 * st[2] + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kw", [((2, 3, 88, 120), dict(upscale_factor=2)), ((1, 3, 144, 160), dict(res_out=(300, 250))),
                                      ((1, 3, 91, 125), dict(upscale_factor=2))])
def test_hip_grads_fixed_cotangent_vs_oracle(shape, kw):
    """Every parameter gradient against the oracle's autograd (CPU fp32) with a smooth cotangent; tolerances and their
    calibration as in test_hip_train.py (bf16 activations)."""
    m = importlib.import_module("models.WindowTransformer.model").TransformerModel()
    sd = wt_deterministic_state_dict(0)
    m.load_state_dict(sd, strict=False)
    m = m.cuda().eval()
    x = torch.rand(shape, generator=torch.Generator().manual_seed(7))
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = WO.forward(leaf, x, **kw)
    cot = torch.rand(ref.shape, generator=torch.Generator().manual_seed(8)) * 2 - 1
    (ref * cot).sum().backward()
    out = m(x.cuda(), **kw)
    assert (out.detach().cpu() - ref.detach()).abs().max() < 2e-2
    (out * cot.cuda()).sum().backward()
    rels = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        gr, g = leaf[k].grad.double().flatten(), p.grad.double().cpu().flatten()
        rels[k] = ((g - gr).norm() / gr.norm().clamp_min(1e-12)).item()
    worst = max(rels, key=rels.get)
    med = float(np.median(list(rels.values())))
    print("worst", worst, rels[worst], "median", med)
    assert rels[worst] <= 0.15, (worst, rels[worst])
    assert med <= 0.10, med


@pytest.mark.gpu
def test_hip_train_grads_match_reference_fixture(golden_dir):
    import torch.nn.functional as F
    d = dict(np.load(os.path.join(golden_dir, "wt_train_g88x120.npz")))
    m = importlib.import_module("models.WindowTransformer.model").TransformerModel()
    m.load_state_dict(wt_deterministic_state_dict(0), strict=False)
    m = m.cuda().eval()
    loss = F.l1_loss(m(torch.from_numpy(d["lr"]).cuda(), upscale_factor=2), torch.from_numpy(d["hr"]).cuda())
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 2e-3
    for k, p in m.named_parameters():
        st = d["gstat_" + k]
        gr = p.grad.detach().double().cpu().flatten()
        e_s = np.abs(gr[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() / max(st[2], 1e-12)
        e_n = abs(gr.norm().item() - st[1]) / max(st[1], 1e-12)
        assert e_s <= 0.12, f"{k}: sampled max err {e_s:.4f} of max|g|"
        assert e_n <= 0.06, f"{k}: norm err {e_n:.4f}"


@pytest.mark.gpu
def test_hip_train_mode_dropout_is_reproducible():
    m = importlib.import_module("models.WindowTransformer.model").TransformerModel(dropout=0.1)
    m.load_state_dict(wt_deterministic_state_dict(0), strict=False)
    m = m.cuda().train()
    x = torch.rand((1, 3, 88, 120), generator=torch.Generator().manual_seed(9)).cuda()
    calls = m._dropout_calls
    out = m(x, upscale_factor=2)
    out.sum().backward()
    m._dropout_calls = calls
    m.zero_grad()
    out2 = m(x, upscale_factor=2)
    assert torch.equal(out.detach(), out2.detach())
    m.eval()
    with torch.no_grad():
        assert (m(x, upscale_factor=2) - out.detach()).abs().max() > 1e-5
