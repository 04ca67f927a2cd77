"""WindowTransformer plugin: the CPU oracle against fixtures generated from the real reference
module (tests/golden/make_golden_wt.py), the plugin surface, and the HIP path against the same fixtures (GPU)."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import window_transformer_oracle as WO
from transformerupscaler_amd.weights import wt_deterministic_state_dict, wt_param_shapes

CASES = [("g88x120_x2", dict(upscale_factor=2)), ("g128x128_res", dict(res_out=(200, 168)))]


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
    assert err <= 2.5e-2 and psnr >= 50.0
    assert np.abs(y[0, :, :24, :24].numpy() - d["out_f32_patch"]).max() <= 2.5e-2


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
    assert worst <= 2.5e-2 and psnr >= 50.0
    assert abs(y.double().mean().item() - d["stats"][0]) < 2e-3
    assert np.abs(y[0].double().mean(dim=(0, 2)).float().numpy() - d["row_means"]).max() < 5e-3
