"""The ResidualTransformer CPU oracle (oracle/residual_transformer_oracle.py) against fixtures generated from the real
reference module (tests/golden/make_golden_rt.py), plus the plugin surface / packing checks that need no GPU."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import residual_transformer_oracle as R


@pytest.fixture(scope="module")
def rt_sd():
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    return rt_deterministic_state_dict(0)


def test_rt_oracle_matches_reference_fixture(golden_dir, rt_sd):
    d = dict(np.load(os.path.join(golden_dir, "rt_fwd_1080p.npz")))
    x = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234))
    cap = {}
    with torch.no_grad():
        y = R.forward(rt_sd, x, res_out=(1080, 1920), capture=cap)
    for i, (a, b) in enumerate(zip(d["ys"].tolist(), d["xs"].tolist())):
        assert np.abs(y[0, :, a:a + 32, b:b + 32].numpy() - d["patches"][i]).max() <= 2e-5
    assert np.abs(cap["residual"].numpy() - d["cap_decoder_conv2"]).max() <= 2e-5
    assert np.abs(cap["block0"][0, :64].numpy() - d["cap_block0_head"]).max() <= 2e-5
    assert np.abs(cap["block7"][0, -64:].numpy() - d["cap_block7_tail"]).max() <= 5e-5
    assert np.abs(cap["feat_down"][0, :, 100:116, 200:216].numpy() - d["cap_downsample_patch"]).max() <= 2e-5
    assert abs(y.double().mean().item() - d["stats"][0]) < 1e-6


def test_rt_plugin_surface(rt_sd):
    from transformerupscaler_amd.weights import rt_param_shapes
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    sd = m.state_dict()
    assert list(sd.keys()) == list(rt_param_shapes().keys())
    assert all(tuple(sd[k].shape) == s for k, s in rt_param_shapes().items())
    assert sum(p.numel() for p in m.parameters()) == 3210051
    m.load_state_dict(rt_sd)           # strict
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 3, 720, 1280))            # CPU tensor: no silent fallback


def test_strided_conv_space_to_depth_packing():
    """pack_conv_c64_stride2: 3x3 conv over the space-to-depth input == Conv2d(stride=2, padding=1)."""
    from transformerupscaler_amd import packing
    from transformerupscaler_amd.packing import _PERM64
    g = torch.Generator().manual_seed(3)
    x = torch.rand((1, 64, 12, 16), generator=g)
    w, b = torch.rand((64, 64, 3, 3), generator=g) - 0.5, torch.rand((64,), generator=g)
    wp, bp = packing.pack_conv_c64_stride2(w, b)
    inv = torch.empty(64, dtype=torch.long); inv[_PERM64] = torch.arange(64)
    s2d = F.pixel_unshuffle(x, 2).view(1, 64, 4, 6, 8).permute(0, 2, 1, 3, 4).reshape(1, 256, 6, 8)     # [chunk sp][c]
    wk = wp[0].float()[:, :, inv]                                # [chunk][tap][cout][cin], rows back in natural order
    wfull = wk.permute(2, 0, 3, 1).reshape(64, 256, 3, 3)        # cout, (chunk, cin), tap
    got = F.conv2d(s2d, wfull, bp[0], padding=1)
    ref = F.conv2d(x, w.to(torch.bfloat16).float(), b, stride=2, padding=1)
    assert (got - ref).abs().max() < 1e-4


def test_bicubic_taps_match_aten():
    from transformerupscaler_amd.resize_taps import bicubic_taps
    for i, o in ((360, 1080), (720, 1080), (640, 3840), (45, 100)):
        idx, w = bicubic_taps(i, o)
        x = torch.rand(1, 1, i, 3, generator=torch.Generator().manual_seed(i))
        ref = F.interpolate(x, size=(o, 3), mode="bicubic", align_corners=False)[0, 0, :, 0]
        mine = (torch.from_numpy(w) * x[0, 0, :, 0][torch.from_numpy(idx).long()]).sum(1)
        assert (ref - mine).abs().max().item() <= 2e-5
        idx2, w2 = R.bicubic_taps(i, o)
        assert (idx == idx2).all() and np.abs(w - w2).max() == 0


def test_rt_oracle_backward_matches_reference_fixture(golden_dir, rt_sd):
    """Autograd through the oracle == autograd through the reference module (fixture rt_train_1080p.npz)."""
    d = dict(np.load(os.path.join(golden_dir, "rt_train_1080p.npz")))
    g = torch.Generator().manual_seed(4321)
    lr = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 1080, 1920), generator=g)
    leaf = {k: v.clone().requires_grad_(True) for k, v in rt_sd.items()}
    loss = F.l1_loss(R.forward(leaf, lr, res_out=(1080, 1920)), hr)
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 1e-6
    for k, v in leaf.items():
        gr = v.grad.double().flatten()
        st = d["gstat_" + k]
        assert abs(gr.norm().item() - st[1]) <= 2e-3 * st[1] + 1e-12, k
        assert np.abs(gr[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() <= 2e-3 * st[2] + 1e-9, k
