"""The CPU oracle (oracle/) must reproduce the fixtures generated from the real reference
module (tests/golden/make_golden.py).  fp32, tolerance 2e-5 abs (same aten ops, different
call graph only)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fast_transformer_oracle as O

TOL = 2e-5


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def _kwargs(d):
    kw = {}
    if int(d["upscale_factor"]) > 0:
        kw["upscale_factor"] = int(d["upscale_factor"])
    else:
        kw["res_out"] = tuple(int(v) for v in d["res_out"])
    kw["require_ratio"] = bool(d["require_ratio"])
    return kw


FWD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fwd_g*.npz")))


@pytest.mark.parametrize("name", FWD)
def test_forward_cases(golden_dir, det_sd, name):
    d = _load(golden_dir, name)
    cap = {}
    with torch.no_grad():
        y = O.forward(det_sd, torch.from_numpy(d["x"]), capture=cap, **_kwargs(d))
    assert tuple(y.shape) == d["y"].shape
    assert np.abs(y.numpy() - d["y"]).max() <= TOL
    # hooked reference intermediates (module outputs; conv outputs are pre-ReLU there)
    pairs = {"cap_patch_embed": cap["tokens"].permute(0, 3, 1, 2), "cap_window_blocks.0": cap["block0"],
             "cap_window_blocks.5": cap["block5"], "cap_decoder_conv2": cap["residual"],
             "cap_up1": cap["up1"], "cap_up1_conv": cap["upscaled_input"],
             "cap_final_upscale_conv": cap["residual_up"]}
    for k, v in pairs.items():
        if k in d:
            assert np.abs(v.numpy() - d[k]).max() <= 1e-4, k
    if "cap_conv2" in d:   # reference hook cloned the pre-ReLU conv output
        assert np.abs(cap["feat"].numpy() - np.maximum(d["cap_conv2"], 0)).max() <= TOL


def test_relative_position_index(golden_dir):
    d = _load(golden_dir, "relative_position_index.npz")
    idx = O.relative_position_index(8).numpy()
    assert (idx == d["index"]).all()
    assert idx[0, 0] == 112 and idx.min() == 0 and idx.max() == 224


def test_unbuilt_scale_raises(det_sd):
    with pytest.raises(ValueError):
        O.forward(det_sd, torch.rand(1, 3, 16, 16), res_out=(80, 80))


def test_256_config1(golden_dir, det_sd):
    d = _load(golden_dir, "fwd_256_s2.npz")
    x = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(int(d["seed"])))
    with torch.no_grad():
        y = O.forward(det_sd, x, upscale_factor=2)
    assert np.abs(y.numpy() - d["y_f16"].astype(np.float32)).max() <= 1e-3   # fp16 storage
    assert abs(y.double().mean().item() - d["stats"][0]) < 1e-6


def test_train_step_grads(golden_dir, det_sd):
    d = _load(golden_dir, "train_g36x44.npz")
    loss, grads = O.train_step_grads(det_sd, torch.from_numpy(d["lr"]), torch.from_numpy(d["hr"]))
    assert abs(loss.item() - float(d["loss"])) < 1e-6
    none = set(d["none_grads"].tolist())
    for k, g in grads.items():
        if k.endswith("relative_position_index"):
            continue
        if k in none:
            assert g is None, k
            continue
        st = d["gstat_" + k]
        gd = g.double().flatten()
        assert abs(gd.norm().item() - st[1]) <= 1e-4 * max(1.0, st[1]), k
        assert np.abs(gd[torch.from_numpy(d["gidx_" + k])].float().numpy() - d["gval_" + k]).max() <= 1e-5 + 1e-4 * st[2], k
    # Adam, train.py:104: default betas/eps, params without grad untouched
    for k in ("conv1.bias", "window_blocks.3.norm2.weight", "decoder_conv2.bias"):
        p = det_sd[k]
        newp, _, _ = O.adam_step(p, grads[k], torch.zeros_like(p), torch.zeros_like(p), 1)
        assert np.abs(newp.numpy() - d["adam_" + k]).max() <= 2e-7, k


def test_aa_taps_match_aten():
    """Explicit tap table == aten's antialiased bilinear (what the GPU kernel consumes)."""
    for (h, w), (oh, ow) in (((72, 96), (54, 72)), ((64, 64), (48, 48)), ((48, 80), (35, 50)),
                             ((30, 40), (45, 47)), ((288, 512), (216, 384))):
        x = torch.rand(1, 2, h, w, generator=torch.Generator().manual_seed(h))
        a = O.aa_resize(x, (oh, ow))
        b = O.aa_resize_explicit(x, (oh, ow))
        assert (a - b).abs().max().item() <= 1e-6
    xmin, xs, wts = O.aa_bilinear_taps(1440, 1080)
    assert np.allclose(wts[1, :2], [0.5, 0.5]) and np.allclose(wts[2, :3], [3 / 11, 7 / 11, 1 / 11], atol=1e-7)
    assert np.allclose(wts[0, :2], [0.7, 0.3], atol=1e-7)


@pytest.mark.parametrize("name", ["fwd_720p_to_1080p.npz", "fwd_540p_x4.npz"])
def test_full_size_statistics(golden_dir, det_sd, name):
    """BASELINE.json config 2/4 geometry on CPU (a few seconds each)."""
    d = _load(golden_dir, name)
    if "720p" in name:
        shape, kw = (1, 3, 720, 1280), dict(res_out=(1080, 1920))
    else:
        shape, kw = (1, 3, 540, 960), dict(upscale_factor=4)
    x = torch.rand(shape, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = O.forward(det_sd, x, **kw)
    for i, (a, b) in enumerate(zip(d["ys"].tolist(), d["xs"].tolist())):
        assert np.abs(y[0, :, a:a + 32, b:b + 32].numpy() - d["patches"][i]).max() <= 5e-5
    assert abs(y.double().mean().item() - d["stats"][0]) < 1e-6
    assert np.abs(y[0].double().mean(dim=(0, 2)).float().numpy() - d["row_means"]).max() < 1e-5
