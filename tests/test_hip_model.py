"""End-to-end parity of the HIP path (through the nn.Module plugin surface and the C ABI) against
the oracle and the reference-generated golden fixtures.

Tolerance (bf16 activations, fp32 accumulation; BASELINE.json north_star): max |diff| <= 4e-3 and PSNR(build,
reference fp32) >= 62 dB on [0,1] images = 4x what the path measures (9.5e-4 / 74 dB; every test prints its own numbers).
For scale: the reference's own CPU bf16-autocast run differs from its fp32 run by max 6.1e-3..6.7e-3 on these weights
(tests/golden/calib_bf16_autocast.json) and 1.9e-3 / 69.7 dB on natural-statistics weights."""
import glob
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import fast_transformer_oracle as O

pytestmark = pytest.mark.gpu
MAX_ABS, MIN_PSNR = 4e-3, 62.0


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


@pytest.fixture(scope="module")
def model(det_sd):
    assert torch.cuda.is_available()
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    return m.to("cuda").eval()


FWD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fwd_g*.npz")))


@pytest.mark.parametrize("name", FWD)
def test_forward_matches_reference_fixture(model, golden_dir, name):
    d = dict(np.load(os.path.join(golden_dir, name)))
    kw = {"require_ratio": bool(d["require_ratio"])}
    if int(d["upscale_factor"]) > 0:
        kw["upscale_factor"] = int(d["upscale_factor"])
    else:
        kw["res_out"] = tuple(int(v) for v in d["res_out"])
    with torch.no_grad():
        y = model(torch.from_numpy(d["x"]).cuda(), **kw).cpu()
    ref = torch.from_numpy(d["y"])
    assert tuple(y.shape) == tuple(ref.shape)
    assert y.min() >= 0 and y.max() <= 1
    print(f"{name}: max|d| {(y - ref).abs().max().item():.2e} PSNR {psnr(y, ref):.1f} dB")
    assert (y - ref).abs().max().item() <= MAX_ABS, (y - ref).abs().max().item()
    assert psnr(y, ref) >= MIN_PSNR, psnr(y, ref)


def test_intermediates_vs_oracle(model, det_sd):
    """Stage-by-stage comparison on a geometry with both padding paths (68x84, batch 2)."""
    from transformerupscaler_amd import engine
    x = torch.rand((2, 3, 68, 84), generator=torch.Generator().manual_seed(5))
    cap_o, cap_h = {}, {}
    with torch.no_grad():
        O.forward(det_sd, x, upscale_factor=2, capture=cap_o)
        pk, frags = model.packed(2)
        engine.forward(pk, frags, x.cuda(), 2, (136, 168), True, capture=cap_h, fuse_branch_a=False)
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    checks = [("feat", nhwc(cap_o["feat"]), 3e-2), ("up1", nhwc(cap_o["up1"]), 3e-2),
              ("upscaled_input", cap_o["upscaled_input"], 3e-2), ("win_in", cap_o["win_in"].reshape(-1, 192), 2e-2),
              ("block0", cap_o["block0"].reshape(-1, 192), 5e-2), ("block5", cap_o["block5"].reshape(-1, 192), 1.5e-1),
              ("combined", nhwc(cap_o["combined"]), 8e-2), ("residual", cap_o["residual"], 2e-2)]
    for name, ref, tol in checks:
        got = cap_h[name].float().cpu()
        err = (got - ref).abs().max().item()
        rel = err / max(ref.abs().max().item(), 1e-6)
        assert rel <= tol, f"{name}: max abs {err:.4f} rel {rel:.4f}"


@pytest.mark.parametrize("scale,shape", [(2, (2, 3, 68, 84)), (3, (1, 3, 20, 28)), (4, (1, 3, 20, 28)), (6, (1, 3, 20, 28)), (2, (1, 3, 8, 32))])
def test_composed_branch_a_matches_explicit(model, det_sd, scale, shape):
    """The inference-only composition (last up-conv + PixelShuffle + up1_conv as one 5x5 conv) against the
    oracle's explicit chain, incl. the HR border ring, and against the explicit HIP kernels."""
    from transformerupscaler_amd import engine
    x = torch.rand(shape, generator=torch.Generator().manual_seed(scale))
    cap_o, cap_c, cap_e = {}, {}, {}
    with torch.no_grad():
        O.forward(det_sd, x, upscale_factor=scale, capture=cap_o)
        pk, frags = model.packed(scale)
        res = (shape[2] * scale, shape[3] * scale)
        engine.forward(pk, frags, x.cuda(), scale, res, True, capture=cap_c, fuse_branch_a=True)
        engine.forward(pk, frags, x.cuda(), scale, res, True, capture=cap_e, fuse_branch_a=False)
    ref = cap_o["upscaled_input"]
    got_c, got_e = cap_c["upscaled_input"].cpu(), cap_e["upscaled_input"].cpu()
    tol = 1.2e-2 * max(1.0, ref.abs().max().item())
    assert (got_c - ref).abs().max().item() <= tol, (got_c - ref).abs().max().item()
    ring = torch.ones_like(ref, dtype=torch.bool); ring[..., 1:-1, 1:-1] = False
    assert (got_c - ref)[ring].abs().max().item() <= tol          # border ring uses the weight variants
    # the composition skips the bf16 rounding of the HR intermediate, so it is at least as close as the explicit path
    assert (got_c - ref).abs().mean().item() <= (got_e - ref).abs().mean().item() * 1.1 + 1e-5


def test_packed_weights_are_kept_per_scale(model):
    """Alternating-scale inference re-packs nothing while the parameters are unchanged (one cache entry);
    a parameter update drops every entry."""
    with torch.no_grad():
        pk2 = model.packed(2)[0]
        pk4 = model.packed(4)[0]
        assert model.packed(2)[0] is pk2 and model.packed(4)[0] is pk4
        p = next(model.parameters())
        p.add_(0.0)                                  # in-place write: version counter moves
        assert model.packed(2)[0] is not pk2
        assert len(model._pack_cache) == 1


def test_unbuilt_scale_and_cpu_inputs_raise(model):
    with pytest.raises(ValueError):
        model(torch.rand(1, 3, 16, 16).cuda(), res_out=(80, 80))
    with pytest.raises(RuntimeError):
        model(torch.rand(1, 3, 16, 16), upscale_factor=2)


def test_config1_256(model, golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "fwd_256_s2.npz")))
    x = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(int(d["seed"])))
    with torch.no_grad():
        y = model(x.cuda(), upscale_factor=2).cpu()
    ref = torch.from_numpy(d["y_f16"].astype(np.float32))
    assert (y - ref).abs().max().item() <= MAX_ABS
    assert psnr(y, ref) >= MIN_PSNR


@pytest.mark.parametrize("name,shape,kw", [("fwd_720p_to_1080p.npz", (1, 3, 720, 1280), dict(res_out=(1080, 1920))),
                                           ("fwd_540p_x4.npz", (1, 3, 540, 960), dict(upscale_factor=4))])
def test_full_size_configs(model, golden_dir, name, shape, kw):
    """BASELINE.json config 2 / 4 geometry: patches + row statistics of the reference output."""
    d = dict(np.load(os.path.join(golden_dir, name)))
    x = torch.rand(shape, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = model(x.cuda(), **kw).cpu()
    worst, se, n = 0.0, 0.0, 0
    for i, (a, b) in enumerate(zip(d["ys"].tolist(), d["xs"].tolist())):
        diff = y[0, :, a:a + 32, b:b + 32] - torch.from_numpy(d["patches"][i])
        worst = max(worst, diff.abs().max().item())
        se += (diff.double() ** 2).sum().item(); n += diff.numel()
    assert worst <= MAX_ABS, worst
    assert 10 * np.log10(1.0 / (se / n)) >= MIN_PSNR
    assert abs(y.double().mean().item() - d["stats"][0]) < 2e-3
    assert np.abs(y[0].double().mean(dim=(0, 2)).float().numpy() - d["row_means"]).max() < 5e-3


def test_batch_equals_per_sample(model):
    x = torch.rand((3, 3, 40, 56), generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        yb = model(x, upscale_factor=2)
        ys = torch.cat([model(x[i:i + 1], upscale_factor=2) for i in range(3)])
    assert torch.equal(yb, ys)


@pytest.mark.parametrize("name", ["fwd_g36x48_resize54x72.npz", "fwd_g20x28_s4.npz", "fwd_g20x28_s6.npz", "fwd_g68x84_s2_b2.npz", "fwd_g24x40_res3.npz"])
def test_fused_paths_equal_unfused(model, golden_dir, name):
    """The inference fusions (composed branch A, LN+QKV, fused MLP, fused tail) against the one-kernel-per-op path."""
    from transformerupscaler_amd import engine
    d = dict(np.load(os.path.join(golden_dir, name)))
    kw = {"require_ratio": bool(d["require_ratio"])}
    if int(d["upscale_factor"]) > 0:
        kw["upscale_factor"] = int(d["upscale_factor"])
    else:
        kw["res_out"] = tuple(int(v) for v in d["res_out"])
    x = torch.from_numpy(d["x"]).cuda()
    with torch.no_grad():
        y_f = model(x, **kw).cpu()
        engine.fuse_blocks, engine.fuse_tail = False, False
        try:
            y_u = model(x, **kw).cpu()
        finally:
            engine.fuse_blocks, engine.fuse_tail = True, True
    assert (y_f - y_u).abs().max().item() <= 6e-3, (y_f - y_u).abs().max().item()
