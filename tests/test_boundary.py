"""CPU-side boundary checks: the C-ABI library loads and exports every symbol include/tupscale_hip.h
declares (no compute without a GPU), the plugin module keeps the reference's surface, and the host
packing logic produces the layouts the kernels assume (emulated with torch on the CPU)."""
import importlib
import inspect
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from transformerupscaler_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tupscale_hip.h")).read()
    declared = set(re.findall(r"\b(?:int|long long)\s+(tup_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                      # raises if the .so or any symbol is missing
    assert lib.tup_abi_version() == _lib.ABI_VERSION
    for name in declared:
        assert hasattr(lib, name)


def test_plugin_surface_matches_reference_contract():
    mod = importlib.import_module("models.FastTransformer.model")
    sig = inspect.signature(mod.TransformerModel.__init__)
    assert [p for p in sig.parameters][1:] == ["in_channels", "base_channels", "transformer_dim", "num_window_blocks",
                                               "num_heads", "mlp_ratio", "dropout", "window_size"]
    fsig = inspect.signature(mod.TransformerModel.forward)
    assert [p for p in fsig.parameters][1:] == ["x", "res_out", "upscale_factor", "require_ratio"]
    assert fsig.parameters["res_out"].default == (1080, 1920) and fsig.parameters["require_ratio"].default is True
    m = mod.TransformerModel()
    from transformerupscaler_amd.weights import param_shapes
    sd = m.state_dict()
    keys = [k for k in sd if not k.endswith("relative_position_index")]
    assert keys == list(param_shapes().keys())
    assert all(tuple(sd[k].shape) == s for k, s in param_shapes().items())
    assert sum(p.numel() for p in m.parameters()) == 6447379
    idx = sd["window_blocks.0.attn.relative_position_index"]
    assert idx.dtype == torch.int64 and idx[0, 0] == 112 and idx.max() == 224
    with pytest.raises(RuntimeError):          # no silent CPU path
        m(torch.rand(1, 3, 16, 16), upscale_factor=2)
    with pytest.raises(ValueError):            # utils.py:96-97
        m(torch.rand(1, 3, 16, 16), res_out=(80, 80))


def test_checkpoint_roundtrip_and_latest(tmp_path, det_sd):
    from tools.utils import get_latest_checkpoint
    mod = importlib.import_module("models.FastTransformer.model")
    m = mod.TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    for e in (1, 2, 10):
        torch.save(m.state_dict(), tmp_path / f"model_epoch_{e}.pth")     # train.py:152-156 naming
    path, epoch = get_latest_checkpoint(str(tmp_path))
    assert epoch == 10 and path.endswith("model_epoch_10.pth")
    m2 = mod.TransformerModel()
    m2.load_state_dict(torch.load(path, map_location="cpu"))               # strict, train.py:90
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_harness_checkpoint_resume_with_optimizer(tmp_path, det_sd):
    """harness.save_checkpoint / load_latest_checkpoint: reference-format weight file + Adam sidecar."""
    from transformerupscaler_amd import harness
    mod = importlib.import_module("models.FastTransformer.model")
    m = mod.TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    opt = harness.make_optimizer(m)
    p = m.conv1.weight
    p.grad = torch.ones_like(p)
    opt.step()
    path = harness.save_checkpoint(m, str(tmp_path), 3, optimizer=opt)
    assert path.endswith("model_epoch_3.pth")
    plain = torch.load(path, map_location="cpu")
    assert list(plain.keys()) == list(m.state_dict().keys())               # what the reference's load_state_dict expects
    m2 = mod.TransformerModel()
    opt2 = harness.make_optimizer(m2)
    assert harness.load_latest_checkpoint(m2, str(tmp_path), optimizer=opt2, map_location="cpu") == 3
    assert torch.equal(m2.conv1.weight, m.conv1.weight)
    st = opt2.state_dict()["state"]
    assert len(st) == 1 and float(next(iter(st.values()))["step"]) == 1.0
    assert harness.load_latest_checkpoint(m2, str(tmp_path / "missing")) == 0


def _emulate_conv_c64(x_nhwc, wp, bp, r):
    """What conv3x3_c64_kernel<4,0> computes from the packed operands (fp32 emulation)."""
    from transformerupscaler_amd.packing import _PERM64
    B, H, W, _ = x_nhwc.shape
    xp = F.pad(x_nhwc.permute(0, 3, 1, 2), (1, 1, 1, 1))
    out = torch.zeros(B, H * r, W * r, 64)
    inv = torch.empty(64, dtype=torch.long); inv[_PERM64] = torch.arange(64)
    for nt in range(r * r):
        acc = torch.zeros(B, 64, H, W)
        for tap in range(9):
            dy, dx = tap // 3, tap % 3
            acc += torch.einsum("nc,bchw->bnhw", wp[nt, 0, tap].float(), xp[:, :, dy:dy + H, dx:dx + W])
        acc = acc[:, inv] + bp[nt].view(1, 64, 1, 1)          # row n_local holds channel perm[n_local]
        out[:, nt // r::r, nt % r::r, :] = acc.permute(0, 2, 3, 1)
    return out


@pytest.mark.parametrize("r", [1, 2, 3])
def test_conv_packing_layout(r):
    from transformerupscaler_amd import packing
    g = torch.Generator().manual_seed(r)
    x = torch.rand((1, 64, 6, 7), generator=g)
    w, b = torch.rand((64 * r * r, 64, 3, 3), generator=g) - 0.5, torch.rand((64 * r * r,), generator=g)
    wp, bp = packing.pack_conv_c64(w, b, r)
    ref = F.pixel_shuffle(F.conv2d(x, w.to(torch.bfloat16).float(), b, padding=1), r).permute(0, 2, 3, 1)
    got = _emulate_conv_c64(x.permute(0, 2, 3, 1), wp, bp, r)
    assert (got - ref).abs().max() < 1e-4


def test_patch_and_linear_packing_layout():
    from transformerupscaler_amd import packing
    from transformerupscaler_amd.packing import _PERM64
    g = torch.Generator().manual_seed(0)
    inv = torch.empty(64, dtype=torch.long); inv[_PERM64] = torch.arange(64)
    unperm = lambda t: t.view(-1, 64, t.shape[-1])[:, inv].reshape(t.shape)      # rows back to natural order
    w = torch.rand((192, 64, 8, 8), generator=g) - 0.5
    feat = torch.rand((1, 64, 16, 24), generator=g)
    wt = unperm(packing.pack_patch_embed(w).float())                  # [192][(i*8+j)*64+c]
    patches = feat.unfold(2, 8, 8).unfold(3, 8, 8)                    # [1][c][ty][tx][i][j]
    a = patches.permute(0, 2, 3, 4, 5, 1).reshape(-1, 4096)
    ref = F.conv2d(feat, w.to(torch.bfloat16).float(), None, stride=8).permute(0, 2, 3, 1).reshape(-1, 192)
    assert (a @ wt.t() - ref).abs().max() < 1e-3
    wu = torch.rand((192, 64, 8, 8), generator=g) - 0.5
    tok = torch.rand((1, 192, 2, 3), generator=g)
    wtu = unperm(packing.pack_patch_unembed(wu).float())              # [(i*8+j)*64+o][192]
    y = (tok.permute(0, 2, 3, 1).reshape(-1, 192) @ wtu.t()).view(1, 2, 3, 8, 8, 64)    # [b][ty][tx][i][j][o]
    y = y.permute(0, 5, 1, 3, 2, 4).reshape(1, 64, 16, 24)
    refu = F.conv_transpose2d(tok, wu.to(torch.bfloat16).float(), None, stride=8)
    assert (y - refu).abs().max() < 1e-3


def test_resize_taps_match_oracle_and_aten():
    from oracle import fast_transformer_oracle as O
    from transformerupscaler_amd.resize_taps import aa_taps
    for i, o in ((1440, 1080), (2560, 1920), (72, 54), (40, 47), (64, 48)):
        lo, n, w, k = aa_taps(i, o)
        lo2, n2, w2 = O.aa_bilinear_taps(i, o)
        assert (lo == lo2).all() and (n == n2).all() and k == w2.shape[1]
        assert abs(w - w2).max() == 0.0


@pytest.mark.parametrize("r", [2, 3])
def test_branch_a_composition_is_exact(r):
    """compose_branch_a (fp32, CPU) == Conv(64->64rr)+PixelShuffle+Conv(64->3) incl. the border variants."""
    from transformerupscaler_amd import packing
    g = torch.Generator().manual_seed(r)
    H, W = 6, 7
    feat = torch.rand((1, 64, H, W), generator=g)
    wu, bu = torch.rand((64 * r * r, 64, 3, 3), generator=g) - 0.5, torch.rand((64 * r * r,), generator=g)
    w3 = torch.rand((3, 64, 3, 3), generator=g) - 0.5
    ref = F.conv2d(F.pixel_shuffle(F.conv2d(feat, wu, bu, padding=1), r), w3, None, padding=1)
    fp = F.pad(feat, (2, 2, 2, 2))
    out = torch.zeros_like(ref)
    Hs, Ws = H * r, W * r
    for Y in range(Hs):
        for X in range(Ws):
            rm = 1 if Y == 0 else (2 if Y == Hs - 1 else 0)
            cm = 1 if X == 0 else (2 if X == Ws - 1 else 0)
            wc, bc = packing.compose_branch_a(wu, bu, w3, r, rm, cm)
            y, si, x, sj = Y // r, Y % r, X // r, X % r
            win = fp[0, :, y:y + 5, x:x + 5].permute(1, 2, 0)            # [5][5][64]
            for c in range(3):
                n = c * r * r + si * r + sj
                out[0, c, Y, X] = (wc[n] * win).sum() + bc[n]
    assert (out - ref).abs().max() < 2e-4


def test_transposed_bicubic_taps_are_the_adjoint():
    """resize_taps.transpose_taps: the per-source CSR lists used by the gather-form bicubic backward are exactly the
    transpose of the forward tap matrix, which itself reproduces F.interpolate(bicubic) (so the backward = autograd)."""
    from transformerupscaler_amd.resize_taps import bicubic_taps, transpose_taps
    for n_in, n_out in ((10, 47), (14, 66), (360, 1080)):
        idx, w = bicubic_taps(n_in, n_out)
        dense = np.zeros((n_out, n_in), dtype=np.float64)
        for o in range(n_out):
            for k in range(4):
                dense[o, idx[o, k]] += w[o, k]
        x = torch.rand(1, 1, n_in, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(n_in))
        ref = F.interpolate(x.float(), size=(n_out, 1), mode="bicubic", align_corners=False)[0, 0, :, 0].double().numpy()
        assert np.abs(dense @ x[0, 0, :, 0].numpy() - ref).max() < 3e-5        # the fp32 rounding of F.interpolate itself
        start, oi, ow = transpose_taps(idx, w, n_in)
        assert start[0] == 0 and start[-1] == 4 * n_out and np.all(np.diff(start) >= 0)
        dense_t = np.zeros((n_in, n_out), dtype=np.float64)
        for i in range(n_in):
            for t in range(start[i], start[i + 1]):
                dense_t[i, oi[t]] += ow[t]
        assert np.array_equal(dense_t, dense.T)


def test_stride2_dgrad_packing_matches_autograd():
    """pack_conv_c64_stride2_dgrad: a 3x3 conv over the output gradient with one output tile per input sub-pixel, stored
    through PixelShuffle(2), equals autograd of Conv2d(stride=2, padding=1) w.r.t. its input."""
    from transformerupscaler_amd import packing
    from transformerupscaler_amd.packing import _PERM64
    g = torch.Generator().manual_seed(5)
    x = torch.rand((1, 64, 12, 16), generator=g, requires_grad=True)
    w = (torch.rand((64, 64, 3, 3), generator=g) - 0.5).to(torch.bfloat16).float()
    y = F.conv2d(x, w, stride=2, padding=1)
    gy = torch.rand(y.shape, generator=g) - 0.5
    y.backward(gy)
    wp = packing.pack_conv_c64_stride2_dgrad(w)[:, 0].float()            # [sub-pixel][tap][ci perm][co]
    inv = torch.empty(64, dtype=torch.long); inv[_PERM64] = torch.arange(64)
    wk = wp[:, :, inv, :]                                                # rows back in natural ci order
    wfull = wk.permute(2, 0, 3, 1).reshape(64, 4, 64, 3, 3)              # [ci][sp][co][ky][kx]
    planes = F.conv2d(gy, wfull.reshape(256, 64, 3, 3), padding=1)       # channel = ci*4 + sp
    got = F.pixel_shuffle(planes, 2)
    assert (got - x.grad).abs().max() <= 1e-4 * max(1.0, x.grad.abs().max().item())
    dwp = torch.zeros(4, 64, 9, 64)
    assert torch.equal(packing.unpack_conv_c64_stride2_wgrad(dwp), torch.zeros(64, 64, 3, 3))


def test_pack_plan_maps_on_cpu(det_sd):
    """The bit-plane trace of pack_plan.PackPlan (no GPU needed to BUILD a plan): applying its maps with torch indexing
    reproduces packing.pack_state_dict(backward=True) bit for bit -- i.e. every training-time packed tensor is a pure
    gather of the parameters and the traced maps are the right ones."""
    import importlib
    from transformerupscaler_amd import packing
    from transformerupscaler_amd.pack_plan import PackPlan
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(det_sd, strict=False)
    params = list(m.named_parameters())
    plan = PackPlan(params, lambda d: packing.pack_state_dict(d, 2, backward=True))
    flat = torch.cat([p.detach().reshape(-1) for _, p in params])
    ref = packing.pack_state_dict(dict(params), 2, backward=True)
    assert set(ref) == set(plan.views)
    for k, (dt, off, shape) in plan.views.items():
        mp = (plan.map_bf16 if dt == torch.bfloat16 else plan.map_f32)[off:off + ref[k].numel()].long()
        got = torch.where(mp >= 0, flat[mp.clamp(min=0)], torch.zeros(())).to(dt).view(shape)
        assert ref[k].dtype == dt and torch.equal(got, ref[k]), k


def test_integration_md_ctypes_snippet_matches_header():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer would write.  Its `argtypes` / `restype` lines are executed
    here against a recording stand-in for the library and compared with _lib.SIGNATURES (held equal to include/tupscale_hip.h
    by test_library_exports_every_declared_symbol), and the example CALL must pass exactly that many arguments (VERDICT r2 #13:
    the document had drifted to 4 pointers + 8 ints)."""
    import ctypes
    from transformerupscaler_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding the C ABI directly"):text.index("## 3. Reference-side mapping")]
    code = sec[sec.index("```python") + len("```python"):]
    code = code[:code.index("```")]

    class _Fn:
        pass

    class _Lib:
        def __init__(self):
            self.fns = {}

        def __getattr__(self, name):
            if name.startswith("tup_"):
                return self.fns.setdefault(name, _Fn())
            raise AttributeError(name)

    lib = _Lib()
    sig_lines = [ln for ln in code.splitlines() if re.match(r"\s*lib\.tup_\w+\.(argtypes|restype)\s*=", ln)]
    assert sig_lines, "no argtypes / restype lines found in INTEGRATION.md section 2"
    exec("\n".join(sig_lines), {"ctypes": ctypes, "lib": lib})
    assert lib.fns, "the snippet binds no entry point"
    for name, fn in lib.fns.items():
        assert name in _lib.SIGNATURES, f"INTEGRATION.md binds {name}, which the header does not declare"
        assert list(fn.argtypes) == list(_lib.SIGNATURES[name]), (name, fn.argtypes, _lib.SIGNATURES[name])
        assert fn.restype is ctypes.c_int
        # the example call passes one value per declared argument
        m = re.search(r"lib\." + name + r"\((.*?)\)\n\s*if err", code, re.S)
        assert m, f"no example call of {name} found"
        call = re.sub(r"#[^\n]*", "", m.group(1))
        depth, nargs, cur = 0, 0, ""
        for ch in call:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                nargs += bool(cur.strip())
                cur = ""
            else:
                cur += ch
        nargs += bool(cur.strip())
        assert nargs == len(_lib.SIGNATURES[name]), (name, nargs, len(_lib.SIGNATURES[name]))


def test_tail_routing_respects_the_streaming_tails_offset_limits(monkeypatch):
    """engine.forward must not hand the streaming tail (32-bit byte offsets, csrc/tail_stream.hip) a batch it refuses: beyond
    B*12*H*W = 2^29 elements of the last stage's HR map (or B*3*Ho*Wo of the resized output) the tiled tail takes over."""
    from transformerupscaler_amd import engine, ops
    assert ops.tail_stream_fits(8, 720, 1280) and ops.tail_stream_fits(8, 720, 1280, (1080, 1920))
    assert ops.tail_stream_fits(12, 1440, 2560) and not ops.tail_stream_fits(13, 1440, 2560)        # 4x of 720p: fails from B = 13
    assert ops.tail_stream_fits(21, 1080, 1920) and not ops.tail_stream_fits(22, 1080, 1920)        # 540p -> 2160p: from B = 22
    assert ops.tail_stream_fits(1, 64, 64, (128, 128))                                               # identity "resize" = no resize
    assert not ops.tail_stream_fits(1, 4000, 4000, (7999, 7999 * 3))                                 # the resized output's own limit
    # the router itself, with the kernels replaced by recorders (no GPU): an oversized last stage must reach tail_fused
    calls = []
    big = torch.empty((13, 3, 1440, 2560), device="meta")
    monkeypatch.setattr(ops, "tail_stream_r2", lambda *a, **k: calls.append("stream") or "s")
    monkeypatch.setattr(ops, "tail_fused", lambda *a, **k: calls.append("tiled") or "t")
    src = inspect.getsource(engine.forward)
    assert "ops.tail_stream_fits(" in src and src.index("ops.tail_stream_fits(") < src.index("ops.tail_stream_r2(")
    fits = ops.tail_stream_fits(big.shape[0], big.shape[2], big.shape[3], None)
    (ops.tail_stream_r2 if fits else ops.tail_fused)(big)
    assert calls == ["tiled"]


def test_rt_attention_dropout_on_an_unsupported_token_count_raises_a_clear_error():
    from transformerupscaler_amd import ops
    with pytest.raises(ValueError, match="multiple of 4"):
        ops.rt_attention(torch.empty((2 * 506, 384), dtype=torch.bfloat16), 2, 506, drop_p=0.1)
    with pytest.raises(ValueError, match="multiple of 4"):
        ops.rt_attention_bwd(None, None, None, None, 1, 22 * 23, drop_p=0.1)


def test_halo_row_read_addresses_of_the_weight_gradient_kernels():
    """The conv / branch-A weight-gradient kernels (csrc/conv_bwd.hip, csrc/branch_a_train.hip) read their transposed X fragments at
    `xb0 + ((xc ^ ((xs + k * hy) & 7)) << 4) + hy * row_bytes` -- lane constants + the halo row -- instead of swz128(q, chunk) of the pixel
    index.  Restated here for every lane, tap column and halo row against common.h's swz128, with the divisions by the halo width the
    kernels replace by a multiply-shift, and with the source-side swizzle of the DMA pieces of the wide weight-gradient kernel."""
    def swz128(row, chunk):
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)

    for halo_w, halo_h, ndx, k in ((34, 10, 3, 1), (36, 12, 5, 2)):          # 3x3 convs; the composed 5x5 branch A
        assert halo_w % 2 == 0 and (halo_w // 2) % 8 == k % 8                  # a halo row shifts the swizzle phase by k
        for lane in range(64):
            g, l16 = lane >> 4, lane & 15
            trq, trp = l16 >> 2, l16 & 3
            for cit in range(4):
                xcol = 16 * cit + 4 * trp
                xc = xcol >> 3
                for dx in range(ndx):
                    for h in range(2):
                        q0 = 8 * g + trq + dx + 4 * h
                        xs, xb0 = q0 >> 1, q0 * 128 + (xcol & 7) * 2
                        for hy in range(halo_h):
                            got = xb0 + ((xc ^ ((xs + k * hy) & 7)) << 4) + hy * halo_w * 128
                            q = hy * halo_w + q0
                            assert got == swz128(q, xcol >> 3) + (xcol & 7) * 2, (halo_w, lane, cit, dx, h, hy)
    # G tiles are 32 pixels wide: the phase of pixel ry * 32 + 8g + trq + 4h does not depend on the row
    for ry in range(8):
        for p in range(32):
            assert ((ry * 32 + p) >> 1) & 7 == (p >> 1) & 7
    assert all((q * 241) >> 13 == q // 34 for q in range(340))
    assert all((q * 911) >> 15 == q // 36 for q in range(432))
    # gemm_wgrad_wide_kernel: thread tid moves physical slot tid & 7 of row (tid >> 3) (+ 32): it must fetch logical chunk
    # dc = slot ^ ((row >> 1) & 7), so that a fragment read at swz128(row, chunk) finds chunk `chunk`
    for tid in range(512):
        row, slot = tid >> 3, tid & 7
        dc = slot ^ ((tid >> 4) & 7)
        assert swz128(row, dc) == row * 128 + slot * 16
        assert swz128(row + 32, dc) == (row + 32) * 128 + slot * 16           # the second row of a 256-thread workgroup
