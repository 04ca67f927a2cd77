#!/usr/bin/env python
"""Golden fixtures for the ResidualTransformer path from the REAL reference module
(/root/reference/models/ResidualTransformer/model.py; runs only in the build container).
The module only accepts 720x1280 inputs (fixed 3600-token pos_embed), so the fixtures are statistics, patches
and hooked intermediates of full-size runs."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True


def main():
    sys.path.insert(0, ROOT)
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    sys.path.insert(0, "/root/reference")
    ref = importlib.import_module("models.ResidualTransformer.model")
    sys.path.pop(0)
    torch.set_num_threads(8)
    model = ref.TransformerModel().eval()
    sd = rt_deterministic_state_dict(0)
    res = model.load_state_dict(sd, strict=True)
    x = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234))
    caps = {}
    mods = dict(model.named_modules())
    hs = []
    for name in ("downsample", "patch_embed", "transformer_blocks.0", "transformer_blocks.7", "patch_unembed", "decoder_conv2"):
        def hook(_m, _i, out, name=name):
            caps[name] = out.detach().clone()
        hs.append(mods[name].register_forward_hook(hook))
    for tag, kw in (("1080p", dict(res_out=(1080, 1920))), ("x2", dict(upscale_factor=2))):
        with torch.no_grad():
            y = model(x, **kw)
        H, W = y.shape[2:]
        gi = torch.Generator().manual_seed(77)
        ys = torch.randint(0, H - 32, (16,), generator=gi); xs = torch.randint(0, W - 32, (16,), generator=gi)
        out = dict(ys=ys.numpy(), xs=xs.numpy(),
                   patches=np.stack([y[0, :, a:a + 32, b:b + 32].numpy() for a, b in zip(ys.tolist(), xs.tolist())]),
                   stats=np.array([y.double().mean().item(), y.double().norm().item(), (y == 0).double().mean().item(), (y == 1).double().mean().item()]),
                   row_means=y[0].double().mean(dim=(0, 2)).float().numpy())
        if tag == "1080p":
            out["cap_decoder_conv2"] = caps["decoder_conv2"].numpy()                       # residual [1,3,360,640]
            out["cap_block0_head"] = caps["transformer_blocks.0"][0, :64].numpy()           # first 64 tokens
            out["cap_block7_head"] = caps["transformer_blocks.7"][0, :64].numpy()
            out["cap_block7_tail"] = caps["transformer_blocks.7"][0, -64:].numpy()
            out["cap_downsample_patch"] = caps["downsample"][0, :, 100:116, 200:216].numpy()
        np.savez_compressed(os.path.join(HERE, f"rt_fwd_{tag}.npz"), **out)
        print(tag, tuple(y.shape), float(y.mean()), float((y == 0).float().mean()), float((y == 1).float().mean()))
    for h in hs:
        h.remove()
    make_train_fixture(model)


def make_train_fixture(model):
    """One training-graph evaluation (train.py:124-138 with dropout off = eval graph): L1 against a seeded HR target,
    backward through the reference module; per parameter: gradient norm / abs-max, 64 sampled entries, and the
    full gradient of the small parameters."""
    g = torch.Generator().manual_seed(4321)
    lr = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 1080, 1920), generator=g)
    model.zero_grad()
    out = model(lr, res_out=(1080, 1920))
    loss = torch.nn.functional.l1_loss(out, hr)
    loss.backward()
    d = {"loss": np.float64(loss.item())}
    gi = torch.Generator().manual_seed(99)
    for k, p in model.named_parameters():
        gr = p.grad.detach().double().flatten()
        idx = torch.randint(0, gr.numel(), (64,), generator=gi)
        d["gstat_" + k] = np.array([gr.mean().item(), gr.norm().item(), gr.abs().max().item()])
        d["gidx_" + k] = idx.numpy()
        d["gval_" + k] = gr[idx].float().numpy()
        if gr.numel() <= 4096:
            d["gfull_" + k] = p.grad.detach().float().numpy()
    np.savez_compressed(os.path.join(HERE, "rt_train_1080p.npz"), **d)
    print("train fixture: loss", loss.item(), "params", len(list(model.named_parameters())))


if __name__ == "__main__":
    main()
