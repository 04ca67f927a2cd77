#!/usr/bin/env python
"""Round-2 golden fixtures from the REAL reference modules (runs only in the build container; needs
/root/reference read-only -- the reference's source never enters this repo, only inputs / outputs):

  train_720p.npz      FastTransformer, train.py:113-140 step at BASELINE configs[2] geometry (B = 1, dropout off):
                      loss + per-parameter gradient statistics / samples / small full gradients
  rt_fwd_x6.npz       ResidualTransformer x6 (720p -> 4320x7680, BASELINE configs[4]): patches, statistics
  rt_train_x6.npz     its L1 training-graph gradients
  psnr_cases.npz      PSNR(reference output, HR) scalars for the ΔPSNR <= 0.01 dB test (inference.py:129-146 prints
                      the same quantity): 720p synthetic, 256^2 synthetic, and a real-image 256^2 crop (HR stored as uint8)
  calib_bf16_autocast.json   how far the REFERENCE graph itself moves when run under torch bf16 autocast on the CPU
                      (forward max |diff|, per-parameter gradient relative L2): the yardstick the GPU gradient
                      tolerances in tests/test_hip_train.py are derived from

    python tests/golden/make_golden_r2.py [--only train720,rtx6,psnr,calib]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import import_reference  # noqa: E402  (same torchvision.transforms.Resize stand-in)


def grad_record(model, n_samples=512, seed=1234):
    out, none_grads = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none_grads.append(k)
            continue
        g = p.grad.detach().double().flatten()
        gi = torch.Generator().manual_seed(seed)
        idx = torch.randperm(g.numel(), generator=gi)[:n_samples].sort().values
        out["gstat_" + k] = np.array([g.sum().item(), g.norm().item(), g.abs().max().item()])
        out["gidx_" + k] = idx.numpy()
        out["gval_" + k] = g[idx].float().numpy()
        if g.numel() <= 4096:
            out["gfull_" + k] = p.grad.detach().float().numpy()
    out["none_grads"] = np.array(none_grads)
    return out


def psnr(a, b):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * np.log10(1.0 / mse)


def make_train720(ref, Resize, sd):
    model = ref.TransformerModel().eval()
    model.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(4321)
    lr = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 1080, 1920), generator=g)
    model.zero_grad()
    o = model(lr, res_out=(1080, 1920), require_ratio=False)            # train.py:124
    assert tuple(o.shape[2:]) == (1440, 2560)
    o = Resize((1080, 1920))(o)                                          # train.py:127-130
    loss = F.l1_loss(o, hr)
    loss.backward()
    out = grad_record(model)
    out["loss"] = np.float64(loss.item())
    out["seed"] = np.array(4321)
    np.savez_compressed(os.path.join(HERE, "train_720p.npz"), **out)
    print("train_720p: loss", loss.item(), "none grads", len(out["none_grads"]))


def make_rtx6(sd_rt):
    sys.path.insert(0, "/root/reference")
    ref = importlib.import_module("models.ResidualTransformer.model")
    sys.path.pop(0)
    model = ref.TransformerModel().eval()
    model.load_state_dict(sd_rt, strict=True)
    x = torch.rand((1, 3, 720, 1280), generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        y = model(x, upscale_factor=6)
    H, W = y.shape[2:]
    assert (H, W) == (4320, 7680)
    gi = torch.Generator().manual_seed(77)
    ys = torch.randint(0, H - 32, (24,), generator=gi); xs = torch.randint(0, W - 32, (24,), generator=gi)
    # corners + edges too: the bicubic index clamping lives there
    ys = torch.cat([ys, torch.tensor([0, 0, H - 32, H - 32])]); xs = torch.cat([xs, torch.tensor([0, W - 32, 0, W - 32])])
    np.savez_compressed(os.path.join(HERE, "rt_fwd_x6.npz"), ys=ys.numpy(), xs=xs.numpy(),
                        patches=np.stack([y[0, :, a:a + 32, b:b + 32].numpy() for a, b in zip(ys.tolist(), xs.tolist())]),
                        stats=np.array([y.double().mean().item(), y.double().norm().item(), (y == 0).double().mean().item(),
                                        (y == 1).double().mean().item()]),
                        row_means=y[0].double().mean(dim=(0, 2)).float().numpy(),
                        col_means=y[0].double().mean(dim=(0, 1)).float().numpy())
    print("rt x6", tuple(y.shape), float(y.mean()))
    g = torch.Generator().manual_seed(9876)
    lr = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 4320, 7680), generator=g)
    model.zero_grad()
    loss = F.l1_loss(model(lr, upscale_factor=6), hr)
    loss.backward()
    d = grad_record(model, n_samples=64, seed=99)
    d["loss"] = np.float64(loss.item())
    np.savez_compressed(os.path.join(HERE, "rt_train_x6.npz"), **d)
    print("rt_train_x6: loss", loss.item())


def make_psnr(ref, sd):
    model = ref.TransformerModel().eval()
    model.load_state_dict(sd, strict=False)
    out = {}
    # 720p synthetic (LR and HR from one generator seeded 1234)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((1, 3, 720, 1280), generator=g)
    hr = torch.rand((1, 3, 1080, 1920), generator=g)
    with torch.no_grad():
        y = model(x, res_out=(1080, 1920))
    out["psnr_720p"] = np.float64(psnr(y, hr))
    # 256^2 synthetic (config 1 input of fwd_256_s2.npz, seed 11; HR from seed 12)
    x = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(11))
    hr = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        y = model(x, upscale_factor=2)
    out["psnr_256"] = np.float64(psnr(y, hr))
    # real image: 512^2 HR crop of the reference's training image, LR = antialiased bilinear /2 (data_class.py:61-71 geometry)
    from PIL import Image
    im = np.asarray(Image.open("/root/reference/images/training_set/image_9.png").convert("RGB"))
    crop = im[800:1312, 1600:2112].copy()
    hr = torch.from_numpy(crop).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
    lr = F.interpolate(hr, size=(256, 256), mode="bilinear", align_corners=False, antialias=True).clamp(0, 1)
    with torch.no_grad():
        y = model(lr, upscale_factor=2)
    out["real_hr_u8"] = crop
    out["real_lr"] = lr.numpy()
    out["real_y_f16"] = y.numpy().astype(np.float16)
    out["psnr_real"] = np.float64(psnr(y, hr))
    np.savez_compressed(os.path.join(HERE, "psnr_cases.npz"), **out)
    print({k: float(v) for k, v in out.items() if k.startswith("psnr")})


def make_calib(ref, Resize, sd):
    """The reference graph under CPU bf16 autocast vs its own fp32 run (same weights / inputs as the GPU tests)."""
    res = {}
    cases = {"s4_20x28": ((1, 3, 20, 28), dict(upscale_factor=4), 4), "s3_24x40": ((2, 3, 24, 40), dict(res_out=(70, 100)), 3),
             "s6_16x24": ((1, 3, 16, 24), dict(upscale_factor=6), 6)}
    for name, (shape, kw, seed) in cases.items():
        x = torch.rand(shape, generator=torch.Generator().manual_seed(seed))
        grads = {}
        outs = {}
        for mode in ("fp32", "bf16"):
            model = ref.TransformerModel().eval()
            model.load_state_dict(sd, strict=False)
            ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cpu", enabled=False)
            with ctx:
                y = model(x, **kw)
            y = y.float()
            R = torch.rand(tuple(y.shape), generator=torch.Generator().manual_seed(99)) - 0.5
            (y * R).sum().backward()
            outs[mode] = y.detach()
            grads[mode] = {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}
        errs = {k: ((grads["bf16"][k] - g).norm() / g.norm().clamp_min(1e-12)).item() for k, g in grads["fp32"].items()}
        v = sorted(errs.values())
        res[name] = {"forward_max_abs": (outs["bf16"] - outs["fp32"]).abs().max().item(),
                     "grad_rel_l2_median": v[len(v) // 2], "grad_rel_l2_worst": v[-1], "worst_param": max(errs, key=errs.get),
                     "n_params": len(v)}
        print(name, res[name])
    res["_note"] = ("reference models/FastTransformer/model.py run under torch.autocast('cpu', bfloat16) vs its fp32 run; smooth cotangent "
                    "sum(out * R); deterministic weights seed 0; generated by tests/golden/make_golden_r2.py --only calib")
    with open(os.path.join(HERE, "calib_bf16_autocast.json"), "w") as f:
        json.dump(res, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="train720,rtx6,psnr,calib")
    args = ap.parse_args()
    only = set(args.only.split(","))
    from transformerupscaler_amd.weights import deterministic_state_dict, rt_deterministic_state_dict
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref, Resize = import_reference()
    sd = deterministic_state_dict(0)
    if "calib" in only:
        make_calib(ref, Resize, sd)
    if "psnr" in only:
        make_psnr(ref, sd)
    if "train720" in only:
        make_train720(ref, Resize, sd)
    if "rtx6" in only:
        make_rtx6(rt_deterministic_state_dict(0))


if __name__ == "__main__":
    main()
