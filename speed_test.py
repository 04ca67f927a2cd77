#!/usr/bin/env python
"""Synced single-image latency / throughput tool with the reference's speed_test.py surface (reference
speed_test.py:20-88: ``--model``, ``--checkpoint_dir``, ``--data_dir``; one image at a time, ``res_out=(2160, 3840)``),
plus what that script lacks on a GPU: device synchronisation around the timed region (the reference times only the
asynchronous launch, speed_test.py:62-66), uint8 frame pre/post-processing on the GPU (app_overlay.py:381-388) and
an optional hipGraph replay of the whole frame (``--graph``) for the live-overlay latency path (app_overlay.py:337-420).

    python speed_test.py --model FastTransformer --frames 50 --res_in 720 --res_out 2160 3840 --graph

Without ``--data_dir`` (or when PIL is missing) it runs on synthetic uint8 frames; without a checkpoint directory it
uses the deterministic synthetic weights of the test-suite.  Prints one JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.utils import get_latest_checkpoint, resolutions  # noqa: E402


def load_frames(args, device):
    h, w = resolutions[str(args.source_res or args.res_in)]
    if args.data_dir:
        try:
            from PIL import Image
            import numpy as np
            names = sorted(f for f in os.listdir(args.data_dir) if f.lower().endswith((".png", ".jpg", ".jpeg")))[: args.frames]
            frames = [torch.from_numpy(np.asarray(Image.open(os.path.join(args.data_dir, n)).convert("RGB").resize((w, h))).copy())
                      for n in names]
            if frames:
                return [f.to(device) for f in frames], "files"
        except ImportError:
            pass
    g = torch.Generator().manual_seed(1234)
    return [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, generator=g).to(device) for _ in range(min(args.frames, 8))], "synthetic"


def frame_latency(model, frames, res_out, n_frames, warmup, graph, bgr=False, lr_hw=None):
    """The live-overlay frame loop (reference app_overlay.py:337-420, speed_test.py:56-75): uint8 HWC frame on the GPU ->
    [Resize to lr_hw] -> ToTensor -> model(res_out) -> uint8 HWC frame, one frame at a time, each timed to completion
    (`lat`) and to the return of the asynchronous call (`launch`, what the reference's unsynchronised timer sees).
    graph=True: the whole frame is captured once into a hipGraph and replayed.  Returns (last output, lat, launch, wall)."""
    from transformerupscaler_amd import ops

    def one_frame(frame_u8):
        if lr_hw is not None and tuple(frame_u8.shape[:2]) != tuple(lr_hw):
            x = ops.resize_frames(frame_u8, lr_hw, to_tensor=True, bgr=bgr)
        else:
            x = ops.frames_to_tensor(frame_u8, bgr=bgr)
        y = model(x, res_out=res_out)
        return ops.tensor_to_frames(y, bgr=bgr)

    g = static_in = static_out = None
    with torch.no_grad():
        for i in range(warmup):
            out = one_frame(frames[i % len(frames)])
        torch.cuda.synchronize()
        if graph:
            static_in = frames[0].clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one_frame(static_in)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = one_frame(static_in)
            torch.cuda.synchronize()

        def run(frame):
            if g is None:
                return one_frame(frame)
            static_in.copy_(frame)
            g.replay()
            return static_out

        lat, launch = [], []
        t_all = time.perf_counter()
        for i in range(n_frames):
            f = frames[i % len(frames)]
            t0 = time.perf_counter()
            out = run(f)
            t1 = time.perf_counter()           # what the reference's speed_test.py measures (no sync)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            launch.append(t1 - t0)
            lat.append(t2 - t0)
        wall = time.perf_counter() - t_all
    return out, lat, launch, wall


def latency_summary(lat):
    lat = sorted(lat)
    return {"mean": 1e3 * sum(lat) / len(lat), "p50": 1e3 * lat[len(lat) // 2], "p99": 1e3 * lat[min(len(lat) - 1, int(0.99 * len(lat)))]}


def main():
    ap = argparse.ArgumentParser(description="Synced speed test for the MI355X upscaler plugins")
    ap.add_argument("--data_dir", type=str, default=None)
    ap.add_argument("--model", type=str, default="FastTransformer")
    ap.add_argument("--checkpoint_dir", type=str, default=None)
    ap.add_argument("--res_in", type=str, default="720", choices=list(resolutions))
    ap.add_argument("--res_out", type=int, nargs=2, default=(2160, 3840))
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graph", action="store_true", help="capture one frame (pre-process, model, post-process) into a hipGraph and replay it")
    ap.add_argument("--bgr", action="store_true", help="frames are BGR (screen grabs), output BGR")
    ap.add_argument("--source_res", type=str, default=None, choices=list(resolutions),
                    help="frames arrive at this resolution and are resized to --res_in on the GPU (transforms.Resize + ToTensor, "
                         "bit-exact with Pillow: the LR transform of data_handling/data_class.py:61-64 and inference.py:65-68)")
    args = ap.parse_args()

    device = torch.device("cuda", 0)
    from transformerupscaler_amd import ops
    model = importlib.import_module(f"models.{args.model}.model").TransformerModel().to(device)
    ckpt_dir = args.checkpoint_dir or f"models/{args.model}/checkpoints"
    weights = "deterministic synthetic"
    if os.path.isdir(ckpt_dir) and any(f.endswith(".pth") for f in os.listdir(ckpt_dir)):
        path, epoch = get_latest_checkpoint(ckpt_dir)
        model.load_state_dict(torch.load(path, map_location=device))
        weights = f"{path} (epoch {epoch})"
    else:
        from transformerupscaler_amd import weights as W
        sd = {"ResidualTransformer": W.rt_deterministic_state_dict, "WindowTransformer": W.wt_deterministic_state_dict}.get(
            args.model, W.deterministic_state_dict)(0)
        model.load_state_dict(sd, strict=False)
    model.eval()
    frames, source = load_frames(args, device)
    res_out = tuple(args.res_out)
    lr_hw = resolutions[str(args.res_in)]
    out, lat, launch, wall = frame_latency(model, frames, res_out, args.frames, args.warmup, args.graph, bgr=args.bgr,
                                           lr_hw=lr_hw if args.source_res else None)
    res = {"model": args.model, "weights": weights, "frames": args.frames, "source": source, "res_in": resolutions[str(args.res_in)],
           "res_out": list(out.shape[1:3]), "graph": bool(args.graph),
           "latency_ms": latency_summary(lat),
           "unsynced_launch_ms_mean": 1e3 * sum(launch) / len(launch), "images_per_sec": args.frames / wall,
           "includes": ("uint8 HWC -> Resize (Pillow-exact) -> model -> uint8 HWC on the GPU (no PCIe)" if args.source_res
                        else "uint8 HWC -> model -> uint8 HWC on the GPU (no PCIe)")}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
