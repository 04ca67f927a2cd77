"""Worker for tests/test_hip_dp.py: python _dp_gpu_worker.py RANK WORLD INIT_METHOD OUTFILE [rt].
Both ranks share cuda:0 (one-GPU box), so the collective runs over gloo; the reducer, bucketing and
backward-overlap logic are the ones the RCCL path uses."""
import faulthandler
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, init, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    faulthandler.dump_traceback_later(300, exit=True)        # a stall leaves every thread's stack on stderr
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=init, rank=rank, world_size=world)
    if len(sys.argv) > 5 and sys.argv[5] == "rt":
        return main_rt(rank, outfile)
    from transformerupscaler_amd.autograd import resize_aa
    from transformerupscaler_amd.dp import DataParallel
    from transformerupscaler_amd.weights import deterministic_state_dict
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "train_g36x44.npz")))
    model = importlib.import_module("models.FastTransformer.model").TransformerModel()
    model.load_state_dict(deterministic_state_dict(0), strict=False)
    model = model.cuda().eval()
    DataParallel(model, scale=2, bucket_mb=2.0)
    lr = torch.from_numpy(d["lr"])[rank:rank + 1].cuda()           # one sample per rank
    hr = torch.from_numpy(d["hr"])[rank:rank + 1].cuda()
    out = resize_aa(model(lr, res_out=(54, 66), require_ratio=False), (54, 66))
    R = torch.rand((2, 3, 54, 66), generator=torch.Generator().manual_seed(5))[rank:rank + 1].cuda() - 0.5
    (out * R).sum().backward()
    if rank == 0:
        torch.save({k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}, outfile)
    dist.barrier()
    dist.destroy_process_group()
    print(f"RANK{rank} OK", flush=True)


def rt_small_model():
    """ResidualTransformer plugin on a 4 x 6 token grid (64 x 96 input)."""
    import torch.nn as nn
    from transformerupscaler_amd.weights import rt_deterministic_state_dict
    sd = rt_deterministic_state_dict(0)
    sd["pos_embed"] = sd["pos_embed"][:, :24].clone()
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.pos_embed = nn.Parameter(torch.empty(1, 24, 128))
    m.num_tokens = 24
    m.load_state_dict(sd)
    return m.cuda().eval()


def rt_inputs():
    g = torch.Generator().manual_seed(11)
    return torch.rand((2, 3, 64, 96), generator=g), torch.rand((2, 3, 96, 144), generator=g) - 0.5


def main_rt(rank, outfile):
    from transformerupscaler_amd.dp import DataParallel
    model = rt_small_model()
    DataParallel(model, bucket_mb=1.0)            # scale=None: every parameter is active
    x, R = rt_inputs()
    out = model(x[rank:rank + 1].cuda(), res_out=(96, 144))
    (out * R[rank:rank + 1].cuda()).sum().backward()
    if rank == 0:
        torch.save({k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None}, outfile)
    dist.barrier()
    dist.destroy_process_group()
    print(f"RANK{rank} OK", flush=True)


if __name__ == "__main__":
    main()
