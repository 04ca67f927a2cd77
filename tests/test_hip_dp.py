"""Data-parallel correctness on the GPU box: two ranks with one sample each must produce the
average of the per-sample gradients = the single-process gradients of the 2-sample batch / 2 x 2 ... i.e. with a
sum-type loss, rank-averaged grads == (full-batch grads) / world."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_two_workers(worker, outfile, extra=()):
    """Two ranks sharing cuda:0 over gloo, run ONCE.  Rendezvous through a file in the test's tmp directory (no TCP port to
    race for); each worker arms faulthandler, so a stall ends with both ranks' Python stacks in the assertion message instead
    of a silent timeout."""
    rdzv = outfile + ".rdzv"
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", "file://" + rdzv, outfile, *extra], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=420)[0] for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("DP workers stalled:\n" + "\n-----\n".join(p.communicate()[0] for p in procs))
    assert all(p.returncode == 0 for p in procs), outs
    return outs


def test_two_rank_dp_grads_equal_single_process(det_sd, golden_dir, tmp_path):
    from transformerupscaler_amd.autograd import resize_aa
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_gpu_worker.py")
    outfile = str(tmp_path / "dp_grads.pt")
    run_two_workers(worker, outfile)
    dp = torch.load(outfile)

    d = dict(np.load(os.path.join(golden_dir, "train_g36x44.npz")))
    model = importlib.import_module("models.FastTransformer.model").TransformerModel()
    model.load_state_dict(det_sd, strict=False)
    model = model.cuda().eval()
    lr, hr = torch.from_numpy(d["lr"]).cuda(), torch.from_numpy(d["hr"]).cuda()
    R = torch.rand((2, 3, 54, 66), generator=torch.Generator().manual_seed(5)).cuda() - 0.5
    (resize_aa(model(lr, res_out=(54, 66), require_ratio=False), (54, 66)) * R).sum().backward()
    worst = 0.0
    for k, p in model.named_parameters():
        if p.grad is None:
            assert k not in dp
            continue
        ref = p.grad.cpu().double() / 2            # mean over the 2 ranks of per-sample sum-loss gradients
        got = dp[k].double()
        rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-12)
        worst = max(worst, rel)
        assert rel <= 2e-2, f"{k}: {rel:.4f}"
    print("worst relative L2 between DP(2 ranks) and single process:", worst)


def test_two_rank_dp_residual_transformer(tmp_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _dp_gpu_worker as W
    outfile = str(tmp_path / "dp_rt_grads.pt")
    run_two_workers(W.__file__, outfile, extra=("rt",))
    dp = torch.load(outfile)
    model = W.rt_small_model()
    x, R = W.rt_inputs()
    (model(x.cuda(), res_out=(96, 144)) * R.cuda()).sum().backward()
    for k, p in model.named_parameters():
        ref, got = p.grad.cpu().double() / 2, dp[k].double()
        rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-12)
        assert rel <= 2e-2, f"{k}: {rel:.4f}"
