"""Whole-block kernels A/B on the MI355X box: interleaved rounds in one process (guide 5.4 rule 24), random data.
    python scripts/microbench_block.py [nwin=1920] [rounds=15]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from transformerupscaler_amd import ops

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dev = "cuda"
import test_hip_kernels as T
raw, args = T._block_operands(dev, nwin)
x0 = raw["x"].to(dev)
variants = {"block x6 launches": 32, "blocks32, one launch": -1}
table6 = ops.block_table([tuple(args)] * 6)
xs = {k: x0.clone() for k in variants}
times = {k: [] for k in variants}
for k, tpw in variants.items():                       # warm-up
    for _ in range(3):
        ops.fused_block(xs[k], *args)
torch.cuda.synchronize()
for r in range(rounds):
    for k, tpw in variants.items():
        xs[k].copy_(x0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if tpw == -1:
            ops.fused_blocks32(xs[k], table6)
        else:
            for _ in range(6):                            # ... as 6 launches
                ops.fused_block(xs[k], *args)
        e.record(); torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) / 6 * 1e3)
gf = 86.1e9 / 6 * nwin / 240
for k, t in times.items():
    t = sorted(t)
    med, mn = t[len(t) // 2], t[0]
    print(f"{k}: median {med:.1f} us  min {mn:.1f} us  -> {gf / (med * 1e-6) / 1e12:.0f} TFLOP/s on the attention GEMM set ({gf / (med * 1e-6) / 2.5e15:.3f} of 2.5 PF)")
