import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_kernels import _block_operands, _block_torch
from transformerupscaler_amd import ops
dev = torch.device("cuda:0")
for nwin in (64, 512, 1024, 1920):
    raw, args = _block_operands(dev, nwin)
    x = raw["x"].to(dev)
    ref = _block_torch(raw, nwin)
    outs = [ops.fused_block(x.clone(), *args, tokens_per_wave=32).cpu() for _ in range(3)]
    b64 = ops.fused_block(x.clone(), *args, tokens_per_wave=64).cpu()
    print(nwin, "determinism", [(outs[0] - o).abs().max().item() for o in outs[1:]])
    e = (outs[0] - ref).abs()
    bad = (e > 2.5e-2).nonzero()
    print(nwin, "max err b32", e.max().item(), "b64", (b64 - ref).abs().max().item(), "n bad", len(bad))
    for r, c in bad[:20].tolist():
        print("   row", r, "win", r // 64, "tok", r % 64, "col", c, "b32", outs[0][r, c].item(), "b64", b64[r, c].item(), "ref", ref[r, c].item())
