"""Where the whole-block kernel's in-model time differs from its harness time (VERDICT r2 #1d): the one-launch six-block kernel,
1,920 windows, event-timed in ONE process, interleaved rounds, on
  shared    the harness operands of scripts/ab_block.py: random x, ONE weight set used for all six blocks
  distinct  random x, six different random weight sets (5.3 MB of weights per launch instead of 0.9 MB)
  model-w   random x, the model's six packed weight sets
  model     the model's own token stream (patch_embed of the bench input) and weights = what bench.py's `blocks` stage runs
    python scripts/ab_block_model.py [lib.so]"""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
if len(sys.argv) > 1:
    os.environ["TUP_LIB_PATH"] = os.path.join(root, sys.argv[1])
from transformerupscaler_amd import ops
from transformerupscaler_amd.weights import deterministic_state_dict, BLOCKS
import importlib
import test_hip_kernels as T

nwin, rounds = 1920, 12
dev = "cuda"
raw, args = T._block_operands(dev, nwin)
x_rand = raw["x"].to(dev)
tabs = {}
tabs["shared"] = (x_rand, ops.block_table([tuple(args)] * 6))
sets = [T._block_operands(dev, 1, seed=100 + i)[1] for i in range(6)]
tabs["distinct"] = (x_rand, ops.block_table([tuple(a) for a in sets]))
m = importlib.import_module("models.FastTransformer.model").TransformerModel()
m.load_state_dict(deterministic_state_dict(0), strict=False)
m = m.to(dev).eval()
pk, frags = m.packed(2)
from transformerupscaler_amd.engine import _block_operands
mt = ops.block_table([_block_operands(pk, i, frags) for i in range(BLOCKS)])
tabs["model-w"] = (x_rand, mt)
g = torch.Generator().manual_seed(1234)
img = torch.rand((8, 3, 720, 1280), generator=g).to(dev)
with torch.no_grad():
    feat = ops.conv_c64(ops.conv1(img, pk["conv1.w"], pk["conv1.b"], relu=True), pk["conv2.w"], pk["conv2.b"], 1, relu=True)
    xw = ops.patch_embed(feat, pk["pe.w"], pk["pe.b"])
assert xw.shape[0] == nwin * 64
tabs["model"] = (xw, mt)
print("x_rand |.| mean", x_rand.abs().mean().item(), " model tokens |.| mean", xw.abs().mean().item(), "max", xw.abs().max().item())
times = {k: [] for k in tabs}
x = torch.empty_like(x_rand)
for r in range(rounds + 2):
    for k, (x0, tab) in tabs.items():
        x.copy_(x0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.fused_blocks32(x, tab); e.record(); torch.cuda.synchronize()
        if r >= 2:
            times[k].append(s.elapsed_time(e) / 6 * 1e3)
for k, t in times.items():
    t = sorted(t)
    print(f"{k:9s}: median {t[len(t) // 2]:.1f} us  min {t[0]:.1f} us per block")
