"""Timing of the ResidualTransformer inference path (BASELINE.json config 5 geometry: 720x1280 -> x6) on one MI355X."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd.weights import rt_deterministic_state_dict
m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
m.load_state_dict(rt_deterministic_state_dict(0)); m = m.cuda().eval()
CONFIGS = ((2, dict(upscale_factor=6)), (8, dict(res_out=(1080, 1920))))
if len(sys.argv) > 1:
    CONFIGS = (CONFIGS[int(sys.argv[1])],)
for B, kw in CONFIGS:
    x = torch.rand((B, 3, 720, 1280)).cuda()
    with torch.no_grad():
        for _ in range(3): m(x, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): y = m(x, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"RT B={B} {kw}: {dt * 1e3:.2f} ms/step, {B / dt:.1f} images/s, out {tuple(y.shape)}", flush=True)
