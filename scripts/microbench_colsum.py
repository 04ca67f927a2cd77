"""Column sums at the training bench's two shapes: the 64-channel map (4 x 720 x 1280 pixels, bf16) and the token rows with a
row mask (61,440 x 192 fp32).  TUP_COLSUM_BLOCKS sets the workgroup target."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerupscaler_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
big = torch.randn(4 * 720 * 1280, 64, device="cuda", generator=g).bfloat16()
tok = torch.randn(61440, 192, device="cuda", generator=g)
mask = (torch.rand(61440, device="cuda", generator=g) > 0.06).to(torch.uint8)
def t(f):
    ts = []
    for r in range(12):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts[2:])[5]
print(f"target {os.environ.get('TUP_COLSUM_BLOCKS', 'default')}: map {t(lambda: ops.colsum(big)):.1f} us, masked tokens {t(lambda: ops.colsum(tok, rowmask=mask)):.1f} us")
