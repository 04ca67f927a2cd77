"""Where the HOST time of a training step goes (cProfile over 8 steps issued back to back, no sync inside): the FastTransformer
step issues ~275 launches and its issue time is within 10 % of its GPU time, so Python overhead per launch decides whether a
slow-CPU box is host-bound.    python scripts/host_profile.py [ft|rt]"""
import cProfile, importlib, io, os, pstats, sys, time
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from transformerupscaler_amd import harness
from transformerupscaler_amd.weights import deterministic_state_dict, rt_deterministic_state_dict
which = sys.argv[1] if len(sys.argv) > 1 else "ft"
dev = "cuda"
g = torch.Generator().manual_seed(1)
if which == "ft":
    m = importlib.import_module("models.FastTransformer.model").TransformerModel()
    m.load_state_dict(deterministic_state_dict(0), strict=False)
    lr = torch.rand((4, 3, 720, 1280), generator=g).to(dev); hr = torch.rand((4, 3, 1080, 1920), generator=g).to(dev)
    step = lambda: harness.train_step(m, opt, lr, hr)
else:
    from transformerupscaler_amd.autograd import l1_loss
    m = importlib.import_module("models.ResidualTransformer.model").TransformerModel()
    m.load_state_dict(rt_deterministic_state_dict(0))
    lr = torch.rand((2, 3, 720, 1280), generator=g).to(dev); hr = torch.rand((2, 3, 4320, 7680), generator=g).to(dev)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = l1_loss(m(lr, upscale_factor=6), hr, fuse_into_model_backward=True)
        loss.backward(); opt.step()
        return loss.detach()
m = m.to(dev).train()
opt = harness.make_optimizer(m, 1e-4)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{which}: issue {1e3 * (t1 - t0) / 8:.2f} ms per step, with sync {1e3 * (t2 - t0) / 8:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
