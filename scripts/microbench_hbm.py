"""HBM ceilings of this box for the three access mixes the HBM-class kernels have (1 GB tensors, torch kernels, event-timed):
write-only (fill), read-only (sum), read + write (copy).  What 'achievable' means for conv1 (write-dominated), decoder_conv2 /
patch_embed (read-dominated) and patch_unembed (1 : 1)."""
import torch
n = 256 * 1024 * 1024          # fp32 elements = 1 GiB
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
def t(f, reps=12):
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e-3)
    return sorted(ts[2:])[len(ts[2:]) // 2]
gb = n * 4 / 1e12
print(f"fill  (write 1 GiB)          {gb / t(lambda: a.fill_(1.0)):.2f} TB/s")
print(f"sum   (read 1 GiB)           {gb / t(lambda: a.sum()):.2f} TB/s")
print(f"copy  (read 1 GiB + write 1) {2 * gb / t(lambda: b.copy_(a)):.2f} TB/s")
print(f"add   (read 2 GiB + write 1) {3 * gb / t(lambda: torch.add(a, b, out=b)):.2f} TB/s")
