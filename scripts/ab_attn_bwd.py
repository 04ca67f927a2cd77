"""A/B of builds of window_attn_bwd_kernel in one process: python scripts/ab_attn_bwd.py name=lib.so[@slots] ..."""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
nwin = 960
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(nwin * 64, 576, device="cuda", generator=g) * 0.5).bfloat16()
gout = (torch.randn(nwin * 64, 192, device="cuda", generator=g) * 0.1).bfloat16()
bt = torch.randn(12, 4, 4, 64, 4, device="cuda", generator=g) * 0.2
bn = torch.randn(12, 4, 4, 64, 4, device="cuda", generator=g) * 0.2
P, I, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_uint
runs, outs = {}, {}
for a in sys.argv[1:]:
    name, path = a.split("=")
    path, _, slots = path.partition("@")
    if slots:
        os.environ["TUP_ATTN_BWD_SLOTS"] = slots          # read once per library at its first launch
    import shutil, tempfile
    cp = os.path.join(tempfile.mkdtemp(), name + ".so"); shutil.copy(os.path.join(root, path), cp)   # a fresh image per variant
    L = ctypes.CDLL(cp)
    gq = torch.empty(nwin * 64, 576, device="cuda", dtype=torch.bfloat16)
    db = torch.zeros(12, 4, 4, 64, 4, device="cuda")
    if hasattr(L, "tup_window_attn_bwd_scratch"):            # ABI 5: per-slot partials + sum kernel instead of atomics
        L.tup_window_attn_bwd.argtypes = [P, P, P, P, P, P, P, I, F, U, P]
        L.tup_window_attn_bwd_scratch.restype = ctypes.c_longlong
        sc = torch.empty(L.tup_window_attn_bwd_scratch(nwin, 12), device="cuda")
        f = lambda L=L, gq=gq, db=db, sc=sc: L.tup_window_attn_bwd(qkv.data_ptr(), gout.data_ptr(), bt.data_ptr(), bn.data_ptr(), gq.data_ptr(), db.data_ptr(), sc.data_ptr(), nwin, 0.1, 1234, None)
    else:
        L.tup_window_attn_bwd.argtypes = [P, P, P, P, P, P, I, F, U, P]
        f = lambda L=L, gq=gq, db=db: (db.zero_(), L.tup_window_attn_bwd(qkv.data_ptr(), gout.data_ptr(), bt.data_ptr(), bn.data_ptr(), gq.data_ptr(), db.data_ptr(), nwin, 0.1, 1234, None))[1]
    assert f() == 0
    torch.cuda.synchronize()
    runs[name], outs[name] = f, (gq, db)
ref = next(iter(outs.values()))
for k, (gq, db) in outs.items():
    print(k, "gqkv max diff", (gq.float() - ref[0].float()).abs().max().item(), "dbias rel", ((db - ref[1]).norm() / ref[1].norm()).item())
times = {k: [] for k in runs}
for r in range(11):
    for k, f in runs.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        times[k].append(s.elapsed_time(e) * 1e3)
for k, t in times.items():
    t = sorted(t)
    print(f"{k}: median {t[len(t) // 2]:.1f} us  min {t[0]:.1f} us")
