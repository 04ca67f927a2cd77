"""Worker for tests/test_dp_gloo.py: python _dp_worker.py RANK WORLD PORT (gloo, CPU)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_grads(names, shapes, rank, step):
    g = {}
    for i, n in enumerate(names):
        gen = torch.Generator().manual_seed(1000 * step + 10 * i + rank)
        g[n] = torch.rand(shapes[n], generator=gen) - 0.5
    return g


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformerupscaler_amd.dp import GradReducer
    from transformerupscaler_amd.weights import active_param_names, param_shapes
    shapes = param_shapes()
    names = active_param_names(2)
    red = GradReducer(2, "cpu", bucket_mb=2.0)
    ok = True
    for step in range(2):
        grads = fake_grads(names, shapes, rank, step)
        order = list(reversed(names))                      # backward order
        for i in range(0, len(order), 7):
            red.on_ready(order[i:i + 7], grads)
        out = red.finish()
        others = [fake_grads(names, shapes, r, step) for r in range(world)]
        ok = ok and set(out) == set(names)
        for n in names:
            ref = sum(o[n] for o in others) / world
            ok = ok and torch.allclose(out[n], ref, atol=1e-6)
    ok = ok and len(red.bucket_ranges) >= 3 and red.bucket_ranges[-1][1] == sum(red.numel.values())
    dist.destroy_process_group()
    print(f"RANK{rank} {'OK' if ok else 'FAIL'} buckets={len(red.bucket_ranges)}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
