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


def mixed_scale_step(rank, world, shapes):
    """SURVEY 8(e) wrinkle: rank 0 trains scale 2 and rank 1 scale 3 in the same step.  Shared parameters average over both
    ranks, a scale's own upsamplers get (that rank's gradient) / world, the scales nobody trained get no gradient at all,
    and both ranks issue the same bucket sequence although their gradients arrive for different parameter sets."""
    from transformerupscaler_amd.dp import GradReducer
    from transformerupscaler_amd.weights import active_param_names
    red = GradReducer(None, "cpu", bucket_mb=2.0, scales=(2, 3, 4, 6))
    ok = True
    for step in range(2):
        my_scale = (2, 3)[(rank + step) % 2]
        names = active_param_names(my_scale)
        grads = fake_grads(names, shapes, rank, 100 + step)
        order = sorted(names, key=lambda n: red.offset[n])          # the backward's order = the layout's order
        red.begin(names)
        for i in range(0, len(order), 5):
            red.on_ready(order[i:i + 5], grads)
        out = red.finish()
        ok = ok and red.launched_order == list(range(len(red.bucket_ranges)))
        per_rank = []
        for r in range(world):
            sc = (2, 3)[(r + step) % 2]
            per_rank.append((set(active_param_names(sc)), fake_grads(active_param_names(sc), shapes, r, 100 + step)))
        expected = set().union(*[a for a, _ in per_rank])
        ok = ok and set(out) == expected
        for n in expected:
            ref = sum(g[n] for a, g in per_rank if n in a) / world
            ok = ok and torch.allclose(out[n], ref, atol=1e-6)
        ok = ok and not any(".upsamplers.4." in n or ".upsamplers.6." in n for n in out)
    # a gradient the layout does not know must raise, not be dropped
    fixed = GradReducer(2, "cpu")
    try:
        fixed.begin(active_param_names(3))
        ok = False
    except RuntimeError:
        pass
    return ok


def main():
    rank, world, init = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=init, rank=rank, world_size=world)
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
    ok = ok and len(red.bucket_ranges) >= 3 and red.bucket_ranges[-1][1] == red.param_floats
    ok = ok and red.launched_order == list(range(len(red.bucket_ranges)))
    ok = ok and mixed_scale_step(rank, world, shapes)
    dist.destroy_process_group()
    print(f"RANK{rank} {'OK' if ok else 'FAIL'} buckets={len(red.bucket_ranges)}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
