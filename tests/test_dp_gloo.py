"""Data-parallel reducer on CPU with gloo, world_size 2 (the kernels need a GPU, so the gradient
dictionaries are synthetic; what is under test is the host logic: arrival-order layout, bucketing,
async all-reduce, averaging, untouched inactive parameters)."""
import os

import pytest
import torch
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _fake_grads(names, shapes, rank, step):
    from _dp_worker import fake_grads
    return fake_grads(names, shapes, rank, step)


def test_reducer_world2_gloo(tmp_path):
    """Fixed-scale steps and a mixed-scale step (ranks on different scales) over gloo; file rendezvous (no port to race for)."""
    init = "file://" + str(tmp_path / "rdzv")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", init], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all(f"RANK{r} OK" in outs[r] for r in range(2)), outs


def test_reducer_single_process_layout():
    from transformerupscaler_amd.dp import GradReducer
    from transformerupscaler_amd.weights import active_param_names, param_shapes
    names = active_param_names(4)
    shapes = param_shapes()
    assert "up1.upsamplers.4.2.weight" in names and "up1.upsamplers.2.0.weight" not in names
    assert sum(torch.Size(shapes[n]).numel() for n in active_param_names(2)) == 4485743
    red = GradReducer(4, "cpu", bucket_mb=1.0)
    grads = _fake_grads(names, shapes, 0, 0)
    red.on_ready(list(reversed(names)), grads)
    out = red.finish()
    assert all(torch.equal(out[n], grads[n]) for n in names)          # world 1: identity
    assert red.launched_order == list(range(len(red.bucket_ranges)))
    late = "up1.upsamplers.2.0.weight"             # not part of scale 4: must raise, never be dropped silently
    with pytest.raises(RuntimeError, match="layout"):
        red.on_ready([late], {late: torch.zeros(shapes[late])})
    red._abort()
    with pytest.raises(RuntimeError, match="layout"):
        red.begin([late])
    # a backward that ends without one of its announced gradients is an error, not a silent zero
    red.begin(names)
    red.on_ready(names[:-1], grads)
    with pytest.raises(RuntimeError, match="without gradients"):
        red.finish()


def test_reducer_static_layout_is_backward_order():
    from transformerupscaler_amd.dp import GradReducer
    red = GradReducer(2, "cpu", bucket_mb=6.0)
    order = red.names
    assert order[0].startswith("final_upscale_conv") and order[-1].startswith("conv1.")
    assert order.index("window_blocks.5.mlp.2.weight") < order.index("window_blocks.0.mlp.2.weight") < order.index("patch_embed.weight")
    assert len(red.bucket_ranges) == 3 and all(b > a for a, b in red.bucket_ranges)
    mixed = GradReducer(None, "cpu", scales=(2, 3, 4, 6))
    # which parameters got a gradient on some rank is exchanged on the host (bitmask words), not through the flat buffer
    assert mixed.mixed and mixed.total_floats == mixed.param_floats and mixed._mask_words == (len(mixed.names) + 61) // 62
    assert mixed.bucket_ranges[-1][1] == mixed.total_floats
    from transformerupscaler_amd.weights import active_param_names
    mixed.begin(active_param_names(3))
    mixed.on_ready(active_param_names(3), {n: torch.zeros(mixed.shapes[n]) for n in active_param_names(3)})
    assert set(mixed.finish()) == set(active_param_names(3))                 # world 1: exactly what this rank produced
