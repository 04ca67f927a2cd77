"""bench.py's rank launcher: `python bench.py --gpus N` must start N ranks itself when it is not already one.
Host logic only; the dry run fails in the CHILD, at device selection (no GPU here), which is exactly what is asserted."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_decision():
    import bench
    assert bench.launcher_decision(1, {}) == "inline"
    assert bench.launcher_decision(2, {}) == "spawn"
    assert bench.launcher_decision(8, {}) == "spawn"
    # already a rank of a torchrun job (the driver's `python -m torch.distributed.run ... bench.py --gpus N` form): never re-spawn
    assert bench.launcher_decision(8, {"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == "inline"
    assert bench.launcher_decision(1, {"RANK": "0", "WORLD_SIZE": "1"}) == "inline"


def test_launcher_command_shape():
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_dry_gpus2_fails_only_in_the_child():
    """No GPU in this container: the parent must not touch the GPU (it would raise here), the two ranks must start and fail at
    device selection, and the parent must hand that failure on as its exit code without printing a JSON line."""
    env = dict(os.environ)
    env["TUP_BENCH_TEST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--mode", "infer"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    err = p.stderr
    # torchrun's failure report names the ranks it started and the script
    assert "bench.py" in err and ("ChildFailedError" in err or "exitcode" in err), err[-2000:]
    assert err.count("local_rank") >= 1 or "rank" in err
