"""bench.py's multi-rank path: `bench.py --gpus 2` launched exactly as the driver launches it
(`python -m torch.distributed.run --nproc-per-node 2 ...`), but with both ranks on the ONE GPU of the test box and gloo carrying
the gradient buckets (COLDDIFF_DIST_BACKEND=gloo, COLDDIFF_SHARE_GPU=1: RCCL refuses two ranks per device).  Checks the JSON line
the driver parses: n_gpus, the global image count, value = images / max-over-ranks time, and the gradient-exchange record."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_two_ranks_json_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--no-sample", "--no-secondary", "--no-cpu-baseline"]
    env = dict(os.environ, COLDDIFF_DIST_BACKEND="gloo", COLDDIFF_SHARE_GPU="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_images_per_step"] == 2 * 8 * 2 and out["config"]["parallelism"] == "dp2"
    assert abs(out["value"] - out["config"]["global_images_per_step"] / (out["ms_per_step"] * 1e-3)) <= 0.01 * out["value"]
    gx = out["gradient_exchange"]
    assert gx["buckets"] >= 6 and gx["bytes_per_step"] >= 4 * 56_615_708 and gx["allreduce_ms_per_step"] > 0
    assert "roofline" in out and out["roofline"]["frac"] > 0


def test_bench_refuses_a_mismatched_world():
    """--gpus N must equal WORLD_SIZE (a bare `python bench.py --gpus 2` is a launch error, not a silent 1-GPU run)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_tail_ms_per_step_reads_the_committed_trace():
    """bench.py's `tail_ms_per_step` (the tail as a driver-visible number): classes from the newest committed
    rocprofv3 kernel trace -- quoted only when its source digest equals the running sources', every kernel in exactly one class."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    t = b.tail_ms_per_step()
    assert t is not None
    if t.get("stale"):                                       # (a tree whose kernels changed after the profile was taken: nothing is quoted)
        assert not t["provenance"]["match"]
        return
    cls = t["by_class_ms"]
    assert abs(sum(cls.values()) - t["kernel_ms"]) < 0.01
    assert abs(t["kernel_ms"] - cls["pre_split_fwd_dgrad_gemm"] - cls["pre_split_wgrad_gemm"] - t["tail_ms"]) < 0.01
    assert cls["other"] < 0.02 * t["kernel_ms"] and t["steps_in_profile"] >= 4 and t["dispatches_per_step"] > 100
