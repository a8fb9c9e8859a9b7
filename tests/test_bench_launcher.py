"""bench.py --gpus N starts N ranks by itself: covered on the CPU with gloo and the oracle as the step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--steps", "3", "--warmup", "1",
                        "--seq", "512", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout      # rank 0 prints the one JSON line
    return json.loads(lines[0])


def test_two_ranks_spawned_weak_scaling():
    out = _run(["--gpus", "2", "--batch", "8"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["rows_per_gpu"] == 8
    # whole-job rate: both ranks' rows over the slowest rank's time
    assert abs(out["value"] - 2 * 8 * 512 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    # every rank's own time beside the reported maximum, and the world torch.distributed itself reports
    r = out["ms_per_step_ranks"]
    assert len(r["per_rank"]) == 2 and r["max"] == max(r["per_rank"]) and abs(r["max"] - out["ms_per_step"]) < 1e-9
    assert out["world"] == {"env_world_size": 2, "torch_distributed_world_size": 2, "backend": "gloo"}


def test_two_ranks_rows_sharded():
    out = _run(["--gpus", "2", "--total-rows", "16"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["rows_per_gpu"] == 8


def test_single_rank_stays_in_process():
    out = _run(["--gpus", "1", "--batch", "4"])
    assert out["n_gpus"] == 1 and out["config"]["rows_per_gpu"] == 4
