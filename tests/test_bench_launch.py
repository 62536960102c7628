"""CPU: `python bench.py --gpus N` starts its own ranks when no launcher is around it — run
here at world size 2 on gloo with the stub renderer (READ_BENCH_STUB=1): one JSON line from rank 0, whole-job value, every rank
verified.  The torchrun form (the driver's) goes through the same main() and is covered by the same stub."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, READ_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 1 and rec["verified_ranks"] == [True, True]
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 5 / (rec["ms_per_step"] * 5e-3)) < 1e-6 * rec["value"]
    assert "without a launcher" in r.stderr


def test_bench_under_torchrun_keeps_working():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _one_json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["verified_ranks"] == [True, True]
    assert "without a launcher" not in r.stderr
