"""bench.py --gpus N must start its own N ranks (the reference's initializeProcesses ... sumAcrossProcesses,
Drivers/monteCarloDriver.f95:187, 1151-1166).  Rehearsed here on CPU: --dry-run swaps RCCL for gloo and skips
the tracing, everything else (child launcher, rendezvous on 127.0.0.1, collective, ONE JSON line from rank 0)
is the code path of the real run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one JSON line, got: %r" % lines
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_reduces():
    out = _run("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["config"]["world_size"] == 2
    assert out["config"]["ranks_seen"] == [0, 1]
    assert out["config"]["photons_all_ranks_per_step"] == 2 * 10 ** 7  # the header of the reduced moment array
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"


def test_gpus_1_stays_in_process():
    out = _run("--dry-run")
    assert out["n_gpus"] == 1 and out["config"]["ranks_seen"] == [0]


def test_under_a_launcher_the_env_decides():
    """The driver's own launch line: torch.distributed.run starts the ranks, bench.py must not start more."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["config"]["ranks_seen"] == [0, 1]
