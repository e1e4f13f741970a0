"""`python bench.py --gpus 2` with no launcher around it (the form the driver uses for N = 1, and what a user types): bench.py starts
the ranks itself as child processes, rank 0 prints the one JSON line, the exit code is the children's.  Exercised on CPU with the fake
workload over gloo (world 2): launcher, rendezvous on 127.0.0.1, barrier + max-over-ranks timing, pipelined steps + flush, the exchange."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_bench_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--workload", "fake"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16
    assert d["detail"]["checksum"] == 8 * (1.0 + 2.0)     # both ranks' payloads arrived through the all-gather
    assert d["detail"]["steps_seen"] == 4                  # warmup + steps, nothing skipped or repeated
    assert d["value"] > 0 and abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_single_rank_runs_in_process():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--workload", "fake"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["detail"]["checksum"] == 4.0
    assert "launching" not in r.stderr                    # no child launcher for one rank


def test_bench_under_an_outer_launcher_does_not_relaunch():
    """the driver's N > 1 form: torch.distributed.run sets WORLD_SIZE - bench.py must join that world, not start another"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--workload", "fake"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    assert "launching" not in r.stderr
