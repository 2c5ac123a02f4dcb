"""`python bench.py --gpus N` forms its own ranks: the parent spawns N children with the torchrun
environment, relays rank 0's JSON line and returns the worst exit code.  Driven here with the CPU rehearsal mode (gloo,
no GPU in this container): launcher + rendezvous + barrier / MAX-over-ranks timing + the flat all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_self_launch_two_ranks_prints_one_json_line():
    p = run(["--gpus", "2", "--rehearse-cpu", "--steps", "5", "--warmup", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["allreduce_ok"] is True and out["config"]["parallelism"] == "dp2"
    assert out["value"] is None  # a rehearsal measures nothing


def test_under_an_external_launcher_no_second_level_of_ranks():
    """RANK / WORLD_SIZE already set (torch.distributed.run did it): bench.py must not spawn again."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = run(["--gpus", "1", "--rehearse-cpu", "--steps", "3", "--warmup", "1"], env_extra=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_a_failing_rank_fails_the_launcher():
    """No GPU here: the real (non-rehearsal) ranks die at device selection; the parent must report it, not hang."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present: ranks would run")
    p = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""


def test_a_rank_that_dies_after_the_rendezvous_ends_the_run_quickly():
    """the parent used to block on rank 0, which waits for its dead peer until the store / collective timeout.  Now
    the parent watches every child: rank 1 exits (code 3) after the rendezvous barrier, rank 0 sits in an all-reduce that can never
    complete; the launcher must stop it and return non-zero well within 30 s, with no JSON line."""
    import time

    t0 = time.time()
    p = run(["--gpus", "2", "--rehearse-cpu", "--steps", "5", "--warmup", "2"],
            env_extra={"HMP_BENCH_TEST_DIE_RANK": "1", "HMP_BENCH_FAIL_GRACE_S": "3"}, timeout=120)
    took = time.time() - t0
    assert p.returncode != 0, p.stdout
    assert p.stdout.strip() == ""
    assert "rank 1 exited with code 3" in p.stderr or "exited with code" in p.stderr, p.stderr[-1500:]
    assert took < 30.0, f"launcher needed {took:.1f} s to give up"
