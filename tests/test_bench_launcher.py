"""`python bench.py --gpus N` must start its own ranks (the driver's SCALE command is the bare form; the reference spawns one process
per GPU itself, yolo/main.py:38-42).  CPU rehearsal: the launcher path with a gloo group and no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *extra], env=env, capture_output=True, text=True,
                          timeout=300)


def test_bare_invocation_spawns_ranks_and_relays_rank0_line():
    r = _run("--launcher-selftest", "ok")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines == [{"launcher_selftest": True, "n_gpus": 2, "rank_sum": 1.0}]


def test_failed_rank_fails_the_launcher():
    r = _run("--launcher-selftest", "fail")
    assert r.returncode != 0


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest", "ok"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
