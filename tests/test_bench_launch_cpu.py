"""`python bench.py --gpus N` launches its own ranks (no GPU needed for these tests): the parent refuses more RCCL ranks
than visible devices with exit code 1 before it starts anything, spawns `torch.distributed.run` as a child otherwise,
passes its arguments on, and returns the ranks' exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e)


def test_more_rccl_ranks_than_devices_is_refused_by_the_parent():
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    r = _run("--gpus", str(n))
    assert r.returncode == 1
    assert "RCCL needs one device per rank" in r.stderr and r.stdout.strip() == ""


def test_plain_bench_gpus_2_spawns_two_ranks_and_prints_one_json_line():
    r = _run("--gpus", "2", "--dist-backend", "gloo-host", "--spawn-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"spawn_check": True, "world_size": 2, "sum_of_ranks": 1.0, "launched_by": "bench.py"}


def test_a_failing_rank_gives_the_parent_a_nonzero_exit_code():
    r = _run("--gpus", "2", "--dist-backend", "gloo-host", "--spawn-check", env={"I2R_SPAWN_CHECK_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_gpus_must_match_an_external_launchers_world_size():
    r = _run("--gpus", "2", "--spawn-check", env={"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE=4" in r.stderr
