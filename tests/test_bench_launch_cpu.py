"""bench.py's launch path on CPU (no GPU is touched: BSLAM_BENCH_DRY_RUN=1 stops every rank before it imports the library).

`python bench.py --gpus N` started WITHOUT a launcher must start its own N ranks and report the size of the group that actually
formed; a rank that dies must fail the whole run; a --gpus / WORLD_SIZE mismatch must fail loudly instead of measuring a
different number of ranks than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(BSLAM_BENCH_DRY_RUN="1", **env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_bare_invocation_spawns_its_own_ranks():
    r = run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 2


def test_single_rank_runs_in_process():
    r = run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_failing_rank_fails_the_run():
    r = run(["--gpus", "2"], BSLAM_BENCH_DRY_FAIL_RANK="1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line when a rank failed"
    assert "rank 1 of 2 exited with code 3" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_more_ranks_than_gpus_is_refused_without_touching_a_gpu():
    """Not a dry run: the parent counts the visible devices (none in the build container, one on the GPU box) and refuses to
    start N > devices ranks on the RCCL path -- instead of measuring one rank and printing n_gpus: 1."""
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BSLAM_BENCH_DRY_RUN", "BSLAM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 64 GPUs" in r.stderr
    assert not r.stdout.strip()
