"""bench.py's own N > 1 launcher (no GPU needed): `python bench.py --gpus N` from a bare shell starts N ranks itself,
hands back rank 0's JSON line and exits with the children's status (it used to refuse)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASLP_COMM_FILE", "ASLP_COMM_TOKEN")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run-ranks"], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


def test_bare_shell_launch_starts_n_ranks_and_returns_rank0_line():
    p = run(3)
    assert p.returncode == 0, p.stderr.decode()
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["dry_run"]
    assert d["comm_file"] and d["token"] and d["token"] in d["comm_file"]   # a per-launch rendezvous file
    assert not os.path.exists(d["comm_file"])


def test_a_failing_rank_fails_the_launch():
    p = run(2, {"ASLP_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert p.returncode == 3


def test_launcher_environment_is_respected():
    """under torch.distributed.run (WORLD_SIZE set) bench.py is one rank and must not start children"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-ranks"], env=env, stdout=subprocess.PIPE, timeout=60)
    assert p.returncode == 0 and json.loads(p.stdout.decode())["n_gpus"] == 2


def test_eight_ranks_the_drivers_scale_run():
    """N = 8 (the SCALE run's largest): eight children, one line, every child's status collected"""
    p = run(8)
    assert p.returncode == 0, p.stderr.decode()
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 8 and d["dry_run"]


def test_a_communicator_that_does_not_come_up_ends_the_run_with_its_error_string():
    """what a failed ncclCommInitRank does to `bench.py --gpus 8`: the rank says so on stderr with the transport's error string and exits 4,
    the launcher returns non-zero and prints no bench line -- no retry, no re-exec"""
    msg = "ncclCommInitRank: unhandled system error (simulated)"
    p = run(8, {"ASLP_BENCH_DRYRUN_COMM_ERROR": msg, "ASLP_BENCH_DRYRUN_COMM_ERROR_RANK": "5"})
    assert p.returncode == 4
    assert ("rank 5 of 8: communicator did not come up: " + msg) in p.stderr.decode()
    # rank 0 failing: no line on stdout at all
    p = run(8, {"ASLP_BENCH_DRYRUN_COMM_ERROR": msg, "ASLP_BENCH_DRYRUN_COMM_ERROR_RANK": "0"})
    assert p.returncode == 4 and p.stdout.decode().strip() == ""
