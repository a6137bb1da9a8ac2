"""N>1 path on CPU: world_size-2 and -4 gloo jobs through yams_amd/dist.py — shard bounds, the rank
launcher bench.py uses for `--gpus N`, rendezvous from the torchrun environment, the two-slot
all-gather + merge pipeline; merged result == single-shard oracle."""
import json
import os

import pytest

from yams_amd import dist as ydist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition_rows():
    for n in (0, 1, 7, 100_000_000):
        for w in (1, 2, 3, 8):
            b = ydist.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            assert all(b[i] <= b[i + 1] for i in range(w))
            assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1
    assert ydist.shard_bounds(100_000_000, 8)[1] == 12_500_000      # BASELINE config 4


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_search_matches_oracle(world, monkeypatch):
    monkeypatch.setenv("YAMS_DIST_TEST_CPU", "1")
    r = ydist.launch_ranks(os.path.join(ROOT, "tests", "_dist_worker.py"), world, [], capture=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["world"] == world and out["bounds"] == [3000 * g // world for g in range(world + 1)]
    # what the N > 1 bench line is built from: one collective per batch, every exchange timed
    ex = out["exchange"]
    assert ex["collectives"] == ex["exchanges"] == out["batches"] and 0 < ex["exchange_ms"] <= ex["exchange_ms_max"], ex


def test_deadline_fires_once_with_a_one_line_diagnosis_and_not_when_the_block_finishes():
    import time
    fired = []
    with ydist.Deadline("a block that finishes", 5.0, on_expire=fired.append):
        pass
    with ydist.Deadline("a collective nobody joins", 0.2, on_expire=fired.append, detail=lambda: {"exchanges_done": 7}):
        time.sleep(0.6)
    time.sleep(0.1)
    assert len(fired) == 1 and "\n" not in fired[0]
    assert "a collective nobody joins" in fired[0] and "exchanges_done" in fired[0] and "0 s" in fired[0]


def test_deadline_without_a_handler_exits_the_process_with_status_3():
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); from yams_amd import dist as d\n"
            "with d.Deadline('barrier', 0.2):\n    time.sleep(30)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "[yams_amd watchdog]" in r.stderr and "'barrier'" in r.stderr, (r.returncode, r.stderr)
