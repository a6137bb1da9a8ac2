"""N>1 path on CPU: world_size-2 gloo job through yams_amd/dist.py (shard bounds, rendezvous from
the torchrun environment, all-gather of per-shard top-k, merge == single-shard oracle)."""
import json
import os
import subprocess
import sys

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


def test_world_size_2_gloo_sharded_search_matches_oracle():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["world"] == 2 and out["bounds"] == [0, 1500, 3000]
