"""N>1 path on ONE GPU: the ranks of a world_size-2/4 job share cuda:0 (gloo carries the collective
on the host — RCCL refuses two ranks on one device), everything else is the product path: the HIP
scan per shard, the two-slot gather pipeline, the product's merge_topk_kernel behind the collective;
and bench.py's own `--gpus N` self-launch with its oracle check of the merged top-k."""
import json
import os
import subprocess
import sys

import pytest

from yams_amd import dist as ydist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_search_with_the_merge_kernel_behind_the_collective(world):
    r = ydist.launch_ranks(os.path.join(ROOT, "tests", "_dist_worker.py"), world, [], capture=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == world and out["gpu"] and out["merge"] == "merge_topk_kernel", r.stderr[-3000:]


def test_rccl_with_one_rank_runs_the_collective_and_the_merge(monkeypatch):
    """What a one-GPU box can run of RCCL: a process group of ONE rank on backend "nccl" (= RCCL), through the
    product pipeline — communicator init under HSA_ENABLE_IPC_MODE_LEGACY=0, all_gather_into_tensor of the packed
    record on the pipeline's side stream, the merge kernel behind it, an all_reduce and a barrier — with the merged
    top-k checked against the oracle.  Two ranks on one device are refused by RCCL, so the multi-rank collective
    itself remains unexecuted here."""
    monkeypatch.setenv("YAMS_DIST_TEST_BACKEND", "nccl")
    r = ydist.launch_ranks(os.path.join(ROOT, "tests", "_dist_worker.py"), 1, [], capture=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["world"] == 1 and out["gpu"] and out["backend"] == "nccl" and out["merge"] == "merge_topk_kernel"


def test_bench_self_launches_n_ranks_and_checks_the_merge_against_the_oracle():
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks itself and prints ONE
    line with n_gpus = 2; the merged top-k of the last timed step equals the oracle over the union
    of the shards (small shards here; --single-device/gloo because this box has one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--dist-backend", "gloo",
           "--rows-per-gpu", "300000", "--dim", "256", "--queries", "300", "--k", "50", "--steps", "3", "--warmup", "1",
           "--no-ingest", "--no-cpu-baseline", "--oracle-queries", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["corpus_rows"] == 600000
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True and out["oracle_queries"] == 3
    assert out["exact_fallback_queries"] == 0 and out["roofline"]["launches"] == 3
    # ... and rank 0's extra process drove both shards from ONE process through the C ABI (here: two shards on one
    # device, so device copies instead of the communicator) and got the very same merged result
    ca = out["c_abi_sharded"]
    assert "error" not in ca, ca
    assert ca["n_devices"] == 2 and ca["collective"] == "peer_copy" and ca["identical_to_the_timed_step"] is True
    assert ca["merged_equals_host_merge_of_per_shard_results"]["ok"] is True


def test_bench_through_the_c_abi_from_one_process_with_an_rccl_communicator():
    """`python bench.py --gpus 1 --via-c-abi`: ONE process, yams_scan_sharded_* with the communicator required (one
    rank here), submit/wait lanes, the contract's JSON line, oracle-checked."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--via-c-abi", "--rows-per-gpu", "300000", "--dim", "256",
           "--queries", "300", "--k", "50", "--steps", "4", "--warmup", "1", "--oracle-queries", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    ca = out["c_abi_sharded"]
    assert out["n_gpus"] == 1 and out["launcher"].startswith("single process") and out["unit"] == "QPS" and out["value"] > 0
    assert ca["collective"] == "rccl" and ca["communicator_ranks"] == 1 and ca["collectives"] >= 5 and ca["rccl_version"] > 20000
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True
    assert ca["merged_equals_host_merge_of_per_shard_results"]["ok"] is True
