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


def test_bench_self_launches_n_ranks_and_checks_the_merge_against_the_oracle(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks itself and prints ONE
    line with n_gpus = 2; the merged top-k of the last timed step equals the oracle over the union
    of the shards (small shards here; --single-device/gloo because this box has one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--dist-backend", "gloo",
           "--rows-per-gpu", "300000", "--dim", "256", "--queries", "300", "--k", "50", "--steps", "3", "--warmup", "1",
           "--no-ingest", "--no-cpu-baseline", "--oracle-queries", "3", "--extra-json", str(tmp_path / "extra.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144 and r.stdout.rstrip().endswith(lines[0])     # the line is short and LAST
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["corpus_rows"] == 600000
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True and out["oracle_queries"] == 3
    assert out["exact_fallback_queries"] == 0 and out["roofline"]["launches"] == 3
    # what a scaling curve is attributed with: the exchange per batch and every rank's sweep time
    co = out["collective"]
    assert co["communicator_ranks"] == 2 and co["collectives"] == co["batches"] == 3 and co["watchdog_s"] == 30.0, co
    assert co["exchange_ms"] > 0 and co["launch_ms_min"] > 0 and co["launch_ms_max"] >= co["launch_ms_min"], co
    full = json.loads((tmp_path / "extra.json").read_text())
    assert [r_["rank"] for r_ in full["collective"]["per_rank"]] == [0, 1]
    out = full          # (the long form: everything the line summarises)
    # ... and rank 0's extra process drove both shards from ONE process through the C ABI (here: two shards on one
    # device, so device copies instead of the communicator) and got the very same merged result
    ca = out["c_abi_sharded"]
    assert "error" not in ca, ca
    assert ca["n_devices"] == 2 and ca["collective"] == "peer_copy" and ca["identical_to_the_timed_step"] is True
    assert ca["merged_equals_host_merge_of_per_shard_results"]["ok"] is True


def test_bench_through_the_c_abi_from_one_process_with_an_rccl_communicator(tmp_path):
    """`python bench.py --gpus 1 --via-c-abi`: ONE process, yams_scan_sharded_* with the communicator required (one
    rank here), submit/wait lanes, the contract's JSON line, oracle-checked."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--via-c-abi", "--rows-per-gpu", "300000", "--dim", "256",
           "--queries", "300", "--k", "50", "--steps", "4", "--warmup", "1", "--oracle-queries", "3",
           "--extra-json", str(tmp_path / "extra.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144
    short = json.loads(lines[0])
    assert short["c_abi_sharded"]["communicator_ranks"] == 1 and short["collective"]["collectives"] == short["collective"]["batches"] == 4
    assert short["collective"]["exchange_ms"] > 0 and short["roofline"]["traffic"] is None
    out = json.loads((tmp_path / "extra.json").read_text())
    ca = out["c_abi_sharded"]
    assert out["n_gpus"] == 1 and out["launcher"].startswith("single process") and out["unit"] == "QPS" and out["value"] > 0
    assert ca["collective"] == "rccl" and ca["communicator_ranks"] == 1 and ca["collectives"] >= 5 and ca["rccl_version"] > 20000
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True
    assert ca["merged_equals_host_merge_of_per_shard_results"]["ok"] is True


def test_bench_through_the_c_abi_with_eight_ranks_on_a_stand_in_collective(tmp_path):
    """The run the driver will launch unattended on an 8-GPU node, rehearsed on one: `bench.py --via-c-abi --gpus 8
    --single-device --rccl-library <tests/stub_coll>` — eight ranks of ONE communicator (the stand-in lets them share
    device 0), every batch one all-gather + merge.  The line must carry what a scaling curve is attributed with:
    communicator_ranks (ncclCommCount), collectives == batches, exchange_ms, per-rank launch_ms min / max, the deadline."""
    import _cpp_build
    stub = _cpp_build.build_stub_collective()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--via-c-abi", "--single-device", "--rccl-library", stub,
           "--rows-per-gpu", "120000", "--dim", "256", "--queries", "300", "--k", "50", "--steps", "6", "--warmup", "2",
           "--oracle-queries", "3", "--extra-json", str(tmp_path / "extra.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144 and r.stdout.rstrip().endswith(lines[0])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["corpus_rows"] == 960000 and out["value"] > 0
    co = out["collective"]
    assert co["communicator_ranks"] == 8 and co["collectives"] == co["batches"] == 6, co
    assert co["exchange_ms"] > 0 and co["exchange_ms_max"] >= co["exchange_ms"] and co["watchdog_s"] == 30.0, co
    assert 0 < co["launch_ms_min"] <= co["launch_ms_max"] and co["fenced"] is True, co
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True and out["oracle_queries"] == 3
    full = json.loads((tmp_path / "extra.json").read_text())
    ca = full["c_abi_sharded"]
    assert ca["collective"] == "rccl" and ca["communicator_ranks_source"] == "ncclCommCount" and len(ca["launch_ms_per_shard"]) == 8
    assert ca["merged_equals_host_merge_of_per_shard_results"]["ok"] is True
