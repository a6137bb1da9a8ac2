"""bench.py's final stdout line (the one the driver parses): built from canned result objects — the largest line this
repository has produced (profiles/r04_bench.json, 22 KB, which the round-4 driver could NOT parse) and an N = 8 shaped
one — it must stay under 6 KB, be strict JSON (no NaN / Infinity), and keep the contract's fields, `roofline` and
`cpu_baseline`.  The full object goes to a side file."""
import importlib.util
import io
import json
import os
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _strict(line):
    def boom(tok):
        raise AssertionError(f"non-finite constant {tok} in the line")
    return json.loads(line, parse_constant=boom)


def _canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        return json.load(f)


def test_line_is_short_strict_and_complete():
    b = _bench()
    full = _canned()
    assert len(json.dumps(full)) > 20000                      # the object that broke the round-4 record
    full["roofline"]["achieved"] = float("nan")               # a failed timing must not produce a bare NaN
    full["ingest"]["verified_blobs"] = 25600
    full["ingest"]["bit_exact_vs_cpu"] = True
    full["ingest"]["blobs_through_reference_tus"] = 3200
    line = b.compact_line(full)
    assert "\n" not in line and len(line) < 6144, len(line)
    out = _strict(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["roofline"]["achieved"] is None
    assert set(out["roofline"]) <= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launches",
                                    "sclk_MHz", "power_W", "limiter"}
    assert len(out["roofline"]["kernel"]) <= 80 and out["roofline"]["bound"] == "mfma" and "traffic" in out["roofline"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] == 16 and "sample" in out["cpu_baseline"]
    assert out["recall_at_k"] == 1.0 and out["bit_exact_vs_oracle"] is True and out["oracle_queries"] == 128
    ing = out["ingest"]
    assert ing["unit"] == "GB/s" and ing["verified_blobs"] == 25600 and ing["bit_exact_vs_cpu"] is True
    assert ing["blobs_through_reference_tus"] == 3200
    assert set(ing["roofline"]) >= {"bound", "achieved", "peak", "frac"} and ing["cpu_baseline"]["cores"] == 16
    assert "model" not in out["config"] and out["config"]["workload"].startswith("12500000x768")
    # nothing below the top level's objects is itself an object, except ingest's two
    for k, v in out.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                assert not isinstance(vv, (dict, list)) or (k == "ingest" and kk in ("roofline", "cpu_baseline")), (k, kk)


def test_n8_line_carries_the_exchange_fields():
    b = _bench()
    full = _canned()
    for k in ("ingest", "cpu_baseline", "boundary", "roofline_hbm_leg", "config3_l2", "config4_shard_q256"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["collective"] = {"backend": "nccl", "communicator_ranks": 8, "bytes_per_rank": 1232896, "collectives": 20, "batches": 20,
                          "exchange_ms": 0.41, "exchange_ms_max": 0.9, "launch_ms_min": 7.1, "launch_ms_max": 7.4, "fenced": True,
                          "watchdog_s": 30.0, "fence": "x" * 400, "per_rank_launch_ms": [7.1] * 8}
    out = _strict(b.compact_line(full))
    c = out["collective"]
    assert c["communicator_ranks"] == 8 and c["collectives"] == c["batches"] == 20 and c["exchange_ms"] == 0.41
    assert c["launch_ms_min"] == 7.1 and c["launch_ms_max"] == 7.4 and "fence" not in c and "per_rank_launch_ms" not in c


def test_emit_writes_the_side_file_and_prints_the_line_last(tmp_path):
    b = _bench()
    full = _canned()
    full["x"] = float("inf")
    side = tmp_path / "extra.json"
    so, se = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
        b.emit(full, str(side))
    lines = so.getvalue().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0]) < 6144
    assert not any(l.startswith("{") for l in se.getvalue().splitlines())       # the log copy never looks like the line
    extra = _strict(side.read_text())
    assert extra["x"] is None and "telemetry" in extra["roofline"] and "boundary" in extra
    assert _strict(lines[0])["extra"] == str(side)


def test_oversized_strings_cannot_push_the_line_over_the_limit():
    b = _bench()
    full = _canned()
    full["config"]["workload"] = "w" * 5000
    full["config"]["parallelism"] = "p" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["c_abi_sharded"] = {"error": "e" * 9000}
    line = b.compact_line(full)
    assert len(line) < 6144
    _strict(line)
