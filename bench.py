#!/usr/bin/env python
"""bench.py — the hot path measured the way BASELINE.json asks.

Metric: k-NN QPS (+ recall@k) for batched exact cosine top-100 over fp32 embeddings, corpus
row-sharded over the GPUs of one node; ingest GB/s (SHA-256 + CDC) reported alongside.

Workload per GPU (weak scaling in corpus size): one 12.5M x 768 fp32 row shard of
BASELINE config 4 (100M x 768 cosine top-100 over 8 GPUs), query batch 1024 — at N GPUs the job
searches N x 12.5M rows; N = 8 is the headline configuration.  A "step" = one query batch: every
rank scans its shard, one RCCL all-gather moves the per-shard top-k, every rank merges (collective +
merge of batch i run on a side stream under the sweep of batch i+1).
`value` is the whole-job rate in the unit of the metric — queries/s against the 100M x 768 headline
corpus: (rows scored x queries) / s / 1e8.  At N = 8 the job holds exactly those 100M rows and
`value` is its measured QPS (batch / step time); at N < 8 the GPUs hold N/8 of the corpus and the
same aggregate rate counts for N/8 of a headline query, so `value` grows with N when per-GPU work
is fixed (weak scaling).  `qps_on_resident_corpus` is batch / step time on the rows actually
resident.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE: starts N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: bf16 MFMA dense peak (NOT the 2:1-sparse 5 PF)
PEAK_I8_MFMA_TOPS = 5000.0     # int8 MFMA dense = 2x the bf16 rate (MI355X_MICROARCH.md: "I8 ... ~2x bf16 rate"; 10 POPS is the sparse figure)
BF16_PASSES = 3                # hi*hi + hi*lo + lo*hi per algorithmic multiply-add
PEAK_HBM_GBPS = 8000.0
INT_VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4   # 256 CUs x 4 SIMDs, one 32-bit integer wave-instruction per 4 cycles
HEADLINE_ROWS = 100_000_000     # BASELINE.json metric: 100M x 768 (= 8 shards of the default --rows-per-gpu)
LINE_LIMIT = 6144               # the driver keeps the last 8 KB of stdout: the final line stays well under it


def _clean(x):
    """JSON-strict copy: NaN / inf -> None, numpy scalars -> Python numbers, floats to 6 significant digits."""
    import math
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if hasattr(x, "item") and not isinstance(x, float):
        try:
            return _clean(x.item())
        except Exception:          # noqa: BLE001
            return str(x)
    if isinstance(x, float):
        return float(f"{x:.6g}") if math.isfinite(x) else None
    return str(x)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _short(s, n=80):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1].rstrip() + "~"


def compact_line(out):
    """The ONE line the driver parses: the contract's fields + `roofline` + `cpu_baseline` + the parity verdicts,
    scalars only below the top level's few objects, always shorter than LINE_LIMIT.  Everything else a run learns
    (telemetry windows, the boundary / breadth / L2 legs, per-query ids) lives in bench_extra.json."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "qps_on_resident_corpus", "exact_fallback_queries", "launcher",
                    "recall_at_k", "bit_exact_vs_oracle", "oracle_queries", "oracle_seconds", "extra"))
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("workload", "rows_per_gpu", "corpus_rows", "dim", "k", "query_batch", "parallelism", "search_lanes"))
    for kk in ("workload", "parallelism"):
        if kk in c["config"]:
            c["config"][kk] = _short(c["config"][kk], 160)
    r = out.get("roofline") or {}
    c["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launches",
                              "sclk_MHz", "power_W", "limiter"))
    if "kernel" in c["roofline"]:
        c["roofline"]["kernel"] = _short(c["roofline"]["kernel"], 80)
    c["roofline"].setdefault("traffic", None)
    b = out.get("cpu_baseline")
    if b:
        c["cpu_baseline"] = _pick(b, ("value", "unit", "cores", "kind", "sample", "sample_rows", "sample_queries", "sample_seconds",
                                      "oracle_agrees_on_the_sample"))
        c["cpu_baseline"]["sample"] = _short(c["cpu_baseline"].get("sample"), 260)
        st = b.get("single_thread") or {}
        if "value" in st:
            c["cpu_baseline"]["single_thread_value"] = st["value"]
    h = out.get("roofline_hbm_leg")
    if h:
        c["roofline_hbm_leg"] = _pick(h, ("bound", "queries", "achieved", "peak", "unit", "frac", "traffic", "launch_ms"))
    l2 = out.get("config3_l2")
    if l2:
        c["config3_l2"] = _pick(l2, ("ms_per_step", "qps", "launch_ms", "frac", "results_identical_to_the_bf16_tier", "parity"))
    c2 = out.get("config2")
    if c2:
        c["config2"] = _pick(c2, ("ms_per_step", "qps", "search_lanes", "launch_ms", "bound", "achieved", "peak", "unit", "frac",
                                  "filter_candidates", "exact_fallback_queries", "results_identical_to_the_oracle_run", "oracle_queries", "error"))
        if "error" in c["config2"]:
            c["config2"]["error"] = _short(c["config2"]["error"], 200)
    pqv = out.get("pq")
    if pqv:
        c["pq"] = _pick(pqv, ("ms_per_step", "qps", "adc_launch_ms", "achieved", "frac", "lds_TBps", "lds_frac", "identical_to_the_restated_oracle", "error"))
        if "error" in c["pq"]:
            c["pq"]["error"] = _short(c["pq"]["error"], 200)
    nu = out.get("non_uniform")
    if nu:      # one summary object: step time per distribution, the proof outcomes, the oracle verdicts
        c["non_uniform"] = {}
        for kd, v in nu.items():
            if not isinstance(v, dict):
                continue
            if "error" in v:
                c["non_uniform"][kd + "_error"] = _short(v["error"], 120)
                continue
            c["non_uniform"][kd + "_ms_per_step"] = v.get("ms_per_step")
            esc = v.get("escalated_queries")
            c["non_uniform"][kd + "_escalated_queries"] = (sum(esc) / len(esc)) if isinstance(esc, list) and esc else esc
            c["non_uniform"][kd + "_bit_exact"] = v.get("bit_exact_vs_oracle")
            c["non_uniform"][kd + "_filter_tier"] = v.get("filter_tier")
            c["non_uniform"][kd + "_i8_layout"] = v.get("i8_layout")
    ca = out.get("c_abi_sharded")
    if ca:
        c["c_abi_sharded"] = _pick(ca, ("n_devices", "collective", "communicator_ranks", "collectives", "batches", "ms_per_step", "value",
                                        "exchange_ms", "launch_ms_min", "launch_ms_max", "identical_to_the_timed_step", "error"))
        if "error" in c["c_abi_sharded"]:
            c["c_abi_sharded"]["error"] = _short(c["c_abi_sharded"]["error"], 200)
        m = ca.get("merged_equals_host_merge_of_per_shard_results")
        if isinstance(m, dict):
            c["c_abi_sharded"]["merge_check_ok"] = m.get("ok")
    co = out.get("collective")
    if co:
        c["collective"] = _pick(co, ("backend", "communicator_ranks", "bytes_per_rank", "collectives", "batches", "exchange_ms",
                                     "exchange_ms_max", "launch_ms_min", "launch_ms_max", "fenced", "watchdog_s"))
    ing = out.get("ingest")
    if ing:
        ci = _pick(ing, ("value", "unit", "bytes", "blobs", "chunks", "ms", "verified_blobs", "blobs_through_reference_tus",
                         "bit_exact_vs_cpu", "error"))
        ci["roofline"] = _pick(ing.get("roofline") or {}, ("bound", "achieved", "peak", "unit", "frac", "sclk_MHz", "limiter"))
        ci["cpu_baseline"] = _pick(ing.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind", "sample_seconds"))
        c["ingest"] = ci
    c = _clean(c)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:            # cannot happen with the fields above; keep the contract whatever happens
        for kk in ("c_abi_sharded", "config3_l2", "roofline_hbm_leg", "collective"):
            c.pop(kk, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


_REAL_STDOUT = None


def quiet_stdout():
    """File descriptor 1 points at stderr from here on: whatever native libraries print on stdout (RCCL's version banner
    sits in C stdio's buffer until the process exits — it used to land AFTER the JSON line) goes to the log, and
    `emit` alone writes to the real stdout.  Every rank calls it; rank 0 emits."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _print_on_real_stdout(line):
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)         # C stdio buffers of this process: out through the redirected descriptor first
    except Exception:                           # noqa: BLE001
        pass
    if _REAL_STDOUT is None:
        print(line, flush=True)
        return
    os.write(_REAL_STDOUT, (line + "\n").encode())


def emit(out, extra_path=None):
    """Write the full result object to bench_extra.json (and gpurun_out/ when that scratch directory exists), put it
    on stderr for the log, and print the compact line LAST on stdout."""
    full = _clean(out)
    paths = [extra_path or os.path.join(ROOT, "bench_extra.json")]
    if extra_path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_extra.json"))
    written = None
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(full, f, allow_nan=False, indent=1)
            written = written or p_
        except OSError:
            pass
    out = dict(out)
    out["extra"] = os.path.relpath(written, ROOT) if written and written.startswith(ROOT) else written
    sys.stderr.write("bench_extra: " + json.dumps(full, allow_nan=False) + "\n")
    sys.stderr.flush()
    _print_on_real_stdout(compact_line(out))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows-per-gpu", type=int, default=12_500_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--no-hbm-leg", action="store_true")
    ap.add_argument("--no-l2-leg", action="store_true", help="skip the BASELINE config 3 (10M x 768 L2) leg")
    ap.add_argument("--oracle-queries", type=int, default=None, help="queries checked against the oracle over the whole resident corpus (default: the WHOLE timed batch at N=1 — 1024 queries, ~46 s on 16 host threads — and 16 at N>1; the batched oracle drivers make a query cost ~0.1 thread-seconds per million rows; 0 = skip)")
    ap.add_argument("--ingest-gib", type=float, default=100.0)  # BASELINE config 5
    ap.add_argument("--verify-all-ingest", action="store_true", default=True,
                    help="(the default) check EVERY blob of the ingest leg's timed call on the CPU: boundaries, every chunk digest, blob "
                         "digest; one in eight also through oracle/_ref — ~40 s on 16 host threads at 100 GiB")
    ap.add_argument("--verify-sample-ingest", dest="verify_all_ingest", action="store_false",
                    help="check a spread of 64 blobs of the ingest leg instead of all of them")
    ap.add_argument("--distribution", default=None, help="comma-separated corpora beside the uniform headline: clustered, anisotropic, gaussian (a throughput leg each on the headline shape; 64 oracle queries each)")
    ap.add_argument("--only-distribution", action="store_true", help="run the --distribution legs alone")
    ap.add_argument("--dist-metric", default="cosine", choices=["cosine", "l2"], help="metric of the distribution legs (measurement)")
    ap.add_argument("--dist-i8-layout", default="auto", choices=["auto", "plain", "rotated"], help="int8 shadow layout of the distribution legs (auto: yams_scan_choose_i8_layout_device decides)")
    ap.add_argument("--dist-flags", type=int, default=0, help="YAMS_SCAN_FLAG_* bits of the distribution legs' searches (measurement: 64 = no int8 tier)")
    ap.add_argument("--no-distribution-legs", action="store_true", help="skip the clustered / anisotropic legs of the default run")
    ap.add_argument("--no-config2-leg", action="store_true", help="skip the BASELINE config 2 (1M x 384, Q = 256) leg")
    ap.add_argument("--only-pq", action="store_true", help="run the product-quantised engine's leg alone")
    ap.add_argument("--no-pq-leg", action="store_true", help="skip the product-quantised engine's leg of the default run")
    ap.add_argument("--only-config2", action="store_true", help="run the BASELINE config 2 leg alone (profiling)")
    ap.add_argument("--config2-lanes", type=int, default=4, help="search lanes of the config 2 leg")
    ap.add_argument("--config2-lane-sweep", default="2,4", help="comma-separated lane counts to time in the config 2 leg (the best is reported)")
    ap.add_argument("--config2-no-gate", action="store_true", help="config 2 leg: no sweep gate between the lanes (measurement)")
    ap.add_argument("--config2-batches", type=int, default=None)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--f32-filter", action="store_true", help="use the exact-f32 MFMA filter kernel")
    ap.add_argument("--no-shadow", action="store_true", help="bare fp32 corpus view: the single-pass filter converts rows in its loop")
    ap.add_argument("--lanes", type=int, default=2, help="search lanes: batches in flight, each on its own context / stream / host thread")
    ap.add_argument("--no-i8", action="store_true", help="no int8 shadow: the bf16 tier filters the large batches too")
    ap.add_argument("--split-filter", action="store_true", help="start with the split-bf16 (3-pass) filter instead of the single-pass bf16 one")
    ap.add_argument("--half-tile", action="store_true", help="int8 tier: keep the per-tile (half-tile) filter kernel instead of the resident-query form (A/B runs)")
    ap.add_argument("--via-c-abi", action="store_true",
                    help="ONE process drives all --gpus devices through yams_scan_sharded_* (the C ABI a C++ host calls: one RCCL "
                         "communicator, all-gather + merge per batch, submit/wait lanes) and prints the same JSON line")
    ap.add_argument("--no-telemetry", action="store_true", help="skip the clock / power / throttle-reason window")
    ap.add_argument("--no-c-abi-leg", action="store_true", help="skip the C-ABI sharded leg of the default run")
    ap.add_argument("--no-boundary-leg", action="store_true", help="skip the plugin-door (vector_scan_v1.search_batch from host memory) leg")
    ap.add_argument("--rccl-library", default=None,
                    help="--via-c-abi: the collective library the sharded handle binds instead of librccl.so.1 (the test "
                         "suite's stand-in, tests/stub_coll, lets --single-device runs take the all-gather path)")
    ap.add_argument("--no-fence", action="store_true", help="--via-c-abi: lift the exchange fence (measurement only)")
    ap.add_argument("--child-json", action="store_true", help=argparse.SUPPRESS)   # --via-c-abi as the child of a torchrun rank 0
    ap.add_argument("--query-batches", type=int, default=4, help="distinct query batches rotated through the steps")
    # dry-run aids (NOT the contract): run the N>1 code path on a box with one GPU
    ap.add_argument("--dist-backend", default=None, help="override the collective backend (gloo for dry runs)")
    ap.add_argument("--single-device", action="store_true", help="all ranks use cuda:0 (dry runs only)")
    ap.add_argument("--watchdog-s", type=float, default=30.0,
                    help="N > 1: an exchange (all-gather + merge) that has not completed after this many seconds aborts the run "
                         "with a one-line diagnosis and exit status 3")
    ap.add_argument("--extra-json", default=None, help="where the full result object goes (default: bench_extra.json beside this file)")
    return ap.parse_args()


def self_launch(a) -> int:
    """`python bench.py --gpus N` with no torchrun environment: start N ranks (one per GPU) under
    torch.distributed.run on this node and pass their output through; rank 0 prints the JSON line."""
    from yams_amd import dist as ydist
    return ydist.launch_ranks(os.path.abspath(__file__), a.gpus, sys.argv[1:]).returncode


def oracle_mod():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    return _oracle


def cpu_baseline_scan(tc, tq, rows_total, k, seed_rows=1_250_000):
    """The oracle (scalar fp64 restatement of sqlite_vec_backend.cpp:4204-4331) timed on this box's
    host cores on a bounded, host-resident sample of the same workload: single-thread (how the
    reference runs a query) and one query per thread on all cores."""
    from concurrent.futures import ThreadPoolExecutor
    _oracle = oracle_mod()
    o = _oracle.oracle()
    n_s = min(seed_rows, rows_total)
    corpus = tc[:n_s].cpu().numpy()
    threads = min(_oracle.host_threads(), tq.shape[0])
    per_thread = 2                                     # ~20 CPU-seconds of scalar scan on a 16-thread box
    n_par = min(threads * per_thread, tq.shape[0])
    queries = tq[:max(n_par, 4)].cpu().numpy()
    n1 = 4
    t0 = time.perf_counter()
    for i in range(n1):
        o.scan_cosine(corpus, queries[i], k, -1.0)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda i: o.scan_cosine(corpus, queries[i], k, -1.0), range(n_par)))
    dtn = time.perf_counter() - t0
    scale = n_s / HEADLINE_ROWS                        # same unit as `value`: queries/s over 100M rows
    # ... and the strongest CPU form of the same arithmetic this repository has: the batched oracle driver (queries in
    # vector lanes, every row read once per 1024-query batch instead of once per query), row slices on all threads
    qb = tq[:256].cpu().numpy()
    step_rows = (n_s + threads - 1) // threads
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda lo: o.scan_cosine_many(corpus[lo:lo + step_rows], qb, k, -1.0), range(0, n_s, step_rows)))
    dtb = time.perf_counter() - t0
    batched = {"value": qb.shape[0] / dtb * scale, "unit": "QPS", "cores": threads, "sample_queries": int(qb.shape[0]),
               "sample_seconds": dtb, "qps_on_one_shard": qb.shape[0] / dtb * n_s / rows_total,
               "what": "oracle_exact_scan_cosine_many: the same fp64 arithmetic per (row, query), 8 queries per vector, "
                       "4-8 rows interleaved; NOT how the reference runs (it loops over the queries: sqlite_vec_backend.cpp:1612-1647)"}
    port = {"value": n_par / dtn * scale, "unit": "QPS", "cores": threads, "kind": "port", "batched_over_queries": batched,
            "sample": f"{n_par} queries (one at a time per thread, {threads} threads) x first {n_s} rows of the shard, scalar fp64 "
                      f"oracle scan, {dtn:.1f} s; scaled by {n_s}/{HEADLINE_ROWS}",
            "sample_rows": n_s, "sample_queries": n_par, "sample_seconds": dtn, "scaled_by": scale,
            "qps_on_one_shard": n_par / dtn * n_s / rows_total,
            "single_thread": {"value": n1 / dt1 * scale, "cores": 1, "sample_queries": n1, "sample_seconds": dt1,
                              "qps_on_one_shard": n1 / dt1 * n_s / rows_total},
            "host_cores_available": os.cpu_count(),
            "note": "the C restatement over a dense matrix: no SQLite row fetch — an upper bound of the reference's rate"}
    # The REFERENCE'S OWN loop when it travelled (oracle/_ref/libyams_scan_ref.so: bruteForceSearchUnlocked cut verbatim from
    # /root/reference, over an in-memory SQLite `vectors` table — row fetch, blob access, heap and recordFromStatement
    # included, as a yams host runs it): one table per thread, one query at a time per thread.
    try:
        ref_rows = min(50_000, n_s)
        tables = []

        def make_table(_):
            t = _oracle.scan_ref()
            if t is not None:
                t.insert_rows(corpus[:ref_rows])
            return t
        with ThreadPoolExecutor(max_workers=threads) as ex:
            tables = list(ex.map(make_table, range(threads)))
        if tables and all(t is not None for t in tables):
            per_t = 16                                                 # ~20 CPU-seconds on a 16-thread box
            tables[0].search(queries[0], k, -1.0)                      # (page in)
            t0 = time.perf_counter()
            r1 = [tables[0].search(queries[i % len(queries)], k, -1.0) for i in range(4)]
            dt1r = time.perf_counter() - t0
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as ex:
                list(ex.map(lambda ti: [tables[ti].search(queries[(ti * per_t + j) % len(queries)], k, -1.0) for j in range(per_t)],
                            range(threads)))
            dtr = time.perf_counter() - t0
            # the same queries through the restatement: the timed reference call and the checker agree on this very sample
            ok = True
            for i in range(4):
                rows, sims, _, _ = o.scan_cosine(corpus[:ref_rows], queries[i % len(queries)], k, -1.0)
                ok &= bool((r1[i][0] == rows).all() and (r1[i][1].view("uint32") == sims.view("uint32")).all())
            for t in tables:
                t.close()
            sc = ref_rows / HEADLINE_ROWS
            return {"value": threads * per_t / dtr * sc, "unit": "QPS", "cores": threads, "kind": "reference",
                    "sample": f"{threads * per_t} queries (one at a time per thread, {threads} threads, a table per thread) x {ref_rows} rows of "
                              f"the shard in an in-memory SQLite `vectors` table, the reference's own bruteForceSearchUnlocked, {dtr:.1f} s; "
                              f"scaled by {ref_rows}/{HEADLINE_ROWS}",
                    "sample_rows": ref_rows, "sample_queries": threads * per_t, "sample_seconds": dtr, "scaled_by": sc,
                    "qps_on_one_shard": threads * per_t / dtr * ref_rows / rows_total,
                    "single_thread": {"value": 4 / dt1r * sc, "cores": 1, "sample_queries": 4, "sample_seconds": dt1r},
                    "oracle_agrees_on_the_sample": ok, "host_cores_available": os.cpu_count(),
                    "what": "oracle/_ref/libyams_scan_ref.so — src/vector/sqlite_vec_backend.cpp:4115-4409 compiled from the reference's "
                            "sources (oracle/gen_scan_ref.py), SQLite row fetch included", "port": port}
    except Exception as e:      # noqa: BLE001 - the restatement's number stands
        port["reference_leg_error"] = repr(e)
    return port


def ingest_cpu_baseline(seed, blen, n_sample=96):
    """The reference's own translation units (oracle/_ref: StreamingChunker::chunkData incl. the
    per-chunk SHA-256, + SHA256Hasher::hash of the whole blob), one blob per call: single-thread and
    one blob per thread on all host cores; falls back to the plain-C port when _ref did not travel."""
    from concurrent.futures import ThreadPoolExecutor
    _oracle = oracle_mod()
    o = _oracle.oracle()
    r = _oracle.ref()

    def one(b):
        if r is not None:
            r.chunks(b, "streaming", with_hashes=True)
            r.sha256_hex(b)
        else:
            off, sz = o.chunks(b, "streaming")
            for x, y in zip(off, sz):
                o.sha256_hex(b[int(x):int(x + y)])
            o.sha256_hex(b)

    threads = _oracle.host_threads()
    blobs = [o.synth_bytes(seed, b, 0, blen) for b in range(min(n_sample, 32))]
    t0 = time.perf_counter()
    for b in blobs:
        one(b)
    dt1 = time.perf_counter() - t0
    n_all = max(threads * 48, 64)                      # ~10 CPU-seconds of the reference's chunker + hasher
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda i: one(blobs[i % len(blobs)]), range(n_all)))
    dtn = time.perf_counter() - t0
    return {"value": n_all * blen / dtn / 1e9, "unit": "GB/s", "cores": threads,
            "kind": "reference" if r is not None else "port",
            "sample": f"{n_all} x {blen >> 20} MiB Philox blobs, StreamingChunker defaults + per-chunk and "
                      f"whole-blob SHA-256, one blob per thread, {threads} threads, {dtn:.1f} s",
            "sample_seconds": dtn,
            "single_thread": {"value": len(blobs) * blen / dt1 / 1e9, "cores": 1, "sample_blobs": len(blobs),
                              "sample_seconds": dt1}}


def ingest_leg(acc, torch, gib, seed, verify_all=False):
    """SHA-256 + CDC over device-resident Philox blobs (4 MiB each, product-default chunker)."""
    from yams_amd.accel import cdc_config
    blen = 4 << 20
    free_b, _ = torch.cuda.mem_get_info()
    gib = min(gib, max(1.0, (free_b * 0.70) / (1 << 30) / 1.2))   # blobs + bitmap/slot workspace
    n_blobs = max(1, int(gib * (1 << 30) // blen))
    tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(seed, 0, n_blobs, blen, tb.data_ptr())
    offs = [i * blen for i in range(n_blobs)]
    lens = [blen] * n_blobs
    cfg = cdc_config("streaming")
    acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=3)   # warm-up (workspace allocation)
    acc.synchronize()
    acc.enable_timing(True)
    reps = 3
    # clock / power / throttle reason of the device while the timed calls run (the roofline below is priced at the
    # nominal 2.4 GHz: what the part actually clocks at under this integer load says how much of the gap is the clock)
    tel = None
    try:
        from yams_amd import telemetry as ytel
        bus = ytel.hip_pci_bus(torch.cuda.current_device())
        hw = ytel.hwmon_dir(bus)
        thr = ytel.Throttle(bus)
        smp = ytel.HwmonSampler(hw) if hw else None
        if smp:
            smp.start()
        snap_a = thr.snapshot()
    except Exception:      # noqa: BLE001
        smp = thr = snap_a = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        res = acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=3)
    acc.synchronize(); t1 = time.perf_counter(); dt = (t1 - t0) / reps
    try:
        if thr is not None:
            snap_b = thr.snapshot()
            if smp:
                smp.stop = True; smp.join()
            tel = {"hwmon": smp.window(t0 + 0.05, t1) if smp else None,
                   "throttle": thr.between(snap_a, snap_b) if snap_a and snap_b else {"error": thr.error}}
    except Exception as e:  # noqa: BLE001
        tel = {"error": repr(e)}
    sha_ms, _ = acc.kernel_ms("sha256")
    cdc_ms, _ = acc.kernel_ms("cdc_candidates")
    acc.enable_timing(False)
    total = n_blobs * blen
    # bit-exactness of the timed call's own output on a spread of blobs (all host cores)
    ht = oracle_mod().host_threads()
    n_check = (n_blobs if ht >= 12 else max(64, n_blobs * ht // 16)) if verify_all else 64   # (16 threads: 39 s for 25 600 blobs)
    verified = verify_ingest_sample(acc, res, n_blobs, blen, seed, n_check=n_check, ref_every=8 if verify_all else 0)
    cpu = ingest_cpu_baseline(seed, blen)
    n_chunks = int(res.n_chunks)
    # Roofline: the call is bound by 32-bit integer VALU issue, not by HBM.
    # Instruction model per wave-instruction (64 lanes): one SHA-256 block per lane = 1400 VALU
    # (sha256_batch_kernel; every byte is hashed twice: chunk digest + blob digest, plus one padding
    # block per message), CDC candidates = 9 VALU per byte per lane-unit of 32 B -> 9/64 per byte.
    blocks = 2.0 * total / 64.0 + n_chunks + n_blobs
    wave_instr = blocks / 64.0 * 1400.0 + total * 9.0 / 64.0
    floor_s = wave_instr / INT_VALU_WAVE_INSTR_PER_S
    out = {"value": total / dt / 1e9, "unit": "GB/s", "bytes": total, "blobs": n_blobs, "cpu_baseline": cpu,
           "blob_bytes": blen, "chunks": n_chunks, "ms": dt * 1e3,
           "sha256_kernel_ms": sha_ms, "cdc_candidates_kernel_ms": cdc_ms,
           "chunker": "StreamingChunker defaults (min 16 KiB, max 1 MiB, mask 0x1FFF)",
           "digests": "per-chunk + whole-blob (every byte hashed twice)",
           "bit_exact_vs_cpu_sample": verified, "bit_exact_vs_cpu": verified.get("ok"), "verified_blobs": verified.get("blobs"),
           "blobs_through_reference_tus": verified.get("blobs_also_through_the_reference_translation_units"),
           "roofline": {"bound": "valu-issue (int32)", "achieved": total / dt / 1e9, "peak": total / floor_s / 1e9,
                        "unit": "GB/s", "frac": floor_s / dt,
                        "sclk_MHz": ((tel or {}).get("hwmon") or {}).get("sclk_MHz", {}).get("mean") if tel else None,
                        "frac_at_measured_clock": (floor_s / dt * 2400.0 / (((tel or {}).get("hwmon") or {}).get("sclk_MHz", {}).get("mean") or 2400.0))
                        if tel and ((tel or {}).get("hwmon") or {}).get("sclk_MHz") else None,
                        "limiter": ((tel or {}).get("throttle") or {}).get("limiter") if tel else None,
                        "telemetry": tel,
                        "model": "wave-instructions = SHA blocks/64 x 1400 + bytes x 9/64; floor = that x 4 cycles / "
                                 "(1024 SIMDs x 2.4 GHz)", "model_wave_instructions": wave_instr,
                        "pmc_cross_check": "profiles/r01_ingest_pmc.json (SQ_INSTS_VALU 1.016e11 for the same call: "
                                           "0.79 of the issue roof; builder run)",
                        "hbm_view": {"bytes_read_per_call_est": 3.0 * total, "achieved_GBps": 3.0 * total / dt / 1e9,
                                     "peak_GBps": PEAK_HBM_GBPS}}}
    # the dedup lookup that follows in ContentStore::store (one exists() per chunk in the reference):
    # all chunk digests of the batch against an empty device set, then again (everything known)
    try:
        n = n_chunks
        dset = acc.dedup_set(n)
        flags = torch.empty(n, dtype=torch.uint8, device="cuda")
        t0 = time.perf_counter()
        n_new, b_new, b_dup = dset.insert_device(res.chunk_digest, n, res.chunk_size, flags.data_ptr())
        t1 = time.perf_counter()
        n_new2, _, _ = dset.insert_device(res.chunk_digest, n, res.chunk_size, flags.data_ptr())
        t2 = time.perf_counter()
        out["dedup"] = {"digests": n, "new_first_pass": n_new, "new_second_pass": n_new2,
                        "insert_ms": (t1 - t0) * 1e3, "reinsert_ms": (t2 - t1) * 1e3,
                        "lookups_per_s": n / (t2 - t1), "bytes_new": b_new, "bytes_deduped": b_dup}
        dset.close()
    except Exception as e:  # the ingest number above stands on its own
        out["dedup"] = {"error": str(e)}
    del tb
    torch.cuda.empty_cache()
    try:
        out["breadth"] = ingest_breadth(acc, torch, seed)
    except Exception as e:  # the headline number above stands on its own
        out["breadth"] = {"error": repr(e)}
    return out


def ingest_breadth(acc, torch, seed):
    """The ingest path away from its friendliest input: (i) host-streamed — blobs in pinned
    host memory cross PCIe in double-buffered batches (yams_ingest_host), reported apart from the device-
    resident number; (ii) a skewed blob set, 1 KiB ... 64 MiB log-uniform plus the sizes around the
    chunker's window / min / max; (iii) the reference's own chunking benchmark configuration
    (core_benchmarks.cpp:225-229: RabinChunker 4 KiB / 16 KiB / 64 KiB on 1 MiB inputs).  Each leg's
    output is checked against the CPU on a spread of blobs."""
    import hashlib
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from yams_amd.accel import cdc_config
    _oracle = oracle_mod()
    o = _oracle.oracle()
    res = {}

    def check(blobs_of, first, co, cs, cd, bd, mode, cfg, pick):
        def verify(bi):
            b = blobs_of(bi)
            ooff, osz = o.chunks(b, mode, **cfg)
            lo, hi = int(first[bi]), int(first[bi + 1])
            if hi - lo != len(ooff) or not (np.array_equal(co[lo:hi], ooff) and np.array_equal(cs[lo:hi], osz)):
                return False
            mv = memoryview(b)
            if bd[bi].tobytes() != hashlib.sha256(mv).digest():
                return False
            return all(cd[j].tobytes() == hashlib.sha256(mv[int(co[j]):int(co[j] + cs[j])]).digest() for j in range(lo, hi))
        with ThreadPoolExecutor(max_workers=_oracle.host_threads(64)) as ex:
            return bool(all(ex.map(verify, pick)))

    # (i) host-streamed: 8 GiB of the config-5 blobs in pinned host memory
    blen, n_blobs = 4 << 20, 2048
    host = torch.empty(n_blobs * blen, dtype=torch.uint8, pin_memory=True)
    stage = torch.empty(256 * blen, dtype=torch.uint8, device="cuda")
    for b0 in range(0, n_blobs, 256):
        acc.synth_bytes(seed, b0, 256, blen, stage.data_ptr()); acc.synchronize()
        host[b0 * blen:(b0 + 256) * blen].copy_(stage)
    del stage
    torch.cuda.empty_cache()
    base = host.data_ptr()
    ptrs = [base + i * blen for i in range(n_blobs)]
    cfg = cdc_config("streaming")
    batch = 2 << 30
    acc.ingest_host(ptrs, [blen] * n_blobs, cfg, flags=3, batch_bytes=batch)         # warm-up (all four batch buffers + lane tables)
    t0 = time.perf_counter()
    h = acc.ingest_host(ptrs, [blen] * n_blobs, cfg, flags=3, batch_bytes=batch)
    dt = time.perf_counter() - t0
    pick = sorted(set(int(x) for x in np.linspace(0, n_blobs - 1, 32).round()))
    ok = check(lambda bi: o.synth_bytes(seed, bi, 0, blen), h["blob_first"], h["chunk_offset"], h["chunk_size"],
               h["chunk_digest"], h["blob_digest"], "streaming", {}, pick)
    res["host_streamed"] = {"value": n_blobs * blen / dt / 1e9, "unit": "GB/s", "bytes": n_blobs * blen, "ms": dt * 1e3,
                            "blobs": n_blobs, "blob_bytes": blen, "batch_bytes": batch, "source": "pinned host memory",
                            "includes": "H2D of every byte + kernels + D2H of chunk tables and digests (never the headline `value`)",
                            "bound": "the SHA-256 chains: one lane per blob, ~120 ms for 4 MiB, 512 chains per 2 GiB batch; each batch's chains run on "
                                     "a low-priority lane of their own and are joined three batches later.  This call is only four batches: the last "
                                     "ones' chains have nothing to hide under.  The steady state is the host_streamed_32GiB leg below",
                            "chunks": h["n_chunks"], "bit_exact_vs_cpu_sample": {"blobs": len(pick), "ok": ok}}
    # the steady state of the same stream ("a >= 32 GiB run"): the 8 GiB of pinned blobs four times over
    # (the device neither knows nor cares that batch i + 4 reads the same host pages as batch i), in batches of the library's choice
    try:
        reps_long = 4
        ptrs_l = ptrs * reps_long
        batch_l = 0     # the library's own choice for this call (8 GiB here: 4 MiB blobs, 32 GiB)
        acc.ingest_host(ptrs_l, [blen] * len(ptrs_l), cfg, flags=3, batch_bytes=batch_l)      # warm-up: all four device buffers and lane tables of that size
        # three consecutive calls (round 5 saw 9 and 46 GB/s for this very call — the slot buffers were allocated
        # and freed by every call; they now wait in the library's pool of call-sized buffers between calls)
        runs, allocs = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            hl = acc.ingest_host(ptrs_l, [blen] * len(ptrs_l), cfg, flags=3, batch_bytes=batch_l)
            runs.append(time.perf_counter() - t0)
            allocs.append((acc.device_info().get("last_host_ingest") or {}).get("alloc_ms"))
        dtl = sorted(runs)[1]
        same = all(np.array_equal(hl["blob_digest"][r * n_blobs:(r + 1) * n_blobs], h["blob_digest"]) for r in range(reps_long)) and \
            hl["n_chunks"] == reps_long * h["n_chunks"] and np.array_equal(hl["chunk_digest"][:h["n_chunks"]], h["chunk_digest"][:h["n_chunks"]])
        res["host_streamed_32GiB"] = {"value": len(ptrs_l) * blen / dtl / 1e9, "unit": "GB/s", "bytes": len(ptrs_l) * blen, "ms": dtl * 1e3,
                                      "blobs": len(ptrs_l), "blob_bytes": blen, "batch_bytes": "library default (about 2048 of the longest blob, 1-8 GiB, >= 4 batches per call): 8 GiB here",
                                      "by_batch_size_GBps": "1 / 2 / 4 / 8 GiB batches: 17.5 / 32 / 41 / 46 before the streams got priority classes of their own, 2 / 8 GiB: 43.6 / 46.2 after (scripts/dbg/host_stream_batches.py, round 4)",
                                      "source": "pinned host memory (the 8 GiB above, streamed four times in one call)",
                                      "equals_the_8GiB_call_repeated": bool(same),
                                      "three_consecutive_calls_GBps": [round(len(ptrs_l) * blen / r_ / 1e9, 2) for r_ in runs],
                                      "spread": (max(runs) - min(runs)) / dtl, "slot_buffer_alloc_ms": allocs}
        del hl
    except Exception as e:      # noqa: BLE001
        res["host_streamed_32GiB"] = {"error": repr(e)}
    # the same bytes as 256 KiB blobs (chains of 4096 blocks: ~8 ms), 512 MiB batches: the link is the bound
    blen2 = 256 << 10
    n2 = n_blobs * blen // blen2
    ptrs2 = [base + i * blen2 for i in range(n2)]
    acc.ingest_host(ptrs2[:2048], [blen2] * 2048, cfg, flags=3, batch_bytes=512 << 20)
    t0 = time.perf_counter()
    h2 = acc.ingest_host(ptrs2, [blen2] * n2, cfg, flags=3, batch_bytes=512 << 20)
    dt2 = time.perf_counter() - t0
    pick2 = sorted(set(int(x) for x in np.linspace(0, n2 - 1, 64).round()))
    ok2 = check(lambda bi: o.synth_bytes(seed, bi // 16, (bi % 16) * blen2, blen2), h2["blob_first"], h2["chunk_offset"],
                h2["chunk_size"], h2["chunk_digest"], h2["blob_digest"], "streaming", {}, pick2)
    res["host_streamed_small_blobs"] = {"value": n2 * blen2 / dt2 / 1e9, "unit": "GB/s", "bytes": n2 * blen2, "ms": dt2 * 1e3,
                                        "blobs": n2, "blob_bytes": blen2, "batch_bytes": 512 << 20, "source": "pinned host memory",
                                        "bound": "PCIe (H2D of every byte, the next batch on its way under this one's kernels); round 2: 26-28 GB/s — each "
                                                 "batch's small table copies queued behind the next batch's 512 MiB upload on the shared DMA engines, "
                                                 "and its results went home as four blocking copies into pageable memory",
                                        "chunks": h2["n_chunks"], "bit_exact_vs_cpu_sample": {"blobs": len(pick2), "ok": ok2}}
    del host

    # (ii) skewed blob set, device-resident
    rng = np.random.default_rng(seed)
    special = [0, 1, 47, 48, 49, 16383, 16384, 16385, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 64 << 20]
    lens, tot = list(special), sum(special)
    while tot < (8 << 30):
        n = int(2.0 ** rng.uniform(10, 26)); lens.append(n); tot += n
    lens = [lens[i] for i in rng.permutation(len(lens))]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    tb = torch.empty(tot + 64, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(seed + 1, 0, 1, tot, tb.data_ptr())              # one Philox stream; blob i = bytes [offs[i], +lens[i])
    acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=3); acc.synchronize()
    t0 = time.perf_counter()
    acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=1)      # boundaries + chunk digests only
    acc.synchronize(); dt_chunks = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=3)
    acc.synchronize(); dt = time.perf_counter() - t0
    out = acc.fetch_ingest(r, len(lens))
    small = [i for i, n in enumerate(lens) if n <= (8 << 20)]
    pick = sorted(set([i for i, n in enumerate(lens) if n in special and n <= (8 << 20)] +
                      [small[int(x)] for x in np.linspace(0, len(small) - 1, 48).round()]))
    ok = check(lambda bi: o.synth_bytes(seed + 1, 0, int(offs[bi]), lens[bi]), out["blob_first"], out["chunk_offset"],
               out["chunk_size"], out["chunk_digest"], out["blob_digest"], "streaming", {}, pick)
    res["skewed_blob_set"] = {"value": tot / dt / 1e9, "unit": "GB/s", "bytes": tot, "ms": dt * 1e3, "blobs": len(lens),
                              "sizes": "1 KiB ... 64 MiB log-uniform + {0, 1, 47, 48, 49, min-1, min, min+1, max-1, max, max+1, 64 MiB}",
                              "largest_blob_bytes": max(lens), "chunks": int(r.n_chunks),
                              "bound": "the whole-blob SHA-256 of the largest blob: ONE sequential chain of 2^20 blocks (64 MiB) at "
                                       "~35 MB/s per chain; every other kernel of the call has finished long before",
                              "without_blob_digests": {"value": tot / dt_chunks / 1e9, "unit": "GB/s", "ms": dt_chunks * 1e3,
                                                       "what": "boundaries + per-chunk digests of the same set"},
                              "bit_exact_vs_cpu_sample": {"blobs": len(pick), "ok": ok}}
    # (ii a) the same set with YAMS_INGEST_DEFER_LONG_BLOB_DIGESTS: blobs whose whole-blob chain would
    # outlast the rest of the call (> max(1 MiB, total / 4096)) are left to the host's hasher, which works on its own copy
    # of the bytes WHILE the device call runs; every digest of the combined result is compared with the all-device run
    # above (itself CPU-checked on the sample)
    try:
        thr = max(1 << 20, tot >> 12)
        deferred = [i for i, n in enumerate(lens) if n > thr]
        host_bytes = tb[:tot].cpu().numpy()                       # (the host has these bytes: it uploaded them)
        threads = _oracle.host_threads(64)
        acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=7); acc.synchronize()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            futs = [ex.submit(lambda i=i: hashlib.sha256(memoryview(host_bytes[int(offs[i]):int(offs[i]) + lens[i]])).digest()) for i in deferred]
            r7 = acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=7)
            acc.synchronize(); dt_dev = time.perf_counter() - t0
            host_dg = [f.result() for f in futs]
        dt_all = time.perf_counter() - t0
        out7 = acc.fetch_ingest(r7, len(lens))
        bd7 = out7["blob_digest"].copy()
        zeros_ok = all(not bd7[i].any() for i in deferred)
        for i, dg in zip(deferred, host_dg):
            bd7[i] = np.frombuffer(dg, np.uint8)
        same = bool(zeros_ok and np.array_equal(bd7, out["blob_digest"]) and np.array_equal(out7["chunk_digest"], out["chunk_digest"]) and
                    np.array_equal(out7["chunk_offset"], out["chunk_offset"]) and np.array_equal(out7["blob_first"], out["blob_first"]))
        res["skewed_blob_set"]["with_long_chains_deferred_to_the_host"] = {
            "value": tot / dt_dev / 1e9, "unit": "GB/s", "what": "device call (boundaries + every chunk digest + the whole-blob digests of "
            "blobs <= threshold), host hashing of the deferred blobs running beside it", "ms_device_call": dt_dev * 1e3,
            "threshold_bytes": thr, "deferred_blobs": len(deferred), "deferred_bytes": int(sum(lens[i] for i in deferred)),
            "host_threads": threads, "host_hasher": "hashlib (OpenSSL)", "ms_until_every_digest_is_there": dt_all * 1e3,
            "overall_GBps": tot / dt_all / 1e9, "all_digests_equal_the_all_device_run": same}
        del host_bytes
    except Exception as e:      # noqa: BLE001
        res["skewed_blob_set"]["with_long_chains_deferred_to_the_host"] = {"error": repr(e)}
    del tb
    torch.cuda.empty_cache()

    # (ii b) ONE long message: a SHA-256 chain is sequential — what the device does with it, what one host core does,
    # and what the plugin door therefore refuses (content_hash_v1: YAMS_ERR_UNSUPPORTED above 1 MiB per lone chain)
    try:
        n_one = 16 << 20
        one = o.synth_bytes(seed + 3, 0, 0, n_one)
        t0 = time.perf_counter(); hex_dev = acc.sha256_hex(one); dt_dev = time.perf_counter() - t0
        t0 = time.perf_counter(); hex_host = hashlib.sha256(memoryview(one)).hexdigest(); dt_host = time.perf_counter() - t0
        res["lone_chain"] = {"bytes": n_one, "device_MBps": n_one / dt_dev / 1e6, "device_ms": dt_dev * 1e3,
                             "host_one_core_MBps": n_one / dt_host / 1e6, "host_ms": dt_host * 1e3, "host_hasher": "hashlib (OpenSSL)",
                             "digests_equal": hex_dev == hex_host,
                             "crossover": "none: one chain is slower on the device at every size (fixed call cost ~0.1 ms + ~35 MB/s "
                                          "against > 1 GB/s on a host core); the device wins with many chains per call",
                             "plugin_policy": {"hash_refused_above_bytes": 1 << 20,
                                               "hash_many_refused_when_longest_exceeds": "max(1 MiB, total bytes / 37)",
                                               "status": "YAMS_ERR_UNSUPPORTED -> the host's own SHA256Hasher (AccelSHA256Hasher takes it as a "
                                                         "constructor argument)"}}
    except Exception as e:      # noqa: BLE001
        res["lone_chain"] = {"error": repr(e)}

    # (iii) the reference benchmark's configuration: RabinChunker 4K / 16K / 64K on 1 MiB inputs
    blen, n_blobs = 1 << 20, 8192
    tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(seed + 2, 0, n_blobs, blen, tb.data_ptr())
    rcfg = dict(min_size=4096, max_size=65536)
    c3 = cdc_config("rabin", **rcfg)
    offs = [i * blen for i in range(n_blobs)]
    acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, c3, flags=1); acc.synchronize()
    t0 = time.perf_counter()
    r = acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, c3, flags=1)   # chunk digests only, like Chunk::hash of the benchmark
    acc.synchronize(); dt = time.perf_counter() - t0
    r = acc.ingest_device(tb.data_ptr(), offs, [blen] * n_blobs, c3, flags=3)
    out = acc.fetch_ingest(r, n_blobs)
    pick = sorted(set(int(x) for x in np.linspace(0, n_blobs - 1, 64).round()))
    ok = check(lambda bi: o.synth_bytes(seed + 2, bi, 0, blen), out["blob_first"], out["chunk_offset"], out["chunk_size"],
               out["chunk_digest"], out["blob_digest"], "rabin", rcfg, pick)
    ref_cpu = None
    rref = _oracle.ref()
    if rref is not None:
        b = o.synth_bytes(seed + 2, 0, 0, blen)
        t0 = time.perf_counter()
        for _ in range(8):
            rref.chunks(b, "rabin", with_hashes=True, **rcfg)
        ref_cpu = (time.perf_counter() - t0) / 8 * 1e3
    res["reference_benchmark_config"] = {"value": n_blobs * blen / dt / 1e9, "unit": "GB/s", "blobs": n_blobs, "blob_bytes": blen,
                                         "config": "RabinChunker min 4096 / target 16384 / max 65536, 1 MiB inputs, per-chunk SHA-256 "
                                                   "(tests/benchmarks/core_benchmarks.cpp:225-229, 'Chunking_Rabin_1MB')",
                                         "ms_per_1MiB_input": dt * 1e3 / n_blobs, "chunks": int(r.n_chunks),
                                         "reference_tus_ms_per_1MiB_input_one_core": ref_cpu,
                                         "reference_published_ms": 19.0242,
                                         "reference_published_source": "tests/benchmarks/baseline/core_benchmarks.baseline.json (other hardware)",
                                         "bit_exact_vs_cpu_sample": {"blobs": len(pick), "ok": ok}}
    return res


def verify_ingest_sample(acc, res, n_blobs, blen, seed, n_check, ref_every=0):
    """Blobs spread over the timed call's result (ALL of them when n_check >= n_blobs: --verify-all-ingest), each
    regenerated on the CPU (Philox), chunked by the oracle and hashed with hashlib; with ref_every = r every r-th blob
    also goes through the reference's own translation units (oracle/_ref) and must give the same chunk list and
    per-chunk hashes.  Returns {"blobs": n, "ok": bool, ...}."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    _oracle = oracle_mod()
    o = _oracle.oracle()
    rref = _oracle.ref() if ref_every else None
    out = acc.fetch_ingest(res, n_blobs)
    first, co, cs, cd, bd = out["blob_first"], out["chunk_offset"], out["chunk_size"], out["chunk_digest"], out["blob_digest"]
    pick = sorted(set(int(x) for x in np.linspace(0, n_blobs - 1, min(n_check, n_blobs)).round()))
    t_begin = time.perf_counter()
    via_ref = [0]

    def verify(bi):
        blob = o.synth_bytes(seed, bi, 0, blen)
        ooff, osz = o.chunks(blob, "streaming")
        lo, hi = int(first[bi]), int(first[bi + 1])
        if hi - lo != len(ooff) or not (np.array_equal(co[lo:hi], ooff) and np.array_equal(cs[lo:hi], osz)):
            return False
        mv = memoryview(blob)
        if bd[bi].tobytes() != hashlib.sha256(mv).digest():
            return False
        if not all(cd[j].tobytes() == hashlib.sha256(mv[int(co[j]):int(co[j] + cs[j])]).digest() for j in range(lo, hi)):
            return False
        if rref is not None and bi % ref_every == 0:     # the reference's StreamingChunker + SHA256Hasher, built from its own sources
            roff, rsz, rhx = rref.chunks(blob, "streaming", with_hashes=True)
            if not (np.array_equal(co[lo:hi], roff) and np.array_equal(cs[lo:hi], rsz)):
                return False
            if any(cd[lo + j].tobytes().hex() != rhx[j] for j in range(hi - lo)):
                return False
            via_ref[0] += 1
        return True

    threads = _oracle.host_threads(64)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        ok = all(ex.map(verify, pick))
    r = {"blobs": len(pick), "ok": bool(ok)}
    if len(pick) == n_blobs or ref_every:
        r.update({"of_blobs": n_blobs, "bytes": len(pick) * blen, "every_byte": len(pick) == n_blobs,
                  "checked": "chunk boundaries, every chunk digest, the whole-blob digest of every blob listed",
                  "blobs_also_through_the_reference_translation_units": via_ref[0] if rref is not None else None,
                  "host_threads": threads, "seconds": time.perf_counter() - t_begin})
    return r


def result_digest(rows, scores, counts):
    """sha256 over the merged result of one batch (rows, score bits, counts): two paths agree iff their digests do."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(rows, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(scores, dtype=np.float32).tobytes())
    h.update(np.ascontiguousarray(counts).astype(np.uint32).tobytes())
    return h.hexdigest()


def c_abi_sharded_run(a, devices, views=None, keep=None, n_query_batches=4, oracle_queries=0, collective="rccl"):
    """The multi-GPU form a C++ host gets (sharded_api.cpp): ONE process, `devices` driven through yams_scan_sharded_*
    — persistent shard workers, one RCCL communicator (of one rank on a one-GPU box), per batch one ncclAllGather of
    the packed per-shard records on a side stream + merge_topk_kernel, `lanes` batches in flight; with more than one
    shard the exchange of batch i is fenced in front of each shard's sweep of batch i + 1.  Queries and results live in HOST memory here
    (pinned staging + PCIe both ways are inside the timed region).  `views`: per-device corpus views to reuse
    (the default N = 1 run hands over its resident shard); otherwise every shard is generated and its shadows built
    through the C ABI alone — no torch tensor is involved."""
    import numpy as np
    from yams_amd.accel import ShardedScan
    from yams_amd._lib import SCAN_COSINE
    rccl_library = getattr(a, "rccl_library", None)
    if rccl_library:
        collective = "rccl"
    n, d, nq, k = a.rows_per_gpu, a.dim, a.queries, a.k
    world = len(devices)
    # three batches in flight: with queries and results crossing the host, two lanes leave the GPU waiting for the
    # host's turn-around between a wait() and the next submit() (8.66 ms per step against 8.49 / 8.44 with 3 / 4 lanes)
    lanes = max(3, a.lanes)
    t_setup = time.perf_counter()
    # RCCL prints a version banner on STDOUT when its first communicator is formed: keep this process's stdout
    # for the one JSON line (file descriptor 1 points at stderr while the handle is created)
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        sh = ShardedScan(devices, lanes=lanes, collective=collective, rccl_library=rccl_library, fence=not getattr(a, "no_fence", False))
    finally:
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # the banner leaves C stdio's buffer while descriptor 1 still points at stderr
        except Exception:                       # noqa: BLE001
            pass
        os.dup2(saved, 1); os.close(saved)
    info = sh.info()
    own = []
    if views is None:
        views = []
        for i in range(world):
            c = sh.ctx(i)
            rows = c.alloc(n * d * 4); own.append(rows)
            c.synth_rows(a.seed, n * i, n, d, rows.ptr)
            bf = c.alloc(n * d * 2); nsq = c.alloc(n * 4); own += [bf, nsq]
            c.build_shadow_device(rows.ptr, n, d, bf.ptr, nsq.ptr)
            i8 = c.alloc((n + 63) // 64 * 64 * d); m8 = c.alloc((n + 63) // 64 * 8 + 64); own += [i8, m8]
            c.build_shadow_i8_device(rows.ptr, n, d, i8.ptr, m8.ptr)
            views.append(c.corpus_view(rows.ptr, n, d, row_base=n * i, rows_bf16_ptr=bf.ptr, rows_nsq_ptr=nsq.ptr,
                                       rows_i8_ptr=i8.ptr, rows_i8_meta_ptr=m8.ptr))
        for i in range(world):
            sh.ctx(i).synchronize()
    setup_s = time.perf_counter() - t_setup
    # the same query batches as the main run: Philox rows (1 << 40) + b * nq ...
    c0 = sh.ctx(0)
    stage = c0.alloc(nq * d * 4)
    qb = []
    for b in range(n_query_batches):
        c0.synth_rows(a.seed, (1 << 40) + b * nq, nq, d, stage.ptr); c0.synchronize()
        qb.append(stage.download(np.float32, nq * d).reshape(nq, d))
    stage.free()

    def run(count, first, collect=None):
        inflight = []
        for i in range(first, first + count):
            if len(inflight) == lanes:
                j, l = inflight.pop(0)
                r = sh.wait(l)
                if collect is not None:
                    collect[j] = r
            inflight.append((i, sh.submit(views, qb[i % n_query_batches], k, -1.0, SCAN_COSINE, want_diag=False)))
        for j, l in inflight:
            r = sh.wait(l)
            if collect is not None:
                collect[j] = r
    # every lane runs once before the timed region (a lane's first batch sizes its workspace and pinned staging: with
    # the default --warmup 2 and three lanes that first batch used to sit inside the timed steps)
    run(max(a.warmup, lanes), 0)
    timed_ctx = [[sh.lane_ctx(i, l) for l in range(lanes)] for i in range(world)]
    for cs in timed_ctx:
        for c in cs:
            c.enable_timing(True)
    info0 = sh.info()
    got = {}
    t0 = time.perf_counter()
    run(a.steps, a.warmup, collect=got)
    dt = time.perf_counter() - t0
    info1 = sh.info()
    per_shard_ms = []          # mean filter-sweep launch per shard (a spread says which GPU is the slow one)
    for cs in timed_ctx:
        tot, cnt = 0.0, 0
        for c in cs:
            ms, n_ = c.kernel_ms("scan_filter")
            if ms is not None and n_:
                tot += ms * n_; cnt += n_
            c.enable_timing(False)
        per_shard_ms.append(tot / cnt if cnt else None)
    filt_ms = per_shard_ms[0]
    have_ms = [m for m in per_shard_ms if m is not None]
    # the exchange over the timed batches alone: device time of all-gather + merge + download on the root's side stream
    ex_n = info1.get("exchanges_timed", 0) - info0.get("exchanges_timed", 0)
    ex_ms = ((info1.get("exchange_ms", 0.0) * info1.get("exchanges_timed", 0) - info0.get("exchange_ms", 0.0) * info0.get("exchanges_timed", 0)) / ex_n) if ex_n else None
    last = a.warmup + a.steps - 1
    res = got[last]
    digests = {}
    for j in sorted(got)[-n_query_batches:]:
        digests[str(j % n_query_batches)] = result_digest(got[j].rows, got[j].scores, got[j].counts)
    # the exchange itself, checked at any N: per-shard top-k through the single-shard C-ABI call, merged on the host
    # with the reference comparator (similarity desc, row id asc), against the handle's merged result
    nchk = min(8, nq)
    qs = qb[last % n_query_batches][:nchk]
    parts = [sh.ctx(i).scan_topk(views[i], qs, k, -1.0, SCAN_COSINE) for i in range(world)]
    exch_ok = True
    for qi in range(nchk):
        rows = np.concatenate([p.rows[qi, :int(p.counts[qi])] for p in parts])
        sims = np.concatenate([p.scores[qi, :int(p.counts[qi])] for p in parts])
        order = np.lexsort((rows, -sims.astype(np.float64)))[:k]
        exch_ok &= bool(int(res.counts[qi]) == len(order) and np.array_equal(res.rows[qi, :len(order)], rows[order])
                        and np.array_equal(res.scores[qi, :len(order)].view(np.uint32), sims[order].view(np.uint32)))
    out = {"what": "one process, yams_scan_sharded_* (C ABI): persistent shard workers, one RCCL communicator, one ncclAllGather "
                   "of the packed records + merge_topk_kernel per batch on a side stream, submit/wait lanes; queries and "
                   "results in host memory (pinned staging + PCIe both ways inside the timed region)",
           "n_devices": world, "devices": list(devices), "lanes": lanes, "collective": info.get("collective"), "fenced": info.get("fenced"),
           "rccl_version": info.get("rccl_version"), "rccl_library": info.get("rccl_library"),
           "communicator_ranks": info.get("communicator_ranks"), "rccl_unavailable": info.get("rccl_unavailable"),
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
           "qps_on_resident_corpus": nq * a.steps / dt, "value": nq * a.steps / dt * (n * world / HEADLINE_ROWS), "unit": "QPS",
           "filter_launch_ms_shard0": filt_ms, "collectives": sh.info().get("collectives"), "batches": sh.info().get("batches"),
           "timed_batches": a.steps, "timed_collectives": info1.get("collectives", 0) - info0.get("collectives", 0),
           "exchange_ms": ex_ms, "exchange_ms_max": info1.get("exchange_ms_max"), "exchanges_timed": ex_n,
           "exchange_what": "events on the root shard's side stream: all-gather + merge + download per batch (the wait for the slowest peer included)",
           "exchange_timeout_ms": info1.get("exchange_timeout_ms"), "communicator_ranks_source": info.get("communicator_ranks_source"),
           "launch_ms_per_shard": per_shard_ms, "launch_ms_min": min(have_ms) if have_ms else None,
           "launch_ms_max": max(have_ms) if have_ms else None,
           "merged_equals_host_merge_of_per_shard_results": {"queries": nchk, "ok": exch_ok},
           "result_digests": digests, "setup_s": setup_s, "distinct_query_batches": n_query_batches}
    if oracle_queries > 0:
        _oracle = oracle_mod()
        qsel = [int(x) for x in np.linspace(0, nq - 1, min(oracle_queries, nq)).round()]
        qh = qb[last % n_query_batches][qsel]
        inter, exact = 0, True
        per_shard = []
        t_or = time.perf_counter()
        for i in range(world):
            base = views[i].rows
            c = sh.ctx(i)

            def fetch(lo, hi, base=base, c=c):
                out_ = np.empty((hi - lo, d), np.float32)
                c._check(c.L.yams_accel_download(c.ctx, out_.ctypes.data, base + lo * d * 4, out_.nbytes))
                return out_
            part = _oracle.scan_threaded(fetch, n, qh, k, slice_rows=32768, threads=min(_oracle.host_threads(), 96))
            per_shard.append([(rows + n * i, sims) for rows, sims in part])
        for j, qi in enumerate(qsel):
            rows = np.concatenate([p[j][0] for p in per_shard]); sims = np.concatenate([p[j][1] for p in per_shard])
            order = np.lexsort((rows, -sims.astype(np.float64)))[:k]
            rows, sims = rows[order], sims[order]
            inter += len(set(res.rows[qi, :k].tolist()) & set(rows.tolist()))
            exact &= bool(int(res.counts[qi]) == len(rows) and np.array_equal(res.rows[qi, :len(rows)], rows)
                          and np.array_equal(res.scores[qi, :len(rows)].view(np.uint32), sims.view(np.uint32)))
        out.update({"recall_at_k": inter / float(len(qsel) * k), "bit_exact_vs_oracle": exact, "oracle_queries": len(qsel),
                    "oracle_seconds": time.perf_counter() - t_or})
    sh.close()
    for o in own:
        o.free()
    return out


def boundary_leg(a, acc, torch, dev, tc_full, rows_c4):
    """The PLUGIN DOOR measured end to end: vector_scan_v1.search_batch as a host calls it — queries in
    pageable host memory, per call: copy into pinned staging, H2D, the scan, D2H, packing of the hit records, free_hits —
    at Q = 1 / 16 / 1024 on BASELINE config 1 (10k x 384, k = 10), config 2 (1M x 384, k = 100) and the config-4 shard
    (k = 100).  The mirrors are the plugin's own (corpus_append from host memory, shadows built at upload)."""
    import ctypes as C
    import numpy as np
    from yams_amd import _lib
    L = acc.L
    L.yams_plugin_shutdown()
    if L.yams_plugin_init(b'{"device": %d, "search_slots": 2}' % dev.index, None) != 0:
        return {"error": "yams_plugin_init failed"}
    p = C.c_void_p()
    L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p))
    vt = C.cast(p, C.POINTER(_lib.VectorScanV1)).contents
    out = {"what": "vector_scan_v1.search_batch from pageable host memory: staging + H2D + scan + D2H + hit packing + free_hits, "
                   "one call at a time (latency) — ms per call, median of the timed calls"}
    legs = [("config1_10k_x384_k10", 10_000, 384, 10), ("config2_1M_x384_k100", 1_000_000, 384, 100)]
    if rows_c4:
        legs.append((f"config4_shard_{rows_c4}_x{a.dim}_k{a.k}", rows_c4, a.dim, a.k))
    for name, n, d, k in legs:
        cid = C.c_uint64()
        if vt.corpus_create(None, d, C.byref(cid)) != 0:
            out[name] = {"error": "corpus_create failed"}; continue
        t0 = time.perf_counter()
        chunk = 1 << 20
        stage = torch.empty((min(n, chunk), d), dtype=torch.float32, device=dev)
        ok = True
        upload_s = 0.0          # the corpus_append calls alone (pageable host memory -> mirror + shadows)
        append_ms = []
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            if tc_full is not None and d == a.dim and n == rows_c4:
                host = tc_full[r0:r0 + m].cpu().numpy()          # the bench's own resident shard, through host memory
            else:
                acc.synth_rows(a.seed + 7, r0, m, d, stage.data_ptr()); acc.synchronize()
                host = stage[:m].cpu().numpy()
            ta = time.perf_counter()
            ok &= vt.corpus_append(None, cid, host.ctypes.data_as(_lib.f32p), m) == 0
            upload_s += time.perf_counter() - ta
            append_ms.append((time.perf_counter() - ta) * 1e3)
        del stage
        staging_s = time.perf_counter() - t0 - upload_s     # this script's own D2H of the rows into fresh pageable arrays (round 3 counted it as upload)
        slowest = None
        try:    # where the slowest append of the series spent its time (health JSON: mapping fresh memory / the copy / the shadows)
            hp = C.c_void_p()
            if L.yams_plugin_get_health_json(C.byref(hp)) == 0:
                hj = json.loads(C.string_at(hp).decode()); L.yams_accel_free_string(hp)
                slowest = hj.get("slowest_append")
        except Exception:      # noqa: BLE001
            slowest = None
        if not ok:
            out[name] = {"error": "corpus_append failed"}; vt.corpus_destroy(None, cid); continue
        qdev = torch.empty((1024, d), dtype=torch.float32, device=dev)
        acc.synth_rows(a.seed + 7, 1 << 40, 1024, d, qdev.data_ptr()); acc.synchronize()
        qh = qdev.cpu().numpy()
        leg = {"rows": n, "dim": d, "k": k, "upload_s_incl_shadows": upload_s,
               "upload_GBps": n * d * 4 / upload_s / 1e9 if upload_s > 0 else None,
               "upload_what": "the corpus_append calls alone: pageable host rows -> device mirror through the pinned staging ring, "
                              "device memory mapped behind the mirrors as they grow, bf16 + int8 shadows built",
               "bench_host_staging_s": staging_s,
               "append_calls": len(append_ms), "append_ms_min_median_max": [round(min(append_ms), 2), round(sorted(append_ms)[len(append_ms) // 2], 2), round(max(append_ms), 2)],
               "slowest_append_so_far": slowest}
        for q in (1, 16, 1024):
            hits = C.POINTER(_lib.ScanHit)(); counts = _lib.u32p()
            reps = 30 if (n <= 1_000_000 or q < 1024) else 8
            ts = []
            for it in range(reps + 3):
                qq = qh[(it * q) % (1024 - q + 1):][:q]
                t1 = time.perf_counter()
                st = vt.search_batch(None, cid, qq.ctypes.data_as(_lib.f32p), q, d, k, -1.0, 0, C.byref(hits), C.byref(counts), None)
                if st == 0:
                    vt.free_hits(None, hits, counts)
                t2 = time.perf_counter()
                if st != 0:
                    leg[f"q{q}"] = {"error": st}; break
                if it >= 3:
                    ts.append(t2 - t1)
            if ts:
                ts.sort()
                leg[f"q{q}"] = {"ms_per_call": ts[len(ts) // 2] * 1e3, "min_ms": ts[0] * 1e3, "qps": q / ts[len(ts) // 2], "calls": len(ts)}
        # the device-entry number of the same shape beside it (queries and results stay in HBM)
        out[name] = leg
        vt.corpus_destroy(None, cid)
        del qdev
    L.yams_plugin_shutdown()
    torch.cuda.empty_cache()
    return out


def config2_leg(a, torch, dev, local, lane_counts=None, batches=None, oracle_queries=None):
    """BASELINE config 2 (BASELINE.json configs[1]; the reference's searchSimilarBatch, sqlite_vec_backend.cpp:1612-1647):
    1M x 384 fp32 cosine top-100, query batch 256, one MI355X.  The corpus, both shadows and four distinct query batches are
    resident before the timed region; a step = one 256-query batch through the whole path (prep, sample, tau, sweep, gather,
    top-k, fp64 re-score + proof, status words); `lanes` batches are in flight, each on its own context / stream / host
    thread, as in the headline loop.  The last timed batch is checked against the oracle over all 1M rows, every query."""
    import threading
    import numpy as np
    from yams_amd.accel import Accel, SweepGate
    from yams_amd._lib import SCAN_COSINE
    n, d, nq, k = 1_000_000, 384, 256, 100
    seed = a.seed + 2
    acc0 = Accel(local, torch.cuda.current_stream().cuda_stream)
    tc = torch.empty((n, d), dtype=torch.float32, device=dev)
    acc0.synth_rows(seed, 0, n, d, tc.data_ptr())
    n_qb = 4
    tqs = []
    for b in range(n_qb):
        t_ = torch.empty((nq, d), dtype=torch.float32, device=dev)
        acc0.synth_rows(seed, (1 << 40) + b * nq, nq, d, t_.data_ptr())
        tqs.append(t_)
    tb = torch.empty((n, d), dtype=torch.bfloat16, device=dev); tn = torch.empty(n, dtype=torch.float32, device=dev)
    acc0.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device=dev)
    tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device=dev)
    acc0.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr())
    view = acc0.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                            rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
    acc0.synchronize()
    lane_counts = list(lane_counts or [a.config2_lanes])
    batches = batches or max(200, 10 * a.steps)
    max_l = max(lane_counts)
    streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(max_l - 1)]
    accs = [acc0] + [Accel(local, st.cuda_stream) for st in streams[1:]]
    outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev),
             torch.empty(nq, dtype=torch.int32, device=dev)) for _ in range(max_l)]
    gate = SweepGate(local) if max_l > 1 else None

    def run(lanes, count, first=0):
        errs = []

        def lane_fn(lane):
            try:
                torch.cuda.set_device(dev)
                o = outs[lane]
                for i in range(first + lane, first + count, lanes):
                    accs[lane].scan_topk_device(view, tqs[i % n_qb].data_ptr(), nq, k, -1.0, SCAN_COSINE, o[0].data_ptr(), o[1].data_ptr(),
                                                o[2].data_ptr(), flags=0, want_diag=False)
            except BaseException as e:       # noqa: BLE001 - re-raised on the main thread
                errs.append(e)
        if lanes == 1:
            lane_fn(0)
        else:
            th = [threading.Thread(target=lane_fn, args=(l,)) for l in range(lanes)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]

    sweep = {}
    best = None
    for lanes in lane_counts:
        for c in accs:
            c.set_gate(gate if (lanes > 1 and not a.config2_no_gate) else None)
        run(lanes, max(4 * lanes, 2 * a.warmup))
        for c in accs[:lanes]:
            c.enable_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(lanes, batches)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / batches
        tot, cnt = 0.0, 0
        for c in accs[:lanes]:
            ms, n_ = c.kernel_ms("scan_filter")
            if ms is not None and n_:
                tot += ms * n_; cnt += n_
            c.enable_timing(False)
        sweep[str(lanes)] = {"ms_per_step": dt * 1e3, "qps": nq / dt, "launch_ms": tot / cnt if cnt else None, "launches": cnt}
        if best is None or dt < best[1]:
            best = (lanes, dt, tot / cnt if cnt else None)
    lanes, dt, launch_ms = best
    # the batch the check looks at: one more call on lane 0 (its buffers), diagnostics on
    last_b = (batches - 1) % n_qb
    o = outs[0]
    diag = accs[0].scan_topk_device(view, tqs[last_b].data_ptr(), nq, k, -1.0, SCAN_COSINE, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                    flags=0, want_diag=True)
    torch.cuda.synchronize()
    tr = 256
    n_tiles = (n + tr - 1) // tr
    stride = max(1, n_tiles // ((min(n, max(n // 64, 8192)) + tr - 1) // tr))
    filt_rows = min(n, (n_tiles - (n_tiles + stride - 1) // stride) * tr)
    byts = filt_rows * d + (filt_rows // 64) * 8 + nq * d      # the int8 shadow + its block meta + the int8 queries, read once
    ops = 2.0 * nq * d * filt_rows
    i8 = diag.get("filter_tier") == 1
    leg = {"workload": "BASELINE config 2: 1M x 384 fp32 cosine top-100, query batch 256, corpus + shadows resident",
           "ms_per_step": dt * 1e3, "qps": nq / dt, "search_lanes": lanes, "batches_timed": batches, "launch_ms": launch_ms,
           # 2 Q = 512 int8 multiply-adds per shadow byte against a machine balance of 5 POP/s / 8 TB/s = 625: the sweep of this
           # shape is bound by the ONE read of the int8 shadow from HBM (the MFMA view is beside it)
           "bound": "hbm", "achieved": byts / (launch_ms * 1e-3) / 1e9 if launch_ms else None, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
           "frac": byts / (launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if launch_ms else None,
           "algorithmic_bytes_per_launch": byts, "mfma_view_TOPs": ops / (launch_ms * 1e-3) / 1e12 if launch_ms else None,
           "step_frac_of_hbm_floor": (byts / PEAK_HBM_GBPS / 1e9) / dt,
           "kernel": "scan_tiles_i8d_kernel (two 128-query tiles per row stream, 128 streams)" if i8 else "bf16 tier",
           "filter_tier": diag.get("filter_tier"), "filter_candidates": diag.get("filter_candidates"),
           "rescored_rows": diag.get("rescored_rows"), "widened_queries": diag.get("widened_queries"),
           "exact_fallback_queries": diag.get("exact_fallback_queries"), "lane_sweep": sweep}
    n_oq = nq if oracle_queries is None else min(nq, oracle_queries)
    if n_oq > 0:
        _o = oracle_mod()
        qsel = [int(x) for x in np.linspace(0, nq - 1, n_oq).round()]
        qh = tqs[last_b][qsel].cpu().numpy()
        t_or = time.perf_counter()
        part = _o.scan_threaded(lambda lo, hi: tc[lo:hi].cpu().numpy(), n, qh, k, slice_rows=32768, threads=min(_o.host_threads(), 96))
        t_or = time.perf_counter() - t_or
        rr = o[1].cpu().numpy(); ss = o[0].cpu().numpy(); cc = o[2].cpu().numpy()
        exact = True
        for j, qi in enumerate(qsel):
            rows, sims = part[j][0], part[j][1]
            exact &= bool(cc[qi] == len(rows) and np.array_equal(rr[qi, :len(rows)], rows)
                          and np.array_equal(ss[qi, :len(rows)].view(np.uint32), sims.view(np.uint32)))
        leg["results_identical_to_the_oracle_run"] = exact
        leg["oracle_queries"] = n_oq
        leg["oracle_seconds"] = t_or
    for c in accs:
        c.set_gate(None)
    for c in accs[1:]:
        c.close()
    if gate is not None:
        gate.close()
    del tc, tb, tn, t8, tm8, view, outs
    acc0.close()
    torch.cuda.empty_cache()
    return leg


def pq_leg(a, torch, dev, local, batches=40, oracle_queries=8):
    """SURVEY 8(f) N4 — the product-quantised engine (the reference's default SimeonPqAdc, simeonPqSearchUnlocked,
    sqlite_vec_backend.cpp:3868-4056) on the shape of BASELINE config 2: 1M x 384 rows, the reference's default index (32
    sub-quantisers, rerank factor 2: sqlite_vec_backend.h:55-64), 256 queries per batch, top-100.  Codes, the per-query tables
    (what simeon's PQInnerProductQuery holds — the host builds them, simeon being the host's), tie ranks and rows are resident;
    a step = one batch through yams_scan_pq_topk_device (ADC scan of every code, the best 200 per query, their cosine re-rank
    over the fp32 rows, final order).  Synthetic index: random code bytes and N(0, 0.05) table entries — the arithmetic and
    the traffic are those of a trained index, the recall is not a quantity here.  A few queries are checked against the
    restated oracle (parity of the ADC sum order is unpinned: simeon is absent from the checkout)."""
    import ctypes as C
    import numpy as np
    from yams_amd import _lib
    from yams_amd.accel import Accel
    n, d, m, nq, k, rf = 1_000_000, 384, 32, 256, 100, 2
    acc = Accel(local, torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(a.seed + 31)
    tc = torch.empty((n, d), dtype=torch.float32, device=dev)
    acc.synth_rows(a.seed + 31, 0, n, d, tc.data_ptr())
    codes = torch.randint(0, 256, (n, m), generator=g, device=dev, dtype=torch.uint8)
    luts = (torch.randn((nq, m, 256), generator=g, device=dev) * 0.05).contiguous()
    tq = torch.empty((nq, d), dtype=torch.float32, device=dev)
    acc.synth_rows(a.seed + 31, 1 << 40, nq, d, tq.data_ptr())
    perm = torch.randperm(n, generator=g, device=dev).to(torch.int32)          # rank of every code's tie-break key
    key_row = torch.empty(n, dtype=torch.int32, device=dev); key_row[perm.long()] = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d)
    pq = _lib.ScanPqIndex(codes.data_ptr(), n, m, 0, perm.data_ptr(), key_row.data_ptr())
    prm = _lib.ScanPqParams(k, -1.0, rf, 0)
    o_s = torch.empty((nq, k), dtype=torch.float32, device=dev); o_r = torch.empty((nq, k), dtype=torch.int64, device=dev); o_c = torch.empty(nq, dtype=torch.int32, device=dev)
    diag = _lib.ScanDiag()

    def step(want_diag=False):
        acc._check(acc.L.yams_scan_pq_topk_device(acc.ctx, C.byref(view), C.byref(pq), tq.data_ptr(), luts.data_ptr(), nq, C.byref(prm), None, 0,
                                                  o_s.data_ptr(), o_r.data_ptr(), o_c.data_ptr(), C.byref(diag) if want_diag else None))
    for _ in range(3):
        step()
    acc.enable_timing(True)
    acc.synchronize(); t0 = time.perf_counter()
    for _ in range(batches):
        step()
    acc.synchronize(); dt = (time.perf_counter() - t0) / batches
    adc_ms, adc_n = acc.kernel_ms("pq_adc")
    acc.enable_timing(False)
    step(want_diag=True); acc.synchronize()
    groups = (nq + 3) // 4
    leg = {"workload": "product-quantised engine (SimeonPqAdc): 1M x 384 rows, 32 sub-quantisers, rerank factor 2, 256 queries per batch, top-100; index and tables resident",
           "ms_per_step": dt * 1e3, "qps": nq / dt, "adc_launch_ms": adc_ms, "adc_launches": adc_n,
           "kernel": "pq_adc_filter_kernel<.., 4, 2> (four queries' tables interleaved in LDS: one 16-byte LDS read per code byte; keys only of the codes that reach the sampled threshold)",
           "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
           "algorithmic_bytes_per_launch": n * m + nq * m * 1024,
           "bytes_moved_by_design_per_launch": n * m * groups + nq * m * 1024,
           "lds_bytes_per_launch": n * m * nq * 4, "lds_peak_TBps": 256 * 256 * 2.4e9 / 1e12,
           "rescored_rows": diag.as_dict().get("rescored_rows"), "filter_candidates": diag.as_dict().get("filter_candidates")}
    if adc_ms:
        leg["achieved"] = leg["algorithmic_bytes_per_launch"] / (adc_ms * 1e-3) / 1e9
        leg["frac"] = leg["achieved"] / 8000.0
        leg["design_bytes_GBps"] = leg["bytes_moved_by_design_per_launch"] / (adc_ms * 1e-3) / 1e9
        leg["lds_TBps"] = leg["lds_bytes_per_launch"] / (adc_ms * 1e-3) / 1e12       # the resource that binds: one table entry per (query, code byte)
        leg["lds_frac"] = leg["lds_TBps"] / leg["lds_peak_TBps"]
        leg["note"] = ("the HBM view prices the launch against reading every code byte once; four queries' tables fill the LDS, so the codes are "
                       "re-read (from L2 / MALL) once per four queries, and the table look-ups — random 16-byte LDS reads — are what the launch spends its time on")
    n_oq = oracle_queries if a.oracle_queries is None else min(oracle_queries, a.oracle_queries)
    if n_oq > 0:
        o = oracle_mod().oracle()
        hc = tc.cpu().numpy(); hcodes = codes.cpu().numpy(); hl = luts.cpu().numpy(); hq = tq.cpu().numpy()
        tie = perm.cpu().numpy().astype(np.uint64)      # (any keys in the same order as the ranks)
        rr = o_r.cpu().numpy(); ss = o_s.cpu().numpy(); cc = o_c.cpu().numpy()
        t_or = time.perf_counter(); exact = True
        for qi in [int(x) for x in np.linspace(0, nq - 1, n_oq).round()]:
            rows, sims, _ = o.pq_search(hc, hcodes, hl[qi], hq[qi], k, -1.0, rf, tie_keys=tie, chunk_rank=np.arange(n, dtype=np.uint64))
            cnt = int(cc[qi])
            exact = exact and cnt == len(rows) and np.array_equal(rr[qi, :cnt], rows) and np.array_equal(ss[qi, :cnt].view(np.uint32), np.asarray(sims, np.float32).view(np.uint32))
        leg["identical_to_the_restated_oracle"] = bool(exact); leg["oracle_queries"] = n_oq; leg["oracle_seconds"] = time.perf_counter() - t_or
        leg["parity"] = "unpinned for the order of the ADC sum (third_party/simeon absent); the rest restated from sqlite_vec_backend.cpp:3868-4056"
    acc.close()
    return leg



def distribution_leg(a, torch, dev, local, kind):
    """The headline shape (rows_per_gpu x dim, query batch, k) on a corpus that is NOT uniform on the sphere — real embedding
    corpora are clustered and anisotropic (the reference stores what its embedding models emit, src/vector/vector_database.cpp:
    1771-1784), and the filter's cost depends on the data: tau comes from a 1/64 row sample, the bound from per-64-row-block
    residues.  kind = "clustered": 10 000 Gaussian clusters (sigma = 0.35 of the centre's norm per dimension scale), queries are
    perturbed cluster centres; "anisotropic": independent normal components with a power-law variance (j^-1), the shape of a text
    embedding's spectrum.  Rows are generated in HBM with torch (plumbing), normalised in fp32.  Reports the step time, the
    filter launch, the candidate and re-score volumes, the proof outcomes, and checks 64 queries against the oracle."""
    import threading
    import numpy as np
    from yams_amd.accel import Accel, SweepGate
    from yams_amd._lib import SCAN_COSINE, SCAN_L2
    metric_id = SCAN_L2 if a.dist_metric == "l2" else SCAN_COSINE
    n, d, nq, k = a.rows_per_gpu, a.dim, a.queries, a.k
    g = torch.Generator(device=dev); g.manual_seed(a.seed + (11 if kind == "clustered" else 13))
    tc = torch.empty((n, d), dtype=torch.float32, device=dev)
    chunk = 1 << 20
    if kind == "clustered":
        n_c = 10_000
        centres = torch.randn((n_c, d), generator=g, device=dev)
        centres /= centres.norm(dim=1, keepdim=True)
        sigma = 0.35 / (d ** 0.5)
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            cid = torch.randint(0, n_c, (m,), generator=g, device=dev)
            x = centres[cid] + sigma * torch.randn((m, d), generator=g, device=dev)
            tc[r0:r0 + m] = x / x.norm(dim=1, keepdim=True)
        def make_queries():
            cid = torch.randint(0, n_c, (nq,), generator=g, device=dev)
            x = centres[cid] + sigma * torch.randn((nq, d), generator=g, device=dev)
            return (x / x.norm(dim=1, keepdim=True)).contiguous()
    elif kind == "anisotropic":
        scale = (torch.arange(1, d + 1, device=dev, dtype=torch.float32) ** -0.5)
        for r0 in range(0, n, chunk):
            m = min(chunk, n - r0)
            x = torch.randn((m, d), generator=g, device=dev) * scale
            tc[r0:r0 + m] = x / x.norm(dim=1, keepdim=True)
        def make_queries():
            x = torch.randn((nq, d), generator=g, device=dev) * scale
            return (x / x.norm(dim=1, keepdim=True)).contiguous()
    elif kind == "gaussian":        # isotropic normal components: what the rotated layout makes of any corpus, and the int8 bound's
        for r0 in range(0, n, chunk):   # usual width (uniform components, the headline's synthetic rows, quantise 2.4x better)
            m = min(chunk, n - r0)
            x = torch.randn((m, d), generator=g, device=dev)
            tc[r0:r0 + m] = x / x.norm(dim=1, keepdim=True)
        def make_queries():
            x = torch.randn((nq, d), generator=g, device=dev)
            return (x / x.norm(dim=1, keepdim=True)).contiguous()
    else:
        raise ValueError(kind)
    n_qb = 4
    tqs = [make_queries() for _ in range(n_qb)]
    torch.cuda.synchronize()        # (the lanes' streams are not torch's: the rows and queries must have landed)
    acc0 = Accel(local, torch.cuda.current_stream().cuda_stream)
    tb = torch.empty((n, d), dtype=torch.bfloat16, device=dev); tn = torch.empty(n, dtype=torch.float32, device=dev)
    acc0.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device=dev)
    tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device=dev)
    # the layout of the int8 shadow: measured, as the plugin does at a corpus' first append ("i8_layout": "auto")
    i8_flags, res_plain, res_rot = acc0.choose_i8_layout(tc.data_ptr(), n, d) if a.dist_i8_layout == "auto" else ((1 if a.dist_i8_layout == "rotated" else 0), None, None)
    acc0.synchronize(); t_b8 = time.perf_counter()
    mean_res = acc0.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr(), want_mean_err=True, i8_flags=i8_flags)
    t_b8 = (time.perf_counter() - t_b8) * 1e3
    view = acc0.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                            rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr(), i8_flags=i8_flags)
    acc0.synchronize()
    lanes = max(1, a.lanes)
    streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
    accs = [acc0] + [Accel(local, st.cuda_stream) for st in streams[1:]]
    outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev),
             torch.empty(nq, dtype=torch.int32, device=dev)) for _ in range(lanes)]
    gate = SweepGate(local) if lanes > 1 else None
    for c in accs:
        c.set_gate(gate)

    def run(count):
        errs = []

        def lane_fn(lane):
            try:
                torch.cuda.set_device(dev)
                o = outs[lane]
                for i in range(lane, count, lanes):
                    accs[lane].scan_topk_device(view, tqs[i % n_qb].data_ptr(), nq, k, -1.0, metric_id, o[0].data_ptr(), o[1].data_ptr(),
                                                o[2].data_ptr(), flags=a.dist_flags, want_diag=False)
            except BaseException as e:       # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=lane_fn, args=(l,)) for l in range(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
    run(max(2 * lanes, a.warmup))
    for c in accs:
        c.enable_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    tot, cnt = 0.0, 0
    for c in accs:
        ms, n_ = c.kernel_ms("scan_filter")
        if ms is not None and n_:
            tot += ms * n_; cnt += n_
        c.enable_timing(False)
    o = outs[0]
    diags = []
    for b in range(n_qb):       # the proof outcomes of every query batch
        diags.append(accs[0].scan_topk_device(view, tqs[b].data_ptr(), nq, k, -1.0, metric_id, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                              flags=a.dist_flags, want_diag=True))
    torch.cuda.synchronize()
    last = n_qb - 1
    leg = {"distribution": kind, "metric": a.dist_metric, "rows": n, "dim": d, "queries": nq, "k": k, "search_lanes": lanes, "ms_per_step": dt * 1e3, "qps_on_resident_corpus": nq / dt,
           "launch_ms": tot / cnt if cnt else None, "shadow_i8_mean_residue": mean_res, "i8_layout": "rotated" if i8_flags & 1 else "plain",
           "sampled_residue_plain": res_plain, "sampled_residue_rotated": res_rot, "shadow_i8_build_ms": t_b8,
           "filter_tier": diags[0].get("filter_tier"),
           **{kk: [dg.get(kk) for dg in diags] for kk in ("filter_candidates", "rescored_rows", "widened_queries", "retried_queries", "escalated_queries", "exact_fallback_queries")}}
    n_oq = 64 if a.oracle_queries is None else min(64, a.oracle_queries)
    if n_oq > 0:
        _o = oracle_mod()
        qsel = [int(x) for x in np.linspace(0, nq - 1, n_oq).round()]
        qh = tqs[last][qsel].cpu().numpy()
        t_or = time.perf_counter()
        part = _o.scan_threaded(lambda lo, hi: tc[lo:hi].cpu().numpy(), n, qh, k, metric=a.dist_metric, slice_rows=32768, threads=min(_o.host_threads(), 96))
        t_or = time.perf_counter() - t_or
        rr = o[1].cpu().numpy(); ss = o[0].cpu().numpy(); cc = o[2].cpu().numpy()
        exact = True
        for j, qi in enumerate(qsel):
            rows, sims = part[j][0], part[j][1]
            ok_q = bool(cc[qi] == len(rows) and np.array_equal(rr[qi, :len(rows)], rows)
                        and np.array_equal(ss[qi, :len(rows)].view(np.uint32), sims.view(np.uint32)))
            if not ok_q and "first_mismatch" not in leg:       # what differs, for whoever has to look
                m = min(int(cc[qi]), len(rows))
                bad = [i for i in range(m) if rr[qi, i] != rows[i] or ss[qi, i] != sims[i]][:3]
                leg["first_mismatch"] = {"query": qi, "count_device": int(cc[qi]), "count_oracle": len(rows),
                                         "at": [{"rank": i, "row_device": int(rr[qi, i]), "row_oracle": int(rows[i]), "sim_device": float(ss[qi, i]),
                                                 "sim_oracle": float(sims[i])} for i in bad]}
            exact &= ok_q
        leg["bit_exact_vs_oracle"] = exact; leg["oracle_queries"] = n_oq; leg["oracle_seconds"] = t_or
    for c in accs:
        c.set_gate(None)
    for c in accs[1:]:
        c.close()
    if gate is not None:
        gate.close()
    del tc, tb, tn, t8, tm8, view, outs, tqs
    acc0.close()
    torch.cuda.empty_cache()
    return leg


def c_abi_main(a):
    """`bench.py --gpus N --via-c-abi`: the whole job from ONE process through the C ABI; prints the contract's line."""
    devices = [0] * a.gpus if a.single_device else list(range(a.gpus))
    oq = a.oracle_queries if a.oracle_queries is not None else (64 if a.gpus == 1 else 0)
    r = c_abi_sharded_run(a, devices, n_query_batches=max(1, a.query_batches), oracle_queries=oq,
                          collective="peer" if a.single_device and a.gpus > 1 else "rccl")
    if a.child_json:
        _print_on_real_stdout(json.dumps(_clean(r), allow_nan=False))
        return
    n, d, k = a.rows_per_gpu, a.dim, a.k
    tr = 256
    n_tiles = (n + tr - 1) // tr
    stride = max(1, n_tiles // ((min(n, max(n // 64, 8192)) + tr - 1) // tr))
    filt_rows = min(n, (n_tiles - (n_tiles + stride - 1) // stride) * tr)
    flops = 2.0 * a.queries * d * filt_rows
    ach = flops / (r["filter_launch_ms_shard0"] * 1e-3) / 1e12 if r["filter_launch_ms_shard0"] else None
    out = {"metric": "k-NN QPS + recall@k, 100M x 768 fp32 cosine top-100; ingest GB/s SHA-256+CDC",
           "value": r["value"], "unit": "QPS", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "i8 (MFMA filter, exact i32 accumulate) + f64 (exact re-score)", "data": "synthetic",
           "config": {"workload": f"{n}x{d} fp32 cosine top-{k} per GPU (row shard of BASELINE config 4: 100Mx768 over 8 GPUs), "
                                  f"query batch {a.queries}", "rows_per_gpu": n, "corpus_rows": n * a.gpus, "dim": d, "k": k,
                      "query_batch": a.queries,
                      "parallelism": f"row-shard x{a.gpus}, ONE process through the C ABI (yams_scan_sharded_*): one all-gather + merge "
                                     "per batch, fenced in front of the shard's next sweep (DESIGN 4)", "search_lanes": a.lanes},
           "launcher": "single process (--via-c-abi)",
           "roofline": {"bound": "mfma", "kernel": "scan_tiles_i8d_kernel (shard 0's lanes)", "achieved": ach, "peak": PEAK_I8_MFMA_TOPS,
                        "unit": "TOP/s", "frac": ach / PEAK_I8_MFMA_TOPS if ach else None, "launch_ms": r["filter_launch_ms_shard0"],
                        "traffic": None},
           "c_abi_sharded": r,
           "collective": {"backend": f"{r.get('collective')} (one process, C ABI)", "communicator_ranks": r.get("communicator_ranks"),
                          "collectives": r.get("timed_collectives"), "batches": r.get("timed_batches"),
                          "exchange_ms": r.get("exchange_ms"), "exchange_ms_max": r.get("exchange_ms_max"),
                          "launch_ms_min": r.get("launch_ms_min"), "launch_ms_max": r.get("launch_ms_max"),
                          "fenced": r.get("fenced"), "watchdog_s": (r.get("exchange_timeout_ms") or 0) / 1e3}}
    for kk in ("recall_at_k", "bit_exact_vs_oracle", "oracle_queries"):
        if kk in r:
            out[kk] = r[kk]
    emit(out, a.extra_json)


def plain_read_ceiling(achieved_gbps):
    """What a plain streaming read reaches on an MI355X (scripts/ubench/hbm_read.hip, the builder's run committed under
    profiles/): the practical ceiling beside the nominal 8 TB/s."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_read_ubench.json")), reverse=True):
        try:
            best = float(json.load(open(f))["best_GBps"])
            return {"best_GBps": best, "source": os.path.relpath(f, ROOT),
                    "frac_of_it": achieved_gbps / best if achieved_gbps else None}
        except Exception:      # noqa: BLE001
            continue
    return None


def hbm_leg_traffic(n, d, nq, i8):
    """HBM bytes per launch of the small-batch filter from the builder's PMC pass (profiles/*_small_batch_pmc.json),
    when it was taken on this very shape and kernel; labelled as such."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_small_batch_pmc.json")), reverse=True):
        try:
            j = json.load(open(f))
            if j.get("rows_per_gpu") == n and j.get("dim") == d and j.get("queries") == nq and (("i8r" in j.get("kernel", "")) or ("i8d" in j.get("kernel", ""))) == bool(i8):
                return j.get("hbm_bytes_per_launch"), (f"profiles/{os.path.basename(f)} (builder run: rocprofv3 --pmc FETCH_SIZE x2 gfx950 "
                                                       "correction; NOT measured in this run)")
        except Exception:
            pass
    return None, None


def main():
    a = parse()
    if not (a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.via_c_abi):     # (the self-launcher passes its ranks' output through)
        quiet_stdout()
    if a.via_c_abi:
        return c_abi_main(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import numpy as np
    import torch
    from yams_amd import dist as ydist
    from yams_amd.accel import Accel
    from yams_amd._lib import SCAN_COSINE, SCAN_L2, FLAG_NO_I8_FILTER

    if a.single_device:
        os.environ["LOCAL_RANK"] = "0"
    with ydist.Deadline("rendezvous of the ranks (init_process_group)", 300):
        rank, world, local = ydist.init_from_env(a.dist_backend)
    assert world == max(1, a.gpus) or world == 1, (world, a.gpus)
    if not a.single_device and local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU ({torch.cuda.device_count()} visible); "
                         "one rank per GPU (use --single-device --dist-backend gloo for a dry run)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.only_distribution:
        res = {kd: distribution_leg(a, torch, dev, local, kd) for kd in (a.distribution or "clustered,anisotropic,gaussian").split(",")}
        sys.stderr.write("distribution: " + json.dumps(_clean(res)) + "\n")
        _print_on_real_stdout(json.dumps(_clean(res), allow_nan=False))
        return
    if a.only_pq:
        leg = pq_leg(a, torch, dev, local)
        sys.stderr.write("pq: " + json.dumps(_clean(leg)) + "\n")
        _print_on_real_stdout(json.dumps(_clean(leg), allow_nan=False))
        return
    if a.only_config2:
        sweep = [int(x) for x in a.config2_lane_sweep.split(",")] if a.config2_lane_sweep else None
        leg = config2_leg(a, torch, dev, local, lane_counts=sweep, batches=a.config2_batches, oracle_queries=a.oracle_queries)
        sys.stderr.write("config2: " + json.dumps(_clean(leg)) + "\n")
        _print_on_real_stdout(json.dumps(_clean(leg), allow_nan=False))
        return
    acc = Accel(local, torch.cuda.current_stream().cuda_stream)
    n, d, nq, k = a.rows_per_gpu, a.dim, a.queries, a.k
    total_rows = n * world
    row_base = n * rank

    # synthetic data, generated in HBM (Philox recipe of SURVEY.md 8d; regenerable on the CPU)
    tc = torch.empty((n, d), dtype=torch.float32, device=dev)
    acc.synth_rows(a.seed, row_base, n, d, tc.data_ptr())
    # `query_batches` distinct query batches rotate through the steps (same on every rank): batch b = Philox rows
    # (1 << 40) + b * nq ...  `tq` is re-pointed at the batch of the LAST TIMED step below (checks, later legs).
    n_qb = max(1, a.query_batches)
    tqs = []
    for b in range(n_qb):
        t_ = torch.empty((nq, d), dtype=torch.float32, device=dev)
        acc.synth_rows(a.seed, (1 << 40) + b * nq, nq, d, t_.data_ptr())
        tqs.append(t_)
    tq = tqs[0]
    # the filter shadow of the mirror (bf16 copy + squared norms), built once when rows are uploaded
    # (plugin.cpp corpus_append does the same); part of the resident index, not of the timed step
    tb = tn = None
    shadow_ms = None
    if not a.no_shadow and d % 32 == 0:
        tb = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
        tn = torch.empty(n, dtype=torch.float32, device=dev)
        acc.synchronize(); t_sh = time.perf_counter()
        acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
        acc.synchronize(); shadow_ms = (time.perf_counter() - t_sh) * 1e3
    # the INT8 shadow (first filter tier of cosine batches > 128 queries), also built at upload time
    t8 = tm8 = None
    shadow_i8_ms = i8_mean_err = None
    i8_flags, i8_res_plain, i8_res_rot = 0, None, None
    if tb is not None and not a.no_i8 and d % 64 == 0 and d >= 256 and not a.f32_filter and not a.split_filter:
        t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device=dev)
        tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device=dev)
        acc.synchronize(); t_sh = time.perf_counter()
        # the layout is measured, as the plugin does at a corpus' first append (the uniform synthetic rows keep the plain one)
        i8_flags, i8_res_plain, i8_res_rot = acc.choose_i8_layout(tc.data_ptr(), n, d)
        i8_mean_err = acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr(), want_mean_err=True, i8_flags=i8_flags)
        shadow_i8_ms = (time.perf_counter() - t_sh) * 1e3
    view = acc.corpus_view(tc.data_ptr(), n, d, row_base=row_base,
                           rows_bf16_ptr=tb.data_ptr() if tb is not None else None,
                           rows_nsq_ptr=tn.data_ptr() if tn is not None else None,
                           rows_i8_ptr=t8.data_ptr() if t8 is not None else None,
                           rows_i8_meta_ptr=tm8.data_ptr() if tm8 is not None else None, i8_flags=i8_flags if t8 is not None else 0)
    acc.synchronize()

    # the step after the scan: all-gather + merge, two batches in flight (yams_amd/dist.py)
    lanes = max(1, a.lanes)
    pipe = ydist.GatherPipeline(nq, k, dev, depth=lanes)
    if world > 1:
        acc_merge = Accel(local, pipe.side_stream_ptr())        # merge kernel on the side stream

        def merge_fn(g, out):
            acc_merge.merge_topk_device(world, nq, k, -1.0, SCAN_COSINE, g["scores"].data_ptr(), g["rows"].data_ptr(),
                                        g["counts"].data_ptr(), None, None, out["scores"].data_ptr(),
                                        out["rows"].data_ptr(), out["counts"].data_ptr(), None)
        pipe.merge_fn = merge_fn
        pipe.watchdog_s = a.watchdog_s
        # the communicator's first collective (RCCL builds its rings / trees here): apart from the steps, under its own deadline
        with ydist.Deadline(f"first collective of the {world}-rank communicator ({torch.distributed.get_backend()})", 300):
            torch.distributed.barrier()
            torch.cuda.synchronize()

    scan_flags = 4 if a.f32_filter else (8 if a.split_filter else 0)   # YAMS_SCAN_FLAG_F32_FILTER / _SPLIT_FILTER
    if a.half_tile:
        scan_flags |= 32                                                # YAMS_SCAN_FLAG_WIDE_TILE: per-tile kernel forms
    # Search lanes: every lane is its own context (stream + workspace) driven by its own host thread, as the
    # plugin serves concurrent search calls from its pool of contexts (plugin.cpp, "search_slots").  While
    # one lane's batch is in its serial tail (candidate selection, fp64 re-score, the host's look at the
    # proof words) the other lane's filter sweep has the GPU.  Batch i runs on lane i % lanes and owns
    # pipeline slot i % lanes; the collectives are issued in batch order on every rank (a turnstile).
    import threading
    lane_streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
    accs = [acc] + [Accel(local, st.cuda_stream) for st in lane_streams[1:]]
    gate = None
    if lanes > 1 and dev.type == "cuda":
        from yams_amd.accel import SweepGate
        gate = SweepGate(local)         # the lanes' filter sweeps run one after the other, the rest overlaps
        for c in accs:
            c.set_gate(gate)
            # N > 1: the collective behind a batch must not be enqueued under another lane's sweep (a persistent grid that
            # owns every CU: the all-gather kernel would wait for a CU and keep its peers spinning) — the gate stays closed
            # behind a lane's sweep until that lane has enqueued its collective + merge ("the fence")
            if pipe.active:
                c.set_sweep_hold(True)
    turn = [0]
    turn_cv = threading.Condition()
    batch_no = [0]

    def step(i, want_diag=False):
        lane = slot = i % lanes
        pipe.wait(slot)                 # the merge of batch i - lanes has long finished: its record is free
        loc = pipe.local(slot)
        # (the call returns after its own host sync on the query status words: results are complete)
        try:
            diag = accs[lane].scan_topk_device(view, tqs[i % n_qb].data_ptr(), nq, k, -1.0, SCAN_COSINE, loc["scores"].data_ptr(),
                                               loc["rows"].data_ptr(), loc["counts"].data_ptr(), flags=scan_flags,
                                               want_diag=want_diag)
            with turn_cv:
                if turn[0] != i and gate is not None and pipe.active:
                    # not this batch's turn yet (two lanes starting together may reach the gate in either order): a lane
                    # that holds the gate never waits for another lane — it gives the gate up first
                    accs[lane].release_sweep_hold(None)
                while turn[0] != i:
                    turn_cv.wait()
                pipe.launch(slot)       # collective + merge on the side stream, in front of the next sweep (N > 1)
                turn[0] += 1
                turn_cv.notify_all()
        finally:
            if gate is not None and pipe.active:
                accs[lane].release_sweep_hold(pipe.side_stream_ptr())   # the next sweep starts behind collective + merge
        return diag, slot

    def run_steps(count):
        """`count` batches, `lanes` at a time; returns the slot of the last one."""
        first = batch_no[0]
        batch_no[0] += count
        errs = []

        def lane_fn(lane):
            try:
                if dev.type == "cuda":
                    torch.cuda.set_device(dev)
                for i in range(first, first + count):
                    if i % lanes == lane:
                        step(i)
            except BaseException as e:       # noqa: BLE001 - re-raised on the main thread
                errs.append(e)
                with turn_cv:
                    turn[0] = 1 << 60
                    turn_cv.notify_all()
        if lanes == 1 or count == 1:
            for i in range(first, first + count):
                step(i)
        else:
            th = [threading.Thread(target=lane_fn, args=(l,)) for l in range(lanes)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
        return (first + count - 1) % lanes

    progress = {"phase": "setup"}

    def fence():
        with ydist.Deadline("fence (drain + barrier + device synchronize)", 4 * a.watchdog_s, detail=lambda: progress):
            pipe.drain()
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

    def guarded_steps(count, phase):
        # a run of steps can only hang in an exchange a peer never joins (GatherPipeline.wait raises CollectiveTimeout
        # after watchdog_s) or under a device that stopped answering: one line of diagnosis, exit 3, no silent hang
        progress["phase"] = phase
        try:
            with ydist.Deadline(f"{count} steps ({phase})", 4 * a.watchdog_s + 0.5 * count,
                                detail=lambda: {**progress, "batches_issued": batch_no[0], "exchanges_done": pipe.exchanges}):
                return run_steps(count)
        except ydist.CollectiveTimeout as e:
            sys.stderr.write(f"[yams_amd watchdog] rank {rank}/{world}: {e}\n"); sys.stderr.flush()
            os._exit(3)

    if a.warmup:
        guarded_steps(a.warmup, "warmup")
    fence()
    for c in accs:
        c.enable_timing(True)
    pipe.reset_exchange_stats()
    fence(); t0 = time.perf_counter()
    last_slot = guarded_steps(a.steps, "timed steps")
    fence(); dt = time.perf_counter() - t0
    ex_stats = pipe.exchange_stats()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    def lane_kernel_ms(name):
        tot, cnt = 0.0, 0
        for c in accs:
            ms, n_ = c.kernel_ms(name)
            if ms is not None and n_:
                tot += ms * n_; cnt += n_
        return (tot / cnt if cnt else None), cnt
    filt_ms, filt_n = lane_kernel_ms("scan_filter")
    samp_ms, samp_n = lane_kernel_ms("scan_sample")
    rank_stats = [{"rank": rank, "launch_ms": filt_ms, **ex_stats}]
    if world > 1:       # every rank's sweep time and view of the exchange: a bad scaling curve can be attributed
        allr = [None] * world
        torch.distributed.all_gather_object(allr, rank_stats[0])
        rank_stats = allr
    for c in accs:
        c.enable_timing(False)
    res = {kk: v.clone() for kk, v in pipe.result(last_slot).items()}   # the merged top-k of the LAST TIMED step
    last_batch = (batch_no[0] - 1) % n_qb
    tq = tqs[last_batch]                  # its queries: every check and every later leg works on this batch
    r_timed = res["rows"].cpu().numpy().copy()
    s_timed = res["scores"].cpu().numpy().copy()
    c_timed = res["counts"].cpu().numpy().copy()

    # ---- the timed configuration against the oracle over the whole resident corpus ----------------
    if a.oracle_queries is not None:
        n_oq = a.oracle_queries
    elif world > 1:
        n_oq = 16
    else:
        # the whole timed batch where the host can afford it (16 threads: 47 s); a box that grants fewer threads checks a
        # proportional spread of the batch instead of stretching the run (never fewer than 128 queries)
        ht = oracle_mod().host_threads()
        n_oq = nq if ht >= 12 else max(min(nq, 128), nq * ht // 16)
    n_oq = min(n_oq, nq)
    check = None
    if n_oq > 0:
        _oracle = oracle_mod()
        qsel = [int(x) for x in np.linspace(0, nq - 1, n_oq).round()]
        qh = tq[qsel].cpu().numpy()
        stats = {}
        threads = max(1, _oracle.host_threads() // world)
        t_or = time.perf_counter()
        part = _oracle.scan_threaded(lambda lo, hi: tc[lo:hi].cpu().numpy(), n, qh, k, slice_rows=32768,
                                     threads=min(threads, 96), stats=stats)
        t_or = time.perf_counter() - t_or
        mine = [(rows + row_base, sims) for rows, sims in part]
        if world > 1:
            allp = [None] * world
            torch.distributed.all_gather_object(allp, mine)
        else:
            allp = [mine]
        if rank == 0:
            inter, exact = 0, True
            for j, qi in enumerate(qsel):
                rows = np.concatenate([p[j][0] for p in allp]); sims = np.concatenate([p[j][1] for p in allp])
                order = np.lexsort((rows, -sims.astype(np.float64)))[:k]      # (similarity desc, row id asc)
                rows, sims = rows[order], sims[order]
                inter += len(set(r_timed[qi, :k].tolist()) & set(rows.tolist()))
                exact &= bool(c_timed[qi] == len(rows) and np.array_equal(r_timed[qi, :len(rows)], rows)
                              and np.array_equal(s_timed[qi, :len(rows)].view(np.uint32), sims.view(np.uint32)))
            check = {"recall_at_k": inter / float(n_oq * k), "bit_exact_vs_oracle": exact,
                     "recall_checked_on": "timed configuration: the merged top-k of the last timed step "
                                          f"({'int8-shadow' if t8 is not None and nq > 128 else ('bf16-shadow' if tb is not None else 'fp32-view')} filter, {total_rows} rows, "
                                          f"Q={nq}) vs the scalar fp64 oracle over all {total_rows} resident rows",
                     "oracle_queries": n_oq, "oracle_query_ids": qsel, "oracle_seconds": t_or,
                     "oracle_threads_per_rank": stats.get("threads"), "oracle_scan_thread_seconds": stats.get("scan_thread_s")}

    # diagnostics of one more batch, outside the timed region
    diag, _ = step(batch_no[0], want_diag=True)
    batch_no[0] += 1
    fence()

    # ---- clock / power / THROTTLE REASON while the same steps run for ~1.5 s (outside the timed region) ----
    telemetry = None
    if world == 1 and not a.no_telemetry:
        try:
            from yams_amd import telemetry as ytel
            bus = ytel.hip_pci_bus(local)
            hw = ytel.hwmon_dir(bus)
            thr = ytel.Throttle(bus)
            smp = ytel.HwmonSampler(hw) if hw else None
            if smp:
                smp.start()
            t_a = time.perf_counter(); snap_a = thr.snapshot()
            n_sus = 0
            while time.perf_counter() - t_a < 1.5:
                run_steps(2 * lanes); n_sus += 2 * lanes
            fence()
            t_b = time.perf_counter(); snap_b = thr.snapshot()
            if smp:
                smp.stop = True; smp.join()
            telemetry = {"window": f"{n_sus} further steps of the timed workload, {t_b - t_a:.2f} s, outside the timed region",
                         "ms_per_step_in_window": (t_b - t_a) / max(1, n_sus) * 1e3, "pci_bus": bus,
                         "hwmon": smp.window(t_a + 0.3, t_b) if smp else None,
                         "power_cap_W": (int(open(os.path.join(hw, "power1_cap")).read()) / 1e6) if hw and os.path.exists(os.path.join(hw, "power1_cap")) else None,
                         "throttle": thr.between(snap_a, snap_b) if snap_a and snap_b else {"error": thr.error}}
        except Exception as e:      # noqa: BLE001 - telemetry never fails the bench
            telemetry = {"error": repr(e)}
    fallbacks = diag["exact_fallback_queries"] * a.steps
    ms_per_step = dt / a.steps * 1e3
    qps_resident = nq * a.steps / dt                       # queries/s against the rows resident on the N GPUs
    # `value`: queries/s against the 100M-row headline corpus = aggregate (rows x queries)/s / 1e8.
    # At N = 8 (8 x 12.5M rows) that IS the measured QPS of the sharded job; at N < 8 the same
    # aggregate rate covers N/8 of the corpus, so the whole-job number grows with N (weak scaling).
    qps = qps_resident * (total_rows / HEADLINE_ROWS)

    # ---- second roofline leg: the HBM-bound regime (Q = 64, narrow kernel form), N = 1 ------------
    hbm_leg = None
    bf16 = (not a.f32_filter) and d % 16 == 0
    tr = 256 if bf16 else 128
    n_tiles = (n + tr - 1) // tr
    s_target = min(n, max(n // 64, 8192))
    stride = max(1, n_tiles // ((s_target + tr - 1) // tr))
    n_sample = (n_tiles + stride - 1) // stride
    filt_rows = min(n, (n_tiles - n_sample) * tr)
    if world == 1 and not a.no_hbm_leg and (tb is not None or t8 is not None) and bf16 and nq >= 64:
        q64 = 64
        s64 = torch.empty((q64, k), dtype=torch.float32, device=dev)
        r64 = torch.empty((q64, k), dtype=torch.int64, device=dev)
        c64 = torch.empty(q64, dtype=torch.int32, device=dev)

        def step64():
            acc.scan_topk_device(view, tq.data_ptr(), q64, k, -1.0, SCAN_COSINE, s64.data_ptr(), r64.data_ptr(),
                                 c64.data_ptr(), flags=scan_flags, want_diag=False)
        for _ in range(max(2, a.warmup)):
            step64()
        acc.enable_timing(True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(a.steps):
            step64()
        torch.cuda.synchronize(); dt64 = (time.perf_counter() - t1) / a.steps
        f64_ms, f64_n = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        d64 = acc.scan_topk_device(view, tq.data_ptr(), q64, k, -1.0, SCAN_COSINE, s64.data_ptr(), r64.data_ptr(),
                                   c64.data_ptr(), flags=scan_flags, want_diag=True)
        i8_64 = d64.get("filter_tier") == 1
        if i8_64:      # the int8 shadow, streamed once by the resident-query kernel (one query tile: every workgroup has its own row stream)
            byts = filt_rows * d + (filt_rows // 64) * 8 + q64 * d
            k64 = ("scan_tiles_i8d_kernel" if d in (384, 768) else "scan_tiles_i8r_kernel") + " (int8 shadow, one 128-query tile resident per CU, 256 row streams)"
        else:
            byts = filt_rows * d * 2 + filt_rows * 4 + q64 * d * 2
            k64 = "scan_tiles_bf16n_kernel<FILTER,COSINE,2> (narrow form over the bf16 shadow, Q <= 64)"
        same = bool(torch.equal(r64, res["rows"][:q64]) and torch.equal(s64, res["scores"][:q64]))
        hbm_leg = {"bound": "hbm", "kernel": k64,
                   "queries": q64, "achieved": byts / (f64_ms * 1e-3) / 1e9 if f64_ms else None, "peak": PEAK_HBM_GBPS,
                   "unit": "GB/s", "frac": byts / (f64_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if f64_ms else None,
                   "algorithmic_bytes_per_launch": byts, "launch_ms": f64_ms, "launches": f64_n,
                   "traffic": hbm_leg_traffic(n, d, q64, i8_64)[0], "traffic_source": hbm_leg_traffic(n, d, q64, i8_64)[1],
                   "ms_per_step": dt64 * 1e3, "qps_on_resident_corpus": q64 / dt64,
                   "plain_read_ubench": plain_read_ceiling(byts / (f64_ms * 1e-3) / 1e9 if f64_ms else None),
                   "results_identical_to_the_q1024_run": same,
                   "bytes_definition": ("the bytes this sweep has to read: the 1-byte-per-element int8 shadow built at upload "
                                        "(+ 2 B bf16 shadow: 1.75x the fp32 rows resident per shard), NOT SURVEY 8(d)'s fp32 rows "
                                        "(38.4 GB per shard) — against those the sweep reads 4x fewer bytes; the fp32 rows are read "
                                        "only by the fp64 re-score of the candidates") if i8_64 else
                                       "the 2-byte-per-element bf16 shadow built at upload, not SURVEY 8(d)'s fp32 rows",
                   "fp32_rows_bytes": n * d * 4}

    # ---- the same shard at 256 queries per batch (Q in {64, 256, 1024} on config 4's shard) ----------
    q256_leg = None
    if world == 1 and not a.no_hbm_leg and (tb is not None or t8 is not None) and bf16 and nq >= 256:
        qm = 256
        sm = torch.empty((qm, k), dtype=torch.float32, device=dev); rm = torch.empty((qm, k), dtype=torch.int64, device=dev)
        cm = torch.empty(qm, dtype=torch.int32, device=dev)

        def stepm():
            acc.scan_topk_device(view, tq.data_ptr(), qm, k, -1.0, SCAN_COSINE, sm.data_ptr(), rm.data_ptr(), cm.data_ptr(),
                                 flags=scan_flags, want_diag=False)
        for _ in range(max(2, a.warmup)):
            stepm()
        acc.enable_timing(True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(a.steps):
            stepm()
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t1) / a.steps
        fm_ms, fm_n = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        flops_m = 2.0 * qm * d * filt_rows
        q256_leg = {"queries": qm, "ms_per_step": dtm * 1e3, "qps_on_resident_corpus": qm / dtm,
                    "value": qm / dtm * (n / HEADLINE_ROWS), "unit": "QPS", "launch_ms": fm_ms, "launches": fm_n,
                    "achieved_TOPs": flops_m / (fm_ms * 1e-3) / 1e12 if fm_ms else None,
                    "achieved_GBps": (filt_rows * d) / (fm_ms * 1e-3) / 1e9 if fm_ms else None,
                    "what": "one batch in flight (one lane): two query tiles per row stream, between the HBM-bound and the MFMA-bound end",
                    "results_identical_to_the_q1024_run": bool(torch.equal(rm, res["rows"][:qm]) and torch.equal(sm, res["scores"][:qm]))}

    # ---- BASELINE config 3 on the same resident rows: 10M x 768, L2 top-k, 1024 queries, N = 1 ------
    # (a prefix of the shard and of its shadows — the blocked int8 shadow of the first 10M rows is a prefix too)
    l2_leg = None
    n3 = 10_000_000
    if (world == 1 and not a.no_l2_leg and tb is not None and t8 is not None and n >= n3 and d == 768 and nq >= 1024
            and scan_flags == 0):
        view3 = acc.corpus_view(tc.data_ptr(), n3, d, row_base=row_base, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                                rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
        q3 = 1024
        s3 = torch.empty((q3, k), dtype=torch.float32, device=dev); r3 = torch.empty((q3, k), dtype=torch.int64, device=dev)
        c3 = torch.empty(q3, dtype=torch.int32, device=dev); d3 = torch.empty((q3, k), dtype=torch.float32, device=dev)

        def step3(flags=0, want_diag=False, out=(s3, r3, c3, d3)):
            return acc.scan_topk_device(view3, tq.data_ptr(), q3, k, -1.0, SCAN_L2, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), flags=flags, want_diag=want_diag)
        # the same search lanes as the headline loop (one context / stream / host thread each, sweeps behind the gate):
        # batch i on lane i % lanes, its own query batch and result buffers
        outs3 = [(s3, r3, c3, d3)] + [(torch.empty_like(s3), torch.empty_like(r3), torch.empty_like(c3), torch.empty_like(d3))
                                      for _ in range(lanes - 1)]

        def run3(count):
            errs = []

            def lane_fn(lane):
                try:
                    if dev.type == "cuda":
                        torch.cuda.set_device(dev)
                    o = outs3[lane]
                    for i in range(lane, count, lanes):
                        accs[lane].scan_topk_device(view3, tqs[i % n_qb].data_ptr(), q3, k, -1.0, SCAN_L2, o[0].data_ptr(), o[1].data_ptr(),
                                                    o[2].data_ptr(), o[3].data_ptr(), flags=0, want_diag=False)
                except BaseException as e:   # noqa: BLE001 - re-raised on the main thread
                    errs.append(e)
            if lanes == 1:
                lane_fn(0)
            else:
                th = [threading.Thread(target=lane_fn, args=(l,)) for l in range(lanes)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
            if errs:
                raise errs[0]
        run3(max(2, a.warmup))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        run3(a.steps)
        torch.cuda.synchronize(); dt3 = (time.perf_counter() - t1) / a.steps
        step3()                         # (the checks below look at THE batch tq in the first lane's buffers)
        acc.enable_timing(True)
        dg3 = step3(want_diag=True)
        torch.cuda.synchronize()
        f3_ms, _ = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        # the same batch through the bf16 tier: every query, bit for bit (the full-size oracle check of this shape is
        # tests/test_scan_gpu.py::test_full_size_config3_10Mx768_l2_top100_q1024)
        sb3 = torch.empty_like(s3); rb3 = torch.empty_like(r3); cb3 = torch.empty_like(c3); db3 = torch.empty_like(d3)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dgb = step3(flags=FLAG_NO_I8_FILTER, want_diag=True, out=(sb3, rb3, cb3, db3))
        torch.cuda.synchronize(); dtb = time.perf_counter() - t1
        tr3 = (n3 + 255) // 256
        st3 = max(1, tr3 // ((min(n3, max(n3 // 64, 8192)) + 255) // 256))
        filt3 = (tr3 - (tr3 + st3 - 1) // st3) * 256
        ops3 = 2.0 * q3 * d * filt3
        i8_3 = dg3.get("filter_tier") == 1
        l2_leg = {"workload": "BASELINE config 3: 10M x 768 fp32, exact L2 top-%d, %d queries per batch (the first 10M resident rows)" % (k, q3),
                  "ms_per_step": dt3 * 1e3, "qps": q3 / dt3, "search_lanes": lanes, "filter_tier": dg3.get("filter_tier"),
                  "kernel": ("scan_tiles_i8r_kernel<L2> (int8 shadow, per-row integer thresholds)" if i8_3
                             else "scan_tiles_bf16s_kernel<FILTER,L2>"),
                  "bound": "mfma", "launch_ms": f3_ms, "achieved": ops3 / (f3_ms * 1e-3) / 1e12 if f3_ms else None,
                  "peak": PEAK_I8_MFMA_TOPS if i8_3 else PEAK_BF16_MFMA_TFLOPS, "unit": "TOP/s" if i8_3 else "TFLOP/s",
                  "frac": ops3 / (f3_ms * 1e-3) / 1e12 / (PEAK_I8_MFMA_TOPS if i8_3 else PEAK_BF16_MFMA_TFLOPS) if f3_ms else None,
                  "filter_candidates": dg3.get("filter_candidates"), "widened_queries": dg3.get("widened_queries"),
                  "escalated_queries": dg3.get("escalated_queries"), "exact_fallback_queries": dg3.get("exact_fallback_queries"),
                  "bf16_tier_ms_per_step_one_cold_call": dtb * 1e3, "bf16_tier_filter_tier": dgb.get("filter_tier"),
                  "results_identical_to_the_bf16_tier": bool(torch.equal(r3, rb3) and torch.equal(s3, sb3) and torch.equal(c3, cb3)
                                                             and torch.equal(d3, db3))}
        # L2 parity is UNPINNED (the vec0 arithmetic lives in the absent sqlite-vec-cpp): this repository accumulates in fp64,
        # the dependency most likely in fp32.  How many of the 1024 top-k sets would differ?  The device's top 2k under our
        # definition, re-measured with fp32 accumulation on the host (sequential / 8 / 16 SIMD lanes): a REPORT of the size
        # of the gap, not a parity claim.
        try:
            K2 = 2 * k
            s2 = torch.empty((q3, K2), dtype=torch.float32, device=dev); r2 = torch.empty((q3, K2), dtype=torch.int64, device=dev)
            c2 = torch.empty(q3, dtype=torch.int32, device=dev); d2 = torch.empty((q3, K2), dtype=torch.float32, device=dev)
            acc.scan_topk_device(view3, tq.data_ptr(), q3, K2, -1.0, SCAN_L2, s2.data_ptr(), r2.data_ptr(), c2.data_ptr(), d2.data_ptr())
            _o = oracle_mod()
            rep = _o.l2_definition_report(_o.oracle(), lambda rows: tc[torch.from_numpy(rows - row_base).to(dev)].cpu().numpy(),
                                          tq[:q3].cpu().numpy(), r2.cpu().numpy(), d2.cpu().numpy(), k)
            rep["note"] = ("L2 parity is unpinned: third_party/sqlite-vec-cpp is absent from the reference checkout; `bit-identical` "
                           "above means identical to THIS repository's fp64-accumulate definition (oracle/yams_oracle.c)")
            l2_leg["definition_gap_report"] = rep
            del s2, r2, c2, d2
        except Exception as e:      # noqa: BLE001
            l2_leg["definition_gap_report"] = {"error": repr(e)}
        # ... and its HBM-bound point: the same shard at 64 queries (one resident query tile, every CU streams its own rows)
        q3s = 64
        def step3s(want_diag=False):
            return acc.scan_topk_device(view3, tq.data_ptr(), q3s, k, -1.0, SCAN_L2, s3.data_ptr(), r3.data_ptr(), c3.data_ptr(),
                                        d3.data_ptr(), want_diag=want_diag)
        for _ in range(max(2, a.warmup)):
            step3s()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(a.steps):
            step3s()
        torch.cuda.synchronize(); dt3s = (time.perf_counter() - t1) / a.steps
        acc.enable_timing(True)
        dg3s = step3s(want_diag=True)
        torch.cuda.synchronize()
        f3s_ms, _ = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        byts3 = filt3 * d + (filt3 // 64) * (8 + 64) + q3s * d          # int8 shadow + thresholds meta + row biases + queries
        l2_leg["q64"] = {"ms_per_step": dt3s * 1e3, "qps": q3s / dt3s, "filter_tier": dg3s.get("filter_tier"), "bound": "hbm",
                         "launch_ms": f3s_ms, "algorithmic_bytes_per_launch": byts3,
                         "achieved": byts3 / (f3s_ms * 1e-3) / 1e9 if f3s_ms else None, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": byts3 / (f3s_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS if f3s_ms else None,
                         "results_identical_to_the_q1024_run": bool(torch.equal(r3[:q3s], rb3[:q3s]) and torch.equal(d3[:q3s], db3[:q3s]))}
        del s3, r3, c3, d3, sb3, rb3, cb3, db3

    # ---- the same workload through the C ABI's sharded entry points (what a C++ host calls) ----------
    c_abi = None
    if not a.no_c_abi_leg and scan_flags == 0 and t8 is not None:
        if world == 1:
            try:
                c_abi = c_abi_sharded_run(a, [local], views=[view], n_query_batches=n_qb, oracle_queries=0)
                c_abi["identical_to_the_timed_step"] = c_abi["result_digests"].get(str(last_batch)) == \
                    result_digest(r_timed, s_timed, c_timed)
                c_abi["note"] = ("the resident shard of this run, handed over as a view; a communicator of ONE rank "
                                 "(all-gather + merge still run per batch)")
            except Exception as e:      # noqa: BLE001 - the headline number above stands on its own
                c_abi = {"error": repr(e)}
        else:
            # N > 1 under torchrun: rank 0 starts ONE more process that drives all N devices through the C ABI while the
            # ranks wait on the HOST (a gloo barrier: an RCCL barrier would spin a kernel on every device meanwhile)
            import subprocess
            hostpg = torch.distributed.new_group(backend="gloo")
            fence()
            torch.distributed.barrier(group=hostpg)
            if rank == 0:
                cmd = [sys.executable, os.path.abspath(__file__), "--via-c-abi", "--child-json", "--gpus", str(world),
                       "--steps", str(a.steps), "--warmup", str(a.warmup), "--rows-per-gpu", str(n), "--dim", str(d),
                       "--queries", str(nq), "--k", str(k), "--lanes", str(lanes), "--seed", str(a.seed),
                       "--query-batches", str(n_qb)] + (["--single-device"] if a.single_device else [])
                env = dict(os.environ)
                for kk in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                           "TORCHELASTIC_RUN_ID"):
                    env.pop(kk, None)
                try:
                    cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
                    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                    if cp.returncode == 0 and line:
                        c_abi = json.loads(line[-1])
                        c_abi["identical_to_the_timed_step"] = c_abi["result_digests"].get(str(last_batch)) == \
                            result_digest(r_timed, s_timed, c_timed)
                        c_abi["note"] = "a second process on the same node, started by rank 0 after the timed region; the ranks idle on the host meanwhile"
                    else:
                        c_abi = {"error": f"child exited {cp.returncode}", "stderr_tail": cp.stderr[-1500:]}
                except Exception as e:  # noqa: BLE001
                    c_abi = {"error": repr(e)}
            torch.distributed.barrier(group=hostpg)

    if rank != 0:
        return
    # ---- roofline of the dominant kernel (the FILTER pass of the scan) -----------------------------
    flops = 2.0 * nq * d * filt_rows            # ALGORITHMIC flops of the contraction per launch
    ach_tf = flops / (filt_ms * 1e-3) / 1e12 if filt_ms else None
    passes = 3 if (a.split_filter or 3 * k + 64 > 2047) else 1
    i8 = t8 is not None and nq > 128 and passes == 1 and diag.get("filter_tier") == 1
    if i8:
        # the library's own choice of kernel form (scan_i8_kernel.hip, i8_resident_plan), restated for the label
        n_qt = (nq + 127) // 128
        n_streams = (32 // n_qt) * 8 if n_qt <= 32 else 0
        n_filter_tiles = n_tiles - n_sample
        resident = (not a.half_tile and d % 128 == 0 and 384 <= d <= 768 and 2 <= n_qt <= 8 and (32 // n_qt) * n_qt * 10 >= 32 * 9
                    and (n_filter_tiles + 1) // 2 >= 12 * n_streams)
        if resident and d in (384, 768):
            kname = ("scan_tiles_i8d_kernel (v_mfma_i32_16x16x64_i8 over the int8 shadow; 128-query tile resident in LDS, one persistent "
                     "workgroup of eight waves per CU, two slabs of row fragments in flight per wave, "
                     "strips drawn per SIMD pair, exact integer accumulate)")
        elif resident:
            kname = ("scan_tiles_i8r_kernel (v_mfma_i32_16x16x64_i8 over the int8 shadow; 128-query tile resident in LDS, one persistent "
                     "workgroup of eight waves per CU, row fragments loaded straight into the register double buffer, "
                     "strips drawn per SIMD pair, exact integer accumulate)")
        else:
            kname = "scan_tiles_i8h_kernel<FILTER> (v_mfma_i32_16x16x64_i8 over the int8 shadow, 128 x 256 half tiles, two workgroups per CU, exact integer accumulate)"
        peak = PEAK_I8_MFMA_TOPS
    elif bf16 and passes == 3:
        kname = "scan_tiles_bf16v2_kernel<FILTER,COSINE,3> (v_mfma_f32_32x32x16_bf16, split hi/lo x3)"
        peak = PEAK_BF16_MFMA_TFLOPS / BF16_PASSES  # each algorithmic multiply-add costs 3 bf16 MFMA passes
    elif bf16:
        kname = ((("scan_tiles_bf16p_kernel (persistent)" if 256 <= d <= 512 else "scan_tiles_bf16s_kernel<FILTER,COSINE>")
                  if tb is not None else "scan_tiles_bf16k32_kernel<FILTER,COSINE>")
                 if d % 32 == 0 else "scan_tiles_bf16v2_kernel<FILTER,COSINE,1>") \
            + " (v_mfma_f32_32x32x16_bf16, RNE bf16 operands, one pass" + (", bf16 corpus shadow)" if tb is not None else ")")
        peak = PEAK_BF16_MFMA_TFLOPS
    else:
        kname = "scan_tiles_kernel<FILTER,COSINE> (v_mfma_f32_32x32x2_f32)"
        peak = PEAK_F32_MFMA_TFLOPS
    traffic = None
    traffic_source = None
    pmc = os.path.join(ROOT, "profiles", "scan_filter_pmc.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("rows_per_gpu") == n and j.get("dim") == d and j.get("queries") == nq \
                    and j.get("bf16", False) == bf16 and j.get("passes", 3) == (passes if bf16 else 0) \
                    and j.get("shadow", False) == (tb is not None) and j.get("i8", False) == i8:
                traffic = j.get("hbm_bytes_per_launch")
                traffic_source = f"profiles/scan_filter_pmc.json ({j.get('round', '?')}, builder run: rocprofv3 --pmc FETCH_SIZE x2 " \
                                 "gfx950 correction; NOT measured in this run)"
        except Exception:
            traffic = None
    row_bytes = 1 if i8 else (2 if (tb is not None and passes == 1 and bf16) else 4)   # what the filter reads per element
    roofline = {"bound": "mfma", "kernel": kname, "achieved": ach_tf, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                "frac": (ach_tf / peak) if ach_tf else None, "traffic": traffic, "traffic_source": traffic_source,
                "launch_ms": filt_ms, "launches": filt_n, "flops_per_launch": flops,
                "executed_mfma_tflops": (ach_tf * passes if bf16 else ach_tf) if ach_tf else None,
                "mfma_peak_for_executed": PEAK_I8_MFMA_TOPS if i8 else (PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS),
                "frac_of_bf16_peak": (ach_tf / PEAK_BF16_MFMA_TFLOPS) if ach_tf else None,
                # MI355X_MICROARCH.md lists no spec peak for I8, only a micro-benchmark ceiling (>= 3944 TOPS, 16x16x64);
                # `peak` above is the nominal 2x-bf16 figure (5000), the stricter of the two
                "frac_of_guide_i8_ubench_ceiling": (ach_tf / 3944.0) if (ach_tf and i8) else None,
                "sample_pass_ms": samp_ms,
                "shadow_build_ms": shadow_ms, "shadow_i8_build_ms": shadow_i8_ms, "shadow_i8_mean_residue": i8_mean_err, "shadow_i8_layout": "rotated" if i8_flags & 1 else "plain",
                "shadow_i8_sampled_residue": {"plain": i8_res_plain, "rotated": i8_res_rot},
                # the other floor of this kernel: one read of the filter rows from HBM per launch
                "hbm_floor_view": {"algorithmic_bytes_per_launch": filt_rows * d * row_bytes,
                                   "achieved_GBps": (filt_rows * d * row_bytes) / (filt_ms * 1e-3) / 1e9 if filt_ms else None,
                                   "peak_GBps": PEAK_HBM_GBPS},
                "hbm_view": {"algorithmic_bytes_per_step": n * d * row_bytes + nq * d * 4 + nq * k * 12,
                             "achieved_GBps": (n * d * row_bytes + nq * d * 4 + nq * k * 12) / (ms_per_step * 1e-3) / 1e9,
                             "peak_GBps": PEAK_HBM_GBPS}}
    out = {"metric": "k-NN QPS + recall@k, 100M x 768 fp32 cosine top-100; ingest GB/s SHA-256+CDC",
           "value": qps, "unit": "QPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None,
           "dtype": ("i8 (MFMA filter, exact i32 accumulate) + f64 (exact re-score)" if i8 else
                     (("bf16" if passes == 1 else "bf16x3 split-f32") + " (MFMA filter) + f64 (exact re-score)" if bf16 else "f32 (MFMA filter) + f64 (exact re-score)")),
           "data": "synthetic",
           "config": {"workload": f"{n}x{d} fp32 cosine top-{k} per GPU (row shard of BASELINE config 4: "
                                  f"100Mx768 over 8 GPUs), query batch {nq}",
                      "rows_per_gpu": n, "corpus_rows": total_rows, "dim": d, "k": k, "query_batch": nq,
                      "parallelism": f"row-shard x{world} + RCCL all-gather top-k merge (overlapped with the next sweep)" if world > 1 else "single shard",
                      "search_lanes": lanes, "distinct_query_batches": n_qb},
           "value_definition": "queries/s against the 100M x 768 headline corpus = (rows scored x queries)/s / 1e8; "
                               "equals qps_on_resident_corpus x corpus_rows / 1e8 (identical at N = 8)",
           "qps_on_resident_corpus": qps_resident,
           "row_queries_per_s": total_rows * nq * a.steps / dt,
           "exact_fallback_queries": fallbacks,
           "scan_diag_per_batch": {kk: diag[kk] for kk in ("filter_tier", "filter_candidates", "rescored_rows", "widened_queries",
                                                             "escalated_queries", "exact_fallback_queries")},
           "roofline": roofline}
    if telemetry is not None:
        roofline["telemetry"] = telemetry
        hw_ = (telemetry.get("hwmon") or {})
        roofline["sclk_MHz"] = (hw_.get("sclk_MHz") or {}).get("mean")
        roofline["power_W"] = (hw_.get("power_W") or {}).get("mean")
        roofline["limiter"] = (telemetry.get("throttle") or {}).get("limiter")
    if c_abi is not None:
        out["c_abi_sharded"] = c_abi
    if hbm_leg is not None:
        out["roofline_hbm_leg"] = hbm_leg
    if q256_leg is not None:
        out["config4_shard_q256"] = q256_leg
    if l2_leg is not None:
        out["config3_l2"] = l2_leg
    if check is not None:
        out.update(check)
    if world > 1:
        lm = [r_["launch_ms"] for r_ in rank_stats if r_.get("launch_ms") is not None]
        em = [r_["exchange_ms"] for r_ in rank_stats if r_.get("exchange_ms") is not None]
        out["collective"] = {"backend": torch.distributed.get_backend(), "bytes_per_rank": pipe.rec_bytes,
                             "communicator_ranks": torch.distributed.get_world_size(),
                             "collectives": ex_stats["collectives"], "batches": a.steps,
                             "exchange_ms": (sum(em) / len(em)) if em else None,
                             "exchange_ms_max": max((r_["exchange_ms_max"] or 0.0) for r_ in rank_stats) if rank_stats else None,
                             "exchange_what": "per batch, on each rank's side stream: events around all_gather_into_tensor + merge_topk_kernel "
                                              "(the wait for the slowest rank's scan included); mean over ranks, max over ranks and batches",
                             "launch_ms_min": min(lm) if lm else None, "launch_ms_max": max(lm) if lm else None,
                             "per_rank": rank_stats, "watchdog_s": a.watchdog_s,
                             "side_stream": pipe.side is not None and torch.distributed.get_backend() != "gloo",
                             "fenced": bool(gate is not None and pipe.active),
                             "fence": "the sweep gate stays closed behind a lane's sweep until that lane has enqueued its all-gather + merge: "
                                      "no collective kernel waits for a CU under a persistent sweep (DESIGN 4)"}
    # CPU baseline and the ingest leg: rank 0 at N = 1 only (at N > 1 the other ranks would sit in the
    # final barrier meanwhile)
    if not a.no_boundary_leg and world == 1:
        try:
            free_b, _ = torch.cuda.mem_get_info()
            need = n * d * 7 * 1.15                      # the plugin's own mirror of the shard: fp32 + bf16 + int8
            out["boundary"] = boundary_leg(a, acc, torch, dev, tc, n if free_b > need + (8 << 30) else 0)
        except Exception as e:          # noqa: BLE001
            out["boundary"] = {"error": repr(e)}
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_scan(tc, tq, n, k)
    if not a.no_distribution_legs and world == 1 and n >= 1_000_000:
        # the same shape on corpora that are not uniform on the sphere (after the headline's tensors are gone: each leg builds
        # a shard of its own)
        pending_distributions = [kd for kd in (a.distribution or "clustered,anisotropic,gaussian").split(",") if kd]
    else:
        pending_distributions = []
    if not a.no_config2_leg and world == 1:
        try:
            sweep = [int(x) for x in a.config2_lane_sweep.split(",")] if a.config2_lane_sweep else None
            out["config2"] = config2_leg(a, torch, dev, local, lane_counts=sweep, batches=a.config2_batches)
        except Exception as e:          # noqa: BLE001 - the headline number above stands on its own
            out["config2"] = {"error": repr(e)}
    if not a.no_pq_leg and world == 1:
        try:
            out["pq"] = pq_leg(a, torch, dev, local)
        except Exception as e:          # noqa: BLE001 - the headline number above stands on its own
            out["pq"] = {"error": repr(e)}
    if not a.no_ingest and world == 1:
        del tc, tb, tn, t8, tm8, view, pipe, res
        acc.L.yams_accel_ctx_destroy(acc.ctx)   # drop the scan workspace before the ingest leg
        acc.ctx = None
        acc = Accel(local, torch.cuda.current_stream().cuda_stream)
        torch.cuda.empty_cache()
        out["ingest"] = ingest_leg(acc, torch, a.ingest_gib, a.seed, verify_all=a.verify_all_ingest)
    if pending_distributions:
        try:
            del tc, tb, tn, t8, tm8, view, pipe, res
        except NameError:
            pass
        torch.cuda.empty_cache()
        nu = {}
        for kd in pending_distributions:
            try:
                nu[kd] = distribution_leg(a, torch, dev, local, kd)
            except Exception as e:      # noqa: BLE001 - the headline number above stands on its own
                nu[kd] = {"error": repr(e)}
        out["non_uniform"] = nu
    emit(out, a.extra_json)


if __name__ == "__main__":
    main()
    import torch
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
