"""Times the BASELINE.json configurations that fit one GPU (informational table for DESIGN.md;
bench.py remains the contract).  C1 10k x 384 cosine k=10 Q=1 | C2 1M x 384 cosine k=100 Q=256 |
C3 10M x 768 L2 k=100 Q in {64, 1024} | C4-shard 12.5M x 768 cosine k=100 Q in {64, 256, 1024}."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE, SCAN_L2

acc = Accel(0, torch.cuda.current_stream().cuda_stream)
rows = []
only = os.environ.get("ONLY")   # e.g. ONLY=C2
for name, n, d, nq, k, metric in [("C1", 10_000, 384, 1, 10, SCAN_COSINE), ("C2", 1_000_000, 384, 256, 100, SCAN_COSINE),
                                  ("C3", 10_000_000, 768, 1024, 100, SCAN_L2), ("C3 Q=64", 10_000_000, 768, 64, 100, SCAN_L2), ("C4/8 Q=64", 12_500_000, 768, 64, 100, SCAN_COSINE),
                                  ("C4/8 Q=256", 12_500_000, 768, 256, 100, SCAN_COSINE), ("C4/8 Q=1024", 12_500_000, 768, 1024, 100, SCAN_COSINE)]:
    if only and not name.startswith(only):
        continue
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda"); dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr()); acc.synchronize()
    t8 = tm8 = None     # both shadows, as the plugin's mirrors carry them ("shadows": "both")
    if d % 64 == 0 and d >= 256:
        t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
        acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                           rows_i8_ptr=t8.data_ptr() if t8 is not None else None,
                           rows_i8_meta_ptr=tm8.data_ptr() if tm8 is not None else None)
    for _ in range(2):
        diag = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, metric, s.data_ptr(), r.data_ptr(), c.data_ptr(), dist.data_ptr())
    reps = 10 if n <= 1_000_000 else 4
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, metric, s.data_ptr(), r.data_ptr(), c.data_ptr(), dist.data_ptr(), want_diag=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    acc.enable_timing(True)     # (HIP events around the two sweeps: serialises the call, so outside the timed loop)
    acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, metric, s.data_ptr(), r.data_ptr(), c.data_ptr(), dist.data_ptr(), want_diag=False)
    torch.cuda.synchronize()
    filt_ms, _ = acc.kernel_ms("scan_filter"); samp_ms, _ = acc.kernel_ms("scan_sample")
    acc.enable_timing(False)
    rows.append({"config": name, "rows": n, "dim": d, "Q": nq, "k": k, "metric": "l2" if metric else "cosine",
                 "ms": dt * 1e3, "QPS": nq / dt, "algorithmic_TFLOPs": 2.0 * n * d * nq / dt / 1e12,
                 "corpus_GBps": n * d * 4 / dt / 1e9, "path": diag["path"], "fallbacks": diag["exact_fallback_queries"], "escalated": diag["escalated_queries"],
                 "widened": diag["widened_queries"], "filter_tier": diag["filter_tier"], "filter_candidates": diag["filter_candidates"],
                 "filter_kernel_ms": filt_ms, "sample_kernel_ms": samp_ms})
    del tc, tq, tb, tn, t8, tm8, view
    torch.cuda.empty_cache()
for r_ in rows:
    print(json.dumps(r_))
