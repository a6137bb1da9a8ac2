"""Small corpora (BASELINE config 1: 10k x 384, one query): the default path (<= 16384 rows and <= 16 queries: ONE fused
launch, scan_small_kernel.hip) against the MFMA filter pipeline and the multi-launch exhaustive fp64 pipeline; device-entry
latency per call (queries and results in HBM), and the fused kernel's own duration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE, FLAG_FORCE_EXACT, FLAG_NO_I8_FILTER
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
d, k = 384, 10
for n in (4096, 10_000, 30_000, 100_000):
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr())
    for nq in (1, 4, 16):
        tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
        s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        c = torch.empty(nq, dtype=torch.int32, device="cuda")
        out = {"n": n, "nq": nq}
        for name, fl in (("default", 0), ("mfma_pipeline", FLAG_NO_I8_FILTER), ("exact_pipeline", FLAG_FORCE_EXACT)):
            for _ in range(5):
                acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
            torch.cuda.synchronize(); out[name + "_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
            if name == "default":
                acc.enable_timing(True)
                for _ in range(10):
                    acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
                try:
                    ms, cnt = acc.kernel_ms("small_scan")
                except Exception:
                    ms = None
                acc.enable_timing(False)
                out["fused_kernel_us"] = round(ms * 1e3, 1) if ms else None
        print(json.dumps(out))
