"""Per-wave begin / end times of the resident-query filter launch (measurement build, 100 MHz ticks): where does a SHORT launch
(BASELINE config 2: 1M x 384, 256 queries) lose its time — start-up ramp, uneven ends, or the steady state?"""
import json, os, sys
os.environ["YAMS_ACCEL_MEASURE_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE
n, d, nq, k = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("DIM", 384)), int(os.environ.get("Q", 256)), 100
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda"); c = torch.zeros(nq, dtype=torch.int32, device="cuda")
out = {}
for v in sys.argv[1:] or ["85"]:
    os.environ["YAMS_ACCEL_BF16_KERNEL"] = v
    for _ in range(4):
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    acc.enable_timing(True)
    for _ in range(6):
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    ms = acc.kernel_ms("scan_filter")[0]
    acc.enable_timing(False)
    os.environ["YAMS_ACCEL_DUMP_SYNC"] = "/tmp/wave_sync.bin"
    acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    acc.synchronize(); del os.environ["YAMS_ACCEL_DUMP_SYNC"]
    w = np.fromfile("/tmp/wave_sync.bin", dtype=np.uint32)
    w = w[: w.size // 2]                      # the filter launch's half
    ns = w.size // (12 * 32); n_qt = (nq + 127) // 128
    dbg = w[ns * 4 * 32:].reshape(ns, 8, 32)
    tb_ = dbg[:, :, 8:8 + n_qt].astype(np.int64); te = dbg[:, :, 16:16 + n_qt].astype(np.int64); un = dbg[:, :, 24:24 + n_qt]
    t0 = tb_.min()
    b = (tb_ - t0) / 100.0; e = (te - t0) / 100.0   # us
    out[v] = {"filter_ms": ms, "streams": int(ns), "n_qt": n_qt,
              "begin_us": {"min": float(b.min()), "p50": float(np.median(b)), "p90": float(np.percentile(b, 90)), "max": float(b.max())},
              "end_us": {"min": float(e.min()), "p10": float(np.percentile(e, 10)), "p50": float(np.median(e)), "max": float(e.max())},
              "busy_us_per_wave": {"mean": float((e - b).mean()), "min": float((e - b).min()), "max": float((e - b).max())},
              "strips_per_wave": {"min": int(un.min()), "mean": float(un.mean()), "max": int(un.max())},
              "us_per_strip": float(((e - b) / np.maximum(un, 1)).mean())}
print(json.dumps(out))
