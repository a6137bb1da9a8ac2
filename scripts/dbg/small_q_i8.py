import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd import _lib
from yams_amd._lib import SCAN_COSINE
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
n, d, k = 12_500_000, 768, 100
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
vboth = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
v8 = acc.corpus_view(tc.data_ptr(), n, d, rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
for nq in (1, 16, 64, 128, 192):
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda"); c = torch.empty(nq, dtype=torch.int32, device="cuda")
    out = {}; res = {}
    for name, view, fl in (("both(default)", vboth, 0), ("i8 resident", v8, _lib.FLAG_RESIDENT_QUERIES), ("i8 half", v8, _lib.FLAG_WIDE_TILE)):
        for _ in range(3):
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
        acc.enable_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6):
            diag = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=True)
        torch.cuda.synchronize(); out[name] = (round((time.perf_counter() - t0) / 6 * 1e3, 3), round(acc.kernel_ms("scan_filter")[0], 3), diag["filter_tier"], diag["exact_fallback_queries"])
        acc.enable_timing(False)
        res[name] = r.clone()
    print(nq, out, "same:", bool(torch.equal(res["both(default)"], res["i8 resident"]) and torch.equal(res["both(default)"], res["i8 half"])))
