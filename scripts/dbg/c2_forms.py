import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd import _lib
from yams_amd._lib import SCAN_COSINE
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
for (n, d, nq) in [(12_500_000, 384, 1024), (12_500_000, 512, 1024), (12_500_000, 640, 1024), (12_500_000, 768, 256), (12_500_000, 768, 384), (12_500_000, 768, 512), (12_500_000, 768, 1024), (12_500_000, 768, 2048), (1_000_000, 384, 256), (2_000_000, 768, 1024)]:
    k = 100
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda"); c = torch.empty(nq, dtype=torch.int32, device="cuda")
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
    out = {}
    for name, fl in (("default", 0), ("resident", _lib.FLAG_RESIDENT_QUERIES), ("half", _lib.FLAG_WIDE_TILE), ("bf16", _lib.FLAG_NO_I8_FILTER)):
        for _ in range(3):
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
        reps = 20 if n <= 2_000_000 else 6
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=fl, want_diag=False)
        torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    print(n, d, nq, out)
    del tc, tq, tb, tn, t8, tm8, view; torch.cuda.empty_cache()
