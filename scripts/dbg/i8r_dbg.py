import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from yams_amd.accel import Accel
from yams_amd import _lib
from yams_amd._lib import SCAN_COSINE
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
def one(n, d, nq, k, flags):
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(35, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(35, 1 << 40, nq, d, tq.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda")
    diag = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=flags, want_diag=True)
    return diag["filter_candidates"], diag["rescored_rows"], diag["exact_fallback_queries"], r.cpu().numpy()
for (n, d, nq) in [(20000, 256, 1), (20000, 256, 16), (200000, 256, 1), (200000, 256, 128), (200000, 256, 130), (20000, 768, 1), (1000000, 256, 1)]:
    a = one(n, d, nq, 10, _lib.FLAG_RESIDENT_QUERIES); b = one(n, d, nq, 10, _lib.FLAG_WIDE_TILE)
    print(n, d, nq, "resident", a[:3], "half", b[:3], "same rows", bool((a[3] == b[3]).all()))
