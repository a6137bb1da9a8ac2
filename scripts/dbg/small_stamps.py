"""Phase stamps of scan_small_kernel (measurement build): per workgroup, 100 MHz wall clock."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["YAMS_ACCEL_MEASURE_LIB"] = "1"
os.environ["YAMS_ACCEL_SMALL_STAMPS"] = "/tmp/small_stamps.bin"
import numpy as np, torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
d, k = 384, 10
for n, nq in ((4096, 1), (10000, 1), (10000, 4)):
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
    view = acc.corpus_view(tc.data_ptr(), n, d)
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda")
    for _ in range(20):
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), want_diag=False)
    st = np.fromfile("/tmp/small_stamps.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0   # microseconds
    print(f"n={n} nq={nq} wgs={len(st)}")
    print(" start  : min %.1f max %.1f" % (rel[:, 0].min(), rel[:, 0].max()))
    for i, name in ((1, "queries in LDS"), (2, "pass 0 issued"), (3, "walk done"), (4, "wg top-k written"), (5, "ticket taken")):
        dlt = rel[:, i] - rel[:, 0]
        print(" %-18s: since wg start min %.1f med %.1f max %.1f | abs max %.1f" % (name, dlt.min(), np.median(dlt), dlt.max(), rel[:, i].max()))
    last = np.argmax(st[:, 7])
    print(" last wg %d: fence %.1f final done %.1f (abs)" % (last, rel[last, 6], rel[last, 7]))
