import os, sys, json
os.environ["YAMS_ACCEL_MEASURE_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE
n, d, nq, k = 12_500_000, 768, 1024, 100
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
view = acc.corpus_view(tc.data_ptr(), n, d, rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda"); c = torch.empty(nq, dtype=torch.int32, device="cuda")
for env in (None, "1", None, "1"):
    if env: os.environ["YAMS_ACCEL_SAMPLE_NODENSE"] = env
    else: os.environ.pop("YAMS_ACCEL_SAMPLE_NODENSE", None)
    for _ in range(2): acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    acc.enable_timing(True)
    for _ in range(5): acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    print("nodense" if env else "dense", acc.kernel_ms("scan_sample")[0]); acc.enable_timing(False)
