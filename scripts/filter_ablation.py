"""Filter-kernel ablations (measurement only): times the FILTER launch alone under
YAMS_ACCEL_BF16_KERNEL = 2 (product), 11 (no refills), 12 (no MFMA), 13 (no refills, no MFMA),
14 (no refills, no fragment reads).  Needs the measurement build of the library
(`python -m yams_amd.build --measure`): the product .so has neither the knob nor the ablation kernels."""
import json, os, sys
os.environ["YAMS_ACCEL_MEASURE_LIB"] = "1"   # the measurement build: python -m yams_amd.build --measure
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE

n, d, nq, k = int(os.environ.get("ROWS", 12_500_000)), int(os.environ.get("DIM", 768)), int(os.environ.get("Q", 1024)), 100
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
c = torch.empty(nq, dtype=torch.int32, device="cuda")
tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr()); acc.synchronize()
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
view = acc.corpus_view(tc.data_ptr(), n, d) if os.environ.get("NO_SHADOW") else \
    acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr())
view8 = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                        rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
out = {}
for v in sys.argv[1:]:
    os.environ["YAMS_ACCEL_BF16_KERNEL"] = v
    acc.enable_timing(True)   # (the host returns right after the filter launch for ablated kernels)
    i8v = v.startswith("i8:")          # "i8:2" = the int8 tier's product kernel, "i8:31/32/37" its ablations
    if i8v:
        os.environ["YAMS_ACCEL_BF16_KERNEL"] = v[3:]
    acc.scan_topk_device(view8 if i8v else view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    out[v] = acc.kernel_ms("scan_filter")[0]
    acc.enable_timing(False)
print(json.dumps(out))
