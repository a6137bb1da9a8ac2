"""128 x 128 wave-tile form of the resident-query filter (measurement build, YAMS_ACCEL_BF16_KERNEL=70) against the product
form: identical results on the bench shard?  launch times?"""
import json, os, sys
os.environ["YAMS_ACCEL_MEASURE_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE
n, d, nq, k = int(os.environ.get("ROWS", 12_500_000)), int(os.environ.get("DIM", 768)), int(os.environ.get("Q", 1024)), 100
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
res = {}
out = {}
for v in sys.argv[1:]:
    os.environ["YAMS_ACCEL_BF16_KERNEL"] = v
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.zeros(nq, dtype=torch.int32, device="cuda")
    for _ in range(3):
        dg = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=True)
    acc.enable_timing(True)
    for _ in range(6):
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
    out[v] = {"filter_ms": acc.kernel_ms("scan_filter")[0], "candidates": dg.get("filter_candidates"), "fallback": dg.get("exact_fallback_queries"),
              "widened": dg.get("widened_queries")}
    acc.enable_timing(False)
    if v in ("70", "71", "72"):  # where the waves' time went (100 MHz ticks per wave and launch)
        import numpy as np
        os.environ["YAMS_ACCEL_DUMP_SYNC"] = "/tmp/i8q_sync.bin"
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=0, want_diag=False)
        acc.synchronize(); del os.environ["YAMS_ACCEL_DUMP_SYNC"]
        w = np.fromfile("/tmp/i8q_sync.bin", dtype=np.uint32)
        ns = w.size // (12 * 32); n_qt = (nq + 127) // 128
        dbg = w[ns * 4 * 32:].reshape(ns, 8, 32)[:, :4, :n_qt * 4].reshape(ns, 4, n_qt, 4).astype(np.float64) / 100.0  # us
        out[v]["phase_us_mean"] = dict(zip(("loop", "sign_test", "emission", "shader_MHz_x100"), [round(float(x), 1) for x in dbg.mean(axis=(0, 1, 2))]))
        out[v]["phase_us_max_wave_total"] = round(float(dbg.sum(axis=3).max()), 1)
        out[v]["strips_per_wave"] = int(w[:ns * 4 * 32].reshape(ns, 4, 32)[:, :, :n_qt].max())
    res[v] = (r.clone(), s.clone(), c.clone())
vs = list(res)
for v in vs[1:]:
    out[v]["identical_to_" + vs[0]] = bool(torch.equal(res[v][0], res[vs[0]][0]) and torch.equal(res[v][1], res[vs[0]][1]) and torch.equal(res[v][2], res[vs[0]][2]))
print(json.dumps(out))
