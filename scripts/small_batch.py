"""Small query batches on one C4 shard (12.5M x 768 cosine, k = 100): the narrow form of the shadow
filter (Q <= 64 / <= 128, HBM-bound) next to the 256-query tile (YAMS_SCAN_FLAG_WIDE_TILE).
One JSON line per (Q, form): whole-step ms, filter-kernel ms (HIP events on the context's stream),
achieved shadow-read GB/s of the filter kernel against the 8 TB/s HBM peak."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE

n, d, k = int(os.environ.get("ROWS", 12_500_000)), int(os.environ.get("DIM", 768)), 100
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr()); acc.synchronize()
# forms: default = both shadows in the view, the library chooses (large shards: the int8 resident-query form);
#        bf16 = only the bf16 shadow (the narrow bf16 form); wide = per-tile kernels (YAMS_SCAN_FLAG_WIDE_TILE)
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
view_bf16 = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr())
view_both = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(),
                            rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr())
keep = {}
for nq in [int(x) for x in os.environ.get("QS", "1,16,64,128,256").split(",")]:
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(42, 1 << 40, nq, d, tq.data_ptr())
    s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c = torch.empty(nq, dtype=torch.int32, device="cuda"); dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for form in os.environ.get("FORMS", "default,bf16,wide").split(","):
        view = view_bf16 if form == "bf16" else view_both
        args = (view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), dist.data_ptr(),
                None, 32 if form == "wide" else 0)   # YAMS_SCAN_FLAG_WIDE_TILE
        for _ in range(2):
            diag = acc.scan_topk_device(*args)
        reps = 6
        acc.enable_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            acc.scan_topk_device(*args, want_diag=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        filt_ms, _ = acc.kernel_ms("scan_filter")
        acc.enable_timing(False)
        res = (r.cpu().numpy().copy(), s.cpu().numpy().view("uint32").copy())
        same = None
        if form != "default" and nq in keep:
            same = bool((keep[nq][0] == res[0]).all() and (keep[nq][1] == res[1]).all())
        elif form == "default":
            keep[nq] = res
        stride = 64
        i8 = diag["filter_tier"] == 1
        filt_bytes = n * d * (1 if i8 else 2) * (stride - 1) / stride
        print(json.dumps({"Q": nq, "form": form, "step_ms": dt * 1e3, "filter_ms": filt_ms,
                          "filter_shadow_GBps": filt_bytes / (filt_ms * 1e-3) / 1e9 if filt_ms else None,
                          "frac_of_8TBps": filt_bytes / (filt_ms * 1e-3) / 8e12 if filt_ms else None,
                          "filter_tier": diag["filter_tier"], "path": diag["path"], "fallbacks": diag["exact_fallback_queries"],
                          "identical_to_default": same}), flush=True)
