"""Run as a subprocess by test_scan_gpu.py with YAMS_ACCEL_MEASURE_LIB=1: the measurement build's alternative forms of
the resident-query int8 filter — 70: 128 x 128 wave tiles, one wave per SIMD, row fragments loaded straight into
registers, block entries in the survivor log; 80: 64 x 128 wave tiles with direct row loads, the plain strip boundary; 87: the
same with the short strip boundary as shipped (thresholds and survivors in LDS, the boundary's work in front of the drain); 90:
two slabs of row fragments in flight per wave (the product's form at dims 384 / 768) — against the LDS-ring form (2; the product's form elsewhere) on the same shard: identical results AND identical candidate sets (count),
on ragged shards (a last strip of 64 rows, a last unit of one tile), with thresholds and an allow-mask.  Prints one
JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE, FLAG_RESIDENT_QUERIES

acc = Accel(0, torch.cuda.current_stream().cuda_stream)
out = []
SHAPES = json.loads(os.environ["FORMS_SHAPES"]) if os.environ.get("FORMS_SHAPES") else None
VERSIONS = os.environ.get("FORMS_VERSIONS", "2,70,80,87,90").split(",")   # the first one is the yardstick (the LDS-ring form)
for (n, d, nq, k, thr, masked) in SHAPES or [(300_001, 768, 1024, 100, -1.0, False), (150_080, 384, 300, 50, 0.02, True),
                                   (90_000, 768, 130, 100, -1.0, True), (70_001, 384, 1024, 10, -1.0, False),
                                   (120_000, 512, 260, 20, -1.0, False)]:
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(77, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(77, 1 << 40, nq, d, tq.data_ptr())
    t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda")
    tm8 = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device="cuda")
    acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr()); acc.synchronize()
    mask = None
    if masked:
        import numpy as np
        allow = np.random.default_rng(n).random((n + 31) // 32 * 32) < 0.5
        allow[n:] = False
        mask = torch.from_numpy(np.packbits(allow, bitorder="little").view(np.uint32).view(np.int32).copy()).cuda()
    view = acc.corpus_view(tc.data_ptr(), n, d, rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr(),
                           row_mask_ptr=mask.data_ptr() if mask is not None else None, row_mask_count=int(allow.sum()) if mask is not None else 0)
    res = {}
    for v in VERSIONS:
        os.environ["YAMS_ACCEL_BF16_KERNEL"] = v
        os.environ["YAMS_ACCEL_I8R_DIRECT"] = "0" if v == "2" else "1"   # "2": the LDS-ring form whatever the launcher's rule says
        s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        c = torch.zeros(nq, dtype=torch.int32, device="cuda")
        dg = acc.scan_topk_device(view, tq.data_ptr(), nq, k, thr, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(),
                                  flags=FLAG_RESIDENT_QUERIES, want_diag=True)
        acc.synchronize()
        if os.environ.get("FORMS_DUMP"):
            import numpy as np
            os.environ["YAMS_ACCEL_DUMP_LCOUNT"] = "/tmp/lc.bin"
            acc.scan_topk_device(view, tq.data_ptr(), nq, k, thr, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(), flags=FLAG_RESIDENT_QUERIES, want_diag=False)
            acc.synchronize(); del os.environ["YAMS_ACCEL_DUMP_LCOUNT"]
            dg["lcount"] = np.fromfile("/tmp/lc.bin", dtype=np.uint32).tolist()
        res[v] = (r.cpu(), s.cpu(), c.cpu(), dg)
    rec = {"shape": [n, d, nq, k, thr, masked], "tier": res["2"][3].get("filter_tier"), "candidates": {v: res[v][3].get("filter_candidates") for v in res},
           "fallback": {v: res[v][3].get("exact_fallback_queries") for v in res}}
    if os.environ.get("FORMS_DUMP") and "70" in res:
        a2, a7 = res["2"][3]["lcount"], res["70"][3]["lcount"]
        rec["lcount_diff"] = [(i, x, y) for i, (x, y) in enumerate(zip(a2, a7)) if x != y][:12]
    for v in VERSIONS[1:]:
        rec["identical_" + v] = bool(torch.equal(res[v][0], res["2"][0]) and torch.equal(res[v][1], res["2"][1]) and torch.equal(res[v][2], res["2"][2]))
    out.append(rec)
print(json.dumps(out))
