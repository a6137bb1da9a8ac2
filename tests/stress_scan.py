"""Randomised self-consistency stress of the scan paths on the GPU (no CPU oracle in the loop, so it
covers far more shapes per second than the parity tests): for random (rows, dim, batch, k, metric,
threshold, allow-mask) the default path — narrow filter for small batches, 256-query tile above —
must return bit-identical rows / scores / counts to the exhaustive fp64 path
(YAMS_SCAN_FLAG_FORCE_EXACT), which shares no filter code with it, and to the wide form
(YAMS_SCAN_FLAG_WIDE_TILE); where the shape allows it also the int8 tier, in its half-tile form and forced into its
resident-query form (YAMS_SCAN_FLAG_RESIDENT_QUERIES; L2 cases with both shadows take the int8 tier there when the
shard's norms allow it).  Prints one summary line; exit code 1 on any mismatch.

    python tests/stress_scan.py [--cases 60] [--seed 1]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yams_amd.accel import Accel
from yams_amd._lib import SCAN_COSINE, SCAN_L2, FLAG_FORCE_EXACT, FLAG_WIDE_TILE, FLAG_RESIDENT_QUERIES
from yams_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--only", type=int, default=None, help="run the scans of this case only (the random draws of the others still happen)")
ap.add_argument("--first", type=int, default=0, help="debug: the scans of earlier cases are skipped")
ap.add_argument("--no-mask", action="store_true", help="debug: drop the allow-mask of the selected case")
ap.add_argument("--thr", type=float, default=None, help="debug: override the threshold of the selected case")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
bad, done, paths = [], 0, {}
for case in range(a.cases):
    d = int(rng.choice([64, 96, 128, 160, 256, 384, 768, 1024]))
    n = int(rng.integers(4096, 120_000 if d <= 256 else 40_000))
    if d == 256 and rng.random() < 0.12:
        n = int(rng.integers(520_000, 700_000))      # large enough for the sample-driven rules (proof-aware threshold, deeper lists)
    nq = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 257, 300, 520]))
    k = int(rng.choice([1, 5, 10, 50, 100, 200]))
    metric = SCAN_L2 if rng.random() < 0.3 else SCAN_COSINE
    thr = float(rng.choice([-1.0, 0.0, 0.1])) if metric == SCAN_COSINE else -1.0
    # L2: the accumulate arithmetic the host names (fp64 / fp32 sequential / 8 / 16 lanes) rides on every form of the case
    acc_flag = int(rng.choice([0, _lib.FLAG_L2_ACC_F32, _lib.FLAG_L2_ACC_F32X8, _lib.FLAG_L2_ACC_F32X16])) if metric == SCAN_L2 else 0
    if acc_flag and rng.random() < 0.5:
        acc_flag |= _lib.FLAG_L2_ACC_FUSED       # (the same lanes, squares accumulated with a fused multiply-add)
    tc = torch.empty((n, d), dtype=torch.float32, device="cuda"); acc.synth_rows(1000 + case, 0, n, d, tc.data_ptr())
    tq = torch.empty((nq, d), dtype=torch.float32, device="cuda"); acc.synth_rows(1000 + case, 1 << 40, nq, d, tq.data_ptr())
    acc.synchronize()           # (synth_rows ran on the library's stream: finished before torch touches the tensors)
    kind = str(rng.choice(["uniform", "uniform", "gauss", "aniso", "outliers"]))   # (round 6: component distributions the int8 bound is sensitive to)
    if kind != "uniform":
        g_ = torch.Generator(device="cuda"); g_.manual_seed(5000 + case)
        for t_ in (tc, tq):
            t_.copy_(torch.randn(t_.shape, generator=g_, device="cuda"))
            if kind == "aniso":
                t_.mul_(torch.arange(1, d + 1, device="cuda", dtype=torch.float32) ** -0.5)
            elif kind == "outliers":
                t_[:, [3, d // 2, d - 7]] *= 12.0
            t_.mul_(float(rng.uniform(0.2, 5.0)))
    if rng.random() < 0.5:      # clustered: a few hundred (sometimes thousands of) near-copies of query 0 -> crowded top
        m = int(rng.integers(50, 600)) if rng.random() < 0.8 else int(rng.integers(600, min(n // 2, 3000)))
        idx = torch.from_numpy(rng.choice(n, m, replace=False)).cuda()
        tc[idx] = tq[0] * float(rng.uniform(0.5, 2.0)) + 1e-3 * torch.randn((m, d), device="cuda")
    if rng.random() < 0.3:
        tc[int(rng.integers(0, n))] = 0.0
    if rng.random() < 0.3:
        tc[n - 1] *= 1e17
    tb = torch.empty((n, d), dtype=torch.bfloat16, device="cuda"); tn = torch.empty(n, dtype=torch.float32, device="cuda")
    # torch's default stream handle is 0, which the library reads as "no stream given" and answers with a non-blocking stream of
    # its own: torch's writes above and the library's reads below are NOT ordered by a stream — they are ordered here
    torch.cuda.synchronize()
    acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
    mask_t, mask_n = None, 0
    if rng.random() < 0.35:
        keep = rng.random(n) < float(rng.uniform(0.3, 0.95))
        bits = np.zeros((n + 31) // 32 * 32, np.uint8); bits[:n] = keep
        words = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
        mask_t = torch.from_numpy(words.view(np.int32)).cuda(); mask_n = int(keep.sum())
        torch.cuda.synchronize()
    view = acc.corpus_view(tc.data_ptr(), n, d, None, None, 0, mask_t.data_ptr() if mask_t is not None else None, mask_n,
                           rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr())
    if a.only is not None and case == a.only:
        if a.no_mask:
            mask_t, mask_n = None, 0
        if a.thr is not None:
            thr = a.thr
    view = acc.corpus_view(tc.data_ptr(), n, d, None, None, 0, mask_t.data_ptr() if mask_t is not None else None, mask_n,
                           rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr())
    forms = ["default", "wide", "exact"]
    view8 = None
    if metric == SCAN_COSINE and d % 64 == 0 and d >= 256:   # the int8 tier: a view that carries only the int8 shadow
        t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
        i8f = _lib.I8_ROTATED if rng.random() < 0.5 else 0        # (either layout of the int8 shadow)
        acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr(), i8_flags=i8f)
        view8 = acc.corpus_view(tc.data_ptr(), n, d, None, None, 0, mask_t.data_ptr() if mask_t is not None else None, mask_n,
                                rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr(), i8_flags=i8f)
        forms.insert(0, "i8#2"); forms.insert(0, "i8")            # (#2: the same call again — what the context learnt about the corpus applies)
        if d % 128 == 0 and d <= 768:     # the resident-query kernel form of the int8 filter, forced on these small shards
            forms.insert(0, "i8r")
    if metric == SCAN_L2 and d % 64 == 0 and d >= 256:   # L2 on the int8 tier: both shadows, half tiles and the resident-query form forced; steps aside for zero / huge rows and wide norm ranges
        t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device="cuda"); tm8 = torch.empty(((n + 15) // 16, 2), dtype=torch.float32, device="cuda")
        i8f = _lib.I8_ROTATED if rng.random() < 0.5 else 0
        acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr(), i8_flags=i8f)
        view8 = acc.corpus_view(tc.data_ptr(), n, d, None, None, 0, mask_t.data_ptr() if mask_t is not None else None, mask_n,
                                rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(), rows_i8_meta_ptr=tm8.data_ptr(), i8_flags=i8f)
        forms.insert(0, "i8#2"); forms.insert(0, "i8")
        if d % 128 == 0 and d <= 768:
            forms.insert(0, "i8r")
    if os.environ.get("STRESS_FORMS"):      # debug: only these forms (comma-separated), in this order
        forms = [f for f in os.environ["STRESS_FORMS"].split(",") if f in forms]
    if os.environ.get("STRESS_VERBOSE"):
        sys.stderr.write(f"case {case}: n={n} d={d} nq={nq} k={k} metric={int(metric)} thr={thr} kind={kind} mask={mask_n} forms={forms}\n"); sys.stderr.flush()
    out = {}
    if (a.only is not None and case != a.only and case < a.first) or (a.only is None and case < a.first):
        torch.cuda.synchronize()
        del tc, tq, tb, tn
        continue
    for form in forms:
        s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); r = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        c = torch.empty(nq, dtype=torch.int32, device="cuda"); dist = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        diag = acc.scan_topk_device(view8 if form in ("i8", "i8#2", "i8r") else view, tq.data_ptr(), nq, k, thr, metric, s.data_ptr(), r.data_ptr(), c.data_ptr(),
                                    dist.data_ptr(), flags=acc_flag | (FLAG_FORCE_EXACT if form == "exact" else
                                    (FLAG_WIDE_TILE if form == "wide" else (FLAG_RESIDENT_QUERIES if form == "i8r" else 0))))
        torch.cuda.synchronize()
        if os.environ.get("STRESS_VERBOSE"):
            sys.stderr.write(f"  form {form}: tier {diag['filter_tier']} widened {diag['widened_queries']} fallback {diag['exact_fallback_queries']}\n"); sys.stderr.flush()
        cn = c.cpu().numpy()
        sel = np.arange(k)[None, :] < cn[:, None]        # only the returned prefix is defined
        out[form] = (cn, np.where(sel, r.cpu().numpy(), -1), np.where(sel, s.cpu().numpy().view(np.uint32), 0),
                     np.where(sel, dist.cpu().numpy().view(np.uint32), 0) if metric == SCAN_L2 else None, diag)
    if "exact" not in out:
        done += 1
        del tc, tq, tb, tn
        continue
    ref = out["exact"]
    for form in [f for f in forms if f != "exact"]:
        o = out[form]
        ok = (o[0] == ref[0]).all() and (o[1] == ref[1]).all() and (o[2] == ref[2]).all()
        if metric == SCAN_L2:
            ok = ok and (o[3] == ref[3]).all()
        if not ok:
            dq = [int(x) for x in np.flatnonzero((o[0] != ref[0]) | (o[1] != ref[1]).any(axis=1) | (o[2] != ref[2]).any(axis=1))[:2]]
            det = []
            for qq in dq:
                ii = np.flatnonzero((o[1][qq] != ref[1][qq]) | (o[2][qq] != ref[2][qq]))
                i0 = int(ii[0]) if ii.size else -1
                det.append({"q": qq, "count": [int(o[0][qq]), int(ref[0][qq])], "at": i0,
                            "row": [int(o[1][qq][i0]), int(ref[1][qq][i0])] if i0 >= 0 else None,
                            "sim_bits": [hex(int(o[2][qq][i0])), hex(int(ref[2][qq][i0]))] if i0 >= 0 else None})
            bad.append({"detail": det, "case": case, "form": form, "n": n, "d": d, "nq": nq, "k": k, "metric": int(metric), "thr": thr, "kind": kind,
                        "mask": mask_n, "l2_acc": acc_flag, "diag": {kk: int(v) for kk, v in o[4].items()}})
    dg = out["default"][4]
    key = f"path{dg['path']}/widened{int(dg['widened_queries'] > 0)}/escalated{int(dg['escalated_queries'] > 0)}/fallback{int(dg['exact_fallback_queries'] > 0)}"
    paths[key] = paths.get(key, 0) + 1
    for form in ("i8", "i8r"):
        if metric == SCAN_L2 and form in out:
            key = f"l2_{form}_tier{int(out[form][4]['filter_tier'])}"
            paths[key] = paths.get(key, 0) + 1
    done += 1
    del tc, tq, tb, tn
print(json.dumps({"cases": done, "mismatches": len(bad), "paths": paths, "first_bad": bad[:3]}))
sys.exit(1 if bad else 0)
