"""Soak of the bench's scan configuration (12.5M x 768 shard, 1024 queries, top-100, two search lanes sharing a sweep gate):
N batches rotating over a few distinct query batches; every result must equal, bit for bit, the first result of its query
batch.  Looks for what parity tests of a handful of launches cannot see: rare races between the lanes, in the strip
counters / pacing of the persistent sweep, in the candidate lists.  Prints one JSON line; exit code 1 on any difference.

    python tests/soak_scan.py [--batches 1200] [--rows 12500000] [--lanes 2]
"""
import argparse, hashlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yams_amd.accel import Accel, SweepGate
from yams_amd._lib import SCAN_COSINE

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=1200)
ap.add_argument("--rows", type=int, default=12_500_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--distinct", type=int, default=4)
a = ap.parse_args()
n, d, nq, k = a.rows, a.dim, a.queries, a.k
dev = torch.device("cuda:0")
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
tc = torch.empty((n, d), dtype=torch.float32, device=dev); acc.synth_rows(42, 0, n, d, tc.data_ptr())
tb = torch.empty((n, d), dtype=torch.bfloat16, device=dev); tn = torch.empty(n, dtype=torch.float32, device=dev)
acc.build_shadow_device(tc.data_ptr(), n, d, tb.data_ptr(), tn.data_ptr())
t8 = torch.empty(((n + 63) // 64 * 64, d), dtype=torch.int8, device=dev); tm8 = torch.empty(((n + 63) // 64, 2), dtype=torch.float32, device=dev)
acc.build_shadow_i8_device(tc.data_ptr(), n, d, t8.data_ptr(), tm8.data_ptr())
view = acc.corpus_view(tc.data_ptr(), n, d, rows_bf16_ptr=tb.data_ptr(), rows_nsq_ptr=tn.data_ptr(), rows_i8_ptr=t8.data_ptr(),
                       rows_i8_meta_ptr=tm8.data_ptr())
tqs = []
for b in range(a.distinct):
    t = torch.empty((nq, d), dtype=torch.float32, device=dev); acc.synth_rows(42, (1 << 40) + b * nq, nq, d, t.data_ptr()); tqs.append(t)
acc.synchronize()
streams = [None] + [torch.cuda.Stream(device=dev) for _ in range(a.lanes - 1)]
accs = [acc] + [Accel(0, s.cuda_stream) for s in streams[1:]]
gate = SweepGate(0) if a.lanes > 1 else None
for c in accs:
    if gate is not None: c.set_gate(gate)
outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev),
         torch.empty(nq, dtype=torch.int32, device=dev)) for _ in accs]
first, bad, lock = {}, [], threading.Lock()
fallbacks = [0]

def lane_fn(lane):
    torch.cuda.set_device(dev)
    s, r, c = outs[lane]
    for i in range(lane, a.batches, a.lanes):
        b = i % a.distinct
        dg = accs[lane].scan_topk_device(view, tqs[b].data_ptr(), nq, k, -1.0, SCAN_COSINE, s.data_ptr(), r.data_ptr(), c.data_ptr(),
                                         want_diag=(i % 50 == 0))
        h = hashlib.sha256(r.cpu().numpy().tobytes() + s.cpu().numpy().tobytes() + c.cpu().numpy().tobytes()).hexdigest()
        with lock:
            if dg and dg.get("exact_fallback_queries"): fallbacks[0] += int(dg["exact_fallback_queries"])
            if b not in first: first[b] = h
            elif first[b] != h: bad.append({"batch": i, "lane": lane, "query_batch": b})

t0 = time.perf_counter()
th = [threading.Thread(target=lane_fn, args=(l,)) for l in range(a.lanes)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
print(json.dumps({"batches": a.batches, "lanes": a.lanes, "rows": n, "dim": d, "queries": nq, "distinct_query_batches": a.distinct,
                  "seconds": round(dt, 1), "ms_per_batch_incl_digest": round(dt / a.batches * 1e3, 2), "differences": len(bad),
                  "first_bad": bad[:3], "exact_fallback_queries_seen": fallbacks[0]}))
sys.exit(1 if bad else 0)
