"""Worker for tests/test_dist_cpu.py (gloo, CPU) and tests/test_dist_gpu.py (one GPU shared by all
ranks): one rank of a world_size-N job through the same shard / all-gather / merge pipeline
bench.py uses (yams_amd/dist.py GatherPipeline).

With a GPU visible the per-shard search is the HIP path and the merge is the product's
merge_topk_kernel (yams_scan_merge_topk_device) behind the collective.  Without one (the CPU suite)
the per-shard search is the oracle and the merge below is a Python restatement of the comparator —
fallbacks of THIS TEST only; the product has no CPU path."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
from yams_amd import dist as ydist  # noqa: E402

use_gpu = torch.cuda.is_available() and os.environ.get("YAMS_DIST_TEST_CPU") is None
backend = os.environ.get("YAMS_DIST_TEST_BACKEND", "gloo")
if use_gpu and backend == "gloo":
    os.environ["LOCAL_RANK"] = "0"             # every rank on cuda:0 (one-GPU box)
rank, world, local = ydist.init_from_env(backend=backend, force=True)   # (a world of one forms its group too: the one-rank RCCL test)
o = _oracle.oracle()
n, d, nq, k = (40000, 64, 6, 20) if use_gpu else (3000, 16, 4, 10)
n_batches = 3
corpus = o.synth_rows(5, 0, n, d)
corpus[7] = corpus[n - 100]                    # cross-shard exact tie
b = ydist.shard_bounds(n, world)
lo, hi = b[rank], b[rank + 1]
dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
pipe = ydist.GatherPipeline(nq, k, dev, depth=2, force_collective=True)

if use_gpu:
    from yams_amd.accel import Accel
    from yams_amd._lib import SCAN_COSINE
    torch.cuda.set_device(local)
    acc = Accel(local, torch.cuda.current_stream().cuda_stream)
    acc_merge = Accel(local, pipe.side_stream_ptr()) if pipe.active else acc
    tc = torch.from_numpy(corpus[lo:hi]).to(dev)
    torch.cuda.synchronize()                   # the contexts run on their own (non-blocking) streams
    view = acc.corpus_view(tc.data_ptr(), hi - lo, d, row_base=lo)

    def merge_fn(g, out):                      # the product's merge kernel behind the collective
        acc_merge.merge_topk_device(world, nq, k, -1.0, SCAN_COSINE, g["scores"].data_ptr(), g["rows"].data_ptr(),
                                    g["counts"].data_ptr(), None, None, out["scores"].data_ptr(),
                                    out["rows"].data_ptr(), out["counts"].data_ptr(), None)
else:
    def merge_fn(g, out):                      # test-only restatement: (similarity desc, row asc)
        for qi in range(nq):
            ent = [(-float(g["scores"][s, qi, i]), int(g["rows"][s, qi, i]))
                   for s in range(world) for i in range(int(g["counts"][s, qi]))]
            ent.sort()
            ent = ent[:k]
            out["counts"][qi] = len(ent)
            for i, e in enumerate(ent):
                out["scores"][qi, i] = -e[0]; out["rows"][qi, i] = e[1]
pipe.merge_fn = merge_fn

ok = True
bad = []
batches = []


def compare(bi, res, queries):
    global ok
    for qi in range(nq):
        r, s, _, _ = o.scan_cosine(corpus, queries[qi], k, -1.0)
        c = int(res["counts"][qi])
        got_r = res["rows"][qi, :c].cpu().numpy(); got_s = res["scores"][qi, :c].cpu().numpy()
        good = c == len(r) and np.array_equal(got_r, r) and np.array_equal(got_s.view(np.uint32), s.view(np.uint32))
        if not good:
            ok = False
            bad.append({"rank": rank, "batch": bi, "query": qi, "count": c, "want_count": len(r),
                        "got": got_r[:6].tolist(), "want": r[:6].tolist()})

for bi in range(n_batches):                    # several batches in flight: slots are reused
    queries = o.synth_rows(5, (1 << 40) + bi * nq, nq, d)
    if bi == 0:
        queries[0] = corpus[7]
    batches.append(queries)
    slot = bi % pipe.depth
    pipe.wait(slot)                            # the batch that owned this slot has been merged
    loc = pipe.local(slot)
    if use_gpu:
        tq = torch.from_numpy(queries).to(dev)
        torch.cuda.synchronize()
        acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, SCAN_COSINE, loc["scores"].data_ptr(),
                             loc["rows"].data_ptr(), loc["counts"].data_ptr(), want_diag=False)
    else:
        loc["scores"].fill_(-np.inf); loc["rows"].fill_(-1); loc["counts"].zero_()
        for qi in range(nq):                   # "local search" on this rank's shard, global row ids
            r, s, _, _ = o.scan_cosine(corpus[lo:hi], queries[qi], k, -1.0)
            loc["scores"][qi, :len(r)] = torch.from_numpy(s); loc["rows"][qi, :len(r)] = torch.from_numpy(r + lo)
            loc["counts"][qi] = len(r)
    pipe.launch(slot)
    # check this batch right away on even batches, late (after the next launch) on odd ones
    if bi % 2 == 0:
        pipe.wait(slot)
    res = pipe.result(slot)
    if bi % 2 == 0:
        compare(bi, res, queries)
pipe.drain()
for bi in range(n_batches - pipe.depth, n_batches):   # the batches still resident in their slots
    if bi % 2 == 0 or bi < 0:
        continue
    compare(bi, pipe.result(bi % pipe.depth), batches[bi])
t = torch.tensor([1.0 if ok else 0.0], device=dev if backend == "nccl" else "cpu")
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
if bad:
    print("MISMATCH " + json.dumps(bad[:4]), file=sys.stderr, flush=True)
if rank == 0:
    print(json.dumps({"ok": bool(t.item() == 1.0), "world": world, "bounds": b, "gpu": bool(use_gpu),
                      "backend": backend, "merge": "merge_topk_kernel" if use_gpu else "python (test fallback)",
                      "exchange": pipe.exchange_stats(), "batches": n_batches}))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
