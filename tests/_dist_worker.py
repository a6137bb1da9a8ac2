"""Worker for tests/test_dist_cpu.py: one rank of a world_size-2 gloo job exercising the same
shard / all-gather / merge plumbing bench.py uses on GPUs (yams_amd/dist.py).  The per-shard
search is injected: here it is the CPU oracle (test infrastructure), on GPUs it is the HIP path."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
from yams_amd import dist as ydist  # noqa: E402

rank, world, _ = ydist.init_from_env(backend="gloo")
o = _oracle.oracle()
n, d, nq, k = 3000, 16, 4, 10
corpus = o.synth_rows(5, 0, n, d)
corpus[7] = corpus[2900]                      # cross-shard exact tie
queries = o.synth_rows(5, 1 << 40, nq, d)
queries[0] = corpus[7]
b = ydist.shard_bounds(n, world)
lo, hi = b[rank], b[rank + 1]
scores = torch.full((nq, k), -np.inf, dtype=torch.float32)
rows = torch.full((nq, k), -1, dtype=torch.int64)
counts = torch.zeros(nq, dtype=torch.int32)
for qi in range(nq):                          # "local search" on this rank's shard, global row ids
    r, s, _, _ = o.scan_cosine(corpus[lo:hi], queries[qi], k, -1.0)
    scores[qi, :len(r)] = torch.from_numpy(s); rows[qi, :len(r)] = torch.from_numpy(r + lo); counts[qi] = len(r)


def merge_fn(g, w):                           # reference merge: (similarity desc, row asc)
    out = []
    for qi in range(nq):
        ent = [(-float(g["scores"][s, qi, i]), int(g["rows"][s, qi, i]))
               for s in range(w) for i in range(int(g["counts"][s, qi]))]
        ent.sort()
        out.append([e[1] for e in ent[:k]])
    return out


merged = ydist.gather_and_merge({"scores": scores, "rows": rows, "counts": counts}, k, merge_fn)
ok = True
for qi in range(nq):
    r, _, _, _ = o.scan_cosine(corpus, queries[qi], k, -1.0)
    ok &= merged[qi] == [int(x) for x in r]
t = torch.tensor([1.0 if ok else 0.0])
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
if rank == 0:
    print(json.dumps({"ok": bool(t.item() == 1.0), "world": world, "bounds": b}))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
