"""Randomised stress of the ingest path on the GPU against the CPU checkers (oracle C restatement
for boundaries, hashlib for digests): random ragged blob sets at arbitrary byte offsets (random,
constant, short-period and text-like content), random chunker configurations on both sides of the
narrow (32-bit, window 48) candidate kernel's limits, both chunkers.  Boundaries, per-chunk
digests and whole-blob digests must be bit-exact.  Test infrastructure (uses oracle/).

    python tests/stress_ingest.py [--cases 40] [--seed 1]
"""
import argparse, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _oracle
from yams_amd.accel import Accel, cdc_config

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
o = _oracle.oracle()


def content(n):
    kind = rng.integers(0, 5)
    if kind == 0 or n < 8:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.full(n, int(rng.integers(0, 256)), np.uint8)                    # constant: max-size chunks
    if kind == 2:
        p = rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8)        # short period
        return np.tile(p, n // p.size + 1)[:n]
    if kind == 3:
        words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(50)]
        s = b" ".join(words[int(i)] for i in rng.integers(0, 50, n // 4 + 8))
        return np.frombuffer(s[:n].ljust(n, b"."), dtype=np.uint8).copy()
    d = rng.integers(0, 256, n, dtype=np.uint8)                                   # random with repeated segments
    seg = int(rng.integers(1, max(2, n // 3)))
    d[n - seg:] = d[:seg]
    return d


CONFIGS = [dict(), dict(min_size=4096, max_size=65536), dict(min_size=64, max_size=256, mask=0xF),
           dict(min_size=1, max_size=100, mask=3, window=16), dict(min_size=2048, max_size=8192, mask=0xFFFFF),
           dict(min_size=512, max_size=4096, mask=(1 << 40) - 1), dict(min_size=100, max_size=100),
           dict(min_size=300, max_size=200, mask=0xFF), dict(min_size=37, max_size=4001, mask=0x155, window=1),
           dict(min_size=5000, max_size=9000, mask=0x3FF, window=7, polynomial=0xBFE6B8A5BF378D83),
           dict(min_size=8, max_size=5000, mask=0x7FFFFFFF), dict(min_size=8, max_size=5000, mask=0x80000000),
           dict(min_size=40, max_size=90, mask=0), dict(min_size=1000, max_size=1 << 20, mask=0x1FFF, window=47)]
bad, chunks_total, bytes_total = [], 0, 0
for case in range(a.cases):
    n_blobs = int(rng.integers(1, 24))
    lens = [int(x) for x in rng.choice([0, 1, 47, 48, 49, 4095, 4096, 16384, 16385, 70_001, 300_000, 1 << 20, 2_500_000], n_blobs)]
    lens = [int(l * rng.uniform(0.5, 1.0)) if l > 100 and rng.random() < 0.5 else l for l in lens]
    mode = "streaming" if rng.random() < 0.5 else "rabin"
    cfg = CONFIGS[int(rng.integers(0, len(CONFIGS)))]
    if cfg.get("max_size", 1 << 20) <= 256:           # tiny chunks: keep the Python-side digest loop bounded
        lens = [min(l, 60_000) for l in lens]
    blobs = [content(n) for n in lens]
    offs, pos, parts = [], int(rng.integers(0, 17)), []
    parts.append(np.zeros(pos, np.uint8))
    for b in blobs:
        g = int(rng.integers(0, 20))
        offs.append(pos); parts.append(b); parts.append(np.full(g, 0xAB, np.uint8)); pos += len(b) + g
    buf = np.concatenate(parts + [np.zeros(64, np.uint8)])
    tb = torch.from_numpy(buf).cuda()
    res = acc.ingest_device(tb.data_ptr(), offs, lens, cdc_config(mode, **cfg), flags=3)
    out = acc.fetch_ingest(res, n_blobs)
    first = out["blob_first"]
    ok = True
    for bi, b in enumerate(blobs):
        ooff, osz = o.chunks(b, mode, **cfg)
        lo, hi = int(first[bi]), int(first[bi + 1])
        ok &= hi - lo == len(ooff) and np.array_equal(out["chunk_offset"][lo:hi], ooff) and np.array_equal(out["chunk_size"][lo:hi], osz)
        ok &= bool((out["chunk_blob"][lo:hi] == bi).all())
        ok &= out["blob_digest"][bi].tobytes() == hashlib.sha256(b.tobytes()).digest()
        if ok:
            for j in range(lo, hi):
                p, s = int(out["chunk_offset"][j]), int(out["chunk_size"][j])
                ok &= out["chunk_digest"][j].tobytes() == hashlib.sha256(b[p:p + s].tobytes()).digest()
        if not ok:
            bad.append({"case": case, "blob": bi, "len": len(b), "mode": mode, "cfg": {k: int(v) for k, v in cfg.items()}})
            break
    chunks_total += int(out["n_chunks"]); bytes_total += sum(lens)
print(json.dumps({"cases": a.cases, "mismatches": len(bad), "chunks": chunks_total, "bytes": bytes_total, "first_bad": bad[:3]}))
sys.exit(1 if bad else 0)
