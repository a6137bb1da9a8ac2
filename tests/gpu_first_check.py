"""First-contact GPU check: loader, synthetic data, SHA-256, CDC, scan (exact + MFMA) vs the oracle."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from yams_amd.accel import Accel, cdc_config
from yams_amd import _lib
import _oracle

o = _oracle.oracle()
print("torch", torch.__version__, "cuda", torch.cuda.is_available(), flush=True)
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
print(acc.device_info(), flush=True)

def step(name):
    print(f"=== {name}", flush=True)

step("synth rows")
n, d = 1000, 384
t = torch.empty((n, d), dtype=torch.float32, device="cuda")
acc.synth_rows(42, 5, n, d, t.data_ptr()); torch.cuda.synchronize()
ref = o.synth_rows(42, 5, n, d)
got = t.cpu().numpy()
print("synth rows bit-exact:", np.array_equal(got.view(np.uint32), ref.view(np.uint32)), np.abs(got - ref).max())

step("synth bytes")
tb = torch.empty(3 * 1000, dtype=torch.uint8, device="cuda")
acc.synth_bytes(7, 10, 3, 1000, tb.data_ptr()); torch.cuda.synchronize()
gb = tb.cpu().numpy()
rb = np.concatenate([o.synth_bytes(7, 10 + b, 0, 1000) for b in range(3)])
print("synth bytes exact:", np.array_equal(gb, rb))

step("sha256")
for m in [b"", b"abc", b"Hello World"]:
    print(acc.sha256_hex(m), acc.sha256_hex(m) == hashlib.sha256(m).hexdigest())
rng = np.random.default_rng(1)
msgs = [rng.integers(0, 256, n, dtype=np.uint8) for n in [0, 1, 17, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 4096, 65537, 1000003]]
hx = acc.sha256_many(msgs)
ok = all(h == hashlib.sha256(m.tobytes()).hexdigest() for h, m in zip(hx, msgs))
print("sha256 many:", ok)
if not ok:
    for h, m in zip(hx, msgs): print(len(m), h == hashlib.sha256(m.tobytes()).hexdigest())
# many random small messages with odd alignment
msgs = [rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8) for _ in range(3000)]
hx = acc.sha256_many(msgs)
print("sha256 3000 ragged:", all(h == hashlib.sha256(m.tobytes()).hexdigest() for h, m in zip(hx, msgs)))

step("cdc")
data = rng.integers(0, 256, 8 << 20, dtype=np.uint8)
for mode in ["streaming", "rabin"]:
    for kw in [dict(), dict(min_size=4096, max_size=65536), dict(min_size=64, max_size=256, mask=0xF),
               dict(min_size=1, max_size=100, mask=0x3, window=16), dict(min_size=2048, max_size=8192, mask=0xFFFFF),
               dict(min_size=512, max_size=4096, mask=0xFFFFFFFFFF)]:
        dd = data if not kw or kw.get("min_size", 0) >= 2048 else data[:1 << 20]
        cfg = cdc_config(mode, **kw)
        t0 = time.time(); off, sz, hx = acc.chunk(dd, cfg, with_hashes=True); t1 = time.time() - t0
        ooff, osz = o.chunks(dd, mode, **kw)
        same = len(off) == len(ooff) and np.array_equal(off, ooff) and np.array_equal(sz, osz)
        hok = all(hx[i] == hashlib.sha256(dd[int(off[i]):int(off[i] + sz[i])].tobytes()).hexdigest() for i in range(0, len(off), max(1, len(off) // 50)))
        print(mode, kw, "chunks", len(off), len(ooff), "boundaries", same, "hashes", hok, "%.3fs" % t1, flush=True)

step("scan exact small")
c = np.array([[1, i, 0, 0] for i in range(6)], np.float32)
dc = acc.to_device(c)
r = acc.scan_topk(acc.corpus_view(dc.ptr, 6, 4), np.array([1, 0, 0, 0], np.float32), 3, -1.0)
print(r.rows, r.scores, r.counts, r.diag)
print("oracle", o.scan_cosine(c, np.array([1, 0, 0, 0], np.float32), 3))

def check_scan(n, d, nq, k, metric=0, thr=-1.0, seed=3, flags=0):
    corpus = o.synth_rows(seed, 0, n, d)
    queries = o.synth_rows(seed, 10_000_000, nq, d)
    dcorp = acc.to_device(corpus)
    t0 = time.time()
    r = acc.scan_topk(acc.corpus_view(dcorp.ptr, n, d), queries, k, thr, metric, flags)
    t1 = time.time() - t0
    bad = 0; maxd = 0.0
    for qi in range(min(nq, 16)):
        if metric == 0:
            rows, sims, _, _ = o.scan_cosine(corpus, queries[qi], k, thr)
        else:
            rows, dist, sims = o.scan_l2(corpus, queries[qi], k, thr)
        cnt = int(r.counts[qi])
        if cnt != len(rows) or not np.array_equal(r.rows[qi, :cnt], rows):
            bad += 1
            if bad <= 2:
                print("  MISMATCH q", qi, cnt, len(rows), r.rows[qi, :8], rows[:8], r.scores[qi, :4], sims[:4])
        else:
            maxd = max(maxd, float(np.abs(r.scores[qi, :cnt].view(np.uint32).astype(np.int64) - sims.view(np.uint32).astype(np.int64)).max()) if cnt else 0)
            if metric == 1:
                maxd = max(maxd, float(np.abs(r.dist[qi, :cnt].view(np.uint32).astype(np.int64) - dist.view(np.uint32).astype(np.int64)).max()))
    print(f"scan n={n} d={d} nq={nq} k={k} metric={metric} thr={thr} flags={flags}: mismatches={bad} max_ulp={maxd} {t1:.3f}s diag={r.diag}", flush=True)
    dcorp.free()

step("scan paths")
check_scan(3000, 64, 5, 10)                 # exact path
check_scan(3000, 384, 3, 100, metric=1)     # exact path L2
check_scan(20000, 64, 5, 10)                # mfma path, all-sample regime
check_scan(20000, 384, 37, 100)
check_scan(200000, 384, 130, 100)           # sample + filter
check_scan(200000, 384, 130, 100, thr=0.15)
check_scan(200000, 768, 16, 10, metric=1)
check_scan(200000, 384, 16, 100, metric=1, thr=0.1)
check_scan(50000, 128, 8, 1000)
check_scan(200000, 384, 16, 100, flags=2)   # forced exact

step("scan timing 1M x 384, Q=256, k=100")
n, d, nq, k = 1_000_000, 384, 256, 100
tc = torch.empty((n, d), dtype=torch.float32, device="cuda")
acc.synth_rows(42, 0, n, d, tc.data_ptr())
tq = torch.empty((nq, d), dtype=torch.float32, device="cuda")
acc.synth_rows(42, n, nq, d, tq.data_ptr())
os_ = torch.empty((nq, k), dtype=torch.float32, device="cuda"); orow = torch.empty((nq, k), dtype=torch.int64, device="cuda")
oc = torch.empty(nq, dtype=torch.int32, device="cuda")
view = acc.corpus_view(tc.data_ptr(), n, d)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    diag = acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, 0, os_.data_ptr(), orow.data_ptr(), oc.data_ptr())
    torch.cuda.synchronize(); t1 = time.time() - t0
    print(f"iter {it}: {t1*1e3:.2f} ms -> {nq/t1:.0f} QPS, TF={2*n*d*nq/t1/1e12:.1f}", diag, flush=True)
acc.enable_timing(True)
for it in range(5):
    acc.scan_topk_device(view, tq.data_ptr(), nq, k, -1.0, 0, os_.data_ptr(), orow.data_ptr(), oc.data_ptr(), want_diag=False)
for nm in ["scan_sample", "scan_filter"]:
    print(nm, acc.kernel_ms(nm))
# verify a few queries against the oracle (regenerate the corpus on the CPU)
corpus = o.synth_rows(42, 0, n, d); queries = o.synth_rows(42, n, nq, d)
print("device corpus == oracle corpus:", np.array_equal(tc.cpu().numpy().view(np.uint32), corpus.view(np.uint32)))
rows_g = orow.cpu().numpy(); sc_g = os_.cpu().numpy()
bad = 0
for qi in [0, 1, 100, 255]:
    rows, sims, _, _ = o.scan_cosine(corpus, queries[qi], k, -1.0)
    if not np.array_equal(rows_g[qi], rows) or not np.array_equal(sc_g[qi].view(np.uint32), sims.view(np.uint32)): bad += 1
print("1M parity mismatches (4 queries):", bad)
print("DONE")
