"""Host-streamed ingest of 4 MiB blobs (yams_ingest_host, pinned source, 32 GiB per call): GB/s by batch size.
A batch's whole-blob chains take ~120 ms whatever its size; up to three batches' chains are in flight."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from yams_amd.accel import Accel, cdc_config
acc = Accel(0)
blen, n_blobs = 4 << 20, 2048
host = torch.empty(n_blobs * blen, dtype=torch.uint8, pin_memory=True)
stage = torch.empty(256 * blen, dtype=torch.uint8, device="cuda")
for b0 in range(0, n_blobs, 256):
    acc.synth_bytes(7, b0, 256, blen, stage.data_ptr()); acc.synchronize()
    host[b0 * blen:(b0 + 256) * blen].copy_(stage)
del stage; torch.cuda.empty_cache()
base = host.data_ptr()
ptrs = [base + i * blen for i in range(n_blobs)] * 4
lens = [blen] * len(ptrs)
cfg = cdc_config("streaming")
out = {}
ref = None
for gib in (1, 2, 4, 8):
    acc.ingest_host(ptrs[:4096], lens[:4096], cfg, flags=3, batch_bytes=gib << 30)      # warm-up: buffers of this size
    t0 = time.perf_counter()
    h = acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=gib << 30)
    dt = time.perf_counter() - t0
    if ref is None: ref = h
    same = bool(np.array_equal(h["blob_digest"], ref["blob_digest"]) and np.array_equal(h["chunk_digest"][:h["n_chunks"]], ref["chunk_digest"][:ref["n_chunks"]]))
    out[f"{gib}GiB"] = {"GBps": round(len(ptrs) * blen / dt / 1e9, 2), "ms": round(dt * 1e3, 1), "same_as_first": same}
    print(json.dumps(out), flush=True)
