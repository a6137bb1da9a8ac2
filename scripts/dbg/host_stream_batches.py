"""Host-streamed ingest of 4 MiB blobs (yams_ingest_host, pinned source, 32 GiB per call): GB/s by batch size.
A batch's whole-blob chains take ~120 ms whatever its size; up to three batches' chains are in flight."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from yams_amd.accel import Accel, cdc_config
import os
acc = Accel(0, torch.cuda.current_stream().cuda_stream) if os.environ.get("TORCH_STREAM") else Accel(0)
blen, n_blobs = 4 << 20, 2048
if os.environ.get("PRE_GIB"):      # what bench.py does before this leg: a device-resident ingest of that many GiB, then its tensor is freed
    g = int(os.environ["PRE_GIB"]); nb = g * (1 << 30) // blen
    tb = torch.empty(nb * blen, dtype=torch.uint8, device="cuda")
    acc.synth_bytes(1, 0, nb, blen, tb.data_ptr())
    if os.environ.get("PRE_INGEST", "1") == "1":
        acc.ingest_device(tb.data_ptr(), [i * blen for i in range(nb)], [blen] * nb, cdc_config("streaming"), flags=3); acc.synchronize()
    del tb; torch.cuda.empty_cache()
host = torch.empty(n_blobs * blen, dtype=torch.uint8, pin_memory=not os.environ.get("PAGEABLE"))   # PAGEABLE=1: what a host that read files into ordinary buffers passes
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
for gib in [int(x) for x in os.environ.get("GIBS", "1 2 4 8").split()]:
    acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=gib << 30)      # warm-up: buffers of this size (gib = 0: the library's choice)
    t0 = time.perf_counter()
    h = acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=gib << 30)
    dt = time.perf_counter() - t0
    if ref is None: ref = h
    same = bool(np.array_equal(h["blob_digest"], ref["blob_digest"]) and np.array_equal(h["chunk_digest"][:h["n_chunks"]], ref["chunk_digest"][:ref["n_chunks"]]))
    out[f"{gib}GiB"] = {"GBps": round(len(ptrs) * blen / dt / 1e9, 2), "ms": round(dt * 1e3, 1), "same_as_first": same}
    print(json.dumps(out), flush=True)
