"""Why is the host-streamed 32 GiB call slow inside bench.py's process?  Runs bench.ingest_leg (parts switched off by
SKIP=...) and then the same host-streamed call as scripts/dbg/host_stream_batches.py."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from yams_amd.accel import Accel, cdc_config
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
skip = os.environ.get("SKIP", "").split(",")
if "verify" in skip: bench.verify_ingest_sample = lambda *a, **k: {"skipped": True}
if "cpu" in skip: bench.ingest_cpu_baseline = lambda *a, **k: {"skipped": True}
if "breadth" in skip or True: bench.ingest_breadth = lambda *a, **k: {"skipped": True}
if "dedup" in skip: acc.dedup_set = lambda n: (_ for _ in ()).throw(RuntimeError("skipped"))
if "leg" not in skip:
    r = bench.ingest_leg(acc, torch, float(os.environ.get("GIB", "100")), 42)
    print("ingest_leg:", r["value"], file=sys.stderr)
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
acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=0)
t0 = time.perf_counter(); acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=0); dt = time.perf_counter() - t0
print(json.dumps({"skip": skip, "GBps": round(len(ptrs) * blen / dt / 1e9, 2)}))
# ... and the order bench.py's breadth leg uses: the 8 GiB call in 2 GiB batches first, then the 32 GiB call
p8 = ptrs[:n_blobs]
acc.ingest_host(p8, [blen] * n_blobs, cfg, flags=3, batch_bytes=2 << 30)
t0 = time.perf_counter(); acc.ingest_host(p8, [blen] * n_blobs, cfg, flags=3, batch_bytes=2 << 30); dt8 = time.perf_counter() - t0
acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=0)
t0 = time.perf_counter(); acc.ingest_host(ptrs, lens, cfg, flags=3, batch_bytes=0); dt = time.perf_counter() - t0
print(json.dumps({"after_2GiB_batches": True, "GBps_8GiB_call": round(n_blobs * blen / dt8 / 1e9, 2), "GBps_32GiB_call": round(len(ptrs) * blen / dt / 1e9, 2)}))
