"""Ingest leg alone (device-resident Philox blobs): used for profiling and for sweeps."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yams_amd.accel import Accel, cdc_config

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--blob-mib", type=float, default=4.0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--flags", type=int, default=3)
a = ap.parse_args()
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
blen = int(a.blob_mib * (1 << 20))
n_blobs = max(1, int(a.gib * (1 << 30) // blen))
tb = torch.empty(n_blobs * blen, dtype=torch.uint8, device="cuda")
acc.synth_bytes(42, 0, n_blobs, blen, tb.data_ptr())
offs = [i * blen for i in range(n_blobs)]; lens = [blen] * n_blobs
cfg = cdc_config("streaming")
acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=a.flags); acc.synchronize()
acc.enable_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.reps):
    res = acc.ingest_device(tb.data_ptr(), offs, lens, cfg, flags=a.flags)
acc.synchronize(); dt = (time.perf_counter() - t0) / a.reps
out = {"GBps": n_blobs * blen / dt / 1e9, "ms": dt * 1e3, "blobs": n_blobs, "blob_bytes": blen,
       "chunks": int(res.n_chunks), "flags": a.flags}
for k in ("sha256", "sha256_blobs", "cdc_candidates", "cdc_walk"):
    out[k + "_ms"] = acc.kernel_ms(k)[0]
print(json.dumps(out))
