"""Host-streamed ingest (yams_ingest_host from pinned memory): end-to-end GB/s against the raw H2D rate, and the kernel
time of every batch (TimedRegion spans) — where a batch's time goes."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yams_amd.accel import Accel, cdc_config
acc = Accel(0, torch.cuda.current_stream().cuda_stream)
n = 8 << 30
host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
stage = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for b0 in range(0, n, 1 << 30):
    acc.synth_bytes(9, b0 >> 22, 256, 4 << 20, stage.data_ptr()); acc.synchronize()
    host[b0:b0 + (1 << 30)].copy_(stage)
del stage
dev = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter()
for off in range(0, n, 2 << 30): dev.copy_(host[off:off + (2 << 30)], non_blocking=True)
torch.cuda.synchronize(); print("raw H2D pinned:", round(n / (time.perf_counter() - t) / 1e9, 1), "GB/s")
del dev
cfg = cdc_config("streaming")
base = host.data_ptr()
for blen, batch in ((256 << 10, 2 << 30), (4 << 20, 2 << 30)):
    nb = n // blen
    ptrs = [base + i * blen for i in range(nb)]
    for flags in (1, 3):
        acc.ingest_host(ptrs, [blen] * nb, cfg, flags=flags, batch_bytes=batch)
        acc.enable_timing(True)
        t = time.perf_counter(); h = acc.ingest_host(ptrs, [blen] * nb, cfg, flags=flags, batch_bytes=batch); dt = time.perf_counter() - t
        spans = {k: [round(x, 2) for x in acc.kernel_ms_all(k)] if hasattr(acc, "kernel_ms_all") else acc.kernel_ms(k) for k in ("cdc_candidates", "cdc_walk", "sha256", "sha256_blobs")}
        acc.enable_timing(False)
        print("blobs of", blen >> 10, "KiB, batches of", batch >> 20, "MiB, flags", flags, ":", round(n / dt / 1e9, 1), "GB/s,", round(dt * 1e3, 1), "ms, chunks", h["n_chunks"], spans)
