"""Per-wave begin/end ticks of the resident-query filter (measurement build, YAMS_ACCEL_DUMP_SYNC)."""
import os, sys, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/i8sync.bin"
env = dict(os.environ, YAMS_ACCEL_DUMP_SYNC=dump)
out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "filter_ablation.py"), sys.argv[1] if len(sys.argv) > 1 else "i8:2"],
                     env=env, capture_output=True, text=True)
print(out.stdout.strip().splitlines()[-1])
raw = np.fromfile(dump, dtype=np.uint32)
ns = raw.size // (12 * 32)
w = raw[ns * 4 * 32:].reshape(ns, 8, 32)      # [stream][wave][32] debug words behind the [stream][4][32] counters
n_streams = w.shape[0]
beg = w[:, :, 8:16].astype(np.int64); end = w[:, :, 16:24].astype(np.int64); units = w[:, :, 24:32]
t0 = beg[beg > 0].min()
dur = ((end - beg) & 0xffffffff) / 100.0      # us
fin = ((end - t0) & 0xffffffff) / 100.0
print("streams", n_streams, "units per wave min/max", units.min(), units.max())
print("wave duration us: min %.0f mean %.0f max %.0f" % (dur.min(), dur.mean(), dur.max()))
print("finish time us:   min %.0f mean %.0f max %.0f" % (fin.min(), fin.mean(), fin.max()))
# per XCD (stream % 8) and per query tile
for x in range(8):
    f = fin[x::8]
    print("xcd", x, "finish min %.0f mean %.0f max %.0f" % (f.min(), f.mean(), f.max()))
print("per query tile finish mean:", [round(float(fin[:, :, q].mean())) for q in range(8)])
print("per wave finish mean:", [round(float(fin[:, v, :].mean())) for v in range(8)])
