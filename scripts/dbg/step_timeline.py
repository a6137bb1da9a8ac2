"""Reads a rocprofv3 kernel trace of the bench's scan leg and prints, for the steady state, what the GPU does between
two filter sweeps: per kernel name the mean duration and how much of it is NOT under a sweep."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void yams_accel::", "").replace("yams_accel::", "")) for r in rows), key=lambda e: e[0])
sweeps = [e for e in ev if e[2].startswith(("scan_tiles_i8d_kernel", "scan_tiles_i8r_kernel<0, false, true, 70"))]
sweeps = sweeps[len(sweeps) // 3:]          # steady state
if len(sweeps) < 4: sys.exit("too few sweeps")
t0, t1 = sweeps[0][0], sweeps[-1][1]
span = (t1 - t0) / 1e6
busy = sum(e - s for s, e, _ in sweeps) / 1e6
print(f"{len(sweeps)} sweeps over {span:.2f} ms: {span / (len(sweeps)):.3f} ms per step (end to end / n), sweep {busy / len(sweeps):.3f} ms each, gaps {(span - busy) / (len(sweeps) - 1):.3f} ms")
inside = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, e, n in ev:
    if s < t0 or e > t1 or n.startswith(("scan_tiles_i8d_kernel", "scan_tiles_i8r_kernel<0, false, true, 70")): continue
    under = 0
    for ss, se, _ in sweeps:
        lo, hi = max(s, ss), min(e, se)
        if hi > lo: under += hi - lo
    v = inside[n]; v[0] += 1; v[1] += (e - s) / 1e3; v[2] += (e - s - under) / 1e3
for n, (c, d, ex) in sorted(inside.items(), key=lambda kv: -kv[1][2]):
    print(f"{n[:64]:64s} calls/step {c / len(sweeps):5.2f}  us/call {d / c:8.1f}  exposed us/step {ex / len(sweeps):8.1f}")
