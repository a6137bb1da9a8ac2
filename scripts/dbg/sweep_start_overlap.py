"""Which kernels are still running when a filter sweep starts, and for how long (kernel trace of the bench's scan leg):
a persistent sweep can only take a CU whole, so whatever sits on CUs at its start delays part of its grid."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void yams_accel::", "").replace("yams_accel::", "")) for r in rows), key=lambda e: e[0])
sweeps = [e for e in ev if e[2].startswith(("scan_tiles_i8d_kernel", "scan_tiles_i8r_kernel<0, false, true, 70"))]
sweeps = sweeps[len(sweeps) // 3:]
agg = collections.defaultdict(lambda: [0, 0.0])
durs = []
for i, (ss, se, _) in enumerate(sweeps):
    durs.append((se - ss) / 1e3)
    tail = 0.0
    for s, e, n in ev:
        if n.startswith(("scan_tiles_i8d_kernel", "scan_tiles_i8r_kernel<0, false, true, 70")): continue
        if s < ss + 20_000 and e > ss:           # running at (or within 20 us after) the sweep's start
            agg[n][0] += 1; agg[n][1] += (e - max(s, ss)) / 1e3
            tail = max(tail, (e - ss) / 1e3)
    if i < 12:
        prev_end = sweeps[i - 1][1] if i else ss
        print(f"sweep {i}: gap before {(ss - prev_end) / 1e3:7.1f} us, duration {(se - ss) / 1e3:7.1f} us, other kernels still running for {tail:7.1f} us after its start")
print("mean sweep duration", sum(durs) / len(durs))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:60]:60s} at {c / len(sweeps):4.2f} of the sweep starts, mean overlap {t / c:7.1f} us")
