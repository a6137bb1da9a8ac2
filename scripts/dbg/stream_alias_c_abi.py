"""Does the sharded handle's step time depend on how many HIP streams the process created BEFORE the handle?
(The bench's c_abi_sharded leg runs after the main loop, the plugin door and the ingest leg have created theirs.)
usage: stream_alias_c_abi.py <n_normal> <n_high> [lanes]  — prints ms_per_step and shard 0's filter launch."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

n_norm, n_high = int(sys.argv[1]), int(sys.argv[2])
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
keep = []
for _ in range(n_norm):
    s = ctypes.c_void_p(); assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0; keep.append(s)
for _ in range(n_high):
    s = ctypes.c_void_p(); assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, hi) == 0; keep.append(s)
sys.argv = ["bench.py", "--gpus", "1", "--via-c-abi", "--lanes", str(lanes), "--steps", "24", "--warmup", "4", "--oracle-queries", "0"]
a = bench.parse()
a.child_json = True
bench.c_abi_main.__globals__["print"] = lambda s_: None
r = bench.c_abi_sharded_run(a, [0], n_query_batches=max(1, a.query_batches))
print(json.dumps({"normal": n_norm, "high": n_high, "lanes": lanes, "ms_per_step": round(r["ms_per_step"], 3),
                  "filter_launch_ms": round(r["filter_launch_ms_shard0"], 3)}))
