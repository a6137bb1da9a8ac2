"""Where does a corpus upload spend its time?  (38.4 GB in 6.2 s through the plugin.)
Prints one JSON line: yams_accel_upload of a pageable / pinned 4 GiB buffer, vector_scan_v1.corpus_append of 4 GiB of
rows with and without shadows, in one call and in 512 MiB calls."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from yams_amd import _lib
from yams_amd.accel import Accel

L = _lib.load()
acc = Accel(0)
n, d = 1_400_000, 768                       # 4.3 GB
rows = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)
out = {"bytes": rows.nbytes}
buf = acc.alloc(rows.nbytes)
for name, src in (("pageable", rows), ("pinned", torch.from_numpy(rows).pin_memory().numpy())):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        acc._check(L.yams_accel_upload(acc.ctx, buf.ptr, src.ctypes.data, src.nbytes))
        ts.append(time.perf_counter() - t0)
    out["upload_" + name + "_GBps"] = [round(rows.nbytes / t / 1e9, 2) for t in ts]
buf.free()
for cfg_name, cfg in (("no_shadows", b'{"device": 0, "shadows": "none"}'), ("both_shadows", b'{"device": 0}')):
    L.yams_plugin_shutdown()
    assert L.yams_plugin_init(cfg, None) == 0
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.VectorScanV1)).contents
    for pieces in (1, 8):
        ts = []
        for rep in range(2):
            cid = C.c_uint64()
            assert vt.corpus_create(None, d, C.byref(cid)) == 0
            t0 = time.perf_counter()
            step = n // pieces
            for i in range(pieces):
                part = rows[i * step:(i + 1) * step]
                assert vt.corpus_append(None, cid, part.ctypes.data_as(_lib.f32p), part.shape[0]) == 0
            ts.append(time.perf_counter() - t0)
            hp = C.c_void_p()
            assert L.yams_plugin_get_health_json(C.byref(hp)) == 0
            out.setdefault(f"last_append_{cfg_name}_{pieces}_calls", []).append(json.loads(C.string_at(hp))["last_append"])
            C.CDLL(None).free(hp)
            assert vt.corpus_destroy(None, cid) == 0
        out[f"append_{cfg_name}_{pieces}_calls_GBps"] = [round(step * pieces * d * 4 / t / 1e9, 2) for t in ts]
    L.yams_plugin_shutdown()
print(json.dumps(out))
