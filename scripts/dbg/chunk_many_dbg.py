import ctypes as C, hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from yams_amd import _lib
import _oracle
o = _oracle.oracle()
L = _lib.load()
assert L.yams_plugin_init(b"{}", None) == 0
p = C.c_void_p(); L.yams_plugin_get_interface(b"chunker_v1", 2, C.byref(p))
vt = C.cast(p, C.POINTER(_lib.ChunkerV1)).contents
cfg = _lib.CdcConfig(); vt.get_default_config(None, _lib.CDC_STREAMING, C.byref(cfg)); cfg.min_size, cfg.max_size = 2048, 65536
data = np.random.default_rng(1).integers(0, 256, (3 << 20) + 12345, dtype=np.uint8)
sizes = [100000, 0, 47, 2048, 2049, data.size - 1000, 65537, 1]
offs = [n % 977 for n in sizes]
ptrs = (C.c_void_p * len(sizes))(*[data.ctypes.data + o_ for o_ in offs])
lens = (C.c_size_t * len(sizes))(*sizes)
batch = C.POINTER(_lib.ChunkBatch)()
print("st", vt.chunk_many(None, ptrs, lens, len(sizes), C.byref(cfg), 1, C.byref(batch)))
bt = batch.contents
for b, (o_, n) in enumerate(zip(offs, sizes)):
    buf = data[o_:o_ + n]
    ooff, osz = o.chunks(buf, "streaming", min_size=2048, max_size=65536)
    lo, hi = bt.first_chunk[b], bt.first_chunk[b + 1]
    ok = hi - lo == len(ooff)
    bad = []
    for i in range(min(hi - lo, len(ooff))):
        ch = bt.chunks[lo + i]
        if (ch.offset, ch.size) != (int(ooff[i]), int(osz[i])) or ch.hash_hex.decode() != hashlib.sha256(buf[int(ooff[i]):int(ooff[i] + osz[i])].tobytes()).hexdigest():
            bad.append(i)
    bh = C.string_at(C.addressof(bt.buffer_hash_hex.contents) + 65 * b).decode()
    print(b, n, o_, "chunks", hi - lo, len(ooff), "bad", bad[:5], "bufhash", bh == hashlib.sha256(buf.tobytes()).hexdigest())
