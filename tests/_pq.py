"""Helpers of the product-quantised engine's tests: a small product quantiser in numpy (what third_party/simeon does for the
reference — absent from the checkout, so this is a stand-in with the same DATA SHAPES: m sub-quantisers x 256 centroids,
one code byte per sub-quantiser, a per-query table of inner products), and stableStringKey (sqlite_vec_backend.cpp:141-148)."""
import numpy as np


def stable_string_key(s: str) -> int:      # FNV-1a 64, sqlite_vec_backend.cpp:141-148
    h = 1469598103934665603
    for b in s.encode():
        h ^= b
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def unit(rows):
    rows = np.asarray(rows, np.float32)
    n = np.sqrt((rows.astype(np.float64) ** 2).sum(axis=1))
    inv = (1.0 / np.sqrt(n.astype(np.float32) ** 2)).astype(np.float32)      # normalizeEmbeddingInPlace: 1 / sqrt((float)norm_sq)
    return (rows * inv[:, None]).astype(np.float32)


class Pq:
    """m x 256 centroids sampled from the unit rows' sub-vectors; encode = nearest centroid (L2) per sub-vector."""

    def __init__(self, rows_unit, m, seed=0):
        n, d = rows_unit.shape
        assert d % m == 0
        self.m, self.ds = m, d // m
        rng = np.random.default_rng(seed)
        pick = rng.choice(n, 256, replace=n < 256)
        self.centroids = np.stack([rows_unit[pick, j * self.ds:(j + 1) * self.ds] for j in range(m)]).astype(np.float32)   # [m][256][ds]

    def encode(self, rows_unit):
        n = rows_unit.shape[0]
        codes = np.empty((n, self.m), np.uint8)
        for j in range(self.m):
            sub = rows_unit[:, j * self.ds:(j + 1) * self.ds]
            d2 = ((sub[:, None, :] - self.centroids[j][None, :, :]) ** 2).sum(axis=2)
            codes[:, j] = d2.argmin(axis=1)
        return codes

    def lut(self, query_raw):
        """lut[j][c] = <sub-vector j of the NORMALISED query, centroid c>, fp32 (what PQInnerProductQuery holds, :3901)."""
        q = unit(np.asarray(query_raw, np.float32)[None, :])[0]
        return np.stack([self.centroids[j] @ q[j * self.ds:(j + 1) * self.ds] for j in range(self.m)]).astype(np.float32)


def numpy_pq_search(corpus, codes, lut, query, k, thr, rerank_factor, tie_keys, row_of_index, chunk_rank, candidates, sum_lanes):
    """An independent restatement of simeonPqSearchUnlocked (:3868-4056) in numpy / Python: pins oracle_pq_search against a
    second reading of the same text (NOT against the reference: simeon is absent)."""
    n, m = codes.shape
    if k == 0 or n == 0:
        return [], []
    nsq = float((query.astype(np.float64) ** 2).sum())
    if not (nsq > 1e-20):
        return [], []
    idxs = np.arange(n) if candidates is None else np.asarray(candidates, np.int64)
    if idxs.size == 0:
        return [], []
    approx = min(idxs.size, max(k, k * max(1, rerank_factor)))
    scored = []
    for i in idxs:
        vals = lut[np.arange(m), codes[i]]
        if sum_lanes == 1:
            s = np.float32(0)
            for v in vals:
                s = np.float32(s + v)
        else:
            part = [np.float32(0)] * sum_lanes
            for j, v in enumerate(vals):
                part[j % sum_lanes] = np.float32(part[j % sum_lanes] + v)
            s = np.float32(0)
            for p_ in part:
                s = np.float32(s + p_)
        scored.append((-float(s), int(tie_keys[i]) if tie_keys is not None else int(i), int(i)))
    scored.sort()
    recs = []
    for _, _, i in scored[:approx]:
        row = int(row_of_index[i]) if row_of_index is not None else i
        if row >= corpus.shape[0]:
            continue
        a, b = query.astype(np.float64), corpus[row].astype(np.float64)
        dp = 0.0; na = 0.0; nb = 0.0
        for x, y in zip(a, b):
            dp += x * y; na += x * x; nb += y * y
        na, nb = np.sqrt(na), np.sqrt(nb)
        sim = np.float32(0.0 if (na == 0 or nb == 0) else dp / (na * nb))
        if sim < np.float32(thr):
            continue
        recs.append((-float(sim), int(chunk_rank[row]) if chunk_rank is not None else row, row, sim))
    recs.sort()
    recs = recs[:k]
    return [r[2] for r in recs], [r[3] for r in recs]
