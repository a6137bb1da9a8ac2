"""Thin numpy-facing wrapper over the C ABI (one context = one device + one stream).

All compute happens in libyams_mi355x_accel.so on the GPU.  Device memory is addressed by integer
pointers (e.g. `tensor.data_ptr()`); `DeviceArray` is a minimal owner for hosts without torch.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (AccelError, CdcConfig, IngestResult, ScanCorpus, ScanDiag, ScanParams,
                   CDC_RABIN, CDC_STREAMING, SCAN_COSINE, SCAN_L2)

DEFAULT_POLY = 0x3DA3358B4DC173


def cdc_config(mode="streaming", window=48, min_size=16 * 1024, max_size=1024 * 1024,
               polynomial=DEFAULT_POLY, mask=0x1FFF, generic_kernel=False) -> CdcConfig:
    m = CDC_STREAMING if mode in ("streaming", CDC_STREAMING) else CDC_RABIN
    return CdcConfig(window, min_size, max_size, polynomial, mask, m,
                     _lib.CDC_FLAG_GENERIC_KERNEL if generic_kernel else 0)


class DeviceArray:
    """Owns a hipMalloc'd buffer; `.ptr` is the device address."""

    def __init__(self, acc: "Accel", nbytes: int):
        self.acc = acc
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        acc._check(acc.L.yams_accel_malloc(acc.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        self.acc._check(self.acc.L.yams_accel_upload(self.acc.ctx, self.ptr + offset,
                                                     arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            self.acc._check(self.acc.L.yams_accel_download(self.acc.ctx, out.ctypes.data,
                                                           self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.acc.L.yams_accel_free(self.acc.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


@dataclass
class ScanResult:
    scores: np.ndarray   # [nq, k] float32 (cosine similarity / relevance_score)
    rows: np.ndarray     # [nq, k] int64
    counts: np.ndarray   # [nq] uint32
    dist: np.ndarray     # [nq, k] float32
    diag: dict


class DedupSet:
    """Device-resident set of SHA-256 digests (chunk dedup lookup)."""

    def __init__(self, acc: "Accel", expected_entries: int = 0):
        self.acc = acc
        self.h = C.c_void_p()
        acc._check(acc.L.yams_dedup_set_create(acc.ctx, expected_entries, C.byref(self.h)))

    def close(self):
        if self.h:
            self.acc.L.yams_dedup_set_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_uint64(0)
        self.acc._check(self.acc.L.yams_dedup_set_size(self.h, C.byref(n)))
        return n.value

    def insert(self, digests: np.ndarray) -> np.ndarray:
        """digests: [n][32] uint8 (host).  Returns is_new[n] (bool)."""
        d = np.ascontiguousarray(digests, np.uint8).reshape(-1, 32)
        out = np.zeros(d.shape[0], np.uint8)
        nn = C.c_uint64(0)
        self.acc._check(self.acc.L.yams_dedup_insert_host(self.h, d.ctypes.data_as(C.c_void_p), d.shape[0],
                                                          out.ctypes.data_as(C.c_void_p), C.byref(nn)))
        assert nn.value == int(out.sum())
        return out.astype(bool)

    def probe(self, digests: np.ndarray) -> np.ndarray:
        d = np.ascontiguousarray(digests, np.uint8).reshape(-1, 32)
        out = np.zeros(d.shape[0], np.uint8)
        self.acc._check(self.acc.L.yams_dedup_probe_host(self.h, d.ctypes.data_as(C.c_void_p), d.shape[0],
                                                         out.ctypes.data_as(C.c_void_p)))
        return out.astype(bool)

    def insert_device(self, digests_ptr: int, n: int, sizes_ptr: int | None, is_new_ptr: int):
        nn, bn, bd = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.acc._check(self.acc.L.yams_dedup_insert_device(self.h, digests_ptr, n, sizes_ptr, is_new_ptr,
                                                            C.byref(nn), C.byref(bn), C.byref(bd)))
        return nn.value, bn.value, bd.value


class ShardedScan:
    """yams_scan_sharded_*: one search over a corpus row-sharded across several devices behind the C ABI — one
    process, one RCCL communicator over the shard devices, all-gather + merge per batch, `lanes` batches in
    flight (shards that share a device, the one-GPU tests, exchange their records with device copies).
    `ctx(i)` is an Accel bound to shard i's lane-0 context — upload the shard's rows and build its shadows
    through it.  collective: "auto" | "rccl" (require the communicator, also for one shard) | "peer";
    rccl_library: the collective library to bind instead of librccl.so.1 (the tests' stand-in lets ranks share a
    device); fence=False lifts the exchange fence (measurements); exchange_timeout_ms: the deadline of wait()
    (0 = 30 s; a batch that misses it raises AccelError TIMEOUT and the handle is stuck)."""

    def __init__(self, devices, lanes: int = 0, collective: str = "auto", rccl_library: str | None = None, fence: bool = True,
                 exchange_timeout_ms: int = 0):
        self.L = _lib.load()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        opt = _lib.ShardedOptions(C.sizeof(_lib.ShardedOptions), lanes,
                                  {"auto": _lib.SHARDED_COLLECTIVE_AUTO, "rccl": _lib.SHARDED_COLLECTIVE_RCCL,
                                   "peer": _lib.SHARDED_COLLECTIVE_PEER}[collective],
                                  _lib.SHARDED_FENCE_AUTO if fence else _lib.SHARDED_FENCE_OFF,
                                  rccl_library.encode() if rccl_library else None, exchange_timeout_ms, 0)
        st = self.L.yams_scan_sharded_create_ex(arr, len(devices), C.byref(opt), C.byref(h))
        if st != 0:
            raise AccelError(st, "yams_scan_sharded_create_ex failed")
        self.h = h
        self.n = len(devices)
        self.devices = list(devices)
        self.lanes = self.L.yams_scan_sharded_lanes(h)
        self._views = [self._borrow(i, 0) for i in range(self.n)]
        self._inflight = {}

    def _borrow(self, shard: int, lane: int) -> "Accel":
        a = Accel.__new__(Accel)
        a.L = self.L; a.ctx = C.c_void_p(self.L.yams_scan_sharded_lane_ctx(self.h, shard, lane)); a.device = self.devices[shard]
        a._borrowed = True
        return a

    def ctx(self, i: int) -> "Accel":
        return self._views[i]

    def lane_ctx(self, shard: int, lane: int) -> "Accel":
        """The context lane `lane` uses on shard `shard` (kernel timings of a pipelined run)."""
        return self._borrow(shard, lane)

    def info(self) -> dict:
        p = C.c_void_p()
        st = self.L.yams_scan_sharded_info_json(self.h, C.byref(p))
        if st != 0:
            raise AccelError(st, "yams_scan_sharded_info_json failed")
        try:
            return json.loads(C.string_at(p).decode())
        finally:
            self.L.yams_accel_free_string(p)

    def close(self):
        if getattr(self, "h", None):
            for a in self._views:
                a.ctx = None
            self.L.yams_scan_sharded_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _error(self, st):
        return AccelError(st, self.L.yams_scan_sharded_last_error(self.h).decode())

    # -- pipelined form ------------------------------------------------------------------------------
    def submit(self, shards, queries: np.ndarray, k: int, threshold: float = 0.0, metric: int = SCAN_COSINE,
               flags: int = 0, rank_of_row_ptr: int | None = None, rank_row_base: int = 0, want_diag: bool = True,
               block: bool = True) -> int:
        """Starts one batch on a free lane and returns the lane; wait(lane) returns its merged result."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        lane = C.c_uint32()
        st = self.L.yams_scan_sharded_lane_acquire(self.h, 1 if block else 0, C.byref(lane))
        if st != 0:
            raise self._error(st)
        prm = ScanParams(k, threshold, metric, flags)
        arr = (ScanCorpus * self.n)(*shards)
        st = self.L.yams_scan_sharded_submit(self.h, lane.value, arr, q.ctypes.data, q.shape[0], C.byref(prm), rank_of_row_ptr,
                                             rank_row_base, _lib.SHARDED_SUBMIT_DIAG if want_diag else 0)
        if st != 0:
            self.L.yams_scan_sharded_lane_release(self.h, lane.value)
            raise self._error(st)
        self._inflight[lane.value] = (q.shape[0], k, arr)       # (the views must outlive the batch)
        return lane.value

    def wait(self, lane: int) -> ScanResult:
        nq, k, _ = self._inflight.pop(lane)
        kk = max(k, 1)
        scores = np.full((nq, kk), -np.inf, np.float32)
        rows = np.full((nq, kk), -1, np.int64)
        counts = np.zeros(nq, np.uint32)
        dist = np.full((nq, kk), np.inf, np.float32)
        diag = ScanDiag()
        st = self.L.yams_scan_sharded_wait(self.h, lane, scores.ctypes.data, rows.ctypes.data, counts.ctypes.data,
                                           dist.ctypes.data, C.byref(diag))
        if st != 0:
            raise self._error(st)
        return ScanResult(scores[:, :k], rows[:, :k], counts, dist[:, :k], diag.as_dict())

    # -- one call ------------------------------------------------------------------------------------
    def topk(self, shards, queries: np.ndarray, k: int, threshold: float = 0.0, metric: int = SCAN_COSINE,
             flags: int = 0, rank_of_row_ptr: int | None = None, rank_row_base: int = 0) -> ScanResult:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq, kk = q.shape[0], max(k, 1)
        scores = np.full((nq, kk), -np.inf, np.float32)
        rows = np.full((nq, kk), -1, np.int64)
        counts = np.zeros(nq, np.uint32)
        dist = np.full((nq, kk), np.inf, np.float32)
        prm = ScanParams(k, threshold, metric, flags)
        diag = ScanDiag()
        arr = (ScanCorpus * self.n)(*shards)
        st = self.L.yams_scan_sharded_topk_host(self.h, arr, q.ctypes.data, nq, C.byref(prm), rank_of_row_ptr,
                                                rank_row_base, scores.ctypes.data, rows.ctypes.data,
                                                counts.ctypes.data, dist.ctypes.data, C.byref(diag))
        if st != 0:
            raise self._error(st)
        return ScanResult(scores[:, :k], rows[:, :k], counts, dist[:, :k], diag.as_dict())


class SweepGate:
    """Shared by the contexts of one device that search concurrently: their filter sweeps run one after the
    other on the GPU, everything around them overlaps (yams_accel_gate_create in the header)."""

    def __init__(self, device: int = 0):
        self.L = _lib.load()
        h = C.c_void_p()
        st = self.L.yams_accel_gate_create(device, C.byref(h))
        if st != 0:
            raise AccelError(st, "yams_accel_gate_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.yams_accel_gate_destroy(self.h)
        self.h = None


class Accel:
    def __init__(self, device: int = 0, stream: int | None = None):
        """stream: a hipStream_t handle, or None / 0 for a non-blocking stream of the context's own.  torch's DEFAULT stream has
        the handle 0: a context created with torch.cuda.current_stream().cuda_stream while no torch.cuda.Stream is current runs
        on its own stream, unordered with torch's work — torch.cuda.synchronize() before handing it tensors torch has just written."""
        self.L = _lib.load()
        if self.L.yams_accel_device_count() <= 0:
            raise AccelError(_lib.YAMS_ERR_UNSUPPORTED,
                             "no HIP device visible (the accelerator path has no CPU fallback)")
        ctx = C.c_void_p()
        st = self.L.yams_accel_ctx_create(device, C.c_void_p(stream) if stream else None,
                                          C.byref(ctx))
        if st != 0:
            raise AccelError(st, "yams_accel_ctx_create failed")
        self.ctx = ctx
        self.device = device

    def close(self):
        if getattr(self, "ctx", None) and not getattr(self, "_borrowed", False):
            self.L.yams_accel_ctx_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != 0:
            raise AccelError(st, self.L.yams_accel_last_error(self.ctx).decode())

    def set_sweep_hold(self, on: bool):
        """Keep the sweep gate closed behind this context's sweeps until release_sweep_hold (yams_accel_ctx_set_sweep_hold)."""
        self._check(self.L.yams_accel_ctx_set_sweep_hold(self.ctx, 1 if on else 0))

    def release_sweep_hold(self, stream_ptr: int | None = None):
        self._check(self.L.yams_accel_ctx_release_sweep_hold(self.ctx, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_gate(self, gate: "SweepGate | None"):
        """Attach this context to a sweep gate (None detaches).  The gate must outlive the context."""
        self._check(self.L.yams_accel_ctx_set_gate(self.ctx, gate.h if gate is not None else None))
        self._gate = gate

    # ---- misc ---------------------------------------------------------------------------------
    def device_info(self) -> dict:
        p = C.c_void_p()
        self._check(self.L.yams_accel_device_info_json(self.ctx, C.byref(p)))
        s = C.string_at(p).decode()
        self.L.yams_accel_free_string(p)
        return json.loads(s)

    def synchronize(self):
        self._check(self.L.yams_accel_ctx_synchronize(self.ctx))

    def alloc(self, nbytes: int) -> DeviceArray:
        return DeviceArray(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DeviceArray:
        arr = np.ascontiguousarray(arr)
        return DeviceArray(self, max(arr.nbytes, 16)).upload(arr)

    def enable_timing(self, on: bool = True):
        self._check(self.L.yams_accel_enable_kernel_timing(self.ctx, 1 if on else 0))

    def kernel_ms(self, name: str):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        st = self.L.yams_accel_last_kernel_ms(self.ctx, name.encode(), C.byref(ms), C.byref(n))
        if st == _lib.YAMS_ERR_NOT_FOUND:
            return None, 0
        self._check(st)
        return ms.value, n.value

    # ---- exact vector scan --------------------------------------------------------------------
    def corpus_view(self, rows_ptr: int, n_rows: int, dim: int, tie_rank_ptr: int | None = None,
                    rank_row_ptr: int | None = None, row_base: int = 0,
                    row_mask_ptr: int | None = None, row_mask_count: int = 0,
                    rows_bf16_ptr: int | None = None, rows_nsq_ptr: int | None = None,
                    rows_i8_ptr: int | None = None, rows_i8_meta_ptr: int | None = None,
                    stripe_rows: int = 0, n_stripes: int = 0, stripe_index: int = 0, i8_flags: int = 0) -> ScanCorpus:
        return ScanCorpus(rows_ptr, n_rows, dim, 0, tie_rank_ptr, rank_row_ptr, row_base,
                          row_mask_ptr, row_mask_count, rows_bf16_ptr, rows_nsq_ptr,
                          rows_i8_ptr, rows_i8_meta_ptr, stripe_rows, n_stripes, stripe_index, i8_flags)

    def build_shadow_device(self, rows_ptr: int, n_rows: int, dim: int, out_bf16_ptr: int,
                            out_nsq_ptr: int) -> None:
        """Filter shadow of the rows (bf16 RNE copy + fp32 squared norms); asynchronous."""
        self._check(self.L.yams_scan_build_shadow_device(self.ctx, rows_ptr, n_rows, dim,
                                                         out_bf16_ptr, out_nsq_ptr))

    def build_shadow_i8_device(self, rows_ptr: int, n_rows: int, dim: int, out_i8_ptr: int,
                               out_meta_ptr: int, want_mean_err: bool = False, first_row: int = 0, i8_flags: int = 0):
        """INT8 filter shadow of rows [first_row, first_row + n_rows) of the mirror whose arrays start
        at the given BASE pointers: int8 rows [n][dim] + {scale, residue bound} per block of 64 rows
        ([ceil(n / 64)][2] fp32), in the layout `i8_flags` names (the view must carry the same bits).
        Asynchronous unless the mean residue bound is asked for."""
        me = C.c_double(0.0)
        self._check(self.L.yams_scan_build_shadow_i8_layout_device(self.ctx, rows_ptr, first_row, n_rows, dim, i8_flags, out_i8_ptr,
                                                                   out_meta_ptr, C.byref(me) if want_mean_err else None))
        return me.value if want_mean_err else None

    def choose_i8_layout(self, rows_ptr: int, n_rows: int, dim: int):
        """(i8_flags, mean residue plain, mean residue rotated) measured on a sample of blocks; synchronises."""
        fl = C.c_uint32(0); a = C.c_double(0.0); b = C.c_double(0.0)
        self._check(self.L.yams_scan_choose_i8_layout_device(self.ctx, rows_ptr, n_rows, dim, C.byref(fl), C.byref(a), C.byref(b)))
        return fl.value, a.value, b.value

    def scan_topk_device(self, corpus: ScanCorpus, queries_ptr: int, nq: int, k: int,
                         threshold: float, metric: int, out_scores: int, out_rows: int,
                         out_counts: int, out_dist: int | None = None,
                         out_ranks: int | None = None, flags: int = 0, want_diag: bool = True):
        prm = ScanParams(k, threshold, metric, flags)
        diag = ScanDiag()
        self._check(self.L.yams_scan_topk_device(self.ctx, C.byref(corpus), queries_ptr, nq,
                                                 C.byref(prm), out_scores, out_rows, out_counts,
                                                 out_dist, out_ranks,
                                                 C.byref(diag) if want_diag else None))
        return diag.as_dict() if want_diag else None

    def scan_topk(self, corpus: ScanCorpus, queries: np.ndarray, k: int, threshold: float = 0.0,
                  metric: int = SCAN_COSINE, flags: int = 0) -> ScanResult:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        kk = max(k, 1)
        scores = np.full((nq, kk), -np.inf, np.float32)
        rows = np.full((nq, kk), -1, np.int64)
        counts = np.zeros(nq, np.uint32)
        dist = np.full((nq, kk), np.inf, np.float32)
        prm = ScanParams(k, threshold, metric, flags)
        diag = ScanDiag()
        self._check(self.L.yams_scan_topk_host(self.ctx, C.byref(corpus), q.ctypes.data, nq,
                                               C.byref(prm), scores.ctypes.data, rows.ctypes.data,
                                               counts.ctypes.data, dist.ctypes.data,
                                               C.byref(diag)))
        return ScanResult(scores[:, :k], rows[:, :k], counts, dist[:, :k], diag.as_dict())

    def scan_pq_topk(self, corpus: ScanCorpus, codes: np.ndarray, luts: np.ndarray, queries: np.ndarray, k: int, threshold: float = -1.0,
                     rerank_factor: int = 2, tie_keys: np.ndarray | None = None, row_of_index: np.ndarray | None = None,
                     candidates: np.ndarray | None = None, sum_lanes: int = 1) -> ScanResult:
        """The product-quantised engine's search (yams_scan_pq_topk_device; simeonPqSearchUnlocked, sqlite_vec_backend.cpp:
        3868-4056) from host arrays: codes u8 [n][m], luts f32 [nq][m][256] (what simeon's PQInnerProductQuery holds), raw
        queries [nq][dim], tie_keys u64 [n] (stableStringKey of the chunk ids), row_of_index u32 [n] (index -> corpus row),
        candidates: ascending indices or None.  The tie ranks and the key -> row table are derived here, as a host does once
        per index build."""
        codes = np.ascontiguousarray(codes, np.uint8)
        n, m = codes.shape
        q = np.ascontiguousarray(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        luts = np.ascontiguousarray(luts, np.float32).reshape(nq, m, 256)
        keep = []
        d_codes = self.to_device(codes); keep.append(d_codes)
        d_tie = d_keyrow = None
        order = None
        if tie_keys is not None:
            tk = np.ascontiguousarray(tie_keys, np.uint64)
            order = np.lexsort((np.arange(n), tk))            # ascending key, equal keys by index
            rank = np.empty(n, np.uint32); rank[order] = np.arange(n, dtype=np.uint32)
            d_tie = self.to_device(rank); keep.append(d_tie)
        if tie_keys is not None or row_of_index is not None:
            roi = np.arange(n, dtype=np.uint32) if row_of_index is None else np.ascontiguousarray(row_of_index, np.uint32)
            key_row = roi[order] if order is not None else roi  # key index r -> row of the code with that key index
            d_keyrow = self.to_device(np.ascontiguousarray(key_row, np.uint32)); keep.append(d_keyrow)
        pq = _lib.ScanPqIndex(d_codes.ptr, n, m, 0, d_tie.ptr if d_tie else None, d_keyrow.ptr if d_keyrow else None)
        d_q = self.to_device(q); d_l = self.to_device(luts); keep += [d_q, d_l]
        d_c = None; n_c = 0
        if candidates is not None:
            cand = np.ascontiguousarray(candidates, np.uint32)
            n_c = cand.size
            d_c = self.to_device(cand if n_c else np.zeros(1, np.uint32)); keep.append(d_c)
        kk = max(k, 1)
        d_s = self.alloc(nq * kk * 4); d_r = self.alloc(nq * kk * 8); d_n = self.alloc(nq * 4)
        prm = _lib.ScanPqParams(k, threshold, rerank_factor, {1: 0, 4: 1, 8: 2, 16: 3}[sum_lanes])
        diag = ScanDiag()
        try:
            self._check(self.L.yams_scan_pq_topk_device(self.ctx, C.byref(corpus), C.byref(pq), d_q.ptr, d_l.ptr, nq, C.byref(prm),
                                                        d_c.ptr if d_c else None, n_c, d_s.ptr, d_r.ptr, d_n.ptr, C.byref(diag)))
            counts = d_n.download(np.uint32, nq)
            scores = d_s.download(np.float32, nq * kk).reshape(nq, kk)
            rows = d_r.download(np.int64, nq * kk).reshape(nq, kk)
        finally:
            for b in keep + [d_s, d_r, d_n]:
                b.free()
        return ScanResult(scores[:, :k], rows[:, :k], counts, None, diag.as_dict())

    def merge_topk_device(self, n_shards, nq, k, threshold, metric, in_scores, in_rows, in_counts,
                          in_dist, in_ranks, out_scores, out_rows, out_counts, out_dist):
        prm = ScanParams(k, threshold, metric, 0)
        self._check(self.L.yams_scan_merge_topk_device(self.ctx, n_shards, nq, C.byref(prm),
                                                       in_scores, in_rows, in_counts, in_dist,
                                                       in_ranks, out_scores, out_rows, out_counts,
                                                       out_dist))

    def synth_rows(self, seed: int, row0: int, n_rows: int, dim: int, out_ptr: int):
        self._check(self.L.yams_synth_rows_device(self.ctx, seed, row0, n_rows, dim, out_ptr))

    def synth_bytes(self, seed: int, blob0: int, n_blobs: int, blob_len: int, out_ptr: int):
        self._check(self.L.yams_synth_bytes_device(self.ctx, seed, blob0, n_blobs, blob_len, out_ptr))

    # ---- SHA-256 ------------------------------------------------------------------------------
    def sha256_hex(self, data: bytes | np.ndarray) -> str:
        a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, np.uint8)
        out = C.create_string_buffer(65)
        self._check(self.L.yams_sha256_host(self.ctx, a.ctypes.data if a.size else None, a.size, out))
        return out.value.decode()

    def sha256_many(self, msgs: list) -> list[str]:
        arrs = [np.frombuffer(bytes(m), np.uint8) if not isinstance(m, np.ndarray)
                else np.ascontiguousarray(m, np.uint8) for m in msgs]
        n = len(arrs)
        if n == 0:
            return []
        ptrs = (C.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * n)(*[a.size for a in arrs])
        out = C.create_string_buffer(65 * n)
        self._check(self.L.yams_sha256_many_host(self.ctx, ptrs, lens, n, out))
        raw = out.raw
        return [raw[65 * i:65 * i + 64].decode() for i in range(n)]

    def sha256_batch_device(self, data_ptr, offsets_ptr, lengths_ptr, n, digests_ptr):
        self._check(self.L.yams_sha256_batch_device(self.ctx, data_ptr, offsets_ptr, lengths_ptr,
                                                    n, digests_ptr))

    # ---- chunking / ingest ----------------------------------------------------------------------
    def verify_chunks_device(self, data_ptr, offsets_ptr, lengths_ptr, n, expected_ptr, valid_ptr) -> int:
        """Batched integrity check; returns the number of mismatching chunks."""
        bad = C.c_uint64(0)
        self._check(self.L.yams_verify_chunks_device(self.ctx, data_ptr, offsets_ptr, lengths_ptr, n,
                                                     expected_ptr, valid_ptr, C.byref(bad)))
        return bad.value

    def dedup_set(self, expected_entries: int = 0) -> "DedupSet":
        return DedupSet(self, expected_entries)

    def chunk(self, data, cfg: CdcConfig | None = None, with_hashes: bool = True, context_len: int = 0):
        """IChunker::chunkDataLazy over host memory -> (offsets, sizes, hex hashes | None).  context_len > 0: one window
        of a stream (yams_cdc_chunk_window_host) — the leading bytes are history, chunks start behind them."""
        cfg = cfg or cdc_config()
        a = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else \
            np.ascontiguousarray(data, np.uint8)
        floor = max(1, int(cfg.min_size))
        cap = a.size // floor + 2
        off = np.zeros(cap, np.uint64)
        sz = np.zeros(cap, np.uint64)
        hexbuf = C.create_string_buffer(65 * cap) if with_hashes else None
        cnt = C.c_size_t(0)
        self._check(self.L.yams_cdc_chunk_window_host(self.ctx, a.ctypes.data if a.size else None, a.size, context_len,
                                                      C.byref(cfg), off.ctypes.data_as(_lib.u64p),
                                                      sz.ctypes.data_as(_lib.u64p), hexbuf, cap,
                                                      C.byref(cnt)))
        n = cnt.value
        hashes = None
        if with_hashes:
            raw = hexbuf.raw
            hashes = [raw[65 * i:65 * i + 64].decode() for i in range(n)]
        return off[:n].copy(), sz[:n].copy(), hashes

    def ingest_device(self, data_ptr: int, blob_offsets, blob_lengths, cfg: CdcConfig | None = None,
                      flags: int = 3) -> IngestResult:
        cfg = cfg or cdc_config()
        bo = np.ascontiguousarray(blob_offsets, np.uint64)
        bl = np.ascontiguousarray(blob_lengths, np.uint64)
        res = IngestResult()
        self._check(self.L.yams_ingest_device(self.ctx, data_ptr, bo.ctypes.data_as(_lib.u64p),
                                              bl.ctypes.data_as(_lib.u64p), bo.size, C.byref(cfg),
                                              flags, C.byref(res)))
        return res

    def ingest_host(self, blob_ptrs, blob_lengths, cfg: CdcConfig | None = None, flags: int = 3,
                    batch_bytes: int = 0, chunk_cap: int | None = None) -> dict:
        """yams_ingest_host: blobs in host memory (addresses in blob_ptrs), streamed through the device
        in batches.  Returns blob_first, chunk_offset, chunk_size, chunk_digest, blob_digest (numpy)."""
        cfg = cfg or cdc_config()
        bl = np.ascontiguousarray(blob_lengths, np.uint64)
        n = int(bl.size)
        ptrs = (C.c_void_p * max(n, 1))(*[int(p) if p else None for p in blob_ptrs])
        if chunk_cap is None:
            chunk_cap = int(sum(int(x) // max(int(cfg.min_size), 1) + 2 for x in bl))
        first = np.zeros(n + 1, np.uint64)
        off = np.empty(chunk_cap, np.uint64); sz = np.empty(chunk_cap, np.uint64)
        dg = np.empty((chunk_cap, 32), np.uint8) if flags & 1 else None
        bd = np.empty((n, 32), np.uint8) if flags & 2 else None
        cnt = C.c_uint64(0)
        rc = self.L.yams_ingest_host(self.ctx, ptrs, bl.ctypes.data_as(_lib.u64p), n, C.byref(cfg), flags, batch_bytes,
                                     first.ctypes.data_as(_lib.u64p), off.ctypes.data_as(_lib.u64p),
                                     sz.ctypes.data_as(_lib.u64p), dg.ctypes.data if dg is not None else None,
                                     chunk_cap, bd.ctypes.data if bd is not None else None, C.byref(cnt))
        self.last_required_chunks = int(cnt.value)
        self._check(rc)
        m = int(cnt.value)
        return {"n_chunks": m, "blob_first": first, "chunk_offset": off[:m], "chunk_size": sz[:m],
                "chunk_digest": dg[:m] if dg is not None else None, "blob_digest": bd}

    def download(self, ptr: int, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            self._check(self.L.yams_accel_download(self.ctx, out.ctypes.data, ptr, out.nbytes))
        return out

    def fetch_ingest(self, res: IngestResult, n_blobs: int) -> dict:
        n = res.n_chunks
        out = {"n_chunks": n,
               "chunk_offset": self.download(res.chunk_offset, np.uint64, n),
               "chunk_size": self.download(res.chunk_size, np.uint64, n),
               "chunk_blob": self.download(res.chunk_blob, np.uint32, n),
               "blob_first": self.download(res.blob_first, np.uint64, n_blobs + 1)}
        if res.chunk_digest:
            out["chunk_digest"] = self.download(res.chunk_digest, np.uint8, n * 32).reshape(n, 32)
        if res.blob_digest:
            out["blob_digest"] = self.download(res.blob_digest, np.uint8, n_blobs * 32).reshape(n_blobs, 32)
        return out
