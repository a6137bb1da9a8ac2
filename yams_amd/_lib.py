"""ctypes binding of libyams_mi355x_accel.so (the C ABI declared in include/yams_mi355x_accel.h).

The library is the product; this module only loads it and declares signatures.  It fails loudly
when the shared object is missing — there is no Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libyams_mi355x_accel.so")

YAMS_OK, YAMS_ERR_INVALID_ARG, YAMS_ERR_NOT_FOUND, YAMS_ERR_IO, YAMS_ERR_INTERNAL, \
    YAMS_ERR_UNSUPPORTED, YAMS_ERR_TIMEOUT, YAMS_ERR_RESOURCE_EXHAUSTED = range(8)
STATUS_NAMES = {0: "OK", 1: "INVALID_ARG", 2: "NOT_FOUND", 3: "IO", 4: "INTERNAL", 5: "UNSUPPORTED", 6: "TIMEOUT",
                7: "RESOURCE_EXHAUSTED"}
SCAN_COSINE, SCAN_L2 = 0, 1
CDC_RABIN, CDC_STREAMING = 0, 1
FLAG_DEFER_THRESHOLD, FLAG_FORCE_EXACT, FLAG_F32_FILTER, FLAG_SPLIT_FILTER, FLAG_RECORD_PATH = 1, 2, 4, 8, 16
FLAG_WIDE_TILE = 32
FLAG_NO_I8_FILTER = 64
FLAG_RESIDENT_QUERIES = 128
FLAG_L2_ACC_F64, FLAG_L2_ACC_F32, FLAG_L2_ACC_F32X8, FLAG_L2_ACC_F32X16 = 0, 256, 512, 768
FLAG_L2_ACC_EXPLICIT = 1024
FLAG_L2_ACC_FUSED = 2048


def i8_shadow_rows(n_rows: int) -> int:
    """YAMS_SCAN_I8_SHADOW_ROWS: the int8 shadow is padded to whole blocks of 64 rows."""
    return (int(n_rows) + 63) // 64 * 64
TIER_NONE, TIER_I8, TIER_BF16, TIER_SPLIT, TIER_F32 = range(5)
CDC_FLAG_GENERIC_KERNEL = 1
INGEST_CHUNK_DIGESTS, INGEST_BLOB_DIGESTS = 1, 2
I8_ROTATED = 1      # YAMS_SCAN_I8_ROTATED (yams_scan_corpus_t.i8_flags)

vp = C.c_void_p
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)


class ScanCorpus(C.Structure):
    _fields_ = [("rows", vp), ("n_rows", C.c_uint64), ("dim", C.c_uint32), ("reserved", C.c_uint32),
                ("tie_rank", vp), ("rank_row", vp), ("row_base", C.c_int64),
                ("row_mask", vp), ("row_mask_count", C.c_uint64),
                ("rows_bf16", vp), ("rows_nsq", vp), ("rows_i8", vp), ("rows_i8_meta", vp),
                ("stripe_rows", C.c_uint32), ("n_stripes", C.c_uint32), ("stripe_index", C.c_uint32),
                ("i8_flags", C.c_uint32)]


class ShardedOptions(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("lanes", C.c_uint32), ("collective", C.c_uint32), ("fence", C.c_uint32),
                ("rccl_library", C.c_char_p), ("exchange_timeout_ms", C.c_uint32), ("reserved0", C.c_uint32)]


SHARDED_COLLECTIVE_AUTO, SHARDED_COLLECTIVE_RCCL, SHARDED_COLLECTIVE_PEER = 0, 1, 2
SHARDED_FENCE_AUTO, SHARDED_FENCE_OFF = 0, 1
SHARDED_SUBMIT_DIAG = 1


class RecordLayout(C.Structure):
    _fields_ = [("scores_off", C.c_uint64), ("rows_off", C.c_uint64), ("counts_off", C.c_uint64),
                ("dist_off", C.c_uint64), ("ranks_off", C.c_uint64), ("bytes", C.c_uint64)]


class ScanParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("similarity_threshold", C.c_float), ("metric", C.c_uint32),
                ("flags", C.c_uint32)]


class ScanPqIndex(C.Structure):     # yams_scan_pq_index_t
    _fields_ = [("codes", vp), ("n_codes", C.c_uint64), ("m", C.c_uint32), ("reserved", C.c_uint32),
                ("tie_rank", vp), ("key_row", vp)]


class ScanPqParams(C.Structure):    # yams_scan_pq_params_t
    _fields_ = [("k", C.c_uint32), ("similarity_threshold", C.c_float), ("rerank_factor", C.c_uint32), ("flags", C.c_uint32)]


PQ_SUM_SEQUENTIAL, PQ_SUM_X4, PQ_SUM_X8, PQ_SUM_X16 = 0, 1, 2, 3


class ScanDiag(C.Structure):
    _fields_ = [("used_exact_scan", C.c_uint32), ("rows_visited_observed", C.c_uint32),
                ("rows_visited", C.c_uint64), ("exact_distance_evaluations", C.c_uint64),
                ("returned_rows", C.c_uint64), ("filter_candidates", C.c_uint64),
                ("rescored_rows", C.c_uint64), ("widened_queries", C.c_uint32),
                ("exact_fallback_queries", C.c_uint32), ("path", C.c_uint32),
                ("escalated_queries", C.c_uint32), ("filter_tier", C.c_uint32), ("retried_queries", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class CdcConfig(C.Structure):
    _fields_ = [("window_size", C.c_uint64), ("min_size", C.c_uint64), ("max_size", C.c_uint64),
                ("polynomial", C.c_uint64), ("mask", C.c_uint64), ("mode", C.c_uint32),
                ("flags", C.c_uint32)]


class IngestResult(C.Structure):
    _fields_ = [("n_chunks", C.c_uint64), ("chunk_offset", vp), ("chunk_size", vp),
                ("chunk_blob", vp), ("blob_first", vp), ("chunk_digest", vp), ("blob_digest", vp)]


class ScanHit(C.Structure):
    _fields_ = [("row", C.c_int64), ("similarity", C.c_float), ("distance", C.c_float)]


class ChunkRef(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("size", C.c_uint64), ("hash_hex", C.c_char * 65),
                ("pad", C.c_char * 7)]


ST = C.c_int
_vs_fn = C.CFUNCTYPE


class VectorScanV1(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("self", vp),
        ("corpus_create", C.CFUNCTYPE(ST, vp, C.c_uint32, u64p)),
        ("corpus_append", C.CFUNCTYPE(ST, vp, C.c_uint64, f32p, C.c_uint64)),
        ("corpus_set_tie_ranks", C.CFUNCTYPE(ST, vp, C.c_uint64, u32p, C.c_uint64)),
        ("corpus_clear", C.CFUNCTYPE(ST, vp, C.c_uint64)),
        ("corpus_destroy", C.CFUNCTYPE(ST, vp, C.c_uint64)),
        ("corpus_size", C.CFUNCTYPE(ST, vp, C.c_uint64, u64p, u32p)),
        ("search_batch", C.CFUNCTYPE(ST, vp, C.c_uint64, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_float, C.c_uint32, C.POINTER(C.POINTER(ScanHit)),
                                     C.POINTER(u32p), C.POINTER(ScanDiag))),
        ("free_hits", C.CFUNCTYPE(None, vp, C.POINTER(ScanHit), u32p)),
        ("get_runtime_info_json", C.CFUNCTYPE(ST, vp, C.POINTER(C.c_void_p))),
        ("free_string", C.CFUNCTYPE(None, vp, C.c_void_p)),
        ("search_batch_masked", C.CFUNCTYPE(ST, vp, C.c_uint64, f32p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_float, C.c_uint32, u32p,
                                            C.POINTER(C.POINTER(ScanHit)), C.POINTER(u32p),
                                            C.POINTER(ScanDiag))),
        ("search_batch_ex", C.CFUNCTYPE(ST, vp, C.c_uint64, f32p, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, u32p,
                                        C.POINTER(C.POINTER(ScanHit)), C.POINTER(u32p),
                                        C.POINTER(ScanDiag))),
        # version 2: the product-quantised engine over the mirror
        ("pq_index_set", C.CFUNCTYPE(ST, vp, C.c_uint64, u8p, C.c_uint64, C.c_uint32, u64p, u32p)),
        ("search_pq", C.CFUNCTYPE(ST, vp, C.c_uint64, f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32,
                                  C.c_uint32, u32p, C.c_uint64, C.POINTER(C.POINTER(ScanHit)), C.POINTER(u32p),
                                  C.POINTER(ScanDiag))),
    ]


class ContentHashV1(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("self", vp),
        ("hash", C.CFUNCTYPE(ST, vp, u8p, C.c_size_t, C.c_char_p)),
        ("hash_many", C.CFUNCTYPE(ST, vp, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_size_t,
                                  C.c_char_p)),
        ("stream_create", C.CFUNCTYPE(ST, vp, C.POINTER(vp))),
        ("stream_init", C.CFUNCTYPE(ST, vp, vp)),
        ("stream_update", C.CFUNCTYPE(ST, vp, vp, u8p, C.c_size_t)),
        ("stream_finalize", C.CFUNCTYPE(ST, vp, vp, C.c_char_p)),
        ("stream_destroy", C.CFUNCTYPE(None, vp, vp)),
        ("verify_many", C.CFUNCTYPE(ST, vp, C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, u8p)),
        ("dedup_create", C.CFUNCTYPE(ST, vp, C.c_uint64, u64p)),
        ("dedup_insert", C.CFUNCTYPE(ST, vp, C.c_uint64, C.c_char_p, C.c_size_t, u8p)),
        ("dedup_contains", C.CFUNCTYPE(ST, vp, C.c_uint64, C.c_char_p, C.c_size_t, u8p)),
        ("dedup_size", C.CFUNCTYPE(ST, vp, C.c_uint64, u64p)),
        ("dedup_destroy", C.CFUNCTYPE(ST, vp, C.c_uint64)),
    ]


class ChunkBatch(C.Structure):
    _fields_ = [("n_buffers", C.c_size_t), ("n_chunks", C.c_size_t), ("first_chunk", C.POINTER(C.c_size_t)),
                ("chunks", C.POINTER(ChunkRef)), ("buffer_hash_hex", C.POINTER(C.c_char))]


CHUNK_MANY_BUFFER_HASHES = 1
HASH_LONE_CHAIN_MAX = 1 << 20


class ChunkerV1(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("self", vp),
        ("get_default_config", C.CFUNCTYPE(ST, vp, C.c_uint32, C.POINTER(CdcConfig))),
        ("chunk_data", C.CFUNCTYPE(ST, vp, u8p, C.c_size_t, C.POINTER(CdcConfig),
                                   C.POINTER(C.POINTER(ChunkRef)), C.POINTER(C.c_size_t))),
        ("free_chunks", C.CFUNCTYPE(None, vp, C.POINTER(ChunkRef), C.c_size_t)),
        ("chunk_many", C.CFUNCTYPE(ST, vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(CdcConfig), C.c_uint32,
                                   C.POINTER(C.POINTER(ChunkBatch)))),
        ("free_chunk_batch", C.CFUNCTYPE(None, vp, C.POINTER(ChunkBatch))),
        ("chunk_window", C.CFUNCTYPE(ST, vp, u8p, C.c_size_t, C.c_size_t, C.POINTER(CdcConfig),
                                     C.POINTER(C.POINTER(ChunkRef)), C.POINTER(C.c_size_t))),
    ]


# Every symbol include/yams_mi355x_accel.h declares + the abi.h plugin entry points.
EXPORTS = [
    "yams_accel_device_count", "yams_accel_ctx_create", "yams_accel_ctx_destroy",
    "yams_accel_ctx_synchronize", "yams_accel_gate_create", "yams_accel_gate_destroy", "yams_accel_ctx_set_gate", "yams_accel_ctx_set_sweep_hold", "yams_accel_ctx_release_sweep_hold",
    "yams_accel_last_error", "yams_accel_device_info_json",
    "yams_accel_free_string", "yams_accel_trim", "yams_accel_malloc", "yams_accel_free", "yams_accel_upload",
    "yams_accel_download", "yams_accel_last_kernel_ms", "yams_accel_enable_kernel_timing",
    "yams_accel_debug_fail_alloc_after", "yams_accel_debug_alloc_faults", "yams_accel_debug_alloc_injection_compiled",
    "yams_scan_topk_device", "yams_scan_topk_host", "yams_scan_merge_topk_device", "yams_scan_pq_topk_device",
    "yams_scan_build_shadow_device", "yams_scan_build_shadow_i8_device", "yams_scan_build_shadow_i8_layout_device", "yams_scan_choose_i8_layout_device",
    "yams_scan_record_layout", "yams_scan_merge_records_device", "yams_scan_sharded_create",
    "yams_scan_sharded_destroy", "yams_scan_sharded_count", "yams_scan_sharded_ctx",
    "yams_scan_sharded_last_error", "yams_scan_sharded_topk_host", "yams_scan_sharded_create_ex",
    "yams_scan_sharded_lanes", "yams_scan_sharded_info_json", "yams_scan_sharded_lane_ctx",
    "yams_scan_sharded_lane_acquire", "yams_scan_sharded_lane_release", "yams_scan_sharded_submit",
    "yams_scan_sharded_wait",
    "yams_synth_rows_device", "yams_synth_bytes_device", "yams_sha256_batch_device",
    "yams_sha256_host", "yams_sha256_many_host", "yams_verify_chunks_device", "yams_cdc_default_config",
    "yams_dedup_set_create", "yams_dedup_set_destroy", "yams_dedup_set_size", "yams_dedup_insert_device",
    "yams_dedup_probe_device", "yams_dedup_insert_host", "yams_dedup_probe_host",
    "yams_cdc_chunk_device", "yams_ingest_device", "yams_ingest_host", "yams_cdc_chunk_host", "yams_cdc_chunk_window_host",
    "yams_plugin_get_abi_version", "yams_plugin_get_name", "yams_plugin_get_version",
    "yams_plugin_get_manifest_json", "yams_plugin_init", "yams_plugin_shutdown",
    "yams_plugin_get_interface", "yams_plugin_get_health_json",
]

_lib = None


def load(share_torch_runtime: bool = True) -> C.CDLL:
    """dlopen the accelerator library.

    torch's ROCm wheel ships its own libamdhip64 (SONAME libamdhip64.so.7).  A process that uses
    both must have ONE HIP runtime, otherwise pointers and streams do not cross; importing torch
    first makes the dynamic loader resolve this library's libamdhip64.so.7 to the copy torch has
    already mapped.  Hosts that never touch torch pass share_torch_runtime=False.
    """
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    if os.environ.get("YAMS_ACCEL_MEASURE_LIB"):
        # scripts/ only: the measurement build (ablation kernels + environment knobs), same ABI.
        # This is the Python loader choosing a file; the product library itself reads no environment.
        LIB_PATH = os.path.join(HERE, "lib", "libyams_mi355x_accel_measure.so")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m yams_amd.build` "
            "(there is no CPU fallback for the accelerator path)")
    if share_torch_runtime and "torch" not in sys.modules \
            and os.environ.get("YAMS_ACCEL_NO_TORCH") is None \
            and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401  (plumbing only: puts torch's HIP runtime in the process first)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols: {missing}")
    L.yams_accel_device_count.restype = C.c_int
    L.yams_accel_debug_fail_alloc_after.argtypes = [C.c_int64]
    L.yams_accel_debug_fail_alloc_after.restype = None
    L.yams_accel_debug_alloc_faults.restype = C.c_uint64
    L.yams_accel_debug_alloc_injection_compiled.restype = C.c_int
    L.yams_accel_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.yams_accel_ctx_destroy.argtypes = [vp]
    L.yams_accel_ctx_destroy.restype = None
    L.yams_accel_ctx_synchronize.argtypes = [vp]
    L.yams_accel_gate_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.yams_accel_gate_destroy.argtypes = [vp]
    L.yams_accel_gate_destroy.restype = None
    L.yams_accel_ctx_set_gate.argtypes = [vp, vp]
    L.yams_accel_ctx_set_sweep_hold.argtypes = [vp, C.c_int]
    L.yams_accel_ctx_release_sweep_hold.argtypes = [vp, vp]
    L.yams_accel_last_error.argtypes = [vp]
    L.yams_accel_last_error.restype = C.c_char_p
    L.yams_accel_device_info_json.argtypes = [vp, C.POINTER(vp)]
    L.yams_accel_trim.argtypes = [C.c_int]
    L.yams_accel_trim.restype = C.c_uint64
    L.yams_accel_free_string.argtypes = [vp]
    L.yams_accel_free_string.restype = None
    L.yams_accel_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.yams_accel_free.argtypes = [vp, vp]
    L.yams_accel_free.restype = None
    L.yams_accel_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.yams_accel_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.yams_accel_last_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), u64p]
    L.yams_accel_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.yams_scan_topk_device.argtypes = [vp, C.POINTER(ScanCorpus), vp, C.c_uint32,
                                        C.POINTER(ScanParams), vp, vp, vp, vp, vp,
                                        C.POINTER(ScanDiag)]
    L.yams_scan_pq_topk_device.argtypes = [vp, C.POINTER(ScanCorpus), C.POINTER(ScanPqIndex), vp, vp, C.c_uint32, C.POINTER(ScanPqParams),
                                           vp, C.c_uint64, vp, vp, vp, C.POINTER(ScanDiag)]
    L.yams_scan_topk_host.argtypes = [vp, C.POINTER(ScanCorpus), vp, C.c_uint32,
                                      C.POINTER(ScanParams), vp, vp, vp, vp, C.POINTER(ScanDiag)]
    L.yams_scan_build_shadow_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp]
    L.yams_scan_build_shadow_i8_device.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, C.POINTER(C.c_double)]
    L.yams_scan_build_shadow_i8_layout_device.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(C.c_double)]
    L.yams_scan_choose_i8_layout_device.argtypes = [vp, vp, C.c_uint64, C.c_uint32, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.yams_scan_merge_topk_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(ScanParams),
                                              vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.yams_scan_record_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(RecordLayout)]
    L.yams_scan_record_layout.restype = None
    L.yams_scan_merge_records_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(ScanParams), vp, C.c_uint64,
                                                 C.POINTER(RecordLayout), vp, C.c_int64, vp, vp, vp, vp]
    L.yams_scan_sharded_create.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    L.yams_scan_sharded_destroy.argtypes = [vp]
    L.yams_scan_sharded_destroy.restype = None
    L.yams_scan_sharded_count.argtypes = [vp]
    L.yams_scan_sharded_count.restype = C.c_uint32
    L.yams_scan_sharded_ctx.argtypes = [vp, C.c_uint32]
    L.yams_scan_sharded_ctx.restype = vp
    L.yams_scan_sharded_last_error.argtypes = [vp]
    L.yams_scan_sharded_last_error.restype = C.c_char_p
    L.yams_scan_sharded_topk_host.argtypes = [vp, C.POINTER(ScanCorpus), vp, C.c_uint32, C.POINTER(ScanParams), vp,
                                              C.c_int64, vp, vp, vp, vp, C.POINTER(ScanDiag)]
    L.yams_scan_sharded_create_ex.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(ShardedOptions), C.POINTER(vp)]
    L.yams_scan_sharded_lanes.argtypes = [vp]
    L.yams_scan_sharded_lanes.restype = C.c_uint32
    L.yams_scan_sharded_info_json.argtypes = [vp, C.POINTER(vp)]
    L.yams_scan_sharded_lane_ctx.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.yams_scan_sharded_lane_ctx.restype = vp
    L.yams_scan_sharded_lane_acquire.argtypes = [vp, C.c_int, u32p]
    L.yams_scan_sharded_lane_release.argtypes = [vp, C.c_uint32]
    L.yams_scan_sharded_lane_release.restype = None
    L.yams_scan_sharded_submit.argtypes = [vp, C.c_uint32, C.POINTER(ScanCorpus), vp, C.c_uint32, C.POINTER(ScanParams), vp,
                                           C.c_int64, C.c_uint32]
    L.yams_scan_sharded_wait.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.POINTER(ScanDiag)]
    L.yams_synth_rows_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp]
    L.yams_synth_bytes_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.yams_sha256_batch_device.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
    L.yams_verify_chunks_device.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp, u64p]
    L.yams_dedup_set_create.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.yams_dedup_set_destroy.argtypes = [vp]
    L.yams_dedup_set_destroy.restype = None
    L.yams_dedup_set_size.argtypes = [vp, u64p]
    L.yams_dedup_insert_device.argtypes = [vp, vp, C.c_uint64, vp, vp, u64p, u64p, u64p]
    L.yams_dedup_probe_device.argtypes = [vp, vp, C.c_uint64, vp]
    L.yams_dedup_insert_host.argtypes = [vp, vp, C.c_uint64, vp, u64p]
    L.yams_dedup_probe_host.argtypes = [vp, vp, C.c_uint64, vp]
    L.yams_sha256_host.argtypes = [vp, vp, C.c_size_t, C.c_char_p]
    L.yams_sha256_many_host.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t,
                                        C.c_char_p]
    L.yams_cdc_default_config.argtypes = [C.POINTER(CdcConfig), C.c_uint32]
    L.yams_cdc_default_config.restype = None
    L.yams_cdc_chunk_device.argtypes = [vp, vp, u64p, u64p, C.c_uint64, C.POINTER(CdcConfig),
                                        C.POINTER(IngestResult)]
    L.yams_ingest_device.argtypes = [vp, vp, u64p, u64p, C.c_uint64, C.POINTER(CdcConfig),
                                     C.c_uint32, C.POINTER(IngestResult)]
    L.yams_ingest_host.argtypes = [vp, C.POINTER(vp), u64p, C.c_uint64, C.POINTER(CdcConfig), C.c_uint32, C.c_uint64,
                                   u64p, u64p, u64p, vp, C.c_uint64, vp, u64p]
    L.yams_cdc_chunk_host.argtypes = [vp, vp, C.c_size_t, C.POINTER(CdcConfig), u64p, u64p,
                                      C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.yams_cdc_chunk_window_host.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(CdcConfig), u64p, u64p,
                                             C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.yams_plugin_get_abi_version.restype = C.c_int
    L.yams_plugin_get_name.restype = C.c_char_p
    L.yams_plugin_get_version.restype = C.c_char_p
    L.yams_plugin_get_manifest_json.restype = C.c_char_p
    L.yams_plugin_init.argtypes = [C.c_char_p, vp]
    L.yams_plugin_init.restype = C.c_int
    L.yams_plugin_shutdown.restype = None
    L.yams_plugin_get_interface.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(vp)]
    L.yams_plugin_get_interface.restype = C.c_int
    L.yams_plugin_get_health_json.argtypes = [C.POINTER(vp)]
    L.yams_plugin_get_health_json.restype = C.c_int
    _lib = L
    return L


class AccelError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"yams_accel status {STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
