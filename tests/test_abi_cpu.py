"""CPU suite part 2: the C-ABI library loads, exports every symbol include/*.h declares, and the
YAMS plugin surface behaves like the reference's plugins (include/yams/plugins/abi.h:18-34;
tools/fuzzing/fuzz_abi_test_plugin.c:48-60; plugins/glint/plugin.cpp:338-359).
No compute call is made here: without a GPU the product path must refuse, not fall back."""
import ctypes as C
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "yams_mi355x_accel.h")).read()
    return set(re.findall(r"YAMS_ACCEL_API\s+[\w\s\*]+?\b(yams_\w+)\s*\(", src))


def test_every_declared_symbol_is_exported(accel_lib):
    from yams_amd import _lib
    decl = declared_functions()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(accel_lib, name), f"{name} declared in the header but not exported"
    # and the Python binding knows every declared symbol
    assert decl <= set(_lib.EXPORTS)
    for name in ["yams_plugin_get_abi_version", "yams_plugin_get_name", "yams_plugin_get_version",
                 "yams_plugin_get_manifest_json", "yams_plugin_init", "yams_plugin_shutdown",
                 "yams_plugin_get_interface", "yams_plugin_get_health_json"]:  # abi.h:26-34
        assert hasattr(accel_lib, name)


def test_library_has_no_oracle_or_torch_dependency():
    import subprocess
    from yams_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any("amdhip64" in n for n in needed)
    for n in needed:
        assert "oracle" not in n and "torch" not in n and "yams_ref" not in n and "crypto" not in n, needed
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in syms and "ref_sha256" not in syms


def test_product_library_reads_no_environment():
    """Kernel-form / ablation selection from the environment exists only in the measurement build
    (-DYAMS_ACCEL_MEASURE); the product library does not even import getenv."""
    import subprocess
    from yams_amd import _lib
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in und
    for name in os.listdir(os.path.join(ROOT, "yams_amd", "csrc")):
        src = open(os.path.join(ROOT, "yams_amd", "csrc", name)).read()
        for m in re.finditer(r"getenv", src):
            guard = src.rfind("#ifdef YAMS_ACCEL_MEASURE", 0, m.start())
            assert guard >= 0 and src.find("#endif", guard, m.start()) < 0, f"unguarded getenv in {name}"


def test_plugin_identity_and_manifest(accel_lib):
    L = accel_lib
    assert L.yams_plugin_get_abi_version() == 1  # YAMS_PLUGIN_ABI_VERSION, abi.h:18
    assert L.yams_plugin_get_name() == b"yams_mi355x_accel"
    assert re.match(rb"\d+\.\d+\.\d+", L.yams_plugin_get_version())
    m = json.loads(L.yams_plugin_get_manifest_json())
    # docs/spec/schemas/manifest.schema.json (reference): name, version, abi, interfaces[{id,version}]
    assert m["name"] == "yams_mi355x_accel" and m["abi"] == 1 and isinstance(m["version"], str)
    ids = {(i["id"], i["version"]) for i in m["interfaces"]}
    assert ids == {("vector_scan_v1", 1), ("content_hash_v1", 1), ("chunker_v1", 3)}


def test_get_interface_contract(accel_lib):
    from yams_amd import _lib
    L = accel_lib
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(None, 1, C.byref(p)) == -4          # ERR_INVALID
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, None) == -4
    assert L.yams_plugin_get_interface(b"nope_v1", 1, C.byref(p)) == -2    # ERR_NOT_FOUND
    assert p.value is None
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 3, C.byref(p)) == -2
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 0, C.byref(p)) == -2
    p1 = C.c_void_p()
    assert L.yams_plugin_get_interface(b"vector_scan_v1", 1, C.byref(p1)) == 0      # version 1 hosts get the same table
    for name, typ, ver in [(b"vector_scan_v1", _lib.VectorScanV1, 2),               # (2 appended pq_index_set / search_pq)
                           (b"content_hash_v1", _lib.ContentHashV1, 1),
                           (b"chunker_v1", _lib.ChunkerV1, 3)]:
        assert L.yams_plugin_get_interface(name, ver, C.byref(p)) == 0
        vt = C.cast(p, C.POINTER(typ)).contents
        assert vt.abi_version == ver  # first field, model_provider_v1.h:50-51 convention
        for fname, _ in typ._fields_[2:]:
            assert getattr(vt, fname), f"{name}.{fname} is NULL"


def test_chunker_v1_serves_every_version_up_to_its_own(accel_lib):
    """One .so accepts any version in [1, max] (plugins/onnx/onnx_plugin.cpp:188-210): a host that asks for chunker_v1
    version 1 gets the same table (chunk_many / free_chunk_batch are appended fields it does not look at)."""
    from yams_amd import _lib
    L = accel_lib
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.yams_plugin_get_interface(b"chunker_v1", 1, C.byref(p1)) == 0
    assert L.yams_plugin_get_interface(b"chunker_v1", 2, C.byref(p2)) == 0 and p1.value == p2.value
    assert L.yams_plugin_get_interface(b"chunker_v1", 3, C.byref(p3)) == 0 and p1.value == p3.value
    assert L.yams_plugin_get_interface(b"chunker_v1", 4, C.byref(p3)) == -2


def test_health_json_is_malloced_json(accel_lib):
    L = accel_lib
    p = C.c_void_p()
    assert L.yams_plugin_get_health_json(None) == -4
    assert L.yams_plugin_get_health_json(C.byref(p)) == 0
    h = json.loads(C.string_at(p))
    assert "status" in h
    C.CDLL(None).free(p)  # the host free()s it (abi_plugin_loader.cpp:481-500)


def test_no_gpu_means_refusal_not_fallback(accel_lib):
    """Without a HIP device every compute door refuses loudly (UNSUPPORTED / INIT_FAILED)."""
    from yams_amd import _lib
    L = accel_lib
    if L.yams_accel_device_count() > 0:
        pytest.skip("a GPU is visible here; the refusal path is exercised on CPU-only hosts")
    ctx = C.c_void_p()
    assert L.yams_accel_ctx_create(0, None, C.byref(ctx)) == _lib.YAMS_ERR_UNSUPPORTED
    assert ctx.value is None
    assert L.yams_plugin_init(b"{}", None) == -3  # YAMS_PLUGIN_ERR_INIT_FAILED
    p = C.c_void_p()
    assert L.yams_plugin_get_interface(b"content_hash_v1", 1, C.byref(p)) == 0
    vt = C.cast(p, C.POINTER(_lib.ContentHashV1)).contents
    out = C.create_string_buffer(65)
    assert vt.hash(None, None, 0, out) == _lib.YAMS_ERR_UNSUPPORTED
    assert L.yams_plugin_get_interface(b"chunker_v1", 1, C.byref(p)) == 0
    ck = C.cast(p, C.POINTER(_lib.ChunkerV1)).contents
    cfg = _lib.CdcConfig()
    assert ck.get_default_config(None, _lib.CDC_STREAMING, C.byref(cfg)) == 0  # pure metadata
    assert (cfg.window_size, cfg.min_size, cfg.max_size, cfg.mask) == (48, 16384, 1048576, 0x1FFF)
    assert cfg.polynomial == 0x3DA3358B4DC173
    from yams_amd.accel import Accel
    with pytest.raises(_lib.AccelError):
        Accel(0)
    L.yams_plugin_shutdown()


def test_plugin_configuration_is_read_strictly(accel_lib):
    """yams_plugin_init's config_json goes through a tokenizer, not substring search: a known key
    with a value of the wrong type, an enumerated value the plugin does not list (the parity-critical "l2_accumulate" above
    all: a typo must not silently serve fp64 results) or text that is not a JSON object fail the init with a message in
    the health JSON; unknown keys — whatever they hold — are ignored.  Runs without a GPU: a well-formed configuration
    gets as far as the device check."""
    import json
    L = accel_lib
    if L.yams_accel_device_count() > 0:
        pytest.skip("a GPU is visible here: the device-independent half of init is what this test looks at")

    def init_error(cfg):
        L.yams_plugin_shutdown()
        assert L.yams_plugin_init(cfg, None) == -3
        p = C.c_void_p()
        assert L.yams_plugin_get_health_json(C.byref(p)) == 0
        try:
            return json.loads(C.string_at(p).decode())["error"]
        finally:
            L.yams_accel_free_string(p)

    for good in (b"{}", b"", None, b'{"device": 0, "search_slots": 4, "shadows": "i8", "i8_layout": "rotated", "l2_accumulate": "f32x8_fma", "collective": "peer", "fence": "off"}',
                 b'{"devices": [0, 1], "stripe_rows": 4096, "note": "both", "nested": {"shadows": ["none", {"x": 1}]}, "ratio": 0.5, "on": true}',
                 b' { "rccl_library" : "/opt/site/lib\\"rccl\\".so" , "exchange_timeout_ms" : 1500 } '):
        assert init_error(good) == "no gfx950 device visible", good
    bad = {b'{"l2_accumulate": "f32x8_fmadd"}': "l2_accumulate", b'{"l2_accumulate": 8}': "must be a string",
           b'{"shadows": "all"}': "shadows", b'{"i8_layout": "rotate"}': "i8_layout", b'{"search_slots": "4"}': "must be an integer", b'{"devices": "0,1"}': "devices",
           b'{"devices": []}': "devices", b'{"collective": "nccl"}': "collective", b'{"device": 0': "expected", b'["device", 0]': "not a JSON object",
           b'{"device": 0} trailing': "text after", b'{"stripe_rows": 1.5}': "must be an integer"}
    for cfg, needle in bad.items():
        e = init_error(cfg)
        assert e.startswith("configuration:") and needle in e, (cfg, e)
    # round 5's reader took the FIRST "both" after "shadows" anywhere in the text: {"shadows":"none","note":"both"} enabled both
    assert init_error(b'{"shadows": "none", "note": "both"}') == "no gfx950 device visible"
    L.yams_plugin_shutdown()


def test_default_config_matches_reference_constants(accel_lib):
    from yams_amd import _lib
    cfg = _lib.CdcConfig()
    accel_lib.yams_cdc_default_config(C.byref(cfg), _lib.CDC_RABIN)
    # chunker.h:44-51, core/types.h:280-285 (reference)
    assert (cfg.window_size, cfg.min_size, cfg.max_size, cfg.polynomial, cfg.mask, cfg.mode) == \
        (48, 16 * 1024, 1024 * 1024, 0x3DA3358B4DC173, 0x1FFF, 0)


def test_product_sources_never_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    bad = []
    for base in ("yams_amd", "include", "scripts"):   # measurement scripts time the product, never the checker
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(?<![\w/])oracle(/|_)|libyams_oracle|libyams_ref|_oracle\b", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_integration_index_names_every_exported_entry_point():
    """INTEGRATION.md section 8 lists the flat C ABI entry point by entry point (with the reference interface each stands
    in for); a symbol added to the library has to appear there.  The table abbreviates families as
    `yams_x_create / _destroy`: a `_suffix` replaces trailing components of the name in front of it."""
    from yams_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 8. Index of the flat C ABI"):]
    assert set(_lib.EXPORTS) == declared_functions()      # (the list the index is held against is the header's)
    names = set()
    for cell in re.findall(r"`([^`]+)`", sec):
        base = None
        for part in (p.strip() for p in re.split(r"\s*/\s*|,\s*", cell)):
            if part.startswith("yams_"):
                names.add(part); base = part
            elif part.startswith("_") and base:
                toks = base.split("_")
                for cut in range(len(toks) - 1, 1, -1):
                    cand = "_".join(toks[:cut]) + part
                    if cand in _lib.EXPORTS:
                        names.add(cand); break
    missing = [e for e in _lib.EXPORTS if e not in names]
    assert not missing, missing
