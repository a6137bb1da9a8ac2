"""The C++ host-side mirror (include/yams_accel/*.hpp: Plugin loader, IContentHasher, IChunker and
the yams::vector query API over the plugin vtables) — compiled with g++, run as its own process
(it dlopens the plugin exactly as AbiPluginLoader would)."""
import subprocess

import pytest


def _exe():
    from yams_amd import build as b
    b.build()
    return b.build_host_tests(), b.LIB


def test_host_mirror_compiles_and_refuses_without_gpu(accel_lib):
    exe, lib = _exe()
    if accel_lib.yams_accel_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked run")
    r = subprocess.run([exe, lib, "--expect-no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_host_mirror_reference_style_suite():
    exe, lib = _exe()
    r = subprocess.run([exe, lib], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK (0 failures)" in r.stdout
