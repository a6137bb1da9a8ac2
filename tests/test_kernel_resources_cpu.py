"""The filter kernels track their LDS-DMA pieces with hand-counted `s_waitcnt vmcnt(N)`: a register spill inside
their k loops would put scratch loads / stores (VMEM operations the count does not know about) between the pieces
and the waits.  So "no scratch traffic between the first and the last MFMA" is a correctness property of these
kernels, checked here on the gfx950 assembly (hipcc cross-compiles without a GPU); the int8 kernels — the bench's
dominant kernel family — must not use scratch at all."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _kernels(src):
    """{mangled kernel name: [assembly lines]} of the product build of one translation unit."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
               "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "yams_amd", "csrc"),
               os.path.join(ROOT, "yams_amd", "csrc", src), "-o", out]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        text = open(out).read().splitlines()
    kernels, cur = {}, None
    for line in text:
        m = re.match(r"^(_ZN10yams_accel\w+):", line)
        if m:
            cur = m.group(1); kernels[cur] = []
        elif cur is not None:
            kernels[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    return kernels


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("src", ["scan_i8_kernel.hip", "scan_bf16_kernel.hip"])
def test_no_scratch_traffic_inside_the_mfma_loops(src):
    tiles = {k: v for k, v in _kernels(src).items() if "scan_tiles" in k}
    assert tiles
    for name, body in tiles.items():
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        assert mf, name
        inside = [l.strip() for l in body[mf[0]:mf[-1] + 1] if "scratch_" in l]
        assert not inside, (name, inside[:3])
        if src == "scan_i8_kernel.hip":
            assert not any("scratch_" in l for l in body), name
    if src == "scan_i8_kernel.hip":     # both metrics of both kernel forms are there
        for frag in ("scan_tiles_i8r_kernelILi0ELb0ELb0", "scan_tiles_i8r_kernelILi0ELb1ELb0", "scan_tiles_i8r_kernelILi0ELb0ELb1",
                     "scan_tiles_i8r_kernelILi0ELb1ELb1", "scan_tiles_i8h_kernelILi1ELi0ELi0",
                     "scan_tiles_i8h_kernelILi1ELi0ELi1", "scan_tiles_i8h_kernelILi0ELi0ELi1"):
            assert any(frag in k for k in tiles), frag
