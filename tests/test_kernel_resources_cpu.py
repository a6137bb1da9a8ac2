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


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_direct_row_loads_are_not_touched_while_in_flight():
    """The direct form of the resident-query filter (scan_tiles_i8r_kernel<..., DIRECT>) loads the next slab's row
    fragments from inline asm into registers the compiler owns, and only an s_waitcnt vmcnt(0) at the head of the next
    slab makes them valid.  Nothing the compiler puts in between — a copy at a loop edge, a spill, a reuse — may read or
    write those registers: checked on the assembly in program order (a linear walk; the kernel's loops are the unrolled
    slab pairs, so program order is what matters between a load and its wait)."""
    kernels = {k: v for k, v in _kernels("scan_i8_kernel.hip").items() if "scan_tiles_i8r_kernel" in k and "ELb1ELi" in k}   # <ABL, L2, DIRECT = true, ZSM>
    assert len(kernels) == 3, list(kernels)      # the cosine filter, the L2 filter, the cosine SAMPLE pass

    def regs(tok):
        tok = tok.strip().split()[0] if tok.strip() else ""
        m = re.match(r"v\[(\d+):(\d+)\]$", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return [int(m.group(1))] if m else []
    for name, body in kernels.items():
        inflight, loads = {}, 0
        for line in body:
            t = line.split(";")[0].strip()
            if not t or t.startswith(".") or t.endswith(":"):
                continue
            ops = t.split(None, 1)
            if ops[0] == "global_load_dwordx4":
                for r in regs(ops[1].split(",")[0]):
                    inflight[r] = t
                loads += 1
                continue
            if ops[0] == "s_waitcnt" and "vmcnt(0)" in t:
                inflight.clear()
                continue
            if len(ops) < 2:
                continue
            toks = [x for x in ops[1].split(",")]
            stores = ops[0].startswith(("global_store", "ds_write", "scratch_store", "global_atomic"))
            for tok in (toks if stores else toks[1:]):
                for r in regs(tok):
                    assert r not in inflight, (name, "reads in-flight v%d" % r, t, inflight[r])
            if not stores and not ops[0].startswith("s_"):
                for r in regs(toks[0]):
                    assert r not in inflight, (name, "writes in-flight v%d" % r, t, inflight[r])
        assert loads >= 12, (name, loads)       # prologue + two slabs of the loop body, four fragments each


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_deep_form_row_loads_are_not_touched_while_in_flight():
    """scan_tiles_i8d_kernel keeps TWO slabs of row fragments in flight per wave and waits for them with COUNTED waits
    (`s_waitcnt vmcnt(4)`: the four loads of the slab one ahead may stay out).  Its vector-memory operations retire in
    order, so a walk over the assembly with a FIFO of outstanding operations knows, at every instruction, which registers
    still have a load due: nothing may read or write them — not a copy, not a spill, not a temporary the allocator parks
    there (round 6: the flush after the last strip did exactly that and stored through an overwritten address; a bus error
    on shards of 4M rows and more).  The kernel must not use scratch at all."""
    kernels = {k: v for k, v in _kernels("scan_i8_kernel.hip").items() if "scan_tiles_i8d_kernel" in k}
    assert len(kernels) == 1, list(kernels)

    def regs(tok):
        tok = tok.strip().split()[0] if tok.strip() else ""
        m = re.match(r"v\[(\d+):(\d+)\]$", tok)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)$", tok)
        return [int(m.group(1))] if m else []
    for name, body in kernels.items():
        assert not any("scratch_" in l for l in body), name
        fifo, loads, counted = [], 0, 0
        for line in body:
            t = line.split(";")[0].strip()
            if not t or t.startswith(".") or t.endswith(":"):
                continue
            ops = t.split(None, 1)
            op = ops[0]
            toks = ops[1].split(",") if len(ops) > 1 else []
            busy = {r: txt for rs, txt in fifo for r in rs}
            vm_load = op.startswith(("global_load", "global_atomic")) and not op.startswith("global_load_lds")
            vm_other = op.startswith(("global_store", "global_load_lds", "buffer_", "flat_"))
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    counted += keep == 4
                    while len(fifo) > keep:
                        fifo.pop(0)
                continue
            sources = toks if (vm_other or op.startswith(("ds_write", "global_atomic"))) else toks[1:]
            for tok in sources:
                for r in regs(tok):
                    assert r not in busy, (name, "reads in-flight v%d" % r, t, busy[r])
            if vm_load:
                dst = regs(toks[0]) if not op.startswith("global_atomic") or len(toks) >= 4 else []
                for r in dst:
                    assert r not in busy, (name, "loads into in-flight v%d" % r, t, busy[r])
                fifo.append((dst, t)); loads += op == "global_load_dwordx4"
                continue
            if vm_other:
                fifo.append(([], t))
                continue
            if toks and not op.startswith("s_"):
                for r in regs(toks[0]):
                    assert r not in busy, (name, "writes in-flight v%d" % r, t, busy[r])
        assert loads >= 8 + 12 and counted >= 4, (name, loads, counted)    # pipeline fill + three unrolled slabs; the counted waits are there
