"""Builds libyams_mi355x_accel.so for gfx950 with hipcc (in-tree, so it travels to the GPU box).

    python -m yams_amd.build [--force]

One hipcc invocation per translation unit (cross-compiles without a GPU), then one link.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libyams_mi355x_accel.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["scan_kernels.hip", "scan_small_kernel.hip", "scan_bf16_kernel.hip", "scan_i8_kernel.hip", "pq_kernels.hip", "ingest_kernels.hip", "dedup_kernels.hip", "accel_ctx.cpp",
           "scan_api.cpp", "pq_api.cpp", "sharded_api.cpp", "dedup_api.cpp",
           "ingest_api.cpp", "plugin.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _stamp() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for name in sorted(os.listdir(root)):
            p = os.path.join(root, name)
            if os.path.isfile(p):
                h.update(name.encode())
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


MEASURE_LIB = os.path.join(LIBDIR, "libyams_mi355x_accel_measure.so")


def build(force: bool = False, verbose: bool = False, measure: bool = False) -> str:
    """measure=True builds libyams_mi355x_accel_measure.so (-DYAMS_ACCEL_MEASURE): the same sources
    plus the ablation kernels and the YAMS_ACCEL_BF16_KERNEL / _PASSES environment knobs that
    scripts/ uses.  The product library contains neither and never reads the environment."""
    os.makedirs(LIBDIR, exist_ok=True)
    LIB = MEASURE_LIB if measure else globals()["LIB"]
    stamp_file = os.path.join(LIBDIR, ".stamp_measure" if measure else ".stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) \
            and open(stamp_file).read() == stamp:
        return LIB
    if not os.path.exists(HIPCC):
        if os.path.exists(LIB):
            return LIB  # GPU box without a toolchain mismatch: use what travelled with the snapshot
        raise RuntimeError(f"hipcc not found at {HIPCC} and no prebuilt {LIB}")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src + (".measure.o" if measure else ".o"))
        cmd = [HIPCC, *FLAGS, *(["-DYAMS_ACCEL_MEASURE"] if measure else []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
            "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-ldl", "-lpthread"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, measure="--measure" in sys.argv))


def build_host_tests() -> str:
    """Compiles the C++ host-side mirror test (plain g++, dlopens the plugin at run time)."""
    root = os.path.dirname(HERE)
    out_dir = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "host_mirror_test")
    src = os.path.join(root, "tests", "cpp", "host_mirror_test.cpp")
    deps = [src] + [os.path.join(root, "include", "yams_accel", f)
                    for f in os.listdir(os.path.join(root, "include", "yams_accel"))]
    deps.append(os.path.join(root, "include", "yams_mi355x_accel.h"))
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    cxx = os.environ.get("CXX", "g++")
    r = subprocess.run([cxx, "-std=c++20", "-O1", "-Wall", "-I" + os.path.join(root, "include"),
                        "-o", exe, src, "-ldl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("host mirror test failed to compile:\n" + r.stdout.decode())
    return exe

