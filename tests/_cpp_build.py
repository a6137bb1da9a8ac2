"""Test infrastructure: builds tests/cpp/real_headers_test against the REFERENCE's own headers and
translation units (needs /root/reference; uses the no-op spdlog shim under oracle/shim).  Lives under
tests/ because nothing under yams_amd/ may depend on oracle/ or the reference tree."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def build_real_headers_test(reference: str = "/root/reference"):
    """Compiles tests/cpp/real_headers_test.cpp against the REFERENCE's own headers
    (-DYAMS_ACCEL_USE_HOST_TYPES -I<reference>/include) and links the reference's own translation
    units next to it (sha256_hasher.cpp, rabin_chunker.cpp, streaming_chunker.cpp with the no-op spdlog
    shim of oracle/shim — the recipe of oracle/Makefile).  Only possible where the reference tree
    exists (the dev container); the binary is git-ignored and travels to the GPU box.  Returns the path
    of the executable, or None when neither the sources nor a prebuilt binary are there."""
    root = ROOT
    out_dir = os.path.join(root, "tests", "cpp", "_build")
    exe = os.path.join(out_dir, "real_headers_test")
    src = os.path.join(root, "tests", "cpp", "real_headers_test.cpp")
    if not os.path.isdir(os.path.join(reference, "include", "yams")):
        return exe if os.path.exists(exe) else None
    os.makedirs(out_dir, exist_ok=True)
    deps = [src, os.path.join(root, "include", "yams_mi355x_accel.h")] + \
        [os.path.join(root, "include", "yams_accel", f) for f in os.listdir(os.path.join(root, "include", "yams_accel"))]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    ref_srcs = [os.path.join(reference, "src", "crypto", "sha256_hasher.cpp"),
                os.path.join(reference, "src", "chunking", "rabin_chunker.cpp"),
                os.path.join(reference, "src", "chunking", "streaming_chunker.cpp")]
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-std=c++20", "-O1", "-g", "-rdynamic", "-Wall", "-Wno-unused-variable", "-DYAMS_ACCEL_USE_HOST_TYPES",
           "-I" + os.path.join(root, "oracle", "shim"), "-I" + os.path.join(reference, "include"),
           "-I" + os.path.join(reference, "src", "chunking"), "-I" + os.path.join(root, "include"),
           "-o", exe, src, *ref_srcs, "-lcrypto", "-lpthread", "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("real_headers_test failed to compile against the reference headers:\n" + r.stdout.decode())
    return exe



def build_stub_collective():
    """Compiles tests/stub_coll/stub_coll.cpp — the stream-ordered stand-in for the five RCCL entry points, with which
    the sharded search's kRccl code path runs with several ranks on ONE device (test infrastructure; reached only
    through yams_scan_sharded_options_t.rccl_library).  Host code on the HIP runtime API: plain g++.  The .so is
    git-ignored and travels to the GPU box; returns its path."""
    out_dir = os.path.join(ROOT, "tests", "stub_coll", "_build")
    lib = os.path.join(out_dir, "libyams_stub_coll.so")
    src = os.path.join(ROOT, "tests", "stub_coll", "stub_coll.cpp")
    if os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(src):
        return lib
    if not os.path.isdir("/opt/rocm/include"):
        if os.path.exists(lib):
            return lib
        raise RuntimeError("no ROCm headers and no prebuilt " + lib)
    os.makedirs(out_dir, exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-o", lib, src, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("the stub collective library failed to compile:\n" + r.stdout.decode())
    return lib


def build_l2_calibration_test():
    """Compiles tests/cpp/l2_calibration_test.cpp (plain g++, no GPU, no reference tree) and links it with the oracle's C
    restatement, whose four L2 definitions play the host's distance function.  Returns the executable."""
    import _oracle
    _oracle.build()
    out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "l2_calibration_test")
    src = os.path.join(ROOT, "tests", "cpp", "l2_calibration_test.cpp")
    hdr = os.path.join(ROOT, "include", "yams_accel", "l2_calibration.hpp")
    so = os.path.join(ROOT, "oracle", "_build", "libyams_oracle.so")
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in (src, hdr, so)):
        return exe
    # -O2 -mavx -mfma -ffp-contract=fast on purpose: the reference compiles its sqlite-vec-cpp dependency with '-mavx', '-mfma'
    # (src/vector/meson.build:80-88) — the header's definitions must hold under those flags, and the test's own AVX-shaped
    # host function must come out FUSED
    r = subprocess.run([os.environ.get("CXX", "g++"), "-std=c++20", "-O2", "-mavx", "-mfma", "-ffp-contract=fast", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        "-o", exe, src, so, "-Wl,-rpath," + os.path.dirname(so), "-lm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("l2_calibration_test failed to compile:\n" + r.stdout.decode())
    return exe
