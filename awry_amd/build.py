"""Builds awry_amd/lib/libawry_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m awry_amd.build [--force]

The kernels + C ABI (csrc/awry_hip.hip) are compiled as HIP for --offload-arch=gfx950; the pure host
code (csrc/host_index.cpp) with the host compiler at -march=x86-64-v3, because the .so built here also
runs on the GPU box's host CPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libawry_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
SOURCES = ["awry_hip.hip", "sa_builder.hip", "host_index.cpp", "host_pack.cpp"]
HEADERS = ["kernels.hip.h", "layout.h", "alphabet.h", "host_index.h", "host_pack.h", "sais.hpp", os.path.join("..", "..", "include", "awry_hip.h")]


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return SO
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: the HIP extension cannot be built" % HIPCC)
    os.makedirs(LIBDIR, exist_ok=True)
    obj = os.path.join(LIBDIR, "_obj")
    os.makedirs(obj, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    cmds = [
        [HIPCC, "--offload-arch=gfx950", "-march=x86-64-v3"] + common + ["-c", os.path.join(CSRC, "awry_hip.hip"), "-o", os.path.join(obj, "awry_hip.o")],
        [HIPCC, "--offload-arch=gfx950", "-march=x86-64-v3"] + common + ["-c", os.path.join(CSRC, "sa_builder.hip"), "-o", os.path.join(obj, "sa_builder.o")],
        ["g++", "-march=x86-64-v3"] + common + ["-c", os.path.join(CSRC, "host_index.cpp"), "-o", os.path.join(obj, "host_index.o")],
        ["g++", "-march=x86-64-v3"] + common + ["-c", os.path.join(CSRC, "host_pack.cpp"), "-o", os.path.join(obj, "host_pack.o")],
        [HIPCC, "--offload-arch=gfx950", "-shared", "-o", SO, os.path.join(obj, "awry_hip.o"), os.path.join(obj, "sa_builder.o"), os.path.join(obj, "host_index.o"),
         os.path.join(obj, "host_pack.o"),
         "-lpthread", "-Wl,-rpath," + os.path.join(ROCM, "lib")],
    ]
    for c in cmds:
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
