"""Builds the HIP engine in-tree:  hipcc --offload-arch=gfx950 ... -> libtrl_hip.so

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libtrl_hip.so")
SOURCES = ["trl_engine.hip", "trl_mcts.hip", "trl_net.hip"]
HEADERS = ["trl_device.h", "trl_tables.h", "trl_internal.h", os.path.join("..", "..", "include", "tetris_engine.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               # the search mirrors the reference's per-operation rounding: no fused multiply-add, IEEE divide/sqrt
               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the engine is HIP-only and cannot be built without ROCm")
    return exe


def needs_build():
    if not os.path.isfile(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
