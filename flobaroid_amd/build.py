"""Build libfbr.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libfbr.so")
SOURCES = ["fbr_api.hip"]
HEADERS = ["fbr_kernels.h", "fbr_math.h", "fbr_program.h", "fbr_tsqr.h", "fbr_signal.h", "fbr_reduce.h", os.path.join("..", "..", "include", "fbr.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libfbr.so)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    # -Wno-inline-asm: the hand-written LDS-DMA of the TSQR kernels (csrc/fbr_tsqr.h fbr_dma16) writes M0 and says so in its clobber
    # list; clang warns about every instantiation that a reserved register is named there (it is still honoured)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-inline-asm", "-o", OUT] + [
        os.path.join(CSRC, s) for s in SOURCES
    ]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
