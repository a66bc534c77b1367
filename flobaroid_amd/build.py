"""Build libfbr.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit, compiled in parallel and only when its inputs changed: the inputs of every object (the unit itself,
every header of csrc/ and include/fbr.h, the compiler flags) are hashed and the hash is kept beside the object, so that a library that
does not match the sources in the tree is never mistaken for an up-to-date one (time stamps do not survive a checkout or a snapshot)."""
from __future__ import annotations

import concurrent.futures
import hashlib
import json
import os
import shutil
import subprocess
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libfbr.so")
OBJDIR = os.path.join(_HERE, "csrc", "_obj")
STAMP = os.path.join(_HERE, "libfbr.build.json")
SOURCES = ["fbr_api.hip", "fbr_gram_api.hip", "fbr_tsqr_api.hip", "fbr_signal_api.hip"]
# -Wno-inline-asm: the hand-written LDS-DMA of the TSQR kernels (csrc/fbr_tsqr.h fbr_dma16) writes M0 and says so in its clobber list;
# clang warns about every instantiation that a reserved register is named there (it is still honoured)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm"]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libfbr.so)")


def _headers() -> list[str]:
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs + [os.path.join(_HERE, "..", "include", "fbr.h")]


def source_hash() -> str:
    """Hash of everything libfbr.so is built from (sources, headers, flags)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [os.path.join(CSRC, s) for s in SOURCES] + _headers():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _lib_digest() -> str | None:
    try:
        with open(OUT, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()
    except OSError:
        return None


def built_hash() -> str | None:
    """Source hash the library in the tree was built from -- None when the stamp does not belong to THIS binary (the stamp carries the
    binary's own digest: a stamp restored by a checkout beside a library built from other sources must not pass for up to date)."""
    try:
        with open(STAMP) as fh:
            st = json.load(fh)
    except (OSError, ValueError):
        return None
    if st.get("lib_sha256") != _lib_digest():
        return None
    return st.get("source_hash")


def needs_build() -> bool:
    return not os.path.exists(OUT) or built_hash() != source_hash()


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile (if the sources changed, or force) and link libfbr.so; returns its path.  ``last_build`` says what happened."""
    global last_build
    want = source_hash()
    if not force and os.path.exists(OUT) and built_hash() == want:
        last_build = {"compiled": False, "source_hash": want, "reason": "libfbr.so matches the sources in the tree"}
        return OUT
    hipcc = hipcc_path()
    os.makedirs(OBJDIR, exist_ok=True)
    t0 = time.time()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    last_build = {"compiled": True, "source_hash": want, "seconds": round(time.time() - t0, 1), "units": len(SOURCES)}
    with open(STAMP, "w") as fh:
        json.dump(dict(last_build, lib_sha256=_lib_digest()), fh)
    return OUT


last_build: dict = {}

if __name__ == "__main__":
    print(build_lib(force=True, verbose=True), last_build)
