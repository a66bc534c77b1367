"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/fbr.h declares;
compute calls fail loudly without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from common import ROOT, load_topo


def _declared():
    src = open(os.path.join(ROOT, "include", "fbr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fbr_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol():
    from flobaroid_amd import _lib

    lib = _lib.load_library()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libfbr.so does not export {n}"
    assert set(_lib._SIGNATURES) == set(names)
    assert lib.fbr_version() >= 100


def test_fails_loudly_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from flobaroid_amd._lib import Engine, FbrError, device_count

    assert device_count() == 0
    with pytest.raises(FbrError, match="no HIP device"):
        Engine(load_topo("threeLinks"))


def test_product_does_not_import_the_oracle():
    """The shipped package must never import, load or call anything under oracle/."""
    pkg = os.path.join(ROOT, "flobaroid_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|fbr_oracle|oracle[./]|orc_[a-z_]+\()")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), (f, "references the oracle")
