"""Process model of the drop-in boundary: the reference's multi-process users build one Model per worker after
fork (analyticalGradient.py:188-210), so the library initialises HIP lazily and per process; a handle or an initialised HIP runtime
inherited through fork() is reported (FBR_E_FORK), not used."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fork_workers.py")


def _run(mode):
    p = subprocess.run([sys.executable, HELPER, mode], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_one_engine_per_forked_worker():
    out = _run("after")
    assert out["workers"][0] == out["workers"][1] == out["parent"] and out["parent"] > 0.0


def test_fork_after_hip_init_is_reported():
    out = _run("before")
    assert "code -5" in out["create"] and "fork" in out["create"]
    assert "code -5" in out["inherited"] and "another process" in out["inherited"]
    assert out["parent_still_works"] is True
