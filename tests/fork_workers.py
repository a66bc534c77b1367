"""Helper of tests/test_gpu_fork.py (run as a script): the reference's worker-pool pattern on the GPU library.

mode "after":  the parent loads the library but does not touch HIP, forks two workers that each build their own Engine
               (the reference builds one Model per worker after the fork, analyticalGradient.py:188-210), then builds its own.
mode "before": the parent builds an Engine first (HIP initialised), then forks: model creation and calls on the inherited
               handle must fail with FBR_E_FORK in the child instead of hanging.
Prints one JSON line."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import load_topo, random_states  # noqa: E402
from flobaroid_amd import _lib  # noqa: E402

TOPO = load_topo("kuka_lwr4")
ST = random_states(TOPO, 500, np.random.default_rng(7), 0)
INHERITED = None


def gram_checksum(_=None):
    eng = _lib.Engine(TOPO)
    G = eng.gram(ST)
    eng.close()
    return float(np.abs(G).sum())


def child_after_parent_init(_=None):
    out = {}
    try:
        _lib.Engine(TOPO)
        out["create"] = "no error"
    except _lib.FbrError as e:
        out["create"] = str(e)
    try:
        INHERITED.gram(ST)
        out["inherited"] = "no error"
    except _lib.FbrError as e:
        out["inherited"] = str(e)
    return out


if __name__ == "__main__":
    mode = sys.argv[1]
    ctx = mp.get_context("fork")
    if mode == "after":
        _lib.load_library()
        with ctx.Pool(2) as pool:
            sums = pool.map(gram_checksum, range(2))
        print(json.dumps({"workers": sums, "parent": gram_checksum()}))
    else:
        INHERITED = _lib.Engine(TOPO)
        ref = float(np.abs(INHERITED.gram(ST)).sum())
        with ctx.Pool(1) as pool:
            res = pool.map(child_after_parent_init, range(1))[0]
        res["parent_still_works"] = float(np.abs(INHERITED.gram(ST)).sum()) == ref
        print(json.dumps(res))
