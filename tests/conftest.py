import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The library takes the column reductions only where they pay -- not for a few hundred samples of a small robot.  The
    # tests run at such sizes and are there to check the reduced paths: they are taken whatever the size
    # (tests/test_gpu_linkmerge.py::test_small_batches_skip_the_reductions looks at the default).
    os.environ.setdefault("FBR_REDUCE_ALWAYS", "1")
