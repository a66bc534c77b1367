import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The library takes the column reductions only where they pay -- not for a few hundred samples of a small robot, which is what
# the tests run at.  The modules that compare whole passes with the oracle / the reference's outputs (parity, model, endtoend, dropin_fb)
# therefore run every test three times (``pytestmark = pytest.mark.usefixtures("reduction_mode")``):
#   default   the product's own choice per call (what a user gets: all columns at these sizes),
#   reduced   the reductions forced whatever the size (option reduce_min_work = 0, row groups of the TSQR from 1 sample on where the
#             test asks for them),
#   allcols   the reductions switched off (option link_merge = 0).
# Options go through the C-ABI (fbr_model_set_option) as defaults of every Engine created during the test; nothing reads the environment.
REDUCTION_MODES = {"default": {}, "reduced": {"reduce_min_work": 0.0, "tsqr_group_min_samples": 1.0}, "allcols": {"link_merge": 0.0}}


@pytest.fixture(params=list(REDUCTION_MODES))
def reduction_mode(request):
    from flobaroid_amd import _lib

    saved = dict(_lib.DEFAULT_OPTIONS)
    _lib.DEFAULT_OPTIONS.update(REDUCTION_MODES[request.param])
    try:
        yield request.param
    finally:
        _lib.DEFAULT_OPTIONS.clear()
        _lib.DEFAULT_OPTIONS.update(saved)


@pytest.fixture
def engine_options():
    """Context manager: defaults of every Engine created inside it (``with engine_options(tsqr_groups=0): ...``)."""
    import contextlib

    from flobaroid_amd import _lib

    @contextlib.contextmanager
    def scope(**opts):
        saved = dict(_lib.DEFAULT_OPTIONS)
        _lib.DEFAULT_OPTIONS.update({k: float(v) for k, v in opts.items()})
        try:
            yield
        finally:
            _lib.DEFAULT_OPTIONS.clear()
            _lib.DEFAULT_OPTIONS.update(saved)

    return scope
