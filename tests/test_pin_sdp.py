"""tools/pin_sdp.py: the reference's OWN ``SDP.identifyFeasibleStandardParameters`` (identification/sdp.py:450-604), unmodified, fed once by
its own ``la.qr(YBase)`` and once by the TSQR factor of the path (``estimation.sdp_inputs``), xStd compared at north_star's 1e-6.

* CPU (here): the plumbing, with ``tests/stub_cvxpy.py`` standing in for cvxpy (affine expressions exact, LMIs ignored, least-squares
  minimiser of the Schur residual) and the oracle stand-in engine on both paths -- cvxpy / CLARABEL are not installed in this image.
* GPU: the same through libfbr; with real cvxpy where it is importable, otherwise with the stub (the residual map the reference's function
  builds from the GPU factor against the one it builds from its own QR).  Both need the reference checkout and skip without it.
"""
import os
import sys

import pytest

from common import ROOT

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "identifier.py")), reason="reference checkout not present")


def _tool():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pin_sdp

    return pin_sdp


@needs_ref
def test_reference_sdp_consumes_the_factor_of_the_path(tmp_path, capsys):
    res = _tool().run(REF, "kuka", 600, True, "cpu", str(tmp_path))
    capsys.readouterr()
    assert res["qr_answered_from_factor"] == 1          # la.qr(YBase) inside the reference's function was answered from R_aug
    assert res["num_base_params"] == 43
    assert res["moved_from_apriori"] > 1e-6            # the solve did something (not the a-priori fallback of a failed solver)
    assert res["rel_err_xstd_gpu_vs_cpu"] <= 1e-6, res


def test_tool_reports_what_is_missing(tmp_path):
    """Without cvxpy / a device the tool says so and exits with 3 ("not pinned"), never with a traceback or a false 0."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_sdp.py"), "--reference", str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "missing" in r.stderr and "reference checkout" in r.stderr


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("robot,samples", [("kuka", 2000), ("walkman", 4000)])
def test_reference_sdp_on_the_hip_factor(tmp_path, capsys, robot, samples):
    try:
        import cvxpy  # noqa: F401
        stub = False
    except Exception:
        stub = True
    res = _tool().run(REF, robot, samples, stub, "hip", str(tmp_path))
    capsys.readouterr()
    assert res["gpu_engine"] == "Engine" and res["qr_answered_from_factor"] == 1
    assert res["rel_err_xstd_gpu_vs_cpu"] <= 1e-6, res
