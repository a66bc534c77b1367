"""Small host helpers on the hot path (work-alikes of identification/helpers.py:89-156, 212-219).

Pure NumPy/SciPy, once-per-run work on S x n arrays; kept on the host as SURVEY.md row A12 says and fed
to the GPU as an input column block.
"""
from __future__ import annotations

import time
from typing import Any

import numpy as np


def _memo(samples: dict[str, np.ndarray], key: str, make) -> np.ndarray:
    """value cached in the samples dict under ``key`` (the reference keeps both series there so that the regressor
    columns, the friction fit and every torque prediction use the same arrays)"""
    if key not in samples:
        samples[key] = make()
    return samples[key]


def getFrictionSignVelocities(samples: dict[str, np.ndarray], opt: dict[str, Any]) -> np.ndarray:
    """Velocities used for the Coulomb sign term (reference: helpers.py:89-132).

    Zero-phase 3rd-order Butterworth low-pass (second-order sections) of ``velocities_raw`` at
    ``frictionVelocityCutoff`` [Hz, default 25] when raw velocities and the sampling frequency are present and the
    cutoff lies below Nyquist; the pipeline velocities otherwise.  Cached as ``velocities_for_sign``."""

    def make():
        fc = float(opt.get("frictionVelocityCutoff", 25.0))
        raw_ok = "velocities_raw" in samples and "frequency" in samples
        fs = float(samples["frequency"]) if raw_ok else 0.0
        if not (raw_ok and fc < fs / 2):
            return samples["velocities"]
        import scipy.signal as sig

        # all joints in one call along the time axis (same per-column arithmetic as filtering joint by joint)
        return sig.sosfiltfilt(sig.butter(3, fc, btype="low", fs=fs, output="sos"), samples["velocities_raw"], axis=0)

    return _memo(samples, "velocities_for_sign", make)


def getFrictionSignSeries(samples: dict[str, np.ndarray], opt: dict[str, Any]) -> np.ndarray:
    """Smoothed Coulomb sign tanh(v_sign / frictionSignThreshold) [default 0.02], cached as ``friction_sign_series``
    (helpers.py:135-156)."""
    return _memo(samples, "friction_sign_series",
                 lambda: np.tanh(getFrictionSignVelocities(samples, opt) / float(opt.get("frictionSignThreshold", 0.02))))


class Timer:
    """``with Timer() as t: ...; t.interval`` (helpers.py:212-219)."""

    def __enter__(self):
        self.start = time.perf_counter()
        return self

    def __exit__(self, *args):
        self.end = time.perf_counter()
        self.interval = self.end - self.start


class Progress:
    def __init__(self, config: dict[str, Any]) -> None:
        self.config = config

    def progress(self, it):
        return it
