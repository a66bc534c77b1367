"""Small host helpers on the hot path (work-alikes of identification/helpers.py:89-156, 212-219).

Pure NumPy/SciPy, once-per-run work on S x n arrays; kept on the host as SURVEY.md row A12 says and fed
to the GPU as an input column block.
"""
from __future__ import annotations

import time
from typing import Any

import numpy as np


def getFrictionSignVelocities(samples: dict[str, np.ndarray], opt: dict[str, Any]) -> np.ndarray:
    """Velocities used for the Coulomb sign term (reference: helpers.py:89-132).

    Zero-phase 3rd-order Butterworth low-pass of ``velocities_raw`` at ``frictionVelocityCutoff`` when raw
    velocities and the sampling frequency exist and the cutoff is below Nyquist, else the pipeline
    velocities.  Cached in the samples dict under ``velocities_for_sign``."""
    if "velocities_for_sign" in samples:
        return samples["velocities_for_sign"]
    cutoff = float(opt.get("frictionVelocityCutoff", 25.0))
    has_raw = "velocities_raw" in samples and "frequency" in samples
    freq = float(samples["frequency"]) if has_raw else 0.0
    if has_raw and cutoff < freq / 2:
        import scipy.signal

        sos = scipy.signal.butter(3, cutoff, btype="low", fs=freq, output="sos")
        raw = samples["velocities_raw"]
        vfs = np.column_stack([scipy.signal.sosfiltfilt(sos, raw[:, j]) for j in range(raw.shape[1])])
    else:
        vfs = samples["velocities"]
    samples["velocities_for_sign"] = vfs
    return vfs


def getFrictionSignSeries(samples: dict[str, np.ndarray], opt: dict[str, Any]) -> np.ndarray:
    """tanh(v_sign / frictionSignThreshold), cached under ``friction_sign_series`` (helpers.py:135-156)."""
    if "friction_sign_series" in samples:
        return samples["friction_sign_series"]
    v = getFrictionSignVelocities(samples, opt)
    thr = float(opt.get("frictionSignThreshold", 0.02))
    s = np.tanh(v / thr)
    samples["friction_sign_series"] = s
    return s


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
