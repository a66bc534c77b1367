"""``Data`` container work-alike (reference: identification/data.py:12-161).

Only the container part that feeds the hot path is provided (loading, startOffset, concatenation,
file_boundaries, skipSamples accounting).  Pre-processing and block selection (data.py:148-345,
369-619) are upstream of the path and out of scope."""
from __future__ import annotations

from typing import Any

import numpy as np

from .helpers import Timer


class Data:
    def __init__(self, opt: dict[str, Any]) -> None:
        self.opt = opt
        self.measurements: dict[str, np.ndarray] = {}
        self.num_loaded_samples = 0
        self.num_used_samples = 0
        self.samples: dict[str, np.ndarray] = {}
        self.usedBlocks: list[tuple[Any, ...]] = []
        self.unusedBlocks: list[tuple[Any, ...]] = []
        self.seenBlocks: list[tuple[Any, ...]] = []
        self.file_boundaries: list[int] = [0]
        self.inited = False

    @staticmethod
    def _validate_required_keys(data: dict[str, np.ndarray]) -> None:
        required = {"positions", "velocities", "accelerations", "torques"}
        missing = required - set(data.keys())
        if missing:
            raise KeyError(
                f"Measurement data is missing required key(s): {sorted(missing)}. "
                f"Available keys: {sorted(data.keys())}. "
                f"Make sure you are loading a measurements file, not a trajectory file."
            )

    def init_from_data(self, data: dict[str, np.ndarray]) -> None:
        self.samples = self.measurements = data.copy()
        self._validate_required_keys(data)
        self.num_loaded_samples = self.samples["positions"].shape[0]
        self.num_used_samples = self.num_loaded_samples // (self.opt["skipSamples"] + 1)
        if self.opt.get("verbose"):
            print(f"loaded {self.num_loaded_samples} data samples (using {self.num_used_samples})")
        self.inited = True

    def init_from_files(self, measurements_files) -> None:
        """Load and concatenate measurement npz files (list of lists of paths), dropping the first
        ``startOffset`` samples of each, re-basing ``times`` and recording ``file_boundaries``."""
        with Timer() as t:
            so = self.opt["startOffset"]
            self.file_boundaries = [0]
            for group in measurements_files:
                for fn in group:
                    m = np.load(fn, encoding="latin1", allow_pickle=True)
                    self.file_boundaries.append(self.file_boundaries[-1] + m["positions"].shape[0] - so)
                    for k in m.keys():
                        v = m[k]
                        first = k not in self.measurements
                        if v.ndim == 0:
                            if isinstance(v.item(0), dict):
                                cd = {c: a[so:, :] for c, a in v.item(0).items() if c != "dummy_sim"}
                                if not first and isinstance(self.measurements[k].item(0), dict):
                                    prev = self.measurements[k].item(0)
                                    cd = {c: np.concatenate((prev[c], a), axis=0) if c in prev else a for c, a in cd.items()}
                                self.measurements[k] = np.array(cd)
                            else:
                                self.measurements[k] = v
                        elif v.ndim == 1:
                            if first:
                                self.measurements[k] = v[so:]
                            else:
                                vv = v
                                if k == "times":
                                    vv = v - v[so] + (v[so + 1] - v[so]) + self.measurements[k][-1]
                                self.measurements[k] = np.concatenate((self.measurements[k], vv[so:]), axis=0)
                        else:
                            if first:
                                self.measurements[k] = v[so:, :]
                            else:
                                self.measurements[k] = np.concatenate((self.measurements[k], v[so:, :]), axis=0)
                    m.close()
            self._validate_required_keys(self.measurements)
            self.num_loaded_samples = self.measurements["positions"].shape[0]
            self.num_used_samples = self.num_loaded_samples // (self.opt["skipSamples"] + 1)
            if self.opt.get("verbose"):
                print(f"loaded {self.num_loaded_samples} measurement samples (using {self.num_used_samples})")
            self.samples = {}
            self.block_pos = 0
            if self.opt.get("selectBlocksFromMeasurements"):
                bs = self.opt["blockSize"]
                for k, v in self.measurements.items():
                    if v.ndim == 0:
                        self.samples[k] = v
                    else:
                        self.samples[k] = v[self.block_pos : self.block_pos + bs]
                self.num_selected_samples = self.samples["positions"].shape[0]
                self.num_used_samples = self.num_selected_samples // (self.opt["skipSamples"] + 1)
            else:
                self.samples = self.measurements
        if self.opt.get("showTiming"):
            print(f"(loading samples from file took {t.interval:.3f} sec.)")
        self.inited = True

    def hasMoreSamples(self) -> bool:
        if not self.opt.get("selectBlocksFromMeasurements"):
            return False
        return not (self.block_pos + self.opt["blockSize"] >= self.num_loaded_samples)

    def updateNumSamples(self) -> None:
        self.num_selected_samples = self.samples["positions"].shape[0]
        self.num_used_samples = self.num_selected_samples // (self.opt["skipSamples"] + 1)
