"""``Data`` container work-alike (reference: identification/data.py:12-161).

The container part that feeds the hot path (loading, startOffset, concatenation, file_boundaries, skipSamples
accounting) and the step before the path, ``preprocess`` (data.py:369-619): zero-phase
Butterworth low-passes, median filters and 4th-order central differences that turn raw logs into q, dq, ddq, tau.
Block selection (data.py:148-345: iterate the measurement blocks, keep the ones whose base regressor is best conditioned,
drop near-duplicates, re-assemble) works from the path's small reductions: the condition numbers come from the triangular
factor / the per-block Grams of the GPU pass (``getBlockStats``, ``getAllBlockStats``), never from a materialised YBase."""
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
        """Load and concatenate measurement npz files (list of groups of paths; reference: data.py:55-146).

        Every file drops its first ``startOffset`` samples.  Per key: 2-D arrays are stacked along the sample axis, 1-D
        arrays appended, scalars (e.g. ``frequency``) keep the last file's value; ``times`` of a later file continue the
        running clock: t - t[so] + (t[so+1] - t[so]) + last time so far.  ``file_boundaries`` records the sample index
        at which every file starts.  Contact dictionaries are concatenated per frame, zero-padded for the files that do not
        carry a frame, so every frame has one row per loaded sample (the reference keeps the last file's dict only)."""
        with Timer() as timer:
            so = self.opt["startOffset"]
            bounds = [0]
            merged: dict[str, Any] = self.measurements

            def append(key: str, arr: np.ndarray) -> None:
                merged[key] = arr if key not in merged else np.concatenate((merged[key], arr), axis=0)

            for path in (p for group in measurements_files for p in group):
                with np.load(path, encoding="latin1", allow_pickle=True) as f:
                    bounds.append(bounds[-1] + f["positions"].shape[0] - so)
                    for key in f.keys():
                        v = f[key]
                        if v.ndim >= 1:
                            if key == "times" and key in merged:
                                v = v - v[so] + (v[so + 1] - v[so]) + merged[key][-1]
                            append(key, v[so:])
                        elif isinstance(v.item(0), dict):  # contact wrenches: {frame: (S, 6)}
                            # (the reference keeps only the LAST file's dict, data.py:55-146; here every frame gets one row per
                            # loaded sample: files that do not know a frame contribute zero wrenches for their samples)
                            nrows = f["positions"].shape[0] - so
                            before = bounds[-2]  # samples merged before this file
                            frames = {c: w[so:, :] for c, w in v.item(0).items() if c != "dummy_sim"}
                            old = merged[key].item(0) if key in merged and isinstance(merged[key].item(0), dict) else {}
                            out = {}
                            for c in list(old.keys()) + [c for c in frames if c not in old]:
                                prev = old[c] if c in old else np.zeros((before, 6))
                                if prev.shape[0] < before:  # frame absent from intermediate files
                                    prev = np.concatenate((prev, np.zeros((before - prev.shape[0], 6))), axis=0)
                                out[c] = np.concatenate((prev, frames[c] if c in frames else np.zeros((nrows, 6))), axis=0)
                            merged[key] = np.array(out)
                        else:
                            merged[key] = v
            self.file_boundaries = bounds
            for key, v in merged.items():  # frames missing from the trailing files: zero wrenches up to the last sample
                if isinstance(v, np.ndarray) and v.ndim == 0 and isinstance(v.item(0), dict):
                    d = v.item(0)
                    for c, w in d.items():
                        if w.shape[0] < bounds[-1]:
                            d[c] = np.concatenate((w, np.zeros((bounds[-1] - w.shape[0], 6))), axis=0)
            self._validate_required_keys(merged)
            self.num_loaded_samples = merged["positions"].shape[0]
            self.num_used_samples = self.num_loaded_samples // (self.opt["skipSamples"] + 1)
            if self.opt.get("verbose"):
                print(f"loaded {self.num_loaded_samples} measurement samples (using {self.num_used_samples})")
            self.block_pos = 0
            if self.opt.get("selectBlocksFromMeasurements"):
                bs = self.opt["blockSize"]
                self.samples = {k: (v if v.ndim == 0 else v[self.block_pos:self.block_pos + bs]) for k, v in merged.items()}
                self.updateNumSamples()
            else:
                self.samples = merged
        if self.opt.get("showTiming"):
            print(f"(loading samples from file took {timer.interval:.3f} sec.)")
        self.inited = True

    def hasMoreSamples(self) -> bool:
        if not self.opt.get("selectBlocksFromMeasurements"):
            return False
        return not (self.block_pos + self.opt["blockSize"] >= self.num_loaded_samples)

    def updateNumSamples(self) -> None:
        self.num_selected_samples = self.samples["positions"].shape[0]
        self.num_used_samples = self.num_selected_samples // (self.opt["skipSamples"] + 1)

    # ------------------------------------------------------------------ block selection (data.py:160-345)
    def _slice_block(self, pos: int, size: int) -> dict[str, np.ndarray]:
        return {k: (v if v.ndim == 0 else v[pos:pos + size]) for k, v in self.measurements.items()}

    def removeLastSampleBlock(self) -> None:
        """Drop the last ``blockSize`` selected samples again (data.py:160-177)."""
        bs = self.opt["blockSize"]
        n = self.num_selected_samples
        for k in self.measurements.keys():
            if self.samples[k].ndim:
                self.samples[k] = self.samples[k][: n - bs]
        self.updateNumSamples()

    def getNextSampleBlock(self) -> None:
        """Replace the working samples by the next measurement block (data.py:179-203); the last block may be shorter, and like
        the reference ``opt['blockSize']`` is shrunk to it."""
        self.block_pos += self.opt["blockSize"]
        if self.block_pos + self.opt["blockSize"] > self.num_loaded_samples:
            self.opt["blockSize"] = self.num_loaded_samples - self.block_pos
        self.samples = self._slice_block(self.block_pos, self.opt["blockSize"])
        self.updateNumSamples()

    def block_positions(self) -> list[tuple[int, int]]:
        """(start, size) of every block the reference's loop visits (identifier.py:1564-1583 with hasMoreSamples /
        getNextSampleBlock): all multiples of blockSize below the number of loaded samples, the last one possibly short."""
        bs, N = int(self.opt["blockSize"]), int(self.num_loaded_samples)
        out, pos = [], 0
        while True:
            out.append((pos, min(bs, N - pos)))
            if pos + bs >= N:
                return out
            pos += bs

    def getBlockStats(self, model) -> None:
        """Record (start, size, cond(YBase), per-link sub-regressor condition numbers) of the block the model was just evaluated
        on (data.py:205-252).  cond(YBase) is the condition number of the triangular factor R with R^T R = YBase^T YBase
        (``Model.base_factor``: the TSQR of the path, or a host QR of a materialised YBase) -- the same singular values as the
        tall matrix the reference hands to ``numpy.linalg.cond``."""
        self.model = model
        R = model.base_factor()
        self.seenBlocks.append((self.block_pos, self.opt["blockSize"], float(np.linalg.cond(R)), model.getSubregressorsConditionNumbers(R)))

    def getAllBlockStats(self, model) -> None:
        """The whole loop of identifier.py:1564-1583 in one pass over the measurements: the blocks do not depend on each other
        (``getNextSampleBlock`` replaces the samples), so ``fbr_gram_grouped`` reduces every full block to its own Gram at once
        (+ one plain Gram for a short last block) and the condition numbers are eigenvalues of nb x nb matrices:
        cond(YBase_b) = sqrt(lmax / lmin) of G_b[ic, ic].  A block whose Gram is too ill conditioned for that (ratio > 1e13:
        the smallest eigenvalue would carry < 3 digits) is re-done through its triangular factor (``fbr_tsqr_cols``)."""
        self.model = model
        # The one-pass form evaluates the raw measurement states against ONE basis.  Where the reference's per-block loop does
        # something the grouped reduction does not see -- velocities / accelerations zeroed for gravity-only identification
        # (model.py:382-385), a projected or filtered base regressor (useBasisProjection, filterRegressor), a basis recomputed from
        # every block's own regressor (useStructuralRegressor 0, model.py:598-601) -- the loop itself is run, block by block.
        o = self.opt
        if o.get("identifyGravityParamsOnly") or o.get("useBasisProjection") or o.get("filterRegressor") or not o.get("useStructuralRegressor", 1):
            return self._block_stats_loop(model)
        skip = int(self.opt["skipSamples"]) + 1
        blocks = self.block_positions()
        ic = np.asarray(model.independent_cols)
        cols_of_link = model.base_columns_of_links()

        def stats_from_gram(G):
            Gb = G[np.ix_(ic, ic)]
            ev = np.linalg.eigvalsh(Gb)
            if not ev[0] > 1e-13 * ev[-1]:
                return None
            conds = []
            for cols in cols_of_link:
                if not cols:
                    conds.append(1e16)
                    continue
                e = np.linalg.eigvalsh(Gb[np.ix_(cols, cols)])
                if not e[0] > 1e-13 * e[-1]:
                    return None
                conds.append(float(np.sqrt(e[-1] / e[0])))
            return float(np.sqrt(ev[-1] / ev[0])), conds

        def states_of(pos, size):
            return model._states_from_samples(self.measurements, pos + np.arange(size // skip) * skip)

        def stats_from_factor(pos, size):
            R = np.asarray(model.engine.tsqr(states_of(pos, size), cols=ic.astype(np.int32)))
            return float(np.linalg.cond(R)), model.getSubregressorsConditionNumbers(R)

        full = [b for b in blocks if b[1] == blocks[0][1]]
        rest = [b for b in blocks if b[1] != blocks[0][1]]
        results = {}
        if full and full[0][1] // skip > 0:
            used = full[0][1] // skip
            idx = np.concatenate([pos + np.arange(used) * skip for pos, _ in full])
            Gs = np.asarray(model.engine.gram_grouped(model._states_from_samples(self.measurements, idx), len(full)))
            for (pos, size), G in zip(full, Gs):
                results[pos] = stats_from_gram(G)
        for pos, size in rest:
            results[pos] = stats_from_gram(np.asarray(model.engine.gram(states_of(pos, size)))) if size // skip > 0 else None
        for pos, size in blocks:
            st = results.get(pos)
            if st is None:
                st = stats_from_factor(pos, size)
            self.seenBlocks.append((pos, size, st[0], st[1]))
        # leave the cursor, the block size and the working samples where the reference's loop leaves them: on the last block
        self.block_pos, self.opt["blockSize"] = blocks[-1]
        self.samples = self._slice_block(*blocks[-1])
        self.updateNumSamples()

    def _block_stats_loop(self, model) -> None:
        """identifier.py:1564-1583 block by block: regressors of the working block, its statistics, next block."""
        self.block_pos = 0
        self.samples = self._slice_block(0, min(int(self.opt["blockSize"]), int(self.num_loaded_samples)))
        self.updateNumSamples()
        while True:
            model.computeRegressors(self)
            self.getBlockStats(model)
            if not self.hasMoreSamples():
                break
            self.getNextSampleBlock()

    def selectBlocks(self) -> None:
        """Keep the blocks whose condition number is within the best ``selectBestPerenctage`` percent, then thin out blocks
        whose per-link condition patterns are near-duplicates (variance of the link condition numbers within 15 % of a
        neighbour in sorted order: of two close ones the first goes, of three the middle one) -- data.py:254-312."""
        conds = np.array([b[2] for b in self.seenBlocks], dtype=float)
        limit = np.percentile(conds, self.opt["selectBestPerenctage"])
        for blk in self.seenBlocks:
            (self.unusedBlocks if blk[2] > limit else self.usedBlocks).append(blk)
        c = len(self.usedBlocks)
        if c == 0:
            return
        var = np.var(np.array([blk[3] for blk in self.usedBlocks], dtype=float).reshape(c, -1), axis=1)
        order = np.argsort(var)
        sv = var[order]
        drop, rel, i = [], 0.15, 1
        while i < c:
            if i < c - 1 and abs(sv[i - 1] - sv[i + 1]) < abs(sv[i + 1]) * rel:
                drop.append(order[i])
                i += 1
            elif abs(sv[i - 1] - sv[i]) < abs(sv[i]) * rel:
                drop.append(order[i - 1])
            i += 1
        for d in np.sort(drop)[::-1]:
            del self.usedBlocks[d]

    def assembleSelectedBlocks(self) -> None:
        """Concatenate the selected blocks into the working samples (data.py:314-345): 2-D channels stacked, every 1-D channel
        continued like a clock (re-based to start one step after the previous block's last value), scalars / the contact
        dictionary kept whole, as the reference does."""
        if self.usedBlocks:
            out = {}
            for k, m in self.measurements.items():
                if m.ndim == 0:
                    out[k] = m
                    continue
                b0, s0 = self.usedBlocks[0][0], self.usedBlocks[0][1]
                acc = m[b0:b0 + s0]
                for b, bs, _, _ in self.usedBlocks[1:]:
                    mv = m[b:b + bs]
                    if m.ndim == 1:
                        mv = mv - mv[0] + (mv[1] - mv[0]) + acc[-1]
                    acc = np.concatenate((acc, mv), axis=0)
                out[k] = acc
            self.samples = out
        self.updateNumSamples()

    # ------------------------------------------------------------------ step before the path (N2)
    @staticmethod
    def _central_diff(a: np.ndarray, times: np.ndarray) -> np.ndarray:
        """4th-order central difference of data.py:396-418 (n = 2), vectorised.  The divisor of sample i is
        times[i] - times[i-1]; the two leading samples use times[1] - times[0] and the two trailing ones the step of the
        last interior sample (the reference's loop variable is still set to it) -- reproduced, not 'corrected'."""
        size = a.shape[0]
        d = np.zeros_like(a)
        div0 = times[1] - times[0]
        d[0] = (a[1] - a[0]) / div0
        d[1] = (a[2] - a[0]) / (2 * div0)
        if size > 4:
            dv = (times[2:size - 2] - times[1:size - 3]).reshape((-1,) + (1,) * (a.ndim - 1))
            d[2:size - 2] = (-a[4:] + 8 * a[3:size - 1] - 8 * a[1:size - 3] + a[:size - 4]) / (12 * dv)
            div_last = times[size - 3] - times[size - 4]
        else:
            div_last = div0
        d[size - 2] = (a[size - 1] - a[size - 3]) / (2 * div_last)
        d[size - 1] = (a[size - 1] - a[size - 2]) / div_last
        return d

    def preprocess(self, Q, V, Vdot, Tau, T, Fs, Q_raw=None, V_raw=None, Tau_raw=None, IMUlinVel=None, IMUrotVel=None,
                   IMUlinAcc=None, IMUrotAcc=None, IMUrpy=None, FT=None, engine=None) -> None:
        """Derivation and filtering of measurements, arrays changed in place like the reference (data.py:369-619):
        Q, Tau filtered; V, Vdot, *_raw overwritten; IMUrotVel, IMUlinAcc, IMUrpy filtered; IMUlinVel, IMUrotAcc
        overwritten; every FT array median + low-pass filtered (first three columns).  All columns of an array are
        filtered in one SciPy call (identical per-column arithmetic).

        ``engine`` (a ``flobaroid_amd._lib.Engine``): the zero-phase low-passes, the median filters and the central differences run on
        the GPU (``fbr_filtfilt`` / ``fbr_medfilt`` / ``fbr_central_diff``: time-parallel blocked scan of the filter state, all
        channels of an array at once) -- same results to rounding, the filter design (``scipy.signal.butter``, a handful of
        coefficients) and the IMU frame bookkeeping stay on the host."""
        import scipy.integrate
        import scipy.signal as sig

        from .topology import rpy_to_matrix

        k = self.opt["filterMedianSize"]
        n = self.opt["num_dofs"]
        if engine is None:
            med = lambda X: sig.medfilt(X, (k, 1))
            filtfilt = lambda b, a, X: sig.filtfilt(b, a, X, axis=0)
            cdiff = self._central_diff
        else:
            # the device kernels cover what the shipped configurations use (median windows <= 31, filter orders <= 11, >= 5 samples);
            # anything beyond runs through the host implementation for that array, like the reference
            host_med = lambda X: sig.medfilt(X, (k, 1))
            host_ff = lambda b, a, X: sig.filtfilt(b, a, X, axis=0)
            med = (lambda X: engine.medfilt(k, np.ascontiguousarray(X, dtype=np.float64).copy())) if k <= 31 else host_med
            filtfilt = lambda b, a, X: (engine.filtfilt(b, a, np.ascontiguousarray(X, dtype=np.float64).copy())
                                        if max(len(a), len(b)) - 1 <= 11 and X.shape[0] > 3 * max(len(a), len(b)) else host_ff(b, a, X))
            cdiff = lambda A, times: (engine.central_diff(np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(times, dtype=np.float64))
                                      if A.shape[0] >= 5 else self._central_diff(A, times))
        if self.opt["useDeg"]:
            np.copyto(Q, np.deg2rad(Q))
            np.copyto(V, np.deg2rad(V))
        lp = {}
        for key in ("filterLowPass1", "filterLowPass2", "filterLowPass3"):
            fc, order = self.opt[key][0], self.opt[key][1]
            lp[key] = sig.butter(order, fc / (Fs / 2), btype="low", analog=False)
        b8, a8 = lp["filterLowPass1"]
        b6, a6 = lp["filterLowPass2"]
        b3, a3 = lp["filterLowPass3"]
        # joint positions: low-pass
        Q_orig = Q.copy()
        Q[:, :n] = filtfilt(b8, a8, Q_orig[:, :n])
        if Q_raw is not None:
            np.copyto(Q_raw, Q_orig)
        # joint velocities: derivative of the filtered positions, median, low-pass
        Vs = cdiff(Q, T)
        if V_raw is not None:
            np.copyto(V_raw, Vs)
        Vs[:, :n] = med(Vs[:, :n].copy())
        Vs[:, :n] = filtfilt(b6, a6, Vs[:, :n].copy())
        np.copyto(V, Vs)
        # joint accelerations: derivative of the velocities, median
        np.copyto(Vdot, cdiff(Vs, T))
        Vdot[:, :n] = med(Vdot[:, :n].copy())
        # joint torques: median, low-pass
        if Tau_raw is not None:
            np.copyto(Tau_raw, Tau)
        Tau[:, :n] = med(Tau[:, :n].copy())
        Tau[:, :n] = filtfilt(b8, a8, Tau[:, :n].copy())
        # IMU
        if IMUlinAcc is not None and IMUrotVel is not None:
            IMUlinAcc[:, :3] = med(IMUlinAcc[:, :3].copy())
            IMUrotVel[:, :3] = med(IMUrotVel[:, :3].copy())
            IMUlinAcc[:, :3] = filtfilt(b8, a8, IMUlinAcc[:, :3].copy())
            IMUrotVel[:, :3] = filtfilt(b8, a8, IMUrotVel[:, :3].copy())
            IMUrpy[:, :3] = filtfilt(b3, a3, IMUrpy[:, :3].copy())
            if IMUlinVel is not None:
                # rotate to the (estimated) world frame
                R = np.stack([rpy_to_matrix(r) for r in IMUrpy])
                accW = np.einsum("sij,sj->si", R, IMUlinAcc)
                np.copyto(IMUrotVel, np.einsum("sij,sj->si", R, IMUrotVel))
                grav_norm = np.mean(np.linalg.norm(accW, axis=1))
                if grav_norm < 9.81 or grav_norm > 9.82:
                    print(f"Warning: mean base acceleration is different than gravity ({grav_norm})!")
                accW -= np.array([0, 0, -9.81])
                if self.opt["waitForZeroAcc"]:
                    means = np.mean(accW, axis=0)
                    accW -= means
                    start = 0
                    for j in range(3):
                        for s_ in range(accW.shape[0]):
                            if np.linalg.norm(accW[s_:s_ + 10, j]) < self.opt["zeroAccThresh"]:
                                start = max(s_, start)
                                break
                    accW[:start, :] = 0
                    accW += means
                elif np.linalg.norm(accW[:, 0]) > 0.1:
                    print("Warning: proper base acceleration not zero at time 0 (assuming start at zero, integrated velocity will be wrong)!")
                accW -= np.mean(accW, axis=0)
                np.copyto(IMUlinAcc, accW)
                for j in range(3):
                    IMUlinVel[:, j] = scipy.integrate.cumulative_trapezoid(IMUlinAcc[:, j], T, initial=0)
                    IMUlinVel[:, j] -= np.mean(IMUlinVel[:, j])
            if IMUrotAcc is not None:
                for j in range(3):
                    IMUrotAcc[:, j] = np.gradient(IMUrotVel[:, j])
        # contact wrenches
        if FT is not None:
            for ft in FT:
                ft[:, :3] = med(ft[:, :3].copy())
                ft[:, :3] = filtfilt(b3, a3, ft[:, :3].copy())
