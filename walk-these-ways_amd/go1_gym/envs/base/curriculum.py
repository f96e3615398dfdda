"""Grid curriculum over the command space — host (numpy) mirror.

Same semantics and public names as the reference's go1_gym/envs/base/curriculum.py
(`Curriculum` :17-89, `RewardThresholdCurriculum` :105-159): an N-d grid of bin centroids
(first key slowest, 'ij' meshgrid flattened), per-bin weights, categorical bin sampling from a
seeded `numpy.random.RandomState`, uniform jitter inside the cell, and the
"success -> +0.2 on the bin and its neighbours" update.

On the MI355X path the per-step sampling/update runs inside the step kernel (device backend,
DESIGN.md "Commands"); this class builds the grid / initial weights / neighbourhood table that
the kernel consumes and remains the reference-exact host implementation for tests and tools.
"""
import numpy as np


def is_met(scale, l2_err, threshold):
    """reference curriculum.py:6-7"""
    return (l2_err / scale) < threshold


def key_is_met(metric_cache, config, ep_len, target_key, env_id, threshold):
    """reference curriculum.py:10-14: a stub there as well (scale 1, error 0: always met for a positive threshold)"""
    return is_met(1, 0, threshold)


class Curriculum:
    def __init__(self, seed, **key_ranges):
        self.rng = np.random.RandomState(seed)
        self.keys = list(key_ranges)
        lo = np.array([r[0] for r in key_ranges.values()], dtype=float)
        hi = np.array([r[1] for r in key_ranges.values()], dtype=float)
        nb = np.array([r[2] for r in key_ranges.values()], dtype=int)
        self.lows, self.highs, self.num_bins_per_key = lo, hi, nb
        width = (hi - lo) / nb
        self.bin_sizes = dict(zip(self.keys, width))
        self.cfg = {k: np.linspace(l + w / 2, h - w / 2, n) for k, l, h, w, n in zip(self.keys, lo, hi, width, nb)}
        self.ls = {k: int(n) for k, n in zip(self.keys, nb)}
        self._raw_grid = np.stack(np.meshgrid(*self.cfg.values(), indexing='ij'))
        self._idx_grid = np.stack(np.meshgrid(*[np.linspace(0, n - 1, n) for n in nb], indexing='ij'))
        self.grid = self._raw_grid.reshape(len(self.keys), -1)
        self.idx_grid = self._idx_grid.reshape(len(self.keys), -1)
        self._l = self.grid.shape[1]
        self.weights = np.zeros(self._l)
        self.indices = np.arange(self._l)

    def __len__(self):
        return self._l

    def __getitem__(self, *keys):
        pass                                           # (a stub in the reference too, curriculum.py:61-62)

    def set_to(self, low, high, value=1.0):
        inside = np.logical_and(self.grid >= low[:, None], self.grid <= high[:, None]).all(axis=0)
        assert len(inside) != 0, "You are intializing your distribution with an empty domain!"
        self.weights[inside] = value

    def sample_bins(self, batch_size, low=None, high=None):
        w = self.weights
        if low is not None and high is not None:
            valid = np.logical_and(self.grid >= low[:, None], self.grid <= high[:, None]).all(axis=0)
            w = np.where(valid, self.weights, 0.0)
        inds = self.rng.choice(self.indices, batch_size, p=w / w.sum())
        return self.grid.T[inds], inds

    def sample_uniform_from_cell(self, centroids):
        half = np.array(list(self.bin_sizes.values())) / 2
        return self.rng.uniform(centroids + half, centroids - half)

    def sample(self, batch_size, low=None, high=None):
        centroids, inds = self.sample_bins(batch_size, low=low, high=high)
        return np.stack([self.sample_uniform_from_cell(c) for c in centroids]), inds

    def update(self, **kwargs):
        pass


class SumCurriculum(Curriculum):
    """Success / trial counts per bin (reference curriculum.py:92-109; not used by scripts/train.py, whose
    `curriculum_type` is RewardThresholdCurriculum)."""

    def __init__(self, seed, **kwargs):
        super().__init__(seed, **kwargs)
        self.success = np.zeros(len(self))
        self.trials = np.zeros(len(self))

    def update(self, bin_inds, l1_error, threshold):
        ok = l1_error < threshold
        self.success[bin_inds[ok]] += 1                # (fancy-index +=: a bin listed twice counts once, as in the reference)
        self.trials[bin_inds] += 1

    def success_rates(self, *keys):
        """success rate per bin on the N-d grid, averaged over the axes whose key is not named (all axes kept if none is left out)"""
        rate = (self.success / (self.trials + 1e-6)).reshape(list(self.ls.values()))
        drop = tuple(i for i, k in enumerate(self.keys) if k not in keys)
        return rate.mean(axis=drop) if drop else rate


class RewardThresholdCurriculum(Curriculum):
    def __init__(self, seed, **kwargs):
        super().__init__(seed, **kwargs)
        n = len(self)
        self.episode_reward_lin = np.zeros(n)
        self.episode_reward_ang = np.zeros(n)
        self.episode_lin_vel_raw = np.zeros(n)
        self.episode_ang_vel_raw = np.zeros(n)
        self.episode_duration = np.zeros(n)

    def get_local_bins(self, bin_inds, ranges=0.1):
        if isinstance(ranges, float):
            ranges = np.ones(self.grid.shape[0]) * ranges
        bin_inds = np.asarray(bin_inds).reshape(-1)
        centre = self.grid[:, bin_inds, None]                      # (K, B, 1)
        span = np.asarray(ranges).reshape(-1, 1, 1)
        g = self.grid[:, None, :]                                  # (K, 1, L)
        return np.logical_and(g >= centre - span, g <= centre + span).all(axis=0)   # (B, L)

    def update(self, bin_inds, task_rewards, success_thresholds, local_range=0.5):
        bin_inds = np.asarray(bin_inds)
        if len(success_thresholds) == 0:
            success = np.zeros(len(bin_inds), dtype=bool)
        else:
            success = np.ones(len(bin_inds), dtype=bool)
            for rew, thr in zip(task_rewards, success_thresholds):
                rew = rew.cpu().numpy() if hasattr(rew, "cpu") else np.asarray(rew)
                success &= rew > thr
        won = bin_inds[success]
        self.weights[won] = np.clip(self.weights[won] + 0.2, 0, 1)
        for row in self.get_local_bins(won, ranges=local_range):
            near = row.nonzero()[0]
            self.weights[near] = np.clip(self.weights[near] + 0.2, 0, 1)

    def log(self, bin_inds, lin_vel_raw=None, ang_vel_raw=None, episode_duration=None):
        self.episode_lin_vel_raw[bin_inds] = lin_vel_raw.cpu().numpy()
        self.episode_ang_vel_raw[bin_inds] = ang_vel_raw.cpu().numpy()
        self.episode_duration[bin_inds] = episode_duration.cpu().numpy()

    def neighbourhood_csr(self, local_range):
        """CSR (ptr, idx) of get_local_bins for every bin — consumed by the device curriculum kernel."""
        ptr, idx = [0], []
        step = 256
        for s in range(0, len(self), step):
            rows = self.get_local_bins(np.arange(s, min(s + step, len(self))), ranges=local_range)
            for row in rows:
                nz = row.nonzero()[0]
                idx.extend(nz.tolist())
                ptr.append(len(idx))
        return np.asarray(ptr, dtype=np.int32), np.asarray(idx, dtype=np.int32)
