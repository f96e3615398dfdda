"""Running-mean caches used for curriculum logging (surface of reference go1_gym_learn/ppo/metrics_caches.py:8-90;
logging only — kept because ppo_cse/__init__.py:34 instantiates them)."""
from collections import defaultdict

import numpy as np


class DistCache:
    def __init__(self):
        self.cache = defaultdict(lambda: 0)

    def log(self, **key_vals):
        for k, v in key_vals.items():
            n = self.cache[k + '@counts'] + 1
            self.cache[k + '@counts'] = n
            self.cache[k] = (v + (n - 1) * self.cache[k]) / n

    def get_summary(self):
        out = {k: v for k, v in self.cache.items() if not k.endswith("@counts")}
        self.cache.clear()
        return out


class SlotCache:
    def __init__(self, n):
        self.n = n
        self.cache = defaultdict(lambda: np.zeros([n]))

    def log(self, slots=None, **key_vals):
        if slots is None:
            slots = range(self.n)
        for k, v in key_vals.items():
            counts = self.cache[k + '@counts'][slots] + 1
            self.cache[k + '@counts'][slots] = counts
            self.cache[k][slots] = (v + (counts - 1) * self.cache[k][slots]) / counts

    def get_summary(self):
        out = {k: v for k, v in self.cache.items() if not k.endswith("@counts")}
        self.cache.clear()
        return out
