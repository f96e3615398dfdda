"""Local stand-in for the `ml_logger` package: the subset of `logger` the Go1 scripts and
the Runner use, writing under a local root instead of a remote instrument server."""
import contextlib
import datetime
import os
import pickle
import shutil
import time
from collections import defaultdict


class _Logger:
    def __init__(self):
        self.root = None
        self.prefix = "."
        self._metrics = defaultdict(list)
        self._metric_prefix = ""
        self._timers = {}
        self._counters = defaultdict(int)
        self.print_summary = True

    # ---- configuration -------------------------------------------------------------------------
    def configure(self, prefix=None, root=None, **kw):
        self.root = str(root) if root is not None else os.path.abspath("runs")
        self.prefix = prefix or "."
        os.makedirs(self._path(""), exist_ok=True)
        # a new run: nothing of the previous one (pending metrics, `every` phases, timers) carries over
        self._metrics.clear()
        self._counters.clear()
        self._timers.clear()

    def utcnow(self, fmt="%Y-%m-%d/%H%M%S.%f"):
        return datetime.datetime.utcnow().strftime(fmt)

    def _path(self, key):
        return os.path.join(self.root or os.path.abspath("runs"), self.prefix, key)

    # ---- text / params ---------------------------------------------------------------------------
    def log_text(self, text, filename="text.log", dedent=False, **kw):
        if dedent:
            import textwrap
            text = textwrap.dedent(text)
        p = self._path(filename)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "a") as f:
            f.write(text)

    def log_params(self, path="parameters.pkl", **kwargs):
        self.save_pkl(kwargs, path)

    def save_pkl(self, data, path, append=False):
        p = self._path(path)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "ab" if append else "wb") as f:
            pickle.dump(data, f)

    def load_pkl(self, path):
        out = []
        with open(self._path(path), "rb") as f:
            while True:
                try:
                    out.append(pickle.load(f))
                except EOFError:
                    return out

    # ---- timers ------------------------------------------------------------------------------------
    def start(self, *keys):
        now = time.perf_counter()
        for k in keys:
            self._timers[k] = now

    def since(self, key):
        return time.perf_counter() - self._timers.get(key, time.perf_counter())

    def split(self, key):
        now = time.perf_counter()
        last = self._timers.get(key, now)
        self._timers[key] = now
        return now - last

    def every(self, n, key="default", start_on=0):
        self._counters[key] += 1
        return (self._counters[key] - start_on) % n == 0 and self._counters[key] >= start_on

    # ---- metrics -------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def Prefix(self, *a, metrics=None, **kw):
        old = self._metric_prefix
        if metrics:
            self._metric_prefix = metrics.rstrip("/") + "/"
        try:
            yield
        finally:
            self._metric_prefix = old

    def store_metrics(self, **kv):
        for k, v in kv.items():
            self._metrics[self._metric_prefix + k].append(v)

    def log_metrics_summary(self, key_values=None, **kw):
        summary = dict(key_values or {})
        for k, vals in self._metrics.items():
            try:
                nums = [float(v) for v in vals]
                summary[k + "/mean"] = sum(nums) / max(len(nums), 1)
            except (TypeError, ValueError):
                pass
        self._metrics.clear()
        self.save_pkl(summary, "metrics.pkl", append=True)
        if self.print_summary:
            keys = ["iterations", "timesteps", "train/episode/rew_total/mean", "adaptation_loss/mean", "time_iter/mean"]
            print(" | ".join(f"{k}={summary[k]:.4g}" for k in keys if k in summary))
        return summary

    def job_running(self):
        pass

    @contextlib.contextmanager
    def Sync(self):
        yield

    # ---- artefacts -------------------------------------------------------------------------------------
    def torch_save(self, obj, path):
        import torch
        p = self._path(path)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        torch.save(obj, p)

    def load_torch(self, path, **kw):
        import torch
        return torch.load(self._path(path), **kw)

    def duplicate(self, src, dst):
        shutil.copyfile(self._path(src), self._path(dst))

    def upload_file(self, file_path, target_path, once=True):
        dst = self._path(os.path.join(target_path, os.path.basename(file_path)))
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(file_path, dst)

    def save_video(self, frames, path, fps=30):
        pass          # no renderer on this stack (SURVEY.md §5 "video")


logger = _Logger()
ML_Logger = _Logger
