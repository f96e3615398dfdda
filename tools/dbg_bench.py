import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
import torch
import bench
# monkeypatch to sync after each phase
from go1_gym_learn.ppo_cse import ppo as P
orig_act = P.PPO.act
def act(self, *a):
    r = orig_act(self, *a); torch.cuda.synchronize(); return r
P.PPO.act = act
orig_pes = P.PPO.process_env_step
def pes(self, *a):
    torch.cuda.synchronize(); print("env.step ok", flush=True)
    r = orig_pes(self, *a); torch.cuda.synchronize(); return r
P.PPO.process_env_step = pes
orig_cr = P.PPO.compute_returns
def cr(self, *a):
    print("compute_returns", flush=True); r = orig_cr(self, *a); torch.cuda.synchronize(); print("cr ok", flush=True); return r
P.PPO.compute_returns = cr
orig_cs = P.PPO._clip_and_step
def cs(self, *a, **k):
    torch.cuda.synchronize(); print("bwd ok", flush=True); r = orig_cs(self, *a, **k); torch.cuda.synchronize(); print("step ok", flush=True); return r
P.PPO._clip_and_step = cs
bench.main()
