"""A/B of step-kernel builds on the sim-only rate (4096 envs, train.py configuration): the product library and variant builds of the same
sources (e.g. -DGO1_PGS_LEGS), each under N(0, 1) actions (robots falling and tangling: the bench's regime) and under zero actions (standing on
four feet: one contact per leg).  Timing only — a variant's results are validated on the CPU (tests/test_emu_parity.py), not here.

    python tools/probes/step_variant_ab.py [variant.so ...]"""
import os
import sys
import time

R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P_ = os.path.join(R_, "walk-these-ways_amd")
for p in (os.path.join(P_, "shims"), P_, R_):
    sys.path.insert(0, p)
import torch  # noqa: E402
import go1sim_host as H  # noqa: E402


def rate(lib_path, zero_actions, envs=4096, steps=480):
    if lib_path:
        H.LIB_PATH, H._lib = os.path.abspath(lib_path), None
    from bench import build_env
    env, cfg = build_env(envs, 0, 0, rough=bool(os.environ.get("AB_ROUGH")))      # AB_ROUGH=1: BASELINE configs[2] (height field with walls, 257 observations)
    env.reset()
    if os.environ.get("AB_RESETS"):      # AB_RESETS=1: episode lengths spread over the horizon as the runner leaves them (ppo_cse/__init__.py:113) — about
        buf = env.episode_length_buf     # envs / max_episode_length time-outs per step, so nearly every launch holds a workgroup that resets
        buf.copy_(torch.randint_like(buf, high=int(env.max_episode_length)))
    acts = (torch.zeros if zero_actions else torch.randn)(24, envs, 12, device="cuda")
    for i in range(48):
        env.step(acts[i % 24])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 24])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return envs * steps / dt, 1e3 * dt / steps


def envs_sweep():
    """AB_ENVS="256,1024,2048,4096,8192": per-step time of the default library against the number of environments (= workgroups of 16): how much
    of the 4096-environment time is contention between workgroups (two CUs share an instruction cache; the step kernel's loop body is ~240 KB
    of code) and how much is the one workgroup's own serial spine"""
    rate(H.LIB_PATH, False, steps=1500)
    for envs in [int(x) for x in os.environ["AB_ENVS"].split(",")]:
        for zero in (False, True):
            v = sorted(rate(H.LIB_PATH, zero, envs=envs)[1] for _ in range(3))
            print(f"{envs:6d} envs ({envs // 16:4d} workgroups) {'zero actions (standing)' if zero else 'N(0,1) actions':24s}: median {v[1]:.4f} ms per env.step", flush=True)


def main():
    """every library in turn, REPS times round-robin (the first measurement of a process runs on a GPU that is still ramping its clocks: round 5's
    first A/B had the product library first and read 8 % slow); the table prints every repetition and the median"""
    default = H.LIB_PATH
    reps = int(os.environ.get("AB_REPS", "4"))
    libs = [None] + sys.argv[1:]
    rate(default, False, steps=1500)                       # warm the clocks
    res = {}
    for rep in range(reps):
        for path in libs:
            for zero in (False, True):
                r, ms = rate(path or default, zero)
                res.setdefault((path, zero), []).append(ms)
    for path in libs:
        for zero in (False, True):
            v = sorted(res[(path, zero)])
            med = 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])
            print(f"{os.path.basename(path or default):28s} {'zero actions (standing)' if zero else 'N(0,1) actions':24s}: median {med:.4f} ms per env.step "
                  f"({4096 / med / 1e3:6.2f} M env-steps/s); runs {' '.join(f'{x:.4f}' for x in res[(path, zero)])}", flush=True)


if __name__ == "__main__":
    envs_sweep() if os.environ.get("AB_ENVS") else main()
