"""SURVEY 8(d), last row: "reference Python" cost of the non-physics part of `LeggedRobot.step` — the reference's OWN
methods (borrowed from /root/reference exactly as tests/golden/make_golden.py does: test-only stubs for the absent
packages, a mock env carrying synthetic state) executed by PyTorch on the host CPU at N = 4096:
4 x `_compute_torques` (TorchScript actuator net) + `_step_contact_targets` + `check_termination` + `compute_reward`
+ `compute_observations` per policy step.  Only runs where /root/reference exists (the authoring container); writes
profiles/reference_python_maps_cpu.json, which bench.py attaches to its JSON line with this provenance.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main(N=4096, reps=20):
    import make_golden as G
    G.install_stubs()
    e, LR = G.make_env("train", N, seed=1)
    net = torch.jit.load(os.path.join(G.REF, "resources/actuator_nets/unitree_go1.pt"), map_location="cpu")
    e.actuator_network = lambda p, pl, pll, v, vl, vll: net(torch.stack((p, pl, pll, v, vl, vll), dim=-1).view(N * 12, 6)).view(N, 12)
    e.lag_buffer = [torch.zeros(N, 12) for _ in range(e.cfg.domain_rand.lag_timesteps + 1)]
    for n in ("joint_pos_err_last_last", "joint_pos_err_last", "joint_vel_last_last", "joint_vel_last"):
        setattr(e, n, torch.zeros(N, 12))
    a = torch.randn(N, 12)
    threads = torch.get_num_threads()

    def one_step():
        with torch.no_grad():
            for _ in range(e.cfg.control.decimation):
                LR._compute_torques(e, a)
            LR._step_contact_targets(e)
            LR.check_termination(e)
            LR.compute_reward(e)
            LR.compute_observations(e)
    for _ in range(3):
        one_step()
    t0 = time.perf_counter()
    for _ in range(reps):
        one_step()
    dt = (time.perf_counter() - t0) / reps
    out = {"value": N / dt, "unit": "env-steps/s", "ms_per_policy_step": 1e3 * dt, "envs": N, "threads": threads,
           "what": "reference go1_gym methods (legged_robot.py:_compute_torques x4, _step_contact_targets, check_termination, "
                   "compute_reward, compute_observations) executed by PyTorch CPU on a mock env; physics (Isaac Gym) NOT included",
           "source": "tools/time_reference_maps.py run in the authoring container (the only place /root/reference exists), "
                     f"{os.cpu_count()} logical CPUs; committed, not measured by the bench run that quotes it"}
    with open(os.path.join(REPO, "profiles", "reference_python_maps_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
