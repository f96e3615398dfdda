"""Debug (CPU, SIMT emulator): the rough-terrain env under a random policy — why do episodes end?"""
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path[:0] = [os.path.join(REPO, "tools"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "emu"), REPO]
os.environ["GO1_DRY_RUN_GPU_TESTS"] = "1"
import dry_run_gpu_tests  # noqa: E402
dry_run_gpu_tests.install()
import torch  # noqa: E402


def main():
    rough = "--flat" not in sys.argv
    N = 64
    import bench
    torch.cuda.current_device = lambda: 0
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=N)
    if rough:
        t = cfg.terrain
        t.mesh_type, t.terrain_proportions, t.curriculum = os.environ.get("MESH", "trimesh"), [0.1, 0.1, 0.35, 0.25, 0.2], True
        t.num_rows, t.num_cols, t.terrain_length, t.terrain_width, t.border_size, t.center_robots = 10, 20, 8.0, 8.0, 25.0, False
        t.measure_heights = True
        cfg.env.observe_heights = True
        cfg.env.num_observations = cfg.env.num_scalar_observations = 70 + 187
    from go1_gym.envs.base.base_task import BaseTask

    def _cpu(self, sim_device):          # (debug only: the product refuses anything but cuda:N)
        self.sim_device_id = 0
        return "cpu"
    BaseTask._resolve_device = _cpu
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cpu", headless=True, cfg=cfg)
    base = env
    env.reset()
    B = base.buffers
    print("terrain levels", base.terrain_levels.tolist()[:16], "types", base.terrain_types.tolist()[:16])
    t0 = time.time()
    reasons = {"base_contact": 0, "body_height": 0, "time_out": 0, "other": 0}
    resets = 0
    vmax = 0.0
    for step in range(int(os.environ.get("STEPS", 120))):
        actions = torch.randn(N, 12)
        pre_z = base.root_states[:, 2].clone()
        obs, rew, done, info = env.step(actions)
        d = done.bool()
        sp = base.root_states[:, 7:10].norm(dim=1)
        vmax = max(vmax, float(sp.max()))
        fast = (sp > 2.5).nonzero().flatten().tolist()
        for e in fast[:3]:
            rs = base.root_states[e]
            print(f"  step {step}: env {e} type {int(base.terrain_types[e])} level {int(base.terrain_levels[e])} |v| {float(sp[e]):.2f} v {[round(float(x), 2) for x in rs[7:10]]} "
                  f"pos-origin {[round(float(x), 3) for x in (rs[:3] - base.env_origins[e])]} quat {[round(float(x), 2) for x in rs[3:7]]} ep_len {int(base.episode_length_buf[e])} "
                  f"Fz feet {[round(float(x), 0) for x in base.contact_forces[e, base.feet_indices, 2]]} |F| max {float(base.contact_forces[e].norm(dim=-1).max()):.0f} body {int(base.contact_forces[e].norm(dim=-1).argmax())}", flush=True)
        if d.any():
            for e in d.nonzero().flatten().tolist():
                resets += 1
        if step % 20 == 19:
            fl = base.extras["sim_faults"].consume()
            print(f"step {step + 1}: resets so far {resets}  time-outs {int(base.time_out_buf.sum())}  max|v| so far {vmax:.2f}  mean reward {float(rew.mean()):.5f}  "
                  f"faults {{k: v for k, v in fl.items() if v}}  [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
