"""Config variants the golden fixtures are generated for (pure data, shared by make_golden.py and the tests).
Each variant = train config (walk-these-ways_amd/scripts/train_config.py) + these overrides."""

VARIANTS = {
    # scripts/train.py as it is: observation noise on (`torch.rand_like` of the reference is fed the kernels' Philox uniforms)
    "train_noise": {},
    # scripts/train.py effective config (noise off: the reference draws it from torch's global RNG)
    "train": {"noise": dict(add_noise=False)},
    # exercises the other branches: 'P' control without lag, velocity/yaw/contact observations, more privileged
    # blocks, every reward function, clipped-positive reward, timing parameter
    "alt": {
        "noise": dict(add_noise=False),
        "control": dict(control_type="P"),
        "domain_rand": dict(randomize_lag_timesteps=False, randomize_Kp_factor=True, randomize_Kd_factor=True),
        "env": dict(observe_vel=True, observe_yaw=True, observe_contact_states=True, observe_timing_parameter=True,
                    observe_two_prev_actions=False, num_observations=70, num_scalar_observations=70,
                    priv_observe_base_mass=True, priv_observe_com_displacement=True, priv_observe_motor_strength=True,
                    priv_observe_motor_offset=True, priv_observe_body_height=True, priv_observe_body_velocity=True,
                    priv_observe_gravity=True, priv_observe_clock_inputs=True, priv_observe_desired_contact_states=True,
                    num_privileged_obs=2 + 1 + 3 + 12 + 12 + 1 + 3 + 3 + 4 + 4),
        "commands": dict(pacing_offset=True),
        "rewards": dict(only_positive_rewards=True, only_positive_rewards_ji22_style=False,
                        use_terminal_body_height=False),
        "reward_scales": dict(tracking_lin_vel=20.0, orientation=-5.0, dof_pos=-0.05, feet_contact_forces=-0.01, feet_impact_vel=-0.1,
                              feet_contact_vel=-0.1),
    },
    # the remaining observation switches: no command block, linear velocity only, no clock inputs (legged_robot.py:319, 358,
    # 336), a single privileged term, roll / pitch termination off.  (`observe_gait_commands=False` cannot be pinned: the
    # reference then never defines `foot_indices`, which train.py's active rewards read, corl_rewards.py:128.)
    "alt2": {
        "noise": dict(add_noise=False),
        "env": dict(observe_command=False, observe_only_lin_vel=True, observe_clock_inputs=False,
                    observe_two_prev_actions=True, num_observations=3 + 12 + 12 + 12 + 12 + 3, num_scalar_observations=54,
                    priv_observe_friction=False, priv_observe_restitution=True, num_privileged_obs=1),
        "rewards": dict(use_terminal_roll_pitch=False),
    },
    # the two remaining combinations of the torque model's switches (train: actuator network with lag; alt: PD without)
    "act_nolag": {"noise": dict(add_noise=False), "domain_rand": dict(randomize_lag_timesteps=False, randomize_Kp_factor=True)},
    "pd_lag": {"noise": dict(add_noise=False), "control": dict(control_type="P"),
               "domain_rand": dict(randomize_lag_timesteps=True, lag_timesteps=3, randomize_Kp_factor=True, randomize_Kd_factor=True)},
    # north_star's "domain-randomisation pushes" and the other step-callback branches that train.py leaves off: velocity
    # pushes (legged_robot.py:1017-1026), edge teleport (:1028-1051), re-drawn rigid-body properties (:706-708,166-168);
    # kernel-vs-oracle only (no fixture: the reference draws these from torch's global RNG)
    "dr": {
        "domain_rand": dict(push_robots=True, push_interval_s=0.25, max_push_vel_xy=0.8, randomize_rigids_after_start=True,
                            rand_interval_s=0.15, randomize_gravity=True),
        "terrain": dict(teleport_robots=True, teleport_thresh=0.4),
    },
}


def random_switches(rng, physics=True):
    """a random but consistent configuration: observation / privileged-observation blocks (widths per legged_robot.py:319-372,
    383-489), reward shaping, termination and command switches; with `physics` also controller and domain-randomisation
    cadence switches (irrelevant for the tensor maps)"""
    pick = lambda p=0.5: bool(rng.random() < p)
    env = dict(observe_command=pick(0.7), observe_two_prev_actions=pick(), observe_timing_parameter=pick(), observe_clock_inputs=pick(),
               observe_vel=False, observe_only_lin_vel=False, observe_yaw=pick(), observe_contact_states=pick())
    vel = rng.integers(0, 3)
    env["observe_vel"], env["observe_only_lin_vel"] = bool(vel == 1), bool(vel == 2)
    width = 39 + 15 * env["observe_command"] + 12 * env["observe_two_prev_actions"] + env["observe_timing_parameter"] + \
        4 * env["observe_clock_inputs"] + 6 * env["observe_vel"] + 3 * env["observe_only_lin_vel"] + env["observe_yaw"] + \
        4 * env["observe_contact_states"]
    env["num_observations"] = env["num_scalar_observations"] = int(width)
    priv_w = dict(friction=1, restitution=1, base_mass=1, com_displacement=3, motor_strength=12, motor_offset=12, body_height=1,
                  body_velocity=3, gravity=3, clock_inputs=4, desired_contact_states=4)
    npv = 0
    for k, w in priv_w.items():
        on = pick(0.4)
        env["priv_observe_" + k] = on
        npv += w * on
    if npv == 0:
        env["priv_observe_friction"], npv = True, 1
    env["num_privileged_obs"] = int(npv)
    positive = pick()
    extra = {"env": env,
             "commands": dict(pacing_offset=pick(), binary_phases=pick(), gaitwise_curricula=pick(),
                              exclusive_phase_offset=False, balance_gait_distribution=pick()),
             "rewards": dict(only_positive_rewards=positive, only_positive_rewards_ji22_style=not positive,
                             use_terminal_body_height=pick(), use_terminal_roll_pitch=pick())}
    if physics:
        extra["control"] = dict(control_type="P" if pick(0.3) else "actuator_net")
        extra["domain_rand"] = dict(randomize_lag_timesteps=pick(0.7), randomize_Kp_factor=pick(), randomize_Kd_factor=pick(),
                                    push_robots=pick(), push_interval_s=0.1, randomize_rigids_after_start=pick(), rand_interval_s=0.1)
    return extra


# tensor-map fixtures under random switch sets (reference-pinned like "alt": tests/golden/maps_fuzz<k>_mild.npz)
FUZZ_VARIANTS = 6
for _k in range(FUZZ_VARIANTS):
    import numpy as _np
    VARIANTS[f"fuzz{_k}"] = dict(random_switches(_np.random.default_rng(500 + _k), physics=False), noise=dict(add_noise=False))


def apply_variant(Cfg, name):
    for section, values in VARIANTS[name].items():
        target = getattr(Cfg, section)
        for k, v in values.items():
            setattr(target, k, v)
    return Cfg


# PPO_Args settings of tests/golden/ppo_fuzz<k>.npz (reference ppo.py:11-31): fixed / adaptive schedule, plain value loss, several
# adaptation sub-steps, selective adaptation loss, other clip / entropy / value coefficients, epochs, mini-batch counts
PPO_FUZZ = [dict(schedule="fixed", use_clipped_value_loss=False, clip_param=0.1, entropy_coef=0.0, num_learning_epochs=2, num_mini_batches=3,
                 num_adaptation_module_substeps=2, max_grad_norm=0.3),
            dict(schedule="adaptive", desired_kl=0.002, value_loss_coef=0.5, entropy_coef=0.03, num_learning_epochs=3, num_mini_batches=2,
                 selective_adaptation_module_loss=True, gamma=0.97, lam=0.9),
            dict(schedule="adaptive", desired_kl=0.05, learning_rate=3.e-4, adaptation_module_learning_rate=3.e-3, clip_param=0.3,
                 num_learning_epochs=1, num_mini_batches=5, max_grad_norm=10.0),
            dict(schedule="fixed", use_clipped_value_loss=True, value_loss_coef=2.0, num_learning_epochs=4, num_mini_batches=1,
                 num_adaptation_module_substeps=3, selective_adaptation_module_loss=True, lam=1.0)]
