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
    # north_star's "domain-randomisation pushes" and the other step-callback branches that train.py leaves off: velocity
    # pushes (legged_robot.py:1017-1026), edge teleport (:1028-1051), re-drawn rigid-body properties (:706-708,166-168);
    # kernel-vs-oracle only (no fixture: the reference draws these from torch's global RNG)
    "dr": {
        "domain_rand": dict(push_robots=True, push_interval_s=0.25, max_push_vel_xy=0.8, randomize_rigids_after_start=True,
                            rand_interval_s=0.15, randomize_gravity=True),
        "terrain": dict(teleport_robots=True, teleport_thresh=0.4),
    },
}


def apply_variant(Cfg, name):
    for section, values in VARIANTS[name].items():
        target = getattr(Cfg, section)
        for k, v in values.items():
            setattr(target, k, v)
    return Cfg
