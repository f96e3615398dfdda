"""The effective configuration of the reference's headline training run, as data.

`apply_train_config(Cfg)` = `config_go1(Cfg)` followed by the attribute assignments that the
reference's scripts/train.py:19-205 performs before constructing the env (BASELINE.json configs
2/4/5 are quoted on this configuration).  Kept as one table so bench.py, the
tests and smoke() can build the flagship env without the reference tree being present.
"""
from go1_gym.envs.go1.go1_config import config_go1

TRAIN_OVERRIDES = {
    "commands": dict(
        num_lin_vel_bins=30, num_ang_vel_bins=30, distributional_commands=True, resampling_time=10, num_commands=15,
        lin_vel_x=[-1.0, 1.0], lin_vel_y=[-0.6, 0.6], ang_vel_yaw=[-1.0, 1.0], body_height_cmd=[-0.25, 0.15],
        gait_frequency_cmd_range=[2.0, 4.0], gait_phase_cmd_range=[0.0, 1.0], gait_offset_cmd_range=[0.0, 1.0],
        gait_bound_cmd_range=[0.0, 1.0], gait_duration_cmd_range=[0.5, 0.5], footswing_height_range=[0.03, 0.35],
        body_pitch_range=[-0.4, 0.4], body_roll_range=[-0.0, 0.0], stance_width_range=[0.10, 0.45],
        stance_length_range=[0.35, 0.45],
        limit_vel_x=[-5.0, 5.0], limit_vel_y=[-0.6, 0.6], limit_vel_yaw=[-5.0, 5.0], limit_body_height=[-0.25, 0.15],
        limit_gait_frequency=[2.0, 4.0], limit_gait_phase=[0.0, 1.0], limit_gait_offset=[0.0, 1.0],
        limit_gait_bound=[0.0, 1.0], limit_gait_duration=[0.5, 0.5], limit_footswing_height=[0.03, 0.35],
        limit_body_pitch=[-0.4, 0.4], limit_body_roll=[-0.0, 0.0], limit_stance_width=[0.10, 0.45],
        limit_stance_length=[0.35, 0.45],
        num_bins_vel_x=21, num_bins_vel_y=1, num_bins_vel_yaw=21, num_bins_body_height=1, num_bins_gait_frequency=1,
        num_bins_gait_phase=1, num_bins_gait_offset=1, num_bins_gait_bound=1, num_bins_gait_duration=1,
        num_bins_footswing_height=1, num_bins_body_roll=1, num_bins_body_pitch=1, num_bins_stance_width=1,
        exclusive_phase_offset=False, pacing_offset=False, binary_phases=True, gaitwise_curricula=True,
    ),
    "curriculum_thresholds": dict(tracking_ang_vel=0.7, tracking_lin_vel=0.8, tracking_contacts_shaped_vel=0.90,
                                  tracking_contacts_shaped_force=0.90),
    "control": dict(control_type="actuator_net"),
    "domain_rand": dict(
        lag_timesteps=6, randomize_lag_timesteps=True, randomize_rigids_after_start=False,
        randomize_friction_indep=False, randomize_friction=True, friction_range=[0.1, 3.0],
        randomize_restitution=True, restitution_range=[0.0, 0.4], randomize_base_mass=True,
        added_mass_range=[-1.0, 3.0], randomize_gravity=True, gravity_range=[-1.0, 1.0],
        gravity_rand_interval_s=8.0, gravity_impulse_duration=0.99, randomize_com_displacement=False,
        com_displacement_range=[-0.15, 0.15], randomize_ground_friction=True, ground_friction_range=[0.0, 0.0],
        randomize_motor_strength=True, motor_strength_range=[0.9, 1.1], randomize_motor_offset=True,
        motor_offset_range=[-0.02, 0.02], push_robots=False, randomize_Kp_factor=False, randomize_Kd_factor=False,
        rand_interval_s=4, tile_height_range=[-0.0, 0.0], tile_height_curriculum=False,
        tile_height_update_interval=1000000, tile_height_curriculum_step=0.01,
    ),
    "env": dict(
        priv_observe_motion=False, priv_observe_gravity_transformed_motion=False, priv_observe_friction_indep=False,
        priv_observe_friction=True, priv_observe_restitution=True, priv_observe_base_mass=False,
        priv_observe_gravity=False, priv_observe_com_displacement=False, priv_observe_ground_friction=False,
        priv_observe_ground_friction_per_foot=False, priv_observe_motor_strength=False,
        priv_observe_motor_offset=False, priv_observe_Kp_factor=False, priv_observe_Kd_factor=False,
        priv_observe_body_velocity=False, priv_observe_body_height=False, priv_observe_desired_contact_states=False,
        priv_observe_contact_forces=False, priv_observe_foot_displacement=False,
        priv_observe_gravity_transformed_foot_displacement=False,
        num_privileged_obs=2, num_observation_history=30, observe_two_prev_actions=True, observe_yaw=False,
        num_observations=70, num_scalar_observations=70, observe_gait_commands=True, observe_timing_parameter=False,
        observe_clock_inputs=True,
    ),
    "terrain": dict(
        border_size=0.0, mesh_type="trimesh", num_cols=30, num_rows=30, terrain_width=5.0, terrain_length=5.0,
        x_init_range=0.2, y_init_range=0.2, teleport_thresh=0.3, teleport_robots=False, center_robots=True,
        center_span=4, horizontal_scale=0.10, yaw_init_range=3.14,
    ),
    "rewards": dict(
        use_terminal_foot_height=False, use_terminal_body_height=True, terminal_body_height=0.05,
        use_terminal_roll_pitch=True, terminal_body_ori=1.6, base_height_target=0.30, kappa_gait_probs=0.07,
        gait_force_sigma=100., gait_vel_sigma=10., reward_container_name="CoRLRewards", only_positive_rewards=False,
        only_positive_rewards_ji22_style=True, sigma_rew_neg=0.02,
    ),
    "reward_scales": dict(
        feet_contact_forces=0.0, feet_slip=-0.04, action_smoothness_1=-0.1, action_smoothness_2=-0.1, dof_vel=-1e-4,
        dof_pos=-0.0, jump=10.0, base_height=0.0, estimation_bonus=0.0, raibert_heuristic=-10.0,
        feet_impact_vel=-0.0, feet_clearance=-0.0, feet_clearance_cmd=-0.0, feet_clearance_cmd_linear=-30.0,
        orientation=0.0, orientation_control=-5.0, tracking_stance_width=-0.0, tracking_stance_length=-0.0,
        lin_vel_z=-0.02, ang_vel_xy=-0.001, feet_air_time=0.0, hop_symmetry=0.0,
        tracking_contacts_shaped_force=4.0, tracking_contacts_shaped_vel=4.0, collision=-5.0,
    ),
    "normalization": dict(friction_range=[0, 1], ground_friction_range=[0, 1], clip_actions=10.0),
}


def apply_train_config(Cfg, num_envs=None):
    config_go1(Cfg)
    for section, values in TRAIN_OVERRIDES.items():
        target = getattr(Cfg, section)
        for key, val in values.items():
            setattr(target, key, val)
    if num_envs is not None:
        Cfg.env.num_envs = num_envs
    return Cfg
