"""Go1 overrides applied on top of `Cfg` (mirror of reference go1_gym/envs/go1/go1_config.py:8-106).

Table-driven: each section maps to the attribute values the reference assigns, in the same
order (later assignments win, e.g. `commands.lin_vel_x`)."""
from typing import Union

from params_proto import Meta

from go1_gym.envs.base.legged_robot_config import Cfg

_LEGS = ("FL", "RL", "FR", "RR")

_GO1 = [
    ("init_state", dict(
        pos=[0.0, 0.0, 0.34],
        default_joint_angles={
            **{f"{leg}_hip_joint": (0.1 if leg[1] == "L" else -0.1) for leg in _LEGS},
            **{f"{leg}_thigh_joint": (0.8 if leg[0] == "F" else 1.0) for leg in _LEGS},
            **{f"{leg}_calf_joint": -1.5 for leg in _LEGS},
        })),
    ("control", dict(control_type='P', stiffness={'joint': 20.}, damping={'joint': 0.5}, action_scale=0.25,
                     hip_scale_reduction=0.5, decimation=4)),
    ("asset", dict(file='{MINI_GYM_ROOT_DIR}/resources/robots/go1/urdf/go1.urdf', foot_name="foot",
                   penalize_contacts_on=["thigh", "calf"], terminate_after_contacts_on=["base"], self_collisions=0,
                   flip_visual_attachments=False, fix_base_link=False)),
    ("rewards", dict(soft_dof_pos_limit=0.9, base_height_target=0.34)),
    ("reward_scales", dict(torques=-0.0001, action_rate=-0.01, dof_pos_limits=-10.0, orientation=-5.,
                           base_height=-30.)),
    ("terrain", dict(mesh_type='trimesh', measure_heights=False, terrain_noise_magnitude=0.0, teleport_robots=True,
                     border_size=50, terrain_proportions=[0, 0, 0, 0, 0, 0, 0, 0, 1.0], curriculum=False)),
    ("env", dict(num_observations=42, observe_vel=False, num_envs=4000)),
    ("commands", dict(heading_command=False, resampling_time=10.0, command_curriculum=True, num_lin_vel_bins=30,
                      num_ang_vel_bins=30, lin_vel_x=[-0.6, 0.6], lin_vel_y=[-0.6, 0.6], ang_vel_yaw=[-1, 1])),
    ("domain_rand", dict(randomize_base_mass=True, added_mass_range=[-1, 3], push_robots=False, max_push_vel_xy=0.5,
                         randomize_friction=True, friction_range=[0.05, 4.5], randomize_restitution=True,
                         restitution_range=[0.0, 1.0], restitution=0.5, randomize_com_displacement=True,
                         com_displacement_range=[-0.1, 0.1], randomize_motor_strength=True,
                         motor_strength_range=[0.9, 1.1], randomize_Kp_factor=False, Kp_factor_range=[0.8, 1.3],
                         randomize_Kd_factor=False, Kd_factor_range=[0.5, 1.5], rand_interval_s=6)),
]


def config_go1(Cnfg: Union[Cfg, Meta]):
    for section, values in _GO1:
        target = getattr(Cnfg, section)
        for key, val in values.items():
            setattr(target, key, val)
