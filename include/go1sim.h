/* go1sim.h — C-ABI of the MI355X-native Go1 vectorised step (libgo1sim.so).
 *
 * This is the drop-in boundary for the hot path of Improbable-AI/walk-these-ways:
 * everything `LeggedRobot.step()` does between receiving `actions` and returning
 * (obs, privileged_obs, rew, reset) — i.e. what the reference delegates to the closed
 * Isaac Gym tensor API plus its own chain of small PyTorch kernels.  Each entry point
 * cites the reference interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - plain C, no torch types; all pointers in Go1SimBuffers are DEVICE pointers owned by
 *     the caller (PyTorch allocates them); the library never allocates per-env memory,
 *     never synchronises the stream, never throws.  Return value 0 = ok, <0 = error code.
 *   - SoA layout: an array documented as [C][N] stores component c of environment e at
 *     index c*N + e (environment index is the fastest dimension -> coalesced per-lane access).
 *     Arrays documented as (N,K) are row-major per environment (consumed by GEMMs).
 *   - quaternions are xyzw (go1_gym/envs/base/legged_robot_config.py:203).
 *   - all arithmetic is fp32 on device (reference: torch default / Isaac Gym float32 tensors).
 */
#ifndef GO1SIM_H
#define GO1SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO1SIM_ABI_VERSION 6

#define GO1_NUM_DOF 12
#define GO1_NUM_BODIES 17        /* base, then FL,FR,RL,RR x (hip, thigh, calf, foot) */
#define GO1_NUM_FEET 4
#define GO1_MAX_COMMANDS 15
#define GO1_MAX_REWARDS 24
#define GO1_MAX_OBS 96
#define GO1_MAX_PRIV_OBS 64
#define GO1_MAX_LAG 8            /* lag_timesteps + 1 <= 8 */
#define GO1_MAX_CATEGORIES 4
#define GO1_MAX_HEIGHT_AXIS 32   /* measured_points_x / _y entries */
#define GO1_MAX_CONTACTS 24      /* solver contacts per environment and substep (a Go1 lying on its side with every link down: 14-18) */
#define GO1_MAX_CURRICULUM_INTERVAL 64

/* Classes of the solver's contact list (priority order); contact_drop_counts[class] counts the active points of a class that
 * found no solver slot, contact_signature names the listed points (tests: which environments differ in their active set) */
enum Go1ContactClass {
  GO1_CC_FOOT = 0,        /* foot sphere on the terrain's top surface */
  GO1_CC_FOOT_WALL = 1,   /* foot sphere against a vertical terrain face */
  GO1_CC_SELF = 2,        /* body-body: lower legs / thighs of different legs, lower legs against the trunk */
  GO1_CC_TRUNK = 3,
  GO1_CC_CALF = 4,
  GO1_CC_THIGH = 5,
  GO1_CC_HIP = 6,
  GO1_CC_WALL = 7,        /* trunk / calf / thigh against a vertical terrain face */
  GO1_CC_COUNT = 8
};
#define GO1_SIG_WORDS 4          /* per substep: [top-surface points | wall + hip points | self pairs (3 bits per pair of legs: 1 + type of the
                                    listed capsule combination, pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) in bits 0..17; lower leg - trunk at
                                    bit 18 + leg; round 5: types 4 / 5 = hip capsule - lower leg) + legs with limit rows (bits 28..31) | hash of: the
                                    height-field cell and candidate point (corner / end) every listed terrain contact came from, the
                                    contacts that took the restitution branch, and the ACTIVE SET after every solver sweep (pressing
                                    contacts, contacts projected on the friction cone in that sweep, limit rows with an impulse)] */
#define GO1_SIG_MAX_SUBSTEPS 4

/* canonical reward ids: one per `_reward_*` in go1_gym/envs/rewards/corl_rewards.py:15-202 */
enum Go1RewardId {
  GO1_REW_TRACKING_LIN_VEL = 0,             /* corl_rewards.py:15 */
  GO1_REW_TRACKING_ANG_VEL = 1,             /* :20 */
  GO1_REW_LIN_VEL_Z = 2,                    /* :25 */
  GO1_REW_ANG_VEL_XY = 3,                   /* :29 */
  GO1_REW_ORIENTATION = 4,                  /* :33 */
  GO1_REW_TORQUES = 5,                      /* :37 */
  GO1_REW_DOF_ACC = 6,                      /* :41 */
  GO1_REW_ACTION_RATE = 7,                  /* :45 */
  GO1_REW_COLLISION = 8,                    /* :49,:143 */
  GO1_REW_DOF_POS_LIMITS = 9,               /* :54 */
  GO1_REW_JUMP = 10,                        /* :60 */
  GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE = 11, /* :67 */
  GO1_REW_TRACKING_CONTACTS_SHAPED_VEL = 12,   /* :77 */
  GO1_REW_DOF_POS = 13,                     /* :86 */
  GO1_REW_DOF_VEL = 14,                     /* :90 */
  GO1_REW_ACTION_SMOOTHNESS_1 = 15,         /* :94 */
  GO1_REW_ACTION_SMOOTHNESS_2 = 16,         /* :100 */
  GO1_REW_FEET_SLIP = 17,                   /* :107 */
  GO1_REW_FEET_CONTACT_VEL = 18,            /* :115 */
  GO1_REW_FEET_CONTACT_FORCES = 19,         /* :122 */
  GO1_REW_FEET_CLEARANCE_CMD_LINEAR = 20,   /* :127 */
  GO1_REW_FEET_IMPACT_VEL = 21,             /* :134 */
  GO1_REW_ORIENTATION_CONTROL = 22,         /* :148 */
  GO1_REW_RAIBERT_HEURISTIC = 23            /* :161 */
};

/* privileged-observation blocks, in the order legged_robot.py:383-488 concatenates them */
enum Go1PrivObsId {
  GO1_PRIV_FRICTION = 0,        /* :383 */
  GO1_PRIV_RESTITUTION = 1,     /* :405 */
  GO1_PRIV_BASE_MASS = 2,       /* :415 */
  GO1_PRIV_COM_DISPLACEMENT = 3,/* :423 */
  GO1_PRIV_MOTOR_STRENGTH = 4,  /* :434 */
  GO1_PRIV_MOTOR_OFFSET = 5,    /* :444 */
  GO1_PRIV_BODY_HEIGHT = 6,     /* :454 */
  GO1_PRIV_BODY_VELOCITY = 7,   /* :464 */
  GO1_PRIV_GRAVITY = 8,         /* :474 */
  GO1_PRIV_CLOCK_INPUTS = 9,    /* :482 */
  GO1_PRIV_DESIRED_CONTACT = 10,/* :486 */
  GO1_PRIV_COUNT = 11
};

/* Fault word: one bit per site that can make the simulation state non-finite.  The step kernel ORs the bits of an
 * environment into fault_flags[e] (sticky until the caller clears it) and counts every occurrence in
 * fault_counts[bit].  PhysX never hands back a non-finite state (legged_robot.py:76-80 reads it unchecked), so a
 * non-zero word is a defect of THIS simulator, not a condition of the reference; the kernel contains it (the episode
 * ends, nothing non-finite leaves the launch) and reports it here instead of hiding it. */
enum Go1FaultBit {
  GO1_FAULT_STATE_IN = 0,       /* root / joint state already non-finite when the step started */
  GO1_FAULT_TORQUE = 1,         /* torque model output non-finite before the clip */
  GO1_FAULT_BASE_PIVOT = 2,     /* Cholesky pivot of the base articulated inertia <= 1e-9 or non-finite */
  GO1_FAULT_JOINT_D = 3,        /* joint-space articulated inertia D_j <= 1e-9 or non-finite */
  GO1_FAULT_CONTACT_FRAME = 4,  /* contact normal parallel to the x axis: tangent basis undefined */
  GO1_FAULT_W_DIAG = 5,         /* Delassus diagonal of an active contact <= 1e-9 or non-finite */
  GO1_FAULT_LAMBDA = 6,         /* contact impulse non-finite after the PGS sweeps */
  GO1_FAULT_STATE_OUT = 7,      /* root / joint state non-finite after a substep's integration */
  GO1_FAULT_REWARD = 8,         /* a reward term or the total was non-finite (counted as 0, episode ended) */
  GO1_FAULT_OBS = 9,            /* an observation column was non-finite (written as 0) */
  GO1_FAULT_CONTACT_DROPPED = 10, /* more active contact points than solver slots: the excess was not solved (count only) */
  GO1_FAULT_LIMIT_SAFETY = 11,  /* a joint left its limit rows' band by more than the safety factor and was cut (count only) */
  GO1_FAULT_BITS = 16
};
#define GO1_FAULT_FATAL_MASK 0x3FFu   /* bits 0..9: "simulation failed"; CONTACT_DROPPED and LIMIT_SAFETY are count-only */

/* Everything that the reference reads from `Cfg` on the hot path, flattened.
 * Filled by the host mirror of LeggedRobot._parse_cfg/_init_buffers
 * (walk-these-ways_amd/go1_gym/envs/base/legged_robot.py). */
typedef struct Go1SimConfig {
  int32_t abi_version;
  int32_t num_envs;
  uint64_t seed;                   /* Philox key; RNG is counter-based: (seed; env, step, purpose, block) */
  int32_t env_id_offset;           /* global id of local env 0 (multi-GPU sharding keeps streams rank-independent) */

  /* --- control: legged_robot.py:907-946, go1_config.py:29-37 --- */
  int32_t decimation;              /* Cfg.control.decimation */
  float sim_dt;                    /* Cfg.sim.dt (float32) */
  int32_t control_type;            /* 0 = 'P', 1 = 'actuator_net' */
  float action_scale;
  float hip_scale_reduction;
  float clip_actions;
  float kp, kd;                    /* 'P' gains (stiffness/damping['joint']) */
  int32_t use_lag;                 /* Cfg.domain_rand.randomize_lag_timesteps */
  int32_t lag_timesteps;           /* Cfg.domain_rand.lag_timesteps */
  float default_dof_pos[GO1_NUM_DOF];
  float torque_limits[GO1_NUM_DOF];
  float dof_pos_soft_lower[GO1_NUM_DOF];  /* legged_robot.py:603-607 */
  float dof_pos_soft_upper[GO1_NUM_DOF];

  /* --- physics (replaces gym.simulate; parameters legged_robot_config.py:402-421) --- */
  float gravity[3];                /* nominal, (0,0,-9.8): legged_robot.py:558 */
  float contact_distance;          /* contact generation distance (2*contact_offset) */
  float max_depenetration_velocity;
  float bounce_threshold_velocity;
  float terrain_friction;          /* Cfg.terrain.static_friction */
  float terrain_dynamic_friction;  /* Cfg.terrain.dynamic_friction (legged_robot.py:1437,1454,1474): the cone of a SLIDING contact */
  float terrain_restitution;
  float max_linear_velocity, max_angular_velocity;   /* Cfg.asset.max_*_velocity (1000): magnitude caps on the base twist */
  float joint_limit_margin;        /* rad/s, rad: a joint's limit row enters the solver when its free rate comes within      */
  float joint_limit_pos_margin;    /* joint_limit_margin of +-vmax, or would carry it to within joint_limit_pos_margin of a stop  */
  int32_t self_collision;          /* 1: lower legs collide with each other and with the trunk (asset self_collisions = 0 in the
                                      reference means "no pair filtered": enabled, go1_config.py:44, legged_robot.py:1563-1564) */
  int32_t solver_iterations;       /* PGS sweeps per substep */
  int32_t warm_start;              /* start PGS from last substep's impulses */
  int32_t terrain_type;            /* 0 plane, 1 height field */
  int32_t hf_rows, hf_cols;        /* height_samples shape */
  float hf_hscale, hf_vscale, hf_border;
  int32_t hf_wall_units;           /* > 0: two neighbouring height samples differing by MORE than this many units are the ends of a
                                      vertical face — the `trimesh` terrain's slope_treshold * horizontal_scale / vertical_scale
                                      (terrain.py:33-36), compared on the int16 samples as the reference does; 0: plain bilinear field */
  /* height scan: legged_robot.py:1756-1806 (_init_height_points, _get_heights) */
  int32_t measure_heights;         /* Cfg.terrain.measure_heights */
  int32_t num_height_x, num_height_y;
  float height_points_x[GO1_MAX_HEIGHT_AXIS], height_points_y[GO1_MAX_HEIGHT_AXIS];
  int32_t observe_heights;         /* append clip(z_base - 0.5 - h, -1, 1) * scale to the observation (BASELINE config 3) */
  float obs_scale_height, height_noise_scale;

  /* --- episode / domain randomisation cadence: legged_robot.py:675-708,1716-1732 --- */
  int32_t max_episode_length;
  int32_t resample_interval;       /* int(resampling_time / dt) */
  int32_t rand_interval;           /* ceil(rand_interval_s / dt) */
  int32_t randomize_gravity;
  int32_t gravity_rand_interval, gravity_rand_duration;
  float gravity_range[2];
  int32_t push_robots, push_interval;
  float max_push_vel_xy;
  int32_t randomize_motor_strength, randomize_motor_offset, randomize_Kp_factor, randomize_Kd_factor;
  float motor_strength_range[2], motor_offset_range[2], Kp_factor_range[2], Kd_factor_range[2];
  /* re-draw of the rigid-body properties together with the DOF properties (legged_robot.py:706-708,611-633) */
  int32_t randomize_rigids_after_start;
  int32_t randomize_base_mass, randomize_com_displacement, randomize_friction, randomize_restitution;
  float added_mass_range[2], com_displacement_range[2], friction_range[2], restitution_range[2];
  int32_t teleport_robots;
  float teleport_thresh, teleport_x_offset, terrain_length, terrain_width;
  int32_t terrain_num_rows, terrain_num_cols;

  /* --- reset distribution: legged_robot.py:948-1001 --- */
  float base_init_state[13];
  int32_t custom_origins;
  float x_init_range, y_init_range, yaw_init_range, x_init_offset, y_init_offset;

  /* --- termination: legged_robot.py:138-148 --- */
  uint32_t termination_body_mask;  /* bit b: body b in termination_contact_indices */
  uint32_t penalised_body_mask;    /* penalised_contact_indices */
  int32_t use_terminal_body_height;
  float terminal_body_height;

  /* --- observations: legged_robot.py:302-376,1053-1120 --- */
  int32_t num_obs, num_privileged_obs, num_obs_history; /* history length H (rows of num_obs) */
  int32_t num_commands;
  int32_t observe_command, observe_two_prev_actions, observe_timing_parameter, observe_clock_inputs;
  int32_t observe_vel, observe_only_ang_vel, observe_only_lin_vel, observe_yaw, observe_contact_states;
  int32_t global_reference;
  int32_t observe_gait_commands, pacing_offset;
  float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel;
  float commands_scale[GO1_MAX_COMMANDS];
  int32_t add_noise;
  float noise_scale_vec[GO1_MAX_OBS];
  float clip_observations;
  int32_t priv_enabled[GO1_PRIV_COUNT];
  float priv_scale[GO1_PRIV_COUNT], priv_shift[GO1_PRIV_COUNT];  /* get_scale_shift, math_utils.py:35-38 */

  /* --- rewards: legged_robot.py:263-300,1385-1429 --- */
  int32_t num_rewards;             /* active terms, in reward_scales dict order */
  int32_t reward_ids[GO1_MAX_REWARDS];
  float reward_scales[GO1_MAX_REWARDS];  /* already multiplied by dt (legged_robot.py:1400) */
  int32_t only_positive_rewards, only_positive_rewards_ji22_style;
  float sigma_rew_neg;
  float dt;                        /* policy dt = decimation * sim_dt */
  float tracking_sigma, tracking_sigma_yaw, base_height_target, max_contact_force;
  float kappa_gait_probs, gait_force_sigma, gait_vel_sigma;
  int32_t reward_heights_above_terrain;  /* 0 (reference): _reward_feet_clearance_cmd_linear / _reward_feet_contact_vel / _reward_jump read WORLD
                                      z (corl_rewards.py:129 `# - reference_heights`, :100, :52-53: written for the flat z = 0 ground of
                                      scripts/train.py).  1 (NOT in the reference; Cfg.rewards.heights_above_terrain): foot heights above the
                                      terrain sample under the foot, base height above the mean of the height scan (or the sample under the
                                      base) — sample convention of _get_heights, legged_robot.py:1793-1806.  What BASELINE configs[2] needs to
                                      have a non-zero reward off the z = 0 tiles. */

  /* --- commands / curriculum: legged_robot.py:710-824, curriculum.py --- */
  int32_t device_curriculum;       /* 1: in-kernel sampling; 0: kernel only raises resample flags */
  int32_t curriculum_update_interval; /* K >= 1: the successes of step t are logged in slot t % K of curriculum_success and the K per-step
                                      weight updates are applied, in order, after every K-th step (K = 1: after every step, the
                                      reference's cadence).  Environments sharded over ranks exchange the K slots in ONE all-reduce */
  int32_t defer_curriculum_update; /* 1: go1sim_step leaves the weight update to an explicit go1sim_curriculum_update call, so that the
                                      caller can all-reduce curriculum_success over the ranks first (envs sharded over GPUs) */
  int32_t num_categories;          /* 4 when gaitwise_curricula (pronk, trot, pace, bound) else 1 */
  int32_t gaitwise_curricula, binary_phases;
  int32_t exclusive_phase_offset, balance_gait_distribution;   /* legged_robot.py:783-812 (only without gaitwise_curricula) */
  int32_t num_bins;                /* product of num_bins_* */
  int32_t grid_bins[GO1_MAX_COMMANDS];
  float grid_low[GO1_MAX_COMMANDS], grid_high[GO1_MAX_COMMANDS];  /* limit_* ranges */
  int32_t curriculum_keys;         /* bit k: key k of {tracking_lin_vel, tracking_ang_vel,
                                      tracking_contacts_shaped_force, tracking_contacts_shaped_vel} present */
  int32_t curriculum_sum_index[4]; /* index into command_sums for those keys */
  float curriculum_threshold[4];   /* curriculum_thresholds[key] * reward_scales[key] */
} Go1SimConfig;

/* Device buffers (caller-owned).  [C][N] = SoA, (N,K) = row-major.  */
typedef struct Go1SimBuffers {
  /* physics state — replaces the Isaac Gym tensors wrapped at legged_robot.py:1127-1157 */
  float* root_states;              /* [13][N] pos3 quat4 linvel3 angvel3, world frame */
  float* dof_pos;                  /* [12][N] */
  float* dof_vel;                  /* [12][N] */
  float* contact_forces;           /* [17*3][N] net contact force per body (x,y,z), last substep, world; also the solver's warm start */
  float* foot_positions;           /* [4*3][N] */
  float* foot_velocities;          /* [4*3][N] */
  float* prev_foot_velocities;     /* [4*3][N] legged_robot.py:72 */
  /* torque model state — legged_robot.py:907-946,1154,1255-1258 */
  float* lag_buffer;               /* [lag+1][12][N] ring, logical entry i at slot (head+i)%(lag+1) */
  float* joint_pos_err_last;       /* [12][N] */
  float* joint_pos_err_last_last;  /* [12][N] */
  float* joint_vel_last;           /* [12][N] */
  float* joint_vel_last_last;      /* [12][N] */
  float* joint_pos_target;         /* [12][N] */
  float* last_joint_pos_target;    /* [12][N] */
  float* last_last_joint_pos_target; /* [12][N] */
  float* actions;                  /* [12][N] clipped actions of the current step */
  float* last_actions;             /* [12][N] */
  float* last_last_actions;        /* [12][N] */
  float* last_dof_vel;             /* [12][N] */
  float* torques;                  /* [12][N] */
  /* derived state — legged_robot.py:106-115 */
  float* base_lin_vel;             /* [3][N] */
  float* base_ang_vel;             /* [3][N] */
  float* projected_gravity;        /* [3][N] */
  /* gait / commands — legged_robot.py:826-905 */
  float* commands;                 /* [15][N] */
  float* gait_indices;             /* [N] */
  float* clock_inputs;             /* [4][N] */
  float* desired_contact_states;   /* [4][N] */
  float* foot_indices;             /* [4][N] */
  /* episode bookkeeping */
  int32_t* episode_length_buf;     /* [N] */
  uint8_t* reset_buf;              /* [N] */
  uint8_t* time_out_buf;           /* [N] */
  uint8_t* last_contacts;          /* [4][N] corl_rewards.py:107-113 */
  uint8_t* resample_flags;         /* [N] bit0: callback resample, bit1: reset resample (host curriculum backend) */
  float* rew_buf;                  /* [N] */
  float* episode_sums;             /* [num_rewards+1][N], last row = "total" */
  float* command_sums;             /* [num_rewards+5][N]: +lin_vel_raw, ang_vel_raw, lin_vel_residual, ang_vel_residual, ep_timesteps */
  float* episode_log;              /* [num_rewards+1 +1]: running sum over reset TRAIN envs of episode_sums, last = count;
                                      accumulated atomically by step/reset_idx, zeroed by the caller when consumed */
  float* episode_sums_eval;        /* [num_rewards+1][N] or NULL: an evaluation environment's episode sums at the end of its first
                                      episode after the caller set the entries to -1 (legged_robot.py:188-195, 1420-1424) */
  /* domain randomisation parameters — legged_robot.py:1260-1288 */
  float* friction_coeffs;          /* [N] */
  float* restitutions;             /* [N] */
  float* payloads;                 /* [N] */
  float* com_displacements;        /* [3][N] */
  float* motor_strengths;          /* [12][N] */
  float* motor_offsets;            /* [12][N] */
  float* Kp_factors;               /* [12][N] */
  float* Kd_factors;               /* [12][N] */
  float* env_origins;              /* [3][N] */
  /* command curriculum (device backend) */
  int32_t* env_command_bins;       /* [N] */
  int32_t* env_command_categories; /* [N] */
  float* curriculum_weights;       /* [num_categories][num_bins] */
  float* curriculum_cdf;           /* [num_categories][num_bins] normalised inclusive prefix sums */
  int32_t* curriculum_success;     /* [curriculum_update_interval][num_categories][num_bins] successes per step slot */
  const int32_t* curriculum_nbr_ptr; /* [num_bins+1] CSR of get_local_bins neighbourhoods (incl. self) */
  const int32_t* curriculum_nbr_idx;
  /* outputs consumed by the policy */
  float* obs_buf;                  /* (N, num_obs) */
  float* privileged_obs_buf;       /* (N, num_privileged_obs) */
  float* obs_history;              /* (N, 2*(H+1)*num_obs): ring of H+1 slots stored twice; window via go1sim_history_window_offset */
  /* fault reporting (enum Go1FaultBit) */
  uint32_t* fault_flags;           /* [N] sticky OR of the fault bits of each environment; cleared by the caller */
  uint32_t* fault_counts;          /* [GO1_FAULT_BITS] occurrences per bit since the caller last zeroed it */
  uint32_t* contact_drop_counts;   /* [GO1_CC_COUNT] or NULL: active contact points without a solver slot, per class */
  uint32_t* contact_signature;     /* [GO1_SIG_MAX_SUBSTEPS * GO1_SIG_WORDS][N] or NULL (tests): the listed contact points, self pairs and
                                      limit-row legs of each substep of the last step, bit per candidate (csrc/go1_physics.h) */
  /* terrain */
  const int16_t* height_samples;   /* (hf_rows, hf_cols) or NULL for plane */
  float* measured_heights;         /* [num_height_x*num_height_y][N] or NULL */
} Go1SimBuffers;

typedef struct Go1Sim Go1Sim;      /* opaque handle */

/* Create a simulator instance bound to caller-owned device buffers.
 * Replaces: gym.create_sim/prepare_sim + acquire_*_tensor (base_task.py:71-72, legged_robot.py:1127-1157). */
int go1sim_create(const Go1SimConfig* cfg, const Go1SimBuffers* buffers, int device, Go1Sim** out);
int go1sim_destroy(Go1Sim* sim);

/* Update config fields that scripts mutate after construction (e.g. play.py changes ranges).  Replaces the TRAIN block only:
 * with an evaluation split installed, follow it with go1sim_set_eval_config(new cfg + the group-dispatched fields) so that the
 * evaluation block does not keep the old rewards / commands / control / physics (the host mirror does: Go1Sim.set_config). */
int go1sim_set_config(Go1Sim* sim, const Go1SimConfig* cfg);

/* Train / evaluation split (reference `eval_cfg`: base_task.py:43-49, legged_robot.py:531-544 `_call_train_eval`): the
 * environments [num_train_envs, num_envs) run under `eval_cfg` wherever the reference dispatches on the environment's group
 * — domain-randomisation ranges and switches, pushes, teleports, the reset distribution — and are kept out of
 * `episode_log`.  `eval_cfg` must describe the same simulation (num_envs = the TOTAL, same dimensions, control, physics,
 * rewards and command distribution: the caller copies those from the train configuration); the kernels select the block
 * per wavefront, so num_train_envs must be a multiple of 16 (-3 otherwise).  num_train_envs == num_envs: no split. */
int go1sim_set_eval_config(Go1Sim* sim, const Go1SimConfig* eval_cfg, int32_t num_train_envs);

/* One policy step for all environments: clip actions, `decimation` x {torque model, physics substep},
 * derived state, gait clock, command resampling, DR cadence, termination, rewards, in-kernel reset,
 * observations (+noise, clip), history append.
 * Replaces: LeggedRobot.step (legged_robot.py:60-88) incl. gym.set_dof_actuation_force_tensor /
 * simulate / fetch_results / refresh_* (:76-80,:95-97), post_physics_step (:90-136) and
 * HistoryWrapper's concat (history_wrapper.py:23).
 * `actions`: device pointer, row-major (N,12) fp32 as produced by the policy.  `stream`: hipStream_t. */
int go1sim_step(Go1Sim* sim, const float* actions, void* stream);

/* Reset the listed environments (device int32 ids) exactly like reset_idx (legged_robot.py:150-239)
 * without stepping.  ids == NULL resets all. */
int go1sim_reset_idx(Go1Sim* sim, const int32_t* ids, int32_t n, void* stream);

/* Piecewise entry points (each is a sub-range of go1sim_step; used by the parity tests). */
int go1sim_compute_torques(Go1Sim* sim, const float* actions_soa, void* stream);   /* legged_robot.py:907-946 */
int go1sim_physics_substep(Go1Sim* sim, void* stream);                             /* one gym.simulate, :76-80 */
int go1sim_curriculum_update(Go1Sim* sim, void* stream);                           /* curriculum.py:135-154 */
/* tensor maps of post_physics_step only (legged_robot.py:90-136) on the state in the buffers; `gravity` = host
 * pointer to the 3 floats of the gravity vector in force during the step.  Advances common_step_counter. */
int go1sim_post_physics(Go1Sim* sim, const float* gravity, void* stream);
/* HistoryWrapper.get_observations' side effect (history_wrapper.py:26-30): append obs_buf to the history once more. */
int go1sim_append_history(Go1Sim* sim, void* stream);
/* Column (in floats) where the reference-ordered (oldest first) obs-history window starts inside each obs_history row. */
int go1sim_history_window_offset(Go1Sim* sim, int32_t* offset_floats);

/* Step counter (legged_robot.py:103 common_step_counter) and lag ring head live in the handle. */
int go1sim_get_counters(Go1Sim* sim, int64_t* common_step_counter, int32_t* lag_head);
int go1sim_set_counters(Go1Sim* sim, int64_t common_step_counter, int32_t lag_head);

/* Per-launch timing of the step kernel with HIP events recorded on the launch stream (bench.py roofline leg).
 * enable_timing(capacity): keep event pairs for the last `capacity` go1sim_step launches (0 disables).
 * read_timings: synchronises on the recorded events and returns up to `max` durations in ms (oldest first). */
int go1sim_enable_timing(Go1Sim* sim, int capacity);
int go1sim_read_timings(Go1Sim* sim, float* ms, int32_t max, int32_t* count);

const char* go1sim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GO1SIM_H */
