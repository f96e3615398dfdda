// go1_maps.h — tensor maps of post_physics_step, commands/curriculum sampling, domain randomisation, reset
// (reference legged_robot.py:90-136,138-239,263-491,675-905; corl_rewards.py:15-202).  One environment per calling lane.
#pragma once
#include "go1_math.h"
#include "../../include/go1sim.h"

// The configuration, the buffer table and the derived tables live in ONE device struct that no kernel ever writes.
// It is addressed through the constant address space: loads from it are invariant for the compiler — scalar
// (s_load), batched and hoisted across the kernels' stores and atomics.  Through a generic pointer every store would
// force the next configuration field or buffer pointer to be re-fetched from memory (one exposed round trip each,
// with a single wave per SIMD).
#ifndef GO1_CONSTANT
#define GO1_CONSTANT __attribute__((address_space(4)))
#endif
typedef const GO1_CONSTANT Go1SimConfig& CfgRef;
typedef const GO1_CONSTANT Go1SimBuffers& BufRef;

#define PI_F 3.14159265358979323846f
enum { P_NOISE = 1, P_RESET = 2, P_DOFPROPS_CB = 3, P_DOFPROPS_RESET = 4, P_CMD_CB = 5, P_CMD_RESET = 6, P_PUSH = 7, P_GRAVITY = 8, P_RIGID = 9, P_RIGID_RESET = 10 };

#define AT(ptr, c, e) ((ptr)[(size_t)(c) * N + (e)])

__device__ __noinline__ float rng_uniform(CfgRef cfg, uint32_t env_global, int64_t step, uint32_t purpose, uint32_t idx) {
  uint32_t out[4];
  philox4x32_10(env_global, (uint32_t)step, purpose, idx >> 2, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), out);
  uint32_t sel = idx & 3;
  uint32_t v = sel == 0 ? out[0] : sel == 1 ? out[1] : sel == 2 ? out[2] : out[3];
  return u32_to_unit(v);
}

DEV V3 gravity_at(CfgRef cfg, int64_t t) {
  V3 g = v3(cfg.gravity[0], cfg.gravity[1], cfg.gravity[2]);
  if (!cfg.randomize_gravity) return g;
  int64_t epoch = t / cfg.gravity_rand_interval, ph = t % cfg.gravity_rand_interval;
  if (ph >= cfg.gravity_rand_duration) return g;
  float span = cfg.gravity_range[1] - cfg.gravity_range[0];
  g.x += rng_uniform(cfg, 0xFFFFFFFFu, epoch, P_GRAVITY, 0) * span + cfg.gravity_range[0];
  g.y += rng_uniform(cfg, 0xFFFFFFFFu, epoch, P_GRAVITY, 1) * span + cfg.gravity_range[0];
  g.z += rng_uniform(cfg, 0xFFFFFFFFu, epoch, P_GRAVITY, 2) * span + cfg.gravity_range[0];
  return g;
}

// ================================================================================================
// commands, domain randomisation, reset
// ================================================================================================
DEV float fmod1(float x) { float r = fmodf(x, 1.0f); return r < 0.f ? r + 1.0f : r; }

// Where the sampling functions below take their uniform draws from.  RngDirect: generated where they are used, one Philox call per draw
// (rng_uniform).  RngTable: read from a table the environment's four lanes filled together (reset_rng_table) — the reset of an environment
// draws 60 numbers from 17 generator blocks, and as 60 calls on one lane they were more than half of the ~80 k cycles a reset added to its
// workgroup's step (and the step kernel's launch lasts as long as its slowest workgroup: with random episode lengths some environment of the
// 4096 resets in nearly every step).  The same (environment, step, purpose, index) -> number map either way.
struct RngDirect {
  CfgRef cfg; uint32_t eg; int64_t step;
  DEV float operator()(uint32_t purpose, uint32_t idx) const { return rng_uniform(cfg, eg, step, purpose, idx); }
};
enum { RT_CMD = 0, RT_DOF = 20, RT_RIGID = 36, RT_RESET = 44, RT_END = 68, RT_BLOCKS = RT_END / 4 };      // table layout: block-aligned runs per purpose
static_assert(RT_END <= GO1_MAX_OBS, "the reset's table of draws lives in the environment's observation staging row");
struct RngTable {
  const float* t;
  DEV float operator()(uint32_t purpose, uint32_t idx) const {
    return t[(purpose == P_CMD_RESET ? RT_CMD : purpose == P_DOFPROPS_RESET ? RT_DOF : purpose == P_RIGID_RESET ? RT_RIGID : RT_RESET) + idx];
  }
};
// the four lanes of an environment (leg = 0 .. 3) fill its table: block b by lane b & 3.  Indices used: P_CMD_RESET 0 .. 17, P_DOFPROPS_RESET
// 0 .. 14, P_RIGID_RESET 0 .. 5, P_RESET 0 .. 20 (resample_commands / randomize_* / reset_env below)
DEV void reset_rng_table(CfgRef cfg, uint32_t eg, int64_t step, float* t, int leg) {
#pragma unroll 1
  for (int b = leg; b < RT_BLOCKS; b += 4) {
    const uint32_t purpose = 4 * b < RT_DOF ? P_CMD_RESET : 4 * b < RT_RIGID ? P_DOFPROPS_RESET : 4 * b < RT_RESET ? P_RIGID_RESET : P_RESET;
    const uint32_t blk = 4 * b < RT_DOF ? b : 4 * b < RT_RIGID ? b - RT_DOF / 4 : 4 * b < RT_RESET ? b - RT_RIGID / 4 : b - RT_RESET / 4;
    uint32_t out[4];
    philox4x32_10(eg, (uint32_t)step, purpose, blk, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), out);
#pragma unroll
    for (int i = 0; i < 4; i++) t[4 * b + i] = u32_to_unit(out[i]);
  }
}

// slot_step: the policy step the success bookkeeping belongs to (oracle resample_commands()): slot slot_step % curriculum_update_interval
template <class Rng>
DEV void resample_commands_t(CfgRef cfg, BufRef B, int e, int N, uint32_t purpose, int64_t slot_step, const Rng& rng) {
  if (cfg.device_curriculum) {
    const int ep_len = cfg.max_episode_length < cfg.resample_interval ? cfg.max_episode_length : cfg.resample_interval;
    bool ok = cfg.curriculum_keys != 0;
#pragma unroll 1
    for (int kx = 0; kx < 4; kx++) {
      if (!(cfg.curriculum_keys & (1 << kx))) continue;
      float val = AT(B.command_sums, cfg.curriculum_sum_index[kx], e) / (float)ep_len;
      if (!(val > cfg.curriculum_threshold[kx])) ok = false;
    }
    int cat_old = B.env_command_categories[e], bin_old = B.env_command_bins[e];
    if (ok) atomicAdd(&B.curriculum_success[((int)(slot_step % cfg.curriculum_update_interval) * cfg.num_categories + cat_old) * cfg.num_bins + bin_old], 1);
    float u0 = rng(purpose, 0), u1 = rng(purpose, 1);
    int cat = (int)(u0 * cfg.num_categories);
    if (cat >= cfg.num_categories) cat = cfg.num_categories - 1;
    const float* cdf = B.curriculum_cdf + (size_t)cat * cfg.num_bins;
    // first bin with u1 < cdf[bin] (cdf is non-decreasing): binary search instead of the oracle's linear scan
    int lo = 0, hi = cfg.num_bins - 1;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (u1 < cdf[mid]) hi = mid; else lo = mid + 1;
    }
    const int bin = lo;
    B.env_command_bins[e] = bin;
    B.env_command_categories[e] = cat;
    int rem = bin;
    float cmd[GO1_MAX_COMMANDS];
#pragma unroll
    for (int kx = GO1_MAX_COMMANDS - 1; kx >= 0; kx--) {
      int nb = cfg.grid_bins[kx], idx = rem % nb;
      rem /= nb;
      float bs = (cfg.grid_high[kx] - cfg.grid_low[kx]) / nb;
      float centroid = cfg.grid_low[kx] + bs * (idx + 0.5f);
      float u = rng(purpose, 2 + kx);
      cmd[kx] = centroid + (u - 0.5f) * bs;
    }
    if (cfg.num_commands > 5) {
      if (cfg.gaitwise_curricula) {
        if (cat == 0) { cmd[5] = fmod1(cmd[5] / 2 - 0.25f); cmd[6] = fmod1(cmd[6] / 2 - 0.25f); cmd[7] = fmod1(cmd[7] / 2 - 0.25f); }
        else if (cat == 1) { cmd[5] = cmd[5] / 2 + 0.25f; cmd[6] = 0.f; cmd[7] = 0.f; }
        else if (cat == 2) { cmd[5] = 0.f; cmd[6] = cmd[6] / 2 + 0.25f; cmd[7] = 0.f; }
        else { cmd[5] = 0.f; cmd[6] = 0.f; cmd[7] = cmd[7] / 2 + 0.25f; }
      } else if (cfg.exclusive_phase_offset) {            // legged_robot.py:783-793: one of phase / offset / bound survives
        const float r = rng(purpose, 2 + GO1_MAX_COMMANDS);
        const bool trot = r < 0.34f, pace = 0.34f <= r && r < 0.67f, bnd = 0.67f <= r;
        if (pace || bnd) cmd[5] = 0.f;
        if (trot || bnd) cmd[6] = 0.f;
        if (trot || pace) cmd[7] = 0.f;
      } else if (cfg.balance_gait_distribution) {         // :795-812, same statement order (the 0.25 boundary belongs to two sets)
        const float r = rng(purpose, 2 + GO1_MAX_COMMANDS);
        const bool pronk = r <= 0.25f, trot = 0.25f <= r && r < 0.50f, pace = 0.50f <= r && r < 0.75f, bnd = 0.75f <= r;
        if (pronk) { cmd[5] = fmod1(cmd[5] / 2 - 0.25f); cmd[6] = fmod1(cmd[6] / 2 - 0.25f); cmd[7] = fmod1(cmd[7] / 2 - 0.25f); }
        if (trot) { cmd[6] = 0.f; cmd[7] = 0.f; }
        if (pace) { cmd[5] = 0.f; cmd[7] = 0.f; }
        if (bnd) { cmd[5] = 0.f; cmd[6] = 0.f; }
        if (trot) cmd[5] = cmd[5] / 2 + 0.25f;
        if (pace) cmd[6] = cmd[6] / 2 + 0.25f;
        if (bnd) cmd[7] = cmd[7] / 2 + 0.25f;
      }
      if (cfg.binary_phases) {
        cmd[5] = fmod1(rintf(2 * cmd[5]) / 2.0f); cmd[6] = fmod1(rintf(2 * cmd[6]) / 2.0f); cmd[7] = fmod1(rintf(2 * cmd[7]) / 2.0f);
      }
    }
    float nrm = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
    if (!(nrm > 0.2f)) { cmd[0] = 0.f; cmd[1] = 0.f; }
#pragma unroll
    for (int kx = 0; kx < GO1_MAX_COMMANDS; kx++)
      if (kx < cfg.num_commands) AT(B.commands, kx, e) = cmd[kx];
  } else {
    B.resample_flags[e] |= (purpose == P_CMD_CB) ? 1 : 2;
  }
#pragma unroll 1
  for (int kx = 0; kx < cfg.num_rewards + 5; kx++) AT(B.command_sums, kx, e) = 0.f;
}
DEV void resample_commands(CfgRef cfg, BufRef B, int e, int N, int64_t step, uint32_t purpose, int64_t slot_step) {
  resample_commands_t(cfg, B, e, N, purpose, slot_step, RngDirect{cfg, (uint32_t)(cfg.env_id_offset + e), step});
}

// j0, js: the joints this lane writes (0, 1: all of them; leg, 4: one lane of the environment's four)
template <class Rng>
DEV void randomize_dof_props_t(CfgRef cfg, BufRef B, int e, int N, uint32_t purpose, const Rng& rng, int j0, int js) {
  if (cfg.randomize_motor_strength) {
    float v = rng(purpose, 0) * (cfg.motor_strength_range[1] - cfg.motor_strength_range[0]) + cfg.motor_strength_range[0];
#pragma unroll 1
    for (int j = j0; j < 12; j += js) AT(B.motor_strengths, j, e) = v;
  }
  if (cfg.randomize_motor_offset) {
#pragma unroll 1
    for (int j = j0; j < 12; j += js)
      AT(B.motor_offsets, j, e) = rng(purpose, 1 + j) * (cfg.motor_offset_range[1] - cfg.motor_offset_range[0]) + cfg.motor_offset_range[0];
  }
  if (cfg.randomize_Kp_factor) {
    float v = rng(purpose, 13) * (cfg.Kp_factor_range[1] - cfg.Kp_factor_range[0]) + cfg.Kp_factor_range[0];
#pragma unroll 1
    for (int j = j0; j < 12; j += js) AT(B.Kp_factors, j, e) = v;
  }
  if (cfg.randomize_Kd_factor) {
    float v = rng(purpose, 14) * (cfg.Kd_factor_range[1] - cfg.Kd_factor_range[0]) + cfg.Kd_factor_range[0];
#pragma unroll 1
    for (int j = j0; j < 12; j += js) AT(B.Kd_factors, j, e) = v;
  }
}
DEV void randomize_dof_props(CfgRef cfg, BufRef B, int e, int N, int64_t step, uint32_t purpose) {
  randomize_dof_props_t(cfg, B, e, N, purpose, RngDirect{cfg, (uint32_t)(cfg.env_id_offset + e), step}, 0, 1);
}

// _randomize_rigid_body_props when called from the step callback (randomize_rigids_after_start, legged_robot.py:706-708).
// Deviation, documented in DESIGN.md: the reference re-draws payload / COM into its tensors but never pushes them into
// PhysX (only friction / restitution of the first 12 of 17 shapes are refreshed, App. D9), so there the privileged
// observation stops describing the simulated body; here the re-drawn values ARE the simulated body, whole robot.
template <class Rng>
DEV void randomize_rigid_props_t(CfgRef cfg, BufRef B, int e, int N, uint32_t purpose, const Rng& rng) {
  if (cfg.randomize_base_mass)
    B.payloads[e] = rng(purpose, 0) * (cfg.added_mass_range[1] - cfg.added_mass_range[0]) + cfg.added_mass_range[0];
  if (cfg.randomize_com_displacement)
#pragma unroll 1
    for (int i = 0; i < 3; i++)
      AT(B.com_displacements, i, e) = rng(purpose, 1 + i) * (cfg.com_displacement_range[1] - cfg.com_displacement_range[0]) + cfg.com_displacement_range[0];
  if (cfg.randomize_friction)
    B.friction_coeffs[e] = rng(purpose, 4) * (cfg.friction_range[1] - cfg.friction_range[0]) + cfg.friction_range[0];
  if (cfg.randomize_restitution)
    B.restitutions[e] = rng(purpose, 5) * (cfg.restitution_range[1] - cfg.restitution_range[0]) + cfg.restitution_range[0];
}
DEV void randomize_rigid_props(CfgRef cfg, BufRef B, int e, int N, int64_t step, uint32_t purpose) {
  randomize_rigid_props_t(cfg, B, e, N, purpose, RngDirect{cfg, (uint32_t)(cfg.env_id_offset + e), step});
}

// is_eval: an evaluation environment (legged_robot.py:188-195): its episode sums stay out of the training log; the first
// finished episode after the caller armed episode_sums_eval with -1 is kept there
// k0, ks: this lane's share of the per-joint / per-term loops (0, 1: everything, one lane per environment — the reset kernel; leg, 4: the
// step kernel, called by all four lanes of the environment: the scalar parts then run on the leg-0 lane, the loops on all four)
template <class Rng>
DEV void reset_env_t(CfgRef cfg, BufRef B, int e, int N, int64_t step, bool is_eval, int64_t slot_step, const Rng& rng, int k0, int ks) {
  const bool lead = k0 == 0;
  if (lead) resample_commands_t(cfg, B, e, N, P_CMD_RESET, slot_step, rng);
  randomize_dof_props_t(cfg, B, e, N, P_DOFPROPS_RESET, rng, k0, ks);
  if (lead && cfg.randomize_rigids_after_start) randomize_rigid_props_t(cfg, B, e, N, P_RIGID_RESET, rng);      // legged_robot.py:166-168
#pragma unroll 1
  for (int j = k0; j < 12; j += ks) {
    AT(B.dof_pos, j, e) = cfg.default_dof_pos[j] * (0.5f + rng(P_RESET, j));
    AT(B.dof_vel, j, e) = 0.f;
    AT(B.last_actions, j, e) = 0.f; AT(B.last_last_actions, j, e) = 0.f; AT(B.last_dof_vel, j, e) = 0.f;
  }
  if (lead) {
    float root[13];
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = cfg.base_init_state[i];
#pragma unroll
    for (int i = 0; i < 3; i++) root[i] += AT(B.env_origins, i, e);
    if (cfg.custom_origins) {
      root[0] += (2 * rng(P_RESET, 12) - 1) * cfg.x_init_range + cfg.x_init_offset;
      root[1] += (2 * rng(P_RESET, 13) - 1) * cfg.y_init_range + cfg.y_init_offset;
    }
    float yaw = (2 * rng(P_RESET, 14) - 1) * cfg.yaw_init_range;
    root[3] = 0.f; root[4] = 0.f; root[5] = sinf(0.5f * yaw); root[6] = cosf(0.5f * yaw);
#pragma unroll
    for (int i = 0; i < 6; i++) root[7 + i] = rng(P_RESET, 15 + i) - 0.5f;
#pragma unroll
    for (int i = 0; i < 13; i++) AT(B.root_states, i, e) = root[i];
    B.episode_length_buf[e] = 0;
    B.reset_buf[e] = 1;
    if (!is_eval) atomicAdd(&B.episode_log[cfg.num_rewards + 1], 1.0f);
    B.gait_indices[e] = 0.f;
  }
#pragma unroll 1
  for (int kx = k0; kx <= cfg.num_rewards; kx += ks) {
    const float sum = AT(B.episode_sums, kx, e);
    if (!is_eval) atomicAdd(&B.episode_log[kx], sum);
    else if (B.episode_sums_eval && AT(B.episode_sums_eval, kx, e) == -1.f) AT(B.episode_sums_eval, kx, e) = sum;
    AT(B.episode_sums, kx, e) = 0.f;
  }
  const int nlag = 12 * (cfg.lag_timesteps + 1);
#pragma unroll 1
  for (int i = k0; i < nlag; i += ks) B.lag_buffer[(size_t)i * N + e] = 0.f;        // ([slot][joint][env])
}
DEV void reset_env(CfgRef cfg, BufRef B, int e, int N, int64_t step, bool is_eval, int64_t slot_step) {
  reset_env_t(cfg, B, e, N, step, is_eval, slot_step, RngDirect{cfg, (uint32_t)(cfg.env_id_offset + e), step}, 0, 1);
}

// ================================================================================================
// rewards (reference corl_rewards.py:15-202) — four lanes per environment
// Each lane evaluates the part of a term that belongs to ITS leg (3 joints, 1 foot, its thigh/calf/hip bodies);
// base-only terms are evaluated on the leg-0 lane.  The caller quad-sums the partials, so every lane ends up with
// the reference's per-environment value.  One case per reference `_reward_*`.
// ================================================================================================
struct Derived {
  V3 base_pos, blv, bav, pg, gvec;
  float qx, qy, qz, qw;
  float base_hz;            // base height the reward terms read: world z (reference) or above the terrain (reward_heights_above_terrain)
};
struct FootCtx {            // the calling lane's foot
  V3 pos, vel, force;
  float fnorm;
  float foot_index, desired_contact;
  float hz;                 // foot height the reward terms read: world z (reference) or above the terrain sample under the foot
};
// terrain height at world (x, y) with the sample convention of _get_heights (reference legged_robot.py:1793-1806): truncated
// index, the lowest of the sample and its +x / +y neighbours
DEV float hf_sample_min3(CfgRef cfg, const int16_t* __restrict__ hs, float x, float y) {
  if (cfg.terrain_type == 0 || hs == nullptr) return 0.f;
  long px = (long)((x + cfg.hf_border) / cfg.hf_hscale), py = (long)((y + cfg.hf_border) / cfg.hf_hscale);
  px = px < 0 ? 0 : (px > cfg.hf_rows - 2 ? cfg.hf_rows - 2 : px);
  py = py < 0 ? 0 : (py > cfg.hf_cols - 2 ? cfg.hf_cols - 2 : py);
  const int16_t* q = hs + px * cfg.hf_cols + py;
  int16_t hm = q[0] < q[cfg.hf_cols] ? q[0] : q[cfg.hf_cols];
  hm = hm < q[1] ? hm : q[1];
  return hm * cfg.hf_vscale;
}

DEV float cf_norm(BufRef B, int b, int e, int N) {
  float x = AT(B.contact_forces, 3 * b, e), y = AT(B.contact_forces, 3 * b + 1, e), z = AT(B.contact_forces, 3 * b + 2, e);
  return sqrtf(x * x + y * y + z * z);
}
DEV float normal_cdf(float x, float sigma) { return 0.5f * (1.f + erff(x / (sigma * 1.41421356237309504880f))); }

// Everything a reward term reads from HBM, fetched as ONE batch of independent loads before the term loop: with a
// single wave per SIMD every dependent load inside the loop would expose a full memory round trip.
struct RewardIn {
  float cmd[14];
  float tq[3], q[3], qd[3], lqd[3], act[3], lact[3], llact[3], jpt[3], ljpt[3], lljpt[3];
  float cfn[4], cfn_base;          // contact-force norms of the own hip/thigh/calf/foot bodies, and of the trunk
  float pfvz;
  bool last_contact;
};
DEV void load_reward_inputs(CfgRef cfg, BufRef B, int e, int N, int leg, RewardIn& in) {
#pragma unroll
  for (int k = 0; k < 14; k++) in.cmd[k] = k < cfg.num_commands ? AT(B.commands, k, e) : 0.f;
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    in.tq[jj] = AT(B.torques, j, e); in.q[jj] = AT(B.dof_pos, j, e); in.qd[jj] = AT(B.dof_vel, j, e);
    in.lqd[jj] = AT(B.last_dof_vel, j, e); in.act[jj] = AT(B.actions, j, e); in.lact[jj] = AT(B.last_actions, j, e);
    in.llact[jj] = AT(B.last_last_actions, j, e); in.jpt[jj] = AT(B.joint_pos_target, j, e);
    in.ljpt[jj] = AT(B.last_joint_pos_target, j, e); in.lljpt[jj] = AT(B.last_last_joint_pos_target, j, e);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) in.cfn[i] = cf_norm(B, 1 + 4 * leg + i, e, N);
  in.cfn_base = cf_norm(B, 0, e, N);
  in.pfvz = AT(B.prev_foot_velocities, 3 * leg + 2, e);
  in.last_contact = AT(B.last_contacts, leg, e) != 0;
}

DEV float reward_partial(CfgRef cfg, BufRef B, int e, int N, int id, const Derived& d,
                         const FootCtx& F, int leg, const RewardIn& in) {
  const bool is0 = leg == 0;
  const int j0 = 3 * leg;
  float r = 0.f;
  switch (id) {
    case GO1_REW_TRACKING_LIN_VEL: {
      float ex = in.cmd[0] - d.blv.x, ey = in.cmd[1] - d.blv.y;
      return is0 ? expf(-(ex * ex + ey * ey) / cfg.tracking_sigma) : 0.f;
    }
    case GO1_REW_TRACKING_ANG_VEL: {
      float ez = in.cmd[2] - d.bav.z;
      return is0 ? expf(-(ez * ez) / cfg.tracking_sigma_yaw) : 0.f;
    }
    case GO1_REW_LIN_VEL_Z: return is0 ? d.blv.z * d.blv.z : 0.f;
    case GO1_REW_ANG_VEL_XY: return is0 ? d.bav.x * d.bav.x + d.bav.y * d.bav.y : 0.f;
    case GO1_REW_ORIENTATION: return is0 ? d.pg.x * d.pg.x + d.pg.y * d.pg.y : 0.f;
    case GO1_REW_TORQUES:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { float t = in.tq[jj]; r = fmaf(t, t, r); }
      return r;
    case GO1_REW_DOF_ACC:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { float a = (in.lqd[jj] - in.qd[jj]) / cfg.dt; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_ACTION_RATE:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { float a = in.lact[jj] - in.act[jj]; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_COLLISION:
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int b = 1 + 4 * leg + i;
        if (cfg.penalised_body_mask & (1u << b)) r += (in.cfn[i] > 0.1f) ? 1.f : 0.f;
      }
      if (is0 && (cfg.penalised_body_mask & 1u)) r += (in.cfn_base > 0.1f) ? 1.f : 0.f;
      return r;
    case GO1_REW_DOF_POS_LIMITS:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        const int j = j0 + jj;
        float q = in.q[jj];
        float lo = q - cfg.dof_pos_soft_lower[j], hi = q - cfg.dof_pos_soft_upper[j];
        r += -fminf(lo, 0.f) + fmaxf(hi, 0.f);
      }
      return r;
    case GO1_REW_JUMP: {
      float t = d.base_hz - (in.cmd[3] + cfg.base_height_target);
      return is0 ? -t * t : 0.f;
    }
    case GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE:
      return -(1.f - F.desired_contact) * (1.f - expf(-F.fnorm * F.fnorm / cfg.gait_force_sigma)) * 0.25f;
    case GO1_REW_TRACKING_CONTACTS_SHAPED_VEL:
      return -(F.desired_contact * (1.f - expf(-dot(F.vel, F.vel) / cfg.gait_vel_sigma))) * 0.25f;
    case GO1_REW_DOF_POS:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { float a = in.q[jj] - cfg.default_dof_pos[j0 + jj]; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_DOF_VEL:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { float a = in.qd[jj]; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_ACTION_SMOOTHNESS_1:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        float a = in.jpt[jj] - in.ljpt[jj];
        r += a * a * (in.lact[jj] != 0.f ? 1.f : 0.f);
      }
      return r;
    case GO1_REW_ACTION_SMOOTHNESS_2:
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        float a = in.jpt[jj] - 2.f * in.ljpt[jj] + in.lljpt[jj];
        r += a * a * (in.lact[jj] != 0.f ? 1.f : 0.f) * (in.llact[jj] != 0.f ? 1.f : 0.f);
      }
      return r;
    case GO1_REW_FEET_SLIP: {
      bool contact = F.force.z > 1.0f;
      bool filt = contact || in.last_contact;
      AT(B.last_contacts, leg, e) = (uint8_t)contact;
      return filt ? (F.vel.x * F.vel.x + F.vel.y * F.vel.y) : 0.f;
    }
    case GO1_REW_FEET_CONTACT_VEL: return (F.hz < 0.03f) ? dot(F.vel, F.vel) : 0.f;
    case GO1_REW_FEET_CONTACT_FORCES: return fmaxf(F.fnorm - cfg.max_contact_force, 0.f);
    case GO1_REW_FEET_CLEARANCE_CMD_LINEAR: {
      float cl = fminf(fmaxf(F.foot_index * 2.0f - 1.0f, 0.f), 1.f);
      float ph = 1.f - fabsf(1.0f - cl * 2.0f);
      float target = in.cmd[9] * ph + 0.02f;
      float df = target - F.hz;
      return df * df * (1.f - F.desired_contact);
    }
    case GO1_REW_FEET_IMPACT_VEL: {
      float pv = fminf(fmaxf(in.pfvz, -100.f), 0.f);
      return (F.fnorm > 1.0f) ? pv * pv : 0.f;
    }
    case GO1_REW_ORIENTATION_CONTROL: {
      float pitch = in.cmd[10], roll = in.cmd[11];
      float sr, cr, sp, cp;
      sincosf(-0.5f * roll, &sr, &cr);
      sincosf(-0.5f * pitch, &sp, &cp);
      float x = sr * cp, y = cr * sp, z = sr * sp, w = cr * cp;       // quat_mul((sr,0,0,cr), (0,sp,0,cp))
      V3 g = quat_rotate_inverse(x, y, z, w, d.gvec);
      float a = d.pg.x - g.x, b = d.pg.y - g.y;
      return is0 ? a * a + b * b : 0.f;
    }
    case GO1_REW_RAIBERT_HEURISTIC: {
      float l = rsqrtf(d.qz * d.qz + d.qw * d.qw);
      float yz = -d.qz * l, yw = d.qw * l;
      float width = cfg.num_commands >= 13 ? in.cmd[12] : 0.3f;
      float length = cfg.num_commands >= 14 ? in.cmd[13] : 0.45f;
      float freq = in.cmd[4], xv = in.cmd[0], yawv = in.cmd[2];
      float yv = yawv * length / 2;
      V3 fb = quat_rotate(0.f, 0.f, yz, yw, F.pos - d.base_pos);
      float ys = (leg % 2 == 0 ? 1.f : -1.f) * width / 2, xs = (leg < 2 ? 1.f : -1.f) * length / 2;
      float ph = fabsf(1.0f - F.foot_index * 2.0f) * 1.0f - 0.5f;
      float yo = ph * yv * (0.5f / freq), xo = ph * xv * (0.5f / freq);
      if (leg >= 2) yo = -yo;
      float ex = fabsf((xs + xo) - fb.x), ey = fabsf((ys + yo) - fb.y);
      return ex * ex + ey * ey;
    }
    default: return 0.f;
  }
}
#define GO1_REW_COUNT 24            // ids 0 .. GO1_REW_RAIBERT_HEURISTIC
// The configuration lists its terms as (id, scale) pairs in dict order; the kernel walks the ids in a fully unrolled
// loop instead (every term's code appears once, no jump table, the per-term scalars are fetched in one batch).
struct RewardPlan {
  int32_t kx_by_id[GO1_REW_COUNT];  // position in the configuration's list (= row of episode_sums / command_sums), -1: inactive
  float scale_by_id[GO1_REW_COUNT];
};
typedef const GO1_CONSTANT RewardPlan& PlanRef;
// the plan as 2 x GO1_REW_COUNT words in LDS (row indices, then the scales' bits): called by the first 48 threads of a workgroup at kernel start
DEV void reward_plan_to_lds(PlanRef plan, int* plan_lds, int t) {
  if (t < GO1_REW_COUNT) plan_lds[t] = plan.kx_by_id[t];
  else if (t < 2 * GO1_REW_COUNT) plan_lds[t] = __float_as_int(plan.scale_by_id[t - GO1_REW_COUNT]);
}
DEV int reward_raw_sign(int id) {
  return (id == GO1_REW_JUMP || id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL) ? -1 : 1;
}

#define QUAD_SYNC() do { __threadfence_block(); LDS_PHASE(); } while (0)      // (one wavefront: memory fence + ordering, no s_barrier)
// ---- compute_observations + privileged observations + the roll of the "last_*" buffers (reference legged_robot.py:302-491, 120-133) --------
// Reads the post-callback / post-reset state from the buffers; pg (projected gravity), clock_own (the own foot's clock input) and force_z
// (the own foot's vertical contact force) are the three values post_physics() holds in registers — a caller that does not have them
// (the helper wavefront of the step kernel) loads them from projected_gravity / clock_inputs / contact_forces, where post_physics() stored
// them.  parts: 1 = observations + roll, 2 = privileged observations (the step kernel gives 1 to the helper and keeps 2 on the master: the
// helper's share then takes as long as the master's rewards), 8 = with 1: leave the joint-position-target pair out of the roll.  Four lanes per environment, must be called by all four.
DEV void post_observations(CfgRef cfg, BufRef B, float* obs_stage, int lane, int e, int N, int64_t counter_post, V3 grav, int history_slot,
                           uint32_t& fault, V3 pg, float clock_own, float force_z, int parts, int hpart, int hparts PROF_PARAM) {
  const int leg = lane & 3;
  const bool is0 = leg == 0;
  const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
  // The configuration's switches and scales this function branches on, fetched as ONE batch: a mask built from all of them forces every
  // scalar load to be issued before the first branch.  Read where they were used, each `if (cfg.x)` was a load - wait - branch round trip
  // through the constant cache, fifteen in a row (round 5: found in the ISA; the helper wavefront's observations had become the longer side
  // of the post-physics phase).
  enum { O_LIN = 1, O_ANG = 2, O_VEL = 4, O_GLOBAL = 8, O_CMD = 16, O_TWO = 32, O_TIMING = 64, O_CLOCK = 128, O_YAW = 256, O_CONTACT = 512,
         O_NOISE = 1024, O_HEIGHTS = 2048 };
  const unsigned om = (cfg.observe_only_lin_vel ? O_LIN : 0) | (cfg.observe_only_ang_vel ? O_ANG : 0) | (cfg.observe_vel ? O_VEL : 0) |
                      (cfg.global_reference ? O_GLOBAL : 0) | (cfg.observe_command ? O_CMD : 0) | (cfg.observe_two_prev_actions ? O_TWO : 0) |
                      (cfg.observe_timing_parameter ? O_TIMING : 0) | (cfg.observe_clock_inputs ? O_CLOCK : 0) | (cfg.observe_yaw ? O_YAW : 0) |
                      (cfg.observe_contact_states ? O_CONTACT : 0) | (cfg.add_noise ? O_NOISE : 0) |
                      ((cfg.observe_heights && cfg.measure_heights) ? O_HEIGHTS : 0);
  const int o_num_obs = cfg.num_obs, o_num_cmd = cfg.num_commands, o_hist = cfg.num_obs_history;
  const float o_s_lin = cfg.obs_scale_lin_vel, o_s_ang = cfg.obs_scale_ang_vel, o_s_q = cfg.obs_scale_dof_pos, o_s_qd = cfg.obs_scale_dof_vel,
              o_clip = cfg.clip_observations;
  if (parts & 1) {         // observations (+ history ring) and the roll of the own joints' "last_*" values
    float* obs_row = B.obs_buf + (size_t)e * o_num_obs;
    const int R = o_hist + 1;     // ring slots (one spare keeps the previous window intact)
    float* h0 = B.obs_history ? B.obs_history + (size_t)e * 2 * R * o_num_obs + (size_t)history_slot * o_num_obs : nullptr;
    float* h1 = h0 ? h0 + (size_t)R * o_num_obs : nullptr;
    // Two passes.  (1) every lane stages the raw values of "its" columns in an LDS row; (2) the row is finished in
    // blocks of 4 columns, block b by lane b & 3: ONE Philox4x32 call yields the noise of all 4 columns (drawing per
    // column would run the generator 4x for the same counter), then clip and the three stores (obs, history x2).
    float* orow = obs_stage + (lane >> 2) * GO1_MAX_OBS;
    auto emit = [&](int n, float v) { orow[n] = v; };
    // everything the default observation and the history roll read, as ONE batch of loads (post-reset values)
    float o_q[3], o_qd[3], o_act[3], o_lact[3], o_jpt[3], o_ljpt[3], o_cmd[4];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      o_q[jj] = AT(B.dof_pos, j, e); o_qd[jj] = AT(B.dof_vel, j, e); o_act[jj] = AT(B.actions, j, e);
      o_lact[jj] = AT(B.last_actions, j, e); o_jpt[jj] = AT(B.joint_pos_target, j, e); o_ljpt[jj] = AT(B.last_joint_pos_target, j, e);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) o_cmd[i] = (leg + 4 * i < o_num_cmd) ? AT(B.commands, leg + 4 * i, e) : 0.f;
    const float o_gait = B.gait_indices[e];
    int n = 0;                // running column (identical on all lanes)
    if ((om & O_LIN)) { if (is0) for (int i = 0; i < 3; i++) emit(n + i, AT(B.base_lin_vel, i, e) * o_s_lin); n += 3; }
    if ((om & O_ANG)) { if (is0) for (int i = 0; i < 3; i++) emit(n + i, AT(B.base_ang_vel, i, e) * o_s_ang); n += 3; }
    if ((om & O_VEL)) {
      if (is0) {
        for (int i = 0; i < 3; i++) emit(n + i, ((om & O_GLOBAL) ? AT(B.root_states, 7 + i, e) : AT(B.base_lin_vel, i, e)) * o_s_lin);
        for (int i = 0; i < 3; i++) emit(n + 3 + i, AT(B.base_ang_vel, i, e) * o_s_ang);
      }
      n += 6;
    }
    if (is0) { emit(n, pg.x); emit(n + 1, pg.y); emit(n + 2, pg.z); }
    n += 3;
    if ((om & O_CMD)) {
      // 15 command columns: spread over the quad
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int kx = leg + 4 * i;
        if (kx < o_num_cmd) emit(n + kx, o_cmd[i] * cfg.commands_scale[kx]);
      }
      n += o_num_cmd;
    }
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      emit(n + j, (o_q[jj] - cfg.default_dof_pos[j]) * o_s_q);
      emit(n + 12 + j, o_qd[jj] * o_s_qd);
      emit(n + 24 + j, o_act[jj]);
      if ((om & O_TWO)) emit(n + 36 + j, o_lact[jj]);
    }
    n += (om & O_TWO) ? 48 : 36;
    if ((om & O_TIMING)) { if (is0) emit(n, o_gait); n += 1; }
    if ((om & O_CLOCK)) { emit(n + leg, clock_own); n += 4; }
    if ((om & O_YAW)) {
      if (is0) {
        V3 fw = quat_rotate(AT(B.root_states, 3, e), AT(B.root_states, 4, e), AT(B.root_states, 5, e), AT(B.root_states, 6, e), v3(1.f, 0.f, 0.f));
        emit(n, atan2f(fw.y, fw.x));
      }
      n += 1;
    }
    if ((om & O_CONTACT)) { emit(n + leg, force_z > 1.0f ? 1.0f : 0.0f); n += 4; }
    {
      const int n_def = n;                      // columns staged so far
      QUAD_SYNC();
#pragma unroll 1
      for (int b = leg; 4 * b < n_def; b += 4) {
        float v[4], sc[4];
        bool any = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int c = 4 * b + i;
          v[i] = c < n_def ? orow[c] : 0.f;
          sc[i] = ((om & O_NOISE) && c < n_def) ? cfg.noise_scale_vec[c] : 0.f;
          any = any || sc[i] != 0.f;
        }
        if (any) {
          uint32_t out[4];
          philox4x32_10(eg, (uint32_t)counter_post, P_NOISE, (uint32_t)b, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), out);
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (sc[i] != 0.f) v[i] += (2 * u32_to_unit(out[i]) - 1) * sc[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int c = 4 * b + i;
          if (c < n_def) {
            float x = fminf(fmaxf(v[i], -o_clip), o_clip);
            if (!(v[i] == v[i])) { x = 0.f; fault |= 1u << GO1_FAULT_OBS; }
            obs_row[c] = x;
            if (h0) { h0[c] = x; h1[c] = x; }
          }
        }
      }
    }

    // ---- roll (own joints): from the values fetched above -------------------------------------------------
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      AT(B.last_last_actions, j, e) = o_lact[jj];
      AT(B.last_actions, j, e) = o_act[jj];
      if (!(parts & 8)) {       // (8: this step's targets were rolled already — the second observation of an environment re-initialised late)
        AT(B.last_last_joint_pos_target, j, e) = o_ljpt[jj];
        AT(B.last_joint_pos_target, j, e) = o_jpt[jj];
      }
      AT(B.last_dof_vel, j, e) = o_qd[jj];
    }
  }
  if ((parts & 5) && (om & O_HEIGHTS) && B.measured_heights) {      // legacy legged_gym height block (BASELINE config 3): the LAST columns
    // Share hpart of hparts (points leg + 4 hpart, + 4 hparts, ...): the step kernel spreads the block over its three helper wavefronts — one
    // generator call per point (the noise stream is indexed by the column), 47 calls per lane on one wavefront were the longest stretch of the
    // rough-terrain step's post-physics phase (profiles/r05_step_kernel_phases_rough.txt)
    float* obs_row = B.obs_buf + (size_t)e * o_num_obs;
    const int R = o_hist + 1;
    float* h0 = B.obs_history ? B.obs_history + (size_t)e * 2 * R * o_num_obs + (size_t)history_slot * o_num_obs : nullptr;
    float* h1 = h0 ? h0 + (size_t)R * o_num_obs : nullptr;
    const int n = ((om & O_LIN) ? 3 : 0) + ((om & O_ANG) ? 3 : 0) + ((om & O_VEL) ? 6 : 0) + 3 + ((om & O_CMD) ? o_num_cmd : 0) +
                  ((om & O_TWO) ? 48 : 36) + ((om & O_TIMING) ? 1 : 0) + ((om & O_CLOCK) ? 4 : 0) + ((om & O_YAW) ? 1 : 0) +
                  ((om & O_CONTACT) ? 4 : 0);                         // the columns staged before it (the running column of the block above)
    const int np = cfg.num_height_x * cfg.num_height_y;
    const float z = AT(B.root_states, 2, e);
    const bool noisy = (om & O_NOISE) && cfg.height_noise_scale != 0.f;
    const float o_s_h = cfg.obs_scale_height, o_hn = cfg.height_noise_scale;
#pragma unroll 2
    for (int p = leg + 4 * hpart; p < np; p += 4 * hparts) {
      float v = fminf(fmaxf(z - 0.5f - AT(B.measured_heights, p, e), -1.f), 1.f) * o_s_h;
      if (noisy) v += (2 * rng_uniform(cfg, eg, counter_post, P_NOISE, n + p) - 1) * o_hn;
      v = fminf(fmaxf(v, -o_clip), o_clip);
      obs_row[n + p] = v;
      if (h0) { h0[n + p] = v; h1[n + p] = v; }
    }
  }
    PROF(13);
  if (parts & 2) {         // privileged observations (leg-0 lane)
    // (the eleven `priv_enabled` switches as one mask: one batch of scalar loads instead of eleven load - wait - branch round trips)
    unsigned pm = 0;
#pragma unroll
    for (int i = 0; i <= GO1_PRIV_DESIRED_CONTACT; i++) pm |= cfg.priv_enabled[i] ? 1u << i : 0u;
    if (is0) {
      float* pv = B.privileged_obs_buf + (size_t)e * cfg.num_privileged_obs;
      int np = 0;
      auto priv = [&](int idx, float val) {
        float v = (val - cfg.priv_shift[idx]) * cfg.priv_scale[idx];
        pv[np++] = fminf(fmaxf(v, -cfg.clip_observations), cfg.clip_observations);
      };
      auto privraw = [&](float v) { pv[np++] = fminf(fmaxf(v, -cfg.clip_observations), cfg.clip_observations); };
      if ((pm & (1u << GO1_PRIV_FRICTION))) priv(GO1_PRIV_FRICTION, B.friction_coeffs[e]);
      if ((pm & (1u << GO1_PRIV_RESTITUTION))) priv(GO1_PRIV_RESTITUTION, B.restitutions[e]);
      if ((pm & (1u << GO1_PRIV_BASE_MASS))) priv(GO1_PRIV_BASE_MASS, B.payloads[e]);
      if ((pm & (1u << GO1_PRIV_COM_DISPLACEMENT))) for (int i = 0; i < 3; i++) priv(GO1_PRIV_COM_DISPLACEMENT, AT(B.com_displacements, i, e));
      if ((pm & (1u << GO1_PRIV_MOTOR_STRENGTH)))
#pragma unroll 1
        for (int j = 0; j < 12; j++) priv(GO1_PRIV_MOTOR_STRENGTH, AT(B.motor_strengths, j, e));
      if ((pm & (1u << GO1_PRIV_MOTOR_OFFSET)))
#pragma unroll 1
        for (int j = 0; j < 12; j++) priv(GO1_PRIV_MOTOR_OFFSET, AT(B.motor_offsets, j, e));
      if ((pm & (1u << GO1_PRIV_BODY_HEIGHT))) priv(GO1_PRIV_BODY_HEIGHT, AT(B.root_states, 2, e));
      if ((pm & (1u << GO1_PRIV_BODY_VELOCITY))) for (int i = 0; i < 3; i++) priv(GO1_PRIV_BODY_VELOCITY, AT(B.base_lin_vel, i, e));
      if ((pm & (1u << GO1_PRIV_GRAVITY))) {
        privraw(((grav.x - cfg.gravity[0]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
        privraw(((grav.y - cfg.gravity[1]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
        privraw(((grav.z - cfg.gravity[2]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
      }
      if ((pm & (1u << GO1_PRIV_CLOCK_INPUTS))) for (int f = 0; f < 4; f++) privraw(AT(B.clock_inputs, f, e));
      if ((pm & (1u << GO1_PRIV_DESIRED_CONTACT))) for (int f = 0; f < 4; f++) privraw(AT(B.desired_contact_states, f, e));
    }
  }
}

// ================================================================================================
// post-physics maps (reference legged_robot.py:90-136), FOUR lanes per environment (lane = leg).
// Per-leg work (gait clock of the own foot, reward partials, own joints' observation columns, history roll) runs
// on every lane; per-environment scalar work on the leg-0 lane; `__syncthreads()` (one-wave workgroup) orders the
// hand-overs through HBM/L2.  Must be called by all four lanes of the environment.
// ================================================================================================

// helper_flag != nullptr (step kernel, workgroups of nw wavefronts): the observations are taken OFF the master's serial spine — once the
// callbacks and the termination test are through, a helper wavefront runs post_observations() (from the buffers) while the master evaluates
// the rewards.  Two workgroup barriers: S1 (the state the observations read is stored, *helper_flag says whether the helper goes) and S2 (the
// helper is through).  The helper does NOT go when an environment of the wavefront resets (the observations then read the re-initialised
// state, which exists only after the rewards: master as before).  A reset that only the rewards bring about (a non-finite term: the
// failed-simulation guard) is met after S2 by evaluating the observations of that environment again.
// PLANE: the step kernel's plane instance (terrain_type 0 or no samples bound — go1sim.hip picks it on exactly that test): the height-field
// branches are not compiled into it
template <bool PLANE = false>
DEV void post_physics(CfgRef cfg, BufRef B, const int* plan_lds, float* obs_stage, int lane, int e, int N, int64_t counter_post, V3 grav,
                      int history_slot, uint32_t& fault, bool is_eval, float* helper_flag, int nw PROF_PARAM) {
  const int leg = lane & 3;
  const bool is0 = leg == 0;
  const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
  // (the switches and intervals the callbacks branch on as one batch of scalar loads: see post_observations)
  enum { F_TELE = 1, F_GAIT = 2, F_PUSH = 4, F_MEAS = 8, F_ABOVE = 16, F_TERMH = 32, F_PACING = 64, F_RIGID = 128 };
  const unsigned fm = (cfg.teleport_robots ? F_TELE : 0) | (cfg.observe_gait_commands ? F_GAIT : 0) | (cfg.push_robots ? F_PUSH : 0) |
                      (cfg.measure_heights ? F_MEAS : 0) | (cfg.reward_heights_above_terrain ? F_ABOVE : 0) |
                      (cfg.use_terminal_body_height ? F_TERMH : 0) | (cfg.pacing_offset ? F_PACING : 0) | (cfg.randomize_rigids_after_start ? F_RIGID : 0);
  const int c_resample = cfg.resample_interval, c_push = cfg.push_interval, c_rand = cfg.rand_interval, c_maxlen = cfg.max_episode_length;
  Derived d;
  const int ep_len = B.episode_length_buf[e] + 1;
  d.base_pos = v3(AT(B.root_states, 0, e), AT(B.root_states, 1, e), AT(B.root_states, 2, e));
  d.qx = AT(B.root_states, 3, e); d.qy = AT(B.root_states, 4, e); d.qz = AT(B.root_states, 5, e); d.qw = AT(B.root_states, 6, e);
  V3 vl = v3(AT(B.root_states, 7, e), AT(B.root_states, 8, e), AT(B.root_states, 9, e));
  V3 va = v3(AT(B.root_states, 10, e), AT(B.root_states, 11, e), AT(B.root_states, 12, e));
  d.blv = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, vl);
  d.bav = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, va);
  d.gvec = (1.f / norm(grav)) * grav;
  d.pg = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, d.gvec);
  const float root_z = AT(B.root_states, 2, e);
  QUAD_SYNC();                // every lane has read the pre-callback state
  if (is0) {
    B.episode_length_buf[e] = ep_len;
    AT(B.base_lin_vel, 0, e) = d.blv.x; AT(B.base_lin_vel, 1, e) = d.blv.y; AT(B.base_lin_vel, 2, e) = d.blv.z;
    AT(B.base_ang_vel, 0, e) = d.bav.x; AT(B.base_ang_vel, 1, e) = d.bav.y; AT(B.base_ang_vel, 2, e) = d.bav.z;
    AT(B.projected_gravity, 0, e) = d.pg.x; AT(B.projected_gravity, 1, e) = d.pg.y; AT(B.projected_gravity, 2, e) = d.pg.z;
    // ---- _post_physics_step_callback: teleport, interval command resampling ----------------------
    if ((fm & F_TELE)) {
      float x = AT(B.root_states, 0, e), y = AT(B.root_states, 1, e), th = cfg.teleport_thresh, xo = cfg.teleport_x_offset;
      if (x < th + xo) x += cfg.terrain_length * (cfg.terrain_num_rows - 1);
      if (x > cfg.terrain_length * cfg.terrain_num_rows - th + xo) x -= cfg.terrain_length * (cfg.terrain_num_rows - 1);
      if (y < th) y += cfg.terrain_width * (cfg.terrain_num_cols - 1);
      if (y > cfg.terrain_width * cfg.terrain_num_cols - th) y -= cfg.terrain_width * (cfg.terrain_num_cols - 1);
      AT(B.root_states, 0, e) = x; AT(B.root_states, 1, e) = y;
    }
    if (GO1_RARE(ep_len % c_resample == 0)) resample_commands(cfg, B, e, N, counter_post, P_CMD_CB, counter_post - 1);
  }
  QUAD_SYNC();                // commands / command_sums of this step are final
  PROF(8);
  // ---- gait clock of the own foot (_step_contact_targets) -----------------------------------------
  FootCtx F;
  F.foot_index = AT(B.foot_indices, leg, e);
  F.desired_contact = AT(B.desired_contact_states, leg, e);
  float clock_own = AT(B.clock_inputs, leg, e);
  if ((fm & F_GAIT)) {
    float freq = AT(B.commands, 4, e), phase = AT(B.commands, 5, e), offset = AT(B.commands, 6, e), bound = AT(B.commands, 7, e), dur = AT(B.commands, 8, e);
    float gi = fmod1(B.gait_indices[e] + cfg.dt * freq);
    float f0 = gi + phase + offset + bound, f3 = gi + phase;
    float f1 = (fm & F_PACING) ? gi + bound : gi + offset;
    float f2 = (fm & F_PACING) ? gi + offset : gi + bound;
    float fi = leg == 0 ? f0 : leg == 1 ? f1 : leg == 2 ? f2 : f3;
    float rem = fmod1(fi);
    float idx = fi;
    if (rem < dur) idx = rem * (0.5f / dur);
    else if (rem > dur) idx = 0.5f + (rem - dur) * (0.5f / (1.f - dur));
    clock_own = sinf(2.f * PI_F * idx);
    float kap = cfg.kappa_gait_probs, x = fmod1(idx);
    float sm = normal_cdf(x, kap) * (1.f - normal_cdf(x - 0.5f, kap)) + normal_cdf(x - 1.f, kap) * (1.f - normal_cdf(x - 0.5f - 1.f, kap));
    F.foot_index = rem;
    F.desired_contact = sm;
    QUAD_SYNC();              // all lanes have read the old gait index
    if (is0) B.gait_indices[e] = gi;
    AT(B.foot_indices, leg, e) = rem;
    AT(B.clock_inputs, leg, e) = clock_own;
    AT(B.desired_contact_states, leg, e) = sm;
  }
  if (is0) {
    if (GO1_RARE((fm & F_PUSH) && ep_len % c_push == 0)) {
      AT(B.root_states, 7, e) = (2 * rng_uniform(cfg, eg, counter_post, P_PUSH, 0) - 1) * cfg.max_push_vel_xy;
      AT(B.root_states, 8, e) = (2 * rng_uniform(cfg, eg, counter_post, P_PUSH, 1) - 1) * cfg.max_push_vel_xy;
    }
    if (GO1_RARE(ep_len % c_rand == 0)) {
      randomize_dof_props(cfg, B, e, N, counter_post, P_DOFPROPS_CB);
      if ((fm & F_RIGID)) randomize_rigid_props(cfg, B, e, N, counter_post, P_RIGID);
    }
  }
  PROF(9);
  // ---- own foot / bodies -------------------------------------------------------------------------
  F.pos = v3(AT(B.foot_positions, 3 * leg, e), AT(B.foot_positions, 3 * leg + 1, e), AT(B.foot_positions, 3 * leg + 2, e));
  F.vel = v3(AT(B.foot_velocities, 3 * leg, e), AT(B.foot_velocities, 3 * leg + 1, e), AT(B.foot_velocities, 3 * leg + 2, e));
  {
    const int b = 4 + 4 * leg;
    F.force = v3(AT(B.contact_forces, 3 * b, e), AT(B.contact_forces, 3 * b + 1, e), AT(B.contact_forces, 3 * b + 2, e));
    F.fnorm = norm(F.force);
  }
  // ---- measured terrain heights (_get_heights, reference legged_robot.py:1772-1806): 187 points over the quad ----
  float mean_height = 0.f;
  if ((fm & F_MEAS) && B.measured_heights) {
    const int np = cfg.num_height_x * cfg.num_height_y;
    const float l = rsqrtf(d.qz * d.qz + d.qw * d.qw);       // quat_apply_yaw: yaw-only rotation of the scan pattern
    const float yz = d.qz * l, yw = d.qw * l;
    const float bx = AT(B.root_states, 0, e), by = AT(B.root_states, 1, e);
    float sum = 0.f;
    const int ny = cfg.num_height_y;
    int ix = leg / ny, iy = leg % ny;          // p = ix * ny + iy, advanced by 4 per point (a divide and a remainder per point otherwise)
    if (!PLANE && cfg.terrain_type != 0 && B.height_samples) {
      // HB points per turn: all their sample loads are issued before the first is used.  One point per turn was a chain of 47 dependent
      // round trips to L2 per lane — 49 k cycles of the rough-terrain step (profiles/r05_step_kernel_phases_rough.txt), none of it arithmetic.
      // The sample index is the reference's fp32 quotient truncated (legged_robot.py:1795-1797); the library is built with correctly rounded
      // fp32 division since round 4 (__graft_entry__.py SIM_FLAGS), so the plain quotient IS that value (rounds 1-3 formed it through fp64).
      constexpr int HB = 8;
      const long rmax = cfg.hf_rows - 2, cmax = cfg.hf_cols - 2;
      const int cols = cfg.hf_cols;
      const float border = cfg.hf_border, hscale = cfg.hf_hscale, vscale = cfg.hf_vscale;
      const int16_t* __restrict__ hs = B.height_samples;
#pragma unroll 1
      for (int p0 = leg; p0 < np; p0 += 4 * HB) {
        int s00[HB], s10[HB], s01[HB];
        float gx[HB], gy[HB];
#pragma unroll
        for (int k = 0; k < HB; k++) {                // (1) the pattern's points: HB gathers from the configuration block, one wait
          const bool live = p0 + 4 * k < np;          // (a dead slot of the last turn reads the cell of point 0 and is dropped)
          gx[k] = cfg.height_points_x[live ? ix : 0];
          gy[k] = cfg.height_points_y[live ? iy : 0];
          iy += 4;
          while (iy >= ny) { iy -= ny; ix++; }
        }
#pragma unroll
        for (int k = 0; k < HB; k++) {                // (2) their cells: 2 loads per point (samples [0], [1] as one), one wait
          V3 w = quat_rotate(0.f, 0.f, yz, yw, v3(gx[k], gy[k], 0.f));
          long px = (long)((w.x + bx + border) / hscale);
          long py = (long)((w.y + by + border) / hscale);
          px = px < 0 ? 0 : (px > rmax ? rmax : px);
          py = py < 0 ? 0 : (py > cmax ? cmax : py);
          const int16_t* q = hs + px * cols + py;
          s00[k] = q[0]; s10[k] = q[cols]; s01[k] = q[1];
        }
#pragma unroll
        for (int k = 0; k < HB; k++) {
          const int p = p0 + 4 * k;
          if (p < np) {
            int hm = s00[k] < s10[k] ? s00[k] : s10[k];
            hm = hm < s01[k] ? hm : s01[k];
            const float hgt = hm * vscale;
            AT(B.measured_heights, p, e) = hgt;
            sum += hgt;
          }
        }
      }
    } else {
#pragma unroll 1
      for (int p = leg; p < np; p += 4) AT(B.measured_heights, p, e) = 0.f;
    }
    mean_height = quad_sum(sum) / np;
  }
  // heights the reward terms read (reference: world z; reward_heights_above_terrain: above the ground — go1sim.h)
  F.hz = F.pos.z;
  d.base_hz = d.base_pos.z;
  if ((fm & F_ABOVE)) {
    F.hz -= PLANE ? 0.f : hf_sample_min3(cfg, B.height_samples, F.pos.x, F.pos.y);
    d.base_hz -= ((fm & F_MEAS) && B.measured_heights) ? mean_height : (PLANE ? 0.f : hf_sample_min3(cfg, B.height_samples, d.base_pos.x, d.base_pos.y));
  }
  // ---- check_termination ---------------------------------------------------------------------------
  PROF(10);
  RewardIn rin;
  load_reward_inputs(cfg, B, e, N, leg, rin);
  PROF(16);
  float term = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int b = 1 + 4 * leg + i;
    if ((cfg.termination_body_mask & (1u << b)) && !(rin.cfn[i] <= 1.0f)) term = 1.f;      // (a NaN force terminates too)
  }
  if (is0 && (cfg.termination_body_mask & 1u) && !(rin.cfn_base <= 1.0f)) term = 1.f;
  bool reset = quad_sum(term) > 0.f;
  const bool time_out = ep_len > c_maxlen;
  reset = reset || time_out;
  if ((fm & F_TERMH) && root_z - mean_height < cfg.terminal_body_height) reset = true;
  if (is0) { B.time_out_buf[e] = (uint8_t)time_out; B.reset_buf[e] = (uint8_t)reset; }

  PROF(17);
  bool sim_failed = quad_sum((fault & GO1_FAULT_FATAL_MASK) ? 1.f : 0.f) > 0.f;
  bool helper_obs = false;
  if (helper_flag != nullptr) {
    helper_obs = __ballot(reset || sim_failed) == 0ull;
    if (lane == 0) *helper_flag = helper_obs ? 1.f : 0.f;
    __threadfence_block();
    BLOCK_SYNC(nw);             // S1
  }
  // ---- compute_reward ----------------------------------------------------------------------------------
  // Failed-simulation containment: a fault raised by the physics of this step (go1sim.h Go1FaultBit), or a reward term
  // that is not finite, ends the episode and counts as reward 0 instead of poisoning the running sums, the curriculum
  // statistics and, through the advantage normalisation, every other environment's gradient.  Every activation is
  // reported through fault_flags / fault_counts.
  float rew = 0.f, pos = 0.f, neg = 0.f;
  // The plan's 2 x 24 scalars come from LDS (reward_plan_to_lds at kernel start) as ONE batch of broadcast reads.  As scalar loads from the
  // constant block inside the term loop they were 48 DEPENDENT round trips — load, wait, branch, load, wait — which was most of this phase
  // (round 5: found in the ISA).  The same bits arrive: results are unchanged.
  int plan_kx[GO1_REW_COUNT], plan_sc[GO1_REW_COUNT];
#pragma unroll
  for (int id = 0; id < GO1_REW_COUNT; id++) { plan_kx[id] = plan_lds[id]; plan_sc[id] = plan_lds[GO1_REW_COUNT + id]; }
#pragma unroll
  for (int id = 0; id < GO1_REW_COUNT; id++) {
    const int kx = WAVE_UNIFORM(plan_kx[id]);
    if (kx >= 0) {                  // wave-uniform
      const float sc = __int_as_float(WAVE_UNIFORM(plan_sc[id]));
      float r = quad_sum(reward_partial(cfg, B, e, N, id, d, F, leg, rin)) * sc;
      if (!(fabsf(r) <= 3.0e38f)) { r = 0.f; sim_failed = true; fault |= 1u << GO1_FAULT_REWARD; }
      rew += r;
      if (reward_raw_sign(id) * sc >= 0) pos += r; else neg += r;
      if ((kx & 3) == leg) {      // running sums: fire-and-forget fp32 atomics (one writer per address, so the result is
                                  // the plain += of the reference; no load to wait for)
        unsafeAtomicAdd(&AT(B.episode_sums, kx, e), r);
        const bool shaped = id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL;
        unsafeAtomicAdd(&AT(B.command_sums, kx, e), shaped ? sc + r : r);
      }
    }
  }
  if (cfg.only_positive_rewards) rew = fmaxf(rew, 0.f);
  else if (cfg.only_positive_rewards_ji22_style) rew = pos * expf(neg / cfg.sigma_rew_neg);
  if (!(fabsf(rew) <= 3.0e38f)) { rew = 0.f; sim_failed = true; fault |= 1u << GO1_FAULT_REWARD; }
  if (GO1_RARE(sim_failed)) {
    rew = 0.f;
    reset = true;
    if (is0) B.reset_buf[e] = 1;
    // nothing non-finite may survive the launch: the reported forces / foot velocities and the actuator-network
    // histories (which reset_idx leaves alone, reference quirk App. D2) of a failed environment are cleared
#pragma unroll
    for (int i = 0; i < 12; i++) AT(B.contact_forces, 3 * (1 + 4 * leg) + i, e) = 0.f;
    if (is0) { AT(B.contact_forces, 0, e) = 0.f; AT(B.contact_forces, 1, e) = 0.f; AT(B.contact_forces, 2, e) = 0.f; }
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      AT(B.foot_velocities, j, e) = 0.f; AT(B.prev_foot_velocities, j, e) = 0.f; AT(B.torques, j, e) = 0.f;
      AT(B.foot_positions, j, e) = 0.f;
      AT(B.joint_pos_err_last, j, e) = 0.f; AT(B.joint_pos_err_last_last, j, e) = 0.f;
      AT(B.joint_vel_last, j, e) = 0.f; AT(B.joint_vel_last_last, j, e) = 0.f;
    }
    d.pg = d.gvec;            // the observation of the re-initialised environment must not see the failed orientation
    if (is0) {
#pragma unroll
      for (int i = 0; i < 3; i++) { AT(B.base_lin_vel, i, e) = 0.f; AT(B.base_ang_vel, i, e) = 0.f; }
      AT(B.projected_gravity, 0, e) = d.pg.x; AT(B.projected_gravity, 1, e) = d.pg.y; AT(B.projected_gravity, 2, e) = d.pg.z;
    }
  }
  if (is0) {
    B.rew_buf[e] = rew;
    unsafeAtomicAdd(&AT(B.episode_sums, cfg.num_rewards, e), rew);
    const int k0 = cfg.num_rewards;
    const float c0 = rin.cmd[0], c2 = rin.cmd[2];
    unsafeAtomicAdd(&AT(B.command_sums, k0 + 0, e), d.blv.x);
    unsafeAtomicAdd(&AT(B.command_sums, k0 + 1, e), d.bav.z);
    unsafeAtomicAdd(&AT(B.command_sums, k0 + 2, e), (d.blv.x - c0) * (d.blv.x - c0));
    unsafeAtomicAdd(&AT(B.command_sums, k0 + 3, e), (d.bav.z - c2) * (d.bav.z - c2));
    unsafeAtomicAdd(&AT(B.command_sums, k0 + 4, e), 1.f);
  }
  PROF(11);
  QUAD_SYNC();                // running sums complete before a reset logs / clears them
  // ---- reset, compute_observations (+ privileged observations, roll) --------------------------------------------------------------
  const bool late = __ballot(reset) != 0ull;          // an environment of the wavefront is re-initialised
  // the reset: all four lanes of the environment — the draws as a table they fill together (in the observation staging row, free until the
  // observations are staged), then the scalar parts on the leg-0 lane and the per-joint / per-term loops on all four (reset_env_t)
  auto do_reset = [&]() {
    float* rt = obs_stage + (lane >> 2) * GO1_MAX_OBS;
    if (GO1_RARE(reset)) reset_rng_table(cfg, (uint32_t)(cfg.env_id_offset + e), counter_post, rt, leg);
    QUAD_SYNC();              // (wavefront-wide ordering points stay outside the divergent blocks)
    if (GO1_RARE(reset)) reset_env_t(cfg, B, e, N, counter_post, is_eval, counter_post - 1, RngTable{rt}, leg, 4);
    QUAD_SYNC();              // the observation sees the post-reset state, as in the reference
  };
  if (helper_flag == nullptr) {                       // one-wavefront workgroup: everything here
    do_reset();
    PROF(12);
    post_observations(cfg, B, obs_stage, lane, e, N, counter_post, grav, history_slot, fault, d.pg, clock_own, F.force.z, 3, 0, 1 PROF_PASS);
  } else if (helper_obs && !late) {                   // the usual step: the helpers are writing the observations, the privileged ones here
    if (lane == 0) helper_flag[1] = 0.f;
    PROF(12);
    post_observations(cfg, B, obs_stage, lane, e, N, counter_post, grav, history_slot, fault, d.pg, clock_own, F.force.z, 2, 0, 1 PROF_PASS);
    BLOCK_SYNC(nw);                                   // S2: the helpers' observations are written
  } else if (helper_obs) {                            // only through the failed-simulation guard above: the helpers wrote the observations of
    if (lane == 0) helper_flag[1] = 0.f;              // the state before; once they are through (their staging rows are free then), this
    BLOCK_SYNC(nw);                                   // S2   environment is re-initialised and observed once more
    do_reset();
    PROF(12);
    // (parts 8: the helpers rolled this step's joint position targets already and the reset leaves the pair alone — a second roll would put
    //  THIS step's target into last_last_joint_pos_target; the action / joint-rate buffers are re-zeroed by the reset, so rolling them again is exact)
    post_observations(cfg, B, obs_stage, lane, e, N, counter_post, grav, history_slot, fault, d.pg, clock_own, F.force.z, reset ? 3 | 8 : 2, 0, 1 PROF_PASS);
  } else {                                            // a reset known at S1: the helpers are waiting at S2
    do_reset();
    PROF(12);
    // LATE HAND-OVER: the helpers take the observations of the re-initialised wavefront as well (from the buffers: the reset leaves
    // projected_gravity, clock_inputs and contact_forces as post_physics stored them — the reference's observation after reset_idx reads the
    // same stale values) while the privileged ones are written here; S3 ends their round.  Not after a failed simulation (the guard above
    // cleared buffers the observation would read differently from the registers passed here): that rare case stays on this wavefront.
    const bool hand_over = __ballot(sim_failed) == 0ull;
    if (lane == 0) helper_flag[1] = hand_over ? 1.f : 0.f;
    __threadfence_block();
    if (hand_over) {
      BLOCK_SYNC(nw);                                 // S2: the helpers start
      post_observations(cfg, B, obs_stage, lane, e, N, counter_post, grav, history_slot, fault, d.pg, clock_own, F.force.z, 2, 0, 1 PROF_PASS);
      BLOCK_SYNC(nw);                                 // S3: their observations are written
    } else {
      post_observations(cfg, B, obs_stage, lane, e, N, counter_post, grav, history_slot, fault, d.pg, clock_own, F.force.z, 3, 0, 1 PROF_PASS);
      BLOCK_SYNC(nw);                                 // S2 (the helpers had nothing to do)
    }
  }
  PROF(14);
  PROF(15);
}
