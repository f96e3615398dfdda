// go1sim.hip — MI355X (gfx950 / CDNA4) Go1 vectorised step: kernels + the C-ABI of include/go1sim.h.
//
// Mapping: FOUR lanes per environment (one per leg), 16 environments per 64-lane wavefront; the step kernel's workgroup is
// 4 wavefronts for those 16 environments — one MASTER that runs the step and three HELPERS on the CU's other SIMDs, fed through
// LDS (4096 envs -> 256 workgroups x 4 wavefronts: one wavefront per SIMD of the chip).  The step is a latency / issue-bound
// O(n_dof) recursion (not a bandwidth- or MFMA-bound kernel: DESIGN.md §5), so the design spends lanes on the parallelism a
// single robot has — its four independent leg chains (kept in registers, go1_physics.h) — and wavefronts on the data-parallel
// blocks of a substep.  State lives in HBM as SoA [component][env]; the tensor maps of post_physics_step (go1_maps.h) run
// 4-wide on the master.
//
// Physics per substep (replaces gym.simulate, reference legged_robot.py:76-80; contract: DESIGN.md §2, oracle/go1_oracle.c):
//   1. torque model: the helpers build the actuator network's input rows (one row per helper lane) from the q, qd the master posts and
//      evaluate the network (hidden layer on MFMA, fp16 hi/lo split) while the master runs
//   2. forward kinematics + contact candidates (terrain top surface, vertical faces of a trimesh terrain, self-collision
//      capsules), the solver's contact list (<= 24 contacts, priority order, overflow counted per class)
//   3. Featherstone articulated-body algorithm, world-aligned frame at the base origin; base terms quad-reduced
//   4. the rows of the solve in factorised coordinates (M^-1 = A A^T from the ABA factors): terrain contacts finished by the
//      helpers, self-contacts and joint-limit rows by the master
//   5. matrix-free projected Gauss-Seidel on (normal, 2 tangents | limit rows), static / dynamic Coulomb cone: trunk and body-body
//      contacts in list order, then the four legs' contacts side by side (a lane walks through ITS leg's contacts; hip / thigh rows with
//      mass splitting), then the limit rows
//   6. one more impulse propagation applies all impulses; semi-implicit Euler
// After the last substep the master runs the post-physics maps; once the callbacks and the termination test are through, helper 1 takes the
// observations (+ history ring, roll) while the master evaluates the rewards (go1_maps.h post_physics / post_observations).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/go1sim.h"
#include "go1_math.h"
#include "go1_maps.h"
#include "go1_physics.h"
#define GO1SIM_STR_(x) #x
#define GO1SIM_STR(x) GO1SIM_STR_(x)

static_assert(A_IO_END <= PKL_Q * WAVE * 4, "the actuator network's transient rows are overlaid on the emission hand-over packets");
static_assert(L_KL + 7 <= L_END && L_END >= GO1_MAX_OBS, "post_physics stages the observation rows in the solver's scalar block");
static_assert(MAXC == GO1_MAX_CONTACTS && GO1_SIG_MAX_SUBSTEPS >= 1, "contact list size is part of the ABI (oracle and kernel share it)");
struct SimConst {            // lives in device memory (one per handle): indexable with scalar loads
  Go1SimConfig cfg;
  Go1SimConfig cfg_eval;     // configuration of the environments [num_train_envs, num_envs) (a copy of cfg without a split)
  int32_t num_train_envs;
  Go1SimBuffers buf;
  RewardPlan rew;            // derived at create / set_config: reward terms indexed by id (go1_maps.h)
};
// the wavefront's configuration block: 16 consecutive environments are all train or all evaluation environments
#define WAVE_CFG(csc, first_env) (((first_env) >= (csc)->num_train_envs) ? (csc)->cfg_eval : (csc)->cfg)
struct StepArgs {
  const SimConst* __restrict__ sc;
  const float* actions;      // (N,12) row-major, or SoA for the piecewise entry points
  int64_t counter;           // common_step_counter before this step
  int32_t lag_head;
  int32_t history_slot;
  int32_t mode;              // 0 full step, 1 torques only, 2 one physics substep, 3 reset ids, 4 post-physics only
  float gravity_override[3];
  const int32_t* ids;
  int32_t n_ids;
};

// ================================================================================================
// kernels
// ================================================================================================
// rare path: publish the fault bits of this environment (go1sim.h Go1FaultBit).  The four lanes of an environment raise
// environment-wide bits identically and per-leg bits on their own: the word is OR-ed over the quad and the leg-0 lane
// publishes it, so every event counts once per environment and step.
DEV void report_fault(BufRef B, int e, uint32_t fault) {
  fault |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fault, 0xB1, 0xF, 0xF, false);
  fault |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fault, 0x4E, 0xF, 0xF, false);
  const bool mine = (threadIdx.x & 3) == 0 && fault != 0 && B.fault_flags != nullptr;
  if (__ballot(mine) == 0ull) return;
  if (mine) atomicOr(&B.fault_flags[e], fault);
  if (B.fault_counts == nullptr) return;
  // one atomic per wavefront and bit (under a learning policy hundreds of environments report the count-only bits in the same
  // launch: per-environment atomics on ONE address serialise into milliseconds)
  for (int b = 0; b < GO1_FAULT_BITS; b++) {
    const unsigned long long m = __ballot(mine && (fault & (1u << b)));
    if (m != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&B.fault_counts[b], (uint32_t)__builtin_popcountll(m));
  }
}
// contact points without a solver slot, per class: accumulated per environment over the substeps (leg-0 lanes), one atomic per
// wavefront and class
DEV void report_drops(BufRef B, const uint32_t (&drops)[GO1_CC_COUNT]) {
  if (B.contact_drop_counts == nullptr) return;
  uint32_t any = 0;
#pragma unroll
  for (int c = 0; c < GO1_CC_COUNT; c++) any |= drops[c];
  if (__ballot(any != 0u) == 0ull) return;
#pragma unroll
  for (int c = 0; c < GO1_CC_COUNT; c++) {
    uint32_t v = 0;                              // wave sum, bit plane by bit plane (an environment drops < 2^10 points per step)
    for (int bit = 0; bit < 10; bit++) v += (uint32_t)__builtin_popcountll(__ballot((drops[c] >> bit) & 1u)) << bit;
    if (v != 0u && (threadIdx.x & 63) == 0) atomicAdd(&B.contact_drop_counts[c], v);
  }
}

// Workgroup = STEP_WAVES wavefronts for 16 environments.  Wavefront 0 (the "master") runs the step — four lanes per
// environment, one per leg; the other wavefronts are helpers on the CU's other SIMDs: they take the two data-parallel
// blocks of every substep (the actuator network's row tiles, the rows of the listed terrain contacts), fed through LDS,
// and wait at workgroup barriers otherwise.  4096 environments -> 256 workgroups x 4 wavefronts: one per SIMD.
// WALLS: the terrain has vertical faces (hf_wall_units > 0); the plain instance carries none of that code or its registers.
#ifndef STEP_WAVES
#define STEP_WAVES 4
#endif
template <bool WALLS, bool SIG, bool PLANE>
DEV void step_body(const StepArgs& A, float* lds, lf4* ldsx, float* act_lds, float* acth, int* plan_lds) {
  const int nw = STEP_WAVES, wv = WAVE_UNIFORM((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63, leg = lane & 3;
  for (int i = threadIdx.x; i < L_END * EPW; i += WAVE * STEP_WAVES) lds[i] = 0.f;
  for (int i = threadIdx.x; i < X_END; i += WAVE * STEP_WAVES) ldsx[i] = (lf4){0.f, 0.f, 0.f, 0.f};      // finite everywhere: stale records are read (with zero weight)
  SolverLds Z;
  Z.lds = lds; Z.x = ldsx;
  const GO1_CONSTANT SimConst* csc = (const GO1_CONSTANT SimConst*)(uintptr_t)A.sc;
  CfgRef cfg = WAVE_CFG(csc, (int)blockIdx.x * EPW);
  BufRef B = csc->buf;
  const int N = cfg.num_envs;
  const int e = blockIdx.x * EPW + (lane >> 2);
  const bool full_wave = (int)(blockIdx.x + 1) * EPW <= N;      // the MFMA torque model needs all 64 lanes
  const bool substep_only = A.mode == 2;                         // piecewise entry point: ONE physics substep with the torques in the buffer
  const bool mfma_torque = full_wave && cfg.control_type == 1 && !substep_only;
#if defined(GO1_ABLATE_TORQUE) || defined(GO1_ABLATE_PHYSICS) || defined(GO1_NO_DEFERRED_TORQUE)
  const bool deferred = false;
#else
  const bool deferred = mfma_torque && nw == 4 && cfg.decimation <= ACT_MAX_DEC;     // torque model entirely on the helper wavefronts (3 x 64 lanes = the 192 rows)
#endif
  if (mfma_torque && wv == 0) actuator_lds_init(act_lds, lane);
  reward_plan_to_lds(csc->rew, plan_lds, (int)threadIdx.x);
  PROF_INIT
  __syncthreads();
  if (e >= N) return;
  const int nsub = substep_only ? 1 : cfg.decimation;
  if (wv != 0) {           // helper wavefront: the same sequence of workgroup barriers as the master's substep loop
#pragma unroll 1
    for (int sub = 0; sub < nsub; sub++) {
#ifndef GO1_ABLATE_TORQUE
      if (substep_only) {
      } else if (deferred) {
        BLOCK_SYNC(nw);                                   // the master's q, qd are in the row slots
        torque_build_row(acth, Z.act_io(), 64 * (wv - 1) + lane, sub);
        LDS_PHASE();                                      // rows 64 (wv - 1) .. + 63 = tiles 4 (wv - 1) .. + 3: built and consumed by this wavefront
        actuator_tiles(act_lds, Z.act_io(), lane, 4 * (wv - 1), 1, 4 * wv);
        BLOCK_SYNC(nw);                                   // (the master arrives here when it needs the torques)
      } else if (mfma_torque) actuator_net_mfma(act_lds, Z.act_io(), lane, wv, nw, false, nullptr, nullptr);
#endif
#ifndef GO1_ABLATE_PHYSICS
      BLOCK_SYNC(nw);                                     // the master's items and hand-over packets are in LDS
      emit_terrain_contacts(cfg, Z, lane >> 2, 4 * (wv - 1) + (lane & 3), 4 * (nw - 1), cfg.sim_dt);
      BLOCK_SYNC(nw);
#endif
    }
#ifndef GO1_ABLATE_POST
    if (!substep_only) {       // post_physics(): the observations on helper 1 while the master evaluates the rewards (go1_maps.h)
      BLOCK_SYNC(nw);          // S1
      // round 0: the usual step (acth[0], written before S1).  round 1: the late hand-over — a reset kept the observations back until the
      // environment was re-initialised; acth[1] (written before S2) says whether the helpers take them then, and S3 ends that round
#pragma unroll 1
      for (int round = 0; round < 2; round++) {
        if (wv == 1 && acth[round] != 0.f) {
          PROF_DECL
          const int leg_ = lane & 3;
          const V3 pg = v3(AT(B.projected_gravity, 0, e), AT(B.projected_gravity, 1, e), AT(B.projected_gravity, 2, e));
          const float clock_own = AT(B.clock_inputs, leg_, e), force_z = AT(B.contact_forces, 3 * (4 + 4 * leg_) + 2, e);
          uint32_t fault_h = 0;
          post_observations(cfg, B, lds, lane, e, N, A.counter + 1, gravity_at(cfg, A.counter), A.history_slot, fault_h, pg, clock_own, force_z, 1, 0, nw - 1 PROF_PASS);
          report_fault(B, e, fault_h);
        } else if (wv > 1 && acth[round] != 0.f) {       // the other helpers: their share of the measured-heights columns (configurations that observe them)
          PROF_DECL
          uint32_t fault_h = 0;
          post_observations(cfg, B, lds, lane, e, N, A.counter + 1, gravity_at(cfg, A.counter), A.history_slot, fault_h, v3(0.f, 0.f, 0.f), 0.f, 0.f, 4, wv - 1, nw - 1 PROF_PASS);
        }
        BLOCK_SYNC(nw);          // S2 (round 0), S3 (round 1)
        if (acth[1] == 0.f) break;
      }
    }
#endif
    return;
  }
  const float h = cfg.sim_dt;
  PROF_DECL
  Base s;
  Leg L;
  // every load of the prologue is issued before its first store (a store in between would pin the later loads behind
  // it — possible aliasing — and expose one HBM round trip per group)
  load_state(B, leg, e, N, s, L);
  LambdaIn lam_in;
  load_lambda_issue(B, lane, e, N, lam_in);
  uint32_t fault = 0;
  uint32_t drops[GO1_CC_COUNT];
#pragma unroll
  for (int c = 0; c < GO1_CC_COUNT; c++) drops[c] = 0u;
  float act_in[3], fv_in[3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    act_in[jj] = substep_only ? 0.f : A.actions[(size_t)e * 12 + j];
    fv_in[jj] = AT(B.foot_velocities, j, e);
  }
  int ep_len = B.episode_length_buf[e];
  StashIn stash_in;
  if (deferred) torque_stash_issue(cfg, B, leg, e, N, A.lag_head, stash_in);
  // ---- (end of the prologue's load batch) ----
  VALUE_BARRIER(ep_len);
  const bool warm = substep_only ? cfg.warm_start != 0 : (cfg.warm_start && ep_len > 0);
  load_lambda_commit(cfg, lds, lane, lam_in, !warm);
  LDS_PHASE();
  PROF(3);           // the trunk's warm-start impulse is read by all four lanes of the environment
  const V3 grav = gravity_at(cfg, A.counter);
  float act_clipped[3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) act_clipped[jj] = fminf(fmaxf(act_in[jj], -cfg.clip_actions), cfg.clip_actions);
  if (deferred) torque_stash_commit(cfg, B, acth, lane, leg, e, N, A.lag_head, act_clipped, stash_in);
  if (substep_only) {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) L.tau[jj] = AT(B.torques, 3 * leg + jj, e);
  } else {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      AT(B.actions, j, e) = act_clipped[jj];
      AT(B.prev_foot_velocities, j, e) = fv_in[jj];
    }
  }
  {
    float a = 0.f;
    a = nonfinite_acc(a, s.pos.x); a = nonfinite_acc(a, s.pos.y); a = nonfinite_acc(a, s.pos.z);
    a = nonfinite_acc(a, s.qx); a = nonfinite_acc(a, s.qy); a = nonfinite_acc(a, s.qz); a = nonfinite_acc(a, s.qw);
    a = nonfinite_acc(a, s.w.x); a = nonfinite_acc(a, s.w.y); a = nonfinite_acc(a, s.w.z);
    a = nonfinite_acc(a, s.v.x); a = nonfinite_acc(a, s.v.y); a = nonfinite_acc(a, s.v.z);
#pragma unroll
    for (int j = 0; j < 3; j++) { a = nonfinite_acc(a, L.q[j]); a = nonfinite_acc(a, L.qd[j]); }
    if (a != a) fault |= 1u << GO1_FAULT_STATE_IN;
  }
  const int nl = cfg.lag_timesteps + 1;
  int head = A.lag_head;
  PROF(0);
  auto substep = [&](int sub) __attribute__((always_inline)) {
    PROF(24);
#ifndef GO1_ABLATE_TORQUE
    if (substep_only) {
    } else if (deferred) {
      torque_post_state(L, Z.act_io(), lane);
      PROF(22);
      BLOCK_SYNC(nw);
    } else compute_torques(cfg, B, L, leg, e, N, head, act_lds, Z.act_io(), mfma_torque, nw, fault);
#endif
    PROF(1);
    head = (head + 1) % nl;
#ifndef GO1_ABLATE_PHYSICS
    physics_substep<WALLS, SIG, PLANE>(cfg, B, Z, lane, nw, s, L, grav, warm || (cfg.warm_start && sub > 0), h, fault, drops, deferred ? acth : nullptr, e, N, sub PROF_PASS);
#endif
    PROF(30);
  };
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GO1_SUBSTEP_LOOP)
  // The reference's decimation of 4 with the body written out four times (`#pragma unroll` is refused by the optimizer for this loop): no
  // far backward branch at the end of a substep, and `sub` folds into each copy.  Same-box A/B against the loop (gpurun call r5o): 0.1583 /
  // 0.1231 against 0.1626 / 0.1292 ms per env.step (-2.6 % / -4.7 %); code size x 4 (the 16-384 KB probe shows no instruction-cache cliff,
  // profiles/r05_icache_probe.txt).  Price: 352 bytes of scratch per lane instead of 144 (+3 MB of HBM traffic per launch); a loop of two
  // written-out pairs needs 720.  Other decimations — and the host build of the SIMT emulator — take the loop.
  if (nsub == 4) { substep(0); substep(1); substep(2); substep(3); }
  else
#endif
#pragma unroll 1
  for (int sub = 0; sub < nsub; sub++) substep(sub);
  if (deferred) torque_stash_store(cfg, B, L, acth, lane, leg, e, N);
  store_state(B, leg, e, N, s, L);
  foot_state(s, L, leg, B, e, N);
  store_forces(cfg, B, lds, lane, e, N);
  __threadfence_block();
  LDS_PHASE();
  PROF(7);
#ifndef GO1_ABLATE_POST
  if (!substep_only)
    post_physics<PLANE>(cfg, B, plan_lds, lds, lane, e, N, A.counter + 1, grav, A.history_slot, fault, (int)blockIdx.x * EPW >= csc->num_train_envs,
                        nw > 1 ? acth : nullptr, nw PROF_PASS);
#endif
  report_fault(B, e, fault);
  report_drops(B, drops);
#define PROF_LAUNCH ((unsigned)A.counter)
  PROF_FLUSH;
#undef PROF_LAUNCH
}
#define STEP_LDS \
  __shared__ float lds[L_END * EPW]; \
  __shared__ __attribute__((aligned(16))) lf4 ldsx[X_END]; \
  __shared__ __attribute__((aligned(16))) float act_lds[A_END]; \
  __shared__ float acth[AH_END * WAVE];          /* per-lane stash of the deferred torque path (torque_stash_issue / _commit) */ \
  __shared__ int plan_lds[2 * GO1_REW_COUNT];    /* the reward plan (go1_maps.h reward_plan_to_lds) */
// instances: terrain (plane | height field | height field with vertical faces) x (plain | contact signature recorded: parity tests)
#define STEP_KERNEL(name, WALLS, SIG, PLANE) \
  extern "C" __global__ void __launch_bounds__(WAVE * STEP_WAVES) name(const StepArgs A) { STEP_LDS step_body<WALLS, SIG, PLANE>(A, lds, ldsx, act_lds, acth, plan_lds); }
STEP_KERNEL(go1_step_kernel, false, false, true)
STEP_KERNEL(go1_step_kernel_hf, false, false, false)
STEP_KERNEL(go1_step_kernel_walls, true, false, false)
STEP_KERNEL(go1_step_kernel_sig, false, true, true)
STEP_KERNEL(go1_step_kernel_hf_sig, false, true, false)
STEP_KERNEL(go1_step_kernel_walls_sig, true, true, false)

// piecewise entry points with the 4-lane mapping (parity tests): torques only / tensor maps only (a single physics substep is
// mode 2 of the step kernels: the production structure, master + helper wavefronts)
extern "C" __global__ void __launch_bounds__(WAVE) go1_aux_kernel(const StepArgs A) {
  __shared__ float lds[L_END * EPW];
  __shared__ __attribute__((aligned(16))) float act_io[A_IO_END];
  __shared__ __attribute__((aligned(16))) float act_lds[A_END];
  __shared__ int plan_lds[2 * GO1_REW_COUNT];
  for (int i = threadIdx.x; i < L_END * EPW; i += WAVE) lds[i] = 0.f;
  const GO1_CONSTANT SimConst* csc = (const GO1_CONSTANT SimConst*)(uintptr_t)A.sc;
  reward_plan_to_lds(csc->rew, plan_lds, (int)threadIdx.x);
  LDS_PHASE();
  CfgRef cfg = WAVE_CFG(csc, (int)blockIdx.x * EPW);
  BufRef B = csc->buf;
  const int N = cfg.num_envs;
  const int lane = threadIdx.x, leg = lane & 3;
  const int e = blockIdx.x * EPW + (lane >> 2);
  const bool full_wave = (int)(blockIdx.x + 1) * EPW <= N;
  if (A.mode == 1 && full_wave && cfg.control_type == 1) actuator_lds_init(act_lds, lane);
  if (e >= N) return;
  if (A.mode == 4) {       // tensor maps only
    PROF_DECL
    uint32_t fault = 0;
    post_physics(cfg, B, plan_lds, lds, lane, e, N, A.counter + 1, v3(A.gravity_override[0], A.gravity_override[1], A.gravity_override[2]), A.history_slot, fault, (int)blockIdx.x * EPW >= csc->num_train_envs,
                 nullptr, 1 PROF_PASS);
    report_fault(B, e, fault);
    return;
  }
  Base s;
  Leg L;
  load_state(B, leg, e, N, s, L);
  LambdaIn lam_in;
  load_lambda_issue(B, lane, e, N, lam_in);
  uint32_t fault = 0;
  if (A.mode == 1) {       // torques only (actions given as SoA)
#pragma unroll
    for (int jj = 0; jj < 3; jj++) AT(B.actions, 3 * leg + jj, e) = AT(A.actions, 3 * leg + jj, e);
    compute_torques(cfg, B, L, leg, e, N, A.lag_head, act_lds, act_io, full_wave, 1, fault);
    report_fault(B, e, fault);
    return;
  }
}

// one environment per lane: reset_idx
extern "C" __global__ void __launch_bounds__(WAVE) go1_env_kernel(const StepArgs A) {
  const GO1_CONSTANT SimConst* csc = (const GO1_CONSTANT SimConst*)(uintptr_t)A.sc;
  BufRef B = csc->buf;
  const int N = csc->cfg.num_envs;
  const int i = blockIdx.x * WAVE + threadIdx.x;
  if (i < A.n_ids) {
    const int e = A.ids ? A.ids[i] : i;
    const bool ev = e >= csc->num_train_envs;              // per lane here: the ids are arbitrary
    CfgRef cfg = ev ? csc->cfg_eval : csc->cfg;
    reset_env(cfg, B, e, N, A.counter, ev, A.counter);
  }
}

// HistoryWrapper.get_observations: append the current obs_buf to the double-length ring
extern "C" __global__ void __launch_bounds__(256) go1_history_kernel(const SimConst* __restrict__ sc, int slot) {
  const GO1_CONSTANT SimConst* csc = (const GO1_CONSTANT SimConst*)(uintptr_t)sc;
  CfgRef cfg = csc->cfg;
  BufRef B = csc->buf;
  const int no = cfg.num_obs, R = cfg.num_obs_history + 1;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)cfg.num_envs * no) return;
  const size_t e = i / no, c = i % no;
  float v = B.obs_buf[i];
  float* row = B.obs_history + e * 2 * R * no;
  row[(size_t)slot * no + c] = v;
  row[(size_t)(slot + R) * no + c] = v;
}

// curriculum weight update + CDF rebuild (reference curriculum.py:135-154): one workgroup per category.  The successes of the
// last curriculum_update_interval steps sit in one slot per step; their updates are applied in step order (each the
// reference's per-step rule: +0.2 on a bin with a success and +0.2 per successful environment on the bins of its
// neighbourhood, clipped at 1), then the CDF is rebuilt once.
extern "C" __global__ void __launch_bounds__(256) go1_curriculum_kernel(const SimConst* __restrict__ sc) {
  const GO1_CONSTANT SimConst* csc = (const GO1_CONSTANT SimConst*)(uintptr_t)sc;
  CfgRef cfg = csc->cfg;
  BufRef B = csc->buf;
  __shared__ float part[256];
  const int c = blockIdx.x, nb = cfg.num_bins, t = threadIdx.x;
  float* w = B.curriculum_weights + (size_t)c * nb;
  float* cdf = B.curriculum_cdf + (size_t)c * nb;
  const int per = (nb + 255) / 256;
  for (int slot = 0; slot < cfg.curriculum_update_interval; slot++) {
    int32_t* s = B.curriculum_success + ((size_t)slot * cfg.num_categories + c) * nb;
    // increments (read all successes of the slot before anyone clears them)
    for (int i = 0; i < per; i++) {
      int b = t * per + i;
      if (b < nb) {
        int cnt = s[b] > 0 ? 1 : 0;
        for (int p = B.curriculum_nbr_ptr[b]; p < B.curriculum_nbr_ptr[b + 1]; p++) cnt += s[B.curriculum_nbr_idx[p]];
        cdf[b] = fminf(1.0f, w[b] + 0.2f * cnt);        // stage the new weight in the cdf array until every thread has read `s`
      }
    }
    __syncthreads();
    for (int i = 0; i < per; i++) {
      int b = t * per + i;
      if (b < nb) { w[b] = cdf[b]; s[b] = 0; }
    }
    __syncthreads();
  }
  float local = 0.f;
  for (int i = 0; i < per; i++) {
    int b = t * per + i;
    if (b < nb) local += w[b];
  }
  // block prefix sum of the per-thread totals
  part[t] = local;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    float v = (t >= off) ? part[t - off] : 0.f;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const float total = part[255];
  float run = (t > 0) ? part[t - 1] : 0.f;
  for (int i = 0; i < per; i++) {
    int b = t * per + i;
    if (b < nb) { run += w[b]; cdf[b] = run / total; }
  }
}

// ================================================================================================
// C-ABI
// ================================================================================================
struct Go1Sim {
  Go1SimConfig cfg;
  Go1SimConfig cfg_eval;
  int32_t num_train;   // == cfg.num_envs: no evaluation environments
  Go1SimBuffers buf;
  int device;
  int64_t counter;
  int32_t lag_head;
  int32_t history_slot;
  SimConst* dconst;    // device copy of {cfg, buf}
  int timing_cap;
  int64_t timing_n;
  hipEvent_t* ev;      // 2 * timing_cap
};

static int check_cfg(const Go1SimConfig* cfg) {
  if (!cfg || cfg->abi_version != GO1SIM_ABI_VERSION) return -2;
  if (cfg->num_envs <= 0 || cfg->lag_timesteps + 1 > GO1_MAX_LAG) return -3;
  if (cfg->curriculum_update_interval < 1 || cfg->curriculum_update_interval > GO1_MAX_CURRICULUM_INTERVAL) return -3;
  const int scan = cfg->observe_heights ? cfg->num_height_x * cfg->num_height_y : 0;
  if (cfg->num_obs - scan > GO1_MAX_OBS || cfg->num_privileged_obs > GO1_MAX_PRIV_OBS || cfg->num_rewards > GO1_MAX_REWARDS) return -4;
  if (cfg->terrain_type != 0 && (cfg->hf_rows < 2 || cfg->hf_cols < 2)) return -5;
  if (cfg->num_height_x > GO1_MAX_HEIGHT_AXIS || cfg->num_height_y > GO1_MAX_HEIGHT_AXIS) return -4;
  return 0;
}

static int upload_const(Go1Sim* s) {
  SimConst h;
  h.cfg = s->cfg; h.buf = s->buf;
  const bool split = s->num_train > 0 && s->num_train < s->cfg.num_envs;
  h.cfg_eval = split ? s->cfg_eval : s->cfg;
  h.num_train_envs = split ? s->num_train : s->cfg.num_envs;
  for (int id = 0; id < GO1_REW_COUNT; id++) { h.rew.kx_by_id[id] = -1; h.rew.scale_by_id[id] = 0.f; }
  for (int kx = 0; kx < s->cfg.num_rewards; kx++) {
    const int id = s->cfg.reward_ids[kx];
    if (id >= 0 && id < GO1_REW_COUNT) { h.rew.kx_by_id[id] = kx; h.rew.scale_by_id[id] = s->cfg.reward_scales[kx]; }
  }
  return hipMemcpy(s->dconst, &h, sizeof(SimConst), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

extern "C" int go1sim_create(const Go1SimConfig* cfg, const Go1SimBuffers* buffers, int device, Go1Sim** out) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!buffers || !out) return -1;
  if (hipSetDevice(device) != hipSuccess) return -10;
  Go1Sim* s = new Go1Sim();
  s->cfg = *cfg; s->buf = *buffers; s->device = device;
  s->cfg_eval = *cfg; s->num_train = cfg->num_envs;
  s->counter = 0; s->lag_head = 0; s->history_slot = 0; s->timing_cap = 0; s->timing_n = 0; s->ev = nullptr;
  s->dconst = nullptr;
  if (hipMalloc((void**)&s->dconst, sizeof(SimConst)) != hipSuccess) { delete s; return -12; }
  if (upload_const(s) != 0) { (void)hipFree(s->dconst); delete s; return -13; }
  *out = s;
  return 0;
}
extern "C" int go1sim_destroy(Go1Sim* s) {
  if (!s) return -1;
  for (int i = 0; i < 2 * s->timing_cap; i++) (void)hipEventDestroy(s->ev[i]);
  delete[] s->ev;
  (void)hipFree(s->dconst);
  delete s;
  return 0;
}
extern "C" int go1sim_set_config(Go1Sim* s, const Go1SimConfig* cfg) {
  if (!s) return -1;
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (cfg->num_envs != s->cfg.num_envs) return -6;
  // curriculum_success was allocated by the caller as [K][categories][bins] for the creation-time K: a larger K would index past it
  if (cfg->curriculum_update_interval != s->cfg.curriculum_update_interval || cfg->num_categories != s->cfg.num_categories ||
      cfg->num_bins != s->cfg.num_bins)
    return -6;
  s->cfg = *cfg;
  return upload_const(s);      // blocking copy: configuration changes are rare and never on the step path
}
extern "C" int go1sim_set_eval_config(Go1Sim* s, const Go1SimConfig* cfg, int32_t num_train_envs) {
  if (!s) return -1;
  int rc = check_cfg(cfg);
  if (rc) return rc;
  const Go1SimConfig& t = s->cfg;
  if (cfg->num_envs != t.num_envs || cfg->num_obs != t.num_obs || cfg->num_privileged_obs != t.num_privileged_obs ||
      cfg->num_obs_history != t.num_obs_history || cfg->num_rewards != t.num_rewards || cfg->decimation != t.decimation ||
      cfg->lag_timesteps != t.lag_timesteps || cfg->control_type != t.control_type || cfg->num_commands != t.num_commands ||
      cfg->num_bins != t.num_bins || cfg->num_categories != t.num_categories || cfg->terrain_type != t.terrain_type ||
      cfg->curriculum_update_interval != t.curriculum_update_interval ||
      cfg->seed != t.seed || cfg->env_id_offset != t.env_id_offset)
    return -6;
  if (num_train_envs <= 0 || num_train_envs > t.num_envs || (num_train_envs < t.num_envs && (num_train_envs % EPW) != 0)) return -3;
  s->cfg_eval = *cfg;
  s->num_train = num_train_envs;
  return upload_const(s);
}

static int launch(Go1Sim* s, int mode, const float* actions, const int32_t* ids, int n_ids, hipStream_t st, bool timed,
                  const float* grav = nullptr) {
  StepArgs A;
  for (int i = 0; i < 3; i++) A.gravity_override[i] = grav ? grav[i] : 0.f;
  A.sc = s->dconst; A.actions = actions; A.counter = s->counter; A.lag_head = s->lag_head;
  A.history_slot = s->history_slot; A.mode = mode; A.ids = ids; A.n_ids = n_ids;
  const int n = (mode == 3) ? n_ids : s->cfg.num_envs;
  const int per_block = (mode == 3) ? WAVE : EPW;
  dim3 grid((n + per_block - 1) / per_block), block((mode == 0 || mode == 2) ? WAVE * STEP_WAVES : WAVE);
  const int slot = timed ? (int)(s->timing_n % s->timing_cap) : 0;
  if (timed) (void)hipEventRecord(s->ev[2 * slot], st);
  if (mode == 0 || mode == 2) {
    static const bool force_hf = getenv("GO1_FORCE_HF_INSTANCE") != nullptr;        // A/B of the plane instance (tools/variant_bench.sh)
    const bool hf = force_hf || (s->cfg.terrain_type != 0 && s->buf.height_samples != nullptr), walls = hf && s->cfg.terrain_type != 0 && s->cfg.hf_wall_units > 0;
    const bool sig = s->buf.contact_signature != nullptr;
    if (walls) { if (sig) hipLaunchKernelGGL(go1_step_kernel_walls_sig, grid, block, 0, st, A); else hipLaunchKernelGGL(go1_step_kernel_walls, grid, block, 0, st, A); }
    else if (hf) { if (sig) hipLaunchKernelGGL(go1_step_kernel_hf_sig, grid, block, 0, st, A); else hipLaunchKernelGGL(go1_step_kernel_hf, grid, block, 0, st, A); }
    else { if (sig) hipLaunchKernelGGL(go1_step_kernel_sig, grid, block, 0, st, A); else hipLaunchKernelGGL(go1_step_kernel, grid, block, 0, st, A); }
  }
  else if (mode == 3) hipLaunchKernelGGL(go1_env_kernel, grid, block, 0, st, A);
  else hipLaunchKernelGGL(go1_aux_kernel, grid, block, 0, st, A);
  if (timed) { (void)hipEventRecord(s->ev[2 * slot + 1], st); s->timing_n++; }
  return hipGetLastError() == hipSuccess ? 0 : -20;
}

extern "C" int go1sim_step(Go1Sim* s, const float* actions, void* stream) {
  if (!s || !actions) return -1;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch(s, 0, actions, nullptr, 0, st, s->timing_cap > 0);
  if (rc) return rc;
  s->counter += 1;
  s->lag_head = (s->lag_head + s->cfg.decimation) % (s->cfg.lag_timesteps + 1);
  s->history_slot = (s->history_slot + 1) % (s->cfg.num_obs_history + 1);
  if (s->cfg.device_curriculum && s->buf.curriculum_weights && !s->cfg.defer_curriculum_update && s->counter % s->cfg.curriculum_update_interval == 0) {
    hipLaunchKernelGGL(go1_curriculum_kernel, dim3(s->cfg.num_categories), dim3(256), 0, st, (const SimConst*)s->dconst);
    if (hipGetLastError() != hipSuccess) return -21;
  }
  return 0;
}
extern "C" int go1sim_reset_idx(Go1Sim* s, const int32_t* ids, int32_t n, void* stream) {
  if (!s) return -1;
  hipStream_t st = (hipStream_t)stream;
  int cnt = ids ? n : s->cfg.num_envs;
  if (cnt <= 0) return 0;
  return launch(s, 3, nullptr, ids, cnt, st, false);
}
extern "C" int go1sim_compute_torques(Go1Sim* s, const float* actions_soa, void* stream) {
  if (!s || !actions_soa) return -1;
  int rc = launch(s, 1, actions_soa, nullptr, 0, (hipStream_t)stream, false);
  s->lag_head = (s->lag_head + 1) % (s->cfg.lag_timesteps + 1);
  return rc;
}
extern "C" int go1sim_physics_substep(Go1Sim* s, void* stream) {
  if (!s) return -1;
  return launch(s, 2, nullptr, nullptr, 0, (hipStream_t)stream, false);
}
extern "C" int go1sim_curriculum_update(Go1Sim* s, void* stream) {
  if (!s) return -1;
  hipLaunchKernelGGL(go1_curriculum_kernel, dim3(s->cfg.num_categories), dim3(256), 0, (hipStream_t)stream, (const SimConst*)s->dconst);
  return hipGetLastError() == hipSuccess ? 0 : -21;
}
extern "C" int go1sim_post_physics(Go1Sim* s, const float* gravity, void* stream) {
  if (!s || !gravity) return -1;
  int rc = launch(s, 4, nullptr, nullptr, 0, (hipStream_t)stream, false, gravity);
  s->counter += 1;
  s->history_slot = (s->history_slot + 1) % (s->cfg.num_obs_history + 1);
  return rc;
}
extern "C" int go1sim_append_history(Go1Sim* s, void* stream) {
  if (!s || !s->buf.obs_history) return -1;
  size_t total = (size_t)s->cfg.num_envs * s->cfg.num_obs;
  hipLaunchKernelGGL(go1_history_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const SimConst*)s->dconst, s->history_slot);
  s->history_slot = (s->history_slot + 1) % (s->cfg.num_obs_history + 1);
  return hipGetLastError() == hipSuccess ? 0 : -22;
}
extern "C" int go1sim_history_window_offset(Go1Sim* s, int32_t* off) {
  if (!s || !off) return -1;
  *off = ((s->history_slot + 1) % (s->cfg.num_obs_history + 1)) * s->cfg.num_obs;   // skip the spare (next-write) slot
  return 0;
}
extern "C" int go1sim_get_counters(Go1Sim* s, int64_t* c, int32_t* h) {
  if (!s) return -1;
  if (c) *c = s->counter;
  if (h) *h = s->lag_head;
  return 0;
}
extern "C" int go1sim_set_counters(Go1Sim* s, int64_t c, int32_t h) {
  if (!s) return -1;
  s->counter = c; s->lag_head = h;
  return 0;
}
extern "C" int go1sim_enable_timing(Go1Sim* s, int capacity) {
  if (!s || capacity < 0) return -1;
  for (int i = 0; i < 2 * s->timing_cap; i++) (void)hipEventDestroy(s->ev[i]);
  delete[] s->ev;
  s->ev = nullptr; s->timing_cap = 0; s->timing_n = 0;
  if (capacity > 0) {
    s->ev = new hipEvent_t[2 * capacity];
    for (int i = 0; i < 2 * capacity; i++)
      if (hipEventCreate(&s->ev[i]) != hipSuccess) return -30;
    s->timing_cap = capacity;
  }
  return 0;
}
extern "C" int go1sim_read_timings(Go1Sim* s, float* ms, int32_t max, int32_t* count) {
  if (!s || !ms || !count) return -1;
  int64_t have = s->timing_n < s->timing_cap ? s->timing_n : s->timing_cap;
  if (have > max) have = max;
  for (int64_t k = 0; k < have; k++) {
    int slot = (int)((s->timing_n - have + k) % s->timing_cap);
    if (hipEventSynchronize(s->ev[2 * slot + 1]) != hipSuccess) return -31;
    if (hipEventElapsedTime(&ms[k], s->ev[2 * slot], s->ev[2 * slot + 1]) != hipSuccess) return -32;
  }
  *count = (int32_t)have;
  return 0;
}
#ifndef GO1_SOURCE_HASH
#define GO1_SOURCE_HASH "unstamped"      // __graft_entry__.build_hip passes the sha256 of the sources + flags; smoke() / tests/test_abi.py compare
#endif
extern "C" const char* go1sim_version(void) { return "go1sim 0.6 (gfx950, abi " GO1SIM_STR(GO1SIM_ABI_VERSION) ", 4 lanes/env) go1-src:" GO1_SOURCE_HASH; }

#ifdef GO1_PROFILE
// debug build only (tools/phase_profile.py): read and clear the per-phase cycle accumulators of workgroup 0, lane 0
extern "C" int go1sim_debug_read_profile(unsigned long long* out64) {
  unsigned long long z[64] = {0};
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
// the phase accumulators of every workgroup, [1024][40] (read and clear)
extern "C" int go1sim_debug_read_wg_phases(unsigned long long* out) {
  static unsigned long long z[1024 * 40];
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_profw), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_profw), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
// per-workgroup totals of the master wavefront [0, 1024), per-launch maxima [1024, 1088) and sums [1088, 1152) (read and clear)
extern "C" int go1sim_debug_read_wg_times(unsigned long long* out1152) {
  static unsigned long long z[1024];
  if (hipMemcpyFromSymbol(out1152, HIP_SYMBOL(g_wgt), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out1152 + 1024, HIP_SYMBOL(g_lmax), 64 * 8) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out1152 + 1088, HIP_SYMBOL(g_lsum), 64 * 8) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_lmax), z, 64 * 8) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(g_lsum), z, 64 * 8) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wgt), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
