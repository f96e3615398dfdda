// go1_physics.h — torque model + one physics substep, FOUR LANES PER ENVIRONMENT (one lane per leg).
//
// Wavefront = 16 environments x 4 legs (lane = 4*env_local + leg).  The four legs of the Go1 are independent
// sub-trees hanging off the floating base, so every O(n_dof) recursion splits four ways:
//   * each lane keeps ITS leg's chain in registers: joint state, motion subspaces S_j, ABA factors U_j, 1/D_j, u_j
//     (static indexing, no LDS / scratch traffic for the chain);
//   * the base (6x6 articulated inertia, its inverse, twist, pose) is replicated in the four lanes; the only
//     cross-lane traffic is quad reductions done with DPP quad_perm moves (no LDS, no barriers):
//     27 floats after ABA pass 2, 6 floats per Delassus column / impulse application, 3 per PGS row;
//   * what is genuinely shared per environment — the <= 6-contact list, the 18x18 Delassus matrix, impulses —
//     lives in LDS as [field][env_local] (a quad reads one address: broadcast; 16 envs -> 16 banks).
// Replaces gym.simulate (reference legged_robot.py:76-80) and _compute_torques (:907-946).  Same contract and
// solver order as oracle/go1_oracle.c.
#pragma once
#include "go1_maps.h"

#define GO1_CONST static __device__ __constant__ const
#define GO1_REAL float
#include "go1_model_data.h"
#include "go1_actuator_data.h"

#define WAVE 64
#define EPW 16                       // environments per wavefront
#define MAXC GO1_MAX_CONTACTS        // solver contacts per env (24; oracle: the same constant of include/go1sim.h)
#define NRJ 12                       // joint-limit rows: joint j
#define SELF_TYPES 6                  // capsule combinations of a pair of legs (oracle: GO1_SELF_TYPES)
#define SELF_LEGLEG (6 * SELF_TYPES)  // leg-leg pair ids [0, 36); 36 + leg: lower leg against the trunk
#define MAXSB 6                      // leg-leg self-contacts per env: one per pair of legs (the deepest of its four capsule combinations)
#define MAXTR 4                      // trunk corners per env (oracle: GO1_MAX_TRUNK_POINTS)
#define GO1_LIMIT_RECOVERY_RATE 10.0f   // rad/s: a joint found beyond a stop is brought back at a bounded rate
#define GO1_LIMIT_SAFETY 2.0f           // x velocity limit: beyond this the limit rows have failed (cut + fault count)
#define GO1_LIMIT_SLACK 0.2f            // rad beyond a stop: same
#define GO1_SELF_LEG_RADIUS ((float)GO1_FOOT_RADIUS)
#define GO1_SELF_THIGH_RADIUS 0.017f
typedef __attribute__((ext_vector_type(4))) float lf4;

// ---- LDS map of the solver ------------------------------------------------------------------------------------------------
// The solve is MATRIX-FREE.  With the ABA factors of the substep, M^-1 factorises: for a row r (a contact direction or a
// joint coordinate) the backward pass of a unit impulse leaves the wrench g_r arriving at the base and the joint residuals
// u_j(r) of the row's leg, and
//     W[r][c] = g_r . I0^-1 g_c + [same leg] sum_j u_j(r) u_j(c) / D_j  =  a_r . a_c,
//     a_r = [ L^-1 g_r (6) ; u_j(r) / sqrt(D_j) (3, in the slots of the row's leg) ],   I0 = L L^T.
// The sweep keeps s = sum_r lambda_r a_r — 6 base components replicated in the four lanes of the environment, 3 leg
// components in the lane of the leg — and evaluates a row's velocity as b_r + a_r . s (9 FMAs) instead of reading a row of an
// explicit Delassus matrix: no O(rows^2) build, no (rows x rows) LDS block, and the number of contacts is only bounded by the
// 40-float record each one keeps in LDS (24 per environment; round 2: 8, with an 83 KB matrix).
//   lds  : per-environment scalars, lds[field * EPW + env_local]
//   cr   : contact records, 12 x 16-byte slots per contact: cr[(k * CR_Q + q) * EPW + env_local]  (a quad reads one address,
//          the 16 environments of the wavefront 256 contiguous bytes: conflict-free)
//          q0 q1 q2  normal row: a_n (6 + 3), c_n = b_n - v*, 1/W_nn, W_t1n          q3 q4 q5  first tangent: a_t1, b_t1, 1/W_t1t1, W_t2n
//          q6 q7 q8  second tangent: a_t2, b_t2, 1/W_t2t2, flags   (rows start on 16-byte slots: their components sit in even-aligned
//          register pairs after the load, which is what the packed-f32 arithmetic of the sweep wants)
//          q9  lambda_n, lambda_t1, lambda_t2          q10 contact point (rel. base origin)   q11 normal
//          before the emission q9..q11 hold the ITEM the master posted: (phi, x) (u_n, n) (depth, leg, body, share)
//   jr   : joint-limit rows, 4 slots per joint: a (6 + 3), b, lower, upper, W, 1/W, lambda
//   sb   : body-B side of the leg-leg self-contacts (exchange between the two lanes of a pair, then u_B of the three rows)
//   pkl / pke : the emission hand-over — per lane the leg's ABA factors, per environment the free base twist and L^-1
//   seg  : lower-leg / thigh segments and body twists of every leg (self-collision)
enum {
  L_LAM = 0,                 // 17 x world impulse (x, y, z) per reported body: solver output / warm start
  L_KL = 51,                 // +0 contacts listed, +1 mask of the legs with limit rows, +2 contacts listed in the previous substep,
                             // +3 / +6 != 0: an emitting lane met a non-finite normal / a degenerate row (faults), +4 slot of the
                             // first self-contact, +5 number of self-contacts
  L_END = GO1_MAX_OBS        // (post_physics stages the observation rows here)
};
#define LDS(f) lds[(f) * EPW + el]
#define CR_Q 12
#define JR_Q 4
#define SB_Q 8
#define PKL_Q 12
#define PKE_Q 9
#define SEG_Q 6
#define TW_Q 6
enum {
  X_CR = 0, X_JR = X_CR + MAXC * CR_Q * EPW, X_SB = X_JR + NRJ * JR_Q * EPW, X_PKL = X_SB + MAXSB * SB_Q * EPW,
  X_PKE = X_PKL + PKL_Q * WAVE, X_SEG = X_PKE + PKE_Q * EPW, X_TW = X_SEG + SEG_Q * WAVE, X_END = X_TW + TW_Q * WAVE       // (in lf4 units)
};
#define CRQ(k, q) crl[((k) * CR_Q + (q)) * EPW + el]
#define JRQ(j, q) jrl[((j) * JR_Q + (q)) * EPW + el]
#define SBQ(i, q) sbl[((i) * SB_Q + (q)) * EPW + el]
struct SolverLds {
  float* lds;            // L_END * EPW floats
  lf4* x;                // X_END lf4
  DEV lf4* cr() const { return x + X_CR; }
  DEV lf4* jr() const { return x + X_JR; }
  DEV lf4* sb() const { return x + X_SB; }
  DEV lf4* pkl() const { return x + X_PKL; }
  DEV lf4* pke() const { return x + X_PKE; }
  DEV lf4* seg() const { return x + X_SEG; }
  DEV lf4* tw() const { return x + X_TW; }
  DEV float* act_io() const { return reinterpret_cast<float*>(x + X_PKL); }        // the actuator network's transient rows (overlay)
};

// ================================================================================================
// torque model (reference legged_robot.py:907-946): the calling lane handles the 3 joints of its leg
// ================================================================================================
DEV float softsign(float x) { return x * __builtin_amdgcn_rcpf(1.f + fabsf(x)); }   // v_rcp_f32: <= 1 ulp

// 6->32->32->1 actuator network for the three joints of one leg: every weight (wave-uniform, scalar loads from
// constant memory) feeds 3 independent accumulation chains, hiding the dependent-FMA latency of a single chain.
DEV void actuator_net3(const float in[3][6], float out[3]) {
  float h0[3][32];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    float a0 = GO1_ACT_B0[i], a1 = a0, a2 = a0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const float w = GO1_ACT_W0[i][k];
      a0 = fmaf(w, in[0][k], a0); a1 = fmaf(w, in[1][k], a1); a2 = fmaf(w, in[2][k], a2);
    }
    h0[0][i] = softsign(a0); h0[1][i] = softsign(a1); h0[2][i] = softsign(a2);
  }
  float o0 = GO1_ACT_B2, o1 = o0, o2 = o0;
#pragma unroll 2
  for (int i = 0; i < 32; i++) {
    float a0 = GO1_ACT_B1[i], a1 = a0, a2 = a0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
      const float w = GO1_ACT_W1[i][k];
      a0 = fmaf(w, h0[0][k], a0); a1 = fmaf(w, h0[1][k], a1); a2 = fmaf(w, h0[2][k], a2);
    }
    const float w2 = GO1_ACT_W2[i];
    o0 = fmaf(w2, softsign(a0), o0); o1 = fmaf(w2, softsign(a1), o1); o2 = fmaf(w2, softsign(a2), o2);
  }
  out[0] = o0; out[1] = o1; out[2] = o2;
}

// ---- the same network on the matrix cores ------------------------------------------------------------------
// A wavefront evaluates 16 envs x 12 joints = 192 rows per substep.  The 32x32 hidden layer (2/3 of the FLOPs) runs
// as H1^T = W1 . H0^T on v_mfma_f32_16x16x32_f16 with both operands split hi + lo in fp16 (3 MFMAs per 16x16 tile:
// hi.hi + hi.lo + lo.hi, fp32 accumulate).  hi + lo carries 22 mantissa bits and the dropped lo.lo term is 2^-22
// relative: measured against the fp64 network the torque error is 4e-6 N m max — the same as evaluating the network in
// plain fp32 (3.5e-6; a bf16 split, 16 bits, gave 1.1e-4), so the matrix-core path is fp32-equivalent and the torque
// parity tests use ONE tolerance for both paths (tests/test_gpu_env.py).  Data distribution, chosen so that nothing needs a
// transpose: the 6 inputs of every row go through LDS (8 floats per row); lane (c = lane & 15, g = lane >> 4)
// evaluates the first layer for row 16 t + c and hidden units 8 g .. 8 g + 7 — exactly its B fragment of tile t;
// the W1 fragments (A operand) are converted once per launch and parked in LDS; the MFMA result leaves every lane
// with hidden units {16 i + 4 g + q} of ONE row, so the output layer is 8 FMAs + a 4-lane butterfly.
// Only used by full wavefronts (all 64 lanes alive): a partial last workgroup takes actuator_net3.
typedef __attribute__((ext_vector_type(8))) _Float16 act_f16x8;
typedef __attribute__((ext_vector_type(4))) float act_f32x4;
enum { A_IN = 0, A_OUT = 192 * 8, A_IO_END = A_OUT + 4 * 192 };      // transient rows in / partial sums out: overlaid on the solver's matrix
enum { A_W0 = 0, A_B1 = A_W0 + 32 * 8, A_W2 = A_B1 + 32, A_WF = A_W2 + 32, A_END = A_WF + 4 * 64 * 4 };      // constants of the launch

DEV void actuator_lds_init(float* a, int lane) {          // once per launch, all 64 lanes
  for (int i = lane; i < 32 * 8; i += WAVE) {
    const int k = i >> 3, c = i & 7;
    a[A_W0 + i] = c < 6 ? GO1_ACT_W0[k][c] : (c == 6 ? GO1_ACT_B0[k] : 0.f);
  }
  if (lane < 32) { a[A_B1 + lane] = GO1_ACT_B1[lane]; a[A_W2 + lane] = GO1_ACT_W2[lane]; }
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    act_f16x8 hi, lo;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const float w = GO1_ACT_W1[16 * i + c][8 * g + kk];
      const _Float16 h = (_Float16)w;
      hi[kk] = h;
      lo[kk] = (_Float16)(w - (float)h);
    }
    *reinterpret_cast<act_f16x8*>(a + A_WF + ((2 * i) * 64 + lane) * 4) = hi;
    *reinterpret_cast<act_f16x8*>(a + A_WF + ((2 * i + 1) * 64 + lane) * 4) = lo;
  }
}

// The 12 row tiles of a substep are split over wavefronts: `actuator_tiles` evaluates tiles t = t0, t0 + ts, ... from the
// 192 input rows in io[A_IN] and leaves the four partial sums of every row in io[A_OUT].
// (t1: one past the last tile; ts: stride)
DEV void actuator_tiles(const float* a, float* io, int lane, int t0, int ts, int t1 = 12) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const int c = lane & 15, g = lane >> 4;
  float w0[8][7];
#pragma unroll
  for (int kk = 0; kk < 8; kk++) {
    const f4* p = reinterpret_cast<const f4*>(a + A_W0 + (8 * g + kk) * 8);
    const f4 u = p[0], v = p[1];
    w0[kk][0] = u[0]; w0[kk][1] = u[1]; w0[kk][2] = u[2]; w0[kk][3] = u[3]; w0[kk][4] = v[0]; w0[kk][5] = v[1]; w0[kk][6] = v[2];
  }
  act_f16x8 whi[2], wlo[2];
  float b1v[8], w2v[8];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    whi[i] = *reinterpret_cast<const act_f16x8*>(a + A_WF + ((2 * i) * 64 + lane) * 4);
    wlo[i] = *reinterpret_cast<const act_f16x8*>(a + A_WF + ((2 * i + 1) * 64 + lane) * 4);
    const f4 bb = *reinterpret_cast<const f4*>(a + A_B1 + 16 * i + 4 * g), ww = *reinterpret_cast<const f4*>(a + A_W2 + 16 * i + 4 * g);
#pragma unroll
    for (int q = 0; q < 4; q++) { b1v[4 * i + q] = bb[q]; w2v[4 * i + q] = ww[q]; }
  }
#pragma unroll 1
  for (int t = t0; t < t1; t += ts) {
    const f4* pin = reinterpret_cast<const f4*>(io + A_IN + (16 * t + c) * 8);
    const f4 x0 = pin[0], x1 = pin[1];
    act_f16x8 bhi, blo;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      float s0 = w0[kk][6];
      s0 = fmaf(w0[kk][0], x0[0], s0); s0 = fmaf(w0[kk][1], x0[1], s0); s0 = fmaf(w0[kk][2], x0[2], s0);
      s0 = fmaf(w0[kk][3], x0[3], s0); s0 = fmaf(w0[kk][4], x1[0], s0); s0 = fmaf(w0[kk][5], x1[1], s0);
      float h = softsign(s0);
      VALUE_BARRIER(h);
      const _Float16 hh = (_Float16)h;
      bhi[kk] = hh;
      blo[kk] = (_Float16)(h - (float)hh);
    }
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      act_f32x4 acc = (act_f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[i], bhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[i], blo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[i], bhi, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) part = fmaf(w2v[4 * i + q], softsign(acc[q] + b1v[4 * i + q]), part);
    }
    io[A_OUT + 192 * g + 16 * t + c] = part;      // the 4 lanes (g = 0..3) holding the same row: partials meet in LDS
  }
}
DEV void actuator_publish(float* io, int lane, const float in[3][6]) {       // the calling lane's three input rows
  typedef __attribute__((ext_vector_type(4))) float f4;
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    f4* p = reinterpret_cast<f4*>(io + A_IN + (3 * lane + jj) * 8);
    p[0] = (f4){in[jj][0], in[jj][1], in[jj][2], in[jj][3]};
    p[1] = (f4){in[jj][4], in[jj][5], 1.f, 0.f};
  }
}
DEV void actuator_collect(const float* io, int lane, float out[3]) {
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int r = 3 * lane + jj;
    out[jj] = ((io[A_OUT + r] + io[A_OUT + 192 + r]) + (io[A_OUT + 384 + r] + io[A_OUT + 576 + r])) + GO1_ACT_B2;
  }
}
// Everything in one call (the piecewise entry points and workgroups of one wavefront): wave wv of nw takes tiles wv, wv + nw, ...
DEV void actuator_net_mfma(float* a, float* io, int lane, int wv, int nw, bool master, const float in[3][6], float out[3]) {
  if (master) actuator_publish(io, lane, in);
  BLOCK_SYNC(nw);
  actuator_tiles(a, io, lane, wv, nw);
  BLOCK_SYNC(nw);
  if (master) actuator_collect(io, lane, out);
}

struct Leg {             // the calling lane's leg
  float q[3], qd[3], tau[3];
};

DEV void compute_torques(CfgRef cfg, BufRef B, Leg& L, int leg, int e, int N, int head, float* act_lds, float* act_io, bool full_wave, int nw, uint32_t& fault) {
  const int nl = cfg.lag_timesteps + 1;
  const int h2 = (head + 1) % nl;
  float in[3][6], tq[3], tgt[3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    float a = AT(B.actions, j, e) * cfg.action_scale;
    if (jj == 0) a *= cfg.hip_scale_reduction;
    float target;
    if (cfg.use_lag) {
      B.lag_buffer[((size_t)head * 12 + j) * N + e] = a;
      target = B.lag_buffer[((size_t)h2 * 12 + j) * N + e] + cfg.default_dof_pos[j];
    } else {
      target = a + cfg.default_dof_pos[j];
    }
    AT(B.joint_pos_target, j, e) = target;
    tgt[jj] = target;
  }
  if (cfg.control_type == 1) {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      float err = L.q[jj] - tgt[jj] + AT(B.motor_offsets, j, e);
      float elast = AT(B.joint_pos_err_last, j, e), ell = AT(B.joint_pos_err_last_last, j, e);
      float vl = AT(B.joint_vel_last, j, e), vll = AT(B.joint_vel_last_last, j, e);
      in[jj][0] = err; in[jj][1] = elast; in[jj][2] = ell; in[jj][3] = L.qd[jj]; in[jj][4] = vl; in[jj][5] = vll;
      AT(B.joint_pos_err_last_last, j, e) = elast;
      AT(B.joint_pos_err_last, j, e) = err;
      AT(B.joint_vel_last_last, j, e) = vl;
      AT(B.joint_vel_last, j, e) = L.qd[jj];
    }
    if (full_wave) actuator_net_mfma(act_lds, act_io, (int)threadIdx.x & 63, 0, nw, true, in, tq);      // wave-uniform choice
    else actuator_net3(in, tq);
  } else {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      tq[jj] = cfg.kp * AT(B.Kp_factors, j, e) * (tgt[jj] - L.q[jj] + AT(B.motor_offsets, j, e)) - cfg.kd * AT(B.Kd_factors, j, e) * L.qd[jj];
    }
  }
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    float t = tq[jj] * AT(B.motor_strengths, j, e);
    if (!(fabsf(t) <= 3.0e38f)) { fault |= 1u << GO1_FAULT_TORQUE; t = 0.f; }
    const float lim = cfg.torque_limits[j];
    t = fminf(fmaxf(t, -lim), lim);
    L.tau[jj] = t;
    AT(B.torques, j, e) = t;
  }
}

// ---- the torque model OFF the critical path (step kernel, workgroups of several wavefronts) ----------------------------
// The actuator network only needs q, qd and the history; the first third of a substep (kinematics, contact candidates,
// ABA pass 1) does not need the torques.  So the master wavefront publishes the 192 input rows, the HELPER wavefronts
// evaluate all 12 tiles while the master runs the kinematics, and the master picks the torques up right before ABA pass 2.
// What compute_torques() keeps in global memory between substeps (actuator history, lagged targets) lives in a per-lane LDS
// stash for the duration of the step: loaded by the prologue in one batch (torque_stash_issue / _commit), written back once
// (torque_stash_store).  Same arithmetic in the same order as compute_torques(): results are bit-identical.
#define ACT_MAX_DEC 4
enum { AH_E1 = 0, AH_E2 = 3, AH_V1 = 6, AH_V2 = 9, AH_MS = 12, AH_MO = 15, AH_TGT = 18, AH_END = AH_TGT + 3 * ACT_MAX_DEC };
#define ACTH(k) acth[(k) * WAVE + lane]

// Split in two so that the prologue has ONE batch of global loads: torque_stash_issue() only issues loads (nothing depends on
// the actions), torque_stash_commit() — after the batch has landed — forms the targets, fills the stash and writes the lag
// buffer.  All loads of the lag buffer come before its stores (a substep reads the slot the NEXT substep overwrites).
struct StashIn { float lagv[ACT_MAX_DEC][3], e1[3], e2[3], v1[3], v2[3], ms[3], mo[3], dpos[3]; };
DEV void torque_stash_issue(CfgRef cfg, BufRef B, int leg, int e, int N, int head, StashIn& in) {
  const int nl = cfg.lag_timesteps + 1;
#pragma unroll
  for (int sb = 0; sb < ACT_MAX_DEC; sb++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      in.lagv[sb][jj] = 0.f;
      if (cfg.use_lag && sb < cfg.decimation && sb + 1 < nl) in.lagv[sb][jj] = B.lag_buffer[((size_t)((head + sb + 1) % nl) * 12 + j) * N + e];
    }
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    in.e1[jj] = AT(B.joint_pos_err_last, j, e);
    in.e2[jj] = AT(B.joint_pos_err_last_last, j, e);
    in.v1[jj] = AT(B.joint_vel_last, j, e);
    in.v2[jj] = AT(B.joint_vel_last_last, j, e);
    in.ms[jj] = AT(B.motor_strengths, j, e);
    in.mo[jj] = AT(B.motor_offsets, j, e);
    in.dpos[jj] = cfg.default_dof_pos[j];        // (indexed by the lane's leg: a vector load from the constant block)
  }
}
// act[jj]: the clipped action
DEV void torque_stash_commit(CfgRef cfg, BufRef B, float* acth, int lane, int leg, int e, int N, int head, const float act[3], const StashIn& in) {
  const int nl = cfg.lag_timesteps + 1;
  float a[3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    a[jj] = act[jj] * cfg.action_scale;
    if (jj == 0) a[jj] *= cfg.hip_scale_reduction;
  }
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    ACTH(AH_E1 + jj) = in.e1[jj];
    ACTH(AH_E2 + jj) = in.e2[jj];
    ACTH(AH_V1 + jj) = in.v1[jj];
    ACTH(AH_V2 + jj) = in.v2[jj];
    ACTH(AH_MS + jj) = in.ms[jj];
    ACTH(AH_MO + jj) = in.mo[jj];
  }
#pragma unroll
  for (int sb = 0; sb < ACT_MAX_DEC; sb++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      const float t = (cfg.use_lag && sb < cfg.decimation && sb + 1 < nl) ? in.lagv[sb][jj] : a[jj];
      ACTH(AH_TGT + 3 * sb + jj) = t + in.dpos[jj];
      if (cfg.use_lag && sb < cfg.decimation) B.lag_buffer[((size_t)((head + sb) % nl) * 12 + j) * N + e] = a[jj];
    }
}
// Input rows of substep `sub`.  The MASTER only posts q and qd of its three joints into the row slots they belong to (6 LDS stores);
// the row itself — position error against the lagged target, the two-deep histories — is built by the HELPER lane that owns the row
// (192 rows = 3 helper wavefronts x 64 lanes: one row per lane), which also advances the histories in the stash.  Until round 5 the
// master did all of it (15 stash reads, 12 stash writes, 6 row stores and their waits on the critical path, four times a step).
// Same arithmetic in the same order as compute_torques(): results are bit-identical.
DEV void torque_post_state(const Leg& L, float* io, int lane) {
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    float* row = io + A_IN + (3 * lane + jj) * 8;
    row[0] = L.q[jj];
    row[3] = L.qd[jj];
  }
}
// helper lane `hl` in [0, 192): row hl = joint jj = hl % 3 of master lane ml = hl / 3
DEV void torque_build_row(float* acth_, float* io, int hl, int sub) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const int ml = hl / 3, jj = hl - 3 * ml;
  float* acth = acth_;
  const int lane = ml;                                   // (ACTH indexes the master lane's stash column)
  float* row = io + A_IN + hl * 8;
  const float q = row[0], qd = row[3];
  const float err = q - ACTH(AH_TGT + 3 * sub + jj) + ACTH(AH_MO + jj);
  const float e1 = ACTH(AH_E1 + jj), e2 = ACTH(AH_E2 + jj), v1 = ACTH(AH_V1 + jj), v2 = ACTH(AH_V2 + jj);
  f4* p = reinterpret_cast<f4*>(row);
  p[0] = (f4){err, e1, e2, qd};
  p[1] = (f4){v1, v2, 1.f, 0.f};
  ACTH(AH_E2 + jj) = e1; ACTH(AH_E1 + jj) = err; ACTH(AH_V2 + jj) = v1; ACTH(AH_V1 + jj) = qd;
}
DEV void torque_collect(CfgRef cfg, Leg& L, const float* acth, const float* io, int lane, int leg, uint32_t& fault) {
  float tq[3];
  actuator_collect(io, lane, tq);
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    float t = tq[jj] * ACTH(AH_MS + jj);
    if (!(fabsf(t) <= 3.0e38f)) { fault |= 1u << GO1_FAULT_TORQUE; t = 0.f; }
    const float lim = cfg.torque_limits[3 * leg + jj];
    L.tau[jj] = fminf(fmaxf(t, -lim), lim);
  }
}
DEV void torque_stash_store(CfgRef cfg, BufRef B, const Leg& L, const float* acth, int lane, int leg, int e, int N) {
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    AT(B.joint_pos_err_last, j, e) = ACTH(AH_E1 + jj);
    AT(B.joint_pos_err_last_last, j, e) = ACTH(AH_E2 + jj);
    AT(B.joint_vel_last, j, e) = ACTH(AH_V1 + jj);
    AT(B.joint_vel_last_last, j, e) = ACTH(AH_V2 + jj);
    AT(B.joint_pos_target, j, e) = ACTH(AH_TGT + 3 * (cfg.decimation - 1) + jj);
    AT(B.torques, j, e) = L.tau[jj];
  }
}

// ================================================================================================
// physics substep
// ================================================================================================
struct Base {             // replicated in the 4 lanes of the environment
  V3 pos;                 // world position of the base origin
  float qx, qy, qz, qw;
  V3 w, v;                // angular velocity, velocity of the base origin (world axes)
  float mass0;            // trunk mass + payload
  V3 com0;                // base com in body axes (= com_displacement, reference legged_robot.py:671)
  float mu, rest;
};

DEV V3 model_v3(const float (*tab)[3], int i) { return v3(tab[i][0], tab[i][1], tab[i][2]); }

struct Cand { float phi, x, y, z, un, nx, ny, nz; uint32_t tag; };     // deepest contact candidate of one group of points; tag (contact
                                                                         // signature only): 16 x height-field cell + index of the winning point
DEV void cand_init(Cand& c) { c.phi = 1e30f; c.x = c.y = c.z = c.un = 0.f; c.nx = c.ny = 0.f; c.nz = 1.f; c.tag = 0u; }

// Terrain at world (x, y): height and unit normal of the TOP surface — plane, or bilinear interpolation of the int16 height
// field (same sample convention as _get_heights, reference legged_robot.py:1793-1806) — and, with WALLS, the vertical face
// next to the point: unit horizontal normal (towards the low side), horizontal distance, height of its upper edge
// (oracle terrain_sample(): the `trimesh` terrain's slope_treshold restated per cell of the height field).
struct Wall { bool on; V3 n; float d, top; int cell; };
// The sample in two halves, so that a caller can have the loads of several points in flight before it evaluates the first (one point at a
// time, a lane's 21 candidate points per substep were 21 dependent round trips to L2: profiles/r05_step_kernel_phases_rough.txt):
// terrain_fetch() = cell, in-cell coordinates and the four int16 loads; terrain_eval() = everything computed from them.
// GO1_NO_CONTRACT (first statement of a function body): no fused multiply-adds formed ACROSS the source's operations in that function.
// hipcc's default (-ffp-contract=fast) lets the backend fuse a product into a sum wherever the product has no other user — a property of the
// surrounding code after inlining, so two template instances of the same source can round differently.  Round 5 met exactly that: once the
// look-ups below ran ahead of their evaluation, go1_step_kernel_hf and its `_sig` twin (which keeps Cand::tag and the cell index alive)
// disagreed in the last bit of the bilinear interpolation, and 1 % of the environment-steps of the product-instance parity test were no longer
// bit-identical between the two (tests/twin_probe.py: deterministic, gone with -ffp-contract=on / off, gone with the pragma in these three
// functions alone; the plane instance's results are untouched by it — same digest).  The terrain sample is 30 operations per point: what
// the fusing saved is not measurable, the twin relation is what rule (a) of the parity tests stands on.
#if defined(__clang__)
#define GO1_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define GO1_NO_CONTRACT
#endif
struct TerrFetch { float ax, ay; int cell, p00, p01, p10, p11; bool flat; };
DEV void terrain_fetch(CfgRef cfg, const int16_t* __restrict__ hs, float x, float y, TerrFetch& f) {
  GO1_NO_CONTRACT
  f.flat = cfg.terrain_type == 0 || hs == nullptr;
  f.ax = f.ay = 0.f; f.cell = 0; f.p00 = f.p01 = f.p10 = f.p11 = 0;
  if (f.flat) return;
  float fx = (x + cfg.hf_border) / cfg.hf_hscale, fy = (y + cfg.hf_border) / cfg.hf_hscale;
  fx = fminf(fmaxf(fx, 0.f), (float)cfg.hf_rows - 1.000001f);
  fy = fminf(fmaxf(fy, 0.f), (float)cfg.hf_cols - 1.000001f);
  const int ix = (int)fx, iy = (int)fy;
  f.ax = fx - ix; f.ay = fy - iy;
  f.cell = ix * cfg.hf_cols + iy;
  const int16_t* p = hs + (size_t)ix * cfg.hf_cols + iy;
  f.p00 = p[0]; f.p01 = p[1]; f.p10 = p[cfg.hf_cols]; f.p11 = p[cfg.hf_cols + 1];
}
template <bool WALLS>
DEV void terrain_eval(CfgRef cfg, const TerrFetch& f, float& h, V3& n, Wall& wall) {
  GO1_NO_CONTRACT
  wall.on = false; wall.n = v3(1.f, 0.f, 0.f); wall.d = 0.f; wall.top = 0.f; wall.cell = 0;
  if (f.flat) { h = 0.f; n = v3(0.f, 0.f, 1.f); return; }
  const float ax = f.ax, ay = f.ay;
  const int p00 = f.p00, p01 = f.p01, p10 = f.p10, p11 = f.p11;
  wall.cell = f.cell;
  float h00 = p00 * cfg.hf_vscale, h01 = p01 * cfg.hf_vscale, h10 = p10 * cfg.hf_vscale, h11 = p11 * cfg.hf_vscale;
  if (WALLS) {
    const int T = cfg.hf_wall_units;
    const int dx0 = p10 - p00, dx1 = p11 - p01, dy0 = p01 - p00, dy1 = p11 - p10;          // steepness is decided on the integer samples
    const bool sx0 = abs(dx0) > T, sx1 = abs(dx1) > T, sy0 = abs(dy0) > T, sy1 = abs(dy1) > T;
    if (sx0 || sx1 || sy0 || sy1) {
      float best = 1e30f;
      if (sx0 && sx1 && (dx0 > 0) == (dx1 > 0)) {
        const bool up = dx0 > 0;
        wall.on = true; wall.n = v3(up ? -1.f : 1.f, 0.f, 0.f);
        wall.d = (up ? (1.f - ax) : ax) * cfg.hf_hscale;
        wall.top = up ? h10 * (1.f - ay) + h11 * ay : h00 * (1.f - ay) + h01 * ay;
        best = wall.d;
      }
      if (sy0 && sy1 && (dy0 > 0) == (dy1 > 0)) {
        const bool up = dy0 > 0;
        const float d = (up ? (1.f - ay) : ay) * cfg.hf_hscale;
        if (d < best) {
          wall.on = true; wall.n = v3(0.f, up ? -1.f : 1.f, 0.f);
          wall.d = d;
          wall.top = up ? h01 * (1.f - ax) + h11 * ax : h00 * (1.f - ax) + h10 * ax;
        }
      }
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        if (sx0) { const float lo = fminf(h00, h10); h00 = h10 = lo; }
        if (sx1) { const float lo = fminf(h01, h11); h01 = h11 = lo; }
        if (sy0) { const float lo = fminf(h00, h01); h00 = h01 = lo; }
        if (sy1) { const float lo = fminf(h10, h11); h10 = h11 = lo; }
      }
    }
  }
  h = h00 * (1.f - ax) * (1.f - ay) + h10 * ax * (1.f - ay) + h01 * (1.f - ax) * ay + h11 * ax * ay;
  const float dhdx = ((h10 - h00) * (1.f - ay) + (h11 - h01) * ay) / cfg.hf_hscale;
  const float dhdy = ((h01 - h00) * (1.f - ax) + (h11 - h10) * ax) / cfg.hf_hscale;
  const float inv = rsqrtf(dhdx * dhdx + dhdy * dhdy + 1.f);
  n = v3(-dhdx * inv, -dhdy * inv, inv);
}
template <bool WALLS, bool PLANE = false>
DEV void terrain_sample(CfgRef cfg, const int16_t* __restrict__ hs, float x, float y, float& h, V3& n, Wall& wall) {
  if (PLANE) { wall.on = false; wall.n = v3(1.f, 0.f, 0.f); wall.d = 0.f; wall.top = 0.f; wall.cell = 0; h = 0.f; n = v3(0.f, 0.f, 1.f); return; }
  TerrFetch f;
  terrain_fetch(cfg, hs, x, y, f);
  terrain_eval<WALLS>(cfg, f, h, n, wall);
}

// x: candidate point relative to the base origin (world axes); bpos: world position of the base origin.  c: deepest
// top-surface candidate of the point's group, cw: closest wall candidate of its shape
// WANTW: also keep the wall candidate (hip capsules do not: their points are too high up to meet a riser)
template <bool WALLS, bool WANTW>
DEV void cand_take(Cand& c, Cand& cw, float h, V3 n, const Wall& wl, V3 x, V3 bpos, float radius, SV vb, int m) {
  const float phi = (bpos.z + x.z) - radius - h;
  if (phi < c.phi) {
    const V3 xs = x - radius * n;               // contact point on the shape surface
    const V3 vp = vb.l + cross(vb.a, xs);
    c.phi = phi; c.x = xs.x; c.y = xs.y; c.z = xs.z; c.un = dot(n, vp);
    c.nx = n.x; c.ny = n.y; c.nz = n.z;
    c.tag = 16u * (uint32_t)wl.cell + (uint32_t)m;
  }
  if (WALLS && WANTW) {
    const float phiw = wl.d - radius;
    if (wl.on && bpos.z + x.z < wl.top && phiw < cw.phi) {
      const V3 xs = x - radius * wl.n;
      const V3 vp = vb.l + cross(vb.a, xs);
      cw.phi = phiw; cw.x = xs.x; cw.y = xs.y; cw.z = xs.z; cw.un = dot(wl.n, vp);
      cw.nx = wl.n.x; cw.ny = wl.n.y; cw.nz = wl.n.z;
      cw.tag = 16u * (uint32_t)wl.cell + (uint32_t)m;
    }
  }
}
template <bool WALLS, bool WANTW = WALLS, bool PLANE = false>
DEV void cand_try(CfgRef cfg, const int16_t* __restrict__ hs, Cand& c, Cand& cw, V3 x, V3 bpos, float radius, SV vb, int m) {
  float h;
  V3 n;
  Wall wl;
  terrain_sample<WALLS, PLANE>(cfg, hs, bpos.x + x.x, bpos.y + x.y, h, n, wl);
  cand_take<WALLS, WANTW>(c, cw, h, n, wl, x, bpos, radius, vb, m);
}
// the same from a sample fetched earlier (terrain_fetch at bpos + x)
// cd: the contact distance.  A point further than cd above the HIGHEST of its cell's four samples cannot become a contact — the bilinear
// surface (and, with walls, the lowered one and the risers' upper edges) lies at or below that sample — and is dropped before the
// evaluation: a group's candidate is only ever read where its depth is below cd (contact list below), so the list does not change.
// (The margin covers the rounding of the interpolation weights' sum.)  A walking robot's trunk, hips and thighs take this exit.
template <bool WALLS, bool WANTW = WALLS>
DEV void cand_eval(CfgRef cfg, const TerrFetch& f, Cand& c, Cand& cw, V3 x, V3 bpos, float radius, SV vb, int m, float cd) {
  GO1_NO_CONTRACT
  {
    const float vs = cfg.hf_vscale;
    const float top = fmaxf(fmaxf(f.p00 * vs, f.p01 * vs), fmaxf(f.p10 * vs, f.p11 * vs));
    if ((bpos.z + x.z) - radius - (top + 1e-4f + 1e-5f * fabsf(top)) >= cd) return;
  }
  float h;
  V3 n;
  Wall wl;
  terrain_eval<WALLS>(cfg, f, h, n, wl);
  cand_take<WALLS, WANTW>(c, cw, h, n, wl, x, bpos, radius, vb, m);
}
DEV void cand_min_dpp(Cand& c, int lane) {      // quad-wide deepest candidate (ties: lower leg index, as the serial scan)
#pragma unroll
  for (int step = 0; step < 2; step++) {
    Cand o;
    if (step == 0) { o.phi = dpp_xor1(c.phi); o.x = dpp_xor1(c.x); o.y = dpp_xor1(c.y); o.z = dpp_xor1(c.z); o.un = dpp_xor1(c.un);
                     o.nx = dpp_xor1(c.nx); o.ny = dpp_xor1(c.ny); o.nz = dpp_xor1(c.nz);
                     o.tag = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.tag, 0xB1, 0xF, 0xF, false); }
    else           { o.phi = dpp_xor2(c.phi); o.x = dpp_xor2(c.x); o.y = dpp_xor2(c.y); o.z = dpp_xor2(c.z); o.un = dpp_xor2(c.un);
                     o.nx = dpp_xor2(c.nx); o.ny = dpp_xor2(c.ny); o.nz = dpp_xor2(c.nz);
                     o.tag = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.tag, 0x4E, 0xF, 0xF, false); }
    const int bit = step == 0 ? 1 : 2;
    const bool other_is_lower = ((lane & bit) != 0);
    bool take = (o.phi < c.phi) || (o.phi == c.phi && other_is_lower);
    if (take) c = o;
  }
}
// contact frame: normal n, t1 = x-axis projected on the tangent plane, t2 = n x t1 (oracle contact_frame()).  A normal along
// the x axis (a wall facing x, a body-body normal) leaves no projection: the y axis takes its place.  Only a non-finite
// normal is a fault.
DEV void contact_frame(V3 n, V3& t1, V3& t2, uint32_t& fault) {
  V3 t = v3(1.f - n.x * n.x, -n.x * n.y, -n.x * n.z);
  float tt = dot(t, t);
  if (!(tt > 1e-12f)) {
    if (!(tt <= 1e-12f)) fault |= 1u << GO1_FAULT_CONTACT_FRAME;       // NaN normal
    t = v3(-n.y * n.x, 1.f - n.y * n.y, -n.y * n.z);
    tt = dot(t, t);
    if (!(tt > 1e-12f)) { t = v3(0.f, 1.f, 0.f); tt = 1.f; }
  }
  t1 = rsqrtf(tt) * t;
  t2 = cross(n, t1);
}

// closest points of the segments p1-q1 and p2-q2 (Ericson 5.1.9; oracle seg_seg())
DEV void seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
  const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), cc = dot(d1, r), b = dot(d1, d2);
  const float den = a * e - b * b;
  float sN = 0.f;
  if (den > 1e-12f) sN = fminf(fmaxf((b * f - cc * e) / den, 0.f), 1.f);
  float tN = (b * sN + f) / e;
  if (tN < 0.f) { tN = 0.f; sN = fminf(fmaxf(-cc / a, 0.f), 1.f); }
  else if (tN > 1.f) { tN = 1.f; sN = fminf(fmaxf((b - cc) / a, 0.f), 1.f); }
  c1 = p1 + sN * d1;
  c2 = p2 + tN * d2;
}
// contact of capsule A (p1-q1, ra) with capsule B (p2-q2, rb): normal from B to A, point midway between the surfaces
DEV bool capsule_contact(V3 p1, V3 q1, float ra, V3 p2, V3 q2, float rb, float cd, Cand& o) {
  V3 c1, c2;
  seg_seg(p1, q1, p2, q2, c1, c2);
  const V3 d = c1 - c2;
  const float d2 = dot(d, d);
  if (!(d2 > 1e-12f)) return false;
  const float dist = sqrtf(d2), phi = dist - ra - rb;
  if (!(phi < cd)) return false;
  const V3 n = (1.f / dist) * d, x = c2 + (rb + 0.5f * phi) * n;
  o.phi = phi; o.x = x.x; o.y = x.y; o.z = x.z; o.nx = n.x; o.ny = n.y; o.nz = n.z; o.un = 0.f;
  return true;
}

// ---- rows ---------------------------------------------------------------------------------------------------------------
// the ABA factors of one leg as the emission reads them (from registers on the lane that owns the leg, from the hand-over
// packet on any other lane)
struct LegFac { SV S[3], U[3]; float Dinv[3], sD[3], qdf[3]; };
DEV void legfac_store(lf4* pkl, int lane, const LegFac& F) {
  float f[48];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    f[6 * j] = F.S[j].a.x; f[6 * j + 1] = F.S[j].a.y; f[6 * j + 2] = F.S[j].a.z; f[6 * j + 3] = F.S[j].l.x; f[6 * j + 4] = F.S[j].l.y; f[6 * j + 5] = F.S[j].l.z;
    f[18 + 6 * j] = F.U[j].a.x; f[19 + 6 * j] = F.U[j].a.y; f[20 + 6 * j] = F.U[j].a.z; f[21 + 6 * j] = F.U[j].l.x; f[22 + 6 * j] = F.U[j].l.y; f[23 + 6 * j] = F.U[j].l.z;
    f[36 + j] = F.Dinv[j]; f[39 + j] = F.sD[j]; f[42 + j] = F.qdf[j];
  }
  f[45] = f[46] = f[47] = 0.f;
#pragma unroll
  for (int q = 0; q < PKL_Q; q++) pkl[q * WAVE + lane] = (lf4){f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
}
DEV void legfac_load(const lf4* pkl, int lane_src, LegFac& F) {
  float f[48];
#pragma unroll
  for (int q = 0; q < PKL_Q; q++) { const lf4 v = pkl[q * WAVE + lane_src]; f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3]; }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    F.S[j] = sv(v3(f[6 * j], f[6 * j + 1], f[6 * j + 2]), v3(f[6 * j + 3], f[6 * j + 4], f[6 * j + 5]));
    F.U[j] = sv(v3(f[18 + 6 * j], f[19 + 6 * j], f[20 + 6 * j]), v3(f[21 + 6 * j], f[22 + 6 * j], f[23 + 6 * j]));
    F.Dinv[j] = f[36 + j]; F.sD[j] = f[39 + j]; F.qdf[j] = f[42 + j];
  }
}
// the base's side of the hand-over: free twist, restitution of the pair, warm-start switch, rows of L^-1
struct EnvPk { V3 w_free, v_free; float e_c, warm; float Li[21]; };
DEV void envpk_store(lf4* pke, int el, const EnvPk& E) {
  float f[36];
  f[0] = E.w_free.x; f[1] = E.w_free.y; f[2] = E.w_free.z; f[3] = E.v_free.x; f[4] = E.v_free.y; f[5] = E.v_free.z; f[6] = E.e_c; f[7] = E.warm;
#pragma unroll
  for (int i = 0; i < 21; i++) f[8 + i] = E.Li[i];
#pragma unroll
  for (int i = 29; i < 36; i++) f[i] = 0.f;
#pragma unroll
  for (int q = 0; q < PKE_Q; q++) pke[q * EPW + el] = (lf4){f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
}
DEV void envpk_load(const lf4* pke, int el, EnvPk& E) {
  float f[36];
#pragma unroll
  for (int q = 0; q < PKE_Q; q++) { const lf4 v = pke[q * EPW + el]; f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3]; }
  E.w_free = v3(f[0], f[1], f[2]); E.v_free = v3(f[3], f[4], f[5]); E.e_c = f[6]; E.warm = f[7];
#pragma unroll
  for (int i = 0; i < 21; i++) E.Li[i] = f[8 + i];
}
// One solver row in the factorised coordinates (header of this file): a = [L^-1 g ; u_j / sqrt(D_j)]
struct Row { float g[6], u[3]; };
DEV float row_dot(const Row& a, const Row& b) {
  float s = a.g[0] * b.g[0];
#pragma unroll
  for (int i = 1; i < 6; i++) s = fmaf(a.g[i], b.g[i], s);
#pragma unroll
  for (int i = 0; i < 3; i++) s = fmaf(a.u[i], b.u[i], s);
  return s;
}
// a unit impulse (spatial force pA0 on the body at `depth` of the leg; depth < 0: on the base itself) propagated to the base
// through the leg's ABA factors: the wrench g arriving at the base and the joint residuals u_j (not yet scaled)
DEV void row_propagate(const LegFac& F, int depth, SV pA, SV& g, float uj[3]) {
  uj[0] = uj[1] = uj[2] = 0.f;
#pragma unroll
  for (int j = 2; j >= 0; j--) {
    if (j <= depth) {
      const float u = -dot(F.S[j], pA);
      uj[j] = u;
      pA = pA + (u * F.Dinv[j]) * F.U[j];
    }
  }
  g = pA;
}
DEV void row_finish(const float Li[21], SV g, const float uj[3], const float sD[3], Row& r) {
  const float in[6] = {g.a.x, g.a.y, g.a.z, g.l.x, g.l.y, g.l.z};
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = Li[i * (i + 1) / 2] * in[0];
#pragma unroll
    for (int j = 1; j <= i; j++) s = fmaf(Li[i * (i + 1) / 2 + j], in[j], s);
    r.g[i] = s;
  }
#pragma unroll
  for (int j = 0; j < 3; j++) r.u[j] = uj[j] * sD[j];
}
// flags word of a contact record: leg of body A (4: the base), self-contact bit, 1 + index of the B-side record (0: none)
DEV float cr_flags(int legA, bool self, int sb1) { return (float)(legA | (self ? 8 : 0) | (sb1 << 4)); }
// a finished contact into its record.  c_n = b_n - v*
DEV void contact_record_store(lf4* crl, int el, int k, const Row& an, const Row& a1, const Row& a2, float flags, float cn, float b1, float b2,
                              float wnn, float w11, float w22, float w1n, float w2n, V3 lam, V3 x, V3 n, bool bounce, uint32_t& fault) {
  if (!(wnn > 1e-9f) || !(w11 > 1e-9f) || !(w22 > 1e-9f)) fault |= 1u << GO1_FAULT_W_DIAG;
  CRQ(k, 0) = (lf4){an.g[0], an.g[1], an.g[2], an.g[3]};
  CRQ(k, 1) = (lf4){an.g[4], an.g[5], an.u[0], an.u[1]};
  CRQ(k, 2) = (lf4){an.u[2], cn, 1.f / wnn, w1n};
  CRQ(k, 3) = (lf4){a1.g[0], a1.g[1], a1.g[2], a1.g[3]};
  CRQ(k, 4) = (lf4){a1.g[4], a1.g[5], a1.u[0], a1.u[1]};
  CRQ(k, 5) = (lf4){a1.u[2], b1, 1.f / w11, w2n};
  CRQ(k, 6) = (lf4){a2.g[0], a2.g[1], a2.g[2], a2.g[3]};
  CRQ(k, 7) = (lf4){a2.g[4], a2.g[5], a2.u[0], a2.u[1]};
  CRQ(k, 8) = (lf4){a2.u[2], b2, 1.f / w22, flags};
  CRQ(k, 9) = (lf4){lam.x, lam.y, lam.z, bounce ? 1.f : 0.f};       // ([3]: the restitution branch was taken — contact signature)
  CRQ(k, 10) = (lf4){x.x, x.y, x.z, 0.f};
  CRQ(k, 11) = (lf4){n.x, n.y, n.z, 0.f};
}
// the ten slots of a contact the sweep reads, and the packed-f32 arithmetic on them: v_pk_fma_f32 does two lanes' worth of FMAs
// per issue slot, and the issue slot is what the sweep is bound by
typedef __attribute__((ext_vector_type(2))) float f2;
struct SweepRec { lf4 q[10]; };
DEV void sweep_rec_load(const lf4* crl, int el, int k, SweepRec& r) {
#pragma unroll
  for (int q = 0; q < 10; q++) r.q[q] = CRQ(k, q);
}
DEV f2 lo2(lf4 v) { return (f2){v[0], v[1]}; }
DEV f2 hi2(lf4 v) { return (f2){v[2], v[3]}; }
DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV f2 splat2(float x) { return (f2){x, x}; }
struct SweepState { f2 z01, z23, z45, y01; float y2; };
// a . s of row r (slots 3r .. 3r + 2): base part, and the leg part (to be masked by "the row is on my leg")
DEV void sweep_row_dot(const SweepRec& R, int r, const SweepState& S, float& base, float& legpart) {
  f2 acc = lo2(R.q[3 * r]) * S.z01;
  acc = fma2(hi2(R.q[3 * r]), S.z23, acc);
  acc = fma2(lo2(R.q[3 * r + 1]), S.z45, acc);
  base = acc[0] + acc[1];
  const f2 pu = hi2(R.q[3 * r + 1]) * S.y01;
  legpart = fmaf(R.q[3 * r + 2][0], S.y2, pu[0] + pu[1]);
}
// s += d_r a_r for the three rows; m: 1 on the lane of the contact's leg, 0 elsewhere
DEV void sweep_add_rows(const SweepRec& R, SweepState& S, float m, float d0, float d1, float d2) {
  const f2 s0 = splat2(d0), s1 = splat2(d1), s2 = splat2(d2);
  S.z01 = fma2(lo2(R.q[0]), s0, fma2(lo2(R.q[3]), s1, fma2(lo2(R.q[6]), s2, S.z01)));
  S.z23 = fma2(hi2(R.q[0]), s0, fma2(hi2(R.q[3]), s1, fma2(hi2(R.q[6]), s2, S.z23)));
  S.z45 = fma2(lo2(R.q[1]), s0, fma2(lo2(R.q[4]), s1, fma2(lo2(R.q[7]), s2, S.z45)));
  const float m0 = m * d0, m1 = m * d1, m2 = m * d2;
  S.y01 = fma2(hi2(R.q[1]), splat2(m0), fma2(hi2(R.q[4]), splat2(m1), fma2(hi2(R.q[7]), splat2(m2), S.y01)));
  S.y2 = fmaf(R.q[2][0], m0, fmaf(R.q[5][0], m1, fmaf(R.q[8][0], m2, S.y2)));
}

// Finish the listed TERRAIN contact k of environment el from the item the master posted in its record (any lane may do this:
// everything comes from LDS): frame, target velocity, b = J v_free, start impulse, the three rows.
DEV void emit_terrain_contact(CfgRef cfg, const SolverLds& Z, int el, int k, float h, uint32_t& fault) {
  float* const lds = Z.lds;
  lf4* const crl = Z.cr();
  const lf4 i0 = CRQ(k, 9), i1 = CRQ(k, 10), i2 = CRQ(k, 11);
  const float phi = i0[0], un_pre = i1[0], share = i2[3];
  const V3 x = v3(i0[1], i0[2], i0[3]), n = v3(i1[1], i1[2], i1[3]);
  const int depth = (int)i2[0], legc = (int)i2[1], leg = legc & 3, body = (int)i2[2];
  const float fsplit = (float)(legc >> 2);            // mass splitting (the sweep's leg phase): n - 1 for a hip / thigh row of an environment with n > 1 such legs
  EnvPk E;
  envpk_load(Z.pke(), el, E);
  LegFac F;
  legfac_load(Z.pkl(), 4 * el + (leg & 3), F);
  V3 t1, t2;
  contact_frame(n, t1, t2, fault);
  float vs = fminf(-phi / h, cfg.max_depenetration_velocity);
  const bool bounce = un_pre < -cfg.bounce_threshold_velocity && -E.e_c * un_pre > vs;
  if (bounce) vs = -E.e_c * un_pre;
  SV vb = sv(E.w_free, E.v_free);
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (j <= depth) vb = vb + F.qdf[j] * F.S[j];
  const V3 vp = vb.l + cross(vb.a, x);
  const V3 wl = v3(LDS(L_LAM + 3 * body), LDS(L_LAM + 3 * body + 1), LDS(L_LAM + 3 * body + 2));
  const float sh = E.warm != 0.f ? share : 0.f;
  const V3 lam = v3(sh * dot(wl, n), sh * dot(wl, t1), sh * dot(wl, t2));
  Row a[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const V3 d = r == 0 ? n : r == 1 ? t1 : t2;
    SV g;
    float uj[3];
    row_propagate(F, depth, -sv(cross(x, d), d), g, uj);
    row_finish(E.Li, g, uj, F.sD, a[r]);
  }
  // the record's W entries of a split row are those of the split system: W + (n - 1) a_z . a_z (base part once more per extra sub-body)
  auto zdot = [](const Row& p, const Row& q) {
    return fmaf(p.g[0], q.g[0], fmaf(p.g[1], q.g[1], fmaf(p.g[2], q.g[2], fmaf(p.g[3], q.g[3], fmaf(p.g[4], q.g[4], p.g[5] * q.g[5])))));
  };
  contact_record_store(crl, el, k, a[0], a[1], a[2], cr_flags(depth < 0 ? 4 : leg, false, 0), dot(n, vp) - vs, dot(t1, vp), dot(t2, vp),
                       fmaf(fsplit, zdot(a[0], a[0]), row_dot(a[0], a[0])), fmaf(fsplit, zdot(a[1], a[1]), row_dot(a[1], a[1])),
                       fmaf(fsplit, zdot(a[2], a[2]), row_dot(a[2], a[2])), fmaf(fsplit, zdot(a[1], a[0]), row_dot(a[1], a[0])),
                       fmaf(fsplit, zdot(a[2], a[0]), row_dot(a[2], a[0])), lam, x, n, bounce, fault);
}
// terrain contacts k = k0, k0 + kstride, ... of environment el (K listed, nF .. nF + nS - 1 are self-contacts: the master's)
DEV void emit_terrain_contacts(CfgRef cfg, const SolverLds& Z, int el, int k0, int kstride, float h) {
  float* const lds = Z.lds;
  const int K = (int)LDS(L_KL), nF = (int)LDS(L_KL + 4), nS = (int)LDS(L_KL + 5);
  uint32_t fl = 0;
#pragma unroll 1
  for (int k = k0; k < K; k += kstride)
    if (!(k >= nF && k < nF + nS)) emit_terrain_contact(cfg, Z, el, k, h, fl);
  if (fl & (1u << GO1_FAULT_CONTACT_FRAME)) LDS(L_KL + 3) = 1.f;
  if (fl & (1u << GO1_FAULT_W_DIAG)) LDS(L_KL + 6) = 1.f;
}

// ---- the substep ------------------------------------------------------------------------------------------------------
// Wave-uniform bounds of the sweep
struct SolveBounds { int Kw; unsigned LAw; };
DEV SolveBounds solve_bounds(int K, bool legact) {
  SolveBounds m;
  m.Kw = 0;
#pragma unroll 1
  for (int kk = 1; kk <= MAXC; kk++) { if (__ballot(K >= kk) == 0ull) break; m.Kw = kk; }
  unsigned long long bl = __ballot(legact);
  bl |= bl >> 32; bl |= bl >> 16; bl |= bl >> 8; bl |= bl >> 4;
  m.LAw = (unsigned)(bl & 0xFull);
  return m;
}
// pair index of the legs lo < hi in the order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
DEV int leg_pair_index(int lo, int hi) { return lo == 0 ? hi - 1 : lo == 1 ? hi + 1 : 5; }
// capsule combination `type` of a pair of legs: the segments of body A / B (0 lower leg, 1 thigh, 2 hip capsule), a segment's radius and the
// joint its body hangs on (= the depth of the impulse propagation: lower leg 2, thigh 1, hip 0)
DEV int self_seg_a(int type) { return type == 2 || type == 3 ? 1 : type == 4 ? 2 : 0; }
DEV int self_seg_b(int type) { return type == 1 || type == 3 ? 1 : type == 5 ? 2 : 0; }
DEV float self_seg_radius(int seg) { return seg == 0 ? GO1_SELF_LEG_RADIUS : seg == 1 ? GO1_SELF_THIGH_RADIUS : (float)GO1_HIP_CAPSULE_RADIUS; }

// acth != nullptr: the torques of this substep are being evaluated by the helper wavefronts (torque_post_state was called, the
// workgroup barrier behind it passed): they are picked up right before ABA pass 2.
// The helper wavefronts (nw > 1, always) run emit_terrain_contacts() between the two workgroup barriers of the emission hand-over.
// PLANE: the terrain is the plane z = 0 (terrain_type 0; never with WALLS): no height samples, and the deepest corner of a box end
// is known from the signs of the box axes' z components — one candidate per end instead of four
template <bool WALLS, bool SIG, bool PLANE>
DEV void physics_substep(CfgRef cfg, BufRef B, const SolverLds& Z, int lane, int nw, Base& s, Leg& L, V3 grav,
                         bool use_warm, float h, uint32_t& fault, uint32_t (&dropacc)[GO1_CC_COUNT], const float* acth, int e, int N, int sub PROF_PARAM) {
  float* const lds = Z.lds;
  lf4* const crl = Z.cr();
  lf4* const jrl = Z.jr();
  lf4* const sbl = Z.sb();
  const int16_t* __restrict__ hs = B.height_samples;
  const int leg = lane & 3, el = lane >> 2;
  const M3 R0 = quat_to_mat(s.qx, s.qy, s.qz, s.qw);
  const SV v0 = sv(s.w, s.v);
  const float cd = cfg.contact_distance;
  // (height field: the samples under the lane's two trunk corners are requested here and used after the base-body block)
  TerrFetch fb[2];
  V3 xb[2];
  if (!PLANE) {
#pragma unroll
    for (int mm = 0; mm < 2; mm++) {
      const int m = 2 * leg + mm;
      xb[mm] = mul(R0, v3((m & 1 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[0], (m & 2 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[1], (m & 4 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[2]));
      terrain_fetch(cfg, hs, s.pos.x + xb[mm].x, s.pos.y + xb[mm].y, fb[mm]);
    }
  }
  // ---- base body (replicated) ----------------------------------------------------------------------
  Sym6 IA0;
  SV pA0;
  {
    float Il[6], Iw[6];
    float scale = s.mass0 / GO1_BODY_MASS[0];          // recomputeInertia=True: mass-proportional (oracle kinematics())
#pragma unroll
    for (int i = 0; i < 6; i++) Il[i] = GO1_BODY_INERTIA[0][i] * scale;
    rotate_inertia(R0, Il, Iw);
    V3 c = mul(R0, s.com0);
    IA0 = rigid_inertia(s.mass0, c, Iw);
    SV hv = sym6_mul(IA0, v0);
    V3 fg = s.mass0 * grav;
    pA0 = cross_force(v0, hv) - sv(cross(c, fg), fg);
  }
  // Contact candidates (oracle detect_contacts()).  Trunk box: every corner is a candidate of its own (the lane holds corners
  // 2 leg and 2 leg + 1); its wall candidate is the quad-wide closest.
  Cand cb[2], cwb;
  cand_init(cwb);
#pragma unroll
  for (int mm = 0; mm < 2; mm++) {
    cand_init(cb[mm]);
    const int m = 2 * leg + mm;
    if (PLANE) {
      V3 l = v3((m & 1 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[0], (m & 2 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[1], (m & 4 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[2]);
      cand_try<WALLS, WALLS, PLANE>(cfg, hs, cb[mm], cwb, mul(R0, l), s.pos, 0.f, v0, m);
    } else cand_eval<WALLS, WALLS>(cfg, fb[mm], cb[mm], cwb, xb[mm], s.pos, 0.f, v0, m, cd);
  }
  if (WALLS) cand_min_dpp(cwb, lane);

  // ---- own leg: kinematics, contact candidates, ABA pass 1 -------------------------------------------
  LegFac F;
  float uu[3];
  SV cj[3];
  Cand ch[2], ct[2], ck[2], cf, cwt, cwk, cwf;      // hip, thigh, calf: one candidate per end; foot; wall candidates of thigh, calf, foot
  V3 pthigh, pknee, pfoot, phipa, phipb;      // own thigh / lower-leg segments and hip capsule for the self-collision test (rel. base origin)
  SV vthigh, vlow, vhip;                      //   and the three bodies' twists before the step
  SV pA[3];
  Sym6 IA[3];
  {
    M3 R[3];
    V3 p[3];
    SV v[3];
    M3 Rpar = R0;
    V3 ppar = v3(0.f, 0.f, 0.f);
    SV vpar = v0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j, b = ji + 1;
      p[j] = ppar + mul(Rpar, model_v3(GO1_JOINT_ORIGIN, ji));
      V3 ax = (j == 0) ? Rpar.c0 : Rpar.c1;
      float sn, cs;
      sincosf(L.q[j], &sn, &cs);
      R[j] = (j == 0) ? rot_x(Rpar, sn, cs) : rot_y(Rpar, sn, cs);
      F.S[j] = sv(ax, cross(p[j], ax));
      SV vj = L.qd[j] * F.S[j];
      v[j] = vpar + vj;
      cj[j] = cross_motion(v[j], vj);
      float Il[6], Iw[6];
#pragma unroll
      for (int i = 0; i < 6; i++) Il[i] = GO1_BODY_INERTIA[b][i];
      rotate_inertia(R[j], Il, Iw);
      V3 com = p[j] + mul(R[j], model_v3(GO1_BODY_COM, b));
      float m = GO1_BODY_MASS[b];
      IA[j] = rigid_inertia(m, com, Iw);
      SV hv = sym6_mul(IA[j], v[j]);
      V3 fg = m * grav;
      pA[j] = cross_force(v[j], hv) - sv(cross(com, fg), fg);
      Rpar = R[j]; ppar = p[j]; vpar = v[j];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) { cand_init(ch[i]); cand_init(ct[i]); cand_init(ck[i]); }
    cand_init(cf); cand_init(cwt); cand_init(cwk); cand_init(cwf);
    {
      Cand nowall;
      cand_init(nowall);
      V3 hc = model_v3(GO1_HIP_CAPSULE_CENTER, leg);
      if (PLANE) {
#pragma unroll
        for (int m = 0; m < 2; m++) {
          V3 l = v3(hc.x, hc.y + (m ? 1.f : -1.f) * (float)GO1_HIP_CAPSULE_HALF, hc.z);
          cand_try<WALLS, false, PLANE>(cfg, hs, ch[m], nowall, p[0] + mul(R[0], l), s.pos, (float)GO1_HIP_CAPSULE_RADIUS, v[0], m);      // (same top surface as every other shape)
        }
#ifndef GO1_ABLATE_CAND
#pragma unroll
        for (int en = 0; en < 2; en++) {       // thigh / calf boxes: long axis z -> ends by the sign of z (corner bit 2)
          // on the plane the deepest corner of an end minimises z = ... + sx hx R.c0.z + sy hy R.c1.z: sx = -sign(R.c0.z), sy likewise
          // (ties: the lower corner index, i.e. the negative sign — the order the scan over the corners resolves them in)
          const int mt = (R[1].c0.z < 0.f ? 1 : 0) | (R[1].c1.z < 0.f ? 2 : 0) | (4 * en);
          const int mk = (R[2].c0.z < 0.f ? 1 : 0) | (R[2].c1.z < 0.f ? 2 : 0) | (4 * en);
          const V3 lt = v3(GO1_THIGH_BOX_CENTER[0] + (mt & 1 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[0],
                           GO1_THIGH_BOX_CENTER[1] + (mt & 2 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[1],
                           GO1_THIGH_BOX_CENTER[2] + (mt & 4 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[2]);
          const V3 lk = v3(GO1_CALF_BOX_CENTER[0] + (mk & 1 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[0],
                           GO1_CALF_BOX_CENTER[1] + (mk & 2 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[1],
                           GO1_CALF_BOX_CENTER[2] + (mk & 4 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[2]);
          cand_try<false, false, true>(cfg, hs, ct[en], cwt, p[1] + mul(R[1], lt), s.pos, 0.f, v[1], mt);
          cand_try<false, false, true>(cfg, hs, ck[en], cwk, p[2] + mul(R[2], lk), s.pos, 0.f, v[2], mk);
        }
#endif
        cand_try<WALLS, WALLS, PLANE>(cfg, hs, cf, cwf, p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg)), s.pos, (float)GO1_FOOT_RADIUS, v[2], 0);
      } else {
        // Height field: 19 points per lane (2 hip capsule ends, 8 + 8 box corners, the foot), each a cell look-up.  The look-ups run one
        // AHEAD of the evaluations: the hip ends', the foot's and the first thigh corner's loads are requested together, and every turn of the
        // corner loops requests the next corner's sample before it evaluates its own (the candidates of a group are still tried in corner
        // order, which is what decides ties)
        auto thigh_corner = [&](int m) {
          return p[1] + mul(R[1], v3(GO1_THIGH_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[0],
                                     GO1_THIGH_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[1],
                                     GO1_THIGH_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[2]));
        };
        auto calf_corner = [&](int m) {
          return p[2] + mul(R[2], v3(GO1_CALF_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[0],
                                     GO1_CALF_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[1],
                                     GO1_CALF_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[2]));
        };
        TerrFetch fh[2], ff, cur, nxt;
        V3 xh[2], xc, xn;
#pragma unroll
        for (int m = 0; m < 2; m++) {
          xh[m] = p[0] + mul(R[0], v3(hc.x, hc.y + (m ? 1.f : -1.f) * (float)GO1_HIP_CAPSULE_HALF, hc.z));
          terrain_fetch(cfg, hs, s.pos.x + xh[m].x, s.pos.y + xh[m].y, fh[m]);
        }
        const V3 xf = p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg));
        terrain_fetch(cfg, hs, s.pos.x + xf.x, s.pos.y + xf.y, ff);
#ifndef GO1_ABLATE_CAND
        xc = thigh_corner(0);
        terrain_fetch(cfg, hs, s.pos.x + xc.x, s.pos.y + xc.y, cur);
#endif
#pragma unroll
        for (int m = 0; m < 2; m++) cand_eval<WALLS, false>(cfg, fh[m], ch[m], nowall, xh[m], s.pos, (float)GO1_HIP_CAPSULE_RADIUS, v[0], m, cd);
        cand_eval<WALLS, WALLS>(cfg, ff, cf, cwf, xf, s.pos, (float)GO1_FOOT_RADIUS, v[2], 0, cd);
#ifndef GO1_ABLATE_CAND
#pragma unroll
        for (int en = 0; en < 2; en++) {       // thigh / calf boxes: long axis z -> ends by the sign of z (corner bit 2)
#pragma unroll 1
          for (int m = 4 * en; m < 4 * en + 4; m++) {
            xn = (m == 4 * en + 3) ? calf_corner(4 * en) : thigh_corner(m + 1);
            terrain_fetch(cfg, hs, s.pos.x + xn.x, s.pos.y + xn.y, nxt);
            cand_eval<WALLS>(cfg, cur, ct[en], cwt, xc, s.pos, 0.f, v[1], m, cd);
            cur = nxt; xc = xn;
          }
#pragma unroll 1
          for (int m = 4 * en; m < 4 * en + 4; m++) {
            xn = (m == 4 * en + 3) ? thigh_corner((4 * en + 4) & 7) : calf_corner(m + 1);      // (after the last corner: a sample nobody uses)
            terrain_fetch(cfg, hs, s.pos.x + xn.x, s.pos.y + xn.y, nxt);
            cand_eval<WALLS>(cfg, cur, ck[en], cwk, xc, s.pos, 0.f, v[2], m, cd);
            cur = nxt; xc = xn;
          }
        }
#endif
      }
      pthigh = p[1]; pknee = p[2]; pfoot = p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg));
      vthigh = v[1]; vlow = v[2]; vhip = v[0];
      {
        const V3 hc = model_v3(GO1_HIP_CAPSULE_CENTER, leg);
        phipa = p[0] + mul(R[0], v3(hc.x, hc.y - (float)GO1_HIP_CAPSULE_HALF, hc.z));
        phipb = p[0] + mul(R[0], v3(hc.x, hc.y + (float)GO1_HIP_CAPSULE_HALF, hc.z));
      }
    }
  }

  PROF(25);
  // ---- self-collision, geometry (asset self_collisions = 0: enabled): capsules — lower leg (knee -> foot centre, radius of
  // the foot sphere), thigh (thigh joint -> knee), hip (the URDF cylinder as a capsule, the shape the terrain sees too) — of DIFFERENT
  // legs against each other and lower legs against the trunk's capsule.  Every lane publishes its segments and the bodies' twists, tests
  // the pairs it is part of in the canonical order (body A on the lower-numbered leg), and the environment's pair mask is the OR over the quad.
  // pid = 6 type + pair, type 0 lower-lower, 1 lower(A)-thigh(B), 2 thigh(A)-lower(B), 3 thigh-thigh, 4 hip(A)-lower(B), 5 lower(A)-hip(B)
  // (round 5: the one combination with a hip capsule the joint limits let touch); 36 + leg: lower leg - trunk.
  unsigned long long smask = 0;    // listed pairs of the environment
  int nS = 0;
  if (cfg.self_collision) {
    lf4* seg = Z.seg();
    lf4* twx = Z.tw();             // (slot 5 of the free-twist block: not used otherwise, dedicated memory — a failed environment's non-finite
                                   //  values must not reach slots other environments read, tests/test_emu_parity.py failed_state)
    seg[0 * WAVE + lane] = (lf4){pknee.x, pknee.y, pknee.z, pfoot.x};
    seg[1 * WAVE + lane] = (lf4){pfoot.y, pfoot.z, pthigh.x, pthigh.y};
    seg[2 * WAVE + lane] = (lf4){pthigh.z, vlow.a.x, vlow.a.y, vlow.a.z};
    seg[3 * WAVE + lane] = (lf4){vlow.l.x, vlow.l.y, vlow.l.z, vthigh.a.x};
    seg[4 * WAVE + lane] = (lf4){vthigh.a.y, vthigh.a.z, vthigh.l.x, vthigh.l.y};
    seg[5 * WAVE + lane] = (lf4){vthigh.l.z, phipa.x, phipa.y, phipa.z};      // hip capsule: its ends, and the hip joint's rate (the hip body's
    twx[5 * WAVE + lane] = (lf4){phipb.x, phipb.y, phipb.z, L.qd[0]};           //  twist = base twist + rate x the joint's motion subspace)
    LDS_PHASE();
    unsigned long long mybits = 0;
    const V3 own_p[3] = {pknee, pthigh, phipa}, own_q[3] = {pfoot, pknee, phipb};
    // Broad phase 1 (round 5; the capsule tests below were 14 % of the step for pairs that almost never touch): SEPARATING AXES.  The leg's
    // five points (hip capsule ends, thigh joint, knee, foot centre) span all of its capsules; their extents along the base's x and y axes
    // are exchanged in the quad (DPP rotations, no LDS), and two legs whose extents are further apart along either axis than the largest
    // radii sum + the contact distance cannot touch in any of the six capsule combinations.  Conservative (an axis test never rejects a
    // contact), so the listed pairs — and everything downstream — are exactly those of the full test.  Walking robots: every pair is
    // rejected here.
    const float sep = (float)GO1_HIP_CAPSULE_RADIUS + GO1_SELF_LEG_RADIUS + cd + 1e-4f;      // (hip - lower leg: the largest radii sum of a listed combination)
    float ex0, ex1, ey0, ey1;
    {
      const float tx = dot(pthigh, R0.c0), kx = dot(pknee, R0.c0), fx = dot(pfoot, R0.c0), ax = dot(phipa, R0.c0), bx = dot(phipb, R0.c0);
      const float ty = dot(pthigh, R0.c1), ky = dot(pknee, R0.c1), fy = dot(pfoot, R0.c1), ay = dot(phipa, R0.c1), by = dot(phipb, R0.c1);
      ex0 = fminf(fminf(tx, fminf(kx, fx)), fminf(ax, bx)); ex1 = fmaxf(fmaxf(tx, fmaxf(kx, fx)), fmaxf(ax, bx));
      ey0 = fminf(fminf(ty, fminf(ky, fy)), fminf(ay, by)); ey1 = fmaxf(fmaxf(ty, fmaxf(ky, fy)), fmaxf(ay, by));
    }
    // broad phase 2: the two segments' midpoints further apart than both half lengths + radii + contact distance
    const float reach = 2.f * 0.1065f + (float)GO1_HIP_CAPSULE_RADIUS + GO1_SELF_LEG_RADIUS + cd + 0.01f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int j = (leg + 1 + i) & 3, lj = (lane & ~3) | j;
      const float px0 = quad_rot(ex0, i + 1), px1 = quad_rot(ex1, i + 1), py0 = quad_rot(ey0, i + 1), py1 = quad_rot(ey1, i + 1);
      const bool apart = ex0 - px1 > sep || px0 - ex1 > sep || ey0 - py1 > sep || py0 - ey1 > sep;
      if (__ballot(!apart) == 0ull) continue;
      const lf4 a0 = seg[0 * WAVE + lj], a1 = seg[1 * WAVE + lj], a2 = seg[2 * WAVE + lj], h0 = seg[5 * WAVE + lj], h1 = twx[5 * WAVE + lj];
      const V3 par_p[3] = {v3(a0[0], a0[1], a0[2]), v3(a1[2], a1[3], a2[0]), v3(h0[1], h0[2], h0[3])};
      const V3 par_q[3] = {v3(a0[3], a1[0], a1[1]), v3(a0[0], a0[1], a0[2]), v3(h1[0], h1[1], h1[2])};
      const bool lower = leg < j;
      const int pair = lower ? leg_pair_index(leg, j) : leg_pair_index(j, leg);
      float best_phi = 1e30f;                 // two legs touch in ONE point: the deepest of the six capsule combinations (ties: lower type)
      int best_type = -1;
#pragma unroll
      for (int type = 0; type < SELF_TYPES; type++) {
        const int sa = self_seg_a(type), sbq = self_seg_b(type);             // segment of body A / B: 0 lower leg, 1 thigh, 2 hip capsule
        const int so = lower ? sa : sbq, sp = lower ? sbq : sa;               // own / partner segment
        const V3 mo = 0.5f * (own_p[so] + own_q[so]), mp = 0.5f * (par_p[sp] + par_q[sp]), dm = mo - mp;
        const bool near = !apart && dot(dm, dm) < reach * reach;
        if (__ballot(near) != 0ull) {
          Cand c;
          const float ra = self_seg_radius(sa), rb = self_seg_radius(sbq);
          const bool hit = near && (lower ? capsule_contact(own_p[so], own_q[so], ra, par_p[sp], par_q[sp], rb, cd, c)
                                          : capsule_contact(par_p[sp], par_q[sp], ra, own_p[so], own_q[so], rb, cd, c));
          if (hit && c.phi < best_phi) { best_phi = c.phi; best_type = type; }
        }
      }
      if (best_type >= 0) mybits |= 1ull << (6 * best_type + pair);
    }
    {
      // lower leg against the trunk's capsule (axis = the base's x axis through the origin): apart along the base's y or z axis?
      const float ta = (float)(GO1_TRUNK_BOX_HALF[0] - GO1_TRUNK_BOX_HALF[1]);
      const float sept = (float)GO1_TRUNK_BOX_HALF[1] + GO1_SELF_LEG_RADIUS + cd + 1e-4f;
      const float ky = dot(pknee, R0.c1), fy = dot(pfoot, R0.c1), kz = dot(pknee, R0.c2), fz = dot(pfoot, R0.c2);
      const bool apart = fminf(ky, fy) > sept || fmaxf(ky, fy) < -sept || fminf(kz, fz) > sept || fmaxf(kz, fz) < -sept;
      if (__ballot(!apart) != 0ull) {
        Cand c;
        if (!apart && capsule_contact(pknee, pfoot, GO1_SELF_LEG_RADIUS, mul(R0, v3(-ta, 0.f, 0.f)), mul(R0, v3(ta, 0.f, 0.f)), (float)GO1_TRUNK_BOX_HALF[1], cd, c))
          mybits |= 1ull << (SELF_LEGLEG + leg);
      }
    }
    smask = quad_or64(mybits);
    nS = __builtin_popcountll(smask);
  }

  PROF(26);
  // ---- solver contact list (oracle detect_contacts()): slots in the priority order feet, foot walls, self-contacts, trunk
  // corners, trunk wall, calves (first points, walls, second points), thighs (same), hips; at most MAXC, the rest is dropped
  // and counted per class ---------------------------------------------------------------------------------------------------
  // own items: 0 foot, 1 foot wall, 2 calf first, 3 calf wall, 4 calf second, 5 thigh first, 6 thigh wall, 7 thigh second,
  //            8 hip first, 9 hip second, 10 / 11 trunk corners 2 leg, 2 leg + 1, 12 trunk wall (lane 0)
  const int fk = ck[1].phi < ck[0].phi ? 1 : 0, ft = ct[1].phi < ct[0].phi ? 1 : 0, fh = ch[1].phi < ch[0].phi ? 1 : 0;
  enum { IT_FOOT = 0, IT_FOOTW, IT_CALF1, IT_CALFW, IT_CALF2, IT_THIGH1, IT_THIGHW, IT_THIGH2, IT_HIP1, IT_HIP2, IT_TR0, IT_TR1, IT_TRW, IT_N };
  int slot[IT_N];
  int K, nF;
  {
    const unsigned below = (1u << leg) - 1u;
    int base_ofs = 0;
    unsigned sig0 = 0, sig1 = 0;
    int drops[GO1_CC_COUNT];
#pragma unroll
    for (int c = 0; c < GO1_CC_COUNT; c++) drops[c] = 0;
    auto place = [&](int it, bool a, int cls) {
      const unsigned m = quad_ballot(a, lane);
      const int cnt = __popc(m);
      slot[it] = a ? base_ofs + __popc(m & below) : -1;
      const int over = base_ofs + cnt - (base_ofs > MAXC ? base_ofs : MAXC);
      if (over > 0) drops[cls] += over;
      base_ofs += cnt;
    };
    place(IT_FOOT, cf.phi < cd, GO1_CC_FOOT);
    place(IT_FOOTW, WALLS && cwf.phi < cd, GO1_CC_FOOT_WALL);
    nF = base_ofs;
    {
      const int over = base_ofs + nS - (base_ofs > MAXC ? base_ofs : MAXC);
      if (over > 0) drops[GO1_CC_SELF] += over;
      base_ofs += nS;
    }
    {   // trunk corners in corner order, at most MAXTR
      const bool a0 = cb[0].phi < cd, a1 = cb[1].phi < cd;
      const unsigned m0 = quad_ballot(a0, lane), m1 = quad_ballot(a1, lane);
      const int before = __popc(m0 & below) + __popc(m1 & below);
      const int r0 = before, r1 = before + (a0 ? 1 : 0);
      const int tot = __popc(m0) + __popc(m1), listed = tot < MAXTR ? tot : MAXTR;
      slot[IT_TR0] = (a0 && r0 < MAXTR) ? base_ofs + r0 : -1;
      slot[IT_TR1] = (a1 && r1 < MAXTR) ? base_ofs + r1 : -1;
      drops[GO1_CC_TRUNK] += tot - listed;
      const int over = base_ofs + listed - (base_ofs > MAXC ? base_ofs : MAXC);
      if (over > 0) drops[GO1_CC_TRUNK] += over;
      base_ofs += listed;
    }
    place(IT_TRW, WALLS && leg == 0 && cwb.phi < cd, GO1_CC_WALL);
    place(IT_CALF1, (fk ? ck[1].phi : ck[0].phi) < cd, GO1_CC_CALF);
    place(IT_CALFW, WALLS && cwk.phi < cd, GO1_CC_WALL);
    place(IT_CALF2, (fk ? ck[0].phi : ck[1].phi) < cd, GO1_CC_CALF);
    place(IT_THIGH1, (ft ? ct[1].phi : ct[0].phi) < cd, GO1_CC_THIGH);
    place(IT_THIGHW, WALLS && cwt.phi < cd, GO1_CC_WALL);
    place(IT_THIGH2, (ft ? ct[0].phi : ct[1].phi) < cd, GO1_CC_THIGH);
    place(IT_HIP1, (fh ? ch[1].phi : ch[0].phi) < cd, GO1_CC_HIP);
    place(IT_HIP2, (fh ? ch[0].phi : ch[1].phi) < cd, GO1_CC_HIP);
    K = base_ofs > MAXC ? MAXC : base_ofs;
#pragma unroll
    for (int it = 0; it < IT_N; it++) if (slot[it] >= MAXC) slot[it] = -1;
    if (base_ofs > MAXC || drops[GO1_CC_TRUNK] > 0) {
      if (leg == 0) {
        fault |= 1u << GO1_FAULT_CONTACT_DROPPED;
#pragma unroll
        for (int c = 0; c < GO1_CC_COUNT; c++) dropacc[c] += (uint32_t)drops[c];
      }
      sig1 |= 1u << 31;
    }
    if (SIG && B.contact_signature != nullptr && sub < GO1_SIG_MAX_SUBSTEPS) {      // tests only (the SIG instances): which points are listed
      if (slot[IT_FOOT] >= 0) sig0 |= 1u << leg;
      if (slot[IT_TR0] >= 0) sig0 |= 1u << (4 + 2 * leg);
      if (slot[IT_TR1] >= 0) sig0 |= 1u << (5 + 2 * leg);
      if (slot[IT_CALF1] >= 0) sig0 |= 1u << (12 + 2 * leg + fk);
      if (slot[IT_CALF2] >= 0) sig0 |= 1u << (12 + 2 * leg + 1 - fk);
      if (slot[IT_THIGH1] >= 0) sig0 |= 1u << (20 + 2 * leg + ft);
      if (slot[IT_THIGH2] >= 0) sig0 |= 1u << (20 + 2 * leg + 1 - ft);
      if (slot[IT_FOOTW] >= 0) sig1 |= 1u << leg;
      if (slot[IT_TRW] >= 0) sig1 |= 1u << 4;
      if (slot[IT_CALFW] >= 0) sig1 |= 1u << (5 + leg);
      if (slot[IT_THIGHW] >= 0) sig1 |= 1u << (9 + leg);
      if (slot[IT_HIP1] >= 0) sig1 |= 1u << (13 + 2 * leg + fh);
      if (slot[IT_HIP2] >= 0) sig1 |= 1u << (13 + 2 * leg + 1 - fh);
      unsigned sself = 0, pos = nF;          // self pairs that found a slot: 3 bits per pair of legs (1 + type), trunk pairs at 18 + leg
#pragma unroll 1
      for (int pid = 0; pid < SELF_LEGLEG + 4; pid++)
        if (smask & (1ull << pid)) {
          if ((int)pos < MAXC) sself |= pid < SELF_LEGLEG ? (unsigned)(pid / 6 + 1) << (3 * (pid % 6)) : 1u << (18 + pid - SELF_LEGLEG);
          pos++;
        }
      sig0 = quad_or(sig0); sig1 = quad_or(sig1);
      // geometry hash: the cell and candidate point of every listed terrain contact, weighted by its item index
      uint32_t gh = 0;
      auto hash_item = [&](int it, const Cand& c) { if (slot[it] >= 0) gh += c.tag * (uint32_t)(2 * it + 1) * 2654435761u; };
      hash_item(IT_FOOT, cf); hash_item(IT_FOOTW, cwf);
      hash_item(IT_CALF1, fk ? ck[1] : ck[0]); hash_item(IT_CALFW, cwk); hash_item(IT_CALF2, fk ? ck[0] : ck[1]);
      hash_item(IT_THIGH1, ft ? ct[1] : ct[0]); hash_item(IT_THIGHW, cwt); hash_item(IT_THIGH2, ft ? ct[0] : ct[1]);
      hash_item(IT_HIP1, fh ? ch[1] : ch[0]); hash_item(IT_HIP2, fh ? ch[0] : ch[1]);
      hash_item(IT_TR0, cb[0]); hash_item(IT_TR1, cb[1]); hash_item(IT_TRW, cwb);
      gh += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gh, 0xB1, 0xF, 0xF, false);
      gh += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gh, 0x4E, 0xF, 0xF, false);
      if (leg == 0) {
        AT(B.contact_signature, sub * GO1_SIG_WORDS + 3, e) = gh;
        AT(B.contact_signature, sub * GO1_SIG_WORDS + 0, e) = sig0;
        AT(B.contact_signature, sub * GO1_SIG_WORDS + 1, e) = sig1;
        AT(B.contact_signature, sub * GO1_SIG_WORDS + 2, e) = sself;       // (limit-row legs are OR-ed in below)
      }
    }
  }
  PROF(19);
  // ---- post the listed terrain points as items into their records (q9 .. q11), invalidate the records the previous
  // substep used beyond this one's count (their impulses and inverse diagonals: the sweep then leaves them at zero) ----------
  // mass splitting of the sweep's leg phase (see "SWEEP ORDER" below): n = the legs of the environment that hold hip / thigh rows
  const bool has_split = slot[IT_THIGH1] >= 0 || slot[IT_THIGHW] >= 0 || slot[IT_THIGH2] >= 0 || slot[IT_HIP1] >= 0 || slot[IT_HIP2] >= 0;
  const int nsplit_legs = __popc(quad_ballot(has_split, lane));
  const int nsm1i = nsplit_legs > 1 ? nsplit_legs - 1 : 0;            // n - 1 (0: a single leg holds such rows — nothing to split)
  {
    // warm start: a body's previous impulse is shared equally by its listed top-surface points; wall points start from zero
    const float share_k = (slot[IT_CALF1] >= 0 && slot[IT_CALF2] >= 0) ? 0.5f : 1.f, share_t = (slot[IT_THIGH1] >= 0 && slot[IT_THIGH2] >= 0) ? 0.5f : 1.f,
                share_h = (slot[IT_HIP1] >= 0 && slot[IT_HIP2] >= 0) ? 0.5f : 1.f;
    const int ntr = __popc(quad_ballot(slot[IT_TR0] >= 0, lane)) + __popc(quad_ballot(slot[IT_TR1] >= 0, lane));
    const float share_b = ntr > 0 ? 1.f / (float)ntr : 0.f;
    auto post = [&](int k, const Cand& c, int depth, int body, float share) {
      CRQ(k, 9) = (lf4){c.phi, c.x, c.y, c.z};
      CRQ(k, 10) = (lf4){c.un, c.nx, c.ny, c.nz};
      CRQ(k, 11) = (lf4){(float)depth, (float)(leg + ((depth == 0 || depth == 1) ? 4 * nsm1i : 0)), (float)body, share};      // (leg + 4 (n - 1): emit_terrain_contact)
    };
    if (slot[IT_FOOT] >= 0) post(slot[IT_FOOT], cf, 2, 4 + 4 * leg, 1.f);
    if (WALLS && slot[IT_FOOTW] >= 0) post(slot[IT_FOOTW], cwf, 2, 4 + 4 * leg, 0.f);
    if (slot[IT_CALF1] >= 0) post(slot[IT_CALF1], fk ? ck[1] : ck[0], 2, 3 + 4 * leg, share_k);
    if (WALLS && slot[IT_CALFW] >= 0) post(slot[IT_CALFW], cwk, 2, 3 + 4 * leg, 0.f);
    if (slot[IT_CALF2] >= 0) post(slot[IT_CALF2], fk ? ck[0] : ck[1], 2, 3 + 4 * leg, share_k);
    if (slot[IT_THIGH1] >= 0) post(slot[IT_THIGH1], ft ? ct[1] : ct[0], 1, 2 + 4 * leg, share_t);
    if (WALLS && slot[IT_THIGHW] >= 0) post(slot[IT_THIGHW], cwt, 1, 2 + 4 * leg, 0.f);
    if (slot[IT_THIGH2] >= 0) post(slot[IT_THIGH2], ft ? ct[0] : ct[1], 1, 2 + 4 * leg, share_t);
    if (slot[IT_HIP1] >= 0) post(slot[IT_HIP1], fh ? ch[1] : ch[0], 0, 1 + 4 * leg, share_h);
    if (slot[IT_HIP2] >= 0) post(slot[IT_HIP2], fh ? ch[0] : ch[1], 0, 1 + 4 * leg, share_h);
    if (slot[IT_TR0] >= 0) post(slot[IT_TR0], cb[0], -1, 0, share_b);
    if (slot[IT_TR1] >= 0) post(slot[IT_TR1], cb[1], -1, 0, share_b);
    if (WALLS && slot[IT_TRW] >= 0) post(slot[IT_TRW], cwb, -1, 0, 0.f);
    if (leg == 0) {
      const int Kprev = (int)LDS(L_KL + 2);
#pragma unroll 1
      for (int k = K; k < Kprev; k++) { CRQ(k, 2) = CRQ(k, 5) = CRQ(k, 8) = CRQ(k, 9) = (lf4){0.f, 0.f, 0.f, 0.f}; }
      LDS(L_KL) = (float)K; LDS(L_KL + 2) = (float)K; LDS(L_KL + 3) = 0.f; LDS(L_KL + 4) = (float)nF; LDS(L_KL + 5) = (float)nS; LDS(L_KL + 6) = 0.f;
    }
  }
  PROF(20);
  if (acth) {
    BLOCK_SYNC(nw);                       // the helpers' partial sums are in io[A_OUT]
    torque_collect(cfg, L, acth, Z.act_io(), lane, leg, fault);
  }
  // ---- ABA pass 2: calf -> thigh -> hip, then quad-sum into the base -------------------------------------------------------
  {
    SV pa_hip;
#pragma unroll
    for (int j = 2; j >= 0; j--) {
      F.U[j] = sym6_mul(IA[j], F.S[j]);
      float D = dot(F.S[j], F.U[j]);
      if (!(D > 1e-9f)) { fault |= 1u << GO1_FAULT_JOINT_D; D = 1e-9f; }
      const float rs = rsqrtf(D);
      F.sD[j] = rs;
      F.Dinv[j] = rs * rs;
      uu[j] = L.tau[j] - dot(F.S[j], pA[j]);
      sym6_rank1_sub(IA[j], F.U[j], F.Dinv[j]);
      SV pa = pA[j] + sym6_mul(IA[j], cj[j]) + (uu[j] * F.Dinv[j]) * F.U[j];
      if (j > 0) { sym6_add(IA[j - 1], IA[j]); pA[j - 1] = pA[j - 1] + pa; }
      else pa_hip = pa;
    }
#pragma unroll
    for (int i = 0; i < 21; i++) IA0.m[i] += quad_sum(IA[0].m[i]);
    pA0 = pA0 + quad_sum(pa_hip);
  }
  PROF(2);
  // ---- ABA pass 3 ------------------------------------------------------------------------------------
  EnvPk E;
  float min_pivot;
  const Sym6 I0inv = sym6_inverse(IA0, min_pivot, E.Li);
  if (!(min_pivot > 1e-9f)) fault |= 1u << GO1_FAULT_BASE_PIVOT;
  SV a0 = -sym6_mul(I0inv, pA0);
  const V3 w_free = s.w + h * a0.a;
  const V3 v_free = s.v + h * (a0.l + cross(s.w, s.v));
  {
    SV a = a0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      SV ap = a + cj[j];
      float qdd = F.Dinv[j] * (uu[j] - dot(F.U[j], ap));
      a = ap + qdd * F.S[j];
      F.qdf[j] = L.qd[j] + h * qdd;
    }
  }
  E.w_free = w_free; E.v_free = v_free; E.e_c = 0.5f * (s.rest + cfg.terrain_restitution); E.warm = use_warm ? 1.f : 0.f;

  // ---- joint-limit rows of the own leg ------------------------------------------------------------------
  // A joint's position / velocity limits are ONE solver row: a generalised impulse along the joint coordinate (equal and
  // opposite on child and parent) keeps the rate inside [vlo, vhi] = [max((lo - q)/h, -vmax), min((hi - q)/h, vmax)].
  // Clamping the joint coordinate after the solve instead is an unbalanced impulse: a torque held against a stop then
  // acts on the base without reaction (free thrust — the root cause of round 1's non-finite states).
  // The rows of a leg enter the solve together, as soon as one free rate comes within the margins of its band.
  float jlo[3], jhi[3];
  bool legact = false;
  {
    const float mv = cfg.joint_limit_margin, mp = cfg.joint_limit_pos_margin / h;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      float lo = (GO1_JOINT_LOWER[ji] - L.q[j]) / h, hi = (GO1_JOINT_UPPER[ji] - L.q[j]) / h;
      const float vl = GO1_JOINT_VEL_LIMIT[ji];
      lo = fminf(lo, GO1_LIMIT_RECOVERY_RATE);
      hi = fmaxf(hi, -GO1_LIMIT_RECOVERY_RATE);
      jlo[j] = fmaxf(lo, -vl);
      jhi[j] = fminf(hi, vl);
      const float vf = F.qdf[j];
      if (!(vf > lo + mp && vf < hi - mp && vf > -vl + mv && vf < vl - mv)) legact = true;
    }
  }
  const unsigned lact = quad_ballot(legact, lane);      // legs of this environment whose limit rows are in the solve
  if (SIG && B.contact_signature != nullptr && sub < GO1_SIG_MAX_SUBSTEPS && leg == 0) AT(B.contact_signature, sub * GO1_SIG_WORDS + 2, e) |= lact << 28;

  // ---- hand-over: the leg's factors, the base's free twist and L^-1; the terrain contacts are finished by the helper
  // wavefronts (nw > 1) while this one emits the self-contacts and the limit rows --------------------------------------------
  legfac_store(Z.pkl(), lane, F);
  if (leg == 0) { envpk_store(Z.pke(), el, E); LDS(L_KL + 1) = (float)lact; }
  if (GO1_RARE(cfg.self_collision && __ballot(smask != 0ull) != 0ull)) {        // free twists of the own lower leg / thigh / hip for the partners
    SV fl = sv(w_free, v_free), ft_;
#pragma unroll
    for (int j = 0; j < 3; j++) { fl = fl + F.qdf[j] * F.S[j]; if (j == 1) ft_ = fl; }
    lf4* tw = Z.tw();
    tw[0 * WAVE + lane] = (lf4){fl.a.x, fl.a.y, fl.a.z, fl.l.x};
    tw[1 * WAVE + lane] = (lf4){fl.l.y, fl.l.z, ft_.a.x, ft_.a.y};
    tw[2 * WAVE + lane] = (lf4){ft_.a.z, ft_.l.x, ft_.l.y, ft_.l.z};
    tw[3 * WAVE + lane] = (lf4){F.qdf[0], 0.f, 0.f, 0.f};                       // (hip body: free rate of the hip joint; see the geometry block)
  }
  BLOCK_SYNC(nw);
  PROF(4);
  // self-contacts (rare: skipped unless some environment of the wavefront lists one).  Pair pid at slot nF + rank(pid).  The
  // lane of body A (the lower-numbered leg; the leg of a trunk pair) owns the record; for a leg-leg pair the lane of body B
  // hands its side over through the pair's SB slots.
  int sslot[MAXSB + 1];            // slots of the (at most MAXSB + 1) listed pairs this lane is body A or B of ...
  int sdepth[MAXSB + 1];           // ... own body's depth (0 hip, 1 thigh, 2 lower leg) ...
  float ssign[MAXSB + 1];          // ... and the sign of the impulse it receives (+1 body A, -1 body B)
  int nown = 0;
#pragma unroll
  for (int i = 0; i <= MAXSB; i++) { sslot[i] = -1; sdepth[i] = 2; ssign[i] = 0.f; }
  if (GO1_RARE(cfg.self_collision && __ballot(smask != 0ull) != 0ull)) {
    const lf4* seg = Z.seg();
    const lf4* tw = Z.tw();
    int rank = 0, sbi = 0;         // rank among the listed pairs / among the listed leg-leg pairs (wave-uniform loop, per-lane counters)
#pragma unroll 1
    for (int pid = 0; pid < SELF_LEGLEG + 4; pid++) {
      const bool on = (smask & (1ull << pid)) != 0ull;
      if (__ballot(on) == 0ull) continue;
      const int type = pid < SELF_LEGLEG ? pid / 6 : 0, pr = pid % 6;
      const int lo_ = pid < SELF_LEGLEG ? (pr < 3 ? 0 : pr < 5 ? 1 : 2) : pid - SELF_LEGLEG, hi_ = pid < SELF_LEGLEG ? (pr < 3 ? pr + 1 : pr < 5 ? pr - 1 : 3) : -1;
      const int sa = self_seg_a(type), sbq = self_seg_b(type);
      const int k = nF + rank;
      const bool isA = on && leg == lo_ && k < MAXC, isB = on && leg == hi_ && k < MAXC;
      Cand c;
      cand_init(c);
      V3 n = v3(0.f, 0.f, 1.f), x = v3(0.f, 0.f, 0.f), t1 = v3(1.f, 0.f, 0.f), t2 = v3(0.f, 1.f, 0.f);
      SV g[3];
      float uj[3][3];
      int depth = 2;
      const int lj = (lane & ~3) | (isA ? (hi_ < 0 ? leg : hi_) : (lo_ & 3));       // partner lane (trunk pair: unused)
      if (isA || isB) {
        const lf4 a0 = seg[0 * WAVE + lj], a1 = seg[1 * WAVE + lj], a2 = seg[2 * WAVE + lj], h0 = seg[5 * WAVE + lj], h1 = tw[5 * WAVE + lj];
        const V3 par_p[3] = {v3(a0[0], a0[1], a0[2]), v3(a1[2], a1[3], a2[0]), v3(h0[1], h0[2], h0[3])};
        const V3 par_q[3] = {v3(a0[3], a1[0], a1[1]), v3(a0[0], a0[1], a0[2]), v3(h1[0], h1[1], h1[2])};
        const V3 own_p[3] = {pknee, pthigh, phipa}, own_q[3] = {pfoot, pknee, phipb};
        const int so = isA ? sa : sbq, sp = isA ? sbq : sa;
        const float ra = self_seg_radius(sa);
        if (hi_ < 0) {
          const float ta = (float)(GO1_TRUNK_BOX_HALF[0] - GO1_TRUNK_BOX_HALF[1]);
          capsule_contact(pknee, pfoot, ra, mul(R0, v3(-ta, 0.f, 0.f)), mul(R0, v3(ta, 0.f, 0.f)), (float)GO1_TRUNK_BOX_HALF[1], cd, c);
        } else {
          const float rb = self_seg_radius(sbq);
          if (isA) capsule_contact(own_p[so], own_q[so], ra, par_p[sp], par_q[sp], rb, cd, c);
          else capsule_contact(par_p[sp], par_q[sp], ra, own_p[so], own_q[so], rb, cd, c);
        }
        n = v3(c.nx, c.ny, c.nz); x = v3(c.x, c.y, c.z);
        contact_frame(n, t1, t2, fault);
        depth = 2 - so;
        if (nown <= MAXSB) {
#pragma unroll
          for (int i = 0; i <= MAXSB; i++) if (i == nown) { sslot[i] = k; sdepth[i] = depth; ssign[i] = isA ? 1.f : -1.f; }
        }
        nown++;
        // own side of the three rows: +d on body A, -d on body B
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const V3 d = r == 0 ? n : r == 1 ? t1 : t2;
          const SV unit = sv(cross(x, d), d);
          row_propagate(F, depth, isA ? -unit : unit, g[r], uj[r]);
          if (hi_ < 0) g[r] = g[r] + unit;                               // the trunk's side: the opposite unit wrench on the base
        }
        if (isB) {                                                         // hand g_B and u_B over
          SBQ(sbi, 0) = (lf4){g[0].a.x, g[0].a.y, g[0].a.z, g[0].l.x};
          SBQ(sbi, 1) = (lf4){g[0].l.y, g[0].l.z, g[1].a.x, g[1].a.y};
          SBQ(sbi, 2) = (lf4){g[1].a.z, g[1].l.x, g[1].l.y, g[1].l.z};
          SBQ(sbi, 3) = (lf4){g[2].a.x, g[2].a.y, g[2].a.z, g[2].l.x};
          SBQ(sbi, 4) = (lf4){g[2].l.y, g[2].l.z, uj[0][0] * F.sD[0], uj[0][1] * F.sD[1]};
          SBQ(sbi, 5) = (lf4){uj[0][2] * F.sD[2], uj[1][0] * F.sD[0], uj[1][1] * F.sD[1], uj[1][2] * F.sD[2]};
          SBQ(sbi, 6) = (lf4){uj[2][0] * F.sD[0], uj[2][1] * F.sD[1], uj[2][2] * F.sD[2], (float)leg};
        }
      }
      LDS_PHASE();                 // (wave-uniform point: the loop and its skips depend on ballots only)
      if (isA) {
        // relative velocities (A - B) at the contact point: before the step (restitution) and free
        const lf4 s2 = seg[2 * WAVE + lj], s3 = seg[3 * WAVE + lj], s4 = seg[4 * WAVE + lj], s5 = seg[5 * WAVE + lj];
        const lf4 f0 = tw[0 * WAVE + lj], f1 = tw[1 * WAVE + lj], f2 = tw[2 * WAVE + lj];
        SV preB, freeB;
        if (hi_ < 0) { preB = v0; freeB = sv(w_free, v_free); }
        else if (sbq == 2) {         // partner's hip body: base twist + hip rate x the hip joint's motion subspace (axis = the base's x axis)
          const V3 p0 = mul(R0, model_v3(GO1_JOINT_ORIGIN, 3 * (lj & 3)));
          const SV S0 = sv(R0.c0, cross(p0, R0.c0));
          preB = v0 + tw[5 * WAVE + lj][3] * S0;
          freeB = sv(w_free, v_free) + tw[3 * WAVE + lj][0] * S0;
        }
        else if (sbq == 1) { preB = sv(v3(s3[3], s4[0], s4[1]), v3(s4[2], s4[3], s5[0])); freeB = sv(v3(f1[2], f1[3], f2[0]), v3(f2[1], f2[2], f2[3])); }
        else { preB = sv(v3(s2[1], s2[2], s2[3]), v3(s3[0], s3[1], s3[2])); freeB = sv(v3(f0[0], f0[1], f0[2]), v3(f0[3], f1[0], f1[1])); }
        const SV preA = sa == 2 ? vhip : sa == 1 ? vthigh : vlow;
        SV freeA = sv(w_free, v_free);
#pragma unroll
        for (int j = 0; j < 3; j++) if (j <= depth) freeA = freeA + F.qdf[j] * F.S[j];
        const float un_pre = dot(n, (preA.l + cross(preA.a, x)) - (preB.l + cross(preB.a, x)));
        const V3 vrel = (freeA.l + cross(freeA.a, x)) - (freeB.l + cross(freeB.a, x));
        float vs = fminf(-c.phi / h, cfg.max_depenetration_velocity);
        const bool bounce = un_pre < -cfg.bounce_threshold_velocity && -s.rest * un_pre > vs;      // robot-robot: the robot's own material
        if (bounce) vs = -s.rest * un_pre;
        Row a[3], ab[3];
        float wB[5] = {0.f, 0.f, 0.f, 0.f, 0.f};          // u_B . u_B terms: nn, t1t1, t2t2, t1n, t2n
        if (hi_ >= 0) {
          const lf4 b0 = SBQ(sbi, 0), b1 = SBQ(sbi, 1), b2 = SBQ(sbi, 2), b3 = SBQ(sbi, 3), b4 = SBQ(sbi, 4), b5 = SBQ(sbi, 5), b6 = SBQ(sbi, 6);
          g[0] = g[0] + sv(v3(b0[0], b0[1], b0[2]), v3(b0[3], b1[0], b1[1]));
          g[1] = g[1] + sv(v3(b1[2], b1[3], b2[0]), v3(b2[1], b2[2], b2[3]));
          g[2] = g[2] + sv(v3(b3[0], b3[1], b3[2]), v3(b3[3], b4[0], b4[1]));
          ab[0].u[0] = b4[2]; ab[0].u[1] = b4[3]; ab[0].u[2] = b5[0];
          ab[1].u[0] = b5[1]; ab[1].u[1] = b5[2]; ab[1].u[2] = b5[3];
          ab[2].u[0] = b6[0]; ab[2].u[1] = b6[1]; ab[2].u[2] = b6[2];
          auto d3 = [](const Row& p, const Row& q) { return fmaf(p.u[0], q.u[0], fmaf(p.u[1], q.u[1], p.u[2] * q.u[2])); };
          wB[0] = d3(ab[0], ab[0]); wB[1] = d3(ab[1], ab[1]); wB[2] = d3(ab[2], ab[2]); wB[3] = d3(ab[1], ab[0]); wB[4] = d3(ab[2], ab[0]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) row_finish(E.Li, g[r], uj[r], F.sD, a[r]);
        contact_record_store(crl, el, k, a[0], a[1], a[2], cr_flags(leg, true, hi_ >= 0 ? sbi + 1 : 0), dot(n, vrel) - vs, dot(t1, vrel), dot(t2, vrel),
                             row_dot(a[0], a[0]) + wB[0], row_dot(a[1], a[1]) + wB[1], row_dot(a[2], a[2]) + wB[2],
                             row_dot(a[1], a[0]) + wB[3], row_dot(a[2], a[0]) + wB[4], v3(0.f, 0.f, 0.f), x, n, bounce, fault);
      }
      if (on) { rank++; if (pid < SELF_LEGLEG) sbi++; }
    }
  }
  if (legact) {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      float uj[3] = {0.f, 0.f, 0.f};
      uj[jj] = 1.f;
      SV pAj = F.Dinv[jj] * F.U[jj];
#pragma unroll
      for (int j = 1; j >= 0; j--) {
        if (j < jj) {
          const float u = -dot(F.S[j], pAj);
          uj[j] = u;
          pAj = pAj + (u * F.Dinv[j]) * F.U[j];
        }
      }
      Row a;
      row_finish(E.Li, pAj, uj, F.sD, a);
      const float w = row_dot(a, a);
      if (!(w > 1e-9f)) fault |= 1u << GO1_FAULT_W_DIAG;
      const int j = 3 * leg + jj;
      JRQ(j, 0) = (lf4){a.g[0], a.g[1], a.g[2], a.g[3]};
      JRQ(j, 1) = (lf4){a.g[4], a.g[5], a.u[0], a.u[1]};
      JRQ(j, 2) = (lf4){a.u[2], F.qdf[jj], jlo[jj], jhi[jj]};
      JRQ(j, 3) = (lf4){w, 1.f / w, 0.f, 0.f};
    }
  }
  BLOCK_SYNC(nw);
  PROF(21);
  if (LDS(L_KL + 3) != 0.f) fault |= 1u << GO1_FAULT_CONTACT_FRAME;
  if (LDS(L_KL + 6) != 0.f) fault |= 1u << GO1_FAULT_W_DIAG;
  if (SIG && B.contact_signature != nullptr && sub < GO1_SIG_MAX_SUBSTEPS && leg == 0) {      // which contacts took the restitution branch
    uint32_t bh = 0;
#pragma unroll 1
    for (int k = 0; k < K; k++) if (CRQ(k, 9)[3] != 0.f) bh += (uint32_t)(k + 1) * 0x9E3779B1u;
    AT(B.contact_signature, sub * GO1_SIG_WORDS + 3, e) += bh;
  }
  const SolveBounds sm = solve_bounds(K, legact);
  const int Kw = sm.Kw;
  const unsigned LAw = sm.LAw;
  PROF(18);
  // ---- projected Gauss-Seidel on the impulses, matrix-free ---------------------------------------------------------------
  // State s = sum_r lambda_r a_r: z (6, replicated in the quad) and y (3, the own leg's).  A row's velocity is b_r + a_r . s;
  // the leg part of a contact's three rows is evaluated by every lane on its own y, masked by "the contact is on my leg"
  // and summed over the quad.  Contacts in list order (normal, then the two tangents against the updated normal impulse,
  // Coulomb cone: static inside, dynamic when sliding), then the limit rows in joint order.
  const float mu_s = 0.5f * (s.mu + cfg.terrain_friction);       // PhysX default combine mode: average
  const float mu_d = fminf(0.5f * (s.mu + cfg.terrain_dynamic_friction), mu_s);
  float lamj[3] = {0.f, 0.f, 0.f};                                 // limit impulses of the own joints
#ifndef GO1_ABLATE_PGS
  {
    SweepState st;
    st.z01 = st.z23 = st.z45 = st.y01 = splat2(0.f);
    st.y2 = 0.f;
    // B side of a leg-leg self-contact: u_B of the three rows and its leg
    struct RowB { float u[3][3]; int legB; };
    auto load_b = [&](int sb1, RowB& Bq) {
      const int i = sb1 > 0 ? sb1 - 1 : 0;
      const lf4 b4 = SBQ(i, 4), b5 = SBQ(i, 5), b6 = SBQ(i, 6);
      Bq.u[0][0] = b4[2]; Bq.u[0][1] = b4[3]; Bq.u[0][2] = b5[0];
      Bq.u[1][0] = b5[1]; Bq.u[1][1] = b5[2]; Bq.u[1][2] = b5[3];
      Bq.u[2][0] = b6[0]; Bq.u[2][1] = b6[1]; Bq.u[2][2] = b6[2];
      Bq.legB = (int)b6[3];
    };
    // SWEEP ORDER (the contract since round 5; oracle: physics_substep's sweep in oracle/go1_oracle.c, order 3; measured against the list order
    // in profiles/r05_solver_order_study.txt): per sweep, trunk and body-body contacts in list order ("cooperative" turns: the quad works on
    // one contact), then the terrain contacts of the four legs SIDE BY SIDE — every lane walks through the contacts of ITS leg (Gauss-Seidel
    // inside the leg) on a private copy of the base state, the legs' base changes are added up once per sweep (block Jacobi over legs).  The
    // serial length of a sweep is then the cooperative count + the largest count on one leg instead of the environment's total.
    // MASS SPLITTING for hip and thigh contacts: they couple to the base through one or two joints, and plain block Jacobi over them
    // over-corrects the base (every leg stops the WHOLE base: a robot lying on its side creeps, more sweeps do not cure it).  In the leg
    // phase the base therefore answers a hip / thigh row's impulse n times as strongly as it really does — n = the legs of the environment
    // holding such rows; the base as n sub-bodies of 1 / n of its articulated inertia, one per leg — which makes the iteration convergent
    // whatever the coupling; when the legs meet, the TRUE response of the impulse changes is what enters the base state.  Lower-leg rows
    // (foot, calf: three joints away from the base) are not split: walking robots solve exactly as under plain block Jacobi.
    // contact indices: own leg's contacts (all / the hip and thigh ones) and the cooperative ones of the environment — straight from the slots
    // the list gave out
    uint32_t mine = 0u, mine_split = 0u, coop = 0u;
    {
      const int leg_items[10] = {IT_FOOT, IT_FOOTW, IT_CALF1, IT_CALFW, IT_CALF2, IT_THIGH1, IT_THIGHW, IT_THIGH2, IT_HIP1, IT_HIP2};
#pragma unroll
      for (int i = 0; i < 10; i++)
        if (slot[leg_items[i]] >= 0) { mine |= 1u << slot[leg_items[i]]; if (i >= 5) mine_split |= 1u << slot[leg_items[i]]; }
      uint32_t tr = 0u;
      if (slot[IT_TR0] >= 0) tr |= 1u << slot[IT_TR0];
      if (slot[IT_TR1] >= 0) tr |= 1u << slot[IT_TR1];
      if (slot[IT_TRW] >= 0) tr |= 1u << slot[IT_TRW];
      const int s1 = nF + nS < K ? nF + nS : K;                      // body-body contacts: [nF, nF + nS) as far as they were listed
      coop = quad_or(tr) | (s1 > nF ? ((1u << s1) - 1u) & ~((1u << nF) - 1u) : 0u);
    }
    const float nsm1 = (float)nsm1i;
    const bool split_w = __ballot(mine_split != 0u && nsm1 > 0.f) != 0ull;     // wave-uniform: some environment of the wavefront splits
    uint32_t coop_w = 0u;                                          // wave-uniform: indices some environment of the wavefront treats cooperatively
#pragma unroll 1
    for (int k = 0; k < Kw; k++)
      if (__ballot((coop >> k) & 1u) != 0ull) coop_w |= 1u << k;
    int turns = 0;                                                 // wave-uniform: the largest number of contacts on one leg
    {
      const int cnt = __builtin_popcount(mine);
#pragma unroll 1
      for (int tt = 1; tt <= MAXC; tt++) { if (__ballot(cnt >= tt) == 0ull) break; turns = tt; }
    }
    // warm start: the state of the starting impulses (self-contacts and limit rows start from zero).  Like the sweep: the cooperative
    // contacts by the whole quad, the own leg's contacts by their lane — base parts summed over the quad once — instead of every lane
    // walking through every record of the environment (round 4: 6.6 % of the step)
#pragma unroll 1
    for (uint32_t rest = coop_w; rest != 0u; rest &= rest - 1u) {
      const int k = __builtin_ctz(rest);
      SweepRec R;
      sweep_rec_load(crl, el, k, R);
      const float g = ((coop >> k) & 1u) ? 1.f : 0.f;
      sweep_add_rows(R, st, (((int)R.q[8][3]) & 7) == leg ? 1.f : 0.f, g * R.q[9][0], g * R.q[9][1], g * R.q[9][2]);
    }
    {
      SweepState so;
      so.z01 = so.z23 = so.z45 = so.y01 = splat2(0.f);
      so.y2 = 0.f;
      uint32_t rem = mine;
#pragma unroll 1
      for (int tt = 0; tt < turns; tt++) {
        if (rem != 0u) {
          const int k = __builtin_ctz(rem);
          rem &= rem - 1u;
          SweepRec R;
          sweep_rec_load(crl, el, k, R);
          sweep_add_rows(R, so, 1.f, R.q[9][0], R.q[9][1], R.q[9][2]);
        }
      }
      st.z01[0] += quad_sum(so.z01[0]); st.z01[1] += quad_sum(so.z01[1]);
      st.z23[0] += quad_sum(so.z23[0]); st.z23[1] += quad_sum(so.z23[1]);
      st.z45[0] += quad_sum(so.z45[0]); st.z45[1] += quad_sum(so.z45[1]);
      st.y01 += so.y01; st.y2 += so.y2;
    }
    LDS_PHASE();            // every lane has read every contact's starting impulse before a lane overwrites its own contacts' (lock step on the hardware)
    PROF(23);
    // one cooperative contact's turn (its record already in registers).  `on` = this environment takes part; the others run the same
    // instructions — the turn holds wave-level operations — on their record k with their changes gated to zero
    auto contact_turn = [&](int k, const SweepRec& R, bool on) {
      const int fl = (int)R.q[8][3];
      const int legA = fl & 7, sb1 = fl >> 4;
      const bool self = (fl & 8) != 0;
      const float m = legA == leg ? 1.f : 0.f;
      float dn, d1, d2, pn, p1, p2;
      sweep_row_dot(R, 0, st, dn, pn);
      sweep_row_dot(R, 1, st, d1, p1);
      sweep_row_dot(R, 2, st, d2, p2);
      pn *= m; p1 *= m; p2 *= m;
      RowB Bq;
      float mB = 0.f;
      const bool anyB = __ballot(sb1 > 0) != 0ull;                    // leg-leg self-contact somewhere in the wavefront at this slot
      if (anyB) {
        load_b(sb1, Bq);
        mB = (sb1 > 0 && Bq.legB == leg) ? 1.f : 0.f;
        const float y0 = st.y01[0], y1 = st.y01[1];
        pn = fmaf(mB, fmaf(Bq.u[0][0], y0, fmaf(Bq.u[0][1], y1, Bq.u[0][2] * st.y2)), pn);
        p1 = fmaf(mB, fmaf(Bq.u[1][0], y0, fmaf(Bq.u[1][1], y1, Bq.u[1][2] * st.y2)), p1);
        p2 = fmaf(mB, fmaf(Bq.u[2][0], y0, fmaf(Bq.u[2][1], y1, Bq.u[2][2] * st.y2)), p2);
      }
      const float un = R.q[2][1] + dn + quad_sum(pn);                 // = u_n - v*
      float u1 = R.q[5][1] + d1 + quad_sum(p1), u2 = R.q[8][1] + d2 + quad_sum(p2);
      const float ln_old = R.q[9][0];
      const float ln = fmaxf(0.f, ln_old - un * R.q[2][2]);
      const float dln = ln - ln_old;
      u1 = fmaf(R.q[2][3], dln, u1);                                   // the tangential rows see the updated normal impulse
      u2 = fmaf(R.q[5][3], dln, u2);
      float l1 = R.q[9][1] - u1 * R.q[5][2];
      float l2 = R.q[9][2] - u2 * R.q[8][2];
      const float lim = (self ? s.mu : mu_s) * ln, nn = l1 * l1 + l2 * l2;      // robot-robot: the robot's own material
      if (nn > lim * lim) { const float sc = (self ? s.mu : mu_d) * ln * __builtin_amdgcn_rsqf(nn); l1 *= sc; l2 *= sc; }
      float dq = dln, e1 = l1 - R.q[9][1], e2 = l2 - R.q[9][2];
      if (!on) dq = e1 = e2 = 0.f;
      sweep_add_rows(R, st, m, dq, e1, e2);
      if (anyB) {
        const float m0 = mB * dq, m1 = mB * e1, m2 = mB * e2;
        st.y01[0] = fmaf(Bq.u[0][0], m0, fmaf(Bq.u[1][0], m1, fmaf(Bq.u[2][0], m2, st.y01[0])));
        st.y01[1] = fmaf(Bq.u[0][1], m0, fmaf(Bq.u[1][1], m1, fmaf(Bq.u[2][1], m2, st.y01[1])));
        st.y2 = fmaf(Bq.u[0][2], m0, fmaf(Bq.u[1][2], m1, fmaf(Bq.u[2][2], m2, st.y2)));
      }
      // (SIG instances: the 4th word carries "this turn projected the friction impulse on the cone" for the test signature)
      if (leg == 0 && on) CRQ(k, 9) = (lf4){ln, l1, l2, SIG && nn > lim * lim ? 1.f : 0.f};
    };
    uint32_t sig_active = 0u;
#pragma unroll 1
    for (int it = 0; it < cfg.solver_iterations; it++) {
      for (uint32_t rest = coop_w; rest != 0u; rest &= rest - 1u) {
        const int k = __builtin_ctz(rest);
        SweepRec R;
        sweep_rec_load(crl, el, k, R);
        contact_turn(k, R, ((coop >> k) & 1u) != 0u);
      }
      PROF(31);
      {
        SweepState sp = st;                                        // private copy: base state as the leg phase found it + the own contacts' changes
        f2 ds01 = splat2(0.f), ds23 = splat2(0.f), ds45 = splat2(0.f);      // sum of a_z dlambda over the own SPLIT rows (unscaled)
        uint32_t rem = mine;
#pragma unroll 1
        for (int tt = 0; tt < turns; tt++) {
          const int k = rem != 0u ? __builtin_ctz(rem) : 0;
          const bool on = rem != 0u;
          rem &= rem - 1u;
          const bool split = on && ((mine_split >> k) & 1u) != 0u && nsm1 > 0.f;
          const bool any_split = split_w && __ballot(split) != 0ull;      // (wave-uniform: the split arithmetic sits behind a scalar branch)
          if (on) {
            SweepRec R;
            sweep_rec_load(crl, el, k, R);
            float dn, d1, d2, pn, p1, p2;
            sweep_row_dot(R, 0, sp, dn, pn);
            sweep_row_dot(R, 1, sp, d1, p1);
            sweep_row_dot(R, 2, sp, d2, p2);
            // (1 / W of the three rows and the normal's weight in the tangent rows are those of the SPLIT system for a split row: emit_terrain_contact)
            const float iwn = R.q[2][2], iw1 = R.q[5][2], iw2 = R.q[8][2], w1n = R.q[2][3], w2n = R.q[5][3];
            const float un = R.q[2][1] + dn + pn;                  // = u_n - v*
            float u1 = R.q[5][1] + d1 + p1, u2 = R.q[8][1] + d2 + p2;
            const float ln_old = R.q[9][0];
            const float ln = fmaxf(0.f, ln_old - un * iwn);
            const float dln = ln - ln_old;
            u1 = fmaf(w1n, dln, u1);
            u2 = fmaf(w2n, dln, u2);
            float l1 = R.q[9][1] - u1 * iw1;
            float l2 = R.q[9][2] - u2 * iw2;
            const float lim = mu_s * ln, nn = l1 * l1 + l2 * l2;
            if (nn > lim * lim) { const float sc = mu_d * ln * __builtin_amdgcn_rsqf(nn); l1 *= sc; l2 *= sc; }
            const float g0 = dln, g1 = l1 - R.q[9][1], g2 = l2 - R.q[9][2];
            sweep_add_rows(R, sp, 1.f, g0, g1, g2);
            if (any_split) {                                       // the base part once more, n - 1 times, on the private copy; remembered for the merge
              const float f = split ? nsm1 : 0.f;
              const f2 s0 = splat2(f * g0), s1 = splat2(f * g1), s2 = splat2(f * g2);
              const f2 a01 = fma2(lo2(R.q[0]), s0, fma2(lo2(R.q[3]), s1, lo2(R.q[6]) * s2));
              const f2 a23 = fma2(hi2(R.q[0]), s0, fma2(hi2(R.q[3]), s1, hi2(R.q[6]) * s2));
              const f2 a45 = fma2(lo2(R.q[1]), s0, fma2(lo2(R.q[4]), s1, lo2(R.q[7]) * s2));
              sp.z01 += a01; sp.z23 += a23; sp.z45 += a45;
              ds01 += a01; ds23 += a23; ds45 += a45;
            }
            CRQ(k, 9) = (lf4){ln, l1, l2, SIG && nn > lim * lim ? 1.f : 0.f};
          }
        }
        // the legs' base changes meet (split rows: without the n - 1 extra shares); the own leg's part is already final
        if (split_w) { sp.z01 -= ds01; sp.z23 -= ds23; sp.z45 -= ds45; }
        st.z01[0] += quad_sum(sp.z01[0] - st.z01[0]); st.z01[1] += quad_sum(sp.z01[1] - st.z01[1]);
        st.z23[0] += quad_sum(sp.z23[0] - st.z23[0]); st.z23[1] += quad_sum(sp.z23[1] - st.z23[1]);
        st.z45[0] += quad_sum(sp.z45[0] - st.z45[0]); st.z45[1] += quad_sum(sp.z45[1] - st.z45[1]);
        st.y01 = sp.y01; st.y2 = sp.y2;
      }
      PROF(32);
      // limit rows in joint order: the rate without the row's own impulse is projected on [lower, upper]
#pragma unroll
      for (int lgi = 0; lgi < 4; lgi++) {
        if (LAw & (1u << lgi)) {
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            const int j = 3 * lgi + jj;
            const lf4 r0 = JRQ(j, 0), r1 = JRQ(j, 1), r2 = JRQ(j, 2), r3 = JRQ(j, 3);
            const bool rowon = (lact >> lgi) & 1u;
            f2 acc = lo2(r0) * st.z01;
            acc = fma2(hi2(r0), st.z23, acc);
            acc = fma2(lo2(r1), st.z45, acc);
            const f2 pu = hi2(r1) * st.y01;
            float u = r2[1] + acc[0] + acc[1];
            const float pj = fmaf(r2[0], st.y2, pu[0] + pu[1]);
            u += quad_bcast(pj, lgi);
            const float lold = quad_bcast(lamj[jj], lgi);
            const float u0 = u - r3[0] * lold;
            const float ut = fminf(fmaxf(u0, r2[2]), r2[3]);
            const float ln = rowon ? (ut - u0) * r3[1] : 0.f;
            const float dl = rowon ? ln - lold : 0.f;
            const f2 sd = splat2(dl);
            st.z01 = fma2(lo2(r0), sd, st.z01); st.z23 = fma2(hi2(r0), sd, st.z23); st.z45 = fma2(lo2(r1), sd, st.z45);
            if (leg == lgi) {
              st.y01 = fma2(hi2(r1), sd, st.y01); st.y2 = fmaf(r2[0], dl, st.y2);
              lamj[jj] = ln;
            }
          }
        }
      }
      PROF(33);
      LDS_PHASE();          // the next sweep re-reads the impulses the leg-0 lanes stored in this one
      if (SIG && B.contact_signature != nullptr && sub < GO1_SIG_MAX_SUBSTEPS) {
        // tests only: the ACTIVE SET after every sweep — which contacts press (lambda_n > 0), which were projected on the cone in this
        // sweep, which limit rows carry an impulse —, weighted by the sweep.  Same contact list + same active sets = the same
        // smooth map in oracle and kernel; anything else is a discrete flip (tests/test_gpu_parity.py Attribution): a projection
        // that flips in an intermediate sweep sends the unconverged iterate down another path even when the final sets coincide.
        unsigned jm = 0;
#pragma unroll
        for (int jj = 0; jj < 3; jj++) if (lamj[jj] != 0.f) jm |= 1u << (3 * leg + jj);
        jm = quad_or(jm);
        if (leg == 0) {
          uint32_t ah = jm * 0x27D4EB2Fu;
#pragma unroll 1
          for (int k = 0; k < K; k++) {
            const lf4 q9 = CRQ(k, 9);
            if (q9[0] > 0.f) ah += (uint32_t)(k + 1) * 0x85EBCA6Bu;
            if (q9[3] != 0.f) ah += (uint32_t)(k + 1) * 0xC2B2AE35u;
          }
          sig_active += ah * (uint32_t)(2 * it + 1);
        }
      }
    }
    {
      float nf_acc = 0.f;
      nf_acc = nonfinite_acc(nf_acc, st.z01[0]); nf_acc = nonfinite_acc(nf_acc, st.z01[1]); nf_acc = nonfinite_acc(nf_acc, st.z23[0]);
      nf_acc = nonfinite_acc(nf_acc, st.z23[1]); nf_acc = nonfinite_acc(nf_acc, st.z45[0]); nf_acc = nonfinite_acc(nf_acc, st.z45[1]);
      nf_acc = nonfinite_acc(nf_acc, st.y01[0]); nf_acc = nonfinite_acc(nf_acc, st.y01[1]); nf_acc = nonfinite_acc(nf_acc, st.y2);
#pragma unroll
      for (int i = 0; i < 3; i++) nf_acc = nonfinite_acc(nf_acc, lamj[i]);
      if (nf_acc != nf_acc) fault |= 1u << GO1_FAULT_LAMBDA;
    }
    if (SIG && B.contact_signature != nullptr && sub < GO1_SIG_MAX_SUBSTEPS && leg == 0) AT(B.contact_signature, sub * GO1_SIG_WORDS + 3, e) += sig_active;
  }
#endif

  LDS_PHASE();
  PROF(5);
  // ---- apply all impulses with one propagation ---------------------------------------------------------
  SV pAq[3];
#pragma unroll
  for (int j = 0; j < 3; j++) pAq[j] = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
  V3 fbody[4];                                                 // world impulse per own body (hip, thigh, calf, foot)
#pragma unroll
  for (int i = 0; i < 4; i++) fbody[i] = v3(0.f, 0.f, 0.f);
  auto take = [&](int k, V3& x) {                              // world impulse of solver contact k and its point
    const lf4 q9 = CRQ(k, 9), q10 = CRQ(k, 10), q11 = CRQ(k, 11);
    const V3 n = v3(q11[0], q11[1], q11[2]);
    V3 t1, t2;
    contact_frame(n, t1, t2, fault);
    x = v3(q10[0], q10[1], q10[2]);
    return q9[0] * n + q9[1] * t1 + q9[2] * t2;
  };
  SV contrib = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));       // wrench of the impulses acting directly on the base
  V3 ftrunk = v3(0.f, 0.f, 0.f);                                //   and the impulse booked on the trunk
#pragma unroll
  for (int i = 0; i < IT_N; i++) {
    if (!WALLS && (i == IT_FOOTW || i == IT_CALFW || i == IT_THIGHW || i == IT_TRW)) continue;
    if (slot[i] >= 0) {
      V3 x;
      const V3 f = take(slot[i], x);
      const SV ff = sv(cross(x, f), f);
      if (i >= IT_TR0) { contrib = contrib - ff; ftrunk = ftrunk + f; }
      else {
        const int depth = i <= IT_CALF2 ? 2 : i <= IT_THIGH2 ? 1 : 0;     // foot, calf: joint 2; thigh: 1; hip: 0
        const int bi = i <= IT_FOOTW ? 3 : i <= IT_CALF2 ? 2 : i <= IT_THIGH2 ? 1 : 0;      // own body index: hip 0, thigh 1, calf 2, foot 3
#pragma unroll
        for (int j = 0; j < 3; j++)
          if (j == depth) pAq[j] = pAq[j] - ff;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q == bi) fbody[q] = fbody[q] + f;
      }
    }
  }
  if (nown > 0) {
#pragma unroll
    for (int i = 0; i <= MAXSB; i++) {
      if (sslot[i] >= 0) {
        V3 x;
        const V3 f = ssign[i] * take(sslot[i], x);              // body B receives the opposite impulse
        const SV ff = sv(cross(x, f), f);
        if (sdepth[i] == 0) { pAq[0] = pAq[0] - ff; fbody[0] = fbody[0] + f; }        // booked on the hip / the thigh / the calf (penalised
        else if (sdepth[i] == 1) { pAq[1] = pAq[1] - ff; fbody[1] = fbody[1] + f; }   //  bodies: the collision reward sees it,
        else { pAq[2] = pAq[2] - ff; fbody[2] = fbody[2] + f; }                        //  corl_rewards.py:49-52)
        const bool trunk_pair = (((int)CRQ(sslot[i], 8)[3]) & 0x78) == 8;              // self, no B record: the trunk is body B
        if (trunk_pair) { contrib = contrib + ff; ftrunk = ftrunk - f; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int b = 1 + 4 * leg + i;
    LDS(L_LAM + 3 * b) = fbody[i].x; LDS(L_LAM + 3 * b + 1) = fbody[i].y; LDS(L_LAM + 3 * b + 2) = fbody[i].z;      // listed: impulse, else 0
  }
  PROF(27);
  float du[3];
#pragma unroll
  for (int j = 2; j >= 0; j--) {
    float u = lamj[j] - dot(F.S[j], pAq[j]);
    du[j] = u;
    SV pa = pAq[j] + (u * F.Dinv[j]) * F.U[j];
    if (j > 0) pAq[j - 1] = pAq[j - 1] + pa; else contrib = contrib + pa;
  }
  ftrunk = v3(quad_sum(ftrunk.x), quad_sum(ftrunk.y), quad_sum(ftrunk.z));
  if (leg == 0) { LDS(L_LAM) = ftrunk.x; LDS(L_LAM + 1) = ftrunk.y; LDS(L_LAM + 2) = ftrunk.z; }
  SV dv0 = -sym6_mul(I0inv, quad_sum(contrib));
  s.w = w_free + dv0.a;
  s.v = v_free + dv0.l;
  {   // Cfg.asset.max_angular_velocity / max_linear_velocity: magnitude caps on the base twist
    const float wn2 = dot(s.w, s.w), vn2 = dot(s.v, s.v);
    if (wn2 > cfg.max_angular_velocity * cfg.max_angular_velocity) s.w = (cfg.max_angular_velocity * rsqrtf(wn2)) * s.w;
    if (vn2 > cfg.max_linear_velocity * cfg.max_linear_velocity) s.v = (cfg.max_linear_velocity * rsqrtf(vn2)) * s.v;
  }
  PROF(28);
  {
    SV a = dv0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      float dqd = F.Dinv[j] * (du[j] - dot(F.U[j], a));
      a = a + dqd * F.S[j];
      float qd = F.qdf[j] + dqd;
      // What the limit rows leave is the solver's residual; it is NOT clamped away.  Only a failure of the rows far outside
      // the admissible band is cut (and counted): GO1_LIMIT_SAFETY x the rate limit, GO1_LIMIT_SLACK beyond a stop.
      const float vl = GO1_LIMIT_SAFETY * GO1_JOINT_VEL_LIMIT[ji];
      if (!(fabsf(qd) <= vl)) { fault |= 1u << GO1_FAULT_LIMIT_SAFETY; qd = fminf(fmaxf(qd, -vl), vl); }
      float q = L.q[j] + h * qd;
      const float lo = GO1_JOINT_LOWER[ji] - GO1_LIMIT_SLACK, hi = GO1_JOINT_UPPER[ji] + GO1_LIMIT_SLACK;
      if (q < lo) { q = lo; qd = fmaxf(qd, 0.f); fault |= 1u << GO1_FAULT_LIMIT_SAFETY; }
      if (q > hi) { q = hi; qd = fminf(qd, 0.f); fault |= 1u << GO1_FAULT_LIMIT_SAFETY; }
      L.q[j] = q;
      L.qd[j] = qd;
    }
  }
  PROF(29);
  // base pose
  s.pos = s.pos + h * s.v;
  float wn = norm(s.w);
  if (wn > 1e-12f) {
    float half = 0.5f * wn * h, sn, cs;
    sincosf(half, &sn, &cs);
    sn /= wn;
    float dx = s.w.x * sn, dy = s.w.y * sn, dz = s.w.z * sn, dw = cs;
    float nx = dw * s.qx + dx * s.qw + dy * s.qz - dz * s.qy;
    float ny = dw * s.qy - dx * s.qz + dy * s.qw + dz * s.qx;
    float nz = dw * s.qz + dx * s.qy - dy * s.qx + dz * s.qw;
    float nw_ = dw * s.qw - dx * s.qx - dy * s.qy - dz * s.qz;
    float inv = rsqrtf(nx * nx + ny * ny + nz * nz + nw_ * nw_);
    s.qx = nx * inv; s.qy = ny * inv; s.qz = nz * inv; s.qw = nw_ * inv;
  }
  {
    float a = 0.f;
    a = nonfinite_acc(a, s.pos.x); a = nonfinite_acc(a, s.pos.y); a = nonfinite_acc(a, s.pos.z);
    a = nonfinite_acc(a, s.qx); a = nonfinite_acc(a, s.qy); a = nonfinite_acc(a, s.qz); a = nonfinite_acc(a, s.qw);
    a = nonfinite_acc(a, s.w.x); a = nonfinite_acc(a, s.w.y); a = nonfinite_acc(a, s.w.z);
    a = nonfinite_acc(a, s.v.x); a = nonfinite_acc(a, s.v.y); a = nonfinite_acc(a, s.v.z);
#pragma unroll
    for (int j = 0; j < 3; j++) { a = nonfinite_acc(a, L.q[j]); a = nonfinite_acc(a, L.qd[j]); }
    if (a != a) fault |= 1u << GO1_FAULT_STATE_OUT;
  }
  LDS_PHASE();          // the next substep's warm start reads what this one wrote (per-body impulses)
  PROF(6);
}

// own foot position / velocity at the current state (reference legged_robot.py:112-115)
DEV void foot_state(const Base& s, const Leg& L, int leg, BufRef B, int e, int N) {
  M3 Rpar = quat_to_mat(s.qx, s.qy, s.qz, s.qw);
  V3 ppar = v3(0.f, 0.f, 0.f);
  SV vb = sv(s.w, s.v);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int ji = 3 * leg + j;
    V3 p = ppar + mul(Rpar, model_v3(GO1_JOINT_ORIGIN, ji));
    V3 ax = (j == 0) ? Rpar.c0 : Rpar.c1;
    float sn, cs;
    sincosf(L.q[j], &sn, &cs);
    Rpar = (j == 0) ? rot_x(Rpar, sn, cs) : rot_y(Rpar, sn, cs);
    vb = vb + L.qd[j] * sv(ax, cross(p, ax));
    ppar = p;
  }
  V3 x = ppar + mul(Rpar, model_v3(GO1_FOOT_OFFSET, leg));
  V3 vp = vb.l + cross(vb.a, x);
  AT(B.foot_positions, 3 * leg + 0, e) = s.pos.x + x.x;
  AT(B.foot_positions, 3 * leg + 1, e) = s.pos.y + x.y;
  AT(B.foot_positions, 3 * leg + 2, e) = s.pos.z + x.z;
  AT(B.foot_velocities, 3 * leg + 0, e) = vp.x;
  AT(B.foot_velocities, 3 * leg + 1, e) = vp.y;
  AT(B.foot_velocities, 3 * leg + 2, e) = vp.z;
}

// ---- state <-> HBM ----------------------------------------------------------------------------------------
DEV void load_state(BufRef B, int leg, int e, int N, Base& s, Leg& L) {
  s.pos = v3(AT(B.root_states, 0, e), AT(B.root_states, 1, e), AT(B.root_states, 2, e));
  s.qx = AT(B.root_states, 3, e); s.qy = AT(B.root_states, 4, e); s.qz = AT(B.root_states, 5, e); s.qw = AT(B.root_states, 6, e);
  s.v = v3(AT(B.root_states, 7, e), AT(B.root_states, 8, e), AT(B.root_states, 9, e));
  s.w = v3(AT(B.root_states, 10, e), AT(B.root_states, 11, e), AT(B.root_states, 12, e));
#pragma unroll
  for (int j = 0; j < 3; j++) { L.q[j] = AT(B.dof_pos, 3 * leg + j, e); L.qd[j] = AT(B.dof_vel, 3 * leg + j, e); L.tau[j] = 0.f; }
  s.mass0 = GO1_BODY_MASS[0] + B.payloads[e];
  s.com0 = v3(AT(B.com_displacements, 0, e), AT(B.com_displacements, 1, e), AT(B.com_displacements, 2, e));
  s.mu = B.friction_coeffs[e];
  s.rest = B.restitutions[e];
}
DEV void store_state(BufRef B, int leg, int e, int N, const Base& s, const Leg& L) {
  if (leg == 0) {
    AT(B.root_states, 0, e) = s.pos.x; AT(B.root_states, 1, e) = s.pos.y; AT(B.root_states, 2, e) = s.pos.z;
    AT(B.root_states, 3, e) = s.qx; AT(B.root_states, 4, e) = s.qy; AT(B.root_states, 5, e) = s.qz; AT(B.root_states, 6, e) = s.qw;
    AT(B.root_states, 7, e) = s.v.x; AT(B.root_states, 8, e) = s.v.y; AT(B.root_states, 9, e) = s.v.z;
    AT(B.root_states, 10, e) = s.w.x; AT(B.root_states, 11, e) = s.w.y; AT(B.root_states, 12, e) = s.w.z;
  }
#pragma unroll
  for (int j = 0; j < 3; j++) { AT(B.dof_pos, 3 * leg + j, e) = L.q[j]; AT(B.dof_vel, 3 * leg + j, e) = L.qd[j]; }
}
// per-body impulses <-> contact force buffer; each lane moves its leg's 4 bodies, lane 0 also the trunk
// Two halves, so that the loads join the prologue's single batch: the select on `zero` must not pull them into a branch (the
// compiler then issues them one at a time with a full wait each — 15 exposed HBM round trips, measured), hence the pin.
struct LambdaIn { float f[5][3]; };
DEV void load_lambda_issue(BufRef B, int lane, int e, int N, LambdaIn& in) {
  const int leg = lane & 3;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int b = (i == 4) ? 0 : 1 + 4 * leg + i;       // (the trunk's row is loaded by all four lanes: same address, one request)
#pragma unroll
    for (int c = 0; c < 3; c++) in.f[i][c] = AT(B.contact_forces, 3 * b + c, e);
  }
}
DEV void load_lambda_commit(CfgRef cfg, float* lds, int lane, LambdaIn& in, bool zero) {
  const int leg = lane & 3, el = lane >> 2;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int b = (i == 4) ? 0 : 1 + 4 * leg + i;       // world force -> world impulse
#pragma unroll
    for (int c = 0; c < 3; c++) {
      VALUE_BARRIER(in.f[i][c]);
      if (i == 4 && leg != 0) continue;
      LDS(L_LAM + 3 * b + c) = zero ? 0.f : in.f[i][c] * cfg.sim_dt;
    }
  }
}
DEV void store_forces(CfgRef cfg, BufRef B, const float* lds, int lane, int e, int N) {
  const int leg = lane & 3, el = lane >> 2;
  const float inv = 1.f / cfg.sim_dt;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    if (i == 4 && leg != 0) continue;
    const int b = (i == 4) ? 0 : 1 + 4 * leg + i;
#pragma unroll
    for (int c = 0; c < 3; c++) AT(B.contact_forces, 3 * b + c, e) = LDS(L_LAM + 3 * b + c) * inv;
  }
}
