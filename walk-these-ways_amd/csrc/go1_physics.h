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
#define MAXC 8                       // solver contacts per env (oracle: GO1_MAX_CONTACTS)
#define NRC (3 * MAXC)               // contact rows: contact k -> rows 3k (normal), 3k+1, 3k+2 (tangents)
#define NRJ 12                       // joint-limit rows: joint j -> row NRC + j
#define NRT 36                       // rows: NRC + NRJ
#define NCC (NRT / 4)                // Delassus columns per lane (9): lane `leg` owns the columns c = leg + 4 cc
#define GO1_LIMIT_RECOVERY_RATE 10.0f   // rad/s: a joint found beyond a stop is brought back at a bounded rate
#define GO1_LIMIT_SAFETY 2.0f           // x velocity limit: beyond this the limit rows have failed (cut + fault count)
#define GO1_LIMIT_SLACK 0.2f            // rad beyond a stop: same

// ---- LDS map (floats), index = field * EPW + env_local ---------------------------------------------
enum {
  L_LAM = 0,                 // 17 x world impulse (x, y, z) per reported body: solver output / warm start
  L_CX = 51,                 // MAXC x 3 contact points (rel. base origin)
  L_CN = L_CX + 3 * MAXC,    // MAXC x 3 contact normals
  L_RB = L_CN + 3 * MAXC,    // NRC   contact rows: b = J v_free
  L_RP = L_RB + NRC,         // NRC   normal rows: target velocity v*; tangent rows: W[t][n]
  L_RI = L_RP + NRC,         // NRC   1 / W[r][r]
  L_LS = L_RI + NRC,         // NRC   contact impulses: start values in, solution out
  L_I0 = L_LS + NRC,         // 21    inverse of the base's articulated inertia (upper triangle): for the wavefronts that build W
  L_KL = L_I0 + 21,          // 4     contacts in the solver list, mask of the legs whose limit rows are in the solve, 1: the build
                             //       also adds W lambda_start to the rows' right-hand sides (see delassus_rows), != 0: a helper
                             //       wavefront met a degenerate contact normal (emit_contacts_helper)
  L_W = L_RB,                // zero-filled at kernel start from here to L_END
  L_END = L_KL + 4
};
#define LDS(f) lds[(f) * EPW + el]
// Packed records, one per (row, environment), read with 16-byte LDS loads:
//   row functional RF: [0..5] g_r (the row's unit impulse propagated to the base), [6..8] u_j(r) along the row's leg,
//                      [9..11] u_j(r) / D_j, [12] leg of the row (4 = trunk)
//   limit row      JR: [0] b = free joint rate, [1] lower, [2] upper rate bound, [3] W[r][r], [4] 1 / W[r][r] (0: not in the solve)
//   Delassus row   W : the row's NRT entries; lane `leg` owns columns c = leg + 4 cc: slots cc = 0..7 are its 8 contiguous
//                      floats at [8 leg], slot cc = 8 sits at [32 + leg]
#define RF_ST 16
#define RF(r) (rfl + ((r) * EPW + el) * RF_ST)
#define JR_ST 8
#define JR(j) (jrl + ((j) * EPW + el) * JR_ST)
#define WST 36
#define WROW(r) (ldsw + ((r) * EPW + el) * WST)
#define WSH4(r, hf) (reinterpret_cast<lf4*>(WROW(r) + 8 * leg)[hf])       // the lane's column slots 4 hf .. 4 hf + 3 of row r
#define WSH8(r) (WROW(r)[32 + leg])                                        // the lane's column slot 8 of row r
#define LDSW_SIZE (NRT * EPW * WST)
#define MAXSB 2                      // leg-leg self-contacts per env (oracle: GO1_MAX_SELF_LEG_PAIRS)
#define LDSX_SIZE (NRT * EPW * RF_ST + NRJ * EPW * JR_ST + 3 * MAXSB * EPW * RF_ST)       // RF, JR, then the B sides of leg-leg rows
#define SB(i) (rfl + NRT * EPW * RF_ST + NRJ * EPW * JR_ST + ((i) * EPW + el) * RF_ST)
#define GO1_SELF_LEG_RADIUS ((float)GO1_FOOT_RADIUS)
typedef __attribute__((ext_vector_type(4))) float lf4;

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
DEV void actuator_tiles(const float* a, float* io, int lane, int t0, int ts) {
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
  for (int t = t0; t < 12; t += ts) {
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
// stash for the duration of the step: loaded by the prologue in one batch (torque_stash_load), written back once
// (torque_stash_store).  Same arithmetic in the same order as compute_torques(): results are bit-identical.
#define ACT_MAX_DEC 4
enum { AH_E1 = 0, AH_E2 = 3, AH_V1 = 6, AH_V2 = 9, AH_MS = 12, AH_MO = 15, AH_TGT = 18, AH_END = AH_TGT + 3 * ACT_MAX_DEC };
#define ACTH(k) acth[(k) * WAVE + lane]

// act[jj]: the clipped action.  All loads of the lag buffer come before its stores (a substep reads the slot the NEXT
// substep overwrites).
DEV void torque_stash_load(CfgRef cfg, BufRef B, float* acth, int lane, int leg, int e, int N, int head, const float act[3]) {
  const int nl = cfg.lag_timesteps + 1;
  float a[3], tg[ACT_MAX_DEC][3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    a[jj] = act[jj] * cfg.action_scale;
    if (jj == 0) a[jj] *= cfg.hip_scale_reduction;
  }
#pragma unroll
  for (int sb = 0; sb < ACT_MAX_DEC; sb++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      float t = a[jj];
      if (cfg.use_lag && sb < cfg.decimation && sb + 1 < nl) t = B.lag_buffer[((size_t)((head + sb + 1) % nl) * 12 + j) * N + e];
      tg[sb][jj] = t + cfg.default_dof_pos[j];
    }
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const int j = 3 * leg + jj;
    ACTH(AH_E1 + jj) = AT(B.joint_pos_err_last, j, e);
    ACTH(AH_E2 + jj) = AT(B.joint_pos_err_last_last, j, e);
    ACTH(AH_V1 + jj) = AT(B.joint_vel_last, j, e);
    ACTH(AH_V2 + jj) = AT(B.joint_vel_last_last, j, e);
    ACTH(AH_MS + jj) = AT(B.motor_strengths, j, e);
    ACTH(AH_MO + jj) = AT(B.motor_offsets, j, e);
  }
#pragma unroll
  for (int sb = 0; sb < ACT_MAX_DEC; sb++)
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      ACTH(AH_TGT + 3 * sb + jj) = tg[sb][jj];
      if (cfg.use_lag && sb < cfg.decimation) B.lag_buffer[((size_t)((head + sb) % nl) * 12 + 3 * leg + jj) * N + e] = a[jj];
    }
}
// input rows of substep `sub` into io[A_IN]; the history advances
DEV void torque_publish(const Leg& L, float* acth, float* io, int lane, int sub) {
  float in[3][6];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) {
    const float err = L.q[jj] - ACTH(AH_TGT + 3 * sub + jj) + ACTH(AH_MO + jj);
    const float e1 = ACTH(AH_E1 + jj), e2 = ACTH(AH_E2 + jj), v1 = ACTH(AH_V1 + jj), v2 = ACTH(AH_V2 + jj);
    in[jj][0] = err; in[jj][1] = e1; in[jj][2] = e2; in[jj][3] = L.qd[jj]; in[jj][4] = v1; in[jj][5] = v2;
    ACTH(AH_E2 + jj) = e1; ACTH(AH_E1 + jj) = err; ACTH(AH_V2 + jj) = v1; ACTH(AH_V1 + jj) = L.qd[jj];
  }
  actuator_publish(io, lane, in);
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

struct Cand { float phi, x, y, z, un, nx, ny, nz; };     // deepest contact candidate of one reported body
DEV void cand_init(Cand& c) { c.phi = 1e30f; c.x = c.y = c.z = c.un = 0.f; c.nx = c.ny = 0.f; c.nz = 1.f; }

// terrain height and unit normal at world (x, y): plane, or bilinear interpolation of the int16 height field
// (same sample convention as _get_heights, reference legged_robot.py:1793-1806; oracle terrain_sample())
DEV void terrain_sample(CfgRef cfg, const int16_t* __restrict__ hs, float x, float y, float& h, V3& n) {
  if (cfg.terrain_type == 0 || hs == nullptr) { h = 0.f; n = v3(0.f, 0.f, 1.f); return; }
  float fx = (x + cfg.hf_border) / cfg.hf_hscale, fy = (y + cfg.hf_border) / cfg.hf_hscale;
  fx = fminf(fmaxf(fx, 0.f), (float)cfg.hf_rows - 1.000001f);
  fy = fminf(fmaxf(fy, 0.f), (float)cfg.hf_cols - 1.000001f);
  const int ix = (int)fx, iy = (int)fy;
  const float ax = fx - ix, ay = fy - iy;
  const int16_t* p = hs + (size_t)ix * cfg.hf_cols + iy;
  const float h00 = p[0] * cfg.hf_vscale, h01 = p[1] * cfg.hf_vscale, h10 = p[cfg.hf_cols] * cfg.hf_vscale, h11 = p[cfg.hf_cols + 1] * cfg.hf_vscale;
  h = h00 * (1.f - ax) * (1.f - ay) + h10 * ax * (1.f - ay) + h01 * (1.f - ax) * ay + h11 * ax * ay;
  const float dhdx = ((h10 - h00) * (1.f - ay) + (h11 - h01) * ay) / cfg.hf_hscale;
  const float dhdy = ((h01 - h00) * (1.f - ax) + (h11 - h10) * ax) / cfg.hf_hscale;
  const float inv = rsqrtf(dhdx * dhdx + dhdy * dhdy + 1.f);
  n = v3(-dhdx * inv, -dhdy * inv, inv);
}

// x: candidate point relative to the base origin (world axes); bpos: world position of the base origin
DEV void cand_try(CfgRef cfg, const int16_t* __restrict__ hs, Cand& c, V3 x, V3 bpos, float radius, SV vb) {
  float h;
  V3 n;
  terrain_sample(cfg, hs, bpos.x + x.x, bpos.y + x.y, h, n);
  float phi = (bpos.z + x.z) - radius - h;
  if (phi < c.phi) {
    V3 xs = x - radius * n;               // contact point on the shape surface
    V3 vp = vb.l + cross(vb.a, xs);
    c.phi = phi; c.x = xs.x; c.y = xs.y; c.z = xs.z; c.un = dot(n, vp);
    c.nx = n.x; c.ny = n.y; c.nz = n.z;
  }
}
DEV void cand_min_dpp(Cand& c, int lane) {      // quad-wide deepest candidate (ties: lower leg index, as the serial scan)
#pragma unroll
  for (int step = 0; step < 2; step++) {
    Cand o;
    if (step == 0) { o.phi = dpp_xor1(c.phi); o.x = dpp_xor1(c.x); o.y = dpp_xor1(c.y); o.z = dpp_xor1(c.z); o.un = dpp_xor1(c.un);
                     o.nx = dpp_xor1(c.nx); o.ny = dpp_xor1(c.ny); o.nz = dpp_xor1(c.nz); }
    else           { o.phi = dpp_xor2(c.phi); o.x = dpp_xor2(c.x); o.y = dpp_xor2(c.y); o.z = dpp_xor2(c.z); o.un = dpp_xor2(c.un);
                     o.nx = dpp_xor2(c.nx); o.ny = dpp_xor2(c.ny); o.nz = dpp_xor2(c.nz); }
    const int bit = step == 0 ? 1 : 2;
    const bool other_is_lower = ((lane & bit) != 0);
    bool take = (o.phi < c.phi) || (o.phi == c.phi && other_is_lower);
    if (take) c = o;
  }
}
// contact frame: normal n, t1 = x-axis projected on the tangent plane, t2 = n x t1 (oracle detect_contacts())
// A normal along the x axis leaves no projection: fall back to the y axis (terrain normals have n.z > 0, so this only
// guards body-body normals and corrupted input) and report it.
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

// velocity change of the lane's own leg body at `depth` for the current impulse-propagation state
DEV SV leg_response(const SV S[3], const SV U[3], const float Dinv[3], SV a0, int depth, bool on_path, int path_depth, const float pu[3]) {
  SV a = a0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (j <= depth) {
      float uu = (on_path && j <= path_depth) ? pu[j] : 0.f;
      float qdd = Dinv[j] * (uu - dot(U[j], a));
      a = a + qdd * S[j];
    }
  }
  return a;
}

// ---- Delassus matrix W = J M^-1 J^T into LDS ------------------------------------------------------------------------
// Which rows / columns exist: wave-uniform bounds (scalar branches) from the per-environment contact count and limit-row legs.
// ---- contact emission on the helper wavefronts ---------------------------------------------------------------------------
// Publishing a listed contact — frame, target velocity, b = J v_free, start impulse and the three row functionals (a unit
// impulse propagated through the ABA factors of the contact's leg) — is ~230 instructions, and the master wavefront walks
// its up to nine items per lane one after the other.  With several wavefronts per workgroup the master only writes
// PACKETS into the (at that time unused) matrix block — the leg's ABA factors, the environment's free twist, one record per
// listed terrain contact — and 128 helper lanes take one contact each (contact k of environment el: lane 4 el + (k & 3) of
// helper wavefront 1 + (k >> 2)).  Self-contacts and limit rows stay with the master, which emits them meanwhile.
enum { PK_LEG = 0, PK_LEG_ST = 44, PK_ENV = PK_LEG + WAVE * PK_LEG_ST, PK_ENV_ST = 12, PK_ITEM = PK_ENV + EPW * PK_ENV_ST, PK_ITEM_ST = 12,
       PK_END = PK_ITEM + MAXC * EPW * PK_ITEM_ST };

DEV void emit_contact(CfgRef cfg, float* lds, float* rfl, int el, int k, const Cand& c, int depth, int leg, int body, float share,
                      const SV (&S)[3], const SV (&U)[3], const float (&Dinv)[3], const float (&qd_free)[3], V3 w_free, V3 v_free,
                      float e_c, bool use_warm, float h, uint32_t& fault) {
  const V3 n = v3(c.nx, c.ny, c.nz);
  V3 t1, t2;
  contact_frame(n, t1, t2, fault);
  const V3 x = v3(c.x, c.y, c.z);
  LDS(L_CX + 3 * k) = c.x; LDS(L_CX + 3 * k + 1) = c.y; LDS(L_CX + 3 * k + 2) = c.z;
  LDS(L_CN + 3 * k) = c.nx; LDS(L_CN + 3 * k + 1) = c.ny; LDS(L_CN + 3 * k + 2) = c.nz;
  float vs = fminf(-c.phi / h, cfg.max_depenetration_velocity);
  if (c.un < -cfg.bounce_threshold_velocity && -e_c * c.un > vs) vs = -e_c * c.un;
  LDS(L_RP + 3 * k) = vs;
  SV vb = sv(w_free, v_free);
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (j <= depth) vb = vb + qd_free[j] * S[j];
  const V3 vp = vb.l + cross(vb.a, x);
  LDS(L_RB + 3 * k) = dot(n, vp); LDS(L_RB + 3 * k + 1) = dot(t1, vp); LDS(L_RB + 3 * k + 2) = dot(t2, vp);
  const V3 wl = v3(LDS(L_LAM + 3 * body), LDS(L_LAM + 3 * body + 1), LDS(L_LAM + 3 * body + 2));
  const float sh = use_warm ? share : 0.f;
  LDS(L_LS + 3 * k) = sh * dot(wl, n); LDS(L_LS + 3 * k + 1) = sh * dot(wl, t1); LDS(L_LS + 3 * k + 2) = sh * dot(wl, t2);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const V3 d = r == 0 ? n : r == 1 ? t1 : t2;
    SV pA = -sv(cross(x, d), d);
    float uj[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 2; j >= 0; j--) {
      if (j <= depth) {
        const float u = -dot(S[j], pA);
        uj[j] = u;
        pA = pA + (u * Dinv[j]) * U[j];
      }
    }
    lf4* rf = reinterpret_cast<lf4*>(RF(3 * k + r));
    rf[0] = (lf4){pA.a.x, pA.a.y, pA.a.z, pA.l.x};
    rf[1] = (lf4){pA.l.y, pA.l.z, uj[0], uj[1]};
    rf[2] = (lf4){uj[2], uj[0] * Dinv[0], uj[1] * Dinv[1], uj[2] * Dinv[2]};
    rf[3] = (lf4){depth < 0 ? 4.f : (float)leg, -1.f, 0.f, 0.f};
  }
}

// helper wavefront hw (0, 1: contacts 0..3 / 4..7 of every environment; 2: idle): one listed terrain contact per lane
DEV void emit_contacts_helper(CfgRef cfg, float* lds, const float* ldsw, float* rfl, int lane, int hw, float h) {
  if (hw >= (MAXC + 3) / 4) return;
  const int el = lane >> 2, k = 4 * hw + (lane & 3);
  const lf4* ep = reinterpret_cast<const lf4*>(ldsw + PK_ENV + el * PK_ENV_ST);
  const lf4 e0 = ep[0], e1 = ep[1], e2 = ep[2];
  const int K = (int)e2[0], nF = (int)e2[1], nS = (int)e2[2];
  if (k >= K || (k >= nF && k < nF + nS)) return;           // not listed / a self-contact (the master's)
  const lf4* ip = reinterpret_cast<const lf4*>(ldsw + PK_ITEM + (k * EPW + el) * PK_ITEM_ST);
  const lf4 i0 = ip[0], i1 = ip[1], i2 = ip[2];
  Cand c;
  c.phi = i0[0]; c.x = i0[1]; c.y = i0[2]; c.z = i0[3]; c.un = i1[0]; c.nx = i1[1]; c.ny = i1[2]; c.nz = i1[3];
  const int depth = (int)i2[0], leg = (int)i2[1], body = (int)i2[2];
  const lf4* lp = reinterpret_cast<const lf4*>(ldsw + PK_LEG + (4 * el + (leg & 3)) * PK_LEG_ST);
  float f[44];
#pragma unroll
  for (int q = 0; q < 11; q++) { const lf4 v = lp[q]; f[4 * q] = v[0]; f[4 * q + 1] = v[1]; f[4 * q + 2] = v[2]; f[4 * q + 3] = v[3]; }
  SV S[3], U[3];
  float Dinv[3], qd_free[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    S[j] = sv(v3(f[6 * j], f[6 * j + 1], f[6 * j + 2]), v3(f[6 * j + 3], f[6 * j + 4], f[6 * j + 5]));
    U[j] = sv(v3(f[18 + 6 * j], f[19 + 6 * j], f[20 + 6 * j]), v3(f[21 + 6 * j], f[22 + 6 * j], f[23 + 6 * j]));
    Dinv[j] = f[36 + j];
    qd_free[j] = f[39 + j];
  }
  uint32_t fl = 0;
  emit_contact(cfg, lds, rfl, el, k, c, depth, leg, body, i2[3], S, U, Dinv, qd_free, v3(e0[0], e0[1], e0[2]), v3(e0[3], e1[0], e1[1]),
               e1[2], e1[3] != 0.f, h, fl);
  if (fl) LDS(L_KL + 3) = 1.f;
}

struct SolveMasks {
  int Kw;                  // wave-uniform max K
  unsigned LAw;            // wave-uniform: legs with limit rows in some environment
  unsigned ccw;            // wave-uniform: column slots cc with an active column somewhere
  bool colact[NCC];        // column c = leg + 4 cc is a row of THIS environment's solve
};
DEV void solver_masks(int K, unsigned lact, bool legact, int leg, SolveMasks& m) {
  m.Kw = 0;
#pragma unroll
  for (int kk = 1; kk <= MAXC; kk++) m.Kw = (__ballot(K >= kk) != 0ull) ? kk : m.Kw;
  unsigned long long bl = __ballot(legact);
  bl |= bl >> 32; bl |= bl >> 16; bl |= bl >> 8; bl |= bl >> 4;
  m.LAw = (unsigned)(bl & 0xFull);
  m.ccw = 0;
#pragma unroll
  for (int cc = 0; cc < NCC; cc++) {
    const int c = leg + 4 * cc;
    m.colact[cc] = c < NRC ? (c < 3 * K) : (c < NRC + NRJ && ((lact >> ((c - NRC) / 3)) & 1u));
    if (__ballot(m.colact[cc]) != 0ull) m.ccw |= 1u << cc;
  }
}
// the lane's columns: Y_c = I0^-1 g_c, u_j(c) / D_j and the leg of the column
DEV void lane_columns(const float* rfl, const Sym6& I0inv, int leg, int el, unsigned ccw, SV Y[NCC], float ud[NCC][3], float lg[NCC]) {
#pragma unroll
  for (int cc = 0; cc < NCC; cc++) {
    if (ccw & (1u << cc)) {
      int c = leg + 4 * cc;
      c = c < NRC + NRJ ? c : NRC + NRJ - 1;     // lanes 2, 3 have no last column: any finite stand-in (its impulse stays 0)
      const lf4* rf = reinterpret_cast<const lf4*>(RF(c));
      const lf4 r0 = rf[0], r1 = rf[1], r2 = rf[2];
      Y[cc] = sym6_mul(I0inv, sv(v3(r0[0], r0[1], r0[2]), v3(r0[3], r1[0], r1[1])));
      ud[cc][0] = r2[1]; ud[cc][1] = r2[2]; ud[cc][2] = r2[3];
      lg[cc] = rf[3][0];
    } else {                 // a slot no environment of the wavefront uses: zeros, so that the row loop can run branch-free
      Y[cc] = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
      ud[cc][0] = ud[cc][1] = ud[cc][2] = 0.f;
      lg[cc] = -1.f;
    }
  }
}
// Rows wv, wv + nw, ... of the rows that exist somewhere in the wavefront (every wavefront of the workgroup takes its part;
// within a wavefront lane `leg` writes the columns c = leg + 4 cc it owns in the sweep).  Inputs come from LDS only: the row
// functionals, the base's inverse inertia, the per-environment row counts.
// Warm start in the build: the sweep starts from the row velocities u = b + W lambda_start; the wavefront that builds row r
// has the whole row in its lanes, so it adds the row's W lambda_start to the stored b (one FMA per entry and a quad sum)
// instead of the master wavefront streaming the matrix a second time before the first sweep.  Not when some environment
// of the wavefront has leg-leg self-contacts: their rows are corrected after the build (flag in L_KL + 2).
DEV void delassus_rows(float* lds, float* ldsw, float* rfl, int lane, int wv, int nw PROF_PARAM) {
  float* const jrl = rfl + NRT * EPW * RF_ST;
  const int leg = lane & 3, el = lane >> 2;
  Sym6 I0inv;
#pragma unroll
  for (int i = 0; i < 21; i++) I0inv.m[i] = LDS(L_I0 + i);
  const int K = (int)LDS(L_KL);
  const unsigned lact = (unsigned)LDS(L_KL + 1);
  SolveMasks m;
  solver_masks(K, lact, ((lact >> leg) & 1u) != 0u, leg, m);
  SV Y[NCC];
  float ud[NCC][3], lg[NCC];
  lane_columns(rfl, I0inv, leg, el, m.ccw, Y, ud, lg);
  const bool warm_in_build = __ballot(LDS(L_KL + 2) != 0.f) != 0ull;
  float lamc[NCC];
#pragma unroll
  for (int cc = 0; cc < NCC; cc++) {
    const int c = leg + 4 * cc;
    lamc[cc] = (warm_in_build && c < 3 * K && c < NRC) ? LDS(L_LS + (c < NRC ? c : 0)) : 0.f;      // (limit rows start from 0)
  }
  PROF(3);
  // The loop body is branch-free apart from the wave-uniform skips; the next row's record is fetched while this row's
  // entries are computed.
  auto row_here = [&](int rr) { return (rr % nw) == wv && (rr < NRC ? (rr < 3 * m.Kw) : (((m.LAw >> ((rr - NRC) / 3)) & 1u) != 0u)); };
  int r = 0;
  while (r < NRC + NRJ && !row_here(r)) r++;
  lf4 n0, n1, n2, n3;
  if (r < NRC + NRJ) { const lf4* rf = reinterpret_cast<const lf4*>(RF(r)); n0 = rf[0]; n1 = rf[1]; n2 = rf[2]; n3 = rf[3]; }
#pragma unroll 1
  while (r < NRC + NRJ) {
    const lf4 r0 = n0, r1 = n1, r2 = n2;
    const float lr = n3[0];
    int rn = r + 1;
    while (rn < NRC + NRJ && !row_here(rn)) rn++;
    if (rn < NRC + NRJ) { const lf4* rf = reinterpret_cast<const lf4*>(RF(rn)); n0 = rf[0]; n1 = rf[1]; n2 = rf[2]; n3 = rf[3]; }
    const SV g = sv(v3(r0[0], r0[1], r0[2]), v3(r0[3], r1[0], r1[1]));
    const float u0 = r1[2], u1 = r1[3], u2 = r2[0];
    float wv_[NCC];
#pragma unroll
    for (int cc = 0; cc < NCC; cc++) {            // branch-free: one basic block of ~130 independent-enough instructions per row
      const float same = fmaf(u0, ud[cc][0], fmaf(u1, ud[cc][1], u2 * ud[cc][2]));
      wv_[cc] = dot(g, Y[cc]) + (lr == lg[cc] ? same : 0.f);
    }
    WSH4(r, 0) = (lf4){wv_[0], wv_[1], wv_[2], wv_[3]};
    WSH4(r, 1) = (lf4){wv_[4], wv_[5], wv_[6], wv_[7]};
    if (m.ccw & 0x100u) WSH8(r) = wv_[8];
    if (warm_in_build) {
      float uw = 0.f;
#pragma unroll
      for (int cc = 0; cc < NCC; cc++) uw = fmaf(wv_[cc], lamc[cc], uw);      // (inactive slots: entry and impulse are 0)
      uw = quad_sum(uw);
      const bool row_on = r < NRC ? (r < 3 * K) : (((lact >> ((r - NRC) / 3)) & 1u) != 0u);
      if (leg == 0 && row_on) {
        if (r < NRC) LDS(L_RB + (r < NRC ? r : 0)) += uw;
        else JR(r < NRC ? 0 : r - NRC)[0] += uw;
      }
    }
    r = rn;
  }
}

// acth != nullptr: the torques of this substep are being evaluated by the helper wavefronts (torque_publish was called, the
// workgroup barrier behind it passed): they are picked up right before ABA pass 2.
DEV void physics_substep(CfgRef cfg, const int16_t* __restrict__ hs, float* lds, float* ldsw, float* rfl, int lane, int nw, Base& s, Leg& L, V3 grav,
                         bool use_warm, float h, uint32_t& fault, const float* acth PROF_PARAM) {
  float* const jrl = rfl + NRT * EPW * RF_ST;
  const int leg = lane & 3, el = lane >> 2;
  const M3 R0 = quat_to_mat(s.qx, s.qy, s.qz, s.qw);
  const SV v0 = sv(s.w, s.v);
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
  // Contact candidates: every collision shape contributes up to TWO points — its candidate points are split into the two
  // ends of the shape's long axis, the deeper end's deepest point is the first contact, the other end's deepest the second
  // (oracle detect_contacts()).  Trunk box (long axis x): each lane has one corner of either end, quad-wide minimum.
  Cand cb[2];
#pragma unroll
  for (int mm = 0; mm < 2; mm++) {
    cand_init(cb[mm]);
    const int m = 2 * leg + mm;            // m & 1 = mm: the end
    V3 l = v3((m & 1 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[0], (m & 2 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[1], (m & 4 ? 1.f : -1.f) * GO1_TRUNK_BOX_HALF[2]);
    cand_try(cfg, hs, cb[mm], mul(R0, l), s.pos, 0.f, v0);
    cand_min_dpp(cb[mm], lane);
  }

  // ---- own leg: kinematics, contact candidates, ABA passes 1+2 ---------------------------------------
  SV S[3], U[3];
  float Dinv[3], uu[3];
  SV cj[3];
  Cand ch[2], ct[2], ck[2], cf;      // hip, thigh, calf: one candidate per end; foot
  V3 pknee, pfoot;                   // own lower leg for the self-collision test: knee, foot centre (rel. base origin)
  SV vleg2;                          //   and the calf body's twist before the step
  const float cd0 = cfg.contact_distance;
  {
    M3 R[3];
    V3 p[3];
    SV v[3], pA[3];
    Sym6 IA[3];
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
      S[j] = sv(ax, cross(p[j], ax));
      SV vj = L.qd[j] * S[j];
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
    cand_init(cf);
    {
      V3 hc = model_v3(GO1_HIP_CAPSULE_CENTER, leg);
#pragma unroll
      for (int m = 0; m < 2; m++) {
        V3 l = v3(hc.x, hc.y + (m ? 1.f : -1.f) * (float)GO1_HIP_CAPSULE_HALF, hc.z);
        cand_try(cfg, hs, ch[m], p[0] + mul(R[0], l), s.pos, (float)GO1_HIP_CAPSULE_RADIUS, v[0]);
      }
#ifndef GO1_ABLATE_CAND
#pragma unroll
      for (int en = 0; en < 2; en++) {       // thigh / calf boxes: long axis z -> ends by the sign of z (corner bit 2)
#pragma unroll 1
        for (int m = 4 * en; m < 4 * en + 4; m++) {
          V3 l = v3(GO1_THIGH_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[0],
                    GO1_THIGH_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[1],
                    GO1_THIGH_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * GO1_THIGH_BOX_HALF[2]);
          cand_try(cfg, hs, ct[en], p[1] + mul(R[1], l), s.pos, 0.f, v[1]);
        }
#pragma unroll 1
        for (int m = 4 * en; m < 4 * en + 4; m++) {
          V3 l = v3(GO1_CALF_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[0],
                    GO1_CALF_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[1],
                    GO1_CALF_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * GO1_CALF_BOX_HALF[2]);
          cand_try(cfg, hs, ck[en], p[2] + mul(R[2], l), s.pos, 0.f, v[2]);
        }
      }
#endif
      cand_try(cfg, hs, cf, p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg)), s.pos, (float)GO1_FOOT_RADIUS, v[2]);
      pknee = p[2]; pfoot = p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg)); vleg2 = v[2];
    }
    if (acth) {
      BLOCK_SYNC(nw);                       // the helpers' partial sums are in io[A_OUT]
      torque_collect(cfg, L, acth, ldsw, lane, leg, fault);
    }
    // ABA pass 2: calf -> thigh -> hip, then quad-sum into the base
    SV pa_hip;
#pragma unroll
    for (int j = 2; j >= 0; j--) {
      U[j] = sym6_mul(IA[j], S[j]);
      float D = dot(S[j], U[j]);
      if (!(D > 1e-9f)) { fault |= 1u << GO1_FAULT_JOINT_D; D = 1e-9f; }
      Dinv[j] = 1.f / D;
      uu[j] = L.tau[j] - dot(S[j], pA[j]);
      sym6_rank1_sub(IA[j], U[j], Dinv[j]);
      SV pa = pA[j] + sym6_mul(IA[j], cj[j]) + (uu[j] * Dinv[j]) * U[j];
      if (j > 0) { sym6_add(IA[j - 1], IA[j]); pA[j - 1] = pA[j - 1] + pa; }
      else pa_hip = pa;
    }
#pragma unroll
    for (int i = 0; i < 21; i++) IA0.m[i] += quad_sum(IA[0].m[i]);
    pA0 = pA0 + quad_sum(pa_hip);
  }

  PROF(2);
  // ---- ABA pass 3 ------------------------------------------------------------------------------------
  float min_pivot;
  const Sym6 I0inv = sym6_inverse(IA0, min_pivot);
  if (!(min_pivot > 1e-9f)) fault |= 1u << GO1_FAULT_BASE_PIVOT;
  SV a0 = -sym6_mul(I0inv, pA0);
  V3 w_free = s.w + h * a0.a;
  V3 v_free = s.v + h * (a0.l + cross(s.w, s.v));
  float qd_free[3];
  {
    SV a = a0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      SV ap = a + cj[j];
      float qdd = Dinv[j] * (uu[j] - dot(U[j], ap));
      a = ap + qdd * S[j];
      qd_free[j] = L.qd[j] + h * qdd;
    }
  }

  // ---- self-collision (asset self_collisions = 0: enabled): the lower legs (knee -> foot centre, radius of the foot sphere)
  // against each other and against the trunk's capsule.  Every lane publishes its lower leg (segment, body twist before the
  // step, free twist), reads the other three, and evaluates the pairs it is part of in the canonical (lower leg first)
  // order, so both lanes of a pair hold bit-identical contact data. -------------------------------------------------------
  Cand sc[4];                      // contact with partner leg (leg + 1 + i) & 3 for i < 3; [3]: with the trunk (un: A - B normal velocity)
  V3 sfree[4];                     // relative free velocity (A - B) at the contact point
  bool sact[4] = {false, false, false, false};
  unsigned smask = 0;              // environment-wide: bit pid of every active pair (pairs 0..5 leg-leg, 6..9 trunk-leg)
  SV sv_pre_own, sv_free_own;      // own lower leg: twist before the step / free twist (about the base origin, world axes)
  if (cfg.self_collision) {
    SV vfree = sv(w_free, v_free);
#pragma unroll
    for (int j = 0; j < 3; j++) vfree = vfree + qd_free[j] * S[j];
    sv_pre_own = vleg2; sv_free_own = vfree;
    {
      lf4* a = reinterpret_cast<lf4*>(RF(2 * leg));
      a[0] = (lf4){pknee.x, pknee.y, pknee.z, pfoot.x};
      a[1] = (lf4){pfoot.y, pfoot.z, vleg2.a.x, vleg2.a.y};
      a[2] = (lf4){vleg2.a.z, vleg2.l.x, vleg2.l.y, vleg2.l.z};
      lf4* bq = reinterpret_cast<lf4*>(RF(2 * leg + 1));
      bq[0] = (lf4){vfree.a.x, vfree.a.y, vfree.a.z, vfree.l.x};
      bq[1] = (lf4){vfree.l.y, vfree.l.z, 0.f, 0.f};
    }
    LDS_PHASE();
    unsigned mybits = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int j = (leg + 1 + i) & 3;
      const lf4* a = reinterpret_cast<const lf4*>(RF(2 * j));
      const lf4 a0 = a[0], a1 = a[1];
      const V3 pj = v3(a0[0], a0[1], a0[2]), qj = v3(a0[3], a1[0], a1[1]);
      const bool lower = leg < j;                               // canonical order: lower leg index is body A
      sact[i] = lower ? capsule_contact(pknee, pfoot, GO1_SELF_LEG_RADIUS, pj, qj, GO1_SELF_LEG_RADIUS, cd0, sc[i])
                      : capsule_contact(pj, qj, GO1_SELF_LEG_RADIUS, pknee, pfoot, GO1_SELF_LEG_RADIUS, cd0, sc[i]);
      const int lo_ = lower ? leg : j, hi_ = lower ? j : leg;
      const int pid = lo_ == 0 ? hi_ - 1 : lo_ == 1 ? hi_ + 1 : 5;
      if (sact[i] && lower) mybits |= 1u << pid;
      if (sact[i]) {                                            // relative velocity (A - B) at the contact point
        const lf4 a2 = a[2];
        const SV vpreJ = sv(v3(a1[2], a1[3], a2[0]), v3(a2[1], a2[2], a2[3]));
        const lf4* bq = reinterpret_cast<const lf4*>(RF(2 * j + 1));
        const lf4 b0 = bq[0], b1 = bq[1];
        const SV vfreeJ = sv(v3(b0[0], b0[1], b0[2]), v3(b0[3], b1[0], b1[1]));
        const V3 x = v3(sc[i].x, sc[i].y, sc[i].z), n = v3(sc[i].nx, sc[i].ny, sc[i].nz);
        const SV pa = lower ? sv_pre_own : vpreJ, pb = lower ? vpreJ : sv_pre_own;
        const SV fa = lower ? sv_free_own : vfreeJ, fbq = lower ? vfreeJ : sv_free_own;
        sc[i].un = dot(n, (pa.l + cross(pa.a, x)) - (pb.l + cross(pb.a, x)));
        sfree[i] = (fa.l + cross(fa.a, x)) - (fbq.l + cross(fbq.a, x));
      }
    }
    {
      const float ta = (float)(GO1_TRUNK_BOX_HALF[0] - GO1_TRUNK_BOX_HALF[1]);
      sact[3] = capsule_contact(pknee, pfoot, GO1_SELF_LEG_RADIUS, mul(R0, v3(-ta, 0.f, 0.f)), mul(R0, v3(ta, 0.f, 0.f)),
                                (float)GO1_TRUNK_BOX_HALF[1], cd0, sc[3]);
      if (sact[3]) {
        mybits |= 1u << (6 + leg);
        const V3 x = v3(sc[3].x, sc[3].y, sc[3].z), n = v3(sc[3].nx, sc[3].ny, sc[3].nz);
        const SV fb0 = sv(w_free, v_free);
        sc[3].un = dot(n, (sv_pre_own.l + cross(sv_pre_own.a, x)) - (v0.l + cross(v0.a, x)));
        sfree[3] = (sv_free_own.l + cross(sv_free_own.a, x)) - (fb0.l + cross(fb0.a, x));
      }
    }
    mybits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mybits, 0xB1, 0xF, 0xF, false);
    mybits |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mybits, 0x4E, 0xF, 0xF, false);
    smask = mybits;
    LDS_PHASE();                   // the published segments are overwritten by the row functionals below
  }

  // ---- solver contact list (oracle detect_contacts()): feet, trunk (first, second point), calves (first points, second
  // points), thighs, hips; at most MAXC, the rest is dropped and counted -------------------------------------------------
  const float cd = cfg.contact_distance;
  // own items in priority order: 0 foot, 1 calf first, 2 calf second, 3 thigh first, 4 thigh second, 5 hip first, 6 hip second
  const int fk = ck[1].phi < ck[0].phi ? 1 : 0, ft = ct[1].phi < ct[0].phi ? 1 : 0, fh = ch[1].phi < ch[0].phi ? 1 : 0;
  const int fb = cb[1].phi < cb[0].phi ? 1 : 0;
  Cand item[7];
  item[0] = cf;
  item[1] = fk ? ck[1] : ck[0]; item[2] = fk ? ck[0] : ck[1];
  item[3] = ft ? ct[1] : ct[0]; item[4] = ft ? ct[0] : ct[1];
  item[5] = fh ? ch[1] : ch[0]; item[6] = fh ? ch[0] : ch[1];
  const Cand tb0 = fb ? cb[1] : cb[0], tb1 = fb ? cb[0] : cb[1];
  const bool act_b0 = tb0.phi < cd, act_b1 = tb1.phi < cd;
  const unsigned below = (1u << leg) - 1u;
  int slot[7], slot_b0, slot_b1, nF;
  int K;
  {
    int base_ofs = 0;
    unsigned m0 = quad_ballot(item[0].phi < cd, lane);
    slot[0] = (item[0].phi < cd) ? __popc(m0 & below) : -1;
    base_ofs += __popc(m0);
    // self-contacts: pair order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3), then trunk-leg 0..3; at most MAXSB leg-leg pairs
    {
      const unsigned legpairs = smask & 0x3Fu;
      unsigned keep = 0, cnt = 0;
#pragma unroll
      for (int pid = 0; pid < 6; pid++)
        if (legpairs & (1u << pid)) { if (cnt < MAXSB) keep |= 1u << pid; cnt++; }
      if (cnt > MAXSB && leg == 0) fault |= 1u << GO1_FAULT_CONTACT_DROPPED;
      smask = keep | (smask & 0x3C0u);
    }
    nF = base_ofs;
    base_ofs += __popc(smask);
    const int ofs_b0 = base_ofs;
    base_ofs += (act_b0 ? 1 : 0) + (act_b1 ? 1 : 0);
#pragma unroll
    for (int i = 1; i < 7; i++) {
      const bool a = item[i].phi < cd;
      const unsigned m = quad_ballot(a, lane);
      slot[i] = a ? base_ofs + __popc(m & below) : -1;
      base_ofs += __popc(m);
    }
    K = base_ofs;
    if (K > MAXC && leg == 0) fault |= 1u << GO1_FAULT_CONTACT_DROPPED;
    K = K > MAXC ? MAXC : K;
#pragma unroll
    for (int i = 0; i < 7; i++) if (slot[i] >= MAXC) slot[i] = -1;
    // trunk slots (handled by lane 0, known to all)
    slot_b0 = act_b0 && ofs_b0 < MAXC ? ofs_b0 : -1;
    slot_b1 = act_b1 && ofs_b0 + (act_b0 ? 1 : 0) < MAXC ? ofs_b0 + (act_b0 ? 1 : 0) : -1;
  }
  PROF(19);
  const float e_c = 0.5f * (s.rest + cfg.terrain_restitution);
  // warm start: a body's previous impulse is shared equally by its listed points
  const float share_k = (slot[1] >= 0 && slot[2] >= 0) ? 0.5f : 1.f, share_t = (slot[3] >= 0 && slot[4] >= 0) ? 0.5f : 1.f,
              share_h = (slot[5] >= 0 && slot[6] >= 0) ? 0.5f : 1.f, share_b = (slot_b0 >= 0 && slot_b1 >= 0) ? 0.5f : 1.f;

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
      const float vf = qd_free[j];
      if (!(vf > lo + mp && vf < hi - mp && vf > -vl + mv && vf < vl - mv)) legact = true;
    }
  }
  const unsigned lact = quad_ballot(legact, lane);      // legs of this environment whose limit rows are in the solve

  // publish the listed contacts: point, normal, target normal velocity, b = J v_free, start impulse — and the row
  // functionals: each row's unit impulse propagated through the ABA factors of the contact's leg to the base.
  // For a unit impulse along row c the backward ABA pass leaves g_c, the wrench arriving at the base, and the joint
  // residuals u_j(c).  By reciprocity the same vectors are the row functionals, so
  //     W[r][c] = g_r . (I0^-1 g_c)  +  [same leg] sum_j u_j(r) u_j(c) / D_j
  // (legs couple only through the base).  A contact row starts from the spatial force of the unit impulse at the contact
  // point, a joint row from the unit generalised impulse at its joint.
  PROF(20);
  const bool offload = nw > 1;         // terrain contacts on the helper wavefronts (emit_contacts_helper)
  if (offload) {
    lf4* lp = reinterpret_cast<lf4*>(ldsw + PK_LEG + lane * PK_LEG_ST);
    float f[44];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      f[6 * j] = S[j].a.x; f[6 * j + 1] = S[j].a.y; f[6 * j + 2] = S[j].a.z; f[6 * j + 3] = S[j].l.x; f[6 * j + 4] = S[j].l.y; f[6 * j + 5] = S[j].l.z;
      f[18 + 6 * j] = U[j].a.x; f[19 + 6 * j] = U[j].a.y; f[20 + 6 * j] = U[j].a.z; f[21 + 6 * j] = U[j].l.x; f[22 + 6 * j] = U[j].l.y; f[23 + 6 * j] = U[j].l.z;
      f[36 + j] = Dinv[j];
      f[39 + j] = qd_free[j];
    }
    f[42] = f[43] = 0.f;
#pragma unroll
    for (int q = 0; q < 11; q++) lp[q] = (lf4){f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
    if (leg == 0) {
      lf4* ep = reinterpret_cast<lf4*>(ldsw + PK_ENV + el * PK_ENV_ST);
      ep[0] = (lf4){w_free.x, w_free.y, w_free.z, v_free.x};
      ep[1] = (lf4){v_free.y, v_free.z, e_c, use_warm ? 1.f : 0.f};
      ep[2] = (lf4){(float)K, (float)nF, (float)__popc(smask), 0.f};
      LDS(L_KL + 3) = 0.f;
    }
    auto post = [&](int k, const Cand& c, int depth, int body, float share) {
      lf4* ip = reinterpret_cast<lf4*>(ldsw + PK_ITEM + (k * EPW + el) * PK_ITEM_ST);
      ip[0] = (lf4){c.phi, c.x, c.y, c.z};
      ip[1] = (lf4){c.un, c.nx, c.ny, c.nz};
      ip[2] = (lf4){(float)depth, (float)leg, (float)body, share};
    };
    if (slot[0] >= 0) post(slot[0], item[0], 2, 4 + 4 * leg, 1.f);
    if (slot[1] >= 0) post(slot[1], item[1], 2, 3 + 4 * leg, share_k);
    if (slot[2] >= 0) post(slot[2], item[2], 2, 3 + 4 * leg, share_k);
    if (slot[3] >= 0) post(slot[3], item[3], 1, 2 + 4 * leg, share_t);
    if (slot[4] >= 0) post(slot[4], item[4], 1, 2 + 4 * leg, share_t);
    if (slot[5] >= 0) post(slot[5], item[5], 0, 1 + 4 * leg, share_h);
    if (slot[6] >= 0) post(slot[6], item[6], 0, 1 + 4 * leg, share_h);
    if (leg == 0) {
      if (slot_b0 >= 0) post(slot_b0, tb0, -1, 0, share_b);
      if (slot_b1 >= 0) post(slot_b1, tb1, -1, 0, share_b);
    }
    BLOCK_SYNC(nw);                      // the helpers emit while this wavefront goes on with the self-contacts and limit rows
  } else {
    auto emit = [&](int k, const Cand& c, int depth, int body, float share) {      // depth < 0: trunk
      emit_contact(cfg, lds, rfl, el, k, c, depth, leg, body, share, S, U, Dinv, qd_free, w_free, v_free, e_c, use_warm, h, fault);
    };
    // (the trunk's impulse is read by lane 0 only; own bodies by the own lane: no cross-lane hazard on L_LAM here)
    if (slot[0] >= 0) emit(slot[0], item[0], 2, 4 + 4 * leg, 1.f);
    if (slot[1] >= 0) emit(slot[1], item[1], 2, 3 + 4 * leg, share_k);
    if (slot[2] >= 0) emit(slot[2], item[2], 2, 3 + 4 * leg, share_k);
    if (slot[3] >= 0) emit(slot[3], item[3], 1, 2 + 4 * leg, share_t);
    if (slot[4] >= 0) emit(slot[4], item[4], 1, 2 + 4 * leg, share_t);
    if (slot[5] >= 0) emit(slot[5], item[5], 0, 1 + 4 * leg, share_h);
    if (slot[6] >= 0) emit(slot[6], item[6], 0, 1 + 4 * leg, share_h);
    if (leg == 0) {
      if (slot_b0 >= 0) emit(slot_b0, tb0, -1, 0, share_b);
      if (slot_b1 >= 0) emit(slot_b1, tb1, -1, 0, share_b);
    }
  }
  // self-contacts.  Body A (the lower leg of the lower-numbered leg, or the leg of a trunk pair) publishes the contact and
  // its side of the row functionals; for a leg-leg pair body B's lane adds its side (record SB); the trunk as body B has
  // no joints: its side is the unit wrench itself.
  int sslot[4] = {-1, -1, -1, -1};     // solver slot of the pair with partner i / the trunk
  int ssb[3] = {-1, -1, -1};           // index of the leg-leg pair among the listed leg-leg pairs (its SB record)
  if (smask != 0u) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int j = (leg + 1 + i) & 3;
      const bool lower = i == 3 || leg < j;
      const int lo_ = lower ? leg : j, hi_ = lower ? j : leg;
      const int pid = i == 3 ? 6 + leg : (lo_ == 0 ? hi_ - 1 : lo_ == 1 ? hi_ + 1 : 5);
      if (!(smask & (1u << pid))) continue;
      const int k = nF + __popc(smask & ((1u << pid) - 1u));
      if (k >= MAXC) continue;
      sslot[i] = k;
      if (i < 3) ssb[i] = __popc(smask & 0x3Fu & ((1u << pid) - 1u));
    }
  }
  if (smask != 0u) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (sslot[i] < 0) continue;
      const int j = (leg + 1 + i) & 3, k = sslot[i];
      const bool lower = i == 3 || leg < j;
      const Cand& c = sc[i];
      const V3 n = v3(c.nx, c.ny, c.nz), x = v3(c.x, c.y, c.z);
      V3 t1, t2;
      contact_frame(n, t1, t2, fault);
      if (lower) {                           // body A publishes the contact
        LDS(L_CX + 3 * k) = c.x; LDS(L_CX + 3 * k + 1) = c.y; LDS(L_CX + 3 * k + 2) = c.z;
        LDS(L_CN + 3 * k) = c.nx; LDS(L_CN + 3 * k + 1) = c.ny; LDS(L_CN + 3 * k + 2) = c.nz;
        float vs = fminf(-c.phi / h, cfg.max_depenetration_velocity);
        if (c.un < -cfg.bounce_threshold_velocity && -s.rest * c.un > vs) vs = -s.rest * c.un;      // robot-robot: the robot's own material
        LDS(L_RP + 3 * k) = vs;
        LDS(L_RB + 3 * k) = dot(n, sfree[i]); LDS(L_RB + 3 * k + 1) = dot(t1, sfree[i]); LDS(L_RB + 3 * k + 2) = dot(t2, sfree[i]);
        LDS(L_LS + 3 * k) = 0.f; LDS(L_LS + 3 * k + 1) = 0.f; LDS(L_LS + 3 * k + 2) = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const V3 d = r == 0 ? n : r == 1 ? t1 : t2;
        const SV unit = sv(cross(x, d), d);
        SV pA = lower ? -unit : unit;        // +d on body A, -d on body B
        float uj[3];
#pragma unroll
        for (int jq = 2; jq >= 0; jq--) {
          const float u = -dot(S[jq], pA);
          uj[jq] = u;
          pA = pA + (u * Dinv[jq]) * U[jq];
        }
        if (i == 3) pA = pA + unit;          // the trunk's side
        lf4* rf = reinterpret_cast<lf4*>(lower ? RF(3 * k + r) : SB(3 * ssb[i] + r));
        rf[0] = (lf4){pA.a.x, pA.a.y, pA.a.z, pA.l.x};
        rf[1] = (lf4){pA.l.y, pA.l.z, uj[0], uj[1]};
        rf[2] = (lf4){uj[2], uj[0] * Dinv[0], uj[1] * Dinv[1], uj[2] * Dinv[2]};
        rf[3] = (lf4){(float)leg, (lower && i < 3) ? (float)ssb[i] : -1.f, 0.f, 0.f};       // [1]: index of the row's SB record, -1: none
      }
    }
  }
  if (legact) {
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      float* jr = JR(3 * leg + jj);
      jr[0] = qd_free[jj]; jr[1] = jlo[jj]; jr[2] = jhi[jj];
      float uj[3] = {0.f, 0.f, 0.f};
      uj[jj] = 1.f;
      SV pA = Dinv[jj] * U[jj];
#pragma unroll
      for (int j = 1; j >= 0; j--) {
        if (j < jj) {
          const float u = -dot(S[j], pA);
          uj[j] = u;
          pA = pA + (u * Dinv[j]) * U[j];
        }
      }
      lf4* rf = reinterpret_cast<lf4*>(RF(NRC + 3 * leg + jj));
      rf[0] = (lf4){pA.a.x, pA.a.y, pA.a.z, pA.l.x};
      rf[1] = (lf4){pA.l.y, pA.l.z, uj[0], uj[1]};
      rf[2] = (lf4){uj[2], uj[0] * Dinv[0], uj[1] * Dinv[1], uj[2] * Dinv[2]};
      rf[3] = (lf4){(float)leg, -1.f, 0.f, 0.f};
    }
  }
  LDS_PHASE();
  // leg-leg self-contacts: body A's lane folds body B's wrench into the row functional (g = g_A + g_B)
  const bool selfw = __ballot((smask & 0x3Fu) != 0u) != 0ull;      // wave-uniform
  if (selfw) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (sslot[i] >= 0 && leg < ((leg + 1 + i) & 3)) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          lf4* rf = reinterpret_cast<lf4*>(RF(3 * sslot[i] + r));
          const lf4* sb = reinterpret_cast<const lf4*>(SB(3 * ssb[i] + r));
          lf4 a0 = rf[0], a1 = rf[1];
          const lf4 b0 = sb[0], b1 = sb[1];
          a0 = a0 + b0; a1[0] += b1[0]; a1[1] += b1[1];
          rf[0] = a0; rf[1] = a1;
        }
      }
    }
    LDS_PHASE();
  }
  PROF(4);
  // ---- Delassus matrix into LDS: every wavefront of the workgroup builds its share of the rows (delassus_rows) -----------
  if (leg == 0) {
#pragma unroll
    for (int i = 0; i < 21; i++) LDS(L_I0 + i) = I0inv.m[i];
    LDS(L_KL) = (float)K; LDS(L_KL + 1) = (float)lact; LDS(L_KL + 2) = selfw ? 0.f : 1.f;
  }
  BLOCK_SYNC(nw);
  PROF(19);            // (profile builds: the wait at the first barrier is booked on phase 19)
#ifdef GO1_ROWS_HELPERS_ONLY
  if (nw == 1) delassus_rows(lds, ldsw, rfl, lane, 0, 1 PROF_PASS);
#else
  delassus_rows(lds, ldsw, rfl, lane, 0, nw PROF_PASS);
#endif
  BLOCK_SYNC(nw);
  PROF(21);
  if (offload && LDS(L_KL + 3) != 0.f) fault |= 1u << GO1_FAULT_CONTACT_FRAME;
  SolveMasks sm;
  solver_masks(K, lact, legact, leg, sm);
  const int Kw = sm.Kw;
  const unsigned LAw = sm.LAw, ccw = sm.ccw;
  bool colact[NCC];
#pragma unroll
  for (int cc = 0; cc < NCC; cc++) colact[cc] = sm.colact[cc];
  {
    // Rows of leg-leg self-contacts carry TWO leg parts (A: in the row record, B: in its SB record).  The loop above used
    // the total wrench g but only the A parts in the same-leg term; such rows (at most 3 MAXSB per environment) are redone
    // here with all parts, row and mirrored column.  Rare: skipped unless some environment of the wave has such a contact.
    if (selfw) {
      SV Y[NCC];
      float ud[NCC][3], lg[NCC];
      lane_columns(rfl, I0inv, leg, el, ccw, Y, ud, lg);
      int colsb[NCC];
#pragma unroll
      for (int cc = 0; cc < NCC; cc++) {
        int c = leg + 4 * cc;
        c = c < NRC + NRJ ? c : NRC + NRJ - 1;
        colsb[cc] = (ccw & (1u << cc)) ? (int)RF(c)[13] : -1;
      }
      const int nS = __popc(smask);
#pragma unroll 1
      for (int q = 0; q < 3 * MAXC; q++) {                      // candidate rows: the self slots' rows
        const int x = 3 * nF + q;
        const bool inr = q < 3 * nS && x < NRC;
        const int sbx = inr ? (int)RF(x < NRC ? x : 0)[13] : -1;
        if (__ballot(sbx >= 0) == 0ull) { if (__ballot(inr) == 0ull) break; continue; }
        if (sbx >= 0) {
          const lf4* rf = reinterpret_cast<const lf4*>(RF(x));
          const lf4 r0 = rf[0], r1 = rf[1], r2 = rf[2];
          const lf4* sb = reinterpret_cast<const lf4*>(SB(3 * sbx + (x % 3)));
          const lf4 s1 = sb[1], s2 = sb[2];
          const SV g = sv(v3(r0[0], r0[1], r0[2]), v3(r0[3], r1[0], r1[1]));
          const float uA0 = r1[2], uA1 = r1[3], uA2 = r2[0], lA = rf[3][0];
          const float uB0 = s1[2], uB1 = s1[3], uB2 = s2[0], lB = sb[3][0];
#pragma unroll
          for (int cc = 0; cc < NCC; cc++) {
            const int c = leg + 4 * cc;
            if ((ccw & (1u << cc)) && c < NRC + NRJ) {
              float w = dot(g, Y[cc]);
              if (lA == lg[cc]) w += fmaf(uA0, ud[cc][0], fmaf(uA1, ud[cc][1], uA2 * ud[cc][2]));
              if (lB == lg[cc]) w += fmaf(uB0, ud[cc][0], fmaf(uB1, ud[cc][1], uB2 * ud[cc][2]));
              if (colsb[cc] >= 0) {
                const lf4* sc_ = reinterpret_cast<const lf4*>(SB(3 * colsb[cc] + (c % 3)));
                const lf4 c2 = sc_[2];
                const float lgB = sc_[3][0];
                if (lA == lgB) w += fmaf(uA0, c2[1], fmaf(uA1, c2[2], uA2 * c2[3]));
                if (lB == lgB) w += fmaf(uB0, c2[1], fmaf(uB1, c2[2], uB2 * c2[3]));
              }
              if (cc < 8) WROW(x)[8 * leg + cc] = w; else WSH8(x) = w;
              if ((x >> 2) < 8) WROW(c)[8 * (x & 3) + (x >> 2)] = w; else WROW(c)[32 + (x & 3)] = w;      // mirrored entry
            }
          }
        }
      }
    }
  }
  LDS_PHASE();
  PROF(22);
  // row records from the finished matrix: every lane reads back the diagonal entries of its own columns, and the owner
  // of a contact's normal column the two entries that couple the tangent rows to it
#pragma unroll
  for (int cc = 0; cc < NCC; cc++) {
    const int c = leg + 4 * cc;
    if ((ccw & (1u << cc)) && c < NRC + NRJ) {
      const float w = cc < 8 ? WROW(c)[8 * leg + cc] : WSH8(c);
      if (colact[cc] && !(w > 1e-9f)) fault |= 1u << GO1_FAULT_W_DIAG;
      const float iw = colact[cc] ? 1.f / w : 0.f;
      if (c < NRC) LDS(L_RI + c) = iw;
      else { float* jr = JR(c - NRC); jr[3] = w; jr[4] = iw; }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; k++) {
    if (k < Kw && ((3 * k) & 3) == leg) {
      LDS(L_RP + 3 * k + 1) = WROW(3 * k + 1)[8 * leg + ((3 * k) >> 2)];       // (3 k < 24: slots 0..5)
      LDS(L_RP + 3 * k + 2) = WROW(3 * k + 2)[8 * leg + ((3 * k) >> 2)];
    }
  }
  LDS_PHASE();
  PROF(18);
  // ---- projected Gauss-Seidel on the impulses -------------------------------------------------------------
  // The sweep keeps the ROW VELOCITIES u = b + W lambda up to date instead of re-evaluating a row's dot product when its
  // turn comes: lane `leg` holds u for its rows r = leg + 4 cc; a row's turn is then one quad broadcast of its u, the
  // projection, and — off the critical path — 8 independent FMAs per lane that add the impulse change times the lane's
  // share of that row of W (W is symmetric: the share of row c IS the lane's part of column c) to the velocities.
  const float mu = 0.5f * (s.mu + cfg.terrain_friction);       // PhysX default combine mode: average
#ifndef GO1_ABLATE_PGS
  {
    float lam[NRC], lamj[NRJ], uloc[NCC];
    float vst[MAXC], idn[MAXC], id1[MAXC], id2[MAXC], w10[MAXC], w20[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
      const int r0 = 3 * k;
      const bool on = k < K;
      lam[r0] = on ? LDS(L_LS + r0) : 0.f; lam[r0 + 1] = on ? LDS(L_LS + r0 + 1) : 0.f; lam[r0 + 2] = on ? LDS(L_LS + r0 + 2) : 0.f;
      vst[k] = on ? LDS(L_RP + r0) : 0.f;
      idn[k] = on ? LDS(L_RI + r0) : 0.f; id1[k] = on ? LDS(L_RI + r0 + 1) : 0.f; id2[k] = on ? LDS(L_RI + r0 + 2) : 0.f;
      w10[k] = on ? LDS(L_RP + r0 + 1) : 0.f; w20[k] = on ? LDS(L_RP + r0 + 2) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NRJ; j++) lamj[j] = 0.f;
#pragma unroll
    for (int cc = 0; cc < NCC; cc++) {                         // u = b for the lane's rows (rows outside this env's solve: 0)
      const int c = leg + 4 * cc;
      float bb = 0.f;
      if (colact[cc]) bb = c < NRC ? LDS(L_RB + (c < NRC ? c : 0)) : JR(c < NRC + NRJ ? c - NRC : 0)[0];
      uloc[cc] = bb;
    }
    // u += (share of row c) * dl for the lane's rows.  Rows that are not in this environment's solve have impulse 0 and
    // what the build wrote for them is finite (the matrix is zero-filled at kernel start), so their products vanish.
    auto apply_col = [&](int c, float dl) {
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        if (ccw & (0xFu << (4 * hf))) {
          const lf4 w = WSH4(c, hf);
#pragma unroll
          for (int i = 0; i < 4; i++) uloc[4 * hf + i] = fmaf(w[i], dl, uloc[4 * hf + i]);
        }
      }
      if (ccw & 0x100u) uloc[8] = fmaf(WSH8(c), dl, uloc[8]);
    };
    if (selfw)                                                 // (otherwise the build has already added W lambda_start to b)
#pragma unroll
    for (int k = 0; k < MAXC; k++) {                           // warm start: the starting impulses' velocities
      if (k < Kw) {
#pragma unroll
        for (int i = 0; i < 3; i++) apply_col(3 * k + i, lam[3 * k + i]);
      }
    }
    PROF(23);
    // The lane's shares of a row of W do not depend on the iterate: a contact's nine loads (three rows x [4 + 4 + 1]) are
    // issued together, UNCONDITIONALLY, before its projection — one LDS latency per contact, running under the projection
    // arithmetic, instead of nine exposed ones (a load behind a wave-uniform `if (ccw & ..)` sits in its own basic block
    // with its own wait).  Slots of columns that exist nowhere in the wavefront hold finite leftovers (zero fill, earlier
    // substeps) and only ever update row velocities nobody reads.
    struct Share { lf4 a, b; float c; };
    auto fetch = [&](int c) { Share sh; sh.a = WSH4(c, 0); sh.b = WSH4(c, 1); sh.c = WSH8(c); return sh; };
    auto apply_share = [&](const Share& sh, float dl) {
#pragma unroll
      for (int i = 0; i < 4; i++) { uloc[i] = fmaf(sh.a[i], dl, uloc[i]); uloc[4 + i] = fmaf(sh.b[i], dl, uloc[4 + i]); }
      uloc[8] = fmaf(sh.c, dl, uloc[8]);
    };
#pragma unroll 1
    for (int it = 0; it < cfg.solver_iterations; it++) {
#pragma unroll
      for (int k = 0; k < MAXC; k++) {
        if (k < Kw) {
          const int r0 = 3 * k;
          const Share s0 = fetch(r0), s1 = fetch(r0 + 1), s2 = fetch(r0 + 2);
          const float un = quad_bcast(uloc[r0 >> 2], r0);      // row r sits in lane r & 3 at slot r >> 2
          float u1 = quad_bcast(uloc[(r0 + 1) >> 2], r0 + 1);
          float u2 = quad_bcast(uloc[(r0 + 2) >> 2], r0 + 2);
          const float ln_old = lam[r0];
          const float ln = fmaxf(0.f, ln_old - (un - vst[k]) * idn[k]);
          const float dln = ln - ln_old;
          u1 = fmaf(w10[k], dln, u1);                          // the tangential rows see the updated normal impulse
          u2 = fmaf(w20[k], dln, u2);
          float l1 = lam[r0 + 1] - u1 * id1[k];
          float l2 = lam[r0 + 2] - u2 * id2[k];
          const float muk = (k >= nF && k < nF + __popc(smask)) ? s.mu : mu;      // robot-robot: the robot's own material
          const float lim = muk * ln, nn = l1 * l1 + l2 * l2;  // friction cone: |l_t| <= mu l_n
          if (nn > lim * lim) { const float sc = lim * __builtin_amdgcn_rsqf(nn); l1 *= sc; l2 *= sc; }
          const bool on = k < K;                               // lanes of environments with fewer contacts idle here
          const float nl0 = on ? ln : 0.f, nl1 = on ? l1 : 0.f, nl2 = on ? l2 : 0.f;
          apply_share(s0, nl0 - lam[r0]); apply_share(s1, nl1 - lam[r0 + 1]); apply_share(s2, nl2 - lam[r0 + 2]);
          lam[r0] = nl0; lam[r0 + 1] = nl1; lam[r0 + 2] = nl2;
        }
      }
      // limit rows in joint order: the rate without the row's own impulse is projected on [lower, upper]
#pragma unroll
      for (int lgi = 0; lgi < 4; lgi++) {
        if (LAw & (1u << lgi)) {
          Share sj[3];
          lf4 rec[3];
          float idg[3];
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {                     // the leg's three rows: records and shares up front
            const float* jr = JR(3 * lgi + jj);
            rec[jj] = *reinterpret_cast<const lf4*>(jr);       // b, lower, upper, W[r][r]
            idg[jj] = jr[4];                                   // 0: not a row of this environment
            sj[jj] = fetch(NRC + 3 * lgi + jj);
          }
#pragma unroll
          for (int jj = 0; jj < 3; jj++) {
            const int j = 3 * lgi + jj, r = NRC + j;
            const float u = quad_bcast(uloc[r >> 2], r);
            const float u0 = u - rec[jj][3] * lamj[j];
            const float ut = fminf(fmaxf(u0, rec[jj][1]), rec[jj][2]);
            const float ln = (ut - u0) * idg[jj];
            apply_share(sj[jj], ln - lamj[j]);
            lamj[j] = ln;
          }
        }
      }
    }
    {
      float nf_acc = 0.f;
#pragma unroll
      for (int r = 0; r < NRC; r++) nf_acc = nonfinite_acc(nf_acc, lam[r]);
#pragma unroll
      for (int j = 0; j < NRJ; j++) nf_acc = nonfinite_acc(nf_acc, lamj[j]);
      if (nf_acc != nf_acc) fault |= 1u << GO1_FAULT_LAMBDA;
    }
    if (leg == 0) {
#pragma unroll
      for (int k = 0; k < MAXC; k++)
        if (k < K) { LDS(L_LS + 3 * k) = lam[3 * k]; LDS(L_LS + 3 * k + 1) = lam[3 * k + 1]; LDS(L_LS + 3 * k + 2) = lam[3 * k + 2]; }
      if (lact != 0u) {
#pragma unroll
        for (int j = 0; j < NRJ; j++) JR(j)[5] = lamj[j];
      }
    }
  }
#endif

  LDS_PHASE();
  PROF(5);
  // ---- apply all impulses with one propagation ---------------------------------------------------------
  SV pA[3];
#pragma unroll
  for (int j = 0; j < 3; j++) pA[j] = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
  V3 fbody[4];                                                 // world impulse per own body (hip, thigh, calf, foot)
#pragma unroll
  for (int i = 0; i < 4; i++) fbody[i] = v3(0.f, 0.f, 0.f);
  auto take = [&](int k, V3& x) {                              // world impulse of solver contact k and its point
    const V3 n = v3(LDS(L_CN + 3 * k), LDS(L_CN + 3 * k + 1), LDS(L_CN + 3 * k + 2));
    V3 t1, t2;
    contact_frame(n, t1, t2, fault);
    x = v3(LDS(L_CX + 3 * k), LDS(L_CX + 3 * k + 1), LDS(L_CX + 3 * k + 2));
    return LDS(L_LS + 3 * k) * n + LDS(L_LS + 3 * k + 1) * t1 + LDS(L_LS + 3 * k + 2) * t2;
  };
#pragma unroll
  for (int i = 0; i < 7; i++) {
    if (slot[i] >= 0) {
      const int depth = i == 0 ? 2 : i <= 2 ? 2 : i <= 4 ? 1 : 0;       // foot, calf: joint 2; thigh: 1; hip: 0
      const int bi = i == 0 ? 3 : i <= 2 ? 2 : i <= 4 ? 1 : 0;          // own body index: hip 0, thigh 1, calf 2, foot 3
      V3 x;
      const V3 f = take(slot[i], x);
      const SV ff = sv(cross(x, f), f);
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (j == depth) pA[j] = pA[j] - ff;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (q == bi) fbody[q] = fbody[q] + f;
    }
  }
  SV self_base = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));     // trunk-leg self-contacts: wrench on the base
  V3 self_trunk = v3(0.f, 0.f, 0.f);                            //   and the impulse booked on the trunk
  if (smask != 0u) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (sslot[i] >= 0) {
        V3 x;
        V3 f = take(sslot[i], x);
        const bool lower = i == 3 || leg < ((leg + 1 + i) & 3);
        if (!lower) f = -f;                                     // body B receives the opposite impulse
        pA[2] = pA[2] - sv(cross(x, f), f);
        fbody[2] = fbody[2] + f;                                // booked on the calf (a penalised body: corl_rewards.py:49-52)
        if (i == 3) { self_base = self_base + sv(cross(x, f), f); self_trunk = self_trunk - f; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int b = 1 + 4 * leg + i;
    LDS(L_LAM + 3 * b) = fbody[i].x; LDS(L_LAM + 3 * b + 1) = fbody[i].y; LDS(L_LAM + 3 * b + 2) = fbody[i].z;      // listed: impulse, else 0
  }
  float lj[3] = {0.f, 0.f, 0.f};                               // limit impulses of the own joints
  if (legact) {
#pragma unroll
    for (int j = 0; j < 3; j++) lj[j] = JR(3 * leg + j)[5];
  }
  SV contrib = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
  float du[3];
#pragma unroll
  for (int j = 2; j >= 0; j--) {
    float u = lj[j] - dot(S[j], pA[j]);
    du[j] = u;
    SV pa = pA[j] + (u * Dinv[j]) * U[j];
    if (j > 0) pA[j - 1] = pA[j - 1] + pa; else contrib = pa;
  }
  contrib = contrib + self_base;                               // the trunk's share of trunk-leg contacts (-f at x)
  if (cfg.self_collision) self_trunk = v3(quad_sum(self_trunk.x), quad_sum(self_trunk.y), quad_sum(self_trunk.z));
  if (leg == 0) {
    V3 f = self_trunk;
    if (slot_b0 >= 0) { V3 x; const V3 f0 = take(slot_b0, x); contrib = contrib - sv(cross(x, f0), f0); f = f + f0; }
    if (slot_b1 >= 0) { V3 x; const V3 f1 = take(slot_b1, x); contrib = contrib - sv(cross(x, f1), f1); f = f + f1; }
    LDS(L_LAM) = f.x; LDS(L_LAM + 1) = f.y; LDS(L_LAM + 2) = f.z;
  }
  SV dv0 = -sym6_mul(I0inv, quad_sum(contrib));
  s.w = w_free + dv0.a;
  s.v = v_free + dv0.l;
  {   // Cfg.asset.max_angular_velocity / max_linear_velocity: magnitude caps on the base twist
    const float wn2 = dot(s.w, s.w), vn2 = dot(s.v, s.v);
    if (wn2 > cfg.max_angular_velocity * cfg.max_angular_velocity) s.w = (cfg.max_angular_velocity * rsqrtf(wn2)) * s.w;
    if (vn2 > cfg.max_linear_velocity * cfg.max_linear_velocity) s.v = (cfg.max_linear_velocity * rsqrtf(vn2)) * s.v;
  }
  {
    SV a = dv0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      float dqd = Dinv[j] * (du[j] - dot(U[j], a));
      a = a + dqd * S[j];
      float qd = qd_free[j] + dqd;
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
    float nw = dw * s.qw - dx * s.qx - dy * s.qy - dz * s.qz;
    float inv = rsqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
    s.qx = nx * inv; s.qy = ny * inv; s.qz = nz * inv; s.qw = nw * inv;
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
  LDS_PHASE();          // the next substep's warm start reads what this one wrote (per-body impulses, row records)
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
DEV void load_lambda(CfgRef cfg, BufRef B, float* lds, int lane, int e, int N, bool zero) {
  const int leg = lane & 3, el = lane >> 2;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    if (i == 4 && leg != 0) continue;
    const int b = (i == 4) ? 0 : 1 + 4 * leg + i;       // world force -> world impulse
#pragma unroll
    for (int c = 0; c < 3; c++) {          // unconditional load (joins the prologue's batch), then select
      const float f = AT(B.contact_forces, 3 * b + c, e);
      LDS(L_LAM + 3 * b + c) = zero ? 0.f : f * cfg.sim_dt;
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
