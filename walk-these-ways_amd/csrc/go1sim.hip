// go1sim.hip — MI355X (gfx950 / CDNA4) Go1 vectorised step: kernels + the C-ABI of include/go1sim.h.
//
// Mapping: one environment per lane, one 64-lane wavefront per workgroup (4096 envs -> 64 workgroups;
// the step is a latency-bound O(n_dof) recursion, not a bandwidth- or MFMA-bound kernel: DESIGN.md).
// State lives in HBM as SoA [component][env] so every per-lane access of a wavefront is one coalesced
// 256-byte transaction.  The 12-joint chain (joint state, motion subspaces S_j, ABA factors U_j, 1/D_j,
// bias terms) and the contact Delassus matrix are staged in LDS as [field][lane] (bank = lane:
// conflict free), which lets the leg/joint loops stay rolled (small code, no scratch spills) while the
// per-env base quantities (6x6 articulated inertia, its inverse, base twist) stay in VGPRs.
//
// Physics per substep (replaces gym.simulate, reference legged_robot.py:76-80):
//   1. forward kinematics + per-body deepest-point contact detection        (leg loop)
//   2. Featherstone articulated-body algorithm, world-aligned frame at the base origin
//   3. Delassus matrix W = J M^-1 J^T of the <= 6 solver contacts by O(n) impulse propagation
//      through the ABA factors; projected Gauss-Seidel on (normal, 2 tangents) with a Coulomb cone
//   4. one more impulse propagation applies all contact impulses; semi-implicit Euler; joint limits
// followed by the fused tensor maps of post_physics_step (reference legged_robot.py:90-136).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/go1sim.h"
#include "go1_math.h"

#define GO1_CONST static __device__ __constant__ const
#define GO1_REAL float
#include "go1_model_data.h"
#include "go1_actuator_data.h"

#define WAVE 64
#define MAXC 6                       // solver contacts per env (oracle: GO1_MAX_CONTACTS)
#define PI_F 3.14159265358979323846f

// ---- LDS map: float index = field * 64 + lane ------------------------------------------------
enum {
  L_Q = 0, L_QD = 12,
  L_TAU = 24, L_UU = 24, // joint torque, overwritten in place by u_j = tau_j - S_j . pA_j (ABA pass 2)
  L_S = 36, L_U = 108, L_DINV = 180,
  L_LAM = 192,          // 17 x (n, t1, t2) impulses per reported body
  L_CX = 243,           // MAXC x 3 contact points (rel. base origin)
  L_VSTAR = 261,        // MAXC
  L_BODY = 267,         // MAXC reported-body index of the slot (as float)
  L_BV = 273,           // MAXC x 3  b = J v_free
  L_LS = 291,           // MAXC x 3  slot impulses
  L_W = 309,            // (3 MAXC)^2 Delassus matrix ...
  L_C = 309,            //   ... aliased with the velocity-product terms c_j (12 x 6), dead before W is built
  L_CAND = 309 + 72,    //   ... and with the per-body contact candidates (17 x {phi, x, y, z, un}), dead before W
  L_END = 309 + 324
};
static_assert(L_CAND + 17 * 5 <= L_END, "candidate scratch must fit in the W region");
static_assert(L_END * WAVE * 4 <= 163840, "LDS budget (160 KiB per CU)");
#define LDS(f) lds[(f) * WAVE + lane]

// priority order of the solver contact list (oracle: CONTACT_ORDER)
__device__ __constant__ const int CONTACT_ORDER[17] = {4, 8, 12, 16, 0, 3, 7, 11, 15, 2, 6, 10, 14, 1, 5, 9, 13};

enum { P_NOISE = 1, P_RESET = 2, P_DOFPROPS_CB = 3, P_DOFPROPS_RESET = 4, P_CMD_CB = 5, P_CMD_RESET = 6, P_PUSH = 7, P_GRAVITY = 8 };

struct SimConst {            // lives in device memory (one per handle): indexable with scalar loads
  Go1SimConfig cfg;
  Go1SimBuffers buf;
};
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

#define AT(ptr, c, e) ((ptr)[(size_t)(c) * N + (e)])

__device__ __noinline__ float rng_uniform(const Go1SimConfig& cfg, uint32_t env_global, int64_t step, uint32_t purpose, uint32_t idx) {
  uint32_t out[4];
  philox4x32_10(env_global, (uint32_t)step, purpose, idx >> 2, (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32), out);
  uint32_t sel = idx & 3;
  uint32_t v = sel == 0 ? out[0] : sel == 1 ? out[1] : sel == 2 ? out[2] : out[3];
  return u32_to_unit(v);
}

// ================================================================================================
// torque model (reference legged_robot.py:907-946): lag ring, actuator net / PD, strength, clip
// ================================================================================================
DEV float softsign(float x) { return x * __builtin_amdgcn_rcpf(1.f + fabsf(x)); }   // v_rcp_f32: <= 1 ulp

// The 6->32->32->1 actuator network for the three joints of one leg at once: every weight (wave-uniform, fetched
// with scalar loads from constant memory) is used for 3 independent accumulation chains, which hides the
// 4-cycle dependent-FMA latency that a single chain would expose at one wave per SIMD.
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

DEV void compute_torques(const Go1SimConfig& cfg, const Go1SimBuffers& B, float* lds, int lane, int e, int N, int head) {
  const int nl = cfg.lag_timesteps + 1;
  const int h2 = (head + 1) % nl;
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
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
        const float q = LDS(L_Q + j), qd = LDS(L_QD + j);
        float err = q - tgt[jj] + AT(B.motor_offsets, j, e);
        float el = AT(B.joint_pos_err_last, j, e), ell = AT(B.joint_pos_err_last_last, j, e);
        float vl = AT(B.joint_vel_last, j, e), vll = AT(B.joint_vel_last_last, j, e);
        in[jj][0] = err; in[jj][1] = el; in[jj][2] = ell; in[jj][3] = qd; in[jj][4] = vl; in[jj][5] = vll;
        AT(B.joint_pos_err_last_last, j, e) = el;
        AT(B.joint_pos_err_last, j, e) = err;
        AT(B.joint_vel_last_last, j, e) = vl;
        AT(B.joint_vel_last, j, e) = qd;
      }
      actuator_net3(in, tq);
    } else {
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        const int j = 3 * leg + jj;
        const float q = LDS(L_Q + j), qd = LDS(L_QD + j);
        tq[jj] = cfg.kp * AT(B.Kp_factors, j, e) * (tgt[jj] - q + AT(B.motor_offsets, j, e)) - cfg.kd * AT(B.Kd_factors, j, e) * qd;
      }
    }
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
      const int j = 3 * leg + jj;
      float t = tq[jj] * AT(B.motor_strengths, j, e);
      const float lim = cfg.torque_limits[j];
      t = fminf(fmaxf(t, -lim), lim);
      LDS(L_TAU + j) = t;
      AT(B.torques, j, e) = t;
    }
  }
}

// ================================================================================================
// physics substep
// ================================================================================================
struct Base {
  V3 pos;                 // world position of the base origin
  float qx, qy, qz, qw;
  V3 w, v;                // angular velocity, velocity of the base origin (world axes)
  float mass0;            // trunk mass + payload
  V3 com0;                // base com in body axes (= com_displacement, reference legged_robot.py:671)
  float mu, rest;
};

DEV SV lds_sv(const float* lds, int lane, int field) {
  return sv(v3(LDS(field), LDS(field + 1), LDS(field + 2)), v3(LDS(field + 3), LDS(field + 4), LDS(field + 5)));
}
DEV void lds_put_sv(float* lds, int lane, int field, SV s) {
  LDS(field) = s.a.x; LDS(field + 1) = s.a.y; LDS(field + 2) = s.a.z;
  LDS(field + 3) = s.l.x; LDS(field + 4) = s.l.y; LDS(field + 5) = s.l.z;
}
DEV V3 model_v3(const float (*tab)[3], int i) { return v3((float)tab[i][0], (float)tab[i][1], (float)tab[i][2]); }

// record a contact candidate of reported body `rep` (keeps the deepest)
// x: candidate point relative to the base origin (world axes); base_z: world height of the base origin
DEV void candidate(float* lds, int lane, int rep, V3 x, float base_z, float radius, SV vb) {
  float phi = (base_z + x.z) - radius;      // plane terrain: height 0, normal +z
  if (phi < LDS(L_CAND + rep * 5)) {
    V3 xs = v3(x.x, x.y, x.z - radius);     // contact point on the shape surface
    LDS(L_CAND + rep * 5) = phi;
    LDS(L_CAND + rep * 5 + 1) = xs.x;
    LDS(L_CAND + rep * 5 + 2) = xs.y;
    LDS(L_CAND + rep * 5 + 3) = xs.z;
    V3 vp = vb.l + cross(vb.a, xs);
    LDS(L_CAND + rep * 5 + 4) = vp.z;       // normal velocity before the step (restitution test)
  }
}

// body velocity change produced by the current impulse-propagation state, evaluated for the dynamic body
// at depth `depth` of leg `leg` (depth -1: base).  path_leg/path_u describe the loaded chain.
DEV SV body_response(const float* lds, int lane, SV a0, int leg, int depth, int path_leg, int path_depth, const float path_u[3]) {
  SV a = a0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    if (j <= depth) {
      int ji = 3 * leg + j;
      SV U = lds_sv(lds, lane, L_U + 6 * ji), S = lds_sv(lds, lane, L_S + 6 * ji);
      float uu = (leg == path_leg && j <= path_depth) ? path_u[j] : 0.f;
      float qdd = LDS(L_DINV + ji) * (uu - dot(U, a));
      a = a + qdd * S;
    }
  }
  return a;
}

DEV void physics_substep(const Go1SimConfig& cfg, float* lds, int lane, Base& s, V3 grav, bool use_warm, float h) {
  const M3 R0 = quat_to_mat(s.qx, s.qy, s.qz, s.qw);
  const SV v0 = sv(s.w, s.v);
  // ---- base body -------------------------------------------------------------------------------
  Sym6 IA0;
  SV pA0;
  {
    float Il[6], Iw[6];
    float scale = s.mass0 / (float)GO1_BODY_MASS[0];   // recomputeInertia=True: mass-proportional (oracle kinematics())
#pragma unroll
    for (int i = 0; i < 6; i++) Il[i] = (float)GO1_BODY_INERTIA[0][i] * scale;
    rotate_inertia(R0, Il, Iw);
    V3 c = mul(R0, s.com0);
    IA0 = rigid_inertia(s.mass0, c, Iw);
    SV hv = sym6_mul(IA0, v0);
    V3 fg = s.mass0 * grav;
    pA0 = cross_force(v0, hv) - sv(cross(c, fg), fg);
  }
#pragma unroll 1
  for (int b = 0; b < 17; b++) LDS(L_CAND + b * 5) = 1e30f;
  // trunk box corners
#pragma unroll 1
  for (int m = 0; m < 8; m++) {
    V3 l = v3((m & 1 ? 1.f : -1.f) * (float)GO1_TRUNK_BOX_HALF[0], (m & 2 ? 1.f : -1.f) * (float)GO1_TRUNK_BOX_HALF[1],
              (m & 4 ? 1.f : -1.f) * (float)GO1_TRUNK_BOX_HALF[2]);
    candidate(lds, lane, 0, mul(R0, l), s.pos.z, 0.f, v0);
  }

  // ---- legs: kinematics, contacts, ABA passes 1+2 -----------------------------------------------
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
    M3 R[3];
    V3 p[3];
    SV S[3], v[3], c[3], pA[3];
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
      sincosf(LDS(L_Q + ji), &sn, &cs);
      R[j] = (j == 0) ? rot_x(Rpar, sn, cs) : rot_y(Rpar, sn, cs);
      S[j] = sv(ax, cross(p[j], ax));
      float qd = LDS(L_QD + ji);
      SV vj = qd * S[j];
      v[j] = vpar + vj;
      c[j] = cross_motion(v[j], vj);
      float Il[6], Iw[6];
#pragma unroll
      for (int i = 0; i < 6; i++) Il[i] = (float)GO1_BODY_INERTIA[b][i];
      rotate_inertia(R[j], Il, Iw);
      V3 com = p[j] + mul(R[j], model_v3(GO1_BODY_COM, b));
      float m = (float)GO1_BODY_MASS[b];
      IA[j] = rigid_inertia(m, com, Iw);
      SV hv = sym6_mul(IA[j], v[j]);
      V3 fg = m * grav;
      pA[j] = cross_force(v[j], hv) - sv(cross(com, fg), fg);
      lds_put_sv(lds, lane, L_S + 6 * ji, S[j]);
      lds_put_sv(lds, lane, L_C + 6 * ji, c[j]);
      Rpar = R[j]; ppar = p[j]; vpar = v[j];
    }
    // contact candidates of this leg (reported bodies 1+4*leg .. 4+4*leg)
    {
      const int rep = 1 + 4 * leg;
      V3 hc = model_v3(GO1_HIP_CAPSULE_CENTER, leg);
#pragma unroll 1
      for (int m = 0; m < 2; m++) {
        V3 l = v3(hc.x, hc.y + (m ? 1.f : -1.f) * (float)GO1_HIP_CAPSULE_HALF, hc.z);
        candidate(lds, lane, rep, p[0] + mul(R[0], l), s.pos.z, (float)GO1_HIP_CAPSULE_RADIUS, v[0]);
      }
#pragma unroll 1
      for (int m = 0; m < 8; m++) {
        V3 l = v3((float)GO1_THIGH_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * (float)GO1_THIGH_BOX_HALF[0],
                  (float)GO1_THIGH_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * (float)GO1_THIGH_BOX_HALF[1],
                  (float)GO1_THIGH_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * (float)GO1_THIGH_BOX_HALF[2]);
        candidate(lds, lane, rep + 1, p[1] + mul(R[1], l), s.pos.z, 0.f, v[1]);
      }
#pragma unroll 1
      for (int m = 0; m < 8; m++) {
        V3 l = v3((float)GO1_CALF_BOX_CENTER[0] + (m & 1 ? 1.f : -1.f) * (float)GO1_CALF_BOX_HALF[0],
                  (float)GO1_CALF_BOX_CENTER[1] + (m & 2 ? 1.f : -1.f) * (float)GO1_CALF_BOX_HALF[1],
                  (float)GO1_CALF_BOX_CENTER[2] + (m & 4 ? 1.f : -1.f) * (float)GO1_CALF_BOX_HALF[2]);
        candidate(lds, lane, rep + 2, p[2] + mul(R[2], l), s.pos.z, 0.f, v[2]);
      }
      candidate(lds, lane, rep + 3, p[2] + mul(R[2], model_v3(GO1_FOOT_OFFSET, leg)), s.pos.z, (float)GO1_FOOT_RADIUS, v[2]);
    }
    // ABA pass 2: calf -> thigh -> hip -> base
#pragma unroll
    for (int j = 2; j >= 0; j--) {
      const int ji = 3 * leg + j;
      SV U = sym6_mul(IA[j], S[j]);
      float D = dot(S[j], U);
      float Dinv = 1.f / D;
      float u = LDS(L_TAU + ji) - dot(S[j], pA[j]);
      lds_put_sv(lds, lane, L_U + 6 * ji, U);
      LDS(L_DINV + ji) = Dinv;
      LDS(L_UU + ji) = u;
      sym6_rank1_sub(IA[j], U, Dinv);
      SV pa = pA[j] + sym6_mul(IA[j], c[j]) + (u * Dinv) * U;
      if (j > 0) { sym6_add(IA[j - 1], IA[j]); pA[j - 1] = pA[j - 1] + pa; }
      else       { sym6_add(IA0, IA[0]); pA0 = pA0 + pa; }
    }
  }

  // ---- ABA pass 3 --------------------------------------------------------------------------------
  const Sym6 I0inv = sym6_inverse(IA0);
  SV a0 = -sym6_mul(I0inv, pA0);
  // free velocity v_free = v + h * (classical accelerations)
  V3 w_free = s.w + h * a0.a;
  V3 v_free = s.v + h * (a0.l + cross(s.w, s.v));
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
    SV a = a0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      SV ap = a + lds_sv(lds, lane, L_C + 6 * ji);
      float qdd = LDS(L_DINV + ji) * (LDS(L_UU + ji) - dot(lds_sv(lds, lane, L_U + 6 * ji), ap));
      a = ap + qdd * lds_sv(lds, lane, L_S + 6 * ji);
      LDS(L_QD + ji) += h * qdd;
    }
  }

  // ---- solver contact list (priority order, capped) ---------------------------------------------
  int K = 0;
#pragma unroll 1
  for (int o = 0; o < 17; o++) {
    const int b = CONTACT_ORDER[o];
    float phi = LDS(L_CAND + b * 5);
    bool act = (phi < cfg.contact_distance) && (K < MAXC);
    if (act) {
      // note: slot arrays live outside the aliased W region
      float x = LDS(L_CAND + b * 5 + 1), y = LDS(L_CAND + b * 5 + 2), z = LDS(L_CAND + b * 5 + 3), un = LDS(L_CAND + b * 5 + 4);
      LDS(L_CX + 3 * K) = x; LDS(L_CX + 3 * K + 1) = y; LDS(L_CX + 3 * K + 2) = z;
      LDS(L_BODY + K) = (float)b;
      float vs = fminf(-phi / h, cfg.max_depenetration_velocity);
      float e_c = 0.5f * (s.rest + cfg.terrain_restitution);
      if (un < -cfg.bounce_threshold_velocity && -e_c * un > vs) vs = -e_c * un;
      LDS(L_VSTAR + K) = vs;
      K++;
    }
  }
  // impulses of bodies that are not in the list are dropped; listed ones start from the warm value or zero
  {
    float keep[MAXC][3];
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
      int b = (k < K) ? (int)LDS(L_BODY + (k < K ? k : 0)) : 0;
#pragma unroll
      for (int r = 0; r < 3; r++) keep[k][r] = (k < K && use_warm) ? LDS(L_LAM + 3 * b + r) : 0.f;
    }
#pragma unroll 1
    for (int i = 0; i < 51; i++) LDS(L_LAM + i) = 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; k++)
#pragma unroll
      for (int r = 0; r < 3; r++) LDS(L_LS + 3 * k + r) = keep[k][r];
  }

  // ---- Delassus matrix by impulse propagation through the ABA factors ---------------------------
  const int NR = 3 * MAXC;
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    const int b = (int)LDS(L_BODY + k);
    const int leg = (b == 0) ? -1 : (b - 1) / 4;
    const int depth = (b == 0) ? -1 : (((b - 1) % 4) > 2 ? 2 : ((b - 1) % 4));   // foot rides on the calf body
    const V3 x = v3(LDS(L_CX + 3 * k), LDS(L_CX + 3 * k + 1), LDS(L_CX + 3 * k + 2));
    // b = J v_free for this contact
    {
      SV vb = sv(w_free, v_free);
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (j <= depth) vb = vb + LDS(L_QD + 3 * leg + j) * lds_sv(lds, lane, L_S + 6 * (3 * leg + j));
      V3 vp = vb.l + cross(vb.a, x);
      LDS(L_BV + 3 * k) = vp.z; LDS(L_BV + 3 * k + 1) = vp.x; LDS(L_BV + 3 * k + 2) = vp.y;   // (n, t1, t2) = (z, x, y)
    }
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
      V3 d = r == 0 ? v3(0.f, 0.f, 1.f) : r == 1 ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f);
      SV pA = -sv(cross(x, d), d);
      float pu[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 2; j >= 0; j--) {
        if (j <= depth) {
          int ji = 3 * leg + j;
          float u = -dot(lds_sv(lds, lane, L_S + 6 * ji), pA);
          pu[j] = u;
          pA = pA + (u * LDS(L_DINV + ji)) * lds_sv(lds, lane, L_U + 6 * ji);
        }
      }
      SV a0c = -sym6_mul(I0inv, pA);
#pragma unroll 1
      for (int k2 = 0; k2 < K; k2++) {
        const int b2 = (int)LDS(L_BODY + k2);
        const int leg2 = (b2 == 0) ? -1 : (b2 - 1) / 4;
        const int depth2 = (b2 == 0) ? -1 : (((b2 - 1) % 4) > 2 ? 2 : ((b2 - 1) % 4));
        SV ab = body_response(lds, lane, a0c, leg2, depth2, leg, depth, pu);
        V3 x2 = v3(LDS(L_CX + 3 * k2), LDS(L_CX + 3 * k2 + 1), LDS(L_CX + 3 * k2 + 2));
        V3 vp = ab.l + cross(ab.a, x2);
        LDS(L_W + (3 * k2 + 0) * NR + 3 * k + r) = vp.z;
        LDS(L_W + (3 * k2 + 1) * NR + 3 * k + r) = vp.x;
        LDS(L_W + (3 * k2 + 2) * NR + 3 * k + r) = vp.y;
      }
    }
  }

  // ---- projected Gauss-Seidel on the impulses -----------------------------------------------------
  const float mu = 0.5f * (s.mu + cfg.terrain_friction);       // PhysX default combine mode: average
#pragma unroll 1
  for (int it = 0; it < cfg.solver_iterations; it++) {
#pragma unroll 1
    for (int k = 0; k < K; k++) {
      const int r0 = 3 * k;
      float un = LDS(L_BV + r0);
#pragma unroll 1
      for (int c = 0; c < 3 * K; c++) un = fmaf(LDS(L_W + r0 * NR + c), LDS(L_LS + c), un);
      float ln_old = LDS(L_LS + r0);
      float ln = fmaxf(0.f, ln_old - (un - LDS(L_VSTAR + k)) / LDS(L_W + r0 * NR + r0));
      LDS(L_LS + r0) = ln;
      float u1 = LDS(L_BV + r0 + 1), u2 = LDS(L_BV + r0 + 2);
#pragma unroll 1
      for (int c = 0; c < 3 * K; c++) {
        float l = LDS(L_LS + c);
        u1 = fmaf(LDS(L_W + (r0 + 1) * NR + c), l, u1);
        u2 = fmaf(LDS(L_W + (r0 + 2) * NR + c), l, u2);
      }
      float l1 = LDS(L_LS + r0 + 1) - u1 / LDS(L_W + (r0 + 1) * NR + r0 + 1);
      float l2 = LDS(L_LS + r0 + 2) - u2 / LDS(L_W + (r0 + 2) * NR + r0 + 2);
      float lim = mu * ln, nrm = sqrtf(l1 * l1 + l2 * l2);
      if (nrm > lim) { float sc = (nrm > 0.f) ? lim / nrm : 0.f; l1 *= sc; l2 *= sc; }
      LDS(L_LS + r0 + 1) = l1;
      LDS(L_LS + r0 + 2) = l2;
    }
  }

  // ---- apply all contact impulses with one propagation ------------------------------------------
  SV p0 = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    const int b = (int)LDS(L_BODY + k);
    LDS(L_LAM + 3 * b) = LDS(L_LS + 3 * k);
    LDS(L_LAM + 3 * b + 1) = LDS(L_LS + 3 * k + 1);
    LDS(L_LAM + 3 * b + 2) = LDS(L_LS + 3 * k + 2);
    if (b == 0) {
      V3 x = v3(LDS(L_CX + 3 * k), LDS(L_CX + 3 * k + 1), LDS(L_CX + 3 * k + 2));
      V3 f = v3(LDS(L_LS + 3 * k + 1), LDS(L_LS + 3 * k + 2), LDS(L_LS + 3 * k));
      p0 = p0 - sv(cross(x, f), f);
    }
  }
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
    SV pA[3];
#pragma unroll
    for (int j = 0; j < 3; j++) pA[j] = sv(v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f));
#pragma unroll 1
    for (int k = 0; k < K; k++) {
      const int b = (int)LDS(L_BODY + k);
      if (b == 0 || (b - 1) / 4 != leg) continue;
      int depth = (b - 1) % 4;
      depth = depth > 2 ? 2 : depth;
      V3 x = v3(LDS(L_CX + 3 * k), LDS(L_CX + 3 * k + 1), LDS(L_CX + 3 * k + 2));
      V3 f = v3(LDS(L_LS + 3 * k + 1), LDS(L_LS + 3 * k + 2), LDS(L_LS + 3 * k));
      SV ff = sv(cross(x, f), f);
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (j == depth) pA[j] = pA[j] - ff;
    }
#pragma unroll
    for (int j = 2; j >= 0; j--) {
      const int ji = 3 * leg + j;
      float u = -dot(lds_sv(lds, lane, L_S + 6 * ji), pA[j]);
      LDS(L_UU + ji) = u;
      SV pa = pA[j] + (u * LDS(L_DINV + ji)) * lds_sv(lds, lane, L_U + 6 * ji);
      if (j > 0) pA[j - 1] = pA[j - 1] + pa; else p0 = p0 + pa;
    }
  }
  SV dv0 = -sym6_mul(I0inv, p0);
  s.w = w_free + dv0.a;
  s.v = v_free + dv0.l;
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
    SV a = dv0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      float dqd = LDS(L_DINV + ji) * (LDS(L_UU + ji) - dot(lds_sv(lds, lane, L_U + 6 * ji), a));
      a = a + dqd * lds_sv(lds, lane, L_S + 6 * ji);
      float qd = LDS(L_QD + ji) + dqd;
      // joint velocity limit, semi-implicit Euler, hard position limits
      float vl = (float)GO1_JOINT_VEL_LIMIT[ji];
      qd = fminf(fmaxf(qd, -vl), vl);
      float q = LDS(L_Q + ji) + h * qd;
      float lo = (float)GO1_JOINT_LOWER[ji], hi = (float)GO1_JOINT_UPPER[ji];
      if (q < lo) { q = lo; qd = fmaxf(qd, 0.f); }
      if (q > hi) { q = hi; qd = fminf(qd, 0.f); }
      LDS(L_Q + ji) = q;
      LDS(L_QD + ji) = qd;
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
}

// feet positions / velocities at the current state (reference legged_robot.py:112-115)
DEV void feet_state(const float* lds, int lane, const Base& s, const Go1SimBuffers& B, int e, int N) {
  const M3 R0 = quat_to_mat(s.qx, s.qy, s.qz, s.qw);
#pragma unroll 1
  for (int leg = 0; leg < 4; leg++) {
    M3 Rpar = R0;
    V3 ppar = v3(0.f, 0.f, 0.f);
    SV vb = sv(s.w, s.v);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int ji = 3 * leg + j;
      V3 p = ppar + mul(Rpar, model_v3(GO1_JOINT_ORIGIN, ji));
      V3 ax = (j == 0) ? Rpar.c0 : Rpar.c1;
      float sn, cs;
      sincosf(LDS(L_Q + ji), &sn, &cs);
      Rpar = (j == 0) ? rot_x(Rpar, sn, cs) : rot_y(Rpar, sn, cs);
      vb = vb + LDS(L_QD + ji) * sv(ax, cross(p, ax));
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
}

DEV V3 gravity_at(const Go1SimConfig& cfg, int64_t t) {
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

DEV void resample_commands(const Go1SimConfig& cfg, const Go1SimBuffers& B, int e, int N, int64_t step, uint32_t purpose) {
  if (cfg.device_curriculum) {
    const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
    const int ep_len = cfg.max_episode_length < cfg.resample_interval ? cfg.max_episode_length : cfg.resample_interval;
    bool ok = cfg.curriculum_keys != 0;
#pragma unroll 1
    for (int kx = 0; kx < 4; kx++) {
      if (!(cfg.curriculum_keys & (1 << kx))) continue;
      float val = AT(B.command_sums, cfg.curriculum_sum_index[kx], e) / (float)ep_len;
      if (!(val > cfg.curriculum_threshold[kx])) ok = false;
    }
    int cat_old = B.env_command_categories[e], bin_old = B.env_command_bins[e];
    if (ok) atomicAdd(&B.curriculum_success[cat_old * cfg.num_bins + bin_old], 1);
    float u0 = rng_uniform(cfg, eg, step, purpose, 0), u1 = rng_uniform(cfg, eg, step, purpose, 1);
    int cat = (int)(u0 * cfg.num_categories);
    if (cat >= cfg.num_categories) cat = cfg.num_categories - 1;
    const float* cdf = B.curriculum_cdf + (size_t)cat * cfg.num_bins;
    int bin = 0;
    while (bin < cfg.num_bins - 1 && !(u1 < cdf[bin])) bin++;
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
      float u = rng_uniform(cfg, eg, step, purpose, 2 + kx);
      cmd[kx] = centroid + (u - 0.5f) * bs;
    }
    if (cfg.num_commands > 5) {
      if (cfg.gaitwise_curricula) {
        if (cat == 0) { cmd[5] = fmod1(cmd[5] / 2 - 0.25f); cmd[6] = fmod1(cmd[6] / 2 - 0.25f); cmd[7] = fmod1(cmd[7] / 2 - 0.25f); }
        else if (cat == 1) { cmd[5] = cmd[5] / 2 + 0.25f; cmd[6] = 0.f; cmd[7] = 0.f; }
        else if (cat == 2) { cmd[5] = 0.f; cmd[6] = cmd[6] / 2 + 0.25f; cmd[7] = 0.f; }
        else { cmd[5] = 0.f; cmd[6] = 0.f; cmd[7] = cmd[7] / 2 + 0.25f; }
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

DEV void randomize_dof_props(const Go1SimConfig& cfg, const Go1SimBuffers& B, int e, int N, int64_t step, uint32_t purpose) {
  const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
  if (cfg.randomize_motor_strength) {
    float v = rng_uniform(cfg, eg, step, purpose, 0) * (cfg.motor_strength_range[1] - cfg.motor_strength_range[0]) + cfg.motor_strength_range[0];
#pragma unroll 1
    for (int j = 0; j < 12; j++) AT(B.motor_strengths, j, e) = v;
  }
  if (cfg.randomize_motor_offset) {
#pragma unroll 1
    for (int j = 0; j < 12; j++)
      AT(B.motor_offsets, j, e) = rng_uniform(cfg, eg, step, purpose, 1 + j) * (cfg.motor_offset_range[1] - cfg.motor_offset_range[0]) + cfg.motor_offset_range[0];
  }
  if (cfg.randomize_Kp_factor) {
    float v = rng_uniform(cfg, eg, step, purpose, 13) * (cfg.Kp_factor_range[1] - cfg.Kp_factor_range[0]) + cfg.Kp_factor_range[0];
#pragma unroll 1
    for (int j = 0; j < 12; j++) AT(B.Kp_factors, j, e) = v;
  }
  if (cfg.randomize_Kd_factor) {
    float v = rng_uniform(cfg, eg, step, purpose, 14) * (cfg.Kd_factor_range[1] - cfg.Kd_factor_range[0]) + cfg.Kd_factor_range[0];
#pragma unroll 1
    for (int j = 0; j < 12; j++) AT(B.Kd_factors, j, e) = v;
  }
}

DEV void reset_env(const Go1SimConfig& cfg, const Go1SimBuffers& B, int e, int N, int64_t step) {
  const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
  resample_commands(cfg, B, e, N, step, P_CMD_RESET);
  randomize_dof_props(cfg, B, e, N, step, P_DOFPROPS_RESET);
#pragma unroll 1
  for (int j = 0; j < 12; j++) {
    AT(B.dof_pos, j, e) = cfg.default_dof_pos[j] * (0.5f + rng_uniform(cfg, eg, step, P_RESET, j));
    AT(B.dof_vel, j, e) = 0.f;
  }
  float root[13];
#pragma unroll
  for (int i = 0; i < 13; i++) root[i] = cfg.base_init_state[i];
#pragma unroll
  for (int i = 0; i < 3; i++) root[i] += AT(B.env_origins, i, e);
  if (cfg.custom_origins) {
    root[0] += (2 * rng_uniform(cfg, eg, step, P_RESET, 12) - 1) * cfg.x_init_range + cfg.x_init_offset;
    root[1] += (2 * rng_uniform(cfg, eg, step, P_RESET, 13) - 1) * cfg.y_init_range + cfg.y_init_offset;
  }
  float yaw = (2 * rng_uniform(cfg, eg, step, P_RESET, 14) - 1) * cfg.yaw_init_range;
  root[3] = 0.f; root[4] = 0.f; root[5] = sinf(0.5f * yaw); root[6] = cosf(0.5f * yaw);
#pragma unroll
  for (int i = 0; i < 6; i++) root[7 + i] = rng_uniform(cfg, eg, step, P_RESET, 15 + i) - 0.5f;
#pragma unroll
  for (int i = 0; i < 13; i++) AT(B.root_states, i, e) = root[i];
#pragma unroll 1
  for (int j = 0; j < 12; j++) { AT(B.last_actions, j, e) = 0.f; AT(B.last_last_actions, j, e) = 0.f; AT(B.last_dof_vel, j, e) = 0.f; }
  B.episode_length_buf[e] = 0;
  B.reset_buf[e] = 1;
#pragma unroll 1
  for (int kx = 0; kx <= cfg.num_rewards; kx++) {
    atomicAdd(&B.episode_log[kx], AT(B.episode_sums, kx, e));
    AT(B.episode_sums, kx, e) = 0.f;
  }
  atomicAdd(&B.episode_log[cfg.num_rewards + 1], 1.0f);
  B.gait_indices[e] = 0.f;
  const int nl = cfg.lag_timesteps + 1;
#pragma unroll 1
  for (int sl = 0; sl < nl; sl++)
#pragma unroll 1
    for (int j = 0; j < 12; j++) B.lag_buffer[((size_t)sl * 12 + j) * N + e] = 0.f;
}

// ================================================================================================
// rewards (reference corl_rewards.py:15-202), one raw term per id
// ================================================================================================
struct Derived {
  V3 base_pos, blv, bav, pg, gvec;
  float qx, qy, qz, qw;
};

DEV float cf_norm(const Go1SimBuffers& B, int b, int e, int N) {
  float x = AT(B.contact_forces, 3 * b, e), y = AT(B.contact_forces, 3 * b + 1, e), z = AT(B.contact_forces, 3 * b + 2, e);
  return sqrtf(x * x + y * y + z * z);
}
DEV float normal_cdf(float x, float sigma) { return 0.5f * (1.f + erff(x / (sigma * 1.41421356237309504880f))); }

DEV float reward_term(const Go1SimConfig& cfg, const Go1SimBuffers& B, int e, int N, int id, const Derived& d) {
  float r = 0.f;
  switch (id) {
    case GO1_REW_TRACKING_LIN_VEL: {
      float ex = AT(B.commands, 0, e) - d.blv.x, ey = AT(B.commands, 1, e) - d.blv.y;
      return expf(-(ex * ex + ey * ey) / cfg.tracking_sigma);
    }
    case GO1_REW_TRACKING_ANG_VEL: {
      float ez = AT(B.commands, 2, e) - d.bav.z;
      return expf(-(ez * ez) / cfg.tracking_sigma_yaw);
    }
    case GO1_REW_LIN_VEL_Z: return d.blv.z * d.blv.z;
    case GO1_REW_ANG_VEL_XY: return d.bav.x * d.bav.x + d.bav.y * d.bav.y;
    case GO1_REW_ORIENTATION: return d.pg.x * d.pg.x + d.pg.y * d.pg.y;
    case GO1_REW_TORQUES:
#pragma unroll 1
      for (int j = 0; j < 12; j++) { float t = AT(B.torques, j, e); r = fmaf(t, t, r); }
      return r;
    case GO1_REW_DOF_ACC:
#pragma unroll 1
      for (int j = 0; j < 12; j++) { float a = (AT(B.last_dof_vel, j, e) - AT(B.dof_vel, j, e)) / cfg.dt; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_ACTION_RATE:
#pragma unroll 1
      for (int j = 0; j < 12; j++) { float a = AT(B.last_actions, j, e) - AT(B.actions, j, e); r = fmaf(a, a, r); }
      return r;
    case GO1_REW_COLLISION:
#pragma unroll 1
      for (int b = 0; b < 17; b++) if (cfg.penalised_body_mask & (1u << b)) r += (cf_norm(B, b, e, N) > 0.1f) ? 1.f : 0.f;
      return r;
    case GO1_REW_DOF_POS_LIMITS:
#pragma unroll 1
      for (int j = 0; j < 12; j++) {
        float q = AT(B.dof_pos, j, e);
        float lo = q - cfg.dof_pos_soft_lower[j], hi = q - cfg.dof_pos_soft_upper[j];
        r += -fminf(lo, 0.f) + fmaxf(hi, 0.f);
      }
      return r;
    case GO1_REW_JUMP: {
      float t = d.base_pos.z - (AT(B.commands, 3, e) + cfg.base_height_target);
      return -t * t;
    }
    case GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        float fn = cf_norm(B, 4 + 4 * f, e, N);
        r += -(1.f - AT(B.desired_contact_states, f, e)) * (1.f - expf(-fn * fn / cfg.gait_force_sigma));
      }
      return r / 4;
    case GO1_REW_TRACKING_CONTACTS_SHAPED_VEL:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        float vx = AT(B.foot_velocities, 3 * f, e), vy = AT(B.foot_velocities, 3 * f + 1, e), vz = AT(B.foot_velocities, 3 * f + 2, e);
        float vv = vx * vx + vy * vy + vz * vz;
        r += -(AT(B.desired_contact_states, f, e) * (1.f - expf(-vv / cfg.gait_vel_sigma)));
      }
      return r / 4;
    case GO1_REW_DOF_POS:
#pragma unroll 1
      for (int j = 0; j < 12; j++) { float a = AT(B.dof_pos, j, e) - cfg.default_dof_pos[j]; r = fmaf(a, a, r); }
      return r;
    case GO1_REW_DOF_VEL:
#pragma unroll 1
      for (int j = 0; j < 12; j++) { float a = AT(B.dof_vel, j, e); r = fmaf(a, a, r); }
      return r;
    case GO1_REW_ACTION_SMOOTHNESS_1:
#pragma unroll 1
      for (int j = 0; j < 12; j++) {
        float a = AT(B.joint_pos_target, j, e) - AT(B.last_joint_pos_target, j, e);
        r += a * a * (AT(B.last_actions, j, e) != 0.f ? 1.f : 0.f);
      }
      return r;
    case GO1_REW_ACTION_SMOOTHNESS_2:
#pragma unroll 1
      for (int j = 0; j < 12; j++) {
        float a = AT(B.joint_pos_target, j, e) - 2.f * AT(B.last_joint_pos_target, j, e) + AT(B.last_last_joint_pos_target, j, e);
        r += a * a * (AT(B.last_actions, j, e) != 0.f ? 1.f : 0.f) * (AT(B.last_last_actions, j, e) != 0.f ? 1.f : 0.f);
      }
      return r;
    case GO1_REW_FEET_SLIP:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        bool contact = AT(B.contact_forces, 3 * (4 + 4 * f) + 2, e) > 1.0f;
        bool filt = contact || AT(B.last_contacts, f, e);
        AT(B.last_contacts, f, e) = (uint8_t)contact;
        float vx = AT(B.foot_velocities, 3 * f, e), vy = AT(B.foot_velocities, 3 * f + 1, e);
        r += filt ? (vx * vx + vy * vy) : 0.f;
      }
      return r;
    case GO1_REW_FEET_CONTACT_VEL:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        float vx = AT(B.foot_velocities, 3 * f, e), vy = AT(B.foot_velocities, 3 * f + 1, e), vz = AT(B.foot_velocities, 3 * f + 2, e);
        r += (AT(B.foot_positions, 3 * f + 2, e) < 0.03f) ? (vx * vx + vy * vy + vz * vz) : 0.f;
      }
      return r;
    case GO1_REW_FEET_CONTACT_FORCES:
#pragma unroll 1
      for (int f = 0; f < 4; f++) r += fmaxf(cf_norm(B, 4 + 4 * f, e, N) - cfg.max_contact_force, 0.f);
      return r;
    case GO1_REW_FEET_CLEARANCE_CMD_LINEAR:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        float cl = fminf(fmaxf(AT(B.foot_indices, f, e) * 2.0f - 1.0f, 0.f), 1.f);
        float ph = 1.f - fabsf(1.0f - cl * 2.0f);
        float target = AT(B.commands, 9, e) * ph + 0.02f;
        float df = target - AT(B.foot_positions, 3 * f + 2, e);
        r += df * df * (1.f - AT(B.desired_contact_states, f, e));
      }
      return r;
    case GO1_REW_FEET_IMPACT_VEL:
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        float pv = fminf(fmaxf(AT(B.prev_foot_velocities, 3 * f + 2, e), -100.f), 0.f);
        r += (cf_norm(B, 4 + 4 * f, e, N) > 1.0f) ? pv * pv : 0.f;
      }
      return r;
    case GO1_REW_ORIENTATION_CONTROL: {
      float pitch = AT(B.commands, 10, e), roll = AT(B.commands, 11, e);
      float sr, cr, sp, cp;
      sincosf(-0.5f * roll, &sr, &cr);
      sincosf(-0.5f * pitch, &sp, &cp);
      // quat_mul((sr,0,0,cr), (0,sp,0,cp))
      float x = sr * cp, y = cr * sp, z = sr * sp, w = cr * cp;
      V3 g = quat_rotate_inverse(x, y, z, w, d.gvec);
      float a = d.pg.x - g.x, b = d.pg.y - g.y;
      return a * a + b * b;
    }
    case GO1_REW_RAIBERT_HEURISTIC: {
      float l = rsqrtf(d.qz * d.qz + d.qw * d.qw);
      float yz = -d.qz * l, yw = d.qw * l;
      float width = cfg.num_commands >= 13 ? AT(B.commands, 12, e) : 0.3f;
      float length = cfg.num_commands >= 14 ? AT(B.commands, 13, e) : 0.45f;
      float freq = AT(B.commands, 4, e), xv = AT(B.commands, 0, e), yawv = AT(B.commands, 2, e);
      float yv = yawv * length / 2;
#pragma unroll 1
      for (int f = 0; f < 4; f++) {
        V3 rel = v3(AT(B.foot_positions, 3 * f, e) - d.base_pos.x, AT(B.foot_positions, 3 * f + 1, e) - d.base_pos.y,
                    AT(B.foot_positions, 3 * f + 2, e) - d.base_pos.z);
        V3 fb = quat_rotate(0.f, 0.f, yz, yw, rel);
        float ys = (f % 2 == 0 ? 1.f : -1.f) * width / 2, xs = (f < 2 ? 1.f : -1.f) * length / 2;
        float ph = fabsf(1.0f - AT(B.foot_indices, f, e) * 2.0f) * 1.0f - 0.5f;
        float yo = ph * yv * (0.5f / freq), xo = ph * xv * (0.5f / freq);
        if (f >= 2) yo = -yo;
        float ex = fabsf((xs + xo) - fb.x), ey = fabsf((ys + yo) - fb.y);
        r += ex * ex + ey * ey;
      }
      return r;
    }
    default: return 0.f;
  }
}
DEV int reward_raw_sign(int id) {
  return (id == GO1_REW_JUMP || id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL) ? -1 : 1;
}

// ================================================================================================
// post-physics maps (reference legged_robot.py:90-136)
// ================================================================================================
DEV void post_physics(const Go1SimConfig& cfg, const Go1SimBuffers& B, int e, int N, int64_t counter_post, V3 grav, int history_slot) {
  const uint32_t eg = (uint32_t)(cfg.env_id_offset + e);
  Derived d;
  int ep_len = B.episode_length_buf[e] + 1;
  B.episode_length_buf[e] = ep_len;
  d.base_pos = v3(AT(B.root_states, 0, e), AT(B.root_states, 1, e), AT(B.root_states, 2, e));
  d.qx = AT(B.root_states, 3, e); d.qy = AT(B.root_states, 4, e); d.qz = AT(B.root_states, 5, e); d.qw = AT(B.root_states, 6, e);
  V3 vl = v3(AT(B.root_states, 7, e), AT(B.root_states, 8, e), AT(B.root_states, 9, e));
  V3 va = v3(AT(B.root_states, 10, e), AT(B.root_states, 11, e), AT(B.root_states, 12, e));
  d.blv = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, vl);
  d.bav = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, va);
  d.gvec = (1.f / norm(grav)) * grav;
  d.pg = quat_rotate_inverse(d.qx, d.qy, d.qz, d.qw, d.gvec);
  AT(B.base_lin_vel, 0, e) = d.blv.x; AT(B.base_lin_vel, 1, e) = d.blv.y; AT(B.base_lin_vel, 2, e) = d.blv.z;
  AT(B.base_ang_vel, 0, e) = d.bav.x; AT(B.base_ang_vel, 1, e) = d.bav.y; AT(B.base_ang_vel, 2, e) = d.bav.z;
  AT(B.projected_gravity, 0, e) = d.pg.x; AT(B.projected_gravity, 1, e) = d.pg.y; AT(B.projected_gravity, 2, e) = d.pg.z;

  // ---- _post_physics_step_callback -----------------------------------------------------------
  if (cfg.teleport_robots) {
    float x = AT(B.root_states, 0, e), y = AT(B.root_states, 1, e), th = cfg.teleport_thresh, xo = cfg.teleport_x_offset;
    if (x < th + xo) x += cfg.terrain_length * (cfg.terrain_num_rows - 1);
    if (x > cfg.terrain_length * cfg.terrain_num_rows - th + xo) x -= cfg.terrain_length * (cfg.terrain_num_rows - 1);
    if (y < th) y += cfg.terrain_width * (cfg.terrain_num_cols - 1);
    if (y > cfg.terrain_width * cfg.terrain_num_cols - th) y -= cfg.terrain_width * (cfg.terrain_num_cols - 1);
    AT(B.root_states, 0, e) = x; AT(B.root_states, 1, e) = y;
  }
  if (ep_len % cfg.resample_interval == 0) resample_commands(cfg, B, e, N, counter_post, P_CMD_CB);
  if (cfg.observe_gait_commands) {
    float freq = AT(B.commands, 4, e), phase = AT(B.commands, 5, e), offset = AT(B.commands, 6, e), bound = AT(B.commands, 7, e), dur = AT(B.commands, 8, e);
    float gi = fmod1(B.gait_indices[e] + cfg.dt * freq);
    B.gait_indices[e] = gi;
    float fi[4];
    if (cfg.pacing_offset) { fi[0] = gi + phase + offset + bound; fi[1] = gi + bound; fi[2] = gi + offset; fi[3] = gi + phase; }
    else                   { fi[0] = gi + phase + offset + bound; fi[1] = gi + offset; fi[2] = gi + bound; fi[3] = gi + phase; }
#pragma unroll
    for (int f = 0; f < 4; f++) {
      float rem = fmod1(fi[f]);
      AT(B.foot_indices, f, e) = rem;
      float idx = fi[f];
      if (rem < dur) idx = rem * (0.5f / dur);
      else if (rem > dur) idx = 0.5f + (rem - dur) * (0.5f / (1.f - dur));
      AT(B.clock_inputs, f, e) = sinf(2.f * PI_F * idx);
      float kap = cfg.kappa_gait_probs, x = fmod1(idx);
      float sm = normal_cdf(x, kap) * (1.f - normal_cdf(x - 0.5f, kap)) + normal_cdf(x - 1.f, kap) * (1.f - normal_cdf(x - 0.5f - 1.f, kap));
      AT(B.desired_contact_states, f, e) = sm;
    }
  }
  if (cfg.push_robots && ep_len % cfg.push_interval == 0) {
    AT(B.root_states, 7, e) = (2 * rng_uniform(cfg, eg, counter_post, P_PUSH, 0) - 1) * cfg.max_push_vel_xy;
    AT(B.root_states, 8, e) = (2 * rng_uniform(cfg, eg, counter_post, P_PUSH, 1) - 1) * cfg.max_push_vel_xy;
  }
  if (ep_len % cfg.rand_interval == 0) randomize_dof_props(cfg, B, e, N, counter_post, P_DOFPROPS_CB);

  // ---- check_termination -----------------------------------------------------------------------
  bool reset = false;
#pragma unroll 1
  for (int b = 0; b < 17; b++) if ((cfg.termination_body_mask & (1u << b)) && cf_norm(B, b, e, N) > 1.0f) reset = true;
  bool time_out = ep_len > cfg.max_episode_length;
  reset = reset || time_out;
  if (cfg.use_terminal_body_height && AT(B.root_states, 2, e) < cfg.terminal_body_height) reset = true;
  B.time_out_buf[e] = (uint8_t)time_out;
  B.reset_buf[e] = (uint8_t)reset;

  // ---- compute_reward --------------------------------------------------------------------------
  float rew = 0.f, pos = 0.f, neg = 0.f;
#pragma unroll 1
  for (int kx = 0; kx < cfg.num_rewards; kx++) {
    int id = cfg.reward_ids[kx];
    float sc = cfg.reward_scales[kx];
    float r = reward_term(cfg, B, e, N, id, d) * sc;
    rew += r;
    if (reward_raw_sign(id) * sc >= 0) pos += r; else neg += r;
    AT(B.episode_sums, kx, e) += r;
    if (id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL) AT(B.command_sums, kx, e) += sc + r;
    else AT(B.command_sums, kx, e) += r;
  }
  if (cfg.only_positive_rewards) rew = fmaxf(rew, 0.f);
  else if (cfg.only_positive_rewards_ji22_style) rew = pos * expf(neg / cfg.sigma_rew_neg);
  B.rew_buf[e] = rew;
  AT(B.episode_sums, cfg.num_rewards, e) += rew;
  {
    int k0 = cfg.num_rewards;
    float c0 = AT(B.commands, 0, e), c2 = AT(B.commands, 2, e);
    AT(B.command_sums, k0 + 0, e) += d.blv.x;
    AT(B.command_sums, k0 + 1, e) += d.bav.z;
    AT(B.command_sums, k0 + 2, e) += (d.blv.x - c0) * (d.blv.x - c0);
    AT(B.command_sums, k0 + 3, e) += (d.bav.z - c2) * (d.bav.z - c2);
    AT(B.command_sums, k0 + 4, e) += 1.f;
  }

  // ---- reset -----------------------------------------------------------------------------------
  if (reset) reset_env(cfg, B, e, N, counter_post);

  // ---- compute_observations ----------------------------------------------------------------------
  {
    float* obs_row = B.obs_buf + (size_t)e * cfg.num_obs;
    const int R = cfg.num_obs_history + 1;     // ring slots (one spare keeps the previous window intact)
    float* h0 = B.obs_history ? B.obs_history + (size_t)e * 2 * R * cfg.num_obs + (size_t)history_slot * cfg.num_obs : nullptr;
    float* h1 = h0 ? h0 + (size_t)R * cfg.num_obs : nullptr;
    int n = 0;
    auto emit = [&](float v) {
      if (cfg.add_noise && cfg.noise_scale_vec[n] != 0.f) v += (2 * rng_uniform(cfg, eg, counter_post, P_NOISE, n) - 1) * cfg.noise_scale_vec[n];
      v = fminf(fmaxf(v, -cfg.clip_observations), cfg.clip_observations);
      obs_row[n] = v;
      if (h0) { h0[n] = v; h1[n] = v; }
      n++;
    };
    if (cfg.observe_only_lin_vel) for (int i = 0; i < 3; i++) emit(AT(B.base_lin_vel, i, e) * cfg.obs_scale_lin_vel);
    if (cfg.observe_only_ang_vel) for (int i = 0; i < 3; i++) emit(AT(B.base_ang_vel, i, e) * cfg.obs_scale_ang_vel);
    if (cfg.observe_vel) {
      for (int i = 0; i < 3; i++) emit((cfg.global_reference ? AT(B.root_states, 7 + i, e) : AT(B.base_lin_vel, i, e)) * cfg.obs_scale_lin_vel);
      for (int i = 0; i < 3; i++) emit(AT(B.base_ang_vel, i, e) * cfg.obs_scale_ang_vel);
    }
    for (int i = 0; i < 3; i++) emit(AT(B.projected_gravity, i, e));
    if (cfg.observe_command)
#pragma unroll 1
      for (int kx = 0; kx < cfg.num_commands; kx++) emit(AT(B.commands, kx, e) * cfg.commands_scale[kx]);
#pragma unroll 1
    for (int j = 0; j < 12; j++) emit((AT(B.dof_pos, j, e) - cfg.default_dof_pos[j]) * cfg.obs_scale_dof_pos);
#pragma unroll 1
    for (int j = 0; j < 12; j++) emit(AT(B.dof_vel, j, e) * cfg.obs_scale_dof_vel);
#pragma unroll 1
    for (int j = 0; j < 12; j++) emit(AT(B.actions, j, e));
    if (cfg.observe_two_prev_actions)
#pragma unroll 1
      for (int j = 0; j < 12; j++) emit(AT(B.last_actions, j, e));
    if (cfg.observe_timing_parameter) emit(B.gait_indices[e]);
    if (cfg.observe_clock_inputs) for (int f = 0; f < 4; f++) emit(AT(B.clock_inputs, f, e));
    if (cfg.observe_yaw) {
      V3 fw = quat_rotate(AT(B.root_states, 3, e), AT(B.root_states, 4, e), AT(B.root_states, 5, e), AT(B.root_states, 6, e), v3(1.f, 0.f, 0.f));
      emit(atan2f(fw.y, fw.x));
    }
    if (cfg.observe_contact_states) for (int f = 0; f < 4; f++) emit(AT(B.contact_forces, 3 * (4 + 4 * f) + 2, e) > 1.0f ? 1.0f : 0.0f);

    float* pv = B.privileged_obs_buf + (size_t)e * cfg.num_privileged_obs;
    int np = 0;
    auto priv = [&](int idx, float val) {
      float v = (val - cfg.priv_shift[idx]) * cfg.priv_scale[idx];
      pv[np++] = fminf(fmaxf(v, -cfg.clip_observations), cfg.clip_observations);
    };
    auto privraw = [&](float v) { pv[np++] = fminf(fmaxf(v, -cfg.clip_observations), cfg.clip_observations); };
    if (cfg.priv_enabled[GO1_PRIV_FRICTION]) priv(GO1_PRIV_FRICTION, B.friction_coeffs[e]);
    if (cfg.priv_enabled[GO1_PRIV_RESTITUTION]) priv(GO1_PRIV_RESTITUTION, B.restitutions[e]);
    if (cfg.priv_enabled[GO1_PRIV_BASE_MASS]) priv(GO1_PRIV_BASE_MASS, B.payloads[e]);
    if (cfg.priv_enabled[GO1_PRIV_COM_DISPLACEMENT]) for (int i = 0; i < 3; i++) priv(GO1_PRIV_COM_DISPLACEMENT, AT(B.com_displacements, i, e));
    if (cfg.priv_enabled[GO1_PRIV_MOTOR_STRENGTH])
#pragma unroll 1
      for (int j = 0; j < 12; j++) priv(GO1_PRIV_MOTOR_STRENGTH, AT(B.motor_strengths, j, e));
    if (cfg.priv_enabled[GO1_PRIV_MOTOR_OFFSET])
#pragma unroll 1
      for (int j = 0; j < 12; j++) priv(GO1_PRIV_MOTOR_OFFSET, AT(B.motor_offsets, j, e));
    if (cfg.priv_enabled[GO1_PRIV_BODY_HEIGHT]) priv(GO1_PRIV_BODY_HEIGHT, AT(B.root_states, 2, e));
    if (cfg.priv_enabled[GO1_PRIV_BODY_VELOCITY]) for (int i = 0; i < 3; i++) priv(GO1_PRIV_BODY_VELOCITY, AT(B.base_lin_vel, i, e));
    if (cfg.priv_enabled[GO1_PRIV_GRAVITY]) {
      privraw(((grav.x - cfg.gravity[0]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
      privraw(((grav.y - cfg.gravity[1]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
      privraw(((grav.z - cfg.gravity[2]) - cfg.priv_shift[GO1_PRIV_GRAVITY]) / cfg.priv_scale[GO1_PRIV_GRAVITY]);
    }
    if (cfg.priv_enabled[GO1_PRIV_CLOCK_INPUTS]) for (int f = 0; f < 4; f++) privraw(AT(B.clock_inputs, f, e));
    if (cfg.priv_enabled[GO1_PRIV_DESIRED_CONTACT]) for (int f = 0; f < 4; f++) privraw(AT(B.desired_contact_states, f, e));
  }
  // ---- roll ------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int j = 0; j < 12; j++) {
    AT(B.last_last_actions, j, e) = AT(B.last_actions, j, e);
    AT(B.last_actions, j, e) = AT(B.actions, j, e);
    AT(B.last_last_joint_pos_target, j, e) = AT(B.last_joint_pos_target, j, e);
    AT(B.last_joint_pos_target, j, e) = AT(B.joint_pos_target, j, e);
    AT(B.last_dof_vel, j, e) = AT(B.dof_vel, j, e);
  }
}

// ================================================================================================
// kernels
// ================================================================================================
DEV void load_state(const Go1SimBuffers& B, float* lds, int lane, int e, int N, Base& s) {
  s.pos = v3(AT(B.root_states, 0, e), AT(B.root_states, 1, e), AT(B.root_states, 2, e));
  s.qx = AT(B.root_states, 3, e); s.qy = AT(B.root_states, 4, e); s.qz = AT(B.root_states, 5, e); s.qw = AT(B.root_states, 6, e);
  s.v = v3(AT(B.root_states, 7, e), AT(B.root_states, 8, e), AT(B.root_states, 9, e));
  s.w = v3(AT(B.root_states, 10, e), AT(B.root_states, 11, e), AT(B.root_states, 12, e));
#pragma unroll 1
  for (int j = 0; j < 12; j++) { LDS(L_Q + j) = AT(B.dof_pos, j, e); LDS(L_QD + j) = AT(B.dof_vel, j, e); }
  s.mass0 = (float)GO1_BODY_MASS[0] + B.payloads[e];
  s.com0 = v3(AT(B.com_displacements, 0, e), AT(B.com_displacements, 1, e), AT(B.com_displacements, 2, e));
  s.mu = B.friction_coeffs[e];
  s.rest = B.restitutions[e];
}
DEV void store_state(const Go1SimBuffers& B, const float* lds, int lane, int e, int N, const Base& s) {
  AT(B.root_states, 0, e) = s.pos.x; AT(B.root_states, 1, e) = s.pos.y; AT(B.root_states, 2, e) = s.pos.z;
  AT(B.root_states, 3, e) = s.qx; AT(B.root_states, 4, e) = s.qy; AT(B.root_states, 5, e) = s.qz; AT(B.root_states, 6, e) = s.qw;
  AT(B.root_states, 7, e) = s.v.x; AT(B.root_states, 8, e) = s.v.y; AT(B.root_states, 9, e) = s.v.z;
  AT(B.root_states, 10, e) = s.w.x; AT(B.root_states, 11, e) = s.w.y; AT(B.root_states, 12, e) = s.w.z;
#pragma unroll 1
  for (int j = 0; j < 12; j++) { AT(B.dof_pos, j, e) = LDS(L_Q + j); AT(B.dof_vel, j, e) = LDS(L_QD + j); }
}
DEV void load_lambda(const Go1SimConfig& cfg, const Go1SimBuffers& B, float* lds, int lane, int e, int N) {
#pragma unroll 1
  for (int b = 0; b < 17; b++) {   // world force -> (n, t1, t2) = (z, x, y) impulses
    LDS(L_LAM + 3 * b) = AT(B.contact_forces, 3 * b + 2, e) * cfg.sim_dt;
    LDS(L_LAM + 3 * b + 1) = AT(B.contact_forces, 3 * b, e) * cfg.sim_dt;
    LDS(L_LAM + 3 * b + 2) = AT(B.contact_forces, 3 * b + 1, e) * cfg.sim_dt;
  }
}
DEV void store_forces(const Go1SimConfig& cfg, const Go1SimBuffers& B, const float* lds, int lane, int e, int N) {
  const float inv = 1.f / cfg.sim_dt;
#pragma unroll 1
  for (int b = 0; b < 17; b++) {
    AT(B.contact_forces, 3 * b, e) = LDS(L_LAM + 3 * b + 1) * inv;
    AT(B.contact_forces, 3 * b + 1, e) = LDS(L_LAM + 3 * b + 2) * inv;
    AT(B.contact_forces, 3 * b + 2, e) = LDS(L_LAM + 3 * b) * inv;
  }
}

extern "C" __global__ void __launch_bounds__(WAVE) go1_step_kernel(const StepArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Go1SimConfig& cfg = A.sc->cfg;
  const Go1SimBuffers& B = A.sc->buf;
  const int N = cfg.num_envs;
  const int lane = threadIdx.x;
  const int e = blockIdx.x * WAVE + lane;
  if (e >= N) return;
  const float h = cfg.sim_dt;
  Base s;
  load_state(B, lds, lane, e, N, s);
  const V3 grav = gravity_at(cfg, A.counter);
#pragma unroll 1
  for (int j = 0; j < 12; j++) {
    float a = A.actions[(size_t)e * 12 + j];
    AT(B.actions, j, e) = fminf(fmaxf(a, -cfg.clip_actions), cfg.clip_actions);
  }
#pragma unroll 1
  for (int i = 0; i < 12; i++) AT(B.prev_foot_velocities, i, e) = AT(B.foot_velocities, i, e);
  const bool warm = cfg.warm_start && B.episode_length_buf[e] > 0;
  if (warm) load_lambda(cfg, B, lds, lane, e, N);
  else {
#pragma unroll 1
    for (int i = 0; i < 51; i++) LDS(L_LAM + i) = 0.f;
  }
  const int nl = cfg.lag_timesteps + 1;
  int head = A.lag_head;
#pragma unroll 1
  for (int sub = 0; sub < cfg.decimation; sub++) {
    compute_torques(cfg, B, lds, lane, e, N, head);
    head = (head + 1) % nl;
    physics_substep(cfg, lds, lane, s, grav, warm || (cfg.warm_start && sub > 0), h);
  }
  store_state(B, lds, lane, e, N, s);
  feet_state(lds, lane, s, B, e, N);
  store_forces(cfg, B, lds, lane, e, N);
  post_physics(cfg, B, e, N, A.counter + 1, grav, A.history_slot);
}

// piecewise entry points (parity tests, reset_idx): same device functions, kept out of the hot kernel
extern "C" __global__ void __launch_bounds__(WAVE) go1_aux_kernel(const StepArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Go1SimConfig& cfg = A.sc->cfg;
  const Go1SimBuffers& B = A.sc->buf;
  const int N = cfg.num_envs;
  const int lane = threadIdx.x;
  const int e = blockIdx.x * WAVE + lane;
  if (e >= N) return;
  if (A.mode == 3) {       // reset_idx
    if (e < A.n_ids) reset_env(cfg, B, A.ids ? A.ids[e] : e, N, A.counter);
    return;
  }
  if (A.mode == 4) {       // tensor maps only
    post_physics(cfg, B, e, N, A.counter + 1, v3(A.gravity_override[0], A.gravity_override[1], A.gravity_override[2]), A.history_slot);
    return;
  }
  Base s;
  load_state(B, lds, lane, e, N, s);
  if (A.mode == 1) {       // torques only (actions given as SoA)
#pragma unroll 1
    for (int j = 0; j < 12; j++) AT(B.actions, j, e) = AT(A.actions, j, e);
    compute_torques(cfg, B, lds, lane, e, N, A.lag_head);
    return;
  }
  // mode 2: one physics substep with the torques in the buffer
  const V3 grav = gravity_at(cfg, A.counter);
#pragma unroll 1
  for (int j = 0; j < 12; j++) LDS(L_TAU + j) = AT(B.torques, j, e);
  load_lambda(cfg, B, lds, lane, e, N);
  physics_substep(cfg, lds, lane, s, grav, cfg.warm_start != 0, cfg.sim_dt);
  store_state(B, lds, lane, e, N, s);
  feet_state(lds, lane, s, B, e, N);
  store_forces(cfg, B, lds, lane, e, N);
}

// HistoryWrapper.get_observations: append the current obs_buf to the double-length ring
extern "C" __global__ void __launch_bounds__(256) go1_history_kernel(const SimConst* __restrict__ sc, int slot) {
  const Go1SimConfig& cfg = sc->cfg;
  const Go1SimBuffers& B = sc->buf;
  const int no = cfg.num_obs, R = cfg.num_obs_history + 1;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)cfg.num_envs * no) return;
  const size_t e = i / no, c = i % no;
  float v = B.obs_buf[i];
  float* row = B.obs_history + e * 2 * R * no;
  row[(size_t)slot * no + c] = v;
  row[(size_t)(slot + R) * no + c] = v;
}

// curriculum weight update + CDF rebuild (reference curriculum.py:135-154): one workgroup per category
extern "C" __global__ void __launch_bounds__(256) go1_curriculum_kernel(const SimConst* __restrict__ sc) {
  const Go1SimConfig& cfg = sc->cfg;
  const Go1SimBuffers& B = sc->buf;
  __shared__ float part[256];
  const int c = blockIdx.x, nb = cfg.num_bins, t = threadIdx.x;
  float* w = B.curriculum_weights + (size_t)c * nb;
  int32_t* s = B.curriculum_success + (size_t)c * nb;
  float* cdf = B.curriculum_cdf + (size_t)c * nb;
  // 1. increments (read all successes before anyone clears them)
  const int per = (nb + 255) / 256;
  float local = 0.f;
  for (int i = 0; i < per; i++) {
    int b = t * per + i;
    if (b < nb) {
      int cnt = s[b] > 0 ? 1 : 0;
      for (int p = B.curriculum_nbr_ptr[b]; p < B.curriculum_nbr_ptr[b + 1]; p++) cnt += s[B.curriculum_nbr_idx[p]];
      float nw = fminf(1.0f, w[b] + 0.2f * cnt);
      cdf[b] = nw;        // stage the new weight in the cdf array until every thread has read `s`
      local += nw;
    }
  }
  __syncthreads();
  for (int i = 0; i < per; i++) {
    int b = t * per + i;
    if (b < nb) { w[b] = cdf[b]; s[b] = 0; }
  }
  // 2. block prefix sum of the per-thread totals
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
  if (cfg->num_obs > GO1_MAX_OBS || cfg->num_privileged_obs > GO1_MAX_PRIV_OBS || cfg->num_rewards > GO1_MAX_REWARDS) return -4;
  if (cfg->terrain_type != 0) return -5;       // height-field contact: SURVEY.md §8 config 3, next row
  return 0;
}

static int upload_const(Go1Sim* s) {
  SimConst h;
  h.cfg = s->cfg; h.buf = s->buf;
  return hipMemcpy(s->dconst, &h, sizeof(SimConst), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

extern "C" int go1sim_create(const Go1SimConfig* cfg, const Go1SimBuffers* buffers, int device, Go1Sim** out) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!buffers || !out) return -1;
  if (hipSetDevice(device) != hipSuccess) return -10;
  Go1Sim* s = new Go1Sim();
  s->cfg = *cfg; s->buf = *buffers; s->device = device;
  s->counter = 0; s->lag_head = 0; s->history_slot = 0; s->timing_cap = 0; s->timing_n = 0; s->ev = nullptr;
  if (hipFuncSetAttribute((const void*)go1_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L_END * WAVE * 4) != hipSuccess ||
      hipFuncSetAttribute((const void*)go1_aux_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L_END * WAVE * 4) != hipSuccess) {
    delete s;
    return -11;
  }
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
  s->cfg = *cfg;
  return upload_const(s);      // blocking copy: configuration changes are rare and never on the step path
}

static int launch(Go1Sim* s, int mode, const float* actions, const int32_t* ids, int n_ids, hipStream_t st, bool timed,
                  const float* grav = nullptr) {
  StepArgs A;
  for (int i = 0; i < 3; i++) A.gravity_override[i] = grav ? grav[i] : 0.f;
  A.sc = s->dconst; A.actions = actions; A.counter = s->counter; A.lag_head = s->lag_head;
  A.history_slot = s->history_slot; A.mode = mode; A.ids = ids; A.n_ids = n_ids;
  const int n = (mode == 3) ? n_ids : s->cfg.num_envs;
  dim3 grid((n + WAVE - 1) / WAVE), block(WAVE);
  const int slot = timed ? (int)(s->timing_n % s->timing_cap) : 0;
  if (timed) (void)hipEventRecord(s->ev[2 * slot], st);
  if (mode == 0) hipLaunchKernelGGL(go1_step_kernel, grid, block, L_END * WAVE * 4, st, A);
  else hipLaunchKernelGGL(go1_aux_kernel, grid, block, L_END * WAVE * 4, st, A);
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
  if (s->cfg.device_curriculum && s->buf.curriculum_weights) {
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
extern "C" const char* go1sim_version(void) { return "go1sim 0.1 (gfx950, abi 1)"; }
