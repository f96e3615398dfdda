/* go1_oracle.c — CPU ORACLE (test infrastructure, NOT a product path).
 *
 * fp64 restatement of the Go1 vectorised step behind `LeggedRobot.step`
 * (reference: go1_gym/envs/base/legged_robot.py:60-136).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; the shipped path is the HIP library and it
 * fails loudly when that library is missing.
 *
 * PARITY STATUS
 *   - tensor maps (torque model, derived state, gait clock, termination, rewards, observations):
 *     PINNED against the reference's own Python executed by method borrowing
 *     (tests/golden/make_golden.py -> tests/golden/ .npz files, tests/test_oracle_golden.py).
 *   - physics substep: PARITY UNPINNED.  The reference delegates it to the closed Isaac Gym /
 *     PhysX binary (legged_robot.py:76-80), which is absent from /root/reference and cannot be
 *     executed; no golden trajectories exist upstream.  The algorithm below is a
 *     restatement of the *contract* (same model, dt, limits, contact/friction semantics) and is
 *     validated by physical invariants (tests/test_oracle_physics.py).
 *
 * The physics is deliberately formulated differently from the HIP kernel so that agreement
 * between the two checks the mathematics, not a shared implementation:
 *   oracle: classical 3-vector recursive Newton-Euler, dense 18x18 mass matrix from 18 RNEA
 *           columns, Cholesky solves, dense M^-1 J^T.
 *   kernel: spatial-algebra articulated-body algorithm (Featherstone ABA), O(n) impulse
 *           propagation with the ABA factors.
 * Both then run the identical projected Gauss-Seidel sweep order, so results agree to fp32
 * round-off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/go1sim.h"
#include "../walk-these-ways_amd/csrc/go1_model_data.h"
#include "../walk-these-ways_amd/csrc/go1_actuator_data.h"

#ifdef GO1_ORACLE_FP32
typedef float real;      /* fp32 build (oracle/_build/libgo1oracle32.so): the SAME restatement in the kernel's precision — what
                            separates it from the fp64 build is round-off, which is how the tests attribute environments whose
                            contact set flips to precision rather than to logic */
#else
typedef double real;
#endif
#define NV 18
#define PI 3.14159265358979323846

/* ------------------------------------------------------------------ small vector helpers */
static inline void v3set(real* a, real x, real y, real z) { a[0] = x; a[1] = y; a[2] = z; }
static inline void v3cpy(real* a, const real* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static inline void v3add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3axpy(real* o, real s, const real* a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static inline real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3norm(const real* a) { return sqrt(v3dot(a, a)); }
/* o = R (row-major 3x3) * v */
static inline void m3v(real* o, const real* R, const real* v) {
  real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(real* o, const real* R, const real* v) {
  real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  real y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  real z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3m(real* o, const real* A, const real* B) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(o, t, sizeof t);
}
/* quaternion xyzw -> rotation matrix (body -> world) */
static void quat_to_mat(real* R, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
/* isaacgym.torch_utils semantics restated from the maths */
static void quat_rotate(real* o, const real* q, const real* v) {   /* R(q) v */
  real u[3] = {q[0], q[1], q[2]}, t[3], t2[3];
  v3cross(t, u, v);
  v3cross(t2, u, t);
  for (int i = 0; i < 3; i++) o[i] = v[i] + 2 * q[3] * t[i] + 2 * t2[i];
}
static void quat_rotate_inverse(real* o, const real* q, const real* v) {   /* R(q)^T v */
  real qc[4] = {-q[0], -q[1], -q[2], q[3]};
  quat_rotate(o, qc, v);
}
static void quat_mul(real* o, const real* a, const real* b) {   /* Hamilton product, xyzw */
  real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  real y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  real z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* ------------------------------------------------------------------ Philox4x32-10 (Salmon et al., SC'11) */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}
void go1_oracle_philox(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr_in[0], ctr_in[1], ctr_in[2], ctr_in[3]};
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; r++) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c[0], &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c[2], &hi1, &lo1);
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  memcpy(out, c, sizeof c);
}
enum { P_NOISE = 1, P_RESET = 2, P_DOFPROPS_CB = 3, P_DOFPROPS_RESET = 4, P_CMD_CB = 5, P_CMD_RESET = 6,
       P_PUSH = 7, P_GRAVITY = 8, P_RIGID = 9, P_RIGID_RESET = 10 };
/* uniform [0,1) number `idx` of stream (env, step, purpose) */
static float rng_uniform(const Go1SimConfig* cfg, uint32_t env_global, int64_t step, uint32_t purpose, uint32_t idx) {
  uint32_t ctr[4] = {env_global, (uint32_t)step, purpose, idx >> 2};
  uint32_t key[2] = {(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
  uint32_t out[4];
  go1_oracle_philox(ctr, key, out);
  return (float)(out[idx & 3] >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ per-env working state */
typedef struct {
  real pos[3], quat[4], vlin[3], vang[3];  /* root, world */
  real q[12], qd[12];
  real mass0;                              /* base mass incl. payload */
  real com0[3];                            /* base com (= com_displacement, legged_robot.py:671) */
  real mu, rest;                           /* robot material */
} Phys;

typedef struct {
  real R[13][9];       /* body -> world */
  real p[13][3];       /* body frame origin, world, RELATIVE to base origin */
  real axis[12][3];    /* joint axis, world */
  real com[13][3];     /* body com, world, relative to base origin */
  real Iw[13][9];      /* inertia about com, world axes */
  real mass[13];
} Kin;

static void rot_axis(real* R, int axis, real a) {
  real c = cos(a), s = sin(a);
  if (axis == 0) { real t[9] = {1, 0, 0, 0, c, -s, 0, s, c}; memcpy(R, t, sizeof t); }
  else           { real t[9] = {c, 0, s, 0, 1, 0, -s, 0, c}; memcpy(R, t, sizeof t); }
}

static void kinematics(const Phys* s, Kin* k) {
  quat_to_mat(k->R[0], s->quat);
  v3set(k->p[0], 0, 0, 0);
  for (int leg = 0; leg < 4; leg++) {
    for (int j = 0; j < 3; j++) {
      int b = 1 + 3 * leg + j, par = (j == 0) ? 0 : b - 1, ji = b - 1;
      real r[3] = {GO1_JOINT_ORIGIN[ji][0], GO1_JOINT_ORIGIN[ji][1], GO1_JOINT_ORIGIN[ji][2]};
      real off[3];
      m3v(off, k->R[par], r);
      v3add(k->p[b], k->p[par], off);
      real ax[3] = {0, 0, 0};
      ax[GO1_JOINT_AXIS[ji]] = 1;
      m3v(k->axis[ji], k->R[par], ax);
      real Rj[9];
      rot_axis(Rj, GO1_JOINT_AXIS[ji], s->q[ji]);
      m3m(k->R[b], k->R[par], Rj);
    }
  }
  for (int b = 0; b < 13; b++) {
    real c[3], I[9];
    real scale = 1;
    if (b == 0) {
      v3cpy(c, s->com0);
      k->mass[0] = s->mass0;
      scale = s->mass0 / GO1_BODY_MASS[0];   /* recomputeInertia=True restated as mass-proportional scaling */
    } else {
      v3set(c, GO1_BODY_COM[b][0], GO1_BODY_COM[b][1], GO1_BODY_COM[b][2]);
      k->mass[b] = GO1_BODY_MASS[b];
    }
    real cw[3];
    m3v(cw, k->R[b], c);
    v3add(k->com[b], k->p[b], cw);
    const double* i6 = GO1_BODY_INERTIA[b];
    real Il[9] = {i6[0], i6[1], i6[2], i6[1], i6[3], i6[4], i6[2], i6[4], i6[5]};
    for (int i = 0; i < 9; i++) Il[i] *= scale;
    real Rt[9] = {k->R[b][0], k->R[b][3], k->R[b][6], k->R[b][1], k->R[b][4], k->R[b][7], k->R[b][2], k->R[b][5], k->R[b][8]};
    m3m(I, k->R[b], Il);
    m3m(k->Iw[b], I, Rt);
  }
}

/* Classical recursive Newton-Euler.  Generalised coordinates: [omega(3), v_O(3), qd(12)] with
 * omega, v_O the base angular velocity / base-origin velocity in WORLD axes; generalised
 * accelerations are their classical time derivatives.  Output f = M(q) acc + bias(q, vel) - gravity terms,
 * conjugate to those coordinates (moment about the base origin, force, joint torques). */
static void rnea(const Kin* k, const real* vel, const real* acc, const real* grav, real* f) {
  real w[13][3], wd[13][3], a[13][3];   /* angular vel, angular acc, linear acc of body origin */
  real F[13][3], Nm[13][3];
  v3set(w[0], 0, 0, 0); v3set(wd[0], acc[0], acc[1], acc[2]);
  if (vel) v3set(w[0], vel[0], vel[1], vel[2]);
  for (int i = 0; i < 3; i++) a[0][i] = acc[3 + i] - (grav ? grav[i] : 0);
  for (int b = 1; b < 13; b++) {
    int ji = b - 1, par = ((b - 1) % 3 == 0) ? 0 : b - 1;
    real qd = vel ? vel[6 + ji] : 0, qdd = acc[6 + ji];
    real d[3], t[3], t2[3];
    v3sub(d, k->p[b], k->p[par]);
    /* origin of b is fixed in the parent */
    v3cross(t, wd[par], d);
    v3cross(t2, w[par], d);
    v3cross(t2, w[par], t2);
    for (int i = 0; i < 3; i++) a[b][i] = a[par][i] + t[i] + t2[i];
    for (int i = 0; i < 3; i++) w[b][i] = w[par][i] + k->axis[ji][i] * qd;
    v3cross(t, w[par], k->axis[ji]);
    for (int i = 0; i < 3; i++) wd[b][i] = wd[par][i] + k->axis[ji][i] * qdd + t[i] * qd;
  }
  for (int b = 0; b < 13; b++) {
    real d[3], t[3], t2[3], ac[3], Iw[3];
    v3sub(d, k->com[b], k->p[b]);
    v3cross(t, wd[b], d);
    v3cross(t2, w[b], d);
    v3cross(t2, w[b], t2);
    for (int i = 0; i < 3; i++) ac[i] = a[b][i] + t[i] + t2[i];
    for (int i = 0; i < 3; i++) F[b][i] = k->mass[b] * ac[i];
    m3v(Nm[b], k->Iw[b], wd[b]);
    m3v(Iw, k->Iw[b], w[b]);
    v3cross(t, w[b], Iw);
    v3add(Nm[b], Nm[b], t);
  }
  /* backward: accumulate force and moment about each body's origin */
  real fa[13][3], na[13][3];
  for (int b = 12; b >= 0; b--) {
    real d[3], t[3];
    v3cpy(fa[b], F[b]);
    v3sub(d, k->com[b], k->p[b]);
    v3cross(t, d, F[b]);
    v3add(na[b], Nm[b], t);
    /* children */
    for (int c = 1; c < 13; c++) {
      int par = ((c - 1) % 3 == 0) ? 0 : c - 1;
      if (par != b) continue;
      v3add(fa[b], fa[b], fa[c]);
      v3sub(d, k->p[c], k->p[b]);
      v3cross(t, d, fa[c]);
      v3add(na[b], na[b], na[c]);
      v3add(na[b], na[b], t);
    }
    if (b == 0) break;
  }
  /* the child loop above needs children processed before parents: bodies are numbered so that
   * children have larger indices, and fa/na of a child are final when its parent is visited. */
  for (int i = 0; i < 3; i++) { f[i] = na[0][i]; f[3 + i] = fa[0][i]; }
  for (int ji = 0; ji < 12; ji++) f[6 + ji] = v3dot(k->axis[ji], na[ji + 1]);
}

/* dense symmetric positive definite solve, in place Cholesky (lower) */
static int cholesky(real* A, int n) {
  for (int j = 0; j < n; j++) {
    real d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (d <= 0) return -1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      real s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(const real* L, int n, real* b) {
  for (int i = 0; i < n; i++) {
    real s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------ terrain */
typedef struct { const Go1SimConfig* cfg; const int16_t* hs; } Terrain;
typedef struct { int on; real n[3]; real d; real top; } Wall;
static __thread int g_last_cell;      /* height-field cell of the last terrain_sample() call (contact signature) */   /* vertical face: unit horizontal normal (towards the low side),
                                                                    horizontal distance of the sample point, height of its upper edge */
/* Height and unit normal of the terrain's TOP surface at world (x, y), and the vertical face next to the point (if any).
 * Height field: bilinear interpolation of the int16 samples (same sample convention as _get_heights,
 * legged_robot.py:1793-1806: index = (x + border) / hscale).
 * Vertical faces (hf_wall_units T > 0; the `trimesh` terrain's slope_treshold, terrain.py:33-36, legged_robot_config.py:91:
 * where two neighbouring samples differ by more than the threshold the reference's mesh moves the LOWER vertex under the upper
 * one, so the low ground runs on flat to a vertical riser).  Restated on the height field per cell: an edge of the cell whose
 * end heights differ by more than T is "steep"; the cell's corner heights are lowered along steep edges to the lower end (two
 * passes over the four edges: the low level spreads through chains of steep edges) and the top surface is the bilinear
 * interpolant of the LOWERED corners; if both x-edges of the cell are steep in the same direction the cell ends in a wall in
 * the plane x = x(high side), normal towards the low side, upper edge = the interpolated original heights of the high side
 * (same for y; with both, the nearer one is reported). */
static void terrain_sample(const Terrain* t, real x, real y, real* h, real* n, Wall* wall) {
  const Go1SimConfig* c = t->cfg;
  if (wall) wall->on = 0;
  g_last_cell = 0;
  if (c->terrain_type == 0 || !t->hs) { *h = 0; v3set(n, 0, 0, 1); return; }
  real fx = (x + c->hf_border) / c->hf_hscale, fy = (y + c->hf_border) / c->hf_hscale;
  if (fx < 0) fx = 0; if (fy < 0) fy = 0;
  if (fx > c->hf_rows - (real)1.000001) fx = c->hf_rows - (real)1.000001;
  if (fy > c->hf_cols - (real)1.000001) fy = c->hf_cols - (real)1.000001;
  int ix = (int)fx, iy = (int)fy;
  real ax = fx - ix, ay = fy - iy;
  g_last_cell = ix * c->hf_cols + iy;
  const int p00 = t->hs[ix * c->hf_cols + iy], p10 = t->hs[(ix + 1) * c->hf_cols + iy], p01 = t->hs[ix * c->hf_cols + iy + 1], p11 = t->hs[(ix + 1) * c->hf_cols + iy + 1];
  real h00 = p00 * (real)c->hf_vscale, h10 = p10 * (real)c->hf_vscale, h01 = p01 * (real)c->hf_vscale, h11 = p11 * (real)c->hf_vscale;
  const int T = c->hf_wall_units;
  if (T > 0) {
    const int dx0 = p10 - p00, dx1 = p11 - p01, dy0 = p01 - p00, dy1 = p11 - p10;          /* steepness is decided on the integer samples */
    const int sx0 = abs(dx0) > T, sx1 = abs(dx1) > T, sy0 = abs(dy0) > T, sy1 = abs(dy1) > T;
    if (sx0 || sx1 || sy0 || sy1) {
      if (wall) {
        real best = 1e30;
        if (sx0 && sx1 && (dx0 > 0) == (dx1 > 0)) {
          const int up = dx0 > 0;
          wall->on = 1; v3set(wall->n, up ? -1 : 1, 0, 0);
          wall->d = (up ? (1 - ax) : ax) * (real)c->hf_hscale;
          wall->top = up ? h10 * (1 - ay) + h11 * ay : h00 * (1 - ay) + h01 * ay;
          best = wall->d;
        }
        if (sy0 && sy1 && (dy0 > 0) == (dy1 > 0)) {
          const int up = dy0 > 0;
          const real d = (up ? (1 - ay) : ay) * (real)c->hf_hscale;
          if (d < best) {
            wall->on = 1; v3set(wall->n, 0, up ? -1 : 1, 0);
            wall->d = d;
            wall->top = up ? h01 * (1 - ax) + h11 * ax : h00 * (1 - ax) + h10 * ax;
          }
        }
      }
      for (int pass = 0; pass < 2; pass++) {
        if (sx0) { real lo = h00 < h10 ? h00 : h10; h00 = h10 = lo; }
        if (sx1) { real lo = h01 < h11 ? h01 : h11; h01 = h11 = lo; }
        if (sy0) { real lo = h00 < h01 ? h00 : h01; h00 = h01 = lo; }
        if (sy1) { real lo = h10 < h11 ? h10 : h11; h10 = h11 = lo; }
      }
    }
  }
  *h = h00 * (1 - ax) * (1 - ay) + h10 * ax * (1 - ay) + h01 * (1 - ax) * ay + h11 * ax * ay;
  real dhdx = ((h10 - h00) * (1 - ay) + (h11 - h01) * ay) / c->hf_hscale;
  real dhdy = ((h01 - h00) * (1 - ax) + (h11 - h10) * ax) / c->hf_hscale;
  real nn[3] = {-dhdx, -dhdy, 1};
  real l = v3norm(nn);
  v3set(n, nn[0] / l, nn[1] / l, nn[2] / l);
}

/* ------------------------------------------------------------------ contacts
 * Terrain, top surface: every collision shape of the URDF contributes points of its own —
 *   trunk box: every corner within the contact distance, at most four (corner order), so that a trunk lying on a face rests on
 *              that face's four corners (two points per shape left it rocking about the line through them);
 *   hip capsules [replace_cylinder_with_capsule, legged_robot_config.py:232], thigh and calf boxes: up to TWO — the candidate
 *              points (capsule end spheres, box corners) are split into the two ends of the shape's long axis, the deeper end's
 *              deepest point is the first contact, the other end's deepest point the second;
 *   foot spheres: one.
 * Terrain, vertical faces (hf_wall_units > 0): one point per foot, calf, thigh and one for the trunk — the candidate point
 * below the face's upper edge with the smallest horizontal separation from it; normal horizontal.
 * Self-collision (asset self_collisions = 0: nothing filtered, go1_config.py:44, legged_robot.py:1563-1564): capsules —
 * lower leg (knee -> foot centre, radius of the foot sphere), thigh (thigh joint -> knee, GO1_SELF_THIGH_RADIUS), trunk (the
 * box's long axis, radius = its half width) — lower legs and thighs of DIFFERENT legs against each other (24 pairs) and
 * lower legs against the trunk (4), and since round 5 the hip capsules against the OTHER legs' lower legs (12: the one combination with a
 * hip capsule the joint limits let touch); closest points of the two segments; two legs touch in ONE point (the deepest of their six
 * capsule combinations).  Not modelled: pairs within one leg, hip capsules against thighs / each other, trunk against thighs —
 * out of reach within the joint limits (tests/test_self_collision_reach.py samples the limit box).
 * Solver list: at most GO1_MAX_CONTACTS, in the priority order feet, foot walls, self-contacts (at most 6 leg-leg), trunk, trunk wall, calves (first points, walls, second points), thighs (same), hips; what does not fit is dropped
 * and counted per class. */
typedef struct { real phi, x[3], n[3]; int valid; uint32_t tag; } Cand;   /* tag: 16 * height-field cell + candidate point index */
typedef struct {
  int repA, repB;   /* reported bodies (0..16) the force is booked on; repB = -1: terrain */
  int dynA, dynB;   /* dynamic bodies (0..12) carrying the point; dynB = -1: terrain */
  real x[3];        /* contact point, world axes, relative to the base origin */
  real n[3], t1[3], t2[3];
  real phi;         /* signed separation */
  real share;       /* part of repA's previous impulse this point starts from (warm start) */
  int cls;          /* Go1ContactClass */
  int top;          /* 1: on the terrain's top surface (takes part in the warm-start shares) */
} Contact;

/* candidate point `local` of dynamic body dynb: top-surface candidate into *best, wall candidate into *bestw (may be NULL) */
static void candidate(const Terrain* ter, const Kin* k, const real* base_pos, int dynb, const real* local, real radius, Cand* best, Cand* bestw, int m) {
  real w[3], x[3];
  m3v(w, k->R[dynb], local);
  v3add(x, k->p[dynb], w);
  real h, n[3];
  Wall wl;
  terrain_sample(ter, base_pos[0] + x[0], base_pos[1] + x[1], &h, n, bestw ? &wl : NULL);
  real phi = (base_pos[2] + x[2]) - radius - h;
  if (!best->valid || phi < best->phi) {
    best->valid = 1;
    best->tag = 16u * (uint32_t)g_last_cell + (uint32_t)m;
    best->phi = phi;
    for (int i = 0; i < 3; i++) best->x[i] = x[i] - radius * n[i];
    v3cpy(best->n, n);
  }
  if (bestw && wl.on && base_pos[2] + x[2] < wl.top) {
    real phiw = wl.d - radius;
    if (!bestw->valid || phiw < bestw->phi) {
      bestw->valid = 1;
      bestw->tag = 16u * (uint32_t)g_last_cell + (uint32_t)m;
      bestw->phi = phiw;
      for (int i = 0; i < 3; i++) bestw->x[i] = x[i] - radius * wl.n[i];
      v3cpy(bestw->n, wl.n);
    }
  }
}

static void contact_frame(Contact* c) {   /* t1 = x axis projected on the tangent plane (y axis if n is along x), t2 = n x t1 */
  real ex[3] = {1, 0, 0};
  real d = v3dot(ex, c->n);
  for (int i = 0; i < 3; i++) c->t1[i] = ex[i] - d * c->n[i];
  real l2 = v3dot(c->t1, c->t1);
  if (!(l2 > (real)1e-12)) {
    real ey[3] = {0, 1, 0};
    d = v3dot(ey, c->n);
    for (int i = 0; i < 3; i++) c->t1[i] = ey[i] - d * c->n[i];
    l2 = v3dot(c->t1, c->t1);
    if (!(l2 > (real)1e-12)) { v3set(c->t1, 0, 1, 0); l2 = 1; }
  }
  real l = sqrt(l2);
  for (int i = 0; i < 3; i++) c->t1[i] /= l;
  v3cross(c->t2, c->n, c->t1);
}

/* closest points of the segments p1-q1 and p2-q2 (Ericson, Real-Time Collision Detection 5.1.9) */
static void seg_seg(const real* p1, const real* q1, const real* p2, const real* q2, real* c1, real* c2) {
  real d1[3], d2[3], r[3];
  v3sub(d1, q1, p1); v3sub(d2, q2, p2); v3sub(r, p1, p2);
  real a = v3dot(d1, d1), e = v3dot(d2, d2), f = v3dot(d2, r), cc = v3dot(d1, r), b = v3dot(d1, d2);
  real den = a * e - b * b, sN = 0, tN;
  if (den > (real)1e-12) { sN = (b * f - cc * e) / den; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
  tN = (b * sN + f) / e;
  if (tN < 0) { tN = 0; sN = -cc / a; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
  else if (tN > 1) { tN = 1; sN = (b - cc) / a; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
  for (int i = 0; i < 3; i++) { c1[i] = p1[i] + sN * d1[i]; c2[i] = p2[i] + tN * d2[i]; }
}

#define GO1_SELF_TYPES 6                /* capsule combinations of a pair of legs (detect_contacts) */
#define GO1_SELF_LEG_RADIUS GO1_FOOT_RADIUS
#define GO1_SELF_THIGH_RADIUS 0.017    /* half of the thigh box's larger cross-section side (urdf: 0.0245 x 0.034) */
#define GO1_MAX_TRUNK_POINTS 4
#define GO1_LIMIT_RECOVERY_RATE 10.0   /* rad/s */
#define GO1_LIMIT_SAFETY 2.0           /* x velocity limit */
#define GO1_LIMIT_SLACK 0.2            /* rad beyond a stop */

typedef struct { int n; int dropped[GO1_CC_COUNT]; uint32_t sig[GO1_SIG_WORDS]; } ContactList;

/* sigw: signature word (0 top surface, 1 walls, 2 self pairs), sigb: bit (< 0: the caller records it) */
static void add_contact(Contact* list, ContactList* L, const Contact* c, int sigw, int sigb) {
  if (L->n >= GO1_MAX_CONTACTS) { L->dropped[c->cls]++; L->sig[1] |= 1u << 31; return; }
  list[L->n] = *c;
  contact_frame(&list[L->n]);
  L->n++;
  if (sigb >= 0) L->sig[sigw] |= 1u << sigb;
}
/* item: the kernel's item index of the point (go1_physics.h IT_*): its weight in the geometry hash */
enum { IT_FOOT = 0, IT_FOOTW, IT_CALF1, IT_CALFW, IT_CALF2, IT_THIGH1, IT_THIGHW, IT_THIGH2, IT_HIP1, IT_HIP2, IT_TR0, IT_TR1, IT_TRW };
static void add_terrain(Contact* list, ContactList* L, real cd, const Cand* c, int rep, int dyn, int cls, int top, int sigw, int sigb, int item) {
  if (!c->valid || !(c->phi < cd)) return;
  if (L->n < GO1_MAX_CONTACTS) L->sig[3] += c->tag * (uint32_t)(2 * item + 1) * 2654435761u;
  Contact t;
  t.repA = rep; t.repB = -1; t.dynA = dyn; t.dynB = -1; t.phi = c->phi; t.share = 1; t.cls = cls; t.top = top;
  v3cpy(t.x, c->x); v3cpy(t.n, c->n);
  add_contact(list, L, &t, sigw, sigb);
}

/* fills `list`, returns the bookkeeping (count, drops per class, signature words 0..2 without the limit-row bits) */
static ContactList detect_contacts(const Go1SimConfig* cfg, const Terrain* ter, const Kin* k, const real* base_pos, Contact* list) {
  const real cd = cfg->contact_distance;
  const int walls = cfg->terrain_type != 0 && ter->hs && cfg->hf_wall_units > 0;
  Cand trunk[8], hip[4][2], thigh[4][2], calf[4][2], foot[4], trunkw, thighw[4], calfw[4], footw[4];
  memset(trunk, 0, sizeof trunk); memset(hip, 0, sizeof hip); memset(thigh, 0, sizeof thigh); memset(calf, 0, sizeof calf); memset(foot, 0, sizeof foot);
  memset(&trunkw, 0, sizeof trunkw); memset(thighw, 0, sizeof thighw); memset(calfw, 0, sizeof calfw); memset(footw, 0, sizeof footw);
  for (int m = 0; m < 8; m++) {       /* trunk box corners */
    real l[3] = {(m & 1 ? 1 : -1) * GO1_TRUNK_BOX_HALF[0], (m & 2 ? 1 : -1) * GO1_TRUNK_BOX_HALF[1], (m & 4 ? 1 : -1) * GO1_TRUNK_BOX_HALF[2]};
    candidate(ter, k, base_pos, 0, l, 0, &trunk[m], walls ? &trunkw : NULL, m);
  }
  for (int leg = 0; leg < 4; leg++) {
    int hipb = 1 + 3 * leg;
    for (int m = 0; m < 2; m++) {
      real l[3] = {GO1_HIP_CAPSULE_CENTER[leg][0], GO1_HIP_CAPSULE_CENTER[leg][1] + (m ? 1 : -1) * GO1_HIP_CAPSULE_HALF, GO1_HIP_CAPSULE_CENTER[leg][2]};
      candidate(ter, k, base_pos, hipb, l, GO1_HIP_CAPSULE_RADIUS, &hip[leg][m], NULL, m);
    }
    for (int m = 0; m < 8; m++) {     /* thigh / calf boxes: long axis z -> ends by the sign of z */
      real l[3] = {GO1_THIGH_BOX_CENTER[0] + (m & 1 ? 1 : -1) * GO1_THIGH_BOX_HALF[0],
                   GO1_THIGH_BOX_CENTER[1] + (m & 2 ? 1 : -1) * GO1_THIGH_BOX_HALF[1],
                   GO1_THIGH_BOX_CENTER[2] + (m & 4 ? 1 : -1) * GO1_THIGH_BOX_HALF[2]};
      candidate(ter, k, base_pos, hipb + 1, l, 0, &thigh[leg][(m >> 2) & 1], walls ? &thighw[leg] : NULL, m);
    }
    for (int m = 0; m < 8; m++) {
      real l[3] = {GO1_CALF_BOX_CENTER[0] + (m & 1 ? 1 : -1) * GO1_CALF_BOX_HALF[0],
                   GO1_CALF_BOX_CENTER[1] + (m & 2 ? 1 : -1) * GO1_CALF_BOX_HALF[1],
                   GO1_CALF_BOX_CENTER[2] + (m & 4 ? 1 : -1) * GO1_CALF_BOX_HALF[2]};
      candidate(ter, k, base_pos, hipb + 2, l, 0, &calf[leg][(m >> 2) & 1], walls ? &calfw[leg] : NULL, m);
    }
    real fo[3] = {GO1_FOOT_OFFSET[leg][0], GO1_FOOT_OFFSET[leg][1], GO1_FOOT_OFFSET[leg][2]};
    candidate(ter, k, base_pos, hipb + 2, fo, GO1_FOOT_RADIUS, &foot[leg], walls ? &footw[leg] : NULL, 0);
  }
  ContactList L;
  memset(&L, 0, sizeof L);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &foot[leg], 4 + 4 * leg, 3 * leg + 3, GO1_CC_FOOT, 1, 0, leg, IT_FOOT);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &footw[leg], 4 + 4 * leg, 3 * leg + 3, GO1_CC_FOOT_WALL, 0, 1, leg, IT_FOOTW);
  /* self-collision: segments of the lower legs [0], thighs [1] and hip capsules [2], trunk axis */
  real P[4][3][3], Q[4][3][3], TA[3], TB[3];
  for (int leg = 0; leg < 4; leg++) {
    real fo[3] = {GO1_FOOT_OFFSET[leg][0], GO1_FOOT_OFFSET[leg][1], GO1_FOOT_OFFSET[leg][2]}, w[3];
    v3cpy(P[leg][0], k->p[3 * leg + 3]);
    m3v(w, k->R[3 * leg + 3], fo);
    v3add(Q[leg][0], P[leg][0], w);
    v3cpy(P[leg][1], k->p[3 * leg + 2]);
    v3cpy(Q[leg][1], k->p[3 * leg + 3]);
    for (int m = 0; m < 2; m++) {           /* hip capsule: along the hip body's y axis (the collision shape the terrain sees too) */
      real l[3] = {GO1_HIP_CAPSULE_CENTER[leg][0], GO1_HIP_CAPSULE_CENTER[leg][1] + (m ? 1 : -1) * GO1_HIP_CAPSULE_HALF, GO1_HIP_CAPSULE_CENTER[leg][2]};
      m3v(w, k->R[3 * leg + 1], l);
      v3add(m ? Q[leg][2] : P[leg][2], k->p[3 * leg + 1], w);
    }
  }
  {
    real a = GO1_TRUNK_BOX_HALF[0] - GO1_TRUNK_BOX_HALF[1], la[3] = {-a, 0, 0}, lb[3] = {a, 0, 0};
    m3v(TA, k->R[0], la); m3v(TB, k->R[0], lb);
  }
  /* pid = 6 * type + pair for the leg-leg pairs — type 0 lower-lower, 1 lower(i)-thigh(j), 2 thigh(i)-lower(j), 3 thigh-thigh, 4 hip(i)-lower(j),
   * 5 lower(i)-hip(j) (round 5: the one reachable combination with a hip capsule, tests/test_self_collision_reach.py) — pair (i, j) in
   * the order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3); pid 36 + leg: lower leg against the trunk.  Two legs touch in ONE point: of the six
   * capsule combinations of a pair of legs only the deepest is a contact (ties: the lower type), so at most six leg-leg contacts exist and
   * none is ever dropped for lack of a slot of its own.  Signature word 2: 3 bits per pair (type + 1; bits 0..17), trunk pairs at 18 + leg. */
  static const int PI_[6] = {0, 0, 0, 1, 1, 2}, PJ_[6] = {1, 2, 3, 2, 3, 3};
  static const int SEG_A[GO1_SELF_TYPES] = {0, 0, 1, 1, 2, 0}, SEG_B[GO1_SELF_TYPES] = {0, 1, 0, 1, 0, 2};
  const real SEG_R[3] = {(real)GO1_SELF_LEG_RADIUS, (real)GO1_SELF_THIGH_RADIUS, (real)GO1_HIP_CAPSULE_RADIUS};
  int best_type[6];
  Contact pairc[6 * GO1_SELF_TYPES + 4];
  int pair_on[6 * GO1_SELF_TYPES + 4] = {0};
  for (int pr = 0; pr < 6; pr++) best_type[pr] = -1;
  for (int pid = 0; pid < (cfg->self_collision ? 6 * GO1_SELF_TYPES + 4 : 0); pid++) {
    const int leglegs = 6 * GO1_SELF_TYPES;
    const int type = pid < leglegs ? pid / 6 : 0, i = pid < leglegs ? PI_[pid % 6] : pid - leglegs, j = pid < leglegs ? PJ_[pid % 6] : -1;
    const int sa = SEG_A[type], sb = SEG_B[type];          /* segment of body A / B: 0 lower leg, 1 thigh, 2 hip capsule */
    real c1[3], c2[3], d[3];
    const real ra = SEG_R[sa];
    const real rb = j >= 0 ? SEG_R[sb] : (real)GO1_TRUNK_BOX_HALF[1];
    if (j >= 0) seg_seg(P[i][sa], Q[i][sa], P[j][sb], Q[j][sb], c1, c2); else seg_seg(P[i][0], Q[i][0], TA, TB, c1, c2);
    v3sub(d, c1, c2);
    real dist = v3norm(d);
    if (!(dist > (real)1e-6)) continue;
    real phi = dist - ra - rb;
    if (!(phi < cd)) continue;
    Contact t;
    t.repA = 1 + 4 * i + (2 - sa); t.dynA = 3 * i + (3 - sa);
    t.repB = j >= 0 ? 1 + 4 * j + (2 - sb) : 0; t.dynB = j >= 0 ? 3 * j + (3 - sb) : 0;
    t.phi = phi; t.share = 0; t.cls = GO1_CC_SELF; t.top = 0;
    for (int q = 0; q < 3; q++) { t.n[q] = d[q] / dist; t.x[q] = c2[q] + t.n[q] * (rb + (real)0.5 * phi); }
    pairc[pid] = t;
    if (j < 0) { pair_on[pid] = 1; continue; }
    const int pr = pid % 6;
    if (best_type[pr] < 0 || phi < pairc[6 * best_type[pr] + pr].phi) {
      if (best_type[pr] >= 0) pair_on[6 * best_type[pr] + pr] = 0;
      best_type[pr] = type;
      pair_on[pid] = 1;
    }
  }
  for (int pid = 0; pid < 6 * GO1_SELF_TYPES + 4; pid++)
    if (pair_on[pid]) {
      const int before = L.n;
      add_contact(list, &L, &pairc[pid], 2, -1);
      if (L.n > before) L.sig[2] |= pid < 6 * GO1_SELF_TYPES ? (uint32_t)(pid / 6 + 1) << (3 * (pid % 6)) : 1u << (18 + pid - 6 * GO1_SELF_TYPES);
    }
  /* remaining shapes */
#define FIRST(c) (((c)[1].valid && (!(c)[0].valid || (c)[1].phi < (c)[0].phi)) ? 1 : 0)
  {
    int listed = 0;
    for (int m = 0; m < 8; m++) {
      if (!(trunk[m].valid && trunk[m].phi < cd)) continue;
      if (listed >= GO1_MAX_TRUNK_POINTS) { L.dropped[GO1_CC_TRUNK]++; L.sig[1] |= 1u << 31; continue; }
      listed++;
      add_terrain(list, &L, cd, &trunk[m], 0, 0, GO1_CC_TRUNK, 1, 0, 4 + m, IT_TR0 + (m & 1));
    }
  }
  add_terrain(list, &L, cd, &trunkw, 0, 0, GO1_CC_WALL, 0, 1, 4, IT_TRW);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &calf[leg][FIRST(calf[leg])], 3 + 4 * leg, 3 * leg + 3, GO1_CC_CALF, 1, 0, 12 + 2 * leg + FIRST(calf[leg]), IT_CALF1);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &calfw[leg], 3 + 4 * leg, 3 * leg + 3, GO1_CC_WALL, 0, 1, 5 + leg, IT_CALFW);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &calf[leg][1 - FIRST(calf[leg])], 3 + 4 * leg, 3 * leg + 3, GO1_CC_CALF, 1, 0, 12 + 2 * leg + 1 - FIRST(calf[leg]), IT_CALF2);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &thigh[leg][FIRST(thigh[leg])], 2 + 4 * leg, 3 * leg + 2, GO1_CC_THIGH, 1, 0, 20 + 2 * leg + FIRST(thigh[leg]), IT_THIGH1);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &thighw[leg], 2 + 4 * leg, 3 * leg + 2, GO1_CC_WALL, 0, 1, 9 + leg, IT_THIGHW);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &thigh[leg][1 - FIRST(thigh[leg])], 2 + 4 * leg, 3 * leg + 2, GO1_CC_THIGH, 1, 0, 20 + 2 * leg + 1 - FIRST(thigh[leg]), IT_THIGH2);
  /* (hip points: their signature bits sit in word 1 behind the wall bits) */
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &hip[leg][FIRST(hip[leg])], 1 + 4 * leg, 3 * leg + 1, GO1_CC_HIP, 1, 1, 13 + 2 * leg + FIRST(hip[leg]), IT_HIP1);
  for (int leg = 0; leg < 4; leg++) add_terrain(list, &L, cd, &hip[leg][1 - FIRST(hip[leg])], 1 + 4 * leg, 3 * leg + 1, GO1_CC_HIP, 1, 1, 13 + 2 * leg + 1 - FIRST(hip[leg]), IT_HIP2);
#undef FIRST
  /* warm start: a body's previous impulse is shared equally by its listed top-surface points; wall and body-body points
   * start from zero */
  int cnt[17] = {0};
  for (int c = 0; c < L.n; c++) if (list[c].top) cnt[list[c].repA]++;
  for (int c = 0; c < L.n; c++) list[c].share = list[c].top ? (real)1.0 / cnt[list[c].repA] : 0;
  return L;
}

/* Jacobian row for world direction d at point x (relative to base origin) on dynamic body dynb */
static void jac_row(const Kin* k, int dynb, const real* x, const real* d, real* J) {
  memset(J, 0, NV * sizeof(real));
  real t[3];
  v3cross(t, x, d);
  for (int i = 0; i < 3; i++) { J[i] = t[i]; J[3 + i] = d[i]; }
  if (dynb == 0) return;
  int leg = (dynb - 1) / 3, depth = (dynb - 1) % 3;
  for (int j = 0; j <= depth; j++) {
    int ji = 3 * leg + j, b = ji + 1;
    real r[3];
    v3sub(r, x, k->p[b]);
    v3cross(t, r, d);
    J[6 + ji] = v3dot(k->axis[ji], t);
  }
}

/* sweep order.  3 = THE CONTRACT since round 5: trunk and body-body contacts in list order, then the terrain contacts of the four legs SIDE BY
 * SIDE (Gauss-Seidel inside a leg, block Jacobi over legs), the hip and thigh rows with MASS SPLITTING — what the kernel runs
 * (csrc/go1_physics.h "SWEEP ORDER").  Comparison arms of tools/solver_order_study.py and tests/test_emu_parity.py: 0 = every contact in list
 * order (the contract of rounds 1-4); 1 = side by side without splitting (round 4's study build: block Jacobi over hip / thigh contacts, which
 * couple to the base through one or two joints, does not settle on a robot lying on its side); 2 = only the lower-leg contacts side by side,
 * hip / thigh in list order (settles, but no faster than the list order on the hardware: every wavefront holds a fallen robot whose
 * cooperative turns the other fifteen wait for); 4 = every row split (slows the convergence of walking robots).
 * Numbers: profiles/r05_solver_order_study.txt */
static int g_solver_legs_parallel = 3;
void go1_oracle_set_solver_order(int legs_parallel) { g_solver_legs_parallel = legs_parallel; }

/* STUDY option, never the contract (tools/solver_tgs_study.py): a TGS-like solve — what the reference's PhysX setting solver_type 1 with 4
 * position iterations does differently from a plain PGS (legged_robot_config.py:410-414; Macklin et al. 2019, "Small steps in physics
 * simulation", restated for this velocity-level solve with frozen Jacobians).  The substep h is cut into N = solver_iterations parts of
 * hs = h / N; sweep k aims each contact's normal rate at the error that is left,  -(phi + sum_{j<k} hs un_j) / hs  (clamped like the plain
 * target; a restitution target is kept), i.e. the constraint errors are re-evaluated between the sweeps and resolved over hs instead of h.
 * 1: targets only, the pose is integrated with the final velocity over h as in the contract;  2: the pose is also advanced by hs with every
 * sweep's velocity (the TGS "stepping").  0 = off. */
static int g_tgs_like = 0;
void go1_oracle_set_tgs_like(int mode) { g_tgs_like = mode; }

/* ------------------------------------------------------------------ one physics substep (replaces gym.simulate) */
typedef struct { real force[17][3]; int dropped[GO1_CC_COUNT]; uint32_t sig[GO1_SIG_WORDS]; } ContactOut;

/* wl[b]: world impulse vector (x,y,z) of body b's contact in the previous substep (warm start), updated in place */
static void physics_substep(const Go1SimConfig* cfg, const Terrain* ter, Phys* s, const real* tau, const real* grav,
                            real wl[17][3], int use_warm, ContactOut* out) {
  const real h = (real)cfg->sim_dt;
  Kin k;
  kinematics(s, &k);
  real vel[NV], zero[NV] = {0}, bias[NV];
  for (int i = 0; i < 3; i++) { vel[i] = s->vang[i]; vel[3 + i] = s->vlin[i]; }
  for (int j = 0; j < 12; j++) vel[6 + j] = s->qd[j];
  rnea(&k, vel, zero, grav, bias);
  real M[NV * NV];
  for (int c = 0; c < NV; c++) {
    real e[NV] = {0}, col[NV];
    e[c] = 1;
    rnea(&k, NULL, e, NULL, col);
    for (int r = 0; r < NV; r++) M[r * NV + c] = col[r];
  }
  for (int r = 0; r < NV; r++)
    for (int c = r + 1; c < NV; c++) { real m = 0.5 * (M[r * NV + c] + M[c * NV + r]); M[r * NV + c] = M[c * NV + r] = m; }
  real L[NV * NV];
  memcpy(L, M, sizeof M);
  cholesky(L, NV);
  real acc[NV];
  for (int i = 0; i < 6; i++) acc[i] = -bias[i];
  for (int j = 0; j < 12; j++) acc[6 + j] = tau[j] - bias[6 + j];
  chol_solve(L, NV, acc);
  real v[NV];
  for (int i = 0; i < NV; i++) v[i] = vel[i] + h * acc[i];

  /* contacts at the start-of-step configuration */
  Contact C[GO1_MAX_CONTACTS];
  const ContactList CL = detect_contacts(cfg, ter, &k, s->pos, C);
  const int nc = CL.n;
  if (out) { memcpy(out->dropped, CL.dropped, sizeof CL.dropped); memcpy(out->sig, CL.sig, sizeof CL.sig); }
  /* Joint limits are solver rows, one per joint (generalised impulse along the joint coordinate: equal and opposite on
   * child and parent, so an actuator pushing against a stop or against the velocity limit cannot create net momentum).
   * Row j constrains the joint rate to [vlo, vhi] = [max((lo-q)/h, -vmax), min((hi-q)/h, vmax)]; it enters the solve when
   * the free rate comes within joint_limit_margin of that band. */
  real TJ[12][NV], AJ[12], vlo[12], vhi[12], lamj[12];
  int jact[12];
  for (int j = 0; j < 12; j++) {
    real lo = (GO1_JOINT_LOWER[j] - s->q[j]) / h, hi = (GO1_JOINT_UPPER[j] - s->q[j]) / h, vl = GO1_JOINT_VEL_LIMIT[j];
    if (lo > GO1_LIMIT_RECOVERY_RATE) lo = GO1_LIMIT_RECOVERY_RATE;        /* a joint found beyond a stop is brought back at a bounded rate */
    if (hi < -GO1_LIMIT_RECOVERY_RATE) hi = -GO1_LIMIT_RECOVERY_RATE;
    vlo[j] = lo > -vl ? lo : -vl;
    vhi[j] = hi < vl ? hi : vl;
    lamj[j] = 0;
    const real mv = (real)cfg->joint_limit_margin, mp = (real)cfg->joint_limit_pos_margin / h;
    jact[j] = !(v[6 + j] > lo + mp && v[6 + j] < hi - mp && v[6 + j] > -vl + mv && v[6 + j] < vl - mv);
  }
  for (int leg = 0; leg < 4; leg++) {        /* the joints of a leg are strongly coupled: one active row brings in the leg's other two */
    int any = jact[3 * leg] || jact[3 * leg + 1] || jact[3 * leg + 2];
    jact[3 * leg] = jact[3 * leg + 1] = jact[3 * leg + 2] = any;
    if (any && out) out->sig[2] |= 1u << (28 + leg);
  }
  for (int j = 0; j < 12; j++) {
    if (!jact[j]) continue;
    for (int i = 0; i < NV; i++) TJ[j][i] = 0;
    TJ[j][6 + j] = 1;
    chol_solve(L, NV, TJ[j]);
    AJ[j] = TJ[j][6 + j];
  }
  real J[GO1_MAX_CONTACTS][3][NV], T[GO1_MAX_CONTACTS][3][NV], A[GO1_MAX_CONTACTS][3], vstar[GO1_MAX_CONTACTS], cmu[GO1_MAX_CONTACTS], cmud[GO1_MAX_CONTACTS];
  real lamc[GO1_MAX_CONTACTS][3];
  /* orders 3 / 4 (mass splitting): TL = response of the own leg's joints with the base held, TS / AS = the split response / diagonal */
  static real TL[GO1_MAX_CONTACTS][3][NV], TS[GO1_MAX_CONTACTS][3][NV], AS[GO1_MAX_CONTACTS][3];
#pragma omp threadprivate(TL, TS, AS)
  real Lll[12 * 12];
  if (g_solver_legs_parallel >= 3) {
    for (int r = 0; r < 12; r++) for (int cc = 0; cc < 12; cc++) Lll[r * 12 + cc] = M[(6 + r) * NV + 6 + cc];
    cholesky(Lll, 12);
  }
  for (int c = 0; c < nc; c++) {
    const real* dirs[3] = {C[c].n, C[c].t1, C[c].t2};
    for (int r = 0; r < 3; r++) {
      jac_row(&k, C[c].dynA, C[c].x, dirs[r], J[c][r]);
      if (C[c].dynB >= 0) {                       /* body-body contact: +d on A, -d on B */
        real JB[NV];
        jac_row(&k, C[c].dynB, C[c].x, dirs[r], JB);
        for (int i = 0; i < NV; i++) J[c][r][i] -= JB[i];
      }
      memcpy(T[c][r], J[c][r], sizeof(real) * NV);
      chol_solve(L, NV, T[c][r]);
      real a = 0;
      for (int i = 0; i < NV; i++) a += J[c][r][i] * T[c][r][i];
      A[c][r] = a;
      if (g_solver_legs_parallel >= 3) {
        real x[12];
        for (int i = 0; i < 12; i++) x[i] = J[c][r][6 + i];
        chol_solve(Lll, 12, x);
        for (int i = 0; i < 6; i++) TL[c][r][i] = 0;
        for (int i = 0; i < 12; i++) TL[c][r][6 + i] = x[i];
      }
    }
    /* PhysX default combine mode: average of the two materials (robot-robot: the robot's own) */
    const int self = C[c].repB >= 0;
    cmu[c] = self ? s->mu : (real)0.5 * (s->mu + (real)cfg->terrain_friction);
    cmud[c] = self ? s->mu : (real)0.5 * (s->mu + (real)cfg->terrain_dynamic_friction);      /* cone of a sliding contact */
    if (cmud[c] > cmu[c]) cmud[c] = cmu[c];                                                      /* (never wider than the static cone) */
    const real e_c = self ? s->rest : (real)0.5 * (s->rest + (real)cfg->terrain_restitution);
    real vs = -C[c].phi / h;
    if (vs > cfg->max_depenetration_velocity) vs = cfg->max_depenetration_velocity;
    real un_pre = 0;
    for (int i = 0; i < NV; i++) un_pre += J[c][0][i] * vel[i];
    if (un_pre < -(real)cfg->bounce_threshold_velocity && -e_c * un_pre > vs) {
      vs = -e_c * un_pre;
      if (out) out->sig[3] += (uint32_t)(c + 1) * 0x9E3779B1u;       /* signature: this contact took the restitution branch */
    }
    vstar[c] = vs;
    /* the body's last impulse, shared by its listed points and projected on the current contact frame */
    const real* w = wl[C[c].repA];
    const real sh = use_warm ? C[c].share : 0;
    lamc[c][0] = sh * v3dot(w, C[c].n); lamc[c][1] = sh * v3dot(w, C[c].t1); lamc[c][2] = sh * v3dot(w, C[c].t2);
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < NV; i++) v[i] += T[c][r][i] * lamc[c][r];
  }
  int slid[GO1_MAX_CONTACTS];
  for (int c = 0; c < nc; c++) slid[c] = 0;
  /* (study) TGS-like: normal displacement accumulated over the sweeps, the plain target (kept where it is a restitution target), the pose */
  real tgs_delta[GO1_MAX_CONTACTS], tgs_plain[GO1_MAX_CONTACTS], tgs_x[NV];
  const real tgs_hs = h / (cfg->solver_iterations > 0 ? cfg->solver_iterations : 1);
  for (int c = 0; c < nc; c++) {
    real vs = -C[c].phi / h;
    if (vs > cfg->max_depenetration_velocity) vs = cfg->max_depenetration_velocity;
    tgs_plain[c] = vs;
    tgs_delta[c] = 0;
  }
  for (int i = 0; i < NV; i++) tgs_x[i] = 0;
  for (int it = 0; it < cfg->solver_iterations; it++) {
    if (g_tgs_like)
      for (int c = 0; c < nc; c++) {
        if (vstar[c] != tgs_plain[c] && it == 0) tgs_plain[c] = (real)-1e30;      /* marks a restitution target: left alone */
        if (tgs_plain[c] == (real)-1e30) continue;
        real vs = -(C[c].phi + tgs_delta[c]) / tgs_hs;
        if (vs > cfg->max_depenetration_velocity) vs = cfg->max_depenetration_velocity;
        vstar[c] = vs;
      }
    /* one contact's update on the velocity vector vv (projected Gauss-Seidel step: normal row, then the two tangent rows on the cone) */
#define CONTACT_UPDATE(c, vv) CONTACT_UPDATE_X(c, vv, T, A)
#define CONTACT_UPDATE_X(c, vv, T, A) do { \
      real un = 0; \
      for (int i = 0; i < NV; i++) un += J[c][0][i] * (vv)[i]; \
      real ln = lamc[c][0] - (un - vstar[c]) / A[c][0]; \
      if (ln < 0) ln = 0; \
      real dl = ln - lamc[c][0]; \
      lamc[c][0] = ln; \
      for (int i = 0; i < NV; i++) (vv)[i] += T[c][0][i] * dl; \
      real u1 = 0, u2 = 0; \
      for (int i = 0; i < NV; i++) { u1 += J[c][1][i] * (vv)[i]; u2 += J[c][2][i] * (vv)[i]; } \
      real l1 = lamc[c][1] - u1 / A[c][1], l2 = lamc[c][2] - u2 / A[c][2]; \
      /* Coulomb cone: a tangential impulse inside the static cone sticks, beyond it the contact slides on the dynamic cone \
       * (PhysX: static / dynamic friction of the material pair) */ \
      real lim = cmu[c] * ln, nrm = sqrt(l1 * l1 + l2 * l2); \
      slid[c] = nrm > lim; \
      if (nrm > lim) { real sc = (nrm > 0) ? cmud[c] * ln / nrm : 0; l1 *= sc; l2 *= sc; } \
      real d1 = l1 - lamc[c][1], d2 = l2 - lamc[c][2]; \
      lamc[c][1] = l1; lamc[c][2] = l2; \
      for (int i = 0; i < NV; i++) (vv)[i] += T[c][1][i] * d1 + T[c][2][i] * d2; \
    } while (0)
    if (!g_solver_legs_parallel) {
      for (int c = 0; c < nc; c++) CONTACT_UPDATE(c, v);
    } else if (g_solver_legs_parallel >= 3) {
      /* THE CONTRACT (order 3): all terrain contacts of a leg side by side with MASS SPLITTING — in the leg phase the base answers a split
       * row's impulse n times as strongly as it really does (n sub-bodies of 1/n of its articulated inertia, one per leg that holds split
       * rows), which makes block Jacobi over legs convergent whatever the coupling; the true response of all impulse changes is applied
       * when the legs meet.  3: the hip / thigh rows are split (lower-leg rows plain Jacobi); 4 (study): every row */
      for (int c = 0; c < nc; c++)
        if (C[c].dynA == 0 || C[c].dynB >= 0) CONTACT_UPDATE(c, v);
      int legs_on = 0;
      for (int leg = 0; leg < 4; leg++) {
        int any = 0;
        for (int c = 0; c < nc; c++)
          if (C[c].dynB < 0 && C[c].dynA > 0 && (C[c].dynA - 1) / 3 == leg && (g_solver_legs_parallel == 4 || (C[c].dynA - 1) % 3 < 2)) any = 1;
        legs_on += any;
      }
      const real nsplit = legs_on > 1 ? legs_on : 1;
      for (int c = 0; c < nc; c++) {
        if (!(C[c].dynB < 0 && C[c].dynA > 0)) continue;
        const int split = g_solver_legs_parallel == 4 || (C[c].dynA - 1) % 3 < 2;
        for (int r = 0; r < 3; r++) {
          real a = 0;
          for (int i = 0; i < NV; i++) {
            TS[c][r][i] = split ? TL[c][r][i] + nsplit * (T[c][r][i] - TL[c][r][i]) : T[c][r][i];
            a += J[c][r][i] * TS[c][r][i];
          }
          AS[c][r] = a;
        }
      }
      real v0[NV], lam0[GO1_MAX_CONTACTS][3];
      memcpy(v0, v, sizeof v0);
      memcpy(lam0, lamc, sizeof lam0);
      for (int leg = 0; leg < 4; leg++) {
        real vl[NV];
        memcpy(vl, v0, sizeof vl);
        for (int c = 0; c < nc; c++)
          if (C[c].dynB < 0 && C[c].dynA > 0 && (C[c].dynA - 1) / 3 == leg) CONTACT_UPDATE_X(c, vl, TS, AS);
      }
      for (int c = 0; c < nc; c++)
        if (C[c].dynB < 0 && C[c].dynA > 0)
          for (int r = 0; r < 3; r++)
            for (int i = 0; i < NV; i++) v[i] += T[c][r][i] * (lamc[c][r] - lam0[c][r]);
    } else {
      /* study orders 1 / 2: (2: trunk, hip, thigh and) body-body contacts in list order, then the (2: lower-leg) contacts of the four legs
       * SIDE BY SIDE — Gauss-Seidel inside a leg, every leg starting from the same velocity, the legs' velocity changes added up */
#define LEG_PARALLEL(c) (C[c].dynB < 0 && C[c].dynA > 0 && (g_solver_legs_parallel == 1 || (C[c].dynA - 1) % 3 == 2))
      for (int c = 0; c < nc; c++)
        if (!LEG_PARALLEL(c)) CONTACT_UPDATE(c, v);
      real v0[NV], dv[NV];
      memcpy(v0, v, sizeof v0);
      for (int i = 0; i < NV; i++) dv[i] = 0;
      for (int leg = 0; leg < 4; leg++) {
        real vl[NV];
        memcpy(vl, v0, sizeof vl);
        for (int c = 0; c < nc; c++)
          if (LEG_PARALLEL(c) && (C[c].dynA - 1) / 3 == leg) CONTACT_UPDATE(c, vl);
        for (int i = 0; i < NV; i++) dv[i] += vl[i] - v0[i];
      }
      for (int i = 0; i < NV; i++) v[i] = v0[i] + dv[i];
    }
#undef CONTACT_UPDATE
    /* joint rows, plain Gauss-Seidel in joint order.  (Running the four legs side by side — block Jacobi — was tried and
     * does NOT converge: with the robot in the air the legs couple strongly through the light base.) */
    for (int j = 0; j < 12; j++) {
      if (!jact[j]) continue;
      real u0 = v[6 + j] - AJ[j] * lamj[j];                     /* rate without this row's impulse */
      real ut = u0 < vlo[j] ? vlo[j] : (u0 > vhi[j] ? vhi[j] : u0);
      real ln = (ut - u0) / AJ[j];
      real dl = ln - lamj[j];
      lamj[j] = ln;
      for (int i = 0; i < NV; i++) v[i] += TJ[j][i] * dl;
    }
    if (g_tgs_like) {
      for (int c = 0; c < nc; c++) {
        real un = 0;
        for (int i = 0; i < NV; i++) un += J[c][0][i] * v[i];
        tgs_delta[c] += tgs_hs * un;
      }
      for (int i = 0; i < NV; i++) tgs_x[i] += tgs_hs * v[i];
    }
    if (out) {
      /* signature: the active set after EVERY sweep (pressing contacts, contacts projected on the cone in this sweep, limit rows
       * carrying an impulse), weighted by the sweep: a projection that flips in an intermediate sweep sends the unconverged
       * 4-sweep iterate down another path even when the final active sets coincide */
      uint32_t jm = 0;
      for (int j = 0; j < 12; j++) if (jact[j] && lamj[j] != 0) jm |= 1u << j;
      uint32_t ah = jm * 0x27D4EB2Fu;
      for (int c = 0; c < nc; c++) {
        if (lamc[c][0] > 0) ah += (uint32_t)(c + 1) * 0x85EBCA6Bu;
        if (slid[c]) ah += (uint32_t)(c + 1) * 0xC2B2AE35u;
      }
      out->sig[3] += ah * (uint32_t)(2 * it + 1);
    }
  }
  for (int b = 0; b < 17; b++) v3set(wl[b], 0, 0, 0);
  for (int c = 0; c < nc; c++)
    for (int i = 0; i < 3; i++) {
      real f = C[c].n[i] * lamc[c][0] + C[c].t1[i] * lamc[c][1] + C[c].t2[i] * lamc[c][2];
      wl[C[c].repA][i] += f;
      if (C[c].repB >= 0) wl[C[c].repB][i] -= f;
    }
  for (int b = 0; b < 17; b++)
    for (int i = 0; i < 3; i++) out->force[b][i] = wl[b][i] / h;
  /* The limit rows leave at most the solver's residual; it is NOT clamped away (a clamp on the joint coordinate alone is
   * an unbalanced impulse: it breaks momentum conservation, which a learning policy turns into free thrust).  Only a
   * solver failure far outside the admissible band is cut, at GO1_LIMIT_SAFETY x the limit. */
  for (int j = 0; j < 12; j++) {
    real vl = GO1_LIMIT_SAFETY * GO1_JOINT_VEL_LIMIT[j];
    if (v[6 + j] > vl) v[6 + j] = vl;
    if (v[6 + j] < -vl) v[6 + j] = -vl;
  }
  {   /* Cfg.asset.max_angular_velocity / max_linear_velocity: magnitude caps on the base twist */
    real wn2 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), vn2 = sqrt(v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (wn2 > cfg->max_angular_velocity) for (int i = 0; i < 3; i++) v[i] *= cfg->max_angular_velocity / wn2;
    if (vn2 > cfg->max_linear_velocity) for (int i = 0; i < 3; i++) v[3 + i] *= cfg->max_linear_velocity / vn2;
  }
  real xd[NV];          /* displacement of the substep: h v (the contract), or (study, TGS-like 2) the sum of the sweeps' hs v_k */
  for (int i = 0; i < NV; i++) xd[i] = g_tgs_like == 2 ? tgs_x[i] : h * v[i];
  for (int i = 0; i < 3; i++) { s->vang[i] = v[i]; s->vlin[i] = v[3 + i]; s->pos[i] += xd[3 + i]; }
  real wrot[3] = {s->vang[0], s->vang[1], s->vang[2]};       /* mean angular rate of the substep */
  if (g_tgs_like == 2) for (int i = 0; i < 3; i++) wrot[i] = xd[i] / h;
  real wn = v3norm(wrot);
  if (wn > 1e-12) {
    real half = 0.5 * wn * h, sn = sin(half) / wn;
    real dq[4] = {wrot[0] * sn, wrot[1] * sn, wrot[2] * sn, cos(half)}, qn[4];
    quat_mul(qn, dq, s->quat);
    real l = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 4; i++) s->quat[i] = qn[i] / l;
  }
  for (int j = 0; j < 12; j++) {
    s->qd[j] = v[6 + j];
    s->q[j] += g_tgs_like == 2 ? xd[6 + j] : h * s->qd[j];
    if (s->q[j] < GO1_JOINT_LOWER[j] - GO1_LIMIT_SLACK) { s->q[j] = GO1_JOINT_LOWER[j] - GO1_LIMIT_SLACK; if (s->qd[j] < 0) s->qd[j] = 0; }
    if (s->q[j] > GO1_JOINT_UPPER[j] + GO1_LIMIT_SLACK) { s->q[j] = GO1_JOINT_UPPER[j] + GO1_LIMIT_SLACK; if (s->qd[j] > 0) s->qd[j] = 0; }
  }
}

/* exported for the invariant tests: mass matrix, bias and free acceleration at a state */
void go1_oracle_dynamics(const double* root13, const double* q, const double* qd, const double* tau, const double* grav_d,
                         double payload, const double* com_disp, double* M_out, double* bias_out, double* acc_out) {
  Phys s;
  for (int i = 0; i < 3; i++) { s.pos[i] = (real)root13[i]; s.vlin[i] = (real)root13[7 + i]; s.vang[i] = (real)root13[10 + i]; }
  for (int i = 0; i < 4; i++) s.quat[i] = (real)root13[3 + i];
  for (int j = 0; j < 12; j++) { s.q[j] = (real)q[j]; s.qd[j] = (real)qd[j]; }
  s.mass0 = (real)(GO1_BODY_MASS[0] + payload);
  v3set(s.com0, (real)com_disp[0], (real)com_disp[1], (real)com_disp[2]);
  s.mu = 1; s.rest = 0;
  real grav[3] = {(real)grav_d[0], (real)grav_d[1], (real)grav_d[2]};
  Kin k;
  kinematics(&s, &k);
  real vel[NV], zero[NV] = {0}, bias[NV], M[NV * NV];
  for (int i = 0; i < 3; i++) { vel[i] = s.vang[i]; vel[3 + i] = s.vlin[i]; }
  for (int j = 0; j < 12; j++) vel[6 + j] = s.qd[j];
  rnea(&k, vel, zero, grav, bias);
  for (int c = 0; c < NV; c++) {
    real e[NV] = {0}, col[NV];
    e[c] = 1;
    rnea(&k, NULL, e, NULL, col);
    for (int r = 0; r < NV; r++) M[r * NV + c] = col[r];
  }
  for (int i = 0; i < NV * NV; i++) M_out[i] = M[i];
  for (int i = 0; i < NV; i++) bias_out[i] = bias[i];
  real L[NV * NV], acc[NV];
  memcpy(L, M, sizeof M);
  cholesky(L, NV);
  for (int i = 0; i < 6; i++) acc[i] = -bias[i];
  for (int j = 0; j < 12; j++) acc[6 + j] = (real)tau[j] - bias[6 + j];
  chol_solve(L, NV, acc);
  for (int i = 0; i < NV; i++) acc_out[i] = acc[i];
}

/* ------------------------------------------------------------------ torque model (legged_robot.py:907-946) */
static real softsign(real x) { return x / (1 + fabs(x)); }
static real actuator_net(const real* in6) {
  real h0[32], h1[32];
  for (int i = 0; i < 32; i++) {
    real a = GO1_ACT_B0[i];
    for (int k = 0; k < 6; k++) a += (real)GO1_ACT_W0[i][k] * in6[k];
    h0[i] = softsign(a);
  }
  for (int i = 0; i < 32; i++) {
    real a = GO1_ACT_B1[i];
    for (int k = 0; k < 32; k++) a += (real)GO1_ACT_W1[i][k] * h0[k];
    h1[i] = softsign(a);
  }
  real o = GO1_ACT_B2;
  for (int k = 0; k < 32; k++) o += (real)GO1_ACT_W2[k] * h1[k];
  return o;
}
void go1_oracle_actuator_net(const double* in6, int n, double* out) {
  for (int i = 0; i < n; i++) {
    real x[6];
    for (int kx = 0; kx < 6; kx++) x[kx] = (real)in6[6 * i + kx];
    out[i] = actuator_net(x);
  }
}

#define AT(buf, c, e) ((buf)[(size_t)(c) * N + (e)])

static void compute_torques(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int* lag_head, int advance_head,
                            const real* q, const real* qd, real* tau) {
  const int N = cfg->num_envs;
  const int nl = cfg->lag_timesteps + 1;
  int head = *lag_head;
  for (int j = 0; j < 12; j++) {
    real a = (real)AT(B->actions, j, e) * (real)cfg->action_scale;
    if (j % 3 == 0) a *= (real)cfg->hip_scale_reduction;
    real target;
    if (cfg->use_lag) {
      B->lag_buffer[((size_t)head * 12 + j) * N + e] = (float)a;          /* overwrite the oldest entry */
      int h2 = (head + 1) % nl;
      target = (real)B->lag_buffer[((size_t)h2 * 12 + j) * N + e] + (real)cfg->default_dof_pos[j];
    } else {
      target = a + (real)cfg->default_dof_pos[j];
    }
    AT(B->joint_pos_target, j, e) = (float)target;
    target = (real)(float)target;
    real t;
    if (cfg->control_type == 1) {
      real err = q[j] - target + (real)AT(B->motor_offsets, j, e);
      real in6[6] = {err, AT(B->joint_pos_err_last, j, e), AT(B->joint_pos_err_last_last, j, e),
                     qd[j], AT(B->joint_vel_last, j, e), AT(B->joint_vel_last_last, j, e)};
      t = actuator_net(in6);
      AT(B->joint_pos_err_last_last, j, e) = AT(B->joint_pos_err_last, j, e);
      AT(B->joint_pos_err_last, j, e) = (float)err;
      AT(B->joint_vel_last_last, j, e) = AT(B->joint_vel_last, j, e);
      AT(B->joint_vel_last, j, e) = (float)qd[j];
    } else {
      t = (real)cfg->kp * (real)AT(B->Kp_factors, j, e) * (target - q[j] + (real)AT(B->motor_offsets, j, e))
          - (real)cfg->kd * (real)AT(B->Kd_factors, j, e) * qd[j];
    }
    t *= (real)AT(B->motor_strengths, j, e);
    real lim = cfg->torque_limits[j];
    if (t > lim) t = lim;
    if (t < -lim) t = -lim;
    tau[j] = t;
    AT(B->torques, j, e) = (float)t;
  }
  if (advance_head) *lag_head = (head + 1) % nl;
}

/* ------------------------------------------------------------------ state <-> buffers */
static void load_phys(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, Phys* s) {
  const int N = cfg->num_envs;
  for (int i = 0; i < 3; i++) { s->pos[i] = AT(B->root_states, i, e); s->vlin[i] = AT(B->root_states, 7 + i, e); s->vang[i] = AT(B->root_states, 10 + i, e); }
  for (int i = 0; i < 4; i++) s->quat[i] = AT(B->root_states, 3 + i, e);
  for (int j = 0; j < 12; j++) { s->q[j] = AT(B->dof_pos, j, e); s->qd[j] = AT(B->dof_vel, j, e); }
  s->mass0 = GO1_BODY_MASS[0] + (real)B->payloads[e];
  for (int i = 0; i < 3; i++) s->com0[i] = AT(B->com_displacements, i, e);
  s->mu = B->friction_coeffs[e];
  s->rest = B->restitutions[e];
}
static void store_phys(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, const Phys* s) {
  const int N = cfg->num_envs;
  for (int i = 0; i < 3; i++) { AT(B->root_states, i, e) = (float)s->pos[i]; AT(B->root_states, 7 + i, e) = (float)s->vlin[i]; AT(B->root_states, 10 + i, e) = (float)s->vang[i]; }
  for (int i = 0; i < 4; i++) AT(B->root_states, 3 + i, e) = (float)s->quat[i];
  for (int j = 0; j < 12; j++) { AT(B->dof_pos, j, e) = (float)s->q[j]; AT(B->dof_vel, j, e) = (float)s->qd[j]; }
}

/* gravity acting during the policy step whose pre-increment common_step_counter is t
 * (legged_robot.py:546-561,701-705): a random offset drawn every gravity_rand_interval steps is
 * active for gravity_rand_duration steps, then zero.  Global (one draw for all envs, quirk D11). */
static void gravity_at(const Go1SimConfig* cfg, int64_t t, real* g) {
  for (int i = 0; i < 3; i++) g[i] = cfg->gravity[i];
  if (!cfg->randomize_gravity) return;
  int64_t epoch = t / cfg->gravity_rand_interval, ph = t % cfg->gravity_rand_interval;
  if (ph >= cfg->gravity_rand_duration) return;
  for (int i = 0; i < 3; i++) {
    real u = rng_uniform(cfg, 0xFFFFFFFFu, epoch, P_GRAVITY, i);
    g[i] += u * ((real)cfg->gravity_range[1] - (real)cfg->gravity_range[0]) + (real)cfg->gravity_range[0];
  }
}

static void feet_state(const Phys* s, real fp[4][3], real fv[4][3]) {
  Kin k;
  kinematics(s, &k);
  for (int leg = 0; leg < 4; leg++) {
    int b = 3 + 3 * leg;
    real fo[3] = {GO1_FOOT_OFFSET[leg][0], GO1_FOOT_OFFSET[leg][1], GO1_FOOT_OFFSET[leg][2]}, w[3], x[3];
    m3v(w, k.R[b], fo);
    v3add(x, k.p[b], w);
    for (int i = 0; i < 3; i++) fp[leg][i] = s->pos[i] + x[i];
    /* velocity: v_O + omega x r + sum_j (axis_j x (x - p_j)) qd_j */
    real v[3], t[3];
    v3cross(t, s->vang, x);
    v3add(v, s->vlin, t);
    for (int j = 0; j < 3; j++) {
      int ji = 3 * leg + j;
      real r[3];
      v3sub(r, x, k.p[ji + 1]);
      v3cross(t, k.axis[ji], r);
      v3axpy(v, s->qd[ji], t);
    }
    v3cpy(fv[leg], v);
  }
}

/* ------------------------------------------------------------------ commands (device-curriculum semantics) */
/* episode_log is a float sum over the envs that reset in a step: order-dependent.  When post-physics runs in
 * parallel (go1_oracle_step) every resetting env parks its terms here and they are added serially in env order,
 * which reproduces the serial result bit for bit. */
static float* g_log_defer = NULL;
static int g_log_stride = 0;

static real fmod1(real x) { real r = fmod(x, 1.0); if (r < 0) r += 1.0; return r; }   /* torch `%` / remainder */

/* slot_step: the policy step the bookkeeping belongs to (in-step calls: the step in progress; reset_idx between steps: the next
 * one) — its successes go to slot slot_step % curriculum_update_interval of curriculum_success */
static void resample_commands(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int64_t step, uint32_t purpose, int64_t slot_step) {
  const int N = cfg->num_envs;
  if (!cfg->device_curriculum) { B->resample_flags[e] |= (purpose == P_CMD_CB) ? 1 : 2; goto clear; }
  {
    uint32_t eg = (uint32_t)(cfg->env_id_offset + e);
    int ep_len = cfg->max_episode_length < cfg->resample_interval ? cfg->max_episode_length : cfg->resample_interval;
    /* curriculum success bookkeeping (curriculum.py:135-154, legged_robot.py:718-739) */
    int ok = cfg->curriculum_keys != 0;
    for (int kx = 0; kx < 4; kx++) {
      if (!(cfg->curriculum_keys & (1 << kx))) continue;
      float val = AT(B->command_sums, cfg->curriculum_sum_index[kx], e) / (float)ep_len;
      if (!(val > cfg->curriculum_threshold[kx])) ok = 0;
    }
    int cat_old = B->env_command_categories[e], bin_old = B->env_command_bins[e];
    if (ok) {
#pragma omp atomic
      B->curriculum_success[((size_t)(slot_step % cfg->curriculum_update_interval) * cfg->num_categories + cat_old) * cfg->num_bins + bin_old] += 1;
    }
    /* new category and bin */
    float u0 = rng_uniform(cfg, eg, step, purpose, 0), u1 = rng_uniform(cfg, eg, step, purpose, 1);
    int cat = (int)(u0 * cfg->num_categories);
    if (cat >= cfg->num_categories) cat = cfg->num_categories - 1;
    const float* cdf = B->curriculum_cdf + (size_t)cat * cfg->num_bins;
    int bin = 0;
    while (bin < cfg->num_bins - 1 && !(u1 < cdf[bin])) bin++;
    B->env_command_bins[e] = bin;
    B->env_command_categories[e] = cat;
    int rem = bin;
    float cmd[GO1_MAX_COMMANDS];
    for (int kx = GO1_MAX_COMMANDS - 1; kx >= 0; kx--) {
      int nb = cfg->grid_bins[kx], idx = rem % nb;
      rem /= nb;
      float bs = (cfg->grid_high[kx] - cfg->grid_low[kx]) / nb;
      float centroid = cfg->grid_low[kx] + bs * (idx + 0.5f);
      float u = rng_uniform(cfg, eg, step, purpose, 2 + kx);
      cmd[kx] = centroid + (u - 0.5f) * bs;
    }
    if (cfg->num_commands > 5) {
      if (cfg->gaitwise_curricula) {   /* legged_robot.py:764-781 */
        if (cat == 0) { for (int kx = 5; kx < 8; kx++) cmd[kx] = (float)fmod1(cmd[kx] / 2 - 0.25f); }
        else if (cat == 1) { cmd[5] = cmd[5] / 2 + 0.25f; cmd[6] = 0; cmd[7] = 0; }
        else if (cat == 2) { cmd[5] = 0; cmd[6] = cmd[6] / 2 + 0.25f; cmd[7] = 0; }
        else { cmd[5] = 0; cmd[6] = 0; cmd[7] = cmd[7] / 2 + 0.25f; }
      } else if (cfg->exclusive_phase_offset) {   /* legged_robot.py:783-793 */
        float r = rng_uniform(cfg, eg, step, purpose, 2 + GO1_MAX_COMMANDS);
        int trot = r < 0.34f, pace = 0.34f <= r && r < 0.67f, bnd = 0.67f <= r;
        if (pace) cmd[5] = 0;
        if (bnd) cmd[5] = 0;
        if (trot) cmd[6] = 0;
        if (bnd) cmd[6] = 0;
        if (trot) cmd[7] = 0;
        if (pace) cmd[7] = 0;
      } else if (cfg->balance_gait_distribution) {   /* legged_robot.py:795-812, statement by statement */
        float r = rng_uniform(cfg, eg, step, purpose, 2 + GO1_MAX_COMMANDS);
        int pronk = r <= 0.25f, trot = 0.25f <= r && r < 0.50f, pace = 0.50f <= r && r < 0.75f, bnd = 0.75f <= r;
        if (pronk) for (int kx = 5; kx < 8; kx++) cmd[kx] = (float)fmod1(cmd[kx] / 2 - 0.25f);
        if (trot) { cmd[6] = 0; cmd[7] = 0; }
        if (pace) { cmd[5] = 0; cmd[7] = 0; }
        if (bnd) { cmd[5] = 0; cmd[6] = 0; }
        if (trot) cmd[5] = cmd[5] / 2 + 0.25f;
        if (pace) cmd[6] = cmd[6] / 2 + 0.25f;
        if (bnd) cmd[7] = cmd[7] / 2 + 0.25f;
      }
      if (cfg->binary_phases)          /* :814-817; torch.round = round-half-even = rint */
        for (int kx = 5; kx < 8; kx++) cmd[kx] = (float)fmod1(rintf(2 * cmd[kx]) / 2.0f);
    }
    float nrm = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);   /* :820 */
    if (!(nrm > 0.2f)) { cmd[0] = 0; cmd[1] = 0; }
    for (int kx = 0; kx < cfg->num_commands; kx++) AT(B->commands, kx, e) = cmd[kx];
  }
clear:
  for (int kx = 0; kx < cfg->num_rewards + 5; kx++) AT(B->command_sums, kx, e) = 0;   /* :823-824 */
}

/* applies the logged per-step updates in slot order (curriculum.py:135-154 once per step and category), then rebuilds the CDFs */
void go1_oracle_curriculum_update(const Go1SimConfig* cfg, const Go1SimBuffers* B) {
  const int K = cfg->curriculum_update_interval;
  float* add = (float*)calloc(cfg->num_bins, sizeof(float));
  for (int c = 0; c < cfg->num_categories; c++) {
    float* w = B->curriculum_weights + (size_t)c * cfg->num_bins;
    float* cdf = B->curriculum_cdf + (size_t)c * cfg->num_bins;
    for (int slot = 0; slot < K; slot++) {
      int32_t* s = B->curriculum_success + ((size_t)slot * cfg->num_categories + c) * cfg->num_bins;
      for (int b = 0; b < cfg->num_bins; b++) {
        int cnt = s[b] > 0 ? 1 : 0;
        for (int p = B->curriculum_nbr_ptr[b]; p < B->curriculum_nbr_ptr[b + 1]; p++) cnt += s[B->curriculum_nbr_idx[p]];
        add[b] = 0.2f * cnt;
      }
      for (int b = 0; b < cfg->num_bins; b++) { w[b] = fminf(1.0f, w[b] + add[b]); s[b] = 0; }
    }
    double tot = 0;
    for (int b = 0; b < cfg->num_bins; b++) tot += w[b];
    double run = 0;
    for (int b = 0; b < cfg->num_bins; b++) { run += w[b]; cdf[b] = (float)(run / tot); }
  }
  free(add);
}

static void randomize_dof_props(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int64_t step, uint32_t purpose) {
  const int N = cfg->num_envs;
  uint32_t eg = (uint32_t)(cfg->env_id_offset + e);
  if (cfg->randomize_motor_strength) {   /* one value per env, broadcast (legged_robot.py:646-650) */
    float u = rng_uniform(cfg, eg, step, purpose, 0);
    float v = u * (cfg->motor_strength_range[1] - cfg->motor_strength_range[0]) + cfg->motor_strength_range[0];
    for (int j = 0; j < 12; j++) AT(B->motor_strengths, j, e) = v;
  }
  if (cfg->randomize_motor_offset)
    for (int j = 0; j < 12; j++) {
      float u = rng_uniform(cfg, eg, step, purpose, 1 + j);
      AT(B->motor_offsets, j, e) = u * (cfg->motor_offset_range[1] - cfg->motor_offset_range[0]) + cfg->motor_offset_range[0];
    }
  if (cfg->randomize_Kp_factor) {
    float u = rng_uniform(cfg, eg, step, purpose, 13);
    float v = u * (cfg->Kp_factor_range[1] - cfg->Kp_factor_range[0]) + cfg->Kp_factor_range[0];
    for (int j = 0; j < 12; j++) AT(B->Kp_factors, j, e) = v;
  }
  if (cfg->randomize_Kd_factor) {
    float u = rng_uniform(cfg, eg, step, purpose, 14);
    float v = u * (cfg->Kd_factor_range[1] - cfg->Kd_factor_range[0]) + cfg->Kd_factor_range[0];
    for (int j = 0; j < 12; j++) AT(B->Kd_factors, j, e) = v;
  }
}

/* _randomize_rigid_body_props from the step callback (legged_robot.py:706-708,611-633); deviation: DESIGN.md */
static void randomize_rigid_props(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int64_t step, uint32_t purpose) {
  const int N = cfg->num_envs;
  uint32_t eg = (uint32_t)(cfg->env_id_offset + e);
  if (cfg->randomize_base_mass)
    B->payloads[e] = rng_uniform(cfg, eg, step, purpose, 0) * (cfg->added_mass_range[1] - cfg->added_mass_range[0]) + cfg->added_mass_range[0];
  if (cfg->randomize_com_displacement)
    for (int i = 0; i < 3; i++)
      AT(B->com_displacements, i, e) = rng_uniform(cfg, eg, step, purpose, 1 + i) * (cfg->com_displacement_range[1] - cfg->com_displacement_range[0]) + cfg->com_displacement_range[0];
  if (cfg->randomize_friction)
    B->friction_coeffs[e] = rng_uniform(cfg, eg, step, purpose, 4) * (cfg->friction_range[1] - cfg->friction_range[0]) + cfg->friction_range[0];
  if (cfg->randomize_restitution)
    B->restitutions[e] = rng_uniform(cfg, eg, step, purpose, 5) * (cfg->restitution_range[1] - cfg->restitution_range[0]) + cfg->restitution_range[0];
}

/* ---- train / evaluation split (reference eval_cfg: base_task.py:43-49, legged_robot.py:531-544 _call_train_eval) ----
 * environments [g_num_train, N) take their domain-randomisation / push / teleport / reset parameters from a second
 * configuration and stay out of the training episode log (include/go1sim.h go1sim_set_eval_config) */
static Go1SimConfig g_eval_copy;
static const Go1SimConfig* g_eval_cfg = NULL;
static int g_num_train = 0;
void go1_oracle_set_eval(const Go1SimConfig* eval_cfg, int num_train) {
  if (eval_cfg) { g_eval_copy = *eval_cfg; g_eval_cfg = &g_eval_copy; g_num_train = num_train; }
  else g_eval_cfg = NULL;
}
static int env_is_eval(int e) { return g_eval_cfg != NULL && e >= g_num_train; }
static const Go1SimConfig* env_cfg(const Go1SimConfig* cfg, int e) { return env_is_eval(e) ? g_eval_cfg : cfg; }

/* reset_idx for one env (legged_robot.py:150-239,948-1001) */
static void reset_env(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int64_t step, int lag_slots, int64_t slot_step) {
  const int N = cfg->num_envs;
  uint32_t eg = (uint32_t)(cfg->env_id_offset + e);
  resample_commands(cfg, B, e, step, P_CMD_RESET, slot_step);
  randomize_dof_props(cfg, B, e, step, P_DOFPROPS_RESET);
  if (cfg->randomize_rigids_after_start) randomize_rigid_props(cfg, B, e, step, P_RIGID_RESET);   /* :166-168 */
  for (int j = 0; j < 12; j++) {   /* _reset_dofs :956-958 */
    float u = rng_uniform(cfg, eg, step, P_RESET, j);
    AT(B->dof_pos, j, e) = cfg->default_dof_pos[j] * (0.5f + u);
    AT(B->dof_vel, j, e) = 0;
  }
  float root[13];
  for (int i = 0; i < 13; i++) root[i] = cfg->base_init_state[i];   /* _reset_root_states :973-997 */
  for (int i = 0; i < 3; i++) root[i] += AT(B->env_origins, i, e);
  if (cfg->custom_origins) {
    root[0] += (2 * rng_uniform(cfg, eg, step, P_RESET, 12) - 1) * cfg->x_init_range + cfg->x_init_offset;
    root[1] += (2 * rng_uniform(cfg, eg, step, P_RESET, 13) - 1) * cfg->y_init_range + cfg->y_init_offset;
  }
  float yaw = (2 * rng_uniform(cfg, eg, step, P_RESET, 14) - 1) * cfg->yaw_init_range;
  root[3] = 0; root[4] = 0; root[5] = sinf(0.5f * yaw); root[6] = cosf(0.5f * yaw);
  for (int i = 0; i < 6; i++) root[7 + i] = rng_uniform(cfg, eg, step, P_RESET, 15 + i) - 0.5f;
  for (int i = 0; i < 13; i++) AT(B->root_states, i, e) = root[i];
  for (int j = 0; j < 12; j++) { AT(B->last_actions, j, e) = 0; AT(B->last_last_actions, j, e) = 0; AT(B->last_dof_vel, j, e) = 0; }
  B->episode_length_buf[e] = 0;
  B->reset_buf[e] = 1;
  if (env_is_eval(e)) {     /* :188-195: kept out of the training log; the first finished episode is remembered */
    for (int kx = 0; kx <= cfg->num_rewards; kx++) {
      if (B->episode_sums_eval && AT(B->episode_sums_eval, kx, e) == -1.f) AT(B->episode_sums_eval, kx, e) = AT(B->episode_sums, kx, e);
      AT(B->episode_sums, kx, e) = 0;
    }
  } else if (g_log_defer) { /* parallel post-physics: park the per-env terms, the caller adds them in env order */
    float* row = g_log_defer + (size_t)e * g_log_stride;
    for (int kx = 0; kx <= cfg->num_rewards; kx++) { row[kx] = AT(B->episode_sums, kx, e); AT(B->episode_sums, kx, e) = 0; }
    row[cfg->num_rewards + 1] = 1;
  } else {
    for (int kx = 0; kx <= cfg->num_rewards; kx++) {   /* :181-187 logged mean is formed by the host from episode_log */
      B->episode_log[kx] += AT(B->episode_sums, kx, e);
      AT(B->episode_sums, kx, e) = 0;
    }
    B->episode_log[cfg->num_rewards + 1] += 1;
  }
  B->gait_indices[e] = 0;
  for (int sl = 0; sl < lag_slots; sl++)
    for (int j = 0; j < 12; j++) B->lag_buffer[((size_t)sl * 12 + j) * N + e] = 0;
}

void go1_oracle_reset_idx(const Go1SimConfig* cfg, const Go1SimBuffers* B, const int32_t* ids, int n, int64_t step) {
  for (int kx = 0; kx <= cfg->num_rewards + 1; kx++) B->episode_log[kx] = 0;
  for (int i = 0; i < (ids ? n : cfg->num_envs); i++) {
    const int e = ids ? ids[i] : i;
    reset_env(env_cfg(cfg, e), B, e, step, cfg->lag_timesteps + 1, step);
  }
}

/* ------------------------------------------------------------------ rewards (corl_rewards.py) */
static real normal_cdf(real x, real sigma) { return 0.5 * (1 + erf(x / (sigma * sqrt(2.0)))); }

typedef struct {
  real base_pos[3], base_quat[4], base_lin_vel[3], base_ang_vel[3], proj_g[3], gvec[3];
  real q[12], qd[12];
  real fpos[4][3], fvel[4][3], cf[17][3];
  real base_hz, fhz[4];      /* heights the reward terms read: world z (reference) or above the terrain (reward_heights_above_terrain) */
} Derived;

/* terrain height at world (x, y), sample convention of _get_heights (legged_robot.py:1793-1806): fp32 quotient truncated, the lowest
 * of the sample and its +x / +y neighbours */
static float hf_sample_min3(const Go1SimConfig* cfg, const int16_t* hs, float x, float y) {
  if (cfg->terrain_type == 0 || !hs) return 0;
  float fx = (x + cfg->hf_border) / cfg->hf_hscale, fy = (y + cfg->hf_border) / cfg->hf_hscale;
  long px = (long)fx, py = (long)fy;
  if (px < 0) px = 0; if (py < 0) py = 0;
  if (px > cfg->hf_rows - 2) px = cfg->hf_rows - 2;
  if (py > cfg->hf_cols - 2) py = cfg->hf_cols - 2;
  int16_t h1 = hs[px * cfg->hf_cols + py], h2 = hs[(px + 1) * cfg->hf_cols + py], h3 = hs[px * cfg->hf_cols + py + 1];
  int16_t hm = h1 < h2 ? h1 : h2;
  hm = hm < h3 ? hm : h3;
  return hm * cfg->hf_vscale;
}

static real reward_term(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int id, const Derived* d) {
  const int N = cfg->num_envs;
  real r = 0;
  switch (id) {
    case GO1_REW_TRACKING_LIN_VEL: {
      real ex = (real)AT(B->commands, 0, e) - d->base_lin_vel[0], ey = (real)AT(B->commands, 1, e) - d->base_lin_vel[1];
      return exp(-(ex * ex + ey * ey) / (real)cfg->tracking_sigma);
    }
    case GO1_REW_TRACKING_ANG_VEL: {
      real ez = (real)AT(B->commands, 2, e) - d->base_ang_vel[2];
      return exp(-(ez * ez) / (real)cfg->tracking_sigma_yaw);
    }
    case GO1_REW_LIN_VEL_Z: return d->base_lin_vel[2] * d->base_lin_vel[2];
    case GO1_REW_ANG_VEL_XY: return d->base_ang_vel[0] * d->base_ang_vel[0] + d->base_ang_vel[1] * d->base_ang_vel[1];
    case GO1_REW_ORIENTATION: return d->proj_g[0] * d->proj_g[0] + d->proj_g[1] * d->proj_g[1];
    case GO1_REW_TORQUES: for (int j = 0; j < 12; j++) r += (real)AT(B->torques, j, e) * (real)AT(B->torques, j, e); return r;
    case GO1_REW_DOF_ACC: for (int j = 0; j < 12; j++) { real a = ((real)AT(B->last_dof_vel, j, e) - d->qd[j]) / (real)cfg->dt; r += a * a; } return r;
    case GO1_REW_ACTION_RATE: for (int j = 0; j < 12; j++) { real a = (real)AT(B->last_actions, j, e) - (real)AT(B->actions, j, e); r += a * a; } return r;
    case GO1_REW_COLLISION:
      for (int b = 0; b < 17; b++) if (cfg->penalised_body_mask & (1u << b)) r += (v3norm(d->cf[b]) > 0.1) ? 1 : 0;
      return r;
    case GO1_REW_DOF_POS_LIMITS:
      for (int j = 0; j < 12; j++) {
        real lo = d->q[j] - (real)cfg->dof_pos_soft_lower[j], hi = d->q[j] - (real)cfg->dof_pos_soft_upper[j];
        r += -(lo < 0 ? lo : 0) + (hi > 0 ? hi : 0);
      }
      return r;
    case GO1_REW_JUMP: {
      real t = d->base_hz - ((real)AT(B->commands, 3, e) + (real)cfg->base_height_target);
      return -t * t;
    }
    case GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE:
      for (int f = 0; f < 4; f++) {
        real fn = v3norm(d->cf[4 + 4 * f]);
        r += -(1 - (real)AT(B->desired_contact_states, f, e)) * (1 - exp(-fn * fn / (real)cfg->gait_force_sigma));
      }
      return r / 4;
    case GO1_REW_TRACKING_CONTACTS_SHAPED_VEL:
      for (int f = 0; f < 4; f++) {
        real vn = v3norm(d->fvel[f]);
        r += -((real)AT(B->desired_contact_states, f, e) * (1 - exp(-vn * vn / (real)cfg->gait_vel_sigma)));
      }
      return r / 4;
    case GO1_REW_DOF_POS: for (int j = 0; j < 12; j++) { real a = d->q[j] - (real)cfg->default_dof_pos[j]; r += a * a; } return r;
    case GO1_REW_DOF_VEL: for (int j = 0; j < 12; j++) r += d->qd[j] * d->qd[j]; return r;
    case GO1_REW_ACTION_SMOOTHNESS_1:
      for (int j = 0; j < 12; j++) {
        real a = (real)AT(B->joint_pos_target, j, e) - (real)AT(B->last_joint_pos_target, j, e);
        r += a * a * (AT(B->last_actions, j, e) != 0 ? 1 : 0);
      }
      return r;
    case GO1_REW_ACTION_SMOOTHNESS_2:
      for (int j = 0; j < 12; j++) {
        real a = (real)AT(B->joint_pos_target, j, e) - 2 * (real)AT(B->last_joint_pos_target, j, e) + (real)AT(B->last_last_joint_pos_target, j, e);
        r += a * a * (AT(B->last_actions, j, e) != 0 ? 1 : 0) * (AT(B->last_last_actions, j, e) != 0 ? 1 : 0);
      }
      return r;
    case GO1_REW_FEET_SLIP:
      for (int f = 0; f < 4; f++) {
        int contact = d->cf[4 + 4 * f][2] > 1.0;
        int filt = contact || AT(B->last_contacts, f, e);
        AT(B->last_contacts, f, e) = (uint8_t)contact;
        r += filt * (d->fvel[f][0] * d->fvel[f][0] + d->fvel[f][1] * d->fvel[f][1]);
      }
      return r;
    case GO1_REW_FEET_CONTACT_VEL:
      for (int f = 0; f < 4; f++) r += (d->fhz[f] < 0.03 ? 1 : 0) * v3dot(d->fvel[f], d->fvel[f]);
      return r;
    case GO1_REW_FEET_CONTACT_FORCES:
      for (int f = 0; f < 4; f++) { real x = v3norm(d->cf[4 + 4 * f]) - (real)cfg->max_contact_force; r += x > 0 ? x : 0; }
      return r;
    case GO1_REW_FEET_CLEARANCE_CMD_LINEAR:
      for (int f = 0; f < 4; f++) {
        real fi = AT(B->foot_indices, f, e);
        real cl = fi * 2.0 - 1.0; cl = cl < 0 ? 0 : (cl > 1 ? 1 : cl);
        real ph = 1 - fabs(1.0 - cl * 2.0);
        real target = (real)AT(B->commands, 9, e) * ph + 0.02;
        real df = target - d->fhz[f];
        r += df * df * (1 - (real)AT(B->desired_contact_states, f, e));
      }
      return r;
    case GO1_REW_FEET_IMPACT_VEL:
      for (int f = 0; f < 4; f++) {
        real pv = AT(B->prev_foot_velocities, 3 * f + 2, e);
        pv = pv > 0 ? 0 : (pv < -100 ? -100 : pv);
        r += (v3norm(d->cf[4 + 4 * f]) > 1.0 ? 1 : 0) * pv * pv;
      }
      return r;
    case GO1_REW_ORIENTATION_CONTROL: {
      real pitch = AT(B->commands, 10, e), roll = AT(B->commands, 11, e);
      real qr[4] = {sin(-roll / 2), 0, 0, cos(-roll / 2)}, qp[4] = {0, sin(-pitch / 2), 0, cos(-pitch / 2)}, qd_[4], g[3];
      quat_mul(qd_, qr, qp);
      quat_rotate_inverse(g, qd_, d->gvec);
      real a = d->proj_g[0] - g[0], b = d->proj_g[1] - g[1];
      return a * a + b * b;
    }
    case GO1_REW_RAIBERT_HEURISTIC: {
      /* quat_apply_yaw(quat_conjugate(base_quat), foot - base) (math_utils.py:12-17) */
      real qy[4] = {0, 0, -d->base_quat[2], d->base_quat[3]};
      real l = sqrt(qy[2] * qy[2] + qy[3] * qy[3]);
      qy[2] /= l; qy[3] /= l;
      real width = cfg->num_commands >= 13 ? (real)AT(B->commands, 12, e) : 0.3;
      real length = cfg->num_commands >= 14 ? (real)AT(B->commands, 13, e) : 0.45;
      real freq = AT(B->commands, 4, e), xv = AT(B->commands, 0, e), yawv = AT(B->commands, 2, e);
      real yv = yawv * length / 2;
      for (int f = 0; f < 4; f++) {
        real rel[3], fb[3];
        v3sub(rel, d->fpos[f], d->base_pos);
        quat_rotate(fb, qy, rel);
        real ys = (f % 2 == 0 ? 1 : -1) * width / 2, xs = (f < 2 ? 1 : -1) * length / 2;
        real ph = fabs(1.0 - (real)AT(B->foot_indices, f, e) * 2.0) * 1.0 - 0.5;
        real yo = ph * yv * (0.5 / freq), xo = ph * xv * (0.5 / freq);
        if (f >= 2) yo *= -1;
        real ex = fabs((xs + xo) - fb[0]), ey = fabs((ys + yo) - fb[1]);
        r += ex * ex + ey * ey;
      }
      return r;
    }
  }
  return 0;
}
/* sign class of each raw term: +1 raw >= 0, -1 raw <= 0 (DESIGN.md "batch-sum sign") */
static int reward_raw_sign(int id) {
  return (id == GO1_REW_JUMP || id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL) ? -1 : 1;
}

/* ------------------------------------------------------------------ post-physics for one env */
static void post_physics(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int64_t counter_post,
                         const real* grav_used, int lag_slots) {
  const int N = cfg->num_envs;
  uint32_t eg = (uint32_t)(cfg->env_id_offset + e);
  Derived d;
  B->episode_length_buf[e] += 1;
  Phys s;
  load_phys(cfg, B, e, &s);
  v3cpy(d.base_pos, s.pos);
  memcpy(d.base_quat, s.quat, sizeof d.base_quat);
  quat_rotate_inverse(d.base_lin_vel, s.quat, s.vlin);
  quat_rotate_inverse(d.base_ang_vel, s.quat, s.vang);
  real gn = v3norm(grav_used);
  for (int i = 0; i < 3; i++) d.gvec[i] = grav_used[i] / gn;
  quat_rotate_inverse(d.proj_g, s.quat, d.gvec);
  for (int i = 0; i < 3; i++) {
    AT(B->base_lin_vel, i, e) = (float)d.base_lin_vel[i];
    AT(B->base_ang_vel, i, e) = (float)d.base_ang_vel[i];
    AT(B->projected_gravity, i, e) = (float)d.proj_g[i];
  }
  for (int j = 0; j < 12; j++) { d.q[j] = s.q[j]; d.qd[j] = s.qd[j]; }
  for (int f = 0; f < 4; f++)
    for (int i = 0; i < 3; i++) { d.fpos[f][i] = AT(B->foot_positions, 3 * f + i, e); d.fvel[f][i] = AT(B->foot_velocities, 3 * f + i, e); }
  for (int b = 0; b < 17; b++)
    for (int i = 0; i < 3; i++) d.cf[b][i] = AT(B->contact_forces, 3 * b + i, e);

  /* ---- _post_physics_step_callback (legged_robot.py:675-708) ---- */
  if (cfg->teleport_robots) {   /* :1028-1051 */
    float x = AT(B->root_states, 0, e), y = AT(B->root_states, 1, e), th = cfg->teleport_thresh, xo = cfg->teleport_x_offset;
    if (x < th + xo) x += cfg->terrain_length * (cfg->terrain_num_rows - 1);
    if (x > cfg->terrain_length * cfg->terrain_num_rows - th + xo) x -= cfg->terrain_length * (cfg->terrain_num_rows - 1);
    if (y < th) y += cfg->terrain_width * (cfg->terrain_num_cols - 1);
    if (y > cfg->terrain_width * cfg->terrain_num_cols - th) y -= cfg->terrain_width * (cfg->terrain_num_cols - 1);
    AT(B->root_states, 0, e) = x; AT(B->root_states, 1, e) = y;
  }
  if (B->episode_length_buf[e] % cfg->resample_interval == 0) resample_commands(cfg, B, e, counter_post, P_CMD_CB, counter_post - 1);
  if (cfg->observe_gait_commands) {   /* _step_contact_targets :826-905 (float32 arithmetic order preserved where it matters) */
    float freq = AT(B->commands, 4, e), phase = AT(B->commands, 5, e), offset = AT(B->commands, 6, e), bound = AT(B->commands, 7, e), dur = AT(B->commands, 8, e);
    float gi = (float)fmod1((real)(float)(B->gait_indices[e] + cfg->dt * freq));
    B->gait_indices[e] = gi;
    float fi[4];
    if (cfg->pacing_offset) { fi[0] = gi + phase + offset + bound; fi[1] = gi + bound; fi[2] = gi + offset; fi[3] = gi + phase; }
    else                    { fi[0] = gi + phase + offset + bound; fi[1] = gi + offset; fi[2] = gi + bound; fi[3] = gi + phase; }
    for (int f = 0; f < 4; f++) {
      real rem = fmod1(fi[f]);
      AT(B->foot_indices, f, e) = (float)rem;
      real idx = fi[f];
      if (rem < dur) idx = rem * (0.5 / dur);
      else if (rem > dur) idx = 0.5 + (rem - dur) * (0.5 / (1 - dur));
      AT(B->clock_inputs, f, e) = (float)sin(2 * PI * idx);
      real kap = cfg->kappa_gait_probs, x = fmod1(idx);
      real sm = normal_cdf(x, kap) * (1 - normal_cdf(x - 0.5, kap)) + normal_cdf(x - 1, kap) * (1 - normal_cdf(x - 0.5 - 1, kap));
      AT(B->desired_contact_states, f, e) = (float)sm;
    }
  }
  if (cfg->push_robots && B->episode_length_buf[e] % cfg->push_interval == 0) {   /* :1017-1026 */
    for (int i = 0; i < 2; i++)
      AT(B->root_states, 7 + i, e) = (2 * rng_uniform(cfg, eg, counter_post, P_PUSH, i) - 1) * cfg->max_push_vel_xy;
  }
  if (B->episode_length_buf[e] % cfg->rand_interval == 0) {
    randomize_dof_props(cfg, B, e, counter_post, P_DOFPROPS_CB);
    if (cfg->randomize_rigids_after_start) randomize_rigid_props(cfg, B, e, counter_post, P_RIGID);
  }

  /* ---- measured terrain heights (:689-691, _get_heights :1772-1806) ---- */
  real mean_height = 0;
  if (cfg->measure_heights && B->measured_heights) {
    const int np = cfg->num_height_x * cfg->num_height_y;
    real qy[4] = {0, 0, AT(B->root_states, 5, e), AT(B->root_states, 6, e)};     /* quat_apply_yaw (math_utils.py:12-17) */
    real l = sqrt(qy[2] * qy[2] + qy[3] * qy[3]);
    qy[2] /= l; qy[3] /= l;
    for (int p = 0; p < np; p++) {
      real loc[3] = {cfg->height_points_x[p / cfg->num_height_y], cfg->height_points_y[p % cfg->num_height_y], 0}, w[3];
      quat_rotate(w, qy, loc);
      float hgt = 0;
      if (cfg->terrain_type != 0 && B->height_samples) {
        /* float32 arithmetic and truncation toward zero as `(points / horizontal_scale).long()` */
        float fx = ((float)w[0] + AT(B->root_states, 0, e) + cfg->hf_border) / cfg->hf_hscale;
        float fy = ((float)w[1] + AT(B->root_states, 1, e) + cfg->hf_border) / cfg->hf_hscale;
        long px = (long)fx, py = (long)fy;
        if (px < 0) px = 0; if (py < 0) py = 0;
        if (px > cfg->hf_rows - 2) px = cfg->hf_rows - 2;
        if (py > cfg->hf_cols - 2) py = cfg->hf_cols - 2;
        int16_t h1 = B->height_samples[px * cfg->hf_cols + py], h2 = B->height_samples[(px + 1) * cfg->hf_cols + py],
                h3 = B->height_samples[px * cfg->hf_cols + py + 1];
        int16_t hm = h1 < h2 ? h1 : h2;
        hm = hm < h3 ? hm : h3;
        hgt = hm * cfg->hf_vscale;
      }
      AT(B->measured_heights, p, e) = hgt;
      mean_height += hgt;
    }
    mean_height /= np;
  }

  d.base_hz = d.base_pos[2];
  for (int f = 0; f < 4; f++) d.fhz[f] = d.fpos[f][2];
  if (cfg->reward_heights_above_terrain) {          /* NOT in the reference (go1sim.h) */
    for (int f = 0; f < 4; f++) d.fhz[f] -= (real)hf_sample_min3(cfg, B->height_samples, AT(B->foot_positions, 3 * f, e), AT(B->foot_positions, 3 * f + 1, e));
    d.base_hz -= (cfg->measure_heights && B->measured_heights) ? mean_height
                                                               : (real)hf_sample_min3(cfg, B->height_samples, AT(B->root_states, 0, e), AT(B->root_states, 1, e));
  }
  /* ---- check_termination (:138-148) ---- */
  int reset = 0;
  for (int b = 0; b < 17; b++) if ((cfg->termination_body_mask & (1u << b)) && v3norm(d.cf[b]) > 1.0) reset = 1;
  int time_out = B->episode_length_buf[e] > cfg->max_episode_length;
  reset |= time_out;
  if (cfg->use_terminal_body_height && (real)AT(B->root_states, 2, e) - mean_height < (real)cfg->terminal_body_height) reset = 1;
  B->time_out_buf[e] = (uint8_t)time_out;
  B->reset_buf[e] = (uint8_t)reset;

  /* ---- compute_reward (:263-300) ---- */
  real rew = 0, pos = 0, neg = 0;
  for (int kx = 0; kx < cfg->num_rewards; kx++) {
    int id = cfg->reward_ids[kx];
    real r = reward_term(cfg, B, e, id, &d) * (real)cfg->reward_scales[kx];
    rew += r;
    if (reward_raw_sign(id) * cfg->reward_scales[kx] >= 0) pos += r; else neg += r;
    AT(B->episode_sums, kx, e) += (float)r;
    if (id == GO1_REW_TRACKING_CONTACTS_SHAPED_FORCE || id == GO1_REW_TRACKING_CONTACTS_SHAPED_VEL)
      AT(B->command_sums, kx, e) += (float)((real)cfg->reward_scales[kx] + r);
    else
      AT(B->command_sums, kx, e) += (float)r;
  }
  if (cfg->only_positive_rewards) rew = rew < 0 ? 0 : rew;
  else if (cfg->only_positive_rewards_ji22_style) rew = pos * exp(neg / (real)cfg->sigma_rew_neg);
  B->rew_buf[e] = (float)rew;
  AT(B->episode_sums, cfg->num_rewards, e) += (float)rew;
  {
    int k0 = cfg->num_rewards;
    real vx = d.base_lin_vel[0], wz = d.base_ang_vel[2], c0 = AT(B->commands, 0, e), c2 = AT(B->commands, 2, e);
    AT(B->command_sums, k0 + 0, e) += (float)vx;
    AT(B->command_sums, k0 + 1, e) += (float)wz;
    AT(B->command_sums, k0 + 2, e) += (float)((vx - c0) * (vx - c0));
    AT(B->command_sums, k0 + 3, e) += (float)((wz - c2) * (wz - c2));
    AT(B->command_sums, k0 + 4, e) += 1;
  }

  /* ---- reset (:122-123) ---- */
  if (reset) reset_env(cfg, B, e, counter_post, lag_slots, counter_post - 1);

  /* ---- compute_observations (:302-491) ---- */
  {
    float obs[GO1_MAX_OBS];
    int n = 0;
    float core[GO1_MAX_OBS];
    int nc = 0;
    for (int i = 0; i < 3; i++) core[nc++] = AT(B->projected_gravity, i, e);
    if (cfg->observe_command)
      for (int kx = 0; kx < cfg->num_commands; kx++) core[nc++] = AT(B->commands, kx, e) * cfg->commands_scale[kx];
    for (int j = 0; j < 12; j++) core[nc++] = (AT(B->dof_pos, j, e) - cfg->default_dof_pos[j]) * cfg->obs_scale_dof_pos;
    for (int j = 0; j < 12; j++) core[nc++] = AT(B->dof_vel, j, e) * cfg->obs_scale_dof_vel;
    for (int j = 0; j < 12; j++) core[nc++] = AT(B->actions, j, e);
    if (cfg->observe_two_prev_actions) for (int j = 0; j < 12; j++) core[nc++] = AT(B->last_actions, j, e);
    if (cfg->observe_timing_parameter) core[nc++] = B->gait_indices[e];
    if (cfg->observe_clock_inputs) for (int f = 0; f < 4; f++) core[nc++] = AT(B->clock_inputs, f, e);
    /* prefixes are prepended in this order: vel block first, then only_ang_vel, then only_lin_vel in front */
    if (cfg->observe_only_lin_vel) for (int i = 0; i < 3; i++) obs[n++] = AT(B->base_lin_vel, i, e) * cfg->obs_scale_lin_vel;
    if (cfg->observe_only_ang_vel) for (int i = 0; i < 3; i++) obs[n++] = AT(B->base_ang_vel, i, e) * cfg->obs_scale_ang_vel;
    if (cfg->observe_vel) {
      for (int i = 0; i < 3; i++)
        obs[n++] = (cfg->global_reference ? AT(B->root_states, 7 + i, e) : AT(B->base_lin_vel, i, e)) * cfg->obs_scale_lin_vel;
      for (int i = 0; i < 3; i++) obs[n++] = AT(B->base_ang_vel, i, e) * cfg->obs_scale_ang_vel;
    }
    for (int i = 0; i < nc; i++) obs[n++] = core[i];
    if (cfg->observe_yaw) {
      real q[4] = {AT(B->root_states, 3, e), AT(B->root_states, 4, e), AT(B->root_states, 5, e), AT(B->root_states, 6, e)};
      real fw[3] = {1, 0, 0}, o[3];
      quat_rotate(o, q, fw);
      obs[n++] = (float)atan2(o[1], o[0]);
    }
    if (cfg->observe_contact_states) for (int f = 0; f < 4; f++) obs[n++] = AT(B->contact_forces, 3 * (4 + 4 * f) + 2, e) > 1.0f ? 1.0f : 0.0f;
    for (int i = 0; i < n; i++) {
      float v = obs[i];
      if (cfg->add_noise && cfg->noise_scale_vec[i] != 0)
        v += (2 * rng_uniform(cfg, eg, counter_post, P_NOISE, i) - 1) * cfg->noise_scale_vec[i];
      if (v > cfg->clip_observations) v = cfg->clip_observations;
      if (v < -cfg->clip_observations) v = -cfg->clip_observations;
      B->obs_buf[(size_t)e * cfg->num_obs + i] = v;
    }
    if (cfg->observe_heights && cfg->measure_heights && B->measured_heights) {     /* legacy legged_gym block, BASELINE config 3 */
      const int np = cfg->num_height_x * cfg->num_height_y;
      for (int p = 0; p < np; p++) {
        float v = AT(B->root_states, 2, e) - 0.5f - AT(B->measured_heights, p, e);
        v = (v < -1.f ? -1.f : (v > 1.f ? 1.f : v)) * cfg->obs_scale_height;
        if (cfg->add_noise && cfg->height_noise_scale != 0)
          v += (2 * rng_uniform(cfg, eg, counter_post, P_NOISE, n + p) - 1) * cfg->height_noise_scale;
        if (v > cfg->clip_observations) v = cfg->clip_observations;
        if (v < -cfg->clip_observations) v = -cfg->clip_observations;
        B->obs_buf[(size_t)e * cfg->num_obs + n + p] = v;
      }
    }
    /* privileged observations */
    float pv[GO1_MAX_PRIV_OBS];
    int np = 0;
#define PRIV(idx, val) pv[np++] = ((val) - cfg->priv_shift[idx]) * cfg->priv_scale[idx]
    if (cfg->priv_enabled[GO1_PRIV_FRICTION]) PRIV(GO1_PRIV_FRICTION, B->friction_coeffs[e]);
    if (cfg->priv_enabled[GO1_PRIV_RESTITUTION]) PRIV(GO1_PRIV_RESTITUTION, B->restitutions[e]);
    if (cfg->priv_enabled[GO1_PRIV_BASE_MASS]) PRIV(GO1_PRIV_BASE_MASS, B->payloads[e]);
    if (cfg->priv_enabled[GO1_PRIV_COM_DISPLACEMENT]) for (int i = 0; i < 3; i++) PRIV(GO1_PRIV_COM_DISPLACEMENT, AT(B->com_displacements, i, e));
    if (cfg->priv_enabled[GO1_PRIV_MOTOR_STRENGTH]) for (int j = 0; j < 12; j++) PRIV(GO1_PRIV_MOTOR_STRENGTH, AT(B->motor_strengths, j, e));
    if (cfg->priv_enabled[GO1_PRIV_MOTOR_OFFSET]) for (int j = 0; j < 12; j++) PRIV(GO1_PRIV_MOTOR_OFFSET, AT(B->motor_offsets, j, e));
    if (cfg->priv_enabled[GO1_PRIV_BODY_HEIGHT]) PRIV(GO1_PRIV_BODY_HEIGHT, AT(B->root_states, 2, e));
    if (cfg->priv_enabled[GO1_PRIV_BODY_VELOCITY]) for (int i = 0; i < 3; i++) PRIV(GO1_PRIV_BODY_VELOCITY, AT(B->base_lin_vel, i, e));
    if (cfg->priv_enabled[GO1_PRIV_GRAVITY])   /* (gravities - shift) / scale: division, legged_robot.py:477 */
      for (int i = 0; i < 3; i++) pv[np++] = ((float)(grav_used[i] - cfg->gravity[i]) - cfg->priv_shift[GO1_PRIV_GRAVITY]) / cfg->priv_scale[GO1_PRIV_GRAVITY];
    if (cfg->priv_enabled[GO1_PRIV_CLOCK_INPUTS]) for (int f = 0; f < 4; f++) pv[np++] = AT(B->clock_inputs, f, e);
    if (cfg->priv_enabled[GO1_PRIV_DESIRED_CONTACT]) for (int f = 0; f < 4; f++) pv[np++] = AT(B->desired_contact_states, f, e);
    for (int i = 0; i < np; i++) {
      float v = pv[i];
      if (v > cfg->clip_observations) v = cfg->clip_observations;
      if (v < -cfg->clip_observations) v = -cfg->clip_observations;
      B->privileged_obs_buf[(size_t)e * cfg->num_privileged_obs + i] = v;
    }
  }
  /* ---- roll (:126-131) ---- */
  for (int j = 0; j < 12; j++) {
    AT(B->last_last_actions, j, e) = AT(B->last_actions, j, e);
    AT(B->last_actions, j, e) = AT(B->actions, j, e);
    AT(B->last_last_joint_pos_target, j, e) = AT(B->last_joint_pos_target, j, e);
    AT(B->last_joint_pos_target, j, e) = AT(B->joint_pos_target, j, e);
    AT(B->last_dof_vel, j, e) = AT(B->dof_vel, j, e);
  }
}

/* history append (history_wrapper.py:23): ring of R = H+1 slots stored twice back to back; the observation of
 * this step goes to slot w and w+R.  The reference's obs_history (oldest first) after the append is the H-slot
 * window starting at slot (w+2) mod R of the doubled row; the previous step's window is still intact. */
static void history_append(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int slot) {
  const int no = cfg->num_obs, R = cfg->num_obs_history + 1;
  float* row = B->obs_history + (size_t)e * 2 * R * no;
  const float* obs = B->obs_buf + (size_t)e * no;
  memcpy(row + (size_t)slot * no, obs, no * sizeof(float));
  memcpy(row + (size_t)(slot + R) * no, obs, no * sizeof(float));
}

/* per-substep contact bookkeeping into the (optional) buffers: drops per class, signature words of substep `sub` */
static void publish_contact_stats(const Go1SimConfig* cfg, const Go1SimBuffers* B, int e, int sub, const ContactOut* co) {
  const int N = cfg->num_envs;
  if (B->contact_drop_counts)
    for (int c = 0; c < GO1_CC_COUNT; c++)
      if (co->dropped[c]) {
#pragma omp atomic
        B->contact_drop_counts[c] += (uint32_t)co->dropped[c];
      }
  if (B->contact_signature && sub < GO1_SIG_MAX_SUBSTEPS)
    for (int w = 0; w < GO1_SIG_WORDS; w++) AT(B->contact_signature, sub * GO1_SIG_WORDS + w, e) = co->sig[w];
}

/* ------------------------------------------------------------------ public step */
typedef struct { int64_t common_step_counter; int32_t lag_head; int32_t history_slot; } Go1OracleCounters;

/* actions: row-major (N,12).  Advances the counters. */
void go1_oracle_step(const Go1SimConfig* cfg, const Go1SimBuffers* B, const float* actions, Go1OracleCounters* ctr) {
  const int N = cfg->num_envs;
  const int nl = cfg->lag_timesteps + 1;
  Terrain ter = {cfg, B->height_samples};
  real grav[3];
  gravity_at(cfg, ctr->common_step_counter, grav);
  int64_t counter_post = ctr->common_step_counter + 1;
  for (int kx = 0; kx <= cfg->num_rewards + 1; kx++) B->episode_log[kx] = 0;
  int head_end = ctr->lag_head;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < N; e++) {
    for (int j = 0; j < 12; j++) {
      float a = actions[(size_t)e * 12 + j];
      if (a > cfg->clip_actions) a = cfg->clip_actions;
      if (a < -cfg->clip_actions) a = -cfg->clip_actions;
      AT(B->actions, j, e) = a;
    }
    for (int i = 0; i < 12; i++) AT(B->prev_foot_velocities, i, e) = AT(B->foot_velocities, i, e);
    Phys s;
    load_phys(cfg, B, e, &s);
    real lam[17][3];
    int warm = cfg->warm_start && B->episode_length_buf[e] > 0;
    for (int b = 0; b < 17; b++)
      for (int i = 0; i < 3; i++) lam[b][i] = 0;
    if (warm) {
      for (int b = 0; b < 17; b++)       /* world impulse of the previous step's last substep */
        for (int i = 0; i < 3; i++) lam[b][i] = (real)AT(B->contact_forces, 3 * b + i, e) * (real)cfg->sim_dt;
    }
    ContactOut co;
  memset(&co, 0, sizeof co);
    int head = ctr->lag_head;
    for (int sub = 0; sub < cfg->decimation; sub++) {
      real tau[12];
      compute_torques(cfg, B, e, &head, 1, s.q, s.qd, tau);
      physics_substep(cfg, &ter, &s, tau, grav, lam, warm || (cfg->warm_start && sub > 0), &co);
      publish_contact_stats(cfg, B, e, sub, &co);
    }
    store_phys(cfg, B, e, &s);
    real fp[4][3], fv[4][3];
    feet_state(&s, fp, fv);
    for (int f = 0; f < 4; f++)
      for (int i = 0; i < 3; i++) { AT(B->foot_positions, 3 * f + i, e) = (float)fp[f][i]; AT(B->foot_velocities, 3 * f + i, e) = (float)fv[f][i]; }
    for (int b = 0; b < 17; b++)
      for (int i = 0; i < 3; i++) AT(B->contact_forces, 3 * b + i, e) = (float)co.force[b][i];
    if (e == 0) head_end = head;
  }
  if (N > 0) ctr->lag_head = (ctr->lag_head + cfg->decimation) % nl;
  (void)head_end;
  /* post-physics: per-env work in parallel; the two shared accumulators stay deterministic (integer atomics for the
   * curriculum success counts, env-ordered serial sum for episode_log) */
  {
    const int stride = cfg->num_rewards + 2;
    float* defer = (float*)calloc((size_t)N * stride, sizeof(float));
    g_log_defer = defer;
    g_log_stride = stride;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; e++) {
      post_physics(env_cfg(cfg, e), B, e, counter_post, grav, nl);
      if (B->obs_history) history_append(cfg, B, e, ctr->history_slot);
    }
    g_log_defer = NULL;
    for (int e = 0; e < N; e++) {
      const float* row = defer + (size_t)e * stride;
      if (row[stride - 1] == 0) continue;
      for (int kx = 0; kx < stride - 1; kx++) B->episode_log[kx] += row[kx];
      B->episode_log[stride - 1] += 1;
    }
    free(defer);
  }
  ctr->common_step_counter = counter_post;
  ctr->history_slot = (ctr->history_slot + 1) % (cfg->num_obs_history + 1);
  if (cfg->device_curriculum && B->curriculum_weights && !cfg->defer_curriculum_update && counter_post % cfg->curriculum_update_interval == 0)
    go1_oracle_curriculum_update(cfg, B);
}

/* piecewise entry points mirroring go1sim_compute_torques / go1sim_physics_substep */
void go1_oracle_compute_torques(const Go1SimConfig* cfg, const Go1SimBuffers* B, const float* actions_soa, Go1OracleCounters* ctr) {
  const int N = cfg->num_envs;
  for (int e = 0; e < N; e++) {
    for (int j = 0; j < 12; j++) AT(B->actions, j, e) = AT(actions_soa, j, e);
    real q[12], qd[12], tau[12];
    for (int j = 0; j < 12; j++) { q[j] = AT(B->dof_pos, j, e); qd[j] = AT(B->dof_vel, j, e); }
    int head = ctr->lag_head;
    compute_torques(cfg, B, e, &head, 1, q, qd, tau);
  }
  ctr->lag_head = (ctr->lag_head + 1) % (cfg->lag_timesteps + 1);
}
void go1_oracle_physics_substep(const Go1SimConfig* cfg, const Go1SimBuffers* B, Go1OracleCounters* ctr) {
  const int N = cfg->num_envs;
  Terrain ter = {cfg, B->height_samples};
  real grav[3];
  gravity_at(cfg, ctr->common_step_counter, grav);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < N; e++) {
    Phys s;
    load_phys(cfg, B, e, &s);
    real lam[17][3], tau[12];
    for (int b = 0; b < 17; b++)
      for (int i = 0; i < 3; i++) lam[b][i] = (real)AT(B->contact_forces, 3 * b + i, e) * (real)cfg->sim_dt;
    for (int j = 0; j < 12; j++) tau[j] = AT(B->torques, j, e);
    ContactOut co;
  memset(&co, 0, sizeof co);
    physics_substep(cfg, &ter, &s, tau, grav, lam, cfg->warm_start, &co);
    publish_contact_stats(cfg, B, e, 0, &co);
    store_phys(cfg, B, e, &s);
    real fp[4][3], fv[4][3];
    feet_state(&s, fp, fv);
    for (int f = 0; f < 4; f++)
      for (int i = 0; i < 3; i++) { AT(B->foot_positions, 3 * f + i, e) = (float)fp[f][i]; AT(B->foot_velocities, 3 * f + i, e) = (float)fv[f][i]; }
    for (int b = 0; b < 17; b++)
      for (int i = 0; i < 3; i++) AT(B->contact_forces, 3 * b + i, e) = (float)co.force[b][i];
  }
}

/* post-physics only (tensor maps) on whatever is in the buffers: used to pin the maps against the
 * reference's Python (tests/test_oracle_golden.py).  grav_used: gravity vector of the step. */
void go1_oracle_post_physics(const Go1SimConfig* cfg, const Go1SimBuffers* B, const double* grav_d, Go1OracleCounters* ctr) {
  const real grav_used[3] = {(real)grav_d[0], (real)grav_d[1], (real)grav_d[2]};
  int64_t counter_post = ctr->common_step_counter + 1;
  for (int kx = 0; kx <= cfg->num_rewards + 1; kx++) B->episode_log[kx] = 0;
  for (int e = 0; e < cfg->num_envs; e++) post_physics(env_cfg(cfg, e), B, e, counter_post, grav_used, cfg->lag_timesteps + 1);
  ctr->common_step_counter = counter_post;
}

void go1_oracle_gravity_at(const Go1SimConfig* cfg, int64_t t, double* g) {
  real gr[3];
  gravity_at(cfg, t, gr);
  for (int i = 0; i < 3; i++) g[i] = gr[i];
}
int go1_oracle_real_bytes(void) { return (int)sizeof(real); }
float go1_oracle_uniform(const Go1SimConfig* cfg, uint32_t env, int64_t step, uint32_t purpose, uint32_t idx) {
  return rng_uniform(cfg, env, step, purpose, idx);
}
int go1_oracle_sizeof_config(void) { return (int)sizeof(Go1SimConfig); }
int go1_oracle_sizeof_buffers(void) { return (int)sizeof(Go1SimBuffers); }
