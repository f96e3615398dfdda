// go1_math.h — small fixed-size algebra for the Go1 step kernel (device code, fp32).
// 3-vectors, spatial (6D) vectors [angular; linear], symmetric 6x6 articulated inertias,
// xyzw quaternions, Philox4x32-10.  Everything is force-inlined and register resident.
#pragma once

// ---- optional phase profiling (build with -DGO1_PROFILE; tools/phase_profile.py): the first lane of workgroup 0
// accumulates s_memtime deltas per phase into g_prof[]; everything compiles away otherwise ------------------
#ifdef GO1_PROFILE
__device__ unsigned long long g_prof[64];
__device__ unsigned long long g_lmax[64], g_lsum[64];   // per launch (slot = step counter & 63): the slowest workgroup's / all workgroups' master cycles
__device__ unsigned long long g_profw[1024 * 40];       // the phase accumulators of EVERY workgroup (which phases make the slowest workgroups slow)
__device__ unsigned long long g_wgt[1024];          // per workgroup: the master wavefront's cycles, accumulated over the launches (spread between workgroups)
__shared__ unsigned long long s_prof[40];          // accumulated with fire-and-forget LDS adds: no memory stall per marker
#define PROF_PARAM , unsigned long long& prof_t
#define PROF_PASS , prof_t
#define PROF_INIT if (threadIdx.x < 40) s_prof[threadIdx.x] = 0; __syncthreads();
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(); const unsigned long long prof_t0 = prof_t;
// (sched_barrier: the counter read is a scheduling fence — without it the compiler sinks a phase's tail, e.g. the pose update's sincos, past the marker)
#define PROF(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) atomicAdd(&s_prof[i], now_ - prof_t); prof_t = now_; } while (0)
#define PROF_FLUSH do { LDS_PHASE(); if (blockIdx.x == 0 && threadIdx.x < 40) g_prof[threadIdx.x] += s_prof[threadIdx.x]; if (blockIdx.x < 1024 && threadIdx.x < 40) g_profw[blockIdx.x * 40 + threadIdx.x] += s_prof[threadIdx.x]; if (threadIdx.x == 0 && blockIdx.x < 1024) { const unsigned long long dt_ = __builtin_readcyclecounter() - prof_t0; g_wgt[blockIdx.x] += dt_; atomicMax(&g_lmax[PROF_LAUNCH & 63], dt_); atomicAdd(&g_lsum[PROF_LAUNCH & 63], dt_); } } while (0)
#else
#define PROF_PARAM
#define PROF_PASS
#define PROF_INIT
#define PROF_DECL
#define PROF(i) do { } while (0)
#define PROF_FLUSH do { } while (0)
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEV __device__ __forceinline__
// A branch into a large, rarely executed block (a reset, a command resampling, the self-contact rows): the hint moves the block out of the
// fall-through path, so that the common path runs straight through its code.  With ONE wavefront per SIMD a taken branch to a cold line of
// the 240 KB kernel is an exposed instruction fetch (csrc/go1sim.hip: the substep loop written out).  GO1_NO_RARE_HINTS: the probe without.
#ifndef GO1_NO_RARE_HINTS
#define GO1_RARE(x) __builtin_expect(!!(x), 0)
#else
#define GO1_RARE(x) (x)
#endif
// Marks a point where the lanes of the wavefront hand data to each other through LDS.  The hardware executes one wave's
// LDS operations in issue order, so no instruction is needed — only the compiler must not move LDS accesses across it.
// (The SIMT emulator of tests/emu defines it as a real barrier: there the lanes do not run in lock step.)
#ifndef LDS_PHASE
#define LDS_PHASE() __builtin_amdgcn_wave_barrier()
#endif
// Pins a value to ONE rounded fp32 register.  Needed where the same value is rounded twice to a narrower type (the fp16
// hi/lo split of the actuator network): left alone the compiler contracts `(f16)(x * r)` into a single-rounding
// v_fma_mix at one use and keeps the double rounding f16(f32(x * r)) at the other, and hi + lo no longer adds up.
#ifndef VALUE_BARRIER
#define VALUE_BARRIER(x) asm volatile("" : "+v"(x))
#endif
// A value that is the same in every lane of the wavefront but that the compiler cannot prove uniform (the wavefront's index
// in its workgroup): as a scalar it keeps loops over it on the scalar unit instead of turning them into masked vector loops.
#ifndef WAVE_UNIFORM
#define WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
// hand-over between the wavefronts of a workgroup (nw of them); with one wavefront it degenerates to the marker above
#define BLOCK_SYNC(nw) do { if ((nw) > 1) __syncthreads(); else LDS_PHASE(); } while (0)

struct V3 { float x, y, z; };
DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
DEV V3 operator*(V3 a, float s) { return v3(s * a.x, s * a.y, s * a.z); }
DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float norm(V3 a) { return sqrtf(dot(a, a)); }

// rotation matrix stored by columns: world = c0*x + c1*y + c2*z
struct M3 { V3 c0, c1, c2; };
DEV V3 mul(const M3& R, V3 v) { return v.x * R.c0 + v.y * R.c1 + v.z * R.c2; }
DEV V3 mulT(const M3& R, V3 v) { return v3(dot(R.c0, v), dot(R.c1, v), dot(R.c2, v)); }
DEV M3 quat_to_mat(float x, float y, float z, float w) {
  M3 R;
  R.c0 = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + z * w), 2.f * (x * z - y * w));
  R.c1 = v3(2.f * (x * y - z * w), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + x * w));
  R.c2 = v3(2.f * (x * z + y * w), 2.f * (y * z - x * w), 1.f - 2.f * (x * x + y * y));
  return R;
}
// R * Rot(axis, angle) for the two joint axes of the Go1 (x: hips, y: thighs/calves)
DEV M3 rot_x(const M3& R, float s, float c) { M3 o; o.c0 = R.c0; o.c1 = c * R.c1 + s * R.c2; o.c2 = c * R.c2 - s * R.c1; return o; }
DEV M3 rot_y(const M3& R, float s, float c) { M3 o; o.c1 = R.c1; o.c0 = c * R.c0 - s * R.c2; o.c2 = s * R.c0 + c * R.c2; return o; }

// quaternion helpers (xyzw)
DEV V3 quat_rotate(float x, float y, float z, float w, V3 v) {
  V3 u = v3(x, y, z);
  V3 t = cross(u, v);
  V3 t2 = cross(u, t);
  return v + (2.f * w) * t + 2.f * t2;
}
DEV V3 quat_rotate_inverse(float x, float y, float z, float w, V3 v) { return quat_rotate(-x, -y, -z, w, v); }

// spatial vectors: motion [angular a; linear l], force [moment a; force l], common reference point
struct SV { V3 a, l; };
DEV SV sv(V3 a, V3 l) { SV r; r.a = a; r.l = l; return r; }
DEV SV operator+(SV p, SV q) { return sv(p.a + q.a, p.l + q.l); }
DEV SV operator-(SV p, SV q) { return sv(p.a - q.a, p.l - q.l); }
DEV SV operator-(SV p) { return sv(-p.a, -p.l); }
DEV SV operator*(float s, SV p) { return sv(s * p.a, s * p.l); }
DEV float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
DEV SV cross_motion(SV v, SV m) { return sv(cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)); }
DEV SV cross_force(SV v, SV f) { return sv(cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)); }

// symmetric 6x6, upper triangle row-major: index(i,j), i<=j
struct Sym6 { float m[21]; };
DEV constexpr int s6(int i, int j) { return i <= j ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j)); }
DEV float sv_get(const SV& v, int i) { return i == 0 ? v.a.x : i == 1 ? v.a.y : i == 2 ? v.a.z : i == 3 ? v.l.x : i == 4 ? v.l.y : v.l.z; }
DEV SV sym6_mul(const Sym6& A, const SV& v) {
  float in[6] = {v.a.x, v.a.y, v.a.z, v.l.x, v.l.y, v.l.z}, out[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) acc = fmaf(A.m[s6(i, j)], in[j], acc);
    out[i] = acc;
  }
  return sv(v3(out[0], out[1], out[2]), v3(out[3], out[4], out[5]));
}
// A -= u u^T * s
DEV void sym6_rank1_sub(Sym6& A, const SV& u, float s) {
  float in[6] = {u.a.x, u.a.y, u.a.z, u.l.x, u.l.y, u.l.z};
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) A.m[s6(i, j)] = fmaf(-s * in[i], in[j], A.m[s6(i, j)]);
}
DEV void sym6_add(Sym6& A, const Sym6& B) {
#pragma unroll
  for (int i = 0; i < 21; i++) A.m[i] += B.m[i];
}
// rigid-body spatial inertia about the reference point: mass m, com c (rel. reference point),
// rotational inertia about the com in reference axes Ic = (xx,xy,xz,yy,yz,zz)
DEV Sym6 rigid_inertia(float m, V3 c, const float Ic[6]) {
  Sym6 I;
  float cc = dot(c, c);
  I.m[s6(0, 0)] = Ic[0] + m * (cc - c.x * c.x);
  I.m[s6(0, 1)] = Ic[1] - m * c.x * c.y;
  I.m[s6(0, 2)] = Ic[2] - m * c.x * c.z;
  I.m[s6(1, 1)] = Ic[3] + m * (cc - c.y * c.y);
  I.m[s6(1, 2)] = Ic[4] - m * c.y * c.z;
  I.m[s6(2, 2)] = Ic[5] + m * (cc - c.z * c.z);
  // upper-right block m [c]x  (rows angular, cols linear)
  I.m[s6(0, 3)] = 0.f;        I.m[s6(0, 4)] = -m * c.z;   I.m[s6(0, 5)] = m * c.y;
  I.m[s6(1, 3)] = m * c.z;    I.m[s6(1, 4)] = 0.f;        I.m[s6(1, 5)] = -m * c.x;
  I.m[s6(2, 3)] = -m * c.y;   I.m[s6(2, 4)] = m * c.x;    I.m[s6(2, 5)] = 0.f;
  I.m[s6(3, 3)] = m; I.m[s6(3, 4)] = 0.f; I.m[s6(3, 5)] = 0.f;
  I.m[s6(4, 4)] = m; I.m[s6(4, 5)] = 0.f;
  I.m[s6(5, 5)] = m;
  return I;
}
// world-axes inertia R Il R^T from body-axes (xx,xy,xz,yy,yz,zz)
DEV void rotate_inertia(const M3& R, const float Il[6], float out[6]) {
  // columns of (R * Il)
  V3 a0 = Il[0] * R.c0 + Il[1] * R.c1 + Il[2] * R.c2;
  V3 a1 = Il[1] * R.c0 + Il[3] * R.c1 + Il[4] * R.c2;
  V3 a2 = Il[2] * R.c0 + Il[4] * R.c1 + Il[5] * R.c2;
  // (R Il) R^T : entry (i,j) = a0_i R.c0_j + a1_i R.c1_j + a2_i R.c2_j
  out[0] = a0.x * R.c0.x + a1.x * R.c1.x + a2.x * R.c2.x;
  out[1] = a0.x * R.c0.y + a1.x * R.c1.y + a2.x * R.c2.y;
  out[2] = a0.x * R.c0.z + a1.x * R.c1.z + a2.x * R.c2.z;
  out[3] = a0.y * R.c0.y + a1.y * R.c1.y + a2.y * R.c2.y;
  out[4] = a0.y * R.c0.z + a1.y * R.c1.z + a2.y * R.c2.z;
  out[5] = a0.z * R.c0.z + a1.z * R.c1.z + a2.z * R.c2.z;
}
// inverse of a symmetric positive definite 6x6 via Cholesky (fully unrolled, registers)
// min_pivot: smallest Cholesky pivot met (<= 0 or NaN means A was not positive definite in fp32: fault site)
// Li (may be null): the rows of L^-1, lower triangle row-major (entry (i, j), j <= i, at i (i + 1) / 2 + j): A^-1 = L^-T L^-1
DEV Sym6 sym6_inverse(const Sym6& A, float& min_pivot, float* Li_out = nullptr) {
  float L[6][6];
  min_pivot = 3.0e38f;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = A.m[s6(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) d = fmaf(-L[j][k], L[j][k], d);
    min_pivot = (d < min_pivot) ? d : (d == d ? min_pivot : d);      // a NaN pivot sticks
    float inv = rsqrtf(d);
    L[j][j] = d * inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float s = A.m[s6(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) s = fmaf(-L[i][k], L[j][k], s);
      L[i][j] = s * inv;
    }
  }
  // Linv (lower)
  float Li[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    Li[j][j] = 1.f / L[j][j];
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float s = 0.f;
#pragma unroll
      for (int k = j; k < i; k++) s = fmaf(-L[i][k], Li[k][j], s);
      Li[i][j] = s / L[i][i];
    }
  }
  if (Li_out) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) Li_out[i * (i + 1) / 2 + j] = Li[i][j];
  }
  Sym6 inv;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i; j < 6; j++) {
      float s = 0.f;
#pragma unroll
      for (int k = j; k < 6; k++) s = fmaf(Li[k][i], Li[k][j], s);
      inv.m[s6(i, j)] = s;
    }
  return inv;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11); same stream definition as oracle/go1_oracle.c
DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
DEV float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ---- quad (4-lane) collectives on DPP: the four lanes of a quad hold the four legs of one environment -----------
DEV float dpp_xor1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false)); }   // quad_perm [1,0,3,2]
DEV float dpp_xor2(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false)); }   // quad_perm [2,3,0,1]
DEV float quad_sum(float x) { x += dpp_xor1(x); x += dpp_xor2(x); return x; }
// value of lane `src` of the quad in all four lanes (quad_perm [s,s,s,s]); `src` folds to a constant after unrolling
DEV float quad_bcast(float x, int src) {
  const int v = __float_as_int(x);
  switch (src & 3) {
    case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, false));
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x55, 0xF, 0xF, false));
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xAA, 0xF, 0xF, false));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, false));
  }
}
DEV SV quad_sum(SV s) {
  return sv(v3(quad_sum(s.a.x), quad_sum(s.a.y), quad_sum(s.a.z)), v3(quad_sum(s.l.x), quad_sum(s.l.y), quad_sum(s.l.z)));
}
// 0 when every argument seen so far is finite, NaN otherwise (x * 0 is NaN for x = +-Inf and NaN)
DEV float nonfinite_acc(float acc, float x) { return fmaf(x, 0.f, acc); }
DEV unsigned quad_or(unsigned x) {
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
  return x;
}
DEV unsigned long long quad_or64(unsigned long long x) {
  return (unsigned long long)quad_or((unsigned)x) | ((unsigned long long)quad_or((unsigned)(x >> 32)) << 32);
}
// value of the lane `rot` (1..3) places further round the quad (quad_perm rotations): lane leg reads lane (leg + rot) & 3
DEV float quad_rot(float x, int rot) {
  const int v = __float_as_int(x);
  switch (rot & 3) {
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x39, 0xF, 0xF, false));      // [1, 2, 3, 0]
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));      // [2, 3, 0, 1]
    case 3: return __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x93, 0xF, 0xF, false));      // [3, 0, 1, 2]
    default: return x;
  }
}
DEV unsigned quad_ballot(bool p, int lane) { return (unsigned)((__ballot(p) >> (lane & ~3)) & 0xFull); }
