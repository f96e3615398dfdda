// go1ppo.hip — fused kernels of the PPO update for gfx950 (C-ABI in include/go1ppo.h).
//
// The update of go1_gym_learn/ppo_cse/ppo.py:99-205 is a chain of small-MLP GEMMs (hipBLASLt, through torch.mm on
// static buffers) glued by element-wise maps and reductions.  Those maps are the kernels of this file:
//   elu_fwd   ELU (+ the actor's latent columns) in place on a column block of the fused first-layer output
//   elu_bwd   dZ = dH * elu'(H) in place, with the bias gradient (column sums) accumulated on the way
//   loss      surrogate / clipped value / entropy loss + KL, forward AND the analytic gradient w.r.t. mean, value,
//             std and the head biases, gathering the rollout-storage rows through the mini-batch index
//   mse       adaptation-module regression loss and gradient
//   wgrad     dW = dZ^T H for the small layers (n, k <= 512, 24576-row reduction): bf16 MFMA 16x16x32 with the
//             operands transposed through LDS, split over row chunks, fp32 atomic accumulation; all layers of a
//             backward pass in one batched launch.  hipBLASLt serves these shapes at 90-135 us each.
//   tail_fwd  the MLP layers behind the first one with the activations kept on chip (rollout inference)
//   act / store_step / gae / normalize / opt_prestep / opt_adam: rollout glue and the optimiser step
// bf16 activations, fp32 math, fp32 parameter gradients.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/go1ppo.h"

typedef uint16_t bf16_t;
static_assert(sizeof(Go1PpoAdamExtras) == 56 && sizeof(Go1PpoGemmArgs) == 96 && sizeof(Go1PpoWgradProblem) == 96 && sizeof(Go1PpoGradPiece) == 56, "ctypes mirrors (fused.py) assume these sizes");
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {          // round to nearest even: v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
struct alignas(16) Bf8 { bf16_t v[8]; };

// ELU with alpha = 1.  exp(x) - 1 through the hardware exponential loses relative accuracy only for |x| < ~1e-3, where the
// two-term series is exact to fp32; both are far inside the bf16 rounding of the stored activation.
__device__ __forceinline__ float elu1(float x) {
  float em = x > -1e-3f ? fmaf(0.5f * x, x, x) : __expf(x) - 1.f;
  return x > 0.f ? x : em;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// eight floats -> eight bf16 (round to nearest even) on v_cvt_pk_bf16_f32
__device__ __forceinline__ Bf8 pack_bf8(const float (&x)[8]) {
  union { Bf8 b; uint32_t u[4]; } r;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    f32x2 v = {x[2 * e], x[2 * e + 1]};
    r.u[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  return r.b;
}

// ---------------------------------------------------------------------------------------------- elu_fwd
// block = 16 column groups (8 columns each) x 16 row lanes, ELU_ROWS rows per lane: a 128-column x 64-row tile.
#ifndef ELU_ROWS
#define ELU_ROWS 4
#endif
template <bool LAT>
__global__ __launch_bounds__(256) void elu_fwd_kernel(bf16_t* y, int rows, int cols, int ld, const bf16_t* lat, int lat_ld,
                                                      int npv, const bf16_t* wz, int wz_ld, int lat_cols) {
  int c0 = (blockIdx.x * 16 + (threadIdx.x & 15)) << 3;
  int r0 = blockIdx.y * (16 * ELU_ROWS) + (threadIdx.x >> 4);
  if (c0 >= cols) return;
  Bf8 v[ELU_ROWS];
#pragma unroll
  for (int u = 0; u < ELU_ROWS; u++) {
    int r = r0 + u * 16;
    if (r < rows) v[u] = *reinterpret_cast<const Bf8*>(y + (int64_t)r * ld + c0);
  }
  float x[ELU_ROWS][8];
#pragma unroll
  for (int u = 0; u < ELU_ROWS; u++)
#pragma unroll
    for (int e = 0; e < 8; e++) x[u][e] = bf2f(v[u].v[e]);
  if (LAT && c0 < lat_cols) {
    if (npv == 2) {                      // the train.py configuration: both latent columns in one 4-byte load
      float w0[8], w1[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const uint32_t p = *reinterpret_cast<const uint32_t*>(wz + (int64_t)(c0 + e) * wz_ld);
        w0[e] = __uint_as_float(p << 16);
        w1[e] = __uint_as_float(p & 0xffff0000u);
      }
#pragma unroll
      for (int u = 0; u < ELU_ROWS; u++) {
        int r = r0 + u * 16;
        const uint32_t p = r < rows ? *reinterpret_cast<const uint32_t*>(lat + (int64_t)r * lat_ld) : 0u;
        const float l0 = __uint_as_float(p << 16), l1 = __uint_as_float(p & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 8; e++) x[u][e] = fmaf(l1, w1[e], fmaf(l0, w0[e], x[u][e]));
      }
    } else {
      for (int q = 0; q < npv; q++) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; e++) w[e] = bf2f(wz[(int64_t)(c0 + e) * wz_ld + q]);
#pragma unroll
        for (int u = 0; u < ELU_ROWS; u++) {
          int r = r0 + u * 16;
          float l = r < rows ? bf2f(lat[(int64_t)r * lat_ld + q]) : 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) x[u][e] = fmaf(l, w[e], x[u][e]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < ELU_ROWS; u++) {
    int r = r0 + u * 16;
#pragma unroll
    for (int e = 0; e < 8; e++) x[u][e] = elu1(x[u][e]);
    if (r < rows) *reinterpret_cast<Bf8*>(y + (int64_t)r * ld + c0) = pack_bf8(x[u]);
  }
}

// ---------------------------------------------------------------------------------------------- elu_bwd (+ column sums)
// block = 256 threads = CG column groups (8 columns each) x 256/CG row lanes; one block covers `rchunk` rows.
__global__ __launch_bounds__(256) void elu_bwd_kernel(const bf16_t* d, int ld_d, const bf16_t* h, int ld_h, int64_t rows, int cols,
                                                      float* bias_grad, bf16_t* out, int ld_out, int cg_per_block, int rchunk) {
  __shared__ float red[256 * 8];
  int cgi = threadIdx.x % cg_per_block, rl = threadIdx.x / cg_per_block, rlanes = 256 / cg_per_block;
  int c0 = (blockIdx.x * cg_per_block + cgi) << 3;
  int64_t r0 = (int64_t)blockIdx.y * rchunk;
  int64_t r1 = r0 + rchunk < rows ? r0 + rchunk : rows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < cols) {
    for (int64_t r = r0 + rl; r < r1; r += rlanes) {
      Bf8 g = *reinterpret_cast<const Bf8*>(d + r * ld_d + c0);
      Bf8* po = reinterpret_cast<Bf8*>(out + r * ld_out + c0);
      if (h) {
        Bf8 a = *reinterpret_cast<const Bf8*>(h + r * ld_h + c0);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float hv = bf2f(a.v[e]);
          float dz = bf2f(g.v[e]) * (hv > 0.f ? 1.f : hv + 1.f);
          g.v[e] = f2bf(dz);
          s[e] += bf2f(g.v[e]);
        }
        *po = g;
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) s[e] += bf2f(g.v[e]);
        if (out != d) *po = g;
      }
    }
  }
  if (!bias_grad) return;
#pragma unroll
  for (int e = 0; e < 8; e++) red[(rl * cg_per_block + cgi) * 8 + e] = s[e];
  __syncthreads();
  for (int o = threadIdx.x; o < cg_per_block * 8; o += 256) {
    int c = (blockIdx.x * cg_per_block << 3) + o;
    if (c >= cols) continue;
    float t = 0.f;
    for (int q = 0; q < rlanes; q++) t += red[q * cg_per_block * 8 + o];
    atomicAdd(bias_grad + c, t);
  }
}

// the same map without column sums (the weight-gradient kernel produces the bias gradients): elu_fwd's tiling
__global__ __launch_bounds__(256) void elu_bwd_tile_kernel(const bf16_t* d, int ld_d, const bf16_t* h, int ld_h, int rows, int cols,
                                                           bf16_t* out, int ld_out) {
  int c0 = (blockIdx.x * 16 + (threadIdx.x & 15)) << 3;
  int r0 = blockIdx.y * (16 * ELU_ROWS) + (threadIdx.x >> 4);
  if (c0 >= cols) return;
  Bf8 g[ELU_ROWS], a[ELU_ROWS];
#pragma unroll
  for (int u = 0; u < ELU_ROWS; u++) {
    int r = r0 + u * 16;
    if (r < rows) {
      g[u] = *reinterpret_cast<const Bf8*>(d + (int64_t)r * ld_d + c0);
      a[u] = *reinterpret_cast<const Bf8*>(h + (int64_t)r * ld_h + c0);
    }
  }
#pragma unroll
  for (int u = 0; u < ELU_ROWS; u++) {
    int r = r0 + u * 16;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float hv = bf2f(a[u].v[e]);
      x[e] = bf2f(g[u].v[e]) * (hv > 0.f ? 1.f : hv + 1.f);
    }
    if (r < rows) *reinterpret_cast<Bf8*>(out + (int64_t)r * ld_out + c0) = pack_bf8(x);
  }
}

// ---------------------------------------------------------------------------------------------- block reduction helper
template <int NV>
__device__ __forceinline__ void block_reduce_atomic(float (&v)[NV], float* const (&dst)[NV], float* lds /* [4][NV] */) {
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float x = v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    v[i] = x;
  }
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    for (int i = 0; i < NV; i++) lds[wave * NV + i] = v[i];
  __syncthreads();
  if (threadIdx.x < NV) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += lds[w * NV + threadIdx.x];
#if defined(LOSS_ABLATE) && (LOSS_ABLATE & 8)
    if (dst[threadIdx.x] && t == 12345.678f) *dst[threadIdx.x] = t;      // probe build: the reduction without its atomics
#else
    if (dst[threadIdx.x]) atomicAdd(dst[threadIdx.x], t);
#endif
  }
}

// ---------------------------------------------------------------------------------------------- PPO loss
// One thread per sample.  Everything the autograd graph of ppo.py:112-150 computes, in closed form:
//   logp = -1/2 sum z^2 - sum log sigma - A c,  z = (a - mu)/sigma;   ratio = exp(logp - logp_old)
//   surrogate = mean max(-adv ratio, -adv clamp(ratio, 1-eps, 1+eps))
//   d surrogate / d logp = -adv ratio / M where the un-clamped branch is active (inside the clip range both
//   branches coincide and torch.max splits the gradient half/half onto two identical paths), else 0
//   value loss: max((v-R)^2, (v_old + clamp(v - v_old, -eps, eps) - R)^2), same tie rule
//   entropy = sum log sigma + const (independent of the sample)
#if defined(LOSS_ABLATE) && (LOSS_ABLATE & 2)
#define LOSS_LOG(x) __logf(x)
#define LOSS_EXP(x) __expf(x)
#else
#define LOSS_LOG(x) logf(x)
#define LOSS_EXP(x) expf(x)
#endif
#if defined(LOSS_ABLATE) && (LOSS_ABLATE & 4)
#define LOSS_NO_STORES 1
#else
#define LOSS_NO_STORES 0
#endif
// one sample: losses and the analytic gradient w.r.t. its head outputs (written to d_mean / d_value); the sample's shares of the sums
// come back in sur / vl / kl / dvb / dstd[] / dmb[] (zero for a thread past the last row)
__device__ __forceinline__ void loss_sample(const Go1PpoLossArgs& a, int64_t r, float& sur, float& vl, float& kl, float& dvb,
                                            float (&dstd)[GO1PPO_MAX_ACTIONS], float (&dmb)[GO1PPO_MAX_ACTIONS]) {
  const int A = a.num_actions;
  const bool vec4 = (A & 3) == 0 && (a.head_ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.actions) | reinterpret_cast<uintptr_t>(a.old_mu) |
                                                              reinterpret_cast<uintptr_t>(a.old_sigma)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.mean) & 7) == 0;
  bool on = r < a.rows;
  float invM = 1.f / (float)a.rows;
  sur = vl = kl = dvb = 0.f;
#pragma unroll
  for (int j = 0; j < GO1PPO_MAX_ACTIONS; j++) dstd[j] = dmb[j] = 0.f;
  if (on) {
    int64_t s = a.idx[r];
#if defined(LOSS_ABLATE) && (LOSS_ABLATE & 1)
    s = r;                       // probe build (tools/probes/loss_ablate.py): storage rows in order instead of through the permutation
#endif
    const bf16_t* mrow = reinterpret_cast<const bf16_t*>(a.mean) + r * a.head_ld;
    float mu[GO1PPO_MAX_ACTIONS], z[GO1PPO_MAX_ACTIONS], isg[GO1PPO_MAX_ACTIONS];
    float act[GO1PPO_MAX_ACTIONS], omu[GO1PPO_MAX_ACTIONS], osg[GO1PPO_MAX_ACTIONS];
    float logp = 0.f;
    const float HALF_LOG_2PI = 0.9189385332046727f;
    if (vec4) {
      // the storage rows of a sample (A fp32 each, A % 4 == 0: 16-byte aligned) and its bf16 head row as 16 / 8-byte gathers: a wave's
      // 64 samples are 64 different cache lines per load instruction, so the instruction COUNT (4 per element-wise load) is what the
      // texture path's time goes with
#pragma unroll
      for (int j = 0; j < GO1PPO_MAX_ACTIONS; j += 4) {
        if (j < A) {
          const f32x4 va = *reinterpret_cast<const f32x4*>(a.actions + s * A + j), vm = *reinterpret_cast<const f32x4*>(a.old_mu + s * A + j),
                      vs = *reinterpret_cast<const f32x4*>(a.old_sigma + s * A + j);
          const uint2 mr = *reinterpret_cast<const uint2*>(mrow + j);
          mu[j] = __uint_as_float(mr.x << 16); mu[j + 1] = __uint_as_float(mr.x & 0xffff0000u);
          mu[j + 2] = __uint_as_float(mr.y << 16); mu[j + 3] = __uint_as_float(mr.y & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 4; e++) { act[j + e] = va[e]; omu[j + e] = vm[e]; osg[j + e] = vs[e]; }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < GO1PPO_MAX_ACTIONS; j++) {
        if (j < A) { mu[j] = bf2f(mrow[j]); act[j] = a.actions[s * A + j]; omu[j] = a.old_mu[s * A + j]; osg[j] = a.old_sigma[s * A + j]; }
      }
    }
#pragma unroll
    for (int j = 0; j < GO1PPO_MAX_ACTIONS; j++) {
      if (j < A) {
        float sg = a.std[j];
        isg[j] = 1.f / sg;
        z[j] = (act[j] - mu[j]) * isg[j];
        logp += -0.5f * z[j] * z[j] - LOSS_LOG(sg) - HALF_LOG_2PI;
        float so = osg[j], dm = omu[j] - mu[j];
        kl += LOSS_LOG(sg / so + 1.e-5f) + (so * so + dm * dm) / (2.f * sg * sg) - 0.5f;
      }
    }
    float adv = a.advantages[s];
    float ratio = LOSS_EXP(logp - a.old_logp[s]);
    float lo = 1.f - a.clip_param, hi = 1.f + a.clip_param;
    float rc = fminf(fmaxf(ratio, lo), hi);
    float s1 = -adv * ratio, s2 = -adv * rc;
    sur = fmaxf(s1, s2) * invM;
    float dratio = (ratio >= lo && ratio <= hi) ? -adv : (s1 > s2 ? -adv : 0.f);
    float dlogp = dratio * ratio * invM;
    bf16_t* drow = reinterpret_cast<bf16_t*>(a.d_mean) + r * a.head_ld;
    const bool st4 = vec4 && (reinterpret_cast<uintptr_t>(a.d_mean) & 7) == 0;      // four gradients per 8-byte store (12 two-byte stores per
                                                                                    // sample were 2.6-4.9 us of the kernel: r06_loss_kernel_ablation.txt)
#pragma unroll
    for (int j0 = 0; j0 < GO1PPO_MAX_ACTIONS; j0 += 4) {
      if (j0 < A) {
        bf16_t gq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = j0 + q;
          if (j < A) {
            gq[q] = f2bf(dlogp * z[j] * isg[j]);
            if (!LOSS_NO_STORES && !st4) drow[j] = gq[q];
            dmb[j] = bf2f(gq[q]);
            dstd[j] = dlogp * (z[j] * z[j] - 1.f) * isg[j];
          }
        }
        if (!LOSS_NO_STORES && st4) {
          uint2 o;
          o.x = (uint32_t)gq[0] | ((uint32_t)gq[1] << 16);
          o.y = (uint32_t)gq[2] | ((uint32_t)gq[3] << 16);
          *reinterpret_cast<uint2*>(drow + j0) = o;
        }
      }
    }
    float v = bf2f(reinterpret_cast<const bf16_t*>(a.value)[r * a.head_ld]);
    float R = a.returns[s], dv;
    if (a.use_clipped_value_loss) {
      float vo = a.old_values[s];
      float dlt = v - vo;
      float vc = vo + fminf(fmaxf(dlt, -a.clip_param), a.clip_param);
      float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
      vl = fmaxf(l1, l2) * invM;
      bool inside = dlt >= -a.clip_param && dlt <= a.clip_param;
      dv = inside ? 2.f * (v - R) : (l1 > l2 ? 2.f * (v - R) : 0.f);
    } else {
      vl = (R - v) * (R - v) * invM;
      dv = 2.f * (v - R);
    }
    bf16_t g = f2bf(a.value_loss_coef * dv * invM);
    if (!LOSS_NO_STORES) reinterpret_cast<bf16_t*>(a.d_value)[r * a.head_ld] = g;
    dvb = bf2f(g);
    kl *= invM;
  }
}

__global__ __launch_bounds__(256) void loss_kernel(Go1PpoLossArgs a) {
  __shared__ float lds[4 * (4 + 2 * GO1PPO_MAX_ACTIONS)];
  const int A = a.num_actions;
  float sur, vl, kl, dvb;
  float dstd[GO1PPO_MAX_ACTIONS], dmb[GO1PPO_MAX_ACTIONS];
  loss_sample(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, sur, vl, kl, dvb, dstd, dmb);
  // reductions: [sur, vl, kl, dvb] then dstd[A], dmb[A]
  {
    float v4[4] = {sur, vl, kl, dvb};
    float* const d4[4] = {a.surrogate_loss, a.value_loss, a.kl, a.d_value_bias};
    block_reduce_atomic<4>(v4, d4, lds);
  }
  __syncthreads();
#pragma unroll
  for (int j0 = 0; j0 < GO1PPO_MAX_ACTIONS; j0 += 4) {
    if (j0 < A) {      // uniform
      float v8[8];
      float* d8[8];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        bool ok = j0 + q < A;
        v8[q] = ok ? dstd[j0 + q] : 0.f;
        v8[4 + q] = ok ? dmb[j0 + q] : 0.f;
        d8[q] = ok ? a.d_std + j0 + q : nullptr;
        d8[4 + q] = ok ? a.d_mean_bias + j0 + q : nullptr;
      }
      float* const d8c[8] = {d8[0], d8[1], d8[2], d8[3], d8[4], d8[5], d8[6], d8[7]};
      block_reduce_atomic<8>(v8, d8c, lds);
      __syncthreads();
    }
  }
  // entropy term: loss -= entropy_coef * (sum log sigma + const)  ->  d/d sigma_j = -entropy_coef / sigma_j  (once)
  if (blockIdx.x == 0 && threadIdx.x < A) atomicAdd(a.d_std + threadIdx.x, -a.entropy_coef / a.std[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------- adaptation MSE
__global__ __launch_bounds__(256) void mse_kernel(const bf16_t* pred, int pred_ld, const float* target, int npv, const int64_t* idx,
                                                  int64_t rows, int64_t num_train, int selective, bf16_t* d_pred,
                                                  float* d_pred_bias, float* train_loss, float* test_loss) {
  __shared__ float lds[4 * 8];
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int ncol = selective ? 1 : npv;
  float ltrain = 0.f, ltest = 0.f;
  float inv_train = 1.f / ((float)num_train * ncol), inv_test = 1.f / ((float)(rows - num_train) * ncol);
  for (int j0 = 0; j0 < ncol; j0 += 6) {        // 6 columns of bias gradient per reduction round
    float v8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rows) {
      int64_t s = idx[r];
      for (int q = 0; q < 6 && j0 + q < ncol; q++) {
        int j = j0 + q;
        float e = bf2f(pred[r * pred_ld + j]) - target[s * npv + j];
        if (r < num_train) {
          ltrain += e * e * inv_train;
          bf16_t g = f2bf(2.f * e * inv_train);
          d_pred[r * pred_ld + j] = g;
          v8[q] = bf2f(g);
        } else {
          ltest += e * e * inv_test;
          d_pred[r * pred_ld + j] = 0;
        }
      }
    }
    bool last = j0 + 6 >= ncol;
    v8[6] = last ? ltrain : 0.f;
    v8[7] = last ? ltest : 0.f;
    float* d8[8];
    for (int q = 0; q < 6; q++) d8[q] = (j0 + q < ncol) ? d_pred_bias + j0 + q : nullptr;
    d8[6] = last ? train_loss : nullptr;
    d8[7] = last ? test_loss : nullptr;
    float* const d8c[8] = {d8[0], d8[1], d8[2], d8[3], d8[4], d8[5], d8[6], d8[7]};
    block_reduce_atomic<8>(v8, d8c, lds);
    __syncthreads();
  }
  if (selective && r < rows)      // columns the selective loss ignores carry no gradient
    for (int j = 1; j < npv; j++) d_pred[r * pred_ld + j] = 0;
}

// ---------------------------------------------------------------------------------------------- wgrad (MFMA, split rows)
// dW[n0+i][k0+j] += sum_m dz[m][n0+i] h[m][k0+j] for a 64x64 output tile and one row chunk per workgroup.
// MFMA 16x16x32 bf16 wants, per lane, 8 consecutive reduction indices (m) for one output row/column, but both
// operands are m-major in memory; every 64-row step is therefore transposed on its way into LDS (T[op][n|k][m],
// rows padded to 72 elements = 144 B: 16-byte aligned b128 fragment reads).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
// Every thread loads a 4-row x 8-column patch of one operand and stores its transpose with eight 8-byte LDS writes
// (4 consecutive reduction indices each).
#define WG2_LDM 72
__device__ __forceinline__ void wgrad2_body(const bf16_t* dz, int ld_dz, const bf16_t* h, int ld_h, int64_t m_begin, int64_t m_end,
                                            float* dW, int ldw, float* bias_grad, int n0, int k0, bool do_bias,
                                            bf16_t (*T)[2][64][WG2_LDM], int zero_n = 0, int zero_k0 = 0, int zero_k1 = 0) {
  const int t = threadIdx.x, op = t >> 7, rq = (t & 127) >> 3, cg = t & 7, wave = t >> 6, lane = t & 63;
  const bf16_t* src = op ? h + k0 + 8 * cg : dz + n0 + 8 * cg;
  const int ld = op ? ld_h : ld_dz;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 r[4];
  auto gload = [&](int64_t m) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int64_t row = m + 4 * rq + q;
      r[q] = row < m_end ? *reinterpret_cast<const uint4*>(src + row * ld) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
    const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&r[0]);
    const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&r[1]);
    const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&r[2]);
    const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&r[3]);
#pragma unroll
    for (int p = 0; p < 4; p++) {        // dword p holds columns 2p, 2p+1 of each of the 4 rows
      uint2 even, odd;
      even.x = (w0[p] & 0xffffu) | (w1[p] << 16);
      even.y = (w2[p] & 0xffffu) | (w3[p] << 16);
      odd.x = (w0[p] >> 16) | (w1[p] & 0xffff0000u);
      odd.y = (w2[p] >> 16) | (w3[p] & 0xffff0000u);
      *reinterpret_cast<uint2*>(&T[buf][op][8 * cg + 2 * p][4 * rq]) = even;
      *reinterpret_cast<uint2*>(&T[buf][op][8 * cg + 2 * p + 1][4 * rq]) = odd;
    }
  };
  if (m_begin >= m_end) return;
  float bsum = 0.f;
  gload(m_begin);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int64_t m = m_begin; m < m_end; m += 64) {
    bool more = m + 64 < m_end;
    if (more) gload(m + 64);
    if (do_bias) {                                           // thread t: row n = t/4 of the tile, 16 of the 64 m's
      const Bf8* q = reinterpret_cast<const Bf8*>(&T[buf][0][t >> 2][16 * (t & 3)]);
      Bf8 u0 = q[0], u1 = q[1];
#pragma unroll
      for (int e = 0; e < 8; e++) bsum += bf2f(u0.v[e]) + bf2f(u1.v[e]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(&T[buf][0][16 * wave + (lane & 15)][ks * 32 + (lane >> 4) * 8]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bf16x8_t bv = *reinterpret_cast<const bf16x8_t*>(&T[buf][1][16 * j + (lane & 15)][ks * 32 + (lane >> 4) * 8]);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[j], 0, 0, 0);
      }
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int row = n0 + 16 * wave + (lane >> 4) * 4 + q, col = k0 + 16 * j + (lane & 15);
      if (!(row < zero_n && col >= zero_k0 && col < zero_k1)) atomicAdd(dW + (int64_t)row * ldw + col, acc[j][q]);
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    if ((t & 3) == 0) atomicAdd(bias_grad + n0 + (t >> 2), bsum);
  }
}

__global__ __launch_bounds__(256) void wgrad2_kernel(const bf16_t* dz, int ld_dz, const bf16_t* h, int ld_h, int64_t rows,
                                                     int chunk_rows, float* dW, int ldw, float* bias_grad) {
  __shared__ __attribute__((aligned(16))) bf16_t T[2][2][64][WG2_LDM];      // [buffer][operand][n or k][m]
  const int64_t m_begin = (int64_t)blockIdx.z * chunk_rows;
  const int64_t m_end = m_begin + chunk_rows < rows ? m_begin + chunk_rows : rows;
  wgrad2_body(dz, ld_dz, h, ld_h, m_begin, m_end, dW, ldw, bias_grad, blockIdx.x * 64, blockIdx.y * 64, bias_grad && blockIdx.y == 0, T);
}

// every weight gradient of a backward pass in ONE launch: the layers' (n/64)(k/64) output tiles x row chunks are
// independent work items; alone, each layer is too small to fill 256 CUs (2..32 tiles).
__global__ __launch_bounds__(256) void wgrad_batched_kernel(const Go1PpoWgradProblem* __restrict__ probs, int count) {
  __shared__ __attribute__((aligned(16))) bf16_t T[2][2][64][WG2_LDM];
  int p = 0;
  const int wg = blockIdx.x;
  while (p + 1 < count && wg >= probs[p + 1].wg_offset) p++;
  const Go1PpoWgradProblem P = probs[p];
  const int local = wg - P.wg_offset;
  const int tiles_k = P.k / 64, tiles = (P.n / 64) * tiles_k;
  const int tile = local % tiles, split = local / tiles;
  const int n0 = (tile / tiles_k) * 64, k0 = (tile % tiles_k) * 64;
  const int64_t m_begin = (int64_t)split * P.chunk_rows;
  const int64_t m_end = m_begin + P.chunk_rows < P.rows ? m_begin + P.chunk_rows : P.rows;
  wgrad2_body((const bf16_t*)P.dz, P.ld_dz, (const bf16_t*)P.h, P.ld_h, m_begin, m_end, P.dW, P.ldw, P.bias_grad, n0, k0,
              P.bias_grad && k0 == 0, T, P.zero_n, P.zero_k0, P.zero_k1);
}

// ---------------------------------------------------------------------------------------------- rollout glue
// PPO.act (ppo.py:60-77): sample a = mu + sigma * noise, log-prob, and write the transition's policy outputs
// straight into the rollout-storage slot.  One thread per environment.
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* mean, const bf16_t* value, int head_ld, const float* std, int A,
                                                  int64_t rows, const float* noise, float* actions, float* mu, float* sigma,
                                                  float* values, float* logp) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float HALF_LOG_2PI = 0.9189385332046727f;
  float lp = 0.f, logs = 0.f;
  for (int j = 0; j < A; j++) {
    float sg = std[j], m = bf2f(mean[r * head_ld + j]);
    float a = m + sg * noise[r * A + j];
    float z = (a - m) / sg;
    lp += -0.5f * z * z;
    logs += logf(sg);
    actions[r * A + j] = a;
    mu[r * A + j] = m;
    sigma[r * A + j] = sg;
  }
  logp[r] = lp - (logs + A * HALF_LOG_2PI);
  values[r] = bf2f(value[r * head_ld]);
}

// PPO.process_env_step (ppo.py:79-91) + RolloutStorage.add_transitions: reward with the time-out bootstrap
// r + gamma * V * time_out, done flags and curriculum bins into the storage slot.
__global__ __launch_bounds__(256) void store_step_kernel(const float* rewards, const uint8_t* dones, const uint8_t* time_outs,
                                                         const int32_t* env_bins, const float* values, float gamma, int64_t n,
                                                         float* rewards_out, uint8_t* dones_out, float* env_bins_out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = rewards[i];
  if (time_outs) r += gamma * (values[i] * (time_outs[i] ? 1.f : 0.f));
  rewards_out[i] = r;
  dones_out[i] = dones[i];
  if (env_bins) env_bins_out[i] = (float)env_bins[i];
}

// RolloutStorage.compute_returns (rollout_storage.py:72-84): backward GAE scan, one thread per environment, and the
// sums the advantage normalisation needs (double accumulators; reduced over ranks by the caller when envs are sharded).
__global__ __launch_bounds__(256) void gae_kernel(const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
                                                  int T, int64_t N, float gamma, float lam, float* returns, float* advantages, double* stats) {
  __shared__ double red[2][4];
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (e < N) {
    float adv = 0.f, next_v = last_values[e];
    for (int t = T - 1; t >= 0; t--) {
      const int64_t i = (int64_t)t * N + e;
      const float alive = 1.f - (float)dones[i];
      const float v = values[i];
      const float delta = rewards[i] + alive * gamma * next_v - v;
      adv = delta + alive * gamma * lam * adv;
      const float ret = adv + v;
      const float a = ret - v;
      returns[i] = ret;
      advantages[i] = a;
      s1 += a;
      s2 += (double)a * a;
      next_v = v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(stats + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// advantages = (advantages - mean) / (std + 1e-8), unbiased std from stats = [sum, sum of squares, count]
__global__ __launch_bounds__(256) void normalize_kernel(float* adv, int64_t n, const double* stats) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double cnt = stats[2], mean = stats[0] / cnt;
  double var = (stats[1] - cnt * mean * mean) / (cnt - 1.0);
  var = var > 0.0 ? var : 0.0;
  const float m = (float)mean, sd = (float)sqrt(var);
  adv[i] = (adv[i] - m) / (sd + 1e-8f);
}

// ---------------------------------------------------------------------------------------------- optimiser step
// PPO.update's step (ppo.py:126-160): KL-adaptive learning rate, global-norm clip, Adam, and the refresh of the bf16
// compute copy — two kernels instead of ~25.
//   prestep: per-block partial sums of g^2 (no atomics: deterministic), and by one thread the scalar bookkeeping
//            that the second kernel only reads: step += 1, lr <- schedule(kl, lr).
//   adam:    p, m, v update over up to two element ranges with the clip factor from the partials; writes the bf16
//            copy of the updated parameters (and the fp32 tail, the action std) on the way.
#define OPT_BLOCKS 2048
// ---- the flat gradient as pieces (include/go1ppo.h Go1PpoGradPiece): slab pieces are summed in a fixed order (slab 0, 1, ...) into g;
// NORM: the squared norm of (g * gscale) over [0, n) — plain elements and the fresh sums alike — is returned per thread
// One flat index space of 8-element groups over all segments (plain gap, piece, plain gap, ...): a lane's group lies in exactly one segment, so a
// wave makes ONE memory round trip — with a loop per segment the first blocks walked through all ~20 segments one after the other (22 us for 51 MB).
template <bool NORM>
__device__ __forceinline__ float grad_pieces_pass(float* g, int64_t n, const Go1PpoGradPiece* __restrict__ pieces, int num_pieces, float gscale) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  auto gap_groups = [&](int64_t lo, int64_t hi) -> int64_t { return NORM && hi > lo ? (hi - lo + 7) >> 3 : 0; };
  auto piece_groups = [&](const Go1PpoGradPiece& P) -> int64_t { return P.kind == 0 ? gap_groups(P.begin, P.begin + P.count) : P.count >> 3; };
  int64_t total = 0;
  {
    int64_t cur = 0;
    for (int q = 0; q <= num_pieces; q++) {
      const int64_t pb = q < num_pieces ? pieces[q].begin : n;
      total += gap_groups(cur, pb < n ? pb : n);
      if (q == num_pieces) break;
      total += piece_groups(pieces[q]);
      cur = pieces[q].begin + pieces[q].count;
    }
  }
  float s = 0.f;
  auto plain8 = [&](int64_t i, int64_t hi) {                      // g[i, min(i + 8, hi)) into the norm; i is a multiple of 8
    if (i + 8 <= hi) {
      const f32x4 a = reinterpret_cast<const f32x4*>(g + i)[0] * gscale, b = reinterpret_cast<const f32x4*>(g + i)[1] * gscale;
#pragma unroll
      for (int e = 0; e < 4; e++) { s = fmaf(a[e], a[e], s); s = fmaf(b[e], b[e], s); }
    } else {
      for (int64_t j = i; j < hi; j++) { const float x = g[j] * gscale; s = fmaf(x, x, s); }
    }
  };
  for (int64_t f = tid; f < total; f += nth) {
    int64_t cum = 0, cur = 0;
    for (int q = 0; q <= num_pieces; q++) {
      const int64_t pb = q < num_pieces ? pieces[q].begin : n, ge = pb < n ? pb : n;
      const int64_t gg = gap_groups(cur, ge);
      if (f >= cum && f < cum + gg) plain8(cur + ((f - cum) << 3), ge);
      cum += gg;
      if (q == num_pieces) break;
      const Go1PpoGradPiece P = pieces[q];
      const int64_t pg = piece_groups(P);
      cur = P.begin + P.count;
      if (f >= cum && f < cum + pg) {
        const int64_t i = (f - cum) << 3;
        if (P.kind == 0) plain8(P.begin + i, cur);
        else {
          float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (P.kind == 1) {
            const float* src = (const float*)P.src + i;
            int b = 0;
            for (; b + 4 <= P.slabs; b += 4) {                        // four slabs' loads in flight (the sums stay in slab order)
              f32x4 lo[4], hi[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                lo[u] = reinterpret_cast<const f32x4*>(src + (int64_t)(b + u) * P.stride)[0];
                hi[u] = reinterpret_cast<const f32x4*>(src + (int64_t)(b + u) * P.stride)[1];
              }
#pragma unroll
              for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 4; e++) { acc[e] += lo[u][e]; acc[4 + e] += hi[u][e]; }
            }
            for (; b < P.slabs; b++) {
              const f32x4 lo = reinterpret_cast<const f32x4*>(src + (int64_t)b * P.stride)[0], hi = reinterpret_cast<const f32x4*>(src + (int64_t)b * P.stride)[1];
#pragma unroll
              for (int e = 0; e < 4; e++) { acc[e] += lo[e]; acc[4 + e] += hi[e]; }
            }
          } else {
            const bf16_t* src = (const bf16_t*)P.src + i;
            int b = 0;
            for (; b + 4 <= P.slabs; b += 4) {
              Bf8 v[4];
#pragma unroll
              for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const Bf8*>(src + (int64_t)(b + u) * P.stride);
#pragma unroll
              for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[e] += bf2f(v[u].v[e]);
            }
            for (; b < P.slabs; b++) {
              const Bf8 v = *reinterpret_cast<const Bf8*>(src + (int64_t)b * P.stride);
#pragma unroll
              for (int e = 0; e < 8; e++) acc[e] += bf2f(v.v[e]);
            }
          }
          if (P.zero_rows > 0) {
            const int64_t row = i / P.cols;
            const int c = (int)(i - row * P.cols);
            if (row < P.zero_rows && c < P.zero_c1 && c + 8 > P.zero_c0) {
#pragma unroll
              for (int e = 0; e < 8; e++)
                if (c + e >= P.zero_c0 && c + e < P.zero_c1) acc[e] = 0.f;
            }
          }
          float* out = g + P.begin + i;
          reinterpret_cast<f32x4*>(out)[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
          reinterpret_cast<f32x4*>(out)[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
          if (NORM) {
#pragma unroll
            for (int e = 0; e < 8; e++) { const float x = acc[e] * gscale; s = fmaf(x, x, s); }
          }
        }
      }
      cum += pg;
    }
  }
  return s;
}
__global__ __launch_bounds__(256) void grad_reduce_kernel(float* g, const Go1PpoGradPiece* __restrict__ pieces, int num_pieces) {
  grad_pieces_pass<false>(g, 0, pieces, num_pieces, 1.f);
}

__global__ __launch_bounds__(256) void prestep_kernel(const float* g, int64_t n, float gscale, float* partial, float* step, float* lr,
                                                      const float* kl, float kl_scale, float desired_kl, float lr_min, float lr_max,
                                                      const Go1PpoGradPiece* __restrict__ pieces, int num_pieces) {
  __shared__ float red[4];
  float s = 0.f;
  if (partial && pieces) {
    s = grad_pieces_pass<true>(const_cast<float*>(g), n, pieces, num_pieces, gscale);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  } else if (partial) {
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n >> 2 : 0;      // 16-byte loads over the aligned bulk
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)OPT_BLOCKS * 256) {
      const f32x4 x = reinterpret_cast<const f32x4*>(g)[i] * gscale;
      s = fmaf(x[0], x[0], s); s = fmaf(x[1], x[1], s); s = fmaf(x[2], x[2], s); s = fmaf(x[3], x[3], s);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)OPT_BLOCKS * 256) {
      float x = g[i] * gscale;
      s = fmaf(x, x, s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    step[0] += 1.f;
    if (kl) {                         // ppo.py:126-136
      const float k = kl[0] * kl_scale, l = lr[0];
      if (k > desired_kl * 2.f) lr[0] = fmaxf(lr_min, l / 1.5f);
      else if (k < desired_kl / 2.f && k > 0.f) lr[0] = fminf(lr_max, l * 1.5f);
    }
  }
}

// V = 4: four consecutive elements per lane and pass (16-byte loads / stores of p, g, m, v, one 8-byte store of the bf16 copy) — the host picks it when
// both ranges start and end on multiples of 4 elements and the buffers are 16-byte aligned; V = 1 is the general path.
// (Measured on MI355X, 3.2 M elements: 34 us with V = 1 — 2.8 TB/s of the 96 MB it moves — see DESIGN.md section 7 for V = 4.)
template <int V>
__global__ __launch_bounds__(256) void adam_kernel(float* p, float* g, float* m, float* v, int64_t start0, int64_t count0,
                                                   int64_t start1, int64_t count1, float gscale, const float* partial, float max_norm,
                                                   const float* step, const float* lr, float beta1, float beta2, float eps,
                                                   bf16_t* body, int64_t n_body, float* tail, int64_t n_tail, int zero_grad, float* zero_slot,
                                                   Go1PpoAdamExtras ex) {
  // clip factor from the norm pass's per-block partials: all 256 lanes load their OPT_BLOCKS / 256 values at once (one memory round
  // trip; a 64-lane loop of 32 dependent loads cost ~10 us in front of every block's single pass)
  __shared__ float clip_red[4];
  float clip = 1.f;
  if (partial) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < OPT_BLOCKS / 256; u++) s += partial[threadIdx.x + 256 * u];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) clip_red[threadIdx.x >> 6] = s;
    __syncthreads();
    clip = fminf(1.f, max_norm / (sqrtf((clip_red[0] + clip_red[1]) + (clip_red[2] + clip_red[3])) + 1e-6f));
  }
  const float gs = gscale * clip;
  const float t = step[0], l = lr[0];
  const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
  const float step_size = l / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const int64_t total = count0 + count1;
  constexpr int64_t CH = 256 * V;                                          // elements per block and pass
  for (int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; j < total; j += (int64_t)gridDim.x * CH) {
    const int64_t i = j < count0 ? start0 + j : start1 + (j - count0);
    // (block-uniform) can any element of this chunk belong to a transposed block?  first / last element index of the chunk
    const int64_t j0 = j - (int64_t)threadIdx.x * V, j1 = j0 + CH - 1 < total ? j0 + CH - 1 : total - 1;
    const int64_t i0 = j0 < count0 ? start0 + j0 : start1 + (j0 - count0), i1 = j1 < count0 ? start0 + j1 : start1 + (j1 - count0);
    bool tr_chunk = false;
#pragma unroll
    for (int t = 0; t < GO1PPO_ADAM_MAX_TRANSPOSES; t++)
      if (t < ex.num_transposes) {
        const int64_t ts = ex.transpose[t].start, te = ts + (int64_t)ex.transpose[t].rows * ex.transpose[t].cols;
        tr_chunk = tr_chunk || (j0 < count0) != (j1 < count0) || (i0 < te && i1 >= ts);
      }
    float gi[V], mi[V], vi[V], pi[V];
    if constexpr (V == 4) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + i), m4 = *reinterpret_cast<const f32x4*>(m + i);
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + i), p4 = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
      for (int e = 0; e < 4; e++) { gi[e] = g4[e]; mi[e] = m4[e]; vi[e] = v4[e]; pi[e] = p4[e]; }
    } else {
      gi[0] = g[i]; mi[0] = m[i]; vi[0] = v[i]; pi[0] = p[i];
    }
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float ge = gi[e] * gs;
      mi[e] = beta1 * mi[e] + (1.f - beta1) * ge;
      vi[e] = beta2 * vi[e] + (1.f - beta2) * ge * ge;
      pi[e] = pi[e] - step_size * mi[e] / (sqrtf(vi[e]) * inv_sqrt_bc2 + eps);
    }
    if constexpr (V == 4) {
      if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};      // the next backward pass accumulates into a clean gradient: no separate fill pass
      *reinterpret_cast<f32x4*>(m + i) = f32x4{mi[0], mi[1], mi[2], mi[3]};
      *reinterpret_cast<f32x4*>(v + i) = f32x4{vi[0], vi[1], vi[2], vi[3]};
      *reinterpret_cast<f32x4*>(p + i) = f32x4{pi[0], pi[1], pi[2], pi[3]};
    } else {
      if (zero_grad) g[i] = 0.f;
      m[i] = mi[0]; v[i] = vi[0]; p[i] = pi[0];
    }
    bf16_t pb[V];
#pragma unroll
    for (int e = 0; e < V; e++) pb[e] = f2bf(pi[e]);
    if (V == 4 && i + 4 <= n_body) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4_t;
      *reinterpret_cast<us4_t*>(body + i) = us4_t{pb[0], pb[V > 1 ? 1 : 0], pb[V > 2 ? 2 : 0], pb[V > 3 ? 3 : 0]};
    } else {
#pragma unroll
      for (int e = 0; e < V; e++) {
        if (i + e < n_body) body[i + e] = pb[e];
        else if (tail && i + e - n_body < n_tail) tail[i + e - n_body] = pi[e];      // (slots behind the tail — KL, padding — have no compute copy)
      }
    }
    // K-contiguous (transposed) bf16 copies of the weights whose input gradient runs on go1ppo_gemm_nt: kept current here
    // instead of by a transpose-copy launch per backward pass
    if (tr_chunk) {
#pragma unroll
      for (int t = 0; t < GO1PPO_ADAM_MAX_TRANSPOSES; t++) {
        if (t < ex.num_transposes) {
          const uint32_t rows = (uint32_t)ex.transpose[t].rows, cols = (uint32_t)ex.transpose[t].cols;
#pragma unroll
          for (int e = 0; e < V; e++) {
            const uint64_t off = (uint64_t)(i + e - ex.transpose[t].start);
            if (i + e < n_body && off < (uint64_t)rows * cols) {
              const uint32_t r = (uint32_t)off / cols, c = (uint32_t)off % cols;
              reinterpret_cast<bf16_t*>(ex.transpose[t].dst)[(size_t)c * rows + r] = pb[e];
            }
          }
        }
      }
    }
  }
  if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) *zero_slot = 0.f;
}

// ---------------------------------------------------------------------------------------------- fused MLP tail (forward)
// The layers behind the shared first layer (actor / critic 512->256->128->64, adaptation 256->128->64) for a block
// of 32 rows per workgroup, activations resident in LDS from layer to layer, weights streamed from L2 as MFMA B
// fragments (W is [n][k] row-major: the 8 consecutive k a lane needs are 16 contiguous bytes — no staging, no
// transpose).  Replaces 3 hipBLASLt GEMMs + 2 ELU kernels per net whose 6..40 us each are launch/occupancy bound
// (M = 24576 or 4096 rows, N <= 256).  out = elu(in W^T + b) per layer (no ELU on the head); intermediate
// activations are also written to global memory when the backward pass needs them.
#define TF_MAXK 512
template <int BM>          // rows per workgroup: 32 (small batches: more workgroups) or 64 (large: half the weight traffic)
__global__ __launch_bounds__(256) void tail_fwd_kernel(Go1PpoTailArgs A) {
  constexpr int RT = BM / 16;                                            // 16-row MFMA tiles per workgroup
  __shared__ __attribute__((aligned(16))) bf16_t act[2][BM][TF_MAXK + 8];
  const Go1PpoTailNet& N = A.net[blockIdx.y];
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  if (row0 >= N.rows) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, c = lane & 15, g = lane >> 4;
  // the latent weights Wz (k_in rows of npv = 2 bf16: one dword each) once per workgroup into LDS — as global loads inside the
  // staging loop they cost 8 extra load instructions per 16-byte chunk of input and doubled the kernel's time
  __shared__ uint32_t wzl[TF_MAXK];
  const bool lat2 = N.elu_in && N.latent && N.npv == 2 && !(N.wz_ld & 1) && !(N.lat_ld & 1) && !(((uintptr_t)N.wz | (uintptr_t)N.latent) & 3);
  if (lat2) {
    for (int i = t; i < N.layer[0].k_in; i += 256) wzl[i] = *reinterpret_cast<const uint32_t*>((const bf16_t*)N.wz + (int64_t)i * N.wz_ld);
    __syncthreads();
  }
  // stage the input rows
  {
    const int k0 = N.layer[0].k_in, cgs = k0 >> 3;
    for (int i = t; i < BM * cgs; i += 256) {
      const int r = i / cgs, cg = i - r * cgs;
      const int64_t row = row0 + r;
      Bf8 v;
      if (row < N.rows) v = *reinterpret_cast<const Bf8*>((const bf16_t*)N.in + row * N.ld_in + 8 * cg);
      else {
#pragma unroll
        for (int e = 0; e < 8; e++) v.v[e] = 0;
      }
      if (N.elu_in && row < N.rows) {                              // the first layer's activation, on the way in
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = bf2f(v.v[e]);
        if (lat2) {                                                // both latent columns of a Wz row are one dword (in LDS)
          const uint32_t lp = *reinterpret_cast<const uint32_t*>((const bf16_t*)N.latent + row * N.lat_ld);
          const float l0 = __uint_as_float(lp << 16), l1 = __uint_as_float(lp & 0xffff0000u);
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const uint32_t wp = wzl[8 * cg + e];
            x[e] = fmaf(l1, __uint_as_float(wp & 0xffff0000u), fmaf(l0, __uint_as_float(wp << 16), x[e]));
          }
        } else if (N.latent) {
          const bf16_t* lat = (const bf16_t*)N.latent + row * N.lat_ld;
          for (int q = 0; q < N.npv; q++) {
            const float l = bf2f(lat[q]);
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = fmaf(l, bf2f(((const bf16_t*)N.wz)[(int64_t)(8 * cg + e) * N.wz_ld + q]), x[e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = elu1(x[e]);
        v = pack_bf8(x);
      }
      *reinterpret_cast<Bf8*>(&act[0][r][8 * cg]) = v;
    }
  }
  __syncthreads();
  int cur = 0;
  for (int li = 0; li < N.num_layers; li++) {
    const Go1PpoTailLayer L = N.layer[li];
    const bf16_t* W = (const bf16_t*)L.W;
    const bf16_t* bias = (const bf16_t*)L.bias;
    bf16_t* out = (bf16_t*)L.out;
    const int ksteps = L.k_in >> 5;
    const int ntiles = L.n_out >> 4;
    for (int j0 = wave; j0 < ntiles; j0 += 16) {                // this wave's column tiles j0, j0+4, j0+8, j0+12: one pass
      // register blocking: every A fragment read from LDS feeds up to 4 column tiles
      f32x4 acc[4][RT];
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int h = 0; h < RT; h++) acc[jj][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const bf16_t* wrow = W + (int64_t)(16 * j0 + c) * L.k_in + 8 * g;
      const int64_t wstep = (int64_t)64 * L.k_in;                 // 4 tiles = 64 rows of W further
      // The weight fragments come straight from L2 (16 B per lane, no staging): a load waited for where it is issued costs a
      // full L2 round trip (~700 cycles) per k-step with one wavefront per SIMD.  So they are fetched a CHUNK of 4 k-steps
      // ahead into a second register set: 16 loads in flight under the 32 MFMAs of the current chunk.
      constexpr int CH = 4;
      const int nch = (ksteps + CH - 1) / CH;
      bf16x8_t BA[CH][4], BB[CH][4];
      auto fetch = [&](bf16x8_t (&B)[CH][4], int kc) {
#pragma unroll
        for (int s_ = 0; s_ < CH; s_++) {
          const int ks = kc * CH + s_;
#pragma unroll
          for (int jj = 0; jj < 4; jj++)
            if (ks < ksteps && j0 + 4 * jj < ntiles) B[s_][jj] = *reinterpret_cast<const bf16x8_t*>(wrow + jj * wstep + 32 * ks);
        }
      };
      auto compute = [&](const bf16x8_t (&B)[CH][4], int kc) {
#pragma unroll
        for (int s_ = 0; s_ < CH; s_++) {
          const int ks = kc * CH + s_;
          if (ks < ksteps) {
            bf16x8_t a[RT];
#pragma unroll
            for (int h = 0; h < RT; h++) a[h] = *reinterpret_cast<const bf16x8_t*>(&act[cur][16 * h + c][32 * ks + 8 * g]);
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
              if (j0 + 4 * jj < ntiles) {
#pragma unroll
                for (int h = 0; h < RT; h++) acc[jj][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[h], B[s_][jj], acc[jj][h], 0, 0, 0);
              }
          }
        }
      };
      fetch(BA, 0);
#pragma unroll 1
      for (int kc = 0; kc < nch; kc += 2) {
        if (kc + 1 < nch) fetch(BB, kc + 1);
        compute(BA, kc);
        if (kc + 1 < nch) {
          if (kc + 2 < nch) fetch(BA, kc + 2);
          compute(BB, kc + 1);
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        if (j0 + 4 * jj < ntiles) {
          const int n = 16 * (j0 + 4 * jj) + c;
          const float bn = bf2f(bias[n]);
#pragma unroll
          for (int h = 0; h < RT; h++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              float v = acc[jj][h][q] + bn;
              if (L.elu) v = elu1(v);
              const bf16_t o = f2bf(v);
              const int r = 16 * h + 4 * g + q;
              act[cur ^ 1][r][n] = o;
              if (out && row0 + r < N.rows) out[(row0 + r) * L.ld_out + n] = o;
            }
          }
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// row split of one weight-gradient problem: enough workgroups to fill the chip, but a bounded fan-in per output
// element — the partial sums meet in fp32 atomics, which resolve beyond the per-XCD L2 and serialise per address
// (measured: 256x512 best at 32 splits, 128x256 at 64, 64x128 at 128)
static void wgrad_split(int64_t rows, int n, int k, int step, int64_t* S_out, int64_t* chunk_steps_out) {
  const int tiles = (n / 64) * (k / 64);
  const int64_t steps = (rows + step - 1) / step;
  int64_t S = 512 / tiles;
  if (S < 32) S = 32;
  if (S > 128) S = 128;
  if (S > steps) S = steps;
  const int64_t chunk_steps = (steps + S - 1) / S;
  *S_out = (steps + chunk_steps - 1) / chunk_steps;
  *chunk_steps_out = chunk_steps;
}

// ---------------------------------------------------------------------------------------------- C-ABI
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int go1ppo_elu_fwd(void* y, int64_t rows, int cols, int ld, const void* lat, int lat_ld, int npv, const void* wz,
                              int wz_ld, int lat_cols, void* stream) {
  if (!y || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || !aligned16(y)) return -1;
  if (lat && (!wz || npv <= 0 || (lat_cols & 7) || (npv == 2 && ((lat_ld | wz_ld) & 1 || ((uintptr_t)lat | (uintptr_t)wz) & 3)))) return -2;
  if (rows > INT32_MAX) return -1;
  dim3 grid((unsigned)(((cols >> 3) + 15) / 16), (unsigned)((rows + 16 * ELU_ROWS - 1) / (16 * ELU_ROWS)));
  if (lat)
    elu_fwd_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>((bf16_t*)y, (int)rows, cols, ld, (const bf16_t*)lat, lat_ld, npv,
                                                                      (const bf16_t*)wz, wz_ld, lat_cols);
  else
    elu_fwd_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>((bf16_t*)y, (int)rows, cols, ld, nullptr, 0, 0, nullptr, 0, 0);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_elu_bwd(const void* d, int ld_d, const void* h, int ld_h, int64_t rows, int cols, float* bias_grad, void* out,
                              int ld_out, void* stream) {
  if (!d || !out || rows <= 0 || cols <= 0 || (cols & 7) || (ld_d & 7) || (ld_out & 7) || (h && (ld_h & 7)) || !aligned16(d) ||
      !aligned16(out) || (h && !aligned16(h)))
    return -1;
  if (h && !bias_grad && rows <= INT32_MAX) {
    dim3 grid((unsigned)(((cols >> 3) + 15) / 16), (unsigned)((rows + 16 * ELU_ROWS - 1) / (16 * ELU_ROWS)));
    elu_bwd_tile_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)d, ld_d, (const bf16_t*)h, ld_h, (int)rows, cols,
                                                                     (bf16_t*)out, ld_out);
    return hipGetLastError() == hipSuccess ? 0 : -9;
  }
  int cgs = cols >> 3;
  int cgb = cgs >= 32 ? 32 : (cgs >= 16 ? 16 : 8);
  int xblocks = (cgs + cgb - 1) / cgb;
  int rchunk = 256;                    // rows per block: aim for >= ~1500 blocks, at least 4 rows per thread-row
  while (rchunk > 32 && xblocks * ((rows + rchunk - 1) / rchunk) < 1536) rchunk >>= 1;
  dim3 grid(xblocks, (unsigned)((rows + rchunk - 1) / rchunk));
  elu_bwd_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)d, ld_d, (const bf16_t*)h, ld_h, rows, cols, bias_grad,
                                                              (bf16_t*)out, ld_out, cgb, rchunk);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_loss(const Go1PpoLossArgs* a, void* stream) {
  if (!a || a->rows <= 0 || a->num_actions <= 0 || a->num_actions > GO1PPO_MAX_ACTIONS || a->head_ld < a->num_actions) return -1;
  // (measured: 64-thread workgroups — 384 instead of 96 — are slower, 33 vs 26 us: four times the atomics on the same 28 words)
  // (one wavefront per workgroup — 384 workgroups instead of 96 — measured 32 us against 22: four times the same-address atomics of the reductions)
  // (measured, round 6 — profiles/r06_loss_kernel_ablation.txt: the same-address atomics are 1 us of the kernel's 16-22; meeting the sums through
  //  per-workgroup rows and a last-workgroup ticket instead took 45 us: the device-scope fences in front of the ticket write the XCD's L2 back)
  loss_kernel<<<dim3((unsigned)((a->rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(*a);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}
extern "C" int go1ppo_mse(const void* pred, int pred_ld, const float* target, int npv, const int64_t* idx, int64_t rows,
                          int64_t num_train, int selective, void* d_pred, float* d_pred_bias, float* train_loss,
                          float* test_loss, void* stream) {
  if (!pred || !target || !idx || rows <= 0 || num_train <= 0 || num_train >= rows || npv <= 0 || npv > pred_ld) return -1;
  mse_kernel<<<dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      (const bf16_t*)pred, pred_ld, target, npv, idx, rows, num_train, selective, (bf16_t*)d_pred, d_pred_bias, train_loss, test_loss);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_wgrad(const void* dz, int ld_dz, const void* h, int ld_h, int64_t rows, int n, int k, float* dW, int ldw,
                            float* bias_grad, void* stream) {
  if (!dz || !h || !dW || rows <= 0 || n <= 0 || k <= 0 || (n & 63) || (k & 63) || (ld_dz & 7) || (ld_h & 7) || !aligned16(dz) || !aligned16(h))
    return -1;
  const int step = 64;
  int64_t S, chunk_steps;
  wgrad_split(rows, n, k, step, &S, &chunk_steps);
  dim3 grid(n / 64, k / 64, (unsigned)S);
  wgrad2_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)dz, ld_dz, (const bf16_t*)h, ld_h, rows,
                                                             (int)(chunk_steps * step), dW, ldw, bias_grad);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

#include "go1ppo_gemm.h"
#include "go1ppo_mlp.h"

extern "C" int go1ppo_act(const void* mean, const void* value, int head_ld, const float* std, int num_actions, int64_t rows,
                          const float* noise, float* actions, float* mu, float* sigma, float* values, float* logp, void* stream) {
  if (!mean || !value || !std || !noise || !actions || !mu || !sigma || !values || !logp || rows <= 0 || num_actions <= 0 ||
      num_actions > head_ld)
    return -1;
  act_kernel<<<dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream>>>(          // (4096 rows: 64 CUs instead of 16)
      
      (const bf16_t*)mean, (const bf16_t*)value, head_ld, std, num_actions, rows, noise, actions, mu, sigma, values, logp);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_store_step(const float* rewards, const uint8_t* dones, const uint8_t* time_outs, const int32_t* env_bins,
                                 const float* values, float gamma, int64_t n, float* rewards_out, uint8_t* dones_out,
                                 float* env_bins_out, void* stream) {
  if (!rewards || !dones || !values || !rewards_out || !dones_out || n <= 0 || (env_bins && !env_bins_out)) return -1;
  store_step_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      rewards, dones, time_outs, env_bins, values, gamma, n, rewards_out, dones_out, env_bins_out);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_gae(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, int T, int64_t N,
                          float gamma, float lam, float* returns, float* advantages, double* stats, void* stream) {
  if (!rewards || !dones || !values || !last_values || !returns || !advantages || !stats || T <= 0 || N <= 0) return -1;
  gae_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(rewards, dones, values, last_values, T, N, gamma, lam,
                                                                                      returns, advantages, stats);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

// ---------------------------------------------------------------------------------------------- observation ring
// The rollout storage of the observation histories (reference rollout_storage.py:36-38: a (T, N, H * num_obs) block) holds
// every observation H times: consecutive windows overlap in H - 1 entries.  The ring keeps each observation ONCE —
// ring[t][n][num_obs] in bf16, t = 0 .. T + H - 2, the window of rollout step s being rows s .. s + H - 1 — and the
// augmented GEMM rows [window | 1 | privileged | 0 pad] are assembled where they are consumed: for the inference of the
// current step (ring_step) and for the rows of a mini-batch (ring_gather).  One thread per output dword (two bf16).
__device__ __forceinline__ uint32_t ring_pack(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// one augmented row by ONE WAVEFRONT: the window's H entries are `no`/2 dwords each (140 B for the Go1: contiguous, 4-byte
// aligned), entry j read from win + j * stride; lanes walk the entry's dwords, several entries in flight per lane
__device__ __forceinline__ void ring_row(const uint32_t* __restrict__ win, int64_t stride_dw, const float* __restrict__ newest,
                                         const float* __restrict__ priv, int H, int no, int npv, int Kp, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, nod = no >> 1, Kd = H * nod;
  const int Hm = newest ? H - 1 : H;
  for (int c = lane; c < nod; c += 64) {
    int j = 0;
    for (; j + 6 <= Hm; j += 6) {
      uint32_t v[6];
#pragma unroll
      for (int u = 0; u < 6; u++) v[u] = win[(int64_t)(j + u) * stride_dw + c];
#pragma unroll
      for (int u = 0; u < 6; u++) out[(j + u) * nod + c] = v[u];
    }
    for (; j < Hm; j++) out[j * nod + c] = win[(int64_t)j * stride_dw + c];
    if (newest) out[(H - 1) * nod + c] = ring_pack(newest[2 * c], newest[2 * c + 1]);
  }
  for (int d = Kd + lane; d < (Kp >> 1); d += 64) {
    float e[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int col = 2 * d + q - 2 * Kd;               // 0: the bias column, 1 .. npv: privileged observations, then padding
      e[q] = col == 0 ? 1.f : (col <= npv ? priv[col - 1] : 0.f);
    }
    out[d] = ring_pack(e[0], e[1]);
  }
}

// rows 0 .. H-1 of the ring from the environment's fp32 history window (first step of a rollout)
__global__ __launch_bounds__(256) void ring_snapshot_kernel(const float* __restrict__ hist, int64_t ld_hist, int64_t N, int H, int no,
                                                            uint32_t* __restrict__ ring) {
  const int nod = no >> 1;
  const int64_t total = N * H * nod, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % nod);
  const int64_t r = i / nod, n = r % N, j = r / N;
  const float* src = hist + n * ld_hist + j * no + 2 * c;
  ring[(j * N + n) * nod + c] = ring_pack(src[0], src[1]);
}

// one WAVEFRONT per environment (4 per workgroup): append the new observation (row s + H - 1; dst == NULL: the snapshot
// already holds it), assemble the inference row, keep the fp32 copies of obs / privileged obs the storage wants
__global__ __launch_bounds__(256) void ring_step_kernel(const float* __restrict__ obs, const float* __restrict__ priv, uint32_t* __restrict__ dst,
                                                        const uint32_t* __restrict__ window, int64_t N, int H, int no, int npv, int Kp,
                                                        uint32_t* __restrict__ X, float* __restrict__ obs_store, float* __restrict__ priv_store) {
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63, nod = no >> 1;
  const float* o = obs + n * no;
  const float* pv = priv + n * npv;
  for (int c = lane; c < nod; c += 64) {
    if (dst) dst[n * nod + c] = ring_pack(o[2 * c], o[2 * c + 1]);
    if (obs_store) { obs_store[n * no + 2 * c] = o[2 * c]; obs_store[n * no + 2 * c + 1] = o[2 * c + 1]; }
  }
  if (priv_store)
    for (int c = lane; c < npv; c += 64) priv_store[n * npv + c] = pv[c];
  ring_row(window + n * nod, N * nod, o, pv, H, no, npv, Kp, X + n * (Kp >> 1));
}

// one wavefront per mini-batch row: storage index f = s * N + n
__global__ __launch_bounds__(256) void ring_gather_kernel(const uint32_t* __restrict__ ring, const float* __restrict__ priv_store,
                                                          const int64_t* __restrict__ idx, int64_t rows, int64_t N, int H, int no, int npv,
                                                          int Kp, uint32_t* __restrict__ X) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const int64_t f = idx[i], s = f / N, n = f - s * N;
  const int nod = no >> 1;
  ring_row(ring + (s * N + n) * nod, N * nod, nullptr, priv_store + f * npv, H, no, npv, Kp, X + i * (Kp >> 1));
}

extern "C" int go1ppo_ring_snapshot(const float* hist, int64_t ld_hist, int64_t N, int H, int no, void* ring, void* stream) {
  if (!hist || !ring || N <= 0 || H <= 0 || no <= 0 || (no & 1) || ld_hist < (int64_t)H * no) return -1;
  const int64_t total = N * H * (no >> 1);
  ring_snapshot_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(hist, ld_hist, N, H, no, (uint32_t*)ring);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_ring_step(const float* obs, const float* priv, void* ring_dst, const void* ring_window, int64_t N, int H, int no,
                                int npv, int Kp, void* X, float* obs_store, float* priv_store, void* stream) {
  if (!obs || !priv || !ring_window || !X || N <= 0 || H <= 0 || no <= 0 || (no & 1) || npv < 0 || npv > 256 || (Kp & 1) ||
      Kp < H * no + 1 + npv)
    return -1;
  ring_step_kernel<<<dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(obs, priv, (uint32_t*)ring_dst, (const uint32_t*)ring_window, N, H,
                                                                            no, npv, Kp, (uint32_t*)X, obs_store, priv_store);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_ring_gather(const void* ring, const float* priv_store, const int64_t* idx, int64_t rows, int64_t N, int H, int no,
                                  int npv, int Kp, void* X, void* stream) {
  if (!ring || !priv_store || !idx || !X || rows <= 0 || N <= 0 || H <= 0 || no <= 0 || (no & 1) || npv < 0 || (Kp & 1) ||
      Kp < H * no + 1 + npv)
    return -1;
  ring_gather_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>((const uint32_t*)ring, priv_store, idx, rows, N,
                                                                                           H, no, npv, Kp, (uint32_t*)X);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_normalize(float* adv, int64_t n, const double* stats, void* stream) {
  if (!adv || !stats || n <= 0) return -1;
  normalize_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(adv, n, stats);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_opt_prestep(const float* g, int64_t n, float gscale, float* partial, float* step, float* lr, const float* kl,
                                  float kl_scale, float desired_kl, float lr_min, float lr_max, void* stream) {
  if (!step || (partial && (!g || n <= 0)) || (kl && !lr)) return -1;
  prestep_kernel<<<dim3(partial ? OPT_BLOCKS : 1), dim3(256), 0, (hipStream_t)stream>>>(g, n, gscale, partial, step, lr, kl, kl_scale,
                                                                                       desired_kl, lr_min, lr_max, nullptr, 0);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_opt_prestep_pieces(float* g, int64_t n, const Go1PpoGradPiece* device_pieces, int num_pieces, float gscale, float* partial,
                                         float* step, float* lr, const float* kl, float kl_scale, float desired_kl, float lr_min, float lr_max,
                                         void* stream) {
  if (!step || !partial || !g || n <= 0 || (kl && !lr) || !device_pieces || num_pieces <= 0 || !aligned16(g)) return -1;
  prestep_kernel<<<dim3(OPT_BLOCKS), dim3(256), 0, (hipStream_t)stream>>>(g, n, gscale, partial, step, lr, kl, kl_scale, desired_kl, lr_min, lr_max,
                                                                         device_pieces, num_pieces);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_grad_reduce(float* g, const Go1PpoGradPiece* device_pieces, int num_pieces, void* stream) {
  if (!g || !device_pieces || num_pieces <= 0 || !aligned16(g)) return -1;
  grad_reduce_kernel<<<dim3(OPT_BLOCKS), dim3(256), 0, (hipStream_t)stream>>>(g, device_pieces, num_pieces);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_opt_adam(float* p, float* g, float* m, float* v, int64_t start0, int64_t count0, int64_t start1,
                               int64_t count1, float gscale, const float* partial, float max_norm, const float* step, const float* lr,
                               float beta1, float beta2, float eps, void* body, int64_t n_body, float* tail, int64_t n_tail, int zero_grad,
                               float* zero_slot, const Go1PpoAdamExtras* extras, void* stream) {
  if (!p || !g || !m || !v || !step || !lr || !body || count0 < 0 || count1 < 0 || n_tail < 0) return -1;
  Go1PpoAdamExtras ex;
  memset(&ex, 0, sizeof(ex));
  if (extras) {
    ex = *extras;
    if (ex.num_transposes < 0 || ex.num_transposes > GO1PPO_ADAM_MAX_TRANSPOSES) return -2;
    for (int t = 0; t < ex.num_transposes; t++)
      if (!ex.transpose[t].dst || ex.transpose[t].rows <= 0 || ex.transpose[t].cols <= 0) return -2;
  }
  if (count0 + count1 == 0) {           // an empty slice (a rank of a sharded step that owns padding only): just the slot
    if (zero_slot && hipMemsetAsync(zero_slot, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return -9;
    return 0;
  }
  int64_t total = count0 + count1;
  const bool vec = !((start0 | count0 | start1 | count1) & 3) && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && aligned16(body);
  if (vec) {
    int64_t blocks = (total / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    adam_kernel<4><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, start0, count0, start1, count1, gscale, partial, max_norm, step, lr,
                                                                                  beta1, beta2, eps, (bf16_t*)body, n_body, tail, n_tail, zero_grad, zero_slot, ex);
  } else {
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    adam_kernel<1><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, start0, count0, start1, count1, gscale, partial, max_norm, step, lr,
                                                                                  beta1, beta2, eps, (bf16_t*)body, n_body, tail, n_tail, zero_grad, zero_slot, ex);
  }
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

// ---------------------------------------------------------------------------------------------- split-K partial sums
// out[r][c] (fp32) = sum_b part[b][r][c] (bf16): the first-layer weight gradient leaves hipBLASLt as `count` row-chunk partial
// products (a manual split-K, fused.py `_big_wgrad`); this pass sums them into the flat fp32 gradient — replacing the library
// reduction — and writes exact zeros on the columns [zero_c0, zero_c1) of the first zero_rows rows (the privileged-observation
// columns of the adaptation module's and the actor's rows, see adam_kernel) instead of a separate fill launch.
__global__ __launch_bounds__(256) void sum_partials_kernel(const bf16_t* __restrict__ part, int count, int64_t stride, int64_t total8, int cols,
                                                           float* __restrict__ out, int zero_rows, int zero_c0, int zero_c1) {
  for (int64_t i8 = (int64_t)blockIdx.x * 256 + threadIdx.x; i8 < total8; i8 += (int64_t)gridDim.x * 256) {
    const int64_t i = i8 << 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < count; b++) {
      const Bf8 v = *reinterpret_cast<const Bf8*>(part + (int64_t)b * stride + i);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v.v[e]);
    }
    const int64_t row = i / cols;
    const int c = (int)(i - row * cols);
    if (row < zero_rows && c < zero_c1 && c + 8 > zero_c0) {
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (c + e >= zero_c0 && c + e < zero_c1) acc[e] = 0.f;
    }
    reinterpret_cast<f32x4*>(out + i)[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
    reinterpret_cast<f32x4*>(out + i)[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}

extern "C" int go1ppo_sum_partials(const void* partials, int count, int64_t stride, int64_t rows, int cols, float* out, int zero_rows,
                                   int zero_c0, int zero_c1, void* stream) {
  if (!partials || !out || count <= 0 || rows <= 0 || cols <= 0 || (cols & 7) || (stride & 7) || stride < rows * cols || !aligned16(partials) ||
      !aligned16(out) || zero_rows < 0 || zero_c0 < 0 || zero_c1 < zero_c0 || zero_c1 > cols)
    return -1;
  const int64_t total8 = rows * cols / 8;
  int64_t blocks = (total8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  sum_partials_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)partials, count, stride, total8, cols, out,
                                                                                   zero_rows, zero_c0, zero_c1);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_opt_partials(void) { return OPT_BLOCKS; }

extern "C" int go1ppo_tail_fwd(const Go1PpoTailArgs* args, void* stream) {
  if (!args || args->num_nets <= 0 || args->num_nets > GO1PPO_TAIL_MAX_NETS) return -1;
  int64_t max_rows = 0;
  for (int i = 0; i < args->num_nets; i++) {
    const Go1PpoTailNet& N = args->net[i];
    if (!N.in || N.rows <= 0 || N.num_layers <= 0 || N.num_layers > GO1PPO_TAIL_MAX_LAYERS || (N.ld_in & 7) || !aligned16(N.in)) return -1;
    for (int l = 0; l < N.num_layers; l++) {
      const Go1PpoTailLayer& L = N.layer[l];
      if (!L.W || !L.bias || L.k_in <= 0 || L.k_in > TF_MAXK || (L.k_in & 31) || L.n_out <= 0 || L.n_out > TF_MAXK || (L.n_out & 15) ||
          !aligned16(L.W) || (l > 0 && L.k_in != N.layer[l - 1].n_out) || (L.out && L.ld_out < L.n_out))
        return -2;
    }
    if (N.rows > max_rows) max_rows = N.rows;
  }
  static const bool bm64 = getenv("GO1PPO_TAIL_BM64") != nullptr;      // probe switch (tools/probes/tail_first_layer.py)
  if (bm64)
    tail_fwd_kernel<64><<<dim3((unsigned)((max_rows + 63) / 64), args->num_nets), dim3(256), 0, (hipStream_t)stream>>>(*args);
  else
    tail_fwd_kernel<32><<<dim3((unsigned)((max_rows + 31) / 32), args->num_nets), dim3(256), 0, (hipStream_t)stream>>>(*args);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_wgrad_plan(Go1PpoWgradProblem* probs, int count) {
  if (!probs || count <= 0) return -1;
  int total = 0;
  for (int i = 0; i < count; i++) {
    Go1PpoWgradProblem& P = probs[i];
    if (!P.dz || !P.h || !P.dW || P.partials || P.rows <= 0 || P.n <= 0 || P.k <= 0 || (P.n & 63) || (P.k & 63) || (P.ld_dz & 7) || (P.ld_h & 7) ||
        !aligned16(P.dz) || !aligned16(P.h))
      return -1;
    int64_t S, chunk_steps;
    wgrad_split(P.rows, P.n, P.k, 64, &S, &chunk_steps);
    P.chunk_rows = (int32_t)(chunk_steps * 64);
    P.wg_offset = total;
    total += (int)S * (P.n / 64) * (P.k / 64);
  }
  return total;
}

extern "C" int go1ppo_wgrad_batched(const Go1PpoWgradProblem* device_probs, int count, int total_workgroups, void* stream) {
  if (!device_probs || count <= 0 || total_workgroups <= 0) return -1;
  wgrad_batched_kernel<<<dim3((unsigned)total_workgroups), dim3(256), 0, (hipStream_t)stream>>>(device_probs, count);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

#ifndef GO1_SOURCE_HASH
#define GO1_SOURCE_HASH "unstamped"      // __graft_entry__.build_ppo_hip passes the sha256 of the sources + flags
#endif
extern "C" const char* go1ppo_version(void) { return "go1ppo 0.3 (gfx950, abi 3) go1-src:" GO1_SOURCE_HASH; }
