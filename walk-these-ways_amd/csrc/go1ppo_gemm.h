// go1ppo_gemm.h — C = epilogue(A · Bᵀ) for the MLP layers of the PPO update, bf16 in / fp32 accumulate / bf16 out.
//
// A (M x K) and B (N x K) are both K-contiguous ("NT": activations times an nn.Linear weight), so both tiles stage
// into LDS the same way.  One workgroup = 4 wavefronts = one 128 x 128 output tile, K in steps of 64:
//   * staging: global_load_lds (16 B per lane, LDS-DMA), two LDS buffers, the loads of step t+1 in flight while step t
//     is on the MFMAs; the LDS image is [row][64] bf16 (128-B rows) with the 16-B chunk index XOR-ed by (row >> 1) & 7
//     — applied on the per-lane SOURCE address, the DMA destination is lane-linear — which makes every ds_read_b128
//     lane group of a fragment read hit 16 distinct bank quads (MI355X_MICROARCH.md, LDS);
//   * each wave owns a 64 x 64 sub-tile: 4 x 4 accumulators of v_mfma_f32_16x16x32_bf16, with the WEIGHT rows as the
//     A operand so that a lane ends up with 4 consecutive output columns of one output row (8-B bf16 stores);
//   * epilogue in registers: + bias[n], ELU on a column range, or the ELU-backward factor elu'(H) of another matrix
//     (dgrad of a hidden layer), so the activation passes never exist as separate kernels;
//   * workgroup -> tile map is XCD-aware: each of the 8 XCDs walks a contiguous range of tiles with the column tile
//     fastest, so the A row block of a tile row is re-read from that XCD's L2.
// M and N need not be multiples of 128 (rows are clamped on load and masked on store); K % 64 == 0.
#pragma once

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

__device__ __forceinline__ void glds16(const bf16_t* g, bf16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((global_cvoid_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

template <int EPI>   // 0: bias only, 1: bias + ELU on [elu_c0, elu_c1), 2: times elu'(H)
__device__ __forceinline__ void gemm_nt_body(const Go1PpoGemmArgs& a, bf16_t (*lds)[2][GEMM_BM * GEMM_BK]) {   // lds: [buffer][A|B][row][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- XCD-aware tile index (bijective for any grid size)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tiles_n = (a.N + GEMM_BN - 1) / GEMM_BN;
  const int m0 = (logical / tiles_n) * GEMM_BM, n0 = (logical % tiles_n) * GEMM_BN;

  // ---- staging addresses: wave w stages row blocks (8 rows each) w*4 .. w*4+3 of both tiles
  const int srow = lane >> 3;                                  // row within the 8-row block
  const bf16_t* ga[4];
  const bf16_t* gb[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    int R = (wave * 4 + p) * 8 + srow;
    int chunk = (lane & 7) ^ ((R >> 1) & 7);
    int ra = m0 + R < a.M ? m0 + R : a.M - 1;
    int rb = n0 + R < a.N ? n0 + R : a.N - 1;
    ga[p] = (const bf16_t*)a.A + (int64_t)ra * a.lda + chunk * 8;
    gb[p] = (const bf16_t*)a.B + (int64_t)rb * a.ldb + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      glds16(ga[p] + k0, &lds[buf][0][(wave * 4 + p) * 8 * GEMM_BK]);
      glds16(gb[p] + k0, &lds[buf][1][(wave * 4 + p) * 8 * GEMM_BK]);
    }
  };

  // ---- fragment read offsets (elements): row (lane & 15) of a 16-row fragment, k chunk kk*4 + (lane >> 4), swizzled
  const int fr = lane & 15, fg = lane >> 4;
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) foff[kk] = fr * GEMM_BK + (((kk * 4 + fg) ^ (fr >> 1)) << 3);
  const int wm = wave >> 1, wn = wave & 1;                      // the wave's 64 x 64 sub-tile

  f32x4 acc[4][4];                                              // [n fragment][m fragment]
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue 2 multiplies by elu'(H): its 16 H fragments are requested NOW — a load issued in the epilogue is waited for right
  // there (an HBM round trip per fragment with nothing to hide it: 19 such waits made this variant slower than GEMM + a
  // separate element-wise pass)
  // wide epilogue (N, ldc [, ldh] multiples of 8 and 16-byte aligned C [, H]): the finished tile goes through LDS (fp32, the 64 KB the
  // staging buffers occupied) and leaves as whole 256-byte row pieces — 16 lanes x 16 B — instead of fragment-shaped 8-byte stores
  // (16 rows x 32 B per instruction: the stores, and epilogue 2's H reads, were what the 512 <-> 256 layers' time went with);
  // epilogue 2's H row pieces are requested here in that same coalesced shape
  const bool wide = !(a.N & 7) && !(a.ldc & 7) && !((uintptr_t)a.C & 15) && (EPI != 2 || (!(a.ldh & 7) && !((uintptr_t)a.H & 15)));
  uint4 hrow[8];
  if (EPI == 2 && wide) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      int m = m0 + (int)(threadIdx.x >> 4) + 16 * it, n = n0 + (int)(threadIdx.x & 15) * 8;
      m = m < a.M ? m : a.M - 1;
      n = n < a.N ? n : a.N - 8;
      hrow[it] = *reinterpret_cast<const uint4*>((const bf16_t*)a.H + (int64_t)m * a.ldh + n);
    }
  }
  uint2 hpre[4][4];
  if (EPI == 2 && !wide) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int n = n0 + wn * 64 + j * 16 + fg * 4;
      n = n < a.N ? n : a.N - 4;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int m = m0 + wm * 64 + i * 16 + fr;
        m = m < a.M ? m : a.M - 1;
        hpre[j][i] = *reinterpret_cast<const uint2*>((const bf16_t*)a.H + (int64_t)m * a.ldh + n);
      }
    }
  }
  const int KT = a.K / GEMM_BK;
  stage(0, 0);
  for (int kt = 0; kt < KT; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // step kt landed; everyone is done with buffer buf^1
    if (kt + 1 < KT) stage(buf ^ 1, (kt + 1) * GEMM_BK);
    const bf16_t* la = &lds[buf][0][wm * 64 * GEMM_BK];
    const bf16_t* lb = &lds[buf][1][wn * 64 * GEMM_BK];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8_t xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) xa[i] = *reinterpret_cast<const bf16x8_t*>(la + i * 16 * GEMM_BK + foff[kk]);
#pragma unroll
      for (int j = 0; j < 4; j++) wb[j] = *reinterpret_cast<const bf16x8_t*>(lb + j * 16 * GEMM_BK + foff[kk]);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[j][i], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds D[n = 4*fg + e][m = fr] of each 16 x 16 fragment
  if (wide) {
    __syncthreads();                                            // every wave is done with the staging buffers
    float* img = reinterpret_cast<float*>(&lds[0][0][0]);       // [128 rows][128 columns] fp32, 16-byte chunk index XOR-ed with (row & 15)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int nl = wn * 64 + j * 16 + fg * 4, n = n0 + nl;
      f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias && n < a.N) {
        if (a.bias_bf16) {
          const uint2 braw = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.bias) + n);
          bias = f32x4{__uint_as_float(braw.x << 16), __uint_as_float(braw.x & 0xffff0000u), __uint_as_float(braw.y << 16),
                       __uint_as_float(braw.y & 0xffff0000u)};
        } else bias = *reinterpret_cast<const f32x4*>(a.bias + n);
      }
      const bool act = EPI == 1 && n >= a.elu_c0 && n < a.elu_c1 && !(n >= a.elu_skip_c0 && n < a.elu_skip_c1);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int row = wm * 64 + i * 16 + fr;
        f32x4 v = acc[j][i] + bias;
        if (act) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = elu1(v[e]);
        }
        *reinterpret_cast<f32x4*>(img + row * 128 + ((((nl >> 2)) ^ (row & 15)) << 2)) = v;
      }
    }
    __syncthreads();
    const int c8 = (int)(threadIdx.x & 15);
    const int n = n0 + c8 * 8;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = (int)(threadIdx.x >> 4) + 16 * it, m = m0 + row;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(img + row * 128 + (((2 * c8) ^ (row & 15)) << 2));
      const f32x4 hi = *reinterpret_cast<const f32x4*>(img + row * 128 + (((2 * c8 + 1) ^ (row & 15)) << 2));
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (EPI == 2) {
        const uint32_t hw[4] = {hrow[it].x, hrow[it].y, hrow[it].z, hrow[it].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float h0 = __uint_as_float(hw[e] << 16), h1 = __uint_as_float(hw[e] & 0xffff0000u);
          v[2 * e] *= h0 > 0.f ? 1.f : h0 + 1.f;
          v[2 * e + 1] *= h1 > 0.f ? 1.f : h1 + 1.f;
        }
      }
      if (m < a.M && n < a.N) {
        uint4 o;
        uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          f32x2 p2 = {v[2 * e], v[2 * e + 1]};
          ow[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(p2, bf16x2_t));
        }
        *reinterpret_cast<uint4*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int n = n0 + wn * 64 + j * 16 + fg * 4;
    if (n >= a.N) continue;                                     // N % 4 == 0: a 4-column group is in or out as a whole
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
      if (a.bias_bf16) {
        const uint2 braw = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.bias) + n);
        bias = f32x4{__uint_as_float(braw.x << 16), __uint_as_float(braw.x & 0xffff0000u), __uint_as_float(braw.y << 16),
                     __uint_as_float(braw.y & 0xffff0000u)};
      } else bias = *reinterpret_cast<const f32x4*>(a.bias + n);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= a.M) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = acc[j][i][e] + bias[e];
      if (EPI == 1) {
        if (n >= a.elu_c0 && n < a.elu_c1 && !(n >= a.elu_skip_c0 && n < a.elu_skip_c1)) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = elu1(v[e]);
        }
      } else if (EPI == 2) {
        const uint2 hraw = hpre[j][i];
        float h[4] = {__uint_as_float(hraw.x << 16), __uint_as_float(hraw.x & 0xffff0000u), __uint_as_float(hraw.y << 16),
                      __uint_as_float(hraw.y & 0xffff0000u)};
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= h[e] > 0.f ? 1.f : h[e] + 1.f;
      }
      f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
      uint2 o;
      o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
      o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
      *reinterpret_cast<uint2*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = o;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(Go1PpoGemmArgs a) {
  __shared__ __attribute__((aligned(1024))) bf16_t lds[2][2][GEMM_BM * GEMM_BK];
  gemm_nt_body<EPI>(a, lds);
}
// two independent problems with the same tile grid in ONE launch (blockIdx.y picks the problem): the actor's and the critic's
// 512 -> 256 layers.  As two launches on two streams they cost a graph fork and a join (5-10 us of idle device each, tools/timeline.py)
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_pair_kernel(Go1PpoGemmArgs a, Go1PpoGemmArgs b) {
  __shared__ __attribute__((aligned(1024))) bf16_t lds[2][2][GEMM_BM * GEMM_BK];
  gemm_nt_body<EPI>(blockIdx.y == 0 ? a : b, lds);
}

static int gemm_nt_check(const Go1PpoGemmArgs* a) {
  if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return -1;
  if ((a->K % GEMM_BK) || (a->N & 3) || (a->lda & 7) || (a->ldb & 7) || (a->ldc & 3) || !aligned16(a->A) || !aligned16(a->B) ||
      ((uintptr_t)a->C & 7))
    return -2;
  if (a->epilogue == 2 && (!a->H || (a->ldh & 3) || ((uintptr_t)a->H & 7))) return -3;
  if (a->bias && ((uintptr_t)a->bias & (a->bias_bf16 ? 7 : 15))) return -4;
  return 0;
}

extern "C" int go1ppo_gemm_nt_pair(const Go1PpoGemmArgs* a, const Go1PpoGemmArgs* b, void* stream) {
  int rc = gemm_nt_check(a);
  if (rc == 0) rc = gemm_nt_check(b);
  if (rc) return rc;
  const int64_t tiles = (int64_t)((a->M + GEMM_BM - 1) / GEMM_BM) * ((a->N + GEMM_BN - 1) / GEMM_BN);
  const int64_t tiles_b = (int64_t)((b->M + GEMM_BM - 1) / GEMM_BM) * ((b->N + GEMM_BN - 1) / GEMM_BN);
  if (tiles > INT32_MAX) return -5;
  if (tiles != tiles_b || a->epilogue != b->epilogue) return -7;      // same tile grid, same epilogue
  dim3 grid((unsigned)tiles, 2), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (a->epilogue) {
    case 0: gemm_nt_pair_kernel<0><<<grid, block, 0, s>>>(*a, *b); break;
    case 1: gemm_nt_pair_kernel<1><<<grid, block, 0, s>>>(*a, *b); break;
    case 2: gemm_nt_pair_kernel<2><<<grid, block, 0, s>>>(*a, *b); break;
    default: return -6;
  }
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_gemm_nt(const Go1PpoGemmArgs* a, void* stream) {
  const int rc = gemm_nt_check(a);
  if (rc) return rc;
  int64_t tiles = (int64_t)((a->M + GEMM_BM - 1) / GEMM_BM) * ((a->N + GEMM_BN - 1) / GEMM_BN);
  if (tiles > INT32_MAX) return -5;
  dim3 grid((unsigned)tiles), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (a->epilogue) {
    case 0: gemm_nt_kernel<0><<<grid, block, 0, s>>>(*a); break;
    case 1: gemm_nt_kernel<1><<<grid, block, 0, s>>>(*a); break;
    case 2: gemm_nt_kernel<2><<<grid, block, 0, s>>>(*a); break;
    default: return -6;
  }
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

// ================================================================================================ weight gradients
// dW[n][k] += sum_m dZ[m][n] H[m][k] ("TN": both operands are m-major, the reduction runs over the slow index).
// One workgroup = 8 wavefronts = one 128 x 128 output tile over a chunk of rows, 64 reduction rows per step:
//   * staging by LDS-DMA into FOUR buffers, three steps ahead of the MFMAs: a step's operands are 32 KB and the loaded
//     latency under load is ~3000 cycles, so one step in flight (the two-buffer scheme of gemm_nt_kernel) leaves the
//     matrix cores waiting two thirds of the time.  Waits are counted (s_waitcnt vmcnt(8 / 4 / 0): each wave has 4 DMA
//     instructions per step) and the barrier is a raw s_barrier, so that younger steps stay in flight across it;
//   * the LDS image is [m][128 columns] (256-B rows) read through ds_read_b64_tr_b16: a 16-lane group hands the
//     hardware a [4 m][16 columns] block and every lane receives the 4 m-values of ITS column, i.e. half an MFMA
//     operand (the reduction-slot order only has to agree between the two operands, and it does).  A 256-B row is
//     exactly one bank row, so the 8 rows a 32-lane group touches would all collide; the 16-B chunk index is XOR-ed
//     with 2 * ((m & 3) | ((m >> 3) & 1) << 2) on the DMA source address, which spreads them over 8 distinct 32-B
//     slots (tools/probes/tr_bank_probe.hip: same rate as a contiguous image);
//   * each wave owns 64 (n) x 32 (k): 4 x 2 accumulators of v_mfma_f32_16x16x32_bf16;
//   * the reduction is split over row chunks (the output has only (n/128)(k/128) tiles).  The partial tiles go to per-chunk fp32 SLABS
//     (Go1PpoWgradProblem.partials, plain stores; summed in a fixed order by the optimiser's norm pass or go1ppo_grad_reduce) — or, without
//     slabs, meet in fp32 atomics on the gradient buffer itself (~200 G atomic adds/s chip-wide: 15.6 of the 64.5 us of the PPO pass's
//     launch, which is why the product path uses slabs); the bias gradient (column sums of dZ) is one more MFMA per fragment against a
//     fragment of ones, in the workgroups of the first column tile (atomics in both modes: n values per workgroup).
// rows % 64 == 0, n % 8 == 0, k % 8 == 0.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
// The transpose reads are issued as inline asm: for loads the compiler can see, it drains the LDS-DMA queue (vmcnt(0))
// before the first read that follows a DMA into the same array, which would serialise the pipeline.  The price: the
// LDS counter is waited for by hand (lds_wait), with the fragment registers passed through the wait so that no consumer
// can be scheduled above it.
template <int OFF>
__device__ __forceinline__ s16x4_t lds_tr(uint32_t addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
struct TnFrags { s16x4_t lo[6], hi[6]; };                        // 4 dZ fragments + 2 H fragments, two 4-row halves each
template <int KK>
__device__ __forceinline__ void tn_read(TnFrags& f, const uint32_t (&aa)[4], const uint32_t (&ab)[2]) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f.lo[i] = lds_tr<KK * 32 * 256>(aa[i]);
    f.hi[i] = lds_tr<KK * 32 * 256 + 4 * 256>(aa[i]);
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    f.lo[4 + i] = lds_tr<KK * 32 * 256>(ab[i]);
    f.hi[4 + i] = lds_tr<KK * 32 * 256 + 4 * 256>(ab[i]);
  }
}
template <int N>
__device__ __forceinline__ void lds_wait(TnFrags& f) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2]), "+v"(f.lo[3]), "+v"(f.hi[3]),
                 "+v"(f.lo[4]), "+v"(f.hi[4]), "+v"(f.lo[5]), "+v"(f.hi[5])
               : "n"(N));
}
__device__ __forceinline__ bf16x8_t tn_operand(const TnFrags& f, int i) {
  s16x8_t v = __builtin_shufflevector(f.lo[i], f.hi[i], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

#define WTN_T 128            // output tile edge
#define WTN_STEP 64          // reduction rows per step
#define WTN_BUFS 4           // LDS buffers: WTN_BUFS - 1 steps in flight
#define WTN_THREADS 512

__device__ __forceinline__ void wgrad_tn_body(const bf16_t* P, int ldp, const bf16_t* Q, int ldq, int64_t m_begin, int steps,
                                              float* C, int ldc, float* bias_grad, int N, int K, int n0, int k0,
                                              bf16_t (*lds)[2][WTN_STEP * WTN_T], int zero_n, int zero_k0, int zero_k1, bool slab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- staging: wave w moves row blocks (4 rows of 256 B each) 2w, 2w+1 of both operand tiles
  const bf16_t* gp[2];
  const bf16_t* gq[2];
  bool vp[2], vq[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int row = (wave * 2 + p) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (2 * ((row & 3) | (((row >> 3) & 1) << 2)));
    const int cn = n0 + chunk * 8, ck = k0 + chunk * 8;
    // columns past the matrix (n = 64 or k = 64 problems on the 128-wide tile): the lane sits the DMA out — its LDS slot keeps whatever it
    // held, which only ever reaches output elements that are not stored — instead of moving bytes nobody uses through the L2 -> LDS path
    vp[p] = cn < N;
    vq[p] = ck < K;
    gp[p] = P + (m_begin + row) * ldp + (vp[p] ? cn : 0);
    gq[p] = Q + (m_begin + row) * ldq + (vq[p] ? ck : 0);
  }
  auto stage_piece = [&](int step, int p, int op) {              // one of the wave's 4 DMA instructions of a step
    const int buf = step & (WTN_BUFS - 1);
    if (op == 0) { if (vp[p]) glds16(gp[p] + (int64_t)step * WTN_STEP * ldp, &lds[buf][0][(wave * 2 + p) * 4 * WTN_T]); }
    else { if (vq[p]) glds16(gq[p] + (int64_t)step * WTN_STEP * ldq, &lds[buf][1][(wave * 2 + p) * 4 * WTN_T]); }
  };
  auto stage = [&](int step) {
    stage_piece(step, 0, 0); stage_piece(step, 0, 1); stage_piece(step, 1, 0); stage_piece(step, 1, 1);
  };
  // ---- fragment addresses (bytes inside one operand tile): lane (g = lane >> 4, i = lane & 15) supplies the piece
  // (row 8g + (i >> 2) [+4 for the upper half, +32 per k-half], 8 bytes at column 16 f + 4 (i & 3)) of fragment f
  const int g = lane >> 4, i16 = lane & 15;
  const int rowoff = (g * 8 + (i16 >> 2)) * 256 + ((i16 & 3) >> 1) * 16 + (i16 & 1) * 8;
  const int swb = 32 * ((i16 >> 2) | ((g & 1) << 2));
  const int wn = wave >> 2, wk = wave & 3;                        // 64 (n) x 32 (k) per wave
  int fa[4], fb[2];
#pragma unroll
  for (int f = 0; f < 4; f++) fa[f] = rowoff + (((wn * 64 + f * 16) * 2) ^ swb);
#pragma unroll
  for (int f = 0; f < 2; f++) fb[f] = rowoff + (((wk * 32 + f * 16) * 2) ^ swb);
  f32x4 acc[4][2], bacc[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = bias_grad && k0 == 0 && wk == 0;
  const s16x8_t ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)&lds[0][0][0];
#pragma unroll
  for (int s = 0; s < WTN_BUFS - 1; s++)
    if (s < steps) stage(s);
  for (int s = 0; s < steps; s++) {
    // step s has landed once at most the DMAs of the younger steps in flight (4 per step and wave) are outstanding
    const int younger = steps - 1 - s;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // everyone's part of step s is in LDS, step s-1 is consumed
    asm volatile("" ::: "memory");
    const bool more = s + WTN_BUFS - 1 < steps;                  // DMA of step s+3 goes into the buffer step s-1 used
    const int nxt = s + WTN_BUFS - 1;
    const uint32_t base = lds0 + (uint32_t)(s & (WTN_BUFS - 1)) * (2 * WTN_STEP * WTN_T * 2);
    uint32_t aa[4], ab[2];
#pragma unroll
    for (int f = 0; f < 4; f++) aa[f] = base + fa[f];
#pragma unroll
    for (int f = 0; f < 2; f++) ab[f] = base + WTN_STEP * WTN_T * 2 + fb[f];
    // order inside a step: all fragment reads first (LDS latency then runs under the DMA issue, which is the slow part:
    // every wave of the CU queues on the one texture path), the DMA pieces spread between the MFMA groups
    TnFrags f0, f1;
    tn_read<0>(f0, aa, ab);
    tn_read<1>(f1, aa, ab);
    if (more) { stage_piece(nxt, 0, 0); stage_piece(nxt, 0, 1); }
    lds_wait<12>(f0);                                            // LDS returns in order: the first 12 reads are back
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 2; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tn_operand(f0, a), tn_operand(f0, 4 + b), acc[a][b], 0, 0, 0);
      if (a == 1 && more) stage_piece(nxt, 1, 0);
    }
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; a++) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tn_operand(f0, a), ones, bacc[a], 0, 0, 0);
    }
    if (more) stage_piece(nxt, 1, 1);
    lds_wait<0>(f1);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tn_operand(f1, a), tn_operand(f1, 4 + b), acc[a][b], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; a++) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tn_operand(f1, a), ones, bacc[a], 0, 0, 0);
    }
  }
  // ---- lane holds D[n = 4g + e][k = i16] of each fragment pair
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int n = n0 + wn * 64 + a * 16 + 4 * g + e;
      if (n >= N) continue;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int k = k0 + wk * 32 + b * 16 + i16;
        if (k < K && !(n < zero_n && k >= zero_k0 && k < zero_k1)) {
          if (slab) C[(int64_t)n * ldc + k] = acc[a][b][e];           // this row chunk's slab (Go1PpoWgradProblem.partials): summed by the reduction pass
          else atomicAdd(C + (int64_t)n * ldc + k, acc[a][b][e]);
        }
      }
      if (do_bias && i16 == 0) atomicAdd(bias_grad + n, bacc[a][e]);
    }
  }
}

// ---- 256 (n) x 128 (k) tiles for problems whose n is a multiple of 256 (the 256-row first-layer block of the adaptation pass, the
// 512 -> 256 layers, Wz): the H / X operand of a step is staged once for 256 output rows instead of once per 128 — the kernel is bound by the
// L2 -> LDS staging rate, and this tile moves 48 KB per 1 M multiply-adds instead of 64.  Per step and buffer:
// [dZ image, columns 0..127 | dZ image, columns 128..255 | H image], 16 KB each, the images exactly those of the 128-tile body;
// three buffers (two steps in flight: the same 96 KB in flight per workgroup); each wave owns 64 (n) x 64 (k): 4 x 4 accumulators.
#define WTW_BUF_BYTES (3 * WTN_STEP * WTN_T * 2)
#define WTW_BUFS 3
struct TnFragsW { s16x4_t lo[8], hi[8]; };                       // 4 dZ fragments + 4 H fragments
template <int KK>
__device__ __forceinline__ void tnw_read(TnFragsW& f, const uint32_t (&aa)[4], const uint32_t (&ab)[4]) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f.lo[i] = lds_tr<KK * 32 * 256>(aa[i]);
    f.hi[i] = lds_tr<KK * 32 * 256 + 4 * 256>(aa[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f.lo[4 + i] = lds_tr<KK * 32 * 256>(ab[i]);
    f.hi[4 + i] = lds_tr<KK * 32 * 256 + 4 * 256>(ab[i]);
  }
}
__device__ __forceinline__ void ldsw_wait_all(TnFragsW& f) {       // (two statements: an asm takes at most 30 operands; volatile asms keep their order)
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.lo[0]), "+v"(f.lo[1]), "+v"(f.lo[2]), "+v"(f.lo[3]), "+v"(f.lo[4]), "+v"(f.lo[5]), "+v"(f.lo[6]), "+v"(f.lo[7]));
  asm volatile("" : "+v"(f.hi[0]), "+v"(f.hi[1]), "+v"(f.hi[2]), "+v"(f.hi[3]), "+v"(f.hi[4]), "+v"(f.hi[5]), "+v"(f.hi[6]), "+v"(f.hi[7]));
}
__device__ __forceinline__ bf16x8_t tnw_operand(const TnFragsW& f, int i) {
  s16x8_t v = __builtin_shufflevector(f.lo[i], f.hi[i], 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ void wgrad_tn_body_wide(const bf16_t* P, int ldp, const bf16_t* Q, int ldq, int64_t m_begin, int steps,
                                                   float* C, int ldc, float* bias_grad, int N, int K, int n0, int k0, char* lds,
                                                   int zero_n, int zero_k0, int zero_k1, bool slab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- staging: 32 row blocks (4 rows of 256 B) of the two dZ images + 16 of the H image per step; wave w moves dZ blocks 4w .. 4w+3
  // (image = block >> 4) and H blocks 2w, 2w+1
  const bf16_t* src[6];
  bool ok[6];
  int dst[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int id = q < 4 ? wave * 4 + q : wave * 2 + (q - 4);
    const int img = q < 4 ? id >> 4 : 2, rb = q < 4 ? id & 15 : id;
    const int row = rb * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (2 * ((row & 3) | (((row >> 3) & 1) << 2)));
    const int col = (q < 4 ? n0 + img * 128 : k0) + chunk * 8;
    ok[q] = col < (q < 4 ? N : K);                              // (n % 256 == 0 and k0 < K: every instruction keeps active lanes — the vmcnt bookkeeping relies on it)
    src[q] = (q < 4 ? P + (m_begin + row) * ldp : Q + (m_begin + row) * ldq) + (ok[q] ? col : 0);
    dst[q] = img * (WTN_STEP * WTN_T * 2) + rb * 4 * WTN_T * 2;
  }
  auto stage_piece = [&](int step, int buf, int q) {
    if (ok[q]) glds16(src[q] + (int64_t)step * WTN_STEP * (q < 4 ? ldp : ldq), (bf16_t*)(lds + buf * WTW_BUF_BYTES + dst[q]));
  };
  const int g = lane >> 4, i16 = lane & 15;
  const int rowoff = (g * 8 + (i16 >> 2)) * 256 + ((i16 & 3) >> 1) * 16 + (i16 & 1) * 8;
  const int swb = 32 * ((i16 >> 2) | ((g & 1) << 2));
  const int wn = wave >> 1, wk = wave & 1;                        // 64 (n) x 64 (k) per wave
  int fa[4], fb[4];
#pragma unroll
  for (int f = 0; f < 4; f++) {
    fa[f] = (wn >> 1) * (WTN_STEP * WTN_T * 2) + rowoff + ((((wn & 1) * 64 + f * 16) * 2) ^ swb);
    fb[f] = 2 * (WTN_STEP * WTN_T * 2) + rowoff + (((wk * 64 + f * 16) * 2) ^ swb);
  }
  f32x4 acc[4][4], bacc[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = bias_grad && k0 == 0 && wk == 0;
  const s16x8_t ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_bits);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)lds;
#pragma unroll
  for (int s = 0; s < WTW_BUFS - 1; s++)
    if (s < steps) {
#pragma unroll
      for (int q = 0; q < 6; q++) stage_piece(s, s, q);
    }
  int buf = 0;
  for (int s = 0; s < steps; s++) {
    // step s has landed once at most the 6 DMAs of step s+1 are outstanding
    if (s + 1 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // everyone's part of step s is in LDS, step s-1 is consumed
    asm volatile("" ::: "memory");
    const bool more = s + WTW_BUFS - 1 < steps;                  // the DMA of step s+2 goes into the buffer step s-1 used
    const int nxt = s + WTW_BUFS - 1, nbuf = buf == 0 ? WTW_BUFS - 1 : buf - 1;
    const uint32_t base = lds0 + (uint32_t)buf * WTW_BUF_BYTES;
    uint32_t aa[4], ab[4];
#pragma unroll
    for (int f = 0; f < 4; f++) { aa[f] = base + fa[f]; ab[f] = base + fb[f]; }
    TnFragsW f0, f1;
    tnw_read<0>(f0, aa, ab);
    if (more) { stage_piece(nxt, nbuf, 0); stage_piece(nxt, nbuf, 1); stage_piece(nxt, nbuf, 4); }
    ldsw_wait_all(f0);
    tnw_read<1>(f1, aa, ab);
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 4; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tnw_operand(f0, a), tnw_operand(f0, 4 + b), acc[a][b], 0, 0, 0);
      if (more && a < 3) stage_piece(nxt, nbuf, a == 0 ? 2 : (a == 1 ? 3 : 5));
    }
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; a++) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tnw_operand(f0, a), ones, bacc[a], 0, 0, 0);
    }
    ldsw_wait_all(f1);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tnw_operand(f1, a), tnw_operand(f1, 4 + b), acc[a][b], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; a++) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tnw_operand(f1, a), ones, bacc[a], 0, 0, 0);
    }
    buf = buf + 1 == WTW_BUFS ? 0 : buf + 1;
  }
  // ---- lane holds D[n = 4g + e][k = i16] of each fragment pair
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int n = n0 + wn * 64 + a * 16 + 4 * g + e;
      if (n >= N) continue;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int k = k0 + wk * 64 + b * 16 + i16;
        if (k < K && !(n < zero_n && k >= zero_k0 && k < zero_k1)) {
          if (slab) C[(int64_t)n * ldc + k] = acc[a][b][e];
          else atomicAdd(C + (int64_t)n * ldc + k, acc[a][b][e]);
        }
      }
      if (do_bias && i16 == 0) atomicAdd(bias_grad + n, bacc[a][e]);
    }
  }
}
// the tile shape of a problem: 256 x 128 when n is a multiple of 256 AND the partial tiles go to slabs, 128 x 128 otherwise (kernel and
// planner agree through these).  With atomics the wide tile loses: the same number of workgroups means twice the row chunks per output
// element, i.e. twice the atomics (measured: 256 x 512 alone 53 us against 34).
__host__ __device__ __forceinline__ bool wtn_wide(const Go1PpoWgradProblem& P) { return (P.n & 255) == 0 && P.partials != nullptr; }
__host__ __device__ __forceinline__ int wtn_tiles_n(const Go1PpoWgradProblem& P) { return wtn_wide(P) ? P.n >> 8 : (P.n + WTN_T - 1) / WTN_T; }

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// every weight gradient of a backward pass in ONE launch (problem table built by go1ppo_wgrad_tn_plan)
__global__ __launch_bounds__(WTN_THREADS, 1) void wgrad_tn_batched_kernel(const Go1PpoWgradProblem* __restrict__ probs, int count) {
  __shared__ __attribute__((aligned(1024))) char lds_raw[WTW_BUFS * WTW_BUF_BYTES];      // 144 KB (the 128-tile body uses the first 128 KB)
  static_assert(WTW_BUFS * WTW_BUF_BYTES >= WTN_BUFS * 2 * WTN_STEP * WTN_T * 2, "one LDS block for both tile shapes");
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  int p = 0;
  while (p + 1 < count && wg >= probs[p + 1].wg_offset) p++;
  const Go1PpoWgradProblem P = probs[p];
  const int local = wg - P.wg_offset;
  const bool wide = wtn_wide(P);
  const int tiles_k = (P.k + WTN_T - 1) / WTN_T, tiles = wtn_tiles_n(P) * tiles_k;
  const int tile = local % tiles, split = local / tiles;
  const int64_t m_begin = (int64_t)split * P.chunk_rows;
  const int64_t m_end = m_begin + P.chunk_rows < P.rows ? m_begin + P.chunk_rows : P.rows;
  const bool slab = P.partials != nullptr;
  float* C = slab ? P.partials + (int64_t)split * P.partial_stride : P.dW;
  const int steps = (int)((m_end - m_begin) / WTN_STEP);
  if (wide)
    wgrad_tn_body_wide((const bf16_t*)P.dz, P.ld_dz, (const bf16_t*)P.h, P.ld_h, m_begin, steps, C, P.ldw, P.bias_grad, P.n, P.k,
                       (tile / tiles_k) * 2 * WTN_T, (tile % tiles_k) * WTN_T, lds_raw, P.zero_n, P.zero_k0, P.zero_k1, slab);
  else
    wgrad_tn_body((const bf16_t*)P.dz, P.ld_dz, (const bf16_t*)P.h, P.ld_h, m_begin, steps, C, P.ldw, P.bias_grad, P.n, P.k,
                  (tile / tiles_k) * WTN_T, (tile % tiles_k) * WTN_T, reinterpret_cast<bf16_t (*)[2][WTN_STEP * WTN_T]>(lds_raw), P.zero_n, P.zero_k0,
                  P.zero_k1, slab);
}

extern "C" int go1ppo_wgrad_tn_plan(Go1PpoWgradProblem* probs, int count) {
  if (!probs || count <= 0) return -1;
  int64_t tile_steps = 0, tiles = 0;
  for (int i = 0; i < count; i++) {
    Go1PpoWgradProblem& P = probs[i];
    if (!P.dz || !P.h || (!P.dW && !P.partials) || (P.partials && P.partial_stride < (int64_t)P.n * P.ldw) || P.rows <= 0 || (P.rows % WTN_STEP) || P.n < 8 || P.k < 8 || (P.n & 7) || (P.k & 7) ||
        (P.ld_dz & 7) || (P.ld_h & 7) || !aligned16(P.dz) || !aligned16(P.h) || P.zero_n < 0 || P.zero_k0 < 0 || P.zero_k1 < P.zero_k0)
      return -1;
    // (a step of a 256 x 128 tile stages 48 KB instead of 32: it counts 1.5 steps, and its chunks are shorter by that factor)
    const int64_t t = (int64_t)wtn_tiles_n(P) * ((P.k + WTN_T - 1) / WTN_T);
    tiles += t;
    tile_steps += t * (P.rows / WTN_STEP) * (wtn_wide(P) ? 3 : 2) / 2;
  }
  // every partial tile costs a 64 KB slab write (or 16384 fp32 atomics), so chunks are long: one workgroup per CU and round, as few
  // rounds as give each workgroup <= 96 steps
  const int64_t cus = 256;
  int64_t rounds = (tile_steps + cus * 96 - 1) / (cus * 96);
  if (rounds < 1) rounds = 1;
  int64_t chunk_steps = (tile_steps + cus * rounds - 1) / (cus * rounds);
  if (chunk_steps < 4) chunk_steps = 4;
  for (;; chunk_steps++) {                                       // smallest chunk whose launch fits `rounds` full rounds
    int total = 0;
    for (int i = 0; i < count; i++) {
      Go1PpoWgradProblem& P = probs[i];
      const int64_t steps = P.rows / WTN_STEP;
      int64_t want = wtn_wide(P) ? chunk_steps * 2 / 3 : chunk_steps;
      if (want < 4) want = 4;
      const int64_t S0 = (steps + want - 1) / want;               // splits, then equalise the chunks
      const int64_t cs = (steps + S0 - 1) / S0;
      P.chunk_rows = (int32_t)(cs * WTN_STEP);
      P.wg_offset = total;
      total += (int)((steps + cs - 1) / cs) * wtn_tiles_n(P) * ((P.k + WTN_T - 1) / WTN_T);
    }
    if (total <= cus * rounds || chunk_steps >= 4096) return total;
  }
}

extern "C" int go1ppo_wgrad_tn_batched(const Go1PpoWgradProblem* device_probs, int count, int total_workgroups, void* stream) {
  if (!device_probs || count <= 0 || total_workgroups <= 0) return -1;
  wgrad_tn_batched_kernel<<<dim3((unsigned)total_workgroups), dim3(WTN_THREADS), 0, (hipStream_t)stream>>>(device_probs, count);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}
