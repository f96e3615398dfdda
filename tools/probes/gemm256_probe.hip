// gemm256_probe.hip — a 256 x 256-tile LDS-DMA NT GEMM for the update's first-layer forward (X (24576 x 2112) W1^T (1280 x 2112), ELU in the
// epilogue): where does it spend its time, and can it beat hipBLASLt's 96 us?  It cannot (round 4; withdrawn from the product): the main loop
// is bound by the L2 -> LDS staging rate (~10.5 TB/s chip-wide: 1.01 GB staged per launch in 96 us with the MFMAs removed), 117 us with the MFMAs,
// + 17 us epilogue (LDS-staged full-line stores; 8-byte stores from the fragment layout cost 25 us, a per-fragment bounds check serialises them).
// Ablations of the kernel on the production shape
// (24576 x 2112 -> 1280), random bf16 operands, rotating buffers (operands come from HBM / the Infinity Cache, not a hot L2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o gemm256_probe tools/probes/gemm256_probe.hip && ./gemm256_probe
#include "../../walk-these-ways_amd/csrc/go1ppo.hip"
#define G256_T 256
#include <vector>
#include <cstdio>
#include <cstdlib>

// LOADS: 1 = stage every K-step (production), 0 = stage the first step only (MFMA + LDS reads alone)
// MFMA:  1 = production, 0 = fragment reads only
// STORE: 1 = production epilogue (8-byte stores), 0 = none (accumulators kept alive)
template <int LOADS, int MFMA, int STORE>
__global__ __launch_bounds__(512, 1) void probe_kernel(Go1PpoGemmArgs a) {
  __shared__ __attribute__((aligned(1024))) bf16_t lds[2][2][G256_T * GEMM_BK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tiles_n = (a.N + G256_T - 1) / G256_T;
  const int m0 = (logical / tiles_n) * G256_T, n0 = (logical % tiles_n) * G256_T;
  const int srow = lane >> 3;
  const bf16_t* ga[4];
  const bf16_t* gb[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int R = (wave * 4 + p) * 8 + srow;
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);
    const int ra = m0 + R < a.M ? m0 + R : a.M - 1;
    const int rb = n0 + R < a.N ? n0 + R : a.N - 1;
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
  const int fr = lane & 15, fg = lane >> 4;
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) foff[kk] = fr * GEMM_BK + (((kk * 4 + fg) ^ (fr >> 1)) << 3);
  const int wm = wave >> 2, wn = wave & 3;
  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int KT = a.K / GEMM_BK;
  stage(0, 0);
  if (!LOADS) stage(1, GEMM_BK);
  for (int kt = 0; kt < KT; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (LOADS && kt + 1 < KT) stage(buf ^ 1, (kt + 1) * GEMM_BK);
    const bf16_t* la = &lds[buf][0][wm * 128 * GEMM_BK];
    const bf16_t* lb = &lds[buf][1][wn * 64 * GEMM_BK];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8_t xa[8], wb[4];
#pragma unroll
      for (int j = 0; j < 4; j++) wb[j] = *reinterpret_cast<const bf16x8_t*>(lb + j * 16 * GEMM_BK + foff[kk]);
#pragma unroll
      for (int i = 0; i < 8; i++) xa[i] = *reinterpret_cast<const bf16x8_t*>(la + i * 16 * GEMM_BK + foff[kk]);
      if (MFMA) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[j][i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("" ::"v"(xa[i]));
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" ::"v"(wb[j]));
      }
    }
  }
  if (!STORE) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("" ::"v"(acc[j][i]));
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int n = n0 + wn * 64 + j * 16 + fg * 4;
    if (n >= a.N) continue;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int m = m0 + wm * 128 + i * 16 + fr;
      if (m >= a.M) continue;
      f32x2 lo = {acc[j][i][0], acc[j][i][1]}, hi = {acc[j][i][2], acc[j][i][3]};
      uint2 o;
      o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
      o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
      *reinterpret_cast<uint2*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = o;
    }
  }
}


// ---- experimental epilogues on the same main loop.  EP 1: branch-free direct 8-byte stores (interior tiles), ELU by select;
//      EP 2: the tile goes through LDS (bf16, row stride 528 B) and leaves as 16-byte stores of whole 512-byte row segments
template <int EP>
__global__ __launch_bounds__(512, 1) void probe2_kernel(Go1PpoGemmArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[256 * 528];
  bf16_t (*lds)[2][G256_T * GEMM_BK] = reinterpret_cast<bf16_t (*)[2][G256_T * GEMM_BK]>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tiles_n = (a.N + G256_T - 1) / G256_T;
  const int m0 = (logical / tiles_n) * G256_T, n0 = (logical % tiles_n) * G256_T;
  const int srow = lane >> 3;
  const bf16_t* ga[4];
  const bf16_t* gb[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int R = (wave * 4 + p) * 8 + srow;
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);
    const int ra = m0 + R < a.M ? m0 + R : a.M - 1;
    const int rb = n0 + R < a.N ? n0 + R : a.N - 1;
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
  const int fr = lane & 15, fg = lane >> 4;
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) foff[kk] = fr * GEMM_BK + (((kk * 4 + fg) ^ (fr >> 1)) << 3);
  const int wm = wave >> 2, wn = wave & 3;
  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 8; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int KT = a.K / GEMM_BK;
  stage(0, 0);
  for (int kt = 0; kt < KT; kt++) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) stage(buf ^ 1, (kt + 1) * GEMM_BK);
    const bf16_t* la = &lds[buf][0][wm * 128 * GEMM_BK];
    const bf16_t* lb = &lds[buf][1][wn * 64 * GEMM_BK];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8_t xa[8], wb[4];
#pragma unroll
      for (int j = 0; j < 4; j++) wb[j] = *reinterpret_cast<const bf16x8_t*>(lb + j * 16 * GEMM_BK + foff[kk]);
#pragma unroll
      for (int i = 0; i < 8; i++) xa[i] = *reinterpret_cast<const bf16x8_t*>(la + i * 16 * GEMM_BK + foff[kk]);
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[j][i], 0, 0, 0);
    }
  }
  // interior tiles only in this probe (M, N multiples of 256)
  if (EP == 1) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      const bool act = n >= a.elu_c0 && n < a.elu_c1 && !(n >= a.elu_skip_c0 && n < a.elu_skip_c1);
      bf16_t* crow = (bf16_t*)a.C + (int64_t)(m0 + wm * 128 + fr) * a.ldc + n;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { const float x = acc[j][i][e]; v[e] = act ? elu1(x) : x; }
        f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
        uint2 o;
        o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
        o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
        *reinterpret_cast<uint2*>(crow + (int64_t)i * 16 * a.ldc) = o;
      }
    }
  } else {
    __syncthreads();                                    // every wave is done with the operand buffers
    unsigned char* out = smem;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int nl = wn * 64 + j * 16 + fg * 4, n = n0 + nl;
      const bool act = n >= a.elu_c0 && n < a.elu_c1 && !(n >= a.elu_skip_c0 && n < a.elu_skip_c1);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { const float x = acc[j][i][e]; v[e] = act ? elu1(x) : x; }
        f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
        uint2 o;
        o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
        o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
        *reinterpret_cast<uint2*>(out + (wm * 128 + i * 16 + fr) * 528 + nl * 2) = o;
      }
    }
    __syncthreads();
    // 256 rows x 512 B: a wave instruction moves two rows (32 lanes x 16 B each); wave w takes rows 32w .. 32w+31
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const int row = wave * 32 + t * 2 + (lane >> 5), c16 = lane & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(out + row * 528 + c16 * 16);
      *reinterpret_cast<uint4*>((bf16_t*)a.C + (int64_t)(m0 + row) * a.ldc + n0 + c16 * 8) = v;
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void fill_random(bf16_t* d, size_t n, float scale, uint32_t seed) {
  std::vector<bf16_t> h(n);
  uint32_t s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    float f = ((int32_t)s) * (1.0f / 2147483648.0f) * scale;
    uint32_t u; memcpy(&u, &f, 4);
    h[i] = (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  CK(hipMemcpy(d, h.data(), n * sizeof(bf16_t), hipMemcpyHostToDevice));
}

template <typename F>
static float time_us(F&& launch, int iters = 30, int warm = 6) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < warm; i++) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; i++) launch(i);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 24576, N = 1280, K = 2112, R = 4;
  bf16_t *A[R], *C[R], *B;
  for (int i = 0; i < R; i++) {
    CK(hipMalloc(&A[i], (size_t)M * K * 2)); CK(hipMalloc(&C[i], (size_t)M * N * 2));
    fill_random(A[i], (size_t)M * K, 1.0f, 7 + i);
  }
  CK(hipMalloc(&B, (size_t)N * K * 2));
  fill_random(B, (size_t)N * K, 0.03f, 99);
  Go1PpoGemmArgs g[R];
  for (int i = 0; i < R; i++) {
    memset(&g[i], 0, sizeof(g[i]));
    g[i].A = A[i]; g[i].B = B; g[i].C = C[i]; g[i].M = M; g[i].N = N; g[i].K = K; g[i].lda = K; g[i].ldb = K; g[i].ldc = N;
    g[i].epilogue = 1; g[i].elu_c0 = 0; g[i].elu_c1 = N; g[i].elu_skip_c0 = 256; g[i].elu_skip_c1 = 768;
  }
  const double gf = 2.0 * M * N * K / 1e9;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  printf("M %d N %d K %d: %d tiles of 256x256, %.1f GFLOP\n", M, N, K, tiles, gf);
  auto rep = [&](const char* name, float us) { printf("%-58s %8.1f us  %7.0f TF/s\n", name, us, gf / us * 1e-3 * 1e3); };
  rep("go1ppo_gemm_nt (128 tile, ELU epilogue)", time_us([&](int i) { go1ppo_gemm_nt(&g[i % R], 0); }));
  rep("probe <loads, mfma, store>", time_us([&](int i) { probe_kernel<1, 1, 1><<<tiles, 512>>>(g[i % R]); }));
  rep("probe <loads, mfma, NO store>", time_us([&](int i) { probe_kernel<1, 1, 0><<<tiles, 512>>>(g[i % R]); }));
  rep("probe <loads, NO mfma, no store>", time_us([&](int i) { probe_kernel<1, 0, 0><<<tiles, 512>>>(g[i % R]); }));
  rep("probe <NO loads, mfma, no store>", time_us([&](int i) { probe_kernel<0, 1, 0><<<tiles, 512>>>(g[i % R]); }));
  rep("probe <NO loads, NO mfma, no store>  (fragment reads)", time_us([&](int i) { probe_kernel<0, 0, 0><<<tiles, 512>>>(g[i % R]); }));
  rep("probe <NO loads, mfma, store>", time_us([&](int i) { probe_kernel<0, 1, 1><<<tiles, 512>>>(g[i % R]); }));
  rep("probe2 <EP 1: branch-free direct stores + ELU>", time_us([&](int i) { probe2_kernel<1><<<tiles, 512>>>(g[i % R]); }));
  rep("probe2 <EP 2: LDS-staged full-line stores + ELU>", time_us([&](int i) { probe2_kernel<2><<<tiles, 512>>>(g[i % R]); }));
  {   // the two epilogues agree with the 128-tile product kernel bit for bit
    std::vector<bf16_t> h0((size_t)M * N), h1((size_t)M * N), h2((size_t)M * N);
    go1ppo_gemm_nt(&g[0], 0); CK(hipDeviceSynchronize()); CK(hipMemcpy(h0.data(), C[0], h0.size() * 2, hipMemcpyDeviceToHost));
    probe2_kernel<1><<<tiles, 512>>>(g[0]); CK(hipDeviceSynchronize()); CK(hipMemcpy(h1.data(), C[0], h1.size() * 2, hipMemcpyDeviceToHost));
    probe2_kernel<2><<<tiles, 512>>>(g[0]); CK(hipDeviceSynchronize()); CK(hipMemcpy(h2.data(), C[0], h2.size() * 2, hipMemcpyDeviceToHost));
    size_t d1 = 0, d2 = 0;
    for (size_t i = 0; i < h0.size(); i++) { d1 += h0[i] != h1[i]; d2 += h0[i] != h2[i]; }
    printf("elements differing from the 128-tile product kernel (go1ppo_gemm_nt): EP1 %zu, EP2 %zu of %zu\n", d1, d2, h0.size());
  }
  // same kernel on exactly one round of tiles (256) and on a single tile: per-tile latency
  for (int i = 0; i < R; i++) g[i].M = 256 * 256 / 5 / 256 * 256;      // 51 row blocks x 5 = 255 tiles
  const int t1 = (g[0].M / 256) * 5;
  printf("one round (%d tiles): %.1f us\n", t1, time_us([&](int i) { probe_kernel<1, 1, 1><<<t1, 512>>>(g[i % R]); }));
  for (int i = 0; i < R; i++) { g[i].M = 256; g[i].N = 256; }
  printf("one tile: %.1f us (<loads, mfma, store>), %.1f us (no store), %.1f us (no loads, no store)\n",
         time_us([&](int i) { probe_kernel<1, 1, 1><<<1, 512>>>(g[i % R]); }), time_us([&](int i) { probe_kernel<1, 1, 0><<<1, 512>>>(g[i % R]); }),
         time_us([&](int i) { probe_kernel<0, 1, 0><<<1, 512>>>(g[i % R]); }));
  return 0;
}
