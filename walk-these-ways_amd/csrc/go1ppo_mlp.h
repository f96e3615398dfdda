// go1ppo_mlp.h — the 256 -> 128 -> 64 end of the three MLPs (actor / critic behind their 512 -> 256 layer, the
// adaptation module behind the shared first layer), forward and backward, with BOTH weight matrices resident in LDS.
//
// In the reference these are two nn.Linear + ELU per net (actor_critic.py:51-92); as separate launches they are, per
// net and direction, two small GEMMs and two element-wise passes over (rows x 256/128) activations that are all
// launch / latency bound (6-11 us each at 24576 rows).  Here one workgroup owns 64 rows:
//   forward : h0 = elu(x) (x: the pre-activation the previous GEMM left, activated IN PLACE for the backward pass),
//             z2 = elu(h0 W2^T + b2) (kept on chip and written out for the backward pass), out = z2 W3^T + b3;
//   backward: dz2 = (d_out W3) * elu'(z2), dx = (dz2 W2) * elu'(h0); dz2 and dx are what the weight-gradient kernel
//             and the layer in front need, nothing else is written.
// bf16 MFMA 16x16x32 throughout.  LDS images have 256-byte rows (128 bf16; wider matrices = several images side by
// side) with the 16-byte chunk index XOR-swizzled: by (row & 15) for operands whose reduction index is contiguous
// (ds_read_b128 fragments), by 2 ((row & 3) | ((row >> 3) & 1) << 2) for the backward pass's weights, whose reduction
// index is the ROW (ds_read_b64_tr_b16 fragments, as in the weight-gradient kernel).  Weights arrive by LDS-DMA with
// the swizzle applied on the source address; the weight operand is always the MFMA "A" side, so a lane ends up with 4
// consecutive output columns of one row and all global traffic is 8/16-byte wide.
#pragma once

#define MLP2_K1 256
#define MLP2_N2 128
#define MLP2_N3 64
#define MLP2_BM 64
#define MLP2_WAVES 16
#define MLP2_THREADS (64 * MLP2_WAVES)
#define MLP2_IMG (128 * 128)          // elements of a [128 rows][128 columns] image (weights); row tiles use the first 64 rows

struct Mlp2FwdArgs { Go1PpoMlp2Fwd net[GO1PPO_MLP2_MAX_NETS]; };
struct Mlp2BwdArgs { Go1PpoMlp2Bwd net[GO1PPO_MLP2_MAX_NETS]; };

__device__ __forceinline__ int swz_row(int row) { return row & 15; }
__device__ __forceinline__ int swz_red(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

// W [rows][cols] row-major bf16 -> cols/128 images [rows][128], chunk swizzle by `RED ? swz_red : swz_row`; all waves
template <bool RED>
__device__ __forceinline__ void dma_weights(const bf16_t* W, int rows, int cols, bf16_t* img, int wave, int lane) {
  const int windows = cols >> 7, count = (rows >> 2) * windows;         // one DMA instruction = 4 rows of one window
  for (int idx = wave; idx < count; idx += MLP2_WAVES) {
    const int win = idx % windows, blk = idx / windows;
    const int row = blk * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (RED ? swz_red(row) : swz_row(row));
    glds16(W + (int64_t)row * cols + win * 128 + chunk * 8, img + win * MLP2_IMG + blk * 4 * 128);
  }
}

__device__ __forceinline__ uint2 pack_bf4(const float (&v)[4]) {
  f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
  uint2 o;
  o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2_t));
  o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2_t));
  return o;
}
__device__ __forceinline__ void unpack_bf4(uint2 r, float (&v)[4]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ bf16x8_t lds_b128(const bf16_t* img, int byte_off) {
  return *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const char*>(img) + byte_off);
}
// transpose-read operand: rows (reduction) 8g + (i16 >> 2) [+4] of a 32-row slab, 16 columns starting at `col`
__device__ __forceinline__ bf16x8_t lds_tr_operand(const bf16_t* img, int byte_off) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
  const char* p = reinterpret_cast<const char*>(img) + byte_off;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 4 * 256));
  s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// ------------------------------------------------------------------------------------------------------------ forward
// 16 waves per workgroup (one workgroup per CU: the weights take 80 KB of its LDS), so that the element-wise stages and
// the LDS / MFMA latencies of a tile overlap across 4 waves per SIMD; the workgroup keeps the weights and walks over
// row tiles with the next tile's rows in flight during the MFMAs.
__global__ __launch_bounds__(MLP2_THREADS, 1) void mlp2_fwd_kernel(Mlp2FwdArgs A) {
  __shared__ __attribute__((aligned(1024))) bf16_t W2i[2 * MLP2_IMG];     // [k window][n2][128 k]      64 KB
  __shared__ __attribute__((aligned(1024))) bf16_t W3i[64 * 128];         // [n3][128 k]                16 KB
  __shared__ __attribute__((aligned(1024))) bf16_t Ai[2 * 64 * 128];      // [k window][m][128 k]       32 KB
  __shared__ __attribute__((aligned(1024))) bf16_t Z2i[64 * 128];         // [m][128 k]                 16 KB
  const Go1PpoMlp2Fwd& N = A.net[blockIdx.y];
  const int tiles = (int)((N.rows + MLP2_BM - 1) / MLP2_BM);
  if ((int)blockIdx.x >= tiles) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, g = lane >> 4;
  dma_weights<false>((const bf16_t*)N.W2, MLP2_N2, MLP2_K1, W2i, wave, lane);
  dma_weights<false>((const bf16_t*)N.W3, MLP2_N3, MLP2_N2, W3i, wave, lane);
  bf16_t* x = (bf16_t*)N.x;
  bf16_t* z2 = (bf16_t*)N.z2;
  bf16_t* out = (bf16_t*)N.out;
  // lane offset of a b128 fragment read: row r of the fragment, chunk (4 (ks & 3) + g) ^ r = (4 (ks & 3)) ^ (g ^ r)
  const int tx = g ^ r;
  int foff[4];
#pragma unroll
  for (int a = 0; a < 4; a++) foff[a] = r * 256 + (((4 * a) ^ tx) << 4);
  // hidden layer: wave -> n fragment wave >> 1, row fragments 2 (wave & 1), +1; head: n3 fragment wave & 3, row fragment wave >> 2
  const int nf2 = wave >> 1, mp = (wave & 1) * 2, nf3 = wave & 3, m3 = wave >> 2;
  float bias2[4], bias3[4];
  unpack_bf4(*reinterpret_cast<const uint2*>((const bf16_t*)N.b2 + nf2 * 16 + 4 * g), bias2);
  unpack_bf4(*reinterpret_cast<const uint2*>((const bf16_t*)N.b3 + nf3 * 16 + 4 * g), bias3);

  Bf8 v[2];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = t + MLP2_THREADS * j, row = p >> 5, ch = p & 31;
      int64_t grow = (int64_t)tile * MLP2_BM + row;
      grow = grow < N.rows ? grow : N.rows - 1;
      v[j] = *reinterpret_cast<const Bf8*>(x + grow * N.ld_x + ch * 8);
    }
  };
  fetch(blockIdx.x);
  bool first = true;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * MLP2_BM;
    // ---- input rows: ELU, write-back, swizzled LDS image
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = t + MLP2_THREADS * j, row = p >> 5, ch = p & 31;
      if (N.elu_input) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = elu1(bf2f(v[j].v[e]));
        v[j] = pack_bf8(f);
        if (row0 + row < N.rows) *reinterpret_cast<Bf8*>(x + (row0 + row) * N.ld_x + ch * 8) = v[j];
      }
      *reinterpret_cast<Bf8*>(reinterpret_cast<char*>(Ai + (ch >> 4) * 64 * 128) + row * 256 + (((ch & 15) ^ swz_row(row)) << 4)) = v[j];
    }
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the weights
    first = false;
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
    // ---- hidden layer
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < MLP2_K1 / 32; ks++) {
        const int win = ks >> 2, off = foff[ks & 3];
        const bf16x8_t w = lds_b128(W2i + win * MLP2_IMG + nf2 * 16 * 128, off);
#pragma unroll
        for (int m = 0; m < 2; m++)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, lds_b128(Ai + win * 64 * 128 + (mp + m) * 16 * 128, off), acc[m], 0, 0, 0);
      }
      const int n0 = nf2 * 16 + 4 * g;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int row = (mp + m) * 16 + r;
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = elu1(acc[m][e] + bias2[e]);
        const uint2 o = pack_bf4(o4);
        if (row0 + row < N.rows) *reinterpret_cast<uint2*>(z2 + (row0 + row) * N.ld_z2 + n0) = o;
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(Z2i) + row * 256 + (((n0 >> 3) ^ swz_row(row)) << 4) + ((n0 >> 2) & 1) * 8) = o;
      }
    }
    __syncthreads();
    // ---- head
    {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < MLP2_N2 / 32; ks++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_b128(W3i + nf3 * 16 * 128, foff[ks]), lds_b128(Z2i + m3 * 16 * 128, foff[ks]), acc, 0, 0, 0);
      const int n0 = nf3 * 16 + 4 * g, row = m3 * 16 + r;
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; e++) o4[e] = acc[e] + bias3[e];
      if (row0 + row < N.rows) *reinterpret_cast<uint2*>(out + (row0 + row) * N.ld_out + n0) = pack_bf4(o4);
    }
    // the next iteration's Ai / Z2i writes sit behind barriers every wave only reaches once it is done reading here
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
// All global traffic of a tile is whole rows (16 bytes per lane, consecutive lanes on consecutive addresses): d_out, z2
// and h come in through registers one tile ahead and are laid out in LDS; the MFMA epilogues read the activation they
// multiply by from LDS and write the gradient back INTO THE SAME LDS SLOT; the finished dz2 / dx tiles then leave LDS
// row by row.  (Fragment-shaped 8-byte global accesses — 16 rows x 32 B per instruction — made this kernel 3x slower.)
__global__ __launch_bounds__(MLP2_THREADS, 1) void mlp2_bwd_kernel(Mlp2BwdArgs A) {
  __shared__ __attribute__((aligned(1024))) bf16_t W3t[64 * 128];         // [n3][128 k2], reduction = row        16 KB
  __shared__ __attribute__((aligned(1024))) bf16_t W2t[2 * MLP2_IMG];     // [k1 window][n2][128 k1]             64 KB
  __shared__ __attribute__((aligned(1024))) bf16_t D3i[64 * 128];         // [m][64 n3 (of 128)]                 16 KB
  __shared__ __attribute__((aligned(1024))) bf16_t D2i[64 * 128];         // [m][128]: z2, then dz2              16 KB
  __shared__ __attribute__((aligned(1024))) bf16_t Hi[2 * 64 * 128];      // [k1 window][m][128]: h, then dx     32 KB
  const Go1PpoMlp2Bwd& N = A.net[blockIdx.y];
  const int tiles = (int)((N.rows + MLP2_BM - 1) / MLP2_BM);
  if ((int)blockIdx.x >= tiles) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, g = lane >> 4;
  dma_weights<true>((const bf16_t*)N.W3, MLP2_N3, MLP2_N2, W3t, wave, lane);
  dma_weights<true>((const bf16_t*)N.W2, MLP2_N2, MLP2_K1, W2t, wave, lane);
  const bf16_t* d = (const bf16_t*)N.d_out;
  const bf16_t* z2 = (const bf16_t*)N.z2;
  const bf16_t* hin = (const bf16_t*)N.h;
  bf16_t* dz2 = (bf16_t*)N.d_z2;
  bf16_t* dx = (bf16_t*)N.d_x;
  const int tx = g ^ r;
  int foff[4];                                                            // b128 fragments (activation gradients)
#pragma unroll
  for (int a = 0; a < 4; a++) foff[a] = r * 256 + (((4 * a) ^ tx) << 4);
  // transpose-read pieces (weights): row 8g + (r >> 2) of a 32-row slab, 8 bytes at column 16 f + 4 (r & 3)
  const int troff = (g * 8 + (r >> 2)) * 256 + ((r & 3) >> 1) * 16 + (r & 1) * 8;
  const int swb = 32 * ((r >> 2) | ((g & 1) << 2));
  // dz2: wave -> k2 fragment wave >> 1, row fragments 2 (wave & 1), +1; dx: k1 fragment wave, all 4 row fragments
  const int kf2 = wave >> 1, mp = (wave & 1) * 2;
  const int w3off = troff + (((kf2 * 16) * 2) ^ swb);
  const bf16_t* w2img = W2t + (wave >> 3) * MLP2_IMG;                     // 8 fragments per 128-column window
  const int w2off = troff + ((((wave & 7) * 16) * 2) ^ swb);
  // whole-row pieces of this thread: h / dx rows (32 chunks each): pieces t, t + 1024; z2 / dz2 rows (16 chunks): piece t;
  // d_out rows (8 chunks): piece t of the first 512 threads.  LDS slot of chunk c of row q: q * 256 + ((c ^ q) & 15) * 16
  const int hch = t & 31;
#define MLP2_HROW(j) ((t + (j) * MLP2_THREADS) >> 5)
  const int zrow = t >> 4, zch = t & 15, drow = (t >> 3) & 63, dch = t & 7;
#define MLP2_HSLOT(j) ((hch >> 4) * (64 * 128 * 2) + MLP2_HROW(j) * 256 + (((hch & 15) ^ swz_row(MLP2_HROW(j))) << 4))
  const int zslot = zrow * 256 + ((zch ^ swz_row(zrow)) << 4), dslot = drow * 256 + ((dch ^ swz_row(drow)) << 4);

  uint4 dv = make_uint4(0, 0, 0, 0), zv, hv0, hv1;
#define MLP2_BWD_FETCH(tile_)                                                                                          \
  {                                                                                                                    \
    const int64_t base = (int64_t)(tile_) * MLP2_BM, last = N.rows - 1;                                                \
    int64_t q = base + drow; q = q < last ? q : last;                                                                  \
    if (t < 512) dv = *reinterpret_cast<const uint4*>(d + q * N.ld_dout + dch * 8);                                      \
    q = base + zrow; q = q < last ? q : last;                                                                          \
    zv = *reinterpret_cast<const uint4*>(z2 + q * N.ld_z2 + zch * 8);                                                    \
    q = base + MLP2_HROW(0); q = q < last ? q : last;                                                                  \
    hv0 = *reinterpret_cast<const uint4*>(hin + q * N.ld_h + hch * 8);                                                   \
    q = base + MLP2_HROW(1); q = q < last ? q : last;                                                                  \
    hv1 = *reinterpret_cast<const uint4*>(hin + q * N.ld_h + hch * 8);                                                   \
  }
  MLP2_BWD_FETCH(blockIdx.x);
  bool first = true;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * MLP2_BM;
    if (t < 512) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(D3i) + dslot) = dv;
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(D2i) + zslot) = zv;
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Hi) + MLP2_HSLOT(0)) = hv0;
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Hi) + MLP2_HSLOT(1)) = hv1;
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the weights
    first = false;
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) MLP2_BWD_FETCH(tile + gridDim.x);
    // ---- dz2[m][k2] = (d_out W3) * elu'(z2)
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < MLP2_N3 / 32; kk++) {
        const bf16x8_t w = lds_tr_operand(W3t, kk * 32 * 256 + w3off);
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, lds_b128(D3i + (mp + m) * 16 * 128, foff[kk]), acc[m], 0, 0, 0);
      }
      const int k0 = kf2 * 16 + 4 * g;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int row = (mp + m) * 16 + r;
        uint2* slot = reinterpret_cast<uint2*>(reinterpret_cast<char*>(D2i) + row * 256 + (((k0 >> 3) ^ swz_row(row)) << 4) + ((k0 >> 2) & 1) * 8);
        float h[4], o4[4];
        unpack_bf4(*slot, h);
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = acc[m][e] * (h[e] > 0.f ? 1.f : h[e] + 1.f);
        *slot = pack_bf4(o4);
      }
    }
    __syncthreads();
    // ---- dx[m][k1] = (dz2 W2) * elu'(h)
    {
      f32x4 acc[4];
#pragma unroll
      for (int m = 0; m < 4; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < MLP2_N2 / 32; kk++) {
        const bf16x8_t w = lds_tr_operand(w2img, kk * 32 * 256 + w2off);
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, lds_b128(D2i + m * 16 * 128, foff[kk]), acc[m], 0, 0, 0);
      }
      const int k0 = (wave & 7) * 16 + 4 * g;                             // column inside the wave's 128-column window
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int row = m * 16 + r;
        uint2* slot = reinterpret_cast<uint2*>(reinterpret_cast<char*>(Hi + (wave >> 3) * 64 * 128) + row * 256 +
                                               (((k0 >> 3) ^ swz_row(row)) << 4) + ((k0 >> 2) & 1) * 8);
        float h[4], o4[4];
        unpack_bf4(*slot, h);
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = acc[m][e] * (h[e] > 0.f ? 1.f : h[e] + 1.f);
        *slot = pack_bf4(o4);
      }
    }
    __syncthreads();
    // ---- the finished tiles leave row by row
    if (row0 + zrow < N.rows) *reinterpret_cast<uint4*>(dz2 + (row0 + zrow) * N.ld_dz2 + zch * 8) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(D2i) + zslot);
#pragma unroll
    for (int j = 0; j < 2; j++)
      if (row0 + MLP2_HROW(j) < N.rows)
        *reinterpret_cast<uint4*>(dx + (row0 + MLP2_HROW(j)) * N.ld_dx + hch * 8) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Hi) + MLP2_HSLOT(j));
    __syncthreads();                                                      // before the next tile's rows overwrite D2i / Hi
  }
}

static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

extern "C" int go1ppo_mlp2_fwd(const Go1PpoMlp2Fwd* nets, int count, void* stream) {
  if (!nets || count <= 0 || count > GO1PPO_MLP2_MAX_NETS) return -1;
  Mlp2FwdArgs A;
  int64_t max_rows = 0;
  for (int i = 0; i < count; i++) {
    const Go1PpoMlp2Fwd& N = nets[i];
    if (!N.x || !N.W2 || !N.b2 || !N.W3 || !N.b3 || !N.z2 || !N.out || N.rows <= 0) return -1;
    if ((N.ld_x & 7) || (N.ld_z2 & 3) || (N.ld_out & 3) || N.ld_x < MLP2_K1 || N.ld_z2 < MLP2_N2 || N.ld_out < MLP2_N3 || !aligned16(N.x) ||
        !aligned16(N.W2) || !aligned16(N.W3) || !aligned8(N.b2) || !aligned8(N.b3) || !aligned8(N.z2) || !aligned8(N.out))
      return -2;
    A.net[i] = N;
    if (N.rows > max_rows) max_rows = N.rows;
  }
  int64_t wgs = (max_rows + MLP2_BM - 1) / MLP2_BM;                       // one workgroup per CU (the weights take 80 KB of
  if (wgs > 256 / count) wgs = 256 / count;                               // its LDS), each walking over its row tiles
  mlp2_fwd_kernel<<<dim3((unsigned)wgs, count), dim3(MLP2_THREADS), 0, (hipStream_t)stream>>>(A);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

extern "C" int go1ppo_mlp2_bwd(const Go1PpoMlp2Bwd* nets, int count, void* stream) {
  if (!nets || count <= 0 || count > GO1PPO_MLP2_MAX_NETS) return -1;
  Mlp2BwdArgs A;
  int64_t max_rows = 0;
  for (int i = 0; i < count; i++) {
    const Go1PpoMlp2Bwd& N = nets[i];
    if (!N.d_out || !N.z2 || !N.h || !N.W2 || !N.W3 || !N.d_z2 || !N.d_x || N.rows <= 0) return -1;
    if ((N.ld_dout & 7) || (N.ld_z2 & 7) || (N.ld_h & 7) || (N.ld_dz2 & 7) || (N.ld_dx & 7) || N.ld_dout < MLP2_N3 || N.ld_z2 < MLP2_N2 ||
        N.ld_h < MLP2_K1 || N.ld_dz2 < MLP2_N2 || N.ld_dx < MLP2_K1 || !aligned16(N.d_out) || !aligned16(N.W2) || !aligned16(N.W3) ||
        !aligned16(N.z2) || !aligned16(N.h) || !aligned16(N.d_z2) || !aligned16(N.d_x))
      return -2;
    A.net[i] = N;
    if (N.rows > max_rows) max_rows = N.rows;
  }
  int64_t wgs = (max_rows + MLP2_BM - 1) / MLP2_BM;
  if (wgs > 256 / count) wgs = 256 / count;
  mlp2_bwd_kernel<<<dim3((unsigned)wgs, count), dim3(MLP2_THREADS), 0, (hipStream_t)stream>>>(A);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}
