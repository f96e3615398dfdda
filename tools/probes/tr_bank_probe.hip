// LDS read throughput by instruction, address pattern and waves per CU: each wave issues REPS x 16 reads;
// prints LDS cycles per wave-instruction = elapsed / (REPS * 16 * waves).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tr_bank_probe.hip -o tr_bank_probe && ./tr_bank_probe
// (measured: the wgrad_tn / mlp2 swizzled images read at the same rate as a contiguous image; 256-byte rows
//  without swizzle 2.4x slower at 8 waves)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) int i4;
#define REPS 1024
template <int OP>
__global__ __launch_bounds__(512) void k(int pattern, long long* cycles, int* sink) {
  __shared__ __attribute__((aligned(1024))) short lds[65536 / 2 * 2];
  for (int i = threadIdx.x; i < 65536; i += blockDim.x) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x & 63, g = l >> 4, i = l & 15;
  int a;
  if (OP == 0) {                                           // ds_read_b64_tr_b16: 8-byte pieces
    switch (pattern) {
      case 0: a = g * 128 + (i >> 2) * 32 + (i & 3) * 8; break;                                 // contiguous 512 B
      case 1: a = (g * 8 + (i >> 2)) * 256 + 32 * ((i >> 2) | ((g & 1) << 2)) + (i & 3) * 8; break;   // wgrad_tn: 256-B rows + XOR
      case 2: a = (g * 8 + (i >> 2)) * 256 + (i & 3) * 8; break;                                // 256-B rows, same columns
      default: a = (g * 4 + (i >> 2)) * 64 + (i & 3) * 8; break;                                // 64-B rows
    }
  } else {                                                 // ds_read_b128 fragment read of a [16 rows][.] image
    switch (pattern) {
      case 0: a = l * 16; break;                                                                // contiguous 1 KB
      case 1: a = i * 128 + ((g ^ (i >> 1)) & 7) * 16; break;                                   // gemm_nt: 128-B rows + XOR
      case 2: a = i * 128 + g * 16; break;                                                      // 128-B rows linear
      default: a = i * 256 + g * 16; break;                                                     // 256-B rows linear
    }
  }
  a += (threadIdx.x >> 6) * 8192;
  i4 acc = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; r++) {
    if (OP == 0) {
      s4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a), "n"((u & 1) * 4096));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; u++) { acc[0] ^= v[u][0] | (v[u][1] << 16); acc[1] ^= v[u][2] | (v[u][3] << 16); }
    } else {
      i4 v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a), "n"((u & 1) * 4096));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 16; u++) acc ^= v[u];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (l == 0) cycles[threadIdx.x >> 6] = t1 - t0;
  sink[threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
int main() {
  long long* c; int* s; long long h[8];
  (void)hipMalloc(&c, 64); (void)hipMalloc(&s, 2048);
  for (int op = 0; op < 2; op++)
    for (int p = 0; p < 4; p++) {
      printf("%s pattern %d:", op ? "ds_read_b128     " : "ds_read_b64_tr_b16", p);
      for (int waves = 1; waves <= 8; waves *= 2) {
        if (op == 0) k<0><<<1, 64 * waves>>>(p, c, s); else k<1><<<1, 64 * waves>>>(p, c, s);
        (void)hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("  %dw %.1f", waves, (double)h[0] / (REPS * 16.0 * waves));
      }
      printf("   (cycles per wave-instruction at the CU)\n");
    }
  return 0;
}
