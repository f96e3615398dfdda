// ds_read_b64_tr_b16 lane mapping probe: lds[i] = i; lane l (g = l >> 4, i = l & 15) supplies the address of the 4-element
// piece (row 4g + (i >> 2), columns 4 (i & 3) ..) of a [16][128] image; prints what every lane receives.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15;
  int a = (g * 4 + (i >> 2)) * 128 + (i & 3) * 4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + a));
  for (int e = 0; e < 4; e++) out[l * 4 + e] = v[e];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; e++) {
      int row = h[l * 4 + e] / 128, col = h[l * 4 + e] % 128;
      printf(" (%d,%d)", row, col);
      if (row != (l >> 4) * 4 + e || col != (l & 15)) bad++;
    }
    printf("\n");
  }
  printf("hypothesis lane l gets (row 4g+e, col l&15): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  return 0;
}
