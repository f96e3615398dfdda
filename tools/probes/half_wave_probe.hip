// Probe: does a wavefront with only its lower 32 (16, 8) lanes active issue VALU instructions faster than a full one on gfx950?
// (If the SIMD skips the passes of an all-inactive half, 8 environments per wavefront would halve the step kernel's VALU time.)
// hipcc --offload-arch=gfx950 -O3 -o half_wave_probe tools/probes/half_wave_probe.hip && ./half_wave_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(float* out, unsigned long long* cyc, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0001f, c = 0.5f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
      a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent chain: latency of back-to-back dependent FMAs
__global__ void dep(float* out, unsigned long long* cyc, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  float a = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 128; u++) a = fmaf(a, 1.0001f, 0.5f);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&cyc, sizeof(unsigned long long));
  const int iters = 256;
  for (int active : {64, 48, 32, 16, 8, 4}) {
    unsigned long long h1 = 0, h2 = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, cyc, active, iters);
      hipMemcpy(&h1, cyc, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(dep, dim3(1), dim3(64), 0, 0, out, cyc, active, iters);
      hipMemcpy(&h2, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("active lanes %2d: independent FMAs %.2f cycles/instr   dependent FMAs %.2f cycles/instr\n", active,
           (double)h1 / (iters * 128.0), (double)h2 / (iters * 128.0));
  }
  return 0;
}
