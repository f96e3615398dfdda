// Known-traffic micro-kernels for calibrating the rocprofv3 HBM counters (FETCH_SIZE / WRITE_SIZE) against the access
// pattern of the step kernel: SoA rows read / written 4 bytes per lane, 64 contiguous bytes per 16-environment group
// (the 4 lanes of an environment read the same word), rows far apart.  Every kernel moves a known number of bytes once
// (buffers larger than the 256 MB infinity cache, cold), so counter / known bytes is the correction factor.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/traffic_calib tools/probes/traffic_calib.hip && /tmp/traffic_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// the step kernel's pattern: workgroup of 64 lanes = 16 envs x 4 legs; lane reads row r at env e: p[r * N + e]
__global__ void soa_read_quad(const float* __restrict__ p, float* __restrict__ out, int N, int rows) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 2);
  float acc = 0.f;
  for (int r = 0; r < rows; r++) acc += p[(size_t)r * N + e];
  if (acc == 123.456f) out[e] = acc;          // (never true: keeps the loads)
}
__global__ void soa_write_quad(float* __restrict__ p, int N, int rows) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 2);
  if ((threadIdx.x & 3) == 0)
    for (int r = 0; r < rows; r++) p[(size_t)r * N + e] = (float)r;
}
// fully coalesced reference: 256 contiguous bytes per wavefront
__global__ void flat_read(const float* __restrict__ p, float* __restrict__ out, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456f) out[0] = acc;
}
__global__ void flat_write(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}

int main() {
  const int N = 4096, rows = 20000;                       // 20000 rows x 4096 envs x 4 B = 328 MB
  const size_t n = (size_t)N * rows;
  float *a, *out;
  hipMalloc(&a, n * 4); hipMalloc(&out, N * 4);
  hipMemset(a, 0, n * 4);
  flat_write<<<4096, 256>>>(a, n); hipDeviceSynchronize();
  soa_read_quad<<<N / 16, 64>>>(a, out, N, rows); hipDeviceSynchronize();
  soa_write_quad<<<N / 16, 64>>>(a, N, rows); hipDeviceSynchronize();
  flat_read<<<4096, 256>>>(a, out, n); hipDeviceSynchronize();
  printf("known bytes per kernel: %zu\n", n * 4);
  return 0;
}
