// Diagnostic 2: block reduction through LDS + two __syncthreads, then one fp64 atomicAdd per block (the pattern of the
// GroupNorm-statistics epilogues), with enough work in front to be preempted when another process needs the CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ double wave_sum(double v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__global__ __launch_bounds__(256) void probe(double* acc, float* sink, int spin) {
  __shared__ double sm[4];
  float v = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0000001f + 0.5f;
  if (v == 12345.678f) sink[0] = v;
  double mine = 1.0;           // exact: the block sum is 256
  for (int rep = 0; rep < 2; ++rep) {   // two reductions in a row, like r1 / r2
    double w = wave_sum(mine);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wid] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = 0.0;
      for (int i = 0; i < 4; ++i) r += sm[i];
      atomicAdd(acc + 2 * (blockIdx.x % 4) + rep, r);
    }
  }
}
__global__ void zero(double* acc) { if (threadIdx.x < 8) acc[threadIdx.x] = 0.0; }
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000, nblk = 64;   // 64 workgroups, like the 64-tile launch
  double *acc, h[8];
  float* sink;
  hipMalloc(&acc, 64);
  hipMalloc(&sink, 4);
  const double expect = 256.0 * nblk / 4;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, 0, acc);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, acc, sink, 2000 + (it % 7) * 500);
    hipMemcpy(h, acc, 64, hipMemcpyDeviceToHost);
    for (int k = 0; k < 8; ++k)
      if (h[k] != expect) {
        if (bad < 8) printf("iter %d slot %d: got %.1f expected %.1f (diff %.1f)\n", it, k, h[k], expect, h[k] - expect);
        ++bad;
      }
  }
  printf("block-reduce + fp64 atomic probe: %d mismatching slots in %d iterations\n", bad, iters);
  return 0;
}
