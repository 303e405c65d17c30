// Diagnostic: are fp64 atomicAdds from many workgroups exact when several PROCESSES share the GPU?
// Every block adds (blockIdx + 1) to slot blockIdx % 8; the expected sums are exact integers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(double* acc, float* sink, int spin) {
  // some work first, so that blocks of different XCDs arrive spread out in time
  float v = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0000001f + 0.5f;
  if (v == 12345.678f) sink[0] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (blockIdx.x % 8), (double)(blockIdx.x + 1));
}
__global__ void zero(double* acc) { if (threadIdx.x < 8) acc[threadIdx.x] = 0.0; }
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000, nblk = 4096;
  double *acc, h[8], expect[8] = {0};
  float* sink;
  hipMalloc(&acc, 64);
  hipMalloc(&sink, 4);
  for (int b = 0; b < nblk; ++b) expect[b % 8] += (double)(b + 1);
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(zero, dim3(1), dim3(64), 0, 0, acc);
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, 0, acc, sink, 200 + (it % 7) * 100);
    hipMemcpy(h, acc, 64, hipMemcpyDeviceToHost);
    for (int k = 0; k < 8; ++k)
      if (h[k] != expect[k]) {
        if (bad < 5) printf("iter %d slot %d: got %.1f expected %.1f (diff %.1f)\n", it, k, h[k], expect[k], h[k] - expect[k]);
        ++bad;
      }
  }
  printf("fp64 atomic probe: %d mismatching slots in %d iterations\n", bad, iters);
  return 0;
}
