// Diagnostic (not part of libdis_hip.so): how much vector-issue room do the two fp16 MFMA shapes leave on gfx950?
// A "k-step" of conv_f16x2_kernel<32,32> per wave is 12 x v_mfma_f32_16x16x32_f16 (2 rows x 2 cout blocks x 3 products) plus
// ~12 ds_read_b128 and a few dozen VALU / memory instructions that must issue in the MFMAs' shadow.  The same MACs as
// 6 x v_mfma_f32_32x32x16_f16 (one 32-pixel x 32-cout tile, K = 16 per instruction): half the matrix instructions, each busy
// twice as long.  Variant V in {0: 16x16x32, 1: 32x32x16}; per iteration also NR ds_read_b128 and NV dependent-free VALU (v_fma).
// Launch: one workgroup of 512 threads per CU (2 waves per SIMD), `iters` iterations; the host times the launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int V, int NR, int NV>
__global__ __launch_bounds__(512) void issue_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) lds[i] = (unsigned short)(0x3c00 + (i & 15));
  __syncthreads();
  const unsigned short* p = lds + (lane & 15) * 80 + (lane >> 4) * 8 + (tid >> 6) * 1280;
  s16x8 fr[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) fr[j] = *(const s16x8*)(p + (j & 7) * 16);
  float va[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) va[j] = (float)(lane + j);
  f32x4 a4[4] = {};
  f32x16 a16[2] = {};
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          a4[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fr[(q * 4 + t) % 12]),
                                                        __builtin_bit_cast(f16x8, fr[(q + t + 5) % 12]), a4[t], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q)
        a16[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fr[(2 * q) % 12]),
                                                           __builtin_bit_cast(f16x8, fr[(2 * q + 5) % 12]), a16[q & 1], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) fr[j % 12] = *(const s16x8*)(p + ((it + j) & 7) * 16 + (j >> 3) * 640);
#pragma unroll
    for (int j = 0; j < NV; ++j) va[j & 7] = __builtin_fmaf(va[j & 7], 1.0001f, va[(j + 3) & 7]);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) s += a4[t][0] + a4[t][3];
  s += a16[0][0] + a16[1][7];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += va[j];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V, int NR, int NV>
static void run(float* out, int iters, int ncu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((issue_kernel<V, NR, NV>), dim3(ncu), dim3(512), 0, 0, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((issue_kernel<V, NR, NV>), dim3(ncu), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 waves x iters iterations; MACs per wave-iteration: 12 x 8192 = 6 x 16384 = 98304
  const double ns_per_iter = ms * 1e6 / iters;                // wall ns per (2 waves on a SIMD doing one iteration each)
  const double tflops = 2.0 * 98304.0 * 8 * ncu * iters / (ms * 1e-3) / 1e12;   // 8 waves per CU
  printf("%-10s NR %2d NV %3d : %7.1f ns per iteration pair, %7.1f TFLOP/s of fp16 matrix work\n", V ? "32x32x16" : "16x16x32", NR, NV,
         ns_per_iter, tflops);
}

int main(int argc, char** argv) {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int ncu = prop.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  hipMalloc(&out, (size_t)ncu * 512 * 4);
  printf("%s, %d CUs, %d iterations; one workgroup of 8 waves per CU\n", prop.name, ncu, iters);
  run<0, 0, 0>(out, iters, ncu);   run<1, 0, 0>(out, iters, ncu);
  run<0, 12, 0>(out, iters, ncu);  run<1, 12, 0>(out, iters, ncu);
  run<0, 12, 24>(out, iters, ncu); run<1, 12, 24>(out, iters, ncu);
  run<0, 12, 48>(out, iters, ncu); run<1, 12, 48>(out, iters, ncu);
  run<0, 12, 96>(out, iters, ncu); run<1, 12, 96>(out, iters, ncu);
  run<0, 8, 48>(out, iters, ncu);  run<1, 8, 48>(out, iters, ncu);
  hipFree(out);
  return 0;
}
