// Fused Adam over a flat fp32 parameter buffer (torch.optim.Adam defaults, reference train_val.py:55-56).
#include "common.h"

const char* g_dis_last_kernel = nullptr;
// diagnostics: kernel family of the most recent conv dispatch of this process ("" if none since the last clear)
extern "C" int dis_last_kernel(char* name, int cap, int clear) {
  if (!name) return DIS_ERR_NULL;
  if (cap <= 0) return DIS_ERR_BAD_SHAPE;
  const char* r = g_dis_last_kernel ? g_dis_last_kernel : "";
  int i = 0;
  for (; i < cap - 1 && r[i]; ++i) name[i] = r[i];
  name[i] = 0;
  if (clear) g_dis_last_kernel = nullptr;
  return DIS_OK;
}

// omb1 / omb2 = (1 - beta) evaluated in double precision on the host and rounded once, as torch.optim.Adam's python floats
// are (1.f - 0.999f in fp32 is off by 1.3e-5 relative).
__global__ void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                            float4* __restrict__ v, long count4, float lr, float b1, float b2, float omb1, float omb2,
                            float eps, float bc1, float bc2_sqrt, float gscale) {
  const float step = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    float* P = (float*)&pp; float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale;
      M[k] = M[k] * b1 + gr * omb1;
      V[k] = V[k] * b2 + (gr * gr) * omb2;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step * (M[k] / denom);
    }
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

extern "C" int dis_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count,
                             float lr, double beta1, double beta2, float eps, int step_count, float grad_scale,
                             void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return DIS_ERR_NULL;
  if (count <= 0 || step_count <= 0) return DIS_ERR_BAD_SHAPE;
  if (count % 4 != 0) return DIS_ERR_UNSUPPORTED;   // the flat buffer is padded by the caller
  const double bc1 = 1.0 - pow(beta1, (double)step_count);
  const double bc2 = 1.0 - pow(beta2, (double)step_count);
  hipLaunchKernelGGL(adam_kernel, dim3(dis_ew_grid(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (float4*)param, (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, count / 4, lr,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// Graph-safe form: the step counter and the two bias corrections live on the device.  `state` is 4 x 32 bit:
// [0] int step count (number of steps taken so far), [1] float 1 - beta1^step, [2] float sqrt(1 - beta2^step), [3] unused.
// A one-thread kernel advances it, the update kernel reads it, so a hipGraph that captured ONE optimiser step applies
// step k's correction at its k-th replay (with the host-side form above the capture-time correction would be replayed).
__global__ void adam_advance_kernel(int* __restrict__ state, double b1, double b2) {
  const int step = state[0] + 1;
  state[0] = step;
  ((float*)state)[1] = (float)(1.0 - pow(b1, (double)step));
  ((float*)state)[2] = (float)sqrt(1.0 - pow(b2, (double)step));
}
__global__ void adam_dev_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                float4* __restrict__ v, long count4, float lr, float b1, float b2, float omb1, float omb2,
                                float eps, const int* __restrict__ state, float gscale) {
  const float bc1 = ((const float*)state)[1], bc2_sqrt = ((const float*)state)[2];
  const float step = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    float* P = (float*)&pp; float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale;
      M[k] = M[k] * b1 + gr * omb1;
      V[k] = V[k] * b2 + (gr * gr) * omb2;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] = P[k] - step * (M[k] / denom);
    }
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}
extern "C" int dis_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count,
                                 float lr, double beta1, double beta2, float eps, int* state, float grad_scale,
                                 void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  if (count % 4 != 0) return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(dis_ew_grid(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (float4*)param, (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, count / 4, lr,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, (const int*)state,
                     grad_scale);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
