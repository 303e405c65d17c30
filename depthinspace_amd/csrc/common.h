// Shared device/host helpers for libdis_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dis_hip.h"

#define DIS_WAVE 64

#define DIS_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

// Diagnostics only (bench.py's per-call timing labels a call by the kernel family that served it): the conv dispatchers leave
// the name of the kernel template they launched here; dis_last_kernel() hands it out.  Never read by any compute path.
extern const char* g_dis_last_kernel;
#define DIS_TAG(name) (g_dis_last_kernel = (name))

static inline int dis_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, never more than needed
static inline int dis_ew_grid(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// grid size for a grid-stride kernel that ENDS in same-address fp64 atomics (one per block): those serialise at ~7 ns
// each, so 2048 blocks cost more in their tail than the loop saves; 2 blocks per CU keep the latency hidden
static inline int dis_red_grid(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}

// Wave-wide sums with DPP row operations (no LDS traffic, fixed summation order): quad swaps, row rotations, then the
// row_bcast steps; lane 63 holds the total, which is broadcast - the result is valid in EVERY lane.
#define DIS_DPP_STEPS(STEP) STEP(0xB1, 0xf) STEP(0x4E, 0xf) STEP(0x124, 0xf) STEP(0x128, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
__device__ __forceinline__ float wave_sum(float v) {
#ifdef DIS_WAVE_SUM_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
#else
#define DIS_STEP_F(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true));
  DIS_DPP_STEPS(DIS_STEP_F)
#undef DIS_STEP_F
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}
__device__ __forceinline__ double wave_sum_d(double v) {
#ifdef DIS_WAVE_SUM_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
#else
#define DIS_STEP_D(ctrl, rmask)                                                                 \
  {                                                                                             \
    const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, true);  \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, true);  \
    v += __hiloint2double(hi_, lo_);                                                            \
  }
  DIS_DPP_STEPS(DIS_STEP_D)
#undef DIS_STEP_D
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
#endif
}

// Block-wide sum of a double; result valid in thread 0.  `sm` needs blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}

// (fire-and-forget: a returning atomic + wait was tried against the sharing anomaly of round 2 and changed nothing - the
// anomaly was a packed-fp32 hazard, see the Makefile)
__device__ __forceinline__ void atomic_add_d(double* p, double v) { atomicAdd(p, v); }

#define SELU_ALPHA_F 1.6732632423543772848170429916717f
#define SELU_SCALE_F 1.0507009873554804934193349852946f

// exp for the SELU negative branch: v_exp_f32 on x*log2(e) (argument <= 0, result in (0,1]; a few ulp, the same order as
// one fp32 rounding of the result) instead of the ~10-instruction library expf: the conv / Conv3D epilogues are
// VALU-issue-bound.
__device__ __forceinline__ float selu_exp(float x) { return __expf(x); }

__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == DIS_ACT_SELU) return x > 0.f ? SELU_SCALE_F * x : (SELU_SCALE_F * SELU_ALPHA_F) * (selu_exp(x) - 1.f);
  if (act == DIS_ACT_RELU) return x > 0.f ? x : 0.f;
  return x;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  if (act == DIS_ACT_SELU) return y > 0.f ? SELU_SCALE_F : (y + SELU_SCALE_F * SELU_ALPHA_F);
  if (act == DIS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// GroupNorm(1 group) moments of sample n from its fp64 sum / sum of squares over m elements (shared by the GroupNorm kernels
// and by the conv kernels that apply the normalisation while they stage their input: identical bits everywhere)
__device__ __forceinline__ void gn_moments(const double* stats, int n, double m, float eps, float* mean, float* rstd) {
  const double mu = stats[2 * n] / m;
  double var = stats[2 * n + 1] / m - mu * mu;
  if (var < 0.0) var = 0.0;
  *mean = (float)mu;
  *rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// The reference normalises pixel coordinates to [-1,1] and grid_sample maps them back
// (networks.py:363-364 -> ATen grid_sampler, align_corners=True).  Reproduce that round trip so that
// sampling positions carry the same fp32 rounding as the reference's.
__device__ __forceinline__ float gs_roundtrip(float p, int size) {
  float g = 2.f * (p / (float)(size - 1) - 0.5f);
  return (g + 1.f) * ((float)(size - 1) / 2.f);
}

// ------------------------------------------------------------------------------------------------
// 3-way bf16 split of an fp32 value ("bf16x3", see conv2d.hip): x = x1 + x2 + x3, 8 + 8 + 8 significant bits
// ------------------------------------------------------------------------------------------------
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned f2bf_bits(float x) {
  unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf_bits2f(unsigned h) { return __uint_as_float(h << 16); }
// x -> 3 bf16 planes (bits)
__device__ __forceinline__ void split3(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = f2bf_bits(x);
  const float r1 = x - bf_bits2f(h1);
  h2 = f2bf_bits(r1);
  const float r2 = r1 - bf_bits2f(h2);
  h3 = f2bf_bits(r2);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 values -> three packed bf16 pairs (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ void split3_pair(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  const f32x2 v = {x, y};
  const bf16x2 h1 = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h1, f32x2);
  const bf16x2 h2 = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(h2, f32x2);
  const bf16x2 h3 = __builtin_convertvector(r2, bf16x2);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
  p3 = __builtin_bit_cast(unsigned, h3);
}

// buffer-descriptor addressing: a load at an out-of-range offset returns 0 (zero padding without a branch), a store there
// is dropped
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define BX_OOB 0x80000000u  // byte offset beyond any buffer this kernel addresses (sizes are checked < 2 GiB on the host)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bx_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
