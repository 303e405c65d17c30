// GroupNorm(num_groups=1) and activation kernels, nhwc.  Bandwidth-bound elementwise / reduction work:
// float4 per lane, per-sample statistics in fp64 (sum / sum of squares accumulated with fp64 atomics).
#include "common.h"
#include <stdlib.h>

typedef float gn_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gn_ld4(const float4* p, bool nt) {
  if (nt) {
    const gn_v4f t = __builtin_nontemporal_load((const gn_v4f*)p);
    return make_float4(t[0], t[1], t[2], t[3]);
  }
  return *p;
}
__device__ __forceinline__ void gn_st4(float4* p, const float4& v, bool nt) {
  if (nt) __builtin_nontemporal_store((gn_v4f){v.x, v.y, v.z, v.w}, (gn_v4f*)p);
  else *p = v;
}


__global__ void act_bwd_kernel(const float4* __restrict__ gy, const float4* __restrict__ y, float4* __restrict__ gp,
                               int act, long count4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
    const float4 g = gy[i], v = y[i];
    gp[i] = make_float4(g.x * act_grad_from_out(v.x, act), g.y * act_grad_from_out(v.y, act),
                        g.z * act_grad_from_out(v.z, act), g.w * act_grad_from_out(v.w, act));
  }
}
extern "C" int dis_act_bwd(const float* gy, const float* y, float* gpre, int act, long count, void* stream) {
  if (!gy || !y || !gpre) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  if (count % 4 != 0) return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(dis_ew_grid(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)gy, (const float4*)y, (float4*)gpre, act, count / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// The same on channel ranges of wider nhwc buffers (DispNetS: conv outputs live inside the concatenation buffers of the
// decoder): gy / y pixels are ldg / ldy floats apart, gpre is dense (npix, c).  act == DIS_ACT_NONE copies.
__global__ void act_bwd_ld_kernel(const float* __restrict__ gy, int ldg, const float* __restrict__ y, int ldy,
                                  float4* __restrict__ gp, int act, long npix, int c4) {
  const long total = npix * c4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / c4;
    const int q = (int)(i - px * c4) * 4;
    const float4 g = *(const float4*)(gy + px * ldg + q);
    if (act == DIS_ACT_NONE) {
      gp[i] = g;
    } else {
      const float4 v = *(const float4*)(y + px * ldy + q);
      gp[i] = make_float4(g.x * act_grad_from_out(v.x, act), g.y * act_grad_from_out(v.y, act),
                          g.z * act_grad_from_out(v.z, act), g.w * act_grad_from_out(v.w, act));
    }
  }
}
extern "C" int dis_act_bwd_ld(const float* gy, int ldg, const float* y, int ldy, float* gpre, int act, long npix, int c,
                              void* stream) {
  if (!gy || !gpre || (act != DIS_ACT_NONE && !y)) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || ldg < c || (act != DIS_ACT_NONE && ldy < c)) return DIS_ERR_BAD_SHAPE;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || ((uintptr_t)gy & 15) || ((uintptr_t)y & 15)) return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(act_bwd_ld_kernel, dim3(dis_ew_grid(npix * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     ldg, y ? y : gy, ldy, (float4*)gpre, act, npix, c / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// dis_act_bwd_ld that also leaves the bias gradient (column sums of gpre) behind: one pass over the gradient tensor instead of two.
// c / 4 divides 256: a thread keeps its 4 channels while it walks down the pixels; per-block fp32 partials, fp64 totals.
#define ABB_BLOCKS 1024
__global__ __launch_bounds__(256) void act_bwd_ld_bias_kernel(const float* __restrict__ gy, int ldg, const float* __restrict__ y,
                                                               int ldy, float4* __restrict__ gp, int act, long npix, int c4,
                                                               float* __restrict__ part) {
  __shared__ float red[1024];
  const int chunk = threadIdx.x % c4, rows = 256 / c4, row = threadIdx.x / c4, q = chunk * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long px = (long)blockIdx.x * rows + row; px < npix; px += (long)gridDim.x * rows) {
    // (gy and y are dead after this pass - the backward of the layer that consumed y has already run: non-temporal loads)
    float4 o = gn_ld4((const float4*)(gy + px * ldg + q), true);
    if (act != DIS_ACT_NONE) {
      const float4 v = gn_ld4((const float4*)(y + px * ldy + q), true);
      o.x *= act_grad_from_out(v.x, act), o.y *= act_grad_from_out(v.y, act);
      o.z *= act_grad_from_out(v.z, act), o.w *= act_grad_from_out(v.w, act);
    }
    gp[px * c4 + chunk] = o;
    s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
  }
  *(float4*)(red + threadIdx.x * 4) = s;
  __syncthreads();
  const int c = c4 * 4;
  for (int t = threadIdx.x; t < c; t += 256) {
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += red[(r * c4 + (t >> 2)) * 4 + (t & 3)];
    part[(long)blockIdx.x * c + t] = acc;
  }
}
__global__ __launch_bounds__(256) void act_bias_final_kernel(const float* __restrict__ part, int nblocks, int c,
                                                              float* __restrict__ out) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double s = 0.0;
  for (int k = lane; k < nblocks; k += 64) s += (double)part[(long)k * c + ch];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) out[ch] = (float)s;
}
extern "C" long dis_act_bwd_ld_bias_workspace(int c) { return c > 0 ? (long)ABB_BLOCKS * c : -1; }
extern "C" int dis_act_bwd_ld_bias(const float* gy, int ldg, const float* y, int ldy, float* gpre, int act, long npix, int c,
                                   float* bias_grad, float* workspace, void* stream) {
  if (!gy || !gpre || !bias_grad || !workspace || (act != DIS_ACT_NONE && !y)) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || ldg < c || (act != DIS_ACT_NONE && ldy < c)) return DIS_ERR_BAD_SHAPE;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || ((uintptr_t)gy & 15) || ((uintptr_t)y & 15) || c > 1024 || 256 % (c / 4))
    return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int rows = 256 / (c / 4);
  long nb = (npix + rows - 1) / rows;
  if (nb > ABB_BLOCKS) nb = ABB_BLOCKS;
  hipLaunchKernelGGL(act_bwd_ld_bias_kernel, dim3((unsigned)nb), dim3(256), 0, s, gy, ldg, y ? y : gy, ldy, (float4*)gpre, act,
                     npix, c / 4, workspace);
  hipLaunchKernelGGL(act_bias_final_kernel, dim3(dis_cdiv(c, 4)), dim3(256), 0, s, (const float*)workspace, (int)nb, c,
                     bias_grad);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// dst[pixel * ldd + j] = src[pixel * lds + j] for j < c, 0 for c <= j < c + czero: writes a tensor into a channel range
// of a wider nhwc buffer (the up-sampled disparity channel of the DispNetS concatenations and the zero lanes that pad
// them to a multiple of 4 channels) without an ATen in-place op on the buffer.
__global__ void copy_channels_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long npix,
                                     int c, int czero) {
  const int ct = c + czero;
  const long total = npix * ct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / ct;
    const int j = (int)(i - px * ct);
    dst[px * ldd + j] = j < c ? src[px * lds + j] : 0.f;
  }
}
// (1 channel + 3 zero lanes, 16-byte aligned: one float4 store per pixel)
__global__ void copy_channel_pad4_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd,
                                         long npix) {
  for (long px = blockIdx.x * (long)blockDim.x + threadIdx.x; px < npix; px += (long)gridDim.x * blockDim.x)
    *(float4*)(dst + px * ldd) = make_float4(src[px * lds], 0.f, 0.f, 0.f);
}
extern "C" int dis_copy_channels(const float* src, int lds, float* dst, int ldd, long npix, int c, int czero,
                                 void* stream) {
  if (!src || !dst) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || czero < 0 || lds < c || ldd < c + czero) return DIS_ERR_BAD_SHAPE;
  if (c == 1 && czero == 3 && !(ldd & 3) && !((uintptr_t)dst & 15))
    hipLaunchKernelGGL(copy_channel_pad4_kernel, dim3(dis_ew_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       dst, ldd, npix);
  else
    hipLaunchKernelGGL(copy_channels_kernel, dim3(dis_ew_grid(npix * (c + czero), 256)), dim3(256), 0,
                       (hipStream_t)stream, src, lds, dst, ldd, npix, c, czero);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

__global__ void add_act_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y,
                               int act, long count4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count4; i += (long)gridDim.x * blockDim.x) {
    const float4 p = a[i], q = b[i];
    y[i] = make_float4(act_apply(p.x + q.x, act), act_apply(p.y + q.y, act), act_apply(p.z + q.z, act),
                       act_apply(p.w + q.w, act));
  }
}
extern "C" int dis_add_act_fwd(const float* a, const float* b, float* y, int act, long count, void* stream) {
  if (!a || !b || !y) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  if (count % 4 != 0) return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(add_act_kernel, dim3(dis_ew_grid(count / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)a, (const float4*)b, (float4*)y, act, count / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float4* __restrict__ x, double* __restrict__ stats,
                                                       long per4) {
  __shared__ double sm[8];
  const int n = blockIdx.y;
  const float4* p = x + (long)n * per4;
  double s1 = 0.0, s2 = 0.0;
  // 4 float4s per thread and round: their 16 values are summed in fp32 (exact enough: 16 terms), the running sums in
  // fp64.  Few blocks per sample: the launch ends in one fp64 atomic pair per block on the sample's two addresses.
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per4; i += 4 * stride) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long j = i + k * stride;
      const float4 v = j < per4 ? p[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      t1 += (v.x + v.y) + (v.z + v.w);
      t2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s1 += (double)t1;
    s2 += (double)t2;
  }
  const double r1 = block_sum_d(s1, sm), r2 = block_sum_d(s2, sm);
  if (threadIdx.x == 0) {
    atomic_add_d(stats + 2 * n, r1);
    atomic_add_d(stats + 2 * n + 1, r2);
  }
}
extern "C" int dis_gn_stats(const float* x, double* stats, int n, long per_sample, void* stream) {
  if (!x || !stats) return DIS_ERR_NULL;
  if (n <= 0 || per_sample <= 0) return DIS_ERR_BAD_SHAPE;
  if (per_sample % 4 != 0) return DIS_ERR_UNSUPPORTED;
  int gx = dis_ew_grid(per_sample / 16, 256);  // >= 4 float4s per thread
  const int cap = n >= 16 ? 32 : (n >= 4 ? 64 : 256);  // ~512+ blocks in all
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, (const float4*)x, stats,
                     per_sample / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

#define GN_MAXC 128


// y = act(x*scale_c + shift_c (+ residual)),  scale_c = rstd*gamma_c, shift_c = beta_c - scale_c*mean
// Non-temporal loads / stores for the GroupNorm passes (bit 0: forward apply, 1 / 2: backward apply loads / store, 3: backward
// reduce loads).  These kernels stream 180 - 500 MB per launch; measured A/B on MI355X (DIS-MF bs=4): forward apply 2.86 -> 2.65
// ms/step, backward 7.18 -> 6.77 ms/step, 396.5 -> 399.9 frames/s with all four.  DIS_GN_NT=0 switches them off.
static int gn_nt_flags() {
  static const int f = getenv("DIS_GN_NT") ? atoi(getenv("DIS_GN_NT")) : 15;
  return f;
}
__global__ void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ res, float* __restrict__ y, long hw, int c, int act,
                                float eps, int nt) {
  __shared__ float sc[GN_MAXC], sh[GN_MAXC];
  const int n = blockIdx.y;
  if (threadIdx.x < c) {
    float mean, rstd;
    gn_moments(stats, n, (double)hw * c, eps, &mean, &rstd);
    const float s = rstd * gamma[threadIdx.x];
    sc[threadIdx.x] = s;
    sh[threadIdx.x] = beta[threadIdx.x] - s * mean;
  }
  __syncthreads();
  const int cg = c >> 2;
  const long per4 = hw * cg;
  const float4* xp = (const float4*)(x + (long)n * hw * c);
  const float4* rp = res ? (const float4*)(res + (long)n * hw * c) : nullptr;
  float4* yp = (float4*)(y + (long)n * hw * c);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per4; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg) * 4;
    const float4 v = gn_ld4(xp + i, nt & 1);
    float4 o = make_float4(v.x * sc[g] + sh[g], v.y * sc[g + 1] + sh[g + 1], v.z * sc[g + 2] + sh[g + 2],
                           v.w * sc[g + 3] + sh[g + 3]);
    if (rp) {
      const float4 r = rp[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    o = make_float4(act_apply(o.x, act), act_apply(o.y, act), act_apply(o.z, act), act_apply(o.w, act));
    gn_st4(yp + i, o, nt & 1);
  }
}
extern "C" int dis_gn_apply(const float* x, const double* stats, const float* gamma, const float* beta,
                            const float* residual, float* y, int n, long hw, int c, int act, float eps,
                            void* stream) {
  if (!x || !stats || !gamma || !beta || !y) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || c > GN_MAXC) return DIS_ERR_UNSUPPORTED;
  int gx = dis_ew_grid(hw * (c / 4), 256);
  if (gx > 512) gx = 512;
  const int nt = gn_nt_flags();
  hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, residual,
                     y, hw, c, act, eps, nt);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// backward pass 1: per-sample s1 = sum g*gamma, s2 = sum g*gamma*xhat ; per-channel dgamma, dbeta.
// Every block writes its partial sums to its own workspace slot (no atomics: 4096 blocks x 66 fp64 atomics on 96
// addresses used to cost more than the memory pass itself); the second pass and a small reducer add them up in a
// fixed order (deterministic).
#define GN_BWD_BLOCKS 64      // reduce blocks per sample when a launch covers many samples ...
#define GN_BWD_BLOCKS_MAX 512  // ... and when it covers one (a launch should bring >= 512 blocks)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                     const float* __restrict__ x, const double* __restrict__ stats,
                                     const float* __restrict__ gamma, double* __restrict__ red,
                                     double* __restrict__ gparam, long hw, int c, int act, float eps, int n0, int nt) {
  __shared__ float gam[GN_MAXC];
  __shared__ double sm[8];
  __shared__ float pg[256 * 4], pb[256 * 4];
  const int n = n0 + blockIdx.y;  // (this launch covers samples n0 .. n0 + gridDim.y - 1)
  float mean, rstd;
  gn_moments(stats, n, (double)hw * c, eps, &mean, &rstd);
  if (threadIdx.x < c) gam[threadIdx.x] = gamma[threadIdx.x];
  __syncthreads();
  const int cg = c >> 2;
  const long per4 = hw * cg;
  const float4* gp = (const float4*)(gy + (long)n * hw * c);
  const float4* yp = (const float4*)(y + (long)n * hw * c);
  const float4* xp = (const float4*)(x + (long)n * hw * c);
  // blockDim.x (256) is a multiple of cg, and the grid stride is a multiple of cg: each thread keeps one
  // channel group for its whole loop
  const int g = (int)(threadIdx.x % cg) * 4;
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  double s1 = 0.0, s2 = 0.0;
  const long stride = (long)gridDim.x * blockDim.x;
  auto body = [&](float4 gv, const float4 yv, const float4 xv) {
    if (act != DIS_ACT_NONE) {
      gv.x *= act_grad_from_out(yv.x, act); gv.y *= act_grad_from_out(yv.y, act);
      gv.z *= act_grad_from_out(yv.z, act); gv.w *= act_grad_from_out(yv.w, act);
    }
    const float xh[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
    const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dg[k] += ga[k] * xh[k];
      db[k] += ga[k];
      const float t = ga[k] * gam[g + k];
      t1 += t;
      t2 += t * xh[k];
    }
    s1 += (double)t1;
    s2 += (double)t2;
  };
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  // 4 independent loads per tensor in flight
  for (; i + 3 * stride < per4; i += 4 * stride) {
    float4 gv[4], yv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gv[u] = gn_ld4(gp + i + u * stride, nt & 8);
      xv[u] = gn_ld4(xp + i + u * stride, nt & 8);
      yv[u] = (act != DIS_ACT_NONE) ? gn_ld4(yp + i + u * stride, nt & 8) : gv[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) body(gv[u], yv[u], xv[u]);
  }
  for (; i < per4; i += stride) {
    const float4 gv = gp[i];
    body(gv, (act != DIS_ACT_NONE) ? yp[i] : gv, xp[i]);
  }
  const double r1 = block_sum_d(s1, sm), r2 = block_sum_d(s2, sm);
  const long slot = (long)n * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    red[2 * slot] = r1;
    red[2 * slot + 1] = r2;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pg[threadIdx.x * 4 + k] = dg[k];
    pb[threadIdx.x * 4 + k] = db[k];
  }
  __syncthreads();
  if (threadIdx.x < c) {
    const int grp = threadIdx.x >> 2, k = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = grp; t < 256; t += cg) {
      a += (double)pg[t * 4 + k];
      b += (double)pb[t * 4 + k];
    }
    gparam[slot * 2 * c + threadIdx.x] = a;
    gparam[slot * 2 * c + c + threadIdx.x] = b;
  }
}

// backward pass 2: gx = rstd*(g*gamma - s1/M - xhat*s2/M); gres = g
__global__ void gn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                    const float* __restrict__ x, const double* __restrict__ stats,
                                    const float* __restrict__ gamma, const double* __restrict__ red,
                                    float* __restrict__ gx, float* __restrict__ gres, long hw, int c, int act,
                                    float eps, int nred, int in_act, const double* __restrict__ gparam,
                                    float* __restrict__ gg, float* __restrict__ gb, int slots, int n0, int nt) {
  __shared__ float gam[GN_MAXC];
  // Samples and elements are walked in the REVERSE of pass 1's order: what pass 1 read last is what the Infinity Cache
  // still holds, so this pass starts on cached data instead of evicting it first.
  const int n = n0 + gridDim.y - 1 - blockIdx.y;
  // dgamma / dbeta (folded in here to save a launch; slots > 0 only in the LAST group's launch, when every partial exists):
  // wave w of the first 2c/4 blocks of one sample sums the `slots` per-block partials of one parameter in a fixed order
  if (slots > 0 && blockIdx.y == 0 && (int)blockIdx.x * 4 < 2 * c) {
    const int lane = threadIdx.x & 63;
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < 2 * c; j += gridDim.x * 4) {
      double s = 0.0;
      for (int k = lane; k < slots; k += 64) s += gparam[(long)k * 2 * c + j];
      s = wave_sum_d(s);
      if (lane == 0) {
        if (j < c) gg[j] = (float)s;
        else gb[j - c] = (float)s;
      }
    }
  }
  float mean, rstd;
  const double m = (double)hw * c;
  gn_moments(stats, n, m, eps, &mean, &rstd);
  __shared__ double tot[2];
  if (threadIdx.x < 64) {  // wave 0 adds the per-block partials of pass 1 (same order in every block)
    double t1 = 0.0, t2 = 0.0;
    for (int b = threadIdx.x; b < nred; b += 64) {
      t1 += red[2 * ((long)n * nred + b)];
      t2 += red[2 * ((long)n * nred + b) + 1];
    }
    t1 = wave_sum_d(t1);
    t2 = wave_sum_d(t2);
    if (threadIdx.x == 0) {
      tot[0] = t1;
      tot[1] = t2;
    }
  }
  if (threadIdx.x < c) gam[threadIdx.x] = gamma[threadIdx.x];
  __syncthreads();
  const float a1 = (float)(tot[0] / m), a2 = (float)(tot[1] / m);
  const int cg = c >> 2;
  const long per4 = hw * cg;
  const float4* gp = (const float4*)(gy + (long)n * hw * c);
  const float4* yp = (const float4*)(y + (long)n * hw * c);
  const float4* xp = (const float4*)(x + (long)n * hw * c);
  float4* op = (float4*)(gx + (long)n * hw * c);
  float4* rp = gres ? (float4*)(gres + (long)n * hw * c) : nullptr;
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < per4; j += (long)gridDim.x * blockDim.x) {
    const long i = per4 - 1 - j;
    const int g = (int)(i % cg) * 4;
    float4 gv = gn_ld4(gp + i, nt & 2);
    if (act != DIS_ACT_NONE) {
      const float4 yv = gn_ld4(yp + i, nt & 2);
      gv.x *= act_grad_from_out(yv.x, act); gv.y *= act_grad_from_out(yv.y, act);
      gv.z *= act_grad_from_out(yv.z, act); gv.w *= act_grad_from_out(yv.w, act);
    }
    if (rp) rp[i] = gv;
    const float4 xv = gn_ld4(xp + i, nt & 2);
    float4 o;
    o.x = rstd * (gv.x * gam[g] - a1 - ((xv.x - mean) * rstd) * a2);
    o.y = rstd * (gv.y * gam[g + 1] - a1 - ((xv.y - mean) * rstd) * a2);
    o.z = rstd * (gv.z * gam[g + 2] - a1 - ((xv.z - mean) * rstd) * a2);
    o.w = rstd * (gv.w * gam[g + 3] - a1 - ((xv.w - mean) * rstd) * a2);
    if (in_act != DIS_ACT_NONE) {  // x is the activation OUTPUT of the producing conv: emit its pre-activation gradient
      o.x *= act_grad_from_out(xv.x, in_act); o.y *= act_grad_from_out(xv.y, in_act);
      o.z *= act_grad_from_out(xv.z, in_act); o.w *= act_grad_from_out(xv.w, in_act);
    }
    gn_st4(op + i, o, nt & 4);
  }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm backward from per-(sample, channel) sums (round 3).  With A_c = sum_p g, B_c = sum_p g x over a sample's pixels
// (left by the epilogue of the kernel that produced g: dis_conv2d_dgrad_bf16x3_gnsums) everything else follows:
//   dbeta_c = A_c, dgamma_c = rstd (B_c - mean A_c), a1 = sum_c gamma_c A_c / M, a2 = sum_c gamma_c dgamma_c / M,
//   gx = act'(x) rstd (g gamma_c - a1 - xhat a2) = act'(x) (g k1_c + x kx + k0),  k1_c = rstd gamma_c, kx = -rstd^2 a2,
//   k0 = rstd^2 a2 mean - rstd a1
// so the backward is ONE elementwise pass (read g, x; write gx) instead of a reduce pass and an apply pass.
// ------------------------------------------------------------------------------------------------
// grid n: block b writes coef[b][0..c) = k1_c, coef[b][c] = kx, coef[b][c + 1] = k0 and the sample's dgamma_c / dbeta_c (fp64) behind
// the coefficients of all samples (pg); the apply kernel's first block adds those over the samples.  Fixed orders: deterministic.
#define GN_COEF_T 1024
// fin != nullptr (dis_gn_bwd_coef): the block that arrives LAST at the counter *fin (zeroed by the caller) adds the samples' dgamma /
// dbeta parts in sample order and writes gg / gb - the job of gn_apply_coef_kernel's first block when the elementwise pass exists.
__global__ __launch_bounds__(GN_COEF_T) void gn_coef_kernel(const double* __restrict__ ab, int slots, const double* __restrict__ stats,
                                                            const float* __restrict__ gamma, float* __restrict__ coef,
                                                            double* __restrict__ pg, long hw, int c, float eps,
                                                            unsigned* __restrict__ fin = nullptr, float* __restrict__ gg = nullptr,
                                                            float* __restrict__ gb = nullptr) {
  __shared__ double sA[GN_MAXC], sB[GN_MAXC], part[GN_COEF_T];
  __shared__ unsigned s_last;
  const int b = blockIdx.x, t = threadIdx.x;
  const double m = (double)hw * c;
  {  // A_c, B_c of this sample: 16 threads per value, each a sixteenth of the slots with all of its loads in flight (256 slots:
     // one round; with 4 threads per value and 8 loads per round the launch was eight dependent round trips, 9 us, 30 per step)
    const int j = t & 63, q = t >> 6;  // (2 c <= 64 values)
    double v = 0.0;
    if (j < 2 * c) {
      const double* p = ab + (long)b * slots * (2 * c) + j;
      int s_ = q;
      for (; s_ + 15 * 16 < slots; s_ += 256) {
        double u[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) u[k] = p[(long)(s_ + 16 * k) * (2 * c)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v += u[k];
      }
      for (; s_ < slots; s_ += 16) v += p[(long)s_ * (2 * c)];
    }
    part[t] = v;
    __syncthreads();
    if (t < 2 * c) {
      double tot = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < 16; ++q2) tot += part[t + 64 * q2];
      if (t < c) sA[t] = tot;
      else sB[t - c] = tot;
    }
    __syncthreads();
  }
  float mean, rstd;
  gn_moments(stats, b, m, eps, &mean, &rstd);
  if (t < c) {
    coef[(long)b * (c + 2) + t] = rstd * gamma[t];
    pg[(long)b * 2 * c + t] = (double)rstd * (sB[t] - (double)mean * sA[t]);
    pg[(long)b * 2 * c + c + t] = sA[t];
  }
  if (t < 64) {   // (c <= 64: one wave, fixed-order DPP sums instead of a serial loop over the channels in one thread)
    double p1 = 0.0, p2 = 0.0;
    if (t < c) {
      const double gk = (double)gamma[t];
      const double dg = (double)rstd * (sB[t] - (double)mean * sA[t]);
      p1 = gk * sA[t];
      p2 = gk * dg;
    }
    const double s1 = wave_sum_d(p1), s2 = wave_sum_d(p2);
    const float a1 = (float)(s1 / m), a2 = (float)(s2 / m);
    if (t == 0) {
      coef[(long)b * (c + 2) + c] = -(rstd * rstd) * a2;
      coef[(long)b * (c + 2) + c + 1] = (rstd * rstd) * a2 * mean - rstd * a1;
    }
  }
  if (fin) {
    // this block's pg[] entries are visible device-wide before it counts itself in: the barrier completes every thread's stores
    // (to this XCD's L2), ONE thread's release fence writes the L2 back - a fence in every thread cost 15 us per launch
    __syncthreads();
    if (t == 0) {
      // release (agent scope, cumulative over the barrier above: every thread's pg[] stores) + acquire (the other blocks' entries,
      // written through other XCDs' L2s, are read from memory by the threads behind the next barrier) spelled out on the atomic
      s_last = __hip_atomic_fetch_add(fin, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
      if (t < 2 * c) {
        double v = 0.0;
        for (int sn = 0; sn < (int)gridDim.x; ++sn) v += __builtin_nontemporal_load(pg + (long)sn * 2 * c + t);
        if (t < c) gg[t] = (float)v;
        else gb[t - c] = (float)v;
      }
      if (t == 0) *fin = 0u;   // (re-armed: a captured graph replays this launch on the same counter)
    }
  }
}
__global__ void gn_apply_coef_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ coef,
                                     float* __restrict__ gx, long hw, int c, int in_act, int nt, const double* __restrict__ pg,
                                     float* __restrict__ gg, float* __restrict__ gb) {
  __shared__ float k1[GN_MAXC], kk[2];
  const int n = blockIdx.y;
  if (pg && blockIdx.x == 0 && n == 0 && (int)threadIdx.x < 2 * c) {  // dgamma / dbeta: the samples' parts, in order
    double v = 0.0;
    for (int sn = 0; sn < (int)gridDim.y; ++sn) v += pg[(long)sn * 2 * c + threadIdx.x];
    if ((int)threadIdx.x < c) gg[threadIdx.x] = (float)v;
    else gb[threadIdx.x - c] = (float)v;
  }
  if (threadIdx.x < c) k1[threadIdx.x] = coef[(long)n * (c + 2) + threadIdx.x];
  if (threadIdx.x < 2) kk[threadIdx.x] = coef[(long)n * (c + 2) + c + threadIdx.x];
  __syncthreads();
  const float kx = kk[0], k0 = kk[1];
  const int cg = c >> 2;
  const long per4 = hw * cg;
  const float4* gp = (const float4*)(g + (long)n * hw * c);
  const float4* xp = (const float4*)(x + (long)n * hw * c);
  float4* op = (float4*)(gx + (long)n * hw * c);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per4; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cg) * 4;
    const float4 gv = gn_ld4(gp + i, nt & 2), xv = gn_ld4(xp + i, nt & 2);
    float4 o;
    o.x = __builtin_fmaf(gv.x, k1[ch], __builtin_fmaf(xv.x, kx, k0));
    o.y = __builtin_fmaf(gv.y, k1[ch + 1], __builtin_fmaf(xv.y, kx, k0));
    o.z = __builtin_fmaf(gv.z, k1[ch + 2], __builtin_fmaf(xv.z, kx, k0));
    o.w = __builtin_fmaf(gv.w, k1[ch + 3], __builtin_fmaf(xv.w, kx, k0));
    if (in_act != DIS_ACT_NONE) {
      o.x *= act_grad_from_out(xv.x, in_act); o.y *= act_grad_from_out(xv.y, in_act);
      o.z *= act_grad_from_out(xv.z, in_act); o.w *= act_grad_from_out(xv.w, in_act);
    }
    gn_st4(op + i, o, nt & 4);
  }
}
/* GroupNorm(1 group) backward from the sums dis_conv2d_dgrad_bf16x3_gnsums left: g (n, hw, c) the gradient wrt the GroupNorm's
 * output, x its input, stats (n, 2) the forward statistics, ab (n, slots, 2, c) doubles.  Writes gx (the gradient wrt x, times
 * act'(x) when x is the output of the activation in_act of the producing conv), grad_gamma, grad_beta (c).
 * coef: workspace of n * (c + 2) floats followed by n * 2 c doubles (n * (c + 2) + 4 n c + 2 floats in all: 8-byte alignment). */
extern "C" int dis_gn_bwd_from_sums(const float* g, const float* x, const double* stats, const float* gamma, const double* ab,
                                    int slots, float* gx, float* grad_gamma, float* grad_beta, float* coef, int n, long hw, int c,
                                    float eps, int in_act, void* stream) {
  if (!g || !x || !stats || !gamma || !ab || !gx || !grad_gamma || !grad_beta || !coef) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0 || slots <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || 2 * c > 64) return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  double* pg = (double*)(((uintptr_t)(coef + (long)n * (c + 2)) + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(gn_coef_kernel, dim3(n), dim3(GN_COEF_T), 0, s, ab, slots, stats, gamma, coef, pg, hw, c, eps);
  int gxg = dis_ew_grid(hw * (c / 4), 256);
  if (gxg > 512) gxg = 512;
  hipLaunchKernelGGL(gn_apply_coef_kernel, dim3(gxg, n), dim3(256), 0, s, g, x, (const float*)coef, gx, hw, c, in_act,
                     gn_nt_flags(), (const double*)pg, grad_gamma, grad_beta);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// Channel sums for the from-sums backward when NO convolution produced the gradient (round 5): the GroupNorm's output gradient comes
// from a join / a resize / a feature warp, so nobody left A_c = sum g, B_c = sum g x.  One pass over gy (and y: g = gy act'(y), stored
// as the residual gradient when the GroupNorm has a residual; and x) writes them in the slot layout of the convolution epilogues -
// block (s, n) -> ab[n][s][2][c] - after which the backward is dis_gn_bwd_coef + an elementwise pass that the producer of x applies on
// load (dis_conv2d_dgrad_f16x2_gnb): the reduce + apply pair of dis_gn_apply_bwd (read gy, y, x twice, write gx, gres) becomes
// read gy, y, x; write gres.  A thread keeps its 4 channels over its items (the stride is a multiple of c / 4); fixed orders.
#define GN_RS_T 256
__global__ __launch_bounds__(GN_RS_T) void gn_res_sums_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                              const float* __restrict__ x, float* __restrict__ gres,
                                                              double* __restrict__ ab, long hw, int c, int act, int nt) {
  __shared__ float sa[GN_RS_T][4], sb[GN_RS_T][4];
  const int n = blockIdx.y, t = threadIdx.x;
  const int cg = c >> 2;
  const long per4 = hw * cg;
  const float4* gp = (const float4*)(gy + (long)n * hw * c);
  const float4* yp = y ? (const float4*)(y + (long)n * hw * c) : nullptr;
  const float4* xp = (const float4*)(x + (long)n * hw * c);
  float4* rp = gres ? (float4*)(gres + (long)n * hw * c) : nullptr;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)GN_RS_T + t; i < per4; i += (long)gridDim.x * GN_RS_T) {
    float4 g = gn_ld4(gp + i, nt & 2);
    if (yp) {
      const float4 yv = gn_ld4(yp + i, nt & 2);
      g.x *= act_grad_from_out(yv.x, act); g.y *= act_grad_from_out(yv.y, act);
      g.z *= act_grad_from_out(yv.z, act); g.w *= act_grad_from_out(yv.w, act);
    }
    if (rp) rp[i] = g;   // (read again by the consumers of the residual gradient and by the on-load pass: a plain store)
    const float4 xv = gn_ld4(xp + i, nt & 2);
    a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
    b[0] = __builtin_fmaf(g.x, xv.x, b[0]); b[1] = __builtin_fmaf(g.y, xv.y, b[1]);
    b[2] = __builtin_fmaf(g.z, xv.z, b[2]); b[3] = __builtin_fmaf(g.w, xv.w, b[3]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sa[t][k] = a[k], sb[t][k] = b[k];
  __syncthreads();
  if (t < 2 * c) {   // thread j < c: A_j, c <= j < 2c: B_(j - c): the GN_RS_T / cg threads of channel group j / 4, in thread order
    const int ch = t < c ? t : t - c, grp = ch >> 2, k = ch & 3;
    double v = 0.0;
    for (int u = grp; u < GN_RS_T; u += cg) v += (double)(t < c ? sa[u][k] : sb[u][k]);
    ab[((long)n * gridDim.x + blockIdx.x) * (2 * c) + t] = v;
  }
}
/* gy (n, hw, c): gradient wrt the output of a GroupNorm(1 group) with input x; y / act: the output and its activation when the output
 * went through one (g = gy act'(y); NULL / DIS_ACT_NONE: g = gy); gres: g stored (the residual gradient; may be NULL).
 * ab_out (n, slots, 2, c) doubles, every slot written: slots = dis_conv2d_gnsums_slots() when the result feeds dis_gn_bwd_coef.
 * c % 4 == 0, 2 c <= 64, 256 % (c / 4) == 0. */
extern "C" int dis_gn_bwd_res_sums(const float* gy, const float* y, const float* x, float* gres, double* ab_out, int slots, int n,
                                   long hw, int c, int act, void* stream) {
  if (!gy || !x || !ab_out) return DIS_ERR_NULL;
  if (act != DIS_ACT_NONE && !y) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0 || slots <= 0 || n > 65535) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || 2 * c > 64 || GN_RS_T % (c / 4) != 0) return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gn_res_sums_kernel, dim3(slots, n), dim3(GN_RS_T), 0, (hipStream_t)stream, gy, act != DIS_ACT_NONE ? y : nullptr, x,
                     gres, ab_out, hw, c, act, gn_nt_flags());
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

/* The two halves of dis_gn_bwd_from_sums as separate entry points (round 5): the consumer of gx is usually the 3x3 convolution in
 * front of the GroupNorm, whose input-gradient launch can apply the elementwise pass while it stages its operand
 * (dis_conv2d_dgrad_f16x2_gnb) - the pass over g, x and gx then does not exist.
 *   dis_gn_bwd_coef: sums -> coef (n, c + 2) floats [k1_c..., kx, k0 per sample] followed (8-byte aligned) by n * 2 c doubles, and
 *     grad_gamma / grad_beta (c), finished by the last block to arrive at *counter (one unsigned, ZERO on entry, zero again on exit).
 *   dis_gn_bwd_apply_coef: gx = act'(x) (g k1_c + x kx + k0), the elementwise pass alone (the general fallback). */
extern "C" int dis_gn_bwd_coef(const double* stats, const float* gamma, const double* ab, int slots, float* coef, float* grad_gamma,
                               float* grad_beta, unsigned* counter, int n, long hw, int c, float eps, void* stream) {
  if (!stats || !gamma || !ab || !coef || !grad_gamma || !grad_beta || !counter) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0 || slots <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || 2 * c > 64) return DIS_ERR_UNSUPPORTED;
  double* pg = (double*)(((uintptr_t)(coef + (long)n * (c + 2)) + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(gn_coef_kernel, dim3(n), dim3(GN_COEF_T), 0, (hipStream_t)stream, ab, slots, stats, gamma, coef, pg, hw, c, eps,
                     counter, grad_gamma, grad_beta);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_gn_bwd_apply_coef(const float* g, const float* x, const float* coef, float* gx, int n, long hw, int c, int in_act,
                                     void* stream) {
  if (!g || !x || !coef || !gx) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || 2 * c > 64) return DIS_ERR_UNSUPPORTED;
  int gxg = dis_ew_grid(hw * (c / 4), 256);
  if (gxg > 512) gxg = 512;
  hipLaunchKernelGGL(gn_apply_coef_kernel, dim3(gxg, n), dim3(256), 0, (hipStream_t)stream, g, x, coef, gx, hw, c, in_act,
                     gn_nt_flags(), (const double*)nullptr, (float*)nullptr, (float*)nullptr);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" long dis_gn_bwd_workspace(int n, int c) { return (n > 0 && c > 0) ? (long)n * GN_BWD_BLOCKS_MAX * (2 + 2 * c) : -1; }

// Samples per reduce / apply launch pair.  Measured and rejected (round 2): running the two passes over sample GROUPS small
// enough for the 256 MiB Infinity Cache (reduce group, apply group, next group ...), so that the second pass would find what
// the first just read, made the backward SLOWER (7.4 -> 9.7 ms per DIS-MF step: 2 - 16x the launches, grids of 64 - 512
// blocks, no measurable cache benefit).  One group = the whole batch.
static int gn_bwd_group(int n, long per_sample_elems, int tensors) {
  (void)per_sample_elems;
  (void)tensors;
  return n;
}

extern "C" int dis_gn_apply_bwd(const float* gy, const float* y, const float* x, const double* stats,
                                const float* gamma, float* gx, float* gres, float* grad_gamma, float* grad_beta,
                                double* red, double* gparam_acc, int n, long hw, int c, int act, float eps,
                                int in_act, void* stream) {
  if (!gy || !x || !stats || !gamma || !gx || !grad_gamma || !grad_beta || !red || !gparam_acc) return DIS_ERR_NULL;
  if (act != DIS_ACT_NONE && !y) return DIS_ERR_NULL;
  if (n <= 0 || hw <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || c > GN_MAXC || 256 % (c / 4) != 0) return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int gxg = dis_ew_grid(hw * (c / 4), 256);
  if (gxg > 256) gxg = 256;
  const int grp = gn_bwd_group(n, hw * c, act != DIS_ACT_NONE ? 3 : 2);
  int nred = (GN_BWD_BLOCKS_MAX + grp - 1) / grp;  // blocks per sample: ~512 blocks per launch
  nred = (nred + 7) / 8 * 8;
  if (nred < GN_BWD_BLOCKS) nred = GN_BWD_BLOCKS;
  if (nred > GN_BWD_BLOCKS_MAX) nred = GN_BWD_BLOCKS_MAX;
  for (int n0 = 0; n0 < n; n0 += grp) {
    const int ng = n0 + grp <= n ? grp : n - n0;
    const bool last = n0 + ng == n;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nred, ng), dim3(256), 0, s, gy, y ? y : gy, x, stats, gamma, red,
                       gparam_acc, hw, c, act, eps, n0, gn_nt_flags());
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gxg * 2, ng), dim3(256), 0, s, gy, y ? y : gy, x, stats, gamma,
                       (const double*)red, gx, gres, hw, c, act, eps, nred, in_act, (const double*)gparam_acc,
                       grad_gamma, grad_beta, last ? n * nred : 0, n0, gn_nt_flags());
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
