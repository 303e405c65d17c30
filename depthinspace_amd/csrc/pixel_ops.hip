// Per-pixel operators of the DIS training step: LCN, photometric census window, pattern projection,
// scalar reductions, Sobel smoothness, disparity->depth, flow-consistency (geometric) loss.
// All are HBM-bandwidth-bound stencil/gather kernels: coalesced row-major access, LDS halo tiles for the
// windowed ones, wavefront-shuffle + fp64-atomic reductions for the scalar losses.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// LCN  (reference model/networks.py:663-689)
// ------------------------------------------------------------------------------------------------
#define LCN_TX 32
#define LCN_TY 8
#define LCN_MAXR 7

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // ReflectionPad2d (no edge repeat); valid for |overshoot| < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__global__ __launch_bounds__(LCN_TX* LCN_TY) void lcn_fwd_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ out_lcn,
                                                                    float* __restrict__ out_std, int h, int w,
                                                                    int radius, float eps) {
  __shared__ float tile[(LCN_TY + 2 * LCN_MAXR) * (LCN_TX + 2 * LCN_MAXR)];
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * LCN_TX, y0 = blockIdx.y * LCN_TY;
  const int tw = LCN_TX + 2 * radius, th = LCN_TY + 2 * radius;
  const float* img = x + (long)n * h * w;
  for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
    int ty = i / tw, tx = i - ty * tw;
    int gy = reflect_idx(y0 + ty - radius, h), gx = reflect_idx(x0 + tx - radius, w);
    gy = min(max(gy, 0), h - 1);
    gx = min(max(gx, 0), w - 1);
    tile[i] = img[(long)gy * w + gx];
  }
  __syncthreads();
  const int lx = threadIdx.x % LCN_TX, ly = threadIdx.x / LCN_TX;
  const int px = x0 + lx, py = y0 + ly;
  if (px >= w || py >= h) return;
  // fp64 box sums: the reference's fp32 conv result is reproduced to within its own rounding error
  double s = 0.0, s2 = 0.0;
  const int k = 2 * radius + 1;
  for (int dy = 0; dy < k; ++dy) {
    const float* row = tile + (ly + dy) * tw + lx;
    for (int dx = 0; dx < k; ++dx) {
      float v = row[dx];
      s += (double)v;
      s2 += (double)v * (double)v;
    }
  }
  const float kk = (float)(k * k);
  float box = (float)s, box2 = (float)s2;
  float avg = box / kk;
  float var = box2 / kk - avg * avg + 1e-6f;
  var = var < 0.f ? 0.f : var;
  float sd = sqrtf(var) + eps;
  float c = tile[(ly + radius) * tw + lx + radius];
  long o = (long)n * h * w + (long)py * w + px;
  out_lcn[o] = (c - avg) / sd;
  out_std[o] = sd;
}

extern "C" int dis_lcn_fwd(const float* x, float* out_lcn, float* out_std, int n, int h, int w, int radius,
                           float eps, void* stream) {
  if (!x || !out_lcn || !out_std) return DIS_ERR_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (radius < 0 || radius > LCN_MAXR || radius >= h || radius >= w) return DIS_ERR_UNSUPPORTED;
  dim3 grid(dis_cdiv(w, LCN_TX), dis_cdiv(h, LCN_TY), n);
  hipLaunchKernelGGL(lcn_fwd_kernel, grid, dim3(LCN_TX * LCN_TY), 0, (hipStream_t)stream, x, out_lcn, out_std, h,
                     w, radius, eps);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// photometric window loss (reference model/ext_functions.py:115-183)
// ------------------------------------------------------------------------------------------------
#define PH_TX 32
#define PH_TY 8
#define PH_MAXP 7

// soft sign of the census transform, h(d) = 0.5 (1 + d / sqrt(d^2 + eps)), and its derivative 0.5 eps (d^2 + eps)^-3/2.
// Both go through ONE v_rsq_f32 (1 ulp; q >= eps > 0): the kernels are bound by exactly this arithmetic (81 taps x 2
// soft signs per pixel), and an IEEE sqrt + an IEEE division per soft sign cost ~4x as many instructions.
__device__ __forceinline__ float census_rsq(float d, float eps) { return __builtin_amdgcn_rsqf(d * d + eps); }
// ... with the argument as ONE fused multiply-add (the multi-estimate kernels: tolerance-checked, every VALU slot counts)
__device__ __forceinline__ float census_rsq_fma(float d, float eps) { return __builtin_amdgcn_rsqf(__builtin_fmaf(d, d, eps)); }
// sign(x) in {-1, 0, 1} as two VALU instructions (|x| >= 1e-30 or 0: differences of O(1) values)
__device__ __forceinline__ float census_sign(float x) { return __builtin_amdgcn_fmed3f(x * 1e30f, -1.f, 1.f); }
// h(a) - h(b) given the two reciprocal roots (the 0.5 (1 + .) parts cancel)
__device__ __forceinline__ float census_hdiff(float a, float ra, float b, float rb) { return 0.5f * (a * ra - b * rb); }
__device__ __forceinline__ float census_dh_r(float r, float eps) { return (0.5f * eps) * (r * r * r); }
// the reference's own evaluation order (IEEE sqrt and division): used only to decide the SIGN of h(a) - h(b) when the
// fast difference is within rounding of zero, so that |.|' takes the same branch as the reference's autograd
__device__ __forceinline__ float census_h_exact(float d, float eps) { return 0.5f * (1.f + d / sqrtf(d * d + eps)); }

template <int TYPE>
__global__ __launch_bounds__(PH_TX* PH_TY) void photometric_fwd_kernel(const float* __restrict__ es,
                                                                         const float* __restrict__ ta,
                                                                         float* __restrict__ out, int c, int h,
                                                                         int w, int block, float eps) {
  __shared__ float te[(PH_TY + 2 * PH_MAXP) * (PH_TX + 2 * PH_MAXP)];
  __shared__ float tt[(PH_TY + 2 * PH_MAXP) * (PH_TX + 2 * PH_MAXP)];
  const int p = block / 2;
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * PH_TX, y0 = blockIdx.y * PH_TY;
  const int tw = PH_TX + 2 * p, th = PH_TY + 2 * p;
  const int lx = threadIdx.x % PH_TX, ly = threadIdx.x / PH_TX;
  const int px = x0 + lx, py = y0 + ly;
  float acc = 0.f;
  for (int ch = 0; ch < c; ++ch) {
    const float* ie = es + ((long)n * c + ch) * h * w;
    const float* it = ta + ((long)n * c + ch) * h * w;
    __syncthreads();
    for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
      int ty = i / tw, tx = i - ty * tw;
      int gy = min(max(y0 + ty - p, 0), h - 1), gx = min(max(x0 + tx - p, 0), w - 1);  // replicate pad
      te[i] = ie[(long)gy * w + gx];
      tt[i] = it[(long)gy * w + gx];
    }
    __syncthreads();
    const float ec = te[(ly + p) * tw + lx + p], tc = tt[(ly + p) * tw + lx + p];
    for (int dy = 0; dy < block; ++dy) {
      for (int dx = 0; dx < block; ++dx) {
        float e = te[(ly + dy) * tw + lx + dx], t = tt[(ly + dy) * tw + lx + dx];
        float r;
        if (TYPE == 0) {
          r = (e - t) * (e - t);
        } else if (TYPE == 1) {
          r = fabsf(e - t);
        } else {
          const float de = e - ec, dt = t - tc;
          const float diff = census_hdiff(de, census_rsq(de, eps), dt, census_rsq(dt, eps));
          r = (TYPE == 2) ? diff * diff : fabsf(diff);
        }
        acc += r;
      }
    }
  }
  if (px < w && py < h) out[(long)n * h * w + (long)py * w + px] = acc / (float)(block * block);
}

// number of window offsets d in [-p,p] with clamp(pc + d, 0, size-1) == kc
__device__ __forceinline__ int clamp_mult(int pc, int kc, int p, int size) {
  if (kc > 0 && kc < size - 1) return (abs(kc - pc) <= p) ? 1 : 0;
  int m = 0;
  if (kc == 0) {
    int hi = -pc;  // d <= -pc
    if (hi >= -p) m += min(hi, p) - (-p) + 1;
  }
  if (kc == size - 1) {
    int lo = size - 1 - pc;  // d >= lo
    if (lo <= p) m += p - max(lo, -p) + 1;
  }
  if (size == 1) m = 2 * p + 1;
  return m;
}

// Gather formulation of the backward pass (no atomics): pixel k collects
//   (a) its role as a NEIGHBOUR of every window centre p around it (with replicate-pad multiplicity), and
//   (b) its role as the CENTRE of its own window (census types only).
template <int TYPE>
__global__ __launch_bounds__(PH_TX* PH_TY) void photometric_bwd_kernel(const float* __restrict__ es,
                                                                         const float* __restrict__ ta,
                                                                         const float* __restrict__ gout,
                                                                         float* __restrict__ ges, int c, int h,
                                                                         int w, int block, float eps) {
  __shared__ float te[(PH_TY + 2 * PH_MAXP) * (PH_TX + 2 * PH_MAXP)];
  __shared__ float tt[(PH_TY + 2 * PH_MAXP) * (PH_TX + 2 * PH_MAXP)];
  __shared__ float tg[(PH_TY + 2 * PH_MAXP) * (PH_TX + 2 * PH_MAXP)];
  const int p = block / 2;
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * PH_TX, y0 = blockIdx.y * PH_TY;
  const int tw = PH_TX + 2 * p, th = PH_TY + 2 * p;
  const int lx = threadIdx.x % PH_TX, ly = threadIdx.x / PH_TX;
  const int px = x0 + lx, py = y0 + ly;
  const float inv = 1.f / (float)(block * block);
  const float* ig = gout + (long)n * h * w;
  for (int ch = 0; ch < c; ++ch) {
    const float* ie = es + ((long)n * c + ch) * h * w;
    const float* it = ta + ((long)n * c + ch) * h * w;
    __syncthreads();
    for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
      int ty = i / tw, tx = i - ty * tw;
      int ry = y0 + ty - p, rx = x0 + tx - p;
      int gy = min(max(ry, 0), h - 1), gx = min(max(rx, 0), w - 1);
      te[i] = ie[(long)gy * w + gx];
      tt[i] = it[(long)gy * w + gx];
      // window centres outside the image do not exist: zero gradient there
      tg[i] = (ry >= 0 && ry < h && rx >= 0 && rx < w) ? ig[(long)gy * w + gx] * inv : 0.f;
    }
    __syncthreads();
    if (px < w && py < h) {
      const float ek = te[(ly + p) * tw + lx + p], tk = tt[(ly + p) * tw + lx + p];
      const float gk = tg[(ly + p) * tw + lx + p];
      float acc = 0.f;
      const bool interior = (px > 0 && px < w - 1 && py > 0 && py < h - 1);
      for (int dy = 0; dy < block; ++dy) {
        for (int dx = 0; dx < block; ++dx) {
          const int li = (ly + dy) * tw + lx + dx;
          const float e = te[li], t = tt[li], g = tg[li];
          // (a) window centre at p = k + (dy-p, dx-p); te/tt hold the clamped image but tg is 0 outside
          float mult = 1.f;
          if (!interior) {
            int cy = py + dy - p, cx = px + dx - p;
            mult = (float)(clamp_mult(cy, py, p, h) * clamp_mult(cx, px, p, w));
          }
          if (TYPE == 0) {
            acc += mult * g * 2.f * (ek - tk);
          } else if (TYPE == 1) {
            float d = ek - tk;
            acc += mult * g * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
          } else {
            // (a) as neighbour of centre p: des = e_k - e_p, dta = t_k - t_p
            // (b) as centre of its own window with neighbour q = clamp(k + off): des' = e_q - e_k = -des, dta' = -dta.
            //     h(-d) = 1 - h(d) and h' is even, so diff' = -diff and term (b) = +g_k * s * h'(des): both roles share
            //     one evaluation of the two soft signs (half the sqrt/div work of evaluating them separately).
            float des = ek - e, dta = tk - t;
            const float re = census_rsq(des, eps);
            float diff = census_hdiff(des, re, dta, census_rsq(dta, eps));
            if (TYPE == 3 && fabsf(diff) < 1e-5f && (des != 0.f || dta != 0.f))  // (rare: ~1e-5 of the taps)
              diff = census_h_exact(des, eps) - census_h_exact(dta, eps);
            float s = (TYPE == 2) ? 2.f * diff : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
            acc += (mult * g + gk) * (s * census_dh_r(re, eps));
          }
        }
      }
      ges[((long)n * c + ch) * h * w + (long)py * w + px] = acc;
    }
  }
}

extern "C" int dis_photometric_fwd(const float* es, const float* ta, float* out, int n, int c, int h, int w,
                                   int block, int type, float eps, void* stream) {
  if (!es || !ta || !out) return DIS_ERR_NULL;
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (block < 1 || (block & 1) == 0 || block / 2 > PH_MAXP || type < 0 || type > 3) return DIS_ERR_UNSUPPORTED;
  dim3 grid(dis_cdiv(w, PH_TX), dis_cdiv(h, PH_TY), n), blk(PH_TX * PH_TY);
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
    case 0: hipLaunchKernelGGL(photometric_fwd_kernel<0>, grid, blk, 0, s, es, ta, out, c, h, w, block, eps); break;
    case 1: hipLaunchKernelGGL(photometric_fwd_kernel<1>, grid, blk, 0, s, es, ta, out, c, h, w, block, eps); break;
    case 2: hipLaunchKernelGGL(photometric_fwd_kernel<2>, grid, blk, 0, s, es, ta, out, c, h, w, block, eps); break;
    default: hipLaunchKernelGGL(photometric_fwd_kernel<3>, grid, blk, 0, s, es, ta, out, c, h, w, block, eps); break;
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_photometric_bwd(const float* es, const float* ta, const float* grad_out, float* grad_es, int n,
                                   int c, int h, int w, int block, int type, float eps, void* stream) {
  if (!es || !ta || !grad_out || !grad_es) return DIS_ERR_NULL;
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (block < 1 || (block & 1) == 0 || block / 2 > PH_MAXP || type < 0 || type > 3) return DIS_ERR_UNSUPPORTED;
  dim3 grid(dis_cdiv(w, PH_TX), dis_cdiv(h, PH_TY), n), blk(PH_TX * PH_TY);
  hipStream_t s = (hipStream_t)stream;
  switch (type) {
    case 0: hipLaunchKernelGGL(photometric_bwd_kernel<0>, grid, blk, 0, s, es, ta, grad_out, grad_es, c, h, w, block, eps); break;
    case 1: hipLaunchKernelGGL(photometric_bwd_kernel<1>, grid, blk, 0, s, es, ta, grad_out, grad_es, c, h, w, block, eps); break;
    case 2: hipLaunchKernelGGL(photometric_bwd_kernel<2>, grid, blk, 0, s, es, ta, grad_out, grad_es, c, h, w, block, eps); break;
    default: hipLaunchKernelGGL(photometric_bwd_kernel<3>, grid, blk, 0, s, es, ta, grad_out, grad_es, c, h, w, block, eps); break;
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// census window loss of S estimates against ONE target in one launch (DIS-SF: the four output scales are all compared with the
// same LCN image, reference model/single_frame_worker.py:110-118 -> model/networks.py:372 -> ext_functions.py:156-183).
// The soft sign of the TARGET differences, h(ta_nb - ta_c), is the same for every estimate: evaluated once per tap instead of
// once per estimate (5 reciprocal square roots per tap for S = 4 instead of 8; the kernels are bound by exactly this
// arithmetic).  9 x 9 window; 2 pixels per lane, a tap row of both is 10 consecutive LDS values (5 ds_read_b64 instead of 18
// ds_read_b32); the factor 0.5 of h is applied once at the end (a power of two: every partial sum scales exactly).  The results
// equal S calls of dis_photometric_fwd / dis_photometric_bwd TO ROUNDING, not bit for bit: these kernels form d * d + eps as one
// fused multiply-add (census_rsq_fma) and take the sign through a clamp (tests/test_pixel_ops_gpu.py compares both forms with the
// oracle and with each other; DIS_PHOTO_SINGLE_VIA_MULTI=0 keeps the general kernels on the single-estimate calls).
// ------------------------------------------------------------------------------------------------
// XCD-aware tile order for the 3-D grids of the tiled per-pixel kernels: workgroups are dealt to the 8 XCDs round-robin in launch
// order (x fastest), so horizontally / vertically neighbouring tiles - which share their halo - sat on different XCDs and every halo
// was fetched from HBM once per L2 (census_bwd_multi: 1 094 MB read for 368 MB algorithmic, L2 hit 0.17, profiles/r5v8_sf_bf16_kernels.md).
// Each XCD now works through a contiguous eighth of the tiles (same tiles, same arithmetic per tile: bit-identical results).
__device__ __forceinline__ void pm_tile_xcd(int& bx, int& by, int& bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  if ((total & 7u) == 0u) lin = (lin & 7u) * (total >> 3) + (lin >> 3);
  bx = (int)(lin % gx);
  by = (int)((lin / gx) % gy);
  bz = (int)(lin / (gx * gy));
}

#define PM_TX 64
#define PM_TY 8
#define PM_P 4
#define PM_TW (PM_TX + 2 * PM_P)
#define PM_TH (PM_TY + 2 * PM_P)

template <int TYPE, int S>
__global__ __launch_bounds__(256, 4) void census_fwd_multi_kernel(const float* __restrict__ es, const float* __restrict__ ta,
                                                               float* __restrict__ out, long sstride, int h, int w, float eps) {
  __shared__ __attribute__((aligned(16))) float tl[(S + 1) * PM_TH * PM_TW];  // [estimate 0..S-1, target][row][col]
  int bx_, by_, n;
  pm_tile_xcd(bx_, by_, n);
  const int x0 = bx_ * PM_TX, y0 = by_ * PM_TY;
  const long ibase = (long)n * h * w;
  for (int i = threadIdx.x; i < PM_TH * PM_TW; i += 256) {
    const int ty = i / PM_TW, tx = i - ty * PM_TW;
    const int gy = min(max(y0 + ty - PM_P, 0), h - 1), gx = min(max(x0 + tx - PM_P, 0), w - 1);  // replicate pad
    const long o = ibase + (long)gy * w + gx;
    float v[S + 1];
#pragma unroll
    for (int k = 0; k < S; ++k) v[k] = es[k * sstride + o];
    v[S] = ta[o];
#pragma unroll
    for (int k = 0; k <= S; ++k) tl[k * (PM_TH * PM_TW) + i] = v[k];
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const float* tt = tl + S * (PM_TH * PM_TW);
  float tc[2], ec[S][2], acc[S][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    tc[j] = tt[(ly + PM_P) * PM_TW + 2 * lx + j + PM_P];
#pragma unroll
    for (int k = 0; k < S; ++k) {
      ec[k][j] = tl[k * (PM_TH * PM_TW) + (ly + PM_P) * PM_TW + 2 * lx + j + PM_P];
      acc[k][j] = 0.f;
    }
  }
#pragma unroll 1
  for (int dy = 0; dy < 9; ++dy) {
    const int ro = (ly + dy) * PM_TW + 2 * lx;
    float tr[10], tq[2][9];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float2 v = *(const float2*)(tt + ro + 2 * q);
      tr[2 * q] = v.x;
      tr[2 * q + 1] = v.y;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int dx = 0; dx < 9; ++dx) {
        const float dt = tr[j + dx] - tc[j];
        tq[j][dx] = dt * census_rsq_fma(dt, eps);
      }
#pragma unroll
    for (int k = 0; k < S; ++k) {
      float er[10];
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const float2 v = *(const float2*)(tl + k * (PM_TH * PM_TW) + ro + 2 * q);
        er[2 * q] = v.x;
        er[2 * q + 1] = v.y;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          const float de = er[j + dx] - ec[k][j];
          const float v = __builtin_fmaf(de, census_rsq_fma(de, eps), -tq[j][dx]);   // 2 (h(de) - h(dt)): 4 VALU + 1 v_rsq per tap
          acc[k][j] = (TYPE == 2) ? __builtin_fmaf(v, v, acc[k][j]) : acc[k][j] + fabsf(v);
        }
      // (keep the next estimate's row reads behind this estimate's arithmetic: hoisting all S rows costs 40 registers and spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int py = y0 + ly;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = x0 + 2 * lx + j;
    if (px < w && py < h) {
#pragma unroll
      for (int k = 0; k < S; ++k)
        out[k * sstride + ibase + (long)py * w + px] = (acc[k][j] * (TYPE == 2 ? 0.25f : 0.5f)) / 81.f;
    }
  }
}

// census_sad backward, rare path (see census_bwd_multi_kernel): corrections of the S sums for one tap row of one pixel
template <int S>
__device__ __noinline__ float4 census_fix_row(const float* tl, int img, int ro, int cidx, int tw, float tk, float eps, int border,
                                              int px, int py, int dy, int h, int w) {
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  const float* tt = tl + S * img;
#pragma unroll 1
  for (int q = 0; q < 9 * S; ++q) {
    const int k = q / 9, dx = q - 9 * k;
    const float e = tl[k * img + ro + dx], t = tt[ro + dx], g = tl[(S + 1 + k) * img + ro + dx];
    const float ekk = tl[k * img + cidx], gkk = tl[(S + 1 + k) * img + cidx];
    const float des = ekk - e, dta = tk - t;
    const float re = census_rsq_fma(des, eps);
    const float v2 = __builtin_fmaf(des, re, -(dta * census_rsq_fma(dta, eps)));   // exactly the main path's value
    if (fabsf(v2) < 2e-5f && (des != 0.f || dta != 0.f)) {
      const float dx_ = census_h_exact(des, eps) - census_h_exact(dta, eps);
      const float s0 = census_sign(v2), s1 = dx_ > 0.f ? 1.f : (dx_ < 0.f ? -1.f : 0.f);
      float m = 1.f;
      if (border) m = (float)(clamp_mult(py + dy - PM_P, py, PM_P, h) * clamp_mult(px + dx - PM_P, px, PM_P, w));
      const float v = (m * g + gkk) * ((s1 - s0) * (re * re * re));   // (the caller's sum is scaled by 0.5 eps at the end)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) c[kk] += (kk == k) ? v : 0.f;
    }
  }
  return make_float4(c[0], c[1], c[2], c[3]);
}

// Backward: ONE pixel per lane (32 x 8 tile).  The two-pixel form of the forward kernel needs ~164 registers here (3 waves per
// SIMD) and ran slower than four single-estimate launches: the kernel is bound by VALU issue only while enough waves hide the
// LDS and v_rsq latencies (measured: 2.1 ms for 4 estimates against 4 x 0.385 ms).
#define PB_TX 32
#ifndef PB_WPE
#define PB_WPE 6       // waves per SIMD the register allocation aims at
#endif
#ifndef PB_DY_UNROLL
#define PB_DY_UNROLL 1  // tap rows per trip of the row loop
#endif
#define PB_STR_(x) #x
#define PB_STR(x) PB_STR_(x)
#define PB_TW (PB_TX + 2 * PM_P)
template <int TYPE, int S>
__global__ __launch_bounds__(256, PB_WPE) void census_bwd_multi_kernel(const float* __restrict__ es, const float* __restrict__ ta,
                                                               const float* __restrict__ gout, float* __restrict__ ges,
                                                               long sstride, int h, int w, float eps) {
  // [estimate 0..S-1][row][col], [target], [grad_out / 81 of estimate 0..S-1 (0 where no window centre exists)]
  constexpr int IMG = PM_TH * PB_TW;
  __shared__ float tl[(2 * S + 1) * IMG];
  int bx_, by_, n;
  pm_tile_xcd(bx_, by_, n);
  const int x0 = bx_ * PB_TX, y0 = by_ * PM_TY;
  const long ibase = (long)n * h * w;
  const float inv = 1.f / 81.f;
  for (int i = threadIdx.x; i < IMG; i += 256) {
    const int ty = i / PB_TW, tx = i - ty * PB_TW;
    const int ry = y0 + ty - PM_P, rx = x0 + tx - PM_P;
    const int gy = min(max(ry, 0), h - 1), gx = min(max(rx, 0), w - 1);
    const long o = ibase + (long)gy * w + gx;
    const bool in = ry >= 0 && ry < h && rx >= 0 && rx < w;
    float v[2 * S + 1];
#pragma unroll
    for (int k = 0; k < S; ++k) {
      v[k] = es[k * sstride + o];
      v[S + 1 + k] = gout[k * sstride + o];
    }
    v[S] = ta[o];
#pragma unroll
    for (int k = 0; k < S; ++k) {
      tl[k * IMG + i] = v[k];
      tl[(S + 1 + k) * IMG + i] = in ? v[S + 1 + k] * inv : 0.f;
    }
    tl[S * IMG + i] = v[S];
  }
  __syncthreads();
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const float* tt = tl + S * IMG;
  const int px = x0 + lx, py = y0 + ly;
  const int cidx = (ly + PM_P) * PB_TW + lx + PM_P;
  const float tk = tt[cidx];
  float ek[S], gk[S], acc[S];
#pragma unroll
  for (int k = 0; k < S; ++k) {
    ek[k] = tl[k * IMG + cidx];
    gk[k] = tl[(S + 1 + k) * IMG + cidx];
    acc[k] = 0.f;
  }
  const bool interior = px > 0 && px < w - 1 && py > 0 && py < h - 1;
  auto rows = [&](auto border_c) {
    constexpr bool BORDER = decltype(border_c)::value;
_Pragma(PB_STR(unroll PB_DY_UNROLL))
    for (int dy = 0; dy < 9; ++dy) {
      const int ro = (ly + dy) * PB_TW + lx;
      float tq[9], mult[BORDER ? 9 : 1];
#pragma unroll
      for (int dx = 0; dx < 9; ++dx) {
        const float dta = tk - tt[ro + dx];
        tq[dx] = dta * census_rsq_fma(dta, eps);
      }
      if (BORDER) {  // (pixels on the image border: replicate-pad multiplicities of the window centres)
        const int my = clamp_mult(py + dy - PM_P, py, PM_P, h);
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) mult[dx] = (float)(my * clamp_mult(px + dx - PM_P, px, PM_P, w));
      }
      unsigned mnd = 0xffffffffu;   // smallest NON-ZERO |difference| of the row (all estimates), as (float bits << 1) - 1
#pragma unroll
      for (int k = 0; k < S; ++k) {
        // role (a) neighbour of the window centre at this tap, role (b) centre of its own window: one evaluation serves both
        // (see photometric_bwd_kernel).  Branch-free main path; the sign of a near-zero difference (rare) is re-evaluated in the
        // reference's own operation order afterwards, for the whole tap row.
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          // per tap: 10 VALU + 1 v_rsq.  v = 2 (h(des) - h(dta)); the constant 0.5 eps of h' (and the 0.5 of h for census_mse)
          // multiplies the finished sum; sign(v) through a clamp, not compares and selects
          const float des = ek[k] - tl[k * IMG + ro + dx];
          const float re = census_rsq_fma(des, eps);
          const float v = __builtin_fmaf(des, re, -tq[dx]);
          mnd = min(mnd, (__float_as_uint(v) << 1) - 1u);   // 2 |v| as bits, exact zeros (centre tap, flat regions) -> largest
          const float sg = (TYPE == 2) ? v : census_sign(v);
          const float g = tl[(S + 1 + k) * IMG + ro + dx];
          const float wgt = BORDER ? __builtin_fmaf(mult[BORDER ? dx : 0], g, gk[k]) : g + gk[k];
          acc[k] = __builtin_fmaf(wgt, sg * (re * re * re), acc[k]);
        }
      }
      if (TYPE == 3 && mnd < (__float_as_uint(2e-5f) << 1) - 1u) {   // (0 < |h(des) - h(dta)| < 1e-5)
        // (rare: ~1e-5 of the taps) the SIGN of a near-zero difference is the reference's, evaluated in its own operation order
        // (IEEE sqrt and division): an out-of-line function that re-reads the row from LDS and returns the corrections of the
        // sums.  Inlined, its loop cost the main path its registers (543 spills at 6 waves per SIMD).
        const float4 c = census_fix_row<S>(tl, IMG, ro, cidx, PB_TW, tk, eps, BORDER ? 1 : 0, px, py, dy, h, w);
        acc[0] += c.x;
        if (S > 1) acc[S > 1 ? 1 : 0] += c.y;
        if (S > 2) acc[S > 2 ? 2 : 0] += c.z;
        if (S > 3) acc[S > 3 ? 3 : 0] += c.w;
      }
    }
  };
  if (interior) rows(std::false_type{});   // (divergent only in the tiles on the image border)
  else rows(std::true_type{});
  if (px < w && py < h) {
#pragma unroll
    for (int k = 0; k < S; ++k) ges[k * sstride + ibase + (long)py * w + px] = acc[k] * (0.5f * eps);
  }
}

template <int TYPE>
static void census_fwd_multi_launch(int s, dim3 grid, hipStream_t st, const float* es, const float* ta, float* out, long ss, int h,
                                    int w, float eps) {
  switch (s) {
    case 1: hipLaunchKernelGGL((census_fwd_multi_kernel<TYPE, 1>), grid, dim3(256), 0, st, es, ta, out, ss, h, w, eps); break;
    case 2: hipLaunchKernelGGL((census_fwd_multi_kernel<TYPE, 2>), grid, dim3(256), 0, st, es, ta, out, ss, h, w, eps); break;
    case 3: hipLaunchKernelGGL((census_fwd_multi_kernel<TYPE, 3>), grid, dim3(256), 0, st, es, ta, out, ss, h, w, eps); break;
    default: hipLaunchKernelGGL((census_fwd_multi_kernel<TYPE, 4>), grid, dim3(256), 0, st, es, ta, out, ss, h, w, eps); break;
  }
}
template <int TYPE>
static void census_bwd_multi_launch(int s, dim3 grid, hipStream_t st, const float* es, const float* ta, const float* go, float* ge,
                                    long ss, int h, int w, float eps) {
  switch (s) {
    case 1: hipLaunchKernelGGL((census_bwd_multi_kernel<TYPE, 1>), grid, dim3(256), 0, st, es, ta, go, ge, ss, h, w, eps); break;
    case 2: hipLaunchKernelGGL((census_bwd_multi_kernel<TYPE, 2>), grid, dim3(256), 0, st, es, ta, go, ge, ss, h, w, eps); break;
    case 3: hipLaunchKernelGGL((census_bwd_multi_kernel<TYPE, 3>), grid, dim3(256), 0, st, es, ta, go, ge, ss, h, w, eps); break;
    default: hipLaunchKernelGGL((census_bwd_multi_kernel<TYPE, 4>), grid, dim3(256), 0, st, es, ta, go, ge, ss, h, w, eps); break;
  }
}

extern "C" int dis_photometric_fwd_multi(const float* es, const float* ta, float* out, int s, int n, int h, int w, int block,
                                         int type, float eps, void* stream) {
  if (!es || !ta || !out) return DIS_ERR_NULL;
  if (s <= 0 || n <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (s > 4 || block != 9 || (type != 2 && type != 3) || n > 65535) return DIS_ERR_UNSUPPORTED;
  const dim3 grid(dis_cdiv(w, PM_TX), dis_cdiv(h, PM_TY), n);
  const long ss = (long)n * h * w;
  if (type == 2) census_fwd_multi_launch<2>(s, grid, (hipStream_t)stream, es, ta, out, ss, h, w, eps);
  else census_fwd_multi_launch<3>(s, grid, (hipStream_t)stream, es, ta, out, ss, h, w, eps);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_photometric_bwd_multi(const float* es, const float* ta, const float* grad_out, float* grad_es, int s, int n,
                                         int h, int w, int block, int type, float eps, void* stream) {
  if (!es || !ta || !grad_out || !grad_es) return DIS_ERR_NULL;
  if (s <= 0 || n <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (s > 4 || block != 9 || (type != 2 && type != 3) || n > 65535) return DIS_ERR_UNSUPPORTED;
  const dim3 grid(dis_cdiv(w, PB_TX), dis_cdiv(h, PM_TY), n);
  const long ss = (long)n * h * w;
  if (type == 2) census_bwd_multi_launch<2>(s, grid, (hipStream_t)stream, es, ta, grad_out, grad_es, ss, h, w, eps);
  else census_bwd_multi_launch<3>(s, grid, (hipStream_t)stream, es, ta, grad_out, grad_es, ss, h, w, eps);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// pattern projection (reference model/networks.py:358-367), 'border' padding, align_corners=True
// ------------------------------------------------------------------------------------------------
__global__ void pattern_warp_kernel(const float* __restrict__ pattern, const float* __restrict__ disp,
                                    const float* __restrict__ gproj, float* __restrict__ out, int n, int h, int w,
                                    int backward) {
  const long total = (long)n * h * w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    float ix = gs_roundtrip((float)x - disp[i], w);
    float iy = gs_roundtrip((float)y, h);
    // border: clip coordinates; gradient is zero where clipped
    float gmul = 1.f;
    if (ix <= 0.f) { ix = 0.f; gmul = 0.f; }   // ATen clip_coordinates_set_grad: grad 0 at/below 0 ...
    if (ix >= (float)(w - 1)) { ix = (float)(w - 1); gmul = 0.f; }
    iy = fminf(fmaxf(iy, 0.f), (float)(h - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = ix - fx, ex = 1.f - wx, wy = iy - fy, ey = 1.f - wy;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx1 = x1 <= w - 1, vy1 = y1 <= h - 1;
    const float nw = pattern[(long)y0 * w + x0];
    const float ne = vx1 ? pattern[(long)y0 * w + x1] : 0.f;
    const float sw = vy1 ? pattern[(long)y1 * w + x0] : 0.f;
    const float se = (vx1 && vy1) ? pattern[(long)y1 * w + x1] : 0.f;
    if (!backward) {
      out[i] = nw * (ey * ex) + ne * (ey * wx) + sw * (wy * ex) + se * (wy * wx);
    } else {
      // d out / d ix, then ix = (x - disp) up to the (linear) normalisation round trip => d ix / d disp = -1
      float gix = (ne - nw) * ey + (se - sw) * wy;
      out[i] = -gproj[i] * gix * gmul;
    }
  }
}

extern "C" int dis_pattern_warp_fwd(const float* pattern, const float* disp, float* proj, int n, int h, int w,
                                    void* stream) {
  if (!pattern || !disp || !proj) return DIS_ERR_NULL;
  if (n <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pattern_warp_kernel, dim3(dis_ew_grid((long)n * h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, pattern, disp, (const float*)nullptr, proj, n, h, w, 0);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_pattern_warp_bwd(const float* pattern, const float* disp, const float* grad_proj,
                                    float* grad_disp, int n, int h, int w, void* stream) {
  if (!pattern || !disp || !grad_proj || !grad_disp) return DIS_ERR_NULL;
  if (n <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(pattern_warp_kernel, dim3(dis_ew_grid((long)n * h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, pattern, disp, grad_proj, grad_disp, n, h, w, 1);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// scalar reductions
// ------------------------------------------------------------------------------------------------
__global__ void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                    double* __restrict__ acc, long count) {
  __shared__ double sm[8];
  double s0 = 0.0, s1 = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    float wi = w ? w[i] : 1.f;
    s0 += (double)(wi * x[i]);
    s1 += (double)wi;
  }
  double r0 = block_sum_d(s0, sm);
  double r1 = block_sum_d(s1, sm);
  if (threadIdx.x == 0) {
    atomic_add_d(acc, r0);
    atomic_add_d(acc + 1, r1);
  }
}
__global__ void ratio_finalize_kernel(const double* __restrict__ acc, float* __restrict__ out, double eps_den) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)((float)acc[0] / ((float)acc[1] + (float)eps_den));
}
__global__ void weighted_mean_bwd_kernel(const float* __restrict__ w, const double* __restrict__ acc,
                                         const float* __restrict__ gscale, float* __restrict__ gx, long count) {
  const float g = gscale[0] / (float)acc[1];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    gx[i] = g * (w ? w[i] : 1.f);
}

extern "C" int dis_weighted_mean_fwd(const float* x, const float* w, double* acc, float* out, long count,
                                     void* stream) {
  if (!x || !acc || !out) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(dis_red_grid(count, 256)), dim3(256), 0, s, x, w, acc, count);
  hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)acc, out, 0.0);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_weighted_mean_bwd(const float* w, const double* acc, const float* gscale, float* grad_x,
                                     long count, void* stream) {
  if (!acc || !gscale || !grad_x) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(weighted_mean_bwd_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     acc, gscale, grad_x, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

__global__ void l1_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ acc,
                              long count) {
  __shared__ double sm[8];
  double s0 = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    s0 += (double)fabsf(a[i] - b[i]);
  double r0 = block_sum_d(s0, sm);
  if (threadIdx.x == 0) atomic_add_d(acc, r0);
}
__global__ void mean_finalize_kernel(const double* __restrict__ acc, float* __restrict__ out, double count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(acc[0] / count);
}
__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ gscale, float* __restrict__ ga, long count) {
  const float g = gscale[0] / (float)count;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    float d = a[i] - b[i];
    ga[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
  }
}
extern "C" int dis_l1_mean_fwd(const float* a, const float* b, double* acc, float* out, long count, void* stream) {
  if (!a || !b || !acc || !out) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_sum_kernel, dim3(dis_red_grid(count, 256)), dim3(256), 0, s, a, b, acc, count);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)acc, out, (double)count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_l1_mean_bwd(const float* a, const float* b, const float* gscale, float* grad_a, long count,
                               void* stream) {
  if (!a || !b || !gscale || !grad_a) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, a, b, gscale,
                     grad_a, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// real-data warm-up term (reference model/multi_frame_worker.py:168-173, single_frame_worker.py:158-163):
//   valid = sgm > thresh;  val = sum(|o - sgm + noise| * valid) / sum(valid)
__global__ void sgm_l1_sum_kernel(const float* __restrict__ o, const float* __restrict__ sgm,
                                  const float* __restrict__ noise, float thresh, double* __restrict__ acc, long count) {
  __shared__ double sm[8];
  double s0 = 0.0, s1 = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float m = sgm[i] > thresh ? 1.f : 0.f;
    s0 += (double)(fabsf((o[i] - sgm[i]) + noise[i]) * m);
    s1 += (double)m;
  }
  double r0 = block_sum_d(s0, sm);
  double r1 = block_sum_d(s1, sm);
  if (threadIdx.x == 0) {
    atomic_add_d(acc, r0);
    atomic_add_d(acc + 1, r1);
  }
}
__global__ void sgm_l1_bwd_kernel(const float* __restrict__ o, const float* __restrict__ sgm,
                                  const float* __restrict__ noise, float thresh, const double* __restrict__ acc,
                                  const float* __restrict__ gscale, float* __restrict__ go, long count) {
  const float g = gscale[0] / (float)acc[1];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float d = (o[i] - sgm[i]) + noise[i];
    const float m = sgm[i] > thresh ? 1.f : 0.f;
    go[i] = m * (d > 0.f ? g : (d < 0.f ? -g : 0.f));
  }
}
extern "C" int dis_sgm_l1_fwd(const float* out_disp, const float* sgm_disp, const float* noise, float thresh,
                              double* acc, float* out, long count, void* stream) {
  if (!out_disp || !sgm_disp || !noise || !acc || !out) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sgm_l1_sum_kernel, dim3(dis_red_grid(count, 256)), dim3(256), 0, s, out_disp, sgm_disp, noise,
                     thresh, acc, count);
  hipLaunchKernelGGL(ratio_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)acc, out, 0.0);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_sgm_l1_bwd(const float* out_disp, const float* sgm_disp, const float* noise, float thresh,
                              const double* acc, const float* gscale, float* grad_out_disp, long count, void* stream) {
  if (!out_disp || !sgm_disp || !noise || !acc || !gscale || !grad_out_disp) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sgm_l1_bwd_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, out_disp,
                     sgm_disp, noise, thresh, acc, gscale, grad_out_disp, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// Sobel-5 smoothness (reference model/networks.py:411-431, 693-731)
// ------------------------------------------------------------------------------------------------
__constant__ float c_sobel5[25] = {-5.f / 240.f, -4.f / 240.f,  0.f, 4.f / 240.f,  5.f / 240.f,
                                   -8.f / 240.f, -10.f / 240.f, 0.f, 10.f / 240.f, 8.f / 240.f,
                                   -10.f / 240.f, -20.f / 240.f, 0.f, 20.f / 240.f, 10.f / 240.f,
                                   -8.f / 240.f, -10.f / 240.f, 0.f, 10.f / 240.f, 8.f / 240.f,
                                   -5.f / 240.f, -4.f / 240.f,  0.f, 4.f / 240.f,  5.f / 240.f};
#define SM_TX 32
#define SM_TY 8

// mode 0: accumulate sum |g * exp(-|255 ga|)| into acc.  mode 1: write s_x, s_y = sign(g) * e * gscale/count
__global__ __launch_bounds__(SM_TX* SM_TY) void smooth_kernel(const float* __restrict__ disp,
                                                                const float* __restrict__ amb,
                                                                double* __restrict__ acc,
                                                                const float* __restrict__ gscale,
                                                                float* __restrict__ splanes, int n_total, int h,
                                                                int w, int mode) {
  __shared__ float td[(SM_TY + 4) * (SM_TX + 4)];
  __shared__ float tam[(SM_TY + 4) * (SM_TX + 4)];
  __shared__ double sm[8];
  const int tiles_x = (w + SM_TX - 1) / SM_TX, tiles_y = (h + SM_TY - 1) / SM_TY;
  const long ntiles = (long)n_total * tiles_y * tiles_x;
  const int tw = SM_TX + 4, th = SM_TY + 4;
  const int lx = threadIdx.x % SM_TX, ly = threadIdx.x / SM_TX;
  double local = 0.0;
  // workgroups walk the tiles (mode 0 ends in one fp64 atomic per workgroup, not one per tile)
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int n = (int)(tile / ((long)tiles_y * tiles_x));
  const int x0 = (int)(tile % tiles_x) * SM_TX, y0 = (int)((tile / tiles_x) % tiles_y) * SM_TY;
  const float* idp = disp + (long)n * h * w;
  const float* iam = amb + (long)n * h * w;
  __syncthreads();  // the previous tile's LDS reads are done
  for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
    int ty = i / tw, tx = i - ty * tw;
    int gy = min(max(y0 + ty - 2, 0), h - 1), gx = min(max(x0 + tx - 2, 0), w - 1);
    td[i] = idp[(long)gy * w + gx];
    tam[i] = iam[(long)gy * w + gx];
  }
  __syncthreads();
  const int px = x0 + lx, py = y0 + ly;
  if (px < w && py < h) {
    float gx = 0.f, gy = 0.f, ax = 0.f, ay = 0.f;
    for (int dy = 0; dy < 5; ++dy)
      for (int dx = 0; dx < 5; ++dx) {
        float kx = c_sobel5[dy * 5 + dx], ky = c_sobel5[dx * 5 + dy];
        float d = td[(ly + dy) * tw + lx + dx], a = tam[(ly + dy) * tw + lx + dx];
        gx += kx * d;
        gy += ky * d;
        ax += kx * a;
        ay += ky * a;
      }
    float e_x = expf(-fabsf(255.f * ax)), e_y = expf(-fabsf(255.f * ay));
    if (mode == 0) {
      local += (double)fabsf(gx * e_x) + (double)fabsf(gy * e_y);
    } else {
      const float g = gscale[0] / (float)((long)n_total * 2 * h * w);
      float vx = gx * e_x, vy = gy * e_y;
      long o = (long)n * 2 * h * w + (long)py * w + px;
      splanes[o] = (vx > 0.f ? g : (vx < 0.f ? -g : 0.f)) * e_x;
      splanes[o + (long)h * w] = (vy > 0.f ? g : (vy < 0.f ? -g : 0.f)) * e_y;
    }
  }
  }  // tiles
  if (mode == 0) {
    double r = block_sum_d(local, sm);
    if (threadIdx.x == 0) atomic_add_d(acc, r);
  }
}

// grad_disp[k] = sum over window centres p and offsets off with clamp(p+off)==k of coef[off]*s[p]
__global__ void smooth_bwd_gather_kernel(const float* __restrict__ splanes, float* __restrict__ gdisp, int n, int h,
                                         int w) {
  const long total = (long)n * h * w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kx = (int)(i % w), ky = (int)((i / w) % h);
    const int b = (int)(i / ((long)h * w));
    const float* sx = splanes + (long)b * 2 * h * w;
    const float* sy = sx + (long)h * w;
    float acc = 0.f;
    for (int py = max(ky - 2, 0); py <= min(ky + 2, h - 1); ++py) {
      // offsets oy in [-2,2] with clamp(py+oy)==ky
      int oy_lo = ky - py, oy_hi = ky - py;
      if (ky == 0) oy_lo = -2;
      if (ky == h - 1) oy_hi = 2;
      for (int px = max(kx - 2, 0); px <= min(kx + 2, w - 1); ++px) {
        int ox_lo = kx - px, ox_hi = kx - px;
        if (kx == 0) ox_lo = -2;
        if (kx == w - 1) ox_hi = 2;
        float cx = 0.f, cy = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
          for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            cx += c_sobel5[(oy + 2) * 5 + ox + 2];
            cy += c_sobel5[(ox + 2) * 5 + oy + 2];
          }
        acc += cx * sx[(long)py * w + px] + cy * sy[(long)py * w + px];
      }
    }
    gdisp[i] = acc;
  }
}

// The same gather with the two sign planes of a 64 x 4 tile (+ 2 halo) staged in LDS and the 25 window offsets unrolled: the
// generic kernel above re-derives the replicate-pad multiplicities of every (centre, offset) pair with nested loops over constant
// memory - 277 us for 32 x 512 x 432 (0.3 TB/s; round-4 profile of DIS-SF).  Only the pixels ON the image border have
// multiplicities other than 1 (their window centres collect the clamped offsets): they keep the generic arithmetic.
#define SB_TX 64
#define SB_TY 4
__global__ __launch_bounds__(256) void smooth_bwd_gather_tiled_kernel(const float* __restrict__ splanes, float* __restrict__ gdisp,
                                                                      int n, int h, int w) {
  __shared__ float tsx[(SB_TY + 4) * (SB_TX + 4)], tsy[(SB_TY + 4) * (SB_TX + 4)];
  constexpr int TW = SB_TX + 4;
  const int b = blockIdx.z, x0 = blockIdx.x * SB_TX, y0 = blockIdx.y * SB_TY;
  const float* sx = splanes + (long)b * 2 * h * w;
  const float* sy = sx + (long)h * w;
  for (int i = threadIdx.x; i < (SB_TY + 4) * TW; i += 256) {
    const int ty = i / TW, tx = i - ty * TW;
    const int gy = y0 + ty - 2, gx = x0 + tx - 2;
    const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;   // (window centres outside the image do not exist)
    tsx[i] = in ? sx[(long)gy * w + gx] : 0.f;
    tsy[i] = in ? sy[(long)gy * w + gx] : 0.f;
  }
  __syncthreads();
  const int lx = threadIdx.x % SB_TX, ly = threadIdx.x / SB_TX;
  const int kx = x0 + lx, ky = y0 + ly;
  if (kx >= w || ky >= h) return;
  float acc = 0.f;
  if (kx > 0 && kx < w - 1 && ky > 0 && ky < h - 1) {
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {   // centre p = k + (dy, dx) reaches k with the offset (-dy, -dx)
        const int li = (ly + 2 + dy) * TW + lx + 2 + dx;
        acc += c_sobel5[(2 - dy) * 5 + 2 - dx] * tsx[li] + c_sobel5[(2 - dx) * 5 + 2 - dy] * tsy[li];
      }
  } else {  // image border: the generic multiplicities (smooth_bwd_gather_kernel)
    for (int py = max(ky - 2, 0); py <= min(ky + 2, h - 1); ++py) {
      int oy_lo = ky - py, oy_hi = ky - py;
      if (ky == 0) oy_lo = -2;
      if (ky == h - 1) oy_hi = 2;
      for (int px = max(kx - 2, 0); px <= min(kx + 2, w - 1); ++px) {
        int ox_lo = kx - px, ox_hi = kx - px;
        if (kx == 0) ox_lo = -2;
        if (kx == w - 1) ox_hi = 2;
        float cx = 0.f, cy = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
          for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            cx += c_sobel5[(oy + 2) * 5 + ox + 2];
            cy += c_sobel5[(ox + 2) * 5 + oy + 2];
          }
        acc += cx * sx[(long)py * w + px] + cy * sy[(long)py * w + px];
      }
    }
  }
  gdisp[((long)b * h + ky) * w + kx] = acc;
}

extern "C" int dis_smooth_loss_fwd(const float* disp, const float* amb, double* acc, float* out, int n, int h,
                                   int w, void* stream) {
  if (!disp || !amb || !acc || !out) return DIS_ERR_NULL;
  if (n <= 0 || h < 3 || w < 3) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dis_red_grid((long)n * dis_cdiv(w, SM_TX) * dis_cdiv(h, SM_TY), 1));
  hipLaunchKernelGGL(smooth_kernel, grid, dim3(SM_TX * SM_TY), 0, s, disp, amb, acc, (const float*)nullptr,
                     (float*)nullptr, n, h, w, 0);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)acc, out,
                     (double)((long)n * 2 * h * w));
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_smooth_loss_bwd(const float* disp, const float* amb, const float* gscale, float* grad_disp,
                                   float* workspace, int n, int h, int w, void* stream) {
  if (!disp || !amb || !gscale || !grad_disp || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || h < 3 || w < 3) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dis_ew_grid((long)n * dis_cdiv(w, SM_TX) * dis_cdiv(h, SM_TY), 1));
  hipLaunchKernelGGL(smooth_kernel, grid, dim3(SM_TX * SM_TY), 0, s, disp, amb, (double*)nullptr, gscale, workspace,
                     n, h, w, 1);
  if (n <= 65535)
    hipLaunchKernelGGL(smooth_bwd_gather_tiled_kernel, dim3(dis_cdiv(w, SB_TX), dis_cdiv(h, SB_TY), n), dim3(256), 0, s,
                       (const float*)workspace, grad_disp, n, h, w);
  else
    hipLaunchKernelGGL(smooth_bwd_gather_kernel, dim3(dis_ew_grid((long)n * h * w, 256)), dim3(256), 0, s,
                       (const float*)workspace, grad_disp, n, h, w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// disparity -> depth (reference model/networks.py:311-319)
// ------------------------------------------------------------------------------------------------
__global__ void d2d_kernel(const float* __restrict__ disp, const float* __restrict__ gdepth,
                           float* __restrict__ out, float bf, long count, int backward) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    float d = disp[i];
    float r = (d > 0.f ? d : 0.f) + 1e-12f;
    if (!backward) {
      // torch evaluates `python_float / tensor` as tensor.reciprocal() * float (Tensor.__rtruediv__): two roundings
      out[i] = (1.f / r) * bf;
    } else {
      out[i] = d > 0.f ? -gdepth[i] * bf / (r * r) : 0.f;
    }
  }
}
extern "C" int dis_disp_to_depth_fwd(const float* disp, float* depth, float bf, long count, void* stream) {
  if (!disp || !depth) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(d2d_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, disp,
                     (const float*)nullptr, depth, bf, count, 0);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_disp_to_depth_bwd(const float* disp, const float* grad_depth, float* grad_disp, float bf,
                                     long count, void* stream) {
  if (!disp || !grad_depth || !grad_disp) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(d2d_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, disp, grad_depth,
                     grad_disp, bf, count, 1);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// flow-consistency (geometric) loss, one direction (reference model/networks.py:564-601, 619-655)
// ------------------------------------------------------------------------------------------------
struct GeoCam {
  float K[9];
  float Ki[9];
};

// ray = [u,v,1] . Ki^T evaluated in fp64 and rounded to fp32, as numpy does for int64 @ float32
__device__ __forceinline__ void pixel_ray(const float* Ki, int u, int v, float* r) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
    r[c] = (float)((double)u * (double)Ki[c * 3 + 0] + (double)v * (double)Ki[c * 3 + 1] + (double)Ki[c * 3 + 2]);
}
// The masks this kernel thresholds are index-class outputs, so every value that reaches a comparison is rounded exactly
// as the reference's CPU run rounds it (tests/bitexact.py restates these chains in numpy and is checked against the oracle
// bit for bit): torch.bmm accumulates a K = 3 product as the fused chain fma(a2, b2, fma(a1, b1, a0 * b0)), grid_sample
// sums its four taps as fma(se, w, fma(sw, w, fma(ne, w, nw * w))).  The library is built with -ffp-contract=off, so
// the only fused operations are the explicit ones.
// row vector times 3x3 (row-major M): out_c = sum_k a_k M[k][c]
__device__ __forceinline__ void vec_mat(const float* a, const float* M, float* o) {
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = __fmaf_rn(a[2], M[2 * 3 + c], __fmaf_rn(a[1], M[1 * 3 + c], a[0] * M[0 * 3 + c]));
}
// row vector times M^T: out_c = sum_k a_k M[c][k]
__device__ __forceinline__ void vec_matT(const float* a, const float* M, float* o) {
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = __fmaf_rn(a[2], M[c * 3 + 2], __fmaf_rn(a[1], M[c * 3 + 1], a[0] * M[c * 3 + 0]));
}

struct Bilin {
  int x0, y0;
  float nw, ne, sw, se;   // weights
  bool v00, v01, v10, v11;  // tap validity (zeros padding)
};
__device__ __forceinline__ Bilin bilin_zeros(float px, float py, int h, int w) {
  Bilin b;
  float ix = gs_roundtrip(px, w), iy = gs_roundtrip(py, h);
  float fx = floorf(ix), fy = floorf(iy);
  // keep the int conversion defined for far-out coordinates
  fx = fminf(fmaxf(fx, -2.f), (float)w + 1.f);
  fy = fminf(fmaxf(fy, -2.f), (float)h + 1.f);
  ix = fminf(fmaxf(ix, -2.f), (float)w + 2.f);
  iy = fminf(fmaxf(iy, -2.f), (float)h + 2.f);
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  float wx = ix - fx, ex = 1.f - wx, wy = iy - fy, ey = 1.f - wy;
  b.nw = ey * ex; b.ne = ey * wx; b.sw = wy * ex; b.se = wy * wx;
  bool xa = b.x0 >= 0 && b.x0 < w, xb = b.x0 + 1 >= 0 && b.x0 + 1 < w;
  bool ya = b.y0 >= 0 && b.y0 < h, yb = b.y0 + 1 >= 0 && b.y0 + 1 < h;
  b.v00 = xa && ya; b.v01 = xb && ya; b.v10 = xa && yb; b.v11 = xb && yb;
  return b;
}
__device__ __forceinline__ float bilin_fetch(const float* img, const Bilin& b, int w) {
  float a = b.v00 ? img[(long)b.y0 * w + b.x0] : 0.f;
  float c = b.v01 ? img[(long)b.y0 * w + b.x0 + 1] : 0.f;
  float d = b.v10 ? img[(long)(b.y0 + 1) * w + b.x0] : 0.f;
  float e = b.v11 ? img[(long)(b.y0 + 1) * w + b.x0 + 1] : 0.f;
  return __fmaf_rn(e, b.se, __fmaf_rn(d, b.sw, __fmaf_rn(c, b.ne, a * b.nw)));
}

// z (and optionally uv) of pixel (u,v) of camera A with depth d, reprojected into camera B
__device__ __forceinline__ void reproject(const GeoCam& cam, int u, int v, float d, const float* RA,
                                          const float* tA, const float* RB, const float* tB, float* uvw) {
  float ray[3], xyz[3], xw[3], xc[3];
  pixel_ray(cam.Ki, u, v, ray);
#pragma unroll
  for (int c = 0; c < 3; ++c) xyz[c] = d * ray[c] - tA[c];
  vec_mat(xyz, RA, xw);
  vec_matT(xw, RB, xc);
#pragma unroll
  for (int c = 0; c < 3; ++c) xc[c] += tB[c];
  vec_matT(xc, cam.K, uvw);  // uvw = xc . K^T
}

// (bid / nblk: the block's index and the block count of THIS term - of a launch of its own, or its row of a multi-term launch)
__device__ __forceinline__ void geo_loss_fwd_body(double* sm, const float* __restrict__ depth0, const float* __restrict__ depth1,
                                                  const float* __restrict__ flow0, const float* __restrict__ flow1,
                                                  const float* __restrict__ amb0, const float* __restrict__ amb1,
                                                  const float* __restrict__ pdepth1, const float* __restrict__ R0,
                                                  const float* __restrict__ t0, const float* __restrict__ R1,
                                                  const float* __restrict__ t1, const GeoCam& cam, float clampv,
                                                  float* __restrict__ mask_out, double* __restrict__ acc, int bs, int h, int w,
                                                  const int bid, const int nblk) {
  const long hw = (long)h * w, total = (long)bs * hw;
  double s_diff = 0.0, s_mask = 0.0;
  for (long i = bid * (long)blockDim.x + threadIdx.x; i < total; i += (long)nblk * blockDim.x) {
    const int b = (int)(i / hw);
    const long p = i - (long)b * hw;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float *rA = R0 + b * 9, *tA = t0 + b * 3, *rB = R1 + b * 9, *tB = t1 + b * 3;
    float uvw[3];
    reproject(cam, x, y, depth0[i], rA, tA, rB, tB, uvw);
    const float d1 = uvw[2];
    const float f0x = flow0[(long)b * 2 * hw + p], f0y = flow0[(long)b * 2 * hw + hw + p];
    Bilin bl = bilin_zeros(f0x + (float)x, f0y + (float)y, h, w);
    const float depth10 = bilin_fetch(depth1 + (long)b * hw, bl, w);
    float diff = fabsf(d1 - depth10);
    if (clampv > 0.f) diff = fminf(fmaxf(diff, 0.f), clampv);
    // masks (no gradient)
    const float f10x = bilin_fetch(flow1 + (long)b * 2 * hw, bl, w);
    const float f10y = bilin_fetch(flow1 + (long)b * 2 * hw + hw, bl, w);
    const float sx = f0x + f10x, sy = f0y + f10y;
    const float lhs = sx * sx + sy * sy;
    const float rhs = 0.5f + 0.02f * ((f0x * f0x + f0y * f0y) + (f10x * f10x + f10y * f10y));
    float m = lhs < rhs ? 1.f : 0.f;
    const float amb10 = bilin_fetch(amb1 + (long)b * hw, bl, w);
    m *= (fabsf(amb0[i] - amb10) < 0.01f) ? 1.f : 0.f;
    if (pdepth1) {
      // uv0 = projection of frame-1 primary depth into frame 0, sampled at the flow target (4 taps)
      float wu = 0.f, wv = 0.f;
      const float* pd = pdepth1 + (long)b * hw;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int tx = bl.x0 + (k & 1), ty = bl.y0 + (k >> 1);
        const bool valid = (k == 0) ? bl.v00 : (k == 1) ? bl.v01 : (k == 2) ? bl.v10 : bl.v11;
        const float wgt = (k == 0) ? bl.nw : (k == 1) ? bl.ne : (k == 2) ? bl.sw : bl.se;
        if (valid) {
          float q[3];
          reproject(cam, tx, ty, pd[(long)ty * w + tx], rB, tB, rA, tA, q);
          float den = (q[2] > 0.f ? q[2] : 0.f) + 1e-12f;
          wu = __fmaf_rn(q[0] / den, wgt, wu);  // (the first valid tap: fma(v, w, 0) = v * w)
          wv = __fmaf_rn(q[1] / den, wgt, wv);
        }
      }
      const float du = wu - (float)x, dv = wv - (float)y;
      m *= (du * du + dv * dv < 1.f) ? 1.f : 0.f;
    }
    mask_out[i] = m;
    s_diff += (double)(diff * m);
    s_mask += (double)m;
  }
  // per-block partial sums to the block's slot behind acc[0..1] (summed in a fixed order by geo_finalize_kernel): two same-address
  // fp64 atomics per block serialise at ~10 ns each, which capped the grid at 512 blocks (8 waves per CU) for a kernel that is
  // bound by the latency of its dependent gathers - 48 us per launch at 512 blocks with atomics, ~25 us at 1024 blocks with slots
  double r0 = block_sum_d(s_diff, sm);
  double r1 = block_sum_d(s_mask, sm);
  if (threadIdx.x == 0) {
    acc[2 + 2 * bid] = r0;
    acc[3 + 2 * bid] = r1;
  }
}
__global__ void geo_loss_fwd_kernel(const float* __restrict__ depth0, const float* __restrict__ depth1,
                                    const float* __restrict__ flow0, const float* __restrict__ flow1,
                                    const float* __restrict__ amb0, const float* __restrict__ amb1,
                                    const float* __restrict__ pdepth1, const float* __restrict__ R0,
                                    const float* __restrict__ t0, const float* __restrict__ R1,
                                    const float* __restrict__ t1, GeoCam cam, float clampv,
                                    float* __restrict__ mask_out, double* __restrict__ acc, int bs, int h, int w) {
  __shared__ double sm[8];
  geo_loss_fwd_body(sm, depth0, depth1, flow0, flow1, amb0, amb1, pdepth1, R0, t0, R1, t1, cam, clampv, mask_out, acc, bs, h, w,
                    (int)blockIdx.x, (int)gridDim.x);
}
#define GEO_BLOCKS 1024
__global__ __launch_bounds__(256) void geo_finalize_kernel(double* __restrict__ acc, float* __restrict__ out, int nblocks,
                                                           double eps_den) {
  __shared__ double sm[8];
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
    a += acc[2 + 2 * k];
    b += acc[3 + 2 * k];
  }
  a = block_sum_d(a, sm);
  b = block_sum_d(b, sm);
  if (threadIdx.x == 0) {
    acc[0] = a;
    acc[1] = b;
    out[0] = (float)((float)a / ((float)b + (float)eps_den));
  }
}
extern "C" long dis_geo_loss_acc_doubles(void) { return 2 + 2 * GEO_BLOCKS; }

__global__ void geo_loss_bwd_kernel(const float* __restrict__ depth0, const float* __restrict__ depth1,
                                    const float* __restrict__ flow0, const float* __restrict__ R0,
                                    const float* __restrict__ t0, const float* __restrict__ R1,
                                    const float* __restrict__ t1, GeoCam cam, float clampv,
                                    const float* __restrict__ mask, const double* __restrict__ acc,
                                    const float* __restrict__ gscale, float* __restrict__ gdepth0,
                                    float* __restrict__ gdepth1, int bs, int h, int w) {
  const long hw = (long)h * w, total = (long)bs * hw;
  const float gs = gscale[0] / ((float)acc[1] + 1e-8f);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float m = mask[i];
    if (m == 0.f) continue;
    const int b = (int)(i / hw);
    const long p = i - (long)b * hw;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float *rA = R0 + b * 9, *tA = t0 + b * 3, *rB = R1 + b * 9, *tB = t1 + b * 3;
    float uvw[3];
    reproject(cam, x, y, depth0[i], rA, tA, rB, tB, uvw);
    const float d1 = uvw[2];
    const float f0x = flow0[(long)b * 2 * hw + p], f0y = flow0[(long)b * 2 * hw + hw + p];
    Bilin bl = bilin_zeros(f0x + (float)x, f0y + (float)y, h, w);
    const float depth10 = bilin_fetch(depth1 + (long)b * hw, bl, w);
    const float raw = d1 - depth10;
    float a = fabsf(raw);
    if (clampv > 0.f && a > clampv) continue;  // clamp passes gradient only inside [0, clamp]
    const float sgn = raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
    const float g = gs * m * sgn;
    // d d1 / d depth0 = ((ray R0) R1^T K^T)_z
    float ray[3], a1[3], a2[3], a3[3];
    pixel_ray(cam.Ki, x, y, ray);
    vec_mat(ray, rA, a1);
    vec_matT(a1, rB, a2);
    vec_matT(a2, cam.K, a3);
    gdepth0[i] += g * a3[2];
    float* g1 = gdepth1 + (long)b * hw;
    if (bl.v00) atomicAdd(g1 + (long)bl.y0 * w + bl.x0, -g * bl.nw);
    if (bl.v01) atomicAdd(g1 + (long)bl.y0 * w + bl.x0 + 1, -g * bl.ne);
    if (bl.v10) atomicAdd(g1 + (long)(bl.y0 + 1) * w + bl.x0, -g * bl.sw);
    if (bl.v11) atomicAdd(g1 + (long)(bl.y0 + 1) * w + bl.x0 + 1, -g * bl.se);
  }
}

static void fill_cam(GeoCam& cam, const float* K, const float* Ki) {
  for (int i = 0; i < 9; ++i) {
    cam.K[i] = K[i];
    cam.Ki[i] = Ki[i];
  }
}

extern "C" int dis_geo_loss_fwd(const float* depth0, const float* depth1, const float* flow0, const float* flow1,
                                const float* amb0, const float* amb1, const float* primary_depth1,
                                const float* R0, const float* t0, const float* R1, const float* t1,
                                const float* K_host, const float* Kinv_host, float clampv, float* mask_out,
                                double* acc, float* out, int bs, int h, int w, void* stream) {
  if (!depth0 || !depth1 || !flow0 || !flow1 || !amb0 || !amb1 || !R0 || !t0 || !R1 || !t1 || !K_host ||
      !Kinv_host || !mask_out || !acc || !out)
    return DIS_ERR_NULL;
  if (bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  GeoCam cam;
  fill_cam(cam, K_host, Kinv_host);
  hipStream_t s = (hipStream_t)stream;
  long gl = ((long)bs * h * w + 255) / 256;
  const int grid = (int)(gl > GEO_BLOCKS ? GEO_BLOCKS : gl);
  hipLaunchKernelGGL(geo_loss_fwd_kernel, dim3(grid), dim3(256), 0, s, depth0, depth1,
                     flow0, flow1, amb0, amb1, primary_depth1, R0, t0, R1, t1, cam, clampv, mask_out, acc, bs, h, w);
  hipLaunchKernelGGL(geo_finalize_kernel, dim3(1), dim3(256), 0, s, acc, out, grid, 1e-8);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_geo_loss_bwd(const float* depth0, const float* depth1, const float* flow0, const float* R0,
                                const float* t0, const float* R1, const float* t1, const float* K_host,
                                const float* Kinv_host, float clampv, const float* mask, const double* acc,
                                const float* gscale, float* grad_depth0, float* grad_depth1, int bs, int h, int w,
                                void* stream) {
  if (!depth0 || !depth1 || !flow0 || !R0 || !t0 || !R1 || !t1 || !K_host || !Kinv_host || !mask || !acc ||
      !gscale || !grad_depth0 || !grad_depth1)
    return DIS_ERR_NULL;
  if (bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  GeoCam cam;
  fill_cam(cam, K_host, Kinv_host);
  hipLaunchKernelGGL(geo_loss_bwd_kernel, dim3(dis_ew_grid((long)bs * h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, depth0, depth1, flow0, R0, t0, R1, t1, cam, clampv, mask, acc, gscale,
                     grad_depth0, grad_depth1, bs, h, w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ---- all directional terms of a step in one launch each way (round 5) ----
// A training step evaluates tl (tl - 1) = 12 directional terms (model/multi_frame_worker.py:139-158, single_frame_worker.py:126-149),
// each over bs images only: 12 forward, 12 finalize and 12 backward launches of 20 - 40 us that are bound by the latency of their
// dependent gathers, not by their 50 MB.  Here blockIdx.y selects the term (pointer table by value), every term keeps its own block
// slots, sums and 1 / (mask sum) - the forward values are those of the single launches bit for bit; the backward adds BOTH depth
// gradients with float atomics (terms that share a frame run concurrently; the single launch's `+=` would race), so the depth
// gradient repeats to rounding, not bitwise, like the warped frame's always did.
#define GEO_MULTI_MAX 16
#define GEO_MULTI_BLOCKS 1024   // (= GEO_BLOCKS: the same partition of the pixels as a single-term launch, hence the same sums)
struct GeoTermTable {
  DisGeoTerm t[GEO_MULTI_MAX];
};
__global__ __launch_bounds__(256) void geo_loss_fwd_multi_kernel(const GeoTermTable tab, GeoCam cam, float clampv,
                                                                 double* __restrict__ acc, int bs, int h, int w) {
  __shared__ double sm[8];
  const DisGeoTerm& q = tab.t[blockIdx.y];
  geo_loss_fwd_body(sm, q.depth0, q.depth1, q.flow0, q.flow1, q.amb0, q.amb1, q.pdepth1, q.R0, q.t0, q.R1, q.t1, cam, clampv, q.mask,
                    acc + (long)blockIdx.y * (2 + 2 * GEO_MULTI_BLOCKS), bs, h, w, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(256) void geo_finalize_multi_kernel(double* __restrict__ acc, float* __restrict__ out, int nblocks,
                                                                 double eps_den) {
  __shared__ double sm[8];
  double* ac = acc + (long)blockIdx.x * (2 + 2 * GEO_MULTI_BLOCKS);
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
    a += ac[2 + 2 * k];
    b += ac[3 + 2 * k];
  }
  a = block_sum_d(a, sm);
  b = block_sum_d(b, sm);
  if (threadIdx.x == 0) {
    ac[0] = a;
    ac[1] = b;
    out[blockIdx.x] = (float)((float)a / ((float)b + (float)eps_den));
  }
}
__global__ __launch_bounds__(256) void geo_loss_bwd_multi_kernel(const GeoTermTable tab, GeoCam cam, float clampv,
                                                                 const double* __restrict__ acc, const float* __restrict__ gscale,
                                                                 int bs, int h, int w) {
  const DisGeoTerm& q = tab.t[blockIdx.y];
  const float* __restrict__ depth0 = q.depth0;
  const float* __restrict__ depth1 = q.depth1;
  const float* __restrict__ flow0 = q.flow0;
  const float* __restrict__ mask = q.mask;
  const long hw = (long)h * w, total = (long)bs * hw;
  const float gs = gscale[blockIdx.y] / ((float)acc[(long)blockIdx.y * (2 + 2 * GEO_MULTI_BLOCKS) + 1] + 1e-8f);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float m = mask[i];
    if (m == 0.f) continue;
    const int b = (int)(i / hw);
    const long p = i - (long)b * hw;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float *rA = q.R0 + b * 9, *tA = q.t0 + b * 3, *rB = q.R1 + b * 9, *tB = q.t1 + b * 3;
    float uvw[3];
    reproject(cam, x, y, depth0[i], rA, tA, rB, tB, uvw);
    const float d1 = uvw[2];
    const float f0x = flow0[(long)b * 2 * hw + p], f0y = flow0[(long)b * 2 * hw + hw + p];
    Bilin bl = bilin_zeros(f0x + (float)x, f0y + (float)y, h, w);
    const float depth10 = bilin_fetch(depth1 + (long)b * hw, bl, w);
    const float raw = d1 - depth10;
    const float a = fabsf(raw);
    if (clampv > 0.f && a > clampv) continue;
    const float sgn = raw > 0.f ? 1.f : (raw < 0.f ? -1.f : 0.f);
    const float g = gs * m * sgn;
    float ray[3], a1[3], a2[3], a3[3];
    pixel_ray(cam.Ki, x, y, ray);
    vec_mat(ray, rA, a1);
    vec_matT(a1, rB, a2);
    vec_matT(a2, cam.K, a3);
    atomicAdd(q.gdepth0 + i, g * a3[2]);
    float* g1 = q.gdepth1 + (long)b * hw;
    if (bl.v00) atomicAdd(g1 + (long)bl.y0 * w + bl.x0, -g * bl.nw);
    if (bl.v01) atomicAdd(g1 + (long)bl.y0 * w + bl.x0 + 1, -g * bl.ne);
    if (bl.v10) atomicAdd(g1 + (long)(bl.y0 + 1) * w + bl.x0, -g * bl.sw);
    if (bl.v11) atomicAdd(g1 + (long)(bl.y0 + 1) * w + bl.x0 + 1, -g * bl.se);
  }
}
extern "C" long dis_geo_loss_multi_acc_doubles(int nterms) {
  return nterms > 0 && nterms <= GEO_MULTI_MAX ? (long)nterms * (2 + 2 * GEO_MULTI_BLOCKS) : -1;
}
static int geo_multi_table(const DisGeoTerm* terms, int nterms, bool bwd, GeoTermTable* tab) {
  for (int k = 0; k < nterms; ++k) {
    const DisGeoTerm& q = terms[k];
    if (!q.depth0 || !q.depth1 || !q.flow0 || !q.R0 || !q.t0 || !q.R1 || !q.t1 || !q.mask) return DIS_ERR_NULL;
    if (!bwd && (!q.flow1 || !q.amb0 || !q.amb1)) return DIS_ERR_NULL;
    if (bwd && (!q.gdepth0 || !q.gdepth1)) return DIS_ERR_NULL;
    tab->t[k] = q;
  }
  for (int k = nterms; k < GEO_MULTI_MAX; ++k) tab->t[k] = DisGeoTerm{};
  return DIS_OK;
}
extern "C" int dis_geo_loss_fwd_multi(const DisGeoTerm* terms, int nterms, const float* K_host, const float* Kinv_host, float clampv,
                                      double* acc, float* out, int bs, int h, int w, void* stream) {
  if (!terms || !K_host || !Kinv_host || !acc || !out) return DIS_ERR_NULL;
  if (nterms <= 0 || bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  if (nterms > GEO_MULTI_MAX) return DIS_ERR_UNSUPPORTED;
  GeoTermTable tab;
  const int rc = geo_multi_table(terms, nterms, false, &tab);
  if (rc != DIS_OK) return rc;
  GeoCam cam;
  fill_cam(cam, K_host, Kinv_host);
  hipStream_t s = (hipStream_t)stream;
  const long gl = ((long)bs * h * w + 255) / 256;
  const int grid = (int)(gl > GEO_MULTI_BLOCKS ? GEO_MULTI_BLOCKS : gl);
  hipLaunchKernelGGL(geo_loss_fwd_multi_kernel, dim3(grid, nterms), dim3(256), 0, s, tab, cam, clampv, acc, bs, h, w);
  hipLaunchKernelGGL(geo_finalize_multi_kernel, dim3(nterms), dim3(256), 0, s, acc, out, grid, 1e-8);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_geo_loss_bwd_multi(const DisGeoTerm* terms, int nterms, const float* K_host, const float* Kinv_host, float clampv,
                                      const double* acc, const float* gscale, int bs, int h, int w, void* stream) {
  if (!terms || !K_host || !Kinv_host || !acc || !gscale) return DIS_ERR_NULL;
  if (nterms <= 0 || bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  if (nterms > GEO_MULTI_MAX) return DIS_ERR_UNSUPPORTED;
  GeoTermTable tab;
  const int rc = geo_multi_table(terms, nterms, true, &tab);
  if (rc != DIS_OK) return rc;
  GeoCam cam;
  fill_cam(cam, K_host, Kinv_host);
  long gl = ((long)bs * h * w + 255) / 256;
  if (gl > 1024) gl = 1024;
  hipLaunchKernelGGL(geo_loss_bwd_multi_kernel, dim3((unsigned)gl, nterms), dim3(256), 0, (hipStream_t)stream, tab, cam, clampv, acc,
                     gscale, bs, h, w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// training-time image augmentation on the device (reference data/data_manipulation.py:114-195 with the dataset's settings
// data/dataset.py:67-70: max_shift 0 -> no affine part, max_blur 0.5, max_noise 3, max_sp_noise 5e-4), applied to the
// IR image and the ambient image of every frame after the host->device copy instead of by cv2 / numpy in the loader
// processes.  Per image i, params[i] = {blur flag, sigma_im, sigma_amb, noise_im, noise_amb, sp_ratio (<0: none)}, drawn by
// the caller; the per-pixel randomness comes from a counter-based generator keyed by (seed, image, pixel, stream).
//   1. 5x5 Gaussian blur (cv2.GaussianBlur: kernel exp(-x^2 / 2 sigma^2) normalised, BORDER_REFLECT_101), image and ambient
//      with their own sigma                                                              [blur flag]
//   2. + N(0,1) * noise / 255, image and ambient with their own amplitude
//   3. salt (-> the ORIGINAL image's max) then pepper (-> its min), each pixel with probability sp_ratio; image only
//      (the reference sets int(N * ratio) coordinates drawn with replacement: the same marginal distribution)
//   4. clip to [0, 1]
// ------------------------------------------------------------------------------------------------
#define AUG_NP 6
__device__ __forceinline__ unsigned aug_hash(unsigned x) {  // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float aug_uniform(unsigned long long seed, unsigned img, unsigned pix, unsigned stream) {
  unsigned h = aug_hash((unsigned)seed ^ aug_hash((unsigned)(seed >> 32) + 0x9e3779b9U * (stream + 1)));
  h = aug_hash(h ^ aug_hash(img * 0x85ebca6bU + 0xc2b2ae35U) ^ aug_hash(pix + 0x27d4eb2fU * stream));
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
}
__device__ __forceinline__ float aug_normal(unsigned long long seed, unsigned img, unsigned pix, unsigned stream) {
  const float u1 = aug_uniform(seed, img, pix, 2 * stream), u2 = aug_uniform(seed, img, pix, 2 * stream + 1);
  return sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
}
__device__ __forceinline__ int aug_reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__global__ void aug_minmax_kernel(const float* __restrict__ im, unsigned* __restrict__ mm, int n, long hw) {
  // images are >= 0, so the unsigned bit pattern orders like the value; mm[2i] = min bits (init 0x7f800000), mm[2i+1] = max
  const int i = blockIdx.y;
  unsigned lo = 0x7f800000u, hi = 0u;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(fmaxf(im[(long)i * hw + p], 0.f));
    lo = min(lo, b);
    hi = max(hi, b);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(mm + 2 * i, lo);
    atomicMax(mm + 2 * i + 1, hi);
  }
}
__global__ void aug_init_minmax_kernel(unsigned* __restrict__ mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    mm[2 * i] = 0x7f800000u;
    mm[2 * i + 1] = 0u;
  }
}
__device__ __forceinline__ float aug_blur5(const float* __restrict__ img, int x, int y, int h, int w, float sigma) {
  float k[5], ks = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float d = (float)(i - 2);
    k[i] = expf(-(d * d) / (2.f * sigma * sigma));
    ks += k[i];
  }
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy) {
    const int yy = aug_reflect101(y + dy - 2, h);
    float row = 0.f;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) row += k[dx] * img[(long)yy * w + aug_reflect101(x + dx - 2, w)];
    acc += k[dy] * row;
  }
  return acc / (ks * ks);
}
__global__ void augment_kernel(const float* __restrict__ im, const float* __restrict__ amb,
                               const float* __restrict__ params, const unsigned* __restrict__ mm,
                               const long long* __restrict__ seed_dev, float* __restrict__ out_im,
                               float* __restrict__ out_amb, int n, int h, int w) {
  const long hw = (long)h * w, total = (long)n * hw;
  const unsigned long long seed = (unsigned long long)seed_dev[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / hw);
    const long p = i - (long)b * hw;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float* P = params + (long)b * AUG_NP;
    float vi = im[i], va = amb[i];
    if (P[0] > 0.5f) {
      vi = aug_blur5(im + (long)b * hw, x, y, h, w, P[1]);
      va = aug_blur5(amb + (long)b * hw, x, y, h, w, P[2]);
    }
    vi += aug_normal(seed, (unsigned)b, (unsigned)p, 0) * (P[3] / 255.f);
    va += aug_normal(seed, (unsigned)b, (unsigned)p, 1) * (P[4] / 255.f);
    if (P[5] >= 0.f) {
      if (aug_uniform(seed, (unsigned)b, (unsigned)p, 8) < P[5]) vi = __uint_as_float(mm[2 * b + 1]);
      if (aug_uniform(seed, (unsigned)b, (unsigned)p, 9) < P[5]) vi = __uint_as_float(mm[2 * b]);
    }
    out_im[i] = fminf(fmaxf(vi, 0.f), 1.f);
    out_amb[i] = fminf(fmaxf(va, 0.f), 1.f);
  }
}
extern "C" int dis_augment(const float* im, const float* amb, const float* params, const long long* seed,
                           unsigned* minmax_ws, float* out_im, float* out_amb, int n, int h, int w, void* stream) {
  if (!im || !amb || !params || !seed || !minmax_ws || !out_im || !out_amb) return DIS_ERR_NULL;
  if (n <= 0 || h < 3 || w < 3) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const long hw = (long)h * w;
  hipLaunchKernelGGL(aug_init_minmax_kernel, dim3(dis_cdiv(n, 64)), dim3(64), 0, s, minmax_ws, n);
  hipLaunchKernelGGL(aug_minmax_kernel, dim3((unsigned)min((long)64, dis_cdiv(hw, 256)), n), dim3(256), 0, s, im, minmax_ws,
                     n, hw);
  hipLaunchKernelGGL(augment_kernel, dim3(dis_ew_grid((long)n * hw, 256)), dim3(256), 0, s, im, amb, params,
                     (const unsigned*)minmax_ws, seed, out_im, out_amb, n, h, w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_abi_version(void) { return 1; }
