// Layout changes, bilinear resize, flow-guided feature warps and the multi-frame geometry tensors.
// Feature maps are nhwc (one 32-channel pixel = one 128-byte line), so every bilinear tap of a warp or
// a k-NN gather is one fully used cache line; lanes run over channels first => coalesced 16 B/lane.
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// planar <-> nhwc
// ------------------------------------------------------------------------------------------------
__global__ void pack4_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                             const float* __restrict__ s2, const float* __restrict__ s3, float4* __restrict__ out,
                             long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float4 v;
    v.x = s0 ? s0[i] : 0.f;
    v.y = s1 ? s1[i] : 0.f;
    v.z = s2 ? s2[i] : 0.f;
    v.w = s3 ? s3[i] : 0.f;
    out[i] = v;
  }
}
extern "C" int dis_pack4_nhwc(const float* s0, const float* s1, const float* s2, const float* s3, float* out, int n,
                              int h, int w, void* stream) {
  if (!out) return DIS_ERR_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  long total = (long)n * h * w;
  hipLaunchKernelGGL(pack4_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s0, s1, s2, s3,
                     (float4*)out, total);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// The two-channel image tensor `ir` of the module API is (N,2,H,W): sources 0/1 are its planes, which are
// H*W apart inside one sample but N*... apart between samples; a strided variant handles that case.
__global__ void pack4_strided_kernel(const float* __restrict__ s0, long st0, const float* __restrict__ s1, long st1,
                                     const float* __restrict__ s2, long st2, const float* __restrict__ s3, long st3,
                                     float4* __restrict__ out, int n, long hw) {
  const long total = (long)n * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / hw, p = i - b * hw;
    float4 v;
    v.x = s0 ? s0[b * st0 + p] : 0.f;
    v.y = s1 ? s1[b * st1 + p] : 0.f;
    v.z = s2 ? s2[b * st2 + p] : 0.f;
    v.w = s3 ? s3[b * st3 + p] : 0.f;
    out[i] = v;
  }
}
extern "C" int dis_pack4_nhwc_strided(const float* s0, long st0, const float* s1, long st1, const float* s2,
                                      long st2, const float* s3, long st3, float* out, int n, int h, int w,
                                      void* stream) {
  if (!out) return DIS_ERR_NULL;
  if (n <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  long total = (long)n * h * w;
  hipLaunchKernelGGL(pack4_strided_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, s0,
                     st0, s1, st1, s2, st2, s3, st3, (float4*)out, n, (long)h * w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// tiled transpose of a (rows x cols) matrix per batch item: planar (c x hw) <-> nhwc (hw x c)
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
  __shared__ float tile[32][33];
  const long base = (long)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = x[base + (long)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) y[base + (long)c * rows + r] = tile[tx][j];
  }
}
static int launch_transpose(const float* x, float* y, int batch, int rows, int cols, void* stream) {
  dim3 grid(dis_cdiv(cols, 32), dis_cdiv(rows, 32), batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
// planar -> nhwc for 2 .. 4 channels (the flow fields of the DIS-MF step: 64 x 2 x h x w): one thread per pixel reads its C plane
// values (coalesced per plane) and writes them as one vector.  The 32 x 32 tiles of transpose_kernel hold 2 of 32 rows here
// (46 us per launch at 1.2 TB/s in round 6's profile, two launches per step).
template <int C>
__global__ __launch_bounds__(256) void planar_to_nhwc_small_kernel(const float* __restrict__ x, float* __restrict__ y, long hw,
                                                                    long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    float v[C];
#pragma unroll
    for (int k = 0; k < C; ++k) v[k] = x[(n * C + k) * hw + p];
    if (C == 2) *(float2*)(y + i * 2) = make_float2(v[0], v[1]);
    else if (C == 4) *(float4*)(y + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int k = 0; k < C; ++k) y[i * C + k] = v[k];
    }
  }
}
extern "C" int dis_planar_to_nhwc(const float* x, float* y, int n, int c, int h, int w, void* stream) {
  if (!x || !y) return DIS_ERR_NULL;
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (c >= 2 && c <= 4 && (((uintptr_t)y) & 15) == 0) {
    const long hw = (long)h * w, total = (long)n * hw;
    const dim3 grid(dis_ew_grid(total, 256));
    hipStream_t s = (hipStream_t)stream;
    if (c == 2) hipLaunchKernelGGL(planar_to_nhwc_small_kernel<2>, grid, dim3(256), 0, s, x, y, hw, total);
    else if (c == 3) hipLaunchKernelGGL(planar_to_nhwc_small_kernel<3>, grid, dim3(256), 0, s, x, y, hw, total);
    else hipLaunchKernelGGL(planar_to_nhwc_small_kernel<4>, grid, dim3(256), 0, s, x, y, hw, total);
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  return launch_transpose(x, y, n, c, h * w, stream);
}
extern "C" int dis_nhwc_to_planar(const float* x, float* y, int n, int c, int h, int w, void* stream) {
  if (!x || !y) return DIS_ERR_NULL;
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  return launch_transpose(x, y, n, h * w, c, stream);
}

// ------------------------------------------------------------------------------------------------
// bilinear resize (ATen upsample_bilinear2d semantics)
// ------------------------------------------------------------------------------------------------
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ float resize_scale(int in, int out, int align_corners) {
  if (align_corners) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __forceinline__ Lerp resize_src(int dst, float scale, int in, int align_corners) {
  float src;
  if (align_corners) {
    src = scale * (float)dst;
  } else {
    src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  Lerp L;
  L.i0 = min((int)src, in - 1);
  L.i1 = L.i0 + ((L.i0 < in - 1) ? 1 : 0);
  L.l1 = src - (float)L.i0;
  L.l0 = 1.f - L.l1;
  return L;
}
// One bilinear output value with the roundings of ATen's CPU upsample_bilinear2d (UpSampleKernel.cpp).  The resized
// depth / flow / xyz / mask pyramids feed thresholded masks and Conv3D's neighbour keys (index-class outputs), so the
// value must be the reference's bit for bit.  ATen has two kernels and picks by output size
// (_use_vectorized_kernel_cond_2d: out_h + out_w <= 128): the vectorized one multiplies the two 1-D weights first and
// sums the four taps as fma(d, w11, fma(c, w10, fma(a, w00, b * w01))); the generic one nests
// fma(top, ly0, bot * ly1) with top = fma(a, lx0, b * lx1).  (tests/bitexact.py restates both and is checked against
// torch bit for bit.)  a,b = row i0 at columns i0,i1;  c,d = row i1.
__device__ __forceinline__ float bilerp_aten(float a, float b, float c, float d, const Lerp& Ly, const Lerp& Lx,
                                             bool small) {
  if (small) {
    const float w00 = Ly.l0 * Lx.l0, w01 = Ly.l0 * Lx.l1, w10 = Ly.l1 * Lx.l0, w11 = Ly.l1 * Lx.l1;
    return __fmaf_rn(d, w11, __fmaf_rn(c, w10, __fmaf_rn(a, w00, b * w01)));
  }
  const float top = __fmaf_rn(a, Lx.l0, b * Lx.l1), bot = __fmaf_rn(c, Lx.l0, d * Lx.l1);
  return __fmaf_rn(top, Ly.l0, bot * Ly.l1);
}

// nhwc, channels in groups of VEC floats
template <int VEC>
__global__ void resize_nhwc_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int hin, int win,
                                       int hout, int wout, int c, int ac) {
  const int cg = c / VEC;
  const long total = (long)n * hout * wout * cg;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int ox = (int)(p % wout);
    p /= wout;
    const int oy = (int)(p % hout);
    const int b = (int)(p / hout);
    const Lerp Ly = resize_src(oy, sy, hin, ac), Lx = resize_src(ox, sx, win, ac);
    const float* base = x + (long)b * hin * win * c + g * VEC;
    const float* p00 = base + ((long)Ly.i0 * win + Lx.i0) * c;
    const float* p01 = base + ((long)Ly.i0 * win + Lx.i1) * c;
    const float* p10 = base + ((long)Ly.i1 * win + Lx.i0) * c;
    const float* p11 = base + ((long)Ly.i1 * win + Lx.i1) * c;
    float* o = y + (((long)b * hout + oy) * wout + ox) * c + g * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k)
      o[k] = Ly.l0 * (Lx.l0 * p00[k] + Lx.l1 * p01[k]) + Ly.l1 * (Lx.l0 * p10[k] + Lx.l1 * p11[k]);
  }
}

// destination range whose interpolation can touch source index i
__device__ __forceinline__ void resize_dst_range(int i, float scale, int in, int out, int ac, int* lo, int* hi) {
  if (scale <= 0.f) {
    *lo = 0;
    *hi = out - 1;
    return;
  }
  float a = ((float)i - 1.f), b = ((float)i + 1.f);
  float dlo, dhi;
  if (ac) {
    dlo = a / scale;
    dhi = b / scale;
  } else {
    dlo = (a + 0.5f) / scale - 0.5f;
    dhi = (b + 0.5f) / scale - 0.5f;
  }
  int l = (int)floorf(dlo) - 1, h2 = (int)ceilf(dhi) + 1;
  if (!ac && i == 0) l = 0;  // src clamped at 0
  *lo = max(l, 0);
  *hi = min(h2, out - 1);
}

// backward as a gather over the destination pixels that read source pixel (iy,ix): deterministic, no atomics
template <int VEC>
__global__ void resize_nhwc_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int n, int hin, int win,
                                       int hout, int wout, int c, int ac) {
  const int cg = c / VEC;
  const long total = (long)n * hin * win * cg;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int ix = (int)(p % win);
    p /= win;
    const int iy = (int)(p % hin);
    const int b = (int)(p / hin);
    int ylo, yhi, xlo, xhi;
    resize_dst_range(iy, sy, hin, hout, ac, &ylo, &yhi);
    resize_dst_range(ix, sx, win, wout, ac, &xlo, &xhi);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const Lerp Ly = resize_src(oy, sy, hin, ac);
      const float wy = (Ly.i0 == iy ? Ly.l0 : 0.f) + (Ly.i1 == iy ? Ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const Lerp Lx = resize_src(ox, sx, win, ac);
        const float wx = (Lx.i0 == ix ? Lx.l0 : 0.f) + (Lx.i1 == ix ? Lx.l1 : 0.f);
        if (wx == 0.f) continue;
        const float* gp = gy + (((long)b * hout + oy) * wout + ox) * c + g * VEC;
        const float wgt = wy * wx;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] += wgt * gp[k];
      }
    }
    float* o = gx + (((long)b * hin + iy) * win + ix) * c + g * VEC;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = acc[k];
  }
}

// ---- tiled forms (round 5; c = 4 * CG, 16-byte aligned tensors).  The grid-stride kernels above walk the output in row-major order: the
// four taps of an output row and of the row below it are fetched by workgroups that sit on different XCDs (workgroup i -> XCD i % 8),
// each through its own L2 - 155 MB fetched for 38 MB of source (profiles/r5v5_kernels.md).  Here a workgroup owns an 8-row x (256 / CG)-
// column tile of pixels: its source footprint stays in its CU's L1 / its XCD's L2, loads and stores are 16 B per lane.  Same
// arithmetic, term by term.
#define RZ_ROWS 8
template <int CG>
__global__ __launch_bounds__(256) void resize_nhwc_fwd_tiled_kernel(const float4* __restrict__ x, float4* __restrict__ y, int hin,
                                                                    int win, int hout, int wout, int ac, int tiles_x, int tiles_y) {
  constexpr int PX = 256 / CG;
  int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const int g = threadIdx.x % CG, ox = tx * PX + threadIdx.x / CG;
  if (ox >= wout) return;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  const Lerp Lx = resize_src(ox, sx, win, ac);
  const float4* base = x + (long)b * hin * win * CG + g;
  float4* out = y + (long)b * hout * wout * CG + g;
  const int oy_end = min(ty * RZ_ROWS + RZ_ROWS, hout);
  for (int oy = ty * RZ_ROWS; oy < oy_end; ++oy) {
    const Lerp Ly = resize_src(oy, sy, hin, ac);
    const float4 a = base[((long)Ly.i0 * win + Lx.i0) * CG], bb = base[((long)Ly.i0 * win + Lx.i1) * CG];
    const float4 c4 = base[((long)Ly.i1 * win + Lx.i0) * CG], d = base[((long)Ly.i1 * win + Lx.i1) * CG];
    float4 o;
    o.x = Ly.l0 * (Lx.l0 * a.x + Lx.l1 * bb.x) + Ly.l1 * (Lx.l0 * c4.x + Lx.l1 * d.x);
    o.y = Ly.l0 * (Lx.l0 * a.y + Lx.l1 * bb.y) + Ly.l1 * (Lx.l0 * c4.y + Lx.l1 * d.y);
    o.z = Ly.l0 * (Lx.l0 * a.z + Lx.l1 * bb.z) + Ly.l1 * (Lx.l0 * c4.z + Lx.l1 * d.z);
    o.w = Ly.l0 * (Lx.l0 * a.w + Lx.l1 * bb.w) + Ly.l1 * (Lx.l0 * c4.w + Lx.l1 * d.w);
    out[((long)oy * wout + ox) * CG] = o;
  }
}

// Backward, tiled: a workgroup owns 8 rows x (256 / CG) columns of SOURCE pixels.  The destination rows / columns that read a source
// row / column and their weights are worked out once per workgroup into LDS (first destination index + up to RZ_NZ weights: a 2 x
// align_corners upsample has at most 5), so that a thread's loads are RZ_NZ x RZ_NZ independent 16-byte reads with no index arithmetic
// between them; the sum runs in the order of the gather kernel above (destination rows outer, columns inner).  A scale with more
// than RZ_NZ contributing destination indices per source index (upsampling by more than ~2.5) sets `wide` and the workgroup runs
// the general loop.
#define RZ_NZ 5
struct RzTab {
  int first;
  float w[RZ_NZ];
};
__device__ __forceinline__ float resize_weight_of(int dst, int src_i, float scale, int in, int ac) {
  const Lerp L = resize_src(dst, scale, in, ac);
  return (L.i0 == src_i ? L.l0 : 0.f) + (L.i1 == src_i ? L.l1 : 0.f);
}
template <int CG>
__global__ __launch_bounds__(256) void resize_nhwc_bwd_tiled_kernel(const float4* __restrict__ gy, float4* __restrict__ gx, int hin,
                                                                    int win, int hout, int wout, int ac, int tiles_x, int tiles_y) {
  constexpr int PX = 256 / CG;
  __shared__ RzTab tab[RZ_ROWS + PX];
  __shared__ int wide;
  int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  const int t = threadIdx.x;
  if (t == 0) wide = 0;
  __syncthreads();
  if (t < RZ_ROWS + PX) {
    const bool row = t < RZ_ROWS;
    const int i = row ? ty * RZ_ROWS + t : tx * PX + (t - RZ_ROWS);
    const int in = row ? hin : win, out = row ? hout : wout;
    const float sc = row ? sy : sx;
    RzTab T;
    T.first = 0;
#pragma unroll
    for (int k = 0; k < RZ_NZ; ++k) T.w[k] = 0.f;
    tab[t] = T;
    if (i < in) {
      int lo, hi, first = -1;
      resize_dst_range(i, sc, in, out, ac, &lo, &hi);
      for (int d = lo; d <= hi; ++d) {
        const float wgt = resize_weight_of(d, i, sc, in, ac);
        if (wgt == 0.f) continue;
        if (first < 0) {
          first = d;
          tab[t].first = d;
        }
        if (d - first < RZ_NZ)
          tab[t].w[d - first] = wgt;
        else
          wide = 1;
      }
    }
  }
  __syncthreads();
  const int g = t % CG, px = t / CG, ix = tx * PX + px;
  if (ix >= win) return;
  const float4* src = gy + (long)b * hout * wout * CG + g;
  float4* dst = gx + (long)b * hin * win * CG + g;
  const int iy_end = min(ty * RZ_ROWS + RZ_ROWS, hin);
  if (wide) {   // (block-uniform)
    int xlo, xhi;
    resize_dst_range(ix, sx, win, wout, ac, &xlo, &xhi);
    for (int iy = ty * RZ_ROWS; iy < iy_end; ++iy) {
      int ylo, yhi;
      resize_dst_range(iy, sy, hin, hout, ac, &ylo, &yhi);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = resize_weight_of(oy, iy, sy, hin, ac);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = resize_weight_of(ox, ix, sx, win, ac);
          if (wx == 0.f) continue;
          const float4 v = src[((long)oy * wout + ox) * CG];
          const float wgt = wy * wx;
          acc.x += wgt * v.x, acc.y += wgt * v.y, acc.z += wgt * v.z, acc.w += wgt * v.w;
        }
      }
      dst[((long)iy * win + ix) * CG] = acc;
    }
    return;
  }
  const RzTab X = tab[RZ_ROWS + px];
  for (int iy = ty * RZ_ROWS; iy < iy_end; ++iy) {
    RzTab Y = tab[iy - ty * RZ_ROWS];   // (the same for every thread of the workgroup: scalar registers, scalar branches below)
    Y.first = __builtin_amdgcn_readfirstlane(Y.first);
#pragma unroll
    for (int r = 0; r < RZ_NZ; ++r) Y.w[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Y.w[r])));
    float4 v[RZ_NZ][RZ_NZ];
#pragma unroll
    for (int r = 0; r < RZ_NZ; ++r) {
      if (Y.w[r] != 0.f) {
#pragma unroll
        for (int k = 0; k < RZ_NZ; ++k) {
          // (a zero weight: the slot lies outside the contributing range - possibly outside the tensor; read the first slot instead)
          const int ox = X.w[k] != 0.f ? X.first + k : X.first;
          v[r][k] = src[((long)(Y.first + r) * wout + ox) * CG];
        }
      } else {
#pragma unroll
        for (int k = 0; k < RZ_NZ; ++k) v[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < RZ_NZ; ++r)
#pragma unroll
      for (int k = 0; k < RZ_NZ; ++k) {
        if (Y.w[r] != 0.f && X.w[k] != 0.f) {
          const float wgt = Y.w[r] * X.w[k];
          acc.x += wgt * v[r][k].x, acc.y += wgt * v[r][k].y, acc.z += wgt * v[r][k].z, acc.w += wgt * v[r][k].w;
        }
      }
    dst[((long)iy * win + ix) * CG] = acc;
  }
}
static inline bool rz_aligned(const void* a, const void* b) { return ((((size_t)a) | ((size_t)b)) & 15) == 0; }

extern "C" int dis_resize_bilinear_nhwc_fwd(const float* x, float* y, int n, int hin, int win, int hout, int wout,
                                            int c, int align_corners, void* stream) {
  if (!x || !y) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if ((c == 32 || c == 16) && rz_aligned(x, y)) {
    const int px = 256 / (c / 4), tiles_x = dis_cdiv(wout, px), tiles_y = dis_cdiv(hout, RZ_ROWS);
    const long tiles = (long)n * tiles_x * tiles_y;
    if (tiles <= 0x7fffffffL) {
      if (c == 32)
        hipLaunchKernelGGL(resize_nhwc_fwd_tiled_kernel<8>, dim3((unsigned)tiles), dim3(256), 0, s, (const float4*)x, (float4*)y, hin,
                           win, hout, wout, align_corners, tiles_x, tiles_y);
      else
        hipLaunchKernelGGL(resize_nhwc_fwd_tiled_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, s, (const float4*)x, (float4*)y, hin,
                           win, hout, wout, align_corners, tiles_x, tiles_y);
      DIS_CHECK_LAUNCH();
      return DIS_OK;
    }
  }
  if (c % 4 == 0) {
    long total = (long)n * hout * wout * (c / 4);
    hipLaunchKernelGGL(resize_nhwc_fwd_kernel<4>, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, x, y, n, hin, win,
                       hout, wout, c, align_corners);
  } else {
    long total = (long)n * hout * wout * c;
    hipLaunchKernelGGL(resize_nhwc_fwd_kernel<1>, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, x, y, n, hin, win,
                       hout, wout, c, align_corners);
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_resize_bilinear_nhwc_bwd(const float* gy, float* gx, int n, int hin, int win, int hout, int wout,
                                            int c, int align_corners, void* stream) {
  if (!gy || !gx) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if ((c == 32 || c == 16) && rz_aligned(gy, gx)) {
    const int px = 256 / (c / 4), tiles_x = dis_cdiv(win, px), tiles_y = dis_cdiv(hin, RZ_ROWS);
    const long tiles = (long)n * tiles_x * tiles_y;
    if (tiles <= 0x7fffffffL) {
      if (c == 32)
        hipLaunchKernelGGL(resize_nhwc_bwd_tiled_kernel<8>, dim3((unsigned)tiles), dim3(256), 0, s, (const float4*)gy, (float4*)gx,
                           hin, win, hout, wout, align_corners, tiles_x, tiles_y);
      else
        hipLaunchKernelGGL(resize_nhwc_bwd_tiled_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, s, (const float4*)gy, (float4*)gx,
                           hin, win, hout, wout, align_corners, tiles_x, tiles_y);
      DIS_CHECK_LAUNCH();
      return DIS_OK;
    }
  }
  if (c % 4 == 0) {
    long total = (long)n * hin * win * (c / 4);
    hipLaunchKernelGGL(resize_nhwc_bwd_kernel<4>, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, gy, gx, n, hin,
                       win, hout, wout, c, align_corners);
  } else {
    long total = (long)n * hin * win * c;
    hipLaunchKernelGGL(resize_nhwc_bwd_kernel<1>, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, gy, gx, n, hin,
                       win, hout, wout, c, align_corners);
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// planar: nc independent planes; optional per-channel scale (resize_flow_like): plane index % c_for_scale
// selects scale0 (channel 0) / scale1 (channel 1).
__global__ void resize_planar_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int nc, int hin,
                                         int win, int hout, int wout, int ac, float scale0, float scale1,
                                         int c_for_scale) {
  const long total = (long)nc * hout * wout;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  const bool small = hout + wout <= 128;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % wout);
    long p = i / wout;
    const int oy = (int)(p % hout);
    const int pl = (int)(p / hout);
    const Lerp Ly = resize_src(oy, sy, hin, ac), Lx = resize_src(ox, sx, win, ac);
    const float* base = x + (long)pl * hin * win;
    float v = bilerp_aten(base[(long)Ly.i0 * win + Lx.i0], base[(long)Ly.i0 * win + Lx.i1],
                          base[(long)Ly.i1 * win + Lx.i0], base[(long)Ly.i1 * win + Lx.i1], Ly, Lx, small);
    if (c_for_scale > 0) v *= ((pl % c_for_scale) == 0) ? scale0 : scale1;
    y[i] = v;
  }
}
__global__ void resize_planar_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int nc, int hin,
                                         int win, int hout, int wout, int ac) {
  const long total = (long)nc * hin * win;
  const float sy = resize_scale(hin, hout, ac), sx = resize_scale(win, wout, ac);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % win);
    long p = i / win;
    const int iy = (int)(p % hin);
    const int pl = (int)(p / hin);
    int ylo, yhi, xlo, xhi;
    resize_dst_range(iy, sy, hin, hout, ac, &ylo, &yhi);
    resize_dst_range(ix, sx, win, wout, ac, &xlo, &xhi);
    float acc = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const Lerp Ly = resize_src(oy, sy, hin, ac);
      const float wy = (Ly.i0 == iy ? Ly.l0 : 0.f) + (Ly.i1 == iy ? Ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const Lerp Lx = resize_src(ox, sx, win, ac);
        const float wx = (Lx.i0 == ix ? Lx.l0 : 0.f) + (Lx.i1 == ix ? Lx.l1 : 0.f);
        acc += wy * wx * gy[((long)pl * hout + oy) * wout + ox];
      }
    }
    gx[i] = acc;
  }
}
extern "C" int dis_resize_bilinear_planar_fwd(const float* x, float* y, int nc, int hin, int win, int hout, int wout,
                                              int align_corners, float scale0, float scale1, int c_for_scale,
                                              void* stream) {
  if (!x || !y) return DIS_ERR_NULL;
  if (nc <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  long total = (long)nc * hout * wout;
  hipLaunchKernelGGL(resize_planar_fwd_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     nc, hin, win, hout, wout, align_corners, scale0, scale1, c_for_scale);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_resize_bilinear_planar_bwd(const float* gy, float* gx, int nc, int hin, int win, int hout,
                                              int wout, int align_corners, void* stream) {
  if (!gy || !gx) return DIS_ERR_NULL;
  if (nc <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  long total = (long)nc * hin * win;
  hipLaunchKernelGGL(resize_planar_bwd_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, gy,
                     gx, nc, hin, win, hout, wout, align_corners);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// bilinear tap set with zeros padding (shared by the warps)
// ------------------------------------------------------------------------------------------------
struct Taps {
  int x0, y0;
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
};
__device__ __forceinline__ Taps make_taps(float px, float py, int h, int w) {
  Taps t;
  float ix = gs_roundtrip(px, w), iy = gs_roundtrip(py, h);
  ix = fminf(fmaxf(ix, -2.f), (float)w + 2.f);
  iy = fminf(fmaxf(iy, -2.f), (float)h + 2.f);
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  const float wx = ix - fx, ex = 1.f - wx, wy = iy - fy, ey = 1.f - wy;
  t.w00 = ey * ex; t.w01 = ey * wx; t.w10 = wy * ex; t.w11 = wy * wx;
  const bool xa = t.x0 >= 0 && t.x0 < w, xb = t.x0 + 1 >= 0 && t.x0 + 1 < w;
  const bool ya = t.y0 >= 0 && t.y0 < h, yb = t.y0 + 1 >= 0 && t.y0 + 1 < h;
  t.v00 = xa && ya; t.v01 = xb && ya; t.v10 = xa && yb; t.v11 = xb && yb;
  return t;
}

// frame index held by slot s of target t: slot 0 = t, then the other frames in increasing order
__device__ __forceinline__ int slot_frame(int t, int s) { return s == 0 ? t : (s - 1 < t ? s - 1 : s); }

// ------------------------------------------------------------------------------------------------
// gather_warped_feat (reference multi_frame_networks.py:347-360, warp :83-99), all targets at once
//   feat (tl,bs,h,w,c)  flows (tl*tl,bs,h,w,2)  ->  out (tl,bs,h,w,tl,c)
// one thread = 4 channels of one (target, sample, pixel, slot)
// ------------------------------------------------------------------------------------------------
typedef float lo_v4f __attribute__((ext_vector_type(4)));
__global__ void gather_warped_feat_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ flows,
                                              float* __restrict__ out, int tl, int bs, int h, int w, int c, int nt) {
  const int cg = c >> 2;
  const long hw = (long)h * w;
  const long total = (long)tl * bs * hw * tl * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long r = i / cg;
    const int s = (int)(r % tl);
    r /= tl;
    const long p = r % hw;
    r /= hw;
    const int b = (int)(r % bs);
    const int t = (int)(r / bs);
    const int j = slot_frame(t, s);
    const float* src = feat + ((long)j * bs + b) * hw * c + g * 4;
    float4 v;
    if (s == 0) {
      v = *(const float4*)(src + p * c);
    } else {
      const int y = (int)(p / w), x = (int)(p - (long)y * w);
      const float2 f = *(const float2*)(flows + ((((long)t * tl + j) * bs + b) * hw + p) * 2);
      const Taps tp = make_taps(f.x + (float)x, f.y + (float)y, h, w);
      // taps outside the image: clamped address and zero weight (0 * finite = 0 exactly, so the sum is the one over
      // the valid taps) - a `valid ? load : zero` select turns into a select of POINTERS with the zero in scratch
      const int xa = min(max(tp.x0, 0), w - 1), xb = min(max(tp.x0 + 1, 0), w - 1);
      const int ya = min(max(tp.y0, 0), h - 1), yb = min(max(tp.y0 + 1, 0), h - 1);
      const float w00 = tp.v00 ? tp.w00 : 0.f, w01 = tp.v01 ? tp.w01 : 0.f;
      const float w10 = tp.v10 ? tp.w10 : 0.f, w11 = tp.v11 ? tp.w11 : 0.f;
      const float4 a = *(const float4*)(src + ((long)ya * w + xa) * c);
      const float4 bq = *(const float4*)(src + ((long)ya * w + xb) * c);
      const float4 cq = *(const float4*)(src + ((long)yb * w + xa) * c);
      const float4 d = *(const float4*)(src + ((long)yb * w + xb) * c);
      v.x = a.x * w00 + bq.x * w01 + cq.x * w10 + d.x * w11;
      v.y = a.y * w00 + bq.y * w01 + cq.y * w10 + d.y * w11;
      v.z = a.z * w00 + bq.z * w01 + cq.z * w10 + d.z * w11;
      v.w = a.w * w00 + bq.w * w01 + cq.w * w10 + d.w * w11;
    }
    // (the 4-slot output is written once and read much later: a non-temporal store leaves the caches to the bilinear taps)
    float* op = out + ((((long)t * bs + b) * hw + p) * tl + s) * c + g * 4;
    if (nt) __builtin_nontemporal_store((lo_v4f){v.x, v.y, v.z, v.w}, (lo_v4f*)op);
    else *(float4*)op = v;
  }
}

// Tiled form (round 5).  The grid-stride kernel above walks (pixel, slot) in row-major order, 8 pixels per workgroup: the two source rows
// an output row reads are fetched again for the next output row by a workgroup on another XCD, and again by each of the three other
// targets that warp the same frame - ~840 MB leave the L2s for a 113 MB feature tensor at core resolution (profiles/r5v5_kernels.md).
// Here a workgroup owns a GT_TH x GT_TW tile of pixels of ONE target (8-pixel column strips, walked downwards: the lower source row of
// one pass is the upper one of the next), and the blockIdx -> (tile, target) map puts the tl targets of a tile on the SAME XCD one
// after the other (workgroups go to XCD blockIdx % 8): their warped reads of the shared source frames meet in that XCD's L2.
// Arithmetic: the kernel above, term by term.
#define GT_TH 8
#define GT_TW 32
#define GC_E 12   // CSR backward: entries of a destination fetched together
__global__ __launch_bounds__(256) void gather_warped_feat_fwd_tiled_kernel(const float* __restrict__ feat, const float* __restrict__ flows,
                                                                           float* __restrict__ out, int tl, int bs, int h, int w, int c,
                                                                           int nt, int tiles_x, int tiles_y, int ntile) {
  const int cg = c >> 2, tpp = tl * cg, pxp = 256 / tpp;
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int t = k % tl, tile_lin = (k / tl) * 8 + xcd;
  if (tile_lin >= ntile) return;
  const int tx = tile_lin % tiles_x, ty = (tile_lin / tiles_x) % tiles_y, b = tile_lin / (tiles_x * tiles_y);
  const int g = threadIdx.x % cg, s = (threadIdx.x / cg) % tl, pi = threadIdx.x / tpp;
  const long hw = (long)h * w;
  const int j = slot_frame(t, s);
  const float* src = feat + ((long)j * bs + b) * hw * c + g * 4;
  const float* fl = flows + (((long)t * tl + j) * bs + b) * hw * 2;
  float* ob = out + ((long)t * bs + b) * hw * tl * c + (long)s * c + g * 4;
  const int y0 = ty * GT_TH, nrow = min(GT_TH, h - y0);
  for (int x0 = tx * GT_TW; x0 < min(tx * GT_TW + GT_TW, w); x0 += pxp) {
    const int x = x0 + pi;
    if (x >= w) continue;
    // the flows of the strip's rows first (one round trip), then the four taps of four rows in flight together: a thread's chain is
    // 3 dependent memory round trips per strip instead of 2 per row
    float2 f[GT_TH];
#pragma unroll
    for (int r = 0; r < GT_TH; ++r) {
      const int y = min(y0 + r, h - 1);
      f[r] = s == 0 ? make_float2(0.f, 0.f) : *(const float2*)(fl + ((long)y * w + x) * 2);
    }
#pragma unroll
    for (int r0 = 0; r0 < GT_TH; r0 += 4) {
      float4 ta[4], tb[4], tc[4], td[4];
      float w00[4], w01[4], w10[4], w11[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int y = min(y0 + r0 + u, h - 1);
        if (s == 0) {
          ta[u] = *(const float4*)(src + ((long)y * w + x) * c);
        } else {
          const Taps tp = make_taps(f[r0 + u].x + (float)x, f[r0 + u].y + (float)y, h, w);
          const int xa = min(max(tp.x0, 0), w - 1), xb = min(max(tp.x0 + 1, 0), w - 1);
          const int ya = min(max(tp.y0, 0), h - 1), yb = min(max(tp.y0 + 1, 0), h - 1);
          w00[u] = tp.v00 ? tp.w00 : 0.f, w01[u] = tp.v01 ? tp.w01 : 0.f;
          w10[u] = tp.v10 ? tp.w10 : 0.f, w11[u] = tp.v11 ? tp.w11 : 0.f;
          ta[u] = *(const float4*)(src + ((long)ya * w + xa) * c);
          tb[u] = *(const float4*)(src + ((long)ya * w + xb) * c);
          tc[u] = *(const float4*)(src + ((long)yb * w + xa) * c);
          td[u] = *(const float4*)(src + ((long)yb * w + xb) * c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r0 + u >= nrow) break;
        float4 v;
        if (s == 0) {
          v = ta[u];
        } else {
          v.x = ta[u].x * w00[u] + tb[u].x * w01[u] + tc[u].x * w10[u] + td[u].x * w11[u];
          v.y = ta[u].y * w00[u] + tb[u].y * w01[u] + tc[u].y * w10[u] + td[u].y * w11[u];
          v.z = ta[u].z * w00[u] + tb[u].z * w01[u] + tc[u].z * w10[u] + td[u].z * w11[u];
          v.w = ta[u].w * w00[u] + tb[u].w * w01[u] + tc[u].w * w10[u] + td[u].w * w11[u];
        }
        float* op = ob + ((long)(y0 + r0 + u) * w + x) * tl * c;
        if (nt) __builtin_nontemporal_store((lo_v4f){v.x, v.y, v.z, v.w}, (lo_v4f*)op);
        else *(float4*)op = v;
      }
    }
  }
}
static inline bool gather_tiled_on() {   // DIS_GATHER_TILED=0: the grid-stride kernels (read per call: tests compare the two forms)
  const char* e = getenv("DIS_GATHER_TILED");
  return !(e && e[0] == '0');
}

// Backward.  Phase 1 (plain stores, also replaces a memset): grad_feat[t] = grad_out[t, slot 0].
// Phase 2 (float atomics): the warped slots scatter through their 4 bilinear taps.  One LANE = one channel,
// so a wave-instruction's 64 atomics are two contiguous 128-byte rows: the shape the memory-side atomic unit
// runs at full rate (MI355X_MICROARCH.md, "Global float atomics").
__global__ void gather_warped_feat_bwd_own_kernel(const float* __restrict__ gout, float* __restrict__ gfeat,
                                                  long pixels, int tl, int c) {
  const int cg = c >> 2;
  const long total = pixels * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long p = i / cg;
    *(float4*)(gfeat + p * c + g * 4) = *(const float4*)(gout + (p * tl) * c + g * 4);
  }
}

__global__ void gather_warped_feat_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ flows,
                                              float* __restrict__ gfeat, int tl, int bs, int h, int w, int c) {
  const long hw = (long)h * w;
  const int ns = tl - 1;  // warped slots
  const long total = (long)tl * bs * hw * ns * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int s = (int)(r % ns) + 1;
    r /= ns;
    const long p = r % hw;
    r /= hw;
    const int b = (int)(r % bs);
    const int t = (int)(r / bs);
    const int j = slot_frame(t, s);
    const float gv = gout[((((long)t * bs + b) * hw + p) * tl + s) * c + ch];
    float* dst = gfeat + ((long)j * bs + b) * hw * c + ch;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float2 f = *(const float2*)(flows + ((((long)t * tl + j) * bs + b) * hw + p) * 2);
    const Taps tp = make_taps(f.x + (float)x, f.y + (float)y, h, w);
    if (tp.v00) atomicAdd(dst + ((long)tp.y0 * w + tp.x0) * c, gv * tp.w00);
    if (tp.v01) atomicAdd(dst + ((long)tp.y0 * w + tp.x0 + 1) * c, gv * tp.w01);
    if (tp.v10) atomicAdd(dst + ((long)(tp.y0 + 1) * w + tp.x0) * c, gv * tp.w10);
    if (tp.v11) atomicAdd(dst + ((long)(tp.y0 + 1) * w + tp.x0 + 1) * c, gv * tp.w11);
  }
}

extern "C" int dis_gather_warped_feat_fwd(const float* feat, const float* flows, float* out, int tl, int bs, int h,
                                          int w, int c, void* stream) {
  if (!feat || !flows || !out) return DIS_ERR_NULL;
  if (tl <= 0 || bs <= 0 || h <= 1 || w <= 1 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0) return DIS_ERR_UNSUPPORTED;
  long total = (long)tl * bs * h * w * tl * (c / 4);
  static const int nt = getenv("DIS_GATHER_NT") ? atoi(getenv("DIS_GATHER_NT")) : 1;
  const int tpp = tl * (c / 4);
  if (tpp <= 256 && 256 % tpp == 0 && GT_TW % (256 / tpp) == 0 && gather_tiled_on()) {
    const int tiles_x = dis_cdiv(w, GT_TW), tiles_y = dis_cdiv(h, GT_TH);
    const long ntile = (long)bs * tiles_x * tiles_y, blocks = (ntile + 7) / 8 * 8 * tl;
    if (blocks <= 0x7fffffffL) {
      hipLaunchKernelGGL(gather_warped_feat_fwd_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feat, flows,
                         out, tl, bs, h, w, c, nt, tiles_x, tiles_y, (int)ntile);
      DIS_CHECK_LAUNCH();
      return DIS_OK;
    }
  }
  hipLaunchKernelGGL(gather_warped_feat_fwd_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     feat, flows, out, tl, bs, h, w, c, nt);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_gather_warped_feat_bwd(const float* grad_out, const float* flows, float* grad_feat, int tl, int bs,
                                          int h, int w, int c, void* stream) {
  if (!grad_out || !flows || !grad_feat) return DIS_ERR_NULL;
  if (tl <= 0 || bs <= 0 || h <= 1 || w <= 1 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0) return DIS_ERR_UNSUPPORTED;
  const long pixels = (long)tl * bs * h * w;
  hipLaunchKernelGGL(gather_warped_feat_bwd_own_kernel, dim3(dis_ew_grid(pixels * (c / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_out, grad_feat, pixels, tl, c);
  const long total = pixels * (tl - 1) * c;
  int grid = dis_cdiv(total, 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(gather_warped_feat_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, flows,
                     grad_feat, tl, bs, h, w, c);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// Scatter index of gather_warped_feat, built once per step and shared by every backward call at that resolution
// (the flows are data: all four Block2D3D gather the same pattern).  CSR by DESTINATION pixel d = (frame j, b, q):
// entries (source row of grad_out, bilinear weight).  With it the backward is a gather of whole 128-byte rows with
// plain loads/stores (HBM/L2 rate) instead of 12 float-atomic rows per pixel (memory-side atomic rate, ~1.3 TB/s),
// and, the lists being sorted by source row, it is bitwise reproducible.
//   csr (int32 words): [0, nd]          offsets (nd+1)
//                      [nd+1, 2nd+1)    cursor  (scratch of the build)
//                      [2nd+1, ...)     entries: 2 words each (source row, weight bits), nd*... <= pixels*(tl-1)*4
// ------------------------------------------------------------------------------------------------
#define CSR_SCAN_ELEMS 2048
__global__ void csr_zero_kernel(int* __restrict__ p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0;
}
__global__ void csr_count_fill_kernel(const float* __restrict__ flows, int* __restrict__ cursor,
                                      int* __restrict__ entries, int tl, int bs, int h, int w, int fill) {
  const long hw = (long)h * w;
  const int ns = tl - 1;
  const long total = (long)tl * bs * hw * ns;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i % ns) + 1;
    long r = i / ns;
    const long p = r % hw;
    r /= hw;
    const int b = (int)(r % bs);
    const int t = (int)(r / bs);
    const int j = slot_frame(t, s);
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float2 f = *(const float2*)(flows + ((((long)t * tl + j) * bs + b) * hw + p) * 2);
    const Taps tp = make_taps(f.x + (float)x, f.y + (float)y, h, w);
    const int row = (int)(((((long)t * bs + b) * hw + p) * tl) + s);
    const long dbase = ((long)j * bs + b) * hw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool valid = (k == 0) ? tp.v00 : (k == 1) ? tp.v01 : (k == 2) ? tp.v10 : tp.v11;
      if (!valid) continue;
      const float wgt = (k == 0) ? tp.w00 : (k == 1) ? tp.w01 : (k == 2) ? tp.w10 : tp.w11;
      const long d = dbase + (long)(tp.y0 + (k >> 1)) * w + tp.x0 + (k & 1);
      const int pos = atomicAdd(cursor + d, 1);
      if (fill) {
        entries[2 * (long)pos] = row;
        entries[2 * (long)pos + 1] = __float_as_int(wgt);
      }
    }
  }
}
// exclusive scan of cnt[0..n) in three passes (block sums, scan of the sums by one block, block rescans)
__global__ __launch_bounds__(256) void csr_scan1_kernel(const int* __restrict__ cnt, int* __restrict__ bsum, long n) {
  __shared__ int sm[4];
  const long base = (long)blockIdx.x * CSR_SCAN_ELEMS;
  int s = 0;
  for (int k = 0; k < CSR_SCAN_ELEMS / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i < n) s += cnt[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
// exclusive prefix of the block sums in place, bsum[nblk] = total.  ONE wave (launched with 64 threads): every lane scans a contiguous
// chunk, the 64 chunk totals are scanned with shuffles (integers: the order of the additions does not matter).  The one-thread loop
// this replaces was 28 us of dependent loads, twice per step.
__global__ void csr_scan2_kernel(int* __restrict__ bsum, int nblk) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const int lane = threadIdx.x, per = (nblk + 63) / 64;
  const int lo = lane * per < nblk ? lane * per : nblk, hi = lo + per < nblk ? lo + per : nblk;
  int tot = 0;
  for (int k = lo; k < hi; ++k) tot += bsum[k];
  int inc = tot;   // inclusive scan over the lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  int run = inc - tot;
  for (int k = lo; k < hi; ++k) {
    const int v = bsum[k];
    bsum[k] = run;
    run += v;
  }
  if (lane == 63) bsum[nblk] = inc;
}
// offsets[i] = exclusive prefix; cursor[i] = offsets[i] (start position for the fill pass); offsets[n] = total
__global__ __launch_bounds__(256) void csr_scan3_kernel(int* __restrict__ cursor, int* __restrict__ offsets,
                                                         const int* __restrict__ bsum, long n, int nblk) {
  __shared__ int sm[256];
  const long base = (long)blockIdx.x * CSR_SCAN_ELEMS + (long)threadIdx.x * (CSR_SCAN_ELEMS / 256);
  int v[CSR_SCAN_ELEMS / 256], s = 0;
#pragma unroll
  for (int k = 0; k < CSR_SCAN_ELEMS / 256; ++k) {
    v[k] = (base + k < n) ? cursor[base + k] : 0;
    s += v[k];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  int pre = bsum[blockIdx.x];
  for (int k = 0; k < (int)threadIdx.x; ++k) pre += sm[k];
#pragma unroll
  for (int k = 0; k < CSR_SCAN_ELEMS / 256; ++k) {
    if (base + k < n) {
      offsets[base + k] = pre;
      cursor[base + k] = pre;
    }
    pre += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = bsum[nblk];
}
// sort every destination's (short) list by source row: fixed summation order in the gather.  Lists of up to 16
// entries (all of them in practice: 3 source frames x 4 bilinear corners, plus overlaps) are sorted in registers by
// rank - every entry is read once and written once, at list start + number of entries that sort before it (rows are
// unique within a list up to exact duplicates, which the index tie-break orders); longer lists fall back to an
// insertion sort in memory.
#define CSR_SORT_MAX 16
__global__ void csr_sort_kernel(const int* __restrict__ offsets, int* __restrict__ entries, long nd) {
  for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < nd; d += (long)gridDim.x * blockDim.x) {
    const int lo = offsets[d], hi = offsets[d + 1];
    const int len = hi - lo;
    if (len <= 1) continue;
    if (len <= CSR_SORT_MAX) {
      int2 e[CSR_SORT_MAX];
      const int2* src = (const int2*)entries + lo;
#pragma unroll
      for (int k = 0; k < CSR_SORT_MAX; ++k) e[k] = k < len ? src[k] : make_int2(0x7fffffff, 0);
      int2* dst = (int2*)entries + lo;
#pragma unroll
      for (int k = 0; k < CSR_SORT_MAX; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < CSR_SORT_MAX; ++j) rank += (e[j].x < e[k].x || (e[j].x == e[k].x && j < k)) ? 1 : 0;
        if (k < len) dst[rank] = e[k];
      }
    } else {
      for (int a = lo + 1; a < hi; ++a) {
        const int kr = entries[2 * (long)a], kw = entries[2 * (long)a + 1];
        int bpos = a - 1;
        while (bpos >= lo && entries[2 * (long)bpos] > kr) {
          entries[2 * (long)bpos + 2] = entries[2 * (long)bpos];
          entries[2 * (long)bpos + 3] = entries[2 * (long)bpos + 1];
          --bpos;
        }
        entries[2 * (long)bpos + 2] = kr;
        entries[2 * (long)bpos + 3] = kw;
      }
    }
  }
}
// grad_feat[d] = grad_out[d, slot 0] + sum_e w_e * grad_out[row_e];  one thread = 4 channels of one destination
__global__ void gather_warped_feat_bwd_csr_kernel(const float* __restrict__ gout, const int* __restrict__ offsets,
                                                  const int* __restrict__ entries, const float* __restrict__ init,
                                                  float* __restrict__ gfeat, long nd, int tl, int c) {
  const int cg = c >> 2;
  const long total = nd * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long d = i / cg;
    float4 acc = *(const float4*)(gout + (d * tl) * c + g * 4);
    if (init) {
      const float4 q = *(const float4*)(init + d * c + g * 4);
      acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
    const int lo = offsets[d], hi = offsets[d + 1];
    for (int e = lo; e < hi; ++e) {
      const int row = entries[2 * (long)e];
      const float wgt = __int_as_float(entries[2 * (long)e + 1]);
      const float4 v = *(const float4*)(gout + (long)row * c + g * 4);
      acc.x += v.x * wgt; acc.y += v.y * wgt; acc.z += v.z * wgt; acc.w += v.w * wgt;
    }
    *(float4*)(gfeat + d * c + g * 4) = acc;
  }
}

// Tiled form (round 5): a workgroup owns a GT_TH x GT_TW tile of destination pixels of one (frame, sample) (a row segment of 256 / cg
// pixels per pass); consecutive tiles of a frame go to the same XCD (its share of the tiles is one contiguous range), so the rows of
// grad_out that neighbouring destinations share are found in that XCD's L2; four entries and their rows in flight per round instead
// of one.  The sum runs in entry order: bit for bit the kernel above.
// GNRES (round 6): feat IS y = SELU(GroupNorm(x2) + residual) (a Block2D3D / ResNetBlock output) and this launch completes the
// gradient wrt y - the epilogue of dis_gn_bwd_res_sums rides here: what is stored is gres = g SELU'(y) (the pre-activation gradient,
// which doubles as the residual gradient) and the workgroup leaves A_c = sum gres, B_c = sum gres x2 of its tile in slot
// (tile within the image) mod slots of ab (n, slots, 2, c) doubles - the pass that read g, y, x2 again and wrote gres is gone.
struct GatherGnRes {
  const float* y;     // the gather's input (values), shaped like grad_feat
  const float* x2;    // the GroupNorm's input
  double* ab;         // (n, slots, 2, c), zeroed by the caller
  int slots, act, atomic;   // atomic: more tiles per image than slots - several workgroups add to one slot
};
template <bool GNRES>
__global__ __launch_bounds__(256) void gather_warped_feat_bwd_csr_tiled_kernel(const float* __restrict__ gout, const int* __restrict__ offsets,
                                                                               const int* __restrict__ entries,
                                                                               const float* __restrict__ init, float* __restrict__ gfeat,
                                                                               int h, int w, int tl, int c, int tiles_x, int tiles_y,
                                                                               int ntile, int per_xcd, GatherGnRes gr) {
  __shared__ float gsa[GNRES ? 256 : 1][4], gsb[GNRES ? 256 : 1][4];
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  const int cg = c >> 2, pxp = 256 / cg;
  const int tile_lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || tile_lin >= ntile) return;
  const int tx = tile_lin % tiles_x, ty = (tile_lin / tiles_x) % tiles_y;
  const long fb = tile_lin / (tiles_x * tiles_y);   // frame * bs + sample
  const int g = threadIdx.x % cg, pi = threadIdx.x / cg;
  const int y0 = ty * GT_TH, nrow = min(GT_TH, h - y0);
  for (int x0 = tx * GT_TW; x0 < min(tx * GT_TW + GT_TW, w); x0 += pxp) {
    const int x = x0 + pi;
    if (x >= w) continue;
    // offsets of the strip's rows first; per destination: its first GC_E entries (an interior pixel has (tl - 1) * 4 = 12) in one
    // round trip, their rows in a second one - the chain was offsets -> entry -> row per entry.  Sum in entry order.
    int lo[GT_TH], hi[GT_TH];
#pragma unroll
    for (int r = 0; r < GT_TH; ++r) {
      const long d = (fb * h + min(y0 + r, h - 1)) * w + x;
      lo[r] = offsets[d], hi[r] = offsets[d + 1];
    }
#pragma unroll 1
    for (int r = 0; r < nrow; ++r) {
      const long d = (fb * h + y0 + r) * w + x;
      int e0 = lo[0], e1 = hi[0];
#pragma unroll
      for (int q = 1; q < GT_TH; ++q)
        if (q == r) e0 = lo[q], e1 = hi[q];   // (register array, dynamic row: selects, no scratch)
      float4 acc = *(const float4*)(gout + (d * tl) * c + g * 4);
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (init) q4 = *(const float4*)(init + d * c + g * 4);
      float4 yv = make_float4(0.f, 0.f, 0.f, 0.f), xv = yv;
      if (GNRES) {   // (requested with the row's own operands: in flight under the entry -> row chain below)
        yv = *(const float4*)(gr.y + d * c + g * 4);
        xv = *(const float4*)(gr.x2 + d * c + g * 4);
      }
      int row[GC_E];
      float wgt[GC_E];
      float4 v[GC_E];
#pragma unroll
      for (int u = 0; u < GC_E; ++u) {
        const bool on = e0 + u < e1;
        const long e = on ? e0 + u : 0;   // (past the list: entry 0 is read and dropped, row 0 of grad_out stands in - never summed)
        const int rw = entries[2 * e], wb = entries[2 * e + 1];
        row[u] = on ? rw : 0;
        wgt[u] = on ? __int_as_float(wb) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < GC_E; ++u) v[u] = *(const float4*)(gout + (long)row[u] * c + g * 4);
      if (init) { acc.x += q4.x; acc.y += q4.y; acc.z += q4.z; acc.w += q4.w; }
#pragma unroll
      for (int u = 0; u < GC_E; ++u) {
        if (e0 + u < e1) {
          acc.x += v[u].x * wgt[u]; acc.y += v[u].y * wgt[u]; acc.z += v[u].z * wgt[u]; acc.w += v[u].w * wgt[u];
        }
      }
      for (int e = e0 + GC_E; e < e1; ++e) {
        const float wg = __int_as_float(entries[2 * (long)e + 1]);
        const float4 vv = *(const float4*)(gout + (long)entries[2 * (long)e] * c + g * 4);
        acc.x += vv.x * wg; acc.y += vv.y * wg; acc.z += vv.z * wg; acc.w += vv.w * wg;
      }
      if (GNRES) {   // (gn_res_sums_kernel's arithmetic per element)
        acc.x *= act_grad_from_out(yv.x, gr.act), acc.y *= act_grad_from_out(yv.y, gr.act);
        acc.z *= act_grad_from_out(yv.z, gr.act), acc.w *= act_grad_from_out(yv.w, gr.act);
        sa[0] += acc.x, sa[1] += acc.y, sa[2] += acc.z, sa[3] += acc.w;
        sb[0] = __builtin_fmaf(acc.x, xv.x, sb[0]), sb[1] = __builtin_fmaf(acc.y, xv.y, sb[1]);
        sb[2] = __builtin_fmaf(acc.z, xv.z, sb[2]), sb[3] = __builtin_fmaf(acc.w, xv.w, sb[3]);
      }
      *(float4*)(gfeat + d * c + g * 4) = acc;
    }
  }
  if (GNRES) {   // the tile's channel sums: the 256 / cg threads of a channel group in thread order (fixed), then one slot
#pragma unroll
    for (int k = 0; k < 4; ++k) gsa[threadIdx.x][k] = sa[k], gsb[threadIdx.x][k] = sb[k];
    __syncthreads();
    if ((int)threadIdx.x < 2 * c) {
      const int t = threadIdx.x, ch = t < c ? t : t - c, grp = ch >> 2, k = ch & 3;
      double v = 0.0;
      for (int u = grp; u < 256; u += cg) v += (double)(t < c ? gsa[u][k] : gsb[u][k]);
      const int tin = tile_lin % (tiles_x * tiles_y);
      double* dst = gr.ab + ((long)fb * gr.slots + tin % gr.slots) * (2 * c) + t;
      if (gr.atomic) atomic_add_d(dst, v);
      else *dst = v;
    }
  }
}

static long csr_words(int tl, int bs, int h, int w) {
  const long nd = (long)tl * bs * h * w;
  return (nd + 1) + nd + 2 * nd * (tl - 1) * 4;
}
extern "C" long dis_gather_csr_workspace(int tl, int bs, int h, int w) {
  if (tl <= 1 || bs <= 0 || h <= 1 || w <= 1) return -1;
  const long nd = (long)tl * bs * h * w;
  const long nblk = (nd + CSR_SCAN_ELEMS - 1) / CSR_SCAN_ELEMS;
  return csr_words(tl, bs, h, w) + nblk + 1;
}
extern "C" int dis_gather_csr_build(const float* flows, int* csr, int tl, int bs, int h, int w, void* stream) {
  if (!flows || !csr) return DIS_ERR_NULL;
  if (tl <= 1 || bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  const long nd = (long)tl * bs * h * w;
  if (nd * tl >= 2147483647L || nd * (tl - 1) * 4 >= 1073741823L) return DIS_ERR_UNSUPPORTED;  // int32 rows / positions
  hipStream_t s = (hipStream_t)stream;
  int* offsets = csr;
  int* cursor = csr + nd + 1;
  int* entries = csr + 2 * nd + 1;
  int* bsum = csr + csr_words(tl, bs, h, w);
  const int nblk = (int)((nd + CSR_SCAN_ELEMS - 1) / CSR_SCAN_ELEMS);
  // (a kernel, not hipMemsetAsync: captured in a hipGraph, the memset NODE was not reliably ordered in front of the count
  // kernel once eager launches had run between two replays - cursors were then counted on top of the previous replay's end
  // positions and the fill pass wrote out of bounds: "Memory access fault by GPU", round 2, scripts/debug_graph_pure.py)
  hipLaunchKernelGGL(csr_zero_kernel, dim3(dis_ew_grid(nd, 256)), dim3(256), 0, s, cursor, nd);
  const long items = nd * (tl - 1);
  int grid = dis_cdiv(items, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(csr_count_fill_kernel, dim3(grid), dim3(256), 0, s, flows, cursor, entries, tl, bs, h, w, 0);
  hipLaunchKernelGGL(csr_scan1_kernel, dim3(nblk), dim3(256), 0, s, (const int*)cursor, bsum, nd);
  hipLaunchKernelGGL(csr_scan2_kernel, dim3(1), dim3(64), 0, s, bsum, nblk);
  hipLaunchKernelGGL(csr_scan3_kernel, dim3(nblk), dim3(256), 0, s, cursor, offsets, (const int*)bsum, nd, nblk);
  hipLaunchKernelGGL(csr_count_fill_kernel, dim3(grid), dim3(256), 0, s, flows, cursor, entries, tl, bs, h, w, 1);
  hipLaunchKernelGGL(csr_sort_kernel, dim3(dis_ew_grid(nd, 256)), dim3(256), 0, s, (const int*)offsets, entries, nd);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
// ---- CSR of the Conv3D neighbour sets by SOURCE row (deterministic feature gradient, conv3d_knn.hip) ----
// entry = output pixel * 9 + neighbour; its source row = ((tb * h + iy) * w + ix) * 4 + slot with (tap, slot) = (id / 4, id % 4)
// of the selected candidate id and (iy, ix) = (oy, ox) * stride - 1 + (tap / 3, tap % 3), as c3_neighbor() computes it.
// csr = [offsets: nsrc + 1][cursor: nsrc][entries: nent][scan scratch], lists sorted by entry id.
__global__ void c3csr_count_fill_kernel(const unsigned char* __restrict__ idx, int* __restrict__ cursor,
                                        int* __restrict__ entries, long nent, int ho, int wo, int h, int w, int stride,
                                        int fill) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < nent; e += (long)gridDim.x * blockDim.x) {
    const long i = e / 9;
    const int id = idx[e];
    const int tap = id >> 2, slot = id & 3;
    const int ox = (int)(i % wo), oy = (int)((i / wo) % ho);
    const long tb = i / ((long)wo * ho);
    const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
    if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
    const long src = ((tb * h + iy) * w + ix) * 4 + slot;
    const int pos = atomicAdd(cursor + src, 1);
    if (fill) entries[pos] = (int)e;
  }
}
__global__ void c3csr_sort_kernel(const int* __restrict__ offsets, int* __restrict__ entries, long nd) {
  for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < nd; d += (long)gridDim.x * blockDim.x) {
    const int lo = offsets[d], hi = offsets[d + 1];
    const int len = hi - lo;
    if (len <= 1) continue;
    if (len <= CSR_SORT_MAX) {  // rank sort in registers (a source row is selected by at most 9 output pixels)
      int e[CSR_SORT_MAX];
#pragma unroll
      for (int k = 0; k < CSR_SORT_MAX; ++k) e[k] = k < len ? entries[lo + k] : 0x7fffffff;
#pragma unroll
      for (int k = 0; k < CSR_SORT_MAX; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < CSR_SORT_MAX; ++j) rank += (e[j] < e[k]) ? 1 : 0;  // (entry ids are unique)
        if (k < len) entries[lo + rank] = e[k];
      }
    } else {
      for (int a = lo + 1; a < hi; ++a) {
        const int key = entries[a];
        int b = a - 1;
        while (b >= lo && entries[b] > key) {
          entries[b + 1] = entries[b];
          --b;
        }
        entries[b + 1] = key;
      }
    }
  }
}
static int c3csr_dims(int tl, int bs, int h, int w, int stride, long* nsrc, long* nent, int* ho, int* wo) {
  if (tl != 4 || bs <= 0 || h <= 0 || w <= 0 || (stride != 1 && stride != 2)) return DIS_ERR_BAD_SHAPE;
  *ho = (h + 2 - 3) / stride + 1;
  *wo = (w + 2 - 3) / stride + 1;
  *nsrc = (long)tl * bs * h * w * 4;
  *nent = (long)tl * bs * *ho * *wo * 9;
  if (*nsrc >= 2147483647L || *nent >= 2147483647L) return DIS_ERR_UNSUPPORTED;
  return DIS_OK;
}
extern "C" long dis_conv3d_csr_workspace(int tl, int bs, int h, int w, int stride) {
  long nsrc, nent;
  int ho, wo;
  if (c3csr_dims(tl, bs, h, w, stride, &nsrc, &nent, &ho, &wo) != DIS_OK) return -1;
  return (nsrc + 1) + nsrc + nent + (nsrc + CSR_SCAN_ELEMS - 1) / CSR_SCAN_ELEMS + 1;
}
extern "C" int dis_conv3d_csr_build(const unsigned char* idx, int* csr, int tl, int bs, int h, int w, int stride,
                                    void* stream) {
  if (!idx || !csr) return DIS_ERR_NULL;
  long nsrc, nent;
  int ho, wo;
  const int rc = c3csr_dims(tl, bs, h, w, stride, &nsrc, &nent, &ho, &wo);
  if (rc != DIS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  int* offsets = csr;
  int* cursor = csr + nsrc + 1;
  int* entries = csr + 2 * nsrc + 1;
  int* bsum = entries + nent;
  const int nblk = (int)((nsrc + CSR_SCAN_ELEMS - 1) / CSR_SCAN_ELEMS);
  hipLaunchKernelGGL(csr_zero_kernel, dim3(dis_ew_grid(nsrc, 256)), dim3(256), 0, s, cursor, nsrc);
  int grid = dis_cdiv(nent, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(c3csr_count_fill_kernel, dim3(grid), dim3(256), 0, s, idx, cursor, entries, nent, ho, wo, h, w, stride, 0);
  hipLaunchKernelGGL(csr_scan1_kernel, dim3(nblk), dim3(256), 0, s, (const int*)cursor, bsum, nsrc);
  hipLaunchKernelGGL(csr_scan2_kernel, dim3(1), dim3(64), 0, s, bsum, nblk);
  hipLaunchKernelGGL(csr_scan3_kernel, dim3(nblk), dim3(256), 0, s, cursor, offsets, (const int*)bsum, nsrc, nblk);
  hipLaunchKernelGGL(c3csr_count_fill_kernel, dim3(grid), dim3(256), 0, s, idx, cursor, entries, nent, ho, wo, h, w, stride, 1);
  hipLaunchKernelGGL(c3csr_sort_kernel, dim3(dis_ew_grid(nsrc, 256)), dim3(256), 0, s, (const int*)offsets, entries, nsrc);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

static int gather_bwd_csr_tiled(const float* grad_out, const int* csr, const float* init, float* grad_feat, int tl, int bs, int h,
                                int w, int c, const GatherGnRes* gr, void* stream) {
  const long nd = (long)tl * bs * h * w;
  const int cg4 = c / 4;
  if (!(cg4 <= 256 && 256 % cg4 == 0 && GT_TW % (256 / cg4) == 0 && gather_tiled_on())) return DIS_ERR_UNSUPPORTED;
  const int tiles_x = dis_cdiv(w, GT_TW), tiles_y = dis_cdiv(h, GT_TH);
  const long ntile = (long)tl * bs * tiles_x * tiles_y;
  const long per_xcd = (ntile + 7) / 8;
  if (per_xcd * 8 > 0x7fffffffL) return DIS_ERR_UNSUPPORTED;
  if (gr) {
    GatherGnRes g2 = *gr;
    g2.atomic = tiles_x * tiles_y > g2.slots ? 1 : 0;
    hipLaunchKernelGGL(gather_warped_feat_bwd_csr_tiled_kernel<true>, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, csr, csr + 2 * nd + 1, init, grad_feat, h, w, tl, c, tiles_x, tiles_y, (int)ntile, (int)per_xcd, g2);
  } else {
    hipLaunchKernelGGL(gather_warped_feat_bwd_csr_tiled_kernel<false>, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, csr, csr + 2 * nd + 1, init, grad_feat, h, w, tl, c, tiles_x, tiles_y, (int)ntile, (int)per_xcd,
                       GatherGnRes{nullptr, nullptr, nullptr, 1, 0, 0});
  }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
/* dis_gather_warped_feat_bwd_csr for a `feat` that IS y = act(GroupNorm(x2) + residual) (act = SELU; a Block2D3D / ResNetBlock output,
 * reference model/multi_frame_networks.py:428-430, 540-542) when this launch completes the gradient wrt y (init = what the other
 * consumer left): grad_feat receives gres = g act'(y) - the gradient wrt the pre-activation value, which is also the residual
 * gradient - and ab_out (tl * bs, slots, 2, c) doubles (ZEROED by the caller) the channel sums A_c = sum gres, B_c = sum gres x2
 * that dis_gn_bwd_coef takes: the dis_gn_bwd_res_sums pass over g, y, x2 does not exist.  c % 4 == 0, 2 c <= 64.
 * DIS_ERR_UNSUPPORTED: no tiled instance for the shape (the caller runs dis_gather_warped_feat_bwd_csr + dis_gn_bwd_res_sums). */
extern "C" int dis_gather_warped_feat_bwd_csr_gnres(const float* grad_out, const int* csr, const float* init, float* grad_feat,
                                                    const float* y, const float* x2, double* ab_out, int slots, int act, int tl,
                                                    int bs, int h, int w, int c, void* stream) {
  if (!grad_out || !csr || !grad_feat || !y || !x2 || !ab_out) return DIS_ERR_NULL;
  if (tl <= 1 || bs <= 0 || h <= 1 || w <= 1 || c <= 0 || slots <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0 || 2 * c > 64 || (act != DIS_ACT_SELU && act != DIS_ACT_RELU && act != DIS_ACT_NONE)) return DIS_ERR_UNSUPPORTED;
  GatherGnRes gr{y, x2, ab_out, slots, act, 0};
  return gather_bwd_csr_tiled(grad_out, csr, init, grad_feat, tl, bs, h, w, c, &gr, stream);
}
extern "C" int dis_gather_warped_feat_bwd_csr(const float* grad_out, const int* csr, const float* init,
                                              float* grad_feat, int tl, int bs, int h, int w, int c, void* stream) {
  if (!grad_out || !csr || !grad_feat) return DIS_ERR_NULL;
  if (tl <= 1 || bs <= 0 || h <= 1 || w <= 1 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0) return DIS_ERR_UNSUPPORTED;
  const long nd = (long)tl * bs * h * w;
  const long total = nd * (c / 4);
  if (gather_bwd_csr_tiled(grad_out, csr, init, grad_feat, tl, bs, h, w, c, nullptr, stream) == DIS_OK) return DIS_OK;
  int grid = dis_cdiv(total, 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(gather_warped_feat_bwd_csr_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, csr,
                     csr + 2 * nd + 1, init, grad_feat, nd, tl, c);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-frame geometry: warped xyz + forward/backward mask per (target, slot)
// (reference multi_frame_networks.py:172-214, 283-294); out (tl,bs,h,w,tl,4)
// ------------------------------------------------------------------------------------------------
struct GeomCam {
  float Ki[9];
};

__device__ __forceinline__ void core_ray(const float* Ki, int u, int v, float* r) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
    r[c] = (float)((double)u * (double)Ki[c * 3 + 0] + (double)v * (double)Ki[c * 3 + 1] + (double)Ki[c * 3 + 2]);
}

// camera-t coordinates of core pixel (x,y) of frame j:  ((d*ray - t_j) R_j) R_t^T + t_t
__device__ __forceinline__ void xyz_in_view(const GeomCam& cam, int us, int vs, int x, int y, float d,
                                            const float* Rj, const float* tj, const float* Rt, const float* tt,
                                            float* o) {
  float ray[3], a[3], wv[3];
  core_ray(cam.Ki, us * x, vs * y, ray);
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = d * ray[c] - tj[c];
  // torch.matmul on the CPU accumulates K = 3 as fma(a2, b2, fma(a1, b1, a0 * b0)); the masks and Conv3D's neighbour
  // keys derived from these coordinates are index-class outputs, so the chain is reproduced exactly (tests/bitexact.py)
#pragma unroll
  for (int c = 0; c < 3; ++c) wv[c] = __fmaf_rn(a[2], Rj[2 * 3 + c], __fmaf_rn(a[1], Rj[1 * 3 + c], a[0] * Rj[0 * 3 + c]));
#pragma unroll
  for (int c = 0; c < 3; ++c)
    o[c] = __fmaf_rn(wv[2], Rt[c * 3 + 2], __fmaf_rn(wv[1], Rt[c * 3 + 1], wv[0] * Rt[c * 3 + 0])) + tt[c];
}

__global__ void mf_geometry_kernel(const float* __restrict__ depth, const float* __restrict__ R,
                                   const float* __restrict__ tv, const float* __restrict__ flows, GeomCam cam,
                                   int us, int vs, float4* __restrict__ out, int tl, int bs, int h, int w) {
  const long hw = (long)h * w;
  const long total = (long)tl * bs * hw * tl;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i % tl);
    long r = i / tl;
    const long p = r % hw;
    r /= hw;
    const int b = (int)(r % bs);
    const int t = (int)(r / bs);
    const int j = slot_frame(t, s);
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float* Rt = R + ((long)t * bs + b) * 9;
    const float* tt = tv + ((long)t * bs + b) * 3;
    const float* Rj = R + ((long)j * bs + b) * 9;
    const float* tj = tv + ((long)j * bs + b) * 3;
    const float* dj = depth + ((long)j * bs + b) * hw;
    float4 o;
    if (s == 0) {
      float q[3];
      xyz_in_view(cam, us, vs, x, y, dj[p], Rj, tj, Rt, tt, q);
      o = make_float4(q[0], q[1], q[2], 1.f);
    } else {
      const float2 f0 = *(const float2*)(flows + ((((long)t * tl + j) * bs + b) * hw + p) * 2);
      const Taps tp = make_taps(f0.x + (float)x, f0.y + (float)y, h, w);
      const float* fl1 = flows + (((long)j * tl + t) * bs + b) * hw * 2;
      float acc[3] = {0.f, 0.f, 0.f};
      float f10x = 0.f, f10y = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool valid = (k == 0) ? tp.v00 : (k == 1) ? tp.v01 : (k == 2) ? tp.v10 : tp.v11;
        const float wgt = (k == 0) ? tp.w00 : (k == 1) ? tp.w01 : (k == 2) ? tp.w10 : tp.w11;
        if (valid) {
          const int tx = tp.x0 + (k & 1), ty = tp.y0 + (k >> 1);
          float q[3];
          xyz_in_view(cam, us, vs, tx, ty, dj[(long)ty * w + tx], Rj, tj, Rt, tt, q);
          // grid_sample's tap sum fma(se, w, fma(sw, w, fma(ne, w, nw * w))); a tap outside the image adds fma(0, w, acc) = acc
          acc[0] = __fmaf_rn(q[0], wgt, acc[0]);
          acc[1] = __fmaf_rn(q[1], wgt, acc[1]);
          acc[2] = __fmaf_rn(q[2], wgt, acc[2]);
          const float2 g = *(const float2*)(fl1 + ((long)ty * w + tx) * 2);
          f10x = __fmaf_rn(g.x, wgt, f10x);
          f10y = __fmaf_rn(g.y, wgt, f10y);
        }
      }
      const float sx = f0.x + f10x, sy = f0.y + f10y;
      const float lhs = sx * sx + sy * sy;
      const float rhs = 0.5f + 0.01f * ((f0.x * f0.x + f0.y * f0.y) + (f10x * f10x + f10y * f10y));
      o = make_float4(acc[0], acc[1], acc[2], lhs < rhs ? 1.f : 0.f);
    }
    out[i] = o;
  }
}

extern "C" int dis_mf_geometry(const float* depth_core, const float* R, const float* t, const float* flows,
                               const float* Kinv_host, int u_step, int v_step, float* out, int tl, int bs, int h,
                               int w, void* stream) {
  if (!depth_core || !R || !t || !flows || !Kinv_host || !out) return DIS_ERR_NULL;
  if (tl <= 0 || bs <= 0 || h <= 1 || w <= 1) return DIS_ERR_BAD_SHAPE;
  GeomCam cam;
  for (int i = 0; i < 9; ++i) cam.Ki[i] = Kinv_host[i];
  long total = (long)tl * bs * h * w * tl;
  hipLaunchKernelGGL(mf_geometry_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, depth_core,
                     R, t, flows, cam, u_step, v_step, (float4*)out, tl, bs, h, w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// bilinear (align_corners) resize of the (tl*bs, h, w, tl*4) geometry tensor; masks re-binarised (> 0.5)
__global__ void mf_geometry_resize_kernel(const float4* __restrict__ x, float4* __restrict__ y, int n, int hin,
                                          int win, int hout, int wout, int slots) {
  const long total = (long)n * hout * wout * slots;
  const float sy = resize_scale(hin, hout, 1), sx = resize_scale(win, wout, 1);
  const bool small = hout + wout <= 128;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i % slots);
    long p = i / slots;
    const int ox = (int)(p % wout);
    p /= wout;
    const int oy = (int)(p % hout);
    const int b = (int)(p / hout);
    const Lerp Ly = resize_src(oy, sy, hin, 1), Lx = resize_src(ox, sx, win, 1);
    const float4* base = x + (long)b * hin * win * slots + s;
    const float4 a = base[((long)Ly.i0 * win + Lx.i0) * slots], bq = base[((long)Ly.i0 * win + Lx.i1) * slots];
    const float4 c = base[((long)Ly.i1 * win + Lx.i0) * slots], d = base[((long)Ly.i1 * win + Lx.i1) * slots];
    float4 o;
    o.x = bilerp_aten(a.x, bq.x, c.x, d.x, Ly, Lx, small);
    o.y = bilerp_aten(a.y, bq.y, c.y, d.y, Ly, Lx, small);
    o.z = bilerp_aten(a.z, bq.z, c.z, d.z, Ly, Lx, small);
    const float m = bilerp_aten(a.w, bq.w, c.w, d.w, Ly, Lx, small);
    o.w = m > 0.5f ? 1.f : 0.f;
    y[i] = o;
  }
}
extern "C" int dis_mf_geometry_resize(const float* geom, float* out, int tl, int bs, int hin, int win, int hout,
                                      int wout, void* stream) {
  if (!geom || !out) return DIS_ERR_NULL;
  if (tl <= 0 || bs <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  long total = (long)tl * bs * hout * wout * tl;
  hipLaunchKernelGGL(mf_geometry_resize_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)geom, (float4*)out, tl * bs, hin, win, hout, wout, tl);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// slot weighting of the gathered features: out = (wf * mask) / mean_slots(mask)
// (reference multi_frame_networks.py:410).  Linear in wf, so the same call is its own backward.
// ------------------------------------------------------------------------------------------------
__global__ void mask_weight_slots_kernel(const float* __restrict__ wf, const float4* __restrict__ geom,
                                         float* __restrict__ out, long pixels, int tl, int c, int accumulate) {
  const int cg = c >> 2;
  const long total = pixels * tl * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cg;
    const int s = (int)(r % tl);
    const long p = r / tl;
    float msum = 0.f;
    for (int k = 0; k < tl; ++k) msum += geom[p * tl + k].w;
    const float mean = msum / (float)tl;
    const float m = geom[p * tl + s].w;
    float4 v = *(const float4*)(wf + i * 4);
    v.x = (v.x * m) / mean;
    v.y = (v.y * m) / mean;
    v.z = (v.z * m) / mean;
    v.w = (v.w * m) / mean;
    if (accumulate) {
      const float4 o = *(const float4*)(out + i * 4);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *(float4*)(out + i * 4) = v;
  }
}
// wgt[p][s] = mask[p][s] / mean_s(mask[p][.])  (the multiplier of mask_weight_slots as a small tensor of its own, so the
// multi-frame 1x1 convolution can apply it while staging its input instead of reading a pre-scaled copy)
__global__ void slot_weights_kernel(const float4* __restrict__ geom, float* __restrict__ wgt, long pixels, int tl) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < pixels; p += (long)gridDim.x * blockDim.x) {
    float msum = 0.f;
    for (int k = 0; k < tl; ++k) msum += geom[p * tl + k].w;
    const float mean = msum / (float)tl;
    for (int k = 0; k < tl; ++k) wgt[p * tl + k] = geom[p * tl + k].w / mean;
  }
}
extern "C" int dis_slot_weights(const float* geom, float* wgt, long pixels, int tl, void* stream) {
  if (!geom || !wgt) return DIS_ERR_NULL;
  if (pixels <= 0 || tl <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(slot_weights_kernel, dim3(dis_ew_grid(pixels, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)geom, wgt, pixels, tl);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_mask_weight_slots(const float* wf, const float* geom, float* out, long pixels, int tl, int c,
                                     int accumulate, void* stream) {
  if (!wf || !geom || !out) return DIS_ERR_NULL;
  if (pixels <= 0 || tl <= 0 || c <= 0) return DIS_ERR_BAD_SHAPE;
  if (c % 4 != 0) return DIS_ERR_UNSUPPORTED;
  long total = pixels * tl * (c / 4);
  hipLaunchKernelGGL(mask_weight_slots_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, wf,
                     (const float4*)geom, out, pixels, tl, c, accumulate);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
