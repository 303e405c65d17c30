// Conv3D: k-nearest-neighbour continuous convolution over the 3x3xTL candidate window
// (reference model/multi_frame_networks.py:432-512), for all target frames of a step in one launch.
//
// Data layout is built for this gather: geom (tl,bs,h,w,tl,4) keeps xyz+mask of the 4 slots of a pixel
// in one 64-B line, wf (tl,bs,h,w,tl,32) keeps each slot's 32 features in one 128-B line.
//   kernel 1 (select):    one lane per output pixel: 36 keys, masked top-9 (lowest candidate id wins ties)
//   kernel 2 (aggregate): one half-wave (32 lanes = 32 channels) per output pixel: MLP 3->16->32 on the
//                         local coordinates, feature gather (one line per neighbour), 32x32 mix, SELU.
//   backward:             same mapping; per-lane register accumulators for the weight gradients, one
//                         coalesced 128-B float-atomic row per neighbour for the feature gradient.
#include "common.h"
#include <float.h>

#define C3_TL 4
#define C3_NB 9
#define C3_C 32
#define C3_H1 16
#define C3_NCAND (9 * C3_TL)

struct C3Dims {
  int tl, bs, h, w, ho, wo, stride;
};

__global__ void conv3d_select_kernel(const float4* __restrict__ geom, unsigned char* __restrict__ idx, C3Dims d) {
  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % d.wo);
    const int oy = (int)((i / d.wo) % d.ho);
    const long tb = i / ((long)d.wo * d.ho);
    const float4* g = geom + tb * d.h * d.w * C3_TL;
    const int cy = oy * d.stride, cx = ox * d.stride;
    const float4 ctr = g[((long)cy * d.w + cx) * C3_TL];
    const float cden = ctr.z + 1e-12f;
    const float pcx = ctr.x / cden, pcy = ctr.y / cden, pcz = ctr.z / cden;
    float bk[C3_NB];
    int bi[C3_NB];
#pragma unroll
    for (int k = 0; k < C3_NB; ++k) {
      bk[k] = INFINITY;
      bi[k] = 255;
    }
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = cy - 1 + ky;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = cx - 1 + kx;
        const bool inb = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
#pragma unroll
        for (int s = 0; s < C3_TL; ++s) {
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding: xyz = 0, mask = 0
          if (inb) q = g[((long)iy * d.w + ix) * C3_TL + s];
          const float den = q.z + 1e-12f;
          const float dx = q.x / den - pcx, dy = q.y / den - pcy, dz = q.z / den - pcz;
          float key = (dx * dx + dy * dy) + dz * dz;
          key = fminf(key, 0.5f * FLT_MAX);
          if (!(q.w > 0.5f)) key = FLT_MAX;  // masked: the reference's max+1 fill, all such keys tie
          const int id = (ky * 3 + kx) * C3_TL + s;
          // branch-free insertion into the ascending list (static register indexing); strict '<' keeps
          // the earlier (lower) candidate id in front on ties
#pragma unroll
          for (int k = C3_NB - 1; k >= 0; --k) {
            const bool less_prev = (k > 0) && (key < bk[k > 0 ? k - 1 : 0]);
            const bool less_cur = key < bk[k];
            const float nk = less_prev ? bk[k > 0 ? k - 1 : 0] : (less_cur ? key : bk[k]);
            const int ni = less_prev ? bi[k > 0 ? k - 1 : 0] : (less_cur ? id : bi[k]);
            bk[k] = nk;
            bi[k] = ni;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < C3_NB; ++k) idx[i * C3_NB + k] = (unsigned char)bi[k];
  }
}

struct C3Params {
  const float* w1;  // (16,3)
  const float* b1;  // (16)
  const float* w2;  // (32,16)
  const float* b2;  // (32)
  const float* w;   // (32,32)
};

// per-half-wave LDS scratch
struct __attribute__((aligned(16))) C3Scratch {
  float h1[C3_NB][C3_H1];
  float v32a[C3_C];
  float v32b[C3_C];
};

__device__ __forceinline__ float selu_f(float x) {
  return x > 0.f ? SELU_SCALE_F * x : (SELU_SCALE_F * SELU_ALPHA_F) * (expf(x) - 1.f);
}

__device__ __forceinline__ void c3_neighbor(const C3Dims& d, int oy, int ox, int id, int* iy, int* ix, int* slot,
                                            bool* inb) {
  const int tap = id / C3_TL;
  *slot = id % C3_TL;
  *iy = oy * d.stride - 1 + tap / 3;
  *ix = ox * d.stride - 1 + tap % 3;
  *inb = *iy >= 0 && *iy < d.h && *ix >= 0 && *ix < d.w;
}

__global__ __launch_bounds__(256) void conv3d_aggregate_kernel(const float4* __restrict__ geom,
                                                                 const float* __restrict__ wf, C3Params P,
                                                                 const unsigned char* __restrict__ idx,
                                                                 float* __restrict__ y, C3Dims d) {
  __shared__ C3Scratch scr[8];
  const int hw_id = threadIdx.x >> 5;      // half-wave inside the block
  const int c = threadIdx.x & 31;          // channel
  const int k16 = c & 15;
  C3Scratch& S = scr[hw_id];
  float w2r[C3_H1], wcol[C3_C];
#pragma unroll
  for (int k = 0; k < C3_H1; ++k) w2r[k] = P.w2[c * C3_H1 + k];
#pragma unroll
  for (int k = 0; k < C3_C; ++k) wcol[k] = P.w[k * C3_C + c];
  const float b2c = P.b2[c];
  const float w1x = P.w1[k16 * 3], w1y = P.w1[k16 * 3 + 1], w1z = P.w1[k16 * 3 + 2], b1k = P.b1[k16];

  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  const long nhalf = (long)gridDim.x * 8;
  for (long i = (long)blockIdx.x * 8 + hw_id; i < total; i += nhalf) {
    const int ox = (int)(i % d.wo);
    const int oy = (int)((i / d.wo) % d.ho);
    const long tb = i / ((long)d.wo * d.ho);
    const float4* g = geom + tb * d.h * d.w * C3_TL;
    const float* f = wf + tb * d.h * d.w * C3_TL * C3_C;
    const float4 ctr = g[((long)(oy * d.stride) * d.w + ox * d.stride) * C3_TL];
    float agg = 0.f;
#pragma unroll
    for (int n = 0; n < C3_NB; ++n) {
      const int id = idx[i * C3_NB + n];
      int iy, ix, slot;
      bool inb;
      c3_neighbor(d, oy, ox, id, &iy, &ix, &slot, &inb);
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      float fv = 0.f;
      if (inb) {
        q = g[((long)iy * d.w + ix) * C3_TL + slot];
        fv = f[(((long)iy * d.w + ix) * C3_TL + slot) * C3_C + c];
      }
      const float lx = q.x - ctr.x, ly = q.y - ctr.y, lz = q.z - ctr.z;
      const float h1 = selu_f(((lx * w1x + ly * w1y) + lz * w1z) + b1k);
      if (c < C3_H1) S.h1[n][c] = h1;
      __builtin_amdgcn_wave_barrier();
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < C3_H1; ++k) a += S.h1[n][k] * w2r[k];
      const float h2 = selu_f(a + b2c);
      agg += h2 * fv;
    }
    S.v32a[c] = agg;
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
#pragma unroll
    for (int k = 0; k < C3_C; ++k) o += S.v32a[k] * wcol[k];
    y[i * C3_C + c] = selu_f(o);
    __builtin_amdgcn_wave_barrier();
  }
}

// parameter-gradient accumulator layout (doubles / floats): dense1_w 48, dense1_b 16, dense2_w 512,
// dense2_b 32, w 1024
#define C3_OFF_W1 0
#define C3_OFF_B1 48
#define C3_OFF_W2 64
#define C3_OFF_B2 576
#define C3_OFF_W 608
#define C3_NPARAM 1632

__global__ __launch_bounds__(256) void conv3d_bwd_kernel(const float4* __restrict__ geom,
                                                           const float* __restrict__ wf, C3Params P,
                                                           const unsigned char* __restrict__ idx,
                                                           const float* __restrict__ y, const float* __restrict__ gy,
                                                           float* __restrict__ gwf, double* __restrict__ acc,
                                                           C3Dims d) {
  __shared__ C3Scratch scr[8];
  __shared__ float redbuf[8][C3_C];
  const int hw_id = threadIdx.x >> 5;
  const int c = threadIdx.x & 31;
  const int k16 = c & 15;
  C3Scratch& S = scr[hw_id];
  float w2r[C3_H1], wrow[C3_C], wcol[C3_C], w2c[C3_C];
#pragma unroll
  for (int k = 0; k < C3_H1; ++k) w2r[k] = P.w2[c * C3_H1 + k];
#pragma unroll
  for (int k = 0; k < C3_C; ++k) {
    wrow[k] = P.w[c * C3_C + k];
    wcol[k] = P.w[k * C3_C + c];
    w2c[k] = P.w2[k * C3_H1 + k16];  // column k16 of dense2 (used by lanes < 16)
  }
  const float b2c = P.b2[c];
  const float w1x = P.w1[k16 * 3], w1y = P.w1[k16 * 3 + 1], w1z = P.w1[k16 * 3 + 2], b1k = P.b1[k16];

  float dwrow[C3_C], dw2r[C3_H1];
#pragma unroll
  for (int k = 0; k < C3_C; ++k) dwrow[k] = 0.f;
#pragma unroll
  for (int k = 0; k < C3_H1; ++k) dw2r[k] = 0.f;
  float db2 = 0.f, db1 = 0.f, dw1x = 0.f, dw1y = 0.f, dw1z = 0.f;

  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  const long nhalf = (long)gridDim.x * 8;
  for (long i = (long)blockIdx.x * 8 + hw_id; i < total; i += nhalf) {
    const int ox = (int)(i % d.wo);
    const int oy = (int)((i / d.wo) % d.ho);
    const long tb = i / ((long)d.wo * d.ho);
    const float4* g = geom + tb * d.h * d.w * C3_TL;
    const float* f = wf + tb * d.h * d.w * C3_TL * C3_C;
    float* gf = gwf + tb * d.h * d.w * C3_TL * C3_C;
    const float4 ctr = g[((long)(oy * d.stride) * d.w + ox * d.stride) * C3_TL];
    // ---- recompute the forward
    float h2v[C3_NB], fvv[C3_NB], lxv[C3_NB], lyv[C3_NB], lzv[C3_NB];
    long foff[C3_NB];
    float agg = 0.f;
#pragma unroll
    for (int n = 0; n < C3_NB; ++n) {
      const int id = idx[i * C3_NB + n];
      int iy, ix, slot;
      bool inb;
      c3_neighbor(d, oy, ox, id, &iy, &ix, &slot, &inb);
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      float fv = 0.f;
      foff[n] = -1;
      if (inb) {
        q = g[((long)iy * d.w + ix) * C3_TL + slot];
        foff[n] = (((long)iy * d.w + ix) * C3_TL + slot) * C3_C + c;
        fv = f[foff[n]];
      }
      lxv[n] = q.x - ctr.x; lyv[n] = q.y - ctr.y; lzv[n] = q.z - ctr.z;
      const float h1 = selu_f(((lxv[n] * w1x + lyv[n] * w1y) + lzv[n] * w1z) + b1k);
      if (c < C3_H1) S.h1[n][c] = h1;
      __builtin_amdgcn_wave_barrier();
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < C3_H1; ++k) a += S.h1[n][k] * w2r[k];
      h2v[n] = selu_f(a + b2c);
      fvv[n] = fv;
      agg += h2v[n] * fv;
    }
    // ---- output mix backward
    const float yv = y[i * C3_C + c];
    const float gpre = gy[i * C3_C + c] * act_grad_from_out(yv, DIS_ACT_SELU);
    S.v32a[c] = gpre;
    __builtin_amdgcn_wave_barrier();
    float dagg = 0.f;
#pragma unroll
    for (int k = 0; k < C3_C; ++k) {
      const float gk = S.v32a[k];
      dagg += gk * wrow[k];     // d agg[c] = sum_c' gpre[c'] w[c][c']
      dwrow[k] += agg * gk;     // d w[c][c'] += agg[c] gpre[c']
    }
    __builtin_amdgcn_wave_barrier();
    // ---- neighbours
#pragma unroll
    for (int n = 0; n < C3_NB; ++n) {
      if (foff[n] >= 0) atomicAdd(gf + foff[n], dagg * h2v[n]);
      const float dpre2 = (dagg * fvv[n]) * act_grad_from_out(h2v[n], DIS_ACT_SELU);
      db2 += dpre2;
#pragma unroll
      for (int k = 0; k < C3_H1; ++k) dw2r[k] += dpre2 * S.h1[n][k];
      S.v32b[c] = dpre2;
      __builtin_amdgcn_wave_barrier();
      // d h1[k] = sum_c w2[c][k] dpre2[c]   (lanes k < 16; the upper 16 lanes mirror them)
      float dh1 = 0.f;
#pragma unroll
      for (int k = 0; k < C3_C; ++k) dh1 += S.v32b[k] * w2c[k];
      const float h1 = S.h1[n][k16];
      const float dpre1 = dh1 * act_grad_from_out(h1, DIS_ACT_SELU);
      if (c < C3_H1) {
        db1 += dpre1;
        dw1x += dpre1 * lxv[n];
        dw1y += dpre1 * lyv[n];
        dw1z += dpre1 * lzv[n];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // ---- reduce the per-lane parameter gradients over the 8 half-waves, then fp64 atomics
  auto reduce_store = [&](float v, int off, bool active) {
    __syncthreads();
    redbuf[hw_id][c] = active ? v : 0.f;
    __syncthreads();
    if (hw_id == 0 && active) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += (double)redbuf[k][c];
      atomic_add_d(acc + off, t);
    }
  };
#pragma unroll
  for (int k = 0; k < C3_C; ++k) reduce_store(dwrow[k], C3_OFF_W + c * C3_C + k, true);
#pragma unroll
  for (int k = 0; k < C3_H1; ++k) reduce_store(dw2r[k], C3_OFF_W2 + c * C3_H1 + k, true);
  reduce_store(db2, C3_OFF_B2 + c, true);
  reduce_store(db1, C3_OFF_B1 + k16, c < C3_H1);
  reduce_store(dw1x, C3_OFF_W1 + k16 * 3 + 0, c < C3_H1);
  reduce_store(dw1y, C3_OFF_W1 + k16 * 3 + 1, c < C3_H1);
  reduce_store(dw1z, C3_OFF_W1 + k16 * 3 + 2, c < C3_H1);
}

__global__ void c3_cast_kernel(const double* __restrict__ a, float* __restrict__ o, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = (float)a[i];
}

static int c3_dims(C3Dims* d, int tl, int bs, int h, int w, int stride) {
  if (tl != C3_TL) return DIS_ERR_UNSUPPORTED;
  if (bs <= 0 || h <= 0 || w <= 0) return DIS_ERR_BAD_SHAPE;
  if (stride != 1 && stride != 2) return DIS_ERR_UNSUPPORTED;
  d->tl = tl; d->bs = bs; d->h = h; d->w = w; d->stride = stride;
  d->ho = (h + 2 - 3) / stride + 1;
  d->wo = (w + 2 - 3) / stride + 1;
  return DIS_OK;
}

extern "C" int dis_conv3d_knn_select(const float* geom, unsigned char* idx_out, int tl, int bs, int h, int wd,
                                     int stride, void* stream) {
  if (!geom || !idx_out) return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  const long total = (long)tl * bs * d.ho * d.wo;
  hipLaunchKernelGGL(conv3d_select_kernel, dim3(dis_ew_grid(total, 128)), dim3(128), 0, (hipStream_t)stream,
                     (const float4*)geom, idx_out, d);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_conv3d_knn_fwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                  const float* dense2_w, const float* dense2_b, const float* w,
                                  const unsigned char* idx, float* y, int tl, int bs, int h, int wd, int stride,
                                  void* stream) {
  if (!geom || !wf || !dense1_w || !dense1_b || !dense2_w || !dense2_b || !w || !idx || !y) return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)tl * bs * d.ho * d.wo;
  C3Params P{dense1_w, dense1_b, dense2_w, dense2_b, w};
  int grid = dis_cdiv(total, 8);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(conv3d_aggregate_kernel, dim3(grid), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, d);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_conv3d_knn_bwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                  const float* dense2_w, const float* dense2_b, const float* w,
                                  const unsigned char* idx, const float* y, const float* gy, float* grad_wf,
                                  float* gparams, double* acc, int tl, int bs, int h, int wd, int stride,
                                  void* stream) {
  if (!geom || !wf || !dense1_w || !dense1_b || !dense2_w || !dense2_b || !w || !idx || !y || !gy || !grad_wf ||
      !gparams || !acc)
    return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)tl * bs * d.ho * d.wo;
  C3Params P{dense1_w, dense1_b, dense2_w, dense2_b, w};
  int grid = dis_cdiv(total, 8);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(conv3d_bwd_kernel, dim3(grid), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, gy, grad_wf,
                     acc, d);
  hipLaunchKernelGGL(c3_cast_kernel, dim3(7), dim3(256), 0, s, (const double*)acc, gparams, C3_NPARAM);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
