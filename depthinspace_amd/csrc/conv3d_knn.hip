// Conv3D: k-nearest-neighbour continuous convolution over the 3x3xTL candidate window
// (reference model/multi_frame_networks.py:432-512), for all target frames of a step in one launch.
//
// Data layout is built for this gather: geom (tl,bs,h,w,tl,4) keeps xyz+mask of the 4 slots of a pixel
// in one 64-B line, wf (tl,bs,h,w,tl,32) keeps each slot's 32 features in one 128-B line.
//   kernel 1 (select):    one lane per output pixel: 36 keys, masked top-9 exactly as torch.topk picks them
//   kernel 2 (forward):   one wave per group of 16 output pixels.  Every per-pixel matrix product (dense2 16->32 per
//                         neighbour, the 32x32 mix) runs on the matrix cores (v_mfma_f32_16x16x4_f32) in a
//                         "pixel on the lane" layout: lane (li,lg) owns pixel li and channels {16*mt + 4*lg + r},
//                         which is both the MFMA B-operand layout for products that contract over channels and
//                         the MFMA result layout, so the chain h1 -> h2 -> agg -> y needs no data movement.
//   backward:             same mapping for the recomputed forward and the per-pixel products (d agg, d h1); the
//                         weight gradients contract over PIXELS, so their operands are transposed once through
//                         a small per-wave LDS tile and accumulated in MFMA accumulators across all groups of a
//                         wave; feature gradients leave as 128-B float-atomic rows.  Parameter gradients are
//                         reduced through per-block slabs (deterministic, no atomics).
#include "common.h"
#include "nth_select.h"
#include <float.h>

#define C3_TL 4
#define C3_NB 9
#define C3_C 32
#define C3_H1 16
#define C3_NCAND (9 * C3_TL)

struct C3Dims {
  int tl, bs, h, w, ho, wo, stride;
};

// One lane per output pixel.  The 36 keys are the reference's, operation for operation (multi_frame_networks.py:490-497:
// plane = xyz / (z + 1e-12), squared distance to candidate 16 summed x, y, z; -ffp-contract=off keeps every rounding),
// and the 9 neighbours are the ones torch.topk(largest=False, sorted=False) returns on the CPU, in its order: ATen runs
// std::nth_element on the (key, id) row, restated in nth_select.h, so tied keys (masked candidates all share one fill
// value; equidistant planar candidates) resolve exactly as in the reference.
#define C3_SEL_T 64
struct C3SelView {
  NthPair* base;
  __device__ __forceinline__ NthPair& operator[](int i) const { return base[i * C3_SEL_T]; }
};
__global__ __launch_bounds__(C3_SEL_T) void conv3d_select_kernel(const float4* __restrict__ geom,
                                                                 unsigned char* __restrict__ idx, C3Dims d) {
  __shared__ NthPair sq[C3_NCAND * C3_SEL_T];
  const C3SelView q{sq + threadIdx.x};
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
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = cy - 1 + ky;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = cx - 1 + kx;
        const bool inb = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
#pragma unroll
        for (int s = 0; s < C3_TL; ++s) {
          float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding: xyz = 0, mask = 0
          if (inb) qv = g[((long)iy * d.w + ix) * C3_TL + s];
          const float den = qv.z + 1e-12f;
          const float dx = qv.x / den - pcx, dy = qv.y / den - pcy, dz = qv.z / den - pcz;
          float key = (dx * dx + dy * dy) + dz * dz;
          // masked: the reference fills max(dist) + 1, one value above every valid key; FLT_MAX orders the same way
          if (!(qv.w > 0.5f)) key = FLT_MAX;
          const int id = (ky * 3 + kx) * C3_TL + s;
          q[id] = NthPair{key, id};
        }
      }
    }
    nth_element_pairs(q, C3_NCAND, C3_NB - 1);
#pragma unroll
    for (int k = 0; k < C3_NB; ++k) idx[i * C3_NB + k] = (unsigned char)q[k].id;
  }
}

struct C3Params {
  const float* w1;  // (16,3)
  const float* b1;  // (16)
  const float* w2;  // (32,16)
  const float* b2;  // (32)
  const float* w;   // (32,32)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C3_GP 16   // output pixels per wave group
#define C3_WS 36   // LDS row stride of w   (32 + 4: conflict-free b128 row reads and b32 column reads)
#define C3_W2S 20  // LDS row stride of w2  (16 + 4)
#define C3_XS 36   // LDS row stride of the 16x32 transposition tile
#define C3_HS 20   // LDS row stride of the 16x16 transposition tile

// parameter-gradient layout (floats): w 1024, dense1_w 48, dense1_b 16, dense2_w 512, dense2_b 32 - the order of
// Conv3D's parameters() (own parameter first, then the sub-modules), i.e. of the trainer's flat gradient buffer, so the
// reducer can write straight into it
#define C3_OFF_W 0
#define C3_OFF_W1 1024
#define C3_OFF_B1 1072
#define C3_OFF_W2 1088
#define C3_OFF_B2 1600
#define C3_NPARAM 1632

struct __attribute__((aligned(16))) C3Lds {
  float w[C3_C * C3_WS];
  float w2[C3_C * C3_W2S];
  float X[4][C3_GP * C3_XS];
  float H[4][C3_GP * C3_HS];
  long F[4][C3_GP];
  float4 D[4][C3_NB * C3_GP];  // neighbour descriptors of the group a wave works on: (lx, ly, lz, element index | -1)
};

__device__ __forceinline__ float selu_f(float x) {
  return x > 0.f ? SELU_SCALE_F * x : (SELU_SCALE_F * SELU_ALPHA_F) * (selu_exp(x) - 1.f);
}

__device__ __forceinline__ void c3_load_weights(C3Lds& L, const C3Params& P) {
  for (int i = threadIdx.x; i < C3_C * C3_C; i += blockDim.x) L.w[(i >> 5) * C3_WS + (i & 31)] = P.w[i];
  for (int i = threadIdx.x; i < C3_C * C3_H1; i += blockDim.x) L.w2[(i >> 4) * C3_W2S + (i & 15)] = P.w2[i];
}

// Neighbour n of output pixel i (owned by this lane): local coordinates and the element offset of its
// 32-feature line (or -1 for a zero-padded border candidate / an invalid lane).
struct C3Nb {
  float lx, ly, lz;
  long foff;
};
__device__ __forceinline__ C3Nb c3_neighbor(const float4* __restrict__ geom, const unsigned char* __restrict__ idx,
                                            const C3Dims& d, long i, int n, bool pv, int oy, int ox, long tb,
                                            const float4& ctr) {
  C3Nb r;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  r.foff = -1;
  if (pv) {
    const int id = idx[i * C3_NB + n];
    const int tap = id / C3_TL, slot = id % C3_TL;
    const int iy = oy * d.stride - 1 + tap / 3, ix = ox * d.stride - 1 + tap % 3;
    if (iy >= 0 && iy < d.h && ix >= 0 && ix < d.w) {
      const long e = (tb * d.h * d.w + (long)iy * d.w + ix) * C3_TL + slot;
      q = geom[e];
      r.foff = e * C3_C;
    }
  }
  r.lx = q.x - ctr.x;
  r.ly = q.y - ctr.y;
  r.lz = q.z - ctr.z;
  return r;
}

// The neighbour of (pixel, n) is found through two dependent loads (selection index -> geometry); the feature row is a
// third.  Building the 9 x 16 descriptors of a group once, three per lane with all chains in flight together, and
// reading them back from LDS takes those chains out of the per-neighbour loops (the kernels are latency-bound there).
__device__ __forceinline__ void c3_build_desc(float4* D, const float4* __restrict__ geom,
                                              const unsigned char* __restrict__ idx, const C3Dims& d, long ic, bool pv,
                                              int oy, int ox, long tb, const float4& ctr, int li, int lg) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int n = lg + 4 * k;
    if (n < C3_NB) {
      const C3Nb nb = c3_neighbor(geom, idx, d, ic, n, pv, oy, ox, tb, ctr);
      D[n * C3_GP + li] = make_float4(nb.lx, nb.ly, nb.lz, __int_as_float(nb.foff >= 0 ? (int)(nb.foff / C3_C) : -1));
    }
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ C3Nb c3_desc(const float4* D, int n, int li) {
  const float4 v = D[n * C3_GP + li];
  C3Nb r;
  r.lx = v.x;
  r.ly = v.y;
  r.lz = v.z;
  const int e = __float_as_int(v.w);
  r.foff = e >= 0 ? (long)e * C3_C : -1;
  return r;
}

struct C3Lane {
  int li, lg;
  float w1x[4], w1y[4], w1z[4], b1[4];  // dense1 rows k16 = 4*lg + e
  float b2[2][4];                        // dense2 bias of channels 16*mt + 4*lg + r
};
__device__ __forceinline__ void c3_lane_init(C3Lane& Q, const C3Params& P) {
  const int lane = threadIdx.x & 63;
  Q.li = lane & 15;
  Q.lg = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = Q.lg * 4 + e;
    Q.w1x[e] = P.w1[k * 3];
    Q.w1y[e] = P.w1[k * 3 + 1];
    Q.w1z[e] = P.w1[k * 3 + 2];
    Q.b1[e] = P.b1[k];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) Q.b2[mt][r] = P.b2[mt * 16 + Q.lg * 4 + r];
}

// h1 (lane: pixel li, k16 = 4*lg + e) and h2 (lane: pixel li, channels 16*mt + 4*lg + r) of one neighbour
__device__ __forceinline__ void c3_mlp(const C3Lds& L, const C3Lane& Q, const C3Nb& nb, float h1[4], f32x4 h2[2]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) h1[e] = selu_f(((nb.lx * Q.w1x[e] + nb.ly * Q.w1y[e]) + nb.lz * Q.w1z[e]) + Q.b1[e]);
  f32x4 pre[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 wa[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) wa[mt] = *(const f32x4*)(L.w2 + (mt * 16 + Q.li) * C3_W2S + Q.lg * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) pre[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][e], h1[e], pre[mt], 0, 0, 0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[mt][r] = selu_f(pre[mt][r] + Q.b2[mt][r]);
}

__global__ __launch_bounds__(256) void conv3d_fwd_kernel(const float4* __restrict__ geom, const float* __restrict__ wf,
                                                          C3Params P, const unsigned char* __restrict__ idx,
                                                          float* __restrict__ y, C3Dims d) {
  __shared__ C3Lds L;
  c3_load_weights(L, P);
  C3Lane Q;
  c3_lane_init(Q, P);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  const long ngroups = (total + C3_GP - 1) / C3_GP;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long)gridDim.x * 4) {
    const long i = grp * C3_GP + Q.li;
    const bool pv = i < total;
    const long ic = pv ? i : total - 1;
    const int ox = (int)(ic % d.wo), oy = (int)((ic / d.wo) % d.ho);
    const long tb = ic / ((long)d.wo * d.ho);
    const float4 ctr = geom[(tb * d.h * d.w + (long)(oy * d.stride) * d.w + ox * d.stride) * C3_TL];
    f32x4 agg[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    c3_build_desc(L.D[wave], geom, idx, d, ic, pv, oy, ox, tb, ctr, Q.li, Q.lg);
#pragma unroll 3
    for (int n = 0; n < C3_NB; ++n) {
      const C3Nb nb = c3_desc(L.D[wave], n, Q.li);
      f32x4 fv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      if (nb.foff >= 0) {
        fv[0] = *(const f32x4*)(wf + nb.foff + Q.lg * 4);
        fv[1] = *(const f32x4*)(wf + nb.foff + 16 + Q.lg * 4);
      }
      float h1[4];
      f32x4 h2[2];
      c3_mlp(L, Q, nb, h1, h2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) agg[mt] += h2[mt] * fv[mt];
    }
    __builtin_amdgcn_wave_barrier();  // (the next group's descriptors overwrite D)
    // y[px][c'] = selu(sum_c agg[px][c] * w[c][c'])
    f32x4 out[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* wr = L.w + (ms * 16 + Q.lg * 4 + r) * C3_WS + Q.li;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          out[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[mt * 16], agg[ms][r], out[mt], 0, 0, 0);
      }
    if (pv) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = selu_f(out[mt][r]);
        *(f32x4*)(y + i * C3_C + mt * 16 + Q.lg * 4) = o;
      }
    }
  }
}

// sum over the 16 lanes that share lg (xor-shuffles stay inside a 16-lane group)
__device__ __forceinline__ float c3_sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

__global__ __launch_bounds__(256, 2) void conv3d_bwd_kernel(const float4* __restrict__ geom,
                                                              const float* __restrict__ wf, C3Params P,
                                                              const unsigned char* __restrict__ idx,
                                                              const float* __restrict__ y, const float* __restrict__ gy,
                                                              float* __restrict__ gwf, float* __restrict__ part,
                                                              C3Dims d, float* __restrict__ stage) {
  // stage != nullptr: the feature-gradient rows are written to stage[(output pixel * 9 + neighbour) * 32 ..] (plain stores)
  // instead of being scattered with float atomics; conv3d_feat_gather_kernel sums them per source row in a fixed order
  __shared__ C3Lds L;
  __shared__ float red[4][C3_NPARAM];
  c3_load_weights(L, P);
  C3Lane Q;
  c3_lane_init(Q, P);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* X = L.X[wave];
  float* H = L.H[wave];
  long* F = L.F[wave];

  f32x4 accW[2][2], accW2[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    accW2[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) accW[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float db2[2][4], dw1x[4], dw1y[4], dw1z[4], db1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    db2[0][e] = db2[1][e] = 0.f;
    dw1x[e] = dw1y[e] = dw1z[e] = db1[e] = 0.f;
  }

  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  const long ngroups = (total + C3_GP - 1) / C3_GP;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long)gridDim.x * 4) {
    const long i = grp * C3_GP + Q.li;
    const bool pv = i < total;
    const long ic = pv ? i : total - 1;
    const int ox = (int)(ic % d.wo), oy = (int)((ic / d.wo) % d.ho);
    const long tb = ic / ((long)d.wo * d.ho);
    const float4 ctr = geom[(tb * d.h * d.w + (long)(oy * d.stride) * d.w + ox * d.stride) * C3_TL];
    // ---- recompute the forward aggregate (h1/h2 are recomputed again per neighbour below: cheaper than holding
    //      72 registers of h2 across the group, which spilled and halved the occupancy)
    f32x4 agg[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    c3_build_desc(L.D[wave], geom, idx, d, ic, pv, oy, ox, tb, ctr, Q.li, Q.lg);
#pragma unroll 3
    for (int n = 0; n < C3_NB; ++n) {
      const C3Nb nb = c3_desc(L.D[wave], n, Q.li);
      f32x4 fv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
      if (nb.foff >= 0) {
        fv[0] = *(const f32x4*)(wf + nb.foff + Q.lg * 4);
        fv[1] = *(const f32x4*)(wf + nb.foff + 16 + Q.lg * 4);
      }
      float h1[4];
      f32x4 h2[2];
      c3_mlp(L, Q, nb, h1, h2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) agg[mt] += h2[mt] * fv[mt];
    }
    // ---- output mix backward: gpre = gy * selu'(y);  dagg[c] = sum_c' gpre[c'] w[c][c']
    f32x4 gpre[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f}, yv = (f32x4){1.f, 1.f, 1.f, 1.f};
      if (pv) {
        g = *(const f32x4*)(gy + i * C3_C + mt * 16 + Q.lg * 4);
        yv = *(const f32x4*)(y + i * C3_C + mt * 16 + Q.lg * 4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) gpre[mt][r] = g[r] * act_grad_from_out(yv[r], DIS_ACT_SELU);
    }
    f32x4 dagg[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      f32x4 wa[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wa[mt] = *(const f32x4*)(L.w + (mt * 16 + Q.li) * C3_WS + ms * 16 + Q.lg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          dagg[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][r], gpre[ms][r], dagg[mt], 0, 0, 0);
    }
    // ---- dW[c][c'] += sum_px agg[px][c] gpre[px][c']: both operands transposed to "pixel on the k slot"
    f32x4 aggD[2], gpreD[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = agg[mt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) aggD[nt][r] = X[(Q.lg * 4 + r) * C3_XS + nt * 16 + Q.li];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = gpre[mt];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) gpreD[nt][r] = X[(Q.lg * 4 + r) * C3_XS + nt * 16 + Q.li];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          accW[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aggD[mt][r], gpreD[nt][r], accW[mt][nt], 0, 0, 0);

    // ---- neighbours (feature rows fetched one neighbour ahead)
    C3Nb nbn = c3_desc(L.D[wave], 0, Q.li);
    f32x4 fvn[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    if (nbn.foff >= 0) {
      fvn[0] = *(const f32x4*)(wf + nbn.foff + Q.lg * 4);
      fvn[1] = *(const f32x4*)(wf + nbn.foff + 16 + Q.lg * 4);
    }
#pragma unroll 1
    for (int n = 0; n < C3_NB; ++n) {
      const C3Nb nb = nbn;
      const f32x4 fv[2] = {fvn[0], fvn[1]};
      if (n + 1 < C3_NB) {
        nbn = c3_desc(L.D[wave], n + 1, Q.li);
        fvn[0] = fvn[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (nbn.foff >= 0) {
          fvn[0] = *(const f32x4*)(wf + nbn.foff + Q.lg * 4);
          fvn[1] = *(const f32x4*)(wf + nbn.foff + 16 + Q.lg * 4);
        }
      }
      float h1[4];
      f32x4 h2[2];
      c3_mlp(L, Q, nb, h1, h2);
      f32x4 dpre2[2], dfeat[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h2v = h2[mt][r];
          dfeat[mt][r] = dagg[mt][r] * h2v;
          dpre2[mt][r] = (dagg[mt][r] * fv[mt][r]) * act_grad_from_out(h2v, DIS_ACT_SELU);
          db2[mt][r] += dpre2[mt][r];
        }
      // d h1[k16] = sum_c w2[c][k16] dpre2[c]
      f32x4 dh1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dh1 = __builtin_amdgcn_mfma_f32_16x16x4f32(L.w2[(ms * 16 + Q.lg * 4 + r) * C3_W2S + Q.li], dpre2[ms][r], dh1, 0,
                                                     0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dpre1 = dh1[e] * act_grad_from_out(h1[e], DIS_ACT_SELU);
        db1[e] += dpre1;
        dw1x[e] += dpre1 * nb.lx;
        dw1y[e] += dpre1 * nb.ly;
        dw1z[e] += dpre1 * nb.lz;
      }
      // d w2[c][k16] += sum_px dpre2[px][c] h1[px][k16]
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = dpre2[mt];
      *(f32x4*)(H + Q.li * C3_HS + Q.lg * 4) = (f32x4){h1[0], h1[1], h1[2], h1[3]};
      __builtin_amdgcn_wave_barrier();
      f32x4 dpre2D[2];
      float h1D[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h1D[r] = H[(Q.lg * 4 + r) * C3_HS + Q.li];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dpre2D[nt][r] = X[(Q.lg * 4 + r) * C3_XS + nt * 16 + Q.li];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          accW2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpre2D[mt][r], h1D[r], accW2[mt], 0, 0, 0);
      // feature gradient: 128-B rows (2 pixels per wave instruction) of float atomics
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = dfeat[mt];
      if (Q.lg == 0) F[Q.li] = nb.foff;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < C3_GP / 2; ++j) {
        const int px = 2 * j + (lane >> 5), c = lane & 31;
        const long fo = F[px];
        if (fo >= 0) {
          if (stage) stage[((grp * C3_GP + px) * C3_NB + n) * C3_C + c] = X[px * C3_XS + c];
          else atomicAdd(gwf + fo + c, X[px * C3_XS + c]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- parameter gradients: per-wave results -> LDS -> one slab per block
  float* rw = red[wave];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = mt * 16 + Q.lg * 4 + r;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) rw[C3_OFF_W + c * C3_C + nt * 16 + Q.li] = accW[mt][nt][r];
      rw[C3_OFF_W2 + c * C3_H1 + Q.li] = accW2[mt][r];
      const float s = c3_sum16(db2[mt][r]);
      if (Q.li == 0) rw[C3_OFF_B2 + c] = s;
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = Q.lg * 4 + e;
    const float sb = c3_sum16(db1[e]), sx = c3_sum16(dw1x[e]), sy = c3_sum16(dw1y[e]), sz = c3_sum16(dw1z[e]);
    if (Q.li == 0) {
      rw[C3_OFF_B1 + k] = sb;
      rw[C3_OFF_W1 + k * 3] = sx;
      rw[C3_OFF_W1 + k * 3 + 1] = sy;
      rw[C3_OFF_W1 + k * 3 + 2] = sz;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < C3_NPARAM; t += blockDim.x)
    part[(long)blockIdx.x * C3_NPARAM + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// gparams[t] = sum over blocks: one wave per parameter, lanes stride over the block slabs, fixed-order wave reduction
__global__ __launch_bounds__(256) void c3_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ o,
                                                               int nblocks) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= C3_NPARAM) return;
  float s = 0.f;
  for (int k = lane; k < nblocks; k += 64) s += part[(long)k * C3_NPARAM + t];
  s = wave_sum(s);
  if (lane == 0) o[t] = s;
}

#define C3_BWD_BLOCKS 512

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
  hipLaunchKernelGGL(conv3d_select_kernel, dim3(dis_ew_grid(total, C3_SEL_T)), dim3(C3_SEL_T), 0, (hipStream_t)stream,
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
  int grid = dis_cdiv(dis_cdiv(total, C3_GP), 4);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(conv3d_fwd_kernel, dim3(grid), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, d);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" long dis_conv3d_knn_bwd_workspace(void) { return (long)C3_BWD_BLOCKS * C3_NPARAM; }

// grad_wf[row] (+)= sum of the staged rows of the (output pixel, neighbour) entries that selected source row `row`, in the
// order of the CSR lists (ascending entry id: fixed, so the result is bitwise reproducible).  8 threads per row, 4 channels each.
__global__ void conv3d_feat_gather_kernel(const float* __restrict__ stage, const int* __restrict__ offsets,
                                          const int* __restrict__ entries, float* __restrict__ gwf, long nsrc,
                                          int accumulate) {
  const long total = nsrc * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long d = i >> 3;
    const int g = (int)(i & 7);
    const int lo = offsets[d], hi = offsets[d + 1];
    float4* out = (float4*)(gwf + d * C3_C + g * 4);
    if (lo == hi) {
      if (!accumulate) *out = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    float4 acc = accumulate ? *out : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = lo; e < hi; ++e) {
      const float4 v = *(const float4*)(stage + (long)entries[e] * C3_C + g * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *out = acc;
  }
}
extern "C" long dis_conv3d_knn_bwd_stage(int tl, int bs, int h, int wd, int stride) {
  C3Dims d;
  if (c3_dims(&d, tl, bs, h, wd, stride) != DIS_OK) return -1;
  return (long)tl * bs * d.ho * d.wo * C3_NB * C3_C;
}

static int c3_bwd_run(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                      const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                      const float* y, const float* gy, float* grad_wf, float* gparams, float* workspace, const int* csr,
                      float* stage, int accumulate, int tl, int bs, int h, int wd, int stride, void* stream);
extern "C" int dis_conv3d_knn_bwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                  const float* dense2_w, const float* dense2_b, const float* w,
                                  const unsigned char* idx, const float* y, const float* gy, float* grad_wf,
                                  float* gparams, float* workspace, int tl, int bs, int h, int wd, int stride,
                                  void* stream) {
  return c3_bwd_run(geom, wf, dense1_w, dense1_b, dense2_w, dense2_b, w, idx, y, gy, grad_wf, gparams, workspace, nullptr,
                    nullptr, 1, tl, bs, h, wd, stride, stream);
}
// The same with a deterministic feature gradient: csr = dis_conv3d_csr_build(idx, ...) (layout_ops.hip), stage =
// dis_conv3d_knn_bwd_stage() floats of scratch.  accumulate = 0: grad_wf is written (rows nobody selected get zeros: no
// zero fill by the caller); 1: the selected rows are added to grad_wf's contents (shared gradient buffer).
extern "C" int dis_conv3d_knn_bwd_csr(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                      const float* dense2_w, const float* dense2_b, const float* w,
                                      const unsigned char* idx, const float* y, const float* gy, float* grad_wf,
                                      float* gparams, float* workspace, const int* csr, float* stage, int accumulate,
                                      int tl, int bs, int h, int wd, int stride, void* stream) {
  if (!csr || !stage) return DIS_ERR_NULL;
  return c3_bwd_run(geom, wf, dense1_w, dense1_b, dense2_w, dense2_b, w, idx, y, gy, grad_wf, gparams, workspace, csr, stage,
                    accumulate, tl, bs, h, wd, stride, stream);
}
static int c3_bwd_run(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                      const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx,
                      const float* y, const float* gy, float* grad_wf, float* gparams, float* workspace, const int* csr,
                      float* stage, int accumulate, int tl, int bs, int h, int wd, int stride, void* stream) {
  if (!geom || !wf || !dense1_w || !dense1_b || !dense2_w || !dense2_b || !w || !idx || !y || !gy || !grad_wf ||
      !gparams || !workspace)
    return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)tl * bs * d.ho * d.wo;
  C3Params P{dense1_w, dense1_b, dense2_w, dense2_b, w};
  int grid = dis_cdiv(dis_cdiv(total, C3_GP), 4);
  if (grid > C3_BWD_BLOCKS) grid = C3_BWD_BLOCKS;
  hipLaunchKernelGGL(conv3d_bwd_kernel, dim3(grid), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, gy, grad_wf,
                     workspace, d, stage);
  if (csr) {
    const long nsrc = (long)tl * bs * h * wd * C3_TL;
    hipLaunchKernelGGL(conv3d_feat_gather_kernel, dim3(dis_ew_grid(nsrc * 8, 256)), dim3(256), 0, s, (const float*)stage, csr,
                       csr + 2 * nsrc + 1, grad_wf, nsrc, accumulate);
  }
  hipLaunchKernelGGL(c3_param_reduce_kernel, dim3(dis_cdiv(C3_NPARAM, 4)), dim3(256), 0, s, (const float*)workspace,
                     gparams, grid);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
