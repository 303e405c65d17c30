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
__device__ __forceinline__ void c3_mlp(const float* w2s, const C3Lane& Q, const C3Nb& nb, float h1[4], f32x4 h2[2]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) h1[e] = selu_f(((nb.lx * Q.w1x[e] + nb.ly * Q.w1y[e]) + nb.lz * Q.w1z[e]) + Q.b1[e]);
  f32x4 pre[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 wa[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) wa[mt] = *(const f32x4*)(w2s + (mt * 16 + Q.li) * C3_W2S + Q.lg * 4);
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
                                                          float* __restrict__ y, C3Dims d, float* __restrict__ agg_out) {
  // agg_out (optional): the weighted feature sums in front of the 32x32 mix, kept for conv3d_bwd_cls_kernel
  __shared__ C3Lds L;
  c3_load_weights(L, P);
  C3Lane Q;
  c3_lane_init(Q, P);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const long total = (long)d.tl * d.bs * d.ho * d.wo;
  const long ngroups = (total + C3_GP - 1) / C3_GP;
  // (Measured and not kept, round 6: an XCD-contiguous group order - every XCD a contiguous eighth of the groups, every wave a
  //  contiguous run - so that the windows of vertically adjacent rows share their feature rows in ONE L2: 0.075 -> 0.100 ms at
  //  stride 2, 0.065 -> 0.083 ms at stride 1, same box.  The launch is latency-bound; the grid-stride walk spreads a wave's
  //  consecutive groups over the memory channels.)
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
      c3_mlp(L.w2, Q, nb, h1, h2);
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
        if (agg_out) *(f32x4*)(agg_out + i * C3_C + mt * 16 + Q.lg * 4) = agg[mt];
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
      c3_mlp(L.w2, Q, nb, h1, h2);
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
      c3_mlp(L.w2, Q, nb, h1, h2);
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
#ifdef C3_KO_MFMA
          if (r == 0 && ms == 0)
#endif
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
#ifdef C3_KO_MFMA
          if (r == 0)
#endif
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

extern "C" int dis_conv3d_knn_fwd_agg(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                      const float* dense2_w, const float* dense2_b, const float* w,
                                      const unsigned char* idx, float* y, float* agg, int tl, int bs, int h, int wd,
                                      int stride, void* stream) {
  if (!geom || !wf || !dense1_w || !dense1_b || !dense2_w || !dense2_b || !w || !idx || !y) return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)tl * bs * d.ho * d.wo;
  C3Params P{dense1_w, dense1_b, dense2_w, dense2_b, w};
  int grid = dis_cdiv(dis_cdiv(total, C3_GP), 4);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(conv3d_fwd_kernel, dim3(grid), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, d, agg);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_conv3d_knn_fwd(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                  const float* dense2_w, const float* dense2_b, const float* w,
                                  const unsigned char* idx, float* y, int tl, int bs, int h, int wd, int stride,
                                  void* stream) {
  return dis_conv3d_knn_fwd_agg(geom, wf, dense1_w, dense1_b, dense2_w, dense2_b, w, idx, y, nullptr, tl, bs, h, wd, stride,
                                stream);
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

// Branch-free SELU and SELU' (the ternary around an exp compiles to exec-mask control flow: 4 scalar instructions and a basic
// block per value - 12 values per neighbour) and explicit fused multiply-adds: the backward kernel is instruction-issue-bound and
// its results are tolerance-checked, so roundings need not repeat the forward's.
__device__ __forceinline__ float selu_sel(float x) {
  const float neg = __builtin_fmaf(SELU_SCALE_F * SELU_ALPHA_F, __builtin_amdgcn_exp2f(x * 1.4426950408889634f),
                                   -(SELU_SCALE_F * SELU_ALPHA_F));
  const float pos = SELU_SCALE_F * x;
  return x > 0.f ? pos : neg;
}
__device__ __forceinline__ float selu_grad_sel(float y) {
  const float neg = y + SELU_SCALE_F * SELU_ALPHA_F;
  return y > 0.f ? SELU_SCALE_F : neg;
}
struct __attribute__((aligned(16))) C3Lds2 {   // conv3d_bwd2_kernel
  float w[C3_C * C3_WS];
  float w2[C3_C * C3_W2S];
  float X[4][C3_GP * C3_XS];        // tiles of the 1st neighbour of a trip (and of the group's dW operands)
  float H[4][C3_GP * C3_HS];
  float X2[4][2][C3_GP * C3_XS];    // tiles of the 2nd / 3rd neighbour of a trip: the neighbours of a trip are independent chains
  float H2[4][2][C3_GP * C3_HS];
  float4 D[4][C3_NB * C3_GP];       // local coordinates of (neighbour n, pixel li) of a wave's group
  int R[4][C3_NB * C3_GP];          // row index of (neighbour n, pixel li), -1: zero-padded candidate / invalid lane
  float4 W1[C3_H1];                 // dense1 rows (wx, wy, wz, bias): LDS, not 24 registers
  float B2[C3_C];
};
// c3_mlp with the dense1 rows and the dense2 bias read from LDS instead of 24 registers
__device__ __forceinline__ void c3_mlp_lds(const C3Lds2& L, int li, int lg, const C3Nb& nb, float h1[4], f32x4 h2[2]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float4 wv = L.W1[lg * 4 + e];
    h1[e] = selu_sel(__builtin_fmaf(nb.lz, wv.z, __builtin_fmaf(nb.ly, wv.y, __builtin_fmaf(nb.lx, wv.x, wv.w))));
  }
  f32x4 pre[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 wa[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) wa[mt] = *(const f32x4*)(L.w2 + (mt * 16 + li) * C3_W2S + lg * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#ifdef C3_KO_MFMA   // (diagnostic: one product instead of four - WRONG results; what the exact-fp32 products cost)
      if (e == 0)
#endif
      pre[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][e], h1[e], pre[mt], 0, 0, 0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const f32x4 b = *(const f32x4*)(L.B2 + mt * 16 + lg * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[mt][r] = selu_sel(pre[mt][r] + b[r]);
  }
}

// ================================================================================================
// Backward, second form (round 3; the default): conv3d_bwd2_kernel<CLS>.
//   * The forward aggregate - needed for the gradient of the 32x32 mix only - is read back from the forward pass (`agg`, 128 B per
//     output pixel) instead of being recomputed: the first neighbour loop of conv3d_bwd_kernel and its 9 feature-row gathers are
//     gone (a third of the instructions of a kernel that is instruction-issue-bound).
//   * The dependency chain of a group is two levels deep: {centre, selection ids} -> {neighbour geometry, feature rows,
//     gradient rows}; the row indices go to LDS straight from the ids.
//   * CLS = true, the DETERMINISTIC form: the 3x3 windows of output pixels whose coordinates agree modulo cn (3 at stride 1, 2 at
//     stride 2) are disjoint, so within one such CLASS every feature-gradient row is touched by at most one (pixel, neighbour)
//     entry.  One launch per class: the rows are accumulated with plain 16-byte read-modify-writes - no float atomics, a fifth of
//     the scatter's instructions - and the cn*cn launches in stream order fix the summation order of every row (bitwise
//     reproducible, no index structure to build).
//   * CLS = false: one launch over all pixels, float-atomic scatter (cn = 1).
// Parameter gradients: per-wave accumulators over the wave's groups -> one slab per block -> fixed-order two-stage reduction.
// (Tried and dropped: a block of 3 waves per group, 3 neighbours each, to shorten a class launch's critical path - the per-group
//  work every wave repeats made it slower: 44 us per class launch of 1536 groups.)
// ================================================================================================
#ifndef C3D_CAP
#define C3D_CAP 768             // blocks per launch (at most)
#endif
#pragma clang fp contract(fast)  // (tolerance-checked gradients: fused multiply-adds from here on; the forward and the selection keep every rounding)

#ifdef C3_STAMP   // diagnostic build (scripts/diag/conv3d_bwd_modes.py --stamps): phase time stamps of one wave's first group
__device__ unsigned long long c3_stamps[64];
#define C3S(k) { if (blockIdx.x == C3_STAMP && threadIdx.x == 0 && first_) { __builtin_amdgcn_s_waitcnt(0); c3_stamps[k] = __builtin_amdgcn_s_memtime(); } }
extern "C" int dis_c3_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3_stamps), sizeof(c3_stamps));
}
#else
#define C3S(k)
#endif
#ifndef C3D_SB
#define C3D_SB 0
#endif
// LDS accesses of one wave are processed in issue order, so a store -> load hand-over between its lanes needs no wait, only the
// compiler must not reorder the two: a fence at wavefront scope orders memory operations and leaves ALU / MFMA free to move
#define C3_LDS_ORDER()                                       \
  {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  }
#ifndef C3D_WPE
#define C3D_WPE 2               // waves per SIMD the register allocation aims at
#endif
#ifndef C3D_PD
#define C3D_PD 3                // neighbours the row loads run ahead of the math
#endif
union C3DShared {
  C3Lds2 L;
  float red[4][C3_NPARAM];      // (end of the kernel only: aliases the working tiles)
};
template <bool CLS>
__global__ __launch_bounds__(256, C3D_WPE) void conv3d_bwd2_kernel(const float4* __restrict__ geom, const float* __restrict__ wf,
                                                              C3Params P, const unsigned char* __restrict__ idx,
                                                              const float* __restrict__ y, const float* __restrict__ aggp,
                                                              const float* __restrict__ gy, float* __restrict__ gwf,
                                                              float* __restrict__ part, C3Dims d, int cn, int cy, int cx) {
  __shared__ C3DShared S;
  C3Lds2& L = S.L;
  bool first_ = true;
  (void)first_;
  C3S(0)
  {  // weights -> LDS, every load of a thread in flight before the first store
    float wv[4], w2v[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = P.w[threadIdx.x + k * 256];
#pragma unroll
    for (int k = 0; k < 2; ++k) w2v[k] = P.w2[threadIdx.x + k * 256];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * 256;
      L.w[(i >> 5) * C3_WS + (i & 31)] = wv[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = threadIdx.x + k * 256;
      L.w2[(i >> 4) * C3_W2S + (i & 15)] = w2v[k];
    }
  }
  if (threadIdx.x < C3_H1) L.W1[threadIdx.x] = make_float4(P.w1[threadIdx.x * 3], P.w1[threadIdx.x * 3 + 1], P.w1[threadIdx.x * 3 + 2],
                                                           P.b1[threadIdx.x]);
  if (threadIdx.x >= 64 && threadIdx.x < 64 + C3_C) L.B2[threadIdx.x - 64] = P.b2[threadIdx.x - 64];
  struct { int li, lg; } Q;
  Q.li = threadIdx.x & 15;
  Q.lg = (threadIdx.x & 63) >> 4;
  __syncthreads();
  C3S(1)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* X = L.X[wave];
  float4* D = L.D[wave];
  int* R = L.R[wave];

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

  // (32-bit index arithmetic: the launcher has checked that every row index fits an int)
  const int ny = (d.ho - cy + cn - 1) / cn, nx = (d.wo - cx + cn - 1) / cn;
  const int total = d.tl * d.bs * ny * nx;
  const int ngroups = (total + C3_GP - 1) / C3_GP;
  const int hw = d.h * d.w;
  const __amdgpu_buffer_rsrc_t wf_rs = bx_rsrc(wf, (unsigned)d.tl * d.bs * hw * (C3_TL * C3_C * 4u));
  const __amdgpu_buffer_rsrc_t gwf_rs = bx_rsrc(gwf, (unsigned)d.tl * d.bs * hw * (C3_TL * C3_C * 4u));
  // ---- one group ahead: the selection ids of a group's pixels.  ids -> {neighbour geometry, feature rows, gradient rows} were two
  // DEPENDENT global round trips in front of everything else a group does (a wave has ~3 groups per launch and one partner wave on
  // its SIMD: nothing hid them).  The next group's ids are requested before the neighbour loop of the current one - unconditionally
  // (the last group asks for its own again): a load under a condition inside the rolled loop would make every wait behind it
  // conservative - so a group starts with its ids in registers and everything that depends on them is ONE round trip.
  auto decode = [&](int g_, bool& pv_, int& i_, int& oy_, int& ox_, int& tb_) {
    const int il = g_ * C3_GP + Q.li;
    pv_ = il < total;
    const unsigned ilc = pv_ ? il : total - 1;
    const unsigned t1 = ilc / (unsigned)nx;
    ox_ = cx + cn * (int)(ilc - t1 * nx);
    tb_ = (int)(t1 / (unsigned)ny);
    oy_ = cy + cn * (int)(t1 - (unsigned)tb_ * ny);
    i_ = (tb_ * d.ho + oy_) * d.wo + ox_;
  };
  auto load_ids = [&](bool pv_, int i_, unsigned (&id_)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int n = Q.lg + 4 * k;
      id_[k] = 0xffu;
      if (n < C3_NB && pv_) id_[k] = idx[i_ * C3_NB + n];
    }
  };
  const int gstep = gridDim.x * 4;
  bool pvN = false;
  int iN = 0, oyN = 0, oxN = 0, tbN = 0;
  unsigned idN[3] = {0xffu, 0xffu, 0xffu};
  if ((int)(blockIdx.x * 4 + wave) < ngroups) {
    decode(blockIdx.x * 4 + wave, pvN, iN, oyN, oxN, tbN);
    load_ids(pvN, iN, idN);
  }
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gstep) {
    const bool pv = pvN;
    const int i = iN, oy = oyN, ox = oxN, tb = tbN;
    const float4 ctr = geom[(long)(tb * hw + oy * d.stride * d.w + ox * d.stride) * C3_TL];
    // ---- rows of the 9 neighbours (lane (li, lg): neighbours lg, lg + 4, lg + 8 of pixel li), then their geometry
    int myrow[3];
    float4 qn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int n = Q.lg + 4 * k;
      myrow[k] = -1;
      if (n < C3_NB && pv) {
        const unsigned id = idN[k];
        const unsigned tap = id >> 2, slot = id & 3, ty = (tap * 11u) >> 5;   // (C3_TL = 4; tap / 3 for tap < 9)
        const int iy = oy * d.stride - 1 + (int)ty, ix = ox * d.stride - 1 + (int)(tap - 3 * ty);
        if ((unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w) myrow[k] = (tb * hw + iy * d.w + ix) * C3_TL + (int)slot;
      }
      if (n < C3_NB) R[n * C3_GP + Q.li] = myrow[k];
    }
    C3S(2)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      qn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (myrow[k] >= 0) qn[k] = geom[myrow[k]];
    }
    // ---- the next group's ids (this group's own again when it is the wave's last)
    {
      const int gn = grp + gstep < ngroups ? grp + gstep : grp;
      decode(gn, pvN, iN, oyN, oxN, tbN);
      load_ids(pvN, iN, idN);
    }
    // ---- gy, y, agg of the group
    f32x4 gpre[2], aggv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f}, yv = (f32x4){1.f, 1.f, 1.f, 1.f};
      aggv[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (pv) {
        g = *(const f32x4*)(gy + i * C3_C + mt * 16 + Q.lg * 4);
        yv = *(const f32x4*)(y + i * C3_C + mt * 16 + Q.lg * 4);
        aggv[mt] = *(const f32x4*)(aggp + i * C3_C + mt * 16 + Q.lg * 4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) gpre[mt][r] = g[r] * selu_grad_sel(yv[r]);
    }
    __builtin_amdgcn_wave_barrier();
    C3S(3)
    // ---- rows of the first C3D_PD neighbours in flight: the feature rows (matrix layout) and, CLS, the gradient rows to update
    //      (CLS: same layout - a lane updates the 16 bytes of the row it holds the gradient of; 16 rows x 64 B per instruction)
    f32x4 fvq[C3D_PD][2], preq[C3D_PD][2];
    unsigned rowq[C3D_PD];
#define C3D_FETCH(n, q)                                                                                    \
  {                                                                                                        \
    const int rn_ = R[(n) * C3_GP + Q.li];                                                                 \
    rowq[q] = rn_ >= 0 ? (unsigned)rn_ * (C3_C * 4u) + Q.lg * 16u : BX_OOB;  /* byte offset; out of range: loads 0, store dropped */ \
    fvq[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wf_rs, rowq[q], 0, 0));     \
    fvq[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wf_rs, rowq[q], 64, 0));    \
    if (CLS) {                                                                                             \
      preq[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gwf_rs, rowq[q], 0, 0));  \
      preq[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gwf_rs, rowq[q], 64, 0)); \
    }                                                                                                      \
  }
#pragma unroll
    for (int n = 0; n < C3D_PD; ++n) C3D_FETCH(n, n)
    // ---- local coordinates (zero-padded / invalid candidates: 0 - centre: the reference pads xyz with zeros)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int n = Q.lg + 4 * k;
      if (n < C3_NB) D[n * C3_GP + Q.li] = make_float4(qn[k].x - ctr.x, qn[k].y - ctr.y, qn[k].z - ctr.z, 0.f);
    }
    C3S(4)
    // ---- output mix backward: dagg[c] = sum_c' gpre[c'] w[c][c']
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
    {
      f32x4 aggD[2], gpreD[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = aggv[mt];
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
    }
    C3S(5)
    // ---- neighbours, C3D_PD per trip of the rolled loop: queue slot q is consumed by neighbour nb0 + q and refilled with the
    //      rows of neighbour nb0 + q + C3D_PD right after (the scheduling barrier keeps the refill from being hoisted)
    static_assert(C3_NB % C3D_PD == 0, "queue depth must divide the neighbour count");
#pragma unroll 1
    for (int nb0 = 0; nb0 < C3_NB; nb0 += C3D_PD) {
#pragma unroll
    for (int q = 0; q < C3D_PD; ++q) {
      const int n = nb0 + q;
      const f32x4 fv[2] = {fvq[q][0], fvq[q][1]};
      const f32x4 pre[2] = {preq[q][0], preq[q][1]};
      const unsigned rowi = rowq[q];
      (void)pre;
      C3S(8 + 4 * n)
      if (C3D_SB) __builtin_amdgcn_sched_barrier(0);
      if (n + C3D_PD < C3_NB) C3D_FETCH(n + C3D_PD, q)
      if (C3D_SB) __builtin_amdgcn_sched_barrier(0);
      float* X = q == 0 ? L.X[wave] : L.X2[wave][q - 1];   // (per-slot tiles: the neighbours of a trip are independent chains)
      float* H = q == 0 ? L.H[wave] : L.H2[wave][q - 1];
      C3Nb nb;
      {
        const float4 v = D[n * C3_GP + Q.li];
        nb.lx = v.x;
        nb.ly = v.y;
        nb.lz = v.z;
      }
      float h1[4];
      f32x4 h2[2];
      C3S(9 + 4 * n)
      c3_mlp_lds(L, Q.li, Q.lg, nb, h1, h2);
      C3S(10 + 4 * n)
      f32x4 dpre2[2], dfeat[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h2v = h2[mt][r];
          dfeat[mt][r] = dagg[mt][r] * h2v;
          dpre2[mt][r] = (dagg[mt][r] * fv[mt][r]) * selu_grad_sel(h2v);
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
        const float dpre1 = dh1[e] * selu_grad_sel(h1[e]);
        db1[e] += dpre1;
        dw1x[e] = __builtin_fmaf(dpre1, nb.lx, dw1x[e]);
        dw1y[e] = __builtin_fmaf(dpre1, nb.ly, dw1y[e]);
        dw1z[e] = __builtin_fmaf(dpre1, nb.lz, dw1z[e]);
      }
      // d w2[c][k16] += sum_px dpre2[px][c] h1[px][k16]
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = dpre2[mt];
      *(f32x4*)(H + Q.li * C3_HS + Q.lg * 4) = (f32x4){h1[0], h1[1], h1[2], h1[3]};
      C3_LDS_ORDER();
      f32x4 dpre2D[2];
      float h1D[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h1D[r] = H[(Q.lg * 4 + r) * C3_HS + Q.li];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dpre2D[nt][r] = X[(Q.lg * 4 + r) * C3_XS + nt * 16 + Q.li];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          accW2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dpre2D[mt][r], h1D[r], accW2[mt], 0, 0, 0);
      C3S(11 + 4 * n)
      if (CLS) {
        // feature gradient: the lane adds its 2 x 16 bytes to the row (no other entry of this launch touches the row)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pre[0] + dfeat[0]), gwf_rs, rowi, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pre[1] + dfeat[1]), gwf_rs, rowi, 64, 0);
      } else {
        // feature gradient rows: matrix layout -> row layout through LDS, 128-B rows of float atomics
        C3_LDS_ORDER();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) *(f32x4*)(X + Q.li * C3_XS + mt * 16 + Q.lg * 4) = dfeat[mt];
        C3_LDS_ORDER();
#pragma unroll
        for (int j = 0; j < C3_GP / 2; ++j) {
          const int px = 2 * j + (lane >> 5), c = lane & 31;
          const int ro = R[n * C3_GP + px];
          if (ro >= 0) atomicAdd(gwf + (long)ro * C3_C + c, X[px * C3_XS + c]);
        }
      }
    }
    C3_LDS_ORDER();  // (the next trip's tiles)
    }
    __builtin_amdgcn_wave_barrier();  // (the next group's rows / descriptors overwrite R, D)
    C3S(6)
    first_ = false;
  }
  C3S(7)

  // ---- parameter gradients: per-wave results -> LDS (aliases the working tiles) -> one slab per block
  __syncthreads();
  float (*red)[C3_NPARAM] = S.red;
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

// Fixed-order sum of the block slabs, two stages.  Stage 1: block (chunk of 64 parameters, slice s of C3D_RS) sums the slabs
// k = s, s + C3D_RS, ... with coalesced 256-byte reads, 4 waves interleaved, combined in wave order; stage 2 sums the slices.
#define C3D_RS 32
#define C3D_NCH ((C3_NPARAM + 63) / 64)
__global__ __launch_bounds__(256) void c3d_reduce1_kernel(const float* __restrict__ part, float* __restrict__ mid, int nslabs) {
  __shared__ float sm[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 64 + lane, s = blockIdx.y;
  float a = 0.f;
  if (t < C3_NPARAM)
    for (int k = s + C3D_RS * wave; k < nslabs; k += C3D_RS * 4) a += part[(long)k * C3_NPARAM + t];
  sm[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && t < C3_NPARAM) mid[(long)s * C3_NPARAM + t] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}
__global__ __launch_bounds__(64) void c3d_reduce2_kernel(const float* __restrict__ mid, float* __restrict__ o) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= C3_NPARAM) return;
  float a = 0.f;
#pragma unroll 8
  for (int s = 0; s < C3D_RS; ++s) a += mid[(long)s * C3_NPARAM + t];
  o[t] = a;
}

static int c3d_grids(const C3Dims& d, int tl, int bs, int cn, int* grids) {
  int tot = 0, k = 0;
  for (int cy = 0; cy < cn; ++cy)
    for (int cx = 0; cx < cn; ++cx, ++k) {
      const long tc = (long)tl * bs * ((d.ho - cy + cn - 1) / cn) * ((d.wo - cx + cn - 1) / cn);
      long g = ((tc + C3_GP - 1) / C3_GP + 3) / 4;
      if (g > C3D_CAP) g = C3D_CAP;
      grids[k] = (int)g;  // (0: the class is empty - maps narrower than cn)
      tot += (int)g;
    }
  return tot;
}
extern "C" long dis_conv3d_knn_bwd_det_workspace(int tl, int bs, int h, int wd, int stride) {
  C3Dims d;
  if (c3_dims(&d, tl, bs, h, wd, stride) != DIS_OK) return -1;
  int grids[9];
  const int tot = c3d_grids(d, tl, bs, stride == 1 ? 3 : 2, grids);
  return ((long)(tot > C3D_CAP ? tot : C3D_CAP) + C3D_RS) * C3_NPARAM;
}
static int c3_bwd2_run(bool det, const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                       const float* dense2_w, const float* dense2_b, const float* w, const unsigned char* idx, const float* y,
                       const float* agg, const float* gy, float* grad_wf, float* gparams, float* workspace, int tl, int bs, int h,
                       int wd, int stride, void* stream) {
  if (!geom || !wf || !dense1_w || !dense1_b || !dense2_w || !dense2_b || !w || !idx || !y || !agg || !gy || !grad_wf ||
      !gparams || !workspace)
    return DIS_ERR_NULL;
  C3Dims d;
  int rc = c3_dims(&d, tl, bs, h, wd, stride);
  if (rc != DIS_OK) return rc;
  if ((long)tl * bs * h * wd * C3_TL >= (1L << 31)) return DIS_ERR_BAD_SHAPE;  // (row indices are ints)
  // wf / grad_wf are addressed through ONE buffer descriptor each, 32-bit byte offsets, and BX_OOB (2 GiB) is the offset that
  // drops a lane: past 2 GiB the drop offset would land INSIDE the tensor (padded candidates would load real rows and, in the
  // class-ordered form, store into one).  Same bound as the conv launchers; the caller's alternative is dis_conv3d_knn_bwd.
  if ((long)tl * bs * h * wd * C3_TL * C3_C * 4L >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  C3Params P{dense1_w, dense1_b, dense2_w, dense2_b, w};
  int grids[9], tot;
  if (det) {
    const int cn = stride == 1 ? 3 : 2;
    tot = c3d_grids(d, tl, bs, cn, grids);
    int k = 0, base = 0;
    for (int cy = 0; cy < cn; ++cy)
      for (int cx = 0; cx < cn; ++cx, ++k) {
        if (grids[k] == 0) continue;
        hipLaunchKernelGGL(conv3d_bwd2_kernel<true>, dim3(grids[k]), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, agg, gy,
                           grad_wf, workspace + (long)base * C3_NPARAM, d, cn, cy, cx);
        base += grids[k];
      }
  } else {
    tot = c3d_grids(d, tl, bs, 1, grids);
    hipLaunchKernelGGL(conv3d_bwd2_kernel<false>, dim3(tot), dim3(256), 0, s, (const float4*)geom, wf, P, idx, y, agg, gy, grad_wf,
                       workspace, d, 1, 0, 0);
  }
  float* mid = workspace + (long)(tot > C3D_CAP ? tot : C3D_CAP) * C3_NPARAM;
  hipLaunchKernelGGL(c3d_reduce1_kernel, dim3(C3D_NCH, C3D_RS), dim3(256), 0, s, (const float*)workspace, mid, tot);
  hipLaunchKernelGGL(c3d_reduce2_kernel, dim3(C3D_NCH), dim3(64), 0, s, (const float*)mid, gparams);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_conv3d_knn_bwd_det(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                      const float* dense2_w, const float* dense2_b, const float* w,
                                      const unsigned char* idx, const float* y, const float* agg, const float* gy,
                                      float* grad_wf, float* gparams, float* workspace, int tl, int bs, int h, int wd,
                                      int stride, void* stream) {
  return c3_bwd2_run(true, geom, wf, dense1_w, dense1_b, dense2_w, dense2_b, w, idx, y, agg, gy, grad_wf, gparams, workspace, tl,
                     bs, h, wd, stride, stream);
}
extern "C" int dis_conv3d_knn_bwd_agg(const float* geom, const float* wf, const float* dense1_w, const float* dense1_b,
                                      const float* dense2_w, const float* dense2_b, const float* w,
                                      const unsigned char* idx, const float* y, const float* agg, const float* gy,
                                      float* grad_wf, float* gparams, float* workspace, int tl, int bs, int h, int wd,
                                      int stride, void* stream) {
  return c3_bwd2_run(false, geom, wf, dense1_w, dense1_b, dense2_w, dense2_b, w, idx, y, agg, gy, grad_wf, gparams, workspace, tl,
                     bs, h, wd, stride, stream);
}
