// DispNetS with bf16 ACTIVATION STORAGE (BASELINE config 2: "DIS-SF bs=8, full-res default-pattern synthetic, bf16"):
// every nhwc feature map of the encoder-decoder lives in HBM as bf16, the parameters, the gradients of the parameters,
// the Adam state, the disparities and all losses stay fp32.  A convolution is then ONE bf16 x bf16 product per MAC on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the fp32-result path, conv_gen.hip / conv2d.hip, spends six), and
// moves half the bytes.  Results are NOT fp32-accurate: tests compare with the fp32 oracle under a stated looser bound
// (tests/test_sf_bf16_gpu.py) and bench.py reports the mode under its own label.
//
//   convb_fwd_kernel<BN, XB, YB>   the forward-like streaming implicit GEMM of conv_gen.hip (same GenArgs addressing: a tap is
//       an arbitrary (dy,dx) offset, the output grid may be strided, so one kernel runs conv, both input gradients, the
//       transposed conv and its input gradient), 128 pixels x BN couts per workgroup, k-step = 32 channels of one tap;
//       x is bf16 (XB) or fp32 (network input, head gradient), y is bf16 (YB) or fp32 (head pre-activation).
//       The weights are the MFMA's A operand and the pixels its B operand, so a lane ends up with 4 consecutive output
//       channels of one pixel: one 8-byte (bf16) / 16-byte (fp32) store.
//   convb_wgrad_kernel<MTW, NTW, XB, GB>   weight gradient: split-K slabs over pixel ranges, fp32 MFMA on the values the
//       bf16 tensors hold (exact products of bf16 values, fp32 accumulation), fp64 slab sums.
//   element-wise helpers on bf16 nhwc tensors (activation gradient, channel copies, column sums).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

#define CB_MAXTAPS 49
#define CB_BM 128
#define CB_CK 32
#define CB_AS 40  // LDS pixel stride of the A tile (16-bit units): 32 channels + 8 pad (16 consecutive pixels: 16 distinct slots)

__device__ __forceinline__ unsigned cb_pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32: round to nearest even
}
__device__ __forceinline__ float cb_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float cb_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

struct GenArgsB {
  const void* x;
  const bf16_t* w;  // packed [tap][chunk][nblk][lg][BN][8]
  const float* bias;
  void* y;
  int n, hin, win, ldx, xoff, cin;  // channels [xoff, xoff+cin) of a pixel of ldx ELEMENTS
  int nchunk;                       // ceil(cin / 32)
  int hv, wv, S;
  int hf, wf, ldy, yoff, cout;
  int osy, ooy, osx, oox;
  int act, ntaps, nblk;
  short tdy[CB_MAXTAPS], tdx[CB_MAXTAPS];
  unsigned x_bytes;
  // halo form (convb_halo_kernel): first tap offsets, halo extent, LDS pixel stride, taps per k-step, k-steps, k-steps per
  // weight group, tile grid
  int dy0, dx0, HR, HC, PS, TP, nks, GT, tiles_x, tiles_y;
  int trh;         // tile rows: 8 (2 per wave) or 16 (4 per wave)
  int nph;         // output phases of one launch: 1, or the 4 parity classes of a stride-2 transposed form (see convb_halo_kernel, PH)
  unsigned char ph_k0[4], ph_kn[4], ph_oy[4], ph_ox[4];
  int kc;          // column walk: taps ordered dx-major / dy ascending, a weight group = one column of kc taps (0: generic walk)
  int res, wsz16;  // weights of a cout block resident in LDS (all chunks); 16-bit words of the weight region in front of the halo
  // convb_fwd128_kernel, split-K (small maps: fewer workgroups than CUs, each with a long chain of dependent stages):
  // blockIdx.y = split takes the stages [nk * split / ksplit, nk * (split + 1) / ksplit) and leaves its raw fp32 sums in
  // skpart[split][m][nblk * BN]; convb_splitk_reduce_kernel adds the splits in a fixed order, then bias + activation + store
  int ksplit;
  float* skpart;
  long skcap;  // floats available at skpart
};

template <int BN, bool XB, bool YB>
__global__ __launch_bounds__(256, 2) void convb_fwd_kernel(GenArgsB a) {
  constexpr int NT = BN / 16;
  constexpr int A_U16 = CB_BM * CB_AS, B_U16 = 4 * BN * 8;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (A_U16 + B_U16)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int M = a.n * a.hv * a.wv;
  const int nb = blockIdx.x % a.nblk, m0 = (blockIdx.x / a.nblk) * CB_BM;

  // loader role: thread owns channel group pq (8 channels) of pixels p0 and p0 + 64 of the tile
  const int pq = tid & 3, p0 = tid >> 2;
  long pbase[2];
  int piy[2], pix[2];
  bool pval[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + p0 + j * 64;
    pval[j] = m < M;
    const int mm = pval[j] ? m : 0;
    const int vx = mm % a.wv, t = mm / a.wv, vy = t % a.hv, nn = t / a.hv;
    pbase[j] = (long)nn * a.hin * a.win;
    piy[j] = vy * a.S;
    pix[j] = vx * a.S;
  }
  // ring of three register sets; taps outside the image / channel groups beyond cin get an out-of-range buffer offset
  u32x4 ra[3][2], ra2[3][XB ? 1 : 2], rb[3];
  const u32x4* wq = (const u32x4*)a.w;
  const int nk = a.ntaps * a.nchunk;
  const bool wload = tid < 4 * BN;  // 16-byte vectors of the B tile
  int ptap = 0, pchunk = 0, pnext = 0;
  auto prefetch = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    if (pnext >= nk) return;
    const int dy = a.tdy[ptap], dx = a.tdx[ptap];
    const int c = pchunk * CB_CK + pq * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = piy[j] + dy, ix = pix[j] + dx;
      const bool ok = pval[j] && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      const long e = (pbase[j] + (long)iy * a.win + ix) * a.ldx + a.xoff + c;
      if (XB) {
        const unsigned off = (ok && c < a.cin) ? (unsigned)(e * 2) : BX_OOB;
        ra[set][j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), off, 0, 0);
      } else {  // fp32 input: two 16-byte loads of 4 channels each
        const unsigned o0 = (ok && c < a.cin) ? (unsigned)(e * 4) : BX_OOB;
        const unsigned o1 = (ok && c + 4 < a.cin) ? (unsigned)(e * 4 + 16) : BX_OOB;
        ra[set][j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), o0, 0, 0);
        ra2[set][XB ? 0 : j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), o1, 0, 0);
      }
    }
    const long wb = ((long)(ptap * a.nchunk + pchunk) * a.nblk + nb) * (4 * BN);
    rb[set] = wq[wb + (wload ? tid : 0)];
    ++pnext;
    if (++pchunk == a.nchunk) {
      pchunk = 0;
      ++ptap;
    }
  };
  auto stage = [&](auto setc, int buf) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    unsigned short* A = smem + buf * (A_U16 + B_U16);
    unsigned short* B = A + A_U16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4 v = ra[set][j];
      if (!XB) {
        const u32x4 q = ra2[set][XB ? 0 : j];
        v = (u32x4){cb_pack2(__uint_as_float(v[0]), __uint_as_float(v[1])), cb_pack2(__uint_as_float(v[2]), __uint_as_float(v[3])),
                    cb_pack2(__uint_as_float(q[0]), __uint_as_float(q[1])), cb_pack2(__uint_as_float(q[2]), __uint_as_float(q[3]))};
      }
      *(u32x4*)(A + (p0 + j * 64) * CB_AS + pq * 8) = v;
    }
    if (wload) ((u32x4*)B)[tid] = rb[set];
  };

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  prefetch(S0{});
  prefetch(S1{});
  prefetch(S2{});
  stage(S0{}, 0);
  __syncthreads();
  auto body = [&](int s, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    prefetch(setc);
    const unsigned short* A = smem + (s & 1) * (A_U16 + B_U16);
    const unsigned short* B = A + A_U16;
    s16x8 fa[2], fb[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) fa[mt] = *(const s16x8*)(A + (wave * 32 + mt * 16 + li) * CB_AS + lg * 8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const s16x8*)(B + (lg * BN + nt * 16 + li) * 8);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[nt]),
                                                             __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
    stage(std::integral_constant<int, (set + 1) % 3>{}, (s + 1) & 1);
    __syncthreads();
  };
  for (int ks = 0; ks < nk; ks += 3) {
    body(ks, S0{});
    if (ks + 1 < nk) body(ks + 1, S1{});
    if (ks + 2 < nk) body(ks + 2, S2{});
  }

  // epilogue: lane (li, lg) holds couts nb*BN + nt*16 + lg*4 + {0..3} of pixel m0 + wave*32 + mt*16 + li
  auto emit = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + wave * 32 + mt * 16 + li;
      if (m >= M) continue;
      const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
      const long pe = (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = nb * BN + nt * 16 + lg * 4;
        if (co >= a.cout) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b = (a.bias && co + r < a.cout) ? a.bias[co + r] : 0.f;
          o[r] = act_apply(acc[mt][nt][r] + b, ACT);
        }
        if (co + 4 <= a.cout) {
          if (YB) *(uint2*)((bf16_t*)a.y + pe + co) = make_uint2(cb_pack2(o[0], o[1]), cb_pack2(o[2], o[3]));
          else *(float4*)((float*)a.y + pe + co) = make_float4(o[0], o[1], o[2], o[3]);
        } else {  // ragged last group (cout not a multiple of 4: the 1-channel heads)
          for (int r = 0; r < 4 && co + r < a.cout; ++r) {
            if (YB) ((bf16_t*)a.y)[pe + co + r] = (bf16_t)(cb_pack2(o[r], 0.f) & 0xffffu);
            else ((float*)a.y)[pe + co + r] = o[r];
          }
        }
      }
    }
  };
  if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
  else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
  else emit(std::integral_constant<int, DIS_ACT_NONE>{});
}

// The same streaming GEMM with 128 channels of one tap per stage, for the deep layers (>= 128 input channels on maps of a few
// thousand pixels in all): with 32 channels a stage is 8 MFMAs per wave and 12 KB per workgroup, far less than a memory round
// trip covers, and the layer runs at the latency of its 288 dependent stages (1024 -> 512 at 16 x 14: 0.8 us per stage).  Here
// a stage is 4 sub-steps (32 MFMAs per wave, 48 KB per workgroup) and TWO stages are in flight in registers while a third is
// multiplied from a single LDS buffer.  Weights: the packed layout of convb_pack_kernel (sub-step s of a stage = 32-channel
// chunk 4 stage + s).  x and y bf16.
#define CB_AS4 136  // LDS pixel stride of the 128-channel A tile (16-bit units): 128 + 8 (16 consecutive pixels: 16 distinct slots)
// NW waves = 32 NW pixels x BN couts per workgroup: 4 x 64 by default; 8 x 128 (round 4, as conv_gen.hip's convg2_fwd_kernel<128, 8>)
// for layers with >= 128 channels on both sides whose launch still fills the device - twice the MACs per operand byte from L2.
template <int BN, int NW = 4>
struct Cb128Cfg {
  static constexpr int NTHR = 64 * NW, BM = 32 * NW;
  static constexpr int A_U16 = BM * CB_AS4, B_U16 = 4 * (4 * BN * 8);
  static constexpr int LDS_BYTES = (A_U16 + B_U16) * 2;
  static_assert(4 * BN <= NTHR, "one 16-byte vector of a 32-channel weight block per thread");
};
template <int BN, bool YB, int NW = 4>
__global__ __launch_bounds__(64 * NW) void convb_fwd128_kernel(GenArgsB a) {
  using CF = Cb128Cfg<BN, NW>;
  constexpr int NT = BN / 16, BM = CF::BM, PSTEP = CF::NTHR / 4;   // PSTEP: pixels one loader pass covers
  constexpr int A_U16 = CF::A_U16;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* A = smem;
  unsigned short* B = smem + A_U16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int M = a.n * a.hv * a.wv;
  const int nb = blockIdx.x % a.nblk, m0 = (blockIdx.x / a.nblk) * BM;
  // loader role: channel group pq (8 channels of every 32-channel sub-step) of pixels p0 and p0 + PSTEP
  const int pq = tid & 3, p0 = tid >> 2;
  long pbase[2];
  int piy[2], pix[2];
  bool pval[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + p0 + j * PSTEP;
    pval[j] = m < M;
    const int mm = pval[j] ? m : 0;
    const int vx = mm % a.wv, t = mm / a.wv, vy = t % a.hv, nn = t / a.hv;
    pbase[j] = (long)nn * a.hin * a.win;
    piy[j] = vy * a.S;
    pix[j] = vx * a.S;
  }
  const int nst = (a.nchunk + 3) >> 2;  // stages per tap
  const int nk_all = a.ntaps * nst;
  const int k0 = (int)((long)nk_all * blockIdx.y / a.ksplit), nk = (int)((long)nk_all * (blockIdx.y + 1) / a.ksplit);
  u32x4 ra[2][2][4], rb[2][4];
  const u32x4* wq = (const u32x4*)a.w;
  const int bvec = 4 * BN;            // 16-byte vectors of one 32-channel weight block
  const bool wload = tid < bvec;
  int ptap = k0 / nst, pst = k0 % nst, pnext = k0;
  auto prefetch = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    if (pnext >= nk) return;
    const int dy = a.tdy[ptap], dx = a.tdx[ptap];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = piy[j] + dy, ix = pix[j] + dx;
      const bool ok = pval[j] && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      const long e = (pbase[j] + (long)iy * a.win + ix) * a.ldx + a.xoff + pst * 128 + pq * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = pst * 128 + s * 32 + pq * 8;
        ra[set][j][s] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes),
                                                              (ok && c < a.cin) ? (unsigned)((e + s * 32) * 2) : BX_OOB, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int chunk = pst * 4 + s;
      const long wb = ((long)(ptap * a.nchunk + (chunk < a.nchunk ? chunk : 0)) * a.nblk + nb) * bvec;
      u32x4 v = wq[wb + (wload ? tid : 0)];
      if (chunk >= a.nchunk) v = (u32x4){0u, 0u, 0u, 0u};
      rb[set][s] = v;
    }
    ++pnext;
    if (++pst == nst) {
      pst = 0;
      ++ptap;
    }
  };
  auto stage = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) *(u32x4*)(A + (p0 + j * PSTEP) * CB_AS4 + s * 32 + pq * 8) = ra[set][j][s];
    if (wload) {
#pragma unroll
      for (int s = 0; s < 4; ++s) ((u32x4*)B)[s * bvec + tid] = rb[set][s];
    }
  };
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  prefetch(S0{});
  prefetch(S1{});
  auto body = [&](auto setc) __attribute__((always_inline)) {
    __syncthreads();  // every wave is done with the previous stage's tiles
    stage(setc);
    __syncthreads();
    prefetch(setc);   // the set just written takes the stage after next
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      s16x8 fa[2], fb[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *(const s16x8*)(A + (wave * 32 + mt * 16 + li) * CB_AS4 + s * 32 + lg * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const s16x8*)(B + (s * bvec + lg * BN + nt * 16 + li) * 8);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[nt]),
                                                               __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
    }
  };
  for (int ks = k0; ks < nk; ks += 2) {
    body(S0{});
    if (ks + 1 < nk) body(S1{});
  }
  if (a.ksplit > 1) {  // split-K: this split's raw sums; bias, activation and the store in the reduce launch
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + wave * 32 + mt * 16 + li;
      if (m >= M) continue;
      float* pp = a.skpart + ((long)blockIdx.y * M + m) * (a.nblk * BN) + nb * BN + lg * 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *(float4*)(pp + nt * 16) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
    }
    return;
  }
  auto emit = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + wave * 32 + mt * 16 + li;
      if (m >= M) continue;
      const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
      const long pe = (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = nb * BN + nt * 16 + lg * 4;
        if (co >= a.cout) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bv = (a.bias && co + r < a.cout) ? a.bias[co + r] : 0.f;
          o[r] = act_apply(acc[mt][nt][r] + bv, ACT);
        }
        if (co + 4 <= a.cout) {
          if (YB) *(uint2*)((bf16_t*)a.y + pe + co) = make_uint2(cb_pack2(o[0], o[1]), cb_pack2(o[2], o[3]));
          else *(float4*)((float*)a.y + pe + co) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          for (int r = 0; r < 4 && co + r < a.cout; ++r) {
            if (YB) ((bf16_t*)a.y)[pe + co + r] = (bf16_t)(cb_pack2(o[r], 0.f) & 0xffffu);
            else ((float*)a.y)[pe + co + r] = o[r];
          }
        }
      }
    }
  };
  if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
  else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
  else emit(std::integral_constant<int, DIS_ACT_NONE>{});
}

// split-K reduce of convb_fwd128_kernel: y[m][co] = act(sum over the splits (fixed order) + bias); one thread = 4 output channels
template <bool YB>
__global__ __launch_bounds__(256) void convb_splitk_reduce_kernel(GenArgsB a, int coutp) {
  const int M = a.n * a.hv * a.wv;
  const int cq = (a.cout + 3) >> 2;
  const long total = (long)M * cq;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / cq), co = (int)(i - (long)m * cq) * 4;
    float4 sum = *(const float4*)(a.skpart + (long)m * coutp + co);
    for (int sp = 1; sp < a.ksplit; ++sp) {
      const float4 v = *(const float4*)(a.skpart + ((long)sp * M + m) * coutp + co);
      sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    }
    const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
    const long pe = (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff + co;
    const float s4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (co + e >= a.cout) continue;
      const float o = act_apply(s4[e] + (a.bias ? a.bias[co + e] : 0.f), a.act);
      if (YB) ((bf16_t*)a.y)[pe + e] = (bf16_t)(cb_pack2(o, 0.f) & 0xffffu);
      else ((float*)a.y)[pe + e] = o;
    }
  }
}
// split-K factor: only when the launch has fewer workgroups than the device has CUs and a long chain of stages; aims at ~2
// workgroups per CU, at least 6 stages (of 128 channels) per split, at most 8 splits
static int cb_ksplit(long wgs, int nk) {
  if (wgs >= 256 || nk < 24) return 1;
  long ks = (512 + wgs - 1) / wgs;
  if (ks > 8) ks = 8;
  while (ks > 1 && nk / ks < 6) --ks;
  return (int)ks;
}

// 256 x 128 tiles of convb_fwd128_kernel: bf16 input, >= 128 channels on both sides, cout a multiple of 128, and a launch that
// makes ~3/4 of a workgroup per CU WITHOUT split-K (M output positions).  Measured (DispNetS bs=8 x 4 frames): the 32 x 27 maps
// gain (256 -> 512: 0.142 -> 0.116 ms, 256 -> 256: 0.078 -> 0.065 ms), the 16 x 14 maps, which need split-K to fill the device
// with the large tile, lose (1024 -> 512: 0.144 -> 0.153 ms, 512 -> 1024: 0.079 -> 0.097 ms) - unlike the fp32 twin, whose
// operand bytes per MAC are twice these.
static bool cb_big_for(int x_bf16, int cin, int cout, long M, int nk) {
  static const bool off = getenv("DIS_CONVB_BIG") && getenv("DIS_CONVB_BIG")[0] == '0';
  static const bool no128 = getenv("DIS_CONVB_128") && getenv("DIS_CONVB_128")[0] == '0';
  (void)nk;
  if (off || no128 || !x_bf16 || cin < 128 || cout < 128 || cout % 128) return false;
  return ((M + 255) / 256) * (cout / 128) >= 192;
}
// packed[tap][chunk][nb][lg][col][j] = bf16(W(tap, ci = chunk*32 + lg*8 + j, co = nb*BN + col))
struct PackArgsB {
  const float* w;
  bf16_t* packed;
  int ntaps, nchunk, nblk, bn, ci_real, co_real;
  long s_ci, s_co;
  short tsrc[CB_MAXTAPS];
};
__global__ void convb_pack_kernel(PackArgsB a) {
  const long total = (long)a.ntaps * a.nchunk * a.nblk * 4 * a.bn * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (thread order: tap fastest, then the channel of its group of 8 - coalesced reads of the OIHW weights; see convb_pack_batch_kernel)
    const int tap = (int)(i % a.ntaps);
    long r = i / a.ntaps;
    const int j = (int)(r & 7);
    r >>= 3;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    const int chunk = (int)(r / a.nblk);
    const int ci = chunk * CB_CK + lg * 8 + j, co = nb * a.bn + col;
    float v = 0.f;
    if (ci < a.ci_real && co < a.co_real) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
    a.packed[(((((long)tap * a.nchunk + chunk) * a.nblk + nb) * 4 + lg) * a.bn + col) * 8 + j] = (bf16_t)(cb_pack2(v, 0.f) & 0xffffu);
  }
}

// ------------------------------------------------------------------------------------------------
// Halo form of the same convolution family, for feature maps large enough that the per-tap re-read of the input (k*k passes
// through L2 in the streaming kernel: the bound of every high-resolution layer of DispNetS) dominates: the input halo of an
// 8 x 16 tile of (virtual) output positions is staged in LDS ONCE per 32-channel chunk and every tap reads its MFMA operand from
// there at its own offset; only the weights stream (groups of GT k-steps, double-buffered through registers: one barrier per
// group instead of one per tap).
//   k-step = TP taps x 32/TP channels (TP = 1, 2, 4 for cin > 16, <= 16, <= 8: narrow inputs - the network's first layer, the
//   1-channel head gradients, the 16-channel full-resolution layers - fill the MFMA's K = 32 with several taps)
//   LDS pixel = 32/TP (TP = 4: 16, the upper 8 zero) channels + 8 pad: 80 / 48 bytes, conflict-free for 16 consecutive pixels
//   wave w = tile rows 2w, 2w+1 x 16 columns; lane (li, lg): column li, k-slice lg
// ------------------------------------------------------------------------------------------------
#define CBH_TR 8
#define CBH_TC 16
#define CBH_GVEC 1024  // 16-byte vectors of one weight group (16 KB): 4 per thread
// Persistent workgroups: a unit = (tile, cout block); the units of an XCD's share are walked with a grid stride, and the halo
// of the NEXT stage (next 32-channel chunk, or the next unit's first) is fetched into registers (NH 16-byte items per thread)
// while the current stage's MFMAs run; pixels outside the image / channels past cin get an out-of-range buffer offset (zeros).
// MT = output rows per wave: 2 (8 x 16 tiles), or 4 (16 x 16 tiles: twice the MFMAs per streamed weight group and a smaller halo
// overhead - the layers whose weights do not stay resident, 7x7 / 5x5 taps)
// KC > 0 (round 4; streamed weights, stride 1, a full KC x KC window: the 7 x 7 and 5 x 5 layers): the column walk of
// conv_gen.hip's convh2_kernel - taps ordered column by column, a weight group = one column, the MT + KC - 1 row fragments of a
// column read once and kept in registers while ky slides down: 1 + NT ds_read_b128 per k-step instead of MT + NT for MT NT MFMAs
// (MT 4, NT 2: six reads per eight MFMAs had the LDS port 1.5 x as busy as the matrix unit).
// PH = 4 (round 4; resident weights): the four parity classes of a stride-2 transposed form in ONE launch, as convh2_kernel - they
// read the same input tile, so one halo serves four accumulator sets; class p owns the k-steps [ph_k0[p], + ph_kn[p]) of the tap
// list and writes output pixel (2 vy + ph_oy[p], 2 vx + ph_ox[p]).
template <int BN, bool XB, bool YB, int NH, int MT, int KC = 0, int PH = 1>
__global__ __launch_bounds__(256) void convb_halo_kernel(GenArgsB a) {
  static_assert(PH == 1 || KC == 0, "column walk: one phase");
  constexpr int NT = BN / 16, TRH = 4 * MT;
  constexpr int NRB = KC > 0 ? (KC * 4 * BN + 255) / 256 : 4;   // 16-byte vectors of a weight group per thread
  extern __shared__ __attribute__((aligned(16))) unsigned short hsm[];
  int* toff = (int*)hsm;                     // 64 ints
  const int bsz = a.GT * 4 * BN * 8;         // 16-bit words of one weight buffer
  unsigned short* Bb = hsm + 128;
  unsigned short* halo = Bb + a.wsz16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int PS = a.PS, HC = a.HC, ipp = (PS - 8) >> 3;
  // unit -> (cout block, tile): cout block fastest (the blocks of a tile share its halo through L2) when the weights stream,
  // slowest when they are resident (a workgroup then keeps its block's weights for its whole walk)
  const int ntile = a.n * a.tiles_y * a.tiles_x;
  auto unit_nb = [&](int uu) { return a.res ? uu / ntile : uu % a.nblk; };
  auto unit_tile = [&](int uu) { return a.res ? uu % ntile : uu / a.nblk; };
  if (tid < 64) {
    int o = 0;
    if (tid < a.ntaps) o = ((a.tdy[tid] - a.dy0) * HC + (a.tdx[tid] - a.dx0)) * PS;
    toff[tid] = o;
  }
  // units of this workgroup: XCD x (blockIdx.x % 8) owns a contiguous eighth, so that neighbouring tiles (shared halo rows and
  // columns) and the cout blocks of one tile meet in the same L2
  const int units = a.n * a.tiles_y * a.tiles_x * a.nblk;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int u_hi = (int)((long)units * (xcd + 1) / nxcd);
  int u = (int)((long)units * xcd / nxcd) + rank;
  if (u >= u_hi) return;

  const int gpt = 4 / a.TP;  // lane groups per tap
  const int lgq = lg / gpt, cg = lg - lgq * gpt;
  const int a_lane = (wave * MT * a.S * HC + li * a.S) * PS + cg * 8;
  const int rowstep = a.S * HC * PS;
  const u32x4* wq = (const u32x4*)a.w;  // packed [chunk][nb][kstep][lg][BN][8]
  const int ngrp = (a.nks + a.GT - 1) / a.GT;
  u32x4 rb[NRB];
  auto pref_b = [&](int nb, int c, int g) __attribute__((always_inline)) {
    const long base = ((long)(c * a.nblk + nb) * a.nks + g * a.GT) * (4 * BN);
    const int cnt = min(a.GT, a.nks - g * a.GT) * (4 * BN);
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
      const int idx = tid + i * 256;
      rb[i] = wq[base + (idx < cnt ? idx : 0)];
    }
  };
  // halo items of this thread: (row, column) inside the halo, channel group, LDS offset (fixed for the whole launch)
  const int nitems = a.HR * HC * ipp;
  int it_rc[NH], it_cg[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int i = tid + j * 256;
    const int p = i / ipp, cgi = i - p * ipp;
    const int r = p / HC, cc = p - r * HC;
    it_rc[j] = i < nitems ? (r | (cc << 16)) : 0x4000;  // (past the end: a row outside every image)
    it_cg[j] = cgi * 8 | ((p * PS + cgi * 8) << 8);
  }
  u32x4 pre[2][NH], pre2[2][XB ? 1 : NH];  // two register sets: the resident-weight walk keeps two stages in flight
  auto halo_issue = [&](int uu, int c, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    int t = unit_tile(uu);
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y, nn = t / a.tiles_y;
    const int iy0 = ty * TRH * a.S + a.dy0, ix0 = tx * CBH_TC * a.S + a.dx0;
    const long sbase = (long)nn * a.hin * a.win;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int iy = iy0 + (it_rc[j] & 0xffff), ix = ix0 + (it_rc[j] >> 16);
      const int ch = c * CB_CK + (it_cg[j] & 0xff);
      const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win && ch < a.cin;
      const long e = (sbase + (long)iy * a.win + ix) * a.ldx + a.xoff + ch;
      if (XB) {
        pre[set][j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), ok ? (unsigned)(e * 2) : BX_OOB, 0, 0);
      } else {
        pre[set][j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), ok ? (unsigned)(e * 4) : BX_OOB, 0, 0);
        pre2[set][XB ? 0 : j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes),
                                                                 (ok && ch + 4 < a.cin) ? (unsigned)(e * 4 + 16) : BX_OOB, 0, 0);
      }
    }
  };
  auto halo_write = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      if (tid + j * 256 < nitems) {
        u32x4 o = pre[set][j];
        if (!XB) {
          const u32x4 q = pre2[set][XB ? 0 : j];
          o = (u32x4){cb_pack2(__uint_as_float(o[0]), __uint_as_float(o[1])), cb_pack2(__uint_as_float(o[2]), __uint_as_float(o[3])),
                      cb_pack2(__uint_as_float(q[0]), __uint_as_float(q[1])), cb_pack2(__uint_as_float(q[2]), __uint_as_float(q[3]))};
        }
        *(u32x4*)(halo + (it_cg[j] >> 8)) = o;
      }
    }
  };

  f32x4 accp[PH][MT][NT], bias_v[NT];
  f32x4 (&acc)[MT][NT] = accp[0];   // (one phase: the only set)
  int bias_nb = -1;
  // k-steps [0, kn) of a weight block B ([kstep][lg][BN][8]) whose tap offsets start at tq: software-pipelined by hand (a
  // runtime trip count: the compiler does not) - the operands of k-step kk+1 are requested before the MFMAs of kk issue, its
  // tap offset one step earlier still
  auto ksteps = [&](const unsigned short* B, const int* tq, int kn, f32x4 (&acc)[MT][NT]) __attribute__((always_inline)) {
    const unsigned short* bl = B + (lg * BN + li) * 8;
    auto frag = [&](int kk, int to, s16x8 (&fa)[MT], s16x8 (&fb)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = *(const s16x8*)(halo + a_lane + mt * rowstep + to);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = *(const s16x8*)(bl + (kk * 4 * BN + nt * 16) * 8);
    };
    auto mac = [&](const s16x8 (&fa)[MT], const s16x8 (&fb)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[nt]),
                                                               __builtin_bit_cast(bf16x8, fa[mt]), acc[mt][nt], 0, 0, 0);
    };
    s16x8 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    int to0 = tq[0], to1 = kn > 1 ? tq[a.TP] : 0;
    frag(0, to0, fa0, fb0);
    for (int kk = 0; kk < kn; kk += 2) {
      if (kk + 1 < kn) frag(kk + 1, to1, fa1, fb1);
      to0 = kk + 2 < kn ? tq[(kk + 2) * a.TP] : 0;
      mac(fa0, fb0);
      if (kk + 1 < kn) {
        if (kk + 2 < kn) frag(kk + 2, to0, fa0, fb0);
        to1 = kk + 3 < kn ? tq[(kk + 3) * a.TP] : 0;
        mac(fa1, fb1);
      }
    }
  };
  // one column of KC taps (weights B: [ky][lg][BN][8]; `to` = halo offset of the column's first tap): sliding window of row fragments.
  // Rn: the first MT rows of the NEXT column (offset to_next), requested during the last k-step of this one.
  constexpr int KCC = KC > 0 ? KC : 1;
  s16x8 Rn[MT];
  auto col_rows = [&](int to) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < MT; ++j) Rn[j] = *(const s16x8*)(halo + a_lane + j * rowstep + to);
  };
  auto kcolumn = [&](const unsigned short* B, int to, int to_next, bool has_next) __attribute__((always_inline)) {
    const unsigned short* bl = B + (lg * BN + li) * 8;
    s16x8 R[MT + KCC - 1];
    s16x8 fb[2][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j) R[j] = Rn[j];
    auto load_fb = [&](int ky, s16x8 (&f)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) f[nt] = *(const s16x8*)(bl + (ky * 4 * BN + nt * 16) * 8);
    };
    load_fb(0, fb[0]);
#pragma unroll
    for (int ky = 0; ky < KCC; ++ky) {
      if (ky + 1 < KCC) {
        R[ky + MT] = *(const s16x8*)(halo + a_lane + (ky + MT) * rowstep + to);
        load_fb(ky + 1, fb[(ky + 1) & 1]);
      } else if (has_next) {
        col_rows(to_next);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ky & 1][nt]),
                                                               __builtin_bit_cast(bf16x8, R[ky + mt]), acc[mt][nt], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  auto load_bias = [&](int nb) __attribute__((always_inline)) {
    // bias of a cout block (4 consecutive couts per lane and block): requested at the unit's start, added in its epilogue
    if (nb == bias_nb) return;
    bias_nb = nb;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nb * BN + nt * 16 + lg * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < a.cout) bv[r] = a.bias[co + r];
      }
      bias_v[nt] = bv;
    }
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accp[ph][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // epilogue: lane (li, lg) holds couts nb*BN + nt*16 + lg*4 + {0..3} of position (ty*8 + wave*2 + mt, tx*16 + li)
  auto epilogue = [&](int uu, int nb) __attribute__((always_inline)) {
    int t = unit_tile(uu);
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y, nn = t / a.tiles_y;
    auto emit = [&](auto actc) __attribute__((always_inline)) {
      constexpr int ACT = decltype(actc)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int vy = ty * TRH + wave * MT + mt, vx = tx * CBH_TC + li;
        if (vy >= a.hv || vx >= a.wv) continue;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
          const int oy = vy * a.osy + (PH > 1 ? (int)a.ph_oy[ph] : a.ooy), ox = vx * a.osx + (PH > 1 ? (int)a.ph_ox[ph] : a.oox);
          if (PH > 1 && (oy >= a.hf || ox >= a.wf)) continue;   // (the classes of an odd output size differ by one row / column)
          const long pe = (((long)nn * a.hf + oy) * a.wf + ox) * a.ldy + a.yoff;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = nb * BN + nt * 16 + lg * 4;
            if (co >= a.cout) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = act_apply(accp[ph][mt][nt][r] + bias_v[nt][r], ACT);
            if (co + 4 <= a.cout) {
              if (YB) *(uint2*)((bf16_t*)a.y + pe + co) = make_uint2(cb_pack2(o[0], o[1]), cb_pack2(o[2], o[3]));
              else *(float4*)((float*)a.y + pe + co) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
              for (int r = 0; r < 4 && co + r < a.cout; ++r) {
                if (YB) ((bf16_t*)a.y)[pe + co + r] = (bf16_t)(cb_pack2(o[r], 0.f) & 0xffffu);
                else ((float*)a.y)[pe + co + r] = o[r];
              }
            }
          }
        }
      }
    };
    if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
    else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
    else emit(std::integral_constant<int, DIS_ACT_NONE>{});
  };

  if (a.res) {
    // Resident weights.  The walk is a flat sequence of stages (unit, chunk); the halos of the next TWO stages are in flight
    // (register sets alternate), because a stage of a narrow high-resolution layer is far shorter than a memory round trip:
    // one stage of look-ahead leaves the loads exposed, two keep ~2 tiles per workgroup in flight.
    int u1 = u, c1 = 0, u2 = u, c2 = 0;  // cursors of the stages held by the register sets after the current one
    auto advance = [&](int& uu, int& cc) __attribute__((always_inline)) {
      if (cc + 1 < a.nchunk) ++cc;
      else uu += per, cc = 0;
    };
    int c0 = 0;
    halo_issue(u, 0, S0{});
    advance(u1, c1);
    if (u1 < u_hi) halo_issue(u1, c1, S1{});
    u2 = u1, c2 = c1;
    advance(u2, c2);
    int wres_nb = -1;
    auto stage = [&](auto setc) __attribute__((always_inline)) {
      const int nb = unit_nb(u);
      if (c0 == 0) {
        load_bias(nb);
        zero_acc();
      }
      if (nb != wres_nb) {  // (block-uniform) this cout block's weights, all chunks: once per workgroup and block
        __syncthreads();
        const int nv = a.nks * 4 * BN;
        for (int c = 0; c < a.nchunk; ++c)
          for (int i = tid; i < nv; i += 256) ((u32x4*)Bb)[c * nv + i] = wq[(long)(c * a.nblk + nb) * nv + i];
        wres_nb = nb;
      }
      __syncthreads();  // every wave is done with the previous stage's halo
      halo_write(setc);
      __syncthreads();
      if (u2 < u_hi) halo_issue(u2, c2, setc);  // the set just consumed takes the stage after next
      if constexpr (PH > 1) {
#pragma unroll
        for (int ph = 0; ph < PH; ++ph)
          ksteps(Bb + ((long)c0 * a.nks + a.ph_k0[ph]) * (4 * BN * 8), toff + a.ph_k0[ph] * a.TP + lgq, a.ph_kn[ph], accp[ph]);
      } else {
        ksteps(Bb + (long)c0 * a.nks * 4 * BN * 8, toff + lgq, a.nks, acc);
      }
      if (c0 + 1 == a.nchunk) epilogue(u, nb);
      u = u1, c0 = c1;
      u1 = u2, c1 = c2;
      advance(u2, c2);
    };
    while (true) {
      stage(S0{});
      if (u >= u_hi) break;
      stage(S1{});
      if (u >= u_hi) break;
    }
    return;
  }

  // streaming weights: groups of GT k-steps double-buffered through registers, one stage of halo look-ahead
  halo_issue(u, 0, S0{});
  pref_b(unit_nb(u), 0, 0);
  int flat = 0;
  while (true) {
    const int nb = unit_nb(u);
    const int un = u + per;  // this workgroup's next unit
    load_bias(nb);
    zero_acc();
    for (int c = 0; c < a.nchunk; ++c) {
      __syncthreads();  // every wave is done with the previous stage's halo
      halo_write(S0{});
      for (int g = 0; g < ngrp; ++g) {
        unsigned short* B = Bb + (flat & 1) * bsz;
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
          const int idx = tid + i * 256;
          if (idx < a.GT * 4 * BN) ((u32x4*)B)[idx] = rb[i];
        }
        __syncthreads();
        if (g == 0) {  // next stage's halo
          if (c + 1 < a.nchunk) halo_issue(u, c + 1, S0{});
          else if (un < u_hi) halo_issue(un, 0, S0{});
        }
        if (g + 1 < ngrp) pref_b(nb, c, g + 1);
        else if (c + 1 < a.nchunk) pref_b(nb, c + 1, 0);
        else if (un < u_hi) pref_b(unit_nb(un), 0, 0);
        if constexpr (KC > 0) {
          if (g == 0) col_rows(toff[0]);   // (the first column of a stage: its rows were not requested by a predecessor)
          kcolumn(B, toff[g * KC], g + 1 < ngrp ? toff[(g + 1) * KC] : 0, g + 1 < ngrp);
        } else {
          ksteps(B, toff + g * a.GT * a.TP + lgq, min(a.GT, a.nks - g * a.GT), acc);
        }
        ++flat;
      }
    }
    epilogue(u, nb);
    if (un >= u_hi) break;
    u = un;
  }
}

// packed[chunk][nb][kstep][lg][col][j] = bf16(W(tap, ci, co = nb*BN + col)), (tap, ci) = (kstep*TP + lg / (4/TP),
// chunk*32 + (lg % (4/TP))*8 + j); taps past the last one and channels past ci_real hold zeros
struct PackArgsH {
  const float* w;
  bf16_t* packed;
  int ntaps, nchunk, nblk, bn, ci_real, co_real, tp, nks;
  long s_ci, s_co;
  short tsrc[CB_MAXTAPS];
};
__global__ void convb_pack_halo_kernel(PackArgsH a) {
  const long total = (long)a.nchunk * a.nblk * a.nks * 4 * a.bn * 8;
  const int gpt = 4 / a.tp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ks = (int)(i % a.nks);   // (thread order: k-step fastest, as convb_pack_batch_kernel)
    long r = i / a.nks;
    const int j = (int)(r & 7);
    r >>= 3;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    const int chunk = (int)(r / a.nblk);
    const int tap = ks * a.tp + lg / gpt;
    const int ci = chunk * CB_CK + (lg % gpt) * 8 + j, co = nb * a.bn + col;
    float v = 0.f;
    if (tap < a.ntaps && ci < a.ci_real && co < a.co_real) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
    a.packed[(((((long)chunk * a.nblk + nb) * a.nks + ks) * 4 + lg) * a.bn + col) * 8 + j] = (bf16_t)(cb_pack2(v, 0.f) & 0xffffu);
  }
}

// ------------------------------------------------------------------------------------------------
// One packing launch per STEP instead of one per call (round 4).  The packed weights of a call depend on the weights alone, which
// change once per step (Adam): dis_convb_pack_record() makes every packing launch of the following calls also leave its descriptor
// in a HOST array; the caller keeps the calls' wpack buffers alive, uploads the array once, and from then on runs
// dis_convb_pack_batch() at the start of a step and dis_convb_run(mode | DIS_CONVB_PREPACKED) for the calls (~100 launches of
// 5 - 7 us each become one).  The record is process-global host state, like dis_last_kernel: not thread-safe.
// ------------------------------------------------------------------------------------------------
struct PackDesc {
  int kind;  // 0: streaming layout (convb_pack_kernel), 1: halo layout (convb_pack_halo_kernel)
  int blk0;  // first workgroup of this descriptor in the batch launch (set when the record is stopped); nblk workgroups
  int nblk, pad_;
  PackArgsH a;  // (a superset of PackArgsB: tp / nks are the halo layout's)
};
static PackDesc* g_pack_rec = nullptr;
static int g_pack_cap = 0, g_pack_n = 0;
static void cb_pack_note(int kind, const PackArgsH& a) {
  if (!g_pack_rec) return;
  if (g_pack_n < g_pack_cap) {
    g_pack_rec[g_pack_n].kind = kind;
    g_pack_rec[g_pack_n].blk0 = g_pack_rec[g_pack_n].nblk = g_pack_rec[g_pack_n].pad_ = 0;
    g_pack_rec[g_pack_n].a = a;
  }
  ++g_pack_n;  // (past the capacity: the caller sees count > capacity and does not use the record)
}
#define CB_PACK_EPB 4096  // packed values per workgroup (a descriptor gets ceil(total / CB_PACK_EPB) workgroups, at most 512)
static long cb_pack_total(const PackDesc& d) {
  return d.kind == 0 ? (long)d.a.ntaps * d.a.nchunk * d.a.nblk * 4 * d.a.bn * 8 : (long)d.a.nchunk * d.a.nblk * d.a.nks * 4 * d.a.bn * 8;
}
__global__ __launch_bounds__(256) void convb_pack_batch_kernel(const PackDesc* __restrict__ descs, int count) {
  int lo = 0, hi = count - 1;  // the descriptor this workgroup belongs to: last one with blk0 <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackDesc& d = descs[lo];
  const PackArgsH& a = d.a;
  const long first = (long)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x, stride = (long)d.nblk * 256;
  // Thread order: the TAP (k-step) index runs fastest, then the channel within its group of 8 - consecutive threads then read
  // consecutive floats of the weight tensor (the taps of one (co, ci) are adjacent in OIHW, the next ci follows them).  With the
  // packed order as thread order (channel fastest, taps last) every 4-byte read touched its own sector and the taps of a sector
  // were read by workgroups far apart in time: 1.88 GB fetched for 126 MB of weights (PMC, scripts/diag/sf_pmc.sh), 0.30 ms.
  if (d.kind == 0) {
    const long total = (long)a.ntaps * a.nchunk * a.nblk * 4 * a.bn * 8;
    for (long i = first; i < total; i += stride) {
      const int tap = (int)(i % a.ntaps);
      long r = i / a.ntaps;
      const int j = (int)(r & 7);
      r >>= 3;
      const int col = (int)(r % a.bn);
      r /= a.bn;
      const int lg = (int)(r & 3);
      r >>= 2;
      const int nb = (int)(r % a.nblk);
      const int chunk = (int)(r / a.nblk);
      const int ci = chunk * CB_CK + lg * 8 + j, co = nb * a.bn + col;
      float v = 0.f;
      if (ci < a.ci_real && co < a.co_real) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
      a.packed[(((((long)tap * a.nchunk + chunk) * a.nblk + nb) * 4 + lg) * a.bn + col) * 8 + j] = (bf16_t)(cb_pack2(v, 0.f) & 0xffffu);
    }
  } else {
    const long total = (long)a.nchunk * a.nblk * a.nks * 4 * a.bn * 8;
    const int gpt = 4 / a.tp;
    for (long i = first; i < total; i += stride) {
      const int ks = (int)(i % a.nks);
      long r = i / a.nks;
      const int j = (int)(r & 7);
      r >>= 3;
      const int col = (int)(r % a.bn);
      r /= a.bn;
      const int lg = (int)(r & 3);
      r >>= 2;
      const int nb = (int)(r % a.nblk);
      const int chunk = (int)(r / a.nblk);
      const int tap = ks * a.tp + lg / gpt;
      const int ci = chunk * CB_CK + (lg % gpt) * 8 + j, co = nb * a.bn + col;
      float v = 0.f;
      if (tap < a.ntaps && ci < a.ci_real && co < a.co_real) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
      a.packed[(((((long)chunk * a.nblk + nb) * a.nks + ks) * 4 + lg) * a.bn + col) * 8 + j] = (bf16_t)(cb_pack2(v, 0.f) & 0xffffu);
    }
  }
}
extern "C" long dis_convb_pack_desc_bytes(void) { return (long)sizeof(PackDesc); }
extern "C" int dis_convb_pack_record(void* host_descs, int capacity, int* count_out) {
  if (host_descs) {  // start: descriptors of the following packing launches go to host_descs[0 .. capacity)
    if (capacity <= 0) return DIS_ERR_BAD_SHAPE;
    g_pack_rec = (PackDesc*)host_descs;
    g_pack_cap = capacity;
    g_pack_n = 0;
    return DIS_OK;
  }
  if (!count_out) return DIS_ERR_NULL;  // stop: how many launches were seen (> capacity: the record is incomplete)
  count_out[0] = g_pack_n;
  // ... and the batch launch's plan: workgroups in proportion to the packed size; count_out[1] = workgroups in all
  long blk = 0;
  if (g_pack_rec && g_pack_n <= g_pack_cap)
    for (int i = 0; i < g_pack_n; ++i) {
      long nb = (cb_pack_total(g_pack_rec[i]) + CB_PACK_EPB - 1) / CB_PACK_EPB;
      nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
      g_pack_rec[i].blk0 = (int)blk;
      g_pack_rec[i].nblk = (int)nb;
      blk += nb;
    }
  count_out[1] = (int)blk;
  g_pack_rec = nullptr;
  g_pack_cap = g_pack_n = 0;
  return DIS_OK;
}
extern "C" int dis_convb_pack_batch(const void* dev_descs, int count, int workgroups, void* stream) {
  if (!dev_descs) return DIS_ERR_NULL;
  if (count <= 0 || workgroups < count) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(convb_pack_batch_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream,
                     (const PackDesc*)dev_descs, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

static int cb_bn(int cout) {
  static const int cap = getenv("DIS_CONVB_BN") ? atoi(getenv("DIS_CONVB_BN")) : 64;   // (experiments: narrower cout blocks)
  const int bn = cout > 32 ? 64 : (cout > 16 ? 32 : 16);
  return bn > cap ? cap : bn;
}

template <bool XB, bool YB>
static void cb_launch(const GenArgsB& a, int bn, long grid, hipStream_t s) {
  DIS_TAG("convb_fwd_kernel (bf16 streaming)");
  if (bn == 64) hipLaunchKernelGGL((convb_fwd_kernel<64, XB, YB>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (bn == 32) hipLaunchKernelGGL((convb_fwd_kernel<32, XB, YB>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((convb_fwd_kernel<16, XB, YB>), dim3((unsigned)grid), dim3(256), 0, s, a);
}

// halo form: worth it when the image is large (tile quantisation of small maps wastes the tile; their layers are weight-bound
// anyway) and the halo + weight buffers fit LDS.  Fills the halo fields of a.
static int cb_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}
static long cb_halo_lds(const GenArgsB& a, int bn);
static bool cb_halo_plan(GenArgsB& a, int bn) {
  static const bool off = getenv("DIS_CONVB_HALO") && getenv("DIS_CONVB_HALO")[0] == '0';
  if (off || (long)a.hv * a.wv < 1024 || a.ntaps > CB_MAXTAPS) return false;
  int dy0 = a.tdy[0], dy1 = a.tdy[0], dx0 = a.tdx[0], dx1 = a.tdx[0];
  for (int t = 1; t < a.ntaps; ++t) {
    dy0 = a.tdy[t] < dy0 ? a.tdy[t] : dy0; dy1 = a.tdy[t] > dy1 ? a.tdy[t] : dy1;
    dx0 = a.tdx[t] < dx0 ? a.tdx[t] : dx0; dx1 = a.tdx[t] > dx1 ? a.tdx[t] : dx1;
  }
  a.dy0 = dy0; a.dx0 = dx0;
  a.trh = CBH_TR;
  a.HR = (a.trh - 1) * a.S + (dy1 - dy0) + 1;
  a.HC = (CBH_TC - 1) * a.S + (dx1 - dx0) + 1;
  a.TP = a.cin <= 8 ? 4 : (a.cin <= 16 ? 2 : 1);
  a.PS = a.TP == 1 ? 40 : 24;
  a.nks = (a.ntaps + a.TP - 1) / a.TP;
  if (a.nks * a.TP > 64) return false;
  a.GT = CBH_GVEC / (4 * bn);
  if (a.GT > a.nks) a.GT = a.nks;
  a.tiles_x = (a.wv + CBH_TC - 1) / CBH_TC;
  a.tiles_y = (a.hv + a.trh - 1) / a.trh;
  if ((long)a.HR * a.HC * ((a.PS - 8) / 8) > 12 * 256) return false;  // halo items per thread (register prefetch)
  // a cout block's weights (all chunks) stay in LDS when they fit next to the halo: no weight traffic, no group barriers
  const long wall = (long)a.nchunk * a.nks * 4 * bn * 16, hal = (long)a.HR * a.HC * a.PS * 2;
  a.res = (256 + wall + hal <= 75 * 1024) ? 1 : 0;  // (two resident workgroups per CU at least: one alone cannot hide its LDS latency)
  a.wsz16 = (int)((a.res ? wall : 2L * a.GT * 4 * bn * 16) / 2);
  // (16-row tiles for the resident-weight layers as well: measured neutral, 5.51 vs 5.56 ms over the 33 halo launches of a step)
  if (a.nph > 1 && (!a.res || a.TP != 1)) return false;   // (four phases in one launch: resident weights only)
  if (!a.res && a.hv >= 32) {
    // streaming weights: 16 x 16 tiles when the larger halo still leaves two workgroups per CU and 12 prefetch items per thread
    const int hr16 = 15 * a.S + (dy1 - dy0) + 1;
    const long hal16 = (long)hr16 * a.HC * a.PS * 2;
    static const bool no16 = getenv("DIS_CONVB_T16") && getenv("DIS_CONVB_T16")[0] == '0';
    if (!no16 && 256 + 2L * a.wsz16 + hal16 <= 75 * 1024 && (long)hr16 * a.HC * ((a.PS - 8) / 8) <= 12 * 256) {
      a.trh = 16;
      a.HR = hr16;
      a.tiles_y = (a.hv + 15) / 16;
    }
  }
  // column walk (convb_halo_kernel, KC): a full K x K stride-1 window on 16-row tiles, the instances that exist
  a.kc = 0;
  {
    const int K = dy1 - dy0 + 1;
    static const bool nokc = getenv("DIS_CONVB_KC") && getenv("DIS_CONVB_KC")[0] == '0';
    if (!nokc && a.nph == 1 && a.S == 1 && a.TP == 1 && !a.res && a.trh == 16 && dx1 - dx0 + 1 == K && a.ntaps == K * K &&
        ((K == 7 && bn == 32) || (K == 5 && bn == 64))) {
      const long wsz = 2L * K * 4 * bn * 16 / 2;   // two buffers of one column (16-bit words)
      if (256 + 2 * wsz + (long)a.HR * a.HC * a.PS * 2 <= 150 * 1024) {
        a.kc = K;
        a.GT = K;
        a.wsz16 = (int)wsz;
      }
    }
  }
  return cb_halo_lds(a, bn) <= 150 * 1024;
}
static long cb_halo_lds(const GenArgsB& a, int bn) { return 256 + 2L * a.wsz16 + (long)a.HR * a.HC * a.PS * 2; }
template <int BN, bool XB, bool YB, int NH, int MT, int KC = 0, int PH = 1>
static int cbh_launch3(const GenArgsB& a, long grid, long lds, hipStream_t s) {
  static bool attr = false;
  auto kern = convb_halo_kernel<BN, XB, YB, NH, MT, KC, PH>;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  DIS_TAG(KC > 0 ? "convb_halo_kernel (bf16 LDS halo, column walk)" : PH > 1 ? "convb_halo_kernel (bf16 LDS halo, 4 phases)" : "convb_halo_kernel (bf16 LDS halo)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)lds, s, a);
  return DIS_OK;
}
template <int BN, bool XB, bool YB, int NH>
static int cbh_launch2(const GenArgsB& a, long grid, long lds, hipStream_t s) {
  if (a.trh == 16) return cbh_launch3<BN, XB, YB, NH, 4>(a, grid, lds, s);
  return cbh_launch3<BN, XB, YB, NH, 2>(a, grid, lds, s);
}
template <int BN, bool XB, bool YB>
static int cbh_launch1(const GenArgsB& a, int nh, long grid, long lds, hipStream_t s) {
  if (nh <= 4) return cbh_launch2<BN, XB, YB, 4>(a, grid, lds, s);
  if (nh <= 8) return cbh_launch2<BN, XB, YB, 8>(a, grid, lds, s);
  return cbh_launch2<BN, XB, YB, 12>(a, grid, lds, s);
}
template <bool XB, bool YB>
static int cbh_launch(const GenArgsB& a, int bn, int nh, long grid, long lds, hipStream_t s) {
  if (a.nph == 4) {   // (bf16 -> bf16 layers, 8-row tiles, resident weights)
    if constexpr (XB && YB) {
      if (bn == 64 && nh <= 4) return cbh_launch3<64, true, true, 4, 2, 0, 4>(a, grid, lds, s);
      if (bn == 32 && nh <= 4) return cbh_launch3<32, true, true, 4, 2, 0, 4>(a, grid, lds, s);
      if (bn == 16 && nh <= 4) return cbh_launch3<16, true, true, 4, 2, 0, 4>(a, grid, lds, s);
    }
    return DIS_ERR_UNSUPPORTED;
  }
  if (a.kc > 0) {   // (cb_halo_plan grants the column walk to bf16 -> bf16 layers of these two shapes only)
    if constexpr (XB && YB) {
      if (a.kc == 7 && bn == 32 && nh <= 8) return cbh_launch3<32, true, true, 8, 4, 7>(a, grid, lds, s);
      if (a.kc == 5 && bn == 64 && nh <= 8) return cbh_launch3<64, true, true, 8, 4, 5>(a, grid, lds, s);
    }
    return DIS_ERR_UNSUPPORTED;
  }
  if (bn == 64) return cbh_launch1<64, XB, YB>(a, nh, grid, lds, s);
  if (bn == 32) return cbh_launch1<32, XB, YB>(a, nh, grid, lds, s);
  return cbh_launch1<16, XB, YB>(a, nh, grid, lds, s);
}
static int cb_run_halo(GenArgsB a, int x_bf16, int y_bf16, int bn, const float* w_raw, bf16_t* wpack, int ci_real,
                       int co_real, long s_ci, long s_co, const short* tsrc, hipStream_t s) {
  if (a.kc > 0 && !(x_bf16 && y_bf16)) {   // (no instance: the generic walk with its own group size)
    a.kc = 0;
    a.GT = CBH_GVEC / (4 * bn);
    if (a.GT > a.nks) a.GT = a.nks;
    a.wsz16 = (int)((2L * a.GT * 4 * bn * 16) / 2);
  }
  PackArgsH p;
  p.w = w_raw; p.packed = wpack; p.ntaps = a.ntaps; p.nchunk = a.nchunk; p.nblk = a.nblk; p.bn = bn;
  p.ci_real = ci_real; p.co_real = co_real; p.tp = a.TP; p.nks = a.nks; p.s_ci = s_ci; p.s_co = s_co;
  for (int t = 0; t < a.ntaps; ++t) p.tsrc[t] = tsrc[t];
  if (a.kc > 0) {
    // column walk: taps ordered by (dx, dy) ascending - slot (dx - dx0) * kc + (dy - dy0); the window is full (cb_halo_plan)
    short ty[CB_MAXTAPS], tx[CB_MAXTAPS];
    for (int t = 0; t < a.ntaps; ++t) {
      const int slot = (a.tdx[t] - a.dx0) * a.kc + (a.tdy[t] - a.dy0);
      ty[slot] = a.tdy[t]; tx[slot] = a.tdx[t]; p.tsrc[slot] = tsrc[t];
    }
    for (int t = 0; t < a.ntaps; ++t) a.tdy[t] = ty[t], a.tdx[t] = tx[t];
  }
  const long ptotal = (long)a.nchunk * a.nblk * a.nks * 4 * bn * 8;
  if (w_raw) {  // (null: the call runs on weights dis_convb_pack_batch has packed)
    hipLaunchKernelGGL(convb_pack_halo_kernel, dim3(dis_ew_grid(ptotal, 256)), dim3(256), 0, s, p);
    cb_pack_note(1, p);
  }
  a.w = wpack;
  const long units = (long)a.n * a.tiles_y * a.tiles_x * a.nblk;
  if (units > 2147483647L) return DIS_ERR_BAD_SHAPE;
  const long lds = cb_halo_lds(a, bn);
  // persistent grid: as many workgroups as stay resident (LDS-bound, at most 6 per CU), a multiple of the 8 XCDs
  long wpc = (150L * 1024) / lds;
  wpc = wpc > 6 ? 6 : (wpc < 1 ? 1 : wpc);
  long grid = wpc * cb_num_cus();
  if (grid > units) grid = units;
  if (grid >= 8) grid -= grid % 8;
  const int nh = (int)(((long)a.HR * a.HC * ((a.PS - 8) / 8) + 255) / 256);
  int rc;
  if (x_bf16 && y_bf16) rc = cbh_launch<true, true>(a, bn, nh, grid, lds, s);
  else if (x_bf16) rc = cbh_launch<true, false>(a, bn, nh, grid, lds, s);
  else if (y_bf16) rc = cbh_launch<false, true>(a, bn, nh, grid, lds, s);
  else rc = cbh_launch<false, false>(a, bn, nh, grid, lds, s);
  if (rc != DIS_OK) return rc;
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

static int cb_run(GenArgsB a, int x_bf16, int y_bf16, const float* w_raw, bf16_t* wpack, int ci_real, int co_real, long s_ci,
                  long s_co, const short* tsrc, hipStream_t s, bool halo_only = false) {
  if (a.ntaps <= 0) return DIS_OK;
  const long M = (long)a.n * a.hv * a.wv;
  if (M <= 0) return DIS_OK;
  int bn = cb_bn(a.cout);
  a.nblk = (a.cout + bn - 1) / bn;
  const long xb = (long)a.n * a.hin * a.win * a.ldx * (x_bf16 ? 2 : 4);
  if (xb >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;  // 31-bit byte offsets of the buffer descriptor
  a.x_bytes = (unsigned)xb;
  a.nchunk = (a.cin + CB_CK - 1) / CB_CK;
  if (halo_only) {   // (four phases in one launch: bf16 tensors on both sides, a plan with resident weights and <= 4 halo items)
    if (!(x_bf16 && y_bf16) || !cb_halo_plan(a, bn)) return DIS_ERR_UNSUPPORTED;
    if ((long)a.HR * a.HC * ((a.PS - 8) / 8) > 4 * 256) return DIS_ERR_UNSUPPORTED;
    return cb_run_halo(a, x_bf16, y_bf16, bn, w_raw, wpack, ci_real, co_real, s_ci, s_co, tsrc, s);
  }
  if (cb_halo_plan(a, bn)) return cb_run_halo(a, x_bf16, y_bf16, bn, w_raw, wpack, ci_real, co_real, s_ci, s_co, tsrc, s);
  const bool big = cb_big_for(x_bf16, a.cin, a.cout, M, a.ntaps * ((a.nchunk + 3) >> 2));   // deep layers: 256 x 128 tiles
  const int bn0 = bn;
  (void)bn0;
  if (big) {
    bn = 128;
    a.nblk = (a.cout + bn - 1) / bn;
  }
  PackArgsB p;
  p.w = w_raw; p.packed = wpack; p.ntaps = a.ntaps; p.nchunk = a.nchunk; p.nblk = a.nblk; p.bn = bn;
  p.ci_real = ci_real; p.co_real = co_real; p.s_ci = s_ci; p.s_co = s_co;
  for (int t = 0; t < a.ntaps; ++t) p.tsrc[t] = tsrc[t];
  const long ptotal = (long)a.ntaps * a.nchunk * a.nblk * 4 * bn * 8;
  if (w_raw) {
    hipLaunchKernelGGL(convb_pack_kernel, dim3(dis_ew_grid(ptotal, 256)), dim3(256), 0, s, p);
    PackArgsH ph;
    ph.w = p.w; ph.packed = p.packed; ph.ntaps = p.ntaps; ph.nchunk = p.nchunk; ph.nblk = p.nblk; ph.bn = p.bn;
    ph.ci_real = p.ci_real; ph.co_real = p.co_real; ph.tp = 1; ph.nks = p.ntaps; ph.s_ci = p.s_ci; ph.s_co = p.s_co;
    for (int t = 0; t < a.ntaps; ++t) ph.tsrc[t] = p.tsrc[t];
    cb_pack_note(0, ph);
  }
  a.w = wpack;
  const long grid = ((M + (big ? 256 : CB_BM) - 1) / (big ? 256 : CB_BM)) * a.nblk;
  if (grid > 2147483647L) return DIS_ERR_BAD_SHAPE;
  static const bool no128 = getenv("DIS_CONVB_128") && getenv("DIS_CONVB_128")[0] == '0';
  if (x_bf16 && a.cin >= 128 && bn >= 32 && !no128) {  // deep layers: 128-channel stages
    const int nk128 = a.ntaps * ((a.nchunk + 3) >> 2), coutp = a.nblk * bn;
    int ksplit = a.skpart ? cb_ksplit(grid, nk128) : 1;
    while (ksplit > 1 && (long)ksplit * M * coutp > a.skcap) --ksplit;
    a.ksplit = ksplit;
    DIS_TAG(ksplit > 1 ? "convb_fwd128_kernel (bf16 streaming, 128-channel stages, split-K)"
                       : "convb_fwd128_kernel (bf16 streaming, 128-channel stages)");
    const dim3 g((unsigned)grid, (unsigned)ksplit);
    constexpr int lds64 = Cb128Cfg<64>::LDS_BYTES, lds32 = Cb128Cfg<32>::LDS_BYTES, ldsbig = Cb128Cfg<128, 8>::LDS_BYTES;
    if (big) {
      static bool attr[2] = {false, false};
      if (!attr[y_bf16 ? 1 : 0]) {
        hipError_t e = y_bf16 ? hipFuncSetAttribute((const void*)convb_fwd128_kernel<128, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbig)
                              : hipFuncSetAttribute((const void*)convb_fwd128_kernel<128, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbig);
        if (e != hipSuccess) return (int)e;
        attr[y_bf16 ? 1 : 0] = true;
      }
      DIS_TAG(ksplit > 1 ? "convb_fwd128_kernel (bf16 streaming, 128-channel stages, 256 x 128 tiles, split-K)"
                         : "convb_fwd128_kernel (bf16 streaming, 128-channel stages, 256 x 128 tiles)");
      if (y_bf16) hipLaunchKernelGGL((convb_fwd128_kernel<128, true, 8>), g, dim3(512), ldsbig, s, a);
      else hipLaunchKernelGGL((convb_fwd128_kernel<128, false, 8>), g, dim3(512), ldsbig, s, a);
    } else if (bn == 64 && y_bf16) hipLaunchKernelGGL((convb_fwd128_kernel<64, true>), g, dim3(256), lds64, s, a);
    else if (bn == 64) hipLaunchKernelGGL((convb_fwd128_kernel<64, false>), g, dim3(256), lds64, s, a);
    else if (y_bf16) hipLaunchKernelGGL((convb_fwd128_kernel<32, true>), g, dim3(256), lds32, s, a);
    else hipLaunchKernelGGL((convb_fwd128_kernel<32, false>), g, dim3(256), lds32, s, a);
    if (ksplit > 1) {
      const int rg = dis_ew_grid(M * ((a.cout + 3) / 4), 256);
      if (y_bf16) hipLaunchKernelGGL((convb_splitk_reduce_kernel<true>), dim3(rg), dim3(256), 0, s, a, coutp);
      else hipLaunchKernelGGL((convb_splitk_reduce_kernel<false>), dim3(rg), dim3(256), 0, s, a, coutp);
    }
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  if (x_bf16 && y_bf16) cb_launch<true, true>(a, bn, grid, s);
  else if (x_bf16) cb_launch<true, false>(a, bn, grid, s);
  else if (y_bf16) cb_launch<false, true>(a, bn, grid, s);
  else cb_launch<false, false>(a, bn, grid, s);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// floats of split-K partial sums dis_convb_run may need BEHIND its four packing slices (0: none)
extern "C" long dis_convb_splitk_workspace(int mode, int x_bf16, int n, int hin, int win, int hout, int wout, int cin, int cout,
                                           int k, int stride, int pad) {
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || cin <= 0 || cout <= 0 || k <= 0 || k * k > CB_MAXTAPS || pad < 0)
    return -1;
  const int bn0 = cb_bn(cout);
  if (!x_bf16 || cin < 128 || bn0 < 32) return 0;
  const long nst = ((cin + CB_CK - 1) / CB_CK + 3) / 4;
  auto need = [&](long hv, long wv, int ntaps) -> long {
    const long M = (long)n * hv * wv;
    if (M <= 0 || ntaps <= 0) return 0;
    const bool big = cb_big_for(x_bf16, cin, cout, M, (int)(ntaps * nst));   // (the tile cb_run will pick for this launch)
    const int bn = big ? 128 : bn0, bm = big ? 256 : CB_BM;
    const long nblk = (cout + bn - 1) / bn;
    const int ks = cb_ksplit(((M + bm - 1) / bm) * nblk, (int)(ntaps * nst));
    return ks > 1 ? (long)ks * M * nblk * bn : 0;
  };
  const bool phased = (mode == DIS_CONVG_CONV_DGRAD || mode == DIS_CONVG_TCONV) && stride == 2;
  if (!phased) return need(hout, wout, k * k);
  long best = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      int nt = 0;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          if (!((py + pad - ky) & 1) && !((px + pad - kx) & 1)) ++nt;
      const long v = need((hout - py + 1) / 2, (wout - px + 1) / 2, nt);
      best = v > best ? v : best;
    }
  return best;
}

// workspace in 16-bit words for ONE phase
extern "C" long dis_convb_pack_workspace(int cin, int cout, int k) {
  if (cin <= 0 || cout <= 0 || k <= 0 || k * k > CB_MAXTAPS) return -1;
  const int bn = cb_bn(cout);
  const long nblk = (cout + bn - 1) / bn, nchunk = (cin + CB_CK - 1) / CB_CK;
  return (long)k * k * nchunk * nblk * 4 * bn * 8;
}

static int cb_floordiv2(int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); }

// Same modes and argument meaning as dis_convg_run (conv_gen.hip); ldx / xoff / ldy / yoff count ELEMENTS of the tensor's
// own type; x_bf16 / y_bf16 select bf16 (1) or fp32 (0) storage of x / y.  bf16 tensors need ld, offset and cin that
// are multiples of 8 (16-byte vectors), fp32 ones multiples of 4.  wpack: 4 x dis_convb_pack_workspace 16-bit words.
extern "C" int dis_convb_run(int mode, const void* x, int x_bf16, int ldx, int xoff, const float* w, const float* bias,
                             void* y, int y_bf16, int ldy, int yoff, void* wpack, int n, int hin, int win, int cin,
                             int cin_w, int hout, int wout, int cout, int cout_w, int k, int stride, int pad, int act,
                             void* stream) {
  const bool prepacked = (mode & DIS_CONVB_PREPACKED) != 0;  // wpack already holds this call's packed weights (dis_convb_pack_batch)
  mode &= ~DIS_CONVB_PREPACKED;
  if (!x || (!w && !prepacked) || !y || !wpack) return DIS_ERR_NULL;
  if (prepacked) w = nullptr;
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || cin <= 0 || cout <= 0 || cin_w <= 0 || cout_w <= 0 ||
      k <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  if (k * k > CB_MAXTAPS || (stride != 1 && stride != 2) || mode < 0 || mode > 3) return DIS_ERR_UNSUPPORTED;
  const int xa = x_bf16 ? 7 : 3, ya = y_bf16 ? 3 : 0;
  if ((cin & xa) || (xoff & xa) || (ldx & xa) || xoff + cin > ldx || yoff + cout > ldy || cin_w > cin || cout_w > cout)
    return DIS_ERR_BAD_SHAPE;
  if (cout % 4 == 0 && ((yoff & 3) || (ldy & 3))) return DIS_ERR_BAD_SHAPE;
  (void)ya;
  hipStream_t s = (hipStream_t)stream;
  GenArgsB a;
  a.x = x; a.w = nullptr; a.bias = bias; a.y = y;
  a.n = n; a.hin = hin; a.win = win; a.ldx = ldx; a.xoff = xoff; a.cin = cin;
  a.hf = hout; a.wf = wout; a.ldy = ldy; a.yoff = yoff; a.cout = cout; a.act = act;
  short tsrc[CB_MAXTAPS];
  const long kk = (long)k * k;
  bf16_t* wp = (bf16_t*)wpack;
  // (split-K partial sums of the small maps: behind the four packing slices)
  a.ksplit = 1;
  a.nph = 1;
  a.skcap = dis_convb_splitk_workspace(mode, x_bf16, n, hin, win, hout, wout, cin, cout, k, stride, pad);
  a.skpart = a.skcap > 0 ? (float*)(wp + 4 * dis_convb_pack_workspace(cin, cout, k)) : nullptr;
  if (mode == DIS_CONVG_CONV || mode == DIS_CONVG_TCONV_DGRAD) {
    if (mode == DIS_CONVG_CONV) {
      if (hout != (hin + 2 * pad - k) / stride + 1 || wout != (win + 2 * pad - k) / stride + 1) return DIS_ERR_BAD_SHAPE;
    } else if (stride != 2) {
      return DIS_ERR_UNSUPPORTED;
    }
    a.hv = hout; a.wv = wout; a.S = stride; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
    a.ntaps = k * k;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        a.tdy[ky * k + kx] = (short)(ky - pad);
        a.tdx[ky * k + kx] = (short)(kx - pad);
        tsrc[ky * k + kx] = (short)(ky * k + kx);
      }
    return cb_run(a, x_bf16, y_bf16, w, wp, cin_w, cout_w, kk, (long)cin_w * kk, tsrc, s);
  }
  const long s_ci = (long)cout_w * kk, s_co = kk;
  if (stride == 1) {
    a.hv = hout; a.wv = wout; a.S = 1; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
    a.ntaps = k * k;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        a.tdy[ky * k + kx] = (short)(pad - ky);
        a.tdx[ky * k + kx] = (short)(pad - kx);
        tsrc[ky * k + kx] = (short)(ky * k + kx);
      }
    return cb_run(a, x_bf16, y_bf16, w, wp, cin_w, cout_w, s_ci, s_co, tsrc, s);
  }
  const long pstride = dis_convb_pack_workspace(cin, cout, k);
  static const bool nofuse = getenv("DIS_CONVB_NO_FUSED_PHASES") != nullptr;
  if (!nofuse && k * k <= 255) {
    // the four parity classes in ONE launch of the halo kernel (they read the same input tile): taps listed class by class
    GenArgsB b = a;
    b.hv = (hout + 1) / 2; b.wv = (wout + 1) / 2; b.S = 1;
    b.osy = 2; b.ooy = 0; b.osx = 2; b.oox = 0;
    b.nph = 4;
    int nt = 0;
    bool ok = true;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int ph = py * 2 + px, t0 = nt;
        for (int ky = 0; ky < k; ++ky) {
          if ((py + pad - ky) & 1) continue;
          for (int kx = 0; kx < k; ++kx) {
            if ((px + pad - kx) & 1) continue;
            b.tdy[nt] = (short)cb_floordiv2(py + pad - ky);
            b.tdx[nt] = (short)cb_floordiv2(px + pad - kx);
            tsrc[nt] = (short)(ky * k + kx);
            ++nt;
          }
        }
        b.ph_k0[ph] = (unsigned char)t0; b.ph_kn[ph] = (unsigned char)(nt - t0);
        b.ph_oy[ph] = (unsigned char)py; b.ph_ox[ph] = (unsigned char)px;
        ok = ok && nt > t0;
      }
    b.ntaps = nt;
    if (ok) {
      const int rc = cb_run(b, x_bf16, y_bf16, w, wp, cin_w, cout_w, s_ci, s_co, tsrc, s, true);
      if (rc != DIS_ERR_UNSUPPORTED) return rc;
    }
  }
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      GenArgsB b = a;
      b.hv = (hout - py + 1) / 2; b.wv = (wout - px + 1) / 2; b.S = 1;
      b.osy = 2; b.ooy = py; b.osx = 2; b.oox = px;
      int nt = 0;
      for (int ky = 0; ky < k; ++ky) {
        if ((py + pad - ky) & 1) continue;
        for (int kx = 0; kx < k; ++kx) {
          if ((px + pad - kx) & 1) continue;
          b.tdy[nt] = (short)cb_floordiv2(py + pad - ky);
          b.tdx[nt] = (short)cb_floordiv2(px + pad - kx);
          tsrc[nt] = (short)(ky * k + kx);
          ++nt;
        }
      }
      b.ntaps = nt;
      if (b.hv <= 0 || b.wv <= 0) continue;
      if (nt == 0) return DIS_ERR_UNSUPPORTED;
      int rc = cb_run(b, x_bf16, y_bf16, w, wp + (long)(py * 2 + px) * pstride, cin_w, cout_w, s_ci, s_co, tsrc, s);
      if (rc != DIS_OK) return rc;
    }
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[tap][xc][gc] = sum_pixels X[pixel + tap][xc] * G[pixel][gc], X / G bf16 or fp32
// (structure of convg_wgrad_kernel: one tap x TX x-channels x TG g-channels per workgroup, 32 pixels per step, split-K slabs)
// ------------------------------------------------------------------------------------------------
struct WgArgsB {
  const void* X;
  const void* G;
  float* part;  // [ksplit][tap][cXp][cGp]
  int n, hX, wX, ldX, xoff, cX;
  int hG, wG, ldG, goff, cG;
  int S, pad, k;
  int nxb, ngb;
  int mper;
  int cXp, cGp;
};
template <bool BF>
__device__ __forceinline__ float4 cb_load4(const void* p, long e) {
  if (BF) {
    const uint2 v = *(const uint2*)((const bf16_t*)p + e);
    return make_float4(cb_lo(v.x), cb_hi(v.x), cb_lo(v.y), cb_hi(v.y));
  }
  return *(const float4*)((const float*)p + e);
}
template <int MTW, int NTW, bool XB, bool GB>
__global__ __launch_bounds__(256) void convb_wgrad_kernel(WgArgsB a) {
  constexpr int TX = 32 * MTW, TG = 32 * NTW, XS = TX + 16, GS = TG + 16;
  constexpr int X_FL = 32 * XS, G_FL = 32 * GS;
  constexpr int NLX = TX / 32, NLG = TG / 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * (X_FL + G_FL)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int ntaps = a.k * a.k;
  const int tap = blockIdx.x % ntaps;
  const int rest = blockIdx.x / ntaps;
  const int xb = rest % a.nxb, gb = rest / a.nxb;
  const int ky = tap / a.k, kx = tap % a.k;
  const int M = a.n * a.hG * a.wG;
  const int m_lo = blockIdx.y * a.mper;
  const int m_hi = min(M, m_lo + a.mper);
  float4 rx[NLX], rg[NLG];
  auto prefetch = [&](int mbase) {
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      const int item = tid + j * 256;
      const int px = item / (TX / 4), q = item % (TX / 4);
      const int m = mbase + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = xb * TX + q * 4;
      if (m < m_hi && c < a.cX) {
        const int gx = m % a.wG, t = m / a.wG, gy = t % a.hG, nn = t / a.hG;
        const int iy = gy * a.S - a.pad + ky, ix = gx * a.S - a.pad + kx;
        if (iy >= 0 && iy < a.hX && ix >= 0 && ix < a.wX)
          v = cb_load4<XB>(a.X, (((long)nn * a.hX + iy) * a.wX + ix) * a.ldX + a.xoff + c);
      }
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NLG; ++j) {
      const int item = tid + j * 256;
      const int px = item / (TG / 4), q = item % (TG / 4);
      const int m = mbase + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = gb * TG + q * 4;
      if (m < m_hi && c < a.cG) v = cb_load4<GB>(a.G, (long)m * a.ldG + a.goff + c);
      rg[j] = v;
    }
  };
  auto stage = [&](int buf) {
    float* Xt = smem + buf * (X_FL + G_FL);
    float* Gt = Xt + X_FL;
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      const int item = tid + j * 256;
      *(float4*)(Xt + (item / (TX / 4)) * XS + (item % (TX / 4)) * 4) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < NLG; ++j) {
      const int item = tid + j * 256;
      *(float4*)(Gt + (item / (TG / 4)) * GS + (item % (TG / 4)) * 4) = rg[j];
    }
  };
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wx = wave & 1, wg = wave >> 1;
  if (m_lo < m_hi) {
    prefetch(m_lo);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int mb = m_lo; mb < m_hi; mb += 32) {
      const bool more = mb + 32 < m_hi;
      if (more) prefetch(mb + 32);
      const float* Xt = smem + buf * (X_FL + G_FL);
      const float* Gt = Xt + X_FL;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float av[MTW], bv[NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) av[mt] = Xt[(kk * 4 + lg) * XS + wx * 16 * MTW + mt * 16 + li];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) bv[nt] = Gt[(kk * 4 + lg) * GS + wg * 16 * NTW + nt * 16 + li];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
      }
      if (more) stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  float* out = a.part + ((long)blockIdx.y * ntaps + tap) * a.cXp * a.cGp;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int xc = xb * TX + wx * 16 * MTW + mt * 16 + lg * 4 + r;
        const int gc = gb * TG + wg * 16 * NTW + nt * 16 + li;
        out[(long)xc * a.cGp + gc] = acc[mt][nt][r];
      }
}
__global__ void convb_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int nsplit, int ntaps,
                                          int cXp, int cGp, int cXw, int cGw) {
  const long total = (long)ntaps * cXw * cGw;
  const long slab = (long)ntaps * cXp * cGp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cGw);
    long r = i / cGw;
    const int x = (int)(r % cXw);
    const int tap = (int)(r / cXw);
    const float* p = part + ((long)tap * cXp + x) * cGp + g;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += (double)p[k * slab];
    gw[((long)g * cXw + x) * ntaps + tap] = (float)s;
  }
}
// ---- weight gradient of a disparity head: ONE gradient channel (cG_w = 1), 3x3, stride 1, x bf16 ----
// dW[tap][c] = sum_o x[o + tap - pad][c] * g[o] is a reduction with 9 cX outputs: no matrix core needed, HBM-bound on the one
// read of x.  Thread = (8-channel chunk, pixel slot): per x pixel one 16-byte load and 9 gradient scalars (neighbouring
// output pixels: served by L1 / L2), 72 register accumulators; lanes of a chunk are folded with xor-shuffles, waves through LDS,
// workgroups through [block][tap][cX] slabs finished by convb_head_reduce_kernel (fp64, fixed order: deterministic).
#define CBH_BLOCKS 512
template <int C8, bool GB>
__global__ __launch_bounds__(256) void convb_head_wgrad_kernel(const bf16_t* __restrict__ X, int ldX, int xoff,
                                                               const void* __restrict__ G, int ldG, int goff,
                                                               float* __restrict__ part, int n, int h, int w, int pad) {
  constexpr int PPB = 256 / C8, CX = 8 * C8;
  __shared__ float red[4][9 * CX];
  const int chunk = threadIdx.x % C8, slot = threadIdx.x / C8;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  const long npix = (long)n * h * w;
  for (long p = (long)blockIdx.x * PPB + slot; p < npix; p += (long)gridDim.x * PPB) {
    const int xx = (int)(p % w);
    const long r = p / w;
    const int yy = (int)(r % h);
    const uint4 v = *(const uint4*)(X + p * ldX + xoff + chunk * 8);
    const float xv[8] = {cb_lo(v.x), cb_hi(v.x), cb_lo(v.y), cb_hi(v.y), cb_lo(v.z), cb_hi(v.z), cb_lo(v.w), cb_hi(v.w)};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int oy = yy - ky + pad, ox = xx - kx + pad;
        float g = 0.f;
        if ((unsigned)oy < (unsigned)h && (unsigned)ox < (unsigned)w) {
          const long o = p + (long)(pad - ky) * w + (pad - kx);
          g = GB ? __uint_as_float((unsigned)((const bf16_t*)G)[o * ldG + goff] << 16) : ((const float*)G)[o * ldG + goff];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] = fmaf(g, xv[j], acc[ky * 3 + kx][j]);
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[t][j];
#pragma unroll
      for (int m = C8; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
      if (lane < C8) red[wave][t * CX + chunk * 8 + j] = a;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * CX; i += 256)
    part[(long)blockIdx.x * (9 * CX) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// gw[0][x][tap] = sum over the workgroup slabs, one wave per output (lanes stride over the slabs in fp64, fixed fold order)
__global__ __launch_bounds__(256) void convb_head_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
                                                                int nslab, int cX, int cXw) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= 9 * cXw) return;
  const int tap = o / cXw, x = o % cXw;
  double s = 0.0;
  for (int k = lane; k < nslab; k += 64) s += (double)part[(long)k * (9 * cX) + tap * cX + x];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) gw[(long)x * 9 + tap] = (float)s;
}
static bool cbh_eligible(int x_bf16, int cX, int cG_w, int k, int stride, int ldX, int xoff) {
  return x_bf16 && cG_w == 1 && k == 3 && stride == 1 && (cX == 16 || cX == 32 || cX == 64 || cX == 128) &&
         !((ldX | xoff) & 7);
}
template <bool GB>
static void cbh_launch(int c8, int blocks, const void* X, int ldX, int xoff, const void* G, int ldG, int goff, float* part,
                       int n, int h, int w, int pad, hipStream_t s) {
#define CBH_CASE(C8_) \
  if (c8 == C8_)      \
    hipLaunchKernelGGL((convb_head_wgrad_kernel<C8_, GB>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)X, ldX, xoff, G, ldG, \
                       goff, part, n, h, w, pad);
  CBH_CASE(2) CBH_CASE(4) CBH_CASE(8) CBH_CASE(16)
#undef CBH_CASE
}

static void cbw_tiles(int cX, int cG, int* mtw, int* ntw) {
  *mtw = cX > 32 ? 2 : 1;
  *ntw = cG > 32 ? 2 : 1;
}
static void cbw_plan(int n, int hG, int wG, int cX, int cG, int k, int* nxb, int* ngb, int* nsplit, int* mper) {
  int mtw, ntw;
  cbw_tiles(cX, cG, &mtw, &ntw);
  const int TX = 32 * mtw, TG = 32 * ntw;
  *nxb = (cX + TX - 1) / TX;
  *ngb = (cG + TG - 1) / TG;
  const long base = (long)k * k * (*nxb) * (*ngb);
  const long M = (long)n * hG * wG;
  long sp = (2048 + base - 1) / base;
  const long maxsp = (M + 255) / 256;
  if (sp > maxsp) sp = maxsp;
  if (sp < 1) sp = 1;
  long mp = (M + sp - 1) / sp;
  mp = (mp + 31) / 32 * 32;
  sp = (M + mp - 1) / mp;
  *nsplit = (int)sp;
  *mper = (int)mp;
}
// bf16 x AND bf16 gy, >= 16 / 32 channels, 3x3 / 5x5 / 7x7 taps: the one-pass slice-pair kernel of conv2d.hip in its
// bf16-input form (conv_wgrad_bf16x3_kernel<..., BF = true>: one bf16 MFMA product per MAC, operands fetched from the
// [pixel][channel] LDS images with the transposing read)
long dis_wgrad_pairs_workspace(int n, int hX, int wX, int hG, int wG, int cX, int cG, int ldX, int ldG, int k,
                               int stride, int bf);
int dis_wgrad_pairs_run(const float* X, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const float* G, int ldG,
                        int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace, int n, int k,
                        int stride, int pad, int bf, hipStream_t s);

extern "C" long dis_convb_wgrad_workspace(int n, int hG, int wG, int cX, int cG, int k) {
  if (n <= 0 || hG <= 0 || wG <= 0 || cX <= 0 || cG <= 0 || k <= 0 || k * k > CB_MAXTAPS) return -1;
  int nxb, ngb, nsplit, mper, mtw, ntw;
  cbw_plan(n, hG, wG, cX, cG, k, &nxb, &ngb, &nsplit, &mper);
  cbw_tiles(cX, cG, &mtw, &ntw);
  const long f32 = (long)nsplit * k * k * (nxb * 32 * mtw) * (ngb * 32 * ntw);
  // (the slice-pair form, if dis_convb_wgrad takes it for this layer: the stride is not known here, so size for both)
  const long b1 = dis_wgrad_pairs_workspace(n, 1, 1, hG, wG, cX, cG, 8, 8, k, 1, 1);
  const long b2 = dis_wgrad_pairs_workspace(n, 1, 1, hG, wG, cX, cG, 8, 8, k, 2, 1);
  long b3 = b1 > b2 ? b1 : b2;
  if (cG <= 4 && k == 3 && b3 < (long)CBH_BLOCKS * 9 * cX) b3 = (long)CBH_BLOCKS * 9 * cX;  // head form
  return b3 > f32 ? b3 : f32;
}
template <bool XB, bool GB>
static void cbw_launch(const WgArgsB& a, int mtw, int ntw, dim3 grid, hipStream_t s) {
  if (mtw == 2 && ntw == 2) hipLaunchKernelGGL((convb_wgrad_kernel<2, 2, XB, GB>), grid, dim3(256), 0, s, a);
  else if (mtw == 2) hipLaunchKernelGGL((convb_wgrad_kernel<2, 1, XB, GB>), grid, dim3(256), 0, s, a);
  else if (ntw == 2) hipLaunchKernelGGL((convb_wgrad_kernel<1, 2, XB, GB>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((convb_wgrad_kernel<1, 1, XB, GB>), grid, dim3(256), 0, s, a);
}
// argument meaning of dis_convg_wgrad (conv_gen.hip); ld / offsets in elements; x_bf16 / g_bf16 as in dis_convb_run
extern "C" int dis_convb_wgrad(const void* X, int x_bf16, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const void* G,
                               int g_bf16, int ldG, int goff, int hG, int wG, int cG, int cG_w, float* grad_w,
                               float* workspace, int n, int k, int stride, int pad, void* stream) {
  if (!X || !G || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hX <= 0 || wX <= 0 || hG <= 0 || wG <= 0 || cX <= 0 || cG <= 0 || cX_w <= 0 || cG_w <= 0 || cX_w > cX ||
      cG_w > cG || k <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  if ((cX & 3) || (cG & 3) || (xoff & 3) || (goff & 3) || (ldX & 3) || (ldG & 3) || xoff + cX > ldX || goff + cG > ldG)
    return DIS_ERR_BAD_SHAPE;
  if (k * k > CB_MAXTAPS || (stride != 1 && stride != 2)) return DIS_ERR_UNSUPPORTED;
  if ((long)n * hG * wG > 2147483647L - 64) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (x_bf16 && g_bf16 && !((xoff | goff) & 7) && xoff + ((cX + 7) & ~7) <= ldX && goff + ((cG + 7) & ~7) <= ldG &&
      dis_wgrad_pairs_workspace(n, hX, wX, hG, wG, cX, cG, ldX, ldG, k, stride, 1) >= 0)
    return dis_wgrad_pairs_run((const float*)X, ldX, xoff, hX, wX, cX, cX_w, (const float*)G, ldG, goff, hG, wG, cG,
                               cG_w, grad_w, workspace, n, k, stride, pad, 1, s);
  if (cbh_eligible(x_bf16, cX, cG_w, k, stride, ldX, xoff) && hX == hG && wX == wG) {
    const long npix = (long)n * hG * wG;
    const int c8 = cX / 8, ppb = 256 / c8;
    long blocks = (npix + ppb - 1) / ppb;
    if (blocks > CBH_BLOCKS) blocks = CBH_BLOCKS;
    if (g_bf16) cbh_launch<true>(c8, (int)blocks, X, ldX, xoff, G, ldG, goff, workspace, n, hG, wG, pad, s);
    else cbh_launch<false>(c8, (int)blocks, X, ldX, xoff, G, ldG, goff, workspace, n, hG, wG, pad, s);
    const long tot = 9L * cX_w;
    hipLaunchKernelGGL(convb_head_reduce_kernel, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, s, (const float*)workspace,
                       grad_w, (int)blocks, cX, cX_w);
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  WgArgsB a;
  a.X = X; a.G = G; a.part = workspace;
  a.n = n; a.hX = hX; a.wX = wX; a.ldX = ldX; a.xoff = xoff; a.cX = cX;
  a.hG = hG; a.wG = wG; a.ldG = ldG; a.goff = goff; a.cG = cG;
  a.S = stride; a.pad = pad; a.k = k;
  int nsplit, mtw, ntw;
  cbw_plan(n, hG, wG, cX, cG, k, &a.nxb, &a.ngb, &nsplit, &a.mper);
  cbw_tiles(cX, cG, &mtw, &ntw);
  a.cXp = a.nxb * 32 * mtw;
  a.cGp = a.ngb * 32 * ntw;
  const dim3 grid((unsigned)(k * k * a.nxb * a.ngb), (unsigned)nsplit);
  if (x_bf16 && g_bf16) cbw_launch<true, true>(a, mtw, ntw, grid, s);
  else if (x_bf16) cbw_launch<true, false>(a, mtw, ntw, grid, s);
  else if (g_bf16) cbw_launch<false, true>(a, mtw, ntw, grid, s);
  else cbw_launch<false, false>(a, mtw, ntw, grid, s);
  const long total = (long)k * k * cX_w * cG_w;
  hipLaunchKernelGGL(convb_wgrad_reduce_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, (const float*)workspace,
                     grad_w, nsplit, k * k, a.cXp, a.cGp, cX_w, cG_w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// element-wise helpers on bf16 nhwc tensors
// ------------------------------------------------------------------------------------------------
// gpre = gy * act'(y) on channel ranges of wider buffers (4 channels per thread); gy, y, gpre bf16
__global__ void act_bwd_bf16_kernel(const bf16_t* __restrict__ gy, int ldg, const bf16_t* __restrict__ y, int ldy,
                                    bf16_t* __restrict__ gp, int act, long npix, int c4) {
  const long total = npix * c4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / c4;
    const int q = (int)(i - px * c4) * 4;
    const uint2 g = *(const uint2*)(gy + px * ldg + q);
    if (act == DIS_ACT_NONE) {
      *(uint2*)(gp + i * 4) = g;
    } else {
      const uint2 v = *(const uint2*)(y + px * ldy + q);
      *(uint2*)(gp + i * 4) = make_uint2(
          cb_pack2(cb_lo(g.x) * act_grad_from_out(cb_lo(v.x), act), cb_hi(g.x) * act_grad_from_out(cb_hi(v.x), act)),
          cb_pack2(cb_lo(g.y) * act_grad_from_out(cb_lo(v.y), act), cb_hi(g.y) * act_grad_from_out(cb_hi(v.y), act)));
    }
  }
}
// the same with an fp32 result (first layer of DispNetS: its weight gradient runs on the fp32 one-pass kernel, x being fp32)
__global__ void act_bwd_bf16_f32_kernel(const bf16_t* __restrict__ gy, int ldg, const bf16_t* __restrict__ y, int ldy,
                                        float* __restrict__ gp, int act, long npix, int c4) {
  const long total = npix * c4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / c4;
    const int q = (int)(i - px * c4) * 4;
    const uint2 g = *(const uint2*)(gy + px * ldg + q);
    float4 o = make_float4(cb_lo(g.x), cb_hi(g.x), cb_lo(g.y), cb_hi(g.y));
    if (act != DIS_ACT_NONE) {
      const uint2 v = *(const uint2*)(y + px * ldy + q);
      o.x *= act_grad_from_out(cb_lo(v.x), act), o.y *= act_grad_from_out(cb_hi(v.x), act);
      o.z *= act_grad_from_out(cb_lo(v.y), act), o.w *= act_grad_from_out(cb_hi(v.y), act);
    }
    *(float4*)(gp + i * 4) = o;
  }
}
extern "C" int dis_act_bwd_bf16_f32(const void* gy, int ldg, const void* y, int ldy, float* gpre, int act, long npix, int c,
                                    void* stream) {
  if (!gy || !gpre || (act != DIS_ACT_NONE && !y)) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || ldg < c || (act != DIS_ACT_NONE && ldy < c)) return DIS_ERR_BAD_SHAPE;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || ((uintptr_t)gy & 7) || ((uintptr_t)y & 7) || ((uintptr_t)gpre & 15))
    return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(act_bwd_bf16_f32_kernel, dim3(dis_ew_grid(npix * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gy, ldg, (const bf16_t*)(y ? y : gy), ldy, gpre, act, npix, c / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_act_bwd_bf16(const void* gy, int ldg, const void* y, int ldy, void* gpre, int act, long npix, int c,
                                void* stream) {
  if (!gy || !gpre || (act != DIS_ACT_NONE && !y)) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || ldg < c || (act != DIS_ACT_NONE && ldy < c)) return DIS_ERR_BAD_SHAPE;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || ((uintptr_t)gy & 7) || ((uintptr_t)y & 7) || ((uintptr_t)gpre & 7))
    return DIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(act_bwd_bf16_kernel, dim3(dis_ew_grid(npix * (c / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gy, ldg, (const bf16_t*)(y ? y : gy), ldy, (bf16_t*)gpre, act, npix, c / 4);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// dis_act_bwd_bf16 that also leaves the bias gradient (column sums of the unrounded gpre) behind: the gradient tensor is read
// once for both.  c / 4 divides 256: a thread keeps its 4 channels while it walks down the pixels.
#define CSB_BLOCKS 1024
// V = channels per thread: 8 (16-byte vectors; c, ldg, ldy multiples of 8 and 16-byte aligned pointers) or 4
template <int V>
__global__ __launch_bounds__(256) void act_bwd_bf16_bias_kernel(const bf16_t* __restrict__ gy, int ldg,
                                                                 const bf16_t* __restrict__ y, int ldy,
                                                                 bf16_t* __restrict__ gp, int act, long npix, int cv,
                                                                 float* __restrict__ part) {
  __shared__ float red[256 * V];
  const int chunk = threadIdx.x % cv, rows = 256 / cv, row = threadIdx.x / cv, q = chunk * V;
  float s[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = 0.f;
  const int c = cv * V;
  for (long px = (long)blockIdx.x * rows + row; px < npix; px += (long)gridDim.x * rows) {
    unsigned g[V / 2], v[V / 2], o[V / 2];
    // (gy and y are dead after this pass - the backward of the layer that consumed y has already run: non-temporal loads)
    typedef unsigned ab_u4 __attribute__((ext_vector_type(4)));
    typedef unsigned ab_u2 __attribute__((ext_vector_type(2)));
    if (V == 8) {
      const ab_u4 t = __builtin_nontemporal_load((const ab_u4*)(gy + px * ldg + q));
      g[0] = t[0], g[1] = t[1], g[V / 2 - 2] = t[2], g[V / 2 - 1] = t[3];
    } else {
      const ab_u2 t = __builtin_nontemporal_load((const ab_u2*)(gy + px * ldg + q));
      g[0] = t[0], g[1] = t[1];
    }
    if (act != DIS_ACT_NONE) {
      if (V == 8) {
        const ab_u4 t = __builtin_nontemporal_load((const ab_u4*)(y + px * ldy + q));
        v[0] = t[0], v[1] = t[1], v[V / 2 - 2] = t[2], v[V / 2 - 1] = t[3];
      } else {
        const ab_u2 t = __builtin_nontemporal_load((const ab_u2*)(y + px * ldy + q));
        v[0] = t[0], v[1] = t[1];
      }
    }
#pragma unroll
    for (int j = 0; j < V / 2; ++j) {
      float a = cb_lo(g[j]), b = cb_hi(g[j]);
      if (act != DIS_ACT_NONE) a *= act_grad_from_out(cb_lo(v[j]), act), b *= act_grad_from_out(cb_hi(v[j]), act);
      s[2 * j] += a;
      s[2 * j + 1] += b;
      o[j] = cb_pack2(a, b);
    }
    if (V == 8) *(uint4*)(gp + px * c + q) = make_uint4(o[0], o[1], o[V / 2 - 2], o[V / 2 - 1]);
    else *(uint2*)(gp + px * c + q) = make_uint2(o[0], o[1]);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = s[j];
  __syncthreads();
  for (int t = threadIdx.x; t < c; t += 256) {
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += red[(r * cv + t / V) * V + t % V];
    part[(long)blockIdx.x * c + t] = acc;
  }
}

// dst[pixel * ldd + j] = src[pixel * lds + j] (j < c), 0 for c <= j < c + czero; src fp32 or bf16, dst bf16
template <bool SB>
__global__ void copy_channels_bf16_kernel(const void* __restrict__ src, int lds, bf16_t* __restrict__ dst, int ldd,
                                          long npix, int c, int czero) {
  const int ct = c + czero;
  const long total = npix * ct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / ct;
    const int j = (int)(i - px * ct);
    float v = 0.f;
    if (j < c) v = SB ? cb_lo(((const bf16_t*)src)[px * lds + j]) : ((const float*)src)[px * lds + j];
    dst[px * ldd + j] = (bf16_t)(cb_pack2(v, 0.f) & 0xffffu);
  }
}
extern "C" int dis_copy_channels_bf16(const void* src, int src_bf16, int lds, void* dst, int ldd, long npix, int c,
                                      int czero, void* stream) {
  if (!src || !dst) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || czero < 0 || lds < c || ldd < c + czero) return DIS_ERR_BAD_SHAPE;
  const dim3 grid(dis_ew_grid(npix * (c + czero), 256));
  if (src_bf16)
    hipLaunchKernelGGL(copy_channels_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, lds, (bf16_t*)dst,
                       ldd, npix, c, czero);
  else
    hipLaunchKernelGGL(copy_channels_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, lds, (bf16_t*)dst,
                       ldd, npix, c, czero);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// column sums of a bf16 nhwc tensor (bias gradients): out[ch] = sum_pixels G[pixel][goff + ch], fp32 block partials, fp64 total
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ G, int ldG, int goff, long npix, int c,
                                                           float* __restrict__ part) {
  __shared__ float red[256];
  const long lo = npix * blockIdx.x / gridDim.x, hi = npix * (blockIdx.x + 1) / gridDim.x;
  for (int c0 = 0; c0 < c; c0 += 256) {
    const int cw = min(256, c - c0);
    const int rows = 256 / cw > 0 ? 256 / cw : 1;
    const int ch = threadIdx.x % cw, row = threadIdx.x / cw;
    float s0 = 0.f, s1 = 0.f;
    if (row < rows) {
      const bf16_t* gp = G + goff + c0 + ch;
      long p = lo + row;
      for (; p + rows < hi; p += 2L * rows) {
        s0 += cb_lo(gp[p * ldG]);
        s1 += cb_lo(gp[(p + rows) * ldG]);
      }
      for (; p < hi; p += rows) s0 += cb_lo(gp[p * ldG]);
    }
    __syncthreads();
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < cw) {
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += red[k * cw + threadIdx.x];
      part[(long)blockIdx.x * c + c0 + threadIdx.x] = t;
    }
  }
}
__global__ __launch_bounds__(256) void colsum_bf16_final_kernel(const float* __restrict__ part, int nblocks, int c,
                                                                 float* __restrict__ out) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double s = 0.0;
  for (int k = lane; k < nblocks; k += 64) s += (double)part[(long)k * c + ch];
  s = wave_sum_d(s);
  if (lane == 0) out[ch] = (float)s;
}
// gpre = gy * act'(y) as dis_act_bwd_bf16, and bias_grad[ch] = sum_pixels gpre[pixel][ch] (fp32, summed before the bf16 rounding of
// gpre); workspace: dis_colsum_bf16_workspace(c) floats.  c / 4 must divide 256 (DIS_ERR_UNSUPPORTED otherwise: the caller uses
// dis_act_bwd_bf16 + dis_colsum_bf16)
extern "C" int dis_act_bwd_bf16_bias(const void* gy, int ldg, const void* y, int ldy, void* gpre, int act, long npix, int c,
                                     float* bias_grad, float* workspace, void* stream) {
  if (!gy || !gpre || !bias_grad || !workspace || (act != DIS_ACT_NONE && !y)) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || ldg < c || (act != DIS_ACT_NONE && ldy < c)) return DIS_ERR_BAD_SHAPE;
  if ((c & 3) || (ldg & 3) || (ldy & 3) || ((uintptr_t)gy & 7) || ((uintptr_t)y & 7) || ((uintptr_t)gpre & 7) ||
      c > 1024 || 256 % (c / 4))
    return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int rows = 256 / (c / 4);
  long nb = (npix + rows - 1) / rows;
  if (nb > CSB_BLOCKS) nb = CSB_BLOCKS;
  const bool v8 = !(c & 7) && !(ldg & 7) && !(ldy & 7) && !((uintptr_t)gy & 15) && !((uintptr_t)y & 15) &&
                  !((uintptr_t)gpre & 15) && 256 % (c / 8) == 0;
  if (v8) {
    const int rows8 = 256 / (c / 8);
    long nb8 = (npix + rows8 - 1) / rows8;
    if (nb8 > CSB_BLOCKS) nb8 = CSB_BLOCKS;
    nb = nb8;
    hipLaunchKernelGGL(act_bwd_bf16_bias_kernel<8>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)gy, ldg,
                       (const bf16_t*)(y ? y : gy), ldy, (bf16_t*)gpre, act, npix, c / 8, workspace);
  } else {
    hipLaunchKernelGGL(act_bwd_bf16_bias_kernel<4>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)gy, ldg,
                       (const bf16_t*)(y ? y : gy), ldy, (bf16_t*)gpre, act, npix, c / 4, workspace);
  }
  hipLaunchKernelGGL(colsum_bf16_final_kernel, dim3(dis_cdiv(c, 4)), dim3(256), 0, s, (const float*)workspace, (int)nb, c,
                     bias_grad);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" long dis_colsum_bf16_workspace(int c) { return c > 0 ? (long)CSB_BLOCKS * c : -1; }
extern "C" int dis_colsum_bf16(const void* G, int ldG, int goff, long npix, int c, float* out, float* workspace,
                               void* stream) {
  if (!G || !out || !workspace) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || goff < 0 || goff + c > ldG) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  int nb = (int)((npix + 255) / 256);
  if (nb > CSB_BLOCKS) nb = CSB_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)G, ldG, goff, npix, c, workspace);
  hipLaunchKernelGGL(colsum_bf16_final_kernel, dim3(dis_cdiv(c, 4)), dim3(256), 0, s, (const float*)workspace, nb, c, out);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
