// Input gradient AND weight gradient of a 3x3 stride-1 pad-1 convolution C -> C (C = 32) in ONE launch (round 6).
//
// Reference: every Conv2d of ResNetBlock / Block2D3D is one autograd node (model/multi_frame_networks.py:338-345, 514-542); until
// round 5 its backward was two launches here - conv_f16x2_kernel (input gradient) and conv_wgrad_f16x2_kernel (weight gradient) - and
// the second one re-read from HBM what the first had held in LDS a few microseconds earlier: gy (or gpre, the GroupNorm-backward pass
// applied on load, which the first launch had to WRITE for the second) and x.  Here the 18 x 18 gy halo a tile of the input gradient
// stages serves both products:
//     gx[q][ci]        = sum_tap sum_co gy[q + 1 - tap][co] W[co][ci][tap]          (contraction over channels, per tile)
//     dW[tap][ci][co]  = sum_q'  x[q'][ci] gy[q' + 1 - tap][co]                     (contraction over the pixels q' a tile OWNS)
// Two-term fp16 operands, three products on v_mfma_f32_16x16x32_f16, as conv_f16x2.hip (same split, same per-tile scale, same order of
// the accumulation per output element: gx is BIT-identical to conv_f16x2_kernel's).
//
// Shape of the kernel: 4 waves of 64 lanes, ONE per SIMD, 512 registers each, one workgroup per CU (grid = #CUs, persistent over the
// tiles of its XCD's share).  Per 16 x 16 tile:
//   top      maxima of the next halo and of the x tile per wave (DPP), exchanged through LDS
//   barrier A  (every wave has left the previous tile: halo and x tile may be overwritten; the maxima are visible)
//   staging  the gy halo (18 x 18 x 32, two fp16 planes, per-tile scale 2^sg) and the x tile (16 x 16 x 32, scale 2^(S - sg)) go to
//            LDS, ONE copy of each shared by the four waves
//   barrier B
//   D        input gradient: wave w owns tile rows 4 w .. 4 w + 3 (8 accumulator tiles), 9 taps x 3 products; the next tile's halo /
//            x loads are issued inside the first k-steps (their registers were emptied by the staging)
//   W        weight gradient: wave (ah, bh) = (wave >> 1, wave & 1) owns the 9 tap tiles of (ci half ah, co half bh) - 36 accumulation
//            registers, not 144 - over ALL 256 pixels of the tile: 8 k-steps of 32 pixels, x^T and the shifted gy fragments read back
//            with ds_read_b64_tr_b16 (an even halo row pair serves tap row 0 of one k-step and tap row 2 of the next); the input
//            gradient's epilogue (8 stores, channel sums) rides in its first k-steps
// No cross-wave reduction of dW at all: a wave's 9 tiles are its own elements of the workgroup's slab, written once per workgroup and
// summed over workgroups by wgrad_reduce_kernel (fixed order, fp64).
// dW scales: the gy halo carries the input gradient's per-tile scale 2^sg(t); x is split with 2^(S - sg(t)), S a WORKGROUP-uniform
// exponent set by the first tile that has something to add (FB_SMARGIN bits of headroom), so that every term of the sum carries 2^S.
// A later tile whose product magnitude exceeds the headroom makes the workgroup LEAVE the tile loop (inside it only matrix instructions
// touch the dW accumulators - a rescale branch in the loop makes the allocator keep them in vector registers): the accumulators go to
// the slab (scaled by 2^-S, added to what an earlier pass left there), a new pass starts from zero with a new S at the same tile.
//
// History (profiles/r6_bwd_fused.md): the first form of this kernel gave every wave ALL 36 dW tiles over its own 64 pixels (144
// accumulation registers, wave-private x strips, two halo buffers, 162 KB of LDS) - correct, and 1.5 - 3.6 x slower than the two
// launches in every form behind a GroupNorm: 530 - 600 registers of state against 512.  The tile split above needs 110 - 256 + 256 and
// is 0.79 - 0.95 of the two launches (same-box A/B per form); the step uses it wherever an instance exists (DIS_BWD_FUSED=0: off).
// LDS: weights 36.9 KB + halo 51.8 KB + x tile 41.0 KB + 1.2 KB = 130.9 KB.
#include "conv_args.h"
#include <type_traits>


typedef short fb_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x8 fb_tr_read8(const unsigned short* p0, const unsigned short* p1) {
  const fb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fb_s16x4*)p0);
  const fb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fb_s16x4*)p1);
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the tap / channel -> k-slot map of conv_f16x2.hip's f2_weight for 32 channels, input-gradient order (flipped taps, transposed)
__device__ __forceinline__ float fb_weight(const float* w, int stride_row, int wo, int wi, int ks, int lg, int j, int co) {
  const int c = 8 * lg + j;
  return (c < wo && co < wi) ? w[c * stride_row + co * 9 + (8 - ks)] : 0.f;
}
template <int I, int N, class F>
__device__ __forceinline__ void fb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    fb_static_for<I + 1, N>(f);
  }
}

#ifndef FB_ZLIT
#define FB_ZLIT 1   // the first product of an input-gradient accumulator starts from a zero literal (no register moves)
#endif
#ifndef FB_BLIND
#define FB_BLIND 0   // (re-blinding p0 with an empty asm statement produced WRONG results in the INACT instances: off)
#endif
// Diagnostic build only (scripts/diag/fb_stamps.py, -DFB_STAMP): per-wave s_memtime sums of the phases of a tile
#ifdef FB_STAMP
__device__ unsigned long long fb_stamps[256 * 4 * 12];
#define FB_T(k)                                                   \
  {                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    st_[k] += now_ - last_;                                       \
    last_ = now_;                                                 \
  }
extern "C" int dis_debug_fb_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fb_stamps), sizeof(fb_stamps)); }
#else
#define FB_T(k)
#endif
#define FB_TR 16
#define FB_TC 16
#ifndef FB_XGRP
#define FB_XGRP 2   // x pieces staged together, stage by stage (independent chains: one wave per SIMD exposes every dependent latency)
#endif
#ifndef FB_SMARGIN
#define FB_SMARGIN 6   // bits of headroom the dW exponent keeps when it is (re)set: a later tile may be 2^6 larger before the accumulators move again
#endif
struct FbCfg {
  static constexpr int C = 32, IR = FB_TR + 2, IC = FB_TC + 2, CV = C / 4, NP = 2, PS = 80, NT = 2, KS = 9;
  static constexpr int NW = 4, NTHR = 64 * NW, MT = FB_TR / NW;
  static constexpr int W_U16 = KS * NP * 4 * C * 8, X_U16 = IR * IC * PS, XT_U16 = FB_TR * FB_TC * PS;
  static constexpr int NITEMS = IR * IC * CV, NLOAD = (NITEMS + NTHR - 1) / NTHR, NPIECE = MT * NT;
  static constexpr int SMALL_U16 = 32 + 64 + NW * 2 * C * 2 + C + 8;   // red (8 doubles), maxima [parity][gy | x][wave], abw [wave][2 C] floats, write pad
  static constexpr int LDS_BYTES = (W_U16 + X_U16 + XT_U16 + SMALL_U16) * 2;
  static constexpr int NACC = 9;   // dW accumulator tiles per wave: the 9 taps of its (ci half, co half)
};
static_assert(FbCfg::LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(32 * (32 * 9 + 1) * 4 <= FbCfg::XT_U16 * 2, "the weight prologue's fp32 scratch aliases the x tile");

// INACT / INCOEF / ACCUM / EPIAB / EPIACT: conv_f16x2_kernel's forms of the input gradient (see there).
// XSRC: where the weight gradient's x comes from - 0: a.wx, 1: c.ab_x (the GroupNorm input the channel sums are formed with IS the
// conv's input: conv2d_gn_in), 2: c.ab_act_y (the activation output the result is multiplied with IS the conv's input: ResNetBlock
// chains); x always has its own registers (fetched one tile ahead), the epilogue's operands are fetched in their own tile.
// XGN: x is staged as GroupNorm(x).  GST: the staged values of the pixels a tile owns are stored to c.gnb_out as well (the other
// launches of a multi-source node read them).
template <int INACT, bool INCOEF, bool ACCUM, bool EPIAB, int EPIACT, int XSRC, bool XGN, bool GST>
__global__ __launch_bounds__(256) void conv_bwd_fused_kernel(FbArgs fa_) {
  using K = FbCfg;
  const ConvArgs& a = fa_.c;
  constexpr int C = K::C, IC = K::IC, PS = K::PS, NT = K::NT, KS = K::KS, NLOAD = K::NLOAD, NPIECE = K::NPIECE, CV = K::CV, NP = K::NP;
  constexpr int MT = K::MT, NW = K::NW, NTHR = K::NTHR;
  constexpr bool IN2 = INACT != 0 || INCOEF;
  // RIDE: the next tile's halo items become final under the dW products of the current tile (prep_item in W) instead of at the top of
  // their own tile.  Not where the accumulating GroupNorm-backward forms already fill the register file: measured 163 -> 188 us
  // (coef + act' + accumulate) and 196 -> 222 us (chain) with it, 139 -> 129 / 164 -> 157 us for the forms that have room.
  constexpr bool RIDE = !(INCOEF && ACCUM);
  // XSH: x IS one of the epilogue's operands (XSRC 1: the GroupNorm input of the channel sums, XSRC 2: the activation output): one
  // register set and one fetch serve both.  The epilogue of tile t reads piece i in W's k-steps 0 - 3, so the NEXT tile's x pieces
  // are requested right behind it there (not in D's last k-steps): 32 registers and 8 loads per thread and tile less.
  constexpr bool XSH = XSRC != 0;
  static_assert(EPIACT == 0 || EPIAB, "activation gradient at the output: only with the channel sums");
  static_assert(XSRC == 0 || (XSRC == 1 && EPIAB) || (XSRC == 2 && EPIACT), "shared x operand");
  static_assert(!GST || INCOEF, "gpre store: only where the operand is formed on load");
#ifdef FB_STAMP
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* wl = smem16;                              // weights, fragment order, two planes
  unsigned short* xh = smem16 + K::W_U16;                   // the gy halo of the current tile [18 x 18 pixels][plane][channel]
  unsigned short* xt = smem16 + K::W_U16 + K::X_U16;        // the x tile of the current tile  [16 x 16 pixels][plane][channel]
  unsigned short* small = smem16 + K::W_U16 + K::X_U16 + K::XT_U16;
  double* red = (double*)small;
  float* mxs = (float*)(small + 32);      // [parity][gy | x][wave]
  float* abw = (float*)(small + 96);      // EPIAB: [wave][2 C]
  unsigned short* pad16 = small + 96 + NW * 2 * C * 2 + C;   // (idle threads of the last round write pad16 - C and pad16)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4, tq = li >> 2, tp = li & 3;
  const int ah = wave >> 1, bh = wave & 1;   // dW: this wave's (ci half, co half)
  const int tiles_x = (a.wv + FB_TC - 1) / FB_TC, tiles_y = (a.hv + FB_TR - 1) / FB_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);
  const int d_tx = per % tiles_x, d_ty = (per / tiles_x) % tiles_y, d_n = per / (tiles_x * tiles_y);

  // ---- halo items of this thread (conv_f16x2_kernel's, 256 threads): item it = float4 vv of halo pixel pix = p0 + 32 it, p0 = thread / 8.
  // Nothing per item is kept in registers: row / column follow from p0 and compile-time constants (32 it = 18 A + B), the LDS address
  // is a constant offset from one base.
  float4 pre[NLOAD], pre2[IN2 ? NLOAD : 1];
  const int p0 = (int)threadIdx.x >> 3;
  const int vv4 = ((int)threadIdx.x & 7) * 16;   // byte offset of this thread's 4 channels within a pixel (the same for all its items)
  auto item_rc = [&](int it, int& r, int& c) __attribute__((always_inline)) {
    const int A = (32 * it) / IC, B = (32 * it) % IC;
    const int sft = p0 + B;
    const int k = sft >= 2 * IC ? 2 : (sft >= IC ? 1 : 0);
    r = A + k;
    c = sft - IC * k;
    if ((it + 1) * NTHR > K::NITEMS) c = ((int)threadIdx.x + it * NTHR) < K::NITEMS ? c : 0x40000000;   // (past the end of the halo: never in range)
  };
  auto item_own = [&](int it) -> bool {
    int r, c;
    item_rc(it, r, c);
    return r >= 1 && r <= FB_TR && c >= 1 && c <= FB_TC;
  };
  const unsigned x_bytes = (unsigned)a.hin * a.win * (C * 4u), y_bytes = (unsigned)a.hf * a.wf * (C * 4u);
  struct Pf {
    const float* x;
    unsigned bytes;
    int iy0, ix0, off0;
  };
  auto pf_make = [&](int n, int ty, int tx, bool live) -> Pf {
#ifdef FB_KO_LOAD   // (diagnostic: every halo load reads tile (1, 1) of sample 0 - cache hits)
    n = 0, ty = 1, tx = 1;
#endif
    Pf f;
    f.iy0 = ty * FB_TR - 1;
    f.ix0 = tx * FB_TC - 1;
    f.off0 = (f.iy0 * a.win + f.ix0) * (C * 4);
    f.x = a.x + (long)n * a.hin * a.win * C;
    f.bytes = live ? x_bytes : 0u;
    return f;
  };
  auto pf_issue = [&](const Pf& f, int it) {
    int r_, c_;
    item_rc(it, r_, c_);
    const int ix = f.ix0 + c_;
    const unsigned off = (unsigned)ix < (unsigned)a.win ? (unsigned)(f.off0 + (r_ * a.win + c_) * (C * 4) + vv4) : BX_OOB;
    pre[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(f.x, f.bytes), off, 0, 0));
    if (IN2)
      pre2[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.xact + (f.x - a.x), f.bytes), off, 0, 0));
  };
  float4 cf_k1 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cf_kx = 0.f, cf_k0 = 0.f;
  int cf_n = -1;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // bias gradient: this thread's 4 channels over the pixels its tiles own
  // final fp32 values of a tile's halo items (in place) and this lane's largest magnitude.  One wave per SIMD issues a vector
  // instruction per >= 4 cycles and nothing else runs meanwhile: at the top of a tile these 170 - 500 instructions were 1.2 - 2.5 k
  // exposed cycles, so the NEXT tile's items are finished under the dW products of the current one (prep_item rides in W; the first
  // tile's in the prologue).
  auto prep_cf = [&](int n_) {
    if (INCOEF && n_ != cf_n) {
      cf_n = n_;
      const float* cf = a.gnb_coef + (long)n_ * (C + 2);
      cf_k1 = *(const float4*)(cf + ((int)threadIdx.x % CV) * 4);
      cf_kx = cf[C];
      cf_k0 = cf[C + 1];
    }
  };
  auto prep_item = [&](const Pf& f, int it, float& m) __attribute__((always_inline)) {
    float4 v = pre[it];
    if (INCOEF) {   // (gn_apply_coef_kernel's arithmetic, bit for bit; padding: g = q = 0 would give k0, which must not be staged)
      const float4 q = pre2[it];
      int r_, c_;
      item_rc(it, r_, c_);
      const int ix = f.ix0 + c_;
      const unsigned off = (unsigned)ix < (unsigned)a.win ? (unsigned)(f.off0 + (r_ * a.win + c_) * (C * 4) + vv4) : BX_OOB;
      v.x = __builtin_fmaf(v.x, cf_k1.x, __builtin_fmaf(q.x, cf_kx, cf_k0));
      v.y = __builtin_fmaf(v.y, cf_k1.y, __builtin_fmaf(q.y, cf_kx, cf_k0));
      v.z = __builtin_fmaf(v.z, cf_k1.z, __builtin_fmaf(q.z, cf_kx, cf_k0));
      v.w = __builtin_fmaf(v.w, cf_k1.w, __builtin_fmaf(q.w, cf_kx, cf_k0));
      if (INACT) {
        v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
        v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
      }
#ifndef FB_KO_INSIDE   // (diagnostic: padding is not zeroed - WRONG at the image border; what the selects cost)
      // (padding must not be staged as k0.  Knock-out: these selects are 4 - 5 % of the forms with channel sums, the bias sums below
      //  2 - 3 %; skipping them behind a workgroup-uniform "the whole halo lies inside the image" branch was measured SLOWER
      //  (151.7 -> 154.2 us: the branch splits the block the riding arithmetic is scheduled in) - they stay unconditional)
      {
        const bool inside = off < f.bytes;
        v = inside ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#endif
      if (GST) {
        const u32x4 sv = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(sv, bx_rsrc(a.gnb_out + (f.x - a.x), f.bytes), item_own(it) ? off : BX_OOB, 0, 0);
      }
    } else if (INACT) {
      const float4 q = pre2[it];
      v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
      v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
    }
    pre[it] = v;
#ifndef FB_KO_BSUM   // (diagnostic: no bias gradient - what its selects and adds cost)
    {   // (bias gradient: the pixels this tile owns; a select, not a branch)
      const bool own = item_own(it);
      bsum.x += own ? v.x : 0.f, bsum.y += own ? v.y : 0.f, bsum.z += own ? v.z : 0.f, bsum.w += own ? v.w : 0.f;
    }
#endif
    m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.x)), fabsf(v.y));
    m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.z)), fabsf(v.w));
  };
  const int lds_item = p0 * PS + ((int)threadIdx.x & 7) * 4;
  auto stage_item = [&](int it, float sc) {
    const float4 v = pre[it];
    unsigned a1, a2, b1, b2;
    f2_split_pair_scaled(v.x, v.y, sc, a1, a2);
    f2_split_pair_scaled(v.z, v.w, sc, b1, b2);
    const int idx = (int)threadIdx.x + it * NTHR;
    unsigned short* p = xh + lds_item + it * (32 * PS);   // pixel p0 + 32 it, channels 4 vv ..: one base, constant offsets
    if ((it + 1) * NTHR > K::NITEMS) p = idx < K::NITEMS ? p : pad16 - C;
    *(uint2*)(p) = make_uint2(a1, b1);
    *(uint2*)(p + C) = make_uint2(a2, b2);
  };

  // two halo items, stage by stage (see the staging phase)
  auto stage_item2 = [&](int it, float sc) __attribute__((always_inline)) {
    const float4 va = pre[it], vb = pre[it + 1];
    const float x_[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    f32x2 v_[4], r_[4];
    f16x2_t h1_[4], h2_[4];
#ifdef FB_KO_SPLIT   // (diagnostic: the staged bits are not the split - WRONG results; what the split's instructions cost)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = (int)threadIdx.x + (it + j) * NTHR;
      unsigned short* p = xh + lds_item + (it + j) * (32 * PS);
      if ((it + j + 1) * NTHR > K::NITEMS) p = idx < K::NITEMS ? p : pad16 - C;
      *(uint2*)(p) = make_uint2(__float_as_uint(x_[4 * j]) & 0x3fff3fffu, __float_as_uint(x_[4 * j + 1]) & 0x3fff3fffu);
      *(uint2*)(p + C) = make_uint2(__float_as_uint(x_[4 * j + 2]) & 0x3fff3fffu, __float_as_uint(x_[4 * j + 3]) & 0x3fff3fffu);
    }
    (void)sc; (void)v_; (void)r_; (void)h1_; (void)h2_;
    return;
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) v_[k] = (f32x2){x_[2 * k] * sc, x_[2 * k + 1] * sc};
#pragma unroll
    for (int k = 0; k < 4; ++k) h1_[k] = __builtin_convertvector(v_[k], f16x2_t);
#pragma unroll
    for (int k = 0; k < 4; ++k) r_[k] = (f32x2){__builtin_fmaf(x_[2 * k], sc, -(float)h1_[k][0]), __builtin_fmaf(x_[2 * k + 1], sc, -(float)h1_[k][1])};
#pragma unroll
    for (int k = 0; k < 4; ++k) h2_[k] = __builtin_convertvector(r_[k], f16x2_t);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = (int)threadIdx.x + (it + j) * NTHR;
      unsigned short* p = xh + lds_item + (it + j) * (32 * PS);
      if ((it + j + 1) * NTHR > K::NITEMS) p = idx < K::NITEMS ? p : pad16 - C;
      *(uint2*)(p) = make_uint2(__builtin_bit_cast(unsigned, h1_[2 * j]), __builtin_bit_cast(unsigned, h1_[2 * j + 1]));
      *(uint2*)(p + C) = make_uint2(__builtin_bit_cast(unsigned, h2_[2 * j]), __builtin_bit_cast(unsigned, h2_[2 * j + 1]));
    }
  };

  int tile = t_lo + rank;
  int cn = 0, cty = 0, ctx = 0;
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    tx_ += d_tx, ty_ += d_ty, n_ += d_n;
    if (tx_ >= tiles_x) tx_ -= tiles_x, ++ty_;
    if (ty_ >= tiles_y) ty_ -= tiles_y, ++n_;
  };

  // ---- centre operands: this lane's 8 pieces (row MT wave + mt, column li, channels 16 nt + 4 lg ..) of x (fetched one tile ahead: its
  // values are staged at the top of the tile) and of the epilogue's operands - gx so far (ACCUM), the GroupNorm input of the channel
  // sums (EPIAB), the activation output (EPIACT) - fetched at the top of their own tile, used after the input-gradient products
  const int yrow = a.wf * (C * 4);
  const int y_lane = ((wave * MT * a.wf + li) * C + lg * 4) * 4;
  float4 cxw[NPIECE], cy[ACCUM ? NPIECE : 1], cab[EPIAB && XSRC != 1 ? NPIECE : 1], cact[EPIACT && XSRC != 2 ? NPIECE : 1];
  const float* wx_base = XSRC == 1 ? a.ab_x : (XSRC == 2 ? a.ab_act_y : fa_.wx);
  auto centre_off = [&](int ty, int tx, unsigned (&off)[MT]) {
#ifdef FB_KO_LOAD
    ty = 1, tx = 1;
#endif
    const int vy0 = ty * FB_TR + wave * MT, vx0 = tx * FB_TC + li;
    const int t0 = (ty * FB_TR * a.wf + tx * FB_TC) * (C * 4) + y_lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) off[mt] = (vx0 < a.wv && vy0 + mt < a.hv) ? (unsigned)(t0 + mt * yrow) : BX_OOB;
  };
  auto x_issue = [&](int n, int ty, int tx, bool live, int i0 = 0, int i1 = NPIECE) __attribute__((always_inline)) {
    unsigned off[MT];
    centre_off(ty, tx, off);
    const long sb = (long)n * a.hf * a.wf * C;
    const unsigned bytes = live ? y_bytes : 0u;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i)
      if (i >= i0 && i < i1)
        cxw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(wx_base + sb, bytes), off[i / NT] + (i % NT) * 64, 0, 0));
  };
  auto epi_issue = [&](int n, const unsigned (&off)[MT]) {
    const long sb = (long)n * a.hf * a.wf * C;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const unsigned o = off[i / NT] + (i % NT) * 64;
      if (ACCUM) cy[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.y + sb, y_bytes), o, 0, 0));
      if (EPIAB && XSRC != 1) cab[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.ab_x + sb, y_bytes), o, 0, 0));
      if (EPIACT && XSRC != 2) cact[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.ab_act_y + sb, y_bytes), o, 0, 0));
    }
  };

  Pf pfc = pf_make(0, 0, 0, false);   // the current tile's halo (its items are in `pre` when an iteration starts)
  if (tile < t_hi) {
    ctx = tile % tiles_x, cty = (tile / tiles_x) % tiles_y, cn = tile / (tiles_x * tiles_y);
    pfc = pf_make(cn, cty, ctx, true);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) pf_issue(pfc, it);
    x_issue(cn, cty, ctx, true);
  }
  FB_T(10)

  // ---- weights: OIHW fp32 -> scaled fp16 planes in fragment order (conv_f16x2_kernel's prologue with 4 waves; the fp32 copy sits in
  // the x tile, which is first written after the first barrier of the tile loop)
  int sw_e = 0;
  {
    float* ws = (float*)xt;
    float* wmx = (float*)(red + 4);
    const int row = a.w_i * 9;
    const unsigned wbytes = (unsigned)((a.w_o - 1) * a.w_rs + row) * 4u;
    float m = 0.f;
    {
      float v[8][5];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          const int r = wave + NW * rr, j = lane + 64 * jj;
          const unsigned off = (r < a.w_o && j < row) ? (unsigned)(r * a.w_rs + j) * 4u : BX_OOB;
          v[rr][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bx_rsrc(a.w, wbytes), off, 0, 0));
        }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          const int r = wave + NW * rr, j = lane + 64 * jj;
          if (r < a.w_o && j < row) ws[r * (row + 1) + j] = v[rr][jj];
          m = fmaxf(m, fabsf(v[rr][jj]));
        }
    }
    m = f2_wave_max(m);
    if (lane == 0) wmx[wave] = m;
    if (threadIdx.x == 0) *(unsigned*)(red + 3) = 0u;   // (ab_flush's arrival counter)
    __syncthreads();
    const float4 m0 = *(const float4*)(wmx);
    sw_e = f2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)));
    const float sw = __builtin_ldexpf(1.f, sw_e);
    for (int u = threadIdx.x; u < KS * 4 * C; u += NTHR) {
      const int co = u % C, g = (u / C) & 3, ks = u / (4 * C);
      unsigned pl[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = fb_weight(ws, row + 1, a.w_o, a.w_i, ks, g, 2 * j, co);
        const float v1 = fb_weight(ws, row + 1, a.w_o, a.w_i, ks, g, 2 * j + 1, co);
        f2_split_pair(v0 * sw, v1 * sw, pl[0][j], pl[1][j]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
        *(uint4*)(wl + (((ks * NP + p) * 4 + g) * C + co) * 8) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
    // (no barrier here: barrier A of the first tile separates the last read of `ws` from the first write of the x tile, barrier B
    //  publishes the weight planes)
  }
  FB_T(11)

  float mg_lane = 0.f;   // this lane's largest halo magnitude of the tile that comes next (formed one tile ahead, see prep_item)
  if (RIDE && tile < t_hi) {
    prep_cf(cn);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) prep_item(pfc, it, mg_lane);
  }
  f32x4 accw[K::NACC];   // dW: this wave's (ci half, co half), tap j
#pragma unroll
  for (int j = 0; j < K::NACC; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sA[NT][4], sB[NT][4];
  int ab_n = -1;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[nt][r] = sB[nt][r] = 0.f;
  // XGN: the affine map of the GroupNorm in front of the conv, for this lane's 8 channels, per sample
  float4 xg_sc[XGN ? NT : 1], xg_sh[XGN ? NT : 1];
  int xg_n = -1;
#pragma unroll
  for (int nt = 0; nt < (XGN ? NT : 1); ++nt) xg_sc[nt] = xg_sh[nt] = make_float4(0.f, 0.f, 0.f, 0.f);

  // EPIAB: a sample's channel sums leave the workgroup (conv_f16x2_kernel's ab_flush with 4 waves)
  auto ab_flush = [&]() {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float va = sA[nt][r], vb = sB[nt][r];
#define FB_ROW(ctrl)                                                                                \
  va += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(va), ctrl, 0xf, 0xf, true)); \
  vb += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(vb), ctrl, 0xf, 0xf, true));
        FB_ROW(0xB1) FB_ROW(0x4E) FB_ROW(0x124) FB_ROW(0x128)
#undef FB_ROW
        if (li == 0) {
          abw[wave * 2 * C + nt * 16 + lg * 4 + r] = va;
          abw[wave * 2 * C + C + nt * 16 + lg * 4 + r] = vb;
        }
        sA[nt][r] = 0.f;
        sB[nt][r] = 0.f;
      }
    unsigned arrived = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) arrived = __hip_atomic_fetch_add((unsigned*)(red + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (arrived == NW - 1) {
      if (lane < 2 * C) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) t += (double)abw[wv * 2 * C + lane];
        a.ab_out[((long)ab_n * a.ab_slots + blockIdx.x) * (2 * C) + lane] = t;
      }
      if (lane == 0) *(unsigned*)(red + 3) = 0u;
    }
  };

  const int xa_lane = (wave * MT * IC + li) * PS + lg * 8;
  constexpr int PA[3] = {1, 0, 0};
  constexpr int PB[3] = {0, 1, 0};

  // ---- one tile: TOP (final values + maxima | barrier A | exponents | halo and x tile split and staged, next tile requested |
  // barrier B), D (the input gradient's 216 products per wave), W (its epilogue + this wave's 216 dW products over the WHOLE tile).
  // The dW exponent S is workgroup-uniform; when it has to move (rare) every wave leaves the loop in front of the staging, adds its
  // accumulators to the slab, and the pass restarts from zero with this tile setting S.
  unsigned cur_off[MT];
  float gmax = 0.f, xmax = 0.f;
  int sx_e = 0, ex_e = 0, S_w = 120, parity = 0;
  auto xval = [&](int i) -> float4 {   // the x value the products see: GroupNorm applied (XGN), pixels past the map zero
    float4 v = cxw[i];
    if (XGN) {
      const int nt = i % NT;
      const bool ok = cur_off[i / NT] != BX_OOB;
      v.x = v.x * xg_sc[nt].x + xg_sh[nt].x, v.y = v.y * xg_sc[nt].y + xg_sh[nt].y;
      v.z = v.z * xg_sc[nt].z + xg_sh[nt].z, v.w = v.w * xg_sc[nt].w + xg_sh[nt].w;
      v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return v;
  };
  auto top_a = [&]() __attribute__((always_inline)) {   // up to barrier A and the exponents (not repeated when a pass restarts at this tile)
    FB_T(0)
    centre_off(cty, ctx, cur_off);
    if (EPIAB && cn != ab_n) {   // (two flushes are always separated by a tile's barriers)
      if (ab_n >= 0) ab_flush();
      ab_n = cn;
    }
    if (XGN && cn != xg_n) {
      xg_n = cn;
      float mean, rstd;
      gn_moments(fa_.wx_gn_stats, cn, (double)a.hf * a.wf * C, fa_.wx_gn_eps, &mean, &rstd);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 g_ = *(const float4*)(fa_.wx_gn_gamma + nt * 16 + lg * 4), b_ = *(const float4*)(fa_.wx_gn_beta + nt * 16 + lg * 4);
        xg_sc[nt] = make_float4(rstd * g_.x, rstd * g_.y, rstd * g_.z, rstd * g_.w);
        xg_sh[nt] = make_float4(b_.x - xg_sc[nt].x * mean, b_.y - xg_sc[nt].y * mean, b_.z - xg_sc[nt].z * mean, b_.w - xg_sc[nt].w * mean);
      }
    }
    if (!RIDE) {
      prep_cf(cn);
      mg_lane = 0.f;
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) prep_item(pfc, it, mg_lane);
    }
    const float mg = f2_wave_max(mg_lane);
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const float4 v = xval(i);
      mx = __builtin_fmaxf(__builtin_fmaxf(mx, fabsf(v.x)), fabsf(v.y));
      mx = __builtin_fmaxf(__builtin_fmaxf(mx, fabsf(v.z)), fabsf(v.w));
    }
    mx = f2_wave_max(mx);
    if (lane == 0) {
      mxs[parity * 2 * NW + wave] = mg;
      mxs[parity * 2 * NW + NW + wave] = mx;
    }
    FB_T(1)
    // barrier A: every wave has finished the previous tile (halo and x tile may be overwritten), the maxima are visible
    __syncthreads();
    FB_T(2)
    const float4 m0 = *(const float4*)(mxs + parity * 2 * NW), m1 = *(const float4*)(mxs + parity * 2 * NW + NW);
    parity ^= 1;
    gmax = fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w));
    xmax = fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w));
    sx_e = f2_scale_exp(gmax);
    ex_e = f2_scale_exp(xmax);
  };
  f32x4 acc[MT][NT];
  auto rest = [&]() __attribute__((always_inline)) {
    // ---- staging: the halo with its own per-tile scale, x with 2^(S - sx_e) (a term of dW carries 2^S; where the halo or the x tile
    // is all zero the exponent does not matter)
#ifndef FB_KO_STAGE   // (diagnostic: nothing is staged - what the phase costs; WRONG results)
    {
      const float sc = __builtin_ldexpf(1.f, sx_e);
      // (one wave per SIMD: nothing hides the latency of a dependent vector instruction, and the compiler keeps source order - the
      //  split's chain mul -> cvt -> fma -> cvt of TWO items is interleaved by hand: 2.7 k -> cycles per tile for the 19 items)
#pragma unroll
      for (int it = 0; it + 1 < NLOAD; it += 2) stage_item2(it, sc);
      if (NLOAD & 1) stage_item(NLOAD - 1, sc);
      FB_T(8)
      const int es = S_w - sx_e < ex_e ? S_w - sx_e : ex_e;
      const float scx = __builtin_ldexpf(1.f, es);
#pragma unroll
      for (int i = 0; i < NPIECE; i += FB_XGRP) {   // (FB_XGRP pieces at a time, stage by stage: see stage_item2)
        float x_[4 * FB_XGRP];
#pragma unroll
        for (int j = 0; j < FB_XGRP; ++j) {
          const float4 va = xval(i + j);
          x_[4 * j] = va.x, x_[4 * j + 1] = va.y, x_[4 * j + 2] = va.z, x_[4 * j + 3] = va.w;
        }
        f32x2 v_[2 * FB_XGRP], r_[2 * FB_XGRP];
        f16x2_t h1_[2 * FB_XGRP], h2_[2 * FB_XGRP];
#pragma unroll
        for (int k = 0; k < 2 * FB_XGRP; ++k) v_[k] = (f32x2){x_[2 * k] * scx, x_[2 * k + 1] * scx};
#pragma unroll
        for (int k = 0; k < 2 * FB_XGRP; ++k) h1_[k] = __builtin_convertvector(v_[k], f16x2_t);
#pragma unroll
        for (int k = 0; k < 2 * FB_XGRP; ++k) r_[k] = (f32x2){__builtin_fmaf(x_[2 * k], scx, -(float)h1_[k][0]), __builtin_fmaf(x_[2 * k + 1], scx, -(float)h1_[k][1])};
#pragma unroll
        for (int k = 0; k < 2 * FB_XGRP; ++k) h2_[k] = __builtin_convertvector(r_[k], f16x2_t);
#pragma unroll
        for (int j = 0; j < FB_XGRP; ++j) {
          unsigned short* p = xt + ((wave * MT + (i + j) / NT) * FB_TC + li) * PS + ((i + j) % NT) * 16 + lg * 4;
          *(uint2*)(p) = make_uint2(__builtin_bit_cast(unsigned, h1_[2 * j]), __builtin_bit_cast(unsigned, h1_[2 * j + 1]));
          *(uint2*)(p + C) = make_uint2(__builtin_bit_cast(unsigned, h2_[2 * j]), __builtin_bit_cast(unsigned, h2_[2 * j + 1]));
        }
      }
    }
#endif
    FB_T(9)
    // this tile's epilogue operands are requested now (used after D); the next tile's halo and x strip ride in D's first k-steps (a
    // wave issues one 1-KB load per ~16 cycles at best: 19 - 30 of them in a row cost ~2 k cycles per tile in front of barrier B)
    int n1 = cn, ty1 = cty, tx1 = ctx;
    advance(n1, ty1, tx1);
    const Pf pfn = pf_make(n1, ty1, tx1, tile + per < t_hi);
    epi_issue(cn, cur_off);
    if (RIDE) prep_cf(n1);   // (the current tile's items are final: the coefficients may move on to the next tile's sample)
    float mg_next = 0.f;
    FB_T(3)
    // barrier B: halo, x tile (and, first tile, the weight planes) are complete
    __syncthreads();
    FB_T(4)

    // ---------------- input gradient: 9 taps x (4 rows x 2 channel blocks) x 3 products
    {
      // row fragments: ONE set R[halo row 0 .. MT + 1][plane] for the current kx.  Step (kx, ky) multiplies rows ky .. ky + MT - 1:
      // during (kx, 2) rows 0, 1 are dead and take kx + 1's; step (kx + 1, 0) fetches rows 2 .. MT + 1 first and multiplies rows 0, 1
      // (its mt = 0, 1) while they land.  Weight fragments: two sets, the next k-step's fetched under this one's products.
      s16x8 R[MT + 2][NP];
      s16x8 fw[2][NP][NT];
      auto load_rows = [&](int kx, int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MT + 2; ++j)
          if (j >= j0 && j < j1) {
#pragma unroll
            for (int p = 0; p < NP; ++p) R[j][p] = *(const s16x8*)(xh + xa_lane + (j * IC + kx) * PS + p * C);
          }
      };
      auto load_w = [&](int ks, s16x8 (&B)[NP][NT]) __attribute__((always_inline)) {
        const int wt = (ks % 3) * 3 + ks / 3;   // the weights are packed tap-major (ky * 3 + kx); the loop walks kx outer
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) B[p][nt] = *(const s16x8*)(wl + (((wt * NP + p) * 4 + lg) * C + nt * 16 + li) * 8);
      };
      load_rows(0, 0, MT + 2);
      load_w(0, fw[0]);
      fb_static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int kx = ks / 3, ky = ks % 3, b = ks & 1;
        if (ks + 1 < KS) load_w(ks + 1, fw[b ^ 1]);
        if (ky == 0 && kx > 0) load_rows(kx, 2, MT + 2);
        if (ky == 2 && kx < 2) load_rows(kx + 1, 0, 2);   // (this step reads rows 2 .. MT + 1: rows 0, 1 are dead)
        // the next tile's loads that ride in this k-step (their registers were emptied by the staging above)
#pragma unroll
        for (int it = 0; it < NLOAD; ++it)
          if (it * 6 / NLOAD == ks) pf_issue(pfn, it);   // (spread over 3 or 9 k-steps, or the x loads moved into W: no measurable difference)
#ifndef FB_XKS
#define FB_XKS 6   // the k-steps of D (FB_XKS, FB_XKS + 1) in which the next tile's x pieces are requested
#endif
        if (!XSH && ks >= FB_XKS && ks < FB_XKS + 2)
          x_issue(n1, ty1, tx1, tile + per < t_hi, (ks - FB_XKS) * (NPIECE / 2), (ks - FB_XKS + 1) * (NPIECE / 2));
        auto mm = [&](int mt0, int mt1) __attribute__((always_inline)) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              if (mt >= mt0 && mt < mt1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                      __builtin_bit_cast(f16x8_t, fw[b][PB[q]][nt]), __builtin_bit_cast(f16x8_t, R[ky + mt][PA[q]]),
                      (ks == 0 && q == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt][nt], 0, 0, 0);   // (the first product starts from a zero literal)
              }
        };
#ifdef FB_KO_D   // (diagnostic: no input-gradient products)
        if (false)
#endif
        if (ky == 0 && kx > 0) {
          mm(0, 2);
          __builtin_amdgcn_sched_barrier(0);
          mm(2, MT);
        } else {
          mm(0, MT);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    FB_T(5)

    // ---------------- epilogue of the input gradient (conv_f16x2_kernel's arithmetic, undeferred): piece i rides under the dW products
    const float* cur_y = a.y + (long)cn * a.hf * a.wf * C;
    const float desc = __builtin_ldexpf(1.f, -(sx_e + sw_e));
    auto epi_piece = [&](int i) __attribute__((always_inline)) {
      const int mt = i / NT, nt = i % NT;
      const float livef = cur_off[mt] != BX_OOB ? 1.f : 0.f;
      f32x4 o = acc[mt][nt] * desc;
      if (ACCUM) {
        const float4 q = cy[ACCUM ? i : 0];
        o += (f32x4){q.x, q.y, q.z, q.w};
      }
      if (EPIACT) {
        const float4 q = XSRC == 2 ? cxw[i] : cact[EPIACT && XSRC != 2 ? i : 0];
        o *= (f32x4){act_grad_from_out(q.x, EPIACT), act_grad_from_out(q.y, EPIACT), act_grad_from_out(q.z, EPIACT),
                     act_grad_from_out(q.w, EPIACT)};
      }
      const u32x4 ov = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
#ifdef FB_KO_STORE   // (diagnostic: the stores are dropped)
      __builtin_amdgcn_raw_buffer_store_b128(ov, bx_rsrc(cur_y, y_bytes), (cur_off[mt] + nt * 64) | BX_OOB, 0, 0);
#else
      __builtin_amdgcn_raw_buffer_store_b128(ov, bx_rsrc(cur_y, y_bytes), cur_off[mt] + nt * 64, 0, 0);
#endif
      if (EPIAB) {
        const float4 xv = XSRC == 1 ? cxw[i] : cab[EPIAB && XSRC != 1 ? i : 0];
        const float g0 = o[0] * livef, g1 = o[1] * livef, g2 = o[2] * livef, g3 = o[3] * livef;
        sA[nt][0] += g0, sA[nt][1] += g1, sA[nt][2] += g2, sA[nt][3] += g3;
        sB[nt][0] = __builtin_fmaf(g0, xv.x, sB[nt][0]), sB[nt][1] = __builtin_fmaf(g1, xv.y, sB[nt][1]);
        sB[nt][2] = __builtin_fmaf(g2, xv.z, sB[nt][2]), sB[nt][3] = __builtin_fmaf(g3, xv.w, sB[nt][3]);
      }
    };

    // ---------------- weight gradient: this wave's 9 accumulator tiles (tap j; ci half ah, co half bh) over ALL 256 pixels of the
    // tile: 8 k-steps of 32 pixels (tile rows 2 ks, 2 ks + 1), per k-step two x^T fragments (planes) against the gy fragments of the
    // 9 tap shifts - centre pixel (r', c') meets the halo pixel (r' + 2 - ky, c' + 2 - kx)
    {
      // gy fragments by halo row pair: tap row ky of k-step ks reads halo rows (2 ks + 2 - ky, + 1) - the pair of (ks, ky = 2) is the
      // pair of (ks - 1, ky = 0): fetched once, used twice (a k-step fetches two new pairs, not three: the phase is bound by the LDS
      // bandwidth of the transposing reads, 128 B per clock: 314 -> 250 reads per wave and tile).  Order within a k-step: ky = 2, 0, 1.
      s16x8 fx[2][NP];        // x^T: [k-step parity][plane]
      s16x8 G0[2][3][NP];     // even pairs (ky = 0 of k-step ks, ky = 2 of ks + 1): [ks parity][kx][plane]
      s16x8 G1[3][NP];        // odd pairs (ky = 1)
      auto load_x = [&](int ks, s16x8 (&F)[NP]) __attribute__((always_inline)) {
        const unsigned short* xq = xt + (2 * ks * FB_TC + 4 * lg + tq) * PS + ah * 16 + tp * 4;
#pragma unroll
        for (int p = 0; p < NP; ++p) F[p] = fb_tr_read8(xq + p * C, xq + FB_TC * PS + p * C);
      };
      // KX1: the kx = 1 fragments are assembled from the kx = 0 and kx = 2 ones (a one-pixel shift of the same rows: two
      // v_alignbit_b32 per 4 pixels) instead of being read - 20 of 28 transposing reads per k-step.  Same-box A/B per form
      // (profiles/r6_bwd_fused.md): plain 141.5 -> 133.7 us, plain + act' +-0, the GroupNorm-backward forms with channel sums
      // 149 -> 156 us (their W phase is short of vector issue slots, not of LDS bandwidth): on for the plain forms only.
#ifndef FB_KX1_ALIGN
#define FB_KX1_ALIGN (!INCOEF && INACT == 0)
#endif
      auto load_pair = [&](int row0, s16x8 (&G)[3][NP]) __attribute__((always_inline)) {   // halo rows row0, row0 + 1, the three kx shifts
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if (FB_KX1_ALIGN && kx == 1) continue;
          const unsigned short* gq = xh + (row0 * IC + (4 * lg + tq) + 2 - kx) * PS + bh * 16 + tp * 4;
#pragma unroll
          for (int p = 0; p < NP; ++p) G[kx][p] = fb_tr_read8(gq + p * C, gq + IC * PS + p * C);
        }
        if (FB_KX1_ALIGN) {
          // a lane's 4 + 4 values of a fragment are pixels q .. q + 3 of rows row0, row0 + 1 (q = 4 lg + 2 - kx): kx = 2 holds
          // (p0 p1 | p2 p3), kx = 0 holds (p2 p3 | p4 p5), kx = 1 is (p1 p2 | p3 p4)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const u32x4 a = __builtin_bit_cast(u32x4, G[2][p]), b = __builtin_bit_cast(u32x4, G[0][p]);
            u32x4 m;
            m[0] = __builtin_amdgcn_alignbit(a[1], a[0], 16);
            m[1] = __builtin_amdgcn_alignbit(b[1], b[0], 16);
            m[2] = __builtin_amdgcn_alignbit(a[3], a[2], 16);
            m[3] = __builtin_amdgcn_alignbit(b[3], b[2], 16);
            G[1][p] = __builtin_bit_cast(s16x8, m);
          }
        }
      };
      auto mm3 = [&](int ks, int ky, s16x8 (&G)[3][NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            accw[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fx[ks & 1][PA[q]]),
                                                                       __builtin_bit_cast(f16x8_t, G[kx][PB[q]]), accw[ky * 3 + kx], 0, 0, 0);
      };
      load_x(0, fx[0]);
      load_pair(0, G0[1]);   // (k-step 0, ky = 2)
      load_pair(2, G0[0]);   // (k-step 0, ky = 0)
#ifdef FB_KO_W   // (diagnostic: no dW products / reads; the epilogue and the riding work stay)
#define FB_MM3(a_, b_, c_)
#define FB_LP(a_, b_)
#define FB_LX(a_, b_)
#else
#define FB_MM3(a_, b_, c_) mm3(a_, b_, c_)
#define FB_LP(a_, b_) load_pair(a_, b_)
#define FB_LX(a_, b_) load_x(a_, b_)
#endif
      fb_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value;
        // ky = 2 (the pair fetched for ky = 0 of the previous k-step); the odd pair of this k-step is fetched under it
        FB_LP(2 * ks + 1, G1);
        if (ks < NPIECE / 2) epi_piece(2 * ks);       // (the input gradient's epilogue rides in the first k-steps: its stores leave early)
        FB_MM3(ks, 2, G0[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // ky = 0; the next k-step's x^T fragments are fetched under it
        if (ks + 1 < 8) FB_LX(ks + 1, fx[(ks + 1) & 1]);
        if (ks < NPIECE / 2) epi_piece(2 * ks + 1);
        FB_MM3(ks, 0, G0[ks & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // ky = 1; the next k-step's even pair is fetched under it (into the set ky = 2 has just left)
        if (ks + 1 < 8) FB_LP(2 * ks + 4, G0[(ks + 1) & 1]);
        if (XSH && ks < NPIECE / 2) x_issue(n1, ty1, tx1, tile + per < t_hi, 2 * ks, 2 * ks + 2);   // (behind the epilogue's pieces 2 ks, 2 ks + 1)
#pragma unroll
        for (int it = 0; it < NLOAD; ++it)
          if (RIDE && it * 8 / NLOAD == ks) prep_item(pfn, it, mg_next);   // the next tile's halo items (requested in D) become final
        FB_MM3(ks, 1, G1);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    FB_T(6)
    // the next tile becomes the current one
    cn = n1, cty = ty1, ctx = tx1;
    tile += per;
    pfc = pfn;
    if (RIDE) mg_lane = mg_next;
  };

#ifdef FB_STAGGER   // (diagnostic: every second workgroup starts its tile loop FB_STAGGER x 64 cycles late - are the workgroups' phases, all
                    //  requesting their next tile at the same time, what the loads wait for?  Measured with 60 and 110: 135 - 146 us either way, same box)
  if (rank & 1) __builtin_amdgcn_s_sleep(FB_STAGGER);
#endif
  int flushed = 0;
  bool resume = false;
  float* out = fa_.part + (long)blockIdx.x * (9 * C * C);
  auto slab_write = [&](bool add) {   // this wave's 9 tiles leave for the workgroup's slab (other waves own the other elements)
    const float dsc = __builtin_ldexpf(1.f, -S_w);
#pragma unroll
    for (int j = 0; j < K::NACC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = out + ((j * 2 + ah) * 16 + lg * 4 + r) * C + bh * 16 + li;
        const float v = accw[j][r] * dsc;
        *p = add ? *p + v : v;
      }
  };
  for (;;) {   // one pass per dW exponent: almost always exactly one
#pragma unroll
    for (int j = 0; j < K::NACC; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    S_w = 120;
    bool s_set = false, need = false;
    while (tile < t_hi) {
      if (!resume) top_a();
      resume = false;
      const bool adds = xmax > 0.f && gmax > 0.f;   // (workgroup-uniform) this tile has something to add to dW
      if (s_set && adds && ex_e + sx_e < S_w) {     // (rare) a larger product magnitude than the exponent allows: the accumulators leave first
        need = true;
        break;
      }
      if (!s_set && adds) {   // the first tile with something to add sets the exponent, FB_SMARGIN bits of headroom
        S_w = ex_e + sx_e - FB_SMARGIN;
        s_set = true;
      }
      rest();
    }
    if (!need) break;
    slab_write(flushed != 0);
    flushed = 1;
    resume = true;
  }

  FB_T(7)
  // ---------------- the workgroup's results leave: dW slab, channel sums, bias partials
  slab_write(flushed != 0);
  if (EPIAB) {
    __syncthreads();   // (a flush inside the last iteration and the final one must not overlap: see conv_f16x2_kernel)
    if (ab_n >= 0) ab_flush();
  }
  if (fa_.bpart) {
    __syncthreads();   // (every wave is done with the halo: the scratch below aliases it)
    float* bred = (float*)xh;
    const int vv = threadIdx.x % CV, row = threadIdx.x / CV;
    bred[row * C + vv * 4 + 0] = bsum.x;
    bred[row * C + vv * 4 + 1] = bsum.y;
    bred[row * C + vv * 4 + 2] = bsum.z;
    bred[row * C + vv * 4 + 3] = bsum.w;
    __syncthreads();
    if (threadIdx.x < C) {
      float sum = 0.f;
      for (int r = 0; r < NTHR / CV; ++r) sum += bred[r * C + threadIdx.x];
      fa_.bpart[(long)blockIdx.x * C + threadIdx.x] = sum;
    }
  }
#ifdef FB_STAMP
  if (lane == 0 && blockIdx.x < 256)
    for (int k = 0; k < 12; ++k) fb_stamps[(blockIdx.x * 4 + wave) * 12 + k] = st_[k];
#endif
}

// Launch: hipErrorInvalidValue when no instance exists for the combination (the caller keeps the two launches).
hipError_t dis_fb_launch(const FbArgs& f, int inact, bool xgn, int xsrc, long grid, hipStream_t stream) {
  using K = FbCfg;
  const ConvArgs& a = f.c;
  static bool attr_set[16] = {};
  auto launch = [&](auto kern, int slot) -> hipError_t {
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG("conv_bwd_fused_kernel<32>");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(K::NTHR), K::LDS_BYTES, stream, f);
    return hipSuccess;
  };
  constexpr int S = DIS_ACT_SELU;
#ifdef FB_ONLY   // (diagnostic builds: one instance, e.g. -DFB_ONLY=0,false,false,false,0,0,false,false - resource reports, ISA studies)
  return launch(conv_bwd_fused_kernel<FB_ONLY>, 0);
#endif
  if (inact != 0 && inact != S) return hipErrorInvalidValue;
  const bool coef = a.gnb_coef != nullptr, gst = a.gnb_out != nullptr, ab = a.ab_out != nullptr, epiact = a.ab_act_y != nullptr;
  if (!coef) {
    // plain operand (gy itself, or gy act'(y)): no epilogue forms
    if (ab || gst || xgn || xsrc) return hipErrorInvalidValue;
    if (a.accum) return inact ? launch(conv_bwd_fused_kernel<S, false, true, false, 0, 0, false, false>, 0)
                              : launch(conv_bwd_fused_kernel<0, false, true, false, 0, 0, false, false>, 1);
    return inact ? launch(conv_bwd_fused_kernel<S, false, false, false, 0, 0, false, false>, 2)
                 : launch(conv_bwd_fused_kernel<0, false, false, false, 0, 0, false, false>, 3);
  }
  if (ab && !a.accum && !epiact && xsrc == 1 && xgn) {   // conv2d_gn_in: the GroupNorm input of the sums is the conv's input
    if (gst) return inact ? hipErrorInvalidValue : launch(conv_bwd_fused_kernel<0, true, false, true, 0, 1, true, true>, 4);
    return inact ? launch(conv_bwd_fused_kernel<S, true, false, true, 0, 1, true, false>, 5)
                 : launch(conv_bwd_fused_kernel<0, true, false, true, 0, 1, true, false>, 6);
  }
  if (gst || xgn) return hipErrorInvalidValue;
  if (ab && a.accum && epiact && xsrc == 2 && inact == S)   // ResNetBlock chain: x = SELU(GroupNorm(x2) + res) is the conv's input
    return launch(conv_bwd_fused_kernel<S, true, true, true, S, 2, false, false>, 7);
  if (ab && a.accum && !epiact && xsrc == 0 && inact == S)  // two-consumer GroupNorm output (Block2D3D conv1_1)
    return launch(conv_bwd_fused_kernel<S, true, true, true, 0, 0, false, false>, 8);
  if (!ab && a.accum && xsrc == 0 && inact == S) return launch(conv_bwd_fused_kernel<S, true, true, false, 0, 0, false, false>, 9);
  if (!ab && !a.accum && xsrc == 0)
    return inact ? launch(conv_bwd_fused_kernel<S, true, false, false, 0, 0, false, false>, 10)
                 : launch(conv_bwd_fused_kernel<0, true, false, false, 0, 0, false, false>, 11);
  return hipErrorInvalidValue;
}
