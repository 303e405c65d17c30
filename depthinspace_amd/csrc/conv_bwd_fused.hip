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
// Shape of the kernel: 4 waves of 64 lanes, ONE per SIMD, 512 registers each - the 36 accumulator tiles of dW (144 registers) need
// them; with 8 waves (256 registers each) the input-gradient kernel's INCOEF instances alone sit at 236 - 256.  A wave owns 4 rows of
// the 16 x 16 tile: its input-gradient accumulators (4 x 2 tiles) and - for dW - those 64 pixels as TWO k-steps of 32 pixels against
// all 36 (tap, ci half, co half) tiles: no cross-wave reduction until the workgroup's last tile.  The x values of a wave's pixels live
// in a wave-private LDS strip (one k-step at a time, [pixel][plane][channel], read back with ds_read_b64_tr_b16): no barrier for them.
// dW scales: the gy halo carries the input gradient's per-tile scale 2^sg(t); x is split with 2^(S - sg(t)), S a per-wave exponent set by
// the first tile that has something to add (FB_SMARGIN bits of headroom), so that every term of the sum carries 2^S.  A later tile
// whose product magnitude exceeds the headroom makes the wave LEAVE the tile loop: the accumulators go to a spill slab in memory
// (scaled by 2^-S), a new pass starts from zero accumulators with a new S (RESUME copy of D: the tile's products are formed again).
// Status (round 6): parity-green in all forms, NOT faster than the two launches it replaces on MI355X - the register budget
// (profiles/r6_bwd_fused.md); the step uses it only with DIS_BWD_FUSED=1.
// LDS: weights 36.9 KB + two halo buffers 2 x 51.8 KB + x strips 4 x 5 KB + 1.2 KB = 162.2 KB of 160 KiB.
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
__device__ unsigned long long fb_stamps[256 * 4 * 8];
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
#ifndef FB_SMARGIN
#define FB_SMARGIN 6   // bits of headroom the dW exponent keeps when it is (re)set: a later tile may be 2^6 larger before the accumulators move again
#endif
struct FbCfg {
  static constexpr int C = 32, IR = FB_TR + 2, IC = FB_TC + 2, CV = C / 4, NP = 2, PS = 80, NT = 2, KS = 9;
  static constexpr int NW = 4, NTHR = 64 * NW, MT = FB_TR / NW;
  static constexpr int W_U16 = KS * NP * 4 * C * 8, X_U16 = IR * IC * PS, XW_U16 = 32 * PS;
  static constexpr int NITEMS = IR * IC * CV, NLOAD = (NITEMS + NTHR - 1) / NTHR, NPIECE = MT * NT;
  static constexpr int SMALL_U16 = 32 + 32 + NW * 2 * C * 2 + C + 8;   // red (8 doubles), mxs, abw [wave][2 C] floats, write pad (both planes of an idle item)
  static constexpr int LDS_BYTES = (W_U16 + 2 * X_U16 + NW * XW_U16 + SMALL_U16) * 2;
  static constexpr int NACC = 9 * 2 * 2;   // dW accumulator tiles per wave: (tap, ci half, co half)
};
static_assert(FbCfg::LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert((FbCfg::NW - 1) * FbCfg::NACC * 64 * 16 <= (FbCfg::W_U16 + 2 * FbCfg::X_U16) * 2, "end-of-kernel reduction aliases weights + halos");

// INACT / INCOEF / ACCUM / EPIAB / EPIACT: conv_f16x2_kernel's forms of the input gradient (see there).
// XSRC: where the weight gradient's x comes from - 0: a.wx, 1: c.ab_x (the GroupNorm input the channel sums are formed with IS the
// conv's input: conv2d_gn_in), 2: c.ab_act_y (the activation output the result is multiplied with IS the conv's input: ResNetBlock
// chains).  XGN: x is staged as GroupNorm(x).  GST: the staged values of the pixels a tile owns are stored to c.gnb_out as well (the
// other launches of a multi-source node read them).
template <int INACT, bool INCOEF, bool ACCUM, bool EPIAB, int EPIACT, int XSRC, bool XGN, bool GST>
__global__ __launch_bounds__(256) void conv_bwd_fused_kernel(FbArgs fa_) {
  using K = FbCfg;
  const ConvArgs& a = fa_.c;
  constexpr int C = K::C, IC = K::IC, PS = K::PS, NT = K::NT, KS = K::KS, NLOAD = K::NLOAD, NPIECE = K::NPIECE, CV = K::CV, NP = K::NP;
  constexpr int MT = K::MT, NW = K::NW, NTHR = K::NTHR;
  constexpr bool IN2 = INACT != 0 || INCOEF;
  static_assert(EPIACT == 0 || EPIAB, "activation gradient at the output: only with the channel sums");
  static_assert(XSRC == 0 || (XSRC == 1 && EPIAB) || (XSRC == 2 && EPIACT), "shared x operand");
  static_assert(!GST || INCOEF, "gpre store: only where the operand is formed on load");
#ifdef FB_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* wl = smem16;
  unsigned short* xl = smem16 + K::W_U16;
  unsigned short* xw = smem16 + K::W_U16 + 2 * K::X_U16 + (threadIdx.x >> 6) * K::XW_U16;   // this wave's x strip
  unsigned short* small = smem16 + K::W_U16 + 2 * K::X_U16 + NW * K::XW_U16;
  double* red = (double*)small;
  float* mxs = (float*)(small + 32);      // [parity][wave] (2 x 4 floats)
  float* abw = (float*)(small + 64);      // EPIAB: [wave][2 C]
  unsigned short* pad16 = small + 64 + NW * 2 * C * 2 + C;   // (idle threads of the last round write pad16 - C and pad16)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4, tq = li >> 2, tp = li & 3;
  const int tiles_x = (a.wv + FB_TC - 1) / FB_TC, tiles_y = (a.hv + FB_TR - 1) / FB_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);
  const int d_tx = per % tiles_x, d_ty = (per / tiles_x) % tiles_y, d_n = per / (tiles_x * tiles_y);

  // ---- halo items of this thread (conv_f16x2_kernel's, 256 threads): item it = float4 vv of halo pixel pix = p0 + 32 it, p0 = thread / 8.
  // Nothing per item is kept in registers: row / column follow from p0 and compile-time constants (32 it = 18 A + B), the LDS address
  // is a constant offset from one base - 33 loop-invariant registers per thread less than tables of offsets (which the compiler
  // spilled to scratch, whose reloads then made every halo load of the loop synchronous: s_waitcnt vmcnt(0) all over the tap loop).
  float4 pre[NLOAD], pre2[IN2 ? NLOAD : 1];
  const int p0 = (int)threadIdx.x >> 3;
  const int vv4 = ((int)threadIdx.x & 7) * 16;   // byte offset of this thread's 4 channels within a pixel (the same for all its items)
  int p0v = p0;                                  // (re-blinded every iteration: keeps the per-item arithmetic out of loop-invariant registers)
  auto item_rc = [&](int it, int& r, int& c) __attribute__((always_inline)) {
    const int A = (32 * it) / IC, B = (32 * it) % IC;
    const int sft = p0v + B;
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
  // (1) before the barrier: final fp32 values of the next tile's items, this wave's largest magnitude into LDS
  auto prep = [&](const Pf& f, int n_cur, int parity) {
    if (INCOEF && n_cur != cf_n) {
      cf_n = n_cur;
      const float* cf = a.gnb_coef + (long)(n_cur < a.n ? n_cur : a.n - 1) * (C + 2);
      cf_k1 = *(const float4*)(cf + ((int)threadIdx.x % CV) * 4);
      cf_kx = cf[C];
      cf_k0 = cf[C + 1];
    }
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      float4 v = pre[it];
      if (INCOEF) {   // (gn_apply_coef_kernel's arithmetic, bit for bit; padding: g = q = 0 would give k0, which must not be staged)
        const float4 q = pre2[it];
        int r_, c_;
        item_rc(it, r_, c_);
        const int ix = f.ix0 + c_;
        const unsigned off = (unsigned)ix < (unsigned)a.win ? (unsigned)(f.off0 + (r_ * a.win + c_) * (C * 4) + vv4) : BX_OOB;
        const bool inside = off < f.bytes;
        v.x = __builtin_fmaf(v.x, cf_k1.x, __builtin_fmaf(q.x, cf_kx, cf_k0));
        v.y = __builtin_fmaf(v.y, cf_k1.y, __builtin_fmaf(q.y, cf_kx, cf_k0));
        v.z = __builtin_fmaf(v.z, cf_k1.z, __builtin_fmaf(q.z, cf_kx, cf_k0));
        v.w = __builtin_fmaf(v.w, cf_k1.w, __builtin_fmaf(q.w, cf_kx, cf_k0));
        if (INACT) {
          v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
          v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
        }
        v = inside ? v : make_float4(0.f, 0.f, 0.f, 0.f);
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
      {   // (bias gradient: the pixels this tile owns; a select, not a branch)
        const bool own = item_own(it);
        bsum.x += own ? v.x : 0.f, bsum.y += own ? v.y : 0.f, bsum.z += own ? v.z : 0.f, bsum.w += own ? v.w : 0.f;
      }
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.x)), fabsf(v.y));
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.z)), fabsf(v.w));
    }
    m = f2_wave_max(m);
    if (lane == 0) mxs[parity * NW + wave] = m;
  };
  auto tile_max = [&](int parity) -> float {
    const float4 m0 = *(const float4*)(mxs + parity * NW);
    return fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w));
  };
  const int lds_item = p0 * PS + ((int)threadIdx.x & 7) * 4;
  auto stage_item = [&](int it, float sc, unsigned short* xb) {
    const float4 v = pre[it];
    unsigned a1, a2, b1, b2;
    f2_split_pair_scaled(v.x, v.y, sc, a1, a2);
    f2_split_pair_scaled(v.z, v.w, sc, b1, b2);
    const int idx = (int)threadIdx.x + it * NTHR;
    unsigned short* p = xb + lds_item + it * (32 * PS);   // pixel p0 + 32 it, channels 4 vv ..: one base, constant offsets
    if ((it + 1) * NTHR > K::NITEMS) p = idx < K::NITEMS ? p : pad16 - C;
    *(uint2*)(p) = make_uint2(a1, b1);
    *(uint2*)(p + C) = make_uint2(a2, b2);
  };

  int tile = t_lo + rank;
  int cn = 0, cty = 0, ctx = 0;
  Pf pf0 = pf_make(0, 0, 0, false);
  if (tile < t_hi) {
    ctx = tile % tiles_x, cty = (tile / tiles_x) % tiles_y, cn = tile / (tiles_x * tiles_y);
    pf0 = pf_make(cn, cty, ctx, true);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) pf_issue(pf0, it);
  }
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    tx_ += d_tx, ty_ += d_ty, n_ += d_n;
    if (tx_ >= tiles_x) tx_ -= tiles_x, ++ty_;
    if (ty_ >= tiles_y) ty_ -= tiles_y, ++n_;
  };
  int n1 = cn, ty1 = cty, tx1 = ctx;   // the tile after the current one

  // ---- the centre operands of a tile: this lane's 8 pieces (row MT wave + mt, column li, channels 16 nt + 4 lg ..) of gx-so-far
  // (ACCUM), the GroupNorm input of the channel sums (EPIAB), the activation output (EPIACT) and x - fetched one tile ahead
  const int yrow = a.wf * (C * 4);
  const int y_lane = ((wave * MT * a.wf + li) * C + lg * 4) * 4;
  float4 cy[ACCUM ? NPIECE : 1], cab[EPIAB ? NPIECE : 1], cact[EPIACT ? NPIECE : 1], cxw[XSRC == 0 ? NPIECE : 1];
  auto centre_off = [&](int ty, int tx, unsigned (&off)[MT]) {
    const int vy0 = ty * FB_TR + wave * MT, vx0 = tx * FB_TC + li;
    const int t0 = (ty * FB_TR * a.wf + tx * FB_TC) * (C * 4) + y_lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) off[mt] = (vx0 < a.wv && vy0 + mt < a.hv) ? (unsigned)(t0 + mt * yrow) : BX_OOB;
  };
  auto centre_issue = [&](int n, int ty, int tx, bool live) {
    unsigned off[MT];
    centre_off(ty, tx, off);
    const long sb = (long)n * a.hf * a.wf * C;
    const unsigned bytes = live ? y_bytes : 0u;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const unsigned o = off[i / NT] + (i % NT) * 64;
      if (ACCUM) cy[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.y + sb, bytes), o, 0, 0));
      if (EPIAB) cab[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.ab_x + sb, bytes), o, 0, 0));
      if (EPIACT) cact[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.ab_act_y + sb, bytes), o, 0, 0));
      if (XSRC == 0) cxw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(fa_.wx + sb, bytes), o, 0, 0));
    }
  };

  // ---- weights: OIHW fp32 -> scaled fp16 planes in fragment order (conv_f16x2_kernel's prologue with 4 waves); the workgroup's first
  // tile is staged in between
  int sw_e = 0, parity = 0, buf = 0, sx_e = 0;
  float gmax = 0.f;   // largest magnitude of the current tile's halo (0: nothing to add to dW)
  {
    float* ws = (float*)(xl + K::X_U16);
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
    if (tile < t_hi) prep(pf0, cn, parity);
    __syncthreads();
    const float4 m0 = *(const float4*)(wmx);
    sw_e = f2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)));
    const float sw = __builtin_ldexpf(1.f, sw_e);
    if (tile < t_hi) {
      gmax = tile_max(parity);
      sx_e = f2_scale_exp(gmax);
      const float sc = __builtin_ldexpf(1.f, sx_e);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) stage_item(it, sc, xl);
      // (the order of a steady-state iteration - the next tile's halo loads, THEN the current tile's centre loads: the compiler's
      //  wait counts at the loop head are merged over both ways in, and with the other order they waited for everything in flight)
      advance(n1, ty1, tx1);
      pf0 = pf_make(n1, ty1, tx1, tile + per < t_hi);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) pf_issue(pf0, it);
    }
    parity ^= 1;
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
  }

  f32x4 accw[K::NACC];   // dW: tile (tap, ci half a, co half b) at index (tap * 2 + a) * 2 + b, this wave's pixels
#pragma unroll
  for (int j = 0; j < K::NACC; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int S_w = 120;   // running exponent of the dW terms (the clamp's upper end: the first real tile lowers it)
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

  auto xraw = [&](int i) -> float4 { return XSRC == 1 ? cab[EPIAB ? i : 0] : (XSRC == 2 ? cact[EPIACT ? i : 0] : cxw[XSRC == 0 ? i : 0]); };
  auto xval = [&](int i, const unsigned (&cur_off)[MT]) -> float4 {   // the value the products see: GroupNorm applied (XGN), pixels past the map zero
    float4 v = xraw(i);
    if (XGN) {
      const int nt = i % NT;
      const bool ok = cur_off[i / NT] != BX_OOB;
      v.x = v.x * xg_sc[nt].x + xg_sh[nt].x, v.y = v.y * xg_sc[nt].y + xg_sh[nt].y;
      v.z = v.z * xg_sc[nt].z + xg_sh[nt].z, v.w = v.w * xg_sc[nt].w + xg_sh[nt].w;
      v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return v;
  };
  // ---- a tile goes through two phases: D (the input gradient's products; the NEXT tile's halo is split and staged on the way) and W
  // (the input gradient's epilogue and the dW products of the same tile, against the halo D has just read).  The loop runs W(t), D(t + 1):
  // the dW accumulators are carried around a loop in which only matrix instructions touch them, so they stay in accumulation registers.
  // When the dW exponent has to move (a tile whose product magnitude exceeds every earlier one's by more than the headroom - rare), the
  // inner loop is LEFT, the accumulators go to this wave's spill slab in memory, and a new pass starts from zero accumulators.  (With the
  // rescale as a branch inside the loop the compiler kept all 144 accumulators in vector registers across it: ~290 register copies per
  // tile and spills to scratch, whose reloads made every halo load synchronous.)
  f32x4 acc[MT][NT];
  unsigned cur_off[MT];
  const unsigned short* xc = xl;
  int sx_n = 0, n2 = 0, ty2 = 0, tx2 = 0, ex_w = 0;
  float gmax_n = 0.f, xm = 0.f;
  Pf pfn = pf0;
  // RESUME (rare): the tile's D has run already, the wave left the loop in front of W to park its dW accumulators; only the products
  // are formed again (the halo buffer is untouched until the next D stages into the other one) - no barrier, no staging, no loads.
  auto dphase = [&](auto rsm) __attribute__((always_inline)) {
    constexpr bool RESUME = decltype(rsm)::value;
    float sc_n = 0.f;
    unsigned short* xn = xl;
    if constexpr (!RESUME) {
#if FB_BLIND
      asm volatile("" : "+v"(p0v));
#endif
      centre_off(cty, ctx, cur_off);
      // the NEXT tile's items (in flight since the previous D): final values, maxima
      FB_T(0)
      prep(pf0, n1, parity);
      // this tile's centre operands (x strip, gx so far, GroupNorm input, activation output): requested here, used from the end of D on -
      // a whole D phase to land.  (Requested in the previous W they were the youngest loads in flight at the loop head, where the
      // compiler's merged wait counts wait for everything: ~2.2 k cycles per tile, scripts/diag/fb_stamps.py.)
      centre_issue(cn, cty, ctx, true);
      FB_T(1)
      // ONE barrier per tile: every wave has finished reading the other halo buffer (previous tile: both products), this tile's buffer
      // is completely written, the maxima of the next tile are visible
      __syncthreads();
      FB_T(2)
      gmax_n = tile_max(parity);
      sx_n = f2_scale_exp(gmax_n);
      sc_n = __builtin_ldexpf(1.f, sx_n);
      parity ^= 1;
      xc = xl + buf * K::X_U16;
      xn = xl + (buf ^ 1) * K::X_U16;
      n2 = n1, ty2 = ty1, tx2 = tx1;
      advance(n2, ty2, tx2);
      pfn = pf_make(n2, ty2, tx2, tile + 2 * per < t_hi);
    }

    // ---------------- input gradient: 9 taps x (4 rows x 2 channel blocks) x 3 products; the next tile's items are split and written
    // to the other halo buffer on the way, each followed by the load that refills its registers with the tile after next
    if (!FB_ZLIT) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
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
            for (int p = 0; p < NP; ++p) R[j][p] = *(const s16x8*)(xc + xa_lane + (j * IC + kx) * PS + p * C);
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
        // the next tile's items that ride in this k-step: split, LDS write, refill with the tile after next
        if constexpr (!RESUME) {
#pragma unroll
          for (int it = 0; it < NLOAD; ++it)
            if (it * KS / NLOAD == ks) {
              stage_item(it, sc_n, xn);
              __builtin_amdgcn_sched_barrier(0);   // (the load below reuses the registers the item has just left: not before their last read)
              pf_issue(pfn, it);
            }
        }
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
                      (FB_ZLIT && ks == 0 && q == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt][nt], 0, 0, 0);   // (the first product starts from a zero literal)
              }
        };
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


    FB_T(3)
    // this wave's x strip of the tile (requested during the previous W): largest magnitude -> its exponent
    if constexpr (!RESUME) {
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
      xm = 0.f;
  #pragma unroll
      for (int i = 0; i < NPIECE; ++i) {
        const float4 v = xval(i, cur_off);
        xm = __builtin_fmaxf(__builtin_fmaxf(xm, fabsf(v.x)), fabsf(v.y));
        xm = __builtin_fmaxf(__builtin_fmaxf(xm, fabsf(v.z)), fabsf(v.w));
      }
      xm = f2_wave_max(xm);
      ex_w = f2_scale_exp(xm);
    }
  };
  auto wphase = [&]() __attribute__((always_inline)) {
    FB_T(4)
    const float* cur_y = a.y + (long)cn * a.hf * a.wf * C;
    if (EPIAB && cn != ab_n) {   // (two flushes are always separated by D's barrier)
      if (ab_n >= 0) ab_flush();
      ab_n = cn;
    }
    // ---------------- epilogue of the input gradient (conv_f16x2_kernel's arithmetic, undeferred): piece i rides under the dW products
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
        const float4 q = cact[EPIACT ? i : 0];
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
        const float4 xv = cab[EPIAB ? i : 0];
        const float g0 = o[0] * livef, g1 = o[1] * livef, g2 = o[2] * livef, g3 = o[3] * livef;
        sA[nt][0] += g0, sA[nt][1] += g1, sA[nt][2] += g2, sA[nt][3] += g3;
        sB[nt][0] = __builtin_fmaf(g0, xv.x, sB[nt][0]), sB[nt][1] = __builtin_fmaf(g1, xv.y, sB[nt][1]);
        sB[nt][2] = __builtin_fmaf(g2, xv.z, sB[nt][2]), sB[nt][3] = __builtin_fmaf(g3, xv.w, sB[nt][3]);
      }
    };


    // ---------------- weight gradient: this wave's 64 pixels (two k-steps of 32) against the halo in LDS
    {
      // (a term carries 2^(es + sx_e); where that is not 2^S_w - the halo or the strip is all zero - the term is zero anyway)
      const int es = S_w - sx_e < ex_w ? S_w - sx_e : ex_w;
      const float scx = __builtin_ldexpf(1.f, es);
      __builtin_amdgcn_sched_barrier(0);
      fb_static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float4 v = xval((2 * h + mi) * NT + nt, cur_off);
            unsigned a1, a2, b1, b2;
            f2_split_pair_scaled(v.x, v.y, scx, a1, a2);
            f2_split_pair_scaled(v.z, v.w, scx, b1, b2);
            unsigned short* p = xw + (mi * 16 + li) * PS + nt * 16 + lg * 4;
            *(uint2*)(p) = make_uint2(a1, b1);
            *(uint2*)(p + C) = make_uint2(a2, b2);
          }
        s16x8 fx[2][NP];   // x^T: [ci half][plane], k = this k-step's 32 pixels
        const unsigned short* xq = xw + (4 * lg + tq) * PS + tp * 4;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int p = 0; p < NP; ++p) fx[ah][p] = fb_tr_read8(xq + ah * 16 + p * C, xq + 16 * PS + ah * 16 + p * C);
        // units u = (tap, co half): two gy fragments (one per plane) against the four x fragments, 6 products; the next unit's
        // fragments are fetched under this unit's products
        s16x8 fg[3][NP];   // (three sets, fetched two units ahead: one unit = 6 products = ~100 cycles, less than an LDS round trip)
        auto load_g = [&](int u, s16x8 (&G)[NP]) __attribute__((always_inline)) {
          const int tap = u >> 1, bh = u & 1, ky = tap / 3, kx = tap % 3;
          // centre pixel (r', c') of this k-step meets the halo pixel (r' + 2 - ky, c' + 2 - kx)
          const unsigned short* gq = xc + ((MT * wave + 2 * h + 2 - ky) * IC + (4 * lg + tq) + 2 - kx) * PS + tp * 4 + bh * 16;
#pragma unroll
          for (int p = 0; p < NP; ++p) G[p] = fb_tr_read8(gq + p * C, gq + IC * PS + p * C);
        };
        load_g(0, fg[0]);
        load_g(1, fg[1]);
        fb_static_for<0, 18>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value;
          if (u + 2 < 18) load_g(u + 2, fg[(u + 2) % 3]);
          // what rides under the products: the input gradient's epilogue in the first k-step, the next tile's centre loads in the second
          if (h == 0 && u % 2 == 0 && u / 2 < NPIECE) epi_piece(u / 2);
#pragma unroll
          for (int ah = 0; ah < 2; ++ah)
#pragma unroll
            for (int q = 0; q < 3; ++q)
              accw[(u >> 1) * 4 + ah * 2 + (u & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                  __builtin_bit_cast(f16x8_t, fx[ah][PA[q]]), __builtin_bit_cast(f16x8_t, fg[u % 3][PB[q]]),
                  accw[(u >> 1) * 4 + ah * 2 + (u & 1)], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    }
    FB_T(5)
    // the next tile becomes the current one
    cn = n1, cty = ty1, ctx = tx1;
    n1 = n2, ty1 = ty2, tx1 = tx2;
    tile += per;
    sx_e = sx_n;
    gmax = gmax_n;
    buf ^= 1;
    pf0 = pfn;
  };
  int flushed = 0;
  bool resume = false;
  for (;;) {   // one pass per dW exponent: almost always exactly one
#pragma unroll
    for (int j = 0; j < K::NACC; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    S_w = 120;
    bool s_set = false, need = false;
    while (tile < t_hi) {
      if (resume) dphase(std::true_type{});
      else dphase(std::false_type{});
      resume = false;
      const bool adds = xm > 0.f && gmax > 0.f;   // (wave-uniform) this tile has something to add to dW
      if (s_set && adds && ex_w + sx_e < S_w) {   // (rare) a larger product magnitude than the exponent allows: the accumulators leave first
        need = true;
        break;
      }
      if (!s_set && adds) {   // the first tile with something to add sets the exponent, FB_SMARGIN bits of headroom
        S_w = ex_w + sx_e - FB_SMARGIN;
        s_set = true;
      }
      wphase();
    }
    if (!need) break;
    {   // the accumulators so far leave for this wave's spill slab (added to what an earlier pass left there)
      const float dsc = __builtin_ldexpf(1.f, -S_w);
      f32x4* sp = (f32x4*)(fa_.spill + ((long)blockIdx.x * NW + wave) * (K::NACC * 256)) + lane;
#pragma unroll
      for (int j = 0; j < K::NACC; ++j) {
        f32x4 v = accw[j] * dsc;
        if (flushed) v += sp[j * 64];
        sp[j * 64] = v;
      }
      flushed = 1;
      resume = true;
    }
  }

  FB_T(6)
  // ---------------- the workgroup's results leave: channel sums, bias partials, the dW slab
  if (EPIAB) {
    __syncthreads();   // (a flush inside the last iteration and the final one must not overlap: see conv_f16x2_kernel)
    if (ab_n >= 0) ab_flush();
  }
  __syncthreads();     // every wave is done with the weights and the halo buffers: the reductions below alias them
  {
    const float desc = __builtin_ldexpf(1.f, -S_w);
#pragma unroll
    for (int j = 0; j < K::NACC; ++j) accw[j] *= desc;
    if (flushed) {   // (rare: earlier passes of this wave, written by these very lanes)
      const f32x4* sp = (const f32x4*)(fa_.spill + ((long)blockIdx.x * NW + wave) * (K::NACC * 256)) + lane;
#pragma unroll
      for (int j = 0; j < K::NACC; ++j) accw[j] += sp[j * 64];
    }
    f32x4* rbuf = (f32x4*)smem16;   // [wave - 1][tile j][lane]
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < K::NACC; ++j) rbuf[((wave - 1) * K::NACC + j) * 64 + lane] = accw[j];
    }
    __syncthreads();
    if (wave == 0) {
      float* out = fa_.part + (long)blockIdx.x * (9 * C * C);
#pragma unroll
      for (int j = 0; j < K::NACC; ++j) {
        f32x4 s = accw[j];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) s += rbuf[((wv - 1) * K::NACC + j) * 64 + lane];
        const int mb = j >> 1, nb = j & 1;   // (mb = tap * 2 + ci half)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(mb * 16 + lg * 4 + r) * C + nb * 16 + li] = s[r];
      }
    }
  }
  if (fa_.bpart) {
    __syncthreads();
    float* bred = (float*)smem16;
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
  FB_T(7)
  if (lane == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) fb_stamps[(blockIdx.x * 4 + wave) * 8 + k] = st_[k];
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
