// fp32 convolution on the fp16 matrix cores with TWO-term operands ("f16x2"): the 3x3 stride-1 convolutions of FuseNet with 16 / 32
// channels on either side (forward, input gradient, weight gradient), the same launches conv2d.hip's bf16x3 kernels serve.
//
//   x * 2^s = h1 + h2,  h1 = fp16(x * 2^s), h2 = fp16(x * 2^s - h1)        (11 + 11 significant bits)
//   a * b  ~= a1 b1 + a1 b2 + a2 b1                                        (3 products on v_mfma_f32_16x16x32_f16, fp32 accumulate)
//
// Each fp16 x fp16 product is exact in fp32.  What is dropped - a2 b2 and the third-order remainders of the operands - is
// <= 2^-21 of |a b| (typically 2^-23): operands of 22 bits instead of fp32's 24, measured end to end at the level of the fp32
// summation-order noise of the CPU reference itself (scripts/diag/emul_f16x2.py).  Against the three-term
// bf16 split of conv2d.hip (6 products, >= 24 bits) this is HALF the matrix work, 2 operand planes instead of 3 in LDS and
// two thirds of the fragment reads.  fp16 has 5 exponent bits, so every operand block is scaled by a power of two first:
//   * weights: one scale per launch (largest |w| -> [2^14, 2^15)), found in the prologue that splits them into LDS;
//   * pixels: one scale per halo tile.  Every wave reduces the values it is about to stage to a maximum (DPP), the eight
//     maxima are exchanged through LDS across the barrier that already separates two tiles, and the tile is split with
//     2^(14 - exponent of its maximum).  Entries far below the tile's maximum lose RELATIVE precision (they reach the fp16
//     subnormals 2^-39 below it) but their ABSOLUTE error stays 2^-24 * 2^-15 of the maximum: nothing a dot product can see.
//   * the accumulators are multiplied by 2^-(sx + sw) (exact) when a finished tile is handed to the deferred epilogue; the
//     bias joins there.
// DIS_CONV_SPLIT=bf16x3 keeps every launch on the three-term kernels.
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>

#ifndef F2_VALU_PER_GAP
#define F2_VALU_PER_GAP 6
#endif
#ifndef F2_PF2_INRES
#define F2_PF2_INRES 0
#endif
#ifndef F2_PF2
#define F2_PF2 1   // two register sets of halo items where the registers allow it (see PF2 in conv_f16x2_kernel)
#endif
#define F2_TR 16
#define F2_TC 16

// 1: two-term fp16 split (default), 0: three-term bf16 split; DIS_CONV_SPLIT=bf16x3 / f16x2 sets the start value
static int g_f2_mode = -1;
bool dis_f2_enabled() {
  if (g_f2_mode < 0) g_f2_mode = (getenv("DIS_CONV_SPLIT") && getenv("DIS_CONV_SPLIT")[0] == 'b') ? 0 : 1;
  return g_f2_mode == 1;
}
extern "C" int dis_set_conv_split(int mode) {
  if (mode != 0 && mode != 1) return DIS_ERR_UNSUPPORTED;
  g_f2_mode = mode;
  return DIS_OK;
}
extern "C" int dis_get_conv_split(void) { return dis_f2_enabled() ? 1 : 0; }

template <int CIN, int COUT>
struct F2Cfg {
  static constexpr int IR = F2_TR + 2, IC = F2_TC + 2;  // halo tile
  static constexpr int CV = CIN / 4;                      // float4s per pixel
  static constexpr int NP = 2;                            // planes
  // LDS pixel stride (16-bit units): payload NP * CIN, padded to 2 (mod 4) sixteen-byte units - the stride at which the four
  // NON-contiguous 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS table) hit disjoint banks (see BxCfg::PS)
  static constexpr int PS = CIN == 32 ? 80 : 48;
  static constexpr int NT = COUT / 16;
  static constexpr int KS = CIN == 32 ? 9 : 5;  // k-steps: a tap (32 channels) or a pair of taps (16 + 16)
  static constexpr int W_U16 = KS * NP * 4 * COUT * 8;  // packed[kstep][plane][lg][co][8]
  static constexpr int X_U16 = IR * IC * PS;
  // + stats reduction (8 doubles) + wave maxima (2 x 8 floats) + per-wave channel sums of the EPIAB form (8 x 2 COUT floats)
  static constexpr int NXB = 2;  // halo buffers: tile t+1 is split and written while the matrix loop of tile t reads the other one
  static constexpr int LDS_BYTES = W_U16 * 2 + NXB * X_U16 * 2 + 64 + 64 + 8 * 2 * COUT * 4 + 16 + 2 * CIN;  // (+ a write pad)
  static constexpr int NITEMS = IR * IC * CV;
  static constexpr int NLOAD = (NITEMS + 511) / 512;
  static constexpr int NPIECE = 2 * NT;
#ifndef F2_LOAD_SCHED
#define F2_LOAD_SCHED 0
#endif
#ifndef F2_WAIT_MODE
#define F2_WAIT_MODE 0
#endif
  // k-step at which deferred-epilogue piece i is stored / halo item i is staged and its register refilled with the tile after next
  static constexpr int piece_ks(int i) { return KS == 9 ? (F2_LOAD_SCHED ? i + 3 : 2 * i + 1) : i + 1; }
  static constexpr int load_ks(int i) { return KS == 9 ? (F2_LOAD_SCHED ? i / 2 : 2 * (i / 2)) : i; }
  // vector-memory operations a wave issues AFTER its last halo load of a matrix loop (the stores of the pieces that ride at or
  // behind that k-step: within a k-step the store follows the loads): s_waitcnt vmcnt(that many) = every halo load has landed
  static constexpr int stores_behind_loads() {
    int c = 0;
    for (int i = 0; i < NPIECE; ++i) c += piece_ks(i) >= load_ks(NLOAD - 1) ? 1 : 0;
    return c;
  }
  static constexpr int nload_at(int ks) {  // halo items staged (and refilled) at k-step ks
    int c = 0;
    for (int i = 0; i < NLOAD; ++i) c += load_ks(i) == ks ? 1 : 0;
    return c;
  }
};

// same tap / channel -> k-slot map as conv2d.hip's bx_weight
template <int CIN, int COUT>
__device__ __forceinline__ float f2_weight(const float* w, int stride_row, int mode, int wo, int wi, int ks, int lg, int j, int co) {
  const int tap = CIN == 32 ? ks : 2 * ks + (lg >> 1);
  const int c = CIN == 32 ? 8 * lg + j : 8 * (lg & 1) + j;
  if (tap > 8) return 0.f;
  if (mode == 0) return (co < wo && c < wi) ? w[co * stride_row + c * 9 + tap] : 0.f;
  return (c < wo && co < wi) ? w[c * stride_row + co * 9 + (8 - tap)] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// forward / input gradient.  Structure of conv2d.hip's conv_bf16x3_kernel (8 waves = 16 x 16 output pixels, weights resident in
// LDS, halo of tile t+1 fetched into registers and the epilogue of tile t-1 run from a register copy during the MFMA loop of
// tile t, tap loop one basic block with an explicit issue order), with two operand planes and three products per k-step.
// INACT / INGN as there: x staged as x * act'(xact) / as GroupNorm(x).
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (make stamp, scripts/stamp_bf16x3.py f16x2): per-wave s_memtime sums of the kernel's phases
#ifdef BX_STAMP
__device__ unsigned long long f2_stamps[256 * 8 * 8];
#define F2_T(k)                                                   \
  {                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    st_[k] += now_ - last_;                                       \
    last_ = now_;                                                 \
  }
extern "C" int dis_debug_f2_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(f2_stamps), sizeof(f2_stamps));
}
#else
#define F2_T(k)
#endif
// Diagnostic builds only (scripts/diag/conv_bound.py; never part of libdis_hip.so): knock-outs of one resource at a time
// (F2_KO_STORE: the epilogue's stores go out of range and are dropped; F2_KO_LOAD: every halo load reads tile 0 of sample 0,
// i.e. cache hits with the same data statistics; F2_KO_MFMA: no matrix instructions; F2_KO_SPLIT: the staged items are written
// without the two-term split; F2_KO_EPI: no activation / statistics arithmetic) and F2_CLK: the in-kernel clock from
// s_memtime / s_memrealtime around the whole kernel (MI355X_MICROARCH.md, DVFS give-back item 6).  Results are WRONG by design.
#if defined(F2_KO_MFMA) || defined(F2W_KO_MFMA)
// (an empty asm statement keeps the operand registers alive: the fragment reads stay)
__device__ __forceinline__ f32x4 f2_no_mfma(f16x8_t A, f16x8_t B, f32x4 C) {
  asm volatile("; operands kept: %0 %1" ::"v"(A), "v"(B));
  return C;
}
#endif
#ifdef F2_KO_MFMA
#define F2_MFMA(A, B, C) f2_no_mfma(A, B, C)
#else
#define F2_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#endif
#ifdef F2_CLK
__device__ unsigned long long f2_clk[256 * 2];
extern "C" int dis_debug_f2_clk(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(f2_clk), sizeof(f2_clk)); }
#endif

// GEN: channel-slice form (DispNetS layers as 32 x 32 slices of wider tensors, conv2d.hip dis_bx_slices_run): x / y point at the
// slice's first channel, a pixel occupies a.ldx / a.ldy floats, a.cx / a.cy channels of the slice exist (the rest load zeros /
// are not stored), a.nbias bias entries exist; ACT may be ReLU.
// INCOEF: x is staged as act'(xact) * (x * k1_c + xact * kx + k0), the elementwise pass of a GroupNorm backward (ConvArgs::gnb_coef),
// and the staged values of the pixels a tile owns are stored to gnb_out (INACT: the activation between this conv and the GroupNorm).
// INRES (with INGN): x is staged as SELU(GroupNorm(x) + xact) - the output of a ResNetBlock, formed on load (the arithmetic of
// gn_apply_kernel, bit for bit) - and the staged values of the pixels a tile owns are stored to gnb_out: the block's output tensor
// is written by the conv that consumes it first instead of by a pass of its own (ConvArgs::gnb_out).
template <int CIN, int COUT, int ACT, bool ACCUM, bool STATS, int INACT = 0, bool INGN = false, bool EPIAB = false, int EPIACT = 0,
          bool GEN = false, bool INCOEF = false, bool INRES = false>
__global__ __launch_bounds__(512) void conv_f16x2_kernel(ConvArgs a) {
  using C = F2Cfg<CIN, COUT>;
  static_assert(!INCOEF || (!GEN && !INGN && !STATS && ACT == DIS_ACT_NONE && CIN == COUT), "GroupNorm backward on load: input-gradient instances");
  static_assert(!INRES || (INGN && !GEN && !INCOEF && INACT == 0 && !ACCUM && !EPIAB), "residual GroupNorm output on load: forward instances");
  constexpr bool IN2 = INACT != 0 || INCOEF || INRES;   // a second operand rides with every halo item
  static_assert(!GEN || (!STATS && INACT == 0 && !INGN && !EPIAB && CIN == 32 && COUT == 32), "slice form: plain convolution / input gradient");
  const int ldx = GEN ? a.ldx : CIN, ldy = GEN ? a.ldy : COUT;  // floats per pixel
  static_assert(!EPIAB || (!STATS && ACT == DIS_ACT_NONE && 2 * COUT <= 64), "channel sums: plain input-gradient instances");
  static_assert(EPIACT == 0 || EPIAB, "activation gradient at the output: only with the channel sums");
#ifdef BX_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef F2_CLK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int IC = C::IC, PS = C::PS, NT = C::NT, KS = C::KS, NLOAD = C::NLOAD, NPIECE = C::NPIECE, CV = C::CV, NP = C::NP;
  static_assert(!INGN || (INACT == 0 && 512 % CV == 0), "GroupNorm on load: forward instances");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* wl = smem16;
  unsigned short* xl = smem16 + C::W_U16;
  double* red = (double*)(smem16 + C::W_U16 + C::NXB * C::X_U16);
  float* mxs = (float*)(smem16 + C::W_U16 + C::NXB * C::X_U16 + 32);  // [parity][wave]
  float* abw = (float*)(smem16 + C::W_U16 + C::NXB * C::X_U16 + 64);  // EPIAB: [wave][2 COUT]
  unsigned short* pad16 = smem16 + C::W_U16 + C::NXB * C::X_U16 + 64 + 8 * 2 * COUT * 2 + CIN;  // target of idle threads' LDS writes

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.wv + F2_TC - 1) / F2_TC, tiles_y = (a.hv + F2_TR - 1) / F2_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);
  const int d_tx = per % tiles_x, d_ty = (per / tiles_x) % tiles_y, d_n = per / (tiles_x * tiles_y);

  // PF2: TWO register sets of halo items - the loads of tile t + 3 are issued while tile t is multiplied (the items of tile t + 1
  // are staged from one set, tile t + 2 is in flight in the other): a halo load has more than a whole tile period to land instead
  // of half of one.  Measured (scripts/diag/conv_bound.py, profiles/r5_dominant_bound.md): the kernel's loads were issued ~half
  // a tile period ahead of their use while HBM answers in 2 - 3 us at 4 - 5 TB/s of traffic, > 1 us of every tile exposed.
  // Instances that already sit at the register limit of two waves per SIMD (the channel-sum epilogues, the fused activation
  // gradient's second operand) keep one set.
  constexpr bool PF2 = F2_PF2 && !EPIAB && !(INRES && CIN == 32 && !F2_PF2_INRES);   // (32-channel INRES + two sets of two operands: spills)
  constexpr int PFD = PF2 ? 2 : 1;   // tiles between the tile whose items are prepared / staged and the tile whose loads are issued
  float4 pre[NLOAD], preB[NLOAD], pre2[IN2 ? NLOAD : 1], pre2B[IN2 ? NLOAD : 1];   // (preB / pre2B: PF2 only)
  int it_rc[NLOAD], it_off[NLOAD];
  unsigned it_own = 0u;   // INCOEF: bit it = item it lies in the 16 x 16 pixels the tile owns (pad == 1: halo rows / columns 1 .. 16)
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    const int idx = (int)threadIdx.x + it * 512;
    const int vv = idx % CV, pix = idx / CV;
    const int r = pix / IC, c = pix % IC;
    (void)r;
    if ((INCOEF || INRES) && idx < C::NITEMS && r >= 1 && r <= F2_TR && c >= 1 && c <= F2_TC) it_own |= 1u << it;
    // (halo column; items past the end of the halo - and, GEN, channels the slice does not have - are never in range: zeros)
    it_rc[it] = (idx < C::NITEMS && (!GEN || vv * 4 < a.cx)) ? c : 0x40000000;
    it_off[it] = ((r * a.win + c) * ldx + vv * 4) * 4;
  }
  const unsigned x_bytes = GEN ? ((unsigned)a.hin * a.win * ldx - a.x_sub) * 4u : (unsigned)a.hin * a.win * (CIN * 4u);
  const unsigned y_bytes = GEN ? ((unsigned)a.hf * a.wf * ldy - a.y_sub) * 4u : (unsigned)a.hf * a.wf * (COUT * 4u);
  // where a halo tile lies: the sample's base and byte range (0 when the tile does not exist: its loads return zeros), the
  // coordinates of the halo's first pixel and their byte offset
  struct Pf {
    const float* x;
    unsigned bytes;
    int iy0, ix0, off0;
  };
  auto pf_make = [&](int n, int ty, int tx, bool live) -> Pf {
#ifdef F2_KO_LOAD
    n = 0, ty = 1, tx = 1;
#endif
    Pf f;
    f.iy0 = ty * F2_TR - a.pad_y;
    f.ix0 = tx * F2_TC - a.pad_x;
    f.off0 = (f.iy0 * a.win + f.ix0) * (ldx * 4);
    f.x = a.x + (long)n * a.hin * a.win * ldx;
    f.bytes = live ? x_bytes : 0u;
    return f;
  };
  auto pf_issue = [&](float4 (&P)[NLOAD], float4 (&Q)[IN2 ? NLOAD : 1], const Pf& f, int it) {
    // rows above / below the sample leave the sample's buffer range by themselves (the offset wraps below 0 or passes its
    // end): only the column needs a test - 4 vector instructions per item, the loop is vector-issue-bound
    const int ix = f.ix0 + it_rc[it];
    const unsigned off = (unsigned)ix < (unsigned)a.win ? (unsigned)(f.off0 + it_off[it]) : BX_OOB;
    P[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(f.x, f.bytes), off, 0, 0));
    if (IN2)
      Q[it] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.xact + (f.x - a.x), f.bytes), off, 0, 0));
  };
  float4 gn_g = make_float4(0.f, 0.f, 0.f, 0.f), gn_b = gn_g, gn_sc = gn_g, gn_sh = gn_g;
  int gn_n = -1;
  float4 cf_k1 = gn_g;     // INCOEF: this thread's four k1_c, and the sample's kx, k0
  float cf_kx = 0.f, cf_k0 = 0.f;
  if (INGN) {
    gn_g = *(const float4*)(a.gn_gamma + ((int)threadIdx.x % CV) * 4);
    gn_b = *(const float4*)(a.gn_beta + ((int)threadIdx.x % CV) * 4);
  }
  // (1) BEFORE the barrier between two tiles: the final fp32 values of the halo items in flight (activation gradient / GroupNorm
  // applied) and this wave's largest magnitude, left in LDS for the other waves
  auto prep = [&](float4 (&P)[NLOAD], float4 (&Q)[IN2 ? NLOAD : 1], const Pf& f, int n_cur, int parity, bool in_loop = false) {
    if (INGN && n_cur != gn_n) {
      gn_n = n_cur;
      float mean, rstd;
      // (after a workgroup's last tile the look-ahead sample index can pass the batch: its values are never used, the read is clamped)
      gn_moments(a.gn_stats, n_cur < a.n ? n_cur : a.n - 1, (double)a.hin * a.win * CIN, a.gn_eps, &mean, &rstd);
      gn_sc = make_float4(rstd * gn_g.x, rstd * gn_g.y, rstd * gn_g.z, rstd * gn_g.w);
      gn_sh = make_float4(gn_b.x - gn_sc.x * mean, gn_b.y - gn_sc.y * mean, gn_b.z - gn_sc.z * mean, gn_b.w - gn_sc.w * mean);
    }
    if (INCOEF && n_cur != gn_n) {
      gn_n = n_cur;
      const float* cf = a.gnb_coef + (long)(n_cur < a.n ? n_cur : a.n - 1) * (CIN + 2);
      cf_k1 = *(const float4*)(cf + ((int)threadIdx.x % CV) * 4);
      cf_kx = cf[CIN];
      cf_k0 = cf[CIN + 1];
    }
    const bool gn_interior = INGN && f.iy0 >= 0 && f.ix0 >= 0 && f.iy0 + C::IR <= a.hin && f.ix0 + C::IC <= a.win;
    // (one register set: everything in flight is waited for here, in one place; two sets: the other set's loads - and the stores
    //  behind them - stay in flight, the compiler places the counted waits, vmcnt counts loads and stores together in issue order)
#if F2_WAIT_MODE == 0
    if (!PF2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#elif F2_WAIT_MODE == 2
    if (!PF2) {
      if (in_loop) __builtin_amdgcn_s_waitcnt(0x0F70 | C::stores_behind_loads());
      else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
#endif
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      float4 v = P[it];
      if (INGN) {
        f32x2 sh_lo = {gn_sh.x, gn_sh.y}, sh_hi = {gn_sh.z, gn_sh.w};
        if (!gn_interior) {
          const int iy = f.iy0 + ((int)threadIdx.x + it * 512) / (CV * IC), ix = f.ix0 + it_rc[it];
          const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
          sh_lo = ok ? sh_lo : (f32x2){0.f, 0.f};
          sh_hi = ok ? sh_hi : (f32x2){0.f, 0.f};
        }
        const f32x2 lo = (f32x2){v.x, v.y} * (f32x2){gn_sc.x, gn_sc.y} + sh_lo;
        const f32x2 hi = (f32x2){v.z, v.w} * (f32x2){gn_sc.z, gn_sc.w} + sh_hi;
        v = make_float4(lo[0], lo[1], hi[0], hi[1]);
        if (INRES) {   // + residual, SELU (padding: 0 * sc + 0 + 0 -> SELU(0) = 0 exactly), and the block output's owner stores it
          const float4 q = Q[it];
          v = make_float4(act_apply(v.x + q.x, DIS_ACT_SELU), act_apply(v.y + q.y, DIS_ACT_SELU), act_apply(v.z + q.z, DIS_ACT_SELU),
                          act_apply(v.w + q.w, DIS_ACT_SELU));
          const int ixs = f.ix0 + it_rc[it];
          const unsigned offs = (unsigned)ixs < (unsigned)a.win ? (unsigned)(f.off0 + it_off[it]) : BX_OOB;
          const u32x4 sv = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
          __builtin_amdgcn_raw_buffer_store_b128(sv, bx_rsrc(a.gnb_out + (f.x - a.x), f.bytes), ((it_own >> it) & 1u) ? offs : BX_OOB, 0, 0);
        }
      }
      if (INCOEF) {   // (the arithmetic of gn_apply_coef_kernel, bit for bit; padding: g = q = 0 loads give k0, which must not be staged)
        const float4 q = Q[it];
        const int ix = f.ix0 + it_rc[it];
        const unsigned off = (unsigned)ix < (unsigned)a.win ? (unsigned)(f.off0 + it_off[it]) : BX_OOB;
        const bool inside = off < f.bytes;   // (rows above / below the sample: the offset leaves the sample's byte range)
        v.x = __builtin_fmaf(v.x, cf_k1.x, __builtin_fmaf(q.x, cf_kx, cf_k0));
        v.y = __builtin_fmaf(v.y, cf_k1.y, __builtin_fmaf(q.y, cf_kx, cf_k0));
        v.z = __builtin_fmaf(v.z, cf_k1.z, __builtin_fmaf(q.z, cf_kx, cf_k0));
        v.w = __builtin_fmaf(v.w, cf_k1.w, __builtin_fmaf(q.w, cf_kx, cf_k0));
        if (INACT) {
          v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
          v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
        }
        v = inside ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        const u32x4 sv = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(sv, bx_rsrc(a.gnb_out + (f.x - a.x), f.bytes), ((it_own >> it) & 1u) ? off : BX_OOB, 0, 0);
      } else if (INACT) {
        const float4 q = Q[it];
        v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
        v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
      }
      // (items past the end of the halo - last round only - were loaded out of range, zeros, and are not staged; under GroupNorm
      //  on load they would carry the shift into the tile's maximum)
      if (INGN && (it + 1) * 512 > C::NITEMS && (int)threadIdx.x + it * 512 >= C::NITEMS) v = make_float4(0.f, 0.f, 0.f, 0.f);
      P[it] = v;
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.x)), fabsf(v.y));   // -> v_max3_f32 with |.| modifiers
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.z)), fabsf(v.w));
    }
    m = f2_wave_max(m);
    if (lane == 0) mxs[parity * 8 + wave] = m;
  };
  // (2) AFTER it: the tile's scale from the eight maxima (returns its exponent) ...
  auto tile_scale = [&](int parity) -> int {
    const float4 m0 = *(const float4*)(mxs + parity * 8), m1 = *(const float4*)(mxs + parity * 8 + 4);
    const float m = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    return f2_scale_exp(m);
  };
  // ... and, item by item, scale, split, LDS write into halo buffer xb.  For every tile but a workgroup's first these ride in
  // the matrix loop of the tile before (the even k-steps; the deferred epilogue has the odd ones), each item followed by the
  // load that refills its registers with the tile after.
  auto stage_item = [&](float4 (&P)[NLOAD], int it, float sc, unsigned short* xb) {
    const float4 v = P[it];
    unsigned a1, a2, b1, b2;
#ifdef F2_KO_SPLIT
    a1 = __float_as_uint(v.x) & 0x3fff3fffu, a2 = __float_as_uint(v.y) & 0x3fff3fffu, b1 = __float_as_uint(v.z) & 0x3fff3fffu, b2 = __float_as_uint(v.w) & 0x3fff3fffu;
    (void)sc;
#else
    f2_split_pair_scaled(v.x, v.y, sc, a1, a2);
    f2_split_pair_scaled(v.z, v.w, sc, b1, b2);
#endif
    const int idx = (int)threadIdx.x + it * 512;
    unsigned short* p = xb + (idx / CV) * PS + (idx % CV) * 4;
    // (the last round is partial: its idle threads write a pad - a select instead of a branch, the tap loop stays one basic block)
    if ((it + 1) * 512 > C::NITEMS) p = idx < C::NITEMS ? p : pad16 - CIN;
    *(uint2*)(p) = make_uint2(a1, b1);
    *(uint2*)(p + CIN) = make_uint2(a2, b2);
  };

  int tile = t_lo + rank;
  int cn = 0, cty = 0, ctx = 0;
  Pf pfs[PFD];   // pfs[0]: the tile whose items are prepared next (the one after the current tile); pfs[PFD - 1]: the last one issued
  pfs[0] = pf_make(0, 0, 0, false);
  if (PF2) pfs[PFD - 1] = pfs[0];
  if (tile < t_hi) {
    ctx = tile % tiles_x, cty = (tile / tiles_x) % tiles_y, cn = tile / (tiles_x * tiles_y);
    pfs[0] = pf_make(cn, cty, ctx, true);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) pf_issue(pre, pre2, pfs[0], it);  // in flight while the weights are split
  }
  // ---- weights: OIHW fp32 -> scaled fp16 planes in fragment order.  Coalesced copy into the second (still unused) halo buffer,
  // rows padded by one float, largest magnitude over the block, then (k-step, lane group, cout) units of 2 x 8 fp16.  The
  // workgroup's FIRST tile is staged in between (nothing to hide it behind), so that the loads of its second tile are in flight
  // while the weights are split: one barrier for weights + first tile, the next one is the first loop iteration's.
  int sw_e = 0, parity = 0, buf = 0, sx_e = 0;
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    tx_ += d_tx, ty_ += d_ty, n_ += d_n;
    if (tx_ >= tiles_x) tx_ -= tiles_x, ++ty_;
    if (ty_ >= tiles_y) ty_ -= tiles_y, ++n_;
  };
  int n1 = cn, ty1 = cty, tx1 = ctx;   // the tile after the current one
  int nl = cn, tyl = cty, txl = ctx;   // the last tile whose loads were issued
  {
    float* ws = (float*)(xl + C::X_U16);
    float* wmx = (float*)(red + 4);   // the weights' eight wave maxima (their own slots: mxs is the tiles')
    const int row = a.w_i * 9;
    float m = dis_copy_w_rows(a.w, a.w_o, row, a.w_rs, ws);
    m = f2_wave_max(m);
    if (lane == 0) wmx[wave] = m;
    if (threadIdx.x == 0) {  // statistics accumulators (stats_flush)
      red[0] = 0.0;
      red[1] = 0.0;
      *(unsigned*)(red + 2) = 0u;
      *(unsigned*)(red + 3) = 0u;  // (ab_flush)
    }
    if (tile < t_hi) prep(pre, pre2, pfs[0], cn, parity);
    __syncthreads();
    const float4 m0 = *(const float4*)(wmx), m1 = *(const float4*)(wmx + 4);
    sw_e = f2_scale_exp(fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w))));
    const float sw = __builtin_ldexpf(1.f, sw_e);
    if (tile < t_hi) {
      sx_e = tile_scale(parity);
      const float sc = __builtin_ldexpf(1.f, sx_e);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) stage_item(pre, it, sc, xl);
      advance(n1, ty1, tx1);
      nl = n1, tyl = ty1, txl = tx1;
      pfs[0] = pf_make(n1, ty1, tx1, tile + per < t_hi);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it) pf_issue(pre, pre2, pfs[0], it);
      if constexpr (PF2) {   // ... and the tile after that one into the second register set
        advance(nl, tyl, txl);
        pfs[PFD - 1] = pf_make(nl, tyl, txl, tile + 2 * per < t_hi);
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) pf_issue(preB, pre2B, pfs[PFD - 1], it);
      }
    }
    parity ^= 1;
    for (int u = threadIdx.x; u < KS * 4 * COUT; u += 512) {
      const int co = u % COUT, g = (u / COUT) & 3, ks = u / (4 * COUT);
      unsigned pl[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = f2_weight<CIN, COUT>(ws, row + 1, a.wmode, a.w_o, a.w_i, ks, g, 2 * j, co);
        const float v1 = f2_weight<CIN, COUT>(ws, row + 1, a.wmode, a.w_o, a.w_i, ks, g, 2 * j + 1, co);
        f2_split_pair(v0 * sw, v1 * sw, pl[0][j], pl[1][j]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p)
        *(uint4*)(wl + (((ks * NP + p) * 4 + g) * COUT + co) * 8) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
    // (no barrier here: the first loop iteration's publishes the weight planes and the first halo buffer, and by then every
    //  wave has finished reading `ws` - the second halo buffer is first written after that barrier)
  }

  f32x4 acc[2][NT], outv[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) outv[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};  // (the first tile's ride has no finished tile: 0 * livef)
  float4 prevy[NPIECE];
  float4 prevx[EPIAB ? NPIECE : 1], epix[EPIAB ? NPIECE : 1];  // EPIAB: the GroupNorm input at this / the deferred tile's outputs
  float4 prevyv[EPIACT ? NPIECE : 1];                           // EPIACT: the activation output at this tile's outputs
  float sA[NT][4], sB[NT][4];                                   // ... and this lane's sums of g, g * x for the current sample
  int ab_n = -1;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[nt][r] = sB[nt][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (EPIAB ? NPIECE : 1); ++i) epix[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  double s1 = 0.0, s2 = 0.0;
  float t1 = 0.f, t2 = 0.f;
  int stat_n = -1;
  float4 bias_v[NT];
  unsigned ymask[NT];  // GEN: 0 where this lane's 4 output channels of block nt exist in the slice, else the drop-this-store offset
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bias_v[nt] = (a.bias && (!GEN || nt * 16 + lg * 4 < a.nbias)) ? *(const float4*)(a.bias + nt * 16 + lg * 4)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    ymask[nt] = (!GEN || nt * 16 + lg * 4 < a.cy) ? 0u : BX_OOB;
  }
  const int yrow = a.osy * a.wf * (ldy * 4);
  const int y_lane = ((wave * 2 * a.osy * a.wf + li * a.osx) * ldy + lg * 4) * 4;

  const float* prev_y = a.y;
  unsigned prev_off[2] = {BX_OOB, BX_OOB};
  int prev_n = -1;
  // GroupNorm statistics leave through ONE fp64 atomic pair per workgroup and sample, without a block-wide barrier: every
  // wave adds its sums (DPP reduction) to two LDS accumulators and counts itself in; the wave that finds the count complete
  // moves the totals to memory and clears the slots (the next flush is at least one tile - two barriers - away).  The two
  // block_sum_d of the bf16x3 kernel cost 4 barriers and 24 LDS shuffles per flush: ~9 us per launch at 2 - 3 flushes per
  // workgroup (scripts/diag/conv_fixed_cost.py).
  auto stats_flush = [&]() {
    const double r1 = f2_wave_sum_d(s1), r2 = f2_wave_sum_d(s2);
    if (lane == 63) {
      __hip_atomic_fetch_add(red, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(red + 1, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // acq_rel at workgroup scope: this wave's two sums are ordered before its count, and the wave that completes the count sees
      // every other wave's (the hardware completes a wave's LDS operations in order; the ordering is spelled out for the compiler)
      const unsigned arrived = __hip_atomic_fetch_add((unsigned*)(red + 2), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (arrived == 7u) {
        const double q1 = __hip_atomic_load(red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double q2 = __hip_atomic_load(red + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        atomic_add_d(a.stats + 2 * stat_n, q1);
        atomic_add_d(a.stats + 2 * stat_n + 1, q2);
        red[0] = 0.0;
        red[1] = 0.0;
        *(unsigned*)(red + 2) = 0u;
      }
    }
    s1 = 0.0;
    s2 = 0.0;
  };
  auto stats_sample = [&]() {
    if (STATS && prev_n >= 0 && prev_n != stat_n) {
      if (stat_n >= 0) stats_flush();
      stat_n = prev_n;
    }
  };
  // EPIAB: a sample's channel sums leave the workgroup: row reduction over the 16 pixel lanes (DPP), per-wave slots in LDS, the
  // wave that arrives last adds the eight slots in a fixed order and stores the workgroup's slot of ab_out (no atomics)
  auto ab_flush = [&]() {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float va = sA[nt][r], vb = sB[nt][r];
#define F2_ROW(ctrl)                                                                                       \
  va += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(va), ctrl, 0xf, 0xf, true)); \
  vb += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(vb), ctrl, 0xf, 0xf, true));
        F2_ROW(0xB1) F2_ROW(0x4E) F2_ROW(0x124) F2_ROW(0x128)
#undef F2_ROW
        if (li == 0) {
          abw[wave * 2 * COUT + nt * 16 + lg * 4 + r] = va;
          abw[wave * 2 * COUT + COUT + nt * 16 + lg * 4 + r] = vb;
        }
        sA[nt][r] = 0.f;
        sB[nt][r] = 0.f;
      }
    unsigned arrived = 0;
    // release: the abw[] slot stores above (plain stores of other lanes of this wave) are ordered before the count; acquire:
    // the last arriver's slot reads below are ordered after it.  The fences cover the whole wave's accesses, the atomic is
    // lane 0's.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) arrived = __hip_atomic_fetch_add((unsigned*)(red + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (arrived == 7u) {
      if (lane < 2 * COUT) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) t += (double)abw[wv * 2 * COUT + lane];
        a.ab_out[((long)ab_n * a.ab_slots + blockIdx.x) * (2 * COUT) + lane] = t;
      }
      if (lane == 0) *(unsigned*)(red + 3) = 0u;
    }
  };
  auto ab_sample = [&]() {
    if (EPIAB && prev_n >= 0 && prev_n != ab_n) {
      if (ab_n >= 0) ab_flush();
      ab_n = prev_n;
    }
  };
  auto abx_load = [&](const float* xb, const float* yvb, const unsigned (&off)[2]) {
#pragma unroll
    for (int i = 0; i < (EPIAB ? NPIECE : 0); ++i) {
      prevx[i] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(xb, y_bytes), off[i / NT] + (i % NT) * 64, 0, 0));
      if (EPIACT)
        prevyv[i] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(yvb, y_bytes), off[i / NT] + (i % NT) * 64, 0, 0));
    }
  };
  auto epi_load = [&](const float* yb, const unsigned (&off)[2]) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i)
      prevy[i] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(yb, y_bytes), off[i / NT] + (i % NT) * 64, 0, 0));
  };
  float livef[2] = {0.f, 0.f};  // 1 for the deferred tile's pixels of this lane that exist (statistics)
  auto epi_piece = [&](int i) {
    const int mt = i / NT, nt = i % NT;
    f32x2 lo = {outv[mt][nt][0], outv[mt][nt][1]}, hi = {outv[mt][nt][2], outv[mt][nt][3]};
#ifdef F2_KO_EPI
    if (false) {
#else
    if (ACT == DIS_ACT_SELU) {
#endif
      // scale * max(x, 0) + scale * alpha * (exp(min(x, 0)) - 1), branch-free (for x > 0 the second term is sa * 1 - sa = 0
      // exactly), as two packed fused multiply-adds per pair: 5.5 issue slots per element instead of 9
      constexpr float L2E = 1.44269504088896340736f, SA = SELU_SCALE_F * SELU_ALPHA_F;
      const f32x2 nlo = (f32x2){fminf(lo[0], 0.f), fminf(lo[1], 0.f)} * (f32x2){L2E, L2E};
      const f32x2 nhi = (f32x2){fminf(hi[0], 0.f), fminf(hi[1], 0.f)} * (f32x2){L2E, L2E};
      const f32x2 elo = {__builtin_amdgcn_exp2f(nlo[0]), __builtin_amdgcn_exp2f(nlo[1])};
      const f32x2 ehi = {__builtin_amdgcn_exp2f(nhi[0]), __builtin_amdgcn_exp2f(nhi[1])};
      const f32x2 tlo = __builtin_elementwise_fma((f32x2){SA, SA}, elo, (f32x2){-SA, -SA});
      const f32x2 thi = __builtin_elementwise_fma((f32x2){SA, SA}, ehi, (f32x2){-SA, -SA});
      lo = __builtin_elementwise_fma((f32x2){SELU_SCALE_F, SELU_SCALE_F}, (f32x2){fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f)}, tlo);
      hi = __builtin_elementwise_fma((f32x2){SELU_SCALE_F, SELU_SCALE_F}, (f32x2){fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f)}, thi);
    } else if (ACT == DIS_ACT_RELU) {
      lo = (f32x2){fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f)};
      hi = (f32x2){fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f)};
    }
    const u32x4 ov = {__float_as_uint(lo[0]), __float_as_uint(lo[1]), __float_as_uint(hi[0]), __float_as_uint(hi[1])};
#ifdef F2_KO_STORE
    __builtin_amdgcn_raw_buffer_store_b128(ov, bx_rsrc(prev_y, y_bytes), (prev_off[mt] + nt * 64) | BX_OOB, 0, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b128(ov, bx_rsrc(prev_y, y_bytes), GEN ? ((prev_off[mt] + nt * 64) | ymask[nt]) : prev_off[mt] + nt * 64, 0, 0);
#endif
#ifdef F2_KO_EPI
    if (false) {
#else
    if (STATS) {
#endif
      const f32x2 sm = lo + hi, sq = lo * lo + hi * hi;
      t1 = __builtin_fmaf(livef[mt], sm[0] + sm[1], t1);
      t2 = __builtin_fmaf(livef[mt], sq[0] + sq[1], t2);
    }
    if (EPIAB) {
      const float4 xv = epix[i];
      const float g0 = lo[0] * livef[mt], g1 = lo[1] * livef[mt], g2 = hi[0] * livef[mt], g3 = hi[1] * livef[mt];
      sA[nt][0] += g0, sA[nt][1] += g1, sA[nt][2] += g2, sA[nt][3] += g3;
      sB[nt][0] = __builtin_fmaf(g0, xv.x, sB[nt][0]), sB[nt][1] = __builtin_fmaf(g1, xv.y, sB[nt][1]);
      sB[nt][2] = __builtin_fmaf(g2, xv.z, sB[nt][2]), sB[nt][3] = __builtin_fmaf(g3, xv.w, sB[nt][3]);
    }
  };

  const int xa_lane = (wave * 2 * IC + li) * PS + (CIN == 32 ? lg * 8 : (lg & 1) * 8);
  const bool hi_tap = (lg >> 1) != 0;
  auto iter = [&](float4 (&P)[NLOAD], float4 (&Q)[IN2 ? NLOAD : 1]) __attribute__((always_inline)) {
    const int vy0 = cty * F2_TR + wave * 2, vx0 = ctx * F2_TC + li;
    const int tile_yoff = ((cty * F2_TR * a.osy + a.ooy) * a.wf + ctx * F2_TC * a.osx + a.oox) * (ldy * 4) + y_lane;
    const float* cur_y = a.y + (long)cn * a.hf * a.wf * ldy;
    unsigned cur_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) cur_off[mt] = (vx0 < a.wv && vy0 + mt < a.hv) ? (unsigned)(tile_yoff + mt * yrow) : BX_OOB;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stats_sample();
    ab_sample();
    F2_T(0)
    // the NEXT tile's items (in flight since the previous matrix loop; zeros when there is no next tile): final values, maxima
    prep(P, Q, pfs[0], n1, parity, tile != t_lo + rank);   // (a workgroup's first iteration has no stores behind its halo loads yet)
    F2_T(1)
    // ONE barrier per tile: every wave has finished reading the other halo buffer (the previous tile), this tile's buffer is
    // completely written, the maxima of the next tile are visible
    __syncthreads();
    F2_T(2)
    const int sx_n = tile_scale(parity);
    const float sc_n = __builtin_ldexpf(1.f, sx_n);
    parity ^= 1;
    const unsigned short* xc = xl + buf * C::X_U16;        // this tile's halo
    unsigned short* xn = xl + (buf ^ 1) * C::X_U16;        // the next tile's
    int n2 = n1, ty2 = ty1, tx2 = tx1;
    advance(n2, ty2, tx2);
    advance(nl, tyl, txl);   // the tile whose loads this iteration issues (into the registers the staged items leave)
    const Pf pfn = pf_make(nl, tyl, txl, tile + (1 + PFD) * per < t_hi);
    F2_T(3)
    F2_T(4)
    t1 = 0.f;
    t2 = 0.f;

    auto ride = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if (ACCUM && ks == 0) epi_load(cur_y, cur_off);
      if (EPIAB && ks == 0) abx_load(a.ab_x + (cur_y - a.y), EPIACT ? a.ab_act_y + (cur_y - a.y) : nullptr, cur_off);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it)
        if (C::load_ks(it) == ks) stage_item(P, it, sc_n, xn);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it)
        if (C::load_ks(it) == ks) pf_issue(P, Q, pfn, it);
#pragma unroll
      for (int i = 0; i < NPIECE; ++i)
        if (C::piece_ks(i) == ks) epi_piece(i);
    };
    auto pattern = [&](auto nrc, auto ksc) {
      constexpr int NM = 6 * NT, NR = decltype(nrc)::value, ks = decltype(ksc)::value;
      constexpr int NIT = C::nload_at(ks), NSW = 2 * NIT, NLD = NIT * (INACT ? 2 : 1);
      constexpr int NEARLY = ks == 0 ? (ACCUM ? NPIECE : 0) + (EPIAB ? NPIECE * (EPIACT ? 2 : 1) : 0) : 0;
      static_assert(NSW <= NM, "one LDS write per matrix instruction gap");
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        if (g < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        // (an MFMA holds the pipe for 16 cycles and the SIMD's other wave issues one in between: up to ~6 VALU of this wave
        // fit behind each of its own MFMAs; with 3 - the bf16x3 kernel's figure, twice the MFMAs per k-step - the epilogue
        // VALU that did not fit was issued after the k-step's last MFMA, exposed)
        __builtin_amdgcn_sched_group_barrier(0x002, F2_VALU_PER_GAP, 0);
        if (NEARLY && g == (NR < NM ? NR : NM - 1)) __builtin_amdgcn_sched_group_barrier(0x020, NEARLY, 0);  // epilogue operands
        if (NSW && g >= NM - NSW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // the next tile's split items -> LDS
        if (NLD && g == NM - 1) __builtin_amdgcn_sched_group_barrier(0x020, NLD, 0);  // ... and the loads that refill them
      }
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // the k-step's store
      __builtin_amdgcn_sched_barrier(0);
    };
    // products (pixel plane, weight plane), smallest terms first
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
    if constexpr (CIN == 32) {
      s16x8 R[2][4][NP];     // [kx parity][halo row][plane]
      s16x8 fb[2][NP][NT];   // [buffer][plane][nt]
      auto load_row = [&](int kx, int j) {
#pragma unroll
        for (int p = 0; p < NP; ++p) R[kx & 1][j][p] = *(const s16x8*)(xc + xa_lane + (j * IC + kx) * PS + p * CIN);
      };
      auto load_w = [&](int ks, s16x8 (&B)[NP][NT]) {
#ifdef F2_EXP_NOW  // timing experiment (wrong results): no weight-fragment reads after the first tap
        if (ks > 1) return;
#endif
        const int wt = (ks % 3) * 3 + ks / 3;  // the weights are packed tap-major (ky * 3 + kx); the loop walks kx outer
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) B[p][nt] = *(const s16x8*)(wl + (((wt * NP + p) * 4 + lg) * COUT + nt * 16 + li) * 8);
      };
      load_row(0, 0);
      load_row(0, 1);
      load_w(0, fb[0]);
      auto step = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int kx = ks / 3, ky = ks % 3, nkx = (ks + 1) / 3, nky = (ks + 1) % 3;
        if (ks + 1 < KS) {
          if (nky == 0) {
            load_row(nkx, 0);
            load_row(nkx, 1);
          } else {
            load_row(nkx, nky + 1);
          }
          load_w(ks + 1, fb[(ks + 1) & 1]);
        }
        ride(ksc);
        constexpr int b = ks & 1;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
#ifdef F2_SNAKE   // (diagnostic: consecutive matrix instructions always share one operand)
              const int nt = ((mt + q) & 1) ? NT - 1 - nn : nn;
#else
              const int nt = nn;
#endif
              acc[mt][nt] = F2_MFMA(__builtin_bit_cast(f16x8_t, fb[b][PB[q]][nt]),
                                    __builtin_bit_cast(f16x8_t, R[kx & 1][ky + mt][PA[q]]), acc[mt][nt]);
            }
        pattern(std::integral_constant<int, (ks + 1 < KS ? (nky == 0 ? 2 * NP : NP) + NP * NT : 0)>{}, ksc);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    } else {
      s16x8 fa[2][NP][2], fb[2][NP][NT];
      auto load_frag = [&](int ks, s16x8 (&A)[NP][2], s16x8 (&B)[NP][NT]) {
        const int t0 = 2 * ks, t1_ = 2 * ks + 1 > 8 ? 8 : 2 * ks + 1;
        const int xoff = hi_tap ? ((t1_ / 3) * IC + t1_ % 3) * PS : ((t0 / 3) * IC + t0 % 3) * PS;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) A[p][mt] = *(const s16x8*)(xc + xa_lane + xoff + mt * IC * PS + p * CIN);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) B[p][nt] = *(const s16x8*)(wl + (((ks * NP + p) * 4 + lg) * COUT + nt * 16 + li) * 8);
        }
      };
      load_frag(0, fa[0], fb[0]);
      auto step = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if (ks + 1 < KS) load_frag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        ride(ksc);
        constexpr int b = ks & 1;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = F2_MFMA(__builtin_bit_cast(f16x8_t, fb[b][PB[q]][nt]),
                                    __builtin_bit_cast(f16x8_t, fa[b][PA[q]][mt]), acc[mt][nt]);
        pattern(std::integral_constant<int, (ks + 1 < KS ? NP * (2 + NT) : 0)>{}, ksc);
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
    }
    s1 += (double)t1;
    s2 += (double)t2;
    F2_T(5)

    // hand the finished tile over to the deferred epilogue: undo the two scales (exact), add the bias
    const float desc = __builtin_ldexpf(1.f, -(sx_e + sw_e));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      prev_off[mt] = cur_off[mt];
      livef[mt] = cur_off[mt] != BX_OOB ? 1.f : 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 bv = {bias_v[nt].x, bias_v[nt].y, bias_v[nt].z, bias_v[nt].w};
        outv[mt][nt] = acc[mt][nt] * desc + bv;
        if (ACCUM) {
          const float4 q = prevy[mt * NT + nt];
          outv[mt][nt] += (f32x4){q.x, q.y, q.z, q.w};
        }
        if (EPIACT) {  // gradient wrt the pre-activation value: what is stored, and what the channel sums are sums of
          const float4 q = prevyv[mt * NT + nt];
          outv[mt][nt] *= (f32x4){act_grad_from_out(q.x, EPIACT), act_grad_from_out(q.y, EPIACT), act_grad_from_out(q.z, EPIACT),
                                  act_grad_from_out(q.w, EPIACT)};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < (EPIAB ? NPIECE : 0); ++i) epix[i] = prevx[i];
    prev_y = cur_y;
    prev_n = cn;
    cn = n1, cty = ty1, ctx = tx1;
    n1 = n2, ty1 = ty2, tx1 = tx2;
    tile += per;
    sx_e = sx_n;
    buf ^= 1;
    if (PF2) pfs[0] = pfs[PFD - 1];
    pfs[PFD - 1] = pfn;
    F2_T(6)
  };
  if constexpr (PF2) {
    // the two register sets alternate: tiles go in PAIRS through a loop with ONE exit, an odd last tile has its own copy of the
    // body behind it.  (With a `break` between the two bodies the control-flow graph holds an edge from the end of the first
    // body to the loop header, and the compiler's wait-count pass then makes the header wait as if set A had been requested one
    // body ago - vmcnt(2) instead of vmcnt(12): the second set's loads would be waited for right after their issue.)
    const int ntl = tile < t_hi ? (t_hi - tile + per - 1) / per : 0;
    for (int pr = 0; pr < (ntl >> 1); ++pr) {
      iter(pre, pre2);
      iter(preB, pre2B);
    }
    if (ntl & 1) iter(pre, pre2);
  } else {
    while (tile < t_hi) iter(pre, pre2);
  }
  stats_sample();
  ab_sample();
  // A flush above (the last tile belongs to another sample than the sums so far) and the final flush below must not overlap: the
  // flushes are barrier-free - LDS slots, an arrival counter that the last wave clears - and inside the loop the tile barrier
  // separates two of them.  Without this barrier a fast wave's second flush rewrote its slot and bumped the counter while the
  // last wave of the first was still adding the slots up: sums of the wrong sample, and a final count that never completes
  // (seen only when wave timing shifts, e.g. two processes sharing the GPU: scripts/diag/share_gnin.py).
  if (STATS || EPIAB) __syncthreads();
  t1 = 0.f;
  t2 = 0.f;
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) epi_piece(i);
  s1 += (double)t1;
  s2 += (double)t2;
  if (STATS && stat_n >= 0) stats_flush();
  if (EPIAB && ab_n >= 0) ab_flush();
#ifdef F2_CLK
  if (threadIdx.x == 0 && blockIdx.x < 256) {
    f2_clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_t0;
    f2_clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
#ifdef BX_STAMP
  F2_T(7)
  if (lane == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) f2_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_[k];
#endif
}

template <int CIN, int COUT>
static hipError_t f2_launch(const ConvArgs& a, bool stats, int inact, long grid, hipStream_t stream) {
  using C = F2Cfg<CIN, COUT>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  const bool ingn = a.gn_stats != nullptr;
  static bool attr_set[32] = {};
  auto launch = [&](auto kern, int slot) -> hipError_t {
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG(CIN == 32 && COUT == 32 ? "conv_f16x2_kernel<32,32>" : CIN == 16 && COUT == 16 ? "conv_f16x2_kernel<16,16>" : CIN == 16 ? "conv_f16x2_kernel<16,32>" : "conv_f16x2_kernel<32,16>");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, stream, a);
    return hipSuccess;
  };
  if (a.act != DIS_ACT_NONE && a.act != DIS_ACT_SELU) return hipErrorInvalidValue;
  const bool selu = a.act == DIS_ACT_SELU;
  if (a.gnb_coef) {  // input gradient whose operand is the GroupNorm backward's elementwise pass, applied on load (INCOEF)
    if constexpr (CIN == COUT) {
      if (ingn || selu || stats || !a.xact || !a.gnb_out || (inact != 0 && inact != DIS_ACT_SELU)) return hipErrorInvalidValue;
      constexpr int S = DIS_ACT_SELU;
#ifndef F2_INCOEF_FAT
#define F2_INCOEF_FAT 1
#endif
      if (a.ab_out && a.accum && a.ab_act_y) {  // ... + the ResNetBlock-chain epilogue (sums of g, g * x2 behind SELU')
        if (!F2_INCOEF_FAT && CIN == 32) return hipErrorInvalidValue;
        if (inact != S) return hipErrorInvalidValue;
        return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, S, false, true, S, false, true>, 18);
      }
      if (a.ab_out && a.accum) {                // ... + the two-consumer epilogue (Block2D3D conv1_1)
        if (!F2_INCOEF_FAT && CIN == 32) return hipErrorInvalidValue;
        if (inact != S) return hipErrorInvalidValue;
        return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, S, false, true, 0, false, true>, 19);
      }
      if (a.ab_out)                             // ... + the channel sums of a GroupNorm-on-load pair (conv2d_gn_in)
        return inact ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, S, false, true, 0, false, true>, 20)
                     : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, 0, false, true, 0, false, true>, 21);
      if (a.accum)
        return inact ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, S, false, false, 0, false, true>, 22)
                     : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, 0, false, false, 0, false, true>, 23);
      return inact ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, S, false, false, 0, false, true>, 24)
                   : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, 0, false, false, 0, false, true>, 25);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.ab_out && inact) {  // ... of a conv that had an activation itself and whose INPUT was SELU(GroupNorm(.) + residual)
    // (final_conv 32 -> 16 behind ref_res3, reference model/multi_frame_networks.py:262-266, and - round 5 - the 16-channel slice of
    //  ref_conv (48 -> 32) behind amb_res2, :248-256: gy has 32 channels, the gradient 16)
    if constexpr ((CIN == 16 && COUT == 32) || (CIN == 32 && COUT == 16)) {
      if (ingn || selu || stats || a.accum || inact != DIS_ACT_SELU || !a.ab_x || !a.ab_act_y) return hipErrorInvalidValue;
      return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, DIS_ACT_SELU, false, true, DIS_ACT_SELU>, 16);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.ab_out) {  // input gradient + per-(sample, channel) sums for the GroupNorm backward
    if constexpr (CIN == COUT) {
      if (ingn || inact || selu || stats || !a.ab_x) return hipErrorInvalidValue;
      if (a.ab_act_y) {  // accumulating form whose result passes through SELU' (ResNetBlock chains)
        if (!a.accum) return hipErrorInvalidValue;
        return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, 0, false, true, DIS_ACT_SELU>, 11);
      }
      if (a.accum)  // accumulating form without an activation behind the GroupNorm (a GroupNorm output with two consumers)
        return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, 0, false, true, 0>, 17);
      return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, 0, false, true>, 10);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (ingn && a.gnb_out) {  // the residual form: x = SELU(GroupNorm(x2) + res), formed on load and stored by its owner tiles (INRES)
    if (inact || a.accum || !a.xact) return hipErrorInvalidValue;
    if constexpr (CIN == COUT) {
      if (selu)
        return stats ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, true, 0, true, false, 0, false, false, true>, 26)
                     : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, false, 0, true, false, 0, false, false, true>, 27);
      return hipErrorInvalidValue;
    } else if constexpr (CIN == 32 && COUT == 16) {   // (final_conv behind ref_res3)
      if (selu && !stats) return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, false, 0, true, false, 0, false, false, true>, 28);
      return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (ingn) {
    if constexpr (CIN == COUT) {
      if (inact || a.accum) return hipErrorInvalidValue;
      if (selu)
        return stats ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, true, 0, true>, 12)
                     : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, false, 0, true>, 13);
      return stats ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, true, 0, true>, 14)
                   : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, 0, true>, 15);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (inact) {
    if (inact != DIS_ACT_SELU || selu || stats) return hipErrorInvalidValue;
    return a.accum ? launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false, DIS_ACT_SELU>, 8)
                   : launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false, DIS_ACT_SELU>, 9);
  }
  switch ((selu ? 4 : 0) + (a.accum ? 2 : 0) + (stats ? 1 : 0)) {
    case 0: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, false>, 0);
    case 1: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, false, true>, 1);
    case 2: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, false>, 2);
    case 3: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_NONE, true, true>, 3);
    case 4: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, false>, 4);
    case 5: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, false, true>, 5);
    case 6: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, true, false>, 6);
    default: return launch(conv_f16x2_kernel<CIN, COUT, DIS_ACT_SELU, true, true>, 7);
  }
}

// channel-slice form (conv2d.hip dis_bx_slices_run): 32 x 32 slices, act NONE / ReLU, writing or accumulating
hipError_t dis_f2_conv_gen_launch(const ConvArgs& a, long grid, hipStream_t stream) {
  using C = F2Cfg<32, 32>;
  if (a.wmode < 0 || (a.act != DIS_ACT_NONE && a.act != DIS_ACT_RELU)) return hipErrorInvalidValue;
  static bool attr_set[4] = {};
  const int slot = (a.act == DIS_ACT_RELU ? 2 : 0) + (a.accum ? 1 : 0);
  auto launch = [&](auto kern) -> hipError_t {
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG("conv_f16x2_kernel<32,32,GEN> slices");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, stream, a);
    return hipSuccess;
  };
  if (slot == 0) return launch(conv_f16x2_kernel<32, 32, DIS_ACT_NONE, false, false, 0, false, false, 0, true>);
  if (slot == 1) return launch(conv_f16x2_kernel<32, 32, DIS_ACT_NONE, true, false, 0, false, false, 0, true>);
  if (slot == 2) return launch(conv_f16x2_kernel<32, 32, DIS_ACT_RELU, false, false, 0, false, false, 0, true>);
  return launch(conv_f16x2_kernel<32, 32, DIS_ACT_RELU, true, false, 0, false, false, 0, true>);
}

hipError_t dis_f2_conv_launch(const ConvArgs& a, int cin, int cout, bool stats, int inact, long grid, hipStream_t stream) {
  if (a.wmode < 0) return hipErrorInvalidValue;  // pre-packed bf16 planes: the bf16x3 kernel's format
  if (cin == 32 && cout == 32) return f2_launch<32, 32>(a, stats, inact, grid, stream);
  if (cin == 16 && cout == 16) return f2_launch<16, 16>(a, stats, inact, grid, stream);
  if (cin == 16 && cout == 32) return f2_launch<16, 32>(a, stats, inact, grid, stream);
  if (cin == 32 && cout == 16) return f2_launch<32, 16>(a, stats, inact, grid, stream);
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// weight gradient.  Structure of conv2d.hip's conv_wgrad_bf16x3_kernel<CIN, COUT> (4 waves, 8 x 16 output pixels per tile, both
// operands fetched with the transposing read ds_read_b64_tr_b16, the (9 CIN / 16) x (COUT / 16) accumulator tiles split
// statically over the waves and kept in registers across ALL tiles of a workgroup, 2 workgroups per CU), two planes, three
// products.  The contraction runs over pixels, i.e. over tiles, so the block scales of x and gy must not change under the
// accumulators: each is a RUNNING scale that only ever shrinks (when a tile brings a larger magnitude than any before it), and
// the accumulators are multiplied by the (exact, power-of-two) ratio at that moment.  Later tiles with smaller values then
// carry the coarser scale: their absolute error is bounded by 2^-24 x 2^-15 of the largest magnitude seen, which is what a sum
// over all pixels can resolve anyway.  The slab is written as acc * 2^-(sx + sg).
// ------------------------------------------------------------------------------------------------
// K x K taps, stride S, TR x 16 output pixels per tile, KH of the K tap rows per workgroup (blockIdx.z selects the group): the
// FuseNet instances are K = 3, S = 1, TR = 8; the slice-pair instances of DispNetS also use 5 x 5 / 7 x 7 taps and stride 2
// (the geometry of conv2d.hip's WxCfg).
template <int CIN, int COUT, int K_ = 3, int S_ = 1, int TR_ = 8, int KH_ = K_>
struct F2WxCfg {
  static constexpr int NP = 2, K = K_, S = S_, TR = TR_, KH = KH_;
  static constexpr int ps_for(int payload_u16, int step) {  // see WxCfg::ps_for (conflict-free transposing reads)
    int ps = (payload_u16 + 7) / 8 * 8;
    while (((step * (ps / 2)) % 16) != 8) ps += 8;
    return ps;
  }
  static constexpr int PSX = ps_for(NP * CIN, S), PSG = ps_for(NP * COUT, 1);
  static constexpr int CVX = CIN / 4, CVG = COUT / 4;
  static constexpr int IR = (TR - 1) * S + KH, IC = 15 * S + K;
  static constexpr int X_U16 = IR * IC * PSX, G_U16 = TR * 16 * PSG;
  // + wave maxima [parity][x|g][wave]; the bias partials (1024 floats, end of the kernel only) alias the x halo
  static constexpr int LDS_BYTES = (X_U16 + G_U16) * 2 + 64;
  static_assert(X_U16 * 2 >= 1024 * 4, "bias partials alias the x halo");
  static constexpr int NIX = IR * IC * CVX, NLX = (NIX + 255) / 256;
  static constexpr int NLG = TR * 16 * CVG / 256;
  static constexpr int KSN = TR / 2;
  static constexpr int MB = KH * K * CIN / 16, NB = COUT / 16, T = MB * NB, TW = (T + 3) / 4;
  static_assert(TR % 2 == 0 && (TR * 16 * CVG) % 256 == 0, "tile geometry");
};

#ifndef F2W_PF2
// two register sets of staged items in the weight-gradient kernel (see WPF2).  OFF: measured twice (round 4 by hand, round 5 with
// the pair loop and verified counted waits, vmcnt(19..10)): 504.8 / 499.2 frames/s without, 495.6 / 495.0 with it in alternating
// same-box runs (profiles/r5_ab_wgrad_pf2.txt) - the kernel's cycles are its instruction stream (scripts/diag/wgrad_bound.py), the
// loads it would hide are hidden by the CU's second workgroup already, and 68 more registers cost the matrix phase its schedule.
#define F2W_PF2 0
#endif
#ifndef F2W_FD
#define F2W_FD 1   // units the weight-gradient kernel's LDS fragment reads run ahead of their matrix instructions
#endif
#ifndef F2W_WPC
#define F2W_WPC 2   // workgroups per CU (= waves per SIMD) the weight-gradient kernel is built for
#endif
int dis_f2_wgrad_wpc() { return F2W_WPC; }
typedef short f2_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x8 f2_tr_read8(const unsigned short* p0, const unsigned short* p1) {
  const f2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f2_s16x4*)p0);
  const f2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) f2_s16x4*)p1);
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
template <int I, int N, class F>
__device__ __forceinline__ void f2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    f2_static_for<I + 1, N>(f);
  }
}

// GEN: channel-slice PAIRS of a wide layer (DispNetS, conv2d.hip dis_wgrad_pairs_run: 3 x 3, stride 1): blockIdx.y = gb * npx + cb
// selects x channels [32 cb, 32 cb + 32) and gy channels [COUT gb, + COUT) of pixels that occupy a.ldx / a.ldg floats; channels
// past the layer's last one load zeros; one slab per (pair, worker).
template <int CIN, int COUT, int INACT = 0, bool INGN = false, bool GEN = false, int K_ = 3, int S_ = 1, int TR_ = 8, int KH_ = K_,
          bool INCOEF = false>
__global__ __launch_bounds__(256, (K_ == 3 && S_ == 1) ? F2W_WPC : 1) void conv_wgrad_f16x2_kernel(WgArgs a) {
  using C = F2WxCfg<CIN, COUT, K_, S_, TR_, KH_>;
  static_assert(!GEN || (CIN == 32 && INACT == 0 && !INGN), "slice-pair form");
  static_assert(!INCOEF || (!GEN && !INGN && 256 % (COUT / 4) == 0), "GroupNorm backward on load: a thread keeps its 4 gy channels");
  constexpr bool G2 = INACT != 0 || INCOEF;   // the activation output / GroupNorm input rides with every gy item
#ifdef F2_CLK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  static_assert(GEN || (K_ == 3 && S_ == 1 && TR_ == 8 && KH_ == 3) ||
                    (K_ == 4 && S_ == 2 && TR_ == 4 && KH_ == 4 && CIN == 32 && COUT == 32 && (INACT == 0 || INCOEF) && !INGN),
                "the FuseNet forms: 3 x 3 stride 1, and the 4 x 4 stride-2 down convolution (32 -> 32)");
  constexpr int S = S_, KH = KH_;
  const int ky0 = KH < K_ ? (int)blockIdx.z * KH : 0;  // first tap row of this workgroup (7x7: two groups of 4 rows)
  const int ldx = GEN ? a.ldx : CIN, ldg = GEN ? a.ldg : COUT;
  const int cb = GEN ? (int)blockIdx.y % a.npx : 0, gbk = GEN ? (int)blockIdx.y / a.npx : 0;
  const int xc0 = GEN ? a.xoff + 32 * cb : 0, gc0 = GEN ? a.goff + COUT * gbk : 0;
  constexpr int NP = C::NP, K = C::K, TR = C::TR, WX_IC = C::IC;
  constexpr int PSX = C::PSX, PSG = C::PSG, NLX = C::NLX, NLG = C::NLG, NB = C::NB, TW = C::TW;
  static_assert(!INGN || 256 % C::CVX == 0, "a thread keeps its 4 channels over its items");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* xl = smem16;
  unsigned short* gl = smem16 + C::X_U16;
  float* bred = (float*)smem16;                                   // (after the last tile: aliases the x halo)
  float* mxs = (float*)(smem16 + C::X_U16 + C::G_U16);  // [parity][x | g][wave]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane >> 4, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + TR - 1) / TR;
  const int ntiles = a.n * tiles_y * tiles_x;

  f32x4 acc[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  // WPF2: two register sets of staged items (the FuseNet 3 x 3 forms): the loads of tile t + 2 are issued when tile t has been staged,
  // those of tile t + 1 are in flight meanwhile - a load has a whole tile period to land instead of one matrix phase (~1 us against
  // an HBM latency of 2 - 3 us under this traffic: scripts/diag/wgrad_bound.py, profiles/r5_dominant_bound.md)
  constexpr bool WPF2 = F2W_PF2 && !GEN && K_ == 3 && S_ == 1;
  struct WSet {
    float4 x[NLX], g[NLG], g2[G2 ? NLG : 1];
    int iy0, ix0, n;
    bool live;
  };
  WSet SA, SB;
  SA.iy0 = SA.ix0 = 0, SA.n = -1, SA.live = false, SB.iy0 = SB.ix0 = 0, SB.n = -1, SB.live = false;
  float4 cf_k1 = make_float4(0.f, 0.f, 0.f, 0.f);   // INCOEF: this thread's four k1_c, the sample's kx, k0
  float cf_kx = 0.f, cf_k0 = 0.f;
  int cf_n = -1;
  int ix_rc[NLX], ix_off[NLX], ig_rc[NLG], ig_off[NLG];
#pragma unroll
  for (int it = 0; it < NLX; ++it) {
    const int idx = (int)threadIdx.x + it * 256;
    const int vv = idx % C::CVX, pix = idx / C::CVX;
    const int r = pix / WX_IC, c = pix % WX_IC;
    // (items past the end of the halo - and, GEN, channels past the layer's last one: column never in range, zeros)
    ix_rc[it] = (idx < C::NIX && (!GEN || 32 * cb + vv * 4 < a.cx)) ? (r | (c << 16)) : 0x40000000;
    ix_off[it] = ((r * a.win + c) * ldx + xc0 + vv * 4) * 4;
  }
#pragma unroll
  for (int it = 0; it < NLG; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int vv = idx % C::CVG, pix = idx / C::CVG;
    ig_rc[it] = (!GEN || COUT * gbk + vv * 4 < a.cg) ? ((pix >> 4) | ((pix & 15) << 16)) : 0x40000000;
    ig_off[it] = (((pix >> 4) * a.wout + (pix & 15)) * ldg + gc0 + vv * 4) * 4;
  }
  const unsigned x_bytes_all = (unsigned)a.hin * a.win * (ldx * 4u), g_bytes_all = (unsigned)a.hout * a.wout * (ldg * 4u);
  int gn_n = -1;
  float4 gn_g = make_float4(0.f, 0.f, 0.f, 0.f), gn_b = gn_g, gn_sc = gn_g, gn_sh = gn_g;
  if (INGN) {
    gn_g = *(const float4*)(a.gn_gamma + ((int)threadIdx.x % C::CVX) * 4);
    gn_b = *(const float4*)(a.gn_beta + ((int)threadIdx.x % C::CVX) * 4);
  }
  // (a tile past the workgroup's last one: the loads are issued all the same, through empty descriptors - they return zeros nobody
  //  reads - so that the number of memory operations between two waits is the same on every path and the counted waits stay exact)
  // tile of trip j of this workgroup: each XCD (workgroup i sits on XCD i % 8) works through a contiguous eighth of the tiles, its
  // workgroups side by side - neighbouring tiles are in flight on ONE XCD at the same time and find their shared halo rows / columns in
  // its L2 (F2W_XCD=0: tile = blockIdx.x + j gridDim.x, neighbours on different XCDs, every halo item fetched from memory)
#ifndef F2W_XCD
#define F2W_XCD 1
#endif
  const int w_nx = (F2W_XCD && gridDim.x % 8 == 0) ? 8 : 1;
  const int w_xcd = (int)blockIdx.x % w_nx, w_rank = (int)blockIdx.x / w_nx, w_per = (int)gridDim.x / w_nx;
  const int w_lo = (int)((long)ntiles * w_xcd / w_nx), w_hi = (int)((long)ntiles * (w_xcd + 1) / w_nx);
  const int ntl = w_hi - w_lo > w_rank ? (w_hi - w_lo - w_rank + w_per - 1) / w_per : 0;   // trips of this workgroup
  auto prefetch = [&](WSet& W, int trip) __attribute__((always_inline)) {
    const bool live = trip < ntl;
    int tile = live ? w_lo + trip * w_per + w_rank : 0;
#ifdef F2W_KO_LOAD   // (diagnostic build: every tile reads tile 0 - cache hits with the same data statistics)
    tile = 0;
#endif
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * (TR * S) - a.pad + ky0, ix0 = tx * (16 * S) - a.pad;
    W.iy0 = iy0, W.ix0 = ix0, W.n = n, W.live = live;
    const unsigned x_bytes = live ? x_bytes_all : 0u, g_bytes = live ? g_bytes_all : 0u;
    const char* xb = (const char*)a.x + (long)n * a.hin * a.win * ldx * 4;
    const int xoff0 = (iy0 * a.win + ix0) * (ldx * 4);
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      // (rows above / below the sample leave the sample's buffer range by themselves: only the column is tested)
      const int ix = ix0 + (ix_rc[it] >> 16);
      const bool ok = (unsigned)ix < (unsigned)a.win;
      W.x[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                bx_rsrc(xb, x_bytes), ok ? (unsigned)(xoff0 + ix_off[it]) : BX_OOB, 0, 0));
    }
    const char* gb = (const char*)a.gy + (long)n * a.hout * a.wout * ldg * 4;
    const int goff0 = (ty * TR * a.wout + tx * 16) * (ldg * 4);
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int ox = tx * 16 + (ig_rc[it] >> 16);
      const bool ok = ox < a.wout;   // (rows below the sample: past the end of its buffer range)
      W.g[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                bx_rsrc(gb, g_bytes), ok ? (unsigned)(goff0 + ig_off[it]) : BX_OOB, 0, 0));
      if (G2)
        W.g2[it] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc((const char*)a.gact + (gb - (const char*)a.gy), g_bytes),
                                                          ok ? (unsigned)(goff0 + ig_off[it]) : BX_OOB, 0, 0));
    }
  };
  // (1) before the barrier: final fp32 values of the items in flight, this wave's two maxima into LDS
  auto prep = [&](WSet& W, int parity) __attribute__((always_inline)) {
    const int st_iy0 = W.iy0, st_ix0 = W.ix0, st_n = W.n;
    if (INCOEF && st_n != cf_n) {
      cf_n = st_n;
      const float* cf = a.gnb_coef + (long)(st_n < a.n ? st_n : a.n - 1) * (COUT + 2);
      cf_k1 = *(const float4*)(cf + ((int)threadIdx.x % C::CVG) * 4);
      cf_kx = cf[COUT];
      cf_k0 = cf[COUT + 1];
    }
    if (INGN && st_n != gn_n) {
      gn_n = st_n;
      float mean, rstd;
      gn_moments(a.gn_stats, st_n < a.n ? st_n : a.n - 1, (double)a.hin * a.win * CIN, a.gn_eps, &mean, &rstd);
      gn_sc = make_float4(rstd * gn_g.x, rstd * gn_g.y, rstd * gn_g.z, rstd * gn_g.w);
      gn_sh = make_float4(gn_b.x - gn_sc.x * mean, gn_b.y - gn_sc.y * mean, gn_b.z - gn_sc.z * mean, gn_b.w - gn_sc.w * mean);
    }
    const bool gn_interior = INGN && st_iy0 >= 0 && st_ix0 >= 0 && st_iy0 + C::IR <= a.hin && st_ix0 + C::IC <= a.win;
    if (!WPF2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)  (two sets: the other set stays in flight, the compiler counts)
    float mx = 0.f, mg = 0.f;
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      float4 v = W.x[it];
      if (INGN) {
        f32x2 sh_lo = {gn_sh.x, gn_sh.y}, sh_hi = {gn_sh.z, gn_sh.w};
        if (!gn_interior) {
          const int iy = st_iy0 + (ix_rc[it] & 0xffff), ix = st_ix0 + (ix_rc[it] >> 16);
          const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
          sh_lo = ok ? sh_lo : (f32x2){0.f, 0.f};
          sh_hi = ok ? sh_hi : (f32x2){0.f, 0.f};
        }
        const f32x2 lo = (f32x2){v.x, v.y} * (f32x2){gn_sc.x, gn_sc.y} + sh_lo;
        const f32x2 hi = (f32x2){v.z, v.w} * (f32x2){gn_sc.z, gn_sc.w} + sh_hi;
        v = make_float4(lo[0], lo[1], hi[0], hi[1]);
        W.x[it] = v;
      }
      const float mv = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)), fabsf(v.w));
      mx = (!INGN || (it + 1) * 256 <= C::NIX || (int)threadIdx.x + it * 256 < C::NIX) ? fmaxf(mx, mv) : mx;  // (not loaded: zeros)
    }
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      float4 v = W.g[it];
      unsigned gnb_off = BX_OOB;
      if (INCOEF) {   // (gn_apply_coef_kernel's arithmetic, bit for bit; the stored values feed the input-gradient launches)
        const float4 q = W.g2[it];
        // the item's own pixel, or nothing: positions past the map (ragged tiles) and the tiles past the workgroup's last one load
        // g = q = 0, which the affine map would turn into k0 - they must stay zero in the sums, in the bias gradient and in memory
        const int tx_ = (W.ix0 + a.pad) / (16 * S), ty_ = (W.iy0 + a.pad - ky0) / (TR * S);
        const int ox = tx_ * 16 + (ig_rc[it] >> 16);
        gnb_off = ox < a.wout ? (unsigned)((ty_ * TR * a.wout + tx_ * 16) * (ldg * 4) + ig_off[it]) : BX_OOB;
        const bool inside = gnb_off < (W.live ? g_bytes_all : 0u);
        v.x = __builtin_fmaf(v.x, cf_k1.x, __builtin_fmaf(q.x, cf_kx, cf_k0));
        v.y = __builtin_fmaf(v.y, cf_k1.y, __builtin_fmaf(q.y, cf_kx, cf_k0));
        v.z = __builtin_fmaf(v.z, cf_k1.z, __builtin_fmaf(q.z, cf_kx, cf_k0));
        v.w = __builtin_fmaf(v.w, cf_k1.w, __builtin_fmaf(q.w, cf_kx, cf_k0));
        if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (INACT) {
        const float4 q = W.g2[it];
        v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
        v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
        W.g[it] = v;
      }
      if (INCOEF) {
        W.g[it] = v;
        // (every gy pixel belongs to exactly one tile; out-of-range items are dropped by the offset / the descriptor's range)
        const u32x4 sv = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(sv, bx_rsrc((char*)a.gnb_out + (long)W.n * a.hout * a.wout * ldg * 4, W.live ? g_bytes_all : 0u),
                                               gnb_off, 0, 0);
      }
      bsum.x += v.x, bsum.y += v.y, bsum.z += v.z, bsum.w += v.w;
      mg = __builtin_fmaxf(__builtin_fmaxf(mg, fabsf(v.x)), fabsf(v.y));
      mg = __builtin_fmaxf(__builtin_fmaxf(mg, fabsf(v.z)), fabsf(v.w));
    }
    mx = f2_wave_max(mx);
    mg = f2_wave_max(mg);
    if (lane == 0) {
      mxs[parity * 8 + wave] = mx;
      mxs[parity * 8 + 4 + wave] = mg;
    }
  };
  int sx_e = 60, sg_e = 60;  // running scale exponents (the clamp's upper end: any real tile lowers them)
  // (2) after it: running scales (accumulators follow), split, LDS write
  auto stage = [&](WSet& W, int parity) __attribute__((always_inline)) {
    const float4 m0 = *(const float4*)(mxs + parity * 8), m1 = *(const float4*)(mxs + parity * 8 + 4);
    const int ex = f2_scale_exp(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)));
    const int eg = f2_scale_exp(fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    const int nx = ex < sx_e ? ex : sx_e, ng = eg < sg_e ? eg : sg_e;
    if (nx != sx_e || ng != sg_e) {  // (block-uniform, rare) a larger magnitude than any before: the accumulators follow
      const float r = __builtin_ldexpf(1.f, (nx - sx_e) + (ng - sg_e));
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[j] *= r;
      sx_e = nx;
      sg_e = ng;
    }
    const float scx = __builtin_ldexpf(1.f, sx_e), scg = __builtin_ldexpf(1.f, sg_e);
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      const int idx = (int)threadIdx.x + it * 256;
      if (idx < C::NIX) {
        const float4 v = W.x[it];
        unsigned a1, a2, b1, b2;
#ifdef F2W_KO_SPLIT
        a1 = __float_as_uint(v.x) & 0x3fff3fffu, a2 = __float_as_uint(v.y) & 0x3fff3fffu, b1 = __float_as_uint(v.z) & 0x3fff3fffu, b2 = __float_as_uint(v.w) & 0x3fff3fffu;
#else
        f2_split_pair_scaled(v.x, v.y, scx, a1, a2);
        f2_split_pair_scaled(v.z, v.w, scx, b1, b2);
#endif
        unsigned short* p = xl + (idx / C::CVX) * PSX + (idx % C::CVX) * 4;
        *(uint2*)(p) = make_uint2(a1, b1);
        *(uint2*)(p + CIN) = make_uint2(a2, b2);
      }
    }
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int idx = threadIdx.x + it * 256;
      const float4 v = W.g[it];
      unsigned a1, a2, b1, b2;
      f2_split_pair_scaled(v.x, v.y, scg, a1, a2);
      f2_split_pair_scaled(v.z, v.w, scg, b1, b2);
      unsigned short* p = gl + (idx / C::CVG) * PSG + (idx % C::CVG) * 4;
      *(uint2*)(p) = make_uint2(a1, b1);
      *(uint2*)(p + COUT) = make_uint2(a2, b2);
    }
  };

  prefetch(SA, 0);
  if (WPF2) prefetch(SB, 1);
  auto run = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value, T0 = TW * W, T1 = (T0 + TW < C::T) ? T0 + TW : C::T;
    constexpr int NTW = T1 > T0 ? T1 - T0 : 0;
    constexpr int MB0 = T0 / NB, NG = NTW ? (T1 - 1) / NB - MB0 + 1 : 0;
    constexpr int NU = C::KSN * NG, NGD = NG ? NG : 1;
    auto load_fb = [&](int ks, s16x8 (&F)[NP][NB]) __attribute__((always_inline)) {
      const unsigned short* gq = gl + (2 * ks * 16 + 4 * lg + tq) * PSG + tp * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) F[p][nb] = f2_tr_read8(gq + p * COUT + nb * 16, gq + 16 * PSG + p * COUT + nb * 16);
    };
    auto load_fa = [&](int ks, int mb, s16x8 (&F)[NP]) __attribute__((always_inline)) {
      const int tap = CIN == 32 ? mb >> 1 : mb, half = CIN == 32 ? mb & 1 : 0, ky = tap / K, kx = tap - K * ky;
      const unsigned short* xq = xl + ((2 * ks * S + ky) * WX_IC + (4 * lg + tq) * S + kx) * PSX + half * 16 + tp * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p) F[p] = f2_tr_read8(xq + p * CIN, xq + S * WX_IC * PSX + p * CIN);
    };
    int parity = 0;
    auto body = [&](WSet& W, int tile) __attribute__((always_inline)) {
      prep(W, parity);
      __syncthreads();
      stage(W, parity);
      __syncthreads();
      parity ^= 1;
      prefetch(W, tile + (WPF2 ? 2 : 1));
      if (NTW > 0) {
        // fragment reads run FD units (a unit = one x fragment against the gy fragments of its k-step: <= 6 matrix instructions,
        // ~100 cycles) ahead of the matrix instructions that consume them
        constexpr int FD = (F2W_FD > 1 && NGD >= 2) ? 2 : 1, FA = FD + 1;   // (two gy-fragment buffers: a unit two ahead is at most one k-step ahead)
        s16x8 fa[FA][NP], fb[2][NP][NB];
        load_fb(0, fb[0]);
#pragma unroll
        for (int v = 0; v < FD; ++v)
          if (v < NU) {
            if (v > 0 && v % NGD == 0) load_fb(v / NGD, fb[(v / NGD) & 1]);
            load_fa(v / NGD, MB0 + v % NGD, fa[v % FA]);
          }
        f2_static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value;
          constexpr int ks = u / NGD, gi = u % NGD, mb = MB0 + gi;
          int nread = 0;
          if (u + FD < NU) {
            const int ks2 = (u + FD) / NGD, gi2 = (u + FD) % NGD;
            if (gi2 == 0) load_fb(ks2, fb[ks2 & 1]), nread += 2 * NP * NB;
            load_fa(ks2, MB0 + gi2, fa[(u + FD) % FA]);
            nread += 2 * NP;
          }
          constexpr int PA[3] = {1, 0, 0};
          constexpr int PB[3] = {0, 1, 0};
          int nm = 0;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int t = NB * mb + nb;
            if (t >= T0 && t < T1) {
#pragma unroll
              for (int q = 0; q < 3; ++q)
#ifdef F2W_KO_MFMA
                acc[t - T0] = f2_no_mfma(__builtin_bit_cast(f16x8_t, fa[u % FA][PA[q]]), __builtin_bit_cast(f16x8_t, fb[ks & 1][PB[q]][nb]), acc[t - T0]);
#else
                acc[t - T0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[u % FA][PA[q]]),
                                                                     __builtin_bit_cast(f16x8_t, fb[ks & 1][PB[q]][nb]),
                                                                     acc[t - T0], 0, 0, 0);
#endif
              nm += 3;
            }
          }
          const int per = nm ? (nread + nm - 1) / nm : 0;
#pragma unroll
          for (int g = 0; g < 6; ++g)
            if (g < nm) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (per == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              else if (per == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              else if (per == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
              else if (per >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    };
    if constexpr (WPF2) {
      // pairs of tiles through a loop with ONE exit, an odd last tile behind it (see conv_f16x2_kernel: a `break` between the two
      // bodies costs the counted waits)
      int tile = 0;   // (trip index)
      for (int pr = 0; pr < (ntl >> 1); ++pr) {
        body(SA, tile);
        body(SB, tile + 1);
        tile += 2;
      }
      if (ntl & 1) body(SA, tile);
    } else {
      for (int tile = 0; tile < ntl; ++tile) body(SA, tile);
    }
    // partial slab of this workgroup: [m = mb*16 + row][co], scales undone
    const float desc = __builtin_ldexpf(1.f, -(sx_e + sg_e));
    float* out = a.part + (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (C::MB * 16 * COUT);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int t = T0 + j, mb = t / NB, nb = t % NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(mb * 16 + lg * 4 + r) * COUT + nb * 16 + l16] = acc[j][r] * desc;
    }
  };
  switch (wave) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
#ifdef F2_CLK
  if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.y == 0 && blockIdx.z == 0) {
    f2_clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_t0;
    f2_clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
  if (a.bpart) {
    __syncthreads();
    const int vv = threadIdx.x % C::CVG, row = threadIdx.x / C::CVG;
    bred[row * COUT + vv * 4 + 0] = bsum.x;
    bred[row * COUT + vv * 4 + 1] = bsum.y;
    bred[row * COUT + vv * 4 + 2] = bsum.z;
    bred[row * COUT + vv * 4 + 3] = bsum.w;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float sum = 0.f;
      for (int r = 0; r < 256 / C::CVG; ++r) sum += bred[r * COUT + threadIdx.x];
      a.bpart[(long)blockIdx.x * COUT + threadIdx.x] = sum;
    }
  }
}

template <int CIN, int COUT>
static hipError_t f2_wgrad_launch(const WgArgs& a, int inact, long workers, hipStream_t stream) {
  using C = F2WxCfg<CIN, COUT>;
  static_assert(F2W_WPC * C::LDS_BYTES <= 160 * 1024, "workgroups per CU");
  const bool ingn = a.gn_stats != nullptr;
  static bool attr_set[3] = {};
  auto launch = [&](auto kern, int slot) -> hipError_t {
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG(CIN == 32 && COUT == 32 ? "conv_wgrad_f16x2_kernel<32,32>" : CIN == 16 && COUT == 16 ? "conv_wgrad_f16x2_kernel<16,16>" : CIN == 16 ? "conv_wgrad_f16x2_kernel<16,32>" : "conv_wgrad_f16x2_kernel<32,16>");
    hipLaunchKernelGGL(kern, dim3((unsigned)workers), dim3(256), C::LDS_BYTES, stream, a);
    return hipSuccess;
  };
  if (ingn) {
    if constexpr (CIN == COUT) {
      if (inact) return hipErrorInvalidValue;
      return launch(conv_wgrad_f16x2_kernel<CIN, COUT, 0, true>, 2);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (inact == DIS_ACT_SELU) return launch(conv_wgrad_f16x2_kernel<CIN, COUT, DIS_ACT_SELU>, 1);
  if (inact == 0) return launch(conv_wgrad_f16x2_kernel<CIN, COUT, 0>, 0);
  return hipErrorInvalidValue;
}

// slice-pair form (conv2d.hip wgrad_pairs_launch): grid = (workers per pair, pairs, tap-row groups); cob = gy channels per pair;
// (k, stride, kh) as conv2d.hip instantiates the three-term kernel: 3x3 / 5x5 at stride 1 and 2, 7x7 in two groups of 4 tap rows
hipError_t dis_f2_wgrad_pairs_launch(const WgArgs& a, int cob, unsigned workers, unsigned pairs, int k, int stride, int kh,
                                     hipStream_t stream) {
  static bool attr_set[6] = {};
  auto launch = [&](auto kern, int lds, int slot, unsigned ngrp) -> hipError_t {
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG("conv_wgrad_f16x2_kernel slice pairs");
    hipLaunchKernelGGL(kern, dim3(workers, pairs, ngrp), dim3(256), lds, stream, a);
    return hipSuccess;
  };
  if (k == 3 && stride == 1 && kh == 3) {
    if (cob == 32) return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, true>, F2WxCfg<32, 32>::LDS_BYTES, 0, 1);
    if (cob == 16) return launch(conv_wgrad_f16x2_kernel<32, 16, 0, false, true>, F2WxCfg<32, 16>::LDS_BYTES, 1, 1);
    return hipErrorInvalidValue;
  }
  if (cob != 32) return hipErrorInvalidValue;
  if (k == 5 && stride == 1 && kh == 5)
    return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, true, 5, 1, 8, 5>, F2WxCfg<32, 32, 5, 1, 8, 5>::LDS_BYTES, 2, 1);
  if (k == 3 && stride == 2 && kh == 3)
    return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, true, 3, 2, 4, 3>, F2WxCfg<32, 32, 3, 2, 4, 3>::LDS_BYTES, 3, 1);
  if (k == 5 && stride == 2 && kh == 5)
    return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, true, 5, 2, 4, 5>, F2WxCfg<32, 32, 5, 2, 4, 5>::LDS_BYTES, 4, 1);
  if (k == 7 && stride == 1 && kh == 4)
    return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, true, 7, 1, 8, 4>, F2WxCfg<32, 32, 7, 1, 8, 4>::LDS_BYTES, 5, 2);
  return hipErrorInvalidValue;
}

// FuseNet's 4 x 4 stride-2 down convolution (32 -> 32, Block2D3D conv2_1): all 16 taps in one workgroup, 4 x 16 output pixels per
// tile; slab [tap * 32 + ci][co] per worker = conv_wgrad_kernel<32, 32, 4, 4, 2>'s four tap-row splits back to back, so its
// reduce launch finishes the job.
hipError_t dis_f2_wgrad_k4s2_launch(const WgArgs& a, int gnb_act, long workers, hipStream_t stream) {
  using C = F2WxCfg<32, 32, 4, 2, 4, 4>;
  static_assert(C::LDS_BYTES <= 160 * 1024 && C::MB * 16 * 32 == 16 * 32 * 32, "LDS budget / slab size");
  static bool attr_set[3] = {};
  auto launch = [&](auto kern, int slot) -> hipError_t {
    if (!attr_set[slot]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
      attr_set[slot] = true;
    }
    DIS_TAG("conv_wgrad_f16x2_kernel<32,32> 4x4 stride 2");
    hipLaunchKernelGGL(kern, dim3((unsigned)workers), dim3(256), C::LDS_BYTES, stream, a);
    return hipSuccess;
  };
  if (a.gnb_coef) {   // GroupNorm backward applied on load (a.gact: the GroupNorm's input), SELU between conv and GroupNorm or none
    if (!a.gact || !a.gnb_out) return hipErrorInvalidValue;
    return gnb_act == DIS_ACT_SELU ? launch(conv_wgrad_f16x2_kernel<32, 32, DIS_ACT_SELU, false, false, 4, 2, 4, 4, true>, 1)
                                   : launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, false, 4, 2, 4, 4, true>, 2);
  }
  return launch(conv_wgrad_f16x2_kernel<32, 32, 0, false, false, 4, 2, 4, 4>, 0);
}

hipError_t dis_f2_wgrad_launch(const WgArgs& a, int cin, int cout, int inact, long workers, hipStream_t stream) {
  if (a.xscale) return hipErrorInvalidValue;
  if (cin == 32 && cout == 32) return f2_wgrad_launch<32, 32>(a, inact, workers, stream);
  if (cin == 16 && cout == 16) return f2_wgrad_launch<16, 16>(a, inact, workers, stream);
  if (cin == 16 && cout == 32) return f2_wgrad_launch<16, 32>(a, inact, workers, stream);
  if (cin == 32 && cout == 16) return f2_wgrad_launch<32, 16>(a, inact, workers, stream);
  return hipErrorInvalidValue;
}
