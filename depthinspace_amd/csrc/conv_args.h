// Argument blocks of the convolution kernels (shared by conv2d.hip and conv_f16x2.hip).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;
  const float* w;     // packed
  const float* bias;  // may be null
  float* y;
  double* stats;  // may be null: (n,2)
  int n, hin, win;
  int hv, wv;          // virtual output grid the tiles run over
  int pad_y, pad_x;
  int hf, wf;          // full output tensor dims
  int osy, ooy, osx, oox;  // output pixel = (vy*osy+ooy, vx*osx+oox)
  int act;
  int accum;  // epilogue adds the previous contents of y (before the activation): y = act(y_old + conv + bias)
  const float* xscale;  // optional (n,hin,win,NCHUNK): input pixel x chunk multiplier applied when the halo is staged
  const float* yscale;  // optional (n,hf,wf,COUT/32): output pixel x 32-channel-group multiplier (before bias/accum)
  int wmode;  // bf16x3 kernel only: -1 = w is packed; 0 / 1 = w is OIHW fp32, split in the kernel (forward / input gradient)
  int w_o, w_i;  // ... and its leading dims
  int w_rs;      // ... and the floats between its rows (w_i * 9 if dense; larger for a slice w[:, a:b] of a wider weight)
  const float* xact;  // bf16x3 kernel, INACT instances: activation OUTPUT at x's positions; x is multiplied by act'(xact)
  // bf16x3 kernel, GEN instances (32-channel slices of wider tensors, dis_convg_run): x / y point at the slice's first
  // channel, a pixel occupies ldx / ldy floats, cx / cy channels of the slice exist (the rest load zeros / are not
  // stored), x_sub / y_sub = floats between the tensor's start and the slice's (for the buffer range), nbias = bias
  // entries that exist
  int ldx, ldy, cx, cy, x_sub, y_sub, nbias;
  int wtap0, wtap_step;  // tap-row instances (1 x 7 window of a 7x7 weight): see the weight prologue
  // bf16x3 kernel, INGN instances: x is the PRE-GroupNorm tensor (output of the producing conv + activation); the affine map
  // of GroupNorm(1 group) - per sample rstd * gamma_c, beta_c - rstd * gamma_c * mean - is applied while the halo is staged,
  // the zero padding stays zero.  gn_stats (n, 2) fp64 sum / sum of squares, as dis_gn_apply takes them.
  const double* gn_stats;
  const float* gn_gamma;
  const float* gn_beta;
  float gn_eps;
  // f16x2 kernel, EPIAB instance (input gradient of the conv that consumes a GroupNorm output, conv2d_gn_in): besides the
  // gradient g it writes, the epilogue sums g and g * x per (sample, channel), x = the GroupNorm's INPUT at the same positions
  // (ab_x, shaped like y); every workgroup leaves its sums of a sample in its own slot ab_out[(sample * ab_slots + block) * 2 C ..]
  // (fp64: C sums of g, C sums of g * x; the buffer arrives zeroed).  These are all the GroupNorm backward needs besides g.
  const float* ab_x;
  double* ab_out;
  int ab_slots;  // slots per sample in ab_out (>= gridDim.x)
  // ... EPIACT form (the accumulating input gradient whose result is the gradient wrt the OUTPUT of act(GroupNorm(.) + residual),
  // ResNetBlock chains): the finished value is multiplied by act'(ab_act_y) before it is stored and summed (ab_act_y: that output)
  const float* ab_act_y;
  // f16x2 kernel, INCOEF instances (round 5): x is the gradient g wrt the OUTPUT of a GroupNorm whose input q = xact was this conv's
  // own (activated) output: the kernel stages act'(q) (g k1_c + q kx + k0) - the elementwise pass of the GroupNorm backward from the
  // per-sample coefficients gnb_coef (n, CIN + 2) of dis_gn_bwd_coef - and stores those values for the pixels a tile OWNS (its halo
  // minus the rim) to gnb_out (shaped like x): the weight-gradient launch of the same layer reads them from there.
  const float* gnb_coef;
  float* gnb_out;
  int gnb_act;  // conv_fwd_kernel<..., GNB>: the activation between the conv and the GroupNorm (the f16x2 kernel takes it as INACT)
};

struct WgArgs {
  const float* x;
  const float* gy;
  float* part;   // [worker][chunk][split][PART]
  float* bpart;  // [worker][COUT] bias partial sums (written by chunk 0 / split 0 workgroups), may be null
  int n, hin, win, hout, wout, pad;
  const float* xscale;  // optional (n,hin,win,NCHUNK) multiplier of x (see ConvArgs::xscale)
  const float* gact;    // bf16x3 kernel, INACT instances: activation OUTPUT at gy's positions; gy is multiplied by act'(gact)
  // bf16x3 kernel, GEN instance (channel-slice pairs of a wide layer, dis_convg_wgrad): a pixel of x / gy occupies
  // ldx / ldg floats, the layer's channels start at xoff / goff and there are cx / cg of them; blockIdx.y = gb * npx + cb
  // selects x channels [32 cb, 32 cb + 32) and gy channels [32 gb, 32 gb + 32)
  int ldx, xoff, cx, ldg, goff, cg, npx;
  // bf16x3 kernel, INGN instances: x is staged as GroupNorm(x) (see ConvArgs::gn_stats)
  const double* gn_stats;
  const float* gn_gamma;
  const float* gn_beta;
  float gn_eps;
  // f16x2 kernel, INCOEF instance (round 5; FuseNet's 4 x 4 stride-2 conv): gy is the gradient wrt the OUTPUT of the GroupNorm behind
  // the conv, gact the GroupNorm's input (the conv's activated output): staged as act'(gact) (gy k1_c + gact kx + k0) with the
  // coefficients gnb_coef (n, COUT + 2) of dis_gn_bwd_coef, and stored to gnb_out (every gy pixel belongs to one tile) for the
  // input-gradient launches
  const float* gnb_coef;
  float* gnb_out;
  // conv_wgrad_kernel<..., GACT = true> (round 6; the 4 -> 16 stems, whose input gradient is never needed): gy is staged as
  // gy * act'(gact), gact = the conv's activated output - the separate dis_act_bwd pass (read gy, y; write gpre) does not exist
  int gact_act;
};

// conv_bwd_fused.hip: input gradient + weight gradient of a 3x3 stride-1 pad-1 conv C -> C in one launch
struct FbArgs {
  ConvArgs c;           // the input-gradient launch exactly as conv_f16x2_kernel takes it (x = gy / g, y = gx, xact, gnb_*, ab_*)
  const float* wx;      // the conv's INPUT (weight-gradient operand), shaped like c.y; XSRC != 0: taken from c.ab_x / c.ab_act_y instead
  const double* wx_gn_stats;   // XGN: wx is staged as GroupNorm(wx) (n, 2) sums, as dis_gn_apply takes them
  const float* wx_gn_gamma;
  const float* wx_gn_beta;
  float wx_gn_eps;
  float* part;    // weight-gradient slabs [workgroup][9 * C * C], element [(tap * C + ci) * C + co] (conv_wgrad_f16x2_kernel's layout)
  float* bpart;   // bias-gradient partials [workgroup][C], may be null
  float* spill;   // [workgroup][wave][36 * 256]: where a wave parks its dW accumulators when the dW exponent has to move (rare)
};
hipError_t dis_fb_launch(const FbArgs& f, int inact, bool xgn, int xsrc, long grid, hipStream_t stream);

// Weight prologue of the LDS-resident-weight kernels (512 threads): copy an OIHW block - w_o <= 32 rows of `row` <= 288 floats,
// rows w_rs floats apart in memory - into LDS rows padded by one float.  ALL of a thread's loads are issued before the first
// is used: the plain `for (i = tid; i < n; i += 512) lds[..] = w[..]` form waits for every load in turn, 18 memory round trips
// per launch (~15 k cycles: 6 - 10 % of a 3x3 launch at core resolution, measured with the phase stamps of round 3).
// Returns the largest magnitude this thread saw.
__device__ __forceinline__ float dis_copy_w_rows(const float* w, int w_o, int row, int w_rs, float* ws) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned bytes = (unsigned)((w_o - 1) * w_rs + row) * 4u;
  float v[4][5];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int r = wave + 8 * rr, j = lane + 64 * jj;
      const unsigned off = (r < w_o && j < row) ? (unsigned)(r * w_rs + j) * 4u : BX_OOB;
      v[rr][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bx_rsrc(w, bytes), off, 0, 0));
    }
  float m = 0.f;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int jj = 0; jj < 5; ++jj) {
      const int r = wave + 8 * rr, j = lane + 64 * jj;
      if (r < w_o && j < row) ws[r * (row + 1) + j] = v[rr][jj];
      m = fmaxf(m, fabsf(v[rr][jj]));
    }
  return m;
}

// ------------------------------------------------------------------------------------------------
// two-term fp16 split ("f16x2": conv_f16x2.hip, conv_gen.hip): x * 2^s = h1 + h2, 11 + 11 significant bits
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// two fp32 values (already scaled) -> two packed fp16 pairs: h1 = RN(v), h2 = RN(v - h1)
__device__ __forceinline__ void f2_split_pair(float x, float y, unsigned& p1, unsigned& p2) {
  const f32x2 v = {x, y};
  const f16x2_t h1 = __builtin_convertvector(v, f16x2_t);
  const f32x2 r = v - __builtin_convertvector(h1, f32x2);
  const f16x2_t h2 = __builtin_convertvector(r, f16x2_t);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
}
// the same for x * sc, y * sc with sc a power of two (the products are exact): the remainder x * sc - h1 as ONE mixed-precision
// fused multiply-add per value (v_fma_mix_f32 reads h1's halves in place) instead of multiply, convert back, subtract
__device__ __forceinline__ void f2_split_pair_scaled(float x, float y, float sc, unsigned& p1, unsigned& p2) {
  const f32x2 v = {x * sc, y * sc};
  const f16x2_t h1 = __builtin_convertvector(v, f16x2_t);
  const f32x2 r = {__builtin_fmaf(x, sc, -(float)h1[0]), __builtin_fmaf(y, sc, -(float)h1[1])};
  const f16x2_t h2 = __builtin_convertvector(r, f16x2_t);
  p1 = __builtin_bit_cast(unsigned, h1);
  p2 = __builtin_bit_cast(unsigned, h2);
}

// maximum of a non-negative value over the wave (DPP row operations, no LDS traffic); valid in every lane's return value
__device__ __forceinline__ float f2_wave_max(float m) {
  int v = __float_as_int(m);  // non-negative floats order like their bit patterns
#define F2_DPP(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, true))
  F2_DPP(0xB1, 0xf);   // quad_perm [1,0,3,2]
  F2_DPP(0x4E, 0xf);   // quad_perm [2,3,0,1]
  F2_DPP(0x124, 0xf);  // row_ror:4
  F2_DPP(0x128, 0xf);  // row_ror:8   -> every lane holds its row's maximum
  F2_DPP(0x142, 0xa);  // row_bcast:15 into rows 1, 3
  F2_DPP(0x143, 0xc);  // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's maximum
#undef F2_DPP
  return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

// sum of a double over the wave (DPP row operations on the two halves, no LDS traffic); valid in lane 63
__device__ __forceinline__ double f2_wave_sum_d(double v) {
#define F2_DPPD(ctrl, rmask)                                                                              \
  {                                                                                                       \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, true);             \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, true);             \
    v += __hiloint2double(hi, lo);                                                                        \
  }
  F2_DPPD(0xB1, 0xf) F2_DPPD(0x4E, 0xf) F2_DPPD(0x124, 0xf) F2_DPPD(0x128, 0xf) F2_DPPD(0x142, 0xa) F2_DPPD(0x143, 0xc)
#undef F2_DPPD
  return v;
}

// power-of-two scale exponent that brings a block's largest magnitude into [2^14, 2^15) (fp16's largest binade is 2^15);
// clamped so that the product of two scales and its inverse stay representable in fp32
__device__ __forceinline__ int f2_scale_exp(float m) {
  int e = 14 - __builtin_amdgcn_frexp_expf(m) + 1;  // frexp: m = f * 2^e, f in [0.5, 1)  ->  m in [2^(e-1), 2^e)
  e = m > 0.f ? e : 0;
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}

// fp16x2 kernels (conv_f16x2.hip): launched from conv2d.hip's dispatchers.  Return hipErrorInvalidValue when no instance
// exists for the configuration (the caller then takes the bf16x3 kernel).
hipError_t dis_f2_conv_launch(const ConvArgs& a, int cin, int cout, bool stats, int inact, long grid, hipStream_t stream);
hipError_t dis_f2_wgrad_launch(const WgArgs& a, int cin, int cout, int inact, long workers, hipStream_t stream);
hipError_t dis_f2_conv_gen_launch(const ConvArgs& a, long grid, hipStream_t stream);  // 32 x 32 channel slices (DispNetS)
hipError_t dis_f2_wgrad_pairs_launch(const WgArgs& a, int cob, unsigned workers, unsigned pairs, int k, int stride, int kh,
                                     hipStream_t stream);
hipError_t dis_f2_wgrad_k4s2_launch(const WgArgs& a, int gnb_act, long workers, hipStream_t stream);   // 32 -> 32, 4 x 4, stride 2
bool dis_f2_enabled();
int dis_f2_wgrad_wpc();   // workgroups per CU the two-term weight-gradient kernel is built for
