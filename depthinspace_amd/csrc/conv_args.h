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
};

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

// fp16x2 kernels (conv_f16x2.hip): launched from conv2d.hip's dispatchers.  Return hipErrorInvalidValue when no instance
// exists for the configuration (the caller then takes the bf16x3 kernel).
hipError_t dis_f2_conv_launch(const ConvArgs& a, int cin, int cout, bool stats, int inact, long grid, hipStream_t stream);
hipError_t dis_f2_wgrad_launch(const WgArgs& a, int cin, int cout, int inact, long workers, hipStream_t stream);
bool dis_f2_enabled();
