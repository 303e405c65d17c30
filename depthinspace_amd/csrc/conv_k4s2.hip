// FuseNet's 4 x 4 stride-2 pad-1 down convolution (32 -> 32, Block2D3D.conv2_1; reference model/multi_frame_networks.py:338-345) on the
// fp16 matrix cores with two-term operands (the arithmetic of conv_f16x2.hip: x 2^s = h1 + h2, three products per MAC, fp32
// accumulate, power-of-two block scales): forward here, the weight gradient is conv_wgrad_f16x2_kernel<..., 4, 2, 4, 4>.
//
// Why a kernel of its own (round 5): on the exact-fp32 MFMA kernel the layer holds 65 KB of fp32 weights + a 49 KB halo in LDS, i.e.
// ONE four-wave workgroup per CU, and 256 16x16x4 matrix instructions per 64 output pixels: 95 us per launch at 1.4 TB/s, matrix-bound
// at one wave per SIMD (profiles/r5v4_kernels.md).  Here
//   * the WEIGHTS LIVE IN REGISTERS: a wave owns 16 output channels (nt = wave & 1) for the whole launch, i.e. 16 taps x 2 planes of
//     A fragments = 128 VGPRs, split once per workgroup from the OIHW tensor (two passes over 8 KB per lane group, L2 hits) - LDS
//     holds pixels only, so an 8 x 16 output tile (18 x 34 input halo, 96 KB in two fp16 planes) fits;
//   * the halo's columns are stored DE-INTERLEAVED (even columns, then odd ones): a tap of a stride-2 window then reads 16
//     consecutive LDS pixels like a stride-1 tap (the 160-byte pixel stride stays conflict-free);
//   * 8 waves: wave = (row pair mp = wave >> 1, channel half nt = wave & 1) computes 2 output rows x 16 columns x 16 channels:
//     16 taps x 2 rows x 3 products = 96 matrix instructions against 64 ds_read_b128 per tile.
// One scale per halo tile (wave maxima through LDS across the staging barrier), one for the weights.  Two barriers per tile, the next
// tile's loads in flight during the matrix phase.
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>

struct K4Args {
  const float* x;     // (n, hin, win, 32)
  const float* w;     // OIHW (32, 32, 4, 4)
  const float* bias;  // may be null
  float* y;           // (n, hout, wout, 32)
  double* stats;      // may be null: (n, 2) sum / sum of squares of the outputs (GroupNorm statistics)
  int n, hin, win, hout, wout;
  int act;
};

#ifndef K4_TR
#define K4_TR 4       // output rows per tile (4: two 4-wave workgroups per CU; 8: one 8-wave workgroup)
#endif
#define K4_NW (K4_TR / 2 * 2)          // waves: (row pairs) x (two channel halves)
#define K4_NT (K4_NW * 64)
#define K4_TC 16
#define K4_IR ((K4_TR - 1) * 2 + 4)   // 18
#define K4_IC ((K4_TC - 1) * 2 + 4)   // 34
#define K4_PS 80                      // 16-bit units per LDS pixel: 2 planes x 32 channels, padded (see F2Cfg::PS)
#define K4_NIT (K4_IR * K4_IC * 8)    // float4 items of a halo
#define K4_NLOAD ((K4_NIT + K4_NT - 1) / K4_NT)
#define K4_X_U16 (K4_IR * K4_IC * K4_PS)
// the buffer also stages the fp32 weight tensor once (32 rows of 516 floats): the larger of the two sizes, in 16-bit units
#define K4_BUF_U16 ((K4_X_U16 * 2 > 32 * 516 * 4 ? K4_X_U16 * 2 : 32 * 516 * 4) / 2)
#define K4_LDS_BYTES (K4_BUF_U16 * 2 + 2 * K4_PS * 2 + 64 + 64 + 64)

static_assert((K4_TR == 4 ? 2 : 1) * K4_LDS_BYTES <= 160 * 1024, "LDS budget");
__global__ __launch_bounds__(K4_NT, K4_TR == 4 ? 2 : 1) void conv_k4s2_f16x2_fwd_kernel(K4Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* xl = smem16;
  unsigned short* pad16 = smem16 + K4_BUF_U16;                       // target of idle threads' LDS writes (2 pixels)
  float* mxs = (float*)(smem16 + K4_BUF_U16 + 2 * K4_PS);            // the tile's wave maxima
  float* wmx = mxs + 16;                                             // the weights' eight wave maxima
  double* red = (double*)(wmx + 16);                                 // statistics reduction (8 doubles)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int nt = wave & 1, mp = wave >> 1;
  const int tiles_x = (a.wout + K4_TC - 1) / K4_TC, tiles_y = (a.hout + K4_TR - 1) / K4_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);

  // ---- halo items of this thread: (pixel, 4-channel chunk) -> byte offset inside the halo window, LDS position (columns de-interleaved)
  float4 pre[K4_NLOAD];
  int it_c[K4_NLOAD], it_off[K4_NLOAD], it_lds[K4_NLOAD];
#pragma unroll
  for (int it = 0; it < K4_NLOAD; ++it) {
    const int idx = (int)threadIdx.x + it * K4_NT;
    const int ch = idx & 7, pix = idx >> 3;
    const int r = pix / K4_IC, c = pix % K4_IC;
    const bool ok = idx < K4_NIT;
    it_c[it] = ok ? c : 0x40000000;   // (never in range: zeros)
    it_off[it] = ((r * a.win + c) * 32 + ch * 4) * 4;
    it_lds[it] = ok ? (r * K4_IC + (c >> 1) + (c & 1) * (K4_IC / 2)) * K4_PS + ch * 4 : K4_BUF_U16 + (idx & 1) * K4_PS;
  }
  (void)pad16;
  const unsigned x_bytes = (unsigned)a.hin * a.win * 128u;
  auto prefetch = [&](int tile, bool live) __attribute__((always_inline)) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * (2 * K4_TR) - 1, ix0 = tx * (2 * K4_TC) - 1;
    const float* xb = a.x + (long)n * a.hin * a.win * 32;
    const int off0 = (iy0 * a.win + ix0) * 128;
    const unsigned bytes = live ? x_bytes : 0u;
#pragma unroll
    for (int it = 0; it < K4_NLOAD; ++it) {
      // (rows above / below the sample leave the sample's byte range by themselves; only the column is tested)
      const unsigned off = (unsigned)(ix0 + it_c[it]) < (unsigned)a.win ? (unsigned)(off0 + it_off[it]) : BX_OOB;
      pre[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(xb, bytes), off, 0, 0));
    }
  };
  int tile = t_lo + rank;
  prefetch(tile < t_hi ? tile : 0, tile < t_hi);

  // ---- weights -> registers: wave-resident A fragments of its 16 output channels, [tap][plane]; lane (li, lg) holds output channel
  // nt * 16 + li, input channels lg * 8 .. + 7.  Pass 1: the tensor's largest magnitude; pass 2: scale, split.
  s16x8 wf[16][2];
  int sw_e;
  {
    // coalesced copy of the 64 KB tensor into the (still unused) halo buffer, rows of 512 floats padded by 4 (the fragment gather
    // below walks input channels 16 floats apart: the pad keeps its 8 x 16 lanes off each other's banks), all loads of a thread in
    // flight together; 256 scattered global loads per lane took ~10 us of a 53-us launch
    float* ws = (float*)xl;
    constexpr int NW4 = 32 * 32 * 16 / 4, WL = (NW4 + K4_NT - 1) / K4_NT;
    float4 wv[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int q = (int)threadIdx.x + i * K4_NT;
      wv[i] = q < NW4 ? ((const float4*)a.w)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int q = (int)threadIdx.x + i * K4_NT;
      if (q < NW4) *(float4*)(ws + (q / 128) * 516 + (q % 128) * 4) = wv[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(wv[i].x), fabsf(wv[i].y))), fmaxf(fabsf(wv[i].z), fabsf(wv[i].w)));
    }
    m = f2_wave_max(m);
    if (lane == 0) wmx[wave] = m;
    __syncthreads();
    float mm = 0.f;
#pragma unroll
    for (int i = 0; i < K4_NW; ++i) mm = fmaxf(mm, wmx[i]);
    sw_e = f2_scale_exp(mm);
    const float sw = __builtin_ldexpf(1.f, sw_e);
    const float* wp = ws + (nt * 16 + li) * 516 + (lg * 8) * 16;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      unsigned p0[4], p1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f2_split_pair(wp[(2 * j) * 16 + tap] * sw, wp[(2 * j + 1) * 16 + tap] * sw, p0[j], p1[j]);
      wf[tap][0] = __builtin_bit_cast(s16x8, make_uint4(p0[0], p0[1], p0[2], p0[3]));
      wf[tap][1] = __builtin_bit_cast(s16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));
    }
    __syncthreads();   // (the halo buffer is free for the first tile)
  }
  float4 bias_v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bias_v = *(const float4*)(a.bias + nt * 16 + lg * 4);
  // LDS position of this lane's pixel fragment for tap (0, 0) of its first output row: halo row 2 (2 mp), column li
  const int xa_lane = ((4 * mp) * K4_IC + li) * K4_PS + lg * 8;
  double s1 = 0.0, s2 = 0.0;
  int stat_n = -1;
  auto stats_flush = [&]() __attribute__((always_inline)) {
    const double r1 = block_sum_d(s1, red);
    const double r2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
      atomic_add_d(a.stats + 2 * stat_n, r1);
      atomic_add_d(a.stats + 2 * stat_n + 1, r2);
    }
    s1 = 0.0;
    s2 = 0.0;
  };

  while (tile < t_hi) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    if (a.stats && n != stat_n) {   // (block-uniform)
      if (stat_n >= 0) stats_flush();
      stat_n = n;
    }
    // ---- this tile's items have landed: block maximum, scale, split, LDS
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < K4_NLOAD; ++it) {
      const float4 v = pre[it];
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.x)), fabsf(v.y));
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.z)), fabsf(v.w));
    }
    m = f2_wave_max(m);
    if (lane == 0) mxs[wave] = m;
    __syncthreads();   // (also: every wave has finished reading the previous tile's halo)
    float mt_ = 0.f;
#pragma unroll
    for (int i = 0; i < K4_NW; ++i) mt_ = fmaxf(mt_, mxs[i]);
    const int sx_e = f2_scale_exp(mt_);
    const float sc = __builtin_ldexpf(1.f, sx_e);
#pragma unroll
    for (int it = 0; it < K4_NLOAD; ++it) {
      const float4 v = pre[it];
      unsigned a1, a2, b1, b2;
      f2_split_pair_scaled(v.x, v.y, sc, a1, a2);
      f2_split_pair_scaled(v.z, v.w, sc, b1, b2);
      unsigned short* p = xl + it_lds[it];
      *(uint2*)(p) = make_uint2(a1, b1);
      *(uint2*)(p + 32) = make_uint2(a2, b2);
    }
    __syncthreads();
    prefetch(tile + per < t_hi ? tile + per : 0, tile + per < t_hi);   // in flight during the matrix phase

    // ---- matrix phase: 16 taps x 2 output rows x 3 products; products (pixel plane, weight plane), smallest terms first
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
    s16x8 xf[2][2][2];   // [buffer][row][plane]
    auto load_x = [&](int tap, s16x8 (&F)[2][2]) __attribute__((always_inline)) {
      const int ky = tap >> 2, kx = tap & 3;
      const unsigned short* q = xl + xa_lane + (ky * K4_IC + (kx >> 1) + (kx & 1) * (K4_IC / 2)) * K4_PS;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) F[mt][p] = *(const s16x8*)(q + mt * (2 * K4_IC * K4_PS) + p * 32);
    };
    load_x(0, xf[0]);
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      if (tap + 1 < 16) load_x(tap + 1, xf[(tap + 1) & 1]);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[tap][PB[q]]),
                                                          __builtin_bit_cast(f16x8_t, xf[tap & 1][mt][PA[q]]), acc[mt], 0, 0, 0);
    }

    // ---- epilogue: undo the two scales (exact), bias, activation, store, statistics
    const float desc = __builtin_ldexpf(1.f, -(sx_e + sw_e));
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int oy = ty * K4_TR + 2 * mp + mt, ox = tx * K4_TC + li;
      float4 v = make_float4(acc[mt][0] * desc + bias_v.x, acc[mt][1] * desc + bias_v.y, acc[mt][2] * desc + bias_v.z,
                             acc[mt][3] * desc + bias_v.w);
      v = make_float4(act_apply(v.x, a.act), act_apply(v.y, a.act), act_apply(v.z, a.act), act_apply(v.w, a.act));
      if (oy < a.hout && ox < a.wout) {
        *(float4*)(a.y + (((long)n * a.hout + oy) * a.wout + ox) * 32 + nt * 16 + lg * 4) = v;
        t1 += (v.x + v.y) + (v.z + v.w);
        t2 = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.w, v.w, t2))));
      }
    }
    s1 += (double)t1;
    s2 += (double)t2;
    tile += per;
  }
  if (a.stats && stat_n >= 0) stats_flush();
}

// launch: one persistent workgroup per CU (96 KB of LDS, 8 waves)
hipError_t dis_k4s2_fwd_launch(const float* x, const float* w_oihw, const float* bias, float* y, double* stats, int n, int hin, int win,
                               int act, long grid_cus, hipStream_t stream) {
  K4Args a;
  a.x = x; a.w = w_oihw; a.bias = bias; a.y = y; a.stats = stats;
  a.n = n; a.hin = hin; a.win = win; a.hout = (hin + 2 - 4) / 2 + 1; a.wout = (win + 2 - 4) / 2 + 1;
  a.act = act;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_k4s2_f16x2_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, K4_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const long ntiles = (long)n * ((a.hout + K4_TR - 1) / K4_TR) * ((a.wout + K4_TC - 1) / K4_TC);
  long grid = grid_cus * (K4_TR == 4 ? 2 : 1);
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  DIS_TAG("conv_k4s2_f16x2_fwd_kernel");
  hipLaunchKernelGGL(conv_k4s2_f16x2_fwd_kernel, dim3((unsigned)grid), dim3(K4_NT), K4_LDS_BYTES, stream, a);
  return hipSuccess;
}

static int k4_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
bool dis_f2_enabled();   // conv_f16x2.hip: the process-wide operand split (dis_set_conv_split)

/* y = act(conv2d(x, w, stride 2, pad 1) + bias) for FuseNet's 4 x 4 down convolution, 32 -> 32 channels, w OIHW (32, 32, 4, 4) as the
 * module stores it (no packing launch), x (n, hin, win, 32) nhwc, y (n, hin / 2, win / 2, 32) (hin, win even or odd: (h + 2 - 4) / 2
 * + 1); act NONE / SELU / ReLU; stats (n, 2) fp64 sum / sum of squares of y, accumulated (caller zeroes), may be NULL.  Two-term fp16
 * operands (DIS_ERR_UNSUPPORTED under dis_set_conv_split(0): dis_conv2d_fwd serves the shape on the exact-fp32 kernel). */
extern "C" int dis_conv2d_fwd_k4s2_f16x2(const float* x, const float* w_oihw, const float* bias, float* y, double* stats, int n, int hin,
                                         int win, int act, void* stream) {
  if (!x || !w_oihw || !y) return DIS_ERR_NULL;
  if (n <= 0 || hin < 2 || win < 2) return DIS_ERR_BAD_SHAPE;
  if (act < 0 || act > DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  if (!dis_f2_enabled()) return DIS_ERR_UNSUPPORTED;
  if ((long)hin * win * 128 >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;   // (31-bit byte offsets inside a sample)
  hipError_t e = dis_k4s2_fwd_launch(x, w_oihw, bias, y, stats, n, hin, win, act, k4_num_cus(), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// Input gradient of the same convolution: gx[2 i + py][2 j + px] = sum over the class's 2 x 2 taps of gy[i + dy][j + dx] W^T[tap] - the
// four parity classes of the transposed convolution FROM ONE gy HALO TILE IN ONE LAUNCH (the exact-fp32 path ran four launches of a
// 2 x 2-tap kernel and four weight-packing launches: 117 us per call).  A tile is 4 x 16 gy pixels (halo 6 x 18, 17 KB in two fp16
// planes) and writes 8 x 32 gx pixels; wave = (gy row pair mp, input-channel half nt) with the 16 taps' W^T fragments of its 16
// channels resident in registers (A operand: rows = input channels, K = the 32 output channels); per class 4 taps x 2 rows x 3
// products, one epilogue (optionally adding to gx: the layer's input has a second consumer).  Two 4-wave workgroups per CU.
// ------------------------------------------------------------------------------------------------
struct K4DArgs {
  const float* gy;  // (n, hout, wout, 32)
  const float* w;   // OIHW (32, 32, 4, 4)
  float* gx;        // (n, 2 hout, 2 wout, 32)
  int n, hout, wout;
  int accum;
};
#define K4D_TR 4
#define K4D_IR (K4D_TR + 2)
#define K4D_IC (K4_TC + 2)
#define K4D_NT 256
#define K4D_NIT (K4D_IR * K4D_IC * 8)
#define K4D_NLOAD ((K4D_NIT + K4D_NT - 1) / K4D_NT)
#define K4D_X_U16 (K4D_IR * K4D_IC * K4_PS)
#define K4D_BUF_U16 (32 * 516 * 4 / 2)   // the weight staging area (66 KB) is the larger use of the buffer
#define K4D_LDS_BYTES (K4D_BUF_U16 * 2 + 2 * K4_PS * 2 + 64 + 64)
static_assert(K4D_X_U16 <= K4D_BUF_U16 && 2 * K4D_LDS_BYTES <= 160 * 1024, "LDS budget");

__global__ __launch_bounds__(K4D_NT, 2) void conv_k4s2_f16x2_dgrad_kernel(K4DArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* gl = smem16;
  float* mxs = (float*)(smem16 + K4D_BUF_U16 + 2 * K4_PS);
  float* wmx = mxs + 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int nt = wave & 1, mp = wave >> 1;
  const int tiles_x = (a.wout + K4_TC - 1) / K4_TC, tiles_y = (a.hout + K4D_TR - 1) / K4D_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);
  const int hin = 2 * a.hout, win = 2 * a.wout;

  float4 pre[K4D_NLOAD];
  int it_c[K4D_NLOAD], it_off[K4D_NLOAD], it_lds[K4D_NLOAD];
#pragma unroll
  for (int it = 0; it < K4D_NLOAD; ++it) {
    const int idx = (int)threadIdx.x + it * K4D_NT;
    const int ch = idx & 7, pix = idx >> 3;
    const int r = pix / K4D_IC, c = pix % K4D_IC;
    const bool ok = idx < K4D_NIT;
    it_c[it] = ok ? c : 0x40000000;
    it_off[it] = ((r * a.wout + c) * 32 + ch * 4) * 4;
    it_lds[it] = ok ? (r * K4D_IC + c) * K4_PS + ch * 4 : K4D_BUF_U16 + (idx & 1) * K4_PS;
  }
  const unsigned g_bytes = (unsigned)a.hout * a.wout * 128u;
  auto prefetch = [&](int tile, bool live) __attribute__((always_inline)) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * K4D_TR - 1, ix0 = tx * K4_TC - 1;
    const float* gb = a.gy + (long)n * a.hout * a.wout * 32;
    const int off0 = (iy0 * a.wout + ix0) * 128;
    const unsigned bytes = live ? g_bytes : 0u;
#pragma unroll
    for (int it = 0; it < K4D_NLOAD; ++it) {
      const unsigned off = (unsigned)(ix0 + it_c[it]) < (unsigned)a.wout ? (unsigned)(off0 + it_off[it]) : BX_OOB;
      pre[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(gb, bytes), off, 0, 0));
    }
  };
  int tile = t_lo + rank;
  prefetch(tile < t_hi ? tile : 0, tile < t_hi);

  // ---- W^T fragments of this wave's 16 input channels -> registers: lane (li, lg) holds input channel nt * 16 + li, output channels
  // lg * 8 .. + 7 of every tap (staged through LDS like the forward kernel's)
  s16x8 wf[16][2];
  int sw_e;
  {
    float* ws = (float*)gl;
    constexpr int NW4 = 32 * 32 * 16 / 4, WL = (NW4 + K4D_NT - 1) / K4D_NT;
    float m = 0.f;
#pragma unroll 4
    for (int i = 0; i < WL; ++i) {
      const int q = (int)threadIdx.x + i * K4D_NT;
      const float4 v = q < NW4 ? ((const float4*)a.w)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < NW4) *(float4*)(ws + (q / 128) * 516 + (q % 128) * 4) = v;
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = f2_wave_max(m);
    if (lane == 0) wmx[wave] = m;
    __syncthreads();
    sw_e = f2_scale_exp(fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
    const float sw = __builtin_ldexpf(1.f, sw_e);
    const float* wp = ws + (lg * 8) * 516 + (nt * 16 + li) * 16;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      unsigned p0[4], p1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f2_split_pair(wp[(2 * j) * 516 + tap] * sw, wp[(2 * j + 1) * 516 + tap] * sw, p0[j], p1[j]);
      wf[tap][0] = __builtin_bit_cast(s16x8, make_uint4(p0[0], p0[1], p0[2], p0[3]));
      wf[tap][1] = __builtin_bit_cast(s16x8, make_uint4(p1[0], p1[1], p1[2], p1[3]));
    }
    __syncthreads();
  }
  // this lane's gy fragment for halo row 2 mp, halo column li
  const int ga_lane = ((2 * mp) * K4D_IC + li) * K4_PS + lg * 8;

  while (tile < t_hi) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < K4D_NLOAD; ++it) {
      const float4 v = pre[it];
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.x)), fabsf(v.y));
      m = __builtin_fmaxf(__builtin_fmaxf(m, fabsf(v.z)), fabsf(v.w));
    }
    m = f2_wave_max(m);
    if (lane == 0) mxs[wave] = m;
    __syncthreads();
    const int sx_e = f2_scale_exp(fmaxf(fmaxf(mxs[0], mxs[1]), fmaxf(mxs[2], mxs[3])));
    const float sc = __builtin_ldexpf(1.f, sx_e);
#pragma unroll
    for (int it = 0; it < K4D_NLOAD; ++it) {
      const float4 v = pre[it];
      unsigned a1, a2, b1, b2;
      f2_split_pair_scaled(v.x, v.y, sc, a1, a2);
      f2_split_pair_scaled(v.z, v.w, sc, b1, b2);
      unsigned short* p = gl + it_lds[it];
      *(uint2*)(p) = make_uint2(a1, b1);
      *(uint2*)(p + 32) = make_uint2(a2, b2);
    }
    __syncthreads();
    prefetch(tile + per < t_hi ? tile + per : 0, tile + per < t_hi);

    const float desc = __builtin_ldexpf(1.f, -(sx_e + sw_e));
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int py = cls >> 1, px = cls & 1;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // the class's taps: ky in {1, 3} (py = 0) or {0, 2} (py = 1); gy row i + dy with dy = (4 - ky) >> 1 - 1 -> halo row offset (4 - ky) >> 1
        const int ky = (py ? 0 : 1) + 2 * (t >> 1), kx = (px ? 0 : 1) + 2 * (t & 1);
        const int ro = (4 - ky) >> 1, co = (4 - kx) >> 1;
        const unsigned short* q = gl + ga_lane + (ro * K4D_IC + co) * K4_PS;
        s16x8 gf[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p) gf[mt][p] = *(const s16x8*)(q + mt * (K4D_IC * K4_PS) + p * 32);
#pragma unroll
        for (int qq = 0; qq < 3; ++qq)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[ky * 4 + kx][PB[qq]]),
                                                            __builtin_bit_cast(f16x8_t, gf[mt][PA[qq]]), acc[mt], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int oy = ty * K4D_TR + 2 * mp + mt, ox = tx * K4_TC + li;
        if (oy < a.hout && ox < a.wout) {
          float4* dst = (float4*)(a.gx + (((long)n * hin + (2 * oy + py)) * win + (2 * ox + px)) * 32 + nt * 16 + lg * 4);
          float4 v = make_float4(acc[mt][0] * desc, acc[mt][1] * desc, acc[mt][2] * desc, acc[mt][3] * desc);
          if (a.accum) {
            const float4 o = *dst;
            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
          }
          *dst = v;
        }
      }
    }
    tile += per;
  }
}

/* gx (+)= conv_transpose of gy: the input gradient of FuseNet's 4 x 4 stride-2 pad-1 convolution (32 -> 32), gy (n, hin / 2, win / 2, 32),
 * gx (n, hin, win, 32), hin and win even, w OIHW (32, 32, 4, 4) unpacked; accumulate != 0 adds to gx.  One launch for the four parity
 * classes, two-term fp16 operands (DIS_ERR_UNSUPPORTED under dis_set_conv_split(0): dis_conv2d_dgrad_strided remains). */
extern "C" int dis_conv2d_dgrad_k4s2_f16x2(const float* gy, const float* w_oihw, float* gx, int n, int hin, int win, int accumulate,
                                           void* stream) {
  if (!gy || !w_oihw || !gx) return DIS_ERR_NULL;
  if (n <= 0 || hin < 2 || win < 2) return DIS_ERR_BAD_SHAPE;
  if ((hin & 1) || (win & 1) || !dis_f2_enabled()) return DIS_ERR_UNSUPPORTED;
  if ((long)hin * win * 128 >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;
  K4DArgs a;
  a.gy = gy; a.w = w_oihw; a.gx = gx; a.n = n; a.hout = hin / 2; a.wout = win / 2; a.accum = accumulate ? 1 : 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_k4s2_f16x2_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, K4D_LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long ntiles = (long)n * ((a.hout + K4D_TR - 1) / K4D_TR) * ((a.wout + K4_TC - 1) / K4_TC);
  long grid = 2L * k4_num_cus();
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  DIS_TAG("conv_k4s2_f16x2_dgrad_kernel");
  hipLaunchKernelGGL(conv_k4s2_f16x2_dgrad_kernel, dim3((unsigned)grid), dim3(K4D_NT), K4D_LDS_BYTES, (hipStream_t)stream, a);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
