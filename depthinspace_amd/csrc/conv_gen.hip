// General implicit-GEMM convolution family on the CDNA4 matrix cores (fp32 in / fp32 accumulate,
// v_mfma_f32_16x16x4_f32) for the DIS-SF encoder-decoder (DispNetS, reference model/networks.py:170-295):
// channel counts 2..1024, kernels 7/5/3, stride 1/2, ConvTranspose2d(k3,s2,p1,op1) with crop_like, and all
// their input / weight / bias gradients.
//
// Unlike conv2d.hip (whole layer's weights resident in LDS, specialised per FuseNet shape) the weights here
// do not fit LDS (up to 1024x512x9 floats), so both operands are streamed:
//
//   forward-like kernel  D[pixel][co] += A[pixel][(tap,ci)] * B[(tap,ci)][co]
//     tile 128 pixels x BN couts per workgroup (4 waves x 32 pixels), k-step = 16 channels of one tap,
//     A gathered from nhwc global memory by per-thread (tap-shifted) addresses, B read from a pre-packed
//     fragment-ordered weight image; both double-buffered in LDS with register prefetch (one barrier per
//     k-step).  Lane group g of a wave owns channels [4g,4g+4) of the chunk so every fragment is one
//     ds_read_b128 feeding 4 MFMAs.
//     A "tap" is an arbitrary (dy,dx) input offset, the output grid may be strided/offset, so the same kernel
//     runs: conv forward (stride 1/2), stride-1 and stride-2 (4 parity phases) input gradients, the transposed
//     convolution forward (4 parity phases) and its input gradient (a stride-2 conv of gy).
//
//   weight-gradient kernel  dW[tap][xc][gc] = sum_pixels X[pixel+tap][xc] * G[pixel][gc]
//     tile (one tap) x TX x-channels x TG g-channels per workgroup, k = 32 pixels per step, split-K over
//     pixel ranges into partial slabs that a second kernel sums in a fixed order (deterministic).
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CG_MAXTAPS 49
#define CG_BM 128
#define CG_CK 16
#define CG_AS 20  // LDS pixel stride of the A tile (floats): 16 channels + 4 pad

struct GenArgs {
  const float* x;
  const float* w;  // packed [tap][chunk][nblk][4][BN][4]
  const float* bias;
  float* y;
  int n, hin, win, ldx, xoff, cin;  // channels [xoff, xoff+cin) of a pixel of ldx floats; cin % 4 == 0
  int nchunk;                       // ceil(cin / 16)
  int hv, wv, S;                    // virtual output grid and the input step per virtual pixel
  int hf, wf, ldy, yoff, cout;      // output tensor (n,hf,wf,ldy): channels [yoff, yoff+cout) are written
  int osy, ooy, osx, oox;           // output pixel = (vy*osy+ooy, vx*osx+oox)
  int act, ntaps, nblk;
  short tdy[CG_MAXTAPS], tdx[CG_MAXTAPS];
  unsigned x_bytes;  // convg3 only: size of the x tensor in bytes (buffer descriptor range)
  // fp32 kernel, 4-channel inputs (DispNetS conv1: 2 -> 32, 7x7): a 16-wide k-chunk is FOUR TAPS x 4 channels instead of
  // one tap's 4 channels + 12 zeros; ntaps then counts tap groups and ntaps_real the taps
  int tpack, ntaps_real;
  // two-term fp16 streaming kernel (convg2_fwd_kernel): block-scale workspace of this call, see CG2_WS below; null otherwise
  const float* f2ws;
  // ... split-K (small maps: fewer workgroups than CUs, each with hundreds of dependent k-steps): blockIdx.y = split takes the
  // k-steps [nk * split / ksplit, nk * (split + 1) / ksplit) and leaves its raw sums in skpart[split][m][nblk * BN]; the
  // reduce launch adds the splits in a fixed order, then bias + activation.  ksplit = 1: the kernel writes y itself.
  int ksplit;
  float* skpart;
  long skcap;   // floats available at skpart
  // two-term fp16 halo kernel (convh2_kernel): first tap offsets, halo extent, LDS pixel stride (16-bit words), taps per k-step,
  // channels per plane, k-steps, k-steps per weight group, tile grid, tile rows, resident weights, 16-bit words of the weight region
  int dy0, dx0, HR, HC, PS, TP, PL, nks, GT, tiles_x, tiles_y, trh, res, wsz16;
  // ... output phases of one launch: 1, or the 4 parity classes of a stride-2 transposed form (input gradient of a stride-2
  // convolution, ConvTranspose2d forward) sharing one halo: phase p owns the k-steps [ph_k0[p], ph_k0[p] + ph_kn[p]) of the tap
  // list and writes output pixel (2 vy + ph_oy[p], 2 vx + ph_ox[p]) (one phase: (vy osy + ooy, vx osx + oox) as above);
  // streamed weight groups never straddle phases: group g = k-steps [grp_k0[g], + grp_kn[g]), phase p owns groups
  // [ph_g0[p], ph_g0[p] + ph_ng[p])
  int nph, ngrp, kc;   // kc: column walk (taps ordered dx-major, dy ascending; a weight group = one column of kc taps), 0: generic
  unsigned char ph_k0[4], ph_kn[4], ph_oy[4], ph_ox[4], ph_g0[4], ph_ng[4];
  unsigned char grp_k0[32], grp_kn[32];
};

template <int BN>
__global__ __launch_bounds__(256) void convg_fwd_kernel(GenArgs a) {
  constexpr int NT = BN / 16;
  constexpr int A_FL = CG_BM * CG_AS, B_FL = 16 * BN;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_FL + B_FL)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int M = a.n * a.hv * a.wv;
  const int nb = blockIdx.x % a.nblk, m0 = (blockIdx.x / a.nblk) * CG_BM;

  // loader role: thread owns quarter pq of pixels p0 and p0+64 of the tile
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
  // Loads are unconditional (clamped to a valid address) and zeroed when written to LDS: a load under a branch makes
  // hipcc wait for it right at the issue point (see conv2d.hip).
  float4 ra[2], rb = make_float4(0.f, 0.f, 0.f, 0.f);
  bool rok[2] = {false, false};
  const int wtid = (tid < BN * 4) ? tid : 0;
  auto prefetch = [&](int tap, int chunk) {
    // (tap-packed form: quarter pq of the chunk is tap 4*tap + pq, channels 0..3)
    const int tsel = a.tpack ? min(4 * tap + pq, a.ntaps_real - 1) : tap;
    const bool tok = !a.tpack || 4 * tap + pq < a.ntaps_real;
    const int dy = a.tdy[tsel], dx = a.tdx[tsel];
    const int c = a.tpack ? 0 : chunk * CG_CK + pq * 4;
    const int cc = min(c, a.cin - 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = piy[j] + dy, ix = pix[j] + dx;
      rok[j] = pval[j] && tok && c < a.cin && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
      const int cy = min(max(iy, 0), a.hin - 1), cx = min(max(ix, 0), a.win - 1);
      ra[j] = *(const float4*)(a.x + (pbase[j] + (long)cy * a.win + cx) * a.ldx + a.xoff + cc);
    }
    rb = ((const float4*)a.w)[((long)(tap * a.nchunk + chunk) * a.nblk + nb) * (BN * 4) + wtid];
  };
  auto stage = [&](int buf) {
    float* A = smem + buf * (A_FL + B_FL);
    float* B = A + A_FL;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(float4*)(A + (p0 + j * 64) * CG_AS + pq * 4) = rok[j] ? ra[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < BN * 4) ((float4*)B)[tid] = rb;
  };

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.ntaps * a.nchunk;
  prefetch(0, 0);
  stage(0);
  __syncthreads();
  int tap = 0, chunk = 0;
  for (int ks = 0; ks < nk; ++ks) {
    int ntap = tap, nch = chunk + 1;
    if (nch == a.nchunk) {
      nch = 0;
      ++ntap;
    }
    const bool more = ks + 1 < nk;
    if (more) prefetch(ntap, nch);
    const float* A = smem + (ks & 1) * (A_FL + B_FL);
    const float* B = A + A_FL;
    f32x4 av[2], bv[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) av[mt] = *(const f32x4*)(A + (wave * 32 + mt * 16 + li) * CG_AS + lg * 4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const f32x4*)(B + (lg * BN + nt * 16 + li) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][e], bv[nt][e], acc[mt][nt], 0, 0, 0);
    if (more) stage((ks + 1) & 1);
    __syncthreads();
    tap = ntap;
    chunk = nch;
  }

  // epilogue: bias + activation (dispatched once, not per element), masked store of the real output channels
  float bias_v[NT];
  bool cok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nb * BN + nt * 16 + li;
    cok[nt] = co < a.cout;
    bias_v[nt] = (a.bias && cok[nt]) ? a.bias[co] : 0.f;
  }
  auto emit = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 32 + mt * 16 + lg * 4 + r;
        if (m >= M) continue;
        const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
        float* yp = a.y + (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff +
                    nb * BN + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (cok[nt]) yp[nt * 16] = act_apply(acc[mt][nt][r] + bias_v[nt], ACT);
      }
    }
  };
  if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
  else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
  else emit(std::integral_constant<int, DIS_ACT_NONE>{});
}

// packed[tap][chunk][nb][lg][col][e] = W(tap, ci = chunk*16 + lg*4 + e, co = nb*BN + col)
//   W = w[ci*s_ci + co*s_co + tsrc[tap]] for ci < ci_real && co < co_real, else 0
struct PackArgs {
  const float* w;
  float* packed;
  int ntaps, nchunk, nblk, bn, ci_real, co_real;
  int tpack, ntaps_real;  // see GenArgs
  long s_ci, s_co;
  short tsrc[CG_MAXTAPS];
};
__global__ void convg_pack_kernel(PackArgs a) {
  const long total = (long)a.ntaps * a.nchunk * a.nblk * 16 * a.bn;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    long r = i >> 2;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    r /= a.nblk;
    const int chunk = (int)(r % a.nchunk);
    const int tap = (int)(r / a.nchunk);
    const int ci = a.tpack ? e : chunk * CG_CK + lg * 4 + e, co = nb * a.bn + col;
    const int ts = a.tpack ? 4 * tap + lg : tap;
    float v = 0.f;
    if (ci < a.ci_real && co < a.co_real && (!a.tpack || ts < a.ntaps_real)) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[ts]];
    a.packed[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// bf16x3 form of the forward-like kernel (fp32 results on the bf16 matrix cores: 3-way operand split, 6 products per
// MAC, see conv2d.hip) for layers with >= 32 input channels.  Same tile, addressing, modes and epilogue as
// convg_fwd_kernel; a k-step is 32 channels of one tap:
//   A tile  128 pixels x 32 channels, split when it is written to LDS as [pixel][plane][channel] (208-B pixel stride),
//   B tile  32 channels x BN couts, PRE-split by convg3_pack_kernel into [plane][k-group][cout][8] 16-bit words,
//   6 x 2 x BN/16 MFMAs (v_mfma_f32_16x16x32_bf16) per wave and k-step.
// The tiles take 74 KB of LDS (2 workgroups per CU instead of 5), so the global-memory latency is covered inside the
// workgroup: the operands of k-step s+3 are requested into a ring of three register sets while k-step s runs, and
// k-step s+1 is split and written to the other LDS buffer behind the MFMAs of k-step s.
// ------------------------------------------------------------------------------------------------
#define CG3_CK 32
#define CG3_PS 104  // LDS pixel stride in 16-bit units (3 planes x 32 channels + 8 pad)
struct Pack3Args {
  const float* w;
  unsigned short* packed;  // [tap][chunk][nblk][plane][lg][bn][8]
  int ntaps, nchunk, nblk, bn, ci_real, co_real;
  long s_ci, s_co;
  short tsrc[CG_MAXTAPS];
};
__global__ void convg3_pack_kernel(Pack3Args a) {
  const long total = (long)a.ntaps * a.nchunk * a.nblk * 4 * a.bn * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    long r = i >> 3;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    r /= a.nblk;
    const int chunk = (int)(r % a.nchunk);
    const int tap = (int)(r / a.nchunk);
    const int ci = chunk * CG3_CK + lg * 8 + j, co = nb * a.bn + col;
    float v = 0.f;
    if (ci < a.ci_real && co < a.co_real) v = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
    unsigned h1, h2, h3;
    split3(v, h1, h2, h3);
    const long plane = 4L * a.bn * 8;
    const long base = (((long)(tap * a.nchunk + chunk) * a.nblk + nb) * 3) * plane + ((long)lg * a.bn + col) * 8 + j;
    a.packed[base] = (unsigned short)h1;
    a.packed[base + plane] = (unsigned short)h2;
    a.packed[base + 2 * plane] = (unsigned short)h3;
  }
}

template <int BN>
__global__ __launch_bounds__(256, 2) void convg3_fwd_kernel(GenArgs a) {
  constexpr int NT = BN / 16;
  constexpr int NBQ = (3 * 4 * BN + 255) / 256;                   // 16-byte vectors of the B tile per thread
  constexpr int A_U16 = CG_BM * CG3_PS, B_U16 = NBQ * 256 * 8;    // per buffer (B padded to whole rounds of the block)
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * (A_U16 + B_U16)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int M = a.n * a.hv * a.wv;
  const int nb = blockIdx.x % a.nblk, m0 = (blockIdx.x / a.nblk) * CG_BM;

  // loader role: thread owns channel group pq (4 channels) of pixels p0 + 32 j of the tile
  const int pq = tid & 7, p0 = tid >> 3;
  long pbase[4];
  int piy[4], pix[4];
  bool pval[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + p0 + j * 32;
    pval[j] = m < M;
    const int mm = pval[j] ? m : 0;
    const int vx = mm % a.wv, t = mm / a.wv, vy = t % a.hv, nn = t / a.hv;
    pbase[j] = (long)nn * a.hin * a.win;
    piy[j] = vy * a.S;
    pix[j] = vx * a.S;
  }
  // ring of three register sets.  x is addressed through a buffer descriptor over the whole tensor: a tap outside the
  // image (or a channel group beyond cin) gets an out-of-range offset and loads the zero padding - no clamp, no mask,
  // no branch, so that the staging below is straight-line code the scheduler can spread between the MFMAs
  float4 ra[3][4];
  u32x4 rb[3][NBQ];
  const u32x4* wq = (const u32x4*)a.w;
  const int nk = a.ntaps * a.nchunk;
  int ptap = 0, pchunk = 0, pnext = 0;  // cursor of the next k-step to request
  auto prefetch = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    if (pnext >= nk) return;
    const int dy = a.tdy[ptap], dx = a.tdx[ptap];
    const int c = pchunk * CG3_CK + pq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = piy[j] + dy, ix = pix[j] + dx;
      const bool ok = pval[j] && c < a.cin && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      const unsigned off = ok ? (unsigned)(((pbase[j] + (long)iy * a.win + ix) * a.ldx + a.xoff + c) * 4) : BX_OOB;
      ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), off, 0, 0));
    }
    const long wb = ((long)(ptap * a.nchunk + pchunk) * a.nblk + nb) * (3 * 4 * BN);
#pragma unroll
    for (int q = 0; q < NBQ; ++q) rb[set][q] = wq[wb + min(tid + q * 256, 3 * 4 * BN - 1)];
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
    for (int j = 0; j < 4; ++j) {
      const float4 v = ra[set][j];
      unsigned a1, a2, a3, b1, b2, b3;
      split3_pair(v.x, v.y, a1, a2, a3);
      split3_pair(v.z, v.w, b1, b2, b3);
      unsigned short* p = A + (p0 + j * 32) * CG3_PS + pq * 4;
      *(uint2*)(p) = make_uint2(a1, b1);
      *(uint2*)(p + 32) = make_uint2(a2, b2);
      *(uint2*)(p + 64) = make_uint2(a3, b3);
    }
#pragma unroll
    for (int q = 0; q < NBQ; ++q) ((u32x4*)B)[tid + q * 256] = rb[set][q];  // (B is padded to NBQ * 256 vectors)
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
  // one k-step: request k-step s+3 into the set that held k-step s, run the MFMAs of k-step s, stage k-step s+1
  auto body = [&](int s, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    prefetch(setc);
    const unsigned short* A = smem + (s & 1) * (A_U16 + B_U16);
    const unsigned short* B = A + A_U16;
    s16x8 fa[3][2], fb[3][NT];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        fa[p][mt] = *(const s16x8*)(A + (wave * 32 + mt * 16 + li) * CG3_PS + p * 32 + lg * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *(const s16x8*)(B + ((p * 4 + lg) * BN + nt * 16 + li) * 8);
    }
    // smallest terms first
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[PA[q]][mt]),
                                                               __builtin_bit_cast(bf16x8, fb[PB[q]][nt]), acc[mt][nt], 0,
                                                               0, 0);
    // (unconditional: behind the last k-step it writes stale registers into the buffer nobody reads any more; without a
    // branch the split / ds_write instructions share the MFMAs' basic block and are issued between them)
    stage(std::integral_constant<int, (set + 1) % 3>{}, (s + 1) & 1);
    constexpr int NM = 12 * NT;
#pragma unroll
    for (int g = 0; g < NM; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, (100 + NM - 1) / NM + 1, 0);  // its share of the split VALU
      if (g % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
    }
    __syncthreads();
  };
  for (int ks = 0; ks < nk; ks += 3) {
    body(ks, S0{});
    if (ks + 1 < nk) body(ks + 1, S1{});
    if (ks + 2 < nk) body(ks + 2, S2{});
  }

  // epilogue (as convg_fwd_kernel): bias + activation, masked store of the real output channels
  float bias_v[NT];
  bool cok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nb * BN + nt * 16 + li;
    cok[nt] = co < a.cout;
    bias_v[nt] = (a.bias && cok[nt]) ? a.bias[co] : 0.f;
  }
  auto emit = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 32 + mt * 16 + lg * 4 + r;
        if (m >= M) continue;
        const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
        float* yp = a.y + (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff +
                    nb * BN + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (cok[nt]) yp[nt * 16] = act_apply(acc[mt][nt][r] + bias_v[nt], ACT);
      }
    }
  };
  if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
  else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
  else emit(std::integral_constant<int, DIS_ACT_NONE>{});
}

// ------------------------------------------------------------------------------------------------
// The same streaming implicit GEMM with TWO-term fp16 operands ("f16x2", conv_f16x2.hip: x * 2^s = h1 + h2, 3 products per MAC
// on v_mfma_f32_16x16x32_f16 - half the matrix work of the three-term bf16 form above, two operand planes instead of three).
// fp16's 5 exponent bits need block scaling by powers of two (exact).  Here the blocks are
//   * the WEIGHT tensor of the call (one scale, found by convg2_absmax_kernel, applied by the packing launch), and
//   * every SAMPLE of x (one scale per image: the rows of the GEMM are output pixels, and all taps of an output pixel read its own
//     image, so the scale is constant along K and is undone per output row in the epilogue - no running scales, no rescaling of
//     accumulators).  A streaming kernel has no halo tile to take a maximum over; the per-sample maxima come from one
//     launch in front of the packing launch (x was just written by the producing layer: the read is served by the Infinity
//     Cache for all but the full-resolution maps).
// Values within 2^18 of their image's largest magnitude keep 22 significant bits; smaller ones keep an ABSOLUTE error of 2^-39
// of that magnitude (tests/test_sf_gpu.py: against fp64, beside the three-term kernel, incl. an image with a 1e4 outlier).
// DIS_CONV_SPLIT=bf16x3 / dis_set_conv_split(0) keeps every launch on convg3_fwd_kernel.
// Workspace (CG2_WS floats at the end of the call's first packing slice):
//   [0, n * CG2_XB)        partial maxima of |x| per sample     (convg2_absmax_kernel)
//   [CG2_WOFF, +CG2_WB)    partial maxima of |w|
//   [CG2_EOFF, +n] ints    scale exponent per sample; [CG2_EOFF + CG2_NMAX] the weights'   (convg2_pack_kernel, block 0 / all)
// ------------------------------------------------------------------------------------------------
#define CG2_PS 80    // LDS pixel stride in 16-bit units (2 planes x 32 channels + 16 pad: 2 (mod 4) sixteen-byte units, see F2Cfg::PS)
#define CG2_XB 32    // partial maxima per sample (8 left a 3.5 MB image to one workgroup: 31 us per call, 1.5 ms of a DIS-SF step)
#define CG2_WB 256   // partial maxima of the weights (a 1024 x 512 x 3 x 3 weight is 4.7 M values: 64 blocks took longer than the conv)
#define CG2_NMAX 224 // samples per call
#define CG2_WOFF (CG2_NMAX * CG2_XB)
#define CG2_EOFF (CG2_WOFF + CG2_WB)
#define CG2_WS 7936  // >= CG2_EOFF + CG2_NMAX + 1, a multiple of 256

__global__ __launch_bounds__(256) void convg2_absmax_kernel(const float* __restrict__ x, int n, long hw, int ldx, int xoff, int cin,
                                                            const float* __restrict__ w, long wcount, float* __restrict__ ws) {
  __shared__ float sm[4];
  float m = 0.f;
  int slot;
  if ((int)blockIdx.x < n * CG2_XB) {
    const int nn = blockIdx.x / CG2_XB, q = blockIdx.x % CG2_XB;
    const long p_lo = hw * q / CG2_XB, p_hi = hw * (q + 1) / CG2_XB;
    const int cv = cin >> 2;
    const float* xb = x + ((long)nn * hw) * ldx + xoff;
    const long i_hi = p_hi * cv;
    long i = p_lo * cv + threadIdx.x;
    for (; i + 768 < i_hi; i += 1024) {   // four independent loads in flight per thread
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long q = i + u * 256, p = q / cv;
        v[u] = *(const float4*)(xb + p * ldx + (int)(q - p * cv) * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[u].x)), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (; i < i_hi; i += 256) {
      const long p = i / cv;
      const int c = (int)(i - p * cv);
      const float4 v = *(const float4*)(xb + p * ldx + c * 4);
      m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    slot = blockIdx.x;
  } else {
    const int q = blockIdx.x - n * CG2_XB;
    for (long i = wcount * q / CG2_WB + threadIdx.x; i < wcount * (q + 1) / CG2_WB; i += 256) m = fmaxf(m, fabsf(w[i]));
    slot = CG2_WOFF + q;
  }
  m = f2_wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) ws[slot] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

struct Pack2Args {
  const float* w;
  unsigned short* packed;  // [tap][chunk][nblk][plane][lg][bn][8]
  int ntaps, nchunk, nblk, bn, ci_real, co_real;
  long s_ci, s_co;
  short tsrc[CG_MAXTAPS];
  float* ws;   // CG2_WS block-scale workspace
  int n;       // > 0: block 0 also turns the per-sample partial maxima into exponents (first packing launch of the call)
};
__global__ __launch_bounds__(256) void convg2_pack_kernel(Pack2Args a) {
  __shared__ int s_ew;
  if (threadIdx.x < 64) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < CG2_WB / 64; ++q) m = fmaxf(m, a.ws[CG2_WOFF + q * 64 + threadIdx.x]);
    m = f2_wave_max(m);
    if (threadIdx.x == 0) {
      s_ew = f2_scale_exp(m);
      if (blockIdx.x == 0) ((int*)a.ws)[CG2_EOFF + CG2_NMAX] = s_ew;
    }
  }
  if (blockIdx.x == 0 && a.n > 0) {
    for (int nn = threadIdx.x; nn < a.n; nn += 256) {
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < CG2_XB; ++q) m = fmaxf(m, a.ws[nn * CG2_XB + q]);
      ((int*)a.ws)[CG2_EOFF + nn] = f2_scale_exp(m);
    }
  }
  __syncthreads();
  const float sw = __builtin_ldexpf(1.f, s_ew);
  const long total = (long)a.ntaps * a.nchunk * a.nblk * 4 * a.bn * 4;   // pairs of consecutive channels
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (thread order: tap fastest, then the channel pair - consecutive threads read consecutive floats of the OIHW weights, as
    //  conv_bf16.hip's packing kernels; the packed position is computed from the coordinates)
    const int tap = (int)(i % a.ntaps);
    long r = i / a.ntaps;
    const int j = (int)(r & 3) * 2;
    r >>= 2;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    const int chunk = (int)(r / a.nblk);
    const int ci = chunk * CG3_CK + lg * 8 + j, co = nb * a.bn + col;
    float v0 = 0.f, v1 = 0.f;
    if (co < a.co_real) {
      if (ci < a.ci_real) v0 = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
      if (ci + 1 < a.ci_real) v1 = a.w[(ci + 1) * a.s_ci + co * a.s_co + a.tsrc[tap]];
    }
    unsigned p1, p2;
    f2_split_pair(v0 * sw, v1 * sw, p1, p2);
    const long plane = 4L * a.bn * 8;
    const long base = (((long)(tap * a.nchunk + chunk) * a.nblk + nb) * 2) * plane + ((long)lg * a.bn + col) * 8 + j;
    *(unsigned*)(a.packed + base) = p1;
    *(unsigned*)(a.packed + base + plane) = p2;
  }
}

// NW waves per workgroup = 32 NW output pixels x BN couts: 4 x 64 (128-pixel tiles, two workgroups per CU) for most layers;
// 8 x 128 for the deep ones (>= 128 channels on both sides), whose bound is the operand traffic from L2 / Infinity Cache - a
// 128 x 64 tile moves 24 KB per k-step for 128 x 64 x 32 MACs, a 256 x 128 tile 48 KB for four times as many.
template <int BN, int NW = 4>
struct Cg2Cfg {
  static constexpr int NTHR = 64 * NW, BM = 32 * NW;
  static constexpr int NBQ = (2 * 4 * BN + NTHR - 1) / NTHR;           // 16-byte vectors of the B tile per thread
  static constexpr int A_U16 = BM * CG2_PS, B_U16 = NBQ * NTHR * 8;    // per buffer (B padded to whole rounds of the block)
  static constexpr int LDS_BYTES = 2 * (A_U16 + B_U16) * 2;
};
template <int BN, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void convg2_fwd_kernel(GenArgs a) {
  using CF = Cg2Cfg<BN, NW>;
  constexpr int NT = BN / 16, NTHR = CF::NTHR, BM = CF::BM, PSTEP = NTHR / 8;   // PSTEP: pixels one loader pass covers
  constexpr int NBQ = CF::NBQ, A_U16 = CF::A_U16, B_U16 = CF::B_U16;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int M = a.n * a.hv * a.wv;
  const int nb = blockIdx.x % a.nblk, m0 = (blockIdx.x / a.nblk) * BM;
  const int* sexp = (const int*)a.f2ws + CG2_EOFF;

  // loader role: thread owns channel group pq (4 channels) of pixels p0 + PSTEP j of the tile
  const int pq = tid & 7, p0 = tid >> 3;
  long pbase[4];
  int piy[4], pix[4];
  bool pval[4];
  float psc[4];   // 2^(scale exponent of the row's image)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + p0 + j * PSTEP;
    pval[j] = m < M;
    const int mm = pval[j] ? m : 0;
    const int vx = mm % a.wv, t = mm / a.wv, vy = t % a.hv, nn = t / a.hv;
    pbase[j] = (long)nn * a.hin * a.win;
    piy[j] = vy * a.S;
    pix[j] = vx * a.S;
    psc[j] = __builtin_ldexpf(1.f, sexp[nn]);
  }
  // ring of three register sets, as convg3_fwd_kernel
  float4 ra[3][4];
  u32x4 rb[3][NBQ];
  const u32x4* wq = (const u32x4*)a.w;
  const int nk_all = a.ntaps * a.nchunk;
  const int k0 = (int)((long)nk_all * blockIdx.y / a.ksplit), nk = (int)((long)nk_all * (blockIdx.y + 1) / a.ksplit);
  int ptap = k0 / a.nchunk, pchunk = k0 % a.nchunk, pnext = k0;  // cursor of the next k-step to request
  auto prefetch = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    if (pnext >= nk) return;
    const int dy = a.tdy[ptap], dx = a.tdx[ptap];
    const int c = pchunk * CG3_CK + pq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = piy[j] + dy, ix = pix[j] + dx;
      const bool ok = pval[j] && c < a.cin && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      const unsigned off = ok ? (unsigned)(((pbase[j] + (long)iy * a.win + ix) * a.ldx + a.xoff + c) * 4) : BX_OOB;
      ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), off, 0, 0));
    }
    const long wb = ((long)(ptap * a.nchunk + pchunk) * a.nblk + nb) * (2 * 4 * BN);
#pragma unroll
    for (int q = 0; q < NBQ; ++q) rb[set][q] = wq[wb + min(tid + q * NTHR, 2 * 4 * BN - 1)];
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
    for (int j = 0; j < 4; ++j) {
      const float4 v = ra[set][j];
      unsigned a1, a2, b1, b2;
      f2_split_pair_scaled(v.x, v.y, psc[j], a1, a2);
      f2_split_pair_scaled(v.z, v.w, psc[j], b1, b2);
      unsigned short* p = A + (p0 + j * PSTEP) * CG2_PS + pq * 4;
      *(uint2*)(p) = make_uint2(a1, b1);
      *(uint2*)(p + 32) = make_uint2(a2, b2);
    }
#pragma unroll
    for (int q = 0; q < NBQ; ++q) ((u32x4*)B)[tid + q * NTHR] = rb[set][q];  // (B is padded to NBQ * 256 vectors)
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
    s16x8 fa[2][2], fb[2][NT];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        fa[p][mt] = *(const s16x8*)(A + (wave * 32 + mt * 16 + li) * CG2_PS + p * 32 + lg * 8);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *(const s16x8*)(B + ((p * 4 + lg) * BN + nt * 16 + li) * 8);
    }
    // smallest terms first: x2 w1, x1 w2, x1 w1
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[PA[q]][mt]),
                                                              __builtin_bit_cast(f16x8_t, fb[PB[q]][nt]), acc[mt][nt], 0, 0,
                                                              0);
    stage(std::integral_constant<int, (set + 1) % 3>{}, (s + 1) & 1);
    constexpr int NM = 6 * NT;
#pragma unroll
    for (int g = 0; g < NM; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, (80 + NM - 1) / NM + 1, 0);  // its share of the split VALU
      if (NM >= 12 ? g % 2 == 1 : true) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
    }
    __syncthreads();
  };
  for (int ks = k0; ks < nk; ks += 3) {   // (the LDS buffer parity follows the trip count, not the k-step index)
    body(ks - k0, S0{});
    if (ks + 1 < nk) body(ks - k0 + 1, S1{});
    if (ks + 2 < nk) body(ks - k0 + 2, S2{});
  }

  // epilogue: undo the two block scales per output row (exact), then bias + activation, masked store of the real output channels
  const int ew = sexp[CG2_NMAX];
  float bias_v[NT];
  bool cok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nb * BN + nt * 16 + li;
    cok[nt] = co < a.cout;
    bias_v[nt] = (a.bias && cok[nt]) ? a.bias[co] : 0.f;
  }
  auto emit = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 32 + mt * 16 + lg * 4 + r;
        if (m >= M) continue;
        const int vx = m % a.wv, t = m / a.wv, vy = t % a.hv, nn = t / a.hv;
        const float desc = __builtin_ldexpf(1.f, -(sexp[nn] + ew));
        if (a.ksplit > 1) {   // split-K: this split's raw sums (block-uniform branch); bias and activation in the reduce launch
          float* pp = a.skpart + ((long)blockIdx.y * M + m) * (a.nblk * BN) + nb * BN + li;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pp[nt * 16] = acc[mt][nt][r] * desc;
          continue;
        }
        float* yp = a.y + (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff +
                    nb * BN + li;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (cok[nt]) yp[nt * 16] = act_apply(acc[mt][nt][r] * desc + bias_v[nt], ACT);
      }
    }
  };
  if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
  else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
  else emit(std::integral_constant<int, DIS_ACT_NONE>{});
}

// split-K reduce of convg2_fwd_kernel: y[m][co] = act(sum over the splits (fixed order) + bias); one thread = 4 output channels
__global__ __launch_bounds__(256) void convg2_splitk_reduce_kernel(GenArgs a, int coutp) {
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
    float* yp = a.y + (((long)nn * a.hf + (vy * a.osy + a.ooy)) * a.wf + (vx * a.osx + a.oox)) * a.ldy + a.yoff + co;
    const float s4[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (co + e < a.cout) yp[e] = act_apply(s4[e] + (a.bias ? a.bias[co + e] : 0.f), a.act);
  }
}


// ------------------------------------------------------------------------------------------------
// Halo form of the two-term fp16 convolution, for the large maps of DispNetS (round 4; the structure of conv_bf16.hip's
// convb_halo_kernel with fp32 tensors and two operand planes).  The streaming kernel above gathers - and SPLITS - the A tile of
// every tap from global memory: a pixel is fetched through L2 and split k*k times, and a layer with few taps per output (the
// four parity classes of a transposed convolution: 1 - 4 taps) is a chain of three dependent memory round trips per 128 pixels.
// Here the input halo of an 8 x 16 (or 16 x 16) tile of virtual output positions is fetched ONCE per 32-channel chunk, split
// once into [pixel][plane][channel] fp16 and every tap reads its MFMA operand from LDS at its own offset; the weights (packed
// [chunk][cout block][k-step][plane][lane group][cout][8], pre-scaled) stay resident in LDS when a cout block's share fits
// beside the halo, otherwise they stream in groups of GT k-steps double-buffered through registers.
//   k-step = TP taps x 32/TP channels (TP = 2 for <= 16 input channels)
//   wave w = tile rows MT w .. MT w + MT - 1 x 16 columns; lane (li, lg): column li, k-slice lg; 3 MFMAs per operand pair
//   block scales: one per image (the exponents convg2_absmax_kernel / the packing launch leave in f2ws) and one for the weights
// Persistent workgroups, XCD-contiguous shares of the (tile, cout block) units; the halo of the next stage (two stages with
// resident weights) is in flight in registers - 16-byte items, NH per thread - while the current stage's MFMAs run.
// ------------------------------------------------------------------------------------------------
#define CH2_TC 16
#define CH2_GVEC 1024  // 16-byte vectors of one weight group (16 KB): 4 per thread
// KC > 0 (streamed weights, stride 1, one phase, a full KC x KC window - the 7 x 7 and 5 x 5 layers): the tap list is ordered
// column by column (dx outer, dy ascending) and a weight group is one column.  A wave's MT output rows read the halo rows
// r + ky .. r + ky + MT - 1 for tap row ky, so a column needs only the MT + KC - 1 row fragments r .. r + MT + KC - 2, each read
// ONCE and kept in registers while ky slides down (one new row per k-step instead of MT): with the generic tap walk a k-step is
// 2 (MT + NT) ds_read_b128 per wave for 3 MT NT MFMAs, which keeps the LDS port as busy as the matrix unit (MT 4, NT 2: 12 reads
// per 24 MFMAs, the 7 x 7 layer ran at 27 % of the matrix rate); here it is 2 (1 + NT).
template <int BN, int NH, int MT, bool RES, int PH, int KC = 0>
__global__ __launch_bounds__(256) void convh2_kernel(GenArgs a) {
  static_assert(KC == 0 || (!RES && PH == 1), "column walk: streamed weights, one phase");
  constexpr int NT = BN / 16, TRH = 4 * MT, NSET = RES ? 2 : 1;
  constexpr int NRB = KC > 0 ? (KC * 8 * BN + 255) / 256 : 4;   // 16-byte vectors of a weight group per thread
  extern __shared__ __attribute__((aligned(16))) unsigned short hsm[];
  int* toff = (int*)hsm;                       // 64 ints
  const int bsz = a.GT * 2 * 4 * BN * 8;       // 16-bit words of one weight buffer
  unsigned short* Bb = hsm + 128;
  unsigned short* halo = Bb + a.wsz16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int PS = a.PS, HC = a.HC, PL = a.PL, ipp = PL >> 2;   // items (4 channels) per pixel
  const int* sexp = (const int*)a.f2ws + CG2_EOFF;
  const int ntile = a.n * a.tiles_y * a.tiles_x;
  auto unit_nb = [&](int uu) { return RES ? uu / ntile : uu % a.nblk; };
  auto unit_tile = [&](int uu) { return RES ? uu % ntile : uu / a.nblk; };
  if (tid < 56) {   // (toff[56 .. 63] are the wave-maximum slots, written by other waves before the first barrier)
    int o = 0;
    if (tid < a.ntaps) o = ((a.tdy[tid] - a.dy0) * HC + (a.tdx[tid] - a.dx0)) * PS;
    toff[tid] = o;
  }
  const int units = ntile * a.nblk;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int u_hi = (int)((long)units * (xcd + 1) / nxcd);
  int u = (int)((long)units * xcd / nxcd) + rank;
  if (u >= u_hi) return;

  const int gpt = 4 / a.TP;  // lane groups per tap
  const int lgq = lg / gpt, cg = lg - lgq * gpt;
  const int a_lane = (wave * MT * a.S * HC + li * a.S) * PS + cg * 8;
  const int rowstep = a.S * HC * PS;
  const u32x4* wq = (const u32x4*)a.w;
  u32x4 rb[NRB];
  auto pref_b = [&](int nb, int c, int g) __attribute__((always_inline)) {
    const long base = ((long)(c * a.nblk + nb) * a.nks + a.grp_k0[g]) * (8 * BN);
    const int cnt = a.grp_kn[g] * (8 * BN);
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
      const int idx = tid + i * 256;
      rb[i] = wq[base + (idx < cnt ? idx : 0)];
    }
  };
  // halo items of this thread: (row, column) inside the halo, first channel, LDS offset (fixed for the whole launch)
  const int nitems = a.HR * HC * ipp;
  int it_rc[NH], it_cg[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int i = tid + j * 256;
    const int p = i / ipp, cgi = i - p * ipp;
    const int r = p / HC, cc = p - r * HC;
    it_rc[j] = i < nitems ? (r | (cc << 16)) : 0x4000;  // (past the end: a row outside every image)
    it_cg[j] = cgi * 4 | ((p * PS + cgi * 4) << 8);
  }
  u32x4 pre[NSET][NH];
  // Block scale of x: one per halo stage (tile x 32-channel chunk), found in the kernel - no pass over x in front of the launch.
  // The stage's largest magnitude is taken before the barrier that ends the previous stage (thread maxima -> wave maxima by DPP ->
  // four LDS slots); the contraction runs over chunks, so within a unit the scale is a RUNNING one that only ever shrinks, and the
  // accumulators are multiplied by the (exact, power-of-two) ratio when a chunk brings a larger magnitude than those before it
  // (as conv_wgrad_f16x2_kernel does over tiles).  Finer than the per-image scale of the streaming kernel.
  float* mxs = (float*)(toff + 56);   // [parity][wave]  (the tap offsets use toff[0 .. 48])
  int mpar = 0, s_run = 0;
  auto prep = [&](auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NH; ++j) {   // (items past the halo's end and pixels outside the image were loaded as zeros)
      const u32x4 v = pre[set][j];
      m = fmaxf(fmaxf(m, fabsf(__uint_as_float(v[0]))), fabsf(__uint_as_float(v[1])));
      m = fmaxf(fmaxf(m, fabsf(__uint_as_float(v[2]))), fabsf(__uint_as_float(v[3])));
    }
    m = f2_wave_max(m);
    if (lane == 0) mxs[mpar * 4 + wave] = m;
  };
  auto halo_issue = [&](int uu, int c, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    int t = unit_tile(uu);
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y, nn = t / a.tiles_y;
    const int iy0 = ty * TRH * a.S + a.dy0, ix0 = tx * CH2_TC * a.S + a.dx0;
    const long sbase = (long)nn * a.hin * a.win;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int iy = iy0 + (it_rc[j] & 0xffff), ix = ix0 + (it_rc[j] >> 16);
      const int ch = c * CG3_CK + (it_cg[j] & 0xff);
      const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win && ch < a.cin;
      const long e = (sbase + (long)iy * a.win + ix) * a.ldx + a.xoff + ch;
      pre[set][j] = __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.x, a.x_bytes), ok ? (unsigned)(e * 4) : BX_OOB, 0, 0);
    }
  };
  auto halo_write = [&](auto setc, float sc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      if (tid + j * 256 < nitems) {
        const u32x4 v = pre[set][j];
        unsigned a1, a2, b1, b2;
        f2_split_pair_scaled(__uint_as_float(v[0]), __uint_as_float(v[1]), sc, a1, a2);
        f2_split_pair_scaled(__uint_as_float(v[2]), __uint_as_float(v[3]), sc, b1, b2);
        unsigned short* q = halo + (it_cg[j] >> 8);
        *(uint2*)q = make_uint2(a1, b1);
        *(uint2*)(q + PL) = make_uint2(a2, b2);
      }
    }
  };

  f32x4 acc[PH][MT][NT], bias_v[NT];
  // SACC: the two cross terms (x2 w1, x1 w2 - 2^-11 of the main product) accumulate in registers of their own and join the main
  // sums in the epilogue: a third of the additions into the large running sum, whose fp32 rounding is this kernel's error
  // (7 x 7, 32 -> 32 against fp64: largest error 1.0e-6 -> 5.0e-7 of the largest output, mean 6.3e-8 -> 3.7e-8; scripts/diag/halo_err_map.py)
  constexpr bool SACC = (PH == 1) && MT * NT <= 8;   // (16 tiles per wave: 64 more registers would spill)
  f32x4 accs[SACC ? MT : 1][SACC ? NT : 1];
  int bias_nb = -1;
  // k-steps [0, kn) of a weight block B ([k-step][plane][lg][BN][8]) whose tap offsets start at tq, into one phase's accumulators;
  // software-pipelined by hand: the operands of k-step kk+1 are requested before the MFMAs of kk issue
  auto ksteps = [&](const unsigned short* B, const int* tq, int kn, f32x4 (&acc)[MT][NT]) __attribute__((always_inline)) {
    const unsigned short* bl = B + (lg * BN + li) * 8;
    auto frag = [&](int kk, int to, s16x8 (&fa)[2][MT], s16x8 (&fb)[2][NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[p][mt] = *(const s16x8*)(halo + a_lane + mt * rowstep + to + p * PL);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[p][nt] = *(const s16x8*)(bl + ((kk * 2 + p) * 4 * BN + nt * 16) * 8);
      }
    };
    auto mac = [&](const s16x8 (&fa)[2][MT], const s16x8 (&fb)[2][NT]) __attribute__((always_inline)) {
      // smallest terms first: x2 w1, x1 w2, x1 w1
      constexpr int PA[3] = {1, 0, 0};
      constexpr int PB[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4& d = (SACC && q < 2) ? accs[SACC ? mt : 0][SACC ? nt : 0] : acc[mt][nt];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fb[PB[q]][nt]),
                                                      __builtin_bit_cast(f16x8_t, fa[PA[q]][mt]), d, 0, 0, 0);
          }
    };
    s16x8 fa0[2][MT], fb0[2][NT], fa1[2][MT], fb1[2][NT];
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
  // one column of KC taps (weights B: [ky][plane][lg][BN][8]; `to` = halo offset of the column's first tap): sliding window of row
  // fragments.  Rn: the first MT rows of the NEXT column (offset to_next), requested during the last k-steps of this one.
  constexpr int KCC = KC > 0 ? KC : 1;
  s16x8 Rn[MT][2];
  auto col_rows = [&](int to) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) Rn[j][p] = *(const s16x8*)(halo + a_lane + j * rowstep + to + p * PL);
  };
  auto kcolumn = [&](const unsigned short* B, int to, int to_next, bool has_next, f32x4 (&acc)[MT][NT]) __attribute__((always_inline)) {
    const unsigned short* bl = B + (lg * BN + li) * 8;
    s16x8 R[MT + KCC - 1][2];
    s16x8 fb[2][2][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) R[j][p] = Rn[j][p];
    auto load_fb = [&](int ky, s16x8 (&f)[2][NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) f[p][nt] = *(const s16x8*)(bl + ((ky * 2 + p) * 4 * BN + nt * 16) * 8);
    };
    load_fb(0, fb[0]);
    constexpr int PA[3] = {1, 0, 0};
    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
    for (int ky = 0; ky < KCC; ++ky) {
      if (ky + 1 < KCC) {
        // the row that enters the window at the next k-step, and its weights
#pragma unroll
        for (int p = 0; p < 2; ++p) R[ky + MT][p] = *(const s16x8*)(halo + a_lane + (ky + MT) * rowstep + to + p * PL);
        load_fb(ky + 1, fb[(ky + 1) & 1]);
      } else if (has_next) {
        col_rows(to_next);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            f32x4& d = (SACC && q < 2) ? accs[SACC ? mt : 0][SACC ? nt : 0] : acc[mt][nt];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fb[ky & 1][PB[q]][nt]),
                                                      __builtin_bit_cast(f16x8_t, R[ky + mt][PA[q]]), d, 0, 0, 0);
          }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;
  auto load_bias = [&](int nb) __attribute__((always_inline)) {
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
        for (int mt = 0; mt < MT; ++mt) acc[ph][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (SACC) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accs[SACC ? mt : 0][SACC ? nt : 0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // after the barrier: the stage's scale exponent from the four wave maxima; `first`: the unit's first chunk (fresh accumulators)
  auto stage_scale = [&](bool first) __attribute__((always_inline)) -> float {
    const float4 mv = *(const float4*)(mxs + mpar * 4);
    mpar ^= 1;
    const int e = f2_scale_exp(fmaxf(fmaxf(mv.x, mv.y), fmaxf(mv.z, mv.w)));
    if (first) {
      s_run = e;
    } else if (e < s_run) {   // (block-uniform, rare) a larger magnitude than the chunks before it: the accumulators follow
      const float r = __builtin_ldexpf(1.f, e - s_run);
#pragma unroll
      for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[ph][mt][nt] *= r;
      if (SACC) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) accs[SACC ? mt : 0][SACC ? nt : 0] *= r;
      }
      s_run = e;
    }
    return __builtin_ldexpf(1.f, s_run);
  };
  const int ew = sexp[CG2_NMAX];
  // epilogue: lane (li, lg) holds couts nb*BN + nt*16 + lg*4 + {0..3} of position (ty*TRH + wave*MT + mt, tx*16 + li); the two
  // block scales are undone first (exact: powers of two)
  auto epilogue = [&](int uu, int nb) __attribute__((always_inline)) {
    int t = unit_tile(uu);
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y, nn = t / a.tiles_y;
    const float desc = __builtin_ldexpf(1.f, -(s_run + ew));   // (the unit's final running scale and the weights')
    auto emit = [&](auto actc) __attribute__((always_inline)) {
      constexpr int ACT = decltype(actc)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int vy = ty * TRH + wave * MT + mt, vx = tx * CH2_TC + li;
        if (vy >= a.hv || vx >= a.wv) continue;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
          const int oy = vy * a.osy + a.ph_oy[ph], ox = vx * a.osx + a.ph_ox[ph];
          if (PH > 1 && (oy >= a.hf || ox >= a.wf)) continue;   // (the classes of an odd output size differ by one row / column)
          const long pe = (((long)nn * a.hf + oy) * a.wf + ox) * a.ldy + a.yoff;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = nb * BN + nt * 16 + lg * 4;
            if (co >= a.cout) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sum = SACC ? acc[ph][mt][nt][r] + accs[SACC ? mt : 0][SACC ? nt : 0][r] : acc[ph][mt][nt][r];
              o[r] = act_apply(sum * desc + bias_v[nt][r], ACT);
            }
            if (co + 4 <= a.cout && ((pe + co) & 3) == 0) {
              *(float4*)(a.y + pe + co) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
              for (int r = 0; r < 4 && co + r < a.cout; ++r) a.y[pe + co + r] = o[r];
            }
          }
        }
      }
    };
    if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{});
    else if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{});
    else emit(std::integral_constant<int, DIS_ACT_NONE>{});
  };

  if constexpr (RES) {
    // Resident weights.  The walk is a flat sequence of stages (unit, chunk); the halos of the next TWO stages are in flight
    // (register sets alternate): a stage of a narrow high-resolution layer is far shorter than a memory round trip.
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
      prep(setc);
      if (nb != wres_nb) {  // (block-uniform) this cout block's weights, all chunks: once per workgroup and block
        __syncthreads();
        const int nv = a.nks * 8 * BN;
        for (int c = 0; c < a.nchunk; ++c)
          for (int i = tid; i < nv; i += 256) ((u32x4*)Bb)[c * nv + i] = wq[(long)(c * a.nblk + nb) * nv + i];
        wres_nb = nb;
      }
      __syncthreads();  // every wave is done with the previous stage's halo; the stage's wave maxima are visible
      halo_write(setc, stage_scale(c0 == 0));
      __syncthreads();
      if (u2 < u_hi) halo_issue(u2, c2, setc);  // the set just consumed takes the stage after next
#pragma unroll
      for (int ph = 0; ph < PH; ++ph)
        ksteps(Bb + ((long)c0 * a.nks + a.ph_k0[ph]) * (8 * BN * 8), toff + a.ph_k0[ph] * a.TP + lgq, a.ph_kn[ph], acc[ph]);
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
  } else {
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
        prep(S0{});
        __syncthreads();  // every wave is done with the previous stage's halo; the stage's wave maxima are visible
        halo_write(S0{}, stage_scale(c == 0));
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
          const int g_lo = a.ph_g0[ph], g_hi = g_lo + a.ph_ng[ph];
          for (int g = g_lo; g < g_hi; ++g) {
            unsigned short* B = Bb + (flat & 1) * bsz;
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
              const int idx = tid + i * 256;
              if (idx < a.GT * 8 * BN) ((u32x4*)B)[idx] = rb[i];
            }
            __syncthreads();
            if (g == 0) {  // next stage's halo
              if (c + 1 < a.nchunk) halo_issue(u, c + 1, S0{});
              else if (un < u_hi) halo_issue(un, 0, S0{});
            }
            if (g + 1 < a.ngrp) pref_b(nb, c, g + 1);
            else if (c + 1 < a.nchunk) pref_b(nb, c + 1, 0);
            else if (un < u_hi) pref_b(unit_nb(un), 0, 0);
            if constexpr (KC > 0) {
              if (g == 0) col_rows(toff[0]);   // (the first column of a stage: its rows were not requested by a predecessor)
              kcolumn(B, toff[g * KC], g + 1 < a.ngrp ? toff[(g + 1) * KC] : 0, g + 1 < a.ngrp, acc[ph]);
            } else {
              ksteps(B, toff + a.grp_k0[g] * a.TP + lgq, a.grp_kn[g], acc[ph]);
            }
            ++flat;
          }
        }
      }
      epilogue(u, nb);
      if (un >= u_hi) break;
      u = un;
    }
  }
}

// packed[chunk][nb][k-step][plane][lg][col][j] = plane of 2^ew W(tap, ci, co = nb*BN + col), (tap, ci) = (k-step*TP + lg / (4/TP),
// chunk*32 + (lg % (4/TP))*8 + j); taps past the last one and channels past ci_real hold zeros.  Block 0 also turns the
// per-sample partial maxima into exponents (as convg2_pack_kernel).
struct PackH2Args {
  const float* w;
  unsigned short* packed;
  int ntaps, nchunk, nblk, bn, ci_real, co_real, tp, nks;
  long s_ci, s_co;
  short tsrc[CG_MAXTAPS];
  float* ws;
  int n;
};
__global__ __launch_bounds__(256) void convh2_pack_kernel(PackH2Args a) {
  __shared__ int s_ew;
  if (threadIdx.x < 64) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < CG2_WB / 64; ++q) m = fmaxf(m, a.ws[CG2_WOFF + q * 64 + threadIdx.x]);
    m = f2_wave_max(m);
    if (threadIdx.x == 0) {
      s_ew = f2_scale_exp(m);
      if (blockIdx.x == 0) ((int*)a.ws)[CG2_EOFF + CG2_NMAX] = s_ew;
    }
  }
  if (blockIdx.x == 0 && a.n > 0) {
    for (int nn = threadIdx.x; nn < a.n; nn += 256) {
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < CG2_XB; ++q) m = fmaxf(m, a.ws[nn * CG2_XB + q]);
      ((int*)a.ws)[CG2_EOFF + nn] = f2_scale_exp(m);
    }
  }
  __syncthreads();
  const float sw = __builtin_ldexpf(1.f, s_ew);
  const int gpt = 4 / a.tp;
  const long total = (long)a.nchunk * a.nblk * a.nks * 4 * a.bn * 4;   // pairs of consecutive channels
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ks = (int)(i % a.nks);   // (thread order: k-step fastest, then the channel pair: coalesced reads of the weights)
    long r = i / a.nks;
    const int j = (int)(r & 3) * 2;
    r >>= 2;
    const int col = (int)(r % a.bn);
    r /= a.bn;
    const int lg = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % a.nblk);
    const int chunk = (int)(r / a.nblk);
    const int tap = ks * a.tp + lg / gpt;
    const int ci = chunk * CG3_CK + (lg % gpt) * 8 + j, co = nb * a.bn + col;
    float v0 = 0.f, v1 = 0.f;
    if (tap < a.ntaps && co < a.co_real) {
      if (ci < a.ci_real) v0 = a.w[ci * a.s_ci + co * a.s_co + a.tsrc[tap]];
      if (ci + 1 < a.ci_real) v1 = a.w[(ci + 1) * a.s_ci + co * a.s_co + a.tsrc[tap]];
    }
    unsigned p1, p2;
    f2_split_pair(v0 * sw, v1 * sw, p1, p2);
    const long plane = 4L * a.bn * 8;
    const long base = ((((long)chunk * a.nblk + nb) * a.nks + ks) * 2) * plane + ((long)lg * a.bn + col) * 8 + j;
    *(unsigned*)(a.packed + base) = p1;
    *(unsigned*)(a.packed + base + plane) = p2;
  }
}

static int ch2_num_cus() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}
static long ch2_lds(const GenArgs& a) { return 256 + 2L * a.wsz16 + (long)a.HR * a.HC * a.PS * 2; }
#define CH2_LDS_MAX (156 * 1024)
// halo form: worth it when the map is large (tile quantisation wastes small maps; their layers are weight-bound anyway) and the
// halo + weight buffers fit LDS.  Fills the halo fields of a.
static bool ch2_plan(GenArgs& a, int bn) {
  // DIS_CONVG_HALO_MIN: smallest virtual output grid (pixels) that takes this form; read per call (host-side, ~100 ns) so that
  // the tests can run one layer both ways in one process.  0 pixels never qualify; a huge value keeps every call streaming.
  const char* mh = getenv("DIS_CONVG_HALO_MIN");
  const long min_hw = mh ? atol(mh) : 1024;
  if ((long)a.hv * a.wv < min_hw || a.ntaps > CG_MAXTAPS) return false;
  int dy0 = a.tdy[0], dy1 = a.tdy[0], dx0 = a.tdx[0], dx1 = a.tdx[0];
  for (int t = 1; t < a.ntaps; ++t) {
    dy0 = a.tdy[t] < dy0 ? a.tdy[t] : dy0; dy1 = a.tdy[t] > dy1 ? a.tdy[t] : dy1;
    dx0 = a.tdx[t] < dx0 ? a.tdx[t] : dx0; dx1 = a.tdx[t] > dx1 ? a.tdx[t] : dx1;
  }
  a.dy0 = dy0; a.dx0 = dx0;
  a.TP = a.cin <= 16 ? 2 : 1;
  if (a.nph > 1 && a.TP != 1) return false;   // (a k-step of two taps could straddle two phases)
  if (a.nph == 1) {
    a.ph_k0[0] = 0; a.ph_kn[0] = (unsigned char)((a.ntaps + a.TP - 1) / a.TP);
    a.ph_oy[0] = (unsigned char)a.ooy; a.ph_ox[0] = (unsigned char)a.oox;
  }
  a.PL = 32 / a.TP;
  a.PS = a.TP == 1 ? 80 : 48;   // 2 planes + pad: 2 (mod 4) sixteen-byte units (F2Cfg::PS)
  a.nks = (a.ntaps + a.TP - 1) / a.TP;
  if (a.nks * a.TP > 64) return false;
  a.GT = CH2_GVEC / (8 * bn);
  if (a.GT > a.nks) a.GT = a.nks;
  a.HC = (CH2_TC - 1) * a.S + (dx1 - dx0) + 1;
  a.tiles_x = (a.wv + CH2_TC - 1) / CH2_TC;
  const long ipp = a.PL / 4;
  auto rows = [&](int trh) { return (trh - 1) * a.S + (dy1 - dy0) + 1; };
  // a cout block's weights (all chunks) stay in LDS when they fit next to the halo of an 8-row tile with two workgroups per CU
  const long wall = (long)a.nchunk * a.nks * 8 * bn * 16;
  const long hal8 = (long)rows(8) * a.HC * a.PS * 2, items8 = (long)rows(8) * a.HC * ipp;
  a.trh = 8;
  a.HR = rows(8);
  if (256 + wall + hal8 <= 78 * 1024 && items8 <= 8 * 256) {
    a.res = 1;
    a.wsz16 = (int)(wall / 2);
  } else {
    a.res = 0;
    a.wsz16 = (int)((2L * a.GT * 8 * bn * 16) / 2);
    const long hal16 = (long)rows(16) * a.HC * a.PS * 2, items16 = (long)rows(16) * a.HC * ipp;
    static const bool no16 = getenv("DIS_CONVG_T16") && getenv("DIS_CONVG_T16")[0] == '0';
    if (!no16 && a.nph == 1 && a.hv >= 32 && 256 + 2L * a.wsz16 + hal16 <= CH2_LDS_MAX && items16 <= 20 * 256) {
      a.trh = 16;
      a.HR = rows(16);
    } else if (items8 > 20 * 256) {
      return false;
    }
  }
  // column walk (see convh2_kernel, KC): a full K x K stride-1 window, 16-row tiles, the instances that exist
  a.kc = 0;
  {
    const int K = dy1 - dy0 + 1;
    static const bool nokc = getenv("DIS_CONVG_KC") && getenv("DIS_CONVG_KC")[0] == '0';
    if (!nokc && a.nph == 1 && a.S == 1 && a.TP == 1 && !a.res && a.trh == 16 && dx1 - dx0 + 1 == K && a.ntaps == K * K &&
        ((K == 7 && bn == 32) || (K == 5 && bn == 64))) {
      const long wsz = 2L * K * 8 * bn * 16 / 2;   // two buffers of one column
      if (256 + 2 * wsz + (long)a.HR * a.HC * a.PS * 2 <= CH2_LDS_MAX) {
        a.kc = K;
        a.GT = K;
        a.wsz16 = (int)wsz;
      }
    }
  }
  a.tiles_y = (a.hv + a.trh - 1) / a.trh;
  // weight groups (streamed form): GT k-steps each, cut at the phase boundaries
  int ng = 0;
  for (int ph = 0; ph < a.nph; ++ph) {
    a.ph_g0[ph] = (unsigned char)ng;
    for (int k0 = 0; k0 < a.ph_kn[ph]; k0 += a.GT) {
      if (ng >= 32) return false;
      a.grp_k0[ng] = (unsigned char)(a.ph_k0[ph] + k0);
      a.grp_kn[ng] = (unsigned char)(a.ph_kn[ph] - k0 < a.GT ? a.ph_kn[ph] - k0 : a.GT);
      ++ng;
    }
    a.ph_ng[ph] = (unsigned char)(ng - a.ph_g0[ph]);
  }
  a.ngrp = ng;
  // four phases in one launch pay when their weights are resident or the cout block is wide (measured, DispNetS bs=8 x 4 frames:
  // 32 -> 16 at 512 x 432 resident 0.61 -> 0.31 ms; 128 -> 64 streamed 0.21 -> 0.19 ms; but 64 -> 32 streamed 0.24 -> 0.28 ms and
  // the 5 x 5 input gradient 64 -> 32 0.30 -> 0.40 ms: per phase most of their launches keep resident weights)
  if (a.nph > 1 && !a.res && bn < 64) return false;
  return ch2_lds(a) <= CH2_LDS_MAX;
}
template <int BN, int NH, int MT, bool RES, int PH = 1, int KC = 0>
static int ch2_launch3(const GenArgs& a, long grid, long lds, hipStream_t s) {
  static bool attr = false;
  auto kern = convh2_kernel<BN, NH, MT, RES, PH, KC>;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH2_LDS_MAX);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  DIS_TAG(KC > 0 ? "convh2_kernel (f16x2 LDS halo, column walk)" : PH > 1 ? (RES ? "convh2_kernel (f16x2 LDS halo, 4 phases, resident weights)" : "convh2_kernel (f16x2 LDS halo, 4 phases)")
                 : (RES ? "convh2_kernel (f16x2 LDS halo, resident weights)" : "convh2_kernel (f16x2 LDS halo)"));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)lds, s, a);
  return DIS_OK;
}
template <int BN>
static int ch2_launch(const GenArgs& a, int nh, long grid, long lds, hipStream_t s) {
  if (a.nph == 4) {
    if (a.res) {
      if (nh <= 4) return ch2_launch3<BN, 4, 2, true, 4>(a, grid, lds, s);
      return ch2_launch3<BN, 8, 2, true, 4>(a, grid, lds, s);
    }
    if (nh <= 8) return ch2_launch3<BN, 8, 2, false, 4>(a, grid, lds, s);
    if (nh <= 14) return ch2_launch3<BN, 14, 2, false, 4>(a, grid, lds, s);
    return ch2_launch3<BN, 20, 2, false, 4>(a, grid, lds, s);
  }
  if (a.res) {
    if (nh <= 4) return ch2_launch3<BN, 4, 2, true>(a, grid, lds, s);
    return ch2_launch3<BN, 8, 2, true>(a, grid, lds, s);
  }
  if (a.kc > 0) {
    if constexpr (BN == 32) {
      if (a.kc == 7 && nh <= 16) return ch2_launch3<32, 16, 4, false, 1, 7>(a, grid, lds, s);
    }
    if constexpr (BN == 64) {
      if (a.kc == 5 && nh <= 14) return ch2_launch3<64, 14, 4, false, 1, 5>(a, grid, lds, s);
    }
    return DIS_ERR_UNSUPPORTED;
  }
  if (a.trh == 16) {
    if (nh <= 8) return ch2_launch3<BN, 8, 4, false>(a, grid, lds, s);
    if (nh <= 14) return ch2_launch3<BN, 14, 4, false>(a, grid, lds, s);
    return ch2_launch3<BN, 20, 4, false>(a, grid, lds, s);
  }
  if (nh <= 8) return ch2_launch3<BN, 8, 2, false>(a, grid, lds, s);
  if (nh <= 14) return ch2_launch3<BN, 14, 2, false>(a, grid, lds, s);
  return ch2_launch3<BN, 20, 2, false>(a, grid, lds, s);
}
static int ch2_run(GenArgs a, int bn, const float* w_raw, float* wpack, int ci_real, int co_real, long s_ci, long s_co,
                   const short* tsrc, hipStream_t s) {
  PackH2Args p;
  p.w = w_raw; p.packed = (unsigned short*)wpack; p.ntaps = a.ntaps; p.nchunk = a.nchunk; p.nblk = a.nblk; p.bn = bn;
  p.ci_real = ci_real; p.co_real = co_real; p.tp = a.TP; p.nks = a.nks; p.s_ci = s_ci; p.s_co = s_co;
  for (int t = 0; t < a.ntaps; ++t) p.tsrc[t] = tsrc[t];
  if (a.kc > 0) {
    // column walk: taps ordered by (dx, dy) ascending - slot (dx - dx0) * kc + (dy - dy0); the window is full (ch2_plan)
    short ty[CG_MAXTAPS], tx[CG_MAXTAPS];
    for (int t = 0; t < a.ntaps; ++t) {
      const int slot = (a.tdx[t] - a.dx0) * a.kc + (a.tdy[t] - a.dy0);
      ty[slot] = a.tdy[t]; tx[slot] = a.tdx[t]; p.tsrc[slot] = tsrc[t];
    }
    for (int t = 0; t < a.ntaps; ++t) a.tdy[t] = ty[t], a.tdx[t] = tx[t];
  }
  p.ws = const_cast<float*>(a.f2ws); p.n = 0;   // (no per-image exponents: the halo kernel scales x per tile)
  const long ptotal = (long)a.nchunk * a.nblk * a.nks * 4 * bn * 4;
  hipLaunchKernelGGL(convh2_pack_kernel, dim3(dis_ew_grid(ptotal, 256)), dim3(256), 0, s, p);
  a.w = wpack;
  const long units = (long)a.n * a.tiles_y * a.tiles_x * a.nblk;
  if (units > 2147483647L) return DIS_ERR_BAD_SHAPE;
  const long lds = ch2_lds(a);
  // persistent grid: as many workgroups as stay resident (LDS-bound, at most 4 per CU), a multiple of the 8 XCDs
  long wpc = (160L * 1024) / lds;
  wpc = wpc > 4 ? 4 : (wpc < 1 ? 1 : wpc);
  long grid = wpc * ch2_num_cus();
  if (grid > units) grid = units;
  if (grid >= 8) grid -= grid % 8;
  const int nh = (int)(((long)a.HR * a.HC * (a.PL / 4) + 255) / 256);
  int rc;
  if (bn == 64) rc = ch2_launch<64>(a, nh, grid, lds, s);
  else if (bn == 32) rc = ch2_launch<32>(a, nh, grid, lds, s);
  else rc = ch2_launch<16>(a, nh, grid, lds, s);
  if (rc != DIS_OK) return rc;
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// Block maxima of a dis_convg_run call under the two-term split, launched when the first kernel that needs them is chosen: the
// halo form scales x per tile by itself and needs the weights' maximum only (256 blocks); the streaming form also needs the
// per-image maxima of x (a pass over x).
struct F2Maxima {
  const float* x; int n; long hw; int ldx, xoff, cin;
  const float* w; long wcount;
  float* ws;
  bool w_done, x_done;
};
static void f2_maxima(F2Maxima* m, bool need_x, hipStream_t s) {
  if (!m || !m->ws) return;
  if (need_x ? m->x_done : m->w_done) return;
  // (n = 0: the weight blocks only; a full launch recomputes the weights' maximum too - idempotent)
  const int n = need_x ? m->n : 0;
  hipLaunchKernelGGL(convg2_absmax_kernel, dim3(n * CG2_XB + CG2_WB), dim3(256), 0, s, m->x, n, m->hw, m->ldx, m->xoff, m->cin, m->w,
                     m->wcount, m->ws);
  m->w_done = true;
  if (need_x) m->x_done = true;
}
static int cg_bn(int cout) { return cout > 32 ? 64 : (cout > 16 ? 32 : 16); }

// split-K factor of the two-term streaming kernel: only when the launch has fewer workgroups than the device has CUs and a long
// chain of k-steps; aims at ~2 workgroups per CU, at least 12 k-steps per split, at most 8 splits
// 256 x 128 tiles (convg2_fwd_kernel<128, 8>): both channel counts >= 128 and the cout a multiple of 128
static bool cg2_big(int cin, int cout) {
  static const bool off = getenv("DIS_CONVG_BIG") && getenv("DIS_CONVG_BIG")[0] == '0';
  return !off && cin >= 128 && cout >= 128 && cout % 128 == 0;
}
static int cg2_ksplit(long wgs, int nk) {
  if (wgs >= 256 || nk < 48) return 1;
  long ks = (512 + wgs - 1) / wgs;
  if (ks > 8) ks = 8;
  while (ks > 1 && nk / ks < 12) --ks;
  return (int)ks;
}
// ... and enough of them: a launch of M output positions and nk k-steps takes the large tiles when they still make ~3/4 of a
// workgroup per CU (with split-K).  The parity classes of the small stride-2 maps (M = 7 k, 16 - 64 k-steps) do not: 56 workgroups
// of 256 x 128 ran 0.19 -> 0.23 ms.
static bool cg2_big_for(int cin, int cout, long M, int nk) {
  if (!cg2_big(cin, cout)) return false;
  const long wgs = ((M + 255) / 256) * (cout / 128);
  return wgs * cg2_ksplit(wgs, nk) >= 192;
}
// one launch of the forward-like kernel (packs its weights first)
static int cg_run(GenArgs a, const float* w_raw, float* wpack, int ci_real, int co_real, long s_ci, long s_co,
                  const short* tsrc, hipStream_t s, F2Maxima* mx, bool halo_only = false) {
  if (a.ntaps <= 0) return DIS_OK;  // empty phase
  if ((long)a.n * a.hv * a.wv <= 0) return DIS_OK;
  const int bn = cg_bn(a.cout), bn0 = bn;
  a.nblk = (a.cout + bn - 1) / bn;
  static const bool use3 = !(getenv("DIS_CONV_BF16X3") && getenv("DIS_CONV_BF16X3")[0] == '0');
  const long xb3 = (long)a.n * a.hin * a.win * a.ldx * 4;  // the bf16x3 kernel addresses x with 31-bit byte offsets
  if (use3 && a.f2ws && a.cin >= CG3_CK && xb3 < 0x7fff0000L) {  // two-term fp16 form (default): see convg2_fwd_kernel
    a.x_bytes = (unsigned)xb3;
    a.nchunk = (a.cin + CG3_CK - 1) / CG3_CK;
    if (ch2_plan(a, bn)) {   // large maps: LDS halo form
      f2_maxima(mx, false, s);
      return ch2_run(a, bn, w_raw, wpack, ci_real, co_real, s_ci, s_co, tsrc, s);
    }
    if (halo_only) return DIS_ERR_UNSUPPORTED;   // (the caller falls back to one launch per phase)
    f2_maxima(mx, true, s);
    const bool big = cg2_big_for(a.cin, a.cout, (long)a.n * a.hv * a.wv, a.ntaps * a.nchunk);   // deep layers: 256 x 128 tiles
    const int bn = big ? 128 : bn0;
    a.nblk = (a.cout + bn - 1) / bn;
    Pack2Args p2;
    p2.w = w_raw; p2.packed = (unsigned short*)wpack; p2.ntaps = a.ntaps; p2.nchunk = a.nchunk; p2.nblk = a.nblk;
    p2.bn = bn; p2.ci_real = ci_real; p2.co_real = co_real; p2.s_ci = s_ci; p2.s_co = s_co;
    for (int t = 0; t < a.ntaps; ++t) p2.tsrc[t] = tsrc[t];
    p2.ws = const_cast<float*>(a.f2ws); p2.n = a.n;   // (every phase recomputes the same exponents: idempotent)
    const long ptotal2 = (long)a.ntaps * a.nchunk * a.nblk * 4 * bn * 4;
    hipLaunchKernelGGL(convg2_pack_kernel, dim3(dis_ew_grid(ptotal2, 256)), dim3(256), 0, s, p2);
    a.w = wpack;
    const long M2 = (long)a.n * a.hv * a.wv;
    const int bm2 = big ? 256 : CG_BM;
    const long grid2 = ((M2 + bm2 - 1) / bm2) * a.nblk;
    if (grid2 > 2147483647L) return DIS_ERR_BAD_SHAPE;
    // split-K for the small maps (see GenArgs::ksplit): enough workgroups to fill the device twice, >= 12 k-steps per split
    const int nk2 = a.ntaps * a.nchunk, coutp = a.nblk * bn;
    int ksplit = cg2_ksplit(grid2, nk2);
    while (ksplit > 1 && (long)ksplit * M2 * coutp > a.skcap) --ksplit;
    if (!a.skpart) ksplit = 1;
    a.ksplit = ksplit;
    DIS_TAG(ksplit > 1 ? "convg2_fwd_kernel (f16x2 streaming, split-K)" : "convg2_fwd_kernel (f16x2 streaming)");
    const dim3 g2((unsigned)grid2, (unsigned)ksplit);
    if (big) {
      static bool attr = false;
      if (!attr) {
        constexpr int lds_attr = Cg2Cfg<128, 8>::LDS_BYTES;
        hipError_t e = hipFuncSetAttribute((const void*)convg2_fwd_kernel<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);
        if (e != hipSuccess) return (int)e;
        attr = true;
      }
      DIS_TAG(ksplit > 1 ? "convg2_fwd_kernel (f16x2 streaming, 256 x 128 tiles, split-K)" : "convg2_fwd_kernel (f16x2 streaming, 256 x 128 tiles)");
      constexpr int lds_big = Cg2Cfg<128, 8>::LDS_BYTES;
      hipLaunchKernelGGL((convg2_fwd_kernel<128, 8>), g2, dim3(512), lds_big, s, a);
    } else if (bn == 64) hipLaunchKernelGGL((convg2_fwd_kernel<64>), g2, dim3(256), Cg2Cfg<64>::LDS_BYTES, s, a);
    else if (bn == 32) hipLaunchKernelGGL((convg2_fwd_kernel<32>), g2, dim3(256), Cg2Cfg<32>::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((convg2_fwd_kernel<16>), g2, dim3(256), Cg2Cfg<16>::LDS_BYTES, s, a);
    if (ksplit > 1)
      hipLaunchKernelGGL(convg2_splitk_reduce_kernel, dim3(dis_ew_grid(M2 * ((a.cout + 3) / 4), 256)), dim3(256), 0, s, a, coutp);
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  if (halo_only) return DIS_ERR_UNSUPPORTED;
  if (use3 && a.cin >= CG3_CK && xb3 < 0x7fff0000L) {  // bf16x3 form: 32-channel k-steps, weights pre-split
    a.x_bytes = (unsigned)xb3;
    a.nchunk = (a.cin + CG3_CK - 1) / CG3_CK;
    Pack3Args p3;
    p3.w = w_raw; p3.packed = (unsigned short*)wpack; p3.ntaps = a.ntaps; p3.nchunk = a.nchunk; p3.nblk = a.nblk;
    p3.bn = bn; p3.ci_real = ci_real; p3.co_real = co_real; p3.s_ci = s_ci; p3.s_co = s_co;
    for (int t = 0; t < a.ntaps; ++t) p3.tsrc[t] = tsrc[t];
    const long ptotal3 = (long)a.ntaps * a.nchunk * a.nblk * 4 * bn * 8;
    hipLaunchKernelGGL(convg3_pack_kernel, dim3(dis_ew_grid(ptotal3, 256)), dim3(256), 0, s, p3);
    a.w = wpack;
    const long M3 = (long)a.n * a.hv * a.wv;
    const long grid3 = ((M3 + CG_BM - 1) / CG_BM) * a.nblk;
    if (grid3 > 2147483647L) return DIS_ERR_BAD_SHAPE;
    DIS_TAG("convg3_fwd_kernel (bf16x3 streaming)");
    if (bn == 64) hipLaunchKernelGGL(convg3_fwd_kernel<64>, dim3((unsigned)grid3), dim3(256), 0, s, a);
    else if (bn == 32) hipLaunchKernelGGL(convg3_fwd_kernel<32>, dim3((unsigned)grid3), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(convg3_fwd_kernel<16>, dim3((unsigned)grid3), dim3(256), 0, s, a);
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  a.nchunk = (a.cin + CG_CK - 1) / CG_CK;
  a.tpack = (a.cin == 4 && a.ntaps > 4) ? 1 : 0;
  a.ntaps_real = a.ntaps;
  PackArgs p;
  for (int t = 0; t < a.ntaps; ++t) p.tsrc[t] = tsrc[t];
  if (a.tpack) a.ntaps = (a.ntaps_real + 3) / 4;
  p.w = w_raw; p.packed = wpack; p.ntaps = a.ntaps; p.nchunk = a.nchunk; p.nblk = a.nblk; p.bn = bn;
  p.ci_real = ci_real; p.co_real = co_real; p.s_ci = s_ci; p.s_co = s_co;
  p.tpack = a.tpack; p.ntaps_real = a.ntaps_real;
  const long ptotal = (long)a.ntaps * a.nchunk * a.nblk * 16 * bn;
  hipLaunchKernelGGL(convg_pack_kernel, dim3(dis_ew_grid(ptotal, 256)), dim3(256), 0, s, p);
  a.w = wpack;
  const long M = (long)a.n * a.hv * a.wv;
  const long grid = ((M + CG_BM - 1) / CG_BM) * a.nblk;
  if (grid > 2147483647L) return DIS_ERR_BAD_SHAPE;
  DIS_TAG("convg_fwd_kernel (fp32 MFMA streaming)");
  if (bn == 64) hipLaunchKernelGGL(convg_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (bn == 32) hipLaunchKernelGGL(convg_fwd_kernel<32>, dim3((unsigned)grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(convg_fwd_kernel<16>, dim3((unsigned)grid), dim3(256), 0, s, a);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// floats of split-K partial sums dis_convg_run may need BEHIND its packing slices (0: none); wpack must hold
// dis_convg_pack_workspace() x phases + this many floats (phases = 4 for the stride-2 input-gradient / transposed modes)
extern "C" long dis_convg_splitk_workspace(int mode, int n, int hin, int win, int hout, int wout, int cin, int cout, int k,
                                           int stride, int pad) {
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || cin <= 0 || cout <= 0 || k <= 0 || k * k > CG_MAXTAPS || pad < 0)
    return -1;
  if (cin < CG3_CK || n > CG2_NMAX) return 0;
  const long nchunk = (cin + CG3_CK - 1) / CG3_CK;
  auto need = [&](long hv, long wv, int ntaps) -> long {
    const long M = (long)n * hv * wv;
    if (M <= 0 || ntaps <= 0) return 0;
    const bool big = cg2_big_for(cin, cout, M, (int)(ntaps * nchunk));   // (the tile cg_run will pick for this launch)
    const int bn = big ? 128 : cg_bn(cout), bm = big ? 256 : CG_BM;
    const long nblk = (cout + bn - 1) / bn;
    const int ks = cg2_ksplit(((M + bm - 1) / bm) * nblk, (int)(ntaps * nchunk));
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

extern "C" long dis_convg_pack_workspace(int cin, int cout, int k) {
  if (cin <= 0 || cout <= 0 || k <= 0 || k * k > CG_MAXTAPS) return -1;
  const int bn = cg_bn(cout);
  const long nblk = (cout + bn - 1) / bn, nchunk = (cin + CG_CK - 1) / CG_CK, nchunk3 = (cin + CG3_CK - 1) / CG3_CK;
  const long f32 = (long)k * k * nchunk * nblk * 16 * bn;        // fp32 fragment image
  const long b3 = (long)k * k * nchunk3 * nblk * (3 * 4 * 8 / 2) * bn;  // 3 bf16 planes (16-bit words / 2 = floats)
  return (f32 > b3 ? f32 : b3) + CG2_WS;   // (+ the block-scale workspace of the two-term fp16 form, at the end of the first slice)
}

static int floordiv2(int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); }

// conv2d.hip: 3x3 stride-1 layers as 32 x 32 channel-slice launches of the halo-resident bf16x3 kernel
long dis_bx_slices_ok(int n, int h, int wd, int cin, int cout, int ldx, int ldy, int xoff, int yoff, int k, int stride,
                      int pad, int act);
int dis_bx_slices_run(int dgrad, const float* x, int ldx, int xoff, int cin, int cin_w, const float* w,
                      const float* bias, float* y, int ldy, int yoff, int cout, int cout_w, int n, int h, int wd, int k,
                      int act, hipStream_t stream);

extern "C" int dis_convg_run(int mode, const float* x, int ldx, int xoff, const float* w, const float* bias,
                             float* y, int ldy, int yoff, float* wpack, int n, int hin, int win, int cin,
                             int cin_w, int hout, int wout, int cout, int cout_w, int k, int stride, int pad,
                             int act, void* stream) {
  if (!x || !w || !y || !wpack) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || cin <= 0 || cout <= 0 || cin_w <= 0 ||
      cout_w <= 0 || k <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  if (k * k > CG_MAXTAPS || (stride != 1 && stride != 2) || mode < 0 || mode > 3) return DIS_ERR_UNSUPPORTED;
  if ((cin & 3) || (xoff & 3) || (ldx & 3) || xoff + cin > ldx || yoff + cout > ldy || cin_w > cin || cout_w > cout)
    return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  // (7x7 layers: the two-term halo kernel runs all 49 taps in one launch; the tap rows of the resident-weight kernel - seven
  // accumulating launches, each a pass over x and two over y - stay the three-term path's form)
  const bool halo7 = k == 7 && dis_f2_enabled() && cin >= CG3_CK && n <= CG2_NMAX && !getenv("DIS_CONVG_NO_HALO7");
  if ((mode == DIS_CONVG_CONV || mode == DIS_CONVG_CONV_DGRAD) && hin == hout && win == wout && !halo7 && !getenv("DIS_CONVG_NO_SLICES") &&
      dis_bx_slices_ok(n, hin, win, cin, cout, ldx, ldy, xoff, yoff, k, stride, pad, act))
    return dis_bx_slices_run(mode == DIS_CONVG_CONV_DGRAD, x, ldx, xoff, cin, cin_w, w, bias, y, ldy, yoff, cout, cout_w,
                             n, hin, win, k, act, s);
  GenArgs a;
  a.x = x; a.w = nullptr; a.bias = bias; a.y = y;
  a.n = n; a.hin = hin; a.win = win; a.ldx = ldx; a.xoff = xoff; a.cin = cin;
  a.hf = hout; a.wf = wout; a.ldy = ldy; a.yoff = yoff; a.cout = cout; a.act = act;
  short tsrc[CG_MAXTAPS];
  const long kk = (long)k * k;
  a.f2ws = nullptr; a.ksplit = 1; a.skpart = nullptr; a.skcap = 0; a.nph = 1;
  F2Maxima mxv = {};
  F2Maxima* mx = &mxv;
  if (dis_f2_enabled() && cin >= CG3_CK && n <= CG2_NMAX) {
    // (split-K partial sums: behind the packing slices, dis_convg_splitk_workspace floats)
    const int phases = ((mode == DIS_CONVG_CONV_DGRAD || mode == DIS_CONVG_TCONV) && stride == 2) ? 4 : 1;
    a.skcap = dis_convg_splitk_workspace(mode, n, hin, win, hout, wout, cin, cout, k, stride, pad);
    a.skpart = a.skcap > 0 ? wpack + dis_convg_pack_workspace(cin, cout, k) * phases : nullptr;
    // two-term fp16 form: block maxima of this call's x (per sample) and weights, one launch in front of the packing launch(es)
    float* f2ws = wpack + dis_convg_pack_workspace(cin, cout, k) - CG2_WS;
    mxv.x = x; mxv.n = n; mxv.hw = (long)hin * win; mxv.ldx = ldx; mxv.xoff = xoff; mxv.cin = cin;
    mxv.w = w; mxv.wcount = (long)cin_w * cout_w * kk; mxv.ws = f2ws;
    a.f2ws = f2ws;
  }
  if (mode == DIS_CONVG_CONV || mode == DIS_CONVG_TCONV_DGRAD) {
    // direct form: out[vy][vx] = sum_taps in[vy*S + ky - pad][vx*S + kx - pad] * W
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
    // conv: w[co][ci][ky][kx] (co rows of cin_w);  tconv dgrad: w[ci_t = out][co_t = in][ky][kx]
    const long s_ci = kk, s_co = (long)cin_w * kk;
    return cg_run(a, w, wpack, cin_w, cout_w, s_ci, s_co, tsrc, s, mx);
  }
  // transposed form: out[oy][ox] = sum_{ky,kx : parity} in[(oy + pad - ky)/S][(ox + pad - kx)/S] * W
  //   conv dgrad:  w[ci_in(= conv cout)][co_out(= conv cin)]  stored as w[conv_co][conv_ci][ky][kx]
  //   tconv fwd :  w[ci][co][ky][kx]
  // both: in-channel stride = cout_w*k*k, out-channel stride = k*k
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
    return cg_run(a, w, wpack, cin_w, cout_w, s_ci, s_co, tsrc, s, mx);
  }
  const long pstride = dis_convg_pack_workspace(cin, cout, k);  // every phase packs into its own slice
  if (a.f2ws && k * k <= 255 && !getenv("DIS_CONVG_NO_FUSED_PHASES")) {
    // the four parity classes in ONE launch of the halo kernel (they read the same input tile): taps listed class by class
    GenArgs b = a;
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
            b.tdy[nt] = (short)floordiv2(py + pad - ky);
            b.tdx[nt] = (short)floordiv2(px + pad - kx);
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
      const int rc = cg_run(b, w, wpack, cin_w, cout_w, s_ci, s_co, tsrc, s, mx, true);
      if (rc != DIS_ERR_UNSUPPORTED) return rc;
    }
  }
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      GenArgs b = a;
      b.hv = (hout - py + 1) / 2; b.wv = (wout - px + 1) / 2; b.S = 1;
      b.osy = 2; b.ooy = py; b.osx = 2; b.oox = px;
      int nt = 0;
      for (int ky = 0; ky < k; ++ky) {
        if ((py + pad - ky) & 1) continue;
        for (int kx = 0; kx < k; ++kx) {
          if ((px + pad - kx) & 1) continue;
          b.tdy[nt] = (short)floordiv2(py + pad - ky);
          b.tdx[nt] = (short)floordiv2(px + pad - kx);
          tsrc[nt] = (short)(ky * k + kx);
          ++nt;
        }
      }
      b.ntaps = nt;
      if (b.hv <= 0 || b.wv <= 0) continue;
      if (nt == 0) {
        // no tap reaches this parity class (cannot happen for k >= 2): outputs would be bias only
        return DIS_ERR_UNSUPPORTED;
      }
      int rc = cg_run(b, w, wpack + (long)(py * 2 + px) * pstride, cin_w, cout_w, s_ci, s_co, tsrc, s, mx);
      if (rc != DIS_OK) return rc;
    }
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgGenArgs {
  const float* X;
  const float* G;
  float* part;  // [ksplit][tap][cXp][cGp]
  int n, hX, wX, ldX, xoff, cX;
  int hG, wG, ldG, goff, cG;
  int S, pad, k;
  int nxb, ngb;
  int mper;  // G pixels per split (multiple of 32)
  int cXp, cGp;
};

template <int MTW, int NTW>
__global__ __launch_bounds__(256) void convg_wgrad_kernel(WgGenArgs a) {
  constexpr int TX = 32 * MTW, TG = 32 * NTW, XS = TX + 16, GS = TG + 16;
  constexpr int X_FL = 32 * XS, G_FL = 32 * GS;
  constexpr int NLX = TX / 32, NLG = TG / 32;  // float4 loads per thread per step
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
          v = *(const float4*)(a.X + (((long)nn * a.hX + iy) * a.wX + ix) * a.ldX + a.xoff + c);
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
      if (m < m_hi && c < a.cG) v = *(const float4*)(a.G + (long)m * a.ldG + a.goff + c);
      rg[j] = v;
    }
  };
  auto stage = [&](int buf) {
    float* Xt = smem + buf * (X_FL + G_FL);
    float* Gt = Xt + X_FL;
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
      const int item = tid + j * 256;
      const int px = item / (TX / 4), q = item % (TX / 4);
      *(float4*)(Xt + px * XS + q * 4) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < NLG; ++j) {
      const int item = tid + j * 256;
      const int px = item / (TG / 4), q = item % (TG / 4);
      *(float4*)(Gt + px * GS + q * 4) = rg[j];
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

// grad_w[(g*cXw + x)*k*k + tap] = sum_split part[split][tap][x][g]
__global__ void convg_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int nsplit, int ntaps,
                                          int cXp, int cGp, int cXw, int cGw) {
  const long total = (long)ntaps * cXw * cGw;
  const long slab = (long)ntaps * cXp * cGp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cGw);
    long r = i / cGw;
    const int x = (int)(r % cXw);
    const int tap = (int)(r / cXw);
    const float* p = part + ((long)tap * cXp + x) * cGp + g;
    double s = 0.0;  // fp64 over the split-K slabs (sums of ~1e5 signed per-pixel terms at full resolution)
    for (int k = 0; k < nsplit; ++k) s += (double)p[k * slab];
    gw[((long)g * cXw + x) * ntaps + tap] = (float)s;
  }
}

static void wg_tiles(int cX, int cG, int* mtw, int* ntw) {
  *mtw = cX > 32 ? 2 : 1;
  *ntw = cG > 32 ? 2 : 1;
}
static void wg_plan(int n, int hG, int wG, int cX, int cG, int k, int* nxb, int* ngb, int* nsplit, int* mper) {
  int mtw, ntw;
  wg_tiles(cX, cG, &mtw, &ntw);
  const int TX = 32 * mtw, TG = 32 * ntw;
  *nxb = (cX + TX - 1) / TX;
  *ngb = (cG + TG - 1) / TG;
  const long base = (long)k * k * (*nxb) * (*ngb);
  const long M = (long)n * hG * wG;
  long sp = (2048 + base - 1) / base;
  const long maxsp = (M + 255) / 256;  // at least 256 pixels (8 k-steps) per split
  if (sp > maxsp) sp = maxsp;
  if (sp < 1) sp = 1;
  long mp = (M + sp - 1) / sp;
  mp = (mp + 31) / 32 * 32;
  sp = (M + mp - 1) / mp;
  *nsplit = (int)sp;
  *mper = (int)mp;
}

// conv2d.hip: layers with >= 32 channels on both sides and 3x3 / 5x5 taps run as 32 x 32 channel-slice pairs on the
// one-pass bf16x3 kernel (x halo and gy tile staged once for all taps, accumulators in registers)
long dis_wgrad_pairs_workspace(int n, int hX, int wX, int hG, int wG, int cX, int cG, int ldX, int ldG, int k,
                               int stride, int bf);
int dis_wgrad_pairs_run(const float* X, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const float* G, int ldG,
                        int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace, int n, int k,
                        int stride, int pad, int bf, hipStream_t s);

// ---- weight gradient of a disparity head: ONE gradient channel (cG_w = 1), 3x3, stride 1 (fp32 twin of conv_bf16.hip's
// convb_head_wgrad_kernel) ----
// dW[tap][c] = sum_o x[o + tap - pad][c] * g[o] is a reduction with 9 cX outputs: no matrix core needed, HBM-bound on the ONE read
// of x (the generic kernel above takes one pass over x per tap: 1.0 GB for a 113 MB tensor, 0.34 ms for the 64-channel head at
// 128 x 108).  Thread = (8-channel chunk, pixel slot): per x pixel two 16-byte loads and 9 gradient scalars (neighbouring output
// pixels: L1 / L2), 72 register accumulators; lanes of a chunk are folded with xor-shuffles, waves through LDS, workgroups through
// [block][tap][cX] slabs finished by convg_head_reduce_kernel (fp64, fixed order: deterministic).
#define CGH_BLOCKS 512
template <int C8>
__global__ __launch_bounds__(256) void convg_head_wgrad_kernel(const float* __restrict__ X, int ldX, int xoff,
                                                               const float* __restrict__ G, int ldG, int goff,
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
    const float4 v0 = *(const float4*)(X + p * ldX + xoff + chunk * 8);
    const float4 v1 = *(const float4*)(X + p * ldX + xoff + chunk * 8 + 4);
    const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int oy = yy - ky + pad, ox = xx - kx + pad;
        float g = 0.f;
        if ((unsigned)oy < (unsigned)h && (unsigned)ox < (unsigned)w) g = G[(p + (long)(pad - ky) * w + (pad - kx)) * ldG + goff];
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
__global__ __launch_bounds__(256) void convg_head_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
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
// (cG <= 4: the padded gradient tensor of a disparity head - dis_convg_wgrad_workspace sizes the head form's slabs under exactly this
//  condition; pad == 1: the only padding with hX == hG for a 3 x 3 window)
static bool cgh_eligible(int cX, int cG, int cG_w, int k, int stride, int hX, int wX, int hG, int wG, int pad) {
  static const bool off = getenv("DIS_CONVG_HEAD_WGRAD") && getenv("DIS_CONVG_HEAD_WGRAD")[0] == '0';
  return !off && cG <= 4 && cG_w == 1 && k == 3 && stride == 1 && pad == 1 && hX == hG && wX == wG &&
         (cX == 16 || cX == 32 || cX == 64 || cX == 128);
}

extern "C" long dis_convg_wgrad_workspace(int n, int hG, int wG, int cX, int cG, int k) {
  if (n <= 0 || hG <= 0 || wG <= 0 || cX <= 0 || cG <= 0 || k <= 0 || k * k > CG_MAXTAPS) return -1;
  int nxb, ngb, nsplit, mper, mtw, ntw;
  wg_plan(n, hG, wG, cX, cG, k, &nxb, &ngb, &nsplit, &mper);
  wg_tiles(cX, cG, &mtw, &ntw);
  const long f32 = (long)nsplit * k * k * (nxb * 32 * mtw) * (ngb * 32 * ntw);
  // (the slice-pair form, if dis_convg_wgrad takes it for this layer: the stride is not known here, so size for both)
  const long b1 = dis_wgrad_pairs_workspace(n, 1, 1, hG, wG, cX, cG, 4, 4, k, 1, 0);
  const long b2 = dis_wgrad_pairs_workspace(n, 1, 1, hG, wG, cX, cG, 4, 4, k, 2, 0);
  long b3 = b1 > b2 ? b1 : b2;
  if (cG <= 4 && k == 3 && b3 < (long)CGH_BLOCKS * 9 * cX) b3 = (long)CGH_BLOCKS * 9 * cX;  // head form
  return b3 > f32 ? b3 : f32;
}

extern "C" int dis_convg_wgrad(const float* X, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const float* G,
                               int ldG, int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace,
                               int n, int k, int stride, int pad, void* stream) {
  if (!X || !G || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hX <= 0 || wX <= 0 || hG <= 0 || wG <= 0 || cX <= 0 || cG <= 0 || cX_w <= 0 || cG_w <= 0 ||
      cX_w > cX || cG_w > cG || k <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  if ((cX & 3) || (cG & 3) || (xoff & 3) || (goff & 3) || (ldX & 3) || (ldG & 3) || xoff + cX > ldX || goff + cG > ldG)
    return DIS_ERR_BAD_SHAPE;
  if (k * k > CG_MAXTAPS || (stride != 1 && stride != 2)) return DIS_ERR_UNSUPPORTED;
  if ((long)n * hG * wG > 2147483647L - 64) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (cgh_eligible(cX, cG, cG_w, k, stride, hX, wX, hG, wG, pad)) {   // a disparity head: one gradient channel, one pass over x
    const long npix = (long)n * hX * wX;
    const int c8 = cX / 8;
    long blocks = (npix + 256 / c8 - 1) / (256 / c8);
    if (blocks > CGH_BLOCKS) blocks = CGH_BLOCKS;
    DIS_TAG("convg_head_wgrad_kernel (one gradient channel)");
#define CGH_CASE(C8_) \
    if (c8 == C8_) hipLaunchKernelGGL((convg_head_wgrad_kernel<C8_>), dim3((unsigned)blocks), dim3(256), 0, s, X, ldX, xoff, G, ldG, goff, \
                                      workspace, n, hX, wX, pad);
    CGH_CASE(2) CGH_CASE(4) CGH_CASE(8) CGH_CASE(16)
#undef CGH_CASE
    hipLaunchKernelGGL(convg_head_reduce_kernel, dim3((9 * cX_w + 3) / 4), dim3(256), 0, s, (const float*)workspace, grad_w,
                       (int)blocks, cX, cX_w);
    DIS_CHECK_LAUNCH();
    return DIS_OK;
  }
  if (dis_wgrad_pairs_workspace(n, hX, wX, hG, wG, cX, cG, ldX, ldG, k, stride, 0) >= 0)
    return dis_wgrad_pairs_run(X, ldX, xoff, hX, wX, cX, cX_w, G, ldG, goff, hG, wG, cG, cG_w, grad_w, workspace, n, k,
                               stride, pad, 0, s);
  WgGenArgs a;
  a.X = X; a.G = G; a.part = workspace;
  a.n = n; a.hX = hX; a.wX = wX; a.ldX = ldX; a.xoff = xoff; a.cX = cX;
  a.hG = hG; a.wG = wG; a.ldG = ldG; a.goff = goff; a.cG = cG;
  a.S = stride; a.pad = pad; a.k = k;
  int nsplit, mtw, ntw;
  wg_plan(n, hG, wG, cX, cG, k, &a.nxb, &a.ngb, &nsplit, &a.mper);
  wg_tiles(cX, cG, &mtw, &ntw);
  a.cXp = a.nxb * 32 * mtw;
  a.cGp = a.ngb * 32 * ntw;
  const dim3 grid((unsigned)(k * k * a.nxb * a.ngb), (unsigned)nsplit);
  DIS_TAG("convg_wgrad_kernel (fp32 MFMA streaming)");
  if (mtw == 2 && ntw == 2) hipLaunchKernelGGL((convg_wgrad_kernel<2, 2>), grid, dim3(256), 0, s, a);
  else if (mtw == 2) hipLaunchKernelGGL((convg_wgrad_kernel<2, 1>), grid, dim3(256), 0, s, a);
  else if (ntw == 2) hipLaunchKernelGGL((convg_wgrad_kernel<1, 2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((convg_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, a);
  const long total = (long)k * k * cX_w * cG_w;
  hipLaunchKernelGGL(convg_wgrad_reduce_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s,
                     (const float*)workspace, grad_w, nsplit, k * k, a.cXp, a.cGp, cX_w, cG_w);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[c] = sum_pixels G[pixel][goff + c], deterministic two-level sum
// ------------------------------------------------------------------------------------------------
#define CS_BLOCKS 2048
__global__ __launch_bounds__(256) void colsum1_kernel(const float* __restrict__ G, int ldG, int goff, long npix, int c,
                                                       float* __restrict__ part) {
  __shared__ float red[256];
  const long lo = npix * blockIdx.x / gridDim.x, hi = npix * (blockIdx.x + 1) / gridDim.x;
  for (int c0 = 0; c0 < c; c0 += 256) {
    const int cw = min(256, c - c0);          // channels handled in this pass
    const int rows = 256 / cw > 0 ? 256 / cw : 1;
    const int ch = threadIdx.x % cw, row = threadIdx.x / cw;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // 4 independent loads in flight
    if (row < rows) {
      const float* gp = G + goff + c0 + ch;
      long p = lo + row;
      for (; p + 3L * rows < hi; p += 4L * rows) {
        s0 += gp[p * ldG];
        s1 += gp[(p + rows) * ldG];
        s2 += gp[(p + 2L * rows) * ldG];
        s3 += gp[(p + 3L * rows) * ldG];
      }
      for (; p < hi; p += rows) s0 += gp[p * ldG];
    }
    __syncthreads();
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < cw) {
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += red[k * cw + threadIdx.x];
      part[(long)blockIdx.x * c + c0 + threadIdx.x] = t;
    }
  }
}
// one wave per channel: lanes stride over the block partials, fixed-order wave reduction
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ part, int nblocks, int c,
                                                       float* __restrict__ out) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double s = 0.0;
  for (int k = lane; k < nblocks; k += 64) s += (double)part[(long)k * c + ch];
  s = wave_sum_d(s);
  if (lane == 0) out[ch] = (float)s;
}

extern "C" long dis_colsum_workspace(int c) { return c > 0 ? (long)CS_BLOCKS * c : -1; }

extern "C" int dis_colsum(const float* G, int ldG, int goff, long npix, int c, float* out, float* workspace,
                          void* stream) {
  if (!G || !out || !workspace) return DIS_ERR_NULL;
  if (npix <= 0 || c <= 0 || goff < 0 || goff + c > ldG) return DIS_ERR_BAD_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  long want = npix / 64;  // >= 64 pixels per block
  int nb = (int)(want < 1 ? 1 : (want > CS_BLOCKS ? CS_BLOCKS : want));
  hipLaunchKernelGGL(colsum1_kernel, dim3(nb), dim3(256), 0, s, G, ldG, goff, npix, c, workspace);
  hipLaunchKernelGGL(colsum2_kernel, dim3(dis_cdiv(c, 4)), dim3(256), 0, s, (const float*)workspace, nb, c, out);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// SigmoidAffine of the disparity heads (reference model/networks.py:140-149): y = alpha*sigmoid(x - offset)
// ------------------------------------------------------------------------------------------------
__global__ void sigmoid_affine_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, float offset,
                                          long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    y[i] = alpha / (1.f + expf(-(x[i] - offset)));
}
// gpre4 (count,4): channel 0 = gy * alpha * s * (1 - s), channels 1..3 = 0 (padded layout for the conv kernels)
__global__ void sigmoid_affine_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                          float4* __restrict__ gpre4, float alpha, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float sg = y[i] / alpha;
    gpre4[i] = make_float4(gy[i] * alpha * sg * (1.f - sg), 0.f, 0.f, 0.f);
  }
}
extern "C" int dis_sigmoid_affine_fwd(const float* x, float* y, float alpha, float offset, long count, void* stream) {
  if (!x || !y) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sigmoid_affine_fwd_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     alpha, offset, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" int dis_sigmoid_affine_bwd(const float* y, const float* gy, float* gpre4, float alpha, long count,
                                      void* stream) {
  if (!y || !gy || !gpre4) return DIS_ERR_NULL;
  if (count <= 0) return DIS_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(sigmoid_affine_bwd_kernel, dim3(dis_ew_grid(count, 256)), dim3(256), 0, (hipStream_t)stream, y,
                     gy, (float4*)gpre4, alpha, count);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
