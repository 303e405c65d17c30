// Implicit-GEMM 2-D convolution on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, 64 FLOP/clk/SIMD).
//
// Layout: activations nhwc.  GEMM view per output tile:  D[pixel][cout] += A[pixel][k] * B[k][cout],
// k = (tap, cin).  One MFMA consumes 4 k's, one per 16-lane group g; group g owns the channel range
// [g*KG, (g+1)*KG) of the current 32/16/4-channel chunk, so every lane reads CONTIGUOUS channels of its
// pixel from LDS (ds_read_b128).  A workgroup (4 waves) owns a TROWS x 16 pixel tile; wave w owns rows
// [w*MT, (w+1)*MT); each wave keeps MT x (COUT/16) accumulator tiles.
//
//   * all weights of the layer stay resident in LDS in exactly the order the B fragments are read
//     (packed once per step by dis_conv2d_pack_weights); workgroups are persistent over tiles;
//   * the input halo tile of the NEXT (tile, chunk) is fetched into registers while the MFMAs of the
//     current one run (issue-early / write-late staging);
//   * tiles are dealt to workgroups in contiguous ranges per XCD (blockIdx % 8) so halos are re-read
//     from that XCD's L2;
//   * epilogue fuses bias, SELU/ReLU and the GroupNorm(1 group) sum / sum-of-squares of the output.
//
// The same kernel computes stride-1 input gradients (flipped/transposed weights) and, with a 2x2 tap
// set and interleaved output addressing, the four phases of the stride-2 transposed convolution.
#include "conv_args.h"
#include <stdlib.h>
#include <type_traits>


template <int CIN, int COUT, int KH, int KW, int S>
struct ConvCfg {
  static constexpr int CINB = (CIN % 32 == 0) ? 32 : ((CIN % 16 == 0) ? 16 : 4);
  static_assert(CIN % CINB == 0, "cin must be a multiple of the chunk");
  static_assert(COUT % 16 == 0, "cout must be a multiple of 16");
  static constexpr int NCHUNK = CIN / CINB;
  static constexpr int KG = CINB / 4;                 // channels per MFMA k-group
  static constexpr int E = (KG >= 4) ? 4 : KG;        // floats per fragment read
  static constexpr int NQ = KG / E;                   // fragment reads per tap
  static constexpr int NT = COUT / 16;
  static constexpr int MT = 1;  // one 16-pixel row per wave: 4x16 tiles keep the halo tile small enough for 3+ workgroups per CU
  static constexpr int TROWS = 4 * MT;
  static constexpr int TCOLS = 16;
  static constexpr int IN_ROWS = (TROWS - 1) * S + KH;
  static constexpr int IN_COLS = (TCOLS - 1) * S + KW;
  static constexpr int CS = (CINB >= 16) ? CINB + 4 : CINB;  // padded pixel stride in LDS (floats)
  static constexpr int W_FLOATS = KH * KW * CIN * COUT;
  static constexpr int IN_FLOATS = IN_ROWS * IN_COLS * CS;
  static constexpr int RED_FLOATS = 16;  // 8 doubles for the stats reduction
  static constexpr int LDS_BYTES = (W_FLOATS + IN_FLOATS + RED_FLOATS) * 4;
  static constexpr int NV = CINB / 4;  // float4 per pixel per chunk
  static constexpr int NITEMS = IN_ROWS * IN_COLS * NV;
  static constexpr int NLOAD = (NITEMS + 255) / 256;
};

// packed-weight offset of (tap, channel c, cout)
__host__ __device__ inline long packed_w_offset(int tap, int c, int co, int cin, int cout) {
  const int cinb = (cin % 32 == 0) ? 32 : ((cin % 16 == 0) ? 16 : 4);
  const int nchunk = cin / cinb, kg = cinb / 4, e = kg >= 4 ? 4 : kg, nq = kg / e;
  const int chunk = c / cinb, cl = c % cinb;
  const int g = cl / kg, r = cl % kg, q = r / e, ee = r % e;
  return ((((long)(tap * nchunk + chunk) * 4 + g) * nq + q) * cout + co) * e + ee;
}

// Sum of squares of the outputs (GroupNorm statistics): a fused multiply-add per value.  Written as `t1 += v; t2 += v * v` the
// compiler packs the two accumulators into one register pair and emits v_mul_f32 v[n+1], v, v ; v_pk_add_f32 acc[0:1], acc[0:1],
// v[n:n+1] - and that sequence LOSES one half's addend in lanes 48..63 of a wave in ~17 % of the launches once other processes
// time-share the GPU (never alone): scripts/diag/share_repro.hip isolates it (per-thread sums dumped: every deviating thread is in
// lanes 48..63, one of its two sums short by exactly one term, outputs bit-identical), and with this form the count is 0 of
// 8000.  DESIGN.md section 4.  (-DDIS_STATS_PK_ORIG restores the old form for the reproducer.)
#ifdef DIS_STATS_PK_ORIG
#define DIS_T2_ACC(t2, v) t2 += (v) * (v)
#else
#define DIS_T2_ACC(t2, v) t2 = __builtin_fmaf(v, v, t2)
#endif

// GNB (1 x 1 only, round 5): x is the gradient g wrt the OUTPUT of a GroupNorm whose input q = a.xact was this conv's output: the
// tile is staged as act'(q) (g k1_c + q kx + k0) - the elementwise pass of the GroupNorm backward, a.gnb_coef (n, CIN + 2) from
// dis_gn_bwd_coef, gn_apply_coef_kernel's arithmetic bit for bit - and those values are stored to a.gnb_out for the layer's
// weight-gradient launch (a 1 x 1 window has no halo: every staged pixel belongs to exactly one tile).
template <int CIN, int COUT, int KH, int KW, int S, bool GNB = false>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
  using C = ConvCfg<CIN, COUT, KH, KW, S>;
  static_assert(!GNB || (KH == 1 && KW == 1 && S == 1 && C::NCHUNK == 1 && 256 % C::NV == 0), "GroupNorm backward on load: 1 x 1");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
  float* xl = smem + C::W_FLOATS;
  double* red = (double*)(smem + C::W_FLOATS + C::IN_FLOATS);

  for (int i = threadIdx.x; i < C::W_FLOATS / 4; i += 256) ((float4*)wl)[i] = ((const float4*)a.w)[i];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.wv + C::TCOLS - 1) / C::TCOLS, tiles_y = (a.hv + C::TROWS - 1) / C::TROWS;
  const int ntiles = a.n * tiles_y * tiles_x;

  // XCD-aware persistent schedule: workgroups with equal blockIdx%8 share an XCD (speed only)
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);

  // Halo prefetch.  Loads are unconditional (out-of-image taps read a clamped, valid address) and are zeroed only
  // when they are written to LDS: a load under a branch made hipcc wait for it (s_waitcnt vmcnt(0)) right at the
  // issue point.  Everything that does not depend on the tile (item -> row/col/channel-group, element offset
  // inside the halo window) is computed once per thread; interior tiles skip the per-item clamps.
  float4 pre[C::NLOAD], preq[GNB ? C::NLOAD : 1];
  float psc[C::NLOAD];
  unsigned okmask = 0;
  int st_n = 0, st_iy0 = 0, st_ix0 = 0, cf_n = -1;   // GNB: the prefetched tile, the sample whose coefficients are loaded
  float4 cf_k1 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cf_kx = 0.f, cf_k0 = 0.f;
  int it_r[C::NLOAD], it_c[C::NLOAD], it_off[C::NLOAD], it_lds[C::NLOAD];
#pragma unroll
  for (int it = 0; it < C::NLOAD; ++it) {
    const int idx = min((int)threadIdx.x + it * 256, C::NITEMS - 1);
    const int vv = idx % C::NV, pix = idx / C::NV;
    it_c[it] = pix % C::IN_COLS;
    it_r[it] = pix / C::IN_COLS;
    it_off[it] = (it_r[it] * a.win + it_c[it]) * CIN + vv * 4;
    it_lds[it] = pix * C::CS + vv * 4;
    psc[it] = 1.f;
  }
  auto prefetch = [&](int tile, int chunk) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * C::TROWS * S - a.pad_y, ix0 = tx * C::TCOLS * S - a.pad_x;
    const float* xb = a.x + (long)n * a.hin * a.win * CIN + chunk * C::CINB;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + C::IN_ROWS <= a.hin && ix0 + C::IN_COLS <= a.win;
    if (GNB) st_n = n, st_iy0 = iy0, st_ix0 = ix0;
    if (interior) {
      const float* xo = xb + ((long)iy0 * a.win + ix0) * CIN;
      okmask = 0xffffffffu;
#pragma unroll
      for (int it = 0; it < C::NLOAD; ++it) pre[it] = *(const float4*)(xo + it_off[it]);
      if (GNB) {
#pragma unroll
        for (int it = 0; it < C::NLOAD; ++it) preq[it] = *(const float4*)(a.xact + (xo - a.x) + it_off[it]);
      }
    } else {
      okmask = 0;
#pragma unroll
      for (int it = 0; it < C::NLOAD; ++it) {
        const int iy = iy0 + it_r[it], ix = ix0 + it_c[it];
        const bool ok = iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win;
        const int cy = min(max(iy, 0), a.hin - 1), cx = min(max(ix, 0), a.win - 1);
        pre[it] = *(const float4*)(xb + ((long)cy * a.win + cx) * CIN + it_lds[it] % C::CS);
        if (GNB) preq[it] = *(const float4*)(a.xact + (xb - a.x) + ((long)cy * a.win + cx) * CIN + it_lds[it] % C::CS);
        okmask |= (ok ? 1u : 0u) << it;
      }
    }
    if (a.xscale) {  // per (input pixel, chunk) multiplier, e.g. the slot weights of the multi-frame 1x1 conv
#pragma unroll
      for (int it = 0; it < C::NLOAD; ++it) {
        const int cy = min(max(iy0 + it_r[it], 0), a.hin - 1), cx = min(max(ix0 + it_c[it], 0), a.win - 1);
        psc[it] = a.xscale[(((long)n * a.hin + cy) * a.win + cx) * C::NCHUNK + chunk];
      }
    }
  };
  auto stage = [&]() {
    if (GNB && st_n != cf_n) {   // (a thread's items share its channel group: 256 % NV == 0)
      cf_n = st_n;
      const float* cf = a.gnb_coef + (long)st_n * (CIN + 2);
      cf_k1 = *(const float4*)(cf + ((int)threadIdx.x % C::NV) * 4);
      cf_kx = cf[CIN];
      cf_k0 = cf[CIN + 1];
    }
#pragma unroll
    for (int it = 0; it < C::NLOAD; ++it) {
      if ((int)threadIdx.x + it * 256 < C::NITEMS) {
        const bool ok = (okmask >> it) & 1u;
        const float sc = psc[it];
        if (GNB) {
          const float4 q = preq[it];
          float4 v = pre[it];
          v.x = __builtin_fmaf(v.x, cf_k1.x, __builtin_fmaf(q.x, cf_kx, cf_k0));
          v.y = __builtin_fmaf(v.y, cf_k1.y, __builtin_fmaf(q.y, cf_kx, cf_k0));
          v.z = __builtin_fmaf(v.z, cf_k1.z, __builtin_fmaf(q.z, cf_kx, cf_k0));
          v.w = __builtin_fmaf(v.w, cf_k1.w, __builtin_fmaf(q.w, cf_kx, cf_k0));
          if (a.gnb_act != DIS_ACT_NONE) {
            v.x *= act_grad_from_out(q.x, a.gnb_act), v.y *= act_grad_from_out(q.y, a.gnb_act);
            v.z *= act_grad_from_out(q.z, a.gnb_act), v.w *= act_grad_from_out(q.w, a.gnb_act);
          }
          pre[it] = v;
          if (ok)
            *(float4*)(a.gnb_out + (((long)st_n * a.hin + (st_iy0 + it_r[it])) * a.win + (st_ix0 + it_c[it])) * CIN +
                       it_lds[it] % C::CS) = v;
        }
        *(float4*)(xl + it_lds[it]) =
            ok ? make_float4(pre[it].x * sc, pre[it].y * sc, pre[it].z * sc, pre[it].w * sc) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  int tile = t_lo + rank;
  if (tile < t_hi) prefetch(tile, 0);
  f32x4 acc[C::MT][C::NT];
  constexpr bool YS = COUT == 128;   // four 32-channel groups: the multipliers of a pixel are one float4, fetched a matrix loop ahead
  float4 ysc[C::MT][4];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ysc[mt][r] = make_float4(1.f, 1.f, 1.f, 1.f);
  double s1 = 0.0, s2 = 0.0;
  int stat_n = -1;
  float bias_v[C::NT];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) bias_v[nt] = a.bias ? a.bias[nt * 16 + li] : 0.f;
  const long ypix = (long)a.osx * COUT;  // floats between horizontally adjacent outputs of this launch

  while (tile < t_hi) {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
      __syncthreads();  // everyone finished reading the previous LDS tile (and the weights are in)
      stage();
      __syncthreads();
      // issue the next halo tile's loads now; they land while the MFMAs below run
      if (chunk + 1 < C::NCHUNK) prefetch(tile, chunk + 1);
      else if (tile + per < t_hi) prefetch(tile + per, 0);
      // ... and this tile's output multipliers (loaded in the epilogue, right in front of their use, every tile waited for them)
      if (YS && a.yscale && chunk == C::NCHUNK - 1) {
        const int tx_ = tile % tiles_x, ty_ = (tile / tiles_x) % tiles_y, n_ = tile / (tiles_x * tiles_y);
        const int vy_ = ty_ * C::TROWS + wave * C::MT, vx_ = tx_ * C::TCOLS + lg * 4;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int oy = min(vy_ + mt, a.hv - 1) * a.osy + a.ooy, ox = min(vx_ + r, a.wv - 1) * a.osx + a.oox;
            ysc[mt][r] = *(const float4*)(a.yscale + (((long)n_ * a.hf + oy) * a.wf + ox) * 4);
          }
      }

      // software-pipelined fragment reads: the LDS reads of step s+1 are issued before the MFMAs of step s
      constexpr int NS = KH * KW * C::NQ;
      float av[2][C::MT][4], bv[2][C::NT][4];
      auto load_frag = [&](int s, float (&fa)[C::MT][4], float (&fb)[C::NT][4]) {
        const int q = s % C::NQ, tap = s / C::NQ, ky = tap / KW, kx = tap % KW;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
          const int row = wave * C::MT + mt;
          const float* p = xl + (((row * S + ky) * C::IN_COLS) + (li * S + kx)) * C::CS + lg * C::KG + q * C::E;
          if (C::E == 4) {
            const f32x4 t = *(const f32x4*)p;
            fa[mt][0] = t[0]; fa[mt][1] = t[1]; fa[mt][2] = t[2]; fa[mt][3] = t[3];
          } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) fa[mt][e] = p[e];
          }
        }
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const float* p = wl + ((((long)(tap * C::NCHUNK + chunk) * 4 + lg) * C::NQ + q) * COUT + nt * 16 + li) * C::E;
          if (C::E == 4) {
            const f32x4 t = *(const f32x4*)p;
            fb[nt][0] = t[0]; fb[nt][1] = t[1]; fb[nt][2] = t[2]; fb[nt][3] = t[3];
          } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) fb[nt][e] = p[e];
          }
        }
      };
      load_frag(0, av[0], bv[0]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) load_frag(s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
          for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1][mt][e], bv[s & 1][nt][e], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // epilogue: bias + activation + store (+ GroupNorm statistics).  The statistics are summed in fp32 per lane and
    // tile (16 values), then carried in fp64 registers across the consecutive tiles of one sample; the block
    // reduction + 2 fp64 atomics happen only when the sample changes (or at the end), not per tile.
    // The activation is dispatched once per tile (not per element) and interior tiles store without bounds checks.
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    if (a.stats && n != stat_n) {
      if (stat_n >= 0) {
        const double r1 = block_sum_d(s1, red);
        const double r2 = block_sum_d(s2, red);
        if (threadIdx.x == 0) {
          atomic_add_d(a.stats + 2 * stat_n, r1);
          atomic_add_d(a.stats + 2 * stat_n + 1, r2);
        }
      }
      stat_n = n;
      s1 = 0.0;
      s2 = 0.0;
    }
    float t1 = 0.f, t2 = 0.f;
    const int vy0 = ty * C::TROWS + wave * C::MT, vx0 = tx * C::TCOLS + lg * 4;
    float* ybase = a.y + (((long)n * a.hf + ((long)vy0 * a.osy + a.ooy)) * a.wf + ((long)vx0 * a.osx + a.oox)) * COUT + li;
    const long yrow = (long)a.osy * a.wf * COUT;
    const bool full = (ty + 1) * C::TROWS <= a.hv && (tx + 1) * C::TCOLS <= a.wv;
    if (a.yscale) {  // per (output pixel, 32-channel group) multiplier (input gradient of a scaled-input 1x1 conv)
      constexpr int NG = (C::NT + 1) / 2;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (YS) {
            const float sv[4] = {ysc[mt][r].x, ysc[mt][r].y, ysc[mt][r].z, ysc[mt][r].w};
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt][r] *= sv[(nt >> 1) & 3];
          } else {
            const int oy = min(vy0 + mt, a.hv - 1) * a.osy + a.ooy, ox = min(vx0 + r, a.wv - 1) * a.osx + a.oox;
            const float* sp = a.yscale + (((long)n * a.hf + oy) * a.wf + ox) * NG;
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) acc[mt][nt][r] *= sp[nt >> 1];
          }
        }
    }
    auto emit = [&](auto actc, auto accc) {
      constexpr int ACT = decltype(actc)::value;
      constexpr bool ACC = decltype(accc)::value != 0;
      if (full) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) {
              float* yp = ybase + mt * yrow + r * ypix + nt * 16;
              float pre_v = acc[mt][nt][r] + bias_v[nt];
              if (ACC) pre_v += *yp;
              const float v = act_apply(pre_v, ACT);
              *yp = v;
              t1 += v;
              DIS_T2_ACC(t2, v);
            }
      } else {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (vy0 + mt < a.hv && vx0 + r < a.wv) {
#pragma unroll
              for (int nt = 0; nt < C::NT; ++nt) {
                float* yp = ybase + mt * yrow + r * ypix + nt * 16;
                float pre_v = acc[mt][nt][r] + bias_v[nt];
                if (ACC) pre_v += *yp;
                const float v = act_apply(pre_v, ACT);
                *yp = v;
                t1 += v;
                DIS_T2_ACC(t2, v);
              }
            }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (a.accum) {
      if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{}, I1{});
      else if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{}, I1{});
      else emit(std::integral_constant<int, DIS_ACT_NONE>{}, I1{});
    } else {
      if (a.act == DIS_ACT_SELU) emit(std::integral_constant<int, DIS_ACT_SELU>{}, I0{});
      else if (a.act == DIS_ACT_RELU) emit(std::integral_constant<int, DIS_ACT_RELU>{}, I0{});
      else emit(std::integral_constant<int, DIS_ACT_NONE>{}, I0{});
    }
    s1 += (double)t1;
    s2 += (double)t2;
    tile += per;
  }
  if (a.stats && stat_n >= 0) {
#ifdef DIS_STATS_DUMP  // diagnostic build (scripts/diag/share_repro.hip): every thread's sums, behind the n x 2 statistics
    a.stats[2 * a.n + 2 * (blockIdx.x * 256 + threadIdx.x)] = s1;
    a.stats[2 * a.n + 2 * (blockIdx.x * 256 + threadIdx.x) + 1] = s2;
#endif
    const double r1 = block_sum_d(s1, red);
    const double r2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
      atomic_add_d(a.stats + 2 * stat_n, r1);
      atomic_add_d(a.stats + 2 * stat_n + 1, r2);
#ifdef DIS_STATS_DUMP
      a.stats[2 * a.n + 2 * 256 * gridDim.x + 2 * blockIdx.x] = r1;
      a.stats[2 * a.n + 2 * 256 * gridDim.x + 2 * blockIdx.x + 1] = r2;
#endif
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing
//   mode 0: forward          w'(tap=(ky,kx), c=ci, co)            = w[co][ci][ky][kx]
//   mode 1: stride-1 dgrad   w'(tap=(ky,kx), c=co_orig, co'=ci)   = w[co_orig][ci][K-1-ky][K-1-kx]
//   mode 2: stride-2 k4 dgrad phase (py,px), 2x2 taps (dy,dx):    = w[co_orig][ci][3-py-2dy][3-px-2dx]
// `cin_real` <= cin_pad: extra input channels of the padded layout get zero weights (mode 0 only).
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout, int cin_real,
                                    int cin_pad, int k, int mode, int py, int px) {
  const int kk = (mode == 2) ? 2 : k;
  const int ncin = (mode == 0) ? cin_pad : cout;   // channels of the packed conv's input
  const int ncout = (mode == 0) ? cout : cin_real;  // and output
  const long total = (long)kk * kk * ncin * ncout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % ncout);
    long r = i / ncout;
    const int c = (int)(r % ncin);
    const int tap = (int)(r / ncin);
    const int ky = tap / kk, kx = tap % kk;
    float v;
    if (mode == 0) {
      v = (c < cin_real) ? w[(((long)co * cin_real + c) * k + ky) * k + kx] : 0.f;
    } else if (mode == 1) {
      v = w[(((long)c * cin_real + co) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
    } else {
      v = w[(((long)c * cin_real + co) * k + (3 - py - 2 * ky)) * k + (3 - px - 2 * kx)];
    }
    packed[packed_w_offset(tap, c, co, ncin, ncout)] = v;
  }
}

extern "C" int dis_conv2d_pack_weights(const float* w_oihw, float* packed, int cout, int cin_real, int cin_pad,
                                       int k, int mode, void* stream) {
  if (!w_oihw || !packed) return DIS_ERR_NULL;
  if (cout <= 0 || cin_real <= 0 || cin_pad < cin_real || k <= 0) return DIS_ERR_BAD_SHAPE;
  if (mode < 0 || mode > 1) return DIS_ERR_UNSUPPORTED;
  if (mode == 1 && cin_pad != cin_real) return DIS_ERR_UNSUPPORTED;
  long total = (long)k * k * cin_pad * cout;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     packed, cout, cin_real, cin_pad, k, mode, 0, 0);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// launch plumbing
// ------------------------------------------------------------------------------------------------
static int g_num_cu = 0;
static int num_cus() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

template <int CIN, int COUT, int KH, int KW, int S, bool GNB = false>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  using C = ConvCfg<CIN, COUT, KH, KW, S>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  static bool attr_set = false;
  auto kern = conv_fwd_kernel<CIN, COUT, KH, KW, S, GNB>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles_x = (a.wv + C::TCOLS - 1) / C::TCOLS, tiles_y = (a.hv + C::TROWS - 1) / C::TROWS;
  const long ntiles = (long)a.n * tiles_y * tiles_x;
  int per_cu = (160 * 1024) / C::LDS_BYTES;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long grid = (long)num_cus() * per_cu;
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  DIS_TAG("conv_fwd_kernel (fp32 MFMA)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, s, a);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

#define CONV_CASE(CI, CO, KH_, KW_, S_)                                      \
  if (cin == CI && cout == CO && kh == KH_ && kw == KW_ && stride == S_) \
    return launch_conv<CI, CO, KH_, KW_, S_>(a, s);

static int dispatch_conv(const ConvArgs& a, int cin, int cout, int kh, int kw, int stride, hipStream_t s) {
  CONV_CASE(4, 16, 4, 4, 2)
  CONV_CASE(4, 16, 3, 3, 1)
  CONV_CASE(16, 16, 3, 3, 1)
  CONV_CASE(16, 32, 3, 3, 1)
  CONV_CASE(32, 16, 3, 3, 1)
  CONV_CASE(32, 32, 3, 3, 1)
  CONV_CASE(48, 32, 3, 3, 1)
  CONV_CASE(32, 48, 3, 3, 1)
  CONV_CASE(96, 32, 3, 3, 1)
  CONV_CASE(32, 96, 3, 3, 1)
  CONV_CASE(128, 32, 1, 1, 1)
  CONV_CASE(32, 128, 1, 1, 1)
  CONV_CASE(32, 32, 4, 4, 2)
  CONV_CASE(32, 32, 2, 2, 1)
  return DIS_ERR_UNSUPPORTED;
}

extern "C" int dis_conv2d_fwd(const float* x, const float* w_packed, const float* bias, float* y, double* stats,
                              int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act,
                              void* stream) {
  if (!x || !w_packed || !y) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || cin <= 0 || cout <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.stats = stats;
  a.n = n; a.hin = hin; a.win = win; a.hv = hout; a.wv = wout; a.pad_y = pad; a.pad_x = pad;
  a.hf = hout; a.wf = wout; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
  a.act = act & 0xff;
  a.accum = (act & DIS_CONV_ACCUM) ? 1 : 0;
  a.xscale = nullptr;
  a.yscale = nullptr;
  a.xact = nullptr; a.gnb_coef = nullptr; a.gnb_out = nullptr; a.gnb_act = 0;
  if (a.act > DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  return dispatch_conv(a, cin, cout, k, k, stride, (hipStream_t)stream);
}

extern "C" int dis_conv2d_fwd_scaled(const float* x, const float* xscale, const float* w_packed, const float* bias,
                                     float* y, const float* yscale, double* stats, int n, int hin, int win, int cin,
                                     int cout, int k, int stride, int pad, int act, void* stream) {
  if (!x || !w_packed || !y) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || cin <= 0 || cout <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return DIS_ERR_BAD_SHAPE;
  if (yscale && (cout % 32)) return DIS_ERR_UNSUPPORTED;
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.stats = stats;
  a.n = n; a.hin = hin; a.win = win; a.hv = hout; a.wv = wout; a.pad_y = pad; a.pad_x = pad;
  a.hf = hout; a.wf = wout; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
  a.act = act & 0xff;
  a.accum = (act & DIS_CONV_ACCUM) ? 1 : 0;
  a.xscale = xscale;
  a.yscale = yscale;
  a.xact = nullptr; a.gnb_coef = nullptr; a.gnb_out = nullptr; a.gnb_act = 0;
  if (a.act > DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  return dispatch_conv(a, cin, cout, k, k, stride, (hipStream_t)stream);
}

/* Input gradient of the 1 x 1 multi-frame convolution (128 -> 32, slot weights applied to its input: reference
 * model/multi_frame_networks.py:406-413) that is followed by GroupNorm(1 group), WITH that GroupNorm's backward elementwise pass
 * applied while the operand is staged (round 5): gpre = act'(q) (g k1_c + q kx + k0) with g the gradient wrt the GroupNorm's output
 * (n, h, w, 32), q the GroupNorm's input (this conv's output), coef (n, 34) from dis_gn_bwd_coef; gx (n, h, w, 128) (+)= (gpre W^T)
 * times yscale (n, h, w, 4; may be NULL); gpre is stored to gpre_out for dis_conv2d_wgrad_scaled.  w_packed: the mode-1 packing
 * of the conv's weight (dis_conv2d_pack_weights).  Replaces dis_gn_bwd_apply_coef + dis_conv2d_fwd_scaled. */
extern "C" int dis_conv2d_dgrad1x1_scaled_gnb(const float* g, const float* q, const float* coef, int in_act, float* gpre_out,
                                              const float* w_packed, float* gx, const float* yscale, int n, int hin, int win,
                                              int cin, int cout, int accumulate, void* stream) {
  if (!g || !q || !coef || !gpre_out || !w_packed || !gx) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0) return DIS_ERR_BAD_SHAPE;
  if (cin != 32 || cout != 128 || (in_act != DIS_ACT_NONE && in_act != DIS_ACT_SELU)) return DIS_ERR_UNSUPPORTED;
  ConvArgs a;
  a.x = g; a.w = w_packed; a.bias = nullptr; a.y = gx; a.stats = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hv = hin; a.wv = win; a.pad_y = 0; a.pad_x = 0;
  a.hf = hin; a.wf = win; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
  a.act = DIS_ACT_NONE;
  a.accum = accumulate ? 1 : 0;
  a.xscale = nullptr;
  a.yscale = yscale;
  a.xact = q; a.gnb_coef = coef; a.gnb_out = gpre_out; a.gnb_act = in_act;
  return launch_conv<32, 128, 1, 1, 1, true>(a, (hipStream_t)stream);
}

// Input gradient of a k4/stride-2/pad-1 convolution: four 2x2 stride-1 phase convolutions of gy on the
// matrix cores, each writing one parity class of gx.  workspace: 4 * (4*cin*cout) floats.
extern "C" int dis_conv2d_dgrad_strided(const float* gy, const float* w_oihw, float* gx, float* workspace, int n,
                                        int hin, int win, int cin, int cout, int k, int stride, int pad,
                                        int accumulate, void* stream) {
  if (!gy || !w_oihw || !gx || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0) return DIS_ERR_BAD_SHAPE;
  if (k != 4 || stride != 2 || pad != 1 || (hin & 1) || (win & 1)) return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int hout = hin / 2, wout = win / 2;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      float* packed = workspace + (long)(py * 2 + px) * 4 * cin * cout;
      long total = 4L * cin * cout;
      hipLaunchKernelGGL(pack_weights_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, w_oihw, packed, cout,
                         cin, cin, k, 2, py, px);
      ConvArgs a;
      a.x = gy; a.w = packed; a.bias = nullptr; a.y = gx; a.stats = nullptr;
      a.n = n; a.hin = hout; a.win = wout; a.hv = hout; a.wv = wout;
      a.pad_y = (py == 0) ? 1 : 0; a.pad_x = (px == 0) ? 1 : 0;
      a.hf = hin; a.wf = win; a.osy = 2; a.ooy = py; a.osx = 2; a.oox = px; a.act = DIS_ACT_NONE;
      a.accum = accumulate ? 1 : 0;
      a.xscale = nullptr;
      a.yscale = nullptr;
      int rc = dispatch_conv(a, cout, cin, 2, 2, 1, s);
      if (rc != DIS_OK) return rc;
    }
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// fp32 convolution on the bf16 matrix cores ("bf16x3"): every fp32 operand is split into three bf16 terms
// x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 significant bits in all), and a
// product a*b is accumulated as the six terms a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16 (each bf16 x bf16 product is exact in fp32; the dropped terms are <= 2^-24 relative, the
// size of one fp32 rounding).  Six bf16 MFMAs of K=32 replace eight fp32 MFMAs of K=4 at 1/16 of their cost each.
// Specialised for the dominant FuseNet shape: 32 -> 32 channels, 3x3, stride 1 (also its input gradient).
//   workgroup = 8 waves = 16x16 output pixels, wave = 2 rows x all 32 couts; weights (3 planes) resident in LDS,
//   the halo tile is split into its 3 bf16 planes when it is written to LDS.
// ------------------------------------------------------------------------------------------------
#define BX_TR 16  // output tile (pixels); LDS pixel stride: BxCfg::PS
#define BX_TC 16

// Compile-time geometry of conv_bf16x3_kernel<CIN, COUT, ...> (CIN, COUT in {16, 32}; 3x3, stride 1).
//   K of one MFMA is 32: with 32 input channels a k-step is one tap, with 16 it is a PAIR of taps (lane groups 0,1 take
//   the first tap's channels 0-7 / 8-15, groups 2,3 the second tap's; the 10th tap has zero weights).
template <int CIN, int COUT, int KHT = 3, int KWT = 3>
struct BxCfg {
  // KHT x KWT = tap window of ONE launch: 3 x 3, or (slice launches of a 7x7 layer, one per tap row) 1 x 7
  static_assert((KHT == 3 && KWT == 3) || (KHT == 1 && KWT == 7 && CIN == 32), "tap windows of conv_bf16x3_kernel");
  static constexpr int IR = BX_TR + KHT - 1, IC = BX_TC + KWT - 1;  // halo tile
  static constexpr int CV = CIN / 4;                 // float4s per pixel
  // LDS pixel stride (16-bit units).  ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table); the pixel operand's address is li * PS + lg * 8, and those
  // groups are conflict-free exactly when the stride is 2 (mod 4) sixteen-byte units: 112 (32 channels: 3 planes + 32 B pad) and
  // 48 (16 channels, no pad).  With the 104 / 56 of round 1 (laid out for contiguous 16-lane groups) every pixel-operand read took
  // two LDS cycles per group: a third of this kernel's LDS cycles were bank-conflict cycles (profiles/r2v4_pmc_sq.csv).
  static constexpr int PS = CIN == 32 ? 112 : 48;
  static constexpr int NT = COUT / 16;               // cout blocks of a wave's tile
  static constexpr int KS = CIN == 32 ? KHT * KWT : 5;  // k-steps
  static constexpr int W_U16 = KS * 3 * 4 * COUT * 8;  // packed[kstep][plane][lg][co][8]
  static constexpr int X_U16 = IR * IC * PS;
  static constexpr int LDS_BYTES = W_U16 * 2 + X_U16 * 2 + 64;
  static constexpr int NITEMS = IR * IC * CV;
  static constexpr int NLOAD = (NITEMS + 511) / 512;  // 6 / 3
  static constexpr int NPIECE = 2 * NT;               // epilogue float4 stores per lane and tile
  // k-step of epilogue piece i and of halo load i (spread so that no k-step carries more than ~40 VALU)
  static constexpr int piece_ks(int i) { return KS == 9 ? 2 * i + 1 : (KS == 7 ? (i < 2 ? 2 * i + 1 : 2 * i) : i + 1); }
  static constexpr int load_ks(int i) { return KS >= 7 ? 2 * (i / 2) : i; }
};

// packed[kstep][plane][lg][co][j] = plane of W(tap, ci, co) with (tap, ci) = (kstep, 8 lg + j) for 32 input channels and
// (2 kstep + lg/2, 8 (lg%2) + j) for 16;  mode 0: W(tap,ci,co) = w[co][ci][tap], mode 1 (input gradient):
// w[ci][co][8-tap].  w is OIHW with dims (wo, wi); channels beyond them (zero-padded inputs) get zero weights.
template <int CIN, int COUT>
__device__ __forceinline__ float bx_weight(const float* w, int stride_row, int mode, int wo, int wi, int ks, int lg, int j,
                                           int co) {
  const int tap = CIN == 32 ? ks : 2 * ks + (lg >> 1);
  const int c = CIN == 32 ? 8 * lg + j : 8 * (lg & 1) + j;
  if (tap > 8) return 0.f;
  if (mode == 0) return (co < wo && c < wi) ? w[co * stride_row + c * 9 + tap] : 0.f;
  return (c < wo && co < wi) ? w[c * stride_row + co * 9 + (8 - tap)] : 0.f;
}

template <int CIN, int COUT>
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int mode,
                                           int wo, int wi) {
  using C = BxCfg<CIN, COUT>;
  const int total = C::KS * 4 * COUT * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, co = (i >> 3) % COUT, lg = (i / (8 * COUT)) & 3, ks = i / (32 * COUT);
    const float v = bx_weight<CIN, COUT>(w, wi * 9, mode, wo, wi, ks, lg, j, co);
    unsigned h1, h2, h3;
    split3(v, h1, h2, h3);
    const int base = ((ks * 3 * 4 + lg) * COUT + co) * 8 + j;
    packed[base] = (unsigned short)h1;
    packed[base + 4 * COUT * 8] = (unsigned short)h2;
    packed[base + 2 * 4 * COUT * 8] = (unsigned short)h3;
  }
}

// Diagnostic build only (make stamp -> libdis_hip_stamp.so, scripts/stamp_bf16x3.py): per-wave s_memtime sums of the
// phases of conv_bf16x3_kernel, written to a buffer nothing else reads.  The product build contains no stamp.
#ifdef BX_STAMP
__device__ unsigned long long bx_stamps[256 * 8 * 8];
#define BX_T(k)                                                   \
  {                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    st_[k] += now_ - last_;                                       \
    last_ = now_;                                                 \
  }
extern "C" int dis_debug_bx_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bx_stamps), sizeof(bx_stamps));
}
#else
#define BX_T(k)
#endif


// Per tile the matrix work is 216 MFMAs per wave (32 -> 32); everything else is software-pipelined around it, and
// everything that rides inside the MFMA loop is straight-line code (no branch: the scheduler can only interleave VALU and
// memory instructions with MFMAs inside one basic block, and an MFMA leaves half of its issue cycles free):
//   * the halo of tile t+1 is fetched into registers by buffer loads issued during the MFMA loop of tile t; pixels
//     outside the image get an out-of-range buffer offset, which loads zeros (the padding) without a branch,
//   * the epilogue of tile t-1 (bias, accumulate, activation, GroupNorm statistics, float4 stores) runs from a register
//     copy of its accumulators during tile t; pixels outside the output get an out-of-range offset, which drops the
//     store.
// Only the LDS refill (two barriers + split + ds_write) stays serial.
// The weights are the MFMA's A operand (M = cout) and the pixels its B operand (N = pixel): a lane then holds 4
// consecutive output channels of one pixel and stores a float4.
// INACT != 0 (input-gradient launches of a conv that had an activation): x is the gradient wrt the activation's OUTPUT
// and a.xact that output; the halo is staged as x * act'(xact), which replaces a separate pass over the tensor.
// INGN: x is a conv output whose GroupNorm has NOT been applied in memory: the halo is staged as x * scale_c + shift_c of its
// sample (the arithmetic of gn_apply_kernel, bit for bit), pixels outside the image stay zero.  Replaces a dis_gn_apply pass
// (read + write of the whole tensor) between a conv -> act -> GroupNorm -> conv pair (reference ResNetBlock, multi_frame_networks.py:514-542).
template <int CIN, int COUT, int ACT, bool ACCUM, bool STATS, int INACT = 0, bool GEN = false, int KHT = 3, int KWT = 3,
          bool INGN = false>
__global__ __launch_bounds__(512) void conv_bf16x3_kernel(ConvArgs a) {
  using C = BxCfg<CIN, COUT, KHT, KWT>;
  constexpr int BX_IC = C::IC;
  static_assert(!GEN || (!STATS && INACT == 0), "slice form: plain convolution / input gradient");
  static_assert(!INGN || (!GEN && INACT == 0 && 512 % C::CV == 0), "GroupNorm on load: dense forward instances");
  static_assert(GEN || (KHT == 3 && KWT == 3), "tap rows are a slice-launch form");
  const int ldx = GEN ? a.ldx : CIN, ldy = GEN ? a.ldy : COUT;  // floats per pixel
  constexpr int PS = C::PS, NT = C::NT, KS = C::KS, NLOAD = C::NLOAD, NPIECE = C::NPIECE, CV = C::CV;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
#ifdef BX_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
  unsigned short* wl = smem16;
  unsigned short* xl = smem16 + C::W_U16;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.wv + BX_TC - 1) / BX_TC, tiles_y = (a.hv + BX_TR - 1) / BX_TR;
  const int ntiles = a.n * tiles_y * tiles_x;
  const int nxcd = (gridDim.x % 8 == 0) ? 8 : 1;
  const int xcd = blockIdx.x % nxcd, rank = blockIdx.x / nxcd, per = gridDim.x / nxcd;
  const int t_lo = (int)((long)ntiles * xcd / nxcd), t_hi = (int)((long)ntiles * (xcd + 1) / nxcd);
  // a step of `per` tiles in (n, ty, tx) coordinates
  const int d_tx = per % tiles_x, d_ty = (per / tiles_x) % tiles_y, d_n = per / (tiles_x * tiles_y);

  // this thread's halo items: (row, col) inside the 18x18 halo and the byte offset from the halo's first pixel.
  // Items past the end of the halo (last round only) get a row that is outside every image.
  float4 pre[NLOAD], pre2[INACT ? NLOAD : 1];
  int it_rc[NLOAD], it_off[NLOAD];
#pragma unroll
  for (int it = 0; it < NLOAD; ++it) {
    const int idx = (int)threadIdx.x + it * 512;
    const int vv = idx % CV, pix = idx / CV;
    const int r = pix / BX_IC, c = pix % BX_IC;
    // (GEN: channels the slice does not have get the out-of-image row too - they load zeros)
    it_rc[it] = (idx < C::NITEMS && (!GEN || vv * 4 < a.cx)) ? (r | (c << 16)) : 0x4000;
    it_off[it] = ((r * a.win + c) * ldx + vv * 4) * 4;
  }
  const unsigned x_bytes = GEN ? ((unsigned)a.hin * a.win * ldx - a.x_sub) * 4u : (unsigned)a.hin * a.win * (CIN * 4u);
  const unsigned y_bytes = GEN ? ((unsigned)a.hf * a.wf * ldy - a.y_sub) * 4u : (unsigned)a.hf * a.wf * (COUT * 4u);
  // halo fetch state of the tile being prefetched (all wave-uniform)
  const float* pf_x = a.x;
  unsigned pf_bytes = 0;
  int pf_iy0 = 0, pf_ix0 = 0, pf_off0 = 0;
  auto pf_setup = [&](int n, int ty, int tx, bool live) {
    pf_iy0 = ty * BX_TR - a.pad_y;
    pf_ix0 = tx * BX_TC - a.pad_x;
    pf_off0 = (pf_iy0 * a.win + pf_ix0) * (ldx * 4);
    pf_x = a.x + (long)n * a.hin * a.win * ldx;
    pf_bytes = live ? x_bytes : 0u;  // no next tile: every load is out of range
  };
  auto pf_issue = [&](int it) {
    const int iy = pf_iy0 + (it_rc[it] & 0xffff), ix = pf_ix0 + (it_rc[it] >> 16);
    const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
    const unsigned off = ok ? (unsigned)(pf_off0 + it_off[it]) : BX_OOB;
    pre[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(pf_x, pf_bytes), off, 0, 0));
    if (INACT)
      pre2[it] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(a.xact + (pf_x - a.x), pf_bytes), off, 0, 0));
  };
  // INGN: this thread's 4 channels are the same for all its items (512 % CV == 0)
  float4 gn_g = make_float4(0.f, 0.f, 0.f, 0.f), gn_b = gn_g, gn_sc = gn_g, gn_sh = gn_g;
  int gn_n = -1;
  if (INGN) {
    gn_g = *(const float4*)(a.gn_gamma + ((int)threadIdx.x % CV) * 4);
    gn_b = *(const float4*)(a.gn_beta + ((int)threadIdx.x % CV) * 4);
  }
  auto stage = [&](int n_cur) {
    if (INGN && n_cur != gn_n) {  // (block-uniform) a new sample: its scale / shift
      gn_n = n_cur;
      float mean, rstd;
      gn_moments(a.gn_stats, n_cur < a.n ? n_cur : a.n - 1, (double)a.hin * a.win * CIN, a.gn_eps, &mean, &rstd);
      gn_sc = make_float4(rstd * gn_g.x, rstd * gn_g.y, rstd * gn_g.z, rstd * gn_g.w);
      gn_sh = make_float4(gn_b.x - gn_sc.x * mean, gn_b.y - gn_sc.y * mean, gn_b.z - gn_sc.z * mean, gn_b.w - gn_sc.w * mean);
    }
    // (wave-uniform) the whole halo lies inside the image: no padding masks
    const bool gn_interior = INGN && pf_iy0 >= 0 && pf_ix0 >= 0 && pf_iy0 + C::IR <= a.hin && pf_ix0 + C::IC <= a.win;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the halo loads (one unconditional wait, not one per divergent item)
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      if ((int)threadIdx.x + it * 512 < C::NITEMS) {
        float4 v = pre[it];
        if (INGN) {
          // (pf_iy0 / pf_ix0 still describe the tile being staged: pf_setup of the next one runs after this.)  Pixels
          // outside the image were loaded as zeros and must stay zero: their shift is dropped (0 * scale + 0).  Packed fp32
          // multiply / add (two roundings, as gn_apply_kernel computes it).
          f32x2 sh_lo = {gn_sh.x, gn_sh.y}, sh_hi = {gn_sh.z, gn_sh.w};
          if (!gn_interior) {
            const int iy = pf_iy0 + (it_rc[it] & 0xffff), ix = pf_ix0 + (it_rc[it] >> 16);
            const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
            sh_lo = ok ? sh_lo : (f32x2){0.f, 0.f};
            sh_hi = ok ? sh_hi : (f32x2){0.f, 0.f};
          }
          const f32x2 lo = (f32x2){v.x, v.y} * (f32x2){gn_sc.x, gn_sc.y} + sh_lo;
          const f32x2 hi = (f32x2){v.z, v.w} * (f32x2){gn_sc.z, gn_sc.w} + sh_hi;
          v = make_float4(lo[0], lo[1], hi[0], hi[1]);
        }
        if (INACT) {
          const float4 q = pre2[it];
          v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
          v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
        }
        unsigned a1, a2, a3, b1, b2, b3;
        split3_pair(v.x, v.y, a1, a2, a3);
        split3_pair(v.z, v.w, b1, b2, b3);
        const int idx = (int)threadIdx.x + it * 512;
        unsigned short* p = xl + (idx / CV) * PS + (idx % CV) * 4;
        *(uint2*)(p) = make_uint2(a1, b1);
        *(uint2*)(p + CIN) = make_uint2(a2, b2);
        *(uint2*)(p + 2 * CIN) = make_uint2(a3, b3);
      }
    }
  };

  int tile = t_lo + rank;
  int cn = 0, cty = 0, ctx = 0;  // coordinates of `tile`
  if (tile < t_hi) {
    ctx = tile % tiles_x, cty = (tile / tiles_x) % tiles_y, cn = tile / (tiles_x * tiles_y);
    pf_setup(cn, cty, ctx, true);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) pf_issue(it);  // in flight while the weights are copied
  }
  if (KWT == 7) {
    // tap row of a 7x7 weight (w_o x w_i x 49 floats, rows w_rs apart): tap kx of this launch is element
    // wtap0 + kx * wtap_step of a (row, column) pair's 49 (forward: row ky left to right; input gradient: the mirrored
    // row right to left).  7 k floats per workgroup, L2-resident: gathered directly, no LDS staging.
    for (int u = threadIdx.x; u < KS * 4 * COUT; u += 512) {
      const int co = u % COUT, g = (u / COUT) & 3, ks = u / (4 * COUT);
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 8 * g + 2 * j + e;
          const int t = a.wtap0 + ks * a.wtap_step;
          if (a.wmode == 0) v[e] = (co < a.w_o && c < a.w_i) ? a.w[(long)co * a.w_rs + c * 49 + t] : 0.f;
          else v[e] = (c < a.w_o && co < a.w_i) ? a.w[(long)c * a.w_rs + co * 49 + t] : 0.f;
        }
        split3_pair(v[0], v[1], pl[0][j], pl[1][j], pl[2][j]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *(uint4*)(wl + (((ks * 3 + p) * 4 + g) * COUT + co) * 8) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
  } else if (a.wmode < 0) {
    // (packed planes: <= 7 uint4 per thread, all requested before the first is stored - see dis_copy_w_rows)
    constexpr int NWV = (C::W_U16 / 8 + 511) / 512;
    uint4 wv[NWV];
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
      const int i = (int)threadIdx.x + k * 512;
      wv[k] = i < C::W_U16 / 8 ? ((const uint4*)a.w)[i] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
      const int i = (int)threadIdx.x + k * 512;
      if (i < C::W_U16 / 8) ((uint4*)wl)[i] = wv[k];
    }
  } else {
    // OIHW fp32 weights: split here instead of in a launch of their own.  Coalesced copy into the (still unused) halo
    // region, rows padded by one float so that the gather below is conflict-free, then every thread builds
    // (k-step, lane group, cout) units of 3 x 8 bf16.
    float* ws = (float*)xl;
    const int row = a.w_i * 9;
    dis_copy_w_rows(a.w, a.w_o, row, a.w_rs, ws);  // (all loads in flight at once)
    __syncthreads();
    for (int u = threadIdx.x; u < KS * 4 * COUT; u += 512) {
      const int co = u % COUT, g = (u / COUT) & 3, ks = u / (4 * COUT);
      unsigned pl[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = bx_weight<CIN, COUT>(ws, row + 1, a.wmode, a.w_o, a.w_i, ks, g, 2 * j, co);
        const float v1 = bx_weight<CIN, COUT>(ws, row + 1, a.wmode, a.w_o, a.w_i, ks, g, 2 * j + 1, co);
        split3_pair(v0, v1, pl[0][j], pl[1][j], pl[2][j]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *(uint4*)(wl + (((ks * 3 + p) * 4 + g) * COUT + co) * 8) = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
    // (the first __syncthreads of the tile loop orders these reads of `ws` before the halo is staged over it)
  }

  f32x4 acc[2][NT], outv[2][NT];
  float4 prevy[NPIECE];
  double s1 = 0.0, s2 = 0.0;
  float t1 = 0.f, t2 = 0.f;
  int stat_n = -1;
  float4 bias_v[NT];
  bool yok[NT];  // GEN: this lane's 4 output channels of block nt exist in the slice
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    yok[nt] = !GEN || nt * 16 + lg * 4 < a.cy;
    bias_v[nt] = (a.bias && (!GEN || nt * 16 + lg * 4 < a.nbias)) ? *(const float4*)(a.bias + nt * 16 + lg * 4)
                                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int yrow = a.osy * a.wf * (ldy * 4);  // bytes between output rows of this launch
  const int y_lane = ((wave * 2 * a.osy * a.wf + li * a.osx) * ldy + lg * 4) * 4;

  // deferred epilogue state (tile t-1): wave-uniform sample base, per-lane byte offsets of its two rows (or BX_OOB)
  const float* prev_y = a.y;
  unsigned prev_off[2] = {BX_OOB, BX_OOB};
  int prev_n = -1;
  double* red = (double*)(smem16 + C::W_U16 + C::X_U16);
  auto stats_flush = [&]() {  // (block-uniform: every wave of the workgroup walks the same tile sequence)
    // one atomic pair per workgroup: with one per wave the ~8k same-address fp64 atomics at the end of a launch cost 20 us
    const double r1 = block_sum_d(s1, red);
    const double r2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
      atomic_add_d(a.stats + 2 * stat_n, r1);
      atomic_add_d(a.stats + 2 * stat_n + 1, r2);
    }
    s1 = 0.0;
    s2 = 0.0;
  };
  auto stats_sample = [&]() {  // the statistics are per sample: flush when the deferred tile starts a new one
    if (STATS && prev_n >= 0 && prev_n != stat_n) {
      if (stat_n >= 0) stats_flush();
      stat_n = prev_n;
    }
  };
  // accumulate mode: the old values of a tile's outputs are fetched at the START of its own MFMA loop (a whole tile of
  // latency cover) and added when the finished accumulators are handed to the deferred epilogue
  auto epi_load = [&](const float* yb, const unsigned (&off)[2]) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i)
      prevy[i] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc(yb, y_bytes),
                                                        yok[i % NT] ? off[i / NT] + (i % NT) * 64 : BX_OOB, 0, 0));
  };
  auto epi_piece = [&](int i) {
    const int mt = i / NT, nt = i % NT;
    float o[4];  // (the bias is already in: the accumulators start from it)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = outv[mt][nt][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ACT == DIS_ACT_SELU) {
        // act_apply without a select (the compiler turns one into EXEC-masked arms): for x > 0 the second term is
        // sa * (exp(0) - 1) = 0 exactly, for x <= 0 the first is 0, so the sum equals the selected arm bit for bit
        const float e = __builtin_amdgcn_exp2f(fminf(o[r], 0.f) * 1.44269504088896340736f);
        o[r] = SELU_SCALE_F * fmaxf(o[r], 0.f) + (SELU_SCALE_F * SELU_ALPHA_F) * (e - 1.f);
      } else if (ACT == DIS_ACT_RELU) {
        o[r] = fmaxf(o[r], 0.f);
      }
    }
    const bool live = prev_off[mt] != BX_OOB;
    u32x4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = __float_as_uint(o[r]);
    __builtin_amdgcn_raw_buffer_store_b128(ov, bx_rsrc(prev_y, y_bytes), yok[nt] ? prev_off[mt] + nt * 64 : BX_OOB, 0, 0);
    if (STATS) {
      const float q1 = (o[0] + o[1]) + (o[2] + o[3]);
      const float q2 = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
      t1 += live ? q1 : 0.f;
      t2 += live ? q2 : 0.f;
    }
  };

  // per-lane LDS offsets (16-bit units) of the pixel operand: row of the wave, column li, channel group of the lane
  const int xa_lane = (wave * 2 * BX_IC + li) * PS + (CIN == 32 ? lg * 8 : (lg & 1) * 8);
  const bool hi_tap = (lg >> 1) != 0;  // CIN == 16: lane groups 2, 3 take the second tap of the pair

  while (tile < t_hi) {
    // where this tile's outputs go
    const int vy0 = cty * BX_TR + wave * 2, vx0 = ctx * BX_TC + li;
    const int tile_yoff = ((cty * BX_TR * a.osy + a.ooy) * a.wf + ctx * BX_TC * a.osx + a.oox) * (ldy * 4) + y_lane;
    const float* cur_y = a.y + (long)cn * a.hf * a.wf * ldy;
    unsigned cur_off[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) cur_off[mt] = (vx0 < a.wv && vy0 + mt < a.hv) ? (unsigned)(tile_yoff + mt * yrow) : BX_OOB;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){bias_v[nt].x, bias_v[nt].y, bias_v[nt].z, bias_v[nt].w};
    stats_sample();
    BX_T(0)
    __syncthreads();  // every wave has finished reading the previous halo tile
    BX_T(1)
    stage(cn);
    BX_T(2)
    __syncthreads();
    BX_T(3)
    const int ntile = tile + per;
    int ntx = ctx + d_tx, nty = cty + d_ty, nn = cn + d_n;
    if (ntx >= tiles_x) ntx -= tiles_x, ++nty;
    if (nty >= tiles_y) nty -= tiles_y, ++nn;
    pf_setup(nn, nty, ntx, ntile < t_hi);
    t1 = 0.f;
    t2 = 0.f;
    BX_T(4)

    // memory work riding under the MFMAs of k-step ks (straight-line): accumulate-mode reads (this tile's), the next
    // halo, the deferred stores
    auto ride = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if (ACCUM && ks == 0) epi_load(cur_y, cur_off);
#pragma unroll
      for (int it = 0; it < NLOAD; ++it)
        if (C::load_ks(it) == ks) pf_issue(it);
#pragma unroll
      for (int i = 0; i < NPIECE; ++i)
        if (C::piece_ks(i) == ks) epi_piece(i);
    };
    // issue order inside a k-step: every MFMA is followed by what fits into the issue cycles it leaves free - one of
    // the next k-step's NR fragment reads behind each of the first MFMAs, a few VALU instructions behind the rest
    auto pattern = [&](auto nrc) {
      constexpr int NM = 12 * NT, NR = decltype(nrc)::value;
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        if (g < NR) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // VALU
        } else {
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        if (g == NR) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // the k-step's halo loads
      }
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // the k-step's store
      __builtin_amdgcn_sched_barrier(0);
    };
    // smallest terms first
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    if constexpr (CIN == 32) {
      // k-step = tap, walked column by column (kx outer, ky inner): the wave's two output rows r, r+1 read halo rows
      // r+ky and r+1+ky, so one column of taps needs only the 4 row fragments r..r+3 (read once, 36 pixel-fragment
      // reads per tile instead of 54).  LDS bandwidth is what the MFMA rate competes with in this kernel: 12 b128 reads
      // per 24 MFMAs keep the LDS port exactly as busy as the matrix unit.
      s16x8 R[2][4][3];      // [kx parity][halo row][plane]
      s16x8 fb[2][3][NT];    // [buffer][plane][nt]
      auto load_row = [&](int kx, int j) {
#pragma unroll
        for (int p = 0; p < 3; ++p) R[kx & 1][j][p] = *(const s16x8*)(xl + xa_lane + (j * BX_IC + kx) * PS + p * CIN);
      };
      auto load_w = [&](int ks, s16x8 (&B)[3][NT]) {
        const int wt = (ks % KHT) * KWT + ks / KHT;  // the weights are packed tap-major (ky * KWT + kx)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) B[p][nt] = *(const s16x8*)(wl + (((wt * 3 + p) * 4 + lg) * COUT + nt * 16 + li) * 8);
      };
      load_row(0, 0);
      load_row(0, 1);
      load_w(0, fb[0]);
      auto step = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int kx = ks / KHT, ky = ks % KHT, nkx = (ks + 1) / KHT, nky = (ks + 1) % KHT;
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
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[b][PB[q]][nt]),
                                                                   __builtin_bit_cast(bf16x8, R[kx & 1][ky + mt][PA[q]]),
                                                                   acc[mt][nt], 0, 0, 0);
        pattern(std::integral_constant<int, (ks + 1 < KS ? (nky == 0 ? 6 : 3) + 3 * NT : 0)>{});
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      if constexpr (KS == 9) {
        step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{});
      }
    } else {
      s16x8 fa[2][3][2], fb[2][3][NT];  // [buffer][plane][mt|nt]
      auto load_frag = [&](int ks, s16x8 (&A)[3][2], s16x8 (&B)[3][NT]) {
        // k-step = a PAIR of taps (the 10th tap reads tap 8's pixels: zero weights)
        const int t0 = 2 * ks, t1_ = 2 * ks + 1 > 8 ? 8 : 2 * ks + 1;
        const int xoff = hi_tap ? ((t1_ / 3) * BX_IC + t1_ % 3) * PS : ((t0 / 3) * BX_IC + t0 % 3) * PS;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) A[p][mt] = *(const s16x8*)(xl + xa_lane + xoff + mt * BX_IC * PS + p * CIN);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) B[p][nt] = *(const s16x8*)(wl + (((ks * 3 + p) * 4 + lg) * COUT + nt * 16 + li) * 8);
        }
      };
      load_frag(0, fa[0], fb[0]);
      auto step = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if (ks + 1 < KS) load_frag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
        ride(ksc);
        constexpr int b = ks & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[b][PB[q]][nt]),
                                                                   __builtin_bit_cast(bf16x8, fa[b][PA[q]][mt]),
                                                                   acc[mt][nt], 0, 0, 0);
        pattern(std::integral_constant<int, 3 * (2 + NT)>{});
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
    }
    s1 += (double)t1;
    s2 += (double)t2;
    BX_T(5)

    // hand the finished tile over to the deferred epilogue
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      prev_off[mt] = cur_off[mt];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        outv[mt][nt] = acc[mt][nt];
        if (ACCUM) {
          const float4 q = prevy[mt * NT + nt];
          outv[mt][nt] += (f32x4){q.x, q.y, q.z, q.w};
        }
      }
    }
    prev_y = cur_y;
    prev_n = cn;
    cn = nn, cty = nty, ctx = ntx;
    tile = ntile;
    BX_T(6)
  }
  // the last tile's epilogue
  stats_sample();
  t1 = 0.f;
  t2 = 0.f;
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) epi_piece(i);
  s1 += (double)t1;
  s2 += (double)t2;
  if (STATS && stat_n >= 0) stats_flush();
#ifdef BX_STAMP
  BX_T(7)
  if (lane == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) bx_stamps[(blockIdx.x * 8 + wave) * 8 + k] = st_[k];
#endif
}

// (cin, cout) -> kernel instance
template <int CIN, int COUT>
static hipError_t bx_launch(const ConvArgs& a, bool stats, int inact, long grid, hipStream_t stream) {
  using C = BxCfg<CIN, COUT>;
  const int variant = (a.act * 2 + a.accum) * 2 + (stats ? 1 : 0);
  const bool ingn = a.gn_stats != nullptr;
  static bool attr_set[12 + 4 + 4] = {};
  auto launch = [&](auto kern) -> hipError_t {
    bool& set = attr_set[ingn ? 16 + (a.act == DIS_ACT_SELU ? 2 : 0) + (stats ? 1 : 0)
                              : (inact ? 12 + (inact - 1) * 2 + a.accum : variant)];
    if (!set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
      if (e != hipSuccess) return e;
      set = true;
    }
    DIS_TAG(CIN == 32 && COUT == 32 ? "conv_bf16x3_kernel<32,32>" : CIN == 16 && COUT == 16 ? "conv_bf16x3_kernel<16,16>" : CIN == 16 ? "conv_bf16x3_kernel<16,32>" : "conv_bf16x3_kernel<32,16>");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C::LDS_BYTES, stream, a);
    return hipSuccess;
  };
  if (ingn) {  // GroupNorm applied on load: the consumer convs of the conv -> act -> GroupNorm -> conv chains (forward only)
    if constexpr (CIN == COUT) {
      if (inact || a.accum || (a.act != DIS_ACT_NONE && a.act != DIS_ACT_SELU)) return hipErrorInvalidValue;
      if (a.act == DIS_ACT_SELU)
        return stats ? launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_SELU, false, true, 0, false, 3, 3, true>)
                     : launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_SELU, false, false, 0, false, 3, 3, true>);
      return stats ? launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, false, true, 0, false, 3, 3, true>)
                   : launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, false, false, 0, false, 3, 3, true>);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (inact) {  // input gradient with the activation gradient fused into the staging: no epilogue activation, no statistics
    if (a.act != DIS_ACT_NONE || stats) return hipErrorInvalidValue;
    if (inact == DIS_ACT_SELU)
      return a.accum ? launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, true, false, DIS_ACT_SELU>)
                     : launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, false, false, DIS_ACT_SELU>);
    return a.accum ? launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, true, false, DIS_ACT_RELU>)
                   : launch(conv_bf16x3_kernel<CIN, COUT, DIS_ACT_NONE, false, false, DIS_ACT_RELU>);
  }
  switch (variant) {
#define BX_CASE(ACT_, ACC_, ST_) \
  case ((ACT_)*2 + (ACC_)) * 2 + (ST_): return launch(conv_bf16x3_kernel<CIN, COUT, ACT_, (ACC_) != 0, (ST_) != 0>);
    BX_CASE(DIS_ACT_NONE, 0, 0) BX_CASE(DIS_ACT_NONE, 0, 1) BX_CASE(DIS_ACT_NONE, 1, 0) BX_CASE(DIS_ACT_NONE, 1, 1)
    BX_CASE(DIS_ACT_SELU, 0, 0) BX_CASE(DIS_ACT_SELU, 0, 1) BX_CASE(DIS_ACT_SELU, 1, 0) BX_CASE(DIS_ACT_SELU, 1, 1)
    BX_CASE(DIS_ACT_RELU, 0, 0) BX_CASE(DIS_ACT_RELU, 0, 1) BX_CASE(DIS_ACT_RELU, 1, 0) BX_CASE(DIS_ACT_RELU, 1, 1)
#undef BX_CASE
  }
  return hipErrorInvalidValue;
}

static bool bx_shape_ok(int cin, int cout, int k, int stride) {
  return (cin == 16 || cin == 32) && (cout == 16 || cout == 32) && k == 3 && stride == 1;
}

extern "C" long dis_conv2d_pack_bf16x3_size(int cin, int cout) {
  if (!bx_shape_ok(cin, cout, 3, 1)) return DIS_ERR_UNSUPPORTED;
  return (cin == 32 ? 9 : 5) * 3 * 4 * cout * 8;  // 16-bit words
}

/* w_oihw has dims (w_o, w_i, 3, 3); mode 0: conv w_i -> w_o channels (cin >= w_i zero-padded inputs, cout == w_o);
 * mode 1: its input gradient (cin == w_o, cout >= w_i). */
extern "C" int dis_conv2d_pack_weights_bf16x3(const float* w_oihw, void* packed, int cout, int cin, int k, int mode,
                                              void* stream) {
  if (!w_oihw || !packed) return DIS_ERR_NULL;
  if (!bx_shape_ok(cin, cout, k, 1) || mode < 0 || mode > 1) return DIS_ERR_UNSUPPORTED;
  const int wo = mode == 0 ? cout : cin, wi = mode == 0 ? cin : cout;
  unsigned short* pk = (unsigned short*)packed;
  hipStream_t s = (hipStream_t)stream;
  if (cin == 32 && cout == 32) hipLaunchKernelGGL((pack_weights_bf16x3_kernel<32, 32>), dim3(36), dim3(256), 0, s, w_oihw, pk, mode, wo, wi);
  else if (cin == 16 && cout == 16) hipLaunchKernelGGL((pack_weights_bf16x3_kernel<16, 16>), dim3(36), dim3(256), 0, s, w_oihw, pk, mode, wo, wi);
  else if (cin == 16 && cout == 32) hipLaunchKernelGGL((pack_weights_bf16x3_kernel<16, 32>), dim3(36), dim3(256), 0, s, w_oihw, pk, mode, wo, wi);
  else hipLaunchKernelGGL((pack_weights_bf16x3_kernel<32, 16>), dim3(36), dim3(256), 0, s, w_oihw, pk, mode, wo, wi);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

struct GnIn {  // GroupNorm applied to x on load (ConvArgs::gn_stats ...); stats == nullptr: none
  const double* stats = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float eps = 0.f;
  const float* ab_x = nullptr;  // ConvArgs::ab_x / ab_out / ab_act_y (f16x2 kernel only)
  double* ab_out = nullptr;
  const float* ab_act_y = nullptr;
  const float* gnb_coef = nullptr;  // ConvArgs::gnb_coef / gnb_out (f16x2 kernel only; the second operand arrives as xact)
  float* gnb_out = nullptr;
};
static int launch_conv_bf16x3(const float* x, const void* w, int wmode, int w_o, int w_i, int w_rs, const float* bias, float* y,
                              double* stats, int n, int hin, int win, int cin, int cout, int k, int stride, int pad,
                              int act, void* stream, const float* xact = nullptr, int inact = 0, GnIn gn = GnIn()) {
  if (!x || !w || !y) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || pad < 0) return DIS_ERR_BAD_SHAPE;
  if (!bx_shape_ok(cin, cout, k, stride)) return DIS_ERR_UNSUPPORTED;
  const int hout = hin + 2 * pad - 2, wout = win + 2 * pad - 2;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  ConvArgs a;
  a.x = x; a.w = (const float*)w; a.bias = bias; a.y = y; a.stats = stats;
  a.n = n; a.hin = hin; a.win = win; a.hv = hout; a.wv = wout; a.pad_y = pad; a.pad_x = pad;
  a.hf = hout; a.wf = wout; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
  a.act = act & 0xff;
  a.accum = (act & DIS_CONV_ACCUM) ? 1 : 0;
  a.xscale = nullptr;
  a.yscale = nullptr;
  a.wmode = wmode;
  a.w_o = w_o;
  a.w_i = w_i;
  a.w_rs = w_rs;
  a.xact = xact;
  a.gn_stats = gn.stats; a.gn_gamma = gn.gamma; a.gn_beta = gn.beta; a.gn_eps = gn.eps;
  a.ab_x = gn.ab_x; a.ab_out = gn.ab_out; a.ab_slots = num_cus(); a.ab_act_y = gn.ab_act_y;
  a.gnb_coef = gn.gnb_coef; a.gnb_out = gn.gnb_out;
  if ((gn.ab_out || gn.gnb_coef) && !(dis_f2_enabled() && wmode >= 0)) return DIS_ERR_UNSUPPORTED;  // (the two-term kernel's forms)
  if (gn.gnb_coef && (!gn.gnb_out || !xact || pad != 1 || cin != cout)) return DIS_ERR_UNSUPPORTED;
  if (gn.stats && (!gn.gamma || !gn.beta)) return DIS_ERR_NULL;
  const bool inres = gn.stats && gn.gnb_out && !gn.gnb_coef;   // SELU(GroupNorm(x) + xact) formed on load, stored to gnb_out
  if (gn.stats && ((cin != cout && !(inres && cin == 32 && cout == 16)) || inact || (act & DIS_CONV_ACCUM) ||
                   ((act & 0xff) != DIS_ACT_NONE && (act & 0xff) != DIS_ACT_SELU)))
    return DIS_ERR_UNSUPPORTED;
  if (inres && (!xact || pad != 1 || !(dis_f2_enabled() && wmode >= 0))) return DIS_ERR_UNSUPPORTED;
  if (inact < 0 || inact > DIS_ACT_RELU || (inact && !xact)) return DIS_ERR_UNSUPPORTED;
  if (a.act > DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  // the kernel addresses x and y per sample through buffer descriptors with 31-bit byte offsets
  if ((long)hin * win * cin * 4 >= 0x7fff0000L || (long)hout * wout * cout * 4 >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;
  const int tiles_x = (wout + BX_TC - 1) / BX_TC, tiles_y = (hout + BX_TR - 1) / BX_TR;
  const long ntiles = (long)n * tiles_y * tiles_x;
  long grid = num_cus();
#ifdef BX_STAMP
  if (getenv("BX_GRID")) grid = atol(getenv("BX_GRID"));
#endif
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  hipError_t le;
  if (dis_f2_enabled() && wmode >= 0) {  // two-term fp16 split (conv_f16x2.hip); no instance for this configuration: fall through
    le = dis_f2_conv_launch(a, cin, cout, stats != nullptr, inact, grid, (hipStream_t)stream);
    if (le == hipSuccess) {
      DIS_CHECK_LAUNCH();
      return DIS_OK;
    }
    if (le != hipErrorInvalidValue) return (int)le;
  }
  if (gn.ab_out || gn.gnb_coef || inres) return DIS_ERR_UNSUPPORTED;
  if (cin == 32 && cout == 32) le = bx_launch<32, 32>(a, stats != nullptr, inact, grid, (hipStream_t)stream);
  else if (cin == 16 && cout == 16) le = bx_launch<16, 16>(a, stats != nullptr, inact, grid, (hipStream_t)stream);
  else if (cin == 16 && cout == 32) le = bx_launch<16, 32>(a, stats != nullptr, inact, grid, (hipStream_t)stream);
  else le = bx_launch<32, 16>(a, stats != nullptr, inact, grid, (hipStream_t)stream);
  if (le != hipSuccess) return (int)le;
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

extern "C" int dis_conv2d_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* y, double* stats,
                                     int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act,
                                     void* stream) {
  return launch_conv_bf16x3(x, w_packed, -1, 0, 0, 0, bias, y, stats, n, hin, win, cin, cout, k, stride, pad, act, stream);
}
/* w_oihw (w_o, w_i, 3, 3) as stored by the module: mode 0 needs cout == w_o and cin >= w_i (zero-padded input channels),
 * mode 1 (input gradient of that conv) cin == w_o and cout >= w_i.  w_row_stride: floats between consecutive w_o rows
 * (0 or w_i*9: dense; larger when w_oihw points into a slice w[:, a:b] of a wider weight). */
extern "C" int dis_conv2d_fwd_bf16x3_oihw(const float* x, const float* w_oihw, int mode, int w_o, int w_i, int w_row_stride,
                                          const float* bias, float* y, double* stats, int n, int hin, int win, int cin,
                                          int cout, int k, int stride, int pad, int act, void* stream) {
  if (mode < 0 || mode > 1 || w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32) return DIS_ERR_UNSUPPORTED;
  if (mode == 0 ? (cout != w_o || cin < w_i) : (cin != w_o || cout < w_i)) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  return launch_conv_bf16x3(x, w_oihw, mode, w_o, w_i, w_row_stride, bias, y, stats, n, hin, win, cin, cout, k, stride,
                            pad, act, stream);
}
/* dis_conv2d_fwd_bf16x3_oihw(mode 0) of GroupNorm(x): x is the PRE-normalisation tensor, gn_stats (n, 2) its fp64 per-sample
 * sum / sum of squares, gn_gamma / gn_beta (cin) the GroupNorm(1 group) parameters.  y = act(conv(gn(x)) + bias), equal bit for
 * bit to dis_gn_apply followed by the plain call; the normalised tensor never exists in memory.  cin == cout, no accumulate. */
extern "C" int dis_conv2d_fwd_bf16x3_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta,
                                        float gn_eps, const float* w_oihw, int w_o, int w_i, int w_row_stride,
                                        const float* bias, float* y, double* stats, int n, int hin, int win, int cin,
                                        int cout, int k, int stride, int pad, int act, void* stream) {
  if (!gn_stats || !gn_gamma || !gn_beta) return DIS_ERR_NULL;
  if (w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32) return DIS_ERR_UNSUPPORTED;
  if (cout != w_o || cin != w_i) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.stats = gn_stats; gn.gamma = gn_gamma; gn.beta = gn_beta; gn.eps = gn_eps;
  return launch_conv_bf16x3(x, w_oihw, 0, w_o, w_i, w_row_stride, bias, y, stats, n, hin, win, cin, cout, k, stride, pad, act,
                            stream, nullptr, 0, gn);
}
/* y = act(conv3x3(out) + bias) with out = SELU(GroupNorm(x2) + res) - the output of a ResNetBlock (reference
 * model/multi_frame_networks.py:514-542) - FORMED ON LOAD from x2 (the block's second conv output), its statistics gn_stats (n, 2) and
 * the block's residual `res`, with the arithmetic of dis_gn_apply(act = SELU, residual) bit for bit; the conv's tiles also store the
 * values of the pixels they own to `out` (shaped like x2): the pass dis_gn_apply would have made over x2, res and out does not exist,
 * and every later reader of `out` (the next block's residual, the backward passes) finds it in memory.  3x3, stride 1, pad 1,
 * cin == cout in {16, 32} (act SELU, with or without output statistics) or 32 -> 16 (act SELU, no statistics); two-term fp16 kernels
 * only (DIS_ERR_UNSUPPORTED otherwise: the caller runs dis_gn_apply and the plain forward). */
extern "C" int dis_conv2d_fwd_f16x2_gnres(const float* x2, const double* gn_stats, const float* gn_gamma, const float* gn_beta,
                                          float gn_eps, const float* res, float* out, const float* w_oihw, int w_o, int w_i,
                                          int w_row_stride, const float* bias, float* y, double* stats, int n, int hin, int win,
                                          int cin, int cout, int act, void* stream) {
  if (!gn_stats || !gn_gamma || !gn_beta || !res || !out) return DIS_ERR_NULL;
  if (w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32 || cout != w_o || cin != w_i) return DIS_ERR_BAD_SHAPE;
  if (act != DIS_ACT_SELU) return DIS_ERR_UNSUPPORTED;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.stats = gn_stats; gn.gamma = gn_gamma; gn.beta = gn_beta; gn.eps = gn_eps;
  gn.gnb_out = out;
  return launch_conv_bf16x3(x2, w_oihw, 0, w_o, w_i, w_row_stride, bias, y, stats, n, hin, win, cin, cout, 3, 1, 1, act, stream, res,
                            0, gn);
}
/* Input gradient of the conv of dis_conv2d_fwd_bf16x3_gn, g = conv_T(gy, w), which ALSO leaves what the GroupNorm backward needs
 * besides g: per (sample, channel) the sums of g and of g * x over the pixels (x = gn_x, the GroupNorm's input, shaped like g),
 * one fp64 slot per workgroup: ab_out (n, dis_conv2d_gnsums_slots(), 2, cin) doubles, ZEROED by the caller.  Replaces the
 * reduce pass of dis_gn_apply_bwd (a read of g and x).  Two-term fp16 kernels only (DIS_ERR_UNSUPPORTED otherwise: the caller keeps
 * the two-pass form). */
extern "C" long dis_conv2d_gnsums_slots(void) { return num_cus(); }
extern "C" int dis_conv2d_dgrad_bf16x3_gnsums(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g,
                                              const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout,
                                              int pad, void* stream) {
  if (!gn_x || !ab_out) return DIS_ERR_NULL;
  if (w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32 || cin != w_o || cout != w_i || cin != cout) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.ab_x = gn_x; gn.ab_out = ab_out;
  return launch_conv_bf16x3(gy, w_oihw, 1, w_o, w_i, w_row_stride, nullptr, g, nullptr, n, hin, win, cin, cout, 3, 1, pad, 0, stream,
                            nullptr, 0, gn);
}
/* The ResNetBlock-chain form: g = (g + conv_T(gy, w)) * selu'(act_y) in place (g arrives holding the residual branch's gradient;
 * act_y = the block input = output of the previous block's SELU(GroupNorm(.) + residual)), plus the sums of g and g * gn_x (gn_x =
 * that GroupNorm's input) as above: g is then BOTH the previous block's residual gradient and the input of dis_gn_bwd_from_sums. */
extern "C" int dis_conv2d_dgrad_bf16x3_gnsums_res(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride,
                                                  float* g, const float* act_y, const float* gn_x, double* ab_out, int n, int hin,
                                                  int win, int cin, int cout, int pad, void* stream) {
  if (!gn_x || !ab_out) return DIS_ERR_NULL;  // (act_y == NULL: no activation between the GroupNorm and this conv)
  if (w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32 || cin != w_o || cout != w_i || cin != cout) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.ab_x = gn_x; gn.ab_out = ab_out; gn.ab_act_y = act_y;
  return launch_conv_bf16x3(gy, w_oihw, 1, w_o, w_i, w_row_stride, nullptr, g, nullptr, n, hin, win, cin, cout, 3, 1, pad,
                            DIS_CONV_ACCUM, stream, nullptr, 0, gn);
}
/* ... and for a conv that had an activation of its own behind such a block (final_conv: 32 -> 16 + SELU behind ref_res3):
 * g = conv_T(gy * selu'(y), w) * selu'(act_y), written (not accumulated), plus the sums.  cin = 16 (gy), cout = 32 (g). */
extern "C" int dis_conv2d_dgrad_bf16x3_act_gnsums_res(const float* gy, const float* y, const float* w_oihw, int w_o, int w_i,
                                                      int w_row_stride, float* g, const float* act_y, const float* gn_x,
                                                      double* ab_out, int n, int hin, int win, int cin, int cout, int pad,
                                                      void* stream) {
  if (!y || !gn_x || !ab_out || !act_y) return DIS_ERR_NULL;
  if (!((w_o == 16 && w_i == 32) || (w_o == 32 && w_i == 16)) || cin != w_o || cout != w_i) return DIS_ERR_UNSUPPORTED;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.ab_x = gn_x; gn.ab_out = ab_out; gn.ab_act_y = act_y;
  return launch_conv_bf16x3(gy, w_oihw, 1, w_o, w_i, w_row_stride, nullptr, g, nullptr, n, hin, win, cin, cout, 3, 1, pad, 0, stream,
                            y, DIS_ACT_SELU, gn);
}
/* Input gradient of a 3x3 stride-1 convolution that was followed by an activation, with the activation's gradient fused
 * in: gx (+)= conv_T(gy * act'(y), w) where y (same shape as gy) is the activation's output.  Replaces dis_act_bwd +
 * dis_conv2d_fwd_bf16x3_oihw(mode 1).  w_oihw (w_o, w_i, 3, 3) is the forward convolution's weight: gy has w_o channels,
 * gx cout >= w_i.  accumulate != 0 adds into gx. */
extern "C" int dis_conv2d_dgrad_bf16x3_act(const float* gy, const float* y, int act, const float* w_oihw, int w_o, int w_i,
                                           int w_row_stride, float* gx, int n, int hin, int win, int cin, int cout,
                                           int pad, int accumulate, void* stream) {
  if (!y) return DIS_ERR_NULL;
  if (act != DIS_ACT_SELU && act != DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  if (w_o <= 0 || w_i <= 0 || w_o > 32 || w_i > 32 || cin != w_o || cout < w_i) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  return launch_conv_bf16x3(gy, w_oihw, 1, w_o, w_i, w_row_stride, nullptr, gx, nullptr, n, hin, win, cin, cout, 3, 1, pad,
                            accumulate ? DIS_CONV_ACCUM : 0, stream, y, act);
}


/* Input gradient of a 3x3 stride-1 pad-1 convolution c -> c (c in {16, 32}) that is followed by [activation in_act ->] GroupNorm(1
 * group), WITH that GroupNorm's backward elementwise pass applied while the operand is staged (round 5; reference
 * model/multi_frame_networks.py:338-345, :514-542):
 *     gpre = act'(q) * (g * k1_c + q * kx + k0)        g: gradient wrt the GroupNorm's output, q: the GroupNorm's input (this conv's
 *     gx (+)= conv_T(gpre, w)                             activated output), coef (n, c + 2): dis_gn_bwd_coef's coefficients
 * and gpre is stored to gpre_out (shaped like g; every pixel once, by the tile that owns it) for the layer's weight-gradient launch.
 * Replaces dis_gn_bwd_apply_coef (a read of g and q, a write of gpre) + the read of gpre by the plain input-gradient launch.
 * accumulate != 0: gx += ...  Optional epilogue of the dis_conv2d_dgrad_bf16x3_gnsums / _gnsums_res forms (all three NULL: none):
 * ab_out != NULL leaves the channel sums of the finished gx and gx * ab_gn_x; ab_act_y != NULL multiplies the finished value by
 * selu'(ab_act_y) first (accumulate required).  Two-term fp16 kernels only (DIS_ERR_UNSUPPORTED otherwise: the caller keeps the
 * separate pass). */
extern "C" int dis_conv2d_dgrad_f16x2_gnb(const float* g, const float* q, const float* coef, int in_act, float* gpre_out,
                                          const float* w_oihw, int w_o, int w_i, int w_row_stride, float* gx, int accumulate,
                                          const float* ab_gn_x, const float* ab_act_y, double* ab_out, int n, int hin, int win,
                                          int c, void* stream) {
  if (!g || !q || !coef || !gpre_out) return DIS_ERR_NULL;
  if (in_act != DIS_ACT_NONE && in_act != DIS_ACT_SELU) return DIS_ERR_UNSUPPORTED;
  if (w_o != c || w_i != c || (c != 16 && c != 32)) return DIS_ERR_UNSUPPORTED;
  if ((ab_out != nullptr) != (ab_gn_x != nullptr) || (ab_act_y && (!ab_out || !accumulate))) return DIS_ERR_BAD_SHAPE;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  GnIn gn;
  gn.ab_x = ab_gn_x; gn.ab_out = ab_out; gn.ab_act_y = ab_act_y;
  gn.gnb_coef = coef; gn.gnb_out = gpre_out;
  return launch_conv_bf16x3(g, w_oihw, 1, w_o, w_i, w_row_stride, nullptr, gx, nullptr, n, hin, win, c, c, 3, 1, 1,
                            accumulate ? DIS_CONV_ACCUM : 0, stream, q, in_act, gn);
}


// ---- 3x3 stride-1 pad-1 layers of the general family (DispNetS) as 32 x 32 channel-slice launches of the kernel above
// (called from dis_convg_run; conv_gen.hip).  Input-channel slices after the first accumulate into y; the bias enters
// with the first slice and the activation with the last.  dgrad != 0: x is the gradient wrt the conv's output and w its
// weight [cin_w][cout_w][3][3] (DIS_CONVG_CONV_DGRAD), else w is [cout_w][cin_w][3][3].
long dis_bx_slices_ok(int n, int h, int wd, int cin, int cout, int ldx, int ldy, int xoff, int yoff, int k, int stride,
                      int pad, int act) {
  static const bool use3 = !(getenv("DIS_CONV_BF16X3") && getenv("DIS_CONV_BF16X3")[0] == '0');
  if (!use3 || !((k == 3 && pad == 1) || (k == 7 && pad == 3)) || stride != 1 || cin < 16 || cout < 16) return 0;
  if ((cin & 3) || (cout & 3) || (ldx & 3) || (ldy & 3) || (xoff & 3) || (yoff & 3)) return 0;
  if (act != DIS_ACT_NONE && act != DIS_ACT_RELU) return 0;
  if ((long)h * wd * ldx * 4 >= 0x7fff0000L || (long)h * wd * ldy * 4 >= 0x7fff0000L) return 0;
  // one launch per slice pair: worth it for the few-channel layers at high resolution only (the deep layers have
  // hundreds of pairs over a handful of tiles: the streaming kernel stays their path)
  const long pairs = (long)((cin + 31) / 32) * ((cout + 31) / 32);
  if (pairs * (k == 7 ? 7 : 1) > 10 || (long)n * h * wd < 400000L) return 0;  // (7x7: one launch per tap row)
  return 1;
}
int dis_bx_slices_run(int dgrad, const float* x, int ldx, int xoff, int cin, int cin_w, const float* w,
                      const float* bias, float* y, int ldy, int yoff, int cout, int cout_w, int n, int h, int wd, int k,
                      int act, hipStream_t stream) {
  using C = BxCfg<32, 32>;
  using C7 = BxCfg<32, 32, 1, 7>;
  static_assert(C7::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  const int kk = k * k, nrow = k == 7 ? 7 : 1;  // launches per slice pair
  const int ncb = (cin + 31) / 32, ngb = (cout + 31) / 32;
  // input slices that have real weights (the rest are zero-padded lanes and contribute nothing)
  const int ncb_w = (cin_w + 31) / 32;
  const int tiles_x = (wd + BX_TC - 1) / BX_TC, tiles_y = (h + BX_TR - 1) / BX_TR;
  const long ntiles = (long)n * tiles_y * tiles_x;
  long grid = num_cus();
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  static bool attr_set[8] = {};
  for (int gb = 0; gb < ngb; ++gb)
    for (int cb = 0; cb < (ncb_w < ncb ? ncb_w : ncb); ++cb)
     for (int tr = 0; tr < nrow; ++tr) {
      const bool first = cb == 0 && tr == 0, last = cb == (ncb_w < ncb ? ncb_w : ncb) - 1 && tr == nrow - 1;
      ConvArgs a;
      a.x = x + xoff + 32 * cb; a.bias = (first && bias) ? bias + 32 * gb : nullptr; a.y = y + yoff + 32 * gb; a.stats = nullptr;
      a.n = n; a.hin = h; a.win = wd; a.hv = h; a.wv = wd;
      a.pad_y = k == 7 ? 3 - tr : 1;  // tap row tr of a 7x7 window reads input rows vy + tr - 3
      a.pad_x = k == 7 ? 3 : 1;
      a.hf = h; a.wf = wd; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
      a.act = last ? act : DIS_ACT_NONE;
      a.accum = first ? 0 : 1;
      if (!first) a.bias = nullptr;
      a.xscale = nullptr; a.yscale = nullptr; a.xact = nullptr;
      a.gn_stats = nullptr; a.gn_gamma = nullptr; a.gn_beta = nullptr; a.gn_eps = 0.f;
      a.ab_x = nullptr; a.ab_out = nullptr; a.ab_slots = 0; a.ab_act_y = nullptr;
      a.ldx = ldx; a.ldy = ldy;
      a.cx = cin - 32 * cb < 32 ? cin - 32 * cb : 32;
      a.cy = cout - 32 * gb < 32 ? cout - 32 * gb : 32;
      a.x_sub = xoff + 32 * cb; a.y_sub = yoff + 32 * gb;
      const int wi_in = cin_w - 32 * cb < 32 ? cin_w - 32 * cb : 32;      // real input channels of the slice
      int wo_out = cout_w - 32 * gb < 32 ? cout_w - 32 * gb : 32;        // real output channels of the slice
      if (wo_out < 0) wo_out = 0;
      a.nbias = wo_out;
      if (!dgrad) {
        a.wmode = 0; a.w = w + ((long)32 * gb * cin_w + 32 * cb) * kk; a.w_o = wo_out; a.w_i = wi_in; a.w_rs = cin_w * kk;
        a.wtap0 = tr * 7; a.wtap_step = 1;
      } else {
        a.wmode = 1; a.w = w + ((long)32 * cb * cout_w + 32 * gb) * kk; a.w_o = wi_in; a.w_i = wo_out; a.w_rs = cout_w * kk;
        a.wtap0 = (6 - tr) * 7 + 6; a.wtap_step = -1;  // the mirrored window
      }
      if (wo_out == 0) { a.w = w; a.w_o = 0; a.w_i = 0; }  // zero-padded output lanes only: written as zeros (+0 bias)
      const int variant = (a.act == DIS_ACT_RELU ? 2 : 0) + a.accum;
      const int lds = k == 7 ? C7::LDS_BYTES : C::LDS_BYTES, slot = variant + (k == 7 ? 4 : 0);
      auto launch = [&](auto kern) -> hipError_t {
        if (!attr_set[slot]) {
          hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
          if (e != hipSuccess) return e;
          attr_set[slot] = true;
        }
        DIS_TAG("conv_bf16x3_kernel<32,32,GEN> slices");
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, a);
        return hipSuccess;
      };
      hipError_t le = hipErrorInvalidValue;
      // two-term fp16 split (conv_f16x2.hip, default): the 3 x 3 slices; the 7 x 7 tap rows stay on the three-term kernel
      if (k == 3 && dis_f2_enabled()) le = dis_f2_conv_gen_launch(a, grid, stream);
      if (le == hipSuccess) continue;
      if (le != hipErrorInvalidValue) return (int)le;
      if (k == 7) {
        if (variant == 0) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_NONE, false, false, 0, true, 1, 7>);
        else if (variant == 1) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_NONE, true, false, 0, true, 1, 7>);
        else if (variant == 2) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_RELU, false, false, 0, true, 1, 7>);
        else le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_RELU, true, false, 0, true, 1, 7>);
      } else if (variant == 0) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_NONE, false, false, 0, true>);
      else if (variant == 1) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_NONE, true, false, 0, true>);
      else if (variant == 2) le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_RELU, false, false, 0, true>);
      else le = launch(conv_bf16x3_kernel<32, 32, DIS_ACT_RELU, true, false, 0, true>);
      if (le != hipSuccess) return (int)le;
    }
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[(tap,ci)][co] = sum_pixels X[pixel+tap][ci] * G[pixel][co]
// GEMM view: M = (tap,ci) rows, N = cout, K = pixels.  Each wave owns a disjoint pixel subset of the
// tile and keeps the whole (M x N) accumulator of one cin-chunk (and one ky row when split) in registers
// across ALL its tiles; partial results of every wave are written once and summed by a second kernel
// (deterministic, no float atomics).
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int KH, int KW, int S>
struct WgCfg {
  // channels per workgroup.  The 128 -> 32 1x1 conv of Block2D3D (conv_mf) takes all 128 at once: with 32-channel chunks as
  // separate workgroups gy was read once per chunk (828 MB per launch against 566 MB algorithmic, round-2 profile)
  static constexpr bool WIDE1X1 = CIN == 128 && KH == 1 && KW == 1 && S == 1;
  static constexpr int CINB = WIDE1X1 ? 128 : ((CIN % 32 == 0) ? 32 : ((CIN % 16 == 0) ? 16 : 4));
  static constexpr int NCHUNK = CIN / CINB;
  static constexpr int SG = CIN >= 32 ? CIN / 32 : 1;  // 32-channel groups of x: the granularity of WgArgs::xscale
  static constexpr bool KYSPLIT = (KH * KW * CINB * COUT) > 160 * 64;  // accumulator registers > 160
  static constexpr int KHB = KYSPLIT ? 1 : KH;     // tap rows handled per block
  static constexpr int NSPLIT = KYSPLIT ? KH : 1;
  static constexpr int MROWS = KHB * KW * CINB;
  static constexpr int MB = (MROWS + 15) / 16;
  static constexpr int NB = COUT / 16;
  static constexpr int TROWS = (S == 1 && !WIDE1X1) ? 8 : 4;  // (WIDE1X1: 64 pixels x 144 floats = 37 KB, 3 workgroups per CU)
  static constexpr int RPW = TROWS / 4;            // tile rows per wave
  static constexpr int STEPS = RPW * 4;            // MFMA k-steps per wave per tile (16 px per row / 4)
  static constexpr int IN_ROWS = (TROWS - 1) * S + KHB;
  static constexpr int IN_COLS = 15 * S + KW;
  // pixel strides == 16 (mod 32) floats: the ds_read_b32 fragment reads of lanes l and l+16 (neighbouring pixels)
  // then hit disjoint banks
  static constexpr int CS = (CINB >= 16) ? CINB + 16 : CINB;
  static constexpr int GS = (COUT % 32 == 0) ? COUT + 16 : COUT;
  static constexpr int IN_FLOATS = IN_ROWS * IN_COLS * CS;
  static constexpr int G_FLOATS = TROWS * 16 * GS;
  static constexpr int LDS_BYTES = (IN_FLOATS + G_FLOATS) * 4;
  static constexpr int NV = CINB / 4;
  static constexpr int PART = MB * 16 * COUT;      // floats per partial slab
};


template <int CIN, int COUT, int KH, int KW, int S, bool GACT = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  using C = WgCfg<CIN, COUT, KH, KW, S>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xl = smem;
  float* gl = smem + C::IN_FLOATS;
  const int chunk = blockIdx.y, split = blockIdx.z;
  const int ky0 = C::KYSPLIT ? split : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + C::TROWS - 1) / C::TROWS;
  const int ntiles = a.n * tiles_y * tiles_x;

  // per-lane A-row constants: m = mb*16 + li  ->  (tap, ci)
  int aoff[C::MB];
  bool aval[C::MB];
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    const int m = mb * 16 + li;
    aval[mb] = m < C::MROWS;
    const int mm = aval[mb] ? m : 0;
    const int tap = mm / C::CINB, ci = mm % C::CINB;
    const int ky = tap / KW, kx = tap % KW;
    aoff[mb] = (ky * C::IN_COLS + kx) * C::CS + ci;
  }
  f32x4 acc[C::MB][C::NB];
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bsum[C::NB];
#pragma unroll
  for (int nb = 0; nb < C::NB; ++nb) bsum[nb] = 0.f;

  // issue-early / write-late staging: the x halo tile and the gy tile of the NEXT tile are fetched into registers
  // while the MFMAs of the current one run
  constexpr int NIX = C::IN_ROWS * C::IN_COLS * C::NV, NLX = (NIX + 255) / 256;
  constexpr int NIG = C::TROWS * 16 * (COUT / 4), NLG = (NIG + 255) / 256;
  float4 prex[NLX], preg[NLG], preq[GACT ? NLG : 1];   // (GACT: the activated output at gy's positions)
  float presc[NLX];
#pragma unroll
  for (int it = 0; it < NLX; ++it) presc[it] = 1.f;
  // (loads stay under their bounds branches here: the unconditional/clamped form that helps conv_fwd_kernel made
  //  hipcc park the prefetched values in AGPRs right behind each load in this register-bound kernel: 114 -> 77 TFLOP/s)
  auto prefetch = [&](int tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * C::TROWS * S - a.pad + ky0, ix0 = tx * 16 * S - a.pad;
    const float* xb = a.x + (long)n * a.hin * a.win * CIN + chunk * C::CINB;
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      const int idx = threadIdx.x + it * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NIX) {
        const int vv = idx % C::NV, pix = idx / C::NV;
        const int c = pix % C::IN_COLS, r = pix / C::IN_COLS;
        const int iy = iy0 + r, ix = ix0 + c;
        if (iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) {
          v = *(const float4*)(xb + ((long)iy * a.win + ix) * CIN + vv * 4);
          // (the scale is applied in stage(): multiplying here made every item wait for its two loads in turn - the prefetch
          //  was not one: 224 us per launch of the 128 -> 32 1x1 weight gradient at 2.5 TB/s)
          if (a.xscale)
            presc[it] = a.xscale[(((long)n * a.hin + iy) * a.win + ix) * C::SG + chunk * (C::CINB / 32) + (vv * 4) / 32];
        }
      }
      prex[it] = v;
    }
    const float* gb = a.gy + (long)n * a.hout * a.wout * COUT;
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int idx = threadIdx.x + it * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (GACT) preq[GACT ? it : 0] = v;   // (pixels past the map: gy = 0 must not meet an unset act'(.) - 0 x NaN)
      if (idx < NIG) {
        const int vv = idx % (COUT / 4), pix = idx / (COUT / 4);
        const int c = pix % 16, r = pix / 16;
        const int oy = ty * C::TROWS + r, ox = tx * 16 + c;
        if (oy < a.hout && ox < a.wout) {
          v = *(const float4*)(gb + ((long)oy * a.wout + ox) * COUT + vv * 4);
          if (GACT) preq[GACT ? it : 0] = *(const float4*)(a.gact + (gb - a.gy) + ((long)oy * a.wout + ox) * COUT + vv * 4);
        }
      }
      preg[it] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      const int idx = threadIdx.x + it * 256;
      if (idx < NIX) {
        float4 v = prex[it];
        if (a.xscale) v.x *= presc[it], v.y *= presc[it], v.z *= presc[it], v.w *= presc[it];
        *(float4*)(xl + (idx / C::NV) * C::CS + (idx % C::NV) * 4) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int idx = threadIdx.x + it * 256;
      if (idx < NIG) {
        float4 v = preg[it];
        if (GACT) {   // (dis_act_bwd's arithmetic on load: gpre = gy * act'(y); pixels past the map hold gy = 0)
          const float4 q = preq[GACT ? it : 0];
          v.x *= act_grad_from_out(q.x, a.gact_act), v.y *= act_grad_from_out(q.y, a.gact_act);
          v.z *= act_grad_from_out(q.z, a.gact_act), v.w *= act_grad_from_out(q.w, a.gact_act);
        }
        *(float4*)(gl + (idx / (COUT / 4)) * C::GS + (idx % (COUT / 4)) * 4) = v;
      }
    }
  };

  if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    stage();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
#pragma unroll
    for (int st = 0; st < C::STEPS; ++st) {
      const int pk = wave * (C::RPW * 16) + st * 4 + lg;  // pixel inside the tile owned by this k-slot
      const int pr = pk >> 4, pc = pk & 15;
      const float* xp = xl + ((pr * S) * C::IN_COLS + pc * S) * C::CS;
      const float* gp = gl + pk * C::GS + li;
      float bv[C::NB];
#pragma unroll
      for (int nb = 0; nb < C::NB; ++nb) {
        bv[nb] = gp[nb * 16];
        bsum[nb] += bv[nb];
      }
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        float av = xp[aoff[mb]];
        if (!aval[mb]) av = 0.f;
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[mb][nb], 0, 0, 0);
      }
    }
  }
  // cross-wave reduction through LDS (the tile buffers are free now), then ONE slab per workgroup: [m][co]
  float* red = smem;  // needs 4 * NB * 256 floats
  float* out = a.part + (((long)blockIdx.x * C::NCHUNK + chunk) * C::NSPLIT + split) * C::PART;
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * C::NB + nb) * 4 + r) * 64 + lane] = acc[mb][nb][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < C::NB; ++k) {
      const int v = threadIdx.x + k * 256;
      const int nb = v >> 8, r = (v & 255) >> 6, ln = v & 63;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += red[((w * C::NB + nb) * 4 + r) * 64 + ln];
      out[(mb * 16 + (ln >> 4) * 4 + r) * COUT + nb * 16 + (ln & 15)] = sum;
    }
  }
  if (a.bpart && chunk == 0 && split == 0) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) {
      float v = bsum[nb];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0) red[(wave * C::NB + nb) * 16 + li] = v;
    }
    __syncthreads();
    if (threadIdx.x < COUT) {
      const int nb = threadIdx.x >> 4, l2 = threadIdx.x & 15;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += red[(w * C::NB + nb) * 16 + l2];
      a.bpart[(long)blockIdx.x * COUT + threadIdx.x] = sum;
    }
  }
}

// Slab reduce: sums the per-workgroup partial slabs and scatters into OIHW in one launch.  A block = 64 output elements
// x 16 waves; wave w adds slabs [w nslabs/16, (w+1) nslabs/16) of its lane's element (coalesced 256-B rows, 8 loads in
// flight: the kernel is bound by the latency of its strided reads, hence the many short ranges), the sixteen partial sums
// are added in a fixed order (deterministic).  WG_RSPLIT only sizes the (now unused) level-1 scratch that sits between
// the slabs and the bias partials in the workspace layout.
#define WG_RSPLIT 16
#define WG_RW 16  // waves per block of the reduce
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
                                                             int cinb, int nchunk, int nsplit, int khb, int kw, int kh,
                                                             int cout, int cin_real, int partsz,
                                                             const float* __restrict__ bpart, float* __restrict__ gb,
                                                             int workers) {
  __shared__ double red[64 * WG_RW];  // (fp64 sums over the slabs: a weight gradient is a sum of ~1e5 signed per-pixel terms
                                      // at full resolution, and the slab sum is where fp32 would lose the most)
  const int mrows = khb * kw * cinb;
  const long total = (long)nchunk * nsplit * mrows * cout;
  const long elems = (long)nchunk * nsplit * partsz;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long i = base + ln;
    double s = 0.0;
    int co = 0, m = 0, split = 0, chunk = 0;
    if (i < total) {
      co = (int)(i % cout);
      long r = i / cout;
      m = (int)(r % mrows);
      r /= mrows;
      split = (int)(r % nsplit);
      chunk = (int)(r / nsplit);
      const float* p = part + ((long)chunk * nsplit + split) * partsz + (long)m * cout + co;
      const int lo = (int)((long)workers * wv / WG_RW), hi = (int)((long)workers * (wv + 1) / WG_RW);
      int k = lo;
      // (the wave's whole range - 32 slabs at 512 workers - in flight at once: with 8 per round the launch was four dependent
      //  memory round trips long, 8.7 us, and there are 56 of these launches in a DIS-MF step)
      for (; k + 31 < hi; k += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = p[(long)(k + u) * elems];
#pragma unroll
        for (int u = 0; u < 32; u += 8)
          s += (((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3])) +
               (((double)v[u + 4] + (double)v[u + 5]) + ((double)v[u + 6] + (double)v[u + 7]));
      }
      for (; k + 7 < hi; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long)(k + u) * elems];
        s += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) +
             (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
      }
      for (; k < hi; ++k) s += (double)p[(long)k * elems];
    }
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    if (wv == 0 && i < total) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < WG_RW; ++w) t += red[w * 64 + ln];
      const int tap = m / cinb, ci = chunk * cinb + m % cinb;
      const int ky = (nsplit > 1 ? split : 0) + tap / kw, kx = tap % kw;
      if (ci < cin_real) gw[(((long)co * cin_real + ci) * kh + ky) * kw + kx] = (float)t;
    }
  }
  // bias gradient: block b < cout/8 adds the per-workgroup bias partials of channels 8b..8b+7; thread t sums workers
  // {t/8, t/8 + 128, ...} of channel 8b + t%8, then a fixed-order sum over the 128 sub-sums
  if (bpart && (int)blockIdx.x * 8 < cout) {
    const int co = blockIdx.x * 8 + (threadIdx.x & 7), sub = threadIdx.x >> 3;
    double s = 0.0;
    for (int k = sub; k < workers; k += 128) s += (double)bpart[(long)k * cout + co];
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    // (two levels, fixed order: 128 threads add 8 sub-sums each, then 8 threads add 16 - a single 128-term loop in 8 threads was
    //  ~6 us of dependent LDS reads at the end of every launch with a bias)
    double t8 = 0.0;
    if (threadIdx.x < 128) {
      const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) t8 += red[(g * 8 + j) * 8 + c];
    }
    __syncthreads();
    if (threadIdx.x < 128) red[threadIdx.x] = t8;
    __syncthreads();
    if (threadIdx.x < 8) {
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * 8 + threadIdx.x];
      gb[co] = (float)t;
    }
  }
}
static inline int wgrad_reduce_grid(long total, bool bias) {
  long g = (total + 63) / 64;
  if (g > 1024) g = 1024;
  return (bias && g < 16) ? 16 : (int)g;  // the bias part needs cout/8 <= 16 blocks
}
#define WG_WORKERS 768

template <int CIN, int COUT, int KH, int KW, int S>
static long wgrad_ws(void) {
  using C = WgCfg<CIN, COUT, KH, KW, S>;
  const long elems = (long)C::NCHUNK * C::NSPLIT * C::PART;
  return (long)WG_WORKERS * elems + (long)WG_RSPLIT * elems + (long)WG_WORKERS * COUT;
}

template <int CIN, int COUT, int KH, int KW, int S, bool GACT = false>
static int launch_wgrad(WgArgs a, float* gw, float* gb, int cin_real, hipStream_t s) {
  using C = WgCfg<CIN, COUT, KH, KW, S>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  static_assert(C::IN_FLOATS + C::G_FLOATS >= 4 * C::NB * 256, "reduction scratch does not fit the tile buffers");
  static bool attr_set = false;
  auto kern = conv_wgrad_kernel<CIN, COUT, KH, KW, S, GACT>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + C::TROWS - 1) / C::TROWS;
  const long ntiles = (long)a.n * tiles_y * tiles_x;
  long workers = 2L * num_cus();
  if (workers > WG_WORKERS) workers = WG_WORKERS;
  if (workers > ntiles) workers = ntiles;
  const long elems = (long)C::NCHUNK * C::NSPLIT * C::PART;
  float* tmp = a.part + (long)WG_WORKERS * elems;
  a.bpart = gb ? tmp + (long)WG_RSPLIT * elems : nullptr;
  DIS_TAG("conv_wgrad_kernel (fp32 MFMA)");
  hipLaunchKernelGGL(kern, dim3((unsigned)workers, C::NCHUNK, C::NSPLIT), dim3(256), C::LDS_BYTES, s, a);
  const long total = (long)C::NCHUNK * C::NSPLIT * C::MROWS * COUT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_grid(total, gb != nullptr)), dim3(64 * WG_RW), 0, s,
                     (const float*)a.part, gw, C::CINB, C::NCHUNK, C::NSPLIT, C::KHB, KW, KH, COUT, cin_real, C::PART,
                     (const float*)(gb ? a.bpart : nullptr), gb, (int)workers);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ------------------------------------------------------------------------------------------------
// bf16x3 weight gradient of the 3x3 stride-1 convolutions with 16 / 32 channels on either side (same operand split as
// conv_bf16x3_kernel).
// GEMM view dW[(tap,ci)][co] = sum_pixels X[pixel+tap][ci] * G[pixel][co]: the contraction runs over PIXELS, so both
// MFMA operands need, per lane, 8 consecutive pixels of one channel.  The LDS images stay [pixel][plane][channel]
// (written exactly like the forward kernel's halo tile) and the operands are fetched with the hardware transposing
// read ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, delivered channel-major).
// Workgroup = 4 waves, tile = 8x16 output pixels (4 k-steps of 32 pixels); the (9 CIN/16) x (COUT/16) accumulator tiles
// are split over the waves (9 each for 32 -> 32) and stay in registers across all tiles; no cross-wave reduction.
// Partial slabs / bias partials use the layout of conv_wgrad_kernel, so its reducers finish the job.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
// K x K taps, stride S, TR x 16 output pixels per tile (TR / 2 k-steps of 32 pixels).  The FuseNet instances are
// K = 3, S = 1, TR = 8; the slice-pair instances of DispNetS also use 5 x 5 taps and stride 2.
// NP = bf16 planes per value (3: fp32 inputs split on the way into LDS; 1: the inputs ARE bf16, conv_bf16.hip), CPI = channels
// per 16-byte staging item (4 floats / 8 bf16)
template <int CIN, int COUT, int K = 3, int S = 1, int TR = 8, int KH = K, int NP = 3, int CPI = 4>
struct WxCfg {
  // LDS pixel strides (16-bit units).  The transposing reads (ds_read_b64_tr_b16: two groups of 32 lanes, 64 banks) of one group
  // touch 8 pixels (4 columns x 2 lane groups), 32 bytes each, `step` pixels apart (1 for gy and stride-1 x, 2 for stride-2 x):
  // they are conflict-free when the 8 offsets step * p * stride (dwords, mod 64) are distinct multiples of 8, i.e. when
  // step * stride = 8 * odd (mod 64).  Smallest such stride >= the payload, in 16-byte units (measured before: half of the LDS
  // cycles of this kernel were bank-conflict cycles with the 8-element pad).
  static constexpr int ps_for(int payload_u16, int step) {
    int ps = (payload_u16 + 7) / 8 * 8;
    while (((step * (ps / 2)) % 16) != 8) ps += 8;
    return ps;
  }
  static constexpr int PSX = ps_for(NP * CIN, S), PSG = ps_for(NP * COUT, 1);
  static constexpr int CVX = CIN / CPI, CVG = COUT / CPI;
  static constexpr int IR = (TR - 1) * S + KH, IC = 15 * S + K;  // halo of the x tile (KH of the K tap rows per workgroup)
  static constexpr int X_U16 = IR * IC * PSX, G_U16 = TR * 16 * PSG;
  static constexpr int LDS_BYTES = (X_U16 + G_U16) * 2 + 1024 * 4;
  static constexpr int NIX = IR * IC * CVX, NLX = (NIX + 255) / 256;
  static constexpr int NLG = TR * 16 * CVG / 256;
  static constexpr int KSN = TR / 2;                                                          // k-steps per tile
  static constexpr int MB = KH * K * CIN / 16, NB = COUT / 16, T = MB * NB, TW = (T + 3) / 4;  // tiles, tiles per wave
  static_assert(TR % 2 == 0 && (TR * 16 * CVG) % 256 == 0, "tile geometry");
};

template <int I, int N, class F>
__device__ __forceinline__ void wx_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wx_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ s16x8 tr_read8(const unsigned short* p0, const unsigned short* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  return (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// INGN: x is staged as GroupNorm(x) (the consumer conv of a conv -> act -> GroupNorm -> conv chain: its input was never
// written in normalised form, see conv_bf16x3_kernel); padding pixels stay zero.
template <int CIN, int COUT, int INACT = 0, bool GEN = false, int K = 3, int S = 1, int TR = 8, int KH = K, bool BF = false,
          bool INGN = false>
__global__ __launch_bounds__(256) void conv_wgrad_bf16x3_kernel(WgArgs a) {
  // BF: x and gy hold bf16 values (a.x / a.gy point at 16-bit elements; ldx / xoff / ldg / goff count elements): one plane,
  // one product per MAC, no split - the bf16 activation storage mode of DispNetS (conv_bf16.hip)
  constexpr int NP = BF ? 1 : 3, CPI = BF ? 8 : 4, ES = BF ? 2 : 4;
  static_assert(!BF || (GEN && INACT == 0), "bf16 inputs: slice-pair form only");
  static_assert(!INGN || (!GEN && !BF), "GroupNorm on load: the FuseNet form");
  using C = WxCfg<CIN, COUT, K, S, TR, KH, NP, CPI>;
  static_assert(!INGN || 256 % C::CVX == 0, "a thread keeps its 4 channels over its items");
  const int ky0 = KH < K ? (int)blockIdx.z * KH : 0;  // first tap row of this workgroup (7x7: two groups of 4 rows)
  constexpr int WX_IC = C::IC;
  static_assert(GEN || (K == 3 && S == 1 && TR == 8), "the FuseNet form");
  static_assert(!GEN || (CIN == 32 && (COUT == 32 || COUT == 16) && INACT == 0), "slice-pair form: 32 x 32 (x 16) blocks");
  // pixel strides (floats) and first channel of this workgroup's slices
  const int ldx = GEN ? a.ldx : CIN, ldg = GEN ? a.ldg : COUT;
  const int cb = GEN ? (int)blockIdx.y % a.npx : 0, gbk = GEN ? (int)blockIdx.y / a.npx : 0;
  const int xc0 = GEN ? a.xoff + 32 * cb : 0, gc0 = GEN ? a.goff + COUT * gbk : 0;
  constexpr int PSX = C::PSX, PSG = C::PSG, NLX = C::NLX, NLG = C::NLG, NB = C::NB, TW = C::TW;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* xl = smem16;
  unsigned short* gl = smem16 + C::X_U16;
  float* bred = (float*)(smem16 + C::X_U16 + C::G_U16);  // 1024 floats: bias partial reduction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane >> 4, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;  // tr-read role: row tq, column chunk tp
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + TR - 1) / TR;
  const int ntiles = a.n * tiles_y * tiles_x;

  // my accumulator tiles: t = TW*wave + j -> (mb = t / NB, nb = t % NB); mb = (tap, 16-channel half of the input).  The
  // per-wave tile set is a compile-time constant inside run<W>() below (static register indexing of the accumulators).
  f32x4 acc[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);  // channels 4*(tid % CVG).. of the gy pixels this thread stages

  // halo / gradient items of this thread; pixels outside the image get an out-of-range buffer offset, which loads the
  // zero padding without a branch, a clamp or a mask (as in conv_bf16x3_kernel)
  float4 prex[NLX], preg[NLG], preg2[INACT ? NLG : 1];  // (INACT: gy is staged as gy * act'(gact), as in conv_bf16x3_kernel)
  int ix_rc[NLX], ix_off[NLX], ig_rc[NLG], ig_off[NLG];
#pragma unroll
  for (int it = 0; it < NLX; ++it) {
    const int idx = (int)threadIdx.x + it * 256;
    const int vv = idx % C::CVX, pix = idx / C::CVX;
    const int r = pix / WX_IC, c = pix % WX_IC;
    // (channels past the layer's last one - ragged last slice - get the out-of-image row too: they load zeros)
    ix_rc[it] = (idx < C::NIX && (!GEN || 32 * cb + vv * CPI < a.cx)) ? (r | (c << 16)) : 0x4000;
    ix_off[it] = ((r * a.win + c) * ldx + xc0 + vv * CPI) * ES;
  }
#pragma unroll
  for (int it = 0; it < NLG; ++it) {
    const int idx = threadIdx.x + it * 256;
    const int vv = idx % C::CVG, pix = idx / C::CVG;
    ig_rc[it] = (!GEN || COUT * gbk + vv * CPI < a.cg) ? ((pix >> 4) | ((pix & 15) << 16)) : 0x4000;
    ig_off[it] = (((pix >> 4) * a.wout + (pix & 15)) * ldg + gc0 + vv * CPI) * ES;
  }
  const unsigned x_bytes = (unsigned)a.hin * a.win * (ldx * (unsigned)ES), g_bytes = (unsigned)a.hout * a.wout * (ldg * (unsigned)ES);
  // INGN: origin and sample of the tile whose items are in flight (stage() runs before the next prefetch), this thread's
  // GroupNorm parameters and the scale / shift of the current sample
  int st_iy0 = 0, st_ix0 = 0, st_n = -1, gn_n = -1;
  float4 gn_g = make_float4(0.f, 0.f, 0.f, 0.f), gn_b = gn_g, gn_sc = gn_g, gn_sh = gn_g;
  if (INGN) {
    gn_g = *(const float4*)(a.gn_gamma + ((int)threadIdx.x % C::CVX) * 4);
    gn_b = *(const float4*)(a.gn_beta + ((int)threadIdx.x % C::CVX) * 4);
  }
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int iy0 = ty * (TR * S) - a.pad + ky0, ix0 = tx * (16 * S) - a.pad;
    if (INGN) st_iy0 = iy0, st_ix0 = ix0, st_n = n;
    const char* xb = (const char*)a.x + (long)n * a.hin * a.win * ldx * ES;
    const int xoff0 = (iy0 * a.win + ix0) * (ldx * ES);
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      const int iy = iy0 + (ix_rc[it] & 0xffff), ix = ix0 + (ix_rc[it] >> 16);
      const bool ok = (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      prex[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                bx_rsrc(xb, x_bytes), ok ? (unsigned)(xoff0 + ix_off[it]) : BX_OOB, 0, 0));
    }
    const char* gb = (const char*)a.gy + (long)n * a.hout * a.wout * ldg * ES;
    const int goff0 = (ty * TR * a.wout + tx * 16) * (ldg * ES);
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int oy = ty * TR + (ig_rc[it] & 0xffff), ox = tx * 16 + (ig_rc[it] >> 16);
      const bool ok = oy < a.hout && ox < a.wout;
      preg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                bx_rsrc(gb, g_bytes), ok ? (unsigned)(goff0 + ig_off[it]) : BX_OOB, 0, 0));
      if (INACT)
        preg2[it] = __builtin_bit_cast(
            float4, __builtin_amdgcn_raw_buffer_load_b128(bx_rsrc((const char*)a.gact + (gb - (const char*)a.gy), g_bytes),
                                                          ok ? (unsigned)(goff0 + ig_off[it]) : BX_OOB, 0, 0));
    }
  };
  auto put3 = [&](unsigned short* p, const float4& v, int plane) __attribute__((always_inline)) {
    if (BF) {  // 8 bf16 values, as loaded
      *(float4*)p = v;
      return;
    }
    unsigned a1, a2, a3, b1, b2, b3;
    split3_pair(v.x, v.y, a1, a2, a3);
    split3_pair(v.z, v.w, b1, b2, b3);
    *(uint2*)(p) = make_uint2(a1, b1);
    *(uint2*)(p + plane) = make_uint2(a2, b2);
    *(uint2*)(p + 2 * plane) = make_uint2(a3, b3);
  };
  auto stage = [&]() __attribute__((always_inline)) {
    if (INGN && st_n != gn_n) {  // (block-uniform) a new sample
      gn_n = st_n;
      float mean, rstd;
      gn_moments(a.gn_stats, st_n < a.n ? st_n : a.n - 1, (double)a.hin * a.win * CIN, a.gn_eps, &mean, &rstd);
      gn_sc = make_float4(rstd * gn_g.x, rstd * gn_g.y, rstd * gn_g.z, rstd * gn_g.w);
      gn_sh = make_float4(gn_b.x - gn_sc.x * mean, gn_b.y - gn_sc.y * mean, gn_b.z - gn_sc.z * mean, gn_b.w - gn_sc.w * mean);
    }
    const bool gn_interior = INGN && st_iy0 >= 0 && st_ix0 >= 0 && st_iy0 + C::IR <= a.hin && st_ix0 + C::IC <= a.win;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one unconditional wait for the prefetched tile
#pragma unroll
    for (int it = 0; it < NLX; ++it) {
      const int idx = (int)threadIdx.x + it * 256;
      if (idx < C::NIX) {
        float4 v = prex[it];
        if (INGN) {  // (as in conv_bf16x3_kernel::stage)
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
        }
        put3(xl + (idx / C::CVX) * PSX + (idx % C::CVX) * CPI, v, CIN);
      }
    }
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int idx = threadIdx.x + it * 256;
      float4 v = preg[it];
      if (INACT) {
        const float4 q = preg2[it];
        v.x *= act_grad_from_out(q.x, INACT), v.y *= act_grad_from_out(q.y, INACT);
        v.z *= act_grad_from_out(q.z, INACT), v.w *= act_grad_from_out(q.w, INACT);
      }
      if (!BF) bsum.x += v.x, bsum.y += v.y, bsum.z += v.z, bsum.w += v.w;
      put3(gl + (idx / C::CVG) * PSG + (idx % C::CVG) * CPI, v, COUT);
    }
  };

  if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
  auto run = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value, T0 = TW * W, T1 = (T0 + TW < C::T) ? T0 + TW : C::T;
    constexpr int NTW = T1 > T0 ? T1 - T0 : 0;                            // tiles of this wave (0: it only stages)
    constexpr int MB0 = T0 / NB, NG = NTW ? (T1 - 1) / NB - MB0 + 1 : 0;  // row blocks MB0 .. MB0+NG-1
    constexpr int NU = C::KSN * NG, NGD = NG ? NG : 1;                    // units per tile: k-steps x NG row blocks
    // operand fetch (hardware-transposing reads) of k-step ks: lane group lg covers tile row 2 ks + lg/2, 8 columns
    auto load_fb = [&](int ks, s16x8 (&F)[3][NB]) __attribute__((always_inline)) {
      // k-slot <-> pixel: lane group lg takes columns 4 lg .. 4 lg + 3 of tile row 2 ks (first read) and of row 2 ks + 1 (second
      // read) - the same assignment in load_fa, which is all the contraction needs; the 32 lanes of a read group then cover 8
      // CONSECUTIVE pixels of one row
      const unsigned short* gq = gl + (2 * ks * 16 + 4 * lg + tq) * PSG + tp * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) F[p][nb] = tr_read8(gq + p * COUT + nb * 16, gq + 16 * PSG + p * COUT + nb * 16);
    };
    auto load_fa = [&](int ks, int mb, s16x8 (&F)[3]) __attribute__((always_inline)) {
      const int tap = CIN == 32 ? mb >> 1 : mb, half = CIN == 32 ? mb & 1 : 0, ky = tap / K, kx = tap - K * ky;
      const unsigned short* xq = xl + ((2 * ks * S + ky) * WX_IC + (4 * lg + tq) * S + kx) * PSX + half * 16 + tp * 4;
#pragma unroll
      for (int p = 0; p < NP; ++p) F[p] = tr_read8(xq + p * CIN, xq + S * WX_IC * PSX + p * CIN);
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      __syncthreads();
      stage();
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      if (NTW > 0) {
        // NU units, software-pipelined: the operands of unit u+1 are requested before the MFMAs of unit u are issued,
        // and the reads are spread between those MFMAs
        s16x8 fa[2][3], fb[2][3][NB];
        load_fb(0, fb[0]);
        load_fa(0, MB0, fa[0]);
        // (compile-time recursion, not a loop: a loop of this size is only partially unrolled, which would index the
        // accumulator array dynamically and put it into scratch)
        wx_static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value;
          constexpr int ks = u / NGD, gi = u % NGD, mb = MB0 + gi;
          int nread = 0;
          if (u + 1 < NU) {
            const int ks2 = (u + 1) / NGD, gi2 = (u + 1) % NGD;
            if (gi2 == 0) load_fb(ks2, fb[ks2 & 1]), nread += 2 * NP * NB;
            load_fa(ks2, MB0 + gi2, fa[(u + 1) & 1]);
            nread += 2 * NP;
          }
          constexpr int NQ = BF ? 1 : 6;  // products per MAC (bf16 inputs: the one exact product)
          constexpr int PA[6] = {BF ? 0 : 2, 1, 0, 1, 0, 0};
          constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
          int nm = 0;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int t = NB * mb + nb;
            if (t >= T0 && t < T1) {
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                acc[t - T0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u & 1][PA[q]]),
                                                                      __builtin_bit_cast(bf16x8, fb[ks & 1][PB[q]][nb]),
                                                                      acc[t - T0], 0, 0, 0);
              nm += NQ;
            }
          }
          // issue order: MFMA, then up to ceil(nread / nm) of the next unit's reads
          const int per = nm ? (nread + nm - 1) / nm : 0;
#pragma unroll
          for (int g = 0; g < 12; ++g)
            if (g < nm) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (per == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              else if (per == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              else if (per == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
    // partial slab of this workgroup: [m = mb*16 + row][co]
    float* out = a.part + (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (C::MB * 16 * COUT);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int t = T0 + j, mb = t / NB, nb = t % NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(mb * 16 + lg * 4 + r) * COUT + nb * 16 + l16] = acc[j][r];
    }
  };
  switch (wave) {  // wave-uniform; every branch executes the same barriers
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    default: run(std::integral_constant<int, 3>{}); break;
  }
  if (a.bpart) {
    __syncthreads();
    const int vv = threadIdx.x % C::CVG, row = threadIdx.x / C::CVG;  // 256 / CVG rows of partial sums per channel chunk
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

template <int CIN, int COUT, int INACT = 0, bool INGN = false>
static int launch_wgrad_bf16x3(WgArgs a, float* gw, float* gb, int cin_real, hipStream_t s) {
  using C = WgCfg<CIN, COUT, 3, 3, 1>;
  using X = WxCfg<CIN, COUT>;
  static_assert(C::NCHUNK == 1 && C::NSPLIT == 1 && C::PART == X::MB * 16 * COUT, "slab layout of conv_wgrad_kernel");
  // x and gy are addressed per sample through buffer descriptors with 31-bit byte offsets
  if ((long)a.hin * a.win * CIN * 4 >= 0x7fff0000L || (long)a.hout * a.wout * COUT * 4 >= 0x7fff0000L)
    return DIS_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_bf16x3_kernel<CIN, COUT, INACT, false, 3, 1, 8, 3, false, INGN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, X::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + 7) / 8;
  const long ntiles = (long)a.n * tiles_y * tiles_x;
  long workers = (dis_f2_enabled() ? (long)dis_f2_wgrad_wpc() : 2L) * num_cus();
  if (workers > WG_WORKERS) workers = WG_WORKERS;
  if (workers > ntiles) workers = ntiles;
  const long elems = C::PART;
  float* tmp = a.part + (long)WG_WORKERS * elems;
  a.bpart = gb ? tmp + (long)WG_RSPLIT * elems : nullptr;
  // two-term fp16 split (conv_f16x2.hip; same slab layout), else the three-term kernel
  hipError_t le = dis_f2_enabled() ? dis_f2_wgrad_launch(a, CIN, COUT, INACT, workers, s) : hipErrorInvalidValue;
  if (le != hipSuccess && le != hipErrorInvalidValue) return (int)le;
  if (le != hipSuccess) DIS_TAG(CIN == 32 && COUT == 32 ? "conv_wgrad_bf16x3_kernel<32,32>" : "conv_wgrad_bf16x3_kernel<16|32,16|32>");
  if (le != hipSuccess)
    hipLaunchKernelGGL((conv_wgrad_bf16x3_kernel<CIN, COUT, INACT, false, 3, 1, 8, 3, false, INGN>), dim3((unsigned)workers),
                       dim3(256), X::LDS_BYTES, s, a);
  const long total = (long)C::MROWS * COUT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_grid(total, gb != nullptr)), dim3(64 * WG_RW), 0, s,
                     (const float*)a.part, gw, C::CINB, C::NCHUNK, C::NSPLIT, C::KHB, 3, 3, COUT, cin_real, C::PART,
                     (const float*)(gb ? a.bpart : nullptr), gb, (int)workers);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

/* Input gradient AND weight gradient of a 3x3 stride-1 pad-1 convolution c -> c (c = 32) in ONE launch (round 6, conv_bwd_fused.hip):
 * the gy halo an input-gradient tile stages in LDS also feeds the weight-gradient products of the pixels the tile owns, so gy (or
 * gpre, the GroupNorm-backward pass formed on load) is read once and never written for a second launch.  Reference: one autograd node
 * per Conv2d of ResNetBlock / Block2D3D, /root/reference/model/multi_frame_networks.py:338-345,514-542.
 *   operand of both products: coef != null: gpre = act'(q) (g k1_c + q kx + k0) as dis_conv2d_dgrad_f16x2_gnb forms it (stored to gpre_out
 *   too when that is non-null); coef == null: g itself (in_act == 0) or g act'(q) (q = the conv's activated output).
 *   input gradient: exactly dis_conv2d_dgrad_f16x2_gnb's forms (accumulate, ab_gn_x / ab_act_y / ab_out) - gx is bit-identical to it.
 *   weight gradient: x = the conv's input (n, hin, win, c); x_gn_stats != null: x is staged as GroupNorm(x) (dis_conv2d_wgrad_bf16x3_gn).
 *   x may be the same tensor as ab_gn_x or ab_act_y (it is then fetched once).  grad_w (c, c, 3, 3) with rows grad_w_row_stride
 *   floats apart (0: contiguous; the slice of a wider OIHW gradient otherwise), grad_b (c) or null;
 *   workspace: dis_conv2d_bwd_fused_workspace(c) floats.
 * DIS_ERR_UNSUPPORTED: no instance for this combination (the caller keeps the two launches). */
extern "C" long dis_conv2d_bwd_fused_workspace(int c) {
  if (c != 32) return -1;
  // the weight-gradient kernels' slabs + bias partials, then one spill slab per (workgroup, wave) for passes that end early (rare)
  return wgrad_ws<32, 32, 3, 3, 1>() + (long)WG_WORKERS * 4 * 9 * 32 * 32;
}
extern "C" int dis_conv2d_bwd_fused_f16x2(const float* g, const float* q, const float* coef, int in_act, float* gpre_out,
                                          const float* w_oihw, int w_o, int w_i, int w_row_stride, float* gx, int accumulate,
                                          const float* ab_gn_x, const float* ab_act_y, double* ab_out, const float* x,
                                          const double* x_gn_stats, const float* x_gn_gamma, const float* x_gn_beta, float x_gn_eps,
                                          float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int c,
                                          int grad_w_row_stride, void* stream) {
  if (!g || !w_oihw || !gx || !x || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0) return DIS_ERR_BAD_SHAPE;
  if (c != 32 || w_o != c || w_i != c) return DIS_ERR_UNSUPPORTED;
  if (in_act != DIS_ACT_NONE && in_act != DIS_ACT_SELU) return DIS_ERR_UNSUPPORTED;
  if ((coef || in_act) && !q) return DIS_ERR_NULL;
  if (gpre_out && !coef) return DIS_ERR_BAD_SHAPE;
  if ((ab_out != nullptr) != (ab_gn_x != nullptr) || (ab_act_y && (!ab_out || !accumulate))) return DIS_ERR_BAD_SHAPE;
  if (x_gn_stats && (!x_gn_gamma || !x_gn_beta)) return DIS_ERR_NULL;
  if (w_row_stride == 0) w_row_stride = w_i * 9;
  if (w_row_stride < w_i * 9) return DIS_ERR_BAD_SHAPE;
  // grad_w may be the (c, c, 3, 3) slice of a wider OIHW gradient (conv over a channel concatenation): its rows are then
  // grad_w_row_stride floats apart (a multiple of 9: whole input channels), 0 = contiguous
  if (grad_w_row_stride == 0) grad_w_row_stride = c * 9;
  if (grad_w_row_stride < c * 9 || grad_w_row_stride % 9) return DIS_ERR_BAD_SHAPE;
  static const bool off = getenv("DIS_BWD_FUSED") && getenv("DIS_BWD_FUSED")[0] == '0';
  if (off || !dis_f2_enabled()) return DIS_ERR_UNSUPPORTED;
  if ((long)hin * win * c * 4 >= 0x7fff0000L) return DIS_ERR_UNSUPPORTED;
  using C = WgCfg<32, 32, 3, 3, 1>;
  FbArgs f;
  ConvArgs& a = f.c;
  a.x = g; a.w = w_oihw; a.bias = nullptr; a.y = gx; a.stats = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hv = hin; a.wv = win; a.pad_y = 1; a.pad_x = 1;
  a.hf = hin; a.wf = win; a.osy = 1; a.ooy = 0; a.osx = 1; a.oox = 0;
  a.act = DIS_ACT_NONE; a.accum = accumulate ? 1 : 0;
  a.xscale = nullptr; a.yscale = nullptr;
  a.wmode = 1; a.w_o = w_o; a.w_i = w_i; a.w_rs = w_row_stride;
  a.xact = q;
  a.ldx = a.ldy = c; a.cx = a.cy = c; a.x_sub = a.y_sub = 0; a.nbias = 0; a.wtap0 = 0; a.wtap_step = 0;
  a.gn_stats = nullptr; a.gn_gamma = nullptr; a.gn_beta = nullptr; a.gn_eps = 0.f;
  a.ab_x = ab_gn_x; a.ab_out = ab_out; a.ab_slots = num_cus(); a.ab_act_y = ab_act_y;
  a.gnb_coef = coef; a.gnb_out = gpre_out; a.gnb_act = 0;
  const int xsrc = (ab_gn_x && x == ab_gn_x) ? 1 : ((ab_act_y && x == ab_act_y) ? 2 : 0);
  f.wx = x; f.wx_gn_stats = x_gn_stats; f.wx_gn_gamma = x_gn_gamma; f.wx_gn_beta = x_gn_beta; f.wx_gn_eps = x_gn_eps;
  const int tiles_x = (win + 15) / 16, tiles_y = (hin + 15) / 16;
  const long ntiles = (long)n * tiles_y * tiles_x;
  long grid = num_cus();
  if (grid > WG_WORKERS) grid = WG_WORKERS;
  if (grid > ntiles) grid = ntiles;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  const long elems = C::PART;
  static_assert(C::PART == 9 * 32 * 32, "slab layout");
  f.part = workspace;
  float* tmp = workspace + (long)WG_WORKERS * elems;
  f.bpart = grad_b ? tmp + (long)WG_RSPLIT * elems : nullptr;
  f.spill = workspace + wgrad_ws<32, 32, 3, 3, 1>();
  hipStream_t s = (hipStream_t)stream;
  hipError_t le = dis_fb_launch(f, in_act, x_gn_stats != nullptr, xsrc, grid, s);
  if (le == hipErrorInvalidValue) return DIS_ERR_UNSUPPORTED;
  if (le != hipSuccess) return (int)le;
  const long total = (long)C::MROWS * 32;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_grid(total, grad_b != nullptr)), dim3(64 * WG_RW), 0, s,
                     (const float*)f.part, grad_w, C::CINB, C::NCHUNK, C::NSPLIT, C::KHB, 3, 3, 32, grad_w_row_stride / 9, C::PART,
                     (const float*)(grad_b ? f.bpart : nullptr), grad_b, (int)grid);   // (cin_real = the row pitch in input channels)
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ---- wide layers as 32 x 32 channel-slice pairs (DispNetS, called from dis_convg_wgrad) ----
// gw[g][x][tap] = sum over the pair's worker slabs; slab element [tap * 32 + xc][gc] (WxCfg<32, 32, K>: m = 16 mb + row,
// mb = 2 tap + half).  Fixed summation order: deterministic.
// Tap-row groups (k = 7): slabs are [group][pair][worker][kh * k * 1024]; group z holds tap rows z * kh ...
__global__ __launch_bounds__(256) void wgrad_pairs_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw,
                                                                  int workers, int npx, int npairs, int cxw, int cgw,
                                                                  int k, int kh, int cob) {
  // (cob = gy channels per block: 32, or 16 for a layer with 16 output channels)
  const int psz = kh * k * 32 * cob, kk = k * k, ngrp = (k + kh - 1) / kh;
  const long total = (long)ngrp * npairs * psz;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % psz), pair = (int)((i / psz) % npairs), grp = (int)(i / ((long)psz * npairs));
    const int gc = e % cob, xc = (e / cob) & 31, tl = e / (32 * cob);
    const int ky = grp * kh + tl / k, tap = ky * k + tl % k;
    const int x = 32 * (pair % npx) + xc, g = cob * (pair / npx) + gc;
    if (x >= cxw || g >= cgw || ky >= k) continue;
    const float* p = part + ((long)grp * npairs + pair) * workers * psz + e;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // fp64 over the worker slabs (see wgrad_reduce_kernel)
    int wk = 0;
    // (sixteen slabs in flight per round: with four the launch was workers / 4 dependent memory round trips - 19 us at 27 launches per
    //  DIS-SF step, the latency chain, not the 22 MB it reads)
    for (; wk + 15 < workers; wk += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(long)(wk + u) * psz];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        s0 += (double)v[u];
        s1 += (double)v[u + 1];
        s2 += (double)v[u + 2];
        s3 += (double)v[u + 3];
      }
    }
    for (; wk + 3 < workers; wk += 4) {
      s0 += (double)p[(long)wk * psz];
      s1 += (double)p[(long)(wk + 1) * psz];
      s2 += (double)p[(long)(wk + 2) * psz];
      s3 += (double)p[(long)(wk + 3) * psz];
    }
    for (; wk < workers; ++wk) s0 += (double)p[(long)wk * psz];
    gw[((long)g * cxw + x) * kk + tap] = (float)((s0 + s1) + (s2 + s3));
  }
}

// tile rows of the (k, stride) instances: stride 2 halves the tile so that two workgroups still share a CU's LDS
static int wgrad_pairs_tr(int k, int stride) {
  if (stride == 1 && (k == 3 || k == 5 || k == 7)) return 8;
  if (stride == 2 && (k == 3 || k == 5)) return 4;
  return 0;
}
// (a layer with exactly 16 output channels at 3x3 / stride 1 - iconv1 - runs the 32 x 16 instance)
static int wgrad_pairs_cob(int cG, int k, int stride) { return (cG == 16 && k == 3 && stride == 1) ? 16 : 32; }
static void wgrad_pairs_plan(int n, int hG, int wG, int cX, int cG, int tr, int* npx, int* ngb, int* wpp) {
  *npx = (cX + 31) / 32;
  *ngb = cG == 16 ? 1 : (cG + 31) / 32;
  const long ntiles = (long)n * ((hG + tr - 1) / tr) * ((wG + 15) / 16);
  long per = (2L * num_cus() + (long)*npx * *ngb - 1) / ((long)*npx * *ngb);  // ~2 workgroups per CU in all
  if (per > ntiles) per = ntiles;
  if (per < 1) per = 1;
  *wpp = (int)per;
}
// eligibility and workspace (floats) of the slice-pair form; -1: use the fp32 kernel
long dis_wgrad_pairs_workspace(int n, int hX, int wX, int hG, int wG, int cX, int cG, int ldX, int ldG, int k,
                               int stride, int bf) {
  static const bool use3 = !(getenv("DIS_CONV_BF16X3") && getenv("DIS_CONV_BF16X3")[0] == '0');
  const int tr = wgrad_pairs_tr(k, stride);
  // (a 16-channel x slice fills half of its block)
  if (!use3 || tr == 0 || cX < 16 || (cG < 32 && wgrad_pairs_cob(cG, k, stride) != 16)) return -1;
  if ((long)hX * wX * ldX * 4 >= 0x7fff0000L || (long)hG * wG * ldG * 4 >= 0x7fff0000L) return -1;
  if (bf && ((ldX | ldG) & 7)) return -1;  // bf16 inputs: 16-byte staging items of 8 channels
  int npx, ngb, wpp;
  wgrad_pairs_plan(n, hG, wG, cX, cG, tr, &npx, &ngb, &wpp);
  return (long)npx * ngb * wpp * (k == 7 ? 2 * 4 * 7 : k * k) * 1024;  // (7x7: two groups of 4 tap rows)
}
template <int K, int S, int TR, int KH = K, int COB = 32, bool BF = false>
static int wgrad_pairs_launch(WgArgs a, float* grad_w, int cX_w, int cG_w, int ngb, int wpp, hipStream_t s) {
  using XC = WxCfg<32, COB, K, S, TR, KH, BF ? 1 : 3, BF ? 8 : 4>;
  static_assert(XC::LDS_BYTES <= 160 * 1024, "LDS budget exceeded");
  constexpr int NGRP = (K + KH - 1) / KH;
  auto kern = conv_wgrad_bf16x3_kernel<32, COB, 0, true, K, S, TR, KH, BF>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XC::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // two-term fp16 split (conv_f16x2.hip, default) for the fp32 pairs; same slab layout, same reduce launch
  hipError_t le = hipErrorInvalidValue;
  static const bool f2_all = !getenv("DIS_F2_WGRAD_3X3_ONLY");
  if (!BF && dis_f2_enabled() && (f2_all || (K == 3 && S == 1)))
    le = dis_f2_wgrad_pairs_launch(a, COB, (unsigned)wpp, (unsigned)(a.npx * ngb), K, S, KH, s);
  if (le != hipSuccess && le != hipErrorInvalidValue) return (int)le;
  if (le != hipSuccess) {
    DIS_TAG(BF ? "conv_wgrad_bf16x3_kernel<BF> slice pairs" : "conv_wgrad_bf16x3_kernel slice pairs");
    hipLaunchKernelGGL(kern, dim3((unsigned)wpp, (unsigned)(a.npx * ngb), NGRP), dim3(256), XC::LDS_BYTES, s, a);
  }
  const long total = (long)NGRP * a.npx * ngb * KH * K * 32 * COB;
  hipLaunchKernelGGL(wgrad_pairs_reduce_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, (const float*)a.part,
                     grad_w, wpp, a.npx, a.npx * ngb, cX_w, cG_w, K, KH, COB);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
int dis_wgrad_pairs_run(const float* X, int ldX, int xoff, int hX, int wX, int cX, int cX_w, const float* G, int ldG,
                        int goff, int hG, int wG, int cG, int cG_w, float* grad_w, float* workspace, int n, int k,
                        int stride, int pad, int bf, hipStream_t s) {
  // bf: X and G point at bf16 elements (ldX / xoff / ldG / goff in elements, multiples of 8)
  const int tr = wgrad_pairs_tr(k, stride);
  int npx, ngb, wpp;
  wgrad_pairs_plan(n, hG, wG, cX, cG, tr, &npx, &ngb, &wpp);
  WgArgs a;
  a.x = X; a.gy = G; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hX; a.win = wX; a.hout = hG; a.wout = wG; a.pad = pad;
  a.xscale = nullptr; a.gact = nullptr;
  a.ldx = ldX; a.xoff = xoff; a.cx = cX; a.ldg = ldG; a.goff = goff; a.cg = cG; a.npx = npx;
  if (bf) {
    if ((xoff | goff) & 7) return DIS_ERR_UNSUPPORTED;
    if (k == 3 && stride == 1 && wgrad_pairs_cob(cG, k, stride) == 16)
      return wgrad_pairs_launch<3, 1, 8, 3, 16, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    if (k == 3 && stride == 1) return wgrad_pairs_launch<3, 1, 8, 3, 32, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    if (k == 5 && stride == 1) return wgrad_pairs_launch<5, 1, 8, 5, 32, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    if (k == 3 && stride == 2) return wgrad_pairs_launch<3, 2, 4, 3, 32, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    if (k == 5 && stride == 2) return wgrad_pairs_launch<5, 2, 4, 5, 32, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    if (k == 7 && stride == 1) return wgrad_pairs_launch<7, 1, 8, 4, 32, true>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
    return DIS_ERR_UNSUPPORTED;
  }
  if (k == 3 && stride == 1 && wgrad_pairs_cob(cG, k, stride) == 16)
    return wgrad_pairs_launch<3, 1, 8, 3, 16>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  if (k == 3 && stride == 1) return wgrad_pairs_launch<3, 1, 8>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  if (k == 5 && stride == 1) return wgrad_pairs_launch<5, 1, 8>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  if (k == 3 && stride == 2) return wgrad_pairs_launch<3, 2, 4>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  if (k == 5 && stride == 2) return wgrad_pairs_launch<5, 2, 4>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  if (k == 7 && stride == 1) return wgrad_pairs_launch<7, 1, 8, 4>(a, grad_w, cX_w, cG_w, ngb, wpp, s);
  return DIS_ERR_UNSUPPORTED;
}

// 32 -> 32, 4 x 4, stride 2 (FuseNet's down convolution): the two-term fp16 kernel when the split is on (three products per MAC
// instead of the fp32 MFMA's rate; x staged once for all 16 taps instead of once per tap row), same slabs, same reduce launch
struct GnbW {  // GroupNorm backward applied on load of gy (WgArgs::gnb_coef): q = the GroupNorm's input, the gradient is written to out
  const float* q = nullptr;
  const float* coef = nullptr;
  float* out = nullptr;
  int act = 0;
};
static int launch_wgrad_k4s2(WgArgs a, float* gw, float* gb, int cin_real, hipStream_t s, GnbW gnb = GnbW()) {
  using C = WgCfg<32, 32, 4, 4, 2>;
  static_assert(C::NCHUNK == 1 && C::NSPLIT == 4 && C::KHB == 1 && C::PART == 4 * 32 * 32 && C::TROWS == 4, "slab layout of conv_wgrad_kernel");
  static const bool off = getenv("DIS_F2_WGRAD_K4S2") && getenv("DIS_F2_WGRAD_K4S2")[0] == '0';
  if (off || !dis_f2_enabled() || a.xscale || (long)a.hin * a.win * 32 * 4 >= 0x7fff0000L || (long)a.hout * a.wout * 32 * 4 >= 0x7fff0000L)
    return gnb.coef ? DIS_ERR_UNSUPPORTED : launch_wgrad<32, 32, 4, 4, 2>(a, gw, gb, cin_real, s);
  const int tiles_x = (a.wout + 15) / 16, tiles_y = (a.hout + 3) / 4;
  const long ntiles = (long)a.n * tiles_y * tiles_x;
  long workers = 2L * num_cus();
  if (workers > WG_WORKERS) workers = WG_WORKERS;
  if (workers > ntiles) workers = ntiles;
  const long elems = (long)C::NCHUNK * C::NSPLIT * C::PART;
  float* tmp = a.part + (long)WG_WORKERS * elems;
  a.bpart = gb ? tmp + (long)WG_RSPLIT * elems : nullptr;
  a.gact = gnb.q; a.gn_stats = nullptr; a.gn_gamma = nullptr; a.gn_beta = nullptr; a.gn_eps = 0.f;
  a.gnb_coef = gnb.coef; a.gnb_out = gnb.out;
  hipError_t le = dis_f2_wgrad_k4s2_launch(a, gnb.act, workers, s);
  if (le != hipSuccess) return (int)le;
  const long total = (long)C::NCHUNK * C::NSPLIT * C::MROWS * 32;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wgrad_reduce_grid(total, gb != nullptr)), dim3(64 * WG_RW), 0, s,
                     (const float*)a.part, gw, C::CINB, C::NCHUNK, C::NSPLIT, C::KHB, 4, 4, 32, cin_real, C::PART,
                     (const float*)(gb ? a.bpart : nullptr), gb, (int)workers);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

#define WG_CASE(CI, CO, K_, S_) \
  if (cin == CI && cout == CO && k == K_ && stride == S_) return launch_wgrad<CI, CO, K_, K_, S_>(a, gw, gb, cin_real, s);
#define WS_CASE(CI, CO, K_, S_) \
  if (cin == CI && cout == CO && k == K_ && stride == S_) return wgrad_ws<CI, CO, K_, K_, S_>();

static int dispatch_wgrad(const WgArgs& a, float* gw, float* gb, int cin_real, int cin, int cout, int k,
                          int stride, hipStream_t s) {
  WG_CASE(4, 16, 4, 2)
  WG_CASE(4, 16, 3, 1)
  WG_CASE(16, 16, 3, 1)
  WG_CASE(16, 32, 3, 1)
  WG_CASE(32, 16, 3, 1)
  WG_CASE(32, 32, 3, 1)
  WG_CASE(48, 32, 3, 1)
  WG_CASE(96, 32, 3, 1)
  WG_CASE(128, 32, 1, 1)
  if (cin == 32 && cout == 32 && k == 4 && stride == 2) return launch_wgrad_k4s2(a, gw, gb, cin_real, s);
  WG_CASE(4, 32, 7, 2)   // DispNetS conv1 (2 -> 32, k7 s2): all 49 taps in one pass over the pixels
  return DIS_ERR_UNSUPPORTED;
}

/* Weight gradient of FuseNet's 4 x 4 stride-2 pad-1 convolution (32 -> 32; reference model/multi_frame_networks.py:338-345,
 * Block2D3D.conv2_1) that is followed by [SELU ->] GroupNorm, WITH that GroupNorm's backward elementwise pass applied while gy is
 * staged (round 5): gpre = act'(q) (g k1_c + q kx + k0), g the gradient wrt the GroupNorm's output (n, hout, wout, 32), q the
 * GroupNorm's input, coef (n, 34) from dis_gn_bwd_coef; grad_w / grad_b are the gradients for gpre, and gpre is stored to gpre_out
 * for the input-gradient launches (dis_conv2d_dgrad_strided).  workspace: dis_conv2d_wgrad_workspace(32, 32, 4, 2) floats.
 * Two-term fp16 kernel only (DIS_ERR_UNSUPPORTED otherwise: the caller runs dis_gn_bwd_apply_coef and dis_conv2d_wgrad). */
extern "C" int dis_conv2d_wgrad_k4s2_f16x2_gnb(const float* x, const float* g, const float* q, const float* coef, int in_act,
                                               float* gpre_out, float* grad_w, float* grad_b, float* workspace, int n, int hin,
                                               int win, void* stream) {
  if (!x || !g || !q || !coef || !gpre_out || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin < 2 || win < 2) return DIS_ERR_BAD_SHAPE;
  if (in_act != DIS_ACT_NONE && in_act != DIS_ACT_SELU) return DIS_ERR_UNSUPPORTED;
  WgArgs a;
  a.x = x; a.gy = g; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hout = (hin + 2 - 4) / 2 + 1; a.wout = (win + 2 - 4) / 2 + 1; a.pad = 1;
  a.xscale = nullptr; a.gact = nullptr;
  a.gn_stats = nullptr; a.gn_gamma = nullptr; a.gn_beta = nullptr; a.gn_eps = 0.f;
  GnbW gnb;
  gnb.q = q; gnb.coef = coef; gnb.out = gpre_out; gnb.act = in_act;
  return launch_wgrad_k4s2(a, grad_w, grad_b, 32, (hipStream_t)stream, gnb);
}

extern "C" long dis_conv2d_wgrad_workspace(int cin, int cout, int k, int stride) {
  WS_CASE(4, 16, 4, 2)
  WS_CASE(4, 16, 3, 1)
  WS_CASE(16, 16, 3, 1)
  WS_CASE(16, 32, 3, 1)
  WS_CASE(32, 16, 3, 1)
  WS_CASE(32, 32, 3, 1)
  WS_CASE(48, 32, 3, 1)
  WS_CASE(96, 32, 3, 1)
  WS_CASE(128, 32, 1, 1)
  WS_CASE(32, 32, 4, 2)
  WS_CASE(4, 32, 7, 2)
  return -1;
}

// same contract as dis_conv2d_wgrad (and the same workspace size) for cin = cout = 32, k = 3, stride 1
static int wgrad_bf16x3_entry(const float* x, const float* gy, const float* gact, int inact, float* grad_w, float* grad_b,
                              float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k,
                              int stride, int pad, void* stream, GnIn gn = GnIn()) {
  if (!x || !gy || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0) return DIS_ERR_BAD_SHAPE;
  if (!bx_shape_ok(cin_pad, cout, k, stride) || cin_real <= 0 || cin_real > cin_pad) return DIS_ERR_UNSUPPORTED;
  if (inact && (!gact || (inact != DIS_ACT_SELU && inact != DIS_ACT_RELU))) return DIS_ERR_UNSUPPORTED;
  const int hout = hin + 2 * pad - 2, wout = win + 2 * pad - 2;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  WgArgs a;
  a.x = x; a.gy = gy; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.pad = pad;
  a.xscale = nullptr;
  a.gact = gact;
  a.gn_stats = gn.stats; a.gn_gamma = gn.gamma; a.gn_beta = gn.beta; a.gn_eps = gn.eps;
  hipStream_t s = (hipStream_t)stream;
  if (gn.stats) {  // x = GroupNorm(x) on load: the square shapes of the conv -> act -> GroupNorm -> conv chains
    if (!gn.gamma || !gn.beta) return DIS_ERR_NULL;
    if (cin_pad != cout || cin_real != cin_pad) return DIS_ERR_UNSUPPORTED;
    // (the consumers' own activation gradient arrives folded into gy: GroupNorm's backward emits the pre-activation gradient)
    if (inact != 0) return DIS_ERR_UNSUPPORTED;
    if (cout == 32) return launch_wgrad_bf16x3<32, 32, 0, true>(a, grad_w, grad_b, cin_real, s);
    return launch_wgrad_bf16x3<16, 16, 0, true>(a, grad_w, grad_b, cin_real, s);
  }
#define WX_DISPATCH(CI, CO)                                                                                      \
  if (cin_pad == CI && cout == CO) {                                                                             \
    if (inact == DIS_ACT_SELU) return launch_wgrad_bf16x3<CI, CO, DIS_ACT_SELU>(a, grad_w, grad_b, cin_real, s); \
    if (inact == DIS_ACT_RELU) return launch_wgrad_bf16x3<CI, CO, DIS_ACT_RELU>(a, grad_w, grad_b, cin_real, s); \
    return launch_wgrad_bf16x3<CI, CO, 0>(a, grad_w, grad_b, cin_real, s);                                       \
  }
  WX_DISPATCH(32, 32) WX_DISPATCH(16, 16) WX_DISPATCH(16, 32) WX_DISPATCH(32, 16)
#undef WX_DISPATCH
  return DIS_ERR_UNSUPPORTED;
}
extern "C" int dis_conv2d_wgrad_bf16x3(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace,
                                       int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride,
                                       int pad, void* stream) {
  return wgrad_bf16x3_entry(x, gy, nullptr, 0, grad_w, grad_b, workspace, n, hin, win, cin_pad, cin_real, cout, k, stride,
                            pad, stream);
}
/* The same for a convolution that was followed by an activation: gy is the gradient wrt the activation's OUTPUT y, and
 * the kernel stages gy * act'(y) (weight and bias gradient of the pre-activation).  Replaces dis_act_bwd + the above. */
extern "C" int dis_conv2d_wgrad_bf16x3_act(const float* x, const float* gy, const float* y, int act, float* grad_w,
                                           float* grad_b, float* workspace, int n, int hin, int win, int cin_pad,
                                           int cin_real, int cout, int k, int stride, int pad, void* stream) {
  if (!y || act == DIS_ACT_NONE) return DIS_ERR_UNSUPPORTED;
  return wgrad_bf16x3_entry(x, gy, y, act, grad_w, grad_b, workspace, n, hin, win, cin_pad, cin_real, cout, k, stride, pad,
                            stream);
}

/* dis_conv2d_wgrad_bf16x3 with x = GroupNorm(x) applied on load (see dis_conv2d_fwd_bf16x3_gn); gy is the gradient wrt the
 * conv's pre-activation output.  cin_pad == cin_real == cout in {16, 32}. */
extern "C" int dis_conv2d_wgrad_bf16x3_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta,
                                          float gn_eps, const float* gy, float* grad_w, float* grad_b, float* workspace, int n,
                                          int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad,
                                          void* stream) {
  if (!gn_stats || !gn_gamma || !gn_beta) return DIS_ERR_NULL;
  GnIn gn;
  gn.stats = gn_stats; gn.gamma = gn_gamma; gn.beta = gn_beta; gn.eps = gn_eps;
  return wgrad_bf16x3_entry(x, gy, nullptr, 0, grad_w, grad_b, workspace, n, hin, win, cin_pad, cin_real, cout, k, stride, pad,
                            stream, gn);
}

extern "C" int dis_conv2d_wgrad(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace,
                                int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride,
                                int pad, void* stream) {
  if (!x || !gy || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || cin_pad <= 0 || cin_real <= 0 || cin_real > cin_pad || cout <= 0)
    return DIS_ERR_BAD_SHAPE;
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  WgArgs a;
  a.x = x; a.gy = gy; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.pad = pad;
  a.xscale = nullptr;
  return dispatch_wgrad(a, grad_w, grad_b, cin_real, cin_pad, cout, k, stride, (hipStream_t)stream);
}

/* dis_conv2d_wgrad with the activation gradient applied while gy is staged (round 6): grad_w / grad_b are the gradients for
 * gpre = gy * act'(y), y = the conv's activated output - for the layers whose input gradient is never needed (FuseNet's stems:
 * conv1 4 -> 16 k4 s2, amb_conv 4 -> 16 k3 s1; reference model/multi_frame_networks.py:216-227, 229-233), where the separate
 * dis_act_bwd pass (read gy, y; write gpre) existed for this launch alone.  Exact-fp32 MFMA kernel (conv_wgrad_kernel).
 * DIS_ERR_UNSUPPORTED: another shape (the caller runs dis_act_bwd + dis_conv2d_wgrad). */
extern "C" int dis_conv2d_wgrad_act(const float* x, const float* gy, const float* y, int act, float* grad_w, float* grad_b,
                                    float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k,
                                    int stride, int pad, void* stream) {
  if (!x || !gy || !y || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || cin_pad <= 0 || cin_real <= 0 || cin_real > cin_pad || cout <= 0)
    return DIS_ERR_BAD_SHAPE;
  if (act != DIS_ACT_SELU && act != DIS_ACT_RELU) return DIS_ERR_UNSUPPORTED;
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  WgArgs a;
  a.x = x; a.gy = gy; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.pad = pad;
  a.xscale = nullptr;
  a.gact = y; a.gact_act = act;
  hipStream_t s = (hipStream_t)stream;
  if (cin_pad == 4 && cout == 16 && k == 4 && stride == 2) return launch_wgrad<4, 16, 4, 4, 2, true>(a, grad_w, grad_b, cin_real, s);
  if (cin_pad == 4 && cout == 16 && k == 3 && stride == 1) return launch_wgrad<4, 16, 3, 3, 1, true>(a, grad_w, grad_b, cin_real, s);
  return DIS_ERR_UNSUPPORTED;
}

extern "C" int dis_conv2d_wgrad_scaled(const float* x, const float* xscale, const float* gy, float* grad_w, float* grad_b,
                                       float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout,
                                       int k, int stride, int pad, void* stream) {
  if (!x || !gy || !grad_w || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || hin <= 0 || win <= 0 || cin_pad <= 0 || cin_real <= 0 || cin_real > cin_pad || cout <= 0)
    return DIS_ERR_BAD_SHAPE;
  const int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
  if (hout <= 0 || wout <= 0) return DIS_ERR_BAD_SHAPE;
  WgArgs a;
  a.x = x; a.gy = gy; a.part = workspace; a.bpart = nullptr;
  a.n = n; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.pad = pad;
  a.xscale = xscale;
  return dispatch_wgrad(a, grad_w, grad_b, cin_real, cin_pad, cout, k, stride, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// disparity head: Conv2d(cin,1,3,pad 1) + alpha*sigmoid(x-offset)  (bandwidth-bound, VALU)
// ------------------------------------------------------------------------------------------------
// One thread per (pixel, 4-channel group): consecutive lanes read consecutive 16-byte pieces of a pixel's channel line (one
// thread per pixel read 16 bytes every CIN * 4: a quarter of every sector used), the CIN / 4 partial dot products of a pixel are
// folded with xor-shuffles.
template <int CIN>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                const float* __restrict__ b, float* __restrict__ y, int n, int h, int wd,
                                float alpha, float offset) {
  constexpr int NQ = CIN / 4;
  __shared__ float4 ws[9 * NQ];
  for (int i = threadIdx.x; i < 9 * CIN; i += blockDim.x) {
    const int tap = i / CIN, ci = i % CIN;
    ((float*)ws)[i] = w[ci * 9 + tap];  // (1,cin,3,3) -> [tap][ci]
  }
  __syncthreads();
  const int quad = threadIdx.x % NQ;
  const float bias = b[0] - offset;
  const long total = (long)n * h * wd * NQ;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i / NQ;
    const int px = (int)(q % wd), py = (int)((q / wd) % h);
    const long nb = q / ((long)h * wd);
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = py + ky - 1;
      if (iy < 0 || iy >= h) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = px + kx - 1;
        if (ix < 0 || ix >= wd) continue;
        const float4 v = *(const float4*)(x + ((nb * h + iy) * wd + ix) * CIN + quad * 4);
        const float4 wt = ws[(ky * 3 + kx) * NQ + quad];
        acc += (v.x * wt.x + v.y * wt.y) + (v.z * wt.z + v.w * wt.w);
      }
    }
#pragma unroll
    for (int off = NQ / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (quad == 0) y[q] = alpha / (1.f + expf(-(acc + bias)));
  }
}

// Tiled form (round 5): a workgroup owns HD_TH rows x (256 / NQ) columns of output pixels and walks DOWN the rows with three running
// sums per thread (the output rows r - 1, r, r + 1 that input row r feeds), so an input pixel is loaded three times (left / centre /
// right of neighbouring threads of one wave: L1 hits) instead of nine, and never by a workgroup of another XCD except for the two
// halo rows.  Every output accumulates its taps in the order of the kernel above (ky outer, kx inner, out-of-image taps skipped):
// bit-identical results.
#define HD_TH 16
template <int CIN>
__global__ __launch_bounds__(256) void head_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y, int h, int wd,
                                                             float alpha, float offset, int tiles_x, int tiles_y) {
  constexpr int NQ = CIN / 4, PXB = 256 / NQ;
  __shared__ float4 ws[9 * NQ];
  for (int i = threadIdx.x; i < 9 * CIN; i += blockDim.x) {
    const int tap = i / CIN, ci = i % CIN;
    ((float*)ws)[i] = w[ci * 9 + tap];  // (1,cin,3,3) -> [tap][ci]
  }
  __syncthreads();
  int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y;
  const long nb = tile / tiles_y;
  const int quad = threadIdx.x % NQ, px = tx * PXB + (int)threadIdx.x / NQ;
  const float bias = b[0] - offset;
  const int y0 = ty * HD_TH, y1 = min(y0 + HD_TH, h);
  const bool col = px < wd;
  const int pxc = col ? px : wd - 1;   // (idle columns of a ragged tile read the last column and store nothing)
  const bool okl = pxc - 1 >= 0, okr = pxc + 1 < wd;
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = ws[t * NQ + quad];
  auto dot = [](const float4& v, const float4& q) { return (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w); };
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;   // sums of the output rows r - 1, r, r + 1
  for (int r = y0 - 1; r <= y1; ++r) {
    if (r >= 0 && r < h) {
      const float* row = x + ((nb * h + r) * wd + pxc) * CIN + quad * 4;
      const float4 vm = *(const float4*)(row);
      const float4 vl = okl ? *(const float4*)(row - CIN) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vr = okr ? *(const float4*)(row + CIN) : make_float4(0.f, 0.f, 0.f, 0.f);
      // (input row r is tap row ky = 2 of output row r - 1, ky = 1 of row r, ky = 0 of row r + 1)
      if (okl) a0 += dot(vl, wt[6]);
      a0 += dot(vm, wt[7]);
      if (okr) a0 += dot(vr, wt[8]);
      if (okl) a1 += dot(vl, wt[3]);
      a1 += dot(vm, wt[4]);
      if (okr) a1 += dot(vr, wt[5]);
      if (okl) a2 += dot(vl, wt[0]);
      a2 += dot(vm, wt[1]);
      if (okr) a2 += dot(vr, wt[2]);
    }
    if (r - 1 >= y0) {   // output row r - 1 is complete
      float acc = a0;
#pragma unroll
      for (int off = NQ / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (quad == 0 && col) y[(nb * h + (r - 1)) * wd + px] = alpha / (1.f + expf(-(acc + bias)));
    }
    a0 = a1, a1 = a2, a2 = 0.f;
  }
}

// pre-sigmoid gradient plane
__global__ void head_gpre_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gp,
                                 float alpha, long count) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
    const float s = y[i] / alpha;
    gp[i] = gy[i] * alpha * s * (1.f - s);
  }
}

// Input gradient and weight / bias gradient of the head in one pass.  One thread per (pixel q, 4-channel group): with
// g[o] the pre-sigmoid gradient at the 9 output pixels o = q - (tap - 1) that see q,
//   gx[q][c]   = sum_tap g[o] w[tap][c]          dW[tap][c] += g[o] x[q][c]          db += g[q]
// so x is read once (coalesced float4 rows) and the neighbourhood only through the 1-channel g.  Per-thread fp32
// partials -> fp64 sums over the lanes that share a channel group -> one slab per block (head_reduce_kernel adds the
// slabs in a fixed order: deterministic, no atomics).
#define HEAD_SLABS 2048   // (512 until round 5: two workgroups per CU left the kernel waiting on its loads, 175 us for 450 MB)
template <int CIN>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ gp, float* __restrict__ gx,
                                                         double* __restrict__ slab, int n, int h, int wd) {
  constexpr int NQ = CIN / 4, NP = 9 * CIN + 1;
  __shared__ float4 ws[9 * NQ];
  __shared__ double red[4][NP];
  for (int i = threadIdx.x; i < 9 * CIN; i += blockDim.x) {
    const int tap = i / CIN, ci = i % CIN;
    ((float*)ws)[i] = w[ci * 9 + tap];
  }
  __syncthreads();
  const int quad = threadIdx.x % NQ;
  float4 gw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) gw[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  float gb = 0.f;
  const long total = (long)n * h * wd * NQ;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i / NQ;
    const int px = (int)(q % wd), py = (int)((q / wd) % h);
    const long nb = q / ((long)h * wd);
    const float4 xv = *(const float4*)(x + q * CIN + quad * 4);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int oy = py - ky + 1, ox = px - kx + 1;
        const bool ok = oy >= 0 && oy < h && ox >= 0 && ox < wd;
        const float g = ok ? gp[(nb * h + (ok ? oy : 0)) * wd + (ok ? ox : 0)] : 0.f;
        const int t = ky * 3 + kx;
        const float4 wt = ws[t * NQ + quad];
        o.x += g * wt.x, o.y += g * wt.y, o.z += g * wt.z, o.w += g * wt.w;
        gw[t].x += g * xv.x, gw[t].y += g * xv.y, gw[t].z += g * xv.z, gw[t].w += g * xv.w;
        if (t == 4 && quad == 0) gb += g;
      }
    *(float4*)(gx + q * CIN + quad * 4) = o;
  }
  // lanes with the same channel group: l % NQ == quad (NQ divides 64)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto quad_sum = [&](double v) {
#pragma unroll
    for (int off = 32; off >= NQ; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
  };
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const double r0 = quad_sum((double)gw[t].x), r1 = quad_sum((double)gw[t].y), r2 = quad_sum((double)gw[t].z),
                 r3 = quad_sum((double)gw[t].w);
    if (lane < NQ) {
      red[wave][(lane * 4 + 0) * 9 + t] = r0;
      red[wave][(lane * 4 + 1) * 9 + t] = r1;
      red[wave][(lane * 4 + 2) * 9 + t] = r2;
      red[wave][(lane * 4 + 3) * 9 + t] = r3;
    }
  }
  const double rb = wave_sum_d((double)gb);
  if (lane == 0) red[wave][9 * CIN] = rb;
  __syncthreads();
  for (int i = threadIdx.x; i < NP; i += blockDim.x)
    slab[(long)blockIdx.x * NP + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// grad_w (cin*9, [ci][tap]) and grad_b from the block slabs
__global__ __launch_bounds__(64) void head_reduce_kernel(const double* __restrict__ slab, float* __restrict__ gw,
                                                         float* __restrict__ gb, int nslab, int np) {
  const int i = blockIdx.x;  // one parameter per (single-wave) block
  double s = 0.0;
  for (int k = threadIdx.x; k < nslab; k += 64) s += slab[(long)k * np + i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    if (i < np - 1) gw[i] = (float)s;
    else gb[0] = (float)s;
  }
}

extern "C" int dis_disp_head_fwd(const float* x, const float* w, const float* b, float* y, int n, int h, int wd,
                                 int cin, float alpha, float offset, void* stream) {
  if (!x || !w || !b || !y) return DIS_ERR_NULL;
  if (n <= 0 || h <= 0 || wd <= 0) return DIS_ERR_BAD_SHAPE;
  if (cin == 16 || cin == 32) {
    const int pxb = 256 / (cin / 4), tiles_x = dis_cdiv(wd, pxb), tiles_y = dis_cdiv(h, HD_TH);
    const long tiles = (long)n * tiles_x * tiles_y;
    const char* env = getenv("DIS_HEAD_TILED");   // (=0: the grid-stride kernel; read per call: tests compare the two forms)
    const bool tiled = !(env && env[0] == '0');
    if (tiled && tiles <= 0x7fffffffL) {
      if (cin == 16)
        hipLaunchKernelGGL(head_fwd_tiled_kernel<16>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w, b, y, h, wd, alpha,
                           offset, tiles_x, tiles_y);
      else
        hipLaunchKernelGGL(head_fwd_tiled_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w, b, y, h, wd, alpha,
                           offset, tiles_x, tiles_y);
      DIS_CHECK_LAUNCH();
      return DIS_OK;
    }
  }
  const dim3 grid(dis_ew_grid((long)n * h * wd * (cin / 4), 256));
  if (cin == 16) hipLaunchKernelGGL(head_fwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, y, n, h, wd, alpha, offset);
  else if (cin == 32) hipLaunchKernelGGL(head_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, y, n, h, wd, alpha, offset);
  else return DIS_ERR_UNSUPPORTED;
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}
extern "C" long dis_disp_head_bwd_workspace(int n, int h, int wd, int cin) {
  if (n <= 0 || h <= 0 || wd <= 0 || (cin != 16 && cin != 32)) return DIS_ERR_UNSUPPORTED;
  return (((long)n * h * wd + 1) & ~1L) + 2L * HEAD_SLABS * (9 * cin + 1);  // floats: pre-sigmoid gradient + fp64 slabs
}
extern "C" int dis_disp_head_bwd(const float* x, const float* w, const float* y, const float* gy, float* gx,
                                 float* grad_w, float* grad_b, float* workspace, int n, int h, int wd, int cin,
                                 float alpha, void* stream) {
  if (!x || !w || !y || !gy || !gx || !grad_w || !grad_b || !workspace) return DIS_ERR_NULL;
  if (n <= 0 || h <= 0 || wd <= 0) return DIS_ERR_BAD_SHAPE;
  if (cin != 16 && cin != 32) return DIS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)n * h * wd;
  hipLaunchKernelGGL(head_gpre_kernel, dim3(dis_ew_grid(total, 256)), dim3(256), 0, s, y, gy, workspace, alpha, total);
  int grid = dis_ew_grid(total * (cin / 4), 256);
  if (grid > HEAD_SLABS) grid = HEAD_SLABS;
  double* slab = (double*)(workspace + ((total + 1) & ~1L));  // 8-byte aligned behind the gradient plane
  if (cin == 16) hipLaunchKernelGGL(head_bwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, w, (const float*)workspace, gx, slab, n, h, wd);
  else hipLaunchKernelGGL(head_bwd_kernel<32>, dim3(grid), dim3(256), 0, s, x, w, (const float*)workspace, gx, slab, n, h, wd);
  hipLaunchKernelGGL(head_reduce_kernel, dim3(9 * cin + 1), dim3(64), 0, s, (const double*)slab, grad_w, grad_b, grid,
                     9 * cin + 1);
  DIS_CHECK_LAUNCH();
  return DIS_OK;
}

// ---- split-neutral names of the `_bf16x3` entry points (include/dis_hip.h): aliases of the functions above
#ifndef __HIP_DEVICE_COMPILE__
extern "C" int dis_conv2d_fwd_split_oihw(const float* x, const float* w_oihw, int mode, int w_o, int w_i, int w_row_stride, const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act, void* stream) __attribute__((alias("dis_conv2d_fwd_bf16x3_oihw")));
extern "C" int dis_conv2d_fwd_split_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* w_oihw, int w_o, int w_i, int w_row_stride, const float* bias, float* y, double* stats, int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int act, void* stream) __attribute__((alias("dis_conv2d_fwd_bf16x3_gn")));
extern "C" int dis_conv2d_dgrad_split_gnsums(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream) __attribute__((alias("dis_conv2d_dgrad_bf16x3_gnsums")));
extern "C" int dis_conv2d_dgrad_split_gnsums_res(const float* gy, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* act_y, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream) __attribute__((alias("dis_conv2d_dgrad_bf16x3_gnsums_res")));
extern "C" int dis_conv2d_dgrad_split_act_gnsums_res(const float* gy, const float* y, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* g, const float* act_y, const float* gn_x, double* ab_out, int n, int hin, int win, int cin, int cout, int pad, void* stream) __attribute__((alias("dis_conv2d_dgrad_bf16x3_act_gnsums_res")));
extern "C" int dis_conv2d_dgrad_split_act(const float* gy, const float* y, int act, const float* w_oihw, int w_o, int w_i, int w_row_stride, float* gx, int n, int hin, int win, int cin, int cout, int pad, int accumulate, void* stream) __attribute__((alias("dis_conv2d_dgrad_bf16x3_act")));
extern "C" int dis_conv2d_wgrad_split(const float* x, const float* gy, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream) __attribute__((alias("dis_conv2d_wgrad_bf16x3")));
extern "C" int dis_conv2d_wgrad_split_act(const float* x, const float* gy, const float* y, int act, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream) __attribute__((alias("dis_conv2d_wgrad_bf16x3_act")));
extern "C" int dis_conv2d_wgrad_split_gn(const float* x, const double* gn_stats, const float* gn_gamma, const float* gn_beta, float gn_eps, const float* gy, float* grad_w, float* grad_b, float* workspace, int n, int hin, int win, int cin_pad, int cin_real, int cout, int k, int stride, int pad, void* stream) __attribute__((alias("dis_conv2d_wgrad_bf16x3_gn")));
#endif
