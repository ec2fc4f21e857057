// ea_lara_seglin.hip -- LARA 'adaptive-1d' proposals (lara.py:56-63,84-127) WITHOUT the folded projection (round 4):
//     q_bar_l = mean_{n in segment l} LayerNorm(G_q q_n + g_q),     k_bar_l likewise,
// generator Linear, LayerNorm and segment mean in one pass over the stored q / k rows; nothing token-sized is written in the
// forward, and the backward writes dq / dk (accumulated into the attention core's gradient) and the generator gradients only.
//
// Rounds 1-3 folded the generator Linear into the qkv projection (W' = G W_head: two more groups of output columns).  That
// made every one of the three 512-wide library GEMMs of the layer (forward, input gradient, weight gradient) 5C instead of 3C
// columns wide: +67 % of their FLOPs, ~250 us of the 1.27 ms cfg5 step -- for a product that costs 16 k FLOPs per token-head.
// Here the [64 x 64] product runs on the MFMA inside the segment kernels; the projection stays 3C wide.
//
// Layouts (v_mfma_f32_16x16x32, D[m = 4g + r][n = li]):
//   * forward and the dq / dk pass: z^T[out][token] = G X^T -- A = rows of G (resident, 32 VGPRs), B = the token rows as
//     they are loaded (lane (g, li): token li, channels 32 ks + 8 g ..).  A lane owns 16 channels of ONE token: LayerNorm
//     statistics are in-lane sums + two v_permlane swaps.  d x^T = G^T d z^T chains register-for-register: a lane's D values
//     of two adjacent out-tiles are exactly its eight k-slots of the next MFMA (the P -> PV trick of the window kernels).
//   * the dG pass: z[token][out] (operands swapped) -- a lane owns one out-channel of FOUR tokens, LayerNorm statistics are
//     16-lane DPP sums, and dz as well as X (brought into the same layout by four exact MFMAs with a 0 / 1 pattern) are the
//     k-slots of dG[out][in] = sum_tokens dz[token][out] x[token][in] without any LDS transpose.
// One wave per (b, h, side, segment) -- the dG pass: per group of segments, its [64 x 64] partial in registers.
#include <stdlib.h>
#include "ea_lara_segment.h"

namespace ea {

struct SegLinP {
  char *q, *k;                       // stored q / k rows (element type), [B,H,N,64] views
  int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn;
  char *dq, *dk;                     // backward: accumulated into
  int64_t dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn;
  const float *Gq, *Gk, *gqb, *gkb;  // generator Linear [64, 64] (out, in), bias [64]
  const float *lnq_w, *lnq_b, *lnk_w, *lnk_b;   // LayerNorm weight / bias [64]
  float *qbar, *kbar;                // forward outputs [B*H, L, 64]
  const float *d_qbar, *d_kbar;      // backward inputs
  float* part;                       // backward: [B*H*groups, 2, 4, 64] partial sums (d ln_w, d ln_b, d g_b, unused)
  f32x4* stats;                      // backward scratch [B*H, 2, N]: (mean, rstd, mean(a xhat), -) of every token's LayerNorm
  float* dG_part;                    // backward: [nG, 2, 64, 64] partial sums of dG (nG = B*H*groups)
  int B, H, N, L, segs, nshort, groups, seg_per_group, cgroups, cseg_per_group;
  // the estimator's last dq correction riding on the dq / dk pass (ea_lara_seglin_bwd_fin, round 6): dq_n -= s sum_c t[c,n] (u qbar)_c
  const float *fin_qbar, *fin_uq, *fin_lse;   // [B*H, C, 64], [B*H, C, 64], [B*H, C]; null: none
  int C;
  float fin_scale, fin_scale_log2;
};

namespace {

template <typename E> struct GenA {      // G as the A operand of z^T = G X^T: lane (g, li): G[16 mt + li][32 ks + 8 g ..]
  typename E::x8 a[4][2];
  EA_DEV void load(const float* G, int g, int li) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float* s = G + (16 * mt + li) * 64 + 32 * ks + 8 * g;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
        const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[mt][ks] = as_x8<E>(pack8<E>(f));
      }
  }
};

EA_DEV int seg_start(const SegLinP& p, int l) { return l < p.nshort ? l * p.segs : p.nshort * p.segs + (l - p.nshort) * (p.segs + 1); }
EA_DEV int seg_len(const SegLinP& p, int l) { return l < p.nshort ? p.segs : p.segs + 1; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// Cursor over the (segment, step) pairs of a wave's group of consecutive segments -- uniform values.  Round 6: a wave walks
// its segments as ONE stream of TS-token steps with the row loads of the next DEPTH steps in flight ACROSS the segment
// boundaries.  Until then every 84-token segment (cfg5: N = 4096, 49 segments) started cold -- two row loads issued, a full
// memory round trip waited for, then 6 steps with two more steps' rows requested past the segment's end and dropped: the three
// passes ran at 1.4 - 3.6 TB/s (profiles/r06cfg5_lara_kernel_stats.csv: 48 + 111 + 108 us).
template <int TS> struct SegCur {
  int l, it, s0, len, steps;
  EA_DEV void set(const SegLinP& p, int l_) {
    l = l_; it = 0; s0 = seg_start(p, l_); len = seg_len(p, l_); steps = (len + TS - 1) / TS;
  }
  EA_DEV int tok(int o) const { return s0 + min(it * TS + o, len - 1); }          // row o of the step, clamped to the segment
  EA_DEV void advance(const SegLinP& p, int l1) {                                  // past the group's end: its last step again
    if (++it == steps) { if (l + 1 < l1) set(p, l + 1); else it = steps - 1; }    // (a harmless re-read nobody consumes)
  }
};

// ------------------------------------------------------------------------------------------------------------
// token-column passes: forward (BWD = false) and the dq / dk pass (BWD = true)
// NCT > 0 (backward only): the q-side waves also apply the correction ea_lara_bwd_finish applies (lara.py:223 differentiated:
// t = softmax over the sequence of s qbar_c . q_n) to the rows they rewrite anyway -- the q rows are in their registers as the B
// operand of the generator product, t = 2^(s qbar_c . q_n - lse_t) takes NCT x 2 MFMAs against the (b,h)'s qbar rows (wave-
// private LDS image) and the correction rows (u qbar)^T t another NCT x 2, in the D layout of dx itself.  One read-modify-write
// of dq less per step: lara_fin_kernel was 57 us / 201 MB of the 1.1 ms cfg5 step.
template <typename E, bool BWD, int NCT>
__global__ __launch_bounds__(256, BWD ? 2 : 3) void seglin_col_kernel(const SegLinP p) {
  static_assert(NCT == 0 || BWD, "the fused finish belongs to the backward pass");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  // a wave = (b, h, side, group of consecutive segments): the generator operands (64 + 64 registers' worth of loads, the
  // transposed one a 64-way gather) are fetched once per wave, not once per 84-token segment
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= p.B * p.H * p.cgroups * 2) return;
  const int side = unit & 1, rest = unit >> 1;
  const int grp = rest % p.cgroups, bh = rest / p.cgroups, b = bh / p.H, h = bh - b * p.H;
  const int l0 = grp * p.cseg_per_group, l1 = min(p.L, l0 + p.cseg_per_group);
  if (l0 >= l1) return;
  const char* src = side ? p.k + (b * p.k_sb + h * p.k_sh) * 2 : p.q + (b * p.q_sb + h * p.q_sh) * 2;
  const int sn = (int)(side ? p.k_sn : p.q_sn);
  const float* G = side ? p.Gk : p.Gq;
  // wave-private LDS: the LayerNorm weight / bias (needed once per segment -- 32 registers that pay for two more steps of rows
  // in flight) and, in the backward, the segment's incoming gradient row on its way from "lane = channel" to the D layout
  __shared__ float wl[4][3][64];
  float* const wlw = wl[wave][0];
  float* const wlb = wl[wave][1];
  float* const wdb = wl[wave][2];
  wlw[lane] = (side ? p.lnk_w : p.lnq_w)[lane];
  wlb[lane] = (side ? p.lnk_b : p.lnq_b)[lane];
  // fused finish: the (b,h)'s (u qbar) and qbar rows as element-type images in this wave's LDS region, lse_t in log2 units
  constexpr int Cp = NCT * 16, FINB = 2 * Cp * 128 + Cp * 4;
  char *R1 = nullptr, *R2 = nullptr;
  float* SC1 = nullptr;
  LaneOff2<64> lo;
  bool fin = false;
  if constexpr (NCT > 0) {
    fin = side == 0;
    if (fin) {
      extern __shared__ __attribute__((aligned(16))) char smem[];
      R1 = smem + wave * FINB;
      R2 = R1 + Cp * 128;
      SC1 = reinterpret_cast<float*>(R2 + Cp * 128);
      lo.init(lane);
      const size_t lm = (size_t)bh * p.C;
      constexpr int SL = Cp * 8 / 64;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* srcm = (j ? p.fin_qbar : p.fin_uq) + lm * 64;
        char* dstm = j ? R2 : R1;
        f32x4 rb[SL][2];
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
          const int idx = lane + sl * 64, row = idx >> 3, c = idx & 7;
          rb[sl][0] = rb[sl][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (row < p.C) {
            rb[sl][0] = *reinterpret_cast<const f32x4*>(srcm + (size_t)row * 64 + c * 8);
            rb[sl][1] = *reinterpret_cast<const f32x4*>(srcm + (size_t)row * 64 + c * 8 + 4);
          }
        }
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) {
          const int idx = lane + sl * 64, row = idx >> 3, c = idx & 7;
          const float f[8] = {rb[sl][0][0], rb[sl][0][1], rb[sl][0][2], rb[sl][0][3], rb[sl][1][0], rb[sl][1][1], rb[sl][1][2], rb[sl][1][3]};
          sts16(dstm + lds_off2<64>(row, c), pack8<E>(f));
        }
      }
      if (lane < Cp) SC1[lane] = lane < p.C ? p.fin_lse[lm + lane] * LOG2E : INFINITY;     // (rows beyond C: t = 0)
    }
  }
  GenA<E> ga;
  ga.load(G, g, li);
  // per-lane channel constants: channel 16 mt + 4 g + r
  f32x4 gb[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) gb[mt] = *reinterpret_cast<const f32x4*>((side ? p.gkb : p.gqb) + 16 * mt + 4 * g);
  // backward state
  typename E::x8 gt[4][2];             // G^T as the A operand of dx^T = G^T dz^T: lane (g, li): in 16 mi + li, k-slots (out)
  f32x4 as1[4];                        // a - mean(a) of the segment, a = dy ln_w (the LayerNorm backward's row constants)
  float ndb = 0.f;
  char* dst = nullptr;
  int dsn = 0;
  const float* dbar_bh = nullptr;
  if constexpr (BWD) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int out = 16 * (2 * kk + (j >> 2)) + 4 * g + (j & 3);      // k-slot j <-> D value r = j & 3 of out-tile 2 kk + (j >> 2)
          f[j] = G[out * 64 + 16 * mi + li];
        }
        gt[mi][kk] = as_x8<E>(pack8<E>(f));
      }
    dst = side ? p.dk + (b * p.dk_sb + h * p.dk_sh) * 2 : p.dq + (b * p.dq_sb + h * p.dq_sh) * 2;
    dsn = (int)(side ? p.dk_sn : p.dq_sn);
    dbar_bh = (side ? p.d_kbar : p.d_qbar) + (size_t)bh * p.L * 64;
    ndb = dbar_bh[(size_t)l0 * 64 + lane];         // lane = channel; the NEXT segment's row is requested a segment ahead
  }
  // rows of the next DEPTH steps in flight (a wave is one of 8 - 12 per CU), in the backward also the gradient-row pieces
  // (channels 16 g .. + 15) those steps add to.  Slot d of the queue is refilled by the step that consumes it: the step loop
  // is unrolled DEPTH times, no register of an outstanding load is ever moved.
  constexpr int DEPTH = BWD ? 3 : 4;
  u32x4 qx[DEPTH][2], qg[BWD ? DEPTH : 1][2];
  SegCur<16> pc, cc;
  pc.set(p, l0);
  cc.set(p, l0);
  auto issue = [&](u32x4 (&x)[2], u32x4 (&og)[2]) {
    const int tok_ = pc.tok(li);
    x[0] = ldg16(src + (tok_ * sn + 8 * g) * 2);
    x[1] = ldg16(src + (tok_ * sn + 32 + 8 * g) * 2);
    if constexpr (BWD) {
      og[0] = ldg16(dst + ((size_t)tok_ * dsn + 16 * g) * 2);
      og[1] = ldg16(dst + ((size_t)tok_ * dsn + 16 * g + 8) * 2);
    }
    pc.advance(p, l1);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(qx[d], qg[BWD ? d : 0]);
  f32x4 accz[4];                       // forward: sum of the normalised rows of the segment (weight / bias applied at its end)
  float inv_len = 0.f;
  bool done = false;
  while (!done) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (done) break;
      if (cc.it == 0) {                // ---- a segment begins ----
        inv_len = 1.f / (float)cc.len;
        if constexpr (BWD) {
          wdb[lane] = ndb;
          ndb = dbar_bh[(size_t)min(cc.l + 1, l1 - 1) * 64 + lane];
          f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wdb + 16 * mt + 4 * g);
            const f32x4 w = *reinterpret_cast<const f32x4*>(wlw + 16 * mt + 4 * g);
            as1[mt] = v * w * inv_len;
            s4 += as1[mt];
          }
          const float s1 = quad_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.f / 64);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) as1[mt] -= s1;
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) accz[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      const typename E::x8 bx0 = as_x8<E>(qx[d][0]), bx1 = as_x8<E>(qx[d][1]);
      const u32x4 oldg[2] = {qg[BWD ? d : 0][0], qg[BWD ? d : 0][1]};
      const int off = cc.it * 16 + li;
      const bool valid = off < cc.len;
      const int tok = cc.s0 + min(off, cc.len - 1);
      const bool seg_end = cc.it + 1 == cc.steps;
      issue(qx[d], qg[BWD ? d : 0]);
      f32x4 z[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        z[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        z[mt] = E::mma(ga.a[mt][0], bx0, z[mt]);
        z[mt] = E::mma(ga.a[mt][1], bx1, z[mt]);
      }
      // LayerNorm over the token's 64 channels = 16 in-lane values x the 4 lanes of its quad: whole-vector arithmetic (two
      // v_pk_*_f32 per f32x4) -- the scalar form of this stage was ~2/3 of the pass's VALU issue
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) { z[mt] += gb[mt]; s4 += z[mt]; }
      const float mean = quad_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.f / 64);
      f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) { z[mt] -= mean; v4 += z[mt] * z[mt]; }
      const float rstd = rsqrtf(quad_sum((v4[0] + v4[1]) + (v4[2] + v4[3])) * (1.f / 64) + 1e-5f);
      if constexpr (!BWD) {
        const float w = valid ? rstd : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) accz[mt] += z[mt] * w;
      } else {
        f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { z[mt] *= rstd; t4 += as1[mt] * z[mt]; }    // z = xhat from here on
        // mean(a xhat) = mean((a - mean a) xhat) since the normalised row sums to zero
        const float s2 = quad_sum((t4[0] + t4[1]) + (t4[2] + t4[3])) * (1.f / 64);
        // the token's LayerNorm statistics for the dG pass (which would otherwise redo them with 16-lane reductions)
        if (g == 0 && valid) p.stats[((size_t)(bh * 2 + side) * p.N + tok)] = f32x4{mean, rstd, s2, 0.f};
        const float w = valid ? rstd : 0.f;
        f32x4 dz[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dz[mt] = (as1[mt] - z[mt] * s2) * w;
        // dx^T[in][token] = G^T dz^T: the D values of out-tiles (2 kk, 2 kk + 1) are the k-slots of step kk
        f32x4 dx[4];
        const typename E::x8 bz0 = as_x8<E>(u32x4{pack2<E>(dz[0][0], dz[0][1]), pack2<E>(dz[0][2], dz[0][3]),
                                                 pack2<E>(dz[1][0], dz[1][1]), pack2<E>(dz[1][2], dz[1][3])});
        const typename E::x8 bz1 = as_x8<E>(u32x4{pack2<E>(dz[2][0], dz[2][1]), pack2<E>(dz[2][2], dz[2][3]),
                                                 pack2<E>(dz[3][0], dz[3][1]), pack2<E>(dz[3][2], dz[3][3])});
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          dx[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
          dx[mi] = E::mma(gt[mi][0], bz0, dx[mi]);
          dx[mi] = E::mma(gt[mi][1], bz1, dx[mi]);
        }
        if constexpr (NCT > 0) {
          if (fin) {                     // (uniform) dq_n -= s sum_c t[c, n] (u qbar)_c
            f32x4 acc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NCT / 2; ++kk) {          // two landmark tiles = the 32 k-slots of one product
              u32x4 p1;
#pragma unroll
              for (int c2 = 0; c2 < 2; ++c2) {
                const int ct = 2 * kk + c2;
                f32x4 tt = {0.f, 0.f, 0.f, 0.f};
                const int row = ct * 16 + li;
                tt = E::mma(as_x8<E>(lds16(R2 + lds_off2<64>(row, g))), bx0, tt);          // channels 8 g ..: chunk g
                tt = E::mma(as_x8<E>(lds16(R2 + lds_off2<64>(row, 4 + g))), bx1, tt);      // channels 32 + 8 g ..: chunk 4 + g
                const f32x4 ls = *reinterpret_cast<const f32x4*>(SC1 + ct * 16 + 4 * g);
                float w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = fast_exp2(tt[r] * p.fin_scale_log2 - ls[r]);
                p1[2 * c2] = pack2<E>(w[0], w[1]);
                p1[2 * c2 + 1] = pack2<E>(w[2], w[3]);
              }
#pragma unroll
              for (int dt = 0; dt < 4; ++dt) {
                const char* r1 = R1 + (32 * kk) * 128 + lo.tr[dt];
                acc[dt] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * 128)), as_x8<E>(p1), acc[dt]);
              }
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) dx[mi][r] -= acc[mi][r] * p.fin_scale;
          }
        }
        // accumulate into the gradient rows.  The four lanes of a token trade their 4-channel pieces (quad_transpose) so that
        // a lane owns 16 contiguous channels: two 16-byte accesses, the row's 128-byte line complete per instruction pair
        // (8-byte pieces of four different lines per instruction ran this pass at 2 TB/s)
        float f[16];
        quad_transpose_f32(dx, f);
        if (valid) {
          float o[16];
          unpack8<E>(oldg[0], o); unpack8<E>(oldg[1], o + 8);
#pragma unroll
          for (int i = 0; i < 16; ++i) o[i] += f[i];
          stg16(dst + ((size_t)tok * dsn + 16 * g) * 2, pack8<E>(o));
          stg16(dst + ((size_t)tok * dsn + 16 * g + 8) * 2, pack8<E>(o + 8));
        }
      }
      if (seg_end) {                   // ---- the segment ends ----
        if constexpr (!BWD) {
          float* out = (side ? p.kbar : p.qbar) + ((size_t)bh * p.L + cc.l) * 64;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wlw + 16 * mt + 4 * g);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(wlb + 16 * mt + 4 * g);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float a = accz[mt][r]; o[r] = w[r] * (group_sum<16>(a) * inv_len) + bb[r]; }
            if (li == 0) *reinterpret_cast<f32x4*>(out + 16 * mt + 4 * g) = o;
          }
        }
        if (cc.l + 1 < l1) cc.set(p, cc.l + 1);
        else done = true;
      } else {
        ++cc.it;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// token-row pass: dG partials.  z[token = 4 g + r][out = 16 nt + li]
template <typename E>
__global__ __launch_bounds__(256, 2) void seglin_dg_kernel(const SegLinP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  // a unit = (b, h, group of segments, side, half of the out-tiles): two waves share a (group, side).  Round 6: a wave
  // evaluates ONLY its two out-tiles (z, dz, the channel sums of d ln_w / d ln_b / d g_b and the generator operand of those
  // tiles: half the registers and half the z products of the version whose waves both evaluated all four) -- 292 registers
  // had meant one wave per SIMD and the 2048 waves of cfg5 in two rounds.
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= p.B * p.H * p.groups * 4) return;
  const int half = unit & 1, side = (unit >> 1) & 1, rest = unit >> 2;
  const int grp = rest % p.groups, bh = rest / p.groups, b = bh / p.H, h = bh - b * p.H;
  const char* src = side ? p.k + (b * p.k_sb + h * p.k_sh) * 2 : p.q + (b * p.q_sb + h * p.q_sh) * 2;
  const int sn = (int)(side ? p.k_sn : p.q_sn);
  const float* G = side ? p.Gk : p.Gq;
  const int ch0 = 32 * half;                             // first out-channel of this wave
  // B operand of z = X G^T for out-tile j of this wave: lane (g, li): G[ch0 + 16 j + li][32 ks + 8 g ..]
  typename E::x8 gbop[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* s_ = G + (ch0 + 16 * j + li) * 64 + 32 * ks + 8 * g;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(s_), hi = *reinterpret_cast<const f32x4*>(s_ + 4);
      const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      gbop[j][ks] = as_x8<E>(pack8<E>(f));
    }
  // 0 / 1 pattern that brings X into the D layout: XD[token][in = 16 it + li] = sum_ch X[token][ch] I[ch][16 it + li];
  // only k-step ks = it >> 1 has a non-zero: slot j of lane (g, li) is channel 32 ks + 8 g + j
  typename E::x8 iop[4];
  {
    const uint32_t one = (uint32_t)E::from_f(1.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int j = 16 * (it & 1) + li - 8 * g;                       // the slot that holds channel 16 it + li, if any
      u32x4 w = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = (j == 2 * q ? one : 0u) | (j == 2 * q + 1 ? one << 16 : 0u);
      iop[it] = as_x8<E>(w);
    }
  }
  float gb[2], lw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    gb[j] = (side ? p.gkb : p.gqb)[ch0 + 16 * j + li];
    lw[j] = (side ? p.lnk_w : p.lnq_w)[ch0 + 16 * j + li];
  }
  __shared__ f32x4 stat_lds[4][32];
  __shared__ float row_lds[4][2][64];
  f32x4* stw = stat_lds[wave];
  float* const wdb = row_lds[wave][0];                   // the segment's incoming gradient row (lane = channel -> D layout)
  float* const wlw = row_lds[wave][1];                   // LayerNorm weight, all 64 channels: the row's mean of dy lw
  wlw[lane] = (side ? p.lnk_w : p.lnq_w)[lane];
  f32x4 dg[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int it = 0; it < 4; ++it) dg[nt][it] = f32x4{0.f, 0.f, 0.f, 0.f};

  // partial sums of d ln_w, d ln_b, d g_b over the group's tokens: lane (g, li) owns channel ch0 + 16 j + li of rows 4 g + r
  float p_dw[2] = {0.f, 0.f}, p_db[2] = {0.f, 0.f}, p_dgb[2] = {0.f, 0.f};
  const int l0 = grp * p.seg_per_group, l1 = min(p.L, l0 + p.seg_per_group);
  if (l0 < l1) {
    // one stream of 32-token steps over the group's segments, the rows and statistics of the next DEPTH steps in flight across
    // the segment boundaries (SegCur, above); the NEXT segment's incoming gradient row is requested a segment ahead (lane =
    // channel) and reaches the D layout through the wave's LDS slot
    constexpr int DEPTH = 2;
    const float* dbar_bh = (side ? p.d_kbar : p.d_qbar) + (size_t)bh * p.L * 64;
    const f32x4* stb = p.stats + (size_t)(bh * 2 + side) * p.N;
    float ndb = dbar_bh[(size_t)l0 * 64 + lane];
    u32x4 qx[DEPTH][2][2];
    f32x4 qs[DEPTH];                                  // statistics of token (step base + (lane & 31)): one coalesced load per step
    SegCur<32> pc, cc;
    pc.set(p, l0);
    cc.set(p, l0);
    auto issue = [&](u32x4 (&x)[2][2], f32x4& st) {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const int tok = pc.tok(tl * 16 + li);                          // A-operand row li of tile tl
        x[tl][0] = ldg16(src + (tok * sn + 8 * g) * 2);
        x[tl][1] = ldg16(src + (tok * sn + 32 + 8 * g) * 2);
      }
      st = stb[pc.tok(lane & 31)];
      pc.advance(p, l1);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(qx[d], qs[d]);
    float a_[2] = {0.f, 0.f}, dyv[2] = {0.f, 0.f}, s1 = 0.f;
    bool done = false;
    while (!done) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (done) break;
        if (cc.it == 0) {              // ---- a segment begins ----
          const float inv_len = 1.f / (float)cc.len;
          wdb[lane] = ndb;
          ndb = dbar_bh[(size_t)min(cc.l + 1, l1 - 1) * 64 + lane];
          float s = 0.f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) s += wdb[16 * nt + li] * inv_len * wlw[16 * nt + li];
          s1 = group_sum<16>(s) * (1.f / 64);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            dyv[j] = wdb[ch0 + 16 * j + li] * inv_len;
            a_[j] = dyv[j] * lw[j];
            p_db[j] += dyv[j] * (float)cc.len;
          }
        }
        u32x4 pz[2], px[2][2];                                          // packed pieces of the two 16-token tiles
        const u32x4 cur[2][2] = {{qx[d][0][0], qx[d][0][1]}, {qx[d][1][0], qx[d][1][1]}};
        // the step's 32 statistics go through this wave's LDS slot: a lane needs those of its four rows 4 g + r of either tile
        // (fetching them from global memory where they are needed was a dependent round trip per tile: 110 -> 161 us)
        if (lane < 32) stw[lane] = qs[d];
        const int sbase = cc.it * 32, slen = cc.len;
        const bool seg_end = cc.it + 1 == cc.steps;
        issue(qx[d], qs[d]);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const int base = sbase + tl * 16;
          const typename E::x8 ax0 = as_x8<E>(cur[tl][0]);
          const typename E::x8 ax1 = as_x8<E>(cur[tl][1]);
          f32x4 z[2], xd[4];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            z[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            z[j] = E::mma(ax0, gbop[j][0], z[j]);
            z[j] = E::mma(ax1, gbop[j][1], z[j]);
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) xd[nt] = E::mma((nt >> 1) ? ax1 : ax0, iop[nt], f32x4{0.f, 0.f, 0.f, 0.f});
          // rows r = tokens base + 4 g + r; their LayerNorm statistics (mean, rstd, mean(a xhat)) come from the dq / dk pass
          float dzr[2][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 st = stw[tl * 16 + 4 * g + r];
            const bool live = base + 4 * g + r < slen;
            const float w = live ? st[1] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float xh = (z[j][r] + gb[j] - st[0]) * st[1];
              dzr[j][r] = w * (a_[j] - s1 - xh * st[2]);
              if (live) p_dw[j] += dyv[j] * xh;
              p_dgb[j] += dzr[j][r];
            }
          }
          // this tile's four tokens of the lane = four k-slots: out-tile j of dz, in-tile nt of X
          pz[tl] = u32x4{pack2<E>(dzr[0][0], dzr[0][1]), pack2<E>(dzr[0][2], dzr[0][3]), pack2<E>(dzr[1][0], dzr[1][1]), pack2<E>(dzr[1][2], dzr[1][3])};
          px[tl][0] = u32x4{pack2<E>(xd[0][0], xd[0][1]), pack2<E>(xd[0][2], xd[0][3]), pack2<E>(xd[1][0], xd[1][1]), pack2<E>(xd[1][2], xd[1][3])};
          px[tl][1] = u32x4{pack2<E>(xd[2][0], xd[2][1]), pack2<E>(xd[2][2], xd[2][3]), pack2<E>(xd[3][0], xd[3][1]), pack2<E>(xd[3][2], xd[3][3])};
        }
        // dG[out ch0 + 16 nt + ..][in 16 it + ..] += sum over the 32 tokens: k-slots (tile 0 rows 4 g + r, tile 1 rows 4 g + r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int qn = nt * 2;
          const typename E::x8 aop = as_x8<E>(u32x4{pz[0][qn], pz[0][qn + 1], pz[1][qn], pz[1][qn + 1]});
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int hi_ = it >> 1, qi = (it & 1) * 2;
            const typename E::x8 bop = as_x8<E>(u32x4{px[0][hi_][qi], px[0][hi_][qi + 1], px[1][hi_][qi], px[1][hi_][qi + 1]});
            dg[nt][it] = E::mma(aop, bop, dg[nt][it]);
          }
        }
        if (seg_end) {
          if (cc.l + 1 < l1) cc.set(p, cc.l + 1);
          else done = true;
        } else {
          ++cc.it;
        }
      }
    }
  }
  {
    // [group][side][4][64]: (d ln_w, d ln_b, d g_b, 0); the four lane rows g hold different tokens of the same channel
    float* pr = p.part + (((size_t)bh * p.groups + grp) * 2 + side) * 4 * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float v0 = quad_sum(p_dw[j]), v2 = quad_sum(p_dgb[j]);
      if (g == 0) {
        pr[ch0 + 16 * j + li] = v0;
        pr[64 + ch0 + 16 * j + li] = p_db[j];
        pr[128 + ch0 + 16 * j + li] = v2;
        pr[192 + ch0 + 16 * j + li] = 0.f;
      }
    }
  }
  float* out = p.dG_part + (((size_t)bh * p.groups + grp) * 2 + side) * 64 * 64;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * (2 * half + nt) + 4 * g + r) * 64 + 16 * it + li] = dg[nt][it][r];
}

int seglin_groups(int BH, int L) {
  // ~2048 waves for the dG pass (two per (group, side)), every wave at least one segment
  int per = (BH * 4 * L + 2047) / 2048;
  if (per < 1) per = 1;
  return (L + per - 1) / per;
}

int seglin_dispatch(int which, const SegLinP& p0, int dtype, hipStream_t st) {
  SegLinP p = p0;
  p.groups = seglin_groups(p.B * p.H, p.L);
  p.seg_per_group = (p.L + p.groups - 1) / p.groups;
  {
    // forward / dq-dk passes: ONE resident round of waves (12 per CU at 166 registers, 8 at 222) -- 3328 waves on 3072 / 2048
    // slots ran as two rounds, the second nearly empty -- every wave at least one segment
    const long cap = (long)ea_device_cus() * (which == 0 ? 12 : 8);      // (which: 0 forward, 1 dq / dk pass, 3 the same with the fused finish, 2 dG)
    int per = (int)(((long)p.B * p.H * 2 * p.L + cap - 1) / cap);
    if (per < 1) per = 1;
    p.cgroups = (p.L + per - 1) / per;
    p.cseg_per_group = (p.L + p.cgroups - 1) / p.cgroups;
  }
  const long units = (long)p.B * p.H * p.cgroups * 2;
  const dim3 grid((unsigned)((units + 3) / 4)), block(256);
  const long gunits = (long)p.B * p.H * p.groups * 4;
  const dim3 ggrid((unsigned)((gunits + 3) / 4));
#define EA_SL(E_)                                                                                                  \
  do {                                                                                                             \
    if (which == 0) hipLaunchKernelGGL((seglin_col_kernel<E_, false, 0>), grid, block, 0, st, p);                 \
    else if (which == 1) hipLaunchKernelGGL((seglin_col_kernel<E_, true, 0>), grid, block, 0, st, p);             \
    else if (which == 3 && p.C <= 32) EA_SLF(E_, 2);                                                               \
    else if (which == 3) EA_SLF(E_, 4);                                                                            \
    else hipLaunchKernelGGL((seglin_dg_kernel<E_>), ggrid, block, 0, st, p);                                      \
  } while (0)
#define EA_SLF(E_, NCT_)                                                                                           \
  do {                                                                                                             \
    constexpr int lds_ = 4 * (2 * NCT_ * 16 * 128 + NCT_ * 16 * 4);                                                \
    if (lds_ > 48 * 1024) {                                                                                        \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&seglin_col_kernel<E_, true, NCT_>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                      \
      if (e_ != hipSuccess) return (int)e_;                                                                        \
    }                                                                                                              \
    hipLaunchKernelGGL((seglin_col_kernel<E_, true, NCT_>), grid, block, lds_, st, p);                             \
  } while (0)
  if (which == 3 && (p.C < 1 || p.C > 64 || !p.fin_qbar || !p.fin_uq || !p.fin_lse)) return EA_E_BADARG;
  if (dtype == EA_BF16) EA_SL(BF16);
  else if (dtype == EA_F16) EA_SL(F16);
  else return EA_E_UNSUPPORTED;
#undef EA_SL
#undef EA_SLF
  return (int)hipGetLastError();
}

}  // namespace ea
