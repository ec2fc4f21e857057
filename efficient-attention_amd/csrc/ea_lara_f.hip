// ea_lara_f.hip -- fused LARA backward passes (lara.py:201-246 differentiated): each token tensor is
// read ONCE per side.
//
//   lara_fq_kernel  (q, dout -> dq, per-landmark sums)   replaces  LX_BWDQ + LY_BWDQ
//   lara_fk_kernel  (k, v    -> dk, dv, d omega partial)  replaces  LX_BWDK + LY_BWDK
//   lara_fin_kernel (q, dq, dk -> dq, dk)                 replaces  LX_QCORR + the pooling backward
//
// The estimator couples token n and landmark sample c through [C x N] matrices (W, dZ, t dt, t / dB).
// Their per-token contractions (over c: dq, dk, dv) want the token-column register layout
// D[c = 4g+r][n = li] -- reductions over c are in-lane + one 4-lane step -- and their per-landmark
// contractions (over n: d kv_stats, d omega, d q_bar, the scalar sums) want the token-row layout.
// Round 1 evaluated the elementwise stage twice, once per layout, in two passes over q/dout (k/v).
// Here it is evaluated once, in the token-column layout; every wave then drops its 16-token slab of
// the weight matrices (rounded to the MFMA element type, exactly as the token-row pass rounded them)
// and its token rows into LDS, and after one barrier wave w contracts landmark tile w over the 64
// tokens of the chunk: the B operand is a ds_read_b64_tr_b16 of the [n][c] weight slab, the A operand
// a ds_read_b64_tr_b16 of the [n][d] token rows.  The second barrier of a chunk sits right before
// the next slab is written, so the next chunk's score MFMAs and elementwise stage overlap the
// slower waves' contraction.  Per-(b,h) partial results leave in the layout the token-row pass
// used (ea_lara_merge_bwd is unchanged).
//
// The transposed landmark matrices of the contraction over c are ds_read_b64_tr_b16 reads of the
// row-major staging (no second, transposed copy in LDS): 73 KB per workgroup at C <= 64, d = 64,
// two workgroups per CU.
#include <stdlib.h>
#include "ea_lara.h"

namespace ea {

// byte offset of element (row, c) of a bf16/fp16 LDS tile [rows][W] whose 16-byte chunks are
// XOR-swizzled by the row (same scheme as the token tiles, lds_off<>)
template <int W> EA_DEV int wt_off(int row, int c) { return lds_off<W>(row, c >> 3) + ((c & 7) << 1); }

// all global loads of up to three [C][D] fp32 landmark matrices in flight, then convert + store as
// swizzled element-type rows (zero rows beyond C)
template <typename E, int D, int Cp>
EA_DEV void stage_rows3(char* const dst[3], const float* const src[3], int C, int tid) {
  constexpr int CPRs = D / 8;
  constexpr int SL = (Cp * CPRs + 255) / 256;
  float4 rb[3][SL][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int idx = tid + sl * 256;
      const int row = idx / CPRs, c = idx - row * CPRs;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (src[j] && idx < Cp * CPRs && row < C) {
        lo = *reinterpret_cast<const float4*>(src[j] + (size_t)row * D + c * 8);
        hi = *reinterpret_cast<const float4*>(src[j] + (size_t)row * D + c * 8 + 4);
      }
      rb[j][sl][0] = lo; rb[j][sl][1] = hi;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int idx = tid + sl * 256;
      const int row = idx / CPRs, c = idx - row * CPRs;
      if (!src[j] || idx >= Cp * CPRs) continue;
      const float4 lo = rb[j][sl][0], hi = rb[j][sl][1];
      const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      sts16(dst[j] + TileL<D>::off(row, c), pack8<E>(f));
    }
  }
}

// Reduce-scatter over the 16 lanes of a DPP row: every lane brings 16 values, lane li leaves with the row's sum of value li.
// Four halving steps (partners li ^ 8, half-row mirror, quad reverse, li ^ 1: each flips the bit that decides which half of the
// remaining values a lane keeps), 15 adds + 30 selects -- against 16 running sums held in registers for the whole launch.
EA_DEV float row_reduce_scatter16(const float (&v)[16], int li) {
  float w8[8], w4[4], w2[2];
  const bool h8 = (li & 8) != 0, h4 = (li & 4) != 0, h2 = (li & 2) != 0, h1 = (li & 1) != 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) w8[j] = (h8 ? v[8 + j] : v[j]) + dpp_mov<0x128>(h8 ? v[j] : v[8 + j]);        // row_ror:8
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = (h4 ? w8[4 + j] : w8[j]) + dpp_mov<0x141>(h4 ? w8[j] : w8[4 + j]);    // row_half_mirror
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = (h2 ? w4[2 + j] : w4[j]) + dpp_mov<0x1B>(h2 ? w4[j] : w4[2 + j]);     // quad_perm [3,2,1,0]
  return (h1 ? w2[1] : w2[0]) + dpp_mov<0xB1>(h1 ? w2[0] : w2[1]);                                          // quad_perm [1,0,3,2]
}

template <typename E> EA_DEV typename E::x8 ones_x8() {
  const uint32_t o = (uint32_t)E::from_f(1.f) * 0x00010001u;
  return as_x8<E>(u32x4{o, o, o, o});
}

// ------------------------------------------------------------------------------------------
// query side
// ------------------------------------------------------------------------------------------
// LH: r-pairs of the LAST landmark tile that can hold a real sample (C = 49: tile 3 has the single row 48, so only the
// first (r = 0, 1) pair of a lane is ever non-zero -> the elementwise stage skips the other: 7 of 8 pairs per lane).
// The per-token statistics of the estimator's softmax over the samples -- lse_Z (log2 units) and mean_c t -- come from
// the forward (ea_lara_out_fwd writes them: 8 bytes per token-head), so W = alpha 2^(z - lse_Z) directly: no max, no
// sum, no reciprocal, and two cross-lane reductions per tile instead of five.
// MIS: the estimator variant is a template parameter -- as a runtime value every `if (opt)` of the unrolled stage became
// a branch, and the chunk loop fell apart into ~50 basic blocks of 10-25 instructions (no scheduling across them).
template <typename E, int D, int NCT, int LH, int MIS>
__global__ __launch_bounds__(256, 2) void lara_fq_kernel(const LaraP p) {
  constexpr int ROWB = D * 2, KS = D / 32, DT = D / 16, DQ = D / 4;
  constexpr int Cp = NCT * 16, ROWW = Cp * 2, NSUB = 4 / NCT;
  constexpr int WSW = (Cp / 8) >= 8 ? 7 : (Cp / 8) - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;                                  // omega rows
  char* R2 = R1 + Cp * ROWB;                        // qbar rows
  char* R3 = R2 + Cp * ROWB;                        // kv rows
  char* TQ = R3 + Cp * ROWB;                        // q rows of the chunk   [64][D]
  char* TD = TQ + 64 * ROWB;                        // dout rows of the chunk
  char* WT = TD + 64 * ROWB;                        // weight slabs [4][64][Cp]: W, dZ, t dt, t
  float* SC0 = reinterpret_cast<float*>(WT + 4 * 64 * ROWW);
  float* SC1 = SC0 + Cp;
  float* SC2 = SC1 + Cp;
  float* DB = SC2 + Cp;                             // [4][Cp] per-wave sums of d alpha

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nbh = p.B * p.H;
  const int blk = blockIdx.x / nbh, bh = blockIdx.x - blk * nbh;      // slice-major (see lara_f_plan)
  const int b = bh / p.H, h = bh - b * p.H;
  const size_t lm = (size_t)bh * p.C;
  constexpr bool opt = MIS == MIS_OPT;
  constexpr bool use_t = MIS != MIS_BH;
  const char* qb = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* dob = p.dout.p + (b * p.dout.sb + h * p.dout.sh) * 2;
  const float invC = 1.f / (float)p.C;
  const int n0 = p.nsplit == 2 ? p.tok_begin[blk] : blk * p.tok_per_block;
  const int n1 = p.nsplit == 2 ? p.tok_begin[blk + 1] : min(p.N, n0 + p.tok_per_block);
  const int last_tok = n1 - 1;

  EA_BLK(p, 0);
  EA_STAMP(p, 0);
  u32x4 nx1[KS], nx2[KS];
  float nlz, ntm;
  const float* lzb = p.lseZ + (size_t)bh * p.N;
  const float* tmb = p.tmean + (size_t)bh * p.N;
  // (row offsets inside a (b,h) in 32 bits, as on the key side: the dispatcher checks N * stride < 2^30 elements)
  const unsigned qsn = (unsigned)p.q.sn, dosn = (unsigned)p.dout.sn;
  auto issue = [&](int cb_) {
    const unsigned tok_ = (unsigned)min(cb_ + wave * 16 + li, last_tok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned eo = (g * KS + ks) * 8;
      nx1[ks] = ldg16(qb + (size_t)((tok_ * qsn + eo) * 2u));
      nx2[ks] = ldg16(dob + (size_t)((tok_ * dosn + eo) * 2u));
    }
    nlz = lzb[tok_];
    ntm = tmb[tok_];
  };
  const bool empty = n0 >= n1;                      // (uniform) an empty second slice (lara_f_plan): zero partials, no staging
  if (!empty) issue(n0);
  float sc_v0 = -INFINITY, sc_v1 = INFINITY, sc_v2 = 1.f;
  if (tid < p.C && !empty) {
    sc_v0 = p.cst[lm + tid] * LOG2E;
    if (opt) { sc_v2 = p.bhv[lm + tid]; sc_v1 = p.lse_t[lm + tid] * LOG2E; }
  }
  if (!empty) {
    char* const dst[3] = {R1, R2, R3};
    const float* const src[3] = {p.omega + lm * D, use_t ? p.qbar + lm * D : nullptr, p.kv + lm * D};
    stage_rows3<E, D, Cp>(dst, src, p.C, tid);
  }
  if (tid < Cp) { SC0[tid] = sc_v0; SC1[tid] = sc_v1; SC2[tid] = sc_v2; }
  typename LaneOffSel<D>::type lo;
  lo.init(lane);
  // tr-read offset into a weight slab: rows 4g + (li >> 2), column segment 4 (li & 3) of a landmark tile
  const int wr = 4 * g + (li >> 2);
  const int yct = wave % NCT, ysub = wave / NCT;       // token-row phase: landmark tile / 32-token half
  const bool ywave = wave < NCT * NSUB;
  int wtr;
  {
    const int colb = (16 * yct + 4 * (li & 3)) * 2;
    wtr = wr * ROWW + ((((colb >> 4)) ^ (wr & WSW)) << 4) + (colb & 15);
  }
  f32x4 acc0[DT], acc1[DT], acc2[DT], acc3[DT], accR = {0.f, 0.f, 0.f, 0.f}, accU = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc0[dt] = acc1[dt] = acc2[dt] = acc3[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-landmark sums of d alpha (d bh).  Full last tile (LH == 2, four tiles): the 16 running sums per lane are what pushed
  // this instantiation over the 256-register budget (72 B / lane of scratch, an accumulator quad spilled and re-read per
  // chunk) -> one running sum per lane, the chunk's 16 values reduce-scattered over the row's 16 token lanes
  constexpr bool RS = opt && LH == 2 && NCT == 4;
  f32x2 sdbh[RS ? 1 : NCT][2];
  float sdb1 = 0.f;
#pragma unroll
  for (int ct = 0; ct < (RS ? 1 : NCT); ++ct) sdbh[ct][0] = sdbh[ct][1] = f32x2{0.f, 0.f};
  const typename E::x8 ones = ones_x8<E>();
  __syncthreads();
  EA_STAMP(p, 1);
  EA_BLKX(p, 0);
  int prof_it = 0;
  (void)prof_it;

  for (int cb = n0; cb < n1; cb += 64) {
    if (prof_it < 8) EA_STAMP(p, 2 + prof_it * 7);
    const int tok = cb + wave * 16 + li;
    const bool valid = tok < n1;
    typename E::x8 f1[KS], f2[KS];
    u32x4 raw1[KS], raw2[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      raw1[ks] = nx1[ks]; raw2[ks] = nx2[ks];
      f1[ks] = as_x8<E>(nx1[ks]);
      f2[ks] = as_x8<E>(nx2[ks]);
    }
    // tokens beyond the slice contribute nothing to the per-landmark sums: lse_Z = +inf -> W = dZ = d alpha = 0
    const float lz = valid ? nlz : INFINITY;
    const float tmean = ntm;
    issue(cb + 64);
    // ---- score tiles: A = s omega.q, T = s qbar.q, dW = kv.dout  (D[c = 4g+r][n = li]) ----
    f32x4 a[NCT], tt[NCT], dw[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      a[ct] = tt[ct] = dw[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = ct * 16 + li;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[ct] = E::mma(as_x8<E>(lds16(R1 + TileL<D>::off(row, g * KS + ks))), f1[ks], a[ct]);
        if (use_t) tt[ct] = E::mma(as_x8<E>(lds16(R2 + TileL<D>::off(row, g * KS + ks))), f1[ks], tt[ct]);
        dw[ct] = E::mma(as_x8<E>(lds16(R3 + TileL<D>::off(row, g * KS + ks))), f2[ks], dw[ct]);
      }
    }
    if (prof_it < 8) EA_STAMP(p, 3 + prof_it * 7);
    // ---- elementwise stage (lara.py:221-243 differentiated; alpha as a factor, d alpha = 2^z (dW - rd) without a
    // division, lse_Z / mean t from the forward) ----
    const float s2 = p.scale_log2;
    const f32x2 s22 = {s2, s2};
    const f32x2 kap = {p.kappa, p.kappa};
    const f32x2 lz2 = {lz, lz};
    const float kt = -p.kappa * tmean;
    // pair (ct, hh) is live unless it lies in the never-populated part of the last tile
#define EA_LIVE(ct, hh) ((ct) < NCT - 1 || (hh) < LH)
    f32x2 tv[NCT][2], ez[NCT][2], wv[NCT][2];
    f32x2 rd2 = {0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 cs = *reinterpret_cast<const float4*>(SC0 + ct * 16 + 4 * g);
      const float4 ls = *reinterpret_cast<const float4*>(SC1 + ct * 16 + 4 * g);
      const float4 bv = *reinterpret_cast<const float4*>(SC2 + ct * 16 + 4 * g);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (!EA_LIVE(ct, hh)) { tv[ct][hh] = ez[ct][hh] = wv[ct][hh] = f32x2{0.f, 0.f}; continue; }
        const f32x2 av = {a[ct][2 * hh], a[ct][2 * hh + 1]}, tq = {tt[ct][2 * hh], tt[ct][2 * hh + 1]};
        const f32x2 csv = hh ? f32x2{cs.z, cs.w} : f32x2{cs.x, cs.y};
        f32x2 z = av * s22 + (csv - lz2);
        if (MIS == MIS_BIASED) z += tq * s22;
        const f32x2 wz = {fast_exp2(z[0]), fast_exp2(z[1])};                  // 2^(z - lse_Z)
        if (opt) {
          const f32x2 lsv = hh ? f32x2{ls.z, ls.w} : f32x2{ls.x, ls.y};
          const f32x2 bvv = hh ? f32x2{bv.z, bv.w} : f32x2{bv.x, bv.y};
          const f32x2 x = tq * s22 - lsv;
          const f32x2 t = {fast_exp2(x[0]), fast_exp2(x[1])};
          const f32x2 al = kap * t + (bvv + f32x2{kt, kt});
          tv[ct][hh] = t;
          wv[ct][hh] = wz * f32x2{fmaxf(al[0], 1e-8f), fmaxf(al[1], 1e-8f)};   // W
          ez[ct][hh] = f32x2{al[0] > 1e-8f ? wz[0] : 0.f, al[1] > 1e-8f ? wz[1] : 0.f};   // W / alpha where the clamp is off
        } else {
          tv[ct][hh] = f32x2{0.f, 0.f};
          wv[ct][hh] = wz;
          ez[ct][hh] = wz;
        }
        rd2 += wv[ct][hh] * f32x2{dw[ct][2 * hh], dw[ct][2 * hh + 1]};
      }
    }
    const float rd = quad_sum(rd2[0] + rd2[1]);                            // = dout_n . out_n
    const f32x2 rdv = {rd, rd};
    f32x2 sda2 = {0.f, 0.f};
    f32x2 da[NCT][2];
    // W and dZ are rounded to the element type as soon as they exist (the contraction over c and the
    // slabs both take them in that form): half the registers from here on
    u32x2 Wp[NCT], dZp[NCT], tdp[NCT], tp[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      f32x2 dz[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (!EA_LIVE(ct, hh)) { dz[hh] = da[ct][hh] = f32x2{0.f, 0.f}; continue; }
        const f32x2 dd = f32x2{dw[ct][2 * hh], dw[ct][2 * hh + 1]} - rdv;  // dW - rd
        dz[hh] = wv[ct][hh] * dd;
        if (opt) {
          da[ct][hh] = ez[ct][hh] * dd;                                    // dZ / alpha
          sda2 += da[ct][hh];
          if constexpr (!RS) sdbh[ct][hh] += da[ct][hh];
        }
      }
      Wp[ct] = u32x2{pack2<E>(wv[ct][0][0], wv[ct][0][1]), EA_LIVE(ct, 1) ? pack2<E>(wv[ct][1][0], wv[ct][1][1]) : 0u};
      dZp[ct] = u32x2{pack2<E>(dz[0][0], dz[0][1]), EA_LIVE(ct, 1) ? pack2<E>(dz[1][0], dz[1][1]) : 0u};
    }
    const float sda = quad_sum(sda2[0] + sda2[1]);
    if constexpr (RS) {
      float dflat[16];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) { dflat[4 * ct + 2 * hh] = da[ct][hh][0]; dflat[4 * ct + 2 * hh + 1] = da[ct][hh][1]; }
      sdb1 += row_reduce_scatter16(dflat, li);       // lane li: landmark 16 (li >> 2) + 4 g + (li & 3)
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (opt) {
        const float m = sda * invC;
        const f32x2 t0 = tv[ct][0] * kap * (da[ct][0] - f32x2{m, m});      // t * dt
        tdp[ct][0] = pack2<E>(t0[0], t0[1]);
        tp[ct][0] = pack2<E>(tv[ct][0][0], tv[ct][0][1]);
        if (EA_LIVE(ct, 1)) {
          const f32x2 t1 = tv[ct][1] * kap * (da[ct][1] - f32x2{m, m});
          tdp[ct][1] = pack2<E>(t1[0], t1[1]);
          tp[ct][1] = pack2<E>(tv[ct][1][0], tv[ct][1][1]);
        } else {
          tdp[ct][1] = 0u; tp[ct][1] = 0u;
        }
      } else {
        tdp[ct] = dZp[ct];                                                 // mis-biased: dT = dZ
        tp[ct] = u32x2{0u, 0u};
      }
    }
#undef EA_LIVE
    if (prof_it < 8) EA_STAMP(p, 4 + prof_it * 7);
    // ---- contraction over c: dq^T[d][n] = omega^T . dZ (+ qbar^T . t dt) ----
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NCT / 2; ++kk) {
      const u32x4 p1 = {dZp[2 * kk][0], dZp[2 * kk][1], dZp[2 * kk + 1][0], dZp[2 * kk + 1][1]};
      const u32x4 p2 = {tdp[2 * kk][0], tdp[2 * kk][1], tdp[2 * kk + 1][0], tdp[2 * kk + 1][1]};
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* r1 = R1 + (32 * kk) * ROWB + lo.tr[dt];
        acc[dt] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * ROWB)), as_x8<E>(p1), acc[dt]);
        if (use_t) {
          const char* r2 = R2 + (32 * kk) * ROWB + lo.tr[dt];
          acc[dt] = E::mma(as_x8<E>(E::tr4(r2), E::tr4(r2 + 16 * ROWB)), as_x8<E>(p2), acc[dt]);
        }
      }
    }
    {
      // unconditional stores (rows past the slice go to the trash line): static store count, see ea_trash_line()
      char* dst = valid ? p.dq.p + (b * p.dq.sb + h * p.dq.sh + tok * p.dq.sn + DQ * g) * 2 : ea_trash_line();
      if constexpr (TileL<D>::NEWTR) {
        u32x4 o0, o1;
        quad_transpose_pack<E>(acc, p.scale, o0, o1);      // pieces -> the lane's 32 contiguous bytes
        stg16(dst, o0);
        stg16(dst + 16, o1);
      } else {
        float f[DQ];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) f[4 * dt + r] = acc[dt][r] * p.scale;
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
      }
    }
    if (prof_it < 8) EA_STAMP(p, 5 + prof_it * 7);
    // ---- hand the slab to the token-row phase ----
    __syncthreads();                        // the previous chunk's readers are done
    if (prof_it < 8) EA_STAMP(p, 6 + prof_it * 7);
    {
      const int row = wave * 16 + li;
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sts16(TQ + TileL<D>::off(row, g * KS + ks), valid ? raw1[ks] : z);
        sts16(TD + TileL<D>::off(row, g * KS + ks), valid ? raw2[ks] : z);
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int o = wt_off<Cp>(row, 16 * ct + 4 * g);
        *reinterpret_cast<u32x2*>(WT + o) = Wp[ct];
        *reinterpret_cast<u32x2*>(WT + 64 * ROWW + o) = dZp[ct];
        if (opt) {
          *reinterpret_cast<u32x2*>(WT + 2 * 64 * ROWW + o) = tdp[ct];
          *reinterpret_cast<u32x2*>(WT + 3 * 64 * ROWW + o) = tp[ct];
        }
      }
    }
    __syncthreads();
    if (prof_it < 8) EA_STAMP(p, 7 + prof_it * 7);
    // ---- token-row phase: landmark tile yct over the chunk's tokens ----
    if (ywave) {
#pragma unroll
      for (int kq = 0; kq < 2 / NSUB; ++kq) {
        const int kb = (NSUB == 1 ? kq : ysub) * 32;
        const char* w0 = WT + kb * ROWW + wtr;
        const typename E::x8 b0 = as_x8<E>(E::tr4(w0), E::tr4(w0 + 16 * ROWW));
        const typename E::x8 b1 = as_x8<E>(E::tr4(w0 + 64 * ROWW), E::tr4(w0 + 64 * ROWW + 16 * ROWW));
        typename E::x8 b2 = b1, b3 = b1;
        if (opt) {
          b2 = as_x8<E>(E::tr4(w0 + 2 * 64 * ROWW), E::tr4(w0 + 2 * 64 * ROWW + 16 * ROWW));
          b3 = as_x8<E>(E::tr4(w0 + 3 * 64 * ROWW), E::tr4(w0 + 3 * 64 * ROWW + 16 * ROWW));
        }
        accR = E::mma(ones, b1, accR);
        if (opt) accU = E::mma(ones, b2, accU);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* td = TD + kb * ROWB + lo.tr[dt];
          const char* tq = TQ + kb * ROWB + lo.tr[dt];
          const typename E::x8 ado = as_x8<E>(E::tr4(td), E::tr4(td + 16 * ROWB));
          const typename E::x8 aq = as_x8<E>(E::tr4(tq), E::tr4(tq + 16 * ROWB));
          acc0[dt] = E::mma(ado, b0, acc0[dt]);           // d kv_stats
          acc1[dt] = E::mma(aq, b1, acc1[dt]);            // sum dZ q
          if (opt) {
            acc2[dt] = E::mma(aq, b2, acc2[dt]);          // sum t dt q
            acc3[dt] = E::mma(aq, b3, acc3[dt]);          // sum t q
          }
        }
      }
    }
    if (prof_it < 8) EA_STAMP(p, 8 + prof_it * 7);
    ++prof_it;
  }
  EA_STAMP(p, 60);
  EA_BLKX(p, 1);
  // ---- per-landmark sums of d alpha: over the 16 token lanes, then over the four waves ----
  if constexpr (RS) {
    DB[wave * Cp + 16 * (li >> 2) + 4 * g + (li & 3)] = sdb1;
  } else if (opt) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = sdbh[ct][hh][e];
          v = group_sum<16>(v);
          if (li == 0) DB[wave * Cp + 16 * ct + 4 * g + 2 * hh + e] = v;
        }
  }
  __syncthreads();
  // (epilogue coordinates re-derived from an opaque copy of the thread index, as in lara_fk_kernel)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int wave_e = tid_e >> 6, li_e = tid_e & 15, g_e = (tid_e & 63) >> 4;
  const int yct_e = wave_e % NCT, ysub_e = wave_e / NCT;
  const int c = yct_e * 16 + li_e;
  if (!(wave_e < NCT * NSUB) || c >= p.C) { EA_BLK(p, 1); return; }
  const int S = p.nsplit * NSUB;
  const size_t slot = ((size_t)bh * S + blk * NSUB + ysub_e) * p.C + c;
  if (g_e == 0) {
    float* ml = p.p_ml + slot * 4;
    float dbh = 0.f;
    if (opt && ysub_e == 0) dbh = DB[c] + DB[Cp + c] + DB[2 * Cp + c] + DB[3 * Cp + c];
    ml[0] = accR[0]; ml[1] = dbh; ml[2] = accU[0]; ml[3] = 0.f;
  }
  auto put = [&](float* base, const f32x4* av) {
    float* d = base + slot * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<float4*>(d + acc_chan<D>(dt, g_e)) = make_float4(av[dt][0], av[dt][1], av[dt][2], av[dt][3]);
  };
  put(p.p_acc0, acc0);
  put(p.p_acc1, acc1);
  if (opt) { put(p.p_acc2, acc2); put(p.p_acc3, acc3); }
  EA_BLK(p, 1);
}

// ------------------------------------------------------------------------------------------
// key side
// ------------------------------------------------------------------------------------------
// FOLD (round 5): the per-slice partials of lara_fq_kernel are merged HERE (ea_lara_merge.hip's backward arithmetic in the
// prologue: d kv_stats rows, dkk = dkv . kv, r) instead of by a merge launch between the two passes; block 0 of a (b,h)
// also writes what the landmark backward and the finish pass read (dbh, dlp, sum dZ q, d qbar rows, u qbar).
template <typename E, int D, int NCT, bool FOLD>
__global__ __launch_bounds__(256, 3) void lara_fk_kernel(const LaraP p) {
  constexpr int ROWB = D * 2, KS = D / 32, DT = D / 16, DQ = D / 4;
  constexpr int Cp = NCT * 16, ROWW = Cp * 2, NSUB = 4 / NCT;
  constexpr int WSW = (Cp / 8) >= 8 ? 7 : (Cp / 8) - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;                                  // omega rows
  char* R3 = R1 + Cp * ROWB;                        // d kv_stats rows
  char* TK = R3 + Cp * ROWB;                        // k rows of the chunk [64][D]
  char* WT = TK + 64 * ROWB;                        // dB slab [64][Cp]
  float* SC0 = reinterpret_cast<float*>(WT + 64 * ROWW);
  float* SC1 = SC0 + Cp;
  float* SC2 = SC1 + Cp;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nbh = p.B * p.H;
  const int blk = blockIdx.x / nbh, bh = blockIdx.x - blk * nbh;
  const int b = bh / p.H, h = bh - b * p.H;
  const size_t lm = (size_t)bh * p.C;
  const char* kb_ = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const char* vb_ = p.v.p + (b * p.v.sb + h * p.v.sh) * 2;
  const int n0 = p.nsplit == 2 ? p.tok_begin[blk] : blk * p.tok_per_block;
  const int n1 = p.nsplit == 2 ? p.tok_begin[blk + 1] : min(p.N, n0 + p.tok_per_block);
  const int last_tok = n1 - 1;

  EA_BLK(p, 0);
  u32x4 nx1[KS], nx2[KS];
  // row offsets inside a (b,h) in 32 bits (the dispatcher checks N * stride < 2^30 elements): the 64-bit products kept a
  // register pair alive across the chunk loop that it does not have (spilled and re-read per chunk until round 6)
  const unsigned ksn = (unsigned)p.k.sn, vsn = (unsigned)p.v.sn;
  auto issue = [&](int cb_) {
    const unsigned tok_ = (unsigned)min(cb_ + wave * 16 + li, last_tok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const unsigned eo = (g * KS + ks) * 8;
      nx1[ks] = ldg16(kb_ + (size_t)((tok_ * ksn + eo) * 2u));
      nx2[ks] = ldg16(vb_ + (size_t)((tok_ * vsn + eo) * 2u));
    }
  };
  issue(n0);
  float sc_v0 = INFINITY, sc_v1 = 0.f, sc_v2 = 0.f;
  float mg_dbh = 0.f;
  if (tid < p.C) {
    sc_v0 = p.lse_k[lm + tid] * LOG2E;
    if constexpr (FOLD) {
      const int S = p.m_S;
      float4 m4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) m4[u] = *reinterpret_cast<const float4*>(p.m_ml + (((size_t)bh * S + min(u, S - 1)) * p.C + tid) * 4);
      float r = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < S) { r += m4[u].x; mg_dbh += m4[u].y; }
      }
      sc_v2 = r;
    } else {
      sc_v1 = p.dkk[lm + tid]; sc_v2 = p.rsum[lm + tid];
    }
  }
  if constexpr (FOLD) {
    constexpr int CPRs = D / 8;
    constexpr int SL = (Cp * CPRs + 255) / 256;
    const int S = p.m_S;
    {
      char* const dst[3] = {R1, nullptr, nullptr};
      const float* const src[3] = {p.omega + lm * D, nullptr, nullptr};
      stage_rows3<E, D, Cp>(dst, src, p.C, tid);
    }
    // d kv_stats rows = sum of the slices' partials, dkk[c] = dkv[c] . kv[c]
    float4 v0[SL][4][2], kvr[SL][2];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int idx = tid + sl * 256;
      const int row = idx / CPRs, c = idx - row * CPRs;
      const int rr = (idx < Cp * CPRs && row < p.C) ? row : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = p.m_acc0 + (((size_t)bh * S + min(u, S - 1)) * p.C + rr) * D + c * 8;
        v0[sl][u][0] = *reinterpret_cast<const float4*>(src);
        v0[sl][u][1] = *reinterpret_cast<const float4*>(src + 4);
      }
      kvr[sl][0] = *reinterpret_cast<const float4*>(p.kv + (lm + rr) * D + c * 8);
      kvr[sl][1] = *reinterpret_cast<const float4*>(p.kv + (lm + rr) * D + c * 8 + 4);
    }
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int idx = tid + sl * 256;
      const int row = idx / CPRs, c = idx - row * CPRs;
      const bool rok = idx < Cp * CPRs && row < p.C;
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < S) {
          a[0] += v0[sl][u][0].x; a[1] += v0[sl][u][0].y; a[2] += v0[sl][u][0].z; a[3] += v0[sl][u][0].w;
          a[4] += v0[sl][u][1].x; a[5] += v0[sl][u][1].y; a[6] += v0[sl][u][1].z; a[7] += v0[sl][u][1].w;
        }
      }
      if (!rok) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.f;
      }
      float dot = a[0] * kvr[sl][0].x + a[1] * kvr[sl][0].y + a[2] * kvr[sl][0].z + a[3] * kvr[sl][0].w +
                  a[4] * kvr[sl][1].x + a[5] * kvr[sl][1].y + a[6] * kvr[sl][1].z + a[7] * kvr[sl][1].w;
      dot = group_sum<CPRs>(dot);
      if (idx < Cp * CPRs) {
        sts16(R3 + TileL<D>::off(row, c), pack8<E>(a));
        if (c == 0) SC1[row] = rok ? dot : 0.f;
      }
    }
    if (blk == p.nsplit - 1) {                   // (the LAST slice of a (b,h): with uneven slices it is the short one)
      // what the later passes read: sum dZ q (d omega, query side), d qbar rows = s (M1 - u M2) [mis-opt] | s sum dZ q
      // [mis-biased], u qbar (the finish pass), and the per-landmark scalars dbh, dlp = -r
      const bool opt = p.mis == MIS_OPT, biased = p.mis == MIS_BIASED;
      if (tid < p.C) {
        if (p.m_dbh) p.m_dbh[lm + tid] = mg_dbh;
        p.m_dlp[lm + tid] = -sc_v2;
      }
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / CPRs, c = idx - row * CPRs;
        if (idx >= Cp * CPRs || row >= p.C) continue;
        float a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
              a3[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float u_ = 0.f;
        for (int s_ = 0; s_ < S; ++s_) {
          const size_t slot = ((size_t)bh * S + s_) * p.C + row;
          const float* s1 = p.m_acc1 + slot * D + c * 8;
          const float4 x0 = *reinterpret_cast<const float4*>(s1), x1 = *reinterpret_cast<const float4*>(s1 + 4);
          a1[0] += x0.x; a1[1] += x0.y; a1[2] += x0.z; a1[3] += x0.w; a1[4] += x1.x; a1[5] += x1.y; a1[6] += x1.z; a1[7] += x1.w;
          if (opt) {
            u_ += p.m_ml[slot * 4 + 2];
            const float* s2 = p.m_acc2 + slot * D + c * 8;
            const float* s3 = p.m_acc3 + slot * D + c * 8;
            const float4 y0 = *reinterpret_cast<const float4*>(s2), y1 = *reinterpret_cast<const float4*>(s2 + 4);
            const float4 z0 = *reinterpret_cast<const float4*>(s3), z1 = *reinterpret_cast<const float4*>(s3 + 4);
            a2[0] += y0.x; a2[1] += y0.y; a2[2] += y0.z; a2[3] += y0.w; a2[4] += y1.x; a2[5] += y1.y; a2[6] += y1.z; a2[7] += y1.w;
            a3[0] += z0.x; a3[1] += z0.y; a3[2] += z0.z; a3[3] += z0.w; a3[4] += z1.x; a3[5] += z1.y; a3[6] += z1.z; a3[7] += z1.w;
          }
        }
        const size_t o = (lm + row) * D + c * 8;
        *reinterpret_cast<float4*>(p.m_domq + o) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        *reinterpret_cast<float4*>(p.m_domq + o + 4) = make_float4(a1[4], a1[5], a1[6], a1[7]);
        if (opt) {
          const float sc = p.scale;
          const float4 q0 = *reinterpret_cast<const float4*>(p.qbar + o), q1 = *reinterpret_cast<const float4*>(p.qbar + o + 4);
          *reinterpret_cast<float4*>(p.m_dqbar + o) = make_float4(sc * (a2[0] - u_ * a3[0]), sc * (a2[1] - u_ * a3[1]),
                                                                  sc * (a2[2] - u_ * a3[2]), sc * (a2[3] - u_ * a3[3]));
          *reinterpret_cast<float4*>(p.m_dqbar + o + 4) = make_float4(sc * (a2[4] - u_ * a3[4]), sc * (a2[5] - u_ * a3[5]),
                                                                      sc * (a2[6] - u_ * a3[6]), sc * (a2[7] - u_ * a3[7]));
          *reinterpret_cast<float4*>(p.m_uq + o) = make_float4(u_ * q0.x, u_ * q0.y, u_ * q0.z, u_ * q0.w);
          *reinterpret_cast<float4*>(p.m_uq + o + 4) = make_float4(u_ * q1.x, u_ * q1.y, u_ * q1.z, u_ * q1.w);
        } else if (biased && p.m_dqbar) {
          *reinterpret_cast<float4*>(p.m_dqbar + o) = make_float4(p.scale * a1[0], p.scale * a1[1], p.scale * a1[2], p.scale * a1[3]);
          *reinterpret_cast<float4*>(p.m_dqbar + o + 4) = make_float4(p.scale * a1[4], p.scale * a1[5], p.scale * a1[6], p.scale * a1[7]);
        }
      }
    }
    if (tid < Cp) { SC0[tid] = sc_v0; SC2[tid] = sc_v2; }
  } else {
    char* const dst[3] = {R1, nullptr, R3};
    const float* const src[3] = {p.omega + lm * D, nullptr, p.dkv + lm * D};
    stage_rows3<E, D, Cp>(dst, src, p.C, tid);
    if (tid < Cp) { SC0[tid] = sc_v0; SC1[tid] = sc_v1; SC2[tid] = sc_v2; }
  }
  typename LaneOffSel<D>::type lo;
  lo.init(lane);
  const int wr = 4 * g + (li >> 2);
  const int yct = wave % NCT, ysub = wave / NCT;
  const bool ywave = wave < NCT * NSUB;
  int wtr;
  {
    const int colb = (16 * yct + 4 * (li & 3)) * 2;
    wtr = wr * ROWW + ((((colb >> 4)) ^ (wr & WSW)) << 4) + (colb & 15);
  }
  f32x4 acc0[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc0[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int cb = n0; cb < n1; cb += 64) {
    const int tok = cb + wave * 16 + li;
    const bool valid = tok < n1;
    typename E::x8 f1[KS], f2[KS];
    u32x4 raw1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      raw1[ks] = nx1[ks];
      f1[ks] = as_x8<E>(nx1[ks]);
      f2[ks] = as_x8<E>(nx2[ks]);
    }
    issue(cb + 64);
    f32x4 a[NCT], dw[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      a[ct] = dw[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = ct * 16 + li;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[ct] = E::mma(as_x8<E>(lds16(R1 + TileL<D>::off(row, g * KS + ks))), f1[ks], a[ct]);
        dw[ct] = E::mma(as_x8<E>(lds16(R3 + TileL<D>::off(row, g * KS + ks))), f2[ks], dw[ct]);
      }
    }
    float nrm = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float kf[8];
      unpack8<E>(raw1[ks], kf);
#pragma unroll
      for (int i = 0; i < 8; ++i) nrm += kf[i] * kf[i];
    }
    nrm = quad_sum(nrm);
    const bool dead = !valid || (p.mask && p.mask[(size_t)b * p.N + (valid ? tok : 0)]);
    float w1[NCT][4], w2[NCT][4];
    float sdb = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 lk = *reinterpret_cast<const float4*>(SC0 + ct * 16 + 4 * g);
      const float4 dk4 = *reinterpret_cast<const float4*>(SC1 + ct * 16 + 4 * g);
      const float4 rs4 = *reinterpret_cast<const float4*>(SC2 + ct * 16 + 4 * g);
      const float lkv[4] = {lk.x, lk.y, lk.z, lk.w}, dkv4[4] = {dk4.x, dk4.y, dk4.z, dk4.w}, rsv[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bk2 = a[ct][r] * p.scale_log2 - 0.5f * p.scale_log2 * nrm;
        const float pk = dead ? 0.f : fast_exp2(bk2 - lkv[r]);
        const float db = pk * (dw[ct][r] - dkv4[r] + rsv[r]);
        w1[ct][r] = pk;
        w2[ct][r] = db;
        sdb += db;
      }
    }
    sdb = quad_sum(sdb);
    // ---- contraction over c: dv^T = dkv^T . Pk,  dk^T = omega^T . dB ----
    f32x4 acc[DT], acc2[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = acc2[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NCT / 2; ++kk) {
      u32x4 p1, p2;
      p1[0] = pack2<E>(w1[2 * kk][0], w1[2 * kk][1]); p1[1] = pack2<E>(w1[2 * kk][2], w1[2 * kk][3]);
      p1[2] = pack2<E>(w1[2 * kk + 1][0], w1[2 * kk + 1][1]); p1[3] = pack2<E>(w1[2 * kk + 1][2], w1[2 * kk + 1][3]);
      p2[0] = pack2<E>(w2[2 * kk][0], w2[2 * kk][1]); p2[1] = pack2<E>(w2[2 * kk][2], w2[2 * kk][3]);
      p2[2] = pack2<E>(w2[2 * kk + 1][0], w2[2 * kk + 1][1]); p2[3] = pack2<E>(w2[2 * kk + 1][2], w2[2 * kk + 1][3]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* r3 = R3 + (32 * kk) * ROWB + lo.tr[dt];
        const char* r1 = R1 + (32 * kk) * ROWB + lo.tr[dt];
        acc[dt] = E::mma(as_x8<E>(E::tr4(r3), E::tr4(r3 + 16 * ROWB)), as_x8<E>(p1), acc[dt]);
        acc2[dt] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * ROWB)), as_x8<E>(p2), acc2[dt]);
      }
    }
    {
      // unconditional stores (rows past the slice go to the trash line): static store count, see ea_trash_line()
      char* dstv = valid ? p.dv.p + (b * p.dv.sb + h * p.dv.sh + tok * p.dv.sn + DQ * g) * 2 : ea_trash_line();
      // dk: the lane's B-fragment chunks of k hold channels 8 (g KS + ks) .. = DQ g .., the contiguous ownership: the
      // k factor is applied there, after the accumulator pieces have been moved to it
      char* dstk = valid ? p.dk.p + (b * p.dk.sb + h * p.dk.sh + tok * p.dk.sn + DQ * g) * 2 : ea_trash_line() + 32;
      float f2[DQ];
      if constexpr (TileL<D>::NEWTR) {
        u32x4 o0, o1;
        quad_transpose_pack<E>(acc, 1.f, o0, o1);
        stg16(dstv, o0);
        stg16(dstv + 16, o1);
        quad_transpose_f32(acc2, f2);
      } else {
        float f[DQ];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { f[4 * dt + r] = acc[dt][r]; f2[4 * dt + r] = acc2[dt][r]; }
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) stg16(dstv + c * 16, pack8<E>(f + 8 * c));
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float kf[8], o8[8];
        unpack8<E>(raw1[ks], kf);
#pragma unroll
        for (int i = 0; i < 8; ++i) o8[i] = p.scale * f2[8 * ks + i] - p.knorm_coef * kf[i] * sdb;
        stg16(dstk + ks * 16, pack8<E>(o8));
      }
    }
    __syncthreads();
    {
      const int row = wave * 16 + li;
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sts16(TK + TileL<D>::off(row, g * KS + ks), valid ? raw1[ks] : z);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        *reinterpret_cast<u32x2*>(WT + wt_off<Cp>(row, 16 * ct + 4 * g)) =
            u32x2{pack2<E>(w2[ct][0], w2[ct][1]), pack2<E>(w2[ct][2], w2[ct][3])};
    }
    __syncthreads();
    if (ywave) {
#pragma unroll
      for (int kq = 0; kq < 2 / NSUB; ++kq) {
        const int kb = (NSUB == 1 ? kq : ysub) * 32;
        const char* w0 = WT + kb * ROWW + wtr;
        const typename E::x8 b0 = as_x8<E>(E::tr4(w0), E::tr4(w0 + 16 * ROWW));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* tk = TK + kb * ROWB + lo.tr[dt];
          acc0[dt] = E::mma(as_x8<E>(E::tr4(tk), E::tr4(tk + 16 * ROWB)), b0, acc0[dt]);     // sum dB k
        }
      }
    }
  }
  // (the epilogue's lane coordinates are re-derived from an opaque copy of the thread index: kept alive across the chunk
  //  loop they cost three registers the loop does not have -- 16 B / lane of scratch until round 6)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int wave_e = tid_e >> 6, li_e = tid_e & 15, g_e = (tid_e & 63) >> 4;
  const int yct_e = wave_e % NCT, ysub_e = wave_e / NCT;
  const int c = yct_e * 16 + li_e;
  if (!(wave_e < NCT * NSUB) || c >= p.C) { EA_BLK(p, 1); return; }
  const int S = p.nsplit * NSUB;
  const size_t slot = ((size_t)bh * S + blk * NSUB + ysub_e) * p.C + c;
  float* d = p.p_acc0 + slot * D;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<float4*>(d + acc_chan<D>(dt, g_e)) = make_float4(acc0[dt][0], acc0[dt][1], acc0[dt][2], acc0[dt][3]);
  EA_BLK(p, 1);
}

// ------------------------------------------------------------------------------------------
// finish: dq -= s sum_c t[c,n] (u_c qbar_c)  (the softmax-over-sequence correction of t, lara.py:223)
//         dq += d(pooled q)[chunk(n)] / r^2,  dk += d(pooled k)[chunk(n)] / r^2   (lara.py:43,48,145-151)
// ------------------------------------------------------------------------------------------
template <typename E, int D, int NCT>
__global__ __launch_bounds__(256, 3) void lara_fin_kernel(const LaraP p) {
  constexpr int ROWB = D * 2, KS = D / 32, DT = D / 16, DQ = D / 4;
  constexpr int Cp = NCT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;                                  // (u qbar) rows
  char* R2 = R1 + Cp * ROWB;                        // qbar rows
  float* SC1 = reinterpret_cast<float*>(R2 + Cp * ROWB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nbh = p.B * p.H;
  const int blk = blockIdx.x / nbh, bh = blockIdx.x - blk * nbh;
  const int b = bh / p.H, h = bh - b * p.H;
  const size_t lm = (size_t)bh * p.C;
  const bool has_t = p.uq != nullptr;
  const bool pool = p.pool_r > 0;
  const char* qb = p.q.p ? p.q.p + (b * p.q.sb + h * p.q.sh) * 2 : nullptr;
  char* dqb = p.dq.p + (b * p.dq.sb + h * p.dq.sh) * 2;
  char* dkb = p.dk.p ? p.dk.p + (b * p.dk.sb + h * p.dk.sh) * 2 : nullptr;
  const int n0 = p.nsplit == 2 ? p.tok_begin[blk] : blk * p.tok_per_block;
  const int n1 = p.nsplit == 2 ? p.tok_begin[blk + 1] : min(p.N, n0 + p.tok_per_block);
  const int last_tok = n1 - 1;
  const int cpr = pool ? p.pool_gw / p.pool_r : 1;  // chunks per grid row

  u32x4 nx1[KS], nq[DQ / 8], nk[DQ / 8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) nx1[ks] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int c = 0; c < DQ / 8; ++c) nk[c] = u32x4{0u, 0u, 0u, 0u};
  auto issue = [&](int tile_) {
    const int tok_ = min(n0 + tile_ * 16 + li, last_tok);
    if (has_t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) nx1[ks] = ldg16(qb + (tok_ * p.q.sn + (g * KS + ks) * 8) * 2);
    }
#pragma unroll
    for (int c = 0; c < DQ / 8; ++c) {
      nq[c] = ldg16(dqb + (tok_ * p.dq.sn + DQ * g + 8 * c) * 2);
      if (pool) nk[c] = ldg16(dkb + (tok_ * p.dk.sn + DQ * g + 8 * c) * 2);
    }
  };
  if (n0 + wave * 16 < n1) issue(wave);
  float sc_v1 = INFINITY;
  if (has_t && tid < p.C) sc_v1 = p.lse_t[lm + tid] * LOG2E;
  if (has_t) {
    char* const dst[3] = {R1, R2, nullptr};
    const float* const src[3] = {p.uq + lm * D, p.qbar + lm * D, nullptr};
    stage_rows3<E, D, Cp>(dst, src, p.C, tid);
  }
  if (tid < Cp) SC1[tid] = sc_v1;
  typename LaneOffSel<D>::type lo;
  lo.init(lane);
  __syncthreads();

  for (int tile = wave; n0 + tile * 16 < n1; tile += 4) {
    const int tok = n0 + tile * 16 + li;
    const bool valid = tok < n1;
    typename E::x8 f1[KS];
    u32x4 oq[DQ / 8], ok[DQ / 8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f1[ks] = as_x8<E>(nx1[ks]);
#pragma unroll
    for (int c = 0; c < DQ / 8; ++c) { oq[c] = nq[c]; ok[c] = nk[c]; }
    issue(tile + 4);
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_t) {
      float w1[NCT][4];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x4 tt = {0.f, 0.f, 0.f, 0.f};
        const int row = ct * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tt = E::mma(as_x8<E>(lds16(R2 + TileL<D>::off(row, g * KS + ks))), f1[ks], tt);
        const float4 ls = *reinterpret_cast<const float4*>(SC1 + ct * 16 + 4 * g);
        w1[ct][0] = fast_exp2(tt[0] * p.scale_log2 - ls.x); w1[ct][1] = fast_exp2(tt[1] * p.scale_log2 - ls.y);
        w1[ct][2] = fast_exp2(tt[2] * p.scale_log2 - ls.z); w1[ct][3] = fast_exp2(tt[3] * p.scale_log2 - ls.w);
      }
#pragma unroll
      for (int kk = 0; kk < NCT / 2; ++kk) {
        u32x4 p1;
        p1[0] = pack2<E>(w1[2 * kk][0], w1[2 * kk][1]); p1[1] = pack2<E>(w1[2 * kk][2], w1[2 * kk][3]);
        p1[2] = pack2<E>(w1[2 * kk + 1][0], w1[2 * kk + 1][1]); p1[3] = pack2<E>(w1[2 * kk + 1][2], w1[2 * kk + 1][3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* r1 = R1 + (32 * kk) * ROWB + lo.tr[dt];
          acc[dt] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * ROWB)), as_x8<E>(p1), acc[dt]);
        }
      }
    }
    // correction rows in the contiguous ownership (channel DQ g + j) of the dq / dk rows this lane updates
    float cr[DQ];
    if constexpr (TileL<D>::NEWTR) {
      quad_transpose_f32(acc, cr);
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cr[4 * dt + r] = acc[dt][r];
    }
    if (!valid) continue;
    int chunk = 0;
    if (pool) {
      const int y = tok / p.pool_gw, x = tok - y * p.pool_gw;
      chunk = (y / p.pool_r) * cpr + x / p.pool_r;
    }
    const float* pq = pool ? p.dpq + ((size_t)bh * p.pool_L + chunk) * D + DQ * g : nullptr;
    const float* pk = pool ? p.dpk + ((size_t)bh * p.pool_L + chunk) * D + DQ * g : nullptr;
#pragma unroll
    for (int c = 0; c < DQ / 8; ++c) {
      float old[8];
      unpack8<E>(oq[c], old);
#pragma unroll
      for (int i = 0; i < 8; ++i) old[i] -= cr[8 * c + i] * p.scale;
      if (pool) {
        const float4 a0 = *reinterpret_cast<const float4*>(pq + 8 * c), a1 = *reinterpret_cast<const float4*>(pq + 8 * c + 4);
        old[0] += a0.x * p.pool_inv; old[1] += a0.y * p.pool_inv; old[2] += a0.z * p.pool_inv; old[3] += a0.w * p.pool_inv;
        old[4] += a1.x * p.pool_inv; old[5] += a1.y * p.pool_inv; old[6] += a1.z * p.pool_inv; old[7] += a1.w * p.pool_inv;
      }
      stg16(dqb + (tok * p.dq.sn + DQ * g + 8 * c) * 2, pack8<E>(old));
      if (pool) {
        float kk8[8];
        unpack8<E>(ok[c], kk8);
        const float4 b0 = *reinterpret_cast<const float4*>(pk + 8 * c), b1 = *reinterpret_cast<const float4*>(pk + 8 * c + 4);
        kk8[0] += b0.x * p.pool_inv; kk8[1] += b0.y * p.pool_inv; kk8[2] += b0.z * p.pool_inv; kk8[3] += b0.w * p.pool_inv;
        kk8[4] += b1.x * p.pool_inv; kk8[5] += b1.y * p.pool_inv; kk8[6] += b1.z * p.pool_inv; kk8[7] += b1.w * p.pool_inv;
        stg16(dkb + (tok * p.dk.sn + DQ * g + 8 * c) * 2, pack8<E>(kk8));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
size_t lara_f_lds(int which, int D, int Cp) {
  if (which == 0) return (size_t)3 * Cp * D * 2 + (size_t)2 * 64 * D * 2 + (size_t)4 * 64 * Cp * 2 + (size_t)7 * Cp * sizeof(float);
  if (which == 1) return (size_t)2 * Cp * D * 2 + (size_t)64 * D * 2 + (size_t)64 * Cp * 2 + (size_t)3 * Cp * sizeof(float);
  return (size_t)2 * Cp * D * 2 + (size_t)Cp * sizeof(float);
}

// Two slices per (b,h): equal halves unless the 2 BH workgroups do not fit on the chip at once
// (BH < slots < 2 BH) -- then the second "round" would run on a mostly idle chip.  With slice-major
// dispatch the BH first slices start next to slots - BH second slices and the remaining second slices
// follow in rb = ceil(BH / (slots - BH)) rounds, so first : second = rb : 1 keeps every slot busy to
// the end (B*h = 384 on 256 CUs x 2 workgroups: 592 + 192 tokens instead of 448 + 336).
static int f_env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
static void lara_f_plan(LaraP& p, int slots, int F, int last_extra = 0) {
  if (p.nsplit != 2) return;
  const int BH = p.B * p.H, gran = 64;
  p.tok_begin[0] = 0; p.tok_begin[1] = p.tok_per_block < p.N ? p.tok_per_block : p.N; p.tok_begin[2] = p.N;
  // Key side with the query side's partials merged in its prologue (round 6, workgroup timelines: the LAST slice of a (b,h)
  // also forms and writes the merged tensors the later passes read -- two dependent rounds of loads before its first chunk;
  // with equal halves the second slices ended 8 us after the first ones, 42 vs 34 us at cfg3, 20.7 vs 13.7 at cfg2): the last
  // slice is shorter by that work, `last_extra` token-times, when both slices of every (b,h) are resident at once.
  if (last_extra > 0 && slots >= 2 * BH) {
    int first = ((p.N + last_extra) / 2 + gran / 2) / gran * gran;
    if (first >= p.N) first = (p.N - 1) / gran * gran;          // (the last slice keeps at least one token: it writes the merged tensors)
    if (first >= gran && first < p.N) p.tok_begin[1] = first;
    return;
  }
  if (BH < slots && slots < 2 * BH) {
    const int rb = (BH + (slots - BH) - 1) / (slots - BH);
    // F: fixed prologue / epilogue of a workgroup, in token-times (query side, round 3: ~8 us against 0.05 us per token)
    int b = ((p.N - (rb - 1) * F) / (1 + rb) + gran / 2) / gran * gran;
    static const int b_env = f_env_int("EA_LARA_FQ_B", 0);      // dev knob: tokens of the second slice (query side)
    if (b_env > 0 && F != 96) b = b_env;
    if (b >= 16 && b < p.N) p.tok_begin[1] = p.N - b;
    // Short sequences (round 6, N = 196 at B*h = 384): a second slice cannot even carry its own fixed cost -- equal halves run
    // as 1.5 rounds of (F + N / 2) each, i.e. 2 F + N, where ONE slice per (b,h) takes F + N in a single round (-7 us of 28
    // at cfg2, workgroup timelines of the -DEA_PROFILE build).  The slice count stays 2 (the partial buffers are laid out
    // for it): the second slice is EMPTY -- its workgroups skip the staging and write zero partials.
    else if (b < 16 && F != 96) p.tok_begin[1] = p.N;
  }
}

static int f_device_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <typename K>
static int f_occupancy(K kern, size_t lds) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || n <= 0) n = 2;
  return n;
}

template <typename E, int D, int NCT>
static int launch_f(int which, LaraP& p, hipStream_t st) {
  const size_t lds = lara_f_lds(which == 3 ? 1 : which, D, NCT * 16);
  const dim3 grid((unsigned)(p.B * p.H * p.nsplit)), block(256);
  static int occ[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // resident workgroups per CU of the instantiations
  static const int fq_fixed = f_env_int("EA_LARA_FQ_F", 160);
  static const int fk_extra = f_env_int("EA_LARA_FK_X", 128);      // dev knob: the merging slice's extra work in token-times
#define EA_LF(K, slot, ...)                                                                           \
  do {                                                                                                \
    if (lds > 64 * 1024) {                                                                            \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K<__VA_ARGS__>),              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                             \
    }                                                                                                 \
    if (!occ[slot]) occ[slot] = f_occupancy(&K<__VA_ARGS__>, lds);                                    \
    lara_f_plan(p, occ[slot] * f_device_cus(), which == 0 ? fq_fixed : 96, which == 3 ? fk_extra : 0); \
    hipLaunchKernelGGL((K<__VA_ARGS__>), grid, block, lds, st, p);                                    \
  } while (0)
  if (which == 0) {
    // r-pairs of the last landmark tile that can be populated (see lara_fq_kernel); the estimator variant
    const bool lh1 = NCT == 4 && p.C - 16 * (NCT - 1) <= 2;
    if (p.mis == MIS_OPT) { if (lh1) EA_LF(lara_fq_kernel, 0, E, D, NCT, (NCT == 4 ? 1 : 2), MIS_OPT); else EA_LF(lara_fq_kernel, 3, E, D, NCT, 2, MIS_OPT); }
    else if (p.mis == MIS_BIASED) { if (lh1) EA_LF(lara_fq_kernel, 4, E, D, NCT, (NCT == 4 ? 1 : 2), MIS_BIASED); else EA_LF(lara_fq_kernel, 5, E, D, NCT, 2, MIS_BIASED); }
    else { if (lh1) EA_LF(lara_fq_kernel, 6, E, D, NCT, (NCT == 4 ? 1 : 2), MIS_BH); else EA_LF(lara_fq_kernel, 7, E, D, NCT, 2, MIS_BH); }
  } else if (which == 1) EA_LF(lara_fk_kernel, 1, E, D, NCT, false);
  else if (which == 3) EA_LF(lara_fk_kernel, 8, E, D, NCT, true);
  else EA_LF(lara_fin_kernel, 2, E, D, NCT);
#undef EA_LF
  return (int)hipGetLastError();
}

template <typename E, int D>
static int launch_f_nct(int which, LaraP& p, hipStream_t st) {
  if (p.NCT <= 2) return launch_f<E, D, 2>(which, p, st);
  if (p.NCT <= 4) return launch_f<E, D, 4>(which, p, st);
  return EA_E_UNSUPPORTED;
}

// which: 0 query side, 1 key side, 2 finish, 3 key side with the query side's partials merged in its prologue
int lara_f_dispatch(int which, const LaraP& p0, int dtype, hipStream_t st) {
  LaraP p = p0;
  p.prof = nullptr;
  // key side: row offsets inside a (b,h) are formed in 32 bits (lara_fk_kernel)
  if ((which == 1 || which == 3) && ((int64_t)p.N * p.k.sn >= (1ll << 30) || (int64_t)p.N * p.v.sn >= (1ll << 30))) return EA_E_UNSUPPORTED;
  if (which == 0 && ((int64_t)p.N * p.q.sn >= (1ll << 30) || (int64_t)p.N * p.dout.sn >= (1ll << 30))) return EA_E_UNSUPPORTED;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "lara_f", which);
#endif
  if (dtype == EA_BF16) {
    if (p.D == 64) return launch_f_nct<BF16, 64>(which, p, st);
    if (p.D == 32) return launch_f_nct<BF16, 32>(which, p, st);
  } else if (dtype == EA_F16) {
    if (p.D == 64) return launch_f_nct<F16, 64>(which, p, st);
    if (p.D == 32) return launch_f_nct<F16, 32>(which, p, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
