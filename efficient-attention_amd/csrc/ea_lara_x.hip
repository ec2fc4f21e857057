// ea_lara_x.hip -- LARA passes in the token-column layout (see ea_lara.h).
//
//   LX_FWD   out_n = sum_c W[c,n] kv_c,  W = softmax_c(log alpha + s w_c.q_n + lse_k - log_prop)
//            (lara.py:201,221-246; the -s|q_n|^2/2 term of log_proj_q is constant in c and cancels)
//   LX_BWDQ  recomputes W, forms dZ / d(alpha) / dt and writes the part of dq that needs no
//            sequence-wide sum, plus the per-token scalars (lse_Z, mean_c t, dout.out, sum_c dalpha)
//            the token-row pass needs
//   LX_BWDK  dk, dv from Pk = softmax_m(log_proj_k) recomputed with the saved lse_k
//   LX_QCORR dq -= s sum_c t[c,n] (u_c qbar_c): the softmax-over-sequence correction of t
// One 16-token tile per wave step: token rows come straight from global memory as MFMA B
// operands; the landmark matrices live in LDS (row-major for the score MFMAs, transposed for
// the contraction over c); each lane ends up owning D/4 contiguous channels of one token.
#include "ea_lara.h"

namespace ea {

template <int D> struct LxCfg {
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
};

// MIS >= 0: estimator variant at compile time (the forward combine; with mis as a runtime value its tile loop was cut into
// 17 basic blocks); MIS = -1: read mis (the rare two-pass backward modes and the Performer modes).
template <typename E, int D, int NCT, int MODE, int MIS>
__global__ __launch_bounds__(256, NCT <= 4 ? 3 : 1) void lara_x_kernel(const LaraP p) {
  const int mis = MIS >= 0 ? MIS : p.mis;
  using Cfg = LxCfg<D>;
  constexpr int ROWB = Cfg::ROWB, KS = Cfg::KS, DT = Cfg::DT, DQ = Cfg::DQ;
  constexpr int Cp = NCT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;                                  // omega rows
  char* R2 = R1 + Cp * ROWB;                        // qbar rows
  char* R3 = R2 + Cp * ROWB;                        // kv / dkv / uq rows
  // Round 3: the transposed operands of the contraction over c are ds_read_b64_tr_b16 reads of these row-major tiles
  // (round 1 kept second, transposed copies M1 / M2 staged with 2-byte LDS stores and read with 2-way bank conflicts).
  char* const MA = (MODE == LX_BWDQ || MODE == LX_PBWDQ) ? R1 : R3;     // first operand: kv | omega | dkv | uq
  char* const MB = (MODE == LX_BWDQ || MODE == LX_PBWDQ) ? R2 : R1;     // second: qbar (query side) | omega (key side)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = blockIdx.x / p.nsplit, blk = blockIdx.x - bh * p.nsplit;
  const int b = bh / p.H, h = bh - b * p.H;
  const size_t lm = (size_t)bh * p.C;               // landmark row offset
  const size_t lmw = (size_t)(p.w_per_head ? h : bh) * p.C;   // ... of omega / W
  constexpr bool PERF = MODE >= LX_POUT && MODE != LX_FWDM;
  constexpr bool FWD = MODE == LX_FWD || MODE == LX_FWDM;      // LX_FWDM: the statistics pass's slice partials merged on load
  const bool use_t = mis != MIS_BH && !PERF;

  EA_STAMP(p, 0);
  EA_BLK(p, 0);
  constexpr bool KEYS = MODE == LX_BWDK || MODE == LX_PBWDK;
  constexpr bool TWO_TOK = MODE == LX_BWDQ || MODE == LX_BWDK || MODE == LX_PBWDQ || MODE == LX_PBWDK;
  const T4l& tk1 = KEYS ? p.k : p.q;
  const char* t1b = tk1.p + (b * tk1.sb + h * tk1.sh) * 2;
  const T4l& tk2 = KEYS ? p.v : p.dout;
  const char* t2b = tk2.p ? tk2.p + (b * tk2.sb + h * tk2.sh) * 2 : nullptr;
  const float invC = 1.f / (float)p.C;
  const int n0 = blk * p.tok_per_block;
  const int n1 = min(p.N, n0 + p.tok_per_block);

  // Software prefetch: the token fragments of this wave's NEXT tile are in flight while the
  // current tile computes, so a tile costs one exposed memory round trip per wave, not one per tile.
  constexpr bool NEED_O = MODE == LX_PBWDQ;
  // (Straight-line on purpose: the tile index and the token are clamped instead of branched on --
  // rows fetched for tokens >= n1 are never stored -- because with a conditional refill hipcc
  // parks the fragment arrays in scratch memory and the prefetch turns synchronous.)
  u32x4 nx1[KS], nx2[KS], nx3[KS];
  const int last_tok = n1 - 1;
  auto issue = [&](int tile_) {
    const int tok_ = min(n0 + tile_ * 16 + li, last_tok);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int eo = (g * KS + ks) * 8;
      nx1[ks] = ldg16(t1b + (tok_ * tk1.sn + eo) * 2);
      if (TWO_TOK) nx2[ks] = ldg16(t2b + (tok_ * tk2.sn + eo) * 2);
      if (NEED_O) nx3[ks] = ldg16(p.o.p + (b * p.o.sb + h * p.o.sh + tok_ * p.o.sn + eo) * 2);
    }
  };
  // the first tile's token rows are requested before anything else: their round trip hides
  // behind the staging of the landmark matrices
  if (n0 + wave * 16 < n1) issue(wave);   // (uniform per wave)
  // per-landmark scalars: one landmark per thread (Cp <= 128), loaded before the matrices so that
  // every global load of the prologue is in flight together (one exposed round trip, not two)
  float sc_v0 = -INFINITY, sc_v1 = INFINITY, sc_v2 = 1.f;
  float mg_lsek = 0.f, mg_cst = 0.f, mg_lset = 0.f;
  {
    const int c = tid;
    const bool ok = c < p.C;
    if (MODE == LX_BWDK) { sc_v0 = INFINITY; sc_v1 = 0.f; sc_v2 = 0.f; }
    if (MODE == LX_PBWDK) sc_v2 = 0.f;
    if (ok) {
      if (MODE == LX_FWD || MODE == LX_BWDQ) {
        sc_v0 = p.cst[lm + c] * LOG2E;
        if (mis == MIS_OPT) sc_v2 = p.bhv[lm + c];
      }
      if ((MODE == LX_FWD || MODE == LX_BWDQ || MODE == LX_QCORR) && mis == MIS_OPT) sc_v1 = p.lse_t[lm + c] * LOG2E;
      if (MODE == LX_FWDM) {
        // ea_lara_merge.hip (lara_merge_fwd_kernel), per-landmark scalars: log-sum-exp merge of the S slices
        const int S = p.m_S;
        float4 m4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) m4[u] = *reinterpret_cast<const float4*>(p.m_ml + (((size_t)bh * S + min(u, S - 1)) * p.C + c) * 4);
        float mk = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) { mk = fmaxf(mk, m4[u].x); mt = fmaxf(mt, m4[u].z); }
        float lk = 0.f, lt = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < S) {
            lk += m4[u].y * __expf(m4[u].x - mk);
            if (mis == MIS_OPT) lt += m4[u].w * __expf(m4[u].z - mt);
          }
        }
        const float lsek = mk + __logf(lk);
        const float cstv = lsek - p.m_lp[lm + c];
        sc_v0 = cstv * LOG2E;
        float lsetv = 0.f;
        if (mis == MIS_OPT) { sc_v2 = p.bhv[lm + c]; lsetv = mt + __logf(lt); sc_v1 = lsetv * LOG2E; }
        mg_lsek = lsek; mg_cst = cstv; mg_lset = lsetv;       // stored after the staging (a store here would fence the loads below)
      }
      if (MODE == LX_BWDK) { sc_v0 = p.lse_k[lm + c] * LOG2E; sc_v1 = p.dkk[lm + c]; sc_v2 = p.rsum[lm + c]; }
      if (MODE == LX_POUT || MODE == LX_PBWDQ) sc_v0 = p.cst[lm + c];       // sum_n phi(k_n)[j]
      if (MODE == LX_PBWDK) sc_v2 = p.rsum[lm + c];                         // d ksum[j]
    }
  }
  // ---- stage the landmark matrices: ALL global loads are issued before the first conversion /
  // LDS store, so the workgroup pays one memory round trip here instead of one per matrix ----
  {
    constexpr int CPRs = D / 8;
    constexpr int SL = (Cp * CPRs + 255) / 256;        // (row, 8-channel chunk) slots per thread
    const float* rsrc[3] = {nullptr, nullptr, nullptr};
    if (MODE != LX_QCORR) rsrc[0] = p.omega + lmw * D;
    if (MODE != LX_BWDK && MODE != LX_PBWDK && use_t) rsrc[1] = p.qbar + lm * D;
    if (MODE == LX_BWDQ || MODE == LX_PBWDQ || MODE == LX_FWD || MODE == LX_POUT) rsrc[2] = p.kv + lm * D;
    if (MODE == LX_BWDK || MODE == LX_PBWDK) rsrc[2] = p.dkv + lm * D;
    if (MODE == LX_QCORR) rsrc[2] = p.uq + lm * D;
    char* rdst[3] = {R1, R2, R3};
    float4 rb[3][SL][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* src = rsrc[j];
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / CPRs, c = idx - row * CPRs;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (src && idx < Cp * CPRs && row < p.C) {
          lo = *reinterpret_cast<const float4*>(src + (size_t)row * D + c * 8);
          hi = *reinterpret_cast<const float4*>(src + (size_t)row * D + c * 8 + 4);
        }
        rb[j][sl][0] = lo; rb[j][sl][1] = hi;
      }
    }
    // (after the loop above: the loads of omega / qbar are in flight while the partials are fetched and merged)
    if constexpr (MODE == LX_FWDM) {
      // kv rows = sum_s kv_s e^(m_s - m) / sum_s l_s e^(m_s - m): the merge kernel's arithmetic on this thread's (row, chunk)
      // slots; every load first, then the arithmetic and the stores
      const int S = p.m_S;
      float m8[SL][4], l8[SL][4];
      float4 v8[SL][4][2];
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / CPRs, c = idx - row * CPRs;
        const int rr = (idx < Cp * CPRs && row < p.C) ? row : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t slot = ((size_t)bh * S + min(u, S - 1)) * p.C + rr;
          m8[sl][u] = p.m_ml[slot * 4];
          l8[sl][u] = p.m_ml[slot * 4 + 1];
          v8[sl][u][0] = *reinterpret_cast<const float4*>(p.m_acc0 + slot * D + c * 8);
          v8[sl][u][1] = *reinterpret_cast<const float4*>(p.m_acc0 + slot * D + c * 8 + 4);
        }
      }
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / CPRs, c = idx - row * CPRs;
        if (idx >= Cp * CPRs) continue;
        const bool rok = row < p.C;
        float mk = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) mk = fmaxf(mk, m8[sl][u]);
        float lk = 0.f;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < S) {
            const float w = __expf(m8[sl][u] - mk);
            lk += l8[sl][u] * w;
            a0.x += v8[sl][u][0].x * w; a0.y += v8[sl][u][0].y * w; a0.z += v8[sl][u][0].z * w; a0.w += v8[sl][u][0].w * w;
            a1.x += v8[sl][u][1].x * w; a1.y += v8[sl][u][1].y * w; a1.z += v8[sl][u][1].z * w; a1.w += v8[sl][u][1].w * w;
          }
        }
        const float iv = 1.f / lk;
        const float f[8] = {a0.x * iv, a0.y * iv, a0.z * iv, a0.w * iv, a1.x * iv, a1.y * iv, a1.z * iv, a1.w * iv};
        const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sts16(R3 + TileL<D>::off(row, c), pack8<E>(rok ? f : z8));
        if (rok && blk == p.nsplit - 1) {
          float* dst = p.m_kv + (lm + row) * D + c * 8;
          *reinterpret_cast<float4*>(dst) = make_float4(f[0], f[1], f[2], f[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(f[4], f[5], f[6], f[7]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const bool on = rsrc[j] != nullptr;
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int idx = tid + sl * 256;
        const int row = idx / CPRs, c = idx - row * CPRs;
        if (!on || idx >= Cp * CPRs) continue;
        const float4 lo = rb[j][sl][0], hi = rb[j][sl][1];
        const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        sts16(rdst[j] + TileL<D>::off(row, c), pack8<E>(f));
      }
    }
  }

  if (MODE == LX_FWDM && blk == p.nsplit - 1 && tid < p.C) {            // the merged scalars, for the backward
    p.m_lsek[lm + tid] = mg_lsek;
    p.m_cst[lm + tid] = mg_cst;
    if (mis == MIS_OPT) p.m_lset[lm + tid] = mg_lset;
  }
  EA_STAMP(p, 1);
  // per-landmark scalars live in LDS (three [Cp] fp32 vectors); lanes read the entries of their
  // rows c = 16 ct + 4 g + r at the point of use instead of pinning 6 x NCT x 4 registers
  float* SC0 = reinterpret_cast<float*>(R3 + Cp * ROWB);
  float* SC1 = SC0 + Cp;
  float* SC2 = SC1 + Cp;
  if (tid < Cp) { SC0[tid] = sc_v0; SC1[tid] = sc_v1; SC2[tid] = sc_v2; }
  struct LdsVec {
    const float* base; int g;
    EA_DEV float operator()(int ct, int r) const { return base[ct * 16 + 4 * g + r]; }
    EA_DEV float4 v4(int ct) const { return *reinterpret_cast<const float4*>(base + ct * 16 + 4 * g); }
  };
  const LdsVec cst2{SC0, g}, lset2{SC1, g}, bhv{SC2, g}, lsek2{SC0, g}, dkk{SC1, g}, rs{SC2, g};
  const float stabk2 = (MODE == LX_PBWDK) ? p.stab[bh] * LOG2E : 0.f;
  typename LaneOffSel<D>::type lo;
  lo.init(lane);
  __syncthreads();
  EA_STAMP(p, 2);
  int prof_it = 0;
  (void)prof_it;

  for (int tile = wave; n0 + tile * 16 < n1; tile += 4) {
    const int tok = n0 + tile * 16 + li;
    const bool valid = tok < n1;
    typename E::x8 f1[KS], f2[KS];
    u32x4 raw1[KS], raw3[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      raw1[ks] = nx1[ks];
      raw3[ks] = nx3[ks];
      f1[ks] = as_x8<E>(nx1[ks]);
      f2[ks] = as_x8<E>(nx2[ks]);
    }
    issue(tile + 4);
    if (prof_it < 8) EA_STAMP(p, 3 + prof_it * 5);
    // ---- score tiles ----
    f32x4 a[NCT], tt[NCT], dw[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      a[ct] = tt[ct] = dw[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = ct * 16 + li;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (MODE != LX_QCORR) a[ct] = E::mma(as_x8<E>(lds16(R1 + TileL<D>::off(row, g * KS + ks))), f1[ks], a[ct]);
        if (MODE != LX_BWDK && use_t) tt[ct] = E::mma(as_x8<E>(lds16(R2 + TileL<D>::off(row, g * KS + ks))), f1[ks], tt[ct]);
        if (TWO_TOK) dw[ct] = E::mma(as_x8<E>(lds16(R3 + TileL<D>::off(row, g * KS + ks))), f2[ks], dw[ct]);
      }
    }
    if (prof_it < 8) EA_STAMP(p, 4 + prof_it * 5);
    // ---- elementwise stage -> weight tiles w1 (x M1) and w2 (x M2) ----
    float w1[NCT][4], w2[NCT][4];
    float sdb = 0.f, pden = 1.f;
    if (FWD || MODE == LX_BWDQ) {
      // The stage is VALU-bound (16 (c, n) entries per lane and tile), so it is written on float2
      // values (v_pk_fma/mul/add_f32) and avoids per-entry log2 / rcp: with
      //   Z = log alpha + s w.q + cst,  softmax_c Z = alpha 2^z / sum_c alpha 2^z,  z = Z - log alpha
      // alpha enters as a factor, and d(alpha) = dZ / alpha = 2^z (dW - rd) / sum needs no division.
      const float s2 = p.scale_log2;
      const f32x2 s22 = {s2, s2};
      f32x2 tv[NCT][2], ez[NCT][2];     // ez: 2^(z - mx), zeroed in backward where alpha is clamped
      f32x2 tl2 = {0.f, 0.f};
      if (mis == MIS_OPT) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const float4 ls = lset2.v4(ct);
          const f32x2 x0 = f32x2{tt[ct][0], tt[ct][1]} * s22 - f32x2{ls.x, ls.y};
          const f32x2 x1 = f32x2{tt[ct][2], tt[ct][3]} * s22 - f32x2{ls.z, ls.w};
          tv[ct][0] = f32x2{fast_exp2(x0[0]), fast_exp2(x0[1])};
          tv[ct][1] = f32x2{fast_exp2(x1[0]), fast_exp2(x1[1])};
          tl2 += tv[ct][0] + tv[ct][1];
        }
      }
      const float tmean = quad_sum(tl2[0] + tl2[1]) * invC;
      float mx = -INFINITY;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const float4 cs = cst2.v4(ct);
        f32x2 z0 = f32x2{a[ct][0], a[ct][1]} * s22 + f32x2{cs.x, cs.y};
        f32x2 z1 = f32x2{a[ct][2], a[ct][3]} * s22 + f32x2{cs.z, cs.w};
        if (mis == MIS_BIASED) {
          z0 += f32x2{tt[ct][0], tt[ct][1]} * s22;
          z1 += f32x2{tt[ct][2], tt[ct][3]} * s22;
        }
        ez[ct][0] = z0; ez[ct][1] = z1;
        mx = fmaxf(fmaxf(mx, fmaxf(z0[0], z0[1])), fmaxf(z1[0], z1[1]));
      }
      mx = quad_max(mx);
      const f32x2 mx2 = {mx, mx};
      f32x2 ss2 = {0.f, 0.f};
      f32x2 wv[NCT][2];                                  // alpha 2^(z - mx): un-normalised weights
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x2 x = ez[ct][hh] - mx2;
          ez[ct][hh] = f32x2{fast_exp2(x[0]), fast_exp2(x[1])};
        }
        if (mis == MIS_OPT) {
          const float4 bv = bhv.v4(ct);
          const float kt = -p.kappa * tmean;
          const f32x2 kap = {p.kappa, p.kappa};
          const f32x2 a0 = kap * tv[ct][0] + f32x2{bv.x + kt, bv.y + kt};
          const f32x2 a1 = kap * tv[ct][1] + f32x2{bv.z + kt, bv.w + kt};
          wv[ct][0] = ez[ct][0] * f32x2{fmaxf(a0[0], 1e-8f), fmaxf(a0[1], 1e-8f)};
          wv[ct][1] = ez[ct][1] * f32x2{fmaxf(a1[0], 1e-8f), fmaxf(a1[1], 1e-8f)};
          if (MODE == LX_BWDQ) {                         // d(alpha) = 0 where the clamp is active
            ez[ct][0] = f32x2{a0[0] > 1e-8f ? ez[ct][0][0] : 0.f, a0[1] > 1e-8f ? ez[ct][0][1] : 0.f};
            ez[ct][1] = f32x2{a1[0] > 1e-8f ? ez[ct][1][0] : 0.f, a1[1] > 1e-8f ? ez[ct][1][1] : 0.f};
          }
        } else {
          wv[ct][0] = ez[ct][0];
          wv[ct][1] = ez[ct][1];
        }
        ss2 += wv[ct][0] + wv[ct][1];
      }
      const float ssum = quad_sum(ss2[0] + ss2[1]);
      const float inv = fast_rcp(ssum);
      if (FWD) {
        pden = inv;                                       // normalisation folded into the output scale
        if (p.lseZ && valid && g == 0) {                  // kept for the fused backward (ea_lara_bwd_q_fused)
          const size_t o = (size_t)bh * p.N + tok;
          p.lseZ[o] = mx + fast_log2(ssum);
          p.tmean[o] = tmean;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          w1[ct][0] = wv[ct][0][0]; w1[ct][1] = wv[ct][0][1];
          w1[ct][2] = wv[ct][1][0]; w1[ct][3] = wv[ct][1][1];
        }
      } else {
        const f32x2 inv2 = {inv, inv};
        f32x2 rd2 = {0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          wv[ct][0] *= inv2; wv[ct][1] *= inv2;                               // W
          rd2 += wv[ct][0] * f32x2{dw[ct][0], dw[ct][1]} + wv[ct][1] * f32x2{dw[ct][2], dw[ct][3]};
        }
        const float rd = quad_sum(rd2[0] + rd2[1]);                            // = dout_n . out_n
        const f32x2 rdv = {rd, rd};
        f32x2 sda2 = {0.f, 0.f};
        f32x2 da[NCT][2];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x2 dd = f32x2{dw[ct][2 * hh], dw[ct][2 * hh + 1]} - rdv;  // dW - rd
            const f32x2 dz = wv[ct][hh] * dd;
            w1[ct][2 * hh] = dz[0]; w1[ct][2 * hh + 1] = dz[1];
            if (mis == MIS_OPT) {
              da[ct][hh] = ez[ct][hh] * inv2 * dd;                             // dZ / alpha
              sda2 += da[ct][hh];
            }
          }
        }
        const float sda = quad_sum(sda2[0] + sda2[1]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (mis == MIS_OPT) {
              const float m = sda * invC;
              const f32x2 kap = {p.kappa, p.kappa};
              const f32x2 t2 = tv[ct][hh] * kap * (da[ct][hh] - f32x2{m, m});  // t * dt
              w2[ct][2 * hh] = t2[0]; w2[ct][2 * hh + 1] = t2[1];
            } else if (mis == MIS_BIASED) {
              w2[ct][2 * hh] = w1[ct][2 * hh]; w2[ct][2 * hh + 1] = w1[ct][2 * hh + 1];   // dT = dZ
            }
          }
        if (valid && g == 0) {
          const size_t o = (size_t)bh * p.N + tok;
          p.lseZ[o] = mx + fast_log2(ssum);
          p.tmean[o] = tmean;
          p.rowdot[o] = rd;
          p.sda[o] = sda;
        }
      }
    } else if (PERF) {
      // squared norm of this token (lane holds D/4 channels) -> log2-domain diagonal term
      float nrm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float xf[8];
        unpack8<E>(raw1[ks], xf);
#pragma unroll
        for (int i = 0; i < 8; ++i) nrm += xf[i] * xf[i];
      }
      const float diag2 = p.norm_coef2 * quad_sum(nrm);
      float stab2 = stabk2;
      if (MODE != LX_PBWDK) {                       // queries: stabiliser = max over features
        float mx = -INFINITY;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ct * 16 + 4 * g + r < p.C) mx = fmaxf(mx, a[ct][r] * p.scale_log2);
        stab2 = quad_max(mx);
      }
      const bool dead = (MODE == LX_PBWDK) && (!valid || (p.mask && p.mask[(size_t)b * p.N + (valid ? tok : 0)]));
      float phi[NCT][4];
      float den = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = ct * 16 + 4 * g + r < p.C;
          phi[ct][r] = (ok && !dead) ? p.ratio * fast_exp2(a[ct][r] * p.scale_log2 - diag2 - stab2) + p.feps : 0.f;
          if (MODE != LX_PBWDK && ok) den += phi[ct][r] * cst2(ct, r);
        }
      if (MODE == LX_POUT) {
        den = quad_sum(den);
        pden = fast_rcp(fmaxf(den, 1e-2f));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) w1[ct][r] = phi[ct][r];
      } else if (MODE == LX_PBWDQ) {
        den = quad_sum(den);
        const float invden = fast_rcp(fmaxf(den, 1e-2f));
        // dout . out in the B-fragment layout
        float dd = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          float x8[8], y8[8];
          unpack8<E>(raw3[ks], x8);
          unpack8<E>(__builtin_bit_cast(u32x4, f2[ks]), y8);
#pragma unroll
          for (int i = 0; i < 8; ++i) dd += x8[i] * y8[i];
        }
        dd = quad_sum(dd);
        const float dden = den > 1e-2f ? -dd * invden : 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = ct * 16 + 4 * g + r < p.C;
            const float dphi = dw[ct][r] * invden + cst2(ct, r) * dden;
            const float dz = ok ? dphi * (phi[ct][r] - p.feps) : 0.f;
            w1[ct][r] = dz;
            sdb += dz;
          }
        sdb = quad_sum(sdb);
        if (valid && g == 0) {
          const size_t o = (size_t)bh * p.N + tok;
          p.lseZ[o] = stab2; p.tmean[o] = invden; p.rowdot[o] = dden;
        }
      } else {   // LX_PBWDK
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = ct * 16 + 4 * g + r < p.C;
            const float dz = (ok && !dead) ? (dw[ct][r] + rs(ct, r)) * (phi[ct][r] - p.feps) : 0.f;
            w1[ct][r] = phi[ct][r];
            w2[ct][r] = dz;
            sdb += dz;
          }
        sdb = quad_sum(sdb);
      }
    } else if (MODE == LX_QCORR) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) w1[ct][r] = fast_exp2(tt[ct][r] * p.scale_log2 - lset2(ct, r));
    } else {   // LX_BWDK
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
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bk2 = a[ct][r] * p.scale_log2 - 0.5f * p.scale_log2 * nrm;
          const float pk = dead ? 0.f : fast_exp2(bk2 - lsek2(ct, r));
          const float db = pk * (dw[ct][r] - dkk(ct, r) + rs(ct, r));
          w1[ct][r] = pk;
          w2[ct][r] = db;
          sdb += db;
        }
      sdb = quad_sum(sdb);
    }
    if (prof_it < 8) EA_STAMP(p, 5 + prof_it * 5);
    // ---- contraction over c: out^T[d][n] = M1^T . w1 (+ M2^T . w2) ----
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc2[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool two = (MODE == LX_BWDK) || (MODE == LX_PBWDK) || (MODE == LX_BWDQ && use_t);
#pragma unroll
    for (int kk = 0; kk < NCT / 2; ++kk) {
      u32x4 p1, p2;
      p1[0] = pack2<E>(w1[2 * kk][0], w1[2 * kk][1]); p1[1] = pack2<E>(w1[2 * kk][2], w1[2 * kk][3]);
      p1[2] = pack2<E>(w1[2 * kk + 1][0], w1[2 * kk + 1][1]); p1[3] = pack2<E>(w1[2 * kk + 1][2], w1[2 * kk + 1][3]);
      if (two) {
        p2[0] = pack2<E>(w2[2 * kk][0], w2[2 * kk][1]); p2[1] = pack2<E>(w2[2 * kk][2], w2[2 * kk][3]);
        p2[2] = pack2<E>(w2[2 * kk + 1][0], w2[2 * kk + 1][1]); p2[3] = pack2<E>(w2[2 * kk + 1][2], w2[2 * kk + 1][3]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* r1 = MA + (32 * kk) * ROWB + lo.tr[dt];
        acc[dt] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * ROWB)), as_x8<E>(p1), acc[dt]);
        if (two) {
          const char* r2 = MB + (32 * kk) * ROWB + lo.tr[dt];
          if (KEYS) acc2[dt] = E::mma(as_x8<E>(E::tr4(r2), E::tr4(r2 + 16 * ROWB)), as_x8<E>(p2), acc2[dt]);
          else acc[dt] = E::mma(as_x8<E>(E::tr4(r2), E::tr4(r2 + 16 * ROWB)), as_x8<E>(p2), acc[dt]);
        }
      }
    }
    if (prof_it < 8) EA_STAMP(p, 6 + prof_it * 5);
    // ---- store: lane owns channels DQ*g .. DQ*g+DQ-1 of token `tok` (the accumulator pieces of the transpose-read
    // layout are first moved between the four lanes of the token: quad_transpose, ea_common.h) ----
    float f[DQ];
    if (FWD || MODE == LX_POUT) {
      // issued unconditionally (rows past the end go to the trash line): a static store count lets the wait for the next
      // tile's prefetched rows leave this tile's stores in flight
      char* dst = valid ? p.o.p + (b * p.o.sb + h * p.o.sh + tok * p.o.sn + DQ * g) * 2 : ea_trash_line();
      if constexpr (TileL<D>::NEWTR) {
        u32x4 o0, o1;
        quad_transpose_pack<E>(acc, pden, o0, o1);
        stg16(dst, o0);
        stg16(dst + 16, o1);
      } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) f[4 * dt + r] = acc[dt][r] * pden;
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
      }
      continue;
    }
    float fk2[DQ];
    if constexpr (TileL<D>::NEWTR) {
      quad_transpose_f32(acc, f);
      if (KEYS) quad_transpose_f32(acc2, fk2);
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { f[4 * dt + r] = acc[dt][r]; fk2[4 * dt + r] = acc2[dt][r]; }
    }
    if (!valid) continue;
    if (MODE == LX_BWDQ) {
#pragma unroll
      for (int j = 0; j < DQ; ++j) f[j] *= p.scale;
      char* dst = p.dq.p + (b * p.dq.sb + h * p.dq.sh + tok * p.dq.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
    } else if (MODE == LX_PBWDQ) {
      char* dst = p.dq.p + (b * p.dq.sb + h * p.dq.sh + tok * p.dq.sn + DQ * g) * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float qf8[8], o8[8];
        unpack8<E>(raw1[ks], qf8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = 8 * ks + i;
          o8[i] = p.scale * f[j] - p.knorm_coef * qf8[i] * sdb;
        }
        stg16(dst + ks * 16, pack8<E>(o8));
      }
    } else if (MODE == LX_QCORR) {
      char* dst = p.dq.p + (b * p.dq.sb + h * p.dq.sh + tok * p.dq.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) {
        float old[8];
        unpack8<E>(ldg16(dst + c * 16), old);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = 8 * c + i;
          old[i] -= f[j] * p.scale;
        }
        stg16(dst + c * 16, pack8<E>(old));
      }
    } else {   // LX_BWDK: dv = acc, dk = s (acc2 - k * sdb)
      char* dstv = p.dv.p + (b * p.dv.sb + h * p.dv.sh + tok * p.dv.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) stg16(dstv + c * 16, pack8<E>(f + 8 * c));
      char* dstk = p.dk.p + (b * p.dk.sb + h * p.dk.sh + tok * p.dk.sn + DQ * g) * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float kf[8], o8[8];
        unpack8<E>(raw1[ks], kf);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = 8 * ks + i;
          o8[i] = p.scale * fk2[j] - p.knorm_coef * kf[i] * sdb;
        }
        stg16(dstk + ks * 16, pack8<E>(o8));
      }
    }
    if (prof_it < 8) EA_STAMP(p, 7 + prof_it * 5);
    ++prof_it;
  }
  EA_STAMP(p, 60);
  EA_BLK(p, 1);
}

size_t lara_x_lds(int D, int NCT) {
  const int Cp = NCT * 16;
  return (size_t)3 * Cp * D * 2 + (size_t)3 * Cp * sizeof(float);
}

template <typename E, int D, int NCT>
static int launch_x(int mode, const LaraP& p, hipStream_t st) {
  const size_t lds = lara_x_lds(D, NCT);
  const dim3 grid((unsigned)(p.B * p.H * p.nsplit)), block(256);
#define EA_LXM(M, S)                                                                              \
  do {                                                                                            \
    if (lds > 64 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lara_x_kernel<E, D, NCT, M, S>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return (int)e;                                                         \
    }                                                                                             \
    hipLaunchKernelGGL((lara_x_kernel<E, D, NCT, M, S>), grid, block, lds, st, p);                \
  } while (0)
#define EA_LX(M) EA_LXM(M, -1)
  switch (mode) {
    case LX_FWD:
      if (p.mis == MIS_OPT) EA_LXM(LX_FWD, MIS_OPT);
      else if (p.mis == MIS_BIASED) EA_LXM(LX_FWD, MIS_BIASED);
      else EA_LXM(LX_FWD, MIS_BH);
      break;
    case LX_FWDM:
      if constexpr (NCT <= 4) {
        if (p.mis == MIS_OPT) EA_LXM(LX_FWDM, MIS_OPT);
        else if (p.mis == MIS_BIASED) EA_LXM(LX_FWDM, MIS_BIASED);
        else EA_LXM(LX_FWDM, MIS_BH);
      } else {
        return EA_E_UNSUPPORTED;
      }
      break;
    case LX_BWDQ: EA_LX(LX_BWDQ); break;
    case LX_BWDK: EA_LX(LX_BWDK); break;
    case LX_QCORR: EA_LX(LX_QCORR); break;
    case LX_POUT: EA_LX(LX_POUT); break;
    case LX_PBWDQ: EA_LX(LX_PBWDQ); break;
    case LX_PBWDK: EA_LX(LX_PBWDK); break;
    default: return EA_E_BADARG;
  }
#undef EA_LX
#undef EA_LXM
  return (int)hipGetLastError();
}

template <typename E, int D>
static int launch_x_nct(int mode, const LaraP& p, hipStream_t st) {
  if (p.NCT <= 2) return launch_x<E, D, 2>(mode, p, st);
  if (p.NCT <= 4) return launch_x<E, D, 4>(mode, p, st);
  if (p.NCT <= 8) return launch_x<E, D, 8>(mode, p, st);
  return EA_E_UNSUPPORTED;
}

int lara_x_dispatch(int mode, const LaraP& p0, int dtype, hipStream_t st) {
  LaraP p = p0;
  p.prof = nullptr;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "lara_x", mode);
#endif
  if (dtype == EA_BF16) {
    if (p.D == 64) return launch_x_nct<BF16, 64>(mode, p, st);
    if (p.D == 32) return launch_x_nct<BF16, 32>(mode, p, st);
  } else if (dtype == EA_F16) {
    if (p.D == 64) return launch_x_nct<F16, 64>(mode, p, st);
    if (p.D == 32) return launch_x_nct<F16, 32>(mode, p, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
