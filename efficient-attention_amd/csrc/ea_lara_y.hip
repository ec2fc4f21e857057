// ea_lara_y.hip -- LARA passes in the token-row layout (see ea_lara.h): everything that is a
// sum over the sequence for a fixed landmark sample c.
//
//   LY_FWD   kv_stats_c = sum_m softmax_m(log_proj_k[c,:]) v_m with its log-sum-exp (online
//            softmax, lara.py:205-211) and, for mis-opt, LSE_n(s qbar_c.q_n) (lara.py:222-223)
//   LY_BWDQ  d(kv_stats), d(omega) (query side), sum_n t dt q_n, sum_n t q_n and the scalar
//            sums r_c = sum_n dZ, d(bh)_c = sum_n dalpha, u_c = sum_n t dt
//   LY_BWDK  d(omega) (key side) = s sum_m dBk[c,m] k_m
// A workgroup streams its slice of the sequence through LDS in chunks; wave w owns landmark tile
// (w mod ncw) and token sub-chunk (w / ncw); scores are D[n = 4g+r][c = li] so the lane's values
// of two 16-token tiles are the B operand of the contraction over n, whose A operand is a
// ds_read_b64_tr_b16 of the token rows.  Every (split, sub-chunk) writes its own partial result;
// the caller merges partials (tiny tensors).
#include <type_traits>
#include "ea_lara.h"

namespace ea {

// landmark row -> MFMA B-operand fragment, in two steps so that the global loads of all three
// matrices (and of the first token chunk) are in flight together before the first conversion
template <int D> struct LmRaw { float4 v[D / 32][2]; };
template <int D>
EA_DEV void load_lm_raw(LmRaw<D>& dst, const float* src, bool ok, int g) {
  constexpr int KS = D / 32;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float* s = src + (g * KS + ks) * 8;
    dst.v[ks][0] = ok ? *reinterpret_cast<const float4*>(s) : make_float4(0.f, 0.f, 0.f, 0.f);
    dst.v[ks][1] = ok ? *reinterpret_cast<const float4*>(s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <typename E, int D>
EA_DEV void conv_lm_frag(typename E::x8* dst, const LmRaw<D>& raw) {
  constexpr int KS = D / 32;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float f[8] = {raw.v[ks][0].x, raw.v[ks][0].y, raw.v[ks][0].z, raw.v[ks][0].w,
                        raw.v[ks][1].x, raw.v[ks][1].y, raw.v[ks][1].z, raw.v[ks][1].w};
    dst[ks] = as_x8<E>(pack8<E>(f));
  }
}

// MIS >= 0: the estimator variant at compile time (the forward statistics pass; round 3 -- with mis and the phase flag
// `keys` as runtime values the chunk loop was 88 basic blocks); MIS = -1: read mis (the rare two-pass backward modes).
// token tiles [rows][D]: round-3 conflict-free layout for 128-byte rows (ea_common.h: phi2 swizzle, contiguous transpose
// reads -> accumulator tile dt of lane-row g holds channels 16 dt + 4 g ..), round-1 layout for D = 32
template <int D> struct YTile {
  static constexpr bool NEWTR = (D == 64);
  static EA_DEV int off(int row, int chunk16) {
    if constexpr (NEWTR) return lds_off2<D>(row, chunk16);
    else return lds_off<D>(row, chunk16);
  }
  // byte offset of the ds_read_b64_tr_b16 source of lane (li) in row `r`, channel tile dt
  static EA_DEV int tr(int r, int li, int dt) {
    const int colb = NEWTR ? (16 * dt + 4 * (li & 3)) * 2 : ((D / 4) * (li & 3) + 4 * dt) * 2;
    return off(r, colb >> 4) + (colb & 15);
  }
  static EA_DEV int chan(int dt, int g) { return NEWTR ? 16 * dt + 4 * g : (D / 4) * g + 4 * dt; }
};

template <typename E, int D, int MODE, int MIS>
__global__ __launch_bounds__(256, 2) void lara_y_kernel(const LaraP p) {
  const int mis = MIS >= 0 ? MIS : p.mis;
  constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
  constexpr int SW = CPR >= 8 ? 7 : CPR - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nsub = p.NCT == 1 ? 4 : (p.NCT == 2 ? 2 : 1);
  const int ncw = 4 / nsub;                          // landmark tiles handled concurrently
  constexpr int chunk = 128;                         // tokens staged per barrier pair (4 x 32-row blocks)
  constexpr int NSL = chunk * CPR / 256;             // staging slots per thread
  char* T1 = smem;
  char* T2 = T1 + chunk * ROWB;
  float* sc = reinterpret_cast<float*>(T2 + chunk * ROWB);   // [4][chunk] per-row scalars

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  // split-major block order: all first slices are dispatched before any second slice (see
  // lara_y_plan: with uneven slices the long ones must start first)
  const int nbh = p.B * p.H;
  const int split = blockIdx.x / nbh, bh = blockIdx.x - split * nbh;
  const int b = bh / p.H, h = bh - b * p.H;
  const int ct = blockIdx.y * 4 + (wave % ncw), sub = wave / ncw;
  const bool active = ct < p.NCT;
  const int c = ct * 16 + li;
  const bool c_ok = active && c < p.C;
  const size_t lm = (size_t)bh * p.C;
  const size_t lmw = (size_t)(p.w_per_head ? h : bh) * p.C;
  const float invC = 1.f / (float)p.C;
  constexpr bool PERF = MODE >= LY_PMAX;

  EA_STAMP(p, 0);
  EA_BLK(p, 0);
  typename E::x8 r1f[KS], r2f[KS], r3f[KS];
  LmRaw<D> raw1, raw2, raw3;
  load_lm_raw<D>(raw1, p.omega + (lmw + (c_ok ? c : 0)) * D, c_ok, g);
  const bool use_t = mis != MIS_BH && MODE != LY_BWDK && !PERF;
  load_lm_raw<D>(raw2, use_t ? p.qbar + (lm + (c_ok ? c : 0)) * D : p.omega, c_ok && use_t, g);
  const float* r3src = MODE == LY_BWDQ ? p.kv : p.dkv;
  constexpr bool HAS_R3 = MODE == LY_BWDQ || MODE == LY_BWDK;
  load_lm_raw<D>(raw3, HAS_R3 ? r3src + (lm + (c_ok ? c : 0)) * D : p.omega, c_ok && HAS_R3, g);
  const float stabk2 = (MODE == LY_PKV) ? (p.stab_per_feature ? (c_ok ? p.stab[lm + c] : 0.f) : p.stab[bh]) * LOG2E : 0.f;
  float cst2 = -INFINITY, lset2 = INFINITY, bhc = 1.f, lsek2 = INFINITY, dkkc = 0.f, rsc = 0.f;
  if (c_ok) {
    if (MODE == LY_BWDQ) {
      cst2 = p.cst[lm + c] * LOG2E;
      if (mis == MIS_OPT) { bhc = p.bhv[lm + c]; lset2 = p.lse_t[lm + c] * LOG2E; }
    }
    if (MODE == LY_BWDK) { lsek2 = p.lse_k[lm + c] * LOG2E; dkkc = p.dkk[lm + c]; rsc = p.rsum[lm + c]; }
  }

  const int n0 = p.tok_begin[split];
  const int n1 = p.tok_begin[split + 1];
  const int nphase = (MODE == LY_FWD && mis == MIS_OPT) ? 2 : 1;

  f32x4 acc0[DT], acc1[DT], acc2[DT], acc3[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc0[dt] = acc1[dt] = acc2[dt] = acc3[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_k = -INFINITY, l_k = 0.f, m_t = -INFINITY, l_t = 0.f;
  float s_r = 0.f, s_dbh = 0.f, s_u = 0.f;

  auto run_phase = [&](auto keys_tag, int phase) {
    constexpr bool keys = decltype(keys_tag)::value;
    const T4l& a1 = keys ? p.k : p.q;
    const T4l& a2 = keys ? p.v : p.dout;
    constexpr bool need2 = MODE == LY_FWD ? keys : MODE != LY_PMAX;
    const char* a1b = a1.p + (b * a1.sb + h * a1.sh) * 2;
    const char* a2b = need2 ? a2.p + (b * a2.sb + h * a2.sh) * 2 : nullptr;
    // Software pipeline: the rows of chunk i+1 (and their per-token scalars) are loaded into
    // registers while chunk i is being computed; the LDS image is refreshed between two barriers.
    u32x4 pw1[NSL], pw2[NSL];
    float ps0[NSL], ps1[NSL], ps2[NSL], ps3[NSL];
    auto issue = [&](int cb_) {
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / CPR, cc = idx - row * CPR;
        const bool valid = cb_ + row < n1;
        const int tok = valid ? cb_ + row : n1 - 1;       // clamped, not predicated: rows past the slice are zeroed by commit()
        pw2[i] = u32x4{0u, 0u, 0u, 0u};
        ps0[i] = ps1[i] = ps2[i] = ps3[i] = 0.f;
        pw1[i] = ldg16(a1b + (tok * a1.sn + cc * 8) * 2);
        if (need2) pw2[i] = ldg16(a2b + (tok * a2.sn + cc * 8) * 2);
        if (valid) {
          if (cc == 0 && (MODE == LY_BWDQ || MODE == LY_PBWDQ)) {
            const size_t o = (size_t)bh * p.N + tok;
            ps0[i] = p.lseZ[o]; ps1[i] = p.tmean[o]; ps2[i] = p.rowdot[o];
            if (MODE == LY_BWDQ) ps3[i] = p.sda[o];
          }
          if (cc == 0 && keys && (MODE != LY_PMAX || p.stab_per_feature)) ps0[i] = (p.mask && p.mask[(size_t)b * p.N + tok]) ? 1.f : 0.f;
        }
      }
    };
    auto commit = [&](int cb_) {
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / CPR, cc = idx - row * CPR;
        const bool valid = cb_ + row < n1;
        const u32x4 zz = {0u, 0u, 0u, 0u};
        if (!valid) pw1[i] = zz;
        sts16(T1 + YTile<D>::off(row, cc), pw1[i]);
        if (need2) sts16(T2 + YTile<D>::off(row, cc), valid ? pw2[i] : zz);
        if (keys || MODE == LY_PBWDQ) {
          float f[8], part = 0.f;
          unpack8<E>(pw1[i], f);
#pragma unroll
          for (int k = 0; k < 8; ++k) part += f[k] * f[k];
          part = group_sum<CPR>(part);
          if (cc == 0) {
            if (MODE == LY_PMAX && p.stab_per_feature) {
              sc[row] = (!valid || ps0[i] != 0.f) ? -INFINITY : -p.norm_coef2 * part;   // max of the whole log-feature
            } else if (MODE == LY_PMAX) {
              sc[row] = valid ? 0.f : -INFINITY;          // stabiliser: max over ALL keys, no diagonal term
            } else if (MODE == LY_PBWDQ) {
              sc[row] = valid ? -p.norm_coef2 * part - ps0[i] : -INFINITY;
              sc[chunk + row] = ps1[i];                   // 1 / clamp(den)
              sc[2 * chunk + row] = ps2[i];               // d den
            } else {
              const bool dead = !valid || ps0[i] != 0.f;
              sc[row] = dead ? -INFINITY : -p.norm_coef2 * part;
            }
          }
        } else if (MODE == LY_FWD) {
          if (cc == 0) sc[row] = valid ? 0.f : -INFINITY;
        } else if (cc == 0) {     // LY_BWDQ
          sc[row] = valid ? ps0[i] : INFINITY;
          sc[chunk + row] = ps1[i];
          sc[2 * chunk + row] = ps2[i];
          sc[3 * chunk + row] = ps3[i] * invC;
        }
      }
    };
    if (n0 < n1) issue(n0);
    if (phase == 0) {
      conv_lm_frag<E, D>(r1f, raw1);
      conv_lm_frag<E, D>(r2f, raw2);
      conv_lm_frag<E, D>(r3f, raw3);
    }
    EA_STAMP(p, 1);
    int prof_ci = 0;
    (void)prof_ci;
    for (int cb = n0; cb < n1; cb += chunk) {
      __syncthreads();                    // previous chunk's readers are done
      if (prof_ci < 5) EA_STAMP(p, 2 + prof_ci * 8);
      commit(cb);
      __syncthreads();
      if (prof_ci < 5) EA_STAMP(p, 3 + prof_ci * 8);
      if (cb + chunk < n1) issue(cb + chunk);
      if (prof_ci < 5) EA_STAMP(p, 4 + prof_ci * 8);
      if (!active) continue;
      for (int sb = sub; sb < chunk / 32; sb += nsub) {
      const int rb = sb * 32;
      if (cb + rb >= n1) break;
      float w0[2][4], w1v[2][4], w2v[2][4], w3v[2][4];
      float mloc = -INFINITY;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = rb + 16 * mt + li;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, s3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const typename E::x8 ta = as_x8<E>(lds16(T1 + YTile<D>::off(row, g * KS + ks)));
          if (keys || MODE == LY_BWDQ || MODE == LY_PBWDQ) s1 = E::mma(ta, r1f[ks], s1);
          if (!keys && use_t) s2 = E::mma(ta, r2f[ks], s2);
          if (HAS_R3) s3 = E::mma(as_x8<E>(lds16(T2 + YTile<D>::off(row, g * KS + ks))), r3f[ks], s3);
        }
        const int r0 = rb + 16 * mt + 4 * g;
        if (MODE == LY_BWDQ) {
          // VALU-bound stage: float2 arithmetic (v_pk_*_f32), per-row scalars as one float4 each,
          // alpha as a factor instead of log2(alpha) in the exponent, d(alpha) = 2^z (dW - rd)
          // without a division (same algebra as LX_BWDQ in ea_lara_x.hip).
          const float4 lz4 = *reinterpret_cast<const float4*>(sc + r0);
          const float4 tm4 = *reinterpret_cast<const float4*>(sc + chunk + r0);
          const float4 rd4 = *reinterpret_cast<const float4*>(sc + 2 * chunk + r0);
          const float4 sd4 = *reinterpret_cast<const float4*>(sc + 3 * chunk + r0);     // sda / C
          const f32x2 s22 = {p.scale_log2, p.scale_log2}, kap = {p.kappa, p.kappa};
          const f32x2 lzv[2] = {{lz4.x, lz4.y}, {lz4.z, lz4.w}}, tmv[2] = {{tm4.x, tm4.y}, {tm4.z, tm4.w}};
          const f32x2 rdv[2] = {{rd4.x, rd4.y}, {rd4.z, rd4.w}}, sdv[2] = {{sd4.x, sd4.y}, {sd4.z, sd4.w}};
          f32x2 sr2 = {0.f, 0.f}, sdbh2 = {0.f, 0.f}, su2 = {0.f, 0.f};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x2 S1 = {s1[2 * hh], s1[2 * hh + 1]};
            const f32x2 T2 = f32x2{s2[2 * hh], s2[2 * hh + 1]} * s22;
            f32x2 x = S1 * s22 + (f32x2{cst2, cst2} - lzv[hh]);
            f32x2 t = {0.f, 0.f}, al = {1.f, 1.f};
            if (mis == MIS_OPT) {
              const f32x2 tx = T2 - f32x2{lset2, lset2};
              t = f32x2{fast_exp2(tx[0]), fast_exp2(tx[1])};
              al = kap * t + (f32x2{bhc, bhc} - kap * tmv[hh]);
            } else if (mis == MIS_BIASED) {
              x += T2;
            }
            const f32x2 wz = {fast_exp2(x[0]), fast_exp2(x[1])};
            f32x2 w = wz;
            if (mis == MIS_OPT) w = wz * f32x2{fmaxf(al[0], 1e-8f), fmaxf(al[1], 1e-8f)};
            const f32x2 dd = f32x2{s3[2 * hh], s3[2 * hh + 1]} - rdv[hh];
            const f32x2 dz = w * dd;
            f32x2 da = {0.f, 0.f}, tdt = {0.f, 0.f};
            if (mis == MIS_OPT) {
              const f32x2 d0 = wz * dd;
              da = f32x2{al[0] > 1e-8f ? d0[0] : 0.f, al[1] > 1e-8f ? d0[1] : 0.f};
              tdt = t * kap * (da - sdv[hh]);
            }
            w0[mt][2 * hh] = w[0]; w0[mt][2 * hh + 1] = w[1];
            w1v[mt][2 * hh] = dz[0]; w1v[mt][2 * hh + 1] = dz[1];
            w2v[mt][2 * hh] = tdt[0]; w2v[mt][2 * hh + 1] = tdt[1];
            w3v[mt][2 * hh] = t[0]; w3v[mt][2 * hh + 1] = t[1];
            sr2 += dz; sdbh2 += da; su2 += tdt;
          }
          s_r += sr2[0] + sr2[1]; s_dbh += sdbh2[0] + sdbh2[1]; s_u += su2[0] + su2[1];
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (MODE == LY_FWD) {
            const float x = (keys ? s1[r] : s2[r]) * p.scale_log2 + sc[r0 + r];
            w0[mt][r] = c_ok ? x : -INFINITY;
            mloc = fmaxf(mloc, w0[mt][r]);
          } else if (MODE == LY_BWDK) {
            const float pk = fast_exp2(s1[r] * p.scale_log2 + sc[r0 + r] - lsek2);
            w0[mt][r] = pk * (s3[r] - dkkc + rsc);
          } else if (MODE == LY_PMAX) {
            if (c_ok) mloc = fmaxf(mloc, s1[r] * p.scale_log2 + sc[r0 + r]);
          } else if (MODE == LY_PKV) {
            const float sr = sc[r0 + r];
            const float phi = (c_ok && sr != -INFINITY) ? p.ratio * fast_exp2(s1[r] * p.scale_log2 + sr - stabk2) + p.feps : 0.f;
            w0[mt][r] = phi;
            s_r += phi;
          } else if (MODE == LY_PBWDQ) {
            const float sr = sc[r0 + r];
            const float phi = (c_ok && sr != -INFINITY) ? p.ratio * fast_exp2(s1[r] * p.scale_log2 + sr) + p.feps : 0.f;
            w0[mt][r] = phi * sc[chunk + r0 + r];
            s_r += phi * sc[2 * chunk + r0 + r];
          } else {   // LY_BWDQ
            const float lz2 = sc[r0 + r], tm = sc[chunk + r0 + r], rd = sc[2 * chunk + r0 + r], sd = sc[3 * chunk + r0 + r];
            const LaraElem e = lara_alpha(mis, s2[r] * p.scale_log2, lset2, bhc, p.kappa, tm);
            const float z2 = s1[r] * p.scale_log2 + e.la2 + cst2;
            const float w = fast_exp2(z2 - lz2);
            const float dz = w * (s3[r] - rd);
            float da = 0.f, tdt = 0.f;
            if (mis == MIS_OPT) {
              da = e.alpha > 1e-8f ? dz * fast_rcp(e.alpha) : 0.f;
              tdt = e.t * p.kappa * (da - sd);
            }
            w0[mt][r] = w; w1v[mt][r] = dz; w2v[mt][r] = tdt; w3v[mt][r] = e.t;
            s_r += dz; s_dbh += da; s_u += tdt;
          }
        }
        }
      }
      if (prof_ci < 5 && sb == sub) EA_STAMP(p, 5 + prof_ci * 8);
      // tr-read addressing of the two 16-token tiles of this wave
      const int ra = rb + 4 * g + (li >> 2), rb2 = ra + 16;
      if (MODE == LY_FWD) {
        mloc = quad_max(mloc);
        float& m = keys ? m_k : m_t;
        float& l = keys ? l_k : l_t;
        const float mn = fmaxf(m, mloc);
        const float ms = mn == -INFINITY ? 0.f : mn;
        const float alpha = fast_exp2(m - ms);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { w0[mt][r] = fast_exp2(w0[mt][r] - ms); ps += w0[mt][r]; }
        l = l * alpha + ps;
        if (keys) {
          u32x4 pf;
          pf[0] = pack2<E>(w0[0][0], w0[0][1]); pf[1] = pack2<E>(w0[0][2], w0[0][3]);
          pf[2] = pack2<E>(w0[1][0], w0[1][1]); pf[3] = pack2<E>(w0[1][2], w0[1][3]);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const u32x2 lo = E::tr4(T2 + YTile<D>::tr(ra, li, dt));
            const u32x2 hi = E::tr4(T2 + YTile<D>::tr(rb2, li, dt));
            acc0[dt] = acc0[dt] * alpha;
            acc0[dt] = E::mma(as_x8<E>(lo, hi), as_x8<E>(pf), acc0[dt]);
          }
        }
      } else if (MODE == LY_PMAX) {
        m_k = fmaxf(m_k, quad_max(mloc));
      } else {
        u32x4 f0, f1, f2, f3;
#define EA_PK(dst, src)                                                                     \
  dst[0] = pack2<E>(src[0][0], src[0][1]); dst[1] = pack2<E>(src[0][2], src[0][3]);         \
  dst[2] = pack2<E>(src[1][0], src[1][1]); dst[3] = pack2<E>(src[1][2], src[1][3]);
        EA_PK(f0, w0)
        if (MODE == LY_BWDQ) { EA_PK(f1, w1v) EA_PK(f2, w2v) EA_PK(f3, w3v) }
#undef EA_PK
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int oa = YTile<D>::tr(ra, li, dt);
          const int ob = YTile<D>::tr(rb2, li, dt);
          if (MODE == LY_BWDK) {
            acc0[dt] = E::mma(as_x8<E>(E::tr4(T1 + oa), E::tr4(T1 + ob)), as_x8<E>(f0), acc0[dt]);
          } else if (MODE == LY_PKV || MODE == LY_PBWDQ) {
            acc0[dt] = E::mma(as_x8<E>(E::tr4(T2 + oa), E::tr4(T2 + ob)), as_x8<E>(f0), acc0[dt]);
          } else {
            const typename E::x8 dot = as_x8<E>(E::tr4(T2 + oa), E::tr4(T2 + ob));
            const typename E::x8 qt = as_x8<E>(E::tr4(T1 + oa), E::tr4(T1 + ob));
            acc0[dt] = E::mma(dot, as_x8<E>(f0), acc0[dt]);      // d kv_stats
            acc1[dt] = E::mma(qt, as_x8<E>(f1), acc1[dt]);       // sum dZ q
            if (mis == MIS_OPT) {
              acc2[dt] = E::mma(qt, as_x8<E>(f2), acc2[dt]);     // sum t dt q
              acc3[dt] = E::mma(qt, as_x8<E>(f3), acc3[dt]);     // sum t q
            }
          }
        }
      }
      if (prof_ci < 5 && sb == sub) EA_STAMP(p, 6 + prof_ci * 8);
      }   // sub-blocks
      if (prof_ci < 5) EA_STAMP(p, 7 + prof_ci * 8);
      ++prof_ci;
    }
  };
  if constexpr (MODE == LY_FWD) {
    run_phase(std::true_type{}, 0);
    if (nphase == 2) run_phase(std::false_type{}, 1);
  } else if constexpr (MODE == LY_BWDK || MODE == LY_PMAX || MODE == LY_PKV) {
    run_phase(std::true_type{}, 0);
  } else {
    run_phase(std::false_type{}, 0);
  }
  EA_STAMP(p, 60);
  if (!c_ok) { EA_BLK(p, 1); return; }
  // ---- partial results of this (split, sub-chunk): lane (c, g) owns channels DQ*g .. ----
  const int S = p.nsplit * nsub;
  const size_t slot = ((size_t)bh * S + split * nsub + sub) * p.C + c;
  float* ml = p.p_ml + slot * 4;
  if (MODE == LY_FWD) {
    l_k = quad_sum(l_k);
    l_t = quad_sum(l_t);
    if (g == 0) { ml[0] = m_k * LN2; ml[1] = l_k; ml[2] = m_t * LN2; ml[3] = l_t; }
  } else if (MODE == LY_BWDQ) {
    s_r = quad_sum(s_r); s_dbh = quad_sum(s_dbh); s_u = quad_sum(s_u);
    if (g == 0) { ml[0] = s_r; ml[1] = s_dbh; ml[2] = s_u; ml[3] = 0.f; }
  } else if (MODE == LY_PMAX) {
    if (g == 0) { ml[0] = m_k * LN2; ml[1] = 0.f; ml[2] = 0.f; ml[3] = 0.f; }
    EA_BLK(p, 1);
    return;
  } else if (PERF) {
    s_r = quad_sum(s_r);
    if (g == 0) { ml[0] = s_r; ml[1] = 0.f; ml[2] = 0.f; ml[3] = 0.f; }
  }
  auto put = [&](float* base, const f32x4* a) {
    float* d = base + slot * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<float4*>(d + YTile<D>::chan(dt, g)) = make_float4(a[dt][0], a[dt][1], a[dt][2], a[dt][3]);
  };
  put(p.p_acc0, acc0);
  if (MODE == LY_BWDQ) {
    put(p.p_acc1, acc1);
    if (mis == MIS_OPT) { put(p.p_acc2, acc2); put(p.p_acc3, acc3); }
  }
  EA_STAMP(p, 61);
  EA_BLK(p, 1);
}

// Slice boundaries of the sequence for one token-row pass.  Slices are equal unless the 2 BH
// workgroups of a two-slice launch do not fit on the chip at once (BH < slots < 2 BH): then the
// second "round" would run on a mostly idle chip.  With split-major dispatch order the BH first
// slices start immediately next to slots - BH second slices; the remaining second slices follow
// in rb = ceil(BH / (slots - BH)) rounds, so first : second = rb : 1 makes everything end together
// (e.g. B*h = 384 on 256 CUs x 2 resident workgroups: 640 + 144 tokens instead of 392 + 392).
static void lara_y_plan(LaraP& p, int slots) {
  const int BH = p.B * p.H, k = p.nsplit, gran = 16;
  const int tpb = ((p.N + k - 1) / k + gran - 1) / gran * gran;
  for (int i = 0; i <= k; ++i) p.tok_begin[i] = i * tpb < p.N ? i * tpb : p.N;
  p.tok_begin[k] = p.N;
  if (k == 2 && BH < slots && slots < 2 * BH) {
    // a workgroup costs (tokens + F) token-times, F = its fixed prologue / epilogue (measured on the
    // backward statistics pass: ~7.5 us against 0.066 us per token): first + F = rb (second + F)
    const int rb = (BH + (slots - BH) - 1) / (slots - BH);
    const int F = 112;
    int b = (p.N - (rb - 1) * F) / (1 + rb) / gran * gran;
    if (b >= 2 * gran) p.tok_begin[1] = p.N - b;
  }
}

static int device_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <typename E, int D, int MODE, int MIS = -1>
static int launch_y_mode(LaraP& p, hipStream_t st) {
  const size_t lds = (size_t)2 * 128 * D * 2 + (size_t)4 * 128 * sizeof(float);
  static int occ = 0;                       // resident workgroups per CU of this instantiation
  if (!occ) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&lara_y_kernel<E, D, MODE, MIS>), 256, lds) != hipSuccess || n <= 0) n = 2;
    occ = n;
  }
  lara_y_plan(p, occ * device_cus());
  const dim3 grid((unsigned)(p.B * p.H * p.nsplit), (unsigned)((p.NCT + 3) / 4)), block(256);
  hipLaunchKernelGGL((lara_y_kernel<E, D, MODE, MIS>), grid, block, lds, st, p);
  return (int)hipGetLastError();
}

template <typename E, int D>
static int launch_y(int mode, LaraP& p, hipStream_t st) {
  switch (mode) {
    case LY_FWD:
      if (p.mis == MIS_OPT) return launch_y_mode<E, D, LY_FWD, MIS_OPT>(p, st);
      if (p.mis == MIS_BIASED) return launch_y_mode<E, D, LY_FWD, MIS_BIASED>(p, st);
      return launch_y_mode<E, D, LY_FWD, MIS_BH>(p, st);
    case LY_BWDQ: return launch_y_mode<E, D, LY_BWDQ>(p, st);
    case LY_BWDK: return launch_y_mode<E, D, LY_BWDK>(p, st);
    case LY_PMAX: return launch_y_mode<E, D, LY_PMAX>(p, st);
    case LY_PKV: return launch_y_mode<E, D, LY_PKV>(p, st);
    case LY_PBWDQ: return launch_y_mode<E, D, LY_PBWDQ>(p, st);
    default: return EA_E_BADARG;
  }
}

int lara_y_dispatch(int mode, const LaraP& p0, int dtype, hipStream_t st) {
  LaraP p = p0;
  p.prof = nullptr;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "lara_y", mode);
#endif
  if (dtype == EA_BF16) {
    if (p.D == 64) return launch_y<BF16, 64>(mode, p, st);
    if (p.D == 32) return launch_y<BF16, 32>(mode, p, st);
  } else if (dtype == EA_F16) {
    if (p.D == 64) return launch_y<F16, 64>(mode, p, st);
    if (p.D == 32) return launch_y<F16, 32>(mode, p, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
