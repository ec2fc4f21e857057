// ea_softmax.hip -- the softmax baseline (abstract_attention.py:120-133) as streaming
// (flash-style) kernels: softmax(s QK^T, -inf on padded keys) V without materialising the
// [N, N] score matrix.  This is the one MFMA-bound member of the family (AI ~ 0.6 N FLOP/B).
//
//   fwd     : workgroup = 64 queries (a wave per 16-query tile), K/V streamed through LDS in
//             64-key chunks, S^T = K Q^T tiles -> online softmax in registers -> O^T = V^T P^T
//             (same register-level chaining as ea_window_fwd.hip); writes out and lse
//   bwd_dq  : same streaming; P from the saved lse, dS^T = P o (V dO^T - delta) -> dQ^T = K^T dS^T;
//             also writes delta_n = dO_n . O_n for the second pass
//   bwd_dkv : workgroup = 64 keys (a wave per 16-key tile), Q/dO streamed through LDS in 64-query
//             chunks, S = Q K^T, dP = dO V^T tiles [query][key] -> dV^T = dO^T P, dK^T = Q^T dS
// Two recomputing passes instead of fp32 atomics on dQ: deterministic, and no read-modify-write
// traffic on the gradient.
#include <stdlib.h>
#include <type_traits>
#include "ea_softmax.h"

namespace ea {


// DR: attention dropout from an explicit keep mask -- dropped entries leave the numerator (and dP, and
// the P of dV in the backward); the normaliser is that of the full row.
// -0.5 s log2(e) |k|^2 of the key row a staging thread holds one 16-B chunk of (CPR adjacent lanes)
template <typename E, int CPR> EA_DEV float key_norm_term(u32x4 kw, float scale_log2) {
  float f[8], part = 0.f;
  unpack8<E>(kw, f);
#pragma unroll
  for (int j = 0; j < 8; ++j) part += f[j] * f[j];
  part = group_sum<CPR>(part);
  return -0.5f * scale_log2 * part;
}

// KB: per-key bias -s |k|^2 / 2 in the logits (p.key_norm_bias)
// QT: 16-query tiles per wave (round 3).  The K / V fragments a wave reads from LDS feed QT MFMAs instead of one, and a
// chunk's two barriers and its staging are amortised over QT times the work.  Padded / out-of-range keys enter the logits
// as an additive -inf from LDS (kadd_s, together with the KB term) -- one fma per score instead of a byte unpack, a
// compare and a select: the forward is bound by its VALU work (v_exp_f32 alone takes as long as the two MFMAs a score
// costs), not by the matrix cores.
template <typename E, int D, bool DR, bool KB, int QT>
__global__ __launch_bounds__(256, 2) void sm_fwd_kernel(const SmP p) {
  constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
  __shared__ __attribute__((aligned(16))) char Ks[64 * ROWB];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ROWB];
  __shared__ __attribute__((aligned(16))) float kadd_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  constexpr int QPB = 64 * QT;                       // queries per workgroup
  const int nqb = (p.N + QPB - 1) / QPB;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const char* qbase = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* kbase = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const char* vbase = p.v.p + (b * p.v.sb + h * p.v.sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.N : nullptr;
  int qtok[QT];
  bool qvalid[QT];
  typename E::x8 qf[QT][KS];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    qtok[u] = qb * QPB + (wave * QT + u) * 16 + li;
    qvalid[u] = qtok[u] < p.N;
    const int tq = min(qtok[u], p.N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 w = ldg16(qbase + (tq * p.q.sn + (g * KS + ks) * 8) * 2);
      qf[u][ks] = as_x8<E>(qvalid[u] ? w : u32x4{0u, 0u, 0u, 0u});
    }
  }
  float m[QT], lsum[QT];
  f32x4 o[QT][DT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m[u] = -INFINITY; lsum[u] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the next chunk's K/V rows are in flight (registers) while this chunk computes; loads are unconditional from
  // clamped rows (no exec-mask branches), rows past the sequence are zeroed when they are committed to LDS.
  constexpr int NSL = (64 * CPR + 255) / 256;
  u32x4 nk[NSL], nv[NSL];
  uint8_t nm[NSL];
  auto issue = [&](int kc_) {
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, c = idx - row * CPR;
      const int tok = min(kc_ + row, p.N - 1);
      nk[i] = ldg16(kbase + (tok * p.k.sn + c * 8) * 2);
      nv[i] = ldg16(vbase + (tok * p.v.sn + c * 8) * 2);
      nm[i] = mrow ? mrow[tok] : (uint8_t)0;
    }
  };
  // One 64-key chunk.  PL (a template tag, so that the loop body carries no branch: with `plain` tested inside it the
  // compiler kept the S tiles in one set of registers on both arms and copied them out -- 64 v_mov per chunk): a chunk
  // without padded / masked keys and without the KB term (the common case: no mask, N a multiple of 64) needs no additive
  // term: the row maximum is taken over the raw scores and the scale is applied together with the shift,
  // exp2(s * scale - m), one fma per score instead of an fma and a subtraction.
  auto chunk = [&](auto plain_tag, int kc) {
    constexpr bool PL = decltype(plain_tag)::value;
    f32x4 s[QT][4];
    float mloc[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) mloc[u] = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int row = tt * 16 + li;
#pragma unroll
      for (int u = 0; u < QT; ++u) s[u][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const typename E::x8 kfr = as_x8<E>(lds16(Ks + TileL<D>::off(row, g * KS + ks)));
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][tt] = E::mma(kfr, qf[u][ks], s[u][tt]);
      }
      if constexpr (PL) {
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) mloc[u] = fmaxf(mloc[u], s[u][tt][r]);
      } else {
        const float4 ka4 = *reinterpret_cast<const float4*>(kadd_s + tt * 16 + 4 * g);
        const float kav[4] = {ka4.x, ka4.y, ka4.z, ka4.w};
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = fmaf(s[u][tt][r], p.scale_log2, kav[r]);
            s[u][tt][r] = x;
            mloc[u] = fmaxf(mloc[u], x);
          }
      }
    }
    uint32_t pw[QT][4][2];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float ml = quad_max(mloc[u]);
      if (PL) ml *= p.scale_log2;                        // (scale > 0: the maximum commutes with it)
      const float mnew = fmaxf(m[u], ml);
      const float msafe = mnew == -INFINITY ? 0.f : mnew;
      // the running maximum moves in the first few chunks of a row and then hardly ever: the rescale of the 16
      // accumulators is skipped when no lane of the wave needs it -- alpha is exactly 1 there
      const bool moved = mnew != m[u];
      const float alpha = fast_exp2(m[u] - msafe);
      m[u] = mnew;
      // (two scores per instruction: the shift-and-scale as v_pk_fma_f32, the row sum as v_pk_add_f32 -- the forward is
      //  bound by its VALU issue, exp2 alone taking 4 of the ~8 issue slots a score costs against 4 for its two MFMAs)
      f32x2 psum2 = {0.f, 0.f};
      const uint8_t* krow = DR ? p.keep + ((size_t)bh * p.N + (qvalid[u] ? qtok[u] : 0)) * p.keep_ld + kc + 4 * g : nullptr;
      const float sc = PL ? p.scale_log2 : 1.f;
      const f32x2 sc2 = {sc, sc}, nm2 = {-msafe, -msafe};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        uint32_t k4 = 0;
        if (DR) k4 = *reinterpret_cast<const uint32_t*>(krow + tt * 16);
        const f32x2 xa = f32x2{s[u][tt][0], s[u][tt][1]} * sc2 + nm2, xb = f32x2{s[u][tt][2], s[u][tt][3]} * sc2 + nm2;
        f32x2 pa = {fast_exp2(xa[0]), fast_exp2(xa[1])}, pb = {fast_exp2(xb[0]), fast_exp2(xb[1])};
        psum2 += pa;
        psum2 += pb;
        if (DR) {
          pa[0] = (k4 & 0xffu) ? pa[0] * p.keep_scale : 0.f;
          pa[1] = ((k4 >> 8) & 0xffu) ? pa[1] * p.keep_scale : 0.f;
          pb[0] = ((k4 >> 16) & 0xffu) ? pb[0] * p.keep_scale : 0.f;
          pb[1] = ((k4 >> 24) & 0xffu) ? pb[1] * p.keep_scale : 0.f;
        }
        pw[u][tt][0] = pack2<E>(pa[0], pa[1]);
        pw[u][tt][1] = pack2<E>(pb[0], pb[1]);
      }
      const float psum = psum2[0] + psum2[1];
      if (__any(moved)) {
        lsum[u] = lsum[u] * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[u][dt] *= alpha;
      } else {
        lsum[u] += psum;
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      typename E::x8 pf[QT];
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        u32x4 pf4;
        pf4[0] = pw[u][2 * kk][0]; pf4[1] = pw[u][2 * kk][1]; pf4[2] = pw[u][2 * kk + 1][0]; pf4[3] = pw[u][2 * kk + 1][1];
        pf[u] = as_x8<E>(pf4);
      }
      const int r0 = 32 * kk + 4 * g + (li >> 2), r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const u32x2 lo = E::tr4(Vs + tile_tr<D>(r0, li, dt));
        const u32x2 hi = E::tr4(Vs + tile_tr<D>(r1, li, dt));
        const typename E::x8 vfr = as_x8<E>(lo, hi);
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][dt] = E::mma(vfr, pf[u], o[u][dt]);
      }
    }
  };
  // The chunk loop, once per variant of the body (two loops one after the other, not two arms inside one loop: values live
  // across both arms cost registers the kernel does not have): the full unmasked chunks first, then whatever is left.
  auto run = [&](auto plain_tag, int kc0, int kc1) {
    constexpr bool PL = decltype(plain_tag)::value;
    for (int kc = kc0; kc < kc1; kc += 64) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NSL; ++i) {
        const int idx = tid + i * 256;
        const int row = idx / CPR, c = idx - row * CPR;
        if (64 * CPR % 256 != 0 && idx >= 64 * CPR) continue;
        const bool in = PL || kc + row < p.N;
        u32x4 kw = nk[i], vw = nv[i];
        if (!PL) {
          const u32x4 z = {0u, 0u, 0u, 0u};
          kw = in ? kw : z; vw = in ? vw : z;
        }
        sts16(Ks + TileL<D>::off(row, c), kw);
        sts16(Vs + TileL<D>::off(row, c), vw);
        if (!PL) {
          float kn = 0.f;
          if (KB) kn = key_norm_term<E, CPR>(kw, p.scale_log2);
          if (c == 0) kadd_s[row] = (!in || nm[i]) ? -INFINITY : kn;
        }
      }
      __syncthreads();
      if (kc + 64 < p.N) issue(kc + 64);
      chunk(plain_tag, kc);
    }
  };
  const int nfull = (!KB && !mrow) ? (p.N / 64) * 64 : 0;
  issue(0);
  run(std::true_type{}, 0, nfull);
  run(std::false_type{}, nfull, p.N);
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const float ltot = quad_sum(lsum[u]);
    const float inv = fast_rcp(ltot);
    float f[DQ];
    if constexpr (TileL<D>::NEWTR) {
      quad_transpose_f32(o[u], f);                // accumulator pieces -> the lane's contiguous channels (all lanes)
#pragma unroll
      for (int j = 0; j < DQ; ++j) f[j] *= inv;
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) f[4 * dt + r] = o[u][dt][r] * inv;
    }
    if (qvalid[u]) {
      char* dst = p.o.p + (b * p.o.sb + h * p.o.sh + qtok[u] * p.o.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
      if (g == 0) p.lse[(size_t)bh * p.N + qtok[u]] = (m[u] + fast_log2(ltot)) * LN2;
    }
  }
}

// QT: 16-query tiles per wave (see sm_fwd_kernel)
template <typename E, int D, bool DR, bool KB, int QT>
__global__ __launch_bounds__(256) void sm_bwd_dq_kernel(const SmP p) {
  constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
  __shared__ __attribute__((aligned(16))) char Ks[64 * ROWB];
  __shared__ __attribute__((aligned(16))) char Vs[64 * ROWB];
  __shared__ __attribute__((aligned(16))) float kadd_s[64];     // -inf on padded / out-of-range keys, else the KB term
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  constexpr int QPB = 64 * QT;
  const int nqb = (p.N + QPB - 1) / QPB;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const char* kbase = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const char* vbase = p.v.p + (b * p.v.sb + h * p.v.sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.N : nullptr;
  int qtok[QT];
  bool qvalid[QT];
  typename E::x8 qf[QT][KS], dof[QT][KS];
  float delta[QT], lse2[QT];
  f32x4 dq[QT][DT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    qtok[u] = qb * QPB + (wave * QT + u) * 16 + li;
    qvalid[u] = qtok[u] < p.N;
    const int tq = min(qtok[u], p.N - 1);
    delta[u] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int eo = (g * KS + ks) * 8;
      const u32x4 z = {0u, 0u, 0u, 0u};
      u32x4 w = ldg16(p.q.p + (b * p.q.sb + h * p.q.sh + tq * p.q.sn + eo) * 2);
      u32x4 dw = ldg16(p.dout.p + (b * p.dout.sb + h * p.dout.sh + tq * p.dout.sn + eo) * 2);
      u32x4 ow = ldg16(p.o.p + (b * p.o.sb + h * p.o.sh + tq * p.o.sn + eo) * 2);
      w = qvalid[u] ? w : z; dw = qvalid[u] ? dw : z; ow = qvalid[u] ? ow : z;
      qf[u][ks] = as_x8<E>(w);
      dof[u][ks] = as_x8<E>(dw);
      float a8[8], c8[8];
      unpack8<E>(dw, a8);
      unpack8<E>(ow, c8);
#pragma unroll
      for (int i = 0; i < 8; ++i) delta[u] += a8[i] * c8[i];
    }
    delta[u] = quad_sum(delta[u]);
    lse2[u] = qvalid[u] ? p.lse[(size_t)bh * p.N + qtok[u]] * LOG2E : INFINITY;
    if (qvalid[u] && g == 0) p.delta[(size_t)bh * p.N + qtok[u]] = delta[u];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // the next chunk's K/V rows are in flight (registers) while this chunk computes; loads are unconditional from
  // clamped rows (no exec-mask branches), rows past the sequence are zeroed when they are committed to LDS.
  constexpr int NSL = (64 * CPR + 255) / 256;
  u32x4 nk[NSL], nv[NSL];
  uint8_t nm[NSL];
  auto issue = [&](int kc_) {
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, c = idx - row * CPR;
      const int tok = min(kc_ + row, p.N - 1);
      nk[i] = ldg16(kbase + (tok * p.k.sn + c * 8) * 2);
      nv[i] = ldg16(vbase + (tok * p.v.sn + c * 8) * 2);
      nm[i] = mrow ? mrow[tok] : (uint8_t)0;
    }
  };
  issue(0);
  for (int kc = 0; kc < p.N; kc += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, c = idx - row * CPR;
      if (64 * CPR % 256 != 0 && idx >= 64 * CPR) continue;
      const bool in = kc + row < p.N;
      const u32x4 z = {0u, 0u, 0u, 0u};
      const u32x4 kw = in ? nk[i] : z, vw = in ? nv[i] : z;
      sts16(Ks + TileL<D>::off(row, c), kw);
      sts16(Vs + TileL<D>::off(row, c), vw);
      float kn = 0.f;
      if (KB) kn = key_norm_term<E, CPR>(kw, p.scale_log2);
      if (c == 0) kadd_s[row] = (!in || nm[i]) ? -INFINITY : kn;
    }
    __syncthreads();
    if (kc + 64 < p.N) issue(kc + 64);
    const bool plain = !KB && !mrow && kc + 64 <= p.N;      // (uniform) no additive term in this chunk
    uint32_t dsw[QT][4][2];
    // (two straight-line instances of the chunk body: written as one body with `plain ? a : b` hipcc evaluates both
    //  arguments of every score and selects)
    auto score_tiles = [&](auto plain_tag) {
      constexpr bool PL = decltype(plain_tag)::value;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        f32x4 s[QT], dp[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) { s[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int row = tt * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const typename E::x8 kfr = as_x8<E>(lds16(Ks + TileL<D>::off(row, g * KS + ks)));
          const typename E::x8 vfr = as_x8<E>(lds16(Vs + TileL<D>::off(row, g * KS + ks)));
#pragma unroll
          for (int u = 0; u < QT; ++u) {
            s[u] = E::mma(kfr, qf[u][ks], s[u]);
            dp[u] = E::mma(vfr, dof[u][ks], dp[u]);
          }
        }
        float kav[4] = {0.f, 0.f, 0.f, 0.f};
        if (!PL) {
          const float4 ka4 = *reinterpret_cast<const float4*>(kadd_s + tt * 16 + 4 * g);
          kav[0] = ka4.x; kav[1] = ka4.y; kav[2] = ka4.z; kav[3] = ka4.w;
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          uint32_t k4 = 0;
          if (DR) k4 = *reinterpret_cast<const uint32_t*>(
                      p.keep + ((size_t)bh * p.N + (qvalid[u] ? qtok[u] : 0)) * p.keep_ld + kc + tt * 16 + 4 * g);
          float ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (lse2 = +inf on rows past the sequence, kav = -inf on dead keys: either way exp2(-inf) = 0)
            const float pr = PL ? fast_exp2(fmaf(s[u][r], p.scale_log2, -lse2[u]))
                                : fast_exp2(fmaf(s[u][r], p.scale_log2, kav[r]) - lse2[u]);
            float dpr = dp[u][r];
            if (DR) dpr = ((k4 >> (8 * r)) & 0xffu) ? dpr * p.keep_scale : 0.f;
            ds[r] = pr * (dpr - delta[u]);
          }
          dsw[u][tt][0] = pack2<E>(ds[0], ds[1]);
          dsw[u][tt][1] = pack2<E>(ds[2], ds[3]);
        }
      }
    };
    if (plain) score_tiles(std::true_type{}); else score_tiles(std::false_type{});
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      typename E::x8 dsf[QT];
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        u32x4 f4v;
        f4v[0] = dsw[u][2 * kk][0]; f4v[1] = dsw[u][2 * kk][1]; f4v[2] = dsw[u][2 * kk + 1][0]; f4v[3] = dsw[u][2 * kk + 1][1];
        dsf[u] = as_x8<E>(f4v);
      }
      const int r0 = 32 * kk + 4 * g + (li >> 2), r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const u32x2 lo = E::tr4(Ks + tile_tr<D>(r0, li, dt));
        const u32x2 hi = E::tr4(Ks + tile_tr<D>(r1, li, dt));
        const typename E::x8 kt = as_x8<E>(lo, hi);
#pragma unroll
        for (int u = 0; u < QT; ++u) dq[u][dt] = E::mma(kt, dsf[u], dq[u][dt]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    float f[DQ];
    if constexpr (TileL<D>::NEWTR) {
      quad_transpose_f32(dq[u], f);
#pragma unroll
      for (int j = 0; j < DQ; ++j) f[j] *= p.scale;
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) f[4 * dt + r] = dq[u][dt][r] * p.scale;
    }
    if (qvalid[u]) {
      char* dst = p.dq.p + (b * p.dq.sb + h * p.dq.sh + qtok[u] * p.dq.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
    }
  }
}

// KT: 16-key tiles per wave: the Q / dO fragments a wave reads from LDS (plain and transposed) feed KT MFMAs each
template <typename E, int D, bool DR, bool KB, int KT>
__global__ __launch_bounds__(256) void sm_bwd_dkv_kernel(const SmP p) {
  constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
  __shared__ __attribute__((aligned(16))) char Qs[64 * ROWB];
  __shared__ __attribute__((aligned(16))) char dOs[64 * ROWB];
  __shared__ __attribute__((aligned(16))) float lse_s[64];
  __shared__ __attribute__((aligned(16))) float delta_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  constexpr int KPB = 64 * KT;                       // keys per workgroup
  const int nkb = (p.N + KPB - 1) / KPB;
  const int bh = blockIdx.x / nkb, kb = blockIdx.x - bh * nkb;
  const int b = bh / p.H, h = bh - b * p.H;
  const char* qbase = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* dobase = p.dout.p + (b * p.dout.sb + h * p.dout.sh) * 2;
  int ktok[KT];
  bool kvalid[KT], kdead[KT];
  typename E::x8 kf[KT][KS], vf[KT][KS];
  f32x4 dk[KT][DT], dv[KT][DT];
  // KB: this lane's key bias (its key row is spread over the four lane groups) and the running
  // column sum of dS, which is the gradient of that bias
  float kbias[KT], dscol[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    ktok[u] = kb * KPB + (wave * KT + u) * 16 + li;
    kvalid[u] = ktok[u] < p.N;
    const int tk = min(ktok[u], p.N - 1);
    kdead[u] = !kvalid[u] || (p.mask && p.mask[(size_t)b * p.N + tk]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int eo = (g * KS + ks) * 8;
      const u32x4 z = {0u, 0u, 0u, 0u};
      const u32x4 kw = ldg16(p.k.p + (b * p.k.sb + h * p.k.sh + tk * p.k.sn + eo) * 2);
      const u32x4 vw = ldg16(p.v.p + (b * p.v.sb + h * p.v.sh + tk * p.v.sn + eo) * 2);
      kf[u][ks] = as_x8<E>(kvalid[u] ? kw : z);
      vf[u][ks] = as_x8<E>(kvalid[u] ? vw : z);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    kbias[u] = 0.f; dscol[u] = 0.f;
    if (KB) {
      float part = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float f[8];
        unpack8<E>(__builtin_bit_cast(u32x4, kf[u][ks]), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += f[j] * f[j];
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      kbias[u] = -0.5f * p.scale_log2 * part;
    }
  }

  constexpr int NSL = (64 * CPR + 255) / 256;
  u32x4 nq[NSL], nd[NSL];
  float nl[NSL], ndl[NSL];
  auto issue = [&](int qc_) {
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, c = idx - row * CPR;
      const int tok = min(qc_ + row, p.N - 1);
      nq[i] = ldg16(qbase + (tok * p.q.sn + c * 8) * 2);
      nd[i] = ldg16(dobase + (tok * p.dout.sn + c * 8) * 2);
      nl[i] = p.lse[(size_t)bh * p.N + tok];
      ndl[i] = p.delta[(size_t)bh * p.N + tok];
    }
  };
  issue(0);
  for (int qc = 0; qc < p.N; qc += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, c = idx - row * CPR;
      if (64 * CPR % 256 != 0 && idx >= 64 * CPR) continue;
      const bool in = qc + row < p.N;
      const u32x4 z = {0u, 0u, 0u, 0u};
      sts16(Qs + TileL<D>::off(row, c), in ? nq[i] : z);
      sts16(dOs + TileL<D>::off(row, c), in ? nd[i] : z);
      if (c == 0) {
        lse_s[row] = in ? -nl[i] * LOG2E : -INFINITY;       // MINUS the log2-domain lse: added inside the score fma
        delta_s[row] = in ? ndl[i] : 0.f;
      }
    }
    __syncthreads();
    if (qc + 64 < p.N) issue(qc + 64);
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      uint32_t pw[KT][2][2], dsw[KT][2][2];
#pragma unroll
      for (int uq = 0; uq < 2; ++uq) {
        const int rq = (2 * qq + uq) * 16;
        f32x4 s[KT], dp[KT];
#pragma unroll
        for (int u = 0; u < KT; ++u) { s[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const typename E::x8 qfr = as_x8<E>(lds16(Qs + TileL<D>::off(rq + li, g * KS + ks)));
          const typename E::x8 dfr = as_x8<E>(lds16(dOs + TileL<D>::off(rq + li, g * KS + ks)));
#pragma unroll
          for (int u = 0; u < KT; ++u) {
            s[u] = E::mma(qfr, kf[u][ks], s[u]);
            dp[u] = E::mma(dfr, vf[u][ks], dp[u]);
          }
        }
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + rq + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + rq + 4 * g);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int u = 0; u < KT; ++u) {
          float pr[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // a dead key's column is computed like any other and zeroed in the epilogue (columns are independent)
            pr[r] = KB ? fast_exp2(fmaf(s[u][r], p.scale_log2, kbias[u]) + ll[r]) : fast_exp2(fmaf(s[u][r], p.scale_log2, ll[r]));
            float dpr = dp[u][r];
            float km = 1.f;
            if (DR) {
              const int qt = min(qc + rq + 4 * g + r, p.N - 1);
              km = p.keep[((size_t)bh * p.N + qt) * p.keep_ld + (kvalid[u] ? ktok[u] : 0)] ? p.keep_scale : 0.f;
              dpr *= km;
            }
            ds[r] = pr[r] * (dpr - dd[r]);
            if (KB) dscol[u] += ds[r];
            if (DR) pr[r] *= km;
          }
          pw[u][uq][0] = pack2<E>(pr[0], pr[1]); pw[u][uq][1] = pack2<E>(pr[2], pr[3]);
          dsw[u][uq][0] = pack2<E>(ds[0], ds[1]); dsw[u][uq][1] = pack2<E>(ds[2], ds[3]);
        }
      }
      typename E::x8 pf[KT], dsf[KT];
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        u32x4 a4, b4;
        a4[0] = pw[u][0][0]; a4[1] = pw[u][0][1]; a4[2] = pw[u][1][0]; a4[3] = pw[u][1][1];
        b4[0] = dsw[u][0][0]; b4[1] = dsw[u][0][1]; b4[2] = dsw[u][1][0]; b4[3] = dsw[u][1][1];
        pf[u] = as_x8<E>(a4); dsf[u] = as_x8<E>(b4);
      }
      const int r0 = 32 * qq + 4 * g + (li >> 2), r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int o0 = tile_tr<D>(r0, li, dt);
        const int o1 = tile_tr<D>(r1, li, dt);
        const typename E::x8 dot = as_x8<E>(E::tr4(dOs + o0), E::tr4(dOs + o1));
        const typename E::x8 qt_ = as_x8<E>(E::tr4(Qs + o0), E::tr4(Qs + o1));
#pragma unroll
        for (int u = 0; u < KT; ++u) {
          dv[u][dt] = E::mma(dot, pf[u], dv[u][dt]);
          dk[u][dt] = E::mma(qt_, dsf[u], dk[u][dt]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    if (KB) {
      dscol[u] += __shfl_xor(dscol[u], 16);
      dscol[u] += __shfl_xor(dscol[u], 32);
    }
    float fk[DQ], fv[DQ];
    if constexpr (TileL<D>::NEWTR) {
      quad_transpose_f32(dk[u], fk);
      quad_transpose_f32(dv[u], fv);
#pragma unroll
      for (int j = 0; j < DQ; ++j) fk[j] *= p.scale;
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { fk[4 * dt + r] = dk[u][dt][r] * p.scale; fv[4 * dt + r] = dv[u][dt][r]; }
    }
    if (kdead[u]) {
#pragma unroll
      for (int j = 0; j < DQ; ++j) fk[j] = fv[j] = 0.f;
      dscol[u] = 0.f;
    }
    if (kvalid[u]) {
      if (KB) {
        // d/dk_j of -s |k_j|^2 / 2 summed over the queries: -s k_j sum_i dS_ij (this lane's channels)
        const char* krow = p.k.p + (b * p.k.sb + h * p.k.sh + ktok[u] * p.k.sn + DQ * g) * 2;
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) {
          float kv8[8];
          unpack8<E>(ldg16(krow + c * 16), kv8);
#pragma unroll
          for (int j = 0; j < 8; ++j) fk[8 * c + j] -= p.scale * dscol[u] * kv8[j];
        }
      }
      char* d1 = p.dk.p + (b * p.dk.sb + h * p.dk.sh + ktok[u] * p.dk.sn + DQ * g) * 2;
      char* d2 = p.dv.p + (b * p.dv.sb + h * p.dv.sh + ktok[u] * p.dv.sn + DQ * g) * 2;
#pragma unroll
      for (int c = 0; c < DQ / 8; ++c) {
        stg16(d1 + c * 16, pack8<E>(fk + 8 * c));
        stg16(d2 + c * 16, pack8<E>(fv + 8 * c));
      }
    }
  }
}

// ---- one key index per query drawn from softmax(s q.k) (randomized_attention.py:35-37) ----------
// Gumbel-max: argmax_j (logit_ij + G_ij), G_ij = -ln(-ln u_ij) with u from a counter-based hash of
// (seed, b*h, query, key) -- no [N,N] probability matrix, no normalisation, one streaming pass over K.
EA_DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <typename E, int D>
__global__ __launch_bounds__(256) void sm_sample_kernel(const SmP p) {
  constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32;
  __shared__ __attribute__((aligned(16))) char Ks[64 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nqb = (p.N + 63) / 64;
  const int bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const char* qbase = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* kbase = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const int qtok = qb * 64 + wave * 16 + li;
  const bool qvalid = qtok < p.N;
  typename E::x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    u32x4 w = {0u, 0u, 0u, 0u};
    if (qvalid) w = ldg16(qbase + (qtok * p.q.sn + (g * KS + ks) * 8) * 2);
    qf[ks] = as_x8<E>(w);
  }
  const unsigned long long seed = *p.seed;
  const uint32_t base = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)bh)) ^
                        (0x85ebca77u * (uint32_t)qtok);
  float best = -INFINITY;
  int best_j = 0;
  for (int kc = 0; kc < p.N; kc += 64) {
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += 256) {
      const int row = idx / CPR, c = idx - row * CPR;
      const int tok = kc + row;
      u32x4 kw = {0u, 0u, 0u, 0u};
      if (tok < p.N) kw = ldg16(kbase + (tok * p.k.sn + c * 8) * 2);
      sts16(Ks + TileL<D>::off(row, c), kw);
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int row = tt * 16 + li;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc = E::mma(as_x8<E>(lds16(Ks + TileL<D>::off(row, g * KS + ks))), qf[ks], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = kc + tt * 16 + 4 * g + r;
        const uint32_t hsh = mix32(base + 0xc2b2ae3du * (uint32_t)j);
        const float u = ((float)(hsh >> 8) + 0.5f) * (1.f / 16777216.f);
        const float gum = -LN2 * fast_log2(-LN2 * fast_log2(u));      // -ln(-ln u)
        const float val = j < p.N ? acc[r] * p.scale + gum : -INFINITY;
        if (val > best) { best = val; best_j = j; }
      }
    }
  }
  // the four lane groups hold disjoint key subsets of the same query
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    const float ov = __shfl_xor(best, o);
    const int oj = __shfl_xor(best_j, o);
    if (ov > best || (ov == best && oj < best_j)) { best = ov; best_j = oj; }
  }
  if (qvalid && g == 0) p.sample_out[(size_t)bh * p.N + qtok] = best_j;
}
template <typename E, int D>
static int launch_sample(const SmP& p, hipStream_t st) {
  const dim3 grid((unsigned)((long)p.B * p.H * ((p.N + 63) / 64))), block(256);
  hipLaunchKernelGGL((sm_sample_kernel<E, D>), grid, block, 0, st, p);
  return (int)hipGetLastError();
}

template <typename E, int D, bool DR, bool KB>
static int launch_sm_dr(int which, const SmP& p, hipStream_t st) {
  const dim3 grid((unsigned)((long)p.B * p.H * ((p.N + 63) / 64))), block(256);
  if (which == 0) {
    // two (D <= 64: four) query tiles per wave whenever that still leaves every CU a few workgroups
    const long bh = (long)p.B * p.H;
    static const int qt_env = [] { const char* e = getenv("EA_SM_QT"); return e ? atoi(e) : 0; }();   // dev knob
    int qt = 1;
    if (bh * ((p.N + 127) / 128) >= 2 * ea_device_cus()) qt = 2;
    if (D <= 64 && bh * ((p.N + 255) / 256) >= 2 * ea_device_cus()) qt = 4;
    if (qt_env > 0) qt = qt_env;
    if ((D > 64 || DR) && qt > 2) qt = 2;              // (with the keep-mask reads four query tiles do not fit 256 VGPRs)
    const dim3 gq((unsigned)(bh * ((p.N + 64 * qt - 1) / (64 * qt))));
    if (qt == 4) { if constexpr (D <= 64 && !DR) hipLaunchKernelGGL((sm_fwd_kernel<E, D, DR, KB, 4>), gq, block, 0, st, p); }
    else if (qt == 2) hipLaunchKernelGGL((sm_fwd_kernel<E, D, DR, KB, 2>), gq, block, 0, st, p);
    else hipLaunchKernelGGL((sm_fwd_kernel<E, D, DR, KB, 1>), gq, block, 0, st, p);
  } else {
    const long bh = (long)p.B * p.H;
    static const int qt_env = [] { const char* e = getenv("EA_SM_BQT"); return e ? atoi(e) : 0; }();   // dev knob
    int qt = (D <= 64 && bh * ((p.N + 127) / 128) >= 2 * ea_device_cus()) ? 2 : 1;
    if (qt_env > 0) qt = qt_env;
    if (D > 64) qt = 1;
    const dim3 gq((unsigned)(bh * ((p.N + 64 * qt - 1) / (64 * qt))));
    if (qt == 2) { if constexpr (D <= 64) hipLaunchKernelGGL((sm_bwd_dq_kernel<E, D, DR, KB, 2>), gq, block, 0, st, p); }
    else hipLaunchKernelGGL((sm_bwd_dq_kernel<E, D, DR, KB, 1>), gq, block, 0, st, p);
    if (qt == 2) { if constexpr (D <= 64) hipLaunchKernelGGL((sm_bwd_dkv_kernel<E, D, DR, KB, 2>), gq, block, 0, st, p); }
    else hipLaunchKernelGGL((sm_bwd_dkv_kernel<E, D, DR, KB, 1>), gq, block, 0, st, p);
  }
  return (int)hipGetLastError();
}
template <typename E, int D>
static int launch_sm(int which, const SmP& p, hipStream_t st) {
  if (which == 2) return launch_sample<E, D>(p, st);
  if (p.key_norm_bias) return p.keep ? EA_E_UNSUPPORTED : launch_sm_dr<E, D, false, true>(which, p, st);
  return p.keep ? launch_sm_dr<E, D, true, false>(which, p, st) : launch_sm_dr<E, D, false, false>(which, p, st);
}

int softmax_dispatch(int which, const SmP& p, int dtype, int D, hipStream_t st) {
  if (dtype == EA_BF16) {
    if (D == 64) return launch_sm<BF16, 64>(which, p, st);
    if (D == 32) return launch_sm<BF16, 32>(which, p, st);
    if (D == 128) return launch_sm<BF16, 128>(which, p, st);
  } else if (dtype == EA_F16) {
    if (D == 64) return launch_sm<F16, 64>(which, p, st);
    if (D == 32) return launch_sm<F16, 32>(which, p, st);
    if (D == 128) return launch_sm<F16, 128>(which, p, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
