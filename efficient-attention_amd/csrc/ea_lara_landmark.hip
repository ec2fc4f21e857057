// ea_lara_landmark.hip -- LARA's landmark pipeline as ONE kernel per direction (lara.py:145-198,
// 214-238): everything between the pooled q/k and the O(N L d) estimator,
//     q_bar = LN(pq Wq^T + bq),  k0 = LN(pk Wk^T + bk)            (q_bar_gen / k_bar_gen, :45-54)
//     k_bar = softmax(s k0 k0^T) k0                               ('-mixed', :157-174)
//     mu = q_bar + k_bar,  omega_c = mu[c mod L] (+/-) eps        (:182-198)
//     M[c,l] = s omega_c.mu_l - s|mu_l|^2/2;  log-proposal / balanced-heuristic weights (:214-238)
// and its backward (with `eva`: the mu = (q_bar + k0)/2 pipeline of eva.py:178-190).  In the
// reference this is ~45 tiny torch kernels forward and ~90 backward on [B,h,L,d] tensors; here one
// 16-wave workgroup owns one (b,h) and keeps every matrix (<= 64 x 64 fp32) in LDS:
//   * matrix products: 16x16 tiles on the exact-fp32 v_mfma_f32_16x16x4_f32, a wave per tile;
//   * row-wise steps (LayerNorm, softmax, their backward): four lanes per row, 16 columns per
//     lane in registers, reduced with the permlane swaps of ea_common.h;
//   * column sums (parameter gradients): four columns per wave, 16 row-lanes per column.
// All of it exact fp32 like the reference's autocast-exempt LayerNorm/softmax.  The step is
// latency-bound (a few MB of traffic in total), so the design goal is few, short barrier phases:
// 9 forward, 16 backward.  Parameter gradients leave as per-(b,h) partials that the caller sums.
#include <stdlib.h>
#include "ea_common.h"
#include "ea_lara_lmk.h"

namespace ea {

#define STAMP(i) EA_STAMP(p, i)

constexpr int LMK_T = 1024;     // 16 waves: every 16x16 tile of a 64x64 product gets its own wave
constexpr int LMK_W = LMK_T / 64;
constexpr int LD = 65;          // row stride (floats) of every LDS matrix
// Matrix products: fp32 matrices in LDS, operands rounded to fp16 (11 significant bits, 8x finer than
// the bf16 operands autocast gives the reference's own Linear / einsum here) when a tile is fetched,
// v_mfma_f32_16x16x32_f16 with fp32 accumulation -- 1/16 of the matrix-pipe time of the exact-fp32
// v_mfma_f32_16x16x4_f32 (which made this kernel matrix-pipe bound: 2048 cycles per 64^3 product
// and SIMD).  Gradient-side operands (dM, d k_bar, dG, dH) are pre-scaled by a per-matrix power of
// two taken from their block-wide maximum, so loss scaling cannot push them out of fp16 range.
// LMK_HALF = false keeps the exact-fp32 products (debugging).
constexpr bool LMK_HALF = true;
constexpr int BUF = 64 * LD;

// C[m][n] (+)= alpha * sum_k A(m,k) B(k,n) (+ colbias[n]); A, B in LDS (fp32), C in LDS or global;
// TA/TB read A/B transposed.  Operand layout of v_mfma_f32_16x16x4_f32: one float per lane per
// k-step, A[i = lane&15][k' = lane>>4], B[k' = lane>>4][j = lane&15], D[i = 4*(lane>>4)+r][j = lane&15].
// The k order inside a dot product is free, so lane group g = lane>>4 takes the contiguous k range
// [g*steps, (g+1)*steps): with the odd row stride 65 the 64 lanes of every operand read then hit
// (nearly) distinct banks in all four transpose combinations.  Tiles go round-robin over the waves.
struct MMJob {
  float* C; int ldc;
  const float* A; int lda;
  const float* B; int ldb;
  int M, N, K;
  float alpha;
  const float* colbias;
  float sa, sb;                 // power-of-two pre-scales of the A / B operand (fp16 range), undone in alpha
};
struct MMOp { float av[16], bv[16]; };

EA_DEV int mm_tiles(const MMJob& j) { return ((j.M + 15) >> 4) * ((j.N + 15) >> 4); }

// operand fetch of one 16x16 tile: every LDS read is issued here, before any MFMA
template <bool TA, bool TB>
EA_DEV void mm_load(MMOp& o, const MMJob& j, int tile, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int tn = (j.N + 15) >> 4;
  const int m0 = (tile / tn) << 4, n0 = (tile - (tile / tn) * tn) << 4;
  const int am = m0 + li, bn = n0 + li;
  const int steps = (j.K + 3) >> 2, kb = steps * g;
  // Unconditional loads (am, bn, k < 64 always lie inside the 64 x 65 buffers) and a select on the
  // k range only: rows am >= M / columns bn >= N may hold anything, they only reach outputs that
  // mm_store drops.  (Per-load predication costs an exec-mask branch per element.)
  const int nk = min(steps, j.K - kb);                  // valid k-steps of this lane group
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int k = kb + ks;
    const float a = TA ? j.A[k * j.lda + am] : j.A[am * j.lda + k];
    const float b = TB ? j.B[bn * j.ldb + k] : j.B[k * j.ldb + bn];
    o.av[ks] = ks < nk ? a : 0.f;
    o.bv[ks] = ks < nk ? b : 0.f;
  }
}
EA_DEV f32x4 mm_chain(const MMOp& o, const MMJob& j, f32x4 acc) {
  const int steps = (j.K + 3) >> 2;
  if (LMK_HALF) {
    // k-slot (s, j') of the 32-deep MFMA step s <-> this lane's ks = 8 s + j' (same map for A and B)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 xa0, xa1, xb0, xb1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xa0[i] = (_Float16)(o.av[i] * j.sa); xa1[i] = (_Float16)(o.av[8 + i] * j.sa);
      xb0[i] = (_Float16)(o.bv[i] * j.sb); xb1[i] = (_Float16)(o.bv[8 + i] * j.sb);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa0, xb0, acc, 0, 0, 0);
    if (steps > 8) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa1, xb1, acc, 0, 0, 0);
    return acc;
  }
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 16; ks += 2) {             // two accumulation chains (even / odd k-steps)
    if (ks < steps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[ks] * j.sa, o.bv[ks] * j.sb, acc, 0, 0, 0);
    if (ks + 1 < steps) b = __builtin_amdgcn_mfma_f32_16x16x4f32(o.av[ks + 1] * j.sa, o.bv[ks + 1] * j.sb, b, 0, 0, 0);
  }
  return acc + b;
}
template <bool ACC>
EA_DEV void mm_store(const MMJob& j, int tile, int lane, f32x4 acc) {
  const int g = lane >> 4, li = lane & 15;
  const int tn = (j.N + 15) >> 4;
  const int m0 = (tile / tn) << 4, n0 = (tile - (tile / tn) * tn) << 4;
  const int bn = n0 + li;
  if (bn < j.N) {
    const float cb = j.colbias ? j.colbias[bn] : 0.f;
    const float al = j.alpha / (j.sa * j.sb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * g + r;
      if (m < j.M) j.C[m * j.ldc + bn] = (ACC ? j.C[m * j.ldc + bn] + al * acc[r] : al * acc[r]) + cb;
    }
  }
}

// C (+)= alpha * A B (+ colbias); sa / sb: power-of-two pre-scales of the operands (gradient matrices).
// A tile per wave, tiles round-robin over the waves; all LDS reads of a tile precede its MFMAs.
template <bool TA, bool TB, bool ACC>
EA_DEV void mm(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K,
               float alpha, int tid, const float* colbias = nullptr, float sa = 1.f, float sb = 1.f) {
  const MMJob j = {C, ldc, A, lda, B, ldb, M, N, K, alpha, colbias, sa, sb};
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < mm_tiles(j); t += LMK_W) {
    MMOp o;
    mm_load<TA, TB>(o, j, t, lane);
    mm_store<ACC>(j, t, lane, mm_chain(o, j, f32x4{0.f, 0.f, 0.f, 0.f}));
  }
}

// C += alpha * (A1 B1 + A2 B2): two products with the same shape into one accumulator
template <bool TA1, bool TB1, bool TA2, bool TB2>
EA_DEV void mm_sum2(const MMJob& j1, const MMJob& j2, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int tile = wave; tile < mm_tiles(j1); tile += LMK_W) {
    MMOp o1, o2;
    mm_load<TA1, TB1>(o1, j1, tile, lane);
    mm_load<TA2, TB2>(o2, j2, tile, lane);
    f32x4 acc = mm_chain(o1, j1, f32x4{0.f, 0.f, 0.f, 0.f});
    acc = mm_chain(o2, j2, acc);                         // (same operand scales in both)
    mm_store<true>(j1, tile, lane, acc);
  }
}

EA_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// power-of-two 1/scale for a matrix whose per-wave maxima of |x| sit in gm[0..n): m * scale in [0.5, 1)
EA_DEV float pow2_inv_scale(const float* gm, int n) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) m = fmaxf(m, gm[i]);
  if (!(m > 0.f) || m > 3e38f) return 1.f;
  int e;
  (void)frexpf(m, &e);
  return ldexpf(1.f, -e);
}

// sum over the 16 lanes that share lane>>4 (xor shuffles below 16 stay inside the group)
EA_DEV float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int D, bool BWD>
__global__ __launch_bounds__(LMK_T) void lara_lmk_kernel(const LmkP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* S0 = sm;              // MU                       | bwd: dA / dG, then Pq
  float* S1 = S0 + BUF;        // Xq: normalised (pre-affine) q rows, or q_bar itself without MLP
  float* S2 = S1 + BUF;        // Xk: same for k0
  float* S3 = S2 + BUF;        // Pq, then k0             | bwd: dMU, then Pk
  float* S4 = S3 + BUF;        // Pk, then noise          | bwd: dOM, then d k0
  float* S5 = S4 + BUF;        // Wk, then A (mixing softmax) [L][L]   | bwd tail: Wk
  float* S6 = S5 + BUF;        // k_bar, then M           | bwd: dM, then d q_bar
  float* S7 = S6 + BUF;        // Wq, then OM [C][D]      | bwd: k0, then Wq
  float* S8 = S7 + BUF;        // bwd: d qbar_rows (incoming)
  float* vec = S8 + BUF;       // small vectors
  float* rstd_q = vec;         // [64]
  float* rstd_k = vec + 64;
  float* musq = vec + 128;     // |mu_l|^2
  float* dmcol = vec + 192;    // column sums of dM
  float* pv = vec + 256;       // affine params: gq, cq, gk, ck, bq, bk (6 x D)
  float* gmx = pv + 6 * D;     // [4][16] per-wave max |x| of the gradient matrices dM, d k_bar, dG, dH

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = blockIdx.x;
  const int L = p.L, C = p.C;
  const float s = p.scale;
  const size_t oL = (size_t)bh * L * D, oC = (size_t)bh * C * D;
  const int nrep = C / L;                                       // 1, or 2 with duplicated samples
  // row phases: four lanes (g = 0..3) per row, lane (li, g) owns columns 16g .. 16g+15 of row
  // rrow; waves 0-3 cover 64 rows of one matrix, waves 4-7 those of a second one.
  const int rrow = ((wave & 3) << 4) + li, cbase = g << 4;
  // column phases: wave w owns columns 4w .. 4w+3 (+ 4 LMK_W, ...), the 16 lanes li stride over the rows
  const int ccol0 = (wave << 2) + g;

  // Global -> LDS in two steps: `issue` puts a whole [64][D] matrix per workgroup in flight (one
  // float4 per thread), `commit` writes it to the padded LDS rows.  Everything a phase needs is
  // issued long before, so the workgroup pays one exposed memory round trip (the first).
  constexpr int NSLOT = (64 * 64 / 4 + LMK_T - 1) / LMK_T;      // sized for 64 columns (the mixing matrix)
  struct Pre { float4 v[NSLOT]; };
  auto issue_n = [&](Pre& b, const float* src, int rows, int cols) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      b.v[i] = (src && e < rows * cols) ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit_n = [&](float* dst, const Pre& a, int rows, int cols) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      if (e < rows * cols) {
        float* d = dst + (e / cols) * LD + (e % cols);
        d[0] = a.v[i].x; d[1] = a.v[i].y; d[2] = a.v[i].z; d[3] = a.v[i].w;
      }
    }
  };
  auto issue = [&](Pre& b, const float* src, int rows) { issue_n(b, src, rows, D); };
  // dst = sa * a + sb * b
  auto commit2 = [&](float* dst, const Pre& a, float sa, const Pre& b, float sb, int rows) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      if (e < rows * D) {
        float* d = dst + (e / D) * LD + (e % D);
        d[0] = sa * a.v[i].x + sb * b.v[i].x; d[1] = sa * a.v[i].y + sb * b.v[i].y;
        d[2] = sa * a.v[i].z + sb * b.v[i].z; d[3] = sa * a.v[i].w + sb * b.v[i].w;
      }
    }
  };
  auto commit = [&](float* dst, const Pre& a, int rows) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      if (e < rows * D) {
        float* d = dst + (e / D) * LD + (e % D);
        d[0] = a.v[i].x; d[1] = a.v[i].y; d[2] = a.v[i].z; d[3] = a.v[i].w;
      }
    }
  };
  // forward intermediates kept for the backward (LmkP::saved): Xq, Xk, MU [L][D], A [L][64], 1/std
  float* sv = p.saved ? p.saved + (size_t)bh * lara_lmk_saved_per_bh(L, D) : nullptr;
  float* sv_xq = sv, *sv_xk = sv + L * D, *sv_mu = sv + 2 * L * D, *sv_a = sv + 3 * L * D;
  float* sv_rstd = sv + 3 * L * D + L * 64;
  const bool reload = BWD && sv != nullptr;             // backward that skips the forward recomputation
  Pre r_pq, r_pk, r_wq, r_wk, r_noise, r_dom, r_dqr;
  if (reload) {
    issue(r_pq, p.has_mlp ? sv_xq : p.pq + oL, L);      // normalised rows (or q_bar / k_bar themselves)
    issue(r_pk, p.has_mlp ? sv_xk : p.pk + oL, L);
    if (!p.eva) {
      issue(r_wq, sv_mu, L);
      issue_n(r_wk, p.mixed ? sv_a : nullptr, L, 64);
    }
    if (tid < 128) rstd_q[tid] = sv_rstd[tid];          // rstd_q, rstd_k are adjacent
  } else {
    issue(r_pq, p.pq + oL, L);
    issue(r_pk, p.pk + oL, L);
    if (p.has_mlp) { issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
  }
  {
    const float* nsrc = p.noise ? (p.dup == 1 ? p.noise + (size_t)bh * L * D : p.noise + oC) : nullptr;
    issue(r_noise, nsrc, p.dup == 1 ? L : C);
  }
  float pre_dlp = 0.f, pre_dbh = 0.f;
  if (BWD) {
    issue(r_dom, p.d_omega + oC, C);
    issue(r_dqr, p.d_qbar_rows ? p.d_qbar_rows + oC : nullptr, C);
    if (!p.eva && wave < 4 && rrow < C) {
      pre_dlp = p.d_lp[(size_t)bh * C + rrow];
      pre_dbh = p.d_bhv ? p.d_bhv[(size_t)bh * C + rrow] : 0.f;
    }
  }
  if (BWD && tid < 64) gmx[tid] = 0.f;
  if (p.has_mlp) {
    for (int i = tid; i < D; i += LMK_T) {
      pv[i] = p.gq[i]; pv[D + i] = p.cq[i]; pv[2 * D + i] = p.gk[i]; pv[3 * D + i] = p.ck[i];
      pv[4 * D + i] = p.bq[i]; pv[5 * D + i] = p.bk[i];
    }
  }
  // q_bar / k0 accessors: affine applied on the fly to the normalised rows
  auto QB = [&](int r, int j) { return p.has_mlp ? pv[j] * S1[r * LD + j] + pv[D + j] : S1[r * LD + j]; };
  auto K0 = [&](int r, int j) { return p.has_mlp ? pv[2 * D + j] * S2[r * LD + j] + pv[3 * D + j] : S2[r * LD + j]; };

  // =========================== forward (recomputed in backward) ===========================
  STAMP(0);
  // ---- F1/F2/F3: Linear + LayerNorm of the pooled rows, both sides at once ----
  if (reload) {
    commit(S1, r_pq, L); commit(S2, r_pk, L);
    if (!p.eva) {
      commit(S0, r_wq, L);                                       // MU
      if (p.mixed) commit_n(S5, r_wk, L, 64);                    // A
      commit(S4, r_noise, p.dup == 1 ? L : C);
    }
  } else if (p.has_mlp) {
    commit(S3, r_pq, L); commit(S7, r_wq, D);                    // W [out][in]
    commit(S4, r_pk, L); commit(S5, r_wk, D);
  } else {
    commit(S1, r_pq, L); commit(S2, r_pk, L);
  }
  if (BWD && !p.eva) commit(S8, r_dqr, C);
  __syncthreads();
  STAMP(1);
  if (p.has_mlp && !reload) {
    mm<false, true, false>(S1, LD, S3, LD, S7, LD, L, D, D, 1.f, tid, pv + 4 * D);          // H = P W^T + b
    mm<false, true, false>(S2, LD, S4, LD, S5, LD, L, D, D, 1.f, tid, pv + 5 * D);
    __syncthreads();
    STAMP(2);
    if (wave < 8) {
      float* X = wave < 4 ? S1 : S2;
      float* rstd = wave < 4 ? rstd_q : rstd_k;
      const bool r_ok = rrow < L;
      float x[16], sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = (r_ok && cbase + i < D) ? X[rrow * LD + cbase + i] : 0.f;
        sum += x[i];
      }
      const float mean = quad_sum(sum) / D;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float c0 = (cbase + i < D) ? x[i] - mean : 0.f;
        var += c0 * c0;
      }
      const float rs = rsqrtf(quad_sum(var) / D + 1e-5f);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (r_ok && cbase + i < D) X[rrow * LD + cbase + i] = (x[i] - mean) * rs;
      if (g == 0 && r_ok) rstd[rrow] = rs;
    }
    __syncthreads();
    STAMP(3);
  }
  if (p.eva) {
    // EVA (eva.py:178-190): rf_q_bar = q_bar, rf_k_bar = k0, mu = (rf_q_bar + rf_k_bar) / 2, omega = mu + eps
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        const int e = (tid + i * LMK_T) * 4;
        if (e < L * D) {
          const int r = e / D, j = e % D;
          const float nz[4] = {r_noise.v[i].x, r_noise.v[i].y, r_noise.v[i].z, r_noise.v[i].w};
          float4 kk, om;
          float* kp = &kk.x; float* op = &om.x;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float k0v = K0(r, j + t);
            kp[t] = k0v;                                                     // rf_k_bar
            op[t] = 0.5f * (QB(r, j + t) + k0v) + nz[t];
          }
          *reinterpret_cast<float4*>(p.qbar_rows + oC + e) = kk;
          *reinterpret_cast<float4*>(p.omega + oC + e) = om;
        }
      }
      if (sv) {
        for (int idx = tid; idx < L * D; idx += LMK_T) {
          const int o = (idx / D) * LD + (idx % D);
          sv_xq[idx] = S1[o]; sv_xk[idx] = S2[o];
        }
        if (tid < 128) sv_rstd[tid] = rstd_q[tid];
      }
      return;
    }
    commit2(S6, r_dom, 0.5f, r_dom, 0.f, L);                                // d rf_q_bar = d omega / 2
    commit2(S4, r_dom, 0.5f, r_dqr, 1.f, L);                                // d rf_k_bar (total)
    if (p.has_mlp) { issue(r_pq, p.pq + oL, L); issue(r_pk, p.pk + oL, L); issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
    __syncthreads();
  } else {
  // ---- F4..F7: mixing  A = softmax(s k0 k0^T), k_bar = A k0  (S6 = k_bar) ----
  if (!reload) {
  for (int idx = tid; idx < L * D; idx += LMK_T) {
    const int o = (idx / D) * LD + (idx % D);
    const float k0v = K0(idx / D, idx % D);
    S3[o] = k0v;
    if (!p.mixed) S6[o] = k0v;
  }
  commit(S4, r_noise, p.dup == 1 ? L : C);                       // S4 (pk staging) is free again
  __syncthreads();
  STAMP(4);
  if (p.mixed) {
    mm<false, true, false>(S5, LD, S3, LD, S3, LD, L, L, D, s, tid);
    __syncthreads();
    STAMP(5);
    if (wave < 4) {
      const bool r_ok = rrow < L;
      float x[16], mx = -1e30f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = (r_ok && cbase + i < L) ? S5[rrow * LD + cbase + i] : -INFINITY;
        mx = fmaxf(mx, x[i]);
      }
      mx = quad_max(mx);
      float den = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { x[i] = __expf(x[i] - mx); den += x[i]; }
      const float inv = 1.f / quad_sum(den);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (r_ok && cbase + i < L) S5[rrow * LD + cbase + i] = x[i] * inv;
    }
    __syncthreads();
    STAMP(6);
    mm<false, false, false>(S6, LD, S5, LD, S3, LD, L, D, L, 1.f, tid);
    __syncthreads();
    STAMP(7);
  }
  }   // !reload
  // ---- F8: mu (unless reloaded) and the sample rows omega ----
  for (int idx = tid; idx < L * D; idx += LMK_T) {
    const int r = idx / D, j = idx % D;
    const float mu = reload ? S0[r * LD + j] : QB(r, j) + S6[r * LD + j];
    S0[r * LD + j] = mu;
    for (int k = 0; k < nrep; ++k) {
      const int c = r + k * L;
      float eps = 0.f;
      if (p.noise) eps = p.dup == 1 ? (k ? -1.f : 1.f) * S4[r * LD + j] : S4[c * LD + j];
      S7[c * LD + j] = mu + eps;
    }
  }
  __syncthreads();
  STAMP(8);
  // ---- F9: M = s omega mu^T (all waves), |mu_l|^2 (waves 4-7), forward outputs ----
  mm<false, true, false>(S6, LD, S7, LD, S0, LD, C, L, D, s, tid);      // s omega_c . mu_l
  if (wave >= 4 && wave < 8) {
    float a2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float m = (rrow < L && cbase + i < D) ? S0[rrow * LD + cbase + i] : 0.f;
      a2 += m * m;
    }
    a2 = quad_sum(a2);
    if (g == 0 && rrow < L) musq[rrow] = a2;
  }
  if (!BWD) {
    for (int idx = tid; idx < C * D; idx += LMK_T) {
      const int c = idx / D, j = idx % D, l = c % L;
      p.omega[oC + idx] = S7[c * LD + j];
      if (p.mis == 0) p.qbar_rows[oC + idx] = QB(l, j);
      else if (p.mis == 1) p.qbar_rows[oC + idx] = S0[l * LD + j];
    }
    if (sv) {                                                       // keep the intermediates for the backward
      for (int idx = tid; idx < L * D; idx += LMK_T) {
        const int o = (idx / D) * LD + (idx % D);
        if (p.has_mlp) { sv_xq[idx] = S1[o]; sv_xk[idx] = S2[o]; }
        sv_mu[idx] = S0[o];
      }
      if (p.mixed)
        for (int idx = tid; idx < L * 64; idx += LMK_T) sv_a[idx] = S5[(idx >> 6) * LD + (idx & 63)];
      if (tid < 128) sv_rstd[tid] = rstd_q[tid];
    }
  }
  __syncthreads();
  STAMP(9);
  // ---- F10: proposal densities per sample row (and, in backward, dM in place of M) ----
  if (wave < 4) {
    const int c = rrow;
    const bool r_ok = c < C;
    const int cl = r_ok ? c % L : -1;
    float x[16], mx = -1e30f, d0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int l = cbase + i;
      x[i] = (r_ok && l < L) ? S6[c * LD + l] - 0.5f * s * musq[l] : -INFINITY;       // M[c][l]
      mx = fmaxf(mx, x[i]);
      if (l == cl) d0 = x[i];
    }
    mx = quad_max(mx);
    d0 = quad_sum(d0);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) den += __expf(x[i] - mx);
    den = quad_sum(den);
    if (p.mis == 0) den *= (float)nrep;                          // mis-opt: columns repeat nrep times
    const float lse = mx + __logf(den);
    const float lpv = p.mis == 0 ? d0 : lse;
    const float bhv = p.mis == 0 ? __expf(d0 - lse) : 1.f;
    if (!BWD) {
      if (g == 0 && r_ok) {
        p.lp[(size_t)bh * C + c] = lpv;
        if (p.mis == 0) p.bhv[(size_t)bh * C + c] = bhv;
      }
    } else {
      float dlp, dlse;
      if (p.mis == 0) {
        const float dbh = pre_dbh * bhv;
        dlp = pre_dlp + dbh;
        dlse = -dbh;
      } else {
        dlp = 0.f;
        dlse = pre_dlp;
      }
      const float mult = p.mis == 0 ? (float)nrep : 1.f;
      float gm = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = cbase + i;
        if (r_ok && l < L) {
          const float dm = dlse * __expf(x[i] - lse) * mult + ((p.mis == 0 && l == cl) ? dlp : 0.f);
          S6[c * LD + l] = dm;
          gm = fmaxf(gm, fabsf(dm));
        }
      }
      gm = wave_max(gm);
      if (lane == 0) gmx[wave] = gm;
    }
  }
  if (!BWD) return;

  // =================================== backward ===================================
  // live: S0 MU, S1 Xq, S2 Xk, S5 A, S6 dM, S7 OM, S8 d qbar_rows.   S4 <- dOM (incoming; with
  // mis-biased the cotangent of the mu rows joins it: both fold onto mu[c mod L])
  commit2(S4, r_dom, 1.f, r_dqr, (p.mis == 1 && p.d_qbar_rows) ? 1.f : 0.f, C);
  __syncthreads();
  STAMP(10);
  // ---- B2: column sums of dM; dMU = s dM^T OM; dOM += s dM MU ----
  for (int ccol = ccol0; ccol < 64; ccol += 4 * LMK_W) {
    float a = 0.f;
    if (ccol < L)
      for (int r = li; r < C; r += 16) a += S6[r * LD + ccol];
    a = group16_sum(a);
    if (li == 0 && ccol < L) dmcol[ccol] = a;
  }
  {
    const float sdm = pow2_inv_scale(gmx, 4);
    mm<true, false, false>(S3, LD, S6, LD, S7, LD, L, D, C, s, tid, nullptr, sdm);       // dMU = s dM^T OM
    mm<false, false, true>(S4, LD, S6, LD, S0, LD, C, D, L, s, tid, nullptr, sdm);       // dOM += s dM MU
  }
  __syncthreads();
  STAMP(11);
  // ---- B3: fold the C sample rows onto the L landmarks (omega_c = mu[c mod L] +- eps) ----
  //   S3 <- d mu = d k_bar (common part), S6 <- d q_bar, S7 <- k0 (mixed) / S4 <- d k0 (not mixed)
  float gkb = 0.f;
  for (int idx = tid; idx < L * D; idx += LMK_T) {
    const int r = idx / D, j = idx % D, o = r * LD + j;
    float dm = S3[o] - s * dmcol[r] * S0[o], dqx = 0.f;
    for (int k = 0; k < nrep; ++k) {
      const int c = r + k * L;
      dm += S4[c * LD + j];
      if (p.mis == 0 && p.d_qbar_rows) dqx += S8[c * LD + j];
    }
    S3[o] = dm;
    gkb = fmaxf(gkb, fabsf(dm));
    S6[o] = dm + dqx;
    if (p.mixed) S7[o] = K0(r, j); else S4[o] = dm;
  }
  gkb = wave_max(gkb);
  if (lane == 0) gmx[16 + wave] = gkb;
  // re-issue the Linear operands of the parameter-gradient stage; they land during the mixing backward
  if (p.has_mlp) { issue(r_pq, p.pq + oL, L); issue(r_pk, p.pk + oL, L); issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
  __syncthreads();
  STAMP(12);
  // ---- B4..B6: mixing backward.  S4 <- d k0 ----
  if (p.mixed) {
    const float skb = pow2_inv_scale(gmx + 16, 16);
    mm<true, false, false>(S4, LD, S5, LD, S3, LD, L, D, L, 1.f, tid, nullptr, 1.f, skb);   // dK0 = A^T dKb
    mm<false, true, false>(S0, LD, S3, LD, S7, LD, L, L, D, 1.f, tid, nullptr, skb, 1.f);   // dA = dKb K0^T  (MU is dead)
    __syncthreads();
    STAMP(13);
    if (wave < 4) {                                                             // dG = A o (dA - rowsum(A o dA)), in S0
      const bool r_ok = rrow < L;
      float a[16], d[16], rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = r_ok && cbase + i < L;
        a[i] = ok ? S5[rrow * LD + cbase + i] : 0.f;
        d[i] = ok ? S0[rrow * LD + cbase + i] : 0.f;
        rs += a[i] * d[i];
      }
      rs = quad_sum(rs);
      float gm = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (r_ok && cbase + i < L) {
          const float dg = a[i] * (d[i] - rs);
          S0[rrow * LD + cbase + i] = dg;
          gm = fmaxf(gm, fabsf(dg));
        }
      gm = wave_max(gm);
      if (lane == 0) gmx[32 + wave] = gm;
    }
    __syncthreads();
    STAMP(14);
    const float sdg = pow2_inv_scale(gmx + 32, 4);
    mm_sum2<false, false, true, false>(                                         // dK0 += s (dG + dG^T) K0
        MMJob{S4, LD, S0, LD, S7, LD, L, D, L, s, nullptr, sdg, 1.f},
        MMJob{S4, LD, S0, LD, S7, LD, L, D, L, s, nullptr, sdg, 1.f}, tid);
    __syncthreads();
    STAMP(15);
  }
  }   // !p.eva
  // ---- tail: LayerNorm + Linear backward, both sides at once.  dY: S6 (q side), S4 (k side) ----
  if (!p.has_mlp) {
    for (int idx = tid; idx < L * D; idx += LMK_T) {
      const int o = (idx / D) * LD + (idx % D);
      p.dpq[oL + idx] = S6[o];
      p.dpk[oL + idx] = S4[o];
    }
    return;
  }
  // T1: d gamma = sum_r dY xhat, d beta = sum_r dY (column phases); stage W, P of both sides
  for (int side = 0; side < 2; ++side) {
    const float* dY = side == 0 ? S6 : S4;
    const float* X = side == 0 ? S1 : S2;
    for (int ccol = ccol0; ccol < 64; ccol += 4 * LMK_W) {
      float dg = 0.f, db = 0.f;
      if (ccol < D)
        for (int r = li; r < L; r += 16) {
          const float y = dY[r * LD + ccol];
          dg += y * X[r * LD + ccol];
          db += y;
        }
      dg = group16_sum(dg);
      db = group16_sum(db);
      if (li == 0 && ccol < D) {
        float* dvec = p.dvec_part + ((size_t)bh * 2 + side) * 3 * D;
        dvec[D + ccol] = dg;
        dvec[2 * D + ccol] = db;
      }
    }
  }
  commit(S7, r_wq, D); commit(S0, r_pq, L);
  commit(S5, r_wk, D); commit(S3, r_pk, L);
  __syncthreads();
  STAMP(16);
  // T2: dH = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)),  dxh = dY gamma     (in place in dY)
  if (wave < 8) {
    float* dY = wave < 4 ? S6 : S4;
    const float* X = wave < 4 ? S1 : S2;
    const float* gam = pv + (wave < 4 ? 0 : 2 * D);
    const bool r_ok = rrow < L;
    float dxh[16], xh[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = r_ok && cbase + i < D;
      xh[i] = ok ? X[rrow * LD + cbase + i] : 0.f;
      dxh[i] = ok ? dY[rrow * LD + cbase + i] * gam[cbase + i] : 0.f;
      s1 += dxh[i];
      s2 += dxh[i] * xh[i];
    }
    s1 = quad_sum(s1) / D;
    s2 = quad_sum(s2) / D;
    const float rs = r_ok ? (wave < 4 ? rstd_q : rstd_k)[rrow] : 0.f;
    float gm = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (r_ok && cbase + i < D) {
        const float dh = rs * (dxh[i] - s1 - xh[i] * s2);
        dY[rrow * LD + cbase + i] = dh;
        gm = fmaxf(gm, fabsf(dh));
      }
    gm = wave_max(gm);
    if (lane == 0) gmx[48 + wave] = gm;
  }
  __syncthreads();
  STAMP(17);
  // T3: d bias of the Linear (column sums of dH); dP = dH W and dW = dH^T P straight to global
  for (int side = 0; side < 2; ++side) {
    const float* dY = side == 0 ? S6 : S4;
    for (int ccol = ccol0; ccol < 64; ccol += 4 * LMK_W) {
      float db = 0.f;
      if (ccol < D)
        for (int r = li; r < L; r += 16) db += dY[r * LD + ccol];
      db = group16_sum(db);
      if (li == 0 && ccol < D) p.dvec_part[((size_t)bh * 2 + side) * 3 * D + ccol] = db;
    }
  }
  {
    float* dWq = p.dW_part + ((size_t)bh * 2 + 0) * D * D;
    float* dWk = p.dW_part + ((size_t)bh * 2 + 1) * D * D;
    const float shq = pow2_inv_scale(gmx + 48, 4), shk = pow2_inv_scale(gmx + 52, 4);
    mm<false, false, false>(p.dpq + oL, D, S6, LD, S7, LD, L, D, D, 1.f, tid, nullptr, shq);
    mm<false, false, false>(p.dpk + oL, D, S4, LD, S5, LD, L, D, D, 1.f, tid, nullptr, shk);
    mm<true, false, false>(dWq, D, S6, LD, S0, LD, D, D, L, 1.f, tid, nullptr, shq);        // dW[out][in]
    mm<true, false, false>(dWk, D, S4, LD, S3, LD, D, D, L, 1.f, tid, nullptr, shk);
  }
  STAMP(18);
}

size_t lara_lmk_lds(int D) { return ((size_t)9 * BUF + 256 + 6 * D + 64) * sizeof(float); }

template <int D>
static int launch_lmk(bool bwd, const LmkP& p, hipStream_t st) {
  const size_t lds = lara_lmk_lds(D);
  const void* fn = bwd ? reinterpret_cast<const void*>(&lara_lmk_kernel<D, true>)
                       : reinterpret_cast<const void*>(&lara_lmk_kernel<D, false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  if (bwd) hipLaunchKernelGGL((lara_lmk_kernel<D, true>), dim3(p.BH), dim3(LMK_T), lds, st, p);
  else hipLaunchKernelGGL((lara_lmk_kernel<D, false>), dim3(p.BH), dim3(LMK_T), lds, st, p);
  return (int)hipGetLastError();
}

bool lmk2_supported(bool bwd, const LmkP& p);
int lmk2_dispatch(bool bwd, const LmkP& p, hipStream_t st);

int lara_lmk_dispatch(bool bwd, const LmkP& p0, hipStream_t st) {
  LmkP p = p0;
  p.prof = nullptr;
  // second-generation kernels (ea_lmk2.hip: fp16 operand tiles, strips in registers, 2 workgroups per CU);
  // EA_LMK_V1=1 keeps the round-1 kernels below (fp32 matrices in LDS) for A/B comparison
  static const bool v1 = getenv("EA_LMK_V1") && getenv("EA_LMK_V1")[0] == '1';
  if (!v1 && lmk2_supported(bwd, p)) return lmk2_dispatch(bwd, p, st);
  if (p.colbias || p.d_colbias) return EA_E_UNSUPPORTED;      // '-vmixed' column bias: second-generation kernels only
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "lara_lmk", bwd ? 1 : 0);
#endif
  if (p.D == 64) return launch_lmk<64>(bwd, p, st);
  if (p.D == 32) return launch_lmk<32>(bwd, p, st);
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
