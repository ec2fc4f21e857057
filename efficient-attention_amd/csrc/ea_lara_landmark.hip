// ea_lara_landmark.hip -- LARA's landmark pipeline as ONE kernel per direction (lara.py:145-198,
// 214-238): everything between the pooled q/k and the O(N L d) estimator,
//     q_bar = LN(pq Wq^T + bq),  k0 = LN(pk Wk^T + bk)            (q_bar_gen / k_bar_gen, :45-54)
//     k_bar = softmax(s k0 k0^T) k0                               ('-mixed', :157-174)
//     mu = q_bar + k_bar,  omega_c = mu[c mod L] (+/-) eps        (:182-198)
//     M[c,l] = s omega_c.mu_l - s|mu_l|^2/2;  log-proposal / balanced-heuristic weights (:214-238)
// and its backward.  In the reference this is ~45 tiny torch kernels forward and ~90 backward on
// [B,h,L,d] tensors; here one workgroup owns one (b,h) and keeps every matrix (<= 64 x 64 fp32) in
// LDS.  The matrices are far too small for MFMA tiles to matter -- plain fp32 FMA loops, exact
// fp32 like the reference's autocast-exempt LayerNorm/softmax.
// Parameter gradients leave as per-(b,h) partials that the caller sums.
#include "ea_common.h"
#include "ea_lara_lmk.h"

namespace ea {

constexpr int LMK_T = 1024;     // 16 waves: every 16x16 tile of a 64x64 product gets its own wave

// C[m][n] (+)= alpha * sum_k A(m,k) B(k,n), all operands in LDS (fp32); TA/TB read A/B transposed.
// 16x16 output tiles on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 (bit-identical to
// an fmaf chain, same peak rate as the fp32 VALU): one operand float per lane per 4-deep k-step,
// A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[i = 4*(lane>>4)+r][j = lane&15] --
// 2 LDS reads per 1024 MACs instead of 2 per MAC.  A wave per tile, tiles round-robin over waves.
template <bool TA, bool TB, bool ACC>
EA_DEV void mm(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K,
               float alpha, int tid, const float* colbias = nullptr) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int tm = (M + 15) >> 4, tn = (N + 15) >> 4;
  for (int tile = wave; tile < tm * tn; tile += LMK_T / 64) {
    const int m0 = (tile / tn) << 4, n0 = (tile - (tile / tn) * tn) << 4;
    const int am = m0 + li, bn = n0 + li;
    const bool a_ok = am < M, b_ok = bn < N;
    // all LDS operand reads of the tile are issued before the dependent MFMA chain (K <= 64)
    float av[16], bv[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = ks * 4 + g;
      const bool k_ok = k < K;
      av[ks] = (a_ok && k_ok) ? (TA ? A[k * lda + am] : A[am * lda + k]) : 0.f;
      bv[ks] = (b_ok && k_ok) ? (TB ? B[bn * ldb + k] : B[k * ldb + bn]) : 0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks * 4 < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
    if (b_ok) {
      const float cb = colbias ? colbias[bn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (m < M) C[m * ldc + bn] = (ACC ? C[m * ldc + bn] + alpha * acc[r] : alpha * acc[r]) + cb;
      }
    }
  }
}

EA_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int D, bool BWD>
__global__ __launch_bounds__(LMK_T) void lara_lmk_kernel(const LmkP p) {
  constexpr int LD = D + 1;                  // padded row stride (floats)
  constexpr int BUF = 64 * LD;               // one [64][D+1] matrix
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* S0 = sm;              // MU
  float* S1 = S0 + BUF;        // Xq: normalised (pre-affine) q rows, or q_bar itself without MLP
  float* S2 = S1 + BUF;        // Xk: same for k0
  float* S3 = S2 + BUF;        // scratch / dMU
  float* S4 = S3 + BUF;        // scratch / dOM
  float* S5 = S4 + BUF;        // A (mixing softmax) [L][65]
  float* S6 = S5 + BUF;        // k_bar, then M / P [C][65]
  float* S7 = S6 + BUF;        // W, then OM [C][D+1]
  float* vec = S7 + BUF;       // small vectors
  float* rstd_q = vec;         // [64]
  float* rstd_k = vec + 64;
  float* lse_c = vec + 128;    // [64]
  float* lp_c = vec + 192;
  float* bh_c = vec + 256;
  float* colsum = vec + 320;   // [64]
  float* pv = vec + 384;       // affine params: gq, cq, gk, ck, bq, bk (6 x D)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x;
  const int L = p.L, C = p.C;
  const float s = p.scale;
  const size_t oL = (size_t)bh * L * D, oC = (size_t)bh * C * D;

  // Global -> LDS in two steps: `issue` puts up to 4 float4 per thread in flight (a whole [64][D]
  // matrix per workgroup), `commit` writes them to the padded LDS rows.  Everything a phase needs is
  // issued at its start, so the workgroup pays ~one memory round trip per phase, not one per row.
  constexpr int NSLOT = (64 * D / 4 + LMK_T - 1) / LMK_T;
  struct Pre { float4 v[NSLOT]; };
  auto issue = [&](Pre& b, const float* src, int rows) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      b.v[i] = (src && e < rows * D) ? *reinterpret_cast<const float4*>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](float* dst, const Pre& b, int rows) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = (tid + i * LMK_T) * 4;
      if (e < rows * D) {
        float* d = dst + (e / D) * LD + (e % D);
        d[0] = b.v[i].x; d[1] = b.v[i].y; d[2] = b.v[i].z; d[3] = b.v[i].w;
      }
    }
  };
  Pre r_pq, r_pk, r_wq, r_wk, r_noise, r_dom, r_dqr;
  issue(r_pq, p.pq + oL, L);
  issue(r_pk, p.pk + oL, L);
  if (p.has_mlp) { issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
  {
    const float* nsrc = p.noise ? (p.dup == 1 ? p.noise + (size_t)bh * L * D : p.noise + oC) : nullptr;
    issue(r_noise, nsrc, p.dup == 1 ? L : C);
  }
  if (BWD) {
    issue(r_dom, p.d_omega + oC, C);
    issue(r_dqr, p.d_qbar_rows ? p.d_qbar_rows + oC : nullptr, C);
  }
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
  // ---- stage A1: Linear + LayerNorm of the pooled rows ----
  for (int side = 0; side < 2; ++side) {
    float* X = side == 0 ? S1 : S2;
    const float* src = (side == 0 ? p.pq : p.pk) + oL;
    (void)src;
    if (!p.has_mlp) {
      commit(X, side == 0 ? r_pq : r_pk, L);
      __syncthreads();
      continue;
    }
    commit(S3, side == 0 ? r_pq : r_pk, L);
    commit(S7, side == 0 ? r_wq : r_wk, D);                    // W [out][in]
    __syncthreads();
    const float* bias = pv + (side == 0 ? 4 * D : 5 * D);
    mm<false, true, false>(X, LD, S3, LD, S7, LD, L, D, D, 1.f, tid, bias);     // H = P W^T + b
    __syncthreads();
    float* rstd = side == 0 ? rstd_q : rstd_k;
    // a thread per row (row stride D+1 floats: conflict-free across rows); serial over D, no shuffles
    for (int r = tid; r < L; r += LMK_T) {
      float sum = 0.f;
      _Pragma("unroll 16")
      for (int j = 0; j < D; ++j) sum += X[r * LD + j];
      const float mean = sum / D;
      float var = 0.f;
      _Pragma("unroll 16")
      for (int j = 0; j < D; ++j) { const float c0 = X[r * LD + j] - mean; var += c0 * c0; }
      const float rs = rsqrtf(var / D + 1e-5f);
      _Pragma("unroll 16")
      for (int j = 0; j < D; ++j) X[r * LD + j] = (X[r * LD + j] - mean) * rs;
      rstd[r] = rs;
    }
    __syncthreads();
  }
  if (p.eva) {
    // EVA (eva.py:178-190): rf_q_bar = q_bar, rf_k_bar = k0, mu = (rf_q_bar + rf_k_bar) / 2, omega = mu + eps
    if (!BWD) {
      commit(S3, r_noise, L);
      __syncthreads();
      for (int idx = tid; idx < L * D; idx += LMK_T) {
        const int r = idx / D, j = idx % D;
        const float k0v = K0(r, j);
        p.qbar_rows[oC + idx] = k0v;                                        // rf_k_bar
        p.omega[oC + idx] = 0.5f * (QB(r, j) + k0v) + (p.noise ? S3[r * LD + j] : 0.f);
      }
      return;
    }
    commit(S3, r_dom, L);                                                   // d omega
    commit(S0, r_dqr, L);                                                   // d rf_k_bar
    if (p.has_mlp) { issue(r_pq, p.pq + oL, L); issue(r_pk, p.pk + oL, L); issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
    __syncthreads();
    for (int idx = tid; idx < L * D; idx += LMK_T) {
      const int o = (idx / D) * LD + (idx % D);
      S6[o] = 0.5f * S3[o];                                                 // d rf_q_bar
      S4[o] = 0.5f * S3[o] + S0[o];                                         // d rf_k_bar (total)
    }
    __syncthreads();
  } else {
  // ---- stage A2: mixing  A = softmax(s k0 k0^T), k_bar = A k0  (S6 = k_bar) ----
  for (int idx = tid; idx < L * D; idx += LMK_T) S3[(idx / D) * LD + (idx % D)] = K0(idx / D, idx % D);
  __syncthreads();
  if (p.mixed) {
    mm<false, true, false>(S5, LD, S3, LD, S3, LD, L, L, D, s, tid);
    __syncthreads();
    for (int r = tid; r < L; r += LMK_T) {
      float mx = -INFINITY;
      _Pragma("unroll 8")
      for (int j = 0; j < L; ++j) mx = fmaxf(mx, S5[r * LD + j]);
      float den = 0.f;
      _Pragma("unroll 8")
      for (int j = 0; j < L; ++j) { const float e = __expf(S5[r * LD + j] - mx); S5[r * LD + j] = e; den += e; }
      const float inv = 1.f / den;
      _Pragma("unroll 8")
      for (int j = 0; j < L; ++j) S5[r * LD + j] *= inv;
    }
    __syncthreads();
    mm<false, false, false>(S6, LD, S5, LD, S3, LD, L, D, L, 1.f, tid);
  } else {
    for (int idx = tid; idx < L * D; idx += LMK_T) S6[(idx / D) * LD + (idx % D)] = S3[(idx / D) * LD + (idx % D)];
  }
  __syncthreads();
  // ---- stage B: mu, omega, proposal densities ----
  for (int idx = tid; idx < L * D; idx += LMK_T) {
    const int r = idx / D, j = idx % D;
    S0[r * LD + j] = QB(r, j) + S6[r * LD + j];
  }
  __syncthreads();
  const int nrep = C / L;                                       // 1, or 2 with duplicated samples
  commit(S3, r_noise, p.dup == 1 ? L : C);                      // S3 (k0 copy) is free again
  __syncthreads();
  for (int idx = tid; idx < C * D; idx += LMK_T) {
    const int c = idx / D, j = idx % D, l = c % L;
    float eps = 0.f;
    if (p.noise) eps = p.dup == 1 ? (c >= L ? -1.f : 1.f) * S3[l * LD + j] : S3[c * LD + j];
    S7[c * LD + j] = S0[l * LD + j] + eps;
  }
  for (int r = tid; r < L; r += LMK_T) {                          // colsum[l] = |mu_l|^2
    float a2 = 0.f;
    _Pragma("unroll 16")
    for (int j = 0; j < D; ++j) a2 += S0[r * LD + j] * S0[r * LD + j];
    colsum[r] = a2;
  }
  __syncthreads();
  mm<false, true, false>(S6, LD, S7, LD, S0, LD, C, L, D, s, tid);      // s omega_c . mu_l
  __syncthreads();
  for (int c = tid; c < C; c += LMK_T) {
    float mx = -INFINITY;
    _Pragma("unroll 8")
    for (int l = 0; l < L; ++l) {
      const float x = S6[c * LD + l] - 0.5f * s * colsum[l];
      S6[c * LD + l] = x;                                        // M[c][l]
      mx = fmaxf(mx, x);
    }
    float den = 0.f;
    _Pragma("unroll 8")
    for (int l = 0; l < L; ++l) den += __expf(S6[c * LD + l] - mx);
    if (p.mis == 0) den *= (float)nrep;                          // mis-opt: columns repeat nrep times
    const float lse = mx + __logf(den);
    lse_c[c] = lse;
    if (p.mis == 0) {
      const float d0 = S6[c * LD + (c % L)];
      lp_c[c] = d0;
      bh_c[c] = __expf(d0 - lse);
    } else {
      lp_c[c] = lse;
      bh_c[c] = 1.f;
    }
  }
  __syncthreads();

  if (!BWD) {
    for (int idx = tid; idx < C * D; idx += LMK_T) {
      const int c = idx / D, j = idx % D, l = c % L;
      p.omega[oC + idx] = S7[c * LD + j];
      if (p.mis == 0) p.qbar_rows[oC + idx] = QB(l, j);
      else if (p.mis == 1) p.qbar_rows[oC + idx] = S0[l * LD + j];
    }
    for (int c = tid; c < C; c += LMK_T) {
      p.lp[(size_t)bh * C + c] = lp_c[c];
      if (p.mis == 0) p.bhv[(size_t)bh * C + c] = bh_c[c];
    }
    return;
  }

  // =================================== backward ===================================
  // live: S0 MU, S1 Xq, S2 Xk, S5 A, S6 M, S7 OM.   S4 <- dOM (incoming), S3 <- dMU
  commit(S4, r_dom, C);
  for (int idx = tid; idx < L * D; idx += LMK_T) S3[(idx / D) * LD + (idx % D)] = 0.f;
  // re-issue the Linear operands of the parameter-gradient stage now; they land during stage B
  if (p.has_mlp) { issue(r_pq, p.pq + oL, L); issue(r_pk, p.pk + oL, L); issue(r_wq, p.Wq, D); issue(r_wk, p.Wk, D); }
  // dM[c][l] in place of M
  for (int c = tid; c < C; c += LMK_T) {
    const size_t oc = (size_t)bh * C + c;
    const float dlp_in = p.d_lp[oc];
    float dlp, dlse;
    if (p.mis == 0) {
      const float dbh = p.d_bhv ? p.d_bhv[oc] * bh_c[c] : 0.f;
      dlp = dlp_in + dbh;
      dlse = -dbh;
    } else {
      dlp = 0.f;
      dlse = dlp_in;
    }
    const float mult = p.mis == 0 ? (float)nrep : 1.f;
    _Pragma("unroll 8")
    for (int l = 0; l < L; ++l) {
      const float pr = __expf(S6[c * LD + l] - lse_c[c]) * mult;
      S6[c * LD + l] = dlse * pr + ((p.mis == 0 && l == c % L) ? dlp : 0.f);
    }
  }
  __syncthreads();
  // column sums of dM (for the -s|mu_l|^2/2 and the (omega - mu) terms)
  for (int l = tid; l < L; l += LMK_T) {
    float a = 0.f;
    _Pragma("unroll 8")
    for (int c = 0; c < C; ++c) a += S6[c * LD + l];
    colsum[l] = a;
  }
  mm<true, false, true>(S3, LD, S6, LD, S7, LD, L, D, C, s, tid);        // dMU += s dM^T OM
  __syncthreads();
  mm<false, false, true>(S4, LD, S6, LD, S0, LD, C, D, L, s, tid);       // dOM += s dM MU
  for (int idx = tid; idx < L * D; idx += LMK_T) {                        // dMU -= s colsum mu
    const int r = idx / D, j = idx % D;
    S3[r * LD + j] -= s * colsum[r] * S0[r * LD + j];
  }
  __syncthreads();
  // fold the C sample rows onto the L landmarks: omega_c = mu[c mod L] +- eps
  // S6 (dM no longer needed) <- d q_bar extra (mis-opt: from qbar_rows), dMU gets dOM (+ mis-biased rows)
  commit(S0, r_dqr, C);                                          // MU is dead: S0 <- d qbar_rows
  __syncthreads();
  for (int idx = tid; idx < L * D; idx += LMK_T) {
    const int r = idx / D, j = idx % D;
    float dm = S3[r * LD + j], dqx = 0.f;
    for (int k = 0; k < nrep; ++k) {
      const int c = r + k * L;
      dm += S4[c * LD + j];
      if (p.d_qbar_rows) {
        const float g = S0[c * LD + j];
        if (p.mis == 0) dqx += g; else if (p.mis == 1) dm += g;
      }
    }
    S3[r * LD + j] = dm;                 // dMU = d q_bar (common part) = d k_bar
    S6[r * LD + j] = dm + dqx;           // d q_bar
  }
  __syncthreads();
  // ---- stage A backward, k side.  S4 <- d k0 ----
  if (p.mixed) {
    // K0 materialised in S7 (OM no longer needed)
    for (int idx = tid; idx < L * D; idx += LMK_T) S7[(idx / D) * LD + (idx % D)] = K0(idx / D, idx % D);
    __syncthreads();
    mm<true, false, false>(S4, LD, S5, LD, S3, LD, L, D, L, 1.f, tid);   // dK0 = A^T dKb
    mm<false, true, false>(S0, LD, S3, LD, S7, LD, L, L, D, 1.f, tid);   // dA = dKb K0^T  (MU no longer needed)
    __syncthreads();
    for (int r = tid; r < L; r += LMK_T) {                                // dG = A o (dA - rowsum(A o dA)), in S0
      float rs = 0.f;
      _Pragma("unroll 8")
      for (int j = 0; j < L; ++j) rs += S5[r * LD + j] * S0[r * LD + j];
      _Pragma("unroll 8")
      for (int j = 0; j < L; ++j) S0[r * LD + j] = S5[r * LD + j] * (S0[r * LD + j] - rs);
    }
    __syncthreads();
    mm<false, false, true>(S4, LD, S0, LD, S7, LD, L, D, L, s, tid);     // dK0 += s dG K0
    __syncthreads();
    mm<true, false, true>(S4, LD, S0, LD, S7, LD, L, D, L, s, tid);      // dK0 += s dG^T K0
  } else {
    for (int idx = tid; idx < L * D; idx += LMK_T) S4[(idx / D) * LD + (idx % D)] = S3[(idx / D) * LD + (idx % D)];
  }
  __syncthreads();
  }   // !p.eva
  // ---- LayerNorm + Linear backward for both sides: dY in (S4 for k, S6 for q) ----
  for (int side = 0; side < 2; ++side) {
    float* dY = side == 0 ? S6 : S4;
    const float* X = side == 0 ? S1 : S2;
    float* dP = (side == 0 ? p.dpq : p.dpk) + oL;
    if (!p.has_mlp) {
      for (int idx = tid; idx < L * D; idx += LMK_T) dP[idx] = dY[(idx / D) * LD + (idx % D)];
      continue;
    }
    const float* gam = pv + (side == 0 ? 0 : 2 * D);
    const float* rstd = side == 0 ? rstd_q : rstd_k;
    // parameter-gradient partials: d gamma = sum_r dY xhat, d beta = sum_r dY
    float* dvec = p.dvec_part + ((size_t)bh * 2 + side) * 3 * D;
    for (int j = tid; j < D; j += LMK_T) {
      float dg = 0.f, db = 0.f;
      _Pragma("unroll 8")
      for (int r = 0; r < L; ++r) { dg += dY[r * LD + j] * X[r * LD + j]; db += dY[r * LD + j]; }
      dvec[D + j] = dg;
      dvec[2 * D + j] = db;
    }
    __syncthreads();
    // dH = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)),  dxh = dY gamma     (in place in dY)
    for (int r = tid; r < L; r += LMK_T) {
      float s1 = 0.f, s2 = 0.f;
      _Pragma("unroll 16")
      for (int j = 0; j < D; ++j) {
        const float dxh = dY[r * LD + j] * gam[j];
        s1 += dxh; s2 += dxh * X[r * LD + j];
      }
      s1 /= D; s2 /= D;
      const float rs = rstd[r];
      _Pragma("unroll 16")
      for (int j = 0; j < D; ++j) dY[r * LD + j] = rs * (dY[r * LD + j] * gam[j] - s1 - X[r * LD + j] * s2);
    }
    __syncthreads();
    for (int j = tid; j < D; j += LMK_T) {                                  // d bias of the Linear
      float db = 0.f;
      _Pragma("unroll 8")
      for (int r = 0; r < L; ++r) db += dY[r * LD + j];
      dvec[j] = db;
    }
    // dP = dH W ; dW = dH^T P     (S7 <- W, S0 <- P, S5 <- results)
    commit(S7, side == 0 ? r_wq : r_wk, D);
    commit(S0, side == 0 ? r_pq : r_pk, L);
    __syncthreads();
    mm<false, false, false>(S5, LD, dY, LD, S7, LD, L, D, D, 1.f, tid);
    __syncthreads();
    for (int idx = tid; idx < L * D; idx += LMK_T) dP[idx] = S5[(idx / D) * LD + (idx % D)];
    __syncthreads();
    mm<true, false, false>(S5, LD, dY, LD, S0, LD, D, D, L, 1.f, tid);      // dW[out][in]
    __syncthreads();
    float* dW = p.dW_part + ((size_t)bh * 2 + side) * D * D;
    for (int idx = tid; idx < D * D; idx += LMK_T) dW[idx] = S5[(idx / D) * LD + (idx % D)];
    __syncthreads();
  }
}

size_t lara_lmk_lds(int D) { return ((size_t)8 * 64 * (D + 1) + 384 + 6 * D) * sizeof(float); }

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

int lara_lmk_dispatch(bool bwd, const LmkP& p, hipStream_t st) {
  if (p.D == 64) return launch_lmk<64>(bwd, p, st);
  if (p.D == 32) return launch_lmk<32>(bwd, p, st);
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
