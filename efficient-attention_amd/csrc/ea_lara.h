// ea_lara.h -- parameter block and shared device code of the LARA kernels.
//
// LARA (lara.py:177-251) couples every token n with C <= 128 landmark samples c through
// [C x N] score matrices.  Two register layouts cover all of its passes:
//   X ("token-column", ea_lara_x.hip): MFMA tiles D[c = 4g+r][n = li].  Reductions and
//      contractions over c are in-lane / 4-lane, so this layout produces per-token results:
//      the forward combine out_n = sum_c W[c,n] kv_c, dq, dk, dv.
//   Y ("token-row", ea_lara_y.hip): MFMA tiles D[n = 4g+r][c = li].  Reductions and
//      contractions over n are in-lane / 4-lane and accumulate across the whole sequence, so
//      this layout produces per-landmark results: kv_stats / LSE_k / LSE_t in the forward and
//      d(kv_stats), d(omega), d(q_bar) and the scalar sums in the backward.
// Natural-log quantities are carried in the log2 domain inside the kernels (suffix 2).
#pragma once
#include "ea_common.h"

namespace ea {

struct T4l {
  char* p;
  int64_t sb, sh, sn;
};

enum { MIS_OPT = 0, MIS_BIASED = 1, MIS_BH = 2 };
enum { LX_FWD = 0, LX_BWDQ = 1, LX_BWDK = 2, LX_QCORR = 3, LX_POUT = 4, LX_PBWDQ = 5, LX_PBWDK = 6, LX_FWDM = 7 };
enum { LY_FWD = 0, LY_BWDQ = 1, LY_BWDK = 2, LY_PMAX = 3, LY_PKV = 4, LY_PBWDQ = 5 };
// The Performer baseline (kernelized_attention.py:20-56,116-121) runs on the same two skeletons
// with the m random features W_j in the role of the landmark rows (modes LX_P*, LY_P*):
//   phi(x)[j] = m^-1/2 exp(d^-1/4 W_j.x - d^-1/2 |x|^2/2 - stab) + 1e-4

struct LaraP {
  T4l q, k, v, o, dout, dq, dk, dv;
  const uint8_t* mask;          // [B,N] key padding mask or null
  // landmark-side matrices, fp32 [BH, C, D]
  const float *omega, *qbar, *kv, *dkv, *uq;
  // per-landmark scalars, fp32 [BH, C] (natural log units)
  const float *lse_k, *lse_t, *bhv, *cst, *dkk, *rsum;
  // per-token scalars, fp32 [BH, N]
  float *lseZ, *tmean, *rowdot, *sda;
  // Y-pass partial outputs: [BH, nsplit, C, *]
  float *p_ml;                  // FWD: [.., C, 4] = (m_k, l_k, m_t, l_t) ; BWDQ: (r, dbh, u, 0)
  float *p_acc0, *p_acc1, *p_acc2, *p_acc3;   // [.., C, D]: FWD kv ; BWDQ dkv, domega, M1, M2 ; BWDK domega
  int B, H, N, D, C, NCT;       // NCT = 16-landmark tiles (C padded to NCT*16)
  int mis;                      // MIS_*
  int nsplit;                   // Y: sequence splits per (b,h); X: blocks per (b,h)
  int tok_per_block;            // X: tokens handled by one block (multiple of 64)
  int tok_begin[17];            // Y: token range of slice i is [tok_begin[i], tok_begin[i+1])
  float kappa, scale, scale_log2;
  // Performer: omega = W indexed per head, stab [BH] key stabiliser (natural units)
  int w_per_head;
  const float* stab;
  // ScatterBrain's feature statistics (scatterbrain_attention.py:99-131) on the Performer modes: the
  // stabiliser is per FEATURE (stab [BH, C]) and the key max includes the -|k|^2 term and the padding mask
  int stab_per_feature;
  float norm_coef2;             // log2-domain coefficient of |x|^2 in the logits
  float knorm_coef;             // coefficient of x * (sum of logit grads) in dk / dq
  float ratio, feps;
  // finish pass (ea_lara_f.hip): gradients of the pooled q / k rows, fp32 [BH, pool_L, D], spread back
  // over the pool_r x pool_r token blocks of a pool_gw-wide grid (pool_r = 0: no pooling term)
  const float *dpq, *dpk;
  int pool_r, pool_gw, pool_L;
  float pool_inv;
  // merge-on-load (round 5): a consumer pass takes the S <= 4 per-slice partials of the producing token-row pass and merges
  // them in its prologue -- the arithmetic of ea_lara_merge.hip, operation for operation -- instead of waiting for a merge
  // launch of its own (7-16 us each at cfg3 for a few KB per (b,h)).  Block 0 of a (b,h) writes the merged tensors the
  // later passes (and the backward) read.
  //   LX_FWDM (forward combine): m_ml [BH,S,C,4] = (max_k, sum_k, max_t, sum_t), m_acc0 [BH,S,C,D] un-normalised kv, m_lp [BH,C]
  //       -> kv, lse_k, lse_t, cst (m_kv, m_lsek, m_lset, m_cst)
  //   key side of the fused backward (lara_fk_kernel, FOLD): m_ml = (r, dbh, u, -), m_acc0..3 = (dkv, sum dZ q, sum t dt q,
  //       sum t q) partials of lara_fq_kernel -> dkv rows / dkk / r for itself, and m_dbh, m_dlp, m_domq, m_dqbar, m_uq
  const float *m_ml, *m_acc0, *m_acc1, *m_acc2, *m_acc3, *m_lp;
  int m_S;
  float *m_kv, *m_lsek, *m_lset, *m_cst;
  float *m_dbh, *m_dlp, *m_domq, *m_dqbar, *m_uq;
  long long* prof;              // dev builds (-DEA_PROFILE): phase time stamps
};

// ---- the elementwise core of the estimator (lara.py:221-243), one (c, n) entry -------------
// inputs in the log2 domain: A2 = s*log2e*omega_c.q_n, T2 = s*log2e*qbar_c.q_n
struct LaraElem {
  float t, alpha, la2;
};
EA_DEV LaraElem lara_alpha(int mis, float T2, float lse_t2, float bh, float kappa, float tmean) {
  LaraElem e;
  e.t = 0.f; e.alpha = 1.f; e.la2 = 0.f;
  if (mis == MIS_OPT) {
    e.t = fast_exp2(T2 - lse_t2);
    e.alpha = bh + kappa * (e.t - tmean);
    e.la2 = fast_log2(fmaxf(e.alpha, 1e-8f));
  } else if (mis == MIS_BIASED) {
    e.la2 = T2;
  }
  return e;
}

}  // namespace ea
