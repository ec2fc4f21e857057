// ea_lara_lmk.h -- parameter block of the fused LARA landmark kernels (ea_lara_landmark.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct LmkP {
  const float *pq, *pk;                                   // [BH, L, D] pooled q / k (or q_bar / k_bar)
  const float *Wq, *bq, *gq, *cq, *Wk, *bk, *gk, *ck;     // Linear + LayerNorm parameters (has_mlp)
  const float* noise;                                     // standard normal, or null (eval)
  float *omega, *qbar_rows, *bhv, *lp;                    // forward outputs
  const float *d_omega, *d_qbar_rows, *d_bhv, *d_lp;      // backward inputs
  // round 5: d omega = dom_scale (d_omega + sum_s dom_parts[bh][s]) formed while the strip is loaded (ea_slice_sum's launch
  // folded in): dom_parts [BH, dom_S <= 4, C, D] slice partials of the key-side pass, or null (d_omega is final)
  const float* dom_parts;
  int dom_S;
  float dom_scale;
  // round 6 (EVA): d rf_k_bar = sum_s d_qbar_rows[s * dqr_stride + ...], s < dqr_S <= 4 (the window backward's slice partials,
  // ea_slice_sum's order); dqr_S <= 1: d_qbar_rows is final
  int dqr_S;
  long dqr_stride;
  float *dpq, *dpk, *dW_part, *dvec_part;                 // backward outputs
  int BH, L, C, D;
  int has_mlp, mixed, mis, dup;
  int eva;                                                // EVA's mu pipeline (eva.py:178-190) instead of LARA's
  float scale;
  float* saved;                                           // forward intermediates (fwd writes, bwd reads), or null
  const float* colbias;                                   // [BH, L] added to every row of the mixing logits ('-vmixed'), or null
  float* d_colbias;                                       // backward: its gradient, or null
  long long* prof;                                        // dev builds (-DEA_PROFILE): phase time stamps
};

int lara_lmk_dispatch(bool bwd, const LmkP& p, hipStream_t st);
// per-(b,h) floats of LmkP::saved: Xq, Xk, MU [L][D], A [L][64], rstd_q, rstd_k [64]
__host__ __device__ inline size_t lara_lmk_saved_per_bh(int L, int D) { return (size_t)3 * L * D + (size_t)L * 64 + 128; }

}  // namespace ea
