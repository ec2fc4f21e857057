// ea_lara_segment.h -- parameter block of the 1-D landmark-proposal kernels (ea_lara_segment.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct SegP {
  char *q, *k;                       // rows to average (pre-LayerNorm when ln), element type
  int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn;
  char *dq, *dk;                     // backward: gradient of the rows (written)
  int64_t dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn;
  const uint8_t* mask;               // [B, N] key padding mask or null
  const float *bias_q, *bias_k;      // [H, D] added to every unmasked row before the LayerNorm, or null
  const float *mbias_q, *mbias_k;    // [D] the row of a masked token before the LayerNorm, or null
  const float *gq, *cq, *gk, *ck;    // LayerNorm weight / bias [D]; gq == null: plain means
  float *qbar, *kbar;                // forward outputs [B*H, L, D]
  const float *d_qbar, *d_kbar;      // backward inputs
  float* part;                       // backward: [B*H*L, 2, 4, D] partial sums (d gamma, d beta, d bias, d mbias)
  int B, H, N, L, segs, nshort;      // segment l: nshort segments of `segs` tokens, then segments of segs + 1
};

int lara_segment_dispatch(bool bwd, const SegP& p, int dtype, int D, hipStream_t st);

// sampling + proposal densities beyond the fused landmark kernels (lara_sample_kernel)
struct SampP {
  const float *qbar, *mu, *noise;                 // [BH,L,D], [BH,L,D], [BH,L or C,D] or null
  float *omega, *qrows, *bhv, *lp;                // forward outputs ([BH,C,D] x2, [BH,C] x2; qrows / bhv may be null)
  const float *d_omega, *d_qrows, *d_bhv, *d_lp;  // backward inputs (d_qrows / d_bhv may be null)
  float *d_qbar, *d_mu;                           // backward outputs [BH,L,D]
  int BH, L, C, D, mis, mode;
  float scale;
};
int lara_sample_dispatch(bool bwd, const SampP& p, hipStream_t st);

}  // namespace ea
