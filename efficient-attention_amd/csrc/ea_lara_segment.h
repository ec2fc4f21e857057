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

}  // namespace ea
