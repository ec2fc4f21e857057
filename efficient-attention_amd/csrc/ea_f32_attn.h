// ea_f32_attn.h -- parameter blocks of the fp32-faithful attention cores (ea_f32_attn.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct F32T {                 // [B,H,N,D] fp32 view, element strides
  const float* p;
  int64_t sb, sh, sn;
};

struct GaP {
  F32T q, k, v, ek, ev;       // queries [B,H,Nq,D]; keys / values [B,H,Nk,D]; extra keys / values [B,H,L,D] (L = 0: unused)
  F32T o, dout, dq;           // out (fwd: written; bwd: read), d out, d q
  float *dk, *dv;             // bwd: fp32 [B,H,Nk,D] contiguous, ACCUMULATED into (atomics; the caller zeroes them)
  float *dek, *dev;           // bwd: fp32 [B,H,L,D] contiguous, accumulated into; or null
  const int32_t *idx_q, *idx_k;   // [G,Wq], [G,Wk] token tables, -1 = absent
  const float* bias;          // [*, *, Wq, bias_ld] added to the local logits: batch stride bias_bs and head stride bias_hs
  int64_t bias_hs, bias_bs;   // (0: shared); or null.  A per-TOKEN bias is the one-group case (Wq = Nq): LARA's log alpha
  int bias_ld;
  float* dbias;               // bwd: same layout, accumulated into; or null
  const uint8_t *kmask, *qmask;   // [B,Nk] padded keys, [B,Nq] padded queries (causal EVA), or null
  const uint8_t* keep;        // [B,H,Nq,keep_ld] dropout keep decisions over the Wk + L columns, or null
  int64_t keep_ld;
  float keep_scale;
  float* lse;                 // [B,H,Nq] natural log (fwd: written or null)
  float* stat;                // [B,H,Nq,2] = (row max m, sum l of e^(x - m)): fwd writes, bwd reads.  Kept apart because
                              // lse = m + log l loses l's digits when |m| is large (a fully masked row has m = -5e4: one fp32
                              // ulp of lse is 4e-3 there, i.e. a 0.4 % error in every recomputed probability)
  const float* dlse;          // bwd: gradient of lse or null
  int B, H, Nq, Nk, D, G, Wq, Wk, L;
  int knorm;                  // logits -= s |k_j|^2 / 2 on the local keys (prm_projection, attn_utils.py:324-336)
  int zero_mv;                // masked local keys carry a ZERO value row (EVA's beta: `cv = v * keep`, eva.py:167-176,196)
  int neg_inf;                // padded keys take -inf (softmax baseline, lara.py:205-208) instead of the finite -5e4
  int causal_e;               // >= 0: local key slot j is visible to query slot i iff j <= i + causal_e; -1: no rule
  int chunk, lm_base;         // chunk > 0: extra key c is visible to a query token t iff c < lm_base + t / chunk
  float scale;
};

struct GmP {
  F32T x;                     // [B,H,N,D]
  const int32_t* idx;         // [Cn, J], -1 = absent
  const uint8_t* mask;        // [B,N] or null
  float* mean;                // fwd: [B,H,Cn,D]
  const float* dmean;         // bwd
  float* dx;                  // bwd: [B,H,N,D] contiguous, accumulated into
  int B, H, N, D, Cn, J;
};

int ga_dispatch(bool bwd, const GaP& p, hipStream_t st);
int gm_dispatch(bool bwd, const GmP& p, hipStream_t st);

}  // namespace ea
