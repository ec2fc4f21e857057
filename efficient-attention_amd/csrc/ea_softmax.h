// ea_softmax.h -- parameter block of the streaming softmax-attention kernels (ea_softmax.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct SmP {
  struct { char* p; int64_t sb, sh, sn; } q, k, v, o, dout, dq, dk, dv;
  const uint8_t* mask;
  // attention dropout (abstract_attention.py:131): keep mask [B,H,N,keep_ld] u8 over the keys of
  // every query (non-zero = kept), kept probabilities scaled by keep_scale = 1/(1-p); or nullptr
  const uint8_t* keep;
  int keep_ld;
  float keep_scale;
  // key_norm_bias: logits s q.k_j - s |k_j|^2 / 2 (randomized_attention.py:44-50, the second softmax of
  // RA: prm_projection's norm term as a per-key bias, differentiated with respect to k)
  int key_norm_bias;
  // sample: instead of out, draw one key index per query from softmax(s q.k) by Gumbel-max
  // (randomized_attention.py:35-37); counter-based noise from `seed`
  long long* sample_out;           // [BH, N]
  const unsigned long long* seed;  // device scalar
  float* lse;      // [BH, N] natural log
  float* delta;    // [BH, N]
  int B, H, N;
  float scale, scale_log2;
};

int softmax_dispatch(int which, const SmP& p, int dtype, int D, hipStream_t st);

}  // namespace ea
