// ea_landmark_params.h -- parameter block of the EVA landmark kernels (ea_eva_landmark.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct LmP {
  const char *q, *k, *v;
  int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn;
  char *dq, *dk, *dv;
  int64_t dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn;
  const uint8_t* mask;
  const float *omega, *beta, *dbeta, *dqmean, *dkmean;
  float *qmean, *kmean, *beta_out, *domega;
  Geo G;
  int B, H, L, r, e, J;     // J = slots per chunk (r+2e)^dims
  float scale;
  // round 6: d beta = sum_s dbeta[s * dbeta_stride + ...], s < dbeta_S, formed while it is loaded (the window backward's slice
  // partials, ea_slice_sum's order of additions: that launch folded in); dbeta_S <= 1: dbeta is final
  int dbeta_S;
  long dbeta_stride;
};

int landmark_dispatch(int which, const LmP& p, int dtype, int D, hipStream_t st);

}  // namespace ea
