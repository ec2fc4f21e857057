// ea_lara_merge.h -- parameter block of the LARA partial-merge kernels (ea_lara_merge.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct MergeP {
  int BH, S, C, D, has_t;
  float scale;
  const float *p_ml, *p_kv, *lp;            // forward inputs
  float *kv, *lse_k, *lse_t, *cst;          // forward outputs
  const float *acc0, *acc1, *acc2, *acc3;   // backward inputs (p_ml shared)
  const float* qbar;
  float *r, *dbh, *dlp, *dkk, *dkv, *domq, *dqbar, *uq;   // backward outputs (kv is an input here)
};

int lara_merge_dispatch(bool bwd, const MergeP& p, hipStream_t st);

}  // namespace ea
