// ea_scatter.h -- parameter block of the ScatterBrain feature-half kernels (ea_scatter.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct T4s {
  char* p;
  int64_t sb, sh, sn;
};

struct SbP {
  T4s q, k, v, oloc, out, dout, doloc, dq, dk, dv;
  const uint8_t* mask;          // [B,N] key padding mask or null
  const float* Wf;              // [H, M, 64] random features
  const float *mx, *zall, *sall;   // sequence-wide statistics: [BH,M], [BH,M], [BH,M,64]
  const float* lse_loc;         // [BH,N] log-sum-exp of the window half
  float* r;                     // [BH,N] log-sum-exp of the feature columns (forward output, backward input)
  float* dlse;                  // [BH,N] backward output: d lse_loc
  const float *dsall, *dzall;   // backward (global pass) inputs [BH,M,64], [BH,M]
  float *p_dsall, *p_dzall;     // backward (window pass) partial sums [BH,NP,M,64], [BH,NP,M]
  Geo G;
  int B, H, N, M, w, Wq, nwin, wpb;
  float a, b, lconst;           // d^-1/4, d^-1/2 / 2, log(M) / 2
  long long* prof;              // dev builds (-DEA_PROFILE): phase time stamps
};

int sb_fwd_dispatch(const SbP& p, int dtype, hipStream_t st);
int sb_bwd_dispatch(int which, const SbP& p, int dtype, hipStream_t st);

}  // namespace ea
