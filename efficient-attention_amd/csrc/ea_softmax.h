// ea_softmax.h -- parameter block of the streaming softmax-attention kernels (ea_softmax.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct SmP {
  struct { char* p; int64_t sb, sh, sn; } q, k, v, o, dout, dq, dk, dv;
  const uint8_t* mask;
  float* lse;      // [BH, N] natural log
  float* delta;    // [BH, N]
  int B, H, N;
  float scale, scale_log2;
};

int softmax_dispatch(int which, const SmP& p, int dtype, int D, hipStream_t st);

}  // namespace ea
