// ea_rows_mlp.h -- parameter block of the per-row Linear(+LayerNorm) kernels (ea_rows_mlp.hip).
#pragma once
#include "ea_common.h"

namespace ea {

struct RowsP {
  int R;                               // rows per side
  const float* x[2];                   // [R, D] inputs per side
  const float* W[2];                   // [D, D] (out, in)
  const float* b[2];
  const float* g[2];                   // LayerNorm gain / shift (LN only)
  const float* c[2];
  float* y[2];                         // fwd: [R, D] outputs
  float* zhat;                         // [sides, R, D] normalised pre-affine rows (LN; fwd writes, bwd reads)
  float* rstd;                         // [sides, R]
  const float* dy[2];                  // bwd: gradient of y
  float* dx[2];                        // bwd: gradient of x
  float* feed;                         // bwd: [R, planes, sides, D] = (dz [, dy o zhat, dy]) for the column sums
  float* dW_part;                      // bwd: [blocks, sides, D, D]
};

}  // namespace ea
