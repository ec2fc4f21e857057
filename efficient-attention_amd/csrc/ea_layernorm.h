// ea_layernorm.h -- parameter block of the row LayerNorm kernels (ea_layernorm.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace ea {

struct LnP {
  const void* x;          // [rows, C] EA_BF16 / EA_F16 / EA_F32
  const float* gamma;     // [C]
  const float* beta;      // [C] (forward)
  float* y;               // [rows, C] fp32 (forward)
  float* stats;           // [rows, 2] (mean, rstd): forward writes (may be null), backward reads
  const float* dy;        // [rows, C] fp32 (backward)
  void* dx;               // [rows, C] in x's type (backward)
  float* part;            // [layernorm_parts(rows), 2, C]: d gamma, d beta partial sums (backward)
  int rows, C, rows_per_block;
  float eps;
};

int layernorm_parts(int rows);
int layernorm_dispatch(bool bwd, const LnP& p, int xtype, hipStream_t st);

}  // namespace ea
