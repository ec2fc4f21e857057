// ea_f32_mm.h -- 16 x 16 output tiles of an exact-fp32 matrix product out of LDS images (v_mfma_f32_16x16x4_f32)
#pragma once
#include "ea_common.h"

namespace ea {

// acc += A(m0.., k) B(k, n0..) over K (multiple of 16); TA / TB_: operand stored transposed (ea_rows_mlp.hip)
// KC > 0: K is that compile-time constant -- the loop unrolls completely and every operand load of a tile is in flight
// before the first MFMA (with a run-time K each of the K / 16 trips waited for its own eight loads: a workgroup here is
// one wave per SIMD, nothing else hides that latency)
template <bool TA, bool TB_, int KC = 0>
EA_DEV void tile_mm(f32x4& acc, const float* A, int lda, const float* B, int ldb, int m0, int n0, int K, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int steps = (KC > 0 ? KC : K) >> 2, kb = g * steps;
  const int am = m0 + li, bn = n0 + li;
  // four independent accumulation chains: a workgroup of these kernels is one wave per SIMD (its LDS images fill most of
  // the CU), so back-to-back dependent MFMAs would leave the matrix pipe idle three quarters of the time
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1, acc3 = acc1;
#pragma unroll
  for (int k0 = 0; k0 < steps; k0 += 4) {
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb + k0 + i;
      a[i] = TA ? A[k * lda + am] : A[am * lda + k];
      b[i] = TB_ ? B[bn * ldb + k] : B[k * ldb + bn];
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc3, 0, 0, 0);
  }
  acc += (acc1 + acc2) + acc3;
}

}  // namespace ea
