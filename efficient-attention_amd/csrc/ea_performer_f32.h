// ea_performer_f32.h -- parameter block of the exact-fp32 Performer kernels (ea_performer_f32.hip)
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace ea {

struct Pf32T {
  char* p;
  int64_t sb, sh, sn;       // element strides of a [B,H,N,64] view
};

struct Pf32P {
  Pf32T q, k, v, o, dout, dq, dk, dv;
  const uint8_t* mask;                       // [B,N] key padding mask or null
  const float* W;                            // [H, M, 64] random features
  const float *kv, *ksum, *dkv, *dksum;      // [BH, M, 64], [BH, M]
  float* p_max;                              // [BH, S] slice maxima of the key logits
  float *p_kv, *p_ks;                        // slice partials [BH, S, M, 64], [BH, S, M]
  int B, H, N, M, S, tps, dtype;             // dtype: 0 bf16, 1 fp16, 2 fp32; S, tps set by the dispatcher
};

int pf32_slices(int BH, int N);
int pf32_dispatch(int which, const Pf32P& p, hipStream_t st);   // 0 kmax, 1 kv, 2 out, 3 bwd_q, 4 bwd_k

}  // namespace ea
