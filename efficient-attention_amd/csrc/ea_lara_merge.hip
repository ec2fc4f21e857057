// ea_lara_merge.hip -- merge the per-slice partial results of the LARA token-row passes
// (ea_lara_y.hip) into the per-landmark tensors the token-column passes consume.  One workgroup
// per (b,h); everything is [C, D] fp32 with C <= 128 -- tiny, but it replaces ~20 torch kernels
// (amax / exp / mul / sum / log / div ...) per direction.
#include "ea_common.h"
#include "ea_lara_merge.h"

namespace ea {

// forward: log-sum-exp merge of the online-softmax partials
//   p_ml [BH,S,C,4] = (max_k, sum_k, max_t, sum_t), p_kv [BH,S,C,D] un-normalised
//   -> kv [BH,C,D], lse_k, lse_t, cst = lse_k - lp  [BH,C]
__global__ __launch_bounds__(256) void lara_merge_fwd_kernel(const MergeP p) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, D = p.D, S = p.S;
  __shared__ float Mk[128], inv[128];
  for (int c = tid; c < C; c += 256) {
    float mk = -INFINITY, mt = -INFINITY;
    for (int s = 0; s < S; ++s) {
      const float* ml = p.p_ml + (((size_t)bh * S + s) * C + c) * 4;
      mk = fmaxf(mk, ml[0]);
      mt = fmaxf(mt, ml[2]);
    }
    float lk = 0.f, lt = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* ml = p.p_ml + (((size_t)bh * S + s) * C + c) * 4;
      lk += ml[1] * __expf(ml[0] - mk);
      if (p.has_t) lt += ml[3] * __expf(ml[2] - mt);
    }
    const float lsek = mk + __logf(lk);
    Mk[c] = mk;
    inv[c] = 1.f / lk;
    const size_t o = (size_t)bh * C + c;
    p.lse_k[o] = lsek;
    if (p.has_t) p.lse_t[o] = mt + __logf(lt);
    p.cst[o] = lsek - p.lp[o];
  }
  __syncthreads();
  for (int idx = tid; idx < C * D; idx += 256) {
    const int c = idx / D;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t slot = ((size_t)bh * S + s) * C + c;
      acc += p.p_kv[slot * D + (idx - c * D)] * __expf(p.p_ml[slot * 4] - Mk[c]);
    }
    p.kv[(size_t)bh * C * D + idx] = acc * inv[c];
  }
}

// backward: plain sums over the slices + the derived per-landmark quantities
//   p_ml [BH,S,C,4] = (r, dbh, u, -), acc0..3 [BH,S,C,D] = (dkv, sum dZ q, sum t dt q, sum t q)
//   -> r, dbh, u, dkk = dkv.kv [BH,C];  dkv, domq, dqbar = s (M1 - u M2), uq = u qbar [BH,C,D]
__global__ __launch_bounds__(256) void lara_merge_bwd_kernel(const MergeP p) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, D = p.D, S = p.S;
  __shared__ float us[128], dk[128];
  for (int c = tid; c < C; c += 256) {
    float r = 0.f, dbh = 0.f, u = 0.f;
    for (int s = 0; s < S; ++s) {
      const float* ml = p.p_ml + (((size_t)bh * S + s) * C + c) * 4;
      r += ml[0]; dbh += ml[1]; u += ml[2];
    }
    const size_t o = (size_t)bh * C + c;
    p.r[o] = r;
    if (p.dbh) p.dbh[o] = dbh;
    if (p.dlp) p.dlp[o] = -r;
    us[c] = u;
    dk[c] = 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < C * D; idx += 256) {
    const int c = idx / D, j = idx - c * D;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t e = (((size_t)bh * S + s) * C + c) * D + j;
      a0 += p.acc0[e];
      a1 += p.acc1[e];
      if (p.has_t) { a2 += p.acc2[e]; a3 += p.acc3[e]; }
    }
    const size_t o = (size_t)bh * C * D + idx;
    p.dkv[o] = a0;
    p.domq[o] = a1;
    if (p.has_t) {
      p.dqbar[o] = p.scale * (a2 - us[c] * a3);
      p.uq[o] = us[c] * p.qbar[o];
    } else if (p.dqbar) {
      p.dqbar[o] = p.scale * a1;                       // mis-biased: d(mu rows) = s sum dZ q
    }
    atomicAdd(&dk[c], a0 * p.kv[o]);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) p.dkk[(size_t)bh * C + c] = dk[c];
}

int lara_merge_dispatch(bool bwd, const MergeP& p, hipStream_t st) {
  if (p.C > 128) return EA_E_UNSUPPORTED;
  if (bwd) hipLaunchKernelGGL(lara_merge_bwd_kernel, dim3(p.BH), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(lara_merge_fwd_kernel, dim3(p.BH), dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

}  // namespace ea
