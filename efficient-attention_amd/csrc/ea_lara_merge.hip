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
// grid (BH, ceil(C*D/4 / 256)): a thread owns four channels of one landmark row and re-derives that
// row's scalars from the S partials itself (L1-resident), so small B*h launches still spread over
// the chip (cfg5: B*h = 8, S = 64 used to be eight serial workgroups).
__global__ __launch_bounds__(256) void lara_merge_fwd_kernel(const MergeP p) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, D = p.D, S = p.S;
  const int i4 = blockIdx.y * 256 + tid;
  if (blockIdx.y == 0) {
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
      const size_t o = (size_t)bh * C + c;
      p.lse_k[o] = lsek;
      if (p.has_t) p.lse_t[o] = mt + __logf(lt);
      p.cst[o] = lsek - p.lp[o];
    }
  }
  if (i4 * 4 >= C * D) return;
  const int e = i4 * 4, c = e / D, j = e - c * D;
  float mk = -INFINITY;
  for (int s = 0; s < S; ++s) mk = fmaxf(mk, p.p_ml[(((size_t)bh * S + s) * C + c) * 4]);
  float lk = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const size_t slot = ((size_t)bh * S + s) * C + c;
    const float w = __expf(p.p_ml[slot * 4] - mk);
    lk += p.p_ml[slot * 4 + 1] * w;
    const float4 v = *reinterpret_cast<const float4*>(p.p_kv + slot * D + j);
    acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
  }
  const float iv = 1.f / lk;
  *reinterpret_cast<float4*>(p.kv + (size_t)bh * C * D + e) = make_float4(acc.x * iv, acc.y * iv, acc.z * iv, acc.w * iv);
}

// backward: plain sums over the slices + the derived per-landmark quantities
//   p_ml [BH,S,C,4] = (r, dbh, u, -), acc0..3 [BH,S,C,D] = (dkv, sum dZ q, sum t dt q, sum t q)
//   -> r, dbh, u, dkk = dkv.kv [BH,C];  dkv, domq, dqbar = s (M1 - u M2), uq = u qbar [BH,C,D]
// Same grid as the forward merge; D/4 consecutive lanes hold one landmark row.
__global__ __launch_bounds__(256) void lara_merge_bwd_kernel(const MergeP p) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, D = p.D, S = p.S;
  if (blockIdx.y == 0) {
    for (int c = tid; c < C; c += 256) {
      float r = 0.f, dbh = 0.f;
      for (int s = 0; s < S; ++s) {
        const float* ml = p.p_ml + (((size_t)bh * S + s) * C + c) * 4;
        r += ml[0]; dbh += ml[1];
      }
      const size_t o = (size_t)bh * C + c;
      p.r[o] = r;
      if (p.dbh) p.dbh[o] = dbh;
      if (p.dlp) p.dlp[o] = -r;
    }
  }
  // four channels per thread (16-B loads); the D/4 lanes of a row sit in one wave, so
  // dkk[c] = dkv[c] . kv[c] is a fixed-order shuffle reduction (no LDS atomics)
  const int lpr = D >> 2;                                  // lanes per landmark row: 16 (D = 64) or 8
  const int n4 = (C * D) >> 2;
  const int i4 = blockIdx.y * 256 + tid;
  const bool ok = i4 < n4;
  const int e = (ok ? i4 : 0) * 4;
  const int c = e / D, j = e - c * D;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  float u = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t slot = ((size_t)bh * S + s) * C + c;
    const size_t o4 = slot * D + j;
    const float4 v0 = *reinterpret_cast<const float4*>(p.acc0 + o4);
    const float4 v1 = *reinterpret_cast<const float4*>(p.acc1 + o4);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    if (p.has_t) {
      u += p.p_ml[slot * 4 + 2];
      const float4 v2 = *reinterpret_cast<const float4*>(p.acc2 + o4);
      const float4 v3 = *reinterpret_cast<const float4*>(p.acc3 + o4);
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
  }
  const size_t o = (size_t)bh * C * D + e;
  const float4 kv4 = *reinterpret_cast<const float4*>(p.kv + o);
  float dot = a0.x * kv4.x + a0.y * kv4.y + a0.z * kv4.z + a0.w * kv4.w;
  for (int sh = 1; sh < lpr; sh <<= 1) dot += __shfl_xor(dot, sh);
  if (ok) {
    *reinterpret_cast<float4*>(p.dkv + o) = a0;
    *reinterpret_cast<float4*>(p.domq + o) = a1;
    if (p.has_t) {
      const float sc = p.scale;
      const float4 qb = *reinterpret_cast<const float4*>(p.qbar + o);
      *reinterpret_cast<float4*>(p.dqbar + o) =
          make_float4(sc * (a2.x - u * a3.x), sc * (a2.y - u * a3.y), sc * (a2.z - u * a3.z), sc * (a2.w - u * a3.w));
      *reinterpret_cast<float4*>(p.uq + o) = make_float4(u * qb.x, u * qb.y, u * qb.z, u * qb.w);
    } else if (p.dqbar) {                              // mis-biased: d(mu rows) = s sum dZ q
      *reinterpret_cast<float4*>(p.dqbar + o) = make_float4(p.scale * a1.x, p.scale * a1.y, p.scale * a1.z, p.scale * a1.w);
    }
    if (j == 0) p.dkk[(size_t)bh * C + c] = dot;
  }
}

int lara_merge_dispatch(bool bwd, const MergeP& p, hipStream_t st) {
  if (p.C > 128) return EA_E_UNSUPPORTED;
  const dim3 grid((unsigned)p.BH, (unsigned)((p.C * p.D / 4 + 255) / 256));
  if (bwd) hipLaunchKernelGGL(lara_merge_bwd_kernel, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(lara_merge_fwd_kernel, grid, dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

}  // namespace ea
