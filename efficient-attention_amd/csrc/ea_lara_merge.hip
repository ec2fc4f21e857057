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
      for (int s0 = 0; s0 < S; s0 += 8) {
        float4 m8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          m8[u] = *reinterpret_cast<const float4*>(p.p_ml + (((size_t)bh * S + min(s0 + u, S - 1)) * C + c) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) { mk = fmaxf(mk, m8[u].x); mt = fmaxf(mt, m8[u].z); }
      }
      float lk = 0.f, lt = 0.f;
      for (int s0 = 0; s0 < S; s0 += 8) {
        float4 m8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          m8[u] = *reinterpret_cast<const float4*>(p.p_ml + (((size_t)bh * S + min(s0 + u, S - 1)) * C + c) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (s0 + u < S) {
            lk += m8[u].y * __expf(m8[u].x - mk);
            if (p.has_t) lt += m8[u].w * __expf(m8[u].z - mt);
          }
        }
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
  // slices in batches of 8 with every load of a batch in flight (a runtime-S loop issued one dependent round trip per
  // slice: 16 us at B*h = 8, S = 64)
  float mk = -INFINITY;
  for (int s0 = 0; s0 < S; s0 += 8) {
    float m8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) m8[u] = p.p_ml[(((size_t)bh * S + min(s0 + u, S - 1)) * C + c) * 4];
#pragma unroll
    for (int u = 0; u < 8; ++u) mk = fmaxf(mk, m8[u]);
  }
  float lk = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < S; s0 += 8) {
    float m8[8], l8[8];
    float4 v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t slot = ((size_t)bh * S + min(s0 + u, S - 1)) * C + c;
      m8[u] = p.p_ml[slot * 4];
      l8[u] = p.p_ml[slot * 4 + 1];
      v8[u] = *reinterpret_cast<const float4*>(p.p_kv + slot * D + j);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < S) {
        const float w = __expf(m8[u] - mk);
        lk += l8[u] * w;
        acc.x += v8[u].x * w; acc.y += v8[u].y * w; acc.z += v8[u].z * w; acc.w += v8[u].w * w;
      }
    }
  }
  const float iv = 1.f / lk;
  *reinterpret_cast<float4*>(p.kv + (size_t)bh * C * D + e) = make_float4(acc.x * iv, acc.y * iv, acc.z * iv, acc.w * iv);
}

// backward: plain sums over the slices + the derived per-landmark quantities
//   p_ml [BH,S,C,4] = (r, dbh, u, -), acc0..3 [BH,S,C,D] = (dkv, sum dZ q, sum t dt q, sum t q)
//   -> r, dbh, u, dkk = dkv.kv [BH,C];  dkv, domq, dqbar = s (M1 - u M2), uq = u qbar [BH,C,D]
// Same grid as the forward merge; D/4 consecutive lanes hold one landmark row.
template <int SL>
__global__ __launch_bounds__(256) void lara_merge_bwd_kernel(const MergeP p) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const int C = p.C, D = p.D, S = p.S;
  if (blockIdx.y == 0) {
    for (int c = tid; c < C; c += 256) {
      float r = 0.f, dbh = 0.f;
      for (int s0 = 0; s0 < S; s0 += 8) {
        float4 m8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          m8[u] = *reinterpret_cast<const float4*>(p.p_ml + (((size_t)bh * S + min(s0 + u, S - 1)) * C + c) * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (s0 + u < S) { r += m8[u].x; dbh += m8[u].y; }
        }
      }
      const size_t o = (size_t)bh * C + c;
      p.r[o] = r;
      if (p.dbh) p.dbh[o] = dbh;
      if (p.dlp) p.dlp[o] = -r;
    }
  }
  // four channels per thread (16-B loads), FOUR slice lanes per column (round 4: with B*h = 8 and S = 64 slices a thread
  // that walked all slices alone made this a 26-70 us launch); lane sl adds slices sl, sl + 4, .. (four in flight), wave 0
  // then adds the four lane sums in order -- for S <= 4 that is the plain slice order.  The D/4 lanes of a landmark row sit
  // in wave 0, so dkk[c] = dkv[c] . kv[c] is a fixed-order shuffle reduction (no LDS atomics).
  constexpr int NC = 256 / SL;                             // columns per block (SL = 1: few slices, no lane split)
  __shared__ float4 red[4][SL][NC];
  __shared__ float redu[SL][NC];
  const int lpr = D >> 2;                                  // lanes per landmark row: 16 (D = 64) or 8
  const int n4 = (C * D) >> 2;
  const int col = tid % NC, sl = tid / NC;
  const int i4 = blockIdx.y * NC + col;
  const bool ok = i4 < n4;
  const int e = (ok ? i4 : 0) * 4;
  const int c = e / D, j = e - c * D;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  float u = 0.f;
  for (int s0 = sl; s0 < S; s0 += 4 * SL) {
    float4 v0[4], v1[4], v2[4], v3[4];
    float uu[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const size_t slot = ((size_t)bh * S + min(s0 + SL * b, S - 1)) * C + c;
      const size_t o4 = slot * D + j;
      v0[b] = *reinterpret_cast<const float4*>(p.acc0 + o4);
      v1[b] = *reinterpret_cast<const float4*>(p.acc1 + o4);
      if (p.has_t) {
        uu[b] = p.p_ml[slot * 4 + 2];
        v2[b] = *reinterpret_cast<const float4*>(p.acc2 + o4);
        v3[b] = *reinterpret_cast<const float4*>(p.acc3 + o4);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (s0 + SL * b < S) {
        a0.x += v0[b].x; a0.y += v0[b].y; a0.z += v0[b].z; a0.w += v0[b].w;
        a1.x += v1[b].x; a1.y += v1[b].y; a1.z += v1[b].z; a1.w += v1[b].w;
        if (p.has_t) {
          u += uu[b];
          a2.x += v2[b].x; a2.y += v2[b].y; a2.z += v2[b].z; a2.w += v2[b].w;
          a3.x += v3[b].x; a3.y += v3[b].y; a3.z += v3[b].z; a3.w += v3[b].w;
        }
      }
    }
  }
  if (SL > 1) {
    red[0][sl][col] = a0; red[1][sl][col] = a1; red[2][sl][col] = a2; red[3][sl][col] = a3;
    redu[sl][col] = u;
    __syncthreads();
    if (sl != 0) return;
  }
#pragma unroll
  for (int l = 1; l < SL; ++l) {
    const float4 b0 = red[0][l][col], b1 = red[1][l][col], b2 = red[2][l][col], b3 = red[3][l][col];
    a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
    a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
    a3.x += b3.x; a3.y += b3.y; a3.z += b3.z; a3.w += b3.w;
    u += redu[l][col];
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
  const dim3 grid_b((unsigned)p.BH, (unsigned)((p.C * p.D / 4 + 63) / 64));       // 64 columns x 4 slice lanes per block
  if (bwd && p.S > 4) hipLaunchKernelGGL(lara_merge_bwd_kernel<4>, grid_b, dim3(256), 0, st, p);
  else if (bwd) hipLaunchKernelGGL(lara_merge_bwd_kernel<1>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(lara_merge_fwd_kernel, grid, dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

}  // namespace ea
