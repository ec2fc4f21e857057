// ea_rows_mlp.hip -- the per-chunk "mu" networks of EVA / causal EVA as one HIP pass each way:
//     y_s = [LayerNorm_s](x_s W_s^T + b_s),   s = query side, key side
// (eva.py:78-98,178-183: adaptive_mu_q / adaptive_mu_k = Linear(d,d) [+ LayerNorm(d)] applied to
// the chunk means; causal_eva.py:376-392,706-707).  R = B*h*L rows of d <= 128 channels: a handful of
// MB, so what matters is doing it in two launches instead of ~25 framework kernels.
//
// Exact fp32: the products run on v_mfma_f32_16x16x4_f32 (lane (i = lane&15, k' = lane>>4) holds
// A[i][k'], B[k'][j = lane&15]; D[4*(lane>>4)+r][lane&15]); lane group g takes the contiguous k range
// [g*K/4, (g+1)*K/4), which with the odd LDS row stride D+1 keeps the 64 lanes of every operand read
// on distinct banks in both the plain and the transposed access.
//
// forward  : WG = 4 waves; W staged once; per tile of RB rows: X -> LDS, Z = X W^T + b (a 16x16 tile
//            per wave at a time), row statistics with 256/RB lanes per row, y (and for the backward
//            zhat = (z - mean) rstd and rstd) written with 16-byte stores.
// backward : dz from dy (LayerNorm backward per row), dx = dz W, and dW = dz^T X accumulated in
//            registers over all row tiles of the workgroup (one [D,D] partial per workgroup); the
//            column sums that give db, d gamma, d beta are left to ea_colsum_f32 over a "feed" buffer
//            [R, 3, sides, D] = (dz, dy o zhat, dy) this kernel writes on the way.
#include "ea_common.h"
#include "ea_rows_mlp.h"

namespace ea {

template <int D> struct RowsCfg {
  static constexpr int RB = D == 128 ? 32 : 64;      // rows per tile (three row tiles + W within 160 KB)
  static constexpr int LD = D + 1;
  // threads per workgroup: the LDS image (W + three row tiles, 115 KB at D = 128) allows ONE workgroup per CU, so at D = 128
  // it is a 16-wave one -- a row tile's 16 dx tiles and 64 dW tiles spread over 16 waves instead of 4 (round 5: the LM
  // recipe's 9216 rows x 2 sides: 46 -> us backward)
  static constexpr int NT = D == 128 ? 1024 : 256;
  static constexpr int NW = NT / 64;
  static constexpr int TPR = NT / RB;                // lanes per row in the row phases
};

// acc += A(m0.., k) B(k, n0..) over K (multiple of 16); TA / TB: operand stored transposed
template <bool TA, bool TB>
EA_DEV void tile_mm(f32x4& acc, const float* A, int lda, const float* B, int ldb, int m0, int n0, int K, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int steps = K >> 2, kb = g * steps;
  const int am = m0 + li, bn = n0 + li;
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < steps; k0 += 4) {
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb + k0 + i;
      a[i] = TA ? A[k * lda + am] : A[am * lda + k];
      b[i] = TB ? B[bn * ldb + k] : B[k * ldb + bn];
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc2, 0, 0, 0);
  }
  acc += acc2;
}

// W [D, D] -> LDS rows of stride D + 1; batches of eight 16-B loads in flight per thread (a load per
// loop trip would be a global round trip per trip)
template <int D> EA_DEV void stage_weight(float* Ws, const float* W, int tid) {
  constexpr int NT = RowsCfg<D>::NT;
  constexpr int LD = D + 1, N4 = D * D / 4, NB = N4 / NT < 8 ? N4 / NT : 8;
  for (int base = 0; base < N4; base += NT * NB) {
    float4 v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = reinterpret_cast<const float4*>(W)[base + i * NT + tid];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = base + i * NT + tid;
      float* d = Ws + (idx * 4 / D) * LD + (idx * 4) % D;
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
  }
}

template <int TPR> EA_DEV float row_sum(float v) {
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

template <int D, bool LN>
__global__ __launch_bounds__(RowsCfg<D>::NT) void rows_mlp_fwd_kernel(const RowsP p) {
  using C = RowsCfg<D>;
  constexpr int RB = C::RB, LD = C::LD, TPR = C::TPR, CPT = D / TPR;   // columns per lane in a row phase
  constexpr int NT = C::NT, NW = C::NW;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ws = sm;
  float* Xs = Ws + D * LD;
  float* Zs = Xs + RB * LD;
  float* mean_s = Zs + RB * LD;
  float* rstd_s = mean_s + RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int s = blockIdx.y;
  const float* x = p.x[s];
  const float* W = p.W[s];
  const float* bias = p.b[s];
  stage_weight<D>(Ws, W, tid);
  const int ntile = (p.R + RB - 1) / RB;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int r0 = tile * RB;
    __syncthreads();
    for (int idx = tid; idx < RB * D / 4; idx += NT) {
      const int row = idx * 4 / D, c = (idx * 4) % D;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < p.R) v = *reinterpret_cast<const float4*>(x + (size_t)(r0 + row) * D + c);
      float* d = Xs + row * LD + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    for (int t = wave; t < (RB / 16) * (D / 16); t += NW) {
      const int m0 = (t / (D / 16)) * 16, n0 = (t % (D / 16)) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, true>(acc, Xs, LD, Ws, LD, m0, n0, D, lane);
      const float bb = bias[n0 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) Zs[(m0 + 4 * g + r) * LD + n0 + li] = acc[r] + bb;
    }
    __syncthreads();
    if (LN) {
      const int row = tid / TPR, part = tid % TPR;
      float sum = 0.f;
      for (int j = 0; j < CPT; ++j) sum += Zs[row * LD + part + TPR * j];
      const float mean = row_sum<TPR>(sum) * (1.f / D);
      float sq = 0.f;
      for (int j = 0; j < CPT; ++j) { const float dlt = Zs[row * LD + part + TPR * j] - mean; sq += dlt * dlt; }
      const float rstd = rsqrtf(row_sum<TPR>(sq) * (1.f / D) + 1e-5f);
      if (part == 0) {
        mean_s[row] = mean; rstd_s[row] = rstd;
        if (r0 + row < p.R && p.rstd) p.rstd[(size_t)s * p.R + r0 + row] = rstd;
      }
      __syncthreads();
    }
    for (int idx = tid; idx < RB * D / 4; idx += NT) {
      const int row = idx * 4 / D, c = (idx * 4) % D;
      if (r0 + row >= p.R) continue;
      const float* z = Zs + row * LD + c;
      float4 y = make_float4(z[0], z[1], z[2], z[3]);
      if (LN) {
        const float m = mean_s[row], rs = rstd_s[row];
        const float4 zh = make_float4((y.x - m) * rs, (y.y - m) * rs, (y.z - m) * rs, (y.w - m) * rs);
        const float4 gm = *reinterpret_cast<const float4*>(p.g[s] + c);
        const float4 bt = *reinterpret_cast<const float4*>(p.c[s] + c);
        if (p.zhat) *reinterpret_cast<float4*>(p.zhat + ((size_t)s * p.R + r0 + row) * D + c) = zh;
        y = make_float4(zh.x * gm.x + bt.x, zh.y * gm.y + bt.y, zh.z * gm.z + bt.z, zh.w * gm.w + bt.w);
      }
      *reinterpret_cast<float4*>(p.y[s] + (size_t)(r0 + row) * D + c) = y;
    }
  }
}

template <int D, bool LN>
__global__ __launch_bounds__(RowsCfg<D>::NT) void rows_mlp_bwd_kernel(const RowsP p) {
  using C = RowsCfg<D>;
  constexpr int RB = C::RB, LD = C::LD, TPR = C::TPR, CPT = D / TPR;
  constexpr int NT = C::NT, NW = C::NW;
  constexpr int WT = (D / 16) * (D / 16) / NW;        // dW tiles per wave
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ws = sm;
  float* Xs = Ws + D * LD;
  float* As = Xs + RB * LD;                           // g o dy, then dz
  float* Hs = As + RB * LD;                           // zhat
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int s = blockIdx.y, S = gridDim.y;
  const float* x = p.x[s];
  const float* W = p.W[s];
  const float* dy = p.dy[s];
  stage_weight<D>(Ws, W, tid);
  f32x4 dW[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i) dW[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int planes = LN ? 3 : 1;
  const int ntile = (p.R + RB - 1) / RB;
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int r0 = tile * RB;
    __syncthreads();
    // ---- stage X, a = gamma o dy (LN) or dy, zhat; the feed planes that need no row statistics ----
    for (int idx = tid; idx < RB * D / 4; idx += NT) {
      const int row = idx * 4 / D, c = (idx * 4) % D;
      const bool ok = r0 + row < p.R;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv, hv = xv;
      if (ok) {
        xv = *reinterpret_cast<const float4*>(x + (size_t)(r0 + row) * D + c);
        dv = *reinterpret_cast<const float4*>(dy + (size_t)(r0 + row) * D + c);
        if (LN) hv = *reinterpret_cast<const float4*>(p.zhat + ((size_t)s * p.R + r0 + row) * D + c);
      }
      float* d = Xs + row * LD + c;
      d[0] = xv.x; d[1] = xv.y; d[2] = xv.z; d[3] = xv.w;
      float4 av = dv;
      if (LN) {
        const float4 gm = *reinterpret_cast<const float4*>(p.g[s] + c);
        av = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        float* h = Hs + row * LD + c;
        h[0] = hv.x; h[1] = hv.y; h[2] = hv.z; h[3] = hv.w;
        if (ok) {
          float* f = p.feed + (((size_t)(r0 + row) * 3 + 1) * S + s) * D + c;
          *reinterpret_cast<float4*>(f) = make_float4(dv.x * hv.x, dv.y * hv.y, dv.z * hv.z, dv.w * hv.w);
          *reinterpret_cast<float4*>(f + (size_t)S * D) = dv;
        }
      }
      float* a = As + row * LD + c;
      a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
    }
    __syncthreads();
    if (LN) {
      // dz = rstd (a - mean(a) - zhat mean(a o zhat))
      const int row = tid / TPR, part = tid % TPR;
      float s1 = 0.f, s2 = 0.f;
      for (int j = 0; j < CPT; ++j) {
        const float a = As[row * LD + part + TPR * j];
        s1 += a;
        s2 += a * Hs[row * LD + part + TPR * j];
      }
      const float m1 = row_sum<TPR>(s1) * (1.f / D), m2 = row_sum<TPR>(s2) * (1.f / D);
      const float rs = r0 + row < p.R ? p.rstd[(size_t)s * p.R + r0 + row] : 0.f;
      for (int j = 0; j < CPT; ++j) {
        const int o = row * LD + part + TPR * j;
        As[o] = rs * (As[o] - m1 - Hs[o] * m2);
      }
      __syncthreads();
    }
    // ---- feed plane 0: dz (its column sum is the Linear's bias gradient) ----
    for (int idx = tid; idx < RB * D / 4; idx += NT) {
      const int row = idx * 4 / D, c = (idx * 4) % D;
      if (r0 + row >= p.R) continue;
      const float* a = As + row * LD + c;
      *reinterpret_cast<float4*>(p.feed + (((size_t)(r0 + row) * planes) * S + s) * D + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
    // ---- dx = dz W ----
    for (int t = wave; t < (RB / 16) * (D / 16); t += NW) {
      const int m0 = (t / (D / 16)) * 16, n0 = (t % (D / 16)) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, false>(acc, As, LD, Ws, LD, m0, n0, D, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + m0 + 4 * g + r;
        if (row < p.R) p.dx[s][(size_t)row * D + n0 + li] = acc[r];
      }
    }
    // ---- dW += dz^T X  (rows beyond R hold zeros) ----
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      const int t = wave + NW * i;
      const int m0 = (t / (D / 16)) * 16, n0 = (t % (D / 16)) * 16;
      tile_mm<true, false>(dW[i], As, LD, Xs, LD, m0, n0, RB, lane);
    }
  }
  float* dst = p.dW_part + ((size_t)blockIdx.x * S + s) * D * D;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int t = wave + NW * i;
    const int m0 = (t / (D / 16)) * 16, n0 = (t % (D / 16)) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)(m0 + 4 * g + r) * D + n0 + li] = dW[i][r];
  }
}

template <int D> static size_t rows_lds(bool bwd) {
  using C = RowsCfg<D>;
  return ((size_t)D * C::LD + (size_t)C::RB * C::LD * (bwd ? 3 : 2) + 2 * C::RB) * sizeof(float);
}

int rows_mlp_blocks(int R, int D) {
  const int rb = D == 128 ? 32 : 64;
  const int ntile = (R + rb - 1) / rb;
  return ntile < 128 ? ntile : 128;                  // per side; a workgroup loops over its row tiles
}

template <int D, bool LN>
static int launch_rows(const RowsP& p, int sides, bool bwd, hipStream_t st) {
  const size_t lds = rows_lds<D>(bwd);
  const void* fn = bwd ? reinterpret_cast<const void*>(&rows_mlp_bwd_kernel<D, LN>)
                       : reinterpret_cast<const void*>(&rows_mlp_fwd_kernel<D, LN>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid((unsigned)rows_mlp_blocks(p.R, D), (unsigned)sides);
  if (bwd) hipLaunchKernelGGL((rows_mlp_bwd_kernel<D, LN>), grid, dim3(RowsCfg<D>::NT), lds, st, p);
  else hipLaunchKernelGGL((rows_mlp_fwd_kernel<D, LN>), grid, dim3(RowsCfg<D>::NT), lds, st, p);
  return (int)hipGetLastError();
}

int rows_mlp_dispatch(const RowsP& p, int D, int sides, int layer_norm, bool bwd, hipStream_t st) {
#define EA_ROWS(DD) return layer_norm ? launch_rows<DD, true>(p, sides, bwd, st) : launch_rows<DD, false>(p, sides, bwd, st)
  if (D == 64) EA_ROWS(64);
  if (D == 128) EA_ROWS(128);
  if (D == 32) EA_ROWS(32);
#undef EA_ROWS
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
