// ea_layernorm.hip -- LayerNorm over the channels of [rows, C] matrices with fp32 statistics and fp32 output: the LayerNorm of
// LinearRA's model-wide ('dense') landmark generators (lara.py:34-44,64-71: nn.Sequential(pool, flatten, Linear(dim, dim),
// LayerNorm(dim)) on B * L pooled rows; under autocast torch runs layer_norm in fp32 on the 16-bit Linear output).  A few
// thousand rows of 128 .. 1024 channels: one wave per row, the row in registers (two-pass mean / variance like torch's), the
// backward's d gamma / d beta as per-workgroup partial sums that ea_colsum_f32 adds in a fixed order.
#include "ea_common.h"
#include "ea_layernorm.h"

namespace ea {

namespace {

constexpr int LN_MAXE = 16;                       // channels per lane: C <= 1024

EA_DEV float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

template <int XT> EA_DEV float ln_load(const void* x, size_t i) {
  if (XT == 2) return reinterpret_cast<const float*>(x)[i];
  const uint16_t u = reinterpret_cast<const uint16_t*>(x)[i];
  return XT == 0 ? BF16::to_f(u) : F16::to_f(u);
}
template <int XT> EA_DEV void ln_store(void* x, size_t i, float v) {
  if (XT == 2) reinterpret_cast<float*>(x)[i] = v;
  else reinterpret_cast<uint16_t*>(x)[i] = XT == 0 ? BF16::from_f(v) : F16::from_f(v);
}

template <int XT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= p.rows) return;
  const int ne = p.C >> 6;
  float v[LN_MAXE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    v[i] = i < ne ? ln_load<XT>(p.x, (size_t)row * p.C + lane + 64 * i) : 0.f;
    s += v[i];
  }
  const float invC = 1.f / (float)p.C;
  const float mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const float d = i < ne ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i)
    if (i < ne) {
      const int c = lane + 64 * i;
      p.y[(size_t)row * p.C + c] = (v[i] - mean) * rstd * p.gamma[c] + p.beta[c];
    }
  if (lane == 0 && p.stats) { p.stats[2 * row] = mean; p.stats[2 * row + 1] = rstd; }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma;  d gamma = sum_rows dy xhat,  d beta = sum_rows dy
template <int XT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnP p) {
  __shared__ float red[3][2][1024];               // waves 1..3 park their (d gamma, d beta) sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ne = p.C >> 6;
  const float invC = 1.f / (float)p.C;
  float dgam[LN_MAXE], dbet[LN_MAXE], gam[LN_MAXE];
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) { dgam[i] = dbet[i] = 0.f; gam[i] = i < ne ? p.gamma[lane + 64 * i] : 0.f; }
  const int r1 = min(p.rows, (int)(blockIdx.x + 1) * p.rows_per_block);
  for (int row = blockIdx.x * p.rows_per_block + wave; row < r1; row += 4) {
    const float mean = p.stats[2 * row], rstd = p.stats[2 * row + 1];
    float xh[LN_MAXE], g[LN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      xh[i] = g[i] = 0.f;
      if (i < ne) {
        const size_t o = (size_t)row * p.C + lane + 64 * i;
        const float dy = p.dy[o];
        xh[i] = (ln_load<XT>(p.x, o) - mean) * rstd;
        g[i] = dy * gam[i];
        dgam[i] += dy * xh[i];
        dbet[i] += dy;
        s1 += g[i];
        s2 += g[i] * xh[i];
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i)
      if (i < ne) ln_store<XT>(p.dx, (size_t)row * p.C + lane + 64 * i, rstd * (g[i] - s1 - xh[i] * s2));
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i)
      if (i < ne) { red[wave - 1][0][lane + 64 * i] = dgam[i]; red[wave - 1][1][lane + 64 * i] = dbet[i]; }
  }
  __syncthreads();
  if (wave == 0) {
    float* out = p.part + (size_t)blockIdx.x * 2 * p.C;            // [blocks][2][C]
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i)
      if (i < ne) {
        const int c = lane + 64 * i;
        out[c] = ((dgam[i] + red[0][0][c]) + red[1][0][c]) + red[2][0][c];
        out[p.C + c] = ((dbet[i] + red[0][1][c]) + red[1][1][c]) + red[2][1][c];
      }
  }
}

}  // namespace

int layernorm_parts(int rows) {
  if (rows <= 0) return EA_E_BADARG;
  // >= 16 rows per workgroup (four per wave), at most one workgroup per CU-pair: the partial buffer stays small
  int blocks = (rows + 15) / 16;
  const int cap = ea_device_cus() * 2;
  return blocks > cap ? cap : blocks;
}

int layernorm_dispatch(bool bwd, const LnP& p0, int xtype, hipStream_t st) {
  LnP p = p0;
  if (p.rows <= 0 || p.C <= 0 || (p.C & 63) || p.C > 64 * LN_MAXE) return EA_E_UNSUPPORTED;
  if (!bwd) {
    const dim3 grid((unsigned)((p.rows + 3) / 4)), block(256);
    if (xtype == EA_BF16) hipLaunchKernelGGL(ln_fwd_kernel<0>, grid, block, 0, st, p);
    else if (xtype == EA_F16) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, st, p);
    else if (xtype == EA_F32) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, st, p);
    else return EA_E_BADARG;
  } else {
    const int blocks = layernorm_parts(p.rows);
    p.rows_per_block = (p.rows + blocks - 1) / blocks;
    const dim3 grid((unsigned)blocks), block(256);
    if (xtype == EA_BF16) hipLaunchKernelGGL(ln_bwd_kernel<0>, grid, block, 0, st, p);
    else if (xtype == EA_F16) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, 0, st, p);
    else if (xtype == EA_F32) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, 0, st, p);
    else return EA_E_BADARG;
  }
  return (int)hipGetLastError();
}

}  // namespace ea
