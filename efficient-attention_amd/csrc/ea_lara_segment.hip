// ea_lara_segment.hip -- LARA's 1-D landmark proposals (lara.py:84-127): the segment means of
//     LayerNorm(Linear(q)), LayerNorm(Linear(k))      ('adaptive-1d')   or of q, k themselves,
// with the reference's even / uneven split of N tokens into L segments.  The Linear itself is folded
// into the qkv projection by the host module (two more groups of output columns of the same GEMM,
// W' = Wgen Wq), so this kernel reads those pre-LayerNorm rows once, normalises each row in
// registers and averages; nothing token-sized is written in the forward.  The backward recomputes
// the row statistics, writes the gradient of the pre-LayerNorm rows and leaves the LayerNorm / bias
// parameter gradients as per-segment partial sums.
// One wave per (b, h, segment, q|k); a row (D elements) is spread over D/8 lanes (16-B loads).
#include "ea_lara_segment.h"

namespace ea {

template <int CPR> EA_DEV float row_sum(float v) { return group_sum<CPR>(v); }           // over the CPR lanes that share a row
template <int CPR> EA_DEV float col_sum(float v) { return stride_sum<CPR>(v); }          // over the 64/CPR row groups (same channels)

template <typename E, int D, bool BWD>
__global__ __launch_bounds__(256) void lara_segment_kernel(const SegP p) {
  constexpr int CPR = D / 8, RPI = 64 / CPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= p.B * p.H * p.L * 2) return;
  const int side = unit & 1, rest = unit >> 1;
  const int l = rest % p.L, bh = rest / p.L, b = bh / p.H, h = bh - b * p.H;
  const int s0 = l < p.nshort ? l * p.segs : p.nshort * p.segs + (l - p.nshort) * (p.segs + 1);
  const int len = l < p.nshort ? p.segs : p.segs + 1;
  const int rl = lane / CPR, cl = lane - rl * CPR;
  const char* src = side ? p.k + (b * p.k_sb + h * p.k_sh) * 2 : p.q + (b * p.q_sb + h * p.q_sh) * 2;
  const int sn = (int)(side ? p.k_sn : p.q_sn);
  const float* gam = side ? p.gk : p.gq;
  const float* bet = side ? p.ck : p.cq;
  const float* bias = side ? p.bias_k : p.bias_q;
  const float* mbias = side ? p.mbias_k : p.mbias_q;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.N : nullptr;
  const bool ln = gam != nullptr;
  float g8[8], c8[8], b8[8], m8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    g8[i] = ln ? gam[cl * 8 + i] : 1.f;
    c8[i] = ln ? bet[cl * 8 + i] : 0.f;
    b8[i] = bias ? bias[h * D + cl * 8 + i] : 0.f;
    m8[i] = mbias ? mbias[cl * 8 + i] : 0.f;
  }
  const float inv_len = 1.f / (float)len;
  float dy8[8];
  char* dst = nullptr;
  int dsn = 0;
  if (BWD) {
    const float* dbar = (side ? p.d_kbar : p.d_qbar) + ((size_t)bh * p.L + l) * D + cl * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) dy8[i] = dbar[i] * inv_len;
    dst = side ? p.dk + (b * p.dk_sb + h * p.dk_sh) * 2 : p.dq + (b * p.dq_sb + h * p.dq_sh) * 2;
    dsn = (int)(side ? p.dk_sn : p.dq_sn);
  }
  float acc[8], a_dg[8], a_db[8], a_bu[8], a_bm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = a_dg[i] = a_db[i] = a_bu[i] = a_bm[i] = 0.f;

  // one row per lane group and step; two steps are fetched together so that their global loads and
  // the shuffle reductions of the LayerNorm statistics overlap
  auto row = [&](const u32x4 raw, int tok, bool valid, bool masked) {
    float f[8];
    unpack8<E>(raw, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] += masked ? m8[i] : b8[i];
    float xh[8], rs = 1.f;
    if (ln) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += f[i];
      const float mean = row_sum<CPR>(s) * (1.f / D);
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { xh[i] = f[i] - mean; v += xh[i] * xh[i]; }
      rs = rsqrtf(row_sum<CPR>(v) * (1.f / D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 8; ++i) xh[i] *= rs;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) xh[i] = f[i];
    }
    if (!BWD) {
      const float w = valid ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += w * (g8[i] * xh[i] + c8[i]);
    } else {
      float dh[8];
      if (ln) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dh[i] = dy8[i] * g8[i]; s1 += dh[i]; s2 += dh[i] * xh[i]; }
        s1 = row_sum<CPR>(s1) * (1.f / D);
        s2 = row_sum<CPR>(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < 8; ++i) dh[i] = rs * (dh[i] - s1 - xh[i] * s2);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) dh[i] = dy8[i];
      }
      if (valid) {
        stg16(dst + (tok * dsn + cl * 8) * 2, pack8<E>(dh));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          a_dg[i] += dy8[i] * xh[i];
          a_db[i] += dy8[i];
          if (masked) a_bm[i] += dh[i]; else a_bu[i] += dh[i];
        }
      }
    }
  };
  const int iters = (len + RPI - 1) / RPI;
  for (int it = 0; it < iters; it += 2) {
    const int off0 = it * RPI + rl, off1 = off0 + RPI;
    const bool v0 = off0 < len, v1 = off1 < len;
    const int tok0 = s0 + (v0 ? off0 : 0), tok1 = s0 + (v1 ? off1 : 0);     // clamped: unconditional loads
    const u32x4 r0 = ldg16(src + (tok0 * sn + cl * 8) * 2);
    const u32x4 r1 = ldg16(src + (tok1 * sn + cl * 8) * 2);
    const bool k0 = mrow && mrow[tok0], k1 = mrow && mrow[tok1];
    row(r0, tok0, v0, k0);
    row(r1, tok1, v1, k1);
  }
  if (!BWD) {
    float* out = (side ? p.kbar : p.qbar) + ((size_t)bh * p.L + l) * D + cl * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = col_sum<CPR>(acc[i]) * inv_len;
    if (rl == 0) {
      *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(out + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  } else if (p.part) {
    float* pr = p.part + (((size_t)bh * p.L + l) * 2 + side) * 4 * D + cl * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a_dg[i] = col_sum<CPR>(a_dg[i]); a_db[i] = col_sum<CPR>(a_db[i]);
      a_bu[i] = col_sum<CPR>(a_bu[i]); a_bm[i] = col_sum<CPR>(a_bm[i]);
    }
    if (rl == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { pr[i] = a_dg[i]; pr[D + i] = a_db[i]; pr[2 * D + i] = a_bu[i]; pr[3 * D + i] = a_bm[i]; }
    }
  }
}

template <typename E, int D>
static int launch_seg(bool bwd, const SegP& p, hipStream_t st) {
  const long units = (long)p.B * p.H * p.L * 2;
  const dim3 grid((unsigned)((units + 3) / 4));
  if (bwd) hipLaunchKernelGGL((lara_segment_kernel<E, D, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((lara_segment_kernel<E, D, false>), grid, dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

int lara_segment_dispatch(bool bwd, const SegP& p, int dtype, int D, hipStream_t st) {
  if (dtype == EA_BF16) {
    if (D == 64) return launch_seg<BF16, 64>(bwd, p, st);
    if (D == 32) return launch_seg<BF16, 32>(bwd, p, st);
  } else if (dtype == EA_F16) {
    if (D == 64) return launch_seg<F16, 64>(bwd, p, st);
    if (D == 32) return launch_seg<F16, 32>(bwd, p, st);
  }
  return EA_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// nn.AdaptiveAvgPool2d of one of q / k / v over the token grid when the grid does not divide evenly
// (lara.py:43,48,145-151): bin o of an axis of n cells covers [floor(o n / side), ceil((o + 1) n / side)) --
// neighbouring bins share a row / column.  (Evenly dividing grids take the chunk-mean kernels.)
// ------------------------------------------------------------------------------------------
struct PoolP {
  const char* x;          // forward input [B,H,N,D] view (element type)
  char* dx;               // backward: accumulated into
  long sb, sh, sn;
  float* mean;            // [B,H,side*side,D] fp32 (forward output / backward input)
  int B, H, gh, gw, side, D;
};
EA_DEV int bin_lo(int o, int n, int side) { return (o * n) / side; }
EA_DEV int bin_hi(int o, int n, int side) { return ((o + 1) * n + side - 1) / side; }

template <typename E>
__global__ __launch_bounds__(64) void pool2d_fwd_kernel(const PoolP p) {
  const int L = p.side * p.side;
  const int bin = blockIdx.x % L, bh = blockIdx.x / L, b = bh / p.H, h = bh - b * p.H;
  const int oy = bin / p.side, ox = bin - oy * p.side;
  const int y0 = bin_lo(oy, p.gh, p.side), y1 = bin_hi(oy, p.gh, p.side);
  const int x0 = bin_lo(ox, p.gw, p.side), x1 = bin_hi(ox, p.gw, p.side);
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.x) + b * p.sb + h * p.sh;
  const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
  for (int ch = threadIdx.x; ch < p.D; ch += 64) {
    float a = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) a += E::to_f(src[(size_t)(y * p.gw + x) * p.sn + ch]);
    p.mean[((size_t)bh * L + bin) * p.D + ch] = a * inv;
  }
}

template <typename E>
__global__ __launch_bounds__(64) void pool2d_bwd_kernel(const PoolP p) {
  const int N = p.gh * p.gw, L = p.side * p.side;
  const int n = blockIdx.x % N, bh = blockIdx.x / N, b = bh / p.H, h = bh - b * p.H;
  const int y = n / p.gw, x = n - y * p.gw;
  uint16_t* dst = reinterpret_cast<uint16_t*>(p.dx) + b * p.sb + h * p.sh + (size_t)n * p.sn;
  for (int ch = threadIdx.x; ch < p.D; ch += 64) {
    float a = 0.f;
    for (int oy = 0; oy < p.side; ++oy) {
      const int y0 = bin_lo(oy, p.gh, p.side), y1 = bin_hi(oy, p.gh, p.side);
      if (y < y0 || y >= y1) continue;
      for (int ox = 0; ox < p.side; ++ox) {
        const int x0 = bin_lo(ox, p.gw, p.side), x1 = bin_hi(ox, p.gw, p.side);
        if (x < x0 || x >= x1) continue;
        a += p.mean[((size_t)bh * L + oy * p.side + ox) * p.D + ch] / (float)((y1 - y0) * (x1 - x0));
      }
    }
    dst[ch] = E::from_f(E::to_f(dst[ch]) + a);
  }
}

int pool2d_dispatch(bool bwd, int dtype, const void* x, long sb, long sh, long sn, float* mean, int B, int H, int gh, int gw,
                    int side, int D, hipStream_t st) {
  if (B <= 0 || H <= 0 || gh <= 0 || gw <= 0 || side <= 0 || side > gh || side > gw || D <= 0) return EA_E_BADARG;
  PoolP p;
  p.x = (const char*)x; p.dx = (char*)const_cast<void*>(x); p.sb = sb; p.sh = sh; p.sn = sn; p.mean = mean;
  p.B = B; p.H = H; p.gh = gh; p.gw = gw; p.side = side; p.D = D;
  const unsigned grid = (unsigned)(B * H * (bwd ? gh * gw : side * side));
  if (dtype == EA_BF16) {
    if (bwd) hipLaunchKernelGGL(pool2d_bwd_kernel<BF16>, dim3(grid), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(pool2d_fwd_kernel<BF16>, dim3(grid), dim3(64), 0, st, p);
  } else if (dtype == EA_F16) {
    if (bwd) hipLaunchKernelGGL(pool2d_bwd_kernel<F16>, dim3(grid), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(pool2d_fwd_kernel<F16>, dim3(grid), dim3(64), 0, st, p);
  } else {
    return EA_E_BADARG;
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Sampling + the [C x L] proposal-density algebra on the landmarks when the sample count exceeds what the fused
// landmark kernels hold (C > 64: antithetic / multi-sample draws at L = 49; lara.py:187-238):
//     omega_c = mu_{c mod L} +- noise,   prm[c][n] = s <omega_c, mu_n> - s |mu_n|^2 / 2     (n < L)
//     mis-opt   : lp_c = prm[c][c mod L],  bhv_c = exp(lp_c - logsumexp_n prm[c][n]) / nrep,  qbar_rows = rep(q_bar)
//     mis-biased: lp_c = logsumexp_n prm[c][n],                                               qbar_rows = rep(mu)
//     mis-bh    : lp_c = logsumexp_n prm[c][n]
// (the reference takes the log-sum-exp of mis-opt over the C replicated rows of mu: nrep copies of the same L values).
// fp32 throughout, one workgroup per (b,h), mu / omega rows in LDS: tiny next to the estimator passes.
// ------------------------------------------------------------------------------------------

EA_DEV float samp_omega(const SampP& p, const float* mu_s, size_t bh, int c, int d) {
  const int l = c % p.L;
  float v = mu_s[l * (p.D + 1) + d];
  if (p.noise) {
    if (p.mode == 1) v += (c < p.L ? 1.f : -1.f) * p.noise[(bh * p.L + l) * p.D + d];
    else v += p.noise[(bh * p.C + c) * p.D + d];
  }
  return v;
}

template <bool BWD>
__global__ __launch_bounds__(256) void lara_sample_kernel(const SampP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int L = p.L, C = p.C, D = p.D, LD = D + 1, nrep = C / L;
  float* mu_s = reinterpret_cast<float*>(smem);          // [L][D+1]
  float* om_s = mu_s + L * LD;                           // [C][D+1]
  float* m2_s = om_s + C * LD;                           // [L]  s |mu_n|^2 / 2
  float* pr_s = m2_s + ((L + 3) & ~3);                   // backward: [C][L+1] prm -> d prm
  const size_t bh = blockIdx.x;
  const int tid = threadIdx.x;
  const float s = p.scale;
  for (int i = tid; i < L * D; i += 256) mu_s[(i / D) * LD + i % D] = p.mu[bh * L * D + i];
  __syncthreads();
  for (int i = tid; i < C * D; i += 256) {
    const int c = i / D, d = i - c * D;
    const float v = samp_omega(p, mu_s, bh, c, d);
    om_s[c * LD + d] = v;
    if (!BWD) {
      p.omega[(bh * C + c) * D + d] = v;
      if (p.qrows) p.qrows[(bh * C + c) * D + d] = p.mis == EA_MIS_OPT ? p.qbar[(bh * L + c % L) * D + d] : mu_s[(c % L) * LD + d];
    }
  }
  for (int n = tid; n < L; n += 256) {
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += mu_s[n * LD + d] * mu_s[n * LD + d];
    m2_s[n] = 0.5f * s * a;
  }
  __syncthreads();
  // one thread per sample row c: prm[c][:], its log-sum-exp, the diagonal entry
  float lse = 0.f, diag = 0.f;
  const int c = tid;
  if (c < C) {
    float mx = -INFINITY, den = 0.f;
    for (int n = 0; n < L; ++n) {
      float a = 0.f;
      for (int d = 0; d < D; ++d) a += om_s[c * LD + d] * mu_s[n * LD + d];
      a = s * a - m2_s[n];
      if (BWD) pr_s[c * (L + 1) + n] = a;
      if (n == c % L) diag = a;
      const float m = fmaxf(mx, a);
      den = den * __expf(mx - m) + __expf(a - m);
      mx = m;
    }
    lse = mx + __logf(den);
    if (!BWD) {
      if (p.mis == EA_MIS_OPT) {
        p.lp[bh * C + c] = diag;
        if (p.bhv) p.bhv[bh * C + c] = __expf(diag - lse) / (float)nrep;
      } else {
        p.lp[bh * C + c] = lse;
      }
    }
  }
  if (!BWD) return;
  // d prm[c][n] in place
  if (c < C) {
    const float dlp = p.d_lp ? p.d_lp[bh * C + c] : 0.f;
    float dl = 0.f, dd = 0.f;                              // coefficient of P[c][:] and of the diagonal entry
    if (p.mis == EA_MIS_OPT) {
      const float gb = p.d_bhv ? p.d_bhv[bh * C + c] * __expf(diag - lse) / (float)nrep : 0.f;
      dd = dlp + gb; dl = -gb;
    } else {
      dl = dlp;
    }
    for (int n = 0; n < L; ++n) {
      const float pcn = __expf(pr_s[c * (L + 1) + n] - lse);
      pr_s[c * (L + 1) + n] = dl * pcn + (n == c % L ? dd : 0.f);
    }
  }
  __syncthreads();
  // one thread per (n, d): d mu_n[d] = s sum_c dprm[c][n] (omega_c - mu_n)[d] + sum over the replicas c = n + k L of
  // d omega_c[d] (= s sum_n' dprm[c][n'] mu_n'[d] + incoming d_omega) [+ d_qrows of mis-biased]; d q_bar from d_qrows
  for (int i = tid; i < L * D; i += 256) {
    const int n = i / D, d = i - n * D;
    float a = 0.f;
    for (int cc = 0; cc < C; ++cc) a += pr_s[cc * (L + 1) + n] * (om_s[cc * LD + d] - mu_s[n * LD + d]);
    a *= s;
    float dq = 0.f;
    for (int k = 0; k < nrep; ++k) {
      const int cc = n + k * L;
      float t = 0.f;
      for (int n2 = 0; n2 < L; ++n2) t += pr_s[cc * (L + 1) + n2] * mu_s[n2 * LD + d];
      a += s * t + p.d_omega[(bh * C + cc) * D + d];
      if (p.d_qrows) {
        const float g = p.d_qrows[(bh * C + cc) * D + d];
        if (p.mis == EA_MIS_OPT) dq += g; else a += g;
      }
    }
    p.d_mu[bh * L * D + i] = a;
    p.d_qbar[bh * L * D + i] = dq;
  }
}

size_t lara_sample_lds(int L, int C, int D, bool bwd) {
  return ((size_t)(L + C) * (D + 1) + ((L + 3) & ~3) + (bwd ? (size_t)C * (L + 1) : 0)) * 4;
}

int lara_sample_dispatch(bool bwd, const SampP& p, hipStream_t st) {
  if (p.BH <= 0 || p.L <= 0 || p.C <= 0 || p.C > 256 || p.C % p.L || p.D <= 0) return EA_E_UNSUPPORTED;
  const size_t lds = lara_sample_lds(p.L, p.C, p.D, bwd);
  if (lds > 150 * 1024) return EA_E_UNSUPPORTED;
  if (bwd) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lara_sample_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(lara_sample_kernel<true>, dim3((unsigned)p.BH), dim3(256), lds, st, p);
  } else {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lara_sample_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(lara_sample_kernel<false>, dim3((unsigned)p.BH), dim3(256), lds, st, p);
  }
  return (int)hipGetLastError();
}

}  // namespace ea
