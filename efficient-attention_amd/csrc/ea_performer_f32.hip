// ea_performer_f32.hip -- the Performer (FAVOR+) baseline in EXACT fp32 arithmetic (round 4).
//
// The reference runs its linear attention in full precision whatever the AMP state
// (kernelized_attention.py:116-121: `with autocast(enabled=False)`; :343-345: q', k', v `.float()`), and a module
// called outside autocast computes everything in fp32 (abstract_attention.py:120-133).  The 16-bit kernels
// (ea_lara_x / ea_lara_y, modes L?_P*) round the feature matrices phi(q), phi(k) to the MFMA element type -- narrower
// arithmetic than the reference's own.  These kernels keep every operand in fp32: inputs of any of the three I/O types
// (bf16 / fp16 values are exact in fp32), products on v_mfma_f32_16x16x4_f32 out of fp32 LDS tiles, fp32 features,
// fp32 accumulation; outputs in the I/O type of the inputs.
//
//   phi(x)[j] = m^-1/2 exp(d^-1/4 W_j.x - d^-1/2 |x|^2 / 2 - stab) + 1e-4      (favorp_projection, :20-56)
//       stab = max_j d^-1/4 W_j.x for a query, max over (tokens, j) for the keys of a (b,h); detached
//   out_n = phi(q_n) KV / max(phi(q_n).ksum, 1e-2),  KV = sum_n phi(k_n)^T v_n,  ksum = sum_n phi(k_n)   (:116-121)
//   padded keys: phi = 0 (:337-340).
//
// One 8-wave workgroup per (b,h, sequence slice); 64-token tiles; every matrix product is tile_mm() over LDS images with an
// odd row stride (conflict-free for both operand orientations).  Sequence-wide sums (KV, ksum, their gradients) leave as
// per-slice partials, added by ea_slice_sum.  Simple on purpose: this is the faithful path, the 16-bit kernels stay the
// fast one (EA_PERFORMER_16BIT=1).
#include "ea_common.h"
#include "ea_performer_f32.h"
#include "ea_f32_mm.h"

namespace ea {

namespace {

constexpr int TB = 64;            // tokens per tile
constexpr int PD = 64;            // head dim
constexpr int LDD = PD + 1;       // row stride of the [*][64] images
constexpr int NTH = 512;          // threads per workgroup: 8 waves, two per SIMD (the LDS images allow one workgroup per CU)
constexpr int NWV = NTH / 64;     // waves
constexpr int LPR = NTH / TB;     // lanes per token row in the elementwise stages (8)

// rows n0 .. n0 + 63 of a [B,H,N,64] view -> dst[row][65] fp32 (rows >= N: zeros); thread = (row, 8 channels)
EA_DEV void load_tile(float* dst, const Pf32T& t, int b, int h, int n0, int N, int dtype, int tid) {
  const int row = tid / LPR, c0 = (tid % LPR) * 8;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = 0.f;
  if (n0 + row < N) {
    const size_t eo = (size_t)b * t.sb + (size_t)h * t.sh + (size_t)(n0 + row) * t.sn + c0;
    if (dtype == 2) {
      const float* s = reinterpret_cast<const float*>(t.p) + eo;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
      f[0] = v0[0]; f[1] = v0[1]; f[2] = v0[2]; f[3] = v0[3]; f[4] = v1[0]; f[5] = v1[1]; f[6] = v1[2]; f[7] = v1[3];
    } else {
      const u32x4 w0 = ldg16(t.p + eo * 2);
      if (dtype == 0) unpack8<BF16>(w0, f);
      else unpack8<F16>(w0, f);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[row * LDD + c0 + i] = f[i];
}

// src[row][65] fp32 -> rows n0 .. of a [B,H,N,64] view in its I/O type (rows >= N dropped)
EA_DEV void store_tile(const float* src, const Pf32T& t, int b, int h, int n0, int N, int dtype, int tid) {
  const int row = tid / LPR, c0 = (tid % LPR) * 8;
  if (n0 + row >= N) return;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = src[row * LDD + c0 + i];
  const size_t eo = (size_t)b * t.sb + (size_t)h * t.sh + (size_t)(n0 + row) * t.sn + c0;
  if (dtype == 2) {
    float* d = reinterpret_cast<float*>(t.p) + eo;
    *reinterpret_cast<f32x4*>(d) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{f[4], f[5], f[6], f[7]};
  } else {
    char* d = t.p + eo * 2;
    if (dtype == 0) stg16(d, pack8<BF16>(f));
    else stg16(d, pack8<F16>(f));
  }
}

// [rows][64] fp32 global matrix -> dst[rows][65]
EA_DEV void load_mat(float* dst, const float* src, int rows, int tid) {
  for (int idx = tid; idx < rows * (PD / 4); idx += NTH) {
    const int r = idx / (PD / 4), c = (idx - r * (PD / 4)) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)r * PD + c);
    float* d = dst + r * LDD + c;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

// sum over the LPR lanes that share a row (row = tid / LPR); lane q4 of a row walks the CONTIGUOUS part q4 of it, which with
// the odd row stride keeps the 64 lanes of a wave on distinct LDS banks
EA_DEV float row4_sum(float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
EA_DEV float row4_max(float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); return v; }

struct Consts {
  float c, c2, ratio;
};
EA_DEV Consts consts(int M) {
  Consts k;
  k.c = 0.35355339059327373f;            // 64^-1/4
  k.c2 = 0.0625f;                        // 64^-1/2 / 2
  k.ratio = rsqrtf((float)M);
  return k;
}

// Ps[n][j] = c W_j . x_n (the reference's data_dash) for the tile in Xs; diag[n] = c2 |x_n|^2
EA_DEV void logits(float* Ps, int ldp, const float* Xs, const float* Ws, float* diag, int M, float c, float c2, int tid) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int ntn = M / 16;
  for (int t = wave; t < (TB / 16) * ntn; t += NWV) {
    const int m0 = (t / ntn) * 16, n0 = (t % ntn) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    tile_mm<false, true, PD>(acc, Xs, LDD, Ws, LDD, m0, n0, PD, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Ps[(m0 + 4 * g + r) * ldp + n0 + li] = acc[r] * c;
  }
  const int row = tid / LPR, q4 = tid % LPR;
  float s = 0.f;
  for (int e = q4 * (PD / LPR); e < (q4 + 1) * (PD / LPR); ++e) { const float x = Xs[row * LDD + e]; s += x * x; }
  s = row4_sum(s);
  if (q4 == 0) diag[row] = s * c2;
}

// the token range of this workgroup
EA_DEV void slice_range(const Pf32P& p, int s, int& n0, int& n1) {
  n0 = s * p.tps;
  n1 = min(p.N, n0 + p.tps);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// keys, pass 1: partial maximum of data_dash over the slice's tokens and all features (padded keys included, as in the
// reference: the mask is applied to the features afterwards)
__global__ __launch_bounds__(NTH) void pf32_kmax_kernel(const Pf32P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = p.M, ldp = M + 1;
  float* Ws = sm;
  float* Xs = Ws + M * LDD;
  float* Ps = Xs + TB * LDD;
  float* diag = Ps + TB * ldp;
  float* red = diag + TB;
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, s = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const Consts k = consts(M);
  load_mat(Ws, p.W + (size_t)h * M * PD, M, tid);
  int n0, n1;
  slice_range(p, s, n0, n1);
  float mx = -INFINITY;
  for (int t0 = n0; t0 < n1; t0 += TB) {
    __syncthreads();
    load_tile(Xs, p.k, b, h, t0, p.N, p.dtype, tid);
    __syncthreads();
    logits(Ps, ldp, Xs, Ws, diag, M, k.c, k.c2, tid);
    __syncthreads();
    {
      const int row = tid / LPR, q4 = tid % LPR;                         // (no index divisions: four lanes per token row)
      if (t0 + row < n1)
        for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) mx = fmaxf(mx, Ps[row * ldp + j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    float m2 = red[0];
    for (int i = 1; i < NWV; ++i) m2 = fmaxf(m2, red[i]);
    p.p_max[(size_t)bh * p.S + s] = m2;
  }
}

// the key stabiliser of a (b,h): maximum of the slice maxima
EA_DEV float key_stab(const Pf32P& p, int bh) {
  float m = -INFINITY;
  for (int s = 0; s < p.S; ++s) m = fmaxf(m, p.p_max[(size_t)bh * p.S + s]);
  return m;
}

// features of the key tile in Xs (logits already in Ps): phi, 0 for padded keys and rows beyond the slice
EA_DEV void key_features(float* Ps, int ldp, const float* diag, const Pf32P& p, int b, int t0, int n1, float stab, float ratio,
                         int tid) {
  const int M = p.M;
  const int n = tid / LPR, q4 = tid % LPR, tok = t0 + n;
  const bool live = tok < n1 && !(p.mask && p.mask[(size_t)b * p.N + tok]);
  const float sh = diag[n] + stab;
  for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) Ps[n * ldp + j] = live ? ratio * __expf(Ps[n * ldp + j] - sh) + 1e-4f : 0.f;
}

// keys, pass 2: partial KV [M][64] and ksum [M] of the slice
__global__ __launch_bounds__(NTH) void pf32_kv_kernel(const Pf32P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = p.M, ldp = M + 1;
  float* Ws = sm;
  float* Xs = Ws + M * LDD;
  float* Ys = Xs + TB * LDD;
  float* Ps = Ys + TB * LDD;
  float* diag = Ps + TB * ldp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = blockIdx.x, s = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const Consts k = consts(M);
  load_mat(Ws, p.W + (size_t)h * M * PD, M, tid);
  const float stab = key_stab(p, bh);
  int n0, n1;
  slice_range(p, s, n0, n1);
  const int nt = M / 16;                        // feature tiles; wave w owns (jt, et) pairs t = w, w + 4, ..: nt of them
  f32x4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ks = 0.f;
  for (int t0 = n0; t0 < n1; t0 += TB) {
    __syncthreads();
    load_tile(Xs, p.k, b, h, t0, p.N, p.dtype, tid);
    load_tile(Ys, p.v, b, h, t0, p.N, p.dtype, tid);
    __syncthreads();
    logits(Ps, ldp, Xs, Ws, diag, M, k.c, k.c2, tid);
    __syncthreads();
    key_features(Ps, ldp, diag, p, b, t0, n1, stab, k.ratio, tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int t = wave + NWV * i;
      if (t < nt * 4) tile_mm<true, false, TB>(acc[i], Ps, ldp, Ys, LDD, (t >> 2) * 16, (t & 3) * 16, TB, lane);
    }
    if (tid < M) {
      float a = 0.f;
      for (int n = 0; n < TB; ++n) a += Ps[n * ldp + tid];
      ks += a;
    }
  }
  float* okv = p.p_kv + ((size_t)bh * p.S + s) * M * PD;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = wave + NWV * i;
    if (t < nt * 4) {
      const int j0 = (t >> 2) * 16, e0 = (t & 3) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) okv[(size_t)(j0 + 4 * g + r) * PD + e0 + li] = acc[i][r];
    }
  }
  if (tid < M) p.p_ks[((size_t)bh * p.S + s) * M + tid] = ks;
}

// query features of the tile in Xs: Ps <- phi(q), den[n] = phi(q_n) . ksum
EA_DEV void query_features(float* Ps, int ldp, const float* diag, float* den, const float* ksum_s, int M, float ratio, int tid) {
  const int row = tid / LPR, q4 = tid % LPR;
  float mx = -INFINITY;
  for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) mx = fmaxf(mx, Ps[row * ldp + j]);
  mx = row4_max(mx);
  const float sh = diag[row] + mx;
  float dn = 0.f;
  for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) {
    const float f = ratio * __expf(Ps[row * ldp + j] - sh) + 1e-4f;
    Ps[row * ldp + j] = f;
    dn += f * ksum_s[j];
  }
  dn = row4_sum(dn);
  if (q4 == 0) den[row] = dn;
}

// Os[n][e] = (sum_j Ps[n][j] KVs[j][e]) * rowscale(n)
template <typename F>
EA_DEV void feat_times(float* Os, const float* Ps, int ldp, const float* KVs, int M, int tid, F rowscale) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  for (int t = wave; t < 16; t += NWV) {
    const int m0 = (t >> 2) * 16, n0 = (t & 3) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (M == 64) tile_mm<false, false, 64>(acc, Ps, ldp, KVs, LDD, m0, n0, 64, lane);
    else tile_mm<false, false>(acc, Ps, ldp, KVs, LDD, m0, n0, M, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(m0 + 4 * g + r) * LDD + n0 + li] = acc[r] * rowscale(m0 + 4 * g + r);
  }
}

// queries, forward: out
__global__ __launch_bounds__(NTH) void pf32_out_kernel(const Pf32P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = p.M, ldp = M + 1;
  float* Ws = sm;
  float* KVs = Ws + M * LDD;
  float* Xs = KVs + M * LDD;
  float* Os = Xs;                                // the q tile is dead once its logits and |q|^2 are formed
  float* Ps = Xs + TB * LDD;
  float* diag = Ps + TB * ldp;
  float* den = diag + TB;
  float* ksum_s = den + TB;
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, s = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const Consts k = consts(M);
  load_mat(Ws, p.W + (size_t)h * M * PD, M, tid);
  load_mat(KVs, p.kv + (size_t)bh * M * PD, M, tid);
  if (tid < M) ksum_s[tid] = p.ksum[(size_t)bh * M + tid];
  int n0, n1;
  slice_range(p, s, n0, n1);
  for (int t0 = n0; t0 < n1; t0 += TB) {
    __syncthreads();
    load_tile(Xs, p.q, b, h, t0, p.N, p.dtype, tid);
    __syncthreads();
    logits(Ps, ldp, Xs, Ws, diag, M, k.c, k.c2, tid);
    __syncthreads();
    query_features(Ps, ldp, diag, den, ksum_s, M, k.ratio, tid);
    __syncthreads();
    feat_times(Os, Ps, ldp, KVs, M, tid, [&](int n) { return 1.f / fmaxf(den[n], 1e-2f); });
    __syncthreads();
    store_tile(Os, p.o, b, h, t0, n1, p.dtype, tid);
  }
}

// dX[n][e] = c sum_j Gs[n][j] Ws[j][e] - 2 c2 sdl[n] Xs[n][e]  -> Os   (gradient through the logits and the -|x|^2 term)
EA_DEV void logit_grad(float* Os, const float* Gs, int ldp, const float* Ws, const float* Xs, const float* sdl, int M, float c,
                       float c2, int tid) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  for (int t = wave; t < 16; t += NWV) {
    const int m0 = (t >> 2) * 16, n0 = (t & 3) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (M == 64) tile_mm<false, false, 64>(acc, Gs, ldp, Ws, LDD, m0, n0, 64, lane);
    else tile_mm<false, false>(acc, Gs, ldp, Ws, LDD, m0, n0, M, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = m0 + 4 * g + r, e = n0 + li;
      Os[n * LDD + e] = c * acc[r] - 2.f * c2 * sdl[n] * Xs[n * LDD + e];
    }
  }
}

// queries, backward: dq and the partial d KV [M][64], d ksum [M] of the slice
__global__ __launch_bounds__(NTH) void pf32_bwd_q_kernel(const Pf32P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = p.M, ldp = M + 1;
  float* Ws = sm;
  float* KVs = Ws + M * LDD;
  float* Xs = KVs + M * LDD;
  float* Ys = Xs + TB * LDD;                     // dout, then d num
  float* Ps = Ys + TB * LDD;                     // phi(q)
  float* Gs = Ps + TB * ldp;                     // out tile (first 64 x 65 floats), then d phi / d logits
  float* diag = Gs + TB * (ldp > LDD ? ldp : LDD);
  float* den = diag + TB;
  float* dden = den + TB;
  float* sdl = dden + TB;
  float* ksum_s = sdl + TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = blockIdx.x, s = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const Consts k = consts(M);
  load_mat(Ws, p.W + (size_t)h * M * PD, M, tid);
  load_mat(KVs, p.kv + (size_t)bh * M * PD, M, tid);
  if (tid < M) ksum_s[tid] = p.ksum[(size_t)bh * M + tid];
  int n0, n1;
  slice_range(p, s, n0, n1);
  const int nt = M / 16;
  f32x4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dks = 0.f;
  const int row = tid / LPR, q4 = tid % LPR;
  for (int t0 = n0; t0 < n1; t0 += TB) {
    __syncthreads();
    load_tile(Xs, p.q, b, h, t0, p.N, p.dtype, tid);
    load_tile(Ys, p.dout, b, h, t0, n1, p.dtype, tid);          // rows beyond the slice: zero cotangent
    __syncthreads();
    logits(Ps, ldp, Xs, Ws, diag, M, k.c, k.c2, tid);
    __syncthreads();
    query_features(Ps, ldp, diag, den, ksum_s, M, k.ratio, tid);
    __syncthreads();
    float* Os = Gs;
    feat_times(Os, Ps, ldp, KVs, M, tid, [&](int n) { return 1.f / fmaxf(den[n], 1e-2f); });     // out
    __syncthreads();
    {
      // d num = dout / max(den, 1e-2);  d den = -(dout . out) / max(den, 1e-2) where the clamp is inactive
      const float inv = 1.f / fmaxf(den[row], 1e-2f);
      float rd = 0.f;
      for (int e = q4 * (PD / LPR); e < (q4 + 1) * (PD / LPR); ++e) rd += Ys[row * LDD + e] * Os[row * LDD + e];
      rd = row4_sum(rd);
      for (int e = q4 * (PD / LPR); e < (q4 + 1) * (PD / LPR); ++e) Ys[row * LDD + e] *= inv;
      if (q4 == 0) dden[row] = den[row] > 1e-2f ? -rd * inv : 0.f;
    }
    __syncthreads();
    // d phi[n][j] = d num[n] . KV[j] + d den[n] ksum[j];  d logit = d phi (phi - eps)
    for (int t = wave; t < (TB / 16) * nt; t += NWV) {
      const int m0 = (t / nt) * 16, j0 = (t % nt) * 16;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, true, PD>(a, Ys, LDD, KVs, LDD, m0, j0, PD, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = m0 + 4 * g + r, j = j0 + li;
        Gs[n * ldp + j] = (a[r] + dden[n] * ksum_s[j]) * (Ps[n * ldp + j] - 1e-4f);
      }
    }
    __syncthreads();
    {
      float sacc = 0.f;
      for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) sacc += Gs[row * ldp + j];
      sacc = row4_sum(sacc);
      if (q4 == 0) sdl[row] = sacc;
    }
    // partial d KV[j][e] += sum_n phi[n][j] d num[n][e];  d ksum[j] += sum_n phi[n][j] d den[n]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int t = wave + NWV * i;
      if (t < nt * 4) tile_mm<true, false, TB>(acc[i], Ps, ldp, Ys, LDD, (t >> 2) * 16, (t & 3) * 16, TB, lane);
    }
    if (tid < M) {
      float a = 0.f;
      for (int n = 0; n < TB; ++n) a += Ps[n * ldp + tid] * dden[n];
      dks += a;
    }
    __syncthreads();
    logit_grad(Ys, Gs, ldp, Ws, Xs, sdl, M, k.c, k.c2, tid);      // dq tile (d num is dead)
    __syncthreads();
    store_tile(Ys, p.dq, b, h, t0, n1, p.dtype, tid);
  }
  float* okv = p.p_kv + ((size_t)bh * p.S + s) * M * PD;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int t = wave + NWV * i;
    if (t < nt * 4) {
      const int j0 = (t >> 2) * 16, e0 = (t & 3) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) okv[(size_t)(j0 + 4 * g + r) * PD + e0 + li] = acc[i][r];
    }
  }
  if (tid < M) p.p_ks[((size_t)bh * p.S + s) * M + tid] = dks;
}

// keys, backward: dk, dv from the summed d KV, d ksum
__global__ __launch_bounds__(NTH) void pf32_bwd_k_kernel(const Pf32P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = p.M, ldp = M + 1;
  float* Ws = sm;
  float* KVs = Ws + M * LDD;                     // d KV
  float* Xs = KVs + M * LDD;
  float* Ys = Xs + TB * LDD;                     // v, then dv / dk tiles
  float* Ps = Ys + TB * LDD;                     // phi(k)
  float* Gs = Ps + TB * ldp;                     // d phi -> d logits
  float* diag = Gs + TB * ldp;
  float* sdl = diag + TB;
  float* dks_s = sdl + TB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int bh = blockIdx.x, s = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const Consts k = consts(M);
  load_mat(Ws, p.W + (size_t)h * M * PD, M, tid);
  load_mat(KVs, p.dkv + (size_t)bh * M * PD, M, tid);
  if (tid < M) dks_s[tid] = p.dksum[(size_t)bh * M + tid];
  const float stab = key_stab(p, bh);
  int n0, n1;
  slice_range(p, s, n0, n1);
  const int nt = M / 16;
  const int row = tid / LPR, q4 = tid % LPR;
  for (int t0 = n0; t0 < n1; t0 += TB) {
    __syncthreads();
    load_tile(Xs, p.k, b, h, t0, p.N, p.dtype, tid);
    load_tile(Ys, p.v, b, h, t0, p.N, p.dtype, tid);
    __syncthreads();
    logits(Ps, ldp, Xs, Ws, diag, M, k.c, k.c2, tid);
    __syncthreads();
    key_features(Ps, ldp, diag, p, b, t0, n1, stab, k.ratio, tid);
    __syncthreads();
    // d phi[n][j] = v[n] . dKV[j] + d ksum[j];  d logit = d phi (phi - eps), 0 where phi was masked to 0
    for (int t = wave; t < (TB / 16) * nt; t += NWV) {
      const int m0 = (t / nt) * 16, j0 = (t % nt) * 16;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, true, PD>(a, Ys, LDD, KVs, LDD, m0, j0, PD, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = m0 + 4 * g + r, j = j0 + li;
        const float f = Ps[n * ldp + j];
        Gs[n * ldp + j] = f > 0.f ? (a[r] + dks_s[j]) * (f - 1e-4f) : 0.f;
      }
    }
    __syncthreads();
    {
      float sacc = 0.f;
      for (int j = q4 * (M / LPR), je = j + (M / LPR); j < je; ++j) sacc += Gs[row * ldp + j];
      sacc = row4_sum(sacc);
      if (q4 == 0) sdl[row] = sacc;
    }
    // dv[n] = sum_j phi[n][j] dKV[j]   (v is dead: its tile takes dv)
    feat_times(Ys, Ps, ldp, KVs, M, tid, [](int) { return 1.f; });
    __syncthreads();
    store_tile(Ys, p.dv, b, h, t0, n1, p.dtype, tid);
    __syncthreads();
    logit_grad(Ys, Gs, ldp, Ws, Xs, sdl, M, k.c, k.c2, tid);
    __syncthreads();
    store_tile(Ys, p.dk, b, h, t0, n1, p.dtype, tid);
  }
}

// ------------------------------------------------------------------------------------------------------------
int pf32_slices(int BH, int N) {
  const int tiles = (N + TB - 1) / TB;
  int S = (1024 + BH - 1) / BH;
  if (S > tiles) S = tiles;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  return S;
}

static size_t pf32_lds(int which, int M) {
  const size_t ldp = M + 1, big = ldp > LDD ? ldp : LDD;
  switch (which) {
    case 0: return (M * LDD + TB * LDD + TB * ldp + TB + 256) * sizeof(float);
    case 1: return (M * LDD + 2 * TB * LDD + TB * ldp + TB) * sizeof(float);
    case 2: return (2 * M * LDD + TB * LDD + TB * ldp + 2 * TB + M) * sizeof(float);
    case 3: return (2 * M * LDD + 2 * TB * LDD + TB * ldp + TB * big + 4 * TB + M) * sizeof(float);
    default: return (2 * M * LDD + 2 * TB * LDD + 2 * TB * ldp + 2 * TB + M) * sizeof(float);
  }
}

int pf32_dispatch(int which, const Pf32P& p0, hipStream_t st) {
  Pf32P p = p0;
  if (p.M <= 0 || p.M > 96 || (p.M & 15) || p.dtype < 0 || p.dtype > 2) return EA_E_UNSUPPORTED;
  p.S = pf32_slices(p.B * p.H, p.N);
  const int tiles = (p.N + TB - 1) / TB;
  p.tps = ((tiles + p.S - 1) / p.S) * TB;
  const dim3 grid((unsigned)(p.B * p.H), (unsigned)p.S), block(NTH);
  const size_t lds = pf32_lds(which, p.M);
  switch (which) {
    case 0:
      EA_SET_LDS_ONCE((&pf32_kmax_kernel), lds);
      hipLaunchKernelGGL(pf32_kmax_kernel, grid, block, lds, st, p);
      break;
    case 1:
      EA_SET_LDS_ONCE((&pf32_kv_kernel), lds);
      hipLaunchKernelGGL(pf32_kv_kernel, grid, block, lds, st, p);
      break;
    case 2:
      EA_SET_LDS_ONCE((&pf32_out_kernel), lds);
      hipLaunchKernelGGL(pf32_out_kernel, grid, block, lds, st, p);
      break;
    case 3:
      EA_SET_LDS_ONCE((&pf32_bwd_q_kernel), lds);
      hipLaunchKernelGGL(pf32_bwd_q_kernel, grid, block, lds, st, p);
      break;
    case 4:
      EA_SET_LDS_ONCE((&pf32_bwd_k_kernel), lds);
      hipLaunchKernelGGL(pf32_bwd_k_kernel, grid, block, lds, st, p);
      break;
    default:
      return EA_E_BADARG;
  }
  return (int)hipGetLastError();
}

}  // namespace ea
