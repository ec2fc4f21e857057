// ea_eva_landmark.hip -- EVA landmark statistics (eva.py:155-196) as streaming kernels.
//
//   chunk_mean : masked means of q and k over every landmark chunk      (eva.py:160-180)
//   beta       : beta_c = softmax_j(s w_c.k_j - s|k_j|^2/2) . v_j        (eva.py:192-196)
// and their backward passes.  These are HBM/L2-bound O(N*D) passes with no matmul shape, so
// they are plain VALU kernels: a row of D channels is spread over D/8 lanes (16-byte loads),
// 64/(D/8) rows are in flight per wave step.  Short chunks (2-D pooling: 16-64 tokens) get one
// wave each (WPC = 1); long chunks (1-D sequences: hundreds of tokens per landmark, where
// B*h*L waves would not fill the chip) are shared by the four waves of a workgroup (WPC = 4) whose
// partial results are merged through LDS in a fixed order.
//
// The chunk partition (rearrange / pad + as_strided copies in the reference) is address
// arithmetic (part_token); slots outside the sequence and padded tokens count as zeros in the
// means and get the finite -5e4 logit, exactly like the reference's masked_fill sequence.
#include <stdlib.h>
#include "ea_landmark_params.h"

namespace ea {


// reduce across the lanes that hold the same channel chunk (stride CPR in lane id)
template <int CPR> EA_DEV float rows_sum(float v) { return stride_sum<CPR>(v); }
// reduce across the CPR lanes of one row
template <int CPR> EA_DEV float chan_sum(float v) { return group_sum<CPR>(v); }

// row steps requested per memory round trip by the streaming loops below (round 6): long chunks shared by four waves stream
// hundreds of rows per wave, one-wave chunks at most 16 row steps
template <int WPC> struct LmU { static constexpr int value = WPC == 4 ? 8 : 4; };

// ------------------------------------------------------------------------------------------
template <typename E, int D, int WPC>
__global__ __launch_bounds__(256) void chunk_mean_fwd_kernel(const LmP p) {
  constexpr int CPR = D / 8, RPW = 64 / CPR;
  const int lane = threadIdx.x & 63, c = lane % CPR, rg = lane / CPR, wave = threadIdx.x >> 6;
  const int sub = WPC == 1 ? 0 : wave;
  const long chunk_id = WPC == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x;
  if (chunk_id >= (long)p.B * p.H * p.L) return;
  const int cidx = (int)(chunk_id % p.L);
  const int bh = (int)(chunk_id / p.L), b = bh / p.H, h = bh - b * p.H;
  const char* qb = p.q + (b * p.q_sb + h * p.q_sh) * 2;
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  float aq[8], ak[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) aq[i] = ak[i] = 0.f;
  // Round 6: U row steps per memory round trip -- every row (and mask byte) of a macro step is requested up front from a
  // clamped address, then the steps are consumed in the old order (the same additions in the same order: bit-identical).
  // The one-step loop waited a full round trip per 8 (D = 64) rows: 17 dependent trips for a 528-slot chunk of cfg5.
  constexpr int U = LmU<WPC>::value, STRIDE = WPC * RPW;
  for (int jb = sub * RPW; jb < p.J; jb += U * STRIDE) {
    u32x4 rq[U], rk[U];
    int tk[U];
    uint8_t mk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= p.J) break;                       // (uniform)
      const int j = jb + u * STRIDE + rg;
      tk[u] = j < p.J ? part_token(p.G, cidx, j, p.r, p.e) : -1;
      const int tc = max(tk[u], 0);
      rq[u] = ldg16(qb + (tc * p.q_sn + c * 8) * 2);
      rk[u] = ldg16(kb + (tc * p.k_sn + c * 8) * 2);
      mk[u] = mrow ? mrow[tc] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= p.J) break;
      if (tk[u] >= 0 && !mk[u]) {
        float f[8];
        unpack8<E>(rq[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) aq[i] += f[i];
        unpack8<E>(rk[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) ak[i] += f[i];
      }
    }
  }
  const float inv = 1.f / (float)p.J;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    aq[i] = rows_sum<CPR>(aq[i]) * inv;
    ak[i] = rows_sum<CPR>(ak[i]) * inv;
  }
  if (WPC > 1) {
    __shared__ float red[WPC][2][D];
    if (rg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { red[wave][0][c * 8 + i] = aq[i]; red[wave][1][c * 8 + i] = ak[i]; }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      aq[i] = ak[i] = 0.f;
#pragma unroll
      for (int w = 0; w < WPC; ++w) { aq[i] += red[w][0][c * 8 + i]; ak[i] += red[w][1][c * 8 + i]; }
    }
  }
  if (rg == 0) {
    float* dq_ = p.qmean + (size_t)chunk_id * D + c * 8;
    float* dk_ = p.kmean + (size_t)chunk_id * D + c * 8;
    *reinterpret_cast<float4*>(dq_) = make_float4(aq[0], aq[1], aq[2], aq[3]);
    *reinterpret_cast<float4*>(dq_ + 4) = make_float4(aq[4], aq[5], aq[6], aq[7]);
    *reinterpret_cast<float4*>(dk_) = make_float4(ak[0], ak[1], ak[2], ak[3]);
    *reinterpret_cast<float4*>(dk_ + 4) = make_float4(ak[4], ak[5], ak[6], ak[7]);
  }
}

// dq[tok] += dqmean[c]/J, dk[tok] += dkmean[c]/J for every unmasked in-range slot of chunk c.
// With e == 0 every token belongs to exactly one chunk, so the read-modify-write is race free;
// e > 0 (overlapping chunks) is serialised by launching one chunk "colour" at a time (host).
template <typename E, int D, int WPC>
__global__ __launch_bounds__(256) void chunk_mean_bwd_kernel(const LmP p, int colour, int nc) {
  constexpr int CPR = D / 8, RPW = 64 / CPR;
  const int lane = threadIdx.x & 63, c = lane % CPR, rg = lane / CPR, wave = threadIdx.x >> 6;
  const int sub = WPC == 1 ? 0 : wave;
  const long chunk_id = WPC == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x;
  if (chunk_id >= (long)p.B * p.H * p.L) return;
  const int cidx = (int)(chunk_id % p.L);
  if (nc > 1) {
    int col;
    if (p.G.attn2d) { const int per_row = p.G.gw / p.r; col = ((cidx / per_row) % nc) * nc + ((cidx % per_row) % nc); }
    else col = cidx % nc;
    if (col != colour) return;
  }
  const int bh = (int)(chunk_id / p.L), b = bh / p.H, h = bh - b * p.H;
  char* dqb = p.dq + (b * p.dq_sb + h * p.dq_sh) * 2;
  char* dkb = p.dk + (b * p.dk_sb + h * p.dk_sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  const float inv = 1.f / (float)p.J;
  float gq[8], gk[8];
  {
    const float* s1 = p.dqmean + (size_t)chunk_id * D + c * 8;
    const float* s2 = p.dkmean + (size_t)chunk_id * D + c * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { gq[i] = s1[i] * inv; gk[i] = s2[i] * inv; }
  }
  for (int j = sub * RPW + rg; j < p.J; j += WPC * RPW) {
    const int tok = part_token(p.G, cidx, j, p.r, p.e);
    if (tok >= 0 && !(mrow && mrow[tok])) {
      float f[8];
      char* a = dqb + (tok * p.dq_sn + c * 8) * 2;
      unpack8<E>(ldg16(a), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += gq[i];
      stg16(a, pack8<E>(f));
      a = dkb + (tok * p.dk_sn + c * 8) * 2;
      unpack8<E>(ldg16(a), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += gk[i];
      stg16(a, pack8<E>(f));
    }
  }
}

// Overlapping chunks (e > 0): token-centric form of the same update -- a thread owns one 16-B piece of
// one token, sums dmean / J over the (at most nc per axis) chunks that contain the token and does a
// single read-modify-write.  One launch instead of one per colour class, one rounding per token.
template <typename E, int D>
__global__ __launch_bounds__(256) void chunk_mean_bwd_gather_kernel(const LmP p) {
  constexpr int CPR = D / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.B * p.H * p.G.N * CPR;
  if (idx >= total) return;
  const int c = (int)(idx % CPR);
  const long row = idx / CPR;
  const int tok = (int)(row % p.G.N);
  const int bh = (int)(row / p.G.N), b = bh / p.H, h = bh - b * p.H;
  if (p.mask && p.mask[(size_t)b * p.G.N + tok]) return;       // masked slots count as zeros in the mean
  const float inv = 1.f / (float)p.J;
  float gq[8], gk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) gq[i] = gk[i] = 0.f;
  // chunks cy x cx whose extended range [c r - e, c r + r + e) holds the token coordinate
  const int ty = p.G.attn2d ? tok / p.G.gw : 0, tx = p.G.attn2d ? tok - ty * p.G.gw : tok;
  const int per_row = p.G.attn2d ? p.G.gw / p.r : p.L, rows = p.G.attn2d ? p.G.gh / p.r : 1;
  auto lo = [&](int t) { const int v = t - p.r - p.e; return (v >= 0 ? v / p.r : -1) + 1; };
  const int cx0 = lo(tx), cx1 = min((tx + p.e) / p.r, per_row - 1);
  const int cy0 = p.G.attn2d ? lo(ty) : 0, cy1 = p.G.attn2d ? min((ty + p.e) / p.r, rows - 1) : 0;
  for (int cy = cy0; cy <= cy1; ++cy)
    for (int cx = cx0; cx <= cx1; ++cx) {
      const size_t chunk_id = (size_t)bh * p.L + cy * per_row + cx;
      const float* s1 = p.dqmean + chunk_id * D + c * 8;
      const float* s2 = p.dkmean + chunk_id * D + c * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) { gq[i] += s1[i] * inv; gk[i] += s2[i] * inv; }
    }
  float f[8];
  char* a = p.dq + (b * p.dq_sb + h * p.dq_sh + tok * p.dq_sn + c * 8) * 2;
  unpack8<E>(ldg16(a), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] += gq[i];
  stg16(a, pack8<E>(f));
  a = p.dk + (b * p.dk_sb + h * p.dk_sh + tok * p.dk_sn + c * 8) * 2;
  unpack8<E>(ldg16(a), f);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] += gk[i];
  stg16(a, pack8<E>(f));
}

// ------------------------------------------------------------------------------------------
// beta forward: online softmax over the chunk's rows per row-group, merged across row-groups.
template <typename E, int D, int WPC>
__global__ __launch_bounds__(256) void beta_fwd_kernel(const LmP p) {
  constexpr int CPR = D / 8, RPW = 64 / CPR;
  const int lane = threadIdx.x & 63, c = lane % CPR, rg = lane / CPR, wave = threadIdx.x >> 6;
  const int sub = WPC == 1 ? 0 : wave;
  const long chunk_id = WPC == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x;
  if (chunk_id >= (long)p.B * p.H * p.L) return;
  const int cidx = (int)(chunk_id % p.L);
  const int bh = (int)(chunk_id / p.L), b = bh / p.H, h = bh - b * p.H;
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  const char* vb = p.v + (b * p.v_sb + h * p.v_sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  float om[8];
  {
    const float* s = p.omega + (size_t)chunk_id * D + c * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) om[i] = s[i];
  }
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const int jmax = (p.J + WPC * RPW - 1) / (WPC * RPW) * (WPC * RPW);
  constexpr int U = LmU<WPC>::value, STRIDE = WPC * RPW;      // U row steps per memory round trip (see chunk_mean_fwd_kernel)
  for (int jb = sub * RPW; jb < jmax; jb += U * STRIDE) {
    u32x4 rk[U], rv[U];
    int tk[U];
    uint8_t mk[U];
    bool ex[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= jmax) break;                      // (uniform: jmax is a multiple of STRIDE)
      const int j = jb + u * STRIDE + rg;
      ex[u] = j < p.J;
      tk[u] = ex[u] ? part_token(p.G, cidx, j, p.r, p.e) : -1;
      const int tc = max(tk[u], 0);
      rk[u] = ldg16(kb + (tc * p.k_sn + c * 8) * 2);
      rv[u] = ldg16(vb + (tc * p.v_sn + c * 8) * 2);
      mk[u] = mrow ? mrow[tc] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= jmax) break;
      const bool exists = ex[u];
      const bool live = tk[u] >= 0 && !mk[u];
      float kf[8], vf[8];
      unpack8<E>(rk[u], kf);
      unpack8<E>(rv[u], vf);
#pragma unroll
      for (int i = 0; i < 8; ++i) { kf[i] = live ? kf[i] : 0.f; vf[i] = live ? vf[i] : 0.f; }
      float dot = 0.f, nrm = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { dot += om[i] * kf[i]; nrm += kf[i] * kf[i]; }
      dot = chan_sum<CPR>(dot);
      nrm = chan_sum<CPR>(nrm);
      float x = p.scale * (dot - 0.5f * nrm);
      x = live ? x : (exists ? MASK_FILL : -INFINITY);
      const float mn = fmaxf(m, x);
      const float ms = mn == -INFINITY ? 0.f : mn;
      const float a = __expf(m - ms), pj = __expf(x - ms);
      l = l * a + pj;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + pj * vf[i];
      m = mn;
    }
  }
  // merge the RPW row-groups (lanes with equal c)
#pragma unroll
  for (int o = CPR; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o), l2 = __shfl_xor(l, o);
    const float mn = fmaxf(m, m2);
    const float ms = mn == -INFINITY ? 0.f : mn;
    const float a1 = __expf(m - ms), a2 = __expf(m2 - ms);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a1 + __shfl_xor(acc[i], o) * a2;
    m = mn;
  }
  if (WPC > 1) {                                           // merge the waves' (m, l, acc) in wave order
    __shared__ float red_ml[WPC][2];
    __shared__ float red_acc[WPC][D];
    if (rg == 0) {
      if (c == 0) { red_ml[wave][0] = m; red_ml[wave][1] = l; }
#pragma unroll
      for (int i = 0; i < 8; ++i) red_acc[wave][c * 8 + i] = acc[i];
    }
    __syncthreads();
    if (wave != 0) return;
    m = -INFINITY;
#pragma unroll
    for (int w = 0; w < WPC; ++w) m = fmaxf(m, red_ml[w][0]);
    const float ms = m == -INFINITY ? 0.f : m;
    l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int w = 0; w < WPC; ++w) {
      const float aw = __expf(red_ml[w][0] - ms);
      l += red_ml[w][1] * aw;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += red_acc[w][c * 8 + i] * aw;
    }
  }
  if (rg == 0) {
    const float inv = 1.f / l;
    float* dst = p.beta_out + (size_t)chunk_id * D + c * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
  }
}

// beta backward.  With p_j = softmax_j(x_j):  dv_j += p_j dbeta;  dx_j = p_j (v_j.dbeta - beta.dbeta);
// dk_j += dx_j s (w - k_j);  dw += sum_j dx_j s k_j   (masked / outside slots carry no gradient).
template <typename E, int D, int WPC>
__global__ __launch_bounds__(256) void beta_bwd_kernel(const LmP p, int colour, int nc) {
  constexpr int CPR = D / 8, RPW = 64 / CPR;
  const int lane = threadIdx.x & 63, c = lane % CPR, rg = lane / CPR, wave = threadIdx.x >> 6;
  const int sub = WPC == 1 ? 0 : wave;
  const long chunk_id = WPC == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x;
  if (chunk_id >= (long)p.B * p.H * p.L) return;
  const int cidx = (int)(chunk_id % p.L);
  if (nc > 1) {
    int col;
    if (p.G.attn2d) { const int per_row = p.G.gw / p.r; col = ((cidx / per_row) % nc) * nc + ((cidx % per_row) % nc); }
    else col = cidx % nc;
    if (col != colour) return;
  }
  const int bh = (int)(chunk_id / p.L), b = bh / p.H, h = bh - b * p.H;
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  const char* vb = p.v + (b * p.v_sb + h * p.v_sh) * 2;
  char* dkb = p.dk + (b * p.dk_sb + h * p.dk_sh) * 2;
  char* dvb = p.dv + (b * p.dv_sb + h * p.dv_sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  float om[8], db[8], bt[8];
  {
    const size_t off = (size_t)chunk_id * D + c * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { om[i] = p.omega[off + i]; db[i] = p.dbeta[off + i]; bt[i] = p.beta[off + i]; }
    for (int s_ = 1; s_ < p.dbeta_S; ++s_)
#pragma unroll
      for (int i = 0; i < 8; ++i) db[i] += p.dbeta[(size_t)s_ * p.dbeta_stride + off + i];
  }
  float bd = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) bd += bt[i] * db[i];
  bd = chan_sum<CPR>(bd);                                   // beta . dbeta
  const int jmax = (p.J + WPC * RPW - 1) / (WPC * RPW) * (WPC * RPW);
  // pass 1: log-sum-exp of the chunk's logits
  float m = -INFINITY, l = 0.f;
  constexpr int U = LmU<WPC>::value, STRIDE = WPC * RPW;      // U row steps per memory round trip (see chunk_mean_fwd_kernel)
  for (int jb = sub * RPW; jb < jmax; jb += U * STRIDE) {
    u32x4 rk[U];
    int tk[U];
    uint8_t mk[U];
    bool ex[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= jmax) break;
      const int j = jb + u * STRIDE + rg;
      ex[u] = j < p.J;
      tk[u] = ex[u] ? part_token(p.G, cidx, j, p.r, p.e) : -1;
      const int tc = max(tk[u], 0);
      rk[u] = ldg16(kb + (tc * p.k_sn + c * 8) * 2);
      mk[u] = mrow ? mrow[tc] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (jb + u * STRIDE >= jmax) break;
      const bool exists = ex[u];
      const bool live = tk[u] >= 0 && !mk[u];
      float kf[8];
      unpack8<E>(rk[u], kf);
#pragma unroll
      for (int i = 0; i < 8; ++i) kf[i] = live ? kf[i] : 0.f;
      float dot = 0.f, nrm = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { dot += om[i] * kf[i]; nrm += kf[i] * kf[i]; }
      dot = chan_sum<CPR>(dot);
      nrm = chan_sum<CPR>(nrm);
      float x = p.scale * (dot - 0.5f * nrm);
      x = live ? x : (exists ? MASK_FILL : -INFINITY);
      const float mn = fmaxf(m, x);
      const float ms = mn == -INFINITY ? 0.f : mn;
      l = l * __expf(m - ms) + __expf(x - ms);
      m = mn;
    }
  }
#pragma unroll
  for (int o = CPR; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o), l2 = __shfl_xor(l, o);
    const float mn = fmaxf(m, m2);
    const float ms = mn == -INFINITY ? 0.f : mn;
    l = l * __expf(m - ms) + l2 * __expf(m2 - ms);
    m = mn;
  }
  __shared__ float red_ml[WPC][2];
  __shared__ float red_dom[WPC][D];
  if (WPC > 1) {
    if (lane == 0) { red_ml[wave][0] = m; red_ml[wave][1] = l; }
    __syncthreads();
    m = -INFINITY;
#pragma unroll
    for (int w = 0; w < WPC; ++w) m = fmaxf(m, red_ml[w][0]);
    const float ms = m == -INFINITY ? 0.f : m;
    l = 0.f;
#pragma unroll
    for (int w = 0; w < WPC; ++w) l += red_ml[w][1] * __expf(red_ml[w][0] - ms);
  }
  const float lse = m + __logf(l);
  // pass 2: gradients
  float dom[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dom[i] = 0.f;
  // (round 6) the rows AND the gradient rows they are added to, U row steps per round trip: the one-step loop made three
  // dependent trips per step (k / v, then dv, then dk) -- 2 x 90 us for the two colour classes of cfg5 at 2.2 TB/s.  The rows
  // of a macro step are distinct tokens (clamped slots load token 0 and store nothing), so loading every gradient row before
  // the first store of the macro step reads what the one-step loop read.
  constexpr int U2 = WPC == 4 ? 6 : 4;
  for (int jb = sub * RPW; jb < jmax; jb += U2 * STRIDE) {
    u32x4 rk[U2], rv[U2], rdk[U2], rdv[U2];
    int tk[U2];
    uint8_t mk[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      if (jb + u * STRIDE >= jmax) break;
      const int j = jb + u * STRIDE + rg;
      tk[u] = j < p.J ? part_token(p.G, cidx, j, p.r, p.e) : -1;
      const int tc = max(tk[u], 0);
      rk[u] = ldg16(kb + (tc * p.k_sn + c * 8) * 2);
      rv[u] = ldg16(vb + (tc * p.v_sn + c * 8) * 2);
      rdv[u] = ldg16(dvb + (tc * p.dv_sn + c * 8) * 2);
      rdk[u] = ldg16(dkb + (tc * p.dk_sn + c * 8) * 2);
      mk[u] = mrow ? mrow[tc] : (uint8_t)0;
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      if (jb + u * STRIDE >= jmax) break;
      const int tok = tk[u];
      const bool live = tok >= 0 && !mk[u];
      float kf[8], vf[8];
      unpack8<E>(rk[u], kf);
      unpack8<E>(rv[u], vf);
#pragma unroll
      for (int i = 0; i < 8; ++i) { kf[i] = live ? kf[i] : 0.f; vf[i] = live ? vf[i] : 0.f; }
      float dot = 0.f, nrm = 0.f, vd = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { dot += om[i] * kf[i]; nrm += kf[i] * kf[i]; vd += vf[i] * db[i]; }
      dot = chan_sum<CPR>(dot);
      nrm = chan_sum<CPR>(nrm);
      vd = chan_sum<CPR>(vd);
      if (live) {
        const float x = p.scale * (dot - 0.5f * nrm);
        const float pj = __expf(x - lse);
        const float dx = pj * (vd - bd) * p.scale;
        float f[8];
        unpack8<E>(rdv[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += pj * db[i];
        stg16(dvb + (tok * p.dv_sn + c * 8) * 2, pack8<E>(f));
        unpack8<E>(rdk[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { f[i] += dx * (om[i] - kf[i]); dom[i] += dx * kf[i]; }
        stg16(dkb + (tok * p.dk_sn + c * 8) * 2, pack8<E>(f));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dom[i] = rows_sum<CPR>(dom[i]);
  if (WPC > 1) {
    if (rg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red_dom[wave][c * 8 + i] = dom[i];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dom[i] = 0.f;
#pragma unroll
      for (int w = 0; w < WPC; ++w) dom[i] += red_dom[w][c * 8 + i];
    }
  }
  if (rg == 0) {
    float* dst = p.domega + (size_t)chunk_id * D + c * 8;
    *reinterpret_cast<float4*>(dst) = make_float4(dom[0], dom[1], dom[2], dom[3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(dom[4], dom[5], dom[6], dom[7]);
  }
}

// ==========================================================================================
// Register-resident variants (round 3) for short chunks (2-D pooling: one wave per chunk, J <= 8 row steps).
// The loops above fetch a row step, wait, use it, fetch the next: 8 (forward) to 24 (beta backward) dependent memory
// round trips per wave, and hipcc turns the `live ? load : 0` selects back into predicated loads.  Here EVERY row of the
// chunk is requested up front from clamped addresses (k, v -- and dk, dv for the read-modify-writes -- stay packed in
// registers), the wave waits once, and the backward reads k a single time instead of twice.
template <int N> EA_DEV void pin_regs(u32x4 (&x)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
}
template <int D, int NI> struct ChunkRows {
  static constexpr int CPR = D / 8, RPW = 64 / CPR;
  int tok[NI];
  uint32_t flags[NI];          // bit 0: slot exists (j < J), bit 1: inside the sequence, bits 8..: mask byte
  EA_DEV void init(const LmP& p, int cidx, int rg, const uint8_t* mrow) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = rg + i * RPW;
      const bool ex = j < p.J;
      const int t = ex ? part_token(p.G, cidx, j, p.r, p.e) : -1;
      tok[i] = t >= 0 ? t : 0;
      flags[i] = (ex ? 1u : 0u) | (t >= 0 ? 2u : 0u);
      if (mrow) flags[i] |= (uint32_t)mrow[tok[i]] << 8;
    }
  }
  EA_DEV void pin() {
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(flags[i]));
  }
  EA_DEV bool exists(int i) const { return flags[i] & 1u; }
  EA_DEV bool live(int i) const { return (flags[i] & 2u) && !(flags[i] >> 8); }
};
#define EA_CHUNK_PROLOGUE                                                                       \
  constexpr int CPR = D / 8, RPW = 64 / CPR;                                                    \
  const int lane = threadIdx.x & 63, c = lane % CPR, rg = lane / CPR, wave = threadIdx.x >> 6;  \
  const long chunk_id = (long)blockIdx.x * 4 + wave;                                            \
  if (chunk_id >= (long)p.B * p.H * p.L) return;                                                \
  const int cidx = (int)(chunk_id % p.L);                                                       \
  const int bh = (int)(chunk_id / p.L), b = bh / p.H, h = bh - b * p.H;                         \
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;                          \
  (void)RPW; (void)rg; (void)h

EA_DEV void ld8(const float* s, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
EA_DEV void st8(float* d, const float* f) {
  *reinterpret_cast<float4*>(d) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

template <typename E, int D, int NI>
__global__ __launch_bounds__(256) void chunk_mean_fwd_r_kernel(const LmP p) {
  EA_CHUNK_PROLOGUE;
  const char* qb = p.q + (b * p.q_sb + h * p.q_sh) * 2;
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  ChunkRows<D, NI> R;
  R.init(p, cidx, rg, mrow);
  u32x4 qr[NI], kr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    qr[i] = ldg16(qb + (R.tok[i] * p.q_sn + c * 8) * 2);
    kr[i] = ldg16(kb + (R.tok[i] * p.k_sn + c * 8) * 2);
  }
  pin_regs(qr); pin_regs(kr); R.pin();
  float aq[8], ak[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) aq[i] = ak[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float w = R.live(i) ? 1.f : 0.f;
    float f[8];
    unpack8<E>(qr[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) aq[e] = fmaf(w, f[e], aq[e]);
    unpack8<E>(kr[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) ak[e] = fmaf(w, f[e], ak[e]);
  }
  const float inv = 1.f / (float)p.J;
#pragma unroll
  for (int i = 0; i < 8; ++i) { aq[i] = rows_sum<CPR>(aq[i]) * inv; ak[i] = rows_sum<CPR>(ak[i]) * inv; }
  if (rg == 0) {
    st8(p.qmean + (size_t)chunk_id * D + c * 8, aq);
    st8(p.kmean + (size_t)chunk_id * D + c * 8, ak);
  }
}

template <typename E, int D, int NI>
__global__ __launch_bounds__(256) void chunk_mean_bwd_r_kernel(const LmP p) {
  EA_CHUNK_PROLOGUE;
  char* dqb = p.dq + (b * p.dq_sb + h * p.dq_sh) * 2;
  char* dkb = p.dk + (b * p.dk_sb + h * p.dk_sh) * 2;
  ChunkRows<D, NI> R;
  R.init(p, cidx, rg, mrow);
  u32x4 qr[NI], kr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    qr[i] = ldg16(dqb + (R.tok[i] * p.dq_sn + c * 8) * 2);
    kr[i] = ldg16(dkb + (R.tok[i] * p.dk_sn + c * 8) * 2);
  }
  float gq[8], gk[8];
  ld8(p.dqmean + (size_t)chunk_id * D + c * 8, gq);
  ld8(p.dkmean + (size_t)chunk_id * D + c * 8, gk);
  pin_regs(qr); pin_regs(kr); R.pin();
  const float inv = 1.f / (float)p.J;
#pragma unroll
  for (int i = 0; i < 8; ++i) { gq[i] *= inv; gk[i] *= inv; }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (!R.live(i)) continue;
    float f[8];
    unpack8<E>(qr[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += gq[e];
    stg16(dqb + (R.tok[i] * p.dq_sn + c * 8) * 2, pack8<E>(f));
    unpack8<E>(kr[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += gk[e];
    stg16(dkb + (R.tok[i] * p.dk_sn + c * 8) * 2, pack8<E>(f));
  }
}

// logits of the chunk's rows from the packed k rows: x[i] (live), MASK_FILL (padded / outside), -inf (no such slot)
template <typename E, int D, int NI>
EA_DEV void chunk_logits(const ChunkRows<D, NI>& R, const u32x4* kr, const float* om, float scale, float* x) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float kf[8];
    unpack8<E>(kr[i], kf);
    float dot = 0.f, nrm = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dot += om[e] * kf[e]; nrm += kf[e] * kf[e]; }
    dot = chan_sum<CPR>(dot);
    nrm = chan_sum<CPR>(nrm);
    const float v = scale * (dot - 0.5f * nrm);
    x[i] = R.live(i) ? v : (R.exists(i) ? MASK_FILL : -INFINITY);
  }
}
// max and sum of exp over ALL rows of the chunk (the row steps of this lane, then the row groups of the wave)
template <int CPR, int NI> EA_DEV void chunk_lse(const float* x, float& m, float& l) {
  m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) m = fmaxf(m, x[i]);
  m = stride_max<CPR>(m);
  l = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) l += __expf(x[i] - m);
  l = rows_sum<CPR>(l);
}

template <typename E, int D, int NI>
__global__ __launch_bounds__(256) void beta_fwd_r_kernel(const LmP p) {
  EA_CHUNK_PROLOGUE;
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  const char* vb = p.v + (b * p.v_sb + h * p.v_sh) * 2;
  ChunkRows<D, NI> R;
  R.init(p, cidx, rg, mrow);
  u32x4 kr[NI], vr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    kr[i] = ldg16(kb + (R.tok[i] * p.k_sn + c * 8) * 2);
    vr[i] = ldg16(vb + (R.tok[i] * p.v_sn + c * 8) * 2);
  }
  float om[8];
  ld8(p.omega + (size_t)chunk_id * D + c * 8, om);
  pin_regs(kr); pin_regs(vr); R.pin();
  float x[NI], m, l;
  chunk_logits<E, D, NI>(R, kr, om, p.scale, x);
  chunk_lse<CPR, NI>(x, m, l);                             // (slot 0 always exists: m is finite)
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float pj = R.live(i) ? __expf(x[i] - m) : 0.f;   // padded slots weigh in l only (their v counts as zero)
    float vf[8];
    unpack8<E>(vr[i], vf);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, vf[e], acc[e]);
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = rows_sum<CPR>(acc[e]) * inv;
  if (rg == 0) st8(p.beta_out + (size_t)chunk_id * D + c * 8, acc);
}

template <typename E, int D, int NI>
__global__ __launch_bounds__(256) void beta_bwd_r_kernel(const LmP p, int colour, int nc) {
  EA_CHUNK_PROLOGUE;
  if (nc > 1) {
    int col;
    if (p.G.attn2d) { const int per_row = p.G.gw / p.r; col = ((cidx / per_row) % nc) * nc + ((cidx % per_row) % nc); }
    else col = cidx % nc;
    if (col != colour) return;
  }
  const char* kb = p.k + (b * p.k_sb + h * p.k_sh) * 2;
  const char* vb = p.v + (b * p.v_sb + h * p.v_sh) * 2;
  char* dkb = p.dk + (b * p.dk_sb + h * p.dk_sh) * 2;
  char* dvb = p.dv + (b * p.dv_sb + h * p.dv_sh) * 2;
  ChunkRows<D, NI> R;
  R.init(p, cidx, rg, mrow);
  u32x4 kr[NI], vr[NI], dkr[NI], dvr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    kr[i] = ldg16(kb + (R.tok[i] * p.k_sn + c * 8) * 2);
    vr[i] = ldg16(vb + (R.tok[i] * p.v_sn + c * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    dkr[i] = ldg16(dkb + (R.tok[i] * p.dk_sn + c * 8) * 2);
    dvr[i] = ldg16(dvb + (R.tok[i] * p.dv_sn + c * 8) * 2);
  }
  float om[8], db[8], bt[8];
  {
    const size_t off = (size_t)chunk_id * D + c * 8;
    ld8(p.omega + off, om); ld8(p.dbeta + off, db); ld8(p.beta + off, bt);
    for (int s_ = 1; s_ < p.dbeta_S; ++s_) {               // (uniform) slice partials of the window backward, added in slice order
      float ps[8];
      ld8(p.dbeta + (size_t)s_ * p.dbeta_stride + off, ps);
#pragma unroll
      for (int i = 0; i < 8; ++i) db[i] += ps[i];
    }
  }
  pin_regs(kr); pin_regs(vr); pin_regs(dkr); pin_regs(dvr); R.pin();
  float bd = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) bd += bt[e] * db[e];
  bd = chan_sum<CPR>(bd);                                   // beta . dbeta
  float x[NI], m, l;
  chunk_logits<E, D, NI>(R, kr, om, p.scale, x);
  chunk_lse<CPR, NI>(x, m, l);
  const float lse = m + __logf(l);
  float dom[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dom[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float kf[8], vf[8];
    unpack8<E>(kr[i], kf);
    unpack8<E>(vr[i], vf);
    float vd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) vd += vf[e] * db[e];
    vd = chan_sum<CPR>(vd);
    if (R.live(i)) {
      const float pj = __expf(x[i] - lse);
      const float dx = pj * (vd - bd) * p.scale;
      float f[8];
      unpack8<E>(dvr[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += pj * db[e];
      stg16(dvb + (R.tok[i] * p.dv_sn + c * 8) * 2, pack8<E>(f));
      unpack8<E>(dkr[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { f[e] += dx * (om[e] - kf[e]); dom[e] += dx * kf[e]; }
      stg16(dkb + (R.tok[i] * p.dk_sn + c * 8) * 2, pack8<E>(f));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) dom[e] = rows_sum<CPR>(dom[e]);
  if (rg == 0) st8(p.domega + (size_t)chunk_id * D + c * 8, dom);
}
#undef EA_CHUNK_PROLOGUE

// dev switch: EA_LM_REG=0 keeps the looping kernels for short chunks
static bool lm_reg_on() {
  static const bool v = [] { const char* e = getenv("EA_LM_REG"); return !e || atoi(e) != 0; }();
  return v;
}
template <typename E, int D, int NI>
static void launch_lm_r(int which, const LmP& p, dim3 grid, int ncolour, int nc, hipStream_t st) {
  const dim3 block(256);
  switch (which) {
    case 0: hipLaunchKernelGGL((chunk_mean_fwd_r_kernel<E, D, NI>), grid, block, 0, st, p); break;
    case 1: hipLaunchKernelGGL((chunk_mean_bwd_r_kernel<E, D, NI>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((beta_fwd_r_kernel<E, D, NI>), grid, block, 0, st, p); break;
    default:
      for (int col = 0; col < ncolour; ++col) hipLaunchKernelGGL((beta_bwd_r_kernel<E, D, NI>), grid, block, 0, st, p, col, nc);
  }
}

// ------------------------------------------------------------------------------------------
template <typename E, int D>
static int launch_lm(int which, const LmP& p, hipStream_t st) {
  const long chunks = (long)p.B * p.H * p.L;
  // long chunks, or too few chunks to give every CU a few waves: four waves per chunk
  const bool coop = p.J >= 128 || (p.J >= 64 && chunks < 4096);
  const dim3 grid((unsigned)(coop ? chunks : (chunks + 3) / 4)), block(256);
  // overlapping chunks (e > 0): chunks closer than nc = 1 + ceil(2e/r) per dimension share
  // tokens, so nc (1-D) / nc*nc (2-D) colour classes are launched back to back and the
  // read-modify-writes of one launch never touch the same token.
  const int nc = p.e > 0 ? 1 + (2 * p.e + p.r - 1) / p.r : 1;
  const int ncolour = p.G.attn2d ? nc * nc : nc;
  if constexpr (D <= 64) {
    // short chunks, one wave each: the register-resident variants (the chunk-mean backward of overlapping chunks keeps its
    // token-centric gather kernel)
    const int steps = (p.J + 64 / (D / 8) - 1) / (64 / (D / 8));
    if (!coop && steps <= 8 && lm_reg_on() && !(which == 1 && ncolour > 1)) {
      if (steps <= 1) launch_lm_r<E, D, 1>(which, p, grid, ncolour, nc, st);
      else if (steps <= 2) launch_lm_r<E, D, 2>(which, p, grid, ncolour, nc, st);
      else if (steps <= 4) launch_lm_r<E, D, 4>(which, p, grid, ncolour, nc, st);
      else launch_lm_r<E, D, 8>(which, p, grid, ncolour, nc, st);
      return (int)hipGetLastError();
    }
  }
  switch (which) {
    case 0:
      if (coop) hipLaunchKernelGGL((chunk_mean_fwd_kernel<E, D, 4>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((chunk_mean_fwd_kernel<E, D, 1>), grid, block, 0, st, p);
      break;
    case 1:
      if (ncolour > 1) {
        const long pieces = (long)p.B * p.H * p.G.N * (D / 8);
        hipLaunchKernelGGL((chunk_mean_bwd_gather_kernel<E, D>), dim3((unsigned)((pieces + 255) / 256)), block, 0, st, p);
      } else if (coop) hipLaunchKernelGGL((chunk_mean_bwd_kernel<E, D, 4>), grid, block, 0, st, p, 0, 1);
      else hipLaunchKernelGGL((chunk_mean_bwd_kernel<E, D, 1>), grid, block, 0, st, p, 0, 1);
      break;
    case 2:
      if (coop) hipLaunchKernelGGL((beta_fwd_kernel<E, D, 4>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((beta_fwd_kernel<E, D, 1>), grid, block, 0, st, p);
      break;
    case 3:
      for (int col = 0; col < ncolour; ++col) {
        if (coop) hipLaunchKernelGGL((beta_bwd_kernel<E, D, 4>), grid, block, 0, st, p, col, nc);
        else hipLaunchKernelGGL((beta_bwd_kernel<E, D, 1>), grid, block, 0, st, p, col, nc);
      }
      break;
  }
  return (int)hipGetLastError();
}

int landmark_dispatch(int which, const LmP& p, int dtype, int D, hipStream_t st) {
  if (dtype == EA_BF16) {
    if (D == 64) return launch_lm<BF16, 64>(which, p, st);
    if (D == 32) return launch_lm<BF16, 32>(which, p, st);
    if (D == 128) return launch_lm<BF16, 128>(which, p, st);
  } else if (dtype == EA_F16) {
    if (D == 64) return launch_lm<F16, 64>(which, p, st);
    if (D == 32) return launch_lm<F16, 32>(which, p, st);
    if (D == 128) return launch_lm<F16, 128>(which, p, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
