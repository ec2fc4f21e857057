// ea_f32_attn.hip -- fp32-FAITHFUL attention cores (round 5).
//
// Outside torch.autocast the reference computes its attention in fp32 (abstract_attention.py:120-133, local_attention.py:
// 134-182, eva.py:138-233, causal_eva.py:666-783); the 16-bit kernels round q, k, v and the probabilities to bf16 / fp16
// operands.  These kernels keep every operand in fp32 -- products on v_mfma_f32_16x16x4_f32 out of fp32 LDS images
// (ea_f32_mm.h), fp32 softmax -- so a module called in fp32 matches the reference at fp32 tolerances (2e-4 / 1e-4).
//
// ONE generic pair covers every softmax-shaped core of the library, written the way the reference states them -- gathered
// windows under a joint softmax:
//   for every group g (window | chunk | the whole sequence) and query slot i:
//     logits_ij = s q_i.k_j [- s |k_j|^2 / 2] [+ bias[h, i, j]]      j over the group's Wk key slots  (idx_k[g][j], -1 = absent)
//     logits_ic = s q_i.ek_c                                           c over L extra keys shared by all groups (EVA: rf_k_bar)
//     masked entries (padded / absent key, padded query, causal rules) take the finite -5e4 (eva.py:139) or -inf (softmax
//     baseline, lara.py:205-208); out_i = sum_j P_ij v_j + sum_c P_ic ev_c  (ev: EVA's beta), lse_i returned
//   softmax baseline: one group, idx = arange(N); local: windows; EVA: windows + landmark columns; EVA's beta / LARA's
//   kv_stats: queries = omega rows, key-norm term; randomized attention: two passes.
// Backward: flash-style recompute per (query block, key chunk); dq stored, dk / dv / d ek / d ev / d bias accumulated with
// fp32 atomics (windows overlap) -- a fidelity path, not the fast one: the 16-bit kernels stay the training path under AMP.
#include "ea_common.h"
#include "ea_f32_mm.h"
#include "ea_f32_attn.h"

namespace ea {

namespace {

constexpr int QB = 64;                 // query rows per workgroup
constexpr int NT = 256;                // threads: four waves, wave w owns rows 16 w .. 16 w + 15

EA_DEV float group_max16(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x128>(v));
  return v;
}

// rows of a [B,H,N,D] fp32 view addressed through a token table -> dst[rows][D + 1] (absent rows: zeros)
template <int D>
EA_DEV void load_rows(float* dst, const F32T& t, int b, int h, const int* tok, int rows, int tid) {
  constexpr int LD = D + 1, C4 = D / 4;
  for (int idx = tid; idx < rows * C4; idx += NT) {
    const int r = idx / C4, c = (idx - r * C4) * 4;
    const int tk = tok[r];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (tk >= 0) v = *reinterpret_cast<const f32x4*>(t.p + (size_t)b * t.sb + (size_t)h * t.sh + (size_t)tk * t.sn + c);
    float* d = dst + r * LD + c;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

// The key chunk [kc0, kc0 + KC) of group g: local slots then the L extra keys, then nothing.
//   ktok[j] : token index (local) | L-row index (extra) | -1 ; kkind[j]: 0 local, 1 extra, 2 beyond the key list
//   kflag[j]: 0 visible, 1 masked with the finite fill, 2 excluded (-inf)
template <int D, int KC>
EA_DEV void stage_keys(const GaP& p, int b, int h, int g, int kc0, float* Ks, float* Vs, int* ktok, int* kkind, int* kflag,
                       float* kadd, int tid) {
  constexpr int LD = D + 1, C4 = D / 4;
  if (tid < KC) {
    const int j = kc0 + tid;
    int tk = -1, kind = 2, fl = 2;
    if (j < p.Wk) {
      kind = 0;
      tk = p.idx_k[(size_t)g * p.Wk + j];
      fl = tk < 0 ? 1 : 0;
      if (tk >= 0 && p.kmask && p.kmask[(size_t)b * p.Nk + tk]) fl = p.neg_inf ? 2 : 1;
    } else if (j < p.Wk + p.L) {
      kind = 1; tk = j - p.Wk; fl = 0;
    }
    ktok[tid] = tk; kkind[tid] = kind; kflag[tid] = fl;
  }
  __syncthreads();
  for (int idx = tid; idx < KC * C4; idx += NT) {
    const int r = idx / C4, c = (idx - r * C4) * 4;
    const int tk = ktok[r], kind = kkind[r];
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
    if (tk >= 0 && kind == 0) {
      kv = *reinterpret_cast<const f32x4*>(p.k.p + (size_t)b * p.k.sb + (size_t)h * p.k.sh + (size_t)tk * p.k.sn + c);
      if (!(p.zero_mv && kflag[r] != 0))
        vv = *reinterpret_cast<const f32x4*>(p.v.p + (size_t)b * p.v.sb + (size_t)h * p.v.sh + (size_t)tk * p.v.sn + c);
    } else if (kind == 1) {
      kv = *reinterpret_cast<const f32x4*>(p.ek.p + (size_t)b * p.ek.sb + (size_t)h * p.ek.sh + (size_t)tk * p.ek.sn + c);
      vv = *reinterpret_cast<const f32x4*>(p.ev.p + (size_t)b * p.ev.sb + (size_t)h * p.ev.sh + (size_t)tk * p.ev.sn + c);
    }
    float* dk_ = Ks + r * LD + c;
    float* dv_ = Vs + r * LD + c;
    dk_[0] = kv[0]; dk_[1] = kv[1]; dk_[2] = kv[2]; dk_[3] = kv[3];
    dv_[0] = vv[0]; dv_[1] = vv[1]; dv_[2] = vv[2]; dv_[3] = vv[3];
  }
  __syncthreads();
  if (tid < KC) {
    float a = 0.f;
    if (p.knorm && kkind[tid] == 0) {
      float s = 0.f;
      for (int d = 0; d < D; ++d) s += Ks[tid * LD + d] * Ks[tid * LD + d];
      a = -0.5f * p.scale * s;
    }
    kadd[tid] = a;
  }
  __syncthreads();
}

// logit of (query row i of the block, key jl of the chunk) from the raw product `dot`; *live: the entry keeps its gradient
struct QInfo { int tok, slot, lm_lim; bool pad; };
EA_DEV float logit_of(const GaP& p, float dot, int b, int h, const QInfo& qi, int j, int kind, int kflag, float kadd, bool& live) {
  live = false;
  if (kind == 2 || kflag == 2) return -INFINITY;
  float x = dot * p.scale + kadd;
  if (kind == 0) {
    if (p.bias && qi.tok >= 0) x += p.bias[(size_t)b * p.bias_bs + (size_t)h * p.bias_hs + (size_t)qi.slot * p.bias_ld + j];
    if (kflag == 1 || qi.pad || (p.causal_e >= 0 && j > qi.slot + p.causal_e)) return MASK_FILL;
  } else {
    if (p.chunk > 0 && (j - p.Wk) >= qi.lm_lim) return MASK_FILL;
  }
  live = true;
  return x;
}

EA_DEV QInfo query_info(const GaP& p, int b, int g, int slot) {
  QInfo q;
  q.slot = slot;
  q.tok = slot < p.Wq ? p.idx_q[(size_t)g * p.Wq + slot] : -1;
  q.pad = q.tok >= 0 && p.qmask && p.qmask[(size_t)b * p.Nq + q.tok];
  q.lm_lim = (p.chunk > 0 && q.tok >= 0) ? p.lm_base + q.tok / p.chunk : 0x7fffffff;
  return q;
}

template <int D, int KC>
__global__ __launch_bounds__(NT) void ga_fwd_kernel(const GaP p) {
  constexpr int LD = D + 1, DT = D / 16, CT = KC / 16, PLD = KC + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                       // [QB][LD]
  float* Ks = Qs + QB * LD;             // [KC][LD]
  float* Vs = Ks + KC * LD;             // [KC][LD]
  float* Ps = Vs + KC * LD;             // [QB][PLD]
  float* kadd = Ps + QB * PLD;          // [KC]
  int* ktok = reinterpret_cast<int*>(kadd + KC);
  int* kkind = ktok + KC;
  int* kflag = kkind + KC;
  int* qtok = kflag + KC;               // [QB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, li = lane & 15;
  const int nqb = (p.Wq + QB - 1) / QB;
  const int g = blockIdx.x / nqb, qb = blockIdx.x - g * nqb;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  if (tid < QB) {
    const int slot = qb * QB + tid;
    qtok[tid] = slot < p.Wq ? p.idx_q[(size_t)g * p.Wq + slot] : -1;
  }
  __syncthreads();
  load_rows<D>(Qs, p.q, b, h, qtok, QB, tid);
  QInfo qi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qi[r] = query_info(p, b, g, qb * QB + 16 * wave + 4 * gq + r);
  float m[4], l[4];
  f32x4 oacc[DT];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.Wk + p.L;
  for (int kc0 = 0; kc0 < nk; kc0 += KC) {
    stage_keys<D, KC>(p, b, h, g, kc0, Ks, Vs, ktok, kkind, kflag, kadd, tid);
    f32x4 s[CT];
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, true, D>(acc, Qs, LD, Ks, LD, 16 * wave, 16 * ct, D, lane);
      const int jl = 16 * ct + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bool live;
        s[ct][r] = logit_of(p, acc[r], b, h, qi[r], kc0 + jl, kkind[jl], kflag[jl], kadd[jl], live);
        mx[r] = fmaxf(mx[r], s[ct][r]);
      }
    }
    float alpha[4], ps[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m[r], group_max16(mx[r]));
      alpha[r] = mn == -INFINITY ? 1.f : __expf(m[r] - mn);
      m[r] = mn;
      ps[r] = 0.f;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int jl = 16 * ct + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = m[r] == -INFINITY ? 0.f : __expf(s[ct][r] - m[r]);
        ps[r] += pv;
        float kf = 1.f;
        if (p.keep && qi[r].tok >= 0 && kkind[jl] != 2)     // (columns past the key list: the mask row ends at Wk + L)
          kf = p.keep[((size_t)(b * p.H + h) * p.Nq + qi[r].tok) * p.keep_ld + kc0 + jl] ? p.keep_scale : 0.f;
        Ps[(16 * wave + 4 * gq + r) * PLD + jl] = pv * kf;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) l[r] = l[r] * alpha[r] + group_sum<16>(ps[r]);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha[r];
      tile_mm<false, false, KC>(oacc[dt], Ps, PLD, Vs, LD, 16 * wave, 16 * dt, KC, lane);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (qi[r].tok < 0) continue;
    const float inv = 1.f / l[r];
    float* o = const_cast<float*>(p.o.p) + (size_t)b * p.o.sb + (size_t)h * p.o.sh + (size_t)qi[r].tok * p.o.sn;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[16 * dt + li] = oacc[dt][r] * inv;
    if (li == 0) {
      const size_t o = (size_t)(b * p.H + h) * p.Nq + qi[r].tok;
      if (p.lse) p.lse[o] = m[r] + __logf(l[r]);
      if (p.stat) { p.stat[2 * o] = m[r]; p.stat[2 * o + 1] = l[r]; }
    }
  }
}

template <int D, int KC>
__global__ __launch_bounds__(NT) void ga_bwd_kernel(const GaP p) {
  constexpr int LD = D + 1, DT = D / 16, CT = KC / 16, PLD = KC + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Qs = sm;                       // [QB][LD]
  float* dOs = Qs + QB * LD;            // [QB][LD]
  float* Ks = dOs + QB * LD;            // [KC][LD]   (prologue: the out rows, spanning Ks and Vs)
  float* Vs = Ks + KC * LD;             // [KC][LD]
  float* Ps = Vs + KC * LD;             // [QB][PLD]  P (with the dropout factor)
  float* dSs = Ps + QB * PLD;           // [QB][PLD]
  float* kadd = dSs + QB * PLD;         // [KC]
  float* csum = kadd + KC;              // [KC] column sums of dS (key-norm term)
  float* delta = csum + KC;             // [QB]
  float* lses = delta + QB;             // [QB] row max
  float* linv = lses + QB;              // [QB] 1 / row sum
  int* ktok = reinterpret_cast<int*>(linv + QB);
  int* kkind = ktok + KC;
  int* kflag = kkind + KC;
  int* qtok = kflag + KC;               // [QB]
  static_assert(2 * KC * (D + 1) >= QB * (D + 1), "the out rows of the prologue must fit the key / value images");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, li = lane & 15;
  const int nqb = (p.Wq + QB - 1) / QB;
  const int g = blockIdx.x / nqb, qb = blockIdx.x - g * nqb;
  const int b = blockIdx.y / p.H, h = blockIdx.y - b * p.H;
  if (tid < QB) {
    const int slot = qb * QB + tid;
    qtok[tid] = slot < p.Wq ? p.idx_q[(size_t)g * p.Wq + slot] : -1;
  }
  __syncthreads();
  load_rows<D>(Qs, p.q, b, h, qtok, QB, tid);
  load_rows<D>(dOs, p.dout, b, h, qtok, QB, tid);
  load_rows<D>(Ks, p.o, b, h, qtok, QB, tid);
  __syncthreads();
  if (tid < QB) {
    float dl = 0.f, ls = 0.f, li_ = 0.f;
    const int tk = qtok[tid];
    if (tk >= 0) {
      for (int d = 0; d < D; ++d) dl += dOs[tid * LD + d] * Ks[tid * LD + d];
      const size_t o = (size_t)(b * p.H + h) * p.Nq + tk;
      if (p.dlse) dl -= p.dlse[o];
      ls = p.stat[2 * o];
      li_ = 1.f / p.stat[2 * o + 1];
    }
    delta[tid] = dl; lses[tid] = ls; linv[tid] = li_;
  }
  __syncthreads();
  QInfo qi[4];
  float dlt[4], lsr[4], lir[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wave + 4 * gq + r;
    qi[r] = query_info(p, b, g, qb * QB + row);
    dlt[r] = delta[row]; lsr[r] = lses[row]; lir[r] = linv[row];
  }
  f32x4 dqacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dqacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.Wk + p.L;
  const size_t bh = (size_t)b * p.H + h;
  for (int kc0 = 0; kc0 < nk; kc0 += KC) {
    stage_keys<D, KC>(p, b, h, g, kc0, Ks, Vs, ktok, kkind, kflag, kadd, tid);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = acc;
      tile_mm<false, true, D>(acc, Qs, LD, Ks, LD, 16 * wave, 16 * ct, D, lane);
      tile_mm<false, true, D>(dp, dOs, LD, Vs, LD, 16 * wave, 16 * ct, D, lane);
      const int jl = 16 * ct + li, j = kc0 + jl;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bool live;
        const float x = logit_of(p, acc[r], b, h, qi[r], j, kkind[jl], kflag[jl], kadd[jl], live);
        float pv = 0.f, ds = 0.f, kf = 1.f;
        if (qi[r].tok >= 0 && x != -INFINITY) {
          pv = __expf(x - lsr[r]) * lir[r];
          if (p.keep) kf = p.keep[(bh * p.Nq + qi[r].tok) * p.keep_ld + j] ? p.keep_scale : 0.f;
          ds = pv * (dp[r] * kf - dlt[r]);
          if (!live) ds = 0.f;
          else if (p.dbias && kkind[jl] == 0)
            unsafeAtomicAdd(p.dbias + (size_t)b * p.bias_bs + (size_t)h * p.bias_hs + (size_t)qi[r].slot * p.bias_ld + j, ds);
        }
        const int row = 16 * wave + 4 * gq + r;
        Ps[row * PLD + jl] = pv * kf;
        dSs[row * PLD + jl] = ds;
      }
    }
    __syncthreads();
    if (p.knorm && tid < KC) {
      float s = 0.f;
      for (int i = 0; i < QB; ++i) s += dSs[i * PLD + tid];
      csum[tid] = s;
    }
    // dq += dS K
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) tile_mm<false, false, KC>(dqacc[dt], dSs, PLD, Ks, LD, 16 * wave, 16 * dt, KC, lane);
    if (p.knorm) __syncthreads();
    // dk = s dS^T Q (- s k colsum(dS)),  dv = P^T dO : [KC x D] tiles over the four waves
    for (int t = wave; t < CT * DT; t += 4) {
      const int kt = t / DT, dt = t - kt * DT;
      f32x4 dk_ = {0.f, 0.f, 0.f, 0.f}, dv_ = dk_;
      tile_mm<true, false, QB>(dk_, dSs, PLD, Qs, LD, 16 * kt, 16 * dt, QB, lane);
      tile_mm<true, false, QB>(dv_, Ps, PLD, dOs, LD, 16 * kt, 16 * dt, QB, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jl = 16 * kt + 4 * gq + r;
        const int tk = ktok[jl], kind = kkind[jl];
        if (tk < 0 || kind == 2) continue;
        const int c = 16 * dt + li;
        float dkv = dk_[r] * p.scale;
        if (p.knorm && kind == 0) dkv -= p.scale * Ks[jl * LD + c] * csum[jl];
        float* const bk_ = kind == 0 ? p.dk : p.dek;
        float* const bv_ = kind == 0 ? p.dv : p.dev;
        const size_t off = (kind == 0 ? bh * p.Nk + tk : bh * p.L + tk) * D + c;
        if (bk_) unsafeAtomicAdd(bk_ + off, dkv);
        if (bv_ && !(p.zero_mv && kind == 0 && kflag[jl] != 0)) unsafeAtomicAdd(bv_ + off, dv_[r]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (qi[r].tok < 0) continue;
    float* o = const_cast<float*>(p.dq.p) + (size_t)b * p.dq.sb + (size_t)h * p.dq.sh + (size_t)qi[r].tok * p.dq.sn;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[16 * dt + li] = dqacc[dt][r] * p.scale;
  }
}

// ---- masked means of gathered rows (EVA's chunk means, eva.py:167-181; uniform 2-D pooling) ----
//   mean[b,h,c,:] = (1/J) sum_j x[b,h,idx[c,j],:] [token present and not padded]
__global__ __launch_bounds__(256) void gm_fwd_kernel(const GmP p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int C4 = p.D / 4;
  const size_t total = (size_t)p.B * p.H * p.Cn * C4;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  const size_t r = i / C4;
  const int c = (int)(r % p.Cn);
  const size_t bh = r / p.Cn;
  const int b = (int)(bh / p.H), h = (int)(bh - (size_t)b * p.H);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < p.J; ++j) {
    const int tk = p.idx[(size_t)c * p.J + j];
    if (tk < 0 || (p.mask && p.mask[(size_t)b * p.N + tk])) continue;
    acc += *reinterpret_cast<const f32x4*>(p.x.p + (size_t)b * p.x.sb + (size_t)h * p.x.sh + (size_t)tk * p.x.sn + c4 * 4);
  }
  const float inv = 1.f / (float)p.J;
  *reinterpret_cast<f32x4*>(p.mean + (bh * p.Cn + c) * p.D + c4 * 4) = acc * inv;
}
__global__ __launch_bounds__(256) void gm_bwd_kernel(const GmP p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int C4 = p.D / 4;
  const size_t total = (size_t)p.B * p.H * p.Cn * C4;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  const size_t r = i / C4;
  const int c = (int)(r % p.Cn);
  const size_t bh = r / p.Cn;
  const int b = (int)(bh / p.H);
  const f32x4 gmean = *reinterpret_cast<const f32x4*>(p.dmean + (bh * p.Cn + c) * p.D + c4 * 4) * (1.f / (float)p.J);
  for (int j = 0; j < p.J; ++j) {
    const int tk = p.idx[(size_t)c * p.J + j];
    if (tk < 0 || (p.mask && p.mask[(size_t)b * p.N + tk])) continue;
    float* d = p.dx + (bh * p.N + tk) * p.D + c4 * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(d + e, gmean[e]);
  }
}

template <int D, int KC> size_t ga_lds(bool bwd) {
  const size_t LD = D + 1, PLD = KC + 1;
  if (!bwd) return (QB * LD + 2 * KC * LD + QB * PLD + KC) * 4 + (3 * KC + QB) * 4;
  return (2 * QB * LD + 2 * KC * LD + 2 * QB * PLD + 2 * KC + 3 * QB) * 4 + (3 * KC + QB) * 4;
}

template <int D, int KC>
int ga_launch(bool bwd, const GaP& p, hipStream_t st) {
  const size_t lds = ga_lds<D, KC>(bwd);
  const int nqb = (p.Wq + QB - 1) / QB;
  const dim3 grid((unsigned)(p.G * nqb), (unsigned)(p.B * p.H)), block(NT);
  if (bwd) {
    EA_SET_LDS_ONCE((&ga_bwd_kernel<D, KC>), lds);
    hipLaunchKernelGGL((ga_bwd_kernel<D, KC>), grid, block, lds, st, p);
  } else {
    EA_SET_LDS_ONCE((&ga_fwd_kernel<D, KC>), lds);
    hipLaunchKernelGGL((ga_fwd_kernel<D, KC>), grid, block, lds, st, p);
  }
  return (int)hipGetLastError();
}

}  // namespace

int ga_dispatch(bool bwd, const GaP& p, hipStream_t st) {
  if (p.G <= 0 || p.B <= 0 || p.H <= 0 || (long)p.B * p.H > 65535) return EA_E_BADARG;
  if (p.D == 64) return ga_launch<64, 64>(bwd, p, st);
  if (p.D == 32) return ga_launch<32, 64>(bwd, p, st);
  if (p.D == 128) return ga_launch<128, 32>(bwd, p, st);
  return EA_E_UNSUPPORTED;
}

int gm_dispatch(bool bwd, const GmP& p, hipStream_t st) {
  const size_t total = (size_t)p.B * p.H * p.Cn * (p.D / 4);
  if (!total) return EA_OK;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (bwd) hipLaunchKernelGGL(gm_bwd_kernel, grid, block, 0, st, p);
  else hipLaunchKernelGGL(gm_fwd_kernel, grid, block, 0, st, p);
  return (int)hipGetLastError();
}

}  // namespace ea
