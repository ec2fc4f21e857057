// ea_scatter.hip -- ScatterBrain's low-rank half (scatterbrain_attention.py:99-160), one workgroup per window.
//
// Per window g and query i the reference runs ONE softmax over the window's keys and m random-feature
// columns whose logits are  log phi(q_i)[c] + log(sum_{j outside g} phi(k_j)[c])  and whose values are the
// phi-weighted mean of v outside the window.  The window half is the HIP window kernel (it hands back its
// per-query log-sum-exp); this file is the feature half and the exact merge through that log-sum-exp:
//     LQ = a q W^T - b |q|^2 - log(m)/2,   PK = exp(a k W^T - b |k|^2 - mx)            (a = d^-1/4, b = d^-1/2 / 2)
//     z_win = sum_j PK_j,  S_win = PK^T V          (this window)       z_all, S_all, mx: whole sequence
//     KV = (S_all - S_win) / clamp(z_all - z_win, 1e-3)
//     R_ic = LQ_ic + log z_all_c + mx_c - log(m)/2 + log(1 - z_win_c / z_all_c + 1e-5)
//     r_i = LSE_c R_ic,  O_i = softmax_c(R_i) KV,   out_i = sigma(lse_loc_i - r_i) o_loc_i + sigma(r_i - lse_loc_i) O_i
// Everything is 64 x 64 matrix algebra on the window's <= 64 tokens, m <= 64 features and d = 64 channels:
// strips in registers, 16-bit operand tiles in LDS (ea_strip.h), three barrier phases forward.
// The sequence-wide statistics (mx, z_all, S_all) come from the Performer token-row passes with a
// per-feature stabiliser (ea_lara_y.hip, LaraP::stab_per_feature).
#include "ea_strip.h"
#include "ea_scatter.h"

namespace ea {

using namespace strip;

namespace {

constexpr int TB = 64 * 128;      // bytes of a [64][64] 16-bit tile

// Rows of window `win` (slot -> token) of a [B,H,N,64] tensor -> row-major swizzled tile, in two steps so that
// every global load of a phase is in flight before the first use: the loads are unconditional (slots beyond
// the window are clamped and zeroed at commit -- a predicated load is a branch plus a full round trip each).
struct RowRegs { u32x4 v[2]; };
EA_DEV void issue_window(RowRegs& rr, const T4s& t, int b, int h, const Geo& G, int win, int w, int Wq, int tid) {
  const char* base = t.p + (b * t.sb + h * t.sh) * 2;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;
    const int row = min(idx >> 3, Wq - 1), ch = idx & 7;
    const int tok = part_token(G, win, row, w, 0);
    rr.v[k] = ldg16(base + ((size_t)tok * t.sn + ch * 8) * 2);
  }
}
template <typename E>
EA_DEV void commit_window(char* tile, const RowRegs& rr, int Wq, int tid, float* sqnorm /* [64] or null */) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;
    const int row = idx >> 3, ch = idx & 7;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const u32x4 v = row < Wq ? rr.v[k] : z;
    sts16(tile + lds_off<64>(row, ch), v);
    if (sqnorm) {
      float f[8], part = 0.f;
      unpack8<E>(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += f[e] * f[e];
      part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4);
      if (ch == 0) sqnorm[row] = part;
    }
  }
}

// fp32 [rows][64] -> 16-bit row-major tile (zero beyond rows)
struct F32Regs { float4 v[2][2]; };
EA_DEV void issue_f32(F32Regs& fr, const float* src, int rows, int tid) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;
    const int row = min(idx >> 3, rows - 1), ch = idx & 7;
    fr.v[k][0] = *reinterpret_cast<const float4*>(src + (size_t)row * 64 + ch * 8);
    fr.v[k][1] = *reinterpret_cast<const float4*>(src + (size_t)row * 64 + ch * 8 + 4);
  }
}
template <typename E>
EA_DEV void commit_f32(char* tile, const F32Regs& fr, int rows, int tid) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * 256;
    const int row = idx >> 3, ch = idx & 7;
    const bool ok = row < rows;
    const float4 lo = fr.v[k][0], hi = fr.v[k][1];
    const float f[8] = {ok ? lo.x : 0.f, ok ? lo.y : 0.f, ok ? lo.z : 0.f, ok ? lo.w : 0.f,
                        ok ? hi.x : 0.f, ok ? hi.y : 0.f, ok ? hi.z : 0.f, ok ? hi.w : 0.f};
    sts16(tile + lds_off<64>(row, ch), pack8<E>(f));
  }
}

// strip (rows of this wave) -> the wave's 16 rows of a row-major tile (2-byte stores), to be read back by the
// same wave as 16-byte row chunks
template <typename E>
EA_DEV void strip_to_rows(char* tile, const f32x4* s, const Lane& l) {
  const int r0 = 16 * l.w + 4 * l.g;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<uint16_t*>(tile + toff<64>(r0 + r, 16 * ct + l.li)) = E::from_f(s[ct][r]);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256, 3) void sb_fwd_kernel(const SbP p) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* TQ = sm;                 // Q rows        -> AT  [c][i]
  char* TK = TQ + TB;            // K rows        -> KVT [d][c]
  char* TV = TK + TB;            // V rows
  char* TW = TV + TB;            // feature matrix W_h [m][d]
  char* TP = TW + TB;            // PKT [c][j]    -> output rows
  float* qn = reinterpret_cast<float*>(TP + TB);   // [64] |q|^2
  float* kn = qn + 64;                              // [64] |k|^2
  float* kdead = kn + 64;                           // [64] 1 = masked / absent key
  float* cpart = kdead + 64;                        // [4][64]
  float* nlv = cpart + 256;                         // [64] nonlocal log-mass per feature

  const int tid = threadIdx.x, lane = tid & 63;
  Lane l;
  l.w = tid >> 6; l.g = lane >> 4; l.li = lane & 15;
  const int win = blockIdx.x % p.nwin, bh = blockIdx.x / p.nwin;
  const int b = bh / p.H, h = bh - b * p.H;
  const int M = p.M, Wq = p.Wq;
  const int r0 = 16 * l.w + 4 * l.g;

  EA_STAMP(p, 0);
  EA_BLK(p, 0);
  // P0: rows of the window, feature matrix, per-row scalars -- all loads first
  RowRegs rq, rk, rv;
  F32Regs rw;
  issue_window(rq, p.q, b, h, p.G, win, p.w, Wq, tid);
  issue_window(rk, p.k, b, h, p.G, win, p.w, Wq, tid);
  issue_window(rv, p.v, b, h, p.G, win, p.w, Wq, tid);
  issue_f32(rw, p.Wf + (size_t)h * M * 64, M, tid);
  float dead_ = 1.f;
  if (tid < 64) {
    const int tok = part_token(p.G, win, min(tid, Wq - 1), p.w, 0);
    dead_ = (tid >= Wq || (p.mask && p.mask[(size_t)b * p.N + tok])) ? 1.f : 0.f;
  }
  float llr[4];                                       // lse_loc of this lane's rows (used in P3)
  int tokr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    tokr[r] = part_token(p.G, win, min(r0 + r, Wq - 1), p.w, 0);
    llr[r] = p.lse_loc[(size_t)bh * p.N + tokr[r]];
  }
  // per-feature scalars of the columns this lane sees (c = 16 ct + li) and of its rows (c = r0 + r)
  float mxc[4], zac[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c = min(16 * ct + l.li, M - 1);
    mxc[ct] = p.mx[(size_t)bh * M + c];
  }
  float zar[4], mxr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = min(r0 + r, M - 1);
    zar[r] = p.zall[(size_t)bh * M + c];
    mxr[r] = p.mx[(size_t)bh * M + c];
  }
  (void)zac;
  f32x4 sall[4];
  load_strip<4>(sall, p.sall + (size_t)bh * M * 64, 64, M, 64, l);
  commit_window<E>(TQ, rq, Wq, tid, qn);
  commit_window<E>(TK, rk, Wq, tid, kn);
  commit_window<E>(TV, rv, Wq, tid, nullptr);
  commit_f32<E>(TW, rw, M, tid);
  if (tid < 64) kdead[tid] = dead_;
  __syncthreads();

  EA_STAMP(p, 1);
  // P1: PK strip (rows j), LQ strip (rows i)
  f32x4 pk[4], lq[4];
  zero<4>(pk); zero<4>(lq);
  mm<E, 64, false, 64, true, 4>(pk, TK, TW, 2, l);
  mm<E, 64, false, 64, true, 4>(lq, TQ, TW, 2, l);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + r, c = 16 * ct + l.li;
      const bool ok = c < M && kdead[row] == 0.f;
      pk[ct][r] = ok ? __expf(p.a * pk[ct][r] - p.b * kn[row] - mxc[ct]) : 0.f;
      lq[ct][r] = p.a * lq[ct][r] - p.b * qn[row] - p.lconst;
    }
  colsum_part<4>(cpart, pk, 64, l);
  store_t<E, 4>(TP, pk, 1.f, 64, M, l);              // PKT [c][j]
  __syncthreads();

  EA_STAMP(p, 2);
  // P2: window statistics of the feature rows c of this strip
  {
    f32x4 sw[4];
    zero<4>(sw);
    mm<E, 64, false, 64, false, 4>(sw, TP, TV, 2, l);                 // S_win = PK^T V
    f32x4 kv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = r0 + r;
      const float zw = c < 64 ? cpart[c] + cpart[64 + c] + cpart[128 + c] + cpart[192 + c] : 0.f;
      const float inv = 1.f / fmaxf(zar[r] - zw, 1e-3f);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) kv[ct][r] = c < M ? (sall[ct][r] - sw[ct][r]) * inv : 0.f;
      if (l.li == 0 && c < 64)
        nlv[c] = c < M ? __logf(zar[r]) + mxr[r] - p.lconst + __logf(1.f - zw / zar[r] + 1e-5f) : -INFINITY;
    }
    store_t<E, 4>(TK, kv, 1.f, M, 64, l);            // KVT [d][c]  (the K rows are dead: barrier above)
  }
  __syncthreads();

  EA_STAMP(p, 3);
  // P3: joint weights of the feature columns, O = A KV, merge with the window half
  f32x4 A[4];
  float rmax[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, den[4] = {0.f, 0.f, 0.f, 0.f}, rr[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * ct + l.li;
      A[ct][r] = c < M ? lq[ct][r] + nlv[c] : -INFINITY;
      rmax[r] = fmaxf(rmax[r], A[ct][r]);
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) rmax[r] = row16_max(rmax[r]);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { A[ct][r] = __expf(A[ct][r] - rmax[r]); den[r] += A[ct][r]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) { den[r] = row16_sum(den[r]); rr[r] = rmax[r] + __logf(den[r]); den[r] = 1.f / den[r]; }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) A[ct][r] *= den[r];
  store_t<E, 4>(TQ, A, 1.f, 64, M, l);               // AT [c][i]: read back by this wave only
  f32x4 O[4];
  zero<4>(O);
  mm<E, 64, true, 64, true, 4>(O, TQ, TK, 2, l);     // O = A KV
  // merge: beta_i O_i now (strip layout), alpha_i o_loc_i added when the rows are written out
  float beta[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tok = tokr[r];
    const float ll = llr[r];
    const float z = fmaxf(ll, rr[r]) + __logf(1.f + __expf(-fabsf(ll - rr[r])));
    beta[r] = __expf(rr[r] - z);
    if (l.li == 0 && r0 + r < Wq) {
      p.r[(size_t)bh * p.N + tok] = rr[r];
      qn[r0 + r] = __expf(ll - z);                   // alpha_i (qn is dead)
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) O[ct][r] *= beta[r];
  EA_STAMP(p, 4);
  strip_to_rows<E>(TP, O, l);                        // (PKT is dead: every wave passed the barrier of P2 ... P3)
  // each wave writes out its own 16 rows: out = alpha o_loc + beta O
  {
    const char* ob = p.oloc.p + (b * p.oloc.sb + h * p.oloc.sh) * 2;
    char* outb = p.out.p + (b * p.out.sb + h * p.out.sh) * 2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = lane + k * 64;
      const int row = 16 * l.w + (idx >> 3), ch = idx & 7;
      if (row < Wq) {
        const int tok = part_token(p.G, win, row, p.w, 0);
        float fo[8], fl[8];
        unpack8<E>(lds16(TP + lds_off<64>(row, ch)), fo);
        unpack8<E>(ldg16(ob + ((size_t)tok * p.oloc.sn + ch * 8) * 2), fl);
        const float al = qn[row];
#pragma unroll
        for (int e = 0; e < 8; ++e) fo[e] += al * fl[e];
        stg16(outb + ((size_t)tok * p.out.sn + ch * 8) * 2, pack8<E>(fo));
      }
    }
  }
  EA_STAMP(p, 5);
  EA_BLK(p, 1);
}

// ------------------------------------------------------------------------------------------
// backward.  GLOBAL = false: the window pass -- everything that is local to a window (dq, the window's own
//   contributions to dk / dv, d o_loc, d lse_loc) plus this workgroup's partial sums of d S_all / d z_all
//   over its wpb consecutive windows.  GLOBAL = true: the pass that spreads d S_all, d z_all back over
//   every key (accumulating into dk, dv of the window pass).
// With beta_i = sigma(r_i - lse_loc_i), alpha_i = 1 - beta_i and g_i = dout_i:
//     d o_loc_i = alpha_i g_i,   d lse_loc_i = alpha_i beta_i (o_loc_i - O_i) . g_i = - d r_i
//     dA_ic = beta_i g_i . KV_c,   dR_ic = A_ic (dA_ic - beta_i O_i . g_i + d r_i)
//     dq_i = a dR_i W - 2 b q_i sum_c dR_ic
//     dKV_c = sum_i beta_i A_ic g_i;  d S_all_c += dKV_c / den_c,  d S_win_c = - dKV_c / den_c
//     d den_c = - dKV_c . KV_c / den_c (where unclamped);  d nl_c = sum_i dR_ic
//     d z_all_c += d den_c + d nl_c (1 / z_all + z_win / (z_all^2 u)),   d z_win_c = - d den_c - d nl_c / (z_all u)
//     dPK_jc = d z_win_c + v_j . d S_win_c,  dLK = PK o dPK,  dk_j = a dLK_j W - 2 b k_j sum_c dLK_jc,  dv_j = PK_j d S_win
// ------------------------------------------------------------------------------------------
template <typename E, bool GLOBAL>
__global__ __launch_bounds__(256, GLOBAL ? 3 : 1) void sb_bwd_kernel(const SbP p) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  // (the global pass uses six of the nine tiles: TQ, TD and the dedicated TO alias others there)
  char* TK = sm;                 // K rows -> KVT [d][c]
  char* TV = TK + TB;            // V rows
  char* TW = TV + TB;            // W_h [m][d]
  char* TP = TW + TB;            // PKT [c][j]
  char* TR = TP + TB;            // DRT [c][i] -> DLT [c][j]
  char* TS = TR + TB;            // DST [d][c]  (d S_win, or d S_all in the global pass)
  char* TQ = GLOBAL ? TK : TS + TB;     // Q rows -> AT [c][i] -> (beta A)T
  char* TD = GLOBAL ? TK : TQ + TB;     // dout rows
  char* TO = GLOBAL ? TK : TD + TB;     // output rows (global pass: over the K rows, dead once PK exists)
  float* qn = reinterpret_cast<float*>((GLOBAL ? TS : TO) + TB);   // [64]
  float* kn = qn + 64;
  float* kdead = kn + 64;
  float* al_s = kdead + 64;      // alpha_i
  float* be_s = al_s + 64;       // beta_i
  float* od_s = be_s + 64;       // o_loc_i . g_i
  float* nlv = od_s + 64;        // nonlocal log-mass per feature
  float* dzw = nlv + 64;         // d z_win (or d z_all) per feature
  float* cpart = dzw + 64;       // [4][64]
  float* cpartB = cpart + 256;   // [4][64]

  const int tid = threadIdx.x, lane = tid & 63;
  Lane l;
  l.w = tid >> 6; l.g = lane >> 4; l.li = lane & 15;
  const int npb = p.nwin / p.wpb;                  // workgroups per (b,h)
  const int bh = blockIdx.x / npb, np = blockIdx.x - bh * npb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int M = p.M, Wq = p.Wq;
  const int r0 = 16 * l.w + 4 * l.g;

  {
    F32Regs rw;
    issue_f32(rw, p.Wf + (size_t)h * M * 64, M, tid);
    commit_f32<E>(TW, rw, M, tid);
  }
  float mxc[4], zar[4], mxr[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) mxc[ct] = p.mx[(size_t)bh * M + min(16 * ct + l.li, M - 1)];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = min(r0 + r, M - 1);
    zar[r] = p.zall[(size_t)bh * M + c];
    mxr[r] = p.mx[(size_t)bh * M + c];
  }
  f32x4 gs[4];
  float gz[4] = {0.f, 0.f, 0.f, 0.f};
  zero<4>(gs);
  if (GLOBAL) {
    f32x4 ds[4];
    load_strip<4>(ds, p.dsall + (size_t)bh * M * 64, 64, M, 64, l);         // d S_all rows c of this strip
    store_t<E, 4>(TS, ds, 1.f, M, 64, l);                                    // DSAT [d][c]
    if (tid < 64) dzw[tid] = tid < M ? p.dzall[(size_t)bh * M + tid] : 0.f;
  }
  const char* qb = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* kb = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  char* dqb = p.dq.p + (b * p.dq.sb + h * p.dq.sh) * 2;
  char* dkb = p.dk.p + (b * p.dk.sb + h * p.dk.sh) * 2;
  char* dvb = p.dv.p + (b * p.dv.sb + h * p.dv.sh) * 2;

  // the wave's 16 rows of TO -> global rows of window `win`: dst = (acc ? dst : 0) + TO - coef[row] * src
  auto emit = [&](int win, char* dst, int64_t dsn, const char* src, int64_t ssn, const float* coef, bool acc) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = lane + k * 64;
      const int row = 16 * l.w + (idx >> 3), ch = idx & 7;
      if (row < Wq) {
        const int tok = part_token(p.G, win, row, p.w, 0);
        float fo[8], fs[8];
        unpack8<E>(lds16(TO + lds_off<64>(row, ch)), fo);
        if (src) {
          unpack8<E>(ldg16(src + ((size_t)tok * ssn + ch * 8) * 2), fs);
          const float cf = coef[row];
#pragma unroll
          for (int e = 0; e < 8; ++e) fo[e] -= cf * fs[e];
        }
        char* d = dst + ((size_t)tok * dsn + ch * 8) * 2;
        if (acc) {
          unpack8<E>(ldg16(d), fs);
#pragma unroll
          for (int e = 0; e < 8; ++e) fo[e] += fs[e];
        }
        stg16(d, pack8<E>(fo));
      }
    }
  };

  // rows of a window, prefetched one window ahead: the loads of window wi + 1 are in flight while wi computes
  const char* gb = GLOBAL ? nullptr : p.dout.p + (b * p.dout.sb + h * p.dout.sh) * 2;
  const char* ob = GLOBAL ? nullptr : p.oloc.p + (b * p.oloc.sb + h * p.oloc.sh) * 2;
  char* dob = GLOBAL ? nullptr : p.doloc.p + (b * p.doloc.sb + h * p.doloc.sh) * 2;
  RowRegs rq, rk, rv, rg, ro;
  float pll[2], prr[2], pdead = 1.f;
  auto issue = [&](int win) {
    if (!GLOBAL) issue_window(rq, p.q, b, h, p.G, win, p.w, Wq, tid);
    issue_window(rk, p.k, b, h, p.G, win, p.w, Wq, tid);
    issue_window(rv, p.v, b, h, p.G, win, p.w, Wq, tid);
    if (!GLOBAL) {
      issue_window(rg, p.dout, b, h, p.G, win, p.w, Wq, tid);
      issue_window(ro, p.oloc, b, h, p.G, win, p.w, Wq, tid);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int tok = part_token(p.G, win, min((tid + k * 256) >> 3, Wq - 1), p.w, 0);
        pll[k] = p.lse_loc[(size_t)bh * p.N + tok];
        prr[k] = p.r[(size_t)bh * p.N + tok];
      }
    }
    if (tid < 64) {
      const int tok = part_token(p.G, win, min(tid, Wq - 1), p.w, 0);
      pdead = (tid >= Wq || (p.mask && p.mask[(size_t)b * p.N + tok])) ? 1.f : 0.f;
    }
  };
  issue(np * p.wpb);

  for (int wi = 0; wi < p.wpb; ++wi) {
    const int win = np * p.wpb + wi;
    __syncthreads();                                   // tiles of the previous window are free (and TW / TS staged)
    // P0: this window's rows -> tiles
    if (!GLOBAL) commit_window<E>(TQ, rq, Wq, tid, qn);
    commit_window<E>(TK, rk, Wq, tid, kn);
    commit_window<E>(TV, rv, Wq, tid, nullptr);
    if (tid < 64) kdead[tid] = pdead;
    if (!GLOBAL) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + k * 256;
        const int row = idx >> 3, ch = idx & 7;
        const bool ok = row < Wq;
        const float z = fmaxf(pll[k], prr[k]) + __logf(1.f + __expf(-fabsf(pll[k] - prr[k])));
        const float al = ok ? __expf(pll[k] - z) : 0.f, be = ok ? __expf(prr[k] - z) : 0.f;
        const u32x4 zz = {0u, 0u, 0u, 0u};
        const u32x4 gv = ok ? rg.v[k] : zz;
        float fg[8], fo[8], part = 0.f;
        unpack8<E>(gv, fg); unpack8<E>(ro.v[k], fo);
#pragma unroll
        for (int e = 0; e < 8; ++e) { part += fg[e] * fo[e]; fo[e] = al * fg[e]; }
        if (ok) {
          const int tok = part_token(p.G, win, row, p.w, 0);
          stg16(dob + ((size_t)tok * p.doloc.sn + ch * 8) * 2, pack8<E>(fo));        // d o_loc = alpha g
        }
        part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4);
        sts16(TD + lds_off<64>(row, ch), gv);
        if (ch == 0) { al_s[row] = al; be_s[row] = be; od_s[row] = ok ? part : 0.f; }
      }
    }
    __syncthreads();
    if (wi + 1 < p.wpb) issue(win + 1);

    // P1: PK strip (rows j), LQ strip (rows i)
    f32x4 pk[4], lq[4];
    zero<4>(pk); zero<4>(lq);
    mm<E, 64, false, 64, true, 4>(pk, TK, TW, 2, l);
    if (!GLOBAL) mm<E, 64, false, 64, true, 4>(lq, TQ, TW, 2, l);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + r, c = 16 * ct + l.li;
        const bool ok = c < M && kdead[row] == 0.f;
        pk[ct][r] = ok ? __expf(p.a * pk[ct][r] - p.b * kn[row] - mxc[ct]) : 0.f;
        lq[ct][r] = p.a * lq[ct][r] - p.b * qn[row] - p.lconst;
      }
    store_t<E, 4>(TP, pk, 1.f, 64, M, l);              // PKT [c][j]
    f32x4 kv[4];
    float inv[4], zw[4];
    if (!GLOBAL) {
      colsum_part<4>(cpart, pk, 64, l);
      __syncthreads();
      // P2: window statistics of the feature rows c of this strip
      f32x4 sw[4], sall[4];
      load_strip<4>(sall, p.sall + (size_t)bh * M * 64, 64, M, 64, l);       // (L2-resident; not kept across windows)
      zero<4>(sw);
      mm<E, 64, false, 64, false, 4>(sw, TP, TV, 2, l);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = r0 + r;
        zw[r] = c < 64 ? cpart[c] + cpart[64 + c] + cpart[128 + c] + cpart[192 + c] : 0.f;
        inv[r] = 1.f / fmaxf(zar[r] - zw[r], 1e-3f);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) kv[ct][r] = c < M ? (sall[ct][r] - sw[ct][r]) * inv[r] : 0.f;
        if (l.li == 0 && c < 64)
          nlv[c] = c < M ? __logf(zar[r]) + mxr[r] - p.lconst + __logf(1.f - zw[r] / zar[r] + 1e-5f) : -INFINITY;
      }
      store_t<E, 4>(TK, kv, 1.f, M, 64, l);            // KVT [d][c]
      __syncthreads();

      // P3: A, O, d lse_loc, dR, dq
      f32x4 A[4];
      float rmax[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ct + l.li;
          A[ct][r] = c < M ? lq[ct][r] + nlv[c] : -INFINITY;
          rmax[r] = fmaxf(rmax[r], A[ct][r]);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) rmax[r] = row16_max(rmax[r]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { A[ct][r] = __expf(A[ct][r] - rmax[r]); den[r] += A[ct][r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) den[r] = 1.f / row16_sum(den[r]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) A[ct][r] *= den[r];
      store_t<E, 4>(TQ, A, 1.f, 64, M, l);             // AT [c][i] (own columns)
      f32x4 O[4], dA[4];
      zero<4>(O); zero<4>(dA);
      mm<E, 64, true, 64, true, 4>(O, TQ, TK, 2, l);   // O = A KV
      mm<E, 64, false, 64, false, 4>(dA, TD, TK, 2, l);  // g KV^T : B[k=d][n=c] = KVT[d][c]
      float og[4] = {0.f, 0.f, 0.f, 0.f}, be[4], al[4], dr[4], sdr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          og[r] += O[ct][r] * E::to_f(*reinterpret_cast<const uint16_t*>(TD + toff<64>(r0 + r, 16 * ct + l.li)));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        og[r] = row16_sum(og[r]);
        be[r] = be_s[r0 + r]; al[r] = al_s[r0 + r];
        const float dl = al[r] * be[r] * (od_s[r0 + r] - og[r]);          // d lse_loc
        dr[r] = -dl;
        if (l.li == 0 && r0 + r < Wq)
          p.dlse[(size_t)bh * p.N + part_token(p.G, win, r0 + r, p.w, 0)] = dl;
      }
      f32x4 dR[4], Ab[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dR[ct][r] = (r0 + r < Wq) ? A[ct][r] * (be[r] * (dA[ct][r] - og[r]) + dr[r]) : 0.f;
          sdr[r] += dR[ct][r];
          Ab[ct][r] = (r0 + r < Wq) ? A[ct][r] * be[r] : 0.f;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) { sdr[r] = row16_sum(sdr[r]); if (l.li == 0 && r0 + r < 64) qn[r0 + r] = 2.f * p.b * sdr[r]; }
      colsum_part<4>(cpartB, dR, 64, l);               // d nl
      store_t<E, 4>(TR, dR, 1.f, 64, M, l);            // DRT [c][i] (own columns)
      store_t<E, 4>(TQ, Ab, 1.f, 64, M, l);            // (beta A)^T over AT (own columns; O is done)
      {
        f32x4 dq[4];
        zero<4>(dq);
        mm<E, 64, true, 64, false, 4>(dq, TR, TW, 2, l);    // dR W : B[k=c][n=d] = W row-major
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) dq[ct] = dq[ct] * p.a;
        strip_to_rows<E>(TO, dq, l);
        emit(win, dqb, p.dq.sn, qb, p.q.sn, qn, false);     // dq = a dR W - 2 b (sum_c dR) q
      }
      __syncthreads();

      // P4: dKV, the gradients of the window statistics, this workgroup's share of d S_all / d z_all
      f32x4 dkv[4];
      zero<4>(dkv);
      mm<E, 64, false, 64, false, 4>(dkv, TQ, TD, 2, l);    // (beta A)^T g : rows c
      f32x4 dsw[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = r0 + r;
        float dot = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) dot += dkv[ct][r] * kv[ct][r];
        dot = row16_sum(dot);
        const bool live = c < M;
        const float dden = (live && zar[r] - zw[r] >= 1e-3f) ? -dot * inv[r] : 0.f;
        const float dnl = live ? cpartB[c] + cpartB[64 + c] + cpartB[128 + c] + cpartB[192 + c] : 0.f;
        const float u = 1.f - zw[r] / zar[r] + 1e-5f;
        gz[r] += live ? dden + dnl * (1.f / zar[r] + zw[r] / (zar[r] * zar[r] * u)) : 0.f;
        const float dz = live ? -dden - dnl / (zar[r] * u) : 0.f;
        if (l.li == 0 && c < 64) dzw[c] = dz;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float t = live ? dkv[ct][r] * inv[r] : 0.f;
          gs[ct][r] += t;
          dsw[ct][r] = -t;
        }
      }
      store_t<E, 4>(TS, dsw, 1.f, M, 64, l);           // DST [d][c]
      __syncthreads();
    } else {
      __syncthreads();                                  // PKT complete
    }
    // P5: key side -- dPK = dz[c] + v . dS[c],  dLK = PK o dPK,  dk, dv
    {
      f32x4 dpk[4];
      zero<4>(dpk);
      mm<E, 64, false, 64, false, 4>(dpk, TV, TS, 2, l);    // V dS^T : B[k=d][n=c] = DST[d][c]
      float sdl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // PK of this strip is read back from PKT (own columns) rather than kept in registers across the phases
          const float pkv = E::to_f(*reinterpret_cast<const uint16_t*>(TP + toff<64>(16 * ct + l.li, r0 + r)));
          dpk[ct][r] = pkv * (dpk[ct][r] + dzw[16 * ct + l.li]);             // dLK
          sdl[r] += dpk[ct][r];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) { sdl[r] = row16_sum(sdl[r]); if (l.li == 0 && r0 + r < 64) kn[r0 + r] = 2.f * p.b * sdl[r]; }
      store_t<E, 4>(TR, dpk, 1.f, 64, M, l);           // DLT [c][j] (own columns; DRT is dead)
      f32x4 t[4];
      zero<4>(t);
      mm<E, 64, true, 64, false, 4>(t, TR, TW, 2, l);  // dLK W
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) t[ct] = t[ct] * p.a;
      strip_to_rows<E>(TO, t, l);
      emit(win, dkb, p.dk.sn, kb, p.k.sn, kn, GLOBAL);      // dk (+)= a dLK W - 2 b (sum_c dLK) k
      zero<4>(t);
      mm<E, 64, true, 64, true, 4>(t, TP, TS, 2, l);   // PK dS : A-op PKT [k=c][m=j], B = DST [n=d][k=c]
      strip_to_rows<E>(TO, t, l);
      emit(win, dvb, p.dv.sn, nullptr, 0, nullptr, GLOBAL); // dv (+)= PK dS
    }
  }
  if (!GLOBAL) {
    gsave_strip<4>(p.p_dsall + ((size_t)bh * npb + np) * M * 64, gs, 64, M, 64, l);
    if (l.li == 0)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r0 + r < M) p.p_dzall[((size_t)bh * npb + np) * M + r0 + r] = gz[r];
  }
}

size_t sb_lds() { return (size_t)5 * TB + (size_t)(64 * 3 + 256 + 64) * sizeof(float); }

int sb_fwd_dispatch(const SbP& p0, int dtype, hipStream_t st) {
  SbP p = p0;
  p.prof = nullptr;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "sb_fwd", 0);
#endif
  const size_t lds = sb_lds();
  const dim3 grid((unsigned)(p.B * p.H * p.nwin)), block(256);
  if (dtype == EA_BF16) hipLaunchKernelGGL(sb_fwd_kernel<BF16>, grid, block, lds, st, p);
  else if (dtype == EA_F16) hipLaunchKernelGGL(sb_fwd_kernel<F16>, grid, block, lds, st, p);
  else return EA_E_BADARG;
  return (int)hipGetLastError();
}

size_t sb_bwd_lds(bool global) { return (size_t)(global ? 6 : 9) * TB + (size_t)(64 * 8 + 512) * sizeof(float); }

// which: 0 window pass, 1 global pass
int sb_bwd_dispatch(int which, const SbP& p0, int dtype, hipStream_t st) {
  SbP p = p0;
  p.prof = nullptr;
  const size_t lds = sb_bwd_lds(which != 0);
  const dim3 grid((unsigned)(p.B * p.H * (p.nwin / p.wpb))), block(256);
#define EA_SB(E, G)                                                                                         \
  do {                                                                                                      \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sb_bwd_kernel<E, G>),                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
    if (e != hipSuccess) return (int)e;                                                                     \
    hipLaunchKernelGGL((sb_bwd_kernel<E, G>), grid, block, lds, st, p);                                     \
  } while (0)
  if (dtype == EA_BF16) { if (which) EA_SB(BF16, true); else EA_SB(BF16, false); }
  else if (dtype == EA_F16) { if (which) EA_SB(F16, true); else EA_SB(F16, false); }
  else return EA_E_BADARG;
#undef EA_SB
  return (int)hipGetLastError();
}

}  // namespace ea
