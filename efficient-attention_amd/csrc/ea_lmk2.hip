// ea_lmk2.hip -- LARA's landmark pipeline (lara.py:145-198,214-238; eva.py:178-190 in `eva` mode), second
// generation: same mathematics and C ABI as ea_lara_landmark.hip, re-laid-out so that TWO or more
// workgroups fit on a CU.
//
// Round 1 kept nine fp32 [64][65] matrices of a (b,h) in LDS (150 KB, one 16-wave workgroup per CU:
// B*h = 384 (b,h) pairs ran as two rounds on 256 CUs, the second half empty) and fetched MFMA operands
// element by element with a float -> half conversion per fetch.  Here
//   * a workgroup is 4 waves; wave w owns the 16-row strip w of EVERY matrix product and keeps it in
//     registers in the MFMA result layout (row 4g+r, column 16 ct + li), so the row-wise steps --
//     LayerNorm, softmax, their backward, squared norms -- never leave the wave (in-lane over the
//     column tiles + a 16-lane butterfly), and everything the reference evaluates in fp32 stays fp32;
//   * LDS only holds MFMA operands: fp16 [*][64] tiles (8 KB) with the XOR swizzle of every other
//     kernel here.  A strip is stored TRANSPOSED (four consecutive rows of one column = one 8-byte
//     store); both operand orientations of a stored matrix are then available -- index along the rows
//     by two 8-byte reads, index along the columns by two ds_read_b64_tr_b16 -- in the same k-slot
//     order, so no product needs a second copy of an operand;
//   * gradient-side operands carry a per-matrix power-of-two scale taken from the block-wide maximum
//     (loss scaling cannot push them out of fp16 range), exactly as before.
// Peak LDS: 40 KB + 33 KB fp32 exchange (forward), 73 KB (backward): 2 workgroups per CU, 4 forward /
// 9 backward barrier phases instead of 9 / 16.
#include "ea_strip.h"
#include "ea_lara_lmk.h"

namespace ea {

namespace lmk2 {

using namespace strip;
typedef F16 H;

// fp32 global [rows][D] -> fp16 row-major tile [64][D] (zero beyond rows), all 256 threads; in two steps so
// that the loads of several matrices are in flight together
template <int D> struct StageRegs { float4 v[(64 * D / 8 + 255) / 256][2]; };
template <int D> EA_DEV void issue_rows(StageRegs<D>& b, const float* src, int rows, int tid) {
  constexpr int CPR = D / 8, NI = (64 * CPR + 255) / 256;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + i * 256;
    const int row = min(idx / CPR, rows - 1), c = idx % CPR;
    b.v[i][0] = *reinterpret_cast<const float4*>(src + (size_t)row * D + c * 8);
    b.v[i][1] = *reinterpret_cast<const float4*>(src + (size_t)row * D + c * 8 + 4);
  }
}
template <int D> EA_DEV void commit_rows(char* tile, const StageRegs<D>& b, int rows, int tid) {
  constexpr int CPR = D / 8, NI = (64 * CPR + 255) / 256;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / CPR, c = idx % CPR;
    if (idx >= 64 * CPR) continue;
    const bool ok = row < rows;
    const float4 lo = b.v[i][0], hi = b.v[i][1];
    const float f[8] = {ok ? lo.x : 0.f, ok ? lo.y : 0.f, ok ? lo.z : 0.f, ok ? lo.w : 0.f,
                        ok ? hi.x : 0.f, ok ? hi.y : 0.f, ok ? hi.z : 0.f, ok ? hi.w : 0.f};
    sts16(tile + lds_off<D>(row, c), pack8<H>(f));
  }
}

// LayerNorm of a strip over its D columns (in place -> normalised rows), returns 1/std per r
template <int NT> EA_DEV void layer_norm(f32x4* s, float* rstd, int D, const Lane& l) {
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += s[ct][r];
  float mean[4], var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) mean[r] = row16_sum(sum[r]) / D;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float c0 = s[ct][r] - mean[r]; var[r] += c0 * c0; }
#pragma unroll
  for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(row16_sum(var[r]) / D + 1e-5f);
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[ct][r] = (s[ct][r] - mean[r]) * rstd[r];
}

template <int D, bool BWD>
__global__ __launch_bounds__(256, 2) void lmk2_kernel(const LmkP p) {
  constexpr int NT = D / 16;                 // column tiles of an [L][D] matrix
  constexpr int KD = D / 32;                 // 32-deep steps of a contraction over the channels
  constexpr int TB = 64 * 128;               // bytes of a transposed-stored tile [<= 64][64]
  constexpr int RB = 64 * D * 2;             // bytes of a row-major [64][D] tile
  extern __shared__ __attribute__((aligned(16))) char sm[];
  // tiles (see the phase comments for who lives where)
  char* T0 = sm;                  // fwd: PQ -> K0T           | bwd: MUT -> DHT (q)
  char* T1 = T0 + TB;             // fwd: PK -> AST           | bwd: OMT -> DGT
  char* T2 = T1 + TB;             // fwd: WQ -> MUT           | bwd: K0T
  char* T3 = T2 + TB;             // fwd: WK -> OMT           | bwd: AST -> DHT (k)
  char* T4 = T3 + TB;             //                          | bwd: DMT -> DKBT
  char* T5 = T4 + TB;             // fwd: fp32 exchange (mu, qbar rows; 2 x [64][D] floats) | bwd: WQ, WK, PQ, PK
  float* vec = reinterpret_cast<float*>(T5 + (BWD ? 4 * RB : 2 * 64 * D * 4));
  float* musq = vec;              // [64]
  float* cpart = vec + 64;        // [4][64] column-sum partials
  float* cpart2 = cpart + 256;    // [4][64]
  float* gmx = cpart2 + 256;      // [4] per-wave max |x|
  float* pv = gmx + 8;            // gq, cq, gk, ck, bq, bk (6 x D)

  const int tid = threadIdx.x, lane = tid & 63;
  Lane l;
  l.w = tid >> 6; l.g = lane >> 4; l.li = lane & 15;
  const int bh = blockIdx.x;
  const int L = p.L, C = p.C;
  const float s = p.scale;
  const size_t oL = (size_t)bh * L * D, oC = (size_t)bh * C * D;
  const int nrep = C / L;
  const int r0 = 16 * l.w + 4 * l.g;         // first of this lane's four rows
  float* sv = p.saved ? p.saved + (size_t)bh * lara_lmk_saved_per_bh(L, D) : nullptr;
  float* sv_xq = sv, *sv_xk = sv ? sv + L * D : nullptr, *sv_mu = sv ? sv + 2 * L * D : nullptr;
  float* sv_a = sv ? sv + 3 * L * D : nullptr, *sv_rstd = sv ? sv + 3 * L * D + L * 64 : nullptr;

  if (p.has_mlp)
    for (int i = tid; i < D; i += 256) {
      pv[i] = p.gq[i]; pv[D + i] = p.cq[i]; pv[2 * D + i] = p.gk[i]; pv[3 * D + i] = p.ck[i];
      pv[4 * D + i] = p.bq[i]; pv[5 * D + i] = p.bk[i];
    }
  __syncthreads();
  // noise of the sample rows c of this strip (omega_c = mu[c mod L] +- eps): antithetic rows c >= L re-use
  // the first L rows with the opposite sign
  auto noise_strip = [&](f32x4* n, int nrows) {
    if (!p.noise) { zero<NT>(n); return; }
    int ri[4]; bool ok[4]; float sg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = min(r0 + r, nrows - 1);
      ok[r] = r0 + r < nrows;
      ri[r] = p.dup == 1 ? c % L : c;
      sg[r] = (p.dup == 1 && c >= L) ? -1.f : 1.f;
    }
    gather_strip<NT>(n, p.dup == 1 ? p.noise + (size_t)bh * L * D : p.noise + oC, D, p.dup == 1 ? L : nrows, ri, ok, D, l);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) n[ct][r] *= sg[r];
  };
  f32x4 xq[NT], xk[NT], qb[NT], k0[NT], mu[NT], om[NT];
  float rsq[4] = {0.f, 0.f, 0.f, 0.f}, rsk[4] = {0.f, 0.f, 0.f, 0.f};

  if (!BWD) {
    // =========================================== forward ===========================================
    // F0: operands of the two Linear layers
    f32x4 nz[NT];
    noise_strip(nz, p.eva ? L : C);
    if (p.has_mlp) {
      StageRegs<D> b0, b1, b2, b3;
      issue_rows<D>(b0, p.pq + oL, L, tid);
      issue_rows<D>(b1, p.pk + oL, L, tid);
      issue_rows<D>(b2, p.Wq, D, tid);
      issue_rows<D>(b3, p.Wk, D, tid);
      commit_rows<D>(T0, b0, L, tid);
      commit_rows<D>(T1, b1, L, tid);
      commit_rows<D>(T2, b2, D, tid);
      commit_rows<D>(T3, b3, D, tid);
    }
    __syncthreads();
    // F1: H = P W^T + b, LayerNorm, affine  (strip w of both sides)
    if (p.has_mlp) {
      zero<NT>(xq); zero<NT>(xk);
      mm<H, D, false, D, true, NT>(xq, T0, T2, KD, l);
      mm<H, D, false, D, true, NT>(xk, T1, T3, KD, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { xq[ct][r] += pv[4 * D + 16 * ct + l.li]; xk[ct][r] += pv[5 * D + 16 * ct + l.li]; }
      layer_norm<NT>(xq, rsq, D, l);
      layer_norm<NT>(xk, rsk, D, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * ct + l.li;
          qb[ct][r] = pv[o] * xq[ct][r] + pv[D + o];
          k0[ct][r] = pv[2 * D + o] * xk[ct][r] + pv[3 * D + o];
        }
      if (sv) {
        gsave_strip<NT>(sv_xq, xq, D, L, D, l);
        gsave_strip<NT>(sv_xk, xk, D, L, D, l);
        if (l.li == 0)
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r0 + r < 64) { sv_rstd[r0 + r] = rsq[r]; sv_rstd[64 + r0 + r] = rsk[r]; }
      }
    } else {
      load_strip<NT>(qb, p.pq + oL, D, L, D, l);
      load_strip<NT>(k0, p.pk + oL, D, L, D, l);
    }
    if (p.eva) {
      // EVA (eva.py:178-190): rf_k_bar = k0, omega = (q_bar + k0) / 2 + eps
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          om[ct][r] = 0.5f * (qb[ct][r] + k0[ct][r]) + nz[ct][r];
      gsave_strip<NT>(p.qbar_rows + oC, k0, D, L, D, l);
      gsave_strip<NT>(p.omega + oC, om, D, L, D, l);
      return;
    }
    __syncthreads();                                 // every wave is done with PQ / PK / WQ / WK
    f32x4 kb[NT];
    if (p.mixed) {
      store_t<H, NT>(T0, k0, 1.f, L, D, l);             // K0T [o][l]
      __syncthreads();
      // F2: G = s k0 k0^T, A = softmax over the L columns, k_bar = A k0
      f32x4 a[4];
      zero<4>(a);
      mm<H, 64, true, 64, false, 4>(a, T0, T0, KD, l);
      float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, den[4] = {0.f, 0.f, 0.f, 0.f};
      float cb[4];                                     // '-vmixed': log(|v_bar_l'| + 1e-4) on column l' (lara.py:171-172)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) cb[ct] = (p.colbias && 16 * ct + l.li < L) ? p.colbias[(size_t)bh * L + 16 * ct + l.li] : 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[ct][r] = (16 * ct + l.li < L) ? a[ct][r] * s + cb[ct] : -INFINITY;
          mx[r] = fmaxf(mx[r], a[ct][r]);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = row16_max(mx[r]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[ct][r] = __expf(a[ct][r] - mx[r]); den[r] += a[ct][r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) den[r] = 1.f / row16_sum(den[r]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[ct][r] *= den[r];
      if (sv) gsave_strip<4, true>(sv_a, a, 64, L, L, l);
      store_t<H, 4>(T1, a, 1.f, L, L, l);               // AST [l'][l]: read back by this wave only (its own columns)
      zero<NT>(kb);
      mm<H, 64, true, 64, true, NT>(kb, T1, T0, 2, l);
    } else {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) kb[ct] = k0[ct];
    }
    // F3: mu, |mu|^2; the fp32 rows other strips need (dup: omega_c and q_bar rows of c >= L)
    float m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { mu[ct][r] = (r0 + r < L) ? qb[ct][r] + kb[ct][r] : 0.f; m2[r] += mu[ct][r] * mu[ct][r]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { m2[r] = row16_sum(m2[r]); if (l.li == 0 && r0 + r < 64) musq[r0 + r] = m2[r]; }
    if (sv) gsave_strip<NT>(sv_mu, mu, D, L, D, l);
    float* XMU = reinterpret_cast<float*>(T5);        // [64][D] fp32
    float* XQB = XMU + 64 * D;
    save_strip<NT>(XMU, mu, D, 64, D, l);
    save_strip<NT>(XQB, qb, D, 64, D, l);
    store_t<H, NT>(T2, mu, 1.f, L, D, l);               // MUT [o][l]  (WQ is dead: barrier above)
    __syncthreads();
    // F4: omega rows of this strip (sample rows c), outputs
    f32x4 qr[NT];
    {
      // (round 5: reads unconditional from clamped rows + selects -- written as `if (c < C) read` every element became an
      //  exec-mask branch with an LDS round trip inside it, and c % L was redone per element)
      int rowl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rowl[r] = (min(r0 + r, C - 1) % L) * D;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = r0 + r, col = 16 * ct + l.li;
          const float mv = XMU[rowl[r] + col], qv = XQB[rowl[r] + col];
          const float m = c < C ? mv : 0.f, q = c < C ? qv : 0.f;
          om[ct][r] = c < C ? m + nz[ct][r] : 0.f;
          qr[ct][r] = p.mis == 0 ? q : m;
        }
    }
    gsave_strip<NT>(p.omega + oC, om, D, C, D, l);
    if (p.mis != 2) gsave_strip<NT>(p.qbar_rows + oC, qr, D, C, D, l);
    store_t<H, NT>(T3, om, 1.f, C, D, l);               // OMT [o][c]
    __syncthreads();
    // F5: M = s omega mu^T - s |mu_l|^2 / 2 ; proposal densities per sample row
    f32x4 M[4];
    zero<4>(M);
    mm<H, 64, true, 64, false, 4>(M, T3, T2, KD, l);
    float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, d0[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f};
    int cml[4];                                          // c mod L of this lane's four sample rows, once
#pragma unroll
    for (int r = 0; r < 4; ++r) cml[r] = min(r0 + r, C - 1) % L;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int col = 16 * ct + l.li;
      const float hm = 0.5f * s * musq[col];             // (musq holds 64 entries: the read needs no guard)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        M[ct][r] = (col < L) ? s * M[ct][r] - hm : -INFINITY;
        mx[r] = fmaxf(mx[r], M[ct][r]);
        d0[r] = (r0 + r < C && col == cml[r]) ? M[ct][r] : d0[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { mx[r] = row16_max(mx[r]); d0[r] = row16_sum(d0[r]); }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) den[r] += __expf(M[ct][r] - mx[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float dn = row16_sum(den[r]);
      if (p.mis == 0) dn *= (float)nrep;
      const float lse = mx[r] + __logf(dn);
      const int c = r0 + r;
      if (l.li == 0 && c < C) {
        p.lp[(size_t)bh * C + c] = p.mis == 0 ? d0[r] : lse;
        if (p.mis == 0) p.bhv[(size_t)bh * C + c] = __expf(d0[r] - lse);
      }
    }
    return;
  }

  // =========================================== backward ===========================================
  EA_STAMP(p, 0);
  EA_BLK(p, 0);
  // B0: EVERY global load of the backward is issued here, unconditionally, so the workgroup pays one
  // memory round trip: the operand tiles of the tail (W, P), the forward's saved strips, the incoming
  // gradients.
  char* WQt = T5, *WKt = T5 + RB, *PQt = T5 + 2 * RB, *PKt = T5 + 3 * RB;
  StageRegs<D> sb0, sb1, sb2, sb3;
  if (p.has_mlp) {
    issue_rows<D>(sb0, p.Wq, D, tid);
    issue_rows<D>(sb1, p.Wk, D, tid);
    issue_rows<D>(sb2, p.pq + oL, L, tid);
    issue_rows<D>(sb3, p.pk + oL, L, tid);
    load_strip<NT>(xq, sv_xq, D, L, D, l);
    load_strip<NT>(xk, sv_xk, D, L, D, l);
#pragma unroll
    for (int r = 0; r < 4; ++r) { rsq[r] = sv_rstd[min(r0 + r, 63)]; rsk[r] = sv_rstd[64 + min(r0 + r, 63)]; }
  }
  f32x4 dqb[NT], dk0[NT];                              // gradients of q_bar and k0 rows of this strip
  f32x4 gin[NT], gqr[NT];                              // incoming: d omega rows, d (q_bar | mu) rows of this strip
  int rl[4]; bool rok[4];                              // landmark row of sample row c = r0 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) { rok[r] = r0 + r < C; rl[r] = min(r0 + r, C - 1) % L; }
  load_strip<NT>(gin, p.d_omega + oC, D, p.eva ? L : C, D, l);
  if (p.dom_parts) {                                   // (uniform) a + sum_s parts[s], then the scale: ea_slice_sum's order
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      f32x4 part[NT];
      load_strip<NT>(part, s_ < p.dom_S ? p.dom_parts + ((size_t)bh * p.dom_S + s_) * C * D : nullptr, D, C, D, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) gin[ct] = gin[ct] + part[ct];
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) gin[ct] = gin[ct] * p.dom_scale;
  }
  if (p.eva) {
    // d rf_q_bar = d omega / 2, d rf_k_bar = d omega / 2 + d (rf_k_bar output)
    load_strip<NT>(gqr, p.d_qbar_rows ? p.d_qbar_rows + oC : nullptr, D, L, D, l);
    if (p.dqr_S > 1) {                                 // (uniform) slice partials of the window backward, added in slice order
#pragma unroll
      for (int s_ = 1; s_ < 4; ++s_) {
        f32x4 part[NT];
        load_strip<NT>(part, s_ < p.dqr_S ? p.d_qbar_rows + (size_t)s_ * p.dqr_stride + oC : nullptr, D, L, D, l);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) gqr[ct] = gqr[ct] + part[ct];
      }
    }
    if (p.has_mlp) {
      commit_rows<D>(WQt, sb0, D, tid); commit_rows<D>(WKt, sb1, D, tid);
      commit_rows<D>(PQt, sb2, L, tid); commit_rows<D>(PKt, sb3, L, tid);
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { dqb[ct][r] = 0.5f * gin[ct][r]; dk0[ct][r] = 0.5f * gin[ct][r] + gqr[ct][r]; }
    __syncthreads();
  } else {
    const bool have_mu = p.has_mlp || p.mixed;
    f32x4 nz[NT], muc[NT], dqs[NT], asm_[4];
    float pre_dlp[4], pre_dbh[4];
    noise_strip(nz, C);
    load_strip<NT>(gqr, (p.mis == 1 && p.d_qbar_rows) ? p.d_qbar_rows + oC : nullptr, D, C, D, l);
    zero<NT>(dqs);
    if (p.mis == 0 && p.d_qbar_rows) {                 // sum_k d q_bar rows [l + k L] of this strip's landmark rows
      for (int k = 0; k < nrep; ++k) {
        int ri[4]; bool ok[4];
        f32x4 t[NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = r0 + r < L; ri[r] = min(r0 + r, L - 1) + k * L; }
        gather_strip<NT>(t, p.d_qbar_rows + oC, D, C, ri, ok, D, l);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) dqs[ct] = dqs[ct] + t[ct];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)bh * C + min(r0 + r, C - 1);
      pre_dlp[r] = p.d_lp[o];
      pre_dbh[r] = (p.mis == 0 && p.d_bhv) ? p.d_bhv[o] : 0.f;
    }
    if (have_mu) {
      load_strip<NT>(mu, sv_mu, D, L, D, l);
      gather_strip<NT>(muc, sv_mu, D, L, rl, rok, D, l);
    } else {
      // no saved mu: mu = q_bar + k_bar with both given
      f32x4 a[NT], b[NT];
      load_strip<NT>(a, p.pq + oL, D, L, D, l);
      load_strip<NT>(b, p.pk + oL, D, L, D, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) mu[ct] = a[ct] + b[ct];
      gather_strip<NT>(a, p.pq + oL, D, L, rl, rok, D, l);
      gather_strip<NT>(b, p.pk + oL, D, L, rl, rok, D, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) muc[ct] = a[ct] + b[ct];
    }
    if (p.mixed) load_strip<4, true>(asm_, sv_a, 64, L, L, l);
    if (p.has_mlp) {
      commit_rows<D>(WQt, sb0, D, tid); commit_rows<D>(WKt, sb1, D, tid);
      commit_rows<D>(PQt, sb2, L, tid); commit_rows<D>(PKt, sb3, L, tid);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) k0[ct][r] = pv[2 * D + 16 * ct + l.li] * xk[ct][r] + pv[3 * D + 16 * ct + l.li];
    } else if (p.mixed) {
      load_strip<NT>(k0, p.pk + oL, D, L, D, l);
    }
    float m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) m2[r] += mu[ct][r] * mu[ct][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m2[r] = row16_sum(m2[r]); if (l.li == 0 && r0 + r < 64) musq[r0 + r] = m2[r]; }
    // omega rows of this strip: mu[c mod L] +- eps
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) om[ct] = muc[ct] + nz[ct];
    store_t<H, NT>(T0, mu, 1.f, L, D, l);               // MUT
    store_t<H, NT>(T1, om, 1.f, C, D, l);               // OMT
    if (p.mixed) {
      store_t<H, NT>(T2, k0, 1.f, L, D, l);             // K0T
      store_t<H, 4>(T3, asm_, 1.f, L, L, l);            // AST [l'][l]
    }
    __syncthreads();
    EA_STAMP(p, 1);
    // B1: M (recomputed), proposal densities, dM in registers
    f32x4 dM[4];
    {
      f32x4 M[4];
      zero<4>(M);
      mm<H, 64, true, 64, false, 4>(M, T1, T0, KD, l);
      float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, d0[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f};
      int cml[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cml[r] = min(r0 + r, C - 1) % L;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int col = 16 * ct + l.li;
        const float hm = 0.5f * s * musq[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          M[ct][r] = (col < L) ? s * M[ct][r] - hm : -INFINITY;
          mx[r] = fmaxf(mx[r], M[ct][r]);
          d0[r] = (r0 + r < C && col == cml[r]) ? M[ct][r] : d0[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { mx[r] = row16_max(mx[r]); d0[r] = row16_sum(d0[r]); }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) den[r] += __expf(M[ct][r] - mx[r]);
      const float mult = p.mis == 0 ? (float)nrep : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float dn = row16_sum(den[r]);
        if (p.mis == 0) dn *= (float)nrep;
        const float lse = mx[r] + __logf(dn);
        const int c = r0 + r;
        float dlp = 0.f, dlse = 0.f;
        if (p.mis == 0) {
          const float dbh = pre_dbh[r] * __expf(d0[r] - lse);
          dlp = pre_dlp[r] + dbh;
          dlse = -dbh;
        } else {
          dlse = pre_dlp[r];
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int col = 16 * ct + l.li;
          dM[ct][r] = (c < C && col < L) ? dlse * __expf(M[ct][r] - lse) * mult + ((p.mis == 0 && col == cml[r]) ? dlp : 0.f) : 0.f;
        }
      }
    }
    {
      const float gm = strip_absmax<4>(dM, C, L, l);
      if (lane == 0) gmx[l.w] = gm;
    }
    EA_STAMP(p, 2);
    colsum_part<4>(cpart, dM, C, l);                  // column sums of dM
    __syncthreads();
    const float sdm = pow2_scale(gmx);
    store_t<H, 4>(T4, dM, sdm, C, L, l);                 // DMT [l][c]
    __syncthreads();
    EA_STAMP(p, 3);
    // B2: dMU = s dM^T OM - s colsum(dM) mu ;  dOM = d omega (+ d mu rows) + s dM MU
    f32x4 dmu[NT], dom[NT];
    zero<NT>(dmu); zero<NT>(dom);
    mm<H, 64, false, 64, true, NT>(dmu, T4, T1, 2, l);   // A = dM^T rows l (DMT row-major), B = OM (OMT = [n][k])
    mm<H, 64, true, 64, true, NT>(dom, T4, T0, 2, l);    // A = dM rows c (DMT = [k][m]),   B = MU (MUT = [n][k])
    {
      const float al = s / sdm;
      float dmc[4];                                      // column sums of dM for this lane's four rows, once (cpart: 4 x 64)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + r;
        const float v = cpart[row] + cpart[64 + row] + cpart[128 + row] + cpart[192 + row];
        dmc[r] = row < L ? v : 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dmu[ct][r] = al * dmu[ct][r] - s * dmc[r] * mu[ct][r];
          dom[ct][r] = gin[ct][r] + gqr[ct][r] + al * dom[ct][r];
        }
    }
    EA_STAMP(p, 4);
    // B3: fold the sample rows onto the landmarks: d mu[l] += sum_k dOM[l + k L]  (fp32 exchange through LDS)
    __syncthreads();                                  // DMT / cpart readers done; T4.. reused below
    float* XOM = reinterpret_cast<float*>(T0);        // [64][D] fp32 over T0..T1 (MUT, OMT are dead)
    if (nrep > 1) {
      save_strip<NT>(XOM, dom, D, 64, D, l);
      __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + r, col = 16 * ct + l.li;
        float d = 0.f;
        if (row < L) {
          d = dmu[ct][r] + dom[ct][r];
          for (int k = 1; k < nrep; ++k) d += XOM[(row + k * L) * D + col];
        }
        dk0[ct][r] = d;                               // = d k_bar (before the mixing backward)
        dqb[ct][r] = d + dqs[ct][r];
      }
    EA_STAMP(p, 5);
    // B4..B6: mixing backward
    if (p.mixed) {
      {
        const float gm = strip_absmax<NT>(dk0, L, D, l);
        if (lane == 0) gmx[l.w] = gm;
      }
      __syncthreads();
      const float skb = pow2_scale(gmx);
      store_t<H, NT>(T4, dk0, skb, L, D, l);             // DKBT [o][l]  (DMT is dead: barrier at B3)
      __syncthreads();
      f32x4 dkk[NT], dA[4];
      zero<NT>(dkk); zero<4>(dA);
      mm<H, 64, false, 64, true, NT>(dkk, T3, T4, 2, l); // dK0 = A^T dKb : A-op = AST rows l' (row-major), B = dKb (DKBT = [n][k])
      mm<H, 64, true, 64, false, 4>(dA, T4, T2, KD, l);  // dA = dKb K0^T : A-op = dKb rows l (DKBT = [k][m]), B[k=o][n=l'] = K0T
      float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dA[ct][r] *= 1.f / skb; rs[r] += asm_[ct][r] * dA[ct][r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] = row16_sum(rs[r]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) dA[ct][r] = asm_[ct][r] * (dA[ct][r] - rs[r]);      // dG
      if (p.d_colbias) colsum_part<4>(cpart, dA, L, l);  // d colbias = column sums of dG (cpart: free since the barrier at B3)
      {
        const float gm = strip_absmax<4>(dA, L, L, l);
        __syncthreads();                              // gmx readers (skb) are done
        if (lane == 0) gmx[l.w] = gm;
      }
      __syncthreads();
      if (p.d_colbias && tid < L) p.d_colbias[(size_t)bh * L + tid] = cpart[tid] + cpart[64 + tid] + cpart[128 + tid] + cpart[192 + tid];
      const float sdg = pow2_scale(gmx);
      store_t<H, 4>(T1, dA, sdg, L, L, l);               // DGT [l'][l]  (OMT is dead)
      __syncthreads();
      f32x4 t1[NT];
      zero<NT>(t1);
      mm<H, 64, true, 64, true, NT>(t1, T1, T2, 2, l);   // dG K0  : A-op = dG rows l (DGT = [k][m]), B = K0 (K0T = [n][k])
      mm<H, 64, false, 64, true, NT>(t1, T1, T2, 2, l);  // dG^T K0: A-op = dG^T rows l (DGT row-major)
      const float a2 = s / sdg, a1 = 1.f / skb;
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) dk0[ct][r] = (r0 + r < L) ? a1 * dkk[ct][r] + a2 * t1[ct][r] : 0.f;
    }
  }
  EA_STAMP(p, 6);
  // ---- tail: LayerNorm + Linear backward of both sides (dY = dqb / dk0 in registers) ----
  if (!p.has_mlp) {
    gsave_strip<NT>(p.dpq + oL, dqb, D, L, D, l);
    gsave_strip<NT>(p.dpk + oL, dk0, D, L, D, l);
    return;
  }
  __syncthreads();                                    // cpart / tiles of the mixing backward / the exchange buffer are free
  f32x4 dhq[NT], dhk[NT];
  {
    // d gamma = sum_l dY xhat, d beta = sum_l dY
    f32x4 t[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) t[ct] = dqb[ct] * xq[ct];
    colsum_part<NT>(cpart, t, L, l);
    colsum_part<NT>(cpart2, dqb, L, l);
    __syncthreads();
    if (tid < D) {
      float* dvec = p.dvec_part + ((size_t)bh * 2 + 0) * 3 * D;
      dvec[D + tid] = cpart[tid] + cpart[64 + tid] + cpart[128 + tid] + cpart[192 + tid];
      dvec[2 * D + tid] = cpart2[tid] + cpart2[64 + tid] + cpart2[128 + tid] + cpart2[192 + tid];
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) t[ct] = dk0[ct] * xk[ct];
    colsum_part<NT>(cpart, t, L, l);
    colsum_part<NT>(cpart2, dk0, L, l);
    __syncthreads();
    if (tid < D) {
      float* dvec = p.dvec_part + ((size_t)bh * 2 + 1) * 3 * D;
      dvec[D + tid] = cpart[tid] + cpart[64 + tid] + cpart[128 + tid] + cpart[192 + tid];
      dvec[2 * D + tid] = cpart2[tid] + cpart2[64 + tid] + cpart2[128 + tid] + cpart2[192 + tid];
    }
    // dH = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = dY gamma
    auto ln_bwd = [&](f32x4* dh, const f32x4* dy, const f32x4* xh, const float* gam, const float* rs) {
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dh[ct][r] = dy[ct][r] * gam[16 * ct + l.li];
          s1[r] += dh[ct][r];
          s2[r] += dh[ct][r] * xh[ct][r];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = row16_sum(s1[r]) / D; s2[r] = row16_sum(s2[r]) / D; }
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[ct][r] = (r0 + r < L) ? rs[r] * (dh[ct][r] - s1[r] - xh[ct][r] * s2[r]) : 0.f;
    };
    ln_bwd(dhq, dqb, xq, pv, rsq);
    ln_bwd(dhk, dk0, xk, pv + 2 * D, rsk);
  }
  EA_STAMP(p, 7);
  __syncthreads();                                    // cpart readers done
  colsum_part<NT>(cpart, dhq, L, l);                  // d bias of the Linear
  colsum_part<NT>(cpart2, dhk, L, l);
  {
    const float gq_ = strip_absmax<NT>(dhq, L, D, l), gk_ = strip_absmax<NT>(dhk, L, D, l);
    if (lane == 0) { gmx[l.w] = gq_; gmx[4 + l.w] = gk_; }
  }
  __syncthreads();
  if (tid < D) {
    p.dvec_part[((size_t)bh * 2 + 0) * 3 * D + tid] = cpart[tid] + cpart[64 + tid] + cpart[128 + tid] + cpart[192 + tid];
    p.dvec_part[((size_t)bh * 2 + 1) * 3 * D + tid] = cpart2[tid] + cpart2[64 + tid] + cpart2[128 + tid] + cpart2[192 + tid];
  }
  const float shq = pow2_scale(gmx), shk = pow2_scale(gmx + 4);
  EA_STAMP(p, 8);
  store_t<H, NT>(T0, dhq, shq, L, D, l);                 // DHT (q) [o][l]
  store_t<H, NT>(T3, dhk, shk, L, D, l);                 // DHT (k)
  __syncthreads();
  {
    // dP = dH W (rows l) -> global;  dW = dH^T P (rows o) -> per-(b,h) partial
    f32x4 dp[NT];
    zero<NT>(dp);
    mm<H, 64, true, D, false, NT>(dp, T0, WQt, KD, l);   // A-op = dH rows l (DHT = [k][m]); B[k=o][n=i] = W row-major
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) dp[ct] = dp[ct] * (1.f / shq);
    gsave_strip<NT>(p.dpq + oL, dp, D, L, D, l);
    zero<NT>(dp);
    mm<H, 64, true, D, false, NT>(dp, T3, WKt, KD, l);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) dp[ct] = dp[ct] * (1.f / shk);
    gsave_strip<NT>(p.dpk + oL, dp, D, L, D, l);
    if (16 * l.w < D) {                               // rows o of dW: D / 16 strips
      f32x4 dw[NT];
      zero<NT>(dw);
      mm<H, 64, false, D, false, NT>(dw, T0, PQt, 2, l); // A-op = dH^T rows o (DHT row-major); B[k=l][n=i] = P row-major
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) dw[ct] = dw[ct] * (1.f / shq);
      gsave_strip<NT>(p.dW_part + ((size_t)bh * 2 + 0) * D * D, dw, D, D, D, l);
      zero<NT>(dw);
      mm<H, 64, false, D, false, NT>(dw, T3, PKt, 2, l);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) dw[ct] = dw[ct] * (1.f / shk);
      gsave_strip<NT>(p.dW_part + ((size_t)bh * 2 + 1) * D * D, dw, D, D, D, l);
    }
  }
  EA_STAMP(p, 9);
  EA_BLK(p, 1);
}

}  // namespace lmk2

size_t lmk2_lds(int D, bool bwd) {
  const size_t tiles = (size_t)5 * 64 * 128;
  const size_t x = bwd ? (size_t)4 * 64 * D * 2 : (size_t)2 * 64 * D * 4;
  return tiles + x + (size_t)(64 + 256 + 256 + 8 + 6 * D) * sizeof(float);
}

template <int D>
static int launch_lmk2(bool bwd, const LmkP& p, hipStream_t st) {
  const size_t lds = lmk2_lds(D, bwd);
  const void* fn = bwd ? reinterpret_cast<const void*>(&lmk2::lmk2_kernel<D, true>)
                       : reinterpret_cast<const void*>(&lmk2::lmk2_kernel<D, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (bwd) hipLaunchKernelGGL((lmk2::lmk2_kernel<D, true>), dim3(p.BH), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((lmk2::lmk2_kernel<D, false>), dim3(p.BH), dim3(256), lds, st, p);
  return (int)hipGetLastError();
}

// the second-generation kernels need the forward's saved intermediates in the backward
// The landmark pipeline has ONE implementation (the round-1 kernels of ea_lara_landmark.hip -- fp32 matrices in LDS, one
// 16-wave workgroup per CU -- were retired in round 3).  A backward of a parametrised / mixed pipeline needs the workspace the
// forward filled (`saved`): recomputing the forward inside the backward was the only thing the old kernels still offered.
int lara_lmk_dispatch(bool bwd, const LmkP& p0, hipStream_t st) {
  LmkP p = p0;
  p.prof = nullptr;
  if (p.L > 64 || p.C > 64 || (p.D != 64 && p.D != 32)) return EA_E_UNSUPPORTED;
  if (bwd && !p.saved && (p.has_mlp || p.mixed)) return EA_E_BADARG;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "lmk2", bwd ? 1 : 0);
#endif
  if (p.D == 64) return launch_lmk2<64>(bwd, p, st);
  return launch_lmk2<32>(bwd, p, st);
}

}  // namespace ea
