// ea_window_fwd.hip -- forward of window attention with control-variate (landmark) columns.
//
// Replaces, for EVA (eva.py:151-153,200-227) and LocalAttention (local_attention.py:155-181),
// the chain: window_partition copies of q/k/v -> einsum QK^T -> +rpe bias -> masked_fill ->
// cat with the landmark logits -> softmax -> split -> two einsums -> window_merge copy.
//
// One 256-thread workgroup (4 waves) owns a run of windows of one (batch, head):
//   * the landmark keys/values (rf_k_bar, beta: [L, D]) are converted to the MFMA element type
//     and parked in LDS once per workgroup, together with the slot-offset tables;
//   * per iteration the local K/V rows of `wpi` windows are gathered straight from the strided
//     q/k/v tensors into LDS (the window partition is address arithmetic, nothing is copied in
//     HBM); all loads of the iteration -- the K/V rows and the waves' Q fragments -- are in
//     flight before the first LDS store; every key row carries (mul, add) so that
//     logit = mul * (s q.k + bias) + add reproduces masked_fill(-5e4) / absent slots with FMAs;
//   * each wave takes 16-query tiles: S^T = K.Q^T tiles are produced 64 keys at a time, run
//     through an online softmax in registers (two 4-lane shuffles per reduction) and fed,
//     register for register, into O^T = V^T.P^T with V^T fragments from ds_read_b64_tr_b16;
//   * O is written with each lane owning D/4 contiguous channels of one query (32 B stores).
// HBM traffic is the algorithmic minimum (q,k,v read once, out written once) plus the landmark
// rows per workgroup, which come from L2.
#include "ea_window.h"

namespace ea {

// CA: causal_eva.py geometry (ea_geom.causal != 0): per-(query, key) visibility limits on top of the
// per-key (mul, add) pairs.
// DR (with CA): attention dropout from an explicit keep mask.
// SG: static geometry (tile counts as template constants: the query / chunk loops unroll into straight-line code; see
// ea_window_bwd.hip) or SGdyn.

template <typename E, int D, bool CA, bool DR, typename SG>
__global__ __launch_bounds__(256, D == 128 ? 2 : 4) void win_fwd_kernel(const WinP p) {
  constexpr int ROWB = D * 2;      // bytes per LDS row
  constexpr int CPR = D / 8;       // 16-byte chunks per row
  constexpr int KS = D / 32;       // k-steps of the score MFMA
  constexpr int DT = D / 16;       // 16-channel tiles of the output
  constexpr int DQ = D / 4;        // channels per lane in the output layout
  constexpr int NB = 2;            // staging slots per thread per batch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WinTiling& t = p.t;
  constexpr bool STATIC = SG::NQT > 0;
  const int nQT = STATIC ? SG::NQT : t.nQT, nLT = STATIC ? SG::NLT : t.nLT, nCT = STATIC ? SG::NCT : t.nCT;
  const int wpi = STATIC ? SG::WPI : t.wpi;
  const int nchunks = STATIC ? (SG::NLT + SG::NCT + 3) / 4 : t.nchunks;
  const int rowsLocal = STATIC ? SG::WPI * SG::NLT * 16 : t.rowsLocal, rowsLm = STATIC ? SG::NCT * 16 : t.rowsLm;
  const int rowsTotal = rowsLocal + rowsLm + 16;
  const int biasLd = STATIC ? SG::NLT * 16 : t.biasLd;
  char* Ks = smem;
  char* Vs = Ks + rowsTotal * ROWB;
  float* kmul = reinterpret_cast<float*>(Vs + rowsTotal * ROWB);
  float* kadd = kmul + rowsTotal;
  int* kd = reinterpret_cast<int*>(kadd + rowsTotal);
  int* qd = kd + nLT * 16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  typename LaneOffSel<D>::type lo;       // round-3 conflict-free tile layout for D = 64 (ea_common.h)
  lo.init(lane);
  const int bh = blockIdx.x / t.nblk, blk = blockIdx.x - bh * t.nblk;
  const int b = bh / p.H, h = bh - b * p.H;
  const char* qb = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* kb = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const char* vb = p.v.p + (b * p.v.sb + h * p.v.sh) * 2;
  char* ob = p.o.p + (b * p.o.sb + h * p.o.sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  const int rowsPerWin = nLT * 16;

  // ---- once per workgroup: slot tables, landmark rows, the all-zero dummy tile ----
  build_slot_tables(kd, qd, t, p.G, p.w, p.e, nQT * 16, tid);
  for (int idx = tid; idx < (rowsLm + 16) * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    u32x4 kw = {0u, 0u, 0u, 0u}, vw = {0u, 0u, 0u, 0u};
    if (row < p.L) {
      const size_t off = ((size_t)bh * p.L + row) * D + c * 8;
      float f[8], f2[8];
      *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p.lk + off);
      *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p.lk + off + 4);
      *reinterpret_cast<float4*>(f2) = *reinterpret_cast<const float4*>(p.lv + off);
      *reinterpret_cast<float4*>(f2 + 4) = *reinterpret_cast<const float4*>(p.lv + off + 4);
      kw = pack8<E>(f);
      vw = pack8<E>(f2);
    }
    sts16(Ks + TileL<D>::off(rowsLocal + row, c), kw);
    sts16(Vs + TileL<D>::off(rowsLocal + row, c), vw);
    if (c == 0) {
      kmul[rowsLocal + row] = row < p.L ? 1.f : 0.f;
      kadd[rowsLocal + row] = row < p.L ? 0.f : -INFINITY;
    }
  }

  const int it_end = min((blk + 1) * t.ipb, t.niter);
  for (int it = blk * t.ipb; it < it_end; ++it) {
    __syncthreads();   // readers of the previous iteration's local rows are done (and tables ready)
    // ---- this wave's first query tile: Q fragments issued before anything is waited for ----
    typename E::x8 qf[KS];
    int qtok0 = -1;
    {
      const int qi = wave;
      if (qi < wpi * nQT) {
        const int wi = qi / nQT, qt = qi - wi * nQT;
        const int win = it * wpi + wi;
        const int qslot = qt * 16 + li;
        if (win < t.nwin && qslot < t.Wq) {
          int oy, ox;
          win_origin(p.G, win, p.w, oy, ox);
          qtok0 = slot_token(p.G, qd[qslot], oy, ox);
        }
      }
      // unconditional loads from a clamped address, zeroed by a select (a predicated load is an exec-mask branch of its
      // own and the loads behind it wait for it)
      const int qtc = qtok0 >= 0 ? qtok0 : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 w4 = ldg16(qb + (qtc * p.q.sn + (g * KS + ks) * 8) * 2);
        qf[ks] = as_x8<E>(qtok0 >= 0 ? w4 : u32x4{0u, 0u, 0u, 0u});
      }
    }
    // ---- gather the local K/V rows of this iteration's windows (batched loads) ----
    for (int base = 0; base < rowsLocal * CPR; base += 256 * NB) {
      u32x4 kr[NB], vr[NB];
      int rowv[NB];
      float mulv[NB], addv[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = base + tid + i * 256;
        const bool in = idx < rowsLocal * CPR;
        const int idc = in ? idx : 0;
        const int row = idc / CPR, c = idc - row * CPR;
        const int wi = wpi == 1 ? 0 : row / rowsPerWin;
        const int slot = row - wi * rowsPerWin;
        const int win = it * wpi + wi;
        const bool live = in && win < t.nwin && slot < t.Wk;    // otherwise the slot does not exist
        int oy, ox;
        win_origin(p.G, min(win, t.nwin - 1), p.w, oy, ox);
        const int tok = slot_token(p.G, kd[slot], oy, ox);
        const bool has = live && tok >= 0;                       // outside / padded: zero k,v, -5e4
        const int tc = has ? tok : 0;
        const u32x4 k4 = ldg16(kb + (tc * p.k.sn + c * 8) * 2);
        const u32x4 v4 = ldg16(vb + (tc * p.v.sn + c * 8) * 2);
        bool keep = has;
        if (mrow) keep = keep && !mrow[tc];
        const u32x4 z = {0u, 0u, 0u, 0u};
        kr[i] = has ? k4 : z;
        vr[i] = has ? v4 : z;
        rowv[i] = in ? row : -1;
        mulv[i] = keep ? 1.f : 0.f;
        addv[i] = keep ? 0.f : (live ? MASK_FILL * LOG2E : -INFINITY);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (rowv[i] >= 0) {
          const int c = (base + tid + i * 256) - rowv[i] * CPR;
          sts16(Ks + TileL<D>::off(rowv[i], c), kr[i]);
          sts16(Vs + TileL<D>::off(rowv[i], c), vr[i]);
          if (c == 0) { kmul[rowv[i]] = mulv[i]; kadd[rowv[i]] = addv[i]; }
        }
      }
    }
    __syncthreads();

    // (run-time geometries: the query phase re-derives its lane coordinates from an opaque copy of the thread index -- one of
    //  the prologue's lane offsets was spilled across the iteration loop: 8 B / lane of scratch until round 6; the copy is
    //  transparent for the static geometries)
    int tid_q = threadIdx.x;
    if constexpr (!STATIC) asm volatile("" : "+v"(tid_q));
    const int lane = tid_q & 63, g = lane >> 4, li = lane & 15;
    const int wave = STATIC ? (tid_q >> 6) : __builtin_amdgcn_readfirstlane(tid_q >> 6);
    typename LaneOffSel<D>::type lo;
    lo.init(lane);
    for (int qi = wave; qi < wpi * nQT; qi += 4) {
      const int wi = qi / nQT, qt = qi - wi * nQT;
      const int win = it * wpi + wi;
      if (win >= t.nwin) continue;                      // wave-uniform
      const int qslot = qt * 16 + li;
      int qtok = qtok0;
      if (qi != wave) {                                 // further tiles of this wave (rare: nQT*wpi > 4)
        qtok = -1;
        if (qslot < t.Wq) {
          int oy, ox;
          win_origin(p.G, win, p.w, oy, ox);
          qtok = slot_token(p.G, qd[qslot], oy, ox);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          u32x4 w4 = {0u, 0u, 0u, 0u};
          if (qtok >= 0) w4 = ldg16(qb + (qtok * p.q.sn + (g * KS + ks) * 8) * 2);
          qf[ks] = as_x8<E>(w4);
        }
      }
      // bias is pre-multiplied by log2(e) by the caller
      const float* brow = p.bias
          ? p.bias + ((size_t)h * t.Wq + (qslot < t.Wq ? qslot : 0)) * biasLd + 4 * g : nullptr;

      QLim ql;
      ql.local = ql.lm = 0x7fffffff;
      if (CA) ql = query_limits(p.causal, qslot, qtok, p.e, p.chunk, mrow, p.lm_base);

      float m = -INFINITY, lsum = 0.f;
      f32x4 o[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

      for (int ch = 0; ch < nchunks; ++ch) {
        int rowbase[4];
        f32x4 s[4];
        float mloc = -INFINITY;
        float4 b4[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {                 // bias loads first: they overlap the MFMAs
          const int tile = ch * 4 + tt;
          b4[tt] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (brow && tile < nLT) b4[tt] = *reinterpret_cast<const float4*>(brow + tile * 16);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int tile = ch * 4 + tt;
          const bool local = tile < nLT;
          rowbase[tt] = local ? (wi * nLT + tile) * 16
                              : (tile < nLT + nCT ? rowsLocal + (tile - nLT) * 16
                                                      : rowsLocal + rowsLm);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const int row = rowbase[tt] + li;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = E::mma(as_x8<E>(lds16(Ks + rowbase[tt] * ROWB + lo.plain[ks])), qf[ks], acc);
          const float4 m4 = *reinterpret_cast<const float4*>(kmul + rowbase[tt] + 4 * g);
          const float4 a4 = *reinterpret_cast<const float4*>(kadd + rowbase[tt] + 4 * g);
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, aa[4] = {a4.x, a4.y, a4.z, a4.w};
          const float bb[4] = {b4[tt].x, b4[tt].y, b4[tt].z, b4[tt].w};
          const int kidx0 = (local ? tile : tile - nLT) * 16 + 4 * g;   // key slot / landmark id of r = 0
          const int lim = local ? ql.local : ql.lm;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = fmaf(mm[r], fmaf(acc[r], p.scale_log2, bb[r]), aa[r]);
            if (CA) x = kidx0 + r > lim ? fminf(x, MASK_FILL * LOG2E) : x;  // absent slots stay -inf
            acc[r] = x;
            mloc = fmaxf(mloc, x);
          }
          s[tt] = acc;
        }
        mloc = quad_max(mloc);
        const float mnew = fmaxf(m, mloc);
        const float msafe = mnew == -INFINITY ? 0.f : mnew;
        const float alpha = fast_exp2(m - msafe);        // m = -inf -> 0
        m = mnew;
        float psum = 0.f;
        uint32_t pw[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float pv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pv[r] = fast_exp2(s[tt][r] - msafe);
            psum += pv[r];
          }
          pw[tt][0] = pack2<E>(pv[0], pv[1]);
          pw[tt][1] = pack2<E>(pv[2], pv[3]);
        }
        if (DR) {
          // dropped entries leave the numerator only; the normaliser keeps every column
          const uint8_t* krow = p.keep + ((size_t)bh * p.G.N + (qtok >= 0 ? qtok : 0)) * p.keep_ld + 4 * g;
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const int tile = ch * 4 + tt;
            const bool real = tile < nLT + nCT;
            const int col = tile < nLT ? tile * 16 : biasLd + (tile - nLT) * 16;
            const uint32_t m4 = real ? *reinterpret_cast<const uint32_t*>(krow + col) : 0u;
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = ((m4 >> (8 * r)) & 0xffu) ? fast_exp2(s[tt][r] - msafe) * p.keep_scale : 0.f;
            pw[tt][0] = pack2<E>(pv[0], pv[1]);
            pw[tt][1] = pack2<E>(pv[2], pv[3]);
          }
        }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          u32x4 pf4;
          pf4[0] = pw[2 * kk][0]; pf4[1] = pw[2 * kk][1];
          pf4[2] = pw[2 * kk + 1][0]; pf4[3] = pw[2 * kk + 1][1];
          const typename E::x8 pf = as_x8<E>(pf4);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const u32x2 lo_ = E::tr4(Vs + rowbase[2 * kk] * ROWB + lo.tr[dt]);
            const u32x2 hi_ = E::tr4(Vs + rowbase[2 * kk + 1] * ROWB + lo.tr[dt]);
            o[dt] = E::mma(as_x8<E>(lo_, hi_), pf, o[dt]);
          }
        }
      }
      // ---- finalize: normalise, store O (lane: query li, channels DQ*g .. DQ*g+DQ-1), lse ----
      const float ltot = quad_sum(lsum);
      const float inv = fast_rcp(ltot);
      if constexpr (TileL<D>::NEWTR) {
        u32x4 o0, o1;
        quad_transpose_pack<E>(o, inv, o0, o1);           // pieces -> the lane's 32 contiguous bytes (all lanes take part)
        if (qtok >= 0) {
          char* dst = ob + (qtok * p.o.sn + DQ * g) * 2;
          stg16(dst, o0);
          stg16(dst + 16, o1);
          if (g == 0) p.lse[((size_t)bh) * p.G.N + qtok] = (m + fast_log2(ltot)) * LN2;
        }
      } else if (qtok >= 0) {
        float f[DQ];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) f[4 * dt + r] = o[dt][r] * inv;
        char* dst = ob + (qtok * p.o.sn + DQ * g) * 2;
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
        if (g == 0) p.lse[((size_t)bh) * p.G.N + qtok] = (m + fast_log2(ltot)) * LN2;
      }
    }
  }
}

size_t window_fwd_lds(const WinTiling& t, int D) {
  return (size_t)t.rowsTotal * D * 2 * 2 + (size_t)t.rowsTotal * 8 + (size_t)(t.nLT * 16 + t.nQT * 16) * 4;
}

template <typename E, int D, bool CA, bool DR, typename SG = SGdyn>
static int launch_fwd_ca(const WinP& p, hipStream_t st) {
  const size_t lds = window_fwd_lds(p.t, D);
  if (lds > 160 * 1024) return EA_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&win_fwd_kernel<E, D, CA, DR, SG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid((unsigned)(p.B * p.H * p.t.nblk));
  hipLaunchKernelGGL((win_fwd_kernel<E, D, CA, DR, SG>), grid, dim3(256), lds, st, p);
  return (int)hipGetLastError();
}
template <typename E, int D>
static int launch_fwd(const WinP& p, hipStream_t st) {
  if (p.keep) return p.causal ? launch_fwd_ca<E, D, true, true>(p, st) : EA_E_UNSUPPORTED;
  if (p.causal) return launch_fwd_ca<E, D, true, false>(p, st);
  if constexpr (D == 64) {
    const WinTiling& t = p.t;
    if (t.nQT == 4 && t.nLT == 4 && t.wpi == 1) {
      if (t.nCT == 4) return launch_fwd_ca<E, D, false, false, SGs<4, 4, 4, 1>>(p, st);
      if (t.nCT == 3) return launch_fwd_ca<E, D, false, false, SGs<4, 4, 3, 1>>(p, st);
      if (t.nCT == 0) return launch_fwd_ca<E, D, false, false, SGs<4, 4, 0, 1>>(p, st);
    }
    // 16-token 1-D windows, four per iteration (cfg5: EVA with an 8-token extension and 8 landmarks; local attention)
    if (t.nQT == 1 && t.wpi == 4) {
      if (t.nLT == 2 && t.nCT == 1) return launch_fwd_ca<E, D, false, false, SGs<1, 2, 1, 4>>(p, st);
      if (t.nLT == 1 && t.nCT == 0) return launch_fwd_ca<E, D, false, false, SGs<1, 1, 0, 4>>(p, st);
    }
  }
  return launch_fwd_ca<E, D, false, false>(p, st);
}

int window_fwd_dispatch(const WinP& p, int dtype, int D, hipStream_t st) {
#define EA_CASE(EE, DD) return launch_fwd<EE, DD>(p, st)
  if (dtype == EA_BF16) {
    if (D == 64) EA_CASE(BF16, 64);
    if (D == 32) EA_CASE(BF16, 32);
    if (D == 128) EA_CASE(BF16, 128);
  } else if (dtype == EA_F16) {
    if (D == 64) EA_CASE(F16, 64);
    if (D == 32) EA_CASE(F16, 32);
    if (D == 128) EA_CASE(F16, 128);
  }
#undef EA_CASE
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
