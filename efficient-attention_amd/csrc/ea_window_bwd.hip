// ea_window_bwd.hip -- backward of window attention with control-variate (landmark) columns.
//
// Same workgroup <-> windows mapping and LDS images as the forward (ea_window_fwd.hip), plus the
// Q and dO rows of the staged windows.  With P = exp(logits - lse) recomputed from the saved
// log-sum-exp and delta_i = dO_i . O_i:
//     dS = P o (dO V^T - delta)          dQ = s dS K        dK = s dS^T Q       dV = P^T dO
//     d(rf_k_bar) = s dS_cv^T Q          d(beta) = P_cv^T dO                    dbias = dS_local
// Phase A (a wave per 16-query tile): S^T and dP^T tiles [key][query] -> dS^T stays in registers
//   as the B operand of dQ^T = K^T dS^T, with K^T fragments from ds_read_b64_tr_b16.
// Phase B (a wave per 16-key tile): S and dP tiles [query][key] -> P, dS stay in registers as the
//   B operands of dV^T = dO^T P and dK^T = Q^T dS (tr-reads of dO and Q).  Local key tiles are
//   stored straight to dk/dv; landmark key tiles accumulate in registers across all windows of the
//   workgroup and are written once as per-workgroup partial sums; the bias gradient accumulates in
//   LDS and is written once per workgroup.
// There is no barrier between the phases: delta and lse are staged with the Q/dO rows.
#include <stdlib.h>
#include "ea_window.h"
#include <type_traits>
#include <algorithm>

namespace ea {

// GB: the bias table is too large for LDS and is read from global memory (bias / biasT); otherwise
// every bias read goes to LDS -- the staged table, or a block of zeros when there is no bias -- so
// that the inner loops carry no branches and hipcc can interleave the independent tiles.
// CA: causal_eva.py geometry (ea_geom.causal != 0): per-(query, key) visibility limits, staged per
// query row in LDS for phase B.
// DR (with CA): attention dropout from an explicit keep mask (dP and the P of dV carry the mask).
// SG: static geometry (round 3).  With the tile counts as runtime fields of the tiling every loop of the two phases kept its
// bound and its local-vs-landmark selects at run time: the window loop compiled to 219 basic blocks (4.7 k instructions,
// nothing scheduled across them).  For the geometries the models use -- 7 x 7 windows with 49 landmarks (DeiT), local
// attention, 8 x 8 windows with 36 landmarks (PvT) -- the counts are template constants and the loops unroll into
// straight-line code; SGdyn keeps the general kernel.

// EA_RELANE (round 6): the run-time-geometry instantiations (SGdyn) keep ~40 lane-derived offsets alive from the prologue
// through both phases and spill them (68-160 B / lane of scratch; stores in the prologue, reloads at the phase boundaries).
// Each phase re-derives its lane coordinates from an opaque copy of the thread index instead (a dozen integer instructions
// per phase and iteration); for the static geometries the copy is transparent and the code is unchanged.
#define EA_RELANE(tag)                                                                                   \
  int tid_##tag = threadIdx.x;                                                                           \
  if constexpr (!STATIC) asm volatile("" : "+v"(tid_##tag));                                             \
  const int tid = tid_##tag, lane = tid & 63, g = lane >> 4, li = lane & 15;                             \
  const int wave = STATIC ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);                       \
  typename LaneOffSel<D>::type lo;                                                                       \
  lo.init(lane);                                                                                         \
  (void)tid; (void)g; (void)li; (void)wave

template <typename E, int D, bool GB, bool CA, bool DR, typename SG>
__global__ __launch_bounds__(256, D == 128 ? 1 : 2) void win_bwd_kernel(const WinP p, const T4 outp, const float* biasT) {
  constexpr int ROWB = D * 2;
  constexpr int CPR = D / 8;
  constexpr int KS = D / 32;
  constexpr int DT = D / 16;
  constexpr int DQ = D / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // merged query-block launch: this workgroup's tiling and its index within that tiling's range
  // (only in the causal_eva instantiations: a tiling picked at run time keeps its fields out of the
  // preloaded kernel-argument registers, which costs the other variants ~6 %)
  int qi = 0;
  if (CA && p.nq > 1) {
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < p.nq && (int)blockIdx.x >= p.qstart[i]) qi = i;
  }
  const WinTiling& t = (CA && p.nq > 1) ? p.tv[qi] : p.t;
  const int bid = (CA && p.nq > 1) ? (int)blockIdx.x - p.qstart[qi] : (int)blockIdx.x;
  constexpr bool STATIC = SG::NQT > 0;
  const int nQT = STATIC ? SG::NQT : t.nQT, nLT = STATIC ? SG::NLT : t.nLT, nCT = STATIC ? SG::NCT : t.nCT;
  const int wpi = STATIC ? SG::WPI : t.wpi;
  const int nchunks = STATIC ? (SG::NLT + SG::NCT + 3) / 4 : t.nchunks;
  const int rowsLocal = STATIC ? SG::WPI * SG::NLT * 16 : t.rowsLocal, rowsLm = STATIC ? SG::NCT * 16 : t.rowsLm;
  const int rowsTotal = rowsLocal + rowsLm + 16;
  const int biasLd = STATIC ? SG::NLT * 16 : t.biasLd;
  const int nQTe = (nQT + 1) & ~1;                   // query tiles per window, padded to even
  const int rowsQ = wpi * nQTe * 16;
  // HAND (round 3, static geometries, one window per iteration): phase A keeps P and dS of its (query tile, key tile)
  // pairs and hands them to phase B as [query][key] tiles in LDS; phase B reads them back transposed
  // (ds_read_b64_tr_b16) as the B operands of dV^T = dO^T P and dK^T = Q^T dS -- no second S / dP product, no second
  // exp.  The tiles of the LOCAL keys overlay the local K / V rows (dead once every wave has left phase A: phase B
  // needs only Q, dO and the tiles), those of the landmark keys take the room of the bias table and of the bias-gradient
  // accumulator, which live in registers here (a wave owns one query tile for the whole launch, so the bias rows it
  // reads and the bias-gradient entries it produces never change lanes).
  constexpr bool HAND = SG::HO;
  // NQW: (window, query tile) pairs of an iteration -- one per wave, the same pair index every iteration
  constexpr int NQW = STATIC ? SG::WPI * SG::NQT : 0;
  static_assert(!HAND || (STATIC && NQW <= 4 && SG::NLT <= 4 && !GB && !CA && !DR && D == 64),
                "hand-over variant: static geometries with at most four query tiles per iteration");
  constexpr int NKT = SG::NLT + SG::NCT;             // key tiles of a window (HAND)
  char* Ks = smem;
  // HAND: [K local | V local | K landmark (+ zero tile) | V landmark (+ zero tile)], else [K all | V all]
  char* Vs = Ks + (HAND ? rowsLocal : rowsTotal) * ROWB;
  char* Klm = HAND ? Vs + rowsLocal * ROWB : Ks + rowsLocal * ROWB;
  char* Vlm = HAND ? Klm + (rowsLm + 16) * ROWB : Vs + rowsLocal * ROWB;
  char* Qs = Ks + 2 * rowsTotal * ROWB;
  char* dOs = Qs + rowsQ * ROWB;
  float* lse_s = reinterpret_cast<float*>(dOs + rowsQ * ROWB);
  float* delta_s = lse_s + rowsQ;
  char* slabL = Ks;                                   // HAND: P / dS tiles of the local key tiles
  char* slabC = reinterpret_cast<char*>(delta_s + rowsQ);   // HAND: ... of the landmark key tiles
  auto slab_tile = [&](int qt, int kt) -> char* {     // 1 KB per (query tile, key tile): P then dS, [16 queries][16 keys]
    return kt < SG::NLT ? slabL + (qt * SG::NLT + kt) * 1024 : slabC + (qt * SG::NCT + (kt - SG::NLT)) * 1024;   // qt: pair index
  };
  // bias-gradient accumulator [Wq][BLD] and (bias_lds) the head's log2-domain bias [Wq][BLD]; the
  // odd row stride keeps both the row-wise (phase A) and the column-wise (phase B) accesses of
  // the 64 lanes on distinct banks
  const int BLD = biasLd + 1;
  float* dbias_s = HAND ? reinterpret_cast<float*>(slabC + NQW * SG::NCT * 1024) : delta_s + rowsQ;
  const int ntable = (p.bias && !HAND) ? t.Wq * BLD : 0;
  const bool dbd = CA && t.dbd != 0;                   // bias-gradient entries go straight to dbias_part (ea_window.h)
  const int nbias = dbd ? 0 : ntable;
  float* bias_s = dbias_s + nbias;
  float* zero64 = bias_s + (p.bias_lds ? ntable : 0);  // bias reads without a bias table land here
  float* trash64 = zero64 + 64;                        // bias-gradient writes of padded entries
  float* kmul = trash64 + 64;
  const float* bread = p.bias_lds ? bias_s : zero64;
  const int brs = p.bias_lds ? BLD : 0, btm = p.bias_lds ? 16 : 0;
  float* kadd = kmul + rowsTotal;
  int* kd = reinterpret_cast<int*>(kadd + rowsTotal);
  int* qd = kd + nLT * 16;
  int* qlim_s = qd + nQTe * 16;                      // CA only: last visible local slot / landmark per query row
  int* clim_s = qlim_s + rowsQ;
  const int rowsPerWin = nLT * 16;

  constexpr bool PHASE_A_GLOBAL_BIAS = GB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  typename LaneOffSel<D>::type lo;       // round-3 conflict-free tile layout for D = 64 (ea_common.h)
  lo.init(lane);
  const int bh = bid / t.nblk, blk = bid - bh * t.nblk;
  const int b = bh / p.H, h = bh - b * p.H;
  float* const dbias_g = (dbd && p.bias) ? p.dbias_part + ((((size_t)(t.bblk0 + blk) * p.B + b) * p.H + h) * t.WqFull + t.qoff) * (size_t)biasLd
                                         : nullptr;
  const char* qb = p.q.p + (b * p.q.sb + h * p.q.sh) * 2;
  const char* kb = p.k.p + (b * p.k.sb + h * p.k.sh) * 2;
  const char* vb = p.v.p + (b * p.v.sb + h * p.v.sh) * 2;
  const char* dob = p.o.p + (b * p.o.sb + h * p.o.sh) * 2;
  const char* ob = outp.p + (b * outp.sb + h * outp.sh) * 2;
  char* dqb = p.dq.p + (b * p.dq.sb + h * p.dq.sh) * 2;
  char* dkb = p.dk.p + (b * p.dk.sb + h * p.dk.sh) * 2;
  char* dvb = p.dv.p + (b * p.dv.sb + h * p.dv.sh) * 2;
  const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.G.N : nullptr;
  const float* lse_g = p.lse + (size_t)bh * p.G.N;
  // token strides as 32-bit element counts (N * stride < 2^31 is checked on the host): the per-row
  // address arithmetic stays in 32-bit VALU ops and the 64-bit strides free their SGPR pairs
  const int ksn = (int)p.k.sn, vsn = (int)p.v.sn, qsn = (int)p.q.sn, dosn = (int)p.o.sn, osn = (int)outp.sn;
  const int dqsn = (int)p.dq.sn, dksn = (int)p.dk.sn, dvsn = (int)p.dv.sn;

  EA_STAMP(p, 0);
  EA_BLK(p, 0);
  // ---- once per workgroup: landmark rows, zero tile, bias-gradient accumulator ----
  for (int idx = tid; idx < (rowsLm + 16) * CPR; idx += 256) {
    const int row = idx / CPR, c = idx - row * CPR;
    u32x4 kw = {0u, 0u, 0u, 0u}, vw = {0u, 0u, 0u, 0u};
    if (row < p.L) {
      const size_t off = ((size_t)bh * p.L + row) * D + c * 8;
      float f[8];
      *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p.lk + off);
      *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p.lk + off + 4);
      kw = pack8<E>(f);
      *reinterpret_cast<float4*>(f) = *reinterpret_cast<const float4*>(p.lv + off);
      *reinterpret_cast<float4*>(f + 4) = *reinterpret_cast<const float4*>(p.lv + off + 4);
      vw = pack8<E>(f);
    }
    sts16(Klm + TileL<D>::off(row, c), kw);
    sts16(Vlm + TileL<D>::off(row, c), vw);
    if (c == 0) {
      kmul[rowsLocal + row] = row < p.L ? 1.f : 0.f;
      kadd[rowsLocal + row] = row < p.L ? 0.f : -INFINITY;
    }
  }
  for (int idx = tid; idx < nbias; idx += 256) dbias_s[idx] = 0.f;
  if (tid < 128) zero64[tid] = 0.f;
  if (!HAND && p.bias_lds) {
    const float* bsrc = p.bias + ((size_t)h * t.WqFull + t.qoff) * biasLd;
    for (int idx = tid * 4; idx < t.Wq * biasLd; idx += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(bsrc + idx);
      float* d = bias_s + (idx / biasLd) * BLD + (idx % biasLd);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  build_slot_tables(kd, qd, t, p.G, p.w, p.e, nQTe * 16, tid);
  __syncthreads();

  // landmark-gradient accumulators of the landmark tile this wave owns (tile ct = wave)
  f32x4 dlk[DT], dlv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dlk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dlv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // HAND: this wave's query tile (= wave) for the whole launch: its bias rows and its bias-gradient entries
  // (query 16 wave + li, keys 16 tile + 4 g + r) in registers
  constexpr int HT = HAND ? (SG::NLT > 0 ? SG::NLT : 1) : 1;
  f32x4 breg[HT], dbacc[HT];
  uint32_t hp[HAND ? NKT : 1][2], hd[HAND ? NKT : 1][2];
  if constexpr (HAND) {
    const int qs = min((STATIC ? wave % (SG::NQT > 0 ? SG::NQT : 1) : 0) * 16 + li, t.Wq - 1);
#pragma unroll
    for (int tl = 0; tl < HT; ++tl) {
      dbacc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
      breg[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        const float4 v = *reinterpret_cast<const float4*>(p.bias + ((size_t)h * t.WqFull + t.qoff + qs) * biasLd + tl * 16 + 4 * g);
        breg[tl] = f32x4{v.x, v.y, v.z, v.w};
      }
      if (SG::PW) {
        // static key validity (slots beyond the window's keys) rides in the bias registers as an additive -inf
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (tl * 16 + 4 * g + r >= t.Wk) breg[tl][r] = -INFINITY;
      }
    }
  }
  const int it_end = min((blk + 1) * t.ipb, t.niter);
  EA_STAMP(p, 1);
  int prof_it = 0;
  (void)prof_it;
  for (int it = blk * t.ipb; it < it_end; ++it) {
    __syncthreads();
    if (prof_it < 4) EA_STAMP(p, 2 + prof_it * 6);
    // window origins of this iteration, once (static geometries): every staging slot, query tile and key tile used to redo
    // the two integer divisions behind win_origin -- ~23 scalar divisions per iteration in a kernel that already spills SGPRs
    int woy[STATIC ? SG::WPI : 1], wox[STATIC ? SG::WPI : 1];
    if constexpr (STATIC) {
#pragma unroll
      for (int wi = 0; wi < SG::WPI; ++wi)
        win_origin(p.G, colour_win(t, p.G, p.w, min(it * wpi + wi, t.nwin - 1)), p.w, woy[wi], wox[wi]);
    }
    auto origin_of = [&](int wi, int win, int& oy, int& ox) {
      if constexpr (STATIC) {
        oy = woy[0]; ox = wox[0];
#pragma unroll
        for (int k = 1; k < SG::WPI; ++k) { oy = wi == k ? woy[k] : oy; ox = wi == k ? wox[k] : ox; }
      } else {
        win_origin(p.G, colour_win(t, p.G, p.w, win), p.w, oy, ox);
      }
    };
    // ---- stage local K/V rows and Q/dO rows (+ lse, delta = dO.O).  Batches of NB slots per thread:
    // every global load of a batch is in flight before the first conversion / LDS store, so the
    // staging costs ~one memory round trip per iteration instead of one per 256-slot sweep. ----
    constexpr int NB = 2;
    struct KVb { u32x4 kr[NB], vr[NB]; int rowv[NB]; float mulv[NB], addv[NB]; };
    struct Qb { u32x4 qr[NB], dr[NB], orr[NB]; float lsv[NB], dls[NB]; int rowv[NB]; QLim lim[NB]; };
    // Every load of a batch is issued unconditionally from a clamped address and zeroed by a select afterwards: a
    // predicated load is an exec-mask branch of its own and the loads behind it wait for it (round 1's lesson)
    auto issueKV = [&](KVb& x, int base) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = base + tid + i * 256;
        const bool in = idx < rowsLocal * CPR;
        const int idc = in ? idx : 0;
        const int row = idc / CPR, c = idc - row * CPR;
        const int wi = wpi == 1 ? 0 : row / rowsPerWin;
        const int slot = row - wi * rowsPerWin;
        const int win = it * wpi + wi;
        const bool live = in && win < t.nwin && slot < t.Wk;
        int oy, ox;
        origin_of(wi, min(win, t.nwin - 1), oy, ox);
        const int tok = slot_token(p.G, kd[slot], oy, ox);
        const bool has = live && tok >= 0;
        const int tc = has ? tok : 0;
        const u32x4 kr = ldg16(kb + (tc * ksn + c * 8) * 2);
        const u32x4 vr = ldg16(vb + (tc * vsn + c * 8) * 2);
        bool keep = has;
        if (mrow) keep = keep && !mrow[tc];
        const u32x4 z = {0u, 0u, 0u, 0u};
        x.kr[i] = has ? kr : z;
        x.vr[i] = has ? vr : z;
        x.rowv[i] = in ? row : -1;
        x.mulv[i] = keep ? 1.f : 0.f;
        x.addv[i] = keep ? 0.f : (live ? MASK_FILL * LOG2E : -INFINITY);
      }
    };
    auto commitKV = [&](const KVb& x, int base) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (x.rowv[i] >= 0) {
          const int c = (base + tid + i * 256) - x.rowv[i] * CPR;
          sts16(Ks + TileL<D>::off(x.rowv[i], c), x.kr[i]);
          sts16(Vs + TileL<D>::off(x.rowv[i], c), x.vr[i]);
          if (c == 0) { kmul[x.rowv[i]] = x.mulv[i]; kadd[x.rowv[i]] = x.addv[i]; }
        }
      }
    };
    auto issueQ = [&](Qb& x, int base) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = base + tid + i * 256;
        const bool in = idx < rowsQ * CPR;
        const int idc = in ? idx : 0;
        const int row = idc / CPR, c = idc - row * CPR;
        const int wi = wpi == 1 ? 0 : row / (nQTe * 16);
        const int slot = row - wi * (nQTe * 16);
        const int win = it * wpi + wi;
        const bool live = in && win < t.nwin && slot < t.Wq;
        int oy, ox;
        origin_of(wi, min(win, t.nwin - 1), oy, ox);
        const int tokr = slot_token(p.G, qd[slot], oy, ox);
        const int tok = live ? tokr : -1;
        const bool has = tok >= 0;
        const int tc = has ? tok : 0;
        x.rowv[i] = in ? row : -1;
        if (CA && c == 0) x.lim[i] = query_limits(p.causal, t.qoff + slot, tok, p.e, p.chunk, mrow, p.lm_base);
        const u32x4 qr = ldg16(qb + (tc * qsn + c * 8) * 2);
        const u32x4 dr = ldg16(dob + (tc * dosn + c * 8) * 2);
        const u32x4 orr = ldg16(ob + (tc * osn + c * 8) * 2);
        const float ls = lse_g[tc];
        float dl = 0.f;
        if (p.dlse) dl = p.dlse[(size_t)bh * p.G.N + tc];
        const u32x4 z = {0u, 0u, 0u, 0u};
        x.qr[i] = has ? qr : z;
        x.dr[i] = has ? dr : z;
        x.orr[i] = has ? orr : z;
        x.lsv[i] = has ? ls * LOG2E : INFINITY;
        x.dls[i] = has ? dl : 0.f;
      }
    };
    auto commitQ = [&](const Qb& x, int base) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        float part = 0.f;
        float a8[8], o8[8];
        unpack8<E>(x.dr[i], a8);
        unpack8<E>(x.orr[i], o8);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += a8[j] * o8[j];
        part = group_sum<CPR>(part);
        if (x.rowv[i] >= 0) {
          const int c = (base + tid + i * 256) - x.rowv[i] * CPR;
          sts16(Qs + TileL<D>::off(x.rowv[i], c), x.qr[i]);
          sts16(dOs + TileL<D>::off(x.rowv[i], c), x.dr[i]);
          if (c == 0) {
            // a loss that also reads lse adds P o dlse to dS: dS = P o (dP - (delta - dlse))
            delta_s[x.rowv[i]] = part - x.dls[i]; lse_s[x.rowv[i]] = x.lsv[i];
            if (CA) { qlim_s[x.rowv[i]] = x.lim[i].local; clim_s[x.rowv[i]] = x.lim[i].lm; }
          }
        }
      }
    };
    {
      // first batch of both row sets in flight together (the whole iteration when wpi == 1)
      KVb kv0;
      Qb q0;
      issueKV(kv0, 0);
      issueQ(q0, 0);
      commitKV(kv0, 0);
      commitQ(q0, 0);
    }
    for (int base = 256 * NB; base < rowsLocal * CPR; base += 256 * NB) {
      KVb x;
      issueKV(x, base);
      commitKV(x, base);
    }
    if (prof_it < 4) EA_STAMP(p, 3 + prof_it * 6);
    for (int base = 256 * NB; base < rowsQ * CPR; base += 256 * NB) {
      Qb x;
      issueQ(x, base);
      commitQ(x, base);
    }
    if (prof_it < 4) EA_STAMP(p, 4 + prof_it * 6);
    __syncthreads();
    if (prof_it < 4) EA_STAMP(p, 5 + prof_it * 6);

    // =============================== phase A: dQ ===============================
    bool qact = false;                                 // HAND: this wave produced hand-over tiles in this iteration
    {
    EA_RELANE(A);
    for (int qi = wave; qi < wpi * nQT; qi += 4) {
      const int wi = qi / nQT, qt = qi - wi * nQT;
      const int win = it * wpi + wi;
      if (win >= t.nwin) continue;
      qact = true;
      const int qslot = qt * 16 + li;
      const int qrow = (wi * nQTe + qt) * 16 + li;
      int qtok = -1;
      if (qslot < t.Wq) {
        int oy, ox;
        origin_of(wi, win, oy, ox);
        qtok = slot_token(p.G, qd[qslot], oy, ox);
      }
      typename E::x8 qf[KS], dof[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = as_x8<E>(lds16(Qs + (qrow - li) * ROWB + lo.plain[ks]));
        dof[ks] = as_x8<E>(lds16(dOs + (qrow - li) * ROWB + lo.plain[ks]));
      }
      const float lse2 = lse_s[qrow], delta = delta_s[qrow];
      QLim ql;
      ql.local = ql.lm = 0x7fffffff;
      if (CA) { ql.local = qlim_s[qrow]; ql.lm = clim_s[qrow]; }
      const float* brow = (GB && p.bias)
          ? p.bias + ((size_t)h * t.WqFull + t.qoff + (qslot < t.Wq ? qslot : 0)) * biasLd + 4 * g : nullptr;
      const float* brow_s = bread + (qslot < t.Wq ? qslot : 0) * brs + 4 * g;
      f32x4 dq[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

      if (prof_it == 0) EA_STAMP(p, 20);
      // PW (HAND launches without a padding mask whose windows are all complete): key validity is static, so the
      // logit is ONE fma -- s * scale + (bias | -inf) -- and dS needs no gradient mask; otherwise the masked_fill form
      constexpr bool PW = HAND && SG::PW;
      // PFA (round 5, like PF of phase B below): the bias piece read from global memory (GB) and the keep-mask word (DR) of a
      // key tile are requested TWO tiles ahead, unconditionally, from clamped addresses (scalars in rotation, not arrays: the
      // tile loop is not always unrolled and an indexed array would live in scratch)
      constexpr bool PFA = D == 128 && ((PHASE_A_GLOBAL_BIAS && !HAND) || DR);
      const uint8_t* keep_row = DR ? p.keep + ((size_t)bh * p.G.N + (qtok >= 0 ? qtok : 0)) * p.keep_ld + 4 * g : nullptr;
      f32x4 pb0 = {0.f, 0.f, 0.f, 0.f}, pb1 = pb0, pb2 = pb0;     // (vector values, not structs: selects stay in registers)
      uint32_t pk0 = 0, pk1 = 0, pk2 = 0;
      const int ntiles_a = nchunks * 4;
      auto fetchA = [&](int tile_, f32x4& bq, uint32_t& kq) {
        const int tile = min(tile_, ntiles_a - 1);
        const bool local = tile < nLT;
        if (PHASE_A_GLOBAL_BIAS && !HAND) bq = *reinterpret_cast<const f32x4*>(brow + min(tile, nLT - 1) * 16);
        if (DR) {
          const int col = local ? tile * 16 : biasLd + max(min(tile - nLT, nCT - 1), 0) * 16;
          kq = *reinterpret_cast<const uint32_t*>(keep_row + ((nCT > 0 || local) ? col : 0));
        }
      };
      if (PFA) { fetchA(0, pb0, pk0); fetchA(1, pb1, pk1); }
      for (int ch = 0; ch < nchunks; ++ch) {
        int rowbase[4];
        const char* kt_p[4];
        uint32_t dsw[4][2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int tile = ch * 4 + tt;
          const bool local = tile < nLT;
          if (PFA) fetchA(tile + 2, pb2, pk2);
          rowbase[tt] = local ? (wi * nLT + tile) * 16
                              : (tile < nLT + nCT ? rowsLocal + (tile - nLT) * 16
                                                      : rowsLocal + rowsLm);
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          kt_p[tt] = local ? Ks + rowbase[tt] * ROWB : Klm + (rowbase[tt] - rowsLocal) * ROWB;
          const char* vt_p = local ? Vs + rowbase[tt] * ROWB : Vlm + (rowbase[tt] - rowsLocal) * ROWB;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s = E::mma(as_x8<E>(lds16(kt_p[tt] + lo.plain[ks])), qf[ks], s);
            dp = E::mma(as_x8<E>(lds16(vt_p + lo.plain[ks])), dof[ks], dp);
          }
          float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f), a4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!PW) m4 = *reinterpret_cast<const float4*>(kmul + rowbase[tt] + 4 * g);
          if (!PW || !local) a4 = *reinterpret_cast<const float4*>(kadd + rowbase[tt] + 4 * g);
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, aa[4] = {a4.x, a4.y, a4.z, a4.w};
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);                                  // log2-domain bias
          if constexpr (HAND) {
            if (local) b4 = make_float4(breg[tile][0], breg[tile][1], breg[tile][2], breg[tile][3]);
          } else if (PHASE_A_GLOBAL_BIAS) {
            if (PFA) { if (local) b4 = make_float4(pb0[0], pb0[1], pb0[2], pb0[3]); }
            else if (brow && local) b4 = *reinterpret_cast<const float4*>(brow + tile * 16);
          } else {
            const float* bs = brow_s + (local ? tile : 0) * btm;
            const float lf = local ? 1.f : 0.f;
            b4 = make_float4(bs[0] * lf, bs[1] * lf, bs[2] * lf, bs[3] * lf);
          }
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
          const int kidx0 = (local ? tile : tile - nLT) * 16 + 4 * g;
          const int lim = local ? ql.local : ql.lm;
          uint32_t keep4 = 0x01010101u;
          if (DR) {
            const bool real = tile < nLT + nCT;
            if (PFA) keep4 = real ? pk0 : 0u;
            else {
              const int col = local ? tile * 16 : biasLd + (tile - nLT) * 16;
              keep4 = real ? *reinterpret_cast<const uint32_t*>(keep_row + col) : 0u;
            }
          }
          float ds[4], prr[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = PW ? fmaf(s[r], p.scale_log2, local ? bb[r] : aa[r]) : fmaf(mm[r], fmaf(s[r], p.scale_log2, bb[r]), aa[r]);
            float gm = mm[r];
            if (CA) {
              const bool blocked = kidx0 + r > lim;
              x = blocked ? fminf(x, MASK_FILL * LOG2E) : x;
              gm = blocked ? 0.f : gm;
            }
            const float pr = fast_exp2(x - lse2);
            float dpr = dp[r];
            if (DR) dpr = ((keep4 >> (8 * r)) & 0xffu) ? dpr * p.keep_scale : 0.f;
            // masked_fill blocks the gradient of the replaced logits (mul == 0)
            ds[r] = PW ? pr * (dpr - delta) : gm * pr * (dpr - delta);
            prr[r] = pr;
          }
          dsw[tt][0] = pack2<E>(ds[0], ds[1]);
          dsw[tt][1] = pack2<E>(ds[2], ds[3]);
          if (PFA) { pb0 = pb1; pb1 = pb2; pk0 = pk1; pk1 = pk2; }
          if constexpr (HAND) {
            if (tile < NKT) {
              hp[tile][0] = pack2<E>(prr[0], prr[1]); hp[tile][1] = pack2<E>(prr[2], prr[3]);
              hd[tile][0] = dsw[tt][0]; hd[tile][1] = dsw[tt][1];
            }
            if (tile < SG::NLT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) dbacc[tile][r] += ds[r];
            }
          }
        }
        if (prof_it == 0 && ch < 3) EA_STAMP(p, 21 + 2 * ch);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          u32x4 f4v;
          f4v[0] = dsw[2 * kk][0]; f4v[1] = dsw[2 * kk][1];
          f4v[2] = dsw[2 * kk + 1][0]; f4v[3] = dsw[2 * kk + 1][1];
          const typename E::x8 dsf = as_x8<E>(f4v);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const u32x2 lo_ = E::tr4(kt_p[2 * kk] + lo.tr[dt]);
            const u32x2 hi_ = E::tr4(kt_p[2 * kk + 1] + lo.tr[dt]);
            dq[dt] = E::mma(as_x8<E>(lo_, hi_), dsf, dq[dt]);
          }
        }
        if (prof_it == 0 && ch < 3) EA_STAMP(p, 22 + 2 * ch);
      }
      if constexpr (TileL<D>::NEWTR) {
        // accumulator pieces (channels 16 dt + 4 g ..) -> the lane's 32 contiguous bytes (all lanes take part)
        u32x4 o0, o1;
        quad_transpose_pack<E>(dq, p.scale, o0, o1);
        if (qtok >= 0) {
          char* dst = dqb + (qtok * dqsn + DQ * g) * 2;
          stg16(dst, o0);
          stg16(dst + 16, o1);
        }
      } else if (qtok >= 0) {
        float f[DQ];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) f[4 * dt + r] = dq[dt][r] * p.scale;
        char* dst = dqb + (qtok * dqsn + DQ * g) * 2;
#pragma unroll
        for (int c = 0; c < DQ / 8; ++c) stg16(dst + c * 16, pack8<E>(f + 8 * c));
      }
    }
    }

    if constexpr (HAND) {
      __syncthreads();                                 // every wave is done with the local K / V rows
      if (wave < NQW && qact) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          // S^T tile in registers: keys 4 g + r of query li  ->  row li (32 bytes), bytes 8 g .. of the [query][key] tile
          char* dst = slab_tile(wave, kt) + li * 32 + g * 8;
          *reinterpret_cast<u32x2*>(dst) = u32x2{hp[kt][0], hp[kt][1]};
          *reinterpret_cast<u32x2*>(dst + 512) = u32x2{hd[kt][0], hd[kt][1]};
        }
      }
      __syncthreads();
    }
    if (prof_it < 4) EA_STAMP(p, 6 + prof_it * 6);
    // =============================== phase B: dK, dV ===============================
    // work items: (window wi, local tile lt) for all staged windows, then landmark tile = wave
    EA_RELANE(B);
    const int nLocalItems = wpi * nLT;
    // (is_lm as a compile-time tag: the local and the landmark items are two straight-line instances of the body)
    auto process_item = [&](auto lm_tag, int item) {
      constexpr bool is_lm = decltype(lm_tag)::value;
      int tile, wi_lo, wi_hi;
      if (is_lm) {
        if constexpr (HAND && SG::WPI > 1) {
          // several windows per iteration, ONE landmark tile: every wave takes the landmark columns of its own window
          // (summed over the waves after the last iteration) instead of wave 0 taking all of them
          static_assert(SG::NCT <= 1, "hand-over with several windows per iteration: at most one landmark tile");
          tile = 0;
          if (SG::NCT == 0 || wave >= SG::WPI) return;
          wi_lo = wave; wi_hi = wave + 1;
        } else {
          tile = wave;                                 // landmark tile owned by this wave
          if (tile >= nCT) return;
          wi_lo = 0; wi_hi = wpi;
        }
      } else {
        wi_lo = item / nLT; wi_hi = wi_lo + 1;
        tile = item - wi_lo * nLT;
        if (it * wpi + wi_lo >= t.nwin) return;
      }
      const int krow = (is_lm ? rowsLocal + tile * 16 : (wi_lo * nLT + tile) * 16) + li;
      const char* kbp = is_lm ? Klm + tile * 16 * ROWB : Ks + (wi_lo * nLT + tile) * 16 * ROWB;
      const char* vbp = is_lm ? Vlm + tile * 16 * ROWB : Vs + (wi_lo * nLT + tile) * 16 * ROWB;
      typename E::x8 kf[KS], vf[KS];
      if constexpr (!HAND) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          kf[ks] = as_x8<E>(lds16(kbp + lo.plain[ks]));
          vf[ks] = as_x8<E>(lds16(vbp + lo.plain[ks]));
        }
      }
      const float kmu = kmul[krow], kad = kadd[krow];
      const int kslot = tile * 16 + li;                // key slot within the window / landmark id
      const int pb = 30 + (is_lm ? 12 : 0);
      if (prof_it == 0) EA_STAMP(p, pb);
      f32x4 dk[DT], dv[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

      // BM: 0 = no bias terms (landmark keys, or no bias at all); 1 = bias, one window per
      // iteration: the (query, key) entry of the bias gradient belongs to this lane alone, so a plain
      // LDS read-modify-write does (float atomics stall the LDS queue behind them); 2 = bias, several
      // windows in flight: LDS atomics.  Padded entries write to a trash line instead of branching.
      auto sweep = [&](auto bm_tag) {
        constexpr int BM = decltype(bm_tag)::value;
        // PF (round 5): the global-memory operands of a step's elementwise stage -- the 16-B piece of the transposed bias
        // table (GB) and the four keep-mask bytes (DR) of each of the two query tiles -- are requested one step AHEAD, from
        // clamped addresses (no exec-mask branch), so that their latency overlaps the previous step's products.  These
        // instantiations run at one wave per SIMD (the 143 KB image of the LM geometry): a load issued where it is used parks
        // the whole SIMD (SQ counters, LM backward: 48 % of the wave cycles parked).
        constexpr bool PF = D == 128 && ((GB && BM != 0) || DR);     // (the 256-register instantiations have no room for it)
        auto fetch1 = [&](int wi, int qt_, f32x4& bq, uint32_t& kq) {
          const int q0c = min(qt_, nQT - 1) * 16 + 4 * g;
          if (GB && BM != 0)
            bq = *reinterpret_cast<const f32x4*>(
                biasT + ((size_t)h * biasLd + min(kslot, t.Wk - 1)) * (ceil_div(t.WqFull, 16) * 16) + t.qoff + q0c);
          if (DR) {
            const int qbase = colour_win(t, p.G, p.w, it * wpi + wi) * p.w + t.qoff;
            const int kcol = is_lm ? biasLd + kslot : kslot;
            const uint8_t* kp = p.keep + ((size_t)bh * p.G.N + qbase) * p.keep_ld + kcol;
            uint32_t w = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) w |= (uint32_t)kp[(size_t)min(q0c + r, t.Wq - 1) * p.keep_ld] << (8 * r);
            kq = w;
          }
        };
        for (int wi = wi_lo; wi < wi_hi; ++wi) {
          if (it * wpi + wi >= t.nwin) break;
          f32x4 cb0 = {0.f, 0.f, 0.f, 0.f}, cb1 = cb0, nb0 = cb0, nb1 = cb0;
          uint32_t ck0 = 0, ck1 = 0, nk0 = 0, nk1 = 0;
          if (PF) { fetch1(wi, 0, cb0, ck0); fetch1(wi, 1, cb1, ck1); }
          for (int qq = 0; qq < nQTe / 2; ++qq) {
            uint32_t pw[2][2], dsw[2][2];
            int rq[2];
            if (PF) {
              const int qn = min(qq + 1, nQTe / 2 - 1);
              fetch1(wi, 2 * qn, nb0, nk0);
              fetch1(wi, 2 * qn + 1, nb1, nk1);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int qt = 2 * qq + u;
              rq[u] = (wi * nQTe + qt) * 16;
              f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                s = E::mma(as_x8<E>(lds16(Qs + rq[u] * ROWB + lo.plain[ks])), kf[ks], s);
                dp = E::mma(as_x8<E>(lds16(dOs + rq[u] * ROWB + lo.plain[ks])), vf[ks], dp);
              }
              const float4 l4 = *reinterpret_cast<const float4*>(lse_s + rq[u] + 4 * g);
              const float4 d4 = *reinterpret_cast<const float4*>(delta_s + rq[u] + 4 * g);
              const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
              int vis[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};   // CA: last visible key of each query
              if (CA) {
                const int4 v4 = *reinterpret_cast<const int4*>((is_lm ? clim_s : qlim_s) + rq[u] + 4 * g);
                vis[0] = v4.x; vis[1] = v4.y; vis[2] = v4.z; vis[3] = v4.w;
              }
              const int q0 = qt * 16 + 4 * g;            // first of this lane's four query slots
              float bt[4] = {0.f, 0.f, 0.f, 0.f};
              if (BM != 0) {
                if (GB && PF) {
                  // from the transposed copy in global memory: one 16-B load (requested a step ahead)
                  const f32x4 cb = u == 0 ? cb0 : cb1;
                  bt[0] = cb[0]; bt[1] = cb[1]; bt[2] = cb[2]; bt[3] = cb[3];
                } else if (GB) {
                  if (kslot < t.Wk) {
                    const float4 bt4 = *reinterpret_cast<const float4*>(
                        biasT + ((size_t)h * biasLd + kslot) * (ceil_div(t.WqFull, 16) * 16) + t.qoff + q0);
                    bt[0] = bt4.x; bt[1] = bt4.y; bt[2] = bt4.z; bt[3] = bt4.w;
                  }
                } else {
                  const float* bs = bread + kslot;       // rows beyond Wq are masked by bias_on below
#pragma unroll
                  for (int r = 0; r < 4; ++r) bt[r] = bs[min(q0 + r, t.Wq - 1) * brs];
                }
              }
              float pr[4], ds[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int qs = q0 + r;                    // query slot in the window
                const bool bias_on = BM != 0 && qs < t.Wq && kslot < t.Wk;
                const float bias = bias_on ? bt[r] : 0.f;
                float x = fmaf(kmu, fmaf(s[r], p.scale_log2, bias), kad);
                float gm = kmu;
                if (CA) {
                  const bool blocked = kslot > vis[r];
                  x = blocked ? fminf(x, MASK_FILL * LOG2E) : x;
                  gm = blocked ? 0.f : gm;
                }
                pr[r] = fast_exp2(x - ll[r]);
                float dpr = dp[r];
                float km = 1.f;
                if (DR) {
                  // 1-D windows: token of query slot qs of window wi; this lane's softmax column (byte r of the word
                  // fetched a step ahead)
                  if (PF) {
                    km = (((u == 0 ? ck0 : ck1) >> (8 * r)) & 0xffu) ? p.keep_scale : 0.f;
                  } else {
                    const int qtk = colour_win(t, p.G, p.w, it * wpi + wi) * p.w + t.qoff + min(qs, t.Wq - 1);
                    const int kcol = is_lm ? biasLd + kslot : kslot;
                    km = p.keep[((size_t)bh * p.G.N + qtk) * p.keep_ld + kcol] ? p.keep_scale : 0.f;
                  }
                  dpr *= km;
                }
                ds[r] = gm * pr[r] * (dpr - dd[r]);
                if (DR) pr[r] *= km;                        // dV sees the dropped probabilities
                if (BM == 3) {
                  // one window per workgroup: this lane produces the entry exactly once (padded columns of the row as zeros)
                  if (qs < t.Wq) dbias_g[(size_t)qs * biasLd + kslot] = bias_on ? ds[r] : 0.f;
                } else if (BM == 1) {
                  float* dst = bias_on ? dbias_s + qs * BLD + kslot : trash64 + lane;
                  *dst += ds[r];
                } else if (BM == 2) {
                  if (bias_on && ds[r] != 0.f) atomicAdd(dbias_s + qs * BLD + kslot, ds[r]);
                }
              }
              pw[u][0] = pack2<E>(pr[0], pr[1]); pw[u][1] = pack2<E>(pr[2], pr[3]);
              dsw[u][0] = pack2<E>(ds[0], ds[1]); dsw[u][1] = pack2<E>(ds[2], ds[3]);
            }
            u32x4 a4, b4;
            a4[0] = pw[0][0]; a4[1] = pw[0][1]; a4[2] = pw[1][0]; a4[3] = pw[1][1];
            b4[0] = dsw[0][0]; b4[1] = dsw[0][1]; b4[2] = dsw[1][0]; b4[3] = dsw[1][1];
            const typename E::x8 pf = as_x8<E>(a4), dsf = as_x8<E>(b4);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              const int o0 = rq[0] * ROWB + lo.tr[dt];
              const int o1 = rq[1] * ROWB + lo.tr[dt];
              dv[dt] = E::mma(as_x8<E>(E::tr4(dOs + o0), E::tr4(dOs + o1)), pf, dv[dt]);
              dk[dt] = E::mma(as_x8<E>(E::tr4(Qs + o0), E::tr4(Qs + o1)), dsf, dk[dt]);
            }
            if (PF) { cb0 = nb0; cb1 = nb1; ck0 = nk0; ck1 = nk1; }
          }
        }
      };
      if constexpr (HAND) {
        // P and dS of (query tiles 2 qq, 2 qq + 1; this key tile) from the hand-over tiles: lane (g, key li) gets
        // queries 4 g .. 4 g + 3 of each -- the k-slot order of the transposed dO / Q fragments
        const int kt = is_lm ? SG::NLT + tile : tile;
        for (int wi = wi_lo; wi < wi_hi; ++wi) {
          if (it * wpi + wi >= t.nwin) break;
#pragma unroll
          for (int qq = 0; qq < (SG::NQT + 1) / 2; ++qq) {
            const int q0 = wi * SG::NQT + 2 * qq;          // pair index of the first query tile
            const char* t0 = slab_tile(q0, kt) + lane * 8;
            const char* t1 = slab_tile(2 * qq + 1 < SG::NQT ? q0 + 1 : q0, kt) + lane * 8;
            u32x2 p1 = E::tr4(t1), d1 = E::tr4(t1 + 512);
            if (2 * qq + 1 >= SG::NQT) { p1 = u32x2{0u, 0u}; d1 = u32x2{0u, 0u}; }
            const typename E::x8 pf = as_x8<E>(E::tr4(t0), p1), dsf = as_x8<E>(E::tr4(t0 + 512), d1);
            const int rq0 = (wi * nQTe + 2 * qq) * 16, rq1 = rq0 + 16;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              const int o0 = rq0 * ROWB + lo.tr[dt];
              const int o1 = rq1 * ROWB + lo.tr[dt];
              dv[dt] = E::mma(as_x8<E>(E::tr4(dOs + o0), E::tr4(dOs + o1)), pf, dv[dt]);
              dk[dt] = E::mma(as_x8<E>(E::tr4(Qs + o0), E::tr4(Qs + o1)), dsf, dk[dt]);
            }
          }
        }
      } else if (is_lm || !p.bias) sweep(std::integral_constant<int, 0>{});
      else if (CA && dbd) { if constexpr (CA) sweep(std::integral_constant<int, 3>{}); }
      else if (wpi == 1) sweep(std::integral_constant<int, 1>{});
      else sweep(std::integral_constant<int, 2>{});
      if (prof_it == 0) EA_STAMP(p, pb + 9);
      if (is_lm) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dlk[dt] += dk[dt]; dlv[dt] += dv[dt]; }
      } else {
        const int win = it * wpi + wi_lo;
        int tok = -1;
        if (kslot < t.Wk) {
          int oy, ox;
          origin_of(wi_lo, win, oy, ox);
          tok = slot_token(p.G, kd[kslot], oy, ox);
        }
        float fk[DQ], fv[DQ];
        if constexpr (TileL<D>::NEWTR) {
          quad_transpose_f32(dk, fk);                 // (all lanes take part, before the per-token predicate)
          quad_transpose_f32(dv, fv);
#pragma unroll
          for (int j = 0; j < DQ; ++j) fk[j] *= p.scale;
        } else {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { fk[4 * dt + r] = dk[dt][r] * p.scale; fv[4 * dt + r] = dv[dt][r]; }
        }
        if (tok >= 0) {
          if (p.acc_mode == 0) {
            char* d1 = dkb + (tok * dksn + DQ * g) * 2;
            char* d2 = dvb + (tok * dvsn + DQ * g) * 2;
#pragma unroll
            for (int c = 0; c < DQ / 8; ++c) {
              stg16(d1 + c * 16, pack8<E>(fk + 8 * c));
              stg16(d2 + c * 16, pack8<E>(fv + 8 * c));
            }
          } else if (p.acc_mode == 4) {
            // half-window overlap, colour classes run back to back: class 0 stores, class 1 adds to what class 0 stored
            // (a token of the last e is not covered by class 0: plain store)
            char* d1 = dkb + (tok * dksn + DQ * g) * 2;
            char* d2 = dvb + (tok * dvsn + DQ * g) * 2;
            const bool add = t.col_x != 0 && 2 * ((tok + p.e) / (2 * p.w)) < p.G.N / p.w;
#pragma unroll
            for (int c = 0; c < DQ / 8; ++c) {
              if (add) {
                float o1[8], o2[8];
                unpack8<E>(ldg16(d1 + c * 16), o1);
                unpack8<E>(ldg16(d2 + c * 16), o2);
#pragma unroll
                for (int j = 0; j < 8; ++j) { fk[8 * c + j] += o1[j]; fv[8 * c + j] += o2[j]; }
              }
              stg16(d1 + c * 16, pack8<E>(fk + 8 * c));
              stg16(d2 + c * 16, pack8<E>(fv + 8 * c));
            }
          } else if (p.acc_mode >= 2) {
            // merged query blocks: this block's own slice, every (token, channel) written once
            const size_t row = ((size_t)t.slice * p.B * p.H + bh) * p.G.N + tok;
            float4* a1 = reinterpret_cast<float4*>(p.dk32 + row * D + DQ * g);
            float4* a2 = reinterpret_cast<float4*>(p.dv32 + row * D + DQ * g);
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
              a1[i] = make_float4(fk[4 * i], fk[4 * i + 1], fk[4 * i + 2], fk[4 * i + 3]);
              a2[i] = make_float4(fv[4 * i], fv[4 * i + 1], fv[4 * i + 2], fv[4 * i + 3]);
            }
          } else {
            // overlapping windows / query blocks: a token is a key of several (window, query block)
            // pairs, but of only one of this launch -> plain 16-B read-modify-write into the fp32 scratch
            float4* a1 = reinterpret_cast<float4*>(p.dk32 + ((size_t)bh * p.G.N + tok) * D + DQ * g);
            float4* a2 = reinterpret_cast<float4*>(p.dv32 + ((size_t)bh * p.G.N + tok) * D + DQ * g);
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
              float4 x = a1[i], y = a2[i];
              x.x += fk[4 * i]; x.y += fk[4 * i + 1]; x.z += fk[4 * i + 2]; x.w += fk[4 * i + 3];
              y.x += fv[4 * i]; y.y += fv[4 * i + 1]; y.z += fv[4 * i + 2]; y.w += fv[4 * i + 3];
              a1[i] = x; a2[i] = y;
            }
          }
        }
      }
    };
    for (int item = wave; item < nLocalItems; item += 4) process_item(std::false_type{}, item);
    process_item(std::true_type{}, nLocalItems + wave);
    if (prof_it == 0) EA_STAMP(p, 55);
    if (prof_it < 4) EA_STAMP(p, 7 + prof_it * 6);
    ++prof_it;
  }
  EA_STAMP(p, 60);
  {
  EA_RELANE(E);

  // ---- per-workgroup partial sums of the landmark and bias gradients ----
  if constexpr (HAND && SG::WPI > 1 && SG::NCT == 1) {
    // the waves' landmark partials (one window each per iteration) -> wave 0, in wave order
    __syncthreads();
    float* r2 = reinterpret_cast<float*>(smem);        // [wave][dlk | dlv][dt][lane] x 4 floats
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      *reinterpret_cast<f32x4*>(r2 + (((wave * 2 + 0) * DT + dt) * 64 + lane) * 4) = dlk[dt];
      *reinterpret_cast<f32x4*>(r2 + (((wave * 2 + 1) * DT + dt) * 64 + lane) * 4) = dlv[dt];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 1; w < SG::WPI; ++w)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dlk[dt] += *reinterpret_cast<const f32x4*>(r2 + (((w * 2 + 0) * DT + dt) * 64 + lane) * 4);
          dlv[dt] += *reinterpret_cast<const f32x4*>(r2 + (((w * 2 + 1) * DT + dt) * 64 + lane) * 4);
        }
    }
  }
  if (p.L > 0 && wave < nCT) {
    const int lm = wave * 16 + li;
    if (lm < p.L) {
      float* d1 = p.dlk_part + (((size_t)(t.blk0 + blk) * p.B * p.H + bh) * p.L + lm) * D;
      float* d2 = p.dlv_part + (((size_t)(t.blk0 + blk) * p.B * p.H + bh) * p.L + lm) * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        *reinterpret_cast<float4*>(d1 + acc_chan<D>(dt, g)) =
            make_float4(dlk[dt][0] * p.scale, dlk[dt][1] * p.scale, dlk[dt][2] * p.scale, dlk[dt][3] * p.scale);
        *reinterpret_cast<float4*>(d2 + acc_chan<D>(dt, g)) = make_float4(dlv[dt][0], dlv[dt][1], dlv[dt][2], dlv[dt][3]);
      }
    }
  }
  if (p.bias && !dbd) {
    float* dst = p.dbias_part + ((((size_t)(t.bblk0 + blk) * p.B + b) * p.H + h) * t.WqFull + t.qoff) * (size_t)biasLd;
    if constexpr (HAND && SG::WPI == 1) {
      const int qs = wave * 16 + li;
      if (wave < SG::NQT && qs < t.Wq) {
#pragma unroll
        for (int tl = 0; tl < HT; ++tl)
          *reinterpret_cast<float4*>(dst + (size_t)qs * biasLd + tl * 16 + 4 * g) =
              make_float4(dbacc[tl][0], dbacc[tl][1], dbacc[tl][2], dbacc[tl][3]);
      }
    } else if constexpr (HAND) {
      // several windows per iteration: the waves that own the same query tile of different windows add up (fixed order)
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);     // [NQW][16 queries][NLT * 16 keys]
      constexpr int RW = SG::NLT * 16;
      if (wave < NQW) {
#pragma unroll
        for (int tl = 0; tl < HT; ++tl)
          *reinterpret_cast<float4*>(red + (wave * 16 + li) * RW + tl * 16 + 4 * g) =
              make_float4(dbacc[tl][0], dbacc[tl][1], dbacc[tl][2], dbacc[tl][3]);
      }
      __syncthreads();
      for (int idx = tid; idx < t.Wq * biasLd; idx += 256) {
        const int qs = idx / biasLd, k = idx - qs * biasLd;
        const int qt = qs >> 4;
        float sum = 0.f;
#pragma unroll
        for (int wi = 0; wi < SG::WPI; ++wi) sum += red[((wi * SG::NQT + qt) * 16 + (qs & 15)) * RW + k];
        dst[idx] = sum;
      }
    } else {
      __syncthreads();
      for (int idx = tid; idx < t.Wq * biasLd; idx += 256) dst[idx] = dbias_s[(idx / biasLd) * BLD + (idx % biasLd)];
    }
  }
  }
  EA_STAMP(p, 61);
  EA_BLK(p, 1);
}
#undef EA_RELANE

// fp32 scratch -> I/O dtype for the overlapping-window path
template <typename E, int D>
__global__ __launch_bounds__(256) void win_bwd_finish_kernel(const WinP p) {
  const long total = (long)p.B * p.H * p.G.N * (D / 8);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % (D / 8));
    const long row = idx / (D / 8);
    const int tok = (int)(row % p.G.N);
    const int bh = (int)(row / p.G.N), b = bh / p.H, h = bh - b * p.H;
    float f[8], f2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f2[i] = 0.f;
    // slices that hold this token: all of them, or (causal masks: a query block's key list ends at
    // its own last query) the blocks from the token's own on
    int s0 = 0, s1n = 1, ncol = 1;
    if (p.acc_mode == 2) {
      s1n = p.t.qsplit;
      if (p.causal == 2) s0 = (tok % p.w) / p.t.Wq;
    } else if (p.acc_mode == 3) {
      // windows g whose key range [g w - e, g w + w + e_right) holds the token; slice = g mod ncx
      const int er = p.causal ? 0 : p.e;
      const int nwin = (p.G.N + p.w - 1) / p.w;
      const int tl = tok - p.w - er;
      s0 = (tl >= 0 ? tl / p.w : -1) + 1;
      s1n = min((tok + p.e) / p.w, nwin - 1) + 1;
      ncol = p.t.ncx;
    }
    const size_t slice = (size_t)p.B * p.H * p.G.N * D;
    for (int sl = s0; sl < s1n; ++sl) {
      const int si = p.acc_mode == 3 ? sl % ncol : sl;
      const float* s1 = p.dk32 + si * slice + row * D + c * 8;
      const float* s2 = p.dv32 + si * slice + row * D + c * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) { f[i] += s1[i]; f2[i] += s2[i]; }
    }
    stg16(p.dk.p + (b * p.dk.sb + h * p.dk.sh + tok * p.dk.sn + c * 8) * 2, pack8<E>(f));
    stg16(p.dv.p + (b * p.dv.sb + h * p.dv.sh + tok * p.dv.sn + c * 8) * 2, pack8<E>(f2));
  }
}

// dev switch: EA_WIN_HAND_SMALL=0 keeps the general kernel for the 16-token 1-D windows
static bool win_hand_small_on() {
  static const bool v = [] { const char* e = getenv("EA_WIN_HAND_SMALL"); return !e || atoi(e) != 0; }();
  return v;
}

template <typename E, int D>
static int launch_bwd(const WinP& p0, const ea_geom& geom, const T4& outp, const float* biasT, hipStream_t st) {
  using KernelT = void (*)(const WinP, const T4, const float*);
  const bool single = win_bwd_single(p0.t), merged = win_bwd_merged(p0.t);
  if (p0.keep && !p0.causal) return EA_E_UNSUPPORTED;
  auto pick = [&](const WinP& p, bool gb) -> KernelT {
    if (p.keep) return gb ? &win_bwd_kernel<E, D, true, true, true, SGdyn> : &win_bwd_kernel<E, D, false, true, true, SGdyn>;
    if (p.causal) return gb ? &win_bwd_kernel<E, D, true, true, false, SGdyn> : &win_bwd_kernel<E, D, false, true, false, SGdyn>;
    if constexpr (D == 64) {
      // static-geometry instantiations (single launch, bias table in LDS or none)
      if (!gb && p.nq <= 1) {
        const WinTiling& t = p.t;
        if (t.nQT == 4 && t.nLT == 4 && t.wpi == 1) {
          // (the static geometries always take the hand-over kernels; the recomputing phase B lives on in SGdyn only)
          if (p.plain) {
            if (t.nCT == 4) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 4, 1, true, true>>;
            if (t.nCT == 3) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 3, 1, true, true>>;
            if (t.nCT == 0) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 0, 1, true, true>>;
          }
          if (t.nCT == 4) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 4, 1, true>>;
          if (t.nCT == 3) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 3, 1, true>>;
          if (t.nCT == 0) return &win_bwd_kernel<E, D, false, false, false, SGs<4, 4, 0, 1, true>>;
        }
        // 16-token 1-D windows (the LRA / cfg5 geometries), four windows per iteration: with an 8-token extension and up to
        // 16 landmarks (EVA), plain (local attention)
        if (t.nQT == 1 && t.wpi == 4 && win_hand_small_on()) {
          if (t.nLT == 2 && t.nCT == 1) return &win_bwd_kernel<E, D, false, false, false, SGs<1, 2, 1, 4, true>>;
          if (t.nLT == 1 && t.nCT == 0) return &win_bwd_kernel<E, D, false, false, false, SGs<1, 1, 0, 4, true>>;
        }
      }
    }
    return gb ? &win_bwd_kernel<E, D, true, false, false, SGdyn> : &win_bwd_kernel<E, D, false, false, false, SGdyn>;
  };
  auto launch = [&](WinP& p, size_t lds, unsigned blocks) -> int {
    const bool gb = p.bias && !p.bias_lds;
    static const bool plain_on = [] { const char* e = getenv("EA_WIN_PLAIN"); return !e || atoi(e) != 0; }();   // dev switch
    p.plain = (plain_on && !p.mask && p.e == 0 &&
               (p.G.attn2d ? (p.G.gh % p.w == 0 && p.G.gw % p.w == 0) : p.G.N % p.w == 0)) ? 1 : 0;
    const KernelT kern = pick(p, gb);
    const bool hand_big = p.t.nQT == 4 && p.t.nLT == 4 && p.t.wpi == 1 && (p.t.nCT == 4 || p.t.nCT == 3 || p.t.nCT == 0);
    const bool hand_small = p.t.nQT == 1 && p.t.wpi == 4 && win_hand_small_on() &&
                            ((p.t.nLT == 2 && p.t.nCT == 1) || (p.t.nLT == 1 && p.t.nCT == 0));
    if (D == 64 && !p.keep && !p.causal && !gb && p.nq <= 1 && (hand_big || hand_small))
      // hand-over variant: no bias table / bias-gradient image in LDS, P / dS tiles of the landmark keys instead
      lds = window_bwd_lds(p.t, D, false, false) + (size_t)p.t.wpi * p.t.nQT * p.t.nCT * 1024;
    if (lds > WIN_LDS_MAX) return EA_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, p, outp, biasT);
    return EA_OK;
  };
  int rc = EA_OK;
  if (merged) {
    // all query blocks of the (non-overlapping) windows in one launch, the longest key lists first
    WinTiling tl[4];
    int n = 0;
    win_bwd_launches(geom, p0.t, [&](const WinTiling& t) { tl[n] = t; tl[n].slice = t.qoff / t.Wq; ++n; });
    WinP p = p0;
    p.acc_mode = 2;
    p.nq = n;
    // the bias table is staged in LDS only if every block's image has room for it
    bool fits = p.bias != nullptr;
    for (int i = 0; i < n; ++i) fits = fits && window_bwd_lds(tl[i], D, true, true) <= WIN_LDS_MAX;
    p.bias_lds = fits ? 1 : 0;
    size_t lds = 0;
    int start = 0;
    for (int i = 0; i < n; ++i) {
      p.tv[i] = tl[n - 1 - i];
      p.qstart[i] = start;
      start += p.B * p.H * p.tv[i].nblk;
      lds = std::max(lds, window_bwd_lds(p.tv[i], D, p.bias != nullptr, p.bias_lds != 0));
    }
    p.qstart[n] = start;
    rc = launch(p, lds, (unsigned)start);
  } else {
    const bool cslices = win_bwd_colour_slices(p0.t), cdirect = p0.t.cdirect != 0;
    if (!single && !cslices && !cdirect) {
      const size_t bytes = (size_t)p0.B * p0.H * p0.G.N * D * sizeof(float);
      hipError_t e = hipMemsetAsync(p0.dk32, 0, bytes, st);
      if (e == hipSuccess) e = hipMemsetAsync(p0.dv32, 0, bytes, st);
      if (e != hipSuccess) return (int)e;
    }
    // one launch per (colour class, query block); stream order separates them
    win_bwd_launches(geom, p0.t, [&](const WinTiling& t) {
      if (rc != EA_OK) return;
      WinP p = p0;
      p.t = t;
      p.t.slice = t.col_x;
      p.acc_mode = single ? 0 : (cdirect ? 4 : (cslices ? 3 : 1));
      p.nq = 0;
      // the head's bias table lives in LDS whenever it fits (it is read once per (query, key) pair of
      // every window; from global memory those loads sit exposed in the inner loops)
      p.bias_lds = (p.bias && window_bwd_lds(p.t, D, true, true) <= WIN_LDS_MAX) ? 1 : 0;
      rc = launch(p, window_bwd_lds(p.t, D, p.bias != nullptr, p.bias_lds != 0), (unsigned)(p.B * p.H * p.t.nblk));
    });
  }
  if (rc != EA_OK) return rc;
  if (!single && !p0.t.cdirect) {
    WinP pf = p0;
    pf.acc_mode = merged ? 2 : (win_bwd_colour_slices(p0.t) ? 3 : 1);
    hipLaunchKernelGGL((win_bwd_finish_kernel<E, D>), dim3(2048), dim3(256), 0, st, pf);
  }
  return (int)hipGetLastError();
}

int window_bwd_dispatch(const WinP& p0, const ea_geom& geom, const T4& outp, const float* biasT, int dtype, int D,
                        hipStream_t st) {
  WinP p = p0;
  p.prof = nullptr;
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "win_bwd", 0);
#endif
  if (dtype == EA_BF16) {
    if (D == 64) return launch_bwd<BF16, 64>(p, geom, outp, biasT, st);
    if (D == 32) return launch_bwd<BF16, 32>(p, geom, outp, biasT, st);
    if (D == 128) return launch_bwd<BF16, 128>(p, geom, outp, biasT, st);
  } else if (dtype == EA_F16) {
    if (D == 64) return launch_bwd<F16, 64>(p, geom, outp, biasT, st);
    if (D == 32) return launch_bwd<F16, 32>(p, geom, outp, biasT, st);
    if (D == 128) return launch_bwd<F16, 128>(p, geom, outp, biasT, st);
  }
  return EA_E_UNSUPPORTED;
}

}  // namespace ea
