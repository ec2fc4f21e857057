// ea_window.h -- host-side derived geometry + kernel parameter block of the window-attention
// kernels (ea_window_fwd.hip / ea_window_bwd.hip).
#pragma once
#include <stdlib.h>
#include "ea_common.h"

namespace ea {

// Tiling derived from ea_geom.  A "key tile" is 16 keys (one MFMA row block); the key list of a
// window is [nLT local tiles | nCT landmark tiles | dummy tiles up to a multiple of 4]; keys are
// consumed in chunks of 4 tiles (64 keys = two k-steps of the P.V MFMA) with an online softmax.
struct WinTiling {
  int Wq, Wk;          // queries / local keys per window
  int nQT, nLT, nCT;   // 16-row tiles: queries, local keys, landmarks
  int nchunks;         // ceil((nLT + nCT) / 4)
  int nwin;            // windows per (b, h)
  int wpi;             // windows staged per block iteration
  int niter;           // ceil(nwin / wpi)
  int nblk;            // blocks per (b, h)
  int ipb;             // iterations per block
  int biasLd;          // nLT * 16
  int rowsLocal, rowsLm, rowsTotal;
  // Backward with overlapping windows (ext > 0): a token is a key of several windows, so dk/dv
  // accumulate.  The windows are split into colour classes (every ncx-th window per axis) such that
  // no two windows of one class share a key; one launch per class, plain read-modify-write inside.
  int ncx, ncy;        // colour classes per axis (1 without overlap); classes = ncx * ncy
  int col_x, col_y;    // class of this launch
  int sub_x;           // windows per row of this class (2-D)
  int blk0;            // offset of this launch's workgroups in the *_part buffers
  int parts_total;     // workgroups per (b,h) summed over the classes (leading dim of *_part)
  int causal;          // ea_geom.causal: left-only extension, query-padding and causal masks
  // Backward of windows too large for one LDS image: the queries of a window are processed in qsplit
  // blocks of Wq (= WqFull / qsplit) rows, one launch each, like colour classes (they share keys).
  int qsplit, qoff, WqFull;   // blocks per window; first query slot of this launch's block; w (or w*w)
  int slice;                  // merged query-block launch: this block's slice of the dk/dv scratch
  int bblk0;                  // offset of this launch's workgroups in dbias_part (query blocks write
                              // disjoint rows, so merged blocks share the slabs: bblk0 = 0)
  int dbd;                    // causal 1-D backward, one window per workgroup: the bias-gradient entries are stored straight
                              // to dbias_part (each is produced exactly once) -- no [Wq][ld] fp32 accumulator in LDS, so a
                              // 128-token window at D = 128 needs 2 query blocks instead of 4 (round 5)
  int cdirect;                // 1-D windows extended by HALF a window on both sides (2 e = w, two colour classes): the
                              // even windows' key ranges tile the sequence (but its last e tokens) and so do the odd
                              // ones' (but the first e): class 0 STORES dk / dv in the I/O dtype, class 1 adds to them
                              // -- no fp32 scratch slices, no finish pass (round 3)
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// CUs of the current device (cached; 256 = MI355X when no device is visible, e.g. build checks)
inline int device_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    else
      n = 256;
  }
  return n;
}

// LDS image of the backward kernel (ea_window_bwd.hip) for a tiling
inline size_t window_bwd_lds(const WinTiling& t, int D, bool bias, bool bias_lds) {
  const int nQTe = (t.nQT + 1) & ~1;
  const size_t rowsQ = (size_t)t.wpi * nQTe * 16;
  size_t b = (size_t)t.rowsTotal * D * 2 * 2 + rowsQ * D * 2 * 2 + rowsQ * 4 * 2;
  if (bias) b += (size_t)t.Wq * (t.biasLd + 1) * 4 * ((t.dbd ? 0 : 1) + (bias_lds ? 1 : 0));
  b += (size_t)t.rowsTotal * 8 + (size_t)(t.nLT * 16 + nQTe * 16) * 4 + 128 * 4;
  if (t.causal) b += rowsQ * 8;                       // per-query visibility limits
  return b;
}
constexpr size_t WIN_LDS_MAX = 160 * 1024;

// Workgroups per (b,h) for t.niter iterations.  A workgroup pays a fixed prologue (landmark rows,
// bias table, slot tables: ~0.4 window-iterations, measured) and keeps the landmark rows -- in
// backward also its landmark-gradient accumulators -- resident across its windows, so fewer
// workgroups are cheaper; but the launch runs in rounds of (CUs x resident workgroups per CU), and a
// partly filled last round idles the chip (B*h = 384, 6 workgroups each: 4.5 rounds on 512 slots
// -> 5).  Pick the count that minimises rounds x (windows per workgroup + prologue).
// Backward (round 6, workgroup timelines of the -DEA_PROFILE build at cfg3: 19.3 k cycles of staging before the first window,
// 16.5 k per window): the prologue is ~1.1 window-iterations, not the 0.4 measured for the forward in round 1 -- with 0.4 the
// N = 196 backward (4 windows per (b,h), B*h = 384) was cut into one-window workgroups, 3 rounds of (prologue + 1 window)
// = 49 us where one 4-window workgroup per (b,h) takes one round of (prologue + 4 windows).  EA_WIN_BWD_PROLOGUE: dev knob.
inline double win_bwd_prologue() {
  static const double v = [] { const char* e = getenv("EA_WIN_BWD_PROLOGUE"); return e ? atof(e) : 1.1; }();
  return v;
}
// Forward (sweep of round 6 on one box, EVA cfg3 / cfg2, local cfg3, PvT stages, cfg5): 1.2 with landmark rows to stage (EVA cfg3
// forward 50 -> 47 us), 0.4 without (local: 36 us at 0.4, 44 at 1.2).  EA_WIN_FWD_PROLOGUE: dev knob.
inline double win_fwd_prologue(int L) {
  static const double v = [] { const char* e = getenv("EA_WIN_FWD_PROLOGUE"); return e ? atof(e) : -1.0; }();
  return v >= 0 ? v : (L > 0 ? 1.2 : 0.4);
}
inline void win_blocks(const ea_geom& g, WinTiling& t, bool backward) {
  if (backward && t.dbd) { t.ipb = 1; t.nblk = t.niter; return; }    // (direct bias gradient: one iteration per workgroup)
  const long bh = (long)g.B * g.H;
  // resident workgroups per CU: the kernels' launch bounds, or one when the LDS image takes over
  // half of the CU's 160 KB
  int per_cu = backward ? (g.D == 128 ? 1 : 2) : (g.D == 128 ? 2 : 4);
  const size_t lds = backward ? window_bwd_lds(t, g.D, false, false) : (size_t)t.rowsTotal * g.D * 4;
  if (lds > WIN_LDS_MAX / 2) per_cu = 1;
  const long slots = (long)device_cu_count() * per_cu;
  int best = 1;
  double best_cost = 1e30;
  for (int nb = 1; nb <= t.niter; ++nb) {
    const int ipb = ceil_div(t.niter, nb);
    if (ceil_div(t.niter, ipb) != nb) continue;                        // same schedule as a smaller count
    const long rounds = (bh * nb + slots - 1) / slots;
    const double cost = (double)rounds * (ipb + (backward ? win_bwd_prologue() : win_fwd_prologue(g.L)));
    if (cost < best_cost - 1e-9) { best_cost = cost; best = nb; }
  }
  t.ipb = ceil_div(t.niter, best);
  t.nblk = ceil_div(t.niter, t.ipb);
}

// Everything that follows from (Wq, Wk, nwin): tile counts, windows per iteration, LDS rows, blocks.
// Backward: fewer windows per iteration when the image would not fit (sized with a bias table read
// from global memory, the larger-geometry mode of the kernel).
inline void win_derive(const ea_geom& g, WinTiling& t, bool backward) {
  t.nQT = ceil_div(t.Wq, 16);
  t.nLT = ceil_div(t.Wk, 16);
  t.nCT = ceil_div(g.L, 16);
  t.nchunks = ceil_div(t.nLT + t.nCT, 4);
  t.wpi = (t.nQT >= 3 || (backward && t.dbd)) ? 1 : (t.nQT == 2 ? 2 : 4);
  if (t.wpi > t.nwin) t.wpi = t.nwin;
  for (;;) {
    t.rowsLocal = t.wpi * t.nLT * 16;
    t.rowsLm = t.nCT * 16;
    t.rowsTotal = t.rowsLocal + t.rowsLm + 16;
    if (!backward || t.wpi == 1 || window_bwd_lds(t, g.D, true, false) <= WIN_LDS_MAX) break;
    t.wpi /= 2;
  }
  t.niter = ceil_div(t.nwin, t.wpi);
  win_blocks(g, t, backward);
}

// Restrict a backward tiling to one launch: colour class (cy, cx) of the windows and query block qb
// of every window.  Returns false when the class is empty.  With the causal masks, query block qb
// sees no key beyond its own last query, so the key list of the launch ends there.
inline bool win_sub(const ea_geom& g, WinTiling& t, int cy, int cx, int qb) {
  const int w = g.window;
  const int WX = g.attn_2d ? g.gw / w : ceil_div(g.N, w), WY = g.attn_2d ? g.gh / w : 1;
  if (cx >= WX || cy >= WY) return false;
  const int sx = ceil_div(WX - cx, t.ncx), sy = ceil_div(WY - cy, t.ncy);
  t.col_x = cx; t.col_y = cy; t.sub_x = sx;
  t.nwin = sx * sy;
  t.qoff = qb * t.Wq;
  if (t.qsplit > 1 && g.causal == 2) t.Wk = min(t.Wk, g.ext + (qb + 1) * t.Wq);
  win_derive(g, t, true);
  return true;
}

// one launch covers everything: no overlap between windows and the whole window in one query block
__host__ __device__ inline bool win_bwd_single(const WinTiling& t) { return t.ncx * t.ncy * t.qsplit == 1; }
// Query blocks of non-overlapping windows share keys only with each other: given one dk/dv scratch
// slice per block they need no ordering and run as ONE launch (more workgroups in flight, the long
// blocks first), and the slices are summed afterwards.
inline bool win_bwd_merged(const WinTiling& t) {
  return t.causal && t.qsplit > 1 && t.qsplit <= 4 && t.ncx * t.ncy == 1;
}
// fp32 [B,H,N,D] scratch slices the backward needs for dk and for dv
// The backward runs as one launch per (colour class, query block); f(tiling) for each of them, in
// launch order, with blk0 = the launch's offset in the per-workgroup partial buffers.
template <typename F>
inline int win_bwd_launches(const ea_geom& g, const WinTiling& base, F&& f) {
  int blk0 = 0;
  for (int cy = 0; cy < base.ncy; ++cy)
    for (int cx = 0; cx < base.ncx; ++cx)
      for (int qb = 0; qb < base.qsplit; ++qb) {
        WinTiling c = base;
        if (!win_sub(g, c, cy, cx, qb)) continue;
        c.blk0 = blk0;
        c.bblk0 = win_bwd_merged(base) ? 0 : blk0;
        blk0 += c.nblk;
        f(c);
      }
  return blk0;
}
// leading dimension of dbias_part
inline int win_bwd_bias_parts(const ea_geom& g, const WinTiling& t);
// Overlapping 1-D windows: a token is a key of up to ncx consecutive windows, one per colour class.
// Each class stores into its own scratch slice (plain stores: a read-modify-write is a dependent
// global round trip per key tile, 20 % of the kernel at N = 4096) and the finish pass sums the
// slices of the windows that cover a token.
inline bool win_bwd_colour_slices(const WinTiling& t) { return t.qsplit == 1 && t.ncy == 1 && t.ncx > 1 && !t.cdirect; }
// fp32 [B,H,N,D] scratch slices the backward needs for dk and for dv
inline int win_bwd_acc_slices(const WinTiling& t) {
  if (win_bwd_single(t) || t.cdirect) return 0;
  if (win_bwd_merged(t)) return t.qsplit;
  return win_bwd_colour_slices(t) ? t.ncx : 1;
}

inline int win_bwd_bias_parts(const ea_geom& g, const WinTiling& t) {
  if (!win_bwd_merged(t)) return t.parts_total;
  int m = 0;
  win_bwd_launches(g, t, [&](const WinTiling& c) { m = c.nblk > m ? c.nblk : m; });
  return m;
}

// dev switch: EA_WIN_DBD=0 keeps the LDS bias-gradient accumulator (and the 4 query blocks of the LM geometry)
inline bool win_dbd_on() {
  static const bool v = [] { const char* e = getenv("EA_WIN_DBD"); return !e || atoi(e) != 0; }();
  return v;
}
// dev switch: EA_WIN_CDIRECT=0 keeps the scratch slices + finish pass for half-window overlap
inline bool win_cdirect_on() {
  static const bool v = [] { const char* e = getenv("EA_WIN_CDIRECT"); return !e || atoi(e) != 0; }();
  return v;
}
inline int win_tiling(const ea_geom& g, WinTiling& t, bool backward) {
  if (g.window <= 0 || g.D <= 0 || g.B <= 0 || g.H <= 0 || g.N <= 0) return EA_E_BADARG;
  if (g.causal < 0 || g.causal > 2 || (g.causal && (g.attn_2d || g.N % g.window))) return EA_E_BADARG;
  if (g.causal == 2 && g.L > 0 && g.chunk <= 0) return EA_E_BADARG;
  const int w = g.window, e = g.ext;
  const int kext = g.causal ? e : 2 * e;             // keys beyond the window's own tokens
  t.causal = g.causal;
  if (g.attn_2d) {
    if (g.gh <= 0 || g.gw <= 0 || g.gh * g.gw != g.N || g.gh % w || g.gw % w) return EA_E_BADARG;
    t.Wq = w * w;
    t.Wk = (w + 2 * e) * (w + 2 * e);
    t.nwin = (g.gh / w) * (g.gw / w);
  } else {
    t.Wq = w;
    t.Wk = w + kext;
    t.nwin = ceil_div(g.N, w);
  }
  t.WqFull = t.Wq;
  t.biasLd = ceil_div(t.Wk, 16) * 16;
  t.qsplit = 1; t.qoff = 0; t.slice = 0; t.bblk0 = 0; t.cdirect = 0; t.dbd = 0;
  t.ncx = t.ncy = 1;
  t.col_x = t.col_y = 0; t.sub_x = 0; t.blk0 = 0;
  win_derive(g, t, backward);
  if (backward && !g.attn_2d) {
    // a window whose rows do not fit the LDS image is processed in query blocks (halved until it
    // fits; a block keeps whole 32-query MFMA steps), one launch per block
    while (window_bwd_lds(t, g.D, true, false) > WIN_LDS_MAX && t.Wq % 64 == 0) {
      if (!t.dbd && g.causal && t.wpi == 1 && win_dbd_on()) {
        // first give up the bias-gradient accumulator (one window per workgroup, entries stored directly)
        t.dbd = 1;
        win_derive(g, t, true);
        continue;
      }
      t.qsplit *= 2;
      t.Wq /= 2;
      win_derive(g, t, true);
    }
  }
  t.parts_total = t.nblk;
  if (backward && (kext > 0 || t.qsplit > 1)) {
    if (kext > 0) {
      t.ncx = 1 + ceil_div(kext, w);
      t.ncy = g.attn_2d ? t.ncx : 1;
    }
    t.cdirect = (!g.attn_2d && !g.causal && t.qsplit == 1 && t.ncx == 2 && 2 * e == w && g.N % w == 0 && win_cdirect_on()) ? 1 : 0;
    t.parts_total = win_bwd_launches(g, t, [](const WinTiling&) {});
  }
  return EA_OK;
}

// static geometry of the window kernels: tile counts as template constants (0 = read the tiling at run time)
// HO (backward): phase A hands P and dS to phase B through LDS instead of phase B recomputing them (ea_window_bwd.hip)
// PW (with HO): launches without a padding mask whose windows are all complete -- key validity is static
struct SGdyn { static constexpr int NQT = 0, NLT = 0, NCT = 0, WPI = 0; static constexpr bool HO = false, PW = false; };
template <int a, int b, int c, int d, bool ho = false, bool pw = false> struct SGs {
  static constexpr int NQT = a, NLT = b, NCT = c, WPI = d;
  static constexpr bool HO = ho, PW = pw;
};

struct T4 {
  char* p;
  int64_t sb, sh, sn;
};
inline T4 mk(const ea_t4* t) {
  T4 r;
  r.p = t ? (char*)t->ptr : nullptr;
  r.sb = t ? t->sb : 0; r.sh = t ? t->sh : 0; r.sn = t ? t->sn : 0;
  return r;
}

struct WinP {
  T4 q, k, v, o;            // o = out (fwd) / dout (bwd)
  T4 dq, dk, dv;            // bwd only
  const float *lk, *lv, *bias;
  const uint8_t* mask;
  float* lse;
  const float* dlse;                         // bwd, optional: gradient of the returned lse [B,H,N]
  float *dlk_part, *dlv_part, *dbias_part;   // bwd only
  float *dk32, *dv32;                        // bwd, overlap (e > 0): fp32 atomics scratch [B,H,N,D]
  Geo G;
  int B, H, L, w, e;
  int causal, chunk;                         // ea_geom.causal and the landmark chunk length (causal masks)
  int lm_base;                               // ea_geom.lm_base: landmarks before the call's first token (decoding)
  float scale, scale_log2;
  WinTiling t;
  // attention dropout (causal_eva.py:778): keep mask [B,H,N,keep_ld] u8 over the softmax columns of a
  // query -- local slot j at column j, landmark c at column biasLd + c -- or nullptr; kept
  // probabilities are scaled by keep_scale = 1/(1-p)
  const uint8_t* keep;
  int keep_ld;
  float keep_scale;
  int bias_lds;                              // bwd: the bias table of the head is staged in LDS
  int plain;                                 // no padding mask and every window complete: key validity is static (round 3)
  // bwd, how local dk/dv leave the kernel: 0 = stored to dk/dv; 1 = fp32 read-modify-write into dk32/dv32
  // (launches ordered by the stream); 2 / 3 = plain fp32 stores into slice t.slice of dk32/dv32 (2: one
  // slice per query block, 3: one per colour class -- they differ in the finish pass)
  int acc_mode;
  // bwd, merged query-block launch: workgroups [qstart[i], qstart[i+1]) run tiling tv[i] (nq > 1)
  int nq;
  int qstart[5];
  WinTiling tv[4];
  long long* prof;                           // dev builds (-DEA_PROFILE): phase time stamps
};

// ---- slot tables: the (window-independent) offset of every key / query slot, built once per
// workgroup in LDS so that staging needs no integer division per row.
//   2-D: packed (dy & 0xffff) | (dx << 16) relative to the window origin; 1-D: offset from it.
EA_DEV void build_slot_tables(int* kd, int* qd, const WinTiling& t, const Geo& G, int w, int e, int nq, int tid) {
  const int tt = w + 2 * e;
  for (int s = tid; s < t.nLT * 16; s += 256) {
    const int i = s / tt, j = s - i * tt;
    kd[s] = G.attn2d ? (((i - e) & 0xffff) | ((j - e) << 16)) : (s - e);
  }
  for (int s = tid; s < nq; s += 256) {
    const int i = s / w, j = s - i * w;
    qd[s] = G.attn2d ? ((i & 0xffff) | (j << 16)) : s + t.qoff;
  }
}
// window origin (oy, ox) in 2-D, (first token, 0) in 1-D
EA_DEV void win_origin(const Geo& G, int win, int w, int& oy, int& ox) {
  if (G.attn2d) {
    const int per_row = G.gw / w;
    const int wy = win / per_row;
    oy = wy * w;
    ox = (win - wy * per_row) * w;
  } else {
    oy = win * w;
    ox = 0;
  }
}
// window id of logical window `lw` of this launch's colour class
EA_DEV int colour_win(const WinTiling& t, const Geo& G, int w, int lw) {
  if (t.ncx == 1 && t.ncy == 1) return lw;
  if (!G.attn2d) return lw * t.ncx + t.col_x;
  const int sy = lw / t.sub_x, sx = lw - sy * t.sub_x;
  return (sy * t.ncy + t.col_y) * (G.gw / w) + sx * t.ncx + t.col_x;
}
// causal_eva.py visibility limits of one query: the last visible local key slot and the last visible
// landmark.  A padded query sees no local key (:742-755); with the causal masks, query slot i sees
// local slots j <= i + e (:767-773) and the landmarks of the chunks before its own (:716-738).
struct QLim { int local, lm; };
EA_DEV QLim query_limits(int causal, int qslot, int qtok, int e, int chunk, const uint8_t* mrow, int lm_base = 0) {
  QLim r;
  r.local = 0x7fffffff; r.lm = 0x7fffffff;
  if (causal >= 2) {
    r.local = qslot + e;
    r.lm = (qtok >= 0 && chunk > 0) ? lm_base + qtok / chunk - 1 : -1;
  }
  if (qtok < 0 || (mrow && mrow[qtok])) r.local = -1;
  return r;
}
EA_DEV int slot_token(const Geo& G, int packed, int oy, int ox) {
  if (G.attn2d) {
    const int y = oy + (int)(short)(packed & 0xffff), x = ox + (packed >> 16);
    return (y >= 0 && y < G.gh && x >= 0 && x < G.gw) ? y * G.gw + x : -1;
  }
  const int tok = oy + packed;
  return (tok >= 0 && tok < G.N) ? tok : -1;
}

}  // namespace ea
