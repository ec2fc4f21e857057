// ea_capi.hip -- extern "C" entry points of libea_hip.so (declared in include/ea_hip.h).
// Argument validation and launch-parameter construction only; every kernel lives in its own
// translation unit and is reached through a *_dispatch function.
#include <stdlib.h>
#include "ea_window.h"

namespace ea {
int window_fwd_dispatch(const WinP& p, int dtype, int D, hipStream_t st);
int window_bwd_dispatch(const WinP& p, const ea_geom& geom, const T4& outp, const float* biasT, int dtype, int D,
                        hipStream_t st);
}  // namespace ea
#include "ea_landmark_params.h"
#include "ea_lara.h"
#include "ea_softmax.h"
#include "ea_lara_lmk.h"
#include "ea_lara_merge.h"
#include "ea_layernorm.h"
#include "ea_lara_segment.h"
#include "ea_scatter.h"
#include "ea_rows_mlp.h"
#include "ea_performer_f32.h"
#include "ea_f32_attn.h"
namespace ea {
int rows_mlp_dispatch(const RowsP& p, int D, int sides, int layer_norm, bool bwd, hipStream_t st);
int rows_mlp_blocks(int R, int D);
int lara_x_dispatch(int mode, const LaraP& p, int dtype, hipStream_t st);
int lara_y_dispatch(int mode, const LaraP& p, int dtype, hipStream_t st);
int lara_f_dispatch(int which, const LaraP& p, int dtype, hipStream_t st);
int pool2d_dispatch(bool bwd, int dtype, const void* x, long sb, long sh, long sn, float* mean, int B, int H, int gh, int gw,
                    int side, int D, hipStream_t st);
int linear_supported(int K, int NO);
int proj_rs_supported(int K, int NO);
int proj_rs_pool_supported(int K, int NO, int B, int gh, int gw, int r);
int proj_rs_dispatch(int dtype, const void* a, int a_f32, const float* w, const float* bias, void* y, void* a_cast, int rows,
                     long lda, long ldy, hipStream_t st, int B, int gh, int gw, int r, float* pq, float* pk, void* w_cast,
                     const void* wsw);
int w192_prepare_dispatch(int dtype, const float* wq, const float* wp, void* w16q, void* wsw, void* w16p, void* w16pT, hipStream_t st);
int dgrad_rs_supported(int K, int NO);
int dgrad_fin_launch(int dtype, const void* dqkv, long ldy, const void* qkv, long ldq, const void* w, int w_f32, void* dx, int dx_f32,
                     long ldx, int B, int gh, int gw, int pool_r, int C, float scale, const float* qbar, const float* uq,
                     const float* lse_t, const float* dpq, const float* dpk, hipStream_t st);
int dgrad_rs_dispatch(int dtype, const void* dy, const void* w, int w_f32, void* dx, int dx_f32, int rows, long ldy, long ldx,
                      hipStream_t st);
int linear_dispatch(int dtype, const void* a, int a_f32, const void* w, int w_mode, const float* bias, void* y, int y_f32,
                    void* a_cast, int rows, int K, int NO, long lda, long ldy, hipStream_t st);
int wgrad_slices(int rows, int M, int K);
int wgrad_pair_slices(int rows, int M1, int K1, int M2, int K2);
int wgrad_pair_dispatch(int dtype, int rows, const void* dy1, const void* x1, float* part1, float* db1, long ld1, int M1, int K1,
                        const void* dy2, const void* x2, float* part2, float* db2, long ld2, int M2, int K2, hipStream_t st);
int wgrad_dispatch(int dtype, const void* dy, const void* x, float* part, float* db_part, long part_ld, int rows, int M,
                   int K, hipStream_t st);
int part_sum_dispatch(const float* part, float* out, int S, int n, long ld, hipStream_t st);
int multi_sum_dispatch(int K, const float* const* part, const int* S, const int* n, const long long* ld, float* const* out,
                       hipStream_t st);
}

using namespace ea;

static bool t4_ok(const ea_t4* t, int D) {
  // 16-byte vector access: base and every stride must be a multiple of 8 elements
  return t && t->ptr && ((uintptr_t)t->ptr % 16 == 0) && t->sb % 8 == 0 && t->sh % 8 == 0 &&
         t->sn % 8 == 0 && t->sn >= D;
}
// kernels that form token * stride in 32-bit arithmetic additionally need N * sn < 2^31 elements
static bool t4_ok32(const ea_t4* t, int D, int N) {
  return t4_ok(t, D) && (int64_t)N * t->sn < ((int64_t)1 << 31);
}
static bool geom_ok(const ea_geom* g) {
  return g && g->B > 0 && g->H > 0 && g->N > 0 && (g->D == 32 || g->D == 64 || g->D == 128) &&
         (g->dtype == EA_BF16 || g->dtype == EA_F16) && g->ext >= 0 && g->causal >= 0 && g->causal <= 2 && g->lm_base >= 0;
}
static Geo mk_geo(const ea_geom* g) {
  Geo G;
  G.N = g->N; G.attn2d = g->attn_2d; G.gh = g->attn_2d ? g->gh : 1; G.gw = g->attn_2d ? g->gw : g->N;
  return G;
}

extern "C" {

const char* ea_version(void) { return "ea_hip 0.1.0 gfx950"; }
int32_t ea_abi_version(void) { return 15; }

int32_t ea_window_bias_ld(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, false) != EA_OK) return EA_E_BADARG;
  return t.biasLd;
}
int32_t ea_window_keep_ld(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, false) != EA_OK) return EA_E_BADARG;
  return t.biasLd + t.nCT * 16;
}
int32_t ea_window_bwd_parts(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, true) != EA_OK) return EA_E_BADARG;
  return t.parts_total;
}
int32_t ea_window_bwd_needs_bias_t(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, true) != EA_OK) return EA_E_BADARG;
  int any = 0;                                             // a launch reads the bias from global memory
  win_bwd_launches(*g, t, [&](const WinTiling& tl) { any |= window_bwd_lds(tl, g->D, true, true) > WIN_LDS_MAX; });
  return any;
}
int32_t ea_window_bwd_bias_parts(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, true) != EA_OK) return EA_E_BADARG;
  return win_bwd_bias_parts(*g, t);
}
int32_t ea_window_bwd_acc_slices(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, true) != EA_OK) return EA_E_BADARG;
  return win_bwd_acc_slices(t);
}
int32_t ea_window_bwd_query_blocks(const ea_geom* g) {
  WinTiling t;
  if (!geom_ok(g) || win_tiling(*g, t, true) != EA_OK) return EA_E_BADARG;
  return t.qsplit;
}

static int fill_win(const ea_geom* g, WinP& p, bool backward) {
  if (!geom_ok(g)) return EA_E_BADARG;
  int rc = win_tiling(*g, p.t, backward);
  if (rc != EA_OK) return rc;
  if (g->L < 0 || g->L > 64) return EA_E_UNSUPPORTED;      // landmark tiles owned 1:1 by 4 waves
  p.G = mk_geo(g);
  p.B = g->B; p.H = g->H; p.L = g->L; p.w = g->window; p.e = g->ext;
  p.causal = g->causal; p.chunk = g->chunk; p.lm_base = g->causal == 2 ? g->lm_base : 0;
  p.scale = g->scale;
  p.scale_log2 = g->scale * LOG2E;
  return EA_OK;
}

int ea_window_attn_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                       const float* lk, const float* lv, const float* bias, const uint8_t* mask,
                       const ea_t4* out, float* lse, const uint8_t* keep, float keep_scale, void* stream) {
  WinP p = {};
  int rc = fill_win(g, p, false);
  if (rc != EA_OK) return rc;
  if (keep && !g->causal) return EA_E_UNSUPPORTED;
  p.keep = keep; p.keep_scale = keep_scale; p.keep_ld = p.t.biasLd + p.t.nCT * 16;
  if (!t4_ok(q, g->D) || !t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(out, g->D) || !lse) return EA_E_BADARG;
  if (g->L > 0 && (!lk || !lv)) return EA_E_BADARG;
  p.q = mk(q); p.k = mk(k); p.v = mk(v); p.o = mk(out);
  p.lk = lk; p.lv = lv; p.bias = bias; p.mask = mask; p.lse = lse;
  return window_fwd_dispatch(p, g->dtype, g->D, (hipStream_t)stream);
}

int ea_window_attn_bwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                       const float* lk, const float* lv, const float* bias, const uint8_t* mask,
                       const ea_t4* out, const ea_t4* dout, const float* lse,
                       const ea_t4* dq, const ea_t4* dk, const ea_t4* dv,
                       float* dlk_part, float* dlv_part, float* dbias_part,
                       float* dk_acc, float* dv_acc, const float* bias_t,
                       const uint8_t* keep, float keep_scale, const float* dlse, void* stream) {
  WinP p = {};
  int rc = fill_win(g, p, true);
  if (rc != EA_OK) return rc;
  p.dlse = dlse;
  if (keep && !g->causal) return EA_E_UNSUPPORTED;
  p.keep = keep; p.keep_scale = keep_scale; p.keep_ld = p.t.biasLd + p.t.nCT * 16;
  const int N = g->N;
  if (!t4_ok32(q, g->D, N) || !t4_ok32(k, g->D, N) || !t4_ok32(v, g->D, N) || !t4_ok32(dout, g->D, N) ||
      !t4_ok32(out, g->D, N) || (win_bwd_acc_slices(p.t) > 0 && (!dk_acc || !dv_acc)) || !t4_ok32(dq, g->D, N) ||
      !t4_ok32(dk, g->D, N) || !t4_ok32(dv, g->D, N) || !lse) return EA_E_BADARG;
  if (g->L > 0 && (!lk || !lv || !dlk_part || !dlv_part)) return EA_E_BADARG;
  if (bias && (!dbias_part || (!bias_t && ea_window_bwd_needs_bias_t(g) != 0))) return EA_E_BADARG;
  p.q = mk(q); p.k = mk(k); p.v = mk(v); p.o = mk(dout);
  p.dq = mk(dq); p.dk = mk(dk); p.dv = mk(dv);
  p.lk = lk; p.lv = lv; p.bias = bias; p.mask = mask; p.lse = const_cast<float*>(lse);
  p.dlk_part = dlk_part; p.dlv_part = dlv_part; p.dbias_part = dbias_part;
  p.dk32 = dk_acc; p.dv32 = dv_acc;
  return window_bwd_dispatch(p, *g, mk(out), bias ? bias_t : nullptr, g->dtype, g->D, (hipStream_t)stream);
}

// ---- EVA landmark statistics ----
static int fill_lm(const ea_geom* g, LmP& p) {
  if (!geom_ok(g) || g->chunk <= 0 || g->L <= 0) return EA_E_BADARG;
  p.G = mk_geo(g);
  if (g->attn_2d && (g->gh * g->gw != g->N)) return EA_E_BADARG;
  if (g->causal && (g->attn_2d || g->N % g->chunk)) return EA_E_BADARG;
  const int ext = g->causal ? 0 : g->ext;                  // causal_eva.py:688-694: chunks are not extended
  const int side = g->chunk + 2 * ext;
  const int nchunks = g->attn_2d ? (g->gh / g->chunk) * (g->gw / g->chunk) : g->N / g->chunk;
  if (nchunks != g->L) return EA_E_BADARG;
  p.B = g->B; p.H = g->H; p.L = g->L; p.r = g->chunk; p.e = ext;
  p.J = g->attn_2d ? side * side : side;
  p.scale = g->scale;
  return EA_OK;
}
#define SET3(dst, src) do { p.dst = (char*)(src)->ptr; p.dst##_sb = (src)->sb; p.dst##_sh = (src)->sh; p.dst##_sn = (src)->sn; } while (0)

int ea_eva_chunk_mean_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const uint8_t* mask,
                          float* qmean, float* kmean, void* stream) {
  LmP p = {};
  int rc = fill_lm(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(k, g->D) || !qmean || !kmean) return EA_E_BADARG;
  SET3(q, q); SET3(k, k);
  p.mask = mask; p.qmean = qmean; p.kmean = kmean;
  return landmark_dispatch(0, p, g->dtype, g->D, (hipStream_t)stream);
}

int ea_eva_chunk_mean_bwd(const ea_geom* g, const float* dqmean, const float* dkmean,
                          const uint8_t* mask, const ea_t4* dq, const ea_t4* dk, void* stream) {
  LmP p = {};
  int rc = fill_lm(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(dq, g->D) || !t4_ok(dk, g->D) || !dqmean || !dkmean) return EA_E_BADARG;
  SET3(dq, dq); SET3(dk, dk);
  p.mask = mask; p.dqmean = dqmean; p.dkmean = dkmean;
  return landmark_dispatch(1, p, g->dtype, g->D, (hipStream_t)stream);
}

int ea_eva_beta_fwd(const ea_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* omega, float* beta, void* stream) {
  LmP p = {};
  int rc = fill_lm(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !omega || !beta) return EA_E_BADARG;
  SET3(k, k); SET3(v, v);
  p.mask = mask; p.omega = omega; p.beta_out = beta;
  return landmark_dispatch(2, p, g->dtype, g->D, (hipStream_t)stream);
}

// dbeta_S > 1 (composite EVA backward, round 6): dbeta = slice 0 of dbeta_S <= 4 slice partials `dbeta_stride` floats apart
static int eva_beta_bwd_parts(const ea_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                              const float* omega, const float* beta, const float* dbeta, int dbeta_S, long dbeta_stride,
                              const ea_t4* dk, const ea_t4* dv, float* domega, void* stream) {
  LmP p = {};
  int rc = fill_lm(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(dk, g->D) || !t4_ok(dv, g->D) || !omega ||
      !beta || !dbeta || !domega || dbeta_S < 1 || dbeta_S > 4) return EA_E_BADARG;
  SET3(k, k); SET3(v, v); SET3(dk, dk); SET3(dv, dv);
  p.mask = mask; p.omega = omega; p.beta = beta; p.dbeta = dbeta; p.domega = domega;
  p.dbeta_S = dbeta_S; p.dbeta_stride = dbeta_stride;
  return landmark_dispatch(3, p, g->dtype, g->D, (hipStream_t)stream);
}

int ea_eva_beta_bwd(const ea_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* omega, const float* beta, const float* dbeta,
                    const ea_t4* dk, const ea_t4* dv, float* domega, void* stream) {
  return eva_beta_bwd_parts(g, k, v, mask, omega, beta, dbeta, 1, 0, dk, dv, domega, stream);
}

}  // extern "C"

// ---- LARA ----
static T4l mkl(const ea_t4* t) {
  T4l r;
  r.p = t ? (char*)t->ptr : nullptr;
  r.sb = t ? t->sb : 0; r.sh = t ? t->sh : 0; r.sn = t ? t->sn : 0;
  return r;
}
static int lara_nsub(int NCT) { return NCT == 1 ? 4 : (NCT == 2 ? 2 : 1); }
// X passes: blocks per (b,h) and tokens per block; Y passes: sequence splits
static int fill_lara(const ea_lara_geom* g, LaraP& p, bool ypass) {
  if (!g || g->B <= 0 || g->H <= 0 || g->N <= 0 || (g->D != 32 && g->D != 64) || g->C <= 0 ||
      (g->dtype != EA_BF16 && g->dtype != EA_F16) || g->mis < 0 || g->mis > 2) return EA_E_BADARG;
  if (g->C > 128) return EA_E_UNSUPPORTED;
  p.B = g->B; p.H = g->H; p.N = g->N; p.D = g->D; p.C = g->C; p.NCT = (g->C + 15) / 16;
  p.mis = g->mis; p.kappa = g->kappa; p.scale = g->scale; p.scale_log2 = g->scale * LOG2E;
  p.norm_coef2 = 0.5f * g->scale * LOG2E;      // s |k|^2 / 2 in the log2 domain
  p.knorm_coef = g->scale;
  const long bh = (long)g->B * g->H;
  const int gran = ypass ? 128 : 64;
  const int maxblk = (g->N + gran - 1) / gran;
  // X passes re-stage the landmark matrices per workgroup: fewer, longer workgroups (~3 per CU);
  // Y passes only keep landmark fragments in registers: more, shorter slices
  // (Y-pass partial results are [BH, nsplit, C, D] fp32 per accumulator -- as large as the token
  // tensors themselves at 6 slices -- so both passes aim at ~3 workgroups per CU, not more)
  int nblk = (int)((768 + bh - 1) / bh);
  if (nblk < 1) nblk = 1;
  if (nblk > maxblk) nblk = maxblk;
  if (ypass) {
    // the slice boundaries are laid out at launch (lara_y_plan: they depend on how many
    // workgroups of the particular pass fit on the chip); only the count is fixed here
    p.nsplit = nblk > 16 ? 16 : nblk;
    p.tok_per_block = 0;
    return EA_OK;
  }
  int tpb = (g->N + nblk - 1) / nblk;
  tpb = (tpb + gran - 1) / gran * gran;
  p.tok_per_block = tpb;
  p.nsplit = (g->N + tpb - 1) / tpb;
  return EA_OK;
}

extern "C" {

int32_t ea_lara_parts(const ea_lara_geom* g) {
  LaraP p = {};
  if (fill_lara(g, p, true) != EA_OK) return EA_E_BADARG;
  return p.nsplit * lara_nsub(p.NCT);
}

int ea_lara_stats_fwd(const ea_lara_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                      const uint8_t* mask, const float* omega, const float* qbar,
                      float* p_ml, float* p_kv, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(k, g->D) || !t4_ok(v, g->D) || !omega || !p_ml || !p_kv ||
      (g->mis == EA_MIS_OPT && !qbar)) return EA_E_BADARG;
  p.q = mkl(q); p.k = mkl(k); p.v = mkl(v); p.mask = mask; p.omega = omega; p.qbar = qbar;
  p.p_ml = p_ml; p.p_acc0 = p_kv;
  return lara_y_dispatch(LY_FWD, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_out_fwd(const ea_lara_geom* g, const ea_t4* q, const float* omega, const float* qbar,
                    const float* kv, const float* lse_t, const float* bhv, const float* cst,
                    const ea_t4* out, float* lseZ, float* tmean, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(out, g->D) || !omega || !kv || !cst) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !lse_t || !bhv)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar) return EA_E_BADARG;
  if ((lseZ == nullptr) != (tmean == nullptr)) return EA_E_BADARG;
  p.q = mkl(q); p.o = mkl(out); p.omega = omega; p.qbar = qbar; p.kv = kv; p.lse_t = lse_t;
  p.bhv = bhv; p.cst = cst; p.lseZ = lseZ; p.tmean = tmean;
  return lara_x_dispatch(LX_FWD, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_q(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                  const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                  const float* cst, const ea_t4* dq, float* lseZ, float* tmean, float* rowdot,
                  float* sda, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(dout, g->D) || !t4_ok(dq, g->D) || !omega || !kv || !cst ||
      !lseZ || !tmean || !rowdot || !sda) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !lse_t || !bhv)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar) return EA_E_BADARG;
  p.q = mkl(q); p.dout = mkl(dout); p.dq = mkl(dq); p.omega = omega; p.qbar = qbar; p.kv = kv;
  p.lse_t = lse_t; p.bhv = bhv; p.cst = cst;
  p.lseZ = lseZ; p.tmean = tmean; p.rowdot = rowdot; p.sda = sda;
  return lara_x_dispatch(LX_BWDQ, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_qstats(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                       const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                       const float* cst, const float* lseZ, const float* tmean, const float* rowdot,
                       const float* sda, float* p_ml, float* p_dkv, float* p_dom, float* p_m1,
                       float* p_m2, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(dout, g->D) || !omega || !kv || !cst || !lseZ || !tmean ||
      !rowdot || !sda || !p_ml || !p_dkv || !p_dom) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !lse_t || !bhv || !p_m1 || !p_m2)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar) return EA_E_BADARG;
  p.q = mkl(q); p.dout = mkl(dout); p.omega = omega; p.qbar = qbar; p.kv = kv; p.lse_t = lse_t;
  p.bhv = bhv; p.cst = cst;
  p.lseZ = const_cast<float*>(lseZ); p.tmean = const_cast<float*>(tmean);
  p.rowdot = const_cast<float*>(rowdot); p.sda = const_cast<float*>(sda);
  p.p_ml = p_ml; p.p_acc0 = p_dkv; p.p_acc1 = p_dom; p.p_acc2 = p_m1; p.p_acc3 = p_m2;
  return lara_y_dispatch(LY_BWDQ, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_k(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                  const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                  const float* rsum, const ea_t4* dk, const ea_t4* dv, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(dk, g->D) || !t4_ok(dv, g->D) || !omega ||
      !dkv || !lse_k || !dkk || !rsum) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.dk = mkl(dk); p.dv = mkl(dv); p.mask = mask; p.omega = omega;
  p.dkv = dkv; p.lse_k = lse_k; p.dkk = dkk; p.rsum = rsum;
  return lara_x_dispatch(LX_BWDK, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_kstats(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                       const float* rsum, float* p_dom, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !omega || !dkv || !lse_k || !dkk || !rsum || !p_dom)
    return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.mask = mask; p.omega = omega; p.dkv = dkv; p.lse_k = lse_k;
  p.dkk = dkk; p.rsum = rsum; p.p_acc0 = p_dom;
  return lara_y_dispatch(LY_BWDK, p, g->dtype, (hipStream_t)stream);
}

// ---- fused backward (ea_lara_f.hip): one pass per side + one finish pass ----
int32_t ea_lara_fused_parts(const ea_lara_geom* g) {
  LaraP p = {};
  if (fill_lara(g, p, false) != EA_OK) return EA_E_BADARG;
  if (p.NCT > 4) return EA_E_UNSUPPORTED;
  return p.nsplit * (p.NCT <= 2 ? 2 : 1);
}

int ea_lara_bwd_q_fused(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                        const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                        const float* cst, const float* lseZ, const float* tmean, const ea_t4* dq, float* p_ml,
                        float* p_dkv, float* p_dom, float* p_m1, float* p_m2, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (p.NCT > 4) return EA_E_UNSUPPORTED;
  if (!t4_ok(q, g->D) || !t4_ok(dout, g->D) || !t4_ok(dq, g->D) || !omega || !kv || !cst || !p_ml ||
      !p_dkv || !p_dom || !lseZ || !tmean) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !lse_t || !bhv || !p_m1 || !p_m2)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar) return EA_E_BADARG;
  p.q = mkl(q); p.dout = mkl(dout); p.dq = mkl(dq); p.omega = omega; p.qbar = qbar; p.kv = kv;
  p.lse_t = lse_t; p.bhv = bhv; p.cst = cst;
  p.lseZ = const_cast<float*>(lseZ); p.tmean = const_cast<float*>(tmean);
  p.p_ml = p_ml; p.p_acc0 = p_dkv; p.p_acc1 = p_dom; p.p_acc2 = p_m1; p.p_acc3 = p_m2;
  return lara_f_dispatch(0, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_k_fused(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                        const float* rsum, const ea_t4* dk, const ea_t4* dv, float* p_dom, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (p.NCT > 4) return EA_E_UNSUPPORTED;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(dk, g->D) || !t4_ok(dv, g->D) || !omega ||
      !dkv || !lse_k || !dkk || !rsum || !p_dom) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.dk = mkl(dk); p.dv = mkl(dv); p.mask = mask; p.omega = omega;
  p.dkv = dkv; p.lse_k = lse_k; p.dkk = dkk; p.rsum = rsum; p.p_acc0 = p_dom;
  return lara_f_dispatch(1, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_finish(const ea_lara_geom* g, const ea_t4* q, const float* qbar, const float* uq,
                       const float* lse_t, const float* dpq, const float* dpk, int32_t pool_r,
                       int32_t gh, int32_t gw, const ea_t4* dq, const ea_t4* dk, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (p.NCT > 4) return EA_E_UNSUPPORTED;
  if (!t4_ok(dq, g->D)) return EA_E_BADARG;
  const bool has_t = uq != nullptr;
  if (has_t && (g->mis != EA_MIS_OPT || !t4_ok(q, g->D) || !qbar || !lse_t)) return EA_E_BADARG;
  if (pool_r > 0) {
    if (!dpq || !dpk || !t4_ok(dk, g->D) || gh <= 0 || gw <= 0 || gh * gw != g->N || gh % pool_r || gw % pool_r)
      return EA_E_BADARG;
    p.dpq = dpq; p.dpk = dpk; p.pool_r = pool_r; p.pool_gw = gw;
    p.pool_L = (gh / pool_r) * (gw / pool_r);
    p.pool_inv = 1.f / (float)(pool_r * pool_r);
    p.dk = mkl(dk);
  } else if (!has_t) {
    return EA_OK;                                   // nothing to do
  }
  if (has_t) { p.q = mkl(q); p.qbar = qbar; p.uq = uq; p.lse_t = lse_t; }
  p.dq = mkl(dq);
  return lara_f_dispatch(2, p, g->dtype, (hipStream_t)stream);
}

// ---- round 5: consumers that merge the producing pass's slice partials in their prologue (no merge launches) ----
int ea_lara_out_fwd_merge(const ea_lara_geom* g, const ea_t4* q, const float* omega, const float* qbar, const float* bhv,
                          int32_t S, const float* p_ml, const float* p_kv, const float* lp, float* kv, float* lse_k,
                          float* lse_t, float* cst, const ea_t4* out, float* lseZ, float* tmean, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (p.NCT > 4 || S < 1 || S > 4) return EA_E_UNSUPPORTED;
  if (!t4_ok(q, g->D) || !t4_ok(out, g->D) || !omega || !p_ml || !p_kv || !lp || !kv || !lse_k || !cst) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !lse_t || !bhv)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar) return EA_E_BADARG;
  if ((lseZ == nullptr) != (tmean == nullptr)) return EA_E_BADARG;
  p.q = mkl(q); p.o = mkl(out); p.omega = omega; p.qbar = qbar; p.bhv = bhv; p.lseZ = lseZ; p.tmean = tmean;
  p.m_S = S; p.m_ml = p_ml; p.m_acc0 = p_kv; p.m_lp = lp;
  p.m_kv = kv; p.m_lsek = lse_k; p.m_lset = lse_t; p.m_cst = cst;
  return lara_x_dispatch(LX_FWDM, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_k_fused_merge(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* omega,
                              const float* qbar, const float* kv, const float* lse_k, int32_t S, const float* p_ml,
                              const float* p_dkv, const float* p_dom, const float* p_m1, const float* p_m2,
                              const ea_t4* dk, const ea_t4* dv, float* p_domk, float* dbh, float* dlp, float* domq,
                              float* dqbar, float* uq, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (p.NCT > 4 || S < 1 || S > 4) return EA_E_UNSUPPORTED;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(dk, g->D) || !t4_ok(dv, g->D) || !omega || !kv || !lse_k || !p_ml ||
      !p_dkv || !p_dom || !p_domk || !dlp || !domq) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar || !p_m1 || !p_m2 || !dqbar || !uq || !dbh)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !dqbar) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.dk = mkl(dk); p.dv = mkl(dv); p.mask = mask; p.omega = omega; p.qbar = qbar;
  p.kv = kv; p.lse_k = lse_k; p.p_acc0 = p_domk;
  p.m_S = S; p.m_ml = p_ml; p.m_acc0 = p_dkv; p.m_acc1 = p_dom; p.m_acc2 = p_m1; p.m_acc3 = p_m2;
  p.m_dbh = dbh; p.m_dlp = dlp; p.m_domq = domq; p.m_dqbar = dqbar; p.m_uq = uq;
  return lara_f_dispatch(3, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_bwd_qcorr(const ea_lara_geom* g, const ea_t4* q, const float* qbar, const float* uq,
                      const float* lse_t, const ea_t4* dq, void* stream) {
  LaraP p = {};
  int rc = fill_lara(g, p, false);
  if (rc != EA_OK) return rc;
  if (g->mis != EA_MIS_OPT) return EA_E_BADARG;
  if (!t4_ok(q, g->D) || !t4_ok(dq, g->D) || !qbar || !uq || !lse_t) return EA_E_BADARG;
  p.q = mkl(q); p.dq = mkl(dq); p.qbar = qbar; p.uq = uq; p.lse_t = lse_t;
  return lara_x_dispatch(LX_QCORR, p, g->dtype, (hipStream_t)stream);
}

}  // extern "C"

// ---- softmax baseline ----
#define SM_SET(dst, src) do { if (src) { p.dst.p = (char*)(src)->ptr; p.dst.sb = (src)->sb; p.dst.sh = (src)->sh; p.dst.sn = (src)->sn; } } while (0)
static int fill_sm(int B, int H, int N, int D, int dtype, float scale, SmP& p) {
  if (B <= 0 || H <= 0 || N <= 0 || (D != 32 && D != 64 && D != 128) || (dtype != EA_BF16 && dtype != EA_F16))
    return EA_E_BADARG;
  p.B = B; p.H = H; p.N = N; p.scale = scale; p.scale_log2 = scale * LOG2E;
  return EA_OK;
}

extern "C" {

int ea_softmax_attn_fwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                        const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const ea_t4* out, float* lse, const uint8_t* keep, float keep_scale,
                        int32_t key_norm_bias, void* stream) {
  SmP p = {};
  int rc = fill_sm(B, H, N, D, dtype, scale, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, D) || !t4_ok(k, D) || !t4_ok(v, D) || !t4_ok(out, D) || !lse) return EA_E_BADARG;
  if (key_norm_bias && keep) return EA_E_UNSUPPORTED;
  p.key_norm_bias = key_norm_bias;
  p.keep = keep; p.keep_scale = keep_scale; p.keep_ld = (N + 63) / 64 * 64;
  SM_SET(q, q); SM_SET(k, k); SM_SET(v, v); SM_SET(o, out);
  p.mask = mask; p.lse = lse;
  return softmax_dispatch(0, p, dtype, D, (hipStream_t)stream);
}

int ea_softmax_attn_bwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                        const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const ea_t4* out, const ea_t4* dout, const float* lse, float* delta,
                        const ea_t4* dq, const ea_t4* dk, const ea_t4* dv,
                        const uint8_t* keep, float keep_scale, int32_t key_norm_bias, void* stream) {
  SmP p = {};
  int rc = fill_sm(B, H, N, D, dtype, scale, p);
  if (rc != EA_OK) return rc;
  if (key_norm_bias && keep) return EA_E_UNSUPPORTED;
  p.key_norm_bias = key_norm_bias;
  p.keep = keep; p.keep_scale = keep_scale; p.keep_ld = (N + 63) / 64 * 64;
  if (!t4_ok(q, D) || !t4_ok(k, D) || !t4_ok(v, D) || !t4_ok(out, D) || !t4_ok(dout, D) ||
      !t4_ok(dq, D) || !t4_ok(dk, D) || !t4_ok(dv, D) || !lse || !delta) return EA_E_BADARG;
  SM_SET(q, q); SM_SET(k, k); SM_SET(v, v); SM_SET(o, out); SM_SET(dout, dout);
  SM_SET(dq, dq); SM_SET(dk, dk); SM_SET(dv, dv);
  p.mask = mask; p.lse = const_cast<float*>(lse); p.delta = delta;
  return softmax_dispatch(1, p, dtype, D, (hipStream_t)stream);
}

int ea_softmax_sample(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                      const ea_t4* q, const ea_t4* k, const uint64_t* seed, int64_t* index, void* stream) {
  SmP p = {};
  int rc = fill_sm(B, H, N, D, dtype, scale, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, D) || !t4_ok(k, D) || !seed || !index) return EA_E_BADARG;
  SM_SET(q, q); SM_SET(k, k);
  p.seed = reinterpret_cast<const unsigned long long*>(seed);
  p.sample_out = reinterpret_cast<long long*>(index);
  return softmax_dispatch(2, p, dtype, D, (hipStream_t)stream);
}

}  // extern "C"

// ---- Performer (FAVOR+) on the LARA skeletons ----
static int fill_perf(const ea_perf_geom* g, LaraP& p, bool ypass) {
  if (!g) return EA_E_BADARG;
  ea_lara_geom lg;
  lg.B = g->B; lg.H = g->H; lg.N = g->N; lg.D = g->D; lg.dtype = g->dtype; lg.C = g->M;
  lg.mis = EA_MIS_BH; lg.kappa = 0.f;
  lg.scale = 1.f / sqrtf(sqrtf((float)g->D));            // data_normalizer d^-1/4
  int rc = fill_lara(&lg, p, ypass);
  if (rc != EA_OK) return rc;
  p.w_per_head = 1;
  p.norm_coef2 = 0.5f * lg.scale * lg.scale * LOG2E;      // d^-1/2 |x|^2 / 2
  p.knorm_coef = lg.scale * lg.scale;
  p.ratio = 1.f / sqrtf((float)g->M);
  p.feps = 1e-4f;
  return EA_OK;
}

extern "C" {

int32_t ea_performer_parts(const ea_perf_geom* g) {
  LaraP p = {};
  if (fill_perf(g, p, true) != EA_OK) return EA_E_BADARG;
  return p.nsplit * lara_nsub(p.NCT);
}

int ea_performer_kmax(const ea_perf_geom* g, const ea_t4* k, const float* W, float* p_ml, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !W || !p_ml) return EA_E_BADARG;
  p.k = mkl(k); p.omega = W; p.p_ml = p_ml;
  return lara_y_dispatch(LY_PMAX, p, g->dtype, (hipStream_t)stream);
}

int ea_performer_kv(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* W, const float* stab, float* p_ml, float* p_kv, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !W || !stab || !p_ml || !p_kv) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.mask = mask; p.omega = W; p.stab = stab; p.p_ml = p_ml; p.p_acc0 = p_kv;
  return lara_y_dispatch(LY_PKV, p, g->dtype, (hipStream_t)stream);
}

int ea_performer_out(const ea_perf_geom* g, const ea_t4* q, const float* W, const float* kv,
                     const float* ksum, const ea_t4* out, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(out, g->D) || !W || !kv || !ksum) return EA_E_BADARG;
  p.q = mkl(q); p.o = mkl(out); p.omega = W; p.kv = kv; p.cst = ksum;
  return lara_x_dispatch(LX_POUT, p, g->dtype, (hipStream_t)stream);
}

int ea_performer_bwd_q(const ea_perf_geom* g, const ea_t4* q, const ea_t4* out, const ea_t4* dout,
                       const float* W, const float* kv, const float* ksum, const ea_t4* dq,
                       float* stabq, float* invden, float* dden, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(out, g->D) || !t4_ok(dout, g->D) || !t4_ok(dq, g->D) || !W || !kv ||
      !ksum || !stabq || !invden || !dden) return EA_E_BADARG;
  p.q = mkl(q); p.o = mkl(out); p.dout = mkl(dout); p.dq = mkl(dq); p.omega = W; p.kv = kv; p.cst = ksum;
  p.lseZ = stabq; p.tmean = invden; p.rowdot = dden;
  return lara_x_dispatch(LX_PBWDQ, p, g->dtype, (hipStream_t)stream);
}

int ea_performer_bwd_qstats(const ea_perf_geom* g, const ea_t4* q, const ea_t4* dout, const float* W,
                            const float* stabq, const float* invden, const float* dden,
                            float* p_ml, float* p_dkv, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, true);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, g->D) || !t4_ok(dout, g->D) || !W || !stabq || !invden || !dden || !p_ml || !p_dkv)
    return EA_E_BADARG;
  p.q = mkl(q); p.dout = mkl(dout); p.omega = W;
  p.lseZ = const_cast<float*>(stabq); p.tmean = const_cast<float*>(invden); p.rowdot = const_cast<float*>(dden);
  p.p_ml = p_ml; p.p_acc0 = p_dkv;
  return lara_y_dispatch(LY_PBWDQ, p, g->dtype, (hipStream_t)stream);
}

int ea_performer_bwd_k(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* W, const float* stab, const float* dkv, const float* dksum,
                       const ea_t4* dk, const ea_t4* dv, void* stream) {
  LaraP p = {};
  int rc = fill_perf(g, p, false);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !t4_ok(dk, g->D) || !t4_ok(dv, g->D) || !W || !stab ||
      !dkv || !dksum) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.dk = mkl(dk); p.dv = mkl(dv); p.mask = mask; p.omega = W; p.stab = stab;
  p.dkv = dkv; p.rsum = dksum;
  return lara_x_dispatch(LX_PBWDK, p, g->dtype, (hipStream_t)stream);
}

}  // extern "C"

// ---- fused LARA landmark pipeline ----
static int fill_lmk(const ea_lmk_geom* g, LmkP& p) {
  if (!g || g->BH <= 0 || g->L <= 0 || g->C <= 0 || (g->D != 32 && g->D != 64)) return EA_E_BADARG;
  if (g->L > 64 || g->C > 64) return EA_E_UNSUPPORTED;
  if (g->C % g->L != 0 || (g->dup == 0 && g->C != g->L) || (g->dup != 0 && g->C != 2 * g->L)) return EA_E_BADARG;
  if (g->mis < 0 || g->mis > 2 || g->dup < 0 || g->dup > 2) return EA_E_BADARG;
  p.BH = g->BH; p.L = g->L; p.C = g->C; p.D = g->D;
  p.has_mlp = g->has_mlp; p.mixed = g->mixed; p.mis = g->mis; p.dup = g->dup; p.scale = g->scale;
  p.eva = g->eva;
  if (g->eva && (g->dup != 0 || !g->has_mlp)) return EA_E_BADARG;
  return EA_OK;
}
#define LMK_PARAMS(p)                                                                        \
  p.Wq = Wq; p.bq = bq; p.gq = gq; p.cq = cq; p.Wk = Wk; p.bk = bk; p.gk = gk; p.ck = ck;
extern "C" {

int ea_lara_landmarks_fwd(const ea_lmk_geom* g, const float* pq, const float* pk,
                          const float* Wq, const float* bq, const float* gq, const float* cq,
                          const float* Wk, const float* bk, const float* gk, const float* ck,
                          const float* noise, float* omega, float* qbar_rows, float* bhv, float* lp,
                          float* saved, void* stream) {
  LmkP p = {};
  int rc = fill_lmk(g, p);
  if (rc != EA_OK) return rc;
  if (!pq || !pk || !omega || (!lp && !g->eva)) return EA_E_BADARG;
  if (g->has_mlp && (!Wq || !bq || !gq || !cq || !Wk || !bk || !gk || !ck)) return EA_E_BADARG;
  if (!g->eva && g->mis == EA_MIS_OPT && (!qbar_rows || !bhv)) return EA_E_BADARG;
  if ((g->eva || g->mis == EA_MIS_BIASED) && !qbar_rows) return EA_E_BADARG;
  if (g->dup != 0 && !noise) return EA_E_BADARG;
  p.pq = pq; p.pk = pk; LMK_PARAMS(p)
  p.noise = noise; p.omega = omega; p.qbar_rows = qbar_rows; p.bhv = bhv; p.lp = lp;
  p.saved = saved;
  return lara_lmk_dispatch(false, p, (hipStream_t)stream);
}

int ea_lara_landmarks_fwd_cb(const ea_lmk_geom* g, const float* pq, const float* pk,
                             const float* Wq, const float* bq, const float* gq, const float* cq,
                             const float* Wk, const float* bk, const float* gk, const float* ck,
                             const float* noise, const float* colbias, float* omega, float* qbar_rows, float* bhv,
                             float* lp, float* saved, void* stream) {
  LmkP p = {};
  int rc = fill_lmk(g, p);
  if (rc != EA_OK) return rc;
  if (!pq || !pk || !omega || !lp || g->eva || !g->mixed || !colbias) return EA_E_BADARG;
  if (g->has_mlp && (!Wq || !bq || !gq || !cq || !Wk || !bk || !gk || !ck)) return EA_E_BADARG;
  if (g->mis == EA_MIS_OPT && (!qbar_rows || !bhv)) return EA_E_BADARG;
  if (g->mis == EA_MIS_BIASED && !qbar_rows) return EA_E_BADARG;
  if (g->dup != 0 && !noise) return EA_E_BADARG;
  p.pq = pq; p.pk = pk; LMK_PARAMS(p)
  p.noise = noise; p.omega = omega; p.qbar_rows = qbar_rows; p.bhv = bhv; p.lp = lp;
  p.saved = saved; p.colbias = colbias;
  return lara_lmk_dispatch(false, p, (hipStream_t)stream);
}

int64_t ea_lara_landmarks_saved_floats(const ea_lmk_geom* g) {
  LmkP p = {};
  if (fill_lmk(g, p) != EA_OK) return EA_E_BADARG;
  return (int64_t)g->BH * (int64_t)lara_lmk_saved_per_bh(g->L, g->D);
}

// dqr_S > 1 (composite EVA backward, round 6): d_qbar_rows = slice 0 of dqr_S <= 4 slice partials `dqr_stride` floats apart
static int lara_landmarks_bwd_qparts(const ea_lmk_geom* g, const float* pq, const float* pk,
                                     const float* Wq, const float* bq, const float* gq, const float* cq,
                                     const float* Wk, const float* bk, const float* gk, const float* ck,
                                     const float* noise, const float* d_omega, const float* d_qbar_rows, int dqr_S, long dqr_stride,
                                     const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                                     float* dW_part, float* dvec_part, const float* saved, void* stream) {
  LmkP p = {};
  int rc = fill_lmk(g, p);
  if (rc != EA_OK) return rc;
  if (!pq || !pk || !d_omega || (!d_lp && !g->eva) || !dpq || !dpk) return EA_E_BADARG;
  if (g->eva && !d_qbar_rows) return EA_E_BADARG;
  if (dqr_S < 1 || dqr_S > 4 || (dqr_S > 1 && !g->eva)) return EA_E_BADARG;
  if (g->has_mlp && (!Wq || !bq || !gq || !cq || !Wk || !bk || !gk || !ck || !dW_part || !dvec_part))
    return EA_E_BADARG;
  if (g->dup != 0 && !noise) return EA_E_BADARG;
  p.pq = pq; p.pk = pk; LMK_PARAMS(p)
  p.noise = noise; p.d_omega = d_omega; p.d_qbar_rows = d_qbar_rows; p.d_bhv = d_bhv; p.d_lp = d_lp;
  p.dqr_S = dqr_S; p.dqr_stride = dqr_stride;
  p.dpq = dpq; p.dpk = dpk; p.dW_part = dW_part; p.dvec_part = dvec_part;
  p.saved = const_cast<float*>(saved);
  return lara_lmk_dispatch(true, p, (hipStream_t)stream);
}

int ea_lara_landmarks_bwd(const ea_lmk_geom* g, const float* pq, const float* pk,
                          const float* Wq, const float* bq, const float* gq, const float* cq,
                          const float* Wk, const float* bk, const float* gk, const float* ck,
                          const float* noise, const float* d_omega, const float* d_qbar_rows,
                          const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                          float* dW_part, float* dvec_part, const float* saved, void* stream) {
  return lara_landmarks_bwd_qparts(g, pq, pk, Wq, bq, gq, cq, Wk, bk, gk, ck, noise, d_omega, d_qbar_rows, 1, 0, d_bhv, d_lp, dpq,
                                   dpk, dW_part, dvec_part, saved, stream);
}

int ea_lara_landmarks_bwd_parts(const ea_lmk_geom* g, const float* pq, const float* pk,
                                const float* Wq, const float* bq, const float* gq, const float* cq,
                                const float* Wk, const float* bk, const float* gk, const float* ck,
                                const float* noise, const float* d_omega, int32_t dom_S, const float* dom_parts, float dom_scale,
                                const float* d_qbar_rows, const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                                float* dW_part, float* dvec_part, const float* saved, void* stream) {
  LmkP p = {};
  int rc = fill_lmk(g, p);
  if (rc != EA_OK) return rc;
  if (g->eva || !pq || !pk || !d_omega || !d_lp || !dpq || !dpk || !dom_parts || dom_S < 1 || dom_S > 4) return EA_E_BADARG;
  if (g->has_mlp && (!Wq || !bq || !gq || !cq || !Wk || !bk || !gk || !ck || !dW_part || !dvec_part))
    return EA_E_BADARG;
  if (g->dup != 0 && !noise) return EA_E_BADARG;
  p.pq = pq; p.pk = pk; LMK_PARAMS(p)
  p.noise = noise; p.d_omega = d_omega; p.d_qbar_rows = d_qbar_rows; p.d_bhv = d_bhv; p.d_lp = d_lp;
  p.dom_parts = dom_parts; p.dom_S = dom_S; p.dom_scale = dom_scale;
  p.dpq = dpq; p.dpk = dpk; p.dW_part = dW_part; p.dvec_part = dvec_part;
  p.saved = const_cast<float*>(saved);
  return lara_lmk_dispatch(true, p, (hipStream_t)stream);
}

int ea_lara_landmarks_bwd_cb(const ea_lmk_geom* g, const float* pq, const float* pk,
                             const float* Wq, const float* bq, const float* gq, const float* cq,
                             const float* Wk, const float* bk, const float* gk, const float* ck,
                             const float* noise, const float* colbias, const float* d_omega, const float* d_qbar_rows,
                             const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                             float* dW_part, float* dvec_part, float* d_colbias, const float* saved, void* stream) {
  LmkP p = {};
  int rc = fill_lmk(g, p);
  if (rc != EA_OK) return rc;
  if (!pq || !pk || !d_omega || !d_lp || !dpq || !dpk || g->eva || !g->mixed || !colbias || !d_colbias || !saved)
    return EA_E_BADARG;
  if (g->has_mlp && (!Wq || !bq || !gq || !cq || !Wk || !bk || !gk || !ck || !dW_part || !dvec_part))
    return EA_E_BADARG;
  if (g->dup != 0 && !noise) return EA_E_BADARG;
  p.pq = pq; p.pk = pk; LMK_PARAMS(p)
  p.noise = noise; p.d_omega = d_omega; p.d_qbar_rows = d_qbar_rows; p.d_bhv = d_bhv; p.d_lp = d_lp;
  p.dpq = dpq; p.dpk = dpk; p.dW_part = dW_part; p.dvec_part = dvec_part;
  p.saved = const_cast<float*>(saved); p.colbias = colbias; p.d_colbias = d_colbias;
  return lara_lmk_dispatch(true, p, (hipStream_t)stream);
}

}  // extern "C"

// ---- LARA partial merges ----
extern "C" {

int ea_lara_merge_fwd(int32_t BH, int32_t S, int32_t C, int32_t D, int32_t has_t,
                      const float* p_ml, const float* p_kv, const float* lp,
                      float* kv, float* lse_k, float* lse_t, float* cst, void* stream) {
  if (BH <= 0 || S <= 0 || C <= 0 || D <= 0 || !p_ml || !p_kv || !lp || !kv || !lse_k || !cst ||
      (has_t && !lse_t)) return EA_E_BADARG;
  MergeP p = {};
  p.BH = BH; p.S = S; p.C = C; p.D = D; p.has_t = has_t;
  p.p_ml = p_ml; p.p_kv = p_kv; p.lp = lp; p.kv = kv; p.lse_k = lse_k; p.lse_t = lse_t; p.cst = cst;
  return lara_merge_dispatch(false, p, (hipStream_t)stream);
}

int ea_lara_merge_bwd(int32_t BH, int32_t S, int32_t C, int32_t D, int32_t has_t, float scale,
                      const float* p_ml, const float* p_dkv, const float* p_dom, const float* p_m1,
                      const float* p_m2, const float* kv, const float* qbar,
                      float* r, float* dbh, float* dlp, float* dkk, float* dkv, float* domq,
                      float* dqbar, float* uq, void* stream) {
  if (BH <= 0 || S <= 0 || C <= 0 || D <= 0 || !p_ml || !p_dkv || !p_dom || !kv || !r || !dkk ||
      !dkv || !domq) return EA_E_BADARG;
  if (has_t && (!p_m1 || !p_m2 || !qbar || !dqbar || !uq)) return EA_E_BADARG;
  MergeP p = {};
  p.BH = BH; p.S = S; p.C = C; p.D = D; p.has_t = has_t; p.scale = scale;
  p.p_ml = p_ml; p.acc0 = p_dkv; p.acc1 = p_dom; p.acc2 = p_m1; p.acc3 = p_m2;
  p.kv = const_cast<float*>(kv); p.qbar = qbar;
  p.r = r; p.dbh = dbh; p.dlp = dlp; p.dkk = dkk; p.dkv = dkv; p.domq = domq; p.dqbar = dqbar; p.uq = uq;
  return lara_merge_dispatch(true, p, (hipStream_t)stream);
}

}  // extern "C"

// ---- projection bias gradient ----
namespace ea {
int colsum_parts(int rows, int cols);
int colsum_dispatch(int dtype, const void* x, float* part, float* out, int rows, int cols, hipStream_t st);
int colsum_f32_dispatch(const float* x, float* out, int rows, int cols, const float* x2, float* out2, int cols2, hipStream_t st);
int gather_sum_dispatch(const float* g, const int* inv, float* out, int rows, int K, int cols, hipStream_t st);
int slice_sum_dispatch(const float* a, const float* p, float* out, int BH, int S, int n, float scale, hipStream_t st);
int multi_cast_dispatch(int dtype, int K, const float* const* src, const long long* n, void* const* dst, hipStream_t st);
int table_bias_fwd_dispatch(const float* table, const int* idx, float* out, int h, int th, int Wq, int Wk, int ld, float scale,
                            hipStream_t st);
int table_bias_bwd_dispatch(const float* g, const int* inv, float* dtable, int rows, int K, int h, int Wq, int Wk, int ld,
                            float scale, hipStream_t st);
int stream_copy_dispatch(const void* src, void* dst, size_t bytes, hipStream_t st);
}  // namespace ea

extern "C" {

int ea_bias_grad_parts(int32_t rows, int32_t cols) { return ea::colsum_parts(rows, cols); }

int ea_bias_grad(int32_t dtype, int32_t rows, int32_t cols, const void* dy, float* part, float* db, void* stream) {
  if (!dy || !part || !db) return EA_E_BADARG;
  return ea::colsum_dispatch(dtype, dy, part, db, rows, cols, (hipStream_t)stream);
}

int ea_colsum_f32(int32_t rows, int32_t cols, const float* x, float* out, void* stream) {
  if (!x || !out) return EA_E_BADARG;
  return ea::colsum_f32_dispatch(x, out, rows, cols, nullptr, nullptr, 0, (hipStream_t)stream);
}

int ea_colsum2_f32(int32_t rows, int32_t cols1, const float* x1, float* out1, int32_t cols2, const float* x2, float* out2,
                   void* stream) {
  if (!x1 || !out1 || !x2 || !out2 || cols2 <= 0) return EA_E_BADARG;
  return ea::colsum_f32_dispatch(x1, out1, rows, cols1, x2, out2, cols2, (hipStream_t)stream);
}

int ea_gather_sum(int32_t rows, int32_t K, int32_t cols, const float* g, const int32_t* inv, float* out, void* stream) {
  if (!g || !inv || !out) return EA_E_BADARG;
  return ea::gather_sum_dispatch(g, inv, out, rows, K, cols, (hipStream_t)stream);
}

int ea_multi_cast(int32_t dtype, int32_t K, const float* const* src, const int64_t* n, void* const* dst, void* stream) {
  if (!src || !n || !dst) return EA_E_BADARG;
  return ea::multi_cast_dispatch(dtype, K, src, (const long long*)n, dst, (hipStream_t)stream);
}

int ea_table_bias_fwd(int32_t h, int32_t th, int32_t Wq, int32_t Wk, int32_t ld, float scale, const float* table, const int32_t* idx,
                      float* out, void* stream) {
  if (!table || !idx || !out) return EA_E_BADARG;
  return ea::table_bias_fwd_dispatch(table, idx, out, h, th, Wq, Wk, ld, scale, (hipStream_t)stream);
}

int ea_table_bias_bwd(int32_t rows, int32_t K, int32_t h, int32_t Wq, int32_t Wk, int32_t ld, float scale, const float* g,
                      const int32_t* inv, float* dtable, void* stream) {
  if (!g || !inv || !dtable) return EA_E_BADARG;
  return ea::table_bias_bwd_dispatch(g, inv, dtable, rows, K, h, Wq, Wk, ld, scale, (hipStream_t)stream);
}


int ea_stream_copy(const void* src, void* dst, int64_t bytes, void* stream) {
  if (!src || !dst || bytes <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return EA_E_BADARG;
  return ea::stream_copy_dispatch(src, dst, (size_t)bytes, (hipStream_t)stream);
}

int ea_slice_sum(int32_t BH, int32_t S, int32_t n, float scale, const float* a, const float* parts,
                 float* out, void* stream) {
  if (!parts || !out) return EA_E_BADARG;
  return ea::slice_sum_dispatch(a, parts, out, BH, S, n, scale, (hipStream_t)stream);
}

// ---- LARA 1-D landmark proposals: segment means of (LayerNorm'd) rows ----
static int fill_seg(const ea_geom* g, SegP& p) {
  if (!g || g->B <= 0 || g->H <= 0 || g->N <= 0 || g->L <= 0 || (g->D != 32 && g->D != 64) ||
      (g->dtype != EA_BF16 && g->dtype != EA_F16)) return EA_E_BADARG;
  if (g->N <= g->L) return EA_E_UNSUPPORTED;            // the reference uses the rows themselves then
  p.B = g->B; p.H = g->H; p.N = g->N; p.L = g->L;
  p.segs = g->N / g->L;
  p.nshort = g->N % g->L == 0 ? g->L : (p.segs + 1) * g->L - g->N;   // lara.py:111-124
  return EA_OK;
}

int ea_lara_segment_fwd(const ea_geom* g, const ea_t4* q2, const ea_t4* k2, const uint8_t* mask,
                        const float* bias_q, const float* bias_k, const float* mbias_q, const float* mbias_k,
                        const float* gq, const float* cq, const float* gk, const float* ck,
                        float* qbar, float* kbar, void* stream) {
  SegP p = {};
  int rc = fill_seg(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok32(q2, g->D, g->N) || !t4_ok32(k2, g->D, g->N) || !qbar || !kbar) return EA_E_BADARG;
  if ((gq == nullptr) != (gk == nullptr) || (gq && (!cq || !ck))) return EA_E_BADARG;
  SET3(q, q2); SET3(k, k2);
  p.mask = mask; p.bias_q = bias_q; p.bias_k = bias_k; p.mbias_q = mbias_q; p.mbias_k = mbias_k;
  p.gq = gq; p.cq = cq; p.gk = gk; p.ck = ck; p.qbar = qbar; p.kbar = kbar;
  return lara_segment_dispatch(false, p, g->dtype, g->D, (hipStream_t)stream);
}

int ea_lara_segment_bwd(const ea_geom* g, const ea_t4* q2, const ea_t4* k2, const uint8_t* mask,
                        const float* bias_q, const float* bias_k, const float* mbias_q, const float* mbias_k,
                        const float* gq, const float* cq, const float* gk, const float* ck,
                        const float* d_qbar, const float* d_kbar, const ea_t4* dq2, const ea_t4* dk2,
                        float* part, void* stream) {
  SegP p = {};
  int rc = fill_seg(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok32(q2, g->D, g->N) || !t4_ok32(k2, g->D, g->N) || !t4_ok32(dq2, g->D, g->N) ||
      !t4_ok32(dk2, g->D, g->N) || !d_qbar || !d_kbar) return EA_E_BADARG;
  if ((gq == nullptr) != (gk == nullptr) || (gq && (!cq || !ck || !part))) return EA_E_BADARG;
  SET3(q, q2); SET3(k, k2); SET3(dq, dq2); SET3(dk, dk2);
  p.mask = mask; p.bias_q = bias_q; p.bias_k = bias_k; p.mbias_q = mbias_q; p.mbias_k = mbias_k;
  p.gq = gq; p.cq = cq; p.gk = gk; p.ck = ck; p.d_qbar = d_qbar; p.d_kbar = d_kbar; p.part = part;
  return lara_segment_dispatch(true, p, g->dtype, g->D, (hipStream_t)stream);
}

}  // extern "C"

// ---- LARA 'adaptive-1d' with the generator Linear inside the segment kernels (ea_lara_seglin.hip) ----
namespace ea {
struct SegLinP {
  char *q, *k;
  int64_t q_sb, q_sh, q_sn, k_sb, k_sh, k_sn;
  char *dq, *dk;
  int64_t dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn;
  const float *Gq, *Gk, *gqb, *gkb;
  const float *lnq_w, *lnq_b, *lnk_w, *lnk_b;
  float *qbar, *kbar;
  const float *d_qbar, *d_kbar;
  float* part;
  void* stats;
  float* dG_part;
  int B, H, N, L, segs, nshort, groups, seg_per_group, cgroups, cseg_per_group;
  const float *fin_qbar, *fin_uq, *fin_lse;
  int C;
  float fin_scale, fin_scale_log2;
};
int seglin_groups(int BH, int L);
int seglin_dispatch(int which, const SegLinP& p, int dtype, hipStream_t st);
}  // namespace ea

static int fill_seglin(const ea_geom* g, ea::SegLinP& p) {
  SegP sp = {};
  const int rc = fill_seg(g, sp);
  if (rc != EA_OK) return rc;
  if (g->D != 64) return EA_E_UNSUPPORTED;
  p.B = sp.B; p.H = sp.H; p.N = sp.N; p.L = sp.L; p.segs = sp.segs; p.nshort = sp.nshort;
  return EA_OK;
}

extern "C" {

int32_t ea_lara_seglin_groups(const ea_geom* g) {
  ea::SegLinP p = {};
  const int rc = fill_seglin(g, p);
  return rc != EA_OK ? rc : ea::seglin_groups(p.B * p.H, p.L);
}

int ea_lara_seglin_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                       const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                       float* qbar, float* kbar, void* stream) {
  ea::SegLinP p = {};
  const int rc = fill_seglin(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok32(q, 64, g->N) || !t4_ok32(k, 64, g->N) || !Gq || !gq_b || !Gk || !gk_b || !lnq_w || !lnq_b || !lnk_w || !lnk_b ||
      !qbar || !kbar) return EA_E_BADARG;
  SET3(q, q); SET3(k, k);
  p.Gq = Gq; p.gqb = gq_b; p.Gk = Gk; p.gkb = gk_b; p.lnq_w = lnq_w; p.lnq_b = lnq_b; p.lnk_w = lnk_w; p.lnk_b = lnk_b;
  p.qbar = qbar; p.kbar = kbar;
  return ea::seglin_dispatch(0, p, g->dtype, (hipStream_t)stream);
}

static int seglin_bwd_impl(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                           const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                           const float* d_qbar, const float* d_kbar, const ea_t4* dq, const ea_t4* dk, float* part, float* dG_part,
                           float* stats, const float* fin_qbar, const float* fin_uq, const float* fin_lse, int32_t C, float scale,
                           bool fin, void* stream) {
  ea::SegLinP p = {};
  int rc = fill_seglin(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok32(q, 64, g->N) || !t4_ok32(k, 64, g->N) || !t4_ok32(dq, 64, g->N) || !t4_ok32(dk, 64, g->N) || !Gq || !gq_b || !Gk ||
      !gk_b || !lnq_w || !lnq_b || !lnk_w || !lnk_b || !d_qbar || !d_kbar || !part || !dG_part || !stats || ((uintptr_t)stats & 15))
    return EA_E_BADARG;
  if (fin) {
    if (!fin_qbar || !fin_uq || !fin_lse || C < 1 || !(scale > 0.f)) return EA_E_BADARG;
    if (C > 64) return EA_E_UNSUPPORTED;
    p.fin_qbar = fin_qbar; p.fin_uq = fin_uq; p.fin_lse = fin_lse; p.C = C;
    p.fin_scale = scale; p.fin_scale_log2 = scale * 1.4426950408889634f;
  }
  SET3(q, q); SET3(k, k); SET3(dq, dq); SET3(dk, dk);
  p.Gq = Gq; p.gqb = gq_b; p.Gk = Gk; p.gkb = gk_b; p.lnq_w = lnq_w; p.lnq_b = lnq_b; p.lnk_w = lnk_w; p.lnk_b = lnk_b;
  p.d_qbar = d_qbar; p.d_kbar = d_kbar; p.part = part; p.dG_part = dG_part; p.stats = stats;
  rc = ea::seglin_dispatch(fin ? 3 : 1, p, g->dtype, (hipStream_t)stream);
  if (rc != EA_OK) return rc;
  return ea::seglin_dispatch(2, p, g->dtype, (hipStream_t)stream);
}

int ea_lara_seglin_bwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                       const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                       const float* d_qbar, const float* d_kbar, const ea_t4* dq, const ea_t4* dk, float* part, float* dG_part,
                       float* stats, void* stream) {
  return seglin_bwd_impl(g, q, k, Gq, gq_b, Gk, gk_b, lnq_w, lnq_b, lnk_w, lnk_b, d_qbar, d_kbar, dq, dk, part, dG_part, stats,
                         nullptr, nullptr, nullptr, 0, 0.f, false, stream);
}

int ea_lara_seglin_bwd_fin(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                           const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                           const float* d_qbar, const float* d_kbar, const ea_t4* dq, const ea_t4* dk, float* part,
                           float* dG_part, float* stats, const float* fin_qbar, const float* fin_uq, const float* fin_lse_t,
                           int32_t C, float scale, void* stream) {
  return seglin_bwd_impl(g, q, k, Gq, gq_b, Gk, gk_b, lnq_w, lnq_b, lnk_w, lnk_b, d_qbar, d_kbar, dq, dk, part, dG_part, stats,
                         fin_qbar, fin_uq, fin_lse_t, C, scale, true, stream);
}

}  // extern "C"

// ---- mu networks: per-row Linear (+ LayerNorm) on the chunk means ----
extern "C" {

int32_t ea_rows_mlp_parts(int32_t R, int32_t D) {
  if (R <= 0 || (D != 32 && D != 64 && D != 128)) return EA_E_BADARG;
  return ea::rows_mlp_blocks(R, D);
}

int ea_rows_mlp_fwd(int32_t R, int32_t D, int32_t sides, int32_t layer_norm,
                    const float* x0, const float* x1, const float* W0, const float* W1,
                    const float* b0, const float* b1, const float* g0, const float* g1,
                    const float* c0, const float* c1, float* y0, float* y1,
                    float* zhat, float* rstd, void* stream) {
  if (R <= 0 || sides < 1 || sides > 2 || !x0 || !W0 || !b0 || !y0) return EA_E_BADARG;
  if (sides == 2 && (!x1 || !W1 || !b1 || !y1)) return EA_E_BADARG;
  if (layer_norm && (!g0 || !c0 || (sides == 2 && (!g1 || !c1)))) return EA_E_BADARG;
  if ((zhat == nullptr) != (rstd == nullptr)) return EA_E_BADARG;
  RowsP p = {};
  p.R = R;
  p.x[0] = x0; p.x[1] = x1; p.W[0] = W0; p.W[1] = W1; p.b[0] = b0; p.b[1] = b1;
  p.g[0] = g0; p.g[1] = g1; p.c[0] = c0; p.c[1] = c1; p.y[0] = y0; p.y[1] = y1;
  p.zhat = zhat; p.rstd = rstd;
  return ea::rows_mlp_dispatch(p, D, sides, layer_norm, false, (hipStream_t)stream);
}

int ea_rows_mlp_bwd(int32_t R, int32_t D, int32_t sides, int32_t layer_norm,
                    const float* dy0, const float* dy1, const float* x0, const float* x1,
                    const float* W0, const float* W1, const float* g0, const float* g1,
                    const float* zhat, const float* rstd, float* dx0, float* dx1,
                    float* feed, float* dW_part, void* stream) {
  if (R <= 0 || sides < 1 || sides > 2 || !dy0 || !x0 || !W0 || !dx0 || !feed || !dW_part) return EA_E_BADARG;
  if (sides == 2 && (!dy1 || !x1 || !W1 || !dx1)) return EA_E_BADARG;
  if (layer_norm && (!g0 || !zhat || !rstd || (sides == 2 && !g1))) return EA_E_BADARG;
  RowsP p = {};
  p.R = R;
  p.dy[0] = dy0; p.dy[1] = dy1; p.x[0] = x0; p.x[1] = x1; p.W[0] = W0; p.W[1] = W1;
  p.g[0] = g0; p.g[1] = g1; p.zhat = const_cast<float*>(zhat); p.rstd = const_cast<float*>(rstd);
  p.dx[0] = dx0; p.dx[1] = dx1; p.feed = feed; p.dW_part = dW_part;
  return ea::rows_mlp_dispatch(p, D, sides, layer_norm, true, (hipStream_t)stream);
}

}  // extern "C"

// ---- projection weight + bias gradient (ea_wgrad.hip) ----
extern "C" {

int32_t ea_wgrad_parts(int32_t rows, int32_t out_features, int32_t in_features) {
  return wgrad_slices(rows, out_features, in_features);
}

int ea_wgrad(int32_t dtype, int32_t rows, int32_t out_features, int32_t in_features, const void* dy, const void* x,
             float* dw_part, float* db_part, int64_t part_ld, void* stream) {
  if (!dy || !x || !dw_part || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dw_part & 15)) return EA_E_BADARG;
  if (part_ld < (int64_t)out_features * in_features || (part_ld & 3)) return EA_E_BADARG;
  return wgrad_dispatch(dtype, dy, x, dw_part, db_part, (long)part_ld, rows, out_features, in_features, (hipStream_t)stream);
}

int32_t ea_wgrad_pair_parts(int32_t rows, int32_t out1, int32_t in1, int32_t out2, int32_t in2) {
  return wgrad_pair_slices(rows, out1, in1, out2, in2);
}

int ea_wgrad_pair(int32_t dtype, int32_t rows, int32_t out1, int32_t in1, const void* dy1, const void* x1, float* dw_part1,
                  float* db_part1, int64_t part_ld1, int32_t out2, int32_t in2, const void* dy2, const void* x2,
                  float* dw_part2, float* db_part2, int64_t part_ld2, void* stream) {
  if ((dtype != EA_BF16 && dtype != EA_F16) || rows <= 0 || !dy1 || !x1 || !dw_part1 || !dy2 || !x2 || !dw_part2) return EA_E_BADARG;
  if (part_ld1 < (int64_t)out1 * in1 || part_ld2 < (int64_t)out2 * in2 || (part_ld1 & 3) || (part_ld2 & 3)) return EA_E_BADARG;
  if (((uintptr_t)dy1 | (uintptr_t)x1 | (uintptr_t)dw_part1 | (uintptr_t)dy2 | (uintptr_t)x2 | (uintptr_t)dw_part2) & 15)
    return EA_E_BADARG;
  return wgrad_pair_dispatch(dtype, rows, dy1, x1, dw_part1, db_part1, (long)part_ld1, out1, in1, dy2, x2, dw_part2, db_part2,
                             (long)part_ld2, out2, in2, (hipStream_t)stream);
}

int ea_part_sum(int32_t S, int32_t n, int64_t ld, const float* parts, float* out, void* stream) {
  if (!parts || !out || ((uintptr_t)parts & 15) || ((uintptr_t)out & 15)) return EA_E_BADARG;
  return part_sum_dispatch(parts, out, S, n, (long)ld, (hipStream_t)stream);
}

int ea_multi_sum(int32_t K, const float* const* parts, const int32_t* S, const int32_t* n, const int64_t* ld, float* const* out,
                 void* stream) {
  if (!parts || !S || !n || !ld || !out) return EA_E_BADARG;
  return multi_sum_dispatch(K, parts, S, n, (const long long*)ld, out, (hipStream_t)stream);
}

}  // extern "C"

// ---- LARA sampling + proposal densities beyond the fused landmark kernels (ea_lara_segment.hip) ----
extern "C" {

int ea_lara_sample_fwd(int32_t BH, int32_t L, int32_t C, int32_t D, int32_t mis, int32_t mode, float scale,
                       const float* qbar, const float* mu, const float* noise, float* omega, float* qbar_rows, float* bhv,
                       float* lp, void* stream) {
  if (!qbar || !mu || !omega || !lp || (mode != 0 && !noise) || mis < 0 || mis > 2) return EA_E_BADARG;
  if (mis == EA_MIS_OPT && (!qbar_rows || !bhv)) return EA_E_BADARG;
  if (mis == EA_MIS_BIASED && !qbar_rows) return EA_E_BADARG;
  if ((mode == 0) != (C == L) || (mode != 0 && C != 2 * L)) return EA_E_BADARG;
  ea::SampP p = {};
  p.qbar = qbar; p.mu = mu; p.noise = noise; p.omega = omega; p.qrows = mis == EA_MIS_BH ? nullptr : qbar_rows;
  p.bhv = mis == EA_MIS_OPT ? bhv : nullptr; p.lp = lp;
  p.BH = BH; p.L = L; p.C = C; p.D = D; p.mis = mis; p.mode = mode; p.scale = scale;
  return ea::lara_sample_dispatch(false, p, (hipStream_t)stream);
}

int ea_lara_sample_bwd(int32_t BH, int32_t L, int32_t C, int32_t D, int32_t mis, int32_t mode, float scale,
                       const float* qbar, const float* mu, const float* noise, const float* d_omega, const float* d_qbar_rows,
                       const float* d_bhv, const float* d_lp, float* d_qbar, float* d_mu, void* stream) {
  if (!qbar || !mu || !d_omega || !d_qbar || !d_mu || (mode != 0 && !noise) || mis < 0 || mis > 2) return EA_E_BADARG;
  if ((mode == 0) != (C == L) || (mode != 0 && C != 2 * L)) return EA_E_BADARG;
  ea::SampP p = {};
  p.qbar = qbar; p.mu = mu; p.noise = noise;
  p.d_omega = d_omega; p.d_qrows = mis == EA_MIS_BH ? nullptr : d_qbar_rows; p.d_bhv = mis == EA_MIS_OPT ? d_bhv : nullptr;
  p.d_lp = d_lp; p.d_qbar = d_qbar; p.d_mu = d_mu;
  p.BH = BH; p.L = L; p.C = C; p.D = D; p.mis = mis; p.mode = mode; p.scale = scale;
  return ea::lara_sample_dispatch(true, p, (hipStream_t)stream);
}

}  // extern "C"

// ---- adaptive 2-D pooling of a token grid that does not divide evenly (ea_lara_segment.hip) ----
extern "C" {

int ea_adaptive_pool2d_fwd(int32_t dtype, int32_t B, int32_t H, int32_t gh, int32_t gw, int32_t side, int32_t D,
                           const ea_t4* x, float* mean, void* stream) {
  if (!x || !x->ptr || !mean) return EA_E_BADARG;
  return pool2d_dispatch(false, dtype, x->ptr, (long)x->sb, (long)x->sh, (long)x->sn, mean, B, H, gh, gw, side, D,
                         (hipStream_t)stream);
}

int ea_adaptive_pool2d_bwd(int32_t dtype, int32_t B, int32_t H, int32_t gh, int32_t gw, int32_t side, int32_t D,
                           const float* dmean, const ea_t4* dx, void* stream) {
  if (!dx || !dx->ptr || !dmean) return EA_E_BADARG;
  return pool2d_dispatch(true, dtype, dx->ptr, (long)dx->sb, (long)dx->sh, (long)dx->sn, const_cast<float*>(dmean), B, H, gh,
                         gw, side, D, (hipStream_t)stream);
}

}  // extern "C"

// ---- LARA 'adaptive-1d': the generators' Linear folded into the qkv projection (ea_fold.hip) ----
namespace ea {
struct FoldP {
  const float *W, *b, *Gq, *Gk, *gqb, *gkb;
  char* w_ext;
  char* b_ext;
  float *bias_q, *bias_k;
  const float *dW_ext, *db_ext, *dbias_q, *dbias_k;
  float *dW, *db, *dGq, *dGk, *dgqb, *dgkb;
  float *dG_part, *dG_base;
  long ldw;
  int C, h, d;
};
int fold_dispatch(bool bwd, int dtype, const FoldP& p, hipStream_t st);
int fold_bwd_parts(int heads);
}  // namespace ea

extern "C" {

int ea_lara_fold_fwd(int32_t dtype, int32_t C, int32_t heads, const float* W, const float* b, const float* Gq, const float* gq_b,
                     const float* Gk, const float* gk_b, void* w_ext, void* b_ext, float* bias_q, float* bias_k, void* stream) {
  if (!W || !Gq || !Gk || !gq_b || !gk_b || !w_ext || !bias_q || !bias_k || heads <= 0 || C % heads) return EA_E_BADARG;
  ea::FoldP p = {};
  p.W = W; p.b = b; p.Gq = Gq; p.Gk = Gk; p.gqb = gq_b; p.gkb = gk_b; p.w_ext = (char*)w_ext; p.b_ext = (char*)b_ext;
  p.bias_q = bias_q; p.bias_k = bias_k; p.C = C; p.h = heads; p.d = C / heads;
  return ea::fold_dispatch(false, dtype, p, (hipStream_t)stream);
}

int32_t ea_lara_fold_parts(int32_t heads) { return ea::fold_bwd_parts(heads); }

int ea_lara_fold_bwd(int32_t C, int32_t heads, const float* W, const float* b, const float* Gq, const float* Gk,
                     const float* dW_ext, int64_t ldw, const float* db_ext, const float* dbias_q, const float* dbias_k,
                     float* dW, float* db, float* dG, float* dG_part, float* dgq_b, float* dgk_b, void* stream) {
  if (!W || !Gq || !Gk || !dW_ext || !dbias_q || !dbias_k || !dW || !dG || !dG_part || !dgq_b || !dgk_b || heads <= 0 ||
      C % heads || ldw < C) return EA_E_BADARG;
  ea::FoldP p = {};
  p.W = W; p.b = b; p.Gq = Gq; p.Gk = Gk; p.dW_ext = dW_ext; p.ldw = (long)ldw; p.db_ext = db_ext; p.dbias_q = dbias_q;
  p.dbias_k = dbias_k; p.dW = dW; p.db = db; p.dgqb = dgq_b; p.dgkb = dgk_b;
  p.C = C; p.h = heads; p.d = C / heads;
  // the bias part of dG goes to a scratch behind the partials; one slice sum adds everything up into dG [2, d, d]
  const int S = ea::fold_bwd_parts(heads);
  const size_t dd = (size_t)p.d * p.d;
  p.dG_part = dG_part; p.dG_base = dG_part + (size_t)2 * S * dd;
  int rc = ea::fold_dispatch(true, 0, p, (hipStream_t)stream);
  if (rc != EA_OK) return rc;
  return ea_slice_sum(2, S, (int32_t)dd, 1.0f, p.dG_base, dG_part, dG, stream);
}

}  // extern "C"

// ---- Performer in exact fp32 arithmetic (ea_performer_f32.hip) ----
static bool pf_t4_ok(const ea_t4* t, int dtype) {
  const int a = dtype == EA_F32 ? 4 : 8;                  // 16-byte vector access
  return t && t->ptr && ((uintptr_t)t->ptr % 16 == 0) && t->sb % a == 0 && t->sh % a == 0 && t->sn % a == 0 && t->sn >= 64;
}
static Pf32T pf_mk(const ea_t4* t) {
  Pf32T r;
  r.p = t ? (char*)t->ptr : nullptr;
  r.sb = t ? t->sb : 0; r.sh = t ? t->sh : 0; r.sn = t ? t->sn : 0;
  return r;
}
static int pf_fill(const ea_perf_geom* g, Pf32P& p) {
  if (!g || g->B <= 0 || g->H <= 0 || g->N <= 0 || g->dtype < 0 || g->dtype > EA_F32) return EA_E_BADARG;
  if (g->D != 64 || g->M <= 0 || g->M > 96 || (g->M & 15)) return EA_E_UNSUPPORTED;
  p.B = g->B; p.H = g->H; p.N = g->N; p.M = g->M; p.dtype = g->dtype;
  return EA_OK;
}

extern "C" {

int32_t ea_performer_f32_parts(const ea_perf_geom* g) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  return rc != EA_OK ? rc : pf32_slices(g->B * g->H, g->N);
}

int ea_performer_f32_kmax(const ea_perf_geom* g, const ea_t4* k, const float* W, float* p_max, void* stream) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  if (rc != EA_OK) return rc;
  if (!pf_t4_ok(k, g->dtype) || !W || !p_max) return EA_E_BADARG;
  p.k = pf_mk(k); p.W = W; p.p_max = p_max;
  return pf32_dispatch(0, p, (hipStream_t)stream);
}

int ea_performer_f32_kv(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                        const float* p_max, float* p_kv, float* p_ksum, void* stream) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  if (rc != EA_OK) return rc;
  if (!pf_t4_ok(k, g->dtype) || !pf_t4_ok(v, g->dtype) || !W || !p_max || !p_kv || !p_ksum) return EA_E_BADARG;
  p.k = pf_mk(k); p.v = pf_mk(v); p.mask = mask; p.W = W; p.p_max = const_cast<float*>(p_max); p.p_kv = p_kv; p.p_ks = p_ksum;
  return pf32_dispatch(1, p, (hipStream_t)stream);
}

int ea_performer_f32_out(const ea_perf_geom* g, const ea_t4* q, const float* W, const float* kv, const float* ksum,
                         const ea_t4* out, void* stream) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  if (rc != EA_OK) return rc;
  if (!pf_t4_ok(q, g->dtype) || !pf_t4_ok(out, g->dtype) || !W || !kv || !ksum) return EA_E_BADARG;
  p.q = pf_mk(q); p.o = pf_mk(out); p.W = W; p.kv = kv; p.ksum = ksum;
  return pf32_dispatch(2, p, (hipStream_t)stream);
}

int ea_performer_f32_bwd_q(const ea_perf_geom* g, const ea_t4* q, const ea_t4* dout, const float* W, const float* kv,
                           const float* ksum, const ea_t4* dq, float* p_dkv, float* p_dksum, void* stream) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  if (rc != EA_OK) return rc;
  if (!pf_t4_ok(q, g->dtype) || !pf_t4_ok(dout, g->dtype) || !pf_t4_ok(dq, g->dtype) || !W || !kv || !ksum || !p_dkv || !p_dksum)
    return EA_E_BADARG;
  p.q = pf_mk(q); p.dout = pf_mk(dout); p.dq = pf_mk(dq); p.W = W; p.kv = kv; p.ksum = ksum; p.p_kv = p_dkv; p.p_ks = p_dksum;
  return pf32_dispatch(3, p, (hipStream_t)stream);
}

int ea_performer_f32_bwd_k(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                           const float* p_max, const float* dkv, const float* dksum, const ea_t4* dk, const ea_t4* dv,
                           void* stream) {
  Pf32P p = {};
  const int rc = pf_fill(g, p);
  if (rc != EA_OK) return rc;
  if (!pf_t4_ok(k, g->dtype) || !pf_t4_ok(v, g->dtype) || !pf_t4_ok(dk, g->dtype) || !pf_t4_ok(dv, g->dtype) || !W || !p_max ||
      !dkv || !dksum) return EA_E_BADARG;
  p.k = pf_mk(k); p.v = pf_mk(v); p.mask = mask; p.W = W; p.p_max = const_cast<float*>(p_max); p.dkv = dkv; p.dksum = dksum;
  p.dk = pf_mk(dk); p.dv = pf_mk(dv);
  return pf32_dispatch(4, p, (hipStream_t)stream);
}

}  // extern "C"

// ---- projections as streaming kernels (ea_linear.hip) ----
extern "C" {

int32_t ea_linear_supported(int32_t in_features, int32_t out_features) { return linear_supported(in_features, out_features); }

int ea_linear(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* a, int32_t a_f32,
              int64_t lda, const void* w, const float* bias, void* y, int32_t y_f32, int64_t ldy, void* a_cast,
              void* stream) {
  if (!a || !w || !y || ((uintptr_t)a & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)a_cast & 15))
    return EA_E_BADARG;
  if (lda < in_features || ldy < out_features || (lda & 7) || (ldy & 7)) return EA_E_BADARG;
  return linear_dispatch(dtype, a, a_f32, w, 0, bias, y, y_f32, a_cast, rows, in_features, out_features, (long)lda,
                         (long)ldy, (hipStream_t)stream);
}

int ea_linear_w32(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* a, int32_t a_f32,
                  int64_t lda, const float* w, int32_t w_transposed, const float* bias, void* y, int32_t y_f32, int64_t ldy,
                  void* a_cast, void* stream) {
  if (!a || !w || !y || ((uintptr_t)a & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)a_cast & 15))
    return EA_E_BADARG;
  if (lda < in_features || ldy < out_features || (lda & 7) || (ldy & 7)) return EA_E_BADARG;
  // the qkv projection of a 192-wide model: weight resident in registers, activations through LDS once (ea_proj_rs.hip)
  static const bool rs_on = !(getenv("EA_PROJ_RS") && getenv("EA_PROJ_RS")[0] == '0');
  // (a workgroup first loads the whole 576 x 192 weight into its registers: worth it from ~8 token tiles per workgroup on --
  //  at N = 196 x batch 128 the LDS-resident kernel is faster, 25.7 against 30.1 us)
  if (rs_on && !w_transposed && !y_f32 && rows >= 65536 && proj_rs_supported(in_features, out_features))
    return proj_rs_dispatch(dtype, a, a_f32, w, bias, y, a_cast, rows, (long)lda, (long)ldy, (hipStream_t)stream, 0, 0, 0, 0,
                            nullptr, nullptr, nullptr, nullptr);
  return linear_dispatch(dtype, a, a_f32, w, w_transposed ? 2 : 1, bias, y, y_f32, a_cast, rows, in_features, out_features,
                         (long)lda, (long)ldy, (hipStream_t)stream);
}

// input gradient of the 192 -> 576 projection, weight resident in registers (ea_dgrad_rs.hip)
int32_t ea_linear_dgrad_supported(int32_t in_features, int32_t out_features) {
  return dgrad_rs_supported(out_features, in_features);
}

int ea_linear_dgrad(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* dy, int64_t ldy,
                    const void* w, int32_t w_f32, void* dx, int32_t dx_f32, int64_t ldx, void* stream) {
  if (!dy || !w || !dx || ((uintptr_t)dy & 15) || ((uintptr_t)w & 15) || ((uintptr_t)dx & 15)) return EA_E_BADARG;
  if (rows < 0 || ldy < out_features || ldx < in_features || (ldy & 7) || (ldx & 3) || (!dx_f32 && (ldx & 7))) return EA_E_BADARG;
  if (!dgrad_rs_supported(out_features, in_features)) return EA_E_UNSUPPORTED;
  if ((int64_t)rows * ldy >= ((int64_t)1 << 40)) return EA_E_BADARG;
  return dgrad_rs_dispatch(dtype, dy, w, w_f32, dx, dx_f32, rows, (long)ldy, (long)ldx, (hipStream_t)stream);
}

// the same product fused with the last corrections of dq / dk (ea_dgrad_rs.hip, dgrad_fin_kernel)
int ea_linear_dgrad_finish(int32_t dtype, int32_t B, int32_t gh, int32_t gw, int32_t pool_r, int32_t C, float scale,
                           void* dqkv, int64_t ldy, const void* qkv, int64_t ldq, const void* w, int32_t w_f32, void* dx,
                           int32_t dx_f32, int64_t ldx, const float* qbar, const float* uq, const float* lse_t,
                           const float* dpq, const float* dpk, void* stream) {
  if (!dqkv || !w || !dx || ((uintptr_t)dqkv & 15) || ((uintptr_t)w & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)qkv & 15))
    return EA_E_BADARG;
  if (B <= 0 || gh <= 0 || gw <= 0 || ldy < 576 || ldx < 192 || (ldy & 7) || (ldx & 3) || (!dx_f32 && (ldx & 7))) return EA_E_BADARG;
  if ((dpq == nullptr) != (dpk == nullptr)) return EA_E_BADARG;
  if (dpq && (pool_r <= 0 || gh % pool_r || gw % pool_r)) return EA_E_BADARG;
  if (uq && (!qkv || !qbar || !lse_t || ldq < 192 || (ldq & 7) || C <= 0)) return EA_E_BADARG;
  if (uq && C > 64) return EA_E_UNSUPPORTED;
  if (!uq && !dpq) return EA_E_BADARG;                        // nothing to correct: ea_linear_dgrad
  if ((int64_t)B * gh * gw >= ((int64_t)1 << 31)) return EA_E_BADARG;
  return dgrad_fin_launch(dtype, dqkv, (long)ldy, qkv, (long)ldq, w, w_f32, dx, dx_f32, (long)ldx, B, gh, gw, pool_r, C, scale,
                          qbar, uq, lse_t, dpq, dpk, (hipStream_t)stream);
}

// qkv projection + pooled q / k rows in one pass (ea_proj_rs.hip, POOL variants)
int32_t ea_linear_pool_supported(int32_t in_features, int32_t out_features, int32_t B, int32_t gh, int32_t gw, int32_t r) {
  return proj_rs_pool_supported(in_features, out_features, B, gh, gw, r);
}

int ea_linear_w32_pool(int32_t dtype, int32_t B, int32_t gh, int32_t gw, int32_t r, int32_t in_features, int32_t out_features,
                       const void* a, int32_t a_f32, int64_t lda, const float* w, const float* bias, void* y, int64_t ldy,
                       void* a_cast, float* pooled_q, float* pooled_k, void* w_cast, void* stream) {
  if (!a || !w || !y || !pooled_q || !pooled_k || ((uintptr_t)a & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) ||
      ((uintptr_t)bias & 15) || ((uintptr_t)a_cast & 15) || ((uintptr_t)pooled_q & 15) || ((uintptr_t)pooled_k & 15) ||
      ((uintptr_t)w_cast & 15))
    return EA_E_BADARG;
  if (lda < in_features || ldy < out_features || (lda & 7) || (ldy & 7)) return EA_E_BADARG;
  if (!proj_rs_pool_supported(in_features, out_features, B, gh, gw, r)) return EA_E_UNSUPPORTED;
  return proj_rs_dispatch(dtype, a, a_f32, w, bias, y, a_cast, B * gh * gw, (long)lda, (long)ldy, (hipStream_t)stream, B, gh, gw,
                          r, pooled_q, pooled_k, w_cast, nullptr);
}

int ea_linear_w192_prepare(int32_t dtype, const float* wq, const float* wp, void* w16q, void* wq_sw, void* w16p, void* w16pT,
                           void* stream) {
  if (!wq || !w16q || !wq_sw || ((uintptr_t)wq & 15) || ((uintptr_t)wp & 15) || ((uintptr_t)w16q & 15) || ((uintptr_t)wq_sw & 15) ||
      ((uintptr_t)w16p & 15) || ((uintptr_t)w16pT & 15))
    return EA_E_BADARG;
  if (wp && (!w16p || !w16pT)) return EA_E_BADARG;
  return w192_prepare_dispatch(dtype, wq, wp, w16q, wq_sw, w16p, w16pT, (hipStream_t)stream);
}

int ea_linear_wsw(int32_t dtype, int32_t rows, int32_t B, int32_t gh, int32_t gw, int32_t r, const void* a, int32_t a_f32, int64_t lda,
                  const void* wq_sw, const float* bias, void* y, int64_t ldy, void* a_cast, float* pooled_q, float* pooled_k,
                  void* stream) {
  if (!a || !wq_sw || !y || ((uintptr_t)a & 15) || ((uintptr_t)wq_sw & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)a_cast & 15) || ((uintptr_t)pooled_q & 15) || ((uintptr_t)pooled_k & 15))
    return EA_E_BADARG;
  if (lda < 192 || ldy < 576 || (lda & 7) || (ldy & 7) || rows <= 0) return EA_E_BADARG;
  if (r != 0) {
    if (!pooled_q || !pooled_k || rows != B * gh * gw) return EA_E_BADARG;
    if (!proj_rs_pool_supported(192, 576, B, gh, gw, r)) return EA_E_UNSUPPORTED;
  }
  return proj_rs_dispatch(dtype, a, a_f32, nullptr, bias, y, a_cast, rows, (long)lda, (long)ldy, (hipStream_t)stream, B, gh, gw, r,
                          r ? pooled_q : nullptr, r ? pooled_k : nullptr, nullptr, wq_sw);
}

}  // extern "C"

// ---- ScatterBrain feature half (ea_scatter.hip) ----
static T4s mks(const ea_t4* t) {
  T4s r;
  r.p = t ? (char*)t->ptr : nullptr;
  r.sb = t ? t->sb : 0; r.sh = t ? t->sh : 0; r.sn = t ? t->sn : 0;
  return r;
}
static int fill_sb(const ea_sb_geom* g, SbP& p) {
  if (!g || g->B <= 0 || g->H <= 0 || g->N <= 0 || g->window <= 0 || (g->dtype != EA_BF16 && g->dtype != EA_F16))
    return EA_E_BADARG;
  if (g->D != 64 || g->M <= 0 || g->M > 64) return EA_E_UNSUPPORTED;
  p.B = g->B; p.H = g->H; p.N = g->N; p.M = g->M; p.w = g->window;
  p.G.N = g->N; p.G.attn2d = g->attn_2d; p.G.gh = g->attn_2d ? g->gh : 1; p.G.gw = g->attn_2d ? g->gw : g->N;
  if (g->attn_2d) {
    if (g->gh % g->window || g->gw % g->window || g->gh * g->gw != g->N) return EA_E_BADARG;
    p.Wq = g->window * g->window;
    p.nwin = (g->gh / g->window) * (g->gw / g->window);
  } else {
    if (g->N % g->window) return EA_E_BADARG;
    p.Wq = g->window;
    p.nwin = g->N / g->window;
  }
  if (p.Wq > 64) return EA_E_UNSUPPORTED;
  p.a = 1.f / sqrtf(sqrtf(64.f));
  p.b = 0.5f / sqrtf(64.f);
  p.lconst = 0.5f * logf((float)g->M);
  p.wpb = 1;
  for (int d = 2; d <= 4; ++d)                      // windows per workgroup of the backward window pass
    if (p.nwin % d == 0) p.wpb = d;
  return EA_OK;
}
static int fill_sb_stats(const ea_sb_geom* g, LaraP& p) {
  ea_perf_geom pg;
  pg.B = g->B; pg.H = g->H; pg.N = g->N; pg.D = g->D; pg.dtype = g->dtype; pg.M = g->M;
  int rc = fill_perf(&pg, p, true);
  if (rc != EA_OK) return rc;
  p.ratio = 1.f; p.feps = 0.f; p.stab_per_feature = 1;
  return EA_OK;
}

extern "C" {

int32_t ea_scatter_parts(const ea_sb_geom* g) {
  LaraP p = {};
  if (!g || fill_sb_stats(g, p) != EA_OK) return EA_E_BADARG;
  return p.nsplit * lara_nsub(p.NCT);
}

int ea_scatter_kmax(const ea_sb_geom* g, const ea_t4* k, const uint8_t* mask, const float* W, float* p_ml, void* stream) {
  LaraP p = {};
  if (!g) return EA_E_BADARG;
  int rc = fill_sb_stats(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !W || !p_ml) return EA_E_BADARG;
  p.k = mkl(k); p.mask = mask; p.omega = W; p.p_ml = p_ml;
  return lara_y_dispatch(LY_PMAX, p, g->dtype, (hipStream_t)stream);
}

int ea_scatter_kv(const ea_sb_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                  const float* mx, float* p_ml, float* p_kv, void* stream) {
  LaraP p = {};
  if (!g) return EA_E_BADARG;
  int rc = fill_sb_stats(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, g->D) || !t4_ok(v, g->D) || !W || !mx || !p_ml || !p_kv) return EA_E_BADARG;
  p.k = mkl(k); p.v = mkl(v); p.mask = mask; p.omega = W; p.stab = mx; p.p_ml = p_ml; p.p_acc0 = p_kv;
  return lara_y_dispatch(LY_PKV, p, g->dtype, (hipStream_t)stream);
}

int32_t ea_scatter_bwd_parts(const ea_sb_geom* g) {
  SbP p = {};
  if (fill_sb(g, p) != EA_OK) return EA_E_BADARG;
  return p.nwin / p.wpb;
}

int ea_scatter_bwd_window(const ea_sb_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                          const float* W, const float* mx, const float* zall, const float* sall, const ea_t4* oloc,
                          const float* lse_loc, const float* r, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                          const ea_t4* dv, const ea_t4* doloc, float* dlse, float* p_dsall, float* p_dzall,
                          void* stream) {
  SbP p = {};
  int rc = fill_sb(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, 64) || !t4_ok(k, 64) || !t4_ok(v, 64) || !t4_ok(oloc, 64) || !t4_ok(dout, 64) || !t4_ok(dq, 64) ||
      !t4_ok(dk, 64) || !t4_ok(dv, 64) || !t4_ok(doloc, 64) || !W || !mx || !zall || !sall || !lse_loc || !r || !dlse ||
      !p_dsall || !p_dzall) return EA_E_BADARG;
  p.q = mks(q); p.k = mks(k); p.v = mks(v); p.oloc = mks(oloc); p.dout = mks(dout); p.dq = mks(dq); p.dk = mks(dk);
  p.dv = mks(dv); p.doloc = mks(doloc); p.mask = mask; p.Wf = W; p.mx = mx; p.zall = zall; p.sall = sall;
  p.lse_loc = lse_loc; p.r = const_cast<float*>(r); p.dlse = dlse; p.p_dsall = p_dsall; p.p_dzall = p_dzall;
  return sb_bwd_dispatch(0, p, g->dtype, (hipStream_t)stream);
}

int ea_scatter_bwd_global(const ea_sb_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                          const float* mx, const float* dsall, const float* dzall, const ea_t4* dk, const ea_t4* dv,
                          void* stream) {
  SbP p = {};
  int rc = fill_sb(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(k, 64) || !t4_ok(v, 64) || !t4_ok(dk, 64) || !t4_ok(dv, 64) || !W || !mx || !dsall || !dzall)
    return EA_E_BADARG;
  p.k = mks(k); p.v = mks(v); p.dk = mks(dk); p.dv = mks(dv); p.mask = mask; p.Wf = W; p.mx = mx;
  p.zall = mx; p.sall = dsall;                       // (unused in the global pass; kept valid)
  p.dsall = dsall; p.dzall = dzall;
  p.wpb = 1;
  return sb_bwd_dispatch(1, p, g->dtype, (hipStream_t)stream);
}

int ea_scatter_fwd(const ea_sb_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                   const float* W, const float* mx, const float* zall, const float* sall, const ea_t4* oloc,
                   const float* lse_loc, const ea_t4* out, float* r, void* stream) {
  SbP p = {};
  int rc = fill_sb(g, p);
  if (rc != EA_OK) return rc;
  if (!t4_ok(q, 64) || !t4_ok(k, 64) || !t4_ok(v, 64) || !t4_ok(oloc, 64) || !t4_ok(out, 64) || !W || !mx || !zall ||
      !sall || !lse_loc || !r) return EA_E_BADARG;
  p.q = mks(q); p.k = mks(k); p.v = mks(v); p.oloc = mks(oloc); p.out = mks(out); p.mask = mask; p.Wf = W;
  p.mx = mx; p.zall = zall; p.sall = sall; p.lse_loc = lse_loc; p.r = r;
  return sb_fwd_dispatch(p, g->dtype, (hipStream_t)stream);
}

}  // extern "C"

// ---- composite per-module entry points (round 3): one call = the whole LARA core forward, one = the whole backward ----
// The reference's call sites run eagerly (vit/engine.py:47-64); issued from Python the core was ~15 C-ABI calls and ~30
// tensor allocations per step.  Here the launch sequence lives in C++ and every intermediate sits in two caller-owned
// workspaces (`saved`: what the backward needs again; `tmp`: scratch of one direction) whose sizes ea_lara_layer_ws reports.
namespace {
struct LaraLayerPlan {
  ea_geom pg;            // pooling geometry (uniform r x r average pooling of q, k: lara.py:43,48,145-151)
  ea_lmk_geom lg;        // landmark pipeline
  ea_lara_geom g;        // estimator
  int L, C, BH, S_fwd, S_bwd;
  // offsets (floats) into the saved workspace
  size_t o_omega, o_qrows, o_bhv, o_cst, o_kv, o_lsek, o_lset, o_pq, o_pk, o_lmk, o_tok, n_saved;
  // forward scratch: lp, p_ml, p_kv
  size_t f_lp, f_ml, f_kv, n_ftmp;
  // backward scratch: p_ml, p_acc[4], big[4], small[4], d_omega, dpq, dpk, dW, dvec
  size_t b_ml, b_acc, b_big, b_small, b_dom, b_dpq, b_dpk, b_dW, b_dvec, b_domk, n_btmp;
  bool fold_f, fold_b;   // round 5: merge launches folded into their consumers (S <= 4 slices; EA_LARA_FOLD=0 restores them)
};
size_t al4(size_t n) { return (n + 3) & ~(size_t)3; }       // 16-byte aligned sub-buffers
int lara_layer_plan(const ea_lara_layer* c, LaraLayerPlan& P) {
  if (!c || c->B <= 0 || c->H <= 0 || c->gh <= 0 || c->gw <= 0 || c->pool_r <= 0 || c->gh % c->pool_r || c->gw % c->pool_r)
    return EA_E_BADARG;
  const int N = c->gh * c->gw;
  P.L = (c->gh / c->pool_r) * (c->gw / c->pool_r);
  P.C = P.L * (c->dup ? 2 : 1);
  P.BH = c->B * c->H;
  if (P.C > 64 || P.L > 64) return EA_E_UNSUPPORTED;        // larger sample counts: the step-by-step entry points
  ea_geom pg = {};
  pg.B = c->B; pg.H = c->H; pg.N = N; pg.D = c->D; pg.dtype = c->dtype; pg.attn_2d = 1; pg.gh = c->gh; pg.gw = c->gw;
  pg.window = c->pool_r; pg.ext = 0; pg.chunk = c->pool_r; pg.L = P.L; pg.scale = c->scale; pg.causal = 0; pg.lm_base = 0;
  P.pg = pg;
  ea_lmk_geom lg = {};
  lg.BH = P.BH; lg.L = P.L; lg.C = P.C; lg.D = c->D; lg.has_mlp = c->has_mlp; lg.mixed = c->mixed; lg.mis = c->mis;
  lg.dup = c->dup; lg.scale = c->scale; lg.eva = 0;
  P.lg = lg;
  ea_lara_geom g = {};
  g.B = c->B; g.H = c->H; g.N = N; g.D = c->D; g.dtype = c->dtype; g.C = P.C; g.mis = c->mis; g.kappa = c->kappa; g.scale = c->scale;
  P.g = g;
  P.S_fwd = ea_lara_parts(&g);
  P.S_bwd = ea_lara_fused_parts(&g);
  if (P.S_fwd <= 0 || P.S_bwd <= 0) return EA_E_UNSUPPORTED;
  const size_t CD = (size_t)P.BH * P.C * c->D, Cs = (size_t)P.BH * P.C, LD = (size_t)P.BH * P.L * c->D;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += al4(n); return at; };
  P.o_omega = take(CD); P.o_qrows = take(CD); P.o_bhv = take(Cs); P.o_cst = take(Cs); P.o_kv = take(CD);
  P.o_lsek = take(Cs); P.o_lset = take(Cs); P.o_pq = take(LD); P.o_pk = take(LD);
  const int64_t ls = ea_lara_landmarks_saved_floats(&lg);
  if (ls < 0) return EA_E_UNSUPPORTED;
  P.o_lmk = take((size_t)ls); P.o_tok = take((size_t)2 * P.BH * N);
  P.n_saved = o;
  o = 0;
  P.f_lp = take(Cs); P.f_ml = take(Cs * P.S_fwd * 4); P.f_kv = take(CD * P.S_fwd);
  P.n_ftmp = o;
  o = 0;
  P.b_ml = take(Cs * P.S_bwd * 4); P.b_acc = take(4 * CD * P.S_bwd); P.b_big = take(4 * CD); P.b_small = take(4 * Cs);
  P.b_dom = take(CD); P.b_dpq = take(LD); P.b_dpk = take(LD);
  P.b_dW = take((size_t)P.BH * 2 * c->D * c->D); P.b_dvec = take((size_t)P.BH * 6 * c->D);
  const char* fold_env = getenv("EA_LARA_FOLD");             // (read per call: the tests switch it inside one process)
  const bool fold_on = !(fold_env && fold_env[0] == '0');
  P.fold_f = fold_on && P.S_fwd <= 4;
  P.fold_b = fold_on && P.S_bwd <= 4;
  // reserved whatever the switch says: the caller memoises the workspace sizes per geometry, and a size that followed the
  // environment would let a later call with the fold ON write p_domk past a buffer sized with it OFF
  P.b_domk = take(P.S_bwd <= 4 ? CD * P.S_bwd : 0);
  P.n_btmp = o;
  return EA_OK;
}
}  // namespace

extern "C" {

int64_t ea_lara_layer_ws(const ea_lara_layer* c, int32_t which) {
  LaraLayerPlan P;
  const int rc = lara_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  switch (which) {
    case 0: return (int64_t)P.n_saved;
    case 1: return (int64_t)P.n_ftmp;
    case 2: return (int64_t)P.n_btmp;
    case 3: return (int64_t)P.o_pq;
    case 4: return (int64_t)P.o_pk;
    case 5: return (int64_t)P.b_dW;
    case 6: return (int64_t)P.b_dvec;
    // round 5 (ea_lara_layer_bwd2 with EA_LARA_DEFER_FINISH): what ea_linear_dgrad_finish needs -- in the backward scratch
    // u qbar rows (7), d(pooled q) (8), d(pooled k) (9); in `saved` the qbar rows (10) and lse_t (11)
    case 7: return (int64_t)(P.b_big + 3 * (size_t)P.BH * P.C * c->D);
    case 8: return (int64_t)P.b_dpq;
    case 9: return (int64_t)P.b_dpk;
    case 10: return (int64_t)P.o_qrows;
    case 11: return (int64_t)P.o_lset;
    default: return EA_E_BADARG;
  }
}

int ea_lara_layer_fwd(const ea_lara_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                      const float* noise, const float* const* params, const ea_t4* out, float* saved, float* tmp,
                      int32_t keep_for_backward, void* stream) {
  LaraLayerPlan P;
  int rc = lara_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  if (!saved || !tmp || (c->has_mlp && !params)) return EA_E_BADARG;
  const float* pr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (c->has_mlp) for (int i = 0; i < 8; ++i) pr[i] = params[i];
  const bool opt = c->mis == EA_MIS_OPT;
  float* omega = saved + P.o_omega;
  float* qrows = c->mis != EA_MIS_BH ? saved + P.o_qrows : nullptr;
  float* bhv = opt ? saved + P.o_bhv : nullptr;
  float* lse_t = opt ? saved + P.o_lset : nullptr;
  float* pq = saved + P.o_pq; float* pk = saved + P.o_pk;
  if (!(keep_for_backward & EA_LARA_POOLED_READY)) {
    rc = ea_eva_chunk_mean_fwd(&P.pg, q, k, nullptr, pq, pk, stream);
    if (rc != EA_OK) return rc;
  }
  keep_for_backward &= 1;
  rc = ea_lara_landmarks_fwd(&P.lg, pq, pk, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], noise, omega, qrows, bhv,
                             tmp + P.f_lp, keep_for_backward ? saved + P.o_lmk : nullptr, stream);
  if (rc != EA_OK) return rc;
  rc = ea_lara_stats_fwd(&P.g, q, k, v, mask, omega, qrows, tmp + P.f_ml, tmp + P.f_kv, stream);
  if (rc != EA_OK) return rc;
  float* tok = keep_for_backward ? saved + P.o_tok : nullptr;
  if (P.fold_f)
    return ea_lara_out_fwd_merge(&P.g, q, omega, qrows, bhv, P.S_fwd, tmp + P.f_ml, tmp + P.f_kv, tmp + P.f_lp, saved + P.o_kv,
                                 saved + P.o_lsek, lse_t, saved + P.o_cst, out, tok, tok ? tok + (size_t)P.BH * P.g.N : nullptr,
                                 stream);
  rc = ea_lara_merge_fwd(P.BH, P.S_fwd, P.C, c->D, opt ? 1 : 0, tmp + P.f_ml, tmp + P.f_kv, tmp + P.f_lp, saved + P.o_kv,
                         saved + P.o_lsek, lse_t, saved + P.o_cst, stream);
  if (rc != EA_OK) return rc;
  return ea_lara_out_fwd(&P.g, q, omega, qrows, saved + P.o_kv, lse_t, bhv, saved + P.o_cst, out, tok,
                         tok ? tok + (size_t)P.BH * P.g.N : nullptr, stream);
}

int ea_lara_layer_bwd(const ea_lara_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                      const float* noise, const float* const* params, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                      const ea_t4* dv, const float* saved, float* tmp, float* dparams, void* stream) {
  return ea_lara_layer_bwd2(c, q, k, v, mask, noise, params, dout, dq, dk, dv, saved, tmp, dparams, 0, stream);
}

int ea_lara_layer_bwd2(const ea_lara_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* noise, const float* const* params, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                       const ea_t4* dv, const float* saved, float* tmp, float* dparams, int32_t flags, void* stream) {
  const bool defer_finish = (flags & EA_LARA_DEFER_FINISH) != 0;
  LaraLayerPlan P;
  int rc = lara_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  if (!saved || !tmp || (c->has_mlp && !params)) return EA_E_BADARG;
  const float* pr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (c->has_mlp) for (int i = 0; i < 8; ++i) pr[i] = params[i];
  const bool opt = c->mis == EA_MIS_OPT;
  const int D = c->D;
  const size_t CD = (size_t)P.BH * P.C * D, Cs = (size_t)P.BH * P.C;
  const float* omega = saved + P.o_omega;
  const float* qrows = c->mis != EA_MIS_BH ? saved + P.o_qrows : nullptr;
  const float* bhv = opt ? saved + P.o_bhv : nullptr;
  const float* lse_t = opt ? saved + P.o_lset : nullptr;
  const float* kv = saved + P.o_kv;
  const float* tok = saved + P.o_tok;
  float* p_ml = tmp + P.b_ml;
  float* acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = tmp + P.b_acc + (size_t)i * CD * P.S_bwd;
  float *dkv = tmp + P.b_big, *dom_q = dkv + CD, *dqbar_m = dom_q + CD, *uq = dqbar_m + CD;
  float *r = tmp + P.b_small, *dbh = r + Cs, *dlp_m = dbh + Cs, *dkk = dlp_m + Cs;
  rc = ea_lara_bwd_q_fused(&P.g, q, dout, omega, qrows, kv, lse_t, bhv, saved + P.o_cst, tok, tok + (size_t)P.BH * P.g.N, dq, p_ml,
                           acc[0], acc[1], acc[2], acc[3], stream);
  if (rc != EA_OK) return rc;
  const bool want_dqbar = c->mis == EA_MIS_OPT || c->mis == EA_MIS_BIASED;
  if (P.fold_b) {
    // the key-side pass merges the query side's partials itself and block 0 of each (b,h) writes dbh, dlp, sum dZ q, d qbar
    // rows and u qbar; the landmark backward adds the key side's d omega partials while it loads its strips
    float* domk = tmp + P.b_domk;
    rc = ea_lara_bwd_k_fused_merge(&P.g, k, v, mask, omega, qrows, kv, saved + P.o_lsek, P.S_bwd, p_ml, acc[0], acc[1], acc[2],
                                   acc[3], dk, dv, domk, opt ? dbh : nullptr, dlp_m, dom_q, want_dqbar ? dqbar_m : nullptr,
                                   opt ? uq : nullptr, stream);
    if (rc != EA_OK) return rc;
    float *dpq = tmp + P.b_dpq, *dpk = tmp + P.b_dpk;
    float* dW = c->has_mlp ? tmp + P.b_dW : nullptr;
    float* dvec = c->has_mlp ? tmp + P.b_dvec : nullptr;
    rc = ea_lara_landmarks_bwd_parts(&P.lg, saved + P.o_pq, saved + P.o_pk, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7],
                                     noise, dom_q, P.S_bwd, domk, c->scale, want_dqbar ? dqbar_m : nullptr, opt ? dbh : nullptr,
                                     dlp_m, dpq, dpk, dW, dvec, saved + P.o_lmk, stream);
    if (rc != EA_OK) return rc;
    if (!defer_finish) {
      rc = ea_lara_bwd_finish(&P.g, q, qrows, opt ? uq : nullptr, lse_t, dpq, dpk, c->pool_r, c->gh, c->gw, dq, dk, stream);
      if (rc != EA_OK) return rc;
    }
    if (c->has_mlp && dparams) rc = ea_colsum2_f32(P.BH, 2 * D * D, dW, dparams, 6 * D, dvec, dparams + (size_t)2 * D * D, stream);
    return rc;
  }
  rc = ea_lara_merge_bwd(P.BH, P.S_bwd, P.C, D, opt ? 1 : 0, c->scale, p_ml, acc[0], acc[1], acc[2], acc[3], kv, qrows, r, dbh,
                         dlp_m, dkk, dkv, dom_q, want_dqbar ? dqbar_m : nullptr, opt ? uq : nullptr, stream);
  if (rc != EA_OK) return rc;
  rc = ea_lara_bwd_k_fused(&P.g, k, v, mask, omega, dkv, saved + P.o_lsek, dkk, r, dk, dv, acc[1], stream);
  if (rc != EA_OK) return rc;
  float* d_omega = tmp + P.b_dom;
  rc = ea_slice_sum(P.BH, P.S_bwd, P.C * D, c->scale, dom_q, acc[1], d_omega, stream);
  if (rc != EA_OK) return rc;
  float *dpq = tmp + P.b_dpq, *dpk = tmp + P.b_dpk;
  float* dW = c->has_mlp ? tmp + P.b_dW : nullptr;
  float* dvec = c->has_mlp ? tmp + P.b_dvec : nullptr;
  rc = ea_lara_landmarks_bwd(&P.lg, saved + P.o_pq, saved + P.o_pk, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], noise,
                             d_omega, want_dqbar ? dqbar_m : nullptr, opt ? dbh : nullptr, dlp_m, dpq, dpk, dW, dvec,
                             saved + P.o_lmk, stream);
  if (rc != EA_OK) return rc;
  if (!defer_finish) {
    rc = ea_lara_bwd_finish(&P.g, q, qrows, opt ? uq : nullptr, lse_t, dpq, dpk, c->pool_r, c->gh, c->gw, dq, dk, stream);
    if (rc != EA_OK) return rc;
  }
  // dparams == NULL: the caller adds the per-(b,h) partials up itself (tmp + ea_lara_layer_ws(cfg, 5 / 6): [B*H, 2 D D] and
  // [B*H, 6 D]), e.g. together with other terminal sums of its backward in one ea_multi_sum launch
  if (c->has_mlp && dparams) rc = ea_colsum2_f32(P.BH, 2 * D * D, dW, dparams, 6 * D, dvec, dparams + (size_t)2 * D * D, stream);
  return rc;
}

}  // extern "C"

// ---- composite per-module entry points, EVA (round 4): the 2-D core of eva.py:145-227 in one call each way ----
// chunk means (or the projection kernel's) -> mu networks + omega (fused landmark kernel, eva mode) -> beta -> window
// attention with the control-variate columns; backward: window backward -> landmark-gradient slice sum (+ bias-gradient
// column sum) -> beta backward -> mu-network backward -> chunk-mean backward (-> parameter sums).  Same contract as the
// LARA pair: caller-owned workspaces, sizes from ea_eva_layer_ws.
namespace {
struct EvaLayerPlan {
  ea_geom g;
  ea_lmk_geom lg;
  int L, BH, N, Wq, ld, parts, bparts;
  size_t o_lse, o_qm, o_km, o_omega, o_beta, o_rfk, o_lmk, n_saved;
  size_t b_dlp, b_dl, b_dbp, b_dom, b_dqm, b_dkm, b_dW, b_dvec, n_btmp;
};
int eva_layer_plan(const ea_eva_layer* c, EvaLayerPlan& P) {
  if (!c || c->B <= 0 || c->H <= 0 || c->gh <= 0 || c->gw <= 0 || c->window <= 0 || c->chunk <= 0 ||
      c->gh % c->window || c->gw % c->window || c->gh % c->chunk || c->gw % c->chunk) return EA_E_BADARG;
  if (c->D != 32 && c->D != 64) return EA_E_UNSUPPORTED;
  P.N = c->gh * c->gw;
  P.L = (c->gh / c->chunk) * (c->gw / c->chunk);
  P.BH = c->B * c->H;
  if (P.L > 64) return EA_E_UNSUPPORTED;                    // the fused mu / omega kernel; larger: the step-by-step entries
  ea_geom g = {};
  g.B = c->B; g.H = c->H; g.N = P.N; g.D = c->D; g.dtype = c->dtype; g.attn_2d = 1; g.gh = c->gh; g.gw = c->gw;
  g.window = c->window; g.ext = 0; g.chunk = c->chunk; g.L = P.L; g.scale = c->scale; g.causal = 0; g.lm_base = 0;
  P.g = g;
  ea_lmk_geom lg = {};
  lg.BH = P.BH; lg.L = P.L; lg.C = P.L; lg.D = c->D; lg.has_mlp = 1; lg.mixed = 0; lg.mis = 0; lg.dup = 0;
  lg.scale = c->scale; lg.eva = 1;
  P.lg = lg;
  P.Wq = c->window * c->window;
  P.ld = ea_window_bias_ld(&g);
  P.parts = ea_window_bwd_parts(&g);
  P.bparts = c->has_bias ? ea_window_bwd_bias_parts(&g) : 0;
  if (P.ld <= 0 || P.parts <= 0 || (c->has_bias && P.bparts <= 0)) return EA_E_UNSUPPORTED;
  // geometries whose backward needs scratch slices, several query blocks or the transposed bias copy: step-by-step entries
  if (ea_window_bwd_acc_slices(&g) != 0 || ea_window_bwd_query_blocks(&g) != 1 ||
      (c->has_bias && ea_window_bwd_needs_bias_t(&g) != 0)) return EA_E_UNSUPPORTED;
  const size_t LD = (size_t)P.BH * P.L * c->D;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += al4(n); return at; };
  P.o_lse = take((size_t)P.BH * P.N);
  P.o_qm = take(LD); P.o_km = take(LD); P.o_omega = take(LD); P.o_beta = take(LD); P.o_rfk = take(LD);
  const int64_t ls = ea_lara_landmarks_saved_floats(&lg);
  if (ls < 0) return EA_E_UNSUPPORTED;
  P.o_lmk = take((size_t)ls);
  P.n_saved = o;
  o = 0;
  P.b_dlp = take((size_t)2 * P.parts * LD); P.b_dl = take(2 * LD);
  P.b_dbp = take((size_t)P.bparts * c->B * c->H * P.Wq * P.ld);
  P.b_dom = take(LD); P.b_dqm = take(LD); P.b_dkm = take(LD);
  P.b_dW = take((size_t)P.BH * 2 * c->D * c->D); P.b_dvec = take((size_t)P.BH * 6 * c->D);
  P.n_btmp = o;
  return EA_OK;
}
}  // namespace

extern "C" {

int64_t ea_eva_layer_ws(const ea_eva_layer* c, int32_t which) {
  EvaLayerPlan P;
  const int rc = eva_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  switch (which) {
    case 0: return (int64_t)P.n_saved;
    case 1: return 0;
    case 2: return (int64_t)P.n_btmp;
    case 3: return (int64_t)P.o_qm;
    case 4: return (int64_t)P.o_km;
    case 5: return (int64_t)P.b_dW;
    case 6: return (int64_t)P.b_dvec;
    case 7: return (int64_t)P.ld;
    case 8: return (int64_t)P.o_lse;
    case 9: return (int64_t)P.b_dbp;
    case 10: return (int64_t)P.bparts * c->B;
    // round 5 (ea_eva_layer_bwd2 with EA_EVA_DEFER_CHUNK_MEAN): the chunk-mean gradients ea_linear_dgrad_finish adds to dq / dk
    case 11: return (int64_t)P.b_dqm;
    case 12: return (int64_t)P.b_dkm;
    default: return EA_E_BADARG;
  }
}

int ea_eva_layer_fwd(const ea_eva_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                     const float* noise, const float* const* params, const ea_t4* out, float* saved,
                     int32_t keep_for_backward, void* stream) {
  EvaLayerPlan P;
  int rc = eva_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  if (!saved || !params || (c->has_bias != 0) != (bias != nullptr)) return EA_E_BADARG;
  float *qm = saved + P.o_qm, *km = saved + P.o_km, *omega = saved + P.o_omega, *beta = saved + P.o_beta, *rfk = saved + P.o_rfk;
  if (!(keep_for_backward & EA_LARA_POOLED_READY)) {
    rc = ea_eva_chunk_mean_fwd(&P.g, q, k, nullptr, qm, km, stream);
    if (rc != EA_OK) return rc;
  }
  keep_for_backward &= 1;
  rc = ea_lara_landmarks_fwd(&P.lg, qm, km, params[0], params[1], params[2], params[3], params[4], params[5], params[6],
                             params[7], noise, omega, rfk, nullptr, nullptr, keep_for_backward ? saved + P.o_lmk : nullptr, stream);
  if (rc != EA_OK) return rc;
  rc = ea_eva_beta_fwd(&P.g, k, v, nullptr, omega, beta, stream);
  if (rc != EA_OK) return rc;
  return ea_window_attn_fwd(&P.g, q, k, v, rfk, beta, bias, nullptr, out, saved + P.o_lse, nullptr, 1.f, stream);
}

int ea_eva_layer_bwd(const ea_eva_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                     const float* noise, const float* const* params, const ea_t4* out, const ea_t4* dout, const ea_t4* dq,
                     const ea_t4* dk, const ea_t4* dv, const float* saved, float* tmp, float* dbias, float* dparams,
                     void* stream) {
  return ea_eva_layer_bwd2(c, q, k, v, bias, noise, params, out, dout, dq, dk, dv, saved, tmp, dbias, dparams, 0, stream);
}

int ea_eva_layer_bwd2(const ea_eva_layer* c, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                      const float* noise, const float* const* params, const ea_t4* out, const ea_t4* dout, const ea_t4* dq,
                      const ea_t4* dk, const ea_t4* dv, const float* saved, float* tmp, float* dbias, float* dparams,
                      int32_t flags, void* stream) {
  EvaLayerPlan P;
  int rc = eva_layer_plan(c, P);
  if (rc != EA_OK) return rc;
  if (!saved || !tmp || !params || (c->has_bias != 0) != (bias != nullptr)) return EA_E_BADARG;
  const int D = c->D;
  const size_t LD = (size_t)P.BH * P.L * D;
  const float *qm = saved + P.o_qm, *km = saved + P.o_km, *omega = saved + P.o_omega, *beta = saved + P.o_beta, *rfk = saved + P.o_rfk;
  float* dl_p = tmp + P.b_dlp;
  float* dl = tmp + P.b_dl;                 // [2][B*H, L, D]: d rf_k_bar, d beta
  float* dbp = c->has_bias ? tmp + P.b_dbp : nullptr;
  rc = ea_window_attn_bwd(&P.g, q, k, v, rfk, beta, bias, nullptr, out, dout, saved + P.o_lse, dq, dk, dv, dl_p,
                          dl_p + (size_t)P.parts * LD, dbp, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, stream);
  if (rc != EA_OK) return rc;
  // round 6: up to four slice partials are added up by their two consumers while they load them (ea_slice_sum's order of
  // additions: bit-identical) -- one launch less; more slices keep the reduction launch (EA_EVA_FOLD_SLICES=0: always)
  static const bool fold_on = !(getenv("EA_EVA_FOLD_SLICES") && getenv("EA_EVA_FOLD_SLICES")[0] == '0');
  const bool fold = fold_on && P.parts <= 4;
  if (!fold) {
    rc = ea_slice_sum(2, P.parts, (int32_t)LD, 1.f, nullptr, dl_p, dl, stream);
    if (rc != EA_OK) return rc;
  }
  if (c->has_bias && dbias) {        // dbias == NULL: the partials stay in tmp for the caller's reduction (selectors 9 / 10)
    rc = ea_colsum_f32(P.bparts * c->B, c->H * P.Wq * P.ld, dbp, dbias, stream);
    if (rc != EA_OK) return rc;
  }
  float* d_omega = tmp + P.b_dom;
  rc = fold ? eva_beta_bwd_parts(&P.g, k, v, nullptr, omega, beta, dl_p + (size_t)P.parts * LD, P.parts, (long)LD, dk, dv, d_omega, stream)
            : ea_eva_beta_bwd(&P.g, k, v, nullptr, omega, beta, dl + LD, dk, dv, d_omega, stream);
  if (rc != EA_OK) return rc;
  float *dqm = tmp + P.b_dqm, *dkm = tmp + P.b_dkm, *dW = tmp + P.b_dW, *dvec = tmp + P.b_dvec;
  rc = lara_landmarks_bwd_qparts(&P.lg, qm, km, params[0], params[1], params[2], params[3], params[4], params[5], params[6],
                                 params[7], noise, d_omega, fold ? dl_p : dl, fold ? P.parts : 1, fold ? (long)LD : 0, nullptr,
                                 nullptr, dqm, dkm, dW, dvec, saved + P.o_lmk, stream);
  if (rc != EA_OK) return rc;
  if (!(flags & EA_EVA_DEFER_CHUNK_MEAN)) {
    rc = ea_eva_chunk_mean_bwd(&P.g, dqm, dkm, nullptr, dq, dk, stream);
    if (rc != EA_OK) return rc;
  }
  // dparams == NULL: the per-(b,h) partials stay in tmp (offsets ea_eva_layer_ws(cfg, 5 / 6)) for the caller's own reduction
  if (dparams) rc = ea_colsum2_f32(P.BH, 2 * D * D, dW, dparams, 6 * D, dvec, dparams + (size_t)2 * D * D, stream);
  return rc;
}

}  // extern "C"

// ---- fp32-faithful gathered attention (ea_f32_attn.hip) ----
static bool f32_t4_ok(const ea_t4* t, int D) {
  return t && t->ptr && ((uintptr_t)t->ptr % 16 == 0) && t->sb % 4 == 0 && t->sh % 4 == 0 && t->sn % 4 == 0 && t->sn >= D;
}
static F32T f32_mk(const ea_t4* t) {
  F32T r;
  r.p = t ? (const float*)t->ptr : nullptr;
  r.sb = t ? t->sb : 0; r.sh = t ? t->sh : 0; r.sn = t ? t->sn : 0;
  return r;
}
static int fill_ga(const ea_f32_attn* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const ea_t4* ek, const ea_t4* ev,
                   const int32_t* idx_q, const int32_t* idx_k, const float* bias, const uint8_t* kmask, const uint8_t* qmask,
                   const uint8_t* keep, GaP& p) {
  if (!g || g->B <= 0 || g->H <= 0 || g->Nq <= 0 || g->Nk <= 0 || g->G <= 0 || g->Wq <= 0 || g->Wk < 0 || g->L < 0 ||
      g->Wk + g->L <= 0 || !idx_q || (g->Wk > 0 && !idx_k)) return EA_E_BADARG;
  if (g->D != 32 && g->D != 64 && g->D != 128) return EA_E_UNSUPPORTED;
  if (!f32_t4_ok(q, g->D) || !f32_t4_ok(k, g->D) || !f32_t4_ok(v, g->D)) return EA_E_BADARG;
  if (g->L > 0 && (!f32_t4_ok(ek, g->D) || !f32_t4_ok(ev, g->D))) return EA_E_BADARG;
  if (bias && g->bias_ld < g->Wk) return EA_E_BADARG;
  if (keep && g->keep_ld < g->Wk + g->L) return EA_E_BADARG;
  p.q = f32_mk(q); p.k = f32_mk(k); p.v = f32_mk(v); p.ek = f32_mk(g->L > 0 ? ek : nullptr); p.ev = f32_mk(g->L > 0 ? ev : nullptr);
  p.idx_q = idx_q; p.idx_k = idx_k; p.bias = bias; p.bias_hs = g->bias_hs; p.bias_bs = g->bias_bs; p.bias_ld = g->bias_ld;
  p.kmask = kmask; p.qmask = qmask; p.keep = keep; p.keep_ld = g->keep_ld; p.keep_scale = g->keep_scale;
  p.B = g->B; p.H = g->H; p.Nq = g->Nq; p.Nk = g->Nk; p.D = g->D; p.G = g->G; p.Wq = g->Wq; p.Wk = g->Wk; p.L = g->L;
  p.knorm = g->knorm & 1; p.zero_mv = (g->knorm >> 1) & 1; p.neg_inf = g->neg_inf; p.causal_e = g->causal_e; p.chunk = g->chunk; p.lm_base = g->lm_base;
  p.scale = g->scale;
  return EA_OK;
}

extern "C" {

int ea_f32_attn_fwd(const ea_f32_attn* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const ea_t4* ek, const ea_t4* ev,
                    const int32_t* idx_q, const int32_t* idx_k, const float* bias, const uint8_t* kmask, const uint8_t* qmask,
                    const uint8_t* keep, const ea_t4* out, float* lse, float* stat, void* stream) {
  GaP p = {};
  int rc = fill_ga(g, q, k, v, ek, ev, idx_q, idx_k, bias, kmask, qmask, keep, p);
  if (rc != EA_OK) return rc;
  if (!f32_t4_ok(out, g->D)) return EA_E_BADARG;
  p.o = f32_mk(out); p.lse = lse; p.stat = stat;
  return ga_dispatch(false, p, (hipStream_t)stream);
}

int ea_f32_attn_bwd(const ea_f32_attn* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const ea_t4* ek, const ea_t4* ev,
                    const int32_t* idx_q, const int32_t* idx_k, const float* bias, const uint8_t* kmask, const uint8_t* qmask,
                    const uint8_t* keep, const ea_t4* out, const ea_t4* dout, const float* stat, const float* dlse,
                    const ea_t4* dq, float* dk, float* dv, float* dek, float* dev, float* dbias, void* stream) {
  GaP p = {};
  int rc = fill_ga(g, q, k, v, ek, ev, idx_q, idx_k, bias, kmask, qmask, keep, p);
  if (rc != EA_OK) return rc;
  if (!f32_t4_ok(out, g->D) || !f32_t4_ok(dout, g->D) || !f32_t4_ok(dq, g->D) || !stat) return EA_E_BADARG;
  if (dbias && !bias) return EA_E_BADARG;
  p.o = f32_mk(out); p.dout = f32_mk(dout); p.dq = f32_mk(dq); p.stat = const_cast<float*>(stat); p.dlse = dlse;
  p.dk = dk; p.dv = dv; p.dek = g->L > 0 ? dek : nullptr; p.dev = g->L > 0 ? dev : nullptr; p.dbias = dbias;
  return ga_dispatch(true, p, (hipStream_t)stream);
}

int ea_f32_gather_mean_fwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t Cn, int32_t J, const ea_t4* x, const int32_t* idx,
                           const uint8_t* mask, float* mean, void* stream) {
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || (D & 3) || Cn <= 0 || J <= 0 || !f32_t4_ok(x, D) || !idx || !mean) return EA_E_BADARG;
  GmP p = {};
  p.x = f32_mk(x); p.idx = idx; p.mask = mask; p.mean = mean; p.B = B; p.H = H; p.N = N; p.D = D; p.Cn = Cn; p.J = J;
  return gm_dispatch(false, p, (hipStream_t)stream);
}

int ea_f32_gather_mean_bwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t Cn, int32_t J, const int32_t* idx,
                           const uint8_t* mask, const float* dmean, float* dx, void* stream) {
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || (D & 3) || Cn <= 0 || J <= 0 || !idx || !dmean || !dx) return EA_E_BADARG;
  GmP p = {};
  p.idx = idx; p.mask = mask; p.dmean = dmean; p.dx = dx; p.B = B; p.H = H; p.N = N; p.D = D; p.Cn = Cn; p.J = J;
  return gm_dispatch(true, p, (hipStream_t)stream);
}

}  // extern "C"

// ---- row LayerNorm (ea_layernorm.hip) ----
extern "C" {

int32_t ea_layernorm_parts(int32_t rows) { return ea::layernorm_parts(rows); }

int ea_layernorm_fwd(int32_t xtype, int32_t rows, int32_t C, const void* x, const float* gamma, const float* beta, float eps,
                     float* y, float* stats, void* stream) {
  if (!x || !gamma || !beta || !y) return EA_E_BADARG;
  ea::LnP p = {};
  p.x = x; p.gamma = gamma; p.beta = beta; p.y = y; p.stats = stats; p.rows = rows; p.C = C; p.eps = eps;
  return ea::layernorm_dispatch(false, p, xtype, (hipStream_t)stream);
}

int ea_layernorm_bwd(int32_t xtype, int32_t rows, int32_t C, const void* x, const float* gamma, const float* stats,
                     const float* dy, void* dx, float* part, void* stream) {
  if (!x || !gamma || !stats || !dy || !dx || !part) return EA_E_BADARG;
  ea::LnP p = {};
  p.x = x; p.gamma = gamma; p.stats = const_cast<float*>(stats); p.dy = dy; p.dx = dx; p.part = part; p.rows = rows; p.C = C;
  return ea::layernorm_dispatch(true, p, xtype, (hipStream_t)stream);
}

}  // extern "C"
