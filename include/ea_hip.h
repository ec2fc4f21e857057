/*
 * ea_hip.h -- C ABI of libea_hip.so: the MI355X (gfx950) attention hot path of
 * HKUNLP/efficient-attention.
 *
 * The reference has no FFI: its hot path is chains of torch ops inside
 *   efficient_attention/{abstract_attention,local_attention,eva,lara,kernelized_attention}.py
 * (SURVEY.md 8a/8b).  Each entry point below replaces one such chain with hand-written HIP
 * kernels; the citation on every function names the reference lines it replaces.  The host
 * mirror of the reference's nn.Module / AttentionFactory surface
 * (efficient-attention_amd/efficient_attention) binds these symbols with ctypes and passes raw
 * device pointers -- no torch types cross this boundary (INTEGRATION.md shows the binding a
 * reference maintainer would add).
 *
 * Conventions
 *   - All pointers are DEVICE pointers.  `stream` is a hipStream_t (NULL = default stream).
 *   - q/k/v/out-like tensors are logical [B, H, N, D] with the last dim contiguous and
 *     arbitrary ELEMENT strides for the other three (ea_t4), so the [B, N, 3, H, D] output of
 *     the fused qkv Linear and the [B, N, H, D] input of the output projection are addressed
 *     in place (no permute/contiguous copies: abstract_attention.py:72-78,86).
 *   - I/O element type: EA_BF16 or EA_F16 (ea_geom.dtype).  MFMA operands have that type,
 *     accumulation / softmax / statistics are fp32 (the autocast contract of vit/engine.py:47).
 *   - key_padding_mask: uint8 [B, N], 1 = padded key, or NULL.
 *   - Every function returns 0 on success, a negative EA_E* code on invalid arguments or
 *     unsupported geometry, or a positive hipError_t from the launch.  Nothing is ever
 *     computed on the host.
 */
#ifndef EA_HIP_H
#define EA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EA_BF16 0
#define EA_F16  1
#define EA_F32  2          /* accepted by the ea_performer_f32_* entry points only */

#define EA_OK            0
#define EA_E_BADARG     -1   /* null pointer / inconsistent sizes */
#define EA_E_UNSUPPORTED -2  /* geometry outside what the kernels are built for */

/* logical [B,H,N,D] tensor, D contiguous */
typedef struct {
  void*   ptr;
  int64_t sb, sh, sn;        /* element strides of batch, head, token */
} ea_t4;

/* Window / landmark geometry shared by the local, EVA and LARA entry points. */
typedef struct {
  int32_t B, H, N, D;        /* N = tokens per (b,h) AFTER EVA's 1-D padding (eva.py:127-136) */
  int32_t dtype;             /* EA_BF16 | EA_F16 */
  int32_t attn_2d;           /* 1: tokens form a gh x gw grid (row-major), 0: 1-D sequence */
  int32_t gh, gw;            /* grid (2-D); ignored in 1-D */
  int32_t window;            /* window side w (local_attention.py:36) */
  int32_t ext;               /* overlap extension e = max(1, w/2) or 0 (local_attention.py:38-41) */
  int32_t chunk;             /* EVA landmark chunk side r (eva.py:155-158); 0 when unused */
  int32_t L;                 /* number of landmarks / chunks actually produced */
  float   scale;             /* D^-0.5 */
  int32_t causal;            /* 0: symmetric windows (eva.py, local_attention.py).  causal_eva.py (1-D only):
                              * 1: the keys of window g are [g*w - e, g*w + w) (causal_window_1d_partition,
                              *    causal_eva.py:104-116), the landmark chunks carry no extension (:688-694)
                              *    and the local logits of padded QUERIES are masked as well (:742-755);
                              * 2: 1 plus the causal masks -- local key j of query i is masked when
                              *    j > i + e (:767-773) and landmark c unless c < token / chunk (:716-738).
                              * Masked logits are REPLACED by -5e4 (masked_fill), their gradient is zero. */
  int32_t lm_base;           /* causal == 2 only: landmarks that lie BEFORE the tokens of this call and are visible to
                              * every query -- landmark c is masked unless c < lm_base + token / chunk.  0 for a whole
                              * sequence; incremental decoding (causal_eva.py:537-665) runs the suffix [previous window,
                              * current window] against the landmarks of all completed chunks with lm_base = first chunk
                              * of that suffix. */
} ea_geom;

/* ---- library info -------------------------------------------------------------------- */
const char* ea_version(void);                 /* "ea_hip <semver> gfx950" */
int32_t     ea_abi_version(void);             /* bumped on any signature change */

/* ---- EVA landmark statistics (eva.py:155-196) ------------------------------------------
 * ea_eva_chunk_mean_fwd: masked means of q and k over every landmark chunk (chunk side r,
 *   same overlap extension e as the windows; masked/out-of-range slots count as zeros in the
 *   mean: rf_w_q.masked_fill(...).mean(-2), eva.py:174-180).  qmean,kmean: fp32 [B,H,L,D].
 * ea_eva_chunk_mean_bwd: dq[b,h,n,:] += sum_{c ni n} dqmean[c]/J (same for k); dq/dk are
 *   ACCUMULATED INTO (I/O dtype, fp32 math).
 * ea_eva_beta_fwd: beta_c = sum_j softmax_j(s*omega_c.k_j - s|k_j|^2/2, -5e4 on masked) v_j
 *   (prm_projection normalize=False, attn_utils.py:324-347; eva.py:192-196). omega, beta: fp32
 *   [B,H,L,D].
 * ea_eva_beta_bwd: given dbeta, ACCUMULATES dk, dv and writes domega (fp32 [B,H,L,D]). */
int ea_eva_chunk_mean_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const uint8_t* mask,
                          float* qmean, float* kmean, void* stream);
int ea_eva_chunk_mean_bwd(const ea_geom* g, const float* dqmean, const float* dkmean,
                          const uint8_t* mask, const ea_t4* dq, const ea_t4* dk, void* stream);
int ea_eva_beta_fwd(const ea_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* omega, float* beta, void* stream);
int ea_eva_beta_bwd(const ea_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* omega, const float* beta, const float* dbeta,
                    const ea_t4* dk, const ea_t4* dv, float* domega, void* stream);

/* ---- window attention with control-variate columns -----------------------------------------
 * One joint softmax per query over [ local window keys | L landmark keys ]:
 *     logits_local = s q.k_j + bias[h,i,j], masked_fill(-5e4)   (eva.py:204-218,
 *                                                                local_attention.py:159-170)
 *     logits_cv    = s q.lk_c                                    (eva.py:200)
 *     out = P_local V_window + P_cv lv                           (eva.py:222-227)
 * With L == 0 this is LocalAttention._apply_attention (local_attention.py:134-182).
 *   lk, lv : fp32 [B,H,L,D] (rf_k_bar and beta), may be NULL iff L == 0
 *   bias   : fp32 [H, Wq, ea_window_bias_ld(g)] dense per-head bias MULTIPLIED BY log2(e) (the
 *            softmax runs in the log2 domain), rows padded, or NULL; dbias_part is d/d(natural bias)
 *   lse    : fp32 [B,H,N] joint log-sum-exp (natural log), saved for backward; a caller that merges this
 *            attention with further softmax columns of its own (ScatterBrain's low-rank part) uses it
 *            as an output and passes its gradient `dlse` [B,H,N] to the backward (NULL otherwise)
 * With ea_geom.causal the windows, masks and chunk visibility follow causal_eva.py:666-783 (see the
 * field's comment); Wk = window + ext there.
 *   keep   : attention dropout of causal_eva.py:778 (`attn = dropout(attn)` on the joint softmax):
 *            uint8 [B,H,N, ea_window_keep_ld(g)] -- for query n, column j < Wk is local slot j and
 *            column ea_window_bias_ld(g) + c is landmark c; non-zero = kept.  Kept probabilities are
 *            multiplied by keep_scale = 1/(1-p); the normaliser (lse) is that of the full row.
 *            NULL = no dropout.  Only with ea_geom.causal != 0 (the only module that applies it).
 * Backward consumes the forward's out, lse and dout and produces dq, dk, dv (WRITTEN, not
 * accumulated).  It runs as one launch per group of (window, query block) pairs that share no key:
 * with overlapping windows (ext > 0) a token is a key of several windows, and a 1-D window whose
 * rows do not fit one LDS image (e.g. window 128 at D = 128) is processed in
 * ea_window_bwd_query_blocks(g) blocks of queries (launched together when the windows do not
 * overlap, each block with its own scratch slice).  The caller then passes two fp32 scratch buffers
 * dk_acc, dv_acc of [ea_window_bwd_acc_slices(g), B,H,N,D] (initialised inside) that collect the
 * key/value gradients before they are summed and converted into dk/dv; they may be NULL when that
 * count is 0.  With more than one query block dbias_part must be ZEROED by the caller (a block
 * writes only its own rows).
 * The landmark and bias gradients come back as per-workgroup partial sums which the caller
 * reduces over the leading axes:
 *   dlk_part, dlv_part : fp32 [ea_window_bwd_parts(g), B*H, L, D]
 *   dbias_part         : fp32 [ea_window_bwd_bias_parts(g), B, H, Wq, ld]   (NULL iff bias NULL)
 *   bias_t             : fp32 [H, ld, 16*ceil(Wq/16)] transposed copy of `bias` (rows padded with
 *                        zeros).  Only read when the head's bias table does not fit next to the
 *                        window in LDS (ea_window_bwd_needs_bias_t(g) == 1); may be NULL otherwise. */
int32_t ea_window_bias_ld(const ea_geom* g);        /* padded row length of `bias`           */
int32_t ea_window_bwd_parts(const ea_geom* g);      /* leading dim of dlk_part / dlv_part    */
int32_t ea_window_bwd_bias_parts(const ea_geom* g); /* leading dim of dbias_part             */
int32_t ea_window_bwd_needs_bias_t(const ea_geom* g);
int32_t ea_window_bwd_acc_slices(const ea_geom* g); /* [B,H,N,D] slices of dk_acc / dv_acc  */
int32_t ea_window_bwd_query_blocks(const ea_geom* g); /* > 1: dbias_part must be zeroed      */
int32_t ea_window_keep_ld(const ea_geom* g);        /* row length of `keep`                  */
int ea_window_attn_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                       const float* lk, const float* lv, const float* bias, const uint8_t* mask,
                       const ea_t4* out, float* lse, const uint8_t* keep, float keep_scale,
                       void* stream);
int ea_window_attn_bwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                       const float* lk, const float* lv, const float* bias, const uint8_t* mask,
                       const ea_t4* out, const ea_t4* dout, const float* lse,
                       const ea_t4* dq, const ea_t4* dk, const ea_t4* dv,
                       float* dlk_part, float* dlv_part, float* dbias_part,
                       float* dk_acc, float* dv_acc, const float* bias_t,
                       const uint8_t* keep, float keep_scale, const float* dlse, void* stream);

/* ---- LARA: linear randomized attention (lara.py:177-251) -------------------------------------
 * C landmark samples omega_c (C = L, or 2L with antithetic / multi-sample noise), each token n:
 *   log_proj_k[c,m] = s w_c.k_m - s|k_m|^2/2 (-inf on padded keys)          (lara.py:202-208)
 *   kv_stats_c = sum_m softmax_m(log_proj_k[c,:]) v_m,  lse_k[c] = LSE_m log_proj_k   (:211,241)
 *   mis-opt:  t[c,n] = softmax_n(s qbar_c.q_n); alpha = bh_c + kappa (t - mean_c t);
 *             log_alpha = log max(alpha, 1e-8)                               (:221-232)
 *   mis-biased: log_alpha = s qbar_c.q_n (qbar := mu rows)                   (:214-220)
 *   mis-bh:   log_alpha = 0                                                  (:233-236)
 *   W[c,n] = softmax_c(log_alpha + s w_c.q_n + cst_c),  cst_c = lse_k[c] - log_proposal_c
 *   out_n = sum_c W[c,n] kv_stats_c                                          (:241-246)
 * Landmark-side tensors are fp32: omega, qbar, kv, dkv, uq [B*H, C, D]; per-landmark scalars
 * [B*H, C]; per-token scalars [B*H, N].  The sequence-wide sums come back as partial results over
 * S = ea_lara_parts(g) slices that the caller merges (log-sum-exp merge for the forward
 * statistics, plain sums for the backward ones):
 *   stats_fwd : p_ml [B*H, S, C, 4] = (max_k, sum_k, max_t, sum_t) in natural-log units of the
 *               running maxima, p_kv [B*H, S, C, D] un-normalised sum_m exp(lpk - max_k) v_m
 *   bwd_qstats: p_ml = (r_c = sum_n dZ, sum_n dalpha, u_c = sum_n t dt, 0);
 *               p_dkv = d kv_stats; p_dom = sum_n dZ q_n; p_m1 = sum_n t dt q_n; p_m2 = sum_n t q_n
 *   bwd_kstats: p_dom = sum_m dBk[c,m] k_m  (caller scales d omega by s)
 * bwd_q writes the sequence-local part of dq plus the per-token scalars bwd_qstats consumes;
 * bwd_qcorr subtracts s sum_c t[c,n] uq_c (uq_c = u_c qbar_c) from dq in place (mis-opt only);
 * bwd_k writes dk, dv given dkv, lse_k, dkk_c = dkv_c.kv_c and rsum_c = r_c. */
#define EA_MIS_OPT    0
#define EA_MIS_BIASED 1
#define EA_MIS_BH     2
typedef struct {
  int32_t B, H, N, D;
  int32_t dtype;             /* EA_BF16 | EA_F16 */
  int32_t C;                 /* landmark samples, <= 128 */
  int32_t mis;               /* EA_MIS_* */
  float   kappa;             /* alpha_coeff (lara.py:231) */
  float   scale;             /* D^-0.5 */
} ea_lara_geom;

int32_t ea_lara_parts(const ea_lara_geom* g);
int ea_lara_stats_fwd(const ea_lara_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v,
                      const uint8_t* mask, const float* omega, const float* qbar,
                      float* p_ml, float* p_kv, void* stream);
/* lseZ / tmean (both or neither; fp32 [B*H, N]): the per-token statistics of the estimator's softmax over the samples,
 * log2(sum_c alpha 2^z) and mean_c t, kept for ea_lara_bwd_q_fused (8 bytes per token-head). */
int ea_lara_out_fwd(const ea_lara_geom* g, const ea_t4* q, const float* omega, const float* qbar,
                    const float* kv, const float* lse_t, const float* bhv, const float* cst,
                    const ea_t4* out, float* lseZ, float* tmean, void* stream);
int ea_lara_bwd_q(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                  const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                  const float* cst, const ea_t4* dq, float* lseZ, float* tmean, float* rowdot,
                  float* sda, void* stream);
int ea_lara_bwd_qstats(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                       const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                       const float* cst, const float* lseZ, const float* tmean, const float* rowdot,
                       const float* sda, float* p_ml, float* p_dkv, float* p_dom, float* p_m1,
                       float* p_m2, void* stream);
int ea_lara_bwd_k(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                  const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                  const float* rsum, const ea_t4* dk, const ea_t4* dv, void* stream);
int ea_lara_bwd_kstats(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                       const float* rsum, float* p_dom, void* stream);
int ea_lara_bwd_qcorr(const ea_lara_geom* g, const ea_t4* q, const float* qbar, const float* uq,
                      const float* lse_t, const ea_t4* dq, void* stream);

/* Fused backward (round 2): the elementwise stage of the estimator is evaluated once per side and its
 * [C x N] weight matrices are transposed through LDS, so q/dout and k/v are each read ONCE
 * (ea_lara_bwd_q + ea_lara_bwd_qstats, ea_lara_bwd_k + ea_lara_bwd_kstats of round 1 read them
 * twice), and (round 3) the softmax statistics lseZ / tmean of ea_lara_out_fwd are re-used instead of re-derived.
 * C <= 64.  Partial outputs [BH, ea_lara_fused_parts(g), C, *] feed ea_lara_merge_bwd /
 * ea_slice_sum unchanged.  Replaces the autograd of lara.py:201-246.
 * ea_lara_bwd_finish: dq -= s sum_c t[c,n] (u q_bar)_c (softmax-over-sequence correction, uq may be
 * NULL) and, with pool_r > 0, the backward of the uniform pool_r x pool_r average pooling of q and k
 * over the gh x gw token grid (lara.py:43,48,145-151): dq += dpq[chunk(n)] / r^2, dk += dpk[...] / r^2. */
int32_t ea_lara_fused_parts(const ea_lara_geom* g);
int ea_lara_bwd_q_fused(const ea_lara_geom* g, const ea_t4* q, const ea_t4* dout, const float* omega,
                        const float* qbar, const float* kv, const float* lse_t, const float* bhv,
                        const float* cst, const float* lseZ, const float* tmean, const ea_t4* dq, float* p_ml,
                        float* p_dkv, float* p_dom, float* p_m1, float* p_m2, void* stream);
int ea_lara_bwd_k_fused(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const float* omega, const float* dkv, const float* lse_k, const float* dkk,
                        const float* rsum, const ea_t4* dk, const ea_t4* dv, float* p_dom, void* stream);
int ea_lara_bwd_finish(const ea_lara_geom* g, const ea_t4* q, const float* qbar, const float* uq,
                       const float* lse_t, const float* dpq, const float* dpk, int32_t pool_r,
                       int32_t gh, int32_t gw, const ea_t4* dq, const ea_t4* dk, void* stream);

/* ---- softmax baseline (abstract_attention.py:120-133) ----------------------------------------
 * out = dropout(softmax(s Q K^T, -inf on padded keys)) V, streamed (no [N,N] score matrix).
 * lse: fp32 [B*H, N] saved for backward; delta: fp32 [B*H, N] scratch (dO.O) written by the dQ
 * pass and read by the dK/dV pass of ea_softmax_attn_bwd.
 * keep: attention dropout (`attn = self.attn_drop(attn)`, :131): uint8 [B,H,N, 64*ceil(N/64)], entry
 * (n, j) non-zero = probability of key j for query n is kept and multiplied by keep_scale = 1/(1-p);
 * NULL = no dropout.  q, k, v may be views of different tensors (any strides, rows contiguous). */
int ea_softmax_attn_fwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                        const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const ea_t4* out, float* lse, const uint8_t* keep, float keep_scale,
                        int32_t key_norm_bias, void* stream);
int ea_softmax_attn_bwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                        const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                        const ea_t4* out, const ea_t4* dout, const float* lse, float* delta,
                        const ea_t4* dq, const ea_t4* dk, const ea_t4* dv,
                        const uint8_t* keep, float keep_scale, int32_t key_norm_bias, void* stream);
/* Randomized attention (randomized_attention.py:21-52) is two passes of the kernels above:
 *   mu = q + softmax(s q k^T) k (num_samples = -1) or q + k[index] with one index per query drawn
 *   from softmax(s q k^T) (ea_softmax_sample: Gumbel-max with counter-based noise from the 64-bit
 *   device scalar `seed`; index: int64 [B*H, N]); then out = softmax_j(s w.k_j - s |k_j|^2 / 2) v_j
 *   with w = mu (+ noise): key_norm_bias = 1 adds the per-key term and its gradient to dk
 *   (not combined with `keep`). */
int ea_softmax_sample(int32_t B, int32_t H, int32_t N, int32_t D, int32_t dtype, float scale,
                      const ea_t4* q, const ea_t4* k, const uint64_t* seed, int64_t* index, void* stream);

/* ---- Performer / FAVOR+ baseline (kernelized_attention.py:20-56,116-121,326-346) ---------------
 * phi(x)[j] = M^-1/2 exp(d^-1/4 W_j.x - d^-1/2 |x|^2/2 - stab) + 1e-4 with W fp32 [H, M, D] (fresh
 * Gaussian features per training call, `eval_proj` otherwise); stab = max_j for queries, the max
 * over all keys and features of one (b,h) for keys (detached); phi(k) is zeroed at padded keys.
 *   kv[j] = sum_n phi(k_n)[j] v_n,  ksum[j] = sum_n phi(k_n)[j]
 *   out_n = phi(q_n).kv / max(phi(q_n).ksum, 1e-2)
 * Sequence-wide sums come back as partials over S = ea_performer_parts(g) slices (p_ml[...,0] holds
 * the scalar per feature: max of d^-1/4 W_j.k_n for kmax, ksum for kv, d(ksum) for bwd_qstats). */
typedef struct {
  int32_t B, H, N, D;
  int32_t dtype;
  int32_t M;                 /* number of random features (approx_attn_dim), <= 128 */
} ea_perf_geom;
int32_t ea_performer_parts(const ea_perf_geom* g);
int ea_performer_kmax(const ea_perf_geom* g, const ea_t4* k, const float* W, float* p_ml, void* stream);
int ea_performer_kv(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                    const float* W, const float* stab, float* p_ml, float* p_kv, void* stream);
int ea_performer_out(const ea_perf_geom* g, const ea_t4* q, const float* W, const float* kv,
                     const float* ksum, const ea_t4* out, void* stream);
int ea_performer_bwd_q(const ea_perf_geom* g, const ea_t4* q, const ea_t4* out, const ea_t4* dout,
                       const float* W, const float* kv, const float* ksum, const ea_t4* dq,
                       float* stabq, float* invden, float* dden, void* stream);
int ea_performer_bwd_qstats(const ea_perf_geom* g, const ea_t4* q, const ea_t4* dout, const float* W,
                            const float* stabq, const float* invden, const float* dden,
                            float* p_ml, float* p_dkv, void* stream);
int ea_performer_bwd_k(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* W, const float* stab, const float* dkv, const float* dksum,
                       const ea_t4* dk, const ea_t4* dv, void* stream);

/* ---- LARA 'adaptive-1d' proposals with the generator INSIDE the segment kernels (ea_lara_seglin.hip; round 4) ----
 * q_bar_l = mean over segment l of LayerNorm(G_q q_n + g_q), k_bar_l likewise (lara.py:56-63,84-127), straight from the
 * stored q / k rows: generator Linear (a [64 x 64] MFMA product per 16 tokens), LayerNorm and segment mean in one pass; the
 * qkv projection stays 3C wide (the folded form below makes all three GEMMs of the layer 5C wide).  g: B, H, N, D = 64,
 * dtype, L (segments as in ea_lara_segment_*); G [64, 64] (out, in), g_b, ln_w, ln_b [64]: fp32.
 *   fwd: qbar, kbar [B*H, L, 64] fp32.
 *   bwd: ACCUMULATES G^T d z into dq / dk (the attention core's gradient rows, I/O dtype); with S = B*H*
 *        ea_lara_seglin_groups(g): part [S, 2, 4, 64] = partial sums of (d ln_w, d ln_b, d g_b, 0) per side, dG_part
 *        [S, 2, 64, 64] = partial sums of dG (side 0: q generator) -- the caller adds them over S; stats: scratch of
 *        B*H*2*N*4 floats (16-byte aligned) in which the dq / dk pass leaves every token's LayerNorm statistics for the dG
 *        pass. */
int32_t ea_lara_seglin_groups(const ea_geom* g);
int ea_lara_seglin_fwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                       const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                       float* qbar, float* kbar, void* stream);
int ea_lara_seglin_bwd(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                       const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                       const float* d_qbar, const float* d_kbar, const ea_t4* dq, const ea_t4* dk, float* part, float* dG_part,
                       float* stats, void* stream);
/* round 6: the same backward with the estimator's LAST correction of dq riding on the dq / dk pass -- what ea_lara_bwd_finish
 * applies in a pass of its own (lara.py:223 differentiated: t = softmax over the sequence of s qbar_c . q_n,
 *     dq_n -= s sum_c t[c, n] (u qbar)_c ),
 * added to the rows this pass rewrites anyway (the q rows are its MFMA operand already): one read-modify-write of dq less per
 * step.  fin_qbar, fin_uq [B*H, C, 64], fin_lse_t [B*H, C] fp32 as ea_lara_bwd_finish takes them (qbar, uq, lse_t); C <= 64
 * samples (EA_E_UNSUPPORTED beyond); scale = d^-1/2.  The caller must NOT run ea_lara_bwd_finish for the same step. */
int ea_lara_seglin_bwd_fin(const ea_geom* g, const ea_t4* q, const ea_t4* k, const float* Gq, const float* gq_b, const float* Gk,
                           const float* gk_b, const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                           const float* d_qbar, const float* d_kbar, const ea_t4* dq, const ea_t4* dk, float* part,
                           float* dG_part, float* stats, const float* fin_qbar, const float* fin_uq, const float* fin_lse_t,
                           int32_t C, float scale, void* stream);

/* ---- LARA 'adaptive-1d' proposals: the generators' per-token Linear folded into the qkv projection (ea_fold.hip) ----
 * q_bar_gen / k_bar_gen start with Linear(d, d) on every token's q / k row (lara.py:56-63,100-103).  The module computes
 * Linear_gen(q) as two more groups of output columns of the qkv GEMM (W' = G W_head per head); these two entry points
 * build the extended weight and take its gradient apart in one launch each way (round 4; ~30 framework kernels before).
 *   fwd: w_ext [5C, C] (EA dtype) = [ W ; Gq W_q,head ; Gk W_k,head ], b_ext [5C] (EA dtype, may be NULL) = [ b ; 0 ],
 *        bias_q / bias_k [heads, d] fp32 = G b_head + g_b: the bias of a folded row (consumed by ea_lara_segment_*).
 *        W [3C, C], b [3C] (may be NULL), Gq / Gk [d, d], gq_b / gk_b [d]: fp32 parameters.
 *   bwd: from dW_ext [5C, ldw] / db_ext [5C] (the weight / bias gradient of the extended projection, e.g. ea_wgrad's sums)
 *        and dbias_q / dbias_k [heads, d]: dW [3C, C], db [3C] (NULL when b is), dG [2, d, d] = (dGq, dGk), dgq_b, dgk_b [d].
 *        dG_part: scratch of (2 * ea_lara_fold_parts(heads) + 2) * d * d floats (per-head partials of dG; the entry point
 *        adds them up with ea_slice_sum).  d = 64 (EA_E_UNSUPPORTED otherwise). */
int ea_lara_fold_fwd(int32_t dtype, int32_t C, int32_t heads, const float* W, const float* b, const float* Gq, const float* gq_b,
                     const float* Gk, const float* gk_b, void* w_ext, void* b_ext, float* bias_q, float* bias_k, void* stream);
int32_t ea_lara_fold_parts(int32_t heads);
int ea_lara_fold_bwd(int32_t C, int32_t heads, const float* W, const float* b, const float* Gq, const float* Gk,
                     const float* dW_ext, int64_t ldw, const float* db_ext, const float* dbias_q, const float* dbias_k,
                     float* dW, float* db, float* dG, float* dG_part, float* dgq_b, float* dgk_b, void* stream);

/* ---- Performer in EXACT fp32 arithmetic (ea_performer_f32.hip; round 4) --------------------
 * The reference computes its linear attention in full precision whatever the AMP state (kernelized_attention.py:116-121
 * `autocast(enabled=False)`, :343-345 `.float()`), and a module called outside autocast computes everything in fp32
 * (abstract_attention.py:120-133).  These entry points are that arithmetic: q, k, v, dout of type g->dtype = EA_BF16 /
 * EA_F16 / EA_F32 (16-bit values are exact in fp32), every product on v_mfma_f32_16x16x4_f32 with fp32 operands, fp32
 * features phi, outputs in g->dtype.  D = 64, M <= 96 (multiple of 16); W [H, M, 64] fp32.
 *   S = ea_performer_f32_parts(g) sequence slices per (b,h);
 *   kmax : p_max [BH,S]  = slice maxima of d^-1/4 W_j.k_n (the key stabiliser is their maximum; padded keys included,
 *          as in the reference, which masks the features afterwards);
 *   kv   : p_kv [BH,S,M,64], p_ksum [BH,S,M] = partial sum_n phi(k_n)^T v_n, sum_n phi(k_n) (padded keys: phi = 0);
 *          the caller adds the slices (ea_slice_sum) -> kv [BH,M,64], ksum [BH,M];
 *   out  : out_n = phi(q_n) kv / max(phi(q_n).ksum, 1e-2);
 *   bwd_q: dq, and the slice partials of d kv, d ksum (same shapes as kv's);  bwd_k: dk, dv from the summed d kv, d ksum.
 * Replaces favorp_projection + linear_attention and their autograd (kernelized_attention.py:20-56,116-121). */
int32_t ea_performer_f32_parts(const ea_perf_geom* g);
int ea_performer_f32_kmax(const ea_perf_geom* g, const ea_t4* k, const float* W, float* p_max, void* stream);
int ea_performer_f32_kv(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                        const float* p_max, float* p_kv, float* p_ksum, void* stream);
int ea_performer_f32_out(const ea_perf_geom* g, const ea_t4* q, const float* W, const float* kv, const float* ksum,
                         const ea_t4* out, void* stream);
int ea_performer_f32_bwd_q(const ea_perf_geom* g, const ea_t4* q, const ea_t4* dout, const float* W, const float* kv,
                           const float* ksum, const ea_t4* dq, float* p_dkv, float* p_dksum, void* stream);
int ea_performer_f32_bwd_k(const ea_perf_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                           const float* p_max, const float* dkv, const float* dksum, const ea_t4* dk, const ea_t4* dv,
                           void* stream);

/* ---- LARA landmark pipeline, fused (lara.py:145-198,214-238) -----------------------------------
 * One workgroup per (b,h), all matrices in LDS, exact fp32:
 *   q_bar = LN(pq Wq^T + bq), k0 = LN(pk Wk^T + bk)         (has_mlp; else q_bar = pq, k0 = pk)
 *   k_bar = softmax(s k0 k0^T) k0                            (mixed; else k_bar = k0)
 *   mu = q_bar + k_bar;  omega_c = mu[c mod L] + eps_c       (dup 0: C = L; 1 antithetic: C = 2L,
 *                                                             eps [BH,L,D], second half negated;
 *                                                             2 multi-sample: C = 2L, eps [BH,2L,D])
 *   M[c,l] = s omega_c.mu_l - s|mu_l|^2/2
 *   mis-opt   : lp_c = M[c, c mod L]; bh_c = exp(lp_c - LSE over the C (repeated) columns);
 *               qbar_rows_c = q_bar[c mod L]
 *   mis-biased: lp_c = LSE_l M[c,l]; qbar_rows_c = mu[c mod L];   mis-bh: lp_c = LSE_l M[c,l]
 * Outputs omega, qbar_rows [BH,C,D], bhv, lp [BH,C] feed ea_lara_*.  The backward takes their
 * gradients and returns d pq, d pk [BH,L,D] plus per-(b,h) partial parameter gradients
 * dW_part [BH,2,D,D] (q then k; [out][in]) and dvec_part [BH,2,3,D] (Linear bias, LN weight, LN
 * bias) which the caller sums over BH.  L, C <= 64.
 * `saved`: workspace of ea_lara_landmarks_saved_floats(g) floats in which the forward keeps its
 * intermediates (normalised rows, LayerNorm 1/std, mixing matrix, mu) for the backward.  NULL in a
 * forward that will not be differentiated; the backward of a parametrised (has_mlp) or mixed pipeline
 * REQUIRES it (EA_E_BADARG otherwise -- round 3 retired the kernels that recomputed the forward). */
typedef struct {
  int32_t BH, L, C, D;
  int32_t has_mlp, mixed, mis, dup;
  float   scale;
  int32_t eva;               /* 1: EVA's mu pipeline instead (eva.py:178-190, adaptive_proj='default'):
                                qbar_rows = rf_k_bar = LN(pk Wk^T + bk), omega = (rf_q_bar + rf_k_bar)/2 + eps;
                                bhv / lp unused; the backward takes d_omega and d_qbar_rows = d rf_k_bar */
} ea_lmk_geom;
int ea_lara_landmarks_fwd(const ea_lmk_geom* g, const float* pq, const float* pk,
                          const float* Wq, const float* bq, const float* gq, const float* cq,
                          const float* Wk, const float* bk, const float* gk, const float* ck,
                          const float* noise, float* omega, float* qbar_rows, float* bhv, float* lp,
                          float* saved, void* stream);
int64_t ea_lara_landmarks_saved_floats(const ea_lmk_geom* g);
int ea_lara_landmarks_bwd(const ea_lmk_geom* g, const float* pq, const float* pk,
                          const float* Wq, const float* bq, const float* gq, const float* cq,
                          const float* Wk, const float* bk, const float* gk, const float* ck,
                          const float* noise, const float* d_omega, const float* d_qbar_rows,
                          const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                          float* dW_part, float* dvec_part, const float* saved, void* stream);
/* '-vmixed' proposals (lara.py:171-172): colbias [BH, L] = log(|v_bar_l'| + 1e-4) is added to column l' of the mixing
 * logits s k0 k0^T before their softmax; the backward also returns d_colbias [BH, L] (column sums of the logit
 * gradients).  g->mixed must be set and `saved` (the forward intermediates) is required in the backward; geometries
 * outside the second-generation kernels (L, C <= 64, D in {32, 64}) return EA_E_UNSUPPORTED. */
int ea_lara_landmarks_fwd_cb(const ea_lmk_geom* g, const float* pq, const float* pk,
                             const float* Wq, const float* bq, const float* gq, const float* cq,
                             const float* Wk, const float* bk, const float* gk, const float* ck,
                             const float* noise, const float* colbias, float* omega, float* qbar_rows, float* bhv,
                             float* lp, float* saved, void* stream);
int ea_lara_landmarks_bwd_cb(const ea_lmk_geom* g, const float* pq, const float* pk,
                             const float* Wq, const float* bq, const float* gq, const float* cq,
                             const float* Wk, const float* bk, const float* gk, const float* ck,
                             const float* noise, const float* colbias, const float* d_omega, const float* d_qbar_rows,
                             const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                             float* dW_part, float* dvec_part, float* d_colbias, const float* saved, void* stream);

/* ---- LARA: merging the sequence slices of the token-row passes (tiny, one workgroup per (b,h)) ----
 * merge_fwd: (p_ml, p_kv of ea_lara_stats_fwd over S slices, lp) -> kv_stats [BH,C,D], lse_k,
 *   lse_t (has_t: mis-opt) and cst = lse_k - lp [BH,C].
 * merge_bwd: (p_ml, p_dkv, p_dom, p_m1, p_m2 of ea_lara_bwd_qstats) -> r, d(bh), d(lp) = -r,
 *   dkk = dkv.kv [BH,C]; dkv, domq = sum_n dZ q_n [BH,C,D]; with has_t also
 *   dqbar = s (M1 - u M2) and uq = u qbar; without, dqbar (if non-NULL) = s domq (mis-biased). */
int ea_lara_merge_fwd(int32_t BH, int32_t S, int32_t C, int32_t D, int32_t has_t,
                      const float* p_ml, const float* p_kv, const float* lp,
                      float* kv, float* lse_k, float* lse_t, float* cst, void* stream);
int ea_lara_merge_bwd(int32_t BH, int32_t S, int32_t C, int32_t D, int32_t has_t, float scale,
                      const float* p_ml, const float* p_dkv, const float* p_dom, const float* p_m1,
                      const float* p_m2, const float* kv, const float* qbar,
                      float* r, float* dbh, float* dlp, float* dkk, float* dkv, float* domq,
                      float* dqbar, float* uq, void* stream);

/* ---- projection bias gradient: db[c] = sum_t dY[t][c] ----
 * Replaces the bias-gradient reduction autograd runs for the qkv / output nn.Linear of every
 * attention module (abstract_attention.py:34-36 `self.qkv`, `self.proj`; eva.py:76-77, lara.py:88-90):
 * dY is the contiguous [rows = B*N, cols] cotangent of the projection output in the I/O dtype,
 * db is fp32 [cols].  Deterministic two-stage sum; `part` is caller workspace of
 * ea_bias_grad_parts(rows, cols) * cols floats.  cols % 8 == 0, cols <= 16384. */
int ea_bias_grad_parts(int32_t rows, int32_t cols);
int ea_bias_grad(int32_t dtype, int32_t rows, int32_t cols, const void* dy, float* part, float* db,
                 void* stream);

/* ---- small fp32 reductions around the cores (replace chains of tiny torch kernels) ----
 * ea_colsum_f32: out[c] = sum_r x[r][c], x fp32 [rows, cols] contiguous, fixed summation order.
 *   Used for the per-(b,h) partials of the landmark-MLP parameter gradients
 *   (ea_lara_landmarks_bwd dW_part / dvec_part: autograd's accumulation over the batch for
 *   q_bar_gen / k_bar_gen, lara.py:45-54, eva.py:93-103).
 * ea_slice_sum: out[bh][j] = scale * (a[bh][j] + sum_s parts[bh][s][j]), j < n (n % 4 == 0), a may
 *   be NULL.  Used for d(omega) = s (d_omega_q + sum over sequence slices of ea_lara_bwd_kstats). */
int ea_colsum_f32(int32_t rows, int32_t cols, const float* x, float* out, void* stream);
/* Two column sums over the same rows in one launch (dW_part and dvec_part of the landmark backward). */
int ea_colsum2_f32(int32_t rows, int32_t cols1, const float* x1, float* out1, int32_t cols2, const float* x2, float* out2,
                   void* stream);
/* Gradient of a table gather (the relative-position bias table read through `relative_position_index`,
 * local_attention.py:70-79): out[row][c] = sum_k g[inv[row][k]][c], inv [rows, K] int32 = the gather positions that read
 * table row `row` (-1 = unused slot), g [n, cols] fp32.  Fixed order (deterministic). */
int ea_gather_sum(int32_t rows, int32_t K, int32_t cols, const float* g, const int32_t* inv, float* out, void* stream);
/* Round 6: the dense per-head bias of the window kernels straight out of its table, in ONE launch each way (replaces the
 * index_select / permute / multiply / pad / copy chain around `relative_position_bias_table[relative_position_index]`,
 * local_attention.py:70-79, and around T5RelativePositionBias.forward, eva.py:53-65, causal_eva.py:206-300):
 *   fwd: out[hd][i][j] = scale * table[idx[i*Wk + j]][hd] for j < Wk, 0 for Wk <= j < ld      (table [rows, th] fp32, idx int32
 *        [Wq*Wk], out [h, Wq, ld] fp32 -- ld = ea_window_bias_ld(geom); `scale` carries log2(e), the kernels' logit unit)
 *   bwd: dtable[row][hd] = scale * sum_k g[hd][p / Wk][p % Wk], p = inv[row][k] >= 0           (g [h, Wq, ld] fp32 = the bias
 *        gradient the window backward returns; inv [rows, K] int32 as for ea_gather_sum).  Fixed order (deterministic).
 *   th (fwd, ABI 13) = heads of the table: h, or 1 = one column broadcast over the h heads (causal EVA's single-head T5
 *        table); bwd always fills dtable [rows, h] -- the gradient of a one-column table is its sum over the heads. */
int ea_table_bias_fwd(int32_t h, int32_t th, int32_t Wq, int32_t Wk, int32_t ld, float scale, const float* table, const int32_t* idx,
                      float* out, void* stream);
int ea_table_bias_bwd(int32_t rows, int32_t K, int32_t h, int32_t Wq, int32_t Wk, int32_t ld, float scale, const float* g,
                      const int32_t* inv, float* dtable, void* stream);
/* Round 6 (ABI 12): the autocast casts of a layer's parameters in ONE launch -- dst[k][i] = (dtype) src[k][i], i < n[k], for
 * K <= 8 fp32 tensors (round to nearest even, exactly torch's `.to(dtype)`).  The 320 / 512 / 1024-wide layers run their two
 * projections (abstract_attention.py:72-78,86-87) as library GEMMs on 16-bit operands; their weights and biases were four
 * separate cast launches per step. */
int ea_multi_cast(int32_t dtype, int32_t K, const float* const* src, const int64_t* n, void* const* dst, void* stream);
int ea_slice_sum(int32_t BH, int32_t S, int32_t n, float scale, const float* a, const float* parts,
                 float* out, void* stream);
/* Measurement aid, not on the path: dst[0 .. bytes) = src[0 .. bytes) by a plain 16-byte-per-lane device copy kernel
 * (src, dst 16-byte aligned, bytes a multiple of 16).  bench.py times it as the achievable-HBM-bandwidth yardstick next
 * to the nominal 8 TB/s ("babel-stream-style copy kernel on the same GPU"). */
int ea_stream_copy(const void* src, void* dst, int64_t bytes, void* stream);

/* ---- LARA 1-D landmark proposals (LinearRA._proposal_gen_1d, lara.py:84-127) ----
 * Segment means over the sequence, q_bar_l = mean_{n in segment l} row_n, with the reference's
 * split of N tokens into L = g->L segments (N % L == 0: equal; otherwise (segs+1)L - N segments of
 * segs = N / L tokens followed by segments of segs + 1).  Geometry: ea_geom with B, H, N, D, dtype,
 * L (other fields ignored); N > L.
 *   'adaptive-1d' (gq != NULL): row_n = LayerNorm(x_n + bias), x = the rows q2 / k2 [B,H,N,D]
 *   (element type) holding Linear(q) WITHOUT its bias -- the host folds that Linear into the qkv
 *   projection --, bias = bias_q/k [H,D] for ordinary tokens and mbias_q/k [D] for tokens under
 *   `mask` (the reference zeroes q, k of padded tokens BEFORE the Linear, so their row is
 *   LayerNorm(Linear bias)); gq, cq / gk, ck = LayerNorm weight, bias [D].
 *   Plain means (gq == NULL): row_n = x_n (+ bias when given).
 * Forward: qbar, kbar fp32 [B*H, L, D].  Backward: d_qbar, d_kbar -> dq2, dk2 (gradient of the
 * rows x, WRITTEN) and per-segment partials part [B*H*L, 2 (q,k), 4, D] = (d LN weight, d LN bias,
 * d bias (sum over unmasked rows), d mbias (sum over masked rows)) which the caller sums. */
int ea_lara_segment_fwd(const ea_geom* g, const ea_t4* q2, const ea_t4* k2, const uint8_t* mask,
                        const float* bias_q, const float* bias_k, const float* mbias_q, const float* mbias_k,
                        const float* gq, const float* cq, const float* gk, const float* ck,
                        float* qbar, float* kbar, void* stream);
int ea_lara_segment_bwd(const ea_geom* g, const ea_t4* q2, const ea_t4* k2, const uint8_t* mask,
                        const float* bias_q, const float* bias_k, const float* mbias_q, const float* mbias_k,
                        const float* gq, const float* cq, const float* gk, const float* ck,
                        const float* d_qbar, const float* d_kbar, const ea_t4* dq2, const ea_t4* dk2,
                        float* part, void* stream);

/* ---- mu networks on the chunk means (eva.py:78-98,178-183; causal_eva.py:376-392,706-707) ------
 * y_s = [LayerNorm_s](x_s W_s^T + b_s) for sides = 1 (key side only: adaptive_proj 'none') or 2
 * (query side 0, key side 1), rows [R, D] fp32 with D in {32, 64, 128}, exact fp32 arithmetic.
 * layer_norm: 1 = Linear + LayerNorm (eps 1e-5), 0 = Linear only ('no-ln').
 *   fwd: zhat [sides,R,D] and rstd [sides,R] receive the normalised rows and 1/std for the backward
 *        (both NULL when no backward follows or layer_norm == 0).
 *   bwd: dx_s = d/dx; feed [R, planes, sides, D] with planes = 3 (dz, dy o zhat, dy) or 1 (dz = dy)
 *        whose column sums over R are (d bias, d gamma, d beta); dW_part
 *        [ea_rows_mlp_parts(R, D), sides, D, D] per-workgroup partials of d W. */
int32_t ea_rows_mlp_parts(int32_t R, int32_t D);
int ea_rows_mlp_fwd(int32_t R, int32_t D, int32_t sides, int32_t layer_norm,
                    const float* x0, const float* x1, const float* W0, const float* W1,
                    const float* b0, const float* b1, const float* g0, const float* g1,
                    const float* c0, const float* c1, float* y0, float* y1,
                    float* zhat, float* rstd, void* stream);
int ea_rows_mlp_bwd(int32_t R, int32_t D, int32_t sides, int32_t layer_norm,
                    const float* dy0, const float* dy1, const float* x0, const float* x1,
                    const float* W0, const float* W1, const float* g0, const float* g1,
                    const float* zhat, const float* rstd, float* dx0, float* dx1,
                    float* feed, float* dW_part, void* stream);

/* Weight and bias gradient of a projection in one pass over the activations (ea_wgrad.hip): the autograd
 * of `qkv = self.qkv(x)` / `x = self.proj(x)` (abstract_attention.py:72-78,86-87) with respect to the
 * Linear's parameters.  dy [rows, out_features], x [rows, in_features]: contiguous, EA_BF16 / EA_F16;
 * out_features and in_features multiples of 64 (EA_E_UNSUPPORTED otherwise).
 *   Slice s leaves its partial sums at dw_part + s * part_ld ([out_features, in_features] fp32) and, when db_part is not
 *   NULL, db_part + s * part_ld ([out_features] fp32); S = ea_wgrad_parts(rows, out_features, in_features), part_ld >=
 *   out_features * in_features and a multiple of 4.  With db_part = dw_part + out_features * in_features and part_ld =
 *   out_features * (in_features + 1) one ea_part_sum adds both up.
 * ea_part_sum: out[j] = sum_s parts[s * ld + j], j < n (n, ld multiples of 4), slices added in a fixed order
 *   (deterministic; no atomics). */
int32_t ea_wgrad_parts(int32_t rows, int32_t out_features, int32_t in_features);
int ea_wgrad(int32_t dtype, int32_t rows, int32_t out_features, int32_t in_features, const void* dy, const void* x,
             float* dw_part, float* db_part, int64_t part_ld, void* stream);
int ea_part_sum(int32_t S, int32_t n, int64_t ld, const float* parts, float* out, void* stream);
/* The weight (+ bias) gradients of TWO projections over the same token rows in one launch -- a layer's qkv and output
 * projections (abstract_attention.py:72-78,86-87 differentiated): dW1 = dY1^T X1, dW2 = dY2^T X2, rows shared.  With both
 * products' tiles in one grid a token slice is longer (S = ea_wgrad_pair_parts(...) slices for BOTH, fewer than either
 * product alone takes): half the partial-sum bytes and one launch less.  Arguments per product as for ea_wgrad; both need
 * the same tile edges (EA_E_UNSUPPORTED from ea_wgrad_pair_parts otherwise: use two ea_wgrad calls). */
int32_t ea_wgrad_pair_parts(int32_t rows, int32_t out1, int32_t in1, int32_t out2, int32_t in2);
int ea_wgrad_pair(int32_t dtype, int32_t rows, int32_t out1, int32_t in1, const void* dy1, const void* x1, float* dw_part1,
                  float* db_part1, int64_t part_ld1, int32_t out2, int32_t in2, const void* dy2, const void* x2,
                  float* dw_part2, float* db_part2, int64_t part_ld2, void* stream);
/* K <= 6 such reductions in one launch: out[k][j] = sum_s parts[k][s * ld[k] + j], j < n[k] (same order of additions as
 * ea_part_sum).  The terminal sums of a layer's backward -- both projections' slice partials and the per-(b,h) partials of
 * the landmark parameters (ea_lara_layer_bwd with dparams == NULL) -- share one launch this way. */
int ea_multi_sum(int32_t K, const float* const* parts, const int32_t* S, const int32_t* n, const int64_t* ld, float* const* out,
                 void* stream);

/* LARA sampling + proposal densities for sample counts beyond the fused landmark kernels (C > 64: antithetic /
 * multi-sample draws at 49 landmarks; lara.py:187-238).  qbar, mu = q_bar + k_bar: fp32 [BH,L,D]; noise: [BH,L,D] (mode 1,
 * antithetic: omega = [mu + eps; mu - eps]) or [BH,C,D] (mode 2, multi-sample: omega = [mu; mu] + eps), NULL for mode 0
 * (C = L).  With prm[c][n] = s <omega_c, mu_n> - s |mu_n|^2 / 2 over the L landmarks:
 *   EA_MIS_OPT   : lp_c = prm[c][c mod L], bhv_c = exp(lp_c - logsumexp_n prm[c][n]) L / C, qbar_rows = q_bar repeated;
 *   EA_MIS_BIASED: lp_c = logsumexp_n prm[c][n], qbar_rows = mu repeated;      EA_MIS_BH: lp_c likewise, no qbar_rows.
 * The backward takes the gradients of omega / qbar_rows / bhv / lp (NULL = zero, d_omega required) and returns d_qbar and
 * d_mu [BH,L,D].  fp32, one workgroup per (b,h); EA_E_UNSUPPORTED when the rows do not fit 150 KB of LDS. */
int ea_lara_sample_fwd(int32_t BH, int32_t L, int32_t C, int32_t D, int32_t mis, int32_t mode, float scale,
                       const float* qbar, const float* mu, const float* noise, float* omega, float* qbar_rows, float* bhv,
                       float* lp, void* stream);
int ea_lara_sample_bwd(int32_t BH, int32_t L, int32_t C, int32_t D, int32_t mis, int32_t mode, float scale,
                       const float* qbar, const float* mu, const float* noise, const float* d_omega, const float* d_qbar_rows,
                       const float* d_bhv, const float* d_lp, float* d_qbar, float* d_mu, void* stream);

/* nn.AdaptiveAvgPool2d of one of q / k / v over a gh x gw token grid that `side` does not divide (lara.py:43,48,145-151:
 * bin o of an axis of n cells = [floor(o n / side), ceil((o + 1) n / side)), neighbouring bins overlap).  x: [B,H,N,D]
 * view in the EA dtype; mean: fp32 [B,H,side*side,D].  The backward ACCUMULATES into dx (I/O dtype, fp32 math). */
int ea_adaptive_pool2d_fwd(int32_t dtype, int32_t B, int32_t H, int32_t gh, int32_t gw, int32_t side, int32_t D,
                           const ea_t4* x, float* mean, void* stream);
int ea_adaptive_pool2d_bwd(int32_t dtype, int32_t B, int32_t H, int32_t gh, int32_t gw, int32_t side, int32_t D,
                           const float* dmean, const ea_t4* dx, void* stream);

/* The projections themselves as streaming kernels (ea_linear.hip): `qkv = self.qkv(x)`, `x = self.proj(x)`
 * (abstract_attention.py:72-78,86-87) and, with the transposed weight, their input gradients.
 *   y[rows, out] = a[rows, in] w[out, in]^T (+ bias[out])
 * a: EA dtype, or fp32 when a_f32 != 0 (the autocast cast is folded into the load; a_cast, if not NULL, receives the
 * rounded [rows, in] copy for the weight-gradient product); w: [out, in] contiguous in the EA dtype; bias fp32 or
 * NULL (rounded to the EA dtype before the add, as F.linear under autocast does); y: EA dtype, or fp32 when
 * y_f32 != 0; lda / ldy: row strides in elements.  ea_linear_supported(in, out) != 0 for the built geometries
 * (in one of 64..768 in steps the models use, out a multiple of 64); EA_E_UNSUPPORTED otherwise. */
int32_t ea_linear_supported(int32_t in_features, int32_t out_features);
int ea_linear(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* a, int32_t a_f32,
              int64_t lda, const void* w, const float* bias, void* y, int32_t y_f32, int64_t ldy, void* a_cast,
              void* stream);
/* The same product straight from the fp32 MASTER weight (nn.Linear keeps fp32 parameters under autocast,
 * abstract_attention.py:34-36): the weight is rounded to the EA dtype while it is staged, so a training step needs no
 * per-step cast kernels.  w_transposed == 0: w is [out, in]; != 0: w is [in, out] and the product is a w (the input
 * gradient dX = dY W of a layer whose weight is W [out_layer, in_layer]: in = out_layer, out = in_layer), without a
 * transposed copy. */
int ea_linear_w32(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* a, int32_t a_f32,
                  int64_t lda, const float* w, int32_t w_transposed, const float* bias, void* y, int32_t y_f32, int64_t ldy,
                  void* a_cast, void* stream);
/* `qkv = self.qkv(x)` of a 192-wide, three-head model (abstract_attention.py:72-78) TOGETHER WITH the pooled q / k rows the
 * 2-D landmark generators start from (LinearRA: adaptive average pooling, lara.py:43,48,145-151; EVA: the chunk means of
 * eva.py:178-181 with 2-D chunks and no window extension): a [B*gh*gw, 192] tokens of B images, row-major gh x gw grids;
 * pooled_q / pooled_k: fp32 [B*3, (gh/r)*(gw/r), 64], the means of the ROUNDED q / k rows over the r x r cells -- exactly what
 * ea_eva_chunk_mean_fwd computes from the stored rows, without re-reading them (the kernel walks the tokens cell by cell and
 * sums the cell's rows while they are still in registers).  r = 2 or 4, in = 192, out = 576
 * (ea_linear_pool_supported != 0), EA_E_UNSUPPORTED otherwise; the other arguments as for ea_linear_w32.
 * w_cast (ABI 9): NULL, or [out_features, in_features] in the I/O type: the rounded weight, written by the same launch for
 * the backward's input-gradient GEMM (a cast launch of its own costs more than the 220 KB it moves). */
int32_t ea_linear_pool_supported(int32_t in_features, int32_t out_features, int32_t B, int32_t gh, int32_t gw, int32_t r);
int ea_linear_w32_pool(int32_t dtype, int32_t B, int32_t gh, int32_t gw, int32_t r, int32_t in_features, int32_t out_features,
                       const void* a, int32_t a_f32, int64_t lda, const float* w, const float* bias, void* y, int64_t ldy,
                       void* a_cast, float* pooled_q, float* pooled_k, void* w_cast, void* stream);

/* Round 6 (ABI 14): the 16-bit copies of a 192-wide layer's two weights in ONE launch, and the qkv projection fed by them.
 *   ea_linear_w192_prepare: wq fp32 [576, 192], wp fp32 [192, 192] (or NULL) -> w16q [576, 192] (what ea_linear_dgrad* take),
 *       wq_sw (576 * 192 elements: the same values in the order the register-resident projection kernel stages them),
 *       w16p [192, 192] and w16pT = its transpose (the output projection and its input gradient through ea_linear).
 *       Round to nearest even, as every cast of the library.
 *   ea_linear_wsw: y[rows, 576] = a[rows, 192] W^T + bias with W given as wq_sw; r = 0: plain rows; r = 2 | 4: tokens of
 *       B images of gh x gw, pooled q / k rows written as by ea_linear_w32_pool.
 * With the fp32 master weight every CU pulls 442 KB through its L2 port before its first tile -- 13 us at ANY row count
 * (workgroup timelines) -- and the two 192 x 192 projections another 2 x 147 KB per workgroup; the prepared
 * copies halve all of it for one ~4 us launch per step. */
int ea_linear_w192_prepare(int32_t dtype, const float* wq, const float* wp, void* w16q, void* wq_sw, void* w16p, void* w16pT,
                           void* stream);
int ea_linear_wsw(int32_t dtype, int32_t rows, int32_t B, int32_t gh, int32_t gw, int32_t r, const void* a, int32_t a_f32, int64_t lda,
                  const void* wq_sw, const float* bias, void* y, int64_t ldy, void* a_cast, float* pooled_q, float* pooled_k,
                  void* stream);

/* The INPUT gradient of that qkv projection (abstract_attention.py:72-78 differentiated; ABI 10, ea_dgrad_rs.hip):
 *   dx[rows, in] = dy[rows, out] w[out, in]          in = 192, out = 576 (ea_linear_dgrad_supported != 0)
 * dy: EA dtype, row stride ldy elements; w: the layer's weight [out, in] contiguous -- the fp32 master (w_f32 != 0, rounded
 * to the EA dtype on its way into the registers) or a copy in the EA dtype (e.g. ea_linear_w32_pool's w_cast); dx: fp32
 * (dx_f32 != 0: the gradient of an fp32 module input under autocast) or the EA dtype, row stride ldx elements.  The weight
 * stays resident in the registers of a 12-wave workgroup per CU, dy passes LDS once: no library GEMM, no cast kernel. */
int32_t ea_linear_dgrad_supported(int32_t in_features, int32_t out_features);
int ea_linear_dgrad(int32_t dtype, int32_t rows, int32_t in_features, int32_t out_features, const void* dy, int64_t ldy,
                    const void* w, int32_t w_f32, void* dx, int32_t dx_f32, int64_t ldx, void* stream);

/* ... and the same product FUSED with the last corrections of dq / dk of a 192-wide, three-head layer (ABI 10): one pass over
 * the gradient rows instead of ea_lara_bwd_finish (or ea_eva_chunk_mean_bwd) followed by the product:
 *     dq_n -= s sum_c t[c,n] (u_c qbar_c)     (uq != NULL; lara.py:223 differentiated; qbar, uq [B*3, C, 64], lse_t [B*3, C], C <= 64;
 *                                              qkv: the forward's [B*gh*gw, 576] rows, q = columns 0 .. 191)
 *     dq_n += dpq[cell(n)] / r^2,  dk_n += dpk[cell(n)] / r^2   (dpq != NULL: fp32 [B*3, (gh/r)(gw/r), 64]; lara.py:43,48,145-151,
 *                                              eva.py:178-181 differentiated)
 *     dx = [dq | dk | dv] w
 * dqkv [B*gh*gw, 576] (EA dtype) is read AND its dq / dk columns rewritten with the corrected rows (the weight-gradient pass
 * reads them).  w, dx as for ea_linear_dgrad. */
int ea_linear_dgrad_finish(int32_t dtype, int32_t B, int32_t gh, int32_t gw, int32_t pool_r, int32_t C, float scale,
                           void* dqkv, int64_t ldy, const void* qkv, int64_t ldq, const void* w, int32_t w_f32, void* dx,
                           int32_t dx_f32, int64_t ldx, const float* qbar, const float* uq, const float* lse_t,
                           const float* dpq, const float* dpk, void* stream);

/* Round 5 (ABI 10): consumer passes that merge the producing pass's per-slice partials in their prologue -- the arithmetic of
 * ea_lara_merge_fwd / _bwd and ea_slice_sum, operation for operation -- so a layer step has three launches less (7 + 14 + 6
 * us at cfg3 for a few KB per (b,h)).  S <= 4 slices, C <= 64 samples (EA_E_UNSUPPORTED otherwise: keep the merge launches).
 *   ea_lara_out_fwd_merge: ea_lara_out_fwd fed by ea_lara_stats_fwd's partials (p_ml [BH,S,C,4], p_kv [BH,S,C,D]) and lp;
 *       block 0 of every (b,h) writes the merged kv, lse_k, lse_t, cst for the backward (lara.py:205-211,241-246).
 *   ea_lara_bwd_k_fused_merge: ea_lara_bwd_k_fused fed by ea_lara_bwd_q_fused's partials (p_ml, p_dkv, p_dom, p_m1, p_m2);
 *       forms d kv_stats, dkk, r itself; block 0 writes dbh, dlp (= -r), domq (sum dZ q), dqbar, uq (NULL-able as in
 *       ea_lara_merge_bwd); p_domk [BH,S,C,D]: its own d omega partials (must not alias the inputs).
 *   ea_lara_landmarks_bwd_parts: ea_lara_landmarks_bwd with d omega = dom_scale (d_omega + sum_s dom_parts[bh][s]) formed on
 *       load (ea_slice_sum's order of additions). */
int ea_lara_out_fwd_merge(const ea_lara_geom* g, const ea_t4* q, const float* omega, const float* qbar, const float* bhv,
                          int32_t S, const float* p_ml, const float* p_kv, const float* lp, float* kv, float* lse_k,
                          float* lse_t, float* cst, const ea_t4* out, float* lseZ, float* tmean, void* stream);
int ea_lara_bwd_k_fused_merge(const ea_lara_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* omega,
                              const float* qbar, const float* kv, const float* lse_k, int32_t S, const float* p_ml,
                              const float* p_dkv, const float* p_dom, const float* p_m1, const float* p_m2,
                              const ea_t4* dk, const ea_t4* dv, float* p_domk, float* dbh, float* dlp, float* domq,
                              float* dqbar, float* uq, void* stream);
int ea_lara_landmarks_bwd_parts(const ea_lmk_geom* g, const float* pq, const float* pk,
                                const float* Wq, const float* bq, const float* gq, const float* cq,
                                const float* Wk, const float* bk, const float* gk, const float* ck,
                                const float* noise, const float* d_omega, int32_t dom_S, const float* dom_parts, float dom_scale,
                                const float* d_qbar_rows, const float* d_bhv, const float* d_lp, float* dpq, float* dpk,
                                float* dW_part, float* dvec_part, const float* saved, void* stream);

/* ---- fp32-FAITHFUL cores (round 5, ABI 10; ea_f32_attn.hip) -------------------------------------------------------------
 * Outside torch.autocast the reference computes attention in fp32 (abstract_attention.py:120-133, local_attention.py:134-182,
 * eva.py:138-233, causal_eva.py:666-783).  One gathered-attention pair in exact fp32 arithmetic (operands, products on
 * v_mfma_f32_16x16x4_f32, softmax) covers every softmax-shaped core, stated as the reference states them:
 *   group g (a window, a landmark chunk, or the whole sequence), query slot i -> token idx_q[g][i] of q [B,H,Nq,D],
 *   Wk local key slots j -> token idx_k[g][j] of k, v [B,H,Nk,D] (-1: absent = zero row, masked), then L extra keys / values
 *   ek, ev [B,H,L,D] shared by all groups (EVA: rf_k_bar / beta):
 *     logit_ij = scale q_i.k_j [- scale |k_j|^2 / 2 (knorm)] [+ bias[b bias_bs + h bias_hs + i bias_ld + j]]
 *     masked (padded / absent key; padded query (qmask); causal_e >= 0 and j > i + causal_e): -5e4, or -inf for padded keys when
 *     neg_inf; extra key c masked (-5e4) when chunk > 0 and c >= lm_base + token(i) / chunk  (causal_eva.py:716-738)
 *     out_i = softmax over the Wk + L columns . [v ; ev] (dropout: keep [B,H,Nq,keep_ld] over those columns, kept entries x
 *     keep_scale in the value product only), lse_i = log-sum-exp (natural log, optional); stat [B,H,Nq,2] = (row max, sum of
 *     e^(logit - max)) is what the backward recomputes the probabilities from (lse alone loses the sum's digits when the
 *     maximum is the -5e4 fill of a fully masked row).
 * All tensors fp32 (ea_t4 strides in elements).  Backward: dq stored; dk, dv [B,H,Nk,D], dek, dev [B,H,L,D], dbias (bias layout)
 * are contiguous fp32 buffers ACCUMULATED into with atomics (the caller zeroes them; windows overlap); dlse optional. */
typedef struct {
  int32_t B, H, Nq, Nk, D;           /* D in {32, 64, 128} */
  int32_t G, Wq, Wk, L;
  int32_t knorm;                     /* bit 0: the key-norm term; bit 1: masked local keys carry a zero VALUE row (eva.py:167-176) */
  int32_t neg_inf, causal_e, chunk, lm_base;
  int32_t bias_ld;
  int64_t bias_hs, bias_bs, keep_ld;  /* bias index = b bias_bs + h bias_hs + i bias_ld + j (0 strides: shared) */
  float   keep_scale, scale;
} ea_f32_attn;
int ea_f32_attn_fwd(const ea_f32_attn* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const ea_t4* ek, const ea_t4* ev,
                    const int32_t* idx_q, const int32_t* idx_k, const float* bias, const uint8_t* kmask, const uint8_t* qmask,
                    const uint8_t* keep, const ea_t4* out, float* lse, float* stat, void* stream);
int ea_f32_attn_bwd(const ea_f32_attn* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const ea_t4* ek, const ea_t4* ev,
                    const int32_t* idx_q, const int32_t* idx_k, const float* bias, const uint8_t* kmask, const uint8_t* qmask,
                    const uint8_t* keep, const ea_t4* out, const ea_t4* dout, const float* stat, const float* dlse,
                    const ea_t4* dq, float* dk, float* dv, float* dek, float* dev, float* dbias, void* stream);
/* mean[b,h,c,:] = (1/J) sum_j x[b,h,idx[c][j],:] over the present, unpadded tokens (EVA's masked chunk means, eva.py:167-181;
 * uniform pooling); backward accumulates into dx [B,H,N,D] contiguous fp32 (atomics). */
int ea_f32_gather_mean_fwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t Cn, int32_t J, const ea_t4* x, const int32_t* idx,
                           const uint8_t* mask, float* mean, void* stream);
int ea_f32_gather_mean_bwd(int32_t B, int32_t H, int32_t N, int32_t D, int32_t Cn, int32_t J, const int32_t* idx,
                           const uint8_t* mask, const float* dmean, float* dx, void* stream);

/* ---- composite per-module entry points: the whole LARA core in one call each way (round 3) --------------------
 * lara.py:129-175,187-246 for the 2-D pooled proposals ('pool', 'pool-mixed'): uniform r x r pooling of q, k -> landmark
 * pipeline -> estimator, i.e. the launch sequences of ea_eva_chunk_mean_fwd / ea_lara_landmarks_* / ea_lara_stats_fwd /
 * ea_lara_merge_* / ea_lara_out_fwd / ea_lara_bwd_*_fused / ea_slice_sum / ea_lara_bwd_finish / ea_colsum2_f32 issued from
 * C++ on caller-owned workspaces, so that an eagerly stepping caller (vit/engine.py:47-64) pays two FFI calls per layer
 * step instead of ~15 (and two allocations instead of ~30).  C = L (x 2 with antithetic / multi-sample noise) <= 64.
 *   ea_lara_layer_ws(cfg, which): floats of workspace 0 = `saved` (forward -> backward), 1 = forward scratch, 2 = backward
 *       scratch; 3 / 4 = offsets (floats) of the pooled q / k rows [B*H, L, D] inside `saved`; 5 / 6 = offsets inside the
 *       backward scratch of the per-(b,h) parameter-gradient partials [B*H, 2 D D] (dW_q, dW_k) and [B*H, 6 D]
 *       (negative: EA_E_*).  ea_lara_layer_bwd with dparams == NULL leaves those partials to the caller (ea_multi_sum).
 *   keep_for_backward: bit 0 = keep the intermediates the backward needs; bit 1 (EA_LARA_POOLED_READY) = the pooled q / k
 *       rows are already in `saved` at those offsets (written by ea_linear_w32_pool): the pooling pass is skipped.
 *   params: NULL or 8 pointers (Wq, bq, gamma_q, beta_q, Wk, bk, gamma_k, beta_k: q_bar_gen / k_bar_gen, lara.py:45-54);
 *   noise: [B*H, C, D] standard normal or NULL (eval); dparams: [2*D*D + 6*D] fp32 = dW_q, dW_k, then (db, dgamma, dbeta)
 *   of q and of k, summed over the batch and heads. */
typedef struct {
  int32_t B, H, D;
  int32_t dtype;             /* EA_BF16 | EA_F16 */
  int32_t gh, gw;            /* token grid (N = gh * gw) */
  int32_t pool_r;            /* pooling side: landmarks L = (gh / r) * (gw / r) */
  int32_t has_mlp, mixed, mis, dup;   /* as in ea_lmk_geom */
  float   kappa, scale;      /* alpha_coeff (lara.py:231), D^-0.5 */
} ea_lara_layer;
#define EA_LARA_POOLED_READY 2
int64_t ea_lara_layer_ws(const ea_lara_layer* cfg, int32_t which);
int ea_lara_layer_fwd(const ea_lara_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                      const float* noise, const float* const* params, const ea_t4* out, float* saved, float* tmp,
                      int32_t keep_for_backward, void* stream);
int ea_lara_layer_bwd(const ea_lara_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                      const float* noise, const float* const* params, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                      const ea_t4* dv, const float* saved, float* tmp, float* dparams, void* stream);
/* ABI 10: flags = EA_LARA_DEFER_FINISH leaves out the finish pass -- dq lacks the softmax-over-sequence correction and dq, dk the
 * pooling terms; ea_linear_dgrad_finish applies them on its way (operands at ea_lara_layer_ws(cfg, 7..9) inside the backward
 * scratch and (10, 11) inside `saved`). */
#define EA_LARA_DEFER_FINISH 1
int ea_lara_layer_bwd2(const ea_lara_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                       const float* noise, const float* const* params, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                       const ea_t4* dv, const float* saved, float* tmp, float* dparams, int32_t flags, void* stream);

/* ---- composite per-module entry points, EVA (round 4) -----------------------------------------------------------
 * The 2-D EVA core (eva.py:145-227: non-overlapping w x w windows, r x r landmark chunks, adaptive_proj 'default', no pad
 * mask) in one call each way -- the launch sequences of ea_eva_chunk_mean_fwd / ea_lara_landmarks_* (eva mode) /
 * ea_eva_beta_* / ea_window_attn_* / ea_slice_sum / ea_colsum_f32 / ea_eva_chunk_mean_bwd / ea_colsum2_f32 on caller-owned
 * workspaces, for the same reason as the LARA pair above (eagerly stepping call sites, vit/engine.py:47-64).
 *   ea_eva_layer_ws(cfg, which): 0 = floats of `saved` (forward -> backward), 1 = 0 (no forward scratch), 2 = floats of the
 *       backward scratch; 3 / 4 = offsets (floats) inside `saved` of the chunk means of q / k [B*H, L, D]; 5 / 6 = offsets
 *       inside the backward scratch of the per-(b,h) parameter-gradient partials [B*H, 2 D D] and [B*H, 6 D]; 7 = padded
 *       row length of `bias` (= ea_window_bias_ld); 8 = offset inside `saved` of the joint log-sum-exp [B,H,N]
 *       (negative: EA_E_*; EA_E_UNSUPPORTED: L > 64 or a window geometry whose backward needs scratch slices -- use the
 *       step-by-step entry points).
 *   bias: fp32 [H, w*w, ld] dense per-head bias multiplied by log2(e), rows padded to ld (as for ea_window_attn_fwd), or
 *       NULL (then cfg->has_bias == 0); dbias: fp32 [H, w*w, ld], gradient with respect to the NATURAL-unit bias, or NULL:
 *       its partial sums then stay in the backward scratch -- ea_eva_layer_ws(cfg, 9) = their offset, (cfg, 10) = their row
 *       count, rows of H * w*w * ld floats -- for the caller's own reduction (ea_multi_sum, with the other terminal sums).
 *   params: 8 pointers (W, b, gamma, beta of the q and of the k mu network: eva.py:93-103); noise: [B*H, L, D] or NULL.
 *   keep_for_backward: bit 0 = keep the intermediates; bit 1 (EA_LARA_POOLED_READY) = the chunk means are already in
 *       `saved` (ea_linear_w32_pool wrote them).  dparams: [2 D D + 6 D] fp32 as for ea_lara_layer_bwd, or NULL (partials
 *       left in the scratch for the caller's reduction).  dq, dk, dv are WRITTEN. */
typedef struct {
  int32_t B, H, D;
  int32_t dtype;             /* EA_BF16 | EA_F16 */
  int32_t gh, gw;            /* token grid (N = gh * gw) */
  int32_t window;            /* window side w (gh, gw multiples of it; no overlap extension) */
  int32_t chunk;             /* landmark chunk side r: L = (gh / r) * (gw / r) <= 64 */
  int32_t has_bias;
  float   scale;             /* D^-0.5 */
} ea_eva_layer;
int64_t ea_eva_layer_ws(const ea_eva_layer* cfg, int32_t which);
int ea_eva_layer_fwd(const ea_eva_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                     const float* noise, const float* const* params, const ea_t4* out, float* saved,
                     int32_t keep_for_backward, void* stream);
int ea_eva_layer_bwd(const ea_eva_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                     const float* noise, const float* const* params, const ea_t4* out, const ea_t4* dout, const ea_t4* dq,
                     const ea_t4* dk, const ea_t4* dv, const float* saved, float* tmp, float* dbias, float* dparams,
                     void* stream);
/* ABI 10: flags = EA_EVA_DEFER_CHUNK_MEAN leaves out the chunk-mean backward (eva.py:178-181 differentiated): dq / dk lack the
 * d(chunk mean) / r^2 terms, which ea_linear_dgrad_finish adds on its way (their gradients [B*H, L, D] fp32 sit at
 * ea_eva_layer_ws(cfg, 11 / 12) inside the backward scratch). */
#define EA_EVA_DEFER_CHUNK_MEAN 1
int ea_eva_layer_bwd2(const ea_eva_layer* cfg, const ea_t4* q, const ea_t4* k, const ea_t4* v, const float* bias,
                      const float* noise, const float* const* params, const ea_t4* out, const ea_t4* dout, const ea_t4* dq,
                      const ea_t4* dk, const ea_t4* dv, const float* saved, float* tmp, float* dbias, float* dparams,
                      int32_t flags, void* stream);

/* ---- row LayerNorm (ea_layernorm.hip) -------------------------------------------------------------------------
 * The LayerNorm of LinearRA's model-wide ('dense') landmark generators (lara.py:34-44,64-71: Linear(dim, dim) + LayerNorm(dim)
 * on the B * L pooled rows).  x [rows, C] contiguous, xtype EA_BF16 / EA_F16 / EA_F32 (under autocast torch evaluates
 * layer_norm in fp32 on the 16-bit Linear output); C a multiple of 64, <= 1024 (EA_E_UNSUPPORTED otherwise).
 *   fwd: y fp32 [rows, C] = (x - mean) rstd gamma + beta, biased variance, two-pass statistics in fp32; stats [rows, 2] =
 *        (mean, rstd) for the backward (may be NULL).
 *   bwd: dx [rows, C] in x's type; part [ea_layernorm_parts(rows), 2, C] fp32 = per-workgroup partial sums of d gamma and
 *        d beta (add them with ea_colsum_f32(parts, 2 * C, ...)). */
int32_t ea_layernorm_parts(int32_t rows);
int ea_layernorm_fwd(int32_t xtype, int32_t rows, int32_t C, const void* x, const float* gamma, const float* beta, float eps,
                     float* y, float* stats, void* stream);
int ea_layernorm_bwd(int32_t xtype, int32_t rows, int32_t C, const void* x, const float* gamma, const float* stats,
                     const float* dy, void* dx, float* part, void* stream);

/* ---- ScatterBrain, low-rank half (scatterbrain_attention.py:99-160; ea_scatter.hip) --------------------
 * The window half is ea_window_attn_fwd/bwd (it returns / takes the gradient of its per-query log-sum-exp);
 * these entry points evaluate the m random-feature columns of the same softmax and merge the two halves:
 *   ea_scatter_kmax / ea_scatter_kv : mx[c] = max_j log phi(k_j)[c] and the partial sums of
 *       z_all[c] = sum_j exp(lk_jc - mx_c), S_all[c] = sum_j exp(lk_jc - mx_c) v_j  over sequence slices
 *       (p_ml [BH,S,M,4] (first field), p_kv [BH,S,M,64], S = ea_scatter_parts(g)); padded keys excluded;
 *   ea_scatter_fwd : per window, statistics of the window's own keys, KV = (S_all - S_win) / clamp(z_all -
 *       z_win, 1e-3), joint weights of the feature columns, and out = sigma(lse_loc - r) o_loc + sigma(r -
 *       lse_loc) O with r [B,H,N] = log-sum-exp of the feature logits (returned for the backward).
 * D = 64, M <= 64 features, windows of <= 64 tokens, no window overlap.  W [H, M, 64] fp32. */
typedef struct {
  int32_t B, H, N, D;
  int32_t dtype;
  int32_t M;                 /* random features */
  int32_t attn_2d, gh, gw;   /* token grid (2-D) */
  int32_t window;            /* window side */
} ea_sb_geom;
int32_t ea_scatter_parts(const ea_sb_geom* g);
int ea_scatter_kmax(const ea_sb_geom* g, const ea_t4* k, const uint8_t* mask, const float* W, float* p_ml, void* stream);
int ea_scatter_kv(const ea_sb_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                  const float* mx, float* p_ml, float* p_kv, void* stream);
int ea_scatter_fwd(const ea_sb_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                   const float* W, const float* mx, const float* zall, const float* sall, const ea_t4* oloc,
                   const float* lse_loc, const ea_t4* out, float* r, void* stream);
/* Backward of ea_scatter_fwd (mx is a detached stabiliser, as in the reference):
 *   ea_scatter_bwd_window: per window -- dq, the window's own contributions to dk / dv (written, not
 *       accumulated), d o_loc = alpha dout and d lse_loc [B,H,N] (the cotangents of the window half, to be fed to
 *       ea_window_attn_bwd), and partial sums p_dsall [BH, P, M, 64], p_dzall [BH, P, M] of d S_all, d z_all,
 *       P = ea_scatter_bwd_parts(g);
 *   ea_scatter_bwd_global: given the summed d S_all, d z_all, ADDS every key's share to dk, dv. */
int32_t ea_scatter_bwd_parts(const ea_sb_geom* g);
int ea_scatter_bwd_window(const ea_sb_geom* g, const ea_t4* q, const ea_t4* k, const ea_t4* v, const uint8_t* mask,
                          const float* W, const float* mx, const float* zall, const float* sall, const ea_t4* oloc,
                          const float* lse_loc, const float* r, const ea_t4* dout, const ea_t4* dq, const ea_t4* dk,
                          const ea_t4* dv, const ea_t4* doloc, float* dlse, float* p_dsall, float* p_dzall,
                          void* stream);
int ea_scatter_bwd_global(const ea_sb_geom* g, const ea_t4* k, const ea_t4* v, const uint8_t* mask, const float* W,
                          const float* mx, const float* dsall, const float* dzall, const ea_t4* dk, const ea_t4* dv,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EA_HIP_H */
