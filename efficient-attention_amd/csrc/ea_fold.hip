// ea_fold.hip -- LARA 'adaptive-1d' proposals (lara.py:84-127): the per-token Linear of q_bar_gen / k_bar_gen folded into
// the qkv projection (round 1: W' = W_gen W_head as two more groups of output columns of the same GEMM).  Building the
// extended weight and taking its gradient apart were ~30 framework kernels per step (cat / permute copies / small GEMMs /
// casts / adds: a quarter of the cfg5 step at the recipe's batch of one, `--max-tokens 4096`); here they are one launch each
// way (round 4).
//   forward : w_ext [5C, C] (element type) = [ W ; Gq W_q,head ; Gk W_k,head ],  b_ext [5C] = [ b ; 0 ],
//             bias_q [h, d] = Gq b_q,head + gq_b,  bias_k likewise (the bias of a folded row, applied by ea_lara_segment_*)
//   backward: dW [3C, C] = dW_ext[:3C] + Gq^T dW_ext[3C + ...] (q rows) + Gk^T ... (k rows);
//             dGq [d, d] = sum_head dW_ext,q'[head] W_q[head]^T + sum_head dbias_q[head] b_q[head]^T;   d gq_b = sum_head dbias_q
//             db [3C] = db_ext[:3C] + Gq^T dbias_q (q part) + Gk^T dbias_k (k part)
#include "ea_common.h"
#include "ea_f32_mm.h"

namespace ea {

struct FoldP {
  const float *W, *b, *Gq, *Gk, *gqb, *gkb;     // [3C, C], [3C] | null, [d, d] x 2, [d] x 2
  char* w_ext;                                   // [5C, C] element type
  char* b_ext;                                   // [5C] element type | null
  float *bias_q, *bias_k;                        // [h, d]
  // backward
  const float *dW_ext, *db_ext, *dbias_q, *dbias_k;   // [5C, ldw], [5C] | null, [h, d] x 2
  float *dW, *db, *dGq, *dGk, *dgqb, *dgkb;            // [3C, C], [3C] | null, [d, d] x 2, [d] x 2
  float *dG_part, *dG_base;                            // [2][h FB_SPLIT][64 x 64] partials, [2][64 x 64] bias part
  long ldw;
  int C, h, d;
};

// 64 x 64 fp32 tile of a row-major matrix (row stride ld) -> LDS image [64][65]
EA_DEV void fold_load(float* dst, const float* src, long ld, int tid) {
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + 256 * u, r = idx >> 4, c = (idx & 15) * 4;
    v[u] = *reinterpret_cast<const float4*>(src + (size_t)r * ld + c);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + 256 * u, r = idx >> 4, c = (idx & 15) * 4;
    float* d = dst + r * 65 + c;
    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
  }
}

// Forward.  blocks [0, 3C): cast rows of W;  then 2 h (C / 64) tile blocks: w_ext[3C + side C + head 64 + i][c] =
// sum_j G_side[i][j] W[side C + head 64 + j][c] as one [64 x 64] x [64 x 64] product out of LDS (d = 64);  last block: b_ext
// and the biases of the folded rows.  (A first version without tiles re-read the head's 64 weight rows for every output
// row: 134 MB through the CUs' L2 ports, 40 us; the backward's dG part re-read a whole side per output row: 97 us.)
template <typename E>
__global__ __launch_bounds__(256) void fold_fwd_kernel(const FoldP p) {
  __shared__ float Gs[64 * 65], Ws[64 * 65];
  const int C = p.C, d = p.d, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nct = C / 64, ntile = 2 * p.h * nct;
  const int blk = blockIdx.x;
  if (blk < 3 * C) {
    uint16_t* out = reinterpret_cast<uint16_t*>(p.w_ext) + (size_t)blk * C;
    for (int c = tid; c < C; c += 256) out[c] = E::from_f(p.W[(size_t)blk * C + c]);
    return;
  }
  if (blk < 3 * C + ntile) {
    const int t = blk - 3 * C, side = t / (p.h * nct), hr = t - side * p.h * nct, head = hr / nct, c0 = (hr - head * nct) * 64;
    fold_load(Gs, side ? p.Gk : p.Gq, d, tid);
    fold_load(Ws, p.W + ((size_t)side * C + (size_t)head * 64) * C + c0, C, tid);
    __syncthreads();
    uint16_t* out = reinterpret_cast<uint16_t*>(p.w_ext) + ((size_t)3 * C + (size_t)side * C + (size_t)head * 64) * C + c0;
    for (int tt = wave; tt < 16; tt += 4) {
      const int m0 = (tt >> 2) * 16, n0 = (tt & 3) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      tile_mm<false, false, 64>(acc, Gs, 65, Ws, 65, m0, n0, 64, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(m0 + 4 * g + r) * C + n0 + li] = E::from_f(acc[r]);
    }
    return;
  }
  // last block: b_ext and the biases of the folded rows
  if (p.b_ext) {
    uint16_t* be = reinterpret_cast<uint16_t*>(p.b_ext);
    for (int c = tid; c < 5 * C; c += 256) be[c] = E::from_f(c < 3 * C && p.b ? p.b[c] : 0.f);
  }
  for (int idx = tid; idx < 2 * p.h * d; idx += 256) {
    const int side = idx / (p.h * d), hr = idx - side * p.h * d, head = hr / d, i = hr - head * d;
    const float* G = (side ? p.Gk : p.Gq) + (size_t)i * d;
    float a = (side ? p.gkb : p.gqb)[i];
    if (p.b) {
      const float* bh = p.b + side * C + head * 64;
#pragma unroll                                              // (d = 64: every load of the dot product in flight at once)
      for (int j = 0; j < 64; ++j) a = fmaf(G[j], bh[j], a);
    }
    (side ? p.bias_k : p.bias_q)[head * d + i] = a;
  }
}

// Backward.  blocks [0, C): copy the v rows of dW;  then 2 h FB_SPLIT tile blocks (side, head, group of column tiles): per
// 64-column tile  dW[side C + head 64 + j][c] = dW_ext[same][c] + sum_i G[i][j] dW2[i][c]  and, accumulated over the
// group's tiles, the partial  dG[i][j] += sum_c dW2[i][c] W[head 64 + j][c]  -> dG_part [2][h FB_SPLIT][64 x 64] (added by
// ea_slice_sum together with the bias term the next 32 blocks leave in dG_base);  last block: d g_b, db.  (All scalar loops
// have compile-time trip counts: a run-time loop of dependent-latency loads in ONE block was 30-50 us of each launch.)
constexpr int FB_SPLIT = 8;          // (2: 35 us -- four dependent tile trips per block; 8: one trip, 64 partials per side)
__global__ __launch_bounds__(256) void fold_bwd_kernel(const FoldP p) {
  __shared__ float Gs[64 * 65], Ws[64 * 65], Ds[64 * 65];
  const int C = p.C, d = p.d, h = p.h, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int nct = C / 64, ntile = 2 * h * FB_SPLIT;
  const int blk = blockIdx.x;
  if (blk < C) {
    const int r = 2 * C + blk;
    for (int c = tid; c < C; c += 256) p.dW[(size_t)r * C + c] = p.dW_ext[(size_t)r * p.ldw + c];
    return;
  }
  if (blk < C + ntile) {
    const int t = blk - C, side = t / (h * FB_SPLIT), hr = t - side * h * FB_SPLIT, head = hr / FB_SPLIT, part = hr - head * FB_SPLIT;
    fold_load(Gs, side ? p.Gk : p.Gq, d, tid);
    f32x4 dg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t row0 = (size_t)side * C + (size_t)head * 64;
    for (int ct = part; ct < nct; ct += FB_SPLIT) {
      const int c0 = ct * 64;
      __syncthreads();
      fold_load(Ds, p.dW_ext + ((size_t)3 * C + row0) * p.ldw + c0, p.ldw, tid);        // dW2[i][c]
      fold_load(Ws, p.W + row0 * C + c0, C, tid);                                        // W[j][c]
      __syncthreads();
      for (int tt = wave, k = 0; tt < 16; tt += 4, ++k) {
        const int m0 = (tt >> 2) * 16, n0 = (tt & 3) * 16;
        // dG[i = m][j = n] += sum_c Ds[i][c] Ws[j][c]
        tile_mm<false, true, 64>(dg[k], Ds, 65, Ws, 65, m0, n0, 64, lane);
        // dW[j = m][c = n] = dW_ext[j][c] + sum_i Gs[i][j] Ds[i][c]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        tile_mm<true, false, 64>(acc, Gs, 65, Ds, 65, m0, n0, 64, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t row = row0 + m0 + 4 * g + r;
          p.dW[row * C + c0 + n0 + li] = acc[r] + p.dW_ext[row * p.ldw + c0 + n0 + li];
        }
      }
    }
    float* part_out = p.dG_part + ((size_t)side * h * FB_SPLIT + hr) * 64 * 64;
    for (int tt = wave, k = 0; tt < 16; tt += 4, ++k) {
      const int m0 = (tt >> 2) * 16, n0 = (tt & 3) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) part_out[(m0 + 4 * g + r) * 64 + n0 + li] = dg[k][r];
    }
    return;
  }
  // the 32 blocks after the tiles: the bias part of dG (one output per thread)
  if (blk < C + ntile + 32) {
    const int idx = (blk - C - ntile) * 256 + tid;
    const int side = idx / (d * d), ij = idx - side * d * d, i = ij / d, j = ij - i * d;
    float a = 0.f;
    if (p.b) {
#pragma unroll 8
      for (int head = 0; head < h; ++head) a = fmaf((side ? p.dbias_k : p.dbias_q)[head * d + i], p.b[side * C + head * d + j], a);
    }
    p.dG_base[idx] = a;
    return;
  }
  // last block: d gq_b / d gk_b and the bias gradient of the qkv Linear
  for (int idx = tid; idx < 2 * d; idx += 256) {
    const int side = idx / d, i = idx - side * d;
    float a = 0.f;
    for (int head = 0; head < h; ++head) a += (side ? p.dbias_k : p.dbias_q)[head * d + i];
    (side ? p.dgkb : p.dgqb)[i] = a;
  }
  if (p.db) {
    for (int c = tid; c < 3 * C; c += 256) {
      float a = p.db_ext ? p.db_ext[c] : 0.f;
      if (c < 2 * C) {
        const int side = c / C, hr = c - side * C, head = hr / d, j = hr - head * d;
        const float* G = side ? p.Gk : p.Gq;
        const float* dbf = (side ? p.dbias_k : p.dbias_q) + head * d;
#pragma unroll
        for (int i = 0; i < 64; ++i) a = fmaf(G[(size_t)i * 64 + j], dbf[i], a);
      }
      p.db[c] = a;
    }
  }
}

int fold_bwd_parts(int heads) { return heads * FB_SPLIT; }

int fold_dispatch(bool bwd, int dtype, const FoldP& p, hipStream_t st) {
  if (p.C <= 0 || p.h <= 0 || p.d != 64 || p.h * p.d != p.C) return EA_E_UNSUPPORTED;     // d = 64 tiles
  const int nct = p.C / 64;
  if (!bwd) {
    const dim3 grid((unsigned)(3 * p.C + 2 * p.h * nct + 1)), block(256);
    if (dtype == EA_BF16) hipLaunchKernelGGL(fold_fwd_kernel<BF16>, grid, block, 0, st, p);
    else if (dtype == EA_F16) hipLaunchKernelGGL(fold_fwd_kernel<F16>, grid, block, 0, st, p);
    else return EA_E_BADARG;
  } else {
    if ((p.ldw & 3) || ((uintptr_t)p.dW_ext & 15)) return EA_E_BADARG;
    hipLaunchKernelGGL(fold_bwd_kernel, dim3((unsigned)(p.C + 2 * p.h * FB_SPLIT + 32 + 1)), dim3(256), 0, st, p);
  }
  return (int)hipGetLastError();
}

}  // namespace ea
