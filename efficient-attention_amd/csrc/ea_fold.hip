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

namespace ea {

struct FoldP {
  const float *W, *b, *Gq, *Gk, *gqb, *gkb;     // [3C, C], [3C] | null, [d, d] x 2, [d] x 2
  char* w_ext;                                   // [5C, C] element type
  char* b_ext;                                   // [5C] element type | null
  float *bias_q, *bias_k;                        // [h, d]
  // backward
  const float *dW_ext, *db_ext, *dbias_q, *dbias_k;   // [5C, ldw], [5C] | null, [h, d] x 2
  float *dW, *db, *dGq, *dGk, *dgqb, *dgkb;            // [3C, C], [3C] | null, [d, d] x 2, [d] x 2
  long ldw;
  int C, h, d;
};

template <typename E>
__global__ __launch_bounds__(256) void fold_fwd_kernel(const FoldP p) {
  const int C = p.C, d = p.d, tid = threadIdx.x;
  const int r = blockIdx.x;
  if (r < 5 * C) {
    uint16_t* out = reinterpret_cast<uint16_t*>(p.w_ext) + (size_t)r * C;
    if (r < 3 * C) {
      for (int c = tid; c < C; c += 256) out[c] = E::from_f(p.W[(size_t)r * C + c]);
    } else {
      const int side = (r - 3 * C) / C, hr = (r - 3 * C) - side * C, head = hr / d, i = hr - head * d;
      const float* G = (side ? p.Gk : p.Gq) + (size_t)i * d;
      const float* Wh = p.W + ((size_t)side * C + (size_t)head * d) * C;
      for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < d; ++j) a = fmaf(G[j], Wh[(size_t)j * C + c], a);
        out[c] = E::from_f(a);
      }
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
    if (p.b)
      for (int j = 0; j < d; ++j) a = fmaf(G[j], p.b[side * C + head * d + j], a);
    (side ? p.bias_k : p.bias_q)[head * d + i] = a;
  }
}

// blocks [0, 3C): rows of dW;  [3C, 3C + 2d): rows of dGq / dGk;  last: db, d gq_b, d gk_b
__global__ __launch_bounds__(256) void fold_bwd_kernel(const FoldP p) {
  __shared__ float red[4][64];
  const int C = p.C, d = p.d, h = p.h, tid = threadIdx.x;
  const int blk = blockIdx.x;
  if (blk < 3 * C) {
    const int r = blk;
    float* out = p.dW + (size_t)r * C;
    const float* src = p.dW_ext + (size_t)r * p.ldw;
    if (r >= 2 * C) {
      for (int c = tid; c < C; c += 256) out[c] = src[c];
      return;
    }
    const int side = r / C, hr = r - side * C, head = hr / d, j = hr - head * d;
    const float* G = side ? p.Gk : p.Gq;                                 // column j of G
    const float* dWf = p.dW_ext + ((size_t)3 * C + (size_t)side * C + (size_t)head * d) * p.ldw;
    for (int c = tid; c < C; c += 256) {
      float a = src[c];
      for (int i = 0; i < d; ++i) a = fmaf(G[(size_t)i * d + j], dWf[(size_t)i * p.ldw + c], a);
      out[c] = a;
    }
    return;
  }
  if (blk < 3 * C + 2 * d) {
    // dG_side[i][j] = sum_head sum_c dW_ext[3C + side C + head d + i][c] W[side C + head d + j][c] + sum_head dbias[head][i] b[..j]
    const int side = (blk - 3 * C) / d, i = (blk - 3 * C) - side * d;
    // a wave per (head, j) pair: both rows read along c by the 64 lanes (coalesced), fixed-order wave sum
    const int lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < 4 * 64; j += 256) red[j >> 6][j & 63] = 0.f;
    __syncthreads();
    for (int pair = wave; pair < h * d; pair += 4) {
      const int head = pair / d, j = pair - head * d;
      const float* dr = p.dW_ext + ((size_t)3 * C + (size_t)side * C + (size_t)head * d + i) * p.ldw;
      const float* wr = p.W + ((size_t)side * C + (size_t)head * d + j) * C;
      float a = 0.f;
      for (int c = lane; c < C; c += 64) a = fmaf(dr[c], wr[c], a);
      a = wave_sum(a);
      if (lane == 0) {
        if (p.b) a = fmaf((side ? p.dbias_k : p.dbias_q)[head * d + i], p.b[side * C + head * d + j], a);
        red[wave][j] += a;                                   // (this wave only: heads of a j in increasing order)
      }
    }
    __syncthreads();
    if (tid < d) (side ? p.dGk : p.dGq)[(size_t)i * d + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    return;
  }
  // d gq_b / d gk_b and the bias gradient of the qkv Linear
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
        for (int i = 0; i < d; ++i) a = fmaf(G[(size_t)i * d + j], dbf[i], a);
      }
      p.db[c] = a;
    }
  }
}

int fold_dispatch(bool bwd, int dtype, const FoldP& p, hipStream_t st) {
  if (p.C <= 0 || p.h <= 0 || p.d <= 0 || p.d > 64 || p.h * p.d != p.C || (p.C & 3)) return EA_E_UNSUPPORTED;
  if (!bwd) {
    const dim3 grid((unsigned)(5 * p.C + 1)), block(256);
    if (dtype == EA_BF16) hipLaunchKernelGGL(fold_fwd_kernel<BF16>, grid, block, 0, st, p);
    else if (dtype == EA_F16) hipLaunchKernelGGL(fold_fwd_kernel<F16>, grid, block, 0, st, p);
    else return EA_E_BADARG;
  } else {
    hipLaunchKernelGGL(fold_bwd_kernel, dim3((unsigned)(3 * p.C + 2 * p.d + 1)), dim3(256), 0, st, p);
  }
  return (int)hipGetLastError();
}

}  // namespace ea
