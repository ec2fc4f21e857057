// ea_proj.hip -- the small reductions around the attention cores:
//   * bias gradient of the qkv / output projections, db[c] = sum_t dY[t][c] over all B*N tokens
//     (torch does this with a generic strided reduce at ~1.7 TB/s; this is a plain 16-B/lane
//     streaming column sum with a fixed-order two-stage reduction, so it is deterministic);
//   * fp32 column sums of per-(b,h) parameter-gradient partials;
//   * scale * (a + sum over sequence slices) of a per-landmark tensor.
#include "ea_common.h"

namespace ea {

constexpr int CS_UNROLL = 8;

// stage 1: block b sums rows [b*rpb, (b+1)*rpb) -> part[b][cols].  A thread owns one 8-column
// group (16 B) and every R-th row of the slab.
template <class T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const uint16_t* __restrict__ x, float* __restrict__ part,
                                                           int rows, int cols, int rpb, int pitch, int pcols) {
  // x: this launch's column window of a [rows, pitch] matrix; part[b][pcols] (window offset applied by the host)
  extern __shared__ float red[];              // [R][cols]
  const int tpr = cols >> 3, R = 256 / tpr, tid = threadIdx.x;
  const int r = tid / tpr, cg = tid - r * tpr;
  const int row0 = blockIdx.x * rpb, row1 = min(rows, row0 + rpb);
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (r < R) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4* base = (const u4*)(x + (size_t)cg * 8);
    const size_t ld = (size_t)pitch >> 3;     // row pitch in uint4
    int row = row0 + r;
    for (; row + (CS_UNROLL - 1) * R < row1; row += CS_UNROLL * R) {
      u4 v[CS_UNROLL];
#pragma unroll
      for (int u = 0; u < CS_UNROLL; ++u) v[u] = __builtin_nontemporal_load(base + (size_t)(row + u * R) * ld);
#pragma unroll
      for (int u = 0; u < CS_UNROLL; ++u) {
        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += T::to_f((uint16_t)(w[i] & 0xffff));
          acc[2 * i + 1] += T::to_f((uint16_t)(w[i] >> 16));
        }
      }
    }
    for (; row < row1; row += R) {
      const u4 v = base[(size_t)row * ld];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += T::to_f((uint16_t)(w[i] & 0xffff));
        acc[2 * i + 1] += T::to_f((uint16_t)(w[i] >> 16));
      }
    }
    float* dst = red + (size_t)r * cols + cg * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = acc[i];
  }
  __syncthreads();
  for (int c = tid; c < cols; c += 256) {
    float s = 0.f;
    for (int j = 0; j < R; ++j) s += red[(size_t)j * cols + c];
    part[(size_t)blockIdx.x * pcols + c] = s;
  }
}

// stage 2 / generic fp32 column sum: out[c] = sum_r x[r][c] (fixed order).  A block owns 16 columns
// (64-B row segments), 64 row-lanes per column; the partials are L2-resident, so what matters is
// having every load of a thread in flight at once.
// (blocks >= nb1 work on the second matrix x2 / out2 / cols2: two column sums over the same rows in ONE launch)
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int rows, int cols, int nb1, const float* __restrict__ x2,
                                                           float* __restrict__ out2, int cols2) {
  __shared__ float red[64][17];
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  int blk = blockIdx.x;
  if (blk >= nb1) { blk -= nb1; x = x2; out = out2; cols = cols2; }       // (uniform per block)
  const int c = blk * 16 + cl;
  float s = 0.f;
  if (c < cols) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rl;
    for (; r + 192 < rows; r += 256) {
      s0 += x[(size_t)r * cols + c];
      s1 += x[(size_t)(r + 64) * cols + c];
      s2 += x[(size_t)(r + 128) * cols + c];
      s3 += x[(size_t)(r + 192) * cols + c];
    }
    for (; r < rows; r += 64) s0 += x[(size_t)r * cols + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (tid < 16 && blk * 16 + tid < cols) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) t += red[j][tid];
    out[blk * 16 + tid] = t;
  }
}

// out[bh][j] = scale * (a[bh][j] + sum_s p[bh][s][j]),  j < n  (float4 granularity)
__global__ __launch_bounds__(256) void slice_sum_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                        float* __restrict__ out, int S, int n4, float scale, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const size_t bh = i / n4, j = i - bh * n4;
  float4 acc = a ? reinterpret_cast<const float4*>(a)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < S; s0 += 8) {              // eight slices in flight; summation order unchanged
    float4 v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = reinterpret_cast<const float4*>(p)[(bh * S + min(s0 + u, S - 1)) * n4 + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < S) { acc.x += v8[u].x; acc.y += v8[u].y; acc.z += v8[u].z; acc.w += v8[u].w; }
    }
  }
  reinterpret_cast<float4*>(out)[i] = make_float4(scale * acc.x, scale * acc.y, scale * acc.z, scale * acc.w);
}

constexpr int CS_MAXW = 2048;     // columns per launch (256 threads x 8); wider matrices go in windows

int colsum_parts(int rows, int cols) {
  if (rows <= 0 || cols <= 0 || (cols & 7) || cols > 16384) return EA_E_BADARG;
  const int R = 256 / ((cols < CS_MAXW ? cols : CS_MAXW) >> 3);
  // ~512 slabs (two per CU; measured best on MI355X: 3.3 TB/s cold at 100352 x 576), each at least
  // two unrolled sweeps deep
  int rpb = (rows + 511) / 512;
  const int min_rpb = R * CS_UNROLL * 2;
  if (rpb < min_rpb) rpb = min_rpb;
  return (rows + rpb - 1) / rpb;
}

int colsum_dispatch(int dtype, const void* x, float* part, float* out, int rows, int cols, hipStream_t st) {
  const int nblk = colsum_parts(rows, cols);
  if (nblk < 0) return nblk;
  const int rpb = (rows + nblk - 1) / nblk;
  for (int c0 = 0; c0 < cols; c0 += CS_MAXW) {
    const int w = cols - c0 < CS_MAXW ? cols - c0 : CS_MAXW;
    const int Rw = 256 / (w >> 3);
    const size_t lds = (size_t)Rw * w * sizeof(float);
    const uint16_t* xw = (const uint16_t*)x + c0;
    if (dtype == EA_BF16)
      hipLaunchKernelGGL(colsum_part_kernel<BF16>, dim3(nblk), dim3(256), lds, st, xw, part + c0, rows, w, rpb, cols, cols);
    else if (dtype == EA_F16)
      hipLaunchKernelGGL(colsum_part_kernel<F16>, dim3(nblk), dim3(256), lds, st, xw, part + c0, rows, w, rpb, cols, cols);
    else
      return EA_E_BADARG;
  }
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols + 15) / 16), dim3(1024), 0, st, part, out, nblk, cols, (cols + 15) / 16,
                     (const float*)nullptr, (float*)nullptr, 0);
  return (int)hipGetLastError();
}

int colsum_f32_dispatch(const float* x, float* out, int rows, int cols, const float* x2, float* out2, int cols2, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || cols2 < 0) return EA_E_BADARG;
  const int nb1 = (cols + 15) / 16, nb2 = x2 ? (cols2 + 15) / 16 : 0;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(nb1 + nb2), dim3(1024), 0, st, x, out, rows, cols, nb1, x2, out2, cols2);
  return (int)hipGetLastError();
}

int slice_sum_dispatch(const float* a, const float* p, float* out, int BH, int S, int n, float scale, hipStream_t st) {
  if (BH <= 0 || S <= 0 || n <= 0 || (n & 3)) return EA_E_BADARG;
  const size_t total4 = (size_t)BH * (n / 4);
  hipLaunchKernelGGL(slice_sum_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, a, p, out, S, n / 4, scale, total4);
  return (int)hipGetLastError();
}

// Gradient of a table gather (relative-position bias table -> dense per-head bias, local_attention.py:70-79):
//     out[row][c] = sum_k g[inv[row][k]][c]     (inv: the positions that read table row `row`, -1 = none)
// fixed order.
__global__ __launch_bounds__(64) void gather_sum_kernel(const float* __restrict__ g, const int* __restrict__ inv,
                                                         float* __restrict__ out, int rows, int K, int cols) {
  // one wave per table row: lane k holds position k (k + 64, ...) of the row's list, all loads in flight at once; the
  // lanes are added by a fixed butterfly
  const int row = blockIdx.x, lane = threadIdx.x;
  for (int c = 0; c < cols; ++c) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
      const int j = inv[(size_t)row * K + k];
      s += j >= 0 ? g[(size_t)j * cols + c] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) out[(size_t)row * cols + c] = s;
  }
}

int gather_sum_dispatch(const float* g, const int* inv, float* out, int rows, int K, int cols, hipStream_t st) {
  if (rows <= 0 || K <= 0 || cols <= 0) return EA_E_BADARG;
  hipLaunchKernelGGL(gather_sum_kernel, dim3((unsigned)rows), dim3(64), 0, st, g, inv, out, rows, K, cols);
  return (int)hipGetLastError();
}

// Dense per-head bias of the window kernels straight out of its table (round 6; local_attention.py:70-79 `table[index]` ->
// permute, eva.py:15-65 `relative_attention_bias(bucket) * scale`):
//     out[hd][i][j] = scale * table[idx[i Wk + j]][hd]   (j < Wk),   0   (Wk <= j < ld)
// in the layout and units the kernels stage (rows padded to `ld`, `scale` carries log2 e): one launch for the index_select,
// permute, scalar multiply, pad (fill + copy) and contiguous copy the framework spent on it.
// th = heads of the TABLE: h (one column per head) or 1 (causal_eva.py's single-head T5 table, broadcast over the heads).
__global__ __launch_bounds__(256) void table_bias_fwd_kernel(const float* __restrict__ table, const int* __restrict__ idx,
                                                              float* __restrict__ out, int h, int th, int Wq, int Wk, int ld,
                                                              float scale) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= h * Wq * ld) return;
  const int hd = e / (Wq * ld), rem = e - hd * (Wq * ld);
  const int i = rem / ld, j = rem - i * ld;
  out[e] = j < Wk ? scale * table[(size_t)idx[i * Wk + j] * th + (th == 1 ? 0 : hd)] : 0.f;
}

// ... and its gradient from the padded gradient image g [h][Wq][ld]:
//     dtable[row][hd] = scale * sum_k g[hd][p / Wk][p % Wk],  p = inv[row][k] >= 0   (fixed order: lanes, then a butterfly)
__global__ __launch_bounds__(256) void table_bias_bwd_kernel(const float* __restrict__ g, const int* __restrict__ inv,
                                                              float* __restrict__ dtable, int K, int h, int Wq, int Wk, int ld,
                                                              float scale) {
  // block = (table row, head): 256 lanes walk the row's position list (a far T5 bucket of a 128 x 256 window holds thousands
  // of positions: one wave per row took 0.4 ms there), lane sums -> wave butterflies -> the four wave sums added in order.
  // Every block writes its own (row, head) cell of dtable [rows, h]; the caller adds the heads of a one-column table.
  __shared__ float red[4];
  const int row = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
  float s = 0.f;
  for (int k = tid; k < K; k += 256) {
    const int p = inv[(size_t)row * K + k];
    const int pc = p >= 0 ? p : 0;
    const int i = pc / Wk, j = pc - i * Wk;
    const float v = g[((size_t)hd * Wq + i) * ld + j];
    s += p >= 0 ? v : 0.f;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) dtable[(size_t)row * h + hd] = (((red[0] + red[1]) + red[2]) + red[3]) * scale;
}

int table_bias_fwd_dispatch(const float* table, const int* idx, float* out, int h, int th, int Wq, int Wk, int ld, float scale,
                            hipStream_t st) {
  if (h <= 0 || (th != h && th != 1) || Wq <= 0 || Wk <= 0 || ld < Wk) return EA_E_BADARG;
  const int n = h * Wq * ld;
  hipLaunchKernelGGL(table_bias_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, idx, out, h, th, Wq, Wk, ld, scale);
  return (int)hipGetLastError();
}

int table_bias_bwd_dispatch(const float* g, const int* inv, float* dtable, int rows, int K, int h, int Wq, int Wk, int ld,
                            float scale, hipStream_t st) {
  if (rows <= 0 || K <= 0 || h <= 0 || Wq <= 0 || Wk <= 0 || ld < Wk) return EA_E_BADARG;
  hipLaunchKernelGGL(table_bias_bwd_kernel, dim3((unsigned)rows, (unsigned)h), dim3(256), 0, st, g, inv, dtable, K, h, Wq, Wk, ld,
                     scale);
  return (int)hipGetLastError();
}

// The autocast casts of a layer's parameters in ONE launch (round 6): dst_k[i] = (16-bit) src_k[i], up to eight fp32 tensors
// (both projections' weights and biases of a 320 / 512 / 1024-wide layer, whose projections are library GEMMs on 16-bit
// operands: four `bfloat16_copy` launches of ~4-10 us each per step until now).  Round to nearest even, exactly `.to(dtype)`.
struct MultiCastP {
  const float* src[8];
  uint16_t* dst[8];
  long n[8];
  int blk0[9];
  int nseg;
};
template <typename E>
__global__ __launch_bounds__(256) void multi_cast_kernel(const MultiCastP p) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) k += (i < p.nseg && (int)blockIdx.x >= p.blk0[i]) ? 1 : 0;
  const long n = p.n[k];
  const long i0 = ((long)((int)blockIdx.x - p.blk0[k]) * 256 + threadIdx.x) * 8;
  if (i0 >= n) return;
  const float* s = p.src[k] + i0;
  uint16_t* d = p.dst[k] + i0;
  if (i0 + 8 <= n && (((uintptr_t)s & 15) == 0) && (((uintptr_t)d & 15) == 0)) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *reinterpret_cast<u32x4*>(d) = pack8<E>(f);
  } else {
    for (int e = 0; e < 8 && i0 + e < n; ++e) d[e] = (uint16_t)(pack2<E>(s[e], 0.f) & 0xffffu);
  }
}

int multi_cast_dispatch(int dtype, int K, const float* const* src, const long long* n, void* const* dst, hipStream_t st) {
  if (K <= 0 || K > 8) return EA_E_BADARG;
  MultiCastP p = {};
  int blk = 0;
  for (int k = 0; k < K; ++k) {
    if (!src[k] || !dst[k] || n[k] <= 0) return EA_E_BADARG;
    p.src[k] = src[k]; p.dst[k] = (uint16_t*)dst[k]; p.n[k] = (long)n[k];
    p.blk0[k] = blk;
    blk += (int)((n[k] + 2047) / 2048);
  }
  p.blk0[K] = blk;
  p.nseg = K;
  if (dtype == EA_BF16) hipLaunchKernelGGL(multi_cast_kernel<BF16>, dim3((unsigned)blk), dim3(256), 0, st, p);
  else if (dtype == EA_F16) hipLaunchKernelGGL(multi_cast_kernel<F16>, dim3((unsigned)blk), dim3(256), 0, st, p);
  else return EA_E_BADARG;
  return (int)hipGetLastError();
}

// Bandwidth yardstick (bench.py `hbm_measured_copy_gbs`, SURVEY 8d "babel-stream-style copy kernel on the same GPU"):
// dst[i] = src[i] in 16-byte pieces.  Measured on MI355X over 512 MB (tools/probe/copy_probe.hip): a workgroup walking
// CONTIGUOUS 16 KB pieces (four 4 KB rows of its 256 lanes in flight) with non-temporal loads / stores reaches 6.25 TB/s
// (plain: 5.7); the classic grid-stride loop whose four loads lie a whole grid (8 MB) apart 4.4 TB/s -- DRAM page locality
// of what is in flight at one time matters more than anything else in a streaming kernel; hipMemcpyDtoD 4.7 TB/s.
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t step = (size_t)gridDim.x * 1024;
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 768 < n16; i += step) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
  for (int u = 0; u < 4; ++u)
    if (i + u * 256 < n16 && i + 768 >= n16) dst[i + u * 256] = src[i + u * 256];
}

int stream_copy_dispatch(const void* src, void* dst, size_t bytes, hipStream_t st) {
  if (bytes == 0 || (bytes & 15)) return EA_E_BADARG;
  const size_t n16 = bytes / 16;
  size_t grid = (n16 + 1023) / 1024;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const u32x4*)src, (u32x4*)dst, n16);
  return (int)hipGetLastError();
}

}  // namespace ea
