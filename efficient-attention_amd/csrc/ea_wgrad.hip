// ea_wgrad.hip -- weight and bias gradient of a projection in ONE pass over the activations:
//     dW[o][i] = sum_t dY[t][o] X[t][i],   db[o] = sum_t dY[t][o]        (t over all B*N tokens)
// for the qkv / output Linear layers around the attention cores (abstract_attention.py:72-78,86-87
// differentiated).  The contraction runs over ~1e5 tokens and leaves a [out, in] result of a few
// tiles: a library GEMM either occupies a handful of CUs or, split-K batched, is limited by the
// few workgroups it gets (round 1: 45 us per projection + a 13 us bias-gradient pass re-reading dY).
//
// Decomposition: token slices x output tiles.  A workgroup owns a [BM out x 64 in] tile of dW for one
// token slice and streams its slice of dY[:, BM] and X[:, 64] through LDS in 64-token stages (double
// buffered: the next stage's global loads are in flight during the MFMAs, one barrier per stage).  Both
// MFMA operands are token-contracted, i.e. transposed reads of row-major tiles: ds_read_b64_tr_b16 on
// the XOR-swizzled [64 tokens][64 channels] sub-tiles every kernel here uses.  All tiles of a slice
// are placed on ONE XCD (block id -> XCD is round-robin), so a dY / X row is fetched from HBM once
// and re-read by the other tiles of its slice out of that XCD's L2.  The bias gradient rides along:
// the workgroups of in-tile 0 add up the dY rows they stage anyway.  Slice partials [S, out, in] fp32
// are summed in a fixed order by ea_slice_sum (deterministic; no atomics).
#include <stdlib.h>
#include "ea_common.h"

namespace ea {

struct WgP {
  const char* dy;     // [rows, M] element type
  const char* x;      // [rows, K]
  float* part;        // [S, M, K]
  float* db_part;     // [S, M] or null
  int rows, M, K;
  int S, rows_per_slice, tiles_m, tiles_n;
};

template <typename E, int BM>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgP p) {
  constexpr int SA = BM / 64;                 // 64-channel sub-tiles of the dY stage
  constexpr int NAF = SA * 2;                 // A fragments (16 out-channels each) per wave: half of SA*4
  constexpr int NA = BM / 32;                 // 16-B staging chunks of dY per thread and stage
  constexpr int CPRA = BM / 8;                // chunks per dY row
  constexpr int STAGE = (SA + 1) * 64 * 128;  // bytes of one stage: SA + 1 sub-tiles of [64][64]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  // block -> (slice, tile): every tile of a slice on the same XCD (block id modulo 8)
  const int T = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = j % T, slice = (j / T) * 8 + xcd;
  if (slice >= p.S) return;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * 64;
  const int r0 = slice * p.rows_per_slice;
  const int r1 = min(p.rows, r0 + p.rows_per_slice);
  const bool with_bias = p.db_part != nullptr && tn == 0;

  u32x4 pa[NA], pb[2];
  auto issue = [&](int rb) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int idx = tid + k * 256;
      const int row = idx / CPRA, ch = idx - row * CPRA;
      const int t = rb + row;
      pa[k] = t < r1 ? ldg16(p.dy + ((size_t)t * p.M + m0 + ch * 8) * 2) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * 256;
      const int row = idx >> 3, ch = idx & 7;
      const int t = rb + row;
      pb[k] = t < r1 ? ldg16(p.x + ((size_t)t * p.K + n0 + ch * 8) * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  float accb[NA][8];
#pragma unroll
  for (int k = 0; k < NA; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) accb[k][e] = 0.f;
  auto commit = [&](char* st) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int idx = tid + k * 256;
      const int row = idx / CPRA, ch = idx - row * CPRA;
      sts16(st + (ch >> 3) * (64 * 128) + lds_off<64>(row, ch & 7), pa[k]);
      if (with_bias) {
        float f[8];
        unpack8<E>(pa[k], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) accb[k][e] += f[e];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + k * 256;
      sts16(st + SA * (64 * 128) + lds_off<64>(idx >> 3, idx & 7), pb[k]);
    }
  };

  const int wa = wave & 1, wb = wave >> 1;
  const int wr = 4 * g + (li >> 2);
  // per-fragment LDS offsets, computed (not looked up: an array indexed by the runtime wave id lives in scratch,
  // and a scratch access shares -- and drains -- the vmcnt queue of the prefetched global loads)
  int aoff[NAF];
#pragma unroll
  for (int f = 0; f < NAF; ++f) {
    const int fa = wa * NAF + f, dt = fa & 3;
    const int colb = (16 * (li & 3) + 4 * dt) * 2;
    aoff[f] = (fa >> 2) * (64 * 128) + wr * 128 + ((((colb >> 4)) ^ (wr & 7)) << 4) + (colb & 15);
  }
  int btr[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int colb = (16 * (2 * wb + c) + 4 * (li & 3)) * 2;
    btr[c] = wr * 128 + ((((colb >> 4)) ^ (wr & 7)) << 4) + (colb & 15);
  }
  f32x4 acc[NAF][2];
#pragma unroll
  for (int f = 0; f < NAF; ++f) acc[f][0] = acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(r0);
  commit(smem);
  __syncthreads();
  int buf = 0;
  for (int rb = r0; rb < r1; rb += 64) {
    char* cur = smem + buf * STAGE;
    const bool more = rb + 64 < r1;
    if (more) issue(rb + 64);
#pragma unroll
    for (int kb = 0; kb < 64; kb += 32) {
      typename E::x8 bf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const char* bp = cur + SA * (64 * 128) + kb * 128 + btr[c];
        bf[c] = as_x8<E>(E::tr4(bp), E::tr4(bp + 16 * 128));
      }
#pragma unroll
      for (int f = 0; f < NAF; ++f) {
        const char* ap = cur + kb * 128 + aoff[f];       // fragment wa * NAF + f: sub-tile fa / 4, channel group fa % 4
        const typename E::x8 af = as_x8<E>(E::tr4(ap), E::tr4(ap + 16 * 128));
        acc[f][0] = E::mma(af, bf[0], acc[f][0]);
        acc[f][1] = E::mma(af, bf[1], acc[f][1]);
      }
    }
    if (more) commit(smem + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }
  // ---- partial tile -> part[slice][m][n]: D row 4g+r of fragment fa <-> out channel 64 sa + 16 g + 4 dt + r ----
  float* out = p.part + (size_t)slice * p.M * p.K;
#pragma unroll
  for (int f = 0; f < NAF; ++f) {
    const int fa = wa * NAF + f, sa = fa >> 2, dt = fa & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = n0 + 16 * (2 * wb + c) + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 64 * sa + 16 * g + 4 * dt + r;
        out[(size_t)m * p.K + n] = acc[f][c][r];
      }
    }
  }
  if (!with_bias) return;
  // ---- bias partial: 64 staged rows per column group, summed through LDS in a fixed order ----
  float* red = reinterpret_cast<float*>(smem);            // [NA * 256][8] (the stage buffers are free)
#pragma unroll
  for (int k = 0; k < NA; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(size_t)(tid + k * 256) * 8 + e] = accb[k][e];
  __syncthreads();
  if (tid < BM) {
    const int ch = tid >> 3, e = tid & 7;
    float s = 0.f;
    for (int row = 0; row < 64; ++row) s += red[(size_t)(row * CPRA + ch) * 8 + e];
    p.db_part[(size_t)slice * p.M + m0 + tid] = s;
  }
}

static int wg_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

static int wg_bm(int M) { return M % 192 == 0 ? 192 : (M % 128 == 0 ? 128 : 64); }

// token slices: about one workgroup per CU in total, a multiple of 8 (one XCD each), >= 256 tokens each
int wgrad_slices(int rows, int M, int K) {
  if (rows <= 0 || M <= 0 || K <= 0 || (M & 63) || (K & 63)) return EA_E_UNSUPPORTED;
  const int T = (M / wg_bm(M)) * (K / 64);
  static const int per_cu = getenv("EA_WGRAD_PER_CU") ? atoi(getenv("EA_WGRAD_PER_CU")) : 1;
  int S = (per_cu * wg_cus() + T / 2) / T;
  S = (S + 4) / 8 * 8;
  if (S < 8) S = 8;
  while (S > 8 && rows / S < 256) S -= 8;
  if (S > 128) S = 128;
  return S;
}

template <typename E, int BM>
static int launch_wg(const WgP& p, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM / 64 + 1) * 64 * 128;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<E, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int T = p.tiles_m * p.tiles_n;
  const dim3 grid((unsigned)(((p.S + 7) / 8) * 8 * T)), block(256);
  hipLaunchKernelGGL((wgrad_kernel<E, BM>), grid, block, lds, st, p);
  return (int)hipGetLastError();
}

int wgrad_dispatch(int dtype, const void* dy, const void* x, float* part, float* db_part, int rows, int M, int K,
                   hipStream_t st) {
  const int S = wgrad_slices(rows, M, K);
  if (S < 0) return S;
  WgP p;
  p.dy = (const char*)dy; p.x = (const char*)x; p.part = part; p.db_part = db_part;
  p.rows = rows; p.M = M; p.K = K; p.S = S;
  p.rows_per_slice = ((rows + S - 1) / S + 63) / 64 * 64;
  const int bm = wg_bm(M);
  p.tiles_m = M / bm; p.tiles_n = K / 64;
#define EA_WG(E)                                                        \
  do {                                                                  \
    if (bm == 192) return launch_wg<E, 192>(p, st);                     \
    if (bm == 128) return launch_wg<E, 128>(p, st);                     \
    return launch_wg<E, 64>(p, st);                                     \
  } while (0)
  if (dtype == EA_BF16) EA_WG(BF16);
  if (dtype == EA_F16) EA_WG(F16);
#undef EA_WG
  return EA_E_BADARG;
}

}  // namespace ea
