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
#include <algorithm>
#include "ea_common.h"

namespace ea {

struct WgP {
  const char* dy;     // [rows, M] element type
  const char* x;      // [rows, K]
  float* part;        // slice s: [M, K] at part + s * part_ld
  float* db_part;     // slice s: [M] at db_part + s * part_ld, or null
  long part_ld;
  int rows, M, K;
  int S, rows_per_slice, tiles_m, tiles_n;
  // a SECOND product over the same token rows in the same launch (ea_wgrad_pair, round 4): its tiles follow the first one's
  // in the tile index.  T2 = 0: none.
  const char* dy2;
  const char* x2;
  float* part2;
  float* db_part2;
  long part_ld2;
  int M2, K2, tiles_n2, T2;
};

template <typename E, int BM, int BN>
__global__ __launch_bounds__(512, 1) void wgrad_kernel(const WgP p) {
  constexpr int SA = BM / 64, SB = BN / 64;   // 64-channel sub-tiles of the dY / X stage
  constexpr int FA = BM / 64;                 // A fragments (16 out-channels) per wave: BM / 4 waves / 16
  constexpr int FB = BN / 32;                 // B fragments (16 in-channels) per wave:  BN / 2 waves / 16
  constexpr int CPRA = BM / 8, CPRB = BN / 8; // 16-B chunks per staged row
  constexpr int STAGE = (SA + SB) * 64 * 128; // bytes of one stage: sub-tiles of [64 tokens][64 channels]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  // block -> (slice, tile): every tile of a slice on the same XCD (block id modulo 8)
  const int T1 = p.tiles_m * p.tiles_n, T = T1 + p.T2;
  int tile, slice;
  if ((p.S & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = j % T; slice = (j / T) * 8 + xcd;
  } else {                                     // slice count chosen to fill the CUs (wgrad_slices): tiles of a slice side by side
    tile = blockIdx.x % T; slice = blockIdx.x / T;
  }
  if (slice >= p.S) return;
  // which product this workgroup belongs to (uniform): everything below reads these copies
  const bool second = tile >= T1;
  const char* const dyp = second ? p.dy2 : p.dy;
  const char* const xp = second ? p.x2 : p.x;
  float* const partp = second ? p.part2 : p.part;
  float* const dbp = second ? p.db_part2 : p.db_part;
  const long part_ld = second ? p.part_ld2 : p.part_ld;
  const int pM = second ? p.M2 : p.M, pK = second ? p.K2 : p.K, tiles_n = second ? p.tiles_n2 : p.tiles_n;
  if (second) tile -= T1;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int r0 = slice * p.rows_per_slice;
  const int r1 = min(p.rows, r0 + p.rows_per_slice);
  const bool with_bias = dbp != nullptr && tn == 0;

  // Round 5 (from the loop's disassembly: 161 of its 333 VALU instructions were v_mov -- every `t < r1 ? load : 0` built a
  // 64-bit address pair per load and selected around it): the loads are unconditional, from a 32-bit byte offset per staged
  // chunk that advances by 64 rows per stage (base pointer in SGPRs); only the LAST stage of a slice can reach past it, and
  // only there the rows are clamped (issue) and zeroed (commit) -- a uniform branch.
  u32x4 pa[SA], pb[SB];
  uint32_t offa[SA], offb[SB];
#pragma unroll
  for (int k = 0; k < SA; ++k) {
    const int idx = tid + k * 512;
    const int row = idx / CPRA, ch = idx - row * CPRA;
    offa[k] = (uint32_t)(((size_t)row * pM + m0 + ch * 8) * 2);   // relative to the slice's first row (dys / xs below)
  }
#pragma unroll
  for (int k = 0; k < SB; ++k) {
    const int idx = tid + k * 512;
    const int row = idx / CPRB, ch = idx - row * CPRB;
    offb[k] = (uint32_t)(((size_t)row * pK + n0 + ch * 8) * 2);
  }
  const uint32_t stepa = (uint32_t)(64 * pM * 2), stepb = (uint32_t)(64 * pK * 2);
  const char* const dys = dyp + (size_t)r0 * pM * 2;       // (a slice spans < 4 GB: checked at dispatch)
  const char* const xs = xp + (size_t)r0 * pK * 2;
  auto issue = [&](int rb) {
    if (rb + 64 <= r1) {
#pragma unroll
      for (int k = 0; k < SA; ++k) pa[k] = ldg16(dys + offa[k]);
#pragma unroll
      for (int k = 0; k < SB; ++k) pb[k] = ldg16(xs + offb[k]);
    } else {
#pragma unroll
      for (int k = 0; k < SA; ++k) {
        const int idx = tid + k * 512;
        const int row = idx / CPRA, ch = idx - row * CPRA;
        const int t = min(rb + row, r1 - 1);
        pa[k] = ldg16(dyp + ((size_t)t * pM + m0 + ch * 8) * 2);
      }
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int idx = tid + k * 512;
        const int row = idx / CPRB, ch = idx - row * CPRB;
        const int t = min(rb + row, r1 - 1);
        pb[k] = ldg16(xp + ((size_t)t * pK + n0 + ch * 8) * 2);
      }
    }
#pragma unroll
    for (int k = 0; k < SA; ++k) offa[k] += stepa;
#pragma unroll
    for (int k = 0; k < SB; ++k) offb[k] += stepb;
  };
  // bias partials: a thread's staged chunks all sit in ONE 8-channel group when 512 is a multiple of the chunks per row
  // (BM = 64, 128, 256), so they share one set of accumulators
  constexpr int NB = (512 % CPRA == 0) ? 1 : SA;
  float accb[NB][8];
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) accb[k][e] = 0.f;
  auto commit = [&](char* st, int rb) {
    if (rb + 64 > r1) {                                // (uniform) the slice's last stage: rows past it count as zeros
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int k = 0; k < SA; ++k) { if (rb + (tid + k * 512) / CPRA >= r1) pa[k] = z; }
#pragma unroll
      for (int k = 0; k < SB; ++k) { if (rb + (tid + k * 512) / CPRB >= r1) pb[k] = z; }
    }
#pragma unroll
    for (int k = 0; k < SA; ++k) {
      const int idx = tid + k * 512;
      const int row = idx / CPRA, ch = idx - row * CPRA;
      sts16(st + (ch >> 3) * (64 * 128) + lds_off2<64>(row, ch & 7), pa[k]);
      if (with_bias) {
        float f[8];
        unpack8<E>(pa[k], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) accb[NB == 1 ? 0 : k][e] += f[e];
      }
    }
#pragma unroll
    for (int k = 0; k < SB; ++k) {
      const int idx = tid + k * 512;
      const int row = idx / CPRB, ch = idx - row * CPRB;
      sts16(st + (SA + (ch >> 3)) * (64 * 128) + lds_off2<64>(row, ch & 7), pb[k]);
    }
  };

  // 8 waves as 4 (out channels) x 2 (in channels): wave (wa, wb) owns [BM / 4] x [BN / 2] of the tile
  const int wa = wave & 3, wb = wave >> 2;
  const int wr = 4 * g + (li >> 2);
  // per-fragment LDS offsets, computed (not looked up: an array indexed by the runtime wave id lives in scratch,
  // and a scratch access shares -- and drains -- the vmcnt queue of the prefetched global loads)
  // Both fragments are transposed reads in which the four lanes of a token row take 32 CONTIGUOUS bytes (16 channels) of
  // the phi2-swizzled row (ea_common.h, round 3): the round-2 dY pattern -- 8-byte pieces 32 bytes apart -- was 4-way
  // bank-conflicted and made the kernel LDS-bound.
  int aoff[FA], boff[FB];
  const int q = li & 3;
#pragma unroll
  for (int f = 0; f < FA; ++f) {
    const int fa = wa * FA + f;                        // 16-channel group fa of the dY stage: sub-tile fa / 4
    aoff[f] = (fa >> 2) * (64 * 128) + lds_off2<64>(wr, 2 * (fa & 3) + (q >> 1)) + 8 * (q & 1);
  }
#pragma unroll
  for (int c = 0; c < FB; ++c) {
    const int fb = wb * FB + c;                        // 16-channel group fb of the X stage
    boff[c] = (SA + (fb >> 2)) * (64 * 128) + lds_off2<64>(wr, 2 * (fb & 3) + (q >> 1)) + 8 * (q & 1);
  }
  f32x4 acc[FA][FB];
#pragma unroll
  for (int f = 0; f < FA; ++f)
#pragma unroll
    for (int c = 0; c < FB; ++c) acc[f][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(r0);
  commit(smem, r0);
  __syncthreads();
  int buf = 0;
  for (int rb = r0; rb < r1; rb += 64) {
    char* cur = smem + buf * STAGE;
    const bool more = rb + 64 < r1;
    if (more) issue(rb + 64);
#pragma unroll
    for (int kb = 0; kb < 64; kb += 32) {
      typename E::x8 af[FA];
#pragma unroll
      for (int f = 0; f < FA; ++f) {
        const char* ap = cur + kb * 128 + aoff[f];
        af[f] = as_x8<E>(E::tr4(ap), E::tr4(ap + 16 * 128));
      }
#pragma unroll
      for (int c = 0; c < FB; ++c) {
        const char* bp = cur + kb * 128 + boff[c];
        const typename E::x8 bf = as_x8<E>(E::tr4(bp), E::tr4(bp + 16 * 128));
#pragma unroll
        for (int f = 0; f < FA; ++f) acc[f][c] = E::mma(af[f], bf, acc[f][c]);
      }
    }
    if (more) commit(smem + (buf ^ 1) * STAGE, rb + 64);
    __syncthreads();
    buf ^= 1;
  }
  // ---- partial tile -> part[slice][m][n], through LDS so that it leaves in 16-byte row segments (D row 4g+r of
  // fragment fa <-> out channel 64 sa + 16 dt + 4 g + r, D column li of fragment fb <-> in channel 16 fb + li): 72
  // dword stores per lane straight from the accumulators would cost more than the whole stream.  The stage buffers
  // are free; one half of the in-channels (the waves of one wb) at a time fits them.
  float* out = partp + (size_t)slice * part_ld;
  float* ot = reinterpret_cast<float*>(smem);             // [BM][BN / 2]
  constexpr int HN = BN / 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wb == h) {
#pragma unroll
      for (int f = 0; f < FA; ++f) {
        const int fa = wa * FA + f, sa = fa >> 2, dt = fa & 3;
#pragma unroll
        for (int c = 0; c < FB; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[(64 * sa + 16 * dt + 4 * g + r) * HN + 16 * c + li] = acc[f][c][r];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * (HN / 4); idx += 512) {
      const int m = idx / (HN / 4), c4 = idx - m * (HN / 4);
      *reinterpret_cast<f32x4*>(out + (size_t)(m0 + m) * pK + n0 + h * HN + c4 * 4) =
          *reinterpret_cast<const f32x4*>(ot + m * HN + c4 * 4);
    }
    __syncthreads();
  }
  if (!with_bias) return;
  // ---- bias partial: 64 staged rows per column group, summed through LDS in a fixed order ----
  float* red = reinterpret_cast<float*>(smem);            // [NB * 512][8] (the stage buffers are free)
#pragma unroll
  for (int k = 0; k < NB; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(size_t)(tid + k * 512) * 8 + e] = accb[k][e];
  __syncthreads();
  if (tid < BM) {
    const int ch = tid >> 3, e = tid & 7;
    float s = 0.f;
    for (int row = 0; row < NB * 512 / CPRA; ++row) s += red[(size_t)(row * CPRA + ch) * 8 + e];
    dbp[(size_t)slice * part_ld + m0 + tid] = s;
  }
}

// out[j] = sum_s part[s * ld + j], j < n (n, ld multiples of 4): the slice partials of a weight (+ bias) gradient, added
// in a fixed order.  A block owns 32 float4 columns; its 8 slice lanes each add every 8th slice (four independent
// loads in flight per thread), then lane 0 adds the 8 lane sums in order.
__global__ __launch_bounds__(256) void part_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int S, int n4,
                                                       long ld4) {
  __shared__ f32x4 red[8][32];
  const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  const f32x4* src = reinterpret_cast<const f32x4*>(part) + min(j, n4 - 1);
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int s = sl;
  for (; s + 24 < S; s += 32) {
    const f32x4 v0 = src[(size_t)s * ld4], v1 = src[(size_t)(s + 8) * ld4];
    const f32x4 v2 = src[(size_t)(s + 16) * ld4], v3 = src[(size_t)(s + 24) * ld4];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < S; s += 8) a0 += src[(size_t)s * ld4];
  red[sl][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && j < n4) {
    f32x4 t = red[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][c];
    reinterpret_cast<f32x4*>(out)[j] = t;
  }
}

// Several such reductions in ONE launch (round 4): the terminal sums of a layer's backward -- the slice partials of both
// projections' weight gradients and the per-(b,h) partials of the landmark parameters -- each cost a 6-10 us launch of their
// own.  Segment k: out_k[j] = sum_s part_k[s * ld_k + j]; blocks [blk0_k, blk0_{k+1}) work on it, in part_sum_kernel's order
// of additions for segments of up to 96 slices (a fixed order of their own for the wide ones).
struct MultiSumP {
  const float* part[6];
  float* out[6];
  int S[6], n4[6], blk0[7];
  long ld4[6];
  int nseg;
  int wide;          // bit k: segment k has many slices (the per-(b,h) partials of the landmark parameters, S = B*h): a block
                     // owns 8 float4 columns with 32 slice lanes instead of 32 columns with 8 -- a thread of the narrow
                     // layout walked 48 slices in 12 dependent rounds of loads, the launch's whole duration (round 5)
};
__global__ __launch_bounds__(256) void multi_sum_kernel(const MultiSumP p) {
  __shared__ f32x4 red[256];
  int k = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i) k += (i < p.nseg && (int)blockIdx.x >= p.blk0[i]) ? 1 : 0;
  const bool wide = (p.wide >> k) & 1;
  const int cols = wide ? 8 : 32, lanes = wide ? 32 : 8;
  const int c = wide ? (threadIdx.x & 7) : (threadIdx.x & 31), sl = wide ? (threadIdx.x >> 3) : (threadIdx.x >> 5);
  const int j = ((int)blockIdx.x - p.blk0[k]) * cols + c;
  const int S = p.S[k], n4 = p.n4[k];
  const long ld4 = p.ld4[k];
  const f32x4* src = reinterpret_cast<const f32x4*>(p.part[k]) + min(j, n4 - 1);
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int s = sl;
  for (; s + 3 * lanes < S; s += 4 * lanes) {
    const f32x4 v0 = src[(size_t)s * ld4], v1 = src[(size_t)(s + lanes) * ld4];
    const f32x4 v2 = src[(size_t)(s + 2 * lanes) * ld4], v3 = src[(size_t)(s + 3 * lanes) * ld4];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < S; s += lanes) a0 += src[(size_t)s * ld4];
  red[sl * cols + c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && j < n4) {
    f32x4 t = red[c];
    for (int i = 1; i < lanes; ++i) t += red[i * cols + c];
    reinterpret_cast<f32x4*>(p.out[k])[j] = t;
  }
}

int multi_sum_dispatch(int K, const float* const* part, const int* S, const int* n, const long long* ld, float* const* out,
                       hipStream_t st) {
  if (K <= 0 || K > 6) return EA_E_BADARG;
  MultiSumP p = {};
  int blk = 0;
  for (int k = 0; k < K; ++k) {
    if (!part[k] || !out[k] || S[k] <= 0 || n[k] <= 0 || (n[k] & 3) || (ld[k] & 3) || ld[k] < n[k] ||
        ((uintptr_t)part[k] & 15) || ((uintptr_t)out[k] & 15)) return EA_E_BADARG;
    p.part[k] = part[k]; p.out[k] = out[k]; p.S[k] = S[k]; p.n4[k] = n[k] / 4; p.ld4[k] = (long)(ld[k] / 4);
    p.blk0[k] = blk;
    const bool wide = S[k] > 96;
    if (wide) p.wide |= 1 << k;
    blk += (p.n4[k] + (wide ? 7 : 31)) / (wide ? 8 : 32);
  }
  p.blk0[K] = blk;
  p.nseg = K;
  hipLaunchKernelGGL(multi_sum_kernel, dim3((unsigned)blk), dim3(256), 0, st, p);
  return (int)hipGetLastError();
}

int part_sum_dispatch(const float* part, float* out, int S, int n, long ld, hipStream_t st) {
  if (S <= 0 || n <= 0 || (n & 3) || (ld & 3) || ld < n) return EA_E_BADARG;
  const int n4 = n / 4;
  hipLaunchKernelGGL(part_sum_kernel, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, st, part, out, S, n4, ld / 4);
  return (int)hipGetLastError();
}

static int wg_cus() { return ea_device_cus(); }

// tile edge along one axis.  The kernel is fed from L2 at ~10 B/clk/CU, so its MFMA rate is set by the FLOPs per staged
// byte = (BM * BN) / (BM + BN): a [256 x 256] tile (wide models: 512 / 1024 / 1536 / 3072 channels) does twice the work
// per byte of a [128 x 128] one (measured 378 TFLOP/s with 128 x 128 tiles at 1536 x 512 x 65536).
static int wg_tile_max() {
  static const int v = [] { const char* e = getenv("EA_WGRAD_TILE_MAX"); return e ? atoi(e) : 256; }();   // dev knob
  return v;
}
static int wg_bt(int M) {
  const int mx = wg_tile_max();
  return (M % 256 == 0 && mx >= 256) ? 256 : ((M % 192 == 0 && mx >= 192) ? 192 : ((M % 128 == 0 && mx >= 128) ? 128 : 64));
}

// token slices: one workgroup per CU (96 KB of LDS each), the tiles of a slice on one XCD (32 CUs), >= 256 tokens each
int wgrad_slices(int rows, int M, int K) {
  if (rows <= 0 || M <= 0 || K <= 0 || (M & 63) || (K & 63)) return EA_E_UNSUPPORTED;
  const int T = (M / wg_bt(M)) * (K / wg_bt(K));
  const int per_xcd = wg_cus() / 8;
  int S = per_xcd / T * 8;
  if (S < 8) S = 8;
  while (S > 8 && rows / S < 256) S -= 8;
  // many tiles (wide models): a multiple of 8 slices can leave a third of the CUs idle (20 tiles x 8 slices = 160
  // workgroups).  The kernel is bound by what ONE CU can pull out of L2 (~12.5 B/clk measured), so filling the CUs beats
  // keeping a slice's tiles on one XCD -- the re-reads of a dY / X row by the other XCDs hit the Infinity Cache.
  if (S * T < wg_cus() * 85 / 100 && T <= wg_cus()) {
    int Sf = wg_cus() / T;
    while (Sf > 1 && rows / Sf < 256) --Sf;
    if (Sf * T > S * T && (Sf & 7) != 0) S = Sf;
  }
  // (the kernel addresses a slice's rows by 32-bit byte offsets: more slices before a slice reaches 4 GB)
  while ((long)((rows + S - 1) / S + 64) * std::max(M, K) * 2 >= (1L << 32)) S += 8;
  return S;
}

template <typename E, int BM, int BN>
static int launch_wg(const WgP& p, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM / 64 + BN / 64) * 64 * 128;
  if (lds > 64 * 1024) EA_SET_LDS_ONCE((&wgrad_kernel<E, BM, BN>), lds);
  const int T = p.tiles_m * p.tiles_n + p.T2;
  // (the staging addresses are 32-bit byte offsets from a slice's first row)
  const long widest = std::max(std::max(p.M, p.K), std::max(p.M2, p.K2));
  if ((long)(p.rows_per_slice + 64) * widest * 2 >= (1L << 32)) return EA_E_UNSUPPORTED;
  const dim3 grid((unsigned)(((p.S & 7) == 0 ? ((p.S + 7) / 8) * 8 : p.S) * T)), block(512);
  hipLaunchKernelGGL((wgrad_kernel<E, BM, BN>), grid, block, lds, st, p);
  return (int)hipGetLastError();
}

template <typename E, int BM>
static int launch_wg_n(const WgP& p, int bn, hipStream_t st) {
  if (bn == 256) return launch_wg<E, BM, 256>(p, st);
  if (bn == 192) return launch_wg<E, BM, 192>(p, st);
  if (bn == 128) return launch_wg<E, BM, 128>(p, st);
  return launch_wg<E, BM, 64>(p, st);
}

// Two products over the SAME token rows in one launch (the weight gradients of a layer's qkv and output projections): with the
// tiles of both in the tile index a slice is four [192 x 192] tiles instead of three and one, i.e. 64 slices of 1568 tokens for
// the pair instead of 80 + 256 slices -- half the partial-sum traffic (38 MB written and read back instead of 73 MB at cfg3)
// and one launch's fixed cost less.  Both must take the same tile edges.
int wgrad_pair_slices(int rows, int M1, int K1, int M2, int K2) {
  if (rows <= 0 || M1 <= 0 || K1 <= 0 || M2 <= 0 || K2 <= 0 || ((M1 | K1 | M2 | K2) & 63)) return EA_E_UNSUPPORTED;
  if (wg_bt(M1) != wg_bt(M2) || wg_bt(K1) != wg_bt(K2)) return EA_E_UNSUPPORTED;
  const int T = (M1 / wg_bt(M1)) * (K1 / wg_bt(K1)) + (M2 / wg_bt(M2)) * (K2 / wg_bt(K2));
  const int per_xcd = wg_cus() / 8;
  int S = per_xcd / T * 8;
  if (S < 8) S = 8;
  while (S > 8 && rows / S < 256) S -= 8;
  while ((long)((rows + S - 1) / S + 64) * std::max(std::max(M1, K1), std::max(M2, K2)) * 2 >= (1L << 32)) S += 8;
  return S;
}

int wgrad_pair_dispatch(int dtype, int rows, const void* dy1, const void* x1, float* part1, float* db1, long ld1, int M1, int K1,
                        const void* dy2, const void* x2, float* part2, float* db2, long ld2, int M2, int K2, hipStream_t st) {
  const int S = wgrad_pair_slices(rows, M1, K1, M2, K2);
  if (S < 0) return S;
  WgP p = {};
  p.dy = (const char*)dy1; p.x = (const char*)x1; p.part = part1; p.db_part = db1; p.part_ld = ld1;
  p.rows = rows; p.M = M1; p.K = K1; p.S = S;
  p.rows_per_slice = (rows + S - 1) / S;
  const int bm = wg_bt(M1), bn = wg_bt(K1);
  p.tiles_m = M1 / bm; p.tiles_n = K1 / bn;
  p.dy2 = (const char*)dy2; p.x2 = (const char*)x2; p.part2 = part2; p.db_part2 = db2; p.part_ld2 = ld2;
  p.M2 = M2; p.K2 = K2; p.tiles_n2 = K2 / bn; p.T2 = (M2 / bm) * (K2 / bn);
#define EA_WG(E)                                                        \
  do {                                                                  \
    if (bm == 256) return launch_wg_n<E, 256>(p, bn, st);               \
    if (bm == 192) return launch_wg_n<E, 192>(p, bn, st);               \
    if (bm == 128) return launch_wg_n<E, 128>(p, bn, st);               \
    return launch_wg_n<E, 64>(p, bn, st);                               \
  } while (0)
  if (dtype == EA_BF16) EA_WG(BF16);
  if (dtype == EA_F16) EA_WG(F16);
#undef EA_WG
  return EA_E_BADARG;
}

int wgrad_dispatch(int dtype, const void* dy, const void* x, float* part, float* db_part, long part_ld, int rows, int M,
                   int K, hipStream_t st) {
  const int S = wgrad_slices(rows, M, K);
  if (S < 0) return S;
  WgP p = {};
  p.dy = (const char*)dy; p.x = (const char*)x; p.part = part; p.db_part = db_part;
  p.rows = rows; p.M = M; p.K = K; p.S = S;
  p.part_ld = part_ld;
  p.rows_per_slice = (rows + S - 1) / S;
  const int bm = wg_bt(M), bn = wg_bt(K);
  p.tiles_m = M / bm; p.tiles_n = K / bn;
#define EA_WG(E)                                                        \
  do {                                                                  \
    if (bm == 256) return launch_wg_n<E, 256>(p, bn, st);               \
    if (bm == 192) return launch_wg_n<E, 192>(p, bn, st);               \
    if (bm == 128) return launch_wg_n<E, 128>(p, bn, st);               \
    return launch_wg_n<E, 64>(p, bn, st);                               \
  } while (0)
  if (dtype == EA_BF16) EA_WG(BF16);
  if (dtype == EA_F16) EA_WG(F16);
#undef EA_WG
  return EA_E_BADARG;
}

}  // namespace ea
