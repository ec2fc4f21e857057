// ea_linear.hip -- the projections around the attention cores as streaming kernels:
//     Y[t][o] = sum_k A[t][k] W[o][k] (+ bias[o])          (t over all B*N tokens)
// `qkv = self.qkv(x)` / `x = self.proj(x)` (abstract_attention.py:72-78,86-87) and the input gradient of the
// output projection (the same product with the transposed weight).  K and `out` are a few hundred, the
// token axis ~1e5: the activations stream through once and the weight is tiny -- an HBM-bound kernel, not a
// GEMM to be tiled for operand re-use.
//
// The output columns are cut into `nsplit` parts of <= 72 KB of weight rows; a workgroup stages ITS part
// into LDS once and stays resident (two 4-wave workgroups per CU; the 576 columns of a 192-wide qkv
// projection go as two parts of 288 = 111 KB on one 8-wave workgroup per CU).  From there on there is no barrier: every wave
// walks its own 16 * RT-row token tiles (tile = wave id, + number of waves, ...), keeping the tile's rows, all
// K channels, in registers as MFMA operands -- loaded straight from global memory in the operand layout
// (lane = token li, channels 8g .. 8g+7 of a 32-channel step), the next tile's loads in flight under the
// current tile's MFMAs, converted on arrival when the activations come in fp32 (the autocast cast of x is
// folded into the load; the rounded copy is written out once for the weight-gradient GEMM).  The product is
// formed transposed, D[out][token], with the weight rows of two MFMA tiles interleaved in fours so that a
// lane ends up with EIGHT consecutive output columns of one token: 16-byte stores.  Workgroup w of every
// part sits on the same XCD (block id modulo 8) and walks the same token tiles at the same time, so the
// activations leave HBM once.
#include <stdlib.h>
#include "ea_common.h"

namespace ea {

struct LinP {
  const char* a;        // [rows, K] element type or fp32, row stride lda elements
  const char* w;        // [NO, K] element type, contiguous
  const float* bias;    // [NO] fp32 or null (rounded to the element type before it is added, as the library does)
  char* y;              // [rows, NO] element type or fp32, row stride ldy elements
  char* a_cast;         // [rows, K] element type copy of a (fp32 input only) or null
  int rows, NO, nsplit, wg_per_part, y_f32;
  int w_mode;           // 0: w is [NO, K] in the element type; 1: fp32 [NO, K]; 2: fp32 [K, NO] (the transposed product)
  long lda, ldy;
};

template <typename E, int KT, int RT, bool AF32, int NPR, bool YF32>
__global__ __launch_bounds__((NPR > 8 ? 512 : 256), (NPR > 8 ? 1 : 2)) void lin_kernel(const LinP p) {
  constexpr int NT = NPR > 8 ? 512 : 256, NW = NT / 64;   // more than 256 weight rows: one 8-wave workgroup per CU
  constexpr int K = KT * 32, ROWB = K * 2, CPR = K / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int part = j % p.nsplit, wg = (j / p.nsplit) * 8 + xcd;      // workgroup `wg` of column part `part`
  constexpr int nop = NPR * 32;                                      // this part's output columns
  const int n0 = part * nop;
  const int nwaves = p.wg_per_part * NW, wid = wg * NW + wave;
  const int nsteps = (p.rows + 16 * RT - 1) / (16 * RT);
  const bool cast_out = AF32 && p.a_cast != nullptr && part == 0;
  // chunk swizzle of weight row r (bit 2 of r is not used: the two tiles of a pair, rows r and r + 4, share it).
  // Rows of 8 or 24 chunks start on alternating halves of the 256-B bank row, so three bits do; rows of 16 / 32
  // chunks all start on slot 0 and need four.
  auto wswz = [](int r) { return (CPR & 8) ? ((((r >> 3) & 3) << 1) | ((r >> 1) & 1)) : ((((r >> 3) & 3) << 2) | (r & 3)); };

  // ---- first tile's loads in flight while the weight part is staged ----
  u32x4 nb[RT][KT][AF32 ? 2 : 1];
  auto issue_a = [&](int step) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int tc = min(step * (16 * RT) + 16 * rt + li, p.rows - 1);
      const char* ap = p.a + (size_t)tc * p.lda * (AF32 ? 4 : 2) + g * (AF32 ? 32 : 16);
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        nb[rt][ks][0] = ldg16(ap + ks * (AF32 ? 128 : 64));
        if constexpr (AF32) nb[rt][ks][1] = ldg16(ap + ks * 128 + 16);
      }
    }
  };
  int step = wid;
  if (step < nsteps) issue_a(step);
  {
    // weight part -> LDS, eight 16-B loads per thread in flight at a time (a load-store loop would pay one L2 round
    // trip per iteration)
    float* sb = reinterpret_cast<float*>(smem + nop * ROWB);         // this part's bias, rounded to the element type
    for (int i = tid; i < nop; i += NT) sb[i] = p.bias ? E::to_f(E::from_f(p.bias[n0 + i])) : 0.f;
    const int total = nop * CPR;
    if (p.w_mode == 0) {
      const char* wsrc = p.w + (size_t)n0 * K * 2;
      for (int base = 0; base < total; base += 8 * NT) {
        u32x4 wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = ldg16(wsrc + (size_t)min(base + u * NT + tid, total - 1) * 16);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * NT + tid;
          const int row = idx / CPR, c = idx - row * CPR;
          if (idx < total) sts16(smem + row * ROWB + ((c ^ wswz(row)) << 4), wv[u]);
        }
      }
    } else if (p.w_mode == 1) {
      // fp32 master weight, rounded on the way in (the autocast cast of the weight folded into the staging)
      const float* wsrc = reinterpret_cast<const float*>(p.w) + (size_t)n0 * K;
      for (int base = 0; base < total; base += 4 * NT) {
        f32x4 wv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* s8 = wsrc + (size_t)min(base + u * NT + tid, total - 1) * 8;
          wv[u][0] = *reinterpret_cast<const f32x4*>(s8);
          wv[u][1] = *reinterpret_cast<const f32x4*>(s8 + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * NT + tid;
          const int row = idx / CPR, c = idx - row * CPR;
          const float f[8] = {wv[u][0][0], wv[u][0][1], wv[u][0][2], wv[u][0][3], wv[u][1][0], wv[u][1][1], wv[u][1][2], wv[u][1][3]};
          if (idx < total) sts16(smem + row * ROWB + ((c ^ wswz(row)) << 4), pack8<E>(f));
        }
      }
    } else {
      // fp32 source [K, NO]: row o of the staged part is COLUMN n0 + o of the source (dX = dY W without a transposed
      // copy of W).  Adjacent lanes take adjacent columns, so the eight 4-byte loads of a chunk are coalesced.
      const float* wsrc = reinterpret_cast<const float*>(p.w) + n0;
      for (int base = 0; base < total; base += 4 * NT) {
        float wv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = min(base + u * NT + tid, total - 1);
          const int c = idx / nop, row = idx - c * nop;
#pragma unroll
          for (int i = 0; i < 8; ++i) wv[u][i] = wsrc[(size_t)(8 * c + i) * p.NO + row];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * NT + tid;
          const int c = idx / nop, row = idx - c * nop;
          if (idx < total) sts16(smem + row * ROWB + ((c ^ wswz(row)) << 4), pack8<E>(wv[u]));
        }
      }
    }
  }
  // weight fragments: MFMA row li of the two tiles of a 32-column pair <-> weight row 8 (li >> 2) + (li & 3) [+ 4].
  // The 16-byte chunks of a row are XOR-swizzled by wswz(row), chosen so that the 16 lanes ds_read_b128 serves
  // together (MI355X_MICROARCH.md LDS table: quarter-waves mixing two values of g) land on 16 different slots.
  const int wrow = 8 * (li >> 2) + (li & 3);
  const int wsz = wswz(wrow);
  const int wb = wrow * ROWB + ((g ^ (wsz & 3)) << 4);
  int wq[4];                                                          // k-step ks: wq[ks & 3] + (ks >> 2) * 256
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) wq[jj] = wb + ((jj ^ (wsz >> 2)) << 6);
  __syncthreads();

  for (; step < nsteps; step += nwaves) {
    typename E::x8 af[RT][KT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // rows past the end are clamped to the last row (loads AND stores: those lanes redo its work and write the same
      // values) -- no predicated memory operation in the loop, so the counts behind the s_waitcnt's are static
      const int tok = min(step * (16 * RT) + 16 * rt + li, p.rows - 1);
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        if constexpr (AF32) {
          const f32x4 lo = __builtin_bit_cast(f32x4, nb[rt][ks][0]), hi = __builtin_bit_cast(f32x4, nb[rt][ks][1]);
          const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const u32x4 w8 = pack8<E>(f);
          af[rt][ks] = as_x8<E>(w8);
          if (cast_out) stg16(p.a_cast + ((size_t)tok * K + ks * 32 + g * 8) * 2, w8);
        } else {
          af[rt][ks] = as_x8<E>(nb[rt][ks][0]);
        }
      }
    }
    if (step + nwaves < nsteps) issue_a(step + nwaves);

#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
      f32x4 acc[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* wp = smem + pr * (32 * ROWB);
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        const typename E::x8 w0 = as_x8<E>(lds16(wp + wq[ks & 3] + (ks >> 2) * 256));
        const typename E::x8 w1 = as_x8<E>(lds16(wp + wq[ks & 3] + (ks >> 2) * 256 + 4 * ROWB));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[rt][0] = E::mma(w0, af[rt][ks], acc[rt][0]);
          acc[rt][1] = E::mma(w1, af[rt][ks], acc[rt][1]);
        }
      }
      // lane: columns n .. n+7 of token li
      const int n = n0 + pr * 32 + 8 * g;
      const float* sbp = reinterpret_cast<const float*>(smem + nop * ROWB) + pr * 32 + 8 * g;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbp), b1 = *reinterpret_cast<const f32x4*>(sbp + 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int tok = min(step * (16 * RT) + 16 * rt + li, p.rows - 1);
        const f32x4 v0 = acc[rt][0] + b0, v1 = acc[rt][1] + b1;
        if constexpr (YF32) {
          char* yp = p.y + ((size_t)tok * p.ldy + n) * 4;
          *reinterpret_cast<f32x4*>(yp) = v0;
          *reinterpret_cast<f32x4*>(yp + 16) = v1;
        } else {
          u32x4 o;
          o[0] = pack2<E>(v0[0], v0[1]); o[1] = pack2<E>(v0[2], v0[3]);
          o[2] = pack2<E>(v1[0], v1[1]); o[3] = pack2<E>(v1[2], v1[3]);
          stg16(p.y + ((size_t)tok * p.ldy + n) * 2, o);
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // one column pair at a time: the unrolled pairs are not to be interleaved
    }
  }
}

static int lin_cus() { return ea_device_cus(); }

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// weight rows of a part (its bias, <= 1.5 KB, rides on top): two workgroups per CU / one 8-wave workgroup
constexpr int LIN_LDS_MAX = 72 * 1024, LIN_LDS_MAX1 = 112 * 1024;

template <typename E, int KT, int RT, bool AF32, int NPR, bool YF32>
static int launch_lin1(const LinP& p, hipStream_t st) {
  const size_t lds = (size_t)(NPR * 32) * (KT * 64 + 4);
  EA_SET_LDS_ONCE((&lin_kernel<E, KT, RT, AF32, NPR, YF32>), (NPR > 8 ? LIN_LDS_MAX1 : LIN_LDS_MAX) + 2048);
  const dim3 grid((unsigned)(p.wg_per_part * p.nsplit)), block(NPR > 8 ? 512 : 256);
  hipLaunchKernelGGL((lin_kernel<E, KT, RT, AF32, NPR, YF32>), grid, block, lds, st, p);
  return (int)hipGetLastError();
}


// column parts: the fewest with 64 / 128 / 192 / 256 weight rows each that fit the LDS budget of two workgroups per CU;
// 576 columns of 192 channels (the qkv projection of a 192-wide model) go as two parts of 288 on one 8-wave workgroup
// per CU instead: the activations pass the load pipes twice, not three times
static int lin_nsplit(int K, int NO) {
  static const int wide = env_int("EA_LIN_WIDE", 1);
  if (wide && NO == 576 && K == 192) return 2;
  for (int ns = 1; ns <= 4; ++ns) {
    if (NO % ns) continue;
    const int nop = NO / ns;
    if ((nop == 64 || nop == 128 || nop == 192 || nop == 256) && (size_t)nop * K * 2 <= (size_t)LIN_LDS_MAX) return ns;
  }
  return 0;       // more than four parts: every part re-reads the activations, leave it to the library
}

int linear_supported(int K, int NO) {
  if (K != 64 && K != 128 && K != 192 && K != 256) return 0;
  return NO > 0 && lin_nsplit(K, NO) > 0;
}

template <typename E, int KT, int RT, bool AF32, int NPR>
static int launch_lin(const LinP& p, hipStream_t st) {
  return p.y_f32 ? launch_lin1<E, KT, RT, AF32, NPR, true>(p, st) : launch_lin1<E, KT, RT, AF32, NPR, false>(p, st);
}

template <typename E, bool AF32, int KT, int RT>
static int lin_by_n(const LinP& p, hipStream_t st) {
  switch (p.NO / p.nsplit) {
    case 64: return launch_lin<E, KT, RT, AF32, 2>(p, st);
    case 128: return launch_lin<E, KT, RT, AF32, 4>(p, st);
    case 192: return launch_lin<E, KT, RT, AF32, 6>(p, st);
    case 256: return launch_lin<E, KT, RT, AF32, 8>(p, st);
    case 288:
      if constexpr (KT == 6) return launch_lin<E, KT, RT, AF32, 9>(p, st);
      return EA_E_UNSUPPORTED;
    default: return EA_E_UNSUPPORTED;
  }
}

template <typename E, bool AF32>
static int lin_by_k(const LinP& p, int K, hipStream_t st) {
  switch (K) {
    case 64: return lin_by_n<E, AF32, 2, 2>(p, st);
    case 128: return lin_by_n<E, AF32, 4, 2>(p, st);
    case 192: return lin_by_n<E, AF32, 6, 2>(p, st);
    case 256: return lin_by_n<E, AF32, 8, 1>(p, st);
    default: return EA_E_UNSUPPORTED;
  }
}

int linear_dispatch(int dtype, const void* a, int a_f32, const void* w, int w_mode, const float* bias, void* y, int y_f32,
                    void* a_cast, int rows, int K, int NO, long lda, long ldy, hipStream_t st) {
  if (!linear_supported(K, NO)) return EA_E_UNSUPPORTED;
  if (rows <= 0) return EA_OK;
  LinP p;
  p.a = (const char*)a; p.w = (const char*)w; p.bias = bias; p.y = (char*)y;
  p.a_cast = a_f32 ? (char*)a_cast : nullptr;
  p.rows = rows; p.NO = NO; p.y_f32 = y_f32; p.w_mode = w_mode; p.lda = lda; p.ldy = ldy;
  p.nsplit = lin_nsplit(K, NO);
  // two resident workgroups per CU in total, a multiple of 8 per part (one XCD each), no more than there are tiles
  static const int per_cu_env = env_int("EA_LIN_PER_CU", 0);
  const int per_cu = per_cu_env > 0 ? per_cu_env : (NO / p.nsplit > 256 ? 1 : 2);
  const int rt = K <= 192 ? 2 : 1;
  const int nsteps = (rows + 16 * rt - 1) / (16 * rt);
  int wgp = per_cu * lin_cus() / p.nsplit / 8 * 8;
  const int nw = NO / p.nsplit > 256 ? 8 : 4;
  const int need = ((nsteps + nw - 1) / nw + 7) / 8 * 8;
  if (wgp > need) wgp = need;
  if (wgp < 8) wgp = 8;
  p.wg_per_part = wgp;
  if (dtype == EA_BF16) return a_f32 ? lin_by_k<BF16, true>(p, K, st) : lin_by_k<BF16, false>(p, K, st);
  if (dtype == EA_F16) return a_f32 ? lin_by_k<F16, true>(p, K, st) : lin_by_k<F16, false>(p, K, st);
  return EA_E_BADARG;
}

}  // namespace ea
