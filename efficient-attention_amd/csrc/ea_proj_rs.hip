// ea_proj_rs.hip -- the qkv projection of a 192-wide model with the WEIGHT RESIDENT IN REGISTERS (round 3).
//     qkv[t][o] = sum_k x[t][k] W[o][k] + b[o],   K = 192, 576 output columns, t over all B*N tokens
// (abstract_attention.py:72-78).  ea_linear.hip keeps the weight in LDS, which at 576 x 192 needs two column parts: every
// activation row passes the CU's load pipe twice (385 MB through the pipes for 231 MB of algorithmic traffic: 66 us).
// Here a 12-wave workgroup (one per CU) holds the whole weight in its registers -- wave w owns the 48 output columns
// 48 w .. 48 w + 47 as MFMA A operands, 72 VGPRs per lane, loaded once from the fp32 master weight and rounded on the way
// -- and the activations stream through LDS exactly once: a 32-token tile is fetched by all 768 threads (one 16-byte
// bf16 chunk each: two 16-byte fp32 loads, the autocast cast folded in; the rounded copy is written out for the
// weight-gradient pass), double-buffered, one barrier per tile; every wave then reads the tile's rows as B operands
// (ds_read_b128, conflict-free phi2 layout per 64-channel slab) and forms its [48 x 32] piece transposed, D[out][token].
// The weight rows of a tile PAIR are interleaved in fours, so a lane ends up with eight consecutive columns of one token
// (16-byte stores); the third tile of a wave gives 8-byte stores.
#include "ea_common.h"

namespace ea {

struct RsP {
  const char* a;        // [rows, 192] fp32 (AF32) or element type, row stride lda elements
  const float* w;       // [576, 192] fp32 master weight
  const float* bias;    // [576] fp32 or null (rounded to the element type before it is added, as F.linear under autocast does)
  char* y;              // [rows, 576] element type, row stride ldy elements
  char* a_cast;         // [rows, 192] element-type copy of a (AF32 only) or null
  char* w_cast;         // [576, 192] element-type copy of w or null (workgroup 0 writes it: the input-gradient GEMM of the
                        // backward takes a 16-bit weight, and a cast launch of its own costs 5 us for 110 k elements)
  int rows, ntiles;
  long lda, ldy;
  // pooling epilogue (POOL > 0): the means of the ROUNDED q / k rows over the r x r token cells of a gw-wide grid -- what
  // ea_eva_chunk_mean_fwd recomputes from the stored rows (lara.py:43,48,145-151; eva.py:178-181 for 2-D chunks) -- leave
  // with the projection, fp32 [B*3, Lc, 64] per side.  A tile is then 32 / (r r) WHOLE cells (token slots gathered).
  float *pq, *pk;
  int gw, cw, Lc, ntok, ncell;        // cw = gw / r cells per grid row, Lc cells per image, ntok = gh gw, ncell = B Lc
  unsigned m_Lc, m_cw;                // floor(2^32 / d) + 1: n / d == umulhi(n, m) for n d < 2^32
  long long* prof;                    // dev builds (-DEA_PROFILE): workgroup time stamps
  const char* wsw;                    // WSW: the weight in the element type, PRE-ARRANGED per (wave, column group, k-step, lane)
                                      // by w192_prepare_kernel: the staging below is 18 coalesced 1-KB loads per wave
};

constexpr int RS_K = 192, RS_NO = 576, RS_WAVES = 12, RS_TOK = 32, RS_KT = RS_K / 32, RS_SLABS = RS_K / 64;

// LDS tile: [slab][token][64 channels] element type, phi2-swizzled 128-byte rows
EA_DEV int rs_off(int slab, int tok, int chunk16) { return slab * (RS_TOK * 128) + lds_off2<64>(tok, chunk16); }

// POOL: tokens per pooling cell (0: none; 16: 4 x 4 cells, two per tile; 4: 2 x 2 cells, eight per tile)
// lane (g, li) of wave `wave`, column group j: the output column whose weight row it holds as MFMA A operand
EA_DEV int rs_col(int wave, int j, int li) {
  const int c0 = 48 * wave;
  return j == 2 ? c0 + 32 + li : c0 + 8 * (li >> 2) + (li & 3) + 4 * j;
}

// WSW (round 6): the weight arrives in the element type, pre-arranged (w192_prepare_kernel).  Workgroup timelines of the
// -DEA_PROFILE build: with the fp32 master weight every CU pulls 442 KB through its L2 port in 16-byte pieces that use half a
// line each -- 13.0 us before the first tile at ANY row count, 56 % of the kernel at N = 196; the prepared copy is 221 KB in
// whole lines.
template <typename E, bool AF32, int POOL, bool WSW>
__global__ __launch_bounds__(RS_WAVES * 64, 3) void proj_rs_kernel(const RsP p) {
  __shared__ __attribute__((aligned(16))) char tile[2][RS_SLABS * RS_TOK * 128];
  __shared__ __attribute__((aligned(16))) float bias_s[RS_NO];          // rounded to the element type
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- this thread's staging slot: token st_tok of the tile, channels 8 st_c .. 8 st_c + 7 ----
  const int st_tok = tid / 24, st_c = tid - st_tok * 24;           // 32 tokens x 24 chunks = 768 slots
  // token behind slot s of tile t.  Plain: consecutive rows.  POOL: the tile's cells' tokens (cell-major, then row-major
  // inside the r x r cell) -- every row is still read and written whole, only the order inside a tile changes -- clamped to
  // the last cell (duplicates rewrite their own values, like the clamped rows of the plain order).
  constexpr int PR = POOL == 16 ? 4 : 2;
  auto cell_of = [&](int t, int s, int& img, int& cl) {
    const int cell = min(t * (RS_TOK / (POOL ? POOL : 1)) + s / (POOL ? POOL : 1), p.ncell - 1);
    img = (int)__umulhi((unsigned)cell, p.m_Lc);
    cl = cell - img * p.Lc;
  };
  auto tok_of = [&](int t, int s) {
    if constexpr (POOL == 0) {
      return min(t * RS_TOK + s, p.rows - 1);
    } else {
      int img, cl;
      cell_of(t, s, img, cl);
      const int cy = (int)__umulhi((unsigned)cl, p.m_cw), cx = cl - cy * p.cw;
      const int w = s % POOL, dy = w / PR, dx = w % PR;
      return img * p.ntok + (cy * PR + dy) * p.gw + cx * PR + dx;
    }
  };
  u32x4 nb[AF32 ? 2 : 1];
  auto issue = [&](int t) {
    const int tok = tok_of(t, st_tok);
    const char* ap = p.a + (size_t)tok * p.lda * (AF32 ? 4 : 2) + st_c * (AF32 ? 32 : 16);
    nb[0] = ldg16(ap);
    if constexpr (AF32) nb[1] = ldg16(ap + 16);
  };
  int t = blockIdx.x;
  EA_BLK(p, 0);
  if (t < p.ntiles) issue(t);
  // ---- the weight slice of this wave -> registers (A operands).  MFMA row li of the tile pair (0, 1) <-> output column
  // 8 (li >> 2) + (li & 3) [+ 4] of the wave's first 32 columns, of tile 2 <-> column 32 + li ----
  typename E::x8 wr[3][RS_KT];
  for (int i = tid; i < RS_NO; i += RS_WAVES * 64) bias_s[i] = p.bias ? E::to_f(E::from_f(p.bias[i])) : 0.f;
  if constexpr (WSW) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int ks = 0; ks < RS_KT; ++ks)
        wr[j][ks] = as_x8<E>(ldg16(p.wsw + ((size_t)((wave * 3 + j) * RS_KT + ks) * 64 + lane) * 16));
  } else {
    const int col[3] = {rs_col(wave, 0, li), rs_col(wave, 1, li), rs_col(wave, 2, li)};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int ks = 0; ks < RS_KT; ++ks) {
        const float* s = p.w + (size_t)col[j] * RS_K + ks * 32 + 8 * g;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
        const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        wr[j][ks] = as_x8<E>(pack8<E>(f));
        // the rounded copy for the backward: piece (wave, j, ks) -- 1 KB -- leaves with workgroup `piece mod grid` (round 6;
        // until then workgroup 0 stored all 221 KB: ~13 us of its CU's store pipe, the tail of the launch)
        if (p.w_cast && (unsigned)((wave * 3 + j) * RS_KT + ks) % gridDim.x == blockIdx.x)
          stg16(p.w_cast + ((size_t)col[j] * RS_K + ks * 32 + 8 * g) * 2, __builtin_bit_cast(u32x4, wr[j][ks]));
      }
  }
  const int slab_s = st_c >> 3, ch_s = st_c & 7;
  // ---- pooling epilogue (POOL > 0), all of it on the matrix pipe -- a first version that summed the rounded outputs
  // across the lanes of a cell (12 values x 4 DPP adds x 2 tiles per wave and tile) cost 9 us of VALU time at cfg3:
  //   (1) xbar[cell][ch] = mean over the cell's tokens of the staged (rounded) x rows: ONE MFMA per wave and tile --
  //       A = the tile's rows read transposed (channel tile 16 (wave & 3) of slab wave >> 2), B = a 0 / 1 pattern that
  //       assigns token slots to cells -- split into a high and a low 16-bit part (exact to 2^-17) and parked in LDS;
  //   (2) one tile later (after that tile's barrier) the q / k waves form  W xbar + b  for the parked cells with their
  //       resident weight rows: pooled rows = the cell means of the UNROUNDED product (the stored q / k rows differ from it
  //       by their final rounding, which averages out over the cell).
  constexpr int NC = POOL ? RS_TOK / POOL : 1;                          // cells per tile
  constexpr int XB = 8 * RS_K * 2;                                      // one [8 cells][192] 16-bit image
  __shared__ __attribute__((aligned(16))) char xbar[POOL ? 2 : 1][POOL ? 2 * XB : 16];
  const int pc0 = 48 * wave;
  auto pool_x = [&](int b_) {
    const int slab = wave >> 2, dt = wave & 3;
    const char* tb = tile[b_] + slab * (RS_TOK * 128);
    const int r_ = 4 * g + (li >> 2);
    const int off = lds_off2<64>(r_, 2 * dt + ((li & 3) >> 1)) + 8 * (li & 1);
    const typename E::x8 a = as_x8<E>(E::tr4(tb + off), E::tr4(tb + 16 * 128 + off));
    // (the pattern is rebuilt per tile -- two compares -- instead of living in four registers: the kernel sits at its
    //  register limit, 168 of 170 at three waves per SIMD)
    const uint32_t one = (uint32_t)E::from_f(1.f) * 0x00010001u;
    const uint32_t lo_on = (POOL == 16 ? li == 0 : li == g) ? one : 0u, hi_on = (POOL == 16 ? li == 1 : li == 4 + g) ? one : 0u;
    const typename E::x8 pat = as_x8<E>(u32x4{lo_on, lo_on, hi_on, hi_on});
    const f32x4 d = E::mma(a, pat, f32x4{0.f, 0.f, 0.f, 0.f});
    // lane (g, li = cell): channels 64 slab + 16 dt + 4 g + r of that cell
    float hi[4], lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = d[r] * (1.f / (POOL ? POOL : 1));
      hi[r] = E::to_f(E::from_f(v));
      lo[r] = v - hi[r];
    }
    if (li < NC) {
      char* dst = xbar[b_] + li * (RS_K * 2) + (64 * slab + 16 * dt + 4 * g) * 2;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<E>(hi[0], hi[1]), pack2<E>(hi[2], hi[3])};
      *reinterpret_cast<u32x2*>(dst + XB) = u32x2{pack2<E>(lo[0], lo[1]), pack2<E>(lo[2], lo[3])};
    }
  };
  auto pool_out = [&](int tp, int b_) {
    if (wave >= 8) return;                                              // v columns: nothing to pool
    f32x4 pa[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const char* xb = xbar[b_] + min(li, NC - 1) * (RS_K * 2);
#pragma unroll
    for (int ks = 0; ks < RS_KT; ++ks) {
      const typename E::x8 bh = as_x8<E>(lds16(xb + (4 * ks + g) * 16));
      const typename E::x8 bl = as_x8<E>(lds16(xb + XB + (4 * ks + g) * 16));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        pa[j] = E::mma(wr[j][ks], bh, pa[j]);
        pa[j] = E::mma(wr[j][ks], bl, pa[j]);
      }
    }
    const int cell = tp * NC + li;
    if (li < NC && cell < p.ncell) {
      const int img = (int)__umulhi((unsigned)cell, p.m_Lc), cl = cell - img * p.Lc;
      // columns pc0 + 8 g .. + 7 and pc0 + 32 + 4 g .. + 3; a 4-column piece never straddles a head (64 | 192)
      const int cols[3] = {pc0 + 8 * g, pc0 + 8 * g + 4, pc0 + 32 + 4 * g};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // (q or k is a property of the WAVE -- waves 0-3 own the q columns, 4-7 the k columns -- so the base pointer stays in
        //  scalar registers; taken per lane from cols[j] it cost a 64-bit VGPR select and an 8-byte spill at 168 VGPRs)
        const int side = wave >> 2, hc = cols[j] - side * RS_K, head = hc >> 6, ch = hc & 63;
        float* dst = (side ? p.pk : p.pq) + ((size_t)(img * 3 + head) * p.Lc + cl) * 64 + ch;
        *reinterpret_cast<f32x4*>(dst) = pa[j] + *reinterpret_cast<const f32x4*>(bias_s + cols[j]);
      }
    }
  };
  int buf = 0, tprev = -1;
  EA_BLKX(p, 0);
  for (; t < p.ntiles; tprev = t, t += gridDim.x, buf ^= 1) {
    // ---- commit this tile's slot: round, park in LDS, write the rounded copy ----
    {
      u32x4 w8;
      if constexpr (AF32) {
        const f32x4 lo = __builtin_bit_cast(f32x4, nb[0]), hi = __builtin_bit_cast(f32x4, nb[1]);
        const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        w8 = pack8<E>(f);
      } else {
        w8 = nb[0];
      }
      sts16(tile[buf] + rs_off(slab_s, st_tok, ch_s), w8);
      if (AF32 && p.a_cast) {
        const int tok = tok_of(t, st_tok);                          // (clamped rows rewrite the last row with its own values)
        stg16(p.a_cast + ((size_t)tok * RS_K + st_c * 8) * 2, w8);
      }
    }
    if (t + (int)gridDim.x < p.ntiles) issue(t + gridDim.x);
    __syncthreads();
#ifndef EA_RS_NOPOOLMATH
    if constexpr (POOL > 0) pool_x(buf);
#endif
    // ---- [48 x 32] piece of this wave ----
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < RS_KT; ++ks) {
      const int slab = ks >> 1;                                      // 32-channel step ks = chunks 4 (ks & 1) .. + 3 of slab ks >> 1
      const typename E::x8 b0 = as_x8<E>(lds16(tile[buf] + rs_off(slab, li, 4 * (ks & 1) + g)));
      const typename E::x8 b1 = as_x8<E>(lds16(tile[buf] + rs_off(slab, 16 + li, 4 * (ks & 1) + g)));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j][0] = E::mma(wr[j][ks], b0, acc[j][0]);
        acc[j][1] = E::mma(wr[j][ks], b1, acc[j][1]);
      }
    }
    // ---- store: lane (g, li) holds, for tokens li and 16 + li, columns c0 + 8 g .. + 7 (pair) and c0 + 32 + 4 g .. + 3 ----
    const int c0 = 48 * wave;
    // bias of the D rows 4 g + r this lane holds: pair -> columns c0 + 8 g + r (+ 4); tile 2 -> c0 + 32 + 4 g + r
    const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bias_s + c0 + 8 * g), bv1 = *reinterpret_cast<const f32x4*>(bias_s + c0 + 8 * g + 4);
    const f32x4 bv2 = *reinterpret_cast<const f32x4*>(bias_s + c0 + 32 + 4 * g);
    u32x2 third[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int tok = tok_of(t, 16 * rt + li);
      char* yp = p.y + ((size_t)tok * p.ldy + c0) * 2;
      const f32x4 v0 = acc[0][rt] + bv0, v1 = acc[1][rt] + bv1, v2 = acc[2][rt] + bv2;
      u32x4 o;
      o[0] = pack2<E>(v0[0], v0[1]); o[1] = pack2<E>(v0[2], v0[3]);
      o[2] = pack2<E>(v1[0], v1[1]); o[3] = pack2<E>(v1[2], v1[3]);
      stg16(yp + 16 * g, o);
      third[rt] = u32x2{pack2<E>(v2[0], v2[1]), pack2<E>(v2[2], v2[3])};
    }
    // third tile: lane-row g holds columns c0 + 32 + 4 g .. + 3 of tokens li (rt 0) and 16 + li (rt 1).  Lane-rows 2 m and
    // 2 m + 1 trade pieces (even row's rt-1 piece <-> odd row's rt-0 piece, one v_permlane16_swap per register) so that row
    // 2 m ends up with columns c0 + 32 + 8 m .. + 7 of token li and row 2 m + 1 with the same columns of token 16 + li:
    // 16-byte stores here too (8-byte stores cost the same issue slot for half the bytes; the store pipe is the bound).
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3"
                 : "+v"(third[0][0]), "+v"(third[0][1]), "+v"(third[1][0]), "+v"(third[1][1]));
    {
      const int tok = tok_of(t, 16 * (g & 1) + li);
      const u32x4 o = {third[0][0], third[0][1], third[1][0], third[1][1]};
      stg16(p.y + ((size_t)tok * p.ldy + c0 + 32 + 8 * (g >> 1)) * 2, o);
    }
    // pooled rows of the PREVIOUS tile (parked one barrier ago; its image is overwritten after the next barrier), once the
    // accumulators of this tile are dead
#ifndef EA_RS_NOPOOLMATH
    if constexpr (POOL > 0) {
      if (tprev >= 0) pool_out(tprev, buf ^ 1);
    }
#endif
  }
  EA_BLKX(p, 1);
  if constexpr (POOL > 0) {
    if (tprev >= 0) {                                                  // the last tile's cells
      __syncthreads();
      pool_out(tprev, buf ^ 1);
    }
  }
  EA_BLK(p, 1);
}

int proj_rs_supported(int K, int NO) { return K == RS_K && NO == RS_NO; }

static unsigned rs_magic(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

// pooled variant: B images of gh x gw tokens, r x r cells (r = 2 or 4), three heads of 64 channels
int proj_rs_pool_supported(int K, int NO, int B, int gh, int gw, int r) {
  if (K != RS_K || NO != RS_NO || (r != 2 && r != 4) || B <= 0 || gh <= 0 || gw <= 0 || gh % r || gw % r) return 0;
  const long Lc = (long)(gh / r) * (gw / r), ncell = (long)B * Lc;
  return ncell * Lc < (1l << 32) && (long)B * gh * gw < (1l << 31);
}

// The 16-bit copies of a 192-wide layer's two weights in ONE launch (round 6): wq [576, 192], wp [192, 192] fp32 ->
//   w16q [576, 192] (input gradient), wsw = the same values in proj_rs_kernel<.., WSW>'s staging order, w16p [192, 192] (output
//   projection), w16pT = its transpose (the output projection's input gradient as the same streaming kernel).
// One thread per 8-element piece; round to nearest even like every cast of the library.
template <typename E>
__global__ __launch_bounds__(256) void w192_prepare_kernel(const float* __restrict__ wq, const float* __restrict__ wp,
                                                            char* __restrict__ w16q, char* __restrict__ wsw,
                                                            char* __restrict__ w16p, char* __restrict__ w16pT) {
  constexpr int NQ = RS_WAVES * 3 * RS_KT * 64, NP = RS_K * RS_K / 8;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < NQ) {
    const int lane = i & 63, pc = i >> 6, ks = pc % RS_KT, wj = pc / RS_KT, j = wj % 3, wave = wj / 3;
    const int g = lane >> 4, li = lane & 15;
    const size_t e = (size_t)rs_col(wave, j, li) * RS_K + ks * 32 + 8 * g;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(wq + e), hi = *reinterpret_cast<const f32x4*>(wq + e + 4);
    const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const u32x4 w8 = pack8<E>(f);
    stg16(wsw + (size_t)i * 16, w8);
    stg16(w16q + e * 2, w8);
  } else if (i < NQ + NP && wp) {
    const int k = i - NQ, row = k / (RS_K / 8), c = k - row * (RS_K / 8);
    const size_t e = (size_t)row * RS_K + c * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(wp + e), hi = *reinterpret_cast<const f32x4*>(wp + e + 4);
    const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const u32x4 w8 = pack8<E>(f);
    stg16(w16p + e * 2, w8);
    uint16_t* t = reinterpret_cast<uint16_t*>(w16pT);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t[(size_t)(c * 8 + 2 * q) * RS_K + row] = (uint16_t)(w8[q] & 0xffffu);
      t[(size_t)(c * 8 + 2 * q + 1) * RS_K + row] = (uint16_t)(w8[q] >> 16);
    }
  }
}

int w192_prepare_dispatch(int dtype, const float* wq, const float* wp, void* w16q, void* wsw, void* w16p, void* w16pT,
                          hipStream_t st) {
  constexpr int N = RS_WAVES * 3 * RS_KT * 64 + RS_K * RS_K / 8;
  const dim3 g((N + 255) / 256), b(256);
  if (dtype == EA_BF16) hipLaunchKernelGGL(w192_prepare_kernel<BF16>, g, b, 0, st, wq, wp, (char*)w16q, (char*)wsw, (char*)w16p, (char*)w16pT);
  else if (dtype == EA_F16) hipLaunchKernelGGL(w192_prepare_kernel<F16>, g, b, 0, st, wq, wp, (char*)w16q, (char*)wsw, (char*)w16p, (char*)w16pT);
  else return EA_E_BADARG;
  return (int)hipGetLastError();
}

// wsw != NULL: the prepared weight (w is then unused and may be NULL)
int proj_rs_dispatch(int dtype, const void* a, int a_f32, const float* w, const float* bias, void* y, void* a_cast, int rows,
                     long lda, long ldy, hipStream_t st, int B, int gh, int gw, int r, float* pq, float* pk, void* w_cast,
                     const void* wsw) {
  if (rows <= 0) return EA_OK;
  RsP p = {};
  p.a = (const char*)a; p.w = w; p.bias = bias; p.y = (char*)y; p.a_cast = a_f32 ? (char*)a_cast : nullptr;
  p.w_cast = wsw ? nullptr : (char*)w_cast;
  p.wsw = (const char*)wsw;
  p.rows = rows; p.ntiles = (rows + RS_TOK - 1) / RS_TOK; p.lda = lda; p.ldy = ldy;
  const int pool = r * r;
  if (pool) {
    if (!pq || !pk || rows != B * gh * gw) return EA_E_BADARG;
    p.pq = pq; p.pk = pk; p.gw = gw; p.cw = gw / r; p.Lc = (gh / r) * p.cw; p.ntok = gh * gw; p.ncell = B * p.Lc;
    p.m_Lc = rs_magic(p.Lc); p.m_cw = rs_magic(p.cw);
    p.ntiles = (p.ncell * pool + RS_TOK - 1) / RS_TOK;
  }
  int grid = ea_device_cus();
  if (grid > p.ntiles) grid = p.ntiles;
  const dim3 g((unsigned)grid), b(RS_WAVES * 64);
#ifdef EA_PROFILE
  ProfReport rep;
  p.prof = rep.arm(st, "proj_rs", pool);
#endif
#define EA_RS_LAUNCH2(E_, AF_, W_)                                                                             \
  do {                                                                                                         \
    if (pool == 16) hipLaunchKernelGGL((proj_rs_kernel<E_, AF_, 16, W_>), g, b, 0, st, p);                     \
    else if (pool == 4) hipLaunchKernelGGL((proj_rs_kernel<E_, AF_, 4, W_>), g, b, 0, st, p);                  \
    else hipLaunchKernelGGL((proj_rs_kernel<E_, AF_, 0, W_>), g, b, 0, st, p);                                 \
  } while (0)
#define EA_RS_LAUNCH(E_, AF_)                                                                                  \
  do {                                                                                                         \
    if (p.wsw) EA_RS_LAUNCH2(E_, AF_, true);                                                                   \
    else EA_RS_LAUNCH2(E_, AF_, false);                                                                        \
  } while (0)
  if (dtype == EA_BF16) {
    if (a_f32) EA_RS_LAUNCH(BF16, true);
    else EA_RS_LAUNCH(BF16, false);
  } else if (dtype == EA_F16) {
    if (a_f32) EA_RS_LAUNCH(F16, true);
    else EA_RS_LAUNCH(F16, false);
  } else {
    return EA_E_BADARG;
  }
#undef EA_RS_LAUNCH
#undef EA_RS_LAUNCH2
  return (int)hipGetLastError();
}

}  // namespace ea
