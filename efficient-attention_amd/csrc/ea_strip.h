// ea_strip.h -- small dense matrix algebra on 64 x 64 tiles for 4-wave workgroups (ea_lmk2.hip,
// ea_scatter.hip): wave w owns the 16-row strip w of every matrix product and keeps it in registers in the
// MFMA result layout (row 16 w + 4 g + r, column 16 ct + li); LDS holds only MFMA operands, as 16-bit
// [*][64] tiles with the XOR swizzle of ea_common.h.  A strip is stored TRANSPOSED (four consecutive rows
// of one column = one 8-byte store); both operand orientations of a stored matrix are then available --
// index along the rows by two 8-byte reads (rowfrag), index along the columns by two ds_read_b64_tr_b16
// (colfrag) -- in the same k-slot order, so no product needs a second copy of an operand.
#pragma once
#include "ea_common.h"

namespace ea {
namespace strip {



template <int W> EA_DEV int toff(int row, int col) { return lds_off<W>(row, col >> 3) + ((col & 7) << 1); }

struct Lane { int g, li, w; };

// fragment whose 16 indexed lanes run over ROWS 16 t + li of the tile and whose k-slots run over the columns
template <typename H, int W> EA_DEV typename H::x8 rowfrag(const char* tile, int t, int ks, const Lane& l) {
  const int row = 16 * t + l.li;
  const u32x2 lo = *reinterpret_cast<const u32x2*>(tile + toff<W>(row, 32 * ks + 4 * l.g));
  const u32x2 hi = *reinterpret_cast<const u32x2*>(tile + toff<W>(row, 32 * ks + 16 + 4 * l.g));
  return as_x8<H>(lo, hi);
}
// fragment whose 16 indexed lanes run over COLUMNS 16 t + li and whose k-slots run over the rows
template <typename H, int W> EA_DEV typename H::x8 colfrag(const char* tile, int t, int ks, const Lane& l) {
  const int r = 32 * ks + 4 * l.g + (l.li >> 2);
  const int col = 16 * t + 4 * (l.li & 3);
  return as_x8<H>(H::tr4(tile + toff<W>(r, col)), H::tr4(tile + toff<W>(r + 16, col)));
}

// out[ct] += sum_k A(row 16 w + ., k) B(k, col 16 ct + .): AT = A is stored [k][m] (else [m][k]);
// BT = B is stored [n][k] (else [k][n]); KS 32-deep steps; NT column tiles
template <typename H, int WA, bool AT, int WB, bool BT, int NT>
EA_DEV void mm(f32x4* out, const char* A, const char* B, int KS, const Lane& l) {
  for (int ks = 0; ks < KS; ++ks) {
    const typename H::x8 a = AT ? colfrag<H, WA>(A, l.w, ks, l) : rowfrag<H, WA>(A, l.w, ks, l);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const typename H::x8 b = BT ? rowfrag<H, WB>(B, ct, ks, l) : colfrag<H, WB>(B, ct, ks, l);
      out[ct] = H::mma(a, b, out[ct]);
    }
  }
}

template <int NT> EA_DEV void zero(f32x4* s) {
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) s[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// strip (rows 16 w + 4 g + r, columns 16 ct + li) -> TRANSPOSED fp16 tile T[col][row] ([*][64]); entries
// outside [rows, cols) are stored as zero
template <typename H, int NT> EA_DEV void store_t(char* tile, const f32x4* s, float scale, int rows, int cols, const Lane& l) {
  const int r0 = 16 * l.w + 4 * l.g;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int col = 16 * ct + l.li;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (r0 + r < rows && col < cols) ? s[ct][r] * scale : 0.f;
    *reinterpret_cast<u32x2*>(tile + toff<64>(col, r0)) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
  }
}

// fp32 [rows][ld] matrix in GLOBAL memory <-> strip, through raw buffer instructions (round 3).
// A strip element (row r0 + r, column 16 ct + li) sits at byte offset ((r0 + r) ld + li) 4 + 64 ct: ONE offset register per
// r plus an immediate, no 64-bit address arithmetic; the buffer's size is rows * ld * 4, so rows beyond `rows` read as
// zero and their stores are dropped BY THE HARDWARE bounds check -- no clamp, no select, no exec-mask branch per element.
// (The dword-per-element form with flat addresses was ~40 % of the landmark kernels' instructions: 16 address computations
// per strip, a predicated region per stored element, and `cond ? load : 0` selects that hipcc turned back into branches
// with a full wait behind each.)
EA_DEV __amdgpu_buffer_rsrc_t strip_rsrc(const float* base, int rows, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, rows * ld * 4, 0x00020000);
}
constexpr int STRIP_OOB = 0x40000000;               // an offset no buffer here reaches: reads 0, stores nothing
EA_DEV float strip_ld(__amdgpu_buffer_rsrc_t rs, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
// CM: the matrix is narrower than the strip (cols < 16 NT): mask the columns too (otherwise `cols` is not looked at and
// the column part of every offset is an immediate)
template <int NT, bool CM = false> EA_DEV void load_strip(f32x4* s, const float* src, int ld, int rows, int cols, const Lane& l) {
  if (!src) { zero<NT>(s); return; }                 // (uniform)
  const __amdgpu_buffer_rsrc_t rs = strip_rsrc(src, rows, ld);
  const int base = ((16 * l.w + 4 * l.g) * ld + l.li) * 4;
  constexpr bool cm = CM;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const bool cok = !cm || 16 * ct + l.li < cols;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[ct][r] = strip_ld(rs, cok ? base + r * ld * 4 + 64 * ct : STRIP_OOB);
  }
}
// rows given explicitly (ri[r] < nrows), zero where !ok[r]
template <int NT, bool CM = false>
EA_DEV void gather_strip(f32x4* s, const float* src, int ld, int nrows, const int* ri, const bool* ok, int cols, const Lane& l) {
  const __amdgpu_buffer_rsrc_t rs = strip_rsrc(src, nrows, ld);
  constexpr bool cm = CM;
  int ro[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ro[r] = ok[r] ? (ri[r] * ld + l.li) * 4 : STRIP_OOB;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const bool cok = !cm || 16 * ct + l.li < cols;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[ct][r] = strip_ld(rs, cok ? ro[r] + 64 * ct : STRIP_OOB);
  }
}
template <int NT, bool CM = false> EA_DEV void gsave_strip(float* dst, const f32x4* s, int ld, int rows, int cols, const Lane& l) {
  const __amdgpu_buffer_rsrc_t rs = strip_rsrc(dst, rows, ld);
  const int base = ((16 * l.w + 4 * l.g) * ld + l.li) * 4;
  constexpr bool cm = CM;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const bool cok = !cm || 16 * ct + l.li < cols;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)s[ct][r]), rs, cok ? base + r * ld * 4 + 64 * ct : STRIP_OOB, 0, 0);
  }
}
// strip -> fp32 [rows][ld] matrix behind a plain pointer (LDS exchange buffers)
template <int NT> EA_DEV void save_strip(float* dst, const f32x4* s, int ld, int rows, int cols, const Lane& l) {
  const int r0 = 16 * l.w + 4 * l.g;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int col = 16 * ct + l.li;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r0 + r < rows && col < cols) dst[(size_t)(r0 + r) * ld + col] = s[ct][r];
  }
}

// reductions over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15): four rotations within the row, each fused into
// its add / max as a DPP operand.  (__shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip per step; the landmark
// backward issued 174 of them in dependent chains.)  Every lane ends up with the full result.
EA_DEV float row16_sum(float v) {
  v += dpp_mov<0x128>(v);   // row_ror:8
  v += dpp_mov<0x124>(v);   // row_ror:4
  v += dpp_mov<0x122>(v);   // row_ror:2
  v += dpp_mov<0x121>(v);   // row_ror:1
  return v;
}
EA_DEV float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x121>(v));
  return v;
}
EA_DEV float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <int NT> EA_DEV float strip_absmax(const f32x4* s, int rows, int cols, const Lane& l) {
  const int r0 = 16 * l.w + 4 * l.g;
  float m = 0.f;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r0 + r < rows && 16 * ct + l.li < cols) m = fmaxf(m, fabsf(s[ct][r]));
  return wave_maxf(m);
}
// power-of-two 1/scale: max * scale in [0.5, 1)
EA_DEV float pow2_scale(const float* gm) {
  const float m = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
  if (!(m > 0.f) || m > 3e38f) return 1.f;
  int e;
  (void)frexpf(m, &e);
  return ldexpf(1.f, -e);
}
// column sums of a strip over its rows -> part[w][col] (the caller adds the four waves after a barrier)
template <int NT> EA_DEV void colsum_part(float* part, const f32x4* s, int rows, const Lane& l) {
  const int r0 = 16 * l.w + 4 * l.g;
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) a += (r0 + r < rows) ? s[ct][r] : 0.f;
    a = quad_sum(a);
    if (l.g == 0) part[l.w * 64 + 16 * ct + l.li] = a;
  }
}


}  // namespace strip
}  // namespace ea
