// ea_dgrad_rs.hip -- the INPUT gradient of the qkv projection of a 192-wide model with the weight resident in registers
// (round 5; abstract_attention.py:72-78 differentiated):
//     dx[t][k] = sum_o dqkv[t][o] W[o][k],   576-deep contraction, 192 output columns, t over all B*N tokens.
// Until round 4 this was the one library GEMM of the 192-wide layer step (a hipBLASLt Cijk_* picked by TunableOp, 44 us at
// cfg3 + a 5 us weight cast).  Same plan as the forward projection (ea_proj_rs.hip), mirrored: a 12-wave workgroup per CU
// holds W^T in its registers -- wave w owns the output columns 16 w .. 16 w + 15 as MFMA A operands over all 576 contraction
// slots, 72 VGPRs per lane -- and the dqkv rows stream through LDS exactly once: a 32-token tile (36 KB) is fetched by all 768
// threads (three 16-byte chunks each), double-buffered, one barrier per tile; every wave reads the whole tile as B operands
// (ds_read_b128, phi2 layout per 64-channel slab) and forms its [16 x 32] piece transposed, D[column][token] -> 16-byte fp32
// stores (4 consecutive columns of one token per lane).
// The weight reaches the registers TRANSPOSED without a transposed copy in memory: the [576][192] matrix (fp32 master, rounded
// on the way, or the 16-bit copy the forward projection left) passes LDS in 18 pieces of 32 rows, stored row-permuted so that
// two ds_read_b64_tr_b16 per piece hand lane (g, li) the values W[32 p + 8 g .. + 7][16 w + li] -- its eight k-slots.
#include <stdlib.h>
#include "ea_common.h"

namespace ea {

struct DgP {
  const char* dy;       // [rows, 576] element type, row stride ldy elements
  const char* w;        // [576, 192] fp32 master weight (WF32) or element type
  char* dx;             // [rows, 192] fp32 (OF32) or element type, row stride ldx elements
  int rows, ntiles;
  long ldy, ldx;
  long long* prof;      // dev builds (-DEA_PROFILE): workgroup time stamps
};

constexpr int DG_K = 576, DG_NO = 192, DG_WAVES = 12, DG_TOK = 32, DG_KT = DG_K / 32, DG_SLABS = DG_K / 64;
constexpr int DG_TILE = DG_SLABS * DG_TOK * 128;          // 36 KB
constexpr int DG_PIECE = 32 * DG_NO * 2;                  // one 32-row weight piece, 16-bit: 12 KB
constexpr int DG_LDS = 2 * DG_TILE;

EA_DEV int dg_off(int slab, int tok, int chunk16) { return slab * (DG_TOK * 128) + lds_off2<64>(tok, chunk16); }

template <typename E, bool WF32, bool OF32>
__global__ __launch_bounds__(DG_WAVES * 64, 3) void dgrad_rs_kernel(const DgP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const tile0 = smem;
  char* const tile1 = smem + DG_TILE;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- staging slots of this thread: chunk s = tid + 768 i of the tile's 32 x 72 16-byte chunks ----
  int s_tok[3], s_ch[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int s = tid + i * (DG_WAVES * 64);
    s_tok[i] = s / 72;
    s_ch[i] = s - s_tok[i] * 72;
  }
  u32x4 nb[3];
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tok = min(t * DG_TOK + s_tok[i], p.rows - 1);
      nb[i] = ldg16(p.dy + ((size_t)tok * p.ldy + s_ch[i] * 8) * 2);
    }
  };
  int t = blockIdx.x;
  if (t < p.ntiles) issue(t);
  // ---- W^T -> registers through LDS, 18 pieces of 32 rows (the piece images alias tile 1, which the main loop first
  // writes after its first barrier).  Row o of a piece is parked at row rho(o) = ((o >> 2) & 1) 16 + (o >> 3) 4 + (o & 3):
  // the two transposed reads of lane-row g then cover o = 8 g .. 8 g + 3 and 8 g + 4 .. 8 g + 7, in k-slot order. ----
  typename E::x8 wr[DG_KT];
  {
    const int o_l = tid / 24, c = tid - o_l * 24;                     // 32 rows x 24 chunks of 8 columns = 768 slots
    const int rho = ((o_l >> 2) & 1) * 16 + (o_l >> 3) * 4 + (o_l & 3);
    const int wofs = rho * (DG_NO * 2) + c * 16;
    const int rofs = (4 * g + (li >> 2)) * (DG_NO * 2) + (16 * wave + 4 * (li & 3)) * 2;
    constexpr int BATCH = 6;
#pragma unroll
    for (int pb = 0; pb < DG_KT; pb += BATCH) {
      u32x4 raw[BATCH][WF32 ? 2 : 1];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const size_t e = (size_t)(32 * (pb + j) + o_l) * DG_NO + c * 8;
        if constexpr (WF32) {
          raw[j][0] = ldg16(p.w + e * 4);
          raw[j][1] = ldg16(p.w + e * 4 + 16);
        } else {
          raw[j][0] = ldg16(p.w + e * 2);
        }
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        u32x4 w8;
        if constexpr (WF32) {
          const f32x4 lo = __builtin_bit_cast(f32x4, raw[j][0]), hi = __builtin_bit_cast(f32x4, raw[j][1]);
          const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          w8 = pack8<E>(f);
        } else {
          w8 = raw[j][0];
        }
        char* img = tile1 + ((pb + j) & 1) * DG_PIECE;
        sts16(img + wofs, w8);
        __syncthreads();
        wr[pb + j] = as_x8<E>(E::tr4(img + rofs), E::tr4(img + rofs + 16 * (DG_NO * 2)));
      }
    }
  }
  int buf = 0;
  for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
    char* const tb = buf ? tile1 : tile0;
#pragma unroll
    for (int i = 0; i < 3; ++i) sts16(tb + dg_off(s_ch[i] >> 3, s_tok[i], s_ch[i] & 7), nb[i]);
    issue(min(t + (int)gridDim.x, p.ntiles - 1));         // unconditional (the last tile is fetched once more): a static count of
    __syncthreads();                                      // memory operations lets the next commit wait for ITS loads only
    // ---- [16 columns x 32 tokens] piece of this wave: four accumulation chains (two per token half) ----
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DG_KT; ++ks) {
      const int slab = ks >> 1;
      const typename E::x8 b0 = as_x8<E>(lds16(tb + dg_off(slab, li, 4 * (ks & 1) + g)));
      const typename E::x8 b1 = as_x8<E>(lds16(tb + dg_off(slab, 16 + li, 4 * (ks & 1) + g)));
      acc[0][ks & 1] = E::mma(wr[ks], b0, acc[0][ks & 1]);
      acc[1][ks & 1] = E::mma(wr[ks], b1, acc[1][ks & 1]);
    }
    // ---- store: lane (g, li) holds columns 16 wave + 4 g .. + 3 of tokens li and 16 + li ----
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int tok = min(t * DG_TOK + 16 * rt + li, p.rows - 1);     // (clamped rows rewrite the last row with its own values)
      const f32x4 v = acc[rt][0] + acc[rt][1];
      const int col = 16 * wave + 4 * g;
      if constexpr (OF32) {
        *reinterpret_cast<f32x4*>(p.dx + ((size_t)tok * p.ldx + col) * 4) = v;
      } else {
        *reinterpret_cast<u32x2*>(p.dx + ((size_t)tok * p.ldx + col) * 2) = u32x2{pack2<E>(v[0], v[1]), pack2<E>(v[2], v[3])};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dgrad_fin_kernel (round 5): the LAST corrections of dq / dk and the input gradient in ONE pass over the gradient rows.
//   LARA (lara.py:201-246, 43,48,145-151 differentiated; until now ea_lara_bwd_finish, 242 MB / 53 us at cfg3, followed by
//   the input-gradient product, 192 MB / 44 us):
//       dq_n -= s sum_c t[c,n] (u_c qbar_c)          t = softmax over the sequence of s qbar_c.q_n  (HAS_T)
//       dq_n += d(pooled q)[cell(n)] / r^2,   dk_n += d(pooled k)[cell(n)] / r^2
//       dx_n  = [dq_n | dk_n | dv_n] W
//   EVA (eva.py:178-181 differentiated; until now ea_eva_chunk_mean_bwd): the two pooling terms only (HAS_T = false).
// The 32-token tile of gradient rows is in LDS anyway: the pooling terms are added while it is staged (dk; dq without the
// t term), the t correction of dq is formed on the matrix pipe from the tile's q rows and the (b, head)'s landmark rows
// (u qbar, qbar, lse_t: 48 KB of LDS for the three heads) and applied to the staged rows in place; the corrected dq / dk rows
// go back to memory (the weight-gradient pass reads them) and the input-gradient product runs on the corrected tile.
// A workgroup owns (image, token range) units -- the landmark rows belong to one image -- and loops over them.
struct DgFinP {
  DgP d;                              // d.dy: dqkv rows, read AND (dq / dk columns) rewritten; d.rows = B * ntok
  const char* qkv;                    // forward qkv rows [rows, 576] (q = columns 0 .. 191), row stride ldq (HAS_T)
  long ldq;
  const float *qbar, *uq, *lse_t;     // [B*3, C, 64], [B*3, C, 64], [B*3, C]  (HAS_T)
  const float *dpq, *dpk;             // [B*3, L, 64] gradients of the pooled q / k rows, or null
  int ntok, gw, pool_r, cw, L, C;     // tokens per image, grid width, cell side, cells per grid row / per image, samples
  float pool_inv, scale, scale_log2;
  int splits, tps, nunits;            // token ranges per image, tokens per range (multiple of 32), B * splits
  unsigned m_gw, m_r;                 // floor(2^32 / gw) + 1, floor(2^32 / pool_r) + 1
  // cell-major token order (PR = 2 / 4, round 6): a tile = 32 / (PR PR) WHOLE pooling cells, a unit = `cps` cells of an image
  int cps;                            // cells per unit (multiple of the cells per tile)
  unsigned m_cw;                      // floor(2^32 / cw) + 1
};

constexpr int DG_QT = 3 * DG_TOK * 128;                   // q rows of a tile: 12 KB
constexpr int DG_LMR = 64 * 128;                          // one [64][64] landmark matrix, 16-bit: 8 KB
constexpr int DG_FIN_LDS = 2 * DG_TILE + 2 * DG_QT + 6 * DG_LMR + 3 * 64 * 4;

// PR (round 6): 0 = a tile is 32 consecutive tokens of the image (any pooling cell size); 2 / 4 = a tile is 8 / 2 whole
// PR x PR pooling cells, tokens cell-major (every row is still read and written whole, only the order inside a tile changes
// -- ea_proj_rs.hip walks the forward the same way).  In row-major order a 32-token tile of a 28-wide grid touches 8-9 cells
// per head and side: 12 KB of pooled-gradient rows per tile next to the tile's own 48 KB, a quarter more through the CU's
// L2 port (measured with the reads compiled out: 86 -> 76 us at cfg3); cell-major it is 2 cells = 3 KB.
template <typename E, bool WF32, bool OF32, bool HAS_T, bool POOL, int PR>
__global__ __launch_bounds__(DG_WAVES * 64, 3) void dgrad_fin_kernel(const DgFinP p) {
  static_assert(PR == 0 || (POOL && (PR == 2 || PR == 4)), "cell-major order needs 2 x 2 or 4 x 4 pooling cells");
  constexpr int TPC = PR * PR, CPT = PR ? DG_TOK / (PR * PR) : 1;      // tokens per cell, cells per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const tile0 = smem;
  char* const tile1 = smem + DG_TILE;
  char* const qt0 = smem + 2 * DG_TILE;
  char* const R1 = qt0 + 2 * DG_QT;                       // u qbar rows, three heads
  char* const R2 = R1 + 3 * DG_LMR;                       // qbar rows
  float* const LS = reinterpret_cast<float*>(R2 + 3 * DG_LMR);   // lse_t in log2 units, +inf beyond C
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool pool = POOL;                             // (compile-time: the loop's memory-operation counts must be static)
  EA_BLK(p.d, 0);
  // ---- staging slot of this thread: token s_tok of the tile, 16-byte chunk s_c of its q, its k and its v columns (32 tokens
  // x 24 chunks = 768 threads; chunk s_c = head s_c >> 3, channels 8 (s_c & 7) ..) -- which slot is q / k / v is static, so
  // only the rows that take a pooling term carry one in flight ----
  // (computed where they are used, from an opaque copy of the thread index per tile: as loop invariants they and the LDS /
  //  global offsets derived from them sat in ~16 registers that the tile loop does not have -- they were spilled and re-read)
  // ---- W^T -> registers (as dgrad_rs_kernel) ----
  typename E::x8 wr[DG_KT];
  {
    const int o_l = tid / 24, c = tid - o_l * 24;
    const int rho = ((o_l >> 2) & 1) * 16 + (o_l >> 3) * 4 + (o_l & 3);
    const int wofs = rho * (DG_NO * 2) + c * 16;
    const int rofs = (4 * g + (li >> 2)) * (DG_NO * 2) + (16 * wave + 4 * (li & 3)) * 2;
    constexpr int BATCH = 6;
#pragma unroll
    for (int pb = 0; pb < DG_KT; pb += BATCH) {
      u32x4 raw[BATCH][WF32 ? 2 : 1];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const size_t e = (size_t)(32 * (pb + j) + o_l) * DG_NO + c * 8;
        if constexpr (WF32) {
          raw[j][0] = ldg16(p.d.w + e * 4);
          raw[j][1] = ldg16(p.d.w + e * 4 + 16);
        } else {
          raw[j][0] = ldg16(p.d.w + e * 2);
        }
      }
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        u32x4 w8;
        if constexpr (WF32) {
          const f32x4 lo = __builtin_bit_cast(f32x4, raw[j][0]), hi = __builtin_bit_cast(f32x4, raw[j][1]);
          const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          w8 = pack8<E>(f);
        } else {
          w8 = raw[j][0];
        }
        char* img = tile1 + ((pb + j) & 1) * DG_PIECE;
        sts16(img + wofs, w8);
        __syncthreads();
        wr[pb + j] = as_x8<E>(E::tr4(img + rofs), E::tr4(img + rofs + 16 * (DG_NO * 2)));
      }
    }
  }
  typename LaneOffSel<64>::type lo;
  lo.init(lane);
  const int fh = wave >> 2, fnt = (wave >> 1) & 1, fdh = wave & 1;   // t correction: head, 16-token half, channel half of this wave
  int buf = 0;
  EA_BLKX(p.d, 0);
  for (int u = blockIdx.x; u < p.nunits; u += gridDim.x) {
    const int b = u / p.splits, sp = u - b * p.splits;
    // PR == 0: token range [n0, n1) of the image; PR > 0: cell range [n0, n1)
    const int n0 = PR ? sp * p.cps : sp * p.tps, n1 = PR ? min(p.L, n0 + p.cps) : min(p.ntok, n0 + p.tps);
    const size_t row0 = (size_t)b * p.ntok;
    const int ntile = PR ? (n1 - n0 + CPT - 1) / CPT : (n1 - n0 + DG_TOK - 1) / DG_TOK;
    // pooling cell behind slot row r of tile t (PR > 0), clamped to the unit (duplicates recompute and rewrite identical values)
    auto slot_cell = [&](int t, int r) { return min(n0 + t * CPT + r / (TPC ? TPC : 1), n1 - 1); };
    // slot row r of tile t lies beyond the unit: a DUPLICATE of the unit's last cell / token.  It loads, computes and stores like
    // every slot (static memory-operation counts), but its stores go to the trash line: two waves own the two copies of such a
    // row and no barrier separates one's write-back from the other's load, so a copy that read the corrected row would add
    // the pooling term twice and race the owner's stores (round 6: seen as run-to-run differences of dx at B = 3, 28 x 28, r = 4)
    auto dup = [&](int t, int r) {
      if constexpr (PR == 0) return n0 + t * DG_TOK + r > n1 - 1;
      else return n0 + t * CPT + r / (TPC ? TPC : 1) > n1 - 1;
    };
    // (its address re-derived from an opaque copy of the thread index at every use: as a loop invariant it costs the two registers
    //  that push the PR = 2 instantiations into scratch)
    auto trash_line = [&]() {
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      return ea_trash + (t_ & 511) * 64;
    };
    // token (within the image) behind slot row r of tile t, clamped likewise
    // (divisions as multiply-high by floor(2^32 / d) + 1, exact for n d < 2^32: three integer divisions per slot and tile were
    //  ~1.2 us of VALU time per tile on a kernel whose tile takes 6)
    auto tok_of = [&](int t, int r) {
      if constexpr (PR == 0) {
        return min(n0 + t * DG_TOK + r, n1 - 1);
      } else {
        const int cell = slot_cell(t, r), w = r % TPC;
        const int cy = (int)__umulhi((unsigned)cell, p.m_cw), cx = cell - cy * p.cw;
        return (cy * PR + w / PR) * p.gw + cx * PR + w % PR;
      }
    };
    auto cell_of_tok = [&](int tok) {
      const int y = (int)__umulhi((unsigned)tok, p.m_gw), x = tok - y * p.gw;
      return (int)__umulhi((unsigned)y, p.m_r) * p.cw + (int)__umulhi((unsigned)x, p.m_r);
    };
    // cell of slot row r of tile t
    auto cell_of = [&](int t, int r) {
      if constexpr (PR == 0) return cell_of_tok(tok_of(t, r));
      else return slot_cell(t, r);
    };
    u32x4 nb[3], nq;                                       // dq, dk, dv chunk of the slot; its q chunk (HAS_T)
    f32x4 npk[2], npq[2];                                  // pooled-row gradients of the slot's dk chunk (dq chunk: !HAS_T)
    auto issue = [&](int t, int s_tok, int s_c) {
      const int tok = tok_of(t, s_tok);
      const char* rowp = p.d.dy + ((row0 + tok) * p.d.ldy + s_c * 8) * 2;
      nb[0] = ldg16(rowp);
      nb[1] = ldg16(rowp + 192 * 2);
      nb[2] = ldg16(rowp + 384 * 2);
      if constexpr (HAS_T) nq = ldg16(p.qkv + ((row0 + tok) * p.ldq + s_c * 8) * 2);
#ifdef EA_DGF_NOPOOLREAD        // dev (timing only, wrong results): what do the pooled-gradient reads cost?
      npk[0] = npk[1] = npq[0] = npq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (false) {
#else
      if (pool) {
#endif
        const size_t o = ((size_t)(b * 3 + (s_c >> 3)) * p.L + cell_of(t, s_tok)) * 64 + (s_c & 7) * 8;
        npk[0] = *reinterpret_cast<const f32x4*>(p.dpk + o);
        npk[1] = *reinterpret_cast<const f32x4*>(p.dpk + o + 4);
        if constexpr (!HAS_T) {
          npq[0] = *reinterpret_cast<const f32x4*>(p.dpq + o);
          npq[1] = *reinterpret_cast<const f32x4*>(p.dpq + o + 4);
        }
      }
    };
    {
      int tid_i = tid;
      asm volatile("" : "+v"(tid_i));
      const int st = tid_i / 24;
      issue(0, st, tid_i - st * 24);
    }
    if constexpr (HAS_T) {
      // landmark rows of the image's three heads: (u qbar) and qbar as swizzled 16-bit rows, zero beyond C; lse_t (log2)
      // (the previous unit's readers are past the second barrier of its last tile: nobody reads these any more)
      // The slot arithmetic below is a function of the thread index alone: left alone, hipcc hoists all of it (~40 registers
      // of offsets and flags) out of the unit loop and parks it in scratch for the duration of the tile loop (160 B / lane,
      // profiles/r05_resources.md).  An opaque copy of the thread index per unit keeps it where it is used -- a few dozen
      // integer instructions once per 13 tiles.
      int tid_u = tid;
      asm volatile("" : "+v"(tid_u));
      f32x4 rb[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid_u + j * (DG_WAVES * 64);        // 6 matrices x 64 rows x 8 chunks = 3072 slots
        const int m = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
        const float* src = (m < 3 ? p.uq : p.qbar) + ((size_t)(b * 3 + (m < 3 ? m : m - 3)) * p.C + min(row, p.C - 1)) * 64 + c * 8;
        // unconditional loads from clamped rows, pinned (`cond ? load : 0` comes back as a predicated load with a full wait
        // behind it: DESIGN section 4, round 3): all eight are in flight together, rows beyond C are zeroed afterwards
        rb[j][0] = *reinterpret_cast<const f32x4*>(src);
        rb[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
        asm volatile("" : "+v"(rb[j][0]), "+v"(rb[j][1]));
      }
      float lsv = p.lse_t[(size_t)(b * 3 + min(tid_u >> 6, 2)) * p.C + min(tid_u & 63, p.C - 1)] * LOG2E;
      asm volatile("" : "+v"(lsv));
      if (!(tid_u < 192 && (tid_u & 63) < p.C)) lsv = INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid_u + j * (DG_WAVES * 64);
        const int m = idx >> 9, row = (idx >> 3) & 63, c = idx & 7;
        const float zr = row < p.C ? 1.f : 0.f;
        const float f[8] = {rb[j][0][0] * zr, rb[j][0][1] * zr, rb[j][0][2] * zr, rb[j][0][3] * zr,
                            rb[j][1][0] * zr, rb[j][1][1] * zr, rb[j][1][2] * zr, rb[j][1][3] * zr};
        sts16((m < 3 ? R1 + m * DG_LMR : R2 + (m - 3) * DG_LMR) + TileL<64>::off(row, c), pack8<E>(f));
      }
      if (tid_u < 192) LS[tid_u] = lsv;
    }
    for (int t = 0; t < ntile; ++t, buf ^= 1) {
      char* const tb = buf ? tile1 : tile0;
      char* const qb = qt0 + buf * DG_QT;
      int tid_t = tid;
      asm volatile("" : "+v"(tid_t));
      const int s_tok = tid_t / 24, s_c = tid_t - s_tok * 24;
      // ---- commit: pooling terms added on the way (one rounding), changed rows written back at once ----
      {
        char* grow = const_cast<char*>(p.d.dy) + ((row0 + tok_of(t, s_tok)) * p.d.ldy + s_c * 8) * 2;
        const bool dp = dup(t, s_tok);
        u32x4 wq = nb[0], wk = nb[1];
        if (pool) {
          float f[8];
          unpack8<E>(wk, f);
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[e] += npk[0][e] * p.pool_inv; f[4 + e] += npk[1][e] * p.pool_inv; }
          wk = pack8<E>(f);
          stg16(dp ? trash_line() : grow + 192 * 2, wk);
          if constexpr (!HAS_T) {
            unpack8<E>(wq, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] += npq[0][e] * p.pool_inv; f[4 + e] += npq[1][e] * p.pool_inv; }
            wq = pack8<E>(f);
            stg16(dp ? trash_line() + 16 : grow, wq);
          }
        }
        sts16(tb + dg_off(s_c >> 3, s_tok, s_c & 7), wq);
        sts16(tb + dg_off(3 + (s_c >> 3), s_tok, s_c & 7), wk);
        sts16(tb + dg_off(6 + (s_c >> 3), s_tok, s_c & 7), nb[2]);
        if constexpr (HAS_T) sts16(qb + dg_off(s_c >> 3, s_tok, s_c & 7), nq);
      }
      // the pooled-q rows of this wave's correction piece: requested BEFORE the next tile's rows, so that waiting for them
      // (the memory counter retires in order) leaves that prefetch in flight
      f32x4 pq4[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#ifdef EA_DGF_NOPOOLREAD
      if constexpr (false) {
#else
      if constexpr (HAS_T && POOL) {
#endif
        const float* src = p.dpq + ((size_t)(b * 3 + fh) * p.L + cell_of(t, 16 * fnt + li)) * 64 + 4 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j) pq4[j] = *reinterpret_cast<const f32x4*>(src + 16 * (2 * fdh + j));
      }
      issue(min(t + 1, ntile - 1), s_tok, s_c);            // unconditional (static operation count; the last tile is fetched twice)
      __syncthreads();
      if constexpr (HAS_T) {
        // ---- dq -= s sum_c t[c,n] (u qbar)_c (+ the pooled-q term) for (head fh, tokens 16 fnt .., channels 32 fdh ..) ----
        const char* R1h = R1 + fh * DG_LMR;
        const char* R2h = R2 + fh * DG_LMR;
        const float* LSh = LS + fh * 64;
        typename E::x8 f1[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f1[ks] = as_x8<E>(lds16(qb + dg_off(fh, 16 * fnt + li, g * 2 + ks)));
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          uint32_t pw[4];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int ct = 2 * kk + c2;
            f32x4 tt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) tt = E::mma(as_x8<E>(lds16(R2h + TileL<64>::off(ct * 16 + li, g * 2 + ks))), f1[ks], tt);
            const f32x4 ls = *reinterpret_cast<const f32x4*>(LSh + ct * 16 + 4 * g);
            pw[2 * c2] = pack2<E>(fast_exp2(tt[0] * p.scale_log2 - ls[0]), fast_exp2(tt[1] * p.scale_log2 - ls[1]));
            pw[2 * c2 + 1] = pack2<E>(fast_exp2(tt[2] * p.scale_log2 - ls[2]), fast_exp2(tt[3] * p.scale_log2 - ls[3]));
          }
          const u32x4 p1 = {pw[0], pw[1], pw[2], pw[3]};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const char* r1 = R1h + (32 * kk) * 128 + (fdh ? lo.tr[2 + j] : lo.tr[j]);   // (static indices: a run-time index would park the array in scratch)
            acc[j] = E::mma(as_x8<E>(E::tr4(r1), E::tr4(r1 + 16 * 128)), as_x8<E>(p1), acc[j]);
          }
        }
        // lane (g, li): channels 16 dt + 4 g .. + 3 of token 16 fnt + li -> 8-byte read-modify-write of the staged dq row
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int dt = 2 * fdh + j;
          char* a = tb + dg_off(fh, 16 * fnt + li, 2 * dt + (g >> 1)) + (g & 1) * 8;
          const u32x2 old = *reinterpret_cast<const u32x2*>(a);
          float o4[4];
          unpack2<E>(old[0], o4[0], o4[1]);
          unpack2<E>(old[1], o4[2], o4[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = o4[r] - acc[j][r] * p.scale + pq4[j][r] * p.pool_inv;
          *reinterpret_cast<u32x2*>(a) = u32x2{pack2<E>(o4[0], o4[1]), pack2<E>(o4[2], o4[3])};
        }
        __syncthreads();
        // corrected dq chunk of this thread's slot back to memory (the weight-gradient pass reads it)
        stg16(dup(t, s_tok) ? trash_line() : const_cast<char*>(p.d.dy) + ((row0 + tok_of(t, s_tok)) * p.d.ldy + s_c * 8) * 2,
              lds16(tb + dg_off(s_c >> 3, s_tok, s_c & 7)));
      }
      // ---- input gradient of the corrected tile ----
      f32x4 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DG_KT; ++ks) {
        const int slab = ks >> 1;
        const typename E::x8 b0 = as_x8<E>(lds16(tb + dg_off(slab, li, 4 * (ks & 1) + g)));
        const typename E::x8 b1 = as_x8<E>(lds16(tb + dg_off(slab, 16 + li, 4 * (ks & 1) + g)));
        acc[0][ks & 1] = E::mma(wr[ks], b0, acc[0][ks & 1]);
        acc[1][ks & 1] = E::mma(wr[ks], b1, acc[1][ks & 1]);
      }
      int li_x = lane;                                     // (opaque: the token arithmetic of the two stores stays in the loop,
      asm volatile("" : "+v"(li_x));                       //  not in registers / scratch across it)
      li_x &= 15;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const size_t row = row0 + tok_of(t, 16 * rt + li_x);
        const bool dpx = dup(t, 16 * rt + li_x);
        const f32x4 v = acc[rt][0] + acc[rt][1];
        const int col = 16 * wave + 4 * g;
        if constexpr (OF32) {
          *reinterpret_cast<f32x4*>(dpx ? trash_line() : p.d.dx + (row * p.d.ldx + col) * 4) = v;
        } else {
          *reinterpret_cast<u32x2*>(dpx ? trash_line() : p.d.dx + (row * p.d.ldx + col) * 2) = u32x2{pack2<E>(v[0], v[1]), pack2<E>(v[2], v[3])};
        }
      }
    }
    if constexpr (HAS_T) __syncthreads();     // the next unit restages the landmark rows: this unit's correction stages are done
  }
  EA_BLKX(p.d, 1);
  EA_BLK(p.d, 1);
}

int dgrad_rs_supported(int K, int NO) { return K == DG_K && NO == DG_NO; }

int dgrad_rs_dispatch(int dtype, const void* dy, const void* w, int w_f32, void* dx, int dx_f32, int rows, long ldy, long ldx,
                      hipStream_t st) {
  if (rows <= 0) return EA_OK;
  DgP p = {};
  p.dy = (const char*)dy; p.w = (const char*)w; p.dx = (char*)dx;
  p.rows = rows; p.ntiles = (rows + DG_TOK - 1) / DG_TOK; p.ldy = ldy; p.ldx = ldx;
  int grid = ea_device_cus();
  if (grid > p.ntiles) grid = p.ntiles;
  const dim3 g((unsigned)grid), b(DG_WAVES * 64);
#define EA_DG_LAUNCH(E_, WF_, OF_)                                                        \
  do {                                                                                    \
    EA_SET_LDS_ONCE((&dgrad_rs_kernel<E_, WF_, OF_>), DG_LDS);                            \
    hipLaunchKernelGGL((dgrad_rs_kernel<E_, WF_, OF_>), g, b, DG_LDS, st, p);             \
  } while (0)
#define EA_DG_SEL(E_)                                                                     \
  do {                                                                                    \
    if (w_f32) { if (dx_f32) EA_DG_LAUNCH(E_, true, true); else EA_DG_LAUNCH(E_, true, false); }   \
    else { if (dx_f32) EA_DG_LAUNCH(E_, false, true); else EA_DG_LAUNCH(E_, false, false); }       \
  } while (0)
  if (dtype == EA_BF16) EA_DG_SEL(BF16);
  else if (dtype == EA_F16) EA_DG_SEL(F16);
  else return EA_E_BADARG;
#undef EA_DG_SEL
#undef EA_DG_LAUNCH
  return (int)hipGetLastError();
}

int dgrad_fin_dispatch(int dtype, const DgFinP& p0, int w_f32, int dx_f32, bool has_t, int pr, hipStream_t st) {
  DgFinP p = p0;
  if (p.d.rows <= 0) return EA_OK;
  int grid = ea_device_cus();
  if (grid > p.nunits) grid = p.nunits;
  const dim3 g((unsigned)grid), b(DG_WAVES * 64);
#ifdef EA_PROFILE
  ProfReport rep;
  p.d.prof = rep.arm(st, "dgrad_fin", pr);
#endif
#define EA_DF_LAUNCH(E_, WF_, OF_, T_, P_, R_)                                                    \
  do {                                                                                            \
    EA_SET_LDS_ONCE((&dgrad_fin_kernel<E_, WF_, OF_, T_, P_, R_>), DG_FIN_LDS);                   \
    hipLaunchKernelGGL((dgrad_fin_kernel<E_, WF_, OF_, T_, P_, R_>), g, b, DG_FIN_LDS, st, p);    \
  } while (0)
#define EA_DF_SEL3(E_, WF_, OF_, T_)                                                          \
  do {                                                                                        \
    if (pr == 4) EA_DF_LAUNCH(E_, WF_, OF_, T_, true, 4);                                     \
    else if (pr == 2) EA_DF_LAUNCH(E_, WF_, OF_, T_, true, 2);                                \
    else EA_DF_LAUNCH(E_, WF_, OF_, T_, true, 0);                                             \
  } while (0)
#define EA_DF_SEL2(E_, WF_, OF_)                                                              \
  do {                                                                                        \
    if (has_t && p.dpq) EA_DF_SEL3(E_, WF_, OF_, true);                                       \
    else if (has_t) EA_DF_LAUNCH(E_, WF_, OF_, true, false, 0);                               \
    else EA_DF_SEL3(E_, WF_, OF_, false);                                                     \
  } while (0)
#define EA_DF_SEL(E_)                                                                     \
  do {                                                                                    \
    if (w_f32) { if (dx_f32) EA_DF_SEL2(E_, true, true); else EA_DF_SEL2(E_, true, false); }   \
    else { if (dx_f32) EA_DF_SEL2(E_, false, true); else EA_DF_SEL2(E_, false, false); }       \
  } while (0)
  if (dtype == EA_BF16) EA_DF_SEL(BF16);
  else if (dtype == EA_F16) EA_DF_SEL(F16);
  else return EA_E_BADARG;
#undef EA_DF_SEL
#undef EA_DF_SEL2
#undef EA_DF_SEL3
#undef EA_DF_LAUNCH
  return (int)hipGetLastError();
}

// host side of ea_linear_dgrad_finish: three heads of 64 channels, images of gh x gw tokens
int dgrad_fin_launch(int dtype, const void* dqkv, long ldy, const void* qkv, long ldq, const void* w, int w_f32, void* dx, int dx_f32,
                     long ldx, int B, int gh, int gw, int pool_r, int C, float scale, const float* qbar, const float* uq,
                     const float* lse_t, const float* dpq, const float* dpk, hipStream_t st) {
  DgFinP p = {};
  p.d.dy = (const char*)dqkv; p.d.w = (const char*)w; p.d.dx = (char*)dx;
  p.ntok = gh * gw;
  p.d.rows = B * p.ntok; p.d.ntiles = 0; p.d.ldy = ldy; p.d.ldx = ldx;
  p.qkv = (const char*)qkv; p.ldq = ldq;
  p.qbar = qbar; p.uq = uq; p.lse_t = lse_t; p.dpq = dpq; p.dpk = dpk;
  p.gw = gw; p.pool_r = dpq ? pool_r : 1; p.cw = gw / p.pool_r; p.L = (gh / p.pool_r) * p.cw; p.C = C;
  p.pool_inv = 1.f / (float)(p.pool_r * p.pool_r); p.scale = scale; p.scale_log2 = scale * LOG2E;
  p.m_gw = (unsigned)((1ull << 32) / (unsigned)gw) + 1u; p.m_r = (unsigned)((1ull << 32) / (unsigned)p.pool_r) + 1u;
  // token ranges per image: about one unit per CU, whole 32-token tiles
  const int cus = ea_device_cus();
  int splits = (cus + B - 1) / B;
  if (splits < 1) splits = 1;
  int tps = ((p.ntok + splits - 1) / splits + DG_TOK - 1) / DG_TOK * DG_TOK;
  if (tps < 2 * DG_TOK) tps = 2 * DG_TOK;
  p.tps = tps;
  p.splits = (p.ntok + tps - 1) / tps;
  // 2 x 2 / 4 x 4 pooling cells: cell-major tiles, units of whole cells (EA_DGF_CELLS=0: the row-major order)
  static const bool cells_on = !(getenv("EA_DGF_CELLS") && getenv("EA_DGF_CELLS")[0] == '0');
  int pr = 0;
  if (dpq && cells_on && (pool_r == 2 || pool_r == 4) && gh % pool_r == 0 && gw % pool_r == 0) {
    pr = pool_r;
    const int cpt = DG_TOK / (pr * pr);
    int cps = ((p.L + splits - 1) / splits + cpt - 1) / cpt * cpt;
    if (cps < 2 * cpt) cps = 2 * cpt;
    p.cps = cps;
    p.splits = (p.L + cps - 1) / cps;
    p.m_cw = (unsigned)((1ull << 32) / (unsigned)p.cw) + 1u;
  }
  p.nunits = B * p.splits;
  return dgrad_fin_dispatch(dtype, p, w_f32, dx_f32, uq != nullptr, pr, st);
}

}  // namespace ea
