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
#include "ea_common.h"

namespace ea {

struct DgP {
  const char* dy;       // [rows, 576] element type, row stride ldy elements
  const char* w;        // [576, 192] fp32 master weight (WF32) or element type
  char* dx;             // [rows, 192] fp32 (OF32) or element type, row stride ldx elements
  int rows, ntiles;
  long ldy, ldx;
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
    if (t + (int)gridDim.x < p.ntiles) issue(t + gridDim.x);
    __syncthreads();
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

}  // namespace ea
