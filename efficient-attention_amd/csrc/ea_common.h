// ea_common.h -- device-side building blocks shared by every kernel of libea_hip.so.
//
// Everything here is written for gfx950 only: 64-lane wavefronts, v_mfma_f32_16x16x32_{bf16,f16},
// ds_read_b64_tr_b16.  Lane layouts are verified on hardware by probe_primitives.hip.
//
// MFMA conventions used throughout (g = lane >> 4, li = lane & 15):
//   A operand: lane holds A[row = li][k-slot 8g .. 8g+7]     (8 elements, 16 bytes)
//   B operand: lane holds B[k-slot 8g .. 8g+7][col = li]
//   D result : lane holds D[row = 4g + r][col = li], r = 0..3
// The contraction index is free to be permuted as long as A and B agree, which is what lets
// score tiles computed as S^T[key][query] (row = key, col = query) be re-used, register for
// register, as the B operand of the P.V product: a lane's four D values of two adjacent 16-key
// tiles are exactly its eight k-slots of the next MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ea_hip.h"

// ---- dev-only phase profiling (-DEA_PROFILE, tools/build_prof_lib.sh): thread 0 of workgroup 0
// stamps s_memtime into p.prof[i]; the dispatcher prints the deltas.  Compiled out of the product.
#ifdef EA_PROFILE
#include <stdio.h>
#include <algorithm>
#include <vector>
#define EA_STAMP(p, i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (p).prof) (p).prof[i] = (long long)__builtin_readcyclecounter(); } while (0)
// per-workgroup begin / end on the chip-wide 100 MHz clock
#define EA_BLK(p, e) do { if (threadIdx.x == 0 && (p).prof && blockIdx.x < 8192) (p).prof[128 + 2 * blockIdx.x + (e)] = (long long)wall_clock64(); } while (0)
// two more per-workgroup stamps (after the prologue, after the main loop)
#define EA_BLKX(p, e) do { if (threadIdx.x == 0 && (p).prof && blockIdx.x < 8192) (p).prof[128 + 2 * 8192 + 2 * blockIdx.x + (e)] = (long long)wall_clock64(); } while (0)
namespace ea {
struct ProfReport {
  static constexpr int NW = 128 + 4 * 8192;
  long long* d = nullptr; hipStream_t st; const char* name; int mode;
  long long* arm(hipStream_t s, const char* nm, int md) {
    static long long* buf = nullptr;
    if (!buf) hipMalloc(&buf, NW * sizeof(long long));
    hipMemsetAsync(buf, 0, NW * sizeof(long long), s);
    d = buf; st = s; name = nm; mode = md;
    return buf;
  }
  ~ProfReport() {
    if (!d) return;
    std::vector<long long> h(NW);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, NW * sizeof(long long), hipMemcpyDeviceToHost);
    fprintf(stderr, "%s mode %d:", name, mode);
    long long prev = h[0];
    for (int i = 1; i < 128; ++i) if (h[i]) { fprintf(stderr, " [%d]%lld", i, h[i] - prev); prev = h[i]; }
    fprintf(stderr, " total %lld\n", prev - h[0]);
    std::vector<long long> b, e, dur;
    for (int i = 0; i < 8192; ++i) if (h[128 + 2 * i] && h[129 + 2 * i]) { b.push_back(h[128 + 2 * i]); e.push_back(h[129 + 2 * i]); dur.push_back(h[129 + 2 * i] - h[128 + 2 * i]); }
    if (!b.empty()) {
      const long long t0 = *std::min_element(b.begin(), b.end());
      std::vector<long long> bs(b), es(e), ds(dur);
      for (auto& x : bs) x -= t0; for (auto& x : es) x -= t0;
      std::sort(bs.begin(), bs.end()); std::sort(es.begin(), es.end()); std::sort(ds.begin(), ds.end());
      const size_t n = bs.size();
      fprintf(stderr, "  %zu blocks (10 ns ticks): start p50 %lld p90 %lld max %lld | dur min %lld p50 %lld p90 %lld max %lld | end p50 %lld max %lld\n",
              n, bs[n / 2], bs[n * 9 / 10], bs[n - 1], ds[0], ds[n / 2], ds[n * 9 / 10], ds[n - 1], es[n / 2], es[n - 1]);
      {
        std::vector<long long> pro, epi;
        for (int i = 0; i < 8192; ++i) {
          const long long b0 = h[128 + 2 * i], e0 = h[129 + 2 * i], x0 = h[128 + 2 * 8192 + 2 * i], x1 = h[129 + 2 * 8192 + 2 * i];
          if (b0 && e0 && x0 && x1) { pro.push_back(x0 - b0); epi.push_back(e0 - x1); }
        }
        if (!pro.empty()) {
          std::sort(pro.begin(), pro.end()); std::sort(epi.begin(), epi.end());
          const size_t m = pro.size();
          fprintf(stderr, "  prologue p10 %lld p50 %lld p90 %lld | epilogue p10 %lld p50 %lld p90 %lld\n", pro[m / 10], pro[m / 2], pro[m * 9 / 10], epi[m / 10], epi[m / 2], epi[m * 9 / 10]);
        }
      }
      fprintf(stderr, "  start deciles:");
      for (int q = 1; q <= 10; ++q) fprintf(stderr, " %lld", bs[(n * q) / 10 - 1]);
      fprintf(stderr, " | dur deciles:");
      for (int q = 1; q <= 10; ++q) fprintf(stderr, " %lld", ds[(n * q) / 10 - 1]);
      fprintf(stderr, " | end deciles:");
      for (int q = 1; q <= 10; ++q) fprintf(stderr, " %lld", es[(n * q) / 10 - 1]);
      fprintf(stderr, "\n");
    }
  }
};
}  // namespace ea
#else
#define EA_STAMP(p, i) do { } while (0)
#define EA_BLK(p, e) do { } while (0)
#define EA_BLKX(p, e) do { } while (0)
#endif


namespace ea {

// ---- per-device launch state.  hipFuncSetAttribute acts on the CURRENT device's copy of a kernel and the CU
// count is a device property: a process that drives several GPUs must keep both per device.
constexpr int EA_MAX_DEV = 64;
inline int ea_cur_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EA_MAX_DEV) dev = 0;
  return dev;
}
inline int ea_device_cus() {
  static int n[EA_MAX_DEV] = {};
  const int dev = ea_cur_device();
  if (!n[dev]) {
    hipDeviceProp_t prop;
    int v = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
    n[dev] = v > 0 ? v : 256;
  }
  return n[dev];
}
// once per (call site, device): raise the dynamic-LDS limit of kernel `fn`
#define EA_SET_LDS_ONCE(fn, bytes)                                                                              \
  do {                                                                                                         \
    static bool ea_done_[ea::EA_MAX_DEV] = {};                                                                 \
    const int ea_dev_ = ea::ea_cur_device();                                                                   \
    if (!ea_done_[ea_dev_]) {                                                                                  \
      hipError_t ea_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));       \
      if (ea_e_ != hipSuccess) return (int)ea_e_;                                                              \
      ea_done_[ea_dev_] = true;                                                                                \
    }                                                                                                          \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;   // arithmetic on it maps to v_pk_*_f32
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define EA_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define EA_DEV __device__ __forceinline__

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float MASK_FILL = -5e4f;       // finite mask value of local/EVA (eva.py:139)

// ------------------------------------------------------------------------------------------
// element traits
// ------------------------------------------------------------------------------------------
struct BF16 {
  typedef __bf16 T;
  typedef __attribute__((ext_vector_type(8))) __bf16 x8;
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 x4v;
  static EA_DEV f32x4 mma(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static EA_DEV u32x2 tr4(const char* lds) {     // ds_read_b64_tr_b16
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(EA_LDS(x4v, lds)));
  }
  static EA_DEV float to_f(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
  static EA_DEV uint16_t from_f(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
};

struct F16 {
  typedef _Float16 T;
  typedef __attribute__((ext_vector_type(8))) _Float16 x8;
  typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 x4v;
  static EA_DEV f32x4 mma(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static EA_DEV u32x2 tr4(const char* lds) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16(EA_LDS(x4v, lds)));
  }
  static EA_DEV float to_f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  static EA_DEV uint16_t from_f(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
};

// two floats -> one packed 16-bit pair, round-to-nearest-even, as ONE v_cvt_pk_{bf16,f16}_f32: the vector conversion is
// what selects the packed instruction -- two scalar conversions plus shift-and-or compile to four instructions per pair
// (found in round 3 in the softmax forward, whose VALU work bounds it: 16 conversions + 8 shifts + 8 ors per 64-key chunk).
template <typename E> EA_DEV uint32_t pack2(float lo, float hi) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef typename E::T ex2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ex2_));
}
template <typename E> EA_DEV void unpack2(uint32_t w, float& lo, float& hi) {
  lo = E::to_f((uint16_t)(w & 0xffffu));
  hi = E::to_f((uint16_t)(w >> 16));
}
// eight consecutive elements (16 B) <-> eight floats
template <typename E> EA_DEV void unpack8(u32x4 w, float* f) {
  unpack2<E>(w[0], f[0], f[1]); unpack2<E>(w[1], f[2], f[3]);
  unpack2<E>(w[2], f[4], f[5]); unpack2<E>(w[3], f[6], f[7]);
}
template <typename E> EA_DEV u32x4 pack8(const float* f) {
  u32x4 w;
  w[0] = pack2<E>(f[0], f[1]); w[1] = pack2<E>(f[2], f[3]);
  w[2] = pack2<E>(f[4], f[5]); w[3] = pack2<E>(f[6], f[7]);
  return w;
}
template <typename E> EA_DEV typename E::x8 as_x8(u32x4 w) { return __builtin_bit_cast(typename E::x8, w); }
template <typename E> EA_DEV typename E::x8 as_x8(u32x2 lo, u32x2 hi) {
  u32x4 w; w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
  return __builtin_bit_cast(typename E::x8, w);
}

EA_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
EA_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// reduce over the four lanes {li, li+16, li+32, li+48} that share a query/key column.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane halves between two VGPRs in
// the VALU (no LDS crossbar round trip like ds_bpermute): swap(v, v) leaves one register holding
// the even rows' values and the other the odd rows', so each level is one swap + one add/max.
// (Inline asm: with identical inputs the builtin's two results get folded together by hipcc; the
// s_nop covers the VALU-write -> permlane-read hazard.)
EA_DEV void lane_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
EA_DEV void lane_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
EA_DEV float quad_max(float v) {
  float a = v, b = v;
  lane_swap16(a, b);
  a = fmaxf(a, b); b = a;
  lane_swap32(a, b);
  return fmaxf(a, b);
}
EA_DEV float quad_sum(float v) {
  float a = v, b = v;
  lane_swap16(a, b);
  a = a + b; b = a;
  lane_swap32(a, b);
  return a + b;
}

// ---- lane reductions without the LDS pipe (round 3) ----
// __shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip per step, in kernels whose LDS pipe is already the busy one.
// Within a DPP row (16 lanes) the same sums are row rotations / quad permutes / a half-row mirror fused into the add as
// DPP operands; across the four rows they are the two v_permlane swaps of quad_sum / quad_max above.
template <int CTRL> EA_DEV float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// over the G consecutive lanes of an aligned group (G = 4, 8, 16); every lane gets the result
template <int G> EA_DEV float group_sum(float v) {
  static_assert(G == 4 || G == 8 || G == 16, "group_sum: 4, 8 or 16 lanes");
  v += dpp_mov<0xB1>(v);                              // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);                              // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_mov<0x141>(v);                 // row_half_mirror: the other quad of the 8
  if (G >= 16) v += dpp_mov<0x128>(v);                // row_ror:8: the other half of the row
  return v;
}
// over the lanes with equal (lane % S) of the whole wave (S = 4, 8, 16): strides S .. 8 inside the rows, then the rows
template <int S> EA_DEV float stride_sum(float v) {
  static_assert(S == 4 || S == 8 || S == 16, "stride_sum: stride 4, 8 or 16");
  if (S <= 8) v += dpp_mov<0x128>(v);                 // row_ror:8
  if (S <= 4) v += dpp_mov<0x124>(v);                 // row_ror:4
  return quad_sum(v);
}
template <int S> EA_DEV float stride_max(float v) {
  static_assert(S == 4 || S == 8 || S == 16, "stride_max: stride 4, 8 or 16");
  if (S <= 8) v = fmaxf(v, dpp_mov<0x128>(v));
  if (S <= 4) v = fmaxf(v, dpp_mov<0x124>(v));
  return quad_max(v);
}
EA_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
EA_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Destination of stores that must not land anywhere (rows past the end of a slice): a store issued unconditionally to
// `valid ? dst : ea_trash_line()` keeps the number of memory operations in a loop static, so the compiler can wait for
// the prefetched loads alone (`s_waitcnt vmcnt(n_stores)`) instead of for everything (`vmcnt(0)`, which also waits for the
// stores of the previous tile).  64 bytes per thread; nobody reads it.
static __device__ __attribute__((aligned(64))) char ea_trash[512 * 64];
EA_DEV char* ea_trash_line() { return ea_trash + threadIdx.x * 64; }

#ifdef EA_NT_LOADS
EA_DEV u32x4 ldg16(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
#else
EA_DEV u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
#endif
#ifdef EA_NT_STORES
EA_DEV void stg16(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
#else
EA_DEV void stg16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
#endif
EA_DEV u32x4 lds16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
EA_DEV void sts16(char* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// ------------------------------------------------------------------------------------------
// token geometry: (window | chunk, slot) -> token index, -1 outside the sequence / grid
// (attn_utils.py:155-166 / 190-210 as address arithmetic instead of pad + as_strided copies)
// ------------------------------------------------------------------------------------------
struct Geo {
  int N, attn2d, gh, gw;
};
EA_DEV int part_token(const Geo& G, int part, int slot, int side, int ext) {
  if (G.attn2d) {
    const int t = side + 2 * ext;
    const int per_row = G.gw / side;
    const int p1 = part / per_row, p2 = part - p1 * per_row;
    const int i = slot / t, j = slot - i * t;
    const int y = p1 * side - ext + i, x = p2 * side - ext + j;
    return (y >= 0 && y < G.gh && x >= 0 && x < G.gw) ? y * G.gw + x : -1;
  }
  const int tok = part * side - ext + slot;
  return (tok >= 0 && tok < G.N) ? tok : -1;
}

// LDS row-major tile [rows][D] with the 16-byte chunk index XOR-swizzled by the row, so that
// both the b128 operand reads (16 lanes = 16 rows, same chunk) and the b64 transpose reads are
// spread over the banks (cdna_hip_programming.md T2).
template <int D> EA_DEV int lds_off(int row, int chunk16) {
  constexpr int CPR = D / 8;                       // 16-byte chunks per row
  constexpr int SW = CPR >= 8 ? 7 : CPR - 1;
  return row * (D * 2) + ((chunk16 ^ (row & SW)) << 4);
}

// Per-lane byte offsets into a swizzled LDS tile, computed ONCE per kernel: every tile base used by
// the kernels is a multiple of 16 rows, and the swizzle only looks at (row & 7), so the offset of a
// lane's operand inside "its" 16-row tile is a lane constant -- the hot loops then address LDS with
// one scalar-plus-vector add instead of re-deriving the shift/xor chain per read.
template <int D> struct LaneOff {
  static constexpr int ROWB = D * 2, CPR = D / 8, KS = D / 32, DT = D / 16, DQ = D / 4;
  static constexpr int SW = CPR >= 8 ? 7 : CPR - 1;
  int plain[KS];   // A/B operand chunk (row = lane & 15, k-step ks) of a 16-row tile
  int tr[DT];      // ds_read_b64_tr_b16 source (row = 4g + (li >> 2), channel tile dt)
  EA_DEV void init(int lane) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) plain[ks] = li * ROWB + (((g * KS + ks) ^ (li & SW)) << 4);
    const int r = 4 * g + (li >> 2);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int colb = (DQ * (li & 3) + 4 * dt) * 2;
      tr[dt] = r * ROWB + ((((colb >> 4)) ^ (r & SW)) << 4) + (colb & 15);
    }
  }
};

// ------------------------------------------------------------------------------------------
// Round 3: bank-conflict-free tile layout (D = 64: 128-byte rows).  Measured with tools/probe/lds_probe.hip:
//   * ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS):
//     with the (row & 7) chunk swizzle of lds_off<> two of the four groups are 2-way conflicted (6.1 instead of 4.0
//     cycles per wave-instruction);
//   * the ds_read_b64_tr_b16 pattern of LaneOff<>::tr -- a row's four lanes reading 8-byte pieces 32 bytes apart -- puts
//     the 32 lanes of a group on 8 of the 32 8-byte slots of the bank space whatever the 16-byte swizzle: 4-way
//     conflicts, 7.0 instead of 2.0 cycles.  (SQ_LDS_BANK_CONFLICT was 35-55 % of SQ_LDS_IDX_ACTIVE in every kernel.)
// Fix: (a) chunk swizzle phi2(row) = ((row >> 1) & 3) << 1 | ((row ^ (row >> 3)) & 1), conflict-free for the b128
// row-operand reads, for 16-byte row stores and for (b) the transpose reads re-shaped so that the four lanes of a row
// read 32 CONTIGUOUS bytes (channels 16 dt + 4 (li & 3) ..).  With (b) the MFMA that consumes the fragment produces
// D rows <-> channels 16 dt + 4 g + r: a lane owns four 4-channel pieces 16 channels apart instead of 16 contiguous
// channels; quad_transpose() below moves the pieces between the four lanes of a token (8 v_permlane swaps on packed
// data) so that stores stay 32 contiguous bytes per lane.
EA_DEV int phi2(int row) { return (((row >> 1) & 3) << 1) | ((row ^ (row >> 3)) & 1); }
template <int D> EA_DEV int lds_off2(int row, int chunk16) {
  static_assert(D == 64, "lds_off2: 128-byte rows only");
  return row * (D * 2) + ((chunk16 ^ phi2(row)) << 4);
}
template <int D> struct LaneOff2 {
  static constexpr int ROWB = D * 2, KS = D / 32, DT = D / 16;
  int plain[KS];   // A/B operand chunk (row = lane & 15, k-step ks) of a 16-row tile
  int tr[DT];      // ds_read_b64_tr_b16 source: row 4g + (li >> 2), channels 16 dt + 4 (li & 3) ..
  EA_DEV void init(int lane) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) plain[ks] = lds_off2<D>(li, g * KS + ks);
    const int r = 4 * g + (li >> 2);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) tr[dt] = lds_off2<D>(r, 2 * dt + ((li & 3) >> 1)) + 8 * (li & 1);
  }
};
// x[s][w]: W registers per piece, piece s of lane-row g = channels 16 s + 4 g + r  ->  piece s = channels 16 g + 4 s + r
// (4 x 4 transpose of the pieces across the four lanes li, li + 16, li + 32, li + 48 of a token)
// One asm statement per stage, opening with s_nop 15: hipcc pads nothing inside inline asm (cdna_hip_programming.md 5.7), and
// the operands are often MFMA results -- an XDL write needs up to 11 wait states before a VALU read of an 8-pass result
// (measured: without the pad, accumulators handed straight to the swaps came back as NaN in two channels).
template <int W> EA_DEV void quad_transpose(uint32_t (&x)[4][W]);
template <> EA_DEV void quad_transpose<2>(uint32_t (&x)[4][2]) {
  asm volatile("s_nop 15\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %2, %6\n\t"
               "v_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %3, %7"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %4, %6\n\t"
               "v_permlane16_swap_b32 %1, %3\n\tv_permlane16_swap_b32 %5, %7"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]));
}
template <> EA_DEV void quad_transpose<4>(uint32_t (&x)[4][4]) {
  asm volatile("s_nop 15\n\t"
               "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %4, %12\n\t"
               "v_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %5, %13\n\t"
               "v_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %6, %14\n\t"
               "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %7, %15"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[0][3]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2]), "+v"(x[1][3]),
                 "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[2][2]), "+v"(x[2][3]), "+v"(x[3][0]), "+v"(x[3][1]), "+v"(x[3][2]), "+v"(x[3][3]));
  asm volatile("s_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %8, %12\n\t"
               "v_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %9, %13\n\t"
               "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %10, %14\n\t"
               "v_permlane16_swap_b32 %3, %7\n\tv_permlane16_swap_b32 %11, %15"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[0][3]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2]), "+v"(x[1][3]),
                 "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[2][2]), "+v"(x[2][3]), "+v"(x[3][0]), "+v"(x[3][1]), "+v"(x[3][2]), "+v"(x[3][3]));
}
// fp32 accumulators acc[dt][r] (new ownership) -> f[16] in the contiguous ownership (channel 16 g + j)
EA_DEV void quad_transpose_f32(const f32x4* acc, float* f) {
  uint32_t x[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = acc[s][r];          // (a bit_cast applied directly to the vector element reads element 0: hipcc 7.2)
      x[s][r] = __builtin_bit_cast(uint32_t, t);
    }
  quad_transpose<4>(x);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) f[4 * s + r] = __builtin_bit_cast(float, x[s][r]);
}
// ... rounded to the element type first (values scaled by `mul`): two u32x4 = the lane's 32 contiguous bytes
template <typename E> EA_DEV void quad_transpose_pack(const f32x4* acc, float mul, u32x4& o0, u32x4& o1) {
  uint32_t x[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    x[s][0] = pack2<E>(acc[s][0] * mul, acc[s][1] * mul);
    x[s][1] = pack2<E>(acc[s][2] * mul, acc[s][3] * mul);
  }
  quad_transpose<2>(x);
  o0 = u32x4{x[0][0], x[0][1], x[1][0], x[1][1]};
  o1 = u32x4{x[2][0], x[2][1], x[3][0], x[3][1]};
}

// token / landmark tiles [rows][D]: the conflict-free round-3 layout for 128-byte rows (ea_common.h), the round-1 one for D = 32.
// NEWTR: the transpose reads hand a lane the channels 16 dt + 4 g + r (pieces), not D/4 contiguous ones.
template <int D> struct TileL {
  static constexpr bool NEWTR = (D == 64);
  static EA_DEV int off(int row, int chunk16) {
    if constexpr (NEWTR) return lds_off2<D>(row, chunk16);
    else return lds_off<D>(row, chunk16);
  }
};
// byte offset of the ds_read_b64_tr_b16 source of lane li (row-group lanes 4a .. 4a+3 read row r) for channel tile dt
template <int D> EA_DEV int tile_tr(int r, int li, int dt) {
  const int colb = TileL<D>::NEWTR ? (16 * dt + 4 * (li & 3)) * 2 : ((D / 4) * (li & 3) + 4 * dt) * 2;
  return TileL<D>::off(r, colb >> 4) + (colb & 15);
}
template <int D> struct LaneOffSel { typedef LaneOff<D> type; };
template <> struct LaneOffSel<64> { typedef LaneOff2<64> type; };
// channel offset (within a [*, D] fp32 row) of accumulator tile dt of lane-row g
template <int D> EA_DEV int acc_chan(int dt, int g) { return TileL<D>::NEWTR ? 16 * dt + 4 * g : (D / 4) * g + 4 * dt; }

}  // namespace ea
