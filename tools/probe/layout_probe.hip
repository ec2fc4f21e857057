// dev tool: check the round-3 tile layout helpers on hardware (LaneOff2::tr mapping, quad_transpose)
#include "../../efficient-attention_amd/csrc/ea_common.h"
#include <stdio.h>
using namespace ea;
__global__ void k(float* out, int* bad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15;
  // tile [16 rows][64 ch] bf16: value = row * 64 + ch (exact in bf16 up to 256 -> use row*4 + ... keep small): row + ch/64.
  for (int idx = lane; idx < 16 * 8; idx += 64) {
    const int row = idx / 8, c = idx % 8;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)(row * 8 + (c * 8 + i) / 8) + 0.0f;   // coarse: identifies row and chunk
    // exact identification: store channel index directly, row in another tile
    for (int i = 0; i < 8; ++i) f[i] = (float)(c * 8 + i);
    sts16(smem + lds_off2<64>(row, c), pack8<BF16>(f));
    for (int i = 0; i < 8; ++i) f[i] = (float)row;
    sts16(smem + 4096 + lds_off2<64>(row, c), pack8<BF16>(f));
  }
  __syncthreads();
  LaneOff2<64> lo; lo.init(lane);
  int nbad = 0;
  for (int dt = 0; dt < 4; ++dt) {
    u32x2 v = BF16::tr4(smem + lo.tr[dt]);
    u32x2 w = BF16::tr4(smem + 4096 + lo.tr[dt]);
    float ch[4], rw[4];
    unpack2<BF16>(v[0], ch[0], ch[1]); unpack2<BF16>(v[1], ch[2], ch[3]);
    unpack2<BF16>(w[0], rw[0], rw[1]); unpack2<BF16>(w[1], rw[2], rw[3]);
    for (int a = 0; a < 4; ++a) {
      if ((int)ch[a] != 16 * dt + li) ++nbad;        // output lane li <-> channel 16 dt + li
      if ((int)rw[a] != 4 * g + a) ++nbad;           // element a <-> row 4 g + a
    }
  }
  // quad transpose: acc[dt][r] = channel 16 dt + 4 g + r (+ 100 * li)
  f32x4 acc[4];
  for (int dt = 0; dt < 4; ++dt) for (int r = 0; r < 4; ++r) acc[dt][r] = (float)(16 * dt + 4 * g + r + 100 * li);
  float f[16];
  quad_transpose_f32(acc, f);
  if (lane == 17) { for (int dt = 0; dt < 4; ++dt) for (int r = 0; r < 4; ++r) out[2048 + 4 * dt + r] = acc[dt][r]; }
  for (int j = 0; j < 16; ++j) if ((int)f[j] != 16 * g + j + 100 * li) ++nbad;
  u32x4 o0, o1;
  quad_transpose_pack<BF16>(acc, 1.f, o0, o1);
  float h[16];
  unpack8<BF16>(o0, h); unpack8<BF16>(o1, h + 8);
  for (int j = 0; j < 16; ++j) { const float want = BF16::to_f(BF16::from_f((float)(16 * g + j + 100 * li))); if (h[j] != want) ++nbad; }
  bad[lane] = nbad;
  for (int j = 0; j < 16; ++j) out[lane * 16 + j] = f[j];
}
int main() {
  float* d; int* b; hipMalloc(&d, 4096 * 4); hipMalloc(&b, 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, b);
  int hb[64]; float hf[2100];
  hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost); hipMemcpy(hf, d, sizeof(hf), hipMemcpyDeviceToHost);
  int tot = 0; for (int i = 0; i < 64; ++i) tot += hb[i];
  printf("layout_probe: %d mismatches\n", tot);
  printf("acc lane17:"); for (int j = 0; j < 16; ++j) printf(" %d", (int)hf[2048 + j]); printf("\n");
  for (int l : {0, 17, 35, 63}) { printf("lane %d:", l); for (int j = 0; j < 16; ++j) printf(" %d", (int)hf[l * 16 + j]); printf("\n"); }
  return tot != 0;
}
