// Hardware-assumption probe for gfx950: verifies the lane layouts every kernel in this
// directory is built on.  Built by __graft_entry__.build() into csrc/../lib/probe_primitives
// and run by tests/test_gpu_primitives.py (prints "PROBE ALL OK" on success).
//
//  (1) v_mfma_f32_16x16x32_bf16 operand/result layout:
//        A: lane l holds A[i = l&15][k = 8*(l>>4) .. +7]
//        B: lane l holds B[k = 8*(l>>4) .. +7][j = l&15]
//        D: lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3
//  (2) ds_read_b64_tr_b16: within each 16-lane group, lane i receives element e (0..3) =
//        element (i & 3) of the 8-byte word addressed by supplier lane 4*e + (i >> 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void probe_mfma(const unsigned short* A, const unsigned short* B, float* Dm) {
  // A [16][32] row-major bf16 bits, B [32][16] row-major, D [16][16]
  int l = threadIdx.x, g = l >> 4, i = l & 15;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    unsigned short ua = A[i * 32 + 8 * g + t];
    unsigned short ub = B[(8 * g + t) * 16 + i];
    a[t] = __builtin_bit_cast(__bf16, ua);
    b[t] = __builtin_bit_cast(__bf16, ub);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Dm[(4 * g + r) * 16 + i] = acc[r];
}

__global__ void probe_tr(const int* lane_off, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l = threadIdx.x;
  for (int t = l; t < 4096; t += 64) lds[t] = (short)t;
  __syncthreads();
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + lane_off[l]));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}

static float bf16_to_f(unsigned short u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return f; }

int main() {
  int bad = 0;
  // ---------------- (1) MFMA ----------------
  std::vector<unsigned short> A(16 * 32), B(32 * 16);
  srand(7);
  for (auto& x : A) { float f = (float)((rand() % 17) - 8); uint32_t u; memcpy(&u, &f, 4); x = u >> 16; }
  for (auto& x : B) { float f = (float)((rand() % 13) - 6); uint32_t u; memcpy(&u, &f, 4); x = u >> 16; }
  unsigned short *dA, *dB; float* dD;
  CHECK(hipMalloc(&dA, A.size() * 2)); CHECK(hipMalloc(&dB, B.size() * 2)); CHECK(hipMalloc(&dD, 256 * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  probe_mfma<<<1, 64>>>(dA, dB, dD);
  CHECK(hipDeviceSynchronize());
  std::vector<float> D(256);
  CHECK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float ref = 0.f;
    for (int k = 0; k < 32; ++k) ref += bf16_to_f(A[i * 32 + k]) * bf16_to_f(B[k * 16 + j]);
    if (ref != D[i * 16 + j]) { if (bad < 5) printf("MFMA mismatch D[%d][%d] = %f ref %f\n", i, j, D[i * 16 + j], ref); ++bad; }
  }
  printf("probe mfma_f32_16x16x32_bf16 layout: %s\n", bad ? "FAIL" : "ok");

  // ---------------- (2) tr read ----------------
  int bad_tr = 0;
  std::vector<int> off(64);
  int* dOff; short* dOut;
  CHECK(hipMalloc(&dOff, 64 * 4)); CHECK(hipMalloc(&dOut, 256 * 2));
  std::vector<short> out(256);
  // pattern a: contiguous 8 B per lane; pattern b: scattered rows (row stride 64 elems)
  for (int pat = 0; pat < 2; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) off[l] = l * 4;
      else { int g = l >> 4, s = l & 15; off[l] = (4 * g + (s >> 2)) * 64 + 16 * (s & 3) + 4 * (g & 1); }
    }
    CHECK(hipMemcpy(dOff, off.data(), 64 * 4, hipMemcpyHostToDevice));
    probe_tr<<<1, 64>>>(dOff, dOut);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(out.data(), dOut, 256 * 2, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      int grp = l >> 4, i = l & 15;
      int supplier = 16 * grp + 4 * e + (i >> 2);
      int expect = off[supplier] + (i & 3);
      if (out[l * 4 + e] != (short)expect) {
        if (bad_tr < 8) printf("TR mismatch pat %d lane %d elem %d: got %d expect %d\n", pat, l, e, out[l * 4 + e], expect);
        ++bad_tr;
      }
    }
    if (bad_tr && pat == 0) {
      printf("raw pattern-0 dump (lane: e0 e1 e2 e3):\n");
      for (int l = 0; l < 64; ++l) printf("  %2d: %4d %4d %4d %4d\n", l, out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]);
    }
  }
  printf("probe ds_read_b64_tr_b16 mapping: %s\n", bad_tr ? "FAIL" : "ok");
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch %s CUs %d LDS/block %zu clock %d kHz memclk %d kHz buswidth %d\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.sharedMemPerBlock, prop.clockRate, prop.memoryClockRate, prop.memoryBusWidth);
  if (!bad && !bad_tr) printf("PROBE ALL OK\n");
  return (bad || bad_tr) ? 1 : 0;
}
