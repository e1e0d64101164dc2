// GPU probe (development aid, not part of the product): operand layout + scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4,
// the fp8 conversion instructions, and the sustained rate of mixed f16 / MX-fp8 MFMA streams on the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OA, int OB>
__global__ void mx_one(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x16* c) {
  const int l = threadIdx.x;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, OA, sa[l], OB, sb[l]);
  c[l] = acc;
}

__global__ void cvt_probe(const float* x, int n, float sc, int* o) {
  const int l = threadIdx.x;
  if (l < n) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(x[l], -x[l], r, false);
    o[l] = r;
    s16x2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, x[l], -x[l], sc, false);
    o[64 + l] = __builtin_bit_cast(int, q);
  }
}

// rate: MODE 0 = 192 f16 MFMAs / iteration (the f16x3 block), 1 = 64 f16 + 32 scaled fp8 (2-unit block), 2 = 96 scaled fp8
template <int MODE>
__global__ __launch_bounds__(512, 2) void rate(int iters, float* out) {
  f32x16 c[8];
  for (int r = 0; r < 8; ++r) for (int q = 0; q < 16; ++q) c[r][q] = 0.f;
  const int l = threadIdx.x;
  f16x8 ah, bh; i32x8 a8, b8;
  for (int q = 0; q < 8; ++q) { ah[q] = (_Float16)(0.001f * (l + q)); bh[q] = (_Float16)(0.002f * (l - q)); a8[q] = 0x38383838 + l + q; b8[q] = 0x30303030 + l * 3 + q; }
  int sa = 127, sb = 127;
  asm volatile("" : "+v"(sa), "+v"(sb));
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 24; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[r], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[r], 0, 0, 0, sa, 0, sb);
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c[r], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[r], 0, 0, 0, sa, 0, sb);
    }
    asm volatile("" : "+v"(ah), "+v"(bh), "+v"(a8), "+v"(b8));
  }
  float s = 0.f;
  for (int r = 0; r < 8; ++r) for (int q = 0; q < 16; ++q) s += c[r][q];
  if (s == 12345.678f) out[0] = s;
}

static double dec8(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  double x;
  if (e == 0) x = ldexp(m / 8.0, -6);
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexp(1.0 + m / 8.0, e - 7);
  return s ? -x : x;
}

template <int OA, int OB>
static int check_layout(const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<int>& SA, const std::vector<int>& SB) {
  // A[i][k] (32 x 64), B[k][j]; SA[l], SB[l]: 4 scale bytes per lane
  std::vector<i32x8> ha(64), hb(64);
  for (int l = 0; l < 64; ++l) {
    const int i = l & 31, g = l >> 5;
    unsigned char* pa = (unsigned char*)&ha[l]; unsigned char* pb = (unsigned char*)&hb[l];
    for (int p = 0; p < 32; ++p) { pa[p] = A[i * 64 + g * 32 + p]; pb[p] = B[(g * 32 + p) * 32 + i]; }
  }
  i32x8 *da, *db; int *dsa, *dsb; f32x16* dc;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 64));
  CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((mx_one<OA, OB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  std::vector<float> hc(64 * 16);
  CK(hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost));
  int bad = 0; double worst = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double ref = 0;
    for (int g = 0; g < 2; ++g) {
      const int ea = (SA[i + 32 * g] >> (8 * OA)) & 255, eb = (SB[j + 32 * g] >> (8 * OB)) & 255;
      double s = 0;
      for (int p = 0; p < 32; ++p) s += dec8(A[i * 64 + g * 32 + p]) * dec8(B[(g * 32 + p) * 32 + j]);
      ref += ldexp(s, ea - 127 + eb - 127);
    }
    const double err = fabs(hc[l * 16 + r] - ref) / (fabs(ref) + 1e-3);
    if (err > worst) worst = err;
    if (err > 1e-5) ++bad;
  }
  printf("layout/scale check opsel_a=%d opsel_b=%d: %d bad of 1024, worst rel err %.2e\n", OA, OB, bad, worst);
  return bad;
}

int main() {
  srand(7);
  std::vector<unsigned char> A(32 * 64), B(64 * 32);
  for (auto& v : A) { do v = rand() & 255; while (((v >> 3) & 15) == 15 && (v & 7) == 7); }
  for (auto& v : B) { do v = rand() & 255; while (((v >> 3) & 15) == 15 && (v & 7) == 7); }
  std::vector<int> SA(64), SB(64);
  for (int l = 0; l < 64; ++l) {
    SA[l] = (120 + rand() % 15) | ((120 + rand() % 15) << 8) | ((120 + rand() % 15) << 16) | ((120 + rand() % 15) << 24);
    SB[l] = (120 + rand() % 15) | ((120 + rand() % 15) << 8) | ((120 + rand() % 15) << 16) | ((120 + rand() % 15) << 24);
  }
  check_layout<0, 0>(A, B, SA, SB); check_layout<1, 2>(A, B, SA, SB); check_layout<3, 1>(A, B, SA, SB); check_layout<2, 3>(A, B, SA, SB);

  // conversions
  const float xs[] = {1.0f, 1.0625f, 1.1875f, 1.07f, 448.f, 464.f, 480.f, 500.f, 1e6f, 0.3f, 0.001953125f, 0.0009765625f, 0.0029296875f, 200.f, 3.0f, 0.f};
  const int n = sizeof(xs) / 4;
  float* dx; int* d_o; CK(hipMalloc(&dx, 256)); CK(hipMalloc(&d_o, 512));
  CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice));
  for (float sc : {1.0f, 2.0f, 0.25f}) {
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, n, sc, d_o);
    int ho[128]; CK(hipMemcpy(ho, d_o, 512, hipMemcpyDeviceToHost));
    printf("cvt (scale arg %.2f):\n", sc);
    for (int i = 0; i < n; ++i)
      printf("  x=%-12g pk_fp8 -> %02x (%g) / %02x (%g)   scalef32_pk -> %02x (%g) / %02x (%g)\n", xs[i], ho[i] & 255, dec8(ho[i] & 255), (ho[i] >> 8) & 255,
             dec8((ho[i] >> 8) & 255), ho[64 + i] & 255, dec8(ho[64 + i] & 255), (ho[64 + i] >> 8) & 255, dec8((ho[64 + i] >> 8) & 255));
  }

  // sustained rate, whole chip, 2 waves per SIMD
  float* dout; CK(hipMalloc(&dout, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      const int iters = 4000;
      CK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(cus), dim3(512), 0, 0, iters, dout);
      if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(512), 0, 0, iters, dout);
      if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(cus), dim3(512), 0, 0, iters, dout);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      // passes per iteration per wave: mode0 192*8, mode1 64*8 + 32*16, mode2 96*16; 2 waves per SIMD
      const double passes = (mode == 0 ? 192 * 8 : mode == 1 ? 64 * 8 + 32 * 16 : 96 * 16) * 2.0 * iters;
      printf("rate mode %d: %.3f ms for %d block-iterations/wave -> %.1f us per 1000 blocks, implied MFMA-busy clock %.0f MHz\n", mode, ms, iters,
             ms * 1e3 / iters * 1000 / 1000, passes * 4 / (ms * 1e-3) / 1e6);
    }
  return 0;
}
