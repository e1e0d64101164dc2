// GPU probe 2 (development aid): recover the K-slot and scale association of v_mfma_scale_f32_32x32x64_f8f6f4 by one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void mx_one(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x16* c) {
  const int l = threadIdx.x;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  c[l] = acc;
}
static double dec8(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  double x = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
  return s ? -x : x;
}
static unsigned char HA[64][32], HB[64][32]; static int SA[64], SB[64]; static float C[32][32];
static i32x8 *da, *db; static int *dsa, *dsb; static f32x16* dc;
static void run() {
  CK(hipMemcpy(da, HA, 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(db, HB, 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, SA, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, SB, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_one, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
  float hc[64][16]; CK(hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost));
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = hc[l][r];
}
int main() {
  CK(hipMalloc(&da, 2048)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 4096));
  // 64 distinct positive fp8 values: codes 0x20 .. 0x5f (exp 4..11, all mantissas)
  unsigned char code[2][32];
  for (int g = 0; g < 2; ++g) for (int p = 0; p < 32; ++p) code[g][p] = 0x20 + g * 32 + p;
  for (int l = 0; l < 64; ++l) { SA[l] = 0x7f7f7f7f; SB[l] = 0x7f7f7f7f; }
  printf("K-slot map: A (g,p) multiplies B (g',p')  [g = lane>>5, p = byte in the lane's 32]\n");
  int sym = 1;
  for (int g = 0; g < 2; ++g) for (int p = 0; p < 32; ++p) {
    memset(HA, 0, sizeof HA);
    for (int i = 0; i < 32; ++i) HA[i + 32 * g][p] = 0x38;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 32; ++q) HB[l][q] = code[l >> 5][q];
    run();
    int fg = -1, fp = -1;
    for (int g2 = 0; g2 < 2; ++g2) for (int p2 = 0; p2 < 32; ++p2) if (fabs(C[0][0] - dec8(code[g2][p2])) < 1e-6) { fg = g2; fp = p2; }
    int uni = 1; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (C[i][j] != C[0][0]) uni = 0;
    if (fg != g || fp != p || !uni) { sym = 0; printf("  A(%d,%2d) -> B(%d,%2d) uniform=%d C00=%g\n", g, p, fg, fp, uni, C[0][0]); }
  }
  printf("  symmetric (same slot on both sides, rows = lane&31 for A, cols = lane&31 for B): %s\n", sym ? "YES" : "NO");
  // scale association
  for (int side = 0; side < 2; ++side)
    for (int ls : {0, 5, 32, 37})
      for (int byte = 0; byte < 4; ++byte) {
        for (int l = 0; l < 64; ++l) { SA[l] = 0x7f7f7f7f; SB[l] = 0x7f7f7f7f; }
        (side ? SB : SA)[ls] += 1 << (8 * byte);
        // which K slots of which row/col get doubled: one-hot per slot
        char slots[2][33]; int rows_hit = 0, first_row = -1;
        for (int g = 0; g < 2; ++g) { for (int p = 0; p < 32; ++p) {
          memset(HA, 0, sizeof HA); memset(HB, 0, sizeof HB);
          for (int i = 0; i < 32; ++i) { HA[i + 32 * g][p] = 0x38; HB[i + 32 * g][p] = 0x38; }
          run();
          int hit = 0;
          for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (C[i][j] != 1.0f) { hit = 1; const int rc = side ? j : i; if (first_row < 0) first_row = rc; if (rc != first_row) rows_hit = 2; }
          slots[g][p] = hit ? 'X' : '.';
        } slots[g][32] = 0; }
        printf("scale_%c lane %2d byte %d +1: %s index %d%s, doubled slots g0 %s g1 %s\n", side ? 'b' : 'a', ls, byte, side ? "col" : "row", first_row, rows_hit == 2 ? " (+others)" : "",
               slots[0], slots[1]);
      }
  return 0;
}
