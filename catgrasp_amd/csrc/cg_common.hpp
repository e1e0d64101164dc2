// Shared device helpers for the catgrasp_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CG_OK 0
#define CG_ERR_ARG (-1)
#define CG_ERR_UNSUPPORTED (-2)

// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA (A: lane l -> A[i=l&31][k=l>>5], B: lane l -> B[k=l>>5][j=l&31],
// D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5)).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register r for lane l in a 32x32 MFMA tile
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// max over the 16 accumulator registers of a lane.  Written as max(max(m, a), b) chains so that every pair folds into one
// v_max3_f32 (8 VALU ops per tile); the pointmlp files are compiled with -fno-honor-nans, which removes the per-operand
// canonicalisation (v_max x, x) the IEEE maxnum lowering would otherwise insert for MFMA results.
__device__ __forceinline__ float max16(const f32x16& c) {
  float m = fmaxf(fmaxf(c[0], c[1]), c[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, c[i]), c[i + 1]);
  return fmaxf(m, c[15]);
}

// float atomic max valid for any sign (buffer pre-filled with -inf)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // by sign BIT: -0.0f must take the unsigned-min branch (its int pattern is INT_MIN and would lose every signed max)
  if (!(__float_as_uint(v) >> 31)) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

static inline int cg_hip_status(hipError_t e) { return e == hipSuccess ? CG_OK : (int)e; }

// per-device launch state (function attributes, CU counts) is indexed by the HIP device ordinal
constexpr int CG_MAX_DEVICES = 64;
static inline int cg_device_cu_count(int dev) {
  static int n_cu[CG_MAX_DEVICES] = {};
  if (dev < 0 || dev >= CG_MAX_DEVICES) return -1;
  if (n_cu[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu[dev] = prop.multiProcessorCount;
  }
  return n_cu[dev];
}
