// Shared helpers for the libsug_amd.so kernels (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/sug_amd.h"

void sug_set_error(const char* fmt, ...);

#define SUG_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      sug_set_error(__VA_ARGS__);             \
      return SUG_ERR_ARG;                     \
    }                                         \
  } while (0)

#define SUG_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      sug_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return SUG_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

// BatchNorm batch statistics are accumulated about a PIVOT p[c] (one value of the group's data per channel): the
// partial rows hold sum(x - p) | sum((x - p)^2) in fp32, the finalize kernels form mean = p + S1/n and
// var = S2/n - (S1/n)^2 in fp64.  Unshifted fp32 partials lose the variance when |mean| >> std (E[x^2] - mean^2).
// The pivot row of group g lives in the caller's workspace behind the partial rows: producers use at most
// SUG_STATS_ROWS of its SUG_STATS_BLOCKS rows.
#define SUG_STATS_ROWS (SUG_STATS_BLOCKS - 8)
#define SUG_PIVOT_OFFSET(C) ((size_t)SUG_STATS_ROWS * 2 * (C))      /* + g*C: pivot row of group g (<= 16 groups) */

// edgeconv.hip: fold `nblk` per-workgroup partial rows ws[nblk][2C] (sum | sum of squares about `pivot` [C], or plain
// sums when pivot == nullptr; fixed order, fp64) into the BatchNorm coefficients coef[5][C] and update the running buffers
int sug_stats_finalize(const float* ws, int nblk, int C, const float* gamma, const float* beta, double count, float eps,
                       float momentum, float* running_mean, float* running_var, float* coef, hipStream_t st,
                       const float* pivot = nullptr);

// knn.hip: reverse lists of B clouds with E index entries each (destinations in [0, N)): rev_off [B, N+1],
// rev_ent [B, E] (entry positions; ascending per destination when sorted != 0)
int sug_reverse_lists(const int32_t* idx, int B, int E, int N, int sorted, int32_t* rev_off, int32_t* rev_ent,
                      hipStream_t st);
// edgeconv.hip: out[W] (fp64) = ordered sum of nblk partial rows ws[nblk][W]
int sug_reduce_partials(const float* ws, int nblk, int W, double* out, hipStream_t st);

// knn_pc.hip: producer / consumer MFMA kNN (C in {3, 64, 128}, k <= 20)
int sug_knn_pc_supported(const float* x, int64_t ldx, int C, int k);
int sug_knn_pc(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st);

#define WAVE 64

// Opt-in to more than 64 KB of dynamic LDS for one kernel.  The attribute is per DEVICE, so the
// "already done" note is kept per device (atomics: two host threads racing here both set the same
// value, which is harmless) and a failure is reported, not dropped.  This cache of a driver
// attribute is the only state the library keeps besides the thread-local error string.
#include <atomic>
#define SUG_MAX_DEVICES 64
struct SugLdsOptIn {
  std::atomic<int> bytes[SUG_MAX_DEVICES];
};
template <typename KernelT>
static inline int sug_allow_dynamic_lds(SugLdsOptIn& note, KernelT kernel, int bytes, const char* name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SUG_MAX_DEVICES) {
    sug_set_error("%s: cannot identify the current device", name);
    return SUG_ERR_LAUNCH;
  }
  if (note.bytes[dev].load(std::memory_order_acquire) >= bytes) return SUG_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sug_set_error("%s: %d bytes of dynamic LDS refused on device %d: %s", name, bytes, dev, hipGetErrorString(e));
    return SUG_ERR_LAUNCH;
  }
  note.bytes[dev].store(bytes, std::memory_order_release);
  return SUG_OK;
}

static inline int sug_divup(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Compute units of the current device (a driver attribute, cached per device like the LDS opt-in; 256 on MI355X): launchers
// that choose between a coarse and a fine grid ask for it.
static inline int sug_cu_count() {
  static std::atomic<int> cus[SUG_MAX_DEVICES];
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SUG_MAX_DEVICES) {
    ncu = cus[dev].load(std::memory_order_relaxed);
    if (ncu == 0) {
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
      cus[dev].store(ncu, std::memory_order_relaxed);
    }
  }
  return ncu;
}

// |p|^2 as the CPU reference's torch.sum(x**2) over 3 channels: separately rounded
// squares, summed left to right (verified bitwise against the reference).
__device__ __forceinline__ float sq3(float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}
// <a,b> as the reference's K=3 sgemm: ascending fma chain.
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, __fmul_rn(ax, bx)));
}
// square_distance(src=q, dst=p), model/point_utils.py:128-130: ((-2*dot) + |q|^2) + |p|^2
__device__ __forceinline__ float sqdist_expanded(float dot, float nq, float np) {
  return __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), nq), np);
}

// Wave-wide max / min over all 64 lanes through the DPP cross-lane paths (quad_perm, row_half_mirror, row_mirror,
// row_bcast:15 / :31) instead of six dependent LDS-crossbar shuffles: the result is returned to every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int sug_dpp(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_max_f(float v) {
  v = fmaxf(v, __int_as_float(sug_dpp<0xB1, 0xf>(__float_as_int(v))));    // quad_perm [1,0,3,2]
  v = fmaxf(v, __int_as_float(sug_dpp<0x4E, 0xf>(__float_as_int(v))));    // quad_perm [2,3,0,1]
  v = fmaxf(v, __int_as_float(sug_dpp<0x141, 0xf>(__float_as_int(v))));   // row_half_mirror
  v = fmaxf(v, __int_as_float(sug_dpp<0x140, 0xf>(__float_as_int(v))));   // row_mirror: every lane = its row's max
  v = fmaxf(v, __int_as_float(sug_dpp<0x142, 0xa>(__float_as_int(v))));   // row_bcast:15 into rows 1, 3
  v = fmaxf(v, __int_as_float(sug_dpp<0x143, 0xc>(__float_as_int(v))));   // row_bcast:31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f(float v) {
  v = fminf(v, __int_as_float(sug_dpp<0xB1, 0xf>(__float_as_int(v))));
  v = fminf(v, __int_as_float(sug_dpp<0x4E, 0xf>(__float_as_int(v))));
  v = fminf(v, __int_as_float(sug_dpp<0x141, 0xf>(__float_as_int(v))));
  v = fminf(v, __int_as_float(sug_dpp<0x140, 0xf>(__float_as_int(v))));
  v = fminf(v, __int_as_float(sug_dpp<0x142, 0xa>(__float_as_int(v))));
  v = fminf(v, __int_as_float(sug_dpp<0x143, 0xc>(__float_as_int(v))));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, sug_dpp<0xB1, 0xf>(v));
  v = min(v, sug_dpp<0x4E, 0xf>(v));
  v = min(v, sug_dpp<0x141, 0xf>(v));
  v = min(v, sug_dpp<0x140, 0xf>(v));
  v = min(v, sug_dpp<0x142, 0xa>(v));
  v = min(v, sug_dpp<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
