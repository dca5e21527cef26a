// VALU fillers in the shadow of MFMAs issued by the SAME wave: bf16 32x32x16 (one chain / two chains) and
// fp32 32x32x2 with two independent chains.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int F>   // 0: bf16 one chain, 1: bf16 two chains (F fillers after each), 2: f32 two chains
__global__ __launch_bounds__(256, 1) void k(float* out, int n) {
  f32x16 a0, a1;
  for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  bf16x8 xb, yb;
  for (int i = 0; i < 8; ++i) { xb[i] = (short)(0x3f80 + threadIdx.x); yb[i] = (short)(0x3f00 + i); }
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  float c[16];
  for (int i = 0; i < 16; ++i) c[i] = 0.1f * i;
  const float m = 1.0001f, ad = threadIdx.x;
  for (int it = 0; it < n; ++it) {
    if constexpr (SHAPE == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
    } else if constexpr (SHAPE == 1) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < F; ++f) c[f % 16] = fmaf(c[f % 16], m, ad);
      __builtin_amdgcn_sched_barrier(0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yb, xb, a1, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < F; ++f) c[f % 16] = fmaf(c[f % 16], m, ad);
      __builtin_amdgcn_sched_barrier(0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < F; ++f) c[f % 16] = fmaf(c[f % 16], m, ad);
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + c[i];
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int SHAPE, int F>
void run() {
  float* out;
  (void)hipMalloc(&out, 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms = 0;
  const int n = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<SHAPE, F>), dim3(256), dim3(256), 0, 0, out, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
  }
  const char* nm[] = {"bf16 32x32x16, one chain, F fillers per MFMA ", "bf16 32x32x16, two chains, F fillers per MFMA", "f32 32x32x2, two chains, F fillers per MFMA  "};
  printf("%s F=%2d: %7.1f us  (%.1f ns per MFMA)\n", nm[SHAPE], F, ms * 1e3, ms * 1e6 / n / (SHAPE == 0 ? 1 : 2));
}

int main() {
  run<0, 0>(); run<0, 2>(); run<0, 4>(); run<0, 6>(); run<0, 8>(); run<0, 12>();
  run<1, 0>(); run<1, 2>(); run<1, 4>(); run<1, 6>(); run<1, 8>(); run<1, 12>();
  run<2, 0>(); run<2, 2>(); run<2, 4>(); run<2, 8>(); run<2, 12>(); run<2, 16>();
  return 0;
}
