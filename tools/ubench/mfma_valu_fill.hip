// How many independent VALU instructions hide in the shadow of one fp32-input MFMA issued by the SAME wave
// (one wave per SIMD, 256-thread workgroups, one per CU), gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int F>
__global__ __launch_bounds__(256, 1) void k(float* out, int n) {
  f32x16 a0; f32x4 b0, b1;
  for (int i = 0; i < 16; ++i) a0[i] = 0.f;
  for (int i = 0; i < 4; ++i) { b0[i] = 0.f; b1[i] = 0.f; }
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  float c[16];
  for (int i = 0; i < 16; ++i) c[i] = 0.1f * i;
  const float m = 1.0001f, ad = threadIdx.x;
  for (int it = 0; it < n; ++it) {
    if constexpr (SHAPE == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    } else {
      b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < F; ++f) c[f % 16] = fmaf(c[f % 16], m, ad);
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += a0[i] + c[i];
  for (int i = 0; i < 4; ++i) r += b0[i] + b1[i];
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
  printf("%s + %2d VALU per 64 MFMA-cycles: %7.1f us  (%.1f ns per iteration)\n", SHAPE == 0 ? "f32 32x32x2  " : "f32 2x16x16x4", F, ms * 1e3,
         ms * 1e6 / n);
}

int main() {
  run<0, 0>(); run<0, 4>(); run<0, 8>(); run<0, 12>(); run<0, 16>(); run<0, 24>(); run<0, 32>();
  run<1, 0>(); run<1, 4>(); run<1, 8>(); run<1, 12>(); run<1, 16>(); run<1, 24>(); run<1, 32>();
  return 0;
}
