// Does fp32-input MFMA overlap with VALU work of a second wave on the same SIMD (gfx950)?
// 512-thread workgroups, one per CU: waves 0-3 issue MFMAs, waves 4-7 issue VALU FMAs.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: f32 32x32x2, 1: f32 16x16x4, 2: bf16 32x32x16
__global__ __launch_bounds__(512, 2) void k(float* out, int nm, int nv, int mode) {
  const int wv = threadIdx.x >> 6;
  float r = 0.f;
  if (wv < 4) {
    if (mode & 1) {
      if constexpr (SHAPE == 0) {
        f32x16 a0, a1;
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
        for (int i = 0; i < nm; ++i) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
      } else if constexpr (SHAPE == 1) {
        f32x4 a0, a1, a2, a3;
        for (int i = 0; i < 4; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
        const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
        for (int i = 0; i < nm; ++i) {     // 4 x 16x16x4 = the FLOPs of 2 x ... (each 2048 FLOP; 32x32x2 = 4096)
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) r += a0[i] + a1[i] + a2[i] + a3[i];
      } else {
        f32x16 a0, a1;
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        bf16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + threadIdx.x); y[i] = (short)(0x3f00 + i); }
        for (int i = 0; i < nm; ++i) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
      }
    }
  } else if (mode & 2) {
    float a = threadIdx.x, b = 1.0001f, c0 = 0.1f, c1 = 0.2f, c2 = 0.3f, c3 = 0.4f, c4 = 0.5f, c5 = 0.6f, c6 = 0.7f, c7 = 0.8f;
    for (int i = 0; i < nv; ++i) {          // 8 independent chains of v_fma_f32
      c0 = fmaf(c0, b, a); c1 = fmaf(c1, b, a); c2 = fmaf(c2, b, a); c3 = fmaf(c3, b, a);
      c4 = fmaf(c4, b, a); c5 = fmaf(c5, b, a); c6 = fmaf(c6, b, a); c7 = fmaf(c7, b, a);
    }
    r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <int SHAPE>
void run(const char* name, int nm, int nv) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms[4];
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms[mode], a, b);
    }
  }
  printf("%-14s mfma only %8.1f us   valu only %8.1f us   both %8.1f us   (sum %8.1f, max %8.1f)\n", name, ms[1] * 1e3,
         ms[2] * 1e3, ms[3] * 1e3, (ms[1] + ms[2]) * 1e3, (ms[1] > ms[2] ? ms[1] : ms[2]) * 1e3);
}

int main() {
  run<0>("f32 32x32x2", 2000, 8000);     // 4000 MFMAs x 64 cyc = 256k cycles; 64000 FMAs x 4 cyc = 256k cycles
  run<1>("f32 16x16x4", 2000, 8000);     // 8000 MFMAs x 32 cyc
  run<2>("bf16 32x32x16", 4000, 8000);   // 8000 MFMAs x 32 cyc
  return 0;
}
