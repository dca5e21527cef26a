// Measured ablation for a fused Point Transformer attention block: how fast can ONE workgroup chain the
// three 512 x 512 fp16 linears of Ptran_transformer.py:39-44 (fc_delta[2], fc_gamma[0], fc_gamma[2]) on the k-expanded rows
// without the activations leaving the CU?  The planned tile: 128 rows (8 points x 16 neighbours) x 512 channels of fp16
// activations resident in LDS (130 KB: one workgroup per CU), 8 waves, wave w owns output columns [64 w, 64 w + 64) of every
// layer (8 accumulators of v_mfma_f32_32x32x16_f16), the A operand read from LDS, the WEIGHTS streamed from L2 straight into
// the B operand registers (1.5 MB per workgroup; staging them through LDS does not fit beside the activations, and at 64
// rows per workgroup the L2 would have to deliver 33 TB/s).  No gathers, no softmax, no stores of the intermediates a
// training forward would have to save: this is the upper bound of the chain's MFMA side.
// build: hipcc -O3 --offload-arch=gfx950 ptran_chain.hip -o ptran_chain ; run: ./ptran_chain [rows=1048576]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int D = 512, ROWS = 128, LDX = D + 8;        // fp16 per LDS row (+ pad: conflict-free b128 reads)
constexpr int NT = 512;

template <int LAYERS, bool WEIGHTS_FROM_L2>
__global__ __launch_bounds__(NT) void chain_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ W,
                                                   _Float16* __restrict__ y, int nblk) {
  extern __shared__ __attribute__((aligned(16))) _Float16 sx[];       // [ROWS][LDX]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, j = lane & 31, h = lane >> 5;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    // input tile -> LDS (coalesced 16-byte pieces)
    const _Float16* xb = x + (size_t)blk * ROWS * D;
    for (int e = t; e < ROWS * D / 8; e += NT) {
      const int r = e / (D / 8), c = (e % (D / 8)) * 8;
      *reinterpret_cast<h8*>(sx + r * LDX + c) = *reinterpret_cast<const h8*>(xb + (size_t)r * D + c);
    }
    __syncthreads();
#pragma unroll 1
    for (int layer = 0; layer < LAYERS; ++layer) {
      const _Float16* Wl = W + (size_t)layer * D * D;
      f16v acc[4][2];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
      const _Float16* wp0 = Wl + (size_t)(w * 64 + j) * D + 8 * h;
      const _Float16* wp1 = wp0 + (size_t)32 * D;
      h8 b0[2], b1[2];
      b0[0] = *reinterpret_cast<const h8*>(wp0);
      b1[0] = *reinterpret_cast<const h8*>(wp1);
#pragma unroll 2
      for (int s = 0; s < D / 16; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < D / 16) {
          if (WEIGHTS_FROM_L2) {
            b0[nxt] = *reinterpret_cast<const h8*>(wp0 + 16 * (s + 1));
            b1[nxt] = *reinterpret_cast<const h8*>(wp1 + 16 * (s + 1));
          } else {                                                     // (ablation: no weight traffic at all)
            b0[nxt] = b0[cur]; b1[nxt] = b1[cur];
          }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const h8 a = *reinterpret_cast<const h8*>(sx + (rt * 32 + j) * LDX + 16 * s + 8 * h);
          acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0[cur], acc[rt][0], 0, 0, 0);
          acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1[cur], acc[rt][1], 0, 0, 0);
        }
      }
      __syncthreads();                       // every wave has read the old activations
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + 8 * (r >> 2) + (r & 3) + 4 * h;
            const float v = acc[rt][ct][r];
            sx[row * LDX + w * 64 + ct * 32 + j] = (_Float16)(v > 0.f ? v : 0.f);     // ReLU, back to fp16
          }
      __syncthreads();
    }
    // one output row block (what the attention reduction would leave): 8 rows per workgroup
    if (y) {
      for (int e = t; e < 8 * D / 8; e += NT) {
        const int r = e / (D / 8), c = (e % (D / 8)) * 8;
        *reinterpret_cast<h8*>(y + ((size_t)blk * 8 + r) * D + c) = *reinterpret_cast<const h8*>(sx + r * 16 * LDX + c);
      }
    }
    __syncthreads();
  }
}

template <int LAYERS, bool WL2>
float run(const _Float16* x, const _Float16* W, _Float16* y, int nblk, int grid) {
  const size_t sh = (size_t)ROWS * LDX * sizeof(_Float16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<LAYERS, WL2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((chain_kernel<LAYERS, WL2>), dim3(grid), dim3(NT), sh, 0, x, W, y, nblk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  return ms;
}

int main(int argc, char** argv) {
  const size_t R = argc > 1 ? (size_t)atoll(argv[1]) : 1048576;
  const int nblk = (int)(R / ROWS);
  _Float16 *x, *W, *y;
  hipMalloc(&x, R * D * 2); hipMalloc(&W, 3 * D * D * 2); hipMalloc(&y, (size_t)nblk * 8 * D * 2);
  hipMemset(x, 0x11, R * D * 2); hipMemset(W, 0x22, 3 * D * D * 2);
  const double flop3 = 3.0 * 2.0 * R * D * D, flop1 = flop3 / 3.0;
  for (int grid : {256, 512, nblk}) {
    const float t3 = run<3, true>(x, W, y, nblk, grid), t1 = run<1, true>(x, W, y, nblk, grid), t3n = run<3, false>(x, W, y, nblk, grid);
    printf("rows %zu, grid %5d: 3 layers, weights from L2 %8.1f us = %6.1f TFLOP/s | 1 layer %8.1f us = %6.1f TFLOP/s | 3 layers, no weight loads %8.1f us = %6.1f TFLOP/s\n",
           R, grid, t3 * 1e3, flop3 / t3 * 1e-9, t1 * 1e3, flop1 / t1 * 1e-9, t3n * 1e3, flop3 / t3n * 1e-9);
  }
  printf("(dense fp16 MFMA peak 2500 TFLOP/s; the input read is %0.2f GB of HBM traffic)\n", R * D * 2 / 1e9);
  return 0;
}
