// Weight gradient of a per-point linear layer: dW[M,N] = g^T . x over R rows (R = B*N = 32768 at
// C2, M/N = 3..512).  rocBLAS runs these "K = R, tiny output" GEMMs at ~140 us regardless of
// size (few output tiles -> few workgroups, each walking all of K); here K is split over row
// chunks so thousands of waves stream g and x once at HBM rate:
//   workgroup = (row chunk, 64x64 output super-tile); each of its 4 waves owns a quarter of the
//   chunk's rows and 2x2 tiles of v_mfma_f32_32x32x2_f32 (K = 2 rows per MFMA: lanes 0-31 carry
//   row r, lanes 32-63 row r+1; operands are plain coalesced 128-B row segments of g and x, no
//   LDS staging); waves combine through LDS in a fixed order; per-chunk partials are summed by
//   an ordered second kernel (bit-reproducible, no float atomics).
// Algorithmic bytes 4*R*(M+N) + 4*M*N; FLOPs 2*R*M*N: HBM-bound for M,N <= 128.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// FULL: the 64x64 super-tile lies inside [M,N] and every wave owns a multiple of 8 whole rows: the operand loads
// carry no guards (a guarded load compiles to a branch and the wait counters collapse to vmcnt(0): nothing stays
// in flight across the MFMAs) and four row pairs are in flight ahead of the MFMAs; same accumulation order, same bits.
template <bool FULL>
__global__ __launch_bounds__(256) void linear_dw_kernel(const float* __restrict__ g, int64_t ldg,
                                                        const float* __restrict__ x, int64_t ldx,
                                                        int64_t R, int M, int N, int rows_per_block,
                                                        float* __restrict__ part, int with_db) {
  __shared__ float s_acc[3][4][16][64];          // waves 1..3 x (2x2 tiles) x 16 regs x 64 lanes
  __shared__ float s_db[4][64];                  // column sums of g (bias gradient), per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
  const int64_t rb0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t rb1 = rb0 + rows_per_block;
  if (rb1 > R) rb1 = R;
  const int per = rows_per_block / 4;            // rows_per_block is a multiple of 8
  int64_t r0 = rb0 + (int64_t)wv * per, r1 = r0 + per;
  if (r1 > rb1) r1 = rb1;
  const bool ia0 = i0 + col < M, ia1 = i0 + 32 + col < M;
  const bool jb0 = j0 + col < N, jb1 = j0 + 32 + col < N;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float sa0 = 0.f, sa1 = 0.f;                    // sum of this lane's g elements (rows of its parity)
  if constexpr (FULL) {
    constexpr int ST = 4;                        // (8 in flight measured no faster)
    float a0[ST], a1[ST], b0[ST], b1[ST];
    auto fetch = [&](int64_t r, int s) {
      int64_t row = r + kh;
      row = row < R ? row : R - 1;               // prefetches past the chunk are never consumed
      const float* gr = g + row * ldg + i0 + col;
      const float* xr = x + row * ldx + j0 + col;
      a0[s] = gr[0]; a1[s] = gr[32]; b0[s] = xr[0]; b1[s] = xr[32];
    };
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) fetch(r0 + 2 * s, s);
    for (int64_t r = r0; r < r1; r += 2 * ST) {
#pragma unroll
      for (int s = 0; s < ST; ++s) {
        fetch(r + 2 * (s + ST - 1), (s + ST - 1) % ST);
        sa0 += a0[s];
        sa1 += a1[s];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
      }
    }
  } else
  for (int64_t r = r0; r < r1; r += 2) {
    const int64_t row = r + kh;
    const bool ok = row < r1;
    const float* gr = g + row * ldg + i0 + col;
    const float* xr = x + row * ldx + j0 + col;
    const float a0 = (ok && ia0) ? gr[0] : 0.f, a1 = (ok && ia1) ? gr[32] : 0.f;
    const float b0 = (ok && jb0) ? xr[0] : 0.f, b1 = (ok && jb1) ? xr[32] : 0.f;
    sa0 += a0;
    sa1 += a1;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  if (wv > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[wv - 1][t][r][lane] = acc[t >> 1][t & 1][r];
  }
  const size_t pstride = (size_t)M * N + (with_db ? M : 0);
  if (with_db && blockIdx.z == 0) {              // even + odd rows, then the four waves in a fixed order
    sa0 += __shfl_xor(sa0, 32);
    sa1 += __shfl_xor(sa1, 32);
    s_db[wv][lane] = kh == 0 ? sa0 : sa1;        // lanes 0-31: columns i0+col, lanes 32-63: columns i0+32+col
  }
  __syncthreads();
  if (wv == 0) {
    float* p = part + (size_t)blockIdx.x * pstride;
    if (with_db && blockIdx.z == 0) {
      const int i = i0 + lane;
      if (i < M) p[(size_t)M * N + i] = ((s_db[0][lane] + s_db[1][lane]) + s_db[2][lane]) + s_db[3][lane];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ti = i0 + (t >> 1) * 32, tj = j0 + (t & 1) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ((acc[t >> 1][t & 1][r] + s_acc[0][t][r][lane]) + s_acc[1][t][r][lane]) + s_acc[2][t][r][lane];
        const int i = ti + (r & 3) + 8 * (r >> 2) + 4 * kh, j = tj + col;
        if (i < M && j < N) p[(size_t)i * N + j] = v;
      }
    }
  }
}

// dw[e] = sum over chunks, 16 chunk-lanes x 16 elements per workgroup; fixed combination order.
// Large outputs (M, N >= 128): one wave per workgroup owns a 128x128 output tile (4x4 MFMA tiles,
// 256 accumulator registers) over its row chunk: 8 coalesced 128-B loads feed 16 MFMAs, twice the
// arithmetic intensity of the 64x64 kernel, and no cross-wave combine.
template <bool FULL>
__global__ __launch_bounds__(64) void linear_dw_big_kernel(const float* __restrict__ g, int64_t ldg,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           int64_t R, int M, int N, int rows_per_block,
                                                           float* __restrict__ part, int with_db) {
  const int lane = threadIdx.x;
  const int col = lane & 31, kh = lane >> 5;
  const int i0 = blockIdx.y * 128, j0 = blockIdx.z * 128;
  const bool do_db = with_db && blockIdx.z == 0;
  float sdb[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  bool ia[4], jb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ia[t] = i0 + 32 * t + col < M;
    jb[t] = j0 + 32 * t + col < N;
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // one wave per SIMD: nothing else hides the load latency, so the operands of the next ST-1 row
  // pairs are in flight while the 16 MFMAs of a row pair are issued (rows past r1 load as zeros)
  auto fetch = [&](int64_t r, float (&av)[4], float (&bv)[4]) {
    int64_t row = r + kh;
    if constexpr (FULL) row = row < R ? row : R - 1;   // full tile, whole row groups: no guards (see linear_dw_kernel)
    const bool ok = row < r1;
    const float* gr = g + row * ldg + i0 + col;
    const float* xr = x + row * ldx + j0 + col;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (FULL) {
        av[t] = gr[32 * t];
        bv[t] = xr[32 * t];
      } else {
        av[t] = (ok && ia[t]) ? gr[32 * t] : 0.f;
        bv[t] = (ok && jb[t]) ? xr[32 * t] : 0.f;
      }
    }
  };
  constexpr int ST = 4;                       // row pairs in flight ahead of the MFMAs
  float av[ST][4], bv[ST][4];
#pragma unroll
  for (int s = 0; s < ST - 1; ++s) fetch(r0 + 2 * s, av[s], bv[s]);
  for (int64_t r = r0; r < r1; r += 2 * ST) {
#pragma unroll
    for (int s = 0; s < ST; ++s) {
      fetch(r + 2 * (s + ST - 1), av[(s + ST - 1) % ST], bv[(s + ST - 1) % ST]);
      if (do_db) {
#pragma unroll
        for (int a = 0; a < 4; ++a) sdb[a] += av[s][a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][a], bv[s][b], acc[a][b], 0, 0, 0);
    }
  }
  float* p = part + (size_t)blockIdx.x * ((size_t)M * N + (with_db ? M : 0));
  if (do_db) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float t = sdb[a] + __shfl_xor(sdb[a], 32);
      const int i = i0 + 32 * a + col;
      if (kh == 0 && i < M) p[(size_t)M * N + i] = t;
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh, j = j0 + 32 * b + col;
        if (i < M && j < N) p[(size_t)i * N + j] = acc[a][b][r];
      }
}

__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float* __restrict__ part, int nchunk,
                                                               int64_t MN, int64_t P, float* __restrict__ dw,
                                                               float* __restrict__ db) {
  // partial rows hold P = MN (+ M bias-gradient columns) elements; e < MN -> dw, the rest -> db
  __shared__ double s_p[16][17];
  const int el = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t e = (int64_t)blockIdx.x * 16 + el;
  double acc = 0.0;
  if (e < P) {
    int c = cl;
    for (; c + 112 < nchunk; c += 128) {          // eight rows in flight, added in the order of the plain loop
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + 16 * u) * P + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; c < nchunk; c += 16) acc += (double)part[(size_t)c * P + e];
  }
  s_p[cl][el] = acc;
  __syncthreads();
  if (cl == 0 && e < P) {
    double t = s_p[0][el];
#pragma unroll
    for (int i = 1; i < 16; ++i) t += s_p[i][el];
    if (e < MN)
      dw[e] = (float)t;
    else
      db[e - MN] = (float)t;
  }
}

// one-wave 128x128 tiles for outputs of at least 128 x 128 -- except a single such tile over few rows, where the
// 64x64 kernel's four super-tiles bring four times the waves (128x128 at 65536 rows: 49 against 59 us; at 2^20
// rows the big tile's operand reuse wins, 360 against 578 us)
bool use_big(int64_t R, int M, int N) { return M >= 128 && N >= 128 && ((int64_t)M * N > 128 * 128 || R > 262144); }

int plan_rows_per_block(int64_t R, int M, int N) {
  const int tiles = use_big(R, M, N) ? sug_divup(M, 128) * sug_divup(N, 128) : sug_divup(M, 64) * sug_divup(N, 64);
  // ~1024 one-wave workgroups (128x128 tiles) or ~512 four-wave workgroups (64x64 super-tiles: twice the rows per
  // chunk halve the partial rows the fold reads; measured best of 256 / 512 / 768 / 1024, tools/bench_dw.py)
  int64_t nchunk = (use_big(R, M, N) ? 1024 : 512) / tiles;
  if (nchunk < 8) nchunk = 8;
  int64_t rpb = (R + nchunk - 1) / nchunk;
  if (rpb < 128) rpb = 128;
  return (int)((rpb + 7) / 8 * 8);
}

}  // namespace

extern "C" int64_t sug_linear_dw_workspace(int64_t R, int M, int N) {
  if (R <= 0 || M <= 0 || N <= 0) return 0;
  return (int64_t)sug_divup(R, plan_rows_per_block(R, M, N)) * ((int64_t)M * N + M);     // room for the db columns
}

extern "C" int sug_linear_dw_bias(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M,
                                  int N, float* dw, float* db, float* ws, void* stream) {
  SUG_REQUIRE(g && x && dw && ws, "sug_linear_dw: null pointer");
  SUG_REQUIRE(R > 0 && M > 0 && N > 0 && ldg >= M && ldx >= N, "sug_linear_dw: bad shape");
  const int rpb = plan_rows_per_block(R, M, N);
  const int nchunk = sug_divup(R, rpb);
  SUG_REQUIRE(sug_divup(M, 64) <= 65535 && sug_divup(N, 64) <= 65535, "sug_linear_dw: output too large");
  hipStream_t st = (hipStream_t)stream;
  const int with_db = db ? 1 : 0;
  const bool whole = R % rpb == 0 && rpb % 32 == 0;      // every wave of every chunk: a multiple of 8 rows
  if (use_big(R, M, N)) {
    const dim3 grid(nchunk, sug_divup(M, 128), sug_divup(N, 128));
    if (R % rpb == 0 && M % 128 == 0 && N % 128 == 0)      // (rpb is a multiple of 8)
      hipLaunchKernelGGL(linear_dw_big_kernel<true>, grid, dim3(64), 0, st, g, ldg, x, ldx, R, M, N, rpb, ws, with_db);
    else
      hipLaunchKernelGGL(linear_dw_big_kernel<false>, grid, dim3(64), 0, st, g, ldg, x, ldx, R, M, N, rpb, ws, with_db);
  } else {
    const dim3 grid(nchunk, sug_divup(M, 64), sug_divup(N, 64));
    if (whole && M % 64 == 0 && N % 64 == 0)
      hipLaunchKernelGGL(linear_dw_kernel<true>, grid, dim3(256), 0, st, g, ldg, x, ldx, R, M, N, rpb, ws, with_db);
    else
      hipLaunchKernelGGL(linear_dw_kernel<false>, grid, dim3(256), 0, st, g, ldg, x, ldx, R, M, N, rpb, ws, with_db);
  }
  SUG_LAUNCH_CHECK("sug_linear_dw");
  const int64_t MN = (int64_t)M * N, P = MN + (with_db ? M : 0);
  hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3(sug_divup(P, 16)), dim3(256), 0, st, ws, nchunk, MN, P, dw, db);
  SUG_LAUNCH_CHECK("sug_linear_dw(reduce)");
  return SUG_OK;
}

extern "C" int sug_linear_dw(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M,
                             int N, float* dw, float* ws, void* stream) {
  return sug_linear_dw_bias(g, ldg, x, ldx, R, M, N, dw, nullptr, ws, stream);
}

// Ordered fold of `nchunk` partial outputs part[nchunk][MN] (fp64 accumulation, fixed order, no atomics, no memset):
// the second half of sug_linear_dw, exposed for split-K partials that a batched library GEMM produced.
extern "C" int sug_linear_dw_fold(const float* part, int nchunk, int64_t MN, float* dw, void* stream) {
  SUG_REQUIRE(part && dw && nchunk > 0 && MN > 0, "sug_linear_dw_fold: bad argument");
  hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3(sug_divup(MN, 16)), dim3(256), 0, (hipStream_t)stream, part, nchunk, MN,
                     MN, dw, (float*)nullptr);
  SUG_LAUNCH_CHECK("sug_linear_dw_fold");
  return SUG_OK;
}
