// kNN graph over one cloud (replaces knn(), model/model_utils.py:178-185) and the
// reverse neighbour lists used by the EdgeConv backward.
//
// Layout: x [B,N,C] point-major rows (row stride ldx).  One thread owns one query
// point and keeps its K best (score, index) pairs sorted in registers; candidate
// rows are streamed through LDS in tiles of TJ points, read back as wave-wide
// broadcasts (every lane reads the same candidate -> conflict-free).
//
// Roofline: algorithmic bytes 4*C*N + 4*N*k per cloud, FLOPs N^2*(2C+3): compute
// bound for every C; the [B,N,N] matrix of the reference never exists.
#include "common.h"

namespace {

template <int K>
__device__ __forceinline__ void topk_insert_desc(float (&v)[K], int (&id)[K], float s, int j) {
  // precondition s > v[K-1]; strict compares keep earlier (lower) j first among ties
#pragma unroll
  for (int t = K - 1; t > 0; --t) {
    const bool up = s > v[t - 1];
    const bool here = s > v[t];
    const float nv = up ? v[t - 1] : (here ? s : v[t]);
    const int ni = up ? id[t - 1] : (here ? j : id[t]);
    v[t] = nv;
    id[t] = ni;
  }
  if (s > v[0]) {
    v[0] = s;
    id[0] = j;
  }
}

// row stride of the candidate tile in LDS: +4 keeps 16-B alignment for the
// broadcast float4 reads and de-conflicts the per-candidate norm pass.
template <int C>
struct TileStride { static constexpr int value = (C % 4 == 0) ? C + 4 : C; };

template <int C, int K, int BLOCK, int TJ>
__global__ __launch_bounds__(BLOCK) void knn_self_kernel(const float* __restrict__ x, int64_t ldx,
                                                         int N, int k, int32_t* __restrict__ idx) {
  constexpr int CS = TileStride<C>::value;
  __shared__ __attribute__((aligned(16))) float s_x[TJ * CS];
  __shared__ float s_n[TJ];
  const int b = blockIdx.y;
  const float* xb = x + (int64_t)b * N * ldx;
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  const int qc = q < N ? q : N - 1;

  float xi[C];
#pragma unroll
  for (int c = 0; c < C; ++c) xi[c] = xb[(int64_t)qc * ldx + c];
  float ni = __fmul_rn(xi[0], xi[0]);
#pragma unroll
  for (int c = 1; c < C; ++c) ni = __fadd_rn(ni, __fmul_rn(xi[c], xi[c]));

  float v[K];
  int id[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    v[t] = -INFINITY;
    id[t] = 0;
  }

  for (int j0 = 0; j0 < N; j0 += TJ) {
    __syncthreads();
    for (int e = threadIdx.x; e < TJ * C; e += BLOCK) {
      const int r = e / C, c = e - r * C;
      const int j = j0 + r;
      s_x[r * CS + c] = j < N ? xb[(int64_t)j * ldx + c] : 0.0f;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < TJ; r += BLOCK) {
      const float* xr = s_x + r * CS;
      float nj = __fmul_rn(xr[0], xr[0]);
#pragma unroll 8
      for (int c = 1; c < C; ++c) nj = __fadd_rn(nj, __fmul_rn(xr[c], xr[c]));
      s_n[r] = nj;
    }
    __syncthreads();
    const int rmax = (N - j0) < TJ ? (N - j0) : TJ;
    for (int r = 0; r < rmax; ++r) {
      const float* xr = s_x + r * CS;
      float acc;
      if constexpr (C % 4 == 0) {
        const float4* xr4 = reinterpret_cast<const float4*>(xr);
        float4 t = xr4[0];
        acc = __fmul_rn(xi[0], t.x);
        acc = fmaf(xi[1], t.y, acc);
        acc = fmaf(xi[2], t.z, acc);
        acc = fmaf(xi[3], t.w, acc);
#pragma unroll
        for (int c4 = 1; c4 < C / 4; ++c4) {
          t = xr4[c4];
          acc = fmaf(xi[4 * c4 + 0], t.x, acc);
          acc = fmaf(xi[4 * c4 + 1], t.y, acc);
          acc = fmaf(xi[4 * c4 + 2], t.z, acc);
          acc = fmaf(xi[4 * c4 + 3], t.w, acc);
        }
      } else {
        acc = __fmul_rn(xi[0], xr[0]);
#pragma unroll
        for (int c = 1; c < C; ++c) acc = fmaf(xi[c], xr[c], acc);
      }
      // pairwise_distance = -xx - inner - xx^T, inner = -2*dot (model_utils.py:179-181)
      const float s = __fsub_rn(__fsub_rn(-s_n[r], __fmul_rn(-2.0f, acc)), ni);
      if (s > v[K - 1]) topk_insert_desc<K>(v, id, s, j0 + r);
    }
  }
  if (q < N) {
    int32_t* o = idx + ((int64_t)b * N + q) * k;
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < k) o[t] = id[t];
  }
}

// Any C: the query row stays in global memory (L1-resident), candidates in LDS.
template <int K, int BLOCK, int TJ>
__global__ __launch_bounds__(BLOCK) void knn_self_generic_kernel(const float* __restrict__ x,
                                                                 int64_t ldx, int N, int C, int k,
                                                                 int32_t* __restrict__ idx) {
  extern __shared__ float s_dyn[];  // TJ*C candidate tile
  const int b = blockIdx.y;
  const float* xb = x + (int64_t)b * N * ldx;
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  const int qc = q < N ? q : N - 1;
  const float* xq = xb + (int64_t)qc * ldx;
  float ni = __fmul_rn(xq[0], xq[0]);
  for (int c = 1; c < C; ++c) ni = __fadd_rn(ni, __fmul_rn(xq[c], xq[c]));
  float v[K];
  int id[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    v[t] = -INFINITY;
    id[t] = 0;
  }
  for (int j0 = 0; j0 < N; j0 += TJ) {
    __syncthreads();
    for (int e = threadIdx.x; e < TJ * C; e += BLOCK) {
      const int r = e / C, c = e - r * C;
      const int j = j0 + r;
      s_dyn[e] = j < N ? xb[(int64_t)j * ldx + c] : 0.0f;
    }
    __syncthreads();
    const int rmax = (N - j0) < TJ ? (N - j0) : TJ;
    for (int r = 0; r < rmax; ++r) {
      const float* xr = s_dyn + r * C;
      float acc = __fmul_rn(xq[0], xr[0]);
      float nj = __fmul_rn(xr[0], xr[0]);
      for (int c = 1; c < C; ++c) {
        acc = fmaf(xq[c], xr[c], acc);
        nj = __fadd_rn(nj, __fmul_rn(xr[c], xr[c]));
      }
      const float s = __fsub_rn(__fsub_rn(-nj, __fmul_rn(-2.0f, acc)), ni);
      if (s > v[K - 1]) topk_insert_desc<K>(v, id, s, j0 + r);
    }
  }
  if (q < N) {
    int32_t* o = idx + ((int64_t)b * N + q) * k;
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (t < k) o[t] = id[t];
  }
}

template <int C, int K>
int launch_knn(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  // one wave per block keeps >= 2 blocks per CU at B*N = 32k queries
  constexpr int BLOCK = 64;
  constexpr int TJ = (C >= 64) ? 32 : 128;
  dim3 grid(sug_divup(N, BLOCK), B);
  hipLaunchKernelGGL((knn_self_kernel<C, K, BLOCK, TJ>), grid, dim3(BLOCK), 0, st, x, ldx, N, k, idx);
  SUG_LAUNCH_CHECK("sug_knn");
  return SUG_OK;
}

template <int K>
int dispatch_knn_c(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx,
                   hipStream_t st) {
  switch (C) {
    case 3: return launch_knn<3, K>(x, ldx, B, N, k, idx, st);
    case 64: return launch_knn<64, K>(x, ldx, B, N, k, idx, st);
    case 128: return launch_knn<128, K>(x, ldx, B, N, k, idx, st);
    default: {
      constexpr int BLOCK = 64;
      int TJ = 64;
      while (TJ > 1 && (size_t)TJ * C * sizeof(float) > 48 * 1024) TJ >>= 1;
      SUG_REQUIRE((size_t)TJ * C * sizeof(float) <= 48 * 1024, "sug_knn: C=%d too large", C);
      dim3 grid(sug_divup(N, BLOCK), B);
      // TJ is a template constant of the kernel only through its loop bound: use 64/32/16/...
      size_t sh = (size_t)TJ * C * sizeof(float);
      if (TJ == 64)
        hipLaunchKernelGGL((knn_self_generic_kernel<K, BLOCK, 64>), grid, dim3(BLOCK), sh, st, x, ldx, N, C, k, idx);
      else if (TJ == 32)
        hipLaunchKernelGGL((knn_self_generic_kernel<K, BLOCK, 32>), grid, dim3(BLOCK), sh, st, x, ldx, N, C, k, idx);
      else if (TJ == 16)
        hipLaunchKernelGGL((knn_self_generic_kernel<K, BLOCK, 16>), grid, dim3(BLOCK), sh, st, x, ldx, N, C, k, idx);
      else
        hipLaunchKernelGGL((knn_self_generic_kernel<K, BLOCK, 8>), grid, dim3(BLOCK), (size_t)8 * C * sizeof(float), st, x, ldx, N, C, k, idx);
      SUG_LAUNCH_CHECK("sug_knn(generic)");
      return SUG_OK;
    }
  }
}

// ---- reverse lists ---------------------------------------------------------
// One block per cloud, everything in LDS: counts -> exclusive scan -> fill (atomic cursor)
// -> each destination sorts its own (short) list so float sums downstream are deterministic
// -> coalesced write-out.  LDS: 2N ints + N*k ints (N=1024, k=20: 88 KB).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void knn_reverse_kernel(const int32_t* __restrict__ idx, int N,
                                                            int k, int32_t* __restrict__ rev_off,
                                                            int32_t* __restrict__ rev_ent) {
  extern __shared__ int s_i[];  // cnt[N] | cur[N] | scan scratch[BLOCK] | ent[N*k]
  int* cnt = s_i;
  int* cur = s_i + N;
  int* scr = s_i + 2 * N;
  int* ent = s_i + 2 * N + BLOCK;
  const int b = blockIdx.x;
  const int32_t* ib = idx + (int64_t)b * N * k;
  int32_t* off = rev_off + (int64_t)b * (N + 1);
  int32_t* gent = rev_ent + (int64_t)b * N * k;
  for (int i = threadIdx.x; i < N; i += BLOCK) cnt[i] = 0;
  __syncthreads();
  const int total = N * k;
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int m = ib[e];
    if (m >= 0 && m < N) atomicAdd(&cnt[m], 1);
  }
  __syncthreads();
  const int PER = (N + BLOCK - 1) / BLOCK;
  const int lo = threadIdx.x * PER;
  int local = 0;
  for (int i = lo; i < lo + PER && i < N; ++i) local += cnt[i];
  scr[threadIdx.x] = local;
  __syncthreads();
  for (int o = 1; o < BLOCK; o <<= 1) {
    int t = threadIdx.x >= o ? scr[threadIdx.x - o] : 0;
    __syncthreads();
    scr[threadIdx.x] += t;
    __syncthreads();
  }
  int run = scr[threadIdx.x] - local;
  for (int i = lo; i < lo + PER && i < N; ++i) {
    const int c = cnt[i];
    cur[i] = run;
    off[i] = run;
    run += c;
  }
  const int filled = scr[BLOCK - 1];
  if (threadIdx.x == BLOCK - 1) off[N] = filled;
  __syncthreads();
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int m = ib[e];
    if (m >= 0 && m < N) ent[atomicAdd(&cur[m], 1)] = e;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < N; m += BLOCK) {
    const int n = cnt[m], s = cur[m] - n;
    for (int i = 1; i < n; ++i) {  // insertion sort in LDS, lists average k entries
      const int key = ent[s + i];
      int j = i - 1;
      while (j >= 0 && ent[s + j] > key) {
        ent[s + j + 1] = ent[s + j];
        --j;
      }
      ent[s + j + 1] = key;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < filled; e += BLOCK) gent[e] = ent[e];
}

}  // namespace

extern "C" int sug_knn(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx,
                       void* stream) {
  SUG_REQUIRE(x && idx, "sug_knn: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && C > 0, "sug_knn: bad shape B=%d N=%d C=%d", B, N, C);
  SUG_REQUIRE(k >= 1 && k <= 32 && k <= N, "sug_knn: need 1 <= k <= min(32,N), got k=%d N=%d", k, N);
  SUG_REQUIRE(ldx >= C, "sug_knn: ldx=%lld < C=%d", (long long)ldx, C);
  SUG_REQUIRE(B <= 65535, "sug_knn: B=%d exceeds grid.y", B);
  hipStream_t st = (hipStream_t)stream;
  if (sug_knn_mfma_supported(x, ldx, C, k)) return sug_knn_mfma(x, ldx, B, N, C, k, idx, st);
  if (k <= 16) return dispatch_knn_c<16>(x, ldx, B, N, C, k, idx, st);
  if (k <= 20) return dispatch_knn_c<20>(x, ldx, B, N, C, k, idx, st);
  return dispatch_knn_c<32>(x, ldx, B, N, C, k, idx, st);
}

extern "C" int sug_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off,
                               int32_t* rev_ent, void* stream) {
  SUG_REQUIRE(idx && rev_off && rev_ent, "sug_knn_reverse: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && k > 0, "sug_knn_reverse: bad shape");
  constexpr int BLOCK = 1024;
  size_t sh = (size_t)(2 * N + BLOCK + (size_t)N * k) * sizeof(int);
  SUG_REQUIRE(sh <= 160 * 1024, "sug_knn_reverse: N*k=%d too large for the LDS-resident build", N * k);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_reverse_kernel<BLOCK>, 160 * 1024, "sug_knn_reverse")) return rc;
  hipLaunchKernelGGL((knn_reverse_kernel<BLOCK>), dim3(B), dim3(BLOCK), sh, (hipStream_t)stream, idx,
                     N, k, rev_off, rev_ent);
  SUG_LAUNCH_CHECK("sug_knn_reverse");
  return SUG_OK;
}
