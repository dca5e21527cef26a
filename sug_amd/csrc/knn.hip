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
// A workgroup owns the destinations [lo, hi) of one cloud (RS ranges per cloud, so that 64 clouds fill the chip),
// everything in LDS: it scans the cloud's N*k neighbour entries, counts those that point into its range (and those
// that point below it: its base offset in the cloud's entry array), exclusive scan -> fill (atomic cursor, arbitrary
// order) -> rank sort of every destination's list straight into its final place, so that float sums downstream
// are deterministic: a half-wave per list (lane u counts the entries smaller than entry u: the reads of one list
// are LDS broadcasts with no dependence between them), a whole wave for lists longer than 32 (hub points).
// LDS: 2*Nr + BLOCK/64 ints + the worst case of N*k entries (82 KB).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void knn_reverse_kernel(const int32_t* __restrict__ idx, int N,
                                                            int E, int RS, int sorted, int32_t* __restrict__ rev_off,
                                                            int32_t* __restrict__ rev_ent) {
  // E entries per cloud (N*k for a kNN graph, S*nsample for ball-query lists), destinations in [0, N)
  extern __shared__ int s_i[];  // cnt[Nr] | cur[Nr] | per-wave scratch[2*BLOCK/64] | ent[E]
  constexpr int NW = BLOCK / 64;
  const int Nr = (N + RS - 1) / RS;
  int* cnt = s_i;
  int* cur = s_i + Nr;
  int* scr = s_i + 2 * Nr;
  int* ent = s_i + 2 * Nr + 2 * NW;
  const int b = blockIdx.x / RS, r = blockIdx.x % RS;
  const int lo = r * Nr, hi = (lo + Nr < N) ? lo + Nr : N;
  const int nr = hi > lo ? hi - lo : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* ib = idx + (int64_t)b * E;
  int32_t* off = rev_off + (int64_t)b * (N + 1);
  int32_t* gent = rev_ent + (int64_t)b * E;
  for (int i = threadIdx.x; i < nr; i += BLOCK) cnt[i] = 0;
  __syncthreads();
  const int total = E;
  int below = 0;
  // the neighbour entries are read U at a time (all loads of a batch in flight before the first use: one load per
  // iteration, each waiting for the atomic behind it, made this kernel latency-bound)
  constexpr int U = 5;
  for (int e0 = threadIdx.x; e0 < total; e0 += BLOCK * U) {
    int mv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mv[u] = (e0 + u * BLOCK < total) ? ib[e0 + u * BLOCK] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mv[u];
      if (m >= 0 && m < lo) ++below;
      if (m >= lo && m < hi) atomicAdd(&cnt[m - lo], 1);
    }
  }
  // base = number of valid entries that point below this range (integer sums: order-free)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
  if (lane == 0) scr[wave] = below;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) base += scr[w];
  // exclusive scan of the counts: PER consecutive destinations per thread, wave scan, then across waves
  const int PER = (nr + BLOCK - 1) / BLOCK;
  const int t0 = threadIdx.x * PER;
  int local = 0;
  for (int i = t0; i < t0 + PER && i < nr; ++i) local += cnt[i];
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) scr[NW + wave] = incl;
  __syncthreads();
  int wbase = 0, filled = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int t = scr[NW + w];
    if (w < wave) wbase += t;
    filled += t;
  }
  int run = wbase + incl - local;
  for (int i = t0; i < t0 + PER && i < nr; ++i) {
    const int c = cnt[i];
    cur[i] = run;
    off[lo + i] = base + run;
    run += c;
  }
  if (r == RS - 1 && threadIdx.x == 0) off[N] = base + filled;
  __syncthreads();
  for (int e0 = threadIdx.x; e0 < total; e0 += BLOCK * U) {
    int mv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mv[u] = (e0 + u * BLOCK < total) ? ib[e0 + u * BLOCK] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mv[u];
      if (m >= lo && m < hi) ent[atomicAdd(&cur[m - lo], 1)] = e0 + u * BLOCK;
    }
  }
  __syncthreads();
  // lists up to 32 entries: two per wave (one per half-wave); the loop bound is made uniform over the wave
  // (sorted = 0: the consumer does not need a fixed order -- the lists go out as the atomics left them)
  const int half = lane >> 5, hl = lane & 31;
  for (int i0 = wave * 2; sorted && i0 < nr; i0 += NW * 2) {
    const int i = i0 + half;
    const int n = i < nr ? cnt[i] : 0;
    const int s = i < nr ? cur[i] - n : 0;
    const int nn = n <= 32 ? n : 0;
    const int nmax = max(nn, __shfl_xor(nn, 32));
    const int nlast = nn > 0 ? nn - 1 : 0;
    const int key = ent[s + min(hl, nlast)];
    int rank = 0;
    for (int j0 = 0; j0 < nmax; j0 += 8) {           // 8 independent LDS reads in flight
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ent[s + min(j0 + u, nlast)];   // unconditional (a guarded read is a branch + wait)
#pragma unroll
      for (int u = 0; u < 8; ++u) rank += (j0 + u < nn && v[u] < key) ? 1 : 0;      // entries are distinct
    }
    // in place: every read of this list (same half-wave, program order) is behind us
    if (hl < nn) ent[s + rank] = key;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < filled; e += BLOCK) gent[base + e] = ent[e];
  __syncthreads();            // the long lists below overwrite their (unsorted) copy
  // longer lists (hub points): a wave per list
  for (int i = wave; sorted && i < nr; i += NW) {
    const int n = cnt[i];
    if (n <= 32) continue;                           // uniform over the wave
    const int s = cur[i] - n;
    for (int u = lane; u < n; u += 64) {
      const int key = ent[s + u];
      int rank = 0;
      for (int j0 = 0; j0 < n; j0 += 8) {
        int v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = ent[s + min(j0 + t, n - 1)];
#pragma unroll
        for (int t = 0; t < 8; ++t) rank += (j0 + t < n && v[t] < key) ? 1 : 0;
      }
      gent[base + s + rank] = key;
    }
  }
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
  if (sug_knn_pc_supported(x, ldx, C, k)) return sug_knn_pc(x, ldx, B, N, C, k, idx, st);
  if (k <= 16) return dispatch_knn_c<16>(x, ldx, B, N, C, k, idx, st);
  if (k <= 20) return dispatch_knn_c<20>(x, ldx, B, N, C, k, idx, st);
  return dispatch_knn_c<32>(x, ldx, B, N, C, k, idx, st);
}

// LDS bytes of the reverse-list build for B clouds of E entries into N destinations (and its destination ranges per cloud):
// callers that need to know beforehand whether sug_reverse_lists will take the shape compare this with 160 KB
size_t sug_reverse_lists_lds_bytes(int B, int E, int N, int* ranges) {
  int RS = 1;                                    // destination ranges per cloud: >= 256 workgroups when B allows
  while (RS < 8 && B * RS < 256 && N / (RS * 2) >= 64) RS *= 2;
  const int Nr = (N + RS - 1) / RS;
  if (ranges) *ranges = RS;
  return (size_t)(2 * Nr + 2 * (1024 / 64) + (size_t)E) * sizeof(int);
}

// reverse lists of B clouds with E index entries each, destinations in [0, N)
int sug_reverse_lists(const int32_t* idx, int B, int E, int N, int sorted, int32_t* rev_off, int32_t* rev_ent,
                      hipStream_t st) {
  SUG_REQUIRE(idx && rev_off && rev_ent, "sug_reverse_lists: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && E > 0, "sug_reverse_lists: bad shape");
  constexpr int BLOCK = 1024;
  int RS = 0;
  const size_t sh = sug_reverse_lists_lds_bytes(B, E, N, &RS);
  SUG_REQUIRE(sh <= 160 * 1024, "sug_reverse_lists: %d entries per cloud are too many for the LDS-resident build", E);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_reverse_kernel<BLOCK>, 160 * 1024, "sug_reverse_lists")) return rc;
  hipLaunchKernelGGL((knn_reverse_kernel<BLOCK>), dim3(B * RS), dim3(BLOCK), sh, st, idx, N, E, RS, sorted, rev_off,
                     rev_ent);
  SUG_LAUNCH_CHECK("sug_reverse_lists");
  return SUG_OK;
}

extern "C" int sug_knn_reverse(const int32_t* idx, int B, int N, int k, int32_t* rev_off,
                               int32_t* rev_ent, void* stream) {
  SUG_REQUIRE(B > 0 && N > 0 && k > 0, "sug_knn_reverse: bad shape");
  return sug_reverse_lists(idx, B, N * k, N, 1, rev_off, rev_ent, (hipStream_t)stream);
}
