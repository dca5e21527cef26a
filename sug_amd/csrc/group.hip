// Row gather / scatter-add and grouped max (index_points + max), channel-last rows.
// Reference: model/point_utils.py:60-83, model/pointnet2_utils.py:41-57,
// model/model_utils.py:122-123.  All HBM-bound: one read + one write per element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ feat, int64_t ldf,
                                                          const int32_t* __restrict__ idx, int N,
                                                          int S, int C, int64_t total,
                                                          float* __restrict__ out, int64_t ldo) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t row = e / C;  // b*S + s
    const int b = (int)(row / S);
    const int j = idx[row];
    float v = 0.f;
    if (j >= 0 && j < N) v = feat[((int64_t)b * N + j) * ldf + c];
    out[row * ldo + c] = v;
  }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ g, int64_t ldg,
                                                               const int32_t* __restrict__ idx, int N,
                                                               int S, int C, int64_t total,
                                                               float* __restrict__ dfeat, int64_t ldf) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t row = e / C;
    const int b = (int)(row / S);
    const int j = idx[row];
    if (j >= 0 && j < N) atomicAdd(&dfeat[((int64_t)b * N + j) * ldf + c], g[row * ldg + c]);
  }
}

__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ feat, int64_t ldf,
                                                        const int32_t* __restrict__ idx, int N, int S,
                                                        int ns, int C, int64_t total,
                                                        float* __restrict__ out,
                                                        int32_t* __restrict__ arg) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t row = e / C;  // b*S + s
    const int b = (int)(row / S);
    const int32_t* ir = idx + row * ns;
    float best = -INFINITY;
    int bj = -1;
    // 8 neighbours at a time: all index loads, then all feature loads, then the comparisons in list order
    for (int t0 = 0; t0 < ns; t0 += 8) {
      int j[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j[u] = ir[t0 + u < ns ? t0 + u : ns - 1];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jc = j[u] < 0 ? 0 : (j[u] >= N ? N - 1 : j[u]);
        v[u] = feat[((int64_t)b * N + jc) * ldf + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (t0 + u >= ns || j[u] < 0 || j[u] >= N) continue;
        if (v[u] > best || bj < 0) {
          best = v[u];
          bj = j[u];
        }
      }
    }
    out[e] = bj < 0 ? 0.f : best;
    arg[e] = bj;
  }
}

__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float* __restrict__ g,
                                                            const int32_t* __restrict__ arg, int N,
                                                            int S, int C, int64_t total,
                                                            float* __restrict__ dfeat, int64_t ldf) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t row = e / C;
    const int b = (int)(row / S);
    const int j = arg[e];
    if (j >= 0 && j < N) atomicAdd(&dfeat[((int64_t)b * N + j) * ldf + c], g[e]);
  }
}

inline int ew_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

// dst[r][0..C) = src[r][0..C) for R rows with row strides lds / ldd (floats), C % 4 == 0: a column slice of a wide row
// buffer from a dense tensor or the other way round (torch's strided copy runs this at ~1 TB/s: 30 us for 16.8 MB)
__global__ __launch_bounds__(256) void copy_rows2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                                                          int64_t ldd, int64_t R, int C4) {
  const int64_t total = R * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / C4;
    const int c = (int)(e - r * C4) * 4;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + r * lds + c);
  }
}

// index_points backward without float atomics: dfeat[b, n, :] = sum of g[b, e, :] over the entries e of n's reverse list
// (sug_reverse_lists, sorted: ascending e), every destination row written once -- zero for an empty list, so the caller
// does not zero-fill.  Lanes: C/4 (float4) or C per destination.
template <int V>
__global__ __launch_bounds__(256) void gather_sum_rows_kernel(const float* __restrict__ g, int64_t ldg,
                                                              const int32_t* __restrict__ rev_off,
                                                              const int32_t* __restrict__ rev_ent, int N, int S, int C,
                                                              int64_t total, float* __restrict__ dfeat, int64_t ldf) {
  const int CV = C / V;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t % CV) * V;
    const int64_t row = t / CV;        // b*N + n
    const int64_t b = row / N;
    const int n = (int)(row - b * N);
    const int32_t* off = rev_off + b * (N + 1) + n;
    const int e0 = off[0], e1 = off[1];
    const int32_t* ent = rev_ent + b * S;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int i = e0; i < e1; ++i) {
      const float* gr = g + (b * S + ent[i]) * ldg + c;
      if (V == 4) {
        const float4 q = *reinterpret_cast<const float4*>(gr);
        acc[0] += q.x; acc[1] += q.y; acc[2] += q.z; acc[3] += q.w;
      } else {
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += gr[v];
      }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) dfeat[row * ldf + c + v] = acc[v];
  }
}

// The same without float atomics for S <= 64 groups per cloud (the SA-node module): a wave per (cloud, channel), lane = group s.
// Lanes whose groups picked the same point are combined at the LOWEST such lane in ascending s order, and that lane alone
// stores: one fixed summation order, so the gradient is reproducible bit for bit.  The lowest lane of a point comes from an
// LDS integer atomicMin on a per-wave tag table (integer atomics are order-free); only the lanes that are not their point's
// first (the duplicates) are then walked with uniform readlanes.  The workgroup stages a [S][16-channel] panel of arg / g
// through LDS so that the global reads stay coalesced.
constexpr int GMB_CH = 16;      // channels per workgroup (4 per wave): 4 x more workgroups than a 64-channel panel
__global__ __launch_bounds__(256) void group_max_bwd_ordered_kernel(const float* __restrict__ g,
                                                                    const int32_t* __restrict__ arg, int N, int S, int C,
                                                                    float* __restrict__ dfeat, int64_t ldf) {
  extern __shared__ __attribute__((aligned(16))) int s_gm[];
  int (*s_arg)[GMB_CH + 1] = reinterpret_cast<int (*)[GMB_CH + 1]>(s_gm);                          // [64][17]
  float (*s_g)[GMB_CH + 1] = reinterpret_cast<float (*)[GMB_CH + 1]>(s_gm + 64 * (GMB_CH + 1));    // [64][17]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* tag = s_gm + 2 * 64 * (GMB_CH + 1) + wv * N;                                                // [4][N]
  const int b = blockIdx.y, c0 = blockIdx.x * GMB_CH;
  const int cc = (C - c0 < GMB_CH) ? C - c0 : GMB_CH;
  for (int e = threadIdx.x; e < S * GMB_CH; e += 256) {
    const int sI = e / GMB_CH, c = e % GMB_CH;
    const bool in = c < cc;
    const int64_t o = ((int64_t)b * S + sI) * C + c0 + c;
    s_arg[sI][c] = in ? arg[o] : -1;
    s_g[sI][c] = in ? g[o] : 0.f;
  }
  __syncthreads();
  for (int c = wv; c < cc; c += 4) {
    int j = lane < S ? s_arg[lane][c] : -1;
    if (j >= N) j = -1;
    const float gv = lane < S ? s_g[lane][c] : 0.f;
    if (j >= 0) tag[j] = 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    if (j >= 0) atomicMin(&tag[j], lane);
    __builtin_amdgcn_wave_barrier();
    const int leader = j >= 0 ? tag[j] : lane;
    unsigned long long dup = __ballot(leader != lane);
    float acc = gv;
    while (dup) {                                   // duplicates in ascending lane order, each added at its point's first lane
      const int i = __builtin_ctzll(dup);
      dup &= dup - 1;
      const int li = __builtin_amdgcn_readlane(leader, i);
      const float gi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), i));
      acc = (lane == li) ? acc + gi : acc;
    }
    if (j >= 0 && leader == lane) dfeat[((int64_t)b * N + j) * ldf + c0 + c] = acc;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int sug_gather_rows(const float* feat, int64_t ldf, const int32_t* idx, int B, int N, int S,
                               int C, float* out, int64_t ldo, void* stream) {
  SUG_REQUIRE(feat && idx && out, "sug_gather_rows: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && C > 0 && ldf >= C && ldo >= C, "sug_gather_rows: bad shape");
  const int64_t total = (int64_t)B * S * C;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, feat,
                     ldf, idx, N, S, C, total, out, ldo);
  SUG_LAUNCH_CHECK("sug_gather_rows");
  return SUG_OK;
}

extern "C" int sug_scatter_add_rows(const float* g, int64_t ldg, const int32_t* idx, int B, int N,
                                    int S, int C, float* dfeat, int64_t ldf, void* stream) {
  SUG_REQUIRE(g && idx && dfeat, "sug_scatter_add_rows: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && C > 0 && ldf >= C && ldg >= C, "sug_scatter_add_rows: bad shape");
  const int64_t total = (int64_t)B * S * C;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     g, ldg, idx, N, S, C, total, dfeat, ldf);
  SUG_LAUNCH_CHECK("sug_scatter_add_rows");
  return SUG_OK;
}

extern "C" int64_t sug_scatter_rows_workspace(int B, int N, int S) { return (int64_t)B * (N + 1) + (int64_t)B * S; }

// knn.hip: LDS bytes of sug_reverse_lists for this shape (declared here, not in common.h: the kNN / fused-EdgeConv PMC traffic
// files under profiles/ are keyed by a hash of the sources that include common.h)
size_t sug_reverse_lists_lds_bytes(int B, int E, int N, int* ranges);

extern "C" int sug_scatter_rows_ordered_supported(int B, int N, int S) {
  return (B > 0 && N > 0 && S > 0 && sug_reverse_lists_lds_bytes(B, S, N, nullptr) <= 160 * 1024) ? 1 : 0;
}

extern "C" int sug_scatter_rows_ordered(const float* g, int64_t ldg, const int32_t* idx, int B, int N, int S, int C,
                                        float* dfeat, int64_t ldf, int32_t* ws, void* stream) {
  SUG_REQUIRE(g && idx && dfeat && ws, "sug_scatter_rows_ordered: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && C > 0 && ldf >= C && ldg >= C, "sug_scatter_rows_ordered: bad shape");
  SUG_REQUIRE(sug_scatter_rows_ordered_supported(B, N, S), "sug_scatter_rows_ordered: %d entries per cloud do not fit the LDS-resident "
              "reverse-list build (use sug_scatter_add_rows)", S);
  int32_t* rev_off = ws;
  int32_t* rev_ent = ws + (int64_t)B * (N + 1);
  hipStream_t st = (hipStream_t)stream;
  if (int rc = sug_reverse_lists(idx, B, S, N, 1, rev_off, rev_ent, st)) return rc;
  const bool v4 = C % 4 == 0 && ldg % 4 == 0 && ((uintptr_t)g % 16) == 0;
  const int64_t total = (int64_t)B * N * (v4 ? C / 4 : C);
  if (v4)
    hipLaunchKernelGGL((gather_sum_rows_kernel<4>), dim3(ew_grid(total)), dim3(256), 0, st, g, ldg, rev_off, rev_ent, N, S, C, total,
                       dfeat, ldf);
  else
    hipLaunchKernelGGL((gather_sum_rows_kernel<1>), dim3(ew_grid(total)), dim3(256), 0, st, g, ldg, rev_off, rev_ent, N, S, C, total,
                       dfeat, ldf);
  SUG_LAUNCH_CHECK("sug_scatter_rows_ordered");
  return SUG_OK;
}

extern "C" int sug_group_max(const float* feat, int64_t ldf, const int32_t* idx, int B, int N, int S,
                             int ns, int C, float* out, int32_t* arg, void* stream) {
  SUG_REQUIRE(feat && idx && out && arg, "sug_group_max: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0 && C > 0 && ldf >= C, "sug_group_max: bad shape");
  const int64_t total = (int64_t)B * S * C;
  hipLaunchKernelGGL(group_max_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, feat,
                     ldf, idx, N, S, ns, C, total, out, arg);
  SUG_LAUNCH_CHECK("sug_group_max");
  return SUG_OK;
}

extern "C" int sug_group_max_bwd(const float* g, const int32_t* arg, int B, int N, int S, int C,
                                 float* dfeat, int64_t ldf, void* stream) {
  SUG_REQUIRE(g && arg && dfeat, "sug_group_max_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && C > 0 && ldf >= C, "sug_group_max_bwd: bad shape");
  const int64_t total = (int64_t)B * S * C;
  static const int unordered = getenv("SUG_GROUP_MAX_UNORDERED") ? atoi(getenv("SUG_GROUP_MAX_UNORDERED")) : 0;
  const size_t sh = ((size_t)2 * 64 * (GMB_CH + 1) + (size_t)4 * N) * sizeof(int);
  bool ordered = S <= 64 && B <= 65535 && sh <= 150 * 1024 && !unordered;
  if (ordered && sh > 64 * 1024) {      // more than the default dynamic-LDS limit: opt in.  A device that refuses (none of the gfx950
    // parts) FAILS the call, as sug_node_offset_bwd does: the summation order of a gradient must not depend on which kernel
    // happened to hit a limit; SUG_GROUP_MAX_UNORDERED=1 selects the atomic form explicitly
    static SugLdsOptIn note;
    if (int rc = sug_allow_dynamic_lds(note, &group_max_bwd_ordered_kernel, 150 * 1024, "sug_group_max_bwd")) return rc;
  }
  if (ordered) {      // fixed summation order; dfeat must arrive ZERO-FILLED in both forms (sug_amd.h: this one stores the touched
                      // entries, the atomic one adds -- on a zeroed buffer the same result, on any other an unspecified one)
    hipLaunchKernelGGL(group_max_bwd_ordered_kernel, dim3(sug_divup(C, GMB_CH), B), dim3(256), sh, (hipStream_t)stream, g, arg, N, S, C,
                       dfeat, ldf);
    SUG_LAUNCH_CHECK("sug_group_max_bwd");
    return SUG_OK;
  }
  hipLaunchKernelGGL(group_max_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, g,
                     arg, N, S, C, total, dfeat, ldf);
  SUG_LAUNCH_CHECK("sug_group_max_bwd");
  return SUG_OK;
}

extern "C" int sug_copy_rows2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int C, void* stream) {
  SUG_REQUIRE(src && dst && rows > 0 && C > 0 && C % 4 == 0 && lds >= C && ldd >= C && lds % 4 == 0 && ldd % 4 == 0,
              "sug_copy_rows2d: bad shape rows=%lld C=%d lds=%lld ldd=%lld", (long long)rows, C, (long long)lds, (long long)ldd);
  SUG_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "sug_copy_rows2d: operands must be 16-byte aligned");
  const int64_t total = rows * (C / 4);
  const int grid = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
  hipLaunchKernelGGL(copy_rows2d_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, C / 4);
  SUG_LAUNCH_CHECK("sug_copy_rows2d");
  return SUG_OK;
}
