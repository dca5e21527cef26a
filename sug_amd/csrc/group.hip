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
