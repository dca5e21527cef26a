// Train-mode BatchNorm (+LeakyReLU/ReLU) on point-major rows, its exact backward, and the fused
// DGCNN tail  BatchNorm1d -> LeakyReLU(0.2) -> [max over points | mean over points]
// (model/Model.py:112-116).  Statistics come from col_reduce_kernel (edgeconv.hip) through
// sug_col_stats; this file adds the elementwise / pooling halves.  All HBM-bound:
//   bn_bwd_apply      reads a, y (8 B/elem) writes dy (4 B/elem)
//   bn_act_pool_fwd   reads y once (4 B/elem), writes 3 x [B,C]
//   pool_bwd_reduce   reads y once; pool_bwd_apply reads y once, writes dy once.
#include "common.h"

namespace {

// dy = a - (scale/M) * (dbeta + xhat * dgamma),  a = scale * G  (exact BN gradient)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ coef,
                                                           const double* __restrict__ red, int64_t rows,
                                                           int C, float invM, float* __restrict__ dy,
                                                           int64_t lddy) {
  const int64_t total = rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t r = e / C;
    const float f = coef[c] * invM;
    const float xhat = (y[r * ldy + c] - coef[2 * C + c]) * coef[3 * C + c];
    dy[r * lddy + c] = a[r * C + c] - f * ((float)red[c] + xhat * (float)red[C + c]);
  }
}

// One workgroup: 64 channels x all N rows of one cloud (4 row phases, LDS combine).
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ coef, int N,
                                                              int C, float slope,
                                                              float* __restrict__ omax,
                                                              float* __restrict__ omean,
                                                              int32_t* __restrict__ arg) {
  __shared__ float s_m[4][64];
  __shared__ float s_s[4][64];
  __shared__ int s_a[4][64];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float best = -INFINITY, sum = 0.f;
  int bi = 0;
  if (c < C) {
    const float sc = coef[c], sh = coef[C + c];
    const float* p = y + (int64_t)b * N * ldy + c;
    for (int n = ph; n < N; n += 4) {
      float u = fmaf(sc, p[(int64_t)n * ldy], sh);
      u = u > 0.f ? u : u * slope;
      sum += u;
      if (u > best) {
        best = u;
        bi = n;
      }
    }
  }
  s_m[ph][cl] = best;
  s_s[ph][cl] = sum;
  s_a[ph][cl] = bi;
  __syncthreads();
  if (ph == 0 && c < C) {
    float m = s_m[0][cl], s = s_s[0][cl];
    int a = s_a[0][cl];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      s += s_s[i][cl];
      const float mi = s_m[i][cl];
      const int ai = s_a[i][cl];
      if (mi > m || (mi == m && ai < a)) {      // first maximum, like torch.max
        m = mi;
        a = ai;
      }
    }
    omax[(int64_t)b * C + c] = m;
    omean[(int64_t)b * C + c] = s / (float)N;
    arg[(int64_t)b * C + c] = a;
  }
}

// G[b,n,c] = act'(u) * (gmean[b,c]/N + gmax[b,c]*[n == arg[b,c]]);  per-WG partial sums of G and
// G*xhat into ws (ordered combine by sug's reduce_partials), layout as col_reduce_kernel.
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ gmax,
                                                              const float* __restrict__ gmean,
                                                              const int32_t* __restrict__ arg, int N,
                                                              int C, float slope, float* __restrict__ ws) {
  __shared__ float s_g[4][64];
  __shared__ float s_x[4][64];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float sg = 0.f, sx = 0.f;
  if (c < C) {
    const float sc = coef[c], sh = coef[C + c], mean = coef[2 * C + c], rstd = coef[3 * C + c];
    const float gm = gmean[(int64_t)b * C + c] / (float)N, gx = gmax[(int64_t)b * C + c];
    const int am = arg[(int64_t)b * C + c];
    const float* p = y + (int64_t)b * N * ldy + c;
    for (int n = ph; n < N; n += 4) {
      const float yv = p[(int64_t)n * ldy];
      const float u = fmaf(sc, yv, sh);
      const float g = (u > 0.f ? 1.f : slope) * (gm + (n == am ? gx : 0.f));
      sg += g;
      sx = fmaf(g, (yv - mean) * rstd, sx);
    }
  }
  s_g[ph][cl] = sg;
  s_x[ph][cl] = sx;
  __syncthreads();
  if (ph == 0 && c < C) {
    const float tg = ((s_g[0][cl] + s_g[1][cl]) + s_g[2][cl]) + s_g[3][cl];
    const float tx = ((s_x[0][cl] + s_x[1][cl]) + s_x[2][cl]) + s_x[3][cl];
    float* w = ws + (size_t)blockIdx.y * 2 * C;      // one partial row per cloud
    w[c] = tg;
    w[C + c] = tx;
  }
}

__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(const float* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ coef,
                                                             const double* __restrict__ red,
                                                             const float* __restrict__ gmax,
                                                             const float* __restrict__ gmean,
                                                             const int32_t* __restrict__ arg, int B,
                                                             int N, int C, float slope, float invM,
                                                             float* __restrict__ dy, int64_t lddy) {
  const int64_t total = (int64_t)B * N * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t r = e / C;
    const int b = (int)(r / N), n = (int)(r - (int64_t)b * N);
    const float sc = coef[c];
    const float yv = y[r * ldy + c];
    const float u = fmaf(sc, yv, coef[C + c]);
    const float g = (u > 0.f ? 1.f : slope) *
                    (gmean[(int64_t)b * C + c] / (float)N + (n == arg[(int64_t)b * C + c] ? gmax[(int64_t)b * C + c] : 0.f));
    const float xhat = (yv - coef[2 * C + c]) * coef[3 * C + c];
    dy[r * lddy + c] = sc * (g - invM * ((float)red[c] + xhat * (float)red[C + c]));
  }
}

// out[i] = sum over rows (ascending, 16 strided partials) of ws[row][i] in fp64 (same scheme as
// edgeconv.hip's reduce_partials_kernel).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ ws, int nrow, int W,
                                                          double* __restrict__ out) {
  __shared__ double s_p[16][17];
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < W)
    for (int b = p; b < nrow; b += 16) acc += (double)ws[(size_t)b * W + c];
  s_p[p][cl] = acc;
  __syncthreads();
  if (p == 0 && c < W) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_p[i][cl];
    out[c] = t;
  }
}

inline int ew_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int sug_bn_bwd_apply(const float* a, const float* y, int64_t ldy, const float* coef,
                                const double* red, int64_t rows, int C, float* dy, int64_t lddy,
                                void* stream) {
  SUG_REQUIRE(a && y && coef && red && dy, "sug_bn_bwd_apply: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && ldy >= C && lddy >= C, "sug_bn_bwd_apply: bad shape");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, a, y,
                     ldy, coef, red, rows, C, (float)(1.0 / (double)rows), dy, lddy);
  SUG_LAUNCH_CHECK("sug_bn_bwd_apply");
  return SUG_OK;
}

extern "C" int sug_bn_act_pool_fwd(const float* y, int64_t ldy, const float* coef, int B, int N, int C,
                                   float slope, float* out_max, float* out_mean, int32_t* arg,
                                   void* stream) {
  SUG_REQUIRE(y && coef && out_max && out_mean && arg, "sug_bn_act_pool_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && C > 0 && ldy >= C && B <= 65535, "sug_bn_act_pool_fwd: bad shape");
  hipLaunchKernelGGL(bn_act_pool_fwd_kernel, dim3(sug_divup(C, 64), B), dim3(256), 0, (hipStream_t)stream,
                     y, ldy, coef, N, C, slope, out_max, out_mean, arg);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_fwd");
  return SUG_OK;
}

extern "C" int sug_bn_act_pool_bwd(const float* y, int64_t ldy, const float* coef, const float* gmax,
                                   const float* gmean, const int32_t* arg, int B, int N, int C,
                                   float slope, int train, double* red, float* ws, float* dy,
                                   int64_t lddy, void* stream) {
  SUG_REQUIRE(y && coef && gmax && gmean && arg && red && ws && dy, "sug_bn_act_pool_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && C > 0 && ldy >= C && lddy >= C && B <= SUG_STATS_BLOCKS,
              "sug_bn_act_pool_bwd: bad shape (B must be <= %d)", SUG_STATS_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pool_bwd_reduce_kernel, dim3(sug_divup(C, 64), B), dim3(256), 0, st, y, ldy, coef, gmax,
                     gmean, arg, N, C, slope, ws);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(reduce)");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(sug_divup(2 * C, 16)), dim3(256), 0, st, ws, B, 2 * C, red);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(combine)");
  hipLaunchKernelGGL(pool_bwd_apply_kernel, dim3(ew_grid((int64_t)B * N * C)), dim3(256), 0, st, y, ldy, coef,
                     red, gmax, gmean, arg, B, N, C, slope, train ? (float)(1.0 / ((double)B * N)) : 0.f, dy, lddy);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(apply)");
  return SUG_OK;
}
