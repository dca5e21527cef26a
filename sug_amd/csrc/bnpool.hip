// Train-mode BatchNorm (+LeakyReLU/ReLU) on point-major rows, its exact backward, and the fused
// DGCNN tail  BatchNorm1d -> LeakyReLU(0.2) -> [max over points | mean over points]
// (model/Model.py:112-116).  Statistics come from col_reduce_kernel (edgeconv.hip) through
// sug_col_stats; this file adds the elementwise / pooling halves.  All HBM-bound:
//   bn_bwd_apply      reads a, y (8 B/elem) writes dy (4 B/elem)
//   bn_act_pool_fwd   reads y once (4 B/elem), writes 3 x [B,C]
//   pool_bwd_reduce   reads y once; pool_bwd_apply reads y once, writes dy once.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

// float4 variant of bn_bwd_apply_kernel (C % 4 == 0, aligned rows)
__global__ __launch_bounds__(256) void bn_bwd_apply_vec4_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ coef,
                                                                const double* __restrict__ red,
                                                                int64_t rows, int C, float invM,
                                                                float* __restrict__ dy, int64_t lddy) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C4) * 4;
    const int64_t r = e / C4;
    const float4 av = *reinterpret_cast<const float4*>(a + r * C + c);
    const float4 yv = *reinterpret_cast<const float4*>(y + r * ldy + c);
    const float4 sc = *reinterpret_cast<const float4*>(coef + c);
    const float4 mean = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    const float4 rstd = *reinterpret_cast<const float4*>(coef + 3 * C + c);
    float4 o;
    o.x = av.x - sc.x * invM * ((float)red[c + 0] + (yv.x - mean.x) * rstd.x * (float)red[C + c + 0]);
    o.y = av.y - sc.y * invM * ((float)red[c + 1] + (yv.y - mean.y) * rstd.y * (float)red[C + c + 1]);
    o.z = av.z - sc.z * invM * ((float)red[c + 2] + (yv.z - mean.z) * rstd.z * (float)red[C + c + 2]);
    o.w = av.w - sc.w * invM * ((float)red[c + 3] + (yv.w - mean.w) * rstd.w * (float)red[C + c + 3]);
    *reinterpret_cast<float4*>(dy + r * lddy + c) = o;
  }
}

// bn_bwd_apply_vec4_kernel over `rows` rows in domain groups of rows_g: coefficient set and sums of each row's group.
// FROM_G: `a` is the upstream gradient (row stride lda) and a = scale * g * act'(scale*y + shift) is formed here
// (the reduce kernel then writes no [rows, C] tensor: 5 passes over the layer instead of 6).
template <bool FROM_G>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec4_groups_kernel(const float* __restrict__ a, int64_t lda,
                                                                       const float* __restrict__ y, int64_t ldy,
                                                                       const float* __restrict__ coef,
                                                                       const double* __restrict__ red, int64_t rows,
                                                                       int64_t rows_g, int C, float invM, float slope,
                                                                       float* __restrict__ dy, int64_t lddy) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C4) * 4;
    const int64_t r = e / C4;
    const int64_t g = r / rows_g;
    const float* cg = coef + g * 5 * C;
    const double* rg = red + g * 2 * C;
    float4 av = *reinterpret_cast<const float4*>(a + r * lda + c);
    const float4 yv = *reinterpret_cast<const float4*>(y + r * ldy + c);
    const float4 sc = *reinterpret_cast<const float4*>(cg + c);
    const float4 mean = *reinterpret_cast<const float4*>(cg + 2 * C + c);
    const float4 rstd = *reinterpret_cast<const float4*>(cg + 3 * C + c);
    if (FROM_G) {
      const float4 sh = *reinterpret_cast<const float4*>(cg + C + c);
      av.x = sc.x * (av.x * (fmaf(sc.x, yv.x, sh.x) > 0.f ? 1.f : slope));
      av.y = sc.y * (av.y * (fmaf(sc.y, yv.y, sh.y) > 0.f ? 1.f : slope));
      av.z = sc.z * (av.z * (fmaf(sc.z, yv.z, sh.z) > 0.f ? 1.f : slope));
      av.w = sc.w * (av.w * (fmaf(sc.w, yv.w, sh.w) > 0.f ? 1.f : slope));
    }
    float4 o;
    o.x = av.x - sc.x * invM * ((float)rg[c + 0] + (yv.x - mean.x) * rstd.x * (float)rg[C + c + 0]);
    o.y = av.y - sc.y * invM * ((float)rg[c + 1] + (yv.y - mean.y) * rstd.y * (float)rg[C + c + 1]);
    o.z = av.z - sc.z * invM * ((float)rg[c + 2] + (yv.z - mean.z) * rstd.z * (float)rg[C + c + 2]);
    o.w = av.w - sc.w * invM * ((float)rg[c + 3] + (yv.w - mean.w) * rstd.w * (float)rg[C + c + 3]);
    *reinterpret_cast<float4*>(dy + r * lddy + c) = o;
  }
}

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

constexpr int NS = 4;      // row chunks per cloud (more workgroups -> more loads in flight)

__device__ __forceinline__ float lrelu(float u, float slope) { return u > 0.f ? u : u * slope; }

// Workgroup = 64 channels (16 float4 lanes) x 16 row-lanes over one of NS row chunks of a
// cloud.  Partials per (b, chunk, c): max, arg, sum -> pool_combine_kernel.  Scalar fallback
// (C % 4 != 0) uses 64 scalar lanes x 4 row-lanes through the same code path (VEC = 1).
template <int VEC>
__global__ __launch_bounds__(256) void bn_act_pool_part_kernel(const float* __restrict__ y, int64_t ldy,
                                                               const float* __restrict__ coef, int N,
                                                               int C, float slope,
                                                               float* __restrict__ pmax,
                                                               float* __restrict__ psum,
                                                               int32_t* __restrict__ parg) {
  constexpr int LX = 64 / VEC, LY = 256 / LX;
  __shared__ float s_m[LY][64];
  __shared__ float s_s[LY][64];
  __shared__ int s_a[LY][64];
  const int b = blockIdx.y, ch = blockIdx.z;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  const int c = blockIdx.x * 64 + lx * VEC;
  const int n0 = (int)((int64_t)N * ch / NS), n1 = (int)((int64_t)N * (ch + 1) / NS);
  float best[VEC], sum[VEC];
  int bi[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { best[v] = -INFINITY; sum[v] = 0.f; bi[v] = n0; }
  if (c < C) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { sc[v] = coef[c + v]; sh[v] = coef[C + c + v]; }
    const float* p = y + (int64_t)b * N * ldy + c;
#pragma unroll 4
    for (int n = n0 + ly; n < n1; n += LY) {
      float val[VEC];
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)n * ldy);
        val[0] = t.x; val[1] = t.y; val[2] = t.z; val[3] = t.w;
      } else {
        val[0] = p[(int64_t)n * ldy];
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float u = lrelu(fmaf(sc[v], val[v], sh[v]), slope);
        sum[v] += u;
        if (u > best[v]) { best[v] = u; bi[v] = n; }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s_m[ly][lx * VEC + v] = best[v];
    s_s[ly][lx * VEC + v] = sum[v];
    s_a[ly][lx * VEC + v] = bi[v];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float m = s_m[0][threadIdx.x], sm = s_s[0][threadIdx.x];
      int a = s_a[0][threadIdx.x];
      for (int i = 1; i < LY; ++i) {
        sm += s_s[i][threadIdx.x];
        const float mi = s_m[i][threadIdx.x];
        const int ai = s_a[i][threadIdx.x];
        if (mi > m || (mi == m && ai < a)) { m = mi; a = ai; }    // first maximum, like torch.max
      }
      const int64_t o = ((int64_t)b * NS + ch) * C + cc;
      pmax[o] = m; psum[o] = sm; parg[o] = a;
    }
  }
}

__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ pmax,
                                                           const float* __restrict__ psum,
                                                           const int32_t* __restrict__ parg, int B, int N,
                                                           int C, float* __restrict__ omax,
                                                           float* __restrict__ omean, int64_t ldp,
                                                           int32_t* __restrict__ arg) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * C) return;
  const int b = (int)(e / C), c = (int)(e % C);
  float m = -INFINITY, sm = 0.f;
  int a = 0;
  for (int ch = 0; ch < NS; ++ch) {      // ascending chunks: ties keep the lower row
    const int64_t o = ((int64_t)b * NS + ch) * C + c;
    sm += psum[o];
    if (pmax[o] > m) { m = pmax[o]; a = parg[o]; }
  }
  omax[(int64_t)b * ldp + c] = m; omean[(int64_t)b * ldp + c] = sm / (float)N; arg[e] = a;
}

// G[b,n,c] = act'(u) * (gmean[b,c]/N + gmax[b,c]*[n == arg[b,c]]);  partial sums of G and G*xhat
// per (cloud, chunk) into ws rows (ordered fp64 combine by reduce_rows_kernel).
template <int VEC>
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ gmax,
                                                              const float* __restrict__ gmean, int64_t ldp,
                                                              const int32_t* __restrict__ arg, int N,
                                                              int C, float slope, float* __restrict__ ws) {
  constexpr int LX = 64 / VEC, LY = 256 / LX;
  __shared__ float s_g[LY][64];
  __shared__ float s_x[LY][64];
  const int b = blockIdx.y, ch = blockIdx.z;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  const int c = blockIdx.x * 64 + lx * VEC;
  const int n0 = (int)((int64_t)N * ch / NS), n1 = (int)((int64_t)N * (ch + 1) / NS);
  float sg[VEC], sx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { sg[v] = 0.f; sx[v] = 0.f; }
  if (c < C) {
    float sc[VEC], sh[VEC], mean[VEC], rstd[VEC], gm[VEC], gx[VEC];
    int am[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sc[v] = coef[c + v]; sh[v] = coef[C + c + v]; mean[v] = coef[2 * C + c + v]; rstd[v] = coef[3 * C + c + v];
      gm[v] = gmean[(int64_t)b * ldp + c + v] / (float)N;
      gx[v] = gmax[(int64_t)b * ldp + c + v];
      am[v] = arg[(int64_t)b * C + c + v];
    }
    const float* p = y + (int64_t)b * N * ldy + c;
#pragma unroll 4
    for (int n = n0 + ly; n < n1; n += LY) {
      float val[VEC];
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)n * ldy);
        val[0] = t.x; val[1] = t.y; val[2] = t.z; val[3] = t.w;
      } else {
        val[0] = p[(int64_t)n * ldy];
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float u = fmaf(sc[v], val[v], sh[v]);
        const float g = (u > 0.f ? 1.f : slope) * (gm[v] + (n == am[v] ? gx[v] : 0.f));
        sg[v] += g;
        sx[v] = fmaf(g, (val[v] - mean[v]) * rstd[v], sx[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s_g[ly][lx * VEC + v] = sg[v];
    s_x[ly][lx * VEC + v] = sx[v];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float tg = 0.f, tx = 0.f;
      for (int i = 0; i < LY; ++i) { tg += s_g[i][threadIdx.x]; tx += s_x[i][threadIdx.x]; }
      float* w = ws + ((size_t)b * NS + ch) * 2 * C;
      w[cc] = tg;
      w[C + cc] = tx;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(const float* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ coef,
                                                             const double* __restrict__ red,
                                                             const float* __restrict__ gmax,
                                                             const float* __restrict__ gmean, int64_t ldp,
                                                             const int32_t* __restrict__ arg, int B,
                                                             int N, int C, float slope, float invM,
                                                             float* __restrict__ dy, int64_t lddy) {
  const int CV = C / VEC;
  const int64_t total = (int64_t)B * N * CV;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % CV) * VEC;
    const int64_t r = e / CV;
    const int b = (int)(r / N), n = (int)(r - (int64_t)b * N);
    float val[VEC], out[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(y + r * ldy + c);
      val[0] = t.x; val[1] = t.y; val[2] = t.z; val[3] = t.w;
    } else {
      val[0] = y[r * ldy + c];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float sc = coef[c + v];
      const float u = fmaf(sc, val[v], coef[C + c + v]);
      const float g = (u > 0.f ? 1.f : slope) * (gmean[(int64_t)b * ldp + c + v] / (float)N +
                                                 (n == arg[(int64_t)b * C + c + v] ? gmax[(int64_t)b * ldp + c + v] : 0.f));
      const float xhat = (val[v] - coef[2 * C + c + v]) * coef[3 * C + c + v];
      out[v] = sc * (g - invM * ((float)red[c + v] + xhat * (float)red[C + c + v]));
    }
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(dy + r * lddy + c) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
      dy[r * lddy + c] = out[0];
    }
  }
}

// Same, for C/4 dividing 256: a thread owns one float4 channel group of one cloud for its whole
// life -- the nine per-channel constants are loaded once, no integer division per element -- and
// streams over the points of its row chunk (pure read-y / write-dy traffic).
__global__ __launch_bounds__(256) void pool_bwd_apply_rows_kernel(const float* __restrict__ y, int64_t ldy,
                                                                  const float* __restrict__ coef,
                                                                  const double* __restrict__ red,
                                                                  const float* __restrict__ gmax,
                                                                  const float* __restrict__ gmean, int64_t ldp,
                                                                  const int32_t* __restrict__ arg, int N, int C,
                                                                  int rows_per_block, float slope, float invM,
                                                                  float* __restrict__ dy, int64_t lddy) {
  const int CV = C >> 2;
  const int b = blockIdx.y;
  const int cg = threadIdx.x % CV, rl = threadIdx.x / CV, rstep = 256 / CV;
  const int c = cg * 4;
  float sc[4], sh[4], mu[4], rs[4], r0[4], r1[4], gme[4], gmx[4];
  int am[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sc[v] = coef[c + v];
    sh[v] = coef[C + c + v];
    mu[v] = coef[2 * C + c + v];
    rs[v] = coef[3 * C + c + v];
    r0[v] = (float)red[c + v];
    r1[v] = (float)red[C + c + v];
    gme[v] = gmean[(int64_t)b * ldp + c + v] / (float)N;
    gmx[v] = gmax[(int64_t)b * ldp + c + v];
    am[v] = arg[(int64_t)b * C + c + v];
  }
  const int n0 = blockIdx.x * rows_per_block;
  const int n1 = n0 + rows_per_block < N ? n0 + rows_per_block : N;
  const float* yb = y + (int64_t)b * N * ldy + c;
  float* db = dy + (int64_t)b * N * lddy + c;
#pragma unroll 4
  for (int n = n0 + rl; n < n1; n += rstep) {
    const float4 t = *reinterpret_cast<const float4*>(yb + (int64_t)n * ldy);
    const float val[4] = {t.x, t.y, t.z, t.w};
    float out[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float u = fmaf(sc[v], val[v], sh[v]);
      const float g = (u > 0.f ? 1.f : slope) * (gme[v] + (n == am[v] ? gmx[v] : 0.f));
      const float xhat = (val[v] - mu[v]) * rs[v];
      out[v] = sc[v] * (g - invM * (r0[v] + xhat * r1[v]));
    }
    *reinterpret_cast<float4*>(db + (int64_t)n * lddy) = make_float4(out[0], out[1], out[2], out[3]);
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
  const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (lddy % 4 == 0) && ((uintptr_t)a % 16 == 0) &&
                   ((uintptr_t)y % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)coef % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(bn_bwd_apply_vec4_kernel, dim3(ew_grid(rows * C / 4)), dim3(256), 0, (hipStream_t)stream, a,
                       y, ldy, coef, red, rows, C, (float)(1.0 / (double)rows), dy, lddy);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, a, y,
                       ldy, coef, red, rows, C, (float)(1.0 / (double)rows), dy, lddy);
  SUG_LAUNCH_CHECK("sug_bn_bwd_apply");
  return SUG_OK;
}

// all domain groups (rows_g rows each) in one launch; 1 = layout does not allow it.  from_g: `a` is the upstream
// gradient gout (row stride lda) instead of the precomputed a = scale*G
int sug_bn_bwd_apply_groups(const float* a, int64_t lda, int from_g, float slope, const float* y, int64_t ldy,
                            const float* coef, const double* red, int64_t rows_g, int groups, int C, float* dy,
                            int64_t lddy, hipStream_t st) {
  const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && (lddy % 4 == 0) && (lda % 4 == 0) && ((uintptr_t)a % 16 == 0) &&
                   ((uintptr_t)y % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)coef % 16 == 0);
  if (!vec) return 1;
  const int64_t rows = rows_g * groups;
  const float invM = (float)(1.0 / (double)rows_g);
  if (from_g)
    hipLaunchKernelGGL((bn_bwd_apply_vec4_groups_kernel<true>), dim3(ew_grid(rows * C / 4)), dim3(256), 0, st, a, lda, y, ldy,
                       coef, red, rows, rows_g, C, invM, slope, dy, lddy);
  else
    hipLaunchKernelGGL((bn_bwd_apply_vec4_groups_kernel<false>), dim3(ew_grid(rows * C / 4)), dim3(256), 0, st, a, lda, y, ldy,
                       coef, red, rows, rows_g, C, invM, slope, dy, lddy);
  SUG_LAUNCH_CHECK("sug_bn_bwd_apply");
  return SUG_OK;
}

// ---- LayerNorm + (Leaky)ReLU of the FC heads (fc_layer, model/model_utils.py:35-57): rows <= a few hundred,
// C <= 1024.  Forward: one wave per row (mean, biased variance like nn.LayerNorm, two-pass over registers).
// Backward: wave per row for dx, then one thread per column for dgamma / dbeta (rows walked in order:
// bit-reproducible); torch's own pair of kernels for the parameter gradients takes 21 us at [64, 512].
constexpr int LN_MAXV = 16;                     // C <= 64 * 16

__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int rows, int C, float eps,
                                                         float slope, float* __restrict__ y, float* __restrict__ stat) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + (size_t)r * C;
  float v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    v[u] = c < C ? xr[c] : 0.f;
    s += v[u];
  }
  const float mean = wave_sum_f(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    const float d = c < C ? v[u] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(wave_sum_f(q) / (float)C + eps);
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < C) {
      const float t = fmaf((v[u] - mean) * rstd, gamma[c], beta[c]);
      y[(size_t)r * C + c] = t > 0.f ? t : t * slope;
    }
  }
  if (lane == 0) {
    stat[2 * r] = mean;
    stat[2 * r + 1] = rstd;
  }
}

// dx = rstd * (gh - mean_c(gh) - xhat * mean_c(gh * xhat)), gh = g * act' * gamma; ga = g * act' kept for the column pass
__global__ __launch_bounds__(256) void ln_act_bwd_dx_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ stat, int rows, int C, float slope,
                                                            float* __restrict__ dx, float* __restrict__ ga) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float mean = stat[2 * r], rstd = stat[2 * r + 1];
  float xh[LN_MAXV], gh[LN_MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    xh[u] = 0.f;
    gh[u] = 0.f;
    if (c < C) {
      xh[u] = (x[(size_t)r * C + c] - mean) * rstd;
      const float t = fmaf(xh[u], gamma[c], beta[c]);
      const float a = g[(size_t)r * C + c] * (t > 0.f ? 1.f : slope);
      ga[(size_t)r * C + c] = a;
      gh[u] = a * gamma[c];
      s1 += gh[u];
      s2 = fmaf(gh[u], xh[u], s2);
    }
  }
  s1 = wave_sum_f(s1) / (float)C;
  s2 = wave_sum_f(s2) / (float)C;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < C) dx[(size_t)r * C + c] = rstd * (gh[u] - s1 - xh[u] * s2);
  }
}

__global__ __launch_bounds__(256) void ln_act_bwd_param_kernel(const float* __restrict__ ga, const float* __restrict__ x,
                                                               const float* __restrict__ stat, int rows, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float dg = 0.f, db = 0.f;
  for (int r0 = 0; r0 < rows; r0 += 8) {
    float a[8], xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + u < rows ? r0 + u : rows - 1;
      a[u] = ga[(size_t)r * C + c];
      xv[u] = x[(size_t)r * C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (r0 + u >= rows) continue;
      const int r = r0 + u;
      dg = fmaf(a[u], (xv[u] - stat[2 * r]) * stat[2 * r + 1], dg);
      db += a[u];
    }
  }
  dgamma[c] = dg;
  dbeta[c] = db;
}

static bool vec4_ok(const float* y, int64_t ld, int C) {
  return (C % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)y % 16 == 0);
}

extern "C" int sug_bn_act_pool_fwd(const float* y, int64_t ldy, const float* coef, int B, int N, int C,
                                   float slope, float* out_max, float* out_mean, int64_t ld_pool, int32_t* arg,
                                   float* ws, void* stream) {
  SUG_REQUIRE(y && coef && out_max && out_mean && arg && ws, "sug_bn_act_pool_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && C > 0 && ldy >= C && ld_pool >= C && B <= 65535, "sug_bn_act_pool_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const size_t part = (size_t)B * NS * C;
  float* pmax = ws;
  float* psum = ws + part;
  int32_t* parg = reinterpret_cast<int32_t*>(ws + 2 * part);
  dim3 grid(sug_divup(C, 64), B, NS);
  if (vec4_ok(y, ldy, C))
    hipLaunchKernelGGL((bn_act_pool_part_kernel<4>), grid, dim3(256), 0, st, y, ldy, coef, N, C, slope, pmax, psum, parg);
  else
    hipLaunchKernelGGL((bn_act_pool_part_kernel<1>), grid, dim3(256), 0, st, y, ldy, coef, N, C, slope, pmax, psum, parg);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_fwd");
  hipLaunchKernelGGL(pool_combine_kernel, dim3(sug_divup((int64_t)B * C, 256)), dim3(256), 0, st, pmax, psum, parg,
                     B, N, C, out_max, out_mean, ld_pool, arg);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_fwd(combine)");
  return SUG_OK;
}

extern "C" int sug_bn_act_pool_bwd(const float* y, int64_t ldy, const float* coef, const float* gmax,
                                   const float* gmean, int64_t ld_pool, const int32_t* arg, int B, int N, int C,
                                   float slope, int train, double* red, float* ws, float* dy,
                                   int64_t lddy, void* stream) {
  SUG_REQUIRE(y && coef && gmax && gmean && arg && red && ws && dy, "sug_bn_act_pool_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && C > 0 && ldy >= C && lddy >= C && ld_pool >= C && B * NS <= SUG_STATS_BLOCKS,
              "sug_bn_act_pool_bwd: bad shape (B must be <= %d)", SUG_STATS_BLOCKS / NS);
  hipStream_t st = (hipStream_t)stream;
  const bool vec = vec4_ok(y, ldy, C) && vec4_ok(dy, lddy, C);
  dim3 grid(sug_divup(C, 64), B, NS);
  if (vec)
    hipLaunchKernelGGL((pool_bwd_reduce_kernel<4>), grid, dim3(256), 0, st, y, ldy, coef, gmax, gmean, ld_pool, arg, N, C, slope, ws);
  else
    hipLaunchKernelGGL((pool_bwd_reduce_kernel<1>), grid, dim3(256), 0, st, y, ldy, coef, gmax, gmean, ld_pool, arg, N, C, slope, ws);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(reduce)");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(sug_divup(2 * C, 16)), dim3(256), 0, st, ws, B * NS, 2 * C, red);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(combine)");
  const float invM = train ? (float)(1.0 / ((double)B * N)) : 0.f;
  if (vec && (C / 4) <= 256 && 256 % (C / 4) == 0 && B <= 65535) {
    const int rpb = 64 > 256 / (C / 4) ? 64 : 256 / (C / 4);
    hipLaunchKernelGGL(pool_bwd_apply_rows_kernel, dim3(sug_divup(N, rpb), B), dim3(256), 0, st, y, ldy, coef, red, gmax,
                       gmean, ld_pool, arg, N, C, rpb, slope, invM, dy, lddy);
  } else if (vec)
    hipLaunchKernelGGL((pool_bwd_apply_kernel<4>), dim3(ew_grid((int64_t)B * N * C / 4)), dim3(256), 0, st, y, ldy,
                       coef, red, gmax, gmean, ld_pool, arg, B, N, C, slope, invM, dy, lddy);
  else
    hipLaunchKernelGGL((pool_bwd_apply_kernel<1>), dim3(ew_grid((int64_t)B * N * C)), dim3(256), 0, st, y, ldy, coef,
                       red, gmax, gmean, ld_pool, arg, B, N, C, slope, invM, dy, lddy);
  SUG_LAUNCH_CHECK("sug_bn_act_pool_bwd(apply)");
  return SUG_OK;
}


// out[i] = (float) sum_g red[g][i]: the per-domain-group BatchNorm gradient sums (dbeta | dgamma, fp64) of a
// layer folded into the fp32 parameter gradients, in group order.
namespace {
__global__ __launch_bounds__(256) void fold_groups_kernel(const double* __restrict__ red, int groups, int n,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double t = 0.0;
  for (int g = 0; g < groups; ++g) t += red[(int64_t)g * n + i];
  out[i] = (float)t;
}
}  // namespace

extern "C" int sug_fold_groups(const double* red, int groups, int n, float* out, void* stream) {
  SUG_REQUIRE(red && out && groups >= 1 && n > 0, "sug_fold_groups: bad argument");
  hipLaunchKernelGGL(fold_groups_kernel, dim3(sug_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, red, groups, n, out);
  SUG_LAUNCH_CHECK("sug_fold_groups");
  return SUG_OK;
}

extern "C" int sug_ln_act_fwd(const float* x, const float* gamma, const float* beta, int rows, int C, float eps, float slope,
                              float* y, float* stat, void* stream) {
  SUG_REQUIRE(x && gamma && beta && y && stat, "sug_ln_act_fwd: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAXV, "sug_ln_act_fwd: rows=%d C=%d (C <= %d)", rows, C, 64 * LN_MAXV);
  hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(sug_divup(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, rows, C, eps,
                     slope, y, stat);
  SUG_LAUNCH_CHECK("sug_ln_act_fwd");
  return SUG_OK;
}

extern "C" int sug_ln_act_bwd(const float* g, const float* x, const float* gamma, const float* beta, const float* stat, int rows,
                              int C, float slope, float* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
  SUG_REQUIRE(g && x && gamma && beta && stat && dx && dgamma && dbeta && ws, "sug_ln_act_bwd: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && C <= 64 * LN_MAXV, "sug_ln_act_bwd: rows=%d C=%d (C <= %d)", rows, C, 64 * LN_MAXV);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_act_bwd_dx_kernel, dim3(sug_divup(rows, 4)), dim3(256), 0, st, g, x, gamma, beta, stat, rows, C, slope, dx,
                     ws);
  SUG_LAUNCH_CHECK("sug_ln_act_bwd");
  hipLaunchKernelGGL(ln_act_bwd_param_kernel, dim3(sug_divup(C, 256)), dim3(256), 0, st, ws, x, stat, rows, C, dgamma, dbeta);
  SUG_LAUNCH_CHECK("sug_ln_act_bwd(param)");
  return SUG_OK;
}

// ---- CALayer gate: out = x * sigmoid(z) + x (model/Model.py:28-34 behind the second 1x1 conv), the product and the
// sum rounded separately as the reference's two ops are; backward dx = g*s + g, dz = g*x * s*(1-s)
namespace {

__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, int64_t n,
                                                       float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = 1.0f / (1.0f + expf(-z[i]));
  out[i] = __fadd_rn(__fmul_rn(x[i], s), x[i]);
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                       const float* __restrict__ z, int64_t n, float* __restrict__ dx,
                                                       float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = 1.0f / (1.0f + expf(-z[i]));
  const float gi = g[i];
  dx[i] = __fadd_rn(__fmul_rn(gi, s), gi);
  dz[i] = __fmul_rn(__fmul_rn(gi, x[i]), __fmul_rn(__fsub_rn(1.0f, s), s));
}

}  // namespace

extern "C" int sug_gate_fwd(const float* x, const float* z, int64_t n, float* out, void* stream) {
  SUG_REQUIRE(x && z && out, "sug_gate_fwd: null pointer");
  SUG_REQUIRE(n > 0, "sug_gate_fwd: empty input");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)sug_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, x, z, n, out);
  SUG_LAUNCH_CHECK("sug_gate_fwd");
  return SUG_OK;
}

extern "C" int sug_gate_bwd(const float* g, const float* x, const float* z, int64_t n, float* dx, float* dz, void* stream) {
  SUG_REQUIRE(g && x && z && dx && dz, "sug_gate_bwd: null pointer");
  SUG_REQUIRE(n > 0, "sug_gate_bwd: empty input");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)sug_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, g, x, z, n, dx, dz);
  SUG_LAUNCH_CHECK("sug_gate_bwd");
  return SUG_OK;
}

// ---- gate + BatchNorm1d over a handful of rows: the tail of CALayer.forward (model/Model.py:28-34 + :442-449):
//     out = BN(x * sigmoid(z) + x) with batch statistics over the M rows (M = the clouds of one domain, <= 128).
// One thread per channel walks the M rows (statistics two-pass: mean, then squared deviations, as torch's kernel);
// one launch instead of gate + three native BatchNorm launches, and in the backward one instead of the native
// batch_norm_backward_reduce (32 us for [32, 4096]) + elementwise + gate backward.
namespace {

__device__ __forceinline__ float gate_val(float x, float z) {
  const float s = 1.0f / (1.0f + expf(-z));
  return __fadd_rn(__fmul_rn(x, s), x);
}

// MR: rows kept in registers (M <= MR: one batch of loads, the three passes run on registers -- with a loop over the rows every
// pass waited for its own loads: 13.6 us for 512 KB); MR = 0: any M, rows re-read per pass.
template <int MR>
__global__ __launch_bounds__(64) void gate_bn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, int M, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ rmean, float* __restrict__ rvar, int training,
                                                         float eps, float momentum, float* __restrict__ out,
                                                         float* __restrict__ stat) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float gv[MR > 0 ? MR : 1];
  if constexpr (MR > 0) {
    float xv[MR], zv[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int r = i < M ? i : M - 1;
      xv[i] = x[(int64_t)r * C + c];
      zv[i] = z[(int64_t)r * C + c];
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) gv[i] = gate_val(xv[i], zv[i]);
  }
  auto g_at = [&](int i) { if constexpr (MR > 0) return gv[i]; else return gate_val(x[(int64_t)i * C + c], z[(int64_t)i * C + c]); };
  float mean, invstd;
  if (training) {
    float s = 0.f;
    if constexpr (MR > 0) {
#pragma unroll
      for (int i = 0; i < MR; ++i) s += i < M ? gv[i] : 0.f;
    } else {
#pragma unroll 8
      for (int i = 0; i < M; ++i) s += g_at(i);
    }
    mean = s / (float)M;
    float q = 0.f;
    if constexpr (MR > 0) {
#pragma unroll
      for (int i = 0; i < MR; ++i) { const float d = gv[i] - mean; q = i < M ? fmaf(d, d, q) : q; }
    } else {
#pragma unroll 8
      for (int i = 0; i < M; ++i) { const float d = g_at(i) - mean; q = fmaf(d, d, q); }
    }
    const float var = q / (float)M;
    invstd = 1.0f / sqrtf(var + eps);
    if (rmean) {
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (M > 1 ? q / (float)(M - 1) : var);
    }
  } else {
    mean = rmean[c];
    invstd = 1.0f / sqrtf(rvar[c] + eps);
  }
  stat[c] = mean;
  stat[C + c] = invstd;
  const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
  if constexpr (MR > 0) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
      if (i < M) out[(int64_t)i * C + c] = fmaf(gv[i], sc, sh);
  } else {
#pragma unroll 8
    for (int i = 0; i < M; ++i) out[(int64_t)i * C + c] = fmaf(g_at(i), sc, sh);
  }
}

template <int MR>
__global__ __launch_bounds__(64) void gate_bn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         const float* __restrict__ z, int M, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ stat,
                                                         int training, float* __restrict__ dx, float* __restrict__ dz,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float mean = stat[c], invstd = stat[C + c];
  float sb = 0.f, sg = 0.f;
  float xr[MR > 0 ? MR : 1], sr[MR > 0 ? MR : 1], hr[MR > 0 ? MR : 1], gr[MR > 0 ? MR : 1];      // x, sigmoid(z), x_hat, g
  if constexpr (MR > 0) {
    float zr[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int r = i < M ? i : M - 1;
      xr[i] = x[(int64_t)r * C + c];
      zr[i] = z[(int64_t)r * C + c];
      gr[i] = i < M ? g[(int64_t)r * C + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      sr[i] = 1.0f / (1.0f + expf(-zr[i]));
      hr[i] = (__fadd_rn(__fmul_rn(xr[i], sr[i]), xr[i]) - mean) * invstd;
      sb += gr[i];
      sg = fmaf(gr[i], hr[i], sg);
    }
  } else {
#pragma unroll 8
    for (int i = 0; i < M; ++i) {
      const float gi = g[(int64_t)i * C + c];
      const float xh = (gate_val(x[(int64_t)i * C + c], z[(int64_t)i * C + c]) - mean) * invstd;
      sb += gi;
      sg = fmaf(gi, xh, sg);
    }
  }
  dgamma[c] = sg;
  dbeta[c] = sb;
  const float k = gamma[c] * invstd;
  const float mb = training ? sb / (float)M : 0.f, mg = training ? sg / (float)M : 0.f;
  if constexpr (MR > 0) {
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      if (i < M) {
        const float dg = k * (gr[i] - mb - hr[i] * mg);
        dx[(int64_t)i * C + c] = __fadd_rn(__fmul_rn(dg, sr[i]), dg);
        dz[(int64_t)i * C + c] = __fmul_rn(__fmul_rn(dg, xr[i]), __fmul_rn(__fsub_rn(1.0f, sr[i]), sr[i]));
      }
    }
  } else {
#pragma unroll 8
    for (int i = 0; i < M; ++i) {
      const float xv = x[(int64_t)i * C + c], zv = z[(int64_t)i * C + c];
      const float s = 1.0f / (1.0f + expf(-zv));
      const float gv = __fadd_rn(__fmul_rn(xv, s), xv);
      const float xh = (gv - mean) * invstd;
      const float dg = k * (g[(int64_t)i * C + c] - mb - xh * mg);
      dx[(int64_t)i * C + c] = __fadd_rn(__fmul_rn(dg, s), dg);
      dz[(int64_t)i * C + c] = __fmul_rn(__fmul_rn(dg, xv), __fmul_rn(__fsub_rn(1.0f, s), s));
    }
  }
}

}  // namespace

extern "C" int sug_gate_bn_fwd(const float* x, const float* z, int M, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int training, float eps, float momentum, float* out,
                               float* stat, void* stream) {
  SUG_REQUIRE(x && z && gamma && beta && out && stat, "sug_gate_bn_fwd: null pointer");
  SUG_REQUIRE(M >= 1 && M <= 1024 && C >= 1, "sug_gate_bn_fwd: bad shape M=%d C=%d", M, C);
  SUG_REQUIRE(training || (running_mean && running_var), "sug_gate_bn_fwd: eval mode needs the running buffers");
  SUG_REQUIRE(!running_mean == !running_var, "sug_gate_bn_fwd: running_mean and running_var come together");
  if (M <= 32)
    hipLaunchKernelGGL(gate_bn_fwd_kernel<32>, dim3((unsigned)sug_divup(C, 64)), dim3(64), 0, (hipStream_t)stream, x, z, M, C, gamma,
                       beta, running_mean, running_var, training, eps, momentum, out, stat);
  else
    hipLaunchKernelGGL(gate_bn_fwd_kernel<0>, dim3((unsigned)sug_divup(C, 64)), dim3(64), 0, (hipStream_t)stream, x, z, M, C, gamma,
                       beta, running_mean, running_var, training, eps, momentum, out, stat);
  SUG_LAUNCH_CHECK("sug_gate_bn_fwd");
  return SUG_OK;
}

extern "C" int sug_gate_bn_bwd(const float* g, const float* x, const float* z, int M, int C, const float* gamma,
                               const float* stat, int training, float* dx, float* dz, float* dgamma, float* dbeta,
                               void* stream) {
  SUG_REQUIRE(g && x && z && gamma && stat && dx && dz && dgamma && dbeta, "sug_gate_bn_bwd: null pointer");
  SUG_REQUIRE(M >= 1 && M <= 1024 && C >= 1, "sug_gate_bn_bwd: bad shape M=%d C=%d", M, C);
  if (M <= 32)
    hipLaunchKernelGGL(gate_bn_bwd_kernel<32>, dim3((unsigned)sug_divup(C, 64)), dim3(64), 0, (hipStream_t)stream, g, x, z, M, C,
                       gamma, stat, training, dx, dz, dgamma, dbeta);
  else
    hipLaunchKernelGGL(gate_bn_bwd_kernel<0>, dim3((unsigned)sug_divup(C, 64)), dim3(64), 0, (hipStream_t)stream, g, x, z, M, C,
                       gamma, stat, training, dx, dz, dgamma, dbeta);
  SUG_LAUNCH_CHECK("sug_gate_bn_bwd");
  return SUG_OK;
}

// ---- column sums of [R, C] rows (bias gradients): per-workgroup partial rows + an ordered fp64 fold.  torch's
// sum(dim=0) clears a semaphore buffer with a memset for these tall shapes, and memset nodes are what a replayed
// step graph must not contain.
namespace {

template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, int64_t ld, int64_t R, int C,
                                                          float* __restrict__ part) {
  __shared__ float s_p[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float acc = 0.f;
  if (c < C)
    for (int64_t r = (int64_t)blockIdx.y * 8 + ry; r < R; r += (int64_t)gridDim.y * 8) acc += (float)x[r * ld + c];
  s_p[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < C) {
    float t = s_p[0][cx];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += s_p[i][cx];
    part[(int64_t)blockIdx.y * C + c] = t;
  }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, int nblk, int C, float sign,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double t = 0.0;
  for (int b = 0; b < nblk; ++b) t += (double)part[(int64_t)b * C + c];
  out[c] = sign * (float)t;
}

}  // namespace

extern "C" int64_t sug_colsum_workspace(int64_t rows, int C) {
  int64_t nblk = (rows + 63) / 64;
  if (nblk > 128) nblk = 128;
  if (nblk < 1) nblk = 1;
  return nblk * (int64_t)C;
}

extern "C" int sug_colsum(const void* x, int64_t ld, int64_t rows, int C, int dtype, float sign, float* out, float* ws,
                          void* stream) {
  SUG_REQUIRE(x && out && ws, "sug_colsum: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && ld >= C && (dtype == 0 || dtype == 1), "sug_colsum: bad shape / dtype (0 fp32, 1 fp16)");
  const int nblk = (int)(sug_colsum_workspace(rows, C) / C);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(sug_divup(C, 32), nblk);
  if (dtype == 0) hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ld, rows, C, ws);
  else hipLaunchKernelGGL(colsum_part_kernel<__half>, grid, dim3(256), 0, st, (const __half*)x, ld, rows, C, ws);
  SUG_LAUNCH_CHECK("sug_colsum");
  hipLaunchKernelGGL(colsum_fold_kernel, dim3(sug_divup(C, 256)), dim3(256), 0, st, ws, nblk, C, sign, out);
  SUG_LAUNCH_CHECK("sug_colsum(fold)");
  return SUG_OK;
}
