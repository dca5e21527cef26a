// Point Transformer vector attention (d_model = 512, k <= 16 neighbours): the memory-bound parts of
// TransformerBlock.forward, model/Ptran_transformer.py:32-45, around the three 512 x 512 linears of the
// k-expanded rows (which stay library GEMMs on the matrix pipe, fp32 in the parity mode, fp16 MFMA with
// fp32 accumulation in the 16-bit mode of BASELINE config 5):
//
//   pos1   T0[b,i,j,:] = relu(W1 . (xyz_i - xyz_nbr(i,j)) + b1)              fc_delta[0] + ReLU    (:39)
//   qk     U [b,i,j,:] = q[b,i,:] - K[b,nbr(i,j),:] + delta[b,i,j,:]         input of fc_gamma      (:41)
//   attn   mixed[b,i,:] = sum_j softmax_j(L[b,i,j,:] / sqrt(d)) * (V[b,nbr(i,j),:] + delta[b,i,j,:])   (:42-44)
//
// and their gradients.  The reference materialises, per block, the gathered keys / values, q - k,
// q - k + delta, attn / sqrt(d), the softmax, v + delta and their product: 9 tensors of [B,n,k,512];
// here the k-expanded tensors that exist are the GEMM operands / results T0, delta, U, T1, L.
// Layout: rows r = (b*n + i)*k + j, 512 channels contiguous; one wave per row (or per point), a lane
// owns 8 consecutive channels (16-byte fp16 / 2 x 16-byte fp32 accesses).  T = float or __half for the
// k-expanded tensors; q, K, V, xyz, the weights of pos1 and every reduction stay fp32.
// All kernels are HBM-bound: algorithmic bytes = each k-expanded operand once.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int D = 512;

template <typename T>
__device__ __forceinline__ void ld8(const T* __restrict__ p, float (&v)[8]);
template <>
__device__ __forceinline__ void ld8<float>(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8<__half>(const __half* __restrict__ p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const __half2* h = reinterpret_cast<const __half2*>(&u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 f = __half22float2(h[i]);
    v[2 * i] = f.x;
    v[2 * i + 1] = f.y;
  }
}
template <typename T>
__device__ __forceinline__ void st8(T* __restrict__ p, const float (&v)[8]);
template <>
__device__ __forceinline__ void st8<float>(float* __restrict__ p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void st8<__half>(__half* __restrict__ p, const float (&v)[8]) {
  uint4 u;
  __half2* h = reinterpret_cast<__half2*>(&u);
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<uint4*>(p) = u;
}

// a lane's 8 channels of a row as loaded (fp16 rows stay packed: 4 registers instead of 8 while they wait)
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<__half> { uint4 u; };
__device__ __forceinline__ void ldraw(const float* __restrict__ p, Raw8<float>& r) {
  r.a = *reinterpret_cast<const float4*>(p);
  r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void ldraw(const __half* __restrict__ p, Raw8<__half>& r) {
  r.u = *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ void unpack(const Raw8<float>& r, float (&v)[8]) {
  v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void unpack(const Raw8<__half>& r, float (&v)[8]) {
  const __half2* h = reinterpret_cast<const __half2*>(&r.u);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2 f = __half22float2(h[i]);
    v[2 * i] = f.x;
    v[2 * i + 1] = f.y;
  }
}
// exp(l*scale - max) as one fma + v_exp_f32 (arguments <= ~0: no range handling needed; the forward and the
// backward use the same form, so the weights the backward rebuilds are the forward's)
constexpr float LOG2E = 1.44269504088896340736f;
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
// wave index of a 256-thread workgroup as a scalar: row / point addresses become SGPR bases
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float lane_bcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// ---- pos1: T0 = relu(W1 . rel + b1), rel = xyz_i - xyz_nbr.  Wave per point: lane j < k fetches neighbour j's
// offset once (one dependent load chain per point, not per row), the rows then only compute and store.
template <typename T, int KK>
__global__ __launch_bounds__(256) void ptran_pos1_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ W1, const float* __restrict__ b1,
                                                         int64_t P, int n, int k, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nw = (int64_t)gridDim.x * 4;
  float wx[8], wy[8], wz[8], bb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = lane * 8 + u;
    wx[u] = W1[c * 3 + 0]; wy[u] = W1[c * 3 + 1]; wz[u] = W1[c * 3 + 2]; bb[u] = b1[c];
  }
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (lane < k) {
      const int m = nbr[p * k + lane];
      const float* xi = xyz + p * 3;
      const float* xj = xyz + (b * n + m) * 3;
      dx = xi[0] - xj[0]; dy = xi[1] - xj[1]; dz = xi[2] - xj[2];
    }
#pragma unroll
    for (int j = 0; j < KK; ++j) {
      if (j < k) {
        const float ex = lane_bcast(dx, j), ey = lane_bcast(dy, j), ez = lane_bcast(dz, j);
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float t = fmaf(wz[u], ez, fmaf(wy[u], ey, wx[u] * ex)) + bb[u];
          v[u] = t > 0.f ? t : 0.f;
        }
        st8<T>(out + (p * k + j) * D + lane * 8, v);
      }
    }
  }
}

// dW1[c,0..2] = sum_r g[r,c] * [pre(r,c) > 0] * rel(r), db1[c] likewise: per-workgroup partials [nb][4][512]
template <typename T, int KK>
__global__ __launch_bounds__(256) void ptran_pos1_bwd_kernel(const T* __restrict__ g, const float* __restrict__ xyz,
                                                             const int32_t* __restrict__ nbr, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, int64_t P, int n, int k,
                                                             float* __restrict__ part) {
  __shared__ float s_red[4][4][D];
  const int lane = threadIdx.x & 63, wv = wave_in_block();
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
  float wx[8], wy[8], wz[8], bb[8], ax[8], ay[8], az[8], ab[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = lane * 8 + u;
    wx[u] = W1[c * 3 + 0]; wy[u] = W1[c * 3 + 1]; wz[u] = W1[c * 3 + 2]; bb[u] = b1[c];
    ax[u] = ay[u] = az[u] = ab[u] = 0.f;
  }
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (lane < k) {
      const int m = nbr[p * k + lane];
      const float* xi = xyz + p * 3;
      const float* xj = xyz + (b * n + m) * 3;
      dx = xi[0] - xj[0]; dy = xi[1] - xj[1]; dz = xi[2] - xj[2];
    }
#pragma unroll
    for (int j0 = 0; j0 < KK; j0 += 4) {
      Raw8<T> gr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (j0 + t < k) ldraw(g + (p * k + j0 + t) * D + lane * 8, gr[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          const float ex = lane_bcast(dx, j0 + t), ey = lane_bcast(dy, j0 + t), ez = lane_bcast(dz, j0 + t);
          float gv[8];
          unpack(gr[t], gv);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float tt = fmaf(wz[u], ez, fmaf(wy[u], ey, wx[u] * ex)) + bb[u];
            const float gg = tt > 0.f ? gv[u] : 0.f;
            ax[u] = fmaf(gg, ex, ax[u]); ay[u] = fmaf(gg, ey, ay[u]); az[u] = fmaf(gg, ez, az[u]); ab[u] += gg;
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = lane * 8 + u;
    s_red[wv][0][c] = ax[u]; s_red[wv][1][c] = ay[u]; s_red[wv][2][c] = az[u]; s_red[wv][3][c] = ab[u];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 4 * D; e += 256) {
    const int q = e / D, c = e % D;
    part[(int64_t)blockIdx.x * 4 * D + e] = ((s_red[0][q][c] + s_red[1][q][c]) + s_red[2][q][c]) + s_red[3][q][c];
  }
}

// fold the partials in a fixed order (fp64): 32 outputs per workgroup, 8 row groups each, combined in order
__global__ __launch_bounds__(256) void ptran_pos1_fold_kernel(const float* __restrict__ part, int nb, float* __restrict__ dW1,
                                                              float* __restrict__ db1) {
  __shared__ double s_p[8][33];
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + o;
  double t = 0.0;
  int i = grp;
  for (; i + 56 < nb; i += 64) {                 // eight partial rows in flight, added in the plain loop's order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(i + 8 * u) * 4 * D + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += (double)v[u];
  }
  for (; i < nb; i += 8) t += (double)part[(int64_t)i * 4 * D + e];
  s_p[grp][o] = t;
  __syncthreads();
  if (grp == 0) {
    double a = 0.0;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) a += s_p[g8][o];
    const int q = e / D, c = e % D;
    if (q < 3) dW1[c * 3 + q] = (float)a;
    else db1[c] = (float)a;
  }
}

// ---- qk: U = q_i - K_nbr + delta
template <typename T>
__global__ __launch_bounds__(256) void ptran_qk_kernel(const float* __restrict__ q, const float* __restrict__ kf,
                                                       const T* __restrict__ delta, const int32_t* __restrict__ nbr,
                                                       int64_t P, int n, int k, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nw = (int64_t)gridDim.x * 4;
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    float qv[8];
    ld8<float>(q + p * D + lane * 8, qv);
    for (int j0 = 0; j0 < k; j0 += 4) {          // 4 rows' gathers and delta rows in flight
      float kv[4][8];
      Raw8<T> dr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          const int64_t r = p * k + j0 + t;
          const int m = nbr[r];
          ld8<float>(kf + (b * n + m) * D + lane * 8, kv[t]);
          ldraw(delta + r * D + lane * 8, dr[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          float dv[8], o[8];
          unpack(dr[t], dv);
#pragma unroll
          for (int u = 0; u < 8; ++u) o[u] = (qv[u] - kv[t][u]) + dv[u];
          st8<T>(out + (p * k + j0 + t) * D + lane * 8, o);
        }
      }
    }
  }
}

// Column sums (bias gradients) as a by-product of the kernels that produce the [rows, 512] gradient tensors: every
// lane keeps the running sum of its 8 channels over the rows its wave handles; the four waves of a workgroup are
// combined in LDS in wave order and written as one partial row ws[blockIdx.x][512]; ptran_fold_kernel adds the
// partial rows in a fixed order (fp64).  Replaces a separate torch.sum pass over each of these tensors (1 GB at
// block 1 of config 5).
__device__ __forceinline__ void colsum_flush(const float (&cs)[8], float* __restrict__ ws) {
  __shared__ float s_cs[4][D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < 8; ++u) s_cs[wv][lane * 8 + u] = cs[u];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256)
    ws[(size_t)blockIdx.x * D + c] = ((s_cs[0][c] + s_cs[1][c]) + s_cs[2][c]) + s_cs[3][c];
}

__global__ __launch_bounds__(256) void ptran_fold_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ out) {
  __shared__ double s_p[16][17];
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  int b = p;
  for (; b + 112 < nblk; b += 128) {             // eight partial rows in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(b + 16 * u) * D + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (double)v[u];
  }
  for (; b < nblk; b += 16) acc += (double)ws[(size_t)b * D + c];
  s_p[p][cl] = acc;
  __syncthreads();
  if (p == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_p[i][cl];
    out[c] = (float)t;
  }
}

// in place: G <- G * [T1 > 0] (ReLU backward), column sums as above
template <typename T>
__global__ __launch_bounds__(256) void ptran_relu_bwd_kernel(T* __restrict__ G, const T* __restrict__ T1, int64_t R,
                                                             float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  float cs[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) cs[u] = 0.f;
  for (int64_t r = w0; r < R; r += nw) {
    float g[8], t[8];
    ld8<T>(G + r * D + lane * 8, g);
    ld8<T>(T1 + r * D + lane * 8, t);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      g[u] = t[u] > 0.f ? g[u] : 0.f;
      cs[u] += g[u];
    }
    st8<T>(G + r * D + lane * 8, g);
  }
  colsum_flush(cs, ws);
}

// backward of qk, fused with the sum of delta's two gradients: da (in: the attention's gradient of
// delta, out: d delta = dU + da); dq[p] = sum_j dU[p,j]  (dK: ptran_rev_sum_kernel on dU)
template <typename T>
__global__ __launch_bounds__(256) void ptran_qk_bwd_kernel(const T* __restrict__ dU, T* __restrict__ da, int64_t P, int n,
                                                           int k, float* __restrict__ dq, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  float cs[8];                                   // column sums of d delta (ws != nullptr)
#pragma unroll
  for (int u = 0; u < 8; ++u) cs[u] = 0.f;
  for (int64_t p = w0; p < P; p += nw) {
    float aq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) aq[u] = 0.f;
    for (int j = 0; j < k; ++j) {
      const int64_t r = p * k + j;
      float uv[8], av[8];
      ld8<T>(dU + r * D + lane * 8, uv);
      ld8<T>(da + r * D + lane * 8, av);
#pragma unroll
      for (int u = 0; u < 8; ++u) { aq[u] += uv[u]; av[u] += uv[u]; cs[u] += av[u]; }
      st8<T>(da + r * D + lane * 8, av);
    }
    st8<float>(dq + p * D + lane * 8, aq);
  }
  if (ws) colsum_flush(cs, ws);
}

// ---- attn: softmax over the k neighbours (per channel) of L * scale, applied to V_nbr + delta
template <typename T, int KK>
__global__ __launch_bounds__(256) void ptran_attn_kernel(const T* __restrict__ L, const T* __restrict__ delta,
                                                         const float* __restrict__ vf, const int32_t* __restrict__ nbr,
                                                         int64_t P, int n, int k, float scale, float* __restrict__ mixed,
                                                         float* __restrict__ mx, float* __restrict__ sm) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nw = (int64_t)gridDim.x * 4;
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    Raw8<T> z[KK];                               // the k logit rows as loaded, converted twice (max, then exp)
    float zmax[8], zsum[8], acc[8], zl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { zmax[u] = -INFINITY; zsum[u] = 0.f; acc[u] = 0.f; }
#pragma unroll
    for (int j = 0; j < KK; ++j)
      if (j < k) ldraw(L + (p * k + j) * D + lane * 8, z[j]);
#pragma unroll
    for (int j = 0; j < KK; ++j) {
      if (j < k) {
        float zv[8];
        unpack(z[j], zv);
#pragma unroll
        for (int u = 0; u < 8; ++u) zmax[u] = fmaxf(zmax[u], zv[u] * scale);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) zl[u] = -zmax[u] * LOG2E;
#pragma unroll
    for (int j0 = 0; j0 < KK; j0 += 4) {
      float vv[4][8];
      Raw8<T> dr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          const int64_t r = p * k + j0 + t;
          const int m = nbr[r];
          ld8<float>(vf + (b * n + m) * D + lane * 8, vv[t]);
          ldraw(delta + r * D + lane * 8, dr[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          float zv[8], dv[8];
          unpack(z[j0 + t], zv);
          unpack(dr[t], dv);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float e = exp2_hw(fmaf(zv[u], scale * LOG2E, zl[u]));
            zsum[u] += e;
            acc[u] = fmaf(e, vv[t][u] + dv[u], acc[u]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] /= zsum[u];
    st8<float>(mixed + p * D + lane * 8, acc);
    st8<float>(mx + p * D + lane * 8, zmax);
    st8<float>(sm + p * D + lane * 8, zsum);
  }
}

// backward: dL, da (the gradient of delta through v + delta) per row; dV[m] over the reverse list of m.
// With a_j the softmax weights and y_j = V_nbr + delta, dL_j = scale * a_j * (g*y_j - sum_i a_i g*y_i) and the sum is
// g * mixed (the forward's output): one pass over the rows, nothing per row kept in registers.
template <typename T>
__global__ __launch_bounds__(256) void ptran_attn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mixed,
                                                             const T* __restrict__ L, const T* __restrict__ delta,
                                                             const float* __restrict__ vf, const int32_t* __restrict__ nbr,
                                                             const float* __restrict__ mx, const float* __restrict__ sm,
                                                             int64_t P, int n, int k, float scale, T* __restrict__ dL,
                                                             T* __restrict__ da, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + wave_in_block(), nw = (int64_t)gridDim.x * 4;
  float cs[8];                                   // column sums of dL (ws != nullptr)
#pragma unroll
  for (int u = 0; u < 8; ++u) cs[u] = 0.f;
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    float gv[8], zmax[8], rs[8], dot[8];
    ld8<float>(g + p * D + lane * 8, gv);
    ld8<float>(mx + p * D + lane * 8, zmax);
    ld8<float>(sm + p * D + lane * 8, rs);
    ld8<float>(mixed + p * D + lane * 8, dot);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      rs[u] = 1.0f / rs[u];
      dot[u] *= gv[u];
      zmax[u] *= -LOG2E;
    }
    for (int j0 = 0; j0 < k; j0 += 4) {
      Raw8<T> lr[4], dr[4];
      float vv[4][8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          const int64_t r = p * k + j0 + t;
          const int m = nbr[r];
          ldraw(L + r * D + lane * 8, lr[t]);
          ld8<float>(vf + (b * n + m) * D + lane * 8, vv[t]);
          ldraw(delta + r * D + lane * 8, dr[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j0 + t < k) {
          const int64_t r = p * k + j0 + t;
          float lv[8], dlt[8], av[8], o[8];
          unpack(lr[t], lv);
          unpack(dr[t], dlt);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float pr = exp2_hw(fmaf(lv[u], scale * LOG2E, zmax[u])) * rs[u];
            av[u] = gv[u] * pr;
            o[u] = pr * (gv[u] * (vv[t][u] + dlt[u]) - dot[u]) * scale;
            cs[u] += o[u];
          }
          st8<T>(da + r * D + lane * 8, av);
          st8<T>(dL + r * D + lane * 8, o);
        }
      }
    }
  }
  if (ws) colsum_flush(cs, ws);
}

// out[m,:] = sign * sum over the reverse neighbour list of m of src[e,:] (entries ascending: fixed order).
// dV = + sum of da (= g * attn, the attention's gradient of v + delta), dK = - sum of dU.
template <typename T>
__global__ __launch_bounds__(256) void ptran_rev_sum_kernel(const T* __restrict__ src, const int32_t* __restrict__ rev_off,
                                                            const int32_t* __restrict__ rev_ent, int64_t P, int n, int k,
                                                            float sign, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t p = w0; p < P; p += nw) {
    const int64_t b = p / n;
    const int i = (int)(p - b * n);
    const int32_t* off = rev_off + b * (n + 1) + i;
    const int e0 = off[0], e1 = off[1];
    const int32_t* ent = rev_ent + b * (int64_t)n * k;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int t = e0; t < e1; ++t) {
      float v[8];
      ld8<T>(src + (b * (int64_t)n * k + ent[t]) * D + lane * 8, v);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += v[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] *= sign;
    st8<float>(out + p * D + lane * 8, acc);
  }
}

inline int grid_for(int64_t waves) {
  int64_t g = (waves + 3) / 4;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}
// kernels that also emit a partial column-sum row per workgroup: 2048 workgroups (8 per CU: every resident slot
// taken, grid-stride loops balance the rest) keep the partial rows at 4 MB and their fold at a few microseconds
inline int grid_cs(int64_t waves) {
  const int g = grid_for(waves);
  return g > 2048 ? 2048 : g;
}

}  // namespace

#define PT_REQ_COMMON(name)                                                                                          \
  SUG_REQUIRE(B > 0 && n > 0 && k > 0 && k <= 16 && d == D, name ": bad shape (d_model must be 512, k <= 16)");     \
  SUG_REQUIRE(dtype == 0 || dtype == 1, name ": dtype 0 (fp32) or 1 (fp16)")

extern "C" int sug_ptran_pos1_fwd(const float* xyz, const int32_t* nbr, const float* w1, const float* b1, int B, int n,
                                  int k, int d, int dtype, void* out, void* stream) {
  SUG_REQUIRE(xyz && nbr && w1 && b1 && out, "sug_ptran_pos1_fwd: null pointer");
  PT_REQ_COMMON("sug_ptran_pos1_fwd");
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = (int64_t)B * n;
  if (dtype == 0) hipLaunchKernelGGL((ptran_pos1_kernel<float, 16>), dim3(grid_for(P)), dim3(256), 0, st, xyz, nbr, w1, b1, P, n, k, (float*)out);
  else hipLaunchKernelGGL((ptran_pos1_kernel<__half, 16>), dim3(grid_for(P)), dim3(256), 0, st, xyz, nbr, w1, b1, P, n, k, (__half*)out);
  SUG_LAUNCH_CHECK("sug_ptran_pos1_fwd");
  return SUG_OK;
}

extern "C" int sug_ptran_pos1_bwd(const void* g, const float* xyz, const int32_t* nbr, const float* w1, const float* b1,
                                  int B, int n, int k, int d, int dtype, float* dw1, float* db1, float* ws, void* stream) {
  SUG_REQUIRE(g && xyz && nbr && w1 && b1 && dw1 && db1 && ws, "sug_ptran_pos1_bwd: null pointer");
  PT_REQ_COMMON("sug_ptran_pos1_bwd");
  const int64_t R = (int64_t)B * n * k;
  int nb = (int)((R + 63) / 64);
  if (nb > 1024) nb = 1024;                                 // ws: 1024 * 4 * 512 floats
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = (int64_t)B * n;
  if (dtype == 0) hipLaunchKernelGGL((ptran_pos1_bwd_kernel<float, 16>), dim3(nb), dim3(256), 0, st, (const float*)g, xyz, nbr, w1, b1, P, n, k, ws);
  else hipLaunchKernelGGL((ptran_pos1_bwd_kernel<__half, 16>), dim3(nb), dim3(256), 0, st, (const __half*)g, xyz, nbr, w1, b1, P, n, k, ws);
  SUG_LAUNCH_CHECK("sug_ptran_pos1_bwd");
  hipLaunchKernelGGL(ptran_pos1_fold_kernel, dim3(4 * D / 32), dim3(256), 0, st, ws, nb, dw1, db1);
  SUG_LAUNCH_CHECK("sug_ptran_pos1_bwd(fold)");
  return SUG_OK;
}

extern "C" int sug_ptran_qk_fwd(const float* q, const float* kf, const void* delta, const int32_t* nbr, int B, int n, int k,
                                int d, int dtype, void* out, void* stream) {
  SUG_REQUIRE(q && kf && delta && nbr && out, "sug_ptran_qk_fwd: null pointer");
  PT_REQ_COMMON("sug_ptran_qk_fwd");
  const int64_t P = (int64_t)B * n;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(ptran_qk_kernel<float>, dim3(grid_for(P)), dim3(256), 0, st, q, kf, (const float*)delta, nbr, P, n, k, (float*)out);
  else hipLaunchKernelGGL(ptran_qk_kernel<__half>, dim3(grid_for(P)), dim3(256), 0, st, q, kf, (const __half*)delta, nbr, P, n, k, (__half*)out);
  SUG_LAUNCH_CHECK("sug_ptran_qk_fwd");
  return SUG_OK;
}

static int ptran_fold(const float* ws, int nblk, float* out, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(ptran_fold_kernel, dim3(D / 16), dim3(256), 0, st, ws, nblk, out);
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) {
    sug_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));
    return SUG_ERR_LAUNCH;
  }
  return SUG_OK;
}

extern "C" int64_t sug_ptran_colsum_workspace(int64_t rows) {
  return (int64_t)grid_cs(rows) * D;
}

// db (nullable): column sums of d delta [512]; ws: sug_ptran_colsum_workspace(B*n) floats
extern "C" int sug_ptran_qk_bwd(const void* du, void* da, const int32_t* rev_off, const int32_t* rev_ent, int B, int n,
                                int k, int d, int dtype, float* dq, float* dk, float* db, float* ws, void* stream) {
  SUG_REQUIRE(du && da && rev_off && rev_ent && dq && dk, "sug_ptran_qk_bwd: null pointer");
  SUG_REQUIRE(!db || ws, "sug_ptran_qk_bwd: db needs a workspace");
  PT_REQ_COMMON("sug_ptran_qk_bwd");
  const int64_t P = (int64_t)B * n;
  hipStream_t st = (hipStream_t)stream;
  float* w = db ? ws : nullptr;
  if (dtype == 0) {
    hipLaunchKernelGGL(ptran_rev_sum_kernel<float>, dim3(grid_for(P)), dim3(256), 0, st, (const float*)du, rev_off, rev_ent, P, n, k, -1.0f, dk);
    hipLaunchKernelGGL(ptran_qk_bwd_kernel<float>, dim3(grid_cs(P)), dim3(256), 0, st, (const float*)du, (float*)da, P, n, k, dq, w);
  } else {
    hipLaunchKernelGGL(ptran_rev_sum_kernel<__half>, dim3(grid_for(P)), dim3(256), 0, st, (const __half*)du, rev_off, rev_ent, P, n, k, -1.0f, dk);
    hipLaunchKernelGGL(ptran_qk_bwd_kernel<__half>, dim3(grid_cs(P)), dim3(256), 0, st, (const __half*)du, (__half*)da, P, n, k, dq, w);
  }
  SUG_LAUNCH_CHECK("sug_ptran_qk_bwd");
  if (db) return ptran_fold(ws, grid_cs(P), db, st, "sug_ptran_qk_bwd(fold)");
  return SUG_OK;
}

// in place G <- G * [T1 > 0] over `rows` rows of 512 (the ReLU between the two linears of fc_gamma), db = column sums
extern "C" int sug_ptran_relu_bwd_db(void* G, const void* T1, int64_t rows, int d, int dtype, float* db, float* ws,
                                     void* stream) {
  SUG_REQUIRE(G && T1 && db && ws, "sug_ptran_relu_bwd_db: null pointer");
  SUG_REQUIRE(rows > 0 && d == D && (dtype == 0 || dtype == 1), "sug_ptran_relu_bwd_db: bad shape / dtype");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(ptran_relu_bwd_kernel<float>, dim3(grid_cs(rows)), dim3(256), 0, st, (float*)G, (const float*)T1, rows, ws);
  else
    hipLaunchKernelGGL(ptran_relu_bwd_kernel<__half>, dim3(grid_cs(rows)), dim3(256), 0, st, (__half*)G, (const __half*)T1, rows, ws);
  SUG_LAUNCH_CHECK("sug_ptran_relu_bwd_db");
  return ptran_fold(ws, grid_cs(rows), db, st, "sug_ptran_relu_bwd_db(fold)");
}

extern "C" int sug_ptran_attn_fwd(const void* logits, const void* delta, const float* vf, const int32_t* nbr, int B, int n,
                                  int k, int d, int dtype, float scale, float* mixed, float* mx, float* sm, void* stream) {
  SUG_REQUIRE(logits && delta && vf && nbr && mixed && mx && sm, "sug_ptran_attn_fwd: null pointer");
  PT_REQ_COMMON("sug_ptran_attn_fwd");
  const int64_t P = (int64_t)B * n;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL((ptran_attn_kernel<float, 16>), dim3(grid_for(P)), dim3(256), 0, st, (const float*)logits, (const float*)delta, vf, nbr, P, n, k, scale, mixed, mx, sm);
  else hipLaunchKernelGGL((ptran_attn_kernel<__half, 16>), dim3(grid_for(P)), dim3(256), 0, st, (const __half*)logits, (const __half*)delta, vf, nbr, P, n, k, scale, mixed, mx, sm);
  SUG_LAUNCH_CHECK("sug_ptran_attn_fwd");
  return SUG_OK;
}

// mixed = the forward's output; db (nullable): column sums of dlogits [512]; ws: sug_ptran_colsum_workspace(B*n) floats
extern "C" int sug_ptran_attn_bwd(const float* g, const float* mixed, const void* logits, const void* delta, const float* vf,
                                  const int32_t* nbr, const float* mx, const float* sm, const int32_t* rev_off,
                                  const int32_t* rev_ent, int B, int n, int k, int d, int dtype, float scale, void* dlogits,
                                  void* da, float* dv, float* db, float* ws, void* stream) {
  SUG_REQUIRE(g && mixed && logits && delta && vf && nbr && mx && sm && rev_off && rev_ent && dlogits && da && dv,
              "sug_ptran_attn_bwd: null pointer");
  SUG_REQUIRE(!db || ws, "sug_ptran_attn_bwd: db needs a workspace");
  PT_REQ_COMMON("sug_ptran_attn_bwd");
  const int64_t P = (int64_t)B * n;
  hipStream_t st = (hipStream_t)stream;
  float* w = db ? ws : nullptr;
  if (dtype == 0) {
    hipLaunchKernelGGL(ptran_attn_bwd_kernel<float>, dim3(grid_cs(P)), dim3(256), 0, st, g, mixed, (const float*)logits, (const float*)delta, vf, nbr, mx, sm, P, n, k, scale, (float*)dlogits, (float*)da, w);
    hipLaunchKernelGGL(ptran_rev_sum_kernel<float>, dim3(grid_for(P)), dim3(256), 0, st, (const float*)da, rev_off, rev_ent, P, n, k, 1.0f, dv);
  } else {
    hipLaunchKernelGGL(ptran_attn_bwd_kernel<__half>, dim3(grid_cs(P)), dim3(256), 0, st, g, mixed, (const __half*)logits, (const __half*)delta, vf, nbr, mx, sm, P, n, k, scale, (__half*)dlogits, (__half*)da, w);
    hipLaunchKernelGGL(ptran_rev_sum_kernel<__half>, dim3(grid_for(P)), dim3(256), 0, st, (const __half*)da, rev_off, rev_ent, P, n, k, 1.0f, dv);
  }
  SUG_LAUNCH_CHECK("sug_ptran_attn_bwd");
  if (db) return ptran_fold(ws, grid_cs(P), db, st, "sug_ptran_attn_bwd(fold)");
  return SUG_OK;
}
