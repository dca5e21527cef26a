// First layer of a PointNet++ set-abstraction MLP without the grouped tensor.
// Replaces, for PointNetSetAbstraction.forward (model/pointnet2_utils.py:176-207):
//   index_points(xyz, idx) - new_xyz, index_points(points, idx), torch.cat, mlp_convs[0], mlp_bns[0], F.relu.
//
// The 1x1 conv acts on [x_j - c_s ; f_j] (x_j a neighbour of the centroid c_s, f_j its features), so
//   W.[x_j - c_s ; f_j] + b = (Wx.x_j + Wf.f_j) - (Wx.c_s - b) = P[j] - Q[s]
// with P one row per POINT ([B,N,C], a small GEMM in the caller) and Q one row per centroid ([B,S,C]):
// the [B,S,ns,3+D] grouped tensor (550 MB at the config-3 sa2 shape), the GEMM over its B*S*ns rows and
// the [B,S,ns,C] pre-activation tensor are never formed.  The kernels here walk the segments
// (b, s) -> ns neighbour indices, gather P rows (a cloud's P matrix stays in L2) and
//   forward:  batch statistics of y = P[idx] - Q (pass 1), z = relu(BN(y)) written once (pass 2);
//   backward: BatchNorm sums from gz and the recomputed y (pass 1), then dy -> dQ[s] = -sum_j dy (registers,
//             plain store) and dP[idx] += dy (float adds into an LDS-resident channel slice of the cloud's dP;
//             unordered, like index_points' own backward in the reference: torch's index_put uses atomics too).
// Lanes: C/4 per segment (float4 of channels), 256/(C/4) segments per workgroup pass; C in {64, 128}.
#include "common.h"

// edgeconv.hip
int sug_reduce_partials(const float* ws, int nblk, int W, double* out, hipStream_t st);

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int JB = 8;      // neighbour rows in flight per lane

// MODE 0: statistics of y;  MODE 1: z = relu(scale*y + shift);
// MODE 2: backward sums (g = gz * [u > 0]; sum g, sum g*xhat);  MODE 3: backward apply (dQ, dP)
template <int MODE>
__global__ __launch_bounds__(256) void sa_first_kernel(
    const float* __restrict__ P, int64_t ldp, const float* __restrict__ Q, const int32_t* __restrict__ idx,
    int N, int S, int ns, int C, int64_t seg0, int64_t seg1, int segs_per_block, const float* __restrict__ coef,
    const double* __restrict__ red, float invM, const float* __restrict__ gz, float* __restrict__ Z,
    float* __restrict__ dP, float* __restrict__ dQ, float* __restrict__ ws) {
  extern __shared__ float s_red[];                 // [slots][2C] (MODE 0, 2)
  const int LPS = C >> 2;                          // lanes per segment
  const int slots = 256 / LPS;
  const int lp = threadIdx.x % LPS, slot = threadIdx.x / LPS;
  const int c = lp * 4;
  int64_t b0 = seg0 + (int64_t)blockIdx.x * segs_per_block;
  int64_t b1 = b0 + segs_per_block;
  if (b1 > seg1) b1 = seg1;
  float4 scale = make_float4(0, 0, 0, 0), shift = scale, mean = scale, rstd = scale, f = scale, db = scale, dg = scale;
  if (MODE != 0) {
    scale = ld4(coef + c); shift = ld4(coef + C + c);
  }
  if (MODE >= 2) {
    mean = ld4(coef + 2 * C + c); rstd = ld4(coef + 3 * C + c);
  }
  if (MODE == 3) {
    f = make_float4(scale.x * invM, scale.y * invM, scale.z * invM, scale.w * invM);
    db = make_float4((float)red[c], (float)red[c + 1], (float)red[c + 2], (float)red[c + 3]);
    dg = make_float4((float)red[C + c], (float)red[C + c + 1], (float)red[C + c + 2], (float)red[C + c + 3]);
  }
  float4 a1 = make_float4(0, 0, 0, 0), a2 = a1;
  for (int64_t seg = b0 + slot; seg < b1; seg += slots) {
    const int64_t b = seg / S;
    const int32_t* ir = idx + seg * ns;
    const float* Pb = P + b * N * ldp + c;
    const float4 q = ld4(Q + seg * C + c);
    float4 dq = make_float4(0, 0, 0, 0);
    for (int j0 = 0; j0 < ns; j0 += JB) {
      int m[JB];
      float4 pv[JB], gv[JB];
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        const int j = j0 + t < ns ? j0 + t : ns - 1;
        m[t] = min(max(ir[j], 0), N - 1);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        pv[t] = ld4(Pb + (int64_t)m[t] * ldp);
        if (MODE >= 2) gv[t] = ld4(gz + (seg * ns + (j0 + t < ns ? j0 + t : ns - 1)) * C + c);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (j0 + t >= ns) continue;
        float4 y;
        y.x = pv[t].x - q.x; y.y = pv[t].y - q.y; y.z = pv[t].z - q.z; y.w = pv[t].w - q.w;
        if (MODE == 0) {
          a1.x += y.x; a1.y += y.y; a1.z += y.z; a1.w += y.w;
          a2.x = fmaf(y.x, y.x, a2.x); a2.y = fmaf(y.y, y.y, a2.y); a2.z = fmaf(y.z, y.z, a2.z); a2.w = fmaf(y.w, y.w, a2.w);
        } else {
          float4 u;
          u.x = fmaf(scale.x, y.x, shift.x); u.y = fmaf(scale.y, y.y, shift.y);
          u.z = fmaf(scale.z, y.z, shift.z); u.w = fmaf(scale.w, y.w, shift.w);
          if (MODE == 1) {
            u.x = u.x > 0.f ? u.x : 0.f; u.y = u.y > 0.f ? u.y : 0.f; u.z = u.z > 0.f ? u.z : 0.f; u.w = u.w > 0.f ? u.w : 0.f;
            st4(Z + (seg * ns + j0 + t) * C + c, u);
          } else {
            float4 g;
            g.x = u.x > 0.f ? gv[t].x : 0.f; g.y = u.y > 0.f ? gv[t].y : 0.f;
            g.z = u.z > 0.f ? gv[t].z : 0.f; g.w = u.w > 0.f ? gv[t].w : 0.f;
            float4 xh;
            xh.x = (y.x - mean.x) * rstd.x; xh.y = (y.y - mean.y) * rstd.y;
            xh.z = (y.z - mean.z) * rstd.z; xh.w = (y.w - mean.w) * rstd.w;
            if (MODE == 2) {
              a1.x += g.x; a1.y += g.y; a1.z += g.z; a1.w += g.w;
              a2.x = fmaf(g.x, xh.x, a2.x); a2.y = fmaf(g.y, xh.y, a2.y);
              a2.z = fmaf(g.z, xh.z, a2.z); a2.w = fmaf(g.w, xh.w, a2.w);
            } else {
              // dy = scale*g - (scale/M) * (dbeta + xhat * dgamma)   (exact train-mode BN gradient; eval: red = 0)
              float4 dy;
              dy.x = scale.x * g.x - f.x * (db.x + xh.x * dg.x);
              dy.y = scale.y * g.y - f.y * (db.y + xh.y * dg.y);
              dy.z = scale.z * g.z - f.z * (db.z + xh.z * dg.z);
              dy.w = scale.w * g.w - f.w * (db.w + xh.w * dg.w);
              dq.x -= dy.x; dq.y -= dy.y; dq.z -= dy.z; dq.w -= dy.w;
              float* dp = dP + (b * N + m[t]) * (int64_t)C + c;
              atomicAdd(dp + 0, dy.x); atomicAdd(dp + 1, dy.y); atomicAdd(dp + 2, dy.z); atomicAdd(dp + 3, dy.w);
            }
          }
        }
      }
    }
    if (MODE == 3) st4(dQ + seg * C + c, dq);
  }
  if (MODE == 0 || MODE == 2) {
    st4(s_red + (size_t)slot * 2 * C + c, a1);
    st4(s_red + (size_t)slot * 2 * C + C + c, a2);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      float acc = 0.f;
      for (int sl = 0; sl < slots; ++sl) acc += s_red[(size_t)sl * 2 * C + i];
      ws[(size_t)blockIdx.x * 2 * C + i] = acc;
    }
  }
}

// Backward apply with the dP slice of one cloud in LDS: workgroup = (cloud, SW-channel slice), SW/4 lanes per
// segment; dy is accumulated into s_dp[point][SW] with LDS float adds (the global-atomic form of MODE 3 above
// runs at the L2's atomic rate: 0.94 ms per call at the config-3 shapes, 6 ms per step) and the slice is
// written once with plain stores, so dP needs no zero fill.  dQ as above (registers, one store per segment).
template <int SW>
__global__ __launch_bounds__(1024) void sa_first_bwd_lds_kernel(
    const float* __restrict__ P, int64_t ldp, const float* __restrict__ Q, const int32_t* __restrict__ idx, int N,
    int S, int ns, int C, int Bg, const float* __restrict__ coef_all, const double* __restrict__ red_all,
    int64_t red_stride, float invM, const float* __restrict__ gz, float* __restrict__ dP, float* __restrict__ dQ) {
  extern __shared__ __attribute__((aligned(16))) float s_dp[];      // [N][SW]
  constexpr int LP = SW / 4;
  constexpr int SLOTS = 1024 / LP;
  const int nslice = C / SW;
  const int b = blockIdx.x / nslice, sl = blockIdx.x % nslice;
  const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
  const int c = sl * SW + lp * 4;
  const float* coef = coef_all + (int64_t)(b / Bg) * 5 * C;
  const double* red = red_all + (int64_t)(b / Bg) * red_stride;
  for (int e = threadIdx.x; e < N * LP; e += 1024) st4(s_dp + (size_t)e * 4, make_float4(0, 0, 0, 0));
  const float4 scale = ld4(coef + c), shift = ld4(coef + C + c), mean = ld4(coef + 2 * C + c), rstd = ld4(coef + 3 * C + c);
  const float4 f = make_float4(scale.x * invM, scale.y * invM, scale.z * invM, scale.w * invM);
  const float4 db = make_float4((float)red[c], (float)red[c + 1], (float)red[c + 2], (float)red[c + 3]);
  const float4 dg = make_float4((float)red[C + c], (float)red[C + c + 1], (float)red[C + c + 2], (float)red[C + c + 3]);
  const float* Pb = P + (int64_t)b * N * ldp + c;
  __syncthreads();
  for (int sg = slot; sg < S; sg += SLOTS) {
    const int64_t seg = (int64_t)b * S + sg;
    const int32_t* ir = idx + seg * ns;
    const float4 q = ld4(Q + seg * C + c);
    float4 dq = make_float4(0, 0, 0, 0);
    for (int j0 = 0; j0 < ns; j0 += JB) {
      int m[JB];
      float4 pv[JB], gv[JB];
#pragma unroll
      for (int t = 0; t < JB; ++t) m[t] = min(max(ir[j0 + t < ns ? j0 + t : ns - 1], 0), N - 1);
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        pv[t] = ld4(Pb + (int64_t)m[t] * ldp);
        gv[t] = ld4(gz + (seg * ns + (j0 + t < ns ? j0 + t : ns - 1)) * C + c);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (j0 + t >= ns) continue;
        const float yx = pv[t].x - q.x, yy = pv[t].y - q.y, yz = pv[t].z - q.z, yw = pv[t].w - q.w;
        const float gx = fmaf(scale.x, yx, shift.x) > 0.f ? gv[t].x : 0.f, gy = fmaf(scale.y, yy, shift.y) > 0.f ? gv[t].y : 0.f;
        const float gz_ = fmaf(scale.z, yz, shift.z) > 0.f ? gv[t].z : 0.f, gw = fmaf(scale.w, yw, shift.w) > 0.f ? gv[t].w : 0.f;
        const float dx = scale.x * gx - f.x * (db.x + (yx - mean.x) * rstd.x * dg.x);
        const float dy = scale.y * gy - f.y * (db.y + (yy - mean.y) * rstd.y * dg.y);
        const float dz = scale.z * gz_ - f.z * (db.z + (yz - mean.z) * rstd.z * dg.z);
        const float dw = scale.w * gw - f.w * (db.w + (yw - mean.w) * rstd.w * dg.w);
        dq.x -= dx; dq.y -= dy; dq.z -= dz; dq.w -= dw;
        float* d = s_dp + (size_t)m[t] * SW + lp * 4;
        atomicAdd(d + 0, dx); atomicAdd(d + 1, dy); atomicAdd(d + 2, dz); atomicAdd(d + 3, dw);
      }
    }
    st4(dQ + seg * C + c, dq);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * LP; e += 1024) {
    const int n = e / LP, l = e % LP;
    st4(dP + ((int64_t)b * N + n) * C + sl * SW + l * 4, ld4(s_dp + (size_t)e * 4));
  }
}

struct Plan {
  int segs_per_block, nblk;
};
inline Plan plan(int64_t segs, int C) {
  const int slots = 256 / (C >> 2);
  int64_t nblk = 1024;                               // <= SUG_STATS_BLOCKS partial rows per group
  int64_t spb = (segs + nblk - 1) / nblk;
  spb = (spb + slots - 1) / slots * slots;           // whole passes
  if (spb < slots) spb = slots;
  return Plan{(int)spb, (int)((segs + spb - 1) / spb)};
}

}  // namespace

extern "C" int sug_sa_first_fwd(const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B, int N, int S,
                                int ns, int C, int groups, const float* gamma, const float* beta, int training,
                                float eps, float momentum, float* running_mean, float* running_var, float* coef,
                                float* Z, float* ws, void* stream) {
  SUG_REQUIRE(P && Q && idx && gamma && beta && coef && Z && ws, "sug_sa_first_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0, "sug_sa_first_fwd: bad shape B=%d N=%d S=%d ns=%d", B, N, S, ns);
  SUG_REQUIRE(C == 64 || C == 128, "sug_sa_first_fwd: C=%d (64 or 128)", C);
  SUG_REQUIRE(groups >= 1 && B % groups == 0, "sug_sa_first_fwd: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(ldp >= C && ldp % 4 == 0 && ((uintptr_t)P % 16) == 0 && ((uintptr_t)Q % 16) == 0 && ((uintptr_t)Z % 16) == 0,
              "sug_sa_first_fwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t segs_g = (int64_t)(B / groups) * S;
  const Plan pl = plan(segs_g, C);
  const size_t sh = (size_t)(256 / (C >> 2)) * 2 * C * sizeof(float);
  for (int g = 0; g < groups; ++g) {
    float* cg = coef + (int64_t)g * 5 * C;
    const int64_t s0 = g * segs_g, s1 = s0 + segs_g;
    if (training) {
      hipLaunchKernelGGL((sa_first_kernel<0>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                         pl.segs_per_block, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, ws);
      SUG_LAUNCH_CHECK("sug_sa_first_fwd(stats)");
      if (int rc = sug_stats_finalize(ws, pl.nblk, C, gamma, beta, (double)segs_g * ns, eps, momentum, running_mean,
                                      running_var, cg, st))
        return rc;
    }
    hipLaunchKernelGGL((sa_first_kernel<1>), dim3(pl.nblk), dim3(256), 0, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                       pl.segs_per_block, cg, nullptr, 0.f, nullptr, Z, nullptr, nullptr, nullptr);
    SUG_LAUNCH_CHECK("sug_sa_first_fwd(apply)");
  }
  return SUG_OK;
}

extern "C" int sug_sa_first_bwd(const float* gz, const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B,
                                int N, int S, int ns, int C, int groups, int training, const float* coef, double* red,
                                float* dP, float* dQ, float* ws, float* dgb, void* stream) {
  SUG_REQUIRE(gz && P && Q && idx && coef && red && dP && dQ && ws, "sug_sa_first_bwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0, "sug_sa_first_bwd: bad shape");
  SUG_REQUIRE(C == 64 || C == 128, "sug_sa_first_bwd: C=%d (64 or 128)", C);
  SUG_REQUIRE(groups >= 1 && B % groups == 0, "sug_sa_first_bwd: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(((uintptr_t)gz % 16) == 0 && ((uintptr_t)dQ % 16) == 0, "sug_sa_first_bwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t segs_g = (int64_t)(B / groups) * S;
  const Plan pl = plan(segs_g, C);
  const size_t sh = (size_t)(256 / (C >> 2)) * 2 * C * sizeof(float);
  const float invM = (float)(1.0 / ((double)segs_g * ns));
  // slice width of the LDS-resident dP: the widest of 64 / 32 / 16 channels that fits 128 KB
  int SW = 0;
  for (int w = 64; w >= 16; w >>= 1)
    if (C % w == 0 && (size_t)N * w * sizeof(float) <= 128 * 1024) { SW = w; break; }
  if (!SW && hipMemsetAsync(dP, 0, (size_t)B * N * C * sizeof(float), st) != hipSuccess) {
    sug_set_error("sug_sa_first_bwd: memset failed");
    return SUG_ERR_LAUNCH;
  }
  for (int g = 0; g < groups; ++g) {
    const float* cg = coef + (int64_t)g * 5 * C;
    double* rg = red + (int64_t)g * 2 * C;
    const int64_t s0 = g * segs_g, s1 = s0 + segs_g;
    hipLaunchKernelGGL((sa_first_kernel<2>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                       pl.segs_per_block, cg, nullptr, 0.f, gz, nullptr, nullptr, nullptr, ws);
    SUG_LAUNCH_CHECK("sug_sa_first_bwd(reduce)");
    if (int rc = sug_reduce_partials(ws, pl.nblk, 2 * C, rg, st)) return rc;
    if (SW) continue;
    // eval mode: the statistics are constants (red + groups*2C: a caller-zeroed spare row)
    const double* ru = training ? rg : red + (int64_t)groups * 2 * C;
    hipLaunchKernelGGL((sa_first_kernel<3>), dim3(pl.nblk), dim3(256), 0, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                       pl.segs_per_block, cg, ru, invM, gz, nullptr, dP, dQ, nullptr);
    SUG_LAUNCH_CHECK("sug_sa_first_bwd(apply)");
  }
  if (SW) {
    const double* ru = training ? red : red + (int64_t)groups * 2 * C;
    const int64_t rstride = training ? 2 * C : 0;
    const size_t shl = (size_t)N * SW * sizeof(float);
    const dim3 grid(B * (C / SW));
#define SA_BWD_LDS(W) do { \
      static SugLdsOptIn note; \
      if (int rc = sug_allow_dynamic_lds(note, &sa_first_bwd_lds_kernel<W>, 128 * 1024, "sug_sa_first_bwd(lds)")) return rc; \
      hipLaunchKernelGGL((sa_first_bwd_lds_kernel<W>), grid, dim3(1024), shl, st, P, ldp, Q, idx, N, S, ns, C, B / groups, \
                         coef, ru, rstride, invM, gz, dP, dQ); \
    } while (0)
    if (SW == 64) SA_BWD_LDS(64);
    else if (SW == 32) SA_BWD_LDS(32);
    else SA_BWD_LDS(16);
#undef SA_BWD_LDS
    SUG_LAUNCH_CHECK("sug_sa_first_bwd(lds)");
  }
  if (dgb) return sug_fold_groups(red, groups, 2 * C, dgb, stream);
  return SUG_OK;
}
