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
//   backward: BatchNorm sums from gz and the recomputed y, and per segment sum_j g and sum_j y (pass 1); then, by
//             DESTINATION point over the reverse lists of the ball-query lists (sug_reverse_lists, sorted since round 4),
//             dP[m] = sum over the rows that gathered m of dy -- plain stores, no atomics (global float atomics
//             ran this pass at 0.94 ms per call, LDS float adds at 0.45 ms) -- and dQ[s] = -sum_j dy[s,j] in closed
//             form from the per-segment sums (dy is affine in g and y).  The order of a point's sum follows the
//             sorted reverse list (ascending row): a fixed order, where index_points' backward in the reference is atomics.
// Lanes: C/4 per segment (float4 of channels), 256/(C/4) segments per workgroup pass; C in {64, 128}.
//
// Round 4, the "geometric" form (Geo argument; opt-in, SUG_SA_FIRST_GEO=1): the coordinate part of the layer taken from the
// DIFFERENCE the reference forms,
//     y = Pf[j] + b + Wx . (x_j - c_s)          (Pf = Wf . f_j per point, or absent when the layer has no input features)
// with 3 fma per channel and neighbour instead of P[j] - Q[s] = (Wx.x_j + Wf.f_j) - (Wx.c_s - b).  Written to test whether
// the cancellation in P - Q explains PointNet++'s gradient scatter against fp64 -- it does not: against fp64 the
// pre-activations of the two forms and of the reference composition in fp32 are equally accurate (6.6e-7 / 7.3e-7 / 7.6e-7
// on unit-scale clouds with 0.2 balls, tests/test_gpu_sagroup.py), the 16-seed gradient statistic is the same within its
// scatter (median ratio 1.28 vs 1.01, tests/diagnostics/diag_pn2_seeds.py), and the step is 1 % slower (18.2 vs 18.0 ms at
// config 3: three more gathers per neighbour).  Hence off by default.  Gradients are unchanged in form:
// dP[m] = sum dy (to Pf and to Wx.x as autograd sees them), dQ[s] = -sum_j dy.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int JB = 8;      // neighbour rows in flight per lane

// geometric form: coordinates of the points / centroids, the coordinate columns of the weight and the bias
struct Geo {
  const float* xyz;    // [B, N, 3] or null (= the P - Q form)
  const float* cent;   // [B, S, 3]
  const float* Wx;     // [C, 3] (row stride ldw)
  const float* bias;   // [C] or null
  int ldw;
  int has_p;           // P holds the feature part Pf (0: the layer has no input features)
};
struct GeoLane {       // this lane's 4 channels
  float w[4][3];
  float4 b;
};
__device__ __forceinline__ GeoLane geo_lane(const Geo& g, int c) {
  GeoLane L;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int d = 0; d < 3; ++d) L.w[q][d] = g.Wx[(c + q) * g.ldw + d];
  L.b = g.bias ? ld4(g.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  return L;
}
// y of one (centroid, neighbour) pair from the feature part pf (zero without features) and the offset d = x_j - c_s
__device__ __forceinline__ float4 geo_y(const GeoLane& L, const float4& pf, float dx, float dy, float dz) {
  float4 y;
  y.x = pf.x + fmaf(L.w[0][2], dz, fmaf(L.w[0][1], dy, fmaf(L.w[0][0], dx, L.b.x)));
  y.y = pf.y + fmaf(L.w[1][2], dz, fmaf(L.w[1][1], dy, fmaf(L.w[1][0], dx, L.b.y)));
  y.z = pf.z + fmaf(L.w[2][2], dz, fmaf(L.w[2][1], dy, fmaf(L.w[2][0], dx, L.b.z)));
  y.w = pf.w + fmaf(L.w[3][2], dz, fmaf(L.w[3][1], dy, fmaf(L.w[3][0], dx, L.b.w)));
  return y;
}

// MODE 0: statistics of y;  MODE 1: z = relu(scale*y + shift);
// MODE 2: backward sums (g = gz * [u > 0]; sum g, sum g*xhat; per segment sum_j g -> segsum[0], sum_j y -> segsum[1])
template <int MODE, bool GEO>
__global__ __launch_bounds__(256) void sa_first_kernel(
    const float* __restrict__ P, int64_t ldp, const float* __restrict__ Q, const int32_t* __restrict__ idx,
    int N, int S, int ns, int C, int64_t seg0, int64_t seg1, int segs_per_block, const float* __restrict__ coef,
    const float* __restrict__ gz, float* __restrict__ Z, float* __restrict__ segsum, int64_t segsum_stride,
    float* __restrict__ ws, Geo geo) {
  extern __shared__ float s_red[];                 // [slots][2C] (MODE 0, 2)
  const int LPS = C >> 2;                          // lanes per segment
  const int slots = 256 / LPS;
  const int lp = threadIdx.x % LPS, slot = threadIdx.x / LPS;
  const int c = lp * 4;
  int64_t b0 = seg0 + (int64_t)blockIdx.x * segs_per_block;
  int64_t b1 = b0 + segs_per_block;
  if (b1 > seg1) b1 = seg1;
  float4 scale = make_float4(0, 0, 0, 0), shift = scale, mean = scale, rstd = scale;
  if (MODE != 0) {
    scale = ld4(coef + c); shift = ld4(coef + C + c);
  }
  if (MODE >= 2) {
    mean = ld4(coef + 2 * C + c); rstd = ld4(coef + 3 * C + c);
  }
  float4 a1 = make_float4(0, 0, 0, 0), a2 = a1;
  constexpr bool use_geo = GEO;            // compile-time: the P - Q instantiation carries none of the geometric code
  GeoLane GL;
  if constexpr (GEO) GL = geo_lane(geo, c);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // MODE 0: sums about the pivot y of the group's first row (common.h); workgroup 0 publishes it for the finalize
  float4 piv = make_float4(0, 0, 0, 0);
  if (MODE == 0) {
    const int m0 = min(max(idx[seg0 * ns], 0), N - 1);
    if (use_geo) {
      const float* xm = geo.xyz + ((seg0 / S) * N + m0) * 3;
      const float* cs = geo.cent + seg0 * 3;
      const float4 p0 = geo.has_p ? ld4(P + ((seg0 / S) * N + m0) * ldp + c) : zero4;
      piv = geo_y(GL, p0, xm[0] - cs[0], xm[1] - cs[1], xm[2] - cs[2]);
    } else {
      const float4 p0 = ld4(P + ((seg0 / S) * N + m0) * ldp + c), q0 = ld4(Q + seg0 * C + c);
      piv = make_float4(p0.x - q0.x, p0.y - q0.y, p0.z - q0.z, p0.w - q0.w);
    }
    if (blockIdx.x == 0 && slot == 0) st4(ws + SUG_PIVOT_OFFSET(C) + c, piv);
  }
  for (int64_t seg = b0 + slot; seg < b1; seg += slots) {
    const int64_t b = seg / S;
    const int32_t* ir = idx + seg * ns;
    const float* Pb = P + b * N * ldp + c;
    const float4 q = use_geo ? zero4 : ld4(Q + seg * C + c);
    float cx = 0.f, cy = 0.f, cz = 0.f;
    const float* xb = nullptr;
    if (use_geo) {
      const float* cs = geo.cent + seg * 3;
      cx = cs[0]; cy = cs[1]; cz = cs[2];
      xb = geo.xyz + b * N * 3;
    }
    float4 sgs = make_float4(0, 0, 0, 0), sys = sgs;
    for (int j0 = 0; j0 < ns; j0 += JB) {
      int m[JB];
      float4 pv[JB], gv[JB];
      float px[JB], py[JB], pz[JB];
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        const int j = j0 + t < ns ? j0 + t : ns - 1;
        m[t] = min(max(ir[j], 0), N - 1);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (use_geo) {
          const float* xm = xb + (int64_t)m[t] * 3;
          px[t] = xm[0]; py[t] = xm[1]; pz[t] = xm[2];
          pv[t] = geo.has_p ? ld4(Pb + (int64_t)m[t] * ldp) : zero4;
        } else {
          pv[t] = ld4(Pb + (int64_t)m[t] * ldp);
        }
        if (MODE >= 2) gv[t] = ld4(gz + (seg * ns + (j0 + t < ns ? j0 + t : ns - 1)) * C + c);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (j0 + t >= ns) continue;
        float4 y;
        if (use_geo) {
          y = geo_y(GL, pv[t], px[t] - cx, py[t] - cy, pz[t] - cz);
        } else {
          y.x = pv[t].x - q.x; y.y = pv[t].y - q.y; y.z = pv[t].z - q.z; y.w = pv[t].w - q.w;
        }
        if (MODE == 0) {
          y.x -= piv.x; y.y -= piv.y; y.z -= piv.z; y.w -= piv.w;
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
            a1.x += g.x; a1.y += g.y; a1.z += g.z; a1.w += g.w;
            a2.x = fmaf(g.x, xh.x, a2.x); a2.y = fmaf(g.y, xh.y, a2.y);
            a2.z = fmaf(g.z, xh.z, a2.z); a2.w = fmaf(g.w, xh.w, a2.w);
            sgs.x += g.x; sgs.y += g.y; sgs.z += g.z; sgs.w += g.w;
            sys.x += y.x; sys.y += y.y; sys.z += y.z; sys.w += y.w;
          }
        }
      }
    }
    if (MODE == 2) {
      st4(segsum + seg * C + c, sgs);
      st4(segsum + segsum_stride + seg * C + c, sys);
    }
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

// Backward apply by destination point.  Workgroup = (cloud, chunk of its points and segments); C/4 lanes per
// point.  dP[m] = sum over the reverse list of m (rows e = s*ns + j of this cloud that gathered m) of
//   dy[e] = scale*g[e] - (scale/M) * (dbeta + xhat[e] * dgamma),  y[e] = P[m] - Q[s],  g[e] = gz[e] * [scale*y + shift > 0]
// (gz rows are read whole, once each, in reverse-list order), then dQ[s] = -sum_j dy[s,j] from the segment sums
// sg = sum_j g, sy = sum_j y:  dQ = -(scale*sg - (scale/M) * (ns*dbeta + dgamma*rstd*(sy - ns*mean))).
template <bool GEO>
__global__ __launch_bounds__(256) void sa_first_bwd_point_kernel(
    const float* __restrict__ P, int64_t ldp, const float* __restrict__ Q, int N, int S, int ns, int C, int Bg, int bpc,
    const float* __restrict__ coef_all, const double* __restrict__ red_all, int64_t red_stride, float invM,
    const float* __restrict__ gz, const int32_t* __restrict__ rev_off, const int32_t* __restrict__ rev_ent,
    const float* __restrict__ segsum, int64_t segsum_stride, float* __restrict__ dP, float* __restrict__ dQ, Geo geo) {
  const int LPS = C >> 2;
  const int slots = 256 / LPS;
  const int lp = threadIdx.x % LPS, slot = threadIdx.x / LPS;
  const int c = lp * 4;
  const int b = blockIdx.x / bpc, chunk = blockIdx.x % bpc;
  constexpr bool use_geo = GEO;
  GeoLane GL;
  if constexpr (GEO) GL = geo_lane(geo, c);
  const float* coef = coef_all + (int64_t)(b / Bg) * 5 * C;
  const double* red = red_all + (int64_t)(b / Bg) * red_stride;
  const float4 scale = ld4(coef + c), shift = ld4(coef + C + c), mean = ld4(coef + 2 * C + c), rstd = ld4(coef + 3 * C + c);
  const float4 f = make_float4(scale.x * invM, scale.y * invM, scale.z * invM, scale.w * invM);
  const float4 db = make_float4((float)red[c], (float)red[c + 1], (float)red[c + 2], (float)red[c + 3]);
  const float4 dg = make_float4((float)red[C + c], (float)red[C + c + 1], (float)red[C + c + 2], (float)red[C + c + 3]);
  // k0 = f*dbeta - f*dgamma*rstd*mean (the part of dy that does not depend on the row), k1 = f*dgamma*rstd
  const float4 k1 = make_float4(f.x * dg.x * rstd.x, f.y * dg.y * rstd.y, f.z * dg.z * rstd.z, f.w * dg.w * rstd.w);
  const float4 k0 = make_float4(f.x * db.x - k1.x * mean.x, f.y * db.y - k1.y * mean.y, f.z * db.z - k1.z * mean.z,
                                f.w * db.w - k1.w * mean.w);
  const int32_t* offb = rev_off + (int64_t)b * (N + 1);
  const int32_t* entb = rev_ent + (int64_t)b * S * ns;
  const float* Qb = Q + (int64_t)b * S * C + c;
  const float* gzb = gz + (int64_t)b * S * ns * C + c;
  const bool pow2 = (ns & (ns - 1)) == 0;
  const int sh = 31 - __clz(ns);
  const float* centb = use_geo ? geo.cent + (int64_t)b * S * 3 : nullptr;
  for (int m = chunk * slots + slot; m < N; m += bpc * slots) {
    const int off = offb[m], cnt = offb[m + 1] - off;
    const float4 p = (!use_geo || geo.has_p) ? ld4(P + ((int64_t)b * N + m) * ldp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float xm0 = 0.f, xm1 = 0.f, xm2 = 0.f;
    if (use_geo) {
      const float* xm = geo.xyz + ((int64_t)b * N + m) * 3;
      xm0 = xm[0]; xm1 = xm[1]; xm2 = xm[2];
    }
    float4 acc = make_float4(0, 0, 0, 0);
    for (int t0 = 0; t0 < cnt; t0 += JB) {
      int en[JB];
      float4 qv[JB], gv[JB];
#pragma unroll
      for (int t = 0; t < JB; ++t) en[t] = entb[off + (t0 + t < cnt ? t0 + t : cnt - 1)];
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        const int sgm = pow2 ? (en[t] >> sh) : en[t] / ns;
        if (use_geo) {
          const float* cs = centb + (int64_t)sgm * 3;
          qv[t] = make_float4(cs[0], cs[1], cs[2], 0.f);
        } else {
          qv[t] = ld4(Qb + (int64_t)sgm * C);
        }
        gv[t] = ld4(gzb + (int64_t)en[t] * C);
      }
#pragma unroll
      for (int t = 0; t < JB; ++t) {
        if (t0 + t >= cnt) continue;
        float yx, yy, yz, yw;
        if (use_geo) {
          const float4 y = geo_y(GL, p, xm0 - qv[t].x, xm1 - qv[t].y, xm2 - qv[t].z);
          yx = y.x; yy = y.y; yz = y.z; yw = y.w;
        } else {
          yx = p.x - qv[t].x; yy = p.y - qv[t].y; yz = p.z - qv[t].z; yw = p.w - qv[t].w;
        }
        const float gx = fmaf(scale.x, yx, shift.x) > 0.f ? gv[t].x : 0.f, gy = fmaf(scale.y, yy, shift.y) > 0.f ? gv[t].y : 0.f;
        const float gz_ = fmaf(scale.z, yz, shift.z) > 0.f ? gv[t].z : 0.f, gw = fmaf(scale.w, yw, shift.w) > 0.f ? gv[t].w : 0.f;
        acc.x += scale.x * gx - (k0.x + k1.x * yx);
        acc.y += scale.y * gy - (k0.y + k1.y * yy);
        acc.z += scale.z * gz_ - (k0.z + k1.z * yz);
        acc.w += scale.w * gw - (k0.w + k1.w * yw);
      }
    }
    st4(dP + ((int64_t)b * N + m) * C + c, acc);
  }
  const float nsf = (float)ns;
  for (int sg = chunk * slots + slot; sg < S; sg += bpc * slots) {
    const int64_t seg = (int64_t)b * S + sg;
    const float4 g = ld4(segsum + seg * C + c), y = ld4(segsum + segsum_stride + seg * C + c);
    float4 dq;
    dq.x = -(scale.x * g.x - (nsf * k0.x + k1.x * y.x));
    dq.y = -(scale.y * g.y - (nsf * k0.y + k1.y * y.y));
    dq.z = -(scale.z * g.z - (nsf * k0.z + k1.z * y.z));
    dq.w = -(scale.w * g.w - (nsf * k0.w + k1.w * y.w));
    st4(dQ + seg * C + c, dq);
  }
}

struct Plan {
  int segs_per_block, nblk;
};
inline Plan plan(int64_t segs, int C) {
  const int slots = 256 / (C >> 2);
  int64_t nblk = SUG_STATS_ROWS;                     // partial rows per group (the pivot row lives behind them)
  int64_t spb = (segs + nblk - 1) / nblk;
  spb = (spb + slots - 1) / slots * slots;           // whole passes
  if (spb < slots) spb = slots;
  return Plan{(int)spb, (int)((segs + spb - 1) / spb)};
}

}  // namespace

static int sa_first_fwd(const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B, int N, int S,
                        int ns, int C, int groups, const float* gamma, const float* beta, int training,
                        float eps, float momentum, float* running_mean, float* running_var, float* coef,
                        float* Z, float* ws, void* stream, Geo geo) {
  SUG_REQUIRE((geo.xyz || (P && Q)) && idx && gamma && beta && coef && Z && ws, "sug_sa_first_fwd: null pointer");
  SUG_REQUIRE(!geo.xyz || (geo.cent && geo.Wx && (!geo.has_p || P)), "sug_sa_first_geo_fwd: null pointer");
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
      if (geo.xyz)
        hipLaunchKernelGGL((sa_first_kernel<0, true>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                           pl.segs_per_block, nullptr, nullptr, nullptr, nullptr, 0, ws, geo);
      else
        hipLaunchKernelGGL((sa_first_kernel<0, false>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                           pl.segs_per_block, nullptr, nullptr, nullptr, nullptr, 0, ws, geo);
      SUG_LAUNCH_CHECK("sug_sa_first_fwd(stats)");
      if (int rc = sug_stats_finalize(ws, pl.nblk, C, gamma, beta, (double)segs_g * ns, eps, momentum, running_mean,
                                      running_var, cg, st, ws + SUG_PIVOT_OFFSET(C)))
        return rc;
    }
    if (geo.xyz)
      hipLaunchKernelGGL((sa_first_kernel<1, true>), dim3(pl.nblk), dim3(256), 0, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                         pl.segs_per_block, cg, nullptr, Z, nullptr, 0, nullptr, geo);
    else
      hipLaunchKernelGGL((sa_first_kernel<1, false>), dim3(pl.nblk), dim3(256), 0, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                         pl.segs_per_block, cg, nullptr, Z, nullptr, 0, nullptr, geo);
    SUG_LAUNCH_CHECK("sug_sa_first_fwd(apply)");
  }
  return SUG_OK;
}

extern "C" int sug_sa_first_fwd(const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B, int N, int S,
                                int ns, int C, int groups, const float* gamma, const float* beta, int training,
                                float eps, float momentum, float* running_mean, float* running_var, float* coef,
                                float* Z, float* ws, void* stream) {
  return sa_first_fwd(P, ldp, Q, idx, B, N, S, ns, C, groups, gamma, beta, training, eps, momentum, running_mean, running_var,
                      coef, Z, ws, stream, Geo{nullptr, nullptr, nullptr, nullptr, 0, 0});
}

extern "C" int sug_sa_first_geo_fwd(const float* Pf, int64_t ldp, const float* xyz, const float* cent, const float* Wx,
                                    int ldw, const float* bias, const int32_t* idx, int B, int N, int S, int ns, int C,
                                    int groups, const float* gamma, const float* beta, int training, float eps,
                                    float momentum, float* running_mean, float* running_var, float* coef, float* Z,
                                    float* ws, void* stream) {
  SUG_REQUIRE(xyz && cent && Wx && ldw >= 3, "sug_sa_first_geo_fwd: null pointer");
  SUG_REQUIRE(!Pf || (ldp >= C && ldp % 4 == 0 && ((uintptr_t)Pf % 16) == 0), "sug_sa_first_geo_fwd: Pf rows must be 16-byte aligned");
  SUG_REQUIRE(!bias || ((uintptr_t)bias % 16) == 0, "sug_sa_first_geo_fwd: bias must be 16-byte aligned");
  return sa_first_fwd(Pf, Pf ? ldp : C, Z /* unused, passes the alignment check */, idx, B, N, S, ns, C, groups, gamma, beta,
                      training, eps, momentum, running_mean, running_var, coef, Z, ws, stream,
                      Geo{xyz, cent, Wx, bias, ldw, Pf ? 1 : 0});
}

static int sa_first_bwd(const float* gz, const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B,
                        int N, int S, int ns, int C, int groups, int training, const float* coef, double* red,
                        int32_t* rev_off, int32_t* rev_ent, float* segsum, float* dP, float* dQ, float* ws,
                        float* dgb, void* stream, Geo geo) {
  SUG_REQUIRE(gz && (geo.xyz || (P && Q)) && idx && coef && red && rev_off && rev_ent && segsum && dP && dQ && ws,
              "sug_sa_first_bwd: null pointer");
  if (geo.xyz) {             // (the P / Q alignment checks below see valid stand-ins)
    if (!P) P = dP;
    if (!Q) Q = dQ;
  }
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && ns > 0, "sug_sa_first_bwd: bad shape");
  SUG_REQUIRE(C == 64 || C == 128, "sug_sa_first_bwd: C=%d (64 or 128)", C);
  SUG_REQUIRE(groups >= 1 && B % groups == 0, "sug_sa_first_bwd: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(((uintptr_t)gz % 16) == 0 && ((uintptr_t)dQ % 16) == 0 && ((uintptr_t)dP % 16) == 0 &&
                  ((uintptr_t)segsum % 16) == 0 && ((uintptr_t)P % 16) == 0 && ((uintptr_t)Q % 16) == 0 && ldp % 4 == 0,
              "sug_sa_first_bwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t segs_g = (int64_t)(B / groups) * S;
  const Plan pl = plan(segs_g, C);
  const size_t sh = (size_t)(256 / (C >> 2)) * 2 * C * sizeof(float);
  const float invM = (float)(1.0 / ((double)segs_g * ns));
  const int64_t segsum_stride = (int64_t)B * S * C;
  // which rows gathered each point (order inside a list: as the atomics of the build left it)
  // sorted lists: a point's sum runs over its rows in ascending (centroid, neighbour) order -- reproducible bit for bit
  // (SUG_SA_UNSORTED=1: the lists as the build's atomics left them, the reference's own index_points-backward freedom)
  static const int unsorted = getenv("SUG_SA_UNSORTED") ? atoi(getenv("SUG_SA_UNSORTED")) : 0;
  if (int rc = sug_reverse_lists(idx, B, S * ns, N, unsorted ? 0 : 1, rev_off, rev_ent, st)) return rc;
  for (int g = 0; g < groups; ++g) {
    const float* cg = coef + (int64_t)g * 5 * C;
    double* rg = red + (int64_t)g * 2 * C;
    const int64_t s0 = g * segs_g, s1 = s0 + segs_g;
    if (geo.xyz)
      hipLaunchKernelGGL((sa_first_kernel<2, true>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                         pl.segs_per_block, cg, gz, nullptr, segsum, segsum_stride, ws, geo);
    else
      hipLaunchKernelGGL((sa_first_kernel<2, false>), dim3(pl.nblk), dim3(256), sh, st, P, ldp, Q, idx, N, S, ns, C, s0, s1,
                         pl.segs_per_block, cg, gz, nullptr, segsum, segsum_stride, ws, geo);
    SUG_LAUNCH_CHECK("sug_sa_first_bwd(reduce)");
    if (int rc = sug_reduce_partials(ws, pl.nblk, 2 * C, rg, st)) return rc;
  }
  // eval mode: the statistics are constants (red + groups*2C: a caller-zeroed spare row, shared by the groups)
  const double* ru = training ? red : red + (int64_t)groups * 2 * C;
  const int slots = 256 / (C >> 2);
  int bpc = sug_divup(N > S ? N : S, slots);
  while (bpc > 1 && (int64_t)B * bpc > 8192) bpc = (bpc + 1) / 2;
  if (geo.xyz)
    hipLaunchKernelGGL(sa_first_bwd_point_kernel<true>, dim3(B * bpc), dim3(256), 0, st, P, ldp, Q, N, S, ns, C, B / groups, bpc,
                       coef, ru, (int64_t)(training ? 2 * C : 0), invM, gz, rev_off, rev_ent, segsum, segsum_stride, dP, dQ, geo);
  else
    hipLaunchKernelGGL(sa_first_bwd_point_kernel<false>, dim3(B * bpc), dim3(256), 0, st, P, ldp, Q, N, S, ns, C, B / groups, bpc,
                       coef, ru, (int64_t)(training ? 2 * C : 0), invM, gz, rev_off, rev_ent, segsum, segsum_stride, dP, dQ, geo);
  SUG_LAUNCH_CHECK("sug_sa_first_bwd(apply)");
  if (dgb) return sug_fold_groups(red, groups, 2 * C, dgb, stream);
  return SUG_OK;
}

extern "C" int sug_sa_first_bwd(const float* gz, const float* P, int64_t ldp, const float* Q, const int32_t* idx, int B,
                                int N, int S, int ns, int C, int groups, int training, const float* coef, double* red,
                                int32_t* rev_off, int32_t* rev_ent, float* segsum, float* dP, float* dQ, float* ws,
                                float* dgb, void* stream) {
  return sa_first_bwd(gz, P, ldp, Q, idx, B, N, S, ns, C, groups, training, coef, red, rev_off, rev_ent, segsum, dP, dQ, ws,
                      dgb, stream, Geo{nullptr, nullptr, nullptr, nullptr, 0, 0});
}

extern "C" int sug_sa_first_geo_bwd(const float* gz, const float* Pf, int64_t ldp, const float* xyz, const float* cent,
                                    const float* Wx, int ldw, const float* bias, const int32_t* idx, int B, int N, int S,
                                    int ns, int C, int groups, int training, const float* coef, double* red,
                                    int32_t* rev_off, int32_t* rev_ent, float* segsum, float* dP, float* dQ, float* ws,
                                    float* dgb, void* stream) {
  SUG_REQUIRE(xyz && cent && Wx && ldw >= 3, "sug_sa_first_geo_bwd: null pointer");
  SUG_REQUIRE(!Pf || (ldp >= C && ldp % 4 == 0 && ((uintptr_t)Pf % 16) == 0), "sug_sa_first_geo_bwd: Pf rows must be 16-byte aligned");
  SUG_REQUIRE(!bias || ((uintptr_t)bias % 16) == 0, "sug_sa_first_geo_bwd: bias must be 16-byte aligned");
  return sa_first_bwd(gz, Pf, Pf ? ldp : C, nullptr, idx, B, N, S, ns, C, groups, training, coef, red, rev_off, rev_ent, segsum,
                      dP, dQ, ws, dgb, stream, Geo{xyz, cent, Wx, bias, ldw, Pf ? 1 : 0});
}
