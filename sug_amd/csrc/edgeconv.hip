// EdgeConv gather-reduce (forward + backward) and train-mode BatchNorm helpers for
// channel-last tensors.  Reference: get_graph_feature + conv_2d + max over k,
// model/model_utils.py:188-210, :8-32, model/Model.py:88-109.
//
// y[b,n,j,c] = P[b,idx[b,n,j],c] + Q[b,n,c] with [P|Q] = x.[W1 ; W2-W1]^T is never
// stored: the forward keeps only the per-point extreme z (max or min by sign(gamma),
// BN+LeakyReLU being monotone per channel), its arg, sum_j y and the fp64 BN sums.
//
// Data movement per cloud and layer (fp32): read PQ once for Q (4*N*Co) + k gathered
// P rows per point (L2 hits; 4*N*k*Co), write z (4), arg (1), s1 (4) per element.
// Lanes run along channels (float4 per lane) so every gathered row is one
// contiguous 16*LPP-byte read.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int NCH>
__global__ __launch_bounds__(256) void edgeconv_fwd_kernel(
    const float* __restrict__ pq, int64_t ldpq, const int32_t* __restrict__ idx,
    const float* __restrict__ gamma, int64_t BN, int N, int k, int Co, int LPP, int bpc,
    float* __restrict__ z, uint8_t* __restrict__ arg, float* __restrict__ s1,
    float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float s_red[];  // [256/LPP][2*Co] per-slot partial sums
  const int nchunk = Co >> 2;
  const int ppb = 256 / LPP;
  const int slot = threadIdx.x / LPP, ch0 = threadIdx.x % LPP;
  float4 a1[NCH], a2[NCH], gs[NCH];
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    a1[u] = make_float4(0, 0, 0, 0);
    a2[u] = make_float4(0, 0, 0, 0);
    const int ch = ch0 + u * LPP;
    gs[u] = ch < nchunk ? ld4(gamma + ch * 4) : make_float4(1, 1, 1, 1);
  }
  // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs, so the ones with
  // equal blockIdx % 8 share an L2.  All `bpc` workgroups of a cloud get the same residue:
  // a cloud's P matrix (N*Co*4 B <= 1 MB) then stays in one 4 MB L2 instead of being pulled
  // through every XCD (placement only affects speed).
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int64_t b = (int64_t)(jb / bpc) * 8 + xcd;
  const int lb = jb % bpc;
  const int64_t B_ = BN / N;
  for (int n = lb * ppb + slot; b < B_ && n < N; n += bpc * ppb) {
    const int64_t p = b * N + n;
    const int32_t* ir = idx + p * k;
    const float* base = pq + b * N * ldpq;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = ch0 + u * LPP;
      if (ch >= nchunk) continue;
      const float4 q = ld4(pq + p * ldpq + Co + ch * 4);
      float bx = 0, by = 0, bz = 0, bw = 0;
      int jx = 0, jy = 0, jz = 0, jw = 0;
      float sx = 0, sy = 0, sz = 0, sw = 0, qx = 0, qy = 0, qz = 0, qw = 0;
      const bool px = gs[u].x >= 0.f, py = gs[u].y >= 0.f, pz = gs[u].z >= 0.f, pw = gs[u].w >= 0.f;
      // The gathers are latency-bound (L2 hits, waves mostly waiting): fetch a batch of GB
      // neighbour indices first, then issue all GB row gathers back to back, then reduce --
      // two memory round trips per batch instead of two per neighbour group.
      constexpr int GB = 10;
      for (int j0 = 0; j0 < k; j0 += GB) {
        int nb[GB];
        float4 pv[GB];
#pragma unroll
        for (int t = 0; t < GB; ++t) {
          const int j = j0 + t;
          int v = (j < k) ? ir[j] : 0;
          nb[t] = v < 0 ? 0 : (v >= N ? N - 1 : v);
        }
#pragma unroll
        for (int t = 0; t < GB; ++t) pv[t] = ld4(base + (int64_t)nb[t] * ldpq + ch * 4);
#pragma unroll
        for (int t = 0; t < GB; ++t) {
          const int j = j0 + t;
          if (j < k) {
            const float yx = __fadd_rn(pv[t].x, q.x), yy = __fadd_rn(pv[t].y, q.y);
            const float yz = __fadd_rn(pv[t].z, q.z), yw = __fadd_rn(pv[t].w, q.w);
            sx += yx; sy += yy; sz += yz; sw += yw;
            qx = fmaf(yx, yx, qx); qy = fmaf(yy, yy, qy); qz = fmaf(yz, yz, qz); qw = fmaf(yw, yw, qw);
            if (j == 0 || (px ? yx > bx : yx < bx)) { bx = yx; jx = j; }
            if (j == 0 || (py ? yy > by : yy < by)) { by = yy; jy = j; }
            if (j == 0 || (pz ? yz > bz : yz < bz)) { bz = yz; jz = j; }
            if (j == 0 || (pw ? yw > bw : yw < bw)) { bw = yw; jw = j; }
          }
        }
      }
      const int64_t o = p * Co + ch * 4;
      st4(z + o, make_float4(bx, by, bz, bw));
      *reinterpret_cast<uint32_t*>(arg + o) =
          (uint32_t)jx | ((uint32_t)jy << 8) | ((uint32_t)jz << 16) | ((uint32_t)jw << 24);
      if (s1) st4(s1 + o, make_float4(sx, sy, sz, sw));
      a1[u].x += sx; a1[u].y += sy; a1[u].z += sz; a1[u].w += sw;
      a2[u].x += qx; a2[u].y += qy; a2[u].z += qz; a2[u].w += qw;
    }
  }
  // deterministic block reduction: every (slot, chunk) cell is written by exactly one thread,
  // then summed over slots in a fixed order; blocks are summed in order by reduce_partials.
  float* mine = s_red + (size_t)slot * 2 * Co;
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int ch = ch0 + u * LPP;
    if (ch >= nchunk) continue;
    st4(mine + ch * 4, a1[u]);
    st4(mine + Co + ch * 4, a2[u]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * Co; i += 256) {
    float acc = 0.f;
    for (int sl = 0; sl < ppb; ++sl) acc += s_red[(size_t)sl * 2 * Co + i];
    ws[(size_t)blockIdx.x * 2 * Co + i] = acc;
  }
}


// LDS-resident variant.  The gathers of edgeconv_fwd_kernel are bound by L2 bandwidth: every point
// pulls k full rows of P through L2 (k*4*Co bytes per point, 20x the matrix).  All neighbours of a
// point lie in its own cloud, so a workgroup stages a 16-channel slice of the cloud's whole P matrix
// in LDS (N x 64 B = 64 KB at N = 1024) and gathers from there: P leaves L2 once per slice, the
// 20-fold reuse is served by LDS.  Workgroup = (cloud, 16-channel slice, part of the points);
// 4 lanes per point, 64 points per pass; same per-point arithmetic (and tie rules) as above.
// ws: one partial row per (cloud, part): row (b*psplit + part), only this slice's columns.
template <int SW, int KK>
__global__ __launch_bounds__(256) void edgeconv_fwd_lds_kernel(
    const float* __restrict__ pq, int64_t ldpq, const int32_t* __restrict__ idx,
    const float* __restrict__ gamma, int B, int N, int Co, int psplit,
    float* __restrict__ z, uint8_t* __restrict__ arg, float* __restrict__ s1, float* __restrict__ ws, int Bg) {
  static_assert(KK % 4 == 0, "neighbour rows are fetched as int4");
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  float* s_p = s_lds;                          // [N][SW]
  float* s_red = s_lds + (size_t)N * SW;       // [PPP][2*SW]
  constexpr int LP = SW / 4;                   // lanes per point
  constexpr int PPP = 256 / LP;                // points per pass
  const int nslice = Co / SW;
  const int per_cloud = nslice * psplit;
  int b, r;
  if ((B & 7) == 0) {                          // a cloud's workgroups share an XCD (its L2 holds P once)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = (j / per_cloud) * 8 + xcd;
    r = j % per_cloud;
  } else {
    b = blockIdx.x / per_cloud;
    r = blockIdx.x % per_cloud;
  }
  const int sl = r % nslice, part = r / nslice;
  const int c0 = sl * SW;
  const float* base = pq + (int64_t)b * N * ldpq;
  // LDS holds sign(gamma) * P: "max for gamma >= 0, min for gamma < 0" becomes a plain max of
  // y' = sign*y = sign*P + sign*Q (sign flips are exact), and sum(y) = sign*sum(y'), sum(y^2) = sum(y'^2)
  for (int e = threadIdx.x; e < N * LP; e += 256) {
    const int n = e / LP, c4 = e % LP;
    const float4 gf = ld4(gamma + c0 + c4 * 4);
    float4 pv = ld4(base + (int64_t)n * ldpq + c0 + c4 * 4);
    pv.x = gf.x >= 0.f ? pv.x : -pv.x; pv.y = gf.y >= 0.f ? pv.y : -pv.y;
    pv.z = gf.z >= 0.f ? pv.z : -pv.z; pv.w = gf.w >= 0.f ? pv.w : -pv.w;
    st4(s_p + n * SW + c4 * 4, pv);
  }
  const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
  const float4 g4 = ld4(gamma + c0 + lp * 4);
  const float4 sg = make_float4(g4.x >= 0.f ? 1.f : -1.f, g4.y >= 0.f ? 1.f : -1.f, g4.z >= 0.f ? 1.f : -1.f,
                                g4.w >= 0.f ? 1.f : -1.f);
  float4 a1 = make_float4(0, 0, 0, 0), a2 = make_float4(0, 0, 0, 0);
  // BatchNorm sums about a pivot (common.h; Bg > 0): y of point 0 / its first neighbour slot taken as point 0 of the
  // first cloud of this cloud's domain group, p = P[b0,0] + Q[b0,0] (sign-folded like y'); the group's partial rows
  // share it, the first cloud's workgroups publish it for the fold.  The stored maxima z and s1 = sum_j y stay unshifted.
  float4 pv4 = make_float4(0, 0, 0, 0);
  if (Bg > 0) {
    const float* p0 = pq + (int64_t)(b / Bg) * Bg * N * ldpq;
    const float4 a = ld4(p0 + c0 + lp * 4), c = ld4(p0 + Co + c0 + lp * 4);
    pv4 = make_float4(__fadd_rn(a.x, c.x), __fadd_rn(a.y, c.y), __fadd_rn(a.z, c.z), __fadd_rn(a.w, c.w));
    if (b % Bg == 0 && part == 0 && slot == 0) st4(ws + SUG_PIVOT_OFFSET(Co) + (size_t)(b / Bg) * Co + c0 + lp * 4, pv4);
    pv4.x *= sg.x; pv4.y *= sg.y; pv4.z *= sg.z; pv4.w *= sg.w;
  }
  const int n_per = (N + psplit - 1) / psplit;
  const int n_begin = part * n_per, n_end = (n_begin + n_per < N) ? n_begin + n_per : N;
  // the neighbour list and the Q row of the NEXT point are fetched (global) while the current point
  // gathers from LDS: a thread walks its points serially and few waves share a CU
  int4 nv[KK / 4], nvn[KK / 4];
  float4 q = make_float4(0, 0, 0, 0), qn = q;
  auto fetch = [&](int n, int4 (&iv)[KK / 4], float4& qq) {
    if (n < n_end) {
      const int64_t p = (int64_t)b * N + n;
      const int4* ir = reinterpret_cast<const int4*>(idx + p * KK);
#pragma unroll
      for (int t = 0; t < KK / 4; ++t) iv[t] = ir[t];
      qq = ld4(pq + p * ldpq + Co + c0 + lp * 4);
      qq.x *= sg.x; qq.y *= sg.y; qq.z *= sg.z; qq.w *= sg.w;
    }
  };
  fetch(n_begin + slot, nv, q);
  __syncthreads();
  for (int n = n_begin + slot; n < n_end; n += PPP) {
    fetch(n + PPP, nvn, qn);
    float bx = 0, by = 0, bz = 0, bw = 0;
    int jx = 0, jy = 0, jz = 0, jw = 0;
    float sx = 0, sy = 0, sz = 0, sw = 0, qx = 0, qy = 0, qz = 0, qw = 0;
#pragma unroll
    for (int t = 0; t < KK / 4; ++t) {
      const int m4[4] = {nv[t].x, nv[t].y, nv[t].z, nv[t].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = t * 4 + u;
        const int m = min(max(m4[u], 0), N - 1);                      // clamp to the cloud (v_med3_i32)
        const float4 pv = ld4(s_p + m * SW + lp * 4);
        const float yx = __fadd_rn(pv.x, q.x), yy = __fadd_rn(pv.y, q.y);
        const float yz = __fadd_rn(pv.z, q.z), yw = __fadd_rn(pv.w, q.w);
        sx += yx; sy += yy; sz += yz; sw += yw;
        const float dx = yx - pv4.x, dy = yy - pv4.y, dz = yz - pv4.z, dw = yw - pv4.w;
        qx = fmaf(dx, dx, qx); qy = fmaf(dy, dy, qy); qz = fmaf(dz, dz, qz); qw = fmaf(dw, dw, qw);
        if (j == 0 || yx > bx) { bx = yx; jx = j; }                  // first maximum wins, as torch.max
        if (j == 0 || yy > by) { by = yy; jy = j; }
        if (j == 0 || yz > bz) { bz = yz; jz = j; }
        if (j == 0 || yw > bw) { bw = yw; jw = j; }
      }
    }
    // back to the true sign
    bx *= sg.x; by *= sg.y; bz *= sg.z; bw *= sg.w;
    sx *= sg.x; sy *= sg.y; sz *= sg.z; sw *= sg.w;
    const int64_t o = ((int64_t)b * N + n) * Co + c0 + lp * 4;
    st4(z + o, make_float4(bx, by, bz, bw));
    *reinterpret_cast<uint32_t*>(arg + o) =
        (uint32_t)jx | ((uint32_t)jy << 8) | ((uint32_t)jz << 16) | ((uint32_t)jw << 24);
    if (s1) st4(s1 + o, make_float4(sx, sy, sz, sw));
    constexpr float kf = (float)KK;                                   // sum_j (y - p) = sum_j y - k p  (true sign)
    a1.x += fmaf(-kf, pv4.x * sg.x, sx); a1.y += fmaf(-kf, pv4.y * sg.y, sy);
    a1.z += fmaf(-kf, pv4.z * sg.z, sz); a1.w += fmaf(-kf, pv4.w * sg.w, sw);
    a2.x += qx; a2.y += qy; a2.z += qz; a2.w += qw;
#pragma unroll
    for (int t = 0; t < KK / 4; ++t) nv[t] = nvn[t];
    q = qn;
  }
  // fixed-order reduction over the point slots (fp64), one partial row per (cloud, part)
  st4(s_red + slot * 2 * SW + lp * 4, a1);
  st4(s_red + slot * 2 * SW + SW + lp * 4, a2);
  __syncthreads();
  if (threadIdx.x < 2 * SW) {
    double acc = 0.0;
    for (int t = 0; t < PPP; ++t) acc += (double)s_red[t * 2 * SW + threadIdx.x];
    const int col = threadIdx.x < SW ? c0 + threadIdx.x : Co + c0 + (threadIdx.x - SW);
    ws[(size_t)(b * psplit + part) * 2 * Co + col] = (float)acc;
  }
}

// out[i] = sum over workgroup rows of ws[row][i], in fp64, in a fixed order: 16 strided
// partial sums per column (rows p, p+16, ...) combined in ascending p.  grid = ceil(W/16).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nblk,
                                                              int W, double* __restrict__ out) {
  __shared__ double s_p[16][17];
  ws += (size_t)blockIdx.y * nblk * W;           // blockIdx.y = domain group: its nblk partial rows -> out row g
  out += (size_t)blockIdx.y * W;
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < W) {
#pragma unroll 8
    for (int b = p; b < nblk; b += 16) acc += (double)ws[(size_t)b * W + c];
  }
  s_p[p][cl] = acc;
  __syncthreads();
  if (p == 0 && c < W) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_p[i][cl];
    out[c] = t;
  }
}

// reduce_partials_kernel for every domain group of a layer in one workgroup per 16 columns, plus the group-folded fp32
// copy the parameter gradients want (sug_fold_groups' arithmetic: the groups' fp64 sums added in group order, then
// rounded) -- the separate fold launch of every BatchNorm backward is gone.
__global__ __launch_bounds__(256) void reduce_partials_fold_kernel(const float* __restrict__ ws, int nblk, int W, int groups,
                                                                   double* __restrict__ out, float* __restrict__ folded) {
  __shared__ double s_p[16][17];
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double tot = 0.0;
  for (int g = 0; g < groups; ++g) {
    const float* wg = ws + (size_t)g * nblk * W;
    double acc = 0.0;
    if (c < W) {
#pragma unroll 8
      for (int b = p; b < nblk; b += 16) acc += (double)wg[(size_t)b * W + c];
    }
    __syncthreads();
    s_p[p][cl] = acc;
    __syncthreads();
    if (p == 0 && c < W) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += s_p[i][cl];
      out[(size_t)g * W + c] = t;
      tot += t;
    }
  }
  if (p == 0 && c < W) folded[c] = (float)tot;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C,
                                                          double count, float eps, float momentum,
                                                          float* __restrict__ rmean,
                                                          float* __restrict__ rvar,
                                                          float* __restrict__ coef) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    const double mean = stats[c] / count;
    double var = stats[C + c] / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * rstd;
    coef[c] = scale;
    coef[C + c] = beta[c] - (float)mean * scale;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = rstd;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    coef[4 * C + c] = (float)unb;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// reduce_partials + bn_finalize in one launch: a workgroup folds the partial rows of 8 channels
// (their sum and sum-of-squares columns c and C+c, same 16-way strided fixed order as
// reduce_partials_kernel) and writes the BN coefficients / running statistics of those channels.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ ws, int nblk, int C,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, double count,
                                                             float eps, float momentum, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, float* __restrict__ coef,
                                                             const float* __restrict__ pivot) {
  __shared__ double s_p[16][17];
  __shared__ double s_tot[16];
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int ch = blockIdx.x * 8 + (cl & 7);                 // cl < 8: sum column, cl >= 8: sum of squares
  const int col = (cl < 8) ? ch : C + ch;
  const int W = 2 * C;
  double acc = 0.0;
  if (ch < C) {
#pragma unroll 8
    for (int b = p; b < nblk; b += 16) acc += (double)ws[(size_t)b * W + col];
  }
  s_p[p][cl] = acc;
  __syncthreads();
  if (p == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s_p[i][cl];
    s_tot[cl] = t;
  }
  __syncthreads();
  if (threadIdx.x < 8 && ch < C) {
    const int c = ch;
    const double m1 = s_tot[threadIdx.x] / count;           // mean of (x - pivot)
    double var = s_tot[8 + threadIdx.x] / count - m1 * m1;
    if (var < 0) var = 0;
    const double mean = (pivot ? (double)pivot[c] : 0.0) + m1;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * rstd;
    coef[c] = scale;
    coef[C + c] = beta[c] - (float)mean * scale;
    coef[2 * C + c] = (float)mean;
    coef[3 * C + c] = rstd;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    coef[4 * C + c] = (float)unb;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// stats_finalize_kernel for `groups` consecutive blocks of nblk partial rows (one BatchNorm call per domain group,
// in group order: the running statistics see the groups' updates one after the other, as separate calls would).
__global__ __launch_bounds__(1024) void stats_finalize_groups_kernel(const float* __restrict__ ws, int nblk, int C, int groups,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, double count,
                                                                    float eps, float momentum, float* __restrict__ rmean,
                                                                    float* __restrict__ rvar, float* __restrict__ coef,
                                                                    const float* __restrict__ pivot) {
  constexpr int GMAX = 4;                        // groups folded in one pass over the partial rows (loads of all
  constexpr int RL = 64;                         // groups in flight together); more groups: further passes.  64 row
  __shared__ double s_p[GMAX][RL][17];           // lanes (1024 threads): hundreds of partial rows, walked serially by
  __shared__ double s_tot[GMAX][16];             // 16 row lanes, made this launch 22 us
  const int cl = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int ch = blockIdx.x * 8 + (cl & 7);                 // cl < 8: sum column, cl >= 8: sum of squares
  const int col = (cl < 8) ? ch : C + ch;
  const int W = 2 * C;
  const bool fin = threadIdx.x < 8 && ch < C;
  float rm = (fin && rmean) ? rmean[ch] : 0.f, rv = (fin && rvar) ? rvar[ch] : 0.f;
  for (int g0 = 0; g0 < groups; g0 += GMAX) {
    const int ng = groups - g0 < GMAX ? groups - g0 : GMAX;
    double acc[GMAX] = {0.0, 0.0, 0.0, 0.0};
    if (ch < C) {
#pragma unroll 4
      for (int b = p; b < nblk; b += RL) {
#pragma unroll
        for (int g = 0; g < GMAX; ++g)
          if (g < ng) acc[g] += (double)ws[((size_t)(g0 + g) * nblk + b) * W + col];
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g) s_p[g][p][cl] = acc[g];
    __syncthreads();
    if (p < ng) {                                 // row-lane p folds group p
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < RL; ++i) t += s_p[p][i][cl];
      s_tot[p][cl] = t;
    }
    __syncthreads();
    if (fin) {
      const int c = ch;
      for (int g = 0; g < ng; ++g) {               // in group order: the running buffers see one update after the other
        float* cg = coef + (size_t)(g0 + g) * 5 * C;
        const double m1 = s_tot[g][threadIdx.x] / count;      // mean of (x - pivot)
        double var = s_tot[g][8 + threadIdx.x] / count - m1 * m1;
        if (var < 0) var = 0;
        const double mean = (pivot ? (double)pivot[(size_t)(g0 + g) * C + c] : 0.0) + m1;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * rstd;
        cg[c] = scale;
        cg[C + c] = beta[c] - (float)mean * scale;
        cg[2 * C + c] = (float)mean;
        cg[3 * C + c] = rstd;
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        cg[4 * C + c] = (float)unb;
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
      }
    }
  }
  if (fin) {
    if (rmean) rmean[ch] = rm;
    if (rvar) rvar[ch] = rv;
  }
}

// The running-statistics update of G more train-mode forwards on batches whose statistics are
// already known (coef rows 2 and 4), applied in group order with bn_finalize's arithmetic.
__global__ __launch_bounds__(256) void bn_replay_kernel(const float* __restrict__ coef, int G, int C,
                                                        float momentum, float* __restrict__ rmean,
                                                        float* __restrict__ rvar) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m = rmean[c], v = rvar[c];
  for (int g = 0; g < G; ++g) {
    const float* cg = coef + (size_t)g * 5 * C;
    m = (1.f - momentum) * m + momentum * cg[2 * C + c];
    v = (1.f - momentum) * v + momentum * cg[4 * C + c];
  }
  rmean[c] = m;
  rvar[c] = v;
}

// bn_replay_kernel for up to SUG_REPLAY_MULTI layers in one launch (the BatchNorm layers of a shared encoder prefix:
// replay_bn_stats): blockIdx.y = layer.
#define SUG_REPLAY_MULTI 16
struct ReplayMulti {
  const float* coef[SUG_REPLAY_MULTI];
  float* rmean[SUG_REPLAY_MULTI];
  float* rvar[SUG_REPLAY_MULTI];
  int G[SUG_REPLAY_MULTI], C[SUG_REPLAY_MULTI];
  float momentum[SUG_REPLAY_MULTI];
};
__global__ __launch_bounds__(256) void bn_replay_multi_kernel(ReplayMulti a) {
  const int l = blockIdx.y, C = a.C[l], G = a.G[l];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float momentum = a.momentum[l];
  const float* coef = a.coef[l];
  float m = a.rmean[l][c], v = a.rvar[l][c];
  for (int g = 0; g < G; ++g) {
    const float* cg = coef + (size_t)g * 5 * C;
    m = (1.f - momentum) * m + momentum * cg[2 * C + c];
    v = (1.f - momentum) * v + momentum * cg[4 * C + c];
  }
  a.rmean[l][c] = m;
  a.rvar[l][c] = v;
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ z, int64_t ldz,
                                                         const float* __restrict__ coef, int64_t rows,
                                                         int C, float slope, float* __restrict__ out,
                                                         int64_t ldo) {
  const int64_t total = rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t r = e / C;
    const float u = fmaf(coef[c], z[r * ldz + c], coef[C + c]);
    out[r * ldo + c] = u > 0.f ? u : u * slope;
  }
}

__global__ __launch_bounds__(256) void affine_act_vec4_kernel(const float* __restrict__ z, int64_t ldz,
                                                              const float* __restrict__ coef,
                                                              int64_t rows, int C, float slope,
                                                              float* __restrict__ out, int64_t ldo) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C4) * 4;
    const int64_t r = e / C4;
    const float4 zv = ld4(z + r * ldz + c), sc = ld4(coef + c), sh = ld4(coef + C + c);
    float4 u;
    u.x = fmaf(sc.x, zv.x, sh.x); u.y = fmaf(sc.y, zv.y, sh.y);
    u.z = fmaf(sc.z, zv.z, sh.z); u.w = fmaf(sc.w, zv.w, sh.w);
    u.x = u.x > 0.f ? u.x : u.x * slope; u.y = u.y > 0.f ? u.y : u.y * slope;
    u.z = u.z > 0.f ? u.z : u.z * slope; u.w = u.w > 0.f ? u.w : u.w * slope;
    st4(out + r * ldo + c, u);
  }
}

// affine_act_vec4_kernel with one coefficient set per block of rows_g rows (domain groups)
__global__ __launch_bounds__(256) void affine_act_vec4_groups_kernel(const float* __restrict__ z, int64_t ldz,
                                                                     const float* __restrict__ coef, int64_t rows,
                                                                     int64_t rows_g, int C, float slope,
                                                                     float* __restrict__ out, int64_t ldo) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  constexpr int U = 4;                            // independent 16-byte loads in flight per thread
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < total; e0 += U * stride) {
    float4 zv[U];
    int c[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t e = e0 + u * stride;
      const int64_t ee = e < total ? e : e0;
      c[u] = (int)(ee % C4) * 4;
      r[u] = ee / C4;
      zv[u] = ld4(z + r[u] * ldz + c[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u * stride >= total) continue;
      const float* cg = coef + (r[u] / rows_g) * 5 * C;
      const float4 sc = ld4(cg + c[u]), sh = ld4(cg + C + c[u]);
      float4 v;
      v.x = fmaf(sc.x, zv[u].x, sh.x); v.y = fmaf(sc.y, zv[u].y, sh.y);
      v.z = fmaf(sc.z, zv[u].z, sh.z); v.w = fmaf(sc.w, zv[u].w, sh.w);
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      st4(out + r[u] * ldo + c[u], v);
    }
  }
}

// Column sums (+ squares) of a [rows, C] channel-last tensor.  MODE 0: plain BN stats.
// MODE 1: EdgeConv backward reduce (also writes a = scale*G).
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ z,
                                                         const float* __restrict__ coef, int64_t rows,
                                                         int C, float slope, float* __restrict__ a,
                                                         float* __restrict__ ws, int CW,
                                                         int rows_per_block, int pivoted) {
  extern __shared__ float s_red[];  // [256/CW][2*C]
  const int RY = 256 / CW;
  const int cx = threadIdx.x % CW, ry = threadIdx.x / CW;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int c = cx; c < C; c += CW) {
    float s = 0.f, q = 0.f;
    float scale = 0.f, shift = 0.f, mean = 0.f, rstd = 0.f;
    // MODE 0: sums about the pivot = first row of the data (common.h); workgroup 0 publishes it for the finalize
    const float pv = (MODE == 0 && pivoted) ? y[c] : 0.f;
    if (MODE == 0 && pivoted && blockIdx.x == 0 && ry == 0) ws[SUG_PIVOT_OFFSET(C) + c] = pv;
    if (MODE == 1) {
      scale = coef[c]; shift = coef[C + c]; mean = coef[2 * C + c]; rstd = coef[3 * C + c];
    }
    for (int64_t r = r0 + ry; r < r1; r += RY) {
      if (MODE == 0) {
        const float v = y[r * ldy + c] - pv;
        s += v;
        q = fmaf(v, v, q);
      } else {
        const float zv = z[r * C + c];
        const float u = fmaf(scale, zv, shift);
        const float g = y[r * ldy + c] * (u > 0.f ? 1.f : slope);
        a[r * C + c] = scale * g;
        s += g;
        q = fmaf(g, (zv - mean) * rstd, q);
      }
    }
    s_red[(size_t)ry * 2 * C + c] = s;
    s_red[(size_t)ry * 2 * C + C + c] = q;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float acc = 0.f;
    for (int r = 0; r < RY; ++r) acc += s_red[(size_t)r * 2 * C + i];
    ws[(size_t)blockIdx.x * 2 * C + i] = acc;
  }
}

// float4 variant (C % 4 == 0, 16-B aligned rows): LX lanes along channels (float4 each), LY
// row-lanes, a workgroup covers `rows_per_block` rows; >= 1024 workgroups at the C2 sizes so
// that enough loads are in flight to approach the HBM rate (the scalar kernel above ran at
// ~0.9 TB/s with 128 workgroups).
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_vec4_kernel(const float* __restrict__ y, int64_t ldy,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ coef,
                                                              int64_t rows, int C, float slope,
                                                              float* __restrict__ a,
                                                              float* __restrict__ ws, int LX,
                                                              int rows_per_block, int pivoted) {
  extern __shared__ __attribute__((aligned(16))) float s_red[];  // [LY][2*C]
  float* const ws_pivot = ws + SUG_PIVOT_OFFSET(C) + (size_t)blockIdx.y * C;      // pivot row of this group
  // blockIdx.y = domain group: `rows` rows each, consecutive in y / z / a; coefficient set g; partial rows
  // [g * gridDim.x + blockIdx.x]
  const int64_t g0 = (int64_t)blockIdx.y * rows;
  y += g0 * ldy;
  if (MODE >= 1) {
    z += g0 * C;
    if (MODE == 1) a += g0 * C;
    coef += (int64_t)blockIdx.y * 5 * C;
  }
  ws += (size_t)blockIdx.y * gridDim.x * 2 * C;
  const int LY = 256 / LX;
  const int lx = threadIdx.x % LX, ly = threadIdx.x / LX;
  const int C4 = C >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (int c4 = lx; c4 < C4; c4 += LX) {
    const int c = c4 * 4;
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    float4 sc = s, sh = s, mean = s, rstd = s;
    // MODE 0: sums about the pivot = first row of the group's data (common.h); workgroup 0 publishes it
    float4 pv = s;
    if (MODE == 0 && pivoted) {
      pv = *reinterpret_cast<const float4*>(y + c);
      if (blockIdx.x == 0 && ly == 0) *reinterpret_cast<float4*>(ws_pivot + c) = pv;
    }
    if (MODE >= 1) {
      sc = *reinterpret_cast<const float4*>(coef + c);
      sh = *reinterpret_cast<const float4*>(coef + C + c);
      mean = *reinterpret_cast<const float4*>(coef + 2 * C + c);
      rstd = *reinterpret_cast<const float4*>(coef + 3 * C + c);
    }
#pragma unroll 4
    for (int64_t r = r0 + ly; r < r1; r += LY) {
      float4 v = *reinterpret_cast<const float4*>(y + r * ldy + c);
      if (MODE == 0) {
        v.x -= pv.x; v.y -= pv.y; v.z -= pv.z; v.w -= pv.w;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
      } else {
        const float4 zv = *reinterpret_cast<const float4*>(z + r * C + c);
        float4 g;
        g.x = v.x * (fmaf(sc.x, zv.x, sh.x) > 0.f ? 1.f : slope);
        g.y = v.y * (fmaf(sc.y, zv.y, sh.y) > 0.f ? 1.f : slope);
        g.z = v.z * (fmaf(sc.z, zv.z, sh.z) > 0.f ? 1.f : slope);
        g.w = v.w * (fmaf(sc.w, zv.w, sh.w) > 0.f ? 1.f : slope);
        if (MODE == 1)      // MODE 2: sums only (the apply kernel recomputes a = scale*G from gout and z)
          *reinterpret_cast<float4*>(a + r * C + c) = make_float4(sc.x * g.x, sc.y * g.y, sc.z * g.z, sc.w * g.w);
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        q.x = fmaf(g.x, (zv.x - mean.x) * rstd.x, q.x); q.y = fmaf(g.y, (zv.y - mean.y) * rstd.y, q.y);
        q.z = fmaf(g.z, (zv.z - mean.z) * rstd.z, q.z); q.w = fmaf(g.w, (zv.w - mean.w) * rstd.w, q.w);
      }
    }
    *reinterpret_cast<float4*>(s_red + (size_t)ly * 2 * C + c) = s;
    *reinterpret_cast<float4*>(s_red + (size_t)ly * 2 * C + C + c) = q;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float acc = 0.f;
    for (int r = 0; r < LY; ++r) acc += s_red[(size_t)r * 2 * C + i];
    ws[(size_t)blockIdx.x * 2 * C + i] = acc;
  }
}

// KK: compile-time k (0 = run-time): the reverse entries encode (n, j) as n*k + j, and a division by
// a run-time k costs ~20 instructions per entry and lane
template <int NCH, int KK>
__global__ __launch_bounds__(256) void edgeconv_bwd_scatter_kernel(
    const float* __restrict__ a, const uint8_t* __restrict__ arg, const float* __restrict__ s1,
    const float* __restrict__ pq, int64_t ldpq, const int32_t* __restrict__ rev_off,
    const int32_t* __restrict__ rev_ent, const float* __restrict__ coef,
    const double* __restrict__ red, int64_t BN, int N, int k, int Co, int LPP, int bpc, float invM,
    float* __restrict__ dpq, int64_t lddpq) {
  const int nchunk = Co >> 2;
  const int ppb = 256 / LPP;
  const int slot = threadIdx.x / LPP, ch0 = threadIdx.x % LPP;
  const int kk = KK ? KK : k;
  const float kf = (float)kk;
  // XCD-aware mapping as in edgeconv_fwd_kernel: a cloud's a / arg / Q rows stay in one L2
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int64_t b = (int64_t)(jb / bpc) * 8 + xcd;
  const int lb = jb % bpc;
  const int64_t B_ = BN / N;
  for (int m = lb * ppb + slot; b < B_ && m < N; m += bpc * ppb) {
    const int64_t p = b * N + m;
    const int32_t* offp = rev_off + b * (N + 1) + m;
    const int off = offp[0], cnt = offp[1] - offp[0];
    const int32_t* ent = rev_ent + b * (int64_t)N * kk + off;
    const int64_t rowb = b * N;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = ch0 + u * LPP;
      if (ch >= nchunk) continue;
      const int c = ch * 4;
      const float4 sc = ld4(coef + c), mean = ld4(coef + 2 * Co + c), rstd = ld4(coef + 3 * Co + c);
      const float dbx = (float)red[c + 0], dby = (float)red[c + 1], dbz = (float)red[c + 2], dbw = (float)red[c + 3];
      const float dgx = (float)red[Co + c + 0], dgy = (float)red[Co + c + 1];
      const float dgz = (float)red[Co + c + 2], dgw = (float)red[Co + c + 3];
      // f = scale/M, h = f*rstd*dgamma
      const float fx = sc.x * invM, fy = sc.y * invM, fz = sc.z * invM, fw = sc.w * invM;
      const float hx = fx * rstd.x * dgx, hy = fy * rstd.y * dgy, hz = fz * rstd.z * dgz, hw = fw * rstd.w * dgw;
      float ax = 0, ay = 0, az = 0, aw = 0;
      constexpr int GB = 10;      // entries per batch (8: 674 us, 10: 662 us, 12: 794 us per step at C2)
      for (int t0 = 0; t0 < cnt; t0 += GB) {
        int en[GB];
        float4 av[GB], qv[GB];
        uint32_t aj[GB];
#pragma unroll
        for (int t = 0; t < GB; ++t) en[t] = (t0 + t < cnt) ? ent[t0 + t] : -1;
#pragma unroll
        for (int t = 0; t < GB; ++t) {
          const int n = en[t] >= 0 ? en[t] / kk : 0;
          const int64_t o = (rowb + n) * Co + c;
          av[t] = ld4(a + o);
          aj[t] = *reinterpret_cast<const uint32_t*>(arg + o);
          qv[t] = ld4(pq + (rowb + n) * ldpq + Co + c);
        }
#pragma unroll
        for (int t = 0; t < GB; ++t) {
          if (en[t] >= 0) {
            const int n = en[t] / kk, j = en[t] - n * kk;
            ax += ((int)(aj[t] & 255u) == j ? av[t].x : 0.f) - hx * qv[t].x;
            ay += ((int)((aj[t] >> 8) & 255u) == j ? av[t].y : 0.f) - hy * qv[t].y;
            az += ((int)((aj[t] >> 16) & 255u) == j ? av[t].z : 0.f) - hz * qv[t].z;
            aw += ((int)(aj[t] >> 24) == j ? av[t].w : 0.f) - hw * qv[t].w;
          }
        }
      }
      const float4 pm = ld4(pq + p * ldpq + c);
      const float cf = (float)cnt;
      float4 dP, dQ;
      dP.x = ax - cf * (fx * dbx + hx * (pm.x - mean.x));
      dP.y = ay - cf * (fy * dby + hy * (pm.y - mean.y));
      dP.z = az - cf * (fz * dbz + hz * (pm.z - mean.z));
      dP.w = aw - cf * (fw * dbw + hw * (pm.w - mean.w));
      const float4 ap = ld4(a + p * Co + c), sp = ld4(s1 + p * Co + c);
      dQ.x = ap.x - (kf * fx * dbx + hx * (sp.x - kf * mean.x));
      dQ.y = ap.y - (kf * fy * dby + hy * (sp.y - kf * mean.y));
      dQ.z = ap.z - (kf * fz * dbz + hz * (sp.z - kf * mean.z));
      dQ.w = ap.w - (kf * fw * dbw + hw * (sp.w - kf * mean.w));
      st4(dpq + p * lddpq + c, dP);
      st4(dpq + p * lddpq + Co + c, dQ);
    }
  }
}

// LDS-resident backward (the transposed twin of edgeconv_fwd_lds_kernel).  The scatter above pulls, per reverse
// entry, the a / arg / Q rows of another point through L2 (9*Co bytes per entry, 20x the matrices: it runs at the
// L2's rate, not HBM's).  Here a workgroup = (cloud, 16-channel slice, part of the destination points) stages the
// slice of a, Q and the packed arg bytes of the whole cloud in LDS (N x 144 B = 144 KB at N = 1024) and every entry
// is served from there.  4 lanes per destination point, 16 points per wave; the points are visited in descending
// order of their reverse-list length (counting sort in LDS, over the part's own points) so that the 16 points a wave walks in lock-step have
// lists of similar length.  Which lane computes a point does not influence its value: the sums run over the
// point's own (sorted) reverse list, a-term and Q-sum in separate accumulators.
template <int SW, int KK>
__global__ __launch_bounds__(1024) void edgeconv_bwd_lds_kernel(
    const float* __restrict__ a, const uint8_t* __restrict__ arg, const float* __restrict__ s1,
    const float* __restrict__ pq, int64_t ldpq, const int32_t* __restrict__ rev_off,
    const int32_t* __restrict__ rev_ent, const float* __restrict__ coef_all, const double* __restrict__ red_all,
    int B, int N, int Co, int psplit, float invM, int Bg, int64_t coef_stride, int64_t red_stride,
    float* __restrict__ dpq, int64_t lddpq) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  constexpr int LP = SW / 4;                   // lanes per point
  constexpr int PPP = 1024 / LP;               // points per pass
  float* s_a = s_lds;                                                  // [N][SW]
  float* s_q = s_lds + (size_t)N * SW;                                 // [N][SW]
  uint32_t* s_g = reinterpret_cast<uint32_t*>(s_q + (size_t)N * SW);   // [N][LP] packed arg bytes
  int* s_hist = reinterpret_cast<int*>(s_g + (size_t)N * LP);          // [256] (+ scan scratch [256])
  uint16_t* s_perm = reinterpret_cast<uint16_t*>(s_hist + 512);        // [N]
  const int nslice = Co / SW;
  const int per_cloud = nslice * psplit;
  int b, r;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = (j / per_cloud) * 8 + xcd;
    r = j % per_cloud;
  } else {
    b = blockIdx.x / per_cloud;
    r = blockIdx.x % per_cloud;
  }
  const int sl = r % nslice, part = r / nslice;
  const int c0 = sl * SW;
  const int64_t rowb = (int64_t)b * N;
  // BatchNorm group of this cloud (clouds [g*Bg, (g+1)*Bg) share statistics)
  const float* coef = coef_all + (int64_t)(b / Bg) * coef_stride;
  const double* red = red_all + (int64_t)(b / Bg) * red_stride;
  for (int e = threadIdx.x; e < N * LP; e += 1024) {
    const int n = e / LP, c4 = e % LP;
    const int64_t o = (rowb + n) * Co + c0 + c4 * 4;
    st4(s_a + n * SW + c4 * 4, ld4(a + o));
    st4(s_q + n * SW + c4 * 4, ld4(pq + (rowb + n) * ldpq + Co + c0 + c4 * 4));
    s_g[n * LP + c4] = *reinterpret_cast<const uint32_t*>(arg + o);
  }
  // visiting order: descending reverse-list length (bins 0..255, longer lists share bin 0)
  const int32_t* offb = rev_off + (int64_t)b * (N + 1);
  if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
  __syncthreads();
  // a part owns the points m = part (mod psplit); the order inside a bin (set by the atomics) is free
  for (int m = part + psplit * (int)threadIdx.x; m < N; m += psplit * 1024) {
    const int c = offb[m + 1] - offb[m];
    atomicAdd(&s_hist[255 - (c < 255 ? c : 255)], 1);
  }
  __syncthreads();
  {
    int* scr = s_hist + 256;
    const int mine = threadIdx.x < 256 ? s_hist[threadIdx.x] : 0;
    if (threadIdx.x < 256) scr[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int t = 0;
      if (threadIdx.x < 256 && threadIdx.x >= o) t = scr[threadIdx.x - o];
      __syncthreads();
      if (threadIdx.x < 256) scr[threadIdx.x] += t;
      __syncthreads();
    }
    if (threadIdx.x < 256) s_hist[threadIdx.x] = scr[threadIdx.x] - mine;      // exclusive start of the bin
    __syncthreads();
  }
  for (int m = part + psplit * (int)threadIdx.x; m < N; m += psplit * 1024) {
    const int c = offb[m + 1] - offb[m];
    s_perm[atomicAdd(&s_hist[255 - (c < 255 ? c : 255)], 1)] = (uint16_t)m;
  }
  const int n_own = (N - part + psplit - 1) / psplit;
  const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
  const int c = c0 + lp * 4;
  const float4 sc = ld4(coef + c), mean = ld4(coef + 2 * Co + c), rstd = ld4(coef + 3 * Co + c);
  const float dbx = (float)red[c + 0], dby = (float)red[c + 1], dbz = (float)red[c + 2], dbw = (float)red[c + 3];
  const float dgx = (float)red[Co + c + 0], dgy = (float)red[Co + c + 1];
  const float dgz = (float)red[Co + c + 2], dgw = (float)red[Co + c + 3];
  const float fx = sc.x * invM, fy = sc.y * invM, fz = sc.z * invM, fw = sc.w * invM;
  const float hx = fx * rstd.x * dgx, hy = fy * rstd.y * dgy, hz = fz * rstd.z * dgz, hw = fw * rstd.w * dgw;
  const float kf = (float)KK;
  const int32_t* entb = rev_ent + (int64_t)b * N * KK;
  __syncthreads();
  for (int pos = slot; pos < n_own; pos += PPP) {
    const int m = s_perm[pos];
    const int off = offb[m], cnt = offb[m + 1] - off;
    const int32_t* ent = entb + off;
    float ax = 0, ay = 0, az = 0, aw = 0, qx = 0, qy = 0, qz = 0, qw = 0;
    constexpr int GB = 8;
    for (int t0 = 0; t0 < cnt; t0 += GB) {
      int en[GB];
#pragma unroll
      for (int t = 0; t < GB; ++t) en[t] = (t0 + t < cnt) ? ent[t0 + t] : -1;
#pragma unroll
      for (int t = 0; t < GB; ++t) {
        if (en[t] >= 0) {
          const int n = en[t] / KK, j = en[t] - n * KK;
          const float4 av = ld4(s_a + n * SW + lp * 4), qv = ld4(s_q + n * SW + lp * 4);
          const uint32_t g = s_g[n * LP + lp];
          ax += (int)(g & 255u) == j ? av.x : 0.f;
          ay += (int)((g >> 8) & 255u) == j ? av.y : 0.f;
          az += (int)((g >> 16) & 255u) == j ? av.z : 0.f;
          aw += (int)(g >> 24) == j ? av.w : 0.f;
          qx += qv.x; qy += qv.y; qz += qv.z; qw += qv.w;
        }
      }
    }
    const int64_t p = rowb + m;
    const float4 pm = ld4(pq + p * ldpq + c);
    const float cf = (float)cnt;
    float4 dP, dQ;
    dP.x = ax - (hx * qx + cf * (fx * dbx + hx * (pm.x - mean.x)));
    dP.y = ay - (hy * qy + cf * (fy * dby + hy * (pm.y - mean.y)));
    dP.z = az - (hz * qz + cf * (fz * dbz + hz * (pm.z - mean.z)));
    dP.w = aw - (hw * qw + cf * (fw * dbw + hw * (pm.w - mean.w)));
    const float4 ap = ld4(s_a + m * SW + lp * 4), sp = ld4(s1 + p * Co + c);
    dQ.x = ap.x - (kf * fx * dbx + hx * (sp.x - kf * mean.x));
    dQ.y = ap.y - (kf * fy * dby + hy * (sp.y - kf * mean.y));
    dQ.z = ap.z - (kf * fz * dbz + hz * (sp.z - kf * mean.z));
    dQ.w = ap.w - (kf * fw * dbw + hw * (sp.w - kf * mean.w));
    st4(dpq + p * lddpq + c, dP);
    st4(dpq + p * lddpq + Co + c, dQ);
  }
}

inline int lanes_per_point(int Co) {
  int lpp = 1;
  while (lpp < (Co >> 2) && lpp < 64) lpp <<= 1;
  return lpp;
}

inline int col_width(int C) {
  int cw = 1;
  while (cw < C && cw < 256) cw <<= 1;
  return cw;
}

}  // namespace

int sug_reduce_partials(const float* ws, int nblk, int W, double* out, hipStream_t st) {
  SUG_REQUIRE(ws && out && nblk > 0 && W > 0, "sug_reduce_partials: bad argument");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(sug_divup(W, 16)), dim3(256), 0, st, ws, nblk, W, out);
  SUG_LAUNCH_CHECK("sug_reduce_partials");
  return SUG_OK;
}

int sug_stats_finalize(const float* ws, int nblk, int C, const float* gamma, const float* beta, double count, float eps,
                       float momentum, float* running_mean, float* running_var, float* coef, hipStream_t st,
                       const float* pivot) {
  SUG_REQUIRE(ws && gamma && beta && coef && nblk > 0 && C > 0 && count > 0, "sug_stats_finalize: bad argument");
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(sug_divup(C, 8)), dim3(256), 0, st, ws, nblk, C, gamma, beta, count, eps,
                     momentum, running_mean, running_var, coef, pivot);
  SUG_LAUNCH_CHECK("sug_stats_finalize");
  return SUG_OK;
}

// producer only: per-workgroup partial rows in ws, their number in *nblk
// Bg > 0: sums about the pivot of each domain group of Bg clouds (LDS-resident kernel only; the caller folds with
// sug_edgeconv_bn_act, which reads the pivot rows); Bg = 0: plain sums
static int edgeconv_fwd_partials(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma, int B, int N,
                                 int k, int Co, float* z, uint8_t* arg, float* s1, float* ws, int* nblk,
                                 void* stream, int Bg = 0) {
  SUG_REQUIRE(pq && idx && gamma && z && arg && ws, "sug_edgeconv_fwd: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 255, "sug_edgeconv_fwd: bad shape B=%d N=%d k=%d", B, N, k);
  SUG_REQUIRE(Co > 0 && Co % 4 == 0 && Co <= 1024, "sug_edgeconv_fwd: Co=%d must be a multiple of 4, <= 1024", Co);
  SUG_REQUIRE(ldpq >= 2 * Co && ldpq % 4 == 0, "sug_edgeconv_fwd: ldpq=%lld", (long long)ldpq);
  SUG_REQUIRE(((uintptr_t)pq % 16) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                  (!s1 || ((uintptr_t)s1 % 16) == 0) && ((uintptr_t)arg % 4) == 0,
              "sug_edgeconv_fwd: pointers must be 16-byte aligned");
  hipStream_t st_ = (hipStream_t)stream;
#ifndef SUG_EDGECONV_NO_LDS
  // LDS-resident gathers when a 16-channel slice of one cloud's P matrix fits (N <= 2048)
  if (k == 20 && Co % 16 == 0 && (size_t)N * 16 * 4 + 64 * 32 * 4 <= 150 * 1024 && B <= SUG_STATS_BLOCKS / 4 &&
      ((uintptr_t)idx % 16) == 0) {
    const int nslice = Co / 16;
    int psplit = 512 / (B * nslice);
    psplit = psplit < 1 ? 1 : (psplit > 4 ? 4 : psplit);
    const size_t sh = (size_t)N * 16 * sizeof(float) + 64 * 32 * sizeof(float);
    static SugLdsOptIn note;
    if (int rc = sug_allow_dynamic_lds(note, &edgeconv_fwd_lds_kernel<16, 20>, 150 * 1024, "sug_edgeconv_fwd(lds)")) return rc;
    const int grid = ((B & 7) == 0 ? B : B) * nslice * psplit;
    hipLaunchKernelGGL((edgeconv_fwd_lds_kernel<16, 20>), dim3(grid), dim3(256), sh, st_, pq, ldpq, idx, gamma, B, N, Co,
                       psplit, z, arg, s1, ws, Bg);
    SUG_LAUNCH_CHECK("sug_edgeconv_fwd(lds)");
    *nblk = B * psplit;
    return SUG_OK;
  }
#endif
  const int lpp = lanes_per_point(Co);
  const int nch = sug_divup(Co >> 2, lpp);
  const int64_t BN = (int64_t)B * N;
  const int ppb = 256 / lpp;
  // grid = 8 XCD residues x ceil(B/8) clouds per residue x bpc workgroups per cloud (<= SUG_STATS_BLOCKS)
  const int cpx = sug_divup(B, 8);
  int bpc = sug_divup(N, ppb);
  if (bpc > SUG_STATS_BLOCKS / (8 * cpx)) bpc = SUG_STATS_BLOCKS / (8 * cpx);
  SUG_REQUIRE(bpc >= 1, "sug_edgeconv_fwd: B=%d too large", B);
  const int grid = 8 * cpx * bpc;
  const size_t sh = (size_t)ppb * 2 * Co * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (nch == 1)
    hipLaunchKernelGGL((edgeconv_fwd_kernel<1>), dim3(grid), dim3(256), sh, st, pq, ldpq, idx, gamma, BN, N, k, Co, lpp, bpc, z, arg, s1, ws);
  else if (nch == 2)
    hipLaunchKernelGGL((edgeconv_fwd_kernel<2>), dim3(grid), dim3(256), sh, st, pq, ldpq, idx, gamma, BN, N, k, Co, lpp, bpc, z, arg, s1, ws);
  else
    hipLaunchKernelGGL((edgeconv_fwd_kernel<4>), dim3(grid), dim3(256), sh, st, pq, ldpq, idx, gamma, BN, N, k, Co, lpp, bpc, z, arg, s1, ws);
  SUG_LAUNCH_CHECK("sug_edgeconv_fwd");
  *nblk = grid;
  return SUG_OK;
}

extern "C" int sug_edgeconv_fwd(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                                int B, int N, int k, int Co, float* z, uint8_t* arg, float* s1,
                                double* stats, float* ws, void* stream) {
  SUG_REQUIRE(stats, "sug_edgeconv_fwd: null pointer");
  int grid = 0;
  const int rc = edgeconv_fwd_partials(pq, ldpq, idx, gamma, B, N, k, Co, z, arg, s1, ws, &grid, stream);
  if (rc != SUG_OK) return rc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(sug_divup(2 * Co, 16)), dim3(256), 0, (hipStream_t)stream, ws, grid,
                     2 * Co, stats);
  SUG_LAUNCH_CHECK("sug_edgeconv_fwd(reduce)");
  return SUG_OK;
}

// EdgeConv forward + BatchNorm coefficients (batch statistics folded and finalised in one launch)
extern "C" int sug_edgeconv_fwd_bn(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                                   const float* beta, int B, int N, int k, int Co, float eps, float momentum,
                                   float* running_mean, float* running_var, float* z, uint8_t* arg, float* s1,
                                   float* coef, float* ws, void* stream) {
  SUG_REQUIRE(beta && coef, "sug_edgeconv_fwd_bn: null pointer");
  int grid = 0;
  const int rc = edgeconv_fwd_partials(pq, ldpq, idx, gamma, B, N, k, Co, z, arg, s1, ws, &grid, stream);
  if (rc != SUG_OK) return rc;
  // plain sums here (no pivot): y = P[idx] + Q comes from bias-free convolutions of normalised activations, its mean
  // is of the order of its spread, and the un-shifted maximum keeps z bit-identical to the validated kernel
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(sug_divup(Co, 8)), dim3(256), 0, (hipStream_t)stream, ws, grid, Co,
                     gamma, beta, (double)B * N * k, eps, momentum, running_mean, running_var, coef, nullptr);
  SUG_LAUNCH_CHECK("sug_edgeconv_fwd_bn(finalize)");
  return SUG_OK;
}

// EdgeConv forward + BatchNorm coefficients + activation for `groups` domain groups (B/groups clouds each, one
// BatchNorm call per group) in three launches when the LDS-resident kernel applies to the whole batch: its
// partial rows are (cloud, part)-major, so each group's rows are contiguous.  Otherwise: group by group.
int sug_edgeconv_bn_act(const float* ws, int nblk, int Co, int groups, const float* gamma, const float* beta, double count,
                        float eps, float momentum, float* running_mean, float* running_var, float* coef, const float* z,
                        int64_t rows_g, float slope, float* out, int64_t ldo, hipStream_t st);

int sug_edgeconv_fwd_bn_act_groups(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                                   const float* beta, int B, int N, int k, int Co, int groups, float eps, float momentum,
                                   float slope, float* running_mean, float* running_var, float* z, uint8_t* arg,
                                   float* s1, float* coef, float* out, int64_t ldo, float* ws, void* stream) {
  SUG_REQUIRE(beta && coef && out, "sug_edgeconv_layer_fwd: null pointer");
  SUG_REQUIRE(groups >= 1 && B % groups == 0, "sug_edgeconv_layer_fwd: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(groups <= 16, "sug_edgeconv_layer_fwd: %d groups, the workspace reserves pivot rows for 16", groups);
  hipStream_t st = (hipStream_t)stream;
  const int Bg = B / groups;
  const int64_t rows = (int64_t)Bg * N;
  const bool lds = k == 20 && Co % 16 == 0 && (size_t)N * 16 * 4 + 64 * 32 * 4 <= 150 * 1024 && B <= SUG_STATS_BLOCKS / 4 &&
                   ((uintptr_t)idx % 16) == 0;
  const bool vec = (ldo % 4 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)coef % 16 == 0) && ((uintptr_t)z % 16 == 0);
#ifdef SUG_EDGECONV_NO_LDS
  const bool one = false;
#else
  const bool one = lds && vec;
#endif
  if (one) {
    // two launches: gather / reduce with per-(cloud, part) partial rows, then statistics fold + BatchNorm + activation
    // (edgeconv_fused.hip: every workgroup of the second launch folds its group's partial rows itself)
    int nblk = 0;
    if (int rc = edgeconv_fwd_partials(pq, ldpq, idx, gamma, B, N, k, Co, z, arg, s1, ws, &nblk, stream, Bg)) return rc;
    return sug_edgeconv_bn_act(ws, nblk / groups, Co, groups, gamma, beta, (double)rows * k, eps, momentum, running_mean,
                               running_var, coef, z, rows, slope, out, ldo, st);
  }
  for (int g = 0; g < groups; ++g) {
    const int64_t r0 = (int64_t)g * rows;
    float* cg = coef + (int64_t)g * 5 * Co;
    if (int rc = sug_edgeconv_fwd_bn(pq + r0 * ldpq, ldpq, idx + r0 * k, gamma, beta, Bg, N, k, Co, eps, momentum, running_mean,
                                     running_var, z + r0 * Co, arg + r0 * Co, s1 ? s1 + r0 * Co : nullptr, cg, ws, stream))
      return rc;
    if (int rc = sug_affine_act(z + r0 * Co, Co, cg, rows, Co, slope, out + r0 * ldo, ldo, stream)) return rc;
  }
  return SUG_OK;
}

extern "C" int sug_bn_finalize(const double* stats, const float* gamma, const float* beta, int C,
                               double count, float eps, float momentum, float* running_mean,
                               float* running_var, float* coef, void* stream) {
  SUG_REQUIRE(stats && gamma && beta && coef, "sug_bn_finalize: null pointer");
  SUG_REQUIRE(C > 0 && count > 0, "sug_bn_finalize: bad shape");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(sug_divup(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     stats, gamma, beta, C, count, eps, momentum, running_mean, running_var, coef);
  SUG_LAUNCH_CHECK("sug_bn_finalize");
  return SUG_OK;
}

extern "C" int sug_bn_replay(const float* coef, int G, int C, float momentum, float* running_mean,
                             float* running_var, void* stream) {
  SUG_REQUIRE(coef && running_mean && running_var, "sug_bn_replay: null pointer");
  SUG_REQUIRE(G > 0 && C > 0, "sug_bn_replay: bad shape");
  hipLaunchKernelGGL(bn_replay_kernel, dim3(sug_divup(C, 256)), dim3(256), 0, (hipStream_t)stream, coef, G, C,
                     momentum, running_mean, running_var);
  SUG_LAUNCH_CHECK("sug_bn_replay");
  return SUG_OK;
}

extern "C" int sug_affine_act(const float* z, int64_t ldz, const float* coef, int64_t rows, int C,
                              float slope, float* out, int64_t ldo, void* stream) {
  SUG_REQUIRE(z && coef && out, "sug_affine_act: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && ldz >= C && ldo >= C, "sug_affine_act: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ldz % 4 == 0) && (ldo % 4 == 0) && ((uintptr_t)z % 16 == 0) &&
                   ((uintptr_t)out % 16 == 0) && ((uintptr_t)coef % 16 == 0);
  const int64_t total = vec ? rows * (C / 4) : rows * C;
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (vec)
    hipLaunchKernelGGL(affine_act_vec4_kernel, dim3((int)g), dim3(256), 0, st, z, ldz, coef, rows, C, slope, out, ldo);
  else
    hipLaunchKernelGGL(affine_act_kernel, dim3((int)g), dim3(256), 0, st, z, ldz, coef, rows, C, slope, out, ldo);
  SUG_LAUNCH_CHECK("sug_affine_act");
  return SUG_OK;
}

// Launch the column reduction (vectorised when layout allows); returns the number of partial rows.
template <int MODE>
static int launch_col_reduce(const float* y, int64_t ldy, const float* z, const float* coef, int64_t rows,
                             int C, float slope, float* a, float* ws, hipStream_t st, int groups = 1, int pivoted = 0);

// rows per block such that the grid stays within SUG_STATS_BLOCKS
static int col_rows_per_block(int64_t rows, int cw) {
  int64_t rpb = 64 * (256 / cw) > 256 ? 64 * (256 / cw) : 256;
  while ((rows + rpb - 1) / rpb > SUG_STATS_ROWS) rpb *= 2;
  return (int)rpb;
}

// rows = rows per group; groups > 1 (vectorised layout only, else -1): one launch over all groups, partial rows
// [group][block]; returns the number of partial rows PER GROUP
template <int MODE>
static int launch_col_reduce(const float* y, int64_t ldy, const float* z, const float* coef, int64_t rows,
                             int C, float slope, float* a, float* ws, hipStream_t st, int groups, int pivoted) {
  const bool vec = (C % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)y % 16 == 0) &&
                   (MODE == 0 || (((uintptr_t)z % 16 == 0) && (MODE == 2 || (uintptr_t)a % 16 == 0) && ((uintptr_t)coef % 16 == 0)));
  if (vec) {
    int lx = 1;
    while (lx < (C >> 2) && lx < 256) lx <<= 1;
    const int ly = 256 / lx;
    int64_t rpb = (rows * groups + SUG_STATS_ROWS - 1) / SUG_STATS_ROWS;
    if (rpb < 4 * ly) rpb = 4 * ly;
    rpb = (rpb + ly - 1) / ly * ly;
    const int grid = sug_divup(rows, rpb);
    if ((int64_t)grid * groups > SUG_STATS_ROWS || groups > 16) return -1;
    hipLaunchKernelGGL((col_reduce_vec4_kernel<MODE>), dim3(grid, groups), dim3(256), (size_t)ly * 2 * C * sizeof(float),
                       st, y, ldy, z, coef, rows, C, slope, a, ws, lx, (int)rpb, pivoted);
    return grid;
  }
  if (groups > 1 || MODE == 2) return -1;
  const int cw = col_width(C);
  const int rpb = col_rows_per_block(rows, cw);
  const int grid = sug_divup(rows, rpb);
  hipLaunchKernelGGL((col_reduce_kernel<MODE>), dim3(grid), dim3(256), (size_t)(256 / cw) * 2 * C * sizeof(float),
                     st, y, ldy, z, coef, rows, C, slope, a, ws, cw, rpb, pivoted);
  return grid;
}

extern "C" int sug_col_stats(const float* y, int64_t ldy, int64_t rows, int C, double* stats,
                             float* ws, void* stream) {
  SUG_REQUIRE(y && stats && ws, "sug_col_stats: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && C <= 4096 && ldy >= C, "sug_col_stats: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = launch_col_reduce<0>(y, ldy, nullptr, nullptr, rows, C, 0.f, nullptr, ws, st);
  SUG_LAUNCH_CHECK("sug_col_stats");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(sug_divup(2 * C, 16)), dim3(256), 0, st, ws, grid, 2 * C, stats);
  SUG_LAUNCH_CHECK("sug_col_stats(reduce)");
  return SUG_OK;
}

// Column statistics of y [rows,C] + BatchNorm coefficients in two launches (sug_col_stats + sug_bn_finalize fused)
extern "C" int sug_col_stats_bn(const float* y, int64_t ldy, int64_t rows, int C, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean,
                                float* running_var, float* coef, float* ws, void* stream) {
  SUG_REQUIRE(y && gamma && beta && coef && ws, "sug_col_stats_bn: null pointer");
  SUG_REQUIRE(rows > 0 && C > 0 && C <= 4096 && ldy >= C, "sug_col_stats_bn: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = launch_col_reduce<0>(y, ldy, nullptr, nullptr, rows, C, 0.f, nullptr, ws, st, 1, 1);
  SUG_LAUNCH_CHECK("sug_col_stats_bn");
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(sug_divup(C, 8)), dim3(256), 0, st, ws, grid, C, gamma, beta,
                     (double)rows, eps, momentum, running_mean, running_var, coef, ws + SUG_PIVOT_OFFSET(C));
  SUG_LAUNCH_CHECK("sug_col_stats_bn(finalize)");
  return SUG_OK;
}

// Grouped forms (one launch over `groups` domain groups of `rows` rows each; return 1 when the layout does not
// allow it and the caller must go group by group):
//   statistics + BatchNorm coefficients of y -> coef [groups,5,C], running buffers updated in group order
int sug_col_stats_bn_groups(const float* y, int64_t ldy, int64_t rows, int C, int groups, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                            float* coef, float* ws, hipStream_t st) {
  const int grid = launch_col_reduce<0>(y, ldy, nullptr, nullptr, rows, C, 0.f, nullptr, ws, st, groups, 1);
  if (grid < 0) return 1;
  SUG_LAUNCH_CHECK("sug_bn_act_rows_fwd(stats)");
  hipLaunchKernelGGL(stats_finalize_groups_kernel, dim3(sug_divup(C, 8)), dim3(1024), 0, st, ws, grid, C, groups, gamma, beta,
                     (double)rows, eps, momentum, running_mean, running_var, coef, ws + SUG_PIVOT_OFFSET(C));
  SUG_LAUNCH_CHECK("sug_bn_act_rows_fwd(finalize)");
  return SUG_OK;
}
//   out = act(coef_g . z) with the coefficient set of each row's group
int sug_affine_act_groups(const float* z, int64_t ldz, const float* coef, int64_t rows, int groups, int C, float slope,
                          float* out, int64_t ldo, hipStream_t st) {
  const bool vec = (C % 4 == 0) && (ldz % 4 == 0) && (ldo % 4 == 0) && ((uintptr_t)z % 16 == 0) &&
                   ((uintptr_t)out % 16 == 0) && ((uintptr_t)coef % 16 == 0);
  if (!vec) return 1;
  const int64_t total = rows * groups * (C / 4);
  int64_t g = (total + 1023) / 1024;             // 4 float4 per thread and iteration
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(affine_act_vec4_groups_kernel, dim3((int)g), dim3(256), 0, st, z, ldz, coef, rows * groups, rows, C,
                     slope, out, ldo);
  SUG_LAUNCH_CHECK("sug_bn_act_rows_fwd(act)");
  return SUG_OK;
}
//   a = scale*G and the BatchNorm backward sums red [groups, 2C]
int sug_bwd_reduce_groups(const float* gout, int64_t ldg, const float* z, const float* coef, int64_t rows, int Co,
                          int groups, float slope, float* a, double* red, float* ws, hipStream_t st, float* dgb) {
  // a == nullptr: sums only;  dgb (may be null): dbeta | dgamma [2Co] summed over the groups, fp32 (sug_fold_groups of red)
  const int grid = a ? launch_col_reduce<1>(gout, ldg, z, coef, rows, Co, slope, a, ws, st, groups)
                     : launch_col_reduce<2>(gout, ldg, z, coef, rows, Co, slope, nullptr, ws, st, groups);
  if (grid < 0) return 1;
  SUG_LAUNCH_CHECK("sug_edgeconv_bwd_reduce");
  if (dgb)
    hipLaunchKernelGGL(reduce_partials_fold_kernel, dim3(sug_divup(2 * Co, 16)), dim3(256), 0, st, ws, grid, 2 * Co, groups, red, dgb);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(sug_divup(2 * Co, 16), groups), dim3(256), 0, st, ws, grid, 2 * Co, red);
  SUG_LAUNCH_CHECK("sug_edgeconv_bwd_reduce(reduce)");
  return SUG_OK;
}

extern "C" int sug_edgeconv_bwd_reduce(const float* gout, int64_t ldg, const float* z, const float* coef,
                                       int64_t rows, int Co, float slope, float* a, double* red,
                                       float* ws, void* stream) {
  SUG_REQUIRE(gout && z && coef && a && red && ws, "sug_edgeconv_bwd_reduce: null pointer");
  SUG_REQUIRE(rows > 0 && Co > 0 && Co <= 4096 && ldg >= Co, "sug_edgeconv_bwd_reduce: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int grid = launch_col_reduce<1>(gout, ldg, z, coef, rows, Co, slope, a, ws, st);
  SUG_LAUNCH_CHECK("sug_edgeconv_bwd_reduce");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(sug_divup(2 * Co, 16)), dim3(256), 0, st, ws, grid, 2 * Co, red);
  SUG_LAUNCH_CHECK("sug_edgeconv_bwd_reduce(reduce)");
  return SUG_OK;
}

// groups > 1: the B clouds are `groups` BatchNorm groups of B/groups clouds; group g reads coef + g*coef_stride and
// red + g*red_stride (strides in elements; a zero red_stride shares one row, e.g. the zero row of eval mode)
static int edgeconv_bwd_scatter_groups(const float* a, const uint8_t* arg, const float* s1, const float* pq,
                                       int64_t ldpq, const int32_t* rev_off, const int32_t* rev_ent, const float* coef,
                                       const double* red, int B, int N, int k, int Co, int groups, int64_t coef_stride,
                                       int64_t red_stride, float* dpq, int64_t lddpq, void* stream);

extern "C" int sug_edgeconv_bwd_scatter(const float* a, const uint8_t* arg, const float* s1,
                                        const float* pq, int64_t ldpq, const int32_t* rev_off,
                                        const int32_t* rev_ent, const float* coef, const double* red,
                                        int B, int N, int k, int Co, float* dpq, int64_t lddpq,
                                        void* stream) {
  return edgeconv_bwd_scatter_groups(a, arg, s1, pq, ldpq, rev_off, rev_ent, coef, red, B, N, k, Co, 1, 0, 0, dpq, lddpq,
                                     stream);
}

int sug_edgeconv_bwd_scatter_groups(const float* a, const uint8_t* arg, const float* s1, const float* pq, int64_t ldpq,
                                    const int32_t* rev_off, const int32_t* rev_ent, const float* coef, const double* red,
                                    int B, int N, int k, int Co, int groups, int64_t coef_stride, int64_t red_stride,
                                    float* dpq, int64_t lddpq, void* stream) {
  return edgeconv_bwd_scatter_groups(a, arg, s1, pq, ldpq, rev_off, rev_ent, coef, red, B, N, k, Co, groups, coef_stride,
                                     red_stride, dpq, lddpq, stream);
}

static int edgeconv_bwd_scatter_groups(const float* a, const uint8_t* arg, const float* s1, const float* pq,
                                       int64_t ldpq, const int32_t* rev_off, const int32_t* rev_ent, const float* coef,
                                       const double* red, int B, int N, int k, int Co, int groups, int64_t coef_stride,
                                       int64_t red_stride, float* dpq, int64_t lddpq, void* stream) {
  SUG_REQUIRE(a && arg && s1 && pq && rev_off && rev_ent && coef && red && dpq,
              "sug_edgeconv_bwd_scatter: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && k > 0 && k <= 255, "sug_edgeconv_bwd_scatter: bad shape");
  SUG_REQUIRE(groups >= 1 && B % groups == 0, "sug_edgeconv_bwd_scatter: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(Co > 0 && Co % 4 == 0 && Co <= 1024, "sug_edgeconv_bwd_scatter: Co=%d", Co);
  SUG_REQUIRE(ldpq >= 2 * Co && ldpq % 4 == 0 && lddpq >= 2 * Co && lddpq % 4 == 0,
              "sug_edgeconv_bwd_scatter: bad row strides");
  const int Bg = B / groups;
  const int64_t BN = (int64_t)Bg * N;                 // rows behind one set of BatchNorm statistics
  const float invM = (float)(1.0 / ((double)BN * k));
  hipStream_t st = (hipStream_t)stream;
#ifndef SUG_EDGECONV_NO_LDS
  // LDS-resident gathers when a 16-channel slice of one cloud's a, Q and arg fits (N <= 1024)
  const size_t sh_lds = (size_t)N * 16 * 9 + 512 * sizeof(int) + (size_t)N * sizeof(uint16_t);
  if (k == 20 && Co % 16 == 0 && N <= 65535 && sh_lds <= 158 * 1024 && !getenv("SUG_EDGECONV_BWD_LEGACY") &&
      ((uintptr_t)a % 16) == 0 && ((uintptr_t)pq % 16) == 0 && ((uintptr_t)arg % 4) == 0 && ((uintptr_t)dpq % 16) == 0 &&
      ((uintptr_t)s1 % 16) == 0 && ((uintptr_t)coef % 16) == 0) {
    const int nslice = Co / 16;
    int psplit = 256 / (B * nslice);             // one workgroup per CU (144 KB of LDS each): fill the chip once
    psplit = psplit < 1 ? 1 : (psplit > 4 ? 4 : psplit);
    static SugLdsOptIn note;
    if (int rc = sug_allow_dynamic_lds(note, &edgeconv_bwd_lds_kernel<16, 20>, 158 * 1024, "sug_edgeconv_bwd_scatter(lds)")) return rc;
    hipLaunchKernelGGL((edgeconv_bwd_lds_kernel<16, 20>), dim3(B * nslice * psplit), dim3(1024), sh_lds, st, a, arg, s1, pq,
                       ldpq, rev_off, rev_ent, coef, red, B, N, Co, psplit, invM, Bg, coef_stride, red_stride, dpq, lddpq);
    SUG_LAUNCH_CHECK("sug_edgeconv_bwd_scatter(lds)");
    return SUG_OK;
  }
#endif
  // generic path: one launch per BatchNorm group
  const int lpp = lanes_per_point(Co);
  const int nch = sug_divup(Co >> 2, lpp);
  const int ppb = 256 / lpp;
  const int cpx = sug_divup(Bg, 8);
  int bpc = sug_divup(N, ppb);
  while (bpc > 1 && (int64_t)8 * cpx * bpc > 8192) bpc = (bpc + 1) / 2;
  const int grid = 8 * cpx * bpc;
#define LAUNCH_SCATTER(NC) do { \
    if (k == 20) hipLaunchKernelGGL((edgeconv_bwd_scatter_kernel<NC, 20>), dim3(grid), dim3(256), 0, st, ag, argg, s1g, pqg, ldpq, offg, entg, cg, rg, BN, N, k, Co, lpp, bpc, invM, dpqg, lddpq); \
    else hipLaunchKernelGGL((edgeconv_bwd_scatter_kernel<NC, 0>), dim3(grid), dim3(256), 0, st, ag, argg, s1g, pqg, ldpq, offg, entg, cg, rg, BN, N, k, Co, lpp, bpc, invM, dpqg, lddpq); \
  } while (0)
  for (int g = 0; g < groups; ++g) {
    const int64_t r0 = (int64_t)g * BN;
    const float *ag = a + r0 * Co, *s1g = s1 + r0 * Co, *pqg = pq + r0 * ldpq, *cg = coef + g * coef_stride;
    const uint8_t* argg = arg + r0 * Co;
    const int32_t *offg = rev_off + (int64_t)g * Bg * (N + 1), *entg = rev_ent + r0 * k;
    const double* rg = red + g * red_stride;
    float* dpqg = dpq + r0 * lddpq;
    if (nch == 1)
      LAUNCH_SCATTER(1);
    else if (nch == 2)
      LAUNCH_SCATTER(2);
    else
      LAUNCH_SCATTER(4);
  }
  SUG_LAUNCH_CHECK("sug_edgeconv_bwd_scatter");
  return SUG_OK;
}

extern "C" int sug_bn_replay_multi(int n, const void* const* coef, const int32_t* G, const int32_t* C, const float* momentum,
                                   void* const* running_mean, void* const* running_var, void* stream) {
  SUG_REQUIRE(coef && G && C && momentum && running_mean && running_var, "sug_bn_replay_multi: null pointer");
  SUG_REQUIRE(n >= 1 && n <= SUG_REPLAY_MULTI, "sug_bn_replay_multi: %d layers (1..%d)", n, SUG_REPLAY_MULTI);
  ReplayMulti a;
  int cmax = 0;
  for (int i = 0; i < SUG_REPLAY_MULTI; ++i) {
    const int k = i < n ? i : 0;
    SUG_REQUIRE(coef[k] && running_mean[k] && running_var[k] && G[k] >= 1 && C[k] >= 1, "sug_bn_replay_multi: bad layer %d", k);
    a.coef[i] = (const float*)coef[k];
    a.rmean[i] = (float*)running_mean[k];
    a.rvar[i] = (float*)running_var[k];
    a.G[i] = G[k];
    a.C[i] = C[k];
    a.momentum[i] = momentum[k];
    if (i < n && C[k] > cmax) cmax = C[k];
  }
  hipLaunchKernelGGL(bn_replay_multi_kernel, dim3(sug_divup(cmax, 256), n), dim3(256), 0, (hipStream_t)stream, a);
  SUG_LAUNCH_CHECK("sug_bn_replay_multi");
  return SUG_OK;
}
