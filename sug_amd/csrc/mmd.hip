// Gaussian multi-kernel MMD as an LDS-tiled (2m x 2m) pairwise reduction, and the
// Chamfer distance used for the SDA geometric weights.
// Reference: _mix_rbf_kernel + _mmd2(biased=True), model/mmd.py:239-254, :274-312;
// geometric_weights, model/mmd.py:107-131 (third-party op there; parity unpinned).
//
// Each workgroup owns a 16x16 tile of (i,j) pairs, streams the two 16-row panels of
// Z through LDS in chunks of DK features and accumulates <z_i,z_j>, |z_i|^2, |z_j|^2
// as identical ascending fma chains (so e_ii == 0 exactly, like diag(ZZ^T)).
// Algorithmic bytes 4*2m*D (+4m weights); FLOPs 2*(2m)^2*D.
#include "common.h"

namespace {

constexpr int TI = 16;
constexpr int DK = 64;

// Kernel value, its derivative coefficient and the contribution of pair (i,j) to the three sums.
// Row blocks (batch-sharded global MMD): a rank evaluates only the rows of Z = [X;Y] that
// belong to its own samples -- local row li in [0, 2*mloc) is global row row0 + li (li < mloc: its
// X rows) or m + row0 + li - mloc (its Y rows) -- against all 2m columns.  row0 = 0, mloc = m is the
// whole matrix.  wt is stored by LOCAL row: [2*mloc, 2m].
__device__ __forceinline__ int global_row(int li, int m, int row0, int mloc) {
  return li < mloc ? row0 + li : m + row0 + (li - mloc);
}

__device__ __forceinline__ void pair_epilogue(int li, int i, int j, int m, float g, float ni, float nj,
                                              const float* __restrict__ w, const float* __restrict__ neg_gamma,
                                              int ns, float* __restrict__ wt, float& kxx, float& kyy, float& kxy) {
  const int M2 = 2 * m;
  {
    // exponent = Z_norm_sqr - 2*ZZT + Z_norm_sqr.t()   (model/mmd.py:247)
    const float e = __fadd_rn(__fsub_rn(ni, __fmul_rn(2.0f, g)), nj);
    float K = 0.f, Kp = 0.f;
    for (int s = 0; s < ns; ++s) {
      const float ng = neg_gamma[s];
      const float t = expf(__fmul_rn(ng, e));
      K += t;
      Kp = fmaf(ng, t, Kp);
    }
    const float inv_m2 = 1.0f / ((float)m * (float)m);
    float cf;  // symmetrised coefficient c_ij + c_ji of K_ij in mmd2
    if (i < m && j < m) {
      kxx = K;
      cf = 2.0f * inv_m2;
    } else if (i >= m && j >= m) {
      kyy = K;
      cf = 2.0f * inv_m2;
    } else if (i < m) {  // i in X, j in Y: weight belongs to column j-m
      const float wj = w ? w[j - m] : 1.0f;
      kxy = wj * K;
      cf = -2.0f * wj * inv_m2;
    } else {  // i in Y, j in X (mirror; contributes to wt only)
      const float wi = w ? w[i - m] : 1.0f;
      cf = -2.0f * wi * inv_m2;
    }
    // e_ii is identically 0: no gradient, and keeping cf*Kp_ii (huge) on the diagonal would
    // cancel catastrophically in dZ = 2*(diag(rowsum(wt)) - wt).Z
    if (wt) wt[(int64_t)li * M2 + j] = (i == j) ? 0.f : cf * Kp;
  }
}

__device__ __forceinline__ void mmd_rbf_body(const int bx, const int by, const float* __restrict__ z, int64_t ldz, int m,
                                             int D, const float* __restrict__ w,
                                             const float* __restrict__ neg_gamma, int ns, int row0, int mloc,
                                             double* __restrict__ sums,
                                             float* __restrict__ wt) {
  __shared__ float s_a[TI][DK + 1];
  __shared__ float s_b[TI][DK + 1];
  __shared__ double s_sum[256 / WAVE][3];          // per-wave partial sums, combined in wave order
  const int M2 = 2 * m;
  const int ti = threadIdx.x / TI, tj = threadIdx.x % TI;
  const int i0 = by * TI, j0 = bx * TI;                        // i0: LOCAL row of the tile
  const int li = i0 + ti, j = j0 + tj;
  const int ML = 2 * mloc;
  const int i = li < ML ? global_row(li, m, row0, mloc) : M2;
  float g = 0.f, ni = 0.f, nj = 0.f;
  for (int d0 = 0; d0 < D; d0 += DK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TI * DK; e += 256) {
      const int r = e / DK, c = e % DK;
      const int d = d0 + c;
      const int ra = (i0 + r) < ML ? global_row(i0 + r, m, row0, mloc) : M2, rb = j0 + r;
      s_a[r][c] = (ra < M2 && d < D) ? z[(int64_t)ra * ldz + d] : 0.f;
      s_b[r][c] = (rb < M2 && d < D) ? z[(int64_t)rb * ldz + d] : 0.f;
    }
    __syncthreads();
    const int dm = (D - d0) < DK ? (D - d0) : DK;
    for (int c = 0; c < dm; ++c) {
      const float a = s_a[ti][c], b = s_b[tj][c];
      g = fmaf(a, b, g);
      ni = fmaf(a, a, ni);
      nj = fmaf(b, b, nj);
    }
  }
  float kxx = 0.f, kyy = 0.f, kxy = 0.f;
  if (i < M2 && j < M2) pair_epilogue(li, i, j, m, g, ni, nj, w, neg_gamma, ns, wt, kxx, kyy, kxy);
  double dxx = wave_sum_d((double)kxx), dyy = wave_sum_d((double)kyy), dxy = wave_sum_d((double)kxy);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    double* sw = s_sum[threadIdx.x / WAVE];
    sw[0] = dxx;
    sw[1] = dyy;
    sw[2] = dxy;
  }
  __syncthreads();
  // one hardware fp64 add per workgroup and sum (atomicAdd(double*) is a compare-and-swap loop without
  // -munsafe-fp-atomics: hundreds of workgroups on three addresses)
  if (threadIdx.x < 3)
    unsafeAtomicAdd(&sums[threadIdx.x], ((s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + s_sum[2][threadIdx.x]) + s_sum[3][threadIdx.x]);
}


// Few rows, many features (2m <= 128, e.g. the node features: 64 x 4106): 4x4 pair tiles, the 16
// lanes of a pair split D (stride 16, straight from global / L2) and their partial chains are
// combined by a fixed xor tree -- for i == j the three partial chains are identical lane by lane,
// so e_ii is still exactly 0.  (2m/4)^2 workgroups instead of (2m/16)^2.
__device__ __forceinline__ void mmd_rbf_small_body(const int bx, const int by, const float* __restrict__ z, int64_t ldz,
                                                   int m, int D, const float* __restrict__ w,
                                                   const float* __restrict__ neg_gamma, int ns, int row0,
                                                   int mloc, double* __restrict__ sums,
                                                   float* __restrict__ wt) {
  __shared__ double s_sum[256 / WAVE][3];          // per-wave partial sums, combined in wave order
  const int M2 = 2 * m;
  const int l16 = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int li = by * 4 + (pr >> 2), j = bx * 4 + (pr & 3);
  const int i = li < 2 * mloc ? global_row(li, m, row0, mloc) : M2;
  __syncthreads();
  float g = 0.f, ni = 0.f, nj = 0.f;
  if (i < M2 && j < M2) {
    const float* zi = z + (int64_t)i * ldz;
    const float* zj = z + (int64_t)j * ldz;
    // 16 elements of both rows in flight per trip (with 4 the loop ran at one L2 latency per 4 elements: 13 us for a
    // 64 x 4106 operand); the three fma chains keep the plain loop's order
    int d = l16;
    for (; d + 15 * 16 < D; d += 16 * 16) {
      float av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { av[u] = zi[d + 16 * u]; bv[u] = zj[d + 16 * u]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        g = fmaf(av[u], bv[u], g);
        ni = fmaf(av[u], av[u], ni);
        nj = fmaf(bv[u], bv[u], nj);
      }
    }
    for (; d < D; d += 16) {
      const float a = zi[d], b = zj[d];
      g = fmaf(a, b, g);
      ni = fmaf(a, a, ni);
      nj = fmaf(b, b, nj);
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {
    g += __shfl_xor(g, o);
    ni += __shfl_xor(ni, o);
    nj += __shfl_xor(nj, o);
  }
  float kxx = 0.f, kyy = 0.f, kxy = 0.f;
  if (l16 == 0 && i < M2 && j < M2) pair_epilogue(li, i, j, m, g, ni, nj, w, neg_gamma, ns, wt, kxx, kyy, kxy);
  double dxx = wave_sum_d((double)kxx), dyy = wave_sum_d((double)kyy), dxy = wave_sum_d((double)kxy);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    double* sw = s_sum[threadIdx.x / WAVE];
    sw[0] = dxx;
    sw[1] = dyy;
    sw[2] = dxy;
  }
  __syncthreads();
  // one hardware fp64 add per workgroup and sum (atomicAdd(double*) is a compare-and-swap loop without
  // -munsafe-fp-atomics: hundreds of workgroups on three addresses)
  if (threadIdx.x < 3)
    unsafeAtomicAdd(&sums[threadIdx.x], ((s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + s_sum[2][threadIdx.x]) + s_sum[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void mmd_rbf_kernel(const float* __restrict__ z, int64_t ldz, int m, int D,
                                                      const float* __restrict__ w, const float* __restrict__ neg_gamma,
                                                      int ns, int row0, int mloc, double* __restrict__ sums,
                                                      float* __restrict__ wt) {
  mmd_rbf_body(blockIdx.x, blockIdx.y, z, ldz, m, D, w, neg_gamma, ns, row0, mloc, sums, wt);
}
__global__ __launch_bounds__(256) void mmd_rbf_small_kernel(const float* __restrict__ z, int64_t ldz, int m, int D,
                                                            const float* __restrict__ w, const float* __restrict__ neg_gamma,
                                                            int ns, int row0, int mloc, double* __restrict__ sums,
                                                            float* __restrict__ wt) {
  mmd_rbf_small_body(blockIdx.x, blockIdx.y, z, ldz, m, D, w, neg_gamma, ns, row0, mloc, sums, wt);
}

// ---- several soft-MMD terms of one batch in one launch per stage (sug_soft_mmd_multi_*): a SUG step evaluates three
// (node features, two heads' semantic features: train_dg_single_gpu.py:300-322) on the same m samples and labels; each
// stage is a latency-bound launch of a few workgroups, so the three of a stage run side by side as blockIdx.z.
#define SUG_MMD_MULTI 4
struct MmdTerm {
  const float* fs; const float* ft; int64_t lds, ldt;      // features [m, D] of the two domains
  float* z; int D;                                           // assembled operand [2m, D + ncls] (row stride D + ncls)
  float scale;                                               // label weight
  const float* w; float* wt;                                 // SDA weights [m] or null; derivative weights [2m, 2m] or null
  const float* gscale; float* dz;                            // backward: upstream gradient (device scalar), dZ [2m, D + ncls]
  int small;
};
struct MmdMulti {
  MmdTerm t[SUG_MMD_MULTI];
};

__global__ __launch_bounds__(256) void mmd_assemble_multi_kernel(MmdMulti a, int n, const int64_t* __restrict__ ls,
                                                                 const int64_t* __restrict__ lt, int m, int ncls,
                                                                 double* __restrict__ sums) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < 3 * n) sums[threadIdx.x] = 0.0;   // the rbf launch follows in stream order
  const MmdTerm& t = a.t[blockIdx.y];
  const int row = blockIdx.x, D = t.D;
  const float* src = row < m ? t.fs + (int64_t)row * t.lds : t.ft + (int64_t)(row - m) * t.ldt;
  const int64_t lab = row < m ? ls[row] : lt[row - m];
  float* dst = t.z + (int64_t)row * (D + ncls);
  for (int c = threadIdx.x; c < D; c += 256) dst[c] = src[c];
  for (int c = threadIdx.x; c < ncls; c += 256) dst[D + c] = (c == lab) ? t.scale : 0.f;
}

__global__ __launch_bounds__(256) void mmd_rbf_multi_kernel(MmdMulti a, int m, int ncls, const float* __restrict__ neg_gamma, int ns,
                                                            double* __restrict__ sums) {
  const MmdTerm& t = a.t[blockIdx.z];
  const int Dz = t.D + ncls;
  if (t.small) {
    mmd_rbf_small_body(blockIdx.x, blockIdx.y, t.z, Dz, m, Dz, t.w, neg_gamma, ns, 0, m, sums + 3 * blockIdx.z, t.wt);
  } else {
    const int nb = (2 * m + TI - 1) / TI;                    // the launch is sized for the 4 x 4 tiles of the small form
    if ((int)blockIdx.x >= nb || (int)blockIdx.y >= nb) return;
    mmd_rbf_body(blockIdx.x, blockIdx.y, t.z, Dz, m, Dz, t.w, neg_gamma, ns, 0, m, sums + 3 * blockIdx.z, t.wt);
  }
}

__global__ void mmd_value_multi_kernel(const double* __restrict__ sums, int n, double mm, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (float)((sums[3 * i] + sums[3 * i + 1] - 2.0 * sums[3 * i + 2]) / mm);
}

// part[b][blockIdx.x] = sum over this workgroup's points i of min_j |a_i - b_j|^2   (one direction; called twice, then
// chamfer_fold_kernel adds a cloud's partial sums in block order: no float atomics, the result is reproducible)
__global__ __launch_bounds__(256) void chamfer_dir_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ bpts, int N, int M,
                                                          float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float s_p[];  // M x (x, y, z, pad): one b128 broadcast read per candidate
  const int b = blockIdx.y;                                    // (three b32 reads per candidate made this LDS-issue bound)
  const float* bb = bpts + (int64_t)b * M * 3;
  for (int e = threadIdx.x; e < M * 3; e += 256) s_p[(e / 3) * 4 + e % 3] = bb[e];
  __syncthreads();
  // four lanes per point, each over a quarter of the candidates (j = cl, cl+4, ...): 4 waves per SIMD instead of one
  // hide the LDS latency that a single wave per SIMD left exposed (34 -> ~10 us at 64 clouds x 1024 points)
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t >> 2, cl = t & 3;
  float best = INFINITY;
  if (i < N) {
    const float* p = a + ((int64_t)b * N + i) * 3;
    const float x = p[0], y = p[1], zc = p[2];
    float b0 = INFINITY, b1 = INFINITY;
    int j = cl;
    for (; j + 4 < M; j += 8) {
      const float4 c0 = *reinterpret_cast<const float4*>(s_p + 4 * j), c1 = *reinterpret_cast<const float4*>(s_p + 4 * j + 16);
      const float d0 = sq3(x - c0.x, y - c0.y, zc - c0.z), d1 = sq3(x - c1.x, y - c1.y, zc - c1.z);
      b0 = d0 < b0 ? d0 : b0;
      b1 = d1 < b1 ? d1 : b1;
    }
    for (; j < M; j += 4) {
      const float4 c = *reinterpret_cast<const float4*>(s_p + 4 * j);
      const float d = sq3(x - c.x, y - c.y, zc - c.z);
      b0 = d < b0 ? d : b0;
    }
    best = b1 < b0 ? b1 : b0;
  }
  {                                             // minimum over the point's four lanes (exact)
    float o = __shfl_xor(best, 1);
    best = o < best ? o : best;
    o = __shfl_xor(best, 2);
    best = o < best ? o : best;
  }
  best = (cl == 0 && i < N) ? best : 0.f;
  __shared__ float s_w[256 / WAVE];
  const float s = wave_sum_f(best);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)b * gridDim.x + blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

// out[b] = (sum of the nbn partial sums of direction a -> b) / N + (sum of the nbm partial sums of b -> a) / M, block order
__global__ __launch_bounds__(256) void chamfer_fold_kernel(const float* __restrict__ pn, int nbn, const float* __restrict__ pm,
                                                           int nbm, int B, float inv_n, float inv_m, float* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float sn = 0.f, sm = 0.f;
  for (int i = 0; i < nbn; ++i) sn += pn[(int64_t)b * nbn + i];
  for (int i = 0; i < nbm; ++i) sm += pm[(int64_t)b * nbm + i];
  out[b] = sn * inv_n + sm * inv_m;
}

// chamfer_fold_kernel + distance2weights (model/mmd.py:178-202) for a batch that fits one workgroup's loop: the per-pair
// distances, their fp64 sums in thread / wave order, then the weights -- the mean / reciprocal / cast / multiply launches of
// the torch formulation in the fold's own launch.  method: 1 naive_inverse, 2 exp_inverse, 3 mean2one (1/mean truncated).
__global__ __launch_bounds__(256) void chamfer_fold_weights_kernel(const float* __restrict__ pn, int nbn,
                                                                   const float* __restrict__ pm, int nbm, int B, float inv_n,
                                                                   float inv_m, int method, float* __restrict__ out) {
  __shared__ double s_red[4][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double dsum = 0.0, wsum = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    float sn = 0.f, sm = 0.f;
    for (int i = 0; i < nbn; ++i) sn += pn[(int64_t)b * nbn + i];
    for (int i = 0; i < nbm; ++i) sm += pm[(int64_t)b * nbm + i];
    const float dist = sn * inv_n + sm * inv_m;
    out[b] = dist;
    dsum += (double)dist;
    wsum += method == 1 ? (double)(1.f / (dist + 1e-8f)) : (method == 2 ? (double)expf(-dist) : 0.0);
  }
  dsum = wave_sum_d(dsum);
  wsum = wave_sum_d(wsum);
  if (lane == 0) { s_red[wv][0] = dsum; s_red[wv][1] = wsum; }
  __syncthreads();
  const double td = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
  const double tw = ((s_red[0][1] + s_red[1][1]) + s_red[2][1]) + s_red[3][1];
  const float mean = (float)(td / (double)B);
  const float scale = (float)(int)(1.f / mean);           // .type(torch.int): truncation
  for (int b = threadIdx.x; b < B; b += 256) {            // every thread re-reads its own stores
    const float dist = out[b];
    out[b] = method == 1 ? (1.f / (dist + 1e-8f)) / (float)tw : (method == 2 ? expf(-dist) / (float)tw : dist * scale);
  }
}


// dZ[li,:] = scale * 2 * (rowsum(wt)[li] * Z[i,:] - sum_j wt[li,j] Z[j,:]) for the local rows li (global
// row i, see global_row) over all 2m columns j: workgroup = (64 columns of D) x 4 row groups; the Z
// column panel goes through LDS in chunks of JT rows, wt rows are broadcast reads.
constexpr int JT = 128;
__device__ __forceinline__ void mmd_bwd_body(const int bx, const int by, const float* __restrict__ z, int64_t ldz,
                                             const float* __restrict__ wt, int m, int D, int row0, int mloc,
                                             const float* __restrict__ gscale, float gmul,
                                             float* __restrict__ dz, int64_t lddz) {
  __shared__ float s_z[JT * 64];
  __shared__ __attribute__((aligned(16))) float s_w[JT * 32];       // wt panel, [j][local row]: a row group's 8 weights = 2 x b128
  const int M2 = 2 * m, ML = 2 * mloc;
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int d = bx * 64 + c;
  const int li0 = by * 32;                          // 32 local rows per workgroup, 8 per row group
  float rs[8], acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { rs[u] = 0.f; acc[u] = 0.f; }
  for (int j0 = 0; j0 < M2; j0 += JT) {
    __syncthreads();
    for (int j = rg; j < JT; j += 4) s_z[j * 64 + c] = (d < D && j0 + j < M2) ? z[(int64_t)(j0 + j) * ldz + d] : 0.f;
    // the weights of this panel: rows past ML / columns past M2 as zeros (they add +0 to sums that start at +0)
    for (int e = threadIdx.x; e < JT * 32; e += 256) {
      const int r = e / JT, j = e % JT;                       // consecutive threads along j: coalesced reads of wt
      const int li = li0 + r;
      s_w[j * 32 + r] = (li < ML && j0 + j < M2) ? wt[(int64_t)li * M2 + j0 + j] : 0.f;
    }
    __syncthreads();
    const int jn = (M2 - j0) < JT ? (M2 - j0) : JT;
    // eight independent (row sum, dot) chains per thread, j ascending as before: the serial per-row loops (a global
    // broadcast load + a dependent fma per step) ran at load latency, 17 us for a 64 x 4106 operand
    for (int j = 0; j < jn; ++j) {
      const float zc = s_z[j * 64 + c];
      const float4 w0 = *reinterpret_cast<const float4*>(s_w + j * 32 + rg * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(s_w + j * 32 + rg * 8 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        rs[u] += wv[u];
        acc[u] = fmaf(wv[u], zc, acc[u]);
      }
    }
  }
  if (d >= D) return;
  const float g2 = 2.f * gscale[0] * gmul;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int li = li0 + rg * 8 + u;
    if (li >= ML) continue;
    const int i = global_row(li, m, row0, mloc);
    dz[(int64_t)li * lddz + d] = g2 * (rs[u] * z[(int64_t)i * ldz + d] - acc[u]);
  }
}

__global__ __launch_bounds__(256) void mmd_bwd_kernel(const float* __restrict__ z, int64_t ldz,
                                                      const float* __restrict__ wt, int m, int D, int row0, int mloc,
                                                      const float* __restrict__ gscale, float gmul,
                                                      float* __restrict__ dz, int64_t lddz) {
  mmd_bwd_body(blockIdx.x, blockIdx.y, z, ldz, wt, m, D, row0, mloc, gscale, gmul, dz, lddz);
}
__global__ __launch_bounds__(256) void mmd_bwd_multi_kernel(MmdMulti a, int m, int ncls) {
  const MmdTerm& t = a.t[blockIdx.z];
  if ((int)blockIdx.x * 64 >= t.D) return;                   // the launch is sized for the widest term
  // the feature columns only, dense [2m, D]: the label columns of Z are constants, and the two domains' gradients leave as
  // adjacent contiguous row blocks (no strided-slice copies downstream)
  mmd_bwd_body(blockIdx.x, blockIdx.y, t.z, t.D + ncls, t.wt, m, t.D, 0, m, t.gscale, 1.0f, t.dz, t.D);
}

// SDA sample weights from class probabilities (prob_weights_soft + distance2weights,
// model/mmd.py:134-148, :178-202): one workgroup, thread per sample row.
//   a = [softmax(pred_s) | onehot(label_s) * lw] + 1e-8, normalised by its global sum (b likewise);
//   dist_i = sum_c 0.5*kl(a,b) + 0.5*kl(b,a), kl(x,y) = x log(x/y) - x + y;
//   method 0 none, 1 naive_inverse, 2 exp_inverse, 3 mean2one (1/mean truncated to an integer).
__device__ __forceinline__ void sda_prob_weights_body(const float* __restrict__ ps, int64_t lds,
                                                      const float* __restrict__ pt, int64_t ldt,
                                                      const int64_t* __restrict__ ls,
                                                      const int64_t* __restrict__ lt, int m, float lw,
                                                      int method, float* __restrict__ out) {
  constexpr int NC = 10;
  __shared__ double s_red[4][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto block_sum2 = [&](double a, double b, double& ra, double& rb) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    __syncthreads();
    if (lane == 0) { s_red[wv][0] = a; s_red[wv][1] = b; }
    __syncthreads();
    ra = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
    rb = ((s_red[0][1] + s_red[1][1]) + s_red[2][1]) + s_red[3][1];
  };
  auto row_probs = [&](const float* p, float* q) {        // fp32 softmax as torch: exp(x - max) / sum
    float mx = p[0];
    for (int c = 1; c < NC; ++c) mx = fmaxf(mx, p[c]);
    float sum = 0.f;
    for (int c = 0; c < NC; ++c) { q[c] = expf(p[c] - mx); sum += q[c]; }
    for (int c = 0; c < NC; ++c) q[c] = q[c] / sum;
  };
  // pass 1: global sums of (a + eps), (b + eps)
  double sa = 0.0, sb = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) {
    float q[NC];
    row_probs(ps + (int64_t)i * lds, q);
    for (int c = 0; c < NC; ++c) sa += (double)(q[c] + 1e-8f) + (double)(((int)ls[i] == c ? lw : 0.f) + 1e-8f);
    row_probs(pt + (int64_t)i * ldt, q);
    for (int c = 0; c < NC; ++c) sb += (double)(q[c] + 1e-8f) + (double)(((int)lt[i] == c ? lw : 0.f) + 1e-8f);
  }
  double ta, tb;
  block_sum2(sa, sb, ta, tb);
  const float fa = (float)ta, fb = (float)tb;
  // pass 2: distances (kept in out[]), and their sum / the sum of the inverse weights
  double dsum = 0.0, wsum = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) {
    float qa[NC], qb[NC];
    row_probs(ps + (int64_t)i * lds, qa);
    row_probs(pt + (int64_t)i * ldt, qb);
    float dist = 0.f;
    for (int c = 0; c < 2 * NC; ++c) {
      const float av = c < NC ? qa[c] : ((int)ls[i] == c - NC ? lw : 0.f);
      const float bv = c < NC ? qb[c] : ((int)lt[i] == c - NC ? lw : 0.f);
      const float x = (av + 1e-8f) / fa, y = (bv + 1e-8f) / fb;
      const float k1 = x * logf(x / y) - x + y, k2 = y * logf(y / x) - y + x;
      dist += k1 * 0.5f + k2 * 0.5f;
    }
    out[i] = dist;
    dsum += (double)dist;
    wsum += method == 1 ? (double)(1.f / (dist + 1e-8f)) : (method == 2 ? (double)expf(-dist) : 0.0);
  }
  double td, tw;
  block_sum2(dsum, wsum, td, tw);
  if (method == 0) return;
  const float mean = (float)(td / (double)m);
  const float scale = (float)(int)(1.f / mean);           // .type(torch.int): truncation
  for (int i = threadIdx.x; i < m; i += 256) {
    const float dist = out[i];
    out[i] = method == 1 ? (1.f / (dist + 1e-8f)) / (float)tw : (method == 2 ? expf(-dist) / (float)tw : dist * scale);
  }
}

__global__ __launch_bounds__(256) void sda_prob_weights_kernel(const float* __restrict__ ps, int64_t lds,
                                                               const float* __restrict__ pt, int64_t ldt,
                                                               const int64_t* __restrict__ ls,
                                                               const int64_t* __restrict__ lt, int m, float lw,
                                                               int method, float* __restrict__ out) {
  sda_prob_weights_body(ps, lds, pt, ldt, ls, lt, m, lw, method, out);
}

// the weights of several heads' logits on one batch (same labels): one workgroup per head
struct SdaMulti {
  const float* ps[SUG_MMD_MULTI];
  const float* pt[SUG_MMD_MULTI];
  int64_t lds[SUG_MMD_MULTI], ldt[SUG_MMD_MULTI];
  float* out[SUG_MMD_MULTI];
};
__global__ __launch_bounds__(256) void sda_prob_weights_multi_kernel(SdaMulti a, const int64_t* __restrict__ ls,
                                                                     const int64_t* __restrict__ lt, int m, float lw, int method) {
  const int t = blockIdx.x;
  sda_prob_weights_body(a.ps[t], a.lds[t], a.pt[t], a.ldt[t], ls, lt, m, lw, method, a.out[t]);
}

}  // namespace

extern "C" int sug_mmd_rbf_rows(const float* z, int64_t ldz, int m, int D, const float* w, const float* neg_gamma,
                                int nsigma, int row0, int mloc, double* sums, float* wt, void* stream) {
  SUG_REQUIRE(z && neg_gamma && sums, "sug_mmd_rbf: null pointer");
  SUG_REQUIRE(m > 0 && D > 0 && ldz >= D, "sug_mmd_rbf: bad shape m=%d D=%d", m, D);
  SUG_REQUIRE(nsigma >= 1 && nsigma <= 8, "sug_mmd_rbf: nsigma=%d", nsigma);
  SUG_REQUIRE(row0 >= 0 && mloc > 0 && row0 + mloc <= m, "sug_mmd_rbf: row block [%d, %d) outside 0..%d", row0, row0 + mloc, m);
  if (2 * m <= 128 && D >= 256) {      // few rows, long rows: split D across lanes for parallelism
    hipLaunchKernelGGL(mmd_rbf_small_kernel, dim3(sug_divup(2 * m, 4), sug_divup(2 * mloc, 4)), dim3(256), 0,
                       (hipStream_t)stream, z, ldz, m, D, w, neg_gamma, nsigma, row0, mloc, sums, wt);
  } else {
    hipLaunchKernelGGL(mmd_rbf_kernel, dim3(sug_divup(2 * m, TI), sug_divup(2 * mloc, TI)), dim3(256), 0,
                       (hipStream_t)stream, z, ldz, m, D, w, neg_gamma, nsigma, row0, mloc, sums, wt);
  }
  SUG_LAUNCH_CHECK("sug_mmd_rbf");
  return SUG_OK;
}

extern "C" int sug_mmd_rbf(const float* z, int64_t ldz, int m, int D, const float* w,
                           const float* neg_gamma, int nsigma, double* sums, float* wt, void* stream) {
  return sug_mmd_rbf_rows(z, ldz, m, D, w, neg_gamma, nsigma, 0, m, sums, wt, stream);
}

namespace {
__global__ void mmd_value_kernel(const double* __restrict__ sums, double mm, float* __restrict__ out) {
  if (threadIdx.x == 0) out[0] = (float)((sums[0] + sums[1] - 2.0 * sums[2]) / mm);
}
__global__ void mmd_zero_sums_kernel(double* __restrict__ sums) {
  if (threadIdx.x < 3) sums[threadIdx.x] = 0.0;
}
}  // namespace

extern "C" int sug_mmd_rbf_value(const float* z, int64_t ldz, int m, int D, const float* w, const float* neg_gamma,
                                 int nsigma, double* sums, float* wt, float* value, void* stream) {
  SUG_REQUIRE(sums && value, "sug_mmd_rbf_value: null pointer");
  hipStream_t st = (hipStream_t)stream;
  // a kernel, not hipMemsetAsync: as a memset node of a captured step graph the clear was not reliably ordered in
  // front of the accumulating kernel on ROCm 7 (after a few hundred replays the reported MMD values turned into
  // constants of the size of whatever had occupied the buffer, for several replays at a time; training itself,
  // which does not read the sums, was unaffected)
  hipLaunchKernelGGL(mmd_zero_sums_kernel, dim3(1), dim3(64), 0, st, sums);
  const int rc = sug_mmd_rbf_rows(z, ldz, m, D, w, neg_gamma, nsigma, 0, m, sums, wt, stream);
  if (rc != SUG_OK) return rc;
  hipLaunchKernelGGL(mmd_value_kernel, dim3(1), dim3(64), 0, st, sums, (double)m * (double)m, value);
  SUG_LAUNCH_CHECK("sug_mmd_rbf_value");
  return SUG_OK;
}

extern "C" int64_t sug_chamfer_workspace(int B, int N, int M) {
  return (int64_t)B * (sug_divup(4 * (int64_t)N, 256) + sug_divup(4 * (int64_t)M, 256));
}

extern "C" int sug_chamfer(const float* a, const float* b, int B, int N, int M, float* out, float* ws,
                           void* stream) {
  SUG_REQUIRE(a && b && out && ws, "sug_chamfer: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && M > 0 && N <= 5000 && M <= 5000 && B <= 65535, "sug_chamfer: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int nbn = sug_divup(4 * N, 256), nbm = sug_divup(4 * M, 256);
  float* pm = ws + (int64_t)B * nbn;
  hipLaunchKernelGGL(chamfer_dir_kernel, dim3(nbn, B), dim3(256), (size_t)M * 4 * sizeof(float), st, a, b, N, M, ws);
  hipLaunchKernelGGL(chamfer_dir_kernel, dim3(nbm, B), dim3(256), (size_t)N * 4 * sizeof(float), st, b, a, M, N, pm);
  hipLaunchKernelGGL(chamfer_fold_kernel, dim3(sug_divup(B, 256)), dim3(256), 0, st, ws, nbn, pm, nbm, B, 1.0f / (float)N,
                     1.0f / (float)M, out);
  SUG_LAUNCH_CHECK("sug_chamfer");
  return SUG_OK;
}

extern "C" int sug_mmd_rbf_rows_bwd(const float* z, int64_t ldz, const float* wt, int m, int D, int row0, int mloc,
                                    const float* gscale, float gmul, float* dz, int64_t lddz, void* stream) {
  SUG_REQUIRE(z && wt && gscale && dz, "sug_mmd_rbf_bwd: null pointer");
  SUG_REQUIRE(m > 0 && D > 0 && ldz >= D && lddz >= D, "sug_mmd_rbf_bwd: bad shape");
  SUG_REQUIRE(row0 >= 0 && mloc > 0 && row0 + mloc <= m, "sug_mmd_rbf_bwd: row block outside 0..%d", m);
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(sug_divup(D, 64), sug_divup(2 * mloc, 32)), dim3(256), 0, (hipStream_t)stream, z,
                     ldz, wt, m, D, row0, mloc, gscale, gmul, dz, lddz);
  SUG_LAUNCH_CHECK("sug_mmd_rbf_bwd");
  return SUG_OK;
}

extern "C" int sug_mmd_rbf_bwd(const float* z, int64_t ldz, const float* wt, int m, int D, const float* gscale,
                               float* dz, int64_t lddz, void* stream) {
  return sug_mmd_rbf_rows_bwd(z, ldz, wt, m, D, 0, m, gscale, 1.0f, dz, lddz, stream);
}

extern "C" int sug_sda_prob_weights(const float* pred_s, int64_t lds, const float* pred_t, int64_t ldt,
                                    const int64_t* label_s, const int64_t* label_t, int m, int num_class,
                                    float label_weight, int method, float* weights, void* stream) {
  SUG_REQUIRE(pred_s && pred_t && label_s && label_t && weights, "sug_sda_prob_weights: null pointer");
  SUG_REQUIRE(m > 0 && num_class == 10 && lds >= 10 && ldt >= 10, "sug_sda_prob_weights: bad shape (10 classes)");
  SUG_REQUIRE(method >= 0 && method <= 3, "sug_sda_prob_weights: unknown weighting method");
  hipLaunchKernelGGL(sda_prob_weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred_s, lds, pred_t, ldt,
                     label_s, label_t, m, label_weight, method, weights);
  SUG_LAUNCH_CHECK("sug_sda_prob_weights");
  return SUG_OK;
}

// ---- label-augmented MMD operand and the EdgeConv weight split: small assembly steps of a training step that cost
// four to six tiny torch launches each (cat / scatter / mul / sub) in the reference's formulation
namespace {

__global__ __launch_bounds__(256) void mmd_assemble_kernel(const float* __restrict__ fs, int64_t lds,
                                                           const float* __restrict__ ft, int64_t ldt,
                                                           const int64_t* __restrict__ ls, const int64_t* __restrict__ lt,
                                                           int m, int D, int ncls, float scale, float* __restrict__ z) {
  const int row = blockIdx.x;
  const float* src = row < m ? fs + (int64_t)row * lds : ft + (int64_t)(row - m) * ldt;
  const int64_t lab = row < m ? ls[row] : lt[row - m];
  float* dst = z + (int64_t)row * (D + ncls);
  for (int c = threadIdx.x; c < D; c += 256) dst[c] = src[c];
  for (int c = threadIdx.x; c < ncls; c += 256) dst[D + c] = (c == lab) ? scale : 0.f;
}

// forward: out [2Co, C] = [W[:, :C] ; W[:, C:] - W[:, :C]];  backward: dW [Co, 2C] = [g[:Co] - g[Co:] | g[Co:]]
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ in, int Co, int C, int backward,
                                                           float* __restrict__ out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * Co * C) return;
  if (!backward) {
    const int r = e / C, c = e - r * C;
    out[e] = r < Co ? in[(int64_t)r * 2 * C + c] : in[(int64_t)(r - Co) * 2 * C + C + c] - in[(int64_t)(r - Co) * 2 * C + c];
  } else {
    const int r = e / (2 * C), c = e - r * 2 * C;
    out[e] = c < C ? in[(int64_t)r * C + c] - in[(int64_t)(Co + r) * C + c] : in[(int64_t)(Co + r) * C + c - C];
  }
}

// the same for up to SUG_SPLIT_MULTI weights in one launch (the four EdgeConv layers of the DGCNN encoder)
#define SUG_SPLIT_MULTI 8
struct SplitMultiArgs {
  const float* in[SUG_SPLIT_MULTI];
  float* out[SUG_SPLIT_MULTI];
  int Co[SUG_SPLIT_MULTI], C[SUG_SPLIT_MULTI], first[SUG_SPLIT_MULTI + 1];
};
__global__ __launch_bounds__(256) void split_weight_multi_kernel(SplitMultiArgs a, int n, int backward) {
  int t = 0;
  while (t + 1 < n && (int)blockIdx.x >= a.first[t + 1]) ++t;
  const float* __restrict__ in = a.in[t];
  float* __restrict__ out = a.out[t];
  const int Co = a.Co[t], C = a.C[t];
  const int e = (blockIdx.x - a.first[t]) * 256 + threadIdx.x;
  if (e >= 2 * Co * C) return;
  if (!backward) {
    const int r = e / C, c = e - r * C;
    out[e] = r < Co ? in[(int64_t)r * 2 * C + c] : in[(int64_t)(r - Co) * 2 * C + C + c] - in[(int64_t)(r - Co) * 2 * C + c];
  } else {
    const int r = e / (2 * C), c = e - r * 2 * C;
    out[e] = c < C ? in[(int64_t)r * C + c] - in[(int64_t)(Co + r) * C + c] : in[(int64_t)(Co + r) * C + c - C];
  }
}

}  // namespace

extern "C" int sug_mmd_assemble(const float* feat_s, int64_t lds, const float* feat_t, int64_t ldt, const int64_t* label_s,
                                const int64_t* label_t, int m, int D, int num_class, float label_scale, float* z,
                                void* stream) {
  SUG_REQUIRE(feat_s && feat_t && label_s && label_t && z, "sug_mmd_assemble: null pointer");
  SUG_REQUIRE(m > 0 && D > 0 && num_class > 0 && lds >= D && ldt >= D, "sug_mmd_assemble: bad shape");
  hipLaunchKernelGGL(mmd_assemble_kernel, dim3(2 * m), dim3(256), 0, (hipStream_t)stream, feat_s, lds, feat_t, ldt, label_s,
                     label_t, m, D, num_class, label_scale, z);
  SUG_LAUNCH_CHECK("sug_mmd_assemble");
  return SUG_OK;
}

extern "C" int sug_edge_weight_split(const float* in, int Co, int C, int backward, float* out, void* stream) {
  SUG_REQUIRE(in && out, "sug_edge_weight_split: null pointer");
  SUG_REQUIRE(Co > 0 && C > 0 && (int64_t)Co * C < (1 << 28), "sug_edge_weight_split: bad shape");
  hipLaunchKernelGGL(split_weight_kernel, dim3(sug_divup(2 * Co * C, 256)), dim3(256), 0, (hipStream_t)stream, in, Co, C,
                     backward, out);
  SUG_LAUNCH_CHECK("sug_edge_weight_split");
  return SUG_OK;
}

extern "C" int sug_edge_weight_split_multi(const void* const* in_host, const int32_t* Co_host, const int32_t* C_host, int n,
                                           int backward, void* const* out_host, void* stream) {
  SUG_REQUIRE(in_host && Co_host && C_host && out_host, "sug_edge_weight_split_multi: null pointer");
  SUG_REQUIRE(n > 0 && n <= SUG_SPLIT_MULTI, "sug_edge_weight_split_multi: %d weights (1..%d)", n, SUG_SPLIT_MULTI);
  SplitMultiArgs a;
  int m = 0;
  a.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    if (!in_host[i] || !out_host[i]) continue;                       // a weight without a gradient: skipped
    SUG_REQUIRE(Co_host[i] > 0 && C_host[i] > 0 && (int64_t)Co_host[i] * C_host[i] < (1 << 26),
                "sug_edge_weight_split_multi: bad shape of weight %d", i);
    a.in[m] = (const float*)in_host[i];
    a.out[m] = (float*)out_host[i];
    a.Co[m] = Co_host[i];
    a.C[m] = C_host[i];
    a.first[m + 1] = a.first[m] + sug_divup(2 * Co_host[i] * C_host[i], 256);
    ++m;
  }
  if (m == 0) return SUG_OK;
  for (int i = m; i < SUG_SPLIT_MULTI; ++i) {
    a.in[i] = nullptr;
    a.out[i] = nullptr;
    a.Co[i] = a.C[i] = 0;
    a.first[i + 1] = a.first[m];
  }
  hipLaunchKernelGGL(split_weight_multi_kernel, dim3(a.first[m]), dim3(256), 0, (hipStream_t)stream, a, m, backward);
  SUG_LAUNCH_CHECK("sug_edge_weight_split_multi");
  return SUG_OK;
}

extern "C" int sug_soft_mmd_multi_fwd(int n, const void* const* feat_s, const int64_t* lds, const void* const* feat_t,
                                      const int64_t* ldt, const int32_t* D, const float* label_scale, const int64_t* label_s,
                                      const int64_t* label_t, int m, int num_class, const void* const* w,
                                      const float* neg_gamma, int nsigma, void* const* z, void* const* wt, double* sums,
                                      float* values, void* stream) {
  SUG_REQUIRE(feat_s && lds && feat_t && ldt && D && label_scale && label_s && label_t && w && neg_gamma && z && wt && sums &&
              values, "sug_soft_mmd_multi_fwd: null pointer");
  SUG_REQUIRE(n >= 1 && n <= SUG_MMD_MULTI, "sug_soft_mmd_multi_fwd: %d terms (1..%d)", n, SUG_MMD_MULTI);
  SUG_REQUIRE(m > 0 && m <= 16384 && num_class > 0 && num_class <= 256, "sug_soft_mmd_multi_fwd: bad shape m=%d classes=%d", m, num_class);
  SUG_REQUIRE(nsigma >= 1 && nsigma <= 8, "sug_soft_mmd_multi_fwd: nsigma=%d", nsigma);
  hipStream_t st = (hipStream_t)stream;
  MmdMulti a;
  bool any_small = false, any_big = false;
  for (int i = 0; i < SUG_MMD_MULTI; ++i) {
    const int k = i < n ? i : 0;
    SUG_REQUIRE(feat_s[k] && feat_t[k] && z[k], "sug_soft_mmd_multi_fwd: null operand of term %d", k);
    SUG_REQUIRE(D[k] > 0 && lds[k] >= D[k] && ldt[k] >= D[k], "sug_soft_mmd_multi_fwd: bad shape of term %d", k);
    MmdTerm& t = a.t[i];
    t.fs = (const float*)feat_s[k];
    t.ft = (const float*)feat_t[k];
    t.lds = lds[k];
    t.ldt = ldt[k];
    t.z = (float*)z[k];
    t.D = D[k];
    t.scale = label_scale[k];
    t.w = (const float*)w[k];
    t.wt = (float*)wt[k];
    t.gscale = nullptr;
    t.dz = nullptr;
    t.small = (2 * m <= 128 && D[k] + num_class >= 256) ? 1 : 0;         // sug_mmd_rbf_rows' own choice
    if (i < n) (t.small ? any_small : any_big) = true;
  }
  hipLaunchKernelGGL(mmd_assemble_multi_kernel, dim3(2 * m, n), dim3(256), 0, st, a, n, label_s, label_t, m, num_class, sums);
  const int nb = any_small ? sug_divup(2 * m, 4) : sug_divup(2 * m, TI);
  (void)any_big;
  hipLaunchKernelGGL(mmd_rbf_multi_kernel, dim3(nb, nb, n), dim3(256), 0, st, a, m, num_class, neg_gamma, nsigma, sums);
  hipLaunchKernelGGL(mmd_value_multi_kernel, dim3(1), dim3(64), 0, st, sums, n, (double)m * (double)m, values);
  SUG_LAUNCH_CHECK("sug_soft_mmd_multi_fwd");
  return SUG_OK;
}

extern "C" int sug_soft_mmd_multi_bwd(int n, const void* const* z, const int32_t* D, const void* const* wt,
                                      const void* const* gscale, int m, int num_class, void* const* dz, void* stream) {
  SUG_REQUIRE(z && D && wt && gscale && dz, "sug_soft_mmd_multi_bwd: null pointer");
  SUG_REQUIRE(n >= 1 && n <= SUG_MMD_MULTI && m > 0 && num_class > 0, "sug_soft_mmd_multi_bwd: bad arguments");
  MmdMulti a;
  int na = 0, dmax = 0;
  for (int i = 0; i < n; ++i) {
    if (!gscale[i] || !dz[i]) continue;                                   // a term without an upstream gradient: skipped
    SUG_REQUIRE(z[i] && wt[i] && D[i] > 0, "sug_soft_mmd_multi_bwd: null operand of term %d", i);
    MmdTerm& t = a.t[na++];
    t = MmdTerm{};
    t.z = (float*)z[i];
    t.D = D[i];
    t.wt = (float*)wt[i];
    t.gscale = (const float*)gscale[i];
    t.dz = (float*)dz[i];
    if (D[i] > dmax) dmax = D[i];
  }
  if (na == 0) return SUG_OK;
  for (int i = na; i < SUG_MMD_MULTI; ++i) a.t[i] = a.t[0];
  hipLaunchKernelGGL(mmd_bwd_multi_kernel, dim3(sug_divup(dmax, 64), sug_divup(2 * m, 32), na), dim3(256), 0,
                     (hipStream_t)stream, a, m, num_class);
  SUG_LAUNCH_CHECK("sug_soft_mmd_multi_bwd");
  return SUG_OK;
}

extern "C" int sug_sda_prob_weights_multi(int n, const void* const* pred_s, const int64_t* lds, const void* const* pred_t,
                                          const int64_t* ldt, const int64_t* label_s, const int64_t* label_t, int m,
                                          int num_class, float label_weight, int method, void* const* weights, void* stream) {
  SUG_REQUIRE(pred_s && lds && pred_t && ldt && label_s && label_t && weights, "sug_sda_prob_weights_multi: null pointer");
  SUG_REQUIRE(n >= 1 && n <= SUG_MMD_MULTI, "sug_sda_prob_weights_multi: %d heads (1..%d)", n, SUG_MMD_MULTI);
  SUG_REQUIRE(m > 0 && num_class == 10, "sug_sda_prob_weights_multi: bad shape (10 classes)");
  SUG_REQUIRE(method >= 0 && method <= 3, "sug_sda_prob_weights_multi: unknown weighting method");
  SdaMulti a;
  for (int i = 0; i < SUG_MMD_MULTI; ++i) {
    const int k = i < n ? i : 0;
    SUG_REQUIRE(pred_s[k] && pred_t[k] && weights[k] && lds[k] >= 10 && ldt[k] >= 10, "sug_sda_prob_weights_multi: bad operand %d", k);
    a.ps[i] = (const float*)pred_s[k];
    a.pt[i] = (const float*)pred_t[k];
    a.lds[i] = lds[k];
    a.ldt[i] = ldt[k];
    a.out[i] = (float*)weights[k];
  }
  hipLaunchKernelGGL(sda_prob_weights_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a, label_s, label_t, m,
                     label_weight, method);
  SUG_LAUNCH_CHECK("sug_sda_prob_weights_multi");
  return SUG_OK;
}

extern "C" int sug_chamfer_weights(const float* a, const float* b, int B, int N, int M, int method, float* out, float* ws,
                                   void* stream) {
  SUG_REQUIRE(a && b && out && ws, "sug_chamfer_weights: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && M > 0 && N <= 5000 && M <= 5000 && B <= 65535, "sug_chamfer_weights: bad shape");
  SUG_REQUIRE(method >= 1 && method <= 3, "sug_chamfer_weights: weighting method %d (1 naive_inverse, 2 exp_inverse, 3 mean2one)", method);
  hipStream_t st = (hipStream_t)stream;
  const int nbn = sug_divup(4 * N, 256), nbm = sug_divup(4 * M, 256);
  float* pm = ws + (int64_t)B * nbn;
  hipLaunchKernelGGL(chamfer_dir_kernel, dim3(nbn, B), dim3(256), (size_t)M * 4 * sizeof(float), st, a, b, N, M, ws);
  hipLaunchKernelGGL(chamfer_dir_kernel, dim3(nbm, B), dim3(256), (size_t)N * 4 * sizeof(float), st, b, a, M, N, pm);
  hipLaunchKernelGGL(chamfer_fold_weights_kernel, dim3(1), dim3(256), 0, st, ws, nbn, pm, nbm, B, 1.0f / (float)N,
                     1.0f / (float)M, method, out);
  SUG_LAUNCH_CHECK("sug_chamfer_weights");
  return SUG_OK;
}
