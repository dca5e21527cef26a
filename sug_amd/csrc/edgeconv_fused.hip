// EdgeConv layer with the 1x1 convolution INSIDE the gather kernel (get_graph_feature + conv_2d + max over k,
// model/model_utils.py:188-210, :8-32, model/Model.py:88-109).
//
//   y[b,n,j,c] = W.[x_j - x_n ; x_n] = P[b,idx[b,n,j],c] + Q[b,n,c],   [P | Q] = x.[W1 ; W2-W1]^T
//
// edgeconv.hip's kernels take [P|Q] from a library GEMM through HBM (written once, read once: 16*Co bytes per point)
// and re-stage a 16-channel slice of P in LDS.  Here a workgroup = (cloud, 16-channel slice) forms that slice itself:
// the cloud's x rows stream from L2 in 32-row tiles straight into the B operand of v_mfma_f32_32x32x2_f32 (the A
// operand holds the slice's 16 P rows and 16 Q rows of the weight matrix), the 32x32 result tile is
// [P slice ; Q slice]^T of 32 points: the P half goes into the LDS image the gathers read, the Q half (needed only by
// its own point) waits in the z buffer -- the location its point's result overwrites -- so [P|Q] never exists in
// HBM as a tensor.  The gather / reduce phase is edgeconv_fwd_lds_kernel's: 4 lanes per point, first maximum wins,
// sign(gamma)-folded so that max/min is a plain max, per-(cloud) BatchNorm partial rows in a fixed order.
//
// MFMA orientation (M = channels, N = points): lane l holds D[i][j] for column j = l & 31 (a point) and rows
// i = (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15.  Weight row of A-row i: P channel r' for (i >> 2) & 1 == 0,
// Q channel r' otherwise, r' = (i & 3) + 4 (i >> 3): lanes 0..31 then hold P[j][c0 .. c0+15], lanes 32..63
// Q[j][c0 .. c0+15] -- one 64-byte row each.  k order: step t takes features 2t (lanes 0-31) and 2t+1 (lanes 32-63): an
// ascending fma chain from zero, bit for bit a plain dot product's.
//
// Per cloud and layer (fp32): x once per slice through L2 (HBM: once), idx, z / arg / s1 once.  FLOPs 2*N*C*2Co on
// the fp32 matrix pipe + 6*N*k*Co VALU.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int SW = 16;         // channels per slice
constexpr int LP = SW / 4;     // lanes per point in the gather phase

// CIN: features per point as the MFMA sees them (4 = xyz padded with a zero feature, else 64 / 128); CR: real
// feature count = row length of wcat.  KK: neighbours per point.  NT: threads per workgroup.  The product form is ONE
// 512-thread workgroup per CU (P and Q images: 128 KB of LDS) with up to 256 registers per lane: with the pivot of the
// BatchNorm sums in registers the 1024-thread form (128 registers) spilled into scratch memory (PMC: +30 % HBM-side
// traffic), and 16 waves per CU had measured no faster than 8 (phases add up either way).
// QLDS: the Q slice waits in LDS next to the P image (2 x N x 64 bytes: one workgroup per CU) instead of in the z
// buffer -- parked in global memory it costs its bytes twice in HBM-side traffic (PMC: written back before z
// overwrites it, fetched again by the gather phase), which is what the fusion is there to avoid.
template <int CIN, int KK, int NT, bool QLDS>
__global__ __launch_bounds__(NT, NT / 256) void edgeconv_fused_fwd_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ wcat, const float* __restrict__ qbias,
    const int32_t* __restrict__ idx, const float* __restrict__ gamma, int B, int N, int Co, int Bg,
    float* __restrict__ z, uint8_t* __restrict__ arg, float* __restrict__ s1, float* __restrict__ pq_out,
    int64_t ldpq, float* __restrict__ ws) {
  static_assert(KK % 4 == 0, "neighbour rows are fetched as int4");
  constexpr int CR = CIN == 4 ? 3 : CIN;
  constexpr int HALF = CIN / 2;                    // MFMA k-steps per tile
  constexpr int KC = HALF < 32 ? HALF : (CIN == 128 ? 16 : 32);   // k-steps per register chunk of the x operand (two chunks live)
  constexpr int NCH = HALF / KC;
  constexpr int NW = NT / 64;                      // waves
  constexpr int PPP = NT / LP;                     // points per pass of the gather phase
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  float* s_p = s_lds;                              // [N][SW]  sign(gamma) * P slice; later the reduction scratch
  float* s_q = s_lds + (size_t)N * SW;             // [N][SW]  Q slice (QLDS)
  const int nslice = Co / SW;
  int b, sl;
  if ((B & 7) == 0) {                              // the slices of a cloud share an XCD (its L2 holds x once)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    b = (j / nslice) * 8 + xcd;
    sl = j % nslice;
  } else {
    b = blockIdx.x / nslice;
    sl = blockIdx.x % nslice;
  }
  const int c0 = sl * SW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int jl = lane & 31, h = lane >> 5;
  const float* xb = x + (int64_t)b * N * ldx;
  // the Q half waits where z will be written (or in the caller's [P|Q] buffer when the backward wants it)
  float* qdst = pq_out ? pq_out + (int64_t)b * N * ldpq + Co + c0 : z + (int64_t)b * N * Co + c0;
  const int64_t ldq = pq_out ? ldpq : (int64_t)Co;

  // BatchNorm sums are taken about a PIVOT (common.h): y of point 0 / neighbour slot "point 0" of the first cloud of this
  // cloud's domain group, p = P[b0,0] + Q[b0,0], split as pP + pQ.  Every workgroup forms the pivot of its 16 channels
  // itself -- the same 2 x 16 ascending fma chains over the features everywhere, hence the same bits -- so that the
  // partial rows of a group share it; the workgroups of the group's first cloud publish it for the fold.
  float* s_piv = reinterpret_cast<float*>(s_lds) + (size_t)N * SW * (QLDS ? 2 : 1);          // [2*SW]: pP | pQ (true sign)
  if (threadIdx.x < 2 * SW) {
    const int hh = threadIdx.x >> 4, r = threadIdx.x & 15;
    const float* wr = wcat + (int64_t)(hh * Co + c0 + r) * CR;
    const float* x0 = x + (int64_t)(b / Bg) * Bg * N * ldx;
    float d = 0.f;
    for (int t = 0; t < CR; ++t) d = fmaf(wr[t], x0[t], d);
    if (hh == 1 && qbias) d = __fadd_rn(d, qbias[c0 + r]);
    s_piv[threadIdx.x] = d;
  }
  // ---------------------------------------------------------------- phase 1: [P ; Q] slice by MFMA
  {
    // A operand: weight row of A-row jl.  k order: MFMA step t multiplies feature 2t in lanes 0-31 and feature 2t+1 in
    // lanes 32-63, and the instruction adds the two products in that order: the accumulator runs through the features in
    // ASCENDING order from zero -- a plain dot product's fma chain (with it the free-running forward reproduces the
    // reference run's neighbour lists, tests/test_gpu_model.py::test_dgcnn_parity_free_running; halves-of-the-row order
    // flipped 4 near-tie rows of 6144 there)
    const int arow = ((jl >> 2) & 1) * Co + c0 + (jl & 3) + 4 * (jl >> 3);
    float areg[HALF];
    if constexpr (CIN == 4) {
      const float* wr = wcat + (int64_t)arow * CR;
      areg[0] = wr[h];
      areg[1] = h ? 0.f : wr[2];
    } else {
      const float* wr = wcat + (int64_t)arow * CR;
#pragma unroll
      for (int g = 0; g < HALF / 4; ++g) {
        const float4 a = ld4(wr + 8 * g), c = ld4(wr + 8 * g + 4);
        areg[4 * g + 0] = h ? a.y : a.x; areg[4 * g + 1] = h ? a.w : a.z;
        areg[4 * g + 2] = h ? c.y : c.x; areg[4 * g + 3] = h ? c.w : c.z;
      }
    }
    const int ntile = (N + 31) >> 5;
    // the four ds_write_b128 of a P row are issued in a per-lane rotated ORDER (lane j writes quarter (q + (j >> 1)) & 3
    // in its q-th store): the 8 lanes of a store's lane group (rows 64 B apart: banks 0 / 16 alternate) then cover 8
    // different bank quads instead of two (4-way conflict); the LDS image itself is the plain [N][16] layout
    const int rot = (jl >> 1) & 3;
    // x operand chunks (KC k-steps) are double-buffered: chunk c+1 (of this or of the wave's next tile) is in flight
    // while chunk c feeds the MFMAs
    auto loadB = [&](float (&bv)[KC], int tile, int ch) {
#ifdef SUG_EF_ABL_NOLOADX
#pragma unroll
      for (int t = 0; t < KC; ++t) bv[t] = (float)(tile + t);
      return;
#endif
      int r0 = tile * 32 + jl;
      r0 = r0 < N ? r0 : N - 1;
      if constexpr (CIN == 4) {
        const float* p = xb + (int64_t)r0 * ldx;
        bv[0] = p[h];
        const float zc = p[2];
        bv[1] = h ? 0.f : zc;
      } else {
        // chunk ch = features [2 KC ch, 2 KC (ch + 1)) of the row; this lane half loads its contiguous KC of them (the
        // halves trade registers in mma(): a lane never loads a feature it does not multiply)
        const float* p = xb + (int64_t)r0 * ldx + ch * 2 * KC + h * KC;
#pragma unroll
        for (int g = 0; g < KC / 4; ++g) {
          const float4 a = ld4(p + 4 * g);
          bv[4 * g + 0] = a.x; bv[4 * g + 1] = a.y; bv[4 * g + 2] = a.z; bv[4 * g + 3] = a.w;
        }
      }
    };
    f32x16 acc;
    auto mma = [&](float (&bv)[KC], int ch) {
#ifndef SUG_EF_ABL_NOMFMA          // (-DSUG_EF_ABL_*: timing experiments of tools/bench_edgeconv_fused.py, never in the library)
      if constexpr (CIN == 4) {
#pragma unroll
        for (int t = 0; t < KC; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t], bv[t], acc, 0, 0, 0);
      } else {
        // registers (e, e+1) hold chunk features (e, e+1) in lanes 0-31 and (KC+e, KC+e+1) in lanes 32-63;
        // v_permlane32_swap trades the upper half of the first with the lower half of the second: register e then holds
        // features (e | e+1) and register e+1 features (KC+e | KC+e+1) -- consecutive pairs, in place
#pragma unroll
        for (int e = 0; e < KC; e += 2) {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(bv[e]), __float_as_uint(bv[e + 1]), false, false);
          bv[e] = __uint_as_float(r[0]); bv[e + 1] = __uint_as_float(r[1]);
        }
#pragma unroll
        for (int s = 0; s < KC; ++s) {          // step s: chunk features (2s, 2s+1)
          const int r = s < KC / 2 ? 2 * s : 2 * (s - KC / 2) + 1;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[ch * KC + s], bv[r], acc, 0, 0, 0);
        }
      }
#else
      acc[0] += bv[0] + bv[KC - 1];
#endif
    };
    float bA[KC], bB[KC];
    if (wv < ntile) loadB(bA, wv, 0);
    for (int tile = wv; tile < ntile; tile += NW) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int tnext = tile + NW < ntile ? tile + NW : tile;         // (the last tile re-reads itself: no branch around a load)
      if constexpr (NCH == 1) {
        loadB(bB, tnext, 0);
        mma(bA, 0);
#pragma unroll
        for (int t = 0; t < KC; ++t) bA[t] = bB[t];
      } else {
        static_assert(NCH % 2 == 0 || NCH == 1, "chunk pairs");
#pragma unroll
        for (int ch = 0; ch < NCH; ch += 2) {
          loadB(bB, tile, ch + 1);
          mma(bA, ch);
          if (ch + 2 < NCH) loadB(bA, tile, ch + 2); else loadB(bA, tnext, 0);
          mma(bB, ch + 1);
        }
      }
      const int n = tile * 32 + jl;
      if (n < N) {
        if (h == 0) {
          if (pq_out) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              st4(pq_out + ((int64_t)b * N + n) * ldpq + c0 + 4 * q, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int qq = (q + rot) & 3;
            float4 pv;
            pv.x = qq == 0 ? acc[0] : qq == 1 ? acc[4] : qq == 2 ? acc[8] : acc[12];
            pv.y = qq == 0 ? acc[1] : qq == 1 ? acc[5] : qq == 2 ? acc[9] : acc[13];
            pv.z = qq == 0 ? acc[2] : qq == 1 ? acc[6] : qq == 2 ? acc[10] : acc[14];
            pv.w = qq == 0 ? acc[3] : qq == 1 ? acc[7] : qq == 2 ? acc[11] : acc[15];
            const float4 g4 = ld4(gamma + c0 + 4 * qq);
            st4(s_p + n * SW + 4 * qq, make_float4(g4.x >= 0.f ? pv.x : -pv.x, g4.y >= 0.f ? pv.y : -pv.y,
                                                   g4.z >= 0.f ? pv.z : -pv.z, g4.w >= 0.f ? pv.w : -pv.w));
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int qq = QLDS ? (q + rot) & 3 : q;
            float4 qv;
            qv.x = qq == 0 ? acc[0] : qq == 1 ? acc[4] : qq == 2 ? acc[8] : acc[12];
            qv.y = qq == 0 ? acc[1] : qq == 1 ? acc[5] : qq == 2 ? acc[9] : acc[13];
            qv.z = qq == 0 ? acc[2] : qq == 1 ? acc[6] : qq == 2 ? acc[10] : acc[14];
            qv.w = qq == 0 ? acc[3] : qq == 1 ? acc[7] : qq == 2 ? acc[11] : acc[15];
            const float4 b4 = qbias ? ld4(qbias + c0 + 4 * qq) : make_float4(0, 0, 0, 0);
            qv = make_float4(__fadd_rn(qv.x, b4.x), __fadd_rn(qv.y, b4.y), __fadd_rn(qv.z, b4.z), __fadd_rn(qv.w, b4.w));
            if (QLDS) st4(s_q + n * SW + 4 * qq, qv);
            if (!QLDS || pq_out) st4(qdst + (int64_t)n * ldq + 4 * qq, qv);
          }
        }
      }
    }
  }
  __syncthreads();          // P slice complete in LDS; the Q rows of every wave are visible (vmcnt(0) + barrier)
  if (b % Bg == 0 && threadIdx.x < SW)
    ws[SUG_PIVOT_OFFSET(Co) + (size_t)(b / Bg) * Co + c0 + threadIdx.x] = __fadd_rn(s_piv[threadIdx.x], s_piv[SW + threadIdx.x]);

  // ---------------------------------------------------------------- phase 2: gather, reduce over k
  // y_j = P[idx_j] + Q (all sign-folded): fp32 addition is monotone, so max_j y_j = (max_j P[idx_j]) + Q exactly and
  // the running maximum, its (first) position and the sums are taken over the gathered P values alone:
  // sum y = sum p + k q,  sum y^2 = sum p^2 + 2 q sum p + k q^2  (the sums feed the BatchNorm statistics and s1)
  const int lp = threadIdx.x % LP, slot = threadIdx.x / LP;
  const float4 g4 = ld4(gamma + c0 + lp * 4);
  const float4 sg = make_float4(g4.x >= 0.f ? 1.f : -1.f, g4.y >= 0.f ? 1.f : -1.f, g4.z >= 0.f ? 1.f : -1.f,
                                g4.w >= 0.f ? 1.f : -1.f);
  float4 a1 = make_float4(0, 0, 0, 0), a2 = make_float4(0, 0, 0, 0);
  constexpr float kf = (float)KK;
  // pivot parts of this lane's 4 channels, sign-folded like the LDS image
  const float4 pp = make_float4(s_piv[lp * 4 + 0] * sg.x, s_piv[lp * 4 + 1] * sg.y, s_piv[lp * 4 + 2] * sg.z, s_piv[lp * 4 + 3] * sg.w);
  const float4 pq4 = make_float4(s_piv[SW + lp * 4 + 0] * sg.x, s_piv[SW + lp * 4 + 1] * sg.y, s_piv[SW + lp * 4 + 2] * sg.z,
                                 s_piv[SW + lp * 4 + 3] * sg.w);
  // (no software prefetch of the next point's neighbour list: measured, it changes nothing -- 45.2 vs 45.3 us at 64 -> 64; the
  // phase is bound by the VALU work per gathered value and the LDS issue of the random 64-byte rows)
#ifdef SUG_EF_ABL_NOGATHER
  for (int n = N; n < N; n += PPP) {
#else
  for (int n = slot; n < N; n += PPP) {
#endif
    int4 nv[KK / 4];
    float4 q;
    {
      const int64_t p = (int64_t)b * N + n;
      const int4* ir = reinterpret_cast<const int4*>(idx + p * KK);
#pragma unroll
      for (int t = 0; t < KK / 4; ++t) nv[t] = ir[t];
      q = QLDS ? ld4(s_q + n * SW + lp * 4) : ld4(qdst + (int64_t)n * ldq + lp * 4);
      q.x *= sg.x; q.y *= sg.y; q.z *= sg.z; q.w *= sg.w;
    }
    float bx = 0, by = 0, bz = 0, bw = 0;
    int jx = 0, jy = 0, jz = 0, jw = 0;
    float sx = 0, sy = 0, sz = 0, sw = 0, qx = 0, qy = 0, qz = 0, qw = 0;
    auto gather4 = [&](float4 (&pv)[4], const int4& iv) {
      const int m4[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = min(max(m4[u], 0), N - 1);                      // clamp to the cloud (v_med3_i32)
        pv[u] = ld4(s_p + m * SW + lp * 4);
      }
    };
    auto reduce4 = [&](const float4 (&pv)[4], int t) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = t * 4 + u;
        const float dx = pv[u].x - pp.x, dy = pv[u].y - pp.y, dz = pv[u].z - pp.z, dw = pv[u].w - pp.w;   // about the pivot
        sx += dx; sy += dy; sz += dz; sw += dw;
        qx = fmaf(dx, dx, qx); qy = fmaf(dy, dy, qy); qz = fmaf(dz, dz, qz); qw = fmaf(dw, dw, qw);
        if (j == 0 || pv[u].x > bx) { bx = pv[u].x; jx = j; }          // first maximum wins, as torch.max
        if (j == 0 || pv[u].y > by) { by = pv[u].y; jy = j; }
        if (j == 0 || pv[u].z > bz) { bz = pv[u].z; jz = j; }
        if (j == 0 || pv[u].w > bw) { bw = pv[u].w; jw = j; }
      }
    };
    // batches of 4 neighbours: 4 LDS reads in flight, then their arithmetic (the other wave of the SIMD covers the read
    // latency; a second register set for a software pipeline moved the total by < 5 %)
    float4 pa[4];
#pragma unroll
    for (int t = 0; t < KK / 4; ++t) {
      gather4(pa, nv[t]);
      __builtin_amdgcn_sched_barrier(0);
      reduce4(pa, t);
      __builtin_amdgcn_sched_barrier(0);
    }
    // + Q, back to the true sign
    const float zx = __fadd_rn(bx, q.x) * sg.x, zy = __fadd_rn(by, q.y) * sg.y;
    const float zz = __fadd_rn(bz, q.z) * sg.z, zw = __fadd_rn(bw, q.w) * sg.w;
    // y' - p' = (P' - pP') + (Q' - pQ') = d + e:  sum = sd + k e,  sum of squares = sum d^2 + e (2 sd + k e)
    const float ex = q.x - pq4.x, ey = q.y - pq4.y, ez = q.z - pq4.z, ew = q.w - pq4.w;
    const float yx = fmaf(kf, ex, sx), yy = fmaf(kf, ey, sy), yz = fmaf(kf, ez, sz), yw = fmaf(kf, ew, sw);
    qx = fmaf(ex, yx + sx, qx); qy = fmaf(ey, yy + sy, qy);
    qz = fmaf(ez, yz + sz, qz); qw = fmaf(ew, yw + sw, qw);
    const int64_t o = ((int64_t)b * N + n) * Co + c0 + lp * 4;
#ifndef SUG_EF_ABL_NOZ
    st4(z + o, make_float4(zx, zy, zz, zw));                          // (overwrites this point's parked Q quarter, if any)
#else
    if (zx == 123.456f) st4(z + o, make_float4(zx, zy, zz, zw));
#endif
#ifndef SUG_EF_ABL_NOARG
    if (arg)
      *reinterpret_cast<uint32_t*>(arg + o) =
          (uint32_t)jx | ((uint32_t)jy << 8) | ((uint32_t)jz << 16) | ((uint32_t)jw << 24);
#endif
    const float4 sy4 = make_float4(yx * sg.x, yy * sg.y, yz * sg.z, yw * sg.w);          // sum_j (y - p), true sign
    // s1 = sum_j y (unshifted: the backward's dQ formula wants it) = sum_j (y - p) + k p
    if (s1) st4(s1 + o, make_float4(fmaf(kf, pp.x + pq4.x, yx) * sg.x, fmaf(kf, pp.y + pq4.y, yy) * sg.y,
                                    fmaf(kf, pp.z + pq4.z, yz) * sg.z, fmaf(kf, pp.w + pq4.w, yw) * sg.w));
    a1.x += sy4.x; a1.y += sy4.y; a1.z += sy4.z; a1.w += sy4.w;
    a2.x += qx; a2.y += qy; a2.z += qz; a2.w += qw;
  }
  // fixed-order reduction over the point slots (fp64), one partial row per cloud; the scratch reuses the P image:
  // 2*SW columns x PPP slots, folded by 16 threads per column (slots t, t+16, ...) and then by one
  __syncthreads();
  float* s_red = s_lds;                            // [PPP][2*SW]
  st4(s_red + slot * 2 * SW + lp * 4, a1);
  st4(s_red + slot * 2 * SW + SW + lp * 4, a2);
  __syncthreads();
  double part = 0.0;
  {
    const int col = threadIdx.x & (2 * SW - 1), tl = (threadIdx.x >> 5) & 15;
    if (threadIdx.x < 2 * SW * 16) {
      for (int t = tl; t < PPP; t += 16) part += (double)s_red[t * 2 * SW + col];
    }
  }
  __syncthreads();
  double* s_d = reinterpret_cast<double*>(s_lds);  // [16][2*SW]
  if (threadIdx.x < 2 * SW * 16) s_d[threadIdx.x] = part;
  __syncthreads();
  if (threadIdx.x < 2 * SW) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc += s_d[t * 2 * SW + threadIdx.x];
    const int col = threadIdx.x < SW ? c0 + threadIdx.x : Co + c0 + (threadIdx.x - SW);
    ws[(size_t)b * 2 * Co + col] = (float)acc;
  }
}

// Second (and last) launch of a fused layer forward: fold the per-cloud partial rows into the BatchNorm coefficients
// (every workgroup folds its own group's rows redundantly, in one fixed order: <= 64 rows of 2*Co columns out of
// L2), then out = LeakyReLU_slope(scale * z + shift) for the workgroup's rows.  Workgroup (0, 0) additionally writes
// coef [G,5,Co] and updates the running statistics with the groups' batch statistics in group order (one
// nn.BatchNorm2d call per domain group, as the reference's separate forward calls).
__global__ __launch_bounds__(256) void edgeconv_bn_act_kernel(
    const float* __restrict__ ws, int nblk, int Co, int groups, const float* __restrict__ gamma,
    const float* __restrict__ beta, double count, float eps, float momentum, float* __restrict__ rmean,
    float* __restrict__ rvar, float* __restrict__ coef, const float* __restrict__ z, int64_t rows_g, float slope,
    float* __restrict__ out, int64_t ldo, int rows_per_wg, const float* __restrict__ pivot) {
  extern __shared__ __attribute__((aligned(16))) float s_c[];      // scale [Co] | shift [Co]
  const int g = blockIdx.y;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
  auto fold = [&](int gg, int c, double& mean, double& var) {
    const float* w = ws + (size_t)gg * nblk * 2 * Co;
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int r = 0; r < nblk; ++r) {
      s += (double)w[(size_t)r * 2 * Co + c];
      q += (double)w[(size_t)r * 2 * Co + Co + c];
    }
    const double m1 = s / count;                     // mean of (y - pivot)
    var = q / count - m1 * m1;
    if (var < 0) var = 0;
    mean = (pivot ? (double)pivot[(size_t)gg * Co + c] : 0.0) + m1;
  };
  for (int c = threadIdx.x; c < Co; c += 256) {
    double mean, var;
    fold(g, c, mean, var);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * rstd;
    s_c[c] = scale;
    s_c[Co + c] = beta[c] - (float)mean * scale;
    if (writer) {
      float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
      for (int gg = 0; gg < groups; ++gg) {
        double m2, v2;
        fold(gg, c, m2, v2);
        const float rs = (float)(1.0 / sqrt(v2 + (double)eps));
        const float sc = gamma[c] * rs;
        float* cg = coef + (size_t)gg * 5 * Co;
        const double unb = count > 1.0 ? v2 * count / (count - 1.0) : v2;
        cg[c] = sc;
        cg[Co + c] = beta[c] - (float)m2 * sc;
        cg[2 * Co + c] = (float)m2;
        cg[3 * Co + c] = rs;
        cg[4 * Co + c] = (float)unb;
        rm = (1.f - momentum) * rm + momentum * (float)m2;
        rv = (1.f - momentum) * rv + momentum * (float)unb;
      }
      if (rmean) rmean[c] = rm;
      if (rvar) rvar[c] = rv;
    }
  }
  __syncthreads();
  const int C4 = Co >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  int64_t r1 = r0 + rows_per_wg;
  if (r1 > rows_g) r1 = rows_g;
  const int64_t total = (r1 - r0) * C4;
  const float* zg = z + ((int64_t)g * rows_g + r0) * Co;
  float* og = out + ((int64_t)g * rows_g + r0) * ldo;
  constexpr int U = 4;
  for (int64_t e0 = threadIdx.x; e0 < total; e0 += U * 256) {
    float4 zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t e = e0 + u * 256;
      zv[u] = ld4(zg + 4 * (e < total ? e : e0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t e = e0 + u * 256;
      if (e >= total) continue;
      const int c = (int)(e % C4) * 4;
      const int64_t r = e / C4;
      const float4 sc = ld4(s_c + c), sh = ld4(s_c + Co + c);
      float4 v;
      v.x = fmaf(sc.x, zv[u].x, sh.x); v.y = fmaf(sc.y, zv[u].y, sh.y);
      v.z = fmaf(sc.z, zv[u].z, sh.z); v.w = fmaf(sc.w, zv[u].w, sh.w);
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      st4(og + r * ldo + c, v);
    }
  }
}

template <int CIN, int NT, bool QLDS>
int launch_fused_nt(const float* x, int64_t ldx, const float* wcat, const float* qbias, const int32_t* idx,
                    const float* gamma, int B, int N, int Co, int Bg, float* z, uint8_t* arg, float* s1, float* pq_out,
                    int64_t ldpq, float* ws, hipStream_t st) {
  const size_t plds = (size_t)N * SW * sizeof(float) * (QLDS ? 2 : 1) + 2 * SW * sizeof(float);      // P (| Q) image + pivot row
  const size_t rlds = (size_t)(NT / LP) * 2 * SW * sizeof(float);
  const size_t sh = plds > rlds ? plds : rlds;
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &edgeconv_fused_fwd_kernel<CIN, 20, NT, QLDS>, 160 * 1024, "sug_edgeconv_fused_layer_fwd"))
    return rc;
  hipLaunchKernelGGL((edgeconv_fused_fwd_kernel<CIN, 20, NT, QLDS>), dim3(B * (Co / SW)), dim3(NT), sh, st, x, ldx, wcat, qbias,
                     idx, gamma, B, N, Co, Bg, z, arg, s1, pq_out, ldpq, ws);
  SUG_LAUNCH_CHECK("sug_edgeconv_fused_layer_fwd");
  return SUG_OK;
}

template <int CIN>
int launch_fused(const float* x, int64_t ldx, const float* wcat, const float* qbias, const int32_t* idx,
                 const float* gamma, int B, int N, int Co, int Bg, float* z, uint8_t* arg, float* s1, float* pq_out,
                 int64_t ldpq, float* ws, hipStream_t st) {
  // P and Q slices both in LDS (one 16-wave workgroup per CU) while they fit; larger clouds park Q in the z buffer
  const bool qlds = (size_t)N * SW * sizeof(float) * 2 + 2 * SW * sizeof(float) <= 160 * 1024;
#ifdef SUG_EF_ABL_QGLOBAL
  return launch_fused_nt<CIN, 512, false>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, Bg, z, arg, s1, pq_out, ldpq, ws, st);
#endif
  if (qlds) return launch_fused_nt<CIN, 512, true>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, Bg, z, arg, s1, pq_out, ldpq, ws, st);
  return launch_fused_nt<CIN, 512, false>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, Bg, z, arg, s1, pq_out, ldpq, ws, st);
}

}  // namespace

int sug_affine_act_groups(const float* z, int64_t ldz, const float* coef, int64_t rows, int groups, int C, float slope,
                          float* out, int64_t ldo, hipStream_t st);

// Statistics fold + BatchNorm + LeakyReLU of an EdgeConv layer in ONE launch (edgeconv_bn_act_kernel), shared by the
// fused layer below and by the library-GEMM path (edgeconv.hip): ws = `groups` x nblk partial rows [2Co] (group-major),
// z [groups*rows_g, Co] dense, count = samples per channel and group.
int sug_edgeconv_bn_act(const float* ws, int nblk, int Co, int groups, const float* gamma, const float* beta, double count,
                        float eps, float momentum, float* running_mean, float* running_var, float* coef, const float* z,
                        int64_t rows_g, float slope, float* out, int64_t ldo, hipStream_t st) {
  // the partial rows are sums about the pivot row the producers published behind them (common.h)
  const float* pivot = ws + SUG_PIVOT_OFFSET(Co);
  int rpw = 64;                                   // rows per workgroup: >= 1024 workgroups where the layer has them
  while ((rows_g + rpw - 1) / rpw * groups > 4096) rpw *= 2;
  hipLaunchKernelGGL(edgeconv_bn_act_kernel, dim3((unsigned)((rows_g + rpw - 1) / rpw), groups), dim3(256),
                     (size_t)2 * Co * sizeof(float), st, ws, nblk, Co, groups, gamma, beta, count, eps, momentum,
                     running_mean, running_var, coef, z, rows_g, slope, out, ldo, rpw, pivot);
  SUG_LAUNCH_CHECK("sug_edgeconv_bn_act");
  return SUG_OK;
}

extern "C" int sug_edgeconv_fused_supported(int N, int k, int Cin, int Co) {
  return k == 20 && (Cin == 3 || Cin == 64 || Cin == 128) && Co % 16 == 0 && Co >= 16 && Co <= 1024 && N >= 32 &&
         (size_t)N * SW * 4 + 2 * SW * 4 <= 160 * 1024;
}

extern "C" int sug_edgeconv_fused_layer_fwd(const float* x, int64_t ldx, int Cin, const float* wcat, const float* qbias,
                                            const int32_t* idx, const float* gamma, const float* beta, int B, int N,
                                            int k, int Co, int groups, int training, float eps, float momentum,
                                            float slope, float* running_mean, float* running_var, float* z,
                                            uint8_t* arg, float* s1, float* pq_out, int64_t ldpq, float* coef,
                                            float* out, int64_t ldo, float* ws, void* stream) {
  SUG_REQUIRE(x && wcat && idx && gamma && beta && z && coef && out && ws, "sug_edgeconv_fused_layer_fwd: null pointer");
  SUG_REQUIRE(B > 0 && groups >= 1 && B % groups == 0, "sug_edgeconv_fused_layer_fwd: B=%d does not split into %d groups", B, groups);
  SUG_REQUIRE(groups <= 16, "sug_edgeconv_fused_layer_fwd: %d groups, the workspace reserves pivot rows for 16", groups);
  SUG_REQUIRE(sug_edgeconv_fused_supported(N, k, Cin, Co), "sug_edgeconv_fused_layer_fwd: unsupported shape N=%d k=%d Cin=%d Co=%d", N, k, Cin, Co);
  SUG_REQUIRE(B <= SUG_STATS_ROWS, "sug_edgeconv_fused_layer_fwd: B=%d exceeds the statistics workspace", B);
  SUG_REQUIRE(ldx >= Cin && (Cin == 3 || (ldx % 4 == 0 && ((uintptr_t)x % 16) == 0)), "sug_edgeconv_fused_layer_fwd: x rows must be 16-byte aligned");
  SUG_REQUIRE(((uintptr_t)wcat % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)idx % 16) == 0 &&
                  (!arg || ((uintptr_t)arg % 4) == 0) && (!s1 || ((uintptr_t)s1 % 16) == 0) && (!qbias || ((uintptr_t)qbias % 16) == 0) &&
                  ((uintptr_t)out % 16) == 0 && ldo % 4 == 0 && ldo >= Co,
              "sug_edgeconv_fused_layer_fwd: pointers must be 16-byte aligned");
  SUG_REQUIRE(!pq_out || (ldpq >= 2 * Co && ldpq % 4 == 0 && ((uintptr_t)pq_out % 16) == 0), "sug_edgeconv_fused_layer_fwd: bad [P|Q] buffer");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (Cin == 3)
    rc = launch_fused<4>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, B / groups, z, arg, s1, pq_out, ldpq, ws, st);
  else if (Cin == 64)
    rc = launch_fused<64>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, B / groups, z, arg, s1, pq_out, ldpq, ws, st);
  else
    rc = launch_fused<128>(x, ldx, wcat, qbias, idx, gamma, B, N, Co, B / groups, z, arg, s1, pq_out, ldpq, ws, st);
  if (rc != SUG_OK) return rc;
  const int64_t rows_g = (int64_t)(B / groups) * N;
#ifdef SUG_EF_ABL_NOACT
  return SUG_OK;
#endif
  if (!training)               // eval mode: coef holds the running-statistics coefficients of every group
    return sug_affine_act_groups(z, Co, coef, rows_g, groups, Co, slope, out, ldo, st) == 0 ? SUG_OK : SUG_ERR_ARG;
  return sug_edgeconv_bn_act(ws, B / groups, Co, groups, gamma, beta, (double)rows_g * k, eps, momentum, running_mean,
                             running_var, coef, z, rows_g, slope, out, ldo, st);
}
