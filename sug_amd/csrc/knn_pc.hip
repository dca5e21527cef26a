// kNN graph, producer / consumer formulation (C = 3, 64, 128, k <= 20): the hot index kernel of the
// DGCNN path.  Replaces knn(), model/model_utils.py:178-185.
//
// The score tiles <x_i, x_j> run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: bit for bit the
// ascending-k fmaf chain of the scalar kernel and of the CPU reference's K=3 sgemm), the top-k
// selection is VALU work.  On gfx950 fp32 MFMA time and VALU time ADD on a SIMD (tools/ubench, DESIGN.md
// section 7), and a VALU instruction between two dependent MFMAs also breaks the accumulator forwarding:
// the two kinds of work live in DIFFERENT waves of one 512-thread workgroup, every SIMD hosts one of
// each, the MFMA chains stay back to back and the selection is kept as short as it can be made:
//   waves 0-3  producers: stream 32-candidate tiles global -> registers -> LDS (mfma_tile.h), run
//              the MFMA chains of 64 queries each (two 32-query column blocks sharing the A
//              operand) and write the 64 x 32 score tile to LDS, row = query;
//   waves 4-7  consumers: lane = ONE query; reads its row of the score tile (candidates in ascending
//              index order), forms score = (-|x_j|^2 - (-2<x_i,x_j>)) - |x_i|^2 and compares with a
//              threshold derived from its (K+2)-th best key as of the previous tile; the few candidates that
//              pass are marked in a 32-bit mask and then inserted -- all lanes of the wave in lockstep, one
//              marked candidate per lane per iteration -- into the lane's sorted list of PACKED KEYS in
//              registers: key = (fixed-point bucket of d = -score over the query's range of the first tile)
//              << ceil(log2 N) | candidate index, one v_med3_i32 per slot (the exact (score, index) lists
//              of the first round-2 kernel cost 4 VALU per slot).
//              The bucket map is a monotone coarsening of d, so the K+2 smallest keys contain the exact
//              top K whenever fewer than 3 keys share the boundary bucket; at the end a lane whose first
//              K+1 keys lie in distinct buckets has the exact answer in key order (the common case), a lane
//              with equal buckets has its K+2 candidates re-scored exactly (fma chain in the MFMA's k
//              order, one lane per candidate) and ranked by (score, index), and a lane whose boundary
//              bucket overflows rescans the cloud exactly (scalar, LDS-resident lists): all three paths
//              give the lists of the exact consumer bit for bit (tests/golden/knn_pc_hashes.json).
// One barrier per tile hands tile t's scores to the consumers while the producers work on tile
// t+1.  No candidate ring, no compaction, no merge.
// LDS: tiles 3 x 32 x (C+4) + score tiles 2 x QB x 36 floats = 100 / 125 KB (C = 64 / 128, QB = 256 queries) or 63 KB (C = 64, QB = 128).
// Workgroup = QB queries of one cloud; the workgroups of a cloud share an XCD (one L2).
//
// FLOPs N^2*(2C+3) per cloud on the matrix pipe (157 TFLOP/s); algorithmic bytes 4*C*N + 4*N*k.
#include <float.h>
#include "common.h"
#include "mfma_tile.h"


namespace {
using namespace sug_tile;

constexpr int SROW = 36;        // floats per query row of a score tile (32 + pad: conflict-free b128)


__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// Sorted insertion of key x into the ascending list L[0..KP-1], in place, one VALU per slot:
//   L[u] = med3(x, L[u-1], L[u]) for u = KP-1 .. 1 (L[u-1] is read before it is overwritten), L[0] = min(L[0], x).
// A key that is not smaller than L[KP-1] leaves the list unchanged.  (asm: hipcc's register allocation of
// the C++ loop copies the list registers every iteration.)
template <int KP>
__device__ __forceinline__ void insert_key_hi(int (&L)[KP], int x) {          // slots KP-1 .. KP/2
#pragma unroll
  for (int u = KP - 1; u >= KP / 2; --u) asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(L[u]) : "v"(x), "v"(L[u - 1]));
}
template <int KP>
__device__ __forceinline__ void insert_key_lo(int (&L)[KP], int x) {          // slots KP/2-1 .. 0
#pragma unroll
  for (int u = KP / 2 - 1; u >= 1; --u) asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(L[u]) : "v"(x), "v"(L[u - 1]));
  asm volatile("v_min_i32 %0, %0, %1" : "+v"(L[0]) : "v"(x));
}

// <a, b> over C features in the k order of the MFMA chain (ascending feature index, fma from a zero accumulator)
template <int CP>
__device__ __forceinline__ float exact_dot(const float* __restrict__ a, const float* __restrict__ b) {
  float acc = 0.f;
  if constexpr (CP == 4) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc);
  } else {
    for (int e = 0; e < CP; e += 4) {
      const float4 av = *reinterpret_cast<const float4*>(a + e), bv = *reinterpret_cast<const float4*>(b + e);
      acc = fmaf(av.x, bv.x, acc); acc = fmaf(av.y, bv.y, acc); acc = fmaf(av.z, bv.z, acc); acc = fmaf(av.w, bv.w, acc);
    }
  }
  return acc;
}
// |x_j|^2 exactly as tile_store forms it: fma chain inside each 8-feature chunk, xor tree over the chunks
template <int CP>
__device__ __forceinline__ float exact_norm(const float* __restrict__ r) {
  if constexpr (CP == 4) {
    return sq3(r[0], r[1], r[2]);
  } else {
    constexpr int CH = CP / 8;
    float pc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(r + 8 * c), b = *reinterpret_cast<const float4*>(r + 8 * c + 4);
      float p = __fmul_rn(a.x, a.x);
      p = fmaf(a.y, a.y, p); p = fmaf(a.z, a.z, p); p = fmaf(a.w, a.w, p);
      p = fmaf(b.x, b.x, p); p = fmaf(b.y, b.y, p); p = fmaf(b.z, b.z, p); p = fmaf(b.w, b.w, p);
      pc[c] = p;
    }
#pragma unroll
    for (int o = 1; o < CH; o <<= 1)
#pragma unroll
      for (int c = 0; c < CH; c += 2 * o) pc[c] = pc[c] + pc[c + o];
    return pc[0];
  }
}

// NP producer waves, NC consumer waves per workgroup; the workgroup serves QB = 64 NC queries, a producer wave scores
// QP = QB / NP of them (64: two 32-query column blocks sharing the A operand; 32: one).
//   NP = NC = 4 (product): 256 queries, 512 threads, one workgroup per CU (100 / 125 KB of LDS);
//   NP = NC = 2 (small batches: see launch_pc): 128 queries, 256 threads;
//   NP = 4, NC = 2 (round 6, small batches at C >= 64): 128 queries, 384 threads -- a 128-query workgroup per CU leaves two
//     of the four SIMDs without a producer when NP = 2 (the matrix pipe of half the chip idle, 109 us of MFMA chains per
//     launch at C = 128 on the other half); four producers of 32 queries put a chain on every SIMD.
// (Which waves of the workgroup take which role makes no difference: consumers first, in the middle or last measured within 1 us.)
template <int CP, int K, int NP, int NC>
__global__ __launch_bounds__(64 * (NP + NC), 2) void knn_pc_kernel(const float* __restrict__ x, int64_t ldx, int B, int N,
                                                                   int k, int32_t* __restrict__ idx, int force) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  constexpr int NTP = 64 * NP;                                          // producer threads
  constexpr int QB = 64 * NC;                                           // queries per workgroup
  constexpr int QP = QB / NP;                                           // queries per producer wave
  static_assert(QP == 32 || QP == 64, "a producer wave scores one or two 32-query column blocks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);                       // [3][TJ][RS]
  float* s_norm = s_tile + 3 * TJ * RS;                                 // [3][TJ] (+pad)
  float* s_score = s_norm + 4 * TJ;                                     // [2][QB][SROW]

  const int nq = (N + QB - 1) / QB;
  int b, qb;
  if ((B & 7) == 0) {                        // a cloud's query blocks share an XCD (its L2 holds the cloud once)
    const int grp = blockIdx.x / (8 * nq), rem = blockIdx.x % (8 * nq);
    b = grp * 8 + (rem & 7);
    qb = rem >> 3;
  } else {
    b = blockIdx.x / nq;
    qb = blockIdx.x % nq;
  }
  const float* xb = x + (int64_t)b * N * ldx;
  const int wave = (int)(threadIdx.x >> 6);
  const bool producer = wave < NP;
  const int lane = threadIdx.x & 63;
  const int wv = producer ? wave : wave - NP;                           // index among the producer / the consumer waves
  const int ptid = (int)threadIdx.x;                                    // producer thread number (staging)
  const int qj = lane & 31, h = lane >> 5;
  const int q0 = qb * QB;

  const int ntile = (N + TJ - 1) / TJ;
  auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };
  auto nbuf = [&](int t) { return s_norm + (t % 3) * TJ; };

  // The two roles are separate code paths (disjoint register sets); both execute the same sequence
  // of workgroup barriers: 2 in the pipeline fill, one per tile.
  if (producer) {
    // ---- query operands (B operand of both column blocks, whole kernel in registers), straight from global memory: lane
    // (qj, h) loads the contiguous half [h HALF, h HALF + HALF) of its two query rows; v_permlane32_swap then trades
    // registers between the lane halves: register pair (e, e+1) = features (e, e+1 | HALF+e, HALF+e+1) becomes
    // (e | e+1) and (HALF+e | HALF+e+1), i.e. register breg(s) holds features (2s | 2s+1) = the k pair of MFMA step s --
    // the same operands the de-interleaved LDS image of mfma_tile.h gives the A side.  They are scaled by -2 (exact): the
    // chains then deliver inner = -2<x_i,x_j> itself, bit for bit fl(-2 * dot), and the consumers save the
    // multiplication per candidate.  (Until round 3 the query tiles were staged through the tile buffers: 4 PW + 1
    // barriers and 2 PW dependent load -> store -> read steps, 10 us of the kernel.)
    TileRegs<CP, NTP> tr;
    tile_load<CP, NTP>(tr, xb, ldx, N, 0, ptid);       // candidate tile 0: in flight under the query loads
    float bq0[HALF], bq1[HALF];
    {
      const int r0 = min(q0 + wv * QP + qj, N - 1), r1 = min(q0 + wv * QP + 32 + qj, N - 1);   // (rows past N: results unused; QP = 32: r1 unused)
      const float* p0 = xb + (int64_t)r0 * ldx;
      const float* p1 = xb + (int64_t)r1 * ldx;
      if constexpr (CP == 4) {
        const float x0 = p0[0], y0 = p0[1], z0 = p0[2], x1 = p1[0], y1 = p1[1], z1 = p1[2];
        bq0[0] = -2.f * (h ? y0 : x0); bq0[1] = h ? 0.f : -2.f * z0;
        bq1[0] = -2.f * (h ? y1 : x1); bq1[1] = h ? 0.f : -2.f * z1;
      } else {
#pragma unroll
        for (int g = 0; g < HALF / 4; ++g) {
          const float4 a = *reinterpret_cast<const float4*>(p0 + h * HALF + 4 * g);
          const float4 c = *reinterpret_cast<const float4*>(p1 + h * HALF + 4 * g);
          bq0[4 * g] = -2.f * a.x; bq0[4 * g + 1] = -2.f * a.y; bq0[4 * g + 2] = -2.f * a.z; bq0[4 * g + 3] = -2.f * a.w;
          bq1[4 * g] = -2.f * c.x; bq1[4 * g + 1] = -2.f * c.y; bq1[4 * g + 2] = -2.f * c.z; bq1[4 * g + 3] = -2.f * c.w;
        }
#pragma unroll
        for (int e = 0; e < HALF; e += 2) {
          const auto u = __builtin_amdgcn_permlane32_swap(__float_as_uint(bq0[e]), __float_as_uint(bq0[e + 1]), false, false);
          bq0[e] = __uint_as_float(u[0]); bq0[e + 1] = __uint_as_float(u[1]);
          const auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(bq1[e]), __float_as_uint(bq1[e + 1]), false, false);
          bq1[e] = __uint_as_float(v[0]); bq1[e + 1] = __uint_as_float(v[1]);
        }
      }
    }
    // register of k-step s
    auto breg = [](int s2) constexpr { return CP == 4 ? s2 : (s2 < HALF / 2 ? 2 * s2 : 2 * (s2 - HALF / 2) + 1); };

    // scores of candidate tile t for this wave's 64 queries -> score buffer `buf`
    auto produce = [&](int t, int buf, auto&& mid) {
      const float* arow = tbuf(t) + qj * RS + h * HALF;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      // QP = 64: the two column blocks share the A operand; their chains alternate on the matrix pipe
      if constexpr (CP == 4) {
        const float2 a2 = *reinterpret_cast<const float2*>(arow);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, bq0[0], acc0, 0, 0, 0);
        if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, bq1[0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, bq0[1], acc0, 0, 0, 0);
        if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, bq1[1], acc1, 0, 0, 0);
        mid();
      } else {
#pragma unroll
        for (int g = 0; g < HALF / 4; ++g) {
          if (g == HALF / 8) mid();             // (empty in the product; tools/ubench/knn_pc_ablations.patch: staging inside the chain)
          const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq0[breg(4 * g + 0)], acc0, 0, 0, 0);
          if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq1[breg(4 * g + 0)], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq0[breg(4 * g + 1)], acc0, 0, 0, 0);
          if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq1[breg(4 * g + 1)], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq0[breg(4 * g + 2)], acc0, 0, 0, 0);
          if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq1[breg(4 * g + 2)], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq0[breg(4 * g + 3)], acc0, 0, 0, 0);
          if constexpr (QP == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq1[breg(4 * g + 3)], acc1, 0, 0, 0);
        }
      }
      // S^T tile: lane = query column, registers 4g..4g+3 = candidate rows 8g + 4h .. +3: one b128 per g
      float* d0 = s_score + ((buf * QB + wv * QP) + qj) * SROW + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(d0 + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
      if constexpr (QP == 64) {
        float* d1 = d0 + 32 * SROW;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(d1 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
      }
    };

    // pipeline: iteration t = scores of tile t+1 (consumers are on tile t); registers of tile t+2 -> LDS;
    // the global loads of tile t+3 are in flight for a whole iteration
    tile_store<CP, true, NTP>(tr, tbuf(0), nbuf(0), N, 0, ptid);
    if (ntile > 1) tile_load<CP, NTP>(tr, xb, ldx, N, TJ, ptid);
    __syncthreads();
    produce(0, 0, [] {});
    if (ntile > 1) tile_store<CP, true, NTP>(tr, tbuf(1), nbuf(1), N, TJ, ptid);
    if (ntile > 2) tile_load<CP, NTP>(tr, xb, ldx, N, 2 * TJ, ptid);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
      auto stage = [&] {
        if (t + 2 < ntile) tile_store<CP, true, NTP>(tr, tbuf(t + 2), nbuf(t + 2), N, (t + 2) * TJ, ptid);
        if (t + 3 < ntile) tile_load<CP, NTP>(tr, xb, ldx, N, (t + 3) * TJ, ptid);
      };
      if (t + 1 < ntile) produce(t + 1, (t + 1) & 1, [] {});
      stage();
      __syncthreads();
    }
  } else {
    __builtin_amdgcn_s_setprio(1);     // the selection waves are the second-dispatched half of the workgroup: static priority (-5 us at C=3)
    // ---- consumer: lane = one query; |x_i|^2 in the rounding order of tile_store's norms (exact_norm)
    const float ni = exact_norm<CP>(xb + (int64_t)min(q0 + wv * 64 + lane, N - 1) * ldx);

    constexpr int KP = K + 2;
    constexpr int EMPTY = 0x7fffffff;          // never-filled slot
    int L[KP];                                 // ascending packed keys
#pragma unroll
    for (int t = 0; t < KP; ++t) L[t] = EMPTY;
    const int q = q0 + wv * 64 + lane;
    const int idb = N > 1 ? 32 - __clz(N - 1) : 0;       // index bits
    const int idm = (1 << idb) - 1, nidm = ~idm;
    // bucket = min(floor(d * scale), FXMAX), d = -score: fixed point over [0, R], R = the largest d of the query's
    // first tile (the K+2 smallest d of the cloud cannot exceed it once that tile holds K+2 candidates; larger d
    // saturate, which keeps the map monotone).  31 - idb bits; FXMAX leaves the EMPTY key a bucket of its own.
    const unsigned FXMAX = (1u << (31 - idb)) - 2u;
    float scale = 1.f, inv_scale = 1.f;
    float thr = -FLT_MAX;                      // candidates with score >= thr may still enter the list

    // score of candidate slot c of the current tile: pairwise_distance = -xx - inner - xx^T, inner = -2*dot
    // (model_utils.py:179-181); the same three roundings wherever it is evaluated
    auto score = [&](float inner, float nj) { return __fsub_rn(__fsub_rn(-nj, inner), ni); };

    __syncthreads();                            // pipeline fill: tile 0 staged
    __syncthreads();                            // tile 0 scored
    for (int t = 0; t < ntile; ++t) {
      const float* srow = s_score + (((t & 1) * QB + wv * 64) + lane) * SROW;
      const float* nrm = nbuf(t);
      // pass 1: which of the 32 candidates can still enter (thr: from the (K+2)-th key as of the previous tile:
      // stale by at most one tile, never too high; rows past N score -inf and never pass).  Candidate c -> bit 31-c.
      unsigned int mask = 0u;
      if (t == 0) {
        // first tile: every finite score passes; its lowest score sets the range of the fixed-point buckets
        float smin = INFINITY;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const float4 s4 = *reinterpret_cast<const float4*>(srow + 4 * g);
          const float4 n4 = *reinterpret_cast<const float4*>(nrm + 4 * g);
          const float sx = score(s4.x, n4.x), sy = score(s4.y, n4.y), sz = score(s4.z, n4.z), sw = score(s4.w, n4.w);
          mask = mask + mask + (sx >= thr ? 1u : 0u);
          mask = mask + mask + (sy >= thr ? 1u : 0u);
          mask = mask + mask + (sz >= thr ? 1u : 0u);
          mask = mask + mask + (sw >= thr ? 1u : 0u);
          smin = fminf(smin, sx >= thr ? sx : INFINITY);
          smin = fminf(smin, sy >= thr ? sy : INFINITY);
          smin = fminf(smin, sz >= thr ? sz : INFINITY);
          smin = fminf(smin, sw >= thr ? sw : INFINITY);
        }
        float R = (smin < 0.f && smin > -FLT_MAX) ? -smin : 1.f;            // no finite negative score: any range will do
        scale = (float)FXMAX / R;
        inv_scale = R / (float)FXMAX;
        if (!(scale < FLT_MAX) || !(inv_scale > 0.f)) { scale = 1.f; inv_scale = 1.f; }
      } else {
        // score >= thr, conservatively, in two instructions per candidate: with u = fl(-nj - inner) (the score's own first
        // rounding) the score is fl(u - ni), so score >= thr implies u >= thr + ni - ulp: compare u with
        // thr2 = fl(thr + ni) - 2.4e-7 (|thr| + |ni|) through the sign bit of fl(u - thr2) (a difference has the sign of the
        // exact one; u = -inf for rows past N gives -inf: rejected), shifted into the mask by v_alignbit.  A candidate
        // that passes here and not the exact test only costs an insertion that leaves the list unchanged.
        const float thr2 = thr == -FLT_MAX ? -FLT_MAX : __fsub_rn(__fadd_rn(thr, ni), 2.4e-7f * (fabsf(thr) + fabsf(ni)));
        unsigned int rej = 0u;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const float4 s4 = *reinterpret_cast<const float4*>(srow + 4 * g);
          const float4 n4 = *reinterpret_cast<const float4*>(nrm + 4 * g);
          const float wx = __fsub_rn(__fsub_rn(-n4.x, s4.x), thr2), wy = __fsub_rn(__fsub_rn(-n4.y, s4.y), thr2);
          const float wz = __fsub_rn(__fsub_rn(-n4.z, s4.z), thr2), ww = __fsub_rn(__fsub_rn(-n4.w, s4.w), thr2);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(wx), 31);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(wy), 31);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(wz), 31);
          rej = __builtin_amdgcn_alignbit(rej, __float_as_uint(ww), 31);
        }
        mask = ~rej;
      }
      // pass 2: the marked candidates in ascending index order, one per lane per iteration (all lanes in
      // lockstep); iterations = the largest number of marked candidates of any lane: a wave-wide maximum built
      // bit by bit from ballots (scalar unit only; a shuffle tree would cost six LDS round trips per tile)
      const int pc = __popc(mask);
      unsigned long long cand = ~0ull;
      int nit = 0;
#pragma unroll
      for (int bit = 5; bit >= 0; --bit) {
        const unsigned long long m = __ballot((pc >> bit) & 1) & cand;
        if (m) {
          nit |= 1 << bit;
          cand = m;
        }
      }
      // The LDS reads are issued by hand and waited for only after the insertion of the previous entry
      // (loads are unconditional -- an unmarked lane re-reads slot 0 -- so that no branch wraps them).
      const unsigned srow_a = lds_addr(srow), nrm_a = lds_addr(nrm);
      // Two entries in flight (register sets A and B): the LDS reads of entry n+2 are issued before entry n is
      // inserted, the key of entry n+1 is formed between the two halves of that insertion (its reads have had a
      // whole iteration; lgkmcnt(2) waits for the older pair only) -- with one entry in flight the read latency
      // plus the dependent key arithmetic (~25 quad-cycles) sat on the critical path of every iteration.
      float dotA, njA, dotB, njB;
      bool vA, vB;
      int cA, cB;
      auto fetch_issue = [&](float& dot, float& nj, bool& valid, int& c) {
        valid = mask != 0u;
        c = valid ? __builtin_clz(mask) : 0;
        mask &= ~(0x80000000u >> c);
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(dot), "=&v"(nj) : "v"(srow_a + 4u * c), "v"(nrm_a + 4u * c));
      };
      auto fetch_finish = [&](float& dot, float& nj, bool valid, int c, int& key_out) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(dot), "+v"(nj));
        // key: bucket of d = -score (v_cvt_u32_f32 truncates, maps negative rounding noise to 0 and saturates),
        // low bits = candidate index
        const unsigned fx = min(__float2uint_rz(-score(dot, nj) * scale), FXMAX);
        key_out = valid ? (int)((fx << idb) | (unsigned)(t * TJ + c)) : EMPTY;
      };
      int key_cur, key_nxt;
      fetch_issue(dotA, njA, vA, cA);
      fetch_issue(dotB, njB, vB, cB);
      fetch_finish(dotA, njA, vA, cA, key_cur);
      for (int it = 0; it < nit; it += 2) {
        fetch_issue(dotA, njA, vA, cA);                         // entry it+2
        insert_key_hi<KP>(L, key_cur);
        fetch_finish(dotB, njB, vB, cB, key_nxt);               // entry it+1
        insert_key_lo<KP>(L, key_cur);
        if (it + 1 < nit) {
          fetch_issue(dotB, njB, vB, cB);                       // entry it+3
          insert_key_hi<KP>(L, key_nxt);
          fetch_finish(dotA, njA, vA, cA, key_cur);             // entry it+2
          insert_key_lo<KP>(L, key_nxt);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dotA), "+v"(njA), "+v"(dotB), "+v"(njB));
      // a candidate whose key is below L[KP-1] has floor(d*scale) <= its bucket, so d < (bucket + 1) / scale: the
      // margin (+3, 2e-6 relative) covers the roundings of d*scale, of scale and of this product (buckets < 2^21)
      const int lk = L[KP - 1];
      thr = lk == EMPTY ? -FLT_MAX : -((float)((lk >> idb) + 3) * inv_scale * 1.000002f);
      __syncthreads();
    }

    // ---- result.  Buckets (key >> idb) order the candidates as their exact scores do, except inside a bucket.
    bool amb = false;
#pragma unroll
    for (int u = 0; u < K; ++u) amb |= ((L[u] ^ L[u + 1]) & nidm) == 0 && L[u] != EMPTY;
    bool rescan = amb && ((L[K - 1] ^ L[KP - 1]) & nidm) == 0 && L[KP - 1] != EMPTY;
    if (force >= 1) amb = true;
    if (force >= 2) rescan = true;
    if (q >= N) amb = rescan = false;
    int32_t* o = idx + ((int64_t)b * N + (q < N ? q : 0)) * k;
    if (q < N && !amb) {
#pragma unroll
      for (int t = 0; t < K; ++t)
        if (t < k) o[t] = L[t] == EMPTY ? q : (L[t] & idm);      // never-filled slots point at the query
    }
    if (__ballot(amb) != 0ull) {
      // scratch: this wave's two score rows blocks (no other wave touches them; the producers are done)
      int* s_keys = reinterpret_cast<int*>(s_score + (0 * QB + wv * 64) * SROW);        // [64][KP]
      float* s_fv = s_score + (1 * QB + wv * 64) * SROW;                                 // [K][64]
#pragma unroll
      for (int t = 0; t < KP; ++t) s_keys[lane * KP + t] = L[t];
      // (a) exact re-rank of the K+2 candidates of one query at a time: lane u = candidate u
      unsigned long long todo = __ballot(amb && !rescan);
      while (todo) {
        const int ql = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int qi = q0 + wv * 64 + ql;
        const float niq = __shfl(ni, ql);
        const int key = lane < KP ? s_keys[ql * KP + lane] : EMPTY;
        const bool has = key != EMPTY;
        const int j = has ? (key & idm) : 0;
        const float* xq = xb + (int64_t)qi * ldx;
        const float* xj = xb + (int64_t)j * ldx;
        const float dj = exact_dot<CP>(xj, xq);
        const float nj = exact_norm<CP>(xj);
        const float sc = has ? __fsub_rn(__fsub_rn(-nj, __fmul_rn(-2.0f, dj)), niq) : -INFINITY;
        int rank = 0;
#pragma unroll
        for (int v = 0; v < KP; ++v) {
          const float sv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), v));
          const int jv = __builtin_amdgcn_readlane(has ? j : 0x7fffffff, v);
          rank += (sv > sc || (sv == sc && jv < j)) ? 1 : 0;
        }
        const int nhas = __popcll(__ballot(has));
        int32_t* oq = idx + ((int64_t)b * N + qi) * k;
        if (has && rank < k) oq[rank] = j;
        if (lane < k && lane >= nhas) oq[lane] = qi;
      }
      // (b) exact rescan of the whole cloud, one lane per query, lists in LDS ([slot][lane])
      if (rescan) {
        int* s_fid = s_keys;                    // the keys are no longer needed by this wave
        const float* xq = xb + (int64_t)q * ldx;
        for (int t = 0; t < K; ++t) { s_fv[t * 64 + lane] = -INFINITY; s_fid[t * 64 + lane] = q; }
        for (int j = 0; j < N; ++j) {
          const float* xj = xb + (int64_t)j * ldx;
          const float sc = __fsub_rn(__fsub_rn(-exact_norm<CP>(xj), __fmul_rn(-2.0f, exact_dot<CP>(xj, xq))), ni);
          if (sc > s_fv[(K - 1) * 64 + lane]) {           // strict: an equal score stays behind earlier entries
            int u = K - 1;
            while (u > 0 && sc > s_fv[(u - 1) * 64 + lane]) {
              s_fv[u * 64 + lane] = s_fv[(u - 1) * 64 + lane];
              s_fid[u * 64 + lane] = s_fid[(u - 1) * 64 + lane];
              --u;
            }
            s_fv[u * 64 + lane] = sc;
            s_fid[u * 64 + lane] = j;
          }
        }
        for (int t = 0; t < K; ++t)
          if (t < k) o[t] = s_fid[t * 64 + lane];
      }
    }
  }
}

template <int CP, int K, int NP, int NC>
int launch_pc_form(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  constexpr int RS = CP + 4;
  constexpr int QB = 64 * NC;
  const size_t sh = (size_t)(3 * TJ * RS + 4 * TJ + 2 * QB * SROW) * sizeof(float);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_pc_kernel<CP, K, NP, NC>, (int)sh, "sug_knn(mfma, producer/consumer)")) return rc;
  dim3 grid(sug_divup(N, QB) * B);
  const char* fe = getenv("SUG_KNN_FORCE");      // test knob: 1 = exact re-rank for every query, 2 = exact rescan
  hipLaunchKernelGGL((knn_pc_kernel<CP, K, NP, NC>), grid, dim3(64 * (NP + NC)), sh, st, x, ldx, B, N, k, idx, fe ? atoi(fe) : 0);
  SUG_LAUNCH_CHECK("sug_knn(mfma, producer/consumer)");
  return SUG_OK;
}

template <int CP, int K>
int launch_pc(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  // 256-query workgroups (NP = NC = 4, one per CU) when they fill the chip; 128-query workgroups when the 256-query grid would
  // leave half of the CUs idle -- 32 clouds of 1024 points are 128 workgroups on 256 CUs: NC = 2 selection waves with NP = 2
  // producers of 64 queries (C = 3: the matrix pipe has four MFMAs per tile to do) or NP = 4 producers of 32 queries (C >= 64:
  // a chain on every SIMD).  Measured (tools/bench_knn_pw.py, us per launch at C = 3 / 64 / 128): 32 clouds 4+4: 61 / 119 / 181,
  // 2+2: 55 / 92 / 157, 4+2 (round 6): -- / 88 / 122 (64-query workgroups of 2+1 waves: the same 88 / 123); 64 clouds 4+4: 71 / 133 / 190,
  // 2+2: 77 / 157 / 303, 4+2: -- / 135 / 235 (every workgroup stages all candidate tiles, so two per CU double that work and
  // halve the waves behind each barrier).  Same lists in every form.  SUG_KNN_PW=44|22|42 forces one.
  static const int forced = getenv("SUG_KNN_PW") ? atoi(getenv("SUG_KNN_PW")) : 0;
  const int ncu = sug_cu_count();
  const int64_t grid4 = (int64_t)B * sug_divup(N, 256);
  const bool small = 2 * grid4 <= ncu;
  const int form = (forced == 4 || forced == 44) ? 44 : (forced == 2 || forced == 22) ? 22 : forced == 42 ? 42
                   : !small ? 44 : (CP >= 64 ? 42 : 22);
  if (form == 22) return launch_pc_form<CP, K, 2, 2>(x, ldx, B, N, k, idx, st);
  if (form == 42) return launch_pc_form<CP, K, 4, 2>(x, ldx, B, N, k, idx, st);
  return launch_pc_form<CP, K, 4, 4>(x, ldx, B, N, k, idx, st);
}

template <int K>
int dispatch_pc(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st) {
  if (C == 3) return launch_pc<4, K>(x, ldx, B, N, k, idx, st);
  if (C == 64) return launch_pc<64, K>(x, ldx, B, N, k, idx, st);
  return launch_pc<128, K>(x, ldx, B, N, k, idx, st);
}

}  // namespace

// Returns 1 if the producer / consumer MFMA kernel handles this shape / alignment (else sug_knn uses the scalar kernel).
int sug_knn_pc_supported(const float* x, int64_t ldx, int C, int k) {
  if (k > 20) return 0;                 // K + 2 key registers per lane: larger k uses the scalar kernel
  if (C == 3) return 1;
  if (C != 64 && C != 128) return 0;
  return ((uintptr_t)x % 16 == 0) && (ldx % 4 == 0);
}

int sug_knn_pc(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st) {
  if (k <= 16) return dispatch_pc<16>(x, ldx, B, N, C, k, idx, st);
  return dispatch_pc<20>(x, ldx, B, N, C, k, idx, st);
}
