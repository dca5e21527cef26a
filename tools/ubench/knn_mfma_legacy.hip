// kNN graph, MFMA formulation (C = 3, 64, 128): the hot index kernel of the DGCNN path.
// Replaces knn(), model/model_utils.py:178-185.
//
// The Gram products <x_i, x_j> run on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), which
// on gfx950 is bit-for-bit the ascending-k fmaf chain of the scalar kernel (knn.hip) and of
// the CPU reference's K=3 sgemm -- so indices stay bit-exact -- while the VALU is left for
// the top-k selection, which is what actually bounds this kernel.
//
// Geometry: workgroup = 4 waves = 128 queries of one cloud; a wave owns 32 queries (one
// MFMA column block).  S^T tile = candidates(32) x queries(32): lane l holds query l&31 and
// the 16 candidate rows (reg&3)+8*(reg>>2)+4*(l>>5), i.e. every query is served by TWO lanes
// (l, l+32), each scanning half of the candidates in ascending index order with a private
// sorted top-K in registers; the halves are merged at the end.
// Candidate tiles (32 rows) stream global -> registers -> LDS (double buffered, next tile in
// flight under the MFMA chain).  LDS rows are stored de-interleaved [even feats | odd feats]
// so one ds_read_b128 feeds the A operand of four consecutive k-steps (lanes 0-31 supply
// k=2s, lanes 32-63 k=2s+1); row stride C+4 floats keeps the b128 reads conflict-free.
// Selection: one sweep with a lazily refreshed threshold -- see knn_mfma_kernel.
//
// Algorithmic bytes 4*C*N + 4*N*k per cloud; FLOPs N^2*(2C+3): compute bound.
#include "common.h"
#include "mfma_tile.h"
#include <type_traits>
#include <stdlib.h>

#ifdef SUG_KNN_STAMP      // diagnostic build only (tools/bench_knn.py): per-phase cycle stamps of block 0
__device__ unsigned long long g_knn_stamp[16];
extern "C" int sug_debug_read_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_knn_stamp), sizeof(g_knn_stamp));
}
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_knn_stamp[i] = clock64(); } while (0)
#define ACC_BEGIN() const unsigned long long _t0 = clock64()
#define ACC_END(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_knn_stamp[i] += clock64() - _t0; } while (0)
#else
#define STAMP(i)
#define ACC_BEGIN()
#define ACC_END(i)
#endif

namespace {

using namespace sug_tile;      // f32x16, TJ, TileRegs, tile_load, tile_store (mfma_tile.h)

template <int K>
__device__ __forceinline__ void insert_desc(float (&v)[K], int (&id)[K], float s, int j) {
  // caller guarantees s > v[K-1]; earlier (lower-index) entries win ties
#pragma unroll
  for (int t = K - 1; t > 0; --t) {
    const bool up = s > v[t - 1];
    const bool here = s > v[t];
    v[t] = up ? v[t - 1] : (here ? s : v[t]);
    id[t] = up ? id[t - 1] : (here ? j : id[t]);
  }
  if (s > v[0]) {
    v[0] = s;
    id[0] = j;
  }
}

__device__ __forceinline__ bool better(float s, int j, float v, int i) { return s > v || (s == v && j < i); }

template <int K>
__device__ __forceinline__ void insert_desc_tie(float (&v)[K], int (&id)[K], float s, int j) {
#pragma unroll
  for (int t = K - 1; t > 0; --t) {
    const bool up = better(s, j, v[t - 1], id[t - 1]);
    const bool here = better(s, j, v[t], id[t]);
    const float nv = up ? v[t - 1] : (here ? s : v[t]);
    const int ni = up ? id[t - 1] : (here ? j : id[t]);
    v[t] = nv;
    id[t] = ni;
  }
  if (better(s, j, v[0], id[0])) {
    v[0] = s;
    id[0] = j;
  }
}

// S^T tile of one candidate tile against the wave's 32 queries (k-ordered fp32 fma chain).
template <int CP>
__device__ __forceinline__ f32x16 score_tile(const float* __restrict__ arow, const float (&bq)[CP / 2]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (CP == 4) {
    const float2 a2 = *reinterpret_cast<const float2*>(arow);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, bq[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, bq[1], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int g = 0; g < CP / 8; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bq[4 * g + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bq[4 * g + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bq[4 * g + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bq[4 * g + 3], acc, 0, 0, 0);
    }
  }
  return acc;
}

// Sortable 64-bit key of a (score, index) pair: larger key = better neighbour
// (higher score; among equal scores the lower index).
__device__ __forceinline__ unsigned long long pack_key(float s, int j) {
  unsigned int u = __float_as_uint(s);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((unsigned long long)u << 32) | (unsigned int)(0x7fffffff - j);
}

// Ring capacity per lane (entries of 8 B): the LDS budget is 160 KB per workgroup.
template <int CP>
struct RingCap { static constexpr int value = (CP == 128) ? 48 : 64; };   // multiples of 16

// One sweep over the candidates; per candidate only  score -> compare -> predicated append:
//  * a candidate whose score beats `thr` is appended, with its index, to a per-lane LDS ring by a
//    predicated store (a rejected candidate writes the dummy slot CAP).  `thr` is the lane's K-th
//    best score AS OF THE LAST REFRESH: slightly stale, never too high, so the ring always holds
//    a superset of the lane's top-K.  That is ~7 VALU ops per candidate, which hides under the
//    next tile's MFMA chain (2-4 MFMAs = 128-256 cycles per candidate slot).
//  * refresh: the K best SCORES of a lane stay sorted in registers; inserting s into a descending
//    list is new[t] = med3(old[t-1], old[t], s), one v_med3_f32 per slot.  The chain runs only
//    over the ring entries appended since the last refresh (not over every candidate), whenever
//    some lane of the workgroup has RP of them pending; then thr = v[K-1] is exact again.  About
//    K(1+ln(n/K)) + staleness ~ 100-150 entries pass per lane over a cloud instead of 512.
//  * the ring (CAP ~ 64) is compacted -- keep score > tau plus the first ties at tau, tau = the
//    exact K-th score right after a refresh -- only a few times; all waves of the workgroup
//    refresh / compact in the same iteration (a wave doing it alone would stall the others at
//    the per-tile barrier).
// The <= K survivors of a lane and of its partner lane (l ^ 32, same query, other half of the
// candidates) are then ranked by counting: position = number of better keys among the 2K.
// Same result as a sorted (score desc, index asc) scan of all N candidates.
#ifndef SUG_KNN_RP
#define SUG_KNN_RP 24      // measured 8/12/16/24 (tools/bench_knn.py): fewer, larger refreshes win
#endif
template <int CP, int K>
__global__ __launch_bounds__(256, 1) void knn_mfma_kernel(const float* __restrict__ x, int64_t ldx,
                                                          int B, int N, int k, int32_t* __restrict__ idx) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  constexpr int CAP = RingCap<CP>::value;
  static_assert(CAP >= K + 24, "ring too small");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);                 // [3][TJ][RS]
  float* s_norm = s_tile + 3 * TJ * RS;                           // [3][TJ] (+pad)
  float2* s_ring = reinterpret_cast<float2*>(s_norm + 4 * TJ);    // [CAP+1][256] (score, index bits)

  // Workgroup -> (cloud, query block).  Consecutive workgroup ids go round-robin over the 8 XCDs
  // (each with its own L2): give all query blocks of a cloud the same id % 8, so the cloud is
  // fetched through the fabric once per XCD-resident group instead of once per workgroup
  // (PMC: 140 MB -> ~algorithmic per launch at C=64, 64 clouds).
  const int nq = (N + 127) / 128;
  int b, qb;
  if ((B & 7) == 0) {
    const int grp = blockIdx.x / (8 * nq), rem = blockIdx.x % (8 * nq);
    b = grp * 8 + (rem & 7);
    qb = rem >> 3;
  } else {
    b = blockIdx.x / nq;
    qb = blockIdx.x % nq;
  }
  const float* xb = x + (int64_t)b * N * ldx;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int qj = lane & 31, h = lane >> 5;
  const int q0 = qb * 128;

  STAMP(0);
  // ---- query operands: stage the wave's 32 query rows through the tile buffers
  float bq[HALF];
  float ni = 0.f;
  {
    TileRegs<CP> tr;
    for (int w = 0; w < 4; ++w) {
      __syncthreads();
      tile_load<CP>(tr, xb, ldx, N, q0 + w * TJ);
      tile_store<CP>(tr, s_tile, s_norm, N, q0 + w * TJ);
      __syncthreads();
      if (w == wv) {
        const float* qrow = s_tile + qj * RS + h * HALF;
#pragma unroll
        for (int e = 0; e < HALF; ++e) bq[e] = qrow[e];
        ni = s_norm[qj];
      }
    }
  }
  __syncthreads();
  STAMP(1);

  const int ntile = (N + TJ - 1) / TJ;
  float v[K];
#pragma unroll
  for (int t = 0; t < K; ++t) v[t] = -INFINITY;
  int cnt = 0, done = 0;            // ring entries; how many of them the med3 chain has seen
  float thr = -INFINITY;            // v[K-1] as of the last refresh
  float2* ring = s_ring + threadIdx.x;

  auto chain = [&](float sc) {
#pragma unroll
    for (int u = K - 1; u > 0; --u) v[u] = __builtin_amdgcn_fmed3f(v[u - 1], v[u], sc);
    v[0] = fmaxf(v[0], sc);
  };
  // Bring the sorted scores up to date with the ring: entries [done, cnt) of every lane, the loop
  // running to the wave-wide maximum (padding with -inf is a no-op for the chain).
  auto refresh = [&]() {
    auto fetch = [&](int i, float (&sc)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = i + u < CAP ? i + u : CAP;
        const float r = ring[e * 256].x;
        sc[u] = (i + u < cnt) ? r : -INFINITY;
      }
    };
    float nx[4];
    fetch(done, nx);
    for (int i = done; __any(i < cnt); i += 4) {
      float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) sc[u] = nx[u];
      fetch(i + 4, nx);                       // next batch in flight under this batch's chain
#pragma unroll
      for (int u = 0; u < 4; ++u) chain(sc[u]);
    }
    done = cnt;
    thr = v[K - 1];
  };

  // Compaction reads a batch of 16 entries into registers before it writes any of them back
  // (writes only go to positions <= the ones already read), so the LDS reads of a batch are in
  // flight together instead of one read -> write round trip per entry.
  static_assert(CAP % 16 == 0, "ring capacity must be a multiple of the compaction batch");
  // Exact compaction: keep entries with score > tau and the first (K - #strict) entries equal
  // to tau, in order (<= K entries remain).
  auto compact = [&]() {
    const float tau = v[K - 1];
    int ns = 0;
#pragma unroll 8
    for (int i = 0; i < CAP; ++i) ns += (i < cnt && ring[i * 256].x > tau) ? 1 : 0;
    int room = K - ns, out = 0;
    for (int i0 = 0; i0 < CAP && __any(i0 < cnt); i0 += 16) {
      float2 e[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) e[u] = ring[(i0 + u) * 256];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool live = i0 + u < cnt;
        const bool strict = live && e[u].x > tau;
        const bool tie = live && e[u].x == tau && room > 0 && e[u].x > -INFINITY;
        const bool keep = strict || tie;
        ring[(keep ? out : CAP) * 256] = e[u];
        out += keep ? 1 : 0;
        room -= tie ? 1 : 0;
      }
    }
    cnt = out;
  };
  // Cheap in-flight compaction: one pass, keep score >= tau (all ties).
  auto compact_fast = [&]() {
    const float tau = v[K - 1];
    int out = 0;
    for (int i0 = 0; i0 < CAP && __any(i0 < cnt); i0 += 16) {
      float2 e[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) e[u] = ring[(i0 + u) * 256];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool keep = (i0 + u < cnt) && (e[u].x >= tau) && (e[u].x > -INFINITY);
        ring[(keep ? out : CAP) * 256] = e[u];
        out += keep ? 1 : 0;
      }
    }
    cnt = out;
  };

#define SUG_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef SUG_KNN_ABL
#define SUG_KNN_REFRESH_MASK 3
#define SUG_KNN_COMPACT_MASK 1
#else
#define SUG_KNN_REFRESH_MASK 4
#define SUG_KNN_COMPACT_MASK 4
#endif
#ifdef SUG_KNN_STAMP
#define SUG_KNN_COUNT(need) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_knn_stamp[13] += (need) ? 1 : 0; g_knn_stamp[14] += ((need) & 1); } } while (0)
#else
#define SUG_KNN_COUNT(need)
#endif
  // One pipeline step: the MFMA chain of the NEXT tile (matrix pipe) is issued in pieces
  // between the CURRENT tile's candidates (VALU): per candidate slot c,
  //   [mfma] score + compare + ring append [mfma x (P-1)]
  // with the source order pinned by sched_barrier(0) -- left alone, hipcc emits the whole
  // dependent MFMA chain first and the in-order wave then cannot overlap anything.
  constexpr int NM = HALF;                       // k-steps (MFMAs) per tile
  constexpr int P = (CP == 4) ? 0 : NM / 16;     // MFMAs issued per candidate slot (2 or 4)
  auto step = [&](const f32x16& acc_cur, f32x16& acc_next, const float* __restrict__ arow,
                  const float* __restrict__ nrm, int jbase) {
    float nn[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 n4 = *reinterpret_cast<const float4*>(nrm + 8 * g);
      nn[4 * g + 0] = n4.x; nn[4 * g + 1] = n4.y; nn[4 * g + 2] = n4.z; nn[4 * g + 3] = n4.w;
    }
    float av[(CP == 4) ? 2 : HALF];
    if constexpr (CP == 4) {
      const float2 a2 = *reinterpret_cast<const float2*>(arow);
      av[0] = a2.x; av[1] = a2.y;
    } else {
#pragma unroll
      for (int g = 0; g < HALF / 4; ++g) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g);
        av[4 * g + 0] = a4.x; av[4 * g + 1] = a4.y; av[4 * g + 2] = a4.z; av[4 * g + 3] = a4.w;
      }
    }
    // keep the operand loads HERE: hipcc otherwise sinks the norm loads into the candidate loop,
    // where every s_waitcnt lgkmcnt(0) also drains the ring stores in flight
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(nn[r]));
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_next[r] = 0.f;
    if constexpr (CP == 4) {
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bq[0], acc_next, 0, 0, 0);
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bq[1], acc_next, 0, 0, 0);
    }
    SUG_SB();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if constexpr (P >= 1) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 0], bq[c * P + 0], acc_next, 0, 0, 0);
      SUG_SB();
      // pairwise_distance = -xx - inner - xx^T, inner = -2*dot (model_utils.py:179-181)
      const float s = __fsub_rn(__fsub_rn(-nn[c], __fmul_rn(-2.0f, acc_cur[c])), ni);
#if defined(SUG_KNN_ABL) && SUG_KNN_ABL == 1      // ablation: no selection at all
      thr = fmaxf(thr, s);
#else
      const bool enters = s > thr;
      ring[(enters ? cnt : CAP) * 256] = make_float2(s, __int_as_float(jbase + 8 * (c >> 2) + (c & 3)));
      cnt += enters ? 1 : 0;
#endif
#if defined(SUG_KNN_ABL) && SUG_KNN_ABL == 2      // ablation: append only, never refresh / compact
      cnt = cnt > CAP - 17 ? 0 : cnt;
#endif
      SUG_SB();
      if constexpr (P >= 2) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 1], bq[c * P + 1], acc_next, 0, 0, 0);
      if constexpr (P >= 3) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 2], bq[c * P + 2], acc_next, 0, 0, 0);
      if constexpr (P >= 4) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 3], bq[c * P + 3], acc_next, 0, 0, 0);
      SUG_SB();
    }
  };

  // Software pipeline over candidate tiles (3 LDS buffers, one barrier per tile):
  //   iteration t:  MFMA chain of tile t+1 (matrix pipe)  ||  selection of tile t (VALU)
  //                 registers of tile t+2 -> LDS; global loads of tiles t+3 and t+4 in flight
  //                 (two staging register sets: a load has two iterations to land).
  {
    TileRegs<CP> tra, trb;                      // tile t+2 lives in tra for even t, trb for odd t
    auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };
    auto nbuf = [&](int t) { return s_norm + (t % 3) * TJ; };
    tile_load<CP>(tra, xb, ldx, N, 0);
    tile_store<CP>(tra, tbuf(0), nbuf(0), N, 0);
    __syncthreads();
    if (ntile > 1) tile_load<CP>(trb, xb, ldx, N, TJ);
    if (ntile > 2) tile_load<CP>(tra, xb, ldx, N, 2 * TJ);
    f32x16 acc_cur = score_tile<CP>(tbuf(0) + qj * RS + h * HALF, bq);
    if (ntile > 1) tile_store<CP>(trb, tbuf(1), nbuf(1), N, TJ);
    __syncthreads();
    if (ntile > 3) tile_load<CP>(trb, xb, ldx, N, 3 * TJ);
    // (a macro, not a lambda: a lambda calling the lambdas above keeps their captured state --
    // cnt, v[], the ring pointer -- in a closure object in scratch memory)
    // per tile: (the last iteration's MFMA chain runs on a stale buffer; its result is never used)
    //   block-wide decision inside the per-tile barrier: bit 0 = a ring could overflow during the
    //   next tile (compact), bit 1 = a lane has SUG_KNN_RP entries the chain has not seen
#define SUG_KNN_BODY(T, TR) do { \
      f32x16 acc_next; \
      { ACC_BEGIN(); \
      step(acc_cur, acc_next, tbuf((T) + 1) + qj * RS + h * HALF, nbuf((T)) + 4 * h, (T) * TJ + 4 * h); \
      ACC_END(8); } \
      { ACC_BEGIN(); \
      if ((T) + 2 < ntile) tile_store<CP>(TR, tbuf((T) + 2), nbuf((T) + 2), N, ((T) + 2) * TJ); \
      ACC_END(9); } \
      ACC_BEGIN(); \
      const int need = __syncthreads_or((cnt > CAP - 16 ? 1 : 0) | (cnt - done >= SUG_KNN_RP ? 2 : 0)); \
      ACC_END(10); \
      if ((T) + 4 < ntile) tile_load<CP>(TR, xb, ldx, N, ((T) + 4) * TJ); \
      { ACC_BEGIN(); \
      if (need & SUG_KNN_REFRESH_MASK) refresh(); \
      ACC_END(11); } \
      { ACC_BEGIN(); \
      if (need & SUG_KNN_COMPACT_MASK) { \
        compact_fast(); \
        if (__syncthreads_or(cnt > CAP - 16)) compact(); /* degenerate clouds: many exact ties */ \
        done = cnt; \
      } \
      ACC_END(12); } \
      SUG_KNN_COUNT(need); \
      acc_cur = acc_next; \
    } while (0)
    for (int t = 0; t < ntile; t += 2) {
      SUG_KNN_BODY(t, tra);
      if (t + 1 < ntile) SUG_KNN_BODY(t + 1, trb);
    }
#undef SUG_KNN_BODY
  }
#undef SUG_SB
  STAMP(2);
  refresh();
  compact();
  STAMP(3);

  // ---- this lane's <= K survivors as sortable keys (0 = empty, below every real key)
  unsigned long long key[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float2 e = ring[i * 256];
    key[i] = (i < cnt) ? pack_key(e.x, __float_as_int(e.y)) : 0ull;
  }
  STAMP(4);
  // ---- rank among own + partner keys (all distinct): rank = number of better keys
  const int q = q0 + wv * TJ + qj;
  int32_t* o = idx + ((int64_t)b * N + (q < N ? q : 0)) * k;
  unsigned long long pk[K];
#pragma unroll
  for (int u = 0; u < K; ++u) pk[u] = __shfl_xor(key[u], 32);
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int rank = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) {
      rank += (key[u] > key[i]) ? 1 : 0;
      rank += (pk[u] > key[i]) ? 1 : 0;
    }
    if (key[i] != 0ull && rank < k && q < N) o[rank] = 0x7fffffff - (int)(unsigned int)(key[i] & 0xffffffffull);
  }
#if defined(SUG_KNN_ABL) && SUG_KNN_ABL == 1
  if (q < N && thr == 12345.678f) o[0] = 7;        // keeps the ablated score computation alive
#endif
  // fewer than k comparable candidates (NaN features, N < k): point the remaining slots at the
  // query itself so that downstream gathers stay in range
  int nvalid = 0;
#pragma unroll
  for (int u = 0; u < K; ++u) nvalid += (key[u] != 0ull ? 1 : 0) + (pk[u] != 0ull ? 1 : 0);
  if (h == 0 && q < N)
    for (int r = nvalid; r < k; ++r) o[r] = q;
  STAMP(5);
}


// ---------------------------------------------------------------------------------------------
// Two-pass variant (used for C = 3; selectable for C = 64): no ring, no refresh, no compaction.
//   pass 1: the lane keeps only its K best SCORES, sorted, in registers (one v_med3_f32 per slot
//           per candidate, interleaved with the next tile's MFMA chain) -> tau = exact K-th best
//           score of the lane's half of the candidates, and how many ties at tau belong to the top K;
//   pass 2: the same sweep again (bit-identical scores): a candidate with score > tau, or one of the
//           first `room` candidates with score == tau, is appended with its index to a K-slot list
//           in LDS (never more than K by construction, so no overflow handling).
// The matrix pipe does the distance GEMM twice, but it was ~25% busy; in exchange LDS drops from
// 159 KB to <= 67 KB per workgroup, two workgroups share a CU and hide each other's per-tile
// latencies (barrier, LDS round trips, tile hand-off), and all block-wide decisions disappear.
// The <= K entries of the two lanes of a query are merged by rank counting exactly as above.
// ---------------------------------------------------------------------------------------------
template <int CP, int K, int MODE>
__device__ __forceinline__ void knn_sweep(const float* __restrict__ xb, int64_t ldx, int N, int ntile,
                                          float* __restrict__ s_tile, float* __restrict__ s_norm,
                                          const float (&bq)[CP / 2], float ni, int qj, int h, float (&v)[K],
                                          float tau, int& room, int& cnt, float2* __restrict__ list) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  constexpr int P = (CP == 4) ? 0 : HALF / 16;     // MFMAs issued per candidate slot
  TileRegs<CP> tra, trb;
  auto tbuf = [&](int t) { return s_tile + (t % 3) * TJ * RS; };
  auto nbuf = [&](int t) { return s_norm + (t % 3) * TJ; };
  __syncthreads();                                  // the buffers may still be read by the previous phase
  tile_load<CP>(tra, xb, ldx, N, 0);
  tile_store<CP>(tra, tbuf(0), nbuf(0), N, 0);
  __syncthreads();
  if (ntile > 1) tile_load<CP>(trb, xb, ldx, N, TJ);
  if (ntile > 2) tile_load<CP>(tra, xb, ldx, N, 2 * TJ);
  f32x16 acc_cur = score_tile<CP>(tbuf(0) + qj * RS + h * HALF, bq);
  if (ntile > 1) tile_store<CP>(trb, tbuf(1), nbuf(1), N, TJ);
  __syncthreads();
  if (ntile > 3) tile_load<CP>(trb, xb, ldx, N, 3 * TJ);
#define SUG_SB() __builtin_amdgcn_sched_barrier(0)
#define SUG_KNN_TILE(T, TR) do { \
    const float* arow = tbuf((T) + 1) + qj * RS + h * HALF; \
    const float* nrm = nbuf((T)) + 4 * h; \
    const int jbase = (T) * TJ + 4 * h; \
    float nn[16]; \
_Pragma("unroll") \
    for (int g = 0; g < 4; ++g) { \
      const float4 n4 = *reinterpret_cast<const float4*>(nrm + 8 * g); \
      nn[4 * g + 0] = n4.x; nn[4 * g + 1] = n4.y; nn[4 * g + 2] = n4.z; nn[4 * g + 3] = n4.w; \
    } \
    float av[(CP == 4) ? 2 : HALF]; \
    if constexpr (CP == 4) { \
      const float2 a2 = *reinterpret_cast<const float2*>(arow); \
      av[0] = a2.x; av[1] = a2.y; \
    } else { \
_Pragma("unroll") \
      for (int g = 0; g < HALF / 4; ++g) { \
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * g); \
        av[4 * g + 0] = a4.x; av[4 * g + 1] = a4.y; av[4 * g + 2] = a4.z; av[4 * g + 3] = a4.w; \
      } \
    } \
_Pragma("unroll") \
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(nn[r])); \
    f32x16 acc_next; \
_Pragma("unroll") \
    for (int r = 0; r < 16; ++r) acc_next[r] = 0.f; \
    if constexpr (CP == 4) { \
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bq[0], acc_next, 0, 0, 0); \
      acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bq[1], acc_next, 0, 0, 0); \
    } \
    SUG_SB(); \
_Pragma("unroll") \
    for (int c = 0; c < 16; ++c) { \
      if constexpr (P >= 1) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 0], bq[c * P + 0], acc_next, 0, 0, 0); \
      SUG_SB(); \
      const float s = __fsub_rn(__fsub_rn(-nn[c], __fmul_rn(-2.0f, acc_cur[c])), ni); \
      if constexpr (MODE == 0) { \
_Pragma("unroll") \
        for (int u = K - 1; u >= K / 2; --u) v[u] = __builtin_amdgcn_fmed3f(v[u - 1], v[u], s); \
_Pragma("unroll") \
        for (int u = K - 1; u >= K / 2; --u) asm volatile("" : "+v"(v[u])); \
        SUG_SB(); \
        if constexpr (P >= 2) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 1], bq[c * P + 1], acc_next, 0, 0, 0); \
        if constexpr (P >= 3) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 2], bq[c * P + 2], acc_next, 0, 0, 0); \
        SUG_SB(); \
_Pragma("unroll") \
        for (int u = K / 2 - 1; u > 0; --u) v[u] = __builtin_amdgcn_fmed3f(v[u - 1], v[u], s); \
        v[0] = fmaxf(v[0], s); \
_Pragma("unroll") \
        for (int u = K / 2 - 1; u >= 0; --u) asm volatile("" : "+v"(v[u])); \
        SUG_SB(); \
        if constexpr (P >= 4) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 3], bq[c * P + 3], acc_next, 0, 0, 0); \
        SUG_SB(); \
      } else { \
        const bool tie = (s == tau) && (room > 0) && (s > -INFINITY); \
        const bool keep = (s > tau) || tie; \
        list[(keep ? cnt : K) * 256] = make_float2(s, __int_as_float(jbase + 8 * (c >> 2) + (c & 3))); \
        cnt += keep ? 1 : 0; \
        room -= tie ? 1 : 0; \
        SUG_SB(); \
        if constexpr (P >= 2) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 1], bq[c * P + 1], acc_next, 0, 0, 0); \
        if constexpr (P >= 3) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 2], bq[c * P + 2], acc_next, 0, 0, 0); \
        if constexpr (P >= 4) acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c * P + 3], bq[c * P + 3], acc_next, 0, 0, 0); \
        SUG_SB(); \
      } \
    } \
    if ((T) + 2 < ntile) tile_store<CP>(TR, tbuf((T) + 2), nbuf((T) + 2), N, ((T) + 2) * TJ); \
    __syncthreads(); \
    if ((T) + 4 < ntile) tile_load<CP>(TR, xb, ldx, N, ((T) + 4) * TJ); \
    acc_cur = acc_next; \
   \
  } while (0)
  for (int t = 0; t < ntile; t += 2) {          // (a macro body: selecting tra / trb at run time would spill them)
    SUG_KNN_TILE(t, tra);
    if (t + 1 < ntile) SUG_KNN_TILE(t + 1, trb);
  }
#undef SUG_KNN_TILE
#undef SUG_SB
}

template <int CP, int K>
__global__ __launch_bounds__(256, 2) void knn_mfma2p_kernel(const float* __restrict__ x, int64_t ldx, int B,
                                                            int N, int k, int32_t* __restrict__ idx) {
  constexpr int RS = CP + 4;
  constexpr int HALF = CP / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_tile = reinterpret_cast<float*>(smem);                 // [3][TJ][RS]
  float* s_norm = s_tile + 3 * TJ * RS;                           // [3][TJ] (+pad)
  float2* s_list = reinterpret_cast<float2*>(s_norm + 4 * TJ);    // [K+1][256] (score, index bits)

  const int nq = (N + 127) / 128;
  int b, qb;
  if ((B & 7) == 0) {                                             // XCD-aware mapping, as above
    const int grp = blockIdx.x / (8 * nq), rem = blockIdx.x % (8 * nq);
    b = grp * 8 + (rem & 7);
    qb = rem >> 3;
  } else {
    b = blockIdx.x / nq;
    qb = blockIdx.x % nq;
  }
  const float* xb = x + (int64_t)b * N * ldx;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int qj = lane & 31, h = lane >> 5;
  const int q0 = qb * 128;

  float bq[HALF];
  float ni = 0.f;
  {
    TileRegs<CP> tr;
    for (int w = 0; w < 4; ++w) {
      __syncthreads();
      tile_load<CP>(tr, xb, ldx, N, q0 + w * TJ);
      tile_store<CP>(tr, s_tile, s_norm, N, q0 + w * TJ);
      __syncthreads();
      if (w == wv) {
        const float* qrow = s_tile + qj * RS + h * HALF;
#pragma unroll
        for (int e = 0; e < HALF; ++e) bq[e] = qrow[e];
        ni = s_norm[qj];
      }
    }
  }
  const int ntile = (N + TJ - 1) / TJ;
  float v[K];
#pragma unroll
  for (int t = 0; t < K; ++t) v[t] = -INFINITY;
  float2* list = s_list + threadIdx.x;
  int cnt = 0, room = 0;

  knn_sweep<CP, K, 0>(xb, ldx, N, ntile, s_tile, s_norm, bq, ni, qj, h, v, 0.f, room, cnt, list);
  const float tau = v[K - 1];
#pragma unroll
  for (int u = 0; u < K; ++u) room += (v[u] == tau) ? 1 : 0;     // ties at tau that belong to the top K
  knn_sweep<CP, K, 1>(xb, ldx, N, ntile, s_tile, s_norm, bq, ni, qj, h, v, tau, room, cnt, list);

  unsigned long long key[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const float2 e = list[i * 256];
    key[i] = (i < cnt) ? pack_key(e.x, __float_as_int(e.y)) : 0ull;
  }
  const int q = q0 + wv * TJ + qj;
  int32_t* o = idx + ((int64_t)b * N + (q < N ? q : 0)) * k;
  unsigned long long pk[K];
#pragma unroll
  for (int u = 0; u < K; ++u) pk[u] = __shfl_xor(key[u], 32);
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int rank = 0;
#pragma unroll
    for (int u = 0; u < K; ++u) {
      rank += (key[u] > key[i]) ? 1 : 0;
      rank += (pk[u] > key[i]) ? 1 : 0;
    }
    if (key[i] != 0ull && rank < k && q < N) o[rank] = 0x7fffffff - (int)(unsigned int)(key[i] & 0xffffffffull);
  }
  int nvalid = 0;
#pragma unroll
  for (int u = 0; u < K; ++u) nvalid += (key[u] != 0ull ? 1 : 0) + (pk[u] != 0ull ? 1 : 0);
  if (h == 0 && q < N)
    for (int r = nvalid; r < k; ++r) o[r] = q;                  // NaN features / N < k: stay in range
}

template <int CP, int K>
int launch2p(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  constexpr int RS = CP + 4;
  const size_t sh = (size_t)(3 * TJ * RS + 4 * TJ) * sizeof(float) + (size_t)(K + 1) * 256 * sizeof(float2);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_mfma2p_kernel<CP, K>, (int)sh, "sug_knn(mfma, two-pass)")) return rc;
  dim3 grid(sug_divup(N, 128) * B);
  hipLaunchKernelGGL((knn_mfma2p_kernel<CP, K>), grid, dim3(256), sh, st, x, ldx, B, N, k, idx);
  SUG_LAUNCH_CHECK("sug_knn(mfma, two-pass)");
  return SUG_OK;
}

template <int CP, int K>
int launch(const float* x, int64_t ldx, int B, int N, int k, int32_t* idx, hipStream_t st) {
  constexpr int RS = CP + 4;
  const size_t sh = (size_t)(3 * TJ * RS + 4 * TJ) * sizeof(float) + (size_t)(RingCap<CP>::value + 1) * 256 * sizeof(float2);
  static SugLdsOptIn note;
  if (int rc = sug_allow_dynamic_lds(note, &knn_mfma_kernel<CP, K>, (int)sh, "sug_knn(mfma)")) return rc;
  dim3 grid(sug_divup(N, 128) * B);
  hipLaunchKernelGGL((knn_mfma_kernel<CP, K>), grid, dim3(256), sh, st, x, ldx, B, N, k, idx);
  SUG_LAUNCH_CHECK("sug_knn(mfma)");
  return SUG_OK;
}

// Which shapes take the two-pass kernel: 1 = C=3 only (default), 2 = C=3 and C=64, 0 = none.
// Measured at 64 clouds (tools/bench_knn.py): C=3 113 vs 145 us (3 workgroups per CU);
// C=64 248 vs 213 us -- the doubled MFMA work outweighs the co-residency of 2 workgroups.
#ifndef SUG_KNN_TWO_PASS
#define SUG_KNN_TWO_PASS 1
#endif

template <int K>
int dispatch(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, hipStream_t st) {
  if (C == 3) return SUG_KNN_TWO_PASS >= 1 ? launch2p<4, K>(x, ldx, B, N, k, idx, st) : launch<4, K>(x, ldx, B, N, k, idx, st);
  if (C == 64) return SUG_KNN_TWO_PASS >= 2 ? launch2p<64, K>(x, ldx, B, N, k, idx, st) : launch<64, K>(x, ldx, B, N, k, idx, st);
  return launch<128, K>(x, ldx, B, N, k, idx, st);
}

}  // namespace

#ifdef SUG_KNN_STAMP
// diagnostic: resident workgroups per CU of the two-pass kernel (tools/bench_knn.py)
extern "C" int sug_debug_knn2p_occupancy(int C) {
  int nb = -1;
  if (C == 64) {
    const size_t sh = (size_t)(3 * TJ * 68 + 4 * TJ) * sizeof(float) + (size_t)21 * 256 * sizeof(float2);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&knn_mfma2p_kernel<64, 20>), 256, sh);
  } else {
    const size_t sh = (size_t)(3 * TJ * 8 + 4 * TJ) * sizeof(float) + (size_t)21 * 256 * sizeof(float2);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&knn_mfma2p_kernel<4, 20>), 256, sh);
  }
  return nb;
}
#endif

// Round-1 kernels, kept OUT of the product library for A/B timing only (tools/bench_knn.py builds this file together with
// sug_amd/csrc/capi.cpp).  Shapes as sug_knn's MFMA path: C in {3, 64, 128}, k <= 20, 16-byte aligned rows.
extern "C" int sug_knn_legacy(const float* x, int64_t ldx, int B, int N, int C, int k, int32_t* idx, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (k > 20 || (C != 3 && C != 64 && C != 128)) return SUG_ERR_ARG;
  if (k <= 16) return dispatch<16>(x, ldx, B, N, C, k, idx, st);
  return dispatch<20>(x, ldx, B, N, C, k, idx, st);
}
