// Farthest point sampling, ball query, kNN-of-few-queries and 3-NN.
// Reference semantics: model/point_utils.py:5-26, :86-109, :134-165 and
// model/pointnet2_utils.py:60-104 (the torch path, not the dead CUDA extension).
#include "common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------
// FPS: one workgroup per cloud; every thread keeps PPT points and their running
// min-distance in registers, the cloud also sits in LDS so the winner's coordinates
// are a broadcast read.  One barrier per round (double-buffered wave slots).
// arg-max rule = torch.max: greatest value, lowest index among equals.
// ---------------------------------------------------------------------------
template <int PPT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float* __restrict__ xyz,
                                                    const int32_t* __restrict__ start, int N,
                                                    int npoint, int32_t* __restrict__ out) {
  extern __shared__ float s_xyz[];  // N*3
  constexpr int NW = BLOCK / WAVE;
  __shared__ float s_v[2][NW];
  __shared__ int s_i[2][NW];
  const int b = blockIdx.x;
  const float* pb = xyz + (int64_t)b * N * 3;
  for (int e = threadIdx.x; e < N * 3; e += BLOCK) s_xyz[e] = pb[e];
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int j = p * BLOCK + threadIdx.x;
    const bool ok = j < N;
    px[p] = ok ? pb[j * 3 + 0] : 0.f;
    py[p] = ok ? pb[j * 3 + 1] : 0.f;
    pz[p] = ok ? pb[j * 3 + 2] : 0.f;
    md[p] = ok ? 1e10f : -INFINITY;
  }
  int cur = start[b];
  cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  for (int i = 0; i < npoint; ++i) {
    if (threadIdx.x == 0) out[(int64_t)b * npoint + i] = cur;
    if (i == npoint - 1) break;
    const float cx = s_xyz[cur * 3 + 0], cy = s_xyz[cur * 3 + 1], cz = s_xyz[cur * 3 + 2];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float dx = __fsub_rn(px[p], cx), dy = __fsub_rn(py[p], cy), dz = __fsub_rn(pz[p], cz);
      const float d = sq3(dx, dy, dz);
      const float m = (d < md[p]) ? d : md[p];
      md[p] = m;
      if (m > bv) {  // ascending index within the thread: strict > keeps the lowest
        bv = m;
        bi = p * BLOCK + threadIdx.x;
      }
    }
    // wave arg-max: greatest value, lowest index among equals (two DPP reductions: the value, then the index)
    const float wm = wave_max_f(bv);
    {
      // the lanes that hold the maximum: almost always one -> its index by a scalar read; several (equal
      // distances) -> the lowest index among them by the second reduction
      const unsigned long long mm = __ballot(bv == wm);
      if (__popcll(mm) == 1) bi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(mm));
      else bi = wave_min_i(bv == wm ? bi : 0x7fffffff);
    }
    bv = wm;
    if constexpr (NW > 1) {
      const int par = i & 1;
      if (lane == 0) {
        s_v[par][wv] = bv;
        s_i[par][wv] = bi;
      }
      __syncthreads();
      bv = s_v[par][0];
      bi = s_i[par][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float ov = s_v[par][w];
        const int oi = s_i[par][w];
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
    }
    cur = bi < N ? bi : 0;
  }
}

// ---------------------------------------------------------------------------
// Ball query: one wave per query, 64 candidates per step, ballot + prefix popcount
// gives the ascending-index order of the reference's sort-based implementation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz,
                                                         const float* __restrict__ qry, int N, int S,
                                                         float r2, int ns, int32_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (s >= S) return;  // whole wave exits together
  const int lane = threadIdx.x & (WAVE - 1);
  const float* pb = xyz + (int64_t)b * N * 3;
  const float* q = qry + ((int64_t)b * S + s) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  int32_t* o = out + ((int64_t)b * S + s) * ns;
  int cnt = 0, first = N;
  for (int j0 = 0; j0 < N && cnt < ns; j0 += WAVE) {
    const int j = j0 + lane;
    bool hit = false;
    if (j < N) {
      const float x = pb[j * 3 + 0], y = pb[j * 3 + 1], z = pb[j * 3 + 2];
      const float d = sqdist_expanded(dot3(qx, qy, qz, x, y, z), nq, sq3(x, y, z));
      hit = !(d > r2);
    }
    const unsigned long long m = __ballot(hit);
    if (m) {
      if (cnt == 0) first = j0 + __builtin_ctzll(m);
      const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (hit && pos < ns) o[pos] = j;
      cnt += __builtin_popcountll(m);
    }
  }
  if (cnt > ns) cnt = ns;
  for (int p = cnt + lane; p < ns; p += WAVE) o[p] = first;
}

// The same with the cloud staged in LDS once per workgroup (N <= 4096) and qpw queries per wave: the 32 dependent
// 12-byte-strided global loads of a query's scan become LDS reads.
__global__ __launch_bounds__(256) void ball_query_lds_kernel(const float* __restrict__ xyz,
                                                             const float* __restrict__ qry, int N, int S,
                                                             float r2, int ns, int qpw, int32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_pts[];             // [N*3]
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const float* pb = xyz + (int64_t)b * N * 3;
  for (int i = threadIdx.x; i < 3 * N; i += 256) s_pts[i] = pb[i];
  __syncthreads();
  for (int qi = 0; qi < qpw; ++qi) {
    const int s = (blockIdx.x * (256 / WAVE) + wv) * qpw + qi;
    if (s >= S) break;
    const float* q = qry + ((int64_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float nq = sq3(qx, qy, qz);
    int32_t* o = out + ((int64_t)b * S + s) * ns;
    int cnt = 0, first = N;
    for (int j0 = 0; j0 < N && cnt < ns; j0 += WAVE) {
      const int j = j0 + lane;
      bool hit = false;
      if (j < N) {
        const float x = s_pts[j * 3 + 0], y = s_pts[j * 3 + 1], z = s_pts[j * 3 + 2];
        const float d = sqdist_expanded(dot3(qx, qy, qz, x, y, z), nq, sq3(x, y, z));
        hit = !(d > r2);
      }
      const unsigned long long m = __ballot(hit);
      if (m) {
        if (cnt == 0) first = j0 + __builtin_ctzll(m);
        const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (hit && pos < ns) o[pos] = j;
        cnt += __builtin_popcountll(m);
      }
    }
    if (cnt > ns) cnt = ns;
    for (int p = cnt + lane; p < ns; p += WAVE) o[p] = first;
  }
}

// ---------------------------------------------------------------------------
// k nearest candidates for few queries (full-sort semantics: ascending distance, lower index first among equal
// distances): one wave per query, the N distances live in registers (NPL per lane) as order-preserving integer keys.
//   1. every lane keeps its R smallest keys (R = 1 / 2 / 4 for k <= 16 / 32 / 64): the k-th smallest key T of this
//      pool of 64 R values is an upper bound of the k-th smallest distance that is almost always tight;
//   2. T by bisection over the key space with wave ballots (32 steps, R compares each);
//   3. the candidates with key <= T (>= k of them, a few more at most) are compacted into LDS through ballot
//      prefixes and ranked by counting: rank = number of candidates with a smaller (key, index); ranks < k are the
//      answer in order.
// ~1.5 k instructions per query at N = 1024, k = 64 against ~6 k for k rounds of a wave-wide arg-min, which remain
// as the fallback when more than 128 candidates tie below T.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned dist_key(float d) {
  const unsigned u = __float_as_uint(d + 0.0f);                 // -0 -> +0: equal distances compare equal
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_dist(unsigned key) {
  return __uint_as_float(key ^ ((key >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <int NPL, bool DIRECT>
__global__ __launch_bounds__(256) void knn_query_kernel(const float* __restrict__ xyz,
                                                        const float* __restrict__ qry, int N, int S,
                                                        int k, int qpw, int32_t* __restrict__ idx_out,
                                                        float* __restrict__ dist_out) {
  // The cloud is staged in LDS once per workgroup (coalesced) and serves 4 * qpw queries: the per-lane reads of
  // 12-byte points straight from global memory were the bottleneck (6 cache lines per wave load).
  constexpr int CAP = 128;
  extern __shared__ __attribute__((aligned(16))) float s_xyz[];             // [N*3], then the candidate buffers
  const int b = blockIdx.y;
  const int wv = threadIdx.x / WAVE;
  const int lane = threadIdx.x & (WAVE - 1);
  const float* pb = xyz + (int64_t)b * N * 3;
  for (int i = threadIdx.x; i < 3 * N; i += 256) s_xyz[i] = pb[i];
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(s_xyz + ((3 * N + 3) & ~3)) + wv * CAP;
  __syncthreads();
  for (int qi = 0; qi < qpw; ++qi) {
  const int s = (blockIdx.x * (256 / WAVE) + wv) * qpw + qi;
  if (s >= S) break;
  const float* q = qry + ((int64_t)b * S + s) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  unsigned key[NPL];
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const int j = p * WAVE + lane;
    if (j < N) {
      const float x = s_xyz[j * 3 + 0], y = s_xyz[j * 3 + 1], z = s_xyz[j * 3 + 2];
      // DIRECT: sum((q - p)^2) as square_distance_Ptrans (point_utils.py:43-57 / PTran_utils.py:22-36);
      // otherwise the expanded form of square_distance (point_utils.py:112-131)
      key[p] = dist_key(DIRECT ? sq3(qx - x, qy - y, qz - z) : sqdist_expanded(dot3(qx, qy, qz, x, y, z), nq, sq3(x, y, z)));
    } else {
      key[p] = 0xFFFFFFFFu;                                     // above every real key (+inf is 0xFF800000)
    }
  }
  int32_t* io = idx_out + ((int64_t)b * S + s) * k;
  float* dO = dist_out ? dist_out + ((int64_t)b * S + s) * k : nullptr;

  // 1. the lane's R smallest keys, ascending
  const int R = k <= 16 ? 1 : (k <= 32 ? 2 : 4);
  unsigned a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu, a3 = 0xFFFFFFFFu;
  if (R == 1) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) a0 = min(a0, key[p]);
  } else if (R == 2) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      const unsigned x = max(a0, key[p]);
      a0 = min(a0, key[p]);
      a1 = min(a1, x);
    }
  } else {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      unsigned x = key[p], t;
      t = min(a0, x); x = max(a0, x); a0 = t;
      t = min(a1, x); x = max(a1, x); a1 = t;
      t = min(a2, x); x = max(a2, x); a2 = t;
      a3 = min(a3, x);
    }
  }
  // 2. T = the k-th smallest key of the pool: smallest T with count(pool <= T) >= k
  // (bounds: at least 64 >= k keys lie at or below the largest lane minimum, none below the smallest)
  unsigned lo = (unsigned)wave_min_i((int)(a0 ^ 0x80000000u)) ^ 0x80000000u;
  unsigned hi = ~((unsigned)wave_min_i((int)(~a0 ^ 0x80000000u)) ^ 0x80000000u);
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int c = __popcll(__ballot(a0 <= mid));
    if (R >= 2) c += __popcll(__ballot(a1 <= mid));
    if (R >= 4) c += __popcll(__ballot(a2 <= mid)) + __popcll(__ballot(a3 <= mid));
    if (c >= k) hi = mid;
    else lo = mid + 1u;
  }
  const unsigned T = hi;
  // 3. compaction of {key <= T} (rows past N carry 0xFFFFFFFF and T is a real key: they never pass)
  int base = 0;
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    const bool hit = key[p] <= T && (p * WAVE + lane) < N;
    const unsigned long long m = __ballot(hit);
    if (m) {                                                     // most registers hold no candidate at all
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && pos < CAP) cand[pos] = ((unsigned long long)key[p] << 32) | (unsigned)(p * WAVE + lane);
      base += __popcll(m);
    }
  }
  if (base <= CAP) {
    // ranks by counting (one wave: LDS writes above are visible after the wait the compiler places before the reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long none = ~0ull;
    const unsigned long long c0 = lane < base ? cand[lane] : none;
    const unsigned long long c1 = lane + WAVE < base ? cand[lane + WAVE] : none;
    int r0 = 0, r1 = 0;
    for (int j = 0; j < base; ++j) {
      const unsigned long long cj = cand[j];
      r0 += cj < c0 ? 1 : 0;
      r1 += cj < c1 ? 1 : 0;
    }
    if (c0 != none && r0 < k) {
      io[r0] = (int32_t)(unsigned)c0;
      if (dO) dO[r0] = key_dist((unsigned)(c0 >> 32));
    }
    if (c1 != none && r1 < k) {
      io[r1] = (int32_t)(unsigned)c1;
      if (dO) dO[r1] = key_dist((unsigned)(c1 >> 32));
    }
    __builtin_amdgcn_wave_barrier();                           // the buffer is reused by the wave's next query
    continue;
  }
  // fallback (more than CAP candidates tie at or below T): k rounds of a wave-wide arg-min
  int res_i = 0;
  unsigned res_k = 0u;
  for (int t = 0; t < k; ++t) {
    unsigned lv = 0xFFFFFFFFu;
    int li = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      if (key[p] < lv) {
        lv = key[p];
        li = p * WAVE + lane;
      }
    }
    // smallest key, lowest index among equals (DPP reductions on the sign-flipped key: signed min)
    const unsigned bv = (unsigned)wave_min_i((int)(lv ^ 0x80000000u)) ^ 0x80000000u;
    const int bi = wave_min_i(lv == bv ? li : 0x7fffffff);
    if (lane == (t & (WAVE - 1))) {
      res_i = bi;
      res_k = bv;
    }
    if ((bi & (WAVE - 1)) == lane) {
      const int slot = bi / WAVE;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        if (p == slot) key[p] = 0xFFFFFFFFu;
    }
  }
  if (lane < k) {
    io[lane] = res_i;
    if (dO) dO[lane] = key_dist(res_k);
  }
  }
}

// ---------------------------------------------------------------------------
// 3 nearest of S candidates (LDS-resident) for every point of the dense cloud.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ qry,
                                                       const float* __restrict__ cand, int N, int S,
                                                       int32_t* __restrict__ idx3,
                                                       float* __restrict__ dist3) {
  extern __shared__ float s_c[];  // S*4: x,y,z,|c|^2
  const int b = blockIdx.y;
  const float* cb = cand + (int64_t)b * S * 3;
  for (int j = threadIdx.x; j < S; j += 256) {
    const float x = cb[j * 3 + 0], y = cb[j * 3 + 1], z = cb[j * 3 + 2];
    s_c[j * 4 + 0] = x;
    s_c[j * 4 + 1] = y;
    s_c[j * 4 + 2] = z;
    s_c[j * 4 + 3] = sq3(x, y, z);
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* q = qry + ((int64_t)b * N + n) * 3;
  const float qx = q[0], qy = q[1], qz = q[2];
  const float nq = sq3(qx, qy, qz);
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int j = 0; j < S; ++j) {
    const float4 c = reinterpret_cast<const float4*>(s_c)[j];
    const float d = sqdist_expanded(dot3(qx, qy, qz, c.x, c.y, c.z), nq, c.w);
    if (d < d2) {
      if (d < d1) {
        d2 = d1; i2 = i1;
        if (d < d0) {
          d1 = d0; i1 = i0;
          d0 = d; i0 = j;
        } else {
          d1 = d; i1 = j;
        }
      } else {
        d2 = d; i2 = j;
      }
    }
  }
  const int64_t o = ((int64_t)b * N + n) * 3;
  idx3[o + 0] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
  dist3[o + 0] = d0; dist3[o + 1] = d1; dist3[o + 2] = d2;
}

template <int PPT, int BLOCK>
int launch_fps(const float* xyz, const int32_t* start, int B, int N, int npoint, int32_t* out,
               hipStream_t st) {
  size_t sh = (size_t)N * 3 * sizeof(float);
  hipLaunchKernelGGL((fps_kernel<PPT, BLOCK>), dim3(B), dim3(BLOCK), sh, st, xyz, start, N, npoint, out);
  SUG_LAUNCH_CHECK("sug_fps");
  return SUG_OK;
}

template <int BLOCK>
int dispatch_fps(const float* xyz, const int32_t* start, int B, int N, int npoint, int32_t* out, hipStream_t st) {
  const int ppt = sug_divup(N, BLOCK);
  if (ppt <= 1) return launch_fps<1, BLOCK>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 2) return launch_fps<2, BLOCK>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 4) return launch_fps<4, BLOCK>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 8) return launch_fps<8, BLOCK>(xyz, start, B, N, npoint, out, st);
  if (ppt <= 16) return launch_fps<16, BLOCK>(xyz, start, B, N, npoint, out, st);
  return launch_fps<32, BLOCK>(xyz, start, B, N, npoint, out, st);
}

}  // namespace

extern "C" int sug_fps(const float* xyz, const int32_t* start, int B, int N, int npoint,
                       int32_t* out, void* stream) {
  SUG_REQUIRE(xyz && start && out, "sug_fps: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && npoint > 0, "sug_fps: bad shape B=%d N=%d npoint=%d", B, N, npoint);
  SUG_REQUIRE(N <= 8192, "sug_fps: N=%d > 8192", N);
  hipStream_t st = (hipStream_t)stream;
  // every round is one dependent chain (winner's coordinates -> distances -> wave arg-max -> cross-wave exchange):
  // ~0.33 us with one wave, ~0.45 with two, ~0.57 with four, plus ~0.02 us per point a thread holds -- fewer
  // waves win for small clouds (measured, tools/bench_fps.py)
  static const int forced = getenv("SUG_FPS_BLOCK") ? atoi(getenv("SUG_FPS_BLOCK")) : 0;
  const int blk = forced ? forced : (N <= 768 ? 64 : (N <= 1536 ? 128 : 256));
  if (blk == 64 && N <= 2048) return dispatch_fps<64>(xyz, start, B, N, npoint, out, st);
  if (blk == 128 && N <= 4096) return dispatch_fps<128>(xyz, start, B, N, npoint, out, st);
  return dispatch_fps<256>(xyz, start, B, N, npoint, out, st);
}

extern "C" int sug_ball_query(const float* xyz, const float* query, int B, int N, int S, float r2,
                              int nsample, int32_t* out, void* stream) {
  SUG_REQUIRE(xyz && query && out, "sug_ball_query: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0 && nsample > 0, "sug_ball_query: bad shape");
  SUG_REQUIRE(B <= 65535, "sug_ball_query: B too large");
  if (N <= 4096) {
    int qpw = 1;                  // plenty of workgroups first (>= 8 per CU), then up to 4 queries per staging
    while (qpw < 4 && (int64_t)B * sug_divup(S, 4 * 2 * qpw) >= 2048) qpw *= 2;
    dim3 grid(sug_divup(S, (256 / WAVE) * qpw), B);
    hipLaunchKernelGGL(ball_query_lds_kernel, grid, dim3(256), (size_t)3 * N * sizeof(float), (hipStream_t)stream, xyz,
                       query, N, S, r2, nsample, qpw, out);
  } else {
    dim3 grid(sug_divup(S, 256 / WAVE), B);
    hipLaunchKernelGGL(ball_query_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, query, N, S, r2,
                       nsample, out);
  }
  SUG_LAUNCH_CHECK("sug_ball_query");
  return SUG_OK;
}

template <bool DIRECT>
static int launch_knn_query(const float* xyz, const float* query, int B, int N, int S, int k, int32_t* idx_out,
                            float* dist_out, void* stream) {
  SUG_REQUIRE(xyz && query && idx_out, "sug_knn_query: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S > 0, "sug_knn_query: bad shape");
  SUG_REQUIRE(k >= 1 && k <= 64 && k <= N, "sug_knn_query: need 1 <= k <= min(64,N), got %d", k);
  SUG_REQUIRE(N <= 4096, "sug_knn_query: N=%d > 4096", N);
  SUG_REQUIRE(B <= 65535, "sug_knn_query: B too large");
  // queries per wave: plenty of workgroups first (>= 8 per CU), then up to 4 queries per staging of the cloud
  int qpw = 1;
  while (qpw < 4 && (int64_t)B * sug_divup(S, 4 * 2 * qpw) >= 2048) qpw *= 2;
  dim3 grid(sug_divup(S, (256 / WAVE) * qpw), B);
  hipStream_t st = (hipStream_t)stream;
  const int npl = sug_divup(N, WAVE);
  const size_t sh = (size_t)((3 * N + 3) & ~3) * sizeof(float) + (256 / WAVE) * 128 * sizeof(unsigned long long);
  if (npl <= 4)
    hipLaunchKernelGGL((knn_query_kernel<4, DIRECT>), grid, dim3(256), sh, st, xyz, query, N, S, k, qpw, idx_out, dist_out);
  else if (npl <= 8)
    hipLaunchKernelGGL((knn_query_kernel<8, DIRECT>), grid, dim3(256), sh, st, xyz, query, N, S, k, qpw, idx_out, dist_out);
  else if (npl <= 16)
    hipLaunchKernelGGL((knn_query_kernel<16, DIRECT>), grid, dim3(256), sh, st, xyz, query, N, S, k, qpw, idx_out, dist_out);
  else if (npl <= 32)
    hipLaunchKernelGGL((knn_query_kernel<32, DIRECT>), grid, dim3(256), sh, st, xyz, query, N, S, k, qpw, idx_out, dist_out);
  else
    hipLaunchKernelGGL((knn_query_kernel<64, DIRECT>), grid, dim3(256), sh, st, xyz, query, N, S, k, qpw, idx_out, dist_out);
  SUG_LAUNCH_CHECK("sug_knn_query");
  return SUG_OK;
}

extern "C" int sug_knn_query(const float* xyz, const float* query, int B, int N, int S, int k,
                             int32_t* idx_out, float* dist_out, void* stream) {
  return launch_knn_query<false>(xyz, query, B, N, S, k, idx_out, dist_out, stream);
}

extern "C" int sug_knn_query_direct(const float* xyz, const float* query, int B, int N, int S, int k,
                                    int32_t* idx_out, float* dist_out, void* stream) {
  return launch_knn_query<true>(xyz, query, B, N, S, k, idx_out, dist_out, stream);
}

extern "C" int sug_three_nn(const float* query, const float* cand, int B, int N, int S,
                            int32_t* idx3, float* dist3, void* stream) {
  SUG_REQUIRE(query && cand && idx3 && dist3, "sug_three_nn: null pointer");
  SUG_REQUIRE(B > 0 && N > 0 && S >= 3, "sug_three_nn: bad shape (S=%d must be >= 3)", S);
  SUG_REQUIRE(S <= 2048, "sug_three_nn: S=%d > 2048", S);
  SUG_REQUIRE(B <= 65535, "sug_three_nn: B too large");
  dim3 grid(sug_divup(N, 256), B);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(256), (size_t)S * 4 * sizeof(float),
                     (hipStream_t)stream, query, cand, N, S, idx3, dist3);
  SUG_LAUNCH_CHECK("sug_three_nn");
  return SUG_OK;
}
