// Layer-level entry points: one C call per BatchNorm-carrying layer of the encoders, looping
// over the domain groups of the batch (source clouds, then target clouds: the reference's
// separate forward calls) and chaining the kernels of the finer-grained entry points.  Nothing
// new is computed here; the point is one host call (and no tensor slicing in the binding) per
// layer instead of 4-8.  Host-only translation unit.
#include <stdint.h>
#include "../../include/sug_amd.h"

void sug_set_error(const char* fmt, ...);
// edgeconv.hip: sug_edgeconv_bwd_scatter over `groups` BatchNorm groups in one launch (group g reads
// coef + g*coef_stride, red + g*red_stride)
// grouped forms of the rows-layer kernels (edgeconv.hip, bnpool.hip): 0 = done, 1 = layout not vectorisable (go group
// by group), < 0 = error
struct ihipStream_t;
int sug_col_stats_bn_groups(const float* y, int64_t ldy, int64_t rows, int C, int groups, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                            float* coef, float* ws, ihipStream_t* st);
int sug_affine_act_groups(const float* z, int64_t ldz, const float* coef, int64_t rows, int groups, int C, float slope,
                          float* out, int64_t ldo, ihipStream_t* st);
int sug_bwd_reduce_groups(const float* gout, int64_t ldg, const float* z, const float* coef, int64_t rows, int Co,
                          int groups, float slope, float* a, double* red, float* ws, ihipStream_t* st, float* dgb);
int sug_bn_bwd_apply_groups(const float* a, int64_t lda, int from_g, float slope, const float* y, int64_t ldy,
                            const float* coef, const double* red, int64_t rows_g, int groups, int C, float* dy,
                            int64_t lddy, ihipStream_t* st);
int sug_edgeconv_fwd_bn_act_groups(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                                   const float* beta, int B, int N, int k, int Co, int groups, float eps, float momentum,
                                   float slope, float* running_mean, float* running_var, float* z, uint8_t* arg,
                                   float* s1, float* coef, float* out, int64_t ldo, float* ws, void* stream);
int sug_edgeconv_bwd_scatter_groups(const float* a, const uint8_t* arg, const float* s1, const float* pq, int64_t ldpq,
                                    const int32_t* rev_off, const int32_t* rev_ent, const float* coef, const double* red,
                                    int B, int N, int k, int Co, int groups, int64_t coef_stride, int64_t red_stride,
                                    float* dpq, int64_t lddpq, void* stream);
#define LAYER_REQUIRE(cond, ...) do { if (!(cond)) { sug_set_error(__VA_ARGS__); return SUG_ERR_ARG; } } while (0)
#define LAYER_TRY(call) do { const int rc_ = (call); if (rc_ != SUG_OK) return rc_; } while (0)

extern "C" int sug_edgeconv_layer_fwd(const float* pq, int64_t ldpq, const int32_t* idx, const float* gamma,
                                      const float* beta, int B, int N, int k, int Co, int groups, int training,
                                      float eps, float momentum, float slope, float* running_mean,
                                      float* running_var, float* z, uint8_t* arg, float* s1, float* coef,
                                      float* out, int64_t ldo, double* stats, float* ws, void* stream) {
  LAYER_REQUIRE(groups >= 1 && B > 0 && B % groups == 0, "sug_edgeconv_layer_fwd: B=%d does not split into %d groups", B, groups);
  LAYER_REQUIRE(coef && out && z && arg && stats, "sug_edgeconv_layer_fwd: null pointer");
  const int Bg = B / groups;
  const int64_t rows = (int64_t)Bg * N;
  if (training)      // all groups in three launches when the LDS-resident kernel applies (edgeconv.hip)
    return sug_edgeconv_fwd_bn_act_groups(pq, ldpq, idx, gamma, beta, B, N, k, Co, groups, eps, momentum, slope, running_mean,
                                          running_var, z, arg, s1, coef, out, ldo, ws, stream);
  for (int g = 0; g < groups; ++g) {
    const int64_t r0 = (int64_t)g * rows;
    float* cg = coef + (int64_t)g * 5 * Co;
    LAYER_TRY(sug_edgeconv_fwd(pq + r0 * ldpq, ldpq, idx + r0 * k, gamma, Bg, N, k, Co, z + r0 * Co, arg + r0 * Co,
                               s1 ? s1 + r0 * Co : nullptr, stats, ws, stream));
    LAYER_TRY(sug_affine_act(z + r0 * Co, Co, cg, rows, Co, slope, out + r0 * ldo, ldo, stream));
  }
  return SUG_OK;
}

extern "C" int sug_edgeconv_layer_bwd(const float* gout, int64_t ldg, const float* z, const uint8_t* arg,
                                      const float* s1, const float* pq, int64_t ldpq, const int32_t* idx,
                                      const float* coef, int B, int N, int k, int Co, int groups, int training,
                                      float slope, float* a, double* red, int32_t* rev_off, int32_t* rev_ent,
                                      float* dpq, int64_t lddpq, float* ws, float* dgb, void* stream) {
  LAYER_REQUIRE(groups >= 1 && B > 0 && B % groups == 0, "sug_edgeconv_layer_bwd: B=%d does not split into %d groups", B, groups);
  LAYER_REQUIRE(red && a && rev_off && rev_ent, "sug_edgeconv_layer_bwd: null pointer");
  const int Bg = B / groups;
  const int64_t rows = (int64_t)Bg * N;
  LAYER_TRY(sug_knn_reverse(idx, B, N, k, rev_off, rev_ent, stream));
  int grouped = 1;
  if (groups > 1) {      // (the grouped reduce also writes dgb: no fold launch afterwards)
    grouped = sug_bwd_reduce_groups(gout, ldg, z, coef, rows, Co, groups, slope, a, red, ws, (ihipStream_t*)stream, dgb);
    if (grouped < 0) return grouped;
  }
  for (int g = 0; grouped != 0 && g < groups; ++g) {
    const int64_t r0 = (int64_t)g * rows;
    const float* cg = coef + (int64_t)g * 5 * Co;
    double* rg = red + (int64_t)g * 2 * Co;                  // per group: dbeta | dgamma
    LAYER_TRY(sug_edgeconv_bwd_reduce(gout + r0 * ldg, ldg, z + r0 * Co, cg, rows, Co, slope, a + r0 * Co, rg, ws, stream));
  }
  // eval mode: the statistics are constants, the scatter must see zero BN sums (red + groups*2Co: a
  // caller-zeroed spare row, shared by all groups)
  LAYER_TRY(sug_edgeconv_bwd_scatter_groups(a, arg, s1, pq, ldpq, rev_off, rev_ent, coef,
                                            training ? red : red + (int64_t)groups * 2 * Co, B, N, k, Co, groups,
                                            (int64_t)5 * Co, training ? (int64_t)2 * Co : 0, dpq, lddpq, stream));
  if (dgb && grouped != 0) LAYER_TRY(sug_fold_groups(red, groups, 2 * Co, dgb, stream));
  return SUG_OK;
}

extern "C" int sug_bn_act_rows_fwd(const float* y, int64_t ldy, int64_t rows, int C, int groups,
                                   const float* gamma, const float* beta, int training, float eps, float momentum,
                                   float slope, float* running_mean, float* running_var, float* coef, float* out,
                                   int64_t ldo, double* stats, float* ws, void* stream) {
  LAYER_REQUIRE(groups >= 1 && rows > 0 && rows % groups == 0, "sug_bn_act_rows_fwd: %lld rows do not split into %d groups",
                (long long)rows, groups);
  LAYER_REQUIRE(coef && out, "sug_bn_act_rows_fwd: null pointer");
  const int64_t rg = rows / groups;
  ihipStream_t* st = (ihipStream_t*)stream;
  if (groups > 1 && ldy == C) {                  // all groups per launch (3 launches instead of 3 per group)
    int rc = training ? sug_col_stats_bn_groups(y, ldy, rg, C, groups, gamma, beta, eps, momentum, running_mean, running_var,
                                                coef, ws, st)
                      : 0;
    if (rc < 0) return rc;
    if (rc == 0) {
      rc = sug_affine_act_groups(y, ldy, coef, rg, groups, C, slope, out, ldo, st);
      if (rc <= 0) return rc;
      // statistics done, activation not vectorisable: finish group by group
      for (int g = 0; g < groups; ++g)
        LAYER_TRY(sug_affine_act(y + g * rg * ldy, ldy, coef + (int64_t)g * 5 * C, rg, C, slope, out + g * rg * ldo, ldo, stream));
      return SUG_OK;
    }
  }
  for (int g = 0; g < groups; ++g) {
    float* cg = coef + (int64_t)g * 5 * C;
    if (training)
      LAYER_TRY(sug_col_stats_bn(y + g * rg * ldy, ldy, rg, C, gamma, beta, eps, momentum, running_mean, running_var, cg,
                                 ws, stream));
    LAYER_TRY(sug_affine_act(y + g * rg * ldy, ldy, cg, rg, C, slope, out + g * rg * ldo, ldo, stream));
  }
  return SUG_OK;
}

extern "C" int sug_col_stats_bn_grouped(const float* y, int64_t ldy, int64_t rows, int C, int groups, const float* gamma,
                                        const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* coef, float* ws, void* stream) {
  LAYER_REQUIRE(groups >= 1 && rows > 0 && rows % groups == 0, "sug_col_stats_bn_grouped: %lld rows do not split into %d groups",
                (long long)rows, groups);
  LAYER_REQUIRE(y && gamma && beta && coef && ws, "sug_col_stats_bn_grouped: null pointer");
  const int64_t rg = rows / groups;
  if (groups > 1 && ldy == C) {                  // every group's statistics in one launch pair
    const int rc = sug_col_stats_bn_groups(y, ldy, rg, C, groups, gamma, beta, eps, momentum, running_mean, running_var, coef, ws,
                                           (ihipStream_t*)stream);
    if (rc <= 0) return rc;
  }
  for (int g = 0; g < groups; ++g)
    LAYER_TRY(sug_col_stats_bn(y + g * rg * ldy, ldy, rg, C, gamma, beta, eps, momentum, running_mean, running_var,
                               coef + (int64_t)g * 5 * C, ws, stream));
  return SUG_OK;
}

extern "C" int sug_bn_act_rows_bwd(const float* gout, int64_t ldg, const float* y, int64_t ldy, const float* coef,
                                   int64_t rows, int C, int groups, int training, float slope, float* a,
                                   double* red, float* dy, float* ws, float* dgb, void* stream) {
  LAYER_REQUIRE(groups >= 1 && rows > 0 && rows % groups == 0, "sug_bn_act_rows_bwd: %lld rows do not split into %d groups",
                (long long)rows, groups);
  LAYER_REQUIRE(ldy == C, "sug_bn_act_rows_bwd: y must be dense");
  const int64_t rg = rows / groups;
  ihipStream_t* st = (ihipStream_t*)stream;
  // train mode, vectorisable layout: sums without the [rows, C] intermediate a = scale*G, the apply kernel forms it
  // from gout and y (5 passes over the layer instead of 6), all groups per launch
  if (training && ldg % 4 == 0 && C % 4 == 0 && ((uintptr_t)gout % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
      ((uintptr_t)dy % 16) == 0 && ((uintptr_t)coef % 16) == 0) {
    int rc = sug_bwd_reduce_groups(gout, ldg, y, coef, rg, C, groups, slope, nullptr, red, ws, st, dgb);
    if (rc < 0) return rc;
    if (rc == 0) {
      rc = sug_bn_bwd_apply_groups(gout, ldg, 1, slope, y, C, coef, red, rg, groups, C, dy, C, st);
      if (rc != 0) return rc < 0 ? rc : SUG_ERR_ARG;
      return SUG_OK;
    }
  }
  bool reduced = false;
  if (groups > 1) {
    const int rc = sug_bwd_reduce_groups(gout, ldg, y, coef, rg, C, groups, slope, a, red, ws, st, dgb);
    if (rc < 0) return rc;
    reduced = rc == 0;
  }
  bool applied = false;
  if (reduced && training) {
    const int rc = sug_bn_bwd_apply_groups(a, C, 0, slope, y, C, coef, red, rg, groups, C, dy, C, st);
    if (rc < 0) return rc;
    applied = rc == 0;
  }
  for (int g = 0; g < groups; ++g) {
    const float* cg = coef + (int64_t)g * 5 * C;
    double* rd = red + (int64_t)g * 2 * C;
    if (!reduced)
      LAYER_TRY(sug_edgeconv_bwd_reduce(gout + g * rg * ldg, ldg, y + g * rg * C, cg, rg, C, slope, a + g * rg * C, rd, ws,
                                        stream));
    if (training && !applied)
      LAYER_TRY(sug_bn_bwd_apply(a + g * rg * C, y + g * rg * C, C, cg, rd, rg, C, dy + g * rg * C, C, stream));
  }
  if (dgb && !reduced) LAYER_TRY(sug_fold_groups(red, groups, 2 * C, dgb, stream));
  return SUG_OK;
}

extern "C" int sug_bn_act_pool_layer_fwd(const float* y, int64_t ldy, int B, int N, int C, int groups,
                                         const float* gamma, const float* beta, int training, float eps,
                                         float momentum, float slope, float* running_mean, float* running_var,
                                         float* coef, float* out_max, float* out_mean, int64_t ld_pool, int32_t* arg, double* stats,
                                         float* ws_stats, float* ws_pool, void* stream) {
  LAYER_REQUIRE(groups >= 1 && B > 0 && B % groups == 0, "sug_bn_act_pool_layer_fwd: B=%d does not split into %d groups", B, groups);
  const int Bg = B / groups;
  const int64_t rg = (int64_t)Bg * N;
  for (int g = 0; g < groups; ++g) {
    float* cg = coef + (int64_t)g * 5 * C;
    const float* yg = y + g * rg * ldy;
    if (training)
      LAYER_TRY(sug_col_stats_bn(yg, ldy, rg, C, gamma, beta, eps, momentum, running_mean, running_var, cg, ws_stats,
                                 stream));
    LAYER_TRY(sug_bn_act_pool_fwd(yg, ldy, cg, Bg, N, C, slope, out_max + (int64_t)g * Bg * ld_pool, out_mean + (int64_t)g * Bg * ld_pool,
                                  ld_pool, arg + (int64_t)g * Bg * C, ws_pool, stream));
  }
  return SUG_OK;
}

extern "C" int sug_bn_act_pool_layer_bwd(const float* y, int64_t ldy, const float* coef, const float* gmax,
                                         const float* gmean, int64_t ld_pool, const int32_t* arg, int B, int N, int C, int groups,
                                         float slope, int training, double* red, float* ws, float* dy, int64_t lddy,
                                         float* dgb, void* stream) {
  LAYER_REQUIRE(groups >= 1 && B > 0 && B % groups == 0, "sug_bn_act_pool_layer_bwd: B=%d does not split into %d groups", B, groups);
  const int Bg = B / groups;
  const int64_t rg = (int64_t)Bg * N;
  for (int g = 0; g < groups; ++g)
    LAYER_TRY(sug_bn_act_pool_bwd(y + g * rg * ldy, ldy, coef + (int64_t)g * 5 * C, gmax + (int64_t)g * Bg * ld_pool,
                                  gmean + (int64_t)g * Bg * ld_pool, ld_pool, arg + (int64_t)g * Bg * C, Bg, N, C, slope, training,
                                  red + (int64_t)g * 2 * C, ws, dy + g * rg * lddy, lddy, stream));
  if (dgb) LAYER_TRY(sug_fold_groups(red, groups, 2 * C, dgb, stream));
  return SUG_OK;
}
