/*
 * njode_producer.h -- C ABI of the GPU-side batch producer in libnjode_hip.so
 * (SURVEY.md section 8 row f1): synthetic SDE datasets resident in HBM and the
 * CSR-by-time collate that turns a set of paths into the arrays NJODE.forward consumes.
 *
 * Reference interfaces replaced (all Python, nothing native exists to mirror):
 *   njode_generate_paths       stock_model.py:356-375 (BlackScholes.generate_paths),
 *                              :397-418 (OrnsteinUhlenbeck), :181-221 (Heston)
 *   njode_sample_observations  data_utils.py:73-81 (observation mask of create_dataset)
 *   njode_collate_count/_fill  data_utils.py:278-316 (custom_collate_fn) and
 *                              :352-416 (CustomCollateFnGen, func_appl_X = power-k)
 *
 * Conventions are those of njode_hip.h (device pointers, caller-owned buffers, caller's
 * stream, int return code + njode_last_error()).
 *
 * Dataset layout in HBM ("time-major"): paths f64 [S+1][dim][N], observed u8 [S+1][N].
 * A time slice of all paths is contiguous, so generation (one thread per path walks the grid)
 * writes, and the collate (one workgroup per grid time scans the batch) reads, with unit
 * stride across lanes.  Values stay float64 like the reference's dataset; the collate casts
 * to fp32 at the same place the reference does (data_utils.py:291,314).
 */
#ifndef NJODE_PRODUCER_H
#define NJODE_PRODUCER_H

#include "njode_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NJODE_SDE_BLACK_SCHOLES 0
#define NJODE_SDE_ORNSTEIN_UHLENBECK 1
#define NJODE_SDE_HESTON 2

/* hyper-parameters of stock_model.StockModel (data_utils.hyperparam_default:25-31) */
typedef struct NjodeSde {
  int32_t model;    /* NJODE_SDE_*                                              */
  int32_t n_paths;  /* N                                                        */
  int32_t dim;      /* dimensions = np.size(S0)                                 */
  int32_t n_steps;  /* S  (grid has S + 1 points)                               */
  int32_t has_sine; /* sine_coeff given: periodic_coeff(t) = 1 + sin(coeff * t) */
  int32_t reserved;
  double drift, volatility, mean, speed, correlation, S0, maturity, sine_coeff;
} NjodeSde;

/* Philox4x32-10 on `n` (counter, key) pairs: ctr [n][4], key [n][2] -> out [n][4].
 * Exposed so the generator's random stream can be pinned to the published known-answer
 * vectors of the algorithm (Salmon et al., SC'11). */
int njode_philox4x32_10(int32_t n, const uint32_t* ctr, const uint32_t* key, uint32_t* out,
                        njodeStream_t stream);

/* Euler-Maruyama paths of `sde` into paths_tm f64 [S+1][dim][N].
 * normals == NULL : standard normals from Philox4x32-10 keyed by `seed`, Box-Muller on 53-bit
 *                   uniforms; counter = (path, step, dim) for Heston (both normals of the
 *                   pair are used by one step), (path, ceil(step / 2), dim) otherwise (the
 *                   pair feeds steps 2m - 1 and 2m).
 * normals != NULL : device f64 array in the reference's draw order, [N][S][dim]
 *                   (Heston: [N][S][2][dim]); the recurrences then reproduce the reference's
 *                   float64 arithmetic operation by operation. */
int njode_generate_paths(const NjodeSde* sde, uint64_t seed, const double* normals,
                         double* paths_tm, njodeStream_t stream);

/* observed_tm[t][n] = (u < obs_perc), nb_obs[n] = number of observations at t >= 1.
 * uniforms == NULL: Philox draws; else device f64 [N][S+1] in the reference's order. */
int njode_sample_observations(int32_t n_paths, int32_t n_steps, double obs_perc, uint64_t seed,
                              const double* uniforms, uint8_t* observed_tm, int32_t* nb_obs,
                              njodeStream_t stream);

/* Collate, phase 1.  batch_idx: device int32 [B] dataset rows of the batch in batch order
 * (NULL = rows 0..B-1).  Outputs: count_per_time [S] = observations of the batch at grid
 * time t = 1..S; n_obs_ot [B] = nb_obs gathered.  The caller copies count_per_time to the
 * host: times = the grid times with a positive count, time_ptr = their running sum. */
int njode_collate_count(const uint8_t* observed_tm, const int32_t* nb_obs, int32_t n_paths,
                        int32_t n_steps, const int32_t* batch_idx, int32_t B,
                        int32_t* count_per_time, int32_t* n_obs_ot, njodeStream_t stream);

/* Collate, phase 2.  Rows sorted by time, then batch position (custom_collate_fn's order):
 * X [n_obs][dim * (1 + n_powers)] fp32, obs_idx [n_obs] int32, start_X [B][dim * (1 +
 * n_powers)].  powers: HOST array of the lifts appended by func_appl_X (k >= 1: 'power-k',
 * 0: 'exp'), n_powers <= 4.  count_per_time as written by phase 1 (device). */
int njode_collate_fill(const double* paths_tm, const uint8_t* observed_tm, int32_t n_paths,
                       int32_t dim, int32_t n_steps, const int32_t* batch_idx, int32_t B,
                       const int32_t* count_per_time, const int32_t* powers, int32_t n_powers,
                       float* start_X, float* X, int32_t* obs_idx, njodeStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NJODE_PRODUCER_H */
