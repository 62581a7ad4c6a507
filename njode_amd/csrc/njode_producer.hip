// njode_producer.hip -- GPU-side batch producer (include/njode_producer.h): synthetic SDE
// datasets generated in HBM and the CSR-by-time collate of a batch of their paths.
//
// All of it is HBM-bound byte / f64 streaming work (no matrix cores): one thread per path
// walks the time grid writing time-major slices (unit stride across lanes), one workgroup
// per grid time ranks the batch's observations with a ballot scan.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>

#include "../../include/njode_producer.h"
#include "njode_error.h"

namespace {

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  njode::set_error_v(code, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(NJODE_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

// ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
// 1, 2, 3", SC'11) ---------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// 53-bit uniform in [0, 1) from two words (the construction numpy's random_double uses)
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// two independent standard normals (Box-Muller)
__device__ __forceinline__ void normal_pair(U4 r, double& z1, double& z2) {
  const double u1 = 1.0 - u53(r.x, r.y);   // (0, 1]
  const double u2 = u53(r.z, r.w);
  const double rad = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z1 = rad * c;
  z2 = rad * s;
}

enum : uint32_t { STREAM_PATHS = 0x70617468u, STREAM_OBS = 0x6f627376u };

__global__ void k_philox(int n, const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 r = philox4x32_10(U4{ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]},
                             key[2 * i], key[2 * i + 1]);
  out[4 * i] = r.x;
  out[4 * i + 1] = r.y;
  out[4 * i + 2] = r.z;
  out[4 * i + 3] = r.w;
}

// ---- path generation --------------------------------------------------------------------
// One thread per (dim j, path n); the recurrences spell out the reference's float64
// expression trees (this unit is compiled with -ffp-contract=off: no fused multiply-adds).
template <int MODEL>
__global__ void __launch_bounds__(256) k_generate(NjodeSde p, double dt, double sq,
                                                  double rho_c, uint32_t seed_lo, uint32_t seed_hi,
                                                  const double* __restrict__ normals,
                                                  double* __restrict__ paths) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long N = p.n_paths;
  if (tid >= N * p.dim) return;
  const int j = (int)(tid / N);
  const long long n = tid % N;
  const size_t slice = (size_t)p.dim * N;
  double* out = paths + (size_t)j * N + n;
  double s = p.S0, v = p.mean, z_next = 0.0;
  out[0] = s;
  for (int k = 1; k <= p.n_steps; ++k) {
    double z1, z2;
    if (normals) {
      if (MODEL == NJODE_SDE_HESTON) {
        const size_t base = (((size_t)n * p.n_steps + (k - 1)) * 2) * p.dim + j;
        z1 = normals[base];
        z2 = normals[base + p.dim];
      } else {
        z1 = normals[((size_t)n * p.n_steps + (k - 1)) * p.dim + j];
        z2 = 0.0;
      }
    } else if (MODEL == NJODE_SDE_HESTON) {
      const U4 r = philox4x32_10(U4{(uint32_t)n, (uint32_t)(n >> 32), (uint32_t)k, (uint32_t)j},
                                 seed_lo, seed_hi ^ STREAM_PATHS);
      normal_pair(r, z1, z2);
    } else {
      // one Philox call and one Box-Muller pair feed two consecutive steps (2m - 1, 2m)
      if (k & 1) {
        const U4 r = philox4x32_10(U4{(uint32_t)n, (uint32_t)(n >> 32), (uint32_t)((k + 1) >> 1),
                                      (uint32_t)j}, seed_lo, seed_hi ^ STREAM_PATHS);
        normal_pair(r, z1, z_next);
      } else {
        z1 = z_next;
      }
      z2 = 0.0;
    }
    const double tk = (double)(k - 1) * dt;
    const double pc = p.has_sine ? 1.0 + sin(p.sine_coeff * tk) : 1.0;
    const double dW = z1 * sq;
    if (MODEL == NJODE_SDE_BLACK_SCHOLES) {
      const double mu = p.drift * pc * s;       // stock_model.py:371
      const double sig = p.volatility * s;
      s = (s + mu * dt) + sig * dW;
    } else if (MODEL == NJODE_SDE_ORNSTEIN_UHLENBECK) {
      const double mu = -p.speed * pc * (s - p.mean);   // stock_model.py:413
      s = (s + mu * dt) + p.volatility * dW;
    } else {
      const double dZ = (p.correlation * z1 + rho_c * z2) * sq;   // stock_model.py:208-219
      const double vn = (v + (-p.speed * (v - p.mean)) * dt) + (p.volatility * sqrt(v)) * dZ;
      s = (s + (p.drift * pc * s) * dt) + (sqrt(vn) * s) * dW;
      v = vn;
    }
    out[(size_t)k * slice] = s;
  }
}

// ---- observation mask -------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sample_obs(int N, int S, double perc, uint32_t seed_lo,
                                                    uint32_t seed_hi,
                                                    const double* __restrict__ uniforms,
                                                    uint8_t* __restrict__ observed,
                                                    int* __restrict__ nb_obs) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int cnt = 0;
  for (int t = 0; t <= S; t += 2) {
    double u0, u1 = 1.0;
    if (uniforms) {
      u0 = uniforms[(size_t)n * (S + 1) + t];
      if (t + 1 <= S) u1 = uniforms[(size_t)n * (S + 1) + t + 1];
    } else {
      const U4 r = philox4x32_10(U4{(uint32_t)n, 0u, (uint32_t)(t >> 1), 0u}, seed_lo,
                                 seed_hi ^ STREAM_OBS);
      u0 = u53(r.x, r.y);
      u1 = u53(r.z, r.w);
    }
    const int o0 = u0 < perc;
    observed[(size_t)t * N + n] = (uint8_t)o0;
    cnt += t >= 1 ? o0 : 0;
    if (t + 1 <= S) {
      const int o1 = u1 < perc;
      observed[(size_t)(t + 1) * N + n] = (uint8_t)o1;
      cnt += o1;
    }
  }
  nb_obs[n] = cnt;
}

// ---- collate -----------------------------------------------------------------------------
constexpr int CB = 256;   // workgroup of the per-time kernels (4 waves)

__device__ __forceinline__ int block_sum(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// block t-1 counts the batch's observations at grid time t; the first blocks also gather nb_obs
__global__ void __launch_bounds__(CB) k_collate_count(const uint8_t* __restrict__ observed,
                                                      const int* __restrict__ nb_obs, int N,
                                                      const int* __restrict__ idx, int B,
                                                      int* __restrict__ count,
                                                      int* __restrict__ n_obs_ot) {
  __shared__ int sh[4];
  const int t = blockIdx.x + 1;
  const uint8_t* row = observed + (size_t)t * N;
  int c = 0;
  for (int b = threadIdx.x; b < B; b += CB) c += row[idx ? idx[b] : b];
  c = block_sum(c, sh);
  if (threadIdx.x == 0) count[blockIdx.x] = c;
  for (int b = blockIdx.x * CB + threadIdx.x; b < B; b += gridDim.x * CB)
    n_obs_ot[b] = nb_obs[idx ? idx[b] : b];
}

// func_appl_X lifts: exponent e >= 1 -> x^e by repeated multiplication, e == 0 -> exp(x)
__device__ __forceinline__ double ipow(double x, int e) {
  if (e == 0) return exp(x);
  double r = x;
  for (int i = 1; i < e; ++i) r *= x;
  return r;
}

struct Powers { int n; int e[4]; };

// block t-1 writes the rows of grid time t: rank within the time = number of earlier batch
// positions observed at t (ballot scan), base = observations at earlier times
__global__ void __launch_bounds__(CB) k_collate_fill(const double* __restrict__ paths,
                                                     const uint8_t* __restrict__ observed, int N,
                                                     int dim, const int* __restrict__ idx, int B,
                                                     const int* __restrict__ count, Powers pw,
                                                     float* __restrict__ X,
                                                     int* __restrict__ obs_idx) {
  __shared__ int sh[4];
  __shared__ int wave_cnt[4];
  const int t = blockIdx.x + 1;
  int before = 0;
  for (int i = threadIdx.x; i < blockIdx.x; i += CB) before += count[i];
  int base = block_sum(before, sh);
  if (count[blockIdx.x] == 0) return;   // block-uniform
  const uint8_t* row = observed + (size_t)t * N;
  const double* slice = paths + (size_t)t * dim * N;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int width = dim * (1 + pw.n);
  for (int b0 = 0; b0 < B; b0 += CB) {
    const int b = b0 + threadIdx.x;
    const int n = b < B ? (idx ? idx[b] : b) : 0;
    const bool on = b < B && row[n] != 0;
    const unsigned long long m = __ballot(on);
    __syncthreads();
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int r = base + __popcll(m & ((1ull << lane) - 1ull));
    for (int i = 0; i < w; ++i) r += wave_cnt[i];
    if (on) {
      obs_idx[r] = b;
      float* xr = X + (size_t)r * width;
      for (int j = 0; j < dim; ++j) {
        const double x = slice[(size_t)j * N + n];
        xr[j] = (float)x;
        for (int q = 0; q < pw.n; ++q) xr[(q + 1) * dim + j] = (float)ipow(x, pw.e[q]);
      }
    }
    base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
}

__global__ void __launch_bounds__(CB) k_collate_start(const double* __restrict__ paths, int N,
                                                      int dim, const int* __restrict__ idx, int B,
                                                      Powers pw, float* __restrict__ start_X) {
  const int b = blockIdx.x * CB + threadIdx.x;
  if (b >= B) return;
  const int n = idx ? idx[b] : b;
  const int width = dim * (1 + pw.n);
  for (int j = 0; j < dim; ++j) {
    const double x = paths[(size_t)j * N + n];
    start_X[(size_t)b * width + j] = (float)x;
    for (int q = 0; q < pw.n; ++q) start_X[(size_t)b * width + (q + 1) * dim + j] = (float)ipow(x, pw.e[q]);
  }
}

}  // namespace

extern "C" int njode_philox4x32_10(int32_t n, const uint32_t* ctr, const uint32_t* key,
                                   uint32_t* out, njodeStream_t stream) {
  if (n < 0 || (n > 0 && (!ctr || !key || !out))) return fail(NJODE_E_BADARG, "null argument");
  if (n == 0) return NJODE_OK;
  k_philox<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, ctr, key, out);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

extern "C" int njode_generate_paths(const NjodeSde* sde, uint64_t seed, const double* normals,
                                    double* paths_tm, njodeStream_t stream) {
  if (!sde || !paths_tm) return fail(NJODE_E_BADARG, "null argument");
  if (sde->n_paths <= 0 || sde->dim <= 0 || sde->n_steps <= 0)
    return fail(NJODE_E_BADARG, "n_paths, dim and n_steps must be positive");
  if (sde->model == NJODE_SDE_HESTON && !(sde->correlation >= -1.0 && sde->correlation <= 1.0))
    return fail(NJODE_E_BADARG, "correlation outside [-1, 1]");
  const double dt = sde->maturity / sde->n_steps;   // stock_model.py:358
  const double sq = __builtin_sqrt(dt);
  const double rho_c = __builtin_sqrt(1.0 - sde->correlation * sde->correlation);
  const int grid = cdiv((long long)sde->n_paths * sde->dim, 256);
  hipStream_t st = (hipStream_t)stream;
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  switch (sde->model) {
    case NJODE_SDE_BLACK_SCHOLES:
      k_generate<NJODE_SDE_BLACK_SCHOLES><<<grid, 256, 0, st>>>(*sde, dt, sq, rho_c, lo, hi, normals, paths_tm);
      break;
    case NJODE_SDE_ORNSTEIN_UHLENBECK:
      k_generate<NJODE_SDE_ORNSTEIN_UHLENBECK><<<grid, 256, 0, st>>>(*sde, dt, sq, rho_c, lo, hi, normals, paths_tm);
      break;
    case NJODE_SDE_HESTON:
      k_generate<NJODE_SDE_HESTON><<<grid, 256, 0, st>>>(*sde, dt, sq, rho_c, lo, hi, normals, paths_tm);
      break;
    default:
      return fail(NJODE_E_UNSUPPORTED, "unknown SDE model %d", sde->model);
  }
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

extern "C" int njode_sample_observations(int32_t n_paths, int32_t n_steps, double obs_perc,
                                         uint64_t seed, const double* uniforms,
                                         uint8_t* observed_tm, int32_t* nb_obs,
                                         njodeStream_t stream) {
  if (!observed_tm || !nb_obs) return fail(NJODE_E_BADARG, "null argument");
  if (n_paths <= 0 || n_steps <= 0) return fail(NJODE_E_BADARG, "n_paths and n_steps must be positive");
  k_sample_obs<<<cdiv(n_paths, 256), 256, 0, (hipStream_t)stream>>>(
      n_paths, n_steps, obs_perc, (uint32_t)seed, (uint32_t)(seed >> 32), uniforms, observed_tm,
      nb_obs);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

extern "C" int njode_collate_count(const uint8_t* observed_tm, const int32_t* nb_obs,
                                   int32_t n_paths, int32_t n_steps, const int32_t* batch_idx,
                                   int32_t B, int32_t* count_per_time, int32_t* n_obs_ot,
                                   njodeStream_t stream) {
  if (!observed_tm || !nb_obs || !count_per_time || !n_obs_ot)
    return fail(NJODE_E_BADARG, "null argument");
  if (n_paths <= 0 || n_steps <= 0 || B <= 0) return fail(NJODE_E_BADARG, "sizes must be positive");
  if (!batch_idx && B > n_paths) return fail(NJODE_E_BADARG, "batch larger than the dataset");
  k_collate_count<<<n_steps, CB, 0, (hipStream_t)stream>>>(observed_tm, nb_obs, n_paths, batch_idx,
                                                          B, count_per_time, n_obs_ot);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}

extern "C" int njode_collate_fill(const double* paths_tm, const uint8_t* observed_tm,
                                  int32_t n_paths, int32_t dim, int32_t n_steps,
                                  const int32_t* batch_idx, int32_t B,
                                  const int32_t* count_per_time, const int32_t* powers,
                                  int32_t n_powers, float* start_X, float* X, int32_t* obs_idx,
                                  njodeStream_t stream) {
  if (!paths_tm || !observed_tm || !count_per_time || !start_X)
    return fail(NJODE_E_BADARG, "null argument");
  if (n_paths <= 0 || dim <= 0 || n_steps <= 0 || B <= 0)
    return fail(NJODE_E_BADARG, "sizes must be positive");
  if (n_powers < 0 || n_powers > 4 || (n_powers > 0 && !powers))
    return fail(NJODE_E_BADARG, "n_powers must be in [0, 4]");
  Powers pw{n_powers, {1, 1, 1, 1}};
  for (int i = 0; i < n_powers; ++i) {
    if (powers[i] < 0) return fail(NJODE_E_BADARG, "powers must be >= 0");
    pw.e[i] = powers[i];
  }
  hipStream_t st = (hipStream_t)stream;
  k_collate_start<<<cdiv(B, CB), CB, 0, st>>>(paths_tm, n_paths, dim, batch_idx, B, pw, start_X);
  if (X && obs_idx)   // a batch without any observation has nothing to fill
    k_collate_fill<<<n_steps, CB, 0, st>>>(paths_tm, observed_tm, n_paths, dim, batch_idx, B,
                                           count_per_time, pw, X, obs_idx);
  HIP_TRY(hipGetLastError());
  return NJODE_OK;
}
