// chain_ubench.hip -- what ONE wave can do per Euler step when a lane is a UNIT of the layer
// (weights of the unit's row in that lane's registers, the layer input broadcast from LDS):
// the building block of the wave-per-path kernels (njode_chain.h).  Prints cycles per "step"
// (three dependent layers K1 -> 50 -> 50 -> OUT with tanh) for several broadcast forms.
//   v0  LDS broadcast (ds_read_b128 of a wave-private vector), one fma chain per layer
//   v1  the same, two accumulators (even / odd inputs)
//   v2  v_pk_fma_f32 on (even, odd) input pairs
//   v3  v_readlane broadcast (no LDS)
//   v4  DPP row_newbcast on block-replicated registers (lane 16 g + c = unit 4 c + g; no LDS)
//   v5  LDS broadcast, every read of a layer issued before its fma chain
// Usage: chain_ubench [blocks] [waves_per_block]
#include <hip/hip_runtime.h>
#include "../../njode_amd/csrc/njode_dpp.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float __attribute__((address_space(3))) * lfp;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f4 __attribute__((address_space(3))) * lf4p;

__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <int K, int V> __device__ __forceinline__ float layer(const float (&w)[K + 1], lfp xin, float own, int lane) {
  // own: this lane's value of the input vector (lane < K); result: the unit's pre-activation
  if constexpr (V == 4) {
    float R[4];
    njode::dpp_replicate(own, R);
    float wk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wk[k] = w[k];
    return njode::dpp_dot<K>(w[K], R, wk);
  } else if constexpr (V == 3) {
    float acc = w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc = fmaf(w[k], __builtin_amdgcn_readlane(own, k), acc);
    return acc;
  } else {
    xin[lane] = own;
    constexpr int KQ = (K + 3) / 4;
    f4 x[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) x[q] = *(lf4p)(xin + 4 * q);
    if constexpr (V == 5) {
      __builtin_amdgcn_sched_barrier(0);
      float acc = w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(w[k], x[k / 4][k % 4], acc);
      return acc;
    } else if constexpr (V == 0) {
      float acc = w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(w[k], x[k / 4][k % 4], acc);
      return acc;
    } else if constexpr (V == 1) {
      float a0 = w[K], a1 = 0.0f;
#pragma unroll
      for (int k = 0; k < K; k += 2) {
        a0 = fmaf(w[k], x[k / 4][k % 4], a0);
        if (k + 1 < K) a1 = fmaf(w[k + 1], x[(k + 1) / 4][(k + 1) % 4], a1);
      }
      return a0 + a1;
    } else {
      f2 acc = {w[K], 0.0f};
#pragma unroll
      for (int k = 0; k + 1 < K; k += 2) {
        f2 ww = {w[k], w[k + 1]};
        f2 xx = {x[k / 4][k % 4], x[(k + 1) / 4][(k + 1) % 4]};
        acc = __builtin_elementwise_fma(ww, xx, acc);
      }
      float r = acc.x + acc.y;
      if (K & 1) r = fmaf(w[K - 1], x[(K - 1) / 4][(K - 1) % 4], r);
      return r;
    }
  }
}

template <int K1, int W, int OUT, int V>
__global__ void __launch_bounds__(512) k_chain(const float* __restrict__ P, float* __restrict__ out,
                                               unsigned long long* __restrict__ cyc, int n_steps) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 3 * 128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  lfp X0 = (lfp)lds + wv * 384, X1 = X0 + 128, X2 = X1 + 128;
  float w1[K1 + 1], w2[W + 1], w3[W + 1];
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
#pragma unroll
  for (int k = 0; k <= K1; ++k) w1[k] = P[(k * 64 + lane) % 4096] * 0.1f;
#pragma unroll
  for (int k = 0; k <= W; ++k) w2[k] = P[(k * 64 + lane + 1000) % 4096] * 0.1f;
#pragma unroll
  for (int k = 0; k <= W; ++k) w3[k] = P[(k * 64 + lane + 2000) % 4096] * 0.1f;
  for (int i = lane; i < 384; i += 64) X0[i] = 0.0f;
  float h = P[lane] * 0.5f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < n_steps; ++s) {
    const float th = tanh_fast(h);
    float z = layer<K1, V>(w1, X0, th, lane);
    float a = tanh_fast(z);
    z = layer<W, V>(w2, X1, a, lane);
    a = tanh_fast(z);
    z = layer<W, V>(w3, X2, a, lane);
    h = lane < OUT ? fmaf(0.01f, z, h) : 0.0f;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[wave * 64 + lane] = h;
  if (lane == 0) cyc[wave] = t1 - t0;
}

template <int K1, int W, int OUT, int V> void run(const char* name, int blocks, int wpb, const float* dP, float* dout,
                                                  unsigned long long* dcyc, int n_steps) {
  hipLaunchKernelGGL((k_chain<K1, W, OUT, V>), dim3(blocks), dim3(64 * wpb), 0, 0, dP, dout, dcyc, 16);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_chain<K1, W, OUT, V>), dim3(blocks), dim3(64 * wpb), 0, 0, dP, dout, dcyc, n_steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(blocks * wpb);
  hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0, sum = 0;
  for (auto v : c) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
  printf("{\"variant\": \"%s\", \"K1\": %d, \"W\": %d, \"OUT\": %d, \"blocks\": %d, \"waves_per_block\": %d, "
         "\"cycles_per_step_mean\": %.1f, \"cycles_per_step_max\": %.1f, \"us_per_step_wall\": %.4f}\n",
         name, K1, W, OUT, blocks, wpb, sum / c.size() / n_steps, mx / n_steps, ms * 1000.0 / n_steps);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

// y = W x (K = 50 inputs, 50 outputs) through dpp_dot against the host: checks the unit <-> lane map,
// the replication and the k order (bitwise: the same two fma chains in k order on both sides)
template <int K> __global__ void k_dpp_check(const float* __restrict__ Wm, const float* __restrict__ x, float* __restrict__ y) {
  const int lane = threadIdx.x, u = njode::dpp_unit(lane);
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = u < K ? Wm[u * K + k] : 0.0f;
  float R[4];
  njode::dpp_replicate(u < K ? x[u] : 0.0f, R);
  const float acc = njode::dpp_dot<K>(0.25f, R, w);
  if (u < K) y[u] = acc;
}
template <int K> static int dpp_check() {
  std::vector<float> W(K * K), x(K), y(K), ref(K);
  for (int i = 0; i < K * K; ++i) W[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
  for (int i = 0; i < K; ++i) x[i] = (float)((i * 40503u + 7) % 1999) / 999.0f - 1.0f;
  for (int j = 0; j < K; ++j) {   // dpp_dot's order: blocks of 16 units alternate between two accumulators
    float acc[2] = {0.25f, 0.0f};
    const int full = K / 4;   // full quads
    int cur = 0;
    for (int q0 = 0; q0 < full;) {
      const int n = full - q0 >= 4 ? 4 : full - q0;
      for (int k = 4 * q0; k < 4 * (q0 + n); ++k) acc[cur] = fmaf(x[k], W[j * K + k], acc[cur]);
      q0 += n;
      if (n == 4) cur ^= 1;
    }
    for (int k = 4 * full; k < K; ++k) acc[cur] = fmaf(x[k], W[j * K + k], acc[cur]);
    ref[j] = K > 16 ? acc[0] + acc[1] : acc[0];
  }
  float *dW, *dx, *dy;
  hipMalloc(&dW, K * K * 4); hipMalloc(&dx, K * 4); hipMalloc(&dy, K * 4);
  hipMemcpy(dW, W.data(), K * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), K * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_dpp_check<K>), dim3(1), dim3(64), 0, 0, dW, dx, dy);
  hipMemcpy(y.data(), dy, K * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int j = 0; j < K; ++j) bad += y[j] != ref[j];
  printf("{\"check\": \"dpp_dot\", \"K\": %d, \"mismatches\": %d, \"y0\": %.9g, \"ref0\": %.9g}\n", K, bad, y[0], ref[0]);
  return bad;
}

int main(int argc, char** argv) {
  if (dpp_check<50>() + dpp_check<41>() + dpp_check<42>() + dpp_check<11>() + dpp_check<64>() != 0) return 1;
  const int blocks = argc > 1 ? atoi(argv[1]) : 1, wpb = argc > 2 ? atoi(argv[2]) : 1;
  const int n_steps = 3000;
  float *dP, *dout;
  unsigned long long* dcyc;
  std::vector<float> hP(4096);
  for (int i = 0; i < 4096; ++i) hP[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
  hipMalloc(&dP, 4096 * 4);
  hipMalloc(&dout, (size_t)blocks * wpb * 64 * 4);
  hipMalloc(&dcyc, (size_t)blocks * wpb * 8);
  hipMemcpy(dP, hP.data(), 4096 * 4, hipMemcpyHostToDevice);
  // config 5 with the x part of layer 1 hoisted (41 h inputs + tdiff) and without
  run<42, 50, 41, 0>("lds_1acc", blocks, wpb, dP, dout, dcyc, n_steps);
  run<42, 50, 41, 1>("lds_2acc", blocks, wpb, dP, dout, dcyc, n_steps);
  run<42, 50, 41, 2>("lds_pkfma", blocks, wpb, dP, dout, dcyc, n_steps);
  run<42, 50, 41, 3>("readlane", blocks, wpb, dP, dout, dcyc, n_steps);
  run<42, 50, 41, 4>("dpp_newbcast", blocks, wpb, dP, dout, dcyc, n_steps);
  run<42, 50, 41, 5>("lds_reads_first", blocks, wpb, dP, dout, dcyc, n_steps);
  run<85, 50, 41, 0>("lds_1acc", blocks, wpb, dP, dout, dcyc, n_steps);
  run<85, 50, 41, 2>("lds_pkfma", blocks, wpb, dP, dout, dcyc, n_steps);
  // demo shape (x, tau hoisted: 10 h inputs + tdiff)
  run<11, 50, 10, 0>("lds_1acc", blocks, wpb, dP, dout, dcyc, n_steps);
  run<11, 50, 10, 2>("lds_pkfma", blocks, wpb, dP, dout, dcyc, n_steps);
  run<11, 50, 10, 3>("readlane", blocks, wpb, dP, dout, dcyc, n_steps);
  run<11, 50, 10, 4>("dpp_newbcast", blocks, wpb, dP, dout, dcyc, n_steps);
  run<11, 50, 10, 5>("lds_reads_first", blocks, wpb, dP, dout, dcyc, n_steps);
  return 0;
}
