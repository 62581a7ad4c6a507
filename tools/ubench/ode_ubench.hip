// ode_ubench.hip -- standalone micro-benchmark of the segment plan's ODE kernels on a
// synthetic plan (n_tiles tiles of 16 segments, every segment L Euler steps).  Maintainer
// aid for the kernel work: it instantiates the SAME device code the library ships
// (njode_amd/csrc/*.h) for the demo shape and times it under controlled occupancy, so a loop
// body can be judged by cycles per tile-step per SIMD without the plan / tail effects of a
// real batch.  Not part of the product or of the tests.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ode_ubench.hip -o tools/ubench/ode_ubench
// run:   tools/ubench/ode_ubench            (prints one JSON line per case)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../njode_amd/csrc/njode_mfma_split.h"
#include "../../njode_amd/csrc/njode_ode2.h"
#include "njode_ode2_proto.h"
#include "njode_odex.h"

using namespace njode;
using C0 = Cfg<1, 10, 1, 2, 50, ACT_TANH, false, false, true, false>;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <class T> T* dalloc(size_t n) {
  T* p;
  CK(hipMalloc(&p, n * sizeof(T) + 256));
  CK(hipMemset(p, 0, n * sizeof(T) + 256));
  return p;
}
template <class T> void h2d(T* d, const std::vector<T>& h) {
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
}
static uint32_t rng_state = 12345u;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) * (1.0f / 16777216.0f)) * 2.0f - 1.0f;
}

// ---- wrappers around the shipped device functions with explicit occupancy ------------
template <class C, bool DROP, int WPS>
__global__ void __launch_bounds__(256, WPS) ub_fwd_single(KArgs a, int n_tiles) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode_fwd_single<C, DROP, false>(a, threadIdx.x & 63, wave, gridDim.x * 4, 0, n_tiles);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_bwd_single(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdSingleLds<C>::FLOATS];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode_bwd_single<C, DROP>(a, (lfp)lds_raw, wave, gridDim.x * 4, 0, n_tiles, blockIdx.x);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_bwd2_single(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdSingleLds<C>::FLOATS];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode2_bwd_single<C, DROP>(a, (lfp)lds_raw, wave, gridDim.x * 4, 0, n_tiles, blockIdx.x);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_fwd_split(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeFwdSplitLds<C>::FLOATS];
  ode_fwd_split<C, DROP, false, true>(a, (lfp)lds_raw, blockIdx.x, gridDim.x, 0, n_tiles);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_bwd_split(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdSplitLds<C>::FLOATS];
  ode_bwd_split<C, DROP>(a, (lfp)lds_raw, blockIdx.x, gridDim.x, 0, n_tiles, blockIdx.x);
}

template <class C, bool DROP, int WPS>
__global__ void __launch_bounds__(256, WPS) ub_fwd2_single(KArgs a, int n_tiles) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode2_fwd_single<C, DROP, false, true>(a, threadIdx.x & 63, wave, gridDim.x * 4, 0, n_tiles);
}

template <class C, bool DROP, int WPS>
__global__ void __launch_bounds__(256, WPS) ub_fwdx_single(KArgs a, int n_tiles) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  odex_fwd_single<C, DROP, false, XFwdFragsReg<C>>(a, (lfp)nullptr, threadIdx.x & 63, wave, gridDim.x * 4, 0, n_tiles);
}
// LDS-resident fragments: 512-thread blocks (8 waves share one 42 KB copy), WPS waves per SIMD
template <class C, bool DROP, int WPS>
__global__ void __launch_bounds__(512, WPS) ub_fwdx_lds(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[XFwdFragsLds<C>::LDS_BYTES / 4];
  const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  odex_fwd_single<C, DROP, false, XFwdFragsLds<C>>(a, (lfp)lds_raw, threadIdx.x & 63, wave, gridDim.x * 8, 0, n_tiles);
}

template <class C, bool DROP>
__global__ void __launch_bounds__(256, 1) ub_bwdx(KArgs a, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char ub_lds[];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  odex_bwd_single<C, DROP>(a, (char __attribute__((address_space(3)))*)ub_lds, wave, gridDim.x * 4, 0, n_tiles,
                           blockIdx.x);
}

template <class C, bool DROP>
__global__ void __launch_bounds__(512, 2) ub_bwdx8(KArgs a, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char ub_lds8[];
  const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);
  odex_bwd_wg<C, DROP, 8>(a, (char __attribute__((address_space(3)))*)ub_lds8, wave, gridDim.x * 8, 0, n_tiles,
                          blockIdx.x);
}

template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_bwd3_single(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdActLds<C>::FLOATS];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode3_bwd_single<C, DROP>(a, (lfp)lds_raw, wave, gridDim.x * 4, 0, n_tiles, blockIdx.x);
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) ub_bwd3_split(KArgs a, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdSplitLds<C>::FLOATS];
  ode3_bwd_split<C, DROP>(a, (lfp)lds_raw, blockIdx.x, gridDim.x, 0, n_tiles, blockIdx.x);
}

struct Problem {
  int n_tiles, L, n_items, B, K;
  KArgs a;
  std::vector<float> P;
  float *d_hend, *d_lamstart, *d_slab, *d_traj;
};

static Problem make_problem(int n_tiles, int L, bool drop) {
  using C = C0;
  Problem p;
  p.n_tiles = n_tiles;
  p.L = L;
  const int n = n_tiles * 16;
  p.n_items = n;
  p.B = n;   // one segment per path: prev = -1, start values as h0
  p.K = L;
  KArgs& a = p.a;
  memset(&a, 0, sizeof(a));
  p.P.resize(C::P);
  // xavier-like scale so activations stay in tanh's interesting range
  for (int i = 0; i < C::P; ++i) p.P[i] = 0.35f * frand();
  float* dP = dalloc<float>(C::P);
  h2d(dP, p.P);
  a.P = dP;
  a.frag = dalloc<float>(MF<C>::NALL * 64);
  a.B = p.B;
  a.n_obs = n;
  std::vector<float> sx(n), hx(n * C::H), X(n), lam(n * C::H);
  for (auto& v : sx) v = frand();
  for (auto& v : hx) v = frand();
  for (auto& v : X) v = frand();
  for (auto& v : lam) v = frand();
  float* d;
  d = dalloc<float>(n); h2d(d, sx); a.start_X = d;
  d = dalloc<float>(n); h2d(d, X); a.X = d;
  d = dalloc<float>(n * C::H); h2d(d, hx); a.h0start = d;
  a.h0row = dalloc<float>(n * C::H);
  d = dalloc<float>(n * C::H); h2d(d, lam); a.lam_end = d;
  std::vector<int> idx(n), len(n, L), zero(n, 0), neg(n, -1);
  for (int i = 0; i < n; ++i) idx[i] = i;
  int* di;
  di = dalloc<int>(n); h2d(di, idx); a.order = di;
  di = dalloc<int>(n); h2d(di, idx); a.obs_idx = di;
  di = dalloc<int>(n); h2d(di, len); a.item_len = di;
  di = dalloc<int>(n); h2d(di, zero); a.item_kbeg = di;
  di = dalloc<int>(n); h2d(di, neg); a.item_prev = di;
  di = dalloc<int>(n); h2d(di, zero); a.t_of_row = di;
  std::vector<float> dt(L, 0.01f), tt(L), tf(4, 0.0f);
  for (int s = 0; s < L; ++s) tt[s] = 0.01f * s;
  d = dalloc<float>(L); h2d(d, dt); a.step_dt = d;
  d = dalloc<float>(L); h2d(d, tt); a.step_t = d;
  d = dalloc<float>(4); h2d(d, tf); a.time_f32 = d;
  std::vector<long long> base(L + 4);
  for (int s = 0; s < L + 4; ++s) base[s] = (long long)s * n;
  long long* dl = dalloc<long long>(L + 4); h2d(dl, base); a.base_s = dl;
  {
    std::vector<long long> b16(L + 2);
    const long long n16 = (n + 15) / 16 * 16;
    for (int s = 0; s < L + 2; ++s) b16[s] = (long long)s * n16;
    long long* d16 = dalloc<long long>(L + 2); h2d(d16, b16); a.base16_s = d16;
    a.act = dalloc<float>((size_t)n16 * L * StepRec<C>::PER_CHAIN);   // (round 4: lane-major step records)
  }
  a.K = L;
  a.n_times = 1;
  p.d_traj = dalloc<float>((size_t)n * L * C::H);
  a.traj = p.d_traj;
  p.d_hend = dalloc<float>(n * C::H);
  a.h_end = p.d_hend;
  p.d_lamstart = dalloc<float>(n * C::H);
  a.lam_start = p.d_lamstart;
  a.g_h0 = dalloc<float>(n * C::H);
  p.d_slab = dalloc<float>((size_t)4096 * C::P);
  a.slab = p.d_slab;
  a.trash = dalloc<float>(1024 * 64);
  a.save_traj = 1;
  a.inv_batch = 1.0f / n;
  a.gid0 = 0;
  a.dc.seed_lo = 0x1234567u;
  a.dc.seed_hi = 0x89abcdeu;
  const float pdrop = drop ? 0.1f : 0.0f;
  a.dc.thr16 = (uint32_t)lrintf(pdrop * 65536.0f);
  a.dc.inv_keep = 1.0f / (1.0f - pdrop);
  a.keep = 1.0f - pdrop;
  k_pack_frags<C><<<(MF<C>::NALL * 64 + 255) / 256, 256>>>(a.P, a.frag);
  a.frag2 = dalloc<float>(MF<C>::NALL * 64);
  k_pack_frags2<C><<<(MF<C>::NALL * 64 + 255) / 256, 256>>>(a.P, a.frag2, a.dc.inv_keep);
  {
    uint16_t* fx = dalloc<uint16_t>((size_t)XF<C>::NALL * XNP * 64 * 8);
    k_pack_frags_x<C><<<(XF<C>::NALL * 64 * 8 + 255) / 256, 256>>>(a.P, fx, a.dc.inv_keep);
    a.fragx = fx;
  }
  CK(hipDeviceSynchronize());
  return p;
}

template <class F> static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

static std::vector<float> d2h(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
  return h;
}
static double rel_l2(const std::vector<float>& x, const std::vector<float>& y) {
  double num = 0, den = 0;
  for (size_t i = 0; i < x.size(); ++i) {
    num += (double)(x[i] - y[i]) * (x[i] - y[i]);
    den += (double)y[i] * y[i];
  }
  return std::sqrt(num / (den > 0 ? den : 1));
}

// streams = SIMDs that work concurrently (1 for a lone wave; 1024 for a full chip)
static void report(const char* name, const Problem& p, int blocks, int wps, float ms, int mfma_per_step,
                   const char* extra = "", int streams = 0) {
  const double tile_steps = (double)p.n_tiles * p.L;
  const double simds = streams > 0 ? streams : (blocks * 4 < 1024 ? blocks * 4 : 1024);
  const double us_per_step_simd = 1e3 * ms / (tile_steps / simds);
  const double cyc = us_per_step_simd * 2400.0;
  printf("{\"case\": \"%s\", \"n_tiles\": %d, \"L\": %d, \"blocks\": %d, \"wps\": %d, \"ms\": %.4f, "
         "\"us_per_tilestep_per_simd\": %.4f, \"cyc_at_2.4GHz\": %.0f, \"mfma_util_at_2.4GHz\": %.3f%s}\n",
         name, p.n_tiles, p.L, blocks, wps, ms, us_per_step_simd, cyc, mfma_per_step * 32.0 / cyc, extra);
  fflush(stdout);
}

int main(int argc, char** argv) {
  using C = C0;
  const int L = argc > 1 ? atoi(argv[1]) : 48;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.multiProcessorCount,
         prop.clockRate);
  const int CUS = prop.multiProcessorCount;

  for (int drop = 0; drop < 2; ++drop) {
    // ---- forward, one wave per tile: 1 wave alone, then k waves per SIMD on every CU
    {
      Problem p = make_problem(1, L, drop);
      float ms = drop ? time_ms([&] { ub_fwd_single<C, true, 1><<<1, 64>>>(p.a, 1); }, reps)
                      : time_ms([&] { ub_fwd_single<C, false, 1><<<1, 64>>>(p.a, 1); }, reps);
      report(drop ? "fwd1.lone.drop" : "fwd1.lone", p, 1, 0, ms, 81, "", 1);
      std::vector<float> ref = d2h(p.d_hend, (size_t)p.n_items * C::H);
      ms = drop ? time_ms([&] { ub_fwd2_single<C, true, 1><<<1, 64>>>(p.a, 1); }, reps)
                : time_ms([&] { ub_fwd2_single<C, false, 1><<<1, 64>>>(p.a, 1); }, reps);
      char ex[128];
      snprintf(ex, sizeof ex, ", \"rel_l2_vs_v1\": %.3e", rel_l2(d2h(p.d_hend, (size_t)p.n_items * C::H), ref));
      report(drop ? "fwd2.lone.drop" : "fwd2.lone", p, 1, 0, ms, 81, ex, 1);
    }
    for (int wps = 1; wps <= 4; ++wps) {
      const int blocks = CUS * wps;
      Problem p = make_problem(blocks * 4, L, drop);
      auto run = [&](auto kern) { return time_ms([&] { kern<<<blocks, 256>>>(p.a, p.n_tiles); }, reps); };
      float ms;
      if (drop) {
        ms = wps == 1 ? run(ub_fwd_single<C, true, 1>) : wps == 2 ? run(ub_fwd_single<C, true, 2>)
           : wps == 3 ? run(ub_fwd_single<C, true, 3>) : run(ub_fwd_single<C, true, 4>);
      } else {
        ms = wps == 1 ? run(ub_fwd_single<C, false, 1>) : wps == 2 ? run(ub_fwd_single<C, false, 2>)
           : wps == 3 ? run(ub_fwd_single<C, false, 3>) : run(ub_fwd_single<C, false, 4>);
      }
      report(drop ? "fwd1.drop" : "fwd1", p, blocks, wps, ms, 81);
      // reference result for the new kernels
      std::vector<float> ref = d2h(p.d_hend, (size_t)p.n_items * C::H);
      auto run2 = [&](auto kern) { return time_ms([&] { kern<<<blocks, 256>>>(p.a, p.n_tiles); }, reps); };
      if (drop) {
        ms = wps == 1 ? run2(ub_fwd2_single<C, true, 1>) : wps == 2 ? run2(ub_fwd2_single<C, true, 2>)
           : wps == 3 ? run2(ub_fwd2_single<C, true, 3>) : run2(ub_fwd2_single<C, true, 4>);
      } else {
        ms = wps == 1 ? run2(ub_fwd2_single<C, false, 1>) : wps == 2 ? run2(ub_fwd2_single<C, false, 2>)
           : wps == 3 ? run2(ub_fwd2_single<C, false, 3>) : run2(ub_fwd2_single<C, false, 4>);
      }
      char ex[128];
      snprintf(ex, sizeof ex, ", \"rel_l2_vs_v1\": %.3e", rel_l2(d2h(p.d_hend, (size_t)p.n_items * C::H), ref));
      report(drop ? "fwd2.drop" : "fwd2", p, blocks, wps, ms, 81, ex);
      if (wps <= 2) {
        CK(hipMemset(p.d_hend, 0, (size_t)p.n_items * C::H * sizeof(float)));
        if (drop) ms = wps == 1 ? run2(ub_fwdx_single<C, true, 1>) : run2(ub_fwdx_single<C, true, 2>);
        else ms = wps == 1 ? run2(ub_fwdx_single<C, false, 1>) : run2(ub_fwdx_single<C, false, 2>);
        snprintf(ex, sizeof ex, ", \"rel_l2_vs_v1\": %.3e", rel_l2(d2h(p.d_hend, (size_t)p.n_items * C::H), ref));
        report(drop ? "fwdx.drop" : "fwdx", p, blocks, wps, ms, 84, ex);
      }
      if (wps == 2 || wps == 4) {
        CK(hipMemset(p.d_hend, 0, (size_t)p.n_items * C::H * sizeof(float)));
        const int b8 = blocks / 2;
        auto run8 = [&](auto kern) { return time_ms([&] { kern<<<b8, 512>>>(p.a, p.n_tiles); }, reps); };
        if (drop) ms = wps == 2 ? run8(ub_fwdx_lds<C, true, 2>) : run8(ub_fwdx_lds<C, true, 4>);
        else ms = wps == 2 ? run8(ub_fwdx_lds<C, false, 2>) : run8(ub_fwdx_lds<C, false, 4>);
        snprintf(ex, sizeof ex, ", \"rel_l2_vs_v1\": %.3e", rel_l2(d2h(p.d_hend, (size_t)p.n_items * C::H), ref));
        report(drop ? "fwdx.lds.drop" : "fwdx.lds", p, blocks, wps, ms, 84, ex);
      }
    }
    // ---- backward, one wave per tile
    {
      Problem p = make_problem(1, L, drop);
      ub_fwd_single<C, false, 1><<<1, 64>>>(p.a, 1);
      float ms = drop ? time_ms([&] { ub_bwd_single<C, true><<<1, 256>>>(p.a, 1); }, reps)
                      : time_ms([&] { ub_bwd_single<C, false><<<1, 256>>>(p.a, 1); }, reps);
      report(drop ? "bwd1.lone.drop" : "bwd1.lone", p, 1, 0, ms, 241, "", 1);
    }
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = CUS * wps;
      Problem p = make_problem(blocks * 4, L, drop);
      if (drop) ub_fwd_single<C, true, 2><<<blocks, 256>>>(p.a, p.n_tiles);
      else ub_fwd_single<C, false, 2><<<blocks, 256>>>(p.a, p.n_tiles);
      float ms = drop ? time_ms([&] { ub_bwd_single<C, true><<<blocks, 256>>>(p.a, p.n_tiles); }, reps)
                      : time_ms([&] { ub_bwd_single<C, false><<<blocks, 256>>>(p.a, p.n_tiles); }, reps);
      report(drop ? "bwd1.drop" : "bwd1", p, blocks, wps, ms, 241);
      if (wps == 2) {
        std::vector<float> ref_lam = d2h(p.d_lamstart, (size_t)p.n_items * C::H);
        // gradient of the old kernel: sum of its slab rows
        auto slab_sum = [&](int rows) {
          std::vector<float> sl = d2h(p.d_slab, (size_t)rows * C::P), g(C::Ode::SIZE, 0.0f);
          for (int r = 0; r < rows; ++r)
            for (int i = 0; i < C::Ode::SIZE; ++i) g[i] += sl[(size_t)r * C::P + i];
          return g;
        };
        std::vector<float> ref_g = slab_sum(blocks);
        {
          CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
          CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
          float msb = drop ? time_ms([&] { ub_bwd2_single<C, true><<<blocks, 256>>>(p.a, p.n_tiles); }, reps)
                           : time_ms([&] { ub_bwd2_single<C, false><<<blocks, 256>>>(p.a, p.n_tiles); }, reps);
          char exb[192];
          snprintf(exb, sizeof exb, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e",
                   rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(blocks), ref_g));
          report(drop ? "bwd2.drop" : "bwd2", p, blocks, wps, msb, 241, exb);
        }
        {   // stored activations: forward (scaled one-wave role) stores, backward loads
          if (drop) ub_fwd2_single<C, true, 2><<<blocks, 256>>>(p.a, p.n_tiles);
          else ub_fwd2_single<C, false, 2><<<blocks, 256>>>(p.a, p.n_tiles);
          CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
          CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
          float msb = drop ? time_ms([&] { ub_bwd3_single<C, true><<<blocks, 256>>>(p.a, p.n_tiles); }, reps)
                           : time_ms([&] { ub_bwd3_single<C, false><<<blocks, 256>>>(p.a, p.n_tiles); }, reps);
          char exb[192];
          snprintf(exb, sizeof exb, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e",
                   rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(blocks), ref_g));
          report(drop ? "bwd3.drop" : "bwd3", p, blocks, wps, msb, 173, exb);
          // four-wave role: forward split stores, backward split loads (one tile per block)
          const int nbs = 512;
          if (drop) ub_fwd_split<C, true><<<nbs, 256>>>(p.a, p.n_tiles);
          else ub_fwd_split<C, false><<<nbs, 256>>>(p.a, p.n_tiles);
          CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
          CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
          msb = drop ? time_ms([&] { ub_bwd3_split<C, true><<<nbs, 256>>>(p.a, p.n_tiles); }, reps)
                     : time_ms([&] { ub_bwd3_split<C, false><<<nbs, 256>>>(p.a, p.n_tiles); }, reps);
          snprintf(exb, sizeof exb, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e",
                   rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(nbs), ref_g));
          report(drop ? "bwd3.split.drop" : "bwd3.split", p, nbs, 2, msb, 173, exb, 1024);
        }
        CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
        CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
        const int nb = CUS;
        const size_t lds_bytes = Ode2Img<C>::FLOATS * sizeof(float);
        if (drop) CK(hipFuncSetAttribute((const void*)k_ode2_bwd_pc<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        else CK(hipFuncSetAttribute((const void*)k_ode2_bwd_pc<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        float ms2 = drop ? time_ms([&] { k_ode2_bwd_pc<C, true><<<nb, 512, lds_bytes>>>(p.a); }, reps)
                         : time_ms([&] { k_ode2_bwd_pc<C, false><<<nb, 512, lds_bytes>>>(p.a); }, reps);
        char ex[192];
        snprintf(ex, sizeof ex, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e, \"lds_bytes\": %zu",
                 rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(nb), ref_g), lds_bytes);
        report(drop ? "bwd2pc.drop" : "bwd2pc", p, nb * 1, 2, ms2, 241, ex, 1024);
        // split-bf16 kernel, one wave per SIMD
        CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
        CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
        const size_t ldsx = XImg<C>::BYTES;
        if (drop) CK(hipFuncSetAttribute((const void*)ub_bwdx<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx));
        else CK(hipFuncSetAttribute((const void*)ub_bwdx<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx));
        float ms3 = drop ? time_ms([&] { ub_bwdx<C, true><<<nb, 256, ldsx>>>(p.a, p.n_tiles); }, reps)
                         : time_ms([&] { ub_bwdx<C, false><<<nb, 256, ldsx>>>(p.a, p.n_tiles); }, reps);
        snprintf(ex, sizeof ex, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e, \"lds_bytes\": %zu",
                 rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(nb), ref_g), ldsx);
        report(drop ? "bwdx.drop" : "bwdx", p, nb, 1, ms3, 300, ex, 1024);
        // ... eight waves per block, every fragment in LDS
        CK(hipMemset(p.d_slab, 0, (size_t)4096 * C::P * sizeof(float)));
        CK(hipMemset(p.d_lamstart, 0, (size_t)p.n_items * C::H * sizeof(float)));
        const size_t lds8 = XImg2<C, 8>::BYTES;
        if (drop) CK(hipFuncSetAttribute((const void*)ub_bwdx8<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
        else CK(hipFuncSetAttribute((const void*)ub_bwdx8<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
        float ms4 = drop ? time_ms([&] { ub_bwdx8<C, true><<<nb, 512, lds8>>>(p.a, p.n_tiles); }, reps)
                         : time_ms([&] { ub_bwdx8<C, false><<<nb, 512, lds8>>>(p.a, p.n_tiles); }, reps);
        snprintf(ex, sizeof ex, ", \"rel_l2_lam\": %.3e, \"rel_l2_grad\": %.3e, \"lds_bytes\": %zu",
                 rel_l2(d2h(p.d_lamstart, (size_t)p.n_items * C::H), ref_lam), rel_l2(slab_sum(nb), ref_g), lds8);
        report(drop ? "bwdx8.drop" : "bwdx8", p, nb, 2, ms4, 300, ex, 1024);
#ifdef NJ_XSTAMP
        {
          std::vector<unsigned long long> tsv(11);
          CK(hipMemcpy(tsv.data(), p.a.g_h0, 11 * 8, hipMemcpyDeviceToHost));
          printf("{\"case\": \"bwdx8.stamps%s\", \"steps\": %llu, \"cycles_per_step_by_phase\": [", drop ? ".drop" : "", tsv[10]);
          double tot = 0;
          for (int i = 0; i < 10; ++i) { printf("%s%.0f", i ? ", " : "", (double)tsv[i] / tsv[10]); tot += (double)tsv[i] / tsv[10]; }
          printf("], \"sum\": %.0f, \"phases\": \"L1 | act1+split | L2 | act2+split+img | dW3 | W3T+d2+split | dW2 | W2T+d1+split | dW1 | W1T+lam\"}\n", tot);
        }
#endif
      }
    }
    // ---- four waves per tile (latency form)
    {
      Problem p = make_problem(1, L, drop);
      float ms = drop ? time_ms([&] { ub_fwd_split<C, true><<<1, 256>>>(p.a, 1); }, reps)
                      : time_ms([&] { ub_fwd_split<C, false><<<1, 256>>>(p.a, 1); }, reps);
      report(drop ? "fwd4.lone.drop" : "fwd4.lone", p, 1, 0, ms, 81, "", 1);
      ms = drop ? time_ms([&] { ub_bwd_split<C, true><<<1, 256>>>(p.a, 1); }, reps)
                : time_ms([&] { ub_bwd_split<C, false><<<1, 256>>>(p.a, 1); }, reps);
      report(drop ? "bwd4.lone.drop" : "bwd4.lone", p, 1, 0, ms, 241, "", 1);
      ms = drop ? time_ms([&] { ub_bwd3_split<C, true><<<1, 256>>>(p.a, 1); }, reps)
                : time_ms([&] { ub_bwd3_split<C, false><<<1, 256>>>(p.a, 1); }, reps);
      report(drop ? "bwd3split.lone.drop" : "bwd3split.lone", p, 1, 0, ms, 173, "", 1);
    }
  }
  return 0;
}
