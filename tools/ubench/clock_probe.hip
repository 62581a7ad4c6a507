// clock_probe.hip -- shader clock as a function of how many CUs are busy (maintainer aid).
// Each block spins on dependent f32 MFMAs for `iters` rounds and reports s_memtime (shader
// clock ticks) against s_memrealtime (100 MHz constant clock): a latency-bound kernel that
// occupies 4 CUs (config 5: B = 50 = 4 tiles) may not run at the clock a full chip reaches.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_probe(unsigned long long* out, float a, int iters) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (acc[0] == 12345.f) out[0] = 0;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 2 * 4096 * sizeof(unsigned long long));
  for (int blocks : {1, 4, 16, 64, 256, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 20000;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      k_probe<<<blocks, 256>>>(d, 1.0f, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2 * blocks);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      const double ticks = (double)h[0], real = (double)h[1];
      // 16 dependent MFMAs per round
      printf("{\"blocks\": %d, \"rep\": %d, \"kernel_ms\": %.3f, \"memtime_ticks\": %.0f, \"realtime_ticks_100MHz\": %.0f, "
             "\"memtime_MHz\": %.1f, \"cycles_per_dependent_mfma_by_event\": %.2f}\n",
             blocks, rep, ms, ticks, real, ticks / (real / 100.0), 0.0);
      printf("{\"blocks\": %d, \"ns_per_dependent_mfma\": %.3f}\n", blocks, ms * 1e6 / (16.0 * iters));
    }
  }
  return 0;
}
