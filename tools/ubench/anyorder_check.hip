// anyorder_check -- does a kernel launched with hipExtAnyOrderLaunch overlap the kernel queued in
// front of it on the SAME stream on gfx950, and what ordering is kept?  (hip_ext.h says the flag is
// "not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel and nothing for
// hipExtLaunchKernelGGL: measured here.)  Also: a software grid barrier among blocks launched beside a
// kernel that fills the chip.  Output: one JSON line per test.  (profiles/r05_anyorder_check.txt)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

// every block spins for `us` microseconds of the 100 MHz wall clock; then block 0 writes `val` to *out
__global__ void k_spin(int us, int* out, int val, const int* peek, int* peeked) {
  const unsigned long long t0 = wall_clock64();
  if (peek && threadIdx.x == 0 && blockIdx.x == 0) *peeked = __hip_atomic_load(peek, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (out && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(out, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// nb blocks, `rounds` software grid barriers (one counter per round), `us` of spinning between them
__global__ void k_grid_barrier(int us, int rounds, unsigned* ctr, int* ok) {
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(&ctr[r], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&ctr[r], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
        __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *ok = 1;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int* d;
  CK(hipMalloc(&d, 64 * sizeof(int)));
  unsigned* ctr;
  CK(hipMalloc(&ctr, 64 * sizeof(unsigned)));
  int h[64];
  auto run = [&](const char* name, int flags_b, int blocks_a, int blocks_b, int us_a, int us_b) -> int {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(d, 0, 64 * sizeof(int), st));
      CK(hipStreamSynchronize(st));
      const double t0 = now_us();
      hipLaunchKernelGGL(k_spin, dim3(blocks_a), dim3(256), 0, st, us_a, d + 0, 1, (const int*)nullptr, (int*)nullptr);
      // B peeks at A's flag when it STARTS: 0 = it started before A had finished
      hipExtLaunchKernelGGL(k_spin, dim3(blocks_b), dim3(256), 0, st, nullptr, nullptr, flags_b, us_b, d + 1, 1,
                            (const int*)(d + 0), d + 2);
      // C (ordinary launch) peeks at B's flag when it starts: must be 1
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 1, d + 3, 1, (const int*)(d + 1), d + 4);
      CK(hipStreamSynchronize(st));
      const double t1 = now_us();
      if (t1 - t0 < best) best = t1 - t0;
    }
    CK(hipMemcpy(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost));
    printf("{\"test\": \"%s\", \"flags_b\": %d, \"blocks_a\": %d, \"blocks_b\": %d, \"us_a\": %d, \"us_b\": %d, "
           "\"wall_us\": %.1f, \"b_saw_a_done_at_start\": %d, \"c_saw_b_done_at_start\": %d}\n",
           name, flags_b, blocks_a, blocks_b, us_a, us_b, best, h[2], h[4]);
    return 0;
  };
  if (run("serial", 0, 64, 64, 200, 200)) return 1;
  if (run("anyorder", hipExtAnyOrderLaunch, 64, 64, 200, 200)) return 1;
  if (run("anyorder_short_b", hipExtAnyOrderLaunch, 64, 1, 200, 50)) return 1;
  if (run("anyorder_full_chip_a", hipExtAnyOrderLaunch, 2048, 1, 200, 50)) return 1;   // A: 8 blocks per CU
  if (run("anyorder_long_b", hipExtAnyOrderLaunch, 64, 1, 50, 200)) return 1;
  // chain: A | P1 (any order) | D | P2 (any order, peeks P1's flag): P2 must see P1 done
  {
    CK(hipMemsetAsync(d, 0, 64 * sizeof(int), st));
    CK(hipStreamSynchronize(st));
    const double t0 = now_us();
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st, 100, d + 0, 1, (const int*)nullptr, (int*)nullptr);
    hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 150, d + 1, 1,
                          (const int*)(d + 0), d + 2);
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, st, 100, d + 3, 1, (const int*)(d + 1), d + 4);
    hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 50, d + 5, 1,
                          (const int*)(d + 1), d + 6);
    CK(hipStreamSynchronize(st));
    const double t1 = now_us();
    CK(hipMemcpy(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost));
    printf("{\"test\": \"chain\", \"wall_us\": %.1f, \"p1_saw_a_done\": %d, \"d_saw_p1_done\": %d, \"p2_saw_p1_done\": %d}\n",
           t1 - t0, h[2], h[4], h[6]);
  }
  // software grid barrier: 64 blocks, 7 barriers, beside a kernel that holds every CU (8 blocks per CU, 300 us)
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemsetAsync(ctr, 0, 64 * sizeof(unsigned), st));
    CK(hipMemsetAsync(d, 0, 64 * sizeof(int), st));
    CK(hipStreamSynchronize(st));
    const double t0 = now_us();
    if (mode == 1)
      hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, st, 300, d + 0, 1, (const int*)nullptr, (int*)nullptr);
    hipExtLaunchKernelGGL(k_grid_barrier, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 5, 7, ctr,
                          d + 1);
    CK(hipStreamSynchronize(st));
    const double t1 = now_us();
    CK(hipMemcpy(h, d, 64 * sizeof(int), hipMemcpyDeviceToHost));
    printf("{\"test\": \"grid_barrier_%s\", \"wall_us\": %.1f, \"ok\": %d}\n", mode ? "beside_full_chip" : "alone",
           t1 - t0, h[1]);
  }
  return 0;
}
