// mfma4x4_check.hip -- what the "edge rows" of njode_ode2.h rely on, checked on the device:
//   (1) operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1):
//       D[lane 4b + j][reg i] += A[lane 4b + i] * B[lane 4b + j];
//   (2) the all-reduce over the four lane groups of two registers with v_permlane16_swap +
//       v_permlane32_swap (edge_reduce of njode_mfma.h);
//   (3) issue cost of the 4x4x1 form against 16x16x4 (cycles per instruction, one wave).
// hipcc --offload-arch=gfx950 -O3 -o mfma4x4_check mfma4x4_check.hip && ./mfma4x4_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_layout(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[64 + l], B[64 + l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[r * 64 + l] = acc[r];
}

__device__ float edge_reduce(float p0, float p1) {
  u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
  const float t = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  u32x2 w = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(w[0]) + __uint_as_float(w[1]);
}
__global__ void k_reduce(const float* P, float* R) {
  const int l = threadIdx.x;
  R[l] = edge_reduce(P[l], P[64 + l]);
}

template <int KIND> __global__ void k_time(float* out, int n, long long* cyc) {
  const int l = threadIdx.x;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  float x = 1.0f + l * 1e-3f, y = 0.5f - l * 1e-3f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    } else if (KIND == 1) {
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    } else if (KIND == 2) {   // one dependent chain of 4x4x1
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
    } else {                  // a 16x16x4 chain with a dependent 4x4x1 chain beside it
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const f32x4 s = a0 + a1 + a2 + a3;
  out[l] = s[0] + s[1] + s[2] + s[3];
  if (l == 0) *cyc = t1 - t0;
}

int main() {
  float hA[128], hB[128], hD[256], hP[128], hR[64];
  srand(1);
  for (int i = 0; i < 128; ++i) { hA[i] = rand() / (float)RAND_MAX - 0.5f; hB[i] = rand() / (float)RAND_MAX - 0.5f; hP[i] = rand() / (float)RAND_MAX; }
  float *dA, *dB, *dD, *dP, *dR, *dO; long long* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMalloc(&dP, sizeof hP); hipMalloc(&dR, sizeof hR); hipMalloc(&dO, 256); hipMalloc(&dC, 8);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dP, hP, sizeof hP, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  k_reduce<<<1, 64>>>(dP, dR);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  hipMemcpy(hR, dR, sizeof hR, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int b = l / 4, j = l % 4;
      const float ref = hA[4 * b + r] * hB[4 * b + j] + hA[64 + 4 * b + r] * hB[64 + 4 * b + j];
      if (fabsf(ref - hD[r * 64 + l]) > 1e-6f) ++bad;
    }
  printf("layout D[lane 4b+j][reg i] = sum_k A[4b+i] B[4b+j]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  int bad2 = 0;
  for (int l = 0; l < 64; ++l) {
    const int c = l & 15, g = l >> 4;
    const int src = (g & 1) ? 64 : 0;     // even lane groups: register 0's total, odd: register 1's
    float ref = 0.f;
    for (int gg = 0; gg < 4; ++gg) ref += hP[src + 16 * gg + c];
    if (fabsf(ref - hR[l]) > 1e-5f) ++bad2;
  }
  printf("edge_reduce (lane groups 0/2: sum of p0, 1/3: sum of p1): %s (%d mismatches)\n", bad2 ? "WRONG" : "ok", bad2);
  const int n = 20000;
  const char* names[4] = {"4 x 16x16x4 independent", "4 x 4x4x1 independent", "4 x 4x4x1 dependent chain",
                          "2 x 16x16x4 + 2 x 4x4x1 (dependent) interleaved"};
  for (int kind = 0; kind < 4; ++kind) {
    long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) k_time<0><<<1, 64>>>(dO, n, dC);
      if (kind == 1) k_time<1><<<1, 64>>>(dO, n, dC);
      if (kind == 2) k_time<2><<<1, 64>>>(dO, n, dC);
      if (kind == 3) k_time<3><<<1, 64>>>(dO, n, dC);
      hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    }
    printf("%-52s %.1f memtime ticks per loop body (x clock ratio = cycles)\n", names[kind], (double)c / n);
  }
  return (bad || bad2) ? 1 : 0;
}
