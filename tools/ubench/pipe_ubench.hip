// pipe_ubench.hip -- which instruction classes overlap with v_mfma_f32_16x16x4_f32 on one
// SIMD of gfx950?  (Maintainer aid; decides how the ODE kernels have to be budgeted:
// max(MFMA, VALU) or MFMA + VALU.)  Every block is 256 or 512 threads = 1 or 2 waves per SIMD.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 pipe_ubench.hip -o pipe_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

enum { W_NONE = 0, W_MFMA, W_FMA, W_EXP, W_MFMA_FMA, W_MFMA_EXP, W_BF16, W_BF16_FMA, W_LDS, W_MFMA_LDS,
       W_MFMA_FMA_SPARSE, W_CNDMASK, W_MFMA_XOR, W_MIX_F32FWD, W_MIX_F16FWD, W_MIX_BF16X6FWD, W_MIX_F32BWD, W_MIX_F16BWD, W_MIX_BF16X6BWD, W_BF16_K16, W_F16_K32, W_F16_K32_SUB, W_BF16_K32_SUB };

#define A_MFMA(ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define A_BF16(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(ACC) : "v"(ab))
#define A_BF16K16(ACC) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %1, %0" : "+v"(ACC) : "v"(ab4))
#define A_F16K32(ACC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(ACC) : "v"(ab))
#define A_FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b))
#define A_EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define A_SHLXOR(X) asm volatile("v_lshlrev_b32 %1, 13, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(X), "=&v"(tmp))
#define A_CMPSEL(X, Y) asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %3, %1, vcc" : "+v"(X) : "v"(Y), "v"(a), "v"(b) : "vcc")

template <int M, int NM, int NPL, int NT, bool F32>
__device__ __forceinline__ void mix_step(f32x4 (&acc)[4], float (&v)[16], float a, float b, bf16x8 ab) {
  if constexpr (M < NM) {
    if constexpr (F32) A_MFMA(acc[M & 3]); else A_BF16(acc[M & 3]);
    constexpr int p0 = M * NPL / NM, p1 = (M + 1) * NPL / NM, t0 = M * NT / NM, t1 = (M + 1) * NT / NM;
#pragma unroll
    for (int i = p0; i < p1; ++i) A_FMA(v[i & 15]);
#pragma unroll
    for (int i = t0; i < t1; ++i) A_EXP(v[(i + 7) & 15]);
    mix_step<M + 1, NM, NPL, NT, F32>(acc, v, a, b, ab);
  }
}

template <int WORK>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[16], float a, float b, int iters,
                                     float* lds) {
  bf16x8 ab = {0x3c00, 0x3c10, 0x3c20, 0x3c30, 0x3c40, 0x3c50, 0x3c60, 0x3c70};  // normal in f16 and bf16
  for (int it = 0; it < iters; ++it) {
    if constexpr (WORK == W_MFMA || WORK == W_MFMA_FMA || WORK == W_MFMA_EXP || WORK == W_MFMA_LDS ||
                  WORK == W_MFMA_FMA_SPARSE || WORK == W_MFMA_XOR) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          A_MFMA(acc[i]);
          if constexpr (WORK == W_MFMA_FMA) {
#pragma unroll
            for (int u = 0; u < 4; ++u) A_FMA(v[(4 * i + u) & 15]);
          }
          if constexpr (WORK == W_MFMA_FMA_SPARSE) {
            A_FMA(v[(4 * r + i) & 15]);
          }
          if constexpr (WORK == W_MFMA_XOR) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { float tmp; A_SHLXOR(v[(4 * i + u) & 15]); }
          }
          if constexpr (WORK == W_MFMA_EXP) A_EXP(v[(4 * r + i) & 15]);
          if constexpr (WORK == W_MFMA_LDS) v[(4 * r + i) & 15] += lds[(threadIdx.x + 64 * i + 256 * r) & 4095];
        }
    }
    if constexpr (WORK == W_BF16 || WORK == W_BF16_FMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          A_BF16(acc[i]);
          if constexpr (WORK == W_BF16_FMA) {
#pragma unroll
            for (int u = 0; u < 2; ++u) A_FMA(v[(2 * i + u) & 15]);
          }
        }
    }
    // projected instruction mixes of one Euler step of 16 chains:
    // NM matrix instructions, NP plain VALU, NT transcendental, evenly interleaved
    if constexpr (WORK >= W_MIX_F32FWD && WORK <= W_MIX_BF16X6BWD) {
      constexpr bool F32 = WORK == W_MIX_F32FWD || WORK == W_MIX_F32BWD;
      constexpr int NM = WORK == W_MIX_F32FWD ? 81 : WORK == W_MIX_F16FWD ? 42 : WORK == W_MIX_BF16X6FWD ? 84
                       : WORK == W_MIX_F32BWD ? 241 : WORK == W_MIX_F16BWD ? 150 : 300;
      constexpr int NPL = WORK == W_MIX_F32FWD ? 239 : WORK == W_MIX_F16FWD ? 302 : WORK == W_MIX_BF16X6FWD ? 407
                        : WORK == W_MIX_F32BWD ? 349 : WORK == W_MIX_F16BWD ? 500 : 690;
      constexpr int NT = 58;
      mix_step<0, NM, NPL, NT, F32>(acc, v, a, b, ab);
    }
    if constexpr (WORK == W_F16_K32_SUB || WORK == W_BF16_K32_SUB) {
      bf16x8 sub = {1, 2, 3, 4, 5, 6, 7, 8};   // subnormal bit patterns
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (WORK == W_F16_K32_SUB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ab), "v"(sub));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ab), "v"(sub));
        }
    }
    if constexpr (WORK == W_BF16_K16 || WORK == W_F16_K32) {
      typedef short bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 ab4 = {0x3c00, 0x3c10, 0x3c20, 0x3c30};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (WORK == W_BF16_K16) A_BF16K16(acc[i]); else A_F16K32(acc[i]);
        }
    }
    if constexpr (WORK == W_FMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 16; ++u) A_FMA(v[u]);
    }
    if constexpr (WORK == W_CNDMASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 16; ++u) A_CMPSEL(v[u], v[(u + 1) & 15]);
    }
    if constexpr (WORK == W_EXP) {
#pragma unroll
      for (int u = 0; u < 16; ++u) A_EXP(v[u]);
    }
    if constexpr (WORK == W_LDS) {
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] += lds[(threadIdx.x + 64 * u) & 4095];
    }
  }
}

// waves [0, 4) run WORK_A, waves [4, 8) (their SIMD partners in a 512-thread block) WORK_B
template <int WORK_A, int WORK_B>
__global__ void __launch_bounds__(512, 2) k_pipe(float* out, float a, float b, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a * i;
  __syncthreads();
  f32x4 acc[4];
  float v[16];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{a, b, a, b};
  for (int i = 0; i < 16; ++i) v[i] = a * (i + 1) + threadIdx.x * 1e-6f;
  const int wv = threadIdx.x >> 6;
  if (wv < 4) body<WORK_A>(acc, v, a, b, iters, lds);
  else body<WORK_B>(acc, v, a, b, iters, lds);
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int A, int B> static void run(const char* name, int threads, float* out, int iters, int blocks = 256) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_pipe<A, B><<<blocks, threads>>>(out, 0.5f, 0.25f, 100);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_pipe<A, B><<<blocks, threads>>>(out, 0.5f, 0.25f, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"case\": \"%s\", \"blocks\": %d, \"threads\": %d, \"ms\": %.4f, \"cycles_per_iter_at_2.4GHz\": %.1f}\n", name,
         blocks, threads, ms, ms * 1e-3 * 2.4e9 / iters);
  fflush(stdout);
}

int main() {
  float* out;
  CK(hipMalloc(&out, 1024 * 512 * sizeof(float)));
  const int N = 20000;
  // one wave per SIMD
  run<W_MFMA, W_NONE>("1w: 16 mfma_f32_16x16x4", 256, out, N);
  run<W_FMA, W_NONE>("1w: 64 v_fma", 256, out, N);
  run<W_CNDMASK, W_NONE>("1w: 64 cmp+cndmask", 256, out, N);
  run<W_EXP, W_NONE>("1w: 16 v_exp", 256, out, N);
  run<W_LDS, W_NONE>("1w: 16 ds_read_b32+add", 256, out, N);
  run<W_MFMA_FMA, W_NONE>("1w: 16 mfma + 64 v_fma interleaved", 256, out, N);
  run<W_MFMA_FMA_SPARSE, W_NONE>("1w: 16 mfma + 16 v_fma interleaved", 256, out, N);
  run<W_MFMA_XOR, W_NONE>("1w: 16 mfma + 64 (shl,xor) interleaved", 256, out, N);
  run<W_MFMA_EXP, W_NONE>("1w: 16 mfma + 16 v_exp interleaved", 256, out, N);
  run<W_MFMA_LDS, W_NONE>("1w: 16 mfma + 16 ds_read interleaved", 256, out, N);
  run<W_BF16, W_NONE>("1w: 16 mfma_bf16_16x16x32", 256, out, N);
  run<W_BF16_FMA, W_NONE>("1w: 16 mfma_bf16 + 32 v_fma interleaved", 256, out, N);
  // two waves per SIMD
  run<W_MFMA, W_MFMA>("2w: mfma | mfma", 512, out, N);
  run<W_FMA, W_FMA>("2w: 64 fma | 64 fma", 512, out, N);
  run<W_EXP, W_EXP>("2w: 16 exp | 16 exp", 512, out, N);
  run<W_MFMA, W_FMA>("2w: 16 mfma | 64 fma", 512, out, N);
  run<W_MFMA, W_EXP>("2w: 16 mfma | 16 exp", 512, out, N);
  run<W_MFMA, W_LDS>("2w: 16 mfma | 16 ds_read", 512, out, N);
  run<W_FMA, W_EXP>("2w: 64 fma | 16 exp", 512, out, N);
  run<W_BF16, W_FMA>("2w: 16 mfma_bf16 | 64 fma", 512, out, N);
  run<W_MFMA_FMA, W_MFMA_FMA>("2w: (16 mfma + 64 fma) x2", 512, out, N);
  run<W_BF16_K16, W_NONE>("1w: 16 mfma_bf16_16x16x16", 256, out, N);
  run<W_F16_K32, W_NONE>("1w: 16 mfma_f16_16x16x32", 256, out, N);
  run<W_BF16_K16, W_FMA>("2w: 16 mfma_bf16_16x16x16 | 64 fma", 512, out, N);
  run<W_F16_K32_SUB, W_NONE>("1w: 16 mfma_f16_16x16x32, B subnormal", 256, out, N);
  run<W_BF16_K32_SUB, W_NONE>("1w: 16 mfma_bf16_16x16x32, B subnormal", 256, out, N);
  // projected step mixes: 1, 2 and 4 waves per SIMD (2 blocks of 512 per CU = 4 waves / SIMD)
  const int M = 2000;
  run<W_MIX_F32FWD, W_NONE>("mix f32 fwd (81 mfma_f32, 239 valu, 58 trans) 1w", 256, out, M);
  run<W_MIX_F32FWD, W_MIX_F32FWD>("mix f32 fwd 2w", 512, out, M);
  run<W_MIX_F32FWD, W_MIX_F32FWD>("mix f32 fwd 4w", 512, out, M, 512);
  run<W_MIX_F16FWD, W_NONE>("mix f16x3 fwd (42 mfma_16x16x32, 302 valu, 58 trans) 1w", 256, out, M);
  run<W_MIX_F16FWD, W_MIX_F16FWD>("mix f16x3 fwd 2w", 512, out, M);
  run<W_MIX_F16FWD, W_MIX_F16FWD>("mix f16x3 fwd 4w", 512, out, M, 512);
  run<W_MIX_BF16X6FWD, W_MIX_BF16X6FWD>("mix bf16x6 fwd (84, 407, 58) 2w", 512, out, M);
  run<W_MIX_BF16X6FWD, W_MIX_BF16X6FWD>("mix bf16x6 fwd 4w", 512, out, M, 512);
  run<W_MIX_F32BWD, W_MIX_F32BWD>("mix f32 bwd (241 mfma_f32, 349 valu, 58 trans) 2w", 512, out, M);
  run<W_MIX_F16BWD, W_MIX_F16BWD>("mix f16x3 bwd (150, 500, 58) 2w", 512, out, M);
  run<W_MIX_F16BWD, W_MIX_F16BWD>("mix f16x3 bwd 4w", 512, out, M, 512);
  run<W_MIX_BF16X6BWD, W_MIX_BF16X6BWD>("mix bf16x6 bwd (300, 690, 58) 2w", 512, out, M);
  run<W_MIX_BF16X6BWD, W_MIX_BF16X6BWD>("mix bf16x6 bwd 4w", 512, out, M, 512);
  return 0;
}
