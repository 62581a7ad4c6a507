// njode_device.h -- device-side building blocks of the gfx950 NJ-ODE kernels: activations,
// dropout streams, parameter layout, and the one-chain-per-lane VALU forms of the networks.
//
// The DEFAULT kernels of both plans run on the f32 matrix cores (njode_mfma*.h); what is
// below the helpers in this file is the VALU form they replaced, still used for GRU models,
// for network shapes outside the matrix-core kernels' range and for NJODE_ODE=valu A/B runs.
// Execution model of that VALU form (CDNA4, wave = 64 lanes):
//   * one independent chain (a path, or a (path, inter-observation segment)
//     work item) per lane; its hidden state and the activations of the layer
//     being evaluated live in that lane's VGPRs;
//   * MLP weights are wave-uniform: they are fetched through the scalar data
//     path (s_load_dwordxN from the constant address space into SGPRs) and fed
//     to v_fmac_f32 as the scalar operand, so the vector ALU issues only FMAs
//     and the LDS stays free;
//   * weight gradients are a reduction over chains (lanes): each wave stages
//     (delta, activation) rows in LDS, switches to an 8x8 lane grid that owns
//     register tiles of dW, and accumulates the outer products from LDS.
// (BASELINE.json's north_star asked for exactly this form; DESIGN.md section 4a has the
// measurements that made the matrix cores the default.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace njode {

typedef const float __attribute__((address_space(4))) * cfp;  // scalar-load ptr
typedef float __attribute__((address_space(3))) * lfp;        // LDS ptr
typedef const float __attribute__((address_space(3))) * lcp;  // read-only LDS ptr (weights)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 __attribute__((address_space(3))) * lf4p;           // LDS ptr, 16 B

#define NJ_DEV __device__ __forceinline__

constexpr int ACT_TANH = 0;
constexpr int ACT_RELU = 1;

// Re-materialise a uniform pointer so the compiler cannot hoist the (loop
// invariant) weight loads out of a time loop and spill thousands of SGPRs.
NJ_DEV cfp launder(cfp p) {
  const unsigned long long v = (unsigned long long)p;
  // readfirstlane: the pointer is uniform by construction, but after loops with
  // lane-dependent bodies the compiler may hold it in a VGPR ("+s" then fails)
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return (cfp)(((unsigned long long)hi << 32) | lo);
}
NJ_DEV cfp as_cfp(const float* p) { return (cfp)(unsigned long long)p; }
// same for a (32-bit) LDS address: keeps loop-invariant ds_reads of the weights from
// being hoisted out of the time loop into hundreds of VGPRs
NJ_DEV lcp launder(lcp p) {
  unsigned v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p);
  asm volatile("" : "+s"(v));
  return (lcp)(unsigned long long)v;
}

// Pin values where they are computed.  Without it the compiler sinks a network's
// FMAs into a later (divergent) block that holds their only use, while the scalar
// weight loads stay behind: thousands of SGPRs live across the branch -> spills.
template <int N> NJ_DEV void pin(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// ---- activations -------------------------------------------------------------
// tanh(x) = 1 - 2 / (2^(2 log2(e) x) + 1): v_exp_f32 + v_rcp_f32; saturates cleanly
// (exp -> inf gives 1, exp -> 0 gives -1).  Its ABSOLUTE error is ~1e-7 everywhere
// (cancellation against 1 for small |x|), fine for the unmasked path.  Masked models
// feed their own predictions back as inputs (self-imputation, models.py:465-467),
// which amplifies rounding differences step after step, so those translation units are
// built with NJ_ACC_TANH=1 (tanh_accurate below).
#ifndef NJ_ACC_TANH
#define NJ_ACC_TANH 0
#endif
// Round 5: the float64 truth test (tests/test_hip_f64_truth.py) put a number on it -- with the
// round-1 form (odd Taylor polynomial below |x| = 0.3, the exp form above: up to 3 ulp around
// 0.3 - 0.8 even with an exact exp2, where 1 - 2 r cancels) the masked prediction path ended
// 3.2x as far from the float64 result as the reference's own fp32 run; restating just this tanh
// in the CPU oracle reproduces the factor, a k-ordered summation does not.  Now, for |x| < 1, the
// [7/6] Pade approximant of Lambert's continued fraction written as a CORRECTION of x,
//   tanh x = x - x q,   q = x^2 (45045 + 2772 x^2 + 27 x^4) / (135135 + 62370 x^2 + 3150 x^4 + 28 x^6)
// (truncation < 1e-9 there; q < 0.33, so the error of the division enters scaled down), and the
// exp form above 1, where 2 r < 0.24: <= 1.3 ulp overall with exact exp2 / rcp (the round-1 form:
// 2.8), at the same instruction count.
NJ_DEV float tanh_accurate(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float x2 = x * x;
  const float m = fmaf(fmaf(x2, 27.0f, 2772.0f), x2, 45045.0f);
  const float d = fmaf(fmaf(fmaf(x2, 28.0f, 3150.0f), x2, 62370.0f), x2, 135135.0f);
  const float q = (x2 * m) * __builtin_amdgcn_rcpf(d);
  const float s = fmaf(-x, q, x);
  return fabsf(x) < 1.0f ? s : t;
}
NJ_DEV float tanh_f(float x) {
#if NJ_ACC_TANH
  return tanh_accurate(x);
#else
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
#endif
}
template <int ACT> NJ_DEV float act_f(float z) {
  if constexpr (ACT == ACT_TANH) return tanh_f(z);
  else return fmaxf(z, 0.0f);
}
// derivative expressed through the activation's output
template <int ACT> NJ_DEV float dact_f(float a) {
  if constexpr (ACT == ACT_TANH) return 1.0f - a * a;
  else return a > 0.0f ? 1.0f : 0.0f;
}

// ---- dropout -------------------------------------------------------------------
// Counter-based: the stream of one network evaluation is keyed by
// (seed, global path id, time key, network id) so forward and backward
// regenerate identical masks and results do not depend on the sharding.
struct DropCtx {
  uint32_t seed_lo, seed_hi;
  uint32_t thr16;   // drop if 16 random bits < thr16  (p = thr16 / 65536)
  float inv_keep;   // 1 / (1 - p)
};
NJ_DEV uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
NJ_DEV uint32_t drop_state(const DropCtx& dc, uint32_t gid_lo, uint32_t gid_hi,
                           uint32_t tkey, uint32_t net) {
  uint32_t h = fmix32(dc.seed_lo ^ (gid_lo * 0x9e3779b9u));
  h = fmix32(h ^ dc.seed_hi ^ (gid_hi * 0x7f4a7c15u) ^ (tkey * 0x85ebca6bu));
  h = fmix32(h ^ (net * 0xc2b2ae35u) ^ 0x27d4eb2fu);
  return h ? h : 0x9e3779b9u;
}
// W (<= 64) keep-bits from an xorshift32 stream, two 16-bit draws per word
template <int W> NJ_DEV uint64_t keep_mask(uint32_t& s, uint32_t thr16) {
  static_assert(W <= 64, "dropout masks are held in one 64-bit register pair");
  uint64_t m = 0;
#pragma unroll
  for (int u = 0; u < W; u += 2) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    m |= (uint64_t)((s & 0xffffu) >= thr16) << u;
    if (u + 1 < W) m |= (uint64_t)((s >> 16) >= thr16) << (u + 1);
  }
  return m;
}

// ---- network layout (get_ffnn, reference models.py:140-166) ----------------------
// NH hidden layers of width W; parameters packed [w0, b0, w1, b1, ...] with
// nn.Linear weights [out][in] row-major.
template <int IN_, int OUT_, int NH_, int W_> struct NetL {
  static constexpr int IN = IN_, OUT = OUT_, NH = NH_, W = (NH_ > 0 ? W_ : 1);
  static constexpr int lin(int l) { return l == 0 ? IN : W; }
  static constexpr int lout(int l) { return l == NH ? OUT : W; }
  static constexpr int woff(int l) {
    int o = 0;
    for (int q = 0; q < l; ++q) o += lin(q) * lout(q) + lout(q);
    return o;
  }
  static constexpr int boff(int l) { return woff(l) + lin(l) * lout(l); }
  static constexpr int SIZE = woff(NH + 1);
};

// out[j] = b[j] + sum_i W[j][i] in[i]
template <int K, int N, class WP>
NJ_DEV void dense(WP Wp, WP bp, const float (&in)[K], float (&out)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float acc = bp[j];
#pragma unroll
    for (int i = 0; i < K; ++i) acc = fmaf(Wp[j * K + i], in[i], acc);
    out[j] = acc;
  }
}

// Transposed product from the transposed copy WT[K][N]:
//   g_i = sum_j WT[i][j] dout[j];   io[i] = f(i, g_i, io[i])   (in place)
template <int K, int N, class WP, class F>
NJ_DEV void dense_T_inplace(WP WTp, const float (&dout)[N], float (&io)[K], F f) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) acc = fmaf(WTp[i * N + j], dout[j], acc);
    io[i] = f(i, acc, io[i]);
  }
}
// same, for the index range [LO, HI) of the input only, into out[HI-LO]
template <int K, int N, int LO, int HI, class WP>
NJ_DEV void dense_T_range(WP WTp, const float (&dout)[N], float (&out)[HI - LO]) {
#pragma unroll
  for (int i = LO; i < HI; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) acc = fmaf(WTp[i * N + j], dout[j], acc);
    out[i - LO] = acc;
  }
}

// Forward of one network; a1/a2 receive the (dropout-scaled) hidden activations.
template <class NL, int ACT, bool DROP, class WP>
NJ_DEV void net_fwd(WP P0, const float (&in)[NL::IN], float (&out)[NL::OUT],
                    float (&a1)[NL::W], float (&a2)[NL::W], uint64_t m1,
                    uint64_t m2, float inv_keep) {
  const WP P = launder(P0);  // fresh weight loads per evaluation (no cross-call hoisting)
  if constexpr (NL::NH == 0) {
    dense<NL::IN, NL::OUT>(P + NL::woff(0), P + NL::boff(0), in, out);
  } else {
    dense<NL::IN, NL::W>(P + NL::woff(0), P + NL::boff(0), in, a1);
#pragma unroll
    for (int j = 0; j < NL::W; ++j) {
      float a = act_f<ACT>(a1[j]);
      if constexpr (DROP) a = ((m1 >> j) & 1) ? a * inv_keep : 0.0f;
      a1[j] = a;
    }
    if constexpr (NL::NH == 1) {
      dense<NL::W, NL::OUT>(P + NL::woff(1), P + NL::boff(1), a1, out);
    } else {
      dense<NL::W, NL::W>(P + NL::woff(1), P + NL::boff(1), a1, a2);
#pragma unroll
      for (int j = 0; j < NL::W; ++j) {
        float a = act_f<ACT>(a2[j]);
        if constexpr (DROP) a = ((m2 >> j) & 1) ? a * inv_keep : 0.0f;
        a2[j] = a;
      }
      dense<NL::W, NL::OUT>(P + NL::woff(2), P + NL::boff(2), a2, out);
    }
  }
  pin(out);
}

// ---- wave-level LDS ordering ---------------------------------------------------
// One wave per workgroup: LDS instructions of a wave execute in issue order, so
// only the compiler has to be kept from reordering across the role switch.
NJ_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- dW tiles: outer products summed over the wave's chains -----------------------
// dW[N][K] (+ bias column K) is owned by an 8x8 lane grid: lane (tj, ti) holds
// the TR x TC register tile of rows tj*TR.. and columns ti*TC...  Chains stage
// their delta[N] and act[K] (+1.0) as rows of two LDS images whose tiles are
// padded to 16 B so every lane reads its operands with ds_read_b128; the row
// stride is an odd number of 16-B slots, so the ds_write_b128 of consecutive
// lanes (rows) hit distinct banks.
// CH = chains staged per phase (64 / CH phases per wave): 32 by default, 16 where the
// weights share the LDS with the staging rows.

template <int N_, int K_, int CH_ = 32> struct Tile {
  static constexpr int N = N_, K = K_, CH = CH_;
  static constexpr int TR = (N + 7) / 8, TC = (K + 1 + 7) / 8;
  static constexpr int TRP = (TR + 3) & ~3, TCP = (TC + 3) & ~3;
  static constexpr int SD = 8 * TRP + 4, SA = 8 * TCP + 4;  // row strides, floats
  static constexpr int NACC = TR * TC;
  static constexpr int LDS_FLOATS = CH * (SD + SA);

  // value at padded position `pos` of a delta row / activation row
  template <int POS> static NJ_DEV float dval(const float (&delta)[N]) {
    constexpr int t = POS / TRP, u = POS % TRP, j = t * TR + u;
    if constexpr (u < TR && j < N) return delta[j];
    else return 0.0f;
  }
  template <int POS> static NJ_DEV float aval(const float (&act)[K]) {
    constexpr int t = POS / TCP, v = POS % TCP, i = t * TC + v;
    if constexpr (v < TC && i < K) return act[i];
    else if constexpr (v < TC && i == K) return 1.0f;
    else return 0.0f;
  }
  template <int Q> static NJ_DEV void put_d(lfp row, const float (&delta)[N]) {
    if constexpr (Q < 2 * TRP) {
      constexpr int u0 = (4 * Q) % TRP;
      if constexpr (u0 < TR) {  // skip slots that are padding only
        f4 v = {dval<4 * Q>(delta), dval<4 * Q + 1>(delta), dval<4 * Q + 2>(delta),
                dval<4 * Q + 3>(delta)};
        *(lf4p)(row + 4 * Q) = v;
      }
      put_d<Q + 1>(row, delta);
    }
  }
  template <int Q> static NJ_DEV void put_a(lfp row, const float (&act)[K]) {
    if constexpr (Q < 2 * TCP) {
      constexpr int v0 = (4 * Q) % TCP;
      if constexpr (v0 < TC) {
        f4 v = {aval<4 * Q>(act), aval<4 * Q + 1>(act), aval<4 * Q + 2>(act),
                aval<4 * Q + 3>(act)};
        *(lf4p)(row + 4 * Q) = v;
      }
      put_a<Q + 1>(row, act);
    }
  }

  // acc += sum over the CH staged rows of delta (x) [act, 1]
  static NJ_DEV void accumulate(lfp lds, float (&acc)[NACC], int lane) {
    const int tj = lane >> 3, ti = lane & 7;
    lfp dp = lds + tj * TRP;
    lfp ap = lds + CH * SD + ti * TCP;
#pragma unroll 2
    for (int c = 0; c < CH; ++c) {
      float dv[TRP], av[TCP];
#pragma unroll
      for (int q = 0; q < TRP / 4; ++q) {
        f4 t = *(lf4p)(dp + c * SD + 4 * q);
        dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int q = 0; q < TCP / 4; ++q) {
        f4 t = *(lf4p)(ap + c * SA + 4 * q);
        av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int u = 0; u < TR; ++u)
#pragma unroll
        for (int v = 0; v < TC; ++v)
          acc[u * TC + v] = fmaf(dv[u], av[v], acc[u * TC + v]);
    }
  }

  // Whole-wave update: every lane contributes its chain's (delta, act); lanes
  // whose chain is inactive must pass delta == 0.  Must be called in
  // wave-uniform control flow.
  static NJ_DEV void update(lfp lds, float (&acc)[NACC], const float (&delta)[N],
                            const float (&act)[K], int lane) {
#pragma unroll
    for (int ph = 0; ph < 64 / CH; ++ph) {
      if ((lane / CH) == ph) {
        const int row = lane % CH;
        put_d<0>(lds + row * SD, delta);
        put_a<0>(lds + CH * SD + row * SA, act);
      }
      wave_lds_sync();
      accumulate(lds, acc, lane);
      wave_lds_sync();
    }
  }

  // Store this wave's partial dW into its slab (same layout as the parameters).
  static NJ_DEV void flush(const float (&acc)[NACC], float* w, float* b, int lane) {
    const int tj = lane >> 3, ti = lane & 7;
#pragma unroll
    for (int u = 0; u < TR; ++u)
#pragma unroll
      for (int v = 0; v < TC; ++v) {
        const int j = tj * TR + u, i = ti * TC + v;
        if (j < N) {
          if (i < K) w[j * K + i] = acc[u * TC + v];
          else if (i == K) b[j] = acc[u * TC + v];
        }
      }
  }
};

// Per-network gradient accumulators (register tiles for each layer)
template <class NL, int CH_ = 32> struct NetAcc {
  using T0 = Tile<NL::lout(0), NL::lin(0), CH_>;
  using T1 = Tile<NL::lout(NL::NH >= 1 ? 1 : 0), NL::lin(NL::NH >= 1 ? 1 : 0), CH_>;
  using T2 = Tile<NL::lout(NL::NH >= 2 ? 2 : 0), NL::lin(NL::NH >= 2 ? 2 : 0), CH_>;
  float a0[T0::NACC];
  float a1[NL::NH >= 1 ? T1::NACC : 1];
  float a2[NL::NH >= 2 ? T2::NACC : 1];
  static constexpr int LDS_FLOATS =
      T0::LDS_FLOATS > T1::LDS_FLOATS
          ? (T0::LDS_FLOATS > T2::LDS_FLOATS ? T0::LDS_FLOATS : T2::LDS_FLOATS)
          : (T1::LDS_FLOATS > T2::LDS_FLOATS ? T1::LDS_FLOATS : T2::LDS_FLOATS);
  NJ_DEV void zero() {
#pragma unroll
    for (int q = 0; q < T0::NACC; ++q) a0[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < (NL::NH >= 1 ? T1::NACC : 1); ++q) a1[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < (NL::NH >= 2 ? T2::NACC : 1); ++q) a2[q] = 0.0f;
  }
  NJ_DEV void flush(float* slab, int lane) const {
    T0::flush(a0, slab + NL::woff(0), slab + NL::boff(0), lane);
    if constexpr (NL::NH >= 1) T1::flush(a1, slab + NL::woff(1), slab + NL::boff(1), lane);
    if constexpr (NL::NH >= 2) T2::flush(a2, slab + NL::woff(2), slab + NL::boff(2), lane);
  }
};

// Backward of one network evaluation (whole wave, uniform control flow).
//   dout  : gradient w.r.t. the network output (zero for inactive lanes)
//   a1,a2 : hidden activations saved by net_fwd (dropout-scaled); clobbered
//   din   : gradient w.r.t. inputs [DLO, DHI) (only if DHI > DLO)
// PT = transposed copy of the parameters (same offsets, weights stored [in][out]).
template <class NL, int ACT, bool DROP, int DLO, int DHI, class WP, int CHN>
NJ_DEV void net_bwd(WP PT0, lfp lds, NetAcc<NL, CHN>& g, const float (&in)[NL::IN],
                    const float (&dout)[NL::OUT], float (&a1)[NL::W],
                    float (&a2)[NL::W], uint64_t m1, uint64_t m2, float inv_keep,
                    float keep, float (&din)[(DHI > DLO) ? (DHI - DLO) : 1], int lane) {
  using A = NetAcc<NL, CHN>;
  const WP PT = launder(PT0);
  if constexpr (NL::NH == 0) {
    A::T0::update(lds, g.a0, dout, in, lane);
    if constexpr (DHI > DLO)
      dense_T_range<NL::IN, NL::OUT, DLO, DHI>(PT + NL::woff(0), dout, din);
  } else {
    auto back = [&](uint64_t m) {
      return [=](int i, float gsum, float a) -> float {
        // a = act(z) * (kept ? inv_keep : 0); d/dz = gsum * scale * act'(act(z))
        if constexpr (DROP) {
          return ((m >> i) & 1) ? gsum * inv_keep * dact_f<ACT>(a * keep) : 0.0f;
        } else {
          return gsum * dact_f<ACT>(a);
        }
      };
    };
    if constexpr (NL::NH == 2) {
      A::T2::update(lds, g.a2, dout, a2, lane);
      dense_T_inplace<NL::W, NL::OUT>(PT + NL::woff(2), dout, a2, back(m2));  // a2 <- delta2
      A::T1::update(lds, g.a1, a2, a1, lane);
      dense_T_inplace<NL::W, NL::W>(PT + NL::woff(1), a2, a1, back(m1));      // a1 <- delta1
    } else {
      A::T1::update(lds, g.a1, dout, a1, lane);
      dense_T_inplace<NL::W, NL::OUT>(PT + NL::woff(1), dout, a1, back(m1));  // a1 <- delta1
    }
    A::T0::update(lds, g.a0, a1, in, lane);
    if constexpr (DHI > DLO)
      dense_T_range<NL::IN, NL::W, DLO, DHI>(PT + NL::woff(0), a1, din);
  }
  if constexpr (DHI > DLO) pin(din);
}

// Backward of one network evaluation w.r.t. its inputs only (no weight gradients, no
// LDS, no cross-lane traffic: may run under a divergent branch).  a1 / a2 as saved by
// net_fwd; clobbered.  din receives d/d in[DLO .. DHI).
template <class NL, int ACT, bool DROP, int DLO, int DHI, class WP>
NJ_DEV void net_bwd_inputs(WP PT0, const float (&dout)[NL::OUT], float (&a1)[NL::W],
                           float (&a2)[NL::W], uint64_t m1, uint64_t m2, float inv_keep,
                           float keep, float (&din)[DHI - DLO]) {
  const WP PT = launder(PT0);
  if constexpr (NL::NH == 0) {
    dense_T_range<NL::IN, NL::OUT, DLO, DHI>(PT + NL::woff(0), dout, din);
  } else {
    auto back = [&](uint64_t m) {
      return [=](int i, float gsum, float a) -> float {
        if constexpr (DROP) {
          return ((m >> i) & 1) ? gsum * inv_keep * dact_f<ACT>(a * keep) : 0.0f;
        } else {
          return gsum * dact_f<ACT>(a);
        }
      };
    };
    if constexpr (NL::NH == 2) {
      dense_T_inplace<NL::W, NL::OUT>(PT + NL::woff(2), dout, a2, back(m2));  // a2 <- delta2
      dense_T_inplace<NL::W, NL::W>(PT + NL::woff(1), a2, a1, back(m1));      // a1 <- delta1
    } else {
      dense_T_inplace<NL::W, NL::OUT>(PT + NL::woff(1), dout, a1, back(m1));
    }
    dense_T_range<NL::IN, NL::W, DLO, DHI>(PT + NL::woff(0), a1, din);
  }
  pin(din);
}

}  // namespace njode
