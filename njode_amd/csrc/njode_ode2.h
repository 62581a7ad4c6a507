// njode_ode2.h -- second generation of the segment plan's ODE kernels on the f32 matrix
// cores (same D-layout and fragment tables as njode_mfma.h).  What changed, and why
// (profiles/r02_ubench_*.jsonl):
//
//  * Scaled fragments (k_pack_frags2).  The tanh layers' weights carry the factor
//    2 log2(e), so an activation is exp2 -> add -> rcp -> fma (no multiply), and the
//    layers that consume dropout outputs carry 1 / (1 - p), so applying a mask is a select,
//    not a select and a multiply -- forward AND backward (the inverted-dropout factor
//    commutes with every product it meets; dW tiles are rescaled once, when flushed).
//  * Forward: output tiles are produced in pairs (two accumulators alternate, so
//    back-to-back MFMAs never depend on each other) and the activation of a finished pair
//    runs in the shadow of the next pair's MFMAs; the H-wide output product uses two
//    accumulators as well.
//  * Backward on the forward's STORED hidden activations (ode3_bwd_single below).
//    (Measured and rejected prototypes -- a recomputing one-wave backward on the scaled
//    fragments, a producer / consumer split of the sweep over SIMD partners -- live in
//    tools/ubench/njode_ode2_proto.h, which only tools/ubench/ode_ubench.hip compiles.)
#pragma once
#include "njode_mfma.h"
#include "njode_mfma_rows.h"

namespace njode {

constexpr float TANH_PRESCALE = 2.8853900817779268f;  // 2 log2(e)

template <class C> struct Ode2Scale {
  static constexpr float S = C::ACT == ACT_TANH ? TANH_PRESCALE : 1.0f;
};

// A-fragments of the six products, with the scale factors folded in:
//   F1 = S [W1 | b1]      F2 = S [ik W2 | b2]     F3 = [ik W3 | b3]
//   B3 = ik W3^T          B2 = ik W2^T            B1 = W1^T (h rows)
// S = 2 log2(e) for tanh networks (1 for relu), ik = 1 / (1 - p) (1 without dropout).
template <class C> NJ_DEV float ode_frag_value(const float* __restrict__ P, int idx, float S, float ik) {
  using M = MF<C>;
  const int f = idx >> 6, l = idx & 63, g = l >> 4, c = l & 15;
  const float* Po = P + C::OFF_ODE;
  using NL = typename C::Ode;
  const float *W1 = Po + NL::woff(0), *b1 = Po + NL::boff(0), *W2 = Po + NL::woff(1),
              *b2 = Po + NL::boff(1), *W3 = Po + NL::woff(2), *b3 = Po + NL::boff(2);
  float v = 0.0f;
  if (f < M::F2) {
    const int mt = (f - M::F1) / M::Q0, q = (f - M::F1) % M::Q0;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::W) v = S * (ui < M::IN0 ? W1[uo * M::IN0 + M::col0(ui)] : (ui == M::IN0 ? b1[uo] : 0.0f));
  } else if (f < M::F3) {
    const int mt = (f - M::F2) / M::Q1, q = (f - M::F2) % M::Q1;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::W) v = ui < M::W ? S * ik * W2[uo * M::W + ui] : (ui == M::W ? S * b2[uo] : 0.0f);
  } else if (f < M::NFWD) {
    const int mt = (f - M::F3) / M::Q1, q = (f - M::F3) % M::Q1;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::H) v = ui < M::W ? ik * W3[uo * M::W + ui] : (ui == M::W ? b3[uo] : 0.0f);
  } else if (f < M::B2) {
    const int mt = (f - M::B3) / M::QH, q = (f - M::B3) % M::QH;
    const int ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < M::W && uo < M::H) v = ik * W3[uo * M::W + ui];
  } else if (f < M::B1) {
    const int mt = (f - M::B2) / M::QW, q = (f - M::B2) % M::QW;
    const int ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < M::W && uo < M::W) v = ik * W2[uo * M::W + ui];
  } else {
    const int mt = (f - M::B1) / M::QW, q = (f - M::B1) % M::QW;
    const int u = row_unit(mt, c), uo = 4 * q + g;
    if (u < M::IN0 && uo < M::W) v = W1[uo * M::IN0 + M::col0(u)];
  }
  return v;
}
template <class C>
__global__ void k_pack_frags2(const float* __restrict__ P, float* __restrict__ frag, float ik) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= MF<C>::NALL * 64) return;
  frag[idx] = ode_frag_value<C>(P, idx, Ode2Scale<C>::S, ik);
}
// both tables of the ODE network in one launch: the plain one (njode_mfma.h: four-wave role,
// lockstep kernels) and the scaled one
template <class C>
__global__ void k_pack_frags12(const float* __restrict__ P, float* __restrict__ frag,
                               float* __restrict__ frag2, float ik) {
  constexpr int N = MF<C>::NALL * 64;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * N) return;
  if (idx < N) frag[idx] = ode_frag_value<C>(P, idx, 1.0f, 1.0f);
  else frag2[idx - N] = ode_frag_value<C>(P, idx - N, Ode2Scale<C>::S, ik);
}

// activation of a pre-scaled pre-activation: tanh(z) from z' = 2 log2(e) z
template <int ACT> NJ_DEV float act2_f(float zs) {
  if constexpr (ACT == ACT_TANH) {
    const float e = __builtin_amdgcn_exp2f(zs);
    return fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
  } else {
    return fmaxf(zs, 0.0f);
  }
}


// in0 with the lane's CONSTANT entries (bias unit 1, padding 0) in a register computed once per
// kernel: left as literals in the per-step select chain, `g == 1 ? 1.0f : 0.0f` comes back from
// the compiler as two nested exec-mask branches in the middle of the Euler-step loop, which
// splits the loop body into three blocks and makes every s_waitcnt at their joins conservative
template <class C, int U> constexpr bool in0_var() {
  return U < C::H + C::D + 2 + (C::CURT ? 1 : 0);
}
template <class C> struct In0Const {
  float v[MF<C>::Q0];
  NJ_DEV void init(int g) {
    const float z[C::D] = {};
#pragma unroll
    for (int q = 0; q < MF<C>::Q0; ++q) v[q] = 0.0f;
    init_q<0>(g, z);
  }
  template <int Q> NJ_DEV void init_q(int g, const float (&z)[C::D]) {
    if constexpr (Q < MF<C>::Q0) {
      const float e0 = in0_var<C, 4 * Q + 0>() ? 0.0f : in0_unit<C, 4 * Q + 0>(0.0f, z, 0.0f, 0.0f);
      const float e1 = in0_var<C, 4 * Q + 1>() ? 0.0f : in0_unit<C, 4 * Q + 1>(0.0f, z, 0.0f, 0.0f);
      const float e2 = in0_var<C, 4 * Q + 2>() ? 0.0f : in0_unit<C, 4 * Q + 2>(0.0f, z, 0.0f, 0.0f);
      const float e3 = in0_var<C, 4 * Q + 3>() ? 0.0f : in0_unit<C, 4 * Q + 3>(0.0f, z, 0.0f, 0.0f);
      float x = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
      asm volatile("" : "+v"(x));
      v[Q] = x;
      init_q<Q + 1>(g, z);
    }
  }
};
template <class C, int Q>
NJ_DEV void in0_fill_c(float (&b0)[MF<C>::Q0], const float (&h)[MF<C>::QH], const float (&tx)[C::D],
                       float tau, float tdiff, int g, const In0Const<C>& K) {
  if constexpr (Q < MF<C>::Q0) {
    float th = 0.0f;
    if constexpr (4 * Q < C::H) th = tanh_f(h[Q]);
    float x = K.v[Q];
    if constexpr (in0_var<C, 4 * Q + 3>()) x = g == 3 ? in0_unit<C, 4 * Q + 3>(th, tx, tau, tdiff) : x;
    if constexpr (in0_var<C, 4 * Q + 2>()) x = g == 2 ? in0_unit<C, 4 * Q + 2>(th, tx, tau, tdiff) : x;
    if constexpr (in0_var<C, 4 * Q + 1>()) x = g == 1 ? in0_unit<C, 4 * Q + 1>(th, tx, tau, tdiff) : x;
    if constexpr (in0_var<C, 4 * Q + 0>()) x = g == 0 ? in0_unit<C, 4 * Q + 0>(th, tx, tau, tdiff) : x;
    b0[Q] = x;
    in0_fill_c<C, Q + 1>(b0, h, tx, tau, tdiff, g, K);
  }
}

// ---- stored step records ----------------------------------------------------------------
// While the f32 pipe (matrix + vector: one pipe on gfx950) bounds the ODE kernels, HBM idles:
// the forward stores, for every Euler step of a tile of 16 chains, the state before the step and
// the two hidden activation vectors (WITHOUT the inverted-dropout factor; a dropped unit as
// -0.0f), and the backward reads them instead of recomputing 68 of its 241 MFMAs, 58
// transcendentals and the dropout stream.
//
// Layout (round 4: LANE-MAJOR quads).  The record of tile t at Euler step s starts at chain
// record base16_s[s] + 16 t (KArgs) and holds, per lane, the 2 Q1 + QH registers
//   group A = a2[0 .. Q1), h[0 .. QH)     group B = a1[0 .. Q1)
// in the order the backward needs them.  Four consecutive registers of a lane are one 16-byte
// quad and the 64 lanes' quads are contiguous, so a wave moves a quad with ONE
// global_load/store_dwordx4 (1 KB, fully coalesced); what does not fill a quad is stored as
// single dwords [register][lane].  Per layer the first NF = Q1 / 4 quads are whole (in the
// four-wave role they are exactly the registers wave 0 .. NF-1 own); group A's tail list
// (a2's L1 = Q1 % 4 leftover registers, then h) follows as quads + singles; group B ends with
// a1's L1 leftovers as singles.  W = 50, H = 10: A = 3 + 1 quads, B = 3 quads + 1 dword --
// 8 memory instructions per lane and step where the register-major layout of round 2/3 took
// 26 dwords + 3 dwords of a separate 40-byte checkpoint record (whose address hung on a
// per-step load of its own).  Size: (2 Q1 + QH) 4 floats per chain record (464 B).
template <class C> struct StepRec {
  using M = MF<C>;
  static constexpr int Q1 = M::Q1, QH = M::QH;
  static constexpr int NF = Q1 / 4, L1 = Q1 % 4;
  static constexpr int NT = L1 + QH, NTQ = NT / 4, NTS = NT % 4;   // group A's tail list
  static constexpr int TAILQ = NF * 256;                            // float offsets in the block
  static constexpr int TAILS = (NF + NTQ) * 256;
  static constexpr int GA = TAILS + NTS * 64;
  static constexpr int B1S = GA + NF * 256;
  static constexpr int FLOATS = B1S + L1 * 64;
  static constexpr int PER_CHAIN = FLOATS / 16;
  static constexpr int WT = NF < 3 ? NF : 3;   // four-wave role: the wave that stores the tail list
  static_assert(FLOATS == (2 * Q1 + QH) * 64, "step record: every register exactly once");
};
template <class C> NJ_DEV float* rec_block(const float* act, long long b16, int tile) {
  return (float*)act + (size_t)(b16 + 16 * (long long)tile) * StepRec<C>::PER_CHAIN;
}
NJ_DEV void quad_store(float* p, float x, float y, float z, float w) {
  const f32x4 v = {x, y, z, w};
  *(f32x4*)p = v;
}
// tail list of group A from a2's leftovers and h (compile-time index)
template <class C, int I>
NJ_DEV float rec_tail_get(const float (&a2)[MF<C>::Q1], const float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
  if constexpr (I < R::L1) return a2[4 * R::NF + I];
  else if constexpr (I < R::NT) return h[I - R::L1];
  else return 0.0f;
}
template <class C, int I>
NJ_DEV void rec_tail_put(float v, float (&a2)[MF<C>::Q1], float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
  if constexpr (I < R::L1) a2[4 * R::NF + I] = v;
  else if constexpr (I < R::NT) h[I - R::L1] = v;
}
template <class C, int K = 0>
NJ_DEV void rec_tail_store(float* blk, int lane, const float (&a2)[MF<C>::Q1], const float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
  if constexpr (K < R::NTQ) {
    quad_store(blk + R::TAILQ + K * 256 + lane * 4, rec_tail_get<C, 4 * K>(a2, h),
               rec_tail_get<C, 4 * K + 1>(a2, h), rec_tail_get<C, 4 * K + 2>(a2, h),
               rec_tail_get<C, 4 * K + 3>(a2, h));
    rec_tail_store<C, K + 1>(blk, lane, a2, h);
  } else if constexpr (K < R::NTQ + R::NTS) {
    blk[R::TAILS + (K - R::NTQ) * 64 + lane] = rec_tail_get<C, 4 * R::NTQ + (K - R::NTQ)>(a2, h);
    rec_tail_store<C, K + 1>(blk, lane, a2, h);
  }
}
template <class C, int K = 0>
NJ_DEV void rec_tail_load(const float* blk, int lane, float (&a2)[MF<C>::Q1], float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
  if constexpr (K < R::NTQ) {
    const f32x4 v = *(const f32x4*)(blk + R::TAILQ + K * 256 + lane * 4);
    rec_tail_put<C, 4 * K>(v[0], a2, h);
    rec_tail_put<C, 4 * K + 1>(v[1], a2, h);
    rec_tail_put<C, 4 * K + 2>(v[2], a2, h);
    rec_tail_put<C, 4 * K + 3>(v[3], a2, h);
    rec_tail_load<C, K + 1>(blk, lane, a2, h);
  } else if constexpr (K < R::NTQ + R::NTS) {
    rec_tail_put<C, 4 * R::NTQ + (K - R::NTQ)>(blk[R::TAILS + (K - R::NTQ) * 64 + lane], a2, h);
    rec_tail_load<C, K + 1>(blk, lane, a2, h);
  }
}
// one wave holds the whole tile (bulk role): group A = a2 + h, group B = a1
template <class C>
NJ_DEV void rec_store_A(float* blk, int lane, const float (&a2)[MF<C>::Q1], const float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
#pragma unroll
  for (int k = 0; k < R::NF; ++k)
    quad_store(blk + k * 256 + lane * 4, a2[4 * k], a2[4 * k + 1], a2[4 * k + 2], a2[4 * k + 3]);
  rec_tail_store<C>(blk, lane, a2, h);
}
template <class C>
NJ_DEV void rec_store_B(float* blk, int lane, const float (&a1)[MF<C>::Q1]) {
  using R = StepRec<C>;
#pragma unroll
  for (int k = 0; k < R::NF; ++k)
    quad_store(blk + R::GA + k * 256 + lane * 4, a1[4 * k], a1[4 * k + 1], a1[4 * k + 2], a1[4 * k + 3]);
#pragma unroll
  for (int i = 0; i < R::L1; ++i) blk[R::B1S + i * 64 + lane] = a1[4 * R::NF + i];
}
template <class C>
NJ_DEV void rec_load_A(const float* blk, int lane, float (&a2)[MF<C>::Q1], float (&h)[MF<C>::QH]) {
  using R = StepRec<C>;
#pragma unroll
  for (int k = 0; k < R::NF; ++k) {
    const f32x4 v = *(const f32x4*)(blk + k * 256 + lane * 4);
    a2[4 * k] = v[0]; a2[4 * k + 1] = v[1]; a2[4 * k + 2] = v[2]; a2[4 * k + 3] = v[3];
  }
  rec_tail_load<C>(blk, lane, a2, h);
}
template <class C>
NJ_DEV void rec_load_B(const float* blk, int lane, float (&a1)[MF<C>::Q1]) {
  using R = StepRec<C>;
#pragma unroll
  for (int k = 0; k < R::NF; ++k) {
    const f32x4 v = *(const f32x4*)(blk + R::GA + k * 256 + lane * 4);
    a1[4 * k] = v[0]; a1[4 * k + 1] = v[1]; a1[4 * k + 2] = v[2]; a1[4 * k + 3] = v[3];
  }
#pragma unroll
  for (int i = 0; i < R::L1; ++i) a1[4 * R::NF + i] = blk[R::B1S + i * 64 + lane];
}
// a wave-uniform value the compiler cannot see is uniform (reduced through lane permutes, or
// derived from threadIdx): in an SGPR, so that loops on it are scalar loops and loads indexed
// by it are scalar loads
NJ_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// base16_s[i] through the scalar cache (s_load): the table is written by the plan kernels of an
// earlier launch; through a plain pointer the compiler, seeing the kernel's own stores, would
// fetch it with a vector load + s_waitcnt vmcnt(0) + v_readfirstlane
// Everything in flight lands before a step loop starts (its first step needs the prologue's
// loads at once anyway).  The point is the compiler's bookkeeping: at a loop header it merges
// the counter states of the entry and of the back edge, and a prologue whose loads are issued in
// another order than the steady state's turns a counted wait (vmcnt(8): "all but the eight
// stores of this step") into vmcnt(0) for every step of the loop.  (A builtin, not inline asm:
// the waitcnt pass has to see it.)
NJ_DEV void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), others untouched
typedef const long long __attribute__((address_space(4))) * cllp;
NJ_DEV long long sload_ll(const long long* p, int i) { return ((cllp)(unsigned long long)p)[i]; }

// act'(z) from a stored activation; 0 for a dropped unit (stored as -0.0f)
template <int ACT, bool DROP> NJ_DEV float dact_stored(float av) {
  const float d = dact_f<ACT>(av);
  if constexpr (DROP) return __float_as_uint(av) == 0x80000000u ? 0.0f : d;
  else return d;
}

// one xorshift32 word: two 16-bit keep decisions (same stream as njode_mfma.h)
NJ_DEV uint32_t xs32(uint32_t& s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return s;
}

// registers 4*MT0 .. 4*MT0+7 of a hidden layer from the two finished accumulator tiles
// MT0, MT0+1: activation, dropout select (no scale: folded into the consumers' weights),
// bias unit = 1.  Draws words q/2 of the layer in order, like njode_mfma.h.
template <class C, bool DROP, int MT0>
NJ_DEV void hidden_pair(const f32x4& t0, const f32x4& t1, float (&av)[MF<C>::Q1], uint32_t& s,
                        uint32_t thr16, int g) {
  constexpr int Q1 = MF<C>::Q1;
#pragma unroll
  for (int r = 0; r < 8; r += 2) {
    const int q = 4 * MT0 + r;
    if (q < Q1) {
      float v0 = act2_f<C::ACT>(r < 4 ? t0[r & 3] : t1[r & 3]);
      float v1 = q + 1 < Q1 ? act2_f<C::ACT>(r + 1 < 4 ? t0[(r + 1) & 3] : t1[(r + 1) & 3]) : 0.0f;
      if constexpr (DROP) {
        const uint32_t w = xs32(s);
        // a dropped unit is -0.0f: a zero for every product, and the backward (which reads the
        // stored activations instead of recomputing them) can tell it from a kept unit at 0
        v0 = (w & 0xffffu) >= thr16 ? v0 : -0.0f;
        v1 = (w >> 16) >= thr16 ? v1 : -0.0f;
      }
      av[q] = v0;
      if (q + 1 < Q1) av[q + 1] = v1;
    }
  }
  constexpr int QB = MF<C>::W / 4, GB = MF<C>::W % 4;
  if constexpr (QB >= 4 * MT0 && QB < 4 * MT0 + 8) av[QB] = g == GB ? 1.0f : av[QB];
}

// A x b for the output tiles [MT0, MT0 + 2) over all Q k-steps, two accumulators alternating.
// ET >= 0: tile ET stands for ER edge rows (njode_mfma.h): A[ET][q] holds their 4x4x1 operands
template <int MT, int Q, int MT0, int ET = -1, int ER = 0>
NJ_DEV void mfma_pair(const float (&A)[MT][Q], const float (&bv)[Q], f32x4& t0, f32x4& t1, int g = 0) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  t0 = z;
  t1 = z;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if constexpr (MT0 == ET) t0 = mfma1(A[MT0][q], bv[q], t0);
    else t0 = mfma4(A[MT0][q], bv[q], t0);
    if constexpr (MT0 + 1 < MT) {
      if constexpr (MT0 + 1 == ET) t1 = mfma1(A[MT0 + 1][q], bv[q], t1);
      else t1 = mfma4(A[MT0 + 1][q], bv[q], t1);
    }
  }
  if constexpr (MT0 == ET) t0 = edge_tile<ER>(t0, g);
  if constexpr (MT0 + 1 == ET && MT0 + 1 < MT) t1 = edge_tile<ER>(t1, g);
}

template <class C> struct Ode2FwdFrags {
  using M = MF<C>;
  float A1[M::MT1][M::Q0], A2[M::MT1][M::Q1], A3[M::MTH][M::Q1];
  NJ_DEV void load(const float* frag, int lane) {
#pragma unroll
    for (int mt = 0; mt < M::MT1; ++mt) {
      // (the edge tile: the 4x4x1 operands of its rows, taken from the same table)
      const int l = mt == EdgeRows<M::W>::TILE ? edge_lane(lane) : lane;
#pragma unroll
      for (int q = 0; q < M::Q0; ++q) A1[mt][q] = frag[(M::F1 + mt * M::Q0 + q) * 64 + l];
#pragma unroll
      for (int q = 0; q < M::Q1; ++q) A2[mt][q] = frag[(M::F2 + mt * M::Q1 + q) * 64 + l];
    }
#pragma unroll
    for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
      for (int q = 0; q < M::Q1; ++q) A3[mt][q] = frag[(M::F3 + mt * M::Q1 + q) * 64 + lane];
  }
};

// hidden layer: all MT1 output tiles in pairs, activation of each pair right behind it
template <class C, bool DROP, int QIN, int MT0 = 0>
NJ_DEV void hidden_layer2(const float (&A)[MF<C>::MT1][QIN], const float (&bv)[QIN],
                          float (&av)[MF<C>::Q1], uint32_t& st, uint32_t thr16, int g) {
  if constexpr (MT0 < MF<C>::MT1) {
    f32x4 t0, t1;
    mfma_pair<MF<C>::MT1, QIN, MT0, EdgeRows<MF<C>::W>::TILE, EdgeRows<MF<C>::W>::R>(A, bv, t0, t1, g);
    hidden_pair<C, DROP, MT0>(t0, t1, av, st, thr16, g);
    hidden_layer2<C, DROP, QIN, MT0 + 2>(A, bv, av, st, thr16, g);
  }
}

// f = A3 a2 (+ bias through the unit-1 column): the H-wide output, two accumulators per tile
template <class C>
NJ_DEV void out_layer2(const float (&A3)[MF<C>::MTH][MF<C>::Q1], const float (&a2)[MF<C>::Q1],
                       f32x4 (&f)[MF<C>::MTH]) {
  using M = MF<C>;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mt = 0; mt < M::MTH; ++mt) {
    f32x4 e = z, o = z;
#pragma unroll
    for (int q = 0; q < M::Q1; q += 2) {
      e = mfma4(A3[mt][q], a2[q], e);
      if (q + 1 < M::Q1) o = mfma4(A3[mt][q + 1], a2[q + 1], o);
    }
    f[mt] = e + o;
  }
}

// start state of an item from the encoder (ENC of ode2_fwd_single): h = encoder_map(X of the
// start row) + identity path, D-layout; stored as h0row[start row] (start values: h0start[path])
// for the kernels behind the forward
// forward fragments of the encoder only (S::NFWD vectors), one copy per block
template <class C> struct EncFwdLds {
  using S = typename EncS<C>::type;
  static constexpr int FLOATS = S::NFWD * 64;
  static NJ_DEV void stage(lfp img, const float* frag_enc, int tid, int nthreads) {
    for (int i = tid; i < FLOATS; i += nthreads) img[i] = frag_enc[i];
  }
};
template <class C, bool DROP>
NJ_DEV void ode2_item_start(const KArgs& a, const Item<C>& it, bool valid, lfp enc_img, int lane,
                            float (&h)[MF<C>::QH]) {
  using S = typename EncS<C>::type;
  static_assert(S::QO == MF<C>::QH && S::MTO == MF<C>::MTH, "encoder output = ODE state");
  const int g = lane >> 4;
  const bool is_row = it.prev >= 0;
  const int pv = is_row ? it.prev : 0;
  const float* xp = is_row ? a.X + (size_t)pv * C::D : a.start_X + (size_t)it.b * C::D;
  float b0[S::Q0], a1[S::Q1], a2[S::Q1];
  enc_input<C, S>(xp, b0, g);
  uint32_t k1, k2;
  row_keep_bits<DROP>(a, a.gid0 + it.b, is_row ? (uint32_t)a.k_jump[a.t_of_row[pv]] : TKEY_START, NET_ENC, g,
                      S::Q1, k1, k2);
  LdsFrags<S> F;
  F.init(enc_img, lane);
  f32x4 out[S::MTO];
  mnet_fwd<S, C::ACT, DROP>(F, b0, a1, a2, out, k1, k2, a.dc.inv_keep, g);
  float* const trash = a.trash + lane * C::H;
  float* dstrow = valid ? (is_row ? a.h0row + (size_t)pv * C::H : a.h0start + (size_t)it.b * C::H) : trash;
#pragma unroll
  for (int q = 0; q < S::QO; ++q) {
    const int u = 4 * q + g;
    const float v = out[q / 4][q % 4] + enc_residual<C>(xp, u < C::H ? u : 0);
    h[q] = u < C::H ? v : 0.0f;
    float* dst = u < C::H ? dstrow + u : trash;
    *dst = v;
  }
}

// B (v2): Euler evolve of every item; worker `wave` of `n_waves` walks the tiles
// [tile0, tile1) in snake order.  Same contract as ode_fwd_single (njode_mfma.h) with the
// scaled fragment table a.frag2.
// SAVE (compile time: checkpoints + activations are stored without a branch, so the compiler
// can COUNT them -- with a runtime `if` around the stores every `s_waitcnt vmcnt` that waits for
// the next step's prefetched scalars conservatively also drained the stores: +13 % on the kernel)
// ENC (round 5, NJODE_ENC_FUSED=1; VERDICT r4 item 4's structural option): an item starts at
// encoder(X of its start row) -- one encoder evaluation per item -- so the wave evaluates it itself
// at the head of the item (fragments from the block's LDS image `enc_img`, the S::NFWD forward
// vectors of the encoder; same matrix instructions, same dropout words as k_encode_rows_mfma: the
// same numbers) and stores h0row[start row] for the row pass, instead of reading what a separate
// launch over all rows wrote.  k_encode_rows_items then only covers the start rows of the four-wave
// role's tiles and every path's LAST row (no item starts there).
template <class C, bool DROP, bool TAIL, bool SAVE, bool ENC = false>
NJ_DEV void ode2_fwd_single(const KArgs& a, int lane, int wave, int n_waves, int tile0, int tile1,
                            lfp enc_img = nullptr) {
  static_assert(!(TAIL && SAVE), "tail items are never checkpointed");
  static_assert(!(TAIL && ENC), "tails read the stored state after their last observation");
  using M = MF<C>;
  const int g = lane >> 4, c = lane & 15;
  Ode2FwdFrags<C> F;
  F.load(a.frag2, lane);
  In0Const<C> K0;
  K0.init(g);

  const int n_items = TAIL ? a.B : a.n_obs;
  const int n_tiles = tile1 - tile0;
  float* const trash = a.trash + lane * C::H;
  for (int round = 0; round * n_waves < n_tiles; ++round) {
    const int rel = snake_tile(round, wave, n_waves);
    if (rel >= n_tiles) continue;
    const int tile = tile0 + rel;
    const int j = tile * 16 + c;
    const bool valid = j < n_items;
    Item<C> it;
    it.template load<TAIL>(a, j, valid);
    float h[M::QH];
    if constexpr (ENC) {
      ode2_item_start<C, DROP>(a, it, valid, enc_img, lane, h);
    } else {
      const float* h0 = it.h0(a);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        h[q] = u < C::H ? h0[u < C::H ? u : 0] : 0.0f;
      }
    }
    // (uniform: the step loop is a scalar loop, base16_s[s] a scalar load; the per-lane
    // schedule values of the next step are loaded one step ahead and carried RAW across the
    // back edge -- a select next to a load puts its s_waitcnt there)
    const int nmax = uniform(wave_max(it.n));
    float dt_r = 0.0f, t_r = 0.0f;
    long long b16_n = 0;
    if (nmax > 0) {
      const int k0 = it.n > 0 ? it.kbeg : 0;
      dt_r = a.step_dt[k0];
      t_r = a.step_t[k0];
      b16_n = SAVE ? sload_ll(a.base16_s, 0) : 0;
    }
    if constexpr (SAVE) {
      if (a.item_pack) {   // (for the backward's tile prologue: Item::store_pack)
        if (g == 0) it.store_pack(a.item_pack, j);
        const long long bl = sload_ll(a.base16_s, nmax > 0 ? nmax - 1 : 0);
        if (lane == 0) a.tile_last[tile] = bl;
      }
    }
    vm_drain();
    for (int s = 0; s < nmax; ++s) {
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      const float dt = active ? dt_r : 0.0f, t = t_r;
      const long long b16 = b16_n;
      {
        const int sn = s + 1 < nmax ? s + 1 : s;
        const int kn = sn < it.n ? it.kbeg + sn : 0;
        dt_r = a.step_dt[kn];
        t_r = a.step_t[kn];
        if constexpr (SAVE) b16_n = sload_ll(a.base16_s, sn);
      }
      float b0[M::Q0];
      in0_fill_c<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g, K0);
      uint32_t st = 0;
      if constexpr (DROP) {
        const unsigned long long gid = a.gid0 + it.b;
        st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                        (uint32_t)k, NET_ODE);
      }
      float a1[M::Q1], a2[M::Q1];
      hidden_layer2<C, DROP, M::Q0>(F.A1, b0, a1, st, a.dc.thr16, g);
      float* blk = nullptr;
      if constexpr (SAVE) {
        blk = rec_block<C>(a.act, b16, tile);
        rec_store_B<C>(blk, lane, a1);
      }
      hidden_layer2<C, DROP, M::Q1>(F.A2, a1, a2, st, a.dc.thr16, g);
      if constexpr (SAVE) rec_store_A<C>(blk, lane, a2, h);
      f32x4 f[M::MTH];
      out_layer2<C>(F.A3, a2, f);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = fmaf(dt, f[q / 4][q % 4], h[q]);  // dt = 0: inactive
    }
    float* out = valid ? (TAIL ? a.hT + (size_t)it.b * C::H : a.h_end + (size_t)it.r * C::H) : trash;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      float* dst = u < C::H ? out + u : trash;
      *dst = h[q];
    }
  }
}



// C (stored activations): reverse Euler sweep of every segment + d loss / d ODE params.
// Same contract and worker layout as ode_bwd_single (njode_mfma.h); per Euler step the two
// hidden activation vectors are LOADED (prefetched one step ahead with the checkpoint), so
// only the transposed products (77 MFMAs) and the weight-gradient products (96) remain, the
// LDS holds only the transposed fragments, and no dropout stream is drawn.
template <class C> struct OdeLdsFragsT {
  using M = MF<C>;
  static constexpr int NVEC = M::NALL - M::NFWD;
  lfp base, cur;
  static NJ_DEV void stage(lfp img, const float* frag, int tid, int nthreads) {
    for (int i = tid; i < NVEC * 64; i += nthreads) img[i] = frag[M::NFWD * 64 + i];
  }
  static constexpr int ET = EdgeRows<M::W>::TILE;   // (the W-row products W3^T, W2^T: edge rows)
  NJ_DEV void init(lfp img, int lane) { base = img + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float b3(int mt, int q) const { return cur[(M::B3 - M::NFWD + mt * M::QH + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(M::B2 - M::NFWD + mt * M::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(M::B1 - M::NFWD + mt * M::QW + q) * 64]; }
  // 4x4x1 operands of the edge rows: the edge tile's vectors read at lane edge_lane(lane)
  NJ_DEV float e3(int q, int de) const { return cur[(M::B3 - M::NFWD + ET * M::QH + q) * 64 + de]; }
  NJ_DEV float e2(int q, int de) const { return cur[(M::B2 - M::NFWD + ET * M::QW + q) * 64 + de]; }
};
// image layout of the one-wave backward's weight-gradient products (njode_mfma.h)
#ifndef NJ_IMG_SWZ
#define NJ_IMG_SWZ 1   // (round 5: conflict-free, k_ode_bwd_mixed -0.5 .. -0.9 %, profiles/r05_lds_conflicts.txt; 0: padded rows, A/B)
#endif
#if NJ_IMG_SWZ
using BwdIL = ImgSwz;
#else
using BwdIL = ImgPad;
#endif
template <class C> struct OdeBwdActLds {
  using M = MF<C>;
  static constexpr int NG = M::MTH * ((M::W + 1 + 15) / 16) + M::MT1 * ((M::W + 1 + 15) / 16) +
                            M::MT1 * ((M::IN0 + 1 + 15) / 16);
  static constexpr int BODY = 4 * 2 * BwdIL::FLOATS + OdeLdsFragsT<C>::NVEC * 64;
  static constexpr int RED = 3 * NG * 64 * 4;      // the block's final tile reduction (upper bound: NG of the no-edge form)
  static constexpr int FLOATS = BODY > RED ? BODY : RED;
};
// (-DNJ_BWD_ABL=bits, tools/ubench/bwd_ablate.sh: parts of the Euler-step loop switched off for
// timing -- 1 weight-gradient products, 2 transposed products, 4 wave-level LDS syncs, 8 the loads
// of the next step, 16 image writes, 32 act' multiplies, 64 the whole Euler-step loop, 128 the
// tile reduction and slab flush; the product build has none of it)
#ifndef NJ_BWD_ABL
#define NJ_BWD_ABL 0
#endif
#define BWD_ABL(bit) ((NJ_BWD_ABL) & (bit))
// tiles of dW2 held as 16x16 accumulators: with edge rows (dw_accumulate_edge) ET x ET + GM, GN
template <class C> struct OdeG2 {
  using M = MF<C>;
  static constexpr int ET = EdgeRows<M::W>::TILE, ER = EdgeRows<M::W>::R;
  static constexpr int MT = ER ? ET : M::MT1, NT = ER ? ET : (M::W + 1 + 15) / 16;
  static constexpr int NE = ER ? 4 : 0;            // edge accumulators (GM[2], GN[2])
};
// flush of a 256-thread block's four workers (the bulk role): their register tiles are summed
// through LDS in fixed order and stored as ONE slab row
template <class C, bool DROP>
NJ_DEV void ode3_flush(const KArgs& a, lfp lds_raw, f32x4 (&G3)[MF<C>::MTH][(MF<C>::W + 1 + 15) / 16],
                       f32x4 (&G2)[OdeG2<C>::MT][OdeG2<C>::NT], f32x4 (&GM)[2], f32x4 (&GN)[2],
                       f32x4 (&G1)[MF<C>::MT1][(MF<C>::IN0 + 1 + 15) / 16], int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using E2 = OdeG2<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16, NT0 = (M::IN0 + 1 + 15) / 16;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  if constexpr (E2::ER != 0) {   // (two accumulators each: independent 4x4x1 chains)
    GM[0] += GM[1];
    GN[0] += GN[1];
  }
  // ---- flush (as ode_bwd_single): one slab row per block
  constexpr int NG = M::MTH * NT1 + E2::MT * E2::NT + (E2::ER ? 2 : 0) + M::MT1 * NT0;
  static_assert(3 * NG * 64 * 4 <= OdeBwdActLds<C>::FLOATS, "tile reduction does not fit the LDS");
  __syncthreads();
  f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds_raw;
  auto for_tiles = [&](auto f) {
    int i = 0;
#pragma unroll
    for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) f(G3[mt][nt], i++);
#pragma unroll
    for (int mt = 0; mt < E2::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < E2::NT; ++nt) f(G2[mt][nt], i++);
    if constexpr (E2::ER != 0) {
      f(GM[0], i++);
      f(GN[0], i++);
    }
#pragma unroll
    for (int mt = 0; mt < M::MT1; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT0; ++nt) f(G1[mt][nt], i++);
  };
  if (wv > 0) for_tiles([&](f32x4& t, int i) { red[((wv - 1) * NG + i) * 64 + lane] = t; });
  __syncthreads();
  if (wv != 0) return;
  for_tiles([&](f32x4& t, int i) {
    t += red[(0 * NG + i) * 64 + lane];
    t += red[(1 * NG + i) * 64 + lane];
    t += red[(2 * NG + i) * 64 + lane];
  });
  // the stored activations carry no inverted-dropout factor: it goes on once, here
  const float ik = DROP ? a.dc.inv_keep : 1.0f;
  float* slab = a.slab + (size_t)slab_row * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
#pragma unroll
  for (int mt = 0; mt < E2::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int uo = 16 * mt + 4 * g + r;
      if (uo < M::W) {
#pragma unroll
        for (int nt = 0; nt < E2::NT; ++nt) {
          const int ui = 16 * nt + c;
          if (ui < M::W) W2[uo * M::W + ui] = ik * G2[mt][nt][r];
          else if (ui == M::W) b2[uo] = G2[mt][nt][r];
        }
      }
    }
  if constexpr (E2::ER != 0) {
    // GM: register i of lane l = dW2[16 ET + i][l]; GN: = dW2[l][16 ET + i] for l < 16 ET
#pragma unroll
    for (int i = 0; i < E2::ER; ++i) {
      const int uo = 16 * E2::ET + i;
      if (lane < M::W) W2[uo * M::W + lane] = ik * GM[0][i];
      else if (lane == M::W) b2[uo] = GM[0][i];
    }
    if (lane < 16 * E2::ET) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ui = 16 * E2::ET + i;
        if (ui < M::W) W2[lane * M::W + ui] = ik * GN[0][i];
        else if (ui == M::W) b2[lane] = GN[0][i];
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < M::MT1; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int uo = 16 * mt + 4 * g + r;
      if (uo < M::W) {
#pragma unroll
        for (int nt = 0; nt < NT0; ++nt) {
          const int ui = 16 * nt + c;
          if (ui < M::IN0) W1[uo * M::IN0 + M::col0(ui)] = G1[mt][nt][r];
          else if (ui == M::IN0) b1[uo] = G1[mt][nt][r];
        }
      }
    }
#pragma unroll
  for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int uo = 16 * mt + 4 * g + r;
      if (uo < C::H) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          const int ui = 16 * nt + c;
          if (ui < M::W) W3[uo * M::W + ui] = ik * G3[mt][nt][r];
          else if (ui == M::W) b3[uo] = G3[mt][nt][r];
        }
      }
    }
}

// ---- tile queue (round 5; NJODE_BWD_QUEUE=1, not the default) ----------------------------------
// The backward's gradient accumulators persist across the tiles of a worker, so its blocks cannot
// be handed out by the dispatcher as the forward's are: rounds 2-4 give every worker its tiles
// statically (snake order over 1 024 blocks, two per CU resident: the second half of the blocks
// starts as first-half blocks retire, and a third of the wave slots' time is idle at the end of
// the launch).  With the queue the launch is exactly the resident blocks and every worker pops
// the next tile -- longest first, the tiles are sorted -- from a counter in the workspace:
// KArgs::tile_q[0] for the tiles [0, T) of the four-wave role, [1] for the bulk [T, n_tiles), [2]
// counts finished blocks (the last one clears all three for the next launch; the saving forward
// clears them once, too).  Measured (profiles/r05_bwd_fixed_costs.txt): the idle time goes (34 %
// -> 6 % of wave time) and the launch gets 0.7 % shorter at 20 000 paths, 5 % longer at 125 000,
// because the Euler-step loop is bound by the SIMD's pipe: kept as a switch, the static rounds stay the default -- they also keep the
// gradient bitwise reproducible (with the queue, which tiles meet in one accumulator depends on
// timing: equal to fp32 summation order only).
// (the pop is issued WITHOUT a wait -- lane 0's register holds the counter's old value once the
// atomic has returned; queue_value() reads it when the next tile starts, a whole tile later.
// profiles/r05_bwd_fixed_costs.txt: waited for at once, the round trip of 1 920 waves popping one
// address was 4.1 us per tile, and the burst at launch -- every wave popping its first tile at the
// same instant -- is why the first tile of a worker is static: tile = worker index, pops count on
// from the number of workers)
NJ_DEV int queue_pop_issue(int* q) {
  int t = 0;
  if ((threadIdx.x & 63) == 0) t = atomicAdd(q, 1);
  return t;
}
NJ_DEV int queue_value(int raw) { return __builtin_amdgcn_readfirstlane(raw); }
NJ_DEV void queue_block_done(int* tile_q, int n_blocks) {   // call once per block, all threads
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(tile_q + 2, 1) == n_blocks - 1) {
      tile_q[0] = 0;
      tile_q[1] = 0;
      tile_q[2] = 0;
      __threadfence();
    }
  }
}
#ifdef NJ_BWD_STAMPS
// diagnostic build (tools/ubench/bwd_stamps.sh): per wave of k_ode_bwd_mixed the 100 MHz wall
// clock at kernel entry, after the prologue, after the last tile, after the flush; tiles and
// Euler steps done; hardware id; role; [8] sum over tiles of (tile start -> first Euler step), [9] of
// the step loops, [10] of the queue pops
__device__ unsigned long long g_bwd_stamps[8192 * 16];
#define BWD_STAMP(slot, val) \
  do { if ((threadIdx.x & 63) == 0) g_bwd_stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (slot)] = (val); } while (0)
#else
#define BWD_STAMP(slot, val) do {} while (0)
#endif

template <class C, bool DROP, bool QUEUE = false>
NJ_DEV void ode3_bwd_single(const KArgs& a, lfp lds_raw, int wave, int n_waves, int tile0, int tile1,
                            int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using FR = OdeLdsFragsT<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16;     // column tiles of [a, 1]
  constexpr int NT0 = (M::IN0 + 1 + 15) / 16;   // column tiles of [in0, 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  using IL = BwdIL;
  lfp img_d = lds_raw + wv * 2 * IL::FLOATS, img_a = img_d + IL::FLOATS;
  lfp fimg = lds_raw + 4 * 2 * IL::FLOATS;
  FR::stage(fimg, a.frag2, threadIdx.x, 256);
  for (int i = threadIdx.x; i < 4 * 2 * IL::FLOATS; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  FR F;
  F.init(fimg, lane);
  In0Const<C> K0;
  K0.init(g);

  using E2 = OdeG2<C>;
  constexpr int ET = E2::ER ? E2::ET : -1, ER = E2::ER;
  const int de = edge_lane(lane) - lane;          // edge operands: the fragment vector at that lane
  f32x4 G3[M::MTH][NT1], G2[E2::MT][E2::NT], GM[2], GN[2], G1[M::MT1][NT0];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  GM[0] = GM[1] = GN[0] = GN[1] = zero4;
#pragma unroll
  for (int i = 0; i < M::MTH; ++i)
#pragma unroll
    for (int n = 0; n < NT1; ++n) G3[i][n] = zero4;
#pragma unroll
  for (int i = 0; i < E2::MT; ++i)
#pragma unroll
    for (int n = 0; n < E2::NT; ++n) G2[i][n] = zero4;
#pragma unroll
  for (int i = 0; i < M::MT1; ++i) {
#pragma unroll
    for (int n = 0; n < NT0; ++n) G1[i][n] = zero4;
  }
  float* const trash = a.trash + threadIdx.x * C::H;
  const int n_tiles = tile1 - tile0;
  BWD_STAMP(1, wall_clock64());
  int q_raw = 0;                        // (QUEUE) the pending pop: the tile after this one
  int q_rel = wave;                     // the first tile of a worker is static
  int n_done = 0, n_steps_done = 0;
#ifdef NJ_BWD_STAMPS
  unsigned long long t_pro = 0, t_loop = 0, t_pop = 0;
#endif
  for (int round = 0; QUEUE ? q_rel < n_tiles : round * n_waves < n_tiles; ++round) {
#ifdef NJ_BWD_STAMPS
    const unsigned long long tt0 = wall_clock64();
#endif
    int rel;
    if constexpr (QUEUE) {
      rel = q_rel;
    } else {
      rel = snake_tile(round, wave, n_waves);
      if (rel >= n_tiles) continue;
    }
    const int tile = tile0 + rel;
    const int j = tile * 16 + c;
    const bool valid = j < a.n_obs;
    Item<C> it;
    long long b16_last = 0;
    const bool packed = a.item_pack != nullptr;   // (uniform)
    if (packed) {
      // (round 6: ONE load deep -- the forward left the item and the record base of the tile's last step)
      b16_last = sload_ll(a.tile_last, tile);
      it.load_pack(a.item_pack, j, valid);
    } else {
      it.template load<false>(a, j, valid);
    }
    float lam[M::QH];
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      const float v = a.lam_end[(size_t)it.r * C::H + (u < C::H ? u : 0)];
      lam[q] = (valid && u < C::H) ? v : 0.0f;
    }
    // (uniform: scalar step loop, base16_s[s] a scalar load)
    const int nmax = BWD_ABL(64) ? 0 : uniform(wave_max(it.n));
    ++n_done;
    n_steps_done += nmax;
    // The record of the NEXT step (s - 1) is loaded INTO the registers of the current one as
    // soon as those are dead (a2 and h after delta2, a1 after delta1: no second register set,
    // no copies, waits spread over the step).  Loads are unconditional (step 0 reloads itself)
    // and their values cross the back edge RAW, so that the compiler counts them and no wait
    // sits next to a load; the record's address hangs on scalar loads only.
    auto load_sched = [&](int s, float& dtr, float& tr) {
      const int kk = s < it.n ? it.kbeg + s : 0;
      dtr = a.step_dt[kk];
      tr = a.step_t[kk];
    };
    float h[M::QH], a1[M::Q1], a2[M::Q1], dt_r = 0.0f, t_r = 0.0f;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) h[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < M::Q1; ++q) { a1[q] = 0.0f; a2[q] = 0.0f; }
    if (nmax > 0) {   // (in the order of the steady state: the waits at the loop head count on it)
      const float* blk = rec_block<C>(a.act, packed ? b16_last : sload_ll(a.base16_s, nmax - 1), tile);
      load_sched(nmax - 1, dt_r, t_r);
      rec_load_A<C>(blk, lane, a2, h);
      rec_load_B<C>(blk, lane, a1);
    }
#ifdef NJ_BWD_STAMPS
    vm_drain();
    const unsigned long long tt1 = wall_clock64();
    t_pro += tt1 - tt0;
#endif
    // (the pop for the tile after this one: issued behind the prologue's loads -- vmcnt retires in
    // order, in front of them the atomic's round trip sits on the chain order -> item -> record the
    // first Euler step waits for: 9.0 instead of 5.4 us per tile -- and read at the tile's end.  The
    // loop is written as `while (tile in range) { ...; tile = popped; }`: the first form, `for (;;) {
    // tile = popped; if (out of range) break; ... }`, cost this kernel 72 spilled registers and 16
    // scratch reloads per Euler step -- same live values, another loop shape for the allocator)
    if constexpr (QUEUE) q_raw = queue_pop_issue(a.tile_q + 1);
    for (int s = nmax - 1; s >= 0; --s) {
      const float dt = s < it.n ? dt_r : 0.0f, t = t_r;
      const int sp = s > 0 ? s - 1 : 0;
      const float* blkp = rec_block<C>(a.act, sload_ll(a.base16_s, sp), tile);
      float b0[M::Q0];
      in0_fill_c<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g, K0);
      if (!BWD_ABL(8)) load_sched(sp, dt_r, t_r);
      F.begin();

      // ---- layer 3: h' = h + dt f  =>  delta3 = dt * lam (zero for inactive chains)
      float d3[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam[q];
      if (!BWD_ABL(16)) {
        img_write<M::QH, IL>(img_d, d3, g, c);
        img_write<M::Q1, IL>(img_a, a2, g, c);
      }
      if (!BWD_ABL(4)) wave_lds_sync();
      if (!BWD_ABL(1)) dw_accumulate<M::MTH, NT1, IL>(img_d, img_a, G3, g, c);
      f32x4 acc[M::MT1];
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
      if (!BWD_ABL(2)) {
#pragma unroll
        for (int q = 0; q < M::QH; ++q)
#pragma unroll
          for (int mt = 0; mt < M::MT1; ++mt)
            acc[mt] = mt == ET ? mfma1(F.e3(q, de), d3[q], acc[mt]) : mfma4(F.b3(mt, q), d3[q], acc[mt]);
        if constexpr (ER != 0) acc[ET] = edge_tile<ER>(acc[ET], g);
      } else {
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt][0] = d3[0];
      }
      float d2[M::QW];
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
        d2[q] = acc[q / 4][q % 4] * (BWD_ABL(32) ? 1.0f : dact_stored<C::ACT, DROP>(a2[q]));
      if (!BWD_ABL(8)) rec_load_A<C>(blkp, lane, a2, h);
      if (!BWD_ABL(4)) wave_lds_sync();

      // ---- layer 2
      if (!BWD_ABL(16)) {
        img_write<M::QW, IL>(img_d, d2, g, c);
        img_write<M::Q1, IL>(img_a, a1, g, c);
      }
      if (!BWD_ABL(4)) wave_lds_sync();
      if (!BWD_ABL(1)) {
        if constexpr (ER != 0) dw_accumulate_edge<E2::MT, IL>(img_d, img_a, G2, GM, GN, lane, g, c);
        else dw_accumulate<E2::MT, E2::NT, IL>(img_d, img_a, G2, g, c);
      }
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
      if (!BWD_ABL(2)) {
#pragma unroll
        for (int q = 0; q < M::QW; ++q)
#pragma unroll
          for (int mt = 0; mt < M::MT1; ++mt)
            acc[mt] = mt == ET ? mfma1(F.e2(q, de), d2[q], acc[mt]) : mfma4(F.b2(mt, q), d2[q], acc[mt]);
        if constexpr (ER != 0) acc[ET] = edge_tile<ER>(acc[ET], g);
      } else {
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt][0] = d2[mt];
      }
      float d1[M::QW];
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
        d1[q] = acc[q / 4][q % 4] * (BWD_ABL(32) ? 1.0f : dact_stored<C::ACT, DROP>(a1[q]));
      if (!BWD_ABL(8)) rec_load_B<C>(blkp, lane, a1);
      if (!BWD_ABL(4)) wave_lds_sync();

      // ---- layer 1
      if (!BWD_ABL(16)) {
        img_write<M::QW, IL>(img_d, d1, g, c);
        img_write<M::Q0, IL>(img_a, b0, g, c);
      }
      if (!BWD_ABL(4)) wave_lds_sync();
      if (!BWD_ABL(1)) dw_accumulate<M::MT1, NT0, IL>(img_d, img_a, G1, g, c);
      f32x4 acch[M::MTH];
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) {
        f32x4 e = zero4, o = zero4;
        if (!BWD_ABL(2)) {
#pragma unroll
          for (int q = 0; q < M::QW; q += 2) {
            e = mfma4(F.b1(mt, q), d1[q], e);
            if (q + 1 < M::QW) o = mfma4(F.b1(mt, q + 1), d1[q + 1], o);
          }
        } else {
          e[0] = d1[0];
        }
        acch[mt] = e + o;
      }
      // adjoint of the state: lam += (W1^T delta1)[h rows] * (1 - tanh(h)^2)
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const float th = b0[q];  // = tanh(h) wherever unit 4q + g < H
        const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
        lam[q] = fmaf(acch[q / 4][q % 4], dth, lam[q]);
      }
      if (!BWD_ABL(4)) wave_lds_sync();
    }
#ifdef NJ_BWD_STAMPS
    t_loop += wall_clock64() - tt1;
#endif
    float* out = valid ? a.lam_start + (size_t)it.r * C::H : trash;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      float* dst = u < C::H ? out + u : trash;
      *dst = lam[q];
    }
    if constexpr (QUEUE) {   // the tile after this one: popped when this one started
#ifdef NJ_BWD_STAMPS
      const unsigned long long tq0 = wall_clock64();
#endif
      q_rel = n_waves + queue_value(q_raw);
#ifdef NJ_BWD_STAMPS
      t_pop += wall_clock64() - tq0;
#endif
    }
  }

#ifdef NJ_BWD_STAMPS
  BWD_STAMP(8, t_pro);
  BWD_STAMP(9, t_loop);
  BWD_STAMP(10, t_pop);
#endif
  BWD_STAMP(2, wall_clock64());
  BWD_STAMP(4, (unsigned long long)n_done);
  BWD_STAMP(5, (unsigned long long)n_steps_done);
  (void)n_done; (void)n_steps_done; (void)q_rel;
  if (BWD_ABL(128)) return;
  ode3_flush<C, DROP>(a, lds_raw, G3, G2, GM, GN, G1, slab_row);
  BWD_STAMP(3, wall_clock64());
}


}  // namespace njode
