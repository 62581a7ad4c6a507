// njode_mfma_lock4.h -- the lockstep plan of the MASKED (PhysioNet-shaped) models with one
// tile of 16 paths spread over the FOUR waves (= four SIMDs) of a 256-thread block.
//
// Why: a masked model feeds its own prediction back at every jump (models.py:465-467), so a
// path is one serial chain of n_steps Euler steps and ~n_obs jumps; physionet_train.py runs
// 3 000 steps at B = 50, i.e. FOUR tiles on the whole chip.  The one-wave kernels of
// njode_mfma_lockstep.h issue all 179 (step) / 445 (jump) MFMAs of an evaluation from one
// SIMD, every A-fragment re-read from LDS / L2 each time, and carry every per-path vector
// (x, mask, y: 41 floats each) in full in every lane: ~10 us per step forward, ~20 us backward
// (profiles/r01_config5_bench.jsonl).  Here
//
//   * wave w owns output tile w (units 16w .. 16w+15) of every layer of the three networks and
//     the same slice of every per-path vector (h, x, mask, y, the adjoint, ...): 4 registers
//     per lane, lane (g, c) = unit 16w + 4r + g of path c;
//   * its A-fragments are register resident (the launch bound gives a wave the whole 512-entry
//     register file: 132 fragments forward, 217 in the adjoint sweep);
//   * a layer's input is all-gathered through an LDS image [unit][path] -- one s_barrier per
//     exchange, 3 per Euler step forward, 4 in the sweep;
//   * per-path scalars of the loss are reduced across lane groups and waves in a fixed order.
//
// The kernels write exactly the buffers the one-wave kernels write (ltraj, src_row, h_end, y_row,
// ybj_row, hT, loss_terms; lam_traj, g_y, g_ybj, g_hnew, g_hstart), so pass 2 of the backward
// (k_ode_dw_pairs_mfma, k_dec_dw_rows_mfma, k_enc_dw_rows_mfma) is unchanged.  Dropout masks are
// those of the one-wave kernels.  return_path calls stay on k_paths_fwd_mfma.
#pragma once
#include "njode_mfma_lockstep.h"
#include "njode_mfma_split.h"

namespace njode {

template <class C, bool TWO = (C::NH == 2)> struct Q4Ok { static constexpr bool value = false; };
template <class C> struct Q4Ok<C, true> {
  static constexpr bool value =
      C::MASKED && !C::RNN && MF<C>::MT1 == 4 && C::D == C::DO && C::H <= 64 && C::D <= 64 &&
      MF<C>::MTB1 <= 8 && (C::ENC_CASE == 0 || (C::ENC_CASE == 1 && C::D == C::H)) &&
      (C::DEC_CASE == 0 || (C::DEC_CASE == 1 && C::DO == C::H));
};

// the ODE network's fragment table (MF<C>) under the names of MS<>
template <class C> struct OdeQS {
  using M = MF<C>;
  static constexpr int IN = M::IN0, OUT = C::H, W = C::W;
  static constexpr int Q0 = M::Q0, Q1 = M::Q1, QO = M::QH, QW = M::QW;
  static constexpr int MT1 = M::MT1, MTO = M::MTH, MTI = M::MTB1;
  static constexpr int F1 = M::F1, F2 = M::F2, F3 = M::F3, B3 = M::B3, B2 = M::B2, B1 = M::B1;
};

constexpr int q4_img_rows(int q) { return (4 * q + 15) / 16 * 16; }

// Stored hidden activations of the ODE network (the forward stores what the adjoint sweep would
// otherwise recompute with 35 more MFMAs, a fragment set that no longer fits the registers and
// one more exchange per step): per Euler step and tile, wave w's own 2 x 4 registers, each
// register one coalesced 256-B line.  Values are those the next layer consumed (dropout applied).
constexpr int Q4_ACT_FLOATS = 4 * 8 * 64;
NJ_DEV float* q4_act_ptr(float* lact, int k, int n_tiles, int tile, int w, int lane) {
  return lact + (((size_t)k * n_tiles + tile) * 4 + w) * 8 * 64 + lane;
}

// This wave's fragments of one network, in registers (the networks every step / every jump
// twice needs: ODE, readout) or in a wave-private LDS region (the encoder: once per jump).
// Forward: W1, W2 tile w, W3 tile w.
template <class S> struct Q4FwdReg {
  float A1[S::Q0], A2[S::Q1], A3[S::Q1];
  NJ_DEV void load(const float* frag, int w, int lane) {
    static_assert(S::MT1 == 4, "four hidden tiles, one per wave");
#pragma unroll
    for (int q = 0; q < S::Q0; ++q) A1[q] = frag[(S::F1 + w * S::Q0 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < S::Q1; ++q) A2[q] = frag[(S::F2 + w * S::Q1 + q) * 64 + lane];
    const int wo = w < S::MTO ? w : 0;
#pragma unroll
    for (int q = 0; q < S::Q1; ++q) A3[q] = frag[(S::F3 + wo * S::Q1 + q) * 64 + lane];
  }
  NJ_DEV void begin() {}
  NJ_DEV float a1(int q) const { return A1[q]; }
  NJ_DEV float a2(int q) const { return A2[q]; }
  NJ_DEV float a3(int q) const { return A3[q]; }
};
template <class S> struct Q4FwdLds {
  static constexpr int NVEC = S::Q0 + 2 * S::Q1;
  lfp base, cur;
  NJ_DEV void load(lfp region, const float* frag, int w, int lane) {   // region: NVEC * 64 floats
    static_assert(S::MT1 == 4, "four hidden tiles, one per wave");
    base = region + lane;
    cur = base;
    for (int q = 0; q < S::Q0; ++q) base[q * 64] = frag[(S::F1 + w * S::Q0 + q) * 64 + lane];
    for (int q = 0; q < S::Q1; ++q) base[(S::Q0 + q) * 64] = frag[(S::F2 + w * S::Q1 + q) * 64 + lane];
    const int wo = w < S::MTO ? w : 0;
    for (int q = 0; q < S::Q1; ++q)
      base[(S::Q0 + S::Q1 + q) * 64] = frag[(S::F3 + wo * S::Q1 + q) * 64 + lane];
  }
  NJ_DEV void begin() {   // laundered per evaluation: the reads are not hoisted into registers
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int q) const { return cur[q * 64]; }
  NJ_DEV float a2(int q) const { return cur[(S::Q0 + q) * 64]; }
  NJ_DEV float a3(int q) const { return cur[(S::Q0 + S::Q1 + q) * 64]; }
};
// Adjoint sweep: hidden recompute + the three transposed products; input-gradient tiles w,
// w + 4, ... (NB1 of them)
template <class S, int NB1> struct Q4AdjReg {
  float A1[S::Q0], A2[S::Q1], B3[S::QO], B2[S::QW], B1[NB1][S::QW];
  NJ_DEV void load(const float* frag, int w, int lane) {
    static_assert(S::MT1 == 4, "four hidden tiles, one per wave");
#pragma unroll
    for (int q = 0; q < S::Q0; ++q) A1[q] = frag[(S::F1 + w * S::Q0 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < S::Q1; ++q) A2[q] = frag[(S::F2 + w * S::Q1 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < S::QO; ++q) B3[q] = frag[(S::B3 + w * S::QO + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < S::QW; ++q) B2[q] = frag[(S::B2 + w * S::QW + q) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
      const int t = w + 4 * j < S::MTI ? w + 4 * j : 0;
#pragma unroll
      for (int q = 0; q < S::QW; ++q) B1[j][q] = frag[(S::B1 + t * S::QW + q) * 64 + lane];
    }
  }
  NJ_DEV void begin() {}
  NJ_DEV float a1(int q) const { return A1[q]; }
  NJ_DEV float a2(int q) const { return A2[q]; }
  NJ_DEV float b3(int q) const { return B3[q]; }
  NJ_DEV float b2(int q) const { return B2[q]; }
  NJ_DEV float b1(int j, int q) const { return B1[j][q]; }
};
// the transposed products only (the sweep's Euler steps read the forward's activations)
template <class S, int NB1> struct Q4AdjStep {
  float B3[S::QO], B2[S::QW], B1[NB1][S::QW];
  // input-gradient tiles: j = 0 -> tile w, j >= 1 -> tile t1 + w + 4 (j - 1)
  NJ_DEV void load(const float* frag, int w, int lane, int t1) {
    static_assert(S::MT1 == 4, "four hidden tiles, one per wave");
#pragma unroll
    for (int q = 0; q < S::QO; ++q) B3[q] = frag[(S::B3 + w * S::QO + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < S::QW; ++q) B2[q] = frag[(S::B2 + w * S::QW + q) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
      const int tj = j == 0 ? w : t1 + w + 4 * (j - 1);
      const int t = tj < S::MTI ? tj : 0;
#pragma unroll
      for (int q = 0; q < S::QW; ++q) B1[j][q] = frag[(S::B1 + t * S::QW + q) * 64 + lane];
    }
  }
  NJ_DEV float b3(int q) const { return B3[q]; }
  NJ_DEV float b2(int q) const { return B2[q]; }
  NJ_DEV float b1(int j, int q) const { return B1[j][q]; }
};
template <class S> struct Q4AdjLds {   // one input-gradient tile
  static constexpr int O2 = S::Q0, O3 = O2 + S::Q1, O4 = O3 + S::QO, O5 = O4 + S::QW, NVEC = O5 + S::QW;
  lfp base, cur;
  NJ_DEV void load(lfp region, const float* frag, int w, int lane) {
    static_assert(S::MT1 == 4, "four hidden tiles, one per wave");
    base = region + lane;
    cur = base;
    for (int q = 0; q < S::Q0; ++q) base[q * 64] = frag[(S::F1 + w * S::Q0 + q) * 64 + lane];
    for (int q = 0; q < S::Q1; ++q) base[(O2 + q) * 64] = frag[(S::F2 + w * S::Q1 + q) * 64 + lane];
    for (int q = 0; q < S::QO; ++q) base[(O3 + q) * 64] = frag[(S::B3 + w * S::QO + q) * 64 + lane];
    for (int q = 0; q < S::QW; ++q) base[(O4 + q) * 64] = frag[(S::B2 + w * S::QW + q) * 64 + lane];
    const int t = w < S::MTI ? w : 0;
    for (int q = 0; q < S::QW; ++q) base[(O5 + q) * 64] = frag[(S::B1 + t * S::QW + q) * 64 + lane];
  }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int q) const { return cur[q * 64]; }
  NJ_DEV float a2(int q) const { return cur[(O2 + q) * 64]; }
  NJ_DEV float b3(int q) const { return cur[(O3 + q) * 64]; }
  NJ_DEV float b2(int q) const { return cur[(O4 + q) * 64]; }
  NJ_DEV float b1(int, int q) const { return cur[(O5 + q) * 64]; }
};

// sum_q A(q) (x) b[q] on two accumulators (halves the dependent chain)
template <int NQ, class FA> NJ_DEV f32x4 q4_dot(FA fa, const float (&b)[NQ]) {
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NQ; q += 2) {
    s0 = mfma4(fa(q), b[q], s0);
    if (q + 1 < NQ) s1 = mfma4(fa(q + 1), b[q + 1], s1);
  }
  return s0 + s1;
}
// the gathers of a layer input are issued as one batch, before the products that consume them
NJ_DEV void q4_gathered() { __builtin_amdgcn_sched_barrier(0); }

// own registers <-> image rows.  q4_put: all four rows (images of >= 64 rows, invalid units carry
// zeros); q4_put_n: units < N only, at row offset row0
NJ_DEV void q4_put(lfp X, const float (&v)[4], int g, int c, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(16 * w + 4 * r + g) * IMG_STRIDE + c] = v[r];
}
template <int N> NJ_DEV void q4_put_n(lfp X, int row0, const float (&v)[4], int g, int c, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = 16 * w + 4 * r + g;
    if (u < N) X[(row0 + u) * IMG_STRIDE + c] = v[r];
  }
}
template <int N> NJ_DEV void q4_put_n_if(lfp X, int row0, const float (&v)[4], bool on, int g, int c, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = 16 * w + 4 * r + g;
    if (on && u < N) X[(row0 + u) * IMG_STRIDE + c] = v[r];
  }
}
template <int N> NJ_DEV void q4_get_n(lfp X, int row0, float (&v)[4], int g, int c, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = 16 * w + 4 * r + g;
    const float t = X[(row0 + (u < N ? u : 0)) * IMG_STRIDE + c];
    v[r] = u < N ? t : 0.0f;
  }
}

// D-layout register q of a layer input [v (NV units from the image), 1, 0 ...]
template <int NV, int Q> NJ_DEV float q4_bias_patch(float raw, int g) {
  const float e0 = 4 * Q + 0 < NV ? raw : (4 * Q + 0 == NV ? 1.0f : 0.0f);
  const float e1 = 4 * Q + 1 < NV ? raw : (4 * Q + 1 == NV ? 1.0f : 0.0f);
  const float e2 = 4 * Q + 2 < NV ? raw : (4 * Q + 2 == NV ? 1.0f : 0.0f);
  const float e3 = 4 * Q + 3 < NV ? raw : (4 * Q + 3 == NV ? 1.0f : 0.0f);
  return g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
}
template <int NV, int NQ, int Q = 0> NJ_DEV void q4_input(lfp X, float (&b)[NQ], int g, int c) {
  if constexpr (Q < NQ) {
    const float raw = X[(4 * Q + g) * IMG_STRIDE + c];
    if constexpr (4 * Q + 3 < NV) b[Q] = raw;
    else b[Q] = q4_bias_patch<NV, Q>(raw, g);
    q4_input<NV, NQ, Q + 1>(X, b, g, c);
  }
}
// ... of the ODE network: [tanh(h) (H), tanh(x) (D)] from the image, then tau, t - tau, (t), 1
template <class C, int U> NJ_DEV float q4_in0_unit(float raw, float tau, float tdiff) {
  if constexpr (U < C::H + C::D) return raw;
  else if constexpr (U == C::H + C::D) return tau;
  else if constexpr (U == C::H + C::D + 1) return tdiff;
  else if constexpr (C::CURT && U == C::H + C::D + 2) return tau + tdiff;
  else if constexpr (U == C::ODE_IN) return 1.0f;
  else return 0.0f;
}
template <class C, int Q = 0>
NJ_DEV void q4_in0(lfp X, float (&b)[MF<C>::Q0], float tau, float tdiff, int g, int c) {
  if constexpr (Q < MF<C>::Q0) {
    const float raw = X[(4 * Q + g) * IMG_STRIDE + c];
    if constexpr (4 * Q + 3 < C::H + C::D) {
      b[Q] = raw;
    } else {
      const float e0 = q4_in0_unit<C, 4 * Q + 0>(raw, tau, tdiff);
      const float e1 = q4_in0_unit<C, 4 * Q + 1>(raw, tau, tdiff);
      const float e2 = q4_in0_unit<C, 4 * Q + 2>(raw, tau, tdiff);
      const float e3 = q4_in0_unit<C, 4 * Q + 3>(raw, tau, tdiff);
      b[Q] = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
    }
    q4_in0<C, Q + 1>(X, b, tau, tdiff, g, c);
  }
}

// hidden activation of the own tile (+ dropout by the own 4 keep bits); the bias unit W is 1
template <int W, int ACT, bool DROP>
NJ_DEV void q4_hidden(const f32x4& acc, float (&al)[4], uint32_t keep4, float inv_keep, int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = act_f<ACT>(acc[r]);
    // a dropped unit is -0.0f: the same zero to every product, and a mark the adjoint sweep can
    // read back from the stored activation (a kept unit can be exactly +0: zero biases, zero
    // start values; it is never -0, the accumulators start from +0)
    if constexpr (DROP) v = ((keep4 >> r) & 1) ? v * inv_keep : -0.0f;
    al[r] = v;
  }
  if (w == W / 16) al[(W % 16) / 4] = g == W % 4 ? 1.0f : al[(W % 16) / 4];
}
template <int W, int ACT, bool DROP>
NJ_DEV void q4_delta(const f32x4& acc, const float (&al)[4], float (&dl)[4], uint32_t keep4, float inv_keep,
                     float keepf, int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float d;
    if constexpr (DROP) d = ((keep4 >> r) & 1) ? acc[r] * inv_keep * dact_f<ACT>(al[r] * keepf) : 0.0f;
    else d = acc[r] * dact_f<ACT>(al[r]);
    dl[r] = (16 * w + 4 * r + g) < W ? d : 0.0f;   // bias / padding units carry no delta
  }
}

// ... from an activation the forward stored: a dropped unit is -0.0f there (q4_hidden)
template <int W, int ACT, bool DROP>
NJ_DEV void q4_delta_stored(const f32x4& acc, const float (&al)[4], float (&dl)[4], float inv_keep,
                            float keepf, int g, int w) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float d;
    if constexpr (DROP)
      d = __float_as_uint(al[r]) != 0x80000000u ? acc[r] * inv_keep * dact_f<ACT>(al[r] * keepf) : 0.0f;
    else d = acc[r] * dact_f<ACT>(al[r]);
    dl[r] = (16 * w + 4 * r + g) < W ? d : 0.0f;
  }
}

// own 4 keep bits of the two hidden layers, from the streams of the one-wave kernels
template <bool DROP>
NJ_DEV void q4_row_keep(const KArgs& a, unsigned long long gid, uint32_t tkey, uint32_t net, int g, int w,
                        uint32_t& k1, uint32_t& k2) {
  // row_keep_bits (njode_mfma_rows.h) draws 16 bits = 8 words per layer; this wave's bits
  // 4w .. 4w+3 are words 2w, 2w+1 of each layer: the other words are stepped over
  k1 = k2 = 0;
  if constexpr (DROP) {
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1), tkey, net);
    auto skip = [&](int n) {
      for (int i_ = 0; i_ < n; ++i_) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; }
    };
    skip(2 * w);
    k1 = keep_bits<4>(st, a.dc.thr16);
    skip(6);
    k2 = keep_bits<4>(st, a.dc.thr16);
  }
}
template <class C, bool DROP>
NJ_DEV void q4_ode_keep(const KArgs& a, unsigned long long gid, int k, int g, int w, uint32_t& k1, uint32_t& k2) {
  k1 = k2 = 0;
  if constexpr (DROP) {
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1), (uint32_t)k,
                             NET_ODE);
    // unit 4q + g is bit q of the lane group's stream, NWD words per layer (njode_mfma.h): this
    // wave's bits 4w .. 4w+3 are words 2w, 2w+1 of each layer -- skip the others
    constexpr int NWD = (MF<C>::Q1 + 1) / 2;
    auto skip = [&](int n) {
      for (int i_ = 0; i_ < n; ++i_) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; }
    };
    skip(2 * w);
    k1 = keep_bits<4>(st, a.dc.thr16);
    skip(NWD - 2);
    k2 = keep_bits<4>(st, a.dc.thr16);
  }
}

// The keep bits of the masked forward AHEAD of the kernel (round 4; the segment plan's four-wave
// role does the same, njode_mfma_split.h): the forward is ONE chain of thousands of network
// evaluations per tile, and q4_ode_keep / q4_row_keep -- three hash rounds, up to fifteen serial
// xorshift words, the bit assembly -- cost ~0.4 us of every one of them although they depend on
// nothing the chain computes.  k_q4_bits draws them for every (Euler step, tile) and every
// (observation row, evaluation) in parallel over the chip: the whole lane-group stream of the
// one-wave kernels, k1 | k2 << 16 (bit q = register q), of which wave w then reads bits
// 4w .. 4w+3.  Same masks, bit for bit (the sweep never draws masks: it reads the -0.0 marks).
template <class C>
__global__ void __launch_bounds__(256) k_q4_bits(KArgs a, int n_tiles) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  const long long n_ode = (long long)a.K * n_tiles;
  for (long long i = wave; i < n_ode; i += n_waves) {
    const int k = (int)(i / n_tiles), tile = (int)(i % n_tiles);
    const int b0i = tile * a.q4_pt + c;
    const int b = (c < a.q4_pt && b0i < a.B) ? b0i : a.B - 1;
    const unsigned long long gid = a.gid0 + b;
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1), (uint32_t)k,
                             NET_ODE);
    const uint32_t k1 = keep_bits<MF<C>::Q1>(st, a.dc.thr16);
    const uint32_t k2 = keep_bits<MF<C>::Q1>(st, a.dc.thr16);
    a.dbits[(size_t)i * 64 + lane] = k1 | (k2 << 16);
  }
  // rows: one thread per (row, evaluation, lane group)
  const long long n_row = (long long)a.n_obs * 12;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_row; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / 12), e = (int)(i % 12) >> 2, gg = (int)i & 3;
    const unsigned long long gid = a.gid0 + a.obs_idx[row];
    const uint32_t tkey = (uint32_t)a.k_jump[a.t_of_row[row]];
    const uint32_t net = e == 0 ? NET_DEC_BJ : (e == 1 ? NET_ENC : NET_DEC);
    uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (gg + 1), tkey, net);
    const uint32_t k1 = keep_bits<16>(st, a.dc.thr16);
    const uint32_t k2 = keep_bits<16>(st, a.dc.thr16);
    a.dbits_row[i] = k1 | (k2 << 16);
  }
}

// Stored hidden activations of the three evaluations of a jump (readout before = 0, encoder = 1,
// readout after = 2): per observation row and wave 3 x 2 x 4 registers x 4 lane groups
constexpr int Q4_JACT_FLOATS = 3 * 2 * 4 * 4;   // = 96 per row and wave
NJ_DEV float* q4_jact_ptr(float* jact, int row, int w, int set, int g) {
  return jact + (((size_t)row * 4 + w) * 3 + set) * 32 + g;
}

// forward of one network: b0 (all units) -> this wave's output tile.  Two exchanges.
template <class S, int ACT, bool DROP, class FP>
NJ_DEV f32x4 q4_net_fwd(FP& F, lfp XA, lfp XB, const float (&b0)[S::Q0], uint32_t k1, uint32_t k2,
                        float inv_keep, int g, int c, int w) {
  F.begin();
  f32x4 acc = q4_dot<S::Q0>([&](int q) { return F.a1(q); }, b0);
  float al[4], av[S::Q1];
  q4_hidden<S::W, ACT, DROP>(acc, al, k1, inv_keep, g, w);
  q4_put(XA, al, g, c, w);
  block_lds_barrier();
  split_get<S::Q1>(XA, av, g, c);
  q4_gathered();
  acc = q4_dot<S::Q1>([&](int q) { return F.a2(q); }, av);
  q4_hidden<S::W, ACT, DROP>(acc, al, k2, inv_keep, g, w);
  q4_put(XB, al, g, c, w);
  block_lds_barrier();
  split_get<S::Q1>(XB, av, g, c);
  q4_gathered();
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  if (w < S::MTO) out = q4_dot<S::Q1>([&](int q) { return F.a3(q); }, av);
  return out;
}

// ... the same, handing back the own tiles of the two hidden activations
template <class S, int ACT, bool DROP, class FP>
NJ_DEV f32x4 q4_net_fwd_acts(FP& F, lfp XA, lfp XB, const float (&b0)[S::Q0], uint32_t k1, uint32_t k2,
                             float inv_keep, int g, int c, int w, float (&a1l)[4], float (&a2l)[4]) {
  F.begin();
  f32x4 acc = q4_dot<S::Q0>([&](int q) { return F.a1(q); }, b0);
  float av[S::Q1];
  q4_hidden<S::W, ACT, DROP>(acc, a1l, k1, inv_keep, g, w);
  q4_put(XA, a1l, g, c, w);
  block_lds_barrier();
  split_get<S::Q1>(XA, av, g, c);
  q4_gathered();
  acc = q4_dot<S::Q1>([&](int q) { return F.a2(q); }, av);
  q4_hidden<S::W, ACT, DROP>(acc, a2l, k2, inv_keep, g, w);
  q4_put(XB, a2l, g, c, w);
  block_lds_barrier();
  split_get<S::Q1>(XB, av, g, c);
  q4_gathered();
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  if (w < S::MTO) out = q4_dot<S::Q1>([&](int q) { return F.a3(q); }, av);
  return out;
}

// per-path sums of the own units -> all units (lane groups by shuffle, waves through LDS)
NJ_DEV void q4_path_sums(lfp LR, float& sa, float& sb, int g, int c, int w) {
  sa += __shfl_xor(sa, 16);
  sb += __shfl_xor(sb, 16);
  sa += __shfl_xor(sa, 32);
  sb += __shfl_xor(sb, 32);
  if (g == 0) {
    LR[(2 * w + 0) * 16 + c] = sa;
    LR[(2 * w + 1) * 16 + c] = sb;
  }
  block_lds_barrier();
  sa = (LR[0 * 16 + c] + LR[2 * 16 + c]) + (LR[4 * 16 + c] + LR[6 * 16 + c]);
  sb = (LR[1 * 16 + c] + LR[3 * 16 + c]) + (LR[5 * 16 + c] + LR[7 * 16 + c]);
}

// maintainer aid (-DNJ_Q4_STAMP): phase timestamps of one Euler step, printed per wave
#ifdef NJ_Q4_STAMP
#define Q4_STAMP_DECL unsigned long long q4_ts[12]; int q4_nts = 0; (void)q4_ts; (void)q4_nts
#define Q4_STAMP() do { if (q4_on && q4_nts < 12) q4_ts[q4_nts++] = __builtin_readcyclecounter(); } while (0)
#define Q4_STAMP_PRINT(name) do { if (q4_on && lane == 0) { \
    unsigned long long d_[10]; for (int i_ = 0; i_ < 10; ++i_) d_[i_] = i_ + 1 < q4_nts ? q4_ts[i_ + 1] - q4_ts[i_] : 0; \
    printf("%s w%d: %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu | total %llu\n", name, w, d_[0], d_[1], d_[2], d_[3], \
           d_[4], d_[5], d_[6], d_[7], d_[8], d_[9], q4_ts[q4_nts - 1] - q4_ts[0]); } q4_nts = 0; } while (0)
#else
#define Q4_STAMP_DECL
#define Q4_STAMP()
#define Q4_STAMP_PRINT(name)
#endif

template <class C> struct Q4Lds {
  using ES = typename EncS<C>::type;
  static constexpr int INR = q4_img_rows(MF<C>::Q0) > q4_img_rows(ES::Q0) ? q4_img_rows(MF<C>::Q0)
                                                                           : q4_img_rows(ES::Q0);
  static constexpr int IN_FL = INR * IMG_STRIDE;
  // XA, XB, XD, HN (64 rows), IN, EI, LX (INR rows), LR; then the encoder fragments of the 4 waves
  static constexpr int IMAGES = 4 * XFLOATS + 3 * IN_FL + 8 * 16;
  static constexpr int FWD_FLOATS = IMAGES + 4 * Q4FwdLds<ES>::NVEC * 64;
  static constexpr int ADJ_FLOATS = IMAGES + 4 * Q4AdjLds<ES>::NVEC * 64;
  static_assert(ADJ_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

// =====================================================================================
// forward
// =====================================================================================
template <class C, bool DROP>
__global__ void __launch_bounds__(256) k_paths_fwd_q4(KArgs a) {
  using M = MF<C>;
  using OS = OdeQS<C>;
  using ES = typename EncS<C>::type;
  using DS = typename DecS<C>::type;
  using L = Q4Lds<C>;
  constexpr int D = C::D, H = C::H, DO = C::DO;
  __shared__ __attribute__((aligned(16))) float lds_raw[L::FWD_FLOATS];
  lfp XA = (lfp)lds_raw, XB = XA + XFLOATS, HN = XB + 2 * XFLOATS, IN = HN + XFLOATS, EI = IN + L::IN_FL,
      LR = EI + 2 * L::IN_FL;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < L::IMAGES; i += 256) lds_raw[i] = 0.0f;
  Q4FwdReg<OS> Fo;
  Q4FwdLds<ES> Fe;
  Q4FwdReg<DS> Fd;
  Fo.load(a.frag, w, lane);
  Fe.load((lfp)lds_raw + L::IMAGES + w * Q4FwdLds<ES>::NVEC * 64, a.frag_enc, w, lane);
  Fd.load(a.frag_dec, w, lane);
  __syncthreads();

  const bool LOSS = a.want_loss != 0, SAVE = a.save_traj != 0;
  // (round 4: a tile holds a.q4_pt <= 16 paths -- see q4_paths_per_tile, njode_api.hip; the
  // other lanes shadow a valid path and store nothing)
  const int b0i = blockIdx.x * a.q4_pt + c;
  const bool valid = c < a.q4_pt && b0i < a.B;
  const int b = valid ? b0i : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  float* const trash = a.trash + threadIdx.x;
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const cfp sdt = as_cfp(a.step_dt), stt = as_cfp(a.step_t), tf = as_cfp(a.time_f32);
  int uo[4];   // the own units
#pragma unroll
  for (int r = 0; r < 4; ++r) uo[r] = 16 * w + 4 * r + g;

  // readout of the state whose tanh is in image `TH` rows [0, H): this wave's tile of y
  // store the own hidden activations of a jump evaluation for the sweep (set < 0: not a jump)
  auto keep_acts = [&](int set, int row, bool on, const float (&a1l)[4], const float (&a2l)[4]) {
    if (set < 0 || !SAVE) return;
    float* p = on ? q4_jact_ptr(a.jact, row, w, set, g) : trash;
    const int st4 = on ? 4 : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r * st4] = a1l[r];
      p[(4 + r) * st4] = a2l[r];
    }
  };
  // readout of the state whose tanh is in image `TH` rows [0, H): this wave's tile of y
  // (kw: the evaluation's keep bits drawn ahead, k_q4_bits -- used when `ahead`)
  const bool bits_ahead = DROP && a.dbits_ready != 0;   // (wave-uniform)
  auto own_bits = [&](uint32_t kw, uint32_t& k1, uint32_t& k2) {
    k1 = (kw >> (4 * w)) & 15u;
    k2 = (kw >> (16 + 4 * w)) & 15u;
  };
  auto readout = [&](lfp TH, const float (&hq)[4], uint32_t tkey, uint32_t net, float (&y)[4], int set,
                     int row, bool on, uint32_t kw = 0, bool ahead = false) {
    float b0d[DS::Q0], a1l[4], a2l[4];
    q4_input<H, DS::Q0>(TH, b0d, g, c);
    q4_gathered();
    uint32_t k1, k2;
    if (ahead) own_bits(kw, k1, k2);
    else q4_row_keep<DROP>(a, gid, tkey, net, g, w, k1, k2);
    const f32x4 out = q4_net_fwd_acts<DS, C::ACT, DROP>(Fd, XA, XB, b0d, k1, k2, a.dc.inv_keep, g, c, w,
                                                        a1l, a2l);
    keep_acts(set, row, on, a1l, a2l);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = out[r];
      if constexpr (C::DEC_CASE == 1) v += hq[r];
      y[r] = uo[r] < DO ? v : 0.0f;
    }
  };
  // encoder of [tanh(xin), mask] staged in EI: this wave's tile of the new state
  auto encode = [&](const float (&xin)[4], uint32_t tkey, float (&hq)[4], int set, int row, bool on,
                    uint32_t kw = 0, bool ahead = false) {
    float b0e[ES::Q0], a1l[4], a2l[4];
    q4_input<C::ENC_IN, ES::Q0>(EI, b0e, g, c);
    q4_gathered();
    uint32_t k1, k2;
    if (ahead) own_bits(kw, k1, k2);
    else q4_row_keep<DROP>(a, gid, tkey, NET_ENC, g, w, k1, k2);
    const f32x4 out = q4_net_fwd_acts<ES, C::ACT, DROP>(Fe, XA, XB, b0e, k1, k2, a.dc.inv_keep, g, c, w,
                                                        a1l, a2l);
    keep_acts(set, row, on, a1l, a2l);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = out[r];
      if constexpr (C::ENC_CASE == 1) v += xin[r];
      hq[r] = uo[r] < H ? v : 0.0f;
    }
  };

  // ---- initial state: h = encoder(start_X, mask = 0) ----------------------------------------
  float h[4], th[4], txo[4], xs[4], zero4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = a.start_X[(size_t)b * D + (uo[r] < D ? uo[r] : 0)];
    xs[r] = uo[r] < D ? v : 0.0f;
    txo[r] = tanh_f(xs[r]);
  }
  q4_put_n<D>(EI, 0, txo, g, c, w);
  if constexpr (C::MASKED) q4_put_n<D>(EI, D, zero4, g, c, w);
  q4_put_n<D>(IN, H, txo, g, c, w);
  block_lds_barrier();
  encode(xs, TKEY_START, h, -1, 0, false);
#pragma unroll
  for (int r = 0; r < 4; ++r) th[r] = tanh_f(h[r]);
  q4_put_n<H>(IN, 0, th, g, c, w);
  block_lds_barrier();

  // checkpoint / activation store addresses advance by a constant per step (inactive lanes and
  // units keep writing to `trash`)
  float* lt_p[4];
  long long lt_step[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = SAVE && valid && uo[r] < H;
    lt_p[r] = ok ? a.ltraj + (size_t)b * H + uo[r] : trash;
    lt_step[r] = ok ? (long long)a.B * H : 0;
  }
  const int n_tiles = (int)gridDim.x;
  float* la_p = SAVE ? q4_act_ptr(a.lact, 0, n_tiles, blockIdx.x, w, lane) : trash;
  const long long la_step = SAVE ? (long long)n_tiles * Q4_ACT_FLOATS : 0;
  const int la_r = SAVE ? 64 : 0;

  float tau = 0.0f, loss_acc = 0.0f;
  int cur = a.first_j[b];
  // row and time index of this path's next observation; after a jump they are reloaded, and the
  // loaded values are only looked at when the next jump time comes up (p_*: pending reload)
  int r_cur = a.n_obs > 0 ? a.row_by_path[cur >= 0 ? cur : 0] : 0;
  int next_i = a.n_obs > 0 ? a.t_of_row[r_cur] : 0;
  next_i = cur >= 0 ? next_i : 0x7fffffff;
  bool p_on = false;
  int p_row = 0, p_path = 0, p_time = 0;
  int src = -1;

  // the schedule's scalars are loaded one step / one jump ahead (their latency would otherwise
  // sit on the critical path of every step)
  int i = 0;
  int kj = a.n_times > 0 ? kjump[0] : -1;
  float dt_n = a.K > 0 ? sdt[0] : 0.0f, t_n = a.K > 0 ? stt[0] : 0.0f;
  const uint32_t* kb_p = bits_ahead ? a.dbits + (size_t)blockIdx.x * 64 + lane : nullptr;
  const size_t kb_step = (size_t)n_tiles * 64;
  uint32_t kb_n = (bits_ahead && a.K > 0) ? kb_p[0] : 0u;
  for (int k = 0;; ++k) {
    while (i < a.n_times && kj == k) {
      kj = i + 1 < a.n_times ? kjump[i + 1] : -1;
      {   // resolve the pending reload
        const int nx = (cur < a.n_obs && p_path == b) ? p_time : 0x7fffffff;
        next_i = p_on ? nx : next_i;
        r_cur = p_on ? p_row : r_cur;
        p_on = false;
      }
      const bool has = valid && next_i == i;
      if (__any(has)) {   // the same decision in all four waves
        const int r_ = has ? r_cur : 0;
        float ybj[4], yn[4], x[4], m[4], xin[4], txin[4], hn[4], thn[4], xraw[4], mraw[4];
        // the row's values are needed after the first readout: loaded beside it
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int u = uo[r] < D ? uo[r] : 0;
          xraw[r] = a.X[(size_t)r_ * D + u];
          mraw[r] = C::MASKED ? a.M[(size_t)r_ * D + u] : 1.0f;
        }
        uint32_t kwj[3] = {0u, 0u, 0u};
        if (bits_ahead) {
#pragma unroll
          for (int e = 0; e < 3; ++e) kwj[e] = a.dbits_row[((size_t)r_ * 3 + e) * 4 + g];
        }
        // ... and so is the path's next row (its time index is loaded at the commit below)
        const int cn = cur + 1;
        const int cc = cn < a.n_obs ? cn : 0;
        const int n_row = a.row_by_path[cc], n_path = a.path_sorted[cc];
        if (SAVE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = (has && uo[r] < H) ? a.h_end + (size_t)r_ * H + uo[r] : trash;
            *dst = h[r];
          }
        }
        readout(IN, h, (uint32_t)k, NET_DEC_BJ, ybj, 0, r_, has, kwj[0], bits_ahead);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r] = uo[r] < D ? xraw[r] : 0.0f;
          if constexpr (C::MASKED) {
            m[r] = uo[r] < D ? mraw[r] : 0.0f;
            xin[r] = x[r] * m[r] + (1.0f - m[r]) * ybj[r];
          } else {
            m[r] = uo[r] < D ? 1.0f : 0.0f;
            xin[r] = x[r];
          }
          txin[r] = tanh_f(xin[r]);
        }
        q4_put_n<D>(EI, 0, txin, g, c, w);
        if constexpr (C::MASKED) q4_put_n<D>(EI, D, m, g, c, w);
        block_lds_barrier();
        encode(xin, (uint32_t)k, hn, 1, r_, has, kwj[1], bits_ahead);
#pragma unroll
        for (int r = 0; r < 4; ++r) thn[r] = tanh_f(hn[r]);
        q4_put_n<H>(HN, 0, thn, g, c, w);
        block_lds_barrier();
        readout(HN, hn, (uint32_t)k, NET_DEC, yn, 2, r_, has, kwj[2], bits_ahead);
        if (LOSS) {   // compute_loss (models.py:76-110) of this row, reduced over all units
          float sa = 0.0f, sb = 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = x[r] - yn[r];
            const float f = a.loss_easy ? (ybj[r] - x[r]) : (ybj[r] - yn[r]);
            sa = fmaf(m[r] * e, e, sa);
            sb = fmaf(m[r] * f, f, sb);
          }
          q4_path_sums(LR, sa, sb, g, c, w);
          const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
          const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
          const float ca = a.loss_easy ? a.weight : 2.0f * a.weight;
          const float cb = a.loss_easy ? (1.0f - a.weight) : 2.0f * (1.0f - a.weight);
          const float s = ca * na + cb * nb;
          loss_acc += has ? s * s * scale : 0.0f;
        }
        if (SAVE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool on = has && uo[r] < DO;
            float* d1 = on ? a.y_row + (size_t)r_ * DO + uo[r] : trash;
            float* d2 = on ? a.ybj_row + (size_t)r_ * DO + uo[r] : trash;
            *d1 = yn[r];
            *d2 = ybj[r];
          }
        }
        // commit for the paths that have an observation (models.py:463-489)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[r] = has ? hn[r] : h[r];
          th[r] = has ? thn[r] : th[r];
          const float nt = tanh_f(C::MASKED ? yn[r] : x[r]);
          txo[r] = has ? nt : txo[r];
        }
        q4_put_n<H>(IN, 0, th, g, c, w);
        q4_put_n<D>(IN, H, txo, g, c, w);
        block_lds_barrier();
        const float tnew = tf[i];
        tau = has ? tnew : tau;
        src = has ? r_ : src;
        p_row = n_row;
        p_path = n_path;
        p_time = a.t_of_row[n_row];
        p_on = has;
        cur = has ? cn : cur;
        next_i = has ? 0x7fffffff : next_i;   // (until the reload is resolved)
      }
      ++i;
    }
    if (SAVE) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *lt_p[r] = h[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) lt_p[r] += lt_step[r];
      if (valid && w == 0 && g == 0 && k < a.K) a.src_row[(size_t)k * a.B + b] = src;
    }
    if (k >= a.K) break;
    {
      const float dt = dt_n, t = t_n;
      const uint32_t kb = kb_n;
      if (k + 1 < a.K) {
        dt_n = sdt[k + 1];
        t_n = stt[k + 1];
        if (bits_ahead) kb_n = kb_p[(size_t)(k + 1) * kb_step];
      }
#ifdef NJ_Q4_STAMP
      const bool q4_on = blockIdx.x == 0 && k == a.K / 2;
#endif
      Q4_STAMP_DECL;
      Q4_STAMP();
      float b0[M::Q0];
      q4_in0<C>(IN, b0, tau, t - tau, g, c);
      q4_gathered();
      uint32_t k1, k2;
      if (bits_ahead) own_bits(kb, k1, k2);
      else q4_ode_keep<C, DROP>(a, gid, k, g, w, k1, k2);
      float a1l[4], a2l[4];
      const f32x4 out = q4_net_fwd_acts<OS, C::ACT, DROP>(Fo, XA, XB, b0, k1, k2, a.dc.inv_keep, g, c, w,
                                                          a1l, a2l);
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // (an evaluation forward writes them to `trash`)
        la_p[r * la_r] = a1l[r];
        la_p[(4 + r) * la_r] = a2l[r];
      }
      la_p += la_step;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h[r] = uo[r] < H ? fmaf(dt, out[r], h[r]) : 0.0f;
        th[r] = tanh_f(h[r]);
      }
      q4_put_n<H>(IN, 0, th, g, c, w);
      Q4_STAMP();
      block_lds_barrier();
      Q4_STAMP();
      Q4_STAMP_PRINT("fwd");
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* dst = (valid && uo[r] < H) ? a.hT + (size_t)b * H + uo[r] : trash;
    *dst = h[r];
  }
  if (LOSS && valid && w == 0 && g == 0) a.loss_terms[b] = loss_acc;
}

// =====================================================================================
// adjoint sweep (pass 1 of the backward)
// =====================================================================================
template <class C, bool DROP>
__global__ void __launch_bounds__(256) k_paths_bwd_adj_q4(KArgs a) {
  using M = MF<C>;
  using OS = OdeQS<C>;
  using ES = typename EncS<C>::type;
  using DS = typename DecS<C>::type;
  using L = Q4Lds<C>;
  constexpr int D = C::D, H = C::H, DO = C::DO;
  // Input-gradient tiles of the ODE network (in0 units [h (H), x (D), ..]): the h tiles (wave w:
  // tile w) every step; the x tiles T0X .. MTB1-1 (wave w: tile T0X + w) ONCE PER JUMP -- the
  // input x of a segment is constant, so sum_steps W1x^T delta1 = W1x^T (sum_steps delta1):
  // the sweep accumulates delta1 and the product waits for the jump
  constexpr int T0X = H / 16, NB1 = 2;
  static_assert(M::MTB1 - T0X <= 4 && M::MTH <= 4, "one x tile and one h tile per wave");
  static_assert((D + 15) / 16 <= 4 && DS::MTI <= 4, "one input-gradient tile per wave for the row networks");
  __shared__ __attribute__((aligned(16))) float lds_raw[L::IMAGES];
  lfp XA = (lfp)lds_raw, XB = XA + XFLOATS, XD = XB + XFLOATS, HN = XD + XFLOATS, IN = HN + XFLOATS,
      EI = IN + L::IN_FL, LX = EI + L::IN_FL, LR = LX + L::IN_FL;
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < L::IMAGES; i += 256) lds_raw[i] = 0.0f;
  // every evaluation's hidden activations come from the forward (lact: Euler steps, jact: the
  // three evaluations of a jump), so the sweep only holds the transposed-product fragments:
  // 50 + 37 + 37 registers
  Q4AdjStep<OS, NB1> Fo;
  Q4AdjStep<ES, 1> Fe;
  Q4AdjStep<DS, 1> Fd;
  Fo.load(a.frag, w, lane, T0X);
  Fe.load(a.frag_enc, w, lane, 0);
  Fd.load(a.frag_dec, w, lane, 0);
  __syncthreads();

  // (round 4: a tile holds a.q4_pt <= 16 paths -- see q4_paths_per_tile, njode_api.hip; the
  // other lanes shadow a valid path and store nothing)
  const int b0i = blockIdx.x * a.q4_pt + c;
  const bool valid = c < a.q4_pt && b0i < a.B;
  const int b = valid ? b0i : a.B - 1;
  const unsigned long long gid = a.gid0 + b;
  float* const trash = a.trash + threadIdx.x;
  const int __attribute__((address_space(4)))* kjump =
      (const int __attribute__((address_space(4)))*)(unsigned long long)a.k_jump;
  const cfp sdt = as_cfp(a.step_dt);
  int uo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) uo[r] = 16 * w + 4 * r + g;

  float lam[4] = {0.f, 0.f, 0.f, 0.f};   // adjoint of h, own units
  if (a.g_hT) {                          // (uniform) ... starting from the upstream gradient of hT
#pragma unroll
    for (int r = 0; r < 4; ++r)
      lam[r] = (valid && uo[r] < H) ? a.g_hT[(size_t)b * H + (uo[r] < H ? uo[r] : 0)] : 0.0f;
  }
  float d1acc[4] = {0.f, 0.f, 0.f, 0.f};   // sum of delta1 over the steps of the segment, own units
  // The row this path reverses next (`src`), what the sweep needs of it (its time index, its
  // predecessor row, h before the jump, X, M, y, y_bj: own units), loaded when `src` is assigned --
  // under the lanes' exec mask, so nothing waits for these loads before the jump that uses them.
  int src = a.last_row[b], src_i = -1, src_pp = -1;
  float Rhp[4] = {0.f, 0.f, 0.f, 0.f}, Rx[4] = {0.f, 0.f, 0.f, 0.f}, Rm[4] = {0.f, 0.f, 0.f, 0.f},
        Ry[4] = {0.f, 0.f, 0.f, 0.f}, Rybj[4] = {0.f, 0.f, 0.f, 0.f}, sx0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sx0[r] = a.start_X[(size_t)b * D + (uo[r] < D ? uo[r] : 0)];
  auto load_rows = [&](int srow, bool on) {
    if (on && srow >= 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uh = uo[r] < H ? uo[r] : 0, ud = uo[r] < D ? uo[r] : 0;
        Rhp[r] = a.h_end[(size_t)srow * H + uh];
        Rx[r] = a.X[(size_t)srow * D + ud];
        if constexpr (C::MASKED) Rm[r] = a.M[(size_t)srow * D + ud];
        Ry[r] = a.y_row[(size_t)srow * DO + ud];
        Rybj[r] = a.ybj_row[(size_t)srow * DO + ud];
      }
      src_pp = a.item_prev[srow];
      src_i = a.t_of_row[srow];
    }
    if (on && srow < 0) src_i = -1;
  };
  load_rows(src, true);
  // last_X of the segment that follows row `src` (its prediction; the start value before the
  // first row) -> image rows H + u; refreshed lazily inside the next Euler step
  bool tx_dirty = true;
  auto put_source = [&]() {
    float tx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) tx[r] = tanh_f(src >= 0 ? (C::MASKED ? Ry[r] : Rx[r]) : sx0[r]);
    q4_put_n<D>(IN, H, tx, g, c, w);
  };

  // adjoint of y = readout(hq) w.r.t. hq: own tile of dh from the own tile of dy; a1l / a2l: the
  // evaluation's own hidden activations as the forward stored them
  auto dec_adj = [&](const float (&hq)[4], const float (&dy)[4], const float (&a1l)[4],
                     const float (&a2l)[4], float (&dh)[4]) {
    float thq[4], dl[4], dq[DS::QO], dv[DS::QW];
#pragma unroll
    for (int r = 0; r < 4; ++r) thq[r] = tanh_f(hq[r]);
    q4_put(XD, dy, g, c, w);
    block_lds_barrier();
    split_get<DS::QO>(XD, dq, g, c);
    q4_gathered();
    f32x4 acc = q4_dot<DS::QO>([&](int q) { return Fd.b3(q); }, dq);
    q4_delta_stored<DS::W, C::ACT, DROP>(acc, a2l, dl, a.dc.inv_keep, a.keep, g, w);
    q4_put(XB, dl, g, c, w);
    block_lds_barrier();
    split_get<DS::QW>(XB, dv, g, c);
    q4_gathered();
    acc = q4_dot<DS::QW>([&](int q) { return Fd.b2(q); }, dv);
    q4_delta_stored<DS::W, C::ACT, DROP>(acc, a1l, dl, a.dc.inv_keep, a.keep, g, w);
    q4_put(XA, dl, g, c, w);
    block_lds_barrier();
    split_get<DS::QW>(XA, dv, g, c);
    q4_gathered();
    f32x4 din = {0.f, 0.f, 0.f, 0.f};
    if (w < DS::MTI) din = q4_dot<DS::QW>([&](int q) { return Fd.b1(0, q); }, dv);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = din[r] * (1.0f - thq[r] * thq[r]);
      if constexpr (C::DEC_CASE == 1) v += dy[r];
      dh[r] = uo[r] < H ? v : 0.0f;
    }
  };

  // state before step k and the forward's hidden activations, loaded one step ahead (raw: the
  // selects happen at the use, so that nothing waits for the loads where they are issued)
  const int n_tiles = (int)gridDim.x;
  float h_cur[4] = {0.f, 0.f, 0.f, 0.f};
  const float* lt_p[4];
  float* lm_p[4];
  long long lt_step[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = valid && uo[r] < H;
    const size_t off = (size_t)b * H + (uo[r] < H ? uo[r] : 0);
    lt_p[r] = a.ltraj + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B) * H + off;
    lm_p[r] = ok ? a.lam_traj + ((size_t)(a.K > 0 ? a.K - 1 : 0) * a.B) * H + off : trash;
    lt_step[r] = ok ? (long long)a.B * H : 0;
  }
  const long long lt_back = (long long)a.B * H;
  const float* la_p = q4_act_ptr(a.lact, a.K > 0 ? a.K - 1 : 0, n_tiles, blockIdx.x, w, lane);
  const long long la_step = (long long)n_tiles * Q4_ACT_FLOATS;
  // Two steps ahead (a step is shorter than a trip to HBM): step k consumes set k & 1 and
  // refills it with the data of step k - 2; the set index is a compile-time constant of the two
  // instances of the step body, so there are no register copies that would wait for the loads.
  float hb[2][4], a1b[2][4], a2b[2][4];
  auto fetch = [&](auto SET) {
    constexpr int S_ = decltype(SET)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hb[S_][r] = *lt_p[r];
      a1b[S_][r] = la_p[r * 64];
      a2b[S_][r] = la_p[(4 + r) * 64];
      lt_p[r] -= lt_back;
    }
    la_p -= la_step;
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  if (a.K > 0) {   // step K - 1, then step K - 2
    if ((a.K - 1) & 1) fetch(Set1{}); else fetch(Set0{});
  }
  if (a.K > 1) {
    if ((a.K - 2) & 1) fetch(Set1{}); else fetch(Set0{});
  }

  int i = a.n_times - 1;
  int kj = i >= 0 ? kjump[i] : -1;   // the schedule's scalars: one step / one jump ahead
  float dt_n = a.K > 0 ? sdt[a.K - 1] : 0.0f;
  for (int k = a.K; k >= 0; --k) {
    // ---- reverse Euler step k (hidden activations from the forward: no recompute)
    auto euler_step = [&](auto SET) {
      constexpr int S_ = decltype(SET)::value;
      float th[4], d3[4], a1l[4], a2l[4], dl[4], dq[M::QH], dv[M::QW];
      const float dt = dt_n;
      if (k > 0) dt_n = sdt[k - 1];
#ifdef NJ_Q4_STAMP
      const bool q4_on = blockIdx.x == 0 && k == a.K / 2;
#endif
      Q4_STAMP_DECL;
      Q4_STAMP();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h_cur[r] = hb[S_][r];
        th[r] = uo[r] < H ? tanh_f(hb[S_][r]) : 0.0f;
        a1l[r] = a1b[S_][r];
        a2l[r] = a2b[S_][r];
        d3[r] = dt * lam[r];
      }
      if (k > 1) fetch(SET);
      q4_put(XD, d3, g, c, w);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        *lm_p[r] = lam[r];
        lm_p[r] -= lt_step[r];
      }
      Q4_STAMP();
      block_lds_barrier();                                   // (1) delta3 of all tiles
      Q4_STAMP();
      split_get<M::QH>(XD, dq, g, c);
      q4_gathered();
      f32x4 acc = q4_dot<M::QH>([&](int q) { return Fo.b3(q); }, dq);
#ifdef NJ_Q4_CHECK
      {
        uint32_t c1, c2;
        q4_ode_keep<C, DROP>(a, gid, k, g, w, c1, c2);
        for (int r = 0; r < 4; ++r) {
          const int u = 16 * w + 4 * r + g;
          if (DROP && valid && u < C::W) {
            const bool k1b = (c1 >> r) & 1, k2b = (c2 >> r) & 1;
            if (k1b != (__float_as_uint(a1l[r]) != 0x80000000u) || k2b != (__float_as_uint(a2l[r]) != 0x80000000u))
              printf("mismatch k=%d b=%d u=%d keep=(%d,%d) a=(%g,%g)\n", k, b, u, (int)k1b, (int)k2b, a1l[r], a2l[r]);
          }
        }
      }
#endif
      q4_delta_stored<C::W, C::ACT, DROP>(acc, a2l, dl, a.dc.inv_keep, a.keep, g, w);
      q4_put(XB, dl, g, c, w);
      Q4_STAMP();
      block_lds_barrier();                                   // (2) delta2
      Q4_STAMP();
      split_get<M::QW>(XB, dv, g, c);
      q4_gathered();
      acc = q4_dot<M::QW>([&](int q) { return Fo.b2(q); }, dv);
      q4_delta_stored<C::W, C::ACT, DROP>(acc, a1l, dl, a.dc.inv_keep, a.keep, g, w);
      if constexpr (C::MASKED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d1acc[r] += dl[r];
      }
      q4_put(XA, dl, g, c, w);
      if (tx_dirty) {   // (wave-uniform) the segment's last_X changed at the jump just reversed
        put_source();
        tx_dirty = false;
      }
      Q4_STAMP();
      block_lds_barrier();                                   // (3) delta1, last_X
      Q4_STAMP();
      split_get<M::QW>(XA, dv, g, c);
      q4_gathered();
      // d/dh of the tanh'd state input: tile w for w < MTH
      if (w < M::MTH) {
        const f32x4 din = q4_dot<M::QW>([&](int q) { return Fo.b1(0, q); }, dv);
#pragma unroll
        for (int r = 0; r < 4; ++r) lam[r] += uo[r] < H ? din[r] * (1.0f - th[r] * th[r]) : 0.0f;
      }
      Q4_STAMP();
      Q4_STAMP_PRINT("bwd");
    };
    if (k < a.K) {
      if (k & 1) euler_step(Set1{}); else euler_step(Set0{});
    }
    // ---- reverse the jump applied right before step k
    while (i >= 0 && kj == k) {
      kj = i > 0 ? kjump[i - 1] : -1;
      const bool has = valid && src >= 0 && src_i == i;
      if (__any(has)) {
        const int r_ = has ? src : 0;
        float hn[4], hp[4], x[4], m[4], y[4], ybj[4], dy[4], dybj[4], dh[4], lam_hn[4], lam_new[4];
        float ja[3][2][4];   // hidden activations of the jump's three evaluations (forward's)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const float* p = q4_jact_ptr(a.jact, r_, w, e, g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ja[e][0][r] = p[r * 4];
            ja[e][1][r] = p[(4 + r) * 4];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // the state after the jump = the state before step k: in registers unless k == K
          float v0 = h_cur[r];
          if (k >= a.K) v0 = a.ltraj[((size_t)k * a.B + b) * H + (uo[r] < H ? uo[r] : 0)];
          hn[r] = uo[r] < H ? v0 : 0.0f;
          hp[r] = uo[r] < H ? Rhp[r] : 0.0f;
          x[r] = uo[r] < D ? Rx[r] : 0.0f;
          y[r] = uo[r] < D ? Ry[r] : 0.0f;
          ybj[r] = uo[r] < D ? Rybj[r] : 0.0f;
          m[r] = uo[r] < D ? (C::MASKED ? Rm[r] : 1.0f) : 0.0f;
        }
        if constexpr (C::MASKED) {
          // gradient w.r.t. the segment's input x (the prediction at this row): W1x^T applied to
          // the accumulated delta1, times tanh'; staged by in0 unit for the gather below
          float dacc[M::QW];
          q4_put(XD, d1acc, g, c, w);   // (XD: nobody reads it between the last barrier and here)
          block_lds_barrier();
          split_get<M::QW>(XD, dacc, g, c);
          q4_gathered();
          if (T0X + w < M::MTB1) {
            const f32x4 din = q4_dot<M::QW>([&](int q) { return Fo.b1(1, q); }, dacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = 16 * (T0X + w) + 4 * r + g;
              const float tv = IN[(u >= H && u < H + D ? u : H) * IMG_STRIDE + c];
              LX[u * IMG_STRIDE + c] = (u >= H && u < H + D) ? din[r] * (1.0f - tv * tv) : 0.0f;
            }
          }
        }
        // gradient of compute_loss at this row
        {
          float sa = 0.0f, sb = 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = x[r] - y[r];
            const float f = a.loss_easy ? (ybj[r] - x[r]) : (ybj[r] - y[r]);
            sa = fmaf(m[r] * e, e, sa);
            sb = fmaf(m[r] * f, f, sb);
          }
          q4_path_sums(LR, sa, sb, g, c, w);   // (its barrier also publishes LX)
          const float scale = a.inv_batch * __builtin_amdgcn_rcpf((float)a.n_obs_ot[b]);
          const float na = sqrtf(sa + 1e-10f), nb = sqrtf(sb + 1e-10f);
          const float ca = a.loss_easy ? a.weight : 2.0f * a.weight;
          const float cb = a.loss_easy ? (1.0f - a.weight) : 2.0f * (1.0f - a.weight);
          const float s = ca * na + cb * nb;
          const float gg = 2.0f * s * scale;
          const float ga = gg * ca / na, gb = gg * cb / nb;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = x[r] - y[r];
            if (a.loss_easy) {
              const float f = ybj[r] - x[r];
              dy[r] = -ga * m[r] * e;
              dybj[r] = gb * m[r] * f;
            } else {
              const float f = ybj[r] - y[r];
              dy[r] = -ga * m[r] * e - gb * m[r] * f;
              dybj[r] = gb * m[r] * f;
            }
          }
        }
        if constexpr (C::MASKED) {   // last_X <- Y: the later segment's input gradient
          float lxo[4];
          q4_get_n<D>(LX, H, lxo, g, c, w);
#pragma unroll
          for (int r = 0; r < 4; ++r) dy[r] += lxo[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = (has && uo[r] < DO) ? a.g_y + (size_t)r_ * DO + uo[r] : trash;
          *dst = dy[r];
        }
        dec_adj(hn, dy, ja[2][0], ja[2][1], dh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lam_hn[r] = lam[r] + dh[r];
          float* dst = (has && uo[r] < H) ? a.g_hnew + (size_t)r_ * H + uo[r] : trash;
          *dst = lam_hn[r];
        }
        if constexpr (C::MASKED) {
          // h_new = encoder(x_in, M), x_in = X M + (1 - M) y_bj
          float txin[4], dl[4], dq[ES::QO], dv[ES::QW];
#pragma unroll
          for (int r = 0; r < 4; ++r) txin[r] = tanh_f(x[r] * m[r] + (1.0f - m[r]) * ybj[r]);
          q4_put(XD, lam_hn, g, c, w);
          block_lds_barrier();
          split_get<ES::QO>(XD, dq, g, c);
          q4_gathered();
          f32x4 acc = q4_dot<ES::QO>([&](int q) { return Fe.b3(q); }, dq);
          q4_delta_stored<ES::W, C::ACT, DROP>(acc, ja[1][1], dl, a.dc.inv_keep, a.keep, g, w);
          q4_put(XB, dl, g, c, w);
          block_lds_barrier();
          split_get<ES::QW>(XB, dv, g, c);
          q4_gathered();
          acc = q4_dot<ES::QW>([&](int q) { return Fe.b2(q); }, dv);
          q4_delta_stored<ES::W, C::ACT, DROP>(acc, ja[1][0], dl, a.dc.inv_keep, a.keep, g, w);
          q4_put(XA, dl, g, c, w);
          block_lds_barrier();
          split_get<ES::QW>(XA, dv, g, c);
          q4_gathered();
          f32x4 din = {0.f, 0.f, 0.f, 0.f};
          if (w < (D + 15) / 16) din = q4_dot<ES::QW>([&](int q) { return Fe.b1(0, q); }, dv);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = din[r] * (1.0f - txin[r] * txin[r]);
            if constexpr (C::ENC_CASE == 1) v += lam_hn[r];
            const float dx = uo[r] < D ? v : 0.0f;
            dybj[r] += dx * (1.0f - m[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = (has && uo[r] < DO) ? a.g_ybj + (size_t)r_ * DO + uo[r] : trash;
          *dst = dybj[r];
        }
        dec_adj(hp, dybj, ja[0][0], ja[0][1], lam_new);
        // commit for the paths that have this observation
#pragma unroll
        for (int r = 0; r < 4; ++r) lam[r] = has ? lam_new[r] : lam[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) d1acc[r] = has ? 0.0f : d1acc[r];
        const int src2 = has ? src_pp : src;
        load_rows(src2, has);
        src = src2;
        tx_dirty = true;
      }
      --i;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* dst = (valid && uo[r] < H) ? a.g_hstart + (size_t)b * H + uo[r] : trash;
    *dst = lam[r];
  }
}

}  // namespace njode
