// njode_mfma.h -- ODE-evolve kernels on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Why: with one chain per lane every Euler step is ~4 500 dependent VALU instructions
// fed by ~210 scalar weight loads; a single wave needs ~19 us per step and the longest
// segments (60-70 steps) set the kernel time no matter how many paths run beside them
// (profiles/r01_v1_smem_variants.jsonl: B=100 costs as much as B=20 000).  On the matrix
// cores one wave advances 16 chains with 81 MFMAs per step, the weights stay resident in
// VGPRs as A-fragments (no weight traffic at all inside the time loop) and a step takes
// ~3-4 k cycles.  f32 MFMA is an exact k-ordered fmaf chain, so results keep fp32 parity.
//
// Layout ("D-layout"): a vector of units over the wave's 16 chains is held as registers
// v[q]; lane l = (g = l >> 4, c = l & 15) holds unit 4q + g of chain c.  That is what
// MFMA tile mt returns in its 4 accumulator registers (q = 4 mt + r) when A's row i is
// assigned to unit 16 mt + 4 (i & 3) + (i >> 2), and it is exactly the B operand of
// k-step q of the next layer (B[k = g][j = c] = unit 4q + g), so activations flow from
// layer to layer without leaving their registers.  Unit IN of every layer input is the
// constant 1 (bias column).
#pragma once
#include "njode_kernels.h"

namespace njode {

typedef float f32x4 __attribute__((ext_vector_type(4)));

NJ_DEV f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- edge rows (round 4) ------------------------------------------------------------------
// A layer of W units takes ceil(W / 16) output tiles of 16x16x4; with W % 16 = 1 or 2 (W = 50:
// units 48, 49) the last tile costs a full tile's 32 cycles per k-step for one or two rows.
// Those rows go through v_mfma_f32_4x4x1_16b_f32 instead: 16 independent blocks of 4x4x1,
// 16 cycles (measured: 15 - 18).  With the B operand in D-layout (lane (g, c): unit 4q + g of chain c) block
// 4g + c / 4 sees input unit 4q + g of chains 4 (c / 4) .. + 3, the A operand of lane (g, c) is
// W[edge row c % 4][4q + g], and after the Q k-steps register i of lane (g, c) holds the part of
// out[edge row i][chain c] that runs over the input units = g (mod 4).  edge_reduce adds the four
// parts of rows 0 and 1 (two lane swaps, two adds): lane groups 0 / 2 get row 0's total, 1 / 3
// row 1's -- register 0 of the tile in D-layout once the groups >= R are cleared.
// (tools/ubench/mfma4x4_check.hip checks the layout, the reduction and the issue cost.)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
NJ_DEV f32x4 mfma1(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
NJ_DEV float edge_reduce(float p0, float p1) {
  const u32x2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
  const float t = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  const u32x2 w = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return __uint_as_float(w[0]) + __uint_as_float(w[1]);
}
template <int W> struct EdgeRows {                 // of a layer with W output units
  static constexpr int R = (W % 16 == 1 || W % 16 == 2) ? W % 16 : 0;
  static constexpr int TILE = R ? W / 16 : -1;     // the tile the edge rows stand for
};
// the lane whose A-fragment value of the edge tile is this lane's 4x4x1 operand: row c % 4 of the
// tile is fragment row i with row_unit(mt, i) = 16 mt + i, i.e. i = 4 (c % 4); same lane group
NJ_DEV int edge_lane(int lane) { return (lane & 48) | ((lane & 3) << 2); }
template <int R> NJ_DEV f32x4 edge_tile(const f32x4& part, int g) {
  const float x = edge_reduce(part[0], part[1]);
  return f32x4{g < R ? x : 0.0f, 0.0f, 0.0f, 0.0f};
}

// Fragment tables of the ODE network (NH == 2)
template <class C> struct MF {
  static constexpr int H = C::H, D = C::D, W = C::W, IN0 = C::ODE_IN;
  static constexpr int Q0 = (IN0 + 1 + 3) / 4;  // k-steps of layer 1 (inputs + bias unit)
  static constexpr int Q1 = (W + 1 + 3) / 4;    // k-steps of layers 2, 3
  static constexpr int QH = (H + 3) / 4, QW = (W + 3) / 4;
  static constexpr int MT1 = (W + 15) / 16, MTH = (H + 15) / 16;
  static constexpr int F1 = 0;                  // W1   [MT1][Q0]
  static constexpr int F2 = F1 + MT1 * Q0;      // W2   [MT1][Q1]
  static constexpr int F3 = F2 + MT1 * Q1;      // W3   [MTH][Q1]
  static constexpr int NFWD = F3 + MTH * Q1;
  static constexpr int B3 = NFWD;               // W3^T [MT1][QH]
  static constexpr int B2 = B3 + MT1 * QH;      // W2^T [MT1][QW]
  // W1^T rows: the h inputs; masked models also need d/d x (the prediction is fed back)
  static constexpr int MTB1 = C::MASKED ? (H + D + 15) / 16 : MTH;
  static constexpr int B1 = B2 + MT1 * QW;      // W1^T [MTB1][QW]
  static constexpr int NALL = B1 + MTB1 * QW;
  // in0 unit order: [h (H), x (D), tau, tdiff, (tau + tdiff), 1]; reference column of unit u
  static constexpr int col0(int u) { return u < H ? D + u : (u < H + D ? u - H : u); }
  static_assert(C::NH == 2, "MFMA ODE kernels are written for two hidden layers");
};

// Persistent waves walk the length-sorted tiles in boustrophedon ("snake") order: round i
// gives wave w tile i*G + w, the next round i*G + (G-1-w), so every wave gets long and
// short tiles alike (static, hence deterministic, yet balanced).
NJ_DEV int snake_tile(int round, int wave, int n_waves) {
  return round * n_waves + ((round & 1) ? (n_waves - 1 - wave) : wave);
}

NJ_DEV int row_unit(int mt, int i) { return 16 * mt + 4 * (i & 3) + (i >> 2); }

// Build the A-fragments of all six products from the flat parameter vector.
template <class C> __global__ void k_pack_frags(const float* __restrict__ P, float* __restrict__ frag) {
  using M = MF<C>;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M::NALL * 64) return;
  const int f = idx >> 6, l = idx & 63, g = l >> 4, c = l & 15;
  const float* Po = P + C::OFF_ODE;
  using NL = typename C::Ode;
  const float *W1 = Po + NL::woff(0), *b1 = Po + NL::boff(0), *W2 = Po + NL::woff(1),
              *b2 = Po + NL::boff(1), *W3 = Po + NL::woff(2), *b3 = Po + NL::boff(2);
  float v = 0.0f;
  if (f < M::F2) {            // W1: out unit = row, in unit = 4q + g
    const int mt = (f - M::F1) / M::Q0, q = (f - M::F1) % M::Q0;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::W) v = ui < M::IN0 ? W1[uo * M::IN0 + M::col0(ui)] : (ui == M::IN0 ? b1[uo] : 0.0f);
  } else if (f < M::F3) {     // W2
    const int mt = (f - M::F2) / M::Q1, q = (f - M::F2) % M::Q1;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::W) v = ui < M::W ? W2[uo * M::W + ui] : (ui == M::W ? b2[uo] : 0.0f);
  } else if (f < M::NFWD) {   // W3
    const int mt = (f - M::F3) / M::Q1, q = (f - M::F3) % M::Q1;
    const int uo = row_unit(mt, c), ui = 4 * q + g;
    if (uo < M::H) v = ui < M::W ? W3[uo * M::W + ui] : (ui == M::W ? b3[uo] : 0.0f);
  } else if (f < M::B2) {     // W3^T: row = hidden-2 unit, k = output unit
    const int mt = (f - M::B3) / M::QH, q = (f - M::B3) % M::QH;
    const int ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < M::W && uo < M::H) v = W3[uo * M::W + ui];
  } else if (f < M::B1) {     // W2^T
    const int mt = (f - M::B2) / M::QW, q = (f - M::B2) % M::QW;
    const int ui = row_unit(mt, c), uo = 4 * q + g;
    if (ui < M::W && uo < M::W) v = W2[uo * M::W + ui];
  } else {                    // W1^T: row = in0 unit (only the h rows are consumed)
    const int mt = (f - M::B1) / M::QW, q = (f - M::B1) % M::QW;
    const int u = row_unit(mt, c), uo = 4 * q + g;
    if (u < M::IN0 && uo < M::W) v = W1[uo * M::IN0 + M::col0(u)];
  }
  frag[idx] = v;
}

// value of in0 unit u for this lane's chain (compile-time u)
template <class C, int U>
NJ_DEV float in0_unit(const float th_q, const float (&tx)[C::D], float tau, float tdiff) {
  if constexpr (U < C::H) return th_q;
  else if constexpr (U < C::H + C::D) return tx[U - C::H];
  else if constexpr (U == C::H + C::D) return tau;
  else if constexpr (U == C::H + C::D + 1) return tdiff;
  else if constexpr (C::CURT && U == C::H + C::D + 2) return tau + tdiff;
  else if constexpr (U == C::ODE_IN) return 1.0f;
  else return 0.0f;
}
template <class C, int Q>
NJ_DEV void in0_fill(float (&b0)[MF<C>::Q0], const float (&h)[MF<C>::QH], const float (&tx)[C::D],
                     float tau, float tdiff, int g) {
  if constexpr (Q < MF<C>::Q0) {
    float th = 0.0f;
    if constexpr (4 * Q < C::H) th = tanh_f(h[Q]);
    const float e0 = in0_unit<C, 4 * Q + 0>(th, tx, tau, tdiff);
    const float e1 = in0_unit<C, 4 * Q + 1>(th, tx, tau, tdiff);
    const float e2 = in0_unit<C, 4 * Q + 2>(th, tx, tau, tdiff);
    const float e3 = in0_unit<C, 4 * Q + 3>(th, tx, tau, tdiff);
    b0[Q] = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
    in0_fill<C, Q + 1>(b0, h, tx, tau, tdiff, g);
  }
}

// keep-bits for this lane's NQ units of one hidden layer (lane streams differ per g)
template <int NQ> NJ_DEV uint32_t keep_bits(uint32_t& s, uint32_t thr16) {
  uint32_t m = 0;
#pragma unroll
  for (int u = 0; u < NQ; u += 2) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    m |= (uint32_t)((s & 0xffffu) >= thr16) << u;
    if (u + 1 < NQ) m |= (uint32_t)((s >> 16) >= thr16) << (u + 1);
  }
  return m;
}

// ---- the same streams as 64-bit LANE masks: the wave-per-path / wave-per-item kernels (njode_chain.h,
// njode_chain_seg.h) ----
// keep masks drawn ahead: [(b K + k) 2 + layer] for the Euler steps (path-major: four consecutive steps
// of a path share a 64-byte line of the scalar cache), then per observation row
// [(row 3 + evaluation) 2 + layer] (evaluation: 0 readout before, 1 encoder, 2 readout after), then
// per path [(3 n_obs + b) 2 + layer] for the start encoder
NJ_DEV size_t chain_step_bits(int k, int K, int b) { return ((size_t)b * K + k) * 2; }
NJ_DEV size_t chain_row_bits(int row, int e) { return ((size_t)row * 3 + e) * 2; }

// The lane-group streams of the matrix-core kernels (njode_mfma.h: unit 4 q + g is bit q of the
// stream of lane group g) as 64-bit LANE masks of the DPP layout (njode_dpp.h: lane 16 g + q holds
// unit 4 q + g): bit l = keep decision of the unit lane l holds -- the four streams' words side by side.
template <int NQ>
NJ_DEV void chain_masks(const DropCtx& dc, unsigned long long gid, uint32_t tkey, uint32_t net, uint64_t& m1,
                        uint64_t& m2) {
  static_assert(NQ <= 16, "16 units per lane group");
  m1 = m2 = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint32_t st = drop_state(dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1), tkey, net);
    const uint32_t k1 = keep_bits<NQ>(st, dc.thr16);
    const uint32_t k2 = keep_bits<NQ>(st, dc.thr16);
    m1 |= (uint64_t)k1 << (16 * g);
    m2 |= (uint64_t)k2 << (16 * g);
  }
}

// keep masks of the ODE network for every (path, Euler step), drawn ahead of the wave-per-item forward
// (worker of n_workers, 256 threads each)
template <class C> NJ_DEV void seg_chain_bits_body(const KArgs& a, int worker, int n_workers) {
  uint64_t* sb = (uint64_t*)a.dbits;
  const long long n_ode = (long long)a.K * a.B;
  for (long long i = (long long)worker * 256 + threadIdx.x; i < n_ode; i += (long long)n_workers * 256) {
    const int b = (int)(i / a.K), k = (int)(i % a.K);
    uint64_t m1, m2;
    chain_masks<MF<C>::Q1>(a.dc, a.gid0 + b, (uint32_t)k, NET_ODE, m1, m2);
    sb[i * 2] = m1;
    sb[i * 2 + 1] = m2;
  }
}

// hidden activation from accumulator tiles: a[q] = act(acc[q / 4][q % 4]) (+dropout),
// then the bias unit (unit W) is set to 1
template <int MT1, int Q1, int W, int ACT, bool DROP>
NJ_DEV void hidden_from_acc_g(const f32x4 (&acc)[MT1], float (&av)[Q1], uint32_t keep,
                              float inv_keep, int g) {
#pragma unroll
  for (int q = 0; q < Q1; ++q) {
    float v = act_f<ACT>(acc[q / 4][q % 4]);
    if constexpr (DROP) v = ((keep >> q) & 1) ? v * inv_keep : 0.0f;
    av[q] = v;
  }
  constexpr int QB = W / 4, GB = W % 4;
  av[QB] = g == GB ? 1.0f : av[QB];
}
template <class C, bool DROP>
NJ_DEV void hidden_from_acc(const f32x4 (&acc)[MF<C>::MT1], float (&av)[MF<C>::Q1], uint32_t keep,
                            float inv_keep, int g) {
  hidden_from_acc_g<MF<C>::MT1, MF<C>::Q1, MF<C>::W, C::ACT, DROP>(acc, av, keep, inv_keep, g);
}

// Forward-only form: draws the layer's keep decisions from the stream and applies them at once
// (the same words, in the same order, as keep_bits<Q1> + hidden_from_acc: identical masks),
// without assembling and re-testing a bit mask.
template <class C, bool DROP>
NJ_DEV void hidden_from_acc_draw(const f32x4 (&acc)[MF<C>::MT1], float (&av)[MF<C>::Q1], uint32_t& s,
                                 uint32_t thr16, float inv_keep, int g) {
  constexpr int Q1 = MF<C>::Q1;
#pragma unroll
  for (int q = 0; q < Q1; q += 2) {
    float v0 = act_f<C::ACT>(acc[q / 4][q % 4]);
    float v1 = q + 1 < Q1 ? act_f<C::ACT>(acc[(q + 1) / 4][(q + 1) % 4]) : 0.0f;
    if constexpr (DROP) {
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      v0 = (s & 0xffffu) >= thr16 ? v0 * inv_keep : 0.0f;
      v1 = (s >> 16) >= thr16 ? v1 * inv_keep : 0.0f;
    }
    av[q] = v0;
    if (q + 1 < Q1) av[q + 1] = v1;
  }
  constexpr int QB = MF<C>::W / 4, GB = MF<C>::W % 4;
  av[QB] = g == GB ? 1.0f : av[QB];
}

// B (MFMA): Euler evolve of every item; 16 items per wave, persistent over tiles.
// worker `wave` of `n_waves` walks the tiles [tile0, tile1) in snake order
template <class C, bool DROP, bool TAIL>
NJ_DEV void ode_fwd_single(const KArgs& a, int lane, int wave, int n_waves, int tile0, int tile1) {
  using M = MF<C>;
  const int g = lane >> 4, c = lane & 15;
  float A1[M::MT1][M::Q0], A2[M::MT1][M::Q1], A3[M::MTH][M::Q1];
#pragma unroll
  for (int mt = 0; mt < M::MT1; ++mt) {
#pragma unroll
    for (int q = 0; q < M::Q0; ++q) A1[mt][q] = a.frag[(M::F1 + mt * M::Q0 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < M::Q1; ++q) A2[mt][q] = a.frag[(M::F2 + mt * M::Q1 + q) * 64 + lane];
  }
#pragma unroll
  for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
    for (int q = 0; q < M::Q1; ++q) A3[mt][q] = a.frag[(M::F3 + mt * M::Q1 + q) * 64 + lane];

  const bool SAVE = !TAIL && a.save_traj != 0;
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
    const float* h0 = it.h0(a);
    float h[M::QH];
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      h[q] = u < C::H ? h0[u < C::H ? u : 0] : 0.0f;
    }
    const int nmax = wave_max(it.n);
    // per-step scalars are loaded one step ahead (their latency would otherwise sit on
    // the critical path of every step of a lone wave)
    float dt_n = 0.0f, t_n = 0.0f;
    long long base_n = 0;
    if (nmax > 0) {
      const int k0 = it.n > 0 ? it.kbeg : 0;
      dt_n = it.n > 0 ? a.step_dt[k0] : 0.0f;
      t_n = a.step_t[k0];
      base_n = SAVE ? a.base_s[0] : 0;
    }
    for (int s = 0; s < nmax; ++s) {
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      const float dt = dt_n, t = t_n;
      const long long base = base_n;
      if (s + 1 < nmax) {
        const bool act_n = s + 1 < it.n;
        const int kn = act_n ? it.kbeg + s + 1 : 0;
        dt_n = act_n ? a.step_dt[kn] : 0.0f;
        t_n = a.step_t[kn];
        if (SAVE) base_n = a.base_s[s + 1];
      }
      if (SAVE) {
        float* rec = active ? a.traj + (size_t)(base + j) * C::H : trash;
#pragma unroll
        for (int q = 0; q < M::QH; ++q) {
          const int u = 4 * q + g;
          float* dst = u < C::H ? rec + u : trash;
          *dst = h[q];
        }
      }
      float b0[M::Q0];
      in0_fill<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g);
      uint32_t st = 0;
      if constexpr (DROP) {
        const unsigned long long gid = a.gid0 + it.b;
        st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                        (uint32_t)k, NET_ODE);
      }
      f32x4 acc[M::MT1];
      float a1[M::Q1], a2[M::Q1];
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < M::Q0; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(A1[mt][q], b0[q], acc[mt]);
      hidden_from_acc_draw<C, DROP>(acc, a1, st, a.dc.thr16, a.dc.inv_keep, g);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(A2[mt][q], a1[q], acc[mt]);
      hidden_from_acc_draw<C, DROP>(acc, a2, st, a.dc.thr16, a.dc.inv_keep, g);
      f32x4 acch[M::MTH];
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = mfma4(A3[mt][q], a2[q], acch[mt]);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = fmaf(dt, acch[q / 4][q % 4], h[q]);  // dt = 0: inactive
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
template <class C, bool DROP, bool TAIL>
__global__ void __launch_bounds__(64) k_ode_fwd_mfma(KArgs a) {
  const int n_items = TAIL ? a.B : a.n_obs;
  ode_fwd_single<C, DROP, TAIL>(a, threadIdx.x, blockIdx.x, gridDim.x, 0, (n_items + 15) / 16);
}


// ---- backward ----------------------------------------------------------------------------
// dW of one layer on the matrix cores: dW[uo][ui] = sum_chain delta[uo][chain] * act[ui][chain]
// with K = the wave's 16 chains.  Both operands need the chain index on the MFMA k axis,
// i.e. the transpose of the D-layout, so each vector is staged once in a wave-private LDS
// image img[unit][chain] (row stride 20 floats: 16-B aligned rows, <= 2-way write
// conflicts) and read back as [row = lane & 15][chains 4g .. 4g+3] with one ds_read_b128
// that carries all four k-steps (k-step s of lane group g is chain 4g + s).
constexpr int IMG_STRIDE = 20;
constexpr int IMG_ROWS = 64;
constexpr int IMG_FLOATS = IMG_ROWS * IMG_STRIDE;

// Image layouts.  ImgPad: rows of 16 chains padded to 20 floats (every user but the ones below).
// ImgSwz (round 5): rows of 16 floats, the four 16-byte chain quads of row r stored at quad
// position q ^ swz(r), swz(r) = (bit 3 of r) << 1 | (bit 4 of r).  By the LDS bank rules of gfx950
// (MI355X_MICROARCH.md, LDS: ds_read_b128 is served in the four 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, +32; banks = dword address mod 64; ds_write_b32 in two 32-lane halves, banks
// mod 32) this layout is conflict-free for every access of the weight-gradient products:
//   * img_write (lane (g, c) writes row 4q + g, chain c): within a half g in {0, 1} resp. {2, 3},
//     bank = 16 (g & 1) + (c ^ const): a permutation of the 32 banks;
//   * operand reads of dw_accumulate (lane (g, c) reads quad g of row 16 mt + c): 16-byte slot
//     = 4 (c & 3) + (g ^ swz): within a lane group the (c >> 2, g) pairs {(0,0),(3,0),(1,1),(2,1)}
//     get the four distinct positions {0, 2, 1, 3};
//   * the row-per-lane reads of dw_accumulate_edge (lane l reads quad s of row l): slot
//     4 (l & 3) + (s ^ swz(l)); the rows l, l + 12, l + 20, l + 24 of one group have four
//     different swz values;
// where the padded layout has 3 two-way collisions per 16-lane group of the operand reads
// (slot = (5 row + g) mod 16) and two-way conflicts on every write (profiles/r05_lds_conflicts.txt).
struct ImgPad {
  static constexpr int STRIDE = IMG_STRIDE, FLOATS = IMG_FLOATS;
  static NJ_DEV int elem(int row, int chain) { return row * STRIDE + chain; }
  static NJ_DEV int quad(int row, int q) { return row * STRIDE + 4 * q; }
};
struct ImgSwz {
  static constexpr int STRIDE = 16, FLOATS = IMG_ROWS * 16;
  static NJ_DEV int swz(int row) { return ((row >> 2) & 2) | ((row >> 4) & 1); }
  static NJ_DEV int elem(int row, int chain) { return row * STRIDE + (chain ^ (swz(row) << 2)); }
  static NJ_DEV int quad(int row, int q) { return row * STRIDE + 4 * (q ^ swz(row)); }
};

template <int NQ, class IL = ImgPad> NJ_DEV void img_write(lfp img, const float (&v)[NQ], int g, int c) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) img[IL::elem(4 * q + g, c)] = v[q];
}
template <int MT, int NT, class IL = ImgPad>
NJ_DEV void dw_accumulate(lfp img_d, lfp img_a, f32x4 (&G)[MT][NT], int g, int c) {
  f4 af[MT], bf[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) af[mt] = *(lf4p)(img_d + IL::quad(16 * mt + c, g));
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bf[nt] = *(lf4p)(img_a + IL::quad(16 * nt + c, g));
  // k-step outermost: back-to-back MFMAs never depend on each other
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) G[mt][nt] = mfma4(af[mt][s], bf[nt][s], G[mt][nt]);
}
// dW of a W x (W + 1) layer with edge rows (ET = W / 16 whole tiles each way): the ET x ET tiles
// on 16x16x4 as dw_accumulate; the rows u >= 16 ET against EVERY column (GM: register i of lane l
// = dW[16 ET + i][l]) and the columns v >= 16 ET against the rows (GN: register i of lane l =
// dW[l][16 ET + i]; its lanes >= 16 ET repeat GM's corner and are not flushed) as 4x4x1 outer
// products, one chain per instruction: 32 x 16 cycles where the two padded tile strips took
// (2 ET + 1) x 4 x 32.
template <int ET, class IL = ImgPad>
NJ_DEV void dw_accumulate_edge(lfp img_d, lfp img_a, f32x4 (&G)[ET][ET], f32x4 (&GM)[2], f32x4 (&GN)[2],
                               int lane, int g, int c) {
  f4 af[ET], bf[ET];
#pragma unroll
  for (int mt = 0; mt < ET; ++mt) af[mt] = *(lf4p)(img_d + IL::quad(16 * mt + c, g));
#pragma unroll
  for (int nt = 0; nt < ET; ++nt) bf[nt] = *(lf4p)(img_a + IL::quad(16 * nt + c, g));
  const int er = 16 * ET + (lane & 3);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const f4 x = *(lf4p)(img_d + IL::quad(er, s)), y = *(lf4p)(img_a + IL::quad(lane, s));
    const f4 u = *(lf4p)(img_a + IL::quad(er, s)), v = *(lf4p)(img_d + IL::quad(lane, s));
#pragma unroll
    for (int mt = 0; mt < ET; ++mt)
#pragma unroll
      for (int nt = 0; nt < ET; ++nt) G[mt][nt] = mfma4(af[mt][s], bf[nt][s], G[mt][nt]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      GM[k & 1] = mfma1(x[k], y[k], GM[k & 1]);
      GN[k & 1] = mfma1(u[k], v[k], GN[k & 1]);
    }
  }
}

// delta of a hidden layer from the transposed product and the saved activation
template <int MT1, int Q1, int QW, int ACT, bool DROP>
NJ_DEV void hidden_delta_g(const f32x4 (&acc)[MT1], const float (&av)[Q1], float (&dv)[QW],
                           uint32_t keep, float inv_keep, float keepf) {
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const float gsum = acc[q / 4][q % 4];
    if constexpr (DROP) {
      // av = act / keep:  (1 / keep) (1 - (av keep)^2) = 1 / keep - keep av^2   (one fma)
      float f;
      if constexpr (ACT == ACT_TANH) f = fmaf(av[q] * av[q], -keepf, inv_keep);
      else f = inv_keep * dact_f<ACT>(av[q] * keepf);
      dv[q] = ((keep >> q) & 1) ? gsum * f : 0.0f;
    } else {
      dv[q] = gsum * dact_f<ACT>(av[q]);
    }
  }
}
template <class C, bool DROP>
NJ_DEV void hidden_delta(const f32x4 (&acc)[MF<C>::MT1], const float (&av)[MF<C>::Q1],
                         float (&dv)[MF<C>::QW], uint32_t keep, float inv_keep, float keepf) {
  hidden_delta_g<MF<C>::MT1, MF<C>::Q1, MF<C>::QW, C::ACT, DROP>(acc, av, dv, keep, inv_keep, keepf);
}

// C (MFMA): reverse Euler sweep of every segment, d loss / d ODE params.
// 256-thread blocks: the A-fragments the sweep needs (W1, W2 and the three transposed
// products; W3 itself is not needed) live once per block in LDS, so a wave stays under
// 256 registers and two waves share a SIMD -- one wave's tanh / LDS phases hide behind
// the other's MFMAs.  Fragment reads are laundered per step so they are not hoisted
// back into registers.
template <class C> struct OdeLdsFrags {
  using M = MF<C>;
  static constexpr int SKIP = M::NFWD - M::F3;          // W3 fragments are not staged
  static constexpr int NVEC = M::NALL - SKIP;
  lfp base, cur;
  static NJ_DEV void stage(lfp img, const float* frag, int tid, int nthreads) {
    for (int i = tid; i < M::F3 * 64; i += nthreads) img[i] = frag[i];
    for (int i = tid; i < (M::NALL - M::NFWD) * 64; i += nthreads)
      img[M::F3 * 64 + i] = frag[M::NFWD * 64 + i];
  }
  NJ_DEV void init(lfp img, int lane) { base = img + lane; cur = base; }
  NJ_DEV void begin() {
    unsigned v = (unsigned)(unsigned long long)base;
    asm volatile("" : "+v"(v));
    cur = (lfp)(unsigned long long)v;
  }
  NJ_DEV float a1(int mt, int q) const { return cur[(M::F1 + mt * M::Q0 + q) * 64]; }
  NJ_DEV float a2(int mt, int q) const { return cur[(M::F2 + mt * M::Q1 + q) * 64]; }
  NJ_DEV float b3(int mt, int q) const { return cur[(M::B3 - SKIP + mt * M::QH + q) * 64]; }
  NJ_DEV float b2(int mt, int q) const { return cur[(M::B2 - SKIP + mt * M::QW + q) * 64]; }
  NJ_DEV float b1(int mt, int q) const { return cur[(M::B1 - SKIP + mt * M::QW + q) * 64]; }
};

template <class C> struct OdeBwdSingleLds {
  static constexpr int FLOATS = 4 * 2 * IMG_FLOATS + OdeLdsFrags<C>::NVEC * 64;
};
// one 256-thread block = 4 independent workers; worker `wave` of `n_waves` walks the tiles
// [tile0, tile1) in snake order; the block's workers sum their gradient tiles into slab row
// `slab_row`
template <class C, bool DROP>
NJ_DEV void ode_bwd_single(const KArgs& a, lfp lds_raw, int wave, int n_waves, int tile0, int tile1,
                           int slab_row) {
  using M = MF<C>;
  using NL = typename C::Ode;
  using FR = OdeLdsFrags<C>;
  constexpr int NT1 = (M::W + 1 + 15) / 16;     // column tiles of [a, 1]
  constexpr int NT0 = (M::IN0 + 1 + 15) / 16;   // column tiles of [in0, 1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  lfp img_d = lds_raw + wv * 2 * IMG_FLOATS, img_a = img_d + IMG_FLOATS;
  lfp fimg = lds_raw + 4 * 2 * IMG_FLOATS;
  FR::stage(fimg, a.frag, threadIdx.x, 256);
  // image rows that no vector writes must be finite (they meet zero deltas / feed
  // accumulator entries that are never flushed)
  for (int i = threadIdx.x; i < 4 * 2 * IMG_FLOATS; i += 256) lds_raw[i] = 0.0f;
  __syncthreads();
  FR F;
  F.init(fimg, lane);

  f32x4 G3[M::MTH][NT1], G2[M::MT1][NT1], G1[M::MT1][NT0];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < M::MTH; ++i)
#pragma unroll
    for (int n = 0; n < NT1; ++n) G3[i][n] = zero4;
#pragma unroll
  for (int i = 0; i < M::MT1; ++i) {
#pragma unroll
    for (int n = 0; n < NT1; ++n) G2[i][n] = zero4;
#pragma unroll
    for (int n = 0; n < NT0; ++n) G1[i][n] = zero4;
  }
  float* const trash = a.trash + threadIdx.x * C::H;
  const int n_tiles = tile1 - tile0;
  for (int round = 0; round * n_waves < n_tiles; ++round) {
    const int rel = snake_tile(round, wave, n_waves);
    if (rel >= n_tiles) continue;
    const int tile = tile0 + rel;
    const int j = tile * 16 + c;
    const bool valid = j < a.n_obs;
    Item<C> it;
    it.template load<false>(a, j, valid);
    float lam[M::QH];
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      const float v = a.lam_end[(size_t)it.r * C::H + (u < C::H ? u : 0)];
      lam[q] = (valid && u < C::H) ? v : 0.0f;
    }
    const int nmax = wave_max(it.n);
    // state and scalars of a step are loaded while the previous one is processed
    auto fetch = [&](int s, float (&hh)[M::QH], float& dtt, float& tt) {
      const bool act = s < it.n;
      const int kk = act ? it.kbeg + s : 0;
      const float* rec = a.traj + (act ? (size_t)(a.base_s[s] + j) * C::H : 0);
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const int u = 4 * q + g;
        const float v = rec[u < C::H ? u : 0];
        hh[q] = u < C::H ? v : 0.0f;
      }
      dtt = act ? a.step_dt[kk] : 0.0f;
      tt = a.step_t[kk];
    };
    float h_n[M::QH], dt_n = 0.0f, t_n = 0.0f;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) h_n[q] = 0.0f;
    if (nmax > 0) fetch(nmax - 1, h_n, dt_n, t_n);
    for (int s = nmax - 1; s >= 0; --s) {
      const bool active = s < it.n;
      const int k = active ? it.kbeg + s : 0;
      float h[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) h[q] = h_n[q];
      const float dt = dt_n, t = t_n;
      if (s > 0) fetch(s - 1, h_n, dt_n, t_n);
      float b0[M::Q0];
      in0_fill<C, 0>(b0, h, it.tx, it.tau, t - it.tau, g);
      uint32_t k1 = 0, k2 = 0;
      if constexpr (DROP) {
        const unsigned long long gid = a.gid0 + it.b;
        uint32_t st = drop_state(a.dc, (uint32_t)gid, (uint32_t)(gid >> 32) + 0x5bd1e995u * (g + 1),
                                 (uint32_t)k, NET_ODE);
        k1 = keep_bits<M::Q1>(st, a.dc.thr16);
        k2 = keep_bits<M::Q1>(st, a.dc.thr16);
      }
      // ---- recompute the two hidden layers
      F.begin();
      f32x4 acc[M::MT1];
      float a1[M::Q1], a2[M::Q1];
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::Q0; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.a1(mt, q), b0[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a1, k1, a.dc.inv_keep, g);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::Q1; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.a2(mt, q), a1[q], acc[mt]);
      hidden_from_acc<C, DROP>(acc, a2, k2, a.dc.inv_keep, g);

      // ---- layer 3: h' = h + dt f  =>  delta3 = dt * lam (zero for inactive chains)
      float d3[M::QH];
#pragma unroll
      for (int q = 0; q < M::QH; ++q) d3[q] = dt * lam[q];
      img_write<M::QH>(img_d, d3, g, c);
      img_write<M::Q1>(img_a, a2, g, c);
      wave_lds_sync();
      dw_accumulate<M::MTH, NT1>(img_d, img_a, G3, g, c);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::QH; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.b3(mt, q), d3[q], acc[mt]);
      float d2[M::QW];
      hidden_delta<C, DROP>(acc, a2, d2, k2, a.dc.inv_keep, a.keep);
      wave_lds_sync();

      // ---- layer 2
      img_write<M::QW>(img_d, d2, g, c);
      img_write<M::Q1>(img_a, a1, g, c);
      wave_lds_sync();
      dw_accumulate<M::MT1, NT1>(img_d, img_a, G2, g, c);
#pragma unroll
      for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MT1; ++mt) acc[mt] = mfma4(F.b2(mt, q), d2[q], acc[mt]);
      float d1[M::QW];
      hidden_delta<C, DROP>(acc, a1, d1, k1, a.dc.inv_keep, a.keep);
      wave_lds_sync();

      // ---- layer 1
      img_write<M::QW>(img_d, d1, g, c);
      img_write<M::Q0>(img_a, b0, g, c);
      wave_lds_sync();
      dw_accumulate<M::MT1, NT0>(img_d, img_a, G1, g, c);
      f32x4 acch[M::MTH];
#pragma unroll
      for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = zero4;
#pragma unroll
      for (int q = 0; q < M::QW; ++q)
#pragma unroll
        for (int mt = 0; mt < M::MTH; ++mt) acch[mt] = mfma4(F.b1(mt, q), d1[q], acch[mt]);
      // adjoint of the state: lam += (W1^T delta1)[h rows] * (1 - tanh(h)^2)
#pragma unroll
      for (int q = 0; q < M::QH; ++q) {
        const float th = b0[q];  // = tanh(h) wherever unit 4q + g < H
        const float dth = (4 * q + g) < C::H ? 1.0f - th * th : 0.0f;
        lam[q] = fmaf(acch[q / 4][q % 4], dth, lam[q]);
      }
      wave_lds_sync();
    }
    float* out = valid ? a.lam_start + (size_t)it.r * C::H : trash;
#pragma unroll
    for (int q = 0; q < M::QH; ++q) {
      const int u = 4 * q + g;
      float* dst = u < C::H ? out + u : trash;
      *dst = lam[q];
    }
  }

  // ---- flush: the block's four workers share ONE slab row (parameter layout), so the
  // reduction kernels read one row per block, not per wave.  Waves 1-3 park their register
  // tiles in LDS (free by now), wave 0 adds them in fixed order (deterministic) and stores.
  constexpr int NG = M::MTH * NT1 + M::MT1 * NT1 + M::MT1 * NT0;
  static_assert(3 * NG * 64 * 4 <= OdeBwdSingleLds<C>::FLOATS, "tile reduction does not fit the LDS");
  __syncthreads();
  f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds_raw;
  auto for_tiles = [&](auto f) {
    int i = 0;
#pragma unroll
    for (int mt = 0; mt < M::MTH; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) f(G3[mt][nt], i++);
#pragma unroll
    for (int mt = 0; mt < M::MT1; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) f(G2[mt][nt], i++);
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
  float* slab = a.slab + (size_t)slab_row * C::P + C::OFF_ODE;
  float *W1 = slab + NL::woff(0), *b1 = slab + NL::boff(0), *W2 = slab + NL::woff(1),
        *b2 = slab + NL::boff(1), *W3 = slab + NL::woff(2), *b3 = slab + NL::boff(2);
#pragma unroll
  for (int mt = 0; mt < M::MT1; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int uo = 16 * mt + 4 * g + r;
      if (uo < M::W) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          const int ui = 16 * nt + c;
          if (ui < M::W) W2[uo * M::W + ui] = G2[mt][nt][r];
          else if (ui == M::W) b2[uo] = G2[mt][nt][r];
        }
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
          if (ui < M::W) W3[uo * M::W + ui] = G3[mt][nt][r];
          else if (ui == M::W) b3[uo] = G3[mt][nt][r];
        }
      }
    }
}
template <class C, bool DROP>
__global__ void __launch_bounds__(256, 2) k_ode_bwd_mfma(KArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_raw[OdeBwdSingleLds<C>::FLOATS];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  ode_bwd_single<C, DROP>(a, (lfp)lds_raw, wave, gridDim.x * 4, 0, (a.n_obs + 15) / 16, blockIdx.x);
}

}  // namespace njode
